"""INTEGRATION.md section 2 shows the reference-side binding a gsplat maintainer would add.  This test extracts that
code block and RUNS it against the built library, so the document cannot drift from include/mtgs_rast.h again:
without a GPU through the host-side argument validation (every pointer NULL -> MTGS_EINVAL with a message, which also
proves the 25 arguments marshal), with a GPU as a real projection compared with the product wrapper."""
import re
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]


def _doc_namespace(hip_lib):
    from mtgs_amd import _lib
    text = (ROOT / "INTEGRATION.md").read_text()
    m = re.search(r"```python\n(# gsplat/cuda/_backend_rocm\.py.*?)```", text, re.S)
    assert m, "INTEGRATION.md: the reference-side binding block is missing"
    ns = {"MTGS_RAST_SO": str(_lib.LIB_PATH)}
    exec(compile(m.group(1), "INTEGRATION.md:_backend_rocm", "exec"), ns)
    return ns


def _header_arg_count(name):
    text = (ROOT / "include" / "mtgs_rast.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    m = re.search(r"\bint\s+%s\s*\((.*?)\)\s*;" % name, text, re.S)
    return len([a for a in m.group(1).split(",") if a.strip()])


def test_doc_binding_matches_the_header(hip_lib):
    ns = _doc_namespace(hip_lib)
    assert len(ns["_lib"].mtgs_project_fwd.argtypes) == _header_arg_count("mtgs_project_fwd") == 25


def test_doc_binding_runs_argument_validation_on_cpu(hip_lib):
    """CPU tensors have no device pointers the library could use; the documented stub is driven with EMPTY inputs
    (N = 0 is a valid no-op) and with NULL pointers for N > 0 (rejected by name before any launch)."""
    ns = _doc_namespace(hip_lib)
    f = ns["fully_fused_projection_fwd"]
    vm, K = torch.eye(4)[None], torch.eye(3)[None]
    radii, means2d, depths, conics, comps = f(torch.zeros(0, 3), None, torch.zeros(0, 4), torch.zeros(0, 3), vm, K, 64, 48,
                                              0.3, 0.01, 1e10, 0.0, True, "pinhole")
    assert radii.shape == (1, 0) and comps.shape == (1, 0)

    class _Null:                      # stands for a tensor whose storage the library must refuse
        def __init__(self, *shape):
            self.shape, self.device = shape, torch.device("cpu")

        def data_ptr(self):
            return None

    with pytest.raises(RuntimeError, match="mtgs_project_fwd"):
        f(_Null(5, 3), None, _Null(5, 4), _Null(5, 3), vm, K, 64, 48, 0.3, 0.01, 1e10, 0.0, False, "pinhole")


@pytest.mark.gpu
def test_doc_binding_equals_the_product_wrapper(hip_lib):
    from mtgs_amd import wrapper
    from tests.util import small_scene
    ns = _doc_namespace(hip_lib)
    sc, vm, K = small_scene(N=500, W=96, H=64)
    dev = torch.device("cuda")
    a = {k: v.to(dev) for k, v in sc.items()}
    got = ns["fully_fused_projection_fwd"](a["means"], None, a["quats"], a["scales"], vm.to(dev), K.to(dev), 96, 64, 0.3, 0.01,
                                           1e10, 0.0, True, "pinhole")
    ref = wrapper.fully_fused_projection(a["means"], None, a["quats"], a["scales"], vm.to(dev), K.to(dev), 96, 64,
                                         calc_compensations=True)
    torch.cuda.synchronize()
    vis = ref[0] > 0
    assert torch.equal(got[0], ref[0]) and int(vis.sum()) > 50
    for g, r in zip(got[1:], ref[1:]):
        assert torch.equal(g[vis], r[vis])
