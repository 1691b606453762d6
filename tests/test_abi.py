"""The C-ABI shared library: loads without a GPU, exports every symbol include/mtgs_rast.h declares,
and validates arguments on the host before any launch (no compute calls here)."""
import ctypes as C
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def header_symbols():
    text = (ROOT / "include" / "mtgs_rast.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mtgs_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_operator_table():
    syms = header_symbols()
    for must in ("mtgs_sh_fwd", "mtgs_sh_bwd", "mtgs_project_fwd", "mtgs_project_bwd", "mtgs_isect_count",
                 "mtgs_isect_scan", "mtgs_isect_emit", "mtgs_sort_pairs", "mtgs_isect_offsets", "mtgs_blend_fwd",
                 "mtgs_blend_bwd", "mtgs_tile_schedule", "mtgs_bin_compact", "mtgs_bin_emit", "mtgs_sort_pairs_u32", "mtgs_bin_sort_tiles", "mtgs_bin_build", "mtgs_dp_pack", "mtgs_dp_accumulate", "mtgs_dp_pack_ordered", "mtgs_node_fwd", "mtgs_node_bwd", "mtgs_node_fwd_batch", "mtgs_node_bwd_batch", "mtgs_densify_stats", "mtgs_densify_stats_batch", "mtgs_normals_fwd", "mtgs_normals_bwd", "mtgs_head_fwd", "mtgs_head_bwd", "mtgs_oob_fwd", "mtgs_oob_bwd", "mtgs_ncc_fwd", "mtgs_ncc_bwd", "mtgs_tv_fwd", "mtgs_tv_bwd", "mtgs_ssim_workspace_floats", "mtgs_ssim_fwd", "mtgs_ssim_bwd", "mtgs_l1_workspace_floats", "mtgs_l1_fwd", "mtgs_l1_bwd", "mtgs_dp_reduce", "mtgs_rast_version", "mtgs_rast_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol(hip_lib):
    from mtgs_amd import _lib
    syms = header_symbols()
    assert sorted(_lib.EXPORTS) == syms, "python binding table and header disagree"
    raw = C.CDLL(str(_lib.LIB_PATH))
    for s in syms:
        assert hasattr(raw, s), f"{s} not exported by libmtgs_rast.so"
    assert hip_lib.mtgs_rast_version() == _lib.ABI_VERSION == 28
    assert hip_lib.mtgs_rast_hot_version() == _lib.HOT_ABI_VERSION


def test_host_side_argument_validation(hip_lib):
    """Bad arguments are rejected before anything is launched, with a message."""
    from mtgs_amd import _lib
    n = C.c_size_t(0)
    assert hip_lib.mtgs_sort_workspace_bytes(-1, C.byref(n)) == 1
    assert b"mtgs_sort_workspace_bytes" in hip_lib.mtgs_rast_last_error()
    assert hip_lib.mtgs_scan_workspace_bytes(1 << 20, C.byref(n)) == 0 and n.value >= 8 * (1 << 20) // 2048
    # SH degree 5 / K too small
    assert hip_lib.mtgs_sh_fwd(10, 16, 5, None, None, None, None, None) == 1
    assert hip_lib.mtgs_sh_fwd(10, 4, 3, None, None, None, None, None) == 1
    with pytest.raises(RuntimeError, match="degree"):
        _lib.call("mtgs_sh_fwd", 10, 4, 3, None, None, None, None, None)
    # null pointers
    assert hip_lib.mtgs_project_fwd(1, 10, None, None, None, None, None, 64, 64, 0.3, 0.01, 1e10, 0.0,
                                    None, None, None, None, None, None, None, 0, 0, 0, None, None) == 1
    # tile size other than 16 and unsupported channel counts are refused by name
    assert hip_lib.mtgs_blend_fwd(1, 10, 3, None, None, None, None, None, None, 0, 64, 64, 8, 8, 8, None, None, 0,
                                  None, None, None, None, None) == 4
    assert b"tile_size" in hip_lib.mtgs_rast_last_error()
    assert hip_lib.mtgs_blend_fwd(1, 10, 9, None, None, None, None, None, None, 0, 64, 64, 16, 4, 4, None, None, 0,
                                  None, None, None, None, None) == 4
    # empty problems are a no-op success
    assert hip_lib.mtgs_sh_fwd(0, 16, 3, None, None, None, None, None) == 0
    assert hip_lib.mtgs_sort_pairs(0, 46, None, None, None, None, None, 0, None) == 0


def test_descriptor_tables_match_the_c_structs(hip_lib):
    """The numpy record layouts the Python layer fills (mtgs_amd.nodes._DESC, mtgs_amd.densify._STATS_DESC) have the size
    of the C structs they are uploaded as (include/mtgs_rast.h: mtgs_node_desc, mtgs_stats_desc), and the field offsets
    follow the declaration order with natural alignment."""
    import re
    from mtgs_amd import densify, nodes
    assert hip_lib.mtgs_node_desc_bytes() == nodes._DESC.itemsize == 320
    assert hip_lib.mtgs_stats_desc_bytes() == densify._STATS_DESC.itemsize == 48
    from mtgs_amd import loss
    assert hip_lib.mtgs_oob_desc_bytes() == loss._OOB_DESC.itemsize == 64
    from mtgs_amd import optim
    assert hip_lib.mtgs_adam_group_bytes() == optim._GROUP.itemsize == 232
    header = (ROOT / "include" / "mtgs_rast.h").read_text()

    def c_fields(name):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), header, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        out = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                out.append(re.sub(r"[^A-Za-z0-9_]", " ", re.sub(r"\[\d+\]", "", part)).split()[-1])
        return out

    assert c_fields("mtgs_node_desc") == list(nodes._DESC.names)
    assert c_fields("mtgs_stats_desc") == list(densify._STATS_DESC.names)
    assert c_fields("mtgs_oob_desc") == list(loss._OOB_DESC.names)
    assert c_fields("mtgs_adam_group") == list(optim._GROUP.names)


def test_header_is_plain_c_and_struct_sizes_match(tmp_path):
    """include/mtgs_rast.h compiles as C99 (no C++-isms, no torch types) and the descriptor structs have the sizes the
    Python layer uploads."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("gcc not installed")
    src = tmp_path / "h.c"
    src.write_text('#include "mtgs_rast.h"\n'
                   'int main(void) { return sizeof(mtgs_node_desc) == 320 && sizeof(mtgs_stats_desc) == 48 && '
                   'sizeof(mtgs_oob_desc) == 64 && sizeof(mtgs_adam_group) == 232 ? 0 : 1; }\n')
    exe = tmp_path / "h"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", f"-I{ROOT / 'include'}", str(src), "-o", str(exe)])
    assert subprocess.call([str(exe)]) == 0
