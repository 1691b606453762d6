import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def hip_lib():
    """Builds (if hipcc is present and the .so is stale/missing) and loads libmtgs_rast.so."""
    from mtgs_amd import _lib, build
    try:
        build.build()          # compile / link failures propagate: never test a stale binary
    except build.HipccNotFound:
        # a box without the compiler runs the prebuilt in-tree library -- which must exist and be current
        if not _lib.LIB_PATH.exists():
            raise
        stale = build.stale_sources()
        if stale:
            raise RuntimeError(f"libmtgs_rast.so is older than {[p.name for p in stale]} and hipcc is missing")
    return _lib.load()


@pytest.fixture(params=["tight", "gsplat"])
def lists_mode(request):
    """Both forms of the fused path's tile lists: gsplat's own (the default) and the tight ones (`with mtgs_amd.tight_lists():`)."""
    import mtgs_amd
    with mtgs_amd.tight_lists(request.param == "tight"):
        yield request.param


def pytest_sessionfinish(session, exitstatus):
    """Parity report: the measured errors of every image / gradient comparison of this session (tests/util.py REPORT)."""
    try:
        from tests import util
    except Exception:
        return
    if not util.REPORT:
        return
    import json
    out = ROOT / "gpurun_out"
    out.mkdir(exist_ok=True)
    with open(out / "parity_report.json", "w") as f:
        json.dump(util.REPORT, f, indent=0)
