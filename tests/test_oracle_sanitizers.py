"""The C oracle under AddressSanitizer + UBSan (CPU only; GPU sanitizers are not available on the pool).

The known-answer tests and the oracle-vs-torch tests are re-run in a child process against the sanitizer build of
oracle/gsplat_oracle.c: an out-of-bounds tile index, a read past a ragged list or signed overflow in the key
packing would abort the child."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _libasan():
    try:
        p = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    except (OSError, subprocess.CalledProcessError):
        return None
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.timeout(900)
def test_oracle_known_answers_under_asan_ubsan():
    asan = _libasan()
    if asan is None:
        pytest.skip("gcc's libasan is not installed")
    subprocess.check_call(["make", "-s", "-C", str(ROOT / "oracle"), "libgsplat_oracle_san.so"])
    env = dict(os.environ, LD_PRELOAD=asan, MTGS_ORACLE_LIB=str(ROOT / "oracle" / "libgsplat_oracle_san.so"),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               OMP_NUM_THREADS="4")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "-m", "not gpu",
                        str(ROOT / "tests" / "test_oracle_known_answers.py"),
                        str(ROOT / "tests" / "test_oracle_vs_torch_ref.py")],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-4000:]
    assert "passed" in r.stdout
