"""Builds mtgs_amd/libmtgs_rast.so (gfx950) from mtgs_amd/csrc/*.hip with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so is
git-ignored but travels with the tree to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
OBJ = CSRC / "_obj"
LIB = PKG / "libmtgs_rast.so"
ARCH = "gfx950"

# -fno-slp-vectorize: on gfx950 v_pk_{fma,mul,add}_f32 issue in two passes (no throughput gain for
# fp32) but the SLP vectoriser pays v_mov shuffles to pair their operands -- measured -19 % time on
# the compositing backward without it.
COMMON_FLAGS = [
    f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-fno-slp-vectorize",
    "-Wall", "-Wno-unused-function", f"-I{PKG.parent / 'include'}",
]
# The projection forward must round after every operation (bit-exact tile binning inputs).
PER_FILE_FLAGS = {"project.hip": ["-ffp-contract=off"], "front.hip": ["-ffp-contract=off"]}


class HipccNotFound(RuntimeError):
    """No hipcc on this machine (a GPU box running the prebuilt in-tree .so): the only build error callers may ignore."""


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise HipccNotFound("hipcc not found (set HIPCC or add /opt/rocm/bin to PATH)")


def sources():
    return sorted(CSRC.glob("*.hip"))


def _stale(out: Path, deps) -> bool:
    if not out.exists():
        return True
    t = out.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def stale_sources():
    """csrc files (and headers) newer than the built library -- empty when the .so is up to date."""
    if not LIB.exists():
        return sources()
    t = LIB.stat().st_mtime
    deps = sources() + list(CSRC.glob("*.hpp")) + [PKG.parent / "include" / "mtgs_rast.h"]
    return [d for d in deps if d.stat().st_mtime > t]


def build(force: bool = False, verbose: bool = False) -> Path:
    hipcc = _hipcc()
    OBJ.mkdir(exist_ok=True)
    live = {s.stem for s in sources()}
    for stale in OBJ.glob("*.o"):          # objects of sources that no longer exist (bin2.o after round 2)
        if stale.stem not in live:
            stale.unlink()
    headers = list(CSRC.glob("*.hpp")) + [PKG.parent / "include" / "mtgs_rast.h", Path(__file__)]
    jobs = []
    for src in sources():
        obj = OBJ / (src.stem + ".o")
        if force or _stale(obj, [src] + headers):
            cmd = [hipcc, "-c", str(src), "-o", str(obj)] + COMMON_FLAGS + PER_FILE_FLAGS.get(src.name, [])
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        return r.stderr

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            for warn in ex.map(run, jobs):
                if verbose and warn:
                    print(warn, file=sys.stderr)
    objs = [OBJ / (s.stem + ".o") for s in sources()]
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(LIB)] + [str(o) for o in objs])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
