#!/usr/bin/env python3
"""Development: build an A/B variant of the library with extra compiler flags.
   python scripts/build_variant.py NAME -DMTGS_EARLY_COLOR_LOAD=0 ...
writes mtgs_amd/_variants/libmtgs_rast_NAME.so; select it with scripts/kbench.py --lib <path>
(mtgs_amd._lib.use_library)."""
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mtgs_amd import build as B  # noqa: E402

name, extra = sys.argv[1], sys.argv[2:]
out_dir = B.PKG / "_variants"
obj_dir = out_dir / f"obj_{name}"
obj_dir.mkdir(parents=True, exist_ok=True)
hipcc = B._hipcc()


def cc(src):
    obj = obj_dir / (src.stem + ".o")
    subprocess.run([hipcc, "-c", str(src), "-o", str(obj)] + B.COMMON_FLAGS + B.PER_FILE_FLAGS.get(src.name, []) + extra, check=True)
    return obj


with ThreadPoolExecutor(8) as ex:
    objs = list(ex.map(cc, B.sources()))
lib = out_dir / f"libmtgs_rast_{name}.so"
subprocess.run([hipcc, "-shared", "-fPIC", f"--offload-arch={B.ARCH}", "-o", str(lib)] + [str(o) for o in objs], check=True)
print(lib)
