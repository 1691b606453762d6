#!/bin/bash
# Development: builds mtgs_amd/_variants/libmtgs_rast_base.so with csrc/<file> taken from a git revision (default HEAD),
# everything else from the current objects -- the A side of an A/B of a kernel file.   usage: build_base_variant.sh blend.hip [rev]
set -e
cd "$(dirname "$0")/../.."
f=$1; rev=${2:-HEAD}
mkdir -p mtgs_amd/_variants/src_base
git show $rev:mtgs_amd/csrc/$f > mtgs_amd/_variants/src_base/$f
for h in mtgs_amd/csrc/*.hpp; do git show $rev:$h > mtgs_amd/_variants/src_base/$(basename $h); done
mkdir -p mtgs_amd/_variants/src_base/../../../include_base
/opt/rocm/bin/hipcc -c mtgs_amd/_variants/src_base/$f -o mtgs_amd/_variants/src_base/${f%.hip}.o --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-slp-vectorize -Iinclude -Imtgs_amd/csrc
objs=$(ls mtgs_amd/csrc/_obj/*.o | grep -v "/${f%.hip}.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o mtgs_amd/_variants/libmtgs_rast_base.so $objs mtgs_amd/_variants/src_base/${f%.hip}.o
echo mtgs_amd/_variants/libmtgs_rast_base.so
