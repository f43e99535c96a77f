#!/bin/bash
# Development: build libfnp_hip variants with -DFNP_ABLATE=<mask> into findnpropagate_amd/csrc/ab/ (git-ignored).
set -e
cd "$(dirname "$0")/../findnpropagate_amd/csrc"
mkdir -p ab
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off ${FNP_AB_DEFS:--DFNP_ABLATE=$m} -c spconv.hip -o ab/spconv_$m.o &
done
wait
for m in "$@"; do
  objs=$(ls *.o | grep -v '^spconv.o$')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/libfnp_ab$m.so $objs ab/spconv_$m.o
done
