#!/bin/bash
# Development: build a libfnp_hip variant with extra -D flags for ONE source file into findnpropagate_amd/csrc/ab/
# (git-ignored; travels to the GPU box) for same-box A/B timing with FNP_LIB_PATH.
# usage: tools/build_variant.sh <name> <file.hip> "<-D flags>"
set -e
NAME=$1; SRC=$2; DEFS=$3
cd "$(dirname "$0")/../findnpropagate_amd/csrc"
mkdir -p ab
BASE=${SRC%.hip}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function $DEFS -c $SRC -o ab/${BASE}_$NAME.o
objs=$(ls *.o | grep -v "^${BASE}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/libfnp_$NAME.so $objs ab/${BASE}_$NAME.o
echo built ab/libfnp_$NAME.so
