#!/bin/bash
# Development: tools/probe/gather_probe.hip -> tools/probe/libgather_probe.so (git-ignored, travels to the GPU box)
set -e
cd "$(dirname "$0")/probe"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o libgather_probe.so gather_probe.hip
echo built tools/probe/libgather_probe.so
