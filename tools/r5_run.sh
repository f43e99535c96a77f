#!/bin/bash
# one-off GPU call of round 5 (rewritten per call)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5t; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_postprocess.py tests/test_gpu_ops.py tests/test_gpu_host_abi.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc $?"; tail -15 $O/tests.log
echo done
