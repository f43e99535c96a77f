#!/bin/bash
# one-off GPU call of round 5 (rewritten per call)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5ae; mkdir -p $O
cd $R
timeout -k 10 300 python tools/ab_bnorm.py --variants bn512,bn1024u2,bn256,$R/_r04/findnpropagate_amd/libfnp_hip.so > $O/ab_bn.log 2>$O/ab_bn.err; cat $O/ab_bn.log; tail -3 $O/ab_bn.err
timeout -k 10 300 python tools/ab_bnorm.py --dtype fp16 --scale 1.6 --variants bn512,$R/_r04/findnpropagate_amd/libfnp_hip.so > $O/ab_bn16.log 2>$O/ab_bn16.err; cat $O/ab_bn16.log
echo done
