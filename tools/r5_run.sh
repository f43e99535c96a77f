#!/bin/bash
# one-off GPU call of round 5 (rewritten per call)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5j; mkdir -p $O
cd $R
timeout -k 10 400 python tools/ab_tiled.py --batch 128 --variants prio1,prio2,prio3 --channels 64 > $O/ab64_128.log 2>$O/ab64_128.err; tail -1 $O/ab64_128.log | cut -c1-800
timeout -k 10 400 python tools/ab_tiled.py --batch 128 --variants cprio1,cprio3 --channels 32 > $O/ab32_128.log 2>$O/ab32_128.err; tail -1 $O/ab32_128.log | cut -c1-800
timeout -k 10 400 python tools/ab_tiled.py --batch 128 --variants sprio2 --channels 128 > $O/ab128_128.log 2>$O/ab128_128.err; tail -1 $O/ab128_128.log | cut -c1-800
for v in main sprio2 main sprio2; do
  L=$R/findnpropagate_amd/csrc/ab/libfnp_$v.so; [ $v = main ] && L=$R/findnpropagate_amd/libfnp_hip.so
  FNP_LIB_PATH=$L timeout -k 10 300 python bench.py --no-secondary --no-sweep --cpu-scenes 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value'],1), round(d['ms_per_step'],4), {k: round(v,3) for k,v in d['roofline']['all_conv_classes_ms_per_step'].items()})" | tee -a $O/bench_ab.log
done
echo done
