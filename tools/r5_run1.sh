#!/bin/bash
# round-5 GPU call 1: full GPU suite, tile64 scheduling variants A/B, XCD-swizzle A/B (whole step + kernel stats)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5a; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc $?" | tee $O/tests.rc; tail -3 $O/tests.log
timeout -k 10 300 python tools/ab_tiled.py --batch 64 --variants s2,s3 --channels 64 > $O/ab64.log 2>$O/ab64.err; tail -2 $O/ab64.log | cut -c1-600
timeout -k 10 300 python tools/ab_tiled.py --batch 128 --variants s2,s3 --channels 64 > $O/ab128.log 2>$O/ab128.err; tail -2 $O/ab128.log | cut -c1-600
for v in main noxcd main noxcd; do
  L=$R/findnpropagate_amd/csrc/ab/libfnp_$v.so; [ $v = main ] && L=$R/findnpropagate_amd/libfnp_hip.so
  FNP_LIB_PATH=$L timeout -k 10 300 python bench.py --no-secondary --no-sweep --cpu-scenes 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value'],1), round(d['ms_per_step'],4))" | tee -a $O/bench_ab.log
done
cd /tmp && export TMPDIR=/tmp
for v in main noxcd; do
  L=$R/findnpropagate_amd/csrc/ab/libfnp_$v.so; [ $v = main ] && L=$R/findnpropagate_amd/libfnp_hip.so
  export FNP_LIB_PATH=$L
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$v -- python3 $R/bench.py --batch 128 --cpu-scenes 0 --no-sweep --no-secondary --launch stream > /dev/null 2>&1
  find $O/stats_$v -name "*kernel_trace.csv" -delete
  f=$(find $O/stats_$v -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_$v.csv
done
echo done
