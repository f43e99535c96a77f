#!/bin/bash
# one-off GPU call of round 5 (rewritten per call)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r5y; mkdir -p $O
for v in r05 r04 r05 r04 r05 r04; do
  D=$R; [ $v = r04 ] && D=$R/_r04
  cd $D
  timeout -k 10 300 python bench.py --no-secondary --no-sweep --cpu-scenes 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v b128', round(d['value'],1), round(d['ms_per_step'],4))" | tee -a $O/bench_ab.log
done
for v in r05 r04 r05 r04 r05 r04; do
  D=$R; [ $v = r04 ] && D=$R/_r04
  cd $D
  timeout -k 10 300 python bench.py --batch 64 --no-secondary --no-sweep --cpu-scenes 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v b64', round(d['value'],1), round(d['ms_per_step'],4))" | tee -a $O/bench_ab.log
done
for v in r05 r04 r05 r04; do
  D=$R; [ $v = r04 ] && D=$R/_r04
  cd $D
  timeout -k 10 300 python bench.py --batch 8 --graph --no-secondary --no-sweep --cpu-scenes 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v b8 graph', round(d['value'],1), round(d['ms_per_step'],4))" | tee -a $O/bench_ab.log
done
cd $R; timeout -k 10 600 python -m pytest tests/test_gpu_spconv.py -m gpu -x -q -k "sorted" 2>&1 | tail -1
echo done
