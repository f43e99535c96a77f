#!/bin/bash
# Development (GPU box): one / two streams over batch sizes, stream launches and hipGraph replay.
for B in 1 2 4 8 16 32; do
  for g in "" "--graph"; do
    for t in 0 1 0 1; do
      FNP_TWO_STREAMS=$t python bench.py --batch $B --no-sweep --no-secondary --cpu-scenes 0 --steps 40 --warmup 5 $g 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('B=$B', '${g:-stream}', 'two=$t', round(d['ms_per_step'],4), round(d['value']))"
    done
  done
done
