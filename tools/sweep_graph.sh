#!/bin/bash
# Development (GPU box): scenes/s of the replayed forward (hipGraph, two branches, counts leaving early) and of plain stream
# launches over batch sizes.
for B in 1 2 4 8 16 32 64; do
  for g in "--graph" ""; do
    python bench.py --batch $B --no-sweep --no-secondary --cpu-scenes 0 --steps 40 --warmup 5 --launch stream $g 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('B=$B', '${g:-stream}', round(d['ms_per_step'],4), round(d['value']))"
  done
done
