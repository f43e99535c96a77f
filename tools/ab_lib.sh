#!/bin/bash
# Development (GPU box): the default bench line with the in-tree library and with a variant library (FNP_LIB_PATH), alternating.
# usage: tools/ab_lib.sh <path to variant .so> [extra bench args]
V=$1; shift
for rep in 1 2; do
  for lib in "" "$V"; do
    FNP_LIB_PATH=$lib python bench.py --no-secondary --cpu-scenes 0 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d.get('batch_sweep',{}).get('1',{}); b8=d.get('batch_sweep',{}).get('8',{})
print('${lib:-base}'.split('/')[-1], round(d['value']), round(d['ms_per_step'],3), 'dominant', round(d['roofline']['avg_launch_ms'],4), 'b1 graph', round(b.get('ms_per_step_graph',0),4), 'stream', round(b.get('ms_per_step_stream',0),4), 'b8 graph', round(b8.get('ms_per_step_graph',0),4))"
  done
done
