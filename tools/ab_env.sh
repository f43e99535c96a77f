#!/bin/bash
# Development (GPU box): the default bench line with an environment switch off / on, alternating on ONE box.
# usage: tools/ab_env.sh VAR=off_value VAR=on_value [extra bench args]
A=$1; B=$2; shift; shift
for rep in 1 2; do
  for kv in $A $B; do
    env $kv python bench.py --no-secondary --cpu-scenes 0 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d.get('batch_sweep',{}).get('1',{}); b8=d.get('batch_sweep',{}).get('8',{})
print('$kv', round(d['value']), round(d['ms_per_step'],3), 'dominant', round(d['roofline']['avg_launch_ms'],4), 'b1 graph', round(b.get('ms_per_step_graph',0),4), 'stream', round(b.get('ms_per_step_stream',0),4), 'b8 graph', round(b8.get('ms_per_step_graph',0),4), 'eq', b.get('graph_equals_stream'))"
  done
done
