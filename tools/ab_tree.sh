#!/bin/bash
# Development (GPU box): the same bench line from this tree and from a copy of another revision under _ab_head/ (git archive +
# make there), alternating, on ONE box.  Prints value, ms/step and the one-scene graph rate of each run.
R=${GRAFT_REPO_ROOT:-$PWD}
for rep in 1 2; do
  for tree in $R/_ab_head $R; do
    (cd $tree && python bench.py --no-secondary --cpu-scenes 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d['batch_sweep']['1']
print('$tree'.split('/')[-1][:8], round(d['value']), round(d['ms_per_step'],3), 'b1 graph', round(b['ms_per_step_graph'],4), 'stream', round(b['ms_per_step_stream'],4), 'b8', round(d['batch_sweep']['8']['ms_per_step_graph'],4))")
  done
done
