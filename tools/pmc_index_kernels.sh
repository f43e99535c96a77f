#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/pmc_idx
rm -rf $O; mkdir -p $O/sq $O/st
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 $R/bench.py --cpu-scenes 0 --no-sweep --no-secondary --launch stream --steps 3 --warmup 2 --reps 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $O/sq -- python3 $R/bench.py --cpu-scenes 0 --no-sweep --no-secondary --launch stream --steps 3 --warmup 2 --reps 1 > /dev/null 2>&1
find $O -name "*kernel_trace.csv" -delete
python3 $R/tools/pmc_any.py $(find $O/st -name "*kernel_stats.csv" | head -1) $O/sq > $O/idx.json
python3 - <<P
import json
rows=json.load(open("$O/idx.json"))
for e in rows:
    k=e["kernel"]
    if any(t in k for t in ("nbr_row","ell_build","strided_mark2","summary_pass","vox_","classsort","rg_clear")):
        print(k[:70].ljust(70), e.get("avg_us"), "occ", e.get("occupancy_waves_per_simd"), "wait", e.get("wait"), "stall", e.get("stall"), "active", e.get("active"), "valu/us/simd", e.get("valu_insts_per_us_per_simd"), "cu_busy", e.get("cu_busy"))
P
