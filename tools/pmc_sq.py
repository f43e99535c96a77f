#!/usr/bin/env python3
"""profiles/<name>_pmc_sq_b<B>.json: per-kernel SQ counters of one rocprofv3 --pmc pass over bench.py plus the
derived MFMA utilisation (SQ_VALU_MFMA_BUSY_CYCLES / (average launch duration x 2.4 GHz x 1024 SIMDs), durations
from the kernel-stats pass of the same command) and the wave-time split (SQ_* wave counters are in quad-cycles).

usage: pmc_sq.py <pmc_dir> <kernel_stats.csv> <batch> > out.json"""
import csv, glob, json, re, sys, collections
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from pmc_traffic import short

f = glob.glob(sys.argv[1] + '/*/*counter_collection.csv')[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
dur = {r['Name']: float(r['AverageNs']) for r in csv.DictReader(open(sys.argv[2]))}
out = {"note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY "
               "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace on `python3 bench.py --cpu-scenes 0 --steps 3 --warmup 2`; "
               "per-launch averages; mfma_util = MFMA busy cycles / (avg launch ns x 2.4 x 1024 SIMDs); wait/active/issue-stall "
               "fractions of SQ_WAVE_CYCLES", "batch": int(sys.argv[3]), "kernels": {}}
for k, cs in acc.items():
    if 'spconv' not in k and 'tile' not in k:
        continue
    a = {c: sum(v) / len(v) for c, v in cs.items()}
    e = {"launches": len(next(iter(cs.values()))), **{c: round(v) for c, v in a.items()}}
    if k in dur and a.get('SQ_VALU_MFMA_BUSY_CYCLES'):
        e["avg_launch_us"] = round(dur[k] / 1e3, 1)
        e["mfma_util"] = round(a['SQ_VALU_MFMA_BUSY_CYCLES'] / (dur[k] * 2.4 * 1024), 3)
    if a.get('SQ_WAVE_CYCLES'):
        for name, c in (("wait_frac", 'SQ_WAIT_ANY'), ("issue_stall_frac", 'SQ_WAIT_INST_ANY'), ("active_frac", 'SQ_ACTIVE_INST_ANY')):
            if c in a:
                e[name] = round(a[c] / a['SQ_WAVE_CYCLES'], 3)
    if a.get('SQ_LDS_IDX_ACTIVE'):
        e["lds_conflict_frac"] = round(a.get('SQ_LDS_BANK_CONFLICT', 0.0) / a['SQ_LDS_IDX_ACTIVE'], 3)
    out["kernels"][short(k)] = e
print(json.dumps(out, indent=1))
