#!/usr/bin/env python3
"""Development: per-kernel averages of every counter of one or more rocprofv3 --pmc passes (all kernels, not only the
convolutions), with the average duration from a kernel-stats CSV.  usage: pmc_any.py <kernel_stats.csv> <pmc_dir>..."""
import csv, glob, json, sys, collections
dur = {r['Name']: (float(r['AverageNs']), int(r['Calls'])) for r in csv.DictReader(open(sys.argv[1]))}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[2:]:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
rows = []
for k, cs in acc.items():
    a = {c: sum(v) / len(v) for c, v in cs.items()}
    us = dur.get(k, (0, 0))[0] / 1e3
    e = {"kernel": k[:90], "avg_us": round(us, 1), "calls": dur.get(k, (0, 0))[1]}
    wc = a.get('SQ_WAVE_CYCLES')
    if wc:
        e.update(waves=round(a.get('SQ_WAVES', 0)), wait=round(a.get('SQ_WAIT_ANY', 0) / wc, 3), stall=round(a.get('SQ_WAIT_INST_ANY', 0) / wc, 3),
                 active=round(a.get('SQ_ACTIVE_INST_ANY', 0) / wc, 3))
        if us:
            # SQ wave counters are in quad-cycles: waves resident per SIMD on average = wave_cycles * 4 / (us * 2400 MHz * 1024 SIMDs)
            e["occupancy_waves_per_simd"] = round(wc * 4 / (us * 2400 * 1024), 2)
            e["valu_insts_per_us_per_simd"] = round(a.get('SQ_INSTS_VALU', 0) / us / 1024, 2)
            e["cu_busy"] = round(a.get('SQ_BUSY_CU_CYCLES', 0) * 4 / (us * 2400 * 256), 3) if 'SQ_BUSY_CU_CYCLES' in a else None
    if 'FETCH_SIZE' in a or 'WRITE_SIZE' in a:
        by = (2 * a.get('FETCH_SIZE', 0) + a.get('WRITE_SIZE', 0)) * 1024
        e["hbm_MB"] = round(by / 1e6, 1)
        if us:
            e["hbm_TBps"] = round(by / us / 1e6, 2)
    for c in ('TCC_HIT_sum', 'TCC_MISS_sum', 'TCC_EA0_RDREQ_sum', 'TCC_EA0_WRREQ_sum', 'TCP_TCC_ATOMIC_WITH_RET_REQ_sum', 'TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum', 'TCC_ATOMIC_sum'):
        if c in a:
            e[c] = round(a[c])
    rows.append(e)
rows.sort(key=lambda e: -e["avg_us"] * max(e["calls"], 1))
print(json.dumps(rows, indent=0))
