#!/usr/bin/env python3
"""Development: from a rocprofv3 --kernel-trace CSV of bench.py, the timeline of ONE replayed step — union busy time, idle gaps,
time with >1 kernel in flight, and the longest gaps with the kernels around them.  usage: step_timeline.py <kernel_trace.csv>"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows), key=lambda e: e[0])
# steps are delimited by vox_mark_kernel launches; take the steps of the last timed repetition
marks = [i for i, e in enumerate(ev) if "vox_mark_kernel" in e[2]]
short = lambda n: re.sub(r"\(anonymous namespace\)::|^void |_ZN12_GLOBAL__N_1\d+|\(.*", "", n)[:48]
res = []
for a, b in zip(marks[-12:-2], marks[-11:-1]):
    step = ev[a:b]
    t0, t1 = step[0][0], ev[b][0]
    busy = 0; over = 0; cur_end = t0; gaps = []
    for s, e, n, q in step:
        if s > cur_end:
            gaps.append((s - cur_end, short(n)))
            busy += e - s; cur_end = e
        else:
            over += min(e, cur_end) - s
            if e > cur_end:
                busy += e - cur_end; cur_end = e
    res.append((t1 - t0, busy, sum(g for g, _ in gaps), over, sorted(gaps, reverse=True)[:6], len(step)))
for tot, busy, idle, over, gaps, nk in res[-3:]:
    print(f"step {tot/1e6:.3f} ms: busy(union) {busy/1e6:.3f}, idle {idle/1e6:.3f}, overlapped kernel time {over/1e6:.3f}, kernels {nk}")
    print("   largest gaps (us, before kernel):", [(round(g/1e3, 1), n) for g, n in gaps])
