#!/usr/bin/env python3
"""Development: from a rocprofv3 --kernel-trace CSV of bench.py --launch pipeline, where the time of a pipelined step goes on the
CONVOLUTION chain: per convolution kernel name the average duration, and between consecutive convolution kernels (any batch) the
idle gap — conv chain busy + gaps = the step.  usage: pipe_timeline.py <kernel_trace.csv> [steps]"""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
short = lambda n: re.sub(r"\(anonymous namespace\)::|^void |_ZN12_GLOBAL__N_1\d+", "", n)[:70]
is_conv = lambda n: "spconv_" in n
convs = [e for e in ev if is_conv(e[2])]
per_step = 21
convs = convs[-nsteps * per_step:]
t0, t1 = convs[0][0], convs[-1][1]
busy = sum(e - s for s, e, _ in convs)
gaps = collections.defaultdict(list)
over = 0
for a, b in zip(convs, convs[1:]):
    g = b[0] - a[1]
    if g >= 0:
        gaps[(short(a[2])[:40], short(b[2])[:40])].append(g)
    else:
        over += -g
dur = collections.defaultdict(list)
for s, e, n in convs:
    dur[short(n)].append(e - s)
print(f"{nsteps} steps: span {1e-6*(t1-t0)/nsteps:.3f} ms/step, conv kernel time {1e-6*busy/nsteps:.3f} ms/step, "
      f"gaps between convs {1e-6*sum(sum(v) for v in gaps.values())/nsteps:.3f} ms/step, conv-conv overlap {1e-6*over/nsteps:.3f} ms/step")
for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {1e-3*sum(v)/len(v):9.1f} us x {len(v)/nsteps:5.1f}/step  {n}")
print("largest gap classes (us per step, count per step, after -> before):")
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print(f"  {1e-3*sum(v)/nsteps:8.1f} us  {len(v)/nsteps:4.1f}  {k[0]} -> {k[1]}")
# what else runs: non-conv kernel time per step inside the span
other = [e for e in ev if not is_conv(e[2]) and e[0] >= t0 and e[1] <= t1]
print(f"non-conv kernel time inside the span: {1e-6*sum(e - s for s, e, _ in other)/nsteps:.3f} ms/step in {len(other)/nsteps:.1f} launches/step")
