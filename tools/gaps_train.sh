#!/bin/bash
# Development (GPU box): where the GPU idles during training steps — gaps between consecutive kernels of tools/prof_train_host.py's
# steps (rocprofv3 kernel trace), summed by the kernel that FOLLOWS the gap.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${1:-gaps}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 $R/tools/prof_train_host.py > $O/host.txt 2>&1
python3 - <<P
import csv, glob, collections
f = glob.glob("$O/tr/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 5 steps: split at the optimizer's multi-tensor kernel
marks = [i for i, r in enumerate(rows) if "multi_tensor_apply" in r["Kernel_Name"] or "foreach" in r["Kernel_Name"].lower()]
print("optimizer kernels seen:", len(marks))
lo = marks[-6] + 1 if len(marks) >= 6 else 0
hi = marks[-1] + 1
seg = rows[lo:hi]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
print("window %.2f ms for 5 steps: %.2f ms/step, busy %.2f ms/step, %d kernels/step" % ((t1 - t0) / 1e6, (t1 - t0) / 5e6, busy / 5e6, len(seg) // 5))
gaps = collections.defaultdict(lambda: [0, 0.0]); prevn = collections.defaultdict(lambda: collections.Counter())
for a, b in zip(seg, seg[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    if g > 2000:
        k = b["Kernel_Name"][:70]
        gaps[k][0] += 1; gaps[k][1] += g
        prevn[k][a["Kernel_Name"][:50]] += 1
tot = sum(v[1] for v in gaps.values())
print("gaps > 2 us: %.2f ms/step" % (tot / 5e6))
for k, (c, g) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%7.1f us/step %5.1f x/step avg %6.1f  next=%s   prev=%s" % (g / 5e3, c / 5, g / c / 1e3, k, prevn[k].most_common(1)[0][0]))
P
find $O -name "*kernel_trace.csv" -delete
