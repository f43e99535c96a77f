#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_stats.csv: us per bench step per kernel."""
import csv, glob, sys
d = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 26
f = glob.glob(d + '/*/*kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
tot = sum(int(r['TotalDurationNs']) for r in rows)
print('total GPU ms/step', round(tot / steps / 1e6, 3))
import subprocess, re
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    n = r['Name']
    if n.startswith('_Z'):
        n = subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip()
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'\(.*', '', n)[:70]
    print(f"{n:72s} calls {r['Calls']:>5} {int(r['TotalDurationNs'])/steps/1e3:9.1f} us/step  avg {float(r['AverageNs'])/1e3:8.1f} us  {float(r['Percentage']):5.2f}%")
