#!/usr/bin/env python3
"""Development: per-kernel averages of a rocprofv3 --pmc pass.  usage: pmc_quick.py <dir> [kernel substring]"""
import csv, glob, sys, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
sub = sys.argv[2] if len(sys.argv) > 2 else 'spconv'
for k, cs in acc.items():
    if sub in k:
        print(json.dumps({"kernel": k[:90], "launches": len(next(iter(cs.values()))), **{c: round(sum(v) / len(v)) for c, v in cs.items()}}))
