#!/usr/bin/env python3
"""Average PMC counters per kernel name from rocprofv3 counter_collection.csv files."""
import csv, glob, sys, re, subprocess, collections
for d in sys.argv[1:]:
    f = glob.glob(d + '/*/*counter_collection.csv')[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, cs in acc.items():
        if 'spconv_mfma' not in k: continue
        name = re.sub(r'.*spconv_mfma_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d).*', r'mfma<\1,\2,\3,\4,w\5>', k)
        print(name, {c: round(sum(v) / len(v)) for c, v in cs.items()})
