#!/usr/bin/env python3
"""Build profiles/<name>_pmc_traffic_b<B>.json from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE).

usage: pmc_traffic.py <fetch_dir> <write_dir> <batch> > out.json
HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB and on gfx950
FETCH_SIZE counts 64 B per 128-B request (MI355X_MICROARCH.md, HBM/rocprofv3 section)."""
import csv, glob, json, re, sys, collections


def load(d, counter):
    f = glob.glob(d + '/*/*counter_collection.csv')[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == counter:
            acc[r['Kernel_Name']].append(float(r['Counter_Value']))
    return acc


def short(k):
    m = re.search(r'spconv_mfma_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)E(DF16b|f)(?:Lb(\d)E(?:DF16b|DF16_)Lb(\d)E)?', k)
    if m:
        # (,fused: the rulebook entries of a tile are made inside the kernel — from the rank grid for the strided 32 -> 64 /
        #  64 -> 128 layers, from the compact records for the 16-channel layers; without it: the (27, cap) table)
        return "spconv_mfma_kernel<%s,%s,%s,%s,%s,%s%s%s>" % (m.group(1), m.group(2), m.group(3), m.group(4),
                                                              "true" if m.group(5) == "1" else "false",
                                                              "bf16" if m.group(6) == "DF16b" else "f32",
                                                              ",fused" if m.group(7) == "1" else "",
                                                              ",sorted" if m.group(8) == "1" else "")
    m = re.search(r'spconv_mfma_kernel<(\d+), (\d+), (\d+), (\d+), (true|false), (__bf16|float)>', k)
    if m:
        return "spconv_mfma_kernel<%s,%s,%s,%s,%s,%s>" % (*m.groups()[:5], "bf16" if m.group(6) == "__bf16" else "f32")
    m = re.search(r'spconv_mfma_f32_kernelILi(\d+)ELi(\d+)ELi(\d+)ELb(\d)E', k)
    if m:
        return "spconv_mfma_f32_kernel<%s,%s,%s>" % (m.group(1), m.group(2), m.group(3))
    m = re.search(r'spconv_mfma_f32_kernel<(\d+), (\d+), (\d+), (true|false)(?:, (?:true|false))?>', k)
    if m:
        return "spconv_mfma_f32_kernel<%s,%s,%s>" % m.groups()[:3]
    m = re.search(r'(spconv_tile\d+_kernel)I(DF16b|DF16_)E', k) or re.search(r'(spconv_tile\d+_kernel)<(__bf16|_Float16)>', k)
    if m:
        return "%s<%s>" % (m.group(1), "bf16" if m.group(2) in ("DF16b", "__bf16") else "f16")
    return re.sub(r'\(.*', '', k)[:80]


if __name__ == "__main__":
    fetch, write = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
    out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, --kernel-trace only) on "
                   "`python3 bench.py --cpu-scenes 0 --steps 3 --warmup 2`; KiB units; corrected = "
                   "(2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md (gfx950 FETCH_SIZE counts 64 B per 128-B request)",
           "batch": int(sys.argv[3]), "kernels": {}}
    for k, v in fetch.items():
        w = write.get(k, [0.0])
        name = short(k)
        groups = [(name, v, w)]
        if name == "spconv_mfma_f32_kernel<128,128,3>" and len(v) % 5 == 0 and len(w) == len(v):
            # the f32 kernel's name does not carry the kernel volume: a step launches it four times for the 3x3x3 layers of stage 4
            # and once for conv_out (3x1x1) — dispatch order separates them
            groups = [(name, [x for i, x in enumerate(v) if i % 5 != 4], [x for i, x in enumerate(w) if i % 5 != 4]),
                      (name + ",k3", [x for i, x in enumerate(v) if i % 5 == 4], [x for i, x in enumerate(w) if i % 5 == 4])]
        for nm, vv, ww in groups:
            fk, wk = sum(vv) / len(vv), sum(ww) / len(ww)
            out["kernels"][nm] = {"fetch_size_kib_raw": fk, "write_size_kib": wk, "launches": len(vv),
                                  "hbm_bytes_corrected": (2 * fk + wk) * 1024}
    print(json.dumps(out, indent=1))
