#!/usr/bin/env python3
"""Per-dispatch PMC counters of tools/gpu_cold_layer.py (two rocprofv3 --pmc passes, tools/cold_layer_counters.sh):
the convolution's launches in dispatch order, split into the two rounds the script runs (each starts right after a
1 GiB fill), averaged over the first 3 launches of a round and over launches 40..63, plus the per-launch times the
script printed (pass a).  usage: cold_layer_summary.py <dir_a> <dir_b> <log_a>"""
import collections, csv, glob, sys


def load(d):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if "conv3x3_igemm" in r["Kernel_Name"]:
            per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    return [per[k] for k in sorted(per)]


N = 64
for d in sys.argv[1:3]:
    rows = load(d)
    rows = rows[-2 * N:]          # the two measured rounds (the warm-up launch comes first)
    print(f"== {d.split('/')[-1]}: {len(rows)} launches in two rounds of {N}")
    names = sorted(rows[0])
    for rnd in range(2):
        seg = rows[rnd * N:(rnd + 1) * N]
        cold = {n: sum(r[n] for r in seg[:3]) / 3 for n in names}
        warm = {n: sum(r[n] for r in seg[40:]) / len(seg[40:]) for n in names}
        for n in names:
            ratio = cold[n] / warm[n] if warm[n] else float("nan")
            print(f"  round {rnd} {n:40s} first 3: {cold[n]:14.0f}   launches 40+: {warm[n]:14.0f}   ratio {ratio:6.3f}")
        if "TCC_HIT_sum" in names:
            for tagn, c in (("first 3", cold), ("40+", warm)):
                hr = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
                dram = c["TCC_EA0_RDREQ_DRAM_sum"] / c["TCC_EA0_RDREQ_sum"]
                print(f"  round {rnd} {tagn:8s}: L2 hit rate {hr:.4f}; fabric reads that went to DRAM (not served by the Infinity Cache) {dram:.4f}")
        if "TCC_EA0_RDREQ_LEVEL_sum" in names:
            for tagn, c in (("first 3", cold), ("40+", warm)):
                print(f"  round {rnd} {tagn:8s}: mean fabric read latency {c['TCC_EA0_RDREQ_LEVEL_sum'] / c['TCC_EA0_RDREQ_sum']:.0f} TCC cycles")
print("== per-launch times printed by the script (pass a):")
for line in open(sys.argv[3]):
    if line.startswith("round"):
        print("  " + line.strip())
