"""Aggregate kernel durations of one training step (between two adam launches) of a rocprofv3 trace.
usage: step_breakdown.py trace.csv [adam_index_from_end]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adam_kernel")]
k = int(sys.argv[2]) if len(sys.argv) > 2 else -10
a, b = adam[k], adam[k + 1]
seg = rows[a + 1:b + 1]
t0 = int(seg[0]["Start_Timestamp"]); t1 = int(seg[-1]["End_Timestamp"])
print("step wall ms", (t1 - t0) / 1e6, "kernels", len(seg))
agg = collections.defaultdict(lambda: [0, 0])
for r in seg:
    key = r["Kernel_Name"][:60]
    agg[key][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); agg[key][1] += 1
tot = 0
for key, (d, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    print(f"{key:60s} n {c:3d} tot {d / 1e3:8.1f} us")
print("sum of all kernel durations ms", sum(d for d, c in agg.values()) / 1e6)
