"""Kernel timeline of one training step from a rocprofv3 kernel trace (queue id, start, end, name)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adam_kernel")]
a, b = adam[10], adam[11]
seg = rows[a + 1:b + 1]
t0 = int(seg[0]["Start_Timestamp"])
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 0
hi = int(sys.argv[3]) if len(sys.argv) > 3 else len(seg)
for r in seg[lo:hi]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"q{r['Queue_Id']:>2s} {s / 1e3:9.1f} -> {e / 1e3:9.1f} ({(e - s) / 1e3:7.1f}) {r['Kernel_Name'][:60]}")
