"""Print the kernels between two consecutive p_sample launches of a rocprofv3 kernel trace."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
pat = sys.argv[2] if len(sys.argv) > 2 else "p_sample"
idx = [i for i, r in enumerate(rows) if pat in r["Kernel_Name"]]
k = len(idx) // 2
a, b = idx[k], idx[k + 1]
prev_end = None
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0
    print(f"{r['Kernel_Name'][:64]:64s} dur {(e - s) / 1e3:7.2f} us gap {gap:6.2f} grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}")
    prev_end = e
print("step total us", (int(rows[b]["End_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1e3)
