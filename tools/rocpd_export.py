#!/usr/bin/env python3
"""rocprofv3 (ROCm 7.2) writes a rocpd SQLite database by default.  Export what the repo keeps:

    python tools/rocpd_export.py <results.db> <out_prefix>

  <out_prefix>_kernel_stats.csv   per-kernel calls / total / average / share / min / max (the `--stats` summary)
  <out_prefix>_kernel_trace.csv   one row per dispatch (name, start, end, stream, queue, grid): input of
                                  tools/insitu.py, tools/exposed.py; NOT committed (large) unless asked
"""
import csv
import sqlite3
import sys

db, prefix = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
rows = c.execute("select name, start, end, stream_id, queue_id, grid_x, grid_y, grid_z, workgroup_x, vgpr_count, "
                 "accum_vgpr_count, sgpr_count, lds_size from kernels order by start").fetchall()
with open(prefix + "_kernel_trace.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Stream_Id", "Queue_Id", "Grid_X", "Grid_Y", "Grid_Z",
                "Workgroup_X", "VGPR", "AGPR", "SGPR", "LDS"])
    w.writerows(rows)
stats = {}
for r in rows:
    d = r[2] - r[1]
    s = stats.setdefault(r[0], [0, 0, 1 << 62, 0])
    s[0] += 1; s[1] += d; s[2] = min(s[2], d); s[3] = max(s[3], d)
tot = sum(s[1] for s in stats.values()) or 1
with open(prefix + "_kernel_stats.csv", "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for name, s in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        w.writerow([name, s[0], s[1], round(s[1] / s[0], 1), round(100.0 * s[1] / tot, 3), s[2], s[3]])
print(f"{len(rows)} dispatches, {len(stats)} kernels -> {prefix}_kernel_stats.csv, {prefix}_kernel_trace.csv")
