#!/usr/bin/env python3
"""Per-kernel summary (calls, total / average / min / max duration) of a rocprofv3 --kernel-trace run that wrote the rocpd
sqlite format (<out>_results.db) -- the same table `--stats` prints -- as CSV, for profiles/.
    python tools/rocpd_stats.py gpurun_out/r02e/prof/r02e_results.db profiles/r02e_bench_kernel_stats.csv"""
import csv, sqlite3, sys
db, out = sys.argv[1:3]
c = sqlite3.connect(db)
cols = [d[1] for d in c.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
rows = c.execute(f"select {name_col}, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) from kernels group by {name_col} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
with open(out, "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for n, k, t, a, mn, mx in rows:
        w.writerow([n, k, int(t), round(a, 3), round(100.0 * t / tot, 4), int(mn), int(mx)])
print(f"{len(rows)} kernels, {tot/1e6:.1f} ms of kernel time -> {out}")
for n, k, t, a, mn, mx in rows[:12]:
    print(f"{t/1e6:9.2f} ms {k:6d} x {a/1e3:9.1f} us  {n[:110]}")
