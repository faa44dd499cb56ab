"""Per-kernel summary of a rocprofv3 rocpd (.db) trace: python tools/rocpd_stats.py <results.db> [top]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 15
rows = db.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                  "from kernels group by name order by 3 desc").fetchall()
print("%-64s %8s %12s %10s %10s %10s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us"))
for r in rows[:top]:
    print("%-64s %8d %12.1f %10.2f %10.2f %10.2f" % (r[0][:64], r[1], r[2], r[3], r[4], r[5]))
