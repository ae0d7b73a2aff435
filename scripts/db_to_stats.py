"""rocprofv3 results db (rocpd sqlite) -> per-kernel stats csv (name, calls, total/avg/min/max ns, %)."""
import csv, sqlite3, sys
db, out = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                 "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
with open(out, "w", newline="") as f:
  w = csv.writer(f)
  w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
  for r in rows:
    w.writerow([r[0], r[1], r[2], "%.1f" % r[3], r[4], r[5], "%.2f" % (100.0 * r[2] / tot)])
