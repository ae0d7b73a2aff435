"""rocprofv3 results db (rocpd sqlite) -> per-kernel stats csv (name, calls, total/avg/min/max ns, %).
With a marker kernel name, only dispatches from the marker's first launch on are counted (bench.py:
everything from the first k_lookup = warm-up + timed steps + per-kernel timing leg, i.e. without the
table population)."""
import csv, sqlite3, sys
db, out = sys.argv[1], sys.argv[2]
marker = sys.argv[3] if len(sys.argv) > 3 else None
c = sqlite3.connect(db)
t0 = 0
if marker:
  r = c.execute("select min(start) from kernels where name like ?", ("%" + marker + "%",)).fetchone()
  t0 = r[0] or 0
rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                 "from kernels where start >= ? group by name order by 3 desc", (t0,)).fetchall()
tot = sum(r[2] for r in rows)
with open(out, "w", newline="") as f:
  w = csv.writer(f)
  w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
  for r in rows:
    w.writerow([r[0], r[1], r[2], "%.1f" % r[3], r[4], r[5], "%.2f" % (100.0 * r[2] / tot)])
