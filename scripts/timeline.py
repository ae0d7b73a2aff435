"""Print the kernel timeline of the last step in a rocprofv3 results db (between the last two
occurrences of a marker kernel; an optional third argument k moves that window k occurrences
back: bench.py ends with a ten-step profiling loop that is not the timed step)."""
import sqlite3, sys
db, marker = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
rows = c.execute("select name, start, end, stream_id from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if marker in r[0]]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 0
i0, i1 = idx[-2 - back], idx[-1 - back]
t0 = rows[i0][1]
for r in rows[i0:i1]:
  print("%-84s %8.1f %8.1f  s%s" % (r[0][:84], (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3]))
print("step span us: %.1f" % ((rows[i1][1] - t0) / 1e3))
per = [(rows[b][1] - rows[a][1]) / 1e3 for a, b in zip(idx[:-1], idx[1:])]
print("marker periods us:", " ".join("%.0f" % p for p in per))
