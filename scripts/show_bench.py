"""print the interesting parts of bench JSON lines (files given on the command line)"""
import json, sys
for f in sys.argv[1:]:
  for line in open(f):
    line = line.strip()
    if not line.startswith("{"):
      if line: print(line)
      continue
    d = json.loads(line)
    r = d["roofline"]
    print(f, "%.2f G  %.1f us/step  step_frac %.3f  dom %s frac %.3f" % (
        d["value"] / 1e9, d["ms_per_step"] * 1e3, r.get("step_frac", 0), r["kernel"], r["frac"]))
    for k in ("kernel_ms", "phase_ms"):
      if k in r: print("   ", {a[:14]: round(b * 1e3, 1) for a, b in r[k].items()})
    if "phase_algorithmic_GBps" in r: print("    GB/s", {a[:14]: round(b) for a, b in r["phase_algorithmic_GBps"].items()})
    c = d.get("cpu_baseline")
    if c: print("    cpu %.1f M/s on %d cores" % (c["value"] / 1e6, c["cores"]))
