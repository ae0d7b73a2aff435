"""rocprofv3 --pmc csv files (one per counter pass) -> profiles/<tag>_pmc_summary.json: per kernel
(name cut at the argument list, 'void ' dropped) the counter value of its LAST launch (steady
state). usage: pmc_summary.py out.json FETCH_SIZE=fetch.csv WRITE_SIZE=write.csv"""
import csv, json, sys
out, res = sys.argv[1], {}
for arg in sys.argv[2:]:
  counter, path = arg.split("=", 1)
  last = {}
  for r in csv.DictReader(open(path)):
    if r["Counter_Name"] != counter:
      continue
    name = r["Kernel_Name"].split("(")[0]
    if name.startswith("void "):
      name = name[5:]
    last[name] = (int(r["Dispatch_Id"]), float(r["Counter_Value"]))
  for name, (_, v) in last.items():
    res.setdefault(name, {})[counter + "_KB_last_launch"] = v
json.dump({"kernels": res}, open(out, "w"), indent=1)
