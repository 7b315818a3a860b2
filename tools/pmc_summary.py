"""Collapses the rocprofv3 counter_collection CSVs of tools/pmc_knn3.sh into one row per (kernel, dispatch): counter sums."""
import csv
import glob
import os
import sys
from collections import OrderedDict

out = sys.argv[1]
rows = OrderedDict()  # (pass, kernel, dispatch) -> {counter: value}
for p in sorted(glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True)):
    pas = os.path.relpath(p, out).split(os.sep)[0]
    with open(p) as f:
        for r in csv.DictReader(f):
            k = (pas, r["Kernel_Name"].split("(")[0][:60], int(r["Dispatch_Id"]))
            rows.setdefault(k, {})
            rows[k][r["Counter_Name"]] = rows[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            rows[k]["_grid"] = r.get("Grid_Size", "")
            rows[k]["_vgpr"] = r.get("VGPR_Count", r.get("Arch_VGPR_Count", ""))
print("pass,kernel,dispatch,grid,vgpr,counters...")
for (pas, kern, disp), c in rows.items():
    extra = ",".join("%s=%.6g" % (n, v) for n, v in c.items() if not n.startswith("_"))
    print("%s,%s,%d,%s,%s,%s" % (pas, kern, disp, c.get("_grid"), c.get("_vgpr"), extra))
