"""development aid: every kernel of one bench step in a rocprofv3 kernel trace between the first kernel whose name contains FROM and the
first after it whose name contains TO, in time order with gaps.   python tools/timeline3.py gpurun_out/prof_<tag> step FROM TO [min_us]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
k = int(sys.argv[2])
frm, to = sys.argv[3], sys.argv[4]
min_us = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
loads = [i for i, r in enumerate(rows) if "k_load_tiles" in r["Kernel_Name"]]
starts = [loads[0]]
for a, b in zip(loads, loads[1:]):
    if int(rows[b]["Start_Timestamp"]) - int(rows[a]["End_Timestamp"]) > 5e6:
        starts.append(b)
rows = rows[starts[k]:starts[k + 1] if k + 1 < len(starts) else len(rows)]
on = False
t0 = prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("tmx::", "")
    if not on and frm in r["Kernel_Name"]:
        on, t0, prev_end = True, s, s
    if not on:
        continue
    if to in r["Kernel_Name"] and s > t0:
        break
    if "rocprim" in name:
        name = "rocprim:" + r["Kernel_Name"].split("detail::")[-1][:70]
    d = (e - s) / 1e3
    if d >= min_us or (s - prev_end) / 1e3 >= min_us:
        print("%9.1f us  +%7.1f  gap %7.1f  %s" % ((s - t0) / 1e3, d, (s - prev_end) / 1e3, name[:100]))
    prev_end = max(prev_end, e)
print("span %.2f ms" % ((prev_end - t0) / 1e6))
