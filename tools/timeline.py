"""development aid: the kernels of the LAST bench step in a rocprofv3 kernel trace, in time order, with the gaps between them
usage: python tools/timeline.py gpurun_out/prof_<tag> [min_us]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last STEP: from the k_load_tiles before the last k_knn_scan2 to the next k_load_tiles (the bench's stage measurements come after it)
scans = [i for i, r in enumerate(rows) if "k_knn_scan2" in r["Kernel_Name"]]
loads = [i for i, r in enumerate(rows) if "k_load_tiles" in r["Kernel_Name"]]
first = max(i for i in loads if i < scans[-1])
after = [i for i in loads if i > scans[-1]]
rows = rows[first:(after[0] if after else len(rows))]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
acc = {}
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("tmx::", "")
    if "rocprim" in name:
        name = "rocprim:" + name.split("::")[-1][:40]
    d = (e - s) / 1e3
    if d >= min_us or (s - prev_end) / 1e3 >= min_us:
        print("%9.1f us  +%7.1f  gap %7.1f  %s" % ((s - t0) / 1e3, d, (s - prev_end) / 1e3, name[:80]))
    acc[name[:60]] = acc.get(name[:60], 0) + d
    prev_end = max(prev_end, e)
print("step span %.2f ms" % ((prev_end - t0) / 1e6))
