"""HBM-side traffic of the KNN scan from the PMC passes of tools/pmc_knn.sh, as bench.py's roofline.traffic reads it:
python3 tools/pmc_traffic.py gpurun_out/pmc_<tag> <tag>  ->  JSON on stdout (copied to profiles/<tag>_pmc_knn_traffic.json).
Launches of k_knn_scan2 in dispatch order: the pruned launches of the warm-up and the timed step, then bench.py's dense diagnostic
launch.  FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE is doubled, as /opt/skills/guides/MI355X_MICROARCH.md prescribes for
16-byte-per-lane streams on gfx950."""
import ctypes
import json
import os
import sys

out, tag = sys.argv[1], sys.argv[2]
rows = {}
for line in open(os.path.join(out, "summary.txt")).read().splitlines()[1:]:
    f = line.split(",")
    pas, disp = f[0], int([x for x in f if x.isdigit()][0])  # (the kernel's template arguments hold commas too: the first all-digit field is the dispatch)
    if "k_knn_scan2" not in line:
        continue
    c = {x.split("=")[0]: float(x.split("=")[1]) for x in f if "=" in x}
    rows.setdefault(pas, []).append((disp, c))
for v in rows.values():
    v.sort()
fetch, write = rows["fetch"], rows["write"]
assert len(fetch) >= 3 and len(write) >= 3, "expected two pruned launches and the dense launch"
pruned_f = [c["FETCH_SIZE"] for _, c in fetch[:-1]]
pruned_w = [c["WRITE_SIZE"] for _, c in write[:-1]]
fk, wk = sum(pruned_f) / len(pruned_f), sum(pruned_w) / len(pruned_w)
hit = sum(c["TCC_HIT_sum"] for _, c in write[:-1]) / max(1.0, sum(c["TCC_HIT_sum"] + c["TCC_MISS_sum"] for _, c in write[:-1]))
sq = rows.get("sq", [])
def _busy(c):  # matrix pipe busy time: SQ_VALU_MFMA_BUSY_CYCLES over the launch's SIMD-cycles (GRBM_GUI_ACTIVE is summed over the 8 XCDs)
    return c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
busy_pruned = sum(_busy(c) for _, c in sq[:-1]) / max(1, len(sq) - 1) if len(sq) >= 2 else None
busy_dense = _busy(sq[-1][1]) if sq else None
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tiler_amd", "lib", "libtilemotion.so"))
lib.tm_version.restype = ctypes.c_char_p
dense_f, dense_c = fetch[-1][1]["FETCH_SIZE"], write[-1][1]
print(json.dumps({
    "kernel_build": lib.tm_version().decode(),
    "kernel": "k_knn_scan2<6,4,true>",
    "workload": "1280x720 x 300, 16 palettes (bench.py default)",
    "traffic_bytes": (2.0 * fk + wk) * 1024.0,
    "fetch_size_kib_per_launch": fk,
    "write_size_kib_per_launch": wk,
    "tcc_hit_rate_pruned": hit,
    "mfma_busy_frac": busy_pruned,
    "mfma_busy_note": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), the sq pass, mean of the pruned launches",
    "dense_launch": {"mfma_busy_frac": busy_dense, "fetch_size_kib": dense_f, "traffic_bytes": (2.0 * dense_f + dense_c["WRITE_SIZE"]) * 1024.0,
                     "tcc_hit_rate": dense_c["TCC_HIT_sum"] / max(1.0, dense_c["TCC_HIT_sum"] + dense_c["TCC_MISS_sum"])},
    "source": "profiles/%s_pmc_knn_fetch.csv + %s_pmc_knn_write.csv: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum in separate passes "
              "(tools/pmc_knn.sh), mean of the pruned launches of a run; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16-byte-per-lane streams" % (tag, tag),
}, indent=1))
