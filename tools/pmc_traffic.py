"""HBM-side traffic of the KNN scan from the PMC passes of tools/pmc_knn3.sh, as bench.py's roofline.traffic reads it:
python3 tools/pmc_traffic.py gpurun_out/pmc_<tag> <tag> [--dense]  ->  JSON on stdout (copied to profiles/<tag>_pmc_knn_traffic.json).
Launches of k_knn_consume in dispatch order: the pruned launches of the warm-up and the timed step, then (--dense: the run kept bench.py's
dense diagnostic) the dense launch.  k_knn_seed and k_knn_lists run once per pruned scan before it; their bytes are reported beside the
dominant kernel's, not inside roofline.traffic.  FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE is doubled, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for 16-byte-per-lane streams on gfx950."""
import ctypes
import json
import os
import sys

out, tag = sys.argv[1], sys.argv[2]
has_dense = "--dense" in sys.argv[3:]
rows = {}
consume_name = "k_knn_consume"
for line in open(os.path.join(out, "summary.txt")).read().splitlines()[1:]:
    f = line.split(",")
    pas, disp = f[0], int([x for x in f if x.isdigit()][0])  # (the kernel's template arguments hold commas too: the first all-digit field is the dispatch)
    kern = next((k for k in ("k_knn_seed", "k_knn_lists", "k_knn_consume") if k in line), None)
    if kern is None:
        continue
    if kern == "k_knn_consume":
        consume_name = line.split("tmx::")[1].split(",%d," % disp)[0]
    c = {x.split("=")[0]: float(x.split("=")[1]) for x in f if "=" in x}
    rows.setdefault((pas, kern), []).append((disp, c))
for v in rows.values():
    v.sort()


def launches(pas, kern):  # (pruned launches, dense launch): the run's FIRST search sizes its list arena from a guess and is repeated -- left out
    v = rows.get((pas, kern), [])
    dense = []
    if kern == "k_knn_consume" and has_dense:
        v, dense = v[:-1], v[-1:]
    return v[-2:], dense


def mean(xs):
    xs = list(xs)
    return sum(xs) / len(xs) if xs else None


def busy(c):  # matrix pipe busy time: SQ_VALU_MFMA_BUSY_CYCLES over the launch's SIMD-cycles (GRBM_GUI_ACTIVE is summed over the 8 XCDs)
    return c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)


per_kernel = {}
for kern in ("k_knn_seed", "k_knn_lists", "k_knn_consume"):
    fp, _ = launches("fetch", kern)
    wp, _ = launches("write", kern)
    sp, _ = launches("sq", kern)
    if not fp or not wp:
        continue
    fk, wk = mean(c["FETCH_SIZE"] for _, c in fp), mean(c["WRITE_SIZE"] for _, c in wp)
    hits, miss = sum(c["TCC_HIT_sum"] for _, c in wp), sum(c["TCC_MISS_sum"] for _, c in wp)
    per_kernel[kern] = {"launches": len(fp), "fetch_size_kib_per_launch": fk, "write_size_kib_per_launch": wk, "traffic_bytes": (2.0 * fk + wk) * 1024.0,
                        "tcc_hit_rate": hits / max(1.0, hits + miss), "mfma_busy_frac": mean(busy(c) for _, c in sp) if sp else None}
assert "k_knn_consume" in per_kernel and per_kernel["k_knn_consume"]["launches"] >= 2, "expected the pruned launches of the warm-up and of the timed step"
cons = per_kernel["k_knn_consume"]
_, fd = launches("fetch", "k_knn_consume")
_, wd = launches("write", "k_knn_consume")
_, sd = launches("sq", "k_knn_consume")
dense = None
if fd and wd:
    dc = wd[0][1]
    dense = {"mfma_busy_frac": busy(sd[0][1]) if sd else None, "fetch_size_kib": fd[0][1]["FETCH_SIZE"], "traffic_bytes": (2.0 * fd[0][1]["FETCH_SIZE"] + dc["WRITE_SIZE"]) * 1024.0,
             "tcc_hit_rate": dc["TCC_HIT_sum"] / max(1.0, dc["TCC_HIT_sum"] + dc["TCC_MISS_sum"])}
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tiler_amd", "lib", "libtilemotion.so"))
lib.tm_version.restype = ctypes.c_char_p
print(json.dumps({
    "kernel_build": lib.tm_version().decode(),
    "kernel": consume_name,
    "workload": "1280x720 x 300, 16 palettes (bench.py default)",
    "traffic_bytes": cons["traffic_bytes"],
    "fetch_size_kib_per_launch": cons["fetch_size_kib_per_launch"],
    "write_size_kib_per_launch": cons["write_size_kib_per_launch"],
    "tcc_hit_rate_pruned": cons["tcc_hit_rate"],
    "mfma_busy_frac": cons["mfma_busy_frac"],
    "mfma_busy_note": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), the sq pass, mean of the pruned launches of k_knn_consume",
    "scan_traffic_bytes": sum(v["traffic_bytes"] for v in per_kernel.values()),
    "per_kernel": per_kernel,
    "dense_launch": dense,
    "source": "profiles/%s_pmc_knn_fetch.csv + %s_pmc_knn_write.csv: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum in separate passes "
              "(tools/pmc_knn3.sh), mean of the pruned launches of a run; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16-byte-per-lane streams" % (tag, tag),
}, indent=1))
