"""bench.py --gpus 2 as the driver starts it (no launcher: bench.py spawns python -m torch.distributed.run itself, one process per rank),
rehearsed on ONE device:
TM_BENCH_REHEARSE=1 puts both ranks on the one GPU and the collectives over gloo (RCCL takes one rank per device), so everything of the
multi-process path runs except RCCL itself -- the rendezvous, the sharding of every step (Load by frames, Reduce by keys, the k-means'
data-parallel iterations, Dither, Reconstruct with the gathered database), the merges, rank 0's JSON line.  The ranks are FRESH child
processes; their result must be the single process's, tile for tile in count, and the bytes a step moves must stay where VERDICT r02
item 5 put them (a rehearsal as a test, with ranks that hold unequal counts -- 300 frames do not split evenly over the
key-frame groups, and the shards' distinct-tile counts differ)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(args, env=None, launcher=None):
    e = dict(os.environ)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    e.update(env or {})
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):  # (a launcher around pytest must not look like one around bench.py)
        e.pop(k, None)
    cmd = [sys.executable] + (launcher or []) + [os.path.join(ROOT, "bench.py")] + args
    p = subprocess.run(cmd, env=e, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=400)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads(p.stdout.strip().splitlines()[-1])


def test_two_ranks_on_one_device_equal_the_single_process():
    small = ["--width", "640", "--height", "360", "--frames", "120", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-motion-extra",
             "--no-defaults-extra", "--no-dense-extra", "--no-h2d-extra", "--no-frozen-extra", "--no-kmodes-extra", "--frozen-columns"]
    one = _bench(["--gpus", "1"] + small)
    # no launcher around it: `bench.py --gpus 2` starts its two ranks itself, as the driver's command form needs it to
    two = _bench(["--gpus", "2"] + small, env={"TM_BENCH_REHEARSE": "1"})
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "strong"
    for key in ("final_tiles_after_reindex", "global_tiles_T", "query_tiles", "distinct_database_rows"):
        assert two["config"][key] == one["config"][key], key
    c = two["collectives_per_step"]
    assert c["path"].startswith("host callback") and c["all_gather"] >= 1
    # PreparePalettes: the tile -> palette clustering is run whole by every rank (one resident launch each, no collective) -- the only int64
    # all-reduces a step has belong to the data-parallel Lloyd iterations, which are gone; the whole step stays far below the ~330 collectives
    # (one per Lloyd iteration) it took before
    calls = sum(c[k] for k in ("all_reduce_sum_i32", "all_reduce_max_i32", "all_reduce_sum_i64", "all_gather"))
    assert c["all_reduce_sum_i64"] == 0 and calls <= 40, c
    by = c["bytes_by_stage"]
    # Reduce moves keys and candidates, not every distinct tile: 16 B per distinct frame tile + 264 B per candidate, against 264 B per distinct tile before
    distinct = one["config"]["knn_queries"]  # (the single process searches once per distinct frame tile)
    assert by["esReduce"] < 0.5 * 264 * distinct, (by, distinct)


def test_two_ranks_on_the_literal_clip_and_the_data_parallel_clustering():
    """the same on SURVEY.md 8(d)'s generator as written (the clip `value` is quoted on: no duplicate frame tiles, a tile -> palette clustering
    that runs to its iteration cap), with the clustering replicated (the default: <= 120 collectives per step asked for, ~15 taken) and with
    TM_PP_SHARDED=1 (the data-parallel Lloyd: an int64 all-reduce per iteration) -- both equal the single process"""
    small = ["--width", "640", "--height", "360", "--frames", "120", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-motion-extra",
             "--no-defaults-extra", "--no-dense-extra", "--no-h2d-extra", "--no-frozen-extra", "--no-kmodes-extra"]
    one = _bench(["--gpus", "1"] + small)
    assert one["parity_gate"] == "passed"
    for env in ({"TM_BENCH_REHEARSE": "1"}, {"TM_BENCH_REHEARSE": "1", "TM_PP_SHARDED": "1"}):
        two = _bench(["--gpus", "2"] + small, env=env)
        assert two["n_gpus"] == 2 and two["parity_gate"] == "passed"
        for key in ("final_tiles_after_reindex", "global_tiles_T", "query_tiles", "distinct_database_rows"):
            assert two["config"][key] == one["config"][key], key
        c = two["collectives_per_step"]
        calls = sum(c[k] for k in ("all_reduce_sum_i32", "all_reduce_max_i32", "all_reduce_sum_i64", "all_gather"))
        if "TM_PP_SHARDED" in env:
            assert c["all_reduce_sum_i64"] >= 5, c
        else:
            assert c["all_reduce_sum_i64"] == 0 and calls <= 120, c
