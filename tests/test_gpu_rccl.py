"""The stream-ordered collective path (tm_set_collective_mode 1) over RCCL itself, as far as one GPU allows: a one-rank NCCL group
in a child process (its own process group, so nothing leaks into the other tests).  What it pins: torch accepts the library's raw HIP
stream as the current stream of a collective, the collective is ordered after the work queued on that stream and before the work
queued next, and the byte-wise all-gather / typed all-reduces run on views built from raw device pointers."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from tiler_amd import distributed as D
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1, device_id=torch.device("cuda", 0))
coll = D.Collective(0, 1)
lib_stream = torch.cuda.Stream()            # stands for the encoder's stream: a raw hipStream_t
assert coll.bind_stream(lib_stream.cuda_stream) is True
n = 1 << 22
with torch.cuda.stream(lib_stream):
    a = torch.zeros(n, dtype=torch.int32, device="cuda")
    for _ in range(50):
        a += 1                               # queued work the collective has to wait for
    g = torch.empty(n, dtype=torch.int32, device="cuda")
assert coll._from_library(None, D.KIND_SUM_I32, a.data_ptr(), None, n) == 0
assert coll._from_library(None, D.KIND_ALLGATHER, a.data_ptr(), g.data_ptr(), n * 4) == 0
with torch.cuda.stream(lib_stream):
    h = g * 2                                # queued after the collectives, no host wait in between
    b = torch.arange(8, dtype=torch.int64, device="cuda")
assert coll._from_library(None, D.KIND_SUM_I64, b.data_ptr(), None, 8) == 0
assert coll._from_library(None, D.KIND_MAX_I32, a.data_ptr(), None, n) == 0
lib_stream.synchronize()
assert int(h.min()) == 100 and int(h.max()) == 100, (int(h.min()), int(h.max()))
assert b.tolist() == list(range(8))
assert coll.calls[D.KIND_ALLGATHER] == 1 and coll.calls[D.KIND_SUM_I32] == 1
os.environ["TM_COLL_BLOCKING"] = "1"
assert coll.bind_stream(lib_stream.cuda_stream) is False   # the blocking contract on request
dist.destroy_process_group()
print("RCCL-OK")
"""


def test_stream_ordered_collectives_over_rccl_one_rank():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("TM_COLL_BLOCKING", None)
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "RCCL-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
