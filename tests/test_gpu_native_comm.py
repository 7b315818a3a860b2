"""The native multi-process path: RCCL linked into libtilemotion.so, reached through tm_comm_unique_id /
tm_comm_init from a plain C host -- tests/c/native_comm.c, compiled here with gcc and started as FRESH CHILD PROCESSES (no Python,
no torch in them).  A one-GPU box can hold one rank of a communicator only (RCCL refuses two ranks on one device), so the
communicator has one rank and TM_COMM_FORCE_DIST=1 makes the encoder walk its sharded paths anyway: every merge of every step
goes through ncclAllReduce / ncclAllGather on the encoder's stream, and the result must be byte for byte the single-process run's.
With more devices visible (the driver's multi-GPU node) the same program runs as one rank per device."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "tiler_amd", "lib")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("native") / "native_comm")
    subprocess.check_call(["gcc", "-O1", "-Wall", "-std=c11", "-D_DEFAULT_SOURCE", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "native_comm.c"),
                           "-o", out, "-L", LIBDIR, "-ltilemotion", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath-link,/opt/rocm/lib"])
    return out


def _run(cmd, env=None, timeout=300):
    e = dict(os.environ)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    e.update(env or {})
    return subprocess.Popen(cmd, env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)


def test_c_host_through_the_native_communicator(exe, tmp_path):
    single = str(tmp_path / "single.bin")
    p = _run([exe, "single", single])
    out, _ = p.communicate(timeout=300)
    assert p.returncode == 0, out
    # one rank, sharded paths forced: all collectives of Run(esAll) go through RCCL inside the library
    one = str(tmp_path / "rank0of1.bin")
    p = _run([exe, "rank", "0", "1", str(tmp_path / "id1"), one], env={"TM_COMM_FORCE_DIST": "1"})
    out, _ = p.communicate(timeout=300)
    assert p.returncode == 0, out
    a, b = open(single, "rb").read(), open(one, "rb").read()
    assert len(a) > 1000 and a == b, "the run through the native communicator differs from the single run"


def test_c_hosts_one_rank_per_device(exe, tmp_path):
    """two processes, two devices, one communicator -- only where the box has them (skipped on a one-GPU box)"""
    import ctypes
    lib = ctypes.CDLL(os.path.join(LIBDIR, "libtilemotion.so"))
    lib.tm_device_count.restype = ctypes.c_int
    if lib.tm_device_count() < 2:
        pytest.skip("one device: RCCL takes one rank per device")
    single = str(tmp_path / "single.bin")
    p = _run([exe, "single", single])
    out, _ = p.communicate(timeout=300)
    assert p.returncode == 0, out
    idf = str(tmp_path / "id2")
    procs = [_run([exe, "rank", str(r), "2", idf, str(tmp_path / ("rank%d.bin" % r))]) for r in range(2)]
    for p in procs:
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0, out
    ref = open(single, "rb").read()
    for r in range(2):
        assert open(str(tmp_path / ("rank%d.bin" % r)), "rb").read() == ref
