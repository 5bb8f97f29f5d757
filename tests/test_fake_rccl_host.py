"""The stand-in collective library the GPU tests load through HIPETS_RCCL_LIB (tests/fake_rccl: N ranks = N processes sharing
one GPU) checked on its own, without a GPU: 8 processes, host buffers (FAKE_RCCL_HOST_BUFFERS=1), repeated all-gathers of
different sizes, the communicator's own rank / size answers, and the injected failure."""
import ctypes
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * 128)]


def _bind(path):
    lib = ctypes.CDLL(path)
    lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(UniqueId)]
    lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    lib.ncclAllGather.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
    lib.ncclCommCount.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
    lib.ncclCommUserRank.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
    lib.ncclGetErrorString.restype = ctypes.c_char_p
    return lib


def _rank(path, uid_bytes, rank, world, fail_at, q):
    os.environ["FAKE_RCCL_HOST_BUFFERS"] = "1"
    if fail_at:
        os.environ["FAKE_RCCL_FAIL_AT"] = str(fail_at)
    lib = _bind(path)
    uid = UniqueId()
    ctypes.memmove(ctypes.byref(uid), uid_bytes, 128)
    comm = ctypes.c_void_p()
    assert lib.ncclCommInitRank(ctypes.byref(comm), world, uid, rank) == 0
    n, r = ctypes.c_int(), ctypes.c_int()
    assert lib.ncclCommCount(comm, ctypes.byref(n)) == 0 and lib.ncclCommUserRank(comm, ctypes.byref(r)) == 0
    ok = (n.value, r.value) == (world, rank)
    codes = []
    for call, count in enumerate([63, 1, 250, 63], start=1):
        send = (np.arange(count, dtype=np.float32) + 1000.0 * rank + call).copy()
        recv = np.full(world * count, -1.0, np.float32)
        rc = lib.ncclAllGather(send.ctypes.data, recv.ctypes.data, count, 7, comm, None)
        codes.append(rc)
        if rc == 0:
            want = np.concatenate([np.arange(count, dtype=np.float32) + 1000.0 * k + call for k in range(world)])
            ok = ok and np.array_equal(recv, want)
    lib.ncclCommDestroy(comm)
    q.put((rank, ok, codes, lib.ncclGetErrorString(2).decode()))


@pytest.mark.parametrize("world,fail_at", [(8, 0), (2, 0), (5, 3)])
def test_fake_rccl_allgather_between_processes(world, fail_at):
    import __graft_entry__ as ge

    path = ge.build_fake_rccl()
    lib = _bind(path)
    uid = UniqueId()
    assert lib.ncclGetUniqueId(ctypes.byref(uid)) == 0
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank, args=(path, bytes(uid), r, world, fail_at, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok, codes, msg in res:
        assert ok, rank
        assert codes == [0 if (not fail_at or c != fail_at) else 2 for c in (1, 2, 3, 4)], (rank, codes)
        assert msg == "unhandled system error"
