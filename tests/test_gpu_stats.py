"""vo_stats_gather (include/vo_hip.h): the one collective of the sharded runs, through the C ABI with an RCCL
communicator the HOST made (here: a one-rank communicator built with ctypes on the RCCL that ships with torch -- a
non-Python host would call ncclCommInitRank the same way).  N > 1 ranks: tests/test_dist_cpu.py (gloo, CPU)."""
import ctypes as C
import os

import pytest

from visual_odometry_amd import _lib

pytestmark = pytest.mark.gpu


class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def test_stats_gather_single_rank_and_with_a_communicator(ctx):
    import torch
    local = _lib.Stats(12.0, 600000.0, 590000.0, 240000.0, 1.5)
    job = _lib.Stats()
    ctx.call("vo_stats_gather", None, C.byref(local), C.byref(job))          # no communicator: one rank
    assert [getattr(job, n) for n, _ in _lib.Stats._fields_] == [12.0, 600000.0, 590000.0, 240000.0, 1.5]
    path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    if not os.path.exists(path):
        pytest.skip("torch ships no librccl.so here")
    rccl = C.CDLL(path, mode=C.RTLD_GLOBAL)
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    torch.cuda.set_device(ctx.device)
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        job2 = _lib.Stats()
        ctx.call("vo_stats_gather", comm, C.byref(local), C.byref(job2))    # sum and max over the (one) rank
        assert [getattr(job2, n) for n, _ in _lib.Stats._fields_] == [12.0, 600000.0, 590000.0, 240000.0, 1.5]
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)
