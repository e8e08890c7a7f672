"""The watchdog of a group context (include/kfsp.h kfsp_create_group, csrc/kfsp_group.cpp Group::run) on its own: worker
threads over the loop-back transport, no device anywhere - runs without a GPU.  A rank that fails before a collective,
or hangs, must not keep the caller: the peers are released by aborting the transport, the failing rank is named, the
group is broken (later calls return 2999 at once), and a rank that never returns is abandoned (2998).

The same fan-out with real contexts on the one GPU (a failure injected into rank 1 while the others run into the
collectives of set_matrix_ell) is tests/test_gpu_group.py::test_a_failing_rank_is_reported_and_does_not_hang."""
import ctypes as C

import pytest


@pytest.fixture(scope="module")
def lib():
    from krylovfspssa_amd import host
    return host.load_library()


def _selftest(lib, nranks, failing, hanging, work_ms=20, hang_ms=0, timeout_ms=5000, grace_ms=200, settle_ms=2000):
    rc, who, brk, stk = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
    sec = C.c_double(0.0)
    fn = lib.kfsp_group_selftest
    fn.restype = C.c_int
    fn.argtypes = [C.c_int] * 8 + [C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int),
                                   C.POINTER(C.c_int)]
    ret = fn(nranks, failing, hanging, work_ms, hang_ms, timeout_ms, grace_ms, settle_ms, C.byref(rc), C.byref(who),
             C.byref(sec), C.byref(brk), C.byref(stk))
    assert ret == 0
    return rc.value, who.value, sec.value, bool(brk.value), bool(stk.value)


@pytest.mark.parametrize("nranks", [1, 2, 4, 8])
def test_a_healthy_fan_out_is_untouched(lib, nranks):
    rc, who, sec, broken, stuck = _selftest(lib, nranks, -1, -1)
    assert (rc, who, broken, stuck) == (0, -1, False, False) and sec < 2.0


@pytest.mark.parametrize("nranks,failing", [(2, 1), (2, 0), (4, 2), (8, 7)])
def test_a_rank_that_fails_before_the_collective_is_named_within_the_grace_period(lib, nranks, failing):
    """the peers sit in a collective the failing rank never enters (for ever, with RCCL; 120 s with the loop-back's own
    guard): after grace_ms the transport is aborted, they return, and the caller learns WHICH rank failed and its code"""
    rc, who, sec, broken, stuck = _selftest(lib, nranks, failing, -1, grace_ms=200, timeout_ms=60000)
    assert rc == -77 and who == failing
    assert broken and not stuck
    assert 0.15 < sec < 3.0, sec           # the grace period, not the 60 s deadline and not the 120 s barrier guard


def test_a_lone_rank_that_fails_needs_no_abort(lib):
    rc, who, sec, broken, stuck = _selftest(lib, 1, 0, -1)
    assert (rc, who, broken, stuck) == (-77, 0, False, False)


def test_a_hanging_rank_costs_the_deadline_and_is_blamed_on_nobody(lib):
    """no rank failed: the call's own deadline expires, the others are released (2999), the sleeper comes back within the
    settle period - broken, not stuck"""
    rc, who, sec, broken, stuck = _selftest(lib, 3, -1, 1, hang_ms=900, timeout_ms=400, settle_ms=3000)
    assert rc == 2999 and broken and not stuck
    assert 0.35 < sec < 3.0, sec


def test_a_rank_that_never_returns_is_abandoned(lib):
    """even the abort does not bring rank 2 back within settle_ms: the caller gets control back with 2998"""
    rc, who, sec, broken, stuck = _selftest(lib, 3, -1, 2, hang_ms=4000, timeout_ms=300, settle_ms=300)
    assert rc == 2998 and broken and stuck
    assert sec < 2.0, sec
