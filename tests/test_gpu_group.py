"""The adaptive solve under the row partition (SURVEY.md 8(e) + the FSP loop of KrylovSolver.f90:206-550): a GROUP
context (kfsp_create_group) is one head handle over P rank contexts - here the P contexts of a loop-back group on the
one GPU - that a serial caller drives like a single context.  Device operations through a head against the same
operations on one context, and the reference's adaptive workloads through the Fortran host (CME_SOLVE with
KFSP_NRANKS = P) against the single-rank run and the reference's fixtures."""
import os

import numpy as np
import pytest

from tests.conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _fsp(name, k):
    a = np.load(os.path.join(GOLDEN, f"assembly_{name}_k{k}.npz"))
    return a["adj"], a["offdiag"], a["diag"], a["state"]


@pytest.mark.parametrize("state_order", [0, 1])
@pytest.mark.parametrize("P", [2, 3])
@pytest.mark.parametrize("name,k", [("goutsias", 16), ("repressilator", 10), ("toggle", 20)])
def test_head_of_a_partition_behaves_like_one_context(name, k, P, state_order):
    """reference-assembled FSPs (discovery order: SELL rows, all-gather exchange; with the internal state order
    forced on, the GLOBAL lexicographic order under the partition): product, vectors in and out, Arnoldi pass,
    fixed-(m, tau) steps and norms through a head over P ranks == the same on one context."""
    from krylovfspssa_amd import KfspContext
    adj, off, diag, state = _fsp(name, k)
    n = adj.shape[0]
    rng = np.random.default_rng(11)
    x = rng.random(n)
    p0 = rng.random(n)
    p0 /= p0.sum()
    out = []
    for group in (None, P):
        with KfspContext(0, group=group) as c:
            c.set_option("small_kernel", 0)
            c.set_option("state_order", state_order)
            c.set_option("state_order_min", 1)
            c.set_option("state_order_products", 0)
            c.set_state_coords(state)
            c.set_matrix_ell(adj, off, diag)
            assert c.state_order_active() == bool(state_order)
            assert c.matrix_info()["rows"] == n
            y = c.spmv(x)
            c.set_vector(p0)
            assert np.array_equal(c.get_vector(), p0)
            yw = c.spmv_w()
            nrm, asum = c.nrm2_w(), c.asum_w()
            beta = c.begin_step()
            H, mb, k1, av = c.arnoldi(12)
            v3 = c.get_basis(3)
            c.set_vector(p0)
            ws = c.expv_fixed(12, 0.01, 3)
            out.append(dict(y=y, yw=yw, nrm=nrm, asum=asum, beta=beta, H=H.copy(), mb=mb, k1=k1, av=av, v3=v3, ws=ws,
                            w=c.get_vector()))
    a, b = out
    # every row is summed in FMATVEC's order whoever owns it: products are the same bits
    assert np.array_equal(a["y"], b["y"]) and np.array_equal(a["yw"], b["yw"])
    for key in ("nrm", "asum", "beta", "av"):
        assert abs(a[key] - b[key]) <= 1e-13 * abs(a[key]), key
    assert (a["mb"], a["k1"]) == (b["mb"], b["k1"])
    assert np.abs(a["H"][:9, :8] - b["H"][:9, :8]).max() <= 1e-11 * np.abs(a["H"]).max()
    assert np.abs(a["v3"] - b["v3"]).max() <= 1e-12
    assert np.abs(a["ws"] - b["ws"]).max() < 1e-13 and np.abs(a["w"] - b["w"]).sum() < 1e-12


def test_head_takes_whole_gather_rows_and_boxes():
    """kfsp_set_matrix_csr / kfsp_set_matrix_box through a head: each rank gets its block (banded form, halo
    strips), results as on one context"""
    from krylovfspssa_amd import KfspContext, synth
    mdl = synth.repressilator(dims=(31, 23, 19))
    rowptr, col, val = mdl.csr_rows()
    x = np.random.default_rng(3).random(mdl.n)
    res = []
    for group in (None, 3):
        with KfspContext(0, group=group) as c:
            c.set_matrix_csr(mdl.n, rowptr, col, val)
            y1 = c.spmv(x)
            c.set_matrix_box(mdl)
            y2 = c.spmv(x)
            c.set_matrix_box(mdl, store=True)
            y3 = c.spmv(x)
            c.set_vector(x / x.sum())
            ws = c.expv_fixed(10, 0.005, 2)
            res.append((y1, y2, y3, ws, c.get_vector()))
    for i in range(3):
        assert np.array_equal(res[0][i], res[1][i])
    assert np.abs(res[0][3] - res[1][3]).max() < 1e-13 and np.abs(res[0][4] - res[1][4]).sum() < 1e-12


def test_ranks_that_disagree_are_reported():
    """a head refuses vectors of the wrong size and names the failing rank"""
    from krylovfspssa_amd import KfspContext, KfspError, synth
    mdl = synth.toggle(40, 30)
    with KfspContext(0, group=2) as c:
        c.set_matrix_csr(mdl.n, *mdl.csr_rows())
        with pytest.raises(KfspError):
            c._chk(c._lib.kfsp_set_vector(c._h, mdl.n - 1, None), "kfsp_set_vector")
        c.set_vector(np.ones(mdl.n) / mdl.n)
        assert c.begin_step() == pytest.approx(1.0 / np.sqrt(mdl.n), rel=1e-13)


def test_a_failing_rank_is_reported_and_does_not_hang():
    """Rank 1 of a 3-rank loop-back group fails (injected: option group_inject_failure) at the start of set_matrix_ell while
    ranks 0 and 2 run into the collectives of the generator build, which rank 1 never enters.  The head's watchdog (Group::run)
    releases them after the grace period by aborting the transport, reports 'rank 1: ...' with rank 1's code, and the group
    is broken: the next call returns 2999 at once, destruction still works, and a new group on the same device is fine."""
    import time
    from krylovfspssa_amd import KfspContext, KfspError
    adj, off, diag, state = _fsp("goutsias", 16)
    n = adj.shape[0]
    c = KfspContext(0, group=3)
    try:
        c.set_option("group_grace_ms", 500)
        c.set_matrix_ell(adj, off, diag)                 # a healthy call first
        c.set_option("group_inject_failure", 1)
        t0 = time.time()
        with pytest.raises(KfspError, match=r"-77.*rank 1: injected failure"):
            c.set_matrix_ell(adj, off, diag)
        assert time.time() - t0 < 20.0                   # the grace period, not the 120 s guard of the loop-back barrier
        t0 = time.time()
        with pytest.raises(KfspError, match="2999"):
            c.set_vector(np.ones(n) / n)
        assert time.time() - t0 < 1.0
    finally:
        c.close()
    x = np.random.default_rng(5).random(n)
    with KfspContext(0, group=3) as g, KfspContext(0) as one:
        g.set_matrix_ell(adj, off, diag)
        one.set_matrix_ell(adj, off, diag)
        assert np.array_equal(g.spmv(x), one.spmv(x))
