"""DROP_STATES on the device (kfsp_drop_plan / _flags / _compact) against
the reference: FIND_DROPTOL thresholds (fixture droptol.npz, StateSpace.f90:398-427) and the
whole decision + compaction on reference-assembled FSPs (fixtures drop_*, :431-548)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _spread_vector(n):
    """oracle/ref_dump.f90 SPREAD_VECTOR + the edits of DO_DROPTOL (1-based strides)"""
    i = np.arange(1, n + 1, dtype=np.float64)
    u = i * 0.6180339887498949
    u = u - np.trunc(u)
    w = 10.0 ** (-2.0 - 14.0 * u)
    w[6::13] = 0.0
    w[4::17] = -w[4::17]
    w[2::11] = w[2::11] * 1.0e-12
    return w


def test_find_droptol_thresholds(golden_dir):
    """ten mass bounds from 1e-1 to 1e-27 on a 50 000-vector with zeros, negative and tiny entries:
    the one-sweep threshold sums return the thresholds of the reference's sweep-per-threshold loop
    (the last bounds need a second batch of sixteen thresholds)."""
    from krylovfspssa_amd import KfspContext
    g = np.load(os.path.join(golden_dir, "droptol.npz"))
    n = 50000
    w = _spread_vector(n)
    # any generator will do for the thresholds: a chain
    rowptr = np.arange(0, 2 * n + 1, 2, dtype=np.int64)
    rowptr[-1] = 2 * n - 1
    col = np.empty(2 * n - 1, dtype=np.int32)
    val = np.empty(2 * n - 1)
    col[0::2] = np.arange(n)
    val[0::2] = -1.0
    col[1::2] = np.arange(1, n)
    val[1::2] = 1.0
    rowptr = np.concatenate(([0], np.cumsum(np.r_[np.full(n - 1, 2), 1]))).astype(np.int64)
    with KfspContext(0) as c:
        c.set_matrix_csr(n, rowptr, col, val)
        c.set_vector(w)
        got = [c.drop_plan(float(d))[0] for d in g["dsum"]]
    assert np.array_equal(np.array(got), g["droptol"])
    assert got[-1] < 1e-23                    # beyond the first sixteen thresholds


DROP_CASES = [("toggle", 20, 1e-6), ("goutsias", 16, 1e-12), ("repressilator", 10, 1e-4)]


@pytest.mark.parametrize("rebuild", [False, True])
@pytest.mark.parametrize("ranks", [1, 2, 3])
@pytest.mark.parametrize("state_order", [0, 1])
@pytest.mark.parametrize("name,k,dsum", DROP_CASES)
def test_drop_decision_and_compaction_match_the_reference(golden_dir, name, k, dsum, state_order, ranks, rebuild):
    """The FSP of `k` one-step sweeps as the reference assembles it (fixture assembly_*), the
    decaying vector of oracle/ref_dump.f90 DO_DROP, DROP_STATES on the device: the states kept and
    the compacted vector are what the reference leaves behind (fixture drop_*: its list starts
    with the kept states in order, its vector is the compacted W), the drop count obeys the
    reference's counting rule, and after kfsp_set_matrix_ell of the compacted FSP the resident
    vector IS the compacted one - also with the device keeping its own state order, and also with the FSP
    row-partitioned over 2 and 3 ranks (a group context over a loop-back group: per-rank threshold sums + one
    all-reduce, flags per block all-gathered, the compacted vector re-partitioned).  rebuild: the generator of the compacted
    FSP comes from the device's OWN copy of the reference arrays (kfsp_drop_rebuild: columns moved up, links renumbered
    through the keep-prefix-sum, dropped targets -> 0, StateSpace.f90:540-545) instead of being uploaded again - its products
    must be the bits of the uploaded one."""
    from krylovfspssa_amd import KfspContext
    from oracle import make_golden as MG
    a = np.load(os.path.join(golden_dir, f"assembly_{name}_k{k}.npz"))
    g = np.load(os.path.join(golden_dir, MG.drop_fixture_name(name, k, dsum)))
    adj, off, diag, state = a["adj"], a["offdiag"], a["diag"], a["state"]
    n = adj.shape[0]
    i = np.arange(1, n + 1, dtype=np.float64)
    w = 10.0 ** (-2.0 - 18.0 * (i - 1.0) / max(n - 1, 1))
    w[6::7] *= 1.0e3
    with KfspContext(0, group=ranks if ranks > 1 else None) as c:
        if state_order:
            c.set_option("state_order", 1)
            c.set_option("state_order_min", 1)
            c.set_option("state_order_products", 0)
            c.set_state_coords(state)
        c.set_matrix_ell(adj, off, diag)
        assert c.state_order_active() == bool(state_order)
        c.set_vector(w)
        aw = c.spmv_w()
        droptol, cnt, nflag = c.drop_plan(dsum)
        flags = c.drop_flags().astype(bool)
        # the reference's rule restated (StateSpace.f90:416-426, 475-497)
        tol = 1e-8
        while w[(w < tol) & (w > 0)].sum() >= dsum:
            tol = tol / 10.0
        assert droptol == tol
        near = np.abs(aw - 1e-8) < 1e-20 + 1e-12 * 1e-8       # guard decisions that hinge on rounding of A*w
        want = (w < tol) & ~(aw > 1e-8)
        assert np.array_equal(flags[~near], want[~near])
        assert cnt == int((w < tol).sum()) - int((aw > 1e-8).sum())
        assert nflag == int(flags.sum())
        if not cnt / n > 0.1:                                 # the 10 % rule (:497): the reference left this FSP alone
            assert np.array_equal(state, g["state"][:n])
            assert np.all(np.abs(w - g["vector"][:n]) <= 5e-16 * w)
            return
        keep = ~flags
        nk = int(keep.sum())
        assert np.array_equal(state[keep], g["state"][:nk])
        # (the vector is rebuilt here with numpy's pow, the fixture's with the Fortran runtime's: last-bit differences)
        assert np.all(np.abs(w[keep] - g["vector"][:nk]) <= 5e-16 * g["vector"][:nk]) and not g["vector"][nk:].any()
        # compaction: w stays on the device, the host re-links its lists (done here in numpy)
        assert c.drop_compact() == nk
        newidx = np.zeros(n + 1, dtype=np.int32)
        newidx[1:][keep] = np.arange(1, nk + 1)
        adj2 = adj[keep].copy()
        pos = adj2 > 0
        adj2[pos] = newidx[adj2[pos]]
        if rebuild:
            c.drop_rebuild()
            assert c.state_order_active() == bool(state_order)
        else:
            if state_order:
                c.set_state_coords(state[keep])
            c.set_matrix_ell(adj2, off[keep], diag[keep])
        assert c.n == nk
        assert np.array_equal(c.get_vector(), w[keep])
        if rebuild:
            # the rebuilt generator IS the generator of the compacted arrays: same products as a fresh upload of them
            xk = np.random.default_rng(5).random(nk)
            y = c.spmv(xk)
            with KfspContext(0) as fresh:
                fresh.set_matrix_ell(adj2, off[keep], diag[keep])
                assert np.array_equal(y, fresh.spmv(xk))
        # and the solver goes on from it
        beta = c.begin_step()
        assert beta == pytest.approx(np.sqrt((w[keep] ** 2).sum()), rel=1e-14)


def test_no_compaction_leaves_everything_in_place(golden_dir):
    """below the 10 % rule nothing is flagged for the host and w is untouched"""
    from krylovfspssa_amd import KfspContext
    a = np.load(os.path.join(golden_dir, "assembly_goutsias_k10.npz"))
    n = a["adj"].shape[0]
    w = np.full(n, 1.0 / n)
    with KfspContext(0) as c:
        c.set_matrix_ell(a["adj"], a["offdiag"], a["diag"])
        c.set_vector(w)
        droptol, cnt, nflag = c.drop_plan(1e-9)
        assert droptol <= 1e-8 and nflag == 0 and cnt <= 0
        assert np.array_equal(c.get_vector(), w)
