"""The internal state order (kfsp_set_state_coords; on by default for large long-lived generators): generator and vectors live in
lexicographic state order on the device, while every array that crosses the C ABI
stays in the caller's order.  Checked against a context that never received
coordinates and against the CPU oracle.  Needs a real MI355X."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pair(golden_dir, fixture):
    from krylovfspssa_amd import KfspContext
    g = np.load(os.path.join(golden_dir, fixture))
    plain, ordered = KfspContext(0), KfspContext(0)
    plain.set_option("state_order", 0)
    for c in (plain, ordered):
        c.set_option("format", 1)                 # SELL in both: the banded form sums diagonal by diagonal
    ordered.set_option("state_order", 1)
    ordered.set_option("state_order_min", 1)
    ordered.set_option("state_order_products", 0)
    plain.set_matrix_ell(g["adj"], g["offdiag"], g["diag"])
    ordered.set_state_coords(g["state"])
    ordered.set_matrix_ell(g["adj"], g["offdiag"], g["diag"])
    return g, plain, ordered


@pytest.mark.parametrize("fixture", ["assembly_goutsias_k16.npz", "assembly_repressilator_k10.npz",
                                     "solve_toggle_input.npz"])
def test_internal_order_is_invisible_at_the_boundary(oracle, golden_dir, fixture):
    g, plain, ordered = _pair(golden_dir, fixture)
    try:
        assert ordered.state_order_active() and not plain.state_order_active()
        n = len(g["diag"])
        rng = np.random.default_rng(3)
        x = rng.random(n)
        A = oracle.EllMatrix(g["adj"], g["offdiag"], g["diag"])
        scale = oracle.spmv_ell(oracle.EllMatrix(g["adj"], np.abs(g["offdiag"]), -np.abs(g["diag"])), x)
        # product of a host vector, caller's order in and out
        y = ordered.spmv(x)
        assert np.all(np.abs(y - oracle.spmv_ell(A, x)) <= 1e-13 * scale)
        # the probability vector round-trips untouched
        for c in (plain, ordered):
            c.set_vector(x)
        assert np.array_equal(ordered.get_vector(), x)
        # every row is summed in FMATVEC's order (entries sorted by the CALLER's column) in either
        # layout: the SELL products are the same bits
        assert np.array_equal(ordered.spmv_w(), plain.spmv_w())
        assert np.array_equal(y, plain.spmv(x))
        # one Arnoldi pass: same Hessenberg matrix up to the order of the sums, same basis columns
        b0, b1 = plain.begin_step(), ordered.begin_step()
        assert b1 == pytest.approx(b0, rel=1e-14)
        m = min(20, n - 2)
        H0, brk0, k0, av0 = plain.arnoldi(m)
        H1, brk1, k1, av1 = ordered.arnoldi(m)
        assert (brk0, k0) == (brk1, k1)
        assert np.abs(H1 - H0).max() <= 1e-11 * np.abs(H0).max()
        assert av1 == pytest.approx(av0, rel=1e-11)
        for j in (1, 2, m + 1):
            assert np.abs(ordered.get_basis(j) - plain.get_basis(j)).max() <= 1e-10
        # a few fixed steps end in the same vector
        for c in (plain, ordered):
            c.set_vector(x / x.sum())
        w0, w1 = plain.expv_fixed(12, 0.01, 3), ordered.expv_fixed(12, 0.01, 3)
        assert np.abs(w1 - w0).max() <= 1e-13
        assert np.abs(ordered.get_vector() - plain.get_vector()).sum() <= 1e-13
    finally:
        plain.close()
        ordered.close()


def test_coordinates_hold_for_one_generator_only(golden_dir):
    from krylovfspssa_amd import KfspContext
    g = np.load(os.path.join(golden_dir, "assembly_goutsias_k10.npz"))
    h = np.load(os.path.join(golden_dir, "assembly_goutsias_k16.npz"))
    c = KfspContext(0)
    try:
        c.set_option("state_order", 1)
        c.set_option("state_order_min", 1)
        c.set_option("state_order_products", 0)
        c.set_state_coords(g["state"])
        c.set_matrix_ell(g["adj"], g["offdiag"], g["diag"])
        assert c.state_order_active()
        c.set_matrix_ell(g["adj"], g["offdiag"], g["diag"])          # no fresh coordinates
        assert not c.state_order_active()
        c.set_state_coords(g["state"])
        c.set_matrix_ell(h["adj"], h["offdiag"], h["diag"])          # coordinates of another size
        assert not c.state_order_active()
        c.set_option("state_order", 0)
        c.set_state_coords(h["state"])
        c.set_matrix_ell(h["adj"], h["offdiag"], h["diag"])
        assert not c.state_order_active()
        c.set_option("state_order", 1)
        c.set_option("state_order_min", 10 ** 6)                     # below the size threshold
        c.set_state_coords(h["state"])
        c.set_matrix_ell(h["adj"], h["offdiag"], h["diag"])
        assert not c.state_order_active()
        c.set_option("state_order_min", 1)
        c.set_option("state_order_products", 5)                      # short-lived generators are left alone
        c.set_state_coords(h["state"])
        c.set_matrix_ell(h["adj"], h["offdiag"], h["diag"])
        assert not c.state_order_active()
        c.set_vector(np.ones(len(h["diag"])))
        for _ in range(6):
            c.spmv_w()
        c.set_state_coords(h["state"])
        c.set_matrix_ell(h["adj"], h["offdiag"], h["diag"])
        assert c.state_order_active()
    finally:
        c.close()


def test_shuffled_box(oracle):
    """A lexicographic box listed in random order (what SSA discovery order does to
    locality): with the coordinates the device works on the box order again, and
    the product agrees with the oracle in the caller's order."""
    from krylovfspssa_amd import KfspContext, synth
    mdl = synth.repressilator(dims=(40, 37, 33))
    adj, off, diag = mdl.ell()
    n = mdl.n
    rng = np.random.default_rng(11)
    perm = rng.permutation(n)                                        # new -> old
    iperm = np.empty(n, dtype=np.int64)
    iperm[perm] = np.arange(n)
    adj_s = adj[perm].copy()
    pos = adj_s > 0
    adj_s[pos] = (iperm[adj_s[pos] - 1] + 1).astype(np.int32)
    off_s, diag_s = off[perm], diag[perm]
    coords = np.stack(mdl.coords(np.arange(n)), axis=1)[perm].astype(np.int32)
    c = KfspContext(0)
    try:
        c.set_option("state_order", 1)
        c.set_option("state_order_min", 1)
        c.set_option("state_order_products", 0)
        c.set_state_coords(coords)
        c.set_matrix_ell(adj_s, off_s, diag_s)
        assert c.state_order_active()
        # six full diagonals over 128-row groups: what the banded form stores
        assert c.matrix_info()["slots"] == 6 * ((n + 127) // 128 * 128)
        x = rng.random(n)
        A = oracle.EllMatrix(adj_s, off_s, diag_s)
        scale = oracle.spmv_ell(oracle.EllMatrix(adj_s, np.abs(off_s), -np.abs(diag_s)), x)
        assert np.all(np.abs(c.spmv(x) - oracle.spmv_ell(A, x)) <= 1e-13 * scale)
    finally:
        c.close()


def test_sell_sigma_windows_keep_the_product_bit_identical(golden_dir):
    """option sell_sigma: inside windows of sigma rows of the internal order the longest rows first -
    fewer padded slots, the same bits in every row of the product (rows are still summed in the
    caller's column order) and through the vector transfers"""
    import os
    from krylovfspssa_amd import KfspContext
    g = np.load(os.path.join(golden_dir, "assembly_goutsias_k16.npz"))
    n = int(g["n"])
    x = np.random.default_rng(3).random(n)
    out = {}
    for sigma in (0, 128, 256):
        with KfspContext(0) as c:
            c.set_option("state_order", 1)
            c.set_option("state_order_min", 1)
            c.set_option("state_order_products", 0)
            c.set_option("small_kernel", 0)
            c.set_option("sell_sigma", sigma)
            c.set_state_coords(g["state"])
            c.set_matrix_ell(g["adj"], g["offdiag"], g["diag"])
            assert c.state_order_active()
            c.set_vector(x)
            c.begin_step()
            out[sigma] = (c.spmv_w(), c.matrix_info()["slots"], c.get_vector())
    assert np.array_equal(out[0][0], out[128][0]) and np.array_equal(out[0][0], out[256][0])
    assert np.array_equal(out[0][2], x) and np.array_equal(out[256][2], x)
    assert out[256][1] < out[128][1] < out[0][1]


def test_coordinates_of_a_grown_fsp_travel_incrementally(golden_dir):
    """kfsp_update_state_coords + kfsp_update_matrix_ell after the FSP GREW (one-step sweeps only append: the reference's
    assembly after 10 sweeps is a prefix of the one after 16): only the coordinates and propensity columns behind the first
    n_unchanged states travel, the state order is recomputed from the resident + new coordinates, and the generator is the one a
    fresh context builds from the whole arrays - products bit for bit."""
    from krylovfspssa_amd import KfspContext
    a = np.load(os.path.join(golden_dir, "assembly_goutsias_k10.npz"))
    b = np.load(os.path.join(golden_dir, "assembly_goutsias_k16.npz"))
    n0, n1 = a["adj"].shape[0], b["adj"].shape[0]
    assert n1 > n0 and np.array_equal(b["state"][:n0], a["state"])
    x = np.random.default_rng(4).random(n1)
    with KfspContext(0) as c, KfspContext(0) as fresh:
        for ctx in (c, fresh):
            ctx.set_option("state_order", 1)
            ctx.set_option("state_order_min", 1)
            ctx.set_option("state_order_products", 0)
        c.set_state_coords(a["state"])
        c.set_matrix_ell(a["adj"], a["offdiag"], a["diag"])
        assert c.state_order_active()
        c.update_state_coords(b["state"], n0)
        c.update_matrix_ell(b["adj"], b["offdiag"], b["diag"], n0)
        fresh.set_state_coords(b["state"])
        fresh.set_matrix_ell(b["adj"], b["offdiag"], b["diag"])
        assert c.state_order_active() and fresh.state_order_active()
        assert np.array_equal(c.spmv(x), fresh.spmv(x))
        c.set_vector(x)
        assert np.array_equal(c.get_vector(), x)
