"""SELL-64 with dictionary-coded columns (kernel format 5): in a locality-preserving state order
the entries of a 64-row chunk use a handful of distinct column offsets col - row, stored once per chunk, and
every entry carries a 6-bit index instead of a 4-byte column.  Values, the order in which a row is summed
(FMATVEC's, KrylovSolver.f90:598-604) and the gathered addresses are those of plain SELL: products must be
BIT-IDENTICAL to the plain form - on reference-assembled FSPs under the internal state order, on boxes, on a
non-box Goutsias FSP in search order, under a partition - while the generator moves fewer bytes."""
import os

import numpy as np
import pytest

from tests.conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _ctx(state_order, sell_code, group=None):
    from krylovfspssa_amd import KfspContext
    c = KfspContext(0, group=group)
    c.set_option("small_kernel", 0)
    c.set_option("state_order", state_order)
    c.set_option("state_order_min", 1)
    c.set_option("state_order_products", 0)
    c.set_option("sell_code", sell_code)
    return c


@pytest.mark.parametrize("name,k", [("goutsias", 16), ("goutsias", 10), ("repressilator", 10), ("toggle", 20)])
def test_coded_columns_give_the_bits_of_plain_sell_on_reference_fsps(name, k):
    a = np.load(os.path.join(GOLDEN, f"assembly_{name}_k{k}.npz"))
    adj, off, diag, state = a["adj"], a["offdiag"], a["diag"], a["state"]
    n = adj.shape[0]
    rng = np.random.default_rng(21)
    x = rng.random(n)
    p0 = rng.random(n)
    p0 /= p0.sum()
    res = {}
    for key, (so, sc) in {"plain": (0, 0), "ordered": (1, 0), "coded": (1, 1), "coded, caller order": (0, 1)}.items():
        with _ctx(so, sc) as c:
            c.set_state_coords(state)
            c.set_matrix_ell(adj, off, diag)
            info = c.layout_info()
            y = c.spmv(x)
            c.set_vector(p0)
            ws = c.expv_fixed(10, 0.01, 2)
            res[key] = dict(info=info, y=y, ws=ws, w=c.get_vector(), bytes=c.matrix_bytes(), plain_bytes=c.matrix_bytes(3))
    assert res["plain"]["info"]["format"] == 0 and res["ordered"]["info"]["format"] == 0
    assert res["coded"]["info"]["format"] == 5 and res["coded"]["info"]["state_order"] == 1
    # in the lexicographic order most chunks code even on these small, ragged FSPs (the others keep their columns)
    assert res["coded"]["info"]["coded_chunks"] >= 0.6 * res["coded"]["info"]["chunks"]
    assert res["coded"]["bytes"] < res["coded"]["plain_bytes"]
    for key in ("ordered", "coded", "coded, caller order"):
        assert np.array_equal(res[key]["y"], res["plain"]["y"]), key            # same bits, whatever the layout
    # the coded form IS the ordered form with other index bytes: everything, reductions included, is bit-identical
    assert np.array_equal(res["coded"]["ws"], res["ordered"]["ws"]) and np.array_equal(res["coded"]["w"], res["ordered"]["w"])
    assert np.abs(res["coded"]["w"] - res["plain"]["w"]).sum() < 1e-13


def test_coded_columns_on_boxes_and_a_wide_row():
    """boxes forced into SELL (every chunk has exactly the reaction shifts as offsets, + 0 for padding); the
    6-species network has rows of 12 entries: two code words per row"""
    from krylovfspssa_amd import synth
    for mdl in (synth.repressilator(dims=(31, 23, 19)), synth.birth_death((5, 6, 4, 5, 3, 4)), synth.toggle(97, 61)):
        x = np.random.default_rng(5).random(mdl.n)
        rowptr, col, val = mdl.csr_rows()
        ys = []
        for sc in (0, 1):
            with _ctx(0, sc) as c:
                c.set_option("format", 1)
                c.set_matrix_csr(mdl.n, rowptr, col, val)
                info = c.layout_info()
                assert info["format"] == (5 if sc else 0)
                if sc:
                    assert info["coded_chunks"] == info["chunks"]
                    assert c.matrix_bytes() < c.matrix_bytes(3)              # 4 B of column per entry gone, 8 B of codes per row added
                    ms = c.spmv_bench(3, 0), c.spmv_bench(3, 3)           # both kernels run (coded, plain columns of the same image)
                ys.append(c.spmv(x))
        assert np.array_equal(ys[0], ys[1]), mdl.name


@pytest.mark.parametrize("P", [1, 3])
def test_coded_columns_on_a_non_box_fsp_in_search_order(P):
    """a Goutsias FSP that is no box (ellipsoid x 6 DNA configurations, 69k states) listed in the order of a
    reachability search: discovery order does not code (too many distinct offsets per chunk: plain columns stay),
    the internal lexicographic order does; same bits either way, also row-partitioned over 3 ranks (bounded reach
    -> halo strips)."""
    from krylovfspssa_amd import synth
    g = synth.GoutsiasEllipsoid(center=(12, 10, 5), axes=(20, 18, 12))
    adj, off, diag = g.ell()
    x = np.random.default_rng(9).random(g.n)
    out = {}
    for key, (so, sc) in {"plain": (0, 0), "tried in search order": (0, 1), "coded": (1, 1)}.items():
        with _ctx(so, sc, group=P if P > 1 else None) as c:
            c.set_state_coords(g.state)
            c.set_matrix_ell(adj, off, diag)
            out[key] = (c.layout_info(), c.spmv(x), c.matrix_bytes())
    assert out["coded"][0]["format"] == 5 and out["coded"][0]["coded_chunks"] >= 0.9 * out["coded"][0]["chunks"]
    assert out["tried in search order"][0]["coded_chunks"] < 0.5 * out["tried in search order"][0]["chunks"]
    assert np.array_equal(out["coded"][1], out["plain"][1]) and np.array_equal(out["tried in search order"][1], out["plain"][1])
    assert out["coded"][2] < 0.85 * out["plain"][2]
    if P > 1:
        # a bounded reach (max |col - row| within one block) -> strips, else whole vectors: either way one was agreed
        assert out["coded"][0]["exchange"] in (1, 2) and out["plain"][0]["exchange"] in (1, 2)
