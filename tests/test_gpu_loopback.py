"""The row partition on ONE GPU: P contexts of this process form a loop-back group
(kfsp_comm_init_loopback, include/kfsp.h) whose collectives are device copies between host
barriers; everything around them is the code that runs over RCCL - strip packing, halo
margins, `rank > 0` / `rank + 1 < nranks` placement, the short last block, split interior /
boundary launches, staged scalars.  Products, Arnoldi passes and fixed-(m, tau) expv are
checked against the oracle on the whole problem."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as O
    return O


def _models():
    from krylovfspssa_amd import synth
    return {
        # c3-like box, 3 species / 6 reactions, n = 13 547: not a multiple of 64 * P for P = 2, 3, 4
        "repressilator": synth.repressilator(dims=(31, 23, 19)),
        # c4-like: Goutsias on M, D, RNA x the 6 conserved DNA configurations, diagonals with empty groups
        "goutsias": synth.GoutsiasConserved(20, 16, 12),
        # c2-like: wide reach (one species step = 211 rows)
        "toggle": synth.toggle(300, 211),
    }


MODES = {"halo": {}, "halo+split": {"overlap": 2}, "strips between neighbours": {"halo_p2p": 1},
         "strips between neighbours+split": {"halo_p2p": 1, "overlap": 2}, "allgather": {"halo": 0},
         # SELL-64 rows of the same boxes: their reach max |col - row| is one species stride, so they exchange
         # halo strips like the banded form (round 3); the whole-vector all-gather stays for unbounded reach
         "sell": {"format": 1}, "sell+split": {"format": 1, "overlap": 2}, "sell+allgather": {"format": 1, "halo_sell": 0},
         "sell coded": {"format": 1, "sell_code": 1}, "sell coded+split": {"format": 1, "sell_code": 1, "overlap": 2}}
# (kernel format, exchange) kfsp_layout_info must report: format 0 SELL / 5 coded SELL / 1, 2 banded; exchange 1 strips / 2 all-gather
EXPECT = {"halo": ((1, 2), 1), "halo+split": ((1, 2), 1), "strips between neighbours": ((1, 2), 1),
          "strips between neighbours+split": ((1, 2), 1), "allgather": ((1, 2), 2), "sell": ((0,), 1), "sell+split": ((0,), 1),
          "sell+allgather": ((0,), 2), "sell coded": ((5,), 1), "sell coded+split": ((5,), 1)}


@pytest.mark.parametrize("P", [2, 3, 4])
@pytest.mark.parametrize("model", ["repressilator", "goutsias", "toggle"])
@pytest.mark.parametrize("mode", list(MODES))
def test_partitioned_product_arnoldi_expv_match_the_oracle(oracle, P, model, mode):
    from krylovfspssa_amd import host, synth
    mdl = _models()[model]
    n = mdl.n
    adj, off, diag = mdl.ell()
    A = oracle.EllMatrix(adj, off, diag)
    p0 = np.random.default_rng(11).random(n)
    p0 /= p0.sum()
    m, tau, nsteps = 18, 0.004, 2

    def body(ctx, rank):
        ctx.set_option("small_kernel", 0)
        for k, v in MODES[mode].items():
            ctx.set_option(k, v)
        r0, nr = ctx.row_block(n)
        rp, cc, vv = mdl.csr_rows(r0, nr)
        ctx.set_matrix_csr(n, rp, cc, vv)
        info = ctx.layout_info()
        rows = np.repeat(np.arange(r0, r0 + nr), np.diff(rp))
        info["reach_of_my_rows"] = int(np.abs(cc.astype(np.int64) - rows).max()) if nr else 0
        ctx.set_vector(p0[r0:r0 + nr])
        y = ctx.spmv_w()
        beta = ctx.begin_step()
        H, mb, k1, av = ctx.arnoldi(m)
        v5 = ctx.get_basis(5)
        ctx.set_vector(p0[r0:r0 + nr])
        ws = ctx.expv_fixed(m, tau, nsteps)
        return dict(r0=r0, nr=nr, y=y, beta=beta, H=H.copy(), mb=mb, k1=k1, av=av, v5=v5, ws=ws.copy(), w=ctx.get_vector(),
                    info=info)

    res = host.run_loopback_ranks(P, body)
    # which kernel format and which exchange actually ran (agreed by all ranks)
    # (strips need the reach max |col - row| to stay within one block; a wider generator - the six DNA
    # configurations of the Goutsias set over 4 ranks - takes the whole-vector all-gather whatever was asked)
    reach = max(r["info"]["reach_of_my_rows"] for r in res)
    want_exchange = EXPECT[mode][1] if reach <= host.partition(n, P, 0)[2] else 2
    for r in res:
        if r["nr"] > 0:
            assert r["info"]["format"] in EXPECT[mode][0], r["info"]
        assert r["info"]["exchange"] == want_exchange, r["info"]
        if mode.startswith("sell") and want_exchange == 1 and r["nr"] > 0:
            assert 0 < r["info"]["sell_reach"] <= r["info"]["halo_rows"] <= host.partition(n, P, 0)[2]
        if mode.startswith("sell coded") and r["nr"] > 0:
            assert r["info"]["coded_chunks"] == r["info"]["chunks"] and r["info"]["code_words"] > 0
    # the blocks tile [0, n): equal padded length, short (possibly empty) last block
    L = host.partition(n, P, 0)[2]
    assert [r["r0"] for r in res] == [min(k * L, n) for k in range(P)]
    assert sum(r["nr"] for r in res) == n and res[-1]["nr"] <= L
    y = np.concatenate([r["y"] for r in res])
    yref = oracle.spmv_ell(A, p0)
    scale = oracle.spmv_ell(oracle.EllMatrix(adj, np.abs(off), -np.abs(diag)), np.abs(p0))
    assert np.all(np.abs(y - yref) <= 1e-13 * np.abs(scale) + 1e-300)
    # every rank holds the same scalars
    for r in res[1:]:
        assert r["beta"] == res[0]["beta"] and r["av"] == res[0]["av"] and np.array_equal(r["H"], res[0]["H"])
        assert np.array_equal(r["ws"], res[0]["ws"])
    V, Href, mb, k1, av = oracle.arnoldi(A, p0 / np.sqrt((p0 * p0).sum()), m)
    H = res[0]["H"]
    assert (res[0]["mb"], res[0]["k1"]) == (mb, k1)
    assert np.abs(H[:m + 1, :m] - Href[:m + 1, :m]).max() <= 1e-11 * np.abs(Href).max()
    assert abs(res[0]["av"] - av) <= 1e-10 * av
    v5 = np.concatenate([r["v5"] for r in res])
    assert np.abs(v5 - V[:, 4]).max() <= 1e-10
    wref, wsref = oracle.expv_fixed(A, p0, m, tau, nsteps)
    w = np.concatenate([r["w"] for r in res])
    assert np.abs(w - wref).sum() < 1e-10
    assert np.abs(res[0]["ws"] - wsref).max() < 1e-12


@pytest.mark.parametrize("P", [2, 3])
def test_partitioned_reference_layout_input(oracle, golden_dir, P):
    """kfsp_set_matrix_ell with a communicator: every rank receives the whole FSP_MATRIX arrays and
    builds its own row block on the device (global column indices, row0 /= 0); the source vector
    travels by all-gather (hash-ordered FSP: not banded)."""
    import os
    from krylovfspssa_amd import host
    g = np.load(os.path.join(golden_dir, "assembly_goutsias_k16.npz"))
    adj, off, diag = g["adj"], g["offdiag"], g["diag"]
    n = adj.shape[0]
    A = oracle.EllMatrix(adj, off, diag)
    x = np.random.default_rng(3).random(n)

    def body(ctx, rank):
        ctx.set_matrix_ell(adj, off, diag)
        r0, nr = ctx.row_block(n)
        ctx.set_vector(x[r0:r0 + nr])
        y = ctx.spmv_w()
        beta = ctx.begin_step()
        H, mb, k1, av = ctx.arnoldi(12)
        return dict(y=y, beta=beta, H=H.copy(), av=av)

    res = host.run_loopback_ranks(P, body)
    y = np.concatenate([r["y"] for r in res])
    yref = oracle.spmv_ell(A, x)
    scale = oracle.spmv_ell(oracle.EllMatrix(adj, np.abs(off), -np.abs(diag)), np.abs(x))
    assert np.all(np.abs(y - yref) <= 1e-13 * np.abs(scale) + 1e-300)
    V, Href, mb, k1, av = oracle.arnoldi(A, x / np.sqrt((x * x).sum()), 12)
    assert np.abs(res[0]["H"][:13, :12] - Href[:13, :12]).max() <= 1e-11 * np.abs(Href).max()
    assert all(np.array_equal(r["H"], res[0]["H"]) for r in res)
