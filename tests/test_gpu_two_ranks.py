"""RCCL with MORE THAN ONE RANK: the only part of the row partition the one-GPU boxes of the builder cannot run.
Skipped unless the box has two GPUs; there it starts 2 real ranks through bench.spawn_ranks (fresh child
processes, one per GPU; nothing is re-executed in this process) and checks product, Arnoldi pass and fixed-(m, tau)
exp(tA)v against the oracle for the strip all-gather, neighbour send/recv (halo_p2p), the split interior /
boundary launches, the whole-vector all-gather and SELL rows with halo strips - and that every rank holds
bit-identical scalars."""
import json
import os

import pytest
import torch

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL between real ranks)")
def test_two_real_ranks_over_rccl(capfd):
    import bench
    rc = bench.spawn_ranks(2, [], script=os.path.join(ROOT, "tests", "two_rank_child.py"), deadline_s=900)
    out = capfd.readouterr().out.strip().splitlines()
    assert rc == 0 and out, "a rank failed"
    rep = json.loads(out[-1])
    assert set(rep) == {"halo", "halo_p2p", "overlap", "allgather", "sell strips", "sell coded"}
    for name, r in rep.items():
        assert r["err_product"] <= 1e-13 and r["err_H"] <= 1e-11 and r["l1_expv"] < 1e-10 and r["err_ws"] < 1e-12, (name, r)
        assert r["breakdown"] and r["scalars_identical"], (name, r)
        assert r["exchange"] == (2 if name == "allgather" else 1), (name, r)
        assert r["bytes_in"] > 0
    assert rep["sell coded"]["format"] == 5 and rep["sell strips"]["format"] == 0
    # neighbours only: a rank receives its two strips, not everybody's
    assert rep["halo_p2p"]["bytes_in"] <= rep["halo"]["bytes_in"]


# The two tests below have NEVER executed: no multi-GPU box was available in any round.  They are written to pass, but a first
# run is a first run - they are reported as xfail / xpass instead of turning a suite red that nobody could have checked.
UNVERIFIED = pytest.mark.xfail(strict=False, reason="first execution on real multi-GPU hardware (RCCL from the threads of a group context)")


@UNVERIFIED
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (a group context on distinct devices: RCCL between its ranks)")
@pytest.mark.parametrize("state_order", [0, 1])
def test_group_context_on_two_devices(oracle, state_order, monkeypatch):
    """kfsp_create_group(2, [0, 1]): ONE host thread, two worker threads, each initialising its rank of an RCCL
    communicator (ncclCommInitRank from threads of one process) - the form the serial Fortran host uses with
    KFSP_NRANKS=2 KFSP_DEVICES=0,1.  Product, Arnoldi pass, fixed-(m, tau) exp(tA)v of a reference-assembled FSP against
    the oracle and against one context; comm_allgather_bytes (the drop flags) through a drop plan."""
    import numpy as np
    from krylovfspssa_amd import KfspContext
    from tests.conftest import GOLDEN
    monkeypatch.setenv("KFSP_GROUP_TIMEOUT_S", "120")            # (a first run: a hang must end as an error, not as a stuck suite)
    monkeypatch.setenv("KFSP_GROUP_GRACE_S", "20")
    a = np.load(os.path.join(GOLDEN, "assembly_goutsias_k16.npz"))
    adj, off, diag, state = a["adj"], a["offdiag"], a["diag"], a["state"]
    n = adj.shape[0]
    A = oracle.EllMatrix(adj, off, diag)
    rng = np.random.default_rng(3)
    x = rng.random(n)
    p0 = rng.random(n)
    p0 /= p0.sum()
    m, tau = 12, 0.01
    out = []
    for group in (None, [0, 1]):                                   # (a list of distinct devices: RCCL between the ranks)
        with KfspContext(0, group=group) as c:
            c.set_option("small_kernel", 0)
            c.set_option("state_order", state_order)
            c.set_option("state_order_min", 1)
            c.set_option("state_order_products", 0)
            c.set_state_coords(state)
            c.set_matrix_ell(adj, off, diag)
            y = c.spmv(x)
            c.set_vector(p0)
            beta = c.begin_step()
            H, mb, k1, av = c.arnoldi(m)
            c.set_vector(p0)
            ws = c.expv_fixed(m, tau, 3)
            w = c.get_vector()
            plan = c.drop_plan(1e-7)
            out.append(dict(y=y, H=H.copy(), mb=mb, k1=k1, av=av, ws=ws, w=w, plan=plan, beta=beta))
    one, two = out
    assert np.array_equal(one["y"], two["y"])                       # rows are summed in FMATVEC's order whoever owns them
    assert np.abs(two["y"] - oracle.spmv_ell(A, x)).max() <= 1e-13 * np.abs(off).max()
    V, Href, mbr, k1r, avr = oracle.arnoldi(A, p0 / np.sqrt((p0 * p0).sum()), m)
    assert (two["mb"], two["k1"]) == (mbr, k1r) and np.abs(two["H"] - Href).max() <= 1e-11 * np.abs(Href).max()
    wref, wsref = oracle.expv_fixed(A, p0, m, tau, 3)
    assert np.abs(two["w"] - wref).sum() < 1e-10 and np.abs(two["ws"] - wsref).max() < 1e-12
    assert two["plan"][1:] == one["plan"][1:] and abs(two["plan"][0] - one["plan"][0]) <= 1e-300 + 1e-12 * abs(one["plan"][0])


@UNVERIFIED
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (CME_SOLVE over KFSP_DEVICES=0,1)")
@pytest.mark.parametrize("mode", ["default", "resident"])
def test_cme_solve_on_two_devices(tmp_path, mode):
    """The Fortran host over two real devices (KFSP_NRANKS=2, KFSP_DEVICES=0,1): Goutsias T = 40 against the reference's
    fixture - state list and links bit for bit, probabilities to 1e-10 - in the default mode, and the resident mode against
    the one-device resident run."""
    import numpy as np
    from tests import test_fortran_host as TF
    dump = TF.DUMP
    if not os.path.exists(dump):
        pytest.skip("kfsp_dump not built")
    two = {"KFSP_NRANKS": "2", "KFSP_DEVICES": "0,1", "KFSP_GROUP_GRACE_S": "20", "KFSP_GROUP_TIMEOUT_S": "120"}
    if mode == "default":
        g, d, log = TF._solve(dump, tmp_path, "goutsias_input_T40", "goutsias_input", env=two)
        assert np.array_equal(log["step_n"], g["step_n"]) and np.array_equal(log["step_tau"], g["step_tau"])
        assert np.array_equal(d["state"], g["state"]) and np.array_equal(d["adj"], g["adj"])
        assert np.abs(d["vector"] - g["vector"]).sum() < 1e-10
    else:
        base = {"KFSP_SSA_STREAMS": "1"}
        g, d1, log1 = TF._solve(dump, tmp_path, "goutsias_input_T40", "goutsias_input", env=base)
        g, d, log = TF._solve(dump, tmp_path, "goutsias_input_T40", "goutsias_input", env=dict(base, **two))
        assert np.array_equal(log["step_n"], log1["step_n"]) and np.array_equal(log["step_tau"], log1["step_tau"])
        for key in ("state", "adj", "offdiag", "diag"):
            assert np.array_equal(d[key], d1[key]), key
        assert np.abs(d["vector"] - d1["vector"]).sum() < 1e-10
