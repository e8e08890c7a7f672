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
