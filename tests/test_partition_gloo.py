"""The N > 1 path on CPU: world size 2 over gloo.

The row partition the library uses (kfsp_partition: equal contiguous blocks of
L = ceil(n/P) rounded to 64 rows, global index = rank*L + local index, global
column indices kept) is exercised with the communication pattern of the solver
- all-gather of the source slab before every product, all-reduce of every
scalar - and a CPU double for the local kernels (the oracle).  Each rank builds
only its own rows, as bench.py does.  The result must equal the one-process
oracle.  This file checks the partition ARITHMETIC and the exchange pattern on CPU
processes; the library's own exchange code (strip packing, halo margins, rank > 0
branches, split launches, staged scalars) runs with 2, 3 and 4 ranks on one GPU in
tests/test_gpu_loopback.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, dims, m, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from krylovfspssa_amd import host, synth
        from oracle import oracle as O
        mdl = synth.repressilator(dims=dims)
        n = mdl.n
        row0, nrows, L = host.partition(n, world, rank)
        rowptr, col, val = mdl.csr_rows(row0, nrows)          # only this rank's rows

        def allgather(slab):
            send = torch.zeros(L, dtype=torch.float64)
            send[:nrows] = torch.from_numpy(slab)
            recv = torch.zeros(world * L, dtype=torch.float64)
            dist.all_gather_into_tensor(recv, send)
            return recv.numpy()                               # index p*L + k == global row

        def allsum(x):
            t = torch.tensor([x], dtype=torch.float64)
            dist.all_reduce(t)
            return float(t.item())

        # halo exchange of the banded case (kfsp_api.cpp gather_source): every rank
        # contributes [first HALO rows | last HALO rows], one all-gather of the strips,
        # the previous rank's LAST and the next rank's FIRST strip go into the
        # margins; the product then only reads global indices in
        # [row0 - HALO, row0 + L + HALO)
        HALO = int(np.max(np.abs(col - np.repeat(np.arange(row0, row0 + nrows), np.diff(rowptr))))) if nrows else 1
        HALO = -(-max(HALO, 1) // 8) * 8

        def halo_gather(slab):
            full = np.zeros(L)
            full[:nrows] = slab
            send = torch.from_numpy(np.concatenate([full[:HALO], full[L - HALO:]]))
            recv = torch.zeros(world * 2 * HALO, dtype=torch.float64)
            dist.all_gather_into_tensor(recv, send)
            recv = recv.numpy()
            xg = np.full(world * L + 2 * HALO, np.nan)          # NaN = never delivered
            base = HALO                                          # xg[base + g] holds global index g
            xg[base + row0:base + row0 + L] = full
            if rank > 0:
                xg[base + row0 - HALO:base + row0] = recv[(rank - 1) * 2 * HALO + HALO:(rank - 1) * 2 * HALO + 2 * HALO]
            if rank + 1 < world:
                xg[base + row0 + L:base + row0 + L + HALO] = recv[(rank + 1) * 2 * HALO:(rank + 1) * 2 * HALO + HALO]
            return xg[base:]

        def spmv(slab):                                       # exchange, then local rows
            y = O.spmv_csr(rowptr, col, val, allgather(slab))
            if HALO <= L:
                yh = O.spmv_csr(rowptr, col, val, halo_gather(slab))
                assert np.array_equal(y, yh), "halo exchange delivers a different source vector"
            return y

        p0 = synth.poisson_p0(mdl, 6.0)[row0:row0 + nrows]
        beta = np.sqrt(allsum(float(p0 @ p0)))
        V = [p0 / beta]
        H = np.zeros((m + 2, m + 2))
        for j in range(1, m + 1):                             # IOP, q = 2 (KrylovSolver.f90:238-260)
            w = spmv(V[j - 1])
            for i in range(max(1, j - 1), j + 1):
                h = allsum(float(V[i - 1] @ w))
                w = w - h * V[i - 1]
                H[i - 1, j - 1] = h
            nrm = np.sqrt(allsum(float(w @ w)))
            H[j, j - 1] = nrm
            V.append(w / nrm)
        av = np.sqrt(allsum(float(np.sum(spmv(V[m]) ** 2))))
        full = allgather(V[m])
        if rank == 0:
            np.savez(out, H=H, av=av, vm=np.concatenate([full[p * L:p * L + host.partition(n, world, p)[1]]
                                                         for p in range(world)]), beta=beta)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dims", [(9, 8, 7), (16, 8, 1)])
def test_row_partitioned_arnoldi_equals_single_process(tmp_path, oracle, dims):
    from krylovfspssa_amd import build, synth
    build.build_lib()
    m = 8
    out = str(tmp_path / "dist.npz")
    mp.start_processes(_worker, args=(2, _free_port(), dims, m, out), nprocs=2, join=True, start_method="spawn")
    d = np.load(out)
    mdl = synth.repressilator(dims=dims)
    A = oracle.EllMatrix(*mdl.ell())
    p0 = synth.poisson_p0(mdl, 6.0)
    V, H, mb, k1, av = oracle.arnoldi(A, p0 / np.sqrt(p0 @ p0), m)
    assert np.abs(d["H"][:m + 1, :m] - H[:m + 1, :m]).max() <= 1e-12 * np.abs(H).max()
    assert d["av"] == pytest.approx(av, rel=1e-12)
    assert np.abs(d["vm"] - V[:, m]).max() < 1e-12


def test_partition_covers_every_row_once():
    from krylovfspssa_amd import build, host
    build.build_lib()
    for n in (1, 63, 64, 65, 1000, 5_000_211, 113_379_904):
        for world in (1, 2, 3, 4, 8):
            blocks = [host.partition(n, world, r) for r in range(world)]
            L = blocks[0][2]
            assert L % 64 == 0 and all(b[2] == L for b in blocks)
            assert sum(b[1] for b in blocks) == n
            pos = 0
            for r, (row0, nrows, _) in enumerate(blocks):
                assert row0 == min(r * L, n) and (nrows == 0 or row0 == pos)
                pos += nrows
