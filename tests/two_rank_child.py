"""Child of tests/test_gpu_two_ranks.py: ONE rank of a real RCCL job (started through bench.spawn_ranks with the
environment torch.distributed.run would set).  Product, Arnoldi pass and fixed-(m, tau) exp(tA)v of a row-partitioned
generator against the oracle on the whole problem, for every exchange mode; rank 0 prints one JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

rank, world, local = (int(os.environ[k]) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"))
json_fd = os.dup(1)
os.dup2(2, 1)                                   # RCCL banners must not reach the relayed stdout
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch                                    # noqa: E402
import torch.distributed as dist                # noqa: E402

torch.cuda.set_device(local)
dist.init_process_group("nccl", device_id=torch.device("cuda", local))
from krylovfspssa_amd import KfspContext, synth  # noqa: E402
from oracle import oracle as O                   # noqa: E402  (test infrastructure: the checker)

mdl = synth.repressilator(dims=(40, 36, 30 * world))
n = mdl.n
adj, off, diag = mdl.ell()
A = O.EllMatrix(adj, off, diag)
p0 = np.random.default_rng(11).random(n)
p0 /= p0.sum()
m, tau, nsteps = 16, 0.004, 2
yref = O.spmv_ell(A, p0)
scale = O.spmv_ell(O.EllMatrix(adj, np.abs(off), -np.abs(diag)), np.abs(p0))
V, Href, mb, k1, av = O.arnoldi(A, p0 / np.sqrt((p0 * p0).sum()), m)
wref, wsref = O.expv_fixed(A, p0, m, tau, nsteps)
MODES = {"halo": {}, "halo_p2p": {"halo_p2p": 1}, "overlap": {"overlap": 2}, "allgather": {"halo": 0},
         "sell strips": {"format": 1}, "sell coded": {"format": 1, "sell_code": 1}}
report = {}
for name, opts in MODES.items():
    with KfspContext(local) as ctx:
        idt = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            idt.copy_(torch.from_numpy(KfspContext.unique_id()))
        dist.broadcast(idt, 0)
        ctx.comm_init(world, rank, idt.cpu().numpy())
        ctx.set_option("small_kernel", 0)
        for k, v in opts.items():
            ctx.set_option(k, v)
        r0, nr = ctx.row_block(n)
        ctx.set_matrix_csr(n, *mdl.csr_rows(r0, nr))
        info = ctx.layout_info()
        ctx.set_vector(p0[r0:r0 + nr])
        y = ctx.spmv_w()
        e_y = float(np.max(np.abs(y - yref[r0:r0 + nr]) / (np.abs(scale[r0:r0 + nr]) + 1e-300))) if nr else 0.0
        beta = ctx.begin_step()
        H, mb2, k12, av2 = ctx.arnoldi(m)
        e_h = float(np.abs(H[:m + 1, :m] - Href[:m + 1, :m]).max() / np.abs(Href).max())
        ctx.set_vector(p0[r0:r0 + nr])
        ws = ctx.expv_fixed(m, tau, nsteps)
        w = ctx.get_vector()
        t = torch.tensor([np.abs(w - wref[r0:r0 + nr]).sum()], dtype=torch.float64, device="cuda")
        dist.all_reduce(t)
        # scalars must be the same bits on every rank
        mine = torch.tensor(np.concatenate([[beta, av2], ws, H.ravel()]), dtype=torch.float64, device="cuda")
        ref0 = mine.clone()
        dist.broadcast(ref0, 0)
        same = torch.tensor([1.0 if torch.equal(mine, ref0) else 0.0], dtype=torch.float64, device="cuda")
        dist.all_reduce(same, op=dist.ReduceOp.MIN)
        ex_ms, ex_bytes = ctx.exchange_bench(20)
        report[name] = dict(err_product=e_y, err_H=e_h, l1_expv=float(t.item()), err_ws=float(np.abs(ws - wsref).max()),
                            breakdown=[mb2, k12] == [mb, k1], scalars_identical=bool(same.item() > 0.5),
                            exchange=info["exchange"], format=info["format"], halo_rows=info["halo_rows"],
                            exchange_us=ex_ms / 20 * 1e3, bytes_in=ex_bytes)
dist.barrier()
if rank == 0:
    os.write(json_fd, (json.dumps(report) + "\n").encode())
dist.destroy_process_group()
