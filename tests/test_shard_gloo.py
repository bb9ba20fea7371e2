"""N > 1 path of bench.py on CPU: two processes, gloo backend, 127.0.0.1."""
import os
import socket

import numpy as np
import torch
import torch.distributed as td
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    import blu_amd
    from blu_amd import shard
    from blu_amd.matrices import CONFIGS
    cfg = dict(CONFIGS["C4"], m=400)  # small stand-in for the 50k bases
    mine = shard.bases_of_rank(8, rank, world)
    nnz = 0
    sigs = []
    for b in mine:
        cp, ri, v = blu_amd.gen_lp_basis(cfg["m"], cfg["k"], cfg["bw"], cfg["tri_frac"], shard.seed_of_basis(cfg, b), cfg["offscale"])
        nnz += len(ri)
        sigs.append(float(np.abs(v).sum()))
    shard.fence()
    elapsed = 1.0 + rank  # rank 1 is the slow one
    value, t = shard.whole_job_throughput(nnz, elapsed)
    allsig = [None] * world
    td.all_gather_object(allsig, (mine, sigs, nnz))
    if rank == 0:
        out.put((value, t, allsig))
    td.barrier()
    td.destroy_process_group()


def test_two_ranks_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    value, t, allsig = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert t == 2.0  # max over ranks
    b0, b1 = allsig[0][0], allsig[1][0]
    assert sorted(b0 + b1) == list(range(8)) and not set(b0) & set(b1)  # partition, no overlap
    assert len(set(allsig[0][1] + allsig[1][1])) == 8  # eight different matrices
    total = allsig[0][2] + allsig[1][2]
    assert abs(value - total / 2.0) < 1e-9
