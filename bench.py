#!/usr/bin/env python3
"""bench.py -- factorize throughput of the MI355X hot path (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one factorize (singletons -> setup_bump -> pivot loop -> build/read-out on the device) of
one synthetic LP basis per GPU, with B already resident in HBM when the clock starts.  The workload
is BASELINE.json configs[2] (C3: 100k x 100k, 10 nnz/col).  With N GPUs every rank factorizes its own
basis (seed = 1 + rank): independent matrices, no data-path collective; RCCL is used
only for the barriers around the timed region and the max-over-ranks of the elapsed time.

Prints ONE JSON line (rank 0).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import blu_amd  # noqa: E402
from blu_amd import keys as K, shard  # noqa: E402
from blu_amd.matrices import CONFIGS  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def cpu_baseline(cp, ri, v, budget_s=12.0, max_reps=12):
    """The CPU oracle (C restatement of the reference; the Rust crate cannot be built here) timed on ONE
    host core on the same matrix.  Bounded sample."""
    from oracle import orc  # baseline leg only
    orc.build()
    m = len(cp) - 1
    times = []
    t_all = time.time()
    o = orc.OracleBLU(m, 16 * len(ri))
    while len(times) < max_reps and (time.time() - t_all < budget_s or len(times) < 2):
        t0 = time.perf_counter()
        st = o.factorize(cp[:-1], cp[1:], ri, v)
        times.append(time.perf_counter() - t0)
        assert st == 0, st
    times = sorted(times[1:] if len(times) > 1 else times)  # drop the warm-up run
    med = times[len(times) // 2]
    return {"value": len(ri) / med, "unit": "nnz/s", "cores": 1, "kind": "port",
            "sample": "C3 basis, %d factorizations after 1 warm-up, median %.3f s each (single-threaded C restatement "
                      "of blu 0.2.1 incl. its always-on consistency passes; reference crate not executable here)" % (len(times), med),
            "seconds_per_factorize": med}


def batched_throughput(args, c, dev, local_rank, world):
    """SECONDARY measurement (never `value`): B independent bases of the same configuration in flight
    on this GPU, one workgroup per basis (blu_hip_factorize_batch).  A single factorize is a chain of
    dependent pivots and can not use more than one CU; this is the mode in which the chip fills up.
    8 distinct matrices (seeds) are cycled over the B handles; inputs resident in HBM."""
    B = args.batch
    nd = min(B, 8)
    mats = []
    for s in range(nd):
        cp, ri, v = blu_amd.gen_lp_basis(c["m"], c["k"], c["bw"], c["tri_frac"], 1000 + s, c["offscale"])
        mats.append((torch.from_numpy(cp.view(np.int64)).to(dev), torch.from_numpy(ri.view(np.int64)).to(dev),
                     torch.from_numpy(v).to(dev), len(ri)))
    hs = [blu_amd.BLU(c["m"], mats[k % nd][3] // 2, device=local_rank) for k in range(B)]
    ptrs = [(mats[k % nd][0].data_ptr(), mats[k % nd][0].data_ptr() + 8, mats[k % nd][1].data_ptr(),
             mats[k % nd][2].data_ptr(), mats[k % nd][3]) for k in range(B)]
    nnz = sum(p[4] for p in ptrs)
    best = None
    for rep in range(3):  # rep 0 warms up (storage growth), best of the other two
        shard.fence(dev)
        t0 = time.perf_counter()
        st = blu_amd.factorize_batch(hs, device_ptrs=ptrs, block=args.batch_block)
        shard.fence(dev)
        el = time.perf_counter() - t0
        if any(s != K.OK for s in st):
            raise RuntimeError("batched factorize failed: %r" % (st,))
        if rep > 0 and (best is None or el < best[0]):
            F = sum(h.stat(K.STAT_FACTOR_FLOPS) for h in hs)
            lu = sum(h.stat(K.STAT_L_NZ) + h.stat(K.STAT_U_NZ) for h in hs)
            best = (el, hs[0].stat(K.STAT_DEV_TIME_PIVOT_LOOP), int(hs[0].stat(K.STAT_DEV_RELAUNCHES)), F, lu)
    el = shard.max_over_ranks(best[0], dev)
    t_piv, nl, F, lu = best[1], best[2], best[3], best[4]
    gbs = (32.0 * F + 32.0 * lu) / t_piv / 1e9
    for h in hs:
        h.close()
    return {"bases_in_flight_per_gpu": B, "workgroup_threads": args.batch_block, "nnz_per_s": world * nnz / el,
            "seconds": el, "pivot_kernel_seconds": t_piv, "pivot_kernel_launches": nl,
            "roofline": {"bound": "hbm", "kernel": "k_pivot_loop_batch (grid = %d workgroups)" % B, "achieved": gbs,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None},
            "note": "throughput mode, reported beside the headline; `value` above is ONE basis per GPU"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C3")
    ap.add_argument("--block", type=int, default=0, help="workgroup size of the pivot kernel (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch", type=int, default=1280, help="bases in flight for the secondary throughput measurement (0 = skip)")
    ap.add_argument("--batch-block", type=int, default=256, help="workgroup size of the pivot kernel in batch mode")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = world > 1
    if dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import torch.distributed as td
        torch.cuda.set_device(local_rank)
        td.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    c = dict(CONFIGS[args.config])
    c["seed"] = shard.seed_of_basis(c, shard.bases_of_rank(world, rank, world)[0])  # independent bases, one per GPU
    cp, ri, v = blu_amd.gen_lp_basis(c["m"], c["k"], c["bw"], c["tri_frac"], c["seed"], c["offscale"])
    m, nnz = c["m"], len(ri)

    # inputs resident in HBM before the clock starts (uint64 bit patterns carried in int64 tensors)
    d_cp = torch.from_numpy(cp.view(np.int64)).to(dev)
    d_ri = torch.from_numpy(ri.view(np.int64)).to(dev)
    d_v = torch.from_numpy(v).to(dev)
    p_begin, p_end = d_cp.data_ptr(), d_cp.data_ptr() + 8
    h = blu_amd.BLU(m, nnz, device=local_rank)
    if args.block:
        h.dbg_set_block(args.block)

    def step():
        st = h.factorize_device(p_begin, p_end, d_ri.data_ptr(), d_v.data_ptr(), nnz)
        if st != K.OK:
            raise RuntimeError("factorize status %d" % st)

    def fence():
        shard.fence(dev)  # barrier (RCCL) + torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    t_pivot = 0.0
    t_dev = 0.0
    nlaunch = 0
    for _ in range(args.steps):
        step()
        t_pivot += h.stat(K.STAT_DEV_TIME_PIVOT_LOOP)  # HIP events on the library's stream, around k_pivot_loop
        t_dev += h.stat(K.STAT_DEV_TIME_TOTAL)
        nlaunch += int(h.stat(K.STAT_DEV_RELAUNCHES))
    fence()
    elapsed = shard.max_over_ranks(time.perf_counter() - t0, dev)

    F = h.stat(K.STAT_FACTOR_FLOPS)
    l_nz, u_nz = h.stat(K.STAT_L_NZ), h.stat(K.STAT_U_NZ)
    # algorithmic bytes: every multiply-add reads and rewrites one 16-byte (index,value)
    # entry; every L/U off-diagonal is read from the active submatrix and written to its factor once
    bytes_elim = 32.0 * F + 32.0 * (l_nz + u_nz)
    bytes_all = 16.0 * (nnz + m) + bytes_elim
    t_kernel = t_pivot / max(1, nlaunch)  # average k_pivot_loop launch
    # HBM traffic of the same kernel from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of this command,
    # recorded under profiles/ (PMC can not be collected from inside this script)
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01_pivot_loop_traffic.json")
    if args.config == "C3" and os.path.exists(tpath):
        traffic = json.load(open(tpath))["hbm_bytes_per_launch"] / max(t_kernel, 1e-12) / 1e9
    achieved = bytes_elim * args.steps / max(t_pivot, 1e-12) / 1e9
    out = {
        "metric": "factorize nnz/s + achieved HBM GB/s, 100k x 100k 10-nnz/col basis",
        "value": world * nnz * args.steps / elapsed,
        "unit": "nnz/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "%s: single %dx%d synthetic LP basis per GPU (lp_basis k=%d bw=%d tri_frac=%g offscale=%g seed=1+rank), "
                               "nnz=%d, inputs resident in HBM" % (args.config, m, m, c["k"], c["bw"], c["tri_frac"], c["offscale"], nnz),
                   "m": m, "nnz": nnz, "l_nz": l_nz, "u_nz": u_nz, "factor_flops": F,
                   "rank": h.stat(K.STAT_RANK), "bump_size": h.stat(K.STAT_BUMP_SIZE),
                   "nsearch_pivot": h.stat(K.STAT_NSEARCH_PIVOT), "parallelism": "one basis per GPU, no data-path collective"},
        "roofline": {"bound": "hbm", "kernel": "k_pivot_loop", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_source": "profiles/r01_pivot_loop_traffic.json (rocprofv3 FETCH_SIZE + WRITE_SIZE per launch / avg launch time, GB/s)",
                     "algorithmic_bytes_per_launch": bytes_elim * args.steps / max(1, nlaunch),
                     "avg_launch_ms": 1e3 * t_kernel, "launches_per_step": nlaunch / args.steps},
        "achieved_GBs_whole_factorize": bytes_all * args.steps / elapsed / 1e9,
        "device_ms_per_step": 1e3 * t_dev / args.steps,
    }
    if args.batch > 0:
        out["batched"] = batched_throughput(args, c, dev, local_rank, world)
    if rank == 0 and not args.no_cpu_baseline:
        cp0, ri0, v0 = (cp, ri, v) if not dist else blu_amd.gen_lp_basis(c["m"], c["k"], c["bw"], c["tri_frac"], CONFIGS[args.config]["seed"], c["offscale"])
        out["cpu_baseline"] = cpu_baseline(cp0, ri0, v0)
    if dist:
        td.barrier()
        td.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
