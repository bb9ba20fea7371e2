#!/usr/bin/env python3
"""bench.py -- factorize throughput of the MI355X hot path (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W [--config C3|C2|C4]

A "step" is one factorize (singletons -> setup_bump -> pivot loop -> build/read-out -> statistics tail, all on
the device) of one synthetic LP basis per GPU, with B already resident in HBM when the clock starts.  The
default workload is BASELINE.json configs[2] (C3: 100k x 100k, 10 nnz/col); C4 is the 8 x 50k batch config
(basis b -> rank b mod N).  With N GPUs every rank factorizes its own basis (seed = 1 + rank): independent
matrices, no data-path collective; RCCL is used only for the barriers around the timed region
and the max-over-ranks of the elapsed time.

N > 1: when started WITHOUT a launcher (RANK unset) this script starts `python -m torch.distributed.run
--nproc-per-node N ... bench.py ...` as a CHILD process -- before anything of torch.cuda / the HIP library is
touched -- and exits with its return code.  Under a launcher WORLD_SIZE must equal --gpus.

Prints ONE JSON line (rank 0).
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "pivot_loop_traffic.json")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C3", choices=["C2", "C3", "C4", "C5"])
    ap.add_argument("--mods", type=int, default=1000, help="C5: column modifications per step")
    ap.add_argument("--block", type=int, default=0, help="workgroup size of the pivot kernel (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batch-sizes", action="store_true", help="skip the batched measurement on C4- and C2-size bases")
    ap.add_argument("--batch", type=int, default=1536, help="bases in flight for the secondary throughput measurement (0 = skip)")
    ap.add_argument("--no-verify", action="store_true", help="skip the untimed oracle check of the batched legs")
    ap.add_argument("--batch-block", type=int, default=256, help="workgroup size of the multi-wave pivot kernel in batch mode (BLU_PIVOT_KERNEL=2 only)")
    return ap.parse_args(argv)


def kernel_source_sha16():
    """Hash of the kernel sources the library is built from: the PMC traffic figure under profiles/ is only
    quoted while it was measured on exactly these sources."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "blu_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".inc")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def cpu_baseline(cp, ri, v, label, budget_s=12.0, max_reps=12):
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
            "sample": "%s basis, %d factorizations after 1 warm-up, median %.3f s each (single-threaded C restatement "
                      "of blu 0.2.1 incl. its always-on consistency passes; reference crate not executable here)" % (label, len(times), med),
            "seconds_per_factorize": med}


def _verify_members(hs, member_seed, c, be, which):
    """UNTIMED check of a batch as it was timed: members `which` against the CPU oracle (checker only; the faithful
    restatement, reference defect D3 included, so d3_hits must be 0 on both sides): the six integer arrays of get_factors
    and the values bit for bit, the counters, the pivots per pivot routine, every statistic of the tail."""
    import numpy as np
    from blu_amd import keys as K
    from oracle import orc  # checker only, outside every timed region
    orc.build()
    cache = {}
    for k in which:
        seed = member_seed(k)
        if seed not in cache:
            cp, ri, v = be.gen_lp_basis(c["m"], c["k"], c["bw"], c["tri_frac"], seed, c["offscale"])
            # first with the 64-bit cancellation mask, which counts the events the reference's i32 mask would mishandle
            # (defect D3, pivot.rs:645-659); none (every seed of the three legs so far): the FAITHFUL restatement is the checker
            o = orc.OracleBLU(c["m"], 16 * len(ri))
            o.set_fix_d3(True)
            if o.factorize(cp[:-1], cp[1:], ri, v) != K.OK:
                raise RuntimeError("oracle run of seed %d failed" % seed)
            if o.d3_hits() == 0:
                o = orc.OracleBLU(c["m"], 16 * len(ri))
                if o.factorize(cp[:-1], cp[1:], ri, v) != K.OK or o.d3_hits() != 0:
                    raise RuntimeError("faithful oracle run of seed %d failed" % seed)
            cache[seed] = (o, o.get_factors())
        o, fo = cache[seed]
        h = hs[k]
        fg = h.get_factors()
        for key in ("rowperm", "colperm", "l_colptr", "l_rowidx", "u_colptr", "u_rowidx", "l_value", "u_value"):
            if not np.array_equal(fg[key], fo[key]):
                raise RuntimeError("batched factorize: member %d (seed %d) differs from the oracle in %s" % (k, seed, key))
        keys = [K.STAT_RANK, K.STAT_MATRIX_NZ, K.STAT_BUMP_SIZE, K.STAT_BUMP_NZ, K.STAT_L_NZ, K.STAT_U_NZ, K.STAT_NSEARCH_PIVOT,
                K.STAT_FACTOR_FLOPS, K.STAT_RANKDEF, 50, 51, 52, 53, 54, 55, 56, K.STAT_MIN_PIVOT, K.STAT_MAX_PIVOT, K.STAT_CONDEST_L,
                K.STAT_CONDEST_U, K.STAT_NORM_L, K.STAT_NORM_U, K.STAT_NORMEST_L_INV, K.STAT_NORMEST_U_INV, K.STAT_ONENORM,
                K.STAT_INFNORM, K.STAT_RESIDUAL_TEST]
        for key in keys:
            if h.stat(key) != o.stat(key):
                raise RuntimeError("batched factorize: member %d (seed %d) differs from the oracle in statistic %d" % (k, seed, key))
    return len(which)


def batch_setup(c, B, dev, local_rank, be, hintdiv=2, nd=64):
    """B handles of configuration c over min(B, nd) distinct matrices (seeds 1000 + k mod nd), every handle with device
    copies of its inputs of its own.  Returns (handles, device pointers, the tensors that own the inputs, seed of member k,
    number of distinct matrices, total nnz).  Also used by tools/batch_probe.py, so that probes and PMC passes run what
    bench.py times."""
    import numpy as np
    import torch
    nd = min(B, nd)

    def member_seed(k):
        return 1000 + k % nd

    mats = [be.gen_lp_basis(c["m"], c["k"], c["bw"], c["tri_frac"], member_seed(s), c["offscale"]) for s in range(nd)]
    inputs, ptrs = [], []
    for k in range(B):
        cp, ri, v = mats[k % nd]
        t = (torch.from_numpy(cp.view(np.int64)).to(dev), torch.from_numpy(ri.view(np.int64)).to(dev), torch.from_numpy(v).to(dev))
        inputs.append(t)  # (kept alive: the handles borrow them for every call)
        ptrs.append((t[0].data_ptr(), t[0].data_ptr() + 8, t[1].data_ptr(), t[2].data_ptr(), len(ri)))
    hs = [be.BLU(c["m"], len(mats[k % nd][1]) // hintdiv, device=local_rank) for k in range(B)]
    return hs, ptrs, inputs, member_seed, nd, sum(p[4] for p in ptrs)


def batched_throughput(args, c, dev, local_rank, world, be, B=None, cfg_name=None):
    """SECONDARY measurement (never `value`): B independent bases of the same configuration in flight
    on this GPU (blu_hip_factorize_batch): one wave per basis in the pivot kernel (k_pivot_loop_wave), two while the card
    holds no more than half the bases its registers would allow (k_pivot_loop_wave2: bases of the 100k size).  A
    single factorize is a chain of dependent pivots and can not use more than one CU; this is the mode in
    which the chip fills up.  64 distinct matrices (seeds 1000..1063) over the B handles, EVERY handle with device
    inputs of its own (no two workgroups read the same copy of B); the first step (cold: storage growth, compaction
    rounds, relaunches) is reported on its own, then the MEDIAN of three warm steps; after the timed steps, untimed,
    eight members (first, last, six in between) are checked against the CPU oracle (`verified_members`)."""
    import numpy as np
    import torch
    from blu_amd import keys as K, shard
    B = args.batch if B is None else B
    cfg_name = args.config if cfg_name is None else cfg_name
    hs, ptrs, inputs, member_seed, nd, nnz = batch_setup(c, B, dev, local_rank, be)
    runs = []
    for rep in range(4):  # rep 0 is the cold step, reported on its own; the median of the other three is the result
        shard.fence(dev)
        t0 = time.perf_counter()
        st = be.factorize_batch(hs, device_ptrs=ptrs, block=args.batch_block)
        shard.fence(dev)
        el = time.perf_counter() - t0
        if any(s != K.OK for s in st):
            raise RuntimeError("batched factorize failed: %r" % (st,))
        runs.append((el, hs[0].stat(K.STAT_DEV_TIME_PIVOT_LOOP), int(hs[0].stat(K.STAT_DEV_RELAUNCHES)),
                     [hs[0].stat(k) for k in (44, 45, 46, 47)]))
    cold = runs[0]
    warm = sorted(runs[1:], key=lambda r: r[0])
    med = warm[len(warm) // 2]
    F = sum(h.stat(K.STAT_FACTOR_FLOPS) for h in hs)
    lu = sum(h.stat(K.STAT_L_NZ) + h.stat(K.STAT_U_NZ) for h in hs)
    which = int(hs[0].stat(118))  # the pivot kernel the library chose (blu_driver.inc: batch_pivot_and_finish)
    regs = int(hs[0].stat(120))   # ... and its register budget in waves per SIMD (3: the _r3 variant of a wave kernel)
    fast_share = (sum(h.stat(110) + h.stat(111) for h in hs[:64])) / max(1.0, sum(h.stat(52) + h.stat(54) for h in hs[:64]))
    # the O(nnz) kernels of the step by the same algorithmic bytes as the single-basis line, summed
    # over the bases of the launch
    m_ = c["m"]
    bump = sum(h.stat(K.STAT_BUMP_NZ) for h in hs)
    onnz_bytes = {"k_prep": 16.0 * (nnz + B * m_) + 16.0 * nnz, "k_setup": 16.0 * nnz + 24.0 * bump + 32.0 * B * (2 * m_ + 2),
                  "k_finish": 16.0 * lu + 16.0 * (lu + 2 * B * m_) + 16.0 * B * m_}
    el = shard.max_over_ranks(med[0], dev)
    t_piv, nl, hs_phase = med[1], med[2], med[3]
    gbs = (32.0 * F + 32.0 * lu) / t_piv / 1e9
    sample = sorted(set([0, B - 1] + [(q * B) // 7 for q in range(1, 7)]))
    verified = _verify_members(hs, member_seed, c, be, sample) if not args.no_verify else 0
    for h in hs:
        h.close()
    del inputs, ptrs
    torch.cuda.empty_cache()  # (the next leg needs the card: torch would keep the freed inputs cached)
    traffic = None
    kname = {0: "k_pivot_loop", 1: "k_pivot_loop_wave", 2: "k_pivot_loop_batch", 3: "k_pivot_loop_wave2"}[which]
    wg_threads = {0: args.batch_block, 1: 64, 2: args.batch_block, 3: 128}[which]
    tinfo = _traffic_record("%s@%s" % (kname, cfg_name)) or _traffic_record(kname)
    if tinfo and tinfo.get("bases") == B and tinfo.get("config") == cfg_name:
        traffic = tinfo["hbm_bytes_per_launch"] / max(t_piv / max(nl, 1), 1e-12) / 1e9
    return {"config": "%s-size bases (m = %d; lp_basis k=%d bw=%d tri_frac=%g offscale=%g, seeds 1000..%d)" % (
                cfg_name, c["m"], c["k"], c["bw"], c["tri_frac"], c["offscale"], 999 + nd),
            "bases_in_flight_per_gpu": B, "distinct_matrices": nd, "inputs": "one device copy of B per handle",
            "workgroup_threads": wg_threads, "nnz_per_s": world * nnz / el,
            "seconds": el, "seconds_warm_steps": [r[0] for r in runs[1:]], "timing": "median of 3 warm steps",
            "cold_first_step": {"seconds": cold[0], "pivot_kernel_seconds": cold[1], "pivot_kernel_launches": cold[2],
                                "note": "handles created with hint nnz/2: storage growth, compaction rounds and relaunches included"},
            "pivot_kernel_seconds": t_piv, "pivot_kernel_launches": nl,
            "phases_seconds": {"k_prep": hs_phase[0], "k_setup": hs_phase[1], "k_finish": hs_phase[2], "k_stats": hs_phase[3]},
            "roofline_onnz_kernels": {k: {"bound": "hbm", "achieved": onnz_bytes[k] / max(hs_phase[i], 1e-12) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                          "frac": onnz_bytes[k] / max(hs_phase[i], 1e-12) / 1e9 / HBM_PEAK_GBS,
                                          "algorithmic_bytes_per_launch": onnz_bytes[k], "seconds": hs_phase[i]}
                                      for i, k in enumerate(("k_prep", "k_setup", "k_finish"))},
            "flattened_path_share": fast_share, "verified_members": verified,
            "verified_against": "CPU oracle, untimed, after the timed steps: members %s -- integer arrays, values, counters, "
                                "pivots per routine, statistics bit-identical" % (sample,),
            "roofline": {"bound": "hbm", "kernel": "%s%s (grid = %d workgroups)" % (kname, "_r3" if regs == 3 and which in (1, 3) else "", B), "achieved": gbs,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": traffic},
            "note": "throughput mode, reported beside the headline; `value` above is ONE basis per GPU"}


def _traffic_record(kernel):
    """PMC traffic of `kernel` from profiles/pivot_loop_traffic.json -- only if it was measured on the kernel
    sources this library is built from (PMC can not be collected from inside this script)."""
    try:
        rec = json.load(open(TRAFFIC_FILE))
    except (OSError, ValueError):
        return None
    if rec.get("kernel_source_sha16") != kernel_source_sha16():
        return None  # stale: the kernels changed since the counters were collected
    return rec.get("kernels", {}).get(kernel)


def bench_c5(args):
    """BASELINE.json configs[4]: Forrest-Tomlin update + sparse re-solve loop on the 100k basis.  A step = `--mods`
    column modifications, each: solve_for_update (transposed, solution wanted = the re-solve of the row system),
    solve_for_update (forward, solution wanted = the re-solve with the incoming column), update.  The factorize
    in front of every step is outside the timed region.  Secondary metric (not the headline): modifications/s,
    and the algorithmic byte rate of the solves, 16*(l_flops + u_flops + r_flops) + 16*nz(lhs)."""
    import numpy as np
    import blu_amd
    from blu_amd import keys as K
    from blu_amd.matrices import CONFIGS
    from blu_amd.workloads import column_modifications
    c = CONFIGS["C3"]
    cp, ri, v = blu_amd.gen_lp_basis(c["m"], c["k"], c["bw"], c["tri_frac"], c["seed"], c["offscale"])
    m = c["m"]
    h = blu_amd.BLU(m, len(ri))
    mods = list(column_modifications(cp, ri, args.mods, c["offscale"]))
    tot_t, done, skipped, nzl, flops = 0.0, 0, 0, 0, 0.0
    for step in range(args.warmup + args.steps):
        assert h.factorize(cp[:-1], cp[1:], ri, v) == K.OK
        f0 = sum(h.stat(k) for k in (K.STAT_L_FLOPS, K.STAT_U_FLOPS, K.STAT_R_FLOPS))
        t0 = time.perf_counter()
        d = s_ = nz = 0
        for j, rows, vals in mods:
            assert h.solve_for_update([j], None, "T") == K.OK
            nz += h.nzlhs
            assert h.solve_for_update(rows, vals, "N") == K.OK
            nz += h.nzlhs
            xtbl = h.lhs[j]
            if abs(xtbl) < 1e-3:  # the replacement would make the basis (nearly) singular: not applied
                s_ += 1
                continue
            st = h.update(xtbl)
            if st == K.ERROR_SINGULAR_UPDATE:
                s_ += 1
                continue
            assert st == K.OK, st
            d += 1
        el = time.perf_counter() - t0
        if step >= args.warmup:
            tot_t += el
            done += d
            skipped += s_
            nzl += nz
            flops += sum(h.stat(k) for k in (K.STAT_L_FLOPS, K.STAT_U_FLOPS, K.STAT_R_FLOPS)) - f0
    nmod = args.steps * len(mods)
    gbs = (16.0 * flops + 16.0 * nzl) / tot_t / 1e9
    out = {"metric": "Forrest-Tomlin update + sparse re-solve: column modifications/s, 100k x 100k basis", "value": nmod / tot_t,
           "unit": "modifications/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * tot_t / args.steps,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "C5: C3 basis (m=%d, nnz=%d) + %d column modifications per step (blu_amd/workloads.py, SplitMix64 seed 99): "
                                  "2 x solve_for_update with solution + update each" % (m, len(ri), len(mods)),
                      "updates_applied_per_step": done / args.steps, "not_applied_per_step": skipped / args.steps,
                      "nforrest_end": h.stat(K.STAT_NFORREST), "pivot_error_last": h.stat(K.STAT_PIVOT_ERROR)},
           "roofline": {"bound": "hbm", "kernel": "k_solve_upd + k_update (one wave each: dependent pointer chases)", "achieved": gbs,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                        "algorithmic_bytes": "16*(l_flops+u_flops+r_flops) + 16*nz(lhs)"}}
    print(json.dumps(out), flush=True)
    return out


def rank_main(args, backend=None, device=None):
    """What one rank does.  backend: the module that provides BLU / gen_lp_basis / factorize_batch (blu_amd on
    the GPU; tests/test_shard_gloo.py passes a CPU stand-in to execute the N > 1 code path -- seeds, barriers,
    MAX over ranks, the JSON line -- under gloo)."""
    import numpy as np
    import torch
    from blu_amd import keys as K, shard
    from blu_amd.matrices import CONFIGS

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (start one process per GPU, or run without a launcher)" % (args.gpus, world))
    stub = backend is not None
    if not stub:
        import blu_amd as backend
    # under a launcher (RANK and MASTER_ADDR set) the process group is initialised whatever the world size, so that the
    # RCCL barrier / MAX / SUM path of shard.py runs at N = 1 too (tests/test_gpu_bench.py does exactly that)
    dist = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ and not stub)
    import torch.distributed as td
    if stub:
        dev = device if device is not None else torch.device("cpu")
        if dist and not td.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            td.init_process_group("gloo")
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        if dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            td.init_process_group("nccl", device_id=dev)
            if rank == 0:
                print("process group: %s, world size %d" % (td.get_backend(), td.get_world_size()), file=sys.stderr, flush=True)

    c = dict(CONFIGS[args.config])
    n_bases = 8 if args.config == "C4" else world
    mine = shard.bases_of_rank(max(n_bases, world), rank, world)
    c["seed"] = shard.seed_of_basis(c, mine[0])  # independent bases, one per GPU (C4 at N < 8: the first of this rank's share)
    cp, ri, v = backend.gen_lp_basis(c["m"], c["k"], c["bw"], c["tri_frac"], c["seed"], c["offscale"])
    m, nnz = c["m"], len(ri)

    # inputs resident in HBM before the clock starts (uint64 bit patterns carried in int64 tensors)
    d_cp = torch.from_numpy(cp.view(np.int64)).to(dev)
    d_ri = torch.from_numpy(ri.view(np.int64)).to(dev)
    d_v = torch.from_numpy(v).to(dev)
    p_begin, p_end = d_cp.data_ptr(), d_cp.data_ptr() + 8
    h = backend.BLU(m, nnz, device=local_rank)
    if args.block:
        h.dbg_set_block(args.block)

    def step():
        st = h.factorize_device(p_begin, p_end, d_ri.data_ptr(), d_v.data_ptr(), nnz)
        if st != K.OK:
            raise RuntimeError("factorize status %d" % st)

    def fence():
        shard.fence(dev)  # barrier (RCCL / gloo) + torch.cuda.synchronize()

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
    total_nnz = shard.sum_over_ranks(nnz, dev)

    F = h.stat(K.STAT_FACTOR_FLOPS)
    l_nz, u_nz = h.stat(K.STAT_L_NZ), h.stat(K.STAT_U_NZ)
    # algorithmic bytes: every multiply-add reads and rewrites one 16-byte (index,value)
    # entry; every L/U off-diagonal is read from the active submatrix and written to its factor once
    bytes_elim = 32.0 * F + 32.0 * (l_nz + u_nz)
    bytes_all = 16.0 * (nnz + m) + bytes_elim
    t_kernel = t_pivot / max(1, nlaunch)  # average k_pivot_loop launch
    traffic = None
    tinfo = _traffic_record("k_pivot_loop")
    if tinfo and tinfo.get("config") == args.config:
        traffic = tinfo["hbm_bytes_per_launch"] / max(t_kernel, 1e-12) / 1e9
    achieved = bytes_elim * args.steps / max(t_pivot, 1e-12) / 1e9
    phases = {}
    # k_stats = the whole statistics tail (single matrix: k_rows_grid + k_stats_chains + k_stats_tail, of which the
    # first and the last are also listed on their own)
    for name, key in (("k_prep", 44), ("k_setup", 45), ("k_finish", 46), ("k_stats", 47), ("k_rows_grid", 108), ("k_stats_tail", 109)):
        val = h.stat(key)
        if val == val and val > 0:
            phases[name + "_ms"] = 1e3 * val
    # the O(nnz) phases: algorithmic bytes with the reference's 16-byte (index, value) entries and 8-byte integers
    # (SURVEY.md 8d): k_prep reads B and writes the row-wise copy; k_setup reads it and writes both files of the
    # bump plus the count lists; k_finish reads the stage-ordered factors and writes the canonical ones
    bump_nz = h.stat(K.STAT_BUMP_NZ)
    o_bytes = {"k_prep": 16.0 * (nnz + m) + 16.0 * nnz, "k_setup": 16.0 * nnz + 24.0 * bump_nz + 32.0 * (2 * m + 2),
               "k_finish": 16.0 * (l_nz + u_nz) + 16.0 * (l_nz + u_nz + 2 * m) + 16.0 * m}
    o_rows = {}
    for name in o_bytes:
        if name + "_ms" in phases:
            g = o_bytes[name] / (1e-3 * phases[name + "_ms"]) / 1e9
            o_rows[name] = {"bound": "hbm", "achieved": g, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": g / HBM_PEAK_GBS,
                            "algorithmic_bytes_per_launch": o_bytes[name], "ms": phases[name + "_ms"]}
    out = {
        "metric": "factorize nnz/s + achieved HBM GB/s, %dk x %dk %d-nnz/col basis" % (m // 1000, m // 1000, c["k"]),
        "value": total_nnz * args.steps / elapsed,
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
        "config": {"workload": "%s: single %dx%d synthetic LP basis per GPU (lp_basis k=%d bw=%d tri_frac=%g offscale=%g seed=%d+basis), "
                               "nnz=%d, inputs resident in HBM" % (args.config, m, m, c["k"], c["bw"], c["tri_frac"], c["offscale"],
                                                                  CONFIGS[args.config]["seed"], nnz),
                   "m": m, "nnz": nnz, "l_nz": l_nz, "u_nz": u_nz, "factor_flops": F,
                   "rank": h.stat(K.STAT_RANK), "bump_size": h.stat(K.STAT_BUMP_SIZE),
                   "nsearch_pivot": h.stat(K.STAT_NSEARCH_PIVOT), "parallelism": "one basis per GPU, no data-path collective",
                   "generator_note": "bw = %d, not SURVEY.md 8d's provisional 16: at bw = 16 the reference's i32 cancellation mask (defect D3, "
                                     "pivot.rs:645-659) is hit and the reference has no defined result; bw = %d keeps d3_hits == 0 (fill %.1fx "
                                     "nnz(B)); BASELINE.md section 8" % (c["bw"], c["bw"], (l_nz + u_nz) / max(1, nnz))},
        "roofline": {"bound": "hbm", "kernel": "k_pivot_loop", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_source": ("profiles/pivot_loop_traffic.json (rocprofv3 FETCH_SIZE + WRITE_SIZE per launch / avg launch time, GB/s)"
                                        if traffic is not None else "none valid for these kernel sources (profiles/pivot_loop_traffic.json is "
                                        "keyed by the hash of blu_amd/csrc; re-collect with tools/pmc_traffic.sh)"),
                     "algorithmic_bytes_per_launch": bytes_elim * args.steps / max(1, nlaunch),
                     "avg_launch_ms": 1e3 * t_kernel, "launches_per_step": nlaunch / args.steps},
        "achieved_GBs_whole_factorize": bytes_all * args.steps / elapsed / 1e9,
        "device_ms_per_step": 1e3 * t_dev / args.steps,
        "phases_last_step": phases,
        "roofline_onnz_kernels": o_rows,
    }
    if args.batch > 0 and not stub:
        out["batched"] = batched_throughput(args, c, dev, local_rank, world, backend)
        if args.config == "C3" and not args.no_batch_sizes:
            # the same measurement on smaller bases, of which more fit into HBM (the batch kernel is limited by the number
            # of bases in flight): C4-size (50k) and C2-size (10k)
            out["batched_other_sizes"] = [
                batched_throughput(args, dict(CONFIGS["C4"]), dev, local_rank, world, backend, B=3072, cfg_name="C4"),
                batched_throughput(args, dict(CONFIGS["C2"]), dev, local_rank, world, backend, B=4096, cfg_name="C2")]
    if rank == 0 and not args.no_cpu_baseline:
        c0 = CONFIGS[args.config]
        cp0, ri0, v0 = (cp, ri, v) if c["seed"] == c0["seed"] else backend.gen_lp_basis(c0["m"], c0["k"], c0["bw"], c0["tri_frac"], c0["seed"], c0["offscale"])
        out["cpu_baseline"] = cpu_baseline(cp0, ri0, v0, args.config)
        out["vs_cpu_baseline_per_gpu"] = (out["value"] / world) / out["cpu_baseline"]["value"]
    if dist:
        td.barrier()
        if not stub:
            td.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)
    return out if rank == 0 else None


def main():
    args = parse_args()
    if args.config == "C5":
        if args.gpus != 1:
            raise SystemExit("bench.py: --config C5 is a single-GPU workload")
        bench_c5(args)
        return
    if args.gpus > 1 and "RANK" not in os.environ:
        # No launcher: start one as a child process, before this process has touched the GPU in any way
        # (nothing of torch.cuda or libblu_hip has been imported or called so far), and hand its return code on.
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29511"),
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    rank_main(args)


if __name__ == "__main__":
    main()
