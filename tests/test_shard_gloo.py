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


# ---- the real bench.py rank function under gloo, with a CPU stand-in for the device library --------------------
class _CpuBackend:
    """Stand-in for blu_amd in bench.rank_main: same entry points, the factorize call answered by the CPU oracle
    (test infrastructure) on the host tensors bench.py prepared.  Executes bench.py's N > 1 path on CPU: per-rank
    seeds, barriers, MAX over ranks, the summed throughput and the one JSON line of rank 0."""

    def __init__(self):
        import blu_amd
        from oracle import orc
        self._blu, self._orc = blu_amd, orc
        self.seeds = []

    def gen_lp_basis(self, m, k, bw, tri, seed, offs):
        self.seeds.append(int(seed))
        return self._blu.gen_lp_basis(m, k, bw, tri, seed, offs)

    def BLU(self, m, nnz, device=0):
        import ctypes as C
        import time
        orc, K = self._orc, __import__("blu_amd").keys
        outer = self

        class H:
            def __init__(self):
                self.o = orc.OracleBLU(m, 16 * nnz)
                self.t = 0.0

            def factorize_device(self, pb, pe, pi, px, n):
                u64 = lambda p, cnt: np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint64)), shape=(cnt,))
                bb, be = u64(pb, m), u64(pe, m)
                bi = u64(pi, n)
                bx = np.ctypeslib.as_array(C.cast(px, C.POINTER(C.c_double)), shape=(n,))
                t0 = time.perf_counter()
                st = self.o.factorize(bb, be, bi, bx)
                self.t = time.perf_counter() - t0
                return st

            def stat(self, key):
                if key in (K.STAT_DEV_TIME_PIVOT_LOOP, K.STAT_DEV_TIME_TOTAL):
                    return self.t
                if key == K.STAT_DEV_RELAUNCHES:
                    return 1.0
                if key in (44, 45, 46, 47):
                    return 0.0
                return self.o.stat(key)

            def dbg_set_block(self, n):
                pass
        return H()


def _bench_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    from blu_amd import matrices
    matrices.CONFIGS["C4"] = dict(matrices.CONFIGS["C4"], m=600)  # small stand-in for the 50k bases
    args = bench.parse_args(["--gpus", str(world), "--steps", "2", "--warmup", "1", "--config", "C4", "--batch", "0", "--no-cpu-baseline"])
    be = _CpuBackend()
    res = bench.rank_main(args, backend=be, device=torch.device("cpu"))
    out.put((rank, be.seeds, res))
    td.destroy_process_group()


def test_bench_rank_function_two_ranks_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(world):
        r, seeds, res = q.get(timeout=180)
        got[r] = (seeds, res)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][0] == [1] and got[1][0] == [2]  # rank r factorizes basis r: seed 1 + r, different matrices
    line = got[0][1]
    assert got[1][1] is None  # only rank 0 reports
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["unit"] == "nnz/s" and line["vs_baseline"] is None and line["dtype"] == "f64"
    # whole-job value: nnz of BOTH ranks' bases x steps / max-over-ranks elapsed
    per_step = line["ms_per_step"] * 1e-3
    assert line["value"] > line["config"]["nnz"] / per_step * 1.5  # two bases, not one
    assert line["roofline"]["traffic"] is None or line["roofline"]["traffic"] > 0


def test_bench_rank_function_eight_ranks_gloo_config_c4():
    """The 8-rank dealing of C4 (shard.bases_of_rank(8, r, 8): one 50k basis per rank, seeds 1..8) through bench.rank_main, once,
    before an 8-GPU node ever sees it: eight gloo ranks on the CPU (small stand-in bases), every rank its own seed, the
    whole-job value = nnz of all eight bases x steps / max-over-ranks time, one JSON line from rank 0."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(world):
        r, seeds, res = q.get(timeout=300)
        got[r] = (seeds, res)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [got[r][0] for r in range(world)] == [[1 + r] for r in range(world)]  # basis b -> rank b: seeds 1..8, no two alike
    assert all(got[r][1] is None for r in range(1, world))
    line = got[0][1]
    assert line["n_gpus"] == 8 and line["scaling"] == "weak" and line["config"]["workload"].startswith("C4")
    per_step = line["ms_per_step"] * 1e-3
    assert line["value"] > line["config"]["nnz"] / per_step * 6.0  # eight bases' nnz over the slowest rank's time


def test_bench_refuses_world_size_mismatch():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "1"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    assert r.returncode != 0 and b"WORLD_SIZE" in r.stdout
