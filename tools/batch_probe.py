"""Throughput of blu_hip_factorize_batch on one GPU, the batch built exactly as bench.py builds it (64 distinct matrices, every
handle with device inputs of its own: bench.batch_setup):  python tools/batch_probe.py B block [config] [hintdiv] [reps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import blu_amd
import bench
from blu_amd import keys as K
from blu_amd.matrices import CONFIGS
B = int(sys.argv[1]); block = int(sys.argv[2]); cfg = sys.argv[3] if len(sys.argv) > 3 else "C3"
hintdiv = int(sys.argv[4]) if len(sys.argv) > 4 else 2
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
c = CONFIGS[cfg]
dev = torch.device("cuda", 0)
t0 = time.time()
hs, ptrs, inputs, member_seed, nd, nnz = bench.batch_setup(c, B, dev, 0, blu_amd, hintdiv=hintdiv)
print("alloc %.1fs, free mem %.1f GB" % (time.time() - t0, torch.cuda.mem_get_info()[0] / 1e9), flush=True)
for rep in range(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st = blu_amd.factorize_batch(hs, device_ptrs=ptrs, block=block)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    assert all(s == 0 for s in st), st
    F = sum(h.stat(K.STAT_FACTOR_FLOPS) for h in hs); lu = sum(h.stat(K.STAT_L_NZ) + h.stat(K.STAT_U_NZ) for h in hs)
    tp = hs[0].stat(K.STAT_DEV_TIME_PIVOT_LOOP)
    ph = [hs[0].stat(k) for k in (44, 45, 46, 47)]
    print("B=%d block=%d kernel=%s rep%d: %.3f s  %.1f Mnnz/s  pivot-kernel %.3f s (%d launches) -> %.1f GB/s alg (%.2f%% of 8 TB/s); "
          "prep %.3f setup %.3f finish %.3f stats %.3f s; fast small/scol %d/%d of %d/%d" %
          (B, block, os.environ.get("BLU_PIVOT_KERNEL", "default"), rep, el, nnz / el / 1e6, tp, hs[0].stat(K.STAT_DEV_RELAUNCHES),
           (32 * F + 32 * lu) / tp / 1e9, (32 * F + 32 * lu) / tp / 1e9 / 80.0, ph[0], ph[1], ph[2], ph[3],
           hs[0].stat(110), hs[0].stat(111), hs[0].stat(54), hs[0].stat(52)), "handed", hs[0].stat(116),
          "free mem %.1f GB" % (torch.cuda.mem_get_info()[0] / 1e9),
          "arena used c/r %d/%d of cap %d; L cap %d" % (max(h.stat(112) for h in hs), max(h.stat(113) for h in hs), hs[0].stat(114), hs[0].stat(115)), flush=True)
if "fprof" in os.environ.get("BLU_HIP_LIB", ""):  # (make -C blu_amd/csrc fprof: shader-clock ticks per phase of k_prep / k_finish, summed over the batch)
    names = ["prep: column pointers", "prep: pack + row counts", "prep: row pointers", "prep: fill, plan + phase A", "prep: fill, phase B / window sweeps",
             "prep: rows of 33..48, duplicates", "prep: long rows", "prep: singletons", "finish: permutations", "finish: L columns",
             "finish: L medium / long columns", "finish: U column counts", "finish: U column pointers", "finish: U fill, plan + phase A",
             "finish: U fill, phase B / window sweeps", "finish: U pivots + short columns", "finish: U medium / long columns",
             "setup: columns counted", "setup: columns copied", "setup: rows counted", "setup: rows copied", "setup: lists and marks initialised",
             "setup: count lists built"]
    tot = [sum(h.stat(60 + k) for h in hs) for k in range(len(names))]
    for lo, hi, kern in ((0, 8, 0), (17, 23, 1), (8, 17, 2)):
        s = sum(tot[lo:hi])
        for k in range(lo, hi):
            print("  %-42s %5.1f %%  ~%.4f s of the kernel's %.3f s" % (names[k], 100 * tot[k] / s, ph[kern] * tot[k] / s, ph[kern]))
