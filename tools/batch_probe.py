"""Throughput of blu_hip_factorize_batch on one GPU: python tools/batch_probe.py B block [config] [hintdiv]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import blu_amd
from blu_amd import keys as K
from blu_amd.matrices import CONFIGS
B = int(sys.argv[1]); block = int(sys.argv[2]); cfg = sys.argv[3] if len(sys.argv) > 3 else "C3"
hintdiv = int(sys.argv[4]) if len(sys.argv) > 4 else 2
c = CONFIGS[cfg]
dev = torch.device("cuda", 0)
nd = min(B, 8)
mats = []
for s in range(nd):
    cp, ri, v = blu_amd.gen_lp_basis(c["m"], c["k"], c["bw"], c["tri_frac"], c["seed"] + s, c["offscale"])
    mats.append((torch.from_numpy(cp.view(np.int64)).to(dev), torch.from_numpy(ri.view(np.int64)).to(dev), torch.from_numpy(v).to(dev), len(ri)))
t0 = time.time()
hs = [blu_amd.BLU(c["m"], mats[k % nd][3] // hintdiv) for k in range(B)]
print("alloc %.1fs, free mem %.1f GB" % (time.time() - t0, torch.cuda.mem_get_info()[0] / 1e9), flush=True)
ptrs = [(mats[k % nd][0].data_ptr(), mats[k % nd][0].data_ptr() + 8, mats[k % nd][1].data_ptr(), mats[k % nd][2].data_ptr(), mats[k % nd][3]) for k in range(B)]
nnz = sum(p[4] for p in ptrs)
for rep in range(3):
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
           hs[0].stat(110), hs[0].stat(111), hs[0].stat(54), hs[0].stat(52)), "handed", hs[0].stat(116), flush=True)
