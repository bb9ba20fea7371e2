"""Device memory around two batches in one process (diagnostic): python tools/mem_probe.py"""
import sys, os, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import blu_amd
from blu_amd import keys as K
from blu_amd.matrices import CONFIGS
dev = torch.device("cuda", 0)
def free(): return torch.cuda.mem_get_info()[0] / 1e9
def run(cfg, B):
    c = CONFIGS[cfg]; nd = 8; mats = []
    for s in range(nd):
        cp, ri, v = blu_amd.gen_lp_basis(c["m"], c["k"], c["bw"], c["tri_frac"], 1000 + s, c["offscale"])
        mats.append((torch.from_numpy(cp.view(np.int64)).to(dev), torch.from_numpy(ri.view(np.int64)).to(dev), torch.from_numpy(v).to(dev), len(ri)))
    print(cfg, B, "before alloc: free %.1f GB" % free(), flush=True)
    hs = [blu_amd.BLU(c["m"], mats[k % nd][3] // 2) for k in range(B)]
    print("  after alloc: free %.1f GB" % free(), flush=True)
    ptrs = [(mats[k % nd][0].data_ptr(), mats[k % nd][0].data_ptr() + 8, mats[k % nd][1].data_ptr(), mats[k % nd][2].data_ptr(), mats[k % nd][3]) for k in range(B)]
    for rep in range(2):
        try:
            st = blu_amd.factorize_batch(hs, device_ptrs=ptrs)
        except Exception as e:
            print("  rep", rep, "FAILED", e, "free %.1f GB" % free(), flush=True); break
        torch.cuda.synchronize()
        print("  rep %d: free %.1f GB, relaunches %d, pivot %.3f s" % (rep, free(), hs[0].stat(K.STAT_DEV_RELAUNCHES), hs[0].stat(K.STAT_DEV_TIME_PIVOT_LOOP)), flush=True)
    for h in hs: h.close()
    del hs; gc.collect()
    print("  after close: free %.1f GB" % free(), flush=True)
for a in sys.argv[1:]:
    cfg, B = a.split(":"); run(cfg, int(B))
