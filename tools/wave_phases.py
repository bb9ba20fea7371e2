"""Phase table of k_pivot_loop_wave from the diagnostic build (make prof; BLU_HIP_LIB=blu_amd/libblu_hip_prof.so):
   python tools/wave_phases.py [B] [config]
runs a batch of B bases (default 1280 x C2) and prints, for the first basis, shader-clock ticks and visits per phase."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import blu_amd
from blu_amd import keys as K
from blu_amd.matrices import CONFIGS
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
cfg = sys.argv[2] if len(sys.argv) > 2 else "C2"
c = CONFIGS[cfg]
dev = torch.device("cuda", 0)
nd = min(B, 8)
mats = []
for s in range(nd):
    cp, ri, v = blu_amd.gen_lp_basis(c["m"], c["k"], c["bw"], c["tri_frac"], c["seed"] + s, c["offscale"])
    mats.append((torch.from_numpy(cp.view(np.int64)).to(dev), torch.from_numpy(ri.view(np.int64)).to(dev), torch.from_numpy(v).to(dev), len(ri)))
hs = [blu_amd.BLU(c["m"], mats[k % nd][3] // 2) for k in range(B)]  # (the capacity hint bench.py uses)
ptrs = [(mats[k % nd][0].data_ptr(), mats[k % nd][0].data_ptr() + 8, mats[k % nd][1].data_ptr(), mats[k % nd][2].data_ptr(), mats[k % nd][3]) for k in range(B)]
for rep in range(2):
    st = blu_amd.factorize_batch(hs, device_ptrs=ptrs)
    assert all(s == 0 for s in st), st
names = ["loop head", "search: heads / express / hand-over", "search: walk", "search: stage + reduce", "layout: pivot row+col -> slots",
         "layout: line metadata, sums", "small: row hash, offsets", "small: pass A", "small: column epilogue (+re-append)", "small: pass B",
         "small: column finalize, U", "small: column hash, offsets", "small: rows pass", "small: row epilogue", "small: row append",
         "small: L column", "small: cleanup", "small: list move, walk of the next search begun", "scol: hash, offsets", "scol: pass", "scol: finalize, U",
         "scol: list move, hand-over, cleanup", "record pivot", "general paths"]
if int(hs[0].stat(118)) == 3:  # k_pivot_loop_wave2 ran (a batch all of whose workgroups are resident): the clock of wave 0 (k_pivot_wave2.inc)
    names[2] = "search: walk (begun while wave 1 updates the rows)"; names[6] = "small: publish to LDS, offsets"
    names[10] = "small: column finalize"; names[14] = "small: the walk's first loads issued"
    names[15] = "small: WAIT for wave 1 (the rows)"; names[16] = "small: U row, list move, L column, cleanup"
    names[17] = "small: WAIT for wave 1 (its columns)"
h = hs[0]
tp = h.stat(K.STAT_DEV_TIME_PIVOT_LOOP)
tot = sum(h.stat(60 + k) for k in range(24))
npiv = h.stat(52) + h.stat(54)
print("B=%d %s (%s): pivot kernel %.3f s; basis 0: %.0f ticks in %d pivots (small %d, scol %d; searches handed over %d, with the walk begun early %d)" % (B, cfg, {0: "k_pivot_loop", 1: "k_pivot_loop_wave", 2: "k_pivot_loop_batch", 3: "k_pivot_loop_wave2"}[int(h.stat(118))], tp, tot, npiv, h.stat(54), h.stat(52), h.stat(116), h.stat(117)))
for k, nm in enumerate(names):
    t, n = h.stat(60 + k), h.stat(84 + k)
    if n:
        print("%-42s %6.2f %%  %9.0f ticks/visit  x %7d" % (nm, 100.0 * t / tot, t / n, n))
