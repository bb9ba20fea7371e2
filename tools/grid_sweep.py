"""Sweep of the workgroup count of the chip-wide O(nnz) phases (k_prep_grid / k_setup_grid / k_finish_grid) on C3."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blu_amd
from blu_amd import keys as K
from blu_amd.matrices import CONFIGS
c = CONFIGS["C3"]
cp, ri, v = blu_amd.gen_lp_basis(c["m"], c["k"], c["bw"], c["tri_frac"], c["seed"], c["offscale"])
h = blu_amd.BLU(c["m"], len(ri))
h.set_skip_stats(True)
for g in [int(a) for a in sys.argv[1:]] or [1, 16, 32, 64, 128, 256]:
    h.dbg_set_grid_blocks(g)
    best = None
    for rep in range(3):
        assert h.factorize(cp[:-1], cp[1:], ri, v) == K.OK
        t = [1e3 * h.stat(k) for k in (44, 45, 46)]
        best = t if best is None else [min(a, b) for a, b in zip(best, t)]
    print("grid %3d: k_prep %.3f ms  k_setup %.3f ms  k_finish %.3f ms" % (g, *best), flush=True)
