"""Per-call times of the update path on the C3 basis (first N modifications of the C5 stream)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import blu_amd
from blu_amd import keys as K
from blu_amd.matrices import CONFIGS
from blu_amd.workloads import column_modifications
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
c = CONFIGS["C3"]
cp, ri, v = blu_amd.gen_lp_basis(c["m"], c["k"], c["bw"], c["tri_frac"], c["seed"], c["offscale"])
h = blu_amd.BLU(c["m"], len(ri))
assert h.factorize(cp[:-1], cp[1:], ri, v) == K.OK
tt = tn = tu = 0.0
nzt = nzn = 0
br = {1: 0, 2: 0}
for j, rows, vals in column_modifications(cp, ri, n, c["offscale"]):
    t0 = time.perf_counter(); assert h.solve_for_update([j], None, "T") == K.OK; tt += time.perf_counter() - t0; nzt += h.nzlhs; br[int(h.stat(43))] += 1
    t0 = time.perf_counter(); assert h.solve_for_update(rows, vals, "N") == K.OK; tn += time.perf_counter() - t0; nzn += h.nzlhs; br[int(h.stat(43))] += 1
    t0 = time.perf_counter(); st = h.update(h.lhs[j]); tu += time.perf_counter() - t0
    assert st == K.OK
print("per modification: solve_for_update T %.2f ms (nz %.0f) | N %.2f ms (nz %.0f) | update %.2f ms | branches sparse/sequential %s" %
      (1e3 * tt / n, nzt / n, 1e3 * tn / n, nzn / n, 1e3 * tu / n, br))
