"""solve_sparse / solve_dense timing on one factorized basis, GPU (blu_hip) next to the CPU oracle.
   python tools/solve_probe.py [C2|C3]      (the oracle is the checker and the timed CPU baseline)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import blu_amd
from blu_amd import keys as K
from blu_amd.matrices import CONFIGS
from oracle import orc
c = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C2"]
m = c["m"]
cp, ri, v = blu_amd.gen_lp_basis(m, c["k"], c["bw"], c["tri_frac"], c["seed"], c["offscale"])
g = blu_amd.BLU(m, len(ri)); o = orc.OracleBLU(m, 16 * len(ri)); o.set_fix_d3(True)
assert g.factorize(cp[:-1], cp[1:], ri, v) == 0 and o.factorize(cp[:-1], cp[1:], ri, v) == 0
rng = np.random.default_rng(1)
print("m=%d nnz=%d l_nz=%d u_nz=%d" % (m, len(ri), g.stat(K.STAT_L_NZ), g.stat(K.STAT_U_NZ)))
for trans in "NT":
    for nz in (1, 10, 100, m // 20):
        ir = rng.choice(m, nz, replace=False); xr = rng.standard_normal(nz)
        g.solve_sparse(ir, xr, trans)  # warm-up (allocations, row-wise L)
        f0 = g.stat(K.STAT_L_FLOPS) + g.stat(K.STAT_U_FLOPS)
        t0 = time.perf_counter(); reps = 5
        for _ in range(reps): g.solve_sparse(ir, xr, trans)
        tg = (time.perf_counter() - t0) / reps
        flops = (g.stat(K.STAT_L_FLOPS) + g.stat(K.STAT_U_FLOPS) - f0) / reps
        t0 = time.perf_counter()
        for _ in range(reps): st, il, lhs = o.solve_sparse(ir, xr, trans)
        tc = (time.perf_counter() - t0) / reps
        same = np.array_equal(il, g.ilhs[:g.nzlhs]) and np.array_equal(lhs, g.lhs)
        byts = 16 * flops + 16 * g.nzlhs
        print("trans=%s nzrhs=%6d -> nzlhs=%6d branch=%d flops=%8d  gpu %8.3f ms (%.4f GB/s alg)  cpu oracle %8.3f ms  identical=%s"
              % (trans, nz, g.nzlhs, g.stat(43), flops, 1e3 * tg, byts / tg / 1e9, 1e3 * tc, same))
b = rng.standard_normal(m)
for trans in "NT":
    g.solve_dense(b, trans); t0 = time.perf_counter(); x = g.solve_dense(b, trans); tg = time.perf_counter() - t0
    t0 = time.perf_counter(); y = o.solve_dense(b, trans); tc = time.perf_counter() - t0
    print("solve_dense trans=%s: gpu %.2f ms, cpu oracle %.2f ms, max rel diff %.1e" % (trans, 1e3 * tg, 1e3 * tc, np.abs(x - y).max() / np.abs(y).max()))
