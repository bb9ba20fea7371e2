"""Statistics tail on the chain pipeline: python tools/chain_probe.py C2|C3|m,k,bw,tri,offs,seed  (BLU_HIP_NO_CHAIN=1 for the one-workgroup kernel)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blu_amd
from blu_amd import keys as K
from blu_amd.matrices import CONFIGS
a = sys.argv[1] if len(sys.argv) > 1 else "C2"
if a in CONFIGS:
    c = CONFIGS[a]
else:
    v = a.split(",")
    c = dict(m=int(v[0]), k=int(v[1]), bw=int(v[2]), tri_frac=float(v[3]), offscale=float(v[4]), seed=int(v[5]))
cp, ri, v = blu_amd.gen_lp_basis(c["m"], c["k"], c["bw"], c["tri_frac"], c["seed"], c["offscale"])
h = blu_amd.BLU(c["m"], len(ri))
for rep in range(2):
    t0 = time.time()
    st = h.factorize(cp[:-1], cp[1:], ri, v)
    print("rep", rep, "status", st, "wall %.3f s" % (time.time() - t0), "stats %.2f ms (rows %.2f, tail %.2f)" % (1e3 * h.stat(47), 1e3 * h.stat(108), 1e3 * h.stat(109)), "err_line", h.stat(57), flush=True)
for key in (K.STAT_CONDEST_L, K.STAT_CONDEST_U, K.STAT_NORMEST_L_INV, K.STAT_NORMEST_U_INV, K.STAT_RESIDUAL_TEST, K.STAT_NORM_L, K.STAT_NORM_U, K.STAT_ONENORM, K.STAT_INFNORM):
    print(key, repr(h.stat(key)))
