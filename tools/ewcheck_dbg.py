"""Self-check build of the early search (make -C blu_amd/csrc ewcheck): factorize one generated matrix;
   if the check trips, print what the early search said and what the ordinary search says.
   BLU_HIP_LIB=blu_amd/libblu_hip_ewcheck.so python tools/ewcheck_dbg.py m,k,bw,tri,offscale,seed [block]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blu_amd
m, k, bw, tri, off, seed = sys.argv[1].split(",")
cp, ri, v = blu_amd.gen_lp_basis(int(m), int(k), int(bw), float(tri), int(seed), float(off))
h = blu_amd.BLU(int(m), len(ri))
if len(sys.argv) > 2:
    h.dbg_set_block(int(sys.argv[2]))
try:
    st = h.factorize(cp[:-1], cp[1:], ri, v)
    print("ok status", st, "pivot loop %.1f ms" % (1e3 * h.stat(40)))
except blu_amd.BluError as e:
    print("failed:", e)
    p = [int(h.stat(60 + i)) for i in range(16)]
    print("rank", p[0] - 1, "ncand early/ordinary", p[1], "nsearched early/ordinary", p[2])
    print("ordinary (col, count):", [(x // 1000, x % 1000) for x in p[7:10]], "early cols:", p[10:13])
    sys.exit(1)
