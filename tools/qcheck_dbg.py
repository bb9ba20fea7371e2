"""Self-check build of the pivot search's candidate queue (make -C blu_amd/csrc qcheck): factorize one
   generated matrix and, if the check trips, print what the queue said and what the lists say.
   BLU_HIP_LIB=blu_amd/libblu_hip_qcheck.so python tools/qcheck_dbg.py m,k,bw,tri,offscale,seed"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blu_amd
m, k, bw, tri, off, seed = sys.argv[1].split(",")
cp, ri, v = blu_amd.gen_lp_basis(int(m), int(k), int(bw), float(tri), int(seed), float(off))
h = blu_amd.BLU(int(m), len(ri))
try:
    st = h.factorize(cp[:-1], cp[1:], ri, v)
    print("ok status", st, "queue hits", h.stat(48), "walks", h.stat(49))
except blu_amd.BluError as e:
    print("failed:", e)
    p = [int(h.stat(60 + i)) for i in range(16)]
    print("rank", p[0] - 1, "r/r2", p[1], "qN", p[2], "qMinNew", p[3], "cont", p[4], "contNz", p[5], "min_colnz", p[6])
    print("lists say (col, count):", [(x // 1000, x % 1000) for x in p[7:10]])
    print("queue said (col, count):", [(x // 1000, x % 1000) for x in p[10:13]])
    sys.exit(1)
