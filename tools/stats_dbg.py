"""Debug: work vectors of the statistics chains, chain pipeline vs one-workgroup kernel (library built with -DBLU_STATS_DEBUG).
   BLU_HIP_LIB=.../libblu_hip_sdbg.so python tools/stats_dbg.py --seed 777 --case 36"""
import sys, os, argparse, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from blu_amd import keys as K
from oracle import orc
import tools.fuzz_gpu as fz
ap = argparse.ArgumentParser(); ap.add_argument("--seed", type=int, default=777); ap.add_argument("--case", type=int, default=36)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
for case in range(a.case + 1):
    c, mat = fz.draw(rng)
    o, so = fz.oracle_of(c, mat)
    if so in (K.OK, K.WARNING_SINGULAR_MATRIX):
        fz.draw_solves(rng, c["m"])
print(fz.tag_of(a.case, c), "oracle status", so, "rank", o.stat(K.STAT_RANK))
import blu_amd
from blu_amd.blu import lib
cp, ri, v = mat
m = c["m"]
vecs = {}
for mode in ("chain", "one"):
    if mode == "one":
        os.environ["BLU_HIP_NO_CHAIN"] = "1"
    g = blu_amd.BLU(m, c["hint"])
    for key, val in c["params"].items():
        g.set_param(key, val)
    g.dbg_set_block(c["block"])
    st = g.factorize(cp[:-1], cp[1:], ri, v)
    buf = np.zeros(9 * (m + 1))
    lib().blu_hip_dbg_get_gwork.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    assert lib().blu_hip_dbg_get_gwork(g._h, buf.ctypes.data, len(buf)) == 0
    vecs[mode] = buf.reshape(9, m + 1)
    print(mode, "status", st, "rows ms", 1e3 * g.stat(108))
f = o.get_factors()
names = ["wl", "wu", "lf", "rf", "lb", "rb"]
for q, nm in enumerate(names):
    x, y = vecs["chain"][q][:m], vecs["one"][q][:m]
    if nm == "wl":
        # the one-workgroup kernel keeps wl in row-index coordinates
        y = y[f["rowperm"].astype(np.int64)]
    bad = np.flatnonzero(x != y)
    print(nm, "differences:", len(bad), "first", bad[:8], [(float(x[i]), float(y[i])) for i in bad[:3]])
    if len(bad):
        k = int(bad[0] if nm not in ("lf", "wu") else bad[-1] if nm in ("wu",) else bad[0])
        ucp, uri = f["u_colptr"].astype(np.int64), f["u_rowidx"].astype(np.int64)
        lcp, lri = f["l_colptr"].astype(np.int64), f["l_rowidx"].astype(np.int64)
        print("   position", k, "rank", o.stat(K.STAT_RANK), "U col len", ucp[k + 1] - ucp[k] - 1, "L col len", lcp[k + 1] - lcp[k] - 1,
              "U row len", int((uri == k).sum()) - 1, "L row len", int((lri == k).sum()) - 1)

# ---- the row-wise copies of the chain handle: sortedness and content against the canonical factors
g = None
os.environ.pop("BLU_HIP_NO_CHAIN", None)
g = blu_amd.BLU(m, c["hint"])
for key, val in c["params"].items():
    g.set_param(key, val)
g.dbg_set_block(c["block"])
g.factorize(cp[:-1], cp[1:], ri, v)
lnz, unz = int(g.stat(K.STAT_L_NZ)), int(g.stat(K.STAT_U_NZ)) + m
ltp = np.zeros(m + 1, np.int32); lti = np.zeros(max(lnz, 1) + 8, np.int32); ltv = np.zeros(max(lnz, 1) + 8)
url = np.zeros(m, np.int32); urp = np.zeros(4 * unz + 8, np.int32); urv = np.zeros(4 * unz + 8); ub = np.zeros(m + 1, np.int32)
fn = lib().blu_hip_dbg_get_rows
fn.argtypes = [C.c_void_p] + [C.c_void_p] * 7
# (uused may exceed u_nz: entries in non-pivotal columns) -- buffers are generous
assert fn(g._h, ltp.ctypes.data, lti.ctypes.data, ltv.ctypes.data, url.ctypes.data, urp.ctypes.data, urv.ctypes.data, ub.ctypes.data) == 0
import scipy.sparse as sp
Lc = sp.csc_matrix((f["l_value"], f["l_rowidx"].astype(np.int64), f["l_colptr"].astype(np.int64)), shape=(m, m)).tocsr()
Uc = sp.csc_matrix((f["u_value"], f["u_rowidx"].astype(np.int64), f["u_colptr"].astype(np.int64)), shape=(m, m)).tocsr()
rowperm = f["rowperm"].astype(np.int64); pinv = np.empty(m, np.int64); pinv[rowperm] = np.arange(m)
badl = badu = 0
for k in range(m):
    i = rowperm[k]
    idx = lti[ltp[i]:ltp[i + 1]]; pos = pinv[idx]
    want = Lc.indices[Lc.indptr[k]:Lc.indptr[k + 1]]; wv = Lc.data[Lc.indptr[k]:Lc.indptr[k + 1]]
    sel = want != k
    o_ = np.argsort(want[sel])
    if not (np.array_equal(pos, want[sel][o_]) and np.array_equal(ltv[ltp[i]:ltp[i + 1]], wv[sel][o_])):
        badl += 1
        if badl < 3: print("L row of position", k, "got", pos, "want", want[sel][o_])
    pu = urp[ub[k]:ub[k] + url[k]]
    want = Uc.indices[Uc.indptr[k]:Uc.indptr[k + 1]]; wv = Uc.data[Uc.indptr[k]:Uc.indptr[k + 1]]
    sel = want != k
    o_ = np.argsort(-want[sel])
    if not (np.array_equal(pu, want[sel][o_]) and np.array_equal(urv[ub[k]:ub[k] + url[k]], wv[sel][o_])):
        badu += 1
        if badu < 3: print("U row of position", k, "got", pu, "want", want[sel][o_])
print("row-wise L rows wrong:", badl, " U rows wrong:", badu)
