"""Regenerates tests/golden/*.npz from the CPU oracle.

The reference (rwl/blu, Rust) ships no golden vectors and cannot be executed in
this environment (no Rust toolchain), so these fixtures pin the ORACLE's output,
not the reference's: they guard the oracle against regressions and give the
GPU box (where /root/reference does not exist either) a fixed target.

Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from blu_amd import keys as K  # noqa: E402
from blu_amd.matrices import simple_rs  # noqa: E402
from oracle import orc  # noqa: E402

COUNTERS = ["RANK", "MATRIX_NZ", "BUMP_SIZE", "BUMP_NZ", "L_NZ", "U_NZ", "NSEARCH_PIVOT", "FACTOR_FLOPS", "RANKDEF"]
FSTATS = ["MIN_PIVOT", "MAX_PIVOT", "CONDEST_L", "CONDEST_U", "NORM_L", "NORM_U", "NORMEST_L_INV",
          "NORMEST_U_INV", "ONENORM", "INFNORM", "RESIDUAL_TEST", "UPDATE_COST_DENOM"]

# name -> generator parameters (lp_basis) ; small cases only (oracle runs in ms)
CASES = {
    "lp_m200_k6_bw6": dict(m=200, k=6, bw=6, tri_frac=0.5, offscale=0.3, seed=3),
    "lp_m500_k8_bw8_dense_end": dict(m=500, k=8, bw=16, tri_frac=0.25, offscale=1.0, seed=7),
    "lp_m2000_k8_bw8": dict(m=2000, k=8, bw=8, tri_frac=0.5, offscale=0.3, seed=1),
    "lp_m1000_k10_tri": dict(m=1000, k=10, bw=12, tri_frac=1.0, offscale=0.2, seed=11),
}


def run(colptr, rowidx, values, params=None, cap=None):
    m = len(colptr) - 1
    o = orc.OracleBLU(m, cap if cap is not None else 16 * len(rowidx) + 64)
    for k, v in (params or {}).items():
        o.set_param(k, v)
    st = o.factorize(colptr[:-1], colptr[1:], rowidx, values)
    out = dict(status=np.int64(st), d3_hits=np.int64(o.d3_hits()))
    if st in (K.OK, K.WARNING_SINGULAR_MATRIX):
        out.update(o.get_factors())
        for c in COUNTERS:
            out["stat_" + c] = np.int64(o.stat(getattr(K, "STAT_" + c)))
        for c in FSTATS:
            out["stat_" + c] = np.float64(o.stat(getattr(K, "STAT_" + c)))
    return out


def main():
    orc.build()
    cp, ri, v, b, x = simple_rs()
    out = run(cp, ri, v, cap=len(ri))  # BLU::new(n, a.len()) as in examples/simple.rs:36
    np.savez_compressed(os.path.join(HERE, "simple_rs.npz"), colptr=cp, rowidx=ri, values=v, rhs=b, sol=x, **out)
    for name, g in CASES.items():
        cp, ri, v = orc.gen_lp_basis(g["m"], g["k"], g["bw"], g["tri_frac"], g["seed"], g["offscale"])
        out = run(cp, ri, v)
        assert out["status"] == 0, (name, out["status"])
        np.savez_compressed(os.path.join(HERE, name + ".npz"), gen=np.array([g["m"], g["k"], g["bw"], g["tri_frac"], g["offscale"], g["seed"]]),
                            colptr=cp, rowidx=ri, values=v, **out)
        print(name, "nnz", len(ri), "l_nz", out["stat_L_NZ"], "u_nz", out["stat_U_NZ"], "d3_hits", out["d3_hits"])


if __name__ == "__main__":
    main()
