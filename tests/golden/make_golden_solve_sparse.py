"""Regenerates tests/golden/solve_sparse.npz from the CPU oracle (see make_golden.py for what these
fixtures can and cannot pin).  Cases: the lp_m2000_k8_bw8 basis and the examples/simple.rs matrix, both
systems, right-hand sides from 1 nonzero (hypersparse branch) to m/2 (sequential branch).

Run:  python tests/golden/make_golden_solve_sparse.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from blu_amd.matrices import simple_rs  # noqa: E402
from oracle import orc  # noqa: E402


def cases(m):
    rng = np.random.default_rng(2024)
    for trans in "NT":
        for nz in sorted({1, 2, min(m, 7), max(1, m // 50), max(1, m // 2)}):
            yield trans, rng.choice(m, nz, replace=False).astype(np.int64), rng.standard_normal(nz)


def main():
    orc.build()
    out = {}
    mats = {"simple": simple_rs()[:3], "lp2000": orc.gen_lp_basis(2000, 8, 8, 0.5, 1, 0.3)}
    for name, (cp, ri, v) in mats.items():
        m = len(cp) - 1
        o = orc.OracleBLU(m, 16 * len(ri) + 64)
        assert o.factorize(cp[:-1], cp[1:], ri, v) == 0
        for n, (trans, ir, xr) in enumerate(cases(m)):
            st, il, lhs = o.solve_sparse(ir, xr, trans)
            assert st == 0
            key = "%s_%d" % (name, n)
            out[key + "_trans"] = np.array(ord(trans))
            out[key + "_irhs"], out[key + "_xrhs"] = ir, xr
            out[key + "_ilhs"], out[key + "_xlhs"] = il, lhs[il]
        out[name + "_ncases"] = np.array(n + 1)
    np.savez_compressed(os.path.join(HERE, "solve_sparse.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
