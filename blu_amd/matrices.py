"""Input matrices used by tests, fixtures and the benchmark.

`simple_rs()` is the literal 10x10 matrix of the reference's only executable
artefact, examples/simple.rs:20-33 (data, not code).  The synthetic LP-basis
configurations are the ones BASELINE.json names; their generator parameters
are fixed here (see DESIGN.md "Synthetic inputs" for why bw/offscale differ
from SURVEY.md's provisional values).
"""
import numpy as np


def simple_rs():
    """examples/simple.rs:20-33: returns (colptr, rowidx, values, rhs, solution)."""
    arow = [0, 7, 8, 1, 4, 9, 2, 9, 3, 6, 7, 8, 9, 1, 4, 5, 3, 6, 9, 0, 3, 7, 8, 0, 3, 7, 8, 1, 2, 3, 6, 9]
    acolst = [0, 3, 6, 8, 13, 15, 16, 19, 23, 27, 32]
    a = [2.1, 0.14, 0.09, 1.1, 0.06, 0.03, 1.7, 0.04, 1.0, 0.32, 0.19, 0.32, 0.44, 0.06, 1.6, 2.2,
         0.32, 1.9, 0.43, 0.14, 0.19, 1.1, 0.22, 0.09, 0.32, 0.22, 2.4, 0.03, 0.04, 0.44, 0.43, 3.2]
    b = [0.403, 0.28, 0.55, 1.504, 0.812, 1.32, 1.888, 1.168, 2.473, 3.695]
    x = [0.1 * (i + 1) for i in range(10)]
    return (np.array(acolst, np.uint64), np.array(arow, np.uint64), np.array(a, np.float64),
            np.array(b, np.float64), np.array(x, np.float64))


# name -> lp_basis(m, k, bw, tri_frac, offscale, seed)   (BASELINE.json configs[1..3])
CONFIGS = {
    "C2": dict(m=10_000, k=8, bw=8, tri_frac=0.5, offscale=0.3, seed=1),
    "C3": dict(m=100_000, k=10, bw=9, tri_frac=0.5, offscale=0.3, seed=1),
    # C4: 8 independent bases, seeds 1..8, one per GPU
    "C4": dict(m=50_000, k=10, bw=9, tri_frac=0.5, offscale=0.3, seed=1),
}
