"""Synthetic workloads beyond a single factorize: the column-replacement stream of BASELINE.json configs[4]
(C5: Forrest-Tomlin update + sparse re-solve loop on the 100k basis, 1000 column modifications).

SURVEY.md 8d asks for "column j_t (SplitMix64 seed 99) replaced by a fresh column drawn by the same rule".  The
generator's hidden permutations are not exposed by lp_basis, so the fresh column keeps the PATTERN of the column
it replaces (which is a column "drawn by the same rule") and draws new VALUES by the same rule: the first entry
of a generated column is its diagonal, |v| = 1 + u, the others are |v| = offscale * (0.1 + 0.9 u), signs by a
further draw < 0.5 -- all from one SplitMix64 stream with state = seed.  Deterministic, no numpy RNG."""
import numpy as np

_MASK = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & _MASK

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & _MASK
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK
        return z ^ (z >> 31)

    def u(self):
        return (self.next() >> 11) * (1.0 / 9007199254740992.0)


def column_modifications(colptr, rowidx, nmods, offscale, seed=99):
    """Yields (j, rows, values): column j of the basis is to be replaced by the sparse column (rows, values)."""
    rng = SplitMix64(seed)
    m = len(colptr) - 1
    for _ in range(nmods):
        j = int(rng.u() * m)
        a, b = int(colptr[j]), int(colptr[j + 1])
        rows = np.asarray(rowidx[a:b], dtype=np.uint64).copy()
        vals = np.empty(b - a)
        for t in range(b - a):
            u1, u2 = rng.u(), rng.u()
            mag = (1.0 + u1) if t == 0 else offscale * (0.1 + 0.9 * u1)
            vals[t] = -mag if u2 < 0.5 else mag
        yield j, rows, vals
