"""The batch AS BENCHED (run with -m gpu on an MI355X): the shapes bench.py times, checked against the oracle.

bench.py's batched legs run thousands of bases through blu_hip_factorize_batch with no debug environment: beyond
`wave2_max` (2048) members the library dispatches to k_pivot_loop_wave (one wave per basis, grid >> resident
workgroups), the O(nnz) kernels run `batch_grid` = one workgroup per CU over far more matrices than workgroups, the row /
column counters of k_prep / k_finish sit in their natural 144 KB LDS window (any batch of at least one basis per CU),
and storage that turns out too small is compacted / grown in memory-bounded k_compact rounds.  The small batches of
test_gpu_parity.py and tools/fuzz_batch_gpu.py reach none of that.  Here:

* >= 2304 C2-size bases of 64 distinct seeds, every handle with device inputs of its own (the host-array entry uploads
  into per-handle buffers), default dispatch (statistic 118 == 1: k_pivot_loop_wave);
* >= 320 C4-size bases of 32 distinct seeds (statistic 118 == 3: k_pivot_loop_wave2, natural LDS window: n >= CUs);
* each batch three times: fresh handles at the bench's hint nnz/2; fresh handles of which every other one starts at
  hint nnz/8 (k_compact rounds / growth and relaunches run: asserted); and the same handles again, warm;
* a stratified sample -- first, last and every 67th (9th) member, >= 32 members -- against its own FAITHFUL oracle run
  (d3_hits == 0): the six integer arrays of get_factors (get_factors.rs:48-180) bit-exact, values bit-exact, counters,
  pivots per pivot routine, every statistic of the tail; every member's status, rank, l_nz, u_nz and factor_flops
  against the oracle run of its seed.
"""
import numpy as np
import pytest

from blu_amd import keys as K
from blu_amd.matrices import CONFIGS
from tests import util

pytestmark = pytest.mark.gpu
FSTATS = ("CONDEST_L", "CONDEST_U", "NORM_L", "NORM_U", "NORMEST_L_INV", "NORMEST_U_INV", "ONENORM", "INFNORM",
          "RESIDUAL_TEST", "MIN_PIVOT", "MAX_PIVOT")
CHEAP = ("RANK", "L_NZ", "U_NZ", "FACTOR_FLOPS", "NSEARCH_PIVOT", "BUMP_NZ")


@pytest.fixture(scope="module")
def blu():
    import blu_amd
    if blu_amd.lib().blu_hip_device_count() < 1:
        pytest.fail("no HIP device visible: the GPU tests must run on the MI355X box")
    return blu_amd


def _check_member(h, o, tag):
    fg, fo = h.get_factors(), o.get_factors()
    for k in util.INT_KEYS + util.VAL_KEYS:
        assert np.array_equal(fg[k], fo[k]), (tag, k)
    for c in util.COUNTERS:
        assert int(h.stat(getattr(K, "STAT_" + c))) == int(o.stat(getattr(K, "STAT_" + c))), (tag, c)
    for kind in range(6):
        assert h.stat(51 + kind) == o.stat(51 + kind), (tag, "pivots of kind", kind)
    for c in FSTATS:
        assert h.stat(getattr(K, "STAT_" + c)) == o.stat(getattr(K, "STAT_" + c)), (tag, c)
    assert int(h.stat(50)) == o.d3_hits(), tag  # (reference defect D3: 0 on both sides unless the test allows such a matrix)


def _batch_as_benched(blu, oracle, cfg, n, nseeds, step, expect_kernel, allow_d3=False):
    c = CONFIGS[cfg]
    mats = [blu.gen_lp_basis(c["m"], c["k"], c["bw"], c["tri_frac"], 5000 + s, c["offscale"]) for s in range(nseeds)]
    oracles = {}

    def oracle_of(s):
        if s not in oracles:
            cp, ri, v = mats[s]
            # the FAITHFUL restatement (i32 cancellation mask, defect D3) with d3_hits == 0 asserted; allow_d3: a matrix that does
            # hit D3 (the reference has no defined result there) is checked against the 64-bit-mask oracle instead
            o, so = util.oracle_factorize(oracle, cp, ri, v, cap=16 * len(ri), allow_d3=allow_d3)
            assert so == K.OK
            oracles[s] = o
        return oracles[s]

    sample = sorted(set([0, n - 1] + list(range(0, n, step))))
    assert len(sample) >= 32
    member_mats = [mats[k % nseeds] for k in range(n)]

    def run_and_check(hs, what):
        sts = blu.factorize_batch(hs, mats=member_mats)  # no debug environment, default dispatch
        assert all(s == K.OK for s in sts), (what, [(k, s) for k, s in enumerate(sts) if s != K.OK][:8])
        assert int(hs[0].stat(118)) == expect_kernel and int(hs[-1].stat(118)) == expect_kernel, what
        assert int(hs[0].stat(120)) == 3, what  # (all workgroups resident with the registers of three waves per SIMD: the _r3 variant)
        if what == "hint nnz/2":  # as benched: k_prep and k_finish fill through buckets (k_bucket.h; statistic 119)
            assert all(int(h.stat(119)) == 3 for h in hs), what
        for k in sample:
            _check_member(hs[k], oracle_of(k % nseeds), "%s %s member %d (seed %d)" % (cfg, what, k, 5000 + k % nseeds))
        # every member: the counters that cost one call each, against the oracle run of its seed (run for the sample
        # above when the seed is among its members', else now)
        for k, h in enumerate(hs):
            o = oracle_of(k % nseeds)
            for cn in CHEAP:
                assert int(h.stat(getattr(K, "STAT_" + cn))) == int(o.stat(getattr(K, "STAT_" + cn))), (cfg, what, k, cn)

    # (1) as bench.py creates them
    hs = [blu.BLU(c["m"], len(member_mats[k][1]) // 2) for k in range(n)]
    run_and_check(hs, "hint nnz/2")
    for h in hs:
        h.close()
    # (2) every other handle far too small: arenas / factors outgrow their storage inside the pivot loop, the batch
    # compacts and grows in rounds and relaunches the pivot kernel
    hs = [blu.BLU(c["m"], len(member_mats[k][1]) // (8 if k % 2 else 2)) for k in range(n)]
    run_and_check(hs, "hint nnz/8 for every other handle")
    cold_launches = hs[0].stat(K.STAT_DEV_RELAUNCHES)
    assert cold_launches > 1
    # (3) the same handles again, warm (what the timed repetitions of bench.py are); storage only ever grows, so a warm
    # step never needs more launches of the pivot kernel than the cold one did
    run_and_check(hs, "warm repetition")
    assert 1 <= hs[0].stat(K.STAT_DEV_RELAUNCHES) <= cold_launches
    for h in hs:
        h.close()


def test_c2_size_batch_beyond_wave2_max_default_dispatch(blu, oracle):
    """2304 C2-size bases: more workgroups than k_pivot_loop_wave2's residency limit (2048) -> k_pivot_loop_wave, the
    kernel behind bench.py's C4-size (3072) and C2-size (4096) legs."""
    _batch_as_benched(blu, oracle, "C2", 2304, 64, 67, expect_kernel=1)


def test_c4_size_batch_two_wave_kernel_natural_window(blu, oracle):
    """320 C4-size bases: every workgroup resident -> k_pivot_loop_wave2 (the kernel behind the C3-size leg), at least one
    basis per CU -> the natural LDS window of k_prep / k_finish, several 36 864-line windows per matrix."""
    _batch_as_benched(blu, oracle, "C4", 320, 32, 9, expect_kernel=3)


def test_c3_size_batch_two_wave_kernel(blu, oracle):
    """256 bases of the 100k size (BASELINE.json's headline configuration; bench.py times 1536 of them and verifies eight):
    one basis per CU -> the natural LDS window with three windows per matrix, k_pivot_loop_wave2; 32 distinct seeds, 33
    members compared in full, every member's counters."""
    _batch_as_benched(blu, oracle, "C3", 256, 32, 8, expect_kernel=3, allow_d3=True)  # (one of the 32 seeds hits D3 at this size)
