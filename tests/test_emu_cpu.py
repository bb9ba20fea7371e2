"""The pivot kernels stepped on the CPU: the emulation build of the library (blu_amd/csrc `make emu`: the same HIP
sources compiled for the host, one fiber per GPU thread, emu/hip/hip_runtime.h) factorizes small bases with the
one-wave-per-matrix kernel (k_pivot_loop_wave) and with the general pivot paths, and the results are compared with
the oracle -- canonical factors, counters and the number of pivots per pivot routine, bit for bit.

This is a DIAGNOSTIC build: the product path (libblu_hip.so) never loads it and has no CPU fallback.  It is what lets
the wave-level kernel logic be checked in the CPU suite (and run under AddressSanitizer: `make emu_asan`).
Each case runs in a child process: the library path is fixed when blu_amd is first imported."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "blu_amd", "csrc")
EMU = os.path.join(ROOT, "blu_amd", "libblu_emu.so")

CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %(root)r)
import blu_amd
from blu_amd import keys as K
from oracle import orc
assert b"gfx950" in blu_amd.lib().blu_hip_version()
spec = %(spec)r
no_fast = %(no_fast)r
mats = [orc.gen_lp_basis(m, k, bw, tri, seed, offs) for (m, k, bw, tri, seed, offs) in spec]
hs = [blu_amd.BLU(len(cp) - 1, len(ri)) for cp, ri, v in mats]
for h in hs:
    if no_fast:
        h.dbg_set_no_fast(True)
if len(hs) == 1:
    cp, ri, v = mats[0]
    st = [hs[0].factorize(cp[:-1], cp[1:], ri, v)]
else:
    st = blu_amd.factorize_batch(hs, mats)
fast = 0
for g, (cp, ri, v), s in zip(hs, mats, st):
    m = len(cp) - 1
    o = orc.OracleBLU(m, 64 * len(ri) + 1024)
    o.set_fix_d3(True)
    so = o.factorize(cp[:-1], cp[1:], ri, v)
    assert s == so, (s, so)
    fg, fo = g.get_factors(), o.get_factors()
    for key in ("rowperm", "colperm", "l_colptr", "l_rowidx", "u_colptr", "u_rowidx", "l_value", "u_value"):
        assert np.array_equal(fg[key], fo[key]), key
    for cn in ("RANK", "L_NZ", "U_NZ", "NSEARCH_PIVOT", "FACTOR_FLOPS", "BUMP_NZ", "MATRIX_NZ"):
        assert g.stat(getattr(K, "STAT_" + cn)) == o.stat(getattr(K, "STAT_" + cn)), cn
    for kind in range(6):
        assert g.stat(51 + kind) == o.stat(51 + kind), ("pivot kind", kind)
    assert int(g.stat(50)) == o.d3_hits()
    fast += int(g.stat(110)) + int(g.stat(111))
print("FILLS", " ".join(str(int(g.stat(119))) for g in hs))
print("FAST", fast)
"""


@pytest.fixture(scope="module")
def emu_lib():
    subprocess.check_call(["make", "-s", "-C", CSRC, "emu"])
    assert os.path.exists(EMU)
    return EMU


def run_child(emu_lib, spec, kernel, no_fast=False, extra_env=None, fills=None):
    env = dict(os.environ, BLU_HIP_LIB=emu_lib, BLU_PIVOT_KERNEL=str(kernel), **(extra_env or {}))
    out = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "spec": spec, "no_fast": no_fast}], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    if fills is not None:  # statistic 119 of every handle: which of k_prep (1) / k_finish (2) filled through buckets
        assert [int(x) for x in out.stdout.split("FILLS")[-1].split("FAST")[0].split()] == fills, out.stdout[-500:]
    return int(out.stdout.split("FAST")[-1])


@pytest.mark.parametrize("spec", [(300, 8, 8, 0.5, 1, 0.3), (500, 10, 9, 0.5, 7, 0.3), (400, 6, 20, 0.2, 3, 1.0), (250, 11, 30, 0.0, 5, 0.1)],
                         ids=["banded", "c3-like", "wide", "no-triangle"])
def test_wave_kernel_on_the_cpu(emu_lib, spec):
    """k_pivot_loop_wave (flattened small / singleton-column paths, hand-over, general fallbacks), one basis"""
    fast = run_child(emu_lib, [spec], kernel=1)
    assert fast > spec[0] // 2  # most pivots took the flattened paths


@pytest.mark.parametrize("spec", [(300, 8, 8, 0.5, 1, 0.3), (400, 6, 20, 0.2, 3, 1.0), (250, 11, 30, 0.0, 5, 0.1)], ids=["banded", "wide", "no-triangle"])
def test_two_wave_kernel_on_the_cpu(emu_lib, spec):
    """k_pivot_loop_wave2: the lines of a small pivot dealt out to two waves, LDS-only barriers between the phases, the
    walk of the next search begun while the other wave updates the rows"""
    fast = run_child(emu_lib, [spec], kernel=3)
    assert fast > spec[0] // 2


def test_two_wave_general_paths_on_the_cpu(emu_lib):
    """the general pivot paths as a workgroup of two waves"""
    assert run_child(emu_lib, [(220, 8, 8, 0.5, 1, 0.3)], kernel=3, no_fast=True) == 0


def test_wave_kernel_batch_on_the_cpu(emu_lib):
    """the batch entry: three bases of different sizes, one wave each"""
    fast = run_child(emu_lib, [(200, 8, 8, 0.5, 1, 0.3), (333, 8, 8, 0.5, 2, 0.3), (150, 5, 4, 0.8, 3, 0.6)], kernel=0)
    assert fast > 300


def test_batch_with_two_workgroups_on_the_cpu(emu_lib):
    """k_prep / k_setup / k_finish as two workgroups that take the batch's matrices one after the other (BLU_BATCH_GRID),
    their row / column counters through LDS windows of 1 KB (BLU_LDS_WINDOW: the path of a large batch)"""
    fast = run_child(emu_lib, [(200, 8, 8, 0.5, 1, 0.3), (333, 8, 8, 0.5, 2, 0.3), (150, 5, 4, 0.8, 3, 0.6), (120, 6, 6, 0.5, 4, 0.3),
                               (260, 7, 9, 0.3, 5, 0.5)], kernel=0, extra_env={"BLU_BATCH_GRID": "2", "BLU_LDS_WINDOW": "2", "BLU_LDS_WINDOW_BYTES": "1024"})
    assert fast > 400


def test_batch_fills_through_buckets_on_the_cpu(emu_lib):
    """the two-phase fill of k_prep / k_finish (k_bucket.h) with a window of 16 KB: buckets of 672 entries, so every one of
    these matrices is several buckets (the fourth has a U column of 112 entries, longer than the 96 a bucket keeps as slack:
    its k_finish takes the window sweeps); the 1 KB windows of the test above are too small for buckets (statistic 119 == 0)"""
    specs = [(200, 8, 8, 0.5, 1, 0.3), (333, 8, 8, 0.5, 2, 0.3), (150, 5, 4, 0.8, 3, 0.6), (420, 9, 12, 0.3, 4, 0.3), (260, 7, 9, 0.3, 5, 0.5)]
    env = {"BLU_BATCH_GRID": "2", "BLU_LDS_WINDOW": "2"}
    fast = run_child(emu_lib, specs, kernel=0, extra_env=dict(env, BLU_LDS_WINDOW_BYTES="16384"), fills=[3, 3, 3, 1, 3])
    assert fast > 600
    run_child(emu_lib, specs[:2], kernel=0, extra_env=dict(env, BLU_LDS_WINDOW_BYTES="1024"), fills=[0, 0])


def test_general_paths_on_the_cpu(emu_lib):
    """the general pivot paths alone (no flattened paths), as a workgroup of one wave"""
    assert run_child(emu_lib, [(220, 8, 8, 0.5, 1, 0.3)], kernel=1, no_fast=True) == 0


def test_step_check_of_the_line_records_on_the_cpu(emu_lib):
    """tools/gpu_stepcheck.py under the emulation build: the library is stopped every 25 pivots and its COMPLETE active
    submatrix -- ordered line contents, column maxima, the count lists with their heads, pivots, partial L / U: what the
    32-byte line records (blu_dev.h: LineRec, HeadRec) hold, read out through blu_hip_dbg_active_state -- is compared with
    the oracle's arrays (file.rs / list.rs representation), for the one-wave kernel and for the general paths."""
    for extra, kernel in ((["--block", "64"], "1"), (["--block", "64", "--no-fast"], "1"), (["--block", "128"], "3")):
        env = dict(os.environ, BLU_HIP_LIB=emu_lib, BLU_PIVOT_KERNEL=kernel)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_stepcheck.py"), "160,7,8,0.5,0.3,3", "--step", "25"] + extra,
                             env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0 and "FACTORS IDENTICAL" in out.stdout and "MISMATCH" not in out.stdout, out.stdout[-2500:] + out.stderr[-2500:]
