"""Lock-step comparison of the pivot loop with the oracle on the GPU (run with -m gpu on an MI355X): tools/gpu_stepcheck.py
stops the library every N pivots and compares its COMPLETE active submatrix with the oracle's -- ordered line contents,
column maxima, the count lists (elements and heads: list.rs representation), inverse permutations, partial L and U,
counters -- i.e. everything the 32-byte line records of round 4 (blu_dev.h: LineRec / HeadRec) carry, at hundreds of
intermediate states, for every pivot kernel and for the general paths; then the final factors and the solves."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("matrix,extra,kernel", [
    ("1500,10,9,0.5,0.3,1", ["--step", "40"], "0"),                   # sixteen-wave kernel, low-latency paths
    ("900,8,8,0.5,0.3,2", ["--step", "30", "--no-fast"], "0"),        # general paths only
    ("1200,9,10,0.4,0.4,17", ["--step", "50", "--block", "64"], "1"),  # one wave per matrix
    ("1200,9,10,0.4,0.4,17", ["--step", "50", "--block", "128"], "3"),  # two waves per matrix
    ("700,6,6,0.3,0.5,5", ["--step", "35", "--search-rows", "--nzbias", "-1"], "0"),  # row search: the row count lists too
], ids=["fast", "general", "one-wave", "two-wave", "row-search"])
def test_active_submatrix_in_lock_step_with_the_oracle(matrix, extra, kernel):
    env = dict(os.environ, BLU_PIVOT_KERNEL=kernel)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_stepcheck.py"), matrix] + extra, cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "FACTORS IDENTICAL" in out.stdout and "MISMATCH" not in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count("ok through") >= 5  # (it did stop and compare at intermediate states)
