"""Randomized parity sweep inside the suite (run with -m gpu on an MI355X).

tools/fuzz_gpu.py draws small bases with random generator parameters, LU parameters (nzbias, row search,
maxsearch, reltol, pad, stretch, sparse_thres), workgroup sizes, capacity hints and numerically null columns,
and requires status, canonical factors, counters, statistics, pivot-path counts, d3 events, solve_dense and
solve_sparse to be IDENTICAL to the CPU oracle's.  The sweep is the 1500-case run of seed 777 that lost its GPU
box in round 1 (cause: DESIGN.md section 7), plus 300 cases of seed 12345, in slices of 100.

Every slice runs in a FRESH child process under a timeout (a child is started; a process that has touched
the GPU is never exec'ed over).  The child logs the tag of a case before its first GPU call, so a failure
reports the parameters that caused it.  After a slice that hung or was killed no further slice is started.
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SLICES = [(777, 100 * i, 100) for i in range(15)] + [(12345, 100 * i, 100) for i in range(3)]
_stop = {"why": ""}


@pytest.mark.parametrize("seed,start,count", SLICES, ids=lambda v: str(v))
def test_fuzz_slice(seed, start, count):
    if _stop["why"]:
        pytest.skip("not started: " + _stop["why"])
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    log = os.path.join(out, "fuzz_s%d_%04d.log" % (seed, start))
    cmd = [sys.executable, os.path.join(ROOT, "tools", "fuzz_gpu.py"), "--seed", str(seed), "--start", str(start),
           "--count", str(count), "--log", log]
    try:
        r = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=240)
    except subprocess.TimeoutExpired:
        last = _last_started(log)
        _stop["why"] = "slice (%d, %d) hung; last case started: %s" % (seed, start, last)
        pytest.fail(_stop["why"])
    text = r.stdout.decode(errors="replace")
    if r.returncode < 0 or r.returncode >= 124:
        _stop["why"] = "slice (%d, %d) was killed (rc %d); last case started: %s" % (seed, start, r.returncode, _last_started(log))
        pytest.fail(_stop["why"] + "\n" + text[-2000:])
    assert r.returncode == 0, "first failing case: %s\n%s" % (_last_started(log), text[-3000:])
    assert ("all %d cases of seed %d from %d identical" % (count, seed, start)) in text


def _last_started(log):
    try:
        lines = [ln for ln in open(log).read().splitlines() if ln.startswith("start ")]
        return lines[-1][6:] if lines else "(none)"
    except OSError:
        return "(no log)"


# ---- the update path (tools/fuzz_update_gpu.py): lock step with the CPU twin on random bases and replacement sequences
UPD_SLICES = [(4242, 60 * i, 60) for i in range(8)]


@pytest.mark.parametrize("seed,start,count", UPD_SLICES, ids=lambda v: str(v))
def test_update_fuzz_slice(seed, start, count):
    if _stop["why"]:
        pytest.skip("not started: " + _stop["why"])
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    log = os.path.join(out, "fuzzupd_s%d_%04d.log" % (seed, start))
    cmd = [sys.executable, os.path.join(ROOT, "tools", "fuzz_update_gpu.py"), "--seed", str(seed), "--start", str(start),
           "--count", str(count), "--log", log]
    try:
        r = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    except subprocess.TimeoutExpired:
        _stop["why"] = "update slice (%d, %d) hung; last case started: %s" % (seed, start, _last_started(log))
        pytest.fail(_stop["why"])
    text = r.stdout.decode(errors="replace")
    if r.returncode < 0 or r.returncode >= 124:
        _stop["why"] = "update slice (%d, %d) was killed (rc %d); last case started: %s" % (seed, start, r.returncode, _last_started(log))
        pytest.fail(_stop["why"] + "\n" + text[-2000:])
    assert r.returncode == 0, "first failing case: %s\n%s" % (_last_started(log), text[-3000:])
    assert ("all %d update cases of seed %d from %d identical" % (count, seed, start)) in text


# ---- mid-size bases (m = 1500 .. 9000): the LDS rings of the chain pipeline wrap, operands come from beyond its window
def test_fuzz_slice_mid_size():
    if _stop["why"]:
        pytest.skip("not started: " + _stop["why"])
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    log = os.path.join(out, "fuzz_mid_s31.log")
    cmd = [sys.executable, os.path.join(ROOT, "tools", "fuzz_gpu.py"), "--seed", "31", "--start", "0", "--count", "40", "--mmin", "1500",
           "--mmax", "9000", "--log", log]
    try:
        r = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=400)
    except subprocess.TimeoutExpired:
        _stop["why"] = "mid-size slice hung; last case started: %s" % _last_started(log)
        pytest.fail(_stop["why"])
    text = r.stdout.decode(errors="replace")
    if r.returncode < 0 or r.returncode >= 124:
        _stop["why"] = "mid-size slice was killed (rc %d); last case started: %s" % (r.returncode, _last_started(log))
        pytest.fail(_stop["why"] + "\n" + text[-2000:])
    assert r.returncode == 0, "first failing case: %s\n%s" % (_last_started(log), text[-3000:])
    assert "all 40 cases of seed 31 from 0 identical" in text


# ---- the self-checking library (make ewcheck, built by __graft_entry__.build()): inside the pivot loop every early and
# every speculative search of the next pivot is compared with the ordinary search -- candidates, count, key, staged
# entries -- and the first difference ends the factorization with ST_ERROR.  A sweep of fresh cases and a few mid-size
# bases under it, results identical to the oracle's as everywhere else.
@pytest.mark.parametrize("args,tag", [(["--seed", "2718", "--start", "0", "--count", "120"], "all 120 cases of seed 2718 from 0 identical"),
                                      (["--seed", "32", "--start", "0", "--count", "12", "--mmin", "1500", "--mmax", "9000"],
                                       "all 12 cases of seed 32 from 0 identical")], ids=["small", "mid"])
def test_fuzz_slice_self_checking_library(args, tag):
    if _stop["why"]:
        pytest.skip("not started: " + _stop["why"])
    import blu_amd
    libpath = blu_amd.build_library(selfcheck=True)  # (a no-op when __graft_entry__.build() has run)
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    log = os.path.join(out, "fuzz_selfcheck_s%s.log" % args[1])
    cmd = [sys.executable, os.path.join(ROOT, "tools", "fuzz_gpu.py")] + args + ["--log", log]
    env = dict(os.environ, BLU_HIP_LIB=libpath)
    try:
        r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=400)
    except subprocess.TimeoutExpired:
        _stop["why"] = "self-checking slice hung; last case started: %s" % _last_started(log)
        pytest.fail(_stop["why"])
    text = r.stdout.decode(errors="replace")
    if r.returncode < 0 or r.returncode >= 124:
        _stop["why"] = "self-checking slice was killed (rc %d); last case started: %s" % (r.returncode, _last_started(log))
        pytest.fail(_stop["why"] + "\n" + text[-2000:])
    assert r.returncode == 0, "first failing case: %s\n%s" % (_last_started(log), text[-3000:])
    assert "self-checking build" in text, text[:300]  # (the child really ran on that library)
    assert tag in text


# ---- the batch entry (blu_hip_factorize_batch -> k_pivot_loop_wave, one wave per basis): random batches of mixed sizes, and
# slices of the single-matrix sweep with the one-wave kernel forced onto them (BLU_PIVOT_KERNEL=1, fresh seed)
def _run_slice(cmd, log, what, env=None, timeout=400):
    if _stop["why"]:
        pytest.skip("not started: " + _stop["why"])
    try:
        r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
    except subprocess.TimeoutExpired:
        _stop["why"] = "%s hung; last case started: %s" % (what, _last_started(log))
        pytest.fail(_stop["why"])
    text = r.stdout.decode(errors="replace")
    if r.returncode < 0 or r.returncode >= 124:
        _stop["why"] = "%s was killed (rc %d); last case started: %s" % (what, r.returncode, _last_started(log))
        pytest.fail(_stop["why"] + "\n" + text[-2000:])
    assert r.returncode == 0, "first failing case: %s\n%s" % (_last_started(log), text[-3000:])
    return text


@pytest.mark.parametrize("start", [0, 15])
def test_batch_fuzz_slice(start):
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    log = os.path.join(out, "fuzzbatch_s2025_%04d.log" % start)
    cmd = [sys.executable, os.path.join(ROOT, "tools", "fuzz_batch_gpu.py"), "--seed", "2025", "--start", str(start), "--count", "15", "--log", log]
    text = _run_slice(cmd, log, "batch slice %d" % start)
    assert ("all 15 batches of seed 2025 from %d identical" % start) in text


def test_batch_fuzz_slice_one_wave_kernel_forced():
    """The same kind of random batches with k_pivot_loop_wave forced (BLU_PIVOT_KERNEL=1, read when a handle is created):
    batches of 3..24 members take the two-wave kernel by default, the one-wave kernel is the default only beyond 2048."""
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    log = os.path.join(out, "fuzzbatch_wave_s2026.log")
    cmd = [sys.executable, os.path.join(ROOT, "tools", "fuzz_batch_gpu.py"), "--seed", "2026", "--start", "0", "--count", "12", "--log", log]
    text = _run_slice(cmd, log, "one-wave batch slice", env=dict(os.environ, BLU_PIVOT_KERNEL="1"))
    assert "all 12 batches of seed 2026 from 0 identical" in text and "pivot kernels [1]" in text


@pytest.mark.parametrize("args,tag", [(["--seed", "9090", "--start", "0", "--count", "150"], "all 150 cases of seed 9090 from 0 identical"),
                                      (["--seed", "33", "--start", "0", "--count", "15", "--mmin", "1500", "--mmax", "9000"],
                                       "all 15 cases of seed 33 from 0 identical")], ids=["small", "mid"])
def test_fuzz_slice_one_wave_kernel_forced(args, tag):
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    log = os.path.join(out, "fuzz_wave_s%s.log" % args[1])
    cmd = [sys.executable, os.path.join(ROOT, "tools", "fuzz_gpu.py")] + args + ["--log", log]
    # ("mid" with BLU_PIVOT_REGS=4: the variant of the kernel with the register budget of four waves per SIMD, what a batch of
    # more than 3072 bases runs; "small" takes the _r3 variant like every batch that is resident at three)
    env = dict(os.environ, BLU_PIVOT_KERNEL="1", **({"BLU_PIVOT_REGS": "4"} if "--mmin" in args else {}))
    text = _run_slice(cmd, log, "one-wave slice", env=env)
    assert tag in text


@pytest.mark.parametrize("args,tag", [(["--seed", "9191", "--start", "0", "--count", "150"], "all 150 cases of seed 9191 from 0 identical"),
                                      (["--seed", "34", "--start", "0", "--count", "15", "--mmin", "1500", "--mmax", "9000"],
                                       "all 15 cases of seed 34 from 0 identical")], ids=["small", "mid"])
def test_fuzz_slice_two_wave_kernel_forced(args, tag):
    """the sweep with k_pivot_loop_wave2 forced for every basis (a batch takes it by itself only while every workgroup
    is resident)"""
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    log = os.path.join(out, "fuzz_wave2_s%s.log" % args[1])
    cmd = [sys.executable, os.path.join(ROOT, "tools", "fuzz_gpu.py")] + args + ["--log", log]
    text = _run_slice(cmd, log, "two-wave slice", env=dict(os.environ, BLU_PIVOT_KERNEL="3"))
    assert tag in text
