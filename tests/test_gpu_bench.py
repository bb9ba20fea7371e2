"""bench.py on the GPU box under the launcher the driver uses (`python -m torch.distributed.run`), at world size 1:
the child is started before anything in this process touches the GPU API on its behalf, initialises the `nccl`
(= RCCL) process group, and runs the barrier / MAX / SUM reductions of blu_amd/shard.py around the timed region --
the N > 1 code path, executed on real hardware with the one GPU a box has."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_under_torch_distributed_run_world_size_one():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--config", "C4", "--steps", "1", "--warmup", "0",
           "--batch", "0", "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 1 and rec["scaling"] == "weak" and rec["value"] > 0
    assert rec["config"]["workload"].startswith("C4") and rec["roofline"]["bound"] == "hbm"
    assert "process group: nccl" in out.stderr + out.stdout


def test_bench_batched_leg_verifies_its_members():
    """The secondary (batched) measurement of bench.py end to end on a small batch: 64 distinct matrices with per-handle
    device inputs, cold step + three warm steps, and -- after the timed steps -- eight members checked against the CPU
    oracle (`verified_members`); the roofline object of the leg names the batch pivot kernel the library chose."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "C2", "--steps", "2", "--warmup", "1", "--batch", "300",
           "--no-batch-sizes", "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    b = rec["batched"]
    assert b["bases_in_flight_per_gpu"] == 300 and b["distinct_matrices"] == 64 and b["verified_members"] == 8
    assert len(b["seconds_warm_steps"]) == 3 and b["cold_first_step"]["seconds"] > 0
    assert "k_pivot_loop_wave2" in b["roofline"]["kernel"] and 0 < b["roofline"]["frac"] < 1
    assert rec["config"]["generator_note"].startswith("bw = 8, not SURVEY.md 8d's provisional 16")  # (C2 is lp_basis(10 000, 8, bw 8, ...))
