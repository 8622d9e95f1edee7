"""bench.py's multi-rank launcher path on the one-GPU lease: `python bench.py --gpus 2 --mode X` spawns its two ranks
itself (torch.distributed.run, 127.0.0.1), both share the GPU over gloo (RTGS_DIST_BACKEND=gloo: RCCL refuses two ranks
on one device), every multi-GPU form of the map step runs, rank 0 prints the contract's JSON line.  On the driver's
8-GPU node the same code runs one rank per GPU over RCCL (backend nccl) - this test keeps the launcher, the world-size /
backend assertions and the three forms from rotting in between (BASELINE configs[4])."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["sparse", "sharded", "tileband"])
def test_two_ranks_through_the_launcher(mode):
    env = dict(os.environ, RTGS_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("LOCAL_RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--prewarm", "4",
           "--repeats", "1", "--gaussians", "60000", "--no-cpu-baseline", "--no-surface", "--no-schedule", "--mode", mode]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 0 and d["value"] > 0 and d["steps"] == 3
    assert d["config"]["mode"] == mode and d["scaling"] == ("strong" if mode == "tileband" else "weak")
    assert d["roofline"]["frac"] > 0
    if mode != "sharded":
        assert d["strong_scaling_one_view"]["n_gpus"] == 2
