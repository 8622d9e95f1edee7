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


@pytest.mark.parametrize("mode", ["sparse", "sharded", "tileband", "auto"])
def test_two_ranks_through_the_launcher(mode):
    env = dict(os.environ, RTGS_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("LOCAL_RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--prewarm", "4",
           "--repeats", "1", "--gaussians", "60000", "--no-cpu-baseline", "--no-surface", "--no-schedule",
           "--config5-gaussians", "80000"] + ([] if mode == "auto" else ["--mode", mode])
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 0 and d["value"] > 0 and d["steps"] == 3
    if mode == "auto":                                  # what the driver's N > 1 runs get: ONE stream, split into tile bands
        mode = "tileband"
    assert d["config"]["mode"] == mode and d["scaling"] == ("strong" if mode == "tileband" else "weak")
    assert d["metric"] == "hot_path_units_per_sec" and d["unit"] == "units/s"
    assert d["roofline"]["frac"] > 0
    if mode != "sharded":
        assert d["strong_scaling_one_view"]["n_gpus"] == 2
    # BASELINE configs[4] rides in every line: sharded, 10 iterations, the four-way split of SURVEY.md 8d
    c5 = d["config5"]
    assert c5["n_gpus"] == 2 and c5["mode"] == "sharded" and c5["iterations"] == 10 and c5["ms_per_iteration"] > 0
    assert set(c5["split_ms"]) == {"render_fwd_and_loss", "render_bwd", "collective", "adam"}
    assert c5["split_ms"]["render_fwd_and_loss"] > 0 and c5["split_ms"]["render_bwd"] > 0 and c5["split_ms"]["adam"] > 0
    assert c5["split_ms"]["collective"] > 0                     # two ranks: the exchange exists
    if mode == "tileband":
        w = d["weak_scaling_one_view_per_rank"]
        assert w["n_gpus"] == 2 and w["views_per_sec"] > 0
    else:
        assert d["weak_scaling_one_view_per_rank"] is None
