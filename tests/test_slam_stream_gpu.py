"""BASELINE.json configs[2] in miniature: ICP tracking + online map optimisation on a synthetic RGB-D stream, every
kernel of the path in one loop (tools/mini_slam.py follows the reference's call order, slam.py:56-90)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


def test_tracking_and_mapping_a_synthetic_stream():
    from rtg_slam_amd import synth
    import mini_slam
    cam = synth.CameraSpec(240, 320, 160.0, 160.0, 159.5, 119.5)
    stats = mini_slam.run(cam, n_frames=8, iters_per_frame=10, first_frame_iters=30, samples_first=30000, samples_new=3000)
    for s in stats:
        print(s)
    last = stats[-1]
    # frame-to-model tracking holds the trajectory (the stream moves <= 2 cm / 1 degree per frame: 8 frames ~ 10 cm)
    assert max(s["trans_err_m"] for s in stats) < 0.01, stats
    assert max(s["rot_err_deg"] for s in stats) < 0.3, stats
    # the map explains the frames it was optimised on
    assert last["psnr"] > 24.0 and last["depth_l1_m"] < 0.02, last
    assert min(s["covered"] for s in stats) > 0.97
    assert stats[0]["gaussians"] > 20000 and last["gaussians"] < 2 * stats[0]["gaussians"]
