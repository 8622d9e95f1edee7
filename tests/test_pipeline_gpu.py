"""TrackMapPipeline (rtg_slam_amd/pipeline.py): the tracker || mapper hand-off of SLAM/multiprocess/system.py:12-87 as
two HIP streams of one process.  What the two-process original guarantees through its queues and what this class must
therefore guarantee through stream events:

* a frame processed through the pipeline gives the SAME pose and the SAME map update as running tracker and mapper one
  after the other (pose and rendered maps bit for bit; gradients to the rounding of their slot-summation order);
* the tracker stage sees everything the caller enqueued before `track()` (the frame's inputs, the previous map update)
  and the caller's later work sees the tracker's results after `result()`;
* results come back in submission order with two frames in flight;
* an exception of the tracker stage surfaces in `result()` of that frame and the pipeline keeps working."""
import numpy as np
import pytest
import torch

from rtg_slam_amd import synth
from tests import raster_util as ru

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class Args:
    icp_downscales = [0.25, 0.5, 1.0]
    icp_downscale_iters = [5, 5, 5]
    icp_warmup_frames = 0
    icp_use_model_depth = False
    icp_distance_threshold = 0.1
    icp_normal_threshold = 20
    icp_damping = 1e-4
    icp_sample_distance_threshold = 0.01
    icp_sample_normal_threshold = 0.01
    icp_fail_threshold = 0.02
    verbose = False


CAM = synth.CameraSpec(240, 320, 260.0, 260.0, 159.5, 119.5)


def _frames(n):
    base = synth.look_at_pose(seed=3, max_angle_deg=5, max_trans=0.3)
    poses = [base @ p for p in synth.trajectory(n, seed=4)]
    K = torch.tensor([[CAM.fx, 0, CAM.cx], [0, CAM.fy, CAM.cy], [0, 0, 1]], dtype=torch.float32, device=DEV)
    return [synth.box_room_depth(CAM, p).to(DEV) for p in poses], K


def _map_step(g, s, grads):
    """One differentiable render of a small map: the mapper stage's kernels (forward + backward)."""
    from diff_gaussian_rasterization_depth import GaussianRasterizer
    leaves = {k: g[k].detach().to(DEV).clone().requires_grad_(True) for k in ru.FIELDS}
    rast = GaussianRasterizer(raster_settings=ru.hip_settings(s, DEV))
    outs = rast(means3D=leaves["xyz"], opacities=leaves["opacity"], shs=leaves["shs"], colors_precomp=None,
                scales=leaves["scales"], rotations=leaves["rotations"], cov3D_precomp=None, normal_w=leaves["normal"],
                tile_mask=None)
    ((outs[0] * grads[0]).sum() + (outs[1] * grads[1]).sum()).backward()
    return outs[0].detach(), {k: leaves[k].grad.detach() for k in ru.FIELDS}


@pytest.mark.parametrize("reserve_cus", [0, 32])
def test_pipelined_frames_equal_sequential_execution(reserve_cus):
    """reserve_cus = 32: the mapper runs on `pipe.mapper_stream`, whose CU mask leaves 32 compute units to the tracker
    (rtgs_stream_create_reserving) - same results, it only changes where workgroups may run."""
    from rtg_slam_amd.icp import IcpTracker
    from rtg_slam_amd.pipeline import TrackMapPipeline
    depths, K = _frames(5)
    g, s = ru.make_scene(20_000, CAM, seed=6)
    gen = torch.Generator().manual_seed(1)
    grads = (torch.randn(3, CAM.H, CAM.W, generator=gen).to(DEV), torch.randn(1, CAM.H, CAM.W, generator=gen).to(DEV))

    def sequential():
        tr = IcpTracker(Args())
        tr.update_curr_status(depths[0], K)
        tr.move_last_status()
        out = []
        for i in range(1, len(depths)):
            tr.update_curr_status(depths[i], K)
            pose, ok = tr.predict_pose({"K": K, "frame_id": i})
            tr.move_last_status()
            img, gd = _map_step(g, s, grads)
            out.append((pose.copy(), ok, img.clone(), {k: v.clone() for k, v in gd.items()}))
        return out

    def pipelined():
        import contextlib
        tr = IcpTracker(Args())
        pipe = TrackMapPipeline(DEV, reserve_cus=reserve_cus)
        assert (pipe.mapper_stream is not None) == (reserve_cus > 0)
        tr.update_curr_status(depths[0], K)
        tr.move_last_status()
        out = []
        main = torch.cuda.current_stream()
        ctx = contextlib.nullcontext()
        if pipe.mapper_stream is not None:
            pipe.mapper_stream.wait_stream(main)
            ctx = torch.cuda.stream(pipe.mapper_stream)
        try:
          with ctx:
            for i in range(1, len(depths)):
                # the frame's depth is scaled IN PLACE on the main stream right before track(): the tracker stage must be
                # ordered after it (it sees the restored values or the poses differ)
                depths[i].mul_(2.0).mul_(0.5)
                def stage(i=i):
                    tr.update_curr_status(depths[i], K)
                    res = tr.predict_pose({"K": K, "frame_id": i})
                    tr.move_last_status()
                    return res
                pipe.track(stage)
                img, gd = _map_step(g, s, grads)            # main stream, concurrently with the tracker stage
                pose, ok = pipe.result()
                out.append((pose.copy(), ok, img.clone(), {k: v.clone() for k, v in gd.items()}))
        finally:
            if pipe.mapper_stream is not None:
                main.wait_stream(pipe.mapper_stream)
            torch.cuda.synchronize()
            pipe.close()
        return out

    a, b = sequential(), pipelined()
    torch.cuda.synchronize()
    assert len(a) == len(b) == 4
    for (pa, oka, ia, ga), (pb, okb, ib, gb) in zip(a, b):
        assert np.array_equal(pa, pb) and oka == okb
        assert torch.equal(ia, ib)
        for k in ru.FIELDS:
            # the gradient slots of a Gaussian are taken in arrival order of its tiles (one integer atomic each): their sum
            # is order-dependent in the last bits from run to run, pipelined or not
            assert float((ga[k] - gb[k]).abs().max()) <= 1e-5 * float(ga[k].abs().max()), k
    assert any(float(np.abs(p[0] - np.eye(4)).max()) > 1e-4 for p in a)        # the stream really moves


def test_two_frames_in_flight_come_back_in_order_and_see_the_callers_earlier_work():
    from rtg_slam_amd.pipeline import TrackMapPipeline
    pipe = TrackMapPipeline(DEV)
    try:
        x = torch.zeros(1 << 22, device=DEV)
        seen = []
        for step in range(6):
            x.add_(1.0)                                          # main stream: "the map update of frame `step`"
            pipe.track(lambda step=step: (step, float(x.sum().item()) / x.numel()))
            if step >= 1:                                        # two stages outstanding from here on
                seen.append(pipe.result())
        seen.append(pipe.result())
        assert [s[0] for s in seen] == list(range(6))
        # stage `step` was ordered after the add of its own frame; later adds may or may not have landed, earlier ones must
        for step, val in seen:
            assert val >= step + 1, (step, val)
        # after result() the main stream waits for the tracker stream: a tracker-side write is visible to main-stream work
        y = torch.zeros(1 << 22, device=DEV)
        def writer():
            for _ in range(50):
                y.add_(1.0)
            return None
        pipe.track(writer)
        pipe.result()
        z = y.clone()                                            # main stream, no host synchronisation in between
        torch.cuda.synchronize()
        assert float(z.min()) == 50.0 and float(z.max()) == 50.0
        with pytest.raises(RuntimeError):
            pipe.result()                                        # nothing outstanding
    finally:
        pipe.close()


def test_an_exception_of_the_tracker_stage_surfaces_in_result():
    from rtg_slam_amd.icp import IcpTracker
    from rtg_slam_amd.pipeline import TrackMapPipeline
    pipe = TrackMapPipeline(DEV)
    try:
        tr = IcpTracker(Args())
        K = torch.tensor([[CAM.fx, 0, CAM.cx], [0, CAM.fy, CAM.cy], [0, 0, 1]], dtype=torch.float32, device=DEV)
        empty = torch.zeros(CAM.H, CAM.W, 1, device=DEV)         # no depth at all: the normal equations are singular
        tr.update_curr_status(empty, K)
        tr.move_last_status()
        def stage():
            tr.update_curr_status(empty, K)
            return tr.predict_pose({"K": K, "frame_id": 1})
        pipe.track(stage)
        pipe.track(lambda: "next frame")
        with pytest.raises(Exception) as ei:
            pipe.result()
        assert "singular" in str(ei.value).lower() or "empty" in str(ei.value).lower() or "icp" in str(ei.value).lower()
        assert pipe.result() == "next frame"                     # the pipeline survives the failed frame
    finally:
        pipe.close()
