"""Parity of the HIP ICP tracker with the reference's own outputs (tests/golden/icp_*.npz) and,
on larger seeded frames, with the pinned oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import icp_oracle as io
from rtg_slam_amd import synth
from tests import margins

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, f"icp_{name}.npz"))
    return {k: (torch.from_numpy(z[k]) if z[k].ndim else z[k]) for k in z.files}


@pytest.mark.parametrize("name", ["small_clean", "small_noisy"])
def test_pyramids_vs_reference(golden_dir, name):
    from rtg_slam_amd import icp
    g = load(golden_dir, name)
    for tag in ("0", "1"):
        vp, npyr = icp.build_pyramids(g[f"depth{tag}"].to(DEV), g["K"].to(DEV), 3)
        for l in range(3):
            assert torch.equal(vp[l].cpu(), g[f"v{tag}_{l}"]), "vertex maps bit-exact"
            # Sobel taps in conv2d's order, cross and norm rounded as torch rounds them: the normal maps are the
            # reference's bit for bit (so no gate of the tracker can flip on a last-ulp normal difference)
            d = (npyr[l].cpu() - g[f"n{tag}_{l}"]).abs().amax(dim=-1)
            assert torch.equal(npyr[l].cpu(), g[f"n{tag}_{l}"]), (l, float(d.max()), float((d > 0).float().mean()))


@pytest.mark.parametrize("name", ["small_clean", "small_noisy"])
def test_icp_step_vs_reference(golden_dir, name):
    from rtg_slam_amd import icp
    g = load(golden_dir, name)
    cos_thr = float(np.cos(np.deg2rad(20.0)))
    for l, ds in enumerate([0.25, 0.5, 1.0]):
        Kl = g["K"] * ds
        Kl[2, 2] = 1.0
        JtJ, Jtr, nv = icp.icp_step(g[f"v1_{l}"].to(DEV), g[f"n1_{l}"].to(DEV), g[f"v0_{l}"].to(DEV),
                                    g[f"n0_{l}"].to(DEV), Kl, g["pose_probe"], 0.1, cos_thr)
        assert int(nv.item()) == int(g[f"nvalid_{l}"])                 # identical valid count
        ref = g[f"JtJ_{l}"]
        assert float((JtJ.cpu() - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
        rj = g[f"Jtr_{l}"].reshape(-1)
        assert float((Jtr.cpu() - rj).abs().max()) <= 1e-4 * float(rj.abs().max()) + 1e-6


@pytest.mark.parametrize("persistent", [False, True])
@pytest.mark.parametrize("name", ["small_clean", "small_noisy"])
def test_track_vs_reference(golden_dir, name, persistent):
    """Both forms of the track (one launch per iteration / one persistent kernel) against the reference's own pose."""
    from rtg_slam_amd import icp
    g = load(golden_dir, name)
    mk = lambda p: [g[f"{p}_{l}"].to(DEV) for l in range(3)]
    out = icp.icp_track(mk("v1"), mk("n1"), mk("v0"), mk("n0"), g["K"], [0.25, 0.5, 1.0], [5, 5, 5], 0.1,
                        float(np.cos(np.deg2rad(20.0))), 1e-4, persistent=persistent).cpu()
    assert float(out[19]) == 0
    pose = out[:16].reshape(4, 4)
    assert float((pose - g["pose_final"]).abs().max()) < 1e-5
    assert abs(float(out[16]) - float(g["valid_ratio"])) < 1e-4
    assert abs(float(out[17]) - float(g["p2p_loss"])) <= 1e-4 * max(1.0, float(g["p2p_loss"]))
    assert float(out[18]) == 0


@pytest.mark.parametrize("name", ["small_clean", "small_noisy"])
def test_fill_vs_reference(golden_dir, name):
    from rtg_slam_amd import icp
    g = load(golden_dir, name)
    rd = g["fill_in"].to(DEV).clone()
    icp.fill_model_depth(rd, g["depth1"].to(DEV), g["fill_rn"].to(DEV), g["n1_2"].to(DEV), 0.01, 0.01)
    assert torch.equal(rd.cpu(), g["fill_out"])


class Args:
    icp_downscales = [0.25, 0.5, 1.0]
    icp_downscale_iters = [5, 5, 5]
    icp_warmup_frames = 0
    icp_use_model_depth = True
    icp_distance_threshold = 0.1
    icp_normal_threshold = 20
    icp_damping = 1e-4
    icp_sample_distance_threshold = 0.01
    icp_sample_normal_threshold = 0.01
    icp_fail_threshold = 0.02
    verbose = False


@pytest.mark.parametrize("cam,noise,persistent", [(synth.TUM_FR1, True, False), (synth.REPLICA, False, False),
                                                  (synth.TUM_FR1, True, True), (synth.REPLICA, False, True)])
def test_tracker_class_vs_oracle_full_size(cam, noise, persistent):
    """IcpTracker API (SLAM/icp.py:357-452) on Replica / TUM shaped frames vs the pinned oracle."""
    from rtg_slam_amd.icp import IcpTracker
    poses = synth.trajectory(2, seed=9)
    base = synth.look_at_pose(seed=3, max_angle_deg=5, max_trans=0.3)
    d0 = synth.box_room_depth(cam, base @ poses[0])
    d1 = synth.box_room_depth(cam, base @ poses[1])
    if noise:
        d0, d1 = synth.tum_noise(d0, 1), synth.tum_noise(d1, 2)
    K = torch.tensor([[cam.fx, 0, cam.cx], [0, cam.fy, cam.cy], [0, 0, 1]], dtype=torch.float32)
    args = Args()
    args.icp_persistent = persistent
    tr = IcpTracker(args)
    tr.update_curr_status(d0.to(DEV), K.to(DEV))
    tr.move_last_status()
    tr.update_curr_status(d1.to(DEV), K.to(DEV))
    pose, ok = tr.predict_pose({"K": K.to(DEV), "frame_id": 1})
    vp0 = io.vertex_pyramid(d0, K.clone(), 3); np0 = io.normal_pyramid(vp0)
    vp1 = io.vertex_pyramid(d1, K.clone(), 3); np1 = io.normal_pyramid(vp1)
    pose_o, ratio_o, loss_o = io.track(vp1, np1, vp0, np0, K.clone())
    pose_x, _, _ = io.track(vp1, np1, vp0, np0, K.clone(), exact_sums=True)
    assert pose.shape == (4, 4) and pose.dtype == np.float32
    # Normals are bit-identical to the reference's and no gate flips (test_no_gate_flips_...), so the only difference
    # left is HOW the 27 sums are reduced.  The kernel reduces in float64; against the reference algorithm with exact
    # (float64) sums it holds 2e-5 on both frames.  The reference's own float32 reductions are the noisier side: on
    # the ill-conditioned noisy frame (5 % of the pixels pass the gates, the iteration is still wandering after 15
    # steps) they move its 15-iteration pose by ~1e-4: the SAME torch-CPU oracle gives t_x = -0.0279093 in the build
    # container and -0.0280408 on the MI355X box's host (another BLAS reduction order), the exact-sum variant
    # -0.0278665 on both, the kernel -0.0278793.  So: 3e-5 against the exact-sum oracle on both frames, and against
    # the float32 oracle no further than that oracle's own distance from exact sums.
    err_x = float(np.abs(pose - pose_x.numpy()).max())
    err_o = float(np.abs(pose - pose_o.numpy()).max())
    ref_noise = float((pose_o - pose_x).abs().max())
    print(f"noise={noise}: |hip - exact-sum oracle| = {err_x:.2e}, |hip - f32 oracle| = {err_o:.2e}, "
          f"|f32 oracle - exact-sum oracle| = {ref_noise:.2e}")
    assert err_x < 3e-5, (err_x, err_o, ref_noise)
    assert err_o < 2e-5 + 1.5 * ref_noise, (err_x, err_o, ref_noise)
    assert abs(tr.last_valid_ratio - ratio_o) < 1e-3
    assert ok == (not (loss_o > 0.02))
    if not noise:
        rel = (torch.linalg.inv(poses[0]) @ poses[1]).float().numpy()
        assert float(np.abs(pose - rel).max()) < 5e-3


def full_tol(g):
    """Tolerance of the 15-iteration pose against the reference's float32 output, DERIVED from the golden file:
    north_star's 1e-5, unless the reference itself is not defined that sharply on the frame - its own answer moves by
    `pose_sensitivity` when ONE entry of its initial pose moves by +-1e-7 (six runs of the reference's code,
    oracle/gen_icp_golden.py); the smallest of those six moves is the bound then.  Clean Replica-shaped frame: the
    reference moves by 4e-8..1.4e-7 -> 1e-5.  Noisy TUM-shaped frame: 3.7e-5..2.9e-4 -> 3.7e-5."""
    return max(1e-5, float(np.min(np.asarray(g["pose_sensitivity"]))))


@pytest.mark.parametrize("name,cam,noise", [("full_replica_clean", synth.REPLICA, False), ("full_tum_noisy", synth.TUM_FR1, True)])
def test_full_size_vs_reference_outputs(golden_dir, name, cam, noise):
    """Full-size frames against outputs of the REFERENCE's own tracker (oracle/gen_icp_golden.py::full_size_cases runs
    /root/reference/SLAM/icp.py on these frames; only its outputs are committed, the frames regenerate from the seeds).
    Pyramids: bit-identical (exact checksums - the sum of the float32 bit patterns - of all four maps per level).  Normal equations at a probe pose: the
    reference's valid count exactly, sums to 1e-4.  The 15-iteration pose: north_star's 1e-5 on the clean Replica-shaped
    frame; on the noisy TUM-shaped frame (5 % of the pixels pass the gates, the iteration is ill-conditioned) the kernel's
    float64 sums and the reference's float32 sums end 3e-5 apart - stated here and in DESIGN.md, not hidden behind
    another oracle.  (The reference against ITSELF with 1 vs 8 torch threads moves by 1e-8 on these frames.)"""
    from rtg_slam_amd import icp
    g = load(golden_dir, name)
    poses = synth.trajectory(2, seed=9)
    base = synth.look_at_pose(seed=3, max_angle_deg=5, max_trans=0.3)
    d0 = synth.box_room_depth(cam, base @ poses[0])
    d1 = synth.box_room_depth(cam, base @ poses[1])
    if noise:
        d0, d1 = synth.tum_noise(d0, 1), synth.tum_noise(d1, 2)
    bits = lambda t: int(t.detach().cpu().contiguous().view(torch.int32).to(torch.int64).sum())   # exact, order-independent
    assert bits(d0) == int(g["depth_checksum"][0]) and bits(d1) == int(g["depth_checksum"][1])
    K = g["K"]
    hv0, hn0 = icp.build_pyramids(d0.to(DEV), K.to(DEV), 3)
    hv1, hn1 = icp.build_pyramids(d1.to(DEV), K.to(DEV), 3)
    cos_thr = float(np.cos(np.deg2rad(20.0)))
    for l, ds in enumerate([0.25, 0.5, 1.0]):
        cs = [bits(t) for t in (hv1[l], hn1[l], hv0[l], hn0[l])]
        assert cs == [int(x) for x in g[f"pyr_checksum_{l}"]], (l, cs, g[f"pyr_checksum_{l}"])
        Kl = K * ds
        Kl[2, 2] = 1.0
        JtJ, Jtr, nv = icp.icp_step(hv1[l], hn1[l], hv0[l], hn0[l], Kl, g["pose_probe"], 0.1, cos_thr)
        assert int(nv.item()) == int(g[f"nvalid_{l}"])
        ref = g[f"JtJ_{l}"]
        assert float((JtJ.cpu() - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
        rj = g[f"Jtr_{l}"].reshape(-1)
        assert float((Jtr.cpu() - rj).abs().max()) <= 1e-4 * float(rj.abs().max()) + 1e-6
    for persistent in (False, True):
        out = icp.icp_track(hv1, hn1, hv0, hn0, K, [0.25, 0.5, 1.0], [5, 5, 5], 0.1, cos_thr, 1e-4, persistent=persistent).cpu()
        pose = out[:16].reshape(4, 4)
        err = float((pose - g["pose_final"]).abs().max())
        # the reference algorithm with its normal equations summed and solved in float64 (reference code on float64
        # tensors, per-pixel stage in float32 as always): the answer the kernel's float64 sums / Cholesky aim at
        err64 = float((pose - g["pose_final_f64solve"]).abs().max())
        ref64 = float((g["pose_final"] - g["pose_final_f64solve"]).abs().max())
        tol = full_tol(g)
        margins.record(f"persistent={persistent}", hip_vs_reference_f32=err, hip_vs_reference_f64solve=err64,
                       reference_f32_vs_reference_f64solve=ref64, tolerance_derived_from_reference_sensitivity=tol,
                       reference_moves_under_1e7_perturbation=[float(x) for x in g["pose_sensitivity"]],
                       reference_8_vs_1_threads=float((g['pose_final'] - g['pose_final_1thread']).abs().max()))
        print(f"{name} persistent={persistent}: |hip - reference pose| = {err:.2e}, |hip - f64-solve reference| = {err64:.2e}, "
              f"|reference - f64-solve reference| = {ref64:.2e}, tol {tol:.1e}")
        assert err < tol, (err, tol)
        # the kernel sits at least as close to the exactly-solved reference as the float32 reference does (1e-6: where
        # both distances are rounding noise, as on the clean frame)
        assert err64 < max(1e-6, ref64), (err64, ref64)
        assert abs(float(out[16]) - float(g["valid_ratio"])) < 1e-3
        assert abs(float(out[17]) - float(g["p2p_loss"])) <= 2e-4 * max(1.0, float(g["p2p_loss"]))


def test_no_gate_flips_on_the_noisy_full_size_frame():
    """bound the per-iteration gate flips on the noisy TUM-shaped frame.  Along the oracle's own
    15-iteration pose sequence, the HIP normal equations at every iterate have EXACTLY the oracle's number of valid
    correspondences (association, depth, distance and normal gates all agree) and match its sums to 1e-4 relative;
    the pyramids the two sides work on are bit-identical."""
    from rtg_slam_amd import icp
    cam = synth.TUM_FR1
    poses = synth.trajectory(2, seed=9)
    base = synth.look_at_pose(seed=3, max_angle_deg=5, max_trans=0.3)
    d0 = synth.tum_noise(synth.box_room_depth(cam, base @ poses[0]), 1)
    d1 = synth.tum_noise(synth.box_room_depth(cam, base @ poses[1]), 2)
    K = torch.tensor([[cam.fx, 0, cam.cx], [0, cam.fy, cam.cy], [0, 0, 1]], dtype=torch.float32)
    vp0 = io.vertex_pyramid(d0, K.clone(), 3); np0 = io.normal_pyramid(vp0)
    vp1 = io.vertex_pyramid(d1, K.clone(), 3); np1 = io.normal_pyramid(vp1)
    hv0, hn0 = icp.build_pyramids(d0.to(DEV), K.to(DEV), 3)
    hv1, hn1 = icp.build_pyramids(d1.to(DEV), K.to(DEV), 3)
    for l in range(3):
        for a, b in ((hv0[l], vp0[l]), (hn0[l], np0[l]), (hv1[l], vp1[l]), (hn1[l], np1[l])):
            assert torch.equal(a.cpu(), b), l
    cos_thr = float(np.cos(np.deg2rad(20.0)))
    pose = torch.eye(4)
    flips = 0
    for l, ds in enumerate([0.25, 0.5, 1.0]):
        Kl = K * ds
        Kl[2, 2] = 1.0
        for _ in range(5):
            res, J, valid = io.residuals_jacobian(vp1[l], vp0[l], np1[l], np0[l], pose, Kl, 0.1, cos_thr)
            JtJ_o, Jtr_o = io.normal_equations(J, res)
            JtJ, Jtr, nv = icp.icp_step(hv1[l], hn1[l], hv0[l], hn0[l], Kl, pose, 0.1, cos_thr)
            flips += abs(int(nv.item()) - int(valid.sum()))
            assert float((JtJ.cpu() - JtJ_o).abs().max()) <= 1e-4 * float(JtJ_o.abs().max())
            pose = io.gauss_newton_update(JtJ_o, Jtr_o, pose, 1e-4)
    assert flips == 0, flips


@pytest.mark.parametrize("persistent", [False, True])
def test_frame_without_correspondences_raises_like_the_reference(persistent):
    """An empty current frame (depth 0 everywhere): no pixel passes the gates, J^T J = 0 and the damped system
    (H += trace(H) * damping * I, icp.py:248-256) stays singular.  The reference raises from torch.inverse
    (icp.py:313-325, pinned by the oracle); the drop-in raises a RuntimeError too instead of returning a NaN pose."""
    from rtg_slam_amd.icp import IcpTracker
    cam = synth.CameraSpec(120, 160, 130.0, 130.0, 79.5, 59.5)
    base = synth.look_at_pose(seed=3, max_angle_deg=5, max_trans=0.3)
    d0 = synth.box_room_depth(cam, base)
    d1 = torch.zeros_like(d0)
    K = torch.tensor([[cam.fx, 0, cam.cx], [0, cam.fy, cam.cy], [0, 0, 1]], dtype=torch.float32)
    vp0 = io.vertex_pyramid(d0, K.clone(), 3); np0 = io.normal_pyramid(vp0)
    vp1 = io.vertex_pyramid(d1, K.clone(), 3); np1 = io.normal_pyramid(vp1)
    with pytest.raises(RuntimeError):
        io.track(vp1, np1, vp0, np0, K.clone())
    args = Args()
    args.icp_persistent = persistent
    tr = IcpTracker(args)
    tr.update_curr_status(d0.to(DEV), K.to(DEV))
    tr.move_last_status()
    tr.update_curr_status(d1.to(DEV), K.to(DEV))
    with pytest.raises(RuntimeError):
        tr.predict_pose({"K": K.to(DEV), "frame_id": 1})
    # and the tracker is still usable afterwards
    tr.update_curr_status(d0.to(DEV), K.to(DEV))
    pose, ok = tr.predict_pose({"K": K.to(DEV), "frame_id": 2})
    assert np.isfinite(pose).all()


@pytest.mark.parametrize("H,W", [(101, 135), (98, 130), (97, 128), (64, 66)])
def test_pyramids_on_sizes_that_are_not_multiples_of_four(H, W):
    """The one-pass three-level vertex kernel (a lane per 4x4 block) on ragged sizes: MaxPool2d floor mode drops the odd
    rows / columns at the coarse levels while the full-resolution level keeps every pixel - bit-identical to the pinned
    oracle (itself bit-identical to the reference's pyramids, tests/test_oracle_icp.py)."""
    from rtg_slam_amd import icp
    cam = synth.CameraSpec(H, W, 120.0, 118.0, (W - 1) / 2.0, (H - 1) / 2.0)
    base = synth.look_at_pose(seed=5, max_angle_deg=5, max_trans=0.3)
    d = synth.tum_noise(synth.box_room_depth(cam, base), 3)
    K = torch.tensor([[cam.fx, 0, cam.cx], [0, cam.fy, cam.cy], [0, 0, 1]], dtype=torch.float32)
    vp = io.vertex_pyramid(d, K.clone(), 3)
    npyr = io.normal_pyramid(vp)
    hv, hn = icp.build_pyramids(d.to(DEV), K.to(DEV), 3)
    for l in range(3):
        assert hv[l].shape == vp[l].shape
        assert torch.equal(hv[l].cpu(), vp[l]), l
        assert torch.equal(hn[l].cpu(), npyr[l]), l


@pytest.mark.parametrize("name,cam,noise", [("full_replica_clean", synth.REPLICA, False), ("full_tum_noisy", synth.TUM_FR1, True)])
def test_where_the_distance_to_the_reference_comes_from_iteration_by_iteration(golden_dir, name, cam, noise):
    """VERDICT r5 item 7.  The reference's pose after EACH of its 15 Gauss-Newton iterations (tests/golden/icp_full_*.npz::iter_poses,
    written by oracle/gen_icp_golden.py::add_iteration_poses from the reference's own static methods; its last entry is
    `pose_final` bit for bit) against the kernel stepped one iteration at a time, in both solve modes: float64 Cholesky (the
    product) and float32 in the reference's order of operations (RTGS_ICP_FLAG_F32_SOLVE).  What it settles: on the clean
    frame both modes stay at float32 rounding of the reference through all 15 iterations; on the noisy TUM-shaped frame the
    distance is born in single iterations where association gates flip (it jumps by an order of magnitude between two
    consecutive iterations, in BOTH modes) - it is not the solve's precision, and a float32 solve does not bring it under
    north_star's 1e-5.  The numbers are printed and recorded (profiles/r06_parity_margins.json)."""
    from rtg_slam_amd import icp
    g = load(golden_dir, name)
    assert "iter_poses" in g and tuple(g["iter_poses"].shape) == (15, 4, 4)
    assert torch.equal(g["iter_poses"][-1], g["pose_final"])
    poses = synth.trajectory(2, seed=9)
    base = synth.look_at_pose(seed=3, max_angle_deg=5, max_trans=0.3)
    d0 = synth.box_room_depth(cam, base @ poses[0])
    d1 = synth.box_room_depth(cam, base @ poses[1])
    if noise:
        d0, d1 = synth.tum_noise(d0, 1), synth.tum_noise(d1, 2)
    K = g["K"]
    hv0, hn0 = icp.build_pyramids(d0.to(DEV), K.to(DEV), 3)
    hv1, hn1 = icp.build_pyramids(d1.to(DEV), K.to(DEV), 3)
    cos_thr = float(np.cos(np.deg2rad(20.0)))
    ref = g["iter_poses"]
    curves = {}
    for mode in ("f64", "f32"):
        pose = torch.eye(4, device=DEV)
        errs = []
        for l, ds in enumerate([0.25, 0.5, 1.0]):
            for _ in range(5):
                out = icp.icp_track([hv1[l]], [hn1[l]], [hv0[l]], [hn0[l]], K, [ds], [1], 0.1, cos_thr, 1e-4, pose0=pose,
                                    f32_solve=(mode == "f32"))
                pose = out[:16].reshape(4, 4).clone()
                errs.append(float((pose.cpu() - ref[len(errs)]).abs().max()))
        curves[mode] = errs
        # stepping one iteration per call IS the 15-iteration chain (same launches, same arithmetic)
        whole = icp.icp_track(hv1, hn1, hv0, hn0, K, [0.25, 0.5, 1.0], [5, 5, 5], 0.1, cos_thr, 1e-4, f32_solve=(mode == "f32")).cpu()
        assert torch.equal(whole[:16].reshape(4, 4), pose.cpu())
    # teacher-forced: ONE iteration from the reference's own pose of the iteration before - the solve alone, no history
    forced = {"f64": [], "f32": []}
    for mode in ("f64", "f32"):
        for it in range(15):
            l = it // 5
            start = torch.eye(4) if it == 0 else ref[it - 1]
            out = icp.icp_track([hv1[l]], [hn1[l]], [hv0[l]], [hn0[l]], K, [[0.25, 0.5, 1.0][l]], [1], 0.1, cos_thr, 1e-4,
                                pose0=start.to(DEV), f32_solve=(mode == "f32")).cpu()
            forced[mode].append(float((out[:16].reshape(4, 4) - ref[it]).abs().max()))
    fmt = lambda v: " ".join(f"{x:.1e}" for x in v)
    print(f"{name}: |hip - reference| after each iteration, float64 solve: {fmt(curves['f64'])}")
    print(f"{name}: |hip - reference| after each iteration, float32 solve: {fmt(curves['f32'])}")
    print(f"{name}: ONE iteration from the reference's previous pose, float64 solve: {fmt(forced['f64'])}")
    print(f"{name}: ONE iteration from the reference's previous pose, float32 solve: {fmt(forced['f32'])}")
    margins.record("per_iteration", free_running_f64=curves["f64"], free_running_f32=curves["f32"],
                   one_step_from_reference_pose_f64=forced["f64"], one_step_from_reference_pose_f32=forced["f32"])
    tol = full_tol(g)
    assert curves["f64"][-1] < tol and curves["f32"][-1] < 4 * tol
    if not noise:
        assert max(curves["f64"]) < 1e-6 and max(curves["f32"]) < 1e-6
        assert max(forced["f64"]) < 1e-6 and max(forced["f32"]) < 1e-6
