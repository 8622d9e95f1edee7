"""Differential fuzzing of rtg_slam_amd.mapping.Mapping (CPU doubles) against the reference's OWN Mapping (run on the CPU in
place, oracle/ref_mapper_shim.py): random argument sets and streams, both lifecycles on the same inputs and random streams,
sizes compared after every frame, every raw tensor (in a canonical row order) at the end.

    python -m oracle.fuzz_mapping_vs_reference [n_cases] [first_seed]        (build container only; ~1 minute per case)
    python -m oracle.fuzz_mapping_vs_reference long                           (the 20-frame case with stable deletions)

TEST INFRASTRUCTURE: a search for lifecycle branches the fixed golden streams do not reach.  What it finds becomes a golden
stream in oracle/gen_mapping_golden.py."""
from __future__ import annotations

import os
import random
import sys
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def case(seed):
    r = random.Random(seed)
    over = dict(uniform_sample_num=r.choice([90, 180, 260, 400]), gaussian_update_iter=r.choice([2, 4, 6]),
                gaussian_update_frame=r.choice([1, 2, 3]), memory_length=r.choice([2, 3, 4]),
                stable_confidence_thres=float(r.choice([1, 2, 3, 5])), unstable_time_window=r.choice([2, 3, 5]),
                keyframe_trans_thes=r.choice([0.008, 0.015, 0.03]), keyframe_theta_thes=r.choice([0.5, 2.0, 30.0]),
                add_depth_thres=r.choice([0.05, 0.1]), add_color_thres=r.choice([0.05, 0.1, 0.2]),
                add_transmission_thres=r.choice([0.3, 0.5]), error_sample_ratio=r.choice([0.05, 0.2]),
                transmission_sample_ratio=r.choice([0.5, 1.0]), global_keyframe_num=r.choice([1, 3]),
                final_global_iter=r.choice([1, 2]), history_merge_max_weight=r.choice([0.3, 0.5]), max_depth=8.0, seed=3,
                xyz_factor=r.choice([[1.0, 1.0, 0.1], [1.0, 1.0, 0.1], [1.0, 1.0, 1.0]]), init_opacity=r.choice([0.99, 0.8]))
    if r.random() < 0.3:                                         # other rates and loss weights than the configuration files'
        over.update(opacity_lr=r.choice([0.0, 0.01]), position_lr=r.choice([0.0005, 0.001, 0.002]),
                    rotation_lr=r.choice([0.0005, 0.001]), color_weight=r.choice([0.5, 0.8, 1.0]),
                    depth_weight=r.choice([0.0, 0.5, 1.0]), ssim_weight=r.choice([0.0, 0.2]))
    if r.random() < 0.25:
        over["normal_weight"] = r.choice([0.05, 0.2])     # the normal term of the loss (0.0 in every configuration file)
    if r.random() < 0.25:
        over["type"] = "Scannetpp"          # local AND (on a keyframe) global optimisation per optimised frame; loss without depth-less pixels
    cam = r.choice([(48, 64, 40.0, 31.5, 23.5)] * 2 + [(50, 70, 44.0, 34.5, 24.5), (41, 57, 36.0, 28.0, 20.0)])   # H W f cx cy
    return dict(over=over, tum=r.random() < 0.4, changing=r.random() < 0.5, n_frames=r.choice([6, 8, 10]),
                stream_seed=r.choice([4, 5, 9, 12]), rng_seed=100 + seed, cam=cam)


def long_case():
    """Twenty frames of the changing scene with a sparse optimisation schedule: the one fixed case that reaches the DELETION of
    stable Gaussians by depth-error strikes (mapper.py:560-575; 19 rows in two batches) besides thousands of releases."""
    return dict(over=dict(uniform_sample_num=260, gaussian_update_iter=2, gaussian_update_frame=3, memory_length=2,
                          stable_confidence_thres=1.0, unstable_time_window=3, keyframe_trans_thes=0.05, keyframe_theta_thes=30.0,
                          max_depth=8.0, seed=3, final_global_iter=1),
                tum=False, changing=True, n_frames=20, stream_seed=4, rng_seed=321)


def snapshot_ref(m):
    g = lambda t: t.detach().cpu().clone()
    out = {}
    for tag, pc in (("u", m.pointcloud), ("s", m.stable_pointcloud)):
        out[tag] = dict(xyz=g(pc._xyz), f_dc=g(pc._features_dc), f_rest=g(pc._features_rest), opacity=g(pc._opacity),
                        scaling=g(pc._scaling), rotation=g(pc._rotation), confidence=g(pc._confidence),
                        add_tick=g(pc._add_tick).float(), depth_error_counter=g(pc._depth_error_counter).float(),
                        color_error_counter=g(pc._color_error_counter).float())
    return out


def snapshot_mine(m):
    o = m.opt
    P = o.params[:o.N]
    out = {}
    for tag, r0, r1 in (("s", 0, o.n_frozen), ("u", o.n_frozen, o.N)):
        out[tag] = dict(xyz=P[r0:r1, 0:3].clone(), f_dc=P[r0:r1, 3:6].reshape(-1, 1, 3).clone(),
                        f_rest=P[r0:r1, 6:51].reshape(-1, 15, 3).clone(), opacity=P[r0:r1, 51:52].clone(),
                        scaling=P[r0:r1, 52:55].clone(), rotation=P[r0:r1, 55:59].clone(),
                        confidence=o.aux["confidence"][r0:r1].clone().float(), add_tick=o.aux["add_tick"][r0:r1].clone().float(),
                        depth_error_counter=o.aux["depth_error_counter"][r0:r1].clone().float(),
                        color_error_counter=o.aux["color_error_counter"][r0:r1].clone().float())
    return out


def force_state(M, ref_state):
    """Teacher forcing: this package's map takes the reference's state (same sizes), so that the next frame's decisions start
    from identical parameters - Adam (eps 1e-15) turns float-level gradient differences into learning-rate-sized parameter
    differences, which would otherwise flip later threshold decisions and hide real divergences behind noise."""
    o = M.opt
    nf, N = o.n_frozen, o.N
    for tag, r0, r1 in (("s", 0, nf), ("u", nf, N)):
        x = ref_state[tag]
        if r1 == r0:
            continue
        o.state["xyz"]["p"][r0:r1] = x["xyz"]
        o.state["shs"]["p"][r0:r1] = torch.cat([x["f_dc"].reshape(r1 - r0, 3), x["f_rest"].reshape(r1 - r0, 45)], dim=1)
        o.state["raw8"]["p"][r0:r1] = torch.cat([x["opacity"], x["scaling"], x["rotation"]], dim=1)
        o.aux["confidence"][r0:r1] = x["confidence"].to(o.aux["confidence"].dtype)
        o.aux["add_tick"][r0:r1] = x["add_tick"].to(o.aux["add_tick"].dtype)
        o.aux["depth_error_counter"][r0:r1] = x["depth_error_counter"].to(o.aux["depth_error_counter"].dtype)
        o.aux["color_error_counter"][r0:r1] = x["color_error_counter"].to(o.aux["color_error_counter"].dtype)
    o.version += 1
    o._act_valid = False
    M._render_cache = None


def compare(a, b, tol=3e-3):
    """Rows are matched by position (nearest xyz, which must be a bijection): the reference re-appends released rows, this
    package leaves them in place, so the row ORDER of the stable cloud may differ."""
    worst = 0.0
    for tag in ("s", "u"):
        x, y = a[tag], b[tag]
        if x["xyz"].shape[0] != y["xyz"].shape[0]:
            return f"{tag}: sizes {x['xyz'].shape[0]} vs {y['xyz'].shape[0]}", None
        n = x["xyz"].shape[0]
        if n == 0:
            continue
        d = torch.cdist(x["xyz"].double(), y["xyz"].double())
        match = d.argmin(dim=1)
        if len(set(match.tolist())) != n:
            return f"{tag}: rows cannot be matched one to one by position (nearest distances up to {float(d.min(1).values.max()):.3g})", None
        for k in x:
            rows = (x[k].reshape(n, -1) - y[k][match].reshape(n, -1)).abs().max(dim=1).values
            e = float(rows.max())
            if k in ("depth_error_counter", "color_error_counter"):
                # a strike is `mean error of the Gaussian's pixels > 2 x threshold` on renders of the map the frame has just
                # optimised: with parameters a learning rate apart, a Gaussian sitting on the threshold may count on one side only
                if e > 1 or float((rows > 0).float().mean()) > 0.02:
                    return f"{tag}.{k}: max difference {e} on {int((rows > 0).sum())} of {n} rows", worst
                continue
            worst = max(worst, e)
            if e > (0 if k in ("confidence", "add_tick") else tol):
                return f"{tag}.{k}: max difference {e}", worst
    return None, worst


def run_case(c, ref_mod):
    from tests import test_mapping_cpu as t
    from tests.mapping_doubles import TorchOps
    from rtg_slam_amd import mapping as mp
    from oracle.gen_mapping_golden import reference_args
    if "cam" in c:                                              # image sizes that are no multiples of the 16-pixel tile too
        from rtg_slam_amd import synth
        H, W, f, cx, cy = c["cam"]
        t.CAM = synth.CameraSpec(H, W, f, f, cx, cy)
    args = (mp.tum_args if c["tum"] else mp.replica_args)(**c["over"])
    stream = (t._changing_stream if c["changing"] else t._stream)(c["n_frames"], c["stream_seed"])
    inputs = []
    for fid, (d, col, c2w) in enumerate(stream):
        fr = mp.Frame(t.CAM, c2w, torch.device("cpu"), uid=fid)
        inputs.append((fid, d, col, c2w, t._frame_map(d, col, fr, args)))

    def seed_all():
        random.seed(c["rng_seed"]); np.random.seed(c["rng_seed"]); torch.manual_seed(c["rng_seed"])
    # the reference
    seed_all()
    rargs = reference_args(args)
    os.makedirs(rargs.save_path, exist_ok=True)
    R = ref_mod.Mapping(rargs)
    upd = SimpleNamespace(**vars(rargs))
    ref_sizes, ref_states, ref_rng = [], [], []
    for fid, d, col, c2w, fm in inputs:
        fr = mp.Frame(t.CAM, c2w, torch.device("cpu"), uid=fid)
        fr.original_image, fr.original_depth, fr.move_to_cpu_clone = col, d, (lambda f=fr: f)
        ref_rng.append((random.getstate(), torch.get_rng_state()))
        R.mapping(fr, {k: (v.clone() if torch.is_tensor(v) else v) for k, v in fm.items()}, fid, upd)
        R.get_render_output(fr)
        ref_sizes.append((R.get_unstable_num, R.get_stable_num))
        ref_states.append(snapshot_ref(R))
        R.time += 1
    ref_rng.append((random.getstate(), torch.get_rng_state()))
    R.global_optimization(upd, is_end=True)
    ref_final = snapshot_ref(R)
    # this package
    seed_all()
    ops = TorchOps(args)
    ops.gen = None
    M = mp.Mapping(args, torch.device("cpu"), ops=ops, capacity=900)
    M.rng = random
    worst_frame = 0.0
    for (fid, d, col, c2w, fm), want, state, rng in zip(inputs, ref_sizes, ref_states, ref_rng):
        fr = mp.Frame(t.CAM, c2w, torch.device("cpu"), uid=fid)
        random.setstate(rng[0]); torch.set_rng_state(rng[1])          # the random streams where the reference had them
        M.mapping(fr, {k: (v.clone() if torch.is_tensor(v) else v) for k, v in fm.items()}, fid)
        M.get_render_output(fr)
        got = (M.opt.N - M.opt.n_frozen, M.opt.n_frozen)
        if got != want:
            return f"frame {fid}: sizes (unstable, stable) {got} vs the reference's {want}", ref_sizes
        # float-level gradient differences become learning-rate-sized steps under Adam (eps 1e-15): up to one rate per iteration
        msg, w = compare(snapshot_mine(M), state, tol=max(3e-3, 2e-3 * args.gaussian_update_iter))
        if msg:
            return f"frame {fid}: {msg}", ref_sizes
        worst_frame = max(worst_frame, w)
        force_state(M, state)
        M.time += 1
    random.setstate(ref_rng[-1][0]); torch.set_rng_state(ref_rng[-1][1])
    if M.keyframe_ids != R.keyframe_ids or M.optimize_frames_ids != R.optimize_frames_ids:
        return f"keyframes {M.keyframe_ids} vs {R.keyframe_ids}; optimised {M.optimize_frames_ids} vs {R.optimize_frames_ids}", ref_sizes
    M.global_optimization(select_keyframe_num=-1, is_end=True)
    msg, worst = compare(snapshot_mine(M), ref_final, tol=max(3e-3, 2e-3 * len(M.keyframe_list) * args.final_global_iter *
                                 max(1.0, args.feature_lr_coef, args.scaling_lr_coef, args.rotation_lr_coef)))
    return (f"final: {msg}", ref_sizes) if msg else (None, (ref_sizes, max(worst, worst_frame), dict(M.stats)))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "long":
        cases = [("long", long_case())]
    else:
        n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
        first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
        cases = [(seed, case(seed)) for seed in range(first, first + n)]
    import io
    import contextlib
    from oracle import ref_mapper_shim as rm
    ref_mod = rm.install()
    bad = 0
    for seed, c in cases:
        sink = io.StringIO()
        try:
            with contextlib.redirect_stdout(sink), contextlib.redirect_stderr(sink):
                msg, info = run_case(c, ref_mod)
        except Exception as e:                                   # a crash on either side is a finding too
            msg, info = f"EXCEPTION {type(e).__name__}: {e}", None
        if msg:
            bad += 1
            print(f"seed {seed}: MISMATCH - {msg}\n    case {c}\n    reference sizes {info}", flush=True)
        else:
            sizes, worst, stats = info
            print(f"seed {seed}: ok  frames {c['n_frames']} tum {c['tum']} changing {c['changing']}  final sizes {sizes[-1]}  "
                  f"largest difference {worst:.2e}  fixed {stats['fixed']} deleted {stats['deleted_unstable']}+{stats['deleted_stable']} "
                  f"released {stats['released']} global {stats['global_opts']}", flush=True)
    print(f"{bad} mismatching case(s) of {len(cases)}")


if __name__ == "__main__":
    main()
