"""tests/golden/mapping_ref.npz: the reference's OWN Mapping (SLAM/multiprocess/mapper.py, run on the CPU from
/root/reference through oracle/ref_mapper_shim.py) over the synthetic stream of tests/test_mapping_cpu.py - per frame the
sizes and the raw tensors of both clouds, the optimised frames and the keyframes - for
tests/test_mapping_cpu.py::test_lifecycle_matches_the_references_own_mapping to hold rtg_slam_amd.mapping.Mapping against.

    python -m oracle.gen_mapping_golden            (build container only; ~1 minute)

Both sides get the same rasterizer (oracle/raster_oracle.py), the same k-NN and error accumulation (slam_ops_oracle) and the
same random streams (python `random`, torch's default generator), so the comparison is one of host logic."""
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

N_FRAMES = 7
SEED = 11


def reference_args(a):
    """The reference reads a few names my argument set does not carry (paths, logging, process mode)."""
    d = dict(vars(a))
    d.update(save_path="/tmp/rtgs_ref_mapping", save_step=10 ** 9, verbose=False, mode="single process", use_tensorboard=False,
             device_list=[0], parent="")
    return SimpleNamespace(**d)


N_FRAMES_CHANGING = 15


def stream_inputs(changing=False, tum=False, scannetpp=False, normal=False):
    from tests import test_mapping_cpu as t
    from rtg_slam_amd import mapping as mp
    args = t._args_tum() if tum else (t._args_scannetpp() if scannetpp else t._args(**({"normal_weight": 0.2} if normal else {})))
    frames = []
    for fid, (d, c, c2w) in enumerate(t._changing_stream(N_FRAMES_CHANGING) if changing else t._stream(N_FRAMES)):
        fr = mp.Frame(t.CAM, c2w, torch.device("cpu"), uid=fid)
        fm = t._frame_map(d, c, fr, args)
        frames.append((fid, fr, fm, d, c))
    return args, frames


def snapshot(pc):
    g = lambda t: t.detach().cpu().numpy().copy()
    return dict(xyz=g(pc._xyz), f_dc=g(pc._features_dc), f_rest=g(pc._features_rest), opacity=g(pc._opacity),
                scaling=g(pc._scaling), rotation=g(pc._rotation), confidence=g(pc._confidence), add_tick=g(pc._add_tick),
                depth_error_counter=g(pc._depth_error_counter), color_error_counter=g(pc._color_error_counter))


def run_reference(changing=False, tum=False, scannetpp=False, normal=False):
    from oracle import ref_mapper_shim as rm
    ref = rm.install()
    args, frames = stream_inputs(changing, tum, scannetpp, normal)
    rargs = reference_args(args)
    os.makedirs(rargs.save_path, exist_ok=True)
    random.seed(SEED)
    np.random.seed(SEED)
    torch.manual_seed(SEED)
    m = ref.Mapping(rargs)
    upd = SimpleNamespace(**vars(rargs))
    out = {}
    for fid, fr, fm, d, c in frames:
        fr.original_image = c
        fr.original_depth = d
        fr.move_to_cpu_clone = (lambda f=fr: f)
        fmap = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in fm.items()}
        out[f"f{fid}_rng_py"] = np.array(random.getstate()[1], dtype=np.int64)      # the random streams at the frame's start
        out[f"f{fid}_rng_torch"] = torch.get_rng_state().numpy().copy()
        m.mapping(fr, fmap, fid, upd)
        m.get_render_output(fr)
        for tag, pc in (("u", m.pointcloud), ("s", m.stable_pointcloud)):
            for k, v in snapshot(pc).items():
                out[f"f{fid}_{tag}_{k}"] = v
        out[f"f{fid}_sizes"] = np.array([m.get_unstable_num, m.get_stable_num], dtype=np.int64)
        m.time += 1
    m.global_optimization(upd, is_end=True)                     # slam.py:129: everything becomes stable, all keyframes
    for tag, pc in (("u", m.pointcloud), ("s", m.stable_pointcloud)):
        for k, v in snapshot(pc).items():
            out[f"final_{tag}_{k}"] = v
    out["final_sizes"] = np.array([m.get_unstable_num, m.get_stable_num], dtype=np.int64)
    out["optimize_frames_ids"] = np.array(m.optimize_frames_ids, dtype=np.int64)
    out["keyframe_ids"] = np.array(m.keyframe_ids, dtype=np.int64)
    out["n_frames"] = np.array([len(frames)], dtype=np.int64)
    out["seed"] = np.array([SEED], dtype=np.int64)
    return out


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "both"
    for changing, tum, sc, nrm, name in ((False, False, False, False, "mapping_ref.npz"),
                                         (True, False, False, False, "mapping_ref_changing.npz"),
                                         (False, True, False, False, "mapping_ref_tum.npz"),
                                         (False, False, True, False, "mapping_ref_scannetpp.npz"),
                                         (False, False, False, True, "mapping_ref_normal.npz")):
        kind = "normal" if nrm else ("scannetpp" if sc else ("tum" if tum else ("changing" if changing else "static")))
        if which not in ("both", kind):
            continue
        out = run_reference(changing, tum, sc, nrm)
        path = os.path.join(ROOT, "tests", "golden", name)
        np.savez_compressed(path, **out)
        n = int(out["n_frames"][0])
        print("sizes per frame (unstable, stable):", [tuple(int(x) for x in out[f"f{i}_sizes"]) for i in range(n)], "final",
              tuple(int(x) for x in out["final_sizes"]))
        print("optimised frames", out["optimize_frames_ids"], "keyframes", out["keyframe_ids"])
        print(f"{os.path.getsize(path) / 1024:.0f} KB -> {path}")


if __name__ == "__main__":
    main()
