"""On-disk formats: the model PLY written here has the reference's header and column order
(gaussian_pointcloud.py:407-466) and survives a round trip; `_sibr` files (no confidence column) load with zeros;
merge, trajectories and performance.json follow the reference's writers."""
import json
import os

import numpy as np
import torch

from rtg_slam_amd import io_formats as io


def _model(n, seed=0):
    g = torch.Generator().manual_seed(seed)
    return dict(xyz=torch.randn(n, 3, generator=g), features_dc=torch.randn(n, 1, 3, generator=g),
                features_rest=torch.randn(n, 15, 3, generator=g), opacity=torch.randn(n, 1, generator=g),
                scaling=torch.randn(n, 3, generator=g), rotation=torch.randn(n, 4, generator=g),
                confidence=torch.randint(0, 50, (n, 1), generator=g).float())


def test_model_ply_header_and_round_trip(tmp_path):
    m = _model(257)
    p = str(tmp_path / "save_model" / "frame_0010" / "iter_0050.ply")
    io.save_model_ply(p, **m)
    raw = open(p, "rb").read()
    head = raw[:raw.index(b"end_header\n") + 11].decode().split("\n")
    assert head[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 257"]
    props = [l.split()[-1] for l in head if l.startswith("property")]
    assert props == (["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(45)]
                     + ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3", "confidence"])
    assert all(l.startswith("property float ") for l in head if l.startswith("property"))
    assert len(raw) == len("\n".join(head).encode()) + 257 * 63 * 4
    # f_rest is stored channel-major: f_rest_0..14 = channel 0 of coefficients 1..15 (transpose(1, 2).flatten())
    table = np.frombuffer(raw[-257 * 63 * 4:], dtype="<f4").reshape(257, 63)
    assert np.array_equal(table[:, 9:24], m["features_rest"][:, :, 0].numpy())
    assert np.array_equal(table[:, 3:6], np.zeros((257, 3), np.float32))            # normals column block is zeros
    back = io.load_model_ply(p)
    for k, v in m.items():
        assert np.array_equal(back[k], v.numpy()), k
    # packed [N,59] layout of the optimiser <-> model arrays
    packed = io.model_to_packed(back)
    again = io.packed_to_model(packed)
    for k in ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation"):
        assert np.array_equal(again[k], back[k]), k


def test_sibr_and_merge(tmp_path):
    a, b = _model(10, 1), _model(7, 2)
    pa, pb = str(tmp_path / "iter_0001_sibr.ply"), str(tmp_path / "iter_0001_stable_sibr.ply")
    io.save_model_ply(pa, **a, include_confidence=False)
    io.save_model_ply(pb, **b, include_confidence=False)
    la = io.load_model_ply(pa)
    assert float(np.abs(la["confidence"]).max()) == 0 and np.array_equal(la["xyz"], a["xyz"].numpy())
    pm = str(tmp_path / "iter_0001_merge_sibr.ply")
    io.merge_ply(pa, pb, pm, include_confidence=False)
    lm = io.load_model_ply(pm)
    assert lm["xyz"].shape[0] == 17 and np.array_equal(lm["xyz"][10:], b["xyz"].numpy())
    io.save_model_ply(str(tmp_path / "empty.ply"), **_model(0))
    assert not os.path.exists(str(tmp_path / "empty.ply"))                              # empty cloud: nothing written


def test_trajectories_and_performance_json(tmp_path):
    poses = [np.eye(4) for _ in range(5)]
    io.save_trajectories(str(tmp_path), poses, poses)
    assert np.load(str(tmp_path / "save_traj" / "pose_es.npy")).shape == (5, 4, 4)
    assert np.load(str(tmp_path / "save_traj" / "pose_gt.npy")).shape == (5, 4, 4)
    r = io.Recorder(0)
    r.update_mean("tracking", 0.5, 5)
    r.update_mean("mapping", 1.0, 5)
    r.update_mean("mapping", 0.5, 5)
    r.update_max("gpu_memory", 1234.0)
    r.cal_fps()
    r.save(str(tmp_path))
    d = json.load(open(str(tmp_path / "performance.json")))
    assert abs(d["mapping"] - 0.15) < 1e-12 and abs(d["fps"] - 1 / 0.15) < 1e-9 and d["gpu_memory"] == 1234.0
    assert abs(d["tracking"] - 0.1) < 1e-12


def test_model_ply_bytes_equal_the_reference_writers_recipe(tmp_path):
    """Byte for byte against the reference writer's own recipe (gaussian_pointcloud.py:424-466), re-enacted here without
    `plyfile` (absent from this image): the attribute table is concatenated in its order (xyz, zero normals,
    f_dc.transpose(1,2).flatten, f_rest.transpose(1,2).flatten, opacity, scaling, rotation, confidence), copied into a
    structured array of ('name', 'f4') fields by `elements[:] = list(map(tuple, attributes))`, and serialised the way
    plyfile's PlyData([PlyElement.describe(elements, 'vertex')]).write does for a native little-endian array: the
    header lines `ply / format binary_little_endian 1.0 / element vertex N / property float <name> ... / end_header`
    followed by elements.tobytes()."""
    m = _model(33, seed=4)
    xyz = m["xyz"].numpy()
    normals = np.zeros_like(xyz)
    f_dc = m["features_dc"].transpose(1, 2).flatten(start_dim=1).contiguous().numpy()
    f_rest = m["features_rest"].transpose(1, 2).flatten(start_dim=1).contiguous().numpy()
    for include_confidence in (True, False):
        names = ["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(45)]
        names += ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)]
        cols = [xyz, normals, f_dc, f_rest, m["opacity"].numpy(), m["scaling"].numpy(), m["rotation"].numpy()]
        if include_confidence:
            names.append("confidence")
            cols.append(m["confidence"].numpy())
        attributes = np.concatenate(cols, axis=1)
        elements = np.empty(xyz.shape[0], dtype=[(n, "f4") for n in names])
        elements[:] = list(map(tuple, attributes))
        header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % xyz.shape[0]
        header += "".join(f"property float {n}\n" for n in names) + "end_header\n"
        want = header.encode("ascii") + elements.tobytes()
        p = str(tmp_path / f"m{int(include_confidence)}.ply")
        io.save_model_ply(p, **m, include_confidence=include_confidence)
        assert open(p, "rb").read() == want


def test_model_ply_bytes_equal_what_the_references_own_writer_produces(tmp_path):
    """tests/golden/model_ply_ref.npz: the file the reference's OWN GaussianPointCloud.save_model_ply writes for the seeded
    model below (gaussian_pointcloud.py:424-466 run on the CPU in place by oracle/gen_ply_golden.py; only plyfile's
    serialisation of the finished structured array is a stand-in) - byte for byte, with and without the confidence column."""
    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "model_ply_ref.npz"))
    m = _model(int(ref["n"][0]), seed=int(ref["seed"][0]))
    for inc in (True, False):
        p = str(tmp_path / f"m{int(inc)}.ply")
        io.save_model_ply(p, **m, include_confidence=inc)
        assert open(p, "rb").read() == ref[f"bytes_conf{int(inc)}"].tobytes()
