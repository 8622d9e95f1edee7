"""On-disk formats of RTG-SLAM that sit either side of the hot path, readable / writable without
`plyfile` or `GPUtil`:

  * model snapshots  `iter_XXXX[_stable][_sibr|_merge].ply` - binary little-endian PLY, one `vertex` element, all
    float32, columns `x y z nx ny nz f_dc_0..2 f_rest_0..44 opacity scale_0..2 rot_0..3 [confidence]`, RAW
    (pre-activation) values, SH stored channel-major (`_features_dc.transpose(1, 2).flatten()`):
    SLAM/gaussian_pointcloud.py:407-466 (writer), :118-193 (reader), SLAM/utils.py:321-392 (merge);
  * trajectories     `save_traj/pose_es.npy`, `pose_gt.npy` - float [n,4,4] camera-to-world: tracker.py:352-362;
  * `performance.json` - {"tracking", "mapping": mean seconds per frame, "fps": 1 / mapping, "gpu_memory": MB}:
    utils/monitor.py:22-50.

The packed [N,59] layout of rtg_slam_amd.map_optim (xyz | f_dc | f_rest coefficient-major | opacity | scaling |
rotation) converts to and from the file's columns here."""
from __future__ import annotations

import json
import os
from typing import Dict, Optional

import numpy as np


def model_columns(include_confidence: bool = True, sh_rest: int = 45):
    cols = ["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(sh_rest)]
    cols += ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)]
    if include_confidence:
        cols.append("confidence")
    return cols


def _header(n: int, cols) -> bytes:
    lines = ["ply", "format binary_little_endian 1.0", f"element vertex {n}"]
    lines += [f"property float {c}" for c in cols] + ["end_header"]
    return ("\n".join(lines) + "\n").encode("ascii")


def save_model_ply(path: str, xyz, features_dc, features_rest, opacity, scaling, rotation, confidence=None,
                   include_confidence: bool = True) -> None:
    """`GaussianPointCloud.save_model_ply` (gaussian_pointcloud.py:424-466).  features_dc [N,1,3], features_rest
    [N,15,3] (coefficient-major, as the model holds them); everything raw.  An empty cloud writes nothing, as there."""
    a = lambda t: np.asarray(t.detach().cpu().numpy() if hasattr(t, "detach") else t, dtype=np.float32)
    xyz = a(xyz).reshape(-1, 3)
    n = xyz.shape[0]
    if n == 0:
        return
    f_dc = a(features_dc).reshape(n, -1, 3).transpose(0, 2, 1).reshape(n, -1)          # channel-major on disk
    f_rest = a(features_rest).reshape(n, -1, 3).transpose(0, 2, 1).reshape(n, -1)
    cols = [xyz, np.zeros_like(xyz), f_dc, f_rest, a(opacity).reshape(n, 1), a(scaling).reshape(n, 3), a(rotation).reshape(n, 4)]
    if include_confidence:
        cols.append((np.zeros((n, 1), np.float32) if confidence is None else a(confidence).reshape(n, 1)))
    table = np.ascontiguousarray(np.concatenate(cols, axis=1).astype("<f4"))
    names = model_columns(include_confidence, f_rest.shape[1])
    assert table.shape[1] == len(names)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        f.write(_header(n, names))
        f.write(table.tobytes())


def _read_ply_table(path: str):
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt = f.readline().split()
        if fmt[:2] != [b"format", b"binary_little_endian"]:
            raise ValueError(f"{path}: only binary_little_endian PLY is supported, got {fmt}")
        n, names, in_vertex = 0, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated header")
            tok = line.split()
            if tok[:1] == [b"end_header"]:
                break
            if tok[:1] == [b"element"]:
                in_vertex = tok[1] == b"vertex"
                if in_vertex:
                    n = int(tok[2])
            elif tok[:1] == [b"property"] and in_vertex:
                if tok[1] not in (b"float", b"float32"):
                    raise ValueError(f"{path}: property {tok[2]!r} is {tok[1]!r}; RTG-SLAM models are all float32")
                names.append(tok[2].decode())
        table = np.frombuffer(f.read(n * len(names) * 4), dtype="<f4").reshape(n, len(names))
    return names, table


def load_model_ply(path: str, max_sh_degree: int = 3) -> Dict[str, np.ndarray]:
    """`GaussianPointCloud.load` (gaussian_pointcloud.py:118-193): columns are looked up BY NAME; `confidence` is
    optional (zeros if absent, e.g. the `_sibr` files).  Returns raw float32 arrays: xyz [N,3], features_dc [N,1,3],
    features_rest [N,15,3], opacity [N,1], scaling [N,3], rotation [N,4], confidence [N,1]."""
    names, t = _read_ply_table(path)
    col = {k: i for i, k in enumerate(names)}
    n = t.shape[0]
    pick = lambda prefix: sorted((k for k in names if k.startswith(prefix)), key=lambda k: int(k.split("_")[-1]))
    rest = pick("f_rest_")
    assert len(rest) == 3 * (max_sh_degree + 1) ** 2 - 3, (len(rest), max_sh_degree)
    f_dc = np.stack([t[:, col[f"f_dc_{i}"]] for i in range(3)], axis=1).reshape(n, 3, 1).transpose(0, 2, 1)
    f_rest = np.stack([t[:, col[k]] for k in rest], axis=1).reshape(n, 3, -1).transpose(0, 2, 1)
    conf = t[:, col["confidence"]].reshape(n, 1) if "confidence" in col else np.zeros((n, 1), np.float32)
    return dict(
        xyz=np.stack([t[:, col[k]] for k in ("x", "y", "z")], axis=1).copy(),
        features_dc=np.ascontiguousarray(f_dc), features_rest=np.ascontiguousarray(f_rest),
        opacity=t[:, col["opacity"]].reshape(n, 1).copy(),
        scaling=np.stack([t[:, col[k]] for k in pick("scale_")], axis=1).copy(),
        rotation=np.stack([t[:, col[k]] for k in pick("rot")], axis=1).copy(),
        confidence=conf.copy())


def merge_ply(path_a: str, path_b: str, path_out: str, include_confidence: bool = True) -> None:
    """SLAM/utils.py:321-392 `merge_ply`: the rows of two model files with the same columns, concatenated."""
    na, ta = _read_ply_table(path_a)
    nb, tb = _read_ply_table(path_b)
    want = model_columns(include_confidence, sum(k.startswith("f_rest_") for k in na))
    ia, ib = [na.index(k) for k in want], [nb.index(k) for k in want]
    table = np.ascontiguousarray(np.concatenate([ta[:, ia], tb[:, ib]], axis=0).astype("<f4"))
    with open(path_out, "wb") as f:
        f.write(_header(table.shape[0], want))
        f.write(table.tobytes())


def packed_to_model(packed) -> Dict[str, np.ndarray]:
    """map_optim's packed raw [N,59] -> the model's arrays (see module docstring for the column order)."""
    p = np.asarray(packed.detach().cpu().numpy() if hasattr(packed, "detach") else packed, dtype=np.float32)
    n = p.shape[0]
    return dict(xyz=p[:, 0:3], features_dc=p[:, 3:6].reshape(n, 1, 3), features_rest=p[:, 6:51].reshape(n, 15, 3),
                opacity=p[:, 51:52], scaling=p[:, 52:55], rotation=p[:, 55:59])


def model_to_packed(m: Dict[str, np.ndarray]) -> np.ndarray:
    n = m["xyz"].shape[0]
    return np.concatenate([m["xyz"], m["features_dc"].reshape(n, 3), m["features_rest"].reshape(n, 45), m["opacity"],
                           m["scaling"], m["rotation"]], axis=1).astype(np.float32)


def save_trajectories(save_path: str, pose_es, pose_gt=None) -> None:
    """tracker.py:352-362: `save_traj/pose_es.npy` (and `pose_gt.npy`), stacked [n,4,4] camera-to-world."""
    d = os.path.join(save_path, "save_traj")
    os.makedirs(d, exist_ok=True)
    np.save(os.path.join(d, "pose_es.npy"), np.stack([np.asarray(p) for p in pose_es], axis=0))
    if pose_gt is not None:
        np.save(os.path.join(d, "pose_gt.npy"), np.stack([np.asarray(p) for p in pose_gt], axis=0))


class Recorder:
    """utils/monitor.py:10-50 without GPUtil: running means of per-frame seconds, fps = 1 / mean(mapping),
    peak GPU memory in MB (from torch.cuda instead of nvidia-smi), `performance.json`."""

    def __init__(self, gpu_id: int = 0) -> None:
        self._gpu_id = gpu_id
        self._value: Dict[str, float] = {}
        self._counter: Dict[str, int] = {}

    def update_max(self, name: str, value: float) -> None:
        if name not in self._value:
            self._value[name], self._counter[name] = value, 1
        else:
            self._value[name] = max(self._value[name], value)

    def update_mean(self, name: str, value: float, count: int) -> None:
        if count == 0:
            return
        if name not in self._value:
            self._value[name], self._counter[name] = value / count, count
        else:
            self._value[name] = (self._value[name] * self._counter[name] + value) / (self._counter[name] + count)
            self._counter[name] += count

    def cal_fps(self) -> None:
        self._value["fps"] = 1 / self._value["mapping"]
        self._counter["fps"] = 1

    def watch_gpu(self) -> float:
        import torch
        used = torch.cuda.max_memory_allocated(self._gpu_id) / (1024.0 * 1024.0) if torch.cuda.is_available() else 0.0
        self.update_max("gpu_memory", used)
        return used / 1024.0

    def save(self, directory: str) -> None:
        os.makedirs(directory, exist_ok=True)
        with open(os.path.join(directory, "performance.json"), "w") as f:
            json.dump(self._value, f)
