"""tests/golden/model_ply_ref.npz: the bytes the reference's OWN GaussianPointCloud.save_model_ply
(SLAM/gaussian_pointcloud.py:424-466, run on the CPU in place through oracle/ref_mapper_shim.py) hands to plyfile for a
seeded model, with and without the confidence column, and what its load_ply / load_model_ply path reads back.

    python -m oracle.gen_ply_golden            (build container only)

`plyfile` is not in this image.  The stand-in below implements only what the writer calls - PlyElement.describe(array,
"vertex") and PlyData([el]).write(path) - the way plyfile serialises a native little-endian structured array of f4 fields
(header lines `ply / format binary_little_endian 1.0 / element vertex N / property float <name> / end_header`, then
array.tobytes()).  The ATTRIBUTE TABLE (names, order, transposes, values) is entirely the reference's code."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class PlyElement:
    def __init__(self, data, name):
        self.data, self.name = data, name

    @staticmethod
    def describe(data, name):
        assert data.dtype.names and all(data.dtype[n] == np.dtype("<f4") for n in data.dtype.names)
        return PlyElement(data, name)


class PlyData:
    def __init__(self, elements):
        self.elements = elements

    def write(self, path):
        (el,) = self.elements
        head = "ply\nformat binary_little_endian 1.0\nelement %s %d\n" % (el.name, el.data.shape[0])
        head += "".join("property float %s\n" % n for n in el.data.dtype.names) + "end_header\n"
        with open(path, "wb") as f:
            f.write(head.encode("ascii"))
            f.write(el.data.tobytes())


def main():
    from oracle import ref_mapper_shim as rm
    rm.install()
    gp = sys.modules["SLAM.gaussian_pointcloud"]
    gp.PlyData, gp.PlyElement = PlyData, PlyElement
    from tests.test_io_formats import _model
    from tests.test_mapping_cpu import _args
    from oracle.gen_mapping_golden import reference_args
    m = _model(9, seed=6)
    pc = gp.GaussianPointCloud(reference_args(_args()))
    pc._xyz, pc._features_dc, pc._features_rest = m["xyz"], m["features_dc"], m["features_rest"]
    pc._opacity, pc._scaling, pc._rotation, pc._confidence = m["opacity"], m["scaling"], m["rotation"], m["confidence"]
    out = {"n": np.array([9]), "seed": np.array([6])}
    for inc in (True, False):
        path = f"/tmp/rtgs_ref_model_{int(inc)}.ply"
        pc.save_model_ply(path, include_confidence=inc)
        out[f"bytes_conf{int(inc)}"] = np.frombuffer(open(path, "rb").read(), dtype=np.uint8)
    path = os.path.join(ROOT, "tests", "golden", "model_ply_ref.npz")
    np.savez_compressed(path, **out)
    print({k: v.shape for k, v in out.items()}, "->", path)


if __name__ == "__main__":
    main()
