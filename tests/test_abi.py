"""The C-ABI library builds, loads and exports every symbol include/*.h declares (no compute)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for h in ("rtgs_raster.h", "rtgs_debug.h", "rtgs_icp.h", "rtgs_slam.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(rtgs_[a-z0-9_]+)\s*\(", src))
    names.discard("rtgs_resize_fn")
    return names


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from rtg_slam_amd import _lib
    lib = _lib.load()
    decl = declared_symbols()
    assert decl, "no declarations parsed"
    assert decl == set(_lib.EXPORTED_SYMBOLS)
    for name in decl:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.rtgs_version()


def test_product_path_has_no_oracle_or_cpu_fallback():
    for d in ("rtg_slam_amd", "diff_gaussian_rasterization_depth", "simple_knn", "cuda_utils"):
        for fn in os.listdir(os.path.join(ROOT, d)):
            if fn.endswith(".py"):
                src = open(os.path.join(ROOT, d, fn)).read()
                assert "import oracle" not in src and "from oracle" not in src, fn


def test_cpu_tensors_are_rejected_loudly():
    import pytest
    import torch
    from diff_gaussian_rasterization_depth import GaussianRasterizer
    from tests import raster_util as ru
    from rtg_slam_amd import synth
    cam = synth.CameraSpec(32, 32, 40.0, 40.0, 15.5, 15.5)
    g, s = ru.make_scene(10, cam)
    rast = GaussianRasterizer(raster_settings=ru.hip_settings(s, "cpu"))
    with pytest.raises(RuntimeError, match="HIP device"):
        rast(means3D=g["xyz"], opacities=g["opacity"], shs=g["shs"], colors_precomp=None, scales=g["scales"],
             rotations=g["rotations"], cov3D_precomp=None, normal_w=g["normal"], tile_mask=None)


def test_ctypes_mirrors_match_the_c_structs():
    """Layout guard: the two structs that cross the C ABI have the size their ctypes mirrors have (also enforced at
    load time), and the optimiser's one-call step refuses CPU tensors like every other entry point."""
    import ctypes as C
    import pytest
    import torch
    from rtg_slam_amd import _lib, map_optim as mo
    lib = _lib.load()
    assert C.sizeof(_lib.MapStepArgsC) == lib.rtgs_map_step_args_size()
    assert C.sizeof(_lib.RasterSettingsC) == lib.rtgs_raster_settings_size()
    packed = torch.zeros(8, mo.COLS)
    opt = mo.ShardedMapOptimizer(packed)           # CPU tensors: no arena, HIP kernels unavailable
    assert opt.grad_rows is None
    with pytest.raises(RuntimeError, match="HIP device"):
        opt.step(lambda gd: gd["xyz"].sum())
