"""HIP SLAM-side ops (include/rtgs_slam.h, rtg_slam_amd/slam_ops.py) against the REFERENCE's outputs
(tests/golden/slam_ops.npz, written from /root/reference's own SLAM/utils.py by oracle/gen_slam_ops_golden.py) and,
at larger sizes, against oracle/slam_ops_oracle.py (which is pinned to the same golden file on CPU)."""
import os

import numpy as np
import pytest
import torch

from oracle import slam_ops_oracle as so
from rtg_slam_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def gold(golden_dir):
    z = np.load(os.path.join(golden_dir, "slam_ops.npz"))
    return {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}


def test_tile_mask_producers_vs_reference(gold):
    from rtg_slam_amd import slam_ops as ops
    pm, err = gold["pixelmask"].to(DEV), gold["color_error"].to(DEV)
    assert torch.equal(ops.transmission2tilemask(pm, 16, 0.5).cpu(), gold["t2t"])
    assert torch.equal(ops.pixelmask2tilemask(pm, 16).cpu(), gold["p2t"])
    assert torch.equal(ops.colorerror2tilemask(err, 16, 0.4).cpu(), gold["c2t"])
    # Replica size, several ratios / strides, vs the pinned oracle
    g = torch.Generator().manual_seed(2)
    H, W = 680, 1200
    pm = torch.rand(H, W, generator=g) < torch.linspace(0.1, 0.9, W)[None, :]
    err = torch.rand(H, W, generator=g) ** 2 * torch.linspace(0.2, 1.0, H)[:, None]
    for ratio in (0.25, 0.5, 0.75):
        assert torch.equal(ops.transmission2tilemask(pm.to(DEV), 16, ratio).cpu(), so.transmission2tilemask(pm, 16, ratio))
    assert torch.equal(ops.pixelmask2tilemask(pm.to(DEV), 8).cpu(), so.pixelmask2tilemask(pm, 8))
    for top in (0.1, 0.4, 0.9):
        got, want = ops.colorerror2tilemask(err.to(DEV), 16, top).cpu(), so.colorerror2tilemask(err, 16, top)
        assert int(got.sum()) == int(want.sum()) == int(got.numel() * top)
        # tile means are float sums in another order: the two top-k sets may swap tiles whose means tie to ~1e-7
        assert int((got != want).sum()) <= 2
    # ties: many equal tiles, k cuts through them -> exactly k on, lower tile index first
    flat = torch.zeros(64, 64)
    flat[:16, :32] = 1.0
    got = ops.colorerror2tilemask(flat.to(DEV), 16, 0.5).cpu().reshape(-1)
    assert got.tolist() == [1] * 8 + [0] * 8        # k = 8: the 2 tiles of mean 1, then 6 of the 14 zero-mean ties, in index order


def test_render_range_from_T_map():
    """mapper.py:500-508: render_mask = (T_map != 1); tile_mask = transmission2tilemask(render_mask, 16, 0.5)."""
    from rtg_slam_amd import slam_ops as ops
    g = torch.Generator().manual_seed(3)
    H, W = 70, 101
    T = torch.ones(1, H, W)
    blob = torch.rand(H, W, generator=g) < 0.4
    T[0][blob] = torch.rand(int(blob.sum()), generator=g) * 0.99
    mask, tile, count = ops.render_range(T.to(DEV), 0.5)
    assert torch.equal(mask.cpu(), T[0] != 1)
    assert torch.equal(tile.cpu(), so.transmission2tilemask(T[0] != 1, 16, 0.5))
    assert int(count.item()) == int((T[0] != 1).sum())


@pytest.mark.parametrize("kind,N", [("uniform", 5000), ("surface", 20000), ("clustered", 8000), ("dups", 3000), ("tiny", 3)])
def test_knn3_is_exact(kind, N):
    from rtg_slam_amd import slam_ops as ops
    g = torch.Generator().manual_seed(N)
    if kind == "uniform":
        p = torch.rand(N, 3, generator=g) * 4 - 2
    elif kind == "surface":
        p = synth.surface_gaussians(N, synth.CONFIG2, seed=5)["xyz"]
    elif kind == "clustered":
        c = torch.randn(20, 3, generator=g) * 3
        p = c[torch.randint(0, 20, (N,), generator=g)] + 0.01 * torch.randn(N, 3, generator=g)
        p[:50] += 100.0                                               # far outliers: rings of boxes must not miss them
    elif kind == "dups":
        p = torch.rand(N // 2, 3, generator=g).repeat(2, 1)           # every point has an exact duplicate (distance 0)
    else:
        p = torch.tensor([[0.0, 0, 0], [1, 0, 0], [0, 2, 0]])
    mean, idx, d3 = ops.distCUDA2(p.to(DEV), return_dist2=True)
    mean_o, idx_o, d3_o = so.dist2_knn(p)
    assert torch.equal(d3.cpu(), d3_o), kind                          # same float32 distances, bit for bit
    assert torch.equal(mean.cpu(), mean_o)
    idx = idx.cpu().long()
    if N > 3:
        assert torch.all(idx != torch.arange(N)[:, None]) and torch.all((idx >= 0) & (idx < N))
        for k in range(3):                                            # indices may differ on ties: check they realise the distance
            q = p[idx[:, k]]
            dx, dy, dz = p[:, 0] - q[:, 0], p[:, 1] - q[:, 1], p[:, 2] - q[:, 2]
            assert torch.equal(dx * dx + dy * dy + dz * dz, d3_o[:, k]), (kind, k)
    else:
        assert idx[0].tolist()[:2] == [1, 2] and idx[0, 2] == -1 and float(d3[0, 2]) == torch.finfo(torch.float32).max


def test_knn3_large_map_subset_check():
    """400 k points: brute force is too slow on the CPU for all rows; check 3 000 random rows against all points."""
    from rtg_slam_amd import slam_ops as ops
    N = 400_000
    p = synth.surface_gaussians(N, synth.REPLICA, seed=9)["xyz"]
    mean, idx, d3 = ops.distCUDA2(p.to(DEV), return_dist2=True)
    rows = torch.randperm(N, generator=torch.Generator().manual_seed(1))[:3000]
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    dx, dy, dz = x[rows, None] - x[None, :], y[rows, None] - y[None, :], z[rows, None] - z[None, :]
    d = dx * dx + dy * dy + dz * dz
    d[torch.arange(3000), rows] = float("inf")
    want = torch.topk(d, 3, dim=1, largest=False).values
    assert torch.equal(d3.cpu()[rows], want)
    assert torch.equal(mean.cpu()[rows], want.sum(1) / 3.0)


@pytest.mark.parametrize("kind", ["disjoint", "self", "few", "none", "boxed"])
def test_knn_query_is_exact(kind):
    """rtgs_knn3_query (what Mapping.temp_points_filter asks pytorch3d.knn_points for, and the new-point rows of
    update_geometry's distCUDA2) against brute force: same float32 distances bit for bit, indices realise them."""
    from rtg_slam_amd import slam_ops as ops
    g = torch.Generator().manual_seed(11)
    ref = synth.surface_gaussians(30000, synth.CONFIG2, seed=6)["xyz"]
    off, box = -1, None
    if kind == "disjoint":
        q = ref[torch.randperm(30000, generator=g)[:4000]] + 0.02 * torch.randn(4000, 3, generator=g)
        q[:40] += 50.0                                                # far outside the references' bounding box
    elif kind == "self":
        off = 1234
        q = ref[off:off + 5000].clone()                               # the queries ARE references: must not find themselves
    elif kind == "few":
        ref, q = ref[:2], torch.rand(300, 3, generator=g)
    elif kind == "boxed":                                             # bbox_filter: only references inside the queries' padded box
        q = ref[:3000] + 0.01 * torch.randn(3000, 3, generator=g)
        q = q[(q[:, 0] > -0.5) & (q[:, 0] < 0.7)]
        box = torch.cat([q.min(0).values - 0.05, q.max(0).values + 0.05])
    else:
        ref, q = ref[:0], torch.rand(10, 3, generator=g)
    d3, idx = ops.knn_query(ref.to(DEV), q.to(DEV), off, None if box is None else box.to(DEV))
    d3_o, idx_o = so.knn_query(ref, q, off, box)
    assert torch.equal(d3.cpu(), d3_o), kind
    idx = idx.cpu().long()
    if kind in ("disjoint", "self", "boxed"):
        assert torch.all((idx >= 0) & (idx < ref.shape[0]))
        if box is not None:
            assert torch.all((ref[idx] > box[:3]).all(-1) & (ref[idx] < box[3:]).all(-1))
        if off >= 0:
            assert torch.all(idx != (off + torch.arange(q.shape[0]))[:, None])
        for k in range(3):
            r = ref[idx[:, k]]
            dx, dy, dz = q[:, 0] - r[:, 0], q[:, 1] - r[:, 1], q[:, 2] - r[:, 2]
            assert torch.equal(dx * dx + dy * dy + dz * dz, d3_o[:, k]), (kind, k)
    elif kind == "few":
        assert torch.all(idx[:, 2] == -1) and torch.all(idx[:, :2] >= 0)
    else:
        assert torch.all(idx == -1)


def test_accumulate_gaussian_error_vs_oracle():
    from cuda_utils._C import accumulate_gaussian_error
    g = torch.Generator().manual_seed(4)
    H, W, P = 120, 160, 900
    ce, de, ne = torch.rand(H, W, 1, generator=g), torch.rand(H, W, 1, generator=g) * 0.3, torch.rand(H, W, 1, generator=g)
    # index maps with runs (an opaque disc owns neighbouring pixels) and holes
    ci = (torch.arange(H * W) // 37 % P).reshape(H, W, 1).int()
    di = (torch.arange(H * W) // 11 % (P - 100)).reshape(H, W, 1).int()
    ci[torch.rand(H, W, 1, generator=g) < 0.2] = -1
    di[torch.rand(H, W, 1, generator=g) < 0.3] = -1
    for mean in (True, False):
        got = accumulate_gaussian_error(H, W, P, ce.to(DEV), de.to(DEV), ne.to(DEV), ci.to(DEV), di.to(DEV), 0.6, 0.2, 0.9, mean)
        want = so.accumulate_gaussian_error(H, W, P, ce, de, ne, ci, di, 0.6, 0.2, 0.9, mean)
        for a, b in zip(got[:3], want[:3]):
            assert float((a.cpu() - b).abs().max()) <= 1e-5 * max(1.0, float(b.abs().max()))
        assert torch.equal(got[3].cpu(), want[3])
        assert float(got[1][P - 100:].abs().max()) == 0              # Gaussians nobody points at stay exactly 0


def test_bilateral_and_frame_preprocess_vs_reference(gold):
    from rtg_slam_amd import slam_ops as ops
    depth, K = gold["depth"].to(DEV), gold["K"].to(DEV)
    bf = ops.bilateralFilter_torch(depth, 5, 2, 2).cpu()
    ref = gold["bilateral"]
    assert bf.shape == ref.shape
    assert float((bf - ref).abs().max()) <= 2e-6 * float(ref.abs().max())        # expf on the GPU vs torch.exp
    assert torch.equal(bf == 0, ref == 0)
    for tag, filt in (("raw", False), ("filt", True)):
        out = {k: v.cpu() for k, v in ops.frame_preprocess(depth, K, 0.3, 5.0, filt, 0.2).items()}
        bad, bad_ref = out["invalid_confidence_mask"], gold[f"pre_{tag}_bad"]
        flips = bad != bad_ref                                           # confidence within an ulp of the 0.2 threshold
        assert float(flips.float().mean()) <= (0.0 if not filt else 2e-3), tag
        ok = ~flips
        tol = 0.0 if not filt else 3e-6                                  # the filtered depth itself carries the expf ulp
        for name, key in (("depth_map", "depth"), ("vertex_map_c", "vertex")):
            d = (out[name] - gold[f"pre_{tag}_{key}"]).abs().amax(dim=-1)
            assert float(d[ok].max()) <= tol * 5.0, (tag, name)
        dn = (out["normal_map_c"] - gold[f"pre_{tag}_normal"]).abs().amax(dim=-1)
        if not filt:
            assert float(dn[ok].max()) == 0.0                            # Sobel / cross / norm rounded as torch rounds them
        else:
            assert float((dn[ok] > 1e-3).float().mean()) < 1e-3
        dc = (out["confidence_map"] - gold[f"pre_{tag}_conf"]).abs()[..., 0]
        assert float(dc[ok].max()) <= (1e-6 if not filt else 1e-3)


def test_sample_pixels_candidates_and_draw(gold):
    from rtg_slam_amd import slam_ops as ops
    normal = gold["pre_raw_normal"]
    H, W = normal.shape[:2]
    sel = torch.rand(H, W, 1, generator=torch.Generator().manual_seed(6)) < 0.5
    for s in (None, sel):
        idx, count = ops.sample_candidates(normal.to(DEV), None if s is None else s.to(DEV))
        want = torch.nonzero(so.sample_pixels_mask(normal, s).reshape(-1)).reshape(-1)
        assert int(count.item()) == want.numel()
        assert torch.equal(idx[:want.numel()].cpu().long(), want)        # ascending pixel order
    vertex, color = gold["pre_raw_vertex"].to(DEV), torch.rand(H, W, 3).to(DEV)
    pts, nrm, col = ops.sample_pixels(vertex, normal.to(DEV), color, 500, sel.to(DEV))
    assert pts.shape == (500, 3) and nrm.shape == (500, 3) and col.shape == (500, 3)
    assert torch.all(nrm.sum(-1) != 0)                                   # never a zero-normal pixel
    # without replacement: 500 distinct vertices (the room's vertices are all distinct)
    assert torch.unique(pts, dim=0).shape[0] == 500
    n_all = int(so.sample_pixels_mask(normal, sel).sum())
    pts, _, _ = ops.sample_pixels(vertex, normal.to(DEV), color, 10 ** 9, sel.to(DEV))
    assert pts.shape[0] == n_all                                         # asks for more than there is: gets them all


def test_renderer_wrapper_matches_the_reference_wrapper_logic():
    """rtg_slam_amd.render.Renderer (interface of SLAM/render.py:21-145): same result dict as the reference wrapper's
    own lines restated with torch indexing, including the normal map of render.py:130-133 and its gradient to `normal`."""
    import math
    from types import SimpleNamespace
    from tests import raster_util as ru
    from rtg_slam_amd import synth
    from rtg_slam_amd.render import Renderer
    dev = "cuda:0"
    cam = synth.CameraSpec(136, 200, 180.0, 175.0, 99.5, 67.5)
    N = 6000
    g, s = ru.make_scene(N, cam, seed=31, pose_seed=2, r_range=(0.02, 0.1))
    args = SimpleNamespace(renderer_opaque_threshold=s.opaque_threshold,
                           renderer_normal_threshold=math.degrees(math.acos(s.normal_threshold)),
                           renderer_depth_threshold=s.depth_threshold, max_sh_degree=3, color_sigma=s.color_sigma,
                           active_sh_degree=-1)
    view = SimpleNamespace(FoVx=2 * math.atan(s.tanfovx), FoVy=2 * math.atan(s.tanfovy), image_height=s.image_height,
                           image_width=s.image_width, world_view_transform=s.viewmatrix.to(dev),
                           full_proj_transform=s.projmatrix.to(dev), camera_center=s.campos.to(dev), cx=s.cx, cy=s.cy)
    gd = {k: g[k].to(dev) for k in ru.FIELDS}
    gd["normal"] = gd["normal"].clone().requires_grad_(True)
    r = Renderer(args)
    gy, gx = (s.image_height + 15) // 16, (s.image_width + 15) // 16
    mask = (torch.rand(gy, gx, generator=torch.Generator().manual_seed(3)) < 0.8).int().to(dev)
    for tm in (None, mask):
        out = r.render(view, gd, tile_mask=tm)
        assert set(out) == {"render", "depth", "normal", "color_index_map", "depth_index_map", "color_hit_weight",
                            "depth_hit_weight", "T_map"}
        ref = ru.hip_run(s, g, tile_mask=tm, dev=dev)[0]
        for key, k in (("render", 0), ("depth", 1), ("color_index_map", 2), ("depth_index_map", 3), ("color_hit_weight", 4),
                       ("depth_hit_weight", 5), ("T_map", 6)):
            assert torch.equal(out[key].detach().cpu(), ref[k]), key
        # render.py:130-133, literally
        idx = out["depth_index_map"]
        want = torch.zeros_like(out["render"])
        nrm = gd["normal"].detach().clone().requires_grad_(True)
        want[:, idx[0] > -1] = nrm[idx[idx > -1].long()].permute(1, 0)
        assert torch.equal(out["normal"].detach(), want.detach())
        assert int((idx > -1).sum()) > 100
        w = torch.randn(3, s.image_height, s.image_width, generator=torch.Generator().manual_seed(5)).to(dev)
        (g_hip,) = torch.autograd.grad((out["normal"] * w).sum(), gd["normal"], retain_graph=True)
        (g_ref,) = torch.autograd.grad((want * w).sum(), nrm)
        assert float((g_hip - g_ref).abs().max()) <= 1e-5 * float(g_ref.abs().max() + 1e-12)


def test_transform_map_is_the_references_transform_map():
    """SLAM/utils.py:56-63 (restated: homogeneous coordinate, 4x4 product, first three components)."""
    from rtg_slam_amd import slam_ops as ops
    g = torch.Generator().manual_seed(3)
    m = torch.randn(37, 53, 3, generator=g)
    m[5:9] = 0
    T = torch.eye(4)
    T[:3, :3] = synth.look_at_pose(seed=2, max_angle_deg=40, max_trans=0.5)[:3, :3].float()
    T[:3, 3] = torch.tensor([0.3, -1.2, 2.0])
    out = ops.transform_map(m.to(DEV), T.to(DEV)).cpu()
    hom = torch.cat([m, torch.ones(37, 53, 1)], -1)
    want = torch.matmul(T[None, None].expand(37, 53, -1, -1), hom.unsqueeze(-1)).squeeze(-1)[..., :3]
    assert out.shape == m.shape and float((out - want).abs().max()) < 1e-6
    assert torch.equal(out[5:9], T[:3, 3].expand(4, 53, 3))               # zero vertices land on the translation, as there


def _frame_and_render(seed=5, H=96, W=144):
    g = torch.Generator().manual_seed(seed)
    T = torch.rand(1, H, W, generator=g)
    T[0, :10] = 1.0
    depth = 1.0 + 3.0 * torch.rand(H, W, 1, generator=g)
    depth[20:30] = 0.0
    rdepth = depth.permute(2, 0, 1) + 0.15 * torch.randn(1, H, W, generator=g)
    rdepth[0, 40:50] = 0.0
    didx = torch.randint(-1, 50, (1, H, W), generator=g, dtype=torch.int32)
    fcol, rcol = torch.rand(3, H, W, generator=g), torch.rand(3, H, W, generator=g)
    rcol[:, 60:] = fcol[:, 60:] + 0.01
    return T, depth, rdepth, rcol, fcol, didx


def test_add_masks_and_frame_errors_vs_the_mapper_lines():
    """rtgs_add_masks / rtgs_frame_errors against mapper.py:728-768 / :527-540 restated line by line in torch."""
    from rtg_slam_amd import slam_ops as ops
    T, depth, rdepth, rcol, fcol, didx = _frame_and_render()
    D = lambda t: t.to(DEV)
    tm, em, cnt = ops.add_masks(D(T), D(depth), D(rdepth), D(rcol), D(fcol), D(didx), 0.5, 0.1, 0.1)
    tm_o, em_o, cnt_o = so.add_masks(T, depth, rdepth, rcol, fcol, didx, 0.5, 0.1, 0.1)
    assert torch.equal(tm.cpu().bool(), tm_o) and torch.equal(em.cpu().bool(), em_o)
    assert cnt.cpu().tolist() == cnt_o.tolist() and 0 < cnt_o[0] and 0 < cnt_o[1]
    ce, de = ops.frame_errors(D(depth), D(rdepth), D(rcol), D(fcol), D(didx))
    ce_o, de_o = so.frame_errors(depth, rdepth, rcol, fcol, didx)
    assert torch.equal(de.cpu(), de_o) and float((ce.cpu() - ce_o).abs().max()) < 1e-6
    assert float(de_o[20:30].abs().max()) == 0 and float(ce_o[20:30].abs().max()) == 0


def test_attach_test_vs_the_mapper_lines():
    from rtg_slam_amd import slam_ops as ops
    g = torch.Generator().manual_seed(8)
    H, W, fx, fy, cx, cy = 60, 80, 70.0, 70.0, 39.5, 29.5
    c2w = synth.look_at_pose(seed=4, max_angle_deg=10, max_trans=0.2)
    w2c = torch.linalg.inv(c2w).float()
    pts_c = torch.stack([(torch.rand(3000, generator=g) - 0.5) * 3, (torch.rand(3000, generator=g) - 0.5) * 2.4,
                         0.5 + 3 * torch.rand(3000, generator=g)], -1)
    pts = pts_c @ c2w[:3, :3].float().T + c2w[:3, 3].float()            # many inside the view, some outside
    S = 200
    sxyz = pts[:S] + 0.02 * torch.randn(S, 3, generator=g)
    snrm = torch.nn.functional.normalize(torch.randn(S, 3, generator=g), dim=-1)
    cidx = torch.randint(-1, S, (1, H, W), generator=g, dtype=torch.int32)
    out = ops.attach_test(pts.to(DEV), w2c.to(DEV), fx, fy, cx, cy, H, W, cidx.to(DEV), sxyz.to(DEV), snrm.to(DEV), 0.05).cpu()
    want = so.attach_test(pts, w2c, fx, fy, cx, cy, H, W, cidx, sxyz, snrm, 0.05)
    diff = int((out != want).sum())
    assert 0 < int(want.sum()) < 3000 and diff <= 3, diff                # the projection is float32 either way: a pixel-border tie may flip


def test_new_gaussian_kernels_equal_the_torch_form():
    """rtgs_gather_new_points / rtgs_new_rows (round 6) against the tensor expressions they replace in Mapping._new_points and
    Mapping.temp_to_optimize (this package's restatement of gaussian_pointcloud.py:305-405, SLAM/utils.py:216-221)."""
    from rtg_slam_amd import mapping as mp, map_optim as mo, slam_ops as ops
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(5)
    H, W = 60, 80
    vertex = torch.randn(H, W, 3, generator=gen).to(dev)
    normal = torch.nn.functional.normalize(torch.randn(H, W, 3, generator=gen), dim=-1).to(dev) * 0.97      # not exactly unit: the kernel normalises
    color = torch.rand(H, W, 3, generator=gen).to(dev)
    pick = torch.randperm(H * W, generator=gen)[:777].to(dev)
    for ident in (False, True):
        xyz, nrm, col, rot = ops.gather_new_points(pick, vertex, normal, color, ident)
        n_t = normal.reshape(-1, 3)[pick]
        n_t = n_t / (torch.norm(n_t, p=2, dim=-1, keepdim=True) + 1e-8)
        assert torch.equal(xyz, vertex.reshape(-1, 3)[pick]) and torch.equal(col, color.reshape(-1, 3)[pick])
        assert (nrm - n_t).abs().max() < 1e-6
        r_t = torch.tensor([1.0, 0, 0, 0], device=dev).repeat(777, 1) if ident else mp.compute_rot(n_t)
        assert (rot - r_t).abs().max() < 2e-6
    # new_rows: candidates + existing Gaussians, neighbours from the exact query
    n, ne = 500, 3000
    xyz = (torch.rand(n, 3, generator=gen) * 2).to(dev)
    exist = (torch.rand(ne, 3, generator=gen) * 2).to(dev)
    exist_scales = (0.002 + 0.03 * torch.rand(ne, 3, generator=gen)).to(dev)
    colr = torch.rand(n, 3, generator=gen).to(dev)
    opac = torch.full((n, 1), 4.5951, device=dev)
    rots = torch.nn.functional.normalize(torch.randn(n, 4, generator=gen), dim=-1).to(dev)
    lo, hi = xyz.min(dim=0)[0] - 0.05, xyz.max(dim=0)[0] + 0.05
    d2, idx = ops.knn_query(torch.cat([xyz, exist]), xyz, 0, torch.cat([lo, hi]))
    factor = [1.0, 1.0, 0.1]
    rows, valid = ops.new_rows(xyz, colr, opac, rots, d2, idx, exist_scales, 0.001, 0.05, 1.0, factor)
    radius = (exist_scales.sum(dim=1) - exist_scales.min(dim=1).values) / 2
    total_radius = torch.cat([torch.full((n,), 1e-6, device=dev), radius])
    dist = torch.sqrt(d2) - 3 * total_radius[idx.clamp_min(0).long()]
    dist = torch.where(idx >= 0, dist, torch.full_like(dist, 1e30))
    invalid = (dist < 0).any(dim=-1)
    scales = torch.sqrt((dist ** 2).sum(dim=-1) / 3).clamp(0.001, 0.05)
    log_scales = torch.log(1.0 * scales[:, None] * torch.tensor(factor, device=dev)[None, :])
    assert torch.equal(valid.bool(), ~invalid) and 0 < int(invalid.sum()) < n
    ref = torch.zeros(n, mo.COLS, device=dev)
    ref[:, 0:3] = xyz; ref[:, 3:6] = mp.RGB2SH(colr); ref[:, 51:52] = opac; ref[:, 52:55] = log_scales; ref[:, 55:59] = rots
    assert (rows - ref).abs().max() < 2e-6


def test_the_draw_inside_the_gather_is_a_draw_without_replacement():
    """rtgs_draw_new_points (round 6; SLAM/utils.py:171 `randperm(n_cand)[:k]` + the gather in one launch): the k pixels are
    DISTINCT members of the candidate list for every key and every size (incl. 1, powers of two and their neighbours, k =
    n_cand = a full permutation), the rows equal rtgs_gather_new_points of the same pixels, two passes land behind one
    another in one set of arrays, and over many keys every candidate is drawn equally often (chi-square against k / n_cand)."""
    from rtg_slam_amd import slam_ops as ops
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(11)
    H, W = 60, 80
    vertex = torch.randn(H, W, 3, generator=gen).to(dev)
    normal = torch.nn.functional.normalize(torch.randn(H, W, 3, generator=gen), dim=-1).to(dev) * 0.97
    color = torch.rand(H, W, 3, generator=gen).to(dev)
    for n_cand, k in ((1, 1), (2, 1), (4, 4), (5, 4), (63, 10), (64, 64), (65, 20), (1000, 1000), (1025, 7), (4096, 2000), (4800, 4800)):
        cand = torch.randperm(H * W, generator=gen)[:n_cand].sort().values.to(torch.int32).to(dev)
        for key in (0, 1, 0xDEADBEEFCAFEF00D, 2 ** 64 - 1):
            xyz, nrm, col, rot, pick = ops.draw_new_points([(cand, n_cand, k, key)], vertex, normal, color, False, want_pick=True)
            p = pick.long()
            assert p.unique().numel() == k and torch.isin(p, cand.long()).all(), (n_cand, k, key)
            gx, gn, gc, gr = ops.gather_new_points(p, vertex, normal, color, False)
            if k != 3:
                assert torch.equal(xyz, gx) and torch.equal(nrm, gn) and torch.equal(col, gc) and torch.equal(rot, gr)
    # two passes, one set of arrays; different keys draw different pixels
    c1 = torch.arange(0, 2000, dtype=torch.int32, device=dev)
    c2 = torch.arange(2000, 4800, dtype=torch.int32, device=dev)
    xyz, nrm, col, rot, pick = ops.draw_new_points([(c1, 2000, 300, 7), (c2, 2800, 50, 8)], vertex, normal, color, True, want_pick=True)
    assert xyz.shape == (350, 3) and rot.shape == (350, 4) and (pick[:300] < 2000).all() and (pick[300:] >= 2000).all()
    assert torch.equal(rot, torch.tensor([1.0, 0, 0, 0], device=dev).repeat(350, 1))
    other = ops.draw_new_points([(c1, 2000, 300, 9)], vertex, normal, color, True, want_pick=True)[4]
    assert not torch.equal(other, pick[:300])
    # uniformity: 400 keys x 100 of 1000 -> every candidate expects 40 draws; chi-square with 999 degrees of freedom has mean
    # 999 and standard deviation 44.7: a flat draw stays inside +- 5 sigma, a biased one does not
    n_cand, k, keys = 1000, 100, 400
    cand = torch.arange(n_cand, dtype=torch.int32, device=dev)
    hits = torch.zeros(n_cand, dtype=torch.int64, device=dev)
    first = torch.zeros(n_cand, dtype=torch.int64, device=dev)
    rng = __import__("random").Random(3)
    for _ in range(keys):
        pick = ops.draw_new_points([(cand, n_cand, k, rng.getrandbits(64))], vertex, normal, color, True, want_pick=True)[4].long()
        hits += torch.bincount(pick, minlength=n_cand)
        first[pick[0]] += 1
    expect = keys * k / n_cand
    chi2 = float(((hits.double() - expect) ** 2 / expect).sum())
    assert abs(chi2 - (n_cand - 1) * (1 - k / n_cand)) < 5 * 44.7, chi2       # without replacement: variance shrinks by (1 - k/n)
    assert int(first.max()) <= 6                                               # output 0 does not favour a pixel (expects 0.4 each)


def test_filter_and_box_kernels_equal_the_torch_form():
    """rtgs_filter_keep / rtgs_bbox_pad (round 6) against the tensor expressions they replace in Mapping.temp_points_filter
    (mapper.py:803-827; bbox bounds SLAM/utils.py:737-744): the same float32 operations, so equal bit for bit."""
    from rtg_slam_amd import slam_ops as ops
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(13)
    for n, nu in ((1, 5), (700, 2), (5000, 3000), (40800, 900)):
        xyz = (torch.rand(n, 3, generator=gen) * 2).to(dev)
        exist = (torch.rand(nu, 3, generator=gen) * 2).to(dev)
        scales = (0.002 + 0.08 * torch.rand(nu, 3, generator=gen)).to(dev)
        box = ops.bbox_pad(xyz, 0.05)
        lo, hi = xyz.min(dim=0)[0] - 0.05, xyz.max(dim=0)[0] + 0.05
        assert torch.equal(box, torch.cat([lo, hi]))
        d2, idx = ops.knn_query(exist, xyz, -1, box)
        keep = ops.filter_keep(d2, idx, scales, 0.6)
        radius = (scales.sum(dim=1) - scales.min(dim=1).values) / 2
        rad = radius[idx.clamp_min(0).long()] * 0.6
        want = ~((torch.sqrt(d2) < rad) & (idx >= 0)).any(dim=-1)
        assert keep.dtype == torch.bool and torch.equal(keep, want)
        if n >= 700 and nu >= 900:
            assert 0 < int(want.sum()) < n
        if nu == 2:
            assert (idx[:, 2] < 0).all()                                       # fewer than three neighbours: -1 is "not inside"


def test_the_adds_compactions_equal_the_torch_form():
    """rtgs_compact_points / rtgs_append_valid_rows (round 6) against boolean-mask indexing and ShardedMapOptimizer.append_rows:
    same rows in the same order, same side arrays, same count - for none, some and all rows, across chunk boundaries (1024)."""
    from rtg_slam_amd import slam_ops as ops, map_optim as mo
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(17)
    for n, p_keep in ((1, 1.0), (5, 0.0), (1023, 0.5), (1024, 0.5), (1025, 0.9), (7000, 0.3), (40800, 0.05), (3000, 1.0)):
        keep = (torch.rand(n, generator=gen) < p_keep).to(dev)
        xyz, col = torch.randn(n, 3, generator=gen).to(dev), torch.rand(n, 3, generator=gen).to(dev)
        opa, rot = torch.randn(n, 1, generator=gen).to(dev), torch.randn(n, 4, generator=gen).to(dev)
        cx, cc, co, cr, m = ops.compact_points(keep, xyz, col, opa, rot)
        assert m == int(keep.sum())
        assert torch.equal(cx, xyz[keep]) and torch.equal(cc, col[keep]) and torch.equal(co, opa[keep]) and torch.equal(cr, rot[keep])
    for n, p_valid in ((4, 0.0), (1500, 0.6), (2048, 1.0), (5000, 0.2)):
        base = torch.randn(300, mo.COLS, generator=gen).to(dev)
        rows = torch.randn(n, mo.COLS, generator=gen).to(dev)
        valid = (torch.rand(n, generator=gen) < p_valid).to(torch.uint8).to(dev)
        maps = []
        for masked in (False, True):
            o = mo.ShardedMapOptimizer(base.clone(), lr_col=mo.default_lr_columns(), capacity=400)      # the append has to grow it
            for name, dtype, fill in (("confidence", torch.float32, 0.25), ("add_tick", torch.int32, 0), ("strikes", torch.int32, 7)):
                o.add_aux(name, 1, dtype, fill)
            v0 = o.version
            if masked:
                m = o.append_rows_masked(rows, valid, aux={"add_tick": 41})
            else:
                good = torch.nonzero(valid).reshape(-1)
                m = int(good.shape[0])
                if m:
                    o.append_rows(rows[good], aux={"add_tick": 41})
            assert m == int(valid.sum()) and o.N == 300 + m and (o.version > v0) == (m > 0)
            maps.append(o)
        a, b = maps
        assert torch.equal(a.params, b.params)
        for name in ("confidence", "add_tick", "strikes"):
            assert torch.equal(a.aux[name][:a.N], b.aux[name][:b.N]), name
        if m:
            assert int(b.aux["add_tick"][300, 0]) == 41 and int(b.aux["strikes"][b.N - 1, 0]) == 7 and float(b.aux["confidence"][300, 0]) == 0.25
        ga, gb = a.gaussian_data("all"), b.gaussian_data("all")
        assert all(torch.equal(ga[k], gb[k]) for k in ga)


def test_the_split_neighbour_search_equals_the_one_structure_search():
    """rtgs_knn3_build_ref + _query_built + _dynamic_merge (round 6: the structure over the stable Gaussians is kept between
    frames) against rtgs_knn3_query over cat(new points, stable, unstable): the same three distances bit for bit, the same
    indices wherever the three distances are distinct - with the open box cutting into all three groups, without unstable
    points, with fewer than three references inside the box."""
    from rtg_slam_amd import slam_ops as ops
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(23)
    for nq, ns, nu, shrink in ((300, 60000, 2500, 0.0), (1, 30000, 0, 0.0), (700, 25000, 4000, 0.35), (40, 20000, 3, 0.0), (1500, 50000, 900, 0.1),
                               (2, 20000, 1, -1.0)):
        far = shrink < 0                                                                       # the stable wall lies outside the box
        shrink = max(shrink, 0.0)
        stable = (torch.rand(ns, 3, generator=gen) * torch.tensor([4.0, 3.0, 0.05]) + torch.tensor([0.0, 0.0, 12.0 if far else 2.0])).to(dev)   # a wall
        unstable = (torch.rand(nu, 3, generator=gen) * torch.tensor([4.0, 3.0, 0.3]) + torch.tensor([0.0, 0.0, 1.9])).to(dev)
        q = (torch.rand(nq, 3, generator=gen) * torch.tensor([4.0, 3.0, 0.1]) + torch.tensor([0.0, 0.0, 1.98])).to(dev)
        lo, hi = q.min(0)[0] - 0.05, q.max(0)[0] + 0.05
        mid = (lo + hi) / 2
        box = torch.cat([mid + (lo - mid) * (1 - shrink), mid + (hi - mid) * (1 - shrink)])       # shrink > 0: some QUERIES lie outside too
        want_d, want_i = ops.knn_query(torch.cat([q, stable, unstable]), q, 0, box)
        built = ops.knn_build_ref(stable)
        d2s, ids = ops.knn_query_built(built, ns, q, box)
        got_d, got_i = ops.knn_dynamic_merge(q, unstable, ns, d2s, ids, box)
        assert torch.equal(got_d, want_d), (nq, ns, nu, float((got_d - want_d).abs().max()))
        distinct = (want_d[:, 0] != want_d[:, 1]) & (want_d[:, 1] != want_d[:, 2])
        assert torch.equal(got_i[distinct], want_i[distinct])
        assert torch.equal(got_i.sort(dim=1).values[~distinct] >= -1, torch.ones_like(got_i[~distinct], dtype=torch.bool))
        if far:
            assert bool((want_i[:, 2] < 0).all()) and bool((want_i[:, 0] >= 0).all())   # two references inside the box at most: -1 / FLT_MAX behind them
        # the structure answers a second, different query set too (it holds its own copy of the points)
        q2 = q + 0.01
        d2b, idb = ops.knn_query_built(built, ns, q2, None)
        wd, wi = ops.knn_query(stable, q2, -1, None)
        assert torch.equal(d2b, wd)


def test_bookkeeping_kernels_equal_the_torch_form():
    """rtgs_error_counters / rtgs_delete_mask (round 6) against the tensor expressions they replace in
    Mapping.error_gaussians_remove and Mapping.gaussians_delete (mapper.py:541-565, 298-335)."""
    from rtg_slam_amd import slam_ops as ops
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(9)
    nf, n_all = 5000, 6000
    g_color = (torch.rand(n_all, generator=gen) * 0.5).to(dev)
    g_depth = (torch.rand(n_all, generator=gen) * 0.5).to(dev)
    dcnt = torch.randint(0, 11, (n_all, 1), generator=gen, dtype=torch.int32).to(dev)
    ccnt = torch.randint(0, 11, (n_all, 1), generator=gen, dtype=torch.int32).to(dev)
    d0, c0 = dcnt.clone(), ccnt.clone()
    ddel, crel, (n_del, n_rel) = ops.error_counters(g_color, g_depth, nf, 0.2, 0.2, dcnt, ccnt, 10)
    d_ref, c_ref = d0.clone(), c0.clone()
    d_ref[:nf, 0] += (g_depth[:nf] > 0.2).to(torch.int32)
    c_ref[:nf, 0] += (g_color[:nf] > 0.2).to(torch.int32)
    del_ref = d_ref[:nf, 0] >= 10
    rel_ref = (c_ref[:nf, 0] >= 10) & ~del_ref
    assert torch.equal(dcnt, d_ref) and torch.equal(ccnt, c_ref)                  # rows >= nf untouched
    assert torch.equal(ddel.bool(), del_ref) and torch.equal(crel.bool(), rel_ref)
    assert (n_del, n_rel) == (int(del_ref.sum()), int(rel_ref.sum())) and n_del > 0 and n_rel > 0
    # delete mask: radius > 10 x mean, or older than the window
    n = 3333
    scales = (0.002 + 0.01 * torch.rand(n, 3, generator=gen)).to(dev)
    scales[7] = torch.tensor([0.5, 0.4, 0.001], device=dev)                        # a giant
    tick = torch.randint(0, 300, (n,), generator=gen, dtype=torch.int32).to(dev)
    radius = (scales.sum(dim=1) - scales.min(dim=1).values) / 2
    for with_tick in (True, False):
        mask, k = ops.delete_mask(scales, tick if with_tick else None, 250, 120)
        ref = radius > radius.mean() * 10
        if with_tick:
            ref = ref | ((250 - tick) > 120)
        assert torch.equal(mask.bool(), ref) and k == int(ref.sum()) and bool(mask[7])
