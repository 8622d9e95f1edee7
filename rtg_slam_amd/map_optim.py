"""One map-optimisation iteration (the body of RTG-SLAM's hot loop B,
/root/reference/SLAM/multiprocess/mapper.py:176-205 -> :371-469) with the unstable-Gaussian set
sharded across the GPUs of one node.

Map state is block-SoA, three contiguous float32 tensors:

    xyz  [N,3]    used by the rasterizer as is                    (SLAM/gaussian_pointcloud.py _xyz)
    shs  [N,48]   used by the rasterizer as is, viewed [N,16,3]   (_features_dc | _features_rest)
    raw8 [N,8]    opacity | scaling xyz | rotation wxyz, raw      (_opacity, _scaling, _rotation)

Only raw8 goes through an activation kernel (rtgs_map_activate8_*: sigmoid / exp / normalize and
get_normal, gaussian_pointcloud.py:16-25, 538-550); xyz and the SH block are never copied, and the
rasterizer's backward writes their gradients in exactly the layout Adam consumes.

`step(loss_fn)`, per iteration and rank:  render this rank's view fwd+bwd  ->  for each of the three
tensors: reduce-scatter of the gradient rows over RCCL (each rank receives the rows of its shard), fused
Adam on the rank's N/world rows (optimizer state exists only for those rows), all-gather of the
updated rows.  With one rank the collectives vanish.  `step_slam(...)` is the same iteration with the
built-in SLAM loss, enqueued by one C call; with several ranks it keeps map and Adam state replicated
and exchanges only the gradient rows that exist.  Learning rates:
configs/replica_base.yaml:19-23, gaussian_pointcloud.py:252-283; Adam eps 1e-15 (mapper.py:156).

The render/loss closure, the Adam kernel and the activation are injected (HIP rasterizer +
rtgs_fused_adam + rtgs_map_activate8 in production; the CPU oracle + torch restatements in the
world_size-2 gloo tests) so the sharding / collective logic is testable without a GPU.

The packed [N,59] row layout (xyz 0:3 | f_dc 3:6 | f_rest 6:51 | opacity 51 | scaling 52:55 |
rotation 55:59 - the PLY column order of gaussian_pointcloud.py:407-466) is accepted by the
constructor and returned by `.params` for interchange.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Callable, Dict, Optional

import torch
import torch.distributed as dist

COLS = 59
BLOCKS = (("xyz", 0, 3), ("shs", 3, 51), ("raw8", 51, 59))


def default_lr_columns(position_lr=1e-3, feature_lr=5e-4, opacity_lr=0.0, scaling_lr=4e-3, rotation_lr=1e-3):
    lr = torch.zeros(COLS)
    lr[0:3] = position_lr
    lr[3:6] = feature_lr
    lr[6:51] = feature_lr / 20.0
    lr[51:52] = opacity_lr
    lr[52:55] = scaling_lr
    lr[55:59] = rotation_lr
    return lr


def global_lr_scale(final: bool = False, feature_lr_coef: float = 1.0, scaling_lr_coef: float = 1.0,
                    rotation_lr_coef: float = 1.0) -> torch.Tensor:
    """The learning-rate rescaling of Mapping.global_optimization as a [59] column factor (mapper.py:605-616; group order
    xyz, f_dc, f_rest, opacity, scaling, rotation - gaussian_pointcloud.py:252-283).  Keyframe-triggered form
    (`select_keyframe_num != -1`): position 0, every other group x 0.1.  Final form: position 0, f_dc and f_rest x
    feature_lr_coef, scaling x scaling_lr_coef, rotation x rotation_lr_coef, opacity unchanged."""
    s = torch.ones(COLS)
    s[0:3] = 0.0
    if not final:
        s[3:] = 0.1
    else:
        s[3:51] = feature_lr_coef
        s[52:55] = scaling_lr_coef
        s[55:59] = rotation_lr_coef
    return s


def pack_from_activated(g: Dict[str, torch.Tensor]) -> torch.Tensor:
    """Inverse activations of a `gaussian_data` dict (SLAM/render.py:93-98 keys) -> packed raw [N,59]."""
    N = g["xyz"].shape[0]
    out = torch.empty(N, COLS, dtype=torch.float32, device=g["xyz"].device)
    out[:, 0:3] = g["xyz"]
    out[:, 3:51] = g["shs"].reshape(N, 48)
    o = g["opacity"].clamp(1e-6, 1 - 1e-6)
    out[:, 51:52] = torch.log(o / (1 - o))
    out[:, 52:55] = torch.log(g["scales"])
    out[:, 55:59] = g["rotations"]
    return out


class _Activate8Hip(torch.autograd.Function):
    """raw8 activations as one HIP kernel each way (rtgs_map_activate8_forward / _backward)."""

    @staticmethod
    def forward(ctx, raw8, grad_rows=None):
        from . import _lib
        lib = _lib.load()
        ctx.grad_rows = grad_rows
        if not raw8.is_cuda:
            raise RuntimeError("rtg_slam_amd.map_optim: activate8_hip needs a HIP device tensor; no CPU path.")
        raw8 = raw8.contiguous()
        N, dev = raw8.shape[0], raw8.device
        f = dict(dtype=torch.float32, device=dev)
        op, sc, rot, nrm = torch.empty(N, 1, **f), torch.empty(N, 3, **f), torch.empty(N, 4, **f), torch.empty(N, 3, **f)
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            rc = lib.rtgs_map_activate8_forward(C.c_void_p(raw8.data_ptr()), N, C.c_void_p(op.data_ptr()),
                                                C.c_void_p(sc.data_ptr()), C.c_void_p(rot.data_ptr()),
                                                C.c_void_p(nrm.data_ptr()), C.c_void_p(stream))
        _lib.check(rc, "rtgs_map_activate8_forward")
        ctx.save_for_backward(raw8)
        return op, sc, rot, nrm

    @staticmethod
    def backward(ctx, g_op, g_sc, g_rot, g_nrm):
        from . import _lib
        lib = _lib.load()
        (raw8,) = ctx.saved_tensors
        N, dev = raw8.shape[0], raw8.device

        def z(g, *shape):
            return torch.zeros(*shape, dtype=torch.float32, device=dev) if g is None else g.contiguous()
        arena = ctx.grad_rows
        stream = torch.cuda.current_stream(dev).cuda_stream
        if arena is not None and arena.calls == 1 and arena.P == N and all(
                g is not None and g.data_ptr() == a.data_ptr()
                for g, a in ((g_op, arena.d_opac), (g_sc, arena.d_scales), (g_rot, arena.d_rots), (g_nrm, arena.d_normal))):
            # the four incoming gradients ARE the rasterizer's persistent rows: follow its row states
            with torch.cuda.device(dev):
                rc = lib.rtgs_map_activate8_backward_rows(
                    C.c_void_p(raw8.data_ptr()), N, C.c_void_p(g_op.data_ptr()), C.c_void_p(g_sc.data_ptr()),
                    C.c_void_p(g_rot.data_ptr()), C.c_void_p(g_nrm.data_ptr()), C.c_void_p(arena.row_state.data_ptr()),
                    C.c_void_p(arena.d_raw8.data_ptr()), C.c_void_p(stream))
            _lib.check(rc, "rtgs_map_activate8_backward_rows")
            return arena.d_raw8, None
        gs = (z(g_op, N, 1), z(g_sc, N, 3), z(g_rot, N, 4), z(g_nrm, N, 3))
        out = torch.empty_like(raw8)
        with torch.cuda.device(dev):
            rc = lib.rtgs_map_activate8_backward(C.c_void_p(raw8.data_ptr()), N, *(C.c_void_p(t.data_ptr()) for t in gs),
                                                 C.c_void_p(out.data_ptr()), C.c_void_p(stream))
        _lib.check(rc, "rtgs_map_activate8_backward")
        return out, None


def activate8_hip(raw8: torch.Tensor, grad_rows=None) -> Dict[str, torch.Tensor]:
    op, sc, rot, nrm = _Activate8Hip.apply(raw8, grad_rows)
    return dict(opacity=op, scales=sc, rotations=rot, normal=nrm)


def activate_packed(packed: torch.Tensor) -> Dict[str, torch.Tensor]:
    """The `gaussian_data` dict of SLAM/render.py:93-98 from packed raw parameters [N,59] (HIP activation kernels)."""
    N = packed.shape[0]
    gd = activate8_hip(packed[:, 51:59].contiguous())
    gd["xyz"] = packed[:, 0:3].contiguous()
    gd["shs"] = packed[:, 3:51].contiguous().view(N, 16, 3)
    return gd


def shard_rows(N: int, world: int):
    """Row partition used for reduce-scatter / all-gather: equal shards of ceil(N/world) rows
    (each tensor is padded to world * rows_per_rank rows)."""
    per = (N + world - 1) // world
    return per, per * world


def _adam_hip(p, g, m, v, lr_col, step, eps):
    from . import _lib
    if not p.is_cuda:
        raise RuntimeError("rtg_slam_amd.map_optim: fused Adam needs HIP device tensors; this build has no CPU path.")
    lib = _lib.load()
    stream = torch.cuda.current_stream(p.device).cuda_stream
    with torch.cuda.device(p.device):
        rc = lib.rtgs_fused_adam(C.c_void_p(p.data_ptr()), C.c_void_p(g.data_ptr()), C.c_void_p(m.data_ptr()),
                                 C.c_void_p(v.data_ptr()), C.c_void_p(lr_col.data_ptr()), p.shape[0], p.shape[1],
                                 int(step), 0.9, 0.999, float(eps), C.c_void_p(stream))
    _lib.check(rc, "rtgs_fused_adam")


def _adam_rows_hip(p, g, m, v, lr_col, step, eps, ever, row_state=None):
    """Row-skipping fused Adam (rtgs_fused_adam_rows): bit-identical to the dense kernel, untouched rows skipped."""
    from . import _lib
    if not p.is_cuda:
        raise RuntimeError("rtg_slam_amd.map_optim: fused Adam needs HIP device tensors; this build has no CPU path.")
    lib = _lib.load()
    stream = torch.cuda.current_stream(p.device).cuda_stream
    with torch.cuda.device(p.device):
        rc = lib.rtgs_fused_adam_rows(C.c_void_p(p.data_ptr()), C.c_void_p(g.data_ptr()), C.c_void_p(m.data_ptr()),
                                      C.c_void_p(v.data_ptr()), C.c_void_p(lr_col.data_ptr()), C.c_void_p(ever.data_ptr()),
                                      C.c_void_p(row_state.data_ptr() if row_state is not None else 0),
                                      p.shape[0], p.shape[1], int(step), 0.9, 0.999, float(eps), C.c_void_p(stream))
    _lib.check(rc, "rtgs_fused_adam_rows")


class _GrowArena:
    """Resize-callback target that keeps its buffer between calls (grows geometrically); a request of size 0
    returns the current buffer (the `rtgs_slam_map_step` contract)."""

    def __init__(self, device):
        from . import _lib
        self.device = device
        self.tensor = None
        self.cb = _lib.RESIZE_FN(self._resize)

    def _resize(self, _user, nbytes):
        try:
            nbytes = int(nbytes)
            if self.tensor is None or self.tensor.numel() < nbytes:
                self.tensor = torch.empty(max(int(nbytes * 1.25), 256), dtype=torch.uint8, device=self.device)
            return self.tensor.data_ptr()
        except Exception:
            return 0


class ShardedMapOptimizer:
    """The map (block-SoA, see the module docstring) with its optimiser.

    Row order: rows [0, n_frozen) are FROZEN - rendered, never differentiated, never stepped: the "stable" Gaussians
    of RTG-SLAM (mapper.py:1026-1108 renders cat(unstable, stable) but only the unstable cloud is parametrized, :143-156) -
    and rows [n_frozen, N) are the TRAINABLE ("unstable") ones.  Frozen rows come first so that `append_rows` - the
    reference's per-frame gaussians_add (gaussian_pointcloud.py:286-303) - is an O(new rows) write behind the last row;
    `freeze_rows` (gaussians_fix, mapper.py:253-271) and `remove_rows` (:298-335, gaussian_pointcloud.py:195-235) permute.
    Adam state, the attach snapshot, the row shards of the multi-GPU forms and the sparse gradient exchange cover the
    trainable rows only.  Storage has a capacity (geometric growth); `gaussian_data()` hands out zero-copy views."""

    def __init__(self, packed: torch.Tensor, lr_col: Optional[torch.Tensor] = None, eps: float = 1e-15,
                 group=None, adam_fn: Optional[Callable] = None, activate_fn: Optional[Callable] = None,
                 n_frozen: int = 0, capacity: Optional[int] = None):
        """`packed` [N,59] raw parameters.  `adam_fn(p, g, m, v, lr_col, step, eps)` defaults to the HIP
        fused Adam and `activate_fn(raw8) -> dict` to the HIP activation kernels (device tensors only -
        there is no CPU path in the product); the gloo tests inject torch restatements."""
        self.group = group
        self.row_skip = adam_fn is None and packed.is_cuda     # default HIP path: row-skipping Adam
        self.adam_fn = adam_fn if adam_fn is not None else _adam_hip
        self.activate_fn = activate_fn if activate_fn is not None else activate8_hip
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.backend = dist.get_backend(group) if dist.is_initialized() else "none"
        self.N = int(packed.shape[0])
        self.n_frozen = int(n_frozen)
        if not 0 <= self.n_frozen <= self.N:
            raise ValueError("n_frozen must be in [0, N]")
        self.device = packed.device
        self.lr_full = (default_lr_columns() if lr_col is None else lr_col).to(self.device).float()
        self.eps = eps
        self._use_arena = self.row_skip and activate_fn is None
        self.state = {}
        self.grad_rows = None
        self.act = None                # activated copies of raw8 (opacity, scales, rotations, normal), kept current by step_slam
        self._slam_ws = None
        self._slam_state = None        # world > 1: full-size Adam state of the replicated sparse step (step_slam)
        self.capacity = 0
        self.aux = {}                  # per-row side arrays that follow every append / remove / freeze (add_aux)
        self._aux_spec = {}
        self._scope = "local"          # "global": begin_global_optimization() - the stable prefix is what is rendered and trained
        self._lr_scope = None          # learning-rate columns of the running global optimisation
        self.version = 0               # bumped by everything that changes what a render of the map shows
        self._gd_views = {}            # gaussian_data's views per row range (see there)
        self._adam_dirty = None        # rows of state[.]["m" / "v" / "ever"] that may be non-zero: None, (lo, hi) or "all"
        self._frozen_version = 0       # bumped by everything that may change a FROZEN row (see frozen_key)
        self._allocate(max(int(capacity) if capacity is not None else self.N, self.N, 1), packed)
        self.step_count = 0
        self.total_steps = getattr(self, "total_steps", 0)      # steps ever taken (never reset: Mapping.gaussians_fix keys on it)
        self.last_render = None
        self.last_num_rendered = 0
        self.attach_init = None        # begin_local_optimization(): snapshot for the attach regulariser
        self._history = None           # ... and, with a confidence array, for history_merge()
        self._row_capacity = 16384     # world > 1: rows per rank in the sparse gradient exchange (grows on overflow)
        self._pending = None           # world > 1: the last exchange, until its overflow flag has been looked at
        self.overflow_redos = 0
        self._cap_peak, self._cap_steps, self._shrink_every = 0, 0, 32   # ... and shrinks when it stays mostly empty
        self._act_valid = False        # self.act holds the activation of the current raw8 (step_slam's tail re-activates moved rows)
        # 0: the one-call step's fused per-Gaussian tail; 1: the three-kernel form (A-B, tests; RTGS_TAIL_MODE at construction)
        self.tail_mode = 1 if __import__("os").environ.get("RTGS_TAIL_MODE", "0") == "1" else 0
        self.live_counts = torch.zeros(2, dtype=torch.int32, device=self.device) if packed.is_cuda else None   # fused tail: += {rows with gradient, rows stepped}
        self._mode = None              # world > 1: "sharded" (step) or "replicated" (step_slam); they keep different state

    # ------------------------------------------------------------------ storage
    @property
    def n_train(self) -> int:
        return self.N - self.n_frozen

    def _active(self):
        """(rows rendered, first trainable row) of the running optimisation.  Local optimisation (mapper.py:134-210): the
        whole map is rendered, the unstable suffix [n_frozen, N) is trained.  Global optimisation (mapper.py:594-707): only
        the stable prefix [0, n_frozen) is rendered, and all of it is trained - the unstable suffix is neither rendered nor
        stepped (`self.renderer.render(frame_input, self.stable_params, ...)`, :679-683)."""
        return (self.n_frozen, 0) if self._scope == "global" else (self.N, self.n_frozen)

    @property
    def frozen_key(self):
        """Identity of the frozen prefix's CONTENT: equal keys = the same rows with the same values (what a structure built over
        the stable Gaussians - Mapping's neighbour search - may be kept for).  Changes with a reallocation, a freeze, a removal
        or permutation that reaches into the prefix, and every step or merge of a global optimisation (which trains it)."""
        return (self.state["xyz"]["p"].data_ptr(), int(self.n_frozen), int(self._frozen_version))

    @property
    def n_active_train(self) -> int:
        n, t0 = self._active()
        return n - t0

    @property
    def per(self) -> int:
        """Trainable rows per rank in the row-sharded form."""
        return shard_rows(self.n_active_train, self.world)[0]

    @property
    def Npad(self) -> int:
        """Rows the parameter tensors must hold for the sharded form's all-gather: first trainable row + world * per."""
        return self._active()[1] + shard_rows(self.n_active_train, self.world)[1]

    def _lr(self, name):
        return self._lr_scope[name] if self._lr_scope is not None else self.state[name]["lr"]

    def _allocate(self, cap: int, packed: Optional[torch.Tensor] = None):
        """(Re)allocate every per-row array for `cap` rows (+ `world` rows of all-gather padding) and carry the live rows
        over.  Adam state and gradient rows start from zero: callers re-begin the optimisation after a change of shape."""
        dev, N = self.device, self.N
        rows = cap + self.world
        old = self.state
        self._gd_views = {}            # views of the arrays about to be replaced
        self._frozen_version = getattr(self, "_frozen_version", 0) + 1
        f = dict(dtype=torch.float32, device=dev)
        per_cap = (cap + self.world - 1) // self.world + 1
        adam_rows = cap if self.world == 1 else per_cap
        self.state = {}
        for name, c0, c1 in BLOCKS:
            full = torch.zeros(rows, c1 - c0, **f)
            if packed is not None:
                full[:N] = packed[:, c0:c1]
            elif old:
                full[:N] = old[name]["p"][:N]
            self.state[name] = dict(
                p=full, lr=self.lr_full[c0:c1].contiguous(),
                m=torch.zeros(adam_rows, c1 - c0, **f), v=torch.zeros(adam_rows, c1 - c0, **f),   # row 0 = first trainable row (of the shard)
                ever=torch.zeros(adam_rows, dtype=torch.uint8, device=dev),
                gpad=(torch.zeros(rows, c1 - c0, **f) if self.world > 1 else None),
                gshard=(torch.zeros(per_cap, c1 - c0, **f) if self.world > 1 else None))
            if old and self.world == 1:
                # a larger allocation of the SAME rows: their Adam state moves with them (an append that happens to
                # exhaust the capacity behaves like one that does not); on several ranks append_rows re-shards instead
                k = min(old[name]["m"].shape[0], adam_rows)
                for key in ("m", "v", "ever"):
                    self.state[name][key][:k] = old[name][key][:k]
        for name, (width, dtype, fill) in self._aux_spec.items():
            new = torch.full((rows, width), fill, dtype=dtype, device=dev)
            new[:N] = self.aux[name][:N]
            self.aux[name] = new
        self.capacity = cap
        if self._use_arena:
            # Single-GPU HIP path: persistent gradient rows + row states (rasterizer.RowGradArena).  step() hands the
            # arena to loss_fn as gd["grad_rows"]; a loss_fn that forwards it to the rasterizer (grad_rows=...) gets the
            # row-state backward, one that ignores it gets the dense path - the results are identical.
            from .rasterizer import RowGradArena
            self.grad_rows = RowGradArena(N, 16, dev, capacity=rows)
            self.grad_rows.train = (self.n_frozen, N)
            self.act = dict(opacity=torch.empty(rows, 1, **f), scales=torch.empty(rows, 3, **f),
                            rotations=torch.empty(rows, 4, **f), normal=torch.empty(rows, 3, **f))
        self._act_valid = False
        if self._slam_state is not None:   # the replicated form's full-size Adam state: same rows, larger arrays
            olds = self._slam_state
            self._slam_state = {}
            for n, c0, c1 in BLOCKS:
                k = olds[n]["m"].shape[0]
                new = dict(m=torch.zeros(cap, c1 - c0, **f), v=torch.zeros(cap, c1 - c0, **f),
                           ever=torch.zeros(cap, dtype=torch.uint8, device=dev))
                for key in ("m", "v", "ever"):
                    new[key][:min(k, cap)] = olds[n][key][:min(k, cap)]
                self._slam_state[n] = new
        if self._slam_ws is not None:
            self._slam_ws["radii"] = torch.empty(rows, dtype=torch.int32, device=dev)

    def _shape_changed(self, permuted: bool):
        """After rows were added / removed / permuted.  Appending keeps every existing row where it was: gradient rows,
        row states and Adam moments of the old rows stay valid and the new rows' are zero (never used since the last
        clear) - nothing is touched, the append stays O(new rows).  After a permutation they belong to other Gaussians:
        they are zeroed - lazily, by the next begin_local_optimization() / step (the reference creates a new Adam for
        every local optimisation anyway, mapper.py:156)."""
        if self.grad_rows is not None:
            self.grad_rows.resize(self.N, clear=False)
            self.grad_rows.train = (self.n_frozen, self.N)
        self._stale = getattr(self, "_stale", False) or permuted
        self.attach_init = None            # the snapshot no longer covers the trainable rows: begin_local_optimization()
        self._history = None
        self._pending = None

    def _mark_adam(self, lo: int, hi: int):
        """Rows [lo, hi) of this rank's Adam moment arrays are about to be written."""
        d = self._adam_dirty
        if d != "all" and hi > lo:
            self._adam_dirty = (int(lo), int(hi)) if d is None else (min(d[0], int(lo)), max(d[1], int(hi)))

    def _zero_adam(self):
        """A fresh Adam (mapper.py:156 creates one per optimisation) = zero moments.  Only the rows written since the last
        time are cleared: the arrays are allocated for the CAPACITY (800 000 rows x 59 columns x two moments = 377 MB in the
        SLAM sequence) while a local optimisation steps a few thousand trainable rows."""
        d = self._adam_dirty
        for holder in (self.state, self._slam_state or {}):
            part = holder is self.state and d != "all" and self.world == 1
            for n in holder:
                for k in ("m", "v", "ever"):
                    t = holder[n][k]
                    if not part:
                        t.zero_()
                    elif d is not None:
                        t[d[0]:min(d[1], t.shape[0])].zero_()
        self._adam_dirty = None

    def _clean(self):
        if getattr(self, "_stale", False):
            if self.grad_rows is not None:
                self.grad_rows.clear()
            self._zero_adam()
            self.step_count = 0
            self._stale = False

    @property
    def params(self) -> torch.Tensor:
        """Packed [N,59] copy of the current parameters."""
        self.flush()
        return torch.cat([self.state[n]["p"][:self.N] for n, _, _ in BLOCKS], dim=1)

    def add_aux(self, name: str, width: int = 1, dtype=torch.float32, fill=0):
        """A per-row side array [capacity, width] that lives with the rows: appended rows get `fill` (or what append_rows
        is handed), removed rows take theirs away, frozen rows move with them.  The reference keeps these as members of
        GaussianPointCloud (_confidence, _add_tick, _depth_error_counter, _color_error_counter,
        gaussian_pointcloud.py:286-303, 195-235).  Returns the full array; rows [0, N) are live."""
        rows = self.state["xyz"]["p"].shape[0]
        self._aux_spec[name] = (int(width), dtype, fill)
        self.aux[name] = torch.full((rows, int(width)), fill, dtype=dtype, device=self.device)
        return self.aux[name]

    def gaussian_data(self, rows: str = "all") -> Dict[str, torch.Tensor]:
        """The `gaussian_data` dict of SLAM/render.py:93-98 as ZERO-COPY views of the optimiser's own arrays: xyz / shs are
        the parameter tensors, opacity / scales / rotations / normal the activated copies the one-call step keeps current
        (its tail re-activates exactly the rows it moved) - refreshed here by one activation pass only if something else
        changed raw8.  Valid until the next step / change of shape; do not write through them.
        `rows`: "all" = Mapping.global_params (mapper.py:1069-1108), "stable" = stable_params (:984-1011: the frozen prefix),
        "unstable" = unstable_params (:1013-1037: the trainable suffix) - contiguous row ranges, so still views."""
        self.flush()
        N = self.N
        r0, r1 = {"all": (0, N), "stable": (0, self.n_frozen), "unstable": (self.n_frozen, N)}[rows]
        if self.act is None:
            gd = self.activate_fn(self.state["raw8"]["p"][r0:r1])
            gd = {k: gd[k] for k in ("opacity", "scales", "rotations", "normal")}
        else:
            self._activate_rows(0, N)
            # the views themselves are remembered per (row range, arrays): a SLAM frame asks seven times, and seven slices a
            # time are ~100 us of interpreter per frame; the CONTENT is kept current in place by _activate_rows
            key = (r0, r1, self.state["xyz"]["p"].data_ptr(), self.state["shs"]["p"].data_ptr(), self.act["scales"].data_ptr())
            hit = self._gd_views.get(rows)
            if hit is not None and hit[0] == key:
                return dict(hit[1])
            gd = {k: v[r0:r1] for k, v in self.act.items()}
        gd["xyz"] = self.state["xyz"]["p"][r0:r1]
        gd["shs"] = self.state["shs"]["p"][r0:r1].view(r1 - r0, 16, 3)
        if self.act is not None:
            self._gd_views[rows] = (key, dict(gd))
        return gd

    def _activate_rows(self, r0: int, r1: int, force: bool = False):
        """Bring self.act up to date: everything when it is stale, else (force) just rows [r0, r1)."""
        if self.act is None or (self._act_valid and not force):
            return
        from . import _lib
        lib = _lib.load()
        if not self._act_valid:
            r0, r1 = 0, self.N
        if r1 > r0:
            dev = self.device
            P = lambda t, c: C.c_void_p(t.data_ptr() + 4 * c * r0)
            with torch.cuda.device(dev):
                rc = lib.rtgs_map_activate8_forward(P(self.state["raw8"]["p"], 8), r1 - r0, P(self.act["opacity"], 1),
                                                    P(self.act["scales"], 3), P(self.act["rotations"], 4), P(self.act["normal"], 3),
                                                    C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            _lib.check(rc, "rtgs_map_activate8_forward")
        self._act_valid = True

    def append_rows(self, packed_new: torch.Tensor, aux: Optional[Dict[str, torch.Tensor]] = None):
        """New TRAINABLE rows behind the last row (gaussians_add -> GaussianPointCloud.cat, gaussian_pointcloud.py:286-303:
        the reference concatenates every parameter tensor - an O(N) copy per frame; here O(new) unless the capacity is
        exhausted, which grows by half).  On several ranks every rank must append the same rows.  `aux`: values of the
        side arrays (add_aux) for the new rows, [n, width] or a scalar each; arrays not named get their fill value."""
        if self._scope != "local":
            raise RuntimeError("append_rows() inside a global optimisation: end_global_optimization() first")
        self.flush()
        n = int(packed_new.shape[0])
        if n == 0:
            return
        carried = self._gather_sharded_adam()
        if self.N + n > self.capacity:
            self._allocate(max(self.N + n, self.capacity + self.capacity // 2))
        r0 = self.N
        for name, c0, c1 in BLOCKS:
            self.state[name]["p"][r0:r0 + n] = packed_new[:, c0:c1]
        for name, (width, dtype, fill) in self._aux_spec.items():
            val = fill if aux is None or name not in aux else aux[name]
            if torch.is_tensor(val):
                val = val.reshape(n, width).to(dtype)
            self.aux[name][r0:r0 + n] = val
        self.version += 1
        self.N += n
        self._scatter_sharded_adam(carried)
        self._shape_changed(permuted=False)
        if self.act is not None and self._act_valid:
            self._activate_rows(r0, self.N, force=True)        # only the new rows

    def append_rows_masked(self, packed_new: torch.Tensor, valid: torch.Tensor, aux: Optional[Dict[str, object]] = None) -> int:
        """append_rows(packed_new[valid != 0]) without the compaction on the host's side: one kernel (rtgs_append_valid_rows)
        writes the accepted rows, in order, straight behind the map's last row and counts them; the one host synchronisation
        reads the count (the tensor form: nonzero, a gather, three block copies, one fill per side array).  `aux`: SCALAR values
        of side arrays for the new rows.  One rank, local scope, HIP arrays only; returns how many rows were appended."""
        if self._scope != "local" or self.world != 1 or not packed_new.is_cuda:
            raise RuntimeError("append_rows_masked(): one rank, local scope, device arrays")
        from . import _lib
        self.flush()
        n = int(packed_new.shape[0])
        if n == 0:
            return 0
        if self.N + n > self.capacity:
            self._allocate(max(self.N + n, self.capacity + self.capacity // 2))
        lib, dev, r0 = _lib.load(), self.device, self.N
        rows = packed_new.float().contiguous()
        v8 = valid.view(torch.uint8) if valid.dtype == torch.bool else valid.to(torch.uint8)
        names = list(self._aux_spec)
        ptrs = (C.c_void_p * max(len(names), 1))()
        bits = (C.c_uint32 * max(len(names), 1))()
        import struct
        for k, name in enumerate(names):
            width, dtype, fill = self._aux_spec[name]
            if width != 1 or self.aux[name].element_size() != 4:
                raise RuntimeError("append_rows_masked(): side arrays of one 4-byte element per row")
            val = fill if aux is None or name not in aux else aux[name]
            bits[k] = struct.unpack("<I", struct.pack("<f", float(val)) if dtype.is_floating_point else struct.pack("<i", int(val)))[0]
            ptrs[k] = self.aux[name].data_ptr() + 4 * r0
        count = torch.empty(1, dtype=torch.int32, device=dev)
        P = lambda name, c: C.c_void_p(self.state[name]["p"].data_ptr() + 4 * c * r0)
        with torch.cuda.device(dev):
            rc = lib.rtgs_append_valid_rows(n, C.c_void_p(v8.contiguous().data_ptr()), C.c_void_p(rows.data_ptr()), P("xyz", 3), P("shs", 48),
                                            P("raw8", 8), len(names), ptrs, bits, C.c_void_p(count.data_ptr()),
                                            C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        _lib.check(rc, "rtgs_append_valid_rows")
        m = int(count.item())
        if m == 0:
            return 0
        self.version += 1
        self.N += m
        self._shape_changed(permuted=False)
        if self.act is not None and self._act_valid:
            self._activate_rows(r0, self.N, force=True)        # only the new rows
        return m

    def _gather_sharded_adam(self):
        """Row-sharded form on several ranks: the shard a trainable row belongs to depends on N (shard_rows), so an append
        moves rows between ranks - and their Adam moments must move with them, or the next step pairs moments and rows of
        different Gaussians.  Gathers m / v / ever of every shard into global trainable-row order (None when there is
        nothing to carry: one rank, the replicated form, no step taken yet, or a pending reset)."""
        if self.world == 1 or self._mode != "sharded" or self.step_count == 0 or getattr(self, "_stale", False):
            return None
        per, out = self.per, {}
        for name, _, _ in BLOCKS:
            st = self.state[name]
            out[name] = {}
            for key in ("m", "v", "ever"):
                mine = st[key][:per].contiguous()
                parts = [torch.empty_like(mine) for _ in range(self.world)]
                dist.all_gather(parts, mine, group=self.group)
                out[name][key] = torch.cat(parts, dim=0)[:self.n_train]
        return out

    def _scatter_sharded_adam(self, carried):
        if carried is None:
            return
        self._adam_dirty = "all"
        per, lo = self.per, self.rank * self.per
        for name, _, _ in BLOCKS:
            st = self.state[name]
            for key in ("m", "v", "ever"):
                full = carried[name][key]
                st[key].zero_()
                k = max(0, min(per, full.shape[0] - lo))
                if k:
                    st[key][:k] = full[lo:lo + k]

    def _permute(self, keep_idx: torch.Tensor, n_frozen: int, start: int = 0):
        """Rows [start, N) become the rows `keep_idx` (indices >= start, in their new order); rows before `start` stay where
        they are - a deletion or a freeze among the trainable suffix of a SLAM map moves a few thousand rows, not the map."""
        if self._scope != "local":
            raise RuntimeError("rows cannot be removed or frozen inside a global optimisation: end_global_optimization() first")
        n = int(start) + int(keep_idx.numel())
        if start < self.n_frozen or int(n_frozen) != self.n_frozen:
            self._frozen_version += 1
        for name, _, _ in BLOCKS:
            pfull = self.state[name]["p"]
            pfull[start:n] = pfull.index_select(0, keep_idx)
            if n < self.N:
                pfull[n:self.N].zero_()
        for name, (width, dtype, fill) in self._aux_spec.items():
            a = self.aux[name]
            a[start:n] = a.index_select(0, keep_idx)
            if n < self.N:
                a[n:self.N] = fill
        self.version += 1
        self.N, self.n_frozen = n, int(n_frozen)
        self._act_valid = False
        self._shape_changed(permuted=True)

    def remove_rows(self, mask: torch.Tensor, start: int = 0):
        """Delete the rows where `mask` [N] is set (GaussianPointCloud.delete / remove, gaussian_pointcloud.py:195-235;
        mapper.py:298-335 drops unstable Gaussians that outlived their window or went transparent).  Order is kept.
        `start`: the caller's promise that no row before it is set in the mask (e.g. n_frozen when only trainable rows go) -
        the rows before it are then neither read nor moved."""
        self.flush()
        start = max(0, min(int(start), self.N))
        mask = mask.to(self.device).bool().reshape(-1)
        keep = ~mask[start:] if start else ~mask
        nf = self.n_frozen if start >= self.n_frozen else (start + int(keep[:self.n_frozen - start].sum()) if self.n_frozen else 0)
        idx = torch.nonzero(keep).reshape(-1)
        self._permute(idx + start if start else idx, nf, start)

    def freeze_rows(self, mask: torch.Tensor):
        """Make the trainable rows where `mask` [N] is set FROZEN (gaussians_fix: unstable Gaussians whose confidence passed
        the threshold join the stable cloud, mapper.py:253-271): they move, in order, behind the frozen prefix."""
        self.flush()
        nf0 = self.n_frozen
        sub = mask.to(self.device).bool().reshape(-1)[nf0:]                  # the frozen prefix stays where it is
        order = torch.sort((~sub).to(torch.int8), stable=True).indices      # newly frozen first, order kept inside both parts
        self._permute(order + nf0 if nf0 else order, nf0 + int(sub.sum()), nf0)

    def _adam(self, name, shard, gs, row_state=None):
        self._adam_dirty = "all"                            # shard-relative rows: no range kept for this path
        st, lr = self.state[name], self._lr(name)
        n = shard.shape[0]                                  # the state arrays are allocated for the capacity
        if self.row_skip:
            _adam_rows_hip(shard, gs, st["m"][:n], st["v"][:n], lr, self.step_count, self.eps, st["ever"][:n], row_state)
        else:
            self.adam_fn(shard, gs, st["m"][:n], st["v"][:n], lr, self.step_count, self.eps)

    # ------------------------------------------------------------------ one-call SLAM step (single GPU)
    def begin_local_optimization(self, confidence: Optional[torch.Tensor] = None):
        """(`confidence` float[n_train], optional: also snapshot the SH block and the confidences, for `history_merge`.)
        Snapshot the raw parameters the attach regulariser ties low-opacity Gaussians to (`history_stat` /
        `init_stat` of mapper.py:147-153, 660-666) and reset the Adam state, as the reference does for every
        local / global optimisation (mapper.py:156: a new torch.optim.Adam per call) - in place: nothing is allocated
        once the buffers exist.  Covers the trainable rows.  Ends a global optimisation that was still open."""
        if self._scope != "local":
            self.end_global_optimization()
        self._begin(confidence)

    def begin_global_optimization(self, lr_scale: Optional[torch.Tensor] = None):
        """Mapping.global_optimization (mapper.py:594-707) as a mode of the map object: until `end_global_optimization()`
        (or the next `begin_local_optimization()`) the STABLE prefix [0, n_frozen) is what `step_slam` / `step` render AND
        train - `self.renderer.render(frame_input, self.stable_params, ...)` (:679-683) with `l = self.stable_pointcloud.
        parametrize(update_args)` (:605) - while the unstable suffix is neither rendered nor stepped.  `lr_scale` [59]
        multiplies the learning-rate columns for the duration (:606-616: position 0 and everything else x 0.1 for the
        keyframe-triggered form, position 0 and feature / scaling / rotation x their `*_lr_coef` for the final one; see
        `global_lr_scale`).  Adam state restarts from zero (:627: a new Adam) and the attach snapshot (`init_stat`,
        :621-626) covers the stable rows.  Adam state, snapshot and a `confidence` array handed to `step_slam` index the
        stable rows: element 0 = row 0."""
        self.flush()
        if self.n_frozen == 0:
            raise RuntimeError("begin_global_optimization() on a map without stable rows (mapper.py:603-604 returns early)")
        self._scope = "global"
        if lr_scale is not None:
            full = self.lr_full * lr_scale.to(self.device).float()
            self._lr_scope = {name: full[c0:c1].contiguous() for name, c0, c1 in BLOCKS}
        else:
            self._lr_scope = None
        if self.grad_rows is not None:
            self.grad_rows.resize(self.n_frozen, clear=False)       # the rasterizer sees a map of n_frozen rows
            self.grad_rows.train = (0, self.n_frozen)
        self._stale = True                  # gradient rows / row states of the local optimisation before it: cleared by _begin
        self._begin(None)

    def end_global_optimization(self):
        """Back to the local form: the whole map is rendered, the unstable suffix is trained (`stable_pointcloud.detach()`,
        mapper.py:707).  Adam state and gradient rows restart from zero at the next begin_local_optimization() / step."""
        if self._scope == "local":
            return
        self.flush()
        self._scope = "local"
        self._lr_scope = None
        if self.grad_rows is not None:
            self.grad_rows.resize(self.N, clear=False)
            self.grad_rows.train = (self.n_frozen, self.N)
        self._stale = True
        self.attach_init = None
        self._history = None

    def _begin(self, confidence):
        self.flush()
        self._clean()
        st = self.state
        N, nf = self._active()
        rows = st["xyz"]["p"].shape[0]
        buf = getattr(self, "_attach_buf", None)
        if buf is None or buf["xyz"].shape[0] < rows:
            buf = self._attach_buf = dict(xyz=torch.empty(rows, 3, dtype=torch.float32, device=self.device),
                                          raw8=torch.empty(rows, 8, dtype=torch.float32, device=self.device),
                                          info=torch.zeros(6, dtype=torch.float32, device=self.device))
        buf["xyz"][:N - nf].copy_(st["xyz"]["p"][nf:N])
        buf["raw8"][:N - nf].copy_(st["raw8"]["p"][nf:N])
        self.attach_init = dict(xyz=buf["xyz"][:N - nf], raw8=buf["raw8"][:N - nf], info=buf["info"])
        self._history = None
        if confidence is not None:
            if buf.get("shs") is None or buf["shs"].shape[0] < rows:
                buf["shs"] = torch.empty(rows, 48, dtype=torch.float32, device=self.device)
                buf["conf"] = torch.empty(rows, dtype=torch.float32, device=self.device)
            buf["shs"][:N - nf].copy_(st["shs"]["p"][nf:N])
            buf["conf"][:N - nf].copy_(confidence.reshape(-1))
            self._history = dict(shs=buf["shs"][:N - nf], conf=buf["conf"][:N - nf])
        self._zero_adam()
        self.step_count = 0
        if st["xyz"]["p"].is_cuda and N > nf:
            self.attach_loss()                 # counts the selected rows once: the selection is fixed by the snapshot

    def history_merge(self, confidence: torch.Tensor, max_weight: float = 0.5):
        """Mapping.history_merge (mapper.py:212-251; history_merge_max_weight 0.5, configs/base.yaml:54), called by the
        reference when the iterations of a local optimisation are done: the trainable rows are blended with the snapshot
        `begin_local_optimization(confidence=...)` took, weighted by max_weight * confidence_then / (confidence_now + 1e-6)
        (rtgs_history_merge; the reference's `history_weight[0]` indexing is kept).  `confidence` float[n_train] = now."""
        from . import _lib
        if self._history is None or self.attach_init is None:
            raise RuntimeError("history_merge() needs begin_local_optimization(confidence=...) on the current rows")
        if max_weight <= 0:
            return
        self.flush()
        lib = _lib.load()
        (N, nf), st, ai, h = self._active(), self.state, self.attach_init, self._history
        if N == nf:
            return
        self.version += 1
        if self._scope == "global":
            self._frozen_version += 1
        V = lambda t: C.c_void_p(t.data_ptr())
        O = lambda t, c: C.c_void_p(t.data_ptr() + 4 * c * nf)
        conf = confidence.reshape(-1).contiguous().float()
        dev = self.device
        with torch.cuda.device(dev):
            rc = lib.rtgs_history_merge(O(st["xyz"]["p"], 3), O(st["shs"]["p"], 48), O(st["raw8"]["p"], 8), V(ai["xyz"]), V(h["shs"]),
                                        V(ai["raw8"]), V(h["conf"]), V(conf), N - nf, float(max_weight),
                                        C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        _lib.check(rc, "rtgs_history_merge")
        self._act_valid = False            # scaling and rotation moved outside the step's tail

    def attach_loss(self) -> torch.Tensor:
        """Value of the attach regulariser at the current parameters (mapper.py:384-401; `scale_loss` of the
        reference's report) - a device scalar.  The one-call step applies its gradient without evaluating it."""
        from . import _lib
        lib = _lib.load()
        ai, st, (N, nf) = self.attach_init, self.state, self._active()
        dev = st["xyz"]["p"].device
        attach = _lib.AttachC(ai["xyz"].data_ptr(), ai["raw8"].data_ptr(), ai["info"].data_ptr())
        with torch.cuda.device(dev):
            rc = lib.rtgs_attach_prepare(C.c_void_p(st["xyz"]["p"].data_ptr() + 12 * nf), C.c_void_p(st["raw8"]["p"].data_ptr() + 32 * nf),
                                         C.byref(attach), N - nf, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        _lib.check(rc, "rtgs_attach_prepare")
        return ai["info"][1]

    def step_slam(self, raster_settings, gt_color: torch.Tensor, gt_depth: torch.Tensor,
                  tile_mask: Optional[torch.Tensor] = None, color_weight: float = 0.8,
                  depth_weight: float = 1.0, ssim_weight: float = 0.2, add_depth_thres: float = 0.1,
                  render_mask: Optional[torch.Tensor] = None, confidence: Optional[torch.Tensor] = None,
                  tile_band: bool = False, normal_weight: float = 0.0,
                  gt_normal: Optional[torch.Tensor] = None) -> torch.Tensor:
        """(`confidence`, if given, is float[n_train]: element 0 belongs to the first trainable row.)
        One iteration with the built-in SLAM loss (`slam_losses`): identical kernels and results as
        `step(lambda gd: slam_losses_hip(render(gd), gt_color, gt_depth))`, but enqueued by a single C call
        (`rtgs_slam_map_step`) - no autograd graph, no per-launch Python.  With more than one rank the map and the
        Adam state stay REPLICATED and only the gradient rows that exist travel: every rank renders its view, packs
        the rows that received gradient (`rtgs_rows_pack`), the packed lists are all-gathered (a few MB instead of
        the 283 MB of a dense reduce-scatter + all-gather at 1.2 M Gaussians), every rank adds all lists in rank order
        and takes the same Adam step - the replicas stay bit-identical, no parameter all-gather is needed.  HIP path
        only (injected torch kernels fall back to `step`).  Returns this rank's loss (0-dim device view, overwritten by
        the next call); the rendered images of the step are in `self.last_render`."""
        from . import _lib
        from .rasterizer import GaussianRasterizer, _Keep, current_context
        self._clean()
        if self.grad_rows is None:
            rast = GaussianRasterizer(raster_settings)

            def loss_fn(gd):
                out = rast(means3D=gd["xyz"], opacities=gd["opacity"], shs=gd["shs"], scales=gd["scales"],
                           rotations=gd["rotations"], normal_w=gd["normal"], tile_mask=tile_mask)
                return slam_losses_hip(out, gt_color, gt_depth, color_weight, depth_weight, ssim_weight, add_depth_thres,
                                       render_mask, normal_weight if gt_normal is not None else 0.0, gd["normal"], gt_normal)
            return self.step(loss_fn)
        lib = _lib.load()
        rs = raster_settings
        (N, nf), st, a = self._active(), self.state, self.grad_rows
        dev = st["xyz"]["p"].device
        if N == 0:
            raise RuntimeError("step_slam() on an empty map")
        self.version += 1
        if self._scope == "global":
            self._frozen_version += 1
        if self.world > 1:
            if self._mode == "sharded":
                raise RuntimeError("ShardedMapOptimizer: step_slam() after step() on more than one rank - the two keep "
                                   "different Adam state (replicated vs row-sharded); use one of them per optimizer")
            self._mode = "replicated"
            self._resolve_pending()            # before this step's backward overwrites the arena
            if self._slam_state is None:       # replicated Adam state of the trainable rows (row 0 = first trainable row)
                cap = self.capacity
                self._slam_state = {
                    n: dict(m=torch.zeros(cap, c1 - c0, dtype=torch.float32, device=dev),
                            v=torch.zeros(cap, c1 - c0, dtype=torch.float32, device=dev),
                            ever=torch.zeros(cap, dtype=torch.uint8, device=dev)) for n, c0, c1 in BLOCKS}
        ad = self._slam_state if self.world > 1 else st      # where m / v / ever live
        H, W = int(rs.image_height), int(rs.image_width)
        ws = self._slam_ws
        if ws is None or ws["hw"] != (H, W):
            f = dict(dtype=torch.float32, device=dev)
            i = dict(dtype=torch.int32, device=dev)
            ws = self._slam_ws = dict(
                hw=(H, W), color=torch.empty(3, H, W, **f), depth=torch.empty(1, H, W, **f),
                cidx=torch.empty(1, H, W, **i), didx=torch.empty(1, H, W, **i), cw=torch.empty(1, H, W, **f),
                dw=torch.empty(1, H, W, **f), T=torch.empty(1, H, W, **f), radii=torch.empty(st["xyz"]["p"].shape[0], **i),
                g_color=torch.empty(3, H, W, **f), g_depth=torch.empty(1, H, W, **f), loss=torch.empty(4, **f),
                loss_scratch=None,
                ones=torch.ones((H + 15) // 16, (W + 15) // 16, **i),
                arenas=[_GrowArena(dev), _GrowArena(dev), _GrowArena(dev)])
        act = self.act
        if tile_mask is None:
            tile_mask = ws["ones"]
        tile_mask = tile_mask.to(device=dev, dtype=torch.int32).contiguous()
        gt_color, gt_depth = gt_color.contiguous(), gt_depth.contiguous()
        if normal_weight > 0 and gt_normal is not None:
            # the normal term (mapper.py:433-442) runs between the rasterizer backward and the row exchange / the tail: on
            # several ranks its d_normal rows travel with the other gradient rows (tile bands: each rank owns its pixels)
            gt_normal = gt_normal.to(device=dev, dtype=torch.float32).contiguous()      # [H, W, 3]
        else:
            gt_normal = None
        rm = None if render_mask is None else (render_mask if render_mask.dtype == torch.uint8 else (render_mask != 0).to(torch.uint8)).contiguous()
        need = lib.rtgs_slam_loss_scratch_bytes(H, W, int(rm is None))
        if ws["loss_scratch"] is None or ws["loss_scratch"].numel() < need:
            ws["loss_scratch"] = torch.empty(need, dtype=torch.uint8, device=dev)
        attach = None
        if getattr(self, "attach_init", None) is not None:
            ai = self.attach_init
            attach = _lib.AttachC(ai["xyz"].data_ptr(), ai["raw8"].data_ptr(), ai["info"].data_ptr())
        keep = _Keep(rs, dev)
        self.step_count += 1
        self.total_steps = getattr(self, "total_steps", 0) + 1
        P = lambda t: t.data_ptr()
        geom, binning, img = ws["arenas"]
        args = _lib.MapStepArgsC(
            C.pointer(keep.c), N, 16, P(st["xyz"]["p"]), P(st["shs"]["p"]), P(st["raw8"]["p"]), P(tile_mask), P(gt_color),
            P(gt_depth), _lib.LossCfgC(float(color_weight), float(depth_weight), float(ssim_weight), float(add_depth_thres),
                                       rm.data_ptr() if rm is not None else None),
            P(ws["loss_scratch"]), P(act["opacity"]), P(act["scales"]), P(act["rotations"]),
            P(act["normal"]), P(ws["color"]), P(ws["depth"]), P(ws["cidx"]), P(ws["didx"]), P(ws["cw"]), P(ws["dw"]),
            P(ws["T"]), P(ws["radii"]), P(ws["g_color"]), P(ws["g_depth"]), P(ws["loss"]), P(a.d_means), P(a.d_opac),
            P(a.d_shs), P(a.d_scales), P(a.d_rots), P(a.d_normal), P(a.d_raw8), P(a.scratch), P(a.row_state),
            P(ad["xyz"]["m"]), P(ad["xyz"]["v"]), P(ad["shs"]["m"]), P(ad["shs"]["v"]), P(ad["raw8"]["m"]),
            P(ad["raw8"]["v"]), P(self._lr("xyz")), P(self._lr("shs")), P(self._lr("raw8")), P(ad["xyz"]["ever"]),
            P(ad["shs"]["ever"]), P(ad["raw8"]["ever"]), int(self.step_count), 0.9, 0.999, float(self.eps),
            C.pointer(attach) if attach is not None else None, P(confidence) if confidence is not None else None,
            int(self._act_valid), geom.cb, None, binning.cb, None, img.cb, None,
            float(normal_weight) if gt_normal is not None else 0.0, P(gt_normal) if gt_normal is not None else None,
            int(nf), int(N), int(self.tail_mode), P(self.live_counts) if self.live_counts is not None else None)
        R = C.c_int64(0)
        stream = torch.cuda.current_stream(dev).cuda_stream
        if tile_band and self.world > 1:
            # ONE view split across the ranks: this rank renders, and differentiates, only its band of
            # tiles; the loss normalisers are all-reduced; the partial gradient rows are summed by the same exchange
            if rm is None:
                raise RuntimeError("step_slam(tile_band=True) needs a render_mask (the SSIM window of the unmasked loss "
                                   "crosses band boundaries; the reference always passes one, mapper.py:196-203)")
            self._front_tileband(lib, args, ws, keep, tile_mask, rm, R, dev)
            self._exchange_and_tail(dict(step=int(self.step_count), attach=attach, confidence=confidence, keep=(keep, attach)))
        elif self.world == 1:
            self._mark_adam(0, N - nf)                      # the moments of trainable row t0 + k sit in row k (rtgs_map_step_args)
            with torch.cuda.device(dev):
                rc = lib.rtgs_slam_map_step_ctx(current_context().ptr, C.byref(args), C.byref(R), C.c_void_p(stream))
            _lib.check(rc, "rtgs_slam_map_step")
        else:
            with torch.cuda.device(dev):
                rc = lib.rtgs_slam_map_step_front_ctx(current_context().ptr, C.byref(args), C.byref(R), C.c_void_p(stream))
            _lib.check(rc, "rtgs_slam_map_step_front")
            self._exchange_and_tail(dict(step=int(self.step_count), attach=attach, confidence=confidence, keep=(keep, attach)))
        a.calls = 1
        self._act_valid = True                      # every tail below re-activated the rows it stepped
        self.last_render = (ws["color"], ws["depth"], ws["cidx"], ws["didx"], ws["cw"], ws["dw"], ws["T"])
        self.last_num_rendered = int(R.value)
        self.last_losses = ws["loss"]               # device float[4]: total, colour, depth, ssim (mapper.py:458-466)
        return ws["loss"][0]

    def band_tile_mask(self, tile_mask: torch.Tensor, with_region: bool = False):
        """This rank's share of the switched-on tiles: contiguous runs in row-major order with (almost) equal tile
        counts - computed with device ops only, identically on every rank.  `with_region` also returns the rank's
        REGION: every tile, on or off, belongs to exactly one rank, so that loss pixels inside switched-off tiles (the
        render mask may hold some, mapper.py:500-505) are counted by exactly one rank too."""
        on = (tile_mask.reshape(-1) != 0)
        cum = torch.cumsum(on.to(torch.int64), 0) - on.to(torch.int64)
        owner = ((cum * self.world) // on.sum().clamp_min(1)).clamp_max(self.world - 1)
        mine = owner == self.rank
        band = (on & mine).to(torch.int32).reshape(tile_mask.shape).contiguous()
        return (band, mine.reshape(tile_mask.shape)) if with_region else band

    def _front_tileband(self, lib, args, ws, keep, tile_mask, rm, R, dev):
        from . import _lib
        from .rasterizer import current_context
        N, t0 = self._active()
        H, W = ws["hw"]
        # The band of a mask pair is cached, keyed on the tensor OBJECTS (held here, so their ids cannot be recycled) and
        # their in-place version counters - never on data pointers: the reference picks a random frame of the window per
        # iteration (mapper.py:176-183) and a temporary mask handed in next call may sit at the same address with other
        # content.  A caller that passes fresh tensors every call simply recomputes (a handful of small kernels).
        objs = ws.get("band_objs")
        same = (objs is not None and objs[0] is tile_mask and objs[1] is rm and objs[2] == (tile_mask._version, rm._version))
        if not same:
            band, region = self.band_tile_mask(tile_mask, with_region=True)
            pix = region.repeat_interleave(16, 0).repeat_interleave(16, 1)[:H, :W].to(torch.uint8)
            ws["band"], ws["band_rm"] = band, (pix * (rm != 0).to(torch.uint8)).contiguous()
            ws["band_objs"] = (tile_mask, rm, (tile_mask._version, rm._version))
        band, band_rm = ws["band"], ws["band_rm"]
        V = lambda t: C.c_void_p(t.data_ptr())
        st = lambda: C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        geom, binning, img = ws["arenas"]
        x = self.state
        a = self.grad_rows
        cfg = _lib.LossCfgC(args.loss.color_weight, args.loss.depth_weight, args.loss.ssim_weight, args.loss.add_depth_thres,
                            band_rm.data_ptr())
        ctx = current_context().ptr
        act = self.act
        self._activate_rows(0, N)
        with torch.cuda.device(dev):
            _lib.check(lib.rtgs_raster_forward_ctx(
                ctx, C.byref(keep.c), N, 16, V(x["xyz"]["p"]), V(act["opacity"]), V(x["shs"]["p"]), V(act["scales"]),
                V(act["rotations"]), V(act["normal"]), V(band), V(ws["color"]), V(ws["depth"]), V(ws["cidx"]), V(ws["didx"]),
                V(ws["cw"]), V(ws["dw"]), V(ws["T"]), V(ws["radii"]), geom.cb, None, binning.cb, None, img.cb, None,
                C.byref(R), 0, st()), "rtgs_raster_forward")
            _lib.check(lib.rtgs_slam_loss_sums(V(ws["color"]), V(ws["depth"]), V(ws["didx"]), C.c_void_p(args.gt_color),
                                               C.c_void_p(args.gt_depth), H, W, C.byref(cfg), V(ws["loss_scratch"]), st()),
                       "rtgs_slam_loss_sums")
        sums = ws["loss_scratch"][:32].view(torch.float32)
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=self.group)        # 8 floats: the normalisers are global
        with torch.cuda.device(dev):
            _lib.check(lib.rtgs_slam_loss_grads(V(ws["color"]), V(ws["depth"]), V(ws["didx"]), C.c_void_p(args.gt_color),
                                                C.c_void_p(args.gt_depth), H, W, C.byref(cfg), V(ws["loss_scratch"]),
                                                V(ws["loss"]), V(ws["g_color"]), V(ws["g_depth"]), st()), "rtgs_slam_loss_grads")
            _lib.check(lib.rtgs_raster_backward_range_ctx(
                ctx, C.byref(keep.c), N, 16, R.value, V(x["xyz"]["p"]), V(act["opacity"]), V(x["shs"]["p"]), V(act["scales"]),
                V(act["rotations"]), V(act["normal"]), V(geom.tensor), V(binning.tensor), V(img.tensor), V(ws["color"]), V(ws["T"]),
                V(ws["didx"]), V(ws["g_color"]), V(ws["g_depth"]), V(a.d_means), V(a.d_opac), V(a.d_shs), V(a.d_scales),
                V(a.d_rots), V(a.d_normal), V(a.scratch), V(a.row_state), int(t0), int(N), st()), "rtgs_raster_backward_range")
            if args.normal_weight > 0 and args.gt_normal:
                # the band's pixels only (band_rm: every pixel of the view is counted by exactly one rank); the mean's two
                # sums are all-reduced like the image terms, so that the normaliser is the reference's - the count over the
                # WHOLE image (mapper.py:433-442) - and the partial gradients of the bands add up to the one-rank gradient
                nsum = C.c_void_p(ws["loss_scratch"].data_ptr() + 20)
                _lib.check(lib.rtgs_slam_normal_loss_sums(V(act["normal"]), V(ws["didx"]), C.c_void_p(args.gt_normal), V(band_rm),
                                                          H, W, nsum, None, st()), "rtgs_slam_normal_loss_sums")
        if args.normal_weight > 0 and args.gt_normal:
            nsums = ws["loss_scratch"][20:28].view(torch.float32)
            dist.all_reduce(nsums, op=dist.ReduceOp.SUM, group=self.group)
            with torch.cuda.device(dev):
                _lib.check(lib.rtgs_slam_normal_loss_grads(
                    V(act["normal"]), V(ws["didx"]), C.c_void_p(args.gt_normal), V(band_rm), H, W, float(args.normal_weight),
                    C.c_void_p(ws["loss_scratch"].data_ptr() + 20), V(ws["loss"]), V(a.d_normal), V(a.row_state), None,
                    int(t0), int(N), st()), "rtgs_slam_normal_loss_grads")

    # ------------------------------------------------------------------ sparse exchange (world > 1), no host sync
    def _exchange_and_tail(self, job):
        """All ranks' gradient rows into this rank's arena, summed in rank order, then the Adam tail - every size the
        collective needs is fixed up front (`self._row_capacity` rows per rank, the real count rides in the list's
        header).  If a rank had more rows than fit, a device flag makes EVERY rank skip the apply and the tail of this
        step (all ranks read the same headers); the flag and the counts are copied to pinned host memory without
        waiting, and `_resolve_pending` - run before the next step touches the arena - repeats the exchange with a
        larger capacity.  Steady state: zero host synchronisations per step."""
        from . import _lib
        lib = _lib.load()
        (N, nf), st, a, ws = self._active(), self.state, self.grad_rows, self._slam_ws
        ad = self._slam_state
        dev = st["xyz"]["p"].device
        W, cap = self.world, int(self._row_capacity)
        if ws.get("cap") != cap:
            ws["list"] = torch.empty((1 + cap) * 64, dtype=torch.float32, device=dev)
            ws["gathered"] = torch.empty(W * (1 + cap) * 64, dtype=torch.float32, device=dev)
            ws["flag"] = torch.zeros(2 + W, dtype=torch.int32, device=dev)
            ws["flag_host"] = torch.zeros(2 + W, dtype=torch.int32).pin_memory()
            ws["cap"] = cap
        V = lambda t: C.c_void_p(t.data_ptr())
        arena = (V(a.d_means), V(a.d_shs), V(a.d_opac), V(a.d_scales), V(a.d_rots), V(a.d_normal))
        flag = ws["flag"]
        count_scratch = C.c_void_p(flag.data_ptr() + 4 * (1 + W))
        stream = lambda: C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        with torch.cuda.device(dev):
            _lib.check(lib.rtgs_rows_pack(V(a.row_state), N, *arena, V(ws["list"]), cap, count_scratch, stream()), "rtgs_rows_pack")
        if self.backend == "gloo":
            parts = [torch.empty_like(ws["list"]) for _ in range(W)]
            dist.all_gather(parts, ws["list"], group=self.group)
            ws["gathered"].copy_(torch.cat(parts))
        else:
            dist.all_gather_into_tensor(ws["gathered"], ws["list"], group=self.group)
        stride = (1 + cap) * 64 * 4
        att = job["attach"]
        conf = job["confidence"]
        with torch.cuda.device(dev):
            _lib.check(lib.rtgs_rows_overflow(V(ws["gathered"]), W, cap, V(flag), stream()), "rtgs_rows_overflow")
            own = C.c_void_p(ws["gathered"].data_ptr() + self.rank * stride)
            _lib.check(lib.rtgs_rows_apply(own, cap, 0, *arena, V(a.row_state), V(flag), stream()), "rtgs_rows_apply")
            for r in range(W):                                        # same order on every rank: bit-identical replicas
                lst = C.c_void_p(ws["gathered"].data_ptr() + r * stride)
                _lib.check(lib.rtgs_rows_apply(lst, cap, 1, *arena, V(a.row_state), V(flag), stream()), "rtgs_rows_apply")
            O = lambda t, c: C.c_void_p(t.data_ptr() + 4 * c * nf)          # row nf of a [rows, c] float32 array
            act = self.act
            rc = lib.rtgs_map_tail_rows(
                O(st["xyz"]["p"], 3), O(st["shs"]["p"], 48), O(st["raw8"]["p"], 8), O(a.d_opac, 1), O(a.d_scales, 3), O(a.d_rots, 4),
                O(a.d_normal, 3), O(a.d_means, 3), O(a.d_shs, 48), O(a.d_raw8, 8), C.c_void_p(a.row_state.data_ptr() + nf),
                V(ad["xyz"]["m"]), V(ad["xyz"]["v"]), V(ad["shs"]["m"]), V(ad["shs"]["v"]), V(ad["raw8"]["m"]), V(ad["raw8"]["v"]),
                V(self._lr("xyz")), V(self._lr("shs")), V(self._lr("raw8")), V(ad["xyz"]["ever"]), V(ad["shs"]["ever"]),
                V(ad["raw8"]["ever"]), N - nf, job["step"], 0.9, 0.999, float(self.eps),
                C.byref(att) if att is not None else None, V(conf) if conf is not None else None, V(flag),
                C.byref(_lib.ActivatedC(act["opacity"].data_ptr() + 4 * nf, act["scales"].data_ptr() + 12 * nf,
                                        act["rotations"].data_ptr() + 16 * nf, act["normal"].data_ptr() + 12 * nf)), stream())
            _lib.check(rc, "rtgs_map_tail_rows")
        ws["flag_host"].copy_(flag, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        self._pending = dict(event=ev, job=job)

    def _resolve_pending(self):
        """Did the last exchange overflow?  (Its flag was copied to pinned memory a whole step ago.)  If so nothing was
        applied anywhere: enlarge the capacity - every rank computes the same value from the same counts - and repeat."""
        while self._pending is not None:
            pend, self._pending = self._pending, None
            pend["event"].synchronize()
            host = self._slam_ws["flag_host"]
            need = int(host[1:1 + self.world].max())
            if int(host[0]) == 0:
                # The all-gather moves `capacity` rows per rank whatever the lists hold: when the fullest list of the last
                # `_shrink_every` exchanges used less than a quarter of it, halve it (never below twice that peak).  Every
                # rank reads the same counts, so every rank takes the same decision at the same step.
                self._cap_peak = max(self._cap_peak, need)
                self._cap_steps += 1
                if self._cap_steps >= self._shrink_every:
                    if 4 * self._cap_peak <= self._row_capacity and self._row_capacity > 1024:
                        self._row_capacity = max(1024, self._row_capacity // 2)
                    self._cap_peak, self._cap_steps = 0, 0
                return
            cap = 1024
            while cap < 2 * need:
                cap *= 2
            self._row_capacity = cap
            self._cap_peak, self._cap_steps = 0, 0
            self.overflow_redos += 1
            self._exchange_and_tail(pend["job"])

    def flush(self):
        """Make the parameters final: resolves a pending (possibly overflowed) multi-GPU exchange."""
        if self.world > 1:
            self._resolve_pending()

    def _arena_grad(self, name):
        a = self.grad_rows
        return dict(xyz=a.d_means, shs=a.d_shs, raw8=a.d_raw8)[name]

    def my_rows(self) -> slice:
        """This rank's shard of the trainable rows."""
        t0 = self._active()[1]
        return slice(t0 + self.rank * self.per, t0 + (self.rank + 1) * self.per)

    def step(self, loss_fn: Callable[[Dict[str, torch.Tensor]], torch.Tensor]) -> torch.Tensor:
        """loss_fn(gaussian_data) -> scalar loss of THIS rank's view.  Gradients are summed over
        ranks (the sum of per-view losses is what a single GPU looping over the views optimises).  Only the trainable
        rows [n_frozen, N) are reduced, stepped and gathered; the row shards partition that range."""
        self._clean()
        if self._scope == "global":
            self._frozen_version += 1
        N, nf = self._active()
        per, span = self.per, self.per * self.world        # rows of one shard / of all shards (>= n_train: padded)
        self._act_valid = False            # raw8 moves without step_slam's tail: its activated copies go stale
        self.version += 1
        marks = getattr(self, "phase_marks", None)          # measurement aid (bench.py config5): events at the phase borders

        def mark(name):
            if marks is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(torch.cuda.current_stream(self.device))
                marks.append((name, ev))
        mark("begin")
        if self.world > 1:
            if self._mode == "replicated":
                raise RuntimeError("ShardedMapOptimizer: step() after step_slam() on more than one rank - the two keep "
                                   "different Adam state (row-sharded vs replicated); use one of them per optimizer")
            self._mode = "sharded"
        leaves = {n: self.state[n]["p"][:N].detach().requires_grad_(True) for n, _, _ in BLOCKS}
        arena = self.grad_rows
        if arena is not None:
            arena.begin_step()
            gd = activate8_hip(leaves["raw8"], arena)
            gd["grad_rows"] = arena
        else:
            gd = self.activate_fn(leaves["raw8"])
        gd["xyz"] = leaves["xyz"]
        gd["shs"] = leaves["shs"].view(N, 16, 3)
        gd["raw8"] = leaves["raw8"]          # the raw leaf itself, for loss terms on raw values (the attach regulariser)
        loss = loss_fn(gd)
        mark("forward_and_loss")
        grads = torch.autograd.grad(loss, [leaves[n] for n, _, _ in BLOCKS], allow_unused=True)
        mark("backward")
        self.step_count += 1
        self.total_steps = getattr(self, "total_steps", 0) + 1
        rows = self.my_rows()
        gmap = {}
        for (name, _, _), g in zip(BLOCKS, grads):
            gmap[name] = (torch.zeros_like(leaves[name]) if g is None else g).contiguous()

        if self.world > 1 and self.backend != "gloo":
            # RCCL path.  All three reduce-scatters are queued first (small tensors first), so Adam on the xyz /
            # raw8 shards and their all-gathers run under the SH reduce-scatter, which carries 81 % of the bytes.
            order = ("xyz", "raw8", "shs")
            rs, ag = {}, []
            for name in order:
                st = self.state[name]
                g = gmap[name]
                if nf + span == N:
                    src = g[nf:N]                       # the trainable gradient rows as they lie (no staging copy)
                else:
                    st["gpad"][nf:N] = g[nf:N]
                    st["gpad"][N:nf + span].zero_()     # the shards' padding rows: zero gradient (in a global optimisation they
                    src = st["gpad"][nf:nf + span]      # are real, unstable rows: zero gradient + zero moments leave them as they are)
                rs[name] = dist.reduce_scatter_tensor(st["gshard"][:per], src, op=dist.ReduceOp.SUM, group=self.group,
                                                      async_op=True)
            for name in order:
                st = self.state[name]
                rs[name].wait()                         # stream-side wait, the host does not block
                shard = st["p"][rows]
                mark("adam_begin")
                self._adam(name, shard, st["gshard"][:per])
                mark("adam_end")
                # in place: the shard IS this rank's slice of the gathered range (sendbuff = recvbuff + rank * count, the
                # in-place form of the collective) - no staging copy of the updated rows
                ag.append(dist.all_gather_into_tensor(st["p"][nf:nf + span], shard, group=self.group, async_op=True))
            for w in ag:
                w.wait()
            mark("end")
            return loss.detach()

        for name, _, _ in BLOCKS:
            st = self.state[name]
            g = gmap[name]
            if self.world > 1:                          # gloo (CPU tests): no reduce-scatter -> all-reduce, take the local rows
                st["gpad"][nf:N] = g[nf:N]
                st["gpad"][N:nf + span].zero_()
                red = st["gpad"][nf:nf + span]
                dist.all_reduce(red, op=dist.ReduceOp.SUM, group=self.group)
                gs = st["gpad"][rows].contiguous()
            else:
                gs = g[nf:N]
            shard = st["p"][rows]
            row_state = None
            if arena is not None and arena.calls == 1 and gs.data_ptr() == self._arena_grad(name)[nf:].data_ptr():
                row_state = arena.row_state[nf:]   # gs IS the rasterizer's persistent rows: their states are exact
            mark("adam_begin")
            self._adam(name, shard, gs, row_state)
            mark("adam_end")
            if self.world > 1:
                parts = [torch.empty_like(shard) for _ in range(self.world)]
                dist.all_gather(parts, shard.clone(), group=self.group)
                st["p"][nf:nf + span].copy_(torch.cat(parts, dim=0))
        mark("end")
        return loss.detach()


def normal_loss_term(render, normal_w: torch.Tensor, gt_normal: torch.Tensor, render_mask: Optional[torch.Tensor] = None):
    """The normal term of Mapping.loss_update (mapper.py:433-442; `normal_weight`, 0 in every shipped config):
    mean over {mask & depth_index != -1 & gt normal not all-zero} of 1 - cos(render normal, gt normal), where the
    render normal of a pixel is the world normal of the Gaussian that owns its depth (render.py:130-133).  `render` =
    the rasterizer's tuple, `normal_w` [N,3] the `normal` of gaussian_data (receives the gradient through the gather
    kernel), `gt_normal` [H,W,3].  An empty set gives 0 (nan in the reference).  HIP tensors only."""
    from .render import gather_normal_map
    didx = render[3]
    rn = gather_normal_map(normal_w, didx).permute(1, 2, 0)
    cos_dist = 1 - torch.nn.functional.cosine_similarity(rn, gt_normal, dim=-1)
    valid = (didx[0] != -1) & ~(gt_normal == 0).all(dim=-1)
    if render_mask is not None:
        valid = valid & (render_mask != 0)
    v = valid.to(cos_dist.dtype)
    return (cos_dist * v).sum() / v.sum().clamp_min(1.0)


class _SlamLossHip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, color, depth, didx, gt_color, gt_depth, cw, dw, sw, thr, render_mask):
        from . import _lib
        lib = _lib.load()
        if not color.is_cuda:
            raise RuntimeError("rtg_slam_amd.map_optim: slam_losses_hip needs HIP device tensors; no CPU path.")
        dev = color.device
        H, W = int(color.shape[1]), int(color.shape[2])
        color, depth, didx = color.contiguous(), depth.contiguous(), didx.contiguous()
        gt_color, gt_depth = gt_color.contiguous(), gt_depth.contiguous()
        rm = None if render_mask is None else (render_mask != 0).to(torch.uint8).contiguous()
        loss4 = torch.empty(4, dtype=torch.float32, device=dev)
        scratch = torch.empty(lib.rtgs_slam_loss_scratch_bytes(H, W, int(rm is None)), dtype=torch.uint8, device=dev)
        g_c, g_d = torch.empty_like(color), torch.empty_like(depth)
        cfg = _lib.LossCfgC(float(cw), float(dw), float(sw), float(thr), rm.data_ptr() if rm is not None else None)
        stream = torch.cuda.current_stream(dev).cuda_stream
        P = lambda t: C.c_void_p(t.data_ptr())
        with torch.cuda.device(dev):
            rc = lib.rtgs_slam_loss(P(color), P(depth), P(didx), P(gt_color), P(gt_depth), H, W, C.byref(cfg), P(scratch),
                                    P(loss4), P(g_c), P(g_d), C.c_void_p(stream))
        _lib.check(rc, "rtgs_slam_loss")
        ctx.save_for_backward(g_c, g_d)
        ctx.terms = loss4                      # [total, colour, depth, ssim] for reporting (mapper.py:458-466)
        return loss4[0]

    @staticmethod
    def backward(ctx, g):
        g_c, g_d = ctx.saved_tensors
        return g_c * g, g_d * g, None, None, None, None, None, None, None, None


def slam_losses_hip(render, gt_color, gt_depth, color_weight: float = 0.8, depth_weight: float = 1.0,
                    ssim_weight: float = 0.2, add_depth_thres: float = 0.1, render_mask=None, normal_weight: float = 0.0,
                    normal_w=None, gt_normal=None) -> torch.Tensor:
    """Same loss as `slam_losses`: value and both image gradients from the fused HIP kernels (rtgs_slam_loss); the
    normal term (off in every shipped config) is added through the gather kernel when `normal_weight > 0`."""
    total = _SlamLossHip.apply(render[0], render[1], render[3], gt_color, gt_depth, color_weight, depth_weight, ssim_weight,
                               add_depth_thres, render_mask)
    if normal_weight > 0:
        total = total + normal_weight * normal_loss_term(render, normal_w, gt_normal, render_mask)
    return total
