"""Mirror of ``diff_gaussian_rasterization`` (forward only) on the HIP rasterizer.

The package is NOT part of the GaussReg tree; the API below is the public
upstream one (graphdeco-inria/diff-gaussian-rasterization, diff_gaussian_rasterization/__init__.py):

    GaussianRasterizationSettings(image_height, image_width, tanfovx, tanfovy, bg, scale_modifier,
                                  viewmatrix, projmatrix, sh_degree, campos, prefiltered, debug)
    GaussianRasterizer(raster_settings).forward(means3D, means2D, opacities, shs=None,
        colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None) -> (color, radii)
    GaussianRasterizer.markVisible(positions) -> BoolTensor

plus `rasterize_views(...)`: many cameras over one Gaussian set in one launch sequence (the
throughput path: per-Gaussian inputs are read once per batch).  Forward only: outputs carry no
autograd graph (fine registration in GaussReg only renders).
"""
import ctypes
import os
import threading
from typing import NamedTuple, Optional, Sequence

import torch

from . import _lib


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


FAST_EXP = 1  # GR_RASTER_FAST_EXP (include/gaussreg_hip.h)
SPLIT = 2     # GR_RASTER_SPLIT
SHARE = 4     # GR_RASTER_SHARE
_ENV_FAST = None
_bin_hint = {}  # (device, P, V, W, H) -> (bytes of the binning buffer, largest chunk) the last call of that shape needed


def _flags(fast_exp):
    """fast_exp None: the library default (environment variable GR_RASTER_FAST_EXP=1 switches it on)."""
    global _ENV_FAST
    if fast_exp is None:
        if _ENV_FAST is None:
            import os
            _ENV_FAST = os.environ.get("GR_RASTER_FAST_EXP", "0")[:1] == "1"
        fast_exp = _ENV_FAST
    return FAST_EXP if fast_exp else 0


def _view_struct(rs: GaussianRasterizationSettings) -> _lib.RasterView:
    v = _lib.RasterView()
    v.image_height, v.image_width = int(rs.image_height), int(rs.image_width)
    v.tanfovx, v.tanfovy = float(rs.tanfovx), float(rs.tanfovy)
    v.scale_modifier = float(rs.scale_modifier)
    v.sh_degree = int(rs.sh_degree)
    v.prefiltered, v.debug = int(bool(rs.prefiltered)), int(bool(rs.debug))
    bg = rs.bg.detach().to("cpu", torch.float32).reshape(-1).tolist()
    vm = rs.viewmatrix.detach().to("cpu", torch.float32).reshape(-1).tolist()
    pm = rs.projmatrix.detach().to("cpu", torch.float32).reshape(-1).tolist()
    cp = rs.campos.detach().to("cpu", torch.float32).reshape(-1).tolist()
    if len(bg) != 3 or len(vm) != 16 or len(pm) != 16 or len(cp) != 3:
        raise ValueError("bg/campos must have 3 elements, viewmatrix/projmatrix 16")
    v.bg[:] = bg
    v.viewmatrix[:] = vm
    v.projmatrix[:] = pm
    v.campos[:] = cp
    return v


class ViewBatch:
    """Cameras of one batched call, already marshalled into the C struct array (build once, reuse:
    marshalling V settings costs ~40 us of Python each)."""

    def __init__(self, settings: Sequence[GaussianRasterizationSettings]):
        if len(settings) < 1:
            raise ValueError("need at least one view")
        self.count = len(settings)
        self.height, self.width = int(settings[0].image_height), int(settings[0].image_width)
        self.array = (_lib.RasterView * self.count)(*[_view_struct(s) for s in settings])


def _dev_f32(t: Optional[torch.Tensor], dev, name):
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a tensor")
    if t.device == dev and t.dtype == torch.float32 and t.is_contiguous():
        return t  # only its pointer is read
    return t.detach().to(device=dev, dtype=torch.float32).contiguous()


class _FramePipe:
    """Two rasterizer calls in flight (per device and host thread).

    A one-camera frame is a chain of ~15 dependent launches, most of them a few microseconds of work: alone on a stream the
    chip idles through every hand-over.  Consecutive calls therefore run on two internal streams in turn, so the front of
    frame n+1 (preprocess, depth sort, tile counts) fills the gaps of frame n's scatter and blend (+19 % views/s).  Batched
    calls (32 views per call) gain too, for a different reason: the memory-bound sort and binning of call n+1 run next to
    the instruction-bound blend of call n (+5.6 %).  For the caller nothing changes: the outputs are joined into the
    caller's current stream before the call returns.

    Input readiness: a frame may only start when its input tensors are complete on the caller's stream.  In general that
    is `side.wait_stream(current)` -- which, after the previous frame was joined into `current`, also waits for that
    frame and serialises everything.  When the inputs are THE SAME tensors with the same version counters as in the
    previous frame (a static scene rendered from moving cameras), they were already complete at the previous call, and
    the frame waits only for the event recorded on the caller's stream at the FIRST call with these inputs.  The previous
    frame's inputs are kept referenced until the next call so that their addresses cannot be handed to other tensors in
    between (the stamp compares storage address + version).  An in-place update by the caller is ordered behind every frame
    that read the old values: each call joins its frame into the caller's stream before it returns.

    OPT-IN (`static_scene=True` on rasterize_views / GaussianRasterizer, or GR_RASTER_PIPELINE=1): the stamp only sees
    writes that bump a tensor's version counter.  `tensor.data.<op>_()` (`.data` carries its own counter), a custom
    extension kernel or any other library writing through `data_ptr()` change the scene without changing the stamp; the
    next frame would then start on the side stream next to that write and could render stale or torn values.  The
    default therefore is no pipe: every call is ordered on the caller's stream like upstream's.  Callers that opt in and
    do write behind the counter's back call reset_frame_pipe() after the write."""

    def __init__(self, dev):
        self.streams = tuple(torch.cuda.Stream(dev) for _ in range(2))
        self.turn = 0
        self.stamp = None
        self.keep = None
        self.caller = None
        self.ready = None  # recorded on the caller's stream at the first call with this stamp: the inputs were complete there
        self.ptrs = None   # the marshalled input pointers of the scene with this stamp (set by rasterize_views)
        self.overlapping = False

    def begin(self, dev, inputs):
        cur = torch.cuda.current_stream(dev)
        side = self.streams[self.turn]
        stamp = _lib.tensor_stamp(inputs)
        self.overlapping = False
        if stamp is not None and stamp == self.stamp and self.caller == cur.cuda_stream and self.ready is not None:
            side.wait_event(self.ready)
            self.overlapping = True   # this call runs next to the previous one
        else:
            side.wait_stream(cur)
            self.ptrs = None
            self.ready = None
            if stamp is not None:
                self.ready = torch.cuda.Event()
                self.ready.record(cur)
        self.stamp, self.keep, self.caller = stamp, inputs, cur.cuda_stream
        return cur, side

    def abort(self, cur, side):
        self.stamp = self.keep = self.ready = self.ptrs = None
        cur.wait_stream(side)

    def end(self, cur, side, outputs):
        self.turn = (self.turn + 1) % len(self.streams)
        cur.wait_stream(side)
        for t in outputs:
            t.record_stream(cur)


_pipes = threading.local()  # per host thread (the library's split-call state is per thread too) and device


def _frame_pipe(dev, static_scene):
    """The pipe is opt-in (see _FramePipe): `static_scene=True` per call, or GR_RASTER_PIPELINE=1 for the process
    (GR_RASTER_PIPELINE=0 wins over the argument: the switch tests and benches use to get the serial path)."""
    env = os.environ.get("GR_RASTER_PIPELINE")
    if env == "0" or not (static_scene or env == "1"):
        return None
    table = getattr(_pipes, "table", None)
    if table is None:
        table = _pipes.table = {}
    p = table.get(dev.index)
    if p is None:
        p = table[dev.index] = _FramePipe(dev)
    return p


_geom_bytes = {}


def reset_frame_pipe():
    """Drop this thread's frame pipes: the references they keep to the last frame's input tensors (see _FramePipe) and the
    readiness events.  The next one-camera call starts with a full wait on the caller's stream."""
    table = getattr(_pipes, "table", None)
    if table:
        for p in table.values():
            p.stamp = p.keep = p.ready = p.ptrs = None


def rasterize_views(settings, means3D, opacities, shs=None,
                    colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None, fast_exp=None, _one=False,
                    static_scene=False):
    """Render the same Gaussians from len(settings) cameras (`settings`: a sequence of
    GaussianRasterizationSettings, or a prebuilt ViewBatch).

    Returns (color (V,3,H,W) f32, radii (V,P) i32, num_rendered list[int]).  num_rendered = (tile, Gaussian)
    instances actually binned per view: at most the reference's count (pairs that cannot reach alpha = 1/255
    anywhere in the tile are dropped before the sort; the image is unaffected).

    `fast_exp=True`: the blend uses the hardware exponential (v_exp_f32) instead of the deterministic polynomial of the
    oracle -- the image is within 1e-5 relative of the bit-exact one (default False: bit-exact).
    `static_scene=True` (extension): consecutive calls over the same, unmodified scene tensors overlap on two internal
    streams (_FramePipe: read its contract first -- only writes that bump the tensors' version counters are seen).
    (`_one`: internal, one camera -- the outputs come back as (3,H,W) and (P,), no view ops on the way out.)"""
    dev = means3D.device if means3D.is_cuda else _lib.require_gpu()
    L = _lib.lib()
    n_pts = int(means3D.shape[0])

    def given(t):  # upstream passes empty tensors for "not provided"; with zero Gaussians everything is empty
        return t is not None and (t.numel() > 0 or n_pts == 0)

    if given(shs) == given(colors_precomp):
        raise Exception('Please provide excatly one of either SHs or precomputed colors!')
    has_sr = given(scales) and given(rotations)
    has_cov = given(cov3D_precomp)
    if has_sr == has_cov or (given(scales) != given(rotations)):
        raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
    vb = settings if isinstance(settings, ViewBatch) else ViewBatch(settings)
    V, views, H, W = vb.count, vb.array, vb.height, vb.width
    m = _dev_f32(means3D, dev, "means3D")
    if m.dim() != 2 or m.shape[1] != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    P = m.shape[0]
    op = _dev_f32(opacities, dev, "opacities")
    sh = _dev_f32(shs, dev, "shs") if given(shs) else None
    cp = _dev_f32(colors_precomp, dev, "colors_precomp") if given(colors_precomp) else None
    sc = _dev_f32(scales, dev, "scales") if has_sr else None
    rot = _dev_f32(rotations, dev, "rotations") if has_sr else None
    cov = _dev_f32(cov3D_precomp, dev, "cov3D_precomp") if has_cov else None
    M = 0 if sh is None else (sh.shape[1] if sh.dim() == 3 else sh.reshape(max(P, 1), -1, 3).shape[1])
    nr = (ctypes.c_int64 * (V + 1))()
    pipe = _frame_pipe(dev, static_scene)
    cur = side = None
    # (explicit set_device / set_stream instead of the context managers: a one-camera frame is ~0.2 ms and every
    # microsecond of Python between two library calls is on the critical path)
    home = torch.cuda.current_device()
    if home != dev.index:
        torch.cuda.set_device(dev)
    try:
        if pipe is not None:
            cur, side = pipe.begin(dev, tuple(t for t in (m, op, sh, cp, sc, rot, cov) if t is not None))
            torch.cuda.set_stream(side)  # allocations below come from the side stream's pool
            st = ctypes.c_void_p(side.cuda_stream)
        else:
            st = _lib.stream_ptr(dev)
        one = _one and V == 1
        color = torch.empty((3, H, W) if one else (V, 3, H, W), dtype=torch.float32, device=dev)
        radii = torch.empty((P,) if one else (V, P), dtype=torch.int32, device=dev)  # every entry is written by preprocess
        gkey = (P, V, W, H)
        gbytes = _geom_bytes.get(gkey)
        if gbytes is None:
            if len(_geom_bytes) > 64:
                _geom_bytes.clear()
            gbytes = _geom_bytes[gkey] = L.gr_raster_geom_bytes(P, V, W, H) + 256
        geom = torch.empty(gbytes, dtype=torch.uint8, device=dev)
        flags = _flags(fast_exp)
        # the binning buffer is sized from the last call of this shape (+ 25 %): the library is entered once per frame, and
        # only a frame that needs more comes back for a larger buffer
        key = (dev.index, P, V, W, H)
        hint, chunk_hint = _bin_hint.get(key, (0, 0))
        binb = torch.empty(hint + 256, dtype=torch.uint8, device=dev) if hint else None
        nr[V] = chunk_hint  # in: sizes the scatter's staging block of the speculative launch; out: this call's figure

        # (the inputs' pointers are marshalled once per scene: the pipe keeps them while the stamp stays the same)
        if pipe is not None and pipe.ptrs is not None:
            ptrs = pipe.ptrs
        else:
            ptrs = (_lib.ptr(m), _lib.ptr(sh), _lib.ptr(cp), _lib.ptr(op), _lib.ptr(sc), _lib.ptr(rot), _lib.ptr(cov))
            if pipe is not None:
                pipe.ptrs = ptrs

        def forward(fl):
            return L.gr_raster_forward(P, M, *ptrs, views, V, _lib.ptr(radii), _lib.ptr(geom), gbytes, _lib.ptr(binb),
                                       hint + 256 if binb is not None else 0, _lib.ptr(color), fl, nr, st)
        # with the pipe: the library returns as soon as the frame is enqueued, and the stream joins below run while the GPU
        # works through the front of the frame; gr_raster_forward_finish then waits for the instance counts
        rc = forward(flags | ((SPLIT | (SHARE if pipe.overlapping and V == 1 else 0)) if pipe is not None else 0))
        joined = rejoin = False
        if rc == _lib.GR_PENDING:
            try:
                torch.cuda.set_stream(cur)
                pipe.end(cur, side, (color, radii))
                joined = True
            finally:
                rc = L.gr_raster_forward_finish(nr)
            if rc in (_lib.GR_RETRY_BIN, _lib.GR_RETRY_FULL):  # rare: more work for this frame on the side stream
                torch.cuda.set_stream(side)
                rejoin = True
            if rc == _lib.GR_RETRY_FULL:
                rc = forward(flags)
        _lib.check(rc, allow=(_lib.GR_RETRY_BIN,))
        total = sum(int(nr[v]) for v in range(V))
        need = L.gr_raster_bin_bytes(total, W, H, V)
        if rc == _lib.GR_RETRY_BIN:
            binb = torch.empty(need + 256, dtype=torch.uint8, device=dev)
            _lib.check(L.gr_raster_render_ex(P, views, V, nr, _lib.ptr(geom), geom.numel(), _lib.ptr(binb), binb.numel(),
                                             _lib.ptr(color), flags, st))
        if len(_bin_hint) > 64:
            _bin_hint.clear()
        _bin_hint[key] = (need + need // 4 + 1024, int(nr[V]))
        if pipe is not None:
            torch.cuda.set_stream(cur)
            if not joined:
                pipe.end(cur, side, (color, radii))
            elif rejoin:
                cur.wait_stream(side)
    except BaseException:
        if pipe is not None and cur is not None:
            torch.cuda.set_stream(cur)
            pipe.abort(cur, side)
        raise
    finally:
        if home != dev.index:
            torch.cuda.set_device(home)
    return color, radii, [int(nr[v]) for v in range(V)]


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings, fast_exp=None, static_scene=False):
    """`raster_settings`: GaussianRasterizationSettings, or a one-camera ViewBatch built from it (marshalled once)."""
    vb = raster_settings if isinstance(raster_settings, ViewBatch) else [raster_settings]
    color, radii, _ = rasterize_views(vb, means3D, opacities, sh, colors_precomp, scales, rotations, cov3Ds_precomp,
                                      fast_exp=fast_exp, _one=True, static_scene=static_scene)
    return color, radii


class GaussianRasterizer(torch.nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings, fast_exp=None, static_scene=False):
        """`fast_exp`, `static_scene` (extensions, upstream has no such arguments): see rasterize_views."""
        super().__init__()
        self.raster_settings = raster_settings
        self.fast_exp = fast_exp
        self.static_scene = static_scene
        self._view_batch = None  # (settings object, tensor versions, ViewBatch): the C camera struct is built once

    @staticmethod
    def _stamp(rs):
        # the camera tensors may be updated in place between frames (pose optimisation): their storage and version
        # counters are part of the cache key, so a stale marshalled copy is never rendered
        # (None for inference tensors, which carry no version counter: nothing is cached for them)
        return _lib.tensor_stamp((rs.viewmatrix, rs.projmatrix, rs.campos, rs.bg))

    def _views(self):
        rs = self.raster_settings
        stamp = self._stamp(rs)
        if stamp is None:
            self._view_batch = None
            return ViewBatch([rs])
        if self._view_batch is None or self._view_batch[0] is not rs or self._view_batch[1] != stamp:
            self._view_batch = (rs, stamp, ViewBatch([rs]))
        return self._view_batch[2]

    def markVisible(self, positions):
        with torch.no_grad():
            dev = _lib.require_gpu()
            L = _lib.lib()
            p = positions.detach().to(device=positions.device if positions.is_cuda else dev,
                                      dtype=torch.float32).contiguous()
            present = torch.zeros((p.shape[0],), dtype=torch.uint8, device=p.device)
            vm = (ctypes.c_float * 16)(*self.raster_settings.viewmatrix.detach().to("cpu", torch.float32)
                                       .reshape(-1).tolist())
            with torch.cuda.device(p.device):
                _lib.check(L.gr_raster_mark_visible(p.shape[0], _lib.ptr(p), vm, _lib.ptr(present),
                                                    _lib.stream_ptr(p.device)))
            return present.bool()

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        rs = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                   cov3D_precomp, self._views(), fast_exp=self.fast_exp, static_scene=self.static_scene)
