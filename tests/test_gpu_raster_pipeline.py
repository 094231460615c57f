"""One-camera forward() calls run two frames in flight on internal streams (gaussreg_amd/rasterizer.py _FramePipe).  What
must hold: every frame is bit-equal to the same call with the pipe switched off, an in-place update of a scene tensor
between two calls is rendered, a scene replaced by new tensors is rendered, and a caller on its own stream gets its
outputs ordered on that stream."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _pipe_on(monkeypatch):
    # the pipe is opt-in (static_scene=True / GR_RASTER_PIPELINE=1); "0" inside a test selects the serial path
    monkeypatch.setenv("GR_RASTER_PIPELINE", "1")


def _setup(P=60000, W=320, H=192, V=3, seed=11):
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from helpers import raster_scene
    g, cams = raster_scene(P, W, H, seed=seed, V=V)
    t = {k: torch.from_numpy(v).cuda() for k, v in g.items()}
    rast = [GaussianRasterizer(GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=c["tanfovx"], tanfovy=c["tanfovy"], bg=torch.tensor([0.1, 0.2, 0.3]).cuda(),
        scale_modifier=1.0, viewmatrix=torch.from_numpy(c["viewmatrix"]).cuda(), projmatrix=torch.from_numpy(c["projmatrix"]).cuda(),
        sh_degree=3, campos=torch.from_numpy(c["campos"]).cuda(), prefiltered=False, debug=False)) for c in cams]
    return t, rast


def _render(r, t):
    return r(means3D=t["means3D"], means2D=None, opacities=t["opacities"], shs=t["shs"], scales=t["scales"], rotations=t["rotations"])


def _reference(rast, t, monkeypatch):
    with monkeypatch.context() as mp:
        mp.setenv("GR_RASTER_PIPELINE", "0")
        out = [tuple(x.clone() for x in _render(r, t)) for r in rast]
    torch.cuda.synchronize()
    return out


def test_frames_in_flight_equal_the_serial_frames(monkeypatch):
    t, rast = _setup()
    want = _reference(rast, t, monkeypatch)
    assert not torch.equal(want[0][0], want[1][0])
    # outputs are kept and checked at the end: nothing synchronises between the calls
    got = [_render(rast[i % len(rast)], t) for i in range(12)]
    for i, (img, radii) in enumerate(got):
        assert torch.equal(img, want[i % len(rast)][0]) and torch.equal(radii, want[i % len(rast)][1]), i


def test_in_place_update_and_replaced_scene_are_rendered(monkeypatch):
    t, rast = _setup(seed=12)
    a0 = _render(rast[0], t)[0]
    a1 = _render(rast[1], t)[0]
    # in place, on the caller's stream, right after two frames were queued
    t["means3D"].add_(torch.tensor([0.05, -0.02, 0.03], device="cuda"))
    t["opacities"].mul_(0.9)
    b0 = _render(rast[0], t)[0]
    b1 = _render(rast[1], t)[0]
    # new tensors (new storage, version counters start again)
    t2 = {k: (v * 1.0) for k, v in t.items()}
    t2["shs"] = t2["shs"] * 0.5
    c0 = _render(rast[0], t2)[0]
    c2 = _render(rast[2], t2)[0]
    t_now = {k: v.clone() for k, v in t.items()}
    want_b = _reference(rast, t_now, monkeypatch)
    want_c = _reference(rast, t2, monkeypatch)
    assert torch.equal(b0, want_b[0][0]) and torch.equal(b1, want_b[1][0])
    assert torch.equal(c0, want_c[0][0]) and torch.equal(c2, want_c[2][0])
    assert not torch.equal(a0, b0) and not torch.equal(a1, b1) and not torch.equal(b0, c0)


def test_caller_on_its_own_stream(monkeypatch):
    t, rast = _setup(seed=13)
    want = _reference(rast, t, monkeypatch)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    sums = []
    with torch.cuda.stream(s):
        scene = {k: v + 0.0 for k, v in t.items()}      # produced on s, consumed by the frames
        for i in range(6):
            img, radii = _render(rast[i % 3], scene)
            sums.append((img.double().sum(), radii.long().sum()))  # consumers on s, straight after the call
    s.synchronize()
    for i, (a, b) in enumerate(sums):
        assert float(a) == float(want[i % 3][0].double().sum()) and int(b) == int(want[i % 3][1].long().sum()), i


def test_frames_that_outgrow_the_speculative_list(monkeypatch):
    """A narrow camera (few instances) followed by wide ones: the list sized from the previous frame is too small, the
    split call reports GR_RETRY_BIN and the frame is drawn again inside the pipe."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from gaussreg_amd import synthetic
    W, H, P = 320, 192, 60000
    g = synthetic.gaussians_c2(P, 14, sh_degree=3)
    t = {k: torch.from_numpy(v).cuda() for k, v in g.items()}
    cams = [synthetic.camera(W, H, fovx_deg=4.0), synthetic.camera(W, H, fovx_deg=75.0),
            synthetic.camera(W, H, fovx_deg=60.0, R_c2w=synthetic.rot_yx(0.2, -0.1), C=np.array([0.2, 0.1, -0.1]))]
    rast = [GaussianRasterizer(GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=c["tanfovx"], tanfovy=c["tanfovy"], bg=torch.zeros(3).cuda(), scale_modifier=1.0,
        viewmatrix=torch.from_numpy(c["viewmatrix"]).cuda(), projmatrix=torch.from_numpy(c["projmatrix"]).cuda(), sh_degree=3,
        campos=torch.from_numpy(c["campos"]).cuda(), prefiltered=False, debug=False)) for c in cams]
    want = _reference(rast, t, monkeypatch)
    order = [0, 0, 1, 0, 2, 1, 0, 0, 2, 2, 1]
    got = [_render(rast[i], t) for i in order]
    for k, i in enumerate(order):
        assert torch.equal(got[k][0], want[i][0]) and torch.equal(got[k][1], want[i][1]), (k, i)
    assert int((want[0][1] > 0).sum()) * 4 < int((want[1][1] > 0).sum())  # the narrow view really sees far fewer Gaussians


def test_two_host_threads_render_side_by_side(monkeypatch):
    """The frame pipe (and the library's split-call state) is per host thread: two threads rendering the same scene through
    their own rasterizer objects get the serial frames."""
    import threading
    t, rast = _setup(seed=15)
    want = _reference(rast, t, monkeypatch)
    out, err = {}, []

    def work(k):
        try:
            frames = [_render(rast[(i + k) % 3], t) for i in range(8)]
            torch.cuda.synchronize()
            out[k] = frames
        except Exception as e:  # pragma: no cover
            err.append(e)
    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not err, err
    for k in range(2):
        for i, (img, radii) in enumerate(out[k]):
            assert torch.equal(img, want[(i + k) % 3][0]) and torch.equal(radii, want[(i + k) % 3][1]), (k, i)


def test_reset_frame_pipe_releases_the_inputs(monkeypatch):
    import gc, weakref
    from gaussreg_amd import rasterizer
    t, rast = _setup(seed=16)
    want = _reference(rast, t, monkeypatch)
    scene = {k: v.clone() for k, v in t.items()}
    ref = weakref.ref(scene["means3D"])
    a = _render(rast[0], scene)[0]
    del scene
    gc.collect()
    assert ref() is not None          # the pipe still holds the last frame's inputs
    rasterizer.reset_frame_pipe()
    gc.collect()
    assert ref() is None
    b = _render(rast[1], t)[0]
    assert torch.equal(a, want[0][0]) and torch.equal(b, want[1][0])


def test_batched_calls_in_flight_equal_the_serial_calls(monkeypatch):
    """rasterize_views with several cameras per call goes through the same pipe (and the library's count read-back in the
    middle of every call)."""
    from gaussreg_amd.rasterizer import rasterize_views
    t, rast = _setup(seed=17, V=4)
    sets = [r.raster_settings for r in rast]
    groups = [sets[:3], sets[1:], sets[:2] + sets[3:]]

    def render(g):
        return rasterize_views(g, t["means3D"], t["opacities"], shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
    with monkeypatch.context() as mp:
        mp.setenv("GR_RASTER_PIPELINE", "0")
        want = [tuple(x.clone() if torch.is_tensor(x) else x for x in render(g)) for g in groups]
    torch.cuda.synchronize()
    got = [render(groups[i % 3]) for i in range(9)]
    for i, (img, radii, nr) in enumerate(got):
        w = want[i % 3]
        assert torch.equal(img, w[0]) and torch.equal(radii, w[1]) and nr == w[2], i


def test_pipe_is_opt_in_per_call_and_default_calls_do_not_touch_it(monkeypatch):
    """Without GR_RASTER_PIPELINE and without static_scene=True no internal stream is used and no input is kept alive; the
    per-call argument (GaussianRasterizer(..., static_scene=True) / rasterize_views(static_scene=True)) switches it on and
    GR_RASTER_PIPELINE=0 wins over the argument.  Every variant renders the same bits."""
    import gc, weakref
    from diff_gaussian_rasterization import GaussianRasterizer
    from gaussreg_amd import rasterizer
    monkeypatch.delenv("GR_RASTER_PIPELINE", raising=False)
    rasterizer.reset_frame_pipe()
    t, rast = _setup(seed=21)
    want = [tuple(x.clone() for x in _render(r, t)) for r in rast]
    torch.cuda.synchronize()
    scene = {k: v.clone() for k, v in t.items()}
    ref = weakref.ref(scene["means3D"])
    a = _render(rast[0], scene)[0]
    del scene
    gc.collect()
    assert ref() is None                      # default call: nothing kept
    assert torch.equal(a, want[0][0])
    piped = [GaussianRasterizer(r.raster_settings, static_scene=True) for r in rast]
    got = [_render(piped[i % 3], t) for i in range(9)]
    for i, (img, radii) in enumerate(got):
        assert torch.equal(img, want[i % 3][0]) and torch.equal(radii, want[i % 3][1]), i
    table = getattr(rasterizer._pipes, "table", None)
    assert table and any(p.keep is not None for p in table.values())   # the opt-in calls did go through the pipe
    rasterizer.reset_frame_pipe()
    monkeypatch.setenv("GR_RASTER_PIPELINE", "0")
    b = _render(piped[1], t)[0]
    assert torch.equal(b, want[1][0])
    assert all(p.keep is None for p in rasterizer._pipes.table.values())  # "0" wins: the pipe stayed untouched
