"""Random call sequences through the rasterizer's frame pipe (gaussreg_amd/rasterizer.py _FramePipe): one- and many-camera
calls, in-place updates and replacements of the scene tensors, switches of the caller's stream, the fast exponential on and
off -- every output must equal the serial render (GR_RASTER_PIPELINE=0) of the scene AS IT WAS at that call."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _pipe_on(monkeypatch):
    # the pipe is opt-in (static_scene=True / GR_RASTER_PIPELINE=1); "0" inside a test selects the serial path
    monkeypatch.setenv("GR_RASTER_PIPELINE", "1")


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_call_sequences(seed, monkeypatch):
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from gaussreg_amd import synthetic
    from gaussreg_amd.rasterizer import rasterize_views
    rng = np.random.default_rng(100 + seed)
    P, W, H = 40000, 256, 160
    g = synthetic.gaussians_c2(P, 20 + seed, sh_degree=3)
    cams = synthetic.camera_ring(6, W, H, seed=seed)
    sets = [GaussianRasterizationSettings(H, W, c["tanfovx"], c["tanfovy"], torch.zeros(3).cuda(), 1.0,
                                          torch.from_numpy(c["viewmatrix"]).cuda(), torch.from_numpy(c["projmatrix"]).cuda(), 3,
                                          torch.from_numpy(c["campos"]).cuda(), False, False) for c in cams]
    scene = {k: torch.from_numpy(v).cuda() for k, v in g.items()}
    streams = [torch.cuda.current_stream(), torch.cuda.Stream()]
    cur = 0
    log = []  # (kind, cameras, fast, snapshot of the scene, outputs)

    def snap():
        return {k: v.clone() for k, v in scene.items()}

    for step in range(40):
        op = rng.integers(0, 10)
        with torch.cuda.stream(streams[cur]):
            if op == 0:      # in place
                scene["means3D"].add_(torch.from_numpy(rng.normal(0, 0.01, 3).astype(np.float32)).cuda())
            elif op == 1:    # replaced
                k = ("opacities", "shs", "scales")[int(rng.integers(0, 3))]
                scene[k] = scene[k] * float(rng.uniform(0.9, 1.0))
            elif op == 2:    # the caller moves to its other stream (ordered behind the work of the old one)
                nxt = 1 - cur
                streams[nxt].wait_stream(streams[cur])
                cur = nxt
                continue
            fast = bool(rng.integers(0, 4) == 0)
            kw = dict(shs=scene["shs"], scales=scene["scales"], rotations=scene["rotations"])
            if rng.integers(0, 3) == 0:
                idx = sorted(rng.choice(6, size=int(rng.integers(2, 6)), replace=False).tolist())
                out = rasterize_views([sets[i] for i in idx], scene["means3D"], scene["opacities"], fast_exp=fast, **kw)
                log.append(("many", idx, fast, snap(), (out[0], out[1])))
            else:
                i = int(rng.integers(0, 6))
                out = GaussianRasterizer(sets[i], fast_exp=fast)(scene["means3D"], None, scene["opacities"], **kw)
                log.append(("one", [i], fast, snap(), out))
    for s in streams:
        s.synchronize()
    monkeypatch.setenv("GR_RASTER_PIPELINE", "0")
    for n, (kind, idx, fast, sc, out) in enumerate(log):
        kw = dict(shs=sc["shs"], scales=sc["scales"], rotations=sc["rotations"])
        if kind == "many":
            want = rasterize_views([sets[i] for i in idx], sc["means3D"], sc["opacities"], fast_exp=fast, **kw)[:2]
        else:
            want = GaussianRasterizer(sets[idx[0]], fast_exp=fast)(sc["means3D"], None, sc["opacities"], **kw)
        assert torch.equal(out[0], want[0]) and torch.equal(out[1], want[1]), (n, kind, idx, fast)
