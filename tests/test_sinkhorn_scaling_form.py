"""The scaling form of the Sinkhorn iteration that csrc/sinkhorn.hip runs (K = exp(S - rowmax) fixed, one exponential per row /
column and half-iteration), restated in float64 NumPy, against the log-domain iteration of the reference
(geotransformer/modules/sinkhorn/learnable_sinkhorn.py:13-66) in float64: the same fixed-point map, so the same numbers.
Masked rows / columns (-1e12 stand-ins) are excluded from the comparison as in the GPU tests."""
import numpy as np


def _padded(scores, rm, cm, alpha, inf):
    B, M, N = scores.shape
    P = np.full((B, M + 1, N + 1), alpha, np.float64)
    P[:, :M, :N] = scores
    prm = np.zeros((B, M + 1), bool)
    prm[:, :M] = ~rm
    pcm = np.zeros((B, N + 1), bool)
    pcm[:, :N] = ~cm
    P[prm[:, :, None] | pcm[:, None, :]] = -inf
    nvr, nvc = rm.sum(1).astype(np.float64), cm.sum(1).astype(np.float64)
    norm = -np.log(nvr + nvc)
    log_mu = np.empty((B, M + 1))
    log_mu[:, :M] = norm[:, None]
    log_mu[:, M] = np.log(nvc) + norm
    log_nu = np.empty((B, N + 1))
    log_nu[:, :N] = norm[:, None]
    log_nu[:, N] = np.log(nvr) + norm
    return P, prm, pcm, log_mu, log_nu, norm


def _lse(x, axis):
    m = x.max(axis=axis, keepdims=True)
    return (m + np.log(np.exp(x - m).sum(axis=axis, keepdims=True))).squeeze(axis)


def log_domain(scores, rm, cm, alpha=1.0, iters=100, inf=1e12):
    P, prm, pcm, log_mu, log_nu, norm = _padded(scores, rm, cm, alpha, inf)
    log_mu = np.where(prm, -inf, log_mu)
    log_nu = np.where(pcm, -inf, log_nu)
    u, v = np.zeros_like(log_mu), np.zeros_like(log_nu)
    for _ in range(iters):
        u = log_mu - _lse(P + v[:, None, :], 2)
        v = log_nu - _lse(P + u[:, :, None], 1)
    return P + u[:, :, None] + v[:, None, :] - norm[:, None, None], prm, pcm


def scaling_form(scores, rm, cm, alpha=1.0, iters=100, inf=1e12):
    """What the kernel does: K and the row maxima once; u, v only through exp(v), exp(u + rmax - c)."""
    P, prm, pcm, log_mu, log_nu, norm = _padded(scores, rm, cm, alpha, inf)
    live = ~(prm[:, :, None] | pcm[:, None, :])
    rmax = np.where(live, P, -np.inf).max(2)
    rmax = np.where(prm, 0.0, rmax)
    K = np.where(live, np.exp(P - rmax[:, :, None]), 0.0)
    u, v = np.zeros_like(log_mu), np.zeros_like(log_nu)
    E = np.where(pcm, 0.0, 1.0)                       # exp(v), v = 0
    cw = norm[:, None]
    for _ in range(iters):
        rs = (K * E[:, None, :]).sum(2)
        w = log_mu - np.log(np.where(prm, 1.0, rs))   # u + rmax
        u = np.where(prm, 0.0, w - rmax)
        F = np.where(prm, 0.0, np.exp(w - cw))
        cs = (K * F[:, :, None]).sum(1)
        v = np.where(pcm, 0.0, log_nu - (cw + np.log(np.where(pcm, 1.0, cs))))
        E = np.where(pcm, 0.0, np.exp(v))
    return P + u[:, :, None] + v[:, None, :] - norm[:, None, None]


def test_scaling_form_is_the_log_domain_iteration():
    rng = np.random.default_rng(4)
    for sigma in (1.5, 8.0):
        B, K = 6, 40
        s = rng.normal(size=(B, K, K)) * sigma
        rm, cm = rng.random((B, K)) > 0.3, rng.random((B, K)) > 0.3
        rm[0] = True
        cm[0] = True
        want, prm, pcm = log_domain(s, rm, cm, alpha=0.37)
        got = scaling_form(s, rm, cm, alpha=0.37)
        live = ~(prm[:, :, None] | pcm[:, None, :])
        assert np.abs(got - want)[live].max() <= 1e-9 * np.abs(want[live]).max()
        assert (got[~live] < -1e11).all()
