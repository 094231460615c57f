"""CPU model of the round structure of csrc/fps.hip (NOT a timing model): how many exchange rounds does a 200 k -> 30 k
farthest point sampling need for a given candidate hierarchy (MW per wave, M per workgroup, MG in the global chain) and
acceptance rule?  Used to pick those constants before writing the kernel; results are quoted in DESIGN.md.

    python tools/fps_round_model.py --mg 32 --rule prefix          # the round-2 kernel
    python tools/fps_round_model.py --mg 64 --rule greedy

Rules (both give exactly the sequential FPS order; the script asserts it against a plain arg-max loop on a prefix):
  prefix   candidate i is accepted while every earlier candidate was, key_i > B and no earlier candidate lies within
           sqrt(d_i) of it
  greedy   exact FPS restricted to the candidate set: take the candidate with the largest CURRENT key while it beats B,
           lower the other candidates' distances by the accepted one
B = the largest key that is not among the candidates (runner-ups of threads, overflow of waves / workgroups).
"""
import argparse
import time

import numpy as np

T, WAVE = 1024, 64


def room(n, seed):
    r = np.random.default_rng(seed)
    ext = np.array([4.0, 3.0, 2.5])
    nf = n // 8
    parts = []
    for axis in range(3):
        for side in (0.0, 1.0):
            p = r.random((nf, 3)) * ext
            p[:, axis] = side * ext[axis] + 0.01 * r.standard_normal(nf)
            parts.append(p)
    parts.append(r.random((n - 6 * nf, 3)) * ext)
    return (np.concatenate(parts) - ext / 2).astype(np.float32)


def morton_order(p):
    mn = p.min(0)
    ext = (p.max(0) - mn).max()
    c = np.clip((p - mn) * (1023.0 / ext), 0, 1023).astype(np.uint64)

    def spread(v):
        v = (v | (v << np.uint64(16))) & np.uint64(0x030000ff)
        v = (v | (v << np.uint64(8))) & np.uint64(0x0300f00f)
        v = (v | (v << np.uint64(4))) & np.uint64(0x030c30c3)
        v = (v | (v << np.uint64(2))) & np.uint64(0x09249249)
        return v
    code = spread(c[:, 0]) | (spread(c[:, 1]) << np.uint64(1)) | (spread(c[:, 2]) << np.uint64(2))
    return np.argsort(code, kind="stable")


def hilbert_order(p, bits=10):
    """Position along a 3-D Hilbert curve (Skilling's transpose algorithm) of the 1024^3 cell a point lies in."""
    mn = p.min(0)
    ext = (p.max(0) - mn).max()
    X = np.clip((p - mn) * (((1 << bits) - 1) / ext), 0, (1 << bits) - 1).astype(np.int64)
    M = 1 << (bits - 1)
    Q = M
    while Q > 1:
        P = Q - 1
        for i in range(3):
            m = (X[:, i] & Q) != 0
            X[m, 0] ^= P
            t = (X[~m, 0] ^ X[~m, i]) & P
            X[~m, 0] ^= t
            X[~m, i] ^= t
        Q >>= 1
    for i in range(1, 3):
        X[:, i] ^= X[:, i - 1]
    t = np.zeros(len(X), np.int64)
    Q = M
    while Q > 1:
        m = (X[:, 2] & Q) != 0
        t[m] ^= Q - 1
        Q >>= 1
    X ^= t[:, None]
    key = np.zeros(len(X), np.int64)
    for b in range(bits - 1, -1, -1):
        for i in range(3):
            key = (key << 1) | ((X[:, i] >> b) & 1)
    return np.argsort(key, kind="stable")


def keys_of(d, idx):
    return (d.astype(np.float32).view(np.uint32).astype(np.uint64) << np.uint64(32)) | (np.uint64(0xffffffff) - idx.astype(np.uint64))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=200000)
    ap.add_argument("--k", type=int, default=30000)
    ap.add_argument("--g", type=int, default=64)
    ap.add_argument("--mw", type=int, default=4)
    ap.add_argument("--m", type=int, default=8)
    ap.add_argument("--mg", type=int, default=32)
    ap.add_argument("--rule", default="prefix", choices=["prefix", "greedy"])
    ap.add_argument("--eligible", type=int, default=0,
                    help="U > 0: no global top-MG selection; candidates = every published key above max(workgroup bounds, "
                         "the largest (U+1)-th key of a workgroup), i.e. at most U per workgroup (greedy rule)")
    ap.add_argument("--per-thread", type=int, default=1, help="how many of its points a thread offers (the next one bounds)")
    ap.add_argument("--check", type=int, default=300, help="samples compared with the plain arg-max loop")
    ap.add_argument("--order", default="morton", choices=["morton", "hilbert"], help="order of the points along the slabs")
    ap.add_argument("--fold-stats", action="store_true",
                    help="per round: how many of the accepted samples pass each wave's box test (the fold's critical path)")
    a = ap.parse_args()
    P0 = room(a.n, 0)
    order = hilbert_order(P0) if a.order == "hilbert" else morton_order(P0)
    S = P0[order]                       # Morton order; keys carry the ORIGINAL index
    n, G = a.n, a.g
    per = -(-(-(-n // G)) // T) * T
    ppt = per // T
    nw = T // WAVE
    # slot (g, wave, j, lane) -> Morton position
    pos = (np.arange(G)[:, None, None, None] * per + np.arange(nw)[None, :, None, None] * ppt * WAVE
           + np.arange(ppt)[None, None, :, None] * WAVE + np.arange(WAVE)[None, None, None, :])
    valid = pos < n
    pos = np.minimum(pos, n - 1)
    orig = order[pos].astype(np.uint64)
    d = np.full(n, np.inf, np.float32)   # by Morton position
    acc = [P0[0]]
    out = [0]
    rounds, hist, stops = 0, np.zeros(G * a.m + 1, np.int64), {"bound": 0, "hurt": 0, "all": 0, "k": 0}
    esize, involved, which = [], [], []
    t0 = time.time()
    # boxes of the waves' runs (static) for --fold-stats
    SP = np.where(valid[..., None], S[pos], np.nan)                       # (G, nw, ppt, 64, 3)
    box_lo = np.nanmin(SP, axis=(2, 3)).reshape(-1, 3)
    box_hi = np.nanmax(SP, axis=(2, 3)).reshape(-1, 3)
    fold_max, fold_sum, fold_rounds = 0, 0, 0
    npair = (ppt + 1) // 2
    SPp = np.full((G, nw, 2 * npair, WAVE, 3), np.nan)
    SPp[:, :, :ppt] = SP
    SPp = SPp.reshape(G * nw, npair, 2 * WAVE, 3)
    with np.errstate(all="ignore"):
        rbox_lo, rbox_hi = np.nanmin(SPp, axis=2), np.nanmax(SPp, axis=2)        # (waves, row pairs, 3)
    rbox_lo, rbox_hi = np.where(np.isnan(rbox_lo), np.inf, rbox_lo), np.where(np.isnan(rbox_hi), -np.inf, rbox_hi)
    row_cost = [0, 0, 0]
    while len(out) < a.k:
        A = np.asarray(acc, np.float32)
        if a.fold_stats and rounds > 0:
            wmax = np.where(valid, d[pos], -1.0).max(axis=(2, 3)).reshape(-1)    # largest running distance of every wave
            e = np.maximum(np.maximum(box_lo[:, None, :] - A[None], A[None] - box_hi[:, None, :]), 0.0)
            lb = (e * e).sum(2)                                                   # (waves, samples)
            passed = lb < wmax[:, None]
            hit = passed.sum(1)
            fold_max += int(hit.max())
            fold_sum += int(hit.sum())
            fold_rounds += 1
            # what a second test per PAIR OF ROWS (128 consecutive points, the unit of the packed fold) would leave: cost of a
            # wave = passed samples x (row tests + loop skeleton) + 8 per (sample, row pair) that still has to be folded
            e2 = np.maximum(np.maximum(rbox_lo[:, :, None, :] - A[None, None], A[None, None] - rbox_hi[:, :, None, :]), 0.0)
            rhit = ((e2 * e2).sum(3) < wmax[:, None, None]) & passed[:, None, :]       # (waves, row pairs, samples)
            cost_now = hit * 8 * rbox_lo.shape[1]
            cost_rows = hit * 30 + rhit.sum((1, 2)) * 8
            row_cost[0] += int(cost_now.max())
            row_cost[1] += int(cost_rows.max())
            row_cost[2] += int(rhit.sum())
        for s in A:
            diff = S - s
            d = np.minimum(d, (diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1]) + diff[:, 2] * diff[:, 2])
        kk = np.where(valid, keys_of(d[pos], orig), np.uint64(0))          # (G, nw, ppt, 64)
        ks = np.sort(kk, axis=2)
        pt = min(a.per_thread, ppt)
        offered = ks[:, :, ppt - pt:, :].reshape(G, nw, -1)                 # the pt best points of every thread
        second = ks[:, :, ppt - pt - 1, :] if ppt > pt else np.zeros_like(ks[:, :, 0, :])
        wb = np.sort(offered, axis=2)[:, :, ::-1]                           # (G, nw, 64 pt) descending
        wtop = wb[:, :, :a.mw]
        wbound = np.maximum(wb[:, :, a.mw], second.max(2))                  # (G, nw)
        which.append(int(np.argmax([wb[:, :, a.mw].max(), second.max()])))
        gl = np.sort(wtop.reshape(G, -1), axis=1)[:, ::-1]                  # (G, nw*mw)
        gtop = gl[:, :a.m]
        gbound = np.maximum(gl[:, a.m] if gl.shape[1] > a.m else 0, wbound.max(1))
        al = np.sort(gtop.reshape(-1))[::-1]
        if a.eligible:
            B = int(gbound.max())
            if a.eligible < a.m:
                B = max(B, int(gtop[:, a.eligible].max()))
            cand = al[al > np.uint64(B)]
            esize.append(len(cand))
        else:
            cand = al[:a.mg]
            B = max(int(al[a.mg]) if al.shape[0] > a.mg else 0, int(gbound.max()))
        ci = (np.uint64(0xffffffff) - (cand & np.uint64(0xffffffff))).astype(np.int64)
        cd = (cand >> np.uint64(32)).astype(np.uint32).view(np.float32)
        cp = P0[ci]
        room_left = a.k - len(out)
        got = []
        if a.rule == "prefix":
            why = "all"
            for i in range(len(cand)):
                if cand[i] == 0 or int(cand[i]) <= B:
                    why = "bound"
                    break
                if i:
                    diff = cp[:i] - cp[i]
                    dd = (diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1]) + diff[:, 2] * diff[:, 2]
                    if (dd < cd[i]).any():
                        why = "hurt"
                        break
                got.append(i)
                if len(got) == room_left:
                    why = "k"
                    break
            stops[why] += 1
        else:
            cur = cd.copy()
            live = cand != 0
            if len(cand) > 1:   # candidates in some conflict relation (hurt by, or hurting, another candidate)
                df = cp[:, None, :] - cp[None, :, :]
                dm = (df[..., 0] * df[..., 0] + df[..., 1] * df[..., 1]) + df[..., 2] * df[..., 2]
                h = dm < cd[:, None]
                np.fill_diagonal(h, False)
                involved.append(int((h.any(0) | h.any(1)).sum()))
            why = "all"
            while live.any():
                ck = np.where(live, keys_of(cur, ci), np.uint64(0))
                i = int(ck.argmax())
                if int(ck[i]) <= B:
                    why = "bound"
                    break
                got.append(i)
                live[i] = False
                diff = cp - cp[i]
                cur = np.minimum(cur, (diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1]) + diff[:, 2] * diff[:, 2])
                if len(got) == room_left:
                    why = "k"
                    break
            stops[why] += 1
        assert got, "a round accepts at least the arg-max"
        acc = [cp[i] for i in got]
        out.extend(int(ci[i]) for i in got)
        hist[len(got)] += 1
        rounds += 1
        if rounds % 200 == 0:
            print(f"  round {rounds}: {len(out)} samples, {len(out) / rounds:.1f} per round, {time.time() - t0:.0f} s", flush=True)
    # the plain loop on a prefix
    dd = np.full(n, np.inf, np.float32)
    cur = 0
    for s in range(1, min(a.check, a.k)):
        diff = P0 - P0[cur]
        dd = np.minimum(dd, (diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1]) + diff[:, 2] * diff[:, 2])
        cur = int(keys_of(dd, np.arange(n)).argmax())
        assert cur == out[s], (s, cur, out[s])
    print({"g": G, "ppt": ppt, "mw": a.mw, "m": a.m, "mg": a.mg, "rule": a.rule, "rounds": rounds,
           "per_round": round(a.k / rounds, 2), "stops": stops,
           "accepted_hist_deciles": [int(x) for x in np.percentile(np.repeat(np.arange(len(hist)), hist), [10, 50, 90])],
           "eligible_deciles": [int(x) for x in np.percentile(esize, [10, 50, 90, 100])] if esize else None,
           "bound_from_thread_runner_up": round(float(np.mean(which)), 3),
           "in_conflict_deciles": [int(x) for x in np.percentile(involved, [10, 50, 90, 100])] if involved else None,
           "fold": {"order": a.order, "sum_over_rounds_of_the_busiest_wave": fold_max,
                    "sum_over_rounds_of_the_mean_wave": round(fold_sum / (G * nw), 1),
                    "sample_wave_pairs": fold_sum,
                    "critical_path_cost_units": {"wave_test_only": row_cost[0], "with_row_pair_tests": row_cost[1],
                                                 "sample_rowpair_folds": row_cost[2]}} if a.fold_stats else None})


if __name__ == "__main__":
    main()
