"""Batch coarse registration of scene pairs on one GPU (BASELINE.json configs[4], the per-rank body).

What the reference does per pair (experiments/geotransformer.gaussian_splatting.indoor/test.py:146-212 -> model.py:69-222):
    FPS to 30 000 points (test.py:46) -> collate / 5-level pyramid (utils/data.py:139-189) -> KPConvFPN + GeometricTransformer
    (learned; stock PyTorch in the reference) -> SuperPointMatching -> patch gather -> einsum scores -> log-Sinkhorn ->
    LocalGlobalRegistration -> RANSAC with scale (model.py:209-220) -> 4x4 transform + RRE / RTE (test.py:193-198).

Here every operator of this repo on that path runs at the real shapes, chained (each stage consumes the previous stage's
output).  The learned backbone / transformer weights are not available offline, so the two feature tensors they would
produce (`feats_c`, 256-d per superpoint; `feats_f`, per fine point) are replaced by SYNTHETIC position descriptors:
random Fourier features of the point's coordinates in the reference frame (the source cloud is mapped through the pair's
known ground-truth transform first).  That keeps the matching stack meaningful -- correspondences are real, the estimated
transform can be scored against the ground truth -- without pretending to have run the network.

`PairRegistrar(features="model")` runs the network instead (gaussreg_amd.model.GeoTransformer, 28 M parameters, seeded random
weights -- timing does not need a checkpoint; the estimates are then meaningless and are not scored): KPConvFPN over all
clouds of the batch in ONE pass (GroupNorm statistics per pair, kpconv_blocks.norm_segments), GeometricTransformer per pair,
the fine features of the backbone in the patch scores.

Pairs are independent units: `register_pairs` is what one rank runs on its block of the pair list (gaussreg_amd/sharding.py).
The clouds of a batch are stacked pair by pair: [ref_1, src_1, ref_2, src_2, ...] (cloud 2 b = ref of pair b, 2 b + 1 = src).
"""
import math
import time

import torch

from .data import precompute_data_stack_mode
from .matching import LocalGlobalRegistration, SuperPointMatching
from .ops import point_to_node_partition, point_to_node_partition_batch
from .registration import (farthest_point_sampling, registration_with_ransac_batch,
                           registration_with_ransac_from_correspondences)
from .sinkhorn import LearnableLogOptimalTransport

# experiments/geotransformer.gaussian_splatting.indoor/config.py:78-125
NUM_STAGES = 5
INIT_VOXEL = 0.025
INIT_RADIUS = 0.0625
NEIGHBOR_LIMITS = [89, 30, 43, 49, 49]  # demo.py:136
NUM_CORRESPONDENCES = 256               # coarse_matching.num_correspondences
POINT_LIMIT = 128                       # model.num_points_in_patch
NUM_SINKHORN_ITERATIONS = 100
RESULT_LEN = 16 + 4                     # flattened 4x4 + (RRE deg, RTE m, #correspondences, inlier ratio)
PATCH_CHUNK = 4096                      # patches per descriptor / Sinkhorn pass (4096 x 128 x 256 floats = 0.5 GB per side)


def synthetic_room_pair(seed, n_per_cloud, device):
    """One scene pair in the style of SURVEY.md App. D, generated on the device: both clouds sample the faces + interior
    of the same 4 x 3 x 2.5 m room (face noise sigma 1 cm, independent samples); the source cloud is then moved by a
    random rigid transform.  Returns (ref (n,3), src (n,3), T_gt (4,4) mapping src -> ref), float32."""
    g = torch.Generator(device=device).manual_seed(1_000_003 * int(seed) + 17)
    ext = torch.tensor([4.0, 3.0, 2.5], device=device)

    def cloud():
        nf = n_per_cloud // 8
        parts = []
        for axis in range(3):
            for side in (0.0, 1.0):
                p = torch.rand((nf, 3), generator=g, device=device) * ext
                p[:, axis] = side * ext[axis] + 0.01 * torch.randn((nf,), generator=g, device=device)
                parts.append(p)
        parts.append(torch.rand((n_per_cloud - 6 * nf, 3), generator=g, device=device) * ext)
        p = torch.cat(parts, 0)
        return p - ext / 2

    ref, src_in_ref = cloud(), cloud()
    ang = (torch.rand(3, generator=g, device=device) - 0.5) * torch.tensor([2 * math.pi, 0.6, 0.6], device=device)
    cz, sz, cy, sy, cx, sx = (torch.cos(ang[0]), torch.sin(ang[0]), torch.cos(ang[1]), torch.sin(ang[1]),
                              torch.cos(ang[2]), torch.sin(ang[2]))
    one, zero = torch.ones((), device=device), torch.zeros((), device=device)
    Rz = torch.stack([torch.stack([cz, -sz, zero]), torch.stack([sz, cz, zero]), torch.stack([zero, zero, one])])
    Ry = torch.stack([torch.stack([cy, zero, sy]), torch.stack([zero, one, zero]), torch.stack([-sy, zero, cy])])
    Rx = torch.stack([torch.stack([one, zero, zero]), torch.stack([zero, cx, -sx]), torch.stack([zero, sx, cx])])
    R = Rz @ Ry @ Rx                                            # src -> ref rotation
    t = (torch.rand(3, generator=g, device=device) - 0.5) * 2.0
    src = (src_in_ref - t) @ R                                  # x_ref = R x_src + t  <=>  x_src = R^T (x_ref - t)
    T = torch.eye(4, device=device)
    T[:3, :3] = R
    T[:3, 3] = t
    return ref.float().contiguous(), src.float().contiguous(), T


def spatial_sort(points, lengths, voxel):
    """Rows of a stack-mode cloud list sorted by (cloud, voxel z, y, x) at edge `voxel` (stable inside a voxel): the order
    grid_subsample(order="cell") gives the coarser levels, applied to the input level.  Clouds stay contiguous."""
    dev = points.device
    n = points.shape[0]
    cloud = torch.repeat_interleave(torch.arange(lengths.numel(), device=dev), lengths.to(dev))
    lo = points.amin(0)
    q = torch.floor((points - lo) / voxel).to(torch.int64)
    dim = q.amax(0) + 1
    key = ((cloud * dim[2] + q[:, 2]) * dim[1] + q[:, 1]) * dim[0] + q[:, 0]
    perm = torch.sort(key, stable=True).indices
    return points[perm].contiguous()


class PositionDescriptor:
    """Random Fourier features of a 3-D position: cos(W x + b) * sqrt(2 / C).  `bandwidth` (metres) sets how fast the
    descriptor decorrelates with distance.  ONE HIP launch (csrc/standin.hip, harness support -- not a reference operator):
    optional per-row transform (the pair's ground-truth pose), row mask and L2 normalisation included."""

    def __init__(self, channels, bandwidth, device, seed):
        g = torch.Generator(device=device).manual_seed(seed)
        self.W = (torch.randn((3, channels), generator=g, device=device) / bandwidth).contiguous()
        self.b = (torch.rand((channels,), generator=g, device=device) * (2 * math.pi)).contiguous()
        self.scale = math.sqrt(2.0 / channels)

    def __call__(self, x, transforms=None, transform_id=None, mask=None, normalize=False):
        """x (..., 3) -> (..., C).  transforms (k, 3, 4) / transform_id (...,) int32 (-1: identity): the row's pose;
        mask (...,) bool: rows to zero."""
        from . import _lib
        L = _lib.lib()
        lead = x.shape[:-1]
        p = x.reshape(-1, 3).to(torch.float32).contiguous()
        n, C = p.shape[0], self.W.shape[1]
        out = torch.empty((n, C), dtype=torch.float32, device=p.device)
        T = None if transforms is None else transforms.reshape(-1, 12).to(torch.float32).contiguous()
        tid = None if transform_id is None else transform_id.reshape(-1).to(torch.int32).contiguous()
        m = None if mask is None else mask.reshape(-1).to(torch.bool).contiguous()
        with torch.cuda.device(p.device):
            _lib.check(L.gr_standin_descriptors(_lib.ptr(p), n, _lib.ptr(T), _lib.ptr(tid), _lib.ptr(m), _lib.ptr(self.W),
                                                _lib.ptr(self.b), C, float(self.scale), int(bool(normalize)), _lib.ptr(out),
                                                _lib.stream_ptr(p.device)))
        return out.reshape(*lead, C)


def rotation_error_deg(Ra, Rb):
    c = ((Ra.T @ Rb).diagonal().sum() - 1.0) * 0.5
    return torch.rad2deg(torch.arccos(c.clamp(-1.0, 1.0)))


class _Section:
    def __init__(self, owner, name):
        self.owner, self.name = owner, name

    def __enter__(self):
        if self.owner.profile:
            torch.cuda.synchronize(self.owner.device)
            self.t0 = time.perf_counter()

    def __exit__(self, *exc):
        if self.owner.profile:
            torch.cuda.synchronize(self.owner.device)
            d = self.owner.section_ms
            d[self.name] = d.get(self.name, 0.0) + (time.perf_counter() - self.t0) * 1e3
        return False


class PairRegistrar:
    """Holds the stateless operator modules and the descriptors; `register_pairs` runs a batch."""

    def __init__(self, device, num_samples=30000, fps_clouds_per_call=None, order=None, use_ransac=True,
                 profile=False, pair_streams=0, features="descriptor", transformer_batch=16):
        """`pair_streams` = 0 (default): the per-pair stages (point_to_node_partition, SuperPointMatching, correspondences +
        LocalGlobalRegistration, RANSAC) run for all pairs of a batch through the stack-mode entry points
        (gr_point_to_node_partition_batch, gr_superpoint_matching_batch, gr_lgr_register_seg, gr_ransac_similarity_seg): two
        host read-backs per BATCH (match counts, number of correspondences), no host threads.  The results are those of the
        per-pair path bit for bit (tests/test_gpu_batch_ops.py, tests/test_gpu_pair_pipeline.py).
        `pair_streams` >= 1 (the round-2 path, kept for comparison): after the batched stages (FPS, pyramid) every pair runs its own short chain of launch-bound
        kernels with two host read-backs (correspondence count, RANSAC result); `pair_streams` host threads, each with
        its own HIP stream, work through the pairs so that one pair's read-back waits while the others' kernels run
        (results are identical to the sequential order: nothing is shared between pairs).  1 = one after the other.
        `fps_clouds_per_call` = None: as many clouds per farthest-point-sampling call as keep every cloud's slabs in registers
        (20 480 points per workgroup, one workgroup per CU: 25 clouds of 200 k points on 256 CUs).
        `profile=True`: `section_ms` accumulates wall milliseconds per stage (a device synchronise on both sides of
        every stage, pairs one after the other: the total is slower than an unprofiled run)."""
        if features not in ("descriptor", "model"):
            raise ValueError("features must be 'descriptor' or 'model'")
        # Row order of the pyramid INSIDE the pipeline (nothing of it crosses an API boundary: a pair goes in, a transform comes
        # out).  "reference" (default) = the reference's unordered_map iteration order at every level (what geotransformer.ext
        # returns); "cell" = rows sorted by voxel key, and the sampled input points sorted the same way, so that neighbouring
        # rows are neighbours in space.  Measured with the network (round 4, 64 pairs, tools/pairs_net_time.py): 110.6 vs 110.9
        # pairs/s, backbone 6.03 vs 5.96 ms per pair -- the KPConv gathers are 5 % of the network's kernel time
        # (profiles/r04_pairs_network_kernel_stats.csv), its GEMMs and attention kernels do not care about row order: not
        # worth a different default.  Same estimates either way up to fp32 summation order
        # (tests/test_gpu_model_e2e.py::test_cell_order_network_equals_reference_order).
        if order is None:
            order = "reference"
        if order not in ("reference", "cell"):
            raise ValueError("order must be 'reference' or 'cell'")
        self.features = features
        self.net = None
        if features == "model":
            from .model import GeoTransformer, make_cfg
            torch.manual_seed(20240301)
            self.net = GeoTransformer(make_cfg()).to(device).eval()
        self.device = device
        self.profile = bool(profile)
        self.section_ms = {}
        self.pair_streams = max(0, int(pair_streams))
        self.transformer_batch = max(1, int(transformer_batch))
        self._pool = None
        self._streams = None
        self.num_samples = int(num_samples)
        self.fps_clouds_per_call = None if fps_clouds_per_call is None else int(fps_clouds_per_call)
        self.order = order
        self.use_ransac = use_ransac
        self.coarse_desc = PositionDescriptor(256, 0.35, device, 11)
        self.fine_desc = PositionDescriptor(256, 0.06, device, 12)
        self.spm = SuperPointMatching(NUM_CORRESPONDENCES, dual_normalization=True)
        self.ot = LearnableLogOptimalTransport(NUM_SINKHORN_ITERATIONS).to(device)
        # fine_matching block of config.py:116-125
        self.lgr = LocalGlobalRegistration(3, 0.1, True, 0.05, False, False, 3, None, 5)

    def _sec(self, name):
        return _Section(self, name)

    def close(self):
        """Join the worker threads and drop their streams (so nothing of this object is left for interpreter shutdown)."""
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None
        if self._streams is not None:
            for st in self._streams:
                st.synchronize()
            self._streams = None

    @torch.no_grad()
    def register_pairs(self, pairs):
        """pairs: list of (ref (n,3), src (m,3), T_gt (4,4) or None) device tensors.
        Returns (len(pairs), RESULT_LEN) float32: [T.flatten(), RRE deg, RTE m, #correspondences, inlier ratio]."""
        return self._register_sampled(pairs, self._sample(pairs, self.fps_clouds_per_call))

    @torch.no_grad()
    def register_many(self, pairs, batch=64):
        """register_pairs for a whole rank's list: the farthest point sampling runs over ALL clouds first, in calls of
        `fps_clouds_per_call` clouds (a call's cost per cloud is lowest when its co-operating workgroups fill the device: 25
        clouds x 10 workgroups; a block of 64 pairs alone would make six calls of 21 - 22), then the pairs go through the other
        stages in blocks of `batch`.  Same rows as register_pairs block by block, except that RANSAC's seed is the pair's
        position in its block either way."""
        if len(pairs) == 0:
            return torch.zeros((0, RESULT_LEN), dtype=torch.float32, device=self.device)
        sampled = self._sample(pairs, self.fps_clouds_per_call, equal_calls=False)
        rows = [self._register_sampled(pairs[i:i + batch], sampled[2 * i:2 * (i + batch)]) for i in range(0, len(pairs), batch)]
        return torch.cat(rows, 0)

    def _sample(self, pairs, clouds_per_call, equal_calls=True):
        """FPS, several clouds per call; stack order [ref_1, src_1, ref_2, src_2, ...]: a pair's two clouds are adjacent at
        every pyramid level (data.py:151-155 stacks [ref, src] of ONE pair the same way)."""
        B = len(pairs)
        clouds = [c for p in pairs for c in (p[0], p[1])]
        sampled = []
        if clouds_per_call is None:
            # gr_fps gives a cloud n_cu // clouds workgroups and a workgroup keeps at most 20 x 1024 points in registers
            n_cu = torch.cuda.get_device_properties(self.device).multi_processor_count
            need = max(1, -(-max(c.shape[0] for c in clouds) // 20480))
            clouds_per_call = max(1, n_cu // need)
        with self._sec("fps"):
            # at most clouds_per_call clouds per launch (the co-operating workgroups of a call share the 256 CUs; 25
            # clouds of 200 k points still fit their slabs in registers), in calls of equal size
            n_calls = -(-2 * B // clouds_per_call)
            if equal_calls:
                bounds = [round(i * 2 * B / n_calls) for i in range(n_calls + 1)]
            else:  # full calls, the remainder last
                bounds = [min(i * clouds_per_call, 2 * B) for i in range(n_calls + 1)]
            for lo_, hi_ in zip(bounds[:-1], bounds[1:]):
                chunk = clouds[lo_:hi_]
                lens = [c.shape[0] for c in chunk]
                ks = [min(self.num_samples, n) for n in lens]
                if all(k == n for k, n in zip(ks, lens)):
                    sampled += chunk
                    continue
                sampled += farthest_point_sampling(torch.cat(chunk, 0), lens, ks, gather=True)
        return sampled

    def _register_sampled(self, pairs, sampled):
        B = len(pairs)
        dev = self.device
        # ---- the 5-level pyramid for the whole batch in one stack-mode pass (4 grid_subsample + 13 radius_search)
        with self._sec("pyramid"):
            points = torch.cat(sampled, 0).contiguous()
            lengths = torch.tensor([c.shape[0] for c in sampled], dtype=torch.int64)
            if self.order == "cell":
                points = spatial_sort(points, lengths, 2 * INIT_VOXEL)
            # (the index tensors of the searches are consumed by the network only: with stand-in descriptors nobody needs the
            # level-0 rows -- limit 89, ~23 columns found, searched at 32 -- as a dense copy)
            pyr = precompute_data_stack_mode(points, lengths, NUM_STAGES, INIT_VOXEL, INIT_RADIUS, NEIGHBOR_LIMITS,
                                             order=self.order, contiguous_neighbors=self.features == "model")
            len_c, len_f = pyr["lengths"][-1].tolist(), pyr["lengths"][1].tolist()
        off_c = [0]
        off_f = [0]
        for a, b in zip(len_c, len_f):          # cloud 2 b = ref of pair b, cloud 2 b + 1 = its src
            off_c.append(off_c[-1] + a)
            off_f.append(off_f[-1] + b)
        pts_c, pts_f = pyr["points"][-1], pyr["points"][1]
        out = torch.zeros((B, RESULT_LEN), dtype=torch.float32, device=dev)
        pad = torch.zeros((1, 3), device=dev)                            # model.py:171-172
        gt_mask = torch.tensor([p[2] is not None for p in pairs], device=dev)
        eye = torch.eye(4, device=dev)
        T_all = torch.stack([p[2] if p[2] is not None else eye for p in pairs])             # (B, 4, 4); no GT: identity
        feats_f = None
        if self.features == "model":
            with self._sec("backbone"):
                # KPConvFPN over all 2 B clouds at once; every GroupNorm normalises pair by pair
                from .kpconv_blocks import norm_segments, segment_table
                table = segment_table([pyr["lengths"][lv] for lv in range(NUM_STAGES)], dev)
                feats_in = torch.ones((points.shape[0], 4), device=dev)   # demo.py:109-118: [1, r, g, b]-style 4-d input
                dd = dict(pyr)
                with norm_segments(table):
                    fl = self.net.backbone(feats_in, dd)
                feats_c, feats_f = fl[-1], fl[0]                           # (sum Nc, 2048-d in), (sum Nf, 256)
        else:
            with self._sec("standin_descriptors"):
                # stand-in for the learned features (see module docstring): descriptors in the reference frame, for all the
                # superpoints of the batch at once (the per-pair stage is bound by host-side launch overhead)
                n_c = torch.tensor(len_c, device=dev)
                cloud = torch.repeat_interleave(torch.arange(2 * B, device=dev, dtype=torch.int32), n_c)   # superpoint -> cloud
                tid = torch.where(cloud % 2 == 1, cloud // 2, torch.full_like(cloud, -1))    # src clouds move into the ref frame
                feats_c = self.coarse_desc(pts_c, T_all[:, :3, :], tid, normalize=True)
        ctx = (pairs, B, pts_c, pts_f, off_c, off_f, out, feats_c)
        batched = self.pair_streams == 0
        if batched:
            # ---- stage 1, all pairs at once through the stack-mode entry points: point_to_node_partition of the 2 B clouds,
            #      (GeometricTransformer per pair), SuperPointMatching of the B pairs -- ONE host read-back (the match counts)
            with self._sec("point_to_node"):                            # model.py:99-104
                _, node_masks, knn_idx, knn_masks = point_to_node_partition_batch(pts_f, len_f, pts_c, len_c, POINT_LIMIT)
            if self.net is not None:
                with self._sec("transformer"):                          # model.py:134-148
                    # `transformer_batch` pairs at a time as one padded batch (one pair alone is ~150 launch-bound kernels
                    # on 2 x ~770 superpoints: host-bound): the embedding and the self-attention run per cloud at its true
                    # size, the projections / cross-attention / feed-forward layers over the padded stack with key masks
                    from torch.nn.utils.rnn import pad_sequence
                    parts = []
                    for a in range(0, B, self.transformer_batch):
                        bs = list(range(a, min(B, a + self.transformer_batch)))
                        rl = [off_c[2 * b + 1] - off_c[2 * b] for b in bs]
                        sl = [off_c[2 * b + 2] - off_c[2 * b + 1] for b in bs]
                        rp = pad_sequence([pts_c[off_c[2 * b]:off_c[2 * b + 1]] for b in bs], batch_first=True)
                        sp = pad_sequence([pts_c[off_c[2 * b + 1]:off_c[2 * b + 2]] for b in bs], batch_first=True)
                        rfe = pad_sequence([feats_c[off_c[2 * b]:off_c[2 * b + 1]] for b in bs], batch_first=True)
                        sfe = pad_sequence([feats_c[off_c[2 * b + 1]:off_c[2 * b + 2]] for b in bs], batch_first=True)
                        rmask = torch.arange(rp.shape[1], device=dev)[None, :] >= torch.tensor(rl, device=dev)[:, None]
                        smask = torch.arange(sp.shape[1], device=dev)[None, :] >= torch.tensor(sl, device=dev)[:, None]
                        rf, sf = self.net.transformer(rp, sp, rfe, sfe, rmask, smask, ref_lengths=rl, src_lengths=sl)
                        for i in range(len(bs)):
                            parts += [rf[i, :rl[i]], sf[i, :sl[i]]]
                    feats_cn = torch.nn.functional.normalize(torch.cat(parts, 0), p=2, dim=1)
                    del parts
            else:
                feats_cn = feats_c
            with self._sec("superpoint_matching"):                      # model.py:152-159
                ri, si, nsc, K = self.spm.forward_batch(feats_cn, len_c, node_masks)
            with self._sec("patch_features"):                           # model.py:162-170
                Kt = torch.tensor(K, device=dev)
                sel = torch.arange(ri.shape[1], device=dev)[None, :] < Kt[:, None]        # valid matches, pair-major order
                g_ref = (ri + torch.tensor(off_c[0:2 * B:2], device=dev)[:, None])[sel]   # stacked superpoint indices
                g_src = (si + torch.tensor(off_c[1:2 * B:2], device=dev)[:, None])[sel]
                st1_cat = (knn_idx[g_ref], knn_masks[g_ref], knn_idx[g_src], knn_masks[g_src])
                node_scores_flat = nsc[sel]
        else:
            # ---- stage 1, per pair on `pair_streams` host threads: point_to_node_partition x 2, (GeometricTransformer),
            #      SuperPointMatching, the patches of the matched superpoints
            st1 = [None] * B
            self._fan_out(lambda b: st1.__setitem__(b, self._match_superpoints(b, ctx)), B)
            K = [t[0].shape[0] for t in st1]
            st1_cat = (torch.cat([t[0] for t in st1]), torch.cat([t[1] for t in st1]),
                       torch.cat([t[2] for t in st1]), torch.cat([t[3] for t in st1]))
        # ---- stage 2, all pairs at once: patch coordinates -> descriptors -> scores -> log-Sinkhorn.  (One pair at a time
        #      these are 256-workgroup launches behind ~35 host calls; over the batch they fill the machine.)
        k_off = [0]
        for k in K:
            k_off.append(k_off[-1] + k)
        with self._sec("patch_features"):
            n_f = pts_f.shape[0]
            pts_pad = torch.cat([pts_f, pad], 0)                                           # model.py:171-172: pad row = index N
            Kt = torch.tensor(K, device=dev)
            pid = torch.repeat_interleave(torch.arange(B, device=dev), Kt)                 # patch -> pair
            pid32 = pid.to(torch.int32)
            n_ref = torch.tensor([off_f[2 * b + 1] - off_f[2 * b] for b in range(B)], device=dev)[pid][:, None]
            n_src = torch.tensor([off_f[2 * b + 2] - off_f[2 * b + 1] for b in range(B)], device=dev)[pid][:, None]
            o_ref = torch.tensor([off_f[2 * b] for b in range(B)], device=dev)[pid][:, None]
            o_src = torch.tensor([off_f[2 * b + 1] for b in range(B)], device=dev)[pid][:, None]
            rk, rkm, sk, skm = st1_cat
            rki = torch.where(rk == n_ref, n_f, rk + o_ref)                                # (sum K, 128): local -> stacked index
            ski = torch.where(sk == n_src, n_f, sk + o_src)
            rkp, skp = pts_pad[rki], pts_pad[ski]
            if feats_f is not None:
                feats_pad = torch.cat([feats_f, torch.zeros_like(feats_f[:1])], 0)         # model.py:181-184
        matching = torch.empty((k_off[-1], POINT_LIMIT, POINT_LIMIT), dtype=torch.float32, device=dev)
        for a in range(0, k_off[-1], PATCH_CHUNK):
            e = min(k_off[-1], a + PATCH_CHUNK)
            if feats_f is not None:                                                        # model.py:186-190
                with self._sec("patch_scores"):
                    rkf, skf = feats_pad[rki[a:e]], feats_pad[ski[a:e]]
                    scores = torch.einsum('bnd,bmd->bnm', rkf, skf) / (rkf.shape[-1] ** 0.5)
            else:
                # synthetic descriptors: position features in the reference frame (one HIP launch per side), a x16 temperature
                # instead of the network's 1 / sqrt(C) (descriptor stand-in only -- see the module docstring)
                with self._sec("standin_descriptors"):
                    rkf = self.fine_desc(rkp[a:e], mask=rkm[a:e])
                    skf = self.fine_desc(skp[a:e], T_all[:, :3, :], pid32[a:e, None].expand(-1, POINT_LIMIT), mask=skm[a:e])
                with self._sec("patch_scores"):                                            # the contraction of model.py:186-190
                    scores = torch.bmm(rkf, skf.transpose(1, 2)) * (rkf.shape[-1] ** 0.5)
            del rkf, skf
            with self._sec("sinkhorn"):
                self.ot(scores, rkm[a:e], skm[a:e], drop_dustbin=True, out=matching[a:e])   # model.py:191-198 (dustbins dropped)
                del scores
        if batched:
            # ---- stage 3, all pairs at once: correspondences + LocalGlobalRegistration (one read-back: the number of
            #      correspondences, which sizes the outputs), RANSAC with scale, the result rows -- model.py:200-220
            with self._sec("local_global_registration"):
                rc, sc, cs, T_lgr, rows = self.lgr.forward_batch(rkp, skp, rkm, skm, matching, node_scores_flat, K)
            with self._sec("ransac"):
                T_est = T_lgr
                if self.use_ransac:                                      # ransac_n = 5 (the estimate the reference keeps)
                    T_est = registration_with_ransac_batch(sc, rc, rows, T_lgr, 0.05, 5, 10000, seed=0)
            with self._sec("metrics"):
                n_corr = (rows[1:] - rows[:-1]).to(torch.float32)
                out[:, :16] = T_est.reshape(B, 16)
                out[:, 18] = n_corr
                if rc.shape[0] > 0:
                    # ground-truth inliers per pair: row -> pair, one 12-float gather, a running sum read at the pair
                    # boundaries (0/1 sums stay exact in fp32; index_add_ on a million rows spent 13 ms in atomics)
                    rid = torch.bucketize(torch.arange(rc.shape[0], device=dev, dtype=torch.int32), rows[1:], right=True)
                    Tr = T_all[:, :3, :].reshape(B, 12)[rid.to(torch.int64)]
                    dx = Tr[:, 0] * sc[:, 0] + Tr[:, 1] * sc[:, 1] + Tr[:, 2] * sc[:, 2] + Tr[:, 3] - rc[:, 0]
                    dy = Tr[:, 4] * sc[:, 0] + Tr[:, 5] * sc[:, 1] + Tr[:, 6] * sc[:, 2] + Tr[:, 7] - rc[:, 1]
                    dz = Tr[:, 8] * sc[:, 0] + Tr[:, 9] * sc[:, 1] + Tr[:, 10] * sc[:, 2] + Tr[:, 11] - rc[:, 2]
                    hit = (torch.sqrt(dx * dx + dy * dy + dz * dz) < 0.1).to(torch.float32)
                    run = torch.cat([torch.zeros(1, device=dev), torch.cumsum(hit, 0)])
                    hits = run[rows[1:].to(torch.int64)] - run[rows[:-1].to(torch.int64)]
                    out[:, 19] = torch.where(gt_mask & (n_corr > 0), hits / n_corr.clamp_min(1.0), out[:, 19])
        else:
            # ---- stage 3, per pair: LocalGlobalRegistration, RANSAC, the result row
            st3 = (pairs, out, rkp, skp, rkm, skm, matching, k_off, [t[4] for t in st1])
            self._fan_out(lambda b: self._estimate(b, st3), B)
        if bool(gt_mask.any()):
            with self._sec("metrics"):
                # RRE / RTE of all pairs at once (similarity estimate: the scale is stripped before the angle)
                T_est = out[:, :16].reshape(B, 4, 4)
                R_est = T_est[:, :3, :3]
                sc_ = torch.linalg.det(R_est).abs().clamp_min(1e-12) ** (1.0 / 3.0)
                cosv = (torch.einsum('bij,bij->b', R_est / sc_[:, None, None], T_all[:, :3, :3]) - 1.0) * 0.5
                out[:, 16] = torch.where(gt_mask, torch.rad2deg(torch.arccos(cosv.clamp(-1.0, 1.0))), out[:, 16])
                out[:, 17] = torch.where(gt_mask, torch.linalg.norm(T_est[:, :3, 3] - T_all[:, :3, 3], dim=1), out[:, 17])
        return out

    def _fan_out(self, fn, B):
        """fn(b) for b in range(B): on `pair_streams` host threads, each with its own stream (thread k takes pairs k, k + S,
        ...), or one after the other on the caller's stream."""
        dev = self.device
        S = 1 if self.profile else max(1, min(self.pair_streams, B))
        if S <= 1:
            for b in range(B):
                fn(b)
            return
        if self._pool is None:
            import concurrent.futures
            self._pool = concurrent.futures.ThreadPoolExecutor(max_workers=self.pair_streams)
            self._streams = [torch.cuda.Stream(device=dev) for _ in range(self.pair_streams)]
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(dev))

        def work(k):
            torch.cuda.set_device(dev)
            st = self._streams[k]
            st.wait_event(ready)                       # everything so far was issued on the caller's stream
            with torch.cuda.stream(st), torch.no_grad():   # grad mode is per thread
                for b in range(k, B, S):
                    fn(b)
            done = torch.cuda.Event()
            done.record(st)
            return done

        for ev in [f.result() for f in [self._pool.submit(work, k) for k in range(S)]]:
            torch.cuda.current_stream(dev).wait_event(ev)

    def _match_superpoints(self, b, ctx):
        """model.py:99-104, 152-159, 162-170 for pair b, on the current stream -> (ref patch indices (K, 128) local to the
        pair's fine cloud, their masks, the same for src, node correspondence scores (K,))."""
        pairs, B, pts_c, pts_f, off_c, off_f, out, feats_c = ctx
        r0, r1, s1 = off_c[2 * b], off_c[2 * b + 1], off_c[2 * b + 2]
        ref_c, src_c = pts_c[r0:r1], pts_c[r1:s1]
        ref_f = pts_f[off_f[2 * b]:off_f[2 * b + 1]]
        src_f = pts_f[off_f[2 * b + 1]:off_f[2 * b + 2]]
        with self._sec("point_to_node"):                            # model.py:99-104
            _, ref_node_masks, ref_knn_idx, ref_knn_masks = point_to_node_partition(ref_f, ref_c, POINT_LIMIT)
            _, src_node_masks, src_knn_idx, src_knn_masks = point_to_node_partition(src_f, src_c, POINT_LIMIT)
        if self.net is not None:
            with self._sec("transformer"):                          # model.py:134-148
                rf, sf = self.net.transformer(ref_c.unsqueeze(0), src_c.unsqueeze(0), feats_c[r0:r1].unsqueeze(0),
                                              feats_c[r1:s1].unsqueeze(0))
                ref_feats_c = torch.nn.functional.normalize(rf.squeeze(0), p=2, dim=1)
                src_feats_c = torch.nn.functional.normalize(sf.squeeze(0), p=2, dim=1)
        else:
            ref_feats_c, src_feats_c = feats_c[r0:r1], feats_c[r1:s1]
        with self._sec("superpoint_matching"):                      # model.py:152-159
            ref_ci, src_ci, node_scores = self.spm(ref_feats_c, src_feats_c, ref_node_masks, src_node_masks)
        with self._sec("patch_features"):                           # model.py:162-170
            return ref_knn_idx[ref_ci], ref_knn_masks[ref_ci], src_knn_idx[src_ci], src_knn_masks[src_ci], node_scores

    def _estimate(self, b, st3):
        """model.py:200-220 for pair b, on the current stream: correspondences + LGR, RANSAC with scale, the result row."""
        pairs, out, rkp, skp, rkm, skm, matching, k_off, node_scores = st3
        a, e = k_off[b], k_off[b + 1]
        T_gt = pairs[b][2]
        with self._sec("local_global_registration"):
            rc, sc, cs, T = self.lgr(rkp[a:e], skp[a:e], rkm[a:e], skm[a:e], matching[a:e], node_scores[b])
            n_corr = rc.shape[0]
        with self._sec("ransac"):
            if self.use_ransac and n_corr >= 5:              # model.py:209-220: ransac_n = 5 (the estimate the reference keeps)
                T = registration_with_ransac_from_correspondences(sc, rc, None, 0.05, 5, 10000, seed=b)
        with self._sec("metrics"):
            out[b, :16] = T.reshape(-1)
            out[b, 18] = float(n_corr)
            if T_gt is not None and n_corr > 0:
                out[b, 19] = (torch.linalg.norm(sc @ T_gt[:3, :3].T + T_gt[:3, 3] - rc, dim=1) < 0.1).float().mean()
