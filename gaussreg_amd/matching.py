"""Mirror of the reference's matching nn.Modules, bodies replaced by fused HIP kernels.

    SuperPointMatching   geotransformer/modules/geotransformer/superpoint_matching.py:7-50
    PointMatching        geotransformer/modules/geotransformer/point_matching.py:5-115
                         (compute_correspondence_matrix is also what LocalGlobalRegistration uses,
                          local_global_registration.py:49-83)

Same constructor arguments, same forward signatures and return tuples; no parameters or buffers
(no state-dict keys), like the reference.
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib


def _f32c(t, dev):
    if t.dtype != torch.float32:
        raise RuntimeError("expected a float tensor")
    return (t if t.is_cuda else t.to(dev)).contiguous()


def _boolc(t, dev):
    if t.dtype != torch.bool:
        t = t != 0
    return (t if t.is_cuda else t.to(dev)).contiguous()


class SuperPointMatching(nn.Module):
    def __init__(self, num_correspondences, dual_normalization=True):
        super().__init__()
        self.num_correspondences = num_correspondences
        self.dual_normalization = dual_normalization

    @torch.no_grad()
    def forward(self, ref_feats, src_feats, ref_masks=None, src_masks=None):
        """-> (ref_corr_indices (k,) i64, src_corr_indices (k,) i64, corr_scores (k,) f32 descending)."""
        dev = _lib.require_gpu()
        L = _lib.lib()
        out_device = ref_feats.device
        rf = _f32c(ref_feats, dev)
        dev = rf.device
        sf = _f32c(src_feats, dev)
        rm = None if ref_masks is None else _boolc(ref_masks, dev)
        sm = None if src_masks is None else _boolc(src_masks, dev)
        nr, c = rf.shape
        ns = sf.shape[0]
        k = int(self.num_correspondences)
        ri = torch.empty((k,), dtype=torch.int64, device=dev)
        si = torch.empty((k,), dtype=torch.int64, device=dev)
        sc = torch.empty((k,), dtype=torch.float32, device=dev)
        n_out = ctypes.c_int64(0)
        with torch.cuda.device(dev):
            ws = _lib.workspace(dev, L.gr_superpoint_matching_workspace_bytes(nr, ns))
            _lib.check(L.gr_superpoint_matching(_lib.ptr(rf), _lib.ptr(sf), nr, ns, c, _lib.ptr(rm), _lib.ptr(sm), k,
                                                int(bool(self.dual_normalization)), _lib.ptr(ri), _lib.ptr(si),
                                                _lib.ptr(sc), ctypes.byref(n_out), _lib.ptr(ws), ws.numel(),
                                                _lib.stream_ptr(dev)))
        n = n_out.value
        ri, si, sc = ri[:n], si[:n], sc[:n]
        if out_device.type != "cuda":
            ri, si, sc = ri.to(out_device), si.to(out_device), sc.to(out_device)
        return ri, si, sc


    @torch.no_grad()
    def forward_batch(self, feats, node_lengths, masks=None):
        """The same for a batch of scene pairs in one call (gr_superpoint_matching_batch): `feats` (sum M, C) stacks the
        L2-normalised superpoint features as [ref_0, src_0, ref_1, src_1, ...], `node_lengths` their 2 B sizes, `masks`
        (sum M,) likewise.  -> (ref_corr_indices (B, k), src_corr_indices (B, k), corr_scores (B, k), counts: list of B ints
        -- row b is valid up to counts[b]).  One host read-back for the whole batch."""
        dev = _lib.require_gpu()
        L = _lib.lib()
        f = _f32c(feats, dev)
        dev = f.device
        m = None if masks is None else _boolc(masks, dev)
        off = [0]
        for n in node_lengths:
            off.append(off[-1] + int(n))
        if len(off) % 2 != 1 or off[-1] != f.shape[0]:
            raise ValueError("node_lengths must list ref and src sizes of every pair and sum to feats.shape[0]")
        B = (len(off) - 1) // 2
        k = int(self.num_correspondences)
        ri = torch.zeros((B, k), dtype=torch.int64, device=dev)
        si = torch.zeros((B, k), dtype=torch.int64, device=dev)
        sc = torch.zeros((B, k), dtype=torch.float32, device=dev)
        h_off = _lib.host_i64(off)
        h_n = _lib.host_i64([0] * B)
        with torch.cuda.device(dev):
            ws = _lib.workspace(dev, L.gr_superpoint_matching_batch_workspace_bytes(h_off, B))
            _lib.check(L.gr_superpoint_matching_batch(_lib.ptr(f), h_off, B, f.shape[1], _lib.ptr(m), k,
                                                      int(bool(self.dual_normalization)), _lib.ptr(ri), _lib.ptr(si),
                                                      _lib.ptr(sc), h_n, _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)))
        return ri, si, sc, [int(h_n[b]) for b in range(B)]


class PointMatching(nn.Module):
    def __init__(self, k: int, mutual: bool = True, confidence_threshold: float = 0.05, use_dustbin: bool = False,
                 use_global_score: bool = False, remove_duplicate: bool = False):
        super().__init__()
        self.k = k
        self.mutual = mutual
        self.confidence_threshold = confidence_threshold
        self.use_dustbin = use_dustbin
        self.use_global_score = use_global_score
        self.remove_duplicate = remove_duplicate
        if use_dustbin:
            # the reference's dustbin branch slices corr_mat[:, -1:, -1] (point_matching.py:61-62), which
            # yields a (B, 1) tensor and cannot be combined with the (B, K, K) mask on the next line;
            # GaussReg never enables it (model.py:51-65 uses LocalGlobalRegistration with use_dustbin False)
            raise NotImplementedError("use_dustbin=True is not supported (unusable in the reference as well)")

    def _corr(self, score_mat, ref_knn_masks, src_knn_masks, want_count, scores_are_exp=False):
        dev = _lib.require_gpu()
        L = _lib.lib()
        s = _f32c(score_mat, dev)
        dev = s.device
        rm, sm = _boolc(ref_knn_masks, dev), _boolc(src_knn_masks, dev)
        B, K1, K2 = s.shape
        corr = torch.empty((B, K1, K2), dtype=torch.bool, device=dev)
        n = ctypes.c_int64(0)
        with torch.cuda.device(dev):
            ws = _lib.workspace(dev, L.gr_point_matching_workspace_bytes(B))
            fn = L.gr_corr_matrix_exp if scores_are_exp else L.gr_corr_matrix
            _lib.check(fn(_lib.ptr(s), B, K1, K2, _lib.ptr(rm), _lib.ptr(sm), int(self.k),
                          int(bool(self.mutual)), float(self.confidence_threshold), _lib.ptr(corr),
                          ctypes.byref(n) if want_count else None, _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)))
        return s, corr, n.value, ws

    @torch.no_grad()
    def compute_correspondence_matrix(self, score_mat, ref_knn_masks, src_knn_masks):
        """`score_mat` here is exp(log-scores), as in the reference call sites (point_matching.py:98,
        local_global_registration.py:211): it is thresholded as given (gr_corr_matrix_exp), no log/exp round trip."""
        s, corr, _, _ = self._corr(score_mat, ref_knn_masks, src_knn_masks, False, scores_are_exp=True)
        return corr if score_mat.is_cuda else corr.to(score_mat.device)

    @torch.no_grad()
    def forward(self, ref_knn_points, src_knn_points, ref_knn_masks, src_knn_masks, ref_knn_indices, src_knn_indices,
                score_mat, global_scores):
        out_device = score_mat.device
        s, corr, n, ws = self._corr(score_mat, ref_knn_masks, src_knn_masks, True)
        dev = s.device
        L = _lib.lib()
        B, K1, K2 = s.shape
        rp, sp = _f32c(ref_knn_points, dev), _f32c(src_knn_points, dev)
        ri = ref_knn_indices.to(dev).contiguous()
        si = src_knn_indices.to(dev).contiguous()
        gs = _f32c(global_scores, dev) if (self.use_global_score and global_scores is not None) else None
        o_rp = torch.empty((n, 3), dtype=torch.float32, device=dev)
        o_sp = torch.empty((n, 3), dtype=torch.float32, device=dev)
        o_ri = torch.empty((n,), dtype=torch.int64, device=dev)
        o_si = torch.empty((n,), dtype=torch.int64, device=dev)
        o_sc = torch.empty((n,), dtype=torch.float32, device=dev)
        if n > 0:
            with torch.cuda.device(dev):
                _lib.check(L.gr_corr_gather(_lib.ptr(s), B, K1, K2, _lib.ptr(corr), _lib.ptr(rp), _lib.ptr(sp),
                                            _lib.ptr(ri), _lib.ptr(si), _lib.ptr(gs), int(gs is not None),
                                            _lib.ptr(o_rp), _lib.ptr(o_sp), _lib.ptr(o_ri), _lib.ptr(o_si),
                                            _lib.ptr(o_sc), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)))
        outs = (o_rp, o_sp, o_ri, o_si, o_sc)
        if out_device.type != "cuda":
            outs = tuple(o.to(out_device) for o in outs)
        return outs


class LocalGlobalRegistration(nn.Module):
    """Mirror of geotransformer/modules/geotransformer/local_global_registration.py:11-235 (forward,
    inference).  Correspondence extraction = the PointMatching kernels; hypothesis generation, verification
    and refinement run in three more launches with no GPU->CPU SVD round trips (the reference does
    1 + num_refinement_steps of them, registration/procrustes.py:59)."""

    def __init__(self, k: int, acceptance_radius: float, mutual: bool = True, confidence_threshold: float = 0.05,
                 use_dustbin: bool = False, use_global_score: bool = False, correspondence_threshold: int = 3,
                 correspondence_limit=None, num_refinement_steps: int = 5):
        super().__init__()
        self.k = k
        self.acceptance_radius = acceptance_radius
        self.mutual = mutual
        self.confidence_threshold = confidence_threshold
        self.use_dustbin = use_dustbin
        self.use_global_score = use_global_score
        self.correspondence_threshold = correspondence_threshold
        self.correspondence_limit = correspondence_limit
        self.num_refinement_steps = num_refinement_steps
        if use_dustbin:
            raise NotImplementedError("use_dustbin=True is not supported (GaussReg sets it False, config.py:121)")
        self._pm = PointMatching(k, mutual, confidence_threshold, False, use_global_score)

    @torch.no_grad()
    def compute_correspondence_matrix(self, score_mat, ref_knn_masks, src_knn_masks):
        return self._pm.compute_correspondence_matrix(score_mat, ref_knn_masks, src_knn_masks)

    @torch.no_grad()
    def forward(self, ref_knn_points, src_knn_points, ref_knn_masks, src_knn_masks, score_mat, global_scores):
        """-> (ref_corr_points (C,3), src_corr_points (C,3), corr_scores (C,), estimated_transform (4,4))."""
        out_device = score_mat.device
        s, corr, n, pm_ws = self._pm._corr(score_mat, ref_knn_masks, src_knn_masks, True)
        dev = s.device
        L = _lib.lib()
        B, K1, K2 = s.shape
        rp, sp = _f32c(ref_knn_points, dev), _f32c(src_knn_points, dev)
        gs = _f32c(global_scores, dev) if (self.use_global_score and global_scores is not None) else None
        o_rp = torch.empty((n, 3), dtype=torch.float32, device=dev)
        o_sp = torch.empty((n, 3), dtype=torch.float32, device=dev)
        o_sc = torch.empty((n,), dtype=torch.float32, device=dev)
        idx_dummy = torch.zeros((B, max(K1, K2)), dtype=torch.int64, device=dev)
        o_i = torch.empty((2, max(n, 1)), dtype=torch.int64, device=dev)
        transform = torch.eye(4, dtype=torch.float32, device=dev)
        if n > 0:
            with torch.cuda.device(dev):
                st = _lib.stream_ptr(dev)
                _lib.check(L.gr_corr_gather(_lib.ptr(s), B, K1, K2, _lib.ptr(corr), _lib.ptr(rp), _lib.ptr(sp),
                                            _lib.ptr(idx_dummy), _lib.ptr(idx_dummy), _lib.ptr(gs), int(gs is not None),
                                            _lib.ptr(o_rp), _lib.ptr(o_sp), _lib.ptr(o_i[0]), _lib.ptr(o_i[1]),
                                            _lib.ptr(o_sc), _lib.ptr(pm_ws), pm_ws.numel(), st))
                ws2 = torch.empty(L.gr_lgr_workspace_bytes(B) + 256, dtype=torch.uint8, device=dev)
                v_rp, v_sp, v_sc = o_rp, o_sp, o_sc
                if self.correspondence_limit is not None and n > int(self.correspondence_limit):
                    # verification set = top-`limit` global scores (local_global_registration.py:145-148)
                    v_sc, sel = o_sc.topk(k=int(self.correspondence_limit), largest=True)
                    v_rp, v_sp, v_sc = o_rp[sel].contiguous(), o_sp[sel].contiguous(), v_sc.contiguous()
                _lib.check(L.gr_lgr_register_verify(_lib.ptr(o_rp), _lib.ptr(o_sp), _lib.ptr(o_sc), n, B, _lib.ptr(pm_ws),
                                                    _lib.ptr(v_rp), _lib.ptr(v_sp), _lib.ptr(v_sc), v_sc.shape[0],
                                                    float(self.acceptance_radius), int(self.correspondence_threshold),
                                                    int(self.num_refinement_steps), _lib.ptr(transform), _lib.ptr(ws2),
                                                    ws2.numel(), st))
        outs = (o_rp, o_sp, o_sc, transform)
        if out_device.type != "cuda":
            outs = tuple(o.to(out_device) for o in outs)
        return outs

    @torch.no_grad()
    def forward_batch(self, ref_knn_points, src_knn_points, ref_knn_masks, src_knn_masks, score_mat, global_scores,
                      patches_per_pair):
        """forward for a batch of scene pairs in one pass: the P patches of all pairs are stacked along dim 0, pair b owning
        the next patches_per_pair[b] of them.  -> (ref_corr_points (C,3), src_corr_points (C,3), corr_scores (C,),
        estimated_transforms (B,4,4), row_offsets (B+1,) int32 on the device: pair b's correspondences are rows
        [row_offsets[b], row_offsets[b+1]) -- the single-pair outputs concatenated).  ONE host read-back (C, which sizes the
        outputs) for the whole batch; correspondence_limit is not supported here (the GaussReg config leaves it None)."""
        if self.correspondence_limit is not None:
            raise NotImplementedError("forward_batch: correspondence_limit must be None")
        s, corr, n, pm_ws = self._pm._corr(score_mat, ref_knn_masks, src_knn_masks, True)
        dev = s.device
        L = _lib.lib()
        P, K1, K2 = s.shape
        poff = [0]
        for c in patches_per_pair:
            poff.append(poff[-1] + int(c))
        if poff[-1] != P:
            raise ValueError("patches_per_pair must sum to the number of patches")
        B = len(poff) - 1
        rp, sp = _f32c(ref_knn_points, dev), _f32c(src_knn_points, dev)
        gs = _f32c(global_scores, dev) if (self.use_global_score and global_scores is not None) else None
        o_rp = torch.empty((n, 3), dtype=torch.float32, device=dev)
        o_sp = torch.empty((n, 3), dtype=torch.float32, device=dev)
        o_sc = torch.empty((n,), dtype=torch.float32, device=dev)
        transforms = torch.eye(4, dtype=torch.float32, device=dev).repeat(B, 1, 1)
        rows = torch.zeros((B + 1,), dtype=torch.int32, device=dev)
        if P > 0 and B > 0:
            idx_dummy = torch.zeros((P, max(K1, K2)), dtype=torch.int64, device=dev)
            o_i = torch.empty((2, max(n, 1)), dtype=torch.int64, device=dev)
            d_poff = torch.tensor(poff, dtype=torch.int32).to(dev, non_blocking=False)
            with torch.cuda.device(dev):
                st = _lib.stream_ptr(dev)
                if n > 0:
                    _lib.check(L.gr_corr_gather(_lib.ptr(s), P, K1, K2, _lib.ptr(corr), _lib.ptr(rp), _lib.ptr(sp),
                                                _lib.ptr(idx_dummy), _lib.ptr(idx_dummy), _lib.ptr(gs), int(gs is not None),
                                                _lib.ptr(o_rp), _lib.ptr(o_sp), _lib.ptr(o_i[0]), _lib.ptr(o_i[1]),
                                                _lib.ptr(o_sc), _lib.ptr(pm_ws), pm_ws.numel(), st))
                ws2 = torch.empty(L.gr_lgr_workspace_bytes(P) + 256, dtype=torch.uint8, device=dev)
                _lib.check(L.gr_lgr_register_seg(_lib.ptr(o_rp), _lib.ptr(o_sp), _lib.ptr(o_sc), n, P, _lib.ptr(pm_ws),
                                                 _lib.ptr(d_poff), B, float(self.acceptance_radius),
                                                 int(self.correspondence_threshold), int(self.num_refinement_steps),
                                                 _lib.ptr(transforms), _lib.ptr(rows), _lib.ptr(ws2), ws2.numel(), st))
        return o_rp, o_sp, o_sc, transforms, rows
