"""ctypes binding of libgaussreg_hip.so (include/gaussreg_hip.h).  Fails loudly."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libgaussreg_hip.so")

_lib = None

c_void = ctypes.c_void_p
c_i64 = ctypes.c_int64
c_f32 = ctypes.c_float
c_int = ctypes.c_int
c_size = ctypes.c_size_t
c_i64p = ctypes.POINTER(ctypes.c_int64)

# name -> (restype, argtypes); must list every symbol include/gaussreg_hip.h declares
SIGNATURES = {
    "gr_last_error": (ctypes.c_char_p, []),
    "gr_version": (c_int, []),
    "gr_timing_enable": (None, [c_int]),
    "gr_timing_reset": (None, []),
    "gr_timing_read": (c_int, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_double), c_i64p]),
    "gr_radius_workspace_bytes": (c_size, [c_i64, c_i64, c_i64]),
    "gr_radius_count": (c_int, [c_void, c_void, c_i64p, c_i64p, c_i64, c_i64, c_i64, c_f32, c_void, c_size,
                                c_i64p, c_void]),
    "gr_radius_count_cached": (c_int, [c_void, c_void, c_i64p, c_i64p, c_i64, c_i64, c_i64, c_f32, c_void, c_size,
                                c_i64p, c_i64p, c_int, c_void]),
    "gr_radius_fill": (c_int, [c_void, c_void, c_i64, c_i64, c_i64, c_f32, c_i64, c_i64p, c_void, c_void,
                               c_size, c_void]),
    "gr_radius_search": (c_int, [c_void, c_void, c_i64p, c_i64p, c_i64, c_i64, c_i64, c_f32, c_i64, c_void, c_void, c_size,
                                 c_i64p, c_i64p, c_int, c_void]),
    "gr_radius_search_mode": (c_int, [c_int]),
    "gr_host_unordered_map_order": (c_int, [ctypes.POINTER(ctypes.c_uint64), c_i64, ctypes.POINTER(ctypes.c_int32)]),
    "gr_hash_order_device_workspace_bytes": (c_size, [c_i64, c_i64]),
    "gr_hash_order_device": (c_int, [c_void, c_i64p, c_i64, c_void, c_void, c_size, c_void]),
    "gr_grid_subsample_workspace_bytes": (c_size, [c_i64, c_i64]),
    "gr_grid_subsample": (c_int, [c_void, c_i64p, c_i64, c_i64, c_f32, c_int, c_void, c_i64p, c_i64p, c_void,
                                  c_size, c_void]),
}


class RasterView(ctypes.Structure):
    """struct gr_raster_view (include/gaussreg_hip.h)."""
    _fields_ = [("image_height", ctypes.c_int32), ("image_width", ctypes.c_int32),
                ("tanfovx", c_f32), ("tanfovy", c_f32), ("bg", c_f32 * 3), ("scale_modifier", c_f32),
                ("viewmatrix", c_f32 * 16), ("projmatrix", c_f32 * 16), ("campos", c_f32 * 3),
                ("sh_degree", ctypes.c_int32), ("prefiltered", ctypes.c_int32), ("debug", ctypes.c_int32)]


SIGNATURES.update({
    "gr_raster_geom_bytes": (c_size, [c_i64, c_int, c_int, c_int]),
    "gr_raster_bin_bytes": (c_size, [c_i64, c_int, c_int, c_int]),
    "gr_raster_debug_geom_layout": (c_int, [c_i64, c_int, c_int, c_int, c_void]),
    "gr_raster_debug_bucket_cooldown": (c_int, [c_int]),
    "gr_raster_preprocess": (c_int, [c_i64, c_int] + [c_void] * 7 + [ctypes.POINTER(RasterView), c_int, c_void,
                                                                     c_void, c_size, c_i64p, c_void]),
    "gr_raster_render": (c_int, [c_i64, ctypes.POINTER(RasterView), c_int, c_i64p, c_void, c_size, c_void, c_size,
                                 c_void, c_void]),
    "gr_raster_render_ex": (c_int, [c_i64, ctypes.POINTER(RasterView), c_int, c_i64p, c_void, c_size, c_void, c_size,
                                    c_void, c_int, c_void]),
    "gr_raster_forward": (c_int, [c_i64, c_int] + [c_void] * 7 + [ctypes.POINTER(RasterView), c_int, c_void, c_void, c_size,
                                  c_void, c_size, c_void, c_int, c_i64p, c_void]),
    "gr_raster_forward_finish": (c_int, [c_i64p]),
    "gr_raster_lds_atomics_lane_ordered": (c_int, []),
    "gr_raster_ballot_ranking": (c_int, [c_int]),
    "gr_raster_mark_visible": (c_int, [c_i64, c_void, ctypes.POINTER(c_f32), c_void, c_void]),
})


SIGNATURES.update({
    "gr_sinkhorn_workspace_bytes": (c_size, [c_i64]),
    "gr_sinkhorn": (c_int, [c_void, c_i64, c_i64, c_i64, c_void, c_void, c_void, c_int, c_f32, c_int, c_void, c_void, c_size,
                            c_void]),
    "gr_kpconv_workspace_bytes": (c_size, [c_i64, c_i64, c_i64, c_i64]),
    "gr_kpconv_forward": (c_int, [c_void] * 4 + [c_i64] * 5 + [c_void, c_i64, c_void, c_void, c_f32, c_f32, c_void,
                                                            c_void, c_size, c_void]),
    "gr_gather_rows": (c_int, [c_void, c_i64, c_i64, c_void, c_i64, c_void, c_void, c_void]),
    "gr_neighbor_pool": (c_int, [c_void, c_i64, c_i64, c_void, c_i64, c_i64, c_int, c_void, c_void]),
    "gr_group_norm_workspace_bytes": (c_size, [c_i64]),
    "gr_group_norm": (c_int, [c_void, c_i64, c_i64, c_i64, c_void, c_void, c_f32, c_f32, c_void, c_void, c_size, c_void]),
    "gr_group_norm_seg_workspace_bytes": (c_size, [c_i64, c_i64]),
    "gr_group_norm_seg": (c_int, [c_void, c_i64, c_i64, c_i64, c_void, c_void, c_f32, c_f32, c_void, c_void, c_i64, c_i64, c_void,
                                  c_size, c_void]),
    "gr_group_norm_res": (c_int, [c_void, c_i64, c_i64, c_i64, c_void, c_void, c_f32, c_f32, c_void, c_void, c_void, c_i64, c_i64,
                                  c_void, c_size, c_void]),
    "gr_gs_fuse_workspace_bytes": (c_size, [c_i64, c_i64]),
    "gr_gs_fuse": (c_int, [c_void, c_i64, c_void, c_i64, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                           ctypes.c_double, ctypes.POINTER(c_f32), ctypes.POINTER(c_f32), ctypes.POINTER(c_f32), c_void,
                           c_i64p, c_void, c_size, c_void]),
    "gr_pairwise_distance_workspace_bytes": (c_size, [c_i64, c_i64]),
    "gr_pairwise_distance": (c_int, [c_void, c_void, c_i64, c_i64, c_i64, c_int, c_void, c_void, c_size, c_void]),
    "gr_fps_debug_force_fallback": (c_int, [c_int]),
    "gr_fps_debug_bucket_sort": (c_int, [c_int]),
    "gr_hash_order_debug_force_prescan": (c_int, [c_int]),
    "gr_grid_subsample_debug_bucket_sort": (c_int, [c_int]),
    "gr_standin_descriptors": (c_int, [c_void, c_i64, c_void, c_void, c_void, c_void, c_void, c_i64, c_f32, c_int, c_void, c_void]),
    "gr_pairwise_distance_batch_workspace_bytes": (c_size, [c_i64, c_i64, c_i64]),
    "gr_pairwise_distance_batch": (c_int, [c_void, c_void, c_i64, c_i64, c_i64, c_i64, c_int, c_void, c_void, c_size, c_void]),
    "gr_superpoint_matching_workspace_bytes": (c_size, [c_i64, c_i64]),
    "gr_superpoint_matching": (c_int, [c_void, c_void, c_i64, c_i64, c_i64, c_void, c_void, c_int, c_int, c_void,
                                       c_void, c_void, c_i64p, c_void, c_size, c_void]),
    "gr_point_matching_workspace_bytes": (c_size, [c_i64]),
    "gr_corr_matrix": (c_int, [c_void, c_i64, c_i64, c_i64, c_void, c_void, c_int, c_int, c_f32, c_void, c_i64p,
                               c_void, c_size, c_void]),
    "gr_corr_matrix_exp": (c_int, [c_void, c_i64, c_i64, c_i64, c_void, c_void, c_int, c_int, c_f32, c_void, c_i64p,
                                   c_void, c_size, c_void]),
    "gr_corr_gather": (c_int, [c_void, c_i64, c_i64, c_i64] + [c_void] * 6 + [c_int] + [c_void] * 5 +
                       [c_void, c_size, c_void]),
    "gr_lgr_workspace_bytes": (c_size, [c_i64]),
    "gr_lgr_register": (c_int, [c_void, c_void, c_void, c_i64, c_i64, c_void, c_f32, c_int, c_int, c_void, c_void,
                                c_size, c_void]),
    "gr_lgr_register_verify": (c_int, [c_void, c_void, c_void, c_i64, c_i64, c_void, c_void, c_void, c_void, c_i64,
                                       c_f32, c_int, c_int, c_void, c_void, c_size, c_void]),
    "gr_ransac_sample_hash": (ctypes.c_uint32, [ctypes.c_uint32] * 4),
    "gr_ransac_workspace_bytes": (c_size, [c_i64]),
    "gr_ransac_similarity": (c_int, [c_void, c_void, c_i64, c_int, c_i64, ctypes.c_uint32, c_f32, c_int, c_int, c_void,
                                     c_void, c_void, c_size, c_void]),
    "gr_geo_embedding_workspace_bytes": (c_size, [c_i64, c_i64]),
    "gr_geo_embedding": (c_int, [c_void, c_i64, c_void, c_void, c_void, c_void, c_void, c_i64, c_f32, c_f32, c_i64,
                                 c_int, c_void, c_void, c_size, c_void]),
    "gr_geo_embedding_table": (c_int, [c_void, c_i64, c_void, c_i64, c_void, c_i64, c_f32, c_void, c_void, c_void, c_void, c_void,
                                       c_i64, c_f32, c_f32, c_i64, c_int, c_void, c_void, c_size, c_void]),
    "gr_rpe_attention": (c_int, [c_void] * 9 + [c_i64] * 4 + [c_void, c_void, c_void]),
    "gr_rpe_scores": (c_int, [c_void, c_void, c_void, c_i64, c_i64, c_i64, c_i64, c_void, c_void]),
    "gr_fps_workspace_bytes": (c_size, [c_i64, c_i64]),
    "gr_fps": (c_int, [c_void, c_i64p, c_i64p, c_i64p, c_i64, c_i64, c_void, c_void, c_size, c_void]),
    "gr_point_to_node_workspace_bytes": (c_size, [c_i64, c_i64]),
    "gr_point_to_node_batch_workspace_bytes": (c_size, [c_i64p, c_i64p, c_i64]),
    "gr_point_to_node_partition_batch": (c_int, [c_void, c_i64p, c_void, c_i64p, c_i64, c_int, c_void, c_void, c_void, c_void,
                                                 c_void, c_size, c_void]),
    "gr_superpoint_matching_batch_workspace_bytes": (c_size, [c_i64p, c_i64]),
    "gr_superpoint_matching_batch": (c_int, [c_void, c_i64p, c_i64, c_i64, c_void, c_int, c_int, c_void, c_void, c_void,
                                             c_i64p, c_void, c_size, c_void]),
    "gr_lgr_register_seg": (c_int, [c_void, c_void, c_void, c_i64, c_i64, c_void, c_void, c_i64, c_f32, c_int, c_int,
                                    c_void, c_void, c_void, c_size, c_void]),
    "gr_ransac_seg_workspace_bytes": (c_size, [c_i64, c_i64]),
    "gr_ransac_similarity_seg": (c_int, [c_void, c_void, c_void, c_i64, c_int, c_i64, ctypes.c_uint32, c_f32, c_int, c_int,
                                         c_void, c_void, c_void, c_void, c_size, c_void]),
    "gr_point_to_node_partition": (c_int, [c_void, c_i64, c_void, c_i64, c_int, c_void, c_void, c_void, c_void,
                                           c_void, c_size, c_void]),
})


class HipLibraryError(RuntimeError):
    pass


def lib():
    """Load the HIP library; raise (never fall back) if it has not been built."""
    global _lib
    if _lib is None:
        # torch first: libgaussreg_hip.so must bind to the SAME HIP runtime instance torch uses (the
        # wheel bundles its own libamdhip64); loading ours first leaves two runtimes in the process
        # and ours then reports "no ROCm-capable device".
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise HipLibraryError(
                f"{LIB_PATH} is missing: build it with `python -m gaussreg_amd.build` "
                "(there is no CPU fallback for the gaussreg_amd ops)")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


GR_PENDING = 2     # gr_raster_forward(GR_RASTER_SPLIT): enqueued, gr_raster_forward_finish collects the counts
GR_RETRY_FULL = 3  # gr_raster_forward_finish: repeat the frame with an unsplit gr_raster_forward (depth >= 8192)
GR_RETRY_BIN = 1   # include/gaussreg_hip.h: gr_raster_forward's "bin buffer too small, call gr_raster_render_ex" status


def check(rc, allow=()):
    """Strict: every status but 0 raises, except the positive codes the call site names in `allow`."""
    if rc != 0 and rc not in allow:
        msg = lib().gr_last_error()
        raise RuntimeError("gaussreg_hip: " + (msg.decode() if msg else f"error {rc}"))
    return rc


def tensor_stamp(tensors):
    """Cache key for objects derived from tensors that may be updated in place: (storage address, version counter) per
    tensor, or None when any of them is an inference tensor (torch.inference_mode(): no version counter exists, reading
    `_version` raises) -- the caller then rebuilds instead of caching."""
    out = []
    for t in tensors:
        if t.is_inference():
            return None
        out.append((t.data_ptr(), t._version))
    return tuple(out)


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise HipLibraryError("gaussreg_amd ops need an MI355X (torch.cuda.is_available() is False); "
                              "there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


_ws_cache = {}


def workspace(device, nbytes):
    """Grow-only scratch buffer per (device, current stream, Python thread): two streams or two threads on one GPU never
    share scratch (the C library is re-entrant across streams only with distinct workspaces)."""
    import threading

    import torch
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream, threading.get_ident())
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def stream_ptr(device):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def host_i64(values):
    arr = (ctypes.c_int64 * max(len(values), 1))(*[int(v) for v in values])
    return arr
