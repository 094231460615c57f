"""Mirror of ``geotransformer.modules.ops`` for the hot-path operators (HIP-backed).

reference: geotransformer/modules/ops/grid_subsample.py:7-22, radius_search.py:7-27.
"""
from . import ext


def grid_subsample(points, lengths, voxel_size, order="reference"):
    """Grid subsampling in stack mode -> (s_points (M,3), s_lengths (B,)).  grid_subsample.py:7-22."""
    s_points, s_lengths = ext.grid_subsampling(points, lengths, voxel_size, order=order)
    return s_points, s_lengths


def radius_search(q_points, s_points, q_lengths, s_lengths, radius, neighbor_limit):
    """Radius search in stack mode -> (N, min(max_count, neighbor_limit)) int64, padded with M.

    radius_search.py:7-27.  The reference materialises the full width and returns a NON-contiguous
    column slice (radius_search.py:26); here the truncation happens inside the fill kernel, so only
    the kept columns are ever written and the result is contiguous (what KPConv's index_select
    needs once the CPU->GPU copy that used to re-densify it is gone -- SURVEY.md section 8b)."""
    for t, n in ((q_points, "q_points"), (s_points, "s_points")):
        ext._check_points(t, n)
        ext._check_float(t, n)
        ext._check_contig(t, n)
    for t, n in ((q_lengths, "q_lengths"), (s_lengths, "s_lengths")):
        ext._check_long(t, n)
        ext._check_contig(t, n)
    return ext.radius_neighbors_limited(q_points, s_points, q_lengths, s_lengths, radius, neighbor_limit)
