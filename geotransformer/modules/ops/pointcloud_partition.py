"""modules/ops/pointcloud_partition.py of GaussReg: `point_to_node_partition` (:61-111, the one the model calls,
model.py:99-104) is the HIP kernel; `get_point_to_node_indices`, `knn_partition`, `ball_query_partition` are re-exported
from GaussReg's own file when its checkout is on sys.path (they then run on this repo's pairwise_distance / index_select)."""
from gaussreg_amd._alias import shadowed_module as _shadowed
from gaussreg_amd.ops import point_to_node_partition  # noqa: F401

__all__ = ["point_to_node_partition"]
_u = _shadowed(__name__, __file__)
if _u is not None:
    for _n in ("get_point_to_node_indices", "knn_partition", "ball_query_partition"):
        if hasattr(_u, _n):
            globals()[_n] = getattr(_u, _n)
            __all__.append(_n)
