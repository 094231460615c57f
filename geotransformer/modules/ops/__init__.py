from gaussreg_amd.ops import (  # noqa: F401
    grid_subsample,
    index_select,
    pairwise_distance,
    point_to_node_partition,
    radius_search,
)
