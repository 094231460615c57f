"""geotransformer/modules/ops/__init__.py:1-22 of GaussReg: the same names.  The operators on the hot path are this
repo's HIP kernels; the small torch helpers it does not replace (transformation.py, vector_angle.py and the three
partition variants the model does not call) come from GaussReg's own files when its checkout is on sys.path."""
from gaussreg_amd._alias import chain as _chain

_chain(globals())   # sub-modules this repo does not override resolve to GaussReg's own package, if on sys.path
from gaussreg_amd._alias import upstream_names as _up

from geotransformer.modules.ops.grid_subsample import grid_subsample  # noqa: E402,F401
from geotransformer.modules.ops.index_select import index_select  # noqa: E402,F401
from geotransformer.modules.ops.pairwise_distance import pairwise_distance  # noqa: E402,F401
from geotransformer.modules.ops.radius_search import radius_search  # noqa: E402,F401
from geotransformer.modules.ops.pointcloud_partition import *  # noqa: E402,F401,F403
from geotransformer.modules.ops.pointcloud_partition import point_to_node_partition  # noqa: E402,F401

_up(__name__, "transformation",
    ["apply_transform", "apply_rotation", "inverse_transform", "skew_symmetric_matrix", "rodrigues_rotation_matrix",
     "rodrigues_alignment_matrix", "get_transform_from_rotation_translation", "get_rotation_translation_from_transform",
     "get_rotation_translation_from_transform_w_scale"], globals())
_up(__name__, "vector_angle", ["vector_angle", "rad2deg", "deg2rad"], globals())
