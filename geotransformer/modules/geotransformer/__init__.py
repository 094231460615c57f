"""geotransformer/modules/geotransformer/__init__.py:1-5 of GaussReg: the same five names."""
from gaussreg_amd._alias import chain as _chain

_chain(globals())   # sub-modules this repo does not override resolve to GaussReg's own package, if on sys.path
from gaussreg_amd._alias import upstream_names as _up

from gaussreg_amd.embedding import GeometricStructureEmbedding  # noqa: E402,F401
from gaussreg_amd.matching import LocalGlobalRegistration, PointMatching, SuperPointMatching  # noqa: E402,F401
from gaussreg_amd.transformer import GeometricTransformer  # noqa: E402,F401

# training-only target generator (superpoint_target.py:6-46): GaussReg's own file, when its checkout is on sys.path
_up(__name__, "superpoint_target", ["SuperPointTargetGenerator"], globals())
