from gaussreg_amd.embedding import GeometricStructureEmbedding  # noqa: F401
from gaussreg_amd.transformer import GeometricTransformer  # noqa: F401
