from gaussreg_amd.embedding import SinusoidalPositionalEmbedding  # noqa: F401
from gaussreg_amd.rpe_attention import RPEMultiHeadAttention  # noqa: F401
from gaussreg_amd.transformer import (  # noqa: F401
    AttentionLayer,
    AttentionOutput,
    MultiHeadAttention,
    RPEAttentionLayer,
    RPEConditionalTransformer,
    RPETransformerLayer,
    TransformerLayer,
)
