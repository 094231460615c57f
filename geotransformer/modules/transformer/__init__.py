from gaussreg_amd._alias import chain as _chain

_chain(globals())   # sub-modules this repo does not override resolve to GaussReg's own package, if on sys.path
from gaussreg_amd.embedding import SinusoidalPositionalEmbedding  # noqa: E402,F401
from gaussreg_amd.rpe_attention import RPEMultiHeadAttention  # noqa: E402,F401
from gaussreg_amd.transformer import (  # noqa: E402,F401
    AttentionLayer,
    AttentionOutput,
    MultiHeadAttention,
    RPEAttentionLayer,
    RPEConditionalTransformer,
    RPETransformerLayer,
    TransformerLayer,
)
