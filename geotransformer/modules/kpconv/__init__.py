from gaussreg_amd._alias import chain as _chain

_chain(globals())   # sub-modules this repo does not override resolve to GaussReg's own package, if on sys.path
from gaussreg_amd.kpconv import KPConv, load_kernels, maxpool, nearest_upsample  # noqa: E402,F401
from gaussreg_amd.kpconv_blocks import (  # noqa: E402,F401
    ConvBlock,
    GroupNorm,
    KPConvFPN,
    LastUnaryBlock,
    ResidualBlock,
    UnaryBlock,
)
