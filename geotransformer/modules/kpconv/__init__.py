from gaussreg_amd.kpconv import KPConv, load_kernels, maxpool, nearest_upsample  # noqa: F401
from gaussreg_amd.kpconv_blocks import (  # noqa: F401
    ConvBlock,
    GroupNorm,
    KPConvFPN,
    LastUnaryBlock,
    ResidualBlock,
    UnaryBlock,
)
