from gaussreg_amd.kpconv_blocks import ConvBlock, GroupNorm, LastUnaryBlock, ResidualBlock, UnaryBlock  # noqa: F401
