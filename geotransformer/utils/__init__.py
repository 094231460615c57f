from gaussreg_amd._alias import chain as _chain

_chain(globals())   # sub-modules this repo does not override resolve to GaussReg's own package, if on sys.path
