"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatements of the reference algorithms on the north-star hot path, plus (when built)
``oracle/_ref``: the reference's own C++ core compiled from /root/reference.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package, and only as the checker / the timed CPU baseline.  Nothing under ``gaussreg_amd/``
(the product) imports it; the product raises if its HIP library is missing.
"""
