"""CPU oracle for the MCFOST continuum Monte Carlo packet loop.

TEST INFRASTRUCTURE ONLY: imported by ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg as the checker.  The product
(``mcfost_amd``) never imports this package.
"""
from .binding import (  # noqa: F401
    Oracle,
    RefGeom,
    build_oracle,
    build_ref,
    oracle_lib_path,
    ref_lib_path,
)
