"""Shared helpers for the parity tests."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CONFIGS = {
    "ref41": lambda M: M.ref41(),
    "pascucci": lambda M: M.pascucci(),
    "small2d": lambda M: M.small(),
    "small3d": lambda M: M.small(n_rad=12, nz=6, n_az=8, l3D=True),
}


def load_golden(name):
    return np.load(os.path.join(GOLDEN, f"geom_{name}.npz"))


def mc_similar(x, y, threshold, mask_threshold=0.0):
    """The reference's own Monte-Carlo-aware comparator
    (test_suite/test_mcfost.py:46-57): 75th percentile of |x-y|/x over pixels
    with |x| > mask must be below the threshold."""
    x = np.asarray(x, float).ravel()
    y = np.asarray(y, float).ravel()
    mask = np.abs(x) > mask_threshold
    p75 = np.percentile(np.abs(x[mask] - y[mask]) / x[mask], 75)
    return p75 < threshold, p75


def rel_rms(T, Tref, T_floor):
    sel = Tref > T_floor
    return float(np.sqrt(np.mean(((T[sel] - Tref[sel]) / Tref[sel]) ** 2)))
