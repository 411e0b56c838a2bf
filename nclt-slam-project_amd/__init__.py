"""MI355X-native visual relocalization (ORB -> brute-force Hamming -> PnP-RANSAC).

The directory is named after the reference repository (`nclt-slam-project_amd`); because a
hyphen is not importable, the repository root ships `nclt_slam_project_amd.py`, which loads this
directory as the package `nclt_slam_project_amd`.
"""
from ._native import RelocError, LIB_PATH  # noqa: F401

__all__ = ["RelocError", "LIB_PATH"]
