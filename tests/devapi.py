"""the ctypes view of include/kvz_hip_dev.h lives in the package"""
from kvazaar_amd.dev import TRANSFORM_KINDS, TRANSFORM_SIZE, Dev  # noqa: F401
