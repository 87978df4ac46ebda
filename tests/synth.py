"""the workload generator lives in the package (bench.py uses it too)"""
from kvazaar_amd.synth import MD5, frames, write_yuv  # noqa: F401
