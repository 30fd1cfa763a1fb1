"""esmdiff_amd — MI355X (gfx950) engine for the ESMDiff sampling hot path.

Hand-written HIP kernels behind a C ABI (include/esmdiff_hip.h, esmdiff_amd/lib/libesmdiff_hip.so); this
package is the Python host side that mirrors the reference's call sites.  There is no CPU fallback.
"""
from .config import ESM3_OPEN, TINY, ModelConfig  # noqa: F401

__version__ = "0.1.0"
