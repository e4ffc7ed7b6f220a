"""eddy_currents_3d_amd — MI355X-native BiCGSTAB-with-restart solver and matrix assembly for the
A–V eddy-current system of JNSresearcher/eddy_currents_3d (hot path only).

The product is ``libec3d_hip.so`` (HIP, gfx950) behind the C ABI in ``include/ec3d_hip.h``;
this package is the thin Python host over it.  There is no CPU path: without the built library
and a HIP device every call raises.
"""
from .solver import EC3DSolver, EC3DMulti, EC3DError, sprsBCGstabWR, load_library, probe_csr, probe_csr_multi  # noqa: F401

__all__ = ["EC3DSolver", "EC3DMulti", "EC3DError", "sprsBCGstabWR", "load_library", "probe_csr", "probe_csr_multi"]
