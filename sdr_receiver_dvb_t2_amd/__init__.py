"""MI355X-native DVB-T2 demodulation / FEC back-end (host-side Python binding of libt2gpu.so).

The product is the HIP library behind ``include/t2gpu.h``; this package only loads it and mirrors the
reference's per-stage objects (``src/DVB_T2/*``) on top of the C ABI. Importing the package never touches
the GPU; constructing a stage object does, and raises if the library or a device is missing (there is no
CPU fallback anywhere in this package).
"""
from ._lib import lib, T2GpuError, library_path  # noqa: F401
from .ldpc import ldpc_decoder, FECFRAME_SHORT, FEC_FRAME_NORMAL, C1_2, C3_5, C2_3, C3_4, C4_5, C5_6  # noqa: F401

from .fec import llr_demapper, time_deinterleaver, bch_decoder  # noqa: F401
from .ofdm import t2_ofdm  # noqa: F401
from .chain import t2_chain  # noqa: F401

__all__ = ["lib", "T2GpuError", "library_path", "ldpc_decoder", "llr_demapper", "time_deinterleaver", "bch_decoder", "t2_ofdm", "t2_chain"]
