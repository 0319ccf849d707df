"""f2cnn_amd -- MI355X (gfx950) implementation of the F2CNN hot path:
gammatone filterbank -> Hilbert/LPF envelope -> 11xC window gather -> CNN forward.

Python here is the host-side mirror of the reference's function/CLI surface; all array work is done by
hand-written HIP kernels in lib/libf2cnn_hip.so, reached through the C ABI of include/f2cnn_hip.h.
"""
from ._lib import Context, F2Error, default_context, load  # noqa: F401

__version__ = "0.1.0"
