"""MI355X-native per-cell patch-inference hot path of RIBCA (multiplexed-image-annotator).

Host side is Python (the reference is pure Python); all compute on the path runs in
hand-written HIP kernels for gfx950 behind the C-ABI declared in ``include/ribca_hip.h``.
There is no CPU fallback: importing a compute entry point without the built shared
library raises.
"""
__version__ = "0.1.0"
