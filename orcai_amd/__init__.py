"""orcai_amd -- MI355X-native (gfx950) implementation of orcAI's spectrogram -> label hot path.

Mirrors the reference's Python function API for that path (ethz-tb/orcAI v1.0.3,
``src/orcAI/{spectrogram,predict,architectures,train,io,auxiliary}.py``); the arithmetic runs in
hand-written HIP kernels behind the C ABI in ``include/orcai_hip.h``.  There is no CPU fallback.
"""

__version__ = "0.1.0"
