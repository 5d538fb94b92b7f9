"""melo_gan_amd -- MI355X-native implementation of Melo-GAN's GAN-training hot path.

Host side (Python, mirrors the reference's module surface) over libmelogan_hip.so, a C-ABI
library of hand-written gfx950 HIP kernels (include/melo_gan_hip.h).  There is no CPU or
PyTorch-eager fallback: every op raises if the library is missing.
"""
__version__ = "0.1.0"
