"""lvae -- MI355X-native latent-analysis hot path of the TransformerVAE.

Host-side mirror of the reference's estimator functions (same names and tensor signatures) over
hand-written HIP kernels reached through the C ABI in include/lvae.h.
"""
__version__ = "0.1.0"
