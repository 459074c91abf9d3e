"""codae — MI355X-native build of CODAE's denoising-autoencoder training path.

Same package / class names as the reference (victordeleau/MUI-DeepAutoEncoder) so that
its training scripts import this package unchanged; the arithmetic runs in
libcodae_hip.so (hand-written HIP for gfx950), reached through `codae.hip`.
"""
__all__ = ["model", "tool", "dataset", "hip"]
