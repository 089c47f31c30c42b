"""s2p_amd -- MI355X-native S2P hot path (state-conditioned SPADE/MAT generator + multi-scale PatchGAN
discriminator + GAN/feature-matching/VGG/L1 train step).  Python host code over libs2p_hip.so (HIP, gfx950)."""
__version__ = "0.1.0"
