"""MI355X-native conditional-VAE training and Monte-Carlo scoring path (gfx950 HIP kernels behind a C-ABI)."""
