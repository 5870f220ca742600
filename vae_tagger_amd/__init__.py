"""MI355X-native FLUX-VAE encode + tag hot path (drop-in for spawner1145/vae-tagger's inference path)."""
__version__ = "0.1.0"
