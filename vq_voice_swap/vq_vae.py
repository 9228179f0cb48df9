from vq_voice_swap_amd.vq_vae import VQVAE, make_encoder  # noqa: F401
