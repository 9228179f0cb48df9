from vq_voice_swap_amd.vq import VQ  # noqa: F401
