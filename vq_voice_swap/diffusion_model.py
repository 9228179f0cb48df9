from vq_voice_swap_amd.diffusion_model import DiffusionModel, make_predictor  # noqa: F401
