"""Import-path shim: lets scripts written against the reference (`from vq_voice_swap.diffusion_model import
DiffusionModel`, ...) run on the gfx950 implementation unchanged for the accelerated sampling path."""
