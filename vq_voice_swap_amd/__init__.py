"""
vq_voice_swap_amd: MI355X (gfx950) native DDPM audio sampler behind the
DiffusionModel / VQVAE API of unixpickle/vq-voice-swap.  See DESIGN.md.
"""
# The package does NOT touch process-wide HIP runtime switches.  One of them matters for speed: HIP_FORCE_DEV_KERNARG=1 (kernel
# arguments in device memory; the default on this stack keeps them in host memory, and every launch of the 225-kernel forward then
# starts with scalar loads across PCIe: 2.3 % of the headline throughput).  It is read when the HIP runtime
# initialises, so it belongs to the PROCESS: bench.py, the sample_*.py scripts, __graft_entry__.py and tests/conftest.py set it before
# they import torch; a host application sets it in its own environment.  _native.lib() warns once when it is unset.
from .base import Savable, atomic_save
from .classifier import Classifier
from .conv_encoder import ConvMFCCEncoder
from .diffusion import CosSchedule, Diffusion, ExpSchedule, Schedule, make_schedule, randn_clips
from .diffusion_model import DiffusionModel
from .encoder_predictor import EncoderPredictor
from .unet import ResBlockModule, UNetEncoder, UNetPredictor
from .vq import VQ
from .vq_vae import VQVAE

__all__ = [
    "Savable", "atomic_save", "CosSchedule", "Diffusion", "ExpSchedule", "Schedule", "make_schedule", "randn_clips",
    "DiffusionModel", "Classifier", "ConvMFCCEncoder", "EncoderPredictor", "ResBlockModule", "UNetEncoder", "UNetPredictor", "VQ", "VQVAE",
]
