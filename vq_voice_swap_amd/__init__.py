"""
vq_voice_swap_amd: MI355X (gfx950) native DDPM audio sampler behind the
DiffusionModel / VQVAE API of unixpickle/vq-voice-swap.  See DESIGN.md.
"""
import os as _os

# Kernel arguments in device memory (a HIP runtime switch, read when the runtime initialises, i.e. at the process's first HIP call):
# the default on this stack keeps them in host memory, and every launch of the 225-kernel forward then starts with scalar loads across
# PCIe -- measured 2.3 % of the headline throughput and 3.7 % of a unet32 forward.  Harmless if the
# runtime is already up (then whatever the process was started with stays in force).
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

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
