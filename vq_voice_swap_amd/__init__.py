"""
vq_voice_swap_amd: MI355X (gfx950) native DDPM audio sampler behind the
DiffusionModel / VQVAE API of unixpickle/vq-voice-swap.  See DESIGN.md.
"""

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
