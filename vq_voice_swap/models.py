from vq_voice_swap_amd.base import Savable, atomic_save  # noqa: F401
from vq_voice_swap_amd.diffusion_model import make_predictor  # noqa: F401
from vq_voice_swap_amd.unet import UNetEncoder, UNetPredictor  # noqa: F401
from vq_voice_swap_amd.vq_vae import make_encoder  # noqa: F401


from vq_voice_swap_amd.classifier import Classifier, ClassifierStem  # noqa: E402,F401
from vq_voice_swap_amd.encoder_predictor import EncoderPredictor  # noqa: E402,F401
