from vq_voice_swap_amd.base import Savable, atomic_save  # noqa: F401
from vq_voice_swap_amd.diffusion_model import make_predictor  # noqa: F401
from vq_voice_swap_amd.unet import UNetEncoder, UNetPredictor  # noqa: F401
from vq_voice_swap_amd.vq_vae import make_encoder  # noqa: F401


class _NotBuilt:
    _what = ""

    def __init__(self, *a, **k):
        raise NotImplementedError(f"{self._what} is a 'next' row of the hot-path scope (SURVEY.md 8f.1) and is not built yet")

    @classmethod
    def load(cls, path):
        cls()


from vq_voice_swap_amd.classifier import Classifier, ClassifierStem  # noqa: E402,F401


class EncoderPredictor(_NotBuilt):
    _what = "EncoderPredictor (encoder-predictor guidance)"
