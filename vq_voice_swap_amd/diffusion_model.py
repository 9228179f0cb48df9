"""
DiffusionModel facade: a UNet predictor bound to its diffusion process
(reference vq_voice_swap/diffusion_model.py:9-58), same constructor, attributes
and checkpoint kwargs; the predictor and the sampler run in libvqvs_hip.so.
"""

from __future__ import annotations

from typing import Any, Dict, Optional

import torch

from .base import Savable
from .diffusion import Diffusion, make_schedule
from .unet import UNetPredictor


def make_predictor(pred_name: str, base_channels: int = 32, num_labels: Optional[int] = None,
                   cond_channels: Optional[int] = None, dropout: float = 0.0):
    if pred_name == "unet":
        return UNetPredictor(base_channels=base_channels, cond_channels=cond_channels, num_labels=num_labels, dropout=dropout)
    if pred_name == "wavegrad":
        raise ValueError("predictor 'wavegrad' is outside the accelerated hot path (SURVEY.md section 2a); only 'unet' is built")
    raise ValueError(f"unknown predictor: {pred_name}")


class DiffusionModel(Savable):
    def __init__(self, pred_name: str, base_channels: int, schedule_name: str = "exp", num_labels: Optional[int] = None,
                 cond_channels: Optional[int] = None, dropout: float = 0.0):
        super().__init__()
        self.pred_name = pred_name
        self.base_channels = base_channels
        self.schedule_name = schedule_name
        self.num_labels = num_labels
        self.cond_channels = cond_channels
        # some reference checkpoints stored dropout as a 1-tuple (diffusion_model.py:30-31)
        self.dropout = dropout[0] if isinstance(dropout, tuple) else dropout
        self.predictor = make_predictor(pred_name, base_channels=base_channels, cond_channels=cond_channels,
                                        num_labels=num_labels, dropout=self.dropout)
        self.diffusion = Diffusion(make_schedule(schedule_name))

    def forward(self, *args, **kwargs) -> torch.Tensor:
        return self.predictor(*args, **kwargs)

    def set_precision(self, precision: str):
        self.predictor.set_precision(precision)
        return self

    def add_labels(self, n: int, end: bool = True):
        assert self.num_labels is not None, "model must be class-conditional"
        self.predictor.add_labels(n, end=end)
        self.num_labels += n

    def save_kwargs(self) -> Dict[str, Any]:
        return dict(pred_name=self.pred_name, base_channels=self.base_channels, schedule_name=self.schedule_name,
                    num_labels=self.num_labels, cond_channels=self.cond_channels, dropout=self.dropout)
