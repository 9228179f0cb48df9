"""
EncoderPredictor: predicts the VQ codes of an encoder from noised audio; used as a guidance model by
`VQVAE.decode(enc_pred=..., enc_pred_scale=...)` (reference `vq_voice_swap/models/encoder_predictor.py:14-75`,
`vq_vae.py:99-130`).

Parameters follow the reference's layout (`unet.*` = a UNetPredictor with `out_channels = bottleneck_dim`, `out.*` =
the 1x1 convolution to `num_latents` logits), so reference checkpoints load unchanged.  `forward` and `guidance_fn`
run on the gfx950 library (`vqvs_encpred_forward`, `vqvs_encpred_guidance`): the UNet forward is the fused ResBlock
schedule with every intermediate kept resident, and the gradient of the summed cross-entropy with respect to x_t is
an explicit backward schedule through the whole UNet (concatenating, up- and down-sampling blocks) -- no autograd.
There is no CPU path.
"""

from __future__ import annotations

from typing import Any, Dict

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _native
from .base import Savable
from .unet import UNetPredictor, _NativeModule


class EncoderPredictor(_NativeModule, Savable):
    def __init__(self, base_channels: int, downsample_rate: int, num_latents: int, bottleneck_dim: int = 64):
        super().__init__()
        from .unet import check_base_channels

        check_base_channels(base_channels)
        self.base_channels = base_channels
        self.downsample_rate = downsample_rate
        self.num_latents = num_latents
        self.bottleneck_dim = bottleneck_dim
        self.unet = UNetPredictor(base_channels, out_channels=bottleneck_dim)  # parameter container (encoder_predictor.py:40)
        self.out = nn.Conv1d(bottleneck_dim, num_latents, 1)

    def save_kwargs(self) -> Dict[str, Any]:
        return dict(base_channels=self.base_channels, downsample_rate=self.downsample_rate, num_latents=self.num_latents,
                    bottleneck_dim=self.bottleneck_dim)

    def set_precision(self, precision: str):
        super().set_precision(precision)
        self.unet.set_precision(precision)
        return self

    def _cfg(self) -> _native.Cfg:
        cfg = _native.Cfg()
        cfg.kind = _native.KIND_ENCPRED
        cfg.base_channels = self.base_channels
        cfg.in_channels = 1
        cfg.out_channels = self.bottleneck_dim
        cfg.reserved[1] = self.downsample_rate
        cfg.reserved[2] = self.num_latents
        return cfg

    def _prepare(self, x: torch.Tensor, ts: torch.Tensor):
        _native.require_cuda(x, ts)
        if x.dim() != 3 or x.shape[1] != 1:
            raise ValueError(f"expected x of shape [N, 1, T], got {tuple(x.shape)}")
        B, _, T = x.shape
        if T % 256 or T % self.downsample_rate:
            raise ValueError(f"T={T} must be a multiple of 256 and of the downsample rate {self.downsample_rate}")
        x = x.detach().to(torch.float32).contiguous()
        ts = ts.detach().to(device=x.device, dtype=torch.float32).contiguous()
        if ts.shape != (B,):
            raise ValueError(f"expected ts of shape [{B}], got {tuple(ts.shape)}")
        return x, ts, B, T

    def forward(self, x: torch.Tensor, ts: torch.Tensor, use_checkpoint: bool = False) -> torch.Tensor:
        """[N,1,T] -> logits [N, num_latents, T // downsample_rate] (encoder_predictor.py:43-58)."""
        x, ts, B, T = self._prepare(x, ts)
        h = self.handle(x.device, B, T)
        logits = torch.empty(B, self.num_latents, T // self.downsample_rate, device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _native.check(_native.lib().vqvs_encpred_forward(h.ptr, x.data_ptr(), ts.data_ptr(), logits.data_ptr(), B, T,
                                                            _native._stream_ptr()))
        return logits

    def losses(self, x: torch.Tensor, ts: torch.Tensor, targets: torch.Tensor, **kwargs) -> torch.Tensor:
        """Per-clip mean cross-entropy (encoder_predictor.py:60-64); values only -- the gradient path is `guidance_grad`."""
        return F.cross_entropy(self(x, ts), targets, reduction="none").mean(-1)

    def guidance_grad(self, x: torch.Tensor, ts: torch.Tensor, targets: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
        """-scale * d/dx sum_{n,i} cross_entropy(logits[n,:,i], targets[n,i])   (vq_vae.py:125-130)."""
        x, ts, B, T = self._prepare(x, ts)
        _native.require_cuda(targets)
        targets = targets.detach().to(device=x.device, dtype=torch.int64).contiguous()
        if tuple(targets.shape) != (B, T // self.downsample_rate):
            raise ValueError(f"expected targets of shape {(B, T // self.downsample_rate)}, got {tuple(targets.shape)}")
        h = self.handle(x.device, B, T)
        grad = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _native.check(_native.lib().vqvs_encpred_guidance(h.ptr, x.data_ptr(), ts.data_ptr(), targets.data_ptr(), float(scale),
                                                             grad.data_ptr(), None, B, T, _native._stream_ptr()))
        return grad

    def guidance_fn(self, targets: torch.Tensor, scale: float = 1.0):
        def cond_fn(x, ts):
            return self.guidance_grad(x, ts, targets.to(x.device), scale)

        cond_fn.native_modules = (self,)  # (see Classifier.guidance_fn)
        return cond_fn
