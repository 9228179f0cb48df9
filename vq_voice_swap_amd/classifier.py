"""
Noised-audio speaker classifier used for classifier-guided sampling (BASELINE config 5).

Parameter layout and semantics follow the reference (`vq_voice_swap/models/classifier.py:18-191`): a stem
of 27 FiLM ResBlocks (every level ends with a x0.5 block, so 64000 samples become 125 tokens), GroupNorm +
GELU, an attention pool whose only consumed output is the prepended zero token, GELU and a linear head.

`Classifier.forward` and `Classifier.guidance_fn` run on the gfx950 library (`vqvs_classifier_forward`,
`vqvs_classifier_guidance`): the forward pass reuses the fused ResBlock schedule of the UNets, and the gradient
of log p(y | x_t) with respect to x_t -- which the reference obtains with `torch.autograd.grad`
(`sample_diffusion.py:34-42`) -- is an explicit backward schedule (transposed convolutions on the MFMA kernel,
GroupNorm / GELU / avg-pool / attention-pool backward kernels).  There is no CPU path: CPU tensors raise.

The classes below are parameter containers (reference state-dict names and shapes); there is no stock-torch
evaluation of them in this package -- the CPU restatement lives in oracle/ref_cpu.py (test infrastructure).
"""

from __future__ import annotations

from typing import Any, Dict

import torch
import torch.nn as nn

from .base import Savable
from . import _native
from .unet import CHANNEL_MULT, ResBlock, _NativeModule, _groups, _scaled, _seq, check_base_channels, check_topology


class AttentionPool1d(nn.Module):
    def __init__(self, channels: int, head_channels: int = 64, out_channels: int = None):
        super().__init__()
        assert channels % head_channels == 0
        self.qkv_proj = nn.Conv1d(channels, 3 * channels, 1)
        self.c_proj = nn.Conv1d(channels, out_channels or channels, 1)
        self.num_heads = channels // head_channels


class ClassifierStem(nn.Module):
    def __init__(self, base_channels: int = 32, channel_mult=CHANNEL_MULT, output_mult: int = 16, depth_mult: int = 2):
        super().__init__()
        self.base_channels, self.channel_mult = base_channels, tuple(channel_mult)
        self.output_mult, self.depth_mult = output_mult, depth_mult
        self.out_channels = base_channels * output_mult
        E = self.embed_dim = 4 * base_channels
        self.time_embed = nn.Module()
        self.time_embed.proj = nn.Linear(E, E)
        self.time_embed_extra = _seq(None, nn.Linear(E, E))
        self.in_conv = nn.Conv1d(1, base_channels, 3, padding=1)
        blocks, cur = [], base_channels
        for mult in self.channel_mult:
            for _ in range(depth_mult):
                blocks.append(ResBlock(cur, E, mult * base_channels))
                cur = mult * base_channels
            blocks.append(ResBlock(cur, E, cur, scale_factor=0.5))
        self.blocks = nn.ModuleList(blocks)
        self.out = _seq(_seq(nn.GroupNorm(_groups(cur), cur), None),
                        AttentionPool1d(cur, head_channels=min(cur, 64), out_channels=self.out_channels))


    def load_from_predictor(self, pred) -> int:
        """Copy the stem's shared parameters from a UNetPredictor (classifier.py:123-131: in_conv, both time-embedding layers and the
        blocks that line up with the predictor's down path); returns the number of values copied.  Host-side only."""
        dsts = [self.in_conv, self.time_embed, self.time_embed_extra, *self.blocks]
        srcs = [pred.in_conv, pred.time_embed, pred.time_embed_extra, *pred.down_blocks]
        total = 0
        for dst, src in zip(dsts, srcs):
            dst.load_state_dict(src.state_dict())
            total += sum(int(v.numel()) for v in src.state_dict().values())
        return total


class Classifier(_NativeModule, Savable):
    def __init__(self, num_labels: int, **kwargs):
        super().__init__()
        self.num_labels = num_labels
        self.stem = ClassifierStem(**kwargs)
        check_base_channels(self.stem.base_channels)
        check_topology(self.stem.base_channels, self.stem.channel_mult, self.stem.depth_mult, ())  # (classifier.py:52-58: any of these)
        if int(self.stem.output_mult) != self.stem.output_mult or not 1 <= self.stem.output_mult * self.stem.base_channels <= 4096:
            raise ValueError(f"output_mult={self.stem.output_mult}: the feature width must be in 1..4096")
        self.out = _seq(None, _scaled(nn.Linear(self.stem.out_channels, num_labels), 0.0))

    def save_kwargs(self) -> Dict[str, Any]:
        s = self.stem
        return dict(num_labels=self.num_labels, base_channels=s.base_channels, channel_mult=s.channel_mult,
                    output_mult=s.output_mult, depth_mult=s.depth_mult)

    @property
    def downsample_rate(self) -> int:
        return 2 ** len(self.stem.channel_mult)

    def _cfg(self) -> _native.Cfg:
        cfg = _native.Cfg()
        cfg.kind = _native.KIND_CLASSIFIER
        cfg.base_channels = self.stem.base_channels
        cfg.in_channels = 1
        cfg.num_labels = self.num_labels
        if (tuple(self.stem.channel_mult), self.stem.depth_mult) != (CHANNEL_MULT, 2):
            cfg.set_topology(self.stem.channel_mult, self.stem.depth_mult, ())
        cfg.reserved[1] = 0 if self.stem.output_mult == 16 else int(self.stem.output_mult)
        return cfg

    def _prepare(self, x: torch.Tensor, ts: torch.Tensor):
        _native.require_cuda(x, ts)
        if x.dim() != 3 or x.shape[1] != 1:
            raise ValueError(f"expected x of shape [N, 1, T], got {tuple(x.shape)}")
        B, _, T = x.shape
        if T % self.downsample_rate:
            raise ValueError(f"T={T} is not a multiple of the classifier downsample rate {self.downsample_rate}")
        x = x.detach().to(torch.float32).contiguous()
        ts = ts.detach().to(device=x.device, dtype=torch.float32).contiguous()
        if ts.shape != (B,):
            raise ValueError(f"expected ts of shape [{B}], got {tuple(ts.shape)}")
        return x, ts, B, T

    def forward(self, x: torch.Tensor, ts: torch.Tensor, use_checkpoint: bool = False, **kwargs) -> torch.Tensor:
        """logits [N, num_labels] (classifier.py:31-36) through `vqvs_classifier_forward`."""
        x, ts, B, T = self._prepare(x, ts)
        h = self.handle(x.device, B, T)
        logits = torch.empty(B, self.num_labels, device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _native.check(_native.lib().vqvs_classifier_forward(h.ptr, x.data_ptr(), ts.data_ptr(), logits.data_ptr(), B, T,
                                                               _native._stream_ptr()))
        return logits

    def log_prob_grad(self, x: torch.Tensor, ts: torch.Tensor, labels: torch.Tensor, scale: float = 1.0, return_logits: bool = False):
        """scale * d/dx log_softmax(self(x, ts))[labels] through `vqvs_classifier_guidance` (forward + explicit backward)."""
        x, ts, B, T = self._prepare(x, ts)
        _native.require_cuda(labels)
        labels = labels.detach().to(device=x.device, dtype=torch.int64).contiguous()
        if labels.shape != (B,):
            raise ValueError(f"expected labels of shape [{B}], got {tuple(labels.shape)}")
        h = self.handle(x.device, B, T)
        grad = torch.empty_like(x)
        logits = torch.empty(B, self.num_labels, device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _native.check(_native.lib().vqvs_classifier_guidance(h.ptr, x.data_ptr(), ts.data_ptr(), labels.data_ptr(), float(scale),
                                                                grad.data_ptr(), logits.data_ptr(), B, T, _native._stream_ptr()))
        return (grad, logits) if return_logits else grad

    def guidance_fn(self, labels: torch.Tensor, scale: float = 1.0):
        """cond_fn(x, ts) = scale * d/dx log softmax(classifier(x, ts))[labels]   (sample_diffusion.py:34-42)"""

        def cond_fn(x, ts):
            return self.log_prob_grad(x, ts, labels.to(x.device), scale)

        cond_fn.native_modules = (self,)  # (Diffusion.ddpm_sample promotes few-step guided runs of 2-byte models to the fp32 mode)
        return cond_fn
