"""
ConvMFCCEncoder facade (reference vq_voice_swap/models/conv_encoder.py:14-133): the encoder of the reference's released
speaker-conversion checkpoint (README.md:21).  mu-law expansion, MFCC + deltas and the convolution stack all run in
libvqvs_hip.so (`vqvs_mfcc_encoder_forward`); this module is the parameter container with the reference's state-dict
names, *including* the three persistent buffers of `torchaudio.transforms.MFCC` (window, mel filter bank, DCT matrix),
so a reference checkpoint loads with strict=True and its own constants are the ones the kernels use.

torchaudio is not needed (and not available here): fresh modules fill those buffers with the restated construction
formulas of torchaudio.functional.melscale_fbanks / create_dct and torch.hann_window.
"""

from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import _native
from .unet import _NativeModule, _seq


class _ResConv(nn.Module):
    """Parameter container of conv_encoder.py:113-120 (x + gelu(conv(x)))."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        self.conv = nn.Conv1d(*args, **kwargs)


class _MFCCBuffers(nn.Module):
    """Buffers under the names torchaudio.transforms.MFCC registers: dct_mat, MelSpectrogram.spectrogram.window,
    MelSpectrogram.mel_scale.fb."""

    def __init__(self, sample_rate: int, n_fft: int, n_mels: int, n_mfcc: int = 13):
        super().__init__()
        n_freqs = n_fft // 2 + 1
        all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
        m_max = 2595.0 * math.log10(1.0 + float(sample_rate // 2) / 700.0)
        m_pts = torch.linspace(0.0, m_max, n_mels + 2)
        f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
        f_diff = f_pts[1:] - f_pts[:-1]
        slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
        fb = torch.max(torch.zeros(1), torch.min((-1.0 * slopes[:, :-2]) / f_diff[:-1], slopes[:, 2:] / f_diff[1:]))
        n = torch.arange(float(n_mels))
        k = torch.arange(float(n_mfcc)).unsqueeze(1)
        dct = torch.cos(math.pi / float(n_mels) * (n + 0.5) * k)
        dct[0] *= 1.0 / math.sqrt(2.0)
        dct *= math.sqrt(2.0 / float(n_mels))
        self.register_buffer("dct_mat", dct.t().contiguous())
        self.MelSpectrogram = nn.Module()
        self.MelSpectrogram.spectrogram = nn.Module()
        self.MelSpectrogram.spectrogram.register_buffer("window", torch.hann_window(n_fft))
        self.MelSpectrogram.mel_scale = nn.Module()
        self.MelSpectrogram.mel_scale.register_buffer("fb", fb)


class ConvMFCCEncoder(_NativeModule):
    def __init__(self, base_channels: int, out_channels: int = 64, input_ulaw: bool = True, input_rate: int = 16000,
                 mfcc_rate: int = 100, version: int = 1):
        super().__init__()
        if input_rate != 16000 or mfcc_rate != 100:
            raise ValueError("the gfx950 library implements the reference's 16 kHz / 100 frames-per-second front end only")
        if version not in (1, 2):
            raise ValueError(f"unknown ConvMFCCEncoder version {version}")
        self.base_channels = base_channels
        self.out_channels = out_channels
        self.input_ulaw = input_ulaw
        self.input_rate = input_rate
        self.mfcc_rate = mfcc_rate
        self.mid_channels = mid = base_channels * 12
        self.version = version
        self.precision = "fp32"  # feeds the VQ layer: code indices have to be bit-exact
        n_fft = round(400 * input_rate / 16000) if version == 2 else (input_rate // mfcc_rate) * 2
        self.mfcc = _MFCCBuffers(input_rate, n_fft, 40 if version == 1 else 80)
        self.blocks = nn.ModuleList([
            _seq(nn.Conv1d(13 * 3, mid, 3, padding=1), None),
            _ResConv(mid, mid, 3, padding=1),
            _seq(nn.Conv1d(mid, mid, 4, stride=2, padding=1), None),
            *[_ResConv(mid, mid, 3, padding=1) for _ in range(2)],
            *[_ResConv(mid, mid, 1) for _ in range(4)],
            nn.Conv1d(mid, out_channels, 1),
        ])
        with torch.no_grad():  # zero output so that by default downstream models are unaffected (conv_encoder.py:85-88)
            for p in self.blocks[-1].parameters():
                p.zero_()

    def set_precision(self, precision: str):
        if precision not in ("fp32", "f32", "float32"):
            raise ValueError("ConvMFCCEncoder runs in the fp32 mode only (its output is vector-quantised)")
        return self

    def _cfg(self) -> _native.Cfg:
        cfg = _native.Cfg()
        cfg.kind = _native.KIND_MFCC_ENCODER
        cfg.base_channels = self.base_channels
        cfg.in_channels = 1
        cfg.out_channels = self.out_channels
        cfg.reserved[1] = self.version
        cfg.reserved[2] = 1 if self.input_ulaw else 0
        return cfg

    def out_length(self, T: int) -> int:
        frames = T // (self.input_rate // self.mfcc_rate) + 1
        return (frames - 2) // 2 + 1

    def forward(self, x: torch.Tensor, use_checkpoint: bool = False) -> torch.Tensor:
        _native.require_cuda(x)
        assert x.dim() == 3 and x.shape[1] == 1, "input must only have one channel"
        B, _, T = x.shape
        x = x.detach().to(torch.float32).contiguous()
        h = self.handle(x.device, B, T)
        z = torch.empty(B, self.out_channels, self.out_length(T), device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _native.check(_native.lib().vqvs_mfcc_encoder_forward(h.ptr, x.data_ptr(), z.data_ptr(), B, T, _native._stream_ptr()))
        return z

    def forward_from_mfcc(self, mfcc: torch.Tensor, T: int) -> torch.Tensor:
        """Testing entry (`vqvs_mfcc_encoder_forward_logmel`): run everything BEHIND the MFCC transform -- `deltas` twice, the
        concatenation, the convolution stack (reference conv_encoder.py:97-109) -- on an injected [B, 13, T // 160 + 1] coefficient
        tensor.  The kernels start from log-mel rows, so the injection is logmel = dct_mat . mfcc (orthonormal columns: the
        DCT gives the coefficients back).  Version-1 front end only."""
        _native.require_cuda(mfcc)
        B, n, frames = mfcc.shape
        assert n == 13 and frames == T // (self.input_rate // self.mfcc_rate) + 1 and self.version == 1
        dct = self.mfcc.dct_mat.to(mfcc.device, torch.float64)  # [n_mels, 13]
        logmel = torch.einsum("mk,bkf->bfm", dct, mfcc.to(torch.float64)).to(torch.float32).contiguous()  # [B, frames, n_mels]
        h = self.handle(mfcc.device, B, T)
        z = torch.empty(B, self.out_channels, self.out_length(T), device=mfcc.device, dtype=torch.float32)
        with torch.cuda.device(mfcc.device):
            _native.check(_native.lib().vqvs_mfcc_encoder_forward_logmel(h.ptr, logmel.data_ptr(), z.data_ptr(), B, T, _native._stream_ptr()))
        return z

    @property
    def downsample_rate(self) -> int:
        return self.input_rate // (self.mfcc_rate // 2)
