"""
VQVAE facade: UNet encoder + VQ + conditional diffusion decoder
(reference vq_voice_swap/vq_vae.py:10-240).  encode / decode keep the reference's
signatures; every stage runs in libvqvs_hip.so.
"""

from __future__ import annotations

from typing import Any, Dict, Optional

import torch

from .diffusion import randn_clips
from .diffusion_model import DiffusionModel
from .conv_encoder import ConvMFCCEncoder
from .unet import UNetEncoder
from .vq import VQ


def make_encoder(enc_name: str, base_channels: int = 32, cond_mult: int = 16):
    """reference models/make.py:41-84: the UNet encoder and the three ConvMFCCEncoder variants are built."""
    if enc_name == "unet":
        return UNetEncoder(base_channels=base_channels, out_channels=base_channels * cond_mult)
    if enc_name == "conv-mfcc-ulaw":
        return ConvMFCCEncoder(base_channels=base_channels, out_channels=base_channels * cond_mult)
    if enc_name == "conv-mfcc-ulaw-v2":
        return ConvMFCCEncoder(base_channels=base_channels, out_channels=base_channels * cond_mult, version=2)
    if enc_name == "conv-mfcc-linear":
        return ConvMFCCEncoder(base_channels=base_channels, out_channels=base_channels * cond_mult, input_ulaw=False)
    raise ValueError(f"encoder {enc_name!r} is outside the accelerated hot path (SURVEY.md section 2a): "
                     "'unet' and the 'conv-mfcc-*' encoders are built")


class VQVAE(DiffusionModel):
    def __init__(self, base_channels: int, enc_name: str = "unet", cond_mult: int = 16, dictionary_size: int = 512, **kwargs):
        encoder = make_encoder(enc_name, base_channels=base_channels, cond_mult=cond_mult)
        kwargs["cond_channels"] = base_channels * cond_mult
        super().__init__(base_channels=base_channels, **kwargs)
        self.enc_name = enc_name
        self.cond_mult = cond_mult
        self.dictionary_size = dictionary_size
        self.encoder = encoder
        self.vq = VQ(self.cond_channels, dictionary_size)

    def set_precision(self, precision: str, encoder_precision: str = "fp32"):
        """Precision of the diffusion decoder; the encoder stays in the fp32 mode unless asked otherwise, because VQ code
        indices have to be bit-exact (a 2-byte encoder flips near-tie codes: 23/500 in bf16) and it runs once per clip."""
        self.predictor.set_precision(precision)
        self.encoder.set_precision(encoder_precision)  # (ConvMFCCEncoder accepts fp32 only)
        return self

    def encode(self, inputs: torch.Tensor) -> torch.Tensor:
        """[N,1,T] waveform -> [N,T/256] int64 codes (vq_vae.py:82-90)."""
        with torch.no_grad():
            return self.vq.encode(self.encoder(inputs))

    def decode(self, codes: torch.Tensor, labels: Optional[torch.Tensor] = None, steps: int = 100, progress: bool = False,
               constrain: bool = False, enc_pred=None, enc_pred_scale: float = 1.0, x_T: Optional[torch.Tensor] = None,
               **kwargs) -> torch.Tensor:
        """codes [N,T1] int or [N,C,T1] float -> [N,1,T1*256] waveform (vq_vae.py:92-145)."""
        if codes.dim() == 2:
            cond_seq = self.vq.embed(codes)
        elif codes.dim() == 3:
            cond_seq = codes
        else:
            raise ValueError(f"unsupported codes shape: {codes.shape}")
        cond_fn = None
        if enc_pred is not None:  # vq_vae.py:123-130: guidance towards the codes, gradient from the native backward schedule
            targets = self.vq.encode(cond_seq)
            cond_fn = enc_pred.guidance_fn(targets, enc_pred_scale)

        T = codes.shape[-1] * self.encoder.downsample_rate
        seed = kwargs.pop("seed", None)
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        if x_T is None:
            x_T = randn_clips(codes.shape[0], T, codes.device, seed, kwargs.get("clip_offset", 0))
        out = self.diffusion.ddpm_sample(
            x_T, lambda xs, ts, **kw: self.predictor(xs, ts, cond=cond_seq, labels=labels, **kw),
            steps=steps, progress=progress, constrain=constrain, cond_fn=cond_fn, seed=seed, **kwargs)
        self.predictor.check_status()  # range guard of the decoder's mode (once per sample)
        return out

    def decode_uncond_guidance(self, codes: torch.Tensor, labels: Optional[torch.Tensor] = None, steps: int = 100,
                               progress: bool = False, constrain: bool = False, label_scale: float = 0.0, vq_scale: float = 0.0,
                               x_T: Optional[torch.Tensor] = None, **kwargs) -> torch.Tensor:
        """Decode with classifier-free-style guidance towards the VQ codes and/or the label (reference
        vq_vae.py:147-220): the predictor runs on a 1x-3x batch [conditional | codes dropped | label dropped] and the
        prediction is base + scale * (base - dropped).  Labels are NOT offset by the caller: label 0 is the
        unconditional label of such a model, so `labels + 1` is used as in the reference."""
        if codes.dim() == 2:
            cond_seq = self.vq.embed(codes)
        elif codes.dim() == 3:
            cond_seq = codes
        else:
            raise ValueError(f"unsupported codes shape: {codes.shape}")
        n = cond_seq.shape[0]
        T = codes.shape[-1] * self.encoder.downsample_rate
        seed = kwargs.pop("seed", None)
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        if x_T is None:
            x_T = randn_clips(n, T, codes.device, seed, kwargs.get("clip_offset", 0))

        use_vq = bool(vq_scale)
        use_label = labels is not None and bool(label_scale)
        reps = 1 + int(use_vq) + int(use_label)
        cond_batch = [cond_seq]
        label_batch = [labels + 1] if labels is not None else None
        if use_vq:
            cond_batch.append(torch.zeros_like(cond_seq))
            if label_batch is not None:
                label_batch.append(labels + 1)
        if use_label:
            cond_batch.append(cond_seq)
            label_batch.append(torch.zeros_like(labels))
        cond_batch = torch.cat(cond_batch, dim=0)
        label_batch = torch.cat(label_batch, dim=0) if label_batch is not None else None

        def pred_fn(xs, ts):
            outs = self.predictor(torch.cat([xs] * reps, dim=0), torch.cat([ts] * reps, dim=0), cond=cond_batch, labels=label_batch)
            base = outs[:n]
            pred, k = base, 1
            for flag, scale in ((use_vq, vq_scale), (use_label, label_scale)):
                if flag:
                    pred = pred + scale * (base - outs[k * n:(k + 1) * n])
                    k += 1
            return pred

        # The extrapolation base + s_vq (base - a) + s_label (base - b) multiplies the predictor's rounding error by up to
        # 1 + 2 (s_vq + s_label): a 2-byte decoder mode does not hold the 1e-3 waveform contract here (fixtures F11 / F11b), so the
        # guided predictor runs in the fp32 mode for this call, whatever mode the decoder is set to.
        prev = self.predictor.precision
        promote = (use_vq or use_label) and prev != "fp32"
        if promote:
            import warnings

            warnings.warn(f"decode_uncond_guidance: the predictor runs in the fp32 mode for this call (decoder mode {prev!r} does not "
                          "meet the 1e-3 waveform contract under guidance extrapolation)", stacklevel=2)
        # (precision_override keeps the decoder's own handle and arena; the fp32 handle is cached beside it for the next call)
        with self.predictor.precision_override("fp32" if promote else prev):
            out = self.diffusion.ddpm_sample(x_T, pred_fn, steps=steps, progress=progress, constrain=constrain, seed=seed, **kwargs)
            self.predictor.check_status()
        return out

    @property
    def downsample_rate(self) -> int:
        import math

        a, b = self.predictor.downsample_rate, self.encoder.downsample_rate
        return a * b // math.gcd(a, b)

    def save_kwargs(self) -> Dict[str, Any]:
        res = super().save_kwargs()
        res.update(dict(enc_name=self.enc_name, cond_mult=self.cond_mult, dictionary_size=self.dictionary_size))
        return res
