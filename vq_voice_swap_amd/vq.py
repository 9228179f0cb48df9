"""
Vector-quantisation layer, inference half (reference vq_voice_swap/vq.py:74-143, 199-243):
nearest-codeword search and embedding gather run as HIP kernels (`vqvs_vq_argmin`,
`vqvs_vq_embed`).  The training half (losses, dead-code revival) is out of scope.
"""

from __future__ import annotations

from typing import Dict

import torch
import torch.nn as nn

from . import _native


class VQ(nn.Module):
    def __init__(self, num_channels: int, num_codes: int, dead_rate: int = 100):
        super().__init__()
        self.num_channels = num_channels
        self.num_codes = num_codes
        self.dead_rate = dead_rate
        self.dictionary = nn.Parameter(torch.randn(num_codes, num_channels))
        self.register_buffer("usage_count", dead_rate * torch.ones(num_codes).long())

    def embed(self, idxs: torch.Tensor) -> torch.Tensor:
        """int [N, ...] -> float [N, C, ...] (vq.py:98-110)."""
        _native.require_cuda(idxs)
        n = idxs.shape[0]
        flat = idxs.detach().reshape(n, -1).to(torch.int64).contiguous()
        _native.check_index_range(flat, self.num_codes, "VQ codes")
        d = self.dictionary.detach().to(device=flat.device, dtype=torch.float32).contiguous()
        out = torch.empty(n, self.num_channels, flat.shape[1], device=flat.device, dtype=torch.float32)
        with torch.cuda.device(flat.device):
            _native.check(_native.lib().vqvs_vq_embed(flat.data_ptr(), d.data_ptr(), out.data_ptr(), n, self.num_channels,
                                                      flat.shape[1], self.num_codes, _native._stream_ptr()))
        return out.reshape(n, self.num_channels, *idxs.shape[1:])

    def encode(self, inputs: torch.Tensor) -> torch.Tensor:
        """float [N, C, ...] -> int64 [N, ...] nearest code, first index on ties (vq.py:127-131)."""
        _native.require_cuda(inputs)
        n, c = inputs.shape[:2]
        if c != self.num_channels:
            raise ValueError(f"expected {self.num_channels} channels, got {c}")
        z = inputs.detach().to(torch.float32).reshape(n, c, -1).contiguous()
        d = self.dictionary.detach().to(device=z.device, dtype=torch.float32).contiguous()
        idx = torch.empty(n, z.shape[2], device=z.device, dtype=torch.int64)
        with torch.cuda.device(z.device):
            _native.check(_native.lib().vqvs_vq_argmin(z.data_ptr(), d.data_ptr(), idx.data_ptr(), n, c, z.shape[2], self.num_codes,
                                                       _native._stream_ptr()))
        return idx.reshape(n, *inputs.shape[2:])

    def forward(self, inputs: torch.Tensor) -> Dict[str, torch.Tensor]:
        if self.training:
            raise RuntimeError("VQ training (usage tracking / revival, vq.py:145-196) is outside the accelerated path; call .eval()")
        idxs = self.encode(inputs)
        embedded = self.embed(idxs)
        return {"embedded": embedded, "passthrough": embedded, "idxs": idxs}
