"""
Deterministic, name-keyed parameter initialiser.

No pretrained checkpoints are reachable offline, and the reference's default
init zeroes every ResBlock's second conv (reference vq_voice_swap/models/unet.py:286-294,
352-356), which would hide the hot path.  This initialiser gives every float
tensor a reproducible, non-degenerate value that depends only on its state-dict
name and shape, so the reference (in the build container), the CPU oracle and
the HIP path can all be loaded with identical weights anywhere.

Used by: oracle/gen_golden.py (applied to the reference's modules), tests/,
bench.py (synthetic weights).
"""

from __future__ import annotations

import math
import zlib
from typing import Dict, Iterable, Tuple

import torch


def det_tensor(name: str, shape, scale: float = 0.5) -> torch.Tensor:
    g = torch.Generator().manual_seed(zlib.crc32(name.encode("utf-8")))
    shape = tuple(shape)
    if len(shape) >= 2:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        return torch.randn(shape, generator=g) * (scale / math.sqrt(fan_in))
    if name.endswith("weight"):
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    return 0.1 * torch.randn(shape, generator=g)


def det_init_(named: Iterable[Tuple[str, torch.Tensor]], scale: float = 0.5) -> None:
    """In-place: fill every floating-point tensor from its name."""
    with torch.no_grad():
        for name, t in named:
            if not torch.is_floating_point(t):
                continue
            t.copy_(det_tensor(name, t.shape, scale).to(t.dtype))


def det_state_dict(shapes: Dict[str, Tuple[int, ...]], scale: float = 0.5) -> Dict[str, torch.Tensor]:
    return {k: det_tensor(k, s, scale) for k, s in shapes.items()}
