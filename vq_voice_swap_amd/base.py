"""
Checkpoint container shared by the model facades.

Keeps the reference's on-disk format (reference vq_voice_swap/models/base.py:63-127):
a dict {"kwargs": constructor kwargs, "state_dict": tensors} written with torch.save,
so reference checkpoints load unchanged and ours load in the reference.
"""

from __future__ import annotations

import os
import tempfile
from typing import Any, Dict

import torch
import torch.nn as nn


class Savable(nn.Module):
    def save_kwargs(self) -> Dict[str, Any]:
        raise NotImplementedError

    def save_dict(self) -> Dict[str, Any]:
        return {"kwargs": self.save_kwargs(), "state_dict": self.state_dict()}

    @classmethod
    def load_dict(cls, state: Dict[str, Any]):
        obj = cls(**state["kwargs"])
        obj.load_state_dict(state["state_dict"])
        return obj

    def save(self, path: str) -> None:
        atomic_save(self.save_dict(), path)

    @classmethod
    def load(cls, path: str):
        return cls.load_dict(torch.load(path, map_location="cpu"))

    def load_from_pretrained(self, model: nn.Module) -> int:
        """Copy every parameter that exists in both modules; returns the element count."""
        theirs = dict(model.named_parameters())
        copied = 0
        with torch.no_grad():
            for name, mine in self.named_parameters():
                src = theirs.get(name)
                if src is None:
                    continue
                if src.shape != mine.shape:
                    raise RuntimeError(
                        f"Parameter {name} has shape {tuple(mine.shape)} in destination but {tuple(src.shape)} in source."
                    )
                mine.copy_(src)
                copied += mine.numel()
        return copied


def atomic_save(state: Any, path: str) -> None:
    """Write to a temporary file in the destination directory, then rename over `path`."""
    d = os.path.dirname(os.path.abspath(path))
    fd, tmp = tempfile.mkstemp(dir=d, suffix=".tmp")
    os.close(fd)
    try:
        torch.save(state, tmp)
        os.replace(tmp, path)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
