"""
16 kHz mono PCM I/O without ffmpeg, with the reference's ChunkReader / ChunkWriter call surface
(reference vq_voice_swap/dataset.py:167-303) and its mu-law codec (dataset.py:342-347).

The reference pipes s16le through one `ffmpeg` subprocess per clip (sample_diffusion.py:98-105 spawns one
per sample); at tens of clips per second that pipe is the bottleneck, and ffmpeg is not in this image.
These classes read/write RIFF/WAVE s16 mono directly.
"""

from __future__ import annotations

import wave
from typing import Optional

import numpy as np


def encode_u_law(x: np.ndarray, mu: float = 255.0) -> np.ndarray:
    return np.sign(x) * (np.log(1 + mu * np.abs(x)) / np.log(1 + mu))


def decode_u_law(x: np.ndarray, mu: float = 255.0) -> np.ndarray:
    return np.sign(x) * (1 / mu) * ((1 + mu) ** np.abs(x) - 1)


def encode_from_linear(x: np.ndarray, encoding: str) -> np.ndarray:
    if encoding == "linear":
        return x
    if encoding == "ulaw":
        return encode_u_law(x)
    raise ValueError(f"unknown audio encoding: {encoding}")


def decode_to_linear(x: np.ndarray, encoding: str) -> np.ndarray:
    if encoding == "linear":
        return x
    if encoding == "ulaw":
        return decode_u_law(x)
    raise ValueError(f"unknown audio encoding: {encoding}")


class ChunkWriter:
    """write(chunk of floats in [-1,1]) ... close(); s16 quantisation as the reference: clip, * (2^15 - 1), truncate."""

    def __init__(self, path: str, sample_rate: int, encoding: str = "linear"):
        self.path, self.sample_rate, self.encoding = path, sample_rate, encoding
        self._w = wave.open(path, "wb")
        self._w.setnchannels(1)
        self._w.setsampwidth(2)
        self._w.setframerate(sample_rate)

    def write(self, chunk: np.ndarray) -> None:
        chunk = np.clip(np.asarray(chunk, dtype=np.float32), -1, 1)
        chunk = decode_to_linear(chunk, self.encoding)
        self._w.writeframes((chunk * (2 ** 15 - 1)).astype("<i2").tobytes())

    def close(self) -> None:
        self._w.close()


class ChunkReader:
    """read(n) -> float32 array in [-1,1] (s16 / 2^15, as the reference) or None at end of file."""

    def __init__(self, path: str, sample_rate: int, encoding: str = "linear"):
        self.path, self.sample_rate, self.encoding = path, sample_rate, encoding
        self._r = wave.open(path, "rb")
        if self._r.getsampwidth() != 2:
            raise ValueError(f"{path}: only 16-bit PCM WAV is supported (no ffmpeg in this build)")
        if self._r.getframerate() != sample_rate:
            raise ValueError(f"{path}: sample rate {self._r.getframerate()} != requested {sample_rate}; resample first (no ffmpeg in this build)")
        self._channels = self._r.getnchannels()
        self._done = False

    def read(self, chunk_size: int) -> Optional[np.ndarray]:
        if self._done:
            return None
        buf = self._r.readframes(chunk_size)
        n = len(buf) // (2 * self._channels)
        if n < chunk_size:
            self._done = True
        if n == 0:
            return None
        x = np.frombuffer(buf, dtype="<i2").astype("float32")
        if self._channels > 1:
            x = x.reshape(-1, self._channels).mean(axis=1)
        return encode_from_linear(x / (2 ** 15), self.encoding)

    def close(self) -> None:
        self._r.close()


def parse_time_schedule(text: str):
    """The reference takes `--schedule "lambda t: t**2"` and eval()s it (sample_diffusion.py:22,139).  Only the
    documented power forms are accepted here, parsed without eval: "lambda t: t", "lambda t: t**2", "t**1.5", ..."""
    s = text.replace(" ", "")
    if s.startswith("lambdat:"):
        s = s[len("lambdat:"):]
    if s == "t":
        return None
    if s.startswith("t**"):
        p = float(s[3:])
        return lambda t, _p=p: t ** _p
    raise ValueError(f"unsupported sample-time schedule {text!r}; use 'lambda t: t' or 'lambda t: t**P'")
