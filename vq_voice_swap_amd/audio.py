"""
16 kHz mono PCM I/O without ffmpeg, with the reference's ChunkReader / ChunkWriter call surface
(reference vq_voice_swap/dataset.py:167-303) and its mu-law codec (dataset.py:342-347).

The reference pipes s16le through one `ffmpeg` subprocess per clip (sample_diffusion.py:98-105 spawns one
per sample); at tens of clips per second that pipe is the bottleneck, and ffmpeg is not in this image.
These classes write RIFF/WAVE s16 mono directly and read RIFF/WAVE of any PCM / float encoding, channel count and sample rate
(down-mixed and resampled to the requested rate, as the reference's ffmpeg pipe does); other containers use ffmpeg when it exists.
"""

from __future__ import annotations

import wave
from typing import Optional

import struct

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


def _resample(x: np.ndarray, rate_in: int, rate_out: int) -> np.ndarray:
    """Band-limited rational resampling (polyphase FIR, Kaiser window): what the reference gets from `ffmpeg -ar` (dataset.py:186-195).
    Not bit-identical to ffmpeg's swresample -- no two resamplers are -- but the same operation; the reference's own s16 quantisation
    follows it."""
    if rate_in == rate_out or x.size == 0:
        return x
    from math import gcd

    g = gcd(int(rate_in), int(rate_out))
    up, down = int(rate_out) // g, int(rate_in) // g
    try:
        from scipy.signal import resample_poly

        return resample_poly(x.astype(np.float64), up, down).astype(np.float32)
    except ImportError:  # (plain numpy: windowed-sinc low-pass at the lower Nyquist rate, evaluated at the output instants)
        cutoff = min(1.0, rate_out / rate_in)
        half = int(np.ceil(16 / cutoff))
        n_out = int(np.ceil(x.size * rate_out / rate_in))
        t = np.arange(n_out) * (rate_in / rate_out)
        out = np.zeros(n_out, dtype=np.float64)
        base = np.floor(t).astype(np.int64)
        for k in range(-half, half + 1):
            idx = base + k
            d = t - idx
            w = np.where(np.abs(d) <= half, 0.5 + 0.5 * np.cos(np.pi * d / half), 0.0)
            tap = cutoff * np.sinc(cutoff * d) * w
            ok = (idx >= 0) & (idx < x.size)
            out[ok] += tap[ok] * x[idx[ok]]
        return out.astype(np.float32)


def _read_wav(path: str):
    """(float32 mono samples in [-1, 1), sample rate) of a RIFF/WAVE file: PCM of 8 / 16 / 24 / 32 bits or IEEE float32 / float64, any
    channel count (down-mixed by the mean, as ffmpeg's `-ac 1` does for stereo)."""
    import struct

    with open(path, "rb") as f:
        data = f.read()
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError("not a RIFF/WAVE file")
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            if len(body) < 16:
                raise ValueError("truncated WAVE fmt chunk")
            tag, ch, rate, _, _, bits = struct.unpack("<HHIIHH", body[:16])
            if tag == 0xFFFE and len(body) >= 26:  # WAVE_FORMAT_EXTENSIBLE: the real tag is the sub-format GUID's first two bytes
                tag = struct.unpack("<H", body[24:26])[0]
            fmt = (tag, ch, rate, bits)
        elif cid == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None:
        raise ValueError("WAVE file without fmt / data chunk")
    tag, ch, rate, bits = fmt
    if tag == 1 and bits == 8:
        x = (np.frombuffer(pcm, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    elif tag == 1 and bits == 16:
        x = np.frombuffer(pcm[: len(pcm) // 2 * 2], dtype="<i2").astype(np.float32) / 2 ** 15
    elif tag == 1 and bits == 24:
        b = np.frombuffer(pcm[: len(pcm) // 3 * 3], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        x = (v - ((v & 0x800000) << 1)).astype(np.float32) / 2 ** 23
    elif tag == 1 and bits == 32:
        x = np.frombuffer(pcm[: len(pcm) // 4 * 4], dtype="<i4").astype(np.float32) / 2 ** 31
    elif tag == 3 and bits in (32, 64):
        x = np.frombuffer(pcm[: len(pcm) // (bits // 8) * (bits // 8)], dtype="<f4" if bits == 32 else "<f8").astype(np.float32)
    else:
        raise ValueError(f"unsupported WAVE encoding (format tag {tag}, {bits} bits)")
    if ch > 1:
        x = x[: x.size // ch * ch].reshape(-1, ch).mean(axis=1)
    return x, rate


class ChunkReader:
    """read(n) -> float32 array in [-1,1] (s16 / 2^15, as the reference) or None at end of file.

    The reference decodes, down-mixes and resamples ANY input through an `ffmpeg -f s16le -ar <rate> -ac 1` pipe (dataset.py:177-203).
    Here RIFF/WAVE files (PCM 8 / 16 / 24 / 32 bit, float32 / float64, any channel count, any sample rate) are decoded, down-mixed
    and band-limit resampled natively, then quantised to s16 exactly where the reference's pipe does; any OTHER container goes
    through the reference's own ffmpeg pipe when an `ffmpeg` binary is on PATH, and is refused with that reason otherwise."""

    def __init__(self, path: str, sample_rate: int, encoding: str = "linear"):
        self.path, self.sample_rate, self.encoding = path, sample_rate, encoding
        self._pos = 0
        try:
            x, rate = _read_wav(path)
            x = _resample(x, rate, sample_rate)
            # the reference sees s16le samples at this point (ffmpeg's output format): the same grid here
            self._s16 = np.clip(np.rint(x * 2 ** 15), -2 ** 15, 2 ** 15 - 1).astype("<i2")
        except (ValueError, struct.error) as e:  # (a malformed header must reach the ffmpeg fallback / the friendly error too)
            import shutil
            import subprocess

            if shutil.which("ffmpeg") is None:
                raise ValueError(f"{path}: {e}; other containers need an `ffmpeg` binary on PATH (as in the reference, dataset.py:185-195)") from e
            raw = subprocess.run(["ffmpeg", "-i", path, "-f", "s16le", "-ar", str(sample_rate), "-ac", "1", "pipe:1"], stdin=subprocess.DEVNULL,
                                 stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
            self._s16 = np.frombuffer(raw[: len(raw) // 2 * 2], dtype="<i2")

    def read(self, chunk_size: int) -> Optional[np.ndarray]:
        if self._pos >= self._s16.size:
            return None
        x = self._s16[self._pos:self._pos + chunk_size].astype("float32")
        self._pos += chunk_size
        return encode_from_linear(x / (2 ** 15), self.encoding)

    def close(self) -> None:
        self._s16 = self._s16[:0]


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
