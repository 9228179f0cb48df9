"""
UNetPredictor / UNetEncoder facades backed by the gfx950 library.

The modules below are *parameter containers*: they register parameters under exactly the
reference's state-dict names (reference vq_voice_swap/models/unet.py:16-116, 187-227,
248-305; key list in SURVEY.md 8b) so checkpoints load unchanged, but their `forward`
does not run torch ops -- it hands device pointers to `libvqvs_hip.so`
(`vqvs_unet_forward` / `vqvs_encoder_forward`), which runs the fused HIP schedule.
"""

from __future__ import annotations

import contextlib
import functools
import os
import warnings
from typing import Callable, List, Optional, Tuple

import torch
import torch.nn as nn

from . import _native

CHANNEL_MULT = (1, 1, 2, 2, 2, 4, 4, 8, 8)
MIDDLE_DILATIONS = (4, 8, 16, 32)


SUPPORTED_BASE_CHANNELS = (32, 64, 128)  # the tuned widths (the reference's two published ones and the next)


def pad_blocks(base_channels: int) -> Tuple[int, int]:
    """(q, q_p) of a width that is not a multiple of 32 (csrc/net.cpp pad_map): base = 2^k * q with q odd; the handle is built with
    channels in blocks of q_p -- the next power of two >= q, widened until 2^k * q_p is a multiple of 32 -- whose first q channels
    are the real ones and the rest stay exactly zero.  Multiples of 32: (q, q): nothing is padded."""
    k = 0
    while (base_channels >> k) & 1 == 0 and k < 5:
        k += 1
    q = base_channels >> k
    if base_channels % 32 == 0:
        return q, q
    qp = 1
    while qp < q:
        qp <<= 1
    while ((1 << k) * qp) % 32:
        qp <<= 1
    return q, qp


def physical_base(base_channels: int) -> int:
    q, qp = pad_blocks(base_channels)
    return base_channels // q * qp


def pad_state(state, table, q: int, qp: int):
    """Real-width parameters -> the physical (zero-padded) shapes of a padded handle's parameter table: every axis whose size differs
    is a channel axis -- a whole number of q-channel blocks, concatenations and FiLM's (a | b) halves included -- and its entry c goes to
    (c // q) * qp + c % q; everything else is zero."""
    out = {}
    for name, shape in table:
        t = state[name].detach().to(device="cpu", dtype=torch.float32)
        for ax, (r, p_) in enumerate(zip(t.shape, shape)):
            if r == p_:
                continue
            if r % q or r // q * qp != p_:
                raise ValueError(f"parameter {name}: axis {ax} of size {r} does not pad to {p_} in blocks of {q} -> {qp}")
            idx = torch.arange(r)
            idx = (idx // q) * qp + idx % q
            new = torch.zeros(*t.shape[:ax], p_, *t.shape[ax + 1:], dtype=t.dtype)
            new.index_copy_(ax, idx, t)
            t = new
        out[name] = t
    return out


def check_base_channels(base_channels: int, power_of_two: bool = False, pad_ok: bool = False) -> None:
    """The reference accepts any `base_channels` (models/unet.py:17-30).  UNetPredictor / UNetEncoder: any multiple of 32 up to 256
    (widths other than 32 / 64 / 128 run generic forms of a few kernels: correct, not tuned) and, since round 6, ANY width whose padded
    form (pad_blocks: 48 -> 64, 40 -> 64, 24 -> 32, 100 -> 128, ...) is at most 256 -- built at the padded width with zero channels;
    the guidance models (Classifier, EncoderPredictor): multiples of 32; ConvMFCCEncoder: powers of two.  Fail here, with the reason,
    rather than at handle creation."""
    if pad_ok and int(base_channels) == base_channels and base_channels >= 1 and base_channels % 32:
        if physical_base(base_channels) > 256:
            raise ValueError(f"base_channels={base_channels}: pads to {physical_base(base_channels)} channels, the gfx950 library builds up to 256")
        return
    if base_channels % 32 or not 32 <= base_channels <= 256:
        raise ValueError(f"base_channels={base_channels}: the gfx950 library builds multiples of 32 in 32..256 "
                         "(every convolution works on 32-channel chunks); see INTEGRATION.md")
    if power_of_two and base_channels & (base_channels - 1):
        raise ValueError(f"base_channels={base_channels}: this model's front-end kernels need a power of two ({SUPPORTED_BASE_CHANNELS})")



def check_topology(base_channels: int, channel_mult, depth_mult: int, dilations) -> None:
    """What the native schedule builds of the reference's open topology (models/unet.py:17-30, 188-196): any channel_mult /
    depth_mult / middle_dilations / out_dilations whose widths are multiples of 32 up to 1024, at most 12 levels, depth_mult 1..8,
    dilations 1..32.  Fail here, with the reason, rather than at handle creation."""
    if not 1 <= len(channel_mult) <= 12:
        raise ValueError(f"channel_mult must have 1..12 entries (got {len(channel_mult)})")
    for m in channel_mult:
        if int(m) != m or m < 1 or (m * base_channels) % 32 or m * base_channels > 1024:
            raise ValueError(f"channel_mult entry {m}: widths must be multiples of 32 in base_channels..1024")
    if int(depth_mult) != depth_mult or not 1 <= depth_mult <= 8:
        raise ValueError(f"depth_mult must be an integer in 1..8 (got {depth_mult})")
    if len(dilations) > 12 or any(int(d) != d or not 1 <= d <= 32 for d in dilations):
        raise ValueError(f"dilations must be at most 12 integers in 1..32 (got {tuple(dilations)})")


def default_precision() -> str:
    return os.environ.get("VQVS_PRECISION", "fp32")


def _groups(ch: int) -> int:
    g = 32
    while ch % g:
        g //= 2
    return g


def _scaled(mod: nn.Module, s: float) -> nn.Module:
    with torch.no_grad():
        for p in mod.parameters():
            p.mul_(s)
    return mod


class _Slot(nn.Module):
    """Parameter-free placeholder that keeps nn.Sequential indices aligned with the
    reference's module layout (activations, resizes and dropout own no tensors)."""

    def forward(self, x):  # pragma: no cover - containers are never called
        return x


def _seq(*mods: Optional[nn.Module]) -> nn.Sequential:
    return nn.Sequential(*[m if m is not None else _Slot() for m in mods])


class ResBlock(nn.Module):
    """Parameters of one residual block: pre_cond.{0.0,2,3}, cond_layers.1, post_cond.{1|2}, skip.1."""

    def __init__(self, channels: int, emb_channels: Optional[int] = None, out_channels: Optional[int] = None,
                 scale_factor: float = 1.0, dilation: int = 2, dropout: float = 0.0):
        super().__init__()
        self.channels = channels
        self.emb_channels = emb_channels
        self.out_channels = out_channels or channels
        self.scale_factor = scale_factor
        self.dilation = dilation
        self.dropout = dropout
        co = self.out_channels
        self.skip = _seq(None, nn.Conv1d(channels, co, 1) if channels != co else None)
        if emb_channels:
            self.cond_layers = _seq(None, _scaled(nn.Linear(emb_channels, 2 * co), 0.1))
        self.pre_cond = _seq(_seq(nn.GroupNorm(_groups(channels), channels), None), None,
                             nn.Conv1d(channels, co, 3, padding=1), nn.GroupNorm(_groups(co), co))
        conv2 = _scaled(nn.Conv1d(co, co, 3, padding=dilation, dilation=dilation), 0.0)
        self.post_cond = _seq(None, None, conv2) if dropout else _seq(None, conv2)


class _NativeModule(nn.Module):
    """Shared handle management: the device handle is (re)built lazily from the current
    parameters and dropped whenever they may have changed."""

    def __init__(self):
        super().__init__()
        self._handle = None
        self._handle_key = None
        self.precision = default_precision()
        self.debug_taps = False

    def set_precision(self, precision: str):
        if precision not in _native.PRECISIONS:
            raise ValueError(f"unknown precision {precision!r}; use 'fp32', 'fp16' or 'bf16'")
        self.precision = precision
        self.invalidate()
        return self

    def invalidate(self):
        if self._handle is not None:
            self._handle.close()
        self._handle = None
        self._handle_key = None
        for h, _ in self.__dict__.pop("_alt_handles", {}).values():
            if h is not None:
                h.close()
        for h in self.__dict__.pop("_cond_handles", {}).values():
            h.close()

    COND_HANDLES_KEPT = 3  # handles of other conditioning lengths kept beside the current one (small ones only: ALT_HANDLE_KEEP_BYTES)

    # an alternate-precision handle larger than this is closed when its override ends instead of waiting beside the module's own
    # (weights + an activation arena sized for the largest batch and length it has seen: several GB at 64 clips of 4 s)
    ALT_HANDLE_KEEP_BYTES = int(float(os.environ.get("VQVS_ALT_HANDLE_KEEP_GB", "2")) * 2 ** 30)

    @contextlib.contextmanager
    def precision_override(self, precision: str):
        """Run in another precision mode for the duration of a call WITHOUT discarding the module's own device handle: the module's
        own mode, handle and arena are back in place afterwards.  Used by VQVAE.decode_uncond_guidance, whose extrapolation needs the
        fp32 mode.  Memory: while the override is active BOTH handles exist (weights + arena each); afterwards the override's handle
        is kept for the next call only if it is small (ALT_HANDLE_KEEP_BYTES, VQVS_ALT_HANDLE_KEEP_GB, default 2 GB) -- a large one
        is closed, so repeated calls at full batch pay a rebuild (~0.3 s) instead of holding several GB; release_alt_handles()
        drops a kept one explicitly.  Nested overrides are allowed; a handle is never orphaned."""
        if precision not in _native.PRECISIONS:
            raise ValueError(f"unknown precision {precision!r}; use 'fp32', 'fp16' or 'bf16'")
        if precision == self.precision:
            yield self
            return
        own = (self.precision, self._handle, self._handle_key)
        alt = self.__dict__.setdefault("_alt_handles", {})
        h, k = alt.pop(precision, (None, None))
        self.precision, self._handle, self._handle_key = precision, h, k
        try:
            yield self
        finally:
            alt = self.__dict__.setdefault("_alt_handles", {})
            stale = alt.pop(precision, (None, None))[0]  # (left by a nested override of the same precision: close, do not orphan)
            if stale is not None and stale is not self._handle:
                stale.close()
            if self._handle is not None and self._handle.device_bytes() > self.ALT_HANDLE_KEEP_BYTES:
                self._handle.close()
            elif self._handle is not None:
                alt[precision] = (self._handle, self._handle_key)
            self.precision, self._handle, self._handle_key = own

    def release_alt_handles(self):
        """Close the handles kept by precision_override (device memory back to the driver)."""
        for h, _ in self.__dict__.pop("_alt_handles", {}).values():
            if h is not None:
                h.close()

    def _apply(self, fn, *a, **k):
        self.invalidate()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self.invalidate()
        return super().load_state_dict(*a, **k)

    def _load_from_state_dict(self, *a, **k):
        self.invalidate()
        return super()._load_from_state_dict(*a, **k)

    def _cfg(self) -> _native.Cfg:
        raise NotImplementedError

    def _weights_token(self):
        """Changes whenever a parameter or buffer is written in place through autograd-visible ops (optimizer steps, `p.mul_()`,
        `load_from_pretrained`, EMA copies -- they bump the tensor's version counter) or REPLACED (`mod.weight = nn.Parameter(..)`,
        `load_state_dict(assign=True)`): the device snapshot is then rebuilt instead of silently running old weights (the
        reference reads live parameters on every call).  The module tree is walked on every call, so a replaced tensor is seen.
        NOT seen: writes through `.data` (`p.data.copy_(..)` does not bump the version counter) -- call `invalidate()` after
        those."""
        tok = 0
        for t in list(self.parameters()) + list(self.buffers()):
            tok = (tok * 1000003 + t._version * 8191 + t.data_ptr() + id(t)) & 0xFFFFFFFFFFFF
        return tok

    def check_status(self) -> None:
        """Range guard (vqvs_model_status), one device sync, call once per sample.  Bit 0 -- a GroupNorm partial was not finite: an
        activation overflowed the storage type (or the input held NaN / inf); the result is garbage: raise.  Bit 1 (fp16 mode) -- a
        256-row tile's sum of squares reached 9e8: a WARNING, not an error -- it fires for a single element near 3e4 (half of
        fp16's 65504) but equally for a sustained RMS of ~1.9e3 over the tile, which fp16 holds without loss; a true overflow
        always sets bit 0.  VQVS_STRICT_RANGE=1 turns the warning into the error.  Tensors that feed no GroupNorm (the network's
        final output, the sampler's x_t) are not covered by either bit."""
        h = self._handle
        if h is None:
            return
        w = h.status()
        if w & 1:
            raise _native.NativeError("range guard: non-finite GroupNorm statistics (an activation overflowed the storage type, or the "
                                      "input held NaN/inf); run this model with set_precision('fp32')")
        if w & 2:
            msg = ("range guard: a tile's sum of squares reached 9e8 in the fp16 mode (an activation may be near fp16's limit of 65504); "
                   "consider set_precision('fp32')")
            if os.environ.get("VQVS_STRICT_RANGE", "0") == "1":
                raise _native.NativeError(msg)
            warnings.warn(msg, RuntimeWarning, stacklevel=2)

    def handle(self, device: torch.device, B: int, T: int) -> _native.Handle:
        idx = device.index if device.index is not None else torch.cuda.current_device()
        # (everything the native handle is BUILT from besides the shapes: the conditioning-length code is part of the key, so a handle
        #  restored by precision_override after the code changed inside the override is rebuilt instead of reused)
        key = (idx, self.precision, bool(self.debug_taps), self._weights_token(), getattr(self, "_cond_code", 0))
        h = self._handle
        if h is not None and self._handle_key == key and h.cfg.max_batch >= B and h.cfg.max_T >= T:
            return h
        if h is not None and self._handle_key[:4] == key[:4] and self._handle_key != key:
            # Only the conditioning-length code differs (a caller alternating between cond lengths, e.g. clips of different duration
            # encoded by one encoder): the handle is set aside instead of torn down -- a handle is built for one code (vqvs_cfg
            # reserved[3]) -- and taken back when its length returns.  Small handles only; the oldest goes first.
            kept = self.__dict__.setdefault("_cond_handles", {})
            if h.device_bytes() <= self.ALT_HANDLE_KEEP_BYTES:
                kept[self._handle_key] = h
                while len(kept) > self.COND_HANDLES_KEPT:
                    kept.pop(next(iter(kept))).close()
            else:
                h.close()
            self._handle, self._handle_key = None, None
            back = kept.pop(key, None)
            if back is not None and back.cfg.max_batch >= B and back.cfg.max_T >= T:
                self._handle, self._handle_key = back, key
                return back
            if back is not None:
                B, T = max(B, back.cfg.max_batch), max(T, back.cfg.max_T)
                back.close()
            h = None
        if h is not None:
            same_shape_class = self._handle_key[:3] == key[:3]
            B = max(B, h.cfg.max_batch) if same_shape_class else B
            T = max(T, h.cfg.max_T) if same_shape_class else T
            self.invalidate()
        cfg = self._cfg()
        cfg.precision = _native.PRECISIONS[self.precision]
        cfg.max_batch = B
        cfg.max_T = T
        cfg.debug_taps = 1 if self.debug_taps else 0
        torch.cuda.synchronize(idx)
        state = self.state_dict()
        q, qp = pad_blocks(getattr(self, "base_channels", 32)) if cfg.reserved[4] else (1, 1)
        if q != qp:  # a width that is not a multiple of 32: the handle takes the zero-padded physical shapes
            state = pad_state(state, _native.param_table(cfg), q, qp)
        self._handle = _native.Handle(cfg, state, "", idx)
        self._handle_key = key
        return self._handle

    # the ctypes handle cannot be pickled or deep-copied (EMA copies, torch.save(model)): drop it, the copy rebuilds its own
    def __getstate__(self):
        d = self.__dict__.copy()
        d["_handle"] = None
        d["_handle_key"] = None
        d.pop("_token_tensors", None)
        d.pop("_alt_handles", None)
        d.pop("_cond_handles", None)
        return d

    def __deepcopy__(self, memo):
        import copy

        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k in ("_token_tensors", "_alt_handles", "_cond_handles"):
                continue
            new.__dict__[k] = None if k in ("_handle", "_handle_key") else copy.deepcopy(v, memo)
        return new

    def _check_inference_only(self, *tensors):
        """forward() detaches its inputs and ignores dropout: say so instead of silently returning no gradients."""
        if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
            import warnings

            warnings.warn("the gfx950 UNet path is inference-only: inputs are detached, no gradient flows through forward()",
                          RuntimeWarning, stacklevel=3)
        if self.training and getattr(self, "dropout", 0.0):
            raise RuntimeError("the gfx950 UNet path does not implement dropout: call .eval() before sampling "
                               "(the reference applies dropout only in training, unet.py:295-300)")


class UNetPredictor(_NativeModule):
    def __init__(self, base_channels: int, channel_mult: Tuple[int, ...] = CHANNEL_MULT,
                 middle_dilations: Tuple[int, ...] = MIDDLE_DILATIONS, depth_mult: int = 2,
                 cond_channels: Optional[int] = None, num_labels: Optional[int] = None,
                 in_channels: int = 1, out_channels: int = 1, dropout: float = 0.0):
        super().__init__()
        check_base_channels(base_channels, pad_ok=True)
        check_topology(physical_base(base_channels), tuple(channel_mult), depth_mult, tuple(middle_dilations))
        if channel_mult[0] != 1:  # (the reference constructs such a model and fails in forward: its output head normalises base_channels, unet.py:113)
            raise ValueError("channel_mult[0] must be 1: the output head (GroupNorm + conv) is built for base_channels")
        self.base_channels = base_channels
        self.channel_mult = tuple(channel_mult)
        self.middle_dilations = tuple(middle_dilations)
        self.depth_mult = depth_mult
        self.cond_channels = cond_channels
        self.num_labels = num_labels
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.dropout = dropout
        self._cond_code = 0  # conditioning rows per clip: 0 = T / 256 (UNet encoder), 1 = T / 320 (ConvMFCCEncoder), 1000 + L = exactly L rows

        C = base_channels
        E = 4 * C
        self.time_embed = nn.Module()
        self.time_embed.proj = nn.Linear(E, E)
        self.time_embed.channels = E
        self.time_embed_extra = _seq(None, nn.Linear(E, E))
        if num_labels is not None:
            self.class_embed = nn.Embedding(num_labels, E)
        if cond_channels is not None:
            self.cond_proj = nn.Conv1d(cond_channels, C, 3, padding=1)
        self.in_conv = nn.Conv1d(in_channels, C, 3, padding=1)

        down, stack, cur = [], [C], C
        last = len(channel_mult) - 1
        for depth, mult in enumerate(channel_mult):
            for _ in range(depth_mult):
                down.append(ResBlock(cur, E, mult * C, dropout=dropout))
                cur = mult * C
                stack.append(cur)
            if depth != last:
                down.append(ResBlock(cur, E, scale_factor=0.5, dropout=dropout))
                stack.append(cur)
        self.down_blocks = nn.ModuleList(down)
        self.middle_blocks = nn.ModuleList([ResBlock(cur, E, dilation=d, dropout=dropout) for d in middle_dilations])
        up = []
        for depth in range(last, -1, -1):
            mult = channel_mult[depth]
            for _ in range(depth_mult + 1):
                up.append(ResBlock(cur + stack.pop(), E, mult * C, dropout=dropout))
                cur = mult * C
            if depth:
                up.append(ResBlock(cur, E, scale_factor=2.0, dropout=dropout))
        self.up_blocks = nn.ModuleList(up)
        self.out = _seq(_seq(nn.GroupNorm(_groups(C), C), None), nn.Conv1d(C, out_channels, 3, padding=1))

    def _cfg(self) -> _native.Cfg:
        cfg = _native.Cfg()
        cfg.kind = _native.KIND_PREDICTOR
        cfg.base_channels = physical_base(self.base_channels)
        cfg.reserved[4] = self.base_channels if cfg.base_channels != self.base_channels else 0  # (a padded handle: csrc/net.cpp pad_map)
        cfg.in_channels = self.in_channels
        cfg.out_channels = self.out_channels
        cfg.cond_channels = self.cond_channels or 0
        cfg.num_labels = self.num_labels or 0
        cfg.reserved[0] = 1 if self.dropout else 0
        cfg.reserved[3] = self._cond_code
        if (self.channel_mult, self.middle_dilations, self.depth_mult) != (CHANNEL_MULT, MIDDLE_DILATIONS, 2):
            cfg.set_topology(self.channel_mult, self.depth_mult, self.middle_dilations)
        return cfg

    def forward(self, x: torch.Tensor, ts: torch.Tensor, cond: Optional[torch.Tensor] = None,
                labels: Optional[torch.Tensor] = None, use_checkpoint: bool = False) -> torch.Tensor:
        assert (labels is None) == (self.num_labels is None), "must provide labels if and only if model is class conditional"
        assert (cond is None) == (self.cond_channels is None), "must provide cond sequence if and only if model is conditional"
        _native.require_cuda(x, ts, cond, labels)
        self._check_inference_only(x, cond)
        if x.dim() != 3 or x.shape[1] != self.in_channels:
            raise ValueError(f"expected x of shape [N, {self.in_channels}, T], got {tuple(x.shape)}")
        B, _, T = x.shape
        if T % self.downsample_rate:
            raise ValueError(f"T={T} is not a multiple of the UNet downsample rate {self.downsample_rate}")
        x = x.detach().to(torch.float32).contiguous()
        ts = ts.detach().to(device=x.device, dtype=torch.float32).contiguous()
        if ts.shape != (B,):
            raise ValueError(f"expected ts of shape [{B}], got {tuple(ts.shape)}")
        if cond is not None:
            cond = cond.detach().to(torch.float32).contiguous()
            # the reference up-samples ANY cond length to T (F.interpolate(cond, T), nearest; unet.py:138-139), and so does in_conv.
            # The two lengths the encoders produce -- T / 256 (UNetEncoder) and (T / 160 + 1 - 2) / 2 + 1 = T / 320 (ConvMFCCEncoder)
            # -- are length CODES of the handle (valid for every T); any other length builds a handle for exactly that many rows.
            if tuple(cond.shape[:2]) != (B, self.cond_channels) or cond.shape[2] < 1:
                raise ValueError(f"expected cond of shape {(B, self.cond_channels)} x L, got {tuple(cond.shape)}")
            lens = {(T // 160 + 1 - 2) // 2 + 1: 1, T // 256: 0}
            code = lens.get(cond.shape[2], 1000 + cond.shape[2])
            self._cond_code = code  # (part of the handle's key: handle() builds, or takes back, the handle of this length)
        if labels is not None:
            labels = labels.detach().to(device=x.device, dtype=torch.int64).contiguous()
            if labels.shape != (B,):
                raise ValueError(f"expected labels of shape [{B}], got {tuple(labels.shape)}")
            _native.check_index_range(labels, self.num_labels, "class labels")
        h = self.handle(x.device, B, T)
        out = torch.empty(B, self.out_channels, T, device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _native.check(_native.lib().vqvs_unet_forward(
                h.ptr, x.data_ptr(), ts.data_ptr(), _native._ptr(cond), _native._ptr(labels), out.data_ptr(), B, T,
                _native._stream_ptr()))
        return out

    def condition(self, **kwargs) -> Callable:
        return functools.partial(self, **kwargs)

    def add_labels(self, n: int, end: bool = True):
        assert self.num_labels is not None
        old = self.class_embed.weight.detach()
        count = self.num_labels
        self.num_labels += n
        self.class_embed = nn.Embedding(self.num_labels, old.shape[-1]).to(old.device)
        with torch.no_grad():
            if end:
                self.class_embed.weight[:count].copy_(old)
            else:
                self.class_embed.weight[n:].copy_(old)
        self.invalidate()

    def label_parameters(self) -> List[nn.Parameter]:
        assert self.num_labels is not None
        return list(self.class_embed.parameters())

    @property
    def downsample_rate(self) -> int:
        return 2 ** (len(self.channel_mult) - 1)


class UNetEncoder(_NativeModule):
    def __init__(self, base_channels: int, channel_mult: Tuple[int, ...] = CHANNEL_MULT, out_dilations: Tuple[int, ...] = (),
                 depth_mult: int = 2, in_channels: int = 1, out_channels: int = 512):
        super().__init__()
        check_base_channels(base_channels, pad_ok=True)
        check_topology(physical_base(base_channels), tuple(channel_mult), depth_mult, tuple(out_dilations))
        self.base_channels = base_channels
        self.channel_mult = tuple(channel_mult)
        self.out_dilations = tuple(out_dilations)
        self.depth_mult = depth_mult
        self.in_channels = in_channels
        self.out_channels = out_channels
        C = base_channels
        self.in_conv = nn.Conv1d(in_channels, C, 3, padding=1)
        blocks, cur = [], C
        last = len(channel_mult) - 1
        for depth, mult in enumerate(channel_mult):
            for _ in range(depth_mult):
                blocks.append(ResBlock(cur, None, mult * C))
                cur = mult * C
            if depth != last:
                blocks.append(ResBlock(cur, None, scale_factor=0.5))
        for d in self.out_dilations:  # unet.py:219-220
            blocks.append(ResBlock(cur, None, dilation=d))
        self.blocks = nn.ModuleList(blocks)
        self.out = _seq(_seq(nn.GroupNorm(_groups(cur), cur), None), nn.Conv1d(cur, out_channels, 3, padding=1))

    def _cfg(self) -> _native.Cfg:
        cfg = _native.Cfg()
        cfg.kind = _native.KIND_ENCODER
        cfg.base_channels = physical_base(self.base_channels)
        cfg.reserved[4] = self.base_channels if cfg.base_channels != self.base_channels else 0  # (a padded handle: csrc/net.cpp pad_map)
        cfg.in_channels = self.in_channels
        cfg.out_channels = self.out_channels
        if (self.channel_mult, self.out_dilations, self.depth_mult) != (CHANNEL_MULT, (), 2):
            cfg.set_topology(self.channel_mult, self.depth_mult, self.out_dilations)
        return cfg

    def forward(self, x: torch.Tensor, use_checkpoint: bool = False) -> torch.Tensor:
        _native.require_cuda(x)
        if x.dim() != 3 or x.shape[1] != self.in_channels:
            raise ValueError(f"expected x of shape [N, {self.in_channels}, T], got {tuple(x.shape)}")
        B, _, T = x.shape
        if T % self.downsample_rate:
            raise ValueError(f"T={T} is not a multiple of the UNet downsample rate {self.downsample_rate}")
        x = x.detach().to(torch.float32).contiguous()
        h = self.handle(x.device, B, T)
        z = torch.empty(B, self.out_channels, T // self.downsample_rate, device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _native.check(_native.lib().vqvs_encoder_forward(h.ptr, x.data_ptr(), z.data_ptr(), B, T, _native._stream_ptr()))
        return z

    @property
    def downsample_rate(self) -> int:
        return 2 ** (len(self.channel_mult) - 1)


class ResBlockModule(_NativeModule):
    """A single residual block behind `vqvs_resblock_forward` (unit-test granularity)."""

    def __init__(self, channels: int, emb_channels: Optional[int] = None, out_channels: Optional[int] = None,
                 scale_factor: float = 1.0, dilation: int = 2):
        super().__init__()
        self.block = ResBlock(channels, emb_channels, out_channels, scale_factor, dilation)

    def state_dict(self, *a, **k):
        return self.block.state_dict(*a, **k)

    def _cfg(self) -> _native.Cfg:
        b = self.block
        cfg = _native.Cfg()
        cfg.kind = _native.KIND_RESBLOCK
        cfg.base_channels = 32
        cfg.in_channels = 1
        cfg.rb_cin, cfg.rb_cout = b.channels, b.out_channels
        cfg.rb_resize = 0 if b.scale_factor == 1.0 else (1 if b.scale_factor < 1.0 else 2)
        cfg.rb_dilation = b.dilation
        cfg.rb_emb_channels = b.emb_channels or 0
        return cfg

    def forward(self, x: torch.Tensor, emb: Optional[torch.Tensor] = None) -> torch.Tensor:
        _native.require_cuda(x, emb)
        B, _, L = x.shape
        x = x.detach().to(torch.float32).contiguous()
        emb = None if emb is None else emb.detach().to(torch.float32).contiguous()
        h = self.handle(x.device, B, L)
        b = self.block
        Lo = L if b.scale_factor == 1.0 else (L // 2 if b.scale_factor < 1.0 else L * 2)
        y = torch.empty(B, b.out_channels, Lo, device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _native.check(_native.lib().vqvs_resblock_forward(h.ptr, x.data_ptr(), _native._ptr(emb), y.data_ptr(), B, L,
                                                             _native._stream_ptr()))
        return y
