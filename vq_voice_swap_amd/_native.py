"""
ctypes binding of the C-ABI library `libvqvs_hip.so` (declared in include/vqvs.h).

There is no CPU fallback: if the library is missing, cannot be loaded, or a call
fails, this module raises.  The library is built in-tree by `build()` (hipcc,
--offload-arch=gfx950) so it travels with the source tree.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, List, Optional, Sequence, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VQVS_LIB_PATH") or os.path.join(_HERE, "libvqvs_hip.so")  # override: instrumented builds (tools/)
CSRC = os.path.join(_HERE, "csrc")

KIND_PREDICTOR, KIND_ENCODER, KIND_RESBLOCK, KIND_CLASSIFIER, KIND_ENCPRED, KIND_MFCC_ENCODER = 0, 1, 2, 3, 4, 5
PREC_F32, PREC_BF16, PREC_F16 = 0, 1, 2
DDPM_SIGMA_LARGE, DDPM_CONSTRAIN = 1, 2
PRECISIONS = {"fp32": PREC_F32, "f32": PREC_F32, "float32": PREC_F32, "bf16": PREC_BF16, "bfloat16": PREC_BF16,
              "fp16": PREC_F16, "f16": PREC_F16, "float16": PREC_F16, "half": PREC_F16}


class Cfg(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("base_channels", C.c_int32),
        ("in_channels", C.c_int32),
        ("out_channels", C.c_int32),
        ("cond_channels", C.c_int32),
        ("num_labels", C.c_int32),
        ("precision", C.c_int32),
        ("max_batch", C.c_int32),
        ("max_T", C.c_int32),
        ("debug_taps", C.c_int32),
        ("rb_cin", C.c_int32),
        ("rb_cout", C.c_int32),
        ("rb_resize", C.c_int32),
        ("rb_dilation", C.c_int32),
        ("rb_emb_channels", C.c_int32),
        ("reserved", C.c_int32 * 5),  # reserved[0] = 1: checkpoints trained with dropout (conv is post_cond.2)
        # topology of predictor / encoder handles (include/vqvs.h); topology_set = 0: the reference's defaults
        ("topology_set", C.c_int32),
        ("n_levels", C.c_int32),
        ("channel_mult", C.c_int32 * 12),
        ("depth_mult", C.c_int32),
        ("n_dilations", C.c_int32),
        ("dilations", C.c_int32 * 12),
    ]

    def set_topology(self, channel_mult, depth_mult, dilations):
        """Describe a UNet other than the reference's default one (unet.py:17-30, 188-196)."""
        if len(channel_mult) > 12 or len(dilations) > 12:
            raise ValueError("the gfx950 library builds at most 12 levels and 12 middle / output dilations")
        self.topology_set = 1
        self.n_levels = len(channel_mult)
        for i, m in enumerate(channel_mult):
            self.channel_mult[i] = int(m)
        self.depth_mult = int(depth_mult)
        self.n_dilations = len(dilations)
        for i, d in enumerate(dilations):
            self.dilations[i] = int(d)


EXPORTS = [
    "vqvs_param_count", "vqvs_param_info", "vqvs_model_create", "vqvs_model_destroy", "vqvs_model_device_bytes",
    "vqvs_unet_forward", "vqvs_encoder_forward", "vqvs_mfcc_encoder_forward", "vqvs_mfcc_encoder_forward_logmel", "vqvs_model_status", "vqvs_resblock_forward", "vqvs_classifier_forward",
    "vqvs_classifier_guidance", "vqvs_encpred_forward", "vqvs_encpred_guidance", "vqvs_ddpm_step", "vqvs_ddpm_mean",
    "vqvs_ddpm_guided_eps", "vqvs_randn", "vqvs_vq_argmin", "vqvs_vq_embed", "vqvs_debug_tap_count",
    "vqvs_debug_tap_info", "vqvs_debug_tap_rows", "vqvs_debug_read_tap", "vqvs_debug_read_embedding", "vqvs_forward_kernel_count", "vqvs_forward_model_bytes",
    "vqvs_forward_flops", "vqvs_set_profiling", "vqvs_op_info", "vqvs_op_desc", "vqvs_profile_read", "vqvs_last_error", "vqvs_version",
]

_lib = None


class NativeError(RuntimeError):
    pass


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into libvqvs_hip.so (in-tree)."""
    cmd = ["make", "-C", CSRC, "-j", str(min(8, os.cpu_count() or 1))]
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, capture_output=not verbose)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise NativeError("building libvqvs_hip.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    if verbose:
        print(r.stdout)
    return LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(
            f"{LIB_PATH} is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C vq_voice_swap_amd/csrc`). There is no CPU fallback for the sampling hot path."
        )
    if os.environ.get("HIP_FORCE_DEV_KERNARG") is None:
        import warnings

        warnings.warn("HIP_FORCE_DEV_KERNARG is not set: the HIP runtime keeps kernel arguments in host memory and every launch of the "
                      "sampler starts with loads across PCIe (~2 % slower, DESIGN.md section 7).  Set HIP_FORCE_DEV_KERNARG=1 in the "
                      "process environment before the first HIP call (bench.py and the sample_*.py scripts do).", RuntimeWarning, stacklevel=2)
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, u32, u64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_uint32, C.c_uint64, C.c_float
    L.vqvs_last_error.restype = C.c_char_p
    L.vqvs_version.restype = C.c_char_p
    L.vqvs_param_count.argtypes = [C.POINTER(Cfg)]
    L.vqvs_param_info.argtypes = [C.POINTER(Cfg), i32, C.c_char_p, i32, C.POINTER(i64), C.POINTER(i32)]
    L.vqvs_model_create.argtypes = [C.POINTER(Cfg), C.POINTER(vp), i32, i32, C.POINTER(vp)]
    L.vqvs_model_destroy.argtypes = [vp]
    L.vqvs_model_destroy.restype = None
    L.vqvs_model_device_bytes.argtypes = [vp]
    L.vqvs_model_device_bytes.restype = i64
    L.vqvs_unet_forward.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, vp]
    L.vqvs_encoder_forward.argtypes = [vp, vp, vp, i32, i32, vp]
    L.vqvs_mfcc_encoder_forward.argtypes = [vp, vp, vp, i32, i32, vp]
    L.vqvs_mfcc_encoder_forward_logmel.argtypes = [vp, vp, vp, i32, i32, vp]
    L.vqvs_model_status.argtypes = [vp, vp]
    L.vqvs_resblock_forward.argtypes = [vp, vp, vp, vp, i32, i32, vp]
    L.vqvs_classifier_forward.argtypes = [vp, vp, vp, vp, i32, i32, vp]
    L.vqvs_classifier_guidance.argtypes = [vp, vp, vp, vp, f32, vp, vp, i32, i32, vp]
    L.vqvs_encpred_forward.argtypes = [vp, vp, vp, vp, i32, i32, vp]
    L.vqvs_encpred_guidance.argtypes = [vp, vp, vp, vp, f32, vp, vp, i32, i32, vp]
    L.vqvs_ddpm_step.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, u32, f32, u64, u64, u32, vp]
    L.vqvs_ddpm_mean.argtypes = [vp, vp, vp, vp, vp, i32, i32, vp]
    L.vqvs_ddpm_guided_eps.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, u32, vp]
    L.vqvs_randn.argtypes = [vp, i32, i32, u64, u64, u32, vp]
    L.vqvs_vq_argmin.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp]
    L.vqvs_vq_embed.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp]
    L.vqvs_debug_tap_count.argtypes = [vp]
    L.vqvs_debug_tap_info.argtypes = [vp, i32, C.c_char_p, i32, C.POINTER(i32), C.POINTER(i32)]
    L.vqvs_debug_tap_rows.argtypes = [vp, i32, i32]
    L.vqvs_debug_read_tap.argtypes = [vp, i32, i32, i32, vp]
    L.vqvs_debug_read_embedding.argtypes = [vp, i32, vp]
    L.vqvs_forward_kernel_count.argtypes = [vp]
    L.vqvs_forward_model_bytes.argtypes = [vp, i32, i32]
    L.vqvs_forward_model_bytes.restype = i64
    L.vqvs_forward_flops.argtypes = [vp, i32, i32]
    L.vqvs_forward_flops.restype = i64
    L.vqvs_set_profiling.argtypes = [vp, i32]
    L.vqvs_op_info.argtypes = [vp, i32, C.c_char_p, i32, C.POINTER(i64), C.POINTER(i64), i32, i32]
    L.vqvs_profile_read.argtypes = [vp, vp, i32]
    L.vqvs_op_desc.argtypes = [vp, i32, C.c_char_p, i32]
    _lib = L
    return L


def last_error() -> str:
    return (lib().vqvs_last_error() or b"").decode("utf-8", "replace")


def check(rc: int) -> None:
    """Map the ABI's error codes to the exception types the reference raises
    (AssertionError for the forward-argument asserts of unet.py:126-131, ValueError for
    shape problems, RuntimeError for HIP failures)."""
    if rc == 0:
        return
    msg = last_error()
    if rc == -1:
        if msg.startswith("must provide"):
            raise AssertionError(msg)
        raise ValueError(msg)
    raise NativeError(f"libvqvs_hip error {rc}: {msg}")


def param_table(cfg: Cfg) -> List[Tuple[str, Tuple[int, ...]]]:
    L = lib()
    n = L.vqvs_param_count(C.byref(cfg))
    if n < 0:
        check(n)
    out = []
    name = C.create_string_buffer(256)
    shape = (C.c_int64 * 4)()
    nd = C.c_int()
    for i in range(n):
        check(L.vqvs_param_info(C.byref(cfg), i, name, 256, shape, C.byref(nd)))
        out.append((name.value.decode(), tuple(int(shape[k]) for k in range(nd.value))))
    return out


def _stream_ptr() -> int:
    import torch

    return torch.cuda.current_stream().cuda_stream


def _ptr(t) -> Optional[int]:
    return None if t is None else t.data_ptr()


class Handle:
    """Owns one `vqvs_model*`."""

    def __init__(self, cfg: Cfg, state: Dict[str, "object"], prefix: str, device_index: int):
        import torch

        self.cfg = cfg
        self.device_index = device_index
        table = param_table(cfg)
        keep = []
        ptrs = (C.c_void_p * len(table))()
        for i, (name, shape) in enumerate(table):
            key = prefix + name
            if key not in state:
                raise KeyError(f"state dict has no parameter {key!r} required by the gfx950 model")
            t = state[key].detach().to(device="cpu", dtype=torch.float32).contiguous()
            if tuple(t.shape) != shape:
                raise ValueError(f"parameter {key}: expected shape {shape}, got {tuple(t.shape)}")
            keep.append(t)
            ptrs[i] = t.data_ptr()
        h = C.c_void_p()
        check(lib().vqvs_model_create(C.byref(cfg), ptrs, len(table), device_index, C.byref(h)))
        self._h = h
        del keep

    @property
    def ptr(self):
        return self._h

    def status(self) -> int:
        """Device status word (vqvs_model_status): read and cleared; synchronises the device."""
        w = C.c_uint(0)
        check(lib().vqvs_model_status(self.ptr, C.byref(w)))
        return int(w.value)

    def close(self):
        if getattr(self, "_h", None):
            lib().vqvs_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- introspection
    def device_bytes(self) -> int:
        return int(lib().vqvs_model_device_bytes(self._h))

    def kernel_count(self) -> int:
        return int(lib().vqvs_forward_kernel_count(self._h))

    def model_bytes(self, B: int, T: int) -> int:
        return int(lib().vqvs_forward_model_bytes(self._h, B, T))

    def flops(self, B: int, T: int) -> int:
        return int(lib().vqvs_forward_flops(self._h, B, T))

    def set_profiling(self, on: bool) -> None:
        check(lib().vqvs_set_profiling(self._h, 1 if on else 0))

    def op_info(self, B: int, T: int) -> List[Tuple[str, int, int]]:
        """(kind, algorithmic bytes, flops) of every kernel one forward enqueues."""
        L = lib()
        kind = C.create_string_buffer(64)
        by, fl = C.c_int64(), C.c_int64()
        out = []
        for i in range(self.kernel_count()):
            check(L.vqvs_op_info(self._h, i, kind, 64, C.byref(by), C.byref(fl), B, T))
            out.append((kind.value.decode(), by.value, fl.value))
        return out

    def op_desc(self) -> List[str]:
        buf = C.create_string_buffer(256)
        out = []
        for i in range(self.kernel_count()):
            check(lib().vqvs_op_desc(self._h, i, buf, 256))
            out.append(buf.value.decode())
        return out

    def profile_read(self) -> List[float]:
        n = self.kernel_count()
        buf = (C.c_float * n)()
        r = lib().vqvs_profile_read(self._h, buf, n)
        if r < 0:
            check(r)
        return [float(buf[i]) for i in range(r)]

    def taps(self) -> List[Tuple[str, int, int]]:
        L = lib()
        n = L.vqvs_debug_tap_count(self._h)
        name = C.create_string_buffer(256)
        ch, ls = C.c_int(), C.c_int()
        out = []
        for i in range(n):
            check(L.vqvs_debug_tap_info(self._h, i, name, 256, C.byref(ch), C.byref(ls)))
            out.append((name.value.decode(), ch.value, ls.value))
        return out

    def read_embedding(self, B: int):
        """Conditioning vector [B, 4*base] of the last forward (time embedding [+ class embedding])."""
        import torch

        out = torch.empty(B, 4 * self.cfg.base_channels, dtype=torch.float32)
        r = lib().vqvs_debug_read_embedding(self._h, B, out.data_ptr())
        if r < 0:
            check(r)
        return out[:, :r]

    def read_tap(self, i: int, B: int, T: int):
        import torch

        _, ch, _ls = self.taps()[i]
        Lx = lib().vqvs_debug_tap_rows(self._h, i, T)
        if Lx < 0:
            check(Lx)
        out = torch.empty(B, ch, Lx, dtype=torch.float32)
        check(lib().vqvs_debug_read_tap(self._h, i, B, T, out.data_ptr()))
        return out


_range_checked = {}


def check_index_range(t, n: int, what: str) -> None:
    """Raise IndexError, as nn.Embedding / F.embedding do (reference unet.py:45, vq.py:108), when an index tensor holds
    values outside [0, n).  The kernels clamp instead of faulting, which would turn a caller bug into plausible audio.
    The check costs one device->host sync, so the SAME tensor object at the same version (the labels of a sampling loop, passed
    again every step) is not checked twice; a new tensor -- even one the allocator places at the old address -- always is."""
    import weakref

    prev = _range_checked.get(what)
    if prev is not None and prev[0]() is t and prev[1:] == (t._version, t.numel(), n):
        return
    if t.numel():
        lo, hi = int(t.min().item()), int(t.max().item())
        if lo < 0 or hi >= n:
            raise IndexError(f"{what}: index out of range (values span [{lo}, {hi}], valid range is [0, {n - 1}])")
    _range_checked[what] = (weakref.ref(t), t._version, t.numel(), n)


def require_cuda(*tensors) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise NativeError(
                "the gfx950 sampling path needs tensors on a ROCm device (got a CPU tensor); "
                "there is deliberately no CPU fallback"
            )
