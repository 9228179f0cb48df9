"""
Continuous-time DDPM process with the reverse step running as fused HIP kernels.

Mirrors the reference's `Diffusion` / `Schedule` API (reference
vq_voice_swap/diffusion/diffusion.py:9-151, schedule.py:7-41, make.py:4-13).  The
sampler loop and `ddpm_previous` call `vqvs_ddpm_step` (one fused kernel instead of
~20 full-size elementwise ops with materialised broadcasts, diffusion.py:62-90,154-157).

Additions over the reference (it has no seeding):
  * `noise=` may be passed explicitly to `ddpm_sample` as a list / callable, which is how
    the parity tests drive oracle and HIP path with identical draws;
  * otherwise noise comes from an in-kernel counter-based generator keyed by
    (seed, global clip index, step), so results do not depend on how a batch is sharded.
"""

from __future__ import annotations

import math
from typing import Callable, List, Optional, Sequence, Union

import torch

from . import _native


class Schedule:
    def __call__(self, t: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError


class ExpSchedule(Schedule):
    """alpha_bar(t) = exp(-k t^2), k = -ln(alpha_final) (schedule.py:15-31)."""

    def __init__(self, alpha_final: float = 1e-5):
        self.alpha_final = alpha_final
        self.k = -math.log(alpha_final)

    def __call__(self, t: torch.Tensor) -> torch.Tensor:
        return torch.exp(-self.k * (t ** 2))


class CosSchedule(Schedule):
    """alpha_bar(t) = cos(pi t / 2)^2 (schedule.py:34-41)."""

    def __call__(self, t: torch.Tensor) -> torch.Tensor:
        return torch.cos(t * math.pi / 2) ** 2


def make_schedule(name: str) -> Schedule:
    if name == "exp":
        return ExpSchedule()
    if name == "cos":
        return CosSchedule()
    raise ValueError(f"unknown schedule: {name}")


def _rows(v: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
    return v.to(like).reshape(-1, *([1] * (like.dim() - 1)))


NoiseSource = Union[None, Sequence[torch.Tensor], Callable[[int], torch.Tensor]]

FEW_GUIDED_STEPS = 10  # a guided run of fewer steps is promoted to the fp32 mode (ddpm_sample)


def _native_modules(fn) -> list:
    """The native (precision-mode) modules behind a predictor / cond_fn callable: the module itself, the target of a
    functools.partial (`UNetPredictor.condition`), the owner of a bound method, or what a closure advertises as `native_modules`
    (`Classifier.guidance_fn`).  Anything else -- a plain Python function -- has none: nothing is promoted."""
    seen, out = set(), []

    def add(m):
        if m is not None and hasattr(m, "precision_override") and id(m) not in seen:
            seen.add(id(m))
            out.append(m)

    add(fn)
    add(getattr(fn, "func", None))
    add(getattr(fn, "__self__", None))
    add(getattr(getattr(fn, "func", None), "__self__", None))
    for m in getattr(fn, "native_modules", ()) or ():
        add(m)
    return out


class Diffusion:
    def __init__(self, schedule: Schedule):
        self.schedule = schedule

    # ---- light helpers (training side; plain tensor expressions) -----------------------
    def sample_q(self, x_0: torch.Tensor, ts: torch.Tensor, epsilon: Optional[torch.Tensor] = None) -> torch.Tensor:
        if epsilon is None:
            epsilon = torch.randn_like(x_0)
        a = _rows(self.schedule(ts), x_0)
        return a.sqrt() * x_0 + (1 - a).sqrt() * epsilon

    def eps_to_x0(self, x_t: torch.Tensor, ts: torch.Tensor, epsilon_prediction: torch.Tensor) -> torch.Tensor:
        a = _rows(self.schedule(ts), x_t)
        return (x_t - (1 - a).sqrt() * epsilon_prediction) * a.rsqrt()

    def x0_to_eps(self, x_t: torch.Tensor, ts: torch.Tensor, x_0: torch.Tensor) -> torch.Tensor:
        a = _rows(self.schedule(ts), x_t)
        return (x_t - x_0 * a.sqrt()) * (1 - a).rsqrt()

    def ddpm_losses(self, x, predictor, ts=None, noise=None) -> torch.Tensor:
        if ts is None:
            ts = torch.rand(len(x), device=x.device)
        if noise is None:
            noise = torch.randn_like(x)
        pred = predictor(self.sample_q(x, ts, epsilon=noise), ts)
        return ((noise - pred) ** 2).flatten(1).mean(dim=1)

    # ---- hot path -------------------------------------------------------------------------
    def ddpm_previous(
        self,
        x_t: torch.Tensor,
        ts: torch.Tensor,
        step,
        epsilon_prediction: torch.Tensor,
        noise: Optional[torch.Tensor] = None,
        sigma_large: bool = False,
        constrain: bool = False,
        cond_fn: Optional[Callable] = None,
        *,
        seed: int = 0,
        clip_offset: int = 0,
        step_index: int = 0,
        noise_scale: float = 1.0,
    ) -> torch.Tensor:
        """x_{t-step} ~ p(. | x_t) (reference diffusion.py:48-90).  `noise=None` draws from the
        in-kernel generator keyed by (seed, clip_offset + row, step_index)."""
        _native.require_cuda(x_t, epsilon_prediction, noise)
        ts = ts.detach().to(device=x_t.device, dtype=torch.float32)
        if not torch.is_tensor(step):
            step = torch.full_like(ts, float(step))
        step = step.to(ts)
        return self._step(x_t, epsilon_prediction, self.schedule(ts).contiguous(), self.schedule(ts - step).contiguous(), ts - step,
                          noise=noise, sigma_large=sigma_large, constrain=constrain, cond_fn=cond_fn, seed=seed, clip_offset=clip_offset,
                          step_index=step_index, noise_scale=noise_scale)

    def _step(self, x_t, epsilon_prediction, a_t, a_prev, ts_prev, *, noise, sigma_large, constrain, cond_fn, seed, clip_offset,
              step_index, noise_scale) -> torch.Tensor:
        """The reverse step given alpha_bar(t) and alpha_bar(t - step) as [B] device tensors."""
        if x_t.dim() < 2:
            raise ValueError("x_t must be [N, ..., T]")
        B = x_t.shape[0]
        T = x_t[0].numel()
        x = x_t.detach().to(torch.float32).contiguous()
        eps = epsilon_prediction.detach().to(torch.float32).contiguous()
        flags = (_native.DDPM_SIGMA_LARGE if sigma_large else 0) | (_native.DDPM_CONSTRAIN if constrain else 0)
        L = _native.lib()
        st = _native._stream_ptr()
        with torch.cuda.device(x.device):
            if cond_fn is not None:  # diffusion.py:80-83
                mean = torch.empty_like(x)
                _native.check(L.vqvs_ddpm_mean(x.data_ptr(), eps.data_ptr(), a_t.data_ptr(), a_prev.data_ptr(), mean.data_ptr(), B, T, st))
                grad = cond_fn(mean.view_as(x_t), ts_prev).detach().to(torch.float32).contiguous()
                eps2 = torch.empty_like(x)
                _native.check(L.vqvs_ddpm_guided_eps(x.data_ptr(), mean.data_ptr(), grad.data_ptr(), a_t.data_ptr(), a_prev.data_ptr(),
                                                     eps2.data_ptr(), B, T, flags, st))
                eps = eps2
            if noise is not None:
                noise = noise.detach().to(torch.float32).contiguous()
            out = torch.empty_like(x)
            _native.check(L.vqvs_ddpm_step(x.data_ptr(), eps.data_ptr(), _native._ptr(noise), a_t.data_ptr(), a_prev.data_ptr(),
                                           out.data_ptr(), B, T, flags, float(noise_scale), int(seed), int(clip_offset),
                                           int(step_index), st))
        return out.view_as(x_t)

    def ddpm_sample(
        self,
        x_T: torch.Tensor,
        predictor: Callable[[torch.Tensor, torch.Tensor], torch.Tensor],
        steps: int,
        progress: bool = False,
        sigma_large: bool = False,
        constrain: bool = False,
        cond_fn: Optional[Callable] = None,
        schedule: Optional[Callable] = None,
        *,
        noise: NoiseSource = None,
        seed: Optional[int] = None,
        clip_offset: int = 0,
    ) -> torch.Tensor:
        """Reverse diffusion from x_T (reference diffusion.py:92-133): t runs steps/steps ... 1/steps,
        optional sample-time remap `schedule`, zero noise on the last iteration."""
        _native.require_cuda(x_T)
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        # Few guided steps in a 2-byte mode: the first reverse step multiplies the predictor's rounding error by 1 / sqrt(alpha_bar(1))
        # and a guided run adds the classifier gradient's own; fewer than FEW_GUIDED_STEPS iterations never average that out (measured:
        # config 5 at 3 steps 1.05e-3 in fp16 against the 1e-3 waveform contract, profiles/r05_parity_margins.jsonl).  The call is then
        # promoted to the fp32 mode -- predictor and guidance model -- through precision_override, which keeps each module's own
        # handle; VQVAE.decode_uncond_guidance does the same for its extrapolation.  (VQVS_FEW_STEP_PROMOTE=0: warn only.)
        promote = []
        if cond_fn is not None and steps < FEW_GUIDED_STEPS:
            promote = [m for m in _native_modules(predictor) + _native_modules(cond_fn) if getattr(m, "precision", "fp32") != "fp32"]
        if promote:
            import contextlib
            import os
            import warnings

            modes = sorted({m.precision for m in promote})
            if os.environ.get("VQVS_FEW_STEP_PROMOTE", "1") == "0":
                warnings.warn(f"ddpm_sample: {steps} guided steps in the {modes} mode(s) are outside the 1e-3 waveform contract "
                              f"(fewer than {FEW_GUIDED_STEPS} steps); VQVS_FEW_STEP_PROMOTE=0 keeps the mode", RuntimeWarning, stacklevel=2)
            else:
                warnings.warn(f"ddpm_sample: {steps} guided steps (fewer than {FEW_GUIDED_STEPS}): predictor / guidance model run in the fp32 "
                              f"mode for this call (their {modes} mode(s) do not hold the 1e-3 waveform contract at so few steps)",
                              RuntimeWarning, stacklevel=2)
                with contextlib.ExitStack() as stack:
                    for m in promote:
                        stack.enter_context(m.precision_override("fp32"))
                    return self.ddpm_sample(x_T, predictor, steps, progress=progress, sigma_large=sigma_large, constrain=constrain,
                                            cond_fn=cond_fn, schedule=schedule, noise=noise, seed=seed, clip_offset=clip_offset)
        x_t = x_T
        B = x_T.shape[0]
        t_values = [(i + 1) / steps for i in range(steps)][::-1]
        # Per-step scalars: the reference's float32 tensor expressions (diffusion.py:107-118, schedule.py:30-41), evaluated once
        # on the HOST -- the same arithmetic as the CPU reference -- and uploaded as four [steps, B] tables, instead of ~20
        # device micro-kernels per step.
        rows = []
        for t in t_values:
            ts = torch.tensor([t] * B, dtype=torch.float32)
            t_step = 1 / steps
            if schedule is not None:
                t_step = schedule(ts) - schedule(ts - 1 / steps)
                ts = schedule(ts)
            step = t_step if torch.is_tensor(t_step) else torch.full_like(ts, float(t_step))
            rows.append((ts, self.schedule(ts), self.schedule(ts - step), ts - step))
        ts_all, a_t_all, a_prev_all, ts_prev_all = (torch.stack([r[k] for r in rows]).to(torch.float32).contiguous().to(x_T.device)
                                                    for k in range(4))
        its = range(steps)
        if progress:
            from tqdm.auto import tqdm

            its = tqdm(its, total=steps)
        for i in its:
            with torch.no_grad():
                eps = predictor(x_t, ts_all[i])
                last = i + 1 == steps
                if last or noise is None:
                    nz = None
                elif callable(noise):
                    nz = noise(i)
                else:
                    nz = noise[i]
                _native.require_cuda(eps, nz)
                x_t = self._step(x_t, eps, a_t_all[i], a_prev_all[i], ts_prev_all[i], noise=nz, sigma_large=sigma_large,
                                 constrain=constrain, cond_fn=cond_fn, seed=seed, clip_offset=clip_offset, step_index=i,
                                 noise_scale=0.0 if last else 1.0)
        chk = getattr(predictor, "check_status", None)  # range guard of a native predictor: once per sample, not per step
        if chk is None:
            mods = _native_modules(predictor)
            chk = mods[0].check_status if mods else None
        if chk is not None:
            chk()
            # The library's guard sees the tensors that feed a GroupNorm.  The network's output and x_t are fp32 in every mode and cannot
            # overflow a storage type, but a non-finite value can still reach them (an inf / NaN in x_T, in the conditioning or in a
            # cond_fn's gradient): the finished sample is checked here, on the sync check_status() has just paid for.
            if not bool(torch.isfinite(x_t).all()):
                raise _native.NativeError("ddpm_sample: the sample holds non-finite values (a non-finite x_T, conditioning tensor or "
                                          "guidance gradient, or an overflow the range guard reported as a warning)")
        return x_t


def randn_clips(n: int, T: int, device, seed: int, clip_offset: int = 0, stream_id: int = 1) -> torch.Tensor:
    """x_T ~ N(0,1) of shape [n,1,T] from the counter-based generator (keyed by global clip index)."""
    out = torch.empty(n, 1, T, device=device, dtype=torch.float32)
    _native.require_cuda(out)
    with torch.cuda.device(out.device):
        _native.check(_native.lib().vqvs_randn(out.data_ptr(), n, T, int(seed), int(clip_offset), int(stream_id), _native._stream_ptr()))
    return out
