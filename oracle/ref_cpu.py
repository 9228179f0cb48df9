"""
CPU oracle for the DDPM sampling hot path (TEST INFRASTRUCTURE ONLY).

This file is a plain restatement, in stock torch fp32 CPU ops, of the algorithm
the reference implements on this path.  It is *not* part of the product: only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import it, and only as the checker / reported baseline.  The product path
(`vq_voice_swap_amd`) never imports anything from `oracle/` and fails loudly if
its HIP library is missing.

Parity pin: the reference has no tests or golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference
itself, imported in the build container (`oracle/gen_golden.py`), and the
resulting vectors are committed under `tests/golden/`.

Everything is functional: a model is (cfg dict, state dict with the reference's
parameter names).  Reference citations are `file:line` under /root/reference.
"""

from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
State = Dict[str, Tensor]

CHANNEL_MULT = (1, 1, 2, 2, 2, 4, 4, 8, 8)  # vq_voice_swap/models/unet.py:20
MIDDLE_DILATIONS = (4, 8, 16, 32)  # unet.py:21
DEPTH_MULT = 2  # unet.py:22


# --------------------------------------------------------------------------
# topology (unet.py:51-111, 206-222)
# --------------------------------------------------------------------------


def predictor_block_specs(base: int, channel_mult=CHANNEL_MULT, middle_dilations=MIDDLE_DILATIONS, depth_mult: int = DEPTH_MULT) -> Dict[str, List[dict]]:
    """Per-ResBlock (cin, cout, scale, dilation) for down / middle / up lists (any topology UNetPredictor.__init__ takes, unet.py:17-30)."""
    CHANNEL_MULT, MIDDLE_DILATIONS, DEPTH_MULT = tuple(channel_mult), tuple(middle_dilations), depth_mult  # noqa: N806 (shadow the defaults)
    down, middle, up = [], [], []
    stack = [base]
    cur = base
    last = len(CHANNEL_MULT) - 1
    for depth, mult in enumerate(CHANNEL_MULT):
        for _ in range(DEPTH_MULT):
            down.append(dict(cin=cur, cout=mult * base, scale=1.0, dil=2))
            cur = mult * base
            stack.append(cur)
        if depth != last:
            down.append(dict(cin=cur, cout=cur, scale=0.5, dil=2))
            stack.append(cur)
    for d in MIDDLE_DILATIONS:
        middle.append(dict(cin=cur, cout=cur, scale=1.0, dil=d))
    for depth in range(last, -1, -1):
        mult = CHANNEL_MULT[depth]
        for _ in range(DEPTH_MULT + 1):
            skip = stack.pop()
            up.append(dict(cin=cur + skip, cout=mult * base, scale=1.0, dil=2, cat=True))
            cur = mult * base
        if depth:
            up.append(dict(cin=cur, cout=cur, scale=2.0, dil=2, cat=False))
    return dict(down=down, middle=middle, up=up)


def encoder_block_specs(base: int, channel_mult=CHANNEL_MULT, out_dilations=(), depth_mult: int = DEPTH_MULT) -> List[dict]:
    """unet.py:206-220 (any topology UNetEncoder.__init__ takes, unet.py:188-196)."""
    blocks = []
    cur = base
    last = len(channel_mult) - 1
    for depth, mult in enumerate(channel_mult):
        for _ in range(depth_mult):
            blocks.append(dict(cin=cur, cout=mult * base, scale=1.0, dil=2))
            cur = mult * base
        if depth != last:
            blocks.append(dict(cin=cur, cout=cur, scale=0.5, dil=2))
    for d in out_dilations:
        blocks.append(dict(cin=cur, cout=cur, scale=1.0, dil=d))
    return blocks


def gn_groups(ch: int) -> int:
    """unet.py:345-349: start at 32 groups, halve until it divides `ch`."""
    g = 32
    while ch % g:
        g //= 2
    return g


# --------------------------------------------------------------------------
# layers
# --------------------------------------------------------------------------


def group_norm(x: Tensor, sd: State, prefix: str) -> Tensor:
    w = sd[prefix + ".weight"]
    return F.group_norm(x, gn_groups(w.shape[0]), w, sd[prefix + ".bias"], eps=1e-5)


def resize(x: Tensor, scale: float) -> Tensor:
    """unet.py:324-334."""
    if scale == 1.0:
        return x
    if scale < 1.0:
        return F.avg_pool1d(x, int(round(1 / scale)))
    return F.interpolate(x, scale_factor=scale)


def res_block(x: Tensor, sd: State, p: str, spec: dict, emb: Optional[Tensor]) -> Tensor:
    """unet.py:307-316 with the module layout of unet.py:265-305."""
    scale, dil = spec["scale"], spec["dil"]
    h = F.gelu(group_norm(x, sd, p + ".pre_cond.0.0"))
    h = resize(h, scale)
    h = F.conv1d(h, sd[p + ".pre_cond.2.weight"], sd[p + ".pre_cond.2.bias"], padding=1)
    h = group_norm(h, sd, p + ".pre_cond.3")
    if emb is not None:
        ab = F.linear(F.gelu(emb), sd[p + ".cond_layers.1.weight"], sd[p + ".cond_layers.1.bias"])
        cout = spec["cout"]
        a, b = ab[:, :cout, None], ab[:, cout:, None]
        h = h * (a + 1) + b
    conv2 = p + (".post_cond.2" if (p + ".post_cond.2.weight") in sd else ".post_cond.1")
    h = F.conv1d(F.gelu(h), sd[conv2 + ".weight"], sd[conv2 + ".bias"], padding=dil, dilation=dil)
    s = resize(x, scale)
    if (p + ".skip.1.weight") in sd:
        s = F.conv1d(s, sd[p + ".skip.1.weight"], sd[p + ".skip.1.bias"])
    return s + h


def time_embedding(ts: Tensor, sd: State, p: str) -> Tensor:
    """wavegrad.py:359-373."""
    w = sd[p + ".proj.weight"]
    half = w.shape[1] // 2
    freqs = (
        torch.exp(-math.log(100.0 / 0.1) * torch.arange(0, half, dtype=torch.float32) / (half - 1))
        * 100.0
    ).to(ts)
    args = ts[:, None] * freqs[None]
    return F.linear(torch.cat([torch.cos(args), torch.sin(args)], dim=-1), w, sd[p + ".proj.bias"])


def unet_embedding(sd: State, ts: Tensor, labels: Optional[Tensor] = None, prefix: str = "predictor") -> Tensor:
    """The conditioning vector every ResBlock's FiLM reads (unet.py:133-135): time_embed_extra(time_embed(ts)) [+ class_embed]."""
    p = prefix
    emb = time_embedding(ts, sd, p + ".time_embed")
    emb = F.linear(F.gelu(emb), sd[p + ".time_embed_extra.1.weight"], sd[p + ".time_embed_extra.1.bias"])
    if labels is not None:
        emb = emb + F.embedding(labels, sd[p + ".class_embed.weight"])
    return emb


def unet_predictor(
    sd: State,
    base: int,
    x: Tensor,
    ts: Tensor,
    cond: Optional[Tensor] = None,
    labels: Optional[Tensor] = None,
    prefix: str = "predictor",
    probe: Optional[Callable[[str, Tensor], None]] = None,
    topology: Optional[dict] = None,
) -> Tensor:
    """unet.py:118-163.  `probe(name, tensor)` sees every block output (for bisecting); `topology` = dict(channel_mult=,
    middle_dilations=, depth_mult=) for a network other than the default one."""
    p = prefix
    has_labels = (p + ".class_embed.weight") in sd
    has_cond = (p + ".cond_proj.weight") in sd
    assert (labels is None) == (not has_labels), "must provide labels iff class conditional"
    assert (cond is None) == (not has_cond), "must provide cond iff conditional"
    specs = predictor_block_specs(base, **(topology or {}))

    emb = unet_embedding(sd, ts, labels, p)

    h = F.conv1d(x, sd[p + ".in_conv.weight"], sd[p + ".in_conv.bias"], padding=1)
    if cond is not None:
        c = F.conv1d(cond, sd[p + ".cond_proj.weight"], sd[p + ".cond_proj.bias"], padding=1)
        h = h + F.interpolate(c, h.shape[-1])
    if probe:
        probe("in_conv", h)
    skips = [h]
    for i, spec in enumerate(specs["down"]):
        h = res_block(h, sd, f"{p}.down_blocks.{i}", spec, emb)
        skips.append(h)
        if probe:
            probe(f"down_blocks.{i}", h)
    for i, spec in enumerate(specs["middle"]):
        h = res_block(h, sd, f"{p}.middle_blocks.{i}", spec, emb)
        if probe:
            probe(f"middle_blocks.{i}", h)
    for i, spec in enumerate(specs["up"]):
        if spec["cat"]:
            h = torch.cat([h, skips.pop()], dim=1)
        h = res_block(h, sd, f"{p}.up_blocks.{i}", spec, emb)
        if probe:
            probe(f"up_blocks.{i}", h)
    h = F.gelu(group_norm(h, sd, p + ".out.0.0"))
    return F.conv1d(h, sd[p + ".out.1.weight"], sd[p + ".out.1.bias"], padding=1)


def unet_encoder(sd: State, base: int, x: Tensor, prefix: str = "encoder", topology: Optional[dict] = None) -> Tensor:
    """unet.py:229-241; `topology` = dict(channel_mult=, out_dilations=, depth_mult=) for a network other than the default one."""
    p = prefix
    h = F.conv1d(x, sd[p + ".in_conv.weight"], sd[p + ".in_conv.bias"], padding=1)
    for i, spec in enumerate(encoder_block_specs(base, **(topology or {}))):
        h = res_block(h, sd, f"{p}.blocks.{i}", spec, None)
    h = F.gelu(group_norm(h, sd, p + ".out.0.0"))
    return F.conv1d(h, sd[p + ".out.1.weight"], sd[p + ".out.1.bias"], padding=1)


def classifier(sd: State, base: int, x: Tensor, ts: Tensor, prefix: str = "", topology: Optional[dict] = None) -> Tensor:
    """Classifier.forward (models/classifier.py:31-36, 111-121, 153-158, 170-191): logits [N, num_labels].  `topology` =
    dict(channel_mult=, depth_mult=) for a stem other than the default one (classifier.py:52-58; output_mult is read off the weights)."""
    CHANNEL_MULT, DEPTH_MULT = (tuple(topology["channel_mult"]), topology["depth_mult"]) if topology else (globals()["CHANNEL_MULT"], globals()["DEPTH_MULT"])  # noqa: N806
    p = prefix + "stem"
    emb = time_embedding(ts, sd, p + ".time_embed")
    emb = F.linear(F.gelu(emb), sd[p + ".time_embed_extra.1.weight"], sd[p + ".time_embed_extra.1.bias"])
    h = F.conv1d(x, sd[p + ".in_conv.weight"], sd[p + ".in_conv.bias"], padding=1)
    cur, i = base, 0
    for mult in CHANNEL_MULT:
        for _ in range(DEPTH_MULT):
            h = res_block(h, sd, f"{p}.blocks.{i}", dict(cin=cur, cout=mult * base, scale=1.0, dil=2), emb)
            cur = mult * base
            i += 1
        h = res_block(h, sd, f"{p}.blocks.{i}", dict(cin=cur, cout=cur, scale=0.5, dil=2), emb)
        i += 1
    h = F.gelu(group_norm(h, sd, p + ".out.0.0"))
    n, c, t = h.shape
    h = torch.cat([torch.zeros_like(h[..., :1]), h], dim=-1)
    qkv = F.conv1d(h, sd[p + ".out.1.qkv_proj.weight"], sd[p + ".out.1.qkv_proj.bias"])
    heads = c // min(c, 64)
    ch = c // heads
    q, k, v = qkv.chunk(3, dim=1)
    scale = 1 / math.sqrt(math.sqrt(ch))
    w = torch.einsum("bct,bcs->bts", (q * scale).view(n * heads, ch, t + 1), (k * scale).view(n * heads, ch, t + 1))
    w = torch.softmax(w, dim=-1)
    a = torch.einsum("bts,bcs->bct", w, v.reshape(n * heads, ch, t + 1)).reshape(n, -1, t + 1)
    feat = F.conv1d(a, sd[p + ".out.1.c_proj.weight"], sd[p + ".out.1.c_proj.bias"])[..., 0]
    return F.linear(F.gelu(feat), sd[prefix + "out.1.weight"], sd[prefix + "out.1.bias"])


def classifier_cond_fn(sd: State, base: int, labels: Tensor, scale: float = 1.0, topology: Optional[dict] = None) -> Callable:
    """sample_diffusion.py:34-42."""

    def cond_fn(x, ts):
        with torch.enable_grad():
            xg = x.detach().clone().requires_grad_()
            logp = F.log_softmax(classifier(sd, base, xg, ts, topology=topology), dim=-1)
            grads = torch.autograd.grad(logp[range(len(xg)), labels].sum(), xg)[0]
        return grads.detach() * scale

    return cond_fn


def encoder_predictor(sd: State, base: int, x: Tensor, ts: Tensor, rate: int, prefix: str = "") -> Tensor:
    """EncoderPredictor.forward (models/encoder_predictor.py:43-58): [N,1,T] -> logits [N, num_latents, T // rate]."""
    h = unet_predictor(sd, base, x, ts, prefix=prefix + "unet")
    h = F.interpolate(h, size=(h.shape[-1] // rate,), mode="nearest")
    return F.conv1d(h, sd[prefix + "out.weight"], sd[prefix + "out.bias"])


def encoder_predictor_cond_fn(sd: State, base: int, rate: int, targets: Tensor, scale: float = 1.0) -> Callable:
    """The cond_fn of VQVAE.decode(enc_pred=...) (vq_vae.py:125-130 with encoder_predictor.py:60-64)."""

    def cond_fn(x, ts):
        with torch.enable_grad():
            xg = x.detach().clone().requires_grad_(True)
            losses = F.cross_entropy(encoder_predictor(sd, base, xg, ts, rate), targets, reduction="none").mean(-1)
            losses = losses * targets.shape[-1]
            grads = torch.autograd.grad(losses.sum(), xg)[0]
        return grads * scale * -1

    return cond_fn


# --------------------------------------------------------------------------
# VQ (vq.py:98-143, 199-243)
# --------------------------------------------------------------------------


def vq_distances(dictionary: Tensor, rows: Tensor) -> Tensor:
    """vq.py:199-221, same evaluation order: ((-2*dots) + |e|^2) + |x|^2."""
    dict_norms = torch.sum(torch.pow(dictionary, 2), dim=-1)
    row_norms = torch.sum(torch.pow(rows, 2), dim=-1)
    lhs = dictionary[None].expand(rows.shape[0], *dictionary.shape)
    dots = torch.bmm(lhs, rows[:, :, None])[..., 0]
    return -2 * dots + dict_norms + row_norms[..., None]


def vq_encode(dictionary: Tensor, z: Tensor) -> Tensor:
    """[N,C,T] -> int64 [N,T] code indices (vq.py:127-131, 224-243)."""
    n, c, t = z.shape
    rows = z.permute(0, 2, 1).reshape(-1, c)
    return torch.argmin(vq_distances(dictionary, rows), dim=-1).reshape(n, t)


def vq_embed(dictionary: Tensor, idxs: Tensor) -> Tensor:
    """vq.py:98-110: int [N,T] -> [N,C,T]."""
    return F.embedding(idxs, dictionary).permute(0, 2, 1).contiguous()


# --------------------------------------------------------------------------
# diffusion (diffusion/schedule.py:15-41, diffusion/diffusion.py:28-133)
# --------------------------------------------------------------------------


def schedule_alpha(name: str, t: Tensor) -> Tensor:
    if name == "exp":
        return torch.exp(-(-math.log(1e-5)) * (t ** 2))
    if name == "cos":
        return torch.cos(t * math.pi / 2) ** 2
    raise ValueError(f"unknown schedule: {name}")


def _bcast(v: Tensor, like: Tensor) -> Tensor:
    while v.dim() < like.dim():
        v = v[:, None]
    return v.to(like) + torch.zeros_like(like)  # diffusion.py:154-157


def ddpm_previous(
    schedule: str,
    x_t: Tensor,
    ts: Tensor,
    step,
    eps: Tensor,
    noise: Tensor,
    sigma_large: bool = False,
    constrain: bool = False,
    cond_fn: Optional[Callable] = None,
) -> Tensor:
    """diffusion.py:48-90 (noise is always explicit here)."""
    a_t = _bcast(schedule_alpha(schedule, ts), x_t)
    a_prev = _bcast(schedule_alpha(schedule, ts - step), x_t)
    alphas = a_t / a_prev
    betas = 1 - alphas

    def eps_to_prev(e):
        return alphas.rsqrt() * (x_t - betas * (1 - a_t).rsqrt() * e)

    def prev_to_eps(prev):
        return (-prev * alphas.sqrt() + x_t) * (1 - a_t).sqrt() / betas

    sigmas = betas if sigma_large else betas * (1 - a_prev) / (1 - a_t)
    if cond_fn is not None:
        mean = eps_to_prev(eps)
        mean = mean + sigmas * cond_fn(mean, ts - step)
        eps = prev_to_eps(mean)
    if constrain:
        x0 = (x_t - (1 - a_t).sqrt() * eps) * a_t.rsqrt()
        x0 = (x0 - x0.mean(dim=-1, keepdim=True)).clamp(-1, 1)
        eps = (x_t - x0 * a_t.sqrt()) * (1 - a_t).rsqrt()
    return eps_to_prev(eps) + sigmas.sqrt() * noise


def ddpm_sample(
    schedule: str,
    x_T: Tensor,
    predictor: Callable[[Tensor, Tensor], Tensor],
    steps: int,
    noises: List[Tensor],
    sigma_large: bool = False,
    constrain: bool = False,
    cond_fn: Optional[Callable] = None,
    t_map: Optional[Callable[[Tensor], Tensor]] = None,
    trace: Optional[List[Tensor]] = None,
) -> Tensor:
    """diffusion.py:92-133.  `noises[i]` is the N(0,1) draw of iteration i; the last
    iteration uses zeros regardless (diffusion.py:127)."""
    x_t = x_T
    t_list = [(i + 1) / steps for i in range(steps)]
    for i, t in enumerate(t_list[::-1]):
        ts = torch.tensor([t] * x_T.shape[0]).to(x_T)
        t_step = 1 / steps
        if t_map is not None:
            t_step = t_map(ts) - t_map(ts - 1 / steps)
            ts = t_map(ts)
        with torch.no_grad():
            eps = predictor(x_t, ts)
            noise = torch.zeros_like(x_T) if i + 1 == steps else noises[i]
            x_t = ddpm_previous(
                schedule, x_t, ts, t_step, eps, noise,
                sigma_large=sigma_large, constrain=constrain, cond_fn=cond_fn,
            )
        if trace is not None:
            trace.append(x_t)
    return x_t


# --------------------------------------------------------------------------
# VQ-VAE facade (vq_vae.py:82-145)
# --------------------------------------------------------------------------


def vqvae_encode(sd: State, base: int, inputs: Tensor) -> Tensor:
    with torch.no_grad():
        return vq_encode(sd["vq.dictionary"], unet_encoder(sd, base, inputs))


def vqvae_decode(
    sd: State,
    base: int,
    schedule: str,
    codes: Tensor,
    labels: Optional[Tensor],
    steps: int,
    x_T: Tensor,
    noises: List[Tensor],
    constrain: bool = False,
    cond_fn: Optional[Callable] = None,
) -> Tensor:
    cond = vq_embed(sd["vq.dictionary"], codes) if codes.dim() == 2 else codes
    return ddpm_sample(
        schedule, x_T,
        lambda xs, ts: unet_predictor(sd, base, xs, ts, cond=cond, labels=labels),
        steps, noises, constrain=constrain, cond_fn=cond_fn,
    )


def vqvae_decode_uncond_guidance(
    sd: State,
    base: int,
    schedule: str,
    codes: Tensor,
    labels: Tensor,
    steps: int,
    x_T: Tensor,
    noises: List[Tensor],
    constrain: bool = False,
    label_scale: float = 0.0,
    vq_scale: float = 0.0,
) -> Tensor:
    """VQVAE.decode_uncond_guidance (vq_vae.py:147-220) with both guidance scales on (the only configuration in which
    the reference's always-tripled batch lines up, vq_vae.py:188-203): rows [conditional | codes zeroed | label 0],
    labels offset by one (label 0 is the unconditional label), pred = base + s_vq (base - no_vq) + s_label (base - no_label)."""
    assert vq_scale and label_scale and labels is not None
    cond = vq_embed(sd["vq.dictionary"], codes) if codes.dim() == 2 else codes
    n = cond.shape[0]
    cond3 = torch.cat([cond, torch.zeros_like(cond), cond], dim=0)
    lab3 = torch.cat([labels + 1, labels + 1, torch.zeros_like(labels)], dim=0)

    def pred_fn(xs, ts):
        outs = unet_predictor(sd, base, torch.cat([xs] * 3, dim=0), torch.cat([ts] * 3, dim=0), cond=cond3, labels=lab3)
        b = outs[:n]
        return b + vq_scale * (b - outs[n:2 * n]) + label_scale * (b - outs[2 * n:])

    return ddpm_sample(schedule, x_T, pred_fn, steps, noises, constrain=constrain)


# --------------------------------------------------------------------------
# ConvMFCCEncoder (models/conv_encoder.py:14-133).  PARITY UNPINNED for the MFCC front end: the reference builds it from
# `torchaudio.transforms.MFCC` (conv_encoder.py:42-58), and torchaudio is not installed in the build container (SURVEY.md
# 8c), so the front end below restates torchaudio's published algorithm -- transforms.MFCC / MelSpectrogram / Spectrogram /
# MelScale / AmplitudeToDB and functional.create_dct / melscale_fbanks / amplitude_to_DB (torchaudio 0.8 ... 2.x, same
# arithmetic throughout) -- and cannot be checked against the reference's own output here.  The three constant tensors of
# the transform (Hann window, mel filter bank, DCT matrix) are persistent buffers of the reference module, so a reference
# checkpoint carries them ("encoder.mfcc.*"); both this oracle and the HIP path read them from the state dict.  The
# convolution stack after the front end, `deltas` and `invert_ulaw` ARE the reference's own code and are restated 1:1.
# --------------------------------------------------------------------------


def mfcc_config(version: int = 1, input_rate: int = 16000, mfcc_rate: int = 100) -> dict:
    """conv_encoder.py:44-58."""
    hop = input_rate // mfcc_rate
    n_fft = round(400 * input_rate / 16000) if version == 2 else hop * 2
    return dict(n_fft=n_fft, hop=hop, n_mels=40 if version == 1 else 80, n_mfcc=13, log_mels=version == 1,
                normalized=version == 2, sample_rate=input_rate)


def mfcc_buffers(cfg: dict) -> State:
    """The transform's constant tensors under the reference's buffer names.
    torchaudio.functional.melscale_fbanks(n_freqs, 0, sr/2, n_mels, sr, norm=None, mel_scale="htk") and
    create_dct(n_mfcc, n_mels, norm="ortho"); window = torch.hann_window(n_fft) (periodic)."""
    n_fft, n_mels, n_mfcc, sr = cfg["n_fft"], cfg["n_mels"], cfg["n_mfcc"], cfg["sample_rate"]
    n_freqs = n_fft // 2 + 1
    all_freqs = torch.linspace(0, sr // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + 0.0 / 700.0)
    m_max = 2595.0 * math.log10(1.0 + float(sr // 2) / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = torch.max(torch.zeros(1), torch.min(down, up))
    n = torch.arange(float(n_mels))
    k = torch.arange(float(n_mfcc)).unsqueeze(1)
    dct = torch.cos(math.pi / float(n_mels) * (n + 0.5) * k)
    dct[0] *= 1.0 / math.sqrt(2.0)
    dct *= math.sqrt(2.0 / float(n_mels))
    return {
        "mfcc.MelSpectrogram.spectrogram.window": torch.hann_window(n_fft),
        "mfcc.MelSpectrogram.mel_scale.fb": fb,
        "mfcc.dct_mat": dct.t().contiguous(),
    }


def invert_ulaw(x: Tensor, mu: float = 255.0) -> Tensor:
    """conv_encoder.py:132-133."""
    return x.sign() * (1 / mu) * ((1 + mu) ** x.abs() - 1)


def deltas(seq: Tensor) -> Tensor:
    """conv_encoder.py:123-129."""
    right_shifted = torch.cat([seq[..., :1], seq[..., :-1]], dim=-1)
    left_shifted = torch.cat([seq[..., 1:], seq[..., -1:]], dim=-1)
    d1 = right_shifted - seq
    d2 = seq - left_shifted
    return (d1 + d2) / 2


def mfcc_transform(wave: Tensor, sd: State, cfg: dict, prefix: str = "") -> Tensor:
    """torchaudio.transforms.MFCC.forward on a [N, T] waveform -> [N, n_mfcc, T // hop + 1]."""
    window = sd[prefix + "mfcc.MelSpectrogram.spectrogram.window"]
    fb = sd[prefix + "mfcc.MelSpectrogram.mel_scale.fb"]
    dct = sd[prefix + "mfcc.dct_mat"]
    n_fft, hop = cfg["n_fft"], cfg["hop"]
    # functional.spectrogram: torch.stft(center=True, pad_mode="reflect", onesided, normalized=False), then the "window"
    # normalisation (spec /= sqrt(sum window^2)) when `normalized`, then |.|^2 (power = 2)
    spec = torch.stft(wave, n_fft, hop_length=hop, win_length=n_fft, window=window, center=True, pad_mode="reflect",
                      normalized=False, onesided=True, return_complex=True)
    if cfg["normalized"]:
        spec = spec / window.pow(2.0).sum().sqrt()
    power = spec.abs().pow(2.0)                                            # [N, n_freqs, frames]
    mel = torch.matmul(power.transpose(-1, -2), fb).transpose(-1, -2)      # MelScale.forward
    if cfg["log_mels"]:
        mel = torch.log(mel + 1e-6)
    else:
        # AmplitudeToDB("power", top_db=80): 10 log10(clamp(x, 1e-10)); for a 3-D input the batch axis is taken for the
        # channel axis (functional.amplitude_to_DB reshapes to [-1, shape[-3], freq, time]) -> ONE maximum over the whole batch
        mel = 10.0 * torch.log10(torch.clamp(mel, min=1e-10))
        mel = torch.max(mel, mel.max() - 80.0)
    return torch.matmul(mel.transpose(-1, -2), dct).transpose(-1, -2)


def conv_mfcc_encoder(sd: State, x: Tensor, version: int = 1, input_ulaw: bool = True, prefix: str = "encoder.",
                      probe: Optional[Callable[[str, Tensor], None]] = None, mfcc_override: Optional[Tensor] = None) -> Tensor:
    """ConvMFCCEncoder.forward (conv_encoder.py:90-110): [N,1,T] -> [N, out_channels, (T // hop + 1 - 2) // 2 + 1]."""
    p = prefix
    cfg = mfcc_config(version)
    assert x.shape[1] == 1, "input must only have one channel"
    if input_ulaw:
        x = invert_ulaw(x)
    # (mfcc_override: the [N, 13, frames] tensor the transform would return -- fixture F12 pins everything AROUND the transform)
    h = mfcc_transform(x[:, 0, :], sd, cfg, p) if mfcc_override is None else mfcc_override
    deriv = deltas(h)
    accel = deltas(deriv)
    h = torch.cat([h, deriv, accel], dim=1)
    if probe:
        probe("features", h)

    def res_conv(h, i, pad):
        return h + F.gelu(F.conv1d(h, sd[f"{p}blocks.{i}.conv.weight"], sd[f"{p}blocks.{i}.conv.bias"], padding=pad))

    h = F.gelu(F.conv1d(h, sd[p + "blocks.0.0.weight"], sd[p + "blocks.0.0.bias"], padding=1))
    h = res_conv(h, 1, 1)
    h = F.gelu(F.conv1d(h, sd[p + "blocks.2.0.weight"], sd[p + "blocks.2.0.bias"], stride=2, padding=1))
    for i in (3, 4):
        h = res_conv(h, i, 1)
    for i in (5, 6, 7, 8):
        h = res_conv(h, i, 0)
    return F.conv1d(h, sd[p + "blocks.9.weight"], sd[p + "blocks.9.bias"])
