"""
Generate the golden fixtures under tests/golden/ by importing the *reference*
(read-only, /root/reference) in the build container, and check the CPU oracle
(oracle/ref_cpu.py) against it on the same inputs.

    PYTHONPATH=/root/reference:/root/repo python oracle/gen_golden.py

The reference never travels to the GPU box; only the .npz vectors written here
do.  Fixtures are data only (inputs, expected outputs, seeds, checksums).
"""

from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"
# the repo also ships an import shim called `vq_voice_swap`; the REFERENCE must win here
sys.path = [REFERENCE] + [p for p in sys.path if os.path.abspath(p or ".") not in (ROOT, REFERENCE)] + [ROOT]

from oracle import ref_cpu  # noqa: E402
from vq_voice_swap_amd.det_init import det_init_  # noqa: E402

from vq_voice_swap.diffusion_model import DiffusionModel  # noqa: E402  (reference)
from vq_voice_swap.models.unet import ResBlock  # noqa: E402  (reference)
from vq_voice_swap.vq_vae import VQVAE  # noqa: E402  (reference)

import vq_voice_swap as _ref_pkg  # noqa: E402

assert os.path.abspath(_ref_pkg.__file__).startswith(REFERENCE), f"not the reference: {_ref_pkg.__file__}"

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)


def seeded(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


def state_of(model):
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    return sd


def det_model(model):
    det_init_(model.state_dict().items())
    model.eval()
    return model


def check(name, ref, ours, tol=1e-6):
    err = (ref - ours).abs().max().item()
    scale = ref.abs().max().item()
    print(f"  oracle vs reference [{name}]: max|diff|={err:.3e} (max|ref|={scale:.3e})")
    assert err <= tol * max(1.0, scale), f"oracle diverges from reference on {name}"


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrays.items()})
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


def probe_stats(t):
    flat = t.flatten()
    idx = torch.linspace(0, flat.numel() - 1, 16).long()
    return np.concatenate([[t.mean().item(), t.pow(2).mean().sqrt().item()], flat[idx].numpy()]).astype(np.float32)


# ---------------------------------------------------------------- F1 ResBlocks
def gen_resblocks():
    cases = [
        dict(name="same32", cin=32, cout=32, scale=1.0, dil=2, emb=128, L=256),
        dict(name="widen32_64", cin=32, cout=64, scale=1.0, dil=2, emb=128, L=256),
        dict(name="down64", cin=64, cout=64, scale=0.5, dil=2, emb=128, L=256),
        dict(name="up64", cin=64, cout=64, scale=2.0, dil=2, emb=128, L=128),
        dict(name="cat96_32", cin=96, cout=32, scale=1.0, dil=2, emb=128, L=256),
        dict(name="mid_dil32", cin=64, cout=64, scale=1.0, dil=32, emb=128, L=250),
        dict(name="noemb_enc", cin=32, cout=64, scale=1.0, dil=2, emb=None, L=256),
    ]
    out = {}
    for i, c in enumerate(cases):
        blk = ResBlock(channels=c["cin"], emb_channels=c["emb"],
                       out_channels=c["cout"] if c["cout"] != c["cin"] else None,
                       scale_factor=c["scale"], dilation=c["dil"])
        prefix = "blk." + c["name"]
        det_init_((prefix + "." + k, v) for k, v in blk.state_dict().items())
        blk.eval()
        x = seeded((2, c["cin"], c["L"]), 100 + i)
        emb = seeded((2, c["emb"]), 200 + i) if c["emb"] else None
        with torch.no_grad():
            y = blk(x, emb) if emb is not None else blk(x)
        sd = {prefix + "." + k: v for k, v in blk.state_dict().items()}
        spec = dict(cin=c["cin"], cout=c["cout"], scale=c["scale"], dil=c["dil"])
        y2 = ref_cpu.res_block(x, sd, prefix, spec, emb)
        check("resblock " + c["name"], y, y2)
        out[c["name"] + ".x"] = x
        if emb is not None:
            out[c["name"] + ".emb"] = emb
        out[c["name"] + ".y"] = y
        out[c["name"] + ".spec"] = np.array([c["cin"], c["cout"], c["scale"], c["dil"], c["emb"] or 0, c["L"]], dtype=np.float64)
    save("f1_resblocks", **out)


# ---------------------------------------------------------------- F3 whole unet32
def gen_unet32():
    model = det_model(DiffusionModel("unet", 32))
    sd = state_of(model)
    x = seeded((2, 1, 64000), 1)
    ts = torch.tensor([0.3, 0.9])
    probes = {}

    def hook_for(name):
        def fn(mod, inp, outp):
            probes[name] = probe_stats(outp)
        return fn

    hs = [model.predictor.in_conv.register_forward_hook(hook_for("in_conv"))]
    for grp in ("down_blocks", "middle_blocks", "up_blocks"):
        for i, b in enumerate(getattr(model.predictor, grp)):
            hs.append(b.register_forward_hook(hook_for(f"{grp}.{i}")))
    with torch.no_grad():
        eps = model.predictor(x, ts)
    for h in hs:
        h.remove()
    oprobes = {}
    eps2 = ref_cpu.unet_predictor(sd, 32, x, ts, probe=lambda n, t: oprobes.__setitem__(n, probe_stats(t)))
    check("unet32 eps", eps, eps2)
    for k in probes:
        if k == "in_conv" and "in_conv" not in oprobes:
            continue
        assert np.allclose(probes[k], oprobes[k], atol=1e-5, rtol=1e-5), k
    names = sorted(probes.keys())
    save("f3_unet32_forward", x_seed=1, ts=ts, eps=eps,
         probe_names=np.array(names), probe_vals=np.stack([probes[n] for n in names]))
    print(f"  eps rms={eps.pow(2).mean().sqrt().item():.4f}")
    return model, sd


# ---------------------------------------------------------------- F5 ddpm_previous
def gen_ddpm_previous(model):
    diff = model.diffusion
    cases = [(1.0, 0.1), (0.5, 0.02), (0.02, 0.02), (0.37, 0.01), (0.9, 0.25)]
    out = {}
    for i, (t, step) in enumerate(cases):
        x = seeded((2, 1, 4096), 300 + i)
        eps = seeded((2, 1, 4096), 400 + i)
        noise = seeded((2, 1, 4096), 500 + i)
        ts = torch.tensor([t, t])
        for mode, kw in (("plain", {}), ("sigma_large", dict(sigma_large=True)), ("constrain", dict(constrain=True))):
            y = diff.ddpm_previous(x, ts, step, eps, noise=noise, **kw)
            y2 = ref_cpu.ddpm_previous("exp", x, ts, step, eps, noise, **kw)
            check(f"ddpm_previous t={t} {mode}", y, y2)
            out[f"c{i}.{mode}"] = y
        out[f"c{i}.x"], out[f"c{i}.eps"], out[f"c{i}.noise"] = x, eps, noise
        out[f"c{i}.t_step"] = np.array([t, step])
    # per-row timesteps + a tensor step (sample-time schedule form, diffusion.py:116-118)
    x, eps, noise = seeded((2, 1, 4096), 390), seeded((2, 1, 4096), 490), seeded((2, 1, 4096), 590)
    ts = torch.tensor([0.81, 0.25])
    step = torch.tensor([0.036, 0.019])
    y = diff.ddpm_previous(x, ts, step, eps, noise=noise, constrain=True)
    y2 = ref_cpu.ddpm_previous("exp", x, ts, step, eps, noise, constrain=True)
    check("ddpm_previous per-row", y, y2)
    out.update({"row.x": x, "row.eps": eps, "row.noise": noise, "row.ts": ts, "row.step": step, "row.constrain": y})
    save("f5_ddpm_previous", **out)


# ---------------------------------------------------------------- F6 end-to-end sampler
def run_ref_sampler(model, x_T, steps, noise_seed, constrain, t_map=None):
    """Drive the reference's ddpm_sample with explicit per-step noise by patching randn_like."""
    gen = torch.Generator().manual_seed(noise_seed)
    noises = [torch.randn(x_T.shape, generator=gen) for _ in range(steps)]
    it = iter(noises)
    import vq_voice_swap.diffusion.diffusion as dmod
    orig = torch.randn_like
    calls = []

    def fake_randn_like(t, *a, **k):
        calls.append(1)
        return next(it)

    dmod.torch.randn_like = fake_randn_like
    try:
        x0 = model.diffusion.ddpm_sample(x_T, model.predictor, steps, constrain=constrain, schedule=t_map)
    finally:
        dmod.torch.randn_like = orig
    assert len(calls) == steps - 1  # last iteration uses zeros (diffusion.py:127)
    return x0, noises


def gen_sampler(model, sd):
    x_T = seeded((2, 1, 64000), 7)
    out = {}
    for tag, steps, constrain, tmap in (("s10_plain", 10, False, None), ("s10_constrain", 10, True, None),
                                        ("s50_sq_constrain", 50, True, (lambda t: t ** 2))):
        x0, noises = run_ref_sampler(model, x_T, steps, 11, constrain, tmap)
        trace = []
        x0b = ref_cpu.ddpm_sample("exp", x_T, lambda a, b: ref_cpu.unet_predictor(sd, 32, a, b), steps, noises,
                                  constrain=constrain, t_map=tmap, trace=trace)
        check("sampler " + tag, x0, x0b, tol=1e-5)
        out[tag + ".x0"] = x0
        out[tag + ".rms_trace"] = np.array([t.pow(2).mean().sqrt().item() for t in trace], dtype=np.float32)
        out[tag + ".noise_checksum"] = np.array([n.double().sum().item() for n in noises])
        print(f"  {tag}: x0 rms={x0.pow(2).mean().sqrt().item():.4f}")
    save("f6_sampler_unet32", x_T_seed=7, noise_seed=11, **out)


# ---------------------------------------------------------------- F4/F7/F8 VQ-VAE
def gen_vqvae():
    model = det_model(VQVAE(base_channels=32, pred_name="unet", num_labels=5))
    with torch.no_grad():
        model.vq.dictionary.copy_(seeded(model.vq.dictionary.shape, 77, 0.35))
    sd = state_of(model)
    # F7 encoder + VQ at full length
    wav = seeded((2, 1, 64000), 21, 0.1).clamp(-1, 1)
    with torch.no_grad():
        z = model.encoder(wav)
        codes = model.encode(wav)
    z2 = ref_cpu.unet_encoder(sd, 32, wav)
    check("encoder z", z, z2)
    codes2 = ref_cpu.vq_encode(sd["vq.dictionary"], z2)
    assert torch.equal(codes, codes2), "oracle VQ codes differ from reference"
    d = ref_cpu.vq_distances(sd["vq.dictionary"], z.permute(0, 2, 1).reshape(-1, z.shape[1]))
    top2 = torch.topk(d, 2, dim=-1, largest=False).values
    gap = (top2[:, 1] - top2[:, 0]).reshape(2, -1)
    print(f"  z rms={z.pow(2).mean().sqrt().item():.4f}; top-2 gap min={gap.min().item():.3e} median={gap.median().item():.3e}; distinct codes={codes.unique().numel()}")
    # margin-guaranteed VQ set: dictionary rows + small noise
    idx_m = torch.randint(0, 512, (2, 250), generator=torch.Generator().manual_seed(31))
    zm = ref_cpu.vq_embed(sd["vq.dictionary"], idx_m) + 1e-3 * seeded((2, 512, 250), 32)
    with torch.no_grad():
        codes_m = model.vq(zm)["idxs"]
    assert torch.equal(codes_m, idx_m)
    assert torch.equal(ref_cpu.vq_encode(sd["vq.dictionary"], zm), idx_m)
    save("f7_encoder_vq32", wav_seed=21, z=z.half(), z_rms=z.pow(2).mean().sqrt().item(), codes=codes, gap=gap,
         margin_idx=idx_m, margin_noise_seed=32)
    # F4 conditional predictor forward (cond + labels), short clip
    x = seeded((2, 1, 4096), 41)
    ts = torch.tensor([0.7, 0.15])
    cond = ref_cpu.vq_embed(sd["vq.dictionary"], codes[:, :16])
    labels = torch.tensor([3, 0])
    with torch.no_grad():
        eps = model.predictor(x, ts, cond=cond, labels=labels)
    eps2 = ref_cpu.unet_predictor(sd, 32, x, ts, cond=cond, labels=labels)
    check("vqvae predictor cond+labels", eps, eps2)
    save("f4_cond_forward", x=x, ts=ts, codes16=codes[:, :16], labels=labels, eps=eps)
    # F8 decode, 5 steps, constrain
    import vq_voice_swap.diffusion.diffusion as dmod
    import vq_voice_swap.vq_vae as vmod
    x_T = seeded((2, 1, 4096), 51)
    gen = torch.Generator().manual_seed(52)
    noises = [torch.randn(x_T.shape, generator=gen) for _ in range(5)]
    it = iter(noises)
    orig_rl, orig_r = torch.randn_like, torch.randn
    dmod.torch.randn_like = lambda t, *a, **k: next(it)
    vmod.torch.randn = lambda *a, **k: x_T.clone()
    try:
        with torch.no_grad():
            dec = model.decode(codes[:, :16], labels, steps=5, constrain=True)
    finally:
        dmod.torch.randn_like = orig_rl
        vmod.torch.randn = orig_r
    dec2 = ref_cpu.vqvae_decode(sd, 32, "exp", codes[:, :16], labels, 5, x_T, noises, constrain=True)
    check("vqvae decode 5 steps", dec, dec2, tol=1e-5)
    save("f8_vqvae_decode", x_T_seed=51, noise_seed=52, codes16=codes[:, :16], labels=labels, x0=dec)
    # F8b: the same decode at BASELINE config 4's step count (50): the 5-step fixture above is dominated by the first reverse
    # step's 1 / sqrt(alpha_bar(1)) amplification, which 50 steps average out -- the 1e-3 gate of the 2-byte modes is claimed here
    x_T = seeded((2, 1, 4096), 53)
    gen = torch.Generator().manual_seed(54)
    noises = [torch.randn(x_T.shape, generator=gen) for _ in range(50)]
    it = iter(noises)
    dmod.torch.randn_like = lambda t, *a, **k: next(it)
    vmod.torch.randn = lambda *a, **k: x_T.clone()
    try:
        with torch.no_grad():
            dec = model.decode(codes[:, :16], labels, steps=50, constrain=True)
    finally:
        dmod.torch.randn_like = orig_rl
        vmod.torch.randn = orig_r
    dec2 = ref_cpu.vqvae_decode(sd, 32, "exp", codes[:, :16], labels, 50, x_T, noises, constrain=True)
    check("vqvae decode 50 steps", dec, dec2, tol=1e-5)
    save("f8b_vqvae_decode50", x_T_seed=53, noise_seed=54, codes16=codes[:, :16], labels=labels, x0=dec)


# ---------------------------------------------------------------- F9 classifier (config 5)
def gen_classifier():
    import torch.nn.functional as F
    from vq_voice_swap.models import Classifier  # reference

    clf = det_model(Classifier(num_labels=7, base_channels=32))
    sd = state_of(clf)
    x = seeded((2, 1, 64000), 61)
    ts = torch.tensor([0.25, 0.8])
    labels = torch.tensor([4, 1])
    xg = x.clone().requires_grad_()
    logits = clf(xg, ts)
    grad = torch.autograd.grad(F.log_softmax(logits, dim=-1)[range(2), labels].sum(), xg)[0]
    logits2 = ref_cpu.classifier(sd, 32, x, ts)
    check("classifier logits", logits.detach(), logits2, tol=1e-6)
    grad2 = ref_cpu.classifier_cond_fn(sd, 32, labels)(x, ts)
    check("classifier grad", grad, grad2, tol=1e-6)
    print(f"  logits rms={logits.pow(2).mean().sqrt().item():.4f} grad rms={grad.pow(2).mean().sqrt().item():.3e}")
    save("f9_classifier32", x_seed=61, ts=ts, labels=labels, logits=logits.detach(), grad=grad)


# ---------------------------------------------------------------- F10 encoder predictor (vq_vae.py:125-130 guidance)
def gen_encpred():
    import torch.nn.functional as F
    from vq_voice_swap.models import EncoderPredictor  # reference

    ep = det_model(EncoderPredictor(base_channels=32, downsample_rate=256, num_latents=96, bottleneck_dim=64))
    sd = state_of(ep)
    x = seeded((2, 1, 16384), 71)
    ts = torch.tensor([0.3, 0.85])
    targets = torch.randint(0, 96, (2, 64), generator=torch.Generator().manual_seed(72))
    xg = x.clone().requires_grad_()
    logits = ep(xg, ts)
    losses = ep.losses(xg, ts, targets) * targets.shape[-1]       # vq_vae.py:128
    grad = torch.autograd.grad(losses.sum(), xg)[0] * 1.0 * -1      # vq_vae.py:129-130
    logits2 = ref_cpu.encoder_predictor(sd, 32, x, ts, 256)
    check("encoder predictor logits", logits.detach(), logits2, tol=1e-6)
    grad2 = ref_cpu.encoder_predictor_cond_fn(sd, 32, 256, targets)(x, ts)
    check("encoder predictor grad", grad, grad2, tol=1e-6)
    print(f"  logits rms={logits.pow(2).mean().sqrt().item():.4f} grad rms={grad.pow(2).mean().sqrt().item():.3e}")
    save("f10_encpred32", x_seed=71, ts=ts, targets=targets, logits=logits.detach(), grad=grad)


# ---------------------------------------------------------------- F2 timestep embedding (wavegrad.py:359-373, unet.py:133-135)
def gen_time_embed():
    out = {}
    ts = torch.tensor([0.02, 0.5, 1.0])
    for base in (32, 64):
        model = det_model(DiffusionModel("unet", base, num_labels=6))
        sd = state_of(model)
        labels = torch.tensor([5, 0, 3])
        with torch.no_grad():
            t_emb = model.predictor.time_embed(ts)                              # sinusoid (args up to 100 rad) + Linear
            emb = model.predictor.time_embed_extra(t_emb) + model.predictor.class_embed(labels)
        check(f"time_embed C={base}", t_emb, ref_cpu.time_embedding(ts, sd, "predictor.time_embed"))
        check(f"embedding C={base}", emb, ref_cpu.unet_embedding(sd, ts, labels))
        out[f"c{base}.emb"] = emb
        out[f"c{base}.labels"] = labels
    save("f2_time_embed", ts=ts, **out)


# ---------------------------------------------------------------- F5b ddpm_previous under the cosine schedule (schedule.py:34-41)
def gen_ddpm_previous_cos():
    model = DiffusionModel("unet", 32, schedule_name="cos")   # only .diffusion is used
    diff = model.diffusion
    cases = [(0.9, 0.1), (0.5, 0.02), (0.1, 0.02), (0.37, 0.01), (0.75, 0.25)]  # away from t = 1 (alpha_bar(1) ~ 4e-33)
    out = {}
    for i, (t, step) in enumerate(cases):
        x, eps, noise = seeded((2, 1, 4096), 310 + i), seeded((2, 1, 4096), 410 + i), seeded((2, 1, 4096), 510 + i)
        ts = torch.tensor([t, t])
        for mode, kw in (("plain", {}), ("sigma_large", dict(sigma_large=True)), ("constrain", dict(constrain=True))):
            y = diff.ddpm_previous(x, ts, step, eps, noise=noise, **kw)
            check(f"cos ddpm_previous t={t} {mode}", y, ref_cpu.ddpm_previous("cos", x, ts, step, eps, noise, **kw))
            out[f"c{i}.{mode}"] = y
        out[f"c{i}.x"], out[f"c{i}.eps"], out[f"c{i}.noise"] = x, eps, noise
        out[f"c{i}.t_step"] = np.array([t, step])
    save("f5b_ddpm_previous_cos", **out)


# ---------------------------------------------------------------- F11 decode_uncond_guidance (vq_vae.py:147-220)
def gen_uncond_guidance():
    import vq_voice_swap.diffusion.diffusion as dmod
    import vq_voice_swap.vq_vae as vmod

    model = det_model(VQVAE(base_channels=32, pred_name="unet", num_labels=5))  # label 0 = unconditional, 4 real labels
    with torch.no_grad():
        model.vq.dictionary.copy_(seeded(model.vq.dictionary.shape, 77, 0.35))
    sd = state_of(model)
    codes = torch.randint(0, 512, (2, 8), generator=torch.Generator().manual_seed(81))
    labels = torch.tensor([0, 2])
    steps, vq_scale, label_scale = 4, 1.5, 0.7
    x_T = seeded((2, 1, 2048), 82)
    gen = torch.Generator().manual_seed(83)
    noises = [torch.randn(x_T.shape, generator=gen) for _ in range(steps)]
    it = iter(noises)
    orig_rl, orig_r = torch.randn_like, torch.randn
    dmod.torch.randn_like = lambda t, *a, **k: next(it)
    vmod.torch.randn = lambda *a, **k: x_T.clone()
    try:
        with torch.no_grad():
            dec = model.decode_uncond_guidance(codes, labels, steps=steps, constrain=True, label_scale=label_scale, vq_scale=vq_scale)
    finally:
        dmod.torch.randn_like = orig_rl
        vmod.torch.randn = orig_r
    dec2 = ref_cpu.vqvae_decode_uncond_guidance(sd, 32, "exp", codes, labels, steps, x_T, noises, constrain=True,
                                                label_scale=label_scale, vq_scale=vq_scale)
    check("decode_uncond_guidance", dec, dec2, tol=1e-5)
    print(f"  x0 rms={dec.pow(2).mean().sqrt().item():.4f}")
    save("f11_uncond_guidance", x_T_seed=82, noise_seed=83, codes=codes, labels=labels, steps=steps,
         scales=np.array([vq_scale, label_scale]), x0=dec)


# ---------------------------------------------------------------- F6b: the headline workload itself
def gen_sampler_unet64():
    """BENCH's exact configuration on 2 clips: unet64, 50 steps, t**2 sample-time schedule (README.md:49), constrain=True
    (diffusion.py:92-133).  ~15 TFLOP of CPU work: minutes, run once."""
    model = det_model(DiffusionModel("unet", 64))
    sd = state_of(model)
    x_T = seeded((2, 1, 64000), 17)
    tmap = lambda t: t ** 2  # noqa: E731
    x0, noises = run_ref_sampler(model, x_T, 50, 19, True, tmap)
    print(f"  unet64 s50_sq_constrain: x0 rms={x0.pow(2).mean().sqrt().item():.4f}")
    save("f6b_sampler_unet64", x_T_seed=17, noise_seed=19, steps=50, x0=x0.to(torch.float32),
         noise_checksum=np.array([n.double().sum().item() for n in noises]))


# ---------------------------------------------------------------- F11b: decode_uncond_guidance at a real step count
def gen_uncond_guidance_50():
    import vq_voice_swap.diffusion.diffusion as dmod
    import vq_voice_swap.vq_vae as vmod

    model = det_model(VQVAE(base_channels=32, pred_name="unet", num_labels=5))
    with torch.no_grad():
        model.vq.dictionary.copy_(seeded(model.vq.dictionary.shape, 77, 0.35))
    sd = state_of(model)
    codes = torch.randint(0, 512, (2, 32), generator=torch.Generator().manual_seed(91))
    labels = torch.tensor([0, 3])
    steps, vq_scale, label_scale = 50, 1.5, 0.7
    x_T = seeded((2, 1, 8192), 92)
    gen = torch.Generator().manual_seed(93)
    noises = [torch.randn(x_T.shape, generator=gen) for _ in range(steps)]
    it = iter(noises)
    orig_rl, orig_r = torch.randn_like, torch.randn
    dmod.torch.randn_like = lambda t, *a, **k: next(it)
    vmod.torch.randn = lambda *a, **k: x_T.clone()
    try:
        with torch.no_grad():
            dec = model.decode_uncond_guidance(codes, labels, steps=steps, constrain=True, label_scale=label_scale, vq_scale=vq_scale)
    finally:
        dmod.torch.randn_like = orig_rl
        vmod.torch.randn = orig_r
    dec2 = ref_cpu.vqvae_decode_uncond_guidance(sd, 32, "exp", codes, labels, steps, x_T, noises, constrain=True,
                                                label_scale=label_scale, vq_scale=vq_scale)
    check("decode_uncond_guidance 50 steps", dec, dec2, tol=1e-5)
    print(f"  x0 rms={dec.pow(2).mean().sqrt().item():.4f}")
    save("f11b_uncond_guidance_50", x_T_seed=92, noise_seed=93, codes=codes, labels=labels, steps=steps,
         scales=np.array([vq_scale, label_scale]), x0=dec)


# ---------------------------------------------------------------- F12: what IS the reference's own code in ConvMFCCEncoder
def gen_conv_mfcc_stack():
    """models/conv_encoder.py:90-133 with torchaudio.transforms.MFCC replaced by a stub that returns an injected [B, 13, frames]
    tensor (torchaudio is not installed where fixtures are made; the stub exists in this script only): pins invert_ulaw, deltas
    x 2, the concatenation order, ResConv, the k = 4 / stride-2 convolution and the output convolution.  The MFCC transform itself
    stays unpinned."""
    import types

    captured = {}

    class StubMFCC(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
            self.kwargs = k

        def forward(self, wave):
            captured["wave"] = wave.detach().clone()
            return captured["inject"]

    ta = types.ModuleType("torchaudio")
    tt = types.ModuleType("torchaudio.transforms")
    tt.MFCC = StubMFCC
    ta.transforms = tt
    sys.modules["torchaudio"] = ta
    sys.modules["torchaudio.transforms"] = tt
    from vq_voice_swap.models.conv_encoder import ConvMFCCEncoder  # reference

    out = {}
    for tag, ulaw, T in (("ulaw_even", True, 4000), ("ulaw_odd", True, 4160), ("linear_even", False, 64000)):
        frames = T // 160 + 1
        enc = ConvMFCCEncoder(32, out_channels=512, input_ulaw=ulaw)
        det_init_((("encoder." + k, v) for k, v in enc.state_dict().items()))
        sd = {"encoder." + k: v.detach().clone() for k, v in enc.state_dict().items()}
        x = (0.4 * seeded((2, 1, T), 300 + len(out))).clamp(-1, 1)
        mf = seeded((2, 13, frames), 400 + len(out), 3.0)
        captured["inject"] = mf
        with torch.no_grad():
            z = enc(x)
        z2 = ref_cpu.conv_mfcc_encoder(sd, x, version=1, input_ulaw=ulaw, mfcc_override=mf)
        check("conv-mfcc stack " + tag, z, z2, tol=1e-5)
        want_wave = ref_cpu.invert_ulaw(x)[:, 0] if ulaw else x[:, 0]
        check("conv-mfcc invert_ulaw " + tag, captured["wave"], want_wave, tol=1e-7)
        out[tag + ".x"] = x
        out[tag + ".mfcc"] = mf
        out[tag + ".wave_seen_by_mfcc"] = captured["wave"]
        out[tag + ".z"] = z
        print(f"  {tag}: frames={frames} z {tuple(z.shape)} rms={z.pow(2).mean().sqrt().item():.4f}")
    save("f12_conv_mfcc_stack", **out)


# ---------------------------------------------------------------- F13: BASELINE config 5 at its own step count
def gen_guided_unet64_100():
    """unet64 sampled under classifier32's gradient at every one of 100 steps (BASELINE config 5; reference sample_diffusion.py:34-42
    + diffusion/diffusion.py:80-83, 92-133), 2 clips x T = 16384, constrain=True.  The cond_fn below is the closure sample_diffusion.py
    defines inside main() (it cannot be imported), around the reference's own Classifier and autograd."""
    import torch.nn.functional as F
    from vq_voice_swap.models import Classifier  # reference

    model = det_model(DiffusionModel("unet", 64))
    clf = Classifier(num_labels=7, base_channels=32)
    det_init_(("clf." + k, v) for k, v in clf.state_dict().items())
    clf.eval()
    sd_m, sd_c = state_of(model), state_of(clf)
    T, steps, scale = 16384, 100, 2000.0
    labels = torch.tensor([1, 6])
    x_T = seeded((2, 1, T), 151)

    def cond_fn(x, ts):  # sample_diffusion.py:34-42 with labels fixed and args.classifier_scale = scale
        with torch.enable_grad():
            x = x.detach().clone().requires_grad_()
            logits = clf(x, ts)
            logprobs = F.log_softmax(logits, dim=-1)
            grads = torch.autograd.grad(logprobs[range(len(x)), labels].sum(), x)[0]
            return grads.detach() * scale

    gen = torch.Generator().manual_seed(152)
    noises = [torch.randn(x_T.shape, generator=gen) for _ in range(steps)]
    it = iter(noises)
    import vq_voice_swap.diffusion.diffusion as dmod
    orig = torch.randn_like
    dmod.torch.randn_like = lambda t, *a, **k: next(it)
    try:
        x0 = model.diffusion.ddpm_sample(x_T, model.predictor, steps, constrain=True, cond_fn=cond_fn)
    finally:
        dmod.torch.randn_like = orig
    x0b = ref_cpu.ddpm_sample("exp", x_T, lambda p, q: ref_cpu.unet_predictor(sd_m, 64, p, q), steps, noises, constrain=True,
                              cond_fn=ref_cpu.classifier_cond_fn(sd_c, 32, labels, scale))
    check("guided unet64 + classifier32, 100 steps", x0, x0b, tol=1e-5)
    plain = ref_cpu.ddpm_sample("exp", x_T, lambda p, q: ref_cpu.unet_predictor(sd_m, 64, p, q), steps, noises, constrain=True)
    eff = (x0 - plain).pow(2).mean().sqrt().item()
    print(f"  x0 rms={x0.pow(2).mean().sqrt().item():.4f}; guided vs unguided rms={eff:.3e}; saturated={(x0.abs() >= 1).float().mean().item():.3f}")
    save("f13_guided_unet64_100", x_T_seed=151, noise_seed=152, steps=steps, scale=scale, labels=labels, x0=x0,
         guidance_effect_rms=eff, noise_checksum=np.array([n.double().sum().item() for n in noises]))


# ---------------------------------------------------------------- F8c: BASELINE config 4 at base 64 and its own step count
def gen_vqvae64_decode50():
    """VQVAE(base_channels=64).decode (vq_vae.py:92-145), 50 steps, T = 16384, constrain=True: codes -> vq.embed -> conditional unet64."""
    import vq_voice_swap.diffusion.diffusion as dmod
    import vq_voice_swap.vq_vae as vmod

    model = det_model(VQVAE(base_channels=64, pred_name="unet", num_labels=7))
    with torch.no_grad():
        model.vq.dictionary.copy_(seeded(model.vq.dictionary.shape, 78, 0.35))
    sd = state_of(model)
    T, steps = 16384, 50
    codes = torch.randint(0, 512, (2, T // 256), generator=torch.Generator().manual_seed(141))
    labels = torch.tensor([2, 5])
    x_T = seeded((2, 1, T), 142)
    gen = torch.Generator().manual_seed(143)
    noises = [torch.randn(x_T.shape, generator=gen) for _ in range(steps)]
    it = iter(noises)
    orig_rl, orig_r = torch.randn_like, torch.randn
    dmod.torch.randn_like = lambda t, *a, **k: next(it)
    vmod.torch.randn = lambda *a, **k: x_T.clone()
    try:
        with torch.no_grad():
            dec = model.decode(codes, labels, steps=steps, constrain=True)
    finally:
        dmod.torch.randn_like = orig_rl
        vmod.torch.randn = orig_r
    dec2 = ref_cpu.vqvae_decode(sd, 64, "exp", codes, labels, steps, x_T, noises, constrain=True)
    check("VQVAE(64) decode 50 steps", dec, dec2, tol=1e-5)
    print(f"  x0 rms={dec.pow(2).mean().sqrt().item():.4f}; saturated={(dec.abs() >= 1).float().mean().item():.3f}")
    save("f8c_vqvae64_decode50", x_T_seed=142, noise_seed=143, steps=steps, codes=codes, labels=labels, x0=dec,
         noise_checksum=np.array([n.double().sum().item() for n in noises]))


# ---------------------------------------------------------------- F14: the reference's OPEN topology (unet.py:17-30, 188-196)
def gen_custom_topologies():
    """UNetPredictor / UNetEncoder built with channel_mult / depth_mult / middle_dilations / out_dilations other than the defaults,
    by the reference's own constructors: forward outputs on seeded inputs."""
    from vq_voice_swap.models.unet import UNetEncoder, UNetPredictor  # reference

    out = {}
    preds = [
        ("p_a", dict(channel_mult=(1, 2, 2, 4), middle_dilations=(1, 6), depth_mult=1), dict(num_labels=3), 2048),
        ("p_b", dict(channel_mult=(1, 1, 2), middle_dilations=(), depth_mult=3), {}, 1024),
        ("p_c", dict(channel_mult=(1, 4, 8, 8, 16), middle_dilations=(2, 32, 5), depth_mult=2), dict(cond_channels=64), 4096),  # (channel_mult[0] must be 1: the reference normalises its output with base_channels, unet.py:113)
    ]
    for tag, topo, extra, T in preds:
        m = UNetPredictor(32, **topo, **extra)
        det_init_(("predictor." + tag + "." + k, v) for k, v in m.state_dict().items())
        m.eval()
        sd = {"predictor." + k: v.detach().clone() for k, v in m.state_dict().items()}
        x, ts = seeded((2, 1, T), 700 + len(out)), torch.tensor([0.4, 0.85])
        kw = {}
        if "num_labels" in extra:
            kw["labels"] = torch.tensor([2, 0])
        if "cond_channels" in extra:
            kw["cond"] = seeded((2, 64, 29), 750, 0.5)  # (a conditioning length that is no fixed fraction of T)
        with torch.no_grad():
            eps = m(x, ts, **kw)
        check("custom predictor " + tag, eps, ref_cpu.unet_predictor(sd, 32, x, ts, topology=topo, **kw))
        out[tag + ".x"], out[tag + ".ts"], out[tag + ".eps"] = x, ts, eps
        for k, v in kw.items():
            out[tag + "." + k] = v
        print(f"  {tag}: {topo} eps rms={eps.pow(2).mean().sqrt().item():.4f}, {sum(p.numel() for p in m.parameters()) / 1e6:.2f} M parameters")
    encs = [
        ("e_a", dict(channel_mult=(1, 2, 4), out_dilations=(2, 8), depth_mult=1), 64, 1024),
        ("e_b", dict(channel_mult=(1, 1, 2, 2, 4, 4), out_dilations=(), depth_mult=3), 96, 2048),
    ]
    for tag, topo, oc, T in encs:
        m = UNetEncoder(32, out_channels=oc, **topo)
        det_init_(("encoder." + tag + "." + k, v) for k, v in m.state_dict().items())
        m.eval()
        sd = {"encoder." + k: v.detach().clone() for k, v in m.state_dict().items()}
        x = seeded((2, 1, T), 800 + len(out), 0.3)
        with torch.no_grad():
            z = m(x)
        check("custom encoder " + tag, z, ref_cpu.unet_encoder(sd, 32, x, topology=topo))
        assert z.shape == (2, oc, T // m.downsample_rate)
        out[tag + ".x"], out[tag + ".z"] = x, z
        print(f"  {tag}: {topo} z {tuple(z.shape)} rms={z.pow(2).mean().sqrt().item():.4f}")
    save("f14_custom_topologies", **out)


# ---------------------------------------------------------------- F15: classifiers of non-default topology (classifier.py:52-58)
def gen_custom_classifiers():
    import torch.nn.functional as F
    from vq_voice_swap.models import Classifier  # reference

    out = {}
    for tag, kw, T in (("c_a", dict(channel_mult=(1, 2, 2, 4), output_mult=8, depth_mult=1), 4096),
                       ("c_b", dict(channel_mult=(1, 1, 2, 2, 2, 4), output_mult=4, depth_mult=3), 8192)):
        clf = Classifier(num_labels=5, base_channels=32, **kw)
        det_init_(("clf." + tag + "." + k, v) for k, v in clf.state_dict().items())
        clf.eval()
        sd = state_of(clf)
        x, ts, labels = seeded((2, 1, T), 900 + len(out)), torch.tensor([0.3, 0.75]), torch.tensor([4, 0])
        xg = x.clone().requires_grad_()
        logits = clf(xg, ts)
        grad = torch.autograd.grad(F.log_softmax(logits, dim=-1)[range(2), labels].sum(), xg)[0]
        topo = dict(channel_mult=kw["channel_mult"], depth_mult=kw["depth_mult"])
        check("custom classifier logits " + tag, logits.detach(), ref_cpu.classifier(sd, 32, x, ts, topology=topo), tol=1e-6)
        check("custom classifier grad " + tag, grad, ref_cpu.classifier_cond_fn(sd, 32, labels, topology=topo)(x, ts), tol=1e-6)
        out[tag + ".x"], out[tag + ".ts"], out[tag + ".labels"], out[tag + ".logits"], out[tag + ".grad"] = x, ts, labels, logits.detach(), grad
        print(f"  {tag}: {kw} logits rms={logits.pow(2).mean().sqrt().item():.4f} grad rms={grad.pow(2).mean().sqrt().item():.3e}")
    save("f15_custom_classifiers", **out)


if __name__ == "__main__":
    only = set(sys.argv[1:])
    if not only or "resblocks" in only:
        gen_resblocks()
    if not only or only & {"unet32", "ddpm", "sampler"}:
        m, sd = gen_unet32()
        if not only or "ddpm" in only:
            gen_ddpm_previous(m)
        if not only or "sampler" in only:
            gen_sampler(m, sd)
    if not only or "vqvae" in only:
        gen_vqvae()
    if not only or "classifier" in only:
        gen_classifier()
    if not only or "encpred" in only:
        gen_encpred()
    if not only or "time_embed" in only:
        gen_time_embed()
    if not only or "cos" in only:
        gen_ddpm_previous_cos()
    if not only or "uncond" in only:
        gen_uncond_guidance()
    if not only or "uncond50" in only:
        gen_uncond_guidance_50()
    if not only or "mfccstack" in only:
        gen_conv_mfcc_stack()
    if not only or "topology" in only:
        gen_custom_topologies()
    if not only or "classifier_topology" in only:
        gen_custom_classifiers()
    if "unet64" in only:  # (minutes of CPU time: only on request; the committed fixture is re-verifiable with this argument)
        gen_sampler_unet64()
    if "guided64" in only:  # (BASELINE config 5 at 100 steps: about a minute of CPU time, on request)
        gen_guided_unet64_100()
    if "vqvae64" in only:  # (BASELINE config 4 at base 64, 50 steps)
        gen_vqvae64_decode50()
    print("ok")
