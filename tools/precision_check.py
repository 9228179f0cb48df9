"""Developer diagnostic (not a test): error of every precision mode of the HIP path against the golden vectors
and the CPU oracle -- the numbers DESIGN.md section 4 quotes and the tolerances in tests/ are set from.

    python tools/precision_check.py [--modes fp32,fp16,bf16] [--skip-unet64]
"""
import os
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # kernel arguments in device memory: a process-level HIP switch, before the runtime starts
import argparse
import os
import sys
import time
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_cpu  # noqa: E402
from vq_voice_swap_amd import DiffusionModel, ResBlockModule, VQVAE  # noqa: E402
from vq_voice_swap_amd.det_init import det_init_  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--modes", default="fp32,fp16,bf16")
ap.add_argument("--skip-unet64", action="store_true")
a = ap.parse_args()
MODES = a.modes.split(",")
G = os.path.join(ROOT, "tests", "golden")
dev = torch.device("cuda:0")
torch.set_num_threads(min(32, os.cpu_count() or 8))


def seeded(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


def rel_rms(x, y):
    return ((x - y).pow(2).mean().sqrt() / y.pow(2).mean().sqrt().clamp_min(1e-12)).item()


def rms(x):
    return x.pow(2).mean().sqrt().item()


def section(name):
    print("\n==== " + name, flush=True)


def guarded(fn):
    try:
        fn()
    except Exception:
        traceback.print_exc()


def det(m):
    det_init_(m.state_dict().items())
    return m.eval()


def resblocks():
    section("F1 resblocks (rel. RMS vs golden)")
    z = np.load(os.path.join(G, "f1_resblocks.npz"))
    for name in sorted({k.split(".")[0] for k in z.files}):
        cin, cout, scale, dil, emb, L = z[name + ".spec"]
        row = []
        for prec in MODES:
            m = ResBlockModule(int(cin), int(emb) or None, int(cout) if cout != cin else None, float(scale), int(dil))
            det_init_(("blk." + name + "." + k, v) for k, v in m.block.state_dict().items())
            m.set_precision(prec)
            e = torch.from_numpy(z[name + ".emb"]).to(dev) if emb else None
            y = m(torch.from_numpy(z[name + ".x"]).to(dev), e).cpu()
            row.append(f"{prec}={rel_rms(y, torch.from_numpy(z[name + '.y'])):.2e}")
        print(f"  {name:12s} " + "  ".join(row), flush=True)


def unet32_forward():
    section("F3 unet32 forward (eps rel. RMS vs golden)")
    z = np.load(os.path.join(G, "f3_unet32_forward.npz"))
    x = seeded((2, 1, 64000), int(z["x_seed"]))
    ts = torch.from_numpy(z["ts"])
    want = torch.from_numpy(z["eps"])
    for prec in MODES:
        model = det(DiffusionModel("unet", 32))
        model.set_precision(prec)
        eps = model.predictor(x.to(dev), ts.to(dev)).cpu()
        print(f"  {prec}: rel_rms={rel_rms(eps, want):.3e}  max|d|={(eps - want).abs().max().item():.3e}  finite={bool(torch.isfinite(eps).all())}", flush=True)


def sampler():
    section("F6 sampler end to end (waveform RMS vs golden; plain = relative)")
    z = np.load(os.path.join(G, "f6_sampler_unet32.npz"))
    x_T = seeded((2, 1, 64000), int(z["x_T_seed"]))
    for tag, steps, constrain, sq in (("s10_plain", 10, False, False), ("s10_constrain", 10, True, False), ("s50_sq_constrain", 50, True, True)):
        gen = torch.Generator().manual_seed(int(z["noise_seed"]))
        noises = [torch.randn(x_T.shape, generator=gen).to(dev) for _ in range(steps)]
        want = torch.from_numpy(z[tag + ".x0"])
        row = []
        for prec in MODES:
            model = det(DiffusionModel("unet", 32))
            model.set_precision(prec)
            got = model.diffusion.ddpm_sample(x_T.to(dev), model.predictor, steps, constrain=constrain, schedule=(lambda t: t ** 2) if sq else None,
                                              noise=noises).cpu()
            err = rms(got - want) if constrain else rel_rms(got, want)
            row.append(f"{prec}={err:.3e}")
        print(f"  {tag:18s} " + "  ".join(row), flush=True)


def vqvae():
    section("F7/F8 VQ-VAE: codes (encoder always fp32) and 5-step decode RMS vs golden")
    z7, z8 = np.load(os.path.join(G, "f7_encoder_vq32.npz")), np.load(os.path.join(G, "f8_vqvae_decode.npz"))
    for prec in MODES:
        model = det(VQVAE(base_channels=32, pred_name="unet", num_labels=5))
        with torch.no_grad():
            model.vq.dictionary.copy_(seeded(model.vq.dictionary.shape, 77, 0.35))
        model.set_precision(prec)
        wav = seeded((2, 1, 64000), int(z7["wav_seed"]), 0.1).clamp(-1, 1)
        codes = model.encode(wav.to(dev)).cpu()
        mism = int((codes != torch.from_numpy(z7["codes"])).sum())
        x_T = seeded((2, 1, 4096), int(z8["x_T_seed"]))
        gen = torch.Generator().manual_seed(int(z8["noise_seed"]))
        noises = [torch.randn(x_T.shape, generator=gen).to(dev) for _ in range(5)]
        dec = model.decode(torch.from_numpy(z8["codes16"]).to(dev), torch.from_numpy(z8["labels"]).to(dev), steps=5, constrain=True,
                           x_T=x_T.to(dev), noise=noises).cpu()
        print(f"  {prec}: code mismatches {mism}/500 (encoder precision {model.encoder.precision}), decode RMS={rms(dec - torch.from_numpy(z8['x0'])):.3e}", flush=True)


def unet64():
    section("unet64 full-length forward vs oracle (1 clip) and 10-step constrained sample (2 clips)")
    model = det(DiffusionModel("unet", 64))
    sd = {"predictor." + k: v.detach() for k, v in model.predictor.state_dict().items()}
    x, ts = seeded((1, 1, 64000), 9), torch.tensor([0.4])
    t0 = time.time()
    want = ref_cpu.unet_predictor(sd, 64, x, ts)
    print(f"  (oracle forward {time.time() - t0:.1f}s)")
    for prec in MODES:
        model.set_precision(prec)
        got = model.predictor(x.to(dev), ts.to(dev)).cpu()
        print(f"  forward {prec}: rel_rms={rel_rms(got, want):.3e}", flush=True)
    x_T = seeded((2, 1, 64000), 21)
    gen = torch.Generator().manual_seed(22)
    noises = [torch.randn(x_T.shape, generator=gen) for _ in range(10)]
    t0 = time.time()
    want = ref_cpu.ddpm_sample("exp", x_T, lambda p, q: ref_cpu.unet_predictor(sd, 64, p, q), 10, noises, constrain=True)
    print(f"  (oracle 10-step sample {time.time() - t0:.1f}s)")
    for prec in MODES:
        model.set_precision(prec)
        got = model.diffusion.ddpm_sample(x_T.to(dev), model.predictor, 10, constrain=True, noise=[n.to(dev) for n in noises]).cpu()
        print(f"  10-step constrained sample {prec}: waveform RMS={rms(got - want):.3e}", flush=True)


for fn in (resblocks, unet32_forward, sampler, vqvae):
    guarded(fn)
if not a.skip_unet64:
    guarded(unet64)
