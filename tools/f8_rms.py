"""Waveform RMS of the VQ-VAE decode fixture F8 (5 steps, constrained, 2 x 4096) per precision mode -- the thinnest margin of the
1e-3 gate, used to bisect numerics changes:  [VQVS_LIB_PATH=...] python tools/f8_rms.py"""
import os
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # kernel arguments in device memory: a process-level HIP switch, before the runtime starts
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vq_voice_swap_amd import VQVAE
from vq_voice_swap_amd.det_init import det_init_
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import seeded

dev = torch.device("cuda:0")
FIX = {"F8": ("f8_vqvae_decode.npz", 5), "F8b": ("f8b_vqvae_decode50.npz", 50)}
model = VQVAE(base_channels=32, pred_name="unet", num_labels=5)
det_init_(model.state_dict().items())
model.eval()
with torch.no_grad():
    model.vq.dictionary.copy_(seeded(model.vq.dictionary.shape, 77, 0.35))
for tag, (fn, steps) in FIX.items():
    z8 = np.load(os.path.join(ROOT, "tests", "golden", fn))
    x_T = seeded((2, 1, 4096), int(z8["x_T_seed"]))
    gen = torch.Generator().manual_seed(int(z8["noise_seed"]))
    noises = [torch.randn(x_T.shape, generator=gen).to(dev) for _ in range(steps)]
    want = torch.from_numpy(z8["x0"])
    for prec in ("fp32", "fp16"):
        model.set_precision(prec)
        dec = model.decode(torch.from_numpy(z8["codes16"]).to(dev), torch.from_numpy(z8["labels"]).to(dev), steps=steps, constrain=True,
                           x_T=x_T.to(dev), noise=noises).cpu()
        d = dec - want
        unsat = want.abs() < 1.0
        print(f"{tag} {prec}: rms {d.pow(2).mean().sqrt().item():.4e}  unsaturated-only rms {d[unsat].pow(2).mean().sqrt().item():.4e}  "
              f"saturated fraction {1 - unsat.float().mean().item():.2f}  lib {os.environ.get('VQVS_LIB_PATH', 'default')} RES={os.environ.get('VQVS_WS_RES', '1')} WS={os.environ.get('VQVS_WS', '1')}")
