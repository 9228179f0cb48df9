"""Clips/s of the other BASELINE.json configurations on ONE MI355X (per-GPU share of the 8-GPU batch):
cfg 2 unet32 B=64; cfg 4 VQ-VAE (encoder + VQ + unet64 decoder, labels, cond) B=32; cfg 5 unet64 + classifier32
guidance, 100 steps, B=32.  Writes one JSON object (profiles/r01_configs.json is a committed copy)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vq_voice_swap_amd import Classifier, DiffusionModel, VQVAE, randn_clips
from vq_voice_swap_amd.det_init import det_init_

dev = torch.device("cuda:0")
T = 64000
out = {}


def timed(fn, reps=2):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def det(m):
    det_init_(m.state_dict().items())
    return m.eval().to(dev)


for prec in ("bf16", "fp32"):
    # cfg 2
    m = det(DiffusionModel("unet", 32)); m.set_precision(prec)
    x = randn_clips(64, T, dev, 1)
    dt = timed(lambda: m.diffusion.ddpm_sample(x, m.predictor, 50, constrain=True, schedule=lambda t: t ** 2, seed=3))
    out[f"cfg2_unet32_B64_50steps_{prec}"] = {"clips_per_s": round(64 / dt, 2), "s_per_batch": round(dt, 3)}
    del m
    # cfg 3 share (unet64 B=64) for reference, same path as bench.py
    m = det(DiffusionModel("unet", 64)); m.set_precision(prec)
    dt = timed(lambda: m.diffusion.ddpm_sample(x, m.predictor, 50, constrain=True, schedule=lambda t: t ** 2, seed=3), reps=1)
    out[f"cfg3_unet64_B64_50steps_{prec}"] = {"clips_per_s": round(64 / dt, 2), "s_per_batch": round(dt, 3)}
    # cfg 5: classifier guidance, 100 steps, 32 clips per GPU
    clf = det(Classifier(num_labels=251, base_channels=32)); clf.set_precision(prec)
    labels = torch.arange(32, device=dev) % 251
    x32 = x[:32].contiguous()
    dt = timed(lambda: m.diffusion.ddpm_sample(x32, m.predictor, 100, constrain=True, cond_fn=clf.guidance_fn(labels, 1.0), seed=3), reps=1)
    out[f"cfg5_unet64_classifier32_B32_100steps_{prec}"] = {"clips_per_s": round(32 / dt, 2), "s_per_batch": round(dt, 3)}
    del m, clf
    # cfg 4: VQ-VAE speaker conversion
    v = det(VQVAE(base_channels=64, enc_name="unet", pred_name="unet", num_labels=251)); v.set_precision(prec)
    wav = (0.1 * torch.randn(32, 1, T, device=dev)).clamp(-1, 1)

    def convert():
        codes = v.encode(wav)
        return v.decode(codes, labels, steps=50, constrain=True)

    dt = timed(convert, reps=1)
    out[f"cfg4_vqvae64_B32_50steps_{prec}"] = {"clips_per_s": round(32 / dt, 2), "s_per_batch": round(dt, 3)}
    del v
    torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
