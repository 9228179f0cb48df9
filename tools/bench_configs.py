"""Clips/s of the other BASELINE.json configurations on ONE MI355X (per-GPU share of the 8-GPU batch), in the gate modes:
cfg 2 unet32 B=64; cfg 3 unet64 B=64 (the bench.py workload, for reference); cfg 4 VQ-VAE conversion (UNet or MFCC encoder +
VQ + conditional unet64 decoder) B=32; cfg 5 unet64 + classifier32 guidance, 100 steps, B=32.  One JSON object on stdout;
every entry carries the end-to-end HBM fraction = algorithmic (Model A) bytes of all forward / backward passes of the run /
wall time / 8 TB/s.  `--only cfg5 --precision fp16 --reps 1` is what tools/measure.sh profiles under rocprofv3."""
import os
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # kernel arguments in device memory: a process-level HIP switch, before the runtime starts
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vq_voice_swap_amd import Classifier, DiffusionModel, VQVAE, randn_clips  # noqa: E402
from vq_voice_swap_amd.det_init import det_init_  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="fp16,fp32")
ap.add_argument("--only", default="cfg2,cfg3,cfg4,cfg4mfcc,cfg5")
ap.add_argument("--reps", type=int, default=1)
a = ap.parse_args()
dev = torch.device("cuda:0")
T = 64000
HBM = 8000e9
out = {}


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def det(m):
    det_init_((k, v) for k, v in m.state_dict().items() if ".mfcc." not in k)
    return m.eval().to(dev)


def entry(name, clips, dt, bytes_moved):
    out[name] = {"clips_per_s": round(clips / dt, 2), "s_per_batch": round(dt, 3),
                 "e2e_hbm_frac": round(bytes_moved / dt / HBM, 4), "algorithmic_GB_per_batch": round(bytes_moved / 1e9, 1)}


only = set(a.only.split(","))
for prec in a.precision.split(","):
    x = randn_clips(64, T, dev, 1)
    if "cfg2" in only:
        m = det(DiffusionModel("unet", 32)); m.set_precision(prec)
        dt = timed(lambda: m.diffusion.ddpm_sample(x, m.predictor, 50, constrain=True, schedule=lambda t: t ** 2, seed=3), a.reps)
        entry(f"cfg2_unet32_B64_50steps_{prec}", 64, dt, 50 * m.predictor.handle(dev, 64, T).model_bytes(64, T))
        del m
    if only & {"cfg3", "cfg5"}:
        m = det(DiffusionModel("unet", 64)); m.set_precision(prec)
        if "cfg3" in only:
            dt = timed(lambda: m.diffusion.ddpm_sample(x, m.predictor, 50, constrain=True, schedule=lambda t: t ** 2, seed=3), a.reps)
            entry(f"cfg3_unet64_B64_50steps_{prec}", 64, dt, 50 * m.predictor.handle(dev, 64, T).model_bytes(64, T))
        if "cfg5" in only:  # classifier guidance: classifier32 forward + explicit backward at every one of the 100 steps
            clf = det(Classifier(num_labels=251, base_channels=32)); clf.set_precision(prec)
            labels = torch.arange(32, device=dev) % 251
            x32 = x[:32].contiguous()
            dt = timed(lambda: m.diffusion.ddpm_sample(x32, m.predictor, 100, constrain=True, cond_fn=clf.guidance_fn(labels, 1.0), seed=3), a.reps)
            by = 100 * (m.predictor.handle(dev, 32, T).model_bytes(32, T) + clf.handle(dev, 32, T).model_bytes(32, T))
            entry(f"cfg5_unet64_classifier32_B32_100steps_{prec}", 32, dt, by)
            del clf
        del m
    for tag, enc in (("cfg4", "unet"), ("cfg4mfcc", "conv-mfcc-ulaw")):
        if tag not in only:
            continue
        v = det(VQVAE(base_channels=64, enc_name=enc, pred_name="unet", num_labels=251)); v.set_precision(prec)
        wav = (0.1 * torch.randn(32, 1, T, device=dev)).clamp(-1, 1)
        labels = torch.arange(32, device=dev) % 251

        def convert():
            return v.decode(v.encode(wav), labels, steps=50, constrain=True)

        dt = timed(convert, a.reps)
        by = 50 * v.predictor._handle.model_bytes(32, T) + v.encoder._handle.model_bytes(32, T)
        entry(f"{tag}_vqvae64_{enc}_B32_50steps_{prec}", 32, dt, by)
        t_enc = timed(lambda: v.encode(wav), 3)
        out[f"{tag}_vqvae64_{enc}_B32_50steps_{prec}"]["encode_ms"] = round(t_enc * 1e3, 2)
        del v
    torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
