"""Per-kernel table of one UNet forward (live hipEvent timing): shape, ms, achieved GB/s and TFLOP/s."""
import os
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # kernel arguments in device memory: a process-level HIP switch, before the runtime starts
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vq_voice_swap_amd import DiffusionModel
from vq_voice_swap_amd.det_init import det_init_

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="unet64")
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--precision", default="bf16")
ap.add_argument("--T", type=int, default=64000)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda:0")
base = 64 if a.model == "unet64" else 32
m = DiffusionModel("unet", base)
det_init_(m.state_dict().items())
m.set_precision(a.precision)
x = torch.randn(a.batch, 1, a.T, device=dev)
ts = torch.full((a.batch,), 0.5, device=dev)
m.predictor(x, ts)
h = m.predictor._handle
h.set_profiling(True)
acc = None
for _ in range(a.reps):
    m.predictor(x, ts)
    ms = h.profile_read()
    acc = ms if acc is None else [p + q for p, q in zip(acc, ms)]
ms = [v / a.reps for v in acc]
info, desc = h.op_info(a.batch, a.T), h.op_desc()
tot = sum(ms)
print(f"{a.model} {a.precision} B={a.batch} T={a.T}: {len(ms)} kernels, {tot:.3f} ms/forward (event sum)")
print(f"{'#':>3} {'kind':10s} {'shape':34s} {'ms':>8s} {'%':>5s} {'GB/s':>8s} {'TF/s':>7s}")
for i, (t, (kind, by, fl), d) in enumerate(zip(ms, info, desc)):
    if t < 0.02 and kind != "conv":
        continue
    print(f"{i:3d} {kind:10s} {d:34s} {t:8.3f} {100*t/tot:5.1f} {by/t/1e6 if t else 0:8.0f} {fl/t/1e9 if t else 0:7.1f}")
by_kind = {}
for t, (kind, by, fl) in zip(ms, info):
    d = by_kind.setdefault(kind, [0.0, 0, 0, 0]); d[0] += t; d[1] += by; d[2] += fl; d[3] += 1
for k, (t, by, fl, n) in sorted(by_kind.items(), key=lambda kv: -kv[1][0]):
    print(f"  {k:12s} n={n:3d} {t:8.3f} ms  {by/t/1e6 if t else 0:8.0f} GB/s  {fl/t/1e9 if t else 0:7.1f} TF/s")
