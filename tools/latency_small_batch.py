"""UNet forward latency at small batch sizes (single-clip sampling is the reference scripts' default)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vq_voice_swap_amd import DiffusionModel
from vq_voice_swap_amd.det_init import det_init_
dev = torch.device("cuda:0")
for base in (64, 32):
    m = DiffusionModel("unet", base); det_init_(m.state_dict().items()); m.eval().to(dev); m.set_precision("bf16")
    for B in (1, 2, 4, 8, 16, 32, 64):
        x = torch.randn(B, 1, 64000, device=dev); ts = torch.full((B,), 0.5, device=dev)
        m.predictor(x, ts); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): m.predictor(x, ts)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(f"unet{base} B={B:2d}: {dt * 1e3:7.3f} ms/forward  {dt * 1e3 / B:6.3f} ms/clip  ({B / dt / 50:6.1f} clips/s at 50 steps)", flush=True)
