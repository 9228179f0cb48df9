"""UNet forward latency at small batch sizes (single-clip sampling is the reference scripts' default)."""
import os
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # kernel arguments in device memory: a process-level HIP switch, before the runtime starts
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

# how much of the small-batch latency is host launch cost?  (time to ENQUEUE one forward vs time to finish it)
m = DiffusionModel("unet", 64); det_init_(m.state_dict().items()); m.eval().to(dev); m.set_precision("bf16")
for B in (1, 8):
    x = torch.randn(B, 1, 64000, device=dev); ts = torch.full((B,), 0.5, device=dev)
    m.predictor(x, ts); torch.cuda.synchronize()
    enq = 0.0; tot = 0.0
    for _ in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.predictor(x, ts); t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        enq += t1 - t0; tot += t2 - t0
    print(f"unet64 B={B}: enqueue {enq * 100:.3f} ms, enqueue + execute {tot * 100:.3f} ms (305 launches)")
