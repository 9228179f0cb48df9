import os
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # kernel arguments in device memory: a process-level HIP switch, before the runtime starts
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from oracle import ref_cpu
from vq_voice_swap_amd.det_init import det_tensor
from vq_voice_swap_amd import _native
cfg = _native.Cfg(); cfg.kind, cfg.base_channels, cfg.in_channels, cfg.out_channels = 0, 64, 1, 1
sd = {"predictor." + n: det_tensor("predictor." + n, s) for n, s in _native.param_table(cfg)}
print("hw threads", os.cpu_count())
for nb in (4, 8, 16):
    x = torch.randn(nb, 1, 64000); ts = torch.full((nb,), 0.5)
    for th in (16, 32, 64, 128):
        torch.set_num_threads(th)
        with torch.no_grad():
            ref_cpu.unet_predictor(sd, 64, x[:1], ts[:1])
            t0 = time.time(); ref_cpu.unet_predictor(sd, 64, x, ts); dt = time.time() - t0
        print(f"B={nb} threads={th}: {dt:.2f} s per forward -> {nb/dt/50:.4f} clips/s at 50 steps", flush=True)
