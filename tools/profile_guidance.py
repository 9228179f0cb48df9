"""Per-kernel-kind time of one classifier guidance call (forward + explicit backward), live hipEvent timing."""
import os
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # kernel arguments in device memory: a process-level HIP switch, before the runtime starts
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vq_voice_swap_amd import Classifier, EncoderPredictor
from vq_voice_swap_amd.det_init import det_init_

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "classifier"
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
B, T = 32, 64000
if which == "classifier":
    m = Classifier(num_labels=251, base_channels=32)
else:
    m = EncoderPredictor(64, 256, 512)
det_init_(m.state_dict().items())
m.eval().to(dev)
m.set_precision(prec)
x = torch.randn(B, 1, T, device=dev)
ts = torch.rand(B, device=dev)
if which == "classifier":
    tgt = torch.randint(0, 251, (B,), device=dev)
    run = lambda: m.log_prob_grad(x, ts, tgt)
else:
    tgt = torch.randint(0, 512, (B, T // 256), device=dev)
    run = lambda: m.guidance_grad(x, ts, tgt)
run()
h = m._handle
h.set_profiling(True)
acc = None
for _ in range(3):
    run()
    ms = h.profile_read()
    acc = ms if acc is None else [p + q for p, q in zip(acc, ms)]
ms = [v / 3 for v in acc]
info, desc = h.op_info(B, T), h.op_desc()
by = {}
for t, (kind, nb, fl), d in zip(ms, info, desc):
    e = by.setdefault(kind, [0.0, 0, 0])
    e[0] += t; e[1] += nb; e[2] += 1
print(f"{which} {prec} B={B}: {sum(ms):.3f} ms over {len(ms)} kernels")
for k, (t, nb, n) in sorted(by.items(), key=lambda kv: -kv[1][0]):
    print(f"  {k:12s} n={n:3d} {t:8.3f} ms   {nb / t / 1e6 if t else 0:8.0f} GB/s")
big = sorted(zip(ms, info, desc), key=lambda z: -z[0])[:12]
for t, (kind, nb, fl), d in big:
    print(f"    {kind:10s} {d:34s} {t:7.3f} ms {nb / t / 1e6:8.0f} GB/s")
