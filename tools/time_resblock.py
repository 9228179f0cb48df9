"""Time one ResBlock (both convolutions + the small kernels around them) in isolation:  python tools/time_resblock.py cin cout L B [prec]
Short bursts (5 forwards after a pause) and a sustained run (200 forwards) -- the difference is the clock the chip sustains."""
import os
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # kernel arguments in device memory: a process-level HIP switch, before the runtime starts
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vq_voice_swap_amd.unet import ResBlockModule
from vq_voice_swap_amd.det_init import det_init_

cin, cout, L, B = (int(v) for v in sys.argv[1:5])
prec = sys.argv[5] if len(sys.argv) > 5 else "fp16"
dev = torch.device("cuda:0")
m = ResBlockModule(cin, 256, cout, 1.0, 2)
det_init_(m.state_dict().items())
m.set_precision(prec)
m.to(dev)
x = torch.randn(B, cin, L, device=dev)
e = torch.randn(B, 256, device=dev)
for _ in range(3):
    m(x, e)
torch.cuda.synchronize()


def run(n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        m(x, e)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for k in range(3):
    time.sleep(0.5)
    print(f"burst of 5: {run(5):.4f} ms per ResBlock forward")
print(f"sustained 300: {run(300):.4f} ms per ResBlock forward")
print(f"sustained 300: {run(300):.4f} ms per ResBlock forward")
