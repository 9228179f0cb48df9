"""Run one ResBlock a few times (a target for rocprofv3 --pmc / --kernel-trace):  python tools/run_resblock.py cin cout L B [reps] [prec]"""
import os
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # kernel arguments in device memory: a process-level HIP switch, before the runtime starts
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vq_voice_swap_amd.unet import ResBlockModule
from vq_voice_swap_amd.det_init import det_init_

cin, cout, L, B = (int(v) for v in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
prec = sys.argv[6] if len(sys.argv) > 6 else "fp16"
dev = torch.device("cuda:0")
m = ResBlockModule(cin, 256, cout, 1.0, 2)
det_init_(m.state_dict().items())
m.set_precision(prec)
m.to(dev)
x = torch.randn(B, cin, L, device=dev)
e = torch.randn(B, 256, device=dev)
for _ in range(reps):
    y = m(x, e)
torch.cuda.synchronize()
print("ok", tuple(y.shape))
