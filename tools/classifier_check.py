"""Parity numbers and timing of the native classifier (forward and guidance gradient) on the GPU."""
import os
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # kernel arguments in device memory: a process-level HIP switch, before the runtime starts
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from vq_voice_swap_amd import Classifier
from vq_voice_swap_amd.det_init import det_init_
from util import rel_rms, seeded

dev = torch.device("cuda:0")
z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "f9_classifier32.npz"))
clf = Classifier(num_labels=7, base_channels=32)
det_init_(clf.state_dict().items())
clf.eval().to(dev)
x = seeded((2, 1, 64000), int(z["x_seed"])).to(dev)
ts, labels = torch.from_numpy(z["ts"]).to(dev), torch.from_numpy(z["labels"]).to(dev)
for prec in ("fp32", "bf16"):
    clf.set_precision(prec)
    g, lg = clf.log_prob_grad(x, ts, labels, 1.0, return_logits=True)
    print(prec, "logits rel", rel_rms(lg.cpu(), torch.from_numpy(z["logits"])), "grad rel", rel_rms(g.cpu(), torch.from_numpy(z["grad"])))
for base, B in ((32, 32), (32, 64), (64, 32)):
    clf = Classifier(num_labels=251, base_channels=base)
    det_init_(clf.state_dict().items())
    clf.eval().to(dev)
    xb = torch.randn(B, 1, 64000, device=dev)
    tb = torch.rand(B, device=dev)
    lb = torch.randint(0, 251, (B,), device=dev)
    for prec in ("bf16", "fp32"):
        clf.set_precision(prec)
        for fn, name in ((lambda: clf(xb, tb), "forward"), (lambda: clf.log_prob_grad(xb, tb, lb), "guidance")):
            fn(); torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            print(f"classifier{base} B={B} {prec} {name}: {(time.time() - t0) / 5 * 1e3:.2f} ms", flush=True)
        h = clf.handle(dev, B, 64000)
        print("   device bytes %.2f GB, kernels %d" % (h.device_bytes() / 1e9, h.kernel_count()))

# ---- encoder predictor (guidance through the whole UNet)
from vq_voice_swap_amd import EncoderPredictor
z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "f10_encpred32.npz"))
ep = EncoderPredictor(32, 256, 96)
det_init_(ep.state_dict().items())
ep.eval().to(dev)
x = seeded((2, 1, 16384), int(z["x_seed"])).to(dev)
ts, targets = torch.from_numpy(z["ts"]).to(dev), torch.from_numpy(z["targets"]).to(dev)
for prec in ("fp32", "bf16"):
    ep.set_precision(prec)
    print("encpred", prec, "logits rel", rel_rms(ep(x, ts).cpu(), torch.from_numpy(z["logits"])),
          "grad rel", rel_rms(ep.guidance_grad(x, ts, targets).cpu(), torch.from_numpy(z["grad"])))
for base, B in ((32, 32), (64, 32)):
    ep = EncoderPredictor(base, 256, 512)
    det_init_(ep.state_dict().items())
    ep.eval().to(dev)
    xb = torch.randn(B, 1, 64000, device=dev)
    tb = torch.rand(B, device=dev)
    tg = torch.randint(0, 512, (B, 250), device=dev)
    for prec in ("bf16", "fp32"):
        ep.set_precision(prec)
        for fn, name in ((lambda: ep(xb, tb), "forward"), (lambda: ep.guidance_grad(xb, tb, tg), "guidance")):
            fn(); torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            print(f"encpred{base} B={B} {prec} {name}: {(time.time() - t0) / 3 * 1e3:.2f} ms", flush=True)
        h = ep.handle(dev, B, 64000)
        print("   device bytes %.2f GB, kernels %d" % (h.device_bytes() / 1e9, h.kernel_count()))
