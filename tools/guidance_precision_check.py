import os
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # kernel arguments in device memory: a process-level HIP switch, before the runtime starts
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, torch
from vq_voice_swap_amd import Classifier, EncoderPredictor
from vq_voice_swap_amd.det_init import det_init_
from util import rel_rms, seeded
G = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests", "golden")
dev = torch.device("cuda:0")
z = np.load(os.path.join(G, "f9_classifier32.npz"))
for prec in ("fp32", "fp16", "bf16"):
    clf = Classifier(num_labels=7, base_channels=32); det_init_(clf.state_dict().items()); clf.eval().to(dev); clf.set_precision(prec)
    x = seeded((2, 1, 64000), int(z["x_seed"])).to(dev)
    ts, labels = torch.from_numpy(z["ts"]).to(dev), torch.from_numpy(z["labels"]).to(dev)
    g, lg = clf.log_prob_grad(x, ts, labels, 1.0, return_logits=True)
    g3 = clf.guidance_fn(labels, 3.0)(x, ts)
    print(prec, "classifier logits rel", rel_rms(lg.cpu(), torch.from_numpy(z["logits"])), "grad rel", rel_rms(g.cpu(), torch.from_numpy(z["grad"])), "3x linearity", rel_rms(g3.cpu(), 3.0 * g.cpu()))
z = np.load(os.path.join(G, "f10_encpred32.npz"))
for prec in ("fp32", "fp16", "bf16"):
    ep = EncoderPredictor(base_channels=32, downsample_rate=256, num_latents=96, bottleneck_dim=64); det_init_(ep.state_dict().items()); ep.eval().to(dev); ep.set_precision(prec)
    x = seeded((2, 1, 16384), int(z["x_seed"])).to(dev)
    ts, targets = torch.from_numpy(z["ts"]).to(dev), torch.from_numpy(z["targets"]).to(dev)
    lg = ep(x, ts).cpu(); g = ep.guidance_grad(x, ts, targets, 1.0)
    print(prec, "encpred logits rel", rel_rms(lg, torch.from_numpy(z["logits"])), "grad rel", rel_rms(g.cpu(), torch.from_numpy(z["grad"])), "0.5x linearity", rel_rms(ep.guidance_fn(targets, 0.5)(x, ts).cpu(), 0.5 * g.cpu()))
