"""CPU study (oracle with simulated rounding): which rounding of the bf16 mode costs the accuracy?
   op  = round the MFMA operands (post-GELU activations and weights) to bf16
   h1  = store the block-internal tensor h1 in bf16
   res = store the residual stream (block outputs, in_conv output) in bf16
Reports the relative error of eps (unet32, 2 x 16384 samples) against the unrounded oracle."""
import os
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # kernel arguments in device memory: a process-level HIP switch, before the runtime starts
import os, sys, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
from oracle import ref_cpu
from vq_voice_swap_amd import DiffusionModel
from vq_voice_swap_amd.det_init import det_init_

torch.set_num_threads(16)
bf = lambda t: t.to(torch.bfloat16).to(torch.float32)  # noqa: E731
MODE = dict(op=False, h1=False, res=False, split2=False)
orig_conv = F.conv1d
orig_res_block = ref_cpu.res_block


def conv_hook(x, w, b=None, **kw):
    if MODE["op"] and w.shape[1] > 1:      # the 1 -> C input conv runs in fp32 FMAs in the library
        if MODE["split2"]:                  # activations as hi + lo (two MFMAs), weights bf16
            hi = bf(x)
            return orig_conv(hi, bf(w), b, **kw) + orig_conv(bf(x - hi), bf(w), None, **kw)
        return orig_conv(bf(x), bf(w), b, **kw)
    return orig_conv(x, w, b, **kw)


def res_block(x, sd, p, spec, emb):
    scale, dil = spec["scale"], spec["dil"]
    h = F.gelu(ref_cpu.group_norm(x, sd, p + ".pre_cond.0.0"))
    h = ref_cpu.resize(h, scale)
    h = conv_hook(h, sd[p + ".pre_cond.2.weight"], sd[p + ".pre_cond.2.bias"], padding=1)
    if MODE["h1"]:
        h = bf(h)
    h = ref_cpu.group_norm(h, sd, p + ".pre_cond.3")
    if emb is not None:
        ab = F.linear(F.gelu(emb), sd[p + ".cond_layers.1.weight"], sd[p + ".cond_layers.1.bias"])
        cout = spec["cout"]
        a, b = ab[:, :cout, None], ab[:, cout:, None]
        h = h * (a + 1) + b
    conv2 = p + (".post_cond.2" if (p + ".post_cond.2.weight") in sd else ".post_cond.1")
    h = conv_hook(F.gelu(h), sd[conv2 + ".weight"], sd[conv2 + ".bias"], padding=dil, dilation=dil)
    s = ref_cpu.resize(x, scale)
    if (p + ".skip.1.weight") in sd:
        s = conv_hook(s, sd[p + ".skip.1.weight"], sd[p + ".skip.1.bias"])
    out = s + h
    return bf(out) if MODE["res"] else out


ref_cpu.res_block = res_block
m = DiffusionModel("unet", 32)
det_init_(m.state_dict().items())
sd = {k: v.detach() for k, v in m.state_dict().items()}
g = torch.Generator().manual_seed(3)
x = torch.randn(2, 1, 16384, generator=g)
ts = torch.tensor([0.3, 0.9])
with torch.no_grad():
    ref = ref_cpu.unet_predictor(sd, 32, x, ts)
    for op, h1, res, split2 in [(1, 1, 1, 0), (1, 0, 0, 0), (0, 1, 0, 0), (0, 0, 1, 0), (0, 1, 1, 0), (1, 1, 0, 0), (1, 0, 0, 1), (1, 1, 0, 1), (1, 1, 1, 1)]:
        MODE.update(op=bool(op), h1=bool(h1), res=bool(res), split2=bool(split2))
        got = ref_cpu.unet_predictor(sd, 32, x, ts)
        err = ((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
        print(f"operands bf16={op} (activations hi+lo={split2})  h1 bf16={h1}  residual bf16={res}:  eps rel error {err:.3e}", flush=True)
