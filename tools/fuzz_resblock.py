"""Randomised ResBlock parity sweep (HIP vs oracle): random channel counts, lengths (incl. tile-boundary cases),
dilations, resizes, FiLM on/off, batch sizes, all three precision modes.  Developer tool; tests/ hold the fixed cases.
    python tools/fuzz_resblock.py [seed] [cases] [big]      ("big": every fourth case is a long, many-clip launch)"""
import os
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # kernel arguments in device memory: a process-level HIP switch, before the runtime starts
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import ref_cpu
from vq_voice_swap_amd.unet import ResBlockModule
from vq_voice_swap_amd.det_init import det_init_
from util import rel_rms, seeded

dev = torch.device("cuda:0")
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
BIG = len(sys.argv) > 3 and sys.argv[3] == "big"
worst = {"fp32": 0.0, "fp16": 0.0, "bf16": 0.0}
bad = 0
for i in range(N):
    cin = rng.choice([32, 64, 96, 128, 192, 256, 384, 512])
    scale = rng.choice([1.0, 1.0, 1.0, 0.5, 2.0])
    cout = cin if scale != 1.0 else rng.choice([cin, 32, 64, 128, 256, 512])
    dil = 2 if scale == 2.0 else rng.choice([1, 2, 2, 4, 8, 16, 32])
    emb = rng.choice([None, 128, 256])
    L = rng.choice([2, 6, 64, 126, 250, 252, 254, 256, 258, 500, 508, 510, 1000, 1024, rng.randrange(2, 1500) * 2])
    B = rng.choice([1, 2, 3, 5])
    if scale == 0.5 and L < 4:  # (one output row per channel and clip: torch's group_norm -- the oracle, and the reference -- refuses it)
        L = 4
    if BIG and i % 4 == 0:  # many tiles per workgroup and workgroups that cross clip boundaries (the (scale, shift) ring)
        L = rng.choice([4000, 6002, 8190, 12000])
        B = rng.choice([24, 37, 48])
        cin = rng.choice([64, 128])
        cout = cin if scale != 1.0 else rng.choice([cin, 64, 128])
    m = ResBlockModule(cin, emb, cout if cout != cin else None, scale, dil)
    det_init_((f"fz{i}." + k, v) for k, v in m.block.state_dict().items())
    x = seeded((B, cin, L), 5000 + i)
    e = seeded((B, emb), 6000 + i) if emb else None
    sd = {"b." + k: v.detach() for k, v in m.block.state_dict().items()}
    want = ref_cpu.res_block(x, sd, "b", dict(cin=cin, cout=cout, scale=scale, dil=dil), e)
    for prec, tol in (("fp32", 2e-4), ("fp16", 4e-3), ("bf16", 3e-2)):
        m.set_precision(prec)
        got = m(x.to(dev), None if e is None else e.to(dev)).cpu()
        err = rel_rms(got, want) if got.shape == want.shape else float("inf")
        worst[prec] = max(worst[prec], err)
        if not err < tol:
            bad += 1
            print("MISMATCH", prec, dict(cin=cin, cout=cout, scale=scale, dil=dil, emb=emb, L=L, B=B), err, flush=True)
print("cases", N, "worst", worst, "bad", bad)
