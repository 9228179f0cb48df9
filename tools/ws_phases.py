"""Where the cycles of the wave-specialised convolution kernel (csrc/conv_ws.hip) go, per role: s_memtime totals from the
instrumented library (`make -C vq_voice_swap_amd/csrc timing` -> libvqvs_timing.so), for single-ResBlock shapes.

    VQVS_LIB_PATH=vq_voice_swap_amd/libvqvs_timing.so python tools/ws_phases.py ["((cin, cout, L, B), ...)" [fp16|bf16|fp32]]
"""
import os
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # kernel arguments in device memory: a process-level HIP switch, before the runtime starts
import ast, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("VQVS_LIB_PATH", os.path.join(ROOT, "vq_voice_swap_amd", "libvqvs_timing.so"))
import torch
from vq_voice_swap_amd import _native
from vq_voice_swap_amd.unet import ResBlockModule
from vq_voice_swap_amd.det_init import det_init_

P_PH = ["wait loads (vmcnt)", "prologue + ds_write", "load cursor (prepare)", "lgkmcnt + barrier", "issue next loads"]
C_PH = ["tile start: store + bias", "weight DMA issue", "ds_read + MFMA", "tile end: stats/round/LDS", "vmcnt + barrier", "last tile's store (after the loop)",
        "look-ahead GroupNorm table"]
L = _native.lib()
L.vqvs_debug_ws_timing.argtypes = [C.c_void_p, C.c_int]
dev = torch.device("cuda:0")


def read(reset=True):
    buf = (C.c_uint64 * 32)()
    _native.check(L.vqvs_debug_ws_timing(buf, 1 if reset else 0))
    return [int(v) for v in buf]


def run(cin, cout, Lx, B, prec="fp16", dil=2, emb=256):
    m = ResBlockModule(cin, emb, cout, 1.0, dil)
    det_init_(m.state_dict().items())
    m.set_precision(prec)
    m.to(dev)
    x = torch.randn(B, cin, Lx, device=dev)
    e = torch.randn(B, emb, device=dev)
    m(x, e)
    read()
    for _ in range(3):
        m(x, e)
    t = read()
    pw, cw, steps, tiles = t[16], t[17], t[18], t[19]
    if pw == 0:
        print(f"--- ResBlock {cin}->{cout} L={Lx}: not on the wave-specialised kernel")
        return
    print(f"--- ResBlock {cin}->{cout} L={Lx} B={B} {prec} d={dil} (both convs): {steps / pw:.0f} steps, {tiles / max(cw, 1):.1f} tiles per sampled wave")
    if t[21]:
        print(f"   clock: {t[20] / t[21] * 100:.0f} MHz (s_memtime ticks per 100 MHz s_memrealtime tick, sampled consumer waves); {t[21] / cw / 100:.1f} us per sampled workgroup")
    if t[22]:
        print(f"   startup (ticks per sampled workgroup and launch): entry -> requests out / tables {t[22] / cw:.0f}, -> first barrier {t[23] / cw:.0f}, "
              f"-> weights landed {t[24] / cw:.0f}, -> loop starts {t[25] / cw:.0f}; whole workgroup {t[20] / cw:.0f}")
    ptot, ctot = sum(t[0:5]), sum(t[8:15])
    print(f"   producers: {ptot / steps:8.0f} ticks per step")
    for name, v in zip(P_PH, t[0:5]):
        print(f"      {name:28s} {v / steps:8.0f}  {100 * v / ptot:5.1f}%")
    if cw == 0:  # (an ablation build whose consumers only keep the barriers)
        return
    csteps = steps * cw / pw
    print(f"   consumers: {ctot / csteps:8.0f} ticks per step")
    for name, v in zip(C_PH, t[8:15]):
        print(f"      {name:28s} {v / csteps:8.0f}  {100 * v / ctot:5.1f}%")


shapes = ast.literal_eval(sys.argv[1]) if len(sys.argv) > 1 else ((64, 64, 64000, 64), (128, 128, 16000, 64), (256, 256, 2000, 64), (512, 512, 250, 64))
prec = sys.argv[2] if len(sys.argv) > 2 else "fp16"
for shape in shapes:
    run(*shape, prec=prec)
