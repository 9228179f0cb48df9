"""Where the cycles of the convolution kernel go: per-phase s_memtime totals from the instrumented library
(`make -C vq_voice_swap_amd/csrc timing` -> libvqvs_timing.so), for single-ResBlock shapes.

    VQVS_LIB_PATH=vq_voice_swap_amd/libvqvs_timing.so python tools/conv_phases.py
"""
import os
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # kernel arguments in device memory: a process-level HIP switch, before the runtime starts
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("VQVS_LIB_PATH", os.path.join(ROOT, "vq_voice_swap_amd", "libvqvs_timing.so"))
import torch
from vq_voice_swap_amd import _native
from vq_voice_swap_amd.unet import ResBlockModule
from vq_voice_swap_amd.det_init import det_init_

PH = ["setup", "issue/loop", "wait loads", "prologue+LDS wr", "barrier", "LDS rd+MFMA", "epi barrier", "acc->LDS", "epi barrier2",
      "row phase", "stats reduce", "  issue: ss loads", "  issue: act loads", "  issue: weight loads", "  loop/geom", "-", "back-edge", "matrix-pipe drain"]
L = _native.lib()
L.vqvs_debug_conv_timing.argtypes = [C.c_void_p, C.c_int]
dev = torch.device("cuda:0")


def read(reset=True):
    buf = (C.c_uint64 * 24)()
    _native.check(L.vqvs_debug_conv_timing(buf, 1 if reset else 0))
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
    m(x, e)
    t1 = read()
    print(f"    one ResBlock (two conv launches incl. the gap between them): workgroup starts spread {t1[18] / 100:.1f} us, "
          f"ends spread {t1[19] / 100:.1f} us, first start -> last end {t1[20] / 100:.1f} us, first start -> first end {t1[21] / 100:.1f} us")
    for _ in range(3):
        m(x, e)
    t = read()
    waves = t[23]
    tot = sum(t[:18])
    # NOTE: an s_memtime tick is NOT a shader cycle here: 12.2k ticks = 9.7 us = 19.4k SQ_WAVE_CYCLES for a 64->64 wave-tile,
    # i.e. ~1.26 ticks per ns (~0.63 shader cycles per tick at 2.0 GHz).  Percentages are what this table is for.
    print(f"--- ResBlock {cin}->{cout} L={Lx} B={B} {prec} d={dil}: {waves} waves, {tot / waves:.0f} ticks/wave (both convs)")
    for name, v in zip(PH, t[:18]):
        print(f"   {name:18s} {v / waves:9.0f}  {100 * v / tot:5.1f}%")


import ast
shapes = ast.literal_eval(sys.argv[1]) if len(sys.argv) > 1 else ((64, 64, 64000, 64), (128, 128, 16000, 64), (256, 256, 2000, 64), (512, 512, 250, 64))
for shape in shapes:
    run(*shape)
