"""Calibration, never product: what a vendor GEMM (hipBLASLt / rocBLAS behind torch.matmul; fp16 operands, fp32 accumulation) takes
on the GEMM shapes of the deep convolutions (Cout >= 256) of a unet64 forward at 64 clips, next to what conv_ws_kernel takes on the
same shape with EVERYTHING a convolution launch carries (GroupNorm + GELU prologue, three shifted taps of one staged tile, bias,
identity skip, statistics, one rounding).  The GEMM is the convolution's arithmetic alone: A = im2col rows [M = clips * L][K = 3 * Cin]
already materialised and transformed (which the convolution never materialises), B = [K][N = Cout].

    python tools/ubench/gemm_shapes.py            # one JSON object on stdout

Shapes: M x N x K = 16000 x 512 x 1536 (512x3->512 at L/256), 32000 x 512 x 3072 (512x3+512x3->512 at
L/128), 32000 x 256 x 768 (256x3->256 at L/128), 128000 x 256 x 1536 (256x3+256x3->256 at L/32)."""
import json
import os
import sys

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

dev = torch.device("cuda:0")
MFMA_PEAK = 2.5e15  # dense fp16 / bf16, MI355X_MICROARCH.md


def time_us(fn, reps=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def conv_ws_us(Cin_list, Cout, L, B=64, dil=1, skip=False):
    """The library's own convolution on the same GEMM shape, through a ResBlock handle's per-op profile (conv 1 of a block whose
    input is Cin wide -- a concatenation when two widths are given is timed by the unet64 per-op table instead)."""
    from vq_voice_swap_amd.unet import ResBlockModule
    from vq_voice_swap_amd.det_init import det_init_

    cin = sum(Cin_list)
    m = ResBlockModule(cin, 256, Cout if Cout != cin else None, 1.0, 2)
    det_init_(("gs." + k, v) for k, v in m.block.state_dict().items())
    m.set_precision("fp16")
    m.to(dev)
    x = torch.randn(B, cin, L, device=dev)
    e = torch.randn(B, 256, device=dev)
    m(x, e)
    h = m._handle
    h.set_profiling(True)
    acc = None
    reps = 20
    for _ in range(reps):
        m(x, e)
        ms = h.profile_read()
        acc = ms if acc is None else [p + q for p, q in zip(acc, ms)]
    h.set_profiling(False)
    info = h.op_info(B, L)
    convs = [(t / reps * 1e3, fl) for (kind, _by, fl), t in zip(info, acc) if kind == "conv"]
    m.invalidate()
    return convs  # [(us, flops)] of conv 1 (cin x 3 -> Cout) and conv 2 (Cout x 3 d2 + skip -> Cout)


out = {"device": torch.cuda.get_device_name(0), "torch": torch.__version__, "mfma_peak_tflops": MFMA_PEAK / 1e12, "shapes": []}
for (M, N, K, what, cw) in [
    (16000, 512, 1536, "512x3->512 at L/256 (250 rows x 64 clips)", ([512], 512, 250)),
    (32000, 512, 3072, "512x3+512x3->512 at L/128 (500 rows x 64 clips)", ([1024], 512, 500)),
    (32000, 256, 768, "256x3->256 at L/128 (500 rows x 64 clips)", ([256], 256, 500)),
    (128000, 256, 1536, "256x3+256x3->256 at L/32 (2000 rows x 64 clips)", ([512], 256, 2000)),
]:
    a = torch.randn(M, K, device=dev, dtype=torch.float16)
    b = torch.randn(K, N, device=dev, dtype=torch.float16)
    bt = torch.randn(N, K, device=dev, dtype=torch.float16)
    c = torch.empty(M, N, device=dev, dtype=torch.float16)
    nn = time_us(lambda: torch.matmul(a, b, out=c))
    nt = time_us(lambda: torch.matmul(a, bt.t(), out=c))
    fl = 2.0 * M * N * K
    best = min(nn, nt)
    rec = {"M": M, "N": N, "K": K, "conv": what, "vendor_gemm_us": {"nn": round(nn, 2), "nt": round(nt, 2)},
           "vendor_gemm_tflops": round(fl / best / 1e6, 1), "vendor_gemm_frac_of_mfma_peak": round(fl / (best * 1e-6) / MFMA_PEAK, 3),
           "bytes_MB": round((M * K + K * N + M * N) * 2 / 1e6, 1)}
    try:
        cin, cout, L = cw
        convs = conv_ws_us(cin, cout, L)
        rec["conv_ws_us"] = [round(u, 2) for u, _ in convs]
        rec["conv_ws_tflops"] = [round(f / u / 1e6, 1) for u, f in convs]
        rec["conv_ws_note"] = ("ResBlock(%d -> %d) at L = %d, 64 clips, fp16: conv 1 = %d x 3 -> %d with the GroupNorm + GELU prologue "
                               "(the GEMM above when the widths match), conv 2 = %d x 3 d2 (+ 1x1 skip or identity) -> %d; bracketed "
                               "per-op times (+ ~3.5 us of bracket overhead each)" % (sum(cin), cout, L, sum(cin), cout, cout, cout))
    except Exception as e:  # noqa: BLE001
        rec["conv_ws_error"] = repr(e)[:300]
    out["shapes"].append(rec)
    del a, b, bt, c
print(json.dumps(out, indent=1))
