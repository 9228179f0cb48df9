"""Developer diagnostic: run every golden case through the HIP path on cuda:0 and print error tables
(keeps going on failure).  Not a test; tests/ hold the asserted versions."""
import os
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # kernel arguments in device memory: a process-level HIP switch, before the runtime starts
import os, sys, time, traceback
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_cpu
from vq_voice_swap_amd import DiffusionModel, VQVAE, ResBlockModule, Diffusion, make_schedule, VQ
from vq_voice_swap_amd.det_init import det_init_

G = os.path.join(ROOT, "tests", "golden")
dev = torch.device("cuda:0")


def seeded(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


def rel_rms(a, b):
    return ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-12)).item()


def section(name):
    print("\n==== " + name, flush=True)


def run(fn):
    try:
        fn()
    except Exception:
        traceback.print_exc()


def resblocks():
    section("F1 resblocks")
    z = np.load(os.path.join(G, "f1_resblocks.npz"))
    names = sorted({k.split(".")[0] for k in z.files})
    for name in names:
        cin, cout, scale, dil, emb, L = z[name + ".spec"]
        cin, cout, dil, emb = int(cin), int(cout), int(dil), int(emb)
        for prec in ("fp32", "bf16"):
            try:
                m = ResBlockModule(cin, emb or None, cout if cout != cin else None, float(scale), dil)
                det_init_(("blk." + name + "." + k, v) for k, v in m.block.state_dict().items())
                m.set_precision(prec)
                x = torch.from_numpy(z[name + ".x"]).to(dev)
                e = torch.from_numpy(z[name + ".emb"]).to(dev) if emb else None
                y = m(x, e).cpu()
                ref = torch.from_numpy(z[name + ".y"])
                print(f"  {name:12s} {prec}: rel_rms={rel_rms(y, ref):.3e} max|d|={(y-ref).abs().max().item():.3e}", flush=True)
            except Exception:
                traceback.print_exc()


def unet32():
    section("F3 unet32 forward (taps vs oracle)")
    z = np.load(os.path.join(G, "f3_unet32_forward.npz"))
    x = seeded((2, 1, 64000), int(z["x_seed"]))
    ts = torch.from_numpy(z["ts"])
    ref = torch.from_numpy(z["eps"])
    for prec in ("fp32", "bf16"):
        model = DiffusionModel("unet", 32)
        det_init_(model.state_dict().items())
        model.set_precision(prec)
        model.predictor.debug_taps = True
        t0 = time.time()
        eps = model.predictor(x.to(dev), ts.to(dev)).cpu()
        print(f"  {prec}: eps rel_rms={rel_rms(eps, ref):.3e} max|d|={(eps-ref).abs().max().item():.3e} ({time.time()-t0:.1f}s incl. build)", flush=True)
        if prec == "fp32" or rel_rms(eps, ref) > 0.05:
            sd = {("predictor." + k): v for k, v in model.predictor.state_dict().items()}
            oracle_taps = {}
            ref_cpu.unet_predictor(sd, 32, x, ts, probe=lambda n, t: oracle_taps.__setitem__(n, t))
            h = model.predictor._handle
            for i, (name, ch, ls) in enumerate(h.taps()):
                got = h.read_tap(i, 2, 64000)
                want = oracle_taps[name]
                r = rel_rms(got, want)
                flag = "" if r < (1e-4 if prec == "fp32" else 3e-2) else "   <-- BAD"
                print(f"    tap {name:18s} C={ch:4d} L={want.shape[-1]:6d} rel_rms={r:.3e}{flag}", flush=True)


def ddpm():
    section("F5 ddpm_previous")
    z = np.load(os.path.join(G, "f5_ddpm_previous.npz"))
    d = Diffusion(make_schedule("exp"))
    for i in range(5):
        t, step = z[f"c{i}.t_step"]
        x, eps, noise = (torch.from_numpy(z[f"c{i}.{k}"]).to(dev) for k in ("x", "eps", "noise"))
        ts = torch.tensor([t, t], dtype=torch.float32, device=dev)
        for mode, kw in (("plain", {}), ("sigma_large", dict(sigma_large=True)), ("constrain", dict(constrain=True))):
            y = d.ddpm_previous(x, ts, float(step), eps, noise=noise, **kw).cpu()
            ref = torch.from_numpy(z[f"c{i}.{mode}"])
            print(f"  t={t} step={step} {mode:11s}: max|d|={(y-ref).abs().max().item():.3e} (max|ref|={ref.abs().max().item():.2e})", flush=True)
    x, eps, noise = (torch.from_numpy(z[f"row.{k}"]).to(dev) for k in ("x", "eps", "noise"))
    y = d.ddpm_previous(x, torch.from_numpy(z["row.ts"]).to(dev), torch.from_numpy(z["row.step"]).to(dev), eps, noise=noise, constrain=True).cpu()
    ref = torch.from_numpy(z["row.constrain"])
    print(f"  per-row constrain: max|d|={(y-ref).abs().max().item():.3e}")


def vqvae():
    section("F7/F4/F8 VQ-VAE")
    z7 = np.load(os.path.join(G, "f7_encoder_vq32.npz"))
    model = VQVAE(base_channels=32, pred_name="unet", num_labels=5)
    det_init_(model.state_dict().items())
    with torch.no_grad():
        model.vq.dictionary.copy_(seeded(model.vq.dictionary.shape, 77, 0.35))
    model.eval()
    wav = seeded((2, 1, 64000), int(z7["wav_seed"]), 0.1).clamp(-1, 1)
    zref = torch.from_numpy(z7["z"]).float()
    codes_ref = torch.from_numpy(z7["codes"])
    for prec in ("fp32", "bf16"):
        model.set_precision(prec)
        zz = model.encoder(wav.to(dev))
        codes = model.vq.encode(zz).cpu()
        mism = (codes != codes_ref)
        print(f"  {prec}: encoder z rel_rms={rel_rms(zz.cpu(), zref):.3e}; code mismatches {int(mism.sum())}/{codes.numel()}; "
              f"gaps at mismatches {torch.from_numpy(z7['gap'])[mism].tolist()[:8]}", flush=True)
    # VQ kernel alone on the golden z (fp16-rounded z -> compare with oracle on the same input)
    zin = zref
    want = ref_cpu.vq_encode(model.vq.dictionary.detach(), zin)
    got = model.vq.encode(zin.to(dev)).cpu()
    print(f"  vq_argmin on golden z: mismatches {(got != want).sum().item()}/{want.numel()}")
    idx_m = torch.from_numpy(z7["margin_idx"])
    zm = ref_cpu.vq_embed(model.vq.dictionary.detach(), idx_m) + 1e-3 * seeded((2, 512, 250), int(z7["margin_noise_seed"]))
    got = model.vq.encode(zm.to(dev)).cpu()
    emb = model.vq.embed(idx_m.to(dev)).cpu()
    print(f"  margin set: mismatches {(got != idx_m).sum().item()}; embed exact={torch.equal(emb, ref_cpu.vq_embed(model.vq.dictionary.detach(), idx_m))}")
    z4 = np.load(os.path.join(G, "f4_cond_forward.npz"))
    model.set_precision("fp32")
    cond = model.vq.embed(torch.from_numpy(z4["codes16"]).to(dev))
    eps = model.predictor(torch.from_numpy(z4["x"]).to(dev), torch.from_numpy(z4["ts"]).to(dev), cond=cond,
                          labels=torch.from_numpy(z4["labels"]).to(dev)).cpu()
    ref = torch.from_numpy(z4["eps"])
    print(f"  cond+labels forward fp32: rel_rms={rel_rms(eps, ref):.3e}")
    z8 = np.load(os.path.join(G, "f8_vqvae_decode.npz"))
    x_T = seeded((2, 1, 4096), int(z8["x_T_seed"]))
    gen = torch.Generator().manual_seed(int(z8["noise_seed"]))
    noises = [torch.randn(x_T.shape, generator=gen).to(dev) for _ in range(5)]
    dec = model.decode(torch.from_numpy(z8["codes16"]).to(dev), torch.from_numpy(z8["labels"]).to(dev), steps=5, constrain=True,
                       x_T=x_T.to(dev), noise=noises).cpu()
    ref = torch.from_numpy(z8["x0"])
    print(f"  decode 5 steps fp32: rms diff={(dec-ref).pow(2).mean().sqrt().item():.3e}")


def sampler():
    section("F6 sampler unet32")
    z = np.load(os.path.join(G, "f6_sampler_unet32.npz"))
    model = DiffusionModel("unet", 32)
    det_init_(model.state_dict().items())
    model.eval()
    x_T = seeded((2, 1, 64000), int(z["x_T_seed"]))
    for tag, steps, constrain, tmap in (("s10_plain", 10, False, None), ("s10_constrain", 10, True, None),
                                        ("s50_sq_constrain", 50, True, (lambda t: t ** 2))):
        gen = torch.Generator().manual_seed(int(z["noise_seed"]))
        noises = [torch.randn(x_T.shape, generator=gen) for _ in range(steps)]
        ck = np.array([n.double().sum().item() for n in noises])
        ok = np.allclose(ck, z[tag + ".noise_checksum"], rtol=0, atol=1e-6)
        for prec in ("fp32", "bf16"):
            model.set_precision(prec)
            t0 = time.time()
            x0 = model.diffusion.ddpm_sample(x_T.to(dev), model.predictor, steps, constrain=constrain, schedule=tmap,
                                             noise=[n.to(dev) for n in noises]).cpu()
            ref = torch.from_numpy(z[tag + ".x0"])
            print(f"  {tag} {prec}: noise_ok={ok} rms diff={(x0-ref).pow(2).mean().sqrt().item():.3e} rel={rel_rms(x0, ref):.3e} "
                  f"(ref rms {ref.pow(2).mean().sqrt().item():.3f}) {time.time()-t0:.1f}s", flush=True)


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0), flush=True)
    which = sys.argv[1:] or ["resblocks", "unet32", "ddpm", "vqvae", "sampler"]
    for w in which:
        run(globals()[w])
