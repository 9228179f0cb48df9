"""The CPU oracle (oracle/ref_cpu.py) against the golden vectors generated from the reference itself
(oracle/gen_golden.py).  This is what pins the oracle; the GPU tests then compare the HIP path with it."""
import numpy as np
import torch

from oracle import ref_cpu
from vq_voice_swap_amd import _native
from vq_voice_swap_amd.det_init import det_tensor

from util import seeded

torch.set_num_threads(8)


def det_sd(cfg, prefix):
    return {prefix + n: det_tensor(prefix + n, s) for n, s in _native.param_table(cfg)}


def predictor_cfg(base, cond=0, labels=0):
    cfg = _native.Cfg()
    cfg.kind, cfg.base_channels, cfg.in_channels, cfg.out_channels = _native.KIND_PREDICTOR, base, 1, 1
    cfg.cond_channels, cfg.num_labels = cond, labels
    return cfg


def test_resblocks_match_reference(golden, lib_built):
    z = golden("f1_resblocks")
    for name in sorted({k.split(".")[0] for k in z.files}):
        cin, cout, scale, dil, emb, L = z[name + ".spec"]
        cfg = _native.Cfg()
        cfg.kind = _native.KIND_RESBLOCK
        cfg.rb_cin, cfg.rb_cout, cfg.rb_dilation, cfg.rb_emb_channels = int(cin), int(cout), int(dil), int(emb)
        cfg.rb_resize = 0 if scale == 1.0 else (1 if scale < 1.0 else 2)
        prefix = "blk." + name
        sd = {prefix + "." + n: det_tensor(prefix + "." + n, s) for n, s in _native.param_table(cfg)}
        spec = dict(cin=int(cin), cout=int(cout), scale=float(scale), dil=int(dil))
        x = torch.from_numpy(z[name + ".x"])
        e = torch.from_numpy(z[name + ".emb"]) if emb else None
        y = ref_cpu.res_block(x, sd, prefix, spec, e)
        assert torch.equal(y, torch.from_numpy(z[name + ".y"])), name


def test_unet32_forward_matches_reference(golden, lib_built):
    z = golden("f3_unet32_forward")
    sd = det_sd(predictor_cfg(32), "predictor.")
    x = seeded((2, 1, 64000), int(z["x_seed"]))
    probes = {}
    eps = ref_cpu.unet_predictor(sd, 32, x, torch.from_numpy(z["ts"]), probe=lambda n, t: probes.__setitem__(n, t))
    assert torch.equal(eps, torch.from_numpy(z["eps"]))
    names = [str(n) for n in z["probe_names"]]
    for n, v in zip(names, z["probe_vals"]):
        t = probes[n]
        assert abs(t.mean().item() - v[0]) < 1e-5 and abs(t.pow(2).mean().sqrt().item() - v[1]) < 1e-5, n


def test_ddpm_previous_matches_reference(golden):
    z = golden("f5_ddpm_previous")
    for i in range(5):
        t, step = z[f"c{i}.t_step"]
        x, eps, noise = (torch.from_numpy(z[f"c{i}.{k}"]) for k in ("x", "eps", "noise"))
        ts = torch.tensor([t, t], dtype=torch.float32)
        for mode, kw in (("plain", {}), ("sigma_large", dict(sigma_large=True)), ("constrain", dict(constrain=True))):
            y = ref_cpu.ddpm_previous("exp", x, ts, float(step), eps, noise, **kw)
            assert torch.equal(y, torch.from_numpy(z[f"c{i}.{mode}"])), (i, mode)
    y = ref_cpu.ddpm_previous("exp", torch.from_numpy(z["row.x"]), torch.from_numpy(z["row.ts"]), torch.from_numpy(z["row.step"]),
                              torch.from_numpy(z["row.eps"]), torch.from_numpy(z["row.noise"]), constrain=True)
    assert torch.equal(y, torch.from_numpy(z["row.constrain"]))


def test_sampler_10_steps_matches_reference(golden, lib_built):
    z = golden("f6_sampler_unet32")
    sd = det_sd(predictor_cfg(32), "predictor.")
    x_T = seeded((2, 1, 64000), int(z["x_T_seed"]))
    gen = torch.Generator().manual_seed(int(z["noise_seed"]))
    noises = [torch.randn(x_T.shape, generator=gen) for _ in range(10)]
    assert np.allclose([n.double().sum().item() for n in noises], z["s10_constrain.noise_checksum"], atol=1e-6), "noise stream differs"
    trace = []
    x0 = ref_cpu.ddpm_sample("exp", x_T, lambda a, b: ref_cpu.unet_predictor(sd, 32, a, b), 10, noises, constrain=True, trace=trace)
    assert (x0 - torch.from_numpy(z["s10_constrain.x0"])).abs().max().item() <= 1e-5
    assert np.allclose([t.pow(2).mean().sqrt().item() for t in trace], z["s10_constrain.rms_trace"], atol=1e-5)


def test_vqvae_paths_match_reference(golden, lib_built):
    z7, z4, z8 = golden("f7_encoder_vq32"), golden("f4_cond_forward"), golden("f8_vqvae_decode")
    sd = det_sd(predictor_cfg(32, cond=512, labels=5), "predictor.")
    ecfg = _native.Cfg()
    ecfg.kind, ecfg.base_channels, ecfg.in_channels, ecfg.out_channels = _native.KIND_ENCODER, 32, 1, 512
    sd.update(det_sd(ecfg, "encoder."))
    sd["vq.dictionary"] = seeded((512, 512), 77, 0.35)
    wav = seeded((2, 1, 64000), int(z7["wav_seed"]), 0.1).clamp(-1, 1)
    zz = ref_cpu.unet_encoder(sd, 32, wav)
    assert abs(zz.pow(2).mean().sqrt().item() - float(z7["z_rms"])) < 1e-6
    codes = ref_cpu.vq_encode(sd["vq.dictionary"], zz)
    assert torch.equal(codes, torch.from_numpy(z7["codes"]))
    # margin-guaranteed VQ set
    idx_m = torch.from_numpy(z7["margin_idx"])
    zm = ref_cpu.vq_embed(sd["vq.dictionary"], idx_m) + 1e-3 * seeded((2, 512, 250), int(z7["margin_noise_seed"]))
    assert torch.equal(ref_cpu.vq_encode(sd["vq.dictionary"], zm), idx_m)
    # conditional forward
    cond = ref_cpu.vq_embed(sd["vq.dictionary"], torch.from_numpy(z4["codes16"]))
    eps = ref_cpu.unet_predictor(sd, 32, torch.from_numpy(z4["x"]), torch.from_numpy(z4["ts"]), cond=cond, labels=torch.from_numpy(z4["labels"]))
    assert torch.equal(eps, torch.from_numpy(z4["eps"]))
    # 5-step decode
    x_T = seeded((2, 1, 4096), int(z8["x_T_seed"]))
    gen = torch.Generator().manual_seed(int(z8["noise_seed"]))
    noises = [torch.randn(x_T.shape, generator=gen) for _ in range(5)]
    dec = ref_cpu.vqvae_decode(sd, 32, "exp", torch.from_numpy(z8["codes16"]), torch.from_numpy(z8["labels"]), 5, x_T, noises, constrain=True)
    assert (dec - torch.from_numpy(z8["x0"])).abs().max().item() <= 1e-5
    # 50-step decode (F8b: BASELINE config 4's step count)
    z8b = golden("f8b_vqvae_decode50")
    x_T = seeded((2, 1, 4096), int(z8b["x_T_seed"]))
    gen = torch.Generator().manual_seed(int(z8b["noise_seed"]))
    noises = [torch.randn(x_T.shape, generator=gen) for _ in range(50)]
    dec = ref_cpu.vqvae_decode(sd, 32, "exp", torch.from_numpy(z8b["codes16"]), torch.from_numpy(z8b["labels"]), 50, x_T, noises, constrain=True)
    assert (dec - torch.from_numpy(z8b["x0"])).abs().max().item() <= 1e-5


def test_time_embedding_matches_reference(golden, lib_built):
    z = golden("f2_time_embed")
    ts = torch.from_numpy(z["ts"])
    for base in (32, 64):
        sd = det_sd(predictor_cfg(base, labels=6), "predictor.")
        emb = ref_cpu.unet_embedding(sd, ts, torch.from_numpy(z[f"c{base}.labels"]))
        assert torch.equal(emb, torch.from_numpy(z[f"c{base}.emb"])), base


def test_ddpm_previous_cos_matches_reference(golden):
    z = golden("f5b_ddpm_previous_cos")
    for i in range(5):
        t, step = z[f"c{i}.t_step"]
        x, eps, noise = (torch.from_numpy(z[f"c{i}.{k}"]) for k in ("x", "eps", "noise"))
        ts = torch.tensor([t, t], dtype=torch.float32)
        for mode, kw in (("plain", {}), ("sigma_large", dict(sigma_large=True)), ("constrain", dict(constrain=True))):
            y = ref_cpu.ddpm_previous("cos", x, ts, float(step), eps, noise, **kw)
            assert torch.equal(y, torch.from_numpy(z[f"c{i}.{mode}"])), (i, mode)


def test_decode_uncond_guidance_matches_reference(golden, lib_built):
    z = golden("f11_uncond_guidance")
    sd = det_sd(predictor_cfg(32, cond=512, labels=5), "predictor.")
    sd["vq.dictionary"] = seeded((512, 512), 77, 0.35)
    steps = int(z["steps"])
    x_T = seeded((2, 1, 2048), int(z["x_T_seed"]))
    gen = torch.Generator().manual_seed(int(z["noise_seed"]))
    noises = [torch.randn(x_T.shape, generator=gen) for _ in range(steps)]
    vq_scale, label_scale = (float(v) for v in z["scales"])
    dec = ref_cpu.vqvae_decode_uncond_guidance(sd, 32, "exp", torch.from_numpy(z["codes"]), torch.from_numpy(z["labels"]), steps, x_T, noises,
                                               constrain=True, label_scale=label_scale, vq_scale=vq_scale)
    assert (dec - torch.from_numpy(z["x0"])).abs().max().item() <= 1e-5


def test_decode_uncond_guidance_50_steps_matches_reference(golden, lib_built):
    """F11b: the same path at a real step count (50), from the reference's own decode_uncond_guidance (vq_vae.py:147-220)."""
    z = golden("f11b_uncond_guidance_50")
    sd = det_sd(predictor_cfg(32, cond=512, labels=5), "predictor.")
    sd["vq.dictionary"] = seeded((512, 512), 77, 0.35)
    steps = int(z["steps"])
    x_T = seeded((2, 1, 8192), int(z["x_T_seed"]))
    gen = torch.Generator().manual_seed(int(z["noise_seed"]))
    noises = [torch.randn(x_T.shape, generator=gen) for _ in range(steps)]
    vq_scale, label_scale = (float(v) for v in z["scales"])
    dec = ref_cpu.vqvae_decode_uncond_guidance(sd, 32, "exp", torch.from_numpy(z["codes"]), torch.from_numpy(z["labels"]), steps, x_T, noises,
                                               constrain=True, label_scale=label_scale, vq_scale=vq_scale)
    assert (dec - torch.from_numpy(z["x0"])).abs().max().item() <= 1e-5


def test_conv_mfcc_stack_matches_reference(golden, lib_built):
    """F12: everything of ConvMFCCEncoder.forward that is the reference's own code (conv_encoder.py:90-133: invert_ulaw, deltas x 2,
    concatenation order, ResConv, the k = 4 / stride-2 convolution, the output convolution) on an injected MFCC tensor.  The fixture
    was made from the reference with torchaudio.transforms.MFCC stubbed out; the transform itself stays unpinned."""
    from vq_voice_swap_amd.det_init import det_tensor

    z = golden("f12_conv_mfcc_stack")
    cfg = _native.Cfg()
    cfg.kind, cfg.base_channels, cfg.in_channels, cfg.out_channels = _native.KIND_MFCC_ENCODER, 32, 1, 512
    cfg.reserved[1], cfg.reserved[2] = 1, 1
    sd = {"encoder." + n: det_tensor("encoder." + n, s) for n, s in _native.param_table(cfg) if ".mfcc." not in n and not n.startswith("mfcc.")}
    for tag, ulaw in (("ulaw_even", True), ("ulaw_odd", True), ("linear_even", False)):
        x = torch.from_numpy(z[tag + ".x"])
        mf = torch.from_numpy(z[tag + ".mfcc"])
        got = ref_cpu.conv_mfcc_encoder(sd, x, version=1, input_ulaw=ulaw, mfcc_override=mf)
        assert (got - torch.from_numpy(z[tag + ".z"])).abs().max().item() <= 1e-5, tag
        seen = ref_cpu.invert_ulaw(x)[:, 0] if ulaw else x[:, 0]
        assert (seen - torch.from_numpy(z[tag + ".wave_seen_by_mfcc"])).abs().max().item() <= 1e-7, tag
