"""Classifier guidance (BASELINE config 5, scope row G1 / 8f.1): parameter layout against the golden vectors of the
reference (CPU); native logits, native input gradient and guided DDPM sampling against the oracle (GPU)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_cpu
from vq_voice_swap_amd import Classifier, DiffusionModel
from vq_voice_swap_amd.det_init import det_init_

from util import gate, rel_rms, rms, seeded


def make_classifier():
    clf = Classifier(num_labels=7, base_channels=32)
    det_init_(clf.state_dict().items())
    clf.eval()
    return clf


def test_classifier_matches_reference_golden(golden):
    z = golden("f9_classifier32")
    clf = make_classifier()
    assert len(clf.state_dict()) == 290 and "stem.out.1.qkv_proj.weight" in clf.state_dict()
    assert clf.save_kwargs()["channel_mult"] == (1, 1, 2, 2, 2, 4, 4, 8, 8)
    x = seeded((2, 1, 64000), int(z["x_seed"]))
    ts, labels = torch.from_numpy(z["ts"]), torch.from_numpy(z["labels"])
    sd = {k: v.detach() for k, v in clf.state_dict().items()}
    # oracle pinned by the reference's outputs
    assert torch.equal(ref_cpu.classifier(sd, 32, x, ts), torch.from_numpy(z["logits"]))
    assert torch.equal(ref_cpu.classifier_cond_fn(sd, 32, labels)(x, ts), torch.from_numpy(z["grad"]))
    # the sampling-path entry points have no CPU fallback
    with pytest.raises(RuntimeError):
        clf(x, ts)
    with pytest.raises(RuntimeError):
        clf.guidance_fn(labels, 2.0)(x, ts)


@pytest.mark.gpu
@pytest.mark.parametrize("precision,tol_logit,tol_grad", [("fp32", 2e-4, 2e-3), ("fp16", 8e-3, 3e-2), ("bf16", 5e-2, 1.5e-1)])
def test_native_classifier_vs_golden(golden, precision, tol_logit, tol_grad):
    """logits and d log p(y|x)/dx of the HIP path against the reference's own outputs (fixture F9)."""
    z = golden("f9_classifier32")
    dev = torch.device("cuda:0")
    clf = make_classifier().to(dev)
    clf.set_precision(precision)
    x = seeded((2, 1, 64000), int(z["x_seed"])).to(dev)
    ts, labels = torch.from_numpy(z["ts"]).to(dev), torch.from_numpy(z["labels"]).to(dev)
    want_logits, want_grad = torch.from_numpy(z["logits"]), torch.from_numpy(z["grad"])
    logits = clf(x, ts).cpu()
    assert rel_rms(logits, want_logits) < tol_logit
    grad, logits2 = clf.log_prob_grad(x, ts, labels, 1.0, return_logits=True)
    assert torch.equal(logits2.cpu(), logits)
    assert rel_rms(grad.cpu(), want_grad) < tol_grad
    # linear in the scale, deterministic
    g3 = clf.guidance_fn(labels, 3.0)(x, ts)
    assert rel_rms(g3.cpu(), 3.0 * grad.cpu()) < (1e-4 if precision == "fp32" else 4e-2)  # 3x changes the operand rounding
    assert torch.equal(clf.log_prob_grad(x, ts, labels, 1.0), grad)


def test_load_from_predictor_copies_the_shared_stem():
    """classifier.py:123-131: in_conv, the time embedding and the blocks that line up with the predictor's down path."""
    m = DiffusionModel("unet", 32)
    det_init_(m.state_dict().items())
    clf = Classifier(num_labels=3, base_channels=32)
    n = clf.stem.load_from_predictor(m.predictor)
    assert n == sum(int(v.numel()) for mod in (m.predictor.in_conv, m.predictor.time_embed, m.predictor.time_embed_extra, *m.predictor.down_blocks)
                    for v in mod.state_dict().values())
    assert torch.equal(clf.stem.blocks[25].post_cond[1].weight, m.predictor.down_blocks[25].post_cond[1].weight)
    assert torch.equal(clf.stem.time_embed.proj.weight, m.predictor.time_embed.proj.weight)


CUSTOM = [("c_a", dict(channel_mult=(1, 2, 2, 4), output_mult=8, depth_mult=1), 4096),
          ("c_b", dict(channel_mult=(1, 1, 2, 2, 2, 4), output_mult=4, depth_mult=3), 8192)]


def test_custom_classifier_topology_param_table(lib_built):
    """classifier.py:52-58: any channel_mult / output_mult / depth_mult -- the library's parameter table is the module's state dict."""
    from vq_voice_swap_amd import _native

    for tag, kw, T in CUSTOM:
        clf = Classifier(num_labels=5, base_channels=32, **kw)
        sd = clf.state_dict()
        table = _native.param_table(clf._cfg())
        assert len(table) == len(sd) and set(n for n, _ in table) == set(sd.keys()), tag
        assert all(tuple(sd[n].shape) == s for n, s in table)
        assert clf.downsample_rate == 2 ** len(kw["channel_mult"]) and clf.save_kwargs()["output_mult"] == kw["output_mult"]
    assert Classifier(num_labels=5, base_channels=32)._cfg().topology_set == 0


@pytest.mark.gpu
@pytest.mark.parametrize("precision,tol_logit,tol_grad", [("fp32", 2e-4, 2e-3), ("fp16", 8e-3, 3e-2)])
def test_custom_classifier_topology_vs_reference_fixture(golden, precision, tol_logit, tol_grad):
    """F15: classifiers of non-default topology built by the reference's own constructor: logits and d log p(y|x)/dx."""
    z = golden("f15_custom_classifiers")
    dev = torch.device("cuda:0")
    for tag, kw, T in CUSTOM:
        clf = Classifier(num_labels=5, base_channels=32, **kw)
        det_init_(("clf." + tag + "." + k, v) for k, v in clf.state_dict().items())
        clf.eval().to(dev)
        clf.set_precision(precision)
        x, ts, labels = (torch.from_numpy(z[f"{tag}.{k}"]).to(dev) for k in ("x", "ts", "labels"))
        grad, logits = clf.log_prob_grad(x, ts, labels, 1.0, return_logits=True)
        assert rel_rms(logits.cpu(), torch.from_numpy(z[tag + ".logits"])) < tol_logit, (tag, precision)
        assert rel_rms(grad.cpu(), torch.from_numpy(z[tag + ".grad"])) < tol_grad, (tag, precision, rel_rms(grad.cpu(), torch.from_numpy(z[tag + ".grad"])))
        with pytest.raises(ValueError, match="downsample rate"):
            clf(torch.zeros(1, 1, clf.downsample_rate * 3 + 2, device=dev), torch.zeros(1, device=dev))


@pytest.mark.gpu
def test_native_classifier_shapes_and_errors():
    dev = torch.device("cuda:0")
    clf = make_classifier().to(dev)
    sd = {k: v.detach().cpu() for k, v in clf.state_dict().items()}
    # a short ragged batch: 3 clips of 1536 samples (3 tokens), per-clip timesteps
    x = seeded((3, 1, 1536), 5)
    ts = torch.tensor([0.1, 0.5, 0.95])
    labels = torch.tensor([0, 6, 2])
    want = ref_cpu.classifier(sd, 32, x, ts)
    want_g = ref_cpu.classifier_cond_fn(sd, 32, labels, 1.0)(x, ts)
    got_g, got = clf.log_prob_grad(x.to(dev), ts.to(dev), labels.to(dev), 1.0, return_logits=True)
    assert rel_rms(got.cpu(), want) < 2e-4
    assert rel_rms(got_g.cpu(), want_g) < 2e-3
    with pytest.raises(ValueError):
        clf(torch.zeros(1, 1, 1000, device=dev), torch.zeros(1, device=dev))
    with pytest.raises(ValueError):
        clf.log_prob_grad(x.to(dev), ts.to(dev), labels[:2].to(dev))


def test_classifier_checkpoint_roundtrip(tmp_path):
    clf = make_classifier()
    p = str(tmp_path / "c.pt")
    clf.save(p)
    clf2 = Classifier.load(p)
    assert all(torch.equal(a, b) for a, b in zip(clf.state_dict().values(), clf2.state_dict().values()))


@pytest.mark.gpu
def test_guided_sampling_vs_oracle():
    dev = torch.device("cuda:0")
    model = DiffusionModel("unet", 32)
    det_init_(model.state_dict().items())
    model.eval()
    clf = make_classifier()
    sd_m = {k: v.detach().clone() for k, v in model.state_dict().items()}
    sd_c = {k: v.detach().clone() for k, v in clf.state_dict().items()}
    labels = torch.tensor([3, 5])
    scale = 200.0  # large enough that the guidance term visibly moves the sample
    x_T = seeded((2, 1, 8192), 71)
    gen = torch.Generator().manual_seed(72)
    noises = [torch.randn(x_T.shape, generator=gen) for _ in range(4)]
    want = ref_cpu.ddpm_sample("exp", x_T, lambda a, b: ref_cpu.unet_predictor(sd_m, 32, a, b), 4, noises, constrain=True,
                               cond_fn=ref_cpu.classifier_cond_fn(sd_c, 32, labels, scale))
    plain = ref_cpu.ddpm_sample("exp", x_T, lambda a, b: ref_cpu.unet_predictor(sd_m, 32, a, b), 4, noises, constrain=True)
    clf.to(dev)
    got = model.diffusion.ddpm_sample(x_T.to(dev), model.predictor, 4, constrain=True, cond_fn=clf.guidance_fn(labels.to(dev), scale),
                                      noise=[n.to(dev) for n in noises]).cpu()
    gate("classifier-guided unet32 4 steps vs oracle (fp32)", got, want, 1e-3)
    assert rms(want - plain) > 10 * rms(got - want), "guidance term too small for the comparison to mean anything"


@pytest.mark.gpu
def test_guided_sampling_with_time_remap_and_sigma_large():
    """cond_fn together with a sample-time schedule (diffusion.py:116-118) and sigma_large.
    (Not with the cos noise schedule: its first iteration has alpha_bar(1) = cos^2(pi/2) ~ 2e-15, the constrain step then
    subtracts a float32 mean from values of magnitude 1e7 and the reference's own result depends on its summation order
    -- DESIGN.md section 4.)"""
    dev = torch.device("cuda:0")
    model = DiffusionModel("unet", 32, schedule_name="exp")
    det_init_(model.state_dict().items())
    model.eval()
    clf = make_classifier()
    sd_m = {k: v.detach().clone() for k, v in model.state_dict().items()}
    sd_c = {k: v.detach().clone() for k, v in clf.state_dict().items()}
    labels = torch.tensor([0, 6, 2])
    x_T = seeded((3, 1, 4096), 91)
    gen = torch.Generator().manual_seed(92)
    noises = [torch.randn(x_T.shape, generator=gen) for _ in range(3)]
    tmap = lambda t: t ** 2  # noqa: E731
    want = ref_cpu.ddpm_sample("exp", x_T, lambda a, b: ref_cpu.unet_predictor(sd_m, 32, a, b), 3, noises, sigma_large=True,
                               constrain=True, cond_fn=ref_cpu.classifier_cond_fn(sd_c, 32, labels, 50.0), t_map=tmap)
    clf.to(dev)
    got = model.diffusion.ddpm_sample(x_T.to(dev), model.predictor, 3, sigma_large=True, constrain=True, schedule=tmap,
                                      cond_fn=clf.guidance_fn(labels.to(dev), 50.0), noise=[n.to(dev) for n in noises]).cpu()
    gate("classifier-guided unet32 3 steps, sigma_large, t**2 vs oracle (fp32)", got, want, 1e-3)


@pytest.mark.gpu
def test_guided_sampling_ten_steps_in_both_gate_modes():
    """BASELINE config 5's path (classifier gradient at every step) held to the waveform gate in the benchmarked fp16 mode
    too: unet32 + classifier32, 10 steps, constrained; predictor AND classifier in the mode under test."""
    dev = torch.device("cuda:0")
    model = DiffusionModel("unet", 32)
    det_init_(model.state_dict().items())
    model.eval()
    clf = make_classifier()
    sd_m = {k: v.detach().clone() for k, v in model.state_dict().items()}
    sd_c = {k: v.detach().clone() for k, v in clf.state_dict().items()}
    labels = torch.tensor([3, 5])
    scale, steps = 200.0, 10
    x_T = seeded((2, 1, 8192), 171)
    gen = torch.Generator().manual_seed(172)
    noises = [torch.randn(x_T.shape, generator=gen) for _ in range(steps)]
    want = ref_cpu.ddpm_sample("exp", x_T, lambda a, b: ref_cpu.unet_predictor(sd_m, 32, a, b), steps, noises, constrain=True,
                               cond_fn=ref_cpu.classifier_cond_fn(sd_c, 32, labels, scale))
    plain = ref_cpu.ddpm_sample("exp", x_T, lambda a, b: ref_cpu.unet_predictor(sd_m, 32, a, b), steps, noises, constrain=True)
    clf.to(dev)
    errs = {}
    for prec in ("fp32", "fp16"):
        model.set_precision(prec)
        clf.set_precision(prec)
        got = model.diffusion.ddpm_sample(x_T.to(dev), model.predictor, steps, constrain=True, cond_fn=clf.guidance_fn(labels.to(dev), scale),
                                          noise=[n.to(dev) for n in noises]).cpu()
        errs[prec] = gate(f"classifier-guided unet32 + classifier32, 10 steps vs oracle {prec}", got, want, 1e-3)
    assert rms(want - plain) > 5 * max(errs.values()), ("guidance term too small for the comparison to mean anything", errs, rms(want - plain))


@pytest.mark.gpu
def test_guidance_gradient_is_batch_independent_at_full_size():
    """BASELINE config 5's classifier at its full size (classifier32, B = 32 clips of T = 64000, fp16 mode): the resident backward
    arena of a 32-clip call must give clips 0 / 17 / 31 BITWISE the gradient of a 1-clip call (sample_diffusion.py:34-42,
    models/classifier.py:111-121 are per clip: nothing mixes clips)."""
    dev = torch.device("cuda:0")
    clf = Classifier(num_labels=251, base_channels=32)
    det_init_(clf.state_dict().items())
    clf.eval()
    clf.set_precision("fp16")
    x = seeded((32, 1, 64000), 41).to(dev)
    ts = torch.linspace(0.05, 0.95, 32).to(dev)
    labels = (torch.arange(32) * 7 % 251).to(dev)
    g_all, logits_all = clf.log_prob_grad(x, ts, labels, scale=1.0, return_logits=True)
    assert g_all.shape == x.shape and bool(torch.isfinite(g_all).all())
    for i in (0, 17, 31):
        g1, l1 = clf.log_prob_grad(x[i:i + 1], ts[i:i + 1], labels[i:i + 1], scale=1.0, return_logits=True)
        assert torch.equal(g1[0], g_all[i]) and torch.equal(l1[0], logits_all[i]), i


@pytest.mark.gpu
def test_classifier_width_96_vs_oracle():
    """The reference takes any base width (classifier.py:52-58); the library builds every multiple of 32 (round 6: the guidance
    models no longer need a power of two -- in_conv_bw / bw_act have forms for rows whose octet count is not one).  base 96 with a
    short topology (final width 384 = 6 heads of 64) and base 160 (final width 320 = 5 heads): logits and d log p / dx."""
    dev = torch.device("cuda:0")
    for base, kw, T in ((96, dict(channel_mult=(1, 2, 4), output_mult=2, depth_mult=1), 2048),
                        (160, dict(channel_mult=(1, 2), output_mult=2, depth_mult=2), 1024)):
        clf = Classifier(num_labels=5, base_channels=base, **kw)
        det_init_((f"clf.w{base}." + k, v) for k, v in clf.state_dict().items())
        clf.eval()
        sd = {k: v.detach().clone() for k, v in clf.state_dict().items()}
        x, ts, labels = seeded((2, 1, T), 300 + base), torch.tensor([0.2, 0.7]), torch.tensor([1, 4])
        topo = dict(channel_mult=kw["channel_mult"], depth_mult=kw["depth_mult"])
        want = ref_cpu.classifier(sd, base, x, ts, topology=topo)
        want_g = ref_cpu.classifier_cond_fn(sd, base, labels, 1.0, topology=topo)(x, ts)
        clf.to(dev)
        for prec, tl, tg in (("fp32", 2e-4, 2e-3), ("fp16", 8e-3, 3e-2)):
            clf.set_precision(prec)
            g, lg = clf.log_prob_grad(x.to(dev), ts.to(dev), labels.to(dev), 1.0, return_logits=True)
            assert rel_rms(lg.cpu(), want) < tl, (base, prec, rel_rms(lg.cpu(), want))
            assert rel_rms(g.cpu(), want_g) < tg, (base, prec, rel_rms(g.cpu(), want_g))
