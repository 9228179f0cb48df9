"""Parity at the benchmarked sizes (BASELINE.json configs 3 and 4), on a real MI355X, through the C ABI.

* unet64 at B = 64, T = 64000 -- the bench.py workload: arena offsets, the XCD-aware tile remap and the 32-bit buffer
  offsets of a full batch -- checked through a size-independent property (a clip's result does not depend on the batch it
  sits in: clips 0 / 31 / 63 of the 64-clip forward are BITWISE those of 1-clip forwards) plus the oracle on one clip;
* a unet64 10-step constrained sample against the oracle (<= 1e-3 waveform RMS) in both gate modes;
* BASELINE config 4 at base_channels = 64: UNetEncoder(64) (1024-channel output conv), VQ with Cd = 1024 (bit-exact,
  margin-guaranteed and exact-tie sets), and the cond_proj 1024 -> 64 conditional forward."""
import pytest
import torch

from oracle import ref_cpu
from vq_voice_swap_amd import DiffusionModel, VQVAE
from vq_voice_swap_amd.det_init import det_init_

from util import gate, rel_rms, rms, seeded

pytestmark = pytest.mark.gpu
torch.set_num_threads(8)  # same as every other test module: bit-exact CPU checks depend on the thread count

GATE = {"fp32": 1e-4, "fp16": 4e-3}  # per-forward relative RMS on eps (measured 1.5e-5 / 1.3e-3)
WAVE_RMS = 1e-3


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def det_model(m):
    det_init_(m.state_dict().items())
    m.eval()
    return m


@pytest.fixture(scope="module")
def unet64():
    model = det_model(DiffusionModel("unet", 64))
    sd = {"predictor." + k: v.detach().clone() for k, v in model.predictor.state_dict().items()}
    return model, sd


@pytest.mark.parametrize("prec", ["fp32", "fp16"])
def test_unet64_bench_batch_forward(dev, unet64, prec):
    model, sd = unet64
    model.set_precision(prec)
    B, T = 64, 64000
    x = torch.randn(B, 1, T, generator=torch.Generator().manual_seed(123))
    ts = torch.linspace(0.05, 0.95, B)
    xd, tsd = x.to(dev), ts.to(dev)
    full = model.predictor(xd, tsd)
    assert bool(torch.isfinite(full).all())
    for i in (0, 31, 63):
        one = model.predictor(xd[i:i + 1].contiguous(), tsd[i:i + 1].contiguous())
        assert torch.equal(full[i:i + 1], one), f"clip {i} of the 64-clip batch differs from its 1-clip forward ({prec})"
    want = ref_cpu.unet_predictor(sd, 64, x[31:32], ts[31:32])
    assert rel_rms(full[31:32].cpu(), want) < GATE[prec], prec
    model.predictor.invalidate()  # give the 64-clip arena back


def test_unet64_ten_step_sample_vs_oracle(dev, unet64):
    model, sd = unet64
    x_T = seeded((2, 1, 64000), 21)
    gen = torch.Generator().manual_seed(22)
    noises = [torch.randn(x_T.shape, generator=gen) for _ in range(10)]
    want = ref_cpu.ddpm_sample("exp", x_T, lambda p, q: ref_cpu.unet_predictor(sd, 64, p, q), 10, noises, constrain=True)
    for prec in ("fp32", "fp16"):
        model.set_precision(prec)
        got = model.diffusion.ddpm_sample(x_T.to(dev), model.predictor, 10, constrain=True, noise=[n.to(dev) for n in noises]).cpu()
        gate(f"unet64 10-step constrained sample vs oracle {prec}", got, want, WAVE_RMS)
    model.predictor.invalidate()


def test_config4_base64_encoder_vq_and_cond_forward(dev):
    model = det_model(VQVAE(base_channels=64, pred_name="unet", num_labels=7))
    with torch.no_grad():
        model.vq.dictionary.copy_(seeded(model.vq.dictionary.shape, 78, 0.35))
    assert model.vq.dictionary.shape == (512, 1024)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    dic = sd["vq.dictionary"]
    # UNetEncoder(64): z against the oracle, codes bit-exact
    wav = seeded((2, 1, 64000), 31, 0.1).clamp(-1, 1)
    z_want = ref_cpu.unet_encoder(sd, 64, wav)
    z_got = model.encoder(wav.to(dev)).cpu()
    assert z_got.shape == (2, 1024, 250) and rel_rms(z_got, z_want) < GATE["fp32"]
    codes_want = ref_cpu.vq_encode(dic, z_want)
    codes = model.encode(wav.to(dev)).cpu()
    d = ref_cpu.vq_distances(dic, z_want.permute(0, 2, 1).reshape(-1, 1024))
    top2 = torch.topk(d, 2, dim=-1, largest=False).values
    gap = (top2[:, 1] - top2[:, 0]).reshape(2, -1)
    mism = codes != codes_want
    assert mism.sum().item() == 0, f"{int(mism.sum())} codes differ; top-2 gaps there: {gap[mism].tolist()}"
    # VQ kernel alone at Cd = 1024: same z -> bit-exact; margin-guaranteed set; exact ties -> first index
    assert torch.equal(model.vq.encode(z_want.to(dev)).cpu(), codes_want)
    idx_m = torch.randint(0, 512, (2, 250), generator=torch.Generator().manual_seed(32))
    zm = ref_cpu.vq_embed(dic, idx_m) + 1e-3 * seeded((2, 1024, 250), 33)
    assert torch.equal(ref_cpu.vq_encode(dic, zm), idx_m) and torch.equal(model.vq.encode(zm.to(dev)).cpu(), idx_m)
    assert torch.equal(model.vq.embed(idx_m.to(dev)).cpu(), ref_cpu.vq_embed(dic, idx_m))
    with torch.no_grad():
        d2 = dic.clone()
        d2[400] = d2[23]
        model.vq.dictionary.copy_(d2)
    zt = ref_cpu.vq_embed(d2, torch.full((1, 6), 400))
    assert model.vq.encode(zt.to(dev)).cpu().tolist() == [[23] * 6]
    with torch.no_grad():
        model.vq.dictionary.copy_(dic)
    # conditional unet64 forward (cond_proj 1024 -> 64, labels) in both gate modes
    T = 16384
    x, ts = seeded((2, 1, T), 34), torch.tensor([0.7, 0.2])
    cond = ref_cpu.vq_embed(dic, codes_want[:, : T // 256])
    labels = torch.tensor([6, 0])
    want = ref_cpu.unet_predictor(sd, 64, x, ts, cond=cond, labels=labels)
    for prec in ("fp32", "fp16"):
        model.set_precision(prec)
        got = model.predictor(x.to(dev), ts.to(dev), cond=cond.to(dev), labels=labels.to(dev)).cpu()
        assert rel_rms(got, want) < GATE[prec], (prec, rel_rms(got, want))


def test_headline_workload_vs_reference_fixture(golden):
    """F6b: BENCH's exact configuration (unet64, 50 steps, t**2 sample-time schedule, constrain) on two clips, against the
    REFERENCE's own ddpm_sample output (diffusion.py:92-133 with README.md:49's schedule), in the parity mode and in the benchmarked
    fp16 mode: <= 1e-3 waveform RMS each."""
    import numpy as np

    dev = torch.device("cuda:0")
    z = golden("f6b_sampler_unet64")
    model = det_model(DiffusionModel("unet", 64))
    steps = int(z["steps"])
    x_T = seeded((2, 1, 64000), int(z["x_T_seed"]))
    gen = torch.Generator().manual_seed(int(z["noise_seed"]))
    noises = [torch.randn(x_T.shape, generator=gen) for _ in range(steps)]
    assert np.allclose([n.double().sum().item() for n in noises], z["noise_checksum"], atol=1e-6), "noise stream differs"
    want = torch.from_numpy(z["x0"])
    for prec in ("fp32", "fp16"):
        model.set_precision(prec)
        got = model.diffusion.ddpm_sample(x_T.to(dev), model.predictor, steps, constrain=True, schedule=lambda t: t ** 2,
                                          noise=[n.to(dev) for n in noises]).cpu()
        gate(f"F6b headline workload (unet64, 50 steps, t**2, constrain, 2 x 64000) {prec}", got, want, WAVE_RMS)


def test_fp16_range_guard_trips():
    """An fp16-mode model whose activations leave fp16's range must raise instead of returning garbage (the statistics pass of
    every GroupNorm flags non-finite or >= 3e4-class partial sums; vqvs_model_status); the same weights run clean in fp32."""
    from vq_voice_swap_amd import _native

    dev = torch.device("cuda:0")
    model = det_model(DiffusionModel("unet", 32))
    x_T = seeded((2, 1, 4096), 5)
    noises = [seeded((2, 1, 4096), 100 + i).to(dev) for i in range(3)]
    model.set_precision("fp16")
    model.diffusion.ddpm_sample(x_T.to(dev), model.predictor, 3, constrain=True, noise=noises)  # sane weights: no trip
    with torch.no_grad():
        model.predictor.in_conv.weight.mul_(3.0e5)  # pushes the first tensor far beyond 65504
    with pytest.raises(_native.NativeError, match="range guard"):
        model.diffusion.ddpm_sample(x_T.to(dev), model.predictor, 3, constrain=True, noise=noises)
    model.set_precision("fp32")
    out = model.diffusion.ddpm_sample(x_T.to(dev), model.predictor, 3, constrain=True, noise=noises)
    assert bool(torch.isfinite(out).all())
    # a non-finite value that never passes a GroupNorm in a 2-byte tensor -- here an inf in the LAST step's noise -- is caught on the
    # finished sample (x_t is fp32 in every mode: the library's guard does not see it)
    det_init_(model.state_dict().items())
    model.set_precision("fp16")
    bad = [n.clone() for n in noises]
    bad[1][0, 0, 7] = float("inf")
    with pytest.raises(_native.NativeError, match="non-finite"):
        model.diffusion.ddpm_sample(x_T.to(dev), model.predictor, 3, noise=bad)


def test_config4_base64_decode_vs_oracle(dev):
    """BASELINE config 4 as one composition at base 64: VQ codes -> vq.embed -> conditional unet64 diffusion decoder
    (reference vq_vae.py:92-145), 5 reverse steps at T = 16384 against the oracle: the fp32 mode is held to 1e-3; the fp16 mode's
    value is recorded only -- a 5-step schedule never averages out the first reverse step's 1 / sqrt(alpha_bar(1)) amplification
    (DESIGN.md section 4).  fp16 is gated where BASELINE quotes it: 50 steps, fixture F8c from the reference, below."""
    model = det_model(VQVAE(base_channels=64, pred_name="unet", num_labels=7))
    with torch.no_grad():
        model.vq.dictionary.copy_(seeded(model.vq.dictionary.shape, 78, 0.35))
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    T, steps = 16384, 5
    codes = torch.randint(0, 512, (2, T // 256), generator=torch.Generator().manual_seed(41))
    labels = torch.tensor([2, 5])
    x_T = seeded((2, 1, T), 42)
    gen = torch.Generator().manual_seed(43)
    noises = [torch.randn(x_T.shape, generator=gen) for _ in range(steps)]
    want = ref_cpu.vqvae_decode(sd, 64, "exp", codes, labels, steps, x_T, noises, constrain=True)
    for prec, bound in (("fp32", WAVE_RMS), ("fp16", 2.5e-3)):  # (fp16 at FIVE steps: no 1e-3 claim -- its gate is F8c, 50 steps -- but a regression bound: 6.2e-4 measured)
        model.set_precision(prec)
        got = model.decode(codes.to(dev), labels.to(dev), steps=steps, constrain=True, x_T=x_T.to(dev), noise=[n.to(dev) for n in noises]).cpu()
        gate(f"config 4 at base 64: VQVAE(64).decode {steps} steps, T = {T}, {prec}", got, want, bound)


def test_config5_unet64_with_classifier32_guidance_vs_oracle(dev):
    """BASELINE config 5 as one composition: unet64 sampled under classifier32's gradient at every step (reference
    sample_diffusion.py:34-42 + diffusion.py:80-83), 3 guided steps at T = 16384, both models in the mode under test."""
    from vq_voice_swap_amd import Classifier

    model = det_model(DiffusionModel("unet", 64))
    clf = Classifier(num_labels=7, base_channels=32)
    det_init_(("clf." + k, v) for k, v in clf.state_dict().items())
    clf.eval()
    sd_m = {k: v.detach().clone() for k, v in model.state_dict().items()}
    sd_c = {k: v.detach().clone() for k, v in clf.state_dict().items()}
    T, steps, scale = 16384, 3, 2000.0
    labels = torch.tensor([1, 6])
    x_T = seeded((2, 1, T), 51)
    gen = torch.Generator().manual_seed(52)
    noises = [torch.randn(x_T.shape, generator=gen) for _ in range(steps)]
    want = ref_cpu.ddpm_sample("exp", x_T, lambda p, q: ref_cpu.unet_predictor(sd_m, 64, p, q), steps, noises, constrain=True,
                               cond_fn=ref_cpu.classifier_cond_fn(sd_c, 32, labels, scale))
    plain = ref_cpu.ddpm_sample("exp", x_T, lambda p, q: ref_cpu.unet_predictor(sd_m, 64, p, q), steps, noises, constrain=True)
    clf.to(dev)
    errs = []
    import warnings

    # fp16 at THREE guided steps: ddpm_sample promotes the call to the fp32 mode (fewer than FEW_GUIDED_STEPS guided steps in a 2-byte
    # mode are outside the 1e-3 contract: 1.05e-3 measured, profiles/r05_parity_margins.jsonl) and says so -- the leg is gated at
    # 1e-3 again; the modules' own modes and handles are back afterwards.  The fp16 mode's gate at config 5's step count is F13.
    for prec, bound in (("fp32", WAVE_RMS), ("fp16", WAVE_RMS)):
        model.set_precision(prec)
        clf.set_precision(prec)
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            got = model.diffusion.ddpm_sample(x_T.to(dev), model.predictor, steps, constrain=True, cond_fn=clf.guidance_fn(labels.to(dev), scale),
                                              noise=[n.to(dev) for n in noises]).cpu()
        promoted = [w for w in rec if "run in the fp32" in str(w.message)]
        assert bool(promoted) == (prec == "fp16"), [str(w.message) for w in rec]
        assert model.predictor.precision == prec and clf.precision == prec
        errs.append(gate(f"config 5: unet64 + classifier32 guidance, {steps} steps, T = {T}, {prec}"
                         + (" (promoted to fp32 by ddpm_sample)" if prec == "fp16" else ""), got, want, bound))
    assert rms(want - plain) > 10 * errs[0], ("guidance term too small for the comparison to mean anything", errs, rms(want - plain))
    model.predictor.invalidate()


def test_config5_at_100_steps_vs_reference_fixture(golden, dev):
    """F13: BASELINE config 5 at ITS step count -- unet64 sampled under classifier32's gradient at each of 100 steps (reference
    sample_diffusion.py:34-42 + diffusion/diffusion.py:80-83, 92-133), 2 clips x T = 16384, constrain -- against the REFERENCE's
    own output, in the parity mode and in the quoted fp16 mode: <= 1e-3 waveform RMS each."""
    import numpy as np

    from vq_voice_swap_amd import Classifier

    z = golden("f13_guided_unet64_100")
    model = det_model(DiffusionModel("unet", 64))
    clf = Classifier(num_labels=7, base_channels=32)
    det_init_(("clf." + k, v) for k, v in clf.state_dict().items())
    clf.eval().to(dev)
    steps, scale = int(z["steps"]), float(z["scale"])
    labels = torch.from_numpy(z["labels"])
    x_T = seeded((2, 1, 16384), int(z["x_T_seed"]))
    gen = torch.Generator().manual_seed(int(z["noise_seed"]))
    noises = [torch.randn(x_T.shape, generator=gen) for _ in range(steps)]
    assert np.allclose([n.double().sum().item() for n in noises], z["noise_checksum"], atol=1e-6), "noise stream differs"
    want = torch.from_numpy(z["x0"])
    assert float(z["guidance_effect_rms"]) > 10 * WAVE_RMS  # (the guided sample is far from the unguided one: the comparison means something)
    for prec in ("fp32", "fp16"):
        model.set_precision(prec)
        clf.set_precision(prec)
        got = model.diffusion.ddpm_sample(x_T.to(dev), model.predictor, steps, constrain=True, cond_fn=clf.guidance_fn(labels.to(dev), scale),
                                          noise=[n.to(dev) for n in noises]).cpu()
        gate(f"F13 config 5 (unet64 + classifier32 guidance, 100 steps, constrain, 2 x 16384) {prec}", got, want, WAVE_RMS)
    model.predictor.invalidate()


def test_config4_base64_at_50_steps_vs_reference_fixture(golden, dev):
    """F8c: BASELINE config 4 at base 64 and ITS step count -- VQVAE(64).decode (reference vq_vae.py:92-145), 50 steps, T = 16384,
    constrain -- against the REFERENCE's own output, fp32 and fp16 decoder: <= 1e-3 waveform RMS each."""
    import numpy as np

    z = golden("f8c_vqvae64_decode50")
    model = det_model(VQVAE(base_channels=64, pred_name="unet", num_labels=7))
    with torch.no_grad():
        model.vq.dictionary.copy_(seeded(model.vq.dictionary.shape, 78, 0.35))
    steps = int(z["steps"])
    codes, labels = torch.from_numpy(z["codes"]), torch.from_numpy(z["labels"])
    x_T = seeded((2, 1, 16384), int(z["x_T_seed"]))
    gen = torch.Generator().manual_seed(int(z["noise_seed"]))
    noises = [torch.randn(x_T.shape, generator=gen) for _ in range(steps)]
    assert np.allclose([n.double().sum().item() for n in noises], z["noise_checksum"], atol=1e-6), "noise stream differs"
    want = torch.from_numpy(z["x0"])
    for prec in ("fp32", "fp16"):
        model.set_precision(prec)
        got = model.decode(codes.to(dev), labels.to(dev), steps=steps, constrain=True, x_T=x_T.to(dev), noise=[n.to(dev) for n in noises]).cpu()
        gate(f"F8c config 4 at base 64 (VQVAE(64).decode, 50 steps, constrain, 2 x 16384) {prec}", got, want, WAVE_RMS)


def test_base128_forward_vs_oracle(dev):
    """The reference accepts any width (models/unet.py:17-30); beyond its two published ones this library builds 128."""
    model = det_model(DiffusionModel("unet", 128))
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    x, ts = seeded((2, 1, 4096), 61), torch.tensor([0.3, 0.8])
    want = ref_cpu.unet_predictor(sd, 128, x, ts)
    for prec in ("fp32", "fp16"):
        model.set_precision(prec)
        got = model.predictor(x.to(dev), ts.to(dev)).cpu()
        assert rel_rms(got, want) < GATE[prec], (prec, rel_rms(got, want))
    model.predictor.invalidate()


def test_resblock_many_clips_long_launch_seeded_sweep(dev):
    """tools/fuzz_resblock.py's "big" mode as a seeded case list: many tiles per workgroup and workgroups that cross clip
    boundaries (the (scale, shift) ring, the fused GroupNorm tables, reversed tile order), in both gate modes."""
    import random

    from vq_voice_swap_amd.unet import ResBlockModule

    rng = random.Random(1234)
    for i in range(6):
        scale = rng.choice([1.0, 1.0, 0.5, 2.0])
        cin = rng.choice([64, 128])
        cout = cin if scale != 1.0 else rng.choice([cin, 64, 128])
        dil = 2 if scale == 2.0 else rng.choice([1, 2, 4])
        emb = rng.choice([None, 256])
        L = rng.choice([4000, 6002, 8190, 12000])
        B = rng.choice([24, 37, 48])
        m = ResBlockModule(cin, emb, cout if cout != cin else None, scale, dil)
        det_init_((f"big{i}." + k, v) for k, v in m.block.state_dict().items())
        x = seeded((B, cin, L), 7000 + i)
        e = seeded((B, emb), 7100 + i) if emb else None
        sd = {"b." + k: v.detach() for k, v in m.block.state_dict().items()}
        want = ref_cpu.res_block(x, sd, "b", dict(cin=cin, cout=cout, scale=scale, dil=dil), e)
        for prec, tol in (("fp32", 2e-4), ("fp16", 4e-3)):
            m.set_precision(prec)
            got = m(x.to(dev), None if e is None else e.to(dev)).cpu()
            err = rel_rms(got, want) if got.shape == want.shape else float("inf")
            assert err < tol, (prec, dict(cin=cin, cout=cout, scale=scale, dil=dil, emb=emb, L=L, B=B), err)


def test_one_step_clips_many_per_workgroup(dev):
    """32 x 3 -> 32 at L <= 254 is ONE K chunk per clip: a persistent workgroup then walks one clip per step, the producers' load
    cursor runs four clips ahead of the clip being staged, and the (scale, shift) ring must hold all of them (a ring
    of 4 was overwritten from ~1024 clips on).  1500 and 2600 clips (6 and 10 per workgroup), with and without FiLM."""
    from vq_voice_swap_amd.unet import ResBlockModule

    for i, (B, L, emb, dil) in enumerate([(1500, 200, None, 1), (2600, 96, 128, 2)]):
        m = ResBlockModule(32, emb, None, 1.0, dil)
        det_init_((f"ring{i}." + k, v) for k, v in m.block.state_dict().items())
        x = seeded((B, 32, L), 8000 + i)
        e = seeded((B, emb), 8100 + i) if emb else None
        sd = {"b." + k: v.detach() for k, v in m.block.state_dict().items()}
        want = ref_cpu.res_block(x, sd, "b", dict(cin=32, cout=32, scale=1.0, dil=dil), e)
        for prec, tol in (("fp32", 2e-4), ("fp16", 4e-3)):
            m.set_precision(prec)
            got = m(x.to(dev), None if e is None else e.to(dev)).cpu()
            err = rel_rms(got, want)
            # per clip too: a wrong table for ONE clip must not hide in the batch's RMS
            per_clip = ((got - want).pow(2).mean(dim=(1, 2)).sqrt() / want.pow(2).mean(dim=(1, 2)).sqrt()).max().item()
            assert err < tol and per_clip < 5 * tol, (prec, B, L, err, per_clip)
