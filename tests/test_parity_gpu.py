"""Parity of the HIP path (through the C ABI) with the CPU oracle and the golden vectors, on a real MI355X.

Tolerances (north_star): waveforms within 1e-3 RMS of the CPU reference; VQ code indices bit-exact.  Two modes meet
the waveform gate and are held to it here: "fp32" (fp32 activations, 3-term bf16-split MFMA: ~5e-6) and "fp16" (fp16
activations and weights, f16 MFMA, fp32 statistics / accumulation: 3e-4 ... 8e-4, the benchmarked mode).  The bf16 mode
(8 significant bits: ~1e-2 on eps, 2.6e-3 on waveforms) is outside the gate and only checked against a loose relative bound."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu
from vq_voice_swap_amd import Diffusion, DiffusionModel, ResBlockModule, VQVAE, make_schedule, randn_clips
from vq_voice_swap_amd.det_init import det_init_

from util import gate, rel_rms, rms, seeded

pytestmark = pytest.mark.gpu
torch.set_num_threads(8)

FP32_REL = 1e-4   # per-forward relative RMS bound in fp32 mode (measured ~2e-5)
FP16_REL = 4e-3   # per-forward relative RMS bound on eps in fp16 mode (measured 1.3e-3 ... 1.6e-3; per ResBlock 4.5e-4)
BF16_REL = 3e-2   # per-forward relative RMS bound in bf16 mode (measured ~1.3e-2)
WAVE_RMS = 1e-3   # north_star gate on sampled waveforms (fp32 AND fp16 modes)
MODES = [("fp32", FP32_REL), ("fp16", FP16_REL), ("bf16", BF16_REL)]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def det_model(m):
    det_init_(m.state_dict().items())
    m.eval()
    return m


@pytest.mark.parametrize("prec,tol", MODES)
def test_resblocks_vs_golden(golden, dev, prec, tol):
    z = golden("f1_resblocks")
    for name in sorted({k.split(".")[0] for k in z.files}):
        cin, cout, scale, dil, emb, L = z[name + ".spec"]
        m = ResBlockModule(int(cin), int(emb) or None, int(cout) if cout != cin else None, float(scale), int(dil))
        det_init_(("blk." + name + "." + k, v) for k, v in m.block.state_dict().items())
        m.set_precision(prec)
        e = torch.from_numpy(z[name + ".emb"]).to(dev) if emb else None
        y = m(torch.from_numpy(z[name + ".x"]).to(dev), e).cpu()
        assert rel_rms(y, torch.from_numpy(z[name + ".y"])) < tol, name


def test_resblock_ragged_lengths_vs_oracle(dev):
    """Tile-edge cases: lengths that are not multiples of the 256-row tile, single rows, odd batch."""
    for L, B in ((1, 2), (2, 3), (255, 2), (257, 1), (600, 2)):
        m = ResBlockModule(64, 128, 32, 1.0, 2)
        det_init_(("rag." + k, v) for k, v in m.block.state_dict().items())
        x, emb = seeded((B, 64, L), 5 + L), seeded((B, 128), 6 + L)
        sd = {"b." + k: v.detach() for k, v in m.block.state_dict().items()}
        want = ref_cpu.res_block(x, sd, "b", dict(cin=64, cout=32, scale=1.0, dil=2), emb)
        got = m(x.to(dev), emb.to(dev)).cpu()
        assert rel_rms(got, want) < FP32_REL, (L, B)


def test_unet32_forward_and_every_block_vs_oracle(golden, dev):
    z = golden("f3_unet32_forward")
    x = seeded((2, 1, 64000), int(z["x_seed"]))
    ts = torch.from_numpy(z["ts"])
    want = torch.from_numpy(z["eps"])
    model = det_model(DiffusionModel("unet", 32))
    model.predictor.debug_taps = True
    eps = model.predictor(x.to(dev), ts.to(dev)).cpu()
    assert rel_rms(eps, want) < FP32_REL
    sd = {"predictor." + k: v for k, v in model.predictor.state_dict().items()}
    taps = {}
    ref_cpu.unet_predictor(sd, 32, x, ts, probe=lambda n, t: taps.__setitem__(n, t))
    h = model.predictor._handle
    assert len(h.taps()) == 66
    for i, (name, ch, ls) in enumerate(h.taps()):
        assert rel_rms(h.read_tap(i, 2, 64000), taps[name]) < FP32_REL, name
    model.predictor.debug_taps = False
    for prec, tol in MODES[1:]:
        model.set_precision(prec)
        assert rel_rms(model.predictor(x.to(dev), ts.to(dev)).cpu(), want) < tol, prec


def test_forward_is_deterministic_and_batch_independent(dev):
    model = det_model(DiffusionModel("unet", 32))
    x = seeded((3, 1, 2048), 3).to(dev)
    ts = torch.tensor([0.2, 0.5, 0.9], device=dev)
    a = model.predictor(x, ts)
    b = model.predictor(x, ts)
    assert torch.equal(a, b)
    # a clip's result does not depend on its neighbours in the batch (no cross-clip op anywhere)
    c = model.predictor(x[1:2].contiguous(), ts[1:2].contiguous())
    assert torch.equal(a[1:2], c)


def test_conditioning_of_any_length_vs_oracle(dev):
    """The reference up-samples a conditioning sequence of ANY length to T (F.interpolate(cond, T), nearest; unet.py:138-139): lengths
    that are neither T / 256 nor T / 320, shorter and longer than both, odd, one row, and longer than T / 2."""
    model = det_model(VQVAE(base_channels=32, pred_name="unet", num_labels=4))
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    labels = torch.tensor([3, 1])
    for T, Lc, seed in ((4096, 37, 1), (4096, 1, 2), (8192, 100, 3), (4096, 3000, 4), (4096, 16, 5), (8192, 513, 6)):
        x, ts = seeded((2, 1, T), 70 + seed), torch.tensor([0.35, 0.8])
        cond = seeded((2, 512, Lc), 80 + seed, 0.5)
        want = ref_cpu.unet_predictor(sd, 32, x, ts, cond=cond, labels=labels)
        for prec, tol in (("fp32", FP32_REL), ("fp16", FP16_REL)):
            model.set_precision(prec)
            got = model.predictor(x.to(dev), ts.to(dev), cond=cond.to(dev), labels=labels.to(dev)).cpu()
            assert rel_rms(got, want) < tol, (prec, T, Lc, rel_rms(got, want))
    model.set_precision("fp32")
    # a caller alternating between conditioning lengths gets the handle of a length back instead of a rebuild (a handle is built for
    # one length code), with the same bits
    p = model.predictor
    x, ts = seeded((2, 1, 4096), 91).to(dev), torch.tensor([0.35, 0.8]).to(dev)
    ca, cb = seeded((2, 512, 37), 92, 0.5).to(dev), seeded((2, 512, 5), 93, 0.5).to(dev)
    ya = p(x, ts, cond=ca, labels=labels.to(dev))
    ha = p._handle
    yb = p(x, ts, cond=cb, labels=labels.to(dev))
    assert p._handle is not ha and len(p.__dict__.get("_cond_handles", {})) == 1
    assert torch.equal(p(x, ts, cond=ca, labels=labels.to(dev)), ya) and p._handle is ha
    assert torch.equal(p(x, ts, cond=cb, labels=labels.to(dev)), yb)
    p.invalidate()
    assert not p.__dict__.get("_cond_handles")
    with pytest.raises(ValueError, match="expected cond of shape"):
        model.predictor(torch.zeros(2, 1, 4096, device=dev), torch.zeros(2, device=dev), cond=torch.zeros(2, 256, 16, device=dev), labels=labels.to(dev))


def test_open_topology_vs_reference_fixture(golden, dev):
    """F14: UNetPredictor / UNetEncoder with channel_mult / depth_mult / middle_dilations / out_dilations other than the defaults
    (reference unet.py:17-30, 188-196), built by the reference's own constructors: forward outputs in every mode."""
    from vq_voice_swap_amd import UNetEncoder, UNetPredictor

    z = golden("f14_custom_topologies")
    preds = [("p_a", dict(channel_mult=(1, 2, 2, 4), middle_dilations=(1, 6), depth_mult=1), dict(num_labels=3)),
             ("p_b", dict(channel_mult=(1, 1, 2), middle_dilations=(), depth_mult=3), {}),
             ("p_c", dict(channel_mult=(1, 4, 8, 8, 16), middle_dilations=(2, 32, 5), depth_mult=2), dict(cond_channels=64))]
    for tag, topo, extra in preds:
        m = UNetPredictor(32, **topo, **extra)
        det_init_(("predictor." + tag + "." + k, v) for k, v in m.state_dict().items())
        m.eval()
        kw = {k: torch.from_numpy(z[f"{tag}.{k}"]).to(dev) for k in ("labels", "cond") if f"{tag}.{k}" in z.files}
        want = torch.from_numpy(z[tag + ".eps"])
        for prec, tol in MODES:
            m.set_precision(prec)
            got = m(torch.from_numpy(z[tag + ".x"]).to(dev), torch.from_numpy(z[tag + ".ts"]).to(dev), **kw).cpu()
            assert rel_rms(got, want) < tol, (tag, prec, rel_rms(got, want))
        with pytest.raises(ValueError, match="downsample rate"):
            m(torch.zeros(1, 1, m.downsample_rate * 3 + 1, device=dev), torch.zeros(1, device=dev), **{k: v[:1] for k, v in kw.items()})
    for tag, topo, oc in (("e_a", dict(channel_mult=(1, 2, 4), out_dilations=(2, 8), depth_mult=1), 64),
                          ("e_b", dict(channel_mult=(1, 1, 2, 2, 4, 4), out_dilations=(), depth_mult=3), 96)):
        m = UNetEncoder(32, out_channels=oc, **topo)
        det_init_(("encoder." + tag + "." + k, v) for k, v in m.state_dict().items())
        m.eval()
        want = torch.from_numpy(z[tag + ".z"])
        got = m(torch.from_numpy(z[tag + ".x"]).to(dev)).cpu()
        assert got.shape == want.shape and rel_rms(got, want) < FP32_REL, (tag, rel_rms(got, want))


def test_unusual_widths_vs_oracle(dev):
    """The reference takes any base_channels (unet.py:17-30): widths other than the tuned 32 / 64 / 128 -- 96 (GroupNorm groups of 3
    channels, 32-channel tiles, a 12-octet row in in_conv / out_conv) and 160 -- run generic forms of a few kernels: predictor (with
    labels) and encoder against the oracle."""
    from vq_voice_swap_amd import UNetEncoder, UNetPredictor

    topo = dict(channel_mult=(1, 2, 4), middle_dilations=(3,), depth_mult=1)
    for base, T in ((96, 2048), (160, 1024)):
        m = UNetPredictor(base, num_labels=3, **topo)
        det_init_((f"predictor.w{base}." + k, v) for k, v in m.state_dict().items())
        m.eval()
        sd = {"predictor." + k: v.detach().clone() for k, v in m.state_dict().items()}
        x, ts, labels = seeded((2, 1, T), 95 + base), torch.tensor([0.2, 0.7]), torch.tensor([2, 0])
        want = ref_cpu.unet_predictor(sd, base, x, ts, labels=labels, topology=topo)
        for prec, tol in (("fp32", FP32_REL), ("fp16", FP16_REL)):
            m.set_precision(prec)
            got = m(x.to(dev), ts.to(dev), labels=labels.to(dev)).cpu()
            assert rel_rms(got, want) < tol, (base, prec, rel_rms(got, want))
    e = UNetEncoder(96, channel_mult=(1, 2), depth_mult=2, out_channels=96)
    det_init_(("encoder.w96." + k, v) for k, v in e.state_dict().items())
    e.eval()
    sde = {"encoder." + k: v.detach().clone() for k, v in e.state_dict().items()}
    xe = seeded((2, 1, 1024), 97, 0.3)
    wante = ref_cpu.unet_encoder(sde, 96, xe, topology=dict(channel_mult=(1, 2), out_dilations=(), depth_mult=2))
    assert rel_rms(e(xe.to(dev)).cpu(), wante) < FP32_REL
    with pytest.raises(ValueError, match="builds up to 256"):
        UNetPredictor(300)  # (48, 40, ... are built at a padded width since round 6: test_widths_that_are_not_multiples_of_32_vs_oracle)


def test_multi_channel_input_vs_oracle(dev):
    """in_channels > 1 (unet.py:25, 193: the reference's constructors take it; every caller uses 1): predictor with 3 input channels
    and 2 output channels... the output head keeps its own width; encoder with 2 input channels -- against the oracle."""
    from vq_voice_swap_amd import UNetEncoder, UNetPredictor

    m = UNetPredictor(32, channel_mult=(1, 2, 2), middle_dilations=(2,), depth_mult=1, in_channels=3)
    det_init_(("predictor.mc." + k, v) for k, v in m.state_dict().items())
    m.eval()
    sd = {"predictor." + k: v.detach().clone() for k, v in m.state_dict().items()}
    x, ts = seeded((2, 3, 1024), 91), torch.tensor([0.25, 0.6])
    want = ref_cpu.unet_predictor(sd, 32, x, ts, topology=dict(channel_mult=(1, 2, 2), middle_dilations=(2,), depth_mult=1))
    for prec, tol in (("fp32", FP32_REL), ("fp16", FP16_REL)):
        m.set_precision(prec)
        got = m(x.to(dev), ts.to(dev)).cpu()
        assert got.shape == want.shape == (2, 1, 1024) and rel_rms(got, want) < tol, (prec, rel_rms(got, want))
    with pytest.raises(ValueError, match="expected x of shape"):
        m(torch.zeros(2, 1, 1024, device=dev), ts.to(dev))
    e = UNetEncoder(32, channel_mult=(1, 2), depth_mult=1, in_channels=2, out_channels=64)
    det_init_(("encoder.mc." + k, v) for k, v in e.state_dict().items())
    e.eval()
    sde = {"encoder." + k: v.detach().clone() for k, v in e.state_dict().items()}
    xe = seeded((3, 2, 512), 92, 0.3)
    wante = ref_cpu.unet_encoder(sde, 32, xe, topology=dict(channel_mult=(1, 2), out_dilations=(), depth_mult=1))
    gote = e(xe.to(dev)).cpu()
    assert gote.shape == wante.shape and rel_rms(gote, wante) < FP32_REL


def test_whole_clip_tiles_for_wide_dilations_vs_oracle(dev):
    """The middle blocks' shapes (unet.py:21, 78-88: dilation 4 .. 32 at 250 rows, 256 / 512 channels): a clip of at most 255 rows is
    one zero-padded tile of conv_ws_kernel whatever the dilation (template flag ZP) -- every dilation, clip lengths around the limits
    (255 = last length covered, 256 = back on two tiles, shorter than the dilation's reach, odd), against the oracle."""
    cases = [(256, 4, 250, 3), (256, 8, 250, 2), (256, 16, 255, 2), (256, 32, 250, 3), (512, 32, 250, 2), (512, 4, 131, 2),
             (256, 32, 37, 2), (256, 16, 256, 2), (256, 3, 253, 2)]
    for i, (C, dil, L, B) in enumerate(cases):
        m = ResBlockModule(C, 128, None, 1.0, dil)
        det_init_((f"zp{i}." + k, v) for k, v in m.block.state_dict().items())
        x, e = seeded((B, C, L), 600 + i), seeded((B, 128), 650 + i)
        sd = {"b." + k: v.detach() for k, v in m.block.state_dict().items()}
        want = ref_cpu.res_block(x, sd, "b", dict(cin=C, cout=C, scale=1.0, dil=dil), e)
        for prec, tol in (("fp32", 2e-4), ("fp16", 4e-3), ("bf16", BF16_REL)):
            m.set_precision(prec)
            got = m(x.to(dev), e.to(dev)).cpu()
            assert rel_rms(got, want) < tol, (prec, C, dil, L, rel_rms(got, want))


def test_whole_clip_tiles_wide_instantiation_and_batch_invariance(dev):
    """The whole-clip (ZP) kernel has two instantiations: 64-channel tiles where 128-channel ones would leave half of the chip idle
    (every case above: B <= 3) and 128-channel tiles at production batch.  70 clips of 512 x 250 take the 128-channel form
    (1 x 4 x 70 x 2 > 256 CUs): against the oracle, and BITWISE against the same clips run two at a time on the 64-channel form -- a
    clip's bits must not depend on the batch it travels in."""
    for i, (C, dil) in enumerate([(512, 8), (256, 32)]):
        B, L = (70, 250) if C == 512 else (140, 250)
        m = ResBlockModule(C, 128, None, 1.0, dil)
        det_init_((f"zpw{i}." + k, v) for k, v in m.block.state_dict().items())
        x, e = seeded((B, C, L), 680 + i), seeded((B, 128), 690 + i)
        sd = {"b." + k: v.detach() for k, v in m.block.state_dict().items()}
        pick = [0, B // 2 - 2, B - 1]
        want = ref_cpu.res_block(x[pick], sd, "b", dict(cin=C, cout=C, scale=1.0, dil=dil), e[pick])
        for prec, tol in (("fp16", 4e-3), ("bf16", BF16_REL)):
            m.set_precision(prec)
            full = m(x.to(dev), e.to(dev))
            assert rel_rms(full[pick].cpu(), want) < tol, (prec, C, dil, rel_rms(full[pick].cpu(), want))
            for j in pick:
                j0 = min(j, B - 2)
                two = m(x[j0:j0 + 2].to(dev), e[j0:j0 + 2].to(dev))
                assert torch.equal(two[j - j0], full[j]), (prec, C, dil, j)


def test_handle_less_entry_points_on_two_streams(dev):
    """vqvs_ddpm_step(CONSTRAIN) and vqvs_vq_argmin keep their scratch per (device, stream): the same calls interleaved on two
    streams give the results of running them one after the other."""
    diff = Diffusion(make_schedule("exp"))
    B, T = 6, 64000
    xs = [seeded((B, 1, T), 500 + i).to(dev) for i in range(2)]
    es = [seeded((B, 1, T), 510 + i).to(dev) for i in range(2)]
    ns = [seeded((B, 1, T), 520 + i).to(dev) for i in range(2)]
    ts = torch.full((B,), 0.6, device=dev)
    model = det_model(VQVAE(base_channels=32, pred_name="unet", num_labels=2))
    zs = [seeded((B, 512, 250), 530 + i).to(dev) for i in range(2)]
    want = [(diff.ddpm_previous(xs[i], ts, 0.02, es[i], noise=ns[i], constrain=True), model.vq.encode(zs[i])) for i in range(2)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    got = [None, None]
    for rep in range(8):
        for i in range(2):
            with torch.cuda.stream(streams[i]):
                got[i] = (diff.ddpm_previous(xs[i], ts, 0.02, es[i], noise=ns[i], constrain=True), model.vq.encode(zs[i]))
    torch.cuda.synchronize()
    for i in range(2):
        assert torch.equal(got[i][0], want[i][0]) and torch.equal(got[i][1], want[i][1]), i


def test_bad_lengths_raise(dev):
    model = det_model(DiffusionModel("unet", 32))
    with pytest.raises(ValueError, match="multiple of the UNet downsample rate"):
        model.predictor(torch.zeros(1, 1, 1000, device=dev), torch.zeros(1, device=dev))


def test_ddpm_previous_vs_golden(golden, dev):
    z = golden("f5_ddpm_previous")
    d = Diffusion(make_schedule("exp"))
    for i in range(5):
        t, step = z[f"c{i}.t_step"]
        x, eps, noise = (torch.from_numpy(z[f"c{i}.{k}"]).to(dev) for k in ("x", "eps", "noise"))
        ts = torch.tensor([t, t], dtype=torch.float32, device=dev)
        for mode, kw in (("plain", {}), ("sigma_large", dict(sigma_large=True)), ("constrain", dict(constrain=True))):
            want = torch.from_numpy(z[f"c{i}.{mode}"])
            got = d.ddpm_previous(x, ts, float(step), eps, noise=noise, **kw).cpu()
            assert (got - want).abs().max().item() <= 2e-6 * max(1.0, want.abs().max().item()), (i, mode)
    got = d.ddpm_previous(torch.from_numpy(z["row.x"]).to(dev), torch.from_numpy(z["row.ts"]).to(dev), torch.from_numpy(z["row.step"]).to(dev),
                          torch.from_numpy(z["row.eps"]).to(dev), noise=torch.from_numpy(z["row.noise"]).to(dev), constrain=True).cpu()
    assert (got - torch.from_numpy(z["row.constrain"])).abs().max().item() <= 2e-6


def test_guided_step_matches_oracle(dev):
    """cond_fn path (diffusion.py:80-83) with an analytic cond_fn."""
    d = Diffusion(make_schedule("exp"))
    x, eps, noise = seeded((2, 1, 4096), 1), seeded((2, 1, 4096), 2), seeded((2, 1, 4096), 3)
    ts = torch.tensor([0.6, 0.3])
    fn = lambda m, t: torch.tanh(m) * t.view(-1, 1, 1)  # noqa: E731
    want = ref_cpu.ddpm_previous("exp", x, ts, 0.05, eps, noise, constrain=True, cond_fn=fn)
    got = d.ddpm_previous(x.to(dev), ts.to(dev), 0.05, eps.to(dev), noise=noise.to(dev), constrain=True, cond_fn=fn).cpu()
    assert (got - want).abs().max().item() <= 1e-5


@pytest.mark.parametrize("tag,steps,constrain,sq", [("s10_plain", 10, False, False), ("s10_constrain", 10, True, False),
                                                    ("s50_sq_constrain", 50, True, True)])
def test_sampler_end_to_end_vs_golden(golden, dev, tag, steps, constrain, sq):
    z = golden("f6_sampler_unet32")
    model = det_model(DiffusionModel("unet", 32))
    x_T = seeded((2, 1, 64000), int(z["x_T_seed"]))
    gen = torch.Generator().manual_seed(int(z["noise_seed"]))
    noises = [torch.randn(x_T.shape, generator=gen) for _ in range(steps)]
    assert np.allclose([n.double().sum().item() for n in noises], z[tag + ".noise_checksum"], atol=1e-6), "noise stream differs"
    tmap = (lambda t: t ** 2) if sq else None
    want = torch.from_numpy(z[tag + ".x0"])
    got = model.diffusion.ddpm_sample(x_T.to(dev), model.predictor, steps, constrain=constrain, schedule=tmap,
                                      noise=[n.to(dev) for n in noises]).cpu()
    # bounded waveform: absolute gate; unconstrained: untrained weights blow x up to RMS ~455, relative gate
    gate(f"F6 unet32 {tag} fp32", got, want, WAVE_RMS, relative=not constrain)
    # the benchmarked mode is held to the same gate
    model.set_precision("fp16")
    got = model.diffusion.ddpm_sample(x_T.to(dev), model.predictor, steps, constrain=constrain, schedule=tmap,
                                      noise=[n.to(dev) for n in noises]).cpu()
    gate(f"F6 unet32 {tag} fp16", got, want, WAVE_RMS, relative=not constrain)
    model.set_precision("bf16")
    got = model.diffusion.ddpm_sample(x_T.to(dev), model.predictor, steps, constrain=constrain, schedule=tmap,
                                      noise=[n.to(dev) for n in noises]).cpu()
    assert rel_rms(got, want) < BF16_REL


def test_vq_and_vqvae_vs_golden(golden, dev):
    z7, z4, z8 = golden("f7_encoder_vq32"), golden("f4_cond_forward"), golden("f8_vqvae_decode")
    model = det_model(VQVAE(base_channels=32, pred_name="unet", num_labels=5))
    with torch.no_grad():
        model.vq.dictionary.copy_(seeded(model.vq.dictionary.shape, 77, 0.35))
    dic = model.vq.dictionary.detach()
    # VQ kernel alone: bit-exact indices
    zin = torch.from_numpy(z7["z"]).float()
    assert torch.equal(model.vq.encode(zin.to(dev)).cpu(), ref_cpu.vq_encode(dic, zin))
    idx_m = torch.from_numpy(z7["margin_idx"])
    zm = ref_cpu.vq_embed(dic, idx_m) + 1e-3 * seeded((2, 512, 250), int(z7["margin_noise_seed"]))
    assert torch.equal(model.vq.encode(zm.to(dev)).cpu(), idx_m)
    assert torch.equal(model.vq.embed(idx_m.to(dev)).cpu(), ref_cpu.vq_embed(dic, idx_m))
    # exact ties: duplicate codewords -> the first index must win (torch.argmin semantics)
    with torch.no_grad():
        d2 = dic.clone()
        d2[300] = d2[17]
        model.vq.dictionary.copy_(d2)
    zt = ref_cpu.vq_embed(d2, torch.full((1, 8), 300))
    assert model.vq.encode(zt.to(dev)).cpu().tolist() == [[17] * 8]
    with torch.no_grad():
        model.vq.dictionary.copy_(dic)
    # encoder + VQ end to end (fp32 mode): codes equal the reference's; any mismatch must be a provable near-tie
    wav = seeded((2, 1, 64000), int(z7["wav_seed"]), 0.1).clamp(-1, 1)
    codes = model.encode(wav.to(dev)).cpu()
    want = torch.from_numpy(z7["codes"])
    mism = codes != want
    assert mism.sum().item() == 0, f"{int(mism.sum())} codes differ; top-2 distance gaps there: {torch.from_numpy(z7['gap'])[mism].tolist()}"
    assert codes.dtype == torch.int64 and codes.shape == (2, 250)
    # a 2-byte predictor precision must not touch the encoder: codes stay bit-exact
    model.set_precision("fp16")
    assert model.encoder.precision == "fp32" and torch.equal(model.encode(wav.to(dev)).cpu(), want)
    model.set_precision("fp32")
    # conditional forward + decode
    cond = model.vq.embed(torch.from_numpy(z4["codes16"]).to(dev))
    eps = model.predictor(torch.from_numpy(z4["x"]).to(dev), torch.from_numpy(z4["ts"]).to(dev), cond=cond,
                          labels=torch.from_numpy(z4["labels"]).to(dev)).cpu()
    assert rel_rms(eps, torch.from_numpy(z4["eps"])) < FP32_REL
    x_T = seeded((2, 1, 4096), int(z8["x_T_seed"]))
    gen = torch.Generator().manual_seed(int(z8["noise_seed"]))
    noises = [torch.randn(x_T.shape, generator=gen).to(dev) for _ in range(5)]
    dec = model.decode(torch.from_numpy(z8["codes16"]).to(dev), torch.from_numpy(z8["labels"]).to(dev), steps=5, constrain=True,
                       x_T=x_T.to(dev), noise=noises).cpu()
    gate("F8 vqvae32 decode 5 steps fp32", dec, torch.from_numpy(z8["x0"]), WAVE_RMS)
    # fp16 at FIVE steps is not a gate claim: the first reverse step divides the predictor's rounding error by sqrt(alpha_bar(1)) =
    # 3e-3 and five steps do not average it out; 58 % of this fixture's samples sit on the clamp, and which ones flip depends on
    # the summation order -- equally valid fp16 schedules of this library measure 0.8e-3 ... 1.1e-3 here (tools/f8_rms.py).  The
    # value is RECORDED, not gated (bound None); the 1e-3 claim is made at BASELINE config 4's step count, in the 50-step leg below.
    model.set_precision("fp16")
    dec = model.decode(torch.from_numpy(z8["codes16"]).to(dev), torch.from_numpy(z8["labels"]).to(dev), steps=5, constrain=True,
                       x_T=x_T.to(dev), noise=noises).cpu()
    # (no 1e-3 claim at five un-guided steps in fp16 -- DESIGN.md section 4 -- but a numeric regression bound from the recorded margins:
    #  0.8e-3 ... 1.1e-3 over rounds 3-6 by summation order)
    gate("F8 vqvae32 decode 5 steps fp16 (not a gate claim: regression bound)", dec, torch.from_numpy(z8["x0"]), 2.5e-3)
    # F8b: 50 steps (BASELINE config 4), the reference's own output: fp32 AND fp16 inside 1e-3
    z8b = golden("f8b_vqvae_decode50")
    x_T = seeded((2, 1, 4096), int(z8b["x_T_seed"]))
    gen = torch.Generator().manual_seed(int(z8b["noise_seed"]))
    noises = [torch.randn(x_T.shape, generator=gen).to(dev) for _ in range(50)]
    for prec in ("fp32", "fp16"):
        model.set_precision(prec)
        dec = model.decode(torch.from_numpy(z8b["codes16"]).to(dev), torch.from_numpy(z8b["labels"]).to(dev), steps=50, constrain=True,
                           x_T=x_T.to(dev), noise=noises).cpu()
        gate(f"F8b vqvae32 decode 50 steps {prec}", dec, torch.from_numpy(z8b["x0"]), WAVE_RMS)


def test_time_embedding_vs_golden(golden, dev):
    """F2: the conditioning vector (sinusoid with arguments up to 100 rad -> Linear -> GELU -> Linear + class embedding,
    wavegrad.py:359-373, unet.py:133-135) read back from the handle after a forward."""
    z = golden("f2_time_embed")
    ts = torch.from_numpy(z["ts"])
    for base in (32, 64):
        model = det_model(DiffusionModel("unet", base, num_labels=6))
        labels = torch.from_numpy(z[f"c{base}.labels"])
        model.predictor(seeded((3, 1, 256), 1).to(dev), ts.to(dev), labels=labels.to(dev))
        emb = model.predictor._handle.read_embedding(3)
        want = torch.from_numpy(z[f"c{base}.emb"])
        assert emb.shape == want.shape and (emb - want).abs().max().item() <= 2e-5, (base, (emb - want).abs().max().item())


def test_ddpm_previous_cos_schedule_vs_golden(golden, dev):
    """F5b: CosSchedule (schedule.py:34-41) through the same step kernels, away from the ill-conditioned t = 1."""
    z = golden("f5b_ddpm_previous_cos")
    d = Diffusion(make_schedule("cos"))
    for i in range(5):
        t, step = z[f"c{i}.t_step"]
        x, eps, noise = (torch.from_numpy(z[f"c{i}.{k}"]).to(dev) for k in ("x", "eps", "noise"))
        ts = torch.tensor([t, t], dtype=torch.float32, device=dev)
        for mode, kw in (("plain", {}), ("sigma_large", dict(sigma_large=True)), ("constrain", dict(constrain=True))):
            want = torch.from_numpy(z[f"c{i}.{mode}"])
            got = d.ddpm_previous(x, ts, float(step), eps, noise=noise, **kw).cpu()
            assert (got - want).abs().max().item() <= 4e-6 * max(1.0, want.abs().max().item()), (i, mode, (got - want).abs().max().item())


def test_decode_uncond_guidance_vs_golden(golden, dev):
    """F11: VQVAE.decode_uncond_guidance (vq_vae.py:147-220) against the reference's own output (both scales on)."""
    z = golden("f11_uncond_guidance")
    model = det_model(VQVAE(base_channels=32, pred_name="unet", num_labels=5))
    with torch.no_grad():
        model.vq.dictionary.copy_(seeded(model.vq.dictionary.shape, 77, 0.35))
    steps = int(z["steps"])
    x_T = seeded((2, 1, 2048), int(z["x_T_seed"]))
    gen = torch.Generator().manual_seed(int(z["noise_seed"]))
    noises = [torch.randn(x_T.shape, generator=gen).to(dev) for _ in range(steps)]
    vq_scale, label_scale = (float(v) for v in z["scales"])
    want = torch.from_numpy(z["x0"])
    # fp32 mode pins the algorithm at the north_star gate.  This fixture is deliberately harsh on a 2-byte mode: the guidance
    # extrapolation base + 1.5 (base - a) + 0.7 (base - b) multiplies the predictor's rounding error by up to 5.4, and 4 coarse
    # steps of an untrained network saturate the clamp (x0 RMS 0.86); fp16 measures 7.3e-3 here (plain 5-step decode: 8.3e-4)
    # ... which is why decode_uncond_guidance runs its predictor in the fp32 mode whatever mode the decoder is set to (vq_vae.py
    # here: "guidance extrapolation"): ONE tolerance for both settings, and the reference's real step count in F11b below.
    for prec in ("fp32", "fp16"):
        model.set_precision(prec)
        got = model.decode_uncond_guidance(torch.from_numpy(z["codes"]).to(dev), torch.from_numpy(z["labels"]).to(dev), steps=steps,
                                           constrain=True, vq_scale=vq_scale, label_scale=label_scale, x_T=x_T.to(dev), noise=noises).cpu()
        gate(f"F11 decode_uncond_guidance 4 steps, decoder mode {prec} (predictor promoted to fp32)", got, want, WAVE_RMS)
    assert model.predictor.precision == "fp16"  # the promotion is per call: the decoder's own mode is untouched
    # ... and so is its device handle: the fp32 handle of the guided call lives beside it (precision_override), nothing is rebuilt
    model.predictor(x_T.to(dev), torch.full((2,), 0.5, device=dev), cond=model.vq.embed(torch.from_numpy(z["codes"]).to(dev)),
                    labels=torch.from_numpy(z["labels"]).to(dev))
    own = model.predictor._handle
    assert own is not None
    model.decode_uncond_guidance(torch.from_numpy(z["codes"]).to(dev), torch.from_numpy(z["labels"]).to(dev), steps=2, constrain=True,
                                 vq_scale=vq_scale, label_scale=label_scale, x_T=x_T.to(dev), noise=noises[:2])
    assert model.predictor._handle is own and model.predictor.precision == "fp16"
    alt = model.predictor._alt_handles["fp32"][0]
    model.decode_uncond_guidance(torch.from_numpy(z["codes"]).to(dev), torch.from_numpy(z["labels"]).to(dev), steps=2, constrain=True,
                                 vq_scale=vq_scale, label_scale=label_scale, x_T=x_T.to(dev), noise=noises[:2])
    assert model.predictor._alt_handles["fp32"][0] is alt  # the guided call's fp32 handle is reused, not rebuilt


def test_decode_uncond_guidance_50_steps_vs_golden(golden, dev):
    """F11b: decode_uncond_guidance at a real step count (50 steps, both scales on), against the reference's own output."""
    z = golden("f11b_uncond_guidance_50")
    model = det_model(VQVAE(base_channels=32, pred_name="unet", num_labels=5))
    with torch.no_grad():
        model.vq.dictionary.copy_(seeded(model.vq.dictionary.shape, 77, 0.35))
    steps = int(z["steps"])
    x_T = seeded((2, 1, 8192), int(z["x_T_seed"]))
    gen = torch.Generator().manual_seed(int(z["noise_seed"]))
    noises = [torch.randn(x_T.shape, generator=gen).to(dev) for _ in range(steps)]
    vq_scale, label_scale = (float(v) for v in z["scales"])
    want = torch.from_numpy(z["x0"])
    for prec in ("fp32", "fp16"):
        model.set_precision(prec)
        got = model.decode_uncond_guidance(torch.from_numpy(z["codes"]).to(dev), torch.from_numpy(z["labels"]).to(dev), steps=steps,
                                           constrain=True, vq_scale=vq_scale, label_scale=label_scale, x_T=x_T.to(dev), noise=noises).cpu()
        gate(f"F11b decode_uncond_guidance 50 steps, decoder mode {prec} (predictor promoted to fp32)", got, want, WAVE_RMS)


def test_unet64_full_size_forward_vs_oracle(dev):
    """BASELINE's model (unet64) at full clip length."""
    model = det_model(DiffusionModel("unet", 64))
    x, ts = seeded((1, 1, 64000), 9), torch.tensor([0.4])
    sd = {"predictor." + k: v.detach() for k, v in model.predictor.state_dict().items()}
    want = ref_cpu.unet_predictor(sd, 64, x, ts)
    for prec, tol in MODES:
        model.set_precision(prec)
        assert rel_rms(model.predictor(x.to(dev), ts.to(dev)).cpu(), want) < tol, prec


def test_sharding_invariance_full_size(dev):
    """Size-independent property at full length: a clip's waveform depends only on (seed, global clip index),
    not on which shard / batch slot it was sampled in (what makes 1/2/4/8-GPU runs identical)."""
    model = det_model(DiffusionModel("unet", 32))
    model.set_precision("bf16")
    T, seed, steps = 64000, 99, 3

    def shard(begin, end):
        x_T = randn_clips(end - begin, T, dev, seed, clip_offset=begin)
        return model.diffusion.ddpm_sample(x_T, model.predictor, steps, constrain=True, seed=seed, clip_offset=begin)

    whole = shard(0, 4)
    parts = torch.cat([shard(0, 1), shard(1, 3), shard(3, 4)], dim=0)
    assert torch.equal(whole, parts)
    assert whole.abs().max().item() <= 1.0 + 1e-6 and torch.isfinite(whole).all()
    # consecutive clips are different draws
    assert rms(whole[0] - whole[1]) > 0.1


def test_counter_rng_is_standard_normal(dev):
    x = randn_clips(8, 64000, dev, 5).double()
    assert abs(x.mean().item()) < 5e-3 and abs(x.var().item() - 1) < 1e-2
    assert abs((x ** 4).mean().item() - 3) < 5e-2
    assert torch.equal(randn_clips(2, 4096, dev, 5, clip_offset=3).cpu(), randn_clips(5, 4096, dev, 5).cpu()[3:5])


def test_decode_uncond_guidance_vs_oracle(dev):
    """vq_vae.py:147-220 (3x-batch classifier-free-style guidance) against the same composition on the oracle."""
    model = det_model(VQVAE(base_channels=32, pred_name="unet", num_labels=4))
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    codes = torch.randint(0, 512, (2, 8), generator=torch.Generator().manual_seed(3))
    labels = torch.tensor([0, 2])
    x_T = seeded((2, 1, 2048), 4)
    gen = torch.Generator().manual_seed(5)
    noises = [torch.randn(x_T.shape, generator=gen) for _ in range(3)]
    cond = ref_cpu.vq_embed(sd["vq.dictionary"], codes)
    cond3 = torch.cat([cond, torch.zeros_like(cond), cond])
    lab3 = torch.cat([labels + 1, labels + 1, torch.zeros_like(labels)])

    def pred(xs, ts):
        o = ref_cpu.unet_predictor(sd, 32, torch.cat([xs] * 3), torch.cat([ts] * 3), cond=cond3, labels=lab3)
        base = o[:2]
        return base + 1.5 * (base - o[2:4]) + 0.7 * (base - o[4:6])

    want = ref_cpu.ddpm_sample("exp", x_T, pred, 3, noises, constrain=True)
    got = model.decode_uncond_guidance(codes.to(dev), labels.to(dev), steps=3, constrain=True, vq_scale=1.5, label_scale=0.7,
                                       x_T=x_T.to(dev), noise=[n.to(dev) for n in noises]).cpu()
    gate("decode_uncond_guidance 3 steps vs oracle", got, want, WAVE_RMS)


@pytest.mark.parametrize("prec,tol", MODES)
def test_resblock_config_sweep_vs_oracle(dev, prec, tol):
    """Every kernel variant the schedule can pick (32/64/128-channel tiles, identity / 1x1 skip, avg / up resize,
    small and large dilation halo, ragged lengths, several clips) against the oracle on seeded inputs."""
    cases = [
        # cin, cout, scale, dil, emb, L, B
        (32, 32, 1.0, 1, 128, 300, 2),
        (64, 128, 1.0, 2, 128, 700, 1),
        (128, 128, 1.0, 2, 256, 253, 3),     # exactly one tile + 1 row
        (128, 128, 0.5, 2, 128, 504, 2),     # avg-pool, identity skip
        (128, 128, 2.0, 2, 128, 252, 2),     # upsample, identity skip
        (256, 256, 1.0, 8, 256, 250, 2),     # big-halo variant, 128-channel tiles
        (512, 512, 1.0, 32, 256, 250, 1),    # pre-transformed (raw) segments, dilation 32
        (768, 256, 1.0, 2, 256, 500, 1),     # 1x1 skip conv over a wide input
        (96, 32, 1.0, 2, None, 1000, 2),     # no FiLM, 32-channel tiles, 3 chunks
        (160, 64, 1.0, 4, 128, 129, 2),      # odd chunk count, dilation 4
    ]
    for i, (cin, cout, scale, dil, emb, L, B) in enumerate(cases):
        m = ResBlockModule(cin, emb, cout if cout != cin else None, scale, dil)
        det_init_((f"sweep{i}." + k, v) for k, v in m.block.state_dict().items())
        m.set_precision(prec)
        x = seeded((B, cin, L), 1000 + i)
        e = seeded((B, emb), 2000 + i) if emb else None
        sd = {"b." + k: v.detach() for k, v in m.block.state_dict().items()}
        want = ref_cpu.res_block(x, sd, "b", dict(cin=cin, cout=cout, scale=scale, dil=dil), e)
        got = m(x.to(dev), None if e is None else e.to(dev)).cpu()
        assert got.shape == want.shape and rel_rms(got, want) < tol, (i, cin, cout, scale, dil, L, B, rel_rms(got, want))


def test_resblock_random_configs_vs_oracle(dev):
    """Seeded random ResBlocks (channel counts, lengths around the tile boundaries, dilations, avg / up resizing, FiLM on / off,
    several clips per workgroup) in the two gate modes: the sweep of tools/fuzz_resblock.py with a fixed seed.  Covers the
    wave-specialised convolution kernel's segment / tile / clip bookkeeping well beyond the shapes of the UNets."""
    import random
    rng = random.Random(11)
    worst = {"fp32": 0.0, "fp16": 0.0}
    for i in range(28):
        cin = rng.choice([32, 64, 96, 128, 192, 256, 384, 512])
        scale = rng.choice([1.0, 1.0, 0.5, 2.0])
        cout = cin if scale != 1.0 else rng.choice([cin, 64, 128, 256, 512])
        dil = 2 if scale == 2.0 else rng.choice([1, 2, 2, 4, 8, 16, 32])
        emb = rng.choice([None, 128, 256])
        L = rng.choice([2, 6, 64, 126, 250, 252, 254, 256, 258, 500, 508, 510, 1000, 1024, rng.randrange(2, 1500) * 2])
        B = rng.choice([1, 2, 3, 5])
        m = ResBlockModule(cin, emb, cout if cout != cin else None, scale, dil)
        det_init_((f"rnd{i}." + k, v) for k, v in m.block.state_dict().items())
        x = seeded((B, cin, L), 7000 + i)
        e = seeded((B, emb), 8000 + i) if emb else None
        sd = {"b." + k: v.detach() for k, v in m.block.state_dict().items()}
        want = ref_cpu.res_block(x, sd, "b", dict(cin=cin, cout=cout, scale=scale, dil=dil), e)
        for prec, tol in (("fp32", FP32_REL), ("fp16", FP16_REL)):
            m.set_precision(prec)
            got = m(x.to(dev), None if e is None else e.to(dev)).cpu()
            err = rel_rms(got, want) if got.shape == want.shape else float("inf")
            worst[prec] = max(worst[prec], err)
            assert err < tol, (prec, dict(cin=cin, cout=cout, scale=scale, dil=dil, emb=emb, L=L, B=B), err)


def test_handle_lifecycle_and_small_shapes(dev):
    """The device handle is rebuilt when batch / length grow or precision changes; tiny shapes work; memory is returned."""
    model = det_model(DiffusionModel("unet", 32))
    sd = {"predictor." + k: v.detach() for k, v in model.predictor.state_dict().items()}
    free0 = torch.cuda.mem_get_info()[0]
    for B, T in ((1, 256), (3, 512), (2, 256), (5, 1024)):
        x, ts = seeded((B, 1, T), 10 * B + T), torch.linspace(0.1, 0.9, B)
        want = ref_cpu.unet_predictor(sd, 32, x, ts)
        got = model.predictor(x.to(dev), ts.to(dev)).cpu()
        assert rel_rms(got, want) < FP32_REL, (B, T)
    h = model.predictor._handle
    assert h.cfg.max_batch >= 5 and h.cfg.max_T >= 1024 and h.kernel_count() > 250
    assert h.model_bytes(2, 1024) == 2 * h.model_bytes(1, 1024) and h.flops(1, 2048) == 2 * h.flops(1, 1024)
    model.set_precision("bf16")
    assert model.predictor._handle is None
    model.predictor(seeded((1, 1, 256), 1).to(dev), torch.tensor([0.5], device=dev))
    model.predictor.invalidate()
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] >= free0 - (64 << 20), "device memory of destroyed handles was not returned"


def test_live_parameters_deepcopy_and_index_errors(dev):
    """The reference reads live parameters on every call: an in-place weight update between two forwards must be seen
    (the device snapshot is rebuilt); EMA-style deep copies work after a forward; out-of-range labels / codes raise like
    nn.Embedding / F.embedding instead of being clamped."""
    import copy
    import pickle

    model = det_model(DiffusionModel("unet", 32))
    x, ts = seeded((2, 1, 512), 3).to(dev), torch.tensor([0.3, 0.8], device=dev)
    a = model.predictor(x, ts)
    with torch.no_grad():  # out = Conv(gelu(GN(h))): doubling weight and bias doubles the output exactly
        model.predictor.out[1].weight.mul_(2.0)
        model.predictor.out[1].bias.mul_(2.0)
    b = model.predictor(x, ts)
    assert torch.equal(b, 2.0 * a)
    ema = copy.deepcopy(model)
    assert torch.equal(ema.predictor(x, ts), b)
    # a REPLACED parameter object is seen too (the module tree is walked on every call) ...
    conv = model.predictor.out[1]
    conv.weight = torch.nn.Parameter(conv.weight.detach() * 0.5)
    conv.bias = torch.nn.Parameter(conv.bias.detach() * 0.5)
    assert torch.equal(model.predictor(x, ts), a)
    # ... a write through .data is not (no version bump): the documented contract is invalidate()
    conv.weight.data.mul_(2.0)
    conv.bias.data.mul_(2.0)
    model.predictor.invalidate()
    assert torch.equal(model.predictor(x, ts), b)
    with torch.no_grad():
        conv.weight.mul_(0.5)
        conv.bias.mul_(0.5)
    pickle.dumps(model.predictor)  # the ctypes handle is dropped from the pickled state
    assert torch.equal(model.predictor(x, ts), a)
    src = det_model(DiffusionModel("unet", 32))
    assert model.load_from_pretrained(src) > 0
    assert torch.equal(model.predictor(x, ts), a)
    lab = det_model(DiffusionModel("unet", 32, num_labels=3))
    with pytest.raises(IndexError):
        lab.predictor(x, ts, labels=torch.tensor([0, 3], device=dev))
    vq = VQVAE(base_channels=32, pred_name="unet").vq
    with pytest.raises(IndexError):
        vq.embed(torch.tensor([[0, 512]], device=dev))
    with pytest.raises(IndexError):
        vq.embed(torch.tensor([[-1, 5]], device=dev))


def test_random_topologies_seeded_subset(dev):
    """A seeded 20-case subset of tools/fuzz_topology.py (random widths, channel_mult, depth_mult, dilations, labels, conditioning of
    random length, input channels, batch; predictor in both gate modes, encoder in fp32) against the oracle: the developer sweep's
    generator, run here as a test so that the open topology (reference unet.py:17-30, 188-196) is held by GPUTEST, not by a tool."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_topology.py"), "2026", "20"], capture_output=True, text=True, timeout=1500)
    tail = r.stdout[-3000:] + r.stderr[-3000:]
    assert r.returncode == 0 and "cases 20" in r.stdout and r.stdout.rstrip().endswith("bad 0"), tail


def test_widths_that_are_not_multiples_of_32_vs_oracle(dev):
    """The reference takes ANY base_channels (unet.py:17-30; its GroupNorm halves the group count until it divides the width,
    unet.py:345-349).  Widths that are not multiples of 32 are built at a padded physical width whose extra channels stay exactly
    zero (csrc/net.cpp pad_map: 48 -> 64, 40 -> 64, 24 -> 32, 20 -> 32, 100 -> 128), with the reference's group structure: predictor
    (labels, conditioning, FiLM, concatenating up blocks, default and custom topologies) and encoder against the oracle."""
    from vq_voice_swap_amd import UNetEncoder, UNetPredictor

    cases = [(48, dict(), dict(), 16384),  # the default nine-level topology at base 48 (widths 48 .. 384, built as 64 .. 512)
             (40, dict(channel_mult=(1, 2, 2, 4), middle_dilations=(2, 8), depth_mult=1), dict(num_labels=3, cond_channels=64), 4096),
             (24, dict(channel_mult=(1, 1, 2), middle_dilations=(4,), depth_mult=2), dict(num_labels=2), 2048),
             (20, dict(channel_mult=(1, 3, 4), middle_dilations=(), depth_mult=1), dict(in_channels=2), 1024),
             (100, dict(channel_mult=(1, 2), middle_dilations=(3,), depth_mult=1), dict(cond_channels=32), 1024)]
    for i, (base, topo, kw, T) in enumerate(cases):
        m = UNetPredictor(base, **topo, **kw)
        det_init_((f"predictor.w{base}." + k, v) for k, v in m.state_dict().items())
        m.eval()
        sd = {"predictor." + k: v.detach().clone() for k, v in m.state_dict().items()}
        x, ts = seeded((2, kw.get("in_channels", 1), T), 1200 + i), torch.tensor([0.25, 0.7])
        call = {}
        if "num_labels" in kw:
            call["labels"] = torch.tensor([1, 0])
        if "cond_channels" in kw:
            call["cond"] = seeded((2, kw["cond_channels"], 11), 1300 + i, 0.5)
        want = ref_cpu.unet_predictor(sd, base, x, ts, topology=topo or None, **call)
        for prec, tol in (("fp32", FP32_REL), ("fp16", 6e-3)):
            m.set_precision(prec)
            got = m(x.to(dev), ts.to(dev), **{k: v.to(dev) for k, v in call.items()}).cpu()
            assert got.shape == want.shape and rel_rms(got, want) < tol, (base, prec, rel_rms(got, want))
        m.invalidate()
    for i, (base, topo, oc, T) in enumerate([(40, dict(channel_mult=(1, 2, 4), out_dilations=(2,), depth_mult=1), 64, 2048),
                                             (24, dict(), 96, 16384)]):
        e = UNetEncoder(base, out_channels=oc, **topo)
        det_init_((f"encoder.w{base}." + k, v) for k, v in e.state_dict().items())
        e.eval()
        sd = {"encoder." + k: v.detach().clone() for k, v in e.state_dict().items()}
        x = seeded((2, 1, T), 1400 + i, 0.3)
        want = ref_cpu.unet_encoder(sd, base, x, topology=topo or None)
        got = e(x.to(dev)).cpu()
        assert got.shape == want.shape and rel_rms(got, want) < FP32_REL, (base, rel_rms(got, want))
        e.invalidate()
