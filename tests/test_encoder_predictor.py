"""EncoderPredictor guidance (scope row 8f.1b): logits and the input gradient of the summed cross-entropy through the
whole UNet (explicit HIP backward schedule) against the reference's own outputs (fixture F10) and the oracle; guided
VQ-VAE decoding against the oracle."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu
from vq_voice_swap_amd import EncoderPredictor, VQVAE
from vq_voice_swap_amd.det_init import det_init_

from util import gate, rel_rms, rms, seeded


def make_encpred(num_latents=96):
    ep = EncoderPredictor(base_channels=32, downsample_rate=256, num_latents=num_latents, bottleneck_dim=64)
    det_init_(ep.state_dict().items())
    ep.eval()
    return ep


def test_encpred_oracle_matches_reference_golden(golden, tmp_path):
    z = golden("f10_encpred32")
    ep = make_encpred()
    sd = {k: v.detach() for k, v in ep.state_dict().items()}
    x = seeded((2, 1, 16384), int(z["x_seed"]))
    ts, targets = torch.from_numpy(z["ts"]), torch.from_numpy(z["targets"])
    assert torch.equal(ref_cpu.encoder_predictor(sd, 32, x, ts, 256), torch.from_numpy(z["logits"]))
    assert torch.equal(ref_cpu.encoder_predictor_cond_fn(sd, 32, 256, targets)(x, ts), torch.from_numpy(z["grad"]))
    p = str(tmp_path / "ep.pt")
    ep.save(p)
    ep2 = EncoderPredictor.load(p)
    assert all(torch.equal(a, b) for a, b in zip(ep.state_dict().values(), ep2.state_dict().values()))
    with pytest.raises(RuntimeError):  # no CPU path
        ep(x, ts)


@pytest.mark.gpu
@pytest.mark.parametrize("precision,tol_logit,tol_grad", [("fp32", 2e-4, 2e-3), ("fp16", 8e-3, 4e-2), ("bf16", 5e-2, 2e-1)])
def test_native_encpred_vs_golden(golden, precision, tol_logit, tol_grad):
    z = golden("f10_encpred32")
    dev = torch.device("cuda:0")
    ep = make_encpred().to(dev)
    ep.set_precision(precision)
    x = seeded((2, 1, 16384), int(z["x_seed"])).to(dev)
    ts, targets = torch.from_numpy(z["ts"]).to(dev), torch.from_numpy(z["targets"]).to(dev)
    logits = ep(x, ts).cpu()
    assert rel_rms(logits, torch.from_numpy(z["logits"])) < tol_logit
    grad = ep.guidance_grad(x, ts, targets, 1.0)
    assert rel_rms(grad.cpu(), torch.from_numpy(z["grad"])) < tol_grad
    assert torch.equal(ep.guidance_grad(x, ts, targets, 1.0), grad)  # deterministic
    assert rel_rms(ep.guidance_fn(targets, 0.5)(x, ts).cpu(), 0.5 * grad.cpu()) < (1e-4 if precision == "fp32" else 2e-2)
    want_losses = torch.nn.functional.cross_entropy(torch.from_numpy(z["logits"]), torch.from_numpy(z["targets"]), reduction="none").mean(-1)
    assert rel_rms(ep.losses(x, ts, targets).cpu(), want_losses) < tol_logit


@pytest.mark.gpu
def test_encpred_guided_decode_vs_oracle():
    """VQVAE.decode(enc_pred=, enc_pred_scale=) (vq_vae.py:92-145) against the same composition on the oracle."""
    dev = torch.device("cuda:0")
    model = VQVAE(base_channels=32, enc_name="unet", pred_name="unet", num_labels=5)
    det_init_(model.state_dict().items())
    model.eval()
    ep = make_encpred(num_latents=512)
    sd_m = {k: v.detach().clone() for k, v in model.state_dict().items()}
    sd_e = {k: v.detach().clone() for k, v in ep.state_dict().items()}
    gen = torch.Generator().manual_seed(81)
    codes = torch.randint(0, 512, (2, 16), generator=gen)
    labels = torch.tensor([1, 4])
    x_T = seeded((2, 1, 4096), 82)
    noises = [torch.randn(x_T.shape, generator=gen) for _ in range(3)]
    scale = 30.0
    cond = ref_cpu.vq_embed(sd_m["vq.dictionary"], codes)
    targets = ref_cpu.vq_encode(sd_m["vq.dictionary"], cond)
    want = ref_cpu.vqvae_decode(sd_m, 32, "exp", codes, labels, 3, x_T, noises, constrain=True,
                                cond_fn=ref_cpu.encoder_predictor_cond_fn(sd_e, 32, 256, targets, scale))
    plain = ref_cpu.vqvae_decode(sd_m, 32, "exp", codes, labels, 3, x_T, noises, constrain=True)
    model.to(dev)
    ep.to(dev)
    got = model.decode(codes.to(dev), labels.to(dev), steps=3, constrain=True, enc_pred=ep, enc_pred_scale=scale, x_T=x_T.to(dev),
                       noise=[n.to(dev) for n in noises]).cpu()
    gate("encoder-predictor-guided vqvae32 decode 3 steps vs oracle (fp32)", got, want, 1e-3)
    assert rms(want - plain) > 10 * rms(got - want), "guidance term too small for the comparison to mean anything"


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,rate", [(1, 2048, 256), (3, 4096, 512)])
def test_encpred_small_shapes_vs_oracle(B, T, rate):
    """single clip / odd batch / a coarser latent rate, per-clip timesteps; handle re-used across shapes."""
    dev = torch.device("cuda:0")
    ep = EncoderPredictor(base_channels=32, downsample_rate=rate, num_latents=40, bottleneck_dim=32)
    det_init_(ep.state_dict().items())
    ep.eval()
    sd = {k: v.detach().clone() for k, v in ep.state_dict().items()}
    x = seeded((B, 1, T), 101 + B)
    ts = torch.linspace(0.2, 0.9, B)
    targets = torch.randint(0, 40, (B, T // rate), generator=torch.Generator().manual_seed(5))
    want = ref_cpu.encoder_predictor(sd, 32, x, ts, rate)
    want_g = ref_cpu.encoder_predictor_cond_fn(sd, 32, rate, targets, 2.0)(x, ts)
    ep.to(dev)
    assert rel_rms(ep(x.to(dev), ts.to(dev)).cpu(), want) < 2e-4
    assert rel_rms(ep.guidance_grad(x.to(dev), ts.to(dev), targets.to(dev), 2.0).cpu(), want_g) < 2e-3
    with pytest.raises(ValueError):
        ep.guidance_grad(x.to(dev), ts.to(dev), targets[:, :-1].to(dev))


@pytest.mark.gpu
def test_base64_guidance_models_vs_oracle():
    """Width 64 (512-channel levels: stand-alone prologue kernels, LDS-DMA staging, 8-head attention pool) for both
    guidance models, forward and input gradient, fp32 mode."""
    from vq_voice_swap_amd import Classifier

    dev = torch.device("cuda:0")
    ts = torch.tensor([0.2, 0.7])
    clf = Classifier(num_labels=11, base_channels=64)
    det_init_(clf.state_dict().items())
    clf.eval()
    sd = {k: v.detach().clone() for k, v in clf.state_dict().items()}
    x = seeded((2, 1, 2048), 3)
    labels = torch.tensor([3, 9])
    want, want_g = ref_cpu.classifier(sd, 64, x, ts), ref_cpu.classifier_cond_fn(sd, 64, labels, 1.0)(x, ts)
    clf.to(dev)
    g, lg = clf.log_prob_grad(x.to(dev), ts.to(dev), labels.to(dev), 1.0, return_logits=True)
    assert rel_rms(lg.cpu(), want) < 2e-4 and rel_rms(g.cpu(), want_g) < 2e-3
    ep = EncoderPredictor(64, 256, 64)
    det_init_(ep.state_dict().items())
    ep.eval()
    sd = {k: v.detach().clone() for k, v in ep.state_dict().items()}
    tg = torch.randint(0, 64, (2, 8), generator=torch.Generator().manual_seed(1))
    want, want_g = ref_cpu.encoder_predictor(sd, 64, x, ts, 256), ref_cpu.encoder_predictor_cond_fn(sd, 64, 256, tg, 1.0)(x, ts)
    ep.to(dev)
    assert rel_rms(ep(x.to(dev), ts.to(dev)).cpu(), want) < 2e-4
    assert rel_rms(ep.guidance_grad(x.to(dev), ts.to(dev), tg.to(dev)).cpu(), want_g) < 2e-3


@pytest.mark.gpu
def test_encpred_width_96_vs_oracle():
    """EncoderPredictor at a base width that is not a power of two (the reference takes any, encoder_predictor.py:25-41): logits and
    the input gradient of the summed cross-entropy through the whole UNet against the oracle."""
    dev = torch.device("cuda:0")
    ep = EncoderPredictor(base_channels=96, downsample_rate=256, num_latents=40, bottleneck_dim=64)
    det_init_(("ep96." + k, v) for k, v in ep.state_dict().items())
    ep.eval()
    sd = {k: v.detach().clone() for k, v in ep.state_dict().items()}
    T = 16384
    x, ts = seeded((2, 1, T), 961), torch.tensor([0.3, 0.8])
    targets = torch.randint(0, 40, (2, T // 256), generator=torch.Generator().manual_seed(962))
    want = ref_cpu.encoder_predictor(sd, 96, x, ts, 256)
    want_g = ref_cpu.encoder_predictor_cond_fn(sd, 96, 256, targets, 1.0)(x, ts)
    ep.to(dev)
    for prec, tl, tg in (("fp32", 2e-4, 2e-3), ("fp16", 8e-3, 4e-2)):
        ep.set_precision(prec)
        assert rel_rms(ep(x.to(dev), ts.to(dev)).cpu(), want) < tl, prec
        g = ep.guidance_grad(x.to(dev), ts.to(dev), targets.to(dev), 1.0).cpu()
        assert rel_rms(g, want_g) < tg, (prec, rel_rms(g, want_g))
