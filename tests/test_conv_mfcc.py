"""ConvMFCCEncoder (reference models/conv_encoder.py:14-133, scope row 8f.4).

PARITY UNPINNED for the MFCC front end: the reference builds it from torchaudio, which is not installed where the golden
vectors are generated, so there is no fixture from the reference itself.  What is checked instead:
  * CPU: the oracle's front end (oracle/ref_cpu.py, torch.stft based) against an independent numpy restatement of the
    published algorithm (explicit reflect padding, framing, rfft, filter bank, log / dB, DCT) -- pins the framing,
    normalisation and dB conventions against a second derivation; the state-dict layout of the parameter container;
  * GPU: the HIP path (through the C ABI) against the oracle: feature rows, encoder output, VQ codes, and the conditional
    predictor forward with T/320 conditioning rows (nearest up-sampling as F.interpolate, unet.py:139)."""
import math

import numpy as np
import pytest
import torch

from oracle import ref_cpu
from vq_voice_swap_amd import ConvMFCCEncoder, VQVAE
from vq_voice_swap_amd.det_init import det_init_

from util import rel_rms, seeded

torch.set_num_threads(8)


def numpy_mfcc(wave: np.ndarray, cfg: dict) -> np.ndarray:
    """Independent restatement of torchaudio.transforms.MFCC for one [T] waveform (float64 throughout)."""
    n_fft, hop, n_mels, sr = cfg["n_fft"], cfg["hop"], cfg["n_mels"], cfg["sample_rate"]
    x = np.pad(wave.astype(np.float64), n_fft // 2, mode="reflect")
    frames = 1 + (len(x) - n_fft) // hop
    win = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n_fft) / n_fft)  # periodic Hann
    spec = np.stack([np.fft.rfft(x[f * hop:f * hop + n_fft] * win) for f in range(frames)], axis=1)  # [freq, frames]
    if cfg["normalized"]:
        spec = spec / np.sqrt((win ** 2).sum())
    power = np.abs(spec) ** 2
    # HTK mel filter bank, norm=None
    hz2mel = lambda f: 2595.0 * np.log10(1.0 + f / 700.0)  # noqa: E731
    mel2hz = lambda m: 700.0 * (10 ** (m / 2595.0) - 1.0)  # noqa: E731
    freqs = np.linspace(0, sr // 2, n_fft // 2 + 1)
    pts = mel2hz(np.linspace(hz2mel(0.0), hz2mel(sr // 2), n_mels + 2))
    fb = np.zeros((len(freqs), n_mels))
    for m in range(n_mels):
        lo, ce, hi = pts[m], pts[m + 1], pts[m + 2]
        fb[:, m] = np.maximum(0.0, np.minimum((freqs - lo) / (ce - lo), (hi - freqs) / (hi - ce)))
    mel = fb.T @ power
    if cfg["log_mels"]:
        mel = np.log(mel + 1e-6)
    else:
        mel = 10.0 * np.log10(np.maximum(mel, 1e-10))
        mel = np.maximum(mel, mel.max() - 80.0)
    k = np.arange(cfg["n_mfcc"])[:, None]
    dct = np.cos(np.pi / n_mels * (np.arange(n_mels)[None] + 0.5) * k) * np.sqrt(2.0 / n_mels)
    dct[0] *= 1.0 / np.sqrt(2.0)
    return dct @ mel


@pytest.mark.parametrize("version", [1, 2])
def test_oracle_front_end_matches_independent_restatement(version):
    cfg = ref_cpu.mfcc_config(version)
    bufs = ref_cpu.mfcc_buffers(cfg)
    wave = (0.3 * seeded((1, 4000), 5) + 0.5 * torch.sin(torch.arange(4000) * 2 * math.pi * 440 / 16000)).clamp(-1, 1)
    got = ref_cpu.mfcc_transform(wave, bufs, cfg)[0].double().numpy()
    want = numpy_mfcc(wave[0].numpy(), cfg)
    assert got.shape == want.shape == (13, 4000 // 160 + 1)
    assert np.abs(got - want).max() <= 2e-3 * max(1.0, np.abs(want).max()), np.abs(got - want).max()


def test_parameter_container_layout():
    """State-dict keys and shapes a reference checkpoint of `VQVAE(enc_name="conv-mfcc-ulaw")` carries for its encoder
    (conv_encoder.py:42-88 plus the persistent buffers of torchaudio.transforms.MFCC)."""
    e = ConvMFCCEncoder(32, out_channels=512)
    sd = e.state_dict()
    mid = 384
    want = {"mfcc.dct_mat": (40, 13), "mfcc.MelSpectrogram.spectrogram.window": (320,), "mfcc.MelSpectrogram.mel_scale.fb": (161, 40),
            "blocks.0.0.weight": (mid, 39, 3), "blocks.0.0.bias": (mid,), "blocks.1.conv.weight": (mid, mid, 3), "blocks.1.conv.bias": (mid,),
            "blocks.2.0.weight": (mid, mid, 4), "blocks.2.0.bias": (mid,), "blocks.9.weight": (512, mid, 1), "blocks.9.bias": (512,)}
    for i in range(3, 9):
        want[f"blocks.{i}.conv.weight"] = (mid, mid, 3 if i <= 4 else 1)
        want[f"blocks.{i}.conv.bias"] = (mid,)
    assert {k: tuple(v.shape) for k, v in sd.items()} == want
    assert float(sd["blocks.9.weight"].abs().max()) == 0.0  # zero-initialised output (conv_encoder.py:85-88)
    assert e.downsample_rate == 320 and e.out_length(64000) == 200
    v2 = ConvMFCCEncoder(32, version=2).state_dict()
    assert tuple(v2["mfcc.MelSpectrogram.mel_scale.fb"].shape) == (201, 80) and tuple(v2["mfcc.MelSpectrogram.spectrogram.window"].shape) == (400,)
    m = VQVAE(base_channels=32, enc_name="conv-mfcc-ulaw", pred_name="unet")
    assert m.save_kwargs()["enc_name"] == "conv-mfcc-ulaw" and m.downsample_rate == 1280
    with pytest.raises(RuntimeError):
        e(torch.zeros(1, 1, 64000))  # no CPU fallback


def det_encoder(enc_name, base=32):
    model = VQVAE(base_channels=base, enc_name=enc_name, pred_name="unet", num_labels=3)
    det_init_((k, v) for k, v in model.state_dict().items() if ".mfcc." not in k)  # the transform's constant tensors stay
    with torch.no_grad():
        model.vq.dictionary.copy_(seeded(model.vq.dictionary.shape, 79, 0.35))
    return model.eval()


@pytest.mark.gpu
@pytest.mark.parametrize("enc_name,version,ulaw", [("conv-mfcc-ulaw", 1, True), ("conv-mfcc-ulaw-v2", 2, True), ("conv-mfcc-linear", 1, False)])
def test_hip_encoder_vs_oracle(enc_name, version, ulaw):
    dev = torch.device("cuda:0")
    model = det_encoder(enc_name)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    enc = model.encoder
    for B, T, seed in ((3, 64000, 1), (2, 4000, 2), (1, 4160, 3)):  # 401 / 26 / 27 frames: odd and even frame counts
        x = (0.4 * seeded((B, 1, T), seed)).clamp(-1, 1)
        feats = {}
        want = ref_cpu.conv_mfcc_encoder(sd, x, version=version, input_ulaw=ulaw, probe=lambda n, t: feats.__setitem__(n, t))
        enc.debug_taps = True
        got = enc(x.to(dev)).cpu()
        assert got.shape == want.shape == (B, 512, enc.out_length(T))
        h = enc._handle
        names = [n for n, _, _ in h.taps()]
        f = h.read_tap(names.index("features"), B, T)[:, :39]
        # log / dB features of noise-floor bins amplify the fp32 FFT's rounding: compare in absolute terms against their range
        assert (f - feats["features"]).abs().max().item() <= 5e-3 * feats["features"].abs().max().item(), (enc_name, T)
        assert rel_rms(got, want) < 2e-3, (enc_name, B, T, rel_rms(got, want))
    # Codes through the VQ layer.  (a) The VQ layer itself is BIT-EXACT: the codes of the HIP path are the reference argmin
    # (vq.py:199-221 order, first index on ties) of the HIP encoder's own z.  (b) Against the oracle's codes the only admissible
    # difference is a PROVEN near-tie: with delta = z_hip - z_oracle at a position, replacing z by z + delta moves the difference
    # of two squared distances by exactly 2 delta . (e2 - e1), so the argmin can only move from e1 to e2 where the oracle's top
    # gap d(z, e2) - d(z, e1) is at most 2 |delta| |e2 - e1| (+ the fp32 rounding of the distances themselves, 1e-5 relative).
    # The front end is where delta comes from (fp64 DFT here, fp32 FFT in torchaudio / the oracle: "parity unpinned" for the MFCC
    # transform); everything else is asserted exactly.
    x = (0.4 * seeded((2, 1, 64000), 9)).clamp(-1, 1)
    z = ref_cpu.conv_mfcc_encoder(sd, x, version=version, input_ulaw=ulaw)
    dic = sd["vq.dictionary"]
    codes_want = ref_cpu.vq_encode(dic, z)
    enc.debug_taps = False
    z_hip = model.encoder(x.to(dev)).cpu()
    codes = model.encode(x.to(dev)).cpu()
    assert codes.shape == (2, 200)
    assert torch.equal(codes, ref_cpu.vq_encode(dic, z_hip)), "VQ indices are not the reference argmin of the HIP encoder's output"
    mism = codes != codes_want
    zf, zh = z.permute(0, 2, 1).reshape(-1, z.shape[1]), z_hip.permute(0, 2, 1).reshape(-1, z.shape[1])
    d = ref_cpu.vq_distances(dic, zf)
    for pos in torch.nonzero(mism.reshape(-1)).flatten().tolist():
        e1, e2 = dic[codes_want.reshape(-1)[pos]], dic[codes.reshape(-1)[pos]]
        gap = (d[pos, codes.reshape(-1)[pos]] - d[pos, codes_want.reshape(-1)[pos]]).item()
        bound = 2.0 * (zh[pos] - zf[pos]).norm().item() * (e2 - e1).norm().item() + 1e-5 * d[pos].abs().max().item()
        print(f"[near-tie] {enc_name} position {pos}: oracle gap {gap:.3e} <= bound {bound:.3e}")
        assert 0.0 <= gap <= bound, (enc_name, pos, gap, bound)
    assert mism.sum().item() <= 4, int(mism.sum())


@pytest.mark.gpu
def test_predictor_with_mfcc_rate_conditioning_vs_oracle():
    """cond rows at T/320 (200 for 4 s): nearest up-sampling to T in in_conv, as F.interpolate(cond, T) (unet.py:139)."""
    dev = torch.device("cuda:0")
    model = det_encoder("conv-mfcc-ulaw")
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    T = 64000
    x, ts = seeded((2, 1, T), 11), torch.tensor([0.6, 0.25])
    codes = torch.randint(0, 512, (2, 200), generator=torch.Generator().manual_seed(12))
    labels = torch.tensor([2, 0])
    cond = ref_cpu.vq_embed(sd["vq.dictionary"], codes)
    want = ref_cpu.unet_predictor(sd, 32, x, ts, cond=cond, labels=labels)
    got = model.predictor(x.to(dev), ts.to(dev), cond=model.vq.embed(codes.to(dev)), labels=labels.to(dev)).cpu()
    assert rel_rms(got, want) < 1e-4
    # and back to the UNet-encoder rate on the same module: the handle is rebuilt for the other conditioning length
    cond250 = ref_cpu.vq_embed(sd["vq.dictionary"], torch.randint(0, 512, (2, 250), generator=torch.Generator().manual_seed(13)))
    want = ref_cpu.unet_predictor(sd, 32, x, ts, cond=cond250, labels=labels)
    got = model.predictor(x.to(dev), ts.to(dev), cond=cond250.to(dev), labels=labels.to(dev)).cpu()
    assert rel_rms(got, want) < 1e-4
    # whole conversion path: encode -> decode shapes (sample_vqvae.py:36-54 with this encoder)
    wav = (0.3 * seeded((1, 1, T), 14)).clamp(-1, 1)
    out = model.decode(model.encode(wav.to(dev)), torch.tensor([1], device=dev), steps=2, constrain=True, seed=5)
    assert out.shape == (1, 1, T) and bool(torch.isfinite(out).all())


@pytest.mark.gpu
def test_hip_conv_stack_vs_reference_fixture(golden):
    """F12 through the C ABI (vqvs_mfcc_encoder_forward_logmel): deltas x 2, concatenation and the convolution stack of the HIP
    encoder against the REFERENCE's own output on an injected MFCC tensor (conv_encoder.py:97-110), and -- with a dictionary made of
    the reference's z vectors, so that every position has a guaranteed margin -- VQ codes with ZERO mismatches."""
    dev = torch.device("cuda:0")
    z = golden("f12_conv_mfcc_stack")
    for tag, enc_name, T in (("ulaw_even", "conv-mfcc-ulaw", 4000), ("ulaw_odd", "conv-mfcc-ulaw", 4160), ("linear_even", "conv-mfcc-linear", 64000)):
        model = det_encoder(enc_name)
        want = torch.from_numpy(z[tag + ".z"])
        got = model.encoder.forward_from_mfcc(torch.from_numpy(z[tag + ".mfcc"]).to(dev), T).cpu()
        assert got.shape == want.shape
        assert rel_rms(got, want) < 1e-4, (tag, rel_rms(got, want))
        # VQ on margin-guaranteed inputs: the dictionary holds the reference's own z vectors (+ far-away fillers)
        vecs = want.permute(0, 2, 1).reshape(-1, want.shape[1])
        dic = 10.0 * seeded((512, want.shape[1]), 555)
        n = min(len(vecs), 512)
        dic[:n] = vecs[:n]
        codes_want = ref_cpu.vq_encode(dic, want)
        d = ref_cpu.vq_distances(dic, vecs)
        top2 = torch.topk(d, 2, dim=-1, largest=False).values
        assert ((top2[:, 1] - top2[:, 0]) > 1e-3 * top2[:, 1].abs().clamp_min(1e-6))[:n].all(), "fixture margins"
        with torch.no_grad():
            model.vq.dictionary.copy_(dic)
        codes = model.vq.encode(got.to(dev)).cpu()
        assert torch.equal(codes.reshape(-1)[:n], codes_want.reshape(-1)[:n]), tag
