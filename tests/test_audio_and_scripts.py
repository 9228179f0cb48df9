"""Caller rows C1/C2 of the scope table: WAV I/O + mu-law (CPU), the sample scripts end to end (GPU)."""
import os
import sys

import numpy as np
import pytest
import torch

from vq_voice_swap_amd.audio import ChunkReader, ChunkWriter, decode_u_law, encode_u_law, parse_time_schedule

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_wav_roundtrip_matches_reference_quantisation(tmp_path):
    x = np.linspace(-1.2, 1.2, 16000).astype(np.float32)
    p = str(tmp_path / "a.wav")
    w = ChunkWriter(p, 16000)
    w.write(x[:7000])
    w.write(x[7000:])
    w.close()
    r = ChunkReader(p, 16000)
    a = r.read(10000)
    b = r.read(10000)
    c = r.read(10)
    r.close()
    assert len(a) == 10000 and len(b) == 6000 and c is None
    got = np.concatenate([a, b])
    want = (np.clip(x, -1, 1) * (2 ** 15 - 1)).astype("int16").astype("float32") / 2 ** 15  # dataset.py:296-298, 222-223
    assert np.array_equal(got, want)


def _write_wav(path, x, rate, bits, channels=1, float_fmt=False):
    import struct

    x = np.asarray(x, dtype=np.float64)
    frames = np.stack([x * (1.0 if c == 0 else 0.5) for c in range(channels)], axis=1)  # (other channels: the same tone at half level)
    if float_fmt:
        body, tag = frames.astype("<f4").tobytes(), 3
    elif bits == 8:
        body, tag = (np.clip(np.rint(frames * 128) + 128, 0, 255)).astype(np.uint8).tobytes(), 1
    elif bits == 16:
        body, tag = np.clip(np.rint(frames * 2 ** 15), -2 ** 15, 2 ** 15 - 1).astype("<i2").tobytes(), 1
    elif bits == 24:
        v = np.clip(np.rint(frames * 2 ** 23), -2 ** 23, 2 ** 23 - 1).astype(np.int64).reshape(-1) & 0xFFFFFF
        body, tag = np.stack([v & 0xFF, (v >> 8) & 0xFF, (v >> 16) & 0xFF], axis=1).astype(np.uint8).tobytes(), 1
    else:
        body, tag = np.clip(np.rint(frames * 2 ** 31), -2 ** 31, 2 ** 31 - 1).astype("<i4").tobytes(), 1
    bps = 4 if float_fmt else bits // 8
    fmt = struct.pack("<HHIIHH", tag, channels, rate, rate * channels * bps, channels * bps, bps * 8)
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 4 + 8 + len(fmt) + 8 + len(body)) + b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt
                + b"data" + struct.pack("<I", len(body)) + body)


def test_reader_decodes_downmixes_and_resamples(tmp_path):
    """What the reference's `ffmpeg -ar <rate> -ac 1` pipe does for every input (dataset.py:177-203): other sample rates, channel
    counts and sample encodings come back as mono samples at the requested rate on the s16 grid."""
    dur, f0 = 0.5, 440.0
    for i, (rate, bits, ch, flt) in enumerate([(44100, 16, 2, False), (48000, 24, 1, False), (8000, 8, 1, False), (22050, 32, 2, False),
                                               (32000, 32, 1, True), (16000, 24, 3, False)]):
        t = np.arange(int(rate * dur)) / rate
        p = str(tmp_path / f"in{i}.wav")
        _write_wav(p, 0.6 * np.sin(2 * np.pi * f0 * t), rate, bits, ch, flt)
        r = ChunkReader(p, 16000)
        x = r.read(10 ** 6)
        assert r.read(1) is None
        r.close()
        n = int(16000 * dur)
        assert abs(len(x) - n) <= 1, (rate, len(x))
        gain = np.mean([1.0] + [0.5] * (ch - 1))  # the mean of the channels
        want = gain * 0.6 * np.sin(2 * np.pi * f0 * np.arange(len(x)) / 16000)
        tol = 2e-2 if bits == 8 else 2e-3
        assert np.abs(x - want)[200:-200].max() < tol, (rate, bits, ch, np.abs(x - want)[200:-200].max())
        assert np.array_equal(x * 2 ** 15, np.rint(x * 2 ** 15))  # on the s16 grid, as the reference's pipe output
    bad = tmp_path / "clip.mp3"
    bad.write_bytes(b"ID3" + bytes(64))
    import shutil

    trunc = tmp_path / "truncated.wav"  # a RIFF/WAVE header whose fmt chunk stops short: the same friendly error, not a struct.error
    trunc.write_bytes(b"RIFF" + (20).to_bytes(4, "little") + b"WAVE" + b"fmt " + (8).to_bytes(4, "little") + bytes(8))
    if shutil.which("ffmpeg") is None:
        for f in (bad, trunc):
            with pytest.raises(ValueError, match="ffmpeg"):
                ChunkReader(str(f), 16000)


def test_ulaw_codec():
    x = np.array([-1.0, -0.5, -1e-3, 0.0, 1e-3, 0.25, 1.0])
    e = encode_u_law(x)
    assert np.allclose(e, np.sign(x) * np.log1p(255 * np.abs(x)) / np.log(256))  # dataset.py:342-343
    assert np.allclose(decode_u_law(e), x, atol=1e-12)
    assert abs(e[-1] - 1) < 1e-12 and e[3] == 0


def test_schedule_parser():
    assert parse_time_schedule("lambda t: t") is None
    f = parse_time_schedule("lambda t: t**2")
    assert torch.allclose(f(torch.tensor([0.5, 1.0])), torch.tensor([0.25, 1.0]))
    with pytest.raises(ValueError):
        parse_time_schedule("__import__('os').system('true')")


def test_import_shim_paths():
    sys.path.insert(0, ROOT)
    from vq_voice_swap.dataset import ChunkWriter as W  # noqa: F401
    from vq_voice_swap.diffusion_model import DiffusionModel  # noqa: F401
    from vq_voice_swap.models import Classifier, EncoderPredictor
    from vq_voice_swap.vq_vae import VQVAE  # noqa: F401
    assert Classifier(num_labels=3, base_channels=32).num_labels == 3
    ep = EncoderPredictor(32, 256, 512)
    assert ep.save_kwargs() == dict(base_channels=32, downsample_rate=256, num_latents=512, bottleneck_dim=64)
    assert "unet.up_blocks.34.post_cond.1.weight" in ep.state_dict() and tuple(ep.state_dict()["out.weight"].shape) == (512, 64, 1)


@pytest.mark.gpu
def test_sample_scripts_end_to_end(tmp_path):
    sys.path.insert(0, ROOT)
    import sample_diffusion
    import sample_vqvae
    from vq_voice_swap_amd import DiffusionModel, VQVAE
    from vq_voice_swap_amd.det_init import det_init_

    m = DiffusionModel("unet", 32, num_labels=4)
    det_init_(m.state_dict().items())
    ck = str(tmp_path / "d.pt")
    m.save(ck)
    out = str(tmp_path / "many")
    sample_diffusion.main(["--checkpoint-path", ck, "--sample-steps", "3", "--batch-size", "2", "--num-samples", "3", "--constrain",
                           "--sample-path", out, "--schedule", "lambda t: t**2", "--seed", "5", "--target-class", "1", "--precision", "bf16"])
    files = sorted(os.listdir(out))
    assert files == ["sample_000000.wav", "sample_000001.wav", "sample_000002.wav"]
    r = ChunkReader(os.path.join(out, files[0]), 16000)
    a = r.read(64000)
    r.close()
    assert a.shape == (64000,) and np.isfinite(a).all() and np.abs(a).max() <= 1.0 and a.std() > 0.01
    # same seed -> same audio, independent of the batch size (counter-based RNG keyed by clip index)
    out2 = str(tmp_path / "many2")
    sample_diffusion.main(["--checkpoint-path", ck, "--sample-steps", "3", "--batch-size", "3", "--num-samples", "3", "--constrain",
                           "--sample-path", out2, "--schedule", "lambda t: t**2", "--seed", "5", "--target-class", "1", "--precision", "bf16"])
    for f in files:
        assert open(os.path.join(out, f), "rb").read() == open(os.path.join(out2, f), "rb").read()

    v = VQVAE(base_channels=32, pred_name="unet", num_labels=3)
    det_init_(v.state_dict().items())
    with torch.no_grad():
        v.vq.dictionary.mul_(20.0)
    ckv = str(tmp_path / "v.pt")
    v.save(ckv)
    src = os.path.join(out, files[1])
    dst = str(tmp_path / "converted.wav")
    sample_vqvae.main(["--label", "2", "--input-file", src, "--sample-steps", "3", "--check-vq", "--seed", "9", ckv, dst])
    r = ChunkReader(dst, 16000)
    b = r.read(64000)
    r.close()
    assert b.shape == (64000,) and np.isfinite(b).all()
    # unconditional-guidance decode (reference sample_vqvae_uncond.py) and the MFCC-encoder VQ-VAE through the same scripts
    import sample_vqvae_uncond

    dst2 = str(tmp_path / "guided.wav")
    sample_vqvae_uncond.main(["--label", "1", "--input-file", src, "--sample-steps", "3", "--guide-vq-scale", "1.5", "--guide-label-scale",
                              "0.7", "--schedule", "lambda t: t**2", "--seed", "9", "--precision", "fp16", ckv, dst2])
    r = ChunkReader(dst2, 16000)
    c = r.read(64000)
    r.close()
    assert c.shape == (64000,) and np.isfinite(c).all() and np.abs(c - b).max() > 1e-3
    vm = VQVAE(base_channels=32, enc_name="conv-mfcc-ulaw", pred_name="unet", num_labels=3)
    det_init_((k, t) for k, t in vm.state_dict().items() if ".mfcc." not in k)
    ckm = str(tmp_path / "vm.pt")
    vm.save(ckm)
    assert VQVAE.load(ckm).enc_name == "conv-mfcc-ulaw"
    dst3 = str(tmp_path / "converted_mfcc.wav")
    sample_vqvae.main(["--label", "0", "--input-file", src, "--sample-steps", "3", "--encoding", "ulaw", "--seed", "9", "--precision", "fp16", ckm, dst3])
    r = ChunkReader(dst3, 16000, encoding="ulaw")
    d = r.read(64000)
    r.close()
    assert d.shape == (64000,) and np.isfinite(d).all()


@pytest.mark.gpu
def test_cpp_host_samples_like_python(tmp_path):
    """examples/sample_unet.cpp drives the sampler through include/vqvs.h alone (weights from tools/export_weights.py);
    same seed -> the same clips as Diffusion.ddpm_sample (schedule values come from expf vs torch.exp: a few ulp)."""
    import subprocess

    import numpy as np
    import torch

    from vq_voice_swap_amd import DiffusionModel, _native, randn_clips
    from vq_voice_swap_amd.det_init import det_init_

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import export_weights

    model = DiffusionModel("unet", 32)
    det_init_(model.state_dict().items())
    model.eval()
    wfile, ofile, exe = str(tmp_path / "w.bin"), str(tmp_path / "out.f32"), str(tmp_path / "sample_unet")
    export_weights.export(model, wfile)
    libdir = os.path.dirname(_native.LIB_PATH)
    subprocess.run(["g++", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include",
                    os.path.join(ROOT, "examples", "sample_unet.cpp"), "-L" + libdir, "-lvqvs_hip", "-L/opt/rocm/lib", "-lamdhip64",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
    B, T, steps, seed = 2, 4096, 4, 77
    r = subprocess.run([exe, wfile, ofile, str(B), str(T), str(steps), str(seed), "0"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = torch.from_numpy(np.fromfile(ofile, dtype=np.float32).reshape(B, 1, T))
    dev = torch.device("cuda:0")
    model.to(dev)
    x_T = randn_clips(B, T, dev, seed)
    want = model.diffusion.ddpm_sample(x_T, model.predictor, steps, constrain=True, seed=seed).cpu()
    assert (got - want).abs().max().item() < 1e-4
    assert os.path.getsize(ofile + ".wav") == 44 + 2 * T


@pytest.mark.gpu
def test_bench_distributed_branch_two_ranks_on_one_gpu():
    """bench.py's own multi-rank path end to end on the GPU box: `--gpus 2` starts two ranks (here over gloo, both on the one
    GPU), each samples its shard of the clips, rank 0 gathers, the timing is the maximum over ranks, and the line says
    n_gpus = 2.  (With RCCL the same code runs one rank per GPU; two ranks cannot share a device there.)"""
    import json
    import subprocess
    import sys

    env = dict(os.environ, VQVS_BENCH_BACKEND="gloo", OMP_NUM_THREADS="4")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--model", "unet32", "--batch", "3", "--sample-steps", "2",
                        "--steps", "2", "--warmup", "1", "--T", "4096", "--no-cpu-baseline", "--no-other-modes"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 6 and out["steps"] == 2 and out["value"] > 0
    assert out["dtype"] == "fp16" and out["scaling"] == "weak" and out["roofline"]["frac"] > 0
