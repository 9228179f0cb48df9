#!/usr/bin/env python3
"""
Encode a clip with a VQ-VAE and decode it as another speaker, on MI355X.  Counterpart of the reference's
sample_vqvae.py (same flags and positionals; reference sample_vqvae.py:76-92): read 4 s of 16 kHz audio,
`encode`, `decode(labels, constrain=True)`, clamp, write WAV; `--check-vq` re-encodes the result.
Differences: WAV in/out directly (no ffmpeg); the model is put in eval mode (the reference's train-mode VQ
bookkeeping crashes on current numpy; outputs are identical).  `--enc-pred-path` loads an
EncoderPredictor whose guidance gradient comes from the library's explicit backward schedule (no autograd).
"""
import argparse
import os
import sys

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # kernel arguments in device memory: a process-level HIP switch, set before the runtime starts

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from vq_voice_swap_amd import VQVAE, EncoderPredictor  # noqa: E402
from vq_voice_swap_amd.audio import ChunkReader, ChunkWriter  # noqa: E402


def arg_parser():
    p = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("--sample-rate", type=int, default=16000)
    p.add_argument("--sample-steps", type=int, default=100)
    p.add_argument("--seconds", type=int, default=4)
    p.add_argument("--label", type=int, default=None, required=True)
    p.add_argument("--input-file", type=str, default=None, required=True)
    p.add_argument("--encoding", type=str, default="linear")
    p.add_argument("--enc-pred-path", type=str, default=None)
    p.add_argument("--enc-pred-scale", type=float, default=1.0)
    p.add_argument("--no-vq", action="store_true")
    p.add_argument("--check-vq", action="store_true")
    p.add_argument("--seed", default=None, type=int)
    p.add_argument("--precision", default="fp32", choices=["fp32", "fp16", "bf16"])
    p.add_argument("checkpoint_path", type=str)
    p.add_argument("output_file", type=str)
    return p


def main(argv=None):
    args = arg_parser().parse_args(argv)
    print("loading model from checkpoint...")
    model = VQVAE.load(args.checkpoint_path)
    assert args.label < model.num_labels
    if not torch.cuda.is_available():
        raise SystemExit("no ROCm device visible: the sampler has no CPU path")
    device = torch.device("cuda")
    model.to(device)
    model.eval()
    model.set_precision(args.precision)
    enc_pred = None
    if args.enc_pred_path:  # reference sample_vqvae.py:24-28
        print("loading encoder predictor")
        enc_pred = EncoderPredictor.load(args.enc_pred_path).to(device)
        enc_pred.eval()
        enc_pred.set_precision(args.precision)

    print(f"loading waveform from {args.input_file}...")
    reader = ChunkReader(args.input_file, sample_rate=args.sample_rate, encoding=args.encoding)
    try:
        chunk = reader.read(args.seconds * args.sample_rate)
    finally:
        reader.close()
    rate = model.downsample_rate  # 256 behind a UNet encoder, lcm(256, 320) = 1280 behind the MFCC encoder (4 s = 64000 fits both)
    usable = (len(chunk) // rate) * rate
    in_seq = torch.from_numpy(chunk[None, None, :usable]).to(device)

    print("encoding audio sequence...")
    encoded = model.encoder(in_seq) if args.no_vq else model.encode(in_seq)

    print("decoding audio samples...")
    labels = torch.tensor([args.label]).long().to(device)
    sample = model.decode(encoded, labels, steps=args.sample_steps, progress=True, constrain=True, seed=args.seed,
                          enc_pred=enc_pred, enc_pred_scale=args.enc_pred_scale)

    if args.check_vq:
        assert not args.no_vq
        count = (encoded == model.encode(sample)).float().mean()
        print(f"fraction of consistent VQ codes: {count}")

    print(f"saving result to {args.output_file}...")
    writer = ChunkWriter(args.output_file, sample_rate=args.sample_rate, encoding=args.encoding)
    try:
        writer.write(sample.clamp(-1, 1).cpu().numpy().flatten())
    finally:
        writer.close()


if __name__ == "__main__":
    main()
