#!/usr/bin/env python3
"""
Encode a clip with a VQ-VAE and decode it with unconditional ("classifier-free"-style) guidance towards the VQ codes and / or
the speaker label, on MI355X.  Counterpart of the reference's sample_vqvae_uncond.py (same flags and positionals; reference
sample_vqvae_uncond.py:14-92): the model is one fine-tuned by train_vqvae_uncond.py, whose label 0 is the unconditional label
(hence `--label + 1 < num_labels`).  Differences: WAV in / out directly (no ffmpeg); `--schedule` is parsed, not eval()ed;
eval mode; `--seed`, `--precision` are new; any combination of the two guidance scales works (the reference's always-tripled
batch only lines up when both are non-zero, vq_vae.py:188-203).
"""
import argparse
import os
import sys

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # kernel arguments in device memory: a process-level HIP switch, set before the runtime starts

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from vq_voice_swap_amd import VQVAE  # noqa: E402
from vq_voice_swap_amd.audio import ChunkReader, ChunkWriter, parse_time_schedule  # noqa: E402


def arg_parser():
    p = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("--sample-rate", type=int, default=16000)
    p.add_argument("--sample-steps", type=int, default=100)
    p.add_argument("--seconds", type=int, default=4)
    p.add_argument("--label", type=int, default=None, required=True)
    p.add_argument("--input-file", type=str, default=None, required=True)
    p.add_argument("--encoding", type=str, default="linear")
    p.add_argument("--schedule", default="lambda t: t", type=str)
    p.add_argument("--guide-label-scale", type=float, default=1.0)
    p.add_argument("--guide-vq-scale", type=float, default=0.0)
    p.add_argument("--no-vq", action="store_true")
    p.add_argument("--check-vq", action="store_true")
    p.add_argument("--seed", default=None, type=int)
    p.add_argument("--precision", default="fp32", choices=["fp32", "fp16", "bf16"])
    p.add_argument("checkpoint_path", type=str)
    p.add_argument("output_file", type=str)
    return p


def main(argv=None):
    args = arg_parser().parse_args(argv)
    schedule = parse_time_schedule(args.schedule)
    print("loading model from checkpoint...")
    model = VQVAE.load(args.checkpoint_path)
    assert args.label + 1 < model.num_labels
    if not torch.cuda.is_available():
        raise SystemExit("no ROCm device visible: the sampler has no CPU path")
    device = torch.device("cuda")
    model.to(device)
    model.eval()
    model.set_precision(args.precision)

    print(f"loading waveform from {args.input_file}...")
    reader = ChunkReader(args.input_file, sample_rate=args.sample_rate, encoding=args.encoding)
    try:
        chunk = reader.read(args.seconds * args.sample_rate)
    finally:
        reader.close()
    rate = model.downsample_rate
    in_seq = torch.from_numpy(chunk[None, None, : (len(chunk) // rate) * rate]).to(device)

    print("encoding audio sequence...")
    encoded = model.encoder(in_seq) if args.no_vq else model.encode(in_seq)

    print("decoding audio samples...")
    labels = torch.tensor([args.label]).long().to(device)
    sample = model.decode_uncond_guidance(encoded, labels, steps=args.sample_steps, progress=True, constrain=True,
                                          label_scale=args.guide_label_scale, vq_scale=args.guide_vq_scale, schedule=schedule,
                                          seed=args.seed)

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
