#!/usr/bin/env python3
"""
Sample waveforms from a diffusion model on MI355X.  Counterpart of the reference's sample_diffusion.py
(same flags; reference sample_diffusion.py:125-141), running on the gfx950 library: x_T ~ N(0,1), optional
class labels (uniform or --target-class), `ddpm_sample`, one 16 kHz mono s16 WAV per clip.
Differences: WAV files are written directly (no ffmpeg); `--schedule` accepts "lambda t: t" / "lambda t: t**P"
without eval; `--seed`, `--precision` are new.  Classifier guidance (`--classifier-path`, reference
sample_diffusion.py:30-42) runs on the same library: the classifier forward and the gradient of log p(y | x_t)
are `vqvs_classifier_guidance` (explicit HIP backward schedule, no autograd).
"""
import argparse
import math
import os
import sys

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # kernel arguments in device memory: a process-level HIP switch, set before the runtime starts

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from vq_voice_swap_amd import Classifier, DiffusionModel, randn_clips  # noqa: E402
from vq_voice_swap_amd.audio import ChunkWriter, parse_time_schedule  # noqa: E402


def arg_parser():
    p = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("--checkpoint-path", default="model_diffusion.pt", type=str)
    p.add_argument("--sample-steps", default=100, type=int)
    p.add_argument("--batch-size", default=1, type=int)
    p.add_argument("--constrain", action="store_true")
    p.add_argument("--sample-path", default="sample.wav", type=str)
    p.add_argument("--num-samples", default=None, type=int)
    p.add_argument("--grad-checkpoint", action="store_true")
    p.add_argument("--classifier-path", default=None, type=str)
    p.add_argument("--classifier-scale", default=1.0, type=float)
    p.add_argument("--target-class", default=None, type=int)
    p.add_argument("--schedule", default="lambda t: t", type=str)
    p.add_argument("--encoding", default="linear", type=str)
    p.add_argument("--seed", default=None, type=int)
    p.add_argument("--precision", default="fp32", choices=["fp32", "fp16", "bf16"])
    return p


def sample_labels(args, num_labels, n, device, gen):
    if args.target_class is not None:
        out = torch.tensor([args.target_class] * n)
    else:
        out = torch.randint(low=0, high=num_labels, size=(n,), generator=gen)
    return out.to(dtype=torch.long, device=device)


def sample_batch(args, model, classifier, device, n, seed, clip_offset, schedule, gen):
    x_T = randn_clips(n, 64000, device, seed, clip_offset=clip_offset)
    pred, labels = model.predictor, None
    if model.num_labels is not None:
        labels = sample_labels(args, model.num_labels, n, device, gen)
        pred = lambda xs, ts, _l=labels: model.predictor(xs, ts, labels=_l)  # noqa: E731
    cond_fn = None
    if classifier is not None:  # reference sample_diffusion.py:34-42, 108-114
        if labels is not None:
            cond_fn = classifier.guidance_fn(labels, args.classifier_scale)
        else:
            # unconditional model: the reference draws the guidance labels afresh on EVERY cond_fn call
            # (sample_diffusion.py:34-36) -- constant only with --target-class
            def cond_fn(x, ts):
                return classifier.log_prob_grad(x, ts, sample_labels(args, classifier.num_labels, len(ts), ts.device, gen),
                                                args.classifier_scale)
    return model.diffusion.ddpm_sample(x_T, pred, args.sample_steps, progress=n == 1, constrain=args.constrain,
                                       cond_fn=cond_fn, schedule=schedule, seed=seed, clip_offset=clip_offset)


def write_clip(path, seq, encoding):
    w = ChunkWriter(path, 16000, encoding=encoding)
    w.write(seq.reshape(-1).cpu().numpy())
    w.close()


def main(argv=None):
    args = arg_parser().parse_args(argv)
    schedule = parse_time_schedule(args.schedule)
    model = DiffusionModel.load(args.checkpoint_path)
    if not torch.cuda.is_available():
        raise SystemExit("no ROCm device visible: the sampler has no CPU path")
    device = torch.device("cuda")
    model.to(device)
    model.eval()
    model.set_precision(args.precision)
    classifier = None
    if args.classifier_path:
        classifier = Classifier.load(args.classifier_path).to(device)
        classifier.eval()
        classifier.set_precision(args.precision)
    seed = args.seed if args.seed is not None else int(torch.randint(0, 2 ** 62, (1,)).item())
    gen = torch.Generator().manual_seed(seed % (2 ** 63))
    if args.num_samples is None:
        write_clip(args.sample_path, sample_batch(args, model, classifier, device, 1, seed, 0, schedule, gen)[0], args.encoding)
        return
    os.mkdir(args.sample_path)
    count = 0
    for b in range(int(math.ceil(args.num_samples / args.batch_size))):
        sample = sample_batch(args, model, classifier, device, args.batch_size, seed, b * args.batch_size, schedule, gen)
        for seq in sample:
            if count == args.num_samples:
                break
            write_clip(os.path.join(args.sample_path, f"sample_{count:06}.wav"), seq, args.encoding)
            count += 1


if __name__ == "__main__":
    main()
