#!/usr/bin/env python3
"""
Headline benchmark: audio clips/sec, unet64 50-step DDPM on 4 s @ 16 kHz waveforms (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: `--batch` clips per GPU (default 64, i.e.
BASELINE config 3's 512 clips over 8 GPUs) taken from x_T to x_0 through `--sample-steps` (50) DDPM
iterations of the unet64 predictor, then gathered on rank 0.  x_T is generated on the GPU before the
timed region (inputs resident in HBM); weights are the deterministic synthetic initialiser.

`--precision` defaults to fp16: fp16 activations and weights on the f16 MFMA with fp32 statistics and
accumulation -- the fastest mode that meets the contract (waveforms within 1e-3 RMS of the CPU reference:
tests/test_parity_gpu.py::test_sampler_end_to_end_vs_golden and tests/test_scale_gpu.py hold it to that gate;
VQ codes stay bit-exact because the encoder always runs in the fp32 mode).  bf16 is faster by ~1 % and outside
the gate (2.6e-3); fp32 (3-term split MFMA) is the 5e-6 parity mode.

`--gpus N` with N > 1 and no torchrun environment re-executes itself under `python -m torch.distributed.run
--nproc-per-node N` (one rank per GPU, RCCL); a line is printed only if the process group really has N ranks.

Rank 0 prints ONE JSON line.  Extra objects:
  roofline     - the dominant kernel (fused MFMA conv): algorithmic bytes of all its launches in one
                 forward / their summed duration, measured live with hipEvents on the launch stream
                 (per-launch brackets minus the measured bracket overhead = the launches back to back, what
                 rocprofv3 --kernel-trace --stats reports under profiles/; the bracketed figure is kept beside it).
                 `traffic` / `mfma_busy` come from rocprofv3 --pmc passes and are replayed from profiles/latest_pmc_*
                 only when that summary is stamped with the build id of the library loaded here.
  cpu_baseline - the CPU oracle (oracle/ref_cpu.py, stock torch fp32 on the host cores) on a bounded
                 sample of the same workload, extrapolated to 50 steps.  Reported baseline, not the target.
  other_modes  - the same workload in the other two precision modes (fp32 parity mode over --steps samples, bf16 over --other-steps).
  other_configs- BASELINE.json configs 2, 4 and 5 at their per-GPU share (rank 0, one GPU): one timed batch each after one warm-up
                 batch, clips/s and end-to-end HBM fraction.  Not part of `value`; --no-other-configs skips them.
"""
import argparse

import os
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # kernel arguments in device memory (vq_voice_swap_amd/__init__.py), before the HIP runtime starts
import csv
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="unet64", choices=["unet32", "unet64"])
    ap.add_argument("--batch", type=int, default=64, help="clips per GPU per step")
    ap.add_argument("--sample-steps", type=int, default=50)
    ap.add_argument("--precision", default="fp16", choices=["fp16", "bf16", "fp32"])
    ap.add_argument("--schedule", default="t**2", choices=["t", "t**2"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-modes", action="store_true", help="skip the measurements of the other two precision modes")
    ap.add_argument("--other-steps", type=int, default=5, help="samples timed for the bf16 mode (the fp32 parity mode is timed over --steps, like the headline)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip BASELINE configs 2, 4 and 5 (one timed batch each)")
    ap.add_argument("--init-only", action="store_true", help="stop after process-group initialisation (launch-path self test)")
    ap.add_argument("--T", type=int, default=64000)
    return ap.parse_args()


def cpu_baseline(base: int, T: int, sample_steps: int):
    """Time the oracle on the host: `nb` clips x `ns` DDPM steps, extrapolated linearly to `sample_steps`."""
    from oracle import ref_cpu
    from vq_voice_swap_amd.det_init import det_tensor
    from vq_voice_swap_amd import _native

    # torch's CPU convolutions stop scaling (and then collapse) far below the 256 hardware threads of the GPU box
    # (tools/cpu_baseline_scaling.py, unet64 forward of 4 clips: 16 threads 0.030 clips/s, 32: 0.027, 64: 0.018, 128: 0.009),
    # and 8 processes x 32 threads side by side are slower still in aggregate (0.010 clips/s: the host is memory-bound):
    # use at most 32 threads of ONE process and say so in "cores".
    cores = min(32, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    cfg = _native.Cfg()
    cfg.kind, cfg.base_channels, cfg.in_channels, cfg.out_channels = 0, base, 1, 1
    sd = {"predictor." + n: det_tensor("predictor." + n, s) for n, s in _native.param_table(cfg)}
    nb, ns = (8, 5) if base == 64 else (8, 10)  # unet64, 8 clips x 5 steps, extrapolated linearly (~25 s of CPU work on 32 threads)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(nb, 1, T, generator=g)
    noises = [torch.randn(nb, 1, T, generator=g) for _ in range(ns)]
    pred = lambda a, b: ref_cpu.unet_predictor(sd, base, a, b)  # noqa: E731
    with torch.no_grad():
        pred(x[:1], torch.tensor([0.5]))  # warm the thread pool
        t0 = time.time()
        ref_cpu.ddpm_sample("exp", x, pred, ns, noises, constrain=True)
        dt = time.time() - t0
    clips_per_s = nb / (dt * sample_steps / ns)
    return {"value": clips_per_s, "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": f"oracle/ref_cpu.py (torch fp32 CPU, {cores} threads of {os.cpu_count()} hw threads), unet{base}, {nb} clips x {ns} "
                      f"DDPM steps at T={T} in {dt:.1f}s, extrapolated x{sample_steps}/{ns} steps"}


def other_configs(dev, prec: str, T: int):
    """BASELINE.json configs 2, 4 and 5 on ONE GPU at their per-GPU share of the 8-GPU batch, in mode `prec`: one warm-up batch, one
    timed batch each.  clips/s and the end-to-end HBM fraction = algorithmic (Model A) bytes of every forward / backward pass of the
    batch / wall time / 8 TB/s.  Accuracy of each configuration in this mode: fixtures F6 (config 2's family), F8c (config 4), F13
    (config 5), tests/test_scale_gpu.py."""
    from vq_voice_swap_amd import Classifier, DiffusionModel, VQVAE, randn_clips
    from vq_voice_swap_amd.det_init import det_init_

    def det(m):
        det_init_((k, v) for k, v in m.state_dict().items() if ".mfcc." not in k)
        return m.eval().to(dev)

    def timed(fn, warm):
        warm()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    def entry(workload, clips, dt, by):
        return {"workload": workload, "dtype": prec, "value": round(clips / dt, 2), "unit": "clips/s", "s_per_batch": round(dt, 3),
                "e2e_hbm_frac": round(by / dt / (HBM_PEAK_GBPS * 1e9), 4), "algorithmic_GB_per_batch": round(by / 1e9, 1)}

    out = {}
    sq = lambda t: t ** 2  # noqa: E731
    x = randn_clips(64, T, dev, 1)
    m = det(DiffusionModel("unet", 32))
    m.set_precision(prec)
    dt = timed(lambda: m.diffusion.ddpm_sample(x, m.predictor, 50, constrain=True, schedule=sq, seed=3),
               lambda: m.diffusion.ddpm_sample(x, m.predictor, 10, constrain=True, schedule=sq, seed=2))
    out["config2"] = entry("unet32 50-step DDPM (constrain, schedule t**2), 64 clips of T=%d on 1 GPU" % T, 64, dt,
                           50 * m.predictor.handle(dev, 64, T).model_bytes(64, T))
    m.predictor.invalidate()
    del m

    v = det(VQVAE(base_channels=64, enc_name="unet", pred_name="unet", num_labels=251))
    v.set_precision(prec)
    wav = (0.1 * torch.randn(32, 1, T, device=dev)).clamp(-1, 1)
    labels = torch.arange(32, device=dev) % 251
    dt = timed(lambda: v.decode(v.encode(wav), labels, steps=50, constrain=True), lambda: v.decode(v.encode(wav), labels, steps=10, constrain=True))
    by = 50 * v.predictor._handle.model_bytes(32, T) + v.encoder._handle.model_bytes(32, T)
    out["config4"] = entry("VQ-VAE speaker conversion: unet64 encoder (fp32) + VQ + conditional unet64 decoder, 50 steps, 32 clips/GPU (256 over 8) "
                           "of T=%d" % T, 32, dt, by)
    v.predictor.invalidate()
    v.encoder.invalidate()
    del v

    m = det(DiffusionModel("unet", 64))
    m.set_precision(prec)
    clf = det(Classifier(num_labels=251, base_channels=32))
    clf.set_precision(prec)
    x32 = x[:32].contiguous()
    dt = timed(lambda: m.diffusion.ddpm_sample(x32, m.predictor, 100, constrain=True, cond_fn=clf.guidance_fn(labels, 1.0), seed=3),
               lambda: m.diffusion.ddpm_sample(x32, m.predictor, 10, constrain=True, cond_fn=clf.guidance_fn(labels, 1.0), seed=2))
    by = 100 * (m.predictor.handle(dev, 32, T).model_bytes(32, T) + clf.handle(dev, 32, T).model_bytes(32, T))
    out["config5"] = entry("unet64 classifier-guided sampling (classifier32 forward + backward at every step), 100 steps, 32 clips/GPU (256 over 8) "
                           "of T=%d" % T, 32, dt, by)
    m.predictor.invalidate()
    clf.invalidate()
    del m, clf
    torch.cuda.empty_cache()
    return out


def box_id(dev):
    """Which box measured this line (boxes differ by +- 2 % on this workload: a number is only comparable with numbers of the same box)."""
    import socket

    p = torch.cuda.get_device_properties(dev)
    out = {"hostname": socket.gethostname(), "gpu": p.name, "compute_units": p.multi_processor_count, "hbm_GB": round(p.total_memory / 2 ** 30, 1)}
    uuid = getattr(p, "uuid", None)
    if uuid is not None:
        out["gpu_uuid"] = str(uuid)
    return out


def free_port() -> int:
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def relaunch(n: int) -> None:
    """`python bench.py --gpus N` outside torchrun: become `python -m torch.distributed.run --nproc-per-node N bench.py ...`
    (same arguments, same stdout), one rank per GPU."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    os.environ.setdefault("OMP_NUM_THREADS", "8")
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def main():
    a = parse()
    backend = os.environ.get("VQVS_BENCH_BACKEND", "nccl")  # "gloo": CPU self test of the launch path (--init-only)
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ and "MASTER_PORT" in os.environ
    if not launched and a.gpus > 1:
        if backend == "nccl" and torch.cuda.device_count() < a.gpus:
            sys.exit(f"bench.py: --gpus {a.gpus} but only {torch.cuda.device_count()} GPU(s) are visible")
        relaunch(a.gpus)  # does not return
    world = int(os.environ.get("WORLD_SIZE", "1")) if launched else 1
    rank = int(os.environ.get("RANK", "0")) if launched else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if launched else 0
    use_dist = launched  # under torchrun the RCCL path is exercised at every world size, 1 included
    if use_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        world = dist.get_world_size()
    if world != a.gpus:  # never print a line whose n_gpus is not what was asked for
        if rank == 0:
            print(f"bench.py: --gpus {a.gpus} but the process group has {world} rank(s); refusing to report", file=sys.stderr)
        if use_dist:
            torch.distributed.destroy_process_group()
        sys.exit(2)
    if a.init_only:
        t = torch.tensor([float(rank)], device=torch.device("cuda", local_rank) if backend == "nccl" else "cpu")
        if use_dist:
            torch.distributed.all_reduce(t)
        if rank == 0:
            ok = float(t.item()) == world * (world - 1) / 2
            print(json.dumps({"init_only": True, "world_size": world, "backend": backend if use_dist else None, "allreduce_ok": ok}), flush=True)
        if use_dist:
            torch.distributed.destroy_process_group()
        return
    n_gpus = world
    # (gloo runs are launch-path tests: their ranks may share a GPU; RCCL runs are one rank per GPU)
    dev = torch.device("cuda", local_rank if backend == "nccl" else local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)

    from vq_voice_swap_amd import DiffusionModel, randn_clips
    from vq_voice_swap_amd.det_init import det_init_
    from vq_voice_swap_amd import sampler as _sampler
    from vq_voice_swap_amd.sampler import gather_clips, shard_range

    def gather_path():
        return _sampler.LAST_GATHER_PATH

    base = 64 if a.model == "unet64" else 32
    model = DiffusionModel("unet", base)
    det_init_(model.state_dict().items())
    model.eval()
    model.set_precision(a.precision)
    tmap = (lambda t: t ** 2) if a.schedule == "t**2" else None
    n_total = a.batch * n_gpus
    begin, end = shard_range(n_total, rank, n_gpus)
    seed = 1234

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    def one_step(step_id: int):
        x_T = randn_clips(end - begin, a.T, dev, seed + step_id, clip_offset=begin)
        torch.cuda.synchronize()
        return x_T

    def run(x_T, step_id):
        x0 = model.diffusion.ddpm_sample(x_T, model.predictor, a.sample_steps, constrain=True, schedule=tmap,
                                         seed=seed + step_id, clip_offset=begin)
        return gather_clips(x0, n_total, a.T)

    # one-time initialisation outside any step: pack/upload the weights, size the arena, set kernel attributes
    model.predictor.handle(dev, end - begin, a.T)
    model.predictor(randn_clips(end - begin, a.T, dev, 1), torch.full((end - begin,), 0.5, device=dev))
    barrier()
    for w in range(a.warmup):
        run(one_step(w), w)
    inputs = [one_step(100 + k) for k in range(a.steps)]
    barrier()
    t0 = time.perf_counter()
    out = None
    for k in range(a.steps):
        out = run(inputs[k], 100 + k)
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    if rank == 0:
        assert out is not None and out.shape == (n_total, 1, a.T) and bool(torch.isfinite(out).all())

    # ---- live per-kernel timing of one forward (hipEvents on the launch stream), rank 0 ----
    def kernel_roofline(prec):
        """Roofline object of the convolution launches of one forward in mode `prec`, timed live in this process.

        Two timings, both with hipEvents on the library's launch stream (= torch's current stream):
          * bracketed: the library records an event after EVERY schedule entry (vqvs_set_profiling) -> per-kind times.  Every
            bracket delays the next dispatch by a few microseconds, so their sum exceeds a plain forward;
          * plain: one event pair around whole forwards with profiling off.
        The per-launch bracket overhead is (bracketed sum - plain forward) / launches, the same for every launch; the convolution
        time quoted in `frac` is the bracketed convolution time minus that overhead times the convolution launches -- what the
        launches take when they run back to back, the quantity rocprofv3 --kernel-trace --stats reports (profiles/)."""
        h = model.predictor.handle(dev, end - begin, a.T)
        B = end - begin
        x = inputs[0]
        ts = torch.full((B,), 0.5, device=dev)
        reps = 5
        model.predictor(x, ts)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            model.predictor(x, ts)
        e1.record()
        torch.cuda.synchronize()
        plain_ms = e0.elapsed_time(e1) / reps
        h.set_profiling(True)
        per_kind = {}
        launched = {}
        for _ in range(reps):
            model.predictor(x, ts)
            ms = h.profile_read()
            for (kind, _by, _fl), t in zip(h.op_info(B, a.T), ms):
                d = per_kind.setdefault(kind, dict(ms=0.0, bytes=0, flops=0, launches=0))
                d["ms"] += t / reps
                if t > 5e-4:  # (an entry that launched nothing -- a gn_prepare whose rows the convolution builds -- measures ~0)
                    launched[kind] = launched.get(kind, 0) + 1.0 / reps
        info = h.op_info(B, a.T)
        for kind, by, fl in info:
            per_kind[kind]["bytes"] += by
            per_kind[kind]["flops"] += fl
            per_kind[kind]["launches"] += 1
        h.set_profiling(False)
        bracketed_ms = sum(d["ms"] for d in per_kind.values())
        n_launched = sum(launched.values())
        overhead_us = max(0.0, (bracketed_ms - plain_ms) * 1e3 / max(n_launched, 1.0))
        conv = per_kind["conv"]
        conv_ms = conv["ms"] - overhead_us * 1e-3 * launched.get("conv", conv["launches"])
        # HBM traffic and MFMA-busy fraction of the same launches come from rocprofv3 --pmc passes (tools/measure.sh: every counter
        # set in its own run; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950; SQ_VALU_MFMA_BUSY_CYCLES summed over
        # the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs).  They cannot be collected from inside this process, so the summary under
        # profiles/latest_* is replayed -- but ONLY when its stamp says it was measured on the very library loaded here (the build id
        # in vqvs_version() hashes every kernel source); a summary of other kernels is reported as stale and not replayed.
        from vq_voice_swap_amd import _native

        lib_version = _native.lib().vqvs_version().decode()
        traffic = mfma_busy = None
        src = {}
        convs = ("conv_ws_kernel", "conv_mfma_kernel")
        full = a.model == "unet64" and B == 64 and a.T == 64000
        stamp_path = os.path.join(ROOT, "profiles", f"latest_pmc_stamp_unet64_{prec}.json")
        if full and os.path.exists(stamp_path):
            stamp = json.load(open(stamp_path))
            if stamp.get("library") != lib_version:
                src["stale"] = f"profiles/latest_pmc_*_unet64_{prec}.csv were measured on '{stamp.get('library')}', this is '{lib_version}': not replayed"
            else:
                pmc = os.path.join(ROOT, "profiles", f"latest_pmc_traffic_per_op_unet64_{prec}.csv")
                if os.path.exists(pmc):
                    rows = [r for r in csv.DictReader(open(pmc)) if r["kernel"] in convs]
                    if rows:
                        traffic = (sum(float(r["FETCH_SIZE"]) for r in rows) * 2 + sum(float(r["WRITE_SIZE"]) for r in rows)) * 1024 / len(rows)
                        src["traffic"] = os.path.relpath(pmc, ROOT)
                pmc2 = os.path.join(ROOT, "profiles", f"latest_pmc_per_op_unet64_{prec}.csv")
                if os.path.exists(pmc2):
                    rows = [r for r in csv.DictReader(open(pmc2)) if r["kernel"] in convs]
                    if rows and "SQ_VALU_MFMA_BUSY_CYCLES" in rows[0]:
                        mfma_busy = sum(float(r["SQ_VALU_MFMA_BUSY_CYCLES"]) for r in rows) / (sum(float(r["GRBM_GUI_ACTIVE"]) for r in rows) / 8 * 1024)
                        src["mfma_busy"] = os.path.relpath(pmc2, ROOT)
        ach = conv["bytes"] / (conv_ms * 1e-3) / 1e9
        ach_br = conv["bytes"] / (conv["ms"] * 1e-3) / 1e9
        roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": None if traffic is None else round(traffic),
                "mfma_busy": None if mfma_busy is None else round(mfma_busy, 4),
                "timing": "hipEvents on the launch stream, this process: per-launch brackets minus the measured bracket overhead "
                          "(= launches running back to back, what rocprofv3 --kernel-trace --stats reports)",
                "conv_ms_per_forward": round(conv_ms, 3),
                "conv_ms_per_forward_bracketed": round(conv["ms"], 3),
                "frac_bracketed": round(ach_br / HBM_PEAK_GBPS, 4),
                "bracket_overhead_us_per_launch": round(overhead_us, 2),
                "forward_ms_plain": round(plain_ms, 3),
                "forward_ms_bracketed": round(bracketed_ms, 3),
                "launches_timed_per_forward": round(n_launched, 1),
                # `traffic` and `mfma_busy` are NOT measured by this process (see above); everything else in this object is
                "traffic_measured_live": False,
                "library": lib_version,
                "pmc_sources": src or None,
                "traffic_note": "HBM bytes per launch: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of the same mode, "
                                "FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md); algorithmic bytes per launch = "
                                "algorithmic_bytes_per_forward / launches_per_forward; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / "
                                "(GRBM_GUI_ACTIVE / 8 * 1024 SIMDs) over the same launches",
                "kernel": "fused convolution launches: conv_ws_kernel (wave-specialised, persistent; every mode) + conv_mfma_kernel "
                          "(fp32 mode: avg-pooled launches; shapes conv_ws_kernel declines)",
                "launches_per_forward": conv["launches"],
                "avg_launch_us": round(conv_ms * 1e3 / conv["launches"], 2),
                "algorithmic_bytes_per_forward": conv["bytes"],
                "mfma_tflops": round(conv["flops"] / (conv_ms * 1e-3) / 1e12, 1)}
        return h, per_kind, roof

    roof = None
    extra = {}
    if rank == 0:
        h, per_kind, roof = kernel_roofline(a.precision)
        B = end - begin
        fwd_ms = sum(d["ms"] for d in per_kind.values())
        model_bytes = h.model_bytes(B, a.T)
        extra = {
            "forward_ms_event_sum": round(fwd_ms, 3),
            "kernel_ms_per_forward": {k: round(v["ms"], 3) for k, v in sorted(per_kind.items(), key=lambda kv: -kv[1]["ms"])},
            "model_bytes_per_forward": model_bytes,
            "model_flops_per_forward": h.flops(B, a.T),
            "e2e_hbm_frac": round(model_bytes * a.sample_steps * a.steps * n_gpus / dt / 1e9 / (HBM_PEAK_GBPS * n_gpus), 4),
            "device_mem_GB": round(h.device_bytes() / 2 ** 30, 2),
        }

    cpu = None
    if rank == 0 and not a.no_cpu_baseline and n_gpus == 1:
        cpu = cpu_baseline(base, a.T, a.sample_steps)

    # the same workload in the other two precision modes (--other-steps samples each), so that all three are on record side by side
    others = None
    if rank == 0 and n_gpus == 1 and not a.no_other_modes:
        # (accuracy is NOT measured in this run: the modes are held to the gate by the -m gpu tests, on this very workload by
        #  tests/test_scale_gpu.py::test_headline_workload_vs_reference_fixture, fixture F6b made from the reference)
        notes = {"fp32": "parity mode: fp32 storage, 3-term bf16-split MFMA (gate: tests, fixture F6b)",
                 "fp16": "fp16 storage + f16 MFMA, fp32 accumulation and statistics (gate: tests, fixture F6b)",
                 "bf16": "bf16 storage + bf16 MFMA: OUTSIDE the 1e-3 waveform gate, kept for comparison"}
        others = []
        for prec in ("fp32", "fp16", "bf16"):
            if prec == a.precision:
                continue
            model.set_precision(prec)
            model.predictor.handle(dev, end - begin, a.T)
            x_w = one_step(7)
            model.predictor(x_w, torch.full((end - begin,), 0.5, device=dev))
            torch.cuda.synchronize()
            n_other = a.steps if prec == "fp32" else a.other_steps  # (the parity mode -- the number a strict reader quotes -- over the headline's sample count)
            t1 = time.perf_counter()
            for k in range(n_other):
                model.diffusion.ddpm_sample(x_w, model.predictor, a.sample_steps, constrain=True, schedule=tmap, seed=seed + k, clip_offset=begin)
            torch.cuda.synchronize()
            rate = round((end - begin) * n_other / (time.perf_counter() - t1), 3)
            _h, pk, roof_o = kernel_roofline(prec)
            others.append({"dtype": prec, "value": rate, "unit": "clips/s", "steps": n_other, "note": notes[prec], "roofline": roof_o,
                           "forward_ms_event_sum": round(sum(d["ms"] for d in pk.values()), 3)})
        model.set_precision(a.precision)

    # BASELINE.json's other configurations (2, 4, 5) at their per-GPU share, in the headline's precision mode: one timed batch each
    # after one warm-up batch (tools/bench_configs.py is the builder-run form of the same measurement, profiles/r06_configs_1gpu.json)
    other_cfgs = None
    if rank == 0 and n_gpus == 1 and not a.no_other_configs and a.T == 64000:
        other_cfgs = other_configs(dev, a.precision, a.T)

    if rank == 0:
        clips = n_total * a.steps
        line = {
            "metric": "audio clips/sec (whole node), unet64 50-step DDPM on 4s@16kHz waveforms",
            "value": round(clips / dt, 3),
            "unit": "clips/s",
            "n_gpus": n_gpus,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 2),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": a.precision,
            "data": "synthetic (x_T ~ N(0,1) from the counter-based generator; deterministic synthetic weights)",
            "config": {"workload": f"{a.model} {a.sample_steps}-step DDPM (constrain, schedule {a.schedule}), "
                                   f"{a.batch} clips/GPU x {n_gpus} GPU of T={a.T}",
                       "global_batch": n_total, "parallelism": f"clips sharded over {n_gpus} GPU(s), gather to rank 0"},
            "roofline": roof,
            "distributed": None if not use_dist else {"backend": backend, "world_size": world, "gather_path": gather_path()},
            "cpu_baseline": cpu,
            "other_modes": others,
            "other_configs": other_cfgs,
            "box": box_id(dev),
            "parity": "dtype mode held to <= 1e-3 waveform RMS vs the CPU reference by tests/test_parity_gpu.py and tests/test_scale_gpu.py"
                      if a.precision in ("fp16", "fp32") else "dtype mode is OUTSIDE the 1e-3 waveform gate",
        }
        line.update(extra)
        print(json.dumps(line), flush=True)
    if use_dist:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
