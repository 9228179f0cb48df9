"""
Whole-job sampling: shard independent clips across ranks (one process per GPU), sample each
shard with the HIP path, gather the finished waveforms on rank 0 (one all_gather of equal-sized shards).

The path shards embarrassingly: GroupNorm, the `constrain` mean and VQ are all
per clip, so there is NO collective on the data path; the only communication is the final
gather of [n_local,1,T] float32 shards (RCCL over xGMI when the backend is "nccl").  Noise and
x_T are keyed by the GLOBAL clip index, so every clip is bit-identical whatever the GPU count.
"""

from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) of rank's clips; the first n_total % world ranks get one extra."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(n_total, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def _dist_info():
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


# which branch the last gather_clips call took: "local" (no process group), "device_all_gather" (RCCL: device tensors) or
# "host_all_gather" (gloo test runs) -- bench.py reports it, tests/test_rccl_gpu.py asserts it
LAST_GATHER_PATH = "none"


def gather_clips(local: torch.Tensor, n_total: int, T: int) -> Optional[torch.Tensor]:
    """Gather variable-size shards [n_local,1,T] to rank 0 in global clip order (None elsewhere).  With an initialised process
    group the collective runs at EVERY world size, 1 included: a one-rank job under torchrun exercises the same RCCL calls as an
    8-rank one (16 MB per rank: nothing next to the sampling time)."""
    import torch.distributed as dist

    global LAST_GATHER_PATH
    rank, world = _dist_info()
    if not (dist.is_available() and dist.is_initialized()):
        LAST_GATHER_PATH = "local"
        return local
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    n_max = max(e - b for b, e in sizes)
    pad = local
    if local.shape[0] < n_max:
        pad = torch.cat([local, local.new_zeros(n_max - local.shape[0], 1, T)], dim=0)
    pad = pad.contiguous()
    # all_gather is the one collective every backend (RCCL, gloo) implements for equal-sized device tensors; the
    # payload (16 MB per rank at 64 clips) is negligible next to the sampling time, so rank 0 simply keeps its copy
    if pad.is_cuda and dist.get_backend() == "gloo":  # gloo gathers host tensors only (test runs; production is RCCL)
        LAST_GATHER_PATH = "host_all_gather"
        host = pad.cpu()
        bufs = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(bufs, host)
        if rank != 0:
            return None
        return torch.cat([bufs[r][: e - b] for r, (b, e) in enumerate(sizes)], dim=0).to(pad.device)
    LAST_GATHER_PATH = "device_all_gather" if pad.is_cuda else "host_all_gather"
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    if rank != 0:
        return None
    return torch.cat([bufs[r][: e - b] for r, (b, e) in enumerate(sizes)], dim=0)


def sample_clips(
    model,
    n_total: int,
    T: int,
    steps: int,
    seed: int,
    *,
    device=None,
    constrain: bool = False,
    sigma_large: bool = False,
    schedule: Optional[Callable] = None,
    labels: Optional[torch.Tensor] = None,
    gather: bool = True,
    sample_fn: Optional[Callable[[int, int, int], torch.Tensor]] = None,
) -> Optional[torch.Tensor]:
    """Sample `n_total` clips over all ranks.  `labels` (global, [n_total]) selects per-clip classes for a
    class-conditional model (the counterpart of sample_diffusion.py:108-122).  `sample_fn(begin, end, seed)`
    replaces the HIP sampler (used by the CPU gloo tests of the sharding logic)."""
    rank, world = _dist_info()
    begin, end = shard_range(n_total, rank, world)
    n_local = end - begin
    if sample_fn is not None:
        local = sample_fn(begin, end, seed)
    else:
        from .diffusion import randn_clips

        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if n_local == 0:
            local = torch.empty(0, 1, T, device=device)
        else:
            x_T = randn_clips(n_local, T, device, seed, clip_offset=begin)
            pred = model.predictor
            if labels is not None:
                lab = labels[begin:end].to(device)
                pred = lambda xs, ts, _p=model.predictor, _l=lab: _p(xs, ts, labels=_l)  # noqa: E731
            local = model.diffusion.ddpm_sample(x_T, pred, steps, constrain=constrain, sigma_large=sigma_large,
                                                schedule=schedule, seed=seed, clip_offset=begin)
    if not gather:
        return local
    return gather_clips(local, n_total, T)
