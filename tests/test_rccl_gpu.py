"""The RCCL code path on hardware, at the one world size a 1-GPU box offers: one rank under torch.distributed.run with the
default "nccl" backend (= RCCL on ROCm).  What the 8-GPU scaling run executes first -- init_process_group with a device id, the
barrier, the MAX all-reduce of the timing and the device all_gather of the finished clips -- has then run once on a GPU."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _torchrun(script_args, timeout=600):
    sys.path.insert(0, ROOT)
    from bench import free_port

    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="8")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "VQVS_BENCH_BACKEND"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port())] + script_args
    return subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)


def test_rccl_collectives_one_rank():
    """tools/rccl_selfcheck.py: init (device id), all_gather of a [4, 1, 64000] shard, all_reduce MAX, barrier."""
    r = _torchrun([os.path.join(ROOT, "tools", "rccl_selfcheck.py")])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "rccl ok, world 1" in r.stdout


def test_bench_under_torchrun_one_rank_takes_the_rccl_branch():
    """bench.py as the driver launches it for N > 1, at N = 1: the line must say n_gpus 1 and that the finished clips went through
    the device all_gather (sampler.gather_clips), not around it."""
    r = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--model", "unet32", "--batch", "3", "--sample-steps", "2", "--steps", "1",
                   "--warmup", "0", "--T", "4096", "--no-cpu-baseline", "--no-other-modes"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["config"]["global_batch"] == 3
    assert out["distributed"] == {"backend": "nccl", "world_size": 1, "gather_path": "device_all_gather"}, out.get("distributed")
    assert out["value"] > 0
