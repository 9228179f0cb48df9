"""The documented A/B switches must leave a WORKING library: with VQVS_WS=0 or VQVS_WS_F32=0 every convolution falls back to
conv_mfma_kernel -- including the fp32 launches for which the schedule builder would otherwise pick the 128-row x 128-channel
geometry only conv_ws_kernel has (that used to be a hard 'tile_rows does not match dilation' error).
The switches are read once per process, so each case runs in its own interpreter."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch
from oracle import ref_cpu
from vq_voice_swap_amd.unet import ResBlockModule
from vq_voice_swap_amd.det_init import det_init_
from util import rel_rms, seeded
dev = torch.device("cuda:0")
torch.set_num_threads(8)
for i, (cin, cout, dil, L, B) in enumerate([(128, 128, 2, 1500, 3), (128, 256, 1, 700, 2), (64, 64, 2, 2000, 2)]):
    m = ResBlockModule(cin, 256, cout if cout != cin else None, 1.0, dil)
    det_init_((f"sw{{i}}." + k, v) for k, v in m.block.state_dict().items())
    x, e = seeded((B, cin, L), 900 + i), seeded((B, 256), 950 + i)
    sd = {{"b." + k: v.detach() for k, v in m.block.state_dict().items()}}
    want = ref_cpu.res_block(x, sd, "b", dict(cin=cin, cout=cout, scale=1.0, dil=dil), e)
    for prec, tol in (("fp32", 2e-4), ("fp16", 4e-3)):
        m.set_precision(prec)
        err = rel_rms(m(x.to(dev), e.to(dev)).cpu(), want)
        assert err < tol, (prec, cin, cout, err)
print("SWITCH_OK")
"""


@pytest.mark.parametrize("env", [{"VQVS_WS": "0"}, {"VQVS_WS_F32": "0"}, {"VQVS_WS": "0", "VQVS_WS_F32": "0"}])
def test_conv_ws_switched_off_falls_back(env, tmp_path):
    script = tmp_path / "sw.py"
    script.write_text(SCRIPT.format(root=ROOT))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=dict(os.environ, **env), timeout=600)
    assert r.returncode == 0 and "SWITCH_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
