"""Write the parameters of a DiffusionModel checkpoint as a flat float32 file in the library's own parameter order
(vqvs_param_info), for hosts that cannot read torch checkpoints (examples/sample_unet.cpp).

    python tools/export_weights.py model_diffusion.pt weights.bin        # or:  --synthetic 32 weights.bin

Layout: magic "VQVSW1\\0\\0", int32 base_channels, int32 n_tensors, then per tensor: int32 name_len, name bytes,
int64 numel, numel float32 values."""
import os
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # kernel arguments in device memory: a process-level HIP switch, before the runtime starts
import os, struct, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vq_voice_swap_amd import DiffusionModel, _native
from vq_voice_swap_amd.det_init import det_init_


def export(model: DiffusionModel, path: str) -> None:
    pred = model.predictor
    cfg = pred._cfg()
    sd = pred.state_dict()
    table = _native.param_table(cfg)
    with open(path, "wb") as f:
        f.write(b"VQVSW1\0\0")
        f.write(struct.pack("<ii", pred.base_channels, len(table)))
        for name, shape in table:
            t = sd[name].detach().to(torch.float32).cpu().contiguous()
            assert tuple(t.shape) == shape, (name, tuple(t.shape), shape)
            nb = name.encode()
            f.write(struct.pack("<i", len(nb)))
            f.write(nb)
            f.write(struct.pack("<q", t.numel()))
            f.write(t.numpy().tobytes())


if __name__ == "__main__":
    if sys.argv[1] == "--synthetic":
        m = DiffusionModel("unet", int(sys.argv[2]))
        det_init_(m.state_dict().items())
        export(m, sys.argv[3])
    else:
        export(DiffusionModel.load(sys.argv[1]), sys.argv[2])
