from vq_voice_swap_amd.diffusion import CosSchedule, Diffusion, ExpSchedule, Schedule, make_schedule  # noqa: F401
