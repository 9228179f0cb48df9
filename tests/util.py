import torch


def seeded(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


def rel_rms(a, b):
    return ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-12)).item()


def rms(a):
    return a.pow(2).mean().sqrt().item()


def gate(name, got, want, bound, relative=False):
    """Waveform gate: assert RMS(got - want) < bound (relative to RMS(want) when `relative`; bound None = record the value without
    gating it, only a gross-error check at 1e-2 stays), print the measured value and
    append it to gpurun_out/parity_margins.jsonl (the margins are recorded under profiles/ from a GPU run).  For bounded
    (constrained) waveforms the RMS over the samples the reference did NOT clamp to +-1 is recorded too: clamped samples hide
    error."""
    import json
    import os

    d = got - want
    val = rel_rms(got, want) if relative else rms(d)
    rec = {"test": name, "rms": val, "bound": bound, "relative": bool(relative)}
    if not relative:
        unsat = want.abs() < 1.0
        rec["saturated_fraction"] = 1.0 - unsat.float().mean().item()
        rec["rms_unsaturated"] = d[unsat].pow(2).mean().sqrt().item() if unsat.any() else 0.0
    rec["gated"] = bound is not None
    if bound is None:
        bound = rec["bound"] = 1e-2
    print(f"[margin] {name}: {'rel ' if relative else ''}rms {val:.3e} ({'bound' if rec['gated'] else 'NOT GATED, gross-error bound'} {bound:.1e})"
          + ("" if relative else f", unsaturated-only {rec['rms_unsaturated']:.3e}, saturated {rec['saturated_fraction']:.2f}"))
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "parity_margins.jsonl"), "a") as f:
            f.write(json.dumps(rec) + "\n")
    except OSError:
        pass
    assert val < bound, (name, val, bound)
    return val
