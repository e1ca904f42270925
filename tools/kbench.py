"""Per-stage timing of the hot path at the bench workload (B = 64, 8 objects, C = 13) -- the iteration loop for kernel work.

    python tools/kbench.py [--steps 21] [--precision bf16x3] [--modes strict,invariant,hoisted] [--check]

Prints, per mode, ms/step and the HIP-event average of every stage (lsdm_profile_*), with the algorithmic TFLOP/s of the
MFMA stages.  --check compares the final sample of the strict and invariant schedules bit for bit and one B = 2 step with
the CPU oracle (the checker; never on the timed path).  Environment A/B switches of the library (LSDM_*) apply.
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Runner  # noqa: E402

torch.set_grad_enabled(False)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=21)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--precision", default="bf16x3")
    ap.add_argument("--modes", default="strict,invariant,hoisted")
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    K, B = a.steps, a.batch
    r = Runner(a.precision, B, 13, K, 1, 0, dev)
    T = r.T
    x0 = r.x.clone()
    finals = {}
    for mode in a.modes.split(","):
        kw = dict(hoisted=mode == "hoisted", invariant=mode == "invariant")
        x = x0.clone()
        r.run(T - 1, 0, K, x=x, **kw)              # warm
        torch.cuda.synchronize()
        x.copy_(x0)
        r.ctx.profile_begin()
        t0 = time.perf_counter()
        r.run(T - 1, 0, K, x=x, **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        prof = r.ctx.profile_end()
        x.copy_(x0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r.run(T - 1, 0, K, x=x, **kw)
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t0
        finals[mode] = x.clone()
        print("== %s / %s: %.3f ms/step (%.0f steps/s); without events %.3f ms/step (%.0f steps/s)"
              % (mode, a.precision, dt / K * 1e3, K / dt, dt2 / K * 1e3, K / dt2))
        tot = 0.0
        for p in sorted(prof, key=lambda q: -q["total_ms"]):
            if not p["launches"]:
                continue
            units = B * 9 if p["name"][:4] in ("ball", "sa_k", "thre", "fp_k", "fps_") else B
            avg = p["total_ms"] / p["launches"]
            tf = 2.0 * p["macs_per_unit"] * units / (avg * 1e-3) / 1e12 if avg > 0 else 0.0
            tot += p["total_ms"]
            print("   %-30s %3d x %8.1f us = %7.3f ms/step   %7.1f TFLOP/s" % (p["name"], p["launches"], avg * 1e3, p["total_ms"] / K, tf))
        print("   sum of stages %.3f ms/step" % (tot / K))
    if a.check:
        if "strict" in finals and "invariant" in finals:
            print("strict == invariant bit for bit:", bool(torch.equal(finals["strict"], finals["invariant"])))
        from bench import parity_check
        print("parity (one B=2 step vs the CPU oracle):", parity_check(a.precision, dev))


if __name__ == "__main__":
    main()
