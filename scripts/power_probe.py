#!/usr/bin/env python3
"""Board power and shader clock (rocm-smi) sampled while the bench-shaped call runs in a loop.
   python scripts/power_probe.py [seconds]"""
import os, subprocess, sys, threading, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import MEAN, STD, synthetic_mel
from tts_amd.vocoder import GriffinLimPlan


def sample(tag):
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower"], capture_output=True, text=True).stdout
    keep = [l.split(":", 1)[1].strip() for l in out.splitlines() if ("Power" in l or "sclk" in l) and "GPU[" in l]
    print(tag, " | ".join(keep), flush=True)


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
    dev = torch.device("cuda:0")
    B, T = 64, 430
    plan = GriffinLimPlan(1024, 256, 1024, 80, T, B, dev)
    plan.set_mel_basis(22050, 0.0, 8000)
    mel = torch.from_numpy(np.stack([synthetic_mel(1234 + i, T) for i in range(B)])).to(dev)
    out = torch.empty((B, 256 * (T - 1)), dtype=torch.float32, device=dev)
    sample("idle:")
    stop = False

    def sampler():
        while not stop:
            time.sleep(1.0)
            sample("load:")
    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < secs:
        for _ in range(20):
            plan.mel_to_audio(mel, MEAN, STD, [T] * B, seed=n, n_iter=60, out=out)
            n += 1
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stop = True
    th.join()
    print(f"{n} calls, {1e3 * dt / n:.3f} ms per call")


if __name__ == "__main__":
    main()
