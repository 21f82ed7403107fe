#!/usr/bin/env python3
"""Experiment: the bench batch (64 x 430 frames, 60 iterations) with one or two batches in flight
(one plan and one stream each).  Does a second stream hide the launch boundaries of the first?"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import MEAN, STD, synthetic_mel
from tts_amd.vocoder import GriffinLimPlan

def main():
    dev = torch.device("cuda:0")
    B, T = 64, 430
    mel = torch.from_numpy(np.stack([synthetic_mel(1234 + i, T) for i in range(B)])).to(dev)
    for n_streams in (1, 2, 3):
        plans = [GriffinLimPlan(1024, 256, 1024, 80, T, B, dev) for _ in range(n_streams)]
        for p in plans:
            p.set_mel_basis(22050, 0.0, 8000)
        streams = [torch.cuda.Stream(dev) for _ in range(n_streams)]
        outs = [torch.empty((B, 256 * (T - 1)), dtype=torch.float32, device=dev) for _ in range(n_streams)]
        def step(i):
            k = i % n_streams
            with torch.cuda.stream(streams[k]):
                plans[k].mel_to_audio(mel, MEAN, STD, [T] * B, seed=1000 + i, n_iter=60, out=outs[k])
        for i in range(30):
            step(i)
        torch.cuda.synchronize()
        steps = 60
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        audio = B * 256 * (T - 1) / 22050
        print(f"{n_streams} batches in flight: {1e3 * dt / steps:.3f} ms per batch, {audio * steps / dt:,.0f} audio-s/s")
        for p in plans:
            p.close()

if __name__ == "__main__":
    main()
