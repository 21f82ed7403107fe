#!/usr/bin/env python3
"""Host time to enqueue one bench-shaped step against the time the GPU needs for it, with and
without the hipGraph and the timing events: shows whether the host keeps ahead of the device.
  python scripts/host_overhead.py        (on a GPU box)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from helpers import MEAN, STD, synthetic_mel
from tts_amd.vocoder import GriffinLimPlan
batch, frames, n_fft = 64, 430, 1024
dev = torch.device("cuda:0")
for graph in (1, 0):
    plan = GriffinLimPlan(n_fft, 256, n_fft, 80, frames, batch, dev)
    plan.set_mel_basis(22050, 0.0, 8000)
    plan.set_option("use_graph", graph)
    for timing in (0, 1):
        plan.set_option("timing", timing)
        mel = torch.from_numpy(np.stack([synthetic_mel(1234 + i, frames) for i in range(batch)])).to(dev)
        out = torch.empty((batch, 256 * (frames - 1)), dtype=torch.float32, device=dev)
        fl = [frames] * batch
        for i in range(3):
            plan.mel_to_audio(mel, MEAN, STD, fl, phase0=None, seed=i, n_iter=60, momentum=0.99, out=out)
        torch.cuda.synchronize()
        n = 20
        t0 = time.perf_counter()
        for i in range(n):
            plan.mel_to_audio(mel, MEAN, STD, fl, phase0=None, seed=i, n_iter=60, momentum=0.99, out=out)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"graph {graph} timing {timing}: host enqueue {1e3*(t1-t0)/n:.3f} ms/step, total {1e3*(t2-t0)/n:.3f} ms/step")
