#!/usr/bin/env python3
"""Experiment: the bench batch (64 x 430 frames, 60 iterations) as ONE call, or cut into 2 / 4
sub-batches that run at the same time on streams of their own, each with a run table sized to its
share of the wave slots, so that the launches of the sub-batches are resident together and the
memory burst at the start of one overlaps the arithmetic of the others."""
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
    for parts, target, delay in ((1, 0, 0), (2, 1536, 0), (2, 1536, 20000), (2, 1536, 40000), (2, 1536, 60000),
                                 (2, 1536, 80000), (3, 1024, 30000), (3, 1024, 60000), (1, 0, 0)):
        b = B // parts if B % parts == 0 else (B + parts - 1) // parts
        sizes = [min(b, B - i * b) for i in range(parts)]
        plans = [GriffinLimPlan(1024, 256, 1024, 80, T, s, dev) for s in sizes]
        for p in plans:
            p.set_mel_basis(22050, 0.0, 8000)
            if target:
                p.set_option("run_target", target)
        streams = [torch.cuda.Stream(dev) for _ in range(parts)]
        outs = [torch.empty((s, 256 * (T - 1)), dtype=torch.float32, device=dev) for s in sizes]
        mels = [mel[i * b:i * b + s].contiguous() for i, s in enumerate(sizes)]
        def step(i):
            for k in range(parts):
                with torch.cuda.stream(streams[k]):
                    if delay and k:
                        torch.cuda._sleep(delay * k)   # sets the sub-batches' launches out of step
                    plans[k].mel_to_audio(mels[k], MEAN, STD, [T] * sizes[k], seed=1000 + i, n_iter=60, out=outs[k])
        for i in range(20):
            step(i)
        torch.cuda.synchronize()
        steps = 40
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        audio = B * 256 * (T - 1) / 22050
        print(f"{parts} sub-batches (run_target {target}, offset {delay} cycles): {1e3 * dt / steps:.3f} ms per 64 utterances, "
              f"{audio * steps / dt:,.0f} audio-s/s", flush=True)
        for p in plans:
            p.close()

if __name__ == "__main__":
    main()
