#!/usr/bin/env python3
"""Measurement build (-DGLV_WAVE_TIMES through GLVOC_LIB): the timeline of the MIDDLE iteration
of an iteration-spanning launch, per wave: start of the iteration, neighbours waited for,
preload issued, every frame, end.
  GLVOC_LIB=tts_amd/csrc/variants/times.so python scripts/persist_times.py [batch frames n_fft]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from helpers import MEAN, STD, synthetic_mel
    from tts_amd import _lib
    from tts_amd.vocoder import GriffinLimPlan
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 430
    n_fft = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    device = torch.device("cuda:0")
    plan = GriffinLimPlan(n_fft, 256, n_fft, 80, frames, batch, device)
    plan.set_mel_basis(22050, 0.0, 8000)
    plan.set_option("timing", 1)
    mel = torch.from_numpy(np.stack([synthetic_mel(1234 + i, frames) for i in range(batch)])).to(device)
    out = torch.empty((batch, 256 * (frames - 1)), dtype=torch.float32, device=device)
    for i in range(25):
        plan.mel_to_audio(mel, MEAN, STD, [frames] * batch, phase0=None, seed=1000 + i, n_iter=60, momentum=0.99, out=out)
    torch.cuda.synchronize()
    plan.status()
    print("us per iteration", 1e3 * plan.last_loop_ms() / 60)
    lib = _lib.load()
    lib.glv_debug_wave_times.restype = ctypes.c_int
    lib.glv_debug_wave_times.argtypes = [ctypes.c_void_p, ctypes.c_int]
    n = 16384
    raw = np.zeros((n, 20), np.uint64)
    assert lib.glv_debug_wave_times(raw.ctypes.data, n) == 0
    raw = raw[raw[:, 3] > 0].astype(np.int64)[:, :18]
    t = (raw - raw[:, 0].min()) / 100.0
    t[raw == 0] = np.nan
    print(f"waves {len(t)}")
    nfr = np.sum(~np.isnan(t[:, 4:]), axis=1)
    print("frames per run:", dict(zip(*np.unique(nfr, return_counts=True))))
    def stat(name, d):
        d = d[~np.isnan(d)]
        print(f"{name:26s} mean {d.mean():7.2f}  p10 {np.percentile(d, 10):7.2f}  median {np.median(d):7.2f}  "
              f"p90 {np.percentile(d, 90):7.2f}  max {d.max():7.2f} us")
    stat("iteration start (spread)", t[:, 0])
    stat("wait for neighbours", t[:, 1] - t[:, 0])
    stat("preload (to first frame)", t[:, 4] - t[:, 1])
    last = np.array([t[i, 3 + nfr[i]] for i in range(len(t))])
    stat("frames (first to last start)", last - t[:, 4])
    stat("per frame", (last - t[:, 4]) / np.maximum(1, nfr - 1))
    stat("last frame + flush + flag", t[:, 3] - last)
    stat("whole iteration", t[:, 3] - t[:, 0])
    for w in range(4):
        sel = np.arange(len(t)) % 4 == w
        print(f"wave {w} of its workgroup: wait {np.nanmean(t[sel, 1] - t[sel, 0]):6.2f}  preload {np.nanmean(t[sel, 4] - t[sel, 1]):6.2f}  "
              f"end-part {np.nanmean(t[sel, 3] - last[sel]):6.2f}  whole {np.nanmean(t[sel, 3] - t[sel, 0]):6.2f}")


if __name__ == "__main__":
    main()
