#!/usr/bin/env python3
"""Measurement build (-DGLV_WAVE_TIMES, loaded through GLVOC_LIB): where the waves of one
iteration launch spend their time.  Prints, for the last launch of a bench-shaped call, the
spread of wave starts, the table staging time, the preload time and the loop time.
  GLVOC_LIB=tts_amd/csrc/variants/libglvoc_times.so python scripts/wave_times.py [batch frames n_fft]"""
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
    for i in range(4):
        plan.mel_to_audio(mel, MEAN, STD, [frames] * batch, phase0=None, seed=1000 + i, n_iter=60, momentum=0.99, out=out)
    torch.cuda.synchronize()
    print("ms per launch", plan.last_loop_ms() / 60 if hasattr(plan, "last_loop_ms") else "n/a")
    lib = _lib.load()
    lib.glv_debug_wave_times.restype = ctypes.c_int
    lib.glv_debug_wave_times.argtypes = [ctypes.c_void_p, ctypes.c_int]
    n = 16384
    raw = np.zeros((n, 20), np.uint64)  # 18 stamps + 2 shader-clock readings per wave
    assert lib.glv_debug_wave_times(raw.ctypes.data, n) == 0
    raw = raw[raw[:, 3] > 0].astype(np.int64)
    shader = (raw[:, 19] - raw[:, 18]) / np.maximum(1, raw[:, 3] - raw[:, 0]) * 100.0   # cycles per microsecond
    print(f"shader clock while the waves ran: median {np.median(shader):.0f} MHz  p10 {np.percentile(shader, 10):.0f}  "
          f"p90 {np.percentile(shader, 90):.0f}")
    raw = raw[:, :18]
    t = (raw - raw[:, 0].min()) / 100.0          # microseconds since the first wave started
    t[raw == 0] = np.nan
    print(f"waves {len(t)}")
    for name, col in (("start", t[:, 0]), ("tables staged", t[:, 1]), ("preloaded", t[:, 2]), ("end", t[:, 3])):
        print(f"{name:14s} min {col.min():7.2f}  p10 {np.percentile(col, 10):7.2f}  median {np.median(col):7.2f}  "
              f"p90 {np.percentile(col, 90):7.2f}  max {col.max():7.2f} us")
    for name, d in (("staging", t[:, 1] - t[:, 0]), ("preload", t[:, 2] - t[:, 1]), ("frame loop", t[:, 3] - t[:, 2]),
                    ("whole wave", t[:, 3] - t[:, 0])):
        print(f"{name:14s} mean {d.mean():7.2f}  min {d.min():7.2f}  max {d.max():7.2f} us")
    # waves in launch order: when did each group of 256 start and end
    for i in range(0, len(t), max(1, len(t) // 12)):
        blk = t[i:i + max(1, len(t) // 12)]
        print(f"waves {i:5d}+: start {blk[:, 0].mean():6.2f}  end {blk[:, 3].mean():6.2f}")
    frame_table(t)


def frame_table(t):
    """Mean start time of every frame of the run, for each third of the launch order."""
    third = len(t) // 3
    for k in range(3):
        blk = t[k * third:(k + 1) * third, 4:]
        starts = np.nanmean(blk, axis=0)
        print(f"tier {k}: frame starts " + " ".join(f"{x:6.2f}" for x in starts if not np.isnan(x)))


if __name__ == "__main__":
    main()

