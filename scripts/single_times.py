#!/usr/bin/env python3
"""Measurement build (-DGLV_WAVE_TIMES through GLVOC_LIB): where a wave of the one-frame-per-wave
kernel (one utterance, one launch per iteration) spends its launch: tables staged, frame loaded and
transformed forward, phase updated, transformed back, stored.
  GLVOC_LIB=tts_amd/csrc/variants/times.so python scripts/single_times.py [frames n_fft]"""
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
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 865
    n_fft = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    device = torch.device("cuda:0")
    plan = GriffinLimPlan(n_fft, 256, n_fft, 80, frames, 1, device)
    plan.set_mel_basis(22050, 0.0, 8000)
    plan.set_option("timing", 1)
    plan.set_option("persistent", 0)
    mel = torch.from_numpy(np.stack([synthetic_mel(1234, frames)])).to(device)
    out = torch.empty((1, 256 * (frames - 1)), dtype=torch.float32, device=device)
    for i in range(10):
        plan.mel_to_audio(mel, MEAN, STD, [frames], phase0=None, seed=1000 + i, n_iter=60, momentum=0.99, out=out)
    torch.cuda.synchronize()
    print("us per iteration", 1e3 * plan.last_loop_ms() / 60, "launches", plan.last_loop_launches())
    lib = _lib.load()
    lib.glv_debug_wave_times.restype = ctypes.c_int
    lib.glv_debug_wave_times.argtypes = [ctypes.c_void_p, ctypes.c_int]
    n = 16384
    raw = np.zeros((n, 20), np.uint64)
    assert lib.glv_debug_wave_times(raw.ctypes.data, n) == 0
    raw = raw[raw[:, 3] > 0].astype(np.int64)
    t = (raw[:, :18] - raw[:, 0:1].min()) / 100.0
    print("waves", len(t))
    for name, i in (("start (spread)", 0), ("tables staged", 1), ("frame in, first passes", 4), ("phase updated", 5),
                    ("before the last pass", 6), ("stored", 3)):
        d = t[:, i]
        print(f"{name:24s} mean {d.mean():6.2f}  p10 {np.percentile(d, 10):6.2f}  median {np.median(d):6.2f}  p90 {np.percentile(d, 90):6.2f} us")
    clk = (raw[:, 19] - raw[:, 18]).astype(np.float64)
    print("shader clocks per wave: median", np.median(clk), " -> GHz", np.median(clk) / (np.median(t[:, 3] - t[:, 0]) * 1e3))


if __name__ == "__main__":
    main()
