#!/usr/bin/env python3
"""Where the reference's literal call (one (865, 80) numpy mel in, numpy waveform out, all
defaults) spends its time on the host side: the same call with the waveform left on the device,
with the mel already on the device, and the pieces of host work timed alone.  One JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def best(fn, n=40):
    fn()
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return round(1e3 * ts[len(ts) // 4], 4)


def main():
    from tts_amd import mel_to_audio_griffin_lim
    rng = np.random.default_rng(0)
    mel = rng.standard_normal((865, 80)).astype(np.float32)
    mel_dev = torch.from_numpy(mel).cuda()
    out = {
        "numpy_in_numpy_out_ms": best(lambda: mel_to_audio_griffin_lim(mel, -30.0, 20.0)),
        "numpy_in_tensor_out_ms": best(lambda: mel_to_audio_griffin_lim(mel, -30.0, 20.0, return_tensor=True)),
        "tensor_in_tensor_out_ms": best(lambda: mel_to_audio_griffin_lim(mel_dev, -30.0, 20.0, return_tensor=True)),
        "seed_draw_ms": best(lambda: np.random.SeedSequence().generate_state(1, dtype=np.uint64)),
        "upload_ms": best(lambda: torch.from_numpy(mel).cuda()),
    }
    y = mel_to_audio_griffin_lim(mel_dev, -30.0, 20.0, return_tensor=True)
    out["download_ms"] = best(lambda: y.cpu().numpy())
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
