"""Wall-clock latency of the drop-in call on one utterance (host numpy in, host numpy out),
the way the reference's scripts call it.  Runs on the GPU box."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import MEAN, STD, synthetic_mel  # noqa: E402
from tts_amd import mel_to_audio_griffin_lim  # noqa: E402

for n_fft, frames in ((2048, 865), (1024, 430)):
    mel = synthetic_mel(1, frames)
    kw = dict(n_fft=n_fft, win_length=n_fft)
    for _ in range(3):
        mel_to_audio_griffin_lim(mel, MEAN, STD, **kw)
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        y = mel_to_audio_griffin_lim(mel, MEAN, STD, **kw)
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e3
    print(f"n_fft {n_fft}, {frames} frames ({len(y) / 22050:.2f} s audio): median {np.median(ts):.2f} ms, "
          f"min {ts.min():.2f} ms per call")
