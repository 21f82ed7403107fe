#!/usr/bin/env python3
"""The iteration-spanning launch against one launch per iteration: same bits, and the time of
both (GPU).  scripts/persist_check.py [batch] [frames] [n_fft] [n_iter]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from tts_amd.vocoder import GriffinLimPlan
from tts_amd import filters

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 430
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
K = int(sys.argv[4]) if len(sys.argv) > 4 else 60
dev = torch.device("cuda:0")
plan = GriffinLimPlan(n_fft=N, hop_length=256, win_length=N, n_mels=80, max_frames=T, max_batch=B, device=dev)
plan.set_inverse_basis(filters.inverse_mel_basis(22050, N, 80, 0.0, 8000.0))
print("capacity (workgroups resident together): all bins", plan.persistent_capacity(0), "zero-skip", plan.persistent_capacity(1), flush=True)
rng = np.random.default_rng(0)
mel = torch.from_numpy((rng.standard_normal((B, T, 80)) * 0.7).astype(np.float32)).to(dev)
lengths = [T] * B
if len(sys.argv) > 5 and sys.argv[5] == "ragged":
    lengths = [int(x) for x in np.clip(rng.normal(0.65 * T, 0.2 * T, B), 96, T)]
plan.set_option("timing", 1)
out = {}
for mode in (0, 1, 0, 1):
    plan.set_option("persistent", mode)
    ts = []
    for rep in range(int(os.environ.get("REPS", "6"))):
        y = plan.mel_to_audio(mel, -61.55587, 20.378538, n_frames=lengths, seed=7, n_iter=K)
        torch.cuda.synchronize()
        ts.append(plan.last_loop_ms())
    plan.status()
    out[mode] = y.cpu().numpy()
    print(f"persistent={mode}: loop {min(ts):.4f} ms = {1e3 * min(ts) / K:.2f} us per iteration (runs {ts})", flush=True)
same = np.array_equal(out[0], out[1])
print("bit-identical:", same, "max abs diff", float(np.abs(out[0] - out[1]).max()), flush=True)
sys.exit(0 if same else 1)
