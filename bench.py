#!/usr/bin/env python3
"""Headline benchmark: audio-seconds synthesised per wall-second by 60-iteration
Griffin-Lim (momentum 0.99, fp32) on batches of 64 synthetic LJSpeech-shaped
80-mel utterances (22.05 kHz, n_fft 1024, hop 256, 430 frames), one batch per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A step is one pass of the whole hot path over one batch already resident in HBM:
mel inversion -> 60 Griffin-Lim iterations -> peak normalisation.  For N > 1 the
driver starts one process per GPU with torch.distributed.run; every rank
processes its own batch (weak scaling, no data-path collective) and the slowest
rank's time counts.  Rank 0 prints one JSON line.

`roofline` prices the dominant kernel (one Griffin-Lim iteration) by its
ALGORITHMIC bytes, 44*F*T + 8*L per utterance and iteration, against the 8 TB/s
HBM peak; the kernel time comes from hipEvents recorded on the launch stream
around the 60 iteration launches.  `cpu_baseline` is the numpy/scipy oracle
(a port of librosa's algorithm; librosa itself is not installable here) run over
a bounded sample of the same utterances on the host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SR = 22050
N_FFT = 1024
HOP = 256
N_MELS = 80
FMAX = 8000
N_ITER = 60
MOMENTUM = 0.99
BATCH = 64
FRAMES = 430
HBM_PEAK_GBS = 8000.0


def make_batch(rank, frames_list):
    from helpers import synthetic_mel
    t_max = max(frames_list)
    mel = np.zeros((len(frames_list), t_max, N_MELS), np.float32)
    for i, t in enumerate(frames_list):
        mel[i, :t] = synthetic_mel(1234 + 100000 * rank + i, t)
    return mel


def _cpu_one(args):
    os.environ["OMP_NUM_THREADS"] = "1"
    from oracle import griffinlim_oracle as O
    mel, mean, std = args
    f = N_FFT // 2 + 1
    phase = np.exp(2j * np.pi * np.random.default_rng(7).random((f, mel.shape[0])))
    y = O.mel_to_audio_griffin_lim(mel, mean, std, sr=SR, n_fft=N_FFT, hop_length=HOP,
                                   win_length=N_FFT, n_mels=N_MELS, fmax=FMAX, n_iter=N_ITER,
                                   momentum=MOMENTUM, init_phase=phase)
    return len(y)


def cpu_baseline(mel, mean, std, n_utts):
    """Oracle over the first n_utts utterances of the batch, one process per core."""
    import multiprocessing as mp
    cores = max(1, min(int(os.environ.get("OMP_NUM_THREADS", "16") or 16), os.cpu_count() or 1, 16))
    jobs = [(mel[i], mean, std) for i in range(n_utts)]
    ctx = mp.get_context("spawn")
    with ctx.Pool(cores) as pool:
        pool.map(_cpu_one, jobs[:cores])  # warm-up: imports, FFT plans
        t0 = time.perf_counter()
        samples = pool.map(_cpu_one, jobs)
        dt = time.perf_counter() - t0
    return {"value": round(sum(samples) / SR / dt, 3), "unit": "audio-sec/sec", "cores": cores,
            "kind": "port",
            "sample": f"{n_utts} of the {mel.shape[0]} utterances ({mel.shape[1]} frames, {N_ITER} iterations), "
                      f"numpy/scipy oracle, one process per core, {dt:.1f} s wall"}


def main():
    global N_FFT
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # a step is 3 ms: after an idle period the GPU needs some tens of milliseconds of work to reach
    # its steady clocks (the first 20 steps after idling measure 2.5 % slower than the next 20)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=25)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--frames", type=int, default=FRAMES)
    ap.add_argument("--lengths", choices=["fixed", "ljspeech"], default="fixed")
    ap.add_argument("--n-fft", type=int, default=N_FFT, help="n_fft = win_length (default workload: 1024)")
    ap.add_argument("--frames-per-run", type=int, default=0)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--recompute", type=int, default=-1,
                    help="1/0: force the iteration variant that recomputes the previous spectrum (default: library choice)")
    ap.add_argument("--alias-utts", action="store_true",
                    help="measurement aid: all utterances share one utterance's memory (cache-resident loop)")
    ap.add_argument("--traffic-only", action="store_true",
                    help="measurement aid: iteration kernels move their data but compute nothing")
    ap.add_argument("--linear", type=int, default=-1,
                    help="1/0: momentum term formed in the time domain / in the frequency domain "
                         "(default -1: library default)")
    ap.add_argument("--single-frame", type=int, default=-1,
                    help="1/0: force / forbid the one-frame-per-wave iteration (default -1: library choice)")
    ap.add_argument("--unfused", action="store_true",
                    help="mel inversion, Griffin-Lim and peak normalisation as three library calls")
    ap.add_argument("--cpu-utts", type=int, default=32)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    N_FFT = args.n_fft

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nnodes=1 "
                             "--nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N ...")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # GLV_BENCH_FORCE_DIST=1 takes the multi-rank code path (RCCL init, barrier, all-reduce) with
    # any world size, to rehearse it on a one-GPU box under torchrun --nproc-per-node 1
    use_dist = world > 1 or bool(os.environ.get("GLV_BENCH_FORCE_DIST"))
    if use_dist:
        dist.init_process_group("nccl", device_id=device)

    from helpers import MEAN, STD
    from tts_amd.vocoder import GriffinLimPlan

    if args.lengths == "fixed":
        frames_list = [args.frames] * args.batch
    else:
        rng = np.random.default_rng(4321 + rank)
        frames_list = [int(x) for x in np.clip(rng.normal(560, 190, args.batch), 96, 865)]
    mel_host = make_batch(rank, frames_list)
    t_max = mel_host.shape[1]
    plan = GriffinLimPlan(N_FFT, HOP, N_FFT, N_MELS, t_max, args.batch, device)
    plan.set_mel_basis(SR, 0.0, FMAX)
    plan.set_option("timing", 1)
    if args.frames_per_run:
        plan.set_option("frames_per_run", args.frames_per_run)
    if args.no_graph:
        plan.set_option("use_graph", 0)
    if args.traffic_only:
        plan.set_option("traffic_only", 1)
    if args.alias_utts:
        plan.set_option("alias_utts", 1)
    if args.linear >= 0:
        plan.set_option("linear", args.linear)
    if args.single_frame >= 0:
        plan.set_option("single_frame", args.single_frame)
    if args.recompute >= 0:
        plan.set_option("recompute", args.recompute)
    mel = torch.from_numpy(mel_host).to(device)
    n_samples = [HOP * (t - 1) for t in frames_list]
    out = torch.empty((args.batch, HOP * (t_max - 1)), dtype=torch.float32, device=device)

    def step(i):
        if args.unfused:
            # the three stages as separate library calls (magnitudes pass through a (B, T, F) array)
            s = plan.mel_invert(mel, MEAN, STD)
            plan.griffinlim(s, frames_list, phase0=None, seed=1000 + i, n_iter=N_ITER, momentum=MOMENTUM, out=out)
            plan.peak_normalize(out, n_samples)
        else:
            plan.mel_to_audio(mel, MEAN, STD, frames_list, phase0=None, seed=1000 + i, n_iter=N_ITER,
                              momentum=MOMENTUM, out=out)

    def fence():
        torch.cuda.synchronize(device)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(device)

    for i in range(args.warmup):
        step(i)
    if use_dist:
        fence()  # the first barriers of a process group carry one-time set-up costs
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    elapsed = time.perf_counter() - t0
    # kernel time of the iteration loop: a few extra, untimed steps read back one by one
    loop_ms = []
    for i in range(5):
        step(i)
        loop_ms.append(plan.last_loop_ms())
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert args.traffic_only or args.alias_utts or bool(torch.isfinite(out).all()), "non-finite audio"

    if rank == 0:
        audio_s = sum(n_samples) / SR
        ms_per_step = 1e3 * elapsed / args.steps
        f_bins = N_FFT // 2 + 1
        alg_bytes_iter = sum(44 * f_bins * t + 8 * HOP * (t - 1) for t in frames_list)
        kernel_ms = float(np.mean(loop_ms)) / N_ITER
        achieved = alg_bytes_iter / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if (os.path.exists(tpath) and args.lengths == "fixed" and args.batch == BATCH
                and args.frames == FRAMES and N_FFT == 1024):
            with open(tpath) as fh:
                traffic = json.load(fh).get("gl_engine_iter_bytes_per_launch")
        result = {
            "metric": "audio-sec/sec (whole node), 60-iter Griffin-Lim 80-mel 22.05kHz; % HBM roofline",
            "value": round(world * audio_s / (elapsed / args.steps), 2),
            "unit": "audio-sec/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"configs[1]: batch {args.batch} x {args.lengths} "
                                   f"{args.frames if args.lengths == 'fixed' else 'N(560,190)'} frames per GPU, "
                                   f"80 mels, n_fft {N_FFT}, hop {HOP}, {N_ITER} iterations, momentum {MOMENTUM}, "
                                   "pinv mel inversion + Griffin-Lim + peak normalisation",
                       "global_batch": world * args.batch, "audio_seconds_per_step": round(world * audio_s, 3),
                       "realtime_factor": round(world * audio_s / (elapsed / args.steps), 1),
                       "parallelism": f"batch-sharded x{world}, no data-path collective"},
            "roofline": {"bound": "hbm", "kernel": f"gl_engine_kernel<n_fft {N_FFT}, hop {HOP}, iteration>",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "alg_bytes_per_launch": alg_bytes_iter,
                         "kernel_ms_per_launch": round(kernel_ms, 5), "traffic": traffic,
                         # measured HBM bytes per launch over the same time: the fused kernel moves
                         # about a third of the algorithmic bytes, hence frac > 1 above
                         "traffic_frac": (round(traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                          if traffic else None)},
        }
        if world == 1 and not args.no_cpu:
            result["cpu_baseline"] = cpu_baseline(mel_host, MEAN, STD, min(args.cpu_utts, args.batch))
        print(json.dumps(result), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
