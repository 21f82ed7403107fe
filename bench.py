#!/usr/bin/env python3
"""Headline benchmark: audio-seconds synthesised per wall-second by 60-iteration
Griffin-Lim (momentum 0.99, fp32) on synthetic LJSpeech-shaped 80-mel utterances
(22.05 kHz, n_fft 1024, hop 256, 430 frames).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 1|3|4] [--rehearse]

A step is one pass of the whole hot path over one batch already resident in HBM:
mel inversion -> 60 Griffin-Lim iterations -> peak normalisation.

--config 1 (default; BASELINE.json configs[1]): a batch of 64 utterances per GPU; every rank
  processes its own batch (weak scaling, no data-path collective) and the slowest rank's time
  counts.
--config 3 (BASELINE.json configs[2]): 4096 utterances in all, dealt to the N ranks by length
  (tts_amd.distributed.shard_by_length); a step is every rank's shard plus ONE gather of the
  waveforms to rank 0 on the device (RCCL), inside the timed region; strong scaling.
--config 4 (BASELINE.json configs[3]): ONE 10-minute mel (51 680 frames) cut along time into N
  shards (tts_amd.distributed.griffin_lim_time_sharded): every rank inverts and iterates its
  frames and swaps n_fft - hop samples with each neighbour once per iteration; the step includes
  the all-reduce(MAX) of the peak and the gather of the pieces to rank 0; strong scaling.

With --gpus N > 1 and no torch.distributed environment the script starts the N rank processes
itself (python -m torch.distributed.run ... bench.py, as a child, before this process touches the
GPU), forwards their output and exits with their status; started under torch.distributed.run it
is one rank.  --rehearse puts every rank on GPU 0 and moves the collectives over gloo, so that
the multi-rank code paths can be run on a one-GPU machine (the numbers then mean little).
Rank 0 prints one JSON line.  At N = 1 and the default workload the line also carries
`secondary`: the reference's default shape (64 x 865 frames, n_fft 2048) and the reference's
literal call (one 865-frame numpy mel in, numpy waveform out, copies over PCIe included).

`roofline` prices the dominant kernel (one Griffin-Lim iteration, `gl_engine_kernel`) by the
bytes the fused iteration cannot avoid: per frame the target magnitudes (4*P bytes, P = n_fft/2),
one hop of each of the two newest signals (8*H) and one hop written (4*H), plus the R-1 extra
hops at an utterance's end - against the 8 TB/s HBM peak.  The kernel time comes from hipEvents
recorded on the launch stream around the 60 iteration launches.  `alg_unfused_ratio` keeps the
figure of the unfused chain (iSTFT, STFT and update each reading and writing their operands:
44*F*T + 8*L bytes per utterance and iteration) for continuity; it exceeds 1 because the fused
kernel never moves the spectra.  `traffic` is HBM bytes per launch from the rocprofv3 counter
passes recorded in profiles/traffic.json for exactly this shape, variant and engine source;
null when there is no such record.  `cpu_baseline` is the numpy/scipy oracle (a port of
librosa's algorithm; librosa itself is not installable here) on the host cores.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
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
BATCH_CONFIG3 = 4096
FRAMES = 430
FRAMES_CONFIG4 = 51680  # 600 s of audio at hop 256
NNLS_DEFAULT_STEPS = 40  # GriffinLimPlan.mel_invert_nnls default (accelerated steps)
HBM_PEAK_GBS = 8000.0
ENGINE_SOURCES = ("engine.h", "wave_fft.h", "kernels.h", "elementwise.h", "host_tables.h", "glvoc.hip")


def engine_source_digest():
    """Identifies the kernel sources a counter pass was taken with."""
    h = hashlib.sha256()
    for name in ENGINE_SOURCES:
        with open(os.path.join(ROOT, "tts_amd", "csrc", name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def traffic_key(n_fft, hop, batch, frames, lengths, variant):
    return f"n_fft{n_fft}_hop{hop}_b{batch}_t{frames}_{lengths}_{variant}"


def measured_traffic(key):
    """(bytes per launch, source) from profiles/traffic.json, or (None, reason)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return None, "no counter pass recorded"
    with open(path) as fh:
        entry = json.load(fh).get("entries", {}).get(key)
    if entry is None:
        return None, f"no counter pass recorded for {key}"
    if entry.get("engine_source_digest") != engine_source_digest():
        return None, f"stale: the engine sources changed since the counter pass ({entry.get('profile')})"
    return int(entry["bytes_per_launch"]), entry.get("profile")


def make_batch(seed0, frames_list):
    from helpers import synthetic_mel
    t_max = max(frames_list)
    mel = np.zeros((len(frames_list), t_max, N_MELS), np.float32)
    for i, t in enumerate(frames_list):
        mel[i, :t] = synthetic_mel(seed0 + i, t)
    return mel


_CPU = {}


def _cpu_one(args):
    """One utterance through the oracle.  The mel basis and its pseudo-inverse depend only on the
    configuration and are built once per worker, as a caller of the reference would hoist them."""
    from oracle import griffinlim_oracle as O
    mel, mean, std, n_fft = args
    if _CPU.get("n_fft") != n_fft:
        basis = O.mel_filterbank(sr=SR, n_fft=n_fft, n_mels=N_MELS, fmax=FMAX)
        _CPU.update(n_fft=n_fft, pinv=np.linalg.pinv(basis))
    f = n_fft // 2 + 1
    phase = np.exp(2j * np.pi * np.random.default_rng(7).random((f, mel.shape[0])))
    spec = np.dot(_CPU["pinv"], O.db_to_power(((mel * std) + mean).T))
    y = O.peak_normalise(O.griffinlim(spec, n_iter=N_ITER, hop_length=HOP, win_length=n_fft,
                                      momentum=MOMENTUM, init_phase=phase))
    return len(y)


def cpu_baseline(mel, frames_list, mean, std, n_utts):
    """The oracle over the first n_utts utterances of the batch, one single-threaded process per
    core of this job's CPU share, median of three passes."""
    import multiprocessing as mp
    # the GPU box gives a one-GPU job 16 cores (os.cpu_count() shows the whole machine there)
    cores = int(os.environ.get("GLV_CPU_WORKERS", "0")) or max(1, min(os.cpu_count() or 1, 16))
    saved = {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")}
    for k in saved:  # before the workers start: they read these when numpy loads its BLAS
        os.environ[k] = "1"
    try:
        jobs = [(mel[i, :frames_list[i]], mean, std, N_FFT) for i in range(n_utts)]
        ctx = mp.get_context("spawn")
        with ctx.Pool(cores) as pool:
            pool.map(_cpu_one, jobs[:cores], chunksize=1)  # warm-up: imports, FFT plans, the basis
            times = []
            for _ in range(3):
                t0 = time.perf_counter()
                samples = pool.map(_cpu_one, jobs, chunksize=1)
                times.append(time.perf_counter() - t0)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    dt = float(np.median(times))
    rate = sum(samples) / SR / dt
    return {"value": round(rate, 3), "unit": "audio-sec/sec", "cores": cores, "kind": "port",
            "per_core": round(rate / cores, 3),
            "sample": f"{n_utts} of the {mel.shape[0]} utterances ({N_ITER} iterations, n_fft {N_FFT}), numpy/scipy "
                      f"oracle with the inverse basis hoisted, one single-threaded process per core, median of 3 "
                      f"passes ({', '.join(f'{t:.2f}' for t in times)} s)"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # a step is 3 ms: after an idle period the GPU needs some tens of milliseconds of work to reach
    # its steady clocks (the first 20 steps after idling measure 2.5 % slower than the next 20)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", type=int, choices=[1, 3, 4], default=1,
                    help="1: 64 utterances per GPU (weak scaling); 3: 4096 utterances over all GPUs + final gather; "
                         "4: one 10-minute mel sharded along time, halo swap per iteration")
    ap.add_argument("--no-persistent", action="store_true",
                    help="one launch per iteration instead of the iteration-spanning launch")
    ap.add_argument("--rehearse", action="store_true",
                    help="every rank on GPU 0, collectives over gloo: runs the multi-rank paths on a one-GPU machine")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port of a self-started run (0: a free one)")
    ap.add_argument("--transport", choices=["torch", "library"], default="torch",
                    help="config 4: who moves the halos (torch.distributed point-to-point, or RCCL issued by libglvoc)")
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--frames", type=int, default=None)
    ap.add_argument("--lengths", choices=["fixed", "ljspeech"], default="fixed")
    ap.add_argument("--n-fft", type=int, default=N_FFT, help="n_fft = win_length (default workload: 1024)")
    ap.add_argument("--frames-per-run", type=int, default=0)
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="plan option set before the run (glv_plan_set_option), e.g. fuse_draw=0; may repeat")
    ap.add_argument("--run-target", type=int, default=0,
                    help="number of runs the automatic run table aims at (default 0: the device's wave slots)")
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
    ap.add_argument("--inversion", choices=["pinv", "nnls"], default="pinv",
                    help="nnls: librosa's mel_to_stft (non-negative least squares) in place of the pinv product")
    ap.add_argument("--nnls-steps", type=int, default=None,
                    help="solver steps (default: the library's default for its solver)")
    ap.add_argument("--cpu-utts", type=int, default=64)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="leave the `secondary` measurements out")
    args = ap.parse_args(argv)
    if args.frames is None:
        args.frames = FRAMES_CONFIG4 if args.config == 4 else FRAMES
    if args.steps is None:
        args.steps = {1: 50, 3: 5, 4: 10}[args.config]
    if args.warmup is None:
        args.warmup = {1: 25, 3: 2, 4: 3}[args.config]
    return args


def free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_command(args, argv):
    """The child that runs `argv` (this script's own arguments) as args.gpus ranks on this node."""
    port = args.master_port or free_port()
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def needs_self_launch(args, environ):
    """--gpus N > 1 asked of a plain `python bench.py`: no rank environment is in place."""
    return args.gpus > 1 and "RANK" not in environ and "WORLD_SIZE" not in environ


def variant_name(args, fpr):
    """Names the kernel variant a run exercises, for the look-up of its measured traffic: every
    switch that changes which iteration kernel runs, or how, is part of the key."""
    variant = ("traffic_only" if args.traffic_only else "alias" if args.alias_utts else
               "recompute" if args.recompute == 1 else "spectral" if args.linear == 0 else "default")
    if args.unfused:
        variant += "_unfused"
    if args.inversion != "pinv":
        variant += "_" + args.inversion
    if args.no_persistent:
        variant += "_nopersist"
    if args.no_graph:
        variant += "_nograph"
    if args.config == 4:
        variant += "_timeshard"
    if fpr:
        variant += f"_fpr{fpr}"
    if args.run_target:
        variant += f"_target{args.run_target}"
    for item in sorted(args.option):
        variant += "_" + item.replace("=", "")
    if args.single_frame >= 0:
        variant += f"_single{args.single_frame}"
    return variant


def fused_bytes_per_launch(n_fft, frames_list, s_floats):
    """Bytes one launch of the fused iteration kernel cannot avoid: per frame the magnitudes and one
    hop of each of the two newest signals in, one hop out; R - 1 more hops at an utterance's end."""
    r_cover = n_fft // HOP
    return sum(t * (4 * s_floats + 12 * HOP) + (r_cover - 1) * 12 * HOP for t in frames_list)


def magnitudes_per_frame(n_fft, from_mel_basis):
    """Floats of target magnitudes the iteration moves per frame: P, or 3P/4 where the magnitudes
    come from this mel basis, whose pseudo-inverse has all-zero rows in the upper quarter of every
    lane's bins (the kernel neither stores nor fetches those)."""
    from tts_amd.filters import inverse_mel_basis
    p_half = n_fft // 2
    nz_rows = np.nonzero(np.abs(inverse_mel_basis(SR, n_fft, N_MELS, 0.0, FMAX)).sum(axis=1))[0]
    first_skipped = (3 * p_half) // 4 if n_fft == 1024 else 512 + 256
    return (3 * p_half) // 4 if (from_mel_basis and int(nz_rows.max()) < first_skipped) else p_half


def piece_samples(t0, t1, frames_all, r_cover):
    """Samples of the waveform that the time shard of frames [t0, t1) delivers (the law of
    glv_shard_output_range): blocks [t0, t1) shifted by half a window, the last shard up to the end."""
    half = r_cover // 2
    first = max(t0, half) - half
    end = (max(t1, half) - half) if t1 < frames_all else frames_all - 1
    return HOP * max(0, end - first)


def secondary_measurements(device, mean, std):
    """Two more workloads for the record, both at the reference's own defaults
    (vocoder/mel_audio.py:112: n_fft = win_length = 2048, hop 256, 865 frames as test.py pads to):
    a device-filling batch of 64, and the reference's literal call - one numpy mel in, one numpy
    waveform out through the drop-in function, host-device copies included."""
    import torch
    from tts_amd.vocoder import GriffinLimPlan, mel_to_audio_griffin_lim
    n_fft, frames, batch = 2048, 865, 64
    mel_host = make_batch(1234, [frames] * batch)
    plan = GriffinLimPlan(n_fft, HOP, n_fft, N_MELS, frames, batch, device)
    plan.set_mel_basis(SR, 0.0, FMAX)
    plan.set_option("timing", 1)
    mel = torch.from_numpy(mel_host).to(device)
    out = torch.empty((batch, HOP * (frames - 1)), dtype=torch.float32, device=device)
    lens = [frames] * batch

    def step(i):
        plan.mel_to_audio(mel, mean, std, lens, phase0=None, seed=2000 + i, n_iter=N_ITER, momentum=MOMENTUM, out=out)

    for i in range(5):
        step(i)
    torch.cuda.synchronize(device)
    steps = 10
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize(device)
    dt = (time.perf_counter() - t0) / steps
    kernel_ms = plan.last_loop_ms() / N_ITER
    audio_s = batch * HOP * (frames - 1) / SR
    alg = fused_bytes_per_launch(n_fft, lens, magnitudes_per_frame(n_fft, True))
    key = traffic_key(n_fft, HOP, batch, frames, "fixed", "default")
    traffic, traffic_source = measured_traffic(key)
    # (the counter record is per launch: with all iterations in one launch, N_ITER iterations' worth)
    iters_per_launch = max(1, N_ITER // max(1, plan.last_loop_launches()))
    batch_line = {"workload": f"batch {batch} x {frames} frames, n_fft {n_fft}, hop {HOP}, {N_ITER} iterations "
                              f"(the reference's defaults), inputs resident in HBM",
                  "value": round(audio_s / dt, 2), "unit": "audio-sec/sec", "ms_per_step": round(1e3 * dt, 4),
                  "steps": steps, "iterations_per_launch": iters_per_launch,
                  "kernel_ms_per_iteration": round(kernel_ms, 5), "alg_bytes_per_iteration": alg,
                  "roofline_frac": round(alg / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                  "traffic_per_iteration": round(traffic / iters_per_launch) if traffic else None,
                  "traffic_source": traffic_source, "traffic_key": key}
    plan.close()
    one = mel_host[0]
    for _ in range(5):
        y = mel_to_audio_griffin_lim(one, mean, std, device=device)
    calls = 30
    t0 = time.perf_counter()
    for _ in range(calls):
        y = mel_to_audio_griffin_lim(one, mean, std, device=device)   # returns a host array: synchronous
    dt = (time.perf_counter() - t0) / calls
    assert isinstance(y, np.ndarray) and y.shape == (HOP * (frames - 1),) and np.all(np.isfinite(y))
    call_line = {"workload": "the reference's call: mel_to_audio_griffin_lim(mel (865, 80) numpy, mean, std) with all "
                             "defaults -> numpy waveform; upload, 60 iterations, download per call",
                 "value": round(HOP * (frames - 1) / SR / dt, 2), "unit": "audio-sec/sec",
                 "ms_per_call": round(1e3 * dt, 4), "calls": calls, "pcie_included": True}
    return {"reference_default_shape_batch": batch_line, "reference_literal_call": call_line}


def main():
    global N_FFT
    args = parse_args()
    if needs_self_launch(args, os.environ):
        # One process per GPU.  The ranks are children of this process, which has not touched the
        # GPU (and never will): their output - rank 0's JSON line - goes to the inherited stdout.
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.stdout.flush()
        sys.exit(subprocess.run(launch_command(args, sys.argv[1:]), env=env).returncode)
    N_FFT = args.n_fft
    config3, config4 = args.config == 3, args.config == 4

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the rank environment says WORLD_SIZE={world}")
    device = torch.device("cuda", 0 if args.rehearse else local_rank)
    torch.cuda.set_device(device)
    if args.rehearse and world > 1:
        # ranks that share a GPU: no launch that needs the device to itself (read at plan creation)
        os.environ["GLVOC_PERSISTENT"] = "0"
    if args.no_persistent:
        os.environ["GLVOC_PERSISTENT"] = "0"
    # GLV_BENCH_FORCE_DIST=1 takes the multi-rank code path (process group, barrier, all-reduce,
    # gather) with any world size, to rehearse it under torchrun --nproc-per-node 1
    use_dist = world > 1 or bool(os.environ.get("GLV_BENCH_FORCE_DIST"))
    on_device = not args.rehearse          # collectives on device tensors (RCCL) or through the host (gloo)
    if use_dist:
        if args.rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    def comm(t):
        return t if on_device else t.cpu()

    from helpers import MEAN, STD
    from tts_amd.distributed import (SHARDED_FRAMES_PER_RUN, griffin_lim_time_sharded, shard_by_length,
                                     split_frames)
    from tts_amd.vocoder import GriffinLimPlan, get_plan

    r_cover = N_FFT // HOP
    nnls = args.inversion == "nnls"
    nnls_ms, swap_ms = [], None
    if config4:
        # one long utterance, the whole mel resident on every rank's device; rank r keeps frames
        # split_frames(T, world)[r]
        total_batch, frames_all = 1, args.frames
        from helpers import synthetic_mel
        mel_host = np.concatenate([synthetic_mel(300 + i, FRAMES) for i in range(-(-frames_all // FRAMES))])[:frames_all]
        ranges = split_frames(frames_all, world, min_frames=2 * (r_cover - 1))
        t0_r, t1_r = ranges[rank]
        frames_list = [t1_r - t0_r]
        all_frames = [frames_all]
        mel = torch.from_numpy(mel_host).to(device)
        plan = get_plan(N_FFT, HOP, N_FFT, N_MELS, max(t1_r - t0_r, 2), 1, device)
        fpr = args.frames_per_run
        batch, t_max = 1, frames_all
        block = torch.zeros((max(piece_samples(b, e, frames_all, r_cover) for b, e in ranges),),
                            dtype=torch.float32, device=device)
        gather_parts = None
        if use_dist and rank == 0:
            gather_parts = [torch.empty_like(comm(block)) for _ in range(world)]
    else:
        total_batch = args.batch if args.batch is not None else (BATCH_CONFIG3 if config3 else BATCH)
        if args.lengths == "fixed":
            all_frames = [args.frames] * total_batch
        else:
            rng = np.random.default_rng(4321 + (0 if config3 else rank))
            all_frames = [int(x) for x in np.clip(rng.normal(560, 190, total_batch), 96, 865)]
        if config3:
            # one global list of utterances, dealt to the ranks by length; global index = identity
            shards = shard_by_length(all_frames, world)
            mine = shards[rank]
            frames_list = [all_frames[i] for i in mine]
            mel_host = make_batch(1, [1])[:0] if not mine else np.stack(
                [np.pad(m, ((0, max(frames_list) - m.shape[0]), (0, 0))) for m in
                 (make_batch(1234 + i, [all_frames[i]])[0] for i in mine)])
            utt_ids = list(mine)
            fpr = args.frames_per_run or SHARDED_FRAMES_PER_RUN
        else:
            frames_list = all_frames
            mel_host = make_batch(1234 + 100000 * rank, frames_list)
            utt_ids = None
            fpr = args.frames_per_run
        batch = len(frames_list)
        t_max = mel_host.shape[1]
        plan = GriffinLimPlan(N_FFT, HOP, N_FFT, N_MELS, t_max, batch, device)
        plan.set_utterance_ids(utt_ids)
        mel = torch.from_numpy(mel_host).to(device)
        n_samples = [HOP * (t - 1) for t in frames_list]
        y_len = HOP * (t_max - 1)
        out = torch.empty((batch, y_len), dtype=torch.float32, device=device)
        if nnls:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        # config 3: every rank contributes a (rows, samples) block of the common shape to one gather
        gather_parts = None
        if config3 and use_dist:
            rows = max(len(sh) for sh in shards)
            block = torch.zeros((rows, HOP * (max(all_frames) - 1)), dtype=torch.float32, device=device)
            if rank == 0:
                gather_parts = [torch.empty_like(comm(block)) for _ in range(world)]
    plan.set_mel_basis(SR, 0.0, FMAX)
    plan.set_option("timing", 1)
    if fpr:
        plan.set_option("frames_per_run", fpr)
    if args.run_target:
        plan.set_option("run_target", args.run_target)
    for item in args.option:
        name, _, value = item.partition("=")
        plan.set_option(name, int(value))
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
    nnls_kw = {} if args.nnls_steps is None else {"n_steps": args.nnls_steps}
    last = {}

    def step(i):
        if config4:
            if use_dist:
                y, n_begin = griffin_lim_time_sharded(mel, MEAN, STD, sr=SR, n_fft=N_FFT, hop_length=HOP,
                                                      n_mels=N_MELS, fmax=FMAX, n_iter=N_ITER, momentum=MOMENTUM,
                                                      seed=1000 + i, device=device, transport=args.transport)
            else:
                y, n_begin = one_shard(i)
            last["y"], last["n_begin"] = y, n_begin
            if use_dist:
                block[: y.numel()] = y
                dist.gather(comm(block), gather_parts, dst=0)
        elif nnls:
            # librosa.feature.inverse.mel_to_stft in place of the pinv product, then the same loop
            ev[0].record()
            power = torch.pow(10.0, 0.1 * (mel * float(STD) + float(MEAN)))
            s = plan.mel_invert_nnls(power, power=1.0, **nnls_kw)
            ev[1].record()
            plan.griffinlim(s, frames_list, phase0=None, seed=1000 + i, n_iter=N_ITER, momentum=MOMENTUM, out=out)
            plan.peak_normalize(out, n_samples)
        elif args.unfused:
            # the three stages as separate library calls (magnitudes pass through a (B, T, F) array)
            s = plan.mel_invert(mel, MEAN, STD)
            plan.griffinlim(s, frames_list, phase0=None, seed=1000 + i, n_iter=N_ITER, momentum=MOMENTUM, out=out)
            plan.peak_normalize(out, n_samples)
        else:
            plan.mel_to_audio(mel, MEAN, STD, frames_list, phase0=None, seed=1000 + i, n_iter=N_ITER,
                              momentum=MOMENTUM, out=out)
        if config3 and use_dist:
            block[:batch, :y_len] = out
            dist.gather(comm(block), gather_parts, dst=0)

    def one_shard(i):
        """config 4 without a process group: the whole utterance is the one shard (same calls)."""
        s_local = plan.mel_invert(mel.unsqueeze(0), float(MEAN), float(STD))[0]
        plan.shard_begin(s_local, 0, frames_all, seed=1000 + i)
        for _ in range(N_ITER):
            plan.shard_step(MOMENTUM)
        y, n_begin = plan.shard_end()
        row = y.unsqueeze(0)
        plan.peak_scale(row, plan.peak_reduce(row))
        return y, n_begin

    def shard_loop_ms(swap):
        """config 4: device time of the 60 steps of this rank's shard, without the halo swaps (the
        waveform is then wrong, the work per step the same) or with them."""
        s_local = plan.mel_invert(mel[t0_r:t1_r].unsqueeze(0).contiguous(), float(MEAN), float(STD))[0]
        plan.shard_begin(s_local, t0_r, frames_all, seed=1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        from tts_amd.distributed import exchange_halos

        def do_swap():
            if not swap or not use_dist:
                return
            views = plan.shard_halo_views()
            if not on_device:
                torch.cuda.current_stream(device).synchronize()
            exchange_halos(*views)

        do_swap()
        e0.record()
        for _ in range(N_ITER):
            plan.shard_step(MOMENTUM)
            do_swap()
        e1.record()
        plan.shard_end()
        torch.cuda.synchronize(device)
        return e0.elapsed_time(e1)

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
    if config4:
        finite = bool(torch.isfinite(last["y"]).all())
        steps_only = [shard_loop_ms(False) for _ in range(3)]
        with_swap = [shard_loop_ms(True) for _ in range(3)]
        loop_ms = steps_only
        swap_ms = (float(np.median(with_swap)) - float(np.median(steps_only))) / N_ITER
    else:
        for i in range(3 if config3 else 5):
            step(i)
            loop_ms.append(plan.last_loop_ms())
            if nnls:
                torch.cuda.synchronize(device)
                nnls_ms.append(ev[0].elapsed_time(ev[1]))
        finite = bool(torch.isfinite(out).all())
        plan.status()  # raises if a wait inside an iteration-spanning launch ran out
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device if on_device else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    # (GLV_BENCH_NO_CHECK: timing-only experiment builds of the library produce garbage)
    assert (args.traffic_only or args.alias_utts or os.environ.get("GLV_BENCH_NO_CHECK") or finite), "non-finite audio"

    if rank == 0:
        # whole-job audio: configs 3 and 4 count every utterance once, config 1 a batch per rank
        audio_s = (sum(HOP * (t - 1) for t in all_frames) / SR) if (config3 or config4) \
            else world * sum(n_samples) / SR
        ms_per_step = 1e3 * elapsed / args.steps
        p_half, f_bins = N_FFT // 2, N_FFT // 2 + 1
        # bytes one launch of the iteration kernel cannot avoid (this rank's batch or shard), and
        # the bytes of the unfused chain of the same iteration
        s_floats = magnitudes_per_frame(N_FFT, not nnls and not args.unfused and not config4)
        fused_bytes = fused_bytes_per_launch(N_FFT, frames_list, s_floats)
        unfused_bytes = sum(44 * f_bins * t + 8 * HOP * (t - 1) for t in frames_list)
        # the loop is one iteration-spanning launch where the library could make it one (every run
        # resident at once), else one launch per iteration: bytes and time are those of a LAUNCH
        # (a batch too large for one such launch runs as several, one per group of utterances, one
        # after the other: the figures are then those of the groups' launches taken together)
        groups = 1 if config4 else max(1, plan.last_loop_groups())
        launches = 1 if config4 else max(1, plan.last_loop_launches())
        iters_per_launch = 1 if config4 else N_ITER // launches
        iter_ms = float(np.mean(loop_ms)) / N_ITER
        kernel_ms = iter_ms * iters_per_launch
        iter_bytes, iter_unfused = fused_bytes, unfused_bytes
        fused_bytes, unfused_bytes = fused_bytes * iters_per_launch, unfused_bytes * iters_per_launch
        achieved = fused_bytes / (kernel_ms * 1e-3) / 1e9
        key = traffic_key(N_FFT, HOP, batch, args.frames, args.lengths, variant_name(args, fpr))
        traffic, traffic_source = measured_traffic(key)
        lengths_txt = f"{args.lengths} {args.frames if args.lengths == 'fixed' else 'N(560,190)'} frames"
        if config4:
            workload = (f"configs[3]: one {args.frames}-frame mel ({audio_s:.0f} s) sharded along time over {world} "
                        f"GPU(s) ({frames_list[0]} frames on rank 0), {N_FFT - HOP}-sample halo swap with each "
                        f"neighbour per iteration ({args.transport} transport), all-reduce(MAX) of the peak and "
                        f"gather of the pieces to rank 0 per step")
        elif config3:
            workload = (f"configs[2]: {total_batch} utterances x {lengths_txt} dealt to {world} GPU(s) by "
                        f"length ({batch} on rank 0), one gather of the waveforms to rank 0 per step")
        else:
            workload = f"configs[1]: batch {batch} x {lengths_txt} per GPU"
        workload += (f", 80 mels, n_fft {N_FFT}, hop {HOP}, {N_ITER} iterations, momentum {MOMENTUM}, "
                     + ("NNLS mel inversion" if nnls else "pinv mel inversion")
                     + " + Griffin-Lim + peak normalisation")
        if args.rehearse:
            workload += " [REHEARSAL: all ranks on GPU 0, collectives over gloo]"
        result = {
            "metric": "audio-sec/sec (whole node), 60-iter Griffin-Lim 80-mel 22.05kHz; % HBM roofline",
            "value": round(audio_s / (elapsed / args.steps), 2),
            "unit": "audio-sec/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong" if (config3 or config4) else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload,
                       "global_batch": total_batch if (config3 or config4) else world * batch,
                       "audio_seconds_per_step": round(audio_s, 3),
                       "realtime_factor": round(audio_s / (elapsed / args.steps), 1),
                       "parallelism": (f"time-sharded x{world}, point-to-point halo swap per iteration" if config4 else
                                       f"batch-sharded x{world}, final device-side gather" if config3 else
                                       f"batch-sharded x{world}, no data-path collective")},
            "roofline": {"bound": "hbm",
                         "kernel": (f"gl_engine_kernel<n_fft {N_FFT}, hop {HOP}, iteration-spanning> (all {N_ITER} "
                                    f"iterations in one launch)" if iters_per_launch > 1 else
                                    f"gl_engine_kernel<n_fft {N_FFT}, hop {HOP}, iteration>"),
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "bytes_model": f"fused iteration: T*(4*{s_floats} + 12H) + (R-1)*12H per utterance and iteration "
                                        f"({s_floats} magnitudes per frame, P = {p_half}), x {iters_per_launch} "
                                        f"iteration(s) per launch",
                         "alg_bytes_per_launch": fused_bytes,
                         "kernel_ms_per_launch": round(kernel_ms, 5),
                         "iterations_per_launch": iters_per_launch,
                         "utterance_groups_per_step": groups,
                         "alg_bytes_per_iteration": iter_bytes,
                         "kernel_ms_per_iteration": round(iter_ms, 5),
                         "traffic": traffic, "traffic_source": traffic_source,
                         "traffic_key": key, "engine_source_digest": engine_source_digest(),
                         # measured HBM bytes per launch over the same time
                         "traffic_frac": (round(traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                          if traffic else None),
                         "alg_unfused_bytes_per_launch": unfused_bytes,
                         "alg_unfused_ratio": round(unfused_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
        }
        if config4:
            result["halo"] = {"samples_per_swap_and_side": N_FFT - HOP, "swaps_per_step": N_ITER + 1,
                              "loop_ms_steps_only": round(float(np.median(steps_only)), 4),
                              "loop_ms_with_swaps": round(float(np.median(with_swap)), 4),
                              "swap_ms_per_iteration": round(swap_ms, 5), "transport": args.transport}
        if nnls:
            steps_run = args.nnls_steps if args.nnls_steps is not None else NNLS_DEFAULT_STEPS
            # the mel-domain residual ||A x - b|| / ||b|| (median over the frames) this solve leaves,
            # beside what 200 plain projected-gradient steps - the solver's earlier form - leave
            def mel_residual(x, power):
                from tts_amd.filters import mel_filterbank
                basis = torch.from_numpy(mel_filterbank(SR, N_FFT, N_MELS, 0.0, FMAX)).to(device)
                r = x @ basis.T - power
                return float((r.norm(dim=-1) / power.norm(dim=-1)).median())
            power_mel = torch.pow(10.0, 0.1 * (mel * float(STD) + float(MEAN)))
            res_now = mel_residual(plan.mel_invert_nnls(power_mel, power=1.0, **nnls_kw), power_mel)
            plan.set_option("nnls_accel", 0)
            res_plain200 = mel_residual(plan.mel_invert_nnls(power_mel, power=1.0, n_steps=200), power_mel)
            plan.set_option("nnls_accel", 1)
            result["nnls"] = {"ms_per_solve": round(float(np.mean(nnls_ms)), 4), "steps": steps_run,
                              "form": "projected gradient with Nesterov/FISTA extrapolation",
                              "mel_residual_median": round(res_now, 6),
                              "mel_residual_median_200_plain_steps": round(res_plain200, 6),
                              # flops of the two dense products per step; the solver skips the all-zero
                              # blocks of the banded mel basis, so it executes about a quarter of them
                              # (executed MFMA work and busy cycles: profiles/*_nnls_pmc_counters.json)
                              "dense_flop_per_solve": 4 * steps_run * batch * t_max * f_bins * N_MELS,
                              "dense_equivalent_tflops": round(4 * steps_run * batch * t_max * f_bins * N_MELS
                                                               / (float(np.mean(nnls_ms)) * 1e-3) / 1e12, 2)}
        if world == 1 and not args.no_cpu:
            if config4:
                result["cpu_baseline"] = cpu_baseline(mel_host[None, :8 * FRAMES].reshape(8, FRAMES, N_MELS),
                                                      [FRAMES] * 8, MEAN, STD, 8)
                result["cpu_baseline"]["sample"] = "the first 3440 frames of the mel as 8 pieces of 430 frames; " + \
                    result["cpu_baseline"]["sample"]
            else:
                result["cpu_baseline"] = cpu_baseline(mel_host, frames_list, MEAN, STD, min(args.cpu_utts, batch))
        default_workload = (args.config == 1 and world == 1 and not args.rehearse and args.batch is None
                            and args.frames == FRAMES and N_FFT == 1024 and args.lengths == "fixed"
                            and variant_name(args, fpr) == "default")
        if default_workload and not args.no_secondary:
            plan.close()
            result["secondary"] = secondary_measurements(device, MEAN, STD)
        print(json.dumps(result), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
