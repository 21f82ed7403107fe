#!/usr/bin/env python3
"""Headline benchmark: audio-seconds synthesised per wall-second by 60-iteration
Griffin-Lim (momentum 0.99, fp32) on synthetic LJSpeech-shaped 80-mel utterances
(22.05 kHz, n_fft 1024, hop 256, 430 frames).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 1|3]

A step is one pass of the whole hot path over one batch already resident in HBM:
mel inversion -> 60 Griffin-Lim iterations -> peak normalisation.

--config 1 (default; BASELINE.json configs[1]): a batch of 64 utterances per GPU.  For N > 1 the
  driver starts one process per GPU with torch.distributed.run; every rank processes its own
  batch (weak scaling, no data-path collective) and the slowest rank's time counts.
--config 3 (BASELINE.json configs[2]): 4096 utterances in all, dealt to the N ranks by length
  (tts_amd.distributed.shard_by_length); a step is every rank's shard plus ONE gather of the
  waveforms to rank 0 on the device (RCCL), inside the timed region; strong scaling.
Rank 0 prints one JSON line.

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


def main():
    global N_FFT
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # a step is 3 ms: after an idle period the GPU needs some tens of milliseconds of work to reach
    # its steady clocks (the first 20 steps after idling measure 2.5 % slower than the next 20)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", type=int, choices=[1, 3], default=1,
                    help="1: 64 utterances per GPU (weak scaling); 3: 4096 utterances over all GPUs + final gather")
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--frames", type=int, default=FRAMES)
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
    ap.add_argument("--nnls-steps", type=int, default=200)
    ap.add_argument("--cpu-utts", type=int, default=64)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    N_FFT = args.n_fft
    config3 = args.config == 3
    if args.steps is None:
        args.steps = 5 if config3 else 50
    if args.warmup is None:
        args.warmup = 2 if config3 else 25

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
    # GLV_BENCH_FORCE_DIST=1 takes the multi-rank code path (RCCL init, barrier, all-reduce, gather)
    # with any world size, to rehearse it on a one-GPU box under torchrun --nproc-per-node 1
    use_dist = world > 1 or bool(os.environ.get("GLV_BENCH_FORCE_DIST"))
    if use_dist:
        dist.init_process_group("nccl", device_id=device)

    from helpers import MEAN, STD
    from tts_amd.distributed import SHARDED_FRAMES_PER_RUN, shard_by_length
    from tts_amd.vocoder import GriffinLimPlan

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
        mel_host = make_batch(0, [1])[:0] if not mine else np.stack(
            [np.pad(m, ((0, max(frames_list) - m.shape[0]), (0, 0))) for m in
             (make_batch(1234 + i, [all_frames[i]])[0] for i in mine)])
        utt_ids = list(mine)
        fpr = args.frames_per_run or SHARDED_FRAMES_PER_RUN
    else:
        mine = list(range(total_batch))
        frames_list = all_frames
        mel_host = make_batch(1234 + 100000 * rank, frames_list)
        utt_ids = None
        fpr = args.frames_per_run
    batch = len(frames_list)
    t_max = mel_host.shape[1]
    plan = GriffinLimPlan(N_FFT, HOP, N_FFT, N_MELS, t_max, batch, device)
    plan.set_mel_basis(SR, 0.0, FMAX)
    plan.set_option("timing", 1)
    plan.set_utterance_ids(utt_ids)
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
    mel = torch.from_numpy(mel_host).to(device)
    n_samples = [HOP * (t - 1) for t in frames_list]
    y_len = HOP * (t_max - 1)
    out = torch.empty((batch, y_len), dtype=torch.float32, device=device)
    nnls = args.inversion == "nnls"
    nnls_ms = []
    if nnls:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    # config 3: every rank contributes a (rows, samples) block of the common shape to one gather
    gather_parts = None
    if config3 and use_dist:
        rows = max(len(s) for s in shards)
        block = torch.zeros((rows, HOP * (max(all_frames) - 1)), dtype=torch.float32, device=device)
        if rank == 0:
            gather_parts = [torch.empty_like(block) for _ in range(world)]

    def step(i):
        if nnls:
            # librosa.feature.inverse.mel_to_stft in place of the pinv product, then the same loop
            ev[0].record()
            power = torch.pow(10.0, 0.1 * (mel * float(STD) + float(MEAN)))
            s = plan.mel_invert_nnls(power, n_steps=args.nnls_steps, power=1.0)
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
            dist.gather(block, gather_parts, dst=0)

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
    for i in range(3 if config3 else 5):
        step(i)
        loop_ms.append(plan.last_loop_ms())
        if nnls:
            torch.cuda.synchronize(device)
            nnls_ms.append(ev[0].elapsed_time(ev[1]))
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    # (GLV_BENCH_NO_CHECK: timing-only experiment builds of the library produce garbage)
    assert (args.traffic_only or args.alias_utts or os.environ.get("GLV_BENCH_NO_CHECK")
            or bool(torch.isfinite(out).all())), "non-finite audio"

    if rank == 0:
        # whole-job audio: config 3 counts every utterance once, config 1 a batch per rank
        audio_s = (sum(HOP * (t - 1) for t in all_frames) / SR) if config3 else world * sum(n_samples) / SR
        ms_per_step = 1e3 * elapsed / args.steps
        p_half, f_bins, r_cover = N_FFT // 2, N_FFT // 2 + 1, N_FFT // HOP
        # bytes one launch of the iteration kernel cannot avoid (this rank's batch), and the
        # bytes of the unfused chain of the same iteration
        # (magnitudes: P floats per frame, or 3P/4 where the upper quarter of every lane's bins has
        # all-zero rows in the inverse basis and the kernel neither stores nor fetches them)
        from tts_amd.filters import inverse_mel_basis
        nz_rows = np.nonzero(np.abs(inverse_mel_basis(SR, N_FFT, N_MELS, 0.0, FMAX)).sum(axis=1))[0]
        first_skipped = (3 * p_half) // 4 if N_FFT == 1024 else 512 + 256
        s_floats = (3 * p_half) // 4 if (int(nz_rows.max()) < first_skipped and not nnls and not args.unfused) else p_half
        fused_bytes = sum(t * (4 * s_floats + 12 * HOP) + (r_cover - 1) * 12 * HOP for t in frames_list)
        unfused_bytes = sum(44 * f_bins * t + 8 * HOP * (t - 1) for t in frames_list)
        kernel_ms = float(np.mean(loop_ms)) / N_ITER
        achieved = fused_bytes / (kernel_ms * 1e-3) / 1e9
        variant = ("traffic_only" if args.traffic_only else "alias" if args.alias_utts else
                   "recompute" if args.recompute == 1 else "spectral" if args.linear == 0 else "default")
        if fpr:
            variant += f"_fpr{fpr}"
        if args.run_target:
            variant += f"_target{args.run_target}"
        for item in sorted(args.option):
            variant += "_" + item.replace("=", "")
        if args.single_frame >= 0:
            variant += f"_single{args.single_frame}"
        key = traffic_key(N_FFT, HOP, batch, args.frames, args.lengths, variant)
        traffic, traffic_source = measured_traffic(key)
        if config3:
            workload = (f"configs[2]: {total_batch} utterances x {args.lengths} "
                        f"{args.frames if args.lengths == 'fixed' else 'N(560,190)'} frames dealt to {world} GPU(s) by "
                        f"length ({batch} on rank 0), one gather of the waveforms to rank 0 per step")
        else:
            workload = (f"configs[1]: batch {batch} x {args.lengths} "
                        f"{args.frames if args.lengths == 'fixed' else 'N(560,190)'} frames per GPU")
        workload += (f", 80 mels, n_fft {N_FFT}, hop {HOP}, {N_ITER} iterations, momentum {MOMENTUM}, "
                     + ("NNLS mel inversion (%d projected-gradient steps)" % args.nnls_steps if nnls else "pinv mel inversion")
                     + " + Griffin-Lim + peak normalisation")
        result = {
            "metric": "audio-sec/sec (whole node), 60-iter Griffin-Lim 80-mel 22.05kHz; % HBM roofline",
            "value": round(audio_s / (elapsed / args.steps), 2),
            "unit": "audio-sec/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong" if config3 else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload,
                       "global_batch": total_batch if config3 else world * batch,
                       "audio_seconds_per_step": round(audio_s, 3),
                       "realtime_factor": round(audio_s / (elapsed / args.steps), 1),
                       "parallelism": (f"batch-sharded x{world}, final device-side gather" if config3 else
                                       f"batch-sharded x{world}, no data-path collective")},
            "roofline": {"bound": "hbm", "kernel": f"gl_engine_kernel<n_fft {N_FFT}, hop {HOP}, iteration>",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "bytes_model": f"fused iteration: T*(4*{s_floats} + 12H) + (R-1)*12H per utterance and launch "
                                        f"({s_floats} magnitudes per frame, P = {p_half})",
                         "alg_bytes_per_launch": fused_bytes,
                         "kernel_ms_per_launch": round(kernel_ms, 5),
                         "traffic": traffic, "traffic_source": traffic_source,
                         "traffic_key": key, "engine_source_digest": engine_source_digest(),
                         # measured HBM bytes per launch over the same time
                         "traffic_frac": (round(traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                          if traffic else None),
                         "alg_unfused_bytes_per_launch": unfused_bytes,
                         "alg_unfused_ratio": round(unfused_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
        }
        if nnls:
            result["nnls"] = {"ms_per_solve": round(float(np.mean(nnls_ms)), 4), "steps": args.nnls_steps,
                              # flops of the two dense products per step; the solver skips the all-zero
                              # blocks of the banded mel basis, so it executes about a quarter of them
                              # (executed MFMA work and busy cycles: profiles/*_nnls_pmc_counters.json)
                              "dense_flop_per_solve": 4 * args.nnls_steps * batch * t_max * f_bins * N_MELS,
                              "dense_equivalent_tflops": round(4 * args.nnls_steps * batch * t_max * f_bins * N_MELS
                                                               / (float(np.mean(nnls_ms)) * 1e-3) / 1e12, 2)}
        if world == 1 and not args.no_cpu:
            result["cpu_baseline"] = cpu_baseline(mel_host, frames_list, MEAN, STD, min(args.cpu_utts, batch))
        print(json.dumps(result), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
