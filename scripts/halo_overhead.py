#!/usr/bin/env python3
"""Per-iteration cost of the time-sharded Griffin-Lim at the long-form shard size (BASELINE.json
configs[3]: 51 680 frames over 8 GPUs = 6 460 frames per rank), on the hardware at hand:

    python -m torch.distributed.run --nproc-per-node 1 scripts/halo_overhead.py nccl
    python -m torch.distributed.run --nproc-per-node 2 scripts/halo_overhead.py gloo

Every rank uses cuda:0.  With one RCCL rank there is no neighbour: the figure is the host-side
cost of driving one step per iteration (step launch + the exchange call) against the same
iterations replayed from a hipGraph.  With two gloo ranks the halos really travel (through host
memory, with a stream drain per iteration - the price of a transport that cannot take device
pointers; RCCL needs neither).  Rank 0 prints one JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from helpers import MEAN, STD, synthetic_mel
    from tts_amd.distributed import griffin_lim_time_sharded
    from tts_amd.vocoder import GriffinLimPlan
    backend = sys.argv[1]
    per_rank = int(sys.argv[2]) if len(sys.argv) > 2 else 6460
    n_fft, hop, n_iter = 1024, 256, 60
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)
    rank, world = dist.get_rank(), dist.get_world_size()
    frames = per_rank * world
    mel = np.concatenate([synthetic_mel(900 + i, 430) for i in range(frames // 430 + 1)])[:frames]

    def run():
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        griffin_lim_time_sharded(mel, MEAN, STD, n_fft=n_fft, hop_length=hop, n_iter=n_iter, seed=3, device=dev)
        torch.cuda.synchronize()
        dist.barrier()
        return time.perf_counter() - t0

    run()
    sharded = min(run() for _ in range(3))
    out = None
    if rank == 0:
        # the same number of local frames, unsharded, iterations replayed from a graph
        plan = GriffinLimPlan(n_fft, hop, n_fft, 80, per_rank, 1, dev)
        plan.set_mel_basis(22050, 0.0, 8000)
        plan.set_option("timing", 1)
        plan.set_option("single_frame", 0)
        m = torch.from_numpy(mel[:per_rank][None]).to(dev)
        y = torch.empty((1, hop * (per_rank - 1)), dtype=torch.float32, device=dev)
        for _ in range(3):
            plan.mel_to_audio(m, MEAN, STD, [per_rank], seed=3, n_iter=n_iter, out=y)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        plan.mel_to_audio(m, MEAN, STD, [per_rank], seed=3, n_iter=n_iter, out=y)
        torch.cuda.synchronize()
        whole = time.perf_counter() - t0
        out = {"backend": backend, "ranks_on_one_gpu": world, "frames_per_rank": per_rank, "n_iter": n_iter,
               "sharded_ms_per_iteration": round(1e3 * sharded / (n_iter + 1), 4),
               "sharded_ms_total": round(1e3 * sharded, 3),
               "unsharded_same_frames_ms_total": round(1e3 * whole, 3),
               "unsharded_kernel_ms_per_iteration": round(plan.last_loop_ms() / n_iter, 4)}
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
