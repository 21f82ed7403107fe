"""Worker of test_time_sharded_two_processes: one rank of a gloo (or one-rank RCCL) group, every rank on cuda:0.
Rank 0 assembles the waveform and compares it with the unsharded call; its exit code is the verdict."""
import os
import sys

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
    n_fft, hop, frames, n_iter = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 6
    backend = sys.argv[4] if len(sys.argv) > 4 else "gloo"
    transport = sys.argv[5] if len(sys.argv) > 5 else None
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
    else:
        dist.init_process_group(backend)
    rank, world = dist.get_rank(), dist.get_world_size()
    gather_device = "cuda:0" if backend == "nccl" else "cpu"
    mel = synthetic_mel(77, frames)
    y, n_begin = griffin_lim_time_sharded(mel, MEAN, STD, n_fft=n_fft, hop_length=hop, n_iter=n_iter, seed=5,
                                          device=torch.device("cuda:0"), transport=transport)
    piece = y.to(gather_device)
    meta = torch.tensor([n_begin, piece.numel()], dtype=torch.int64, device=gather_device)
    metas = [torch.zeros(2, dtype=torch.int64, device=gather_device) for _ in range(world)]
    dist.all_gather(metas, meta)
    size = max(int(m[1]) for m in metas)
    padded = torch.zeros(size, dtype=torch.float32, device=gather_device)
    padded[:piece.numel()] = piece
    parts = [torch.zeros(size, dtype=torch.float32, device=gather_device) for _ in range(world)]
    dist.all_gather(parts, padded)
    parts = [p.cpu() for p in parts]
    ok = True
    if rank == 0:
        total = hop * (frames - 1)
        whole = np.zeros(total, np.float32)
        covered = np.zeros(total, bool)
        for m, part in zip(metas, parts):
            b, n = int(m[0]), int(m[1])
            ok &= not covered[b:b + n].any()
            whole[b:b + n] = part[:n].numpy()
            covered[b:b + n] = True
        ok &= bool(covered.all())
        # the unsharded call with the same device-side phase stream (seed 5, indexed by position)
        plan = GriffinLimPlan(n_fft, hop, n_fft, 80, frames, 1, torch.device("cuda:0"))
        plan.set_mel_basis(22050, 0.0, 8000)
        out = torch.empty((1, total), dtype=torch.float32, device="cuda:0")
        plan.mel_to_audio(torch.from_numpy(mel[None]).cuda(), MEAN, STD, [frames], phase0=None, seed=5,
                          n_iter=n_iter, momentum=0.99, out=out)
        ref = out[0].cpu().numpy()
        err = float(np.linalg.norm(whole - ref) / np.linalg.norm(ref))
        print(f"time-sharded over {world} processes vs one call: rel L2 {err:.3e}, peak {np.abs(whole).max():.6f}")
        ok &= err < 1e-4 and abs(float(np.abs(whole).max()) - 1.0) < 1e-4
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
