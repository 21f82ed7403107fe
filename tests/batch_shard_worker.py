"""Worker of test_batch_sharded_processes: one rank of a process group, every rank on cuda:0.
Backend from argv (gloo, or nccl with a single rank).  Rank 0 compares the gathered waveforms
with the same utterances synthesised in one call; its exit code is the verdict."""
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
    from tts_amd import griffin_lim_batch
    from tts_amd.distributed import synthesize_sharded
    backend = sys.argv[1]
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
    else:
        dist.init_process_group(backend)
    rank = dist.get_rank()
    lens = [40, 12, 75, 33, 2, 58, 20]
    mels = [synthetic_mel(300 + i, t) for i, t in enumerate(lens)]
    kw = dict(n_fft=1024, win_length=1024, hop_length=256, n_iter=5)
    # explicit phases, so that a row does not depend on its position in a batch
    rng = np.random.default_rng(3)
    angles = [2 * np.pi * rng.random((t, 513)) for t in lens]

    def phase_block(idx):
        t_max = max(lens[i] for i in idx)
        p = np.zeros((len(idx), t_max, 513, 2), np.float32)
        for r, i in enumerate(idx):
            p[r, :lens[i], :, 0], p[r, :lens[i], :, 1] = np.cos(angles[i]), np.sin(angles[i])
        return torch.from_numpy(p).cuda()

    def compute(sub, mean, std, **kwargs):
        idx = [next(i for i, m in enumerate(mels) if m is s) for s in sub]
        return griffin_lim_batch(sub, mean, std, phase0=phase_block(idx), **kwargs)

    got = synthesize_sharded(mels, MEAN, STD, compute_fn=compute, **kw)
    ok = True
    if rank == 0:
        ref, n = griffin_lim_batch(mels, MEAN, STD, phase0=phase_block(list(range(len(lens)))), **kw)[:2]
        ref = ref.cpu().numpy()
        worst = 0.0
        for i, t in enumerate(lens):
            ok &= got[i].shape == (256 * (t - 1),)
            err = float(np.linalg.norm(got[i] - ref[i, :n[i]]) / max(np.linalg.norm(ref[i, :n[i]]), 1e-30))
            worst = max(worst, err)
        print(f"batch-sharded over {dist.get_world_size()} processes ({backend}) vs one call: worst rel L2 {worst:.3e}")
        ok &= worst < 1e-5
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
