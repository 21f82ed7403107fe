"""Worker of test_batch_sharded_processes: one rank of a process group, every rank on cuda:0.
Backend from argv (gloo, or nccl with a single rank).  Rank 0 compares the gathered waveforms
(a) with the CPU oracle on the same explicit phases and (b), for the device-drawn phase, bit for
bit with the same utterances synthesised in one call; its exit code is the verdict."""
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
    from oracle import griffinlim_oracle as O
    from tts_amd import griffin_lim_batch
    from tts_amd.distributed import SHARDED_FRAMES_PER_RUN, synthesize_sharded
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
    rng = np.random.default_rng(3)
    angles = [2 * np.pi * rng.random((t, 513)) for t in lens]

    def phase_block(idx):
        t_max = max(lens[i] for i in idx)
        p = np.zeros((len(idx), t_max, 513, 2), np.float32)
        for r, i in enumerate(idx):
            p[r, :lens[i], :, 0], p[r, :lens[i], :, 1] = np.cos(angles[i]), np.sin(angles[i])
        return torch.from_numpy(p).cuda()

    def compute(sub, mean, std, utt_ids=None, **kwargs):
        return griffin_lim_batch(sub, mean, std, phase0=phase_block(utt_ids), utt_ids=utt_ids, **kwargs)

    # (a) explicit phases: against the oracle
    got = synthesize_sharded(mels, MEAN, STD, compute_fn=compute, **kw)
    # (b) phases drawn on the device from (seed, utterance identity): against one call
    drawn = synthesize_sharded(mels, MEAN, STD, seed=17, **kw)
    # the same with the gather cut into pieces that travel under the synthesis of the next piece
    pieces = synthesize_sharded(mels, MEAN, STD, seed=17, gather_chunks=3, **kw)
    ok = True
    if rank == 0:
        worst = 0.0
        for i, t in enumerate(lens):
            ok &= got[i].shape == (256 * (t - 1),)
            ref = O.mel_to_audio_griffin_lim(mels[i], MEAN, STD, n_fft=1024, win_length=1024, hop_length=256,
                                             n_iter=5, init_phase=angles[i].T)
            worst = max(worst, float(np.linalg.norm(got[i] - ref) / max(np.linalg.norm(ref), 1e-30)))
        one, n = griffin_lim_batch(mels, MEAN, STD, seed=17, utt_ids=list(range(len(lens))),
                                   frames_per_run=SHARDED_FRAMES_PER_RUN, **kw)[:2]
        one = one.cpu().numpy()
        identical = all(np.array_equal(drawn[i], one[i, :n[i]]) and np.array_equal(pieces[i], one[i, :n[i]])
                        for i in range(len(lens)))
        print(f"batch-sharded over {dist.get_world_size()} processes ({backend}): worst rel L2 vs oracle "
              f"{worst:.3e}; bit-identical to one call: {identical}")
        ok &= worst < 1e-4 and identical
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
