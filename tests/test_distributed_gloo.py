"""Two CPU processes over gloo: the sharding, gather and halo-exchange plumbing of
tts_amd.distributed, with the oracle standing in for the GPU compute (the same
orchestration code drives RCCL on GPUs)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_batch(mels, mean, std, hop_length=256, n_fft=1024, utt_ids=None, seed=0, **kw):
    """compute_fn with the signature of griffin_lim_batch; the phase stream is keyed by the
    utterance's identity, so the result does not depend on which rank gets the utterance."""
    from oracle import griffinlim_oracle as O
    rows, n = [], []
    for m, uid in zip(mels, utt_ids):
        rs = 1000 * seed + uid
        y = O.mel_to_audio_griffin_lim(m, mean, std, n_fft=n_fft, win_length=n_fft, hop_length=hop_length,
                                       n_iter=2, random_state=rs)
        rows.append(y)
        n.append(len(y))
    out = np.zeros((len(rows), max(n)), np.float32)
    for i, y in enumerate(rows):
        out[i, : len(y)] = y
    return torch.from_numpy(out), n


def _make_mels():
    from helpers import synthetic_mel
    return [synthetic_mel(10 + i, t) for i, t in enumerate([12, 30, 7, 22, 16])]


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import MEAN, STD
    from tts_amd import distributed as D
    mels = _make_mels()
    res = D.synthesize_sharded(mels, MEAN, STD, compute_fn=_oracle_batch, hop_length=256, n_fft=1024)
    # the gather in pieces (each started as soon as its piece is synthesised) brings the same arrays
    for chunks in (2, 3, 9):
        again = D.synthesize_sharded(mels, MEAN, STD, compute_fn=_oracle_batch, hop_length=256, n_fft=1024,
                                     gather_chunks=chunks)
        if rank == 0:
            assert len(again) == len(res) and all(np.array_equal(a, b) for a, b in zip(again, res)), chunks
        else:
            assert sorted(again) == sorted(res)
    if rank == 0:
        np.savez(os.path.join(tmpdir, "gathered.npz"), *res)
    else:
        assert isinstance(res, dict) and sorted(res) == D.shard_by_length([m.shape[0] for m in mels], world)[rank]
    # halo exchange: every rank sends its rank number both ways
    n = 768
    to_prev = torch.full((n,), float(10 * rank + 1))
    to_next = torch.full((n,), float(10 * rank + 2))
    from_prev, from_next = torch.zeros(n), torch.zeros(n)
    D.exchange_halos(to_prev, to_next, from_prev, from_next)
    if rank > 0:
        assert torch.all(from_prev == 10 * (rank - 1) + 2)
    else:
        assert torch.all(from_prev == 0)
    if rank < world - 1:
        assert torch.all(from_next == 10 * (rank + 1) + 1)
    else:
        assert torch.all(from_next == 0)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_by_length_balances_and_is_deterministic():
    from tts_amd.distributed import shard_by_length, split_frames
    lengths = [865, 100, 430, 430, 96, 700, 512, 300]
    shards = shard_by_length(lengths, 3)
    assert sorted(i for s in shards for i in s) == list(range(len(lengths)))
    loads = [sum(lengths[i] for i in s) for s in shards]
    assert max(loads) - min(loads) <= max(lengths)
    assert shards == shard_by_length(lengths, 3)
    assert shard_by_length(lengths, 1) == [list(range(len(lengths)))]
    assert shard_by_length([5, 5], 4)[2:] == [[], []]
    assert split_frames(51680, 8) == [(6460 * r, 6460 * (r + 1)) for r in range(8)]
    with pytest.raises(ValueError):
        split_frames(10, 8, min_frames=6)


def test_two_rank_sharded_synthesis_equals_single_process(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import MEAN, STD
    from tts_amd import distributed as D
    mels = _make_mels()
    single = D.synthesize_sharded(mels, MEAN, STD, compute_fn=_oracle_batch, hop_length=256, n_fft=1024)
    got = np.load(os.path.join(str(tmp_path), "gathered.npz"))
    assert len(got.files) == len(mels)
    for i, y in enumerate(single):
        g = got[f"arr_{i}"]
        assert g.shape == (256 * (mels[i].shape[0] - 1),)
        assert np.array_equal(g, y)
