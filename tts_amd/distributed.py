"""Multi-GPU use of the vocoder: one process per GPU, `torch.distributed` for the little
communication there is.

* Batch sharding: utterances are independent, so each rank synthesises its own subset and
  nothing is exchanged until an optional final gather of the waveforms.
* Time sharding of one long utterance: every rank owns a contiguous range of frames and,
  once per Griffin-Lim iteration, swaps the (n_fft - hop) overlap-add samples at each shard
  boundary with its two neighbours (point-to-point, no ring).

The reference has no counterpart for either (it synthesises one utterance on one CPU
thread, vocoder/mel_audio.py:138); the functions only orchestrate calls of the
single-GPU entry points.
"""
from __future__ import annotations

import inspect
import socket

import numpy as np
import torch
import torch.distributed as dist

__all__ = ["shard_by_length", "synthesize_sharded", "exchange_halos", "split_frames",
           "griffin_lim_time_sharded"]


def shard_by_length(lengths, world_size):
    """Deals utterance indices to ranks: longest first, each to the least loaded rank
    (load = total frames).  Returns a list of index lists, one per rank, each in ascending
    index order.  Deterministic, identical on every rank."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    loads = [0] * world_size
    shards = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        shards[r].append(i)
        loads[r] += int(lengths[i])
    return [sorted(s) for s in shards]


def _default_compute(mels, mean, std, **kw):
    from .vocoder import griffin_lim_batch
    return griffin_lim_batch(mels, mean, std, **kw)


# frames one wavefront walks in sharded synthesis: fixed, so that an utterance is cut into the
# same runs - and gives the same bits - on every device count (the library's automatic run
# table is fitted to the whole batch a device holds).  18 is where the library's launches that
# span all iterations of a group of utterances do best (4096 utterances on one MI355X: 152 k
# audio-s/s against 139-145 k with 32, 24 or 14).
SHARDED_FRAMES_PER_RUN = 18


def synthesize_sharded(mels, mean, std, *, group=None, gather_dst=0, compute_fn=None, seed=0,
                       frames_per_run=SHARDED_FRAMES_PER_RUN, return_tensors=False, gather_chunks=1, **kw):
    """Batch-sharded synthesis.  `mels` is the full list of (T_i, n_mels) arrays, present on
    every rank.  Each rank synthesises `shard_by_length(...)[rank]` with `compute_fn`
    (default: `griffin_lim_batch` on its GPU) - no communication - and, unless `gather_dst`
    is None, ONE gather brings the padded (rows, samples) blocks to rank `gather_dst`: device
    tensors under RCCL, host tensors under a transport that only moves host memory (gloo).

    `gather_chunks` > 1 cuts every rank's shard into that many pieces and starts the gather of a
    piece as soon as it is synthesised, so that the transfer of piece i runs (on the
    communicator's stream) under the synthesis of piece i + 1 and only the last piece's transfer
    is exposed; the result is the same bits.

    Every utterance keeps its identity (its index in `mels`) for the device-drawn initial phase
    and is cut into runs by a rule that depends on its own length only, so the waveforms are
    bit-identical for every world size, 1 included.

    `gather_dst` is a rank of `group` (not a global rank).  `frames_per_run=None` leaves the run
    table to the library (fitted to each rank's batch: faster for small shards, but run boundaries
    then depend on the world size and results agree to rounding only).

    `compute_fn(mels, mean, std, **kw) -> (audio (rows, samples), ...)` replaces the default
    (`griffin_lim_batch`); of the keywords `utt_ids`, `seed` and `frames_per_run` it is handed
    those its signature accepts (all of them if it takes **kwargs).

    Returns on the gathering rank (or on every rank if it is not distributed) a list of
    1-D arrays in the order of `mels` - numpy, or with `return_tensors` views of the gathered
    tensors where they are; on other ranks, or with `gather_dst=None`, a dict
    {global index: 1-D array} of the local results.
    """
    compute_fn = compute_fn or _default_compute
    params = inspect.signature(compute_fn).parameters
    takes_any = any(p.kind is inspect.Parameter.VAR_KEYWORD for p in params.values())

    def identity_kw(part):
        extra = {"utt_ids": list(part), "seed": seed, "frames_per_run": frames_per_run}
        return {k: v for k, v in extra.items() if takes_any or k in params}
    distributed = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if distributed else 1
    rank = dist.get_rank(group) if distributed else 0
    lengths = [int(m.shape[0]) for m in mels]
    shards = shard_by_length(lengths, world)
    mine = shards[rank]
    hop = kw.get("hop_length", 256)
    n_out = [hop * (t - 1) for t in lengths]
    max_len = max(n_out)
    gathering = distributed and gather_dst is not None
    # piece c of a rank = a contiguous slice of its (index-ordered) shard; every rank derives all
    # pieces of all ranks from the lengths it knows, hence the common block shapes: no size exchange
    n_chunks = max(1, int(gather_chunks)) if gathering else 1

    def piece(idxs, c):
        per = -(-len(idxs) // n_chunks)
        return idxs[c * per:(c + 1) * per]

    on_device = distributed and dist.get_backend(group) == "nccl"

    def finish(row):
        return row if return_tensors else row.cpu().numpy()

    local, pending = {}, []
    for c in range(n_chunks):
        part = piece(mine, c)
        audio = None
        if part:
            audio = compute_fn([mels[i] for i in part], mean, std, **identity_kw(part), **kw)[0]
            audio = audio.detach() if isinstance(audio, torch.Tensor) else torch.from_numpy(np.asarray(audio))
            for r, i in enumerate(part):
                local[i] = audio[r, : n_out[i]]
        if not gathering:
            continue
        rows = max(len(piece(sh, c)) for sh in shards)
        if rows == 0:
            continue
        where = audio.device if (audio is not None and on_device) else (
            torch.device("cuda", torch.cuda.current_device()) if on_device else torch.device("cpu"))
        block = torch.zeros((rows, max_len), dtype=torch.float32, device=where)
        if audio is not None:
            block[: audio.shape[0], : audio.shape[1]] = audio[:, :max_len].to(where)
        parts = [torch.empty_like(block) for _ in range(world)] if rank == gather_dst else None
        # asynchronous: the transfer waits for this piece on the device and runs beside the next one
        # (dist.gather names its destination by global rank)
        dst = dist.get_global_rank(group, gather_dst) if group is not None else gather_dst
        work = dist.gather(block, parts, dst=dst, group=group, async_op=True)
        pending.append((c, block, parts, work))
    if not distributed:
        return [finish(local[i]) for i in range(len(mels))]
    if gather_dst is None:
        return {i: finish(v) for i, v in local.items()}
    for _, _, _, work in pending:
        work.wait()
    if rank != gather_dst:
        return {i: finish(v) for i, v in local.items()}
    out = [None] * len(mels)
    for c, _, parts, _ in pending:
        for r, idxs in enumerate(shards):
            for k, i in enumerate(piece(idxs, c)):
                out[i] = finish(parts[r][k, : n_out[i]])
    return out


def split_frames(n_frames, world_size, min_frames=1):
    """Contiguous, near-equal frame ranges [(t0, t1), ...] for time sharding."""
    if n_frames < world_size * min_frames:
        raise ValueError(f"{n_frames} frames cannot give {world_size} shards of >= {min_frames} frames")
    return [(r * n_frames // world_size, (r + 1) * n_frames // world_size) for r in range(world_size)]


def exchange_halos(to_prev, to_next, from_prev, from_next, *, group=None):
    """One halo swap of a time-sharded iteration: send `to_prev` / `to_next` to ranks
    rank-1 / rank+1 and receive `from_prev` / `from_next` from them (line topology: the
    first rank has no previous, the last no next neighbour).  Tensors are the
    (n_fft - hop)-sample overlap-add partial sums at the two shard boundaries."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    # a transport that only moves host memory (gloo) gets host copies of device tensors; a side
    # without a neighbour may be None
    tensors = [t for t in (to_prev, to_next, from_prev, from_next) if t is not None]
    staged = dist.get_backend(group) != "nccl" and bool(tensors) and tensors[0].is_cuda

    def host(t):
        return t.cpu() if (staged and t is not None) else t

    send_prev, send_next = host(to_prev), host(to_next)
    recv_prev = torch.empty_like(send_prev) if (staged and rank > 0) else from_prev
    recv_next = torch.empty_like(send_next) if (staged and rank < world - 1) else from_next
    ops = []
    if rank > 0:
        ops.append(dist.P2POp(dist.isend, send_prev, rank - 1, group))
        ops.append(dist.P2POp(dist.irecv, recv_prev, rank - 1, group))
    if rank < world - 1:
        ops.append(dist.P2POp(dist.isend, send_next, rank + 1, group))
        ops.append(dist.P2POp(dist.irecv, recv_next, rank + 1, group))
    if ops:
        # under RCCL wait() only makes the current stream wait for the communicator's stream:
        # the host does not block, and the next step is enqueued behind the transfer
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if staged:
        if rank > 0:
            from_prev.copy_(recv_prev)
        if rank < world - 1:
            from_next.copy_(recv_next)


def _all_reduce_max(x, group):
    """all-reduce(MAX) of a small device tensor; through the host for transports without device support."""
    if dist.get_backend(group) == "nccl" or not x.is_cuda:
        dist.all_reduce(x, op=dist.ReduceOp.MAX, group=group)
        return x
    host = x.cpu()
    dist.all_reduce(host, op=dist.ReduceOp.MAX, group=group)
    return host.to(x.device)


_HALO_CHAINS = {}


def _check_one_gpu_per_rank(group, world, device):
    """Raises unless every rank of the group holds a different GPU (collective)."""
    props = torch.cuda.get_device_properties(device)
    # the uuid survives a per-process device mask; the index is the fallback
    mine = (socket.gethostname(), str(getattr(props, "uuid", int(device.index or 0))))
    seen = [None] * world
    dist.all_gather_object(seen, mine, group=group)
    if len(set(seen)) != world:
        raise RuntimeError("transport='library' needs one GPU per rank: several ranks of this group share a "
                           "device (use transport='torch')")


def _halo_chain(group, rank, world, device):
    """The library-side communicator of `group` on this device: made on first use (collective:
    rank 0's id travels through the group), then kept for the life of the process."""
    from .vocoder import HaloChain
    # (keyed by the group's members, not the group object, whose identity can be reused)
    members = tuple(dist.get_process_group_ranks(group)) if group is not None else None
    key = (members, world, rank, str(device))
    chain = _HALO_CHAINS.get(key)
    if chain is None:
        box = [HaloChain.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        chain = _HALO_CHAINS[key] = HaloChain(box[0], rank, world, device)
    return chain


def griffin_lim_time_sharded(mel_db, mean, std, sr=22050, n_fft=1024, hop_length=256, win_length=None,
                             n_mels=80, fmax=8000, *, n_iter=60, momentum=0.99, seed=0, group=None,
                             device=None, normalize=True, transport=None):
    """Long-form synthesis of ONE utterance sharded along time over the ranks of `group`
    (for example a 10-minute mel over 8 GPUs).

    `mel_db` is the whole (T, n_mels) mel, present on every rank; rank r keeps frames
    `split_frames(T, world)[r]`, inverts and iterates only those, and swaps n_fft - hop
    samples with each neighbour once per iteration (61 swaps for 60 iterations).  The peak
    for the final normalisation is one all-reduce(MAX) of one float.

    `transport`: "torch" (the default) - torch.distributed point-to-point on views of the plan's
    workspace; "library" - the swap is RCCL send / recv issued by libglvoc itself on the compute
    stream (glv_halo_create / glv_shard_exchange; a second communicator beside torch's, made once
    per group from an id that rank 0 broadcasts through it).  "library" needs the RCCL backend and
    a GPU of its own for every rank (checked; two ranks on one GPU cannot join one communicator)
    and has so far only run with a single rank talking to itself, so it is opt-in.

    Returns (y_local, n_begin): a float32 device tensor with this rank's samples
    [n_begin, n_begin + len(y_local)) of the waveform.
    """
    from .vocoder import get_plan
    win_length = n_fft if win_length is None else win_length
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    mel = mel_db if isinstance(mel_db, torch.Tensor) else torch.from_numpy(
        np.ascontiguousarray(mel_db, dtype=np.float32))
    frames = mel.shape[0]
    r_cover = n_fft // hop_length
    t0, t1 = split_frames(frames, world, min_frames=2 * (r_cover - 1))[rank]
    plan = get_plan(n_fft, hop_length, win_length, n_mels, max(t1 - t0, 2), 1, device)
    plan.set_mel_basis(sr, 0.0, fmax)
    local = mel[t0:t1].to(device=plan.device, dtype=torch.float32).contiguous()
    s_local = plan.mel_invert(local.unsqueeze(0), float(mean), float(std))[0]
    plan.shard_begin(s_local, t0, frames, seed=seed)

    on_device = dist.get_backend(group) == "nccl"
    if transport is None:
        transport = "torch"
    if transport not in ("library", "torch"):
        raise ValueError("transport must be 'library' or 'torch'")
    if transport == "library":
        if not on_device:
            raise ValueError("transport='library' needs the RCCL ('nccl') backend")
        _check_one_gpu_per_rank(group, world, plan.device)
        plan.halo_attach(_halo_chain(group, rank, world, plan.device))

    def swap():
        # The halos are sent from and received into the plan's workspace where they lie: no
        # packing.  Under RCCL the transfers are ordered on the device after the step that wrote
        # them and before the step that reads them (wait() makes the compute stream wait for the
        # communicator's, not the host); a host-memory transport has to drain the stream first.
        to_prev, to_next, from_prev, from_next = plan.shard_halo_views()
        if not on_device:
            torch.cuda.current_stream(plan.device).synchronize()
        exchange_halos(to_prev, to_next, from_prev, from_next, group=group)

    if transport == "library":
        plan.shard_run(n_iter, momentum)
    else:
        swap()
        for _ in range(n_iter):
            plan.shard_step(momentum)
            swap()
    y, n_begin = plan.shard_end()
    if transport == "library":
        plan.halo_attach(None)
    if normalize and y.numel():
        row = y.unsqueeze(0)
        peak = _all_reduce_max(plan.peak_reduce(row), group)
        plan.peak_scale(row, peak)
    elif normalize:
        _all_reduce_max(torch.zeros((1,), dtype=torch.float32, device=plan.device), group)
    return y, n_begin
