"""Mel -> waveform Griffin-Lim vocoder on MI355X.

Drop-in for the reference function

    mel_to_audio_griffin_lim(mel_db, mean, std, sr=22050, n_fft=2048, hop_length=256,
                             win_length=2048, n_mels=80, fmax=8000)

(reference vocoder/mel_audio.py:112-132; copies at vocoder/pra.py:65-80 and
test.py:143-158) with the same positional signature, plus keyword-only extras
whose defaults reproduce the reference behaviour, and a batched entry point.
All arithmetic runs in libglvoc.so (hand-written HIP kernels); torch only owns
device memory and the stream.
"""
from __future__ import annotations

import contextlib
import ctypes
import sys
import threading

import numpy as np
import torch

from . import _lib
from .filters import inverse_mel_basis, mel_basis_and_step

__all__ = ["GriffinLimPlan", "HaloChain", "get_plan", "mel_to_audio_griffin_lim", "griffin_lim_batch", "mel_to_audio",
           "audio_to_mel_db"]


def _stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _int_array(values):
    arr = np.ascontiguousarray(values, dtype=np.int32)
    return arr, arr.ctypes.data_as(_lib.c_int_p)


class _DeviceFloats:
    """n float32 values at a raw device address, presented through the array interface torch reads."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}


class HaloChain:
    """This process's place in a chain of time shards: an RCCL communicator inside libglvoc
    (glv_halo_create).  `unique_id()` on one rank, the bytes handed to all, then every rank
    constructs its HaloChain (collective).  Serves any number of plans and calls."""

    def __init__(self, unique_id, rank, world, device=None):
        self.lib = _lib.load()
        if len(unique_id) != 128:
            raise ValueError("unique_id must be the 128 bytes HaloChain.unique_id() returned")
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        handle = ctypes.c_void_p()
        _lib.check(self.lib.glv_halo_create(ctypes.byref(handle), ctypes.c_char_p(bytes(unique_id)), int(rank),
                                            int(world), int(device.index or 0)))
        self._handle, self.rank, self.world, self.device = handle, int(rank), int(world), device

    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(128)
        _lib.check(_lib.load().glv_halo_unique_id(buf))
        return buf.raw

    def close(self):
        if self._handle is not None:
            self.lib.glv_halo_destroy(self._handle)
            self._handle = None

    def __del__(self):
        # at interpreter shutdown the runtimes underneath may be gone already: leave it to the OS
        if sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass


class GriffinLimPlan:
    """Owns one ``glv_plan``: tables and device workspace for a fixed
    (n_fft, hop_length, win_length, n_mels) and capacity (max_frames, max_batch).

    Tensors passed in must be float32, contiguous and on the plan's device; the
    methods enqueue work on torch's current stream for that device and return
    without synchronising.
    """

    def __init__(self, n_fft=2048, hop_length=256, win_length=None, n_mels=80, max_frames=865,
                 max_batch=1, device=None):
        self._handle = None
        if not torch.cuda.is_available():
            raise RuntimeError("tts_amd needs a ROCm GPU: torch.cuda.is_available() is False "
                               "(there is no CPU fallback)")
        self.lib = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None \
            else torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.n_fft = int(n_fft)
        self.hop_length = int(hop_length)
        self.win_length = int(win_length) if win_length is not None else self.n_fft
        self.n_mels = int(n_mels)
        self.max_frames = int(max_frames)
        self.max_batch = int(max_batch)
        self.n_bins = self.n_fft // 2 + 1
        handle = ctypes.c_void_p()
        _lib.check(self.lib.glv_plan_create(ctypes.byref(handle), self.n_fft, self.hop_length,
                                            self.win_length, self.n_mels, self.max_frames,
                                            self.max_batch, self.device.index))
        self._handle = handle
        self._basis_key = None
        self._options = {}
        self._utt_ids = None

    def close(self):
        if self._handle is not None:
            self.lib.glv_plan_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- configuration -------------------------------------------------------
    def set_option(self, name, value):
        _lib.check(self.lib.glv_plan_set_option(self._handle, name.encode(), int(value)))
        self._options[name] = int(value)

    @contextlib.contextmanager
    def call_state(self, utt_ids, frames_per_run):
        """Utterance identities and run length for the calls inside the block only: what the plan
        held before (a user's own set_utterance_ids / set_option on a shared plan) is put back."""
        old_ids, old_fpr = self._utt_ids, self._options.get("frames_per_run", 0)
        self.set_utterance_ids(utt_ids)
        self.set_option("frames_per_run", 0 if frames_per_run is None else int(frames_per_run))
        try:
            yield self
        finally:
            self.set_utterance_ids(old_ids)
            self.set_option("frames_per_run", old_fpr)

    @property
    def workspace_bytes(self):
        return int(self.lib.glv_plan_workspace_bytes(self._handle))

    def reserve(self, batch, frames, n_iter=60, single_frame=False):
        """Sets aside the usage-dependent workspace for calls of this shape ahead of time
        (glv_plan_reserve), so that none of them has to grow it and drain the stream."""
        _lib.check(self.lib.glv_plan_reserve(self._handle, int(batch), int(frames), int(n_iter),
                                             int(bool(single_frame)), _stream_ptr(self.device)))

    def set_utterance_ids(self, ids):
        """Identities keying the device-drawn initial phase of the following calls: utterance b
        draws the stream (seed, ids[b]).  None = identify utterances by their row."""
        if ids is None:
            _lib.check(self.lib.glv_plan_set_utterance_ids(self._handle, None, 0, _stream_ptr(self.device)))
            self._utt_ids = None
            return
        arr = np.ascontiguousarray(ids, dtype=np.int64)
        self._utt_ids = arr
        _lib.check(self.lib.glv_plan_set_utterance_ids(
            self._handle, arr.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), int(arr.shape[0]),
            _stream_ptr(self.device)))

    def status(self):
        """Raises if a wait inside an iteration-spanning launch ran out (glv_plan_status): the
        results of that call are void.  Ask after synchronising with the stream."""
        _lib.check(self.lib.glv_plan_status(self._handle))

    def persistent_capacity(self, variant=1):
        """Workgroups of the iteration-spanning kernel seen resident together at plan creation
        (variant 0: all bins kept, 1: zero-for-good bins left out)."""
        return int(self.lib.glv_plan_persistent_capacity(self._handle, int(variant)))

    def last_loop_launches(self):
        """Kernel launches the iteration loop of the last call took (1: all iterations in one)."""
        return int(self.lib.glv_plan_last_loop_launches(self._handle))

    def last_loop_groups(self):
        """Groups of utterances the last call's loop was cut into (each its own launch or launches)."""
        return int(self.lib.glv_plan_last_loop_groups(self._handle))

    def last_loop_ms(self):
        ms = ctypes.c_float()
        _lib.check(self.lib.glv_plan_last_loop_ms(self._handle, ctypes.byref(ms)))
        return float(ms.value)

    def drawn_phase(self, seed, batch, frames, frame_origin=0):
        """(batch, frames, F, 2) unit phasors: the initial phase a call with ``phase0=None`` and this
        ``seed`` draws on the device for the utterance identities in place (a checking aid: feed
        them to another implementation to compare a device-drawn call element by element)."""
        out = torch.empty((int(batch), int(frames), self.n_bins, 2), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.glv_plan_drawn_phase(self._handle, int(seed) & (2**64 - 1), int(batch), int(frames),
                                                 int(frame_origin), _ptr(out), _stream_ptr(self.device)))
        return out

    def set_inverse_basis(self, pinv):
        pinv = np.ascontiguousarray(pinv, dtype=np.float32)
        if pinv.shape != (self.n_bins, self.n_mels):
            raise ValueError(f"inverse basis must be ({self.n_bins}, {self.n_mels}), got {pinv.shape}")
        _lib.check(self.lib.glv_plan_set_inverse_basis(self._handle, pinv.ctypes.data_as(_lib.c_float_p)))

    def set_mel_basis(self, sr, fmin=0.0, fmax=None):
        key = (int(sr), float(fmin), None if fmax is None else float(fmax))
        if key != self._basis_key:
            args = (int(sr), self.n_fft, self.n_mels, float(fmin), None if fmax is None else float(fmax))
            self.set_inverse_basis(inverse_mel_basis(*args))
            basis, self._nnls_step = mel_basis_and_step(*args)
            _lib.check(self.lib.glv_plan_set_mel_basis(self._handle, basis.ctypes.data_as(_lib.c_float_p)))
            self._basis_key = key

    # -- helpers ---------------------------------------------------------------
    def _check(self, t, shape_tail, name):
        if not (isinstance(t, torch.Tensor) and t.dtype == torch.float32 and t.is_contiguous()
                and t.device == self.device):
            raise ValueError(f"{name} must be a contiguous float32 tensor on {self.device}")
        if tuple(t.shape[-len(shape_tail):]) != tuple(shape_tail):
            raise ValueError(f"{name} must end in shape {tuple(shape_tail)}, got {tuple(t.shape)}")

    def _frames(self, n_frames, batch, frames_stride):
        if n_frames is None:
            n_frames = [frames_stride] * batch
        arr, ptr = _int_array(n_frames)
        if arr.shape != (batch,):
            raise ValueError(f"n_frames must have {batch} entries")
        return arr, ptr

    # -- stages ------------------------------------------------------------------
    def mel_invert(self, mel_db, mean, std):
        """(B, T, n_mels) normalised dB -> (B, T, F) linear "magnitudes"
        (reference vocoder/mel_audio.py:114-123)."""
        self._check(mel_db, (self.n_mels,), "mel_db")
        if mel_db.dim() != 3:
            raise ValueError("mel_db must be (B, T, n_mels)")
        b, t, _ = mel_db.shape
        out = torch.empty((b, t, self.n_bins), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.glv_mel_invert(self._handle, _ptr(mel_db), b, t, float(mean), float(std),
                                           _ptr(out), _stream_ptr(self.device)))
        return out

    def mel_invert_nnls(self, mel_power, n_steps=40, power=2.0, tol=0.0):
        """(B, T, n_mels) mel POWER values -> (B, T, F) non-negative magnitudes: the
        mel_to_stft step of librosa.feature.inverse.mel_to_audio (reference
        test_cnn_model.py:117-119) as a projected-gradient solver from librosa's start, its steps
        extrapolated the Nesterov / FISTA way (option "nnls_accel", default on: 40 steps reach a
        lower mel residual than 200 plain ones).  A frame stops once its mel residual is below
        ``tol`` times the norm of its mel (0: all steps)."""
        self._check(mel_power, (self.n_mels,), "mel_power")
        if mel_power.dim() != 3:
            raise ValueError("mel_power must be (B, T, n_mels)")
        if self._basis_key is None:
            raise RuntimeError("set_mel_basis(sr, fmin, fmax) first")
        b, t, _ = mel_power.shape
        out = torch.empty((b, t, self.n_bins), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.glv_mel_invert_nnls(self._handle, _ptr(mel_power), b, t, int(n_steps),
                                                float(self._nnls_step), float(tol), float(power), _ptr(out),
                                                _stream_ptr(self.device)))
        return out

    def griffinlim(self, S, n_frames=None, phase0=None, seed=0, n_iter=60, momentum=0.99,
                   return_sc=False, out=None):
        """librosa.griffinlim on a (B, T, F) batch (reference vocoder/mel_audio.py:126).

        ``phase0``: (B, T, F, 2) unit phasors, or None to draw them on the device
        from ``seed``.  Returns y (B, hop*(T-1)), rows zero-filled past
        hop*(n_frames[b]-1); with ``return_sc`` also the (B, n_iter) spectral
        convergence curve.
        """
        self._check(S, (self.n_bins,), "S")
        if S.dim() != 3:
            raise ValueError("S must be (B, T, F)")
        b, t, _ = S.shape
        if phase0 is not None:
            self._check(phase0, (t, self.n_bins, 2), "phase0")
        nf, nf_ptr = self._frames(n_frames, b, t)
        y_stride = self.hop_length * (t - 1)
        if out is None:
            out = torch.empty((b, y_stride), dtype=torch.float32, device=self.device)
        else:
            self._check(out, (y_stride,), "out")
        sc = torch.empty((b, n_iter), dtype=torch.float32, device=self.device) \
            if (return_sc and n_iter > 0) else None
        _lib.check(self.lib.glv_griffinlim(self._handle, _ptr(S), _ptr(phase0), int(seed) & (2**64 - 1),
                                           nf_ptr, b, t, int(n_iter), float(momentum), _ptr(out),
                                           y_stride, _ptr(sc), _stream_ptr(self.device)))
        return (out, sc) if return_sc else out

    def mel_to_audio(self, mel_db, mean, std, n_frames=None, phase0=None, seed=0, n_iter=60,
                     momentum=0.99, peak_eps=1e-6, return_sc=False, out=None):
        """mel_invert + griffinlim + peak_normalize as one library call (the whole of reference
        vocoder/mel_audio.py:112-132 on a batch) with bit-identical results: the magnitudes never
        exist as a (B, T, F) array.  ``peak_eps=None`` skips the normalisation."""
        self._check(mel_db, (self.n_mels,), "mel_db")
        if mel_db.dim() != 3:
            raise ValueError("mel_db must be (B, T, n_mels)")
        b, t, _ = mel_db.shape
        if phase0 is not None:
            self._check(phase0, (t, self.n_bins, 2), "phase0")
        nf, nf_ptr = self._frames(n_frames, b, t)
        y_stride = self.hop_length * (t - 1)
        if out is None:
            out = torch.empty((b, y_stride), dtype=torch.float32, device=self.device)
        else:
            self._check(out, (y_stride,), "out")
        sc = torch.empty((b, n_iter), dtype=torch.float32, device=self.device) \
            if (return_sc and n_iter > 0) else None
        _lib.check(self.lib.glv_mel_to_audio(self._handle, _ptr(mel_db), float(mean), float(std),
                                             _ptr(phase0), int(seed) & (2**64 - 1), nf_ptr, b, t,
                                             int(n_iter), float(momentum),
                                             -1.0 if peak_eps is None else float(peak_eps), _ptr(out),
                                             y_stride, _ptr(sc), _stream_ptr(self.device)))
        return (out, sc) if return_sc else out

    def peak_normalize(self, y, n_samples=None, eps=1e-6):
        """In place audio / (max|audio| + eps) per row (reference vocoder/mel_audio.py:129)."""
        self._check(y, (y.shape[-1],), "y")
        b, stride = y.shape
        if n_samples is None:
            n_samples = [stride] * b
        _, ptr = _int_array(n_samples)
        _lib.check(self.lib.glv_peak_normalize(self._handle, _ptr(y), ptr, b, stride, float(eps),
                                               _stream_ptr(self.device)))
        return y

    def peak_reduce(self, y, n_samples=None):
        """max|y| per row as a (B,) float32 device tensor (first half of peak_normalize)."""
        self._check(y, (y.shape[-1],), "y")
        b, stride = y.shape
        _, ptr = _int_array([stride] * b if n_samples is None else n_samples)
        peak = torch.empty((b,), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.glv_peak_reduce(self._handle, _ptr(y), ptr, b, stride, _ptr(peak),
                                            _stream_ptr(self.device)))
        return peak

    def peak_scale(self, y, peak, n_samples=None, eps=1e-6):
        """In place y[b] /= (peak[b] + eps) (second half of peak_normalize)."""
        self._check(y, (y.shape[-1],), "y")
        b, stride = y.shape
        self._check(peak, (b,), "peak")
        _, ptr = _int_array([stride] * b if n_samples is None else n_samples)
        _lib.check(self.lib.glv_peak_scale(self._handle, _ptr(y), ptr, b, stride, _ptr(peak), float(eps),
                                           _stream_ptr(self.device)))
        return y

    # -- time sharding: this plan holds frames [frame_begin, frame_begin + T_local) -----
    def shard_begin(self, S_local, frame_begin, frames_global, phase0=None, seed=0):
        """Starts a time-sharded Griffin-Lim on the local (T_local, F) slice of S; see
        include/glvoc.h (glv_shard_*) for the protocol."""
        self._check(S_local, (self.n_bins,), "S_local")
        if S_local.dim() != 2:
            raise ValueError("S_local must be (T_local, F)")
        t_local = S_local.shape[0]
        if phase0 is not None:
            self._check(phase0, (t_local, self.n_bins, 2), "phase0")
        _lib.check(self.lib.glv_shard_begin(self._handle, _ptr(S_local), _ptr(phase0), int(seed) & (2**64 - 1),
                                            int(frame_begin), int(t_local), int(frames_global),
                                            _stream_ptr(self.device)))
        n = int(self.lib.glv_shard_halo_samples(self._handle))
        if getattr(self, "_halo", None) is None or self._halo[0].numel() != n:
            self._halo = tuple(torch.zeros((n,), dtype=torch.float32, device=self.device) for _ in range(4))

    def shard_step(self, momentum=0.99):
        _lib.check(self.lib.glv_shard_step(self._handle, float(momentum), _stream_ptr(self.device)))

    def shard_halo_pack(self):
        """Returns (to_prev, to_next): the n_fft - hop partial sums each neighbour needs."""
        to_prev, to_next = self._halo[0], self._halo[1]
        _lib.check(self.lib.glv_shard_halo_pack(self._handle, _ptr(to_prev), _ptr(to_next),
                                                _stream_ptr(self.device)))
        return to_prev, to_next

    def shard_halo_buffers(self):
        """(from_prev, from_next) receive buffers to fill before shard_halo_unpack."""
        return self._halo[2], self._halo[3]

    def shard_halo_unpack(self, from_prev=None, from_next=None):
        from_prev = self._halo[2] if from_prev is None else from_prev
        from_next = self._halo[3] if from_next is None else from_next
        _lib.check(self.lib.glv_shard_halo_unpack(self._handle, _ptr(from_prev), _ptr(from_next),
                                                  _stream_ptr(self.device)))

    def shard_halo_views(self):
        """(send_prev, send_next, recv_prev, recv_next): float32 tensors that ALIAS the halo regions
        of the newest generation inside the plan's workspace (None on the side without a
        neighbour) - what a device-pointer transport sends from and receives into, no copies."""
        ptrs = [ctypes.c_void_p() for _ in range(4)]
        _lib.check(self.lib.glv_shard_halo_views(self._handle, *[ctypes.byref(p) for p in ptrs]))
        n = int(self.lib.glv_shard_halo_samples(self._handle))
        cache = self.__dict__.setdefault("_views", {})
        out = []
        for p in ptrs:
            if not p.value:
                out.append(None)
                continue
            if p.value not in cache:
                cache[p.value] = torch.as_tensor(_DeviceFloats(p.value, n), device=self.device)
            out.append(cache[p.value])
        return tuple(out)

    # -- halo exchange done by the library (RCCL on the plan's stream) ----------
    def halo_attach(self, halo):
        """Points the plan at a HaloChain (None detaches); the chain must outlive its use."""
        handle = None if halo is None else halo._handle
        _lib.check(self.lib.glv_halo_attach(self._handle, handle))
        self._halo_chain = halo

    def shard_exchange(self):
        """Swaps the halos of the newest generation with ranks -1 / +1 on the plan's stream."""
        _lib.check(self.lib.glv_shard_exchange(self._handle, _stream_ptr(self.device)))

    def shard_run(self, n_iter, momentum=0.99):
        """The swap after shard_begin, then n_iter steps each followed by its swap, enqueued by
        one library call (needs halo_attach)."""
        _lib.check(self.lib.glv_shard_run(self._handle, int(n_iter), float(momentum), _stream_ptr(self.device)))

    def shard_last_run_form(self):
        """How the last shard_run was enqueued: 0 step by step, 1 replayed from one graph, 2 from
        one graph whose edge workgroups and swaps overlap the interior (option "shard_graph")."""
        return int(self.lib.glv_shard_last_run_form(self._handle))

    def shard_end(self):
        """Returns (y_local, n_begin): this shard's samples [n_begin, n_begin + len(y_local))
        of the whole (un-normalised) waveform."""
        n_begin, n_count = ctypes.c_int64(), ctypes.c_int64()
        _lib.check(self.lib.glv_shard_output_range(self._handle, ctypes.byref(n_begin), ctypes.byref(n_count)))
        y = torch.empty((max(int(n_count.value), 1),), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.glv_shard_end(self._handle, _ptr(y), _stream_ptr(self.device)))
        return y[: int(n_count.value)], int(n_begin.value)

    def _signal_rows(self, y, n_samples):
        """Row lengths of a (B, stride) signal batch -> (n_samples, n_frames) int arrays with
        librosa's frame count 1 + n_samples // hop for a centred signal."""
        b, stride = y.shape
        ns = np.full((b,), stride, np.int32) if n_samples is None else np.ascontiguousarray(n_samples, np.int32)
        if ns.shape != (b,) or ns.min() < self.hop_length or ns.max() > stride:
            raise ValueError(f"n_samples must be {b} values in [hop, {stride}]")
        return ns, (1 + ns // self.hop_length).astype(np.int32)

    def melspectrogram(self, y, n_samples=None):
        """(B, stride) signals of n_samples[b] samples (default: the whole row) -> (B, T, n_mels) mel
        power with T = 1 + stride // hop rows, 1 + n_samples[b] // hop of them used:
        librosa.feature.melspectrogram (reference acoustic/dataset.py:75-83); set_mel_basis first."""
        self._check(y, (y.shape[-1],), "y")
        if self._basis_key is None:
            raise RuntimeError("set_mel_basis(sr, fmin, fmax) first")
        b, stride = y.shape
        ns, nf = self._signal_rows(y, n_samples)
        t = stride // self.hop_length + 1
        out = torch.zeros((b, t, self.n_mels), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.glv_melspectrogram(self._handle, _ptr(y), stride, nf.ctypes.data_as(_lib.c_int_p),
                                               ns.ctypes.data_as(_lib.c_int_p), b, t, _ptr(out),
                                               _stream_ptr(self.device)))
        return out

    def power_to_db(self, x, n_valid=None, ref="max", amin=1e-10, top_db=80.0):
        """librosa.power_to_db row by row on a (B, ...) tensor; ref="max" is ref=np.max
        (reference acoustic/dataset.py:84), a number is used as is; top_db=None disables
        the floor.  n_valid[b] = leading values of the flattened row b that count."""
        b = x.shape[0]
        flat = x.reshape(b, -1)
        self._check(flat, (flat.shape[1],), "x")
        _, ptr = _int_array([flat.shape[1]] * b if n_valid is None else n_valid)
        out = torch.zeros_like(flat)
        ref_is_max = isinstance(ref, str)
        if ref_is_max and ref != "max":
            raise ValueError('ref must be "max" or a number')
        _lib.check(self.lib.glv_power_to_db(self._handle, _ptr(flat), ptr, b, flat.shape[1], int(ref_is_max),
                                            0.0 if ref_is_max else float(ref), float(amin),
                                            -1.0 if top_db is None else float(top_db), _ptr(out),
                                            _stream_ptr(self.device)))
        return out.reshape(x.shape)

    def stft(self, y, n_samples=None):
        """(B, stride) signals of n_samples[b] samples (default: the whole row) -> (B, T, F, 2) with
        T = 1 + stride // hop; librosa.stft(center=True, pad_mode='constant')."""
        self._check(y, (y.shape[-1],), "y")
        b, stride = y.shape
        ns, nf = self._signal_rows(y, n_samples)
        t = stride // self.hop_length + 1
        out = torch.zeros((b, t, self.n_bins, 2), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.glv_stft(self._handle, _ptr(y), stride, nf.ctypes.data_as(_lib.c_int_p),
                                     ns.ctypes.data_as(_lib.c_int_p), b, t, _ptr(out),
                                     _stream_ptr(self.device)))
        return out

    def istft(self, D, n_frames=None):
        """(B, T, F, 2) -> (B, hop*(T-1)); librosa.istft(center=True)."""
        self._check(D, (self.n_bins, 2), "D")
        b, t = D.shape[0], D.shape[1]
        nf, nf_ptr = self._frames(n_frames, b, t)
        stride = self.hop_length * (t - 1)
        out = torch.empty((b, stride), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.glv_istft(self._handle, _ptr(D), nf_ptr, b, t, _ptr(out), stride,
                                      _stream_ptr(self.device)))
        return out


# ---------------------------------------------------------------------------
# plan cache for the functional API
# ---------------------------------------------------------------------------
_plans = {}
_plans_lock = threading.Lock()


def get_plan(n_fft, hop_length, win_length, n_mels, frames, batch, device=None):
    """Returns a cached plan able to hold `batch` utterances of `frames` frames,
    growing (re-creating) it when a larger request arrives."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    key = (int(n_fft), int(hop_length), int(win_length), int(n_mels), dev.index)
    with _plans_lock:
        plan = _plans.get(key)
        if plan is None or plan.max_frames < frames or plan.max_batch < batch:
            max_frames = max(frames, plan.max_frames if plan else 0)
            max_batch = max(batch, plan.max_batch if plan else 0)
            if plan is not None:
                plan.close()
            plan = GriffinLimPlan(n_fft, hop_length, win_length, n_mels, max_frames, max_batch, dev)
            _plans[key] = plan
        return plan


def _phase_tensor(init_phase, n_bins, frames, device):
    """(F, T) phasors / angles / int seed -> (1, T, F, 2) float32 device tensor."""
    if isinstance(init_phase, (int, np.integer)):
        # librosa: rng = np.random.default_rng(seed); angles = phasor(2*pi*rng.random(S.shape))
        rng = np.random.default_rng(int(init_phase))
        init_phase = 2 * np.pi * rng.random(size=(n_bins, frames))
    arr = np.asarray(init_phase)
    if arr.shape != (n_bins, frames):
        raise ValueError(f"init_phase must have shape ({n_bins}, {frames}), got {arr.shape}")
    if not np.iscomplexobj(arr):
        arr = np.cos(arr) + 1j * np.sin(arr)
    arr = arr.astype(np.complex64)
    pair = np.ascontiguousarray(np.stack([arr.real.T, arr.imag.T], axis=-1)[None], dtype=np.float32)
    return torch.from_numpy(pair).to(device)


def _invert_mel(plan, mel, mean, std, inversion, nnls_steps):
    """(B, T, n_mels) normalised dB -> (B, T, F) what the reference calls linear_spec."""
    if inversion == "pinv":
        return plan.mel_invert(mel, float(mean), float(std))
    if inversion == "nnls":
        # the non-negative least-squares solution takes the place of the pinv product; like it,
        # it is in the power domain and handed to Griffin-Lim as it is
        power = torch.pow(10.0, 0.1 * (mel * float(std) + float(mean)))
        return plan.mel_invert_nnls(power, n_steps=nnls_steps, power=1.0)
    raise ValueError(f"inversion must be 'pinv' or 'nnls', got {inversion!r}")


def mel_to_audio_griffin_lim(mel_db, mean, std, sr=22050, n_fft=2048, hop_length=256, win_length=2048,
                             n_mels=80, fmax=8000, *, n_iter=60, momentum=0.99, init_phase=None,
                             inversion="pinv", nnls_steps=40, device=None, return_tensor=False,
                             verbose=False):
    """Normalised-dB mel (T, n_mels) -> peak-normalised waveform of hop_length*(T-1)
    samples, as reference vocoder/mel_audio.py:112-132 computes it:
    denormalise, dB->power, pinv(mel basis) @ mel, librosa.griffinlim(n_iter=60,
    momentum=0.99, random initial phase), divide by (max|y| + 1e-6).

    Keyword-only extras (defaults = reference behaviour):
      n_iter, momentum   Griffin-Lim parameters the reference leaves at 60 / 0.99.
      init_phase         None: fresh random phase per call (unseeded, like the
                         reference); int: ``np.random.default_rng(seed)`` phase;
                         (F, T) array of phasors or radians: that phase.
      inversion          "pinv" (reference: pinv(mel basis) @ mel power, may be negative) or
                         "nnls" (non-negative least squares, `nnls_steps` projected-gradient
                         steps from the clipped pinv product, as librosa's mel_to_stft starts).
      device             torch device (default: current ROCm device).
      return_tensor      return the device tensor instead of a numpy array.
      verbose            print what the reference prints (:130-131).
    """
    mel = mel_db.detach() if isinstance(mel_db, torch.Tensor) else torch.from_numpy(
        np.ascontiguousarray(mel_db, dtype=np.float32))
    if mel.dim() != 2 or mel.shape[1] != n_mels:
        raise ValueError(f"mel_db must be (T, {n_mels}), got {tuple(mel.shape)}")
    if momentum < 0:
        raise ValueError(f"griffinlim() called with momentum={momentum} < 0")
    frames = mel.shape[0]
    plan = get_plan(n_fft, hop_length, win_length, n_mels, frames, 1, device)
    plan.set_mel_basis(sr, 0.0, fmax)
    mel = mel.to(device=plan.device, dtype=torch.float32).contiguous().unsqueeze(0)
    if init_phase is None:
        seed = int(np.random.SeedSequence().generate_state(1, dtype=np.uint64)[0])
        phase0 = None
    else:
        seed = 0
        phase0 = _phase_tensor(init_phase, plan.n_bins, frames, plan.device)
    with plan.call_state(None, None):
        if inversion == "pinv":
            y = plan.mel_to_audio(mel, mean, std, [frames], phase0=phase0, seed=seed, n_iter=n_iter,
                                  momentum=momentum)
        else:
            S = _invert_mel(plan, mel, mean, std, inversion, nnls_steps)
            y = plan.griffinlim(S, [frames], phase0=phase0, seed=seed, n_iter=n_iter, momentum=momentum)
            plan.peak_normalize(y)
    audio = y[0]
    if verbose:
        host = audio.cpu().numpy()
        print(str(host))
        print(len(host) / 22050)  # librosa.get_duration(y=audio) assumes sr=22050
    if return_tensor:
        return audio
    host = audio.cpu().numpy()
    plan.status()  # the copy has waited for the stream: a void result is reported, not returned
    return host


def griffin_lim_batch(mel_db, mean, std, lengths=None, sr=22050, n_fft=2048, hop_length=256,
                      win_length=2048, n_mels=80, fmax=8000, *, n_iter=60, momentum=0.99, phase0=None,
                      seed=0, utt_ids=None, frames_per_run=None, inversion="pinv", nnls_steps=40,
                      device=None, return_sc=False, normalize=True):
    """Batched mel_to_audio_griffin_lim.

    ``mel_db``: (B, T, n_mels) array/tensor (rows past lengths[b] ignored) or a
    list of (T_b, n_mels) arrays.  Returns (audio, n_samples[, sc]) with audio a
    (B, hop*(T_max-1)) device tensor whose row b holds n_samples[b] =
    hop*(lengths[b]-1) samples followed by zeros.

    ``utt_ids``: one integer per utterance keying its device-drawn initial phase (``phase0`` None):
    utterance b draws the stream (seed, utt_ids[b]); default: its row.  With ids that travel with
    the utterances and a fixed ``frames_per_run`` (the number of frames one wavefront walks; the
    default None lets the library fit the run table to the whole batch) an utterance's waveform
    is bit-identical whatever batch, row or device it is synthesised in.
    """
    if isinstance(mel_db, (list, tuple)):
        lengths = [int(m.shape[0]) for m in mel_db]
        t_max = max(lengths)
        if all(isinstance(m, torch.Tensor) and m.is_cuda for m in mel_db):
            # mels produced on the device are padded there
            mel = torch.nn.utils.rnn.pad_sequence([m.detach().float() for m in mel_db], batch_first=True)
        else:
            packed = np.zeros((len(mel_db), t_max, n_mels), dtype=np.float32)
            for i, m in enumerate(mel_db):
                packed[i, : lengths[i]] = m.detach().cpu().numpy() if isinstance(m, torch.Tensor) else m
            mel = torch.from_numpy(packed)
    else:
        mel = mel_db if isinstance(mel_db, torch.Tensor) else torch.from_numpy(
            np.ascontiguousarray(mel_db, dtype=np.float32))
    if mel.dim() != 3 or mel.shape[2] != n_mels:
        raise ValueError(f"mel_db must be (B, T, {n_mels})")
    b, t_max, _ = mel.shape
    if lengths is None:
        lengths = [t_max] * b
    if utt_ids is not None and len(utt_ids) != b:
        raise ValueError(f"utt_ids must have one entry per utterance ({b}), got {len(utt_ids)}")
    plan = get_plan(n_fft, hop_length, win_length, n_mels, t_max, b, device)
    plan.set_mel_basis(sr, 0.0, fmax)
    mel = mel.to(device=plan.device, dtype=torch.float32).contiguous()
    n_samples = [hop_length * (int(n) - 1) for n in lengths]
    with plan.call_state(utt_ids, frames_per_run):
        if inversion == "pinv":
            res = plan.mel_to_audio(mel, mean, std, lengths, phase0=phase0, seed=seed, n_iter=n_iter,
                                    momentum=momentum, peak_eps=1e-6 if normalize else None,
                                    return_sc=return_sc)
            y, sc = res if return_sc else (res, None)
        else:
            S = _invert_mel(plan, mel, mean, std, inversion, nnls_steps)
            res = plan.griffinlim(S, lengths, phase0=phase0, seed=seed, n_iter=n_iter, momentum=momentum,
                                  return_sc=return_sc)
            y, sc = res if return_sc else (res, None)
            if normalize:
                plan.peak_normalize(y, n_samples)
    return (y, n_samples, sc) if return_sc else (y, n_samples)


def mel_to_audio(mel_spectrogram, sr=22050, n_fft=2048, hop_length=None, win_length=None, power=2.0,
                 n_iter=32, *, fmin=0.0, fmax=None, n_steps=40, momentum=0.99, init_phase=None,
                 device=None, return_tensor=False):
    """``librosa.feature.inverse.mel_to_audio(M, sr=sr)`` as the reference's older scripts call
    it (test_cnn_model.py:117-119, test_1.py:52-54, z_AR/test_AR_model.py:12-14): (n_mels, T)
    mel -> non-negative least-squares inversion -> ** (1/power) -> Griffin-Lim with librosa's
    defaults (n_fft 2048, hop n_fft/4, 32 iterations, fmax sr/2).  No peak normalisation.
    ``n_steps`` projected-gradient steps replace librosa's L-BFGS-B (same start, same
    objective)."""
    mel = mel_spectrogram.detach() if isinstance(mel_spectrogram, torch.Tensor) else torch.from_numpy(
        np.ascontiguousarray(mel_spectrogram, dtype=np.float32))
    if mel.dim() != 2:
        raise ValueError("mel_spectrogram must be (n_mels, T)")
    n_mels, frames = mel.shape
    win_length = n_fft if win_length is None else win_length
    hop_length = win_length // 4 if hop_length is None else hop_length
    plan = get_plan(n_fft, hop_length, win_length, n_mels, frames, 1, device)
    plan.set_mel_basis(sr, fmin, fmax)
    mel_tm = mel.t().to(device=plan.device, dtype=torch.float32).contiguous().unsqueeze(0)
    S = plan.mel_invert_nnls(mel_tm, n_steps=n_steps, power=power)
    if init_phase is None:
        seed, phase0 = int(np.random.SeedSequence().generate_state(1, dtype=np.uint64)[0]), None
    else:
        seed, phase0 = 0, _phase_tensor(init_phase, plan.n_bins, frames, plan.device)
    with plan.call_state(None, None):
        y = plan.griffinlim(S, [frames], phase0=phase0, seed=seed, n_iter=n_iter, momentum=momentum)[0]
    if return_tensor:
        return y
    host = y.cpu().numpy()
    plan.status()
    return host


def audio_to_mel_db(y, sr=22050, n_fft=2048, hop_length=256, win_length=2048, n_mels=80, fmax=8000, *,
                    device=None, return_tensor=False):
    """The reference's mel extraction (acoustic/dataset.py:75-84):
    ``librosa.power_to_db(librosa.feature.melspectrogram(y=y, sr=sr, n_mels=80, fmax=8000,
    n_fft=2048, hop_length=256, win_length=2048), ref=np.max).T`` -> (T, n_mels) dB in
    [-80, 0], T = 1 + len(y) // hop_length frames over the whole signal, as librosa frames it."""
    sig = y.detach() if isinstance(y, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(y, dtype=np.float32))
    if sig.dim() != 1:
        raise ValueError("y must be 1-D")
    frames = 1 + sig.shape[0] // hop_length
    if frames < 2:
        raise ValueError("signal shorter than one hop")
    plan = get_plan(n_fft, hop_length, win_length, n_mels, frames, 1, device)
    plan.set_mel_basis(sr, 0.0, fmax)
    sig = sig.to(device=plan.device, dtype=torch.float32).contiguous().unsqueeze(0)
    mel = plan.melspectrogram(sig)
    db = plan.power_to_db(mel)[0]
    return db if return_tensor else db.cpu().numpy()
