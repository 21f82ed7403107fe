"""Host-side, one-time tables of the vocoder: the Slaney mel filterbank and its
pseudo-inverse.

The reference builds both on the host for every call
(vocoder/mel_audio.py:121-122: ``librosa.filters.mel`` then ``np.linalg.pinv``);
here they are built once per (sr, n_fft, n_mels, fmin, fmax) and cached, the
pseudo-inverse is uploaded to the device plan.
"""
from __future__ import annotations

import functools

import numpy as np

_F_SP = 200.0 / 3          # Hz per mel below the knee
_KNEE_HZ = 1000.0
_KNEE_MEL = _KNEE_HZ / _F_SP
_LOGSTEP = np.log(6.4) / 27.0


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    lin = f / _F_SP
    with np.errstate(divide="ignore", invalid="ignore"):
        log = _KNEE_MEL + np.log(np.maximum(f, 1e-300) / _KNEE_HZ) / _LOGSTEP
    return np.where(f >= _KNEE_HZ, log, lin)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    return np.where(m >= _KNEE_MEL, _KNEE_HZ * np.exp(_LOGSTEP * (m - _KNEE_MEL)), _F_SP * m)


def mel_filterbank(sr, n_fft, n_mels=128, fmin=0.0, fmax=None, dtype=np.float32):
    """Slaney-scale triangular filterbank with Slaney area normalisation,
    the matrix ``librosa.filters.mel(sr=, n_fft=, n_mels=, fmin=, fmax=)`` returns
    with its defaults ``htk=False, norm='slaney'``.  Shape (n_mels, 1 + n_fft//2).
    """
    if fmax is None:
        fmax = sr / 2.0
    freqs = np.fft.rfftfreq(n_fft, 1.0 / sr)
    edges = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    width = np.diff(edges)
    ramps = edges[:, None] - freqs[None, :]
    rising = -ramps[:-2] / width[:-1, None]
    falling = ramps[2:] / width[1:, None]
    # the triangles are clipped in float64 and stored in `dtype` before the area
    # normalisation is applied in `dtype`, as librosa does
    tri = np.maximum(0.0, np.minimum(rising, falling)).astype(dtype)
    tri *= (2.0 / (edges[2:] - edges[:-2]))[:, None]
    return tri


@functools.lru_cache(maxsize=32)
def inverse_mel_basis(sr, n_fft, n_mels, fmin, fmax):
    """``np.linalg.pinv(mel_filterbank(...))`` as float32 (F, n_mels), C-contiguous."""
    basis = mel_filterbank(sr, n_fft, n_mels=n_mels, fmin=fmin, fmax=fmax)
    inv = np.ascontiguousarray(np.linalg.pinv(basis), dtype=np.float32)
    inv.setflags(write=False)
    return inv


@functools.lru_cache(maxsize=32)
def mel_basis_and_step(sr, n_fft, n_mels, fmin, fmax):
    """(basis float32 (n_mels, F), step) for the projected-gradient non-negative inversion:
    step = 1 / ||basis||_2^2, the reciprocal Lipschitz constant of the gradient of
    0.5 * ||basis @ X - mel||^2."""
    basis = np.ascontiguousarray(mel_filterbank(sr, n_fft, n_mels=n_mels, fmin=fmin, fmax=fmax))
    step = float(1.0 / np.linalg.norm(basis.astype(np.float64), 2) ** 2)
    basis.setflags(write=False)
    return basis, step
