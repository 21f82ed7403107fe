"""Shared test utilities: synthetic LJSpeech-shaped mels and layout conversions."""
import numpy as np

MEAN, STD = np.float32(-61.55587), np.float32(20.378538)


def synthetic_mel(seed, frames, n_mels=80):
    """Smoothed-noise dB mel in [-80, 0] with its maximum at 0 dB, normalised with the
    reference's dataset constants; float32 (frames, n_mels)."""
    rng = np.random.default_rng(seed)
    g = rng.standard_normal((frames + 4, n_mels + 4))
    c = np.cumsum(np.cumsum(g, axis=0), axis=1)
    box = (c[5:, 5:] - c[:-5, 5:] - c[5:, :-5] + c[:-5, :-5]) / 25.0
    box = np.pad(box, ((0, frames - box.shape[0]), (0, n_mels - box.shape[1])), mode="edge")
    box = box / box.std()
    d = np.clip(-35.0 + 18.0 * box, -80.0, 0.0)
    d = np.clip(d - d.max(), -80.0, 0.0)
    return ((d - MEAN) / STD).astype(np.float32)


def phasors(seed, n_bins, frames):
    """(F, T) complex64 unit phasors, librosa init='random' with default_rng(seed)."""
    rng = np.random.default_rng(seed)
    return np.exp(2j * np.pi * rng.random((n_bins, frames))).astype(np.complex64)


def to_btf(x_ft):
    """(F, T) real -> (1, T, F) contiguous float32."""
    return np.ascontiguousarray(np.asarray(x_ft).T[None], dtype=np.float32)


def to_btf2(x_ft):
    """(F, T) complex -> (1, T, F, 2) contiguous float32."""
    x = np.asarray(x_ft).T
    return np.ascontiguousarray(np.stack([x.real, x.imag], axis=-1)[None], dtype=np.float32)


def from_btf2(x):
    """(T, F, 2) float -> (F, T) complex64."""
    x = np.asarray(x)
    return (x[..., 0] + 1j * x[..., 1]).T.astype(np.complex64)
