"""CPU oracle for the mel -> waveform Griffin-Lim path.  TEST INFRASTRUCTURE ONLY.

This module is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  Nothing under ``tts_amd/`` imports it and the product path has no
CPU fallback.

What it restates
----------------
The pipeline of the reference function ``mel_to_audio_griffin_lim``
(reference ``vocoder/mel_audio.py:112-132``; identical copies at
``vocoder/pra.py:65-80`` and ``test.py:143-158``):

    denormalise -> librosa.db_to_power -> librosa.filters.mel -> np.linalg.pinv
    -> np.dot -> librosa.griffinlim(n_iter=60) -> peak normalise

The arithmetic of that function lives in third-party packages that are NOT
vendored in the reference and are NOT installed here: ``librosa==0.11.0``
(reference ``requirment1.txt:47``), driven through ``numpy==2.1.3`` (``:64``),
``scipy==1.15.2`` (``:93``) and ``numba==0.61.2`` (``:63``).  The functions below
restate the published librosa 0.11.0 algorithms (``db_to_power``, ``filters.mel``,
``stft``, ``istft``, ``window_sumsquare``, ``griffinlim``,
``feature.inverse.mel_to_stft`` / ``util.nnls``) including their dtype
promotions, on top of numpy and scipy.fft (pocketfft, the FFT librosa itself
calls).

Pinning status
--------------
* Pinned by the only input->output pair the reference ships
  (``vocoder/audio/LJ001-0001.npy`` + ``mel_mean_std.npy`` ->
  ``vocoder/audio/reconstructed_griffin.wav``, ``vocoder/reconstructed_griffin.wav``):
  output length, denormalised dB range, peak normalisation, and the spectral
  convergence of the reference's own waveforms against this module's
  ``|linear_spec|`` (``tests/test_oracle_known_answer.py``).
* The reference draws its initial phase from an unseeded generator, so
  element-wise waveform parity with the reference wavs is impossible; and
  librosa cannot be imported here, so element-wise parity with librosa itself
  is **parity unpinned**.
"""
from __future__ import annotations

import numpy as np
import scipy.fft as _fft

__all__ = [
    "hz_to_mel", "mel_to_hz", "mel_filterbank", "db_to_power",
    "hann_window", "pad_center", "window_sumsquare", "stft", "istft",
    "spectral_convergence", "griffinlim", "peak_normalise", "mel_invert",
    "mel_to_audio_griffin_lim", "nnls_init", "nnls_projected_gradient",
    "mel_to_stft", "mel_to_audio", "melspectrogram", "power_to_db",
]


# ----------------------------------------------------------------------------
# librosa.filters.mel  (Slaney scale, Slaney area normalisation)
# ----------------------------------------------------------------------------
def hz_to_mel(freq):
    """librosa.hz_to_mel(htk=False): linear below 1 kHz, log above."""
    freq = np.asanyarray(freq, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = freq / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    if freq.ndim:
        log_t = freq >= min_log_hz
        mels[log_t] = min_log_mel + np.log(freq[log_t] / min_log_hz) / logstep
    elif freq >= min_log_hz:
        mels = min_log_mel + np.log(freq / min_log_hz) / logstep
    return mels


def mel_to_hz(mels):
    """librosa.mel_to_hz(htk=False)."""
    mels = np.asanyarray(mels, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * mels
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    if mels.ndim:
        log_t = mels >= min_log_mel
        freqs[log_t] = min_log_hz * np.exp(logstep * (mels[log_t] - min_log_mel))
    elif mels >= min_log_mel:
        freqs = min_log_hz * np.exp(logstep * (mels - min_log_mel))
    return freqs


def mel_filterbank(sr, n_fft, n_mels=128, fmin=0.0, fmax=None, dtype=np.float32):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax, htk=False, norm='slaney').

    Called at reference vocoder/mel_audio.py:121 with
    (sr=22050, n_fft=2048, n_mels=80, fmax=8000).  Returns (n_mels, 1+n_fft//2).
    """
    if fmax is None:
        fmax = float(sr) / 2
    n_mels = int(n_mels)
    weights = np.zeros((n_mels, int(1 + n_fft // 2)), dtype=dtype)
    fftfreqs = np.fft.rfftfreq(n=n_fft, d=1.0 / sr)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2 : n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]
    return weights


# ----------------------------------------------------------------------------
# elementwise / linear-algebra stages of mel_audio.py:114-123
# ----------------------------------------------------------------------------
def db_to_power(s_db, ref=1.0):
    """librosa.db_to_power: ref * 10**(0.1*S_db); float32 in -> float32 out."""
    return ref * np.power(10.0, 0.1 * s_db)


def mel_invert(mel_db, mean, std, sr=22050, n_fft=2048, n_mels=80, fmax=8000):
    """Reference vocoder/mel_audio.py:114-123: (T, n_mels) normalised dB -> (F, T)."""
    mel_db = np.asarray(mel_db)
    log_mel = (mel_db * std) + mean
    mel_spec = db_to_power(log_mel.T)
    mel_basis = mel_filterbank(sr=sr, n_fft=n_fft, n_mels=n_mels, fmax=fmax)
    inv_mel_basis = np.linalg.pinv(mel_basis)
    return np.dot(inv_mel_basis, mel_spec)


# ----------------------------------------------------------------------------
# librosa.stft / istft / window_sumsquare
# ----------------------------------------------------------------------------
def hann_window(win_length):
    """scipy.signal.get_window('hann', win_length, fftbins=True): periodic, float64."""
    n = np.arange(win_length, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_length)


def pad_center(w, size):
    """librosa.util.pad_center for a 1-D window."""
    n = w.shape[-1]
    lpad = int((size - n) // 2)
    return np.pad(w, (lpad, int(size - n - lpad)), mode="constant")


def _fft_window(n_fft, win_length):
    return pad_center(hann_window(win_length), n_fft)


def window_sumsquare(n_frames, hop_length, win_length, n_fft, dtype=np.float32):
    """librosa.filters.window_sumsquare(window='hann', norm=None).

    A dtype-typed buffer receives the float64 squared window once per frame, in
    frame order, so every addition rounds to ``dtype``.
    """
    n = n_fft + hop_length * (n_frames - 1)
    x = np.zeros(n, dtype=dtype)
    win_sq = pad_center(hann_window(win_length) ** 2, n_fft)
    for i in range(n_frames):
        sample = i * hop_length
        x[sample : min(n, sample + n_fft)] += win_sq[: max(0, min(n_fft, n - sample))]
    return x


def stft(y, n_fft=2048, hop_length=None, win_length=None, out_dtype=None):
    """librosa.stft(window='hann', center=True, pad_mode='constant').

    float32 signal x float64 window -> float64 frames -> double-precision rfft ->
    rounded to complex64 on store (the dtype librosa allocates for a float32
    signal).  Returns (1+n_fft//2, 1+len(y)//hop).
    """
    if win_length is None:
        win_length = n_fft
    if hop_length is None:
        hop_length = win_length // 4
    y = np.asarray(y)
    if out_dtype is None:
        out_dtype = np.complex64 if y.dtype == np.float32 else np.complex128
    w = _fft_window(n_fft, win_length)
    yp = np.pad(y, (n_fft // 2, n_fft // 2), mode="constant")
    n_frames = 1 + (len(yp) - n_fft) // hop_length
    frames = np.lib.stride_tricks.as_strided(
        yp, shape=(n_frames, n_fft), strides=(yp.strides[0] * hop_length, yp.strides[0]),
        writeable=False)
    spec = _fft.rfft(w[np.newaxis, :] * frames, axis=-1)
    return np.ascontiguousarray(spec.T).astype(out_dtype)


def istft(d, hop_length=None, win_length=None, n_fft=None):
    """librosa.istft(window='hann', center=True, length=None).

    irfft keeps the single precision of a complex64 input; the float64 window
    promotes each frame to float64; frames are overlap-added in ascending frame
    order into a float32 buffer; samples whose window-sum-square exceeds
    ``tiny`` are divided by it.  Returns hop*(T-1) samples.
    """
    d = np.asarray(d)
    if n_fft is None:
        n_fft = 2 * (d.shape[-2] - 1)
    if win_length is None:
        win_length = n_fft
    if hop_length is None:
        hop_length = win_length // 4
    dtype = np.float32 if d.dtype == np.complex64 else np.float64
    n_frames = d.shape[-1]
    w = _fft_window(n_fft, win_length)
    ytmp = w[:, np.newaxis] * _fft.irfft(d, n=n_fft, axis=-2)
    full = np.zeros(n_fft + hop_length * (n_frames - 1), dtype=dtype)
    for t in range(n_frames):
        full[t * hop_length : t * hop_length + n_fft] += ytmp[:, t]
    length = hop_length * (n_frames - 1)
    start = n_fft // 2
    y = full[start : start + length].copy()
    wss = window_sumsquare(n_frames, hop_length, win_length, n_fft, dtype=dtype)
    wss = wss[start : start + length]
    nz = wss > np.finfo(dtype).tiny
    y[nz] /= wss[nz]
    return y


# ----------------------------------------------------------------------------
# librosa.griffinlim
# ----------------------------------------------------------------------------
def spectral_convergence(rebuilt, s):
    """|| |rebuilt| - |S| ||_F / ||S||_F  (cf. reference TTS/losses.py:16-19)."""
    num = np.linalg.norm(np.abs(rebuilt).astype(np.float64) - np.abs(s).astype(np.float64))
    return float(num / np.linalg.norm(np.asarray(s, dtype=np.float64)))


def griffinlim(s, n_iter=32, hop_length=None, win_length=None, n_fft=None,
               momentum=0.99, init_phase=None, random_state=None,
               return_sc=False, snapshots=None):
    """librosa.griffinlim(window='hann', center=True, pad_mode='constant').

    Called at reference vocoder/mel_audio.py:126 with n_iter=60 and the
    hop/win of the caller; n_fft is inferred from S.

    ``init_phase``: None -> librosa's init='random' from ``random_state``
    (None = unseeded, as the reference runs it); otherwise an (F, T) array of
    unit phasors (complex) or of angles in radians (real).
    ``snapshots``: optional iterable of iteration counts k; the waveform
    istft(angles_k) after k updates is returned for each (k = n_iter is the
    ordinary result).
    """
    s = np.asarray(s)
    if momentum < 0:
        raise ValueError(f"griffinlim() called with momentum={momentum} < 0")
    if n_fft is None:
        n_fft = 2 * (s.shape[-2] - 1)
    cdtype = np.complex64 if s.dtype == np.float32 else np.complex128
    angles = np.empty(s.shape, dtype=cdtype)
    eps = np.finfo(angles.real.dtype).tiny
    if init_phase is None:
        rng = np.random.default_rng(random_state)
        phi = 2 * np.pi * rng.random(size=s.shape)
        angles[:] = np.cos(phi) + 1j * np.sin(phi)
    else:
        init_phase = np.asarray(init_phase)
        if np.iscomplexobj(init_phase):
            angles[:] = init_phase
        else:
            angles[:] = np.cos(init_phase) + 1j * np.sin(init_phase)
    angles *= s
    tprev = None
    sc = []
    snaps = {}
    want = set(snapshots) if snapshots is not None else set()
    kw = dict(hop_length=hop_length, win_length=win_length, n_fft=n_fft)
    for it in range(n_iter):
        inverse = istft(angles, **kw)
        if it in want:
            snaps[it] = inverse
        rebuilt = stft(inverse, n_fft=n_fft, hop_length=hop_length, win_length=win_length,
                       out_dtype=cdtype)
        if return_sc:
            sc.append(spectral_convergence(rebuilt, s))
        angles[:] = rebuilt
        if tprev is not None:
            angles -= (momentum / (1 + momentum)) * tprev
        angles /= np.abs(angles) + eps
        angles *= s
        tprev = rebuilt
    y = istft(angles, **kw)
    if n_iter in want:
        snaps[n_iter] = y
    out = (y,)
    if return_sc:
        out += (np.asarray(sc, dtype=np.float64),)
    if snapshots is not None:
        out += (snaps,)
    return out[0] if len(out) == 1 else out


def peak_normalise(audio):
    """Reference vocoder/mel_audio.py:129."""
    return audio / (np.max(np.abs(audio)) + 1e-6)


def mel_to_audio_griffin_lim(mel_db, mean, std, sr=22050, n_fft=2048, hop_length=256,
                             win_length=2048, n_mels=80, fmax=8000, *, n_iter=60,
                             momentum=0.99, init_phase=None, random_state=None):
    """Reference vocoder/mel_audio.py:112-132 (the prints at :130-131 dropped)."""
    linear_spec = mel_invert(mel_db, mean, std, sr=sr, n_fft=n_fft, n_mels=n_mels, fmax=fmax)
    audio = griffinlim(linear_spec, n_iter=n_iter, hop_length=hop_length,
                       win_length=win_length, momentum=momentum, init_phase=init_phase,
                       random_state=random_state)
    return peak_normalise(audio)


# ----------------------------------------------------------------------------
# secondary mode: librosa.feature.inverse.mel_to_audio (NNLS inversion)
#   reference call sites: test_cnn_model.py:117-119, test_1.py:52-54,
#   z_AR/test_AR_model.py:12-14
# ----------------------------------------------------------------------------
def nnls_init(a, b):
    """The deterministic start of librosa.util.nnls: clip(pinv(A) @ B, 0, None)."""
    x = np.einsum("fm,...mt->...ft", np.linalg.pinv(a), b, optimize=True)
    np.clip(x, 0, None, out=x)
    return x


def nnls_projected_gradient(a, b, n_steps=200, x0=None, tol=0.0):
    """Projected gradient on 0.5*||A X - B||^2 s.t. X >= 0, step 1/||A||_2^2.  With ``tol`` a
    frame (column) stops moving once ||A x - b|| <= tol * ||b||.

    librosa.util.nnls minimises the same objective with L-BFGS-B from the same
    start; the problem is under-determined (F >> n_mels), so the minimiser is
    not unique and only the mel-domain residual is comparable between solvers.
    """
    a = np.asarray(a)
    b = np.asarray(b)
    x = nnls_init(a, b) if x0 is None else np.array(x0, copy=True)
    step = a.dtype.type(1.0 / (np.linalg.norm(a.astype(np.float64), 2) ** 2))
    for _ in range(n_steps):
        r = a @ x - b
        if tol > 0:
            active = np.linalg.norm(r, axis=0) > tol * np.linalg.norm(b, axis=0)
            if not active.any():
                break
            r = r * active
        x = np.maximum(x - step * (a.T @ r), 0)
    return x


def nnls_accelerated(a, b, n_steps=40, x0=None, tol=0.0):
    """The same projected-gradient step with Nesterov / FISTA extrapolation (Beck & Teboulle 2009):
        x_{k+1} = max(y_k - step * A^T (A y_k - B), 0)
        t_{k+1} = (1 + sqrt(1 + 4 t_k^2)) / 2,   beta_k = (t_k - 1) / t_{k+1}   (t_0 = 1, in double)
        y_{k+1} = x_{k+1} + beta_k (x_{k+1} - x_k)
    Same objective, same start, same step length as nnls_projected_gradient; it reaches that
    solver's 200-step residual in about 35 steps.  With ``tol`` a frame (column) stops for good
    once ||A y - b|| <= tol * ||b||: it takes the projection max(y, 0) of the point that passed
    the test and neither moves nor extrapolates again, so a frame's result does not depend on
    the other frames; the loop ends when no frame moves.  Returns x."""
    a = np.asarray(a)
    b = np.asarray(b)
    x = nnls_init(a, b) if x0 is None else np.array(x0, copy=True)
    y = x.copy()
    step = a.dtype.type(1.0 / (np.linalg.norm(a.astype(np.float64), 2) ** 2))
    t = 1.0
    stopped = np.zeros(b.shape[-1], dtype=bool)
    for _ in range(n_steps):
        r = a @ y - b
        if tol > 0:
            stopped |= np.linalg.norm(r, axis=0) <= tol * np.linalg.norm(b, axis=0)
        active = (~stopped).astype(a.dtype)
        x_new = np.maximum(y - step * (a.T @ (r * active)), 0)
        t_next = 0.5 * (1.0 + np.sqrt(1.0 + 4.0 * t * t))
        beta = a.dtype.type((t - 1.0) / t_next)
        t = t_next
        y = x_new + (beta * active) * (x_new - x)
        x = x_new
        if stopped.all():
            break
    return x


def mel_to_stft(m, sr=22050, n_fft=2048, power=2.0, n_steps=40, accelerated=True, **kw):
    """librosa.feature.inverse.mel_to_stft with the projected-gradient solver (accelerated by
    default, as the device solver is)."""
    m = np.asarray(m)
    a = mel_filterbank(sr=sr, n_fft=n_fft, n_mels=m.shape[-2], dtype=m.dtype, **kw)
    x = nnls_accelerated(a, m, n_steps=n_steps) if accelerated else nnls_projected_gradient(a, m, n_steps=n_steps)
    return np.power(x, 1.0 / power, out=x)


def mel_to_audio(m, sr=22050, n_fft=2048, hop_length=None, win_length=None, power=2.0,
                 n_iter=32, init_phase=None, random_state=None, n_steps=40, **kw):
    """librosa.feature.inverse.mel_to_audio defaults (hop = n_fft//4 = 512)."""
    s = mel_to_stft(m, sr=sr, n_fft=n_fft, power=power, n_steps=n_steps, **kw)
    return griffinlim(s, n_iter=n_iter, hop_length=hop_length, win_length=win_length,
                      n_fft=n_fft, init_phase=init_phase, random_state=random_state)


# ----------------------------------------------------------------------------
# forward direction: the reference's mel extraction, acoustic/dataset.py:75-84
# ----------------------------------------------------------------------------
def melspectrogram(y, sr=22050, n_fft=2048, hop_length=512, win_length=None, power=2.0, n_mels=128,
                   fmin=0.0, fmax=None):
    """librosa.feature.melspectrogram(y=...): |stft|**power contracted with the mel basis."""
    d = stft(y, n_fft=n_fft, hop_length=hop_length, win_length=win_length)
    s = np.abs(d) ** power
    basis = mel_filterbank(sr=sr, n_fft=n_fft, n_mels=n_mels, fmin=fmin, fmax=fmax)
    return np.einsum("ft,mf->mt", s, basis, optimize=True)


def power_to_db(s, ref=1.0, amin=1e-10, top_db=80.0):
    """librosa.power_to_db."""
    s = np.asarray(s)
    magnitude = s
    ref_value = ref(magnitude) if callable(ref) else np.abs(ref)
    log_spec = 10.0 * np.log10(np.maximum(amin, magnitude))
    log_spec -= 10.0 * np.log10(np.maximum(amin, ref_value))
    if top_db is not None:
        log_spec = np.maximum(log_spec, log_spec.max() - top_db)
    return log_spec
