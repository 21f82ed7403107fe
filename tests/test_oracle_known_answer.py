"""The oracle against everything the reference ships for this path, plus its own
regression anchors.  No GPU involved.

Reference-anchored (KAT-1): the reference's only input/output pair
(vocoder/audio/LJ001-0001.npy + mel_mean_std.npy -> the two reconstructed_griffin.wav
files), repacked under tests/golden by make_golden.py.  The reference draws an unseeded
random phase, so only phase-free quantities can be compared: length, ranges, peak
normalisation, and how well a waveform's STFT magnitude matches |pinv(mel basis) @ mel|.
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, rel_l2
from oracle import griffinlim_oracle as O


@pytest.fixture(scope="module")
def kat():
    with open(os.path.join(GOLDEN, "kat.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="module")
def linear_spec(lj001):
    mel, mean, std = lj001
    return O.mel_invert(mel, mean, std)


def best_gain_sc(y, target_abs):
    d = np.abs(O.stft(y, n_fft=2048, hop_length=256, win_length=2048)).astype(np.float64)
    g = (d * target_abs).sum() / (d * d).sum()
    return float(np.linalg.norm(g * d - target_abs) / np.linalg.norm(target_abs))


def test_fixture_shapes_and_ranges(lj001, linear_spec, kat):
    mel, mean, std = lj001
    assert mel.shape == (865, 80) and mel.dtype == np.float32
    assert np.all(mel[793:] == 0.0)                      # zero padding of the acoustic model
    assert float(mean) == pytest.approx(-61.55587) and float(std) == pytest.approx(20.378538)
    log_mel = mel * std + mean
    assert log_mel.min() == pytest.approx(-80.0, abs=1e-4) and log_mel.max() == pytest.approx(0.0, abs=1e-4)
    power = O.db_to_power(log_mel)
    assert power.dtype == np.float32
    assert power.min() == pytest.approx(1e-8, rel=1e-4) and power.max() == pytest.approx(1.0, rel=1e-5)
    s = linear_spec
    assert s.shape == (1025, 865) and s.dtype == np.float32
    assert s.min() == pytest.approx(kat["linear_spec"]["min"], rel=1e-5)
    assert s.max() == pytest.approx(kat["linear_spec"]["max"], rel=1e-5)
    assert (s < 0).mean() == pytest.approx(0.0687, abs=5e-4)   # signed "magnitudes" by design


def test_reference_wavs_fit_the_oracle_spectrogram(linear_spec, kat):
    """The reference's own outputs, analysed with the oracle's STFT, match the oracle's
    linear spectrogram as well as Griffin-Lim can make them (0.1738 / 0.1709)."""
    wavs = np.load(os.path.join(GOLDEN, "ref_wavs.npz"))
    a = np.abs(linear_spec).astype(np.float64)
    for key in ("vocoder_audio", "vocoder"):
        w = wavs[key]
        assert w.shape == (221184,) and w.dtype == np.int16          # hop * (T - 1)
        assert np.abs(w.astype(np.int32)).max() >= 32767              # peak-normalised, PCM16
        sc = best_gain_sc(w.astype(np.float32) / 32768.0, a)
        assert sc == pytest.approx(kat["ref_wav_best_gain_sc"][key], abs=1e-6)
        assert 0.165 < sc < 0.180


def test_oracle_output_sits_in_the_reference_quality_window(lj001, linear_spec):
    mel, mean, std = lj001
    y = O.mel_to_audio_griffin_lim(mel, mean, std, random_state=4)
    assert y.shape == (221184,) and y.dtype == np.float32
    peak = float(np.abs(y).max())
    assert 0.9999 < peak <= 1.0
    sc = best_gain_sc(y, np.abs(linear_spec).astype(np.float64))
    assert 0.165 < sc < 0.180


@pytest.mark.parametrize("name,n_fft,frames", [("n1024_t430", 1024, 430), ("n2048_t865", 2048, 865)])
def test_spectral_convergence_curve(lj001, kat, name, n_fft, frames):
    mel, mean, std = lj001
    s = O.mel_invert(mel[:frames], mean, std, n_fft=n_fft)
    ph = np.exp(2j * np.pi * np.random.default_rng(0).random(s.shape))
    _, sc = O.griffinlim(s, n_iter=60, hop_length=256, win_length=n_fft, init_phase=ph, return_sc=True)
    want = kat["sc_curve_rng0"][name]
    got = [sc[k - 1] for k in want["k"]]
    assert np.allclose(got, want["sc"], rtol=1e-6)
    # the same curve to four decimals, as first recorded for this fixture and seed
    quoted = {"n2048_t865": [0.7078, 0.3730, 0.2519, 0.1917, 0.1798, 0.1733, 0.1739],
              "n1024_t430": [0.5383, 0.3354, 0.2662, 0.2215, 0.1932, 0.1788, 0.1787]}[name]
    assert np.allclose(got, quoted, atol=6e-5)


@pytest.mark.parametrize("name", ["stages_n1024_t16", "stages_n2048_t12"])
def test_stage_fixtures_reproduce(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    n_fft, hop = int(g["n_fft"]), int(g["hop"])
    s = O.mel_invert(g["mel_db"], g["mean"], g["std"], n_fft=n_fft)
    assert np.array_equal(s, g["linear_spec"])
    y, sc, snaps = O.griffinlim(s, n_iter=8, hop_length=hop, win_length=n_fft, init_phase=g["phase0"],
                                return_sc=True, snapshots=[0, 1, 2, 4, 8])
    for k in (0, 1, 2, 4, 8):
        assert rel_l2(snaps[k], g[f"y_k{k}"]) < 1e-6
    assert np.allclose(sc, g["sc"], rtol=1e-6)
    assert rel_l2(O.peak_normalise(y), g["audio"]) < 1e-6
    assert rel_l2(O.stft(snaps[0], n_fft=n_fft, hop_length=hop, win_length=n_fft), g["rebuilt_k1"]) < 1e-6


def test_mel_basis_properties():
    b = O.mel_filterbank(22050, 2048, n_mels=80, fmax=8000)
    assert b.shape == (80, 1025) and b.dtype == np.float32
    assert np.linalg.matrix_rank(b) == 80
    assert int((b.sum(axis=0) == 0).sum()) == 282          # bins above fmax (and DC)
    assert np.linalg.cond(b[:, b.sum(axis=0) > 0].astype(np.float64)) == pytest.approx(4.48, abs=0.02)
    b2 = O.mel_filterbank(22050, 1024, n_mels=80, fmax=8000)
    assert b2.shape == (80, 513) and int((b2.sum(axis=0) == 0).sum()) == 142
    b3 = O.mel_filterbank(44100, 2048, n_mels=128)
    assert b3.shape == (128, 1025) and np.linalg.matrix_rank(b3) == 128


def test_window_sumsquare_and_round_trip():
    wss = O.window_sumsquare(40, 256, 1024, 1024)
    assert wss.dtype == np.float32
    assert np.allclose(wss[1024:-1024], 1.5, atol=1e-6)    # Hann, 75 % overlap
    wss8 = O.window_sumsquare(40, 256, 2048, 2048)
    assert np.allclose(wss8[2048:-2048], 3.0, atol=1e-6)
    y = np.random.default_rng(2).standard_normal(256 * 39).astype(np.float32)
    d = O.stft(y, n_fft=1024, hop_length=256)
    assert d.shape == (513, 40) and d.dtype == np.complex64
    back = O.istft(d, hop_length=256)
    assert back.shape == y.shape and back.dtype == np.float32
    assert rel_l2(back, y) < 1e-6


def test_griffinlim_contract():
    s = (np.random.default_rng(3).standard_normal((513, 12)) ** 2).astype(np.float32)
    with pytest.raises(ValueError):
        O.griffinlim(s, n_iter=2, hop_length=256, momentum=-0.1)
    y = O.griffinlim(s, n_iter=3, hop_length=256, random_state=1)
    assert y.shape == (256 * 11,) and y.dtype == np.float32
    # n_fft is inferred from the row count, as in the reference call (vocoder/mel_audio.py:126)
    y2 = O.griffinlim(s, n_iter=3, hop_length=256, win_length=1024, n_fft=1024, random_state=1)
    assert np.array_equal(y, y2)
    assert np.array_equal(O.peak_normalise(y), y / (np.max(np.abs(y)) + 1e-6))
