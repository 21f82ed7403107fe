"""Host-side pieces of the product that need no GPU: the mel basis tables, argument
checking, and the rule that the product never touches the oracle or a CPU fallback."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from oracle import griffinlim_oracle as O


@pytest.mark.parametrize("sr,n_fft,n_mels,fmax", [(22050, 2048, 80, 8000), (22050, 1024, 80, 8000),
                                                  (44100, 2048, 128, None), (16000, 1024, 40, 7600.0)])
def test_mel_filterbank_equals_oracle(sr, n_fft, n_mels, fmax):
    from tts_amd.filters import inverse_mel_basis, mel_filterbank
    got = mel_filterbank(sr, n_fft, n_mels=n_mels, fmax=fmax)
    want = O.mel_filterbank(sr, n_fft, n_mels=n_mels, fmax=fmax)
    assert got.dtype == np.float32 and got.shape == want.shape
    assert np.array_equal(got, want)
    inv = inverse_mel_basis(sr, n_fft, n_mels, 0.0, fmax)
    assert inv.shape == (n_fft // 2 + 1, n_mels) and inv.dtype == np.float32 and inv.flags.c_contiguous
    assert np.array_equal(inv, np.linalg.pinv(want))
    # pinv is a right inverse of the (full row rank) basis
    assert np.allclose(want.astype(np.float64) @ inv.astype(np.float64), np.eye(n_mels), atol=1e-4)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "tts_amd")
    for dirpath, _, files in os.walk(pkg):
        for name in files:
            if name.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, name)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), name
                assert "griffinlim_oracle" not in text, name


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import tts_amd
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        tts_amd.GriffinLimPlan(1024, 256, 1024, 80, 16, 1)
    with pytest.raises(RuntimeError):
        tts_amd.mel_to_audio_griffin_lim(np.zeros((10, 80), np.float32), -61.5, 20.4)


def test_argument_checks_come_before_the_device():
    import tts_amd
    with pytest.raises(ValueError, match=r"\(T, 80\)"):
        tts_amd.mel_to_audio_griffin_lim(np.zeros((10, 40), np.float32), 0.0, 1.0)
    with pytest.raises(ValueError, match="momentum"):
        tts_amd.mel_to_audio_griffin_lim(np.zeros((10, 80), np.float32), 0.0, 1.0, momentum=-1.0)


def test_synthetic_mel_is_ljspeech_shaped():
    from helpers import MEAN, STD, synthetic_mel
    m = synthetic_mel(1234, 430)
    assert m.shape == (430, 80) and m.dtype == np.float32
    db = m * STD + MEAN
    assert db.max() == pytest.approx(0.0, abs=1e-3) and db.min() >= -80.001
    assert np.array_equal(m, synthetic_mel(1234, 430))


def test_wav_writer_matches_the_reference_files(tmp_path):
    """PCM16 conversion: the reference's wavs hold +32767 / -32768 peaks for audio
    normalised to just under 1.0, i.e. scale 32768, round, clip."""
    from tts_amd.wavio import float_to_pcm16, read_wav, write_wav
    x = np.array([0.0, 0.5, -0.5, 0.99999, -0.99999, 1.0, -1.0, 1.5, -1.5, 1e-5], np.float32)
    pcm = float_to_pcm16(x)
    assert pcm.tolist() == [0, 16384, -16384, 32767, -32768, 32767, -32768, 32767, -32768, 0]
    path = tmp_path / "t.wav"
    y = np.sin(np.linspace(0, 100, 5000)).astype(np.float32) * 0.7
    write_wav(path, y, 22050)
    back, sr = read_wav(path)
    assert sr == 22050 and back.shape == y.shape
    assert np.max(np.abs(back - y)) <= 0.5 / 32768 + 1e-7
    from scipy.io import wavfile
    sr2, raw = wavfile.read(str(path))
    assert sr2 == 22050 and raw.dtype == np.int16 and np.array_equal(raw, float_to_pcm16(y))


def test_bench_reports_measured_traffic_only_for_the_kernel_it_was_measured_on(tmp_path, monkeypatch):
    """bench.py's roofline.traffic comes from a counter pass filed under the workload's key together
    with a digest of the kernel sources; another workload, or sources that changed since, give
    null and the reason."""
    import json
    import bench
    key = bench.traffic_key(1024, 256, 64, 430, "fixed", "default")
    assert key == "n_fft1024_hop256_b64_t430_fixed_default"
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    (tmp_path / "profiles").mkdir()
    assert bench.measured_traffic(key)[0] is None
    monkeypatch.setattr(bench, "engine_source_digest", lambda: "abc")
    doc = {"entries": {key: {"bytes_per_launch": 123, "engine_source_digest": "abc", "profile": "profiles/x.json"}}}
    (tmp_path / "profiles" / "traffic.json").write_text(json.dumps(doc))
    assert bench.measured_traffic(key) == (123, "profiles/x.json")
    assert bench.measured_traffic(bench.traffic_key(2048, 256, 64, 865, "fixed", "default"))[0] is None
    monkeypatch.setattr(bench, "engine_source_digest", lambda: "changed")
    value, why = bench.measured_traffic(key)
    assert value is None and "stale" in why
