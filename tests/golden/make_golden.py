"""Regenerates the fixtures in this directory.  Runs only where /root/reference exists.

Two kinds of fixture are written:

* data the reference itself ships for the vocoder path (its one input mel,
  the normalisation constants and its two output waveforms), repacked as
  compressed .npz so the GPU box, which has no /root/reference, can read them;
* small per-stage tensors computed by ``oracle/griffinlim_oracle.py`` with a
  fixed initial phase.  These are regression anchors for the oracle (they are
  NOT outputs of librosa, which is not installable here).

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np
from scipy.io import wavfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import griffinlim_oracle as O  # noqa: E402

REF = "/root/reference"


def main():
    mel = np.load(f"{REF}/vocoder/audio/LJ001-0001.npy")
    mean, std = np.load(f"{REF}/vocoder/audio/mel_mean_std.npy")
    np.savez_compressed(f"{HERE}/lj001_mel.npz", mel_db=mel, mean=mean, std=std)

    wavs = {}
    for key, rel in (("vocoder_audio", "vocoder/audio/reconstructed_griffin.wav"),
                     ("vocoder", "vocoder/reconstructed_griffin.wav")):
        sr, w = wavfile.read(f"{REF}/{rel}")
        assert sr == 22050 and w.dtype == np.int16
        wavs[key] = w
    np.savez_compressed(f"{HERE}/ref_wavs.npz", **wavs)

    kat = {}
    s = O.mel_invert(mel, mean, std)
    a = np.abs(s).astype(np.float64)
    kat["linear_spec"] = {"shape": list(s.shape), "min": float(s.min()), "max": float(s.max()),
                          "neg_frac": float((s < 0).mean())}
    sc_ref = {}
    for key, w in wavs.items():
        d = np.abs(O.stft(w.astype(np.float32) / 32768.0, n_fft=2048, hop_length=256,
                          win_length=2048)).astype(np.float64)
        g = (d * a).sum() / (d * d).sum()
        sc_ref[key] = float(np.linalg.norm(g * d - a) / np.linalg.norm(a))
    kat["ref_wav_best_gain_sc"] = sc_ref

    ks = [1, 2, 4, 8, 16, 32, 60]
    curves = {}
    for name, n_fft, frames in (("n2048_t865", 2048, 865), ("n1024_t430", 1024, 430)):
        sp = O.mel_invert(mel[:frames], mean, std, n_fft=n_fft)
        ph = np.exp(2j * np.pi * np.random.default_rng(0).random(sp.shape))
        _, sc = O.griffinlim(sp, n_iter=60, hop_length=256, win_length=n_fft, init_phase=ph,
                             return_sc=True)
        curves[name] = {"k": ks, "sc": [float(sc[k - 1]) for k in ks]}
    kat["sc_curve_rng0"] = curves
    with open(f"{HERE}/kat.json", "w") as f:
        json.dump(kat, f, indent=1)

    # small per-stage tensors
    for name, n_fft, hop, t0, frames in (("stages_n1024_t16", 1024, 256, 100, 16),
                                          ("stages_n2048_t12", 2048, 256, 200, 12)):
        m = mel[t0:t0 + frames]
        sp = O.mel_invert(m, mean, std, n_fft=n_fft)
        ph = np.exp(2j * np.pi * np.random.default_rng(7).random(sp.shape)).astype(np.complex64)
        y, sc, snaps = O.griffinlim(sp, n_iter=8, hop_length=hop, win_length=n_fft,
                                    init_phase=ph, return_sc=True, snapshots=[0, 1, 2, 4, 8])
        d1 = O.stft(snaps[0], n_fft=n_fft, hop_length=hop, win_length=n_fft)
        np.savez_compressed(
            f"{HERE}/{name}.npz", mel_db=m, mean=mean, std=std, n_fft=n_fft, hop=hop,
            linear_spec=sp, phase0=ph, y_k0=snaps[0], y_k1=snaps[1], y_k2=snaps[2],
            y_k4=snaps[4], y_k8=snaps[8], rebuilt_k1=d1, sc=sc,
            audio=O.peak_normalise(y))


if __name__ == "__main__":
    main()
