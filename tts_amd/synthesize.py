"""Command-line form of the reference's inference tail (test.py:191-194): a predicted mel
and the dataset statistics in, a PCM16 wav out.

    python -m tts_amd.synthesize --mel mel.npy --stats mel_mean_std.npy --out out.wav

``mel.npy`` is a (T, 80) float32 normalised-dB mel as the reference's acoustic model emits
(test.py:188) or as its preprocessing stores (vocoder/audio/LJ001-0001.npy); several --mel
files are synthesised as one batch.  The acoustic model itself (TensorFlow) is outside this
package.
"""
from __future__ import annotations

import argparse
import os

import numpy as np


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--mel", nargs="+", required=True, help="(T, n_mels) float32 .npy file(s)")
    ap.add_argument("--stats", required=True, help=".npy with [mean, std] (mel_mean_std.npy)")
    ap.add_argument("--out", required=True, help="output .wav (one mel) or directory (several)")
    ap.add_argument("--sr", type=int, default=22050)
    ap.add_argument("--n-fft", type=int, default=2048)
    ap.add_argument("--hop-length", type=int, default=256)
    ap.add_argument("--win-length", type=int, default=2048)
    ap.add_argument("--fmax", type=float, default=8000)
    ap.add_argument("--n-iter", type=int, default=60)
    ap.add_argument("--momentum", type=float, default=0.99)
    ap.add_argument("--seed", type=int, default=None, help="seed of the initial phase (default: random)")
    args = ap.parse_args(argv)

    from . import griffin_lim_batch
    from .wavio import write_wav

    mean, std = np.load(args.stats, allow_pickle=False)[:2]
    mels = [np.load(p, allow_pickle=False).astype(np.float32) for p in args.mel]
    n_mels = mels[0].shape[1]
    seed = args.seed if args.seed is not None else int(np.random.SeedSequence().generate_state(1)[0])
    audio, n_samples = griffin_lim_batch(mels, mean, std, sr=args.sr, n_fft=args.n_fft,
                                         hop_length=args.hop_length, win_length=args.win_length,
                                         n_mels=n_mels, fmax=args.fmax, n_iter=args.n_iter,
                                         momentum=args.momentum, seed=seed)
    audio = audio.cpu().numpy()
    if len(mels) == 1 and not os.path.isdir(args.out):
        targets = [args.out]
    else:
        os.makedirs(args.out, exist_ok=True)
        targets = [os.path.join(args.out, os.path.splitext(os.path.basename(p))[0] + ".wav") for p in args.mel]
    for path, row, n in zip(targets, audio, n_samples):
        write_wav(path, row[:n], args.sr)
        print(f"{path}: {n} samples, {n / args.sr:.2f} s")


if __name__ == "__main__":
    main()
