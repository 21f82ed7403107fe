"""Minimal PCM16 WAV writer/reader standing in for ``soundfile.write(path, audio, sr)`` as the
reference calls it after vocoding (vocoder/mel_audio.py:139, test.py:194): float audio in
[-1, 1] -> mono 16-bit PCM.  The reference's own output files (peaks of +32767 / -32768 for
a waveform normalised to just under 1.0) fix the conversion: scale by 32768, round to
nearest, clip to the int16 range."""
from __future__ import annotations

import wave

import numpy as np


def float_to_pcm16(audio):
    a = np.asarray(audio, dtype=np.float64)
    return np.clip(np.rint(a * 32768.0), -32768, 32767).astype(np.int16)


def write_wav(path, audio, sr=22050):
    pcm = float_to_pcm16(audio)
    if pcm.ndim != 1:
        raise ValueError("mono audio expected")
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(int(sr))
        w.writeframes(pcm.astype("<i2").tobytes())


def read_wav(path):
    """Returns (float32 audio in [-1, 1), sr) of a mono PCM16 file."""
    with wave.open(str(path), "rb") as w:
        if w.getnchannels() != 1 or w.getsampwidth() != 2:
            raise ValueError("mono PCM16 expected")
        sr = w.getframerate()
        pcm = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
    return pcm.astype(np.float32) / 32768.0, sr
