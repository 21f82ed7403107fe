"""Mel producer in front of the vocoder: the FastSpeech-style acoustic model of the reference
(TTS/Fastspeech_1.py:95-128) as an inference-only PyTorch module, so that phoneme ids go in
and a waveform comes out without the mel ever leaving the GPU (the reference predicts the mel
with Keras on the host and hands a numpy array to the vocoder, test.py:188-192).

The layer stack, hyper-parameters and the quirks of the reference's ``call`` are kept:

* embedding with id 0 as padding (``mask_zero=True``, :101); the mask is honoured by the three
  bidirectional LSTMs the way Keras does it (state carried over masked steps, output repeated;
  only right-padded rows are supported here) and then dropped by the first Conv1D;
* encoder = BiLSTM(512) -> BiLSTM(512) -> BiLSTM(128) -> 2 x [Conv1D(k=5, dilation 2, relu),
  LayerNorm] (:12-38); LayerNorm / BatchNorm epsilon are the Keras defaults (1e-3);
* the duration predictor (:40-56) is evaluated but, as in the reference, does not influence
  the output: the length regulator uses the durations that were passed in;
* the length regulator (:58-63) works on batch row 0 only, takes ``round(exp(d))`` frames per
  phoneme and then indexes the expanded sequence a second time with the phoneme index of every
  frame (``tf.gather(tf.repeat(x, d), tf.repeat(range, d))``) - kept as written;
* the post-net (:76-92) applies ONE Conv1D(k=5, tanh) five times with shared weights, the first
  four followed by one shared BatchNorm (moving statistics at inference); reassigning
  ``self.activation`` inside the loop has no effect on an already constructed layer, so the
  fifth application is tanh as well.

No trained weights exist in the reference and TensorFlow is not available, so equality with
the Keras model is unpinned; `load_keras_weights` documents the layout mapping, and the tests
check the module against an independent numpy forward pass written in Keras conventions.
"""
from __future__ import annotations

import numpy as np
import torch
from torch import nn
import torch.nn.functional as F

__all__ = ["FastSpeechAcousticModel", "synthesize"]

_KERAS_EPS = 1e-3


class _MaskedBiLSTM(nn.Module):
    """``Bidirectional(LSTM(units, return_sequences=True))`` with a right-padding mask."""

    def __init__(self, in_dim, units):
        super().__init__()
        self.lstm = nn.LSTM(in_dim, units, batch_first=True, bidirectional=True)
        self.units = units

    def forward(self, x, lengths):
        t_max = x.shape[1]
        if int(lengths.min()) == t_max:
            return self.lstm(x)[0]
        packed = nn.utils.rnn.pack_padded_sequence(x, lengths.cpu(), batch_first=True, enforce_sorted=False)
        out, _ = self.lstm(packed)
        out, _ = nn.utils.rnn.pad_packed_sequence(out, batch_first=True, total_length=t_max)
        # Keras repeats the last valid output over masked steps; in the time-reversed direction
        # the masked steps come first and emit the initial (zero) output, as the packed form does
        last = (lengths.to(x.device) - 1).clamp_min(0)
        fwd_last = out[torch.arange(x.shape[0], device=x.device), last, : self.units]
        pad = torch.arange(t_max, device=x.device)[None, :] >= lengths.to(x.device)[:, None]
        has_valid = (lengths.to(x.device) > 0)[:, None, None]
        fill = torch.where(has_valid, fwd_last[:, None, :], torch.zeros_like(fwd_last[:, None, :]))
        fwd = torch.where(pad[:, :, None], fill, out[..., : self.units])
        return torch.cat([fwd, out[..., self.units:]], dim=-1)


class _ConvNorm(nn.Module):
    """Conv1D(padding='same', activation='relu') followed by LayerNormalization."""

    def __init__(self, in_dim, out_dim, kernel, dilation=1):
        super().__init__()
        self.conv = nn.Conv1d(in_dim, out_dim, kernel, padding="same", dilation=dilation)
        self.norm = nn.LayerNorm(out_dim, eps=_KERAS_EPS)

    def forward(self, x):  # (B, T, C)
        y = F.relu(self.conv(x.transpose(1, 2))).transpose(1, 2)
        return self.norm(y)


class Encoder(nn.Module):
    def __init__(self, in_dim, hidden_dim):
        super().__init__()
        self.bi_lstm1 = _MaskedBiLSTM(in_dim, 512)
        self.bi_lstm2 = _MaskedBiLSTM(1024, 512)
        self.bi_lstm3 = _MaskedBiLSTM(1024, 128)
        self.block1 = _ConvNorm(256, hidden_dim, 5, dilation=2)
        self.block2 = _ConvNorm(hidden_dim, hidden_dim, 5, dilation=2)

    def forward(self, x, lengths):
        x = self.bi_lstm3(self.bi_lstm2(self.bi_lstm1(x, lengths), lengths), lengths)
        return self.block2(self.block1(x))


class DurationPredictor(nn.Module):
    def __init__(self, hidden_dim):
        super().__init__()
        self.blocks = nn.ModuleList([_ConvNorm(hidden_dim, hidden_dim, 3) for _ in range(3)])
        self.out = nn.Linear(hidden_dim, 1)

    def forward(self, x):
        for b in self.blocks:
            x = b(x)
        return self.out(x)


def length_regulate(x, durations):
    """Reference LengthRegulator.call (TTS/Fastspeech_1.py:59-63) on batch row 0."""
    d = torch.round(torch.exp(durations[0].float())).to(torch.int64)
    idx = torch.repeat_interleave(torch.arange(x.shape[1], device=x.device), d)
    expanded = torch.repeat_interleave(x[0], d, dim=0)
    if idx.numel() and int(idx.max()) >= expanded.shape[0]:
        raise IndexError("length regulator: phoneme index beyond the expanded sequence "
                         "(tf.gather would fail the same way)")
    return expanded[idx].unsqueeze(0)


class PostNet(nn.Module):
    def __init__(self, mel_dim):
        super().__init__()
        self.post_conv1 = nn.Conv1d(mel_dim, mel_dim, 5, padding="same")
        self.bnorm1 = nn.BatchNorm1d(mel_dim, eps=_KERAS_EPS, momentum=0.01)

    def forward(self, x):  # (B, T, C)
        y = x.transpose(1, 2)
        for i in range(5):
            y = torch.tanh(self.post_conv1(y))
            if i < 4:
                y = self.bnorm1(y)
        return y.transpose(1, 2)


class FastSpeechAcousticModel(nn.Module):
    """``FastSpeechAcousticModel(vocab_size)((phoneme_ids, durations)) -> (1, T, mel_dim)``.

    phoneme_ids: (B, Tp) integers, 0 = padding (right padding only); durations: (B, Tp)
    log-durations (frames = round(exp(d))).  Only row 0 reaches the output, as in the
    reference.  The module is inference-only: it is put in eval mode on construction and
    `forward` runs without gradient tracking.
    """

    def __init__(self, vocab_size, embedding_dim=256, mel_dim=80, hidden_dim=256):
        super().__init__()
        self.embed = nn.Embedding(vocab_size, embedding_dim)
        self.encoder = Encoder(embedding_dim, hidden_dim)
        self.duration_predictor = DurationPredictor(hidden_dim)
        self.mel_out = nn.Conv1d(hidden_dim, mel_dim, 1)
        self.post_net = PostNet(mel_dim)
        self.last_duration_prediction = None
        self.eval()

    @torch.no_grad()
    def forward(self, inputs):
        phoneme_ids, durations = inputs
        phoneme_ids = torch.as_tensor(phoneme_ids, device=self.embed.weight.device).long()
        durations = torch.as_tensor(durations, device=self.embed.weight.device)
        if phoneme_ids.dim() != 2 or durations.shape != phoneme_ids.shape:
            raise ValueError("phoneme_ids and durations must both be (B, Tp)")
        valid = phoneme_ids != 0
        lengths = valid.sum(dim=1)
        prefix = torch.arange(phoneme_ids.shape[1], device=valid.device)[None, :] < lengths[:, None]
        if not torch.equal(valid, prefix):
            raise ValueError("padding id 0 inside a sequence: only right-padded rows are supported")
        x = self.embed(phoneme_ids)
        x = self.encoder(x, lengths)
        self.last_duration_prediction = self.duration_predictor(x)
        x = length_regulate(x, durations)
        mel_output = self.mel_out(x.transpose(1, 2)).transpose(1, 2)
        return mel_output + self.post_net(mel_output)

    # -- Keras checkpoint layout ------------------------------------------------------
    def load_keras_weights(self, w):
        """Copies weights given in Keras layouts.  `w` maps names to numpy arrays:

        ``embed`` (vocab, E); for n in 1..3 and dir in fwd/bwd ``lstm{n}_{dir}_kernel``
        (in, 4u), ``..._recurrent`` (u, 4u), ``..._bias`` (4u) - Keras gate order i, f, c, o is
        PyTorch's i, f, g, o, and Keras' single bias goes to ``bias_ih`` with ``bias_hh`` zero;
        ``enc_conv{1,2}_kernel`` (k, in, out), ``..._bias``, ``enc_norm{1,2}_gamma/_beta``;
        ``dur_conv{1..3}_kernel/_bias``, ``dur_norm{1..3}_gamma/_beta``, ``dur_out_kernel``
        (in, 1), ``dur_out_bias``; ``mel_out_kernel`` (1, in, mel), ``mel_out_bias``;
        ``post_conv_kernel`` (5, mel, mel), ``post_conv_bias``, ``post_bn_gamma/_beta/_mean/_var``.
        """
        def t(a):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))

        def conv(mod, name):
            mod.weight.copy_(t(np.transpose(w[name + "_kernel"], (2, 1, 0))))
            mod.bias.copy_(t(w[name + "_bias"]))

        def norm(mod, name):
            mod.weight.copy_(t(w[name + "_gamma"]))
            mod.bias.copy_(t(w[name + "_beta"]))

        with torch.no_grad():
            self.embed.weight.copy_(t(w["embed"]))
            for n, layer in enumerate((self.encoder.bi_lstm1, self.encoder.bi_lstm2, self.encoder.bi_lstm3), 1):
                for d, suffix in (("fwd", ""), ("bwd", "_reverse")):
                    getattr(layer.lstm, "weight_ih_l0" + suffix).copy_(t(w[f"lstm{n}_{d}_kernel"].T))
                    getattr(layer.lstm, "weight_hh_l0" + suffix).copy_(t(w[f"lstm{n}_{d}_recurrent"].T))
                    getattr(layer.lstm, "bias_ih_l0" + suffix).copy_(t(w[f"lstm{n}_{d}_bias"]))
                    getattr(layer.lstm, "bias_hh_l0" + suffix).zero_()
            for n, block in enumerate((self.encoder.block1, self.encoder.block2), 1):
                conv(block.conv, f"enc_conv{n}")
                norm(block.norm, f"enc_norm{n}")
            for n, block in enumerate(self.duration_predictor.blocks, 1):
                conv(block.conv, f"dur_conv{n}")
                norm(block.norm, f"dur_norm{n}")
            self.duration_predictor.out.weight.copy_(t(w["dur_out_kernel"].T))
            self.duration_predictor.out.bias.copy_(t(w["dur_out_bias"]))
            conv(self.mel_out, "mel_out")
            conv(self.post_net.post_conv1, "post_conv")
            norm(self.post_net.bnorm1, "post_bn")
            self.post_net.bnorm1.running_mean.copy_(t(w["post_bn_mean"]))
            self.post_net.bnorm1.running_var.copy_(t(w["post_bn_var"]))
        return self


def synthesize(model, phoneme_ids, durations, mean, std, *, return_mel=False, **vocoder_kw):
    """phoneme ids + durations -> waveform with the mel staying on the model's device:
    the tail of the reference's inference script (test.py:182-192) without the host round trip.
    `vocoder_kw` are the keyword arguments of `mel_to_audio_griffin_lim`.  Returns a device
    tensor (and the (T, mel_dim) mel when `return_mel`)."""
    from .vocoder import mel_to_audio_griffin_lim
    mel = model((phoneme_ids, durations))[0]
    vocoder_kw.setdefault("n_mels", mel.shape[1])
    audio = mel_to_audio_griffin_lim(mel, mean, std, device=mel.device, return_tensor=True, **vocoder_kw)
    return (audio, mel) if return_mel else audio
