"""The mel producer (tts_amd/acoustic.py) against an independent numpy forward pass written in
Keras conventions (weight layouts, gate order, masking, epsilon) of the reference model
TTS/Fastspeech_1.py:95-128.  TensorFlow is not available and the reference ships no weights, so
equality with the Keras model itself is unpinned; this pins the layout mapping and the
reference's call-graph quirks."""
import numpy as np
import pytest
import torch

from tts_amd.acoustic import FastSpeechAcousticModel, length_regulate

VOCAB = 72  # TTS/train.py:43


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def keras_lstm(x, mask, kernel, recurrent, bias, go_backwards=False):
    """LSTM(return_sequences=True) on one row; masked steps carry the state and repeat the output."""
    units = recurrent.shape[0]
    steps = range(len(x) - 1, -1, -1) if go_backwards else range(len(x))
    h = np.zeros(units)
    c = np.zeros(units)
    prev_out = np.zeros(units)
    out = np.zeros((len(x), units))
    for t in steps:
        if mask[t]:
            z = x[t] @ kernel + h @ recurrent + bias
            i, f, g, o = np.split(z, 4)
            c = _sigmoid(f) * c + _sigmoid(i) * np.tanh(g)
            h = _sigmoid(o) * np.tanh(c)
            prev_out = h
        out[t] = prev_out
    return out


def keras_conv1d(x, kernel, bias, dilation=1):
    """Conv1D(padding='same'): kernel (k, in, out); x (T, in)."""
    k = kernel.shape[0]
    total = dilation * (k - 1)
    left = total // 2
    xp = np.pad(x, ((left, total - left), (0, 0)))
    out = np.zeros((len(x), kernel.shape[2]))
    for j in range(k):
        out += xp[j * dilation: j * dilation + len(x)] @ kernel[j]
    return out + bias


def layer_norm(x, gamma, beta):
    mu = x.mean(-1, keepdims=True)
    var = x.var(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + 1e-3) * gamma + beta


def random_weights(rng, mel_dim=80, hidden=256, emb=256):
    w = {"embed": rng.normal(0, 0.5, (VOCAB, emb))}
    dims = [(emb, 512), (1024, 512), (1024, 128)]
    for n, (i, u) in enumerate(dims, 1):
        for d in ("fwd", "bwd"):
            w[f"lstm{n}_{d}_kernel"] = rng.normal(0, 1 / np.sqrt(i), (i, 4 * u))
            w[f"lstm{n}_{d}_recurrent"] = rng.normal(0, 1 / np.sqrt(u), (u, 4 * u))
            w[f"lstm{n}_{d}_bias"] = rng.normal(0, 0.1, 4 * u)
    for n, i in ((1, 256), (2, hidden)):
        w[f"enc_conv{n}_kernel"] = rng.normal(0, 1 / np.sqrt(5 * i), (5, i, hidden))
        w[f"enc_conv{n}_bias"] = rng.normal(0, 0.1, hidden)
        w[f"enc_norm{n}_gamma"] = rng.normal(1, 0.1, hidden)
        w[f"enc_norm{n}_beta"] = rng.normal(0, 0.1, hidden)
    for n in (1, 2, 3):
        w[f"dur_conv{n}_kernel"] = rng.normal(0, 1 / np.sqrt(3 * hidden), (3, hidden, hidden))
        w[f"dur_conv{n}_bias"] = rng.normal(0, 0.1, hidden)
        w[f"dur_norm{n}_gamma"] = rng.normal(1, 0.1, hidden)
        w[f"dur_norm{n}_beta"] = rng.normal(0, 0.1, hidden)
    w["dur_out_kernel"] = rng.normal(0, 0.1, (hidden, 1))
    w["dur_out_bias"] = rng.normal(0, 0.1, 1)
    w["mel_out_kernel"] = rng.normal(0, 1 / np.sqrt(hidden), (1, hidden, mel_dim))
    w["mel_out_bias"] = rng.normal(0, 0.1, mel_dim)
    w["post_conv_kernel"] = rng.normal(0, 1 / np.sqrt(5 * mel_dim), (5, mel_dim, mel_dim))
    w["post_conv_bias"] = rng.normal(0, 0.1, mel_dim)
    w["post_bn_gamma"] = rng.normal(1, 0.1, mel_dim)
    w["post_bn_beta"] = rng.normal(0, 0.1, mel_dim)
    w["post_bn_mean"] = rng.normal(0, 0.1, mel_dim)
    w["post_bn_var"] = rng.uniform(0.5, 1.5, mel_dim)
    # the module stores float32 parameters: use values float32 holds exactly
    return {k: v.astype(np.float32).astype(np.float64) for k, v in w.items()}


def numpy_forward(w, ids, durations):
    """Row 0 through the reference's call graph (TTS/Fastspeech_1.py:113-128)."""
    mask = ids != 0
    x = w["embed"][ids]
    for n in (1, 2, 3):
        f = keras_lstm(x, mask, w[f"lstm{n}_fwd_kernel"], w[f"lstm{n}_fwd_recurrent"], w[f"lstm{n}_fwd_bias"])
        b = keras_lstm(x, mask, w[f"lstm{n}_bwd_kernel"], w[f"lstm{n}_bwd_recurrent"], w[f"lstm{n}_bwd_bias"],
                       go_backwards=True)
        x = np.concatenate([f, b], axis=-1)
    for n in (1, 2):
        x = np.maximum(keras_conv1d(x, w[f"enc_conv{n}_kernel"], w[f"enc_conv{n}_bias"], dilation=2), 0)
        x = layer_norm(x, w[f"enc_norm{n}_gamma"], w[f"enc_norm{n}_beta"])
    dur = x
    for n in (1, 2, 3):
        dur = np.maximum(keras_conv1d(dur, w[f"dur_conv{n}_kernel"], w[f"dur_conv{n}_bias"]), 0)
        dur = layer_norm(dur, w[f"dur_norm{n}_gamma"], w[f"dur_norm{n}_beta"])
    dur = dur @ w["dur_out_kernel"] + w["dur_out_bias"]
    d = np.round(np.exp(durations)).astype(np.int64)
    reps = np.repeat(np.arange(len(ids)), d)
    x = np.repeat(x, d, axis=0)[reps]
    mel = keras_conv1d(x, w["mel_out_kernel"], w["mel_out_bias"])
    post = mel
    for i in range(5):
        post = np.tanh(keras_conv1d(post, w["post_conv_kernel"], w["post_conv_bias"]))
        if i < 4:
            post = (post - w["post_bn_mean"]) / np.sqrt(w["post_bn_var"] + 1e-3) * w["post_bn_gamma"] + w["post_bn_beta"]
    return mel + post, dur


@pytest.mark.parametrize("pad", [0, 4])
def test_matches_keras_convention_forward(pad):
    rng = np.random.default_rng(5)
    w = random_weights(rng)
    n = 9
    ids = np.concatenate([rng.integers(1, VOCAB, n), np.zeros(pad, np.int64)])
    durations = np.log(np.concatenate([rng.integers(1, 6, n), np.ones(pad)]).astype(np.float64))
    model = FastSpeechAcousticModel(VOCAB).load_keras_weights(w).double()
    out = model((ids[None], durations[None]))
    want, want_dur = numpy_forward(w, ids, durations)
    assert out.shape == (1, want.shape[0], 80)
    assert np.abs(out[0].numpy() - want).max() < 1e-9
    assert np.abs(model.last_duration_prediction[0].numpy() - want_dur).max() < 1e-9


def test_length_regulator_double_index():
    # durations 2, 1, 3 -> frame owner p = [0,0,1,2,2,2]; the reference gathers the expanded
    # sequence with p again: out[i] = expanded[p[i]] = x[p[p[i]]] = x[[0,0,0,1,1,1]]
    x = torch.arange(3, dtype=torch.float32)[None, :, None]
    out = length_regulate(x, torch.log(torch.tensor([[2.0, 1.0, 3.0]])))
    assert out[0, :, 0].tolist() == [0, 0, 0, 1, 1, 1]


def test_rejects_inner_padding_and_shape_mismatch():
    model = FastSpeechAcousticModel(VOCAB)
    with pytest.raises(ValueError):
        model((np.array([[3, 0, 5]]), np.zeros((1, 3))))
    with pytest.raises(ValueError):
        model((np.array([[3, 4, 5]]), np.zeros((1, 4))))
    # zero frames for a phoneme can make tf.gather's index run past the expanded sequence
    with pytest.raises(IndexError):
        model((np.array([[3, 4, 5]]), np.log(np.array([[0.4, 0.4, 1.0]]))))


def test_float32_close_to_float64():
    rng = np.random.default_rng(6)
    w = random_weights(rng)
    ids = rng.integers(1, VOCAB, 12)
    durations = np.log(rng.integers(2, 7, 12).astype(np.float64))
    m32 = FastSpeechAcousticModel(VOCAB).load_keras_weights(w)
    out = m32((ids[None], durations[None].astype(np.float32)))[0].numpy()
    want, _ = numpy_forward(w, ids, durations)
    assert np.linalg.norm(out - want) / np.linalg.norm(want) < 1e-4
