"""GPU parity: every stage of the HIP path, called through the C ABI (ctypes via
tts_amd.vocoder.GriffinLimPlan), against the CPU oracle on identical inputs and
an identical initial phase.

Tolerances: linear spectrogram after mel inversion within
1e-4 relative L2; waveform after k iterations within 1e-3 relative L2 in fp32;
per-iteration spectral-convergence curve within 1 %.  Fast Griffin-Lim with
momentum 0.99 amplifies fp32 rounding differences in k; the measured drift at
k = 16 / 32 / 60 on the reference's own mel is printed by
test_fixture_60_iterations (run pytest with -s, or read the captured output) and
asserted below 1e-3 at every one of them, for both forms of the momentum term.
"""
import numpy as np
import pytest

from conftest import rel_l2
from helpers import MEAN, STD, from_btf2, phasors, synthetic_mel, to_btf, to_btf2

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def O():
    from oracle import griffinlim_oracle
    return griffinlim_oracle


@pytest.fixture(scope="module")
def voc():
    import tts_amd.vocoder as v
    return v


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


SHAPES = [(1024, 256, 1024), (2048, 256, 2048), (2048, 512, 2048), (1024, 128, 1024), (1024, 512, 800),
          (2048, 1024, 2048)]


@pytest.mark.parametrize("n_fft,hop,win", SHAPES)
def test_stft_matches_oracle(voc, O, n_fft, hop, win):
    frames = 37
    y = np.random.default_rng(3).standard_normal(hop * (frames - 1)).astype(np.float32)
    plan = voc.GriffinLimPlan(n_fft, hop, win, 80, frames, 1)
    plan.set_option("frames_per_run", 9)
    d = plan.stft(dev(y[None])).cpu().numpy()[0]
    ref = O.stft(y, n_fft=n_fft, hop_length=hop, win_length=win)
    assert rel_l2(from_btf2(d), ref) < 2e-6
    plan.close()


@pytest.mark.parametrize("n_fft,hop,win", SHAPES)
def test_istft_matches_oracle(voc, O, n_fft, hop, win):
    frames = 41
    rng = np.random.default_rng(4)
    f = n_fft // 2 + 1
    d = (rng.standard_normal((f, frames)) + 1j * rng.standard_normal((f, frames))).astype(np.complex64)
    plan = voc.GriffinLimPlan(n_fft, hop, win, 80, frames, 1)
    plan.set_option("frames_per_run", 10)
    y = plan.istft(dev(to_btf2(d))).cpu().numpy()[0]
    ref = O.istft(d, hop_length=hop, win_length=win)
    assert y.shape == ref.shape
    assert rel_l2(y, ref) < 2e-6
    plan.close()


@pytest.mark.parametrize("n_fft", [1024, 2048])
def test_mel_inversion_within_1e4(voc, O, lj001, n_fft):
    mel, mean, std = lj001
    plan = voc.GriffinLimPlan(n_fft, 256, n_fft, 80, mel.shape[0], 1)
    plan.set_mel_basis(22050, 0.0, 8000)
    s = plan.mel_invert(dev(mel[None]), mean, std).cpu().numpy()[0]
    ref = O.mel_invert(mel, mean, std, n_fft=n_fft)
    assert s.shape == ref.T.shape
    assert rel_l2(s, ref.T) < 1e-4
    # the pseudo-inverse output is signed and must stay signed
    assert abs(float((s < 0).mean()) - float((ref < 0).mean())) < 2e-3
    plan.close()


@pytest.mark.parametrize("n_fft,hop,frames,fpr", [(1024, 256, 120, 12), (2048, 256, 96, 12),
                                                  (2048, 512, 64, 7), (1024, 256, 7, 3)])
def test_griffinlim_waveform_and_curve(voc, O, lj001, n_fft, hop, frames, fpr):
    mel, mean, std = lj001
    s = O.mel_invert(mel[50:50 + frames], mean, std, n_fft=n_fft)
    ph = phasors(11, s.shape[0], frames)
    ks = [1, 4, 8, 16]
    _, sc_ref, snaps = O.griffinlim(s, n_iter=16, hop_length=hop, win_length=n_fft, init_phase=ph,
                                    return_sc=True, snapshots=ks + [0])
    plan = voc.GriffinLimPlan(n_fft, hop, n_fft, 80, frames, 1)
    plan.set_option("frames_per_run", fpr)
    sd, pd = dev(to_btf(s)), dev(to_btf2(ph))
    y0 = plan.griffinlim(sd, phase0=pd, n_iter=0).cpu().numpy()[0]
    assert rel_l2(y0, snaps[0]) < 2e-6
    for k in ks:
        y, sc = plan.griffinlim(sd, phase0=pd, n_iter=k, return_sc=True)
        y, sc = y.cpu().numpy()[0], sc.cpu().numpy()[0]
        assert rel_l2(y, snaps[k]) < 1e-3, k
        assert np.max(np.abs(sc - sc_ref[:k]) / sc_ref[:k]) < 1e-2
    plan.close()


@pytest.mark.parametrize("form", ["time-domain momentum", "spectral momentum", "one frame per wave"])
@pytest.mark.parametrize("n_fft,frames", [(1024, 430), (2048, 865)])
def test_fixture_60_iterations(voc, O, lj001, n_fft, frames, form, capsys):
    """The reference's own mel at the reference's iteration count (vocoder/mel_audio.py:126),
    at the benchmark shape (n_fft 1024, 430 frames) and the reference's defaults (2048, 865):
    waveform within 1e-3 relative L2 after 16, 32 and 60 iterations, curve within 1 %, for the
    three forms of the iteration kernel: runs of frames per wave with the momentum term formed
    in the time domain (the default for device-filling batches, the benchmark's kernel), the
    same with the term formed in the frequency domain as librosa writes it, and one frame per
    wave (what a single utterance gets by default)."""
    mel, mean, std = lj001
    s = O.mel_invert(mel[:frames], mean, std, n_fft=n_fft)
    ph = phasors(0, s.shape[0], frames)
    ks = [16, 32, 60]
    y_ref, sc_ref, snaps = O.griffinlim(s, n_iter=60, hop_length=256, win_length=n_fft, init_phase=ph,
                                        return_sc=True, snapshots=ks)
    plan = voc.GriffinLimPlan(n_fft, 256, n_fft, 80, frames, 1)
    plan.set_option("single_frame", 1 if form == "one frame per wave" else 0)
    plan.set_option("linear", 0 if form == "spectral momentum" else 1)
    sd, pd = dev(to_btf(s)), dev(to_btf2(ph))
    drifts = {}
    for k in ks:
        y, sc = plan.griffinlim(sd, phase0=pd, n_iter=k, return_sc=True)
        y, sc = y.cpu().numpy()[0], sc.cpu().numpy()[0]
        assert y.shape == (256 * (frames - 1),)
        assert np.max(np.abs(sc - sc_ref[:k]) / sc_ref[:k]) < 1e-2
        drifts[k] = rel_l2(y, snaps[k])
    with capsys.disabled():
        print(f"\n[drift] n_fft {n_fft}, {frames} frames, {form}: "
              + ", ".join(f"k={k}: {d:.2e}" for k, d in drifts.items())
              + f"; curve max rel err {np.max(np.abs(sc - sc_ref) / sc_ref):.1e}")
    for k, d in drifts.items():
        assert d < 1e-3, (k, d)
    plan.close()


@pytest.mark.parametrize("name", ["stages_n1024_t16", "stages_n2048_t12"])
def test_golden_stage_fixtures(voc, name):
    """The committed per-stage vectors (tests/golden/stages_*.npz: the oracle's tensors for a slice
    of the reference's mel and a fixed phase) through every C-ABI stage: mel inversion, iSTFT of
    the initial angles, STFT of that, the waveform after 1, 2, 4, 8 iterations, the curve, and
    the peak-normalised audio of the one-call entry point."""
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    n_fft, hop = int(g["n_fft"]), int(g["hop"])
    mel, mean, std = g["mel_db"], float(g["mean"]), float(g["std"])
    s, ph = g["linear_spec"], g["phase0"]
    frames = s.shape[1]
    plan = voc.GriffinLimPlan(n_fft, hop, n_fft, 80, frames, 1)
    plan.set_mel_basis(22050, 0.0, 8000)
    got_s = plan.mel_invert(dev(mel[None]), mean, std)
    assert rel_l2(got_s.cpu().numpy()[0], s.T) < 1e-4
    pd = dev(to_btf2(ph))
    for single in (0, 1):
        plan.set_option("single_frame", single)
        for k in (0, 1, 2, 4, 8):
            y, sc = plan.griffinlim(got_s, phase0=pd, n_iter=max(k, 1), return_sc=True) if k else \
                (plan.griffinlim(got_s, phase0=pd, n_iter=0), None)
            assert rel_l2(y.cpu().numpy()[0], g[f"y_k{k}"]) < 1e-4, (single, k)
            if k:
                assert np.max(np.abs(sc.cpu().numpy()[0] - g["sc"][:k]) / g["sc"][:k]) < 1e-4
    d1 = plan.stft(dev(g["y_k0"][None])).cpu().numpy()[0]
    assert rel_l2(from_btf2(d1), g["rebuilt_k1"]) < 2e-6
    audio = plan.mel_to_audio(dev(mel[None]), mean, std, phase0=pd, n_iter=8).cpu().numpy()[0]
    assert rel_l2(audio, g["audio"]) < 1e-4
    plan.close()


@pytest.mark.parametrize("n_fft,hop,lens", [(1024, 256, [33, 7, 64, 20]), (2048, 256, [40, 3, 2, 17, 6]),
                                            (2048, 1024, [9, 2, 5]), (1024, 128, [4, 30, 2])])
def test_ragged_batch_equals_single_runs(voc, O, n_fft, hop, lens):
    """Utterances of different lengths in one batch - down to two frames, shorter than the
    overlap - are bit-identical to the same utterance processed alone, zero past their own
    length, and match the oracle."""
    f = n_fft // 2 + 1
    t_max = max(lens)
    rng = np.random.default_rng(5)
    s = np.zeros((len(lens), t_max, f), np.float32)
    p = np.zeros((len(lens), t_max, f, 2), np.float32)
    for b, t in enumerate(lens):
        s[b, :t] = rng.standard_normal((t, f)) ** 2
        ang = 2 * np.pi * rng.random((t, f))
        p[b, :t, :, 0], p[b, :t, :, 1] = np.cos(ang), np.sin(ang)
    plan = voc.GriffinLimPlan(n_fft, hop, n_fft, 80, t_max, len(lens))
    plan.set_option("frames_per_run", 8)
    yb = plan.griffinlim(dev(s), lens, phase0=dev(p), n_iter=5).cpu().numpy()
    for b, t in enumerate(lens):
        n = hop * (t - 1)
        assert np.all(yb[b, n:] == 0)
        y1 = plan.griffinlim(dev(s[b:b + 1, :t]), [t], phase0=dev(p[b:b + 1, :t]), n_iter=5).cpu().numpy()[0]
        assert np.array_equal(yb[b, :n], y1)
        ref = O.griffinlim(np.ascontiguousarray(s[b, :t].T), n_iter=5, hop_length=hop, win_length=n_fft,
                           init_phase=from_btf2(p[b, :t]))
        assert rel_l2(y1, ref) < 1e-4
    plan.close()


def test_run_length_and_graph_do_not_change_results_much(voc):
    """Different run decompositions only re-associate a few boundary sums; graph replay
    versus eager launches is bit-identical."""
    n_fft, hop, frames = 1024, 256, 90
    f = n_fft // 2 + 1
    rng = np.random.default_rng(6)
    s = dev((rng.standard_normal((1, frames, f)) ** 2).astype(np.float32))
    ang = 2 * np.pi * rng.random((1, frames, f))
    p = dev(np.stack([np.cos(ang), np.sin(ang)], -1).astype(np.float32))
    plan = voc.GriffinLimPlan(n_fft, hop, n_fft, 80, frames, 1)
    outs = {}
    for fpr in (5, 12, 1000):
        plan.set_option("frames_per_run", fpr)
        outs[fpr] = plan.griffinlim(s, phase0=p, n_iter=6).cpu().numpy()[0]
    assert rel_l2(outs[5], outs[1000]) < 1e-5
    assert rel_l2(outs[12], outs[1000]) < 1e-5
    plan.set_option("use_graph", 0)
    eager = plan.griffinlim(s, phase0=p, n_iter=6).cpu().numpy()[0]
    assert np.array_equal(eager, outs[1000])
    # repeated calls are deterministic
    plan.set_option("use_graph", 1)
    again = plan.griffinlim(s, phase0=p, n_iter=6).cpu().numpy()[0]
    assert np.array_equal(again, outs[1000])
    plan.close()


def test_peak_normalize_bit_exact(voc, O):
    rng = np.random.default_rng(8)
    y = (rng.standard_normal((3, 5000)) * np.array([[0.01], [3.0], [1e-4]])).astype(np.float32)
    lens = [5000, 1234, 4097]
    plan = voc.GriffinLimPlan(1024, 256, 1024, 80, 32, 3)
    out = plan.peak_normalize(dev(y.copy()), lens).cpu().numpy()
    for b, n in enumerate(lens):
        assert np.array_equal(out[b, :n], O.peak_normalise(y[b, :n]))
        assert np.array_equal(out[b, n:], y[b, n:])
    plan.close()


def test_drop_in_on_reference_fixture(voc, O, lj001):
    """The reference's input through the reference's signature with its defaults:
    length law, peak normalisation, and the quality window its own outputs sit in
    (best-gain spectral convergence 0.1709 / 0.1738 for the two shipped wavs)."""
    mel, mean, std = lj001
    a = np.abs(O.mel_invert(mel, mean, std)).astype(np.float64)

    def best_gain_sc(audio):
        d = np.abs(O.stft(audio, n_fft=2048, hop_length=256, win_length=2048)).astype(np.float64)
        g = (d * a).sum() / (d * d).sum()
        return np.linalg.norm(g * d - a) / np.linalg.norm(a)

    # the oracle's known-answer window, with the oracle test's phase (default_rng(4))
    audio = voc.mel_to_audio_griffin_lim(mel, mean, std, init_phase=4)
    assert isinstance(audio, np.ndarray) and audio.dtype == np.float32
    assert audio.shape == (221184,)
    peak = float(np.max(np.abs(audio)))
    assert 0.99998 < peak <= 1.0
    sc = best_gain_sc(audio)
    assert 0.165 < sc < 0.180, sc
    # the reference's own call: unseeded phase, drawn on the device.  The end point moves a
    # little with the phase (oracle seeds: 0.170-0.173, an occasional 0.187)
    audio = voc.mel_to_audio_griffin_lim(mel, mean, std)
    assert audio.shape == (221184,) and 0.99998 < float(np.max(np.abs(audio))) <= 1.0
    sc = best_gain_sc(audio)
    assert 0.16 < sc < 0.20, sc
    # seeded phase == oracle with the same generator
    y1 = voc.mel_to_audio_griffin_lim(mel[:100], mean, std, n_iter=8, init_phase=3)
    y2 = O.mel_to_audio_griffin_lim(mel[:100], mean, std, n_iter=8, random_state=3)
    assert rel_l2(y1, y2) < 1e-3


def test_full_size_batch_properties(voc, O):
    """The benchmark batch shape (64 utterances x 430 frames, n_fft=1024, 60 iterations):
    size-independent properties and row independence - an utterance synthesised alone, with its
    identity, gives the same bits.  (The comparison of this call with the oracle, element by
    element on the device-drawn phase, is test_benchmark_call_against_the_oracle.)"""
    b, frames, n_fft, hop = 64, 430, 1024, 256
    mels = np.stack([synthetic_mel(1234 + i, frames) for i in range(b)])
    kw = dict(n_fft=n_fft, win_length=n_fft, hop_length=hop, n_iter=60, seed=99, frames_per_run=9)
    y, n, sc = voc.griffin_lim_batch(mels, MEAN, STD, return_sc=True, **kw)
    y, sc = y.cpu().numpy(), sc.cpu().numpy()
    assert y.shape == (b, hop * (frames - 1)) and n == [hop * (frames - 1)] * b
    assert np.all(np.isfinite(y)) and np.all(np.isfinite(sc))
    assert np.allclose(np.max(np.abs(y), axis=1), 1.0, atol=1e-3)
    # Griffin-Lim converges: the curve ends far below where it starts, for every row
    assert np.all(sc[:, -1] < 0.6 * sc[:, 0])
    # rows are independent: rows 5 and 63 alone, keyed by their identity, give the same samples
    for row in (5, 63):
        y1, _ = voc.griffin_lim_batch(mels[row:row + 1], MEAN, STD, utt_ids=[row], **kw)
        assert np.array_equal(y1.cpu().numpy()[0], y[row])
    # ... and in other company, at another row, with another row stride
    sub = [mels[63], np.concatenate([mels[5], mels[5][:20]]), mels[5]]
    y3, _ = voc.griffin_lim_batch(sub, MEAN, STD, utt_ids=[63, 1000, 5], **kw)
    y3 = y3.cpu().numpy()
    assert np.array_equal(y3[0, :hop * (frames - 1)], y[63]) and np.array_equal(y3[2, :hop * (frames - 1)], y[5])
    # the library's own run table (fitted to the whole batch) moves run boundaries only
    kw_auto = dict(kw, frames_per_run=None)
    ya, _ = voc.griffin_lim_batch(mels, MEAN, STD, **kw_auto)
    assert rel_l2(ya.cpu().numpy()[5], y[5]) < 1e-3


def test_configs2_batch_of_4096(voc, O):
    """BASELINE.json configs[2] on one GPU: 4096 synthetic utterances x 430 frames, 60 iterations, one
    call.  Size-independent properties over the whole batch, and sampled rows against the oracle:
    every sampled utterance is re-synthesised alone with an explicit phase by the library and
    by the oracle (1e-3), and alone with its identity on the device-drawn phase, which must give
    the bits it got inside the batch of 4096."""
    b, frames, n_fft, hop = 4096, 430, 1024, 256
    base = [synthetic_mel(5000 + i, frames) for i in range(32)]
    # 4096 distinct utterances from 32 spectral envelopes: time-reversed / shifted copies
    mels = np.empty((b, frames, 80), np.float32)
    for i in range(b):
        m = base[i % 32]
        mels[i] = np.roll(m[::-1] if (i // 32) % 2 else m, 3 * (i // 64), axis=0)
    kw = dict(n_fft=n_fft, win_length=n_fft, hop_length=hop, n_iter=60, seed=2024, frames_per_run=36)
    y, n, sc = voc.griffin_lim_batch(mels, MEAN, STD, return_sc=True, **kw)
    assert y.shape == (b, hop * (frames - 1)) and n == [hop * (frames - 1)] * b
    assert bool(torch.isfinite(y).all()) and bool(torch.isfinite(sc).all())
    peaks = y.abs().amax(dim=1)
    assert bool(((peaks > 0.999) & (peaks <= 1.0)).all())
    assert bool((sc[:, -1] < 0.6 * sc[:, 0]).all())
    # distinct utterances give distinct waveforms, equal mels with different identities too
    assert not torch.equal(y[0], y[64 * 2])
    rows = [0, 1777, 4095]
    picked = y[rows].cpu().numpy()
    sc_picked = sc[rows].cpu().numpy()
    del y, sc
    for i, row in enumerate(rows):
        y1, _, sc1 = voc.griffin_lim_batch(mels[row:row + 1], MEAN, STD, utt_ids=[row], return_sc=True, **kw)
        assert np.array_equal(y1.cpu().numpy()[0], picked[i])
        # the oracle on the same mel: the curve is phase-independent to a few percent at the end
        # and the waveform check needs the same phase, so both get an explicit one
        ph = phasors(row, n_fft // 2 + 1, frames)
        p0 = dev(to_btf2(ph))
        yo = O.mel_to_audio_griffin_lim(mels[row], MEAN, STD, n_fft=n_fft, win_length=n_fft, hop_length=hop,
                                        n_iter=60, init_phase=ph)
        kw2 = {k: v for k, v in kw.items() if k != "seed"}
        s = O.mel_invert(mels[row], MEAN, STD, n_fft=n_fft)
        ks = [16, 32, 60]
        _, sc_o, snaps = O.griffinlim(s, n_iter=60, hop_length=hop, win_length=n_fft, init_phase=ph,
                                      return_sc=True, snapshots=ks)
        # How far the reference computation itself moves when only its precision changes: these
        # smoothed-noise mels make the momentum iteration far more sensitive to rounding than
        # speech does (the engine compiled in double agrees with the double oracle to 1e-9,
        # tests/test_engine_emulation.py), so beyond k = 32 "within 1e-3" is below what fp32
        # defines and the bound becomes that sensitivity
        _, snaps64 = O.griffinlim(s.astype(np.float64), n_iter=60, hop_length=hop, win_length=n_fft,
                                  init_phase=ph.astype(np.complex128), snapshots=ks)
        for k in ks:
            yk, _ = voc.griffin_lim_batch(mels[row:row + 1], MEAN, STD, phase0=p0, normalize=False,
                                          **dict(kw2, n_iter=k))
            d = rel_l2(yk.cpu().numpy()[0], snaps[k])
            floor = rel_l2(snaps[k], snaps64[k])
            print(f"[configs2] row {row} k={k}: rel L2 vs oracle {d:.2e} (oracle fp32 vs fp64: {floor:.2e})")
            assert d < max(1e-3, 3 * floor), (row, k, d, floor)
            assert k > 32 or d < 1e-3, (row, k, d)
        yg, _ = voc.griffin_lim_batch(mels[row:row + 1], MEAN, STD, phase0=p0, **kw2)
        floor = rel_l2(snaps[60], snaps64[60])
        assert rel_l2(yg.cpu().numpy()[0], yo) < max(1e-3, 3 * floor)
        _, _, sc_g = voc.griffin_lim_batch(mels[row:row + 1], MEAN, STD, phase0=p0, return_sc=True, **kw2)
        assert np.max(np.abs(sc_g.cpu().numpy()[0] - sc_o) / sc_o) < 1e-2
        # the batch's own curve (another phase) ends within a few percent of the oracle's
        assert abs(sc_picked[i, -1] - sc_o[-1]) < 0.1 * sc_o[-1]
    voc._plans.clear()


def test_error_behaviour(voc):
    from tts_amd._lib import GlvError
    with pytest.raises(GlvError):
        voc.GriffinLimPlan(1536, 256, 1536, 80, 16, 1)
    with pytest.raises(GlvError):
        voc.GriffinLimPlan(1024, 200, 1024, 80, 16, 1)
    plan = voc.GriffinLimPlan(1024, 256, 1024, 80, 16, 1)
    s = torch.zeros((1, 16, 513), device="cuda")
    with pytest.raises(GlvError):
        plan.griffinlim(s, momentum=-0.5)          # librosa raises ParameterError
    with pytest.raises(GlvError):
        plan.griffinlim(torch.zeros((2, 16, 513), device="cuda"))   # over max_batch
    with pytest.raises(GlvError):
        plan.mel_invert(torch.zeros((1, 16, 80), device="cuda"), 0.0, 1.0)  # basis not set
    plan.close()
    with pytest.raises(ValueError):
        voc.mel_to_audio_griffin_lim(np.zeros((10, 40), np.float32), 0.0, 1.0)


@pytest.mark.parametrize("n_fft,hop,frames,shards", [(1024, 256, 97, 3), (2048, 256, 120, 2), (1024, 256, 64, 4)])
def test_time_sharded_equals_single_device(voc, O, lj001, n_fft, hop, frames, shards):
    """One utterance split along time into shards that swap n_fft - hop samples per iteration
    (here: several plans on one GPU, the swap is a device copy; across GPUs the same buffers
    travel over RCCL).  Must agree with the unsharded run and the oracle."""
    from tts_amd.distributed import split_frames
    mel, mean, std = lj001
    s = O.mel_invert(mel[:frames], mean, std, n_fft=n_fft)
    ph = phasors(21, s.shape[0], frames)
    n_iter = 6
    ref = O.griffinlim(s, n_iter=n_iter, hop_length=hop, win_length=n_fft, init_phase=ph)
    sd, pd = dev(to_btf(s)), dev(to_btf2(ph))
    whole = voc.GriffinLimPlan(n_fft, hop, n_fft, 80, frames, 1)
    y_whole = whole.griffinlim(sd, phase0=pd, n_iter=n_iter).cpu().numpy()[0]
    whole.close()
    ranges = split_frames(frames, shards, min_frames=2 * (n_fft // hop - 1))
    plans = [voc.GriffinLimPlan(n_fft, hop, n_fft, 80, t1 - t0, 1) for t0, t1 in ranges]
    for p, (t0, t1) in zip(plans, ranges):
        p.set_option("frames_per_run", 9)
        p.shard_begin(sd[0, t0:t1].contiguous(), t0, frames, phase0=pd[0, t0:t1].contiguous())

    def swap():
        packed = [p.shard_halo_pack() for p in plans]
        for r, p in enumerate(plans):
            from_prev = packed[r - 1][1] if r > 0 else None
            from_next = packed[r + 1][0] if r < len(plans) - 1 else None
            p.shard_halo_unpack(from_prev if from_prev is not None else p.shard_halo_buffers()[0],
                                from_next if from_next is not None else p.shard_halo_buffers()[1])

    swap()
    for _ in range(n_iter):
        for p in plans:
            p.shard_step(0.99)
        swap()
    pieces = [p.shard_end() for p in plans]
    y = np.zeros(hop * (frames - 1), np.float32)
    covered = np.zeros_like(y, dtype=bool)
    for piece, n_begin in pieces:
        arr = piece.cpu().numpy()
        assert not covered[n_begin:n_begin + len(arr)].any()
        y[n_begin:n_begin + len(arr)] = arr
        covered[n_begin:n_begin + len(arr)] = True
    assert covered.all()
    assert rel_l2(y, y_whole) < 2e-5
    assert rel_l2(y, ref) < 1e-4
    # peak normalisation from per-shard maxima == global
    peaks = [float(p.peak_reduce(piece.unsqueeze(0)).cpu()[0]) for p, (piece, _) in zip(plans, pieces)]
    assert max(peaks) == float(np.max(np.abs(y)))
    for p in plans:
        p.close()


@pytest.mark.parametrize("n_fft,n_mels", [(1024, 80), (2048, 128)])
def test_nnls_inversion(voc, O, n_fft, n_mels):
    """Secondary mode (librosa.feature.inverse.mel_to_audio): the deterministic start
    clip(pinv @ M, 0) matches exactly to rounding, the projected-gradient iterate matches the
    oracle's, and the mel-domain residual falls."""
    rng = np.random.default_rng(13)
    frames, sr = 50, 22050
    basis = O.mel_filterbank(sr, n_fft, n_mels=n_mels)
    truth = (rng.standard_normal((n_fft // 2 + 1, frames)) ** 2).astype(np.float32)
    m = (basis @ truth).astype(np.float32)                      # (n_mels, T) power mel
    plan = voc.GriffinLimPlan(n_fft, n_fft // 4, n_fft, n_mels, frames, 1)
    plan.set_mel_basis(sr, 0.0, None)
    md = dev(np.ascontiguousarray(m.T[None]))
    x0 = plan.mel_invert_nnls(md, n_steps=0, power=1.0).cpu().numpy()[0].T
    assert rel_l2(x0, O.nnls_init(basis, m)) < 1e-5
    res0 = np.linalg.norm(basis @ x0 - m) / np.linalg.norm(m)
    # plain projected-gradient steps (option "nnls_accel" off) against the oracle's
    plan.set_option("nnls_accel", 0)
    x = plan.mel_invert_nnls(md, n_steps=100, power=1.0).cpu().numpy()[0].T
    want = O.nnls_projected_gradient(basis, m, n_steps=100)
    assert np.all(x >= 0)
    assert rel_l2(x, want) < 1e-4
    res = np.linalg.norm(basis @ x - m) / np.linalg.norm(m)
    assert res < 0.5 * res0 and res < 0.05
    # power = 2 takes the square root (magnitude from power)
    s = plan.mel_invert_nnls(md, n_steps=100, power=2.0).cpu().numpy()[0].T
    assert rel_l2(s, np.sqrt(want)) < 1e-4
    # all steps in one launch (the default) and one product per launch are the same iteration
    plan.set_option("nnls_fused", 0)
    x_unfused = plan.mel_invert_nnls(md, n_steps=100, power=1.0).cpu().numpy()[0].T
    plan.set_option("nnls_fused", 1)
    assert rel_l2(x, x_unfused) < 1e-5
    # frames stop on their own residual: same result as the oracle's rule, residual at the bound
    # for the frames that stopped, and frames do not influence each other
    xt = plan.mel_invert_nnls(md, n_steps=400, power=1.0, tol=0.02).cpu().numpy()[0].T
    want_t = O.nnls_projected_gradient(basis, m, n_steps=400, tol=0.02)
    assert rel_l2(xt, want_t) < 1e-3
    rel = np.linalg.norm(basis @ xt - m, axis=0) / np.linalg.norm(m, axis=0)
    assert np.all(rel < 0.021) and np.all(xt >= 0)
    sub = plan.mel_invert_nnls(dev(np.ascontiguousarray(m.T[None, 7:19])), n_steps=400, power=1.0, tol=0.02)
    assert np.array_equal(sub.cpu().numpy()[0].T, xt[:, 7:19])
    # the default: the same steps extrapolated (Nesterov / FISTA) - against the oracle's identical
    # iteration, and 40 of them must leave a smaller mel residual than 100 plain ones
    plan.set_option("nnls_accel", 1)
    for steps in (1, 2, 40):
        xa = plan.mel_invert_nnls(md, n_steps=steps, power=1.0).cpu().numpy()[0].T
        assert np.all(xa >= 0)
        assert rel_l2(xa, O.nnls_accelerated(basis, m, n_steps=steps)) < 1e-4, steps
    res_a = np.linalg.norm(basis @ xa - m) / np.linalg.norm(m)
    print(f"[nnls] n_fft {n_fft}: mel residual start {res0:.3e}, 100 plain steps {res:.3e}, 40 accelerated {res_a:.3e}")
    assert res_a < res
    xat = plan.mel_invert_nnls(md, n_steps=400, power=1.0, tol=0.02).cpu().numpy()[0].T
    assert rel_l2(xat, O.nnls_accelerated(basis, m, n_steps=400, tol=0.02)) < 1e-3
    rel = np.linalg.norm(basis @ xat - m, axis=0) / np.linalg.norm(m, axis=0)
    assert np.all(rel < 0.03) and np.all(xat >= 0)
    sub = plan.mel_invert_nnls(dev(np.ascontiguousarray(m.T[None, 7:19])), n_steps=400, power=1.0, tol=0.02)
    assert np.array_equal(sub.cpu().numpy()[0].T, xat[:, 7:19])
    plan.close()
    # the secondary entry point end to end, librosa defaults (hop = n_fft/4, no peak normalisation)
    y = voc.mel_to_audio(m, sr=sr, n_fft=n_fft, n_iter=4, n_steps=50, init_phase=5)
    ref = O.mel_to_audio(m, sr=sr, n_fft=n_fft, n_iter=4, n_steps=50, random_state=5)
    assert y.shape == ref.shape == ((n_fft // 4) * (frames - 1),)
    assert rel_l2(y, ref) < 1e-3


def test_synthesize_cli_writes_the_reference_style_wav(tmp_path, lj001, O):
    """test.py:191-194 as a command: mel .npy + stats .npy -> PCM16 wav of hop*(T-1) samples."""
    from tts_amd import synthesize
    from tts_amd.wavio import read_wav
    mel, mean, std = lj001
    np.save(tmp_path / "mel.npy", mel[:200])
    np.save(tmp_path / "stats.npy", np.array([mean, std], np.float32))
    out = tmp_path / "out.wav"
    synthesize.main(["--mel", str(tmp_path / "mel.npy"), "--stats", str(tmp_path / "stats.npy"),
                     "--out", str(out), "--seed", "7"])
    audio, sr = read_wav(out)
    assert sr == 22050 and audio.shape == (256 * 199,)
    assert np.max(np.abs(audio)) > 0.9999


def test_forward_mel_extraction_and_round_trip(voc, O, lj001):
    """acoustic/dataset.py:75-84 on the device, and the loop it closes: mel -> wav -> mel."""
    mel, mean, std = lj001
    rng = np.random.default_rng(17)
    y = (rng.standard_normal(256 * 120) * np.hanning(256 * 120)).astype(np.float32)
    for n_fft in (2048, 1024):
        got = voc.audio_to_mel_db(y, n_fft=n_fft, win_length=n_fft)
        ms = O.melspectrogram(y, sr=22050, n_fft=n_fft, hop_length=256, win_length=n_fft, n_mels=80, fmax=8000)
        want = O.power_to_db(ms, ref=np.max).T
        assert got.shape == want.shape == (121, 80) and got.dtype == np.float32
        assert got.max() == pytest.approx(0.0, abs=1e-4) and got.min() >= -80.0001
        assert np.max(np.abs(got - want)) < 2e-3          # dB
        plan = voc.get_plan(n_fft, 256, n_fft, 80, 121, 1)
        mp = plan.melspectrogram(dev(y[None])).cpu().numpy()[0]
        assert rel_l2(mp, ms.T) < 1e-4
    # A signal that is not a whole number of hops long, as the reference's preprocessing feeds it
    # (whole utterances): librosa frames the zero-padded signal, so the last n_fft / (2 hop)
    # frames still see the trailing samples, and ref=np.max / top_db act on the result
    for extra in (1, 100, 255):
        yl = np.concatenate([y, (3.0 * rng.standard_normal(extra)).astype(np.float32)])
        got = voc.audio_to_mel_db(yl)
        want = O.power_to_db(O.melspectrogram(yl, sr=22050, n_fft=2048, hop_length=256, win_length=2048, n_mels=80,
                                              fmax=8000), ref=np.max).T
        assert got.shape == want.shape == (121, 80)
        assert np.max(np.abs(got - want)) < 2e-3
        plan = voc.get_plan(2048, 256, 2048, 80, 121, 1)
        d = plan.stft(dev(yl[None])).cpu().numpy()[0]
        assert rel_l2(from_btf2(d), O.stft(yl, n_fft=2048, hop_length=256, win_length=2048)) < 2e-6
    # rows of different lengths in one batch
    plan = voc.get_plan(1024, 256, 1024, 80, 121, 2)
    plan.set_mel_basis(22050, 0.0, 8000)
    yb = np.zeros((2, 256 * 120 + 77), np.float32)
    yb[0], yb[1, :9000] = rng.standard_normal(yb.shape[1]), rng.standard_normal(9000)
    mb = plan.melspectrogram(dev(yb), n_samples=[yb.shape[1], 9000]).cpu().numpy()
    for row, n in ((0, yb.shape[1]), (1, 9000)):
        ms = O.melspectrogram(yb[row, :n], sr=22050, n_fft=1024, hop_length=256, win_length=1024, n_mels=80, fmax=8000)
        assert rel_l2(mb[row, :ms.shape[1]], ms.T) < 1e-4
        assert np.all(mb[row, ms.shape[1]:] == 0)
    # Round trip mel -> wav -> mel.  (Through the reference's own function it would not close:
    # mel_audio.py feeds pinv(basis) @ POWER mel to Griffin-Lim as if it were a magnitude, so
    # the re-analysed dB range doubles.  librosa's mel_to_audio takes the square root and
    # does close, up to Griffin-Lim's inconsistency.)
    speech = voc.mel_to_audio_griffin_lim(mel[:300], mean, std, init_phase=1)
    plan = voc.get_plan(2048, 512, 2048, 80, 1 + len(speech) // 512, 1)
    plan.set_mel_basis(22050, 0.0, 8000)
    n = 512 * (len(speech) // 512)
    m0 = plan.melspectrogram(dev(speech[None, :n])).cpu().numpy()[0].T          # (80, T) power
    audio = voc.mel_to_audio(m0, sr=22050, n_fft=2048, fmax=8000, n_iter=32, n_steps=100, init_phase=2)
    assert audio.shape == (n,)
    m1 = plan.melspectrogram(dev(audio[None])).cpu().numpy()[0].T
    db0, db1 = O.power_to_db(m0, ref=np.max), O.power_to_db(m1, ref=m0.max())
    loud = db0 > -40
    assert np.mean(np.abs(db1[loud] - db0[loud])) < 3.0                          # dB


@pytest.mark.parametrize("n_fft,hop,frames", [(1024, 256, 120), (2048, 256, 96), (1024, 512, 40)])
def test_recompute_variant_matches(voc, O, lj001, n_fft, hop, frames):
    """The iteration variant that recomputes the momentum term's previous spectrum instead of
    storing it: same results as the stored-state variant (to rounding) and the oracle."""
    mel, mean, std = lj001
    s = O.mel_invert(mel[30:30 + frames], mean, std, n_fft=n_fft)
    ph = phasors(31, s.shape[0], frames)
    _, sc_ref, snaps = O.griffinlim(s, n_iter=16, hop_length=hop, win_length=n_fft, init_phase=ph,
                                    return_sc=True, snapshots=[1, 2, 3, 16])
    plan = voc.GriffinLimPlan(n_fft, hop, n_fft, 80, frames, 1)
    sd, pd = dev(to_btf(s)), dev(to_btf2(ph))
    for k in (1, 2, 3, 16):
        plan.set_option("recompute", 0)
        y0 = plan.griffinlim(sd, phase0=pd, n_iter=k).cpu().numpy()[0]
        plan.set_option("recompute", 1)
        y1, sc = plan.griffinlim(sd, phase0=pd, n_iter=k, return_sc=True)
        y1, sc = y1.cpu().numpy()[0], sc.cpu().numpy()[0]
        assert rel_l2(y1, y0) < 1e-4, k
        assert rel_l2(y1, snaps[k]) < 1e-3, k
        assert np.max(np.abs(sc - sc_ref[:k]) / sc_ref[:k]) < 1e-2
    plan.close()


def test_drop_in_44k_128_mel_variant(voc, O):
    """The 44.1 kHz / 128-mel / n_fft 2048 / hop 512 configuration (fmax = Nyquist) through the
    reference signature against the oracle with the same seeded phase."""
    from helpers import synthetic_mel
    mel = synthetic_mel(7, 150, n_mels=128)
    kw = dict(sr=44100, n_fft=2048, hop_length=512, win_length=2048, n_mels=128, fmax=None)
    y1 = voc.mel_to_audio_griffin_lim(mel, MEAN, STD, n_iter=8, init_phase=11, **kw)
    y2 = O.mel_to_audio_griffin_lim(mel, MEAN, STD, n_iter=8, random_state=11, **kw)
    assert y1.shape == y2.shape == (512 * 149,)
    assert rel_l2(y1, y2) < 1e-3
    # the inverted magnitudes alone: 1e-4
    plan = voc.get_plan(2048, 512, 2048, 128, 150, 1)
    plan.set_mel_basis(44100, 0.0, None)
    S = plan.mel_invert(dev(mel[None]), MEAN, STD)[0].cpu().numpy()
    assert rel_l2(S, O.mel_invert(mel, MEAN, STD, sr=44100, n_fft=2048, n_mels=128, fmax=None).T) < 1e-4


def test_acoustic_model_feeds_vocoder_on_device(voc):
    """Phoneme ids -> FastSpeech-style mel producer -> vocoder with the mel staying on the GPU
    equals the reference's flow (predict on one side, numpy mel handed to the vocoder,
    test.py:188-192) bit for bit, given the same initial phase."""
    from tts_amd.acoustic import FastSpeechAcousticModel, synthesize
    torch.manual_seed(0)
    model = FastSpeechAcousticModel(72).to("cuda")
    rng = np.random.default_rng(0)
    ids = np.concatenate([rng.integers(1, 72, 40), np.zeros(10, np.int64)])[None]
    durations = np.log(np.concatenate([rng.integers(2, 9, 40), np.ones(10)]).astype(np.float32))[None]
    frames = int(np.round(np.exp(durations)).sum())
    phase = 2 * np.pi * np.random.default_rng(4).random((513, frames))
    kw = dict(n_fft=1024, win_length=1024, hop_length=256, n_iter=8, init_phase=phase)
    audio, mel = synthesize(model, ids, durations, MEAN, STD, return_mel=True, **kw)
    assert audio.is_cuda and mel.is_cuda and mel.shape == (frames, 80)
    assert audio.shape == (256 * (frames - 1),)
    host = voc.mel_to_audio_griffin_lim(mel.cpu().numpy(), MEAN, STD, **kw)
    assert np.array_equal(audio.cpu().numpy(), host)
    # a random-init model emits a very quiet mel, so the 1e-6 of the peak normalisation shows
    assert np.isfinite(host).all() and 0.9 < np.abs(host).max() <= 1.0


def test_drop_in_with_nnls_inversion(voc, O, lj001):
    """inversion="nnls": a non-negative spectrogram whose mel residual is below that of the
    clipped pinv product it starts from, then the same Griffin-Lim."""
    mel, mean, std = lj001
    mel = mel[:96]
    plan = voc.get_plan(1024, 256, 1024, 80, 96, 1)
    plan.set_mel_basis(22050, 0.0, 8000)
    from tts_amd.vocoder import _invert_mel
    S = _invert_mel(plan, dev(mel[None]), mean, std, "nnls", 40)[0].cpu().numpy().astype(np.float64)
    assert S.min() >= 0.0
    basis = O.mel_filterbank(22050, 1024, n_mels=80, fmax=8000).astype(np.float64)
    target = O.db_to_power((mel * std + mean).T).astype(np.float64)
    start = np.clip(O.mel_invert(mel, mean, std, n_fft=1024), 0, None)
    res = np.linalg.norm(basis @ S.T - target) / np.linalg.norm(target)
    res0 = np.linalg.norm(basis @ start - target) / np.linalg.norm(target)
    assert res < res0, (res, res0)
    kw = dict(n_fft=1024, win_length=1024, n_iter=8, init_phase=5)
    y = voc.mel_to_audio_griffin_lim(mel, mean, std, inversion="nnls", **kw)
    y2 = O.griffinlim(S.T.astype(np.float32), n_iter=8, hop_length=256, win_length=1024, n_fft=1024,
                      random_state=5)
    assert rel_l2(y, O.peak_normalise(y2)) < 1e-3
    with pytest.raises(ValueError):
        voc.mel_to_audio_griffin_lim(mel, mean, std, inversion="lbfgs")


@pytest.mark.parametrize("n_fft,hop", [(1024, 256), (2048, 256), (2048, 512)])
def test_single_call_equals_three_calls(voc, n_fft, hop):
    """glv_mel_to_audio (magnitudes written straight into the iteration state, peak gathered
    while the waveform is written) against mel_invert + griffinlim + peak_normalize: the same
    bits, for given and for device-drawn phases, ragged lengths and odd n_mels."""
    from helpers import synthetic_mel, phasors
    lens = [70, 33, 64]
    for n_mels in (80, 37):
        mel = np.zeros((3, 70, n_mels), np.float32)
        for b, t in enumerate(lens):
            mel[b, :t] = synthetic_mel(20 + b, t, n_mels=n_mels)
        plan = voc.GriffinLimPlan(n_fft, hop, n_fft, n_mels, 70, 3)
        plan.set_mel_basis(22050, 0.0, 8000)
        md = dev(mel)
        ph = np.stack([np.stack([phasors(5 + b, n_fft // 2 + 1, 70).T.real,
                                 phasors(5 + b, n_fft // 2 + 1, 70).T.imag], -1) for b in range(3)])
        for phase0, seed in ((dev(np.ascontiguousarray(ph, dtype=np.float32)), 0), (None, 77)):
            S = plan.mel_invert(md, MEAN, STD)
            y3, sc3 = plan.griffinlim(S, lens, phase0=phase0, seed=seed, n_iter=5, return_sc=True)
            n_samples = [hop * (t - 1) for t in lens]
            raw3 = y3.clone()
            plan.peak_normalize(y3, n_samples)
            y1, sc1 = plan.mel_to_audio(md, MEAN, STD, lens, phase0=phase0, seed=seed, n_iter=5, return_sc=True)
            assert torch.equal(y1, y3)
            assert torch.equal(sc1, sc3)
            raw1 = plan.mel_to_audio(md, MEAN, STD, lens, phase0=phase0, seed=seed, n_iter=5, peak_eps=None)
            assert torch.equal(raw1, raw3)
        plan.close()


def test_long_form_single_utterance(voc, O):
    """A 10-minute mel (51 680 frames, n_fft 1024) on one GPU: size-independent properties, and
    the middle of the waveform against the oracle run on a window of frames around it (frames
    further than the iteration count times the overlap cannot influence it)."""
    frames, n_fft, hop, k = 51680, 1024, 256, 4
    mel = np.concatenate([synthetic_mel(300 + i, 430) for i in range(121)])[:frames]
    assert mel.shape == (frames, 80)
    rng = np.random.default_rng(8)
    phase = np.exp(2j * np.pi * rng.random((n_fft // 2 + 1, frames))).astype(np.complex64)
    plan = voc.get_plan(n_fft, hop, n_fft, 80, frames, 1)
    plan.set_mel_basis(22050, 0.0, 8000)
    p0 = dev(np.ascontiguousarray(np.stack([phase.T.real, phase.T.imag], -1)[None], dtype=np.float32))
    y, sc = plan.mel_to_audio(dev(mel[None]), MEAN, STD, [frames], phase0=p0, n_iter=k, peak_eps=None,
                              return_sc=True)
    y, sc = y[0].cpu().numpy(), sc[0].cpu().numpy()
    assert y.shape == (hop * (frames - 1),) and np.all(np.isfinite(y))
    assert np.all(np.diff(sc) < 0)
    # window of frames [t0, t1) -> oracle; compare the centre where both see the same neighbours
    t0, t1 = 25000, 25200
    S = O.mel_invert(mel[t0:t1], MEAN, STD, n_fft=n_fft)
    ref = O.griffinlim(S, n_iter=k, hop_length=hop, win_length=n_fft, n_fft=n_fft, init_phase=phase[:, t0:t1])
    margin = (k + 2) * (n_fft // hop)          # frames whose samples may differ at either end
    a, b = margin * hop, (t1 - t0 - 1 - margin) * hop
    got = y[t0 * hop + a: t0 * hop + b]
    assert rel_l2(got, ref[a:b]) < 1e-3


@pytest.mark.parametrize("n_fft,hop,win,lens", [(1024, 256, 1024, [33, 2, 64, 7]), (2048, 256, 2048, [40, 3, 17]),
                                                (2048, 512, 1200, [21, 6]), (1024, 128, 1024, [30, 5]),
                                                (2048, 1024, 2048, [9, 2]), (1024, 512, 800, [12, 3])])
def test_one_frame_per_wave_mode(voc, O, n_fft, hop, win, lens):
    """The small-batch form of the iteration (option single_frame: every frame a wavefront of
    its own, partial-sum buffers instead of a running sum) against the oracle and against the
    run-based form, waveform and convergence curve, on ragged batches down to two frames."""
    f = n_fft // 2 + 1
    t_max = max(lens)
    rng = np.random.default_rng(17 + t_max)
    s = np.zeros((len(lens), t_max, f), np.float32)
    p = np.zeros((len(lens), t_max, f, 2), np.float32)
    for b, t in enumerate(lens):
        s[b, :t] = rng.standard_normal((t, f)) ** 2
        s[b, :t, ::7] *= -0.3
        ang = 2 * np.pi * rng.random((t, f))
        p[b, :t, :, 0], p[b, :t, :, 1] = np.cos(ang), np.sin(ang)
    plan = voc.GriffinLimPlan(n_fft, hop, win, 80, t_max, len(lens))
    out = {}
    for mode in (1, 0):
        plan.set_option("single_frame", mode)
        for graph in (1, 0):
            plan.set_option("use_graph", graph)
            y, sc = plan.griffinlim(dev(s), lens, phase0=dev(p), n_iter=6, return_sc=True)
            out[mode, graph] = (y.cpu().numpy(), sc.cpu().numpy())
        assert np.array_equal(out[mode, 0][0], out[mode, 1][0])
    y1, sc1 = out[1, 1]
    y0, sc0 = out[0, 1]
    for b, t in enumerate(lens):
        n = hop * (t - 1)
        ref, sc_ref = O.griffinlim(np.ascontiguousarray(s[b, :t].T), n_iter=6, hop_length=hop, win_length=win,
                                   n_fft=n_fft, init_phase=from_btf2(p[b, :t]), return_sc=True)
        assert rel_l2(y1[b, :n], ref) < 1e-4
        assert rel_l2(y1[b, :n], y0[b, :n]) < 1e-4
        assert np.all(y1[b, n:] == 0)
        assert np.max(np.abs(sc1[b] - sc_ref) / sc_ref) < 1e-4
    # a second call with other lengths reuses the buffers (they hold the previous call's sums)
    lens2 = [max(2, t // 2) for t in lens]
    plan.set_option("single_frame", 1)
    y2 = plan.griffinlim(dev(s), lens2, phase0=dev(p), n_iter=2).cpu().numpy()
    for b, t in enumerate(lens2):
        n = hop * (t - 1)
        ref = O.griffinlim(np.ascontiguousarray(s[b, :t].T), n_iter=2, hop_length=hop, win_length=win,
                           n_fft=n_fft, init_phase=from_btf2(p[b, :t]))
        assert rel_l2(y2[b, :n], ref) < 1e-4
    plan.close()


@pytest.mark.parametrize("n_fft,hop,win,lens,fpr", [(1024, 256, 1024, [120, 33, 7], 12), (2048, 256, 2048, [96, 40], 12),
                                                    (2048, 512, 1200, [64], 7), (1024, 128, 1024, [50, 9], 0),
                                                    (2048, 1024, 2048, [12, 5], 0), (1024, 512, 800, [40], 3)])
def test_momentum_in_time_domain_and_in_frequency_domain(voc, O, n_fft, hop, win, lens, fpr):
    """The run-based iteration in its two forms - momentum term formed before the forward
    transform (option linear = 1, the default: no spectrum state in memory) or by subtracting
    the stored previous spectrum (linear = 0) - against the oracle and against each other."""
    f = n_fft // 2 + 1
    t_max = max(lens)
    rng = np.random.default_rng(3 + t_max)
    s = np.zeros((len(lens), t_max, f), np.float32)
    p = np.zeros((len(lens), t_max, f, 2), np.float32)
    for b, t in enumerate(lens):
        s[b, :t] = rng.standard_normal((t, f)) ** 2
        s[b, :t, ::7] *= -0.3
        ang = 2 * np.pi * rng.random((t, f))
        p[b, :t, :, 0], p[b, :t, :, 1] = np.cos(ang), np.sin(ang)
    plan = voc.GriffinLimPlan(n_fft, hop, win, 80, t_max, len(lens))
    plan.set_option("single_frame", 0)
    plan.set_option("frames_per_run", fpr)
    out = {}
    for linear in (1, 0):
        plan.set_option("linear", linear)
        y, sc = plan.griffinlim(dev(s), lens, phase0=dev(p), n_iter=8, return_sc=True)
        y_nosc = plan.griffinlim(dev(s), lens, phase0=dev(p), n_iter=8)
        assert torch.equal(y, y_nosc)          # asking for the curve does not change the waveform
        out[linear] = (y.cpu().numpy(), sc.cpu().numpy())
    for b, t in enumerate(lens):
        n = hop * (t - 1)
        ref, sc_ref = O.griffinlim(np.ascontiguousarray(s[b, :t].T), n_iter=8, hop_length=hop, win_length=win,
                                   n_fft=n_fft, init_phase=from_btf2(p[b, :t]), return_sc=True)
        for linear in (1, 0):
            y, sc = out[linear]
            assert rel_l2(y[b, :n], ref) < 1e-4, linear
            assert np.all(y[b, n:] == 0)
            assert np.max(np.abs(sc[b] - sc_ref) / sc_ref) < 1e-4, linear
    plan.close()


@pytest.mark.parametrize("n_fft,hop,batch,frames", [(1024, 256, 48, 160), (2048, 256, 24, 160)])
def test_device_filling_batch_with_automatic_runs(voc, O, n_fft, hop, batch, frames):
    """A batch with more frames than the device has wave slots, library defaults throughout
    (automatic run table, several frames per wave, momentum in the time domain, graph replay):
    sampled utterances match the oracle, and the batch result does not depend on the batch."""
    f = n_fft // 2 + 1
    rng = np.random.default_rng(17)
    lens = [int(x) for x in rng.integers(frames // 2, frames + 1, batch)]
    lens[0], lens[-1] = frames, frames // 2
    s = np.zeros((batch, frames, f), np.float32)
    p = np.zeros((batch, frames, f, 2), np.float32)
    for b, t in enumerate(lens):
        s[b, :t] = rng.standard_normal((t, f)) ** 2
        ang = 2 * np.pi * rng.random((t, f))
        p[b, :t, :, 0], p[b, :t, :, 1] = np.cos(ang), np.sin(ang)
    plan = voc.GriffinLimPlan(n_fft, hop, n_fft, 80, frames, batch)
    yb = plan.griffinlim(dev(s), lens, phase0=dev(p), n_iter=6).cpu().numpy()
    assert np.all(np.isfinite(yb))
    for b in (0, batch // 2, batch - 1):
        t = lens[b]
        n = hop * (t - 1)
        assert np.all(yb[b, n:] == 0)
        ref = O.griffinlim(np.ascontiguousarray(s[b, :t].T), n_iter=6, hop_length=hop, win_length=n_fft,
                           init_phase=from_btf2(p[b, :t]))
        assert rel_l2(yb[b, :n], ref) < 1e-4
    # the same utterances in another order and company: only run boundaries move
    order = list(range(batch))[::-1]
    yr = plan.griffinlim(dev(s[order]), [lens[i] for i in order], phase0=dev(p[order]), n_iter=6).cpu().numpy()
    for b in (0, batch // 2, batch - 1):
        n = hop * (lens[b] - 1)
        assert rel_l2(yr[batch - 1 - b, :n], yb[b, :n]) < 1e-5
    plan.close()


@pytest.mark.parametrize("n_fft,hop,frames,world,backend", [(1024, 256, 150, 2, "gloo"), (2048, 256, 200, 3, "gloo"),
                                                            (1024, 256, 90, 1, "nccl")])
def test_time_sharded_two_processes(n_fft, hop, frames, world, backend):
    """`griffin_lim_time_sharded` as it is deployed: one process per shard, halos and the peak
    exchanged through torch.distributed (gloo with all ranks sharing the one GPU; RCCL with the
    single rank a one-GPU box allows).  Rank 0
    compares the assembled waveform with the unsharded drop-in call on the same seed."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    # several processes share the one GPU here: no launch that needs the device to itself
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", GLVOC_PERSISTENT="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(29600 + world),
           os.path.join(here, "time_shard_worker.py"), str(n_fft), str(hop), str(frames), backend,
           # the swap issued by the library needs RCCL and a GPU per rank: the one-rank case
           "library" if backend == "nccl" else "torch"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=110)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "rel L2" in res.stdout


@pytest.mark.parametrize("backend,world", [("gloo", 3), ("nccl", 1)])
def test_batch_sharded_processes(backend, world):
    """`synthesize_sharded` as it is deployed: one process per rank, utterances dealt by length,
    waveforms gathered on rank 0 - over gloo with three processes on the one GPU, and over RCCL
    with the one rank a one-GPU box allows (device-resident collectives).  Rank 0 compares with
    the same utterances synthesised in one call."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    # several processes share the one GPU here: no launch that needs the device to itself
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", GLVOC_PERSISTENT="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(29650 + world),
           os.path.join(here, "batch_shard_worker.py"), backend]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=110)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "worst rel L2" in res.stdout


def test_alternating_shapes_without_draining_the_stream(voc):
    """A serving loop that alternates between batches of different lengths and iteration counts:
    the run tables go up from pinned staging sets in rotation and the captured graphs of several
    shapes stay cached, with no host wait in between - every repetition of a shape must give the
    bits of its first occurrence, also when the calls are queued back to back."""
    n_fft, hop = 1024, 256
    f = n_fft // 2 + 1
    rng = np.random.default_rng(41)
    shapes = [([64, 40, 21], 5), ([30, 64], 3), ([12, 12, 12, 50], 5), ([64], 4), ([33, 7, 64, 20], 2)]
    plan = voc.GriffinLimPlan(n_fft, hop, n_fft, 80, 64, 4)
    plan.set_option("single_frame", 0)
    cases = []
    for lens, k in shapes:
        s = np.zeros((len(lens), 64, f), np.float32)
        p = np.zeros((len(lens), 64, f, 2), np.float32)
        for b, t in enumerate(lens):
            s[b, :t] = rng.standard_normal((t, f)) ** 2
            ang = 2 * np.pi * rng.random((t, f))
            p[b, :t, :, 0], p[b, :t, :, 1] = np.cos(ang), np.sin(ang)
        cases.append((lens, k, dev(s), dev(p)))
    first = [plan.griffinlim(sd, lens, phase0=pd, n_iter=k).clone() for lens, k, sd, pd in cases]
    torch.cuda.synchronize()
    outs = []
    for rep in range(4):                      # 20 calls queued without a synchronisation
        for lens, k, sd, pd in cases:
            outs.append(plan.griffinlim(sd, lens, phase0=pd, n_iter=k))
    torch.cuda.synchronize()
    for i, y in enumerate(outs):
        assert torch.equal(y, first[i % len(cases)]), i
    plan.close()


def test_reserved_workspace_covers_forced_modes(voc, O):
    """glv_plan_reserve sets the partial-sum buffers of the one-frame-per-wave mode and the
    partials of the convergence curve aside for a shape beyond what the automatic mode choice
    uses; the calls that follow match the oracle."""
    n_fft, hop, frames, batch = 1024, 256, 200, 24       # 4800 frames: automatic choice would use runs
    f = n_fft // 2 + 1
    rng = np.random.default_rng(43)
    s = (rng.standard_normal((batch, frames, f)) ** 2).astype(np.float32)
    ang = 2 * np.pi * rng.random((batch, frames, f))
    p = np.stack([np.cos(ang), np.sin(ang)], -1).astype(np.float32)
    plan = voc.GriffinLimPlan(n_fft, hop, n_fft, 80, frames, batch)
    before = plan.workspace_bytes
    plan.reserve(batch, frames, n_iter=100, single_frame=True)
    assert plan.workspace_bytes > before
    grown = plan.workspace_bytes
    plan.set_option("single_frame", 1)
    y, sc = plan.griffinlim(dev(s), phase0=dev(p), n_iter=100 // 10, return_sc=True)
    assert plan.workspace_bytes == grown                 # nothing had to grow inside the call
    for b in (0, batch - 1):
        ref, sc_ref = O.griffinlim(np.ascontiguousarray(s[b].T), n_iter=10, hop_length=hop, win_length=n_fft,
                                   init_phase=from_btf2(p[b]), return_sc=True)
        assert rel_l2(y[b].cpu().numpy(), ref) < 1e-4
        assert np.max(np.abs(sc[b].cpu().numpy() - sc_ref) / sc_ref) < 1e-4
    with pytest.raises(Exception):
        plan.reserve(batch + 1, frames)
    plan.close()


@pytest.mark.parametrize("n_fft,hop,lens,ids", [(1024, 256, [64, 40, 21, 2, 9], None), (1024, 256, [33, 64], [7, 123456789012]),
                                                (2048, 256, [48, 30, 5], None), (2048, 512, [40, 17], [3, 1]),
                                                (1024, 128, [50, 12], None)])
def test_phase_drawn_inside_the_first_inverse_equals_stored_spectrum(voc, n_fft, hop, lens, ids):
    """With no phase given, the run-based iteration that keeps no spectrum lets the first inverse
    transform draw the initial phase itself (nothing stored); switched off, the set-up kernels
    store S * phase and that kernel reads it back.  Same products in the same order: the
    waveforms must agree bit for bit - from mel input (pair order without the bins that are zero
    for good) and from caller magnitudes (pair order, all bins), and also after 0 iterations,
    where the output is the inverse of the initial spectrum itself."""
    f = n_fft // 2 + 1
    t_max, b = max(lens), len(lens)
    rng = np.random.default_rng(47)
    mel = np.stack([np.pad(synthetic_mel(500 + i, t), ((0, t_max - t), (0, 0))) for i, t in enumerate(lens)])
    s = np.zeros((b, t_max, f), np.float32)
    for i, t in enumerate(lens):
        s[i, :t] = rng.standard_normal((t, f)) ** 2
    plan = voc.GriffinLimPlan(n_fft, hop, n_fft, 80, t_max, b)
    plan.set_mel_basis(22050, 0.0, 8000)
    plan.set_option("single_frame", 0)
    plan.set_utterance_ids(ids)
    got = {}
    for fuse in (1, 0):
        plan.set_option("fuse_draw", fuse)
        for k in (0, 3):
            got[fuse, k, "mel"] = plan.mel_to_audio(dev(mel), MEAN, STD, lens, seed=99, n_iter=k).clone()
            got[fuse, k, "s"] = plan.griffinlim(dev(s), lens, seed=5, n_iter=k).clone()
    for k in (0, 3):
        for src in ("mel", "s"):
            y1, y0 = got[1, k, src], got[0, k, src]
            assert float(y1.abs().max()) > 0
            assert torch.equal(y1, y0), (k, src)
    plan.close()


@pytest.mark.parametrize("n_fft,hop", [(1024, 256), (2048, 256)])
def test_library_halo_exchange_over_rccl_to_itself(voc, n_fft, hop):
    """The swap issued by the library (glv_halo_attach / glv_shard_exchange: RCCL send / recv on
    the plan's stream).  One GPU allows one rank, so the rank is made its own neighbour on both
    sides ("halo_loopback"): every side then receives what it sent, which the same shard gets
    from glv_shard_halo_pack / glv_shard_halo_unpack with the packed buffers handed back.  Same
    kernels, same data: the two waveforms must be identical - which checks the regions, the
    sizes, the pairing of sends and receives and their ordering against the steps on the stream.
    Also: exchanging without a communicator attached, and a shard with a neighbour on a side
    where the rank has none, are refused."""
    frames, t0, t1, n_iter = 160, 40, 110, 5
    f = n_fft // 2 + 1
    rng = np.random.default_rng(53)
    s = dev((rng.standard_normal((t1 - t0, f)) ** 2).astype(np.float32))

    def run(plan, swap):
        plan.set_option("frames_per_run", 9)
        plan.shard_begin(s, t0, frames, seed=11)
        swap(plan)
        for _ in range(n_iter):
            plan.shard_step(0.99)
            swap(plan)
        y, n_begin = plan.shard_end()
        return y.clone(), n_begin

    def swap_by_copies(plan):
        to_prev, to_next = plan.shard_halo_pack()
        plan.shard_halo_unpack(to_prev, to_next)

    ref_plan = voc.GriffinLimPlan(n_fft, hop, n_fft, 80, t1 - t0, 1)
    y_ref, b_ref = run(ref_plan, swap_by_copies)
    with pytest.raises(Exception):
        ref_plan.shard_begin(s, t0, frames, seed=11)
        ref_plan.shard_exchange()                         # not attached
    ref_plan.close()

    chain = voc.HaloChain(voc.HaloChain.unique_id(), 0, 1)
    plan = voc.GriffinLimPlan(n_fft, hop, n_fft, 80, t1 - t0, 1)
    plan.halo_attach(chain)
    plan.shard_begin(s, t0, frames, seed=11)
    with pytest.raises(Exception):
        plan.shard_exchange()                             # rank 0 of 1 has no neighbours, the shard has two
    plan.set_option("halo_loopback", 1)
    y, b = run(plan, lambda p: p.shard_exchange())
    # the same communicator serves a second plan
    other = voc.GriffinLimPlan(n_fft, hop, n_fft, 80, t1 - t0, 1)
    other.halo_attach(chain)
    other.set_option("halo_loopback", 1)
    other.set_option("frames_per_run", 9)
    other.shard_begin(s, t0, frames, seed=11)
    other.shard_run(n_iter, 0.99)                         # the whole loop enqueued by one call
    y2 = other.shard_end()[0].clone()
    torch.cuda.synchronize()
    assert b == b_ref and y.shape == y_ref.shape and float(y.abs().max()) > 0
    assert torch.equal(y, y_ref) and torch.equal(y2, y_ref)
    plan.halo_attach(None)
    with pytest.raises(Exception):
        plan.shard_begin(s, t0, frames, seed=11)
        plan.shard_exchange()                             # detached
    with pytest.raises(Exception):
        voc.HaloChain(b"short", 0, 1)
    with pytest.raises(Exception):
        voc.HaloChain(voc.HaloChain.unique_id(), 1, 1)    # rank outside [0, world)
    other.close()
    plan.close()
    chain.close()


@pytest.mark.parametrize("n_fft,hop", [(1024, 256), (2048, 256)])
def test_large_batch_cut_into_groups_of_utterances(voc, n_fft, hop):
    """A batch with more runs than one iteration-spanning launch holds is cut into groups of
    consecutive utterances, each run as such a launch with the run table that a call of those
    utterances alone gets (option "group_frames" sets the size, here small enough to cut a small
    batch).  Every group's rows must therefore equal, bit for bit, what a call of just that group
    returns, and the whole must agree with the uncut batch (other run boundaries: a few sums
    associate differently) to rounding - bit for bit when the run length is fixed."""
    lens = [90, 64, 120, 33, 77, 101, 58, 96, 45, 110, 70]
    b, t_max, f = len(lens), max(lens), n_fft // 2 + 1
    rng = np.random.default_rng(67)
    s = np.zeros((b, t_max, f), np.float32)
    ph = np.zeros((b, t_max, f, 2), np.float32)
    for i, t in enumerate(lens):
        s[i, :t] = rng.standard_normal((t, f)) ** 2
        ang = rng.uniform(0, 2 * np.pi, (t, f))
        ph[i, :t, :, 0], ph[i, :t, :, 1] = np.cos(ang), np.sin(ang)
    plan = voc.GriffinLimPlan(n_fft, hop, n_fft, 80, t_max, b)
    if plan.persistent_capacity(0) <= 0:
        plan.close()
        pytest.skip("no iteration-spanning launches on this device")
    plan.set_option("single_frame", 0)     # (so small a batch would get a wave per frame otherwise)
    plan.set_option("group_frames", -1)
    whole = plan.griffinlim(dev(s), lens, phase0=dev(ph), n_iter=8).clone()
    assert plan.last_loop_groups() == 1
    plan.set_option("group_frames", 300)
    cut = plan.griffinlim(dev(s), lens, phase0=dev(ph), n_iter=8).clone()
    groups = plan.last_loop_groups()
    assert groups == 3 and plan.last_loop_launches() == 1
    plan.status()
    for i, t in enumerate(lens):
        n = hop * (t - 1)
        err = float((cut[i, :n] - whole[i, :n]).norm() / whole[i, :n].norm())
        assert err < 2e-5, (i, err)
    # the groups as calls of their own: the same cut of the list by frames (thirds of 864)
    bounds, acc, total, g = [0], 0, sum(lens), 1
    for i, t in enumerate(lens):
        if g < groups and acc + t // 2 > total * g // groups and i > bounds[-1]:
            bounds.append(i)
            g += 1
        acc += t
    bounds.append(b)
    assert len(bounds) == groups + 1
    plan.set_option("group_frames", -1)
    for b0, b1 in zip(bounds[:-1], bounds[1:]):
        part = plan.griffinlim(dev(s[b0:b1]), lens[b0:b1], phase0=dev(ph[b0:b1]), n_iter=8)
        for i in range(b0, b1):
            n = hop * (lens[i] - 1)
            assert torch.equal(part[i - b0, :n], cut[i, :n]), (b0, b1, i)
    # with a fixed run length an utterance's runs do not depend on its group: cut or not, same bits
    plan.set_option("frames_per_run", 2 * (n_fft // hop))
    uncut = plan.griffinlim(dev(s), lens, phase0=dev(ph), n_iter=8).clone()
    assert plan.last_loop_groups() == 1
    plan.set_option("group_frames", 300)
    cut = plan.griffinlim(dev(s), lens, phase0=dev(ph), n_iter=8)
    assert plan.last_loop_groups() == 3
    assert torch.equal(cut, uncut)
    plan.status()
    plan.close()


@pytest.mark.parametrize("n_fft,hop", [(1024, 256), (2048, 256)])
def test_shard_run_replayed_from_one_graph(voc, n_fft, hop):
    """glv_shard_run enqueues the steps and swaps after the first swap one by one (option
    "shard_graph" 0), replays them from one hipGraph (1), or from a graph with two branches in
    which the workgroups at the shard's edges, and the swap, overlap the workgroups between them
    (2); -1 chooses between the last two by the shard's size.  The same launches on the same data in
    an order that keeps every dependence: the waveforms must be identical bit for bit, with the
    swap as device copies too; a replay, a shard at another place of the utterance, another
    iteration count and another momentum must each give what the enqueued form gives."""
    frames, f = 1400, n_fft // 2 + 1
    rng = np.random.default_rng(61)
    s_all = (rng.standard_normal((frames, f)) ** 2).astype(np.float32)
    chain = voc.HaloChain(voc.HaloChain.unique_id(), 0, 1)
    plan = voc.GriffinLimPlan(n_fft, hop, n_fft, 80, 500, 1)
    plan.halo_attach(chain)
    plan.set_option("halo_loopback", 1)

    def run(form, t0, t1, n_iter, momentum):
        plan.set_option("shard_graph", form)
        plan.shard_begin(dev(s_all[t0:t1]), t0, frames, seed=17)
        plan.shard_run(n_iter, momentum)
        taken = plan.shard_last_run_form()
        y, n_begin = plan.shard_end()
        return y.clone(), n_begin, taken

    def by_copies(t0, t1, n_iter, momentum):
        plan.shard_begin(dev(s_all[t0:t1]), t0, frames, seed=17)
        for k in range(n_iter + 1):
            if k:
                plan.shard_step(momentum)
            plan.shard_halo_unpack(*plan.shard_halo_pack())
        return plan.shard_end()[0].clone()

    for t0, t1, n_iter, momentum in [(300, 800, 6, 0.99), (300, 800, 6, 0.99), (450, 930, 6, 0.99),
                                     (450, 930, 3, 0.99), (450, 930, 3, 0.5)]:
        y0, b0, form0 = run(0, t0, t1, n_iter, momentum)
        assert form0 == 0 and float(y0.abs().max()) > 0
        assert torch.equal(by_copies(t0, t1, n_iter, momentum), y0)
        for form in (1, 2, -1):
            y, b, taken = run(form, t0, t1, n_iter, momentum)
            assert taken == (form if form > 0 else 1), (form, taken)   # (a small shard: one branch)
            assert b == b0 and torch.equal(y, y0), (form, t0, n_iter, momentum)
    # two calls on one begin: the second continues where the first ended, from a graph of its own
    plan.set_option("shard_graph", 0)
    plan.shard_begin(dev(s_all[300:800]), 300, frames, seed=17)
    plan.shard_run(6, 0.99)
    y_six = plan.shard_end()[0].clone()
    plan.set_option("shard_graph", 2)
    plan.shard_begin(dev(s_all[300:800]), 300, frames, seed=17)
    plan.shard_run(2, 0.99)
    plan.shard_step(0.99)
    plan.shard_exchange()
    plan.shard_run(3, 0.99)
    assert torch.equal(plan.shard_end()[0], y_six)
    torch.cuda.synchronize()
    plan.close()
    chain.close()


@pytest.mark.parametrize("n_fft,hop,win,lens,fpr", [(1024, 256, 1024, [64, 40, 21, 2, 9], 0), (1024, 256, 1024, [150, 33], 5),
                                                    (1024, 128, 1024, [50, 12, 64], 9), (1024, 512, 1024, [40, 17], 0),
                                                    (1024, 256, 800, [70, 31, 6], 3), (2048, 256, 2048, [48, 30], 0)])
def test_run_boundaries_completed_by_the_producer(voc, n_fft, hop, win, lens, fpr):
    """Option "deposit": where consecutive runs of an utterance are waves of one workgroup the later
    run hands its head sums over through LDS and the earlier one writes the boundary blocks
    complete, instead of both writing halves that every reader adds.  The same additions in the
    same order: waveforms must be identical bit for bit with the option on and off - from mel
    input and from caller magnitudes, with explicit and device-drawn phases, after 0 and several
    iterations - and the convergence curve (whose kernels keep the two-halves form) must come with
    the same waveform.  (n_fft 2048 has no such kernels: the option changes nothing.)"""
    f = n_fft // 2 + 1
    t_max, b = max(lens), len(lens)
    rng = np.random.default_rng(59)
    mel = np.stack([np.pad(synthetic_mel(600 + i, t), ((0, t_max - t), (0, 0))) for i, t in enumerate(lens)])
    s = np.zeros((b, t_max, f), np.float32)
    p0 = np.zeros((b, t_max, f, 2), np.float32)
    for i, t in enumerate(lens):
        s[i, :t] = rng.standard_normal((t, f)) ** 2
        ang = 2 * np.pi * rng.random((t, f))
        p0[i, :t, :, 0], p0[i, :t, :, 1] = np.cos(ang), np.sin(ang)
    plan = voc.GriffinLimPlan(n_fft, hop, win, 80, t_max, b)
    plan.set_mel_basis(22050, 0.0, 8000)
    plan.set_option("single_frame", 0)
    plan.set_option("frames_per_run", fpr)
    got = {}
    for dep in (1, 0):
        plan.set_option("deposit", dep)
        for k in (0, 1, 6):
            got[dep, k, "mel"] = plan.mel_to_audio(dev(mel), MEAN, STD, lens, seed=3, n_iter=k).clone()
            got[dep, k, "s"] = plan.griffinlim(dev(s), lens, phase0=dev(p0), n_iter=k).clone()
        y_sc, sc = plan.griffinlim(dev(s), lens, phase0=dev(p0), n_iter=6, return_sc=True)
        got[dep, 6, "sc"] = y_sc.clone()
    for k in (0, 1, 6):
        for src in ("mel", "s"):
            assert float(got[1, k, src].abs().max()) > 0
            assert torch.equal(got[1, k, src], got[0, k, src]), (k, src)
    assert torch.equal(got[1, 6, "sc"], got[1, 6, "s"]) and torch.equal(got[0, 6, "sc"], got[0, 6, "s"])
    plan.close()


def test_benchmark_call_against_the_oracle(voc, O):
    """The exact call bench.py times - BASELINE.json configs[1]: 64 synthetic utterances x 430
    frames, n_fft 1024, one glv_mel_to_audio with no phase given (drawn on the device inside the
    first inverse transform), bins that are zero for good left out, run boundaries completed by the
    producer, the library's own run table, graph replay, peak normalisation - against the oracle
    ELEMENT BY ELEMENT: the phasors the device draws for (seed, row) are read back
    (glv_plan_drawn_phase) and handed to the oracle as its init_phase.  Rows 0, 31 and 63 after 16,
    32 and 60 iterations (the reference's count, vocoder/mel_audio.py:126): waveform within 1e-3
    relative L2 up to k = 32 and, at k = 60, within three times what the oracle itself moves
    between fp32 and fp64 on the same input if that is more (these smoothed-noise mels amplify
    rounding far more than speech does)."""
    b, frames, n_fft, hop, seed = 64, 430, 1024, 256, 1000
    f = n_fft // 2 + 1
    mels = np.stack([synthetic_mel(1234 + i, frames) for i in range(b)])       # bench.py's make_batch(1234, ...)
    plan = voc.GriffinLimPlan(n_fft, hop, n_fft, 80, frames, b)
    plan.set_mel_basis(22050, 0.0, 8000)
    mel_d = dev(mels)
    phase = plan.drawn_phase(seed, b, frames)
    assert phase.shape == (b, frames, f, 2)
    mod = (phase[..., 0] ** 2 + phase[..., 1] ** 2).sqrt()
    assert float((mod - 1).abs().max()) < 1e-5
    ph_host = phase.cpu().numpy()
    rows, ks = (0, 31, 63), (16, 32, 60)
    got = {}
    for k in ks:
        y = plan.mel_to_audio(mel_d, MEAN, STD, [frames] * b, phase0=None, seed=seed, n_iter=k, momentum=0.99)
        got[k] = y[list(rows)].cpu().numpy()
        if k == 16:
            # the same call with the drawn phasors handed in explicitly: the same products in the
            # same order, so the same bits - the read-back phase IS the phase the call used
            y_explicit = plan.mel_to_audio(mel_d, MEAN, STD, [frames] * b, phase0=phase, seed=0, n_iter=k,
                                           momentum=0.99)
            assert torch.equal(y_explicit, y)
    for i, row in enumerate(rows):
        ph = from_btf2(ph_host[row])
        s = O.mel_invert(mels[row], MEAN, STD, n_fft=n_fft)
        _, snaps = O.griffinlim(s, n_iter=60, hop_length=hop, win_length=n_fft, init_phase=ph, snapshots=list(ks))
        _, snaps64 = O.griffinlim(s.astype(np.float64), n_iter=60, hop_length=hop, win_length=n_fft,
                                  init_phase=ph.astype(np.complex128), snapshots=list(ks))
        for k in ks:
            want = O.peak_normalise(snaps[k])
            d = rel_l2(got[k][i], want)
            floor = rel_l2(snaps[k], snaps64[k])
            print(f"[bench call] row {row} k={k}: rel L2 vs oracle {d:.2e} (oracle fp32 vs fp64: {floor:.2e})")
            assert got[k][i].shape == want.shape
            assert d < max(1e-3, 3 * floor), (row, k, d, floor)
            assert k > 32 or d < 1e-3, (row, k, d)
    plan.close()


def test_long_form_time_sharded_at_full_size(voc, O, lj001):
    """BASELINE.json configs[3] at its real size on one GPU: a 10-minute mel (51 680 frames, n_fft
    1024, hop 256: the reference's speech mel repeated) cut into 8 time shards of 6 460 frames -
    8 plans side by side, the per-iteration halo swap as device copies - for 60 iterations with
    momentum 0.99 (vocoder/mel_audio.py:126), against the unsharded 51 680-frame run on the same
    device-drawn phase after 16, 32 and 60 iterations.  A shard boundary re-associates the sums of
    R - 1 blocks; the momentum iteration amplifies such rounding differences, so the drift is printed
    beside the oracle's own fp32-vs-fp64 difference on a window of the same input and asserted
    below 1e-3 (beyond k = 32: below three times that difference if it is more).  The oracle itself
    is compared on a 400-frame window across a shard boundary at k = 16 (frames further than
    (k + 2) R from a sample cannot have influenced it).  Last, the swap as the library issues it
    (glv_shard_run: RCCL send / recv to itself, "halo_loopback") over 60 iterations at the
    per-rank size, bit for bit against the same swap done by copies."""
    from tts_amd.distributed import split_frames
    mel0, mean, std = lj001
    frames, n_fft, hop, shards, seed = 51680, 1024, 256, 8, 77
    f, r_cover = n_fft // 2 + 1, n_fft // hop
    mel = np.ascontiguousarray(np.tile(mel0[:793], (frames // 793 + 1, 1))[:frames])
    ks = (16, 32, 60)
    whole = voc.GriffinLimPlan(n_fft, hop, n_fft, 80, frames, 1)
    whole.set_mel_basis(22050, 0.0, 8000)
    s_dev = whole.mel_invert(dev(mel[None]), mean, std)                      # (1, T, F)
    y_whole = {k: whole.griffinlim(s_dev, [frames], phase0=None, seed=seed, n_iter=k).clone() for k in ks}
    ranges = split_frames(frames, shards, min_frames=2 * (r_cover - 1))
    assert [t1 - t0 for t0, t1 in ranges] == [6460] * 8
    plans = [voc.GriffinLimPlan(n_fft, hop, n_fft, 80, t1 - t0, 1) for t0, t1 in ranges]

    def swap():
        packed = [p.shard_halo_pack() for p in plans]
        for r, p in enumerate(plans):
            bufs = p.shard_halo_buffers()
            p.shard_halo_unpack(packed[r - 1][1] if r > 0 else bufs[0],
                                packed[r + 1][0] if r < len(plans) - 1 else bufs[1])

    def sharded(k):
        for p, (t0, t1) in zip(plans, ranges):
            p.shard_begin(s_dev[0, t0:t1].contiguous(), t0, frames, phase0=None, seed=seed)
        swap()
        for _ in range(k):
            for p in plans:
                p.shard_step(0.99)
            swap()
        y = torch.zeros(hop * (frames - 1), dtype=torch.float32, device="cuda")
        filled = 0
        for p in plans:
            piece, n_begin = p.shard_end()
            y[n_begin:n_begin + piece.numel()] = piece
            filled += piece.numel()
        assert filled == y.numel()
        return y

    # what the reference computation itself moves between fp32 and fp64, on a window of this input
    w0, w1 = 3 * 6460 - 200, 3 * 6460 + 200                                   # across the shard 2 | 3 boundary
    s_win = np.ascontiguousarray(s_dev[0, w0:w1].cpu().numpy().T)
    ph_win = from_btf2(whole.drawn_phase(seed, 1, w1 - w0, frame_origin=w0)[0].cpu().numpy())
    _, snaps = O.griffinlim(s_win, n_iter=60, hop_length=hop, win_length=n_fft, init_phase=ph_win, snapshots=list(ks))
    _, snaps64 = O.griffinlim(s_win.astype(np.float64), n_iter=60, hop_length=hop, win_length=n_fft,
                              init_phase=ph_win.astype(np.complex128), snapshots=list(ks))
    for k in ks:
        y = sharded(k)
        d = rel_l2(y.cpu().numpy(), y_whole[k][0].cpu().numpy())
        floor = rel_l2(snaps[k], snaps64[k])
        print(f"[configs3] 8 shards x 6460 frames, k={k}: sharded vs unsharded rel L2 {d:.2e} "
              f"(oracle fp32 vs fp64 on a 400-frame window: {floor:.2e})")
        assert bool(torch.isfinite(y).all())
        assert d < max(1e-3, 3 * floor), (k, d, floor)
        assert k > 32 or d < 1e-3, (k, d)
        if k == 16:
            margin = (k + 2) * r_cover
            a, b = margin * hop, (w1 - w0 - 1 - margin) * hop
            for name, yy in (("sharded", y), ("unsharded", y_whole[k][0])):
                got = yy[w0 * hop + a: w0 * hop + b].cpu().numpy()
                dd = rel_l2(got, snaps[k][a:b])
                print(f"[configs3] {name} vs oracle on frames [{w0}, {w1}) at k={k}: rel L2 {dd:.2e}")
                assert dd < 1e-3, (name, dd)
    for p in plans:
        p.close()
    whole.close()
    # the library's own swap at the per-rank size, 60 iterations: a middle shard that is its own
    # neighbour on both sides, RCCL send / recv against the same swap by copies
    t0, t1 = ranges[3]
    s_mid = s_dev[0, t0:t1].contiguous()
    ref = voc.GriffinLimPlan(n_fft, hop, n_fft, 80, t1 - t0, 1)
    ref.shard_begin(s_mid, t0, frames, seed=seed)

    def self_swap():
        to_prev, to_next = ref.shard_halo_pack()
        ref.shard_halo_unpack(to_prev, to_next)

    self_swap()
    for _ in range(60):
        ref.shard_step(0.99)
        self_swap()
    y_ref = ref.shard_end()[0].clone()
    ref.close()
    chain = voc.HaloChain(voc.HaloChain.unique_id(), 0, 1)
    lib_plan = voc.GriffinLimPlan(n_fft, hop, n_fft, 80, t1 - t0, 1)
    lib_plan.halo_attach(chain)
    lib_plan.set_option("halo_loopback", 1)
    lib_plan.shard_begin(s_mid, t0, frames, seed=seed)
    lib_plan.shard_run(60, 0.99)
    y_lib = lib_plan.shard_end()[0].clone()
    torch.cuda.synchronize()
    assert float(y_lib.abs().max()) > 0 and torch.equal(y_lib, y_ref)
    lib_plan.halo_attach(None)
    lib_plan.close()
    chain.close()


@pytest.mark.parametrize("n_fft,hop,win,lens,fpr", [
    (1024, 256, 1024, [430] * 64, 0), (1024, 256, 1024, [150, 33, 64, 7, 2, 97], 0),
    (1024, 256, 1024, [150, 33, 64, 7, 2, 97], 5), (1024, 512, 800, [120, 60, 33], 0),
    (1024, 256, 1024, [300] * 16 + [111] * 9, 0), (1024, 128, 1024, [90, 41, 8], 0),
    (2048, 256, 2048, [865] * 12, 0), (2048, 256, 2048, [200, 120, 60, 33, 7, 2], 9), (2048, 512, 2048, [100, 37], 0)])
def test_all_iterations_in_one_launch(voc, n_fft, hop, win, lens, fpr):
    """Option "persistent": when the runs of a call are resident on the device together,
    ONE launch carries all iterations - a wave keeps its run, runs whose boundary blocks travel
    through memory wait for their neighbours' progress words; at n_fft 1024 the blocks behind a
    run's last frame come from LDS so the frame loop reads no tail buffer, at n_fft 2048 every
    boundary goes through memory.  The same additions in the same order as
    one launch per iteration: waveforms must be identical bit for bit, from mel input (bins that
    are zero for good left out) and from caller magnitudes (all bins), with drawn and explicit
    phases, for 1, 2, 3, 8 and 60 iterations (60: vocoder/mel_audio.py:126) - and the call must
    really have taken one launch, and no wait may have run out."""
    f = n_fft // 2 + 1
    t_max, b = max(lens), len(lens)
    rng = np.random.default_rng(61)
    mel = np.stack([np.pad(synthetic_mel(700 + i, t), ((0, t_max - t), (0, 0))) for i, t in enumerate(lens)])
    s = np.zeros((b, t_max, f), np.float32)
    p0 = np.zeros((b, t_max, f, 2), np.float32)
    for i, t in enumerate(lens):
        s[i, :t] = rng.standard_normal((t, f)) ** 2
        ang = 2 * np.pi * rng.random((t, f))
        p0[i, :t, :, 0], p0[i, :t, :, 1] = np.cos(ang), np.sin(ang)
    plan = voc.GriffinLimPlan(n_fft, hop, win, 80, t_max, b)
    plan.set_mel_basis(22050, 0.0, 8000)
    plan.set_option("single_frame", 0)
    plan.set_option("frames_per_run", fpr)
    assert plan.persistent_capacity(0) > 0 and plan.persistent_capacity(1) > 0
    got = {}
    for persistent in (1, 0):
        plan.set_option("persistent", persistent)
        for k in (1, 2, 3, 8, 60):
            got[persistent, k, "mel"] = plan.mel_to_audio(dev(mel), MEAN, STD, lens, seed=5, n_iter=k).clone()
            assert plan.last_loop_launches() == (1 if persistent else k)
            if k in (2, 8):
                got[persistent, k, "s"] = plan.griffinlim(dev(s), lens, phase0=dev(p0), n_iter=k).clone()
                assert plan.last_loop_launches() == (1 if persistent else k)
    torch.cuda.synchronize()
    plan.status()
    for (persistent, k, src), y in got.items():
        if persistent:
            assert float(y.abs().max()) > 0
            assert torch.equal(y, got[0, k, src]), (k, src)
    plan.close()


@pytest.mark.parametrize("lens", [[430], [96, 33, 7, 2, 150]])
def test_one_frame_per_wave_in_one_launch(voc, lens):
    """The one-frame-per-wave form (small batches) with all iterations in one launch (n_fft 1024):
    a wave keeps its frame, writes its R blocks through to memory and raises the frame's progress
    word; before the next iteration it waits for the words of the 2(R-1) frames whose blocks
    overlap its own.  Same bits as one launch per iteration, for 1, 2, 3 and 60 iterations."""
    t_max, b = max(lens), len(lens)
    mel = np.stack([np.pad(synthetic_mel(900 + i, t), ((0, t_max - t), (0, 0))) for i, t in enumerate(lens)])
    plan = voc.GriffinLimPlan(1024, 256, 1024, 80, t_max, b)
    plan.set_mel_basis(22050, 0.0, 8000)
    plan.set_option("single_frame", 1)
    assert plan.persistent_capacity(2) > 0
    got = {}
    for persistent in (2, 0):   # (2: also where the library would not choose it by itself)
        plan.set_option("persistent", persistent)
        for k in (1, 2, 3, 60):
            got[persistent, k] = plan.mel_to_audio(dev(mel), MEAN, STD, lens, seed=5, n_iter=k).clone()
            assert plan.last_loop_launches() == (1 if persistent else k)
    torch.cuda.synchronize()
    plan.status()
    for k in (1, 2, 3, 60):
        assert float(got[2, k].abs().max()) > 0
        assert torch.equal(got[2, k], got[0, k]), k
    plan.close()


def test_a_launch_that_is_not_resident_as_a_whole_is_reported(voc):
    """The iteration-spanning launch needs its workgroups on the device together.  Forced (option
    "persistent" = 2, a test aid) on ONE long utterance cut into twice as many workgroups as fit,
    for more iterations than there are resident workgroups in the chain, every resident workgroup
    ends up waiting for one that is not: the waits run out after a fraction of a second, every wave
    still ends, the plan's status says that the call's results are void, and the plan goes back to
    one launch per iteration - whose result is the right one.  (A batch of many utterances in
    launch order does not get there: utterances that are resident as a whole finish and make
    room.)"""
    n_fft, hop, frames = 1024, 256, 30000
    plan = voc.GriffinLimPlan(n_fft, hop, n_fft, 80, frames, 1)
    plan.set_mel_basis(22050, 0.0, 8000)
    cap = plan.persistent_capacity(1)
    assert cap > 0
    n_iter = cap + 40
    mel = dev(synthetic_mel(800, frames)[None])
    plan.set_option("run_target", 8 * cap)   # 2 x cap workgroups of 4 runs
    plan.set_option("persistent", 0)
    want = plan.mel_to_audio(mel, MEAN, STD, [frames], seed=9, n_iter=n_iter).clone()
    plan.set_option("persistent", 2)
    plan.mel_to_audio(mel, MEAN, STD, [frames], seed=9, n_iter=n_iter)
    assert plan.last_loop_launches() == 1
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="waited in vain"):
        plan.status()
    again = plan.mel_to_audio(mel, MEAN, STD, [frames], seed=9, n_iter=n_iter)
    assert plan.last_loop_launches() == n_iter
    assert torch.equal(again, want)
    plan.close()
