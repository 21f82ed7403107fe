"""The wave engine's exact source (tts_amd/csrc/*.h) executed on the CPU, 64 lanes as a
loop (tests/csrc/emu.cpp), against the oracle.  This checks, without a GPU, the FFT index
algebra, the LDS exchange layouts, the run / head / tail bookkeeping of the register
overlap-add and the window-sum-square edge tables that the HIP kernels execute."""
import ctypes
import os

import numpy as np
import pytest

from conftest import GOLDEN, rel_l2
from helpers import from_btf2, to_btf, to_btf2
from oracle import griffinlim_oracle as O

fp = ctypes.POINTER(ctypes.c_float)
ip = ctypes.POINTER(ctypes.c_int)


def P(a):
    return a.ctypes.data_as(fp)


def emu_griffinlim(emu, s_btf, ph_btf2, lens, n_fft, hop, win, n_iter, fpr, momentum=0.99, want_sc=True,
                   recompute=0):
    b, t, _ = s_btf.shape
    stride = hop * (t - 1)
    y = np.zeros((b, stride), np.float32)
    sc = np.zeros((b, max(n_iter, 1)), np.float32)
    nf = np.asarray(lens, np.int32)
    rc = emu.emu_griffinlim(n_fft, hop, win, P(s_btf), P(ph_btf2), ctypes.c_uint64(0), nf.ctypes.data_as(ip),
                            b, t, n_iter, ctypes.c_float(momentum), fpr, P(y), ctypes.c_int64(stride),
                            P(sc) if (want_sc and n_iter > 0) else None, recompute)
    assert rc == 0
    return y, sc[:, :n_iter]


SHAPES = [(1024, 256, 1024, 23, 7), (1024, 256, 1024, 23, 100), (2048, 256, 2048, 26, 9),
          (2048, 512, 2048, 17, 5), (1024, 512, 1024, 9, 3), (1024, 128, 1024, 30, 9),
          (2048, 1024, 2048, 12, 4), (1024, 256, 800, 15, 4), (1024, 256, 1024, 2, 8),
          (1024, 256, 1024, 5, 2), (2048, 256, 2048, 3, 8), (2048, 512, 2048, 2, 4), (2048, 256, 1200, 6, 7)]


@pytest.mark.parametrize("n_fft,hop,win,frames,fpr", SHAPES)
def test_transforms_match_oracle(emu, n_fft, hop, win, frames, fpr):
    rng = np.random.default_rng(frames)
    f = n_fft // 2 + 1
    length = hop * (frames - 1)
    nf = np.array([frames], np.int32)
    y = rng.standard_normal(length).astype(np.float32)
    d = np.zeros((1, frames, f, 2), np.float32)
    assert emu.emu_stft(n_fft, hop, win, P(y), ctypes.c_int64(length), nf.ctypes.data_as(ip), 1, frames,
                        fpr, P(d)) == 0
    ref = O.stft(y, n_fft=n_fft, hop_length=hop, win_length=win)
    assert rel_l2(from_btf2(d[0]), ref) < 1e-6
    # Hermitian edge bins come out exactly real
    assert np.all(d[0, :, 0, 1] == 0) and np.all(d[0, :, -1, 1] == 0)
    spec = (rng.standard_normal((f, frames)) + 1j * rng.standard_normal((f, frames))).astype(np.complex64)
    out = np.zeros((1, length), np.float32)
    dc = to_btf2(spec)
    assert emu.emu_istft(n_fft, hop, win, P(dc), nf.ctypes.data_as(ip), 1, frames, fpr, P(out),
                         ctypes.c_int64(length)) == 0
    assert rel_l2(out[0], O.istft(spec, hop_length=hop, win_length=win)) < 1e-6


@pytest.mark.parametrize("n_fft,hop,win,frames,fpr", SHAPES)
def test_griffinlim_matches_oracle(emu, n_fft, hop, win, frames, fpr):
    rng = np.random.default_rng(100 + frames)
    f = n_fft // 2 + 1
    s = (rng.standard_normal((f, frames)) ** 2).astype(np.float32)
    s[::7] *= -0.3                                   # signed input, as the pseudo-inverse produces
    ph = np.exp(2j * np.pi * rng.random((f, frames))).astype(np.complex64)
    y_ref, sc_ref = O.griffinlim(s, n_iter=4, hop_length=hop, win_length=win, init_phase=ph, return_sc=True)
    y, sc = emu_griffinlim(emu, to_btf(s), to_btf2(ph), [frames], n_fft, hop, win, 4, fpr)
    assert rel_l2(y[0], y_ref) < 5e-6
    assert np.max(np.abs(sc[0] - sc_ref) / sc_ref) < 1e-5
    # no momentum, no iterations
    y0, _ = emu_griffinlim(emu, to_btf(s), to_btf2(ph), [frames], n_fft, hop, win, 0, fpr)
    assert rel_l2(y0[0], O.griffinlim(s, n_iter=0, hop_length=hop, win_length=win, init_phase=ph)) < 1e-6
    ym, _ = emu_griffinlim(emu, to_btf(s), to_btf2(ph), [frames], n_fft, hop, win, 3, fpr, momentum=0.0)
    assert rel_l2(ym[0], O.griffinlim(s, n_iter=3, hop_length=hop, win_length=win, init_phase=ph,
                                      momentum=0.0)) < 5e-6


def test_ragged_batch_and_run_length_independence(emu):
    n_fft, hop = 1024, 256
    f = n_fft // 2 + 1
    lens = [19, 3, 40, 8]
    t_max = max(lens)
    rng = np.random.default_rng(9)
    s = np.zeros((len(lens), t_max, f), np.float32)
    p = np.zeros((len(lens), t_max, f, 2), np.float32)
    for b, t in enumerate(lens):
        s[b, :t] = rng.standard_normal((t, f)) ** 2
        ang = 2 * np.pi * rng.random((t, f))
        p[b, :t, :, 0], p[b, :t, :, 1] = np.cos(ang), np.sin(ang)
    ys = {}
    for fpr in (4, 11, 1000):
        ys[fpr], _ = emu_griffinlim(emu, s, p, lens, n_fft, hop, n_fft, 3, fpr)
    for b, t in enumerate(lens):
        n = hop * (t - 1)
        ref = O.griffinlim(np.ascontiguousarray(s[b, :t].T), n_iter=3, hop_length=hop, win_length=n_fft,
                           init_phase=from_btf2(p[b, :t]))
        for fpr in ys:
            assert rel_l2(ys[fpr][b, :n], ref) < 5e-6
            assert np.all(ys[fpr][b, n:] == 0)


@pytest.mark.parametrize("name", ["stages_n1024_t16", "stages_n2048_t12"])
def test_golden_stage_fixtures(emu, name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    n_fft, hop = int(g["n_fft"]), int(g["hop"])
    s, ph = g["linear_spec"], g["phase0"]
    frames = s.shape[1]
    for k in (0, 1, 2, 4, 8):
        y, sc = emu_griffinlim(emu, to_btf(s), to_btf2(ph), [frames], n_fft, hop, n_fft, k, 5)
        assert rel_l2(y[0], g[f"y_k{k}"]) < 2e-5, k
        if k:
            assert np.max(np.abs(sc[0] - g["sc"][:k]) / g["sc"][:k]) < 1e-5


def test_fixture_drift_stays_at_the_fp32_noise_floor(emu, lj001):
    """16 iterations on 430 frames of the reference's mel: fp32 rounding differences grow
    with k under momentum 0.99, and must stay two orders below the 1e-3 bound here."""
    mel, mean, std = lj001
    s = O.mel_invert(mel[:430], mean, std, n_fft=1024)
    ph = np.exp(2j * np.pi * np.random.default_rng(0).random(s.shape)).astype(np.complex64)
    _, sc_ref, snaps = O.griffinlim(s, n_iter=16, hop_length=256, win_length=1024, init_phase=ph,
                                    return_sc=True, snapshots=[16])
    y, sc = emu_griffinlim(emu, to_btf(s), to_btf2(ph), [430], 1024, 256, 1024, 16, 14)
    assert rel_l2(y[0], snaps[16]) < 2e-4
    assert np.max(np.abs(sc[0] - sc_ref) / sc_ref) < 1e-5


@pytest.mark.parametrize("n_fft,hop,win,frames,fpr", [(1024, 256, 1024, 23, 7), (2048, 256, 2048, 26, 9),
                                                      (1024, 512, 1024, 12, 3), (1024, 128, 1024, 30, 9)])
def test_recompute_mode_is_the_same_iteration(emu, n_fft, hop, win, frames, fpr):
    """The variant that recomputes the previous spectrum from the previous signal instead of
    keeping it in memory performs the same arithmetic: results agree with the stored-spectrum
    variant to the last bits and with the oracle."""
    rng = np.random.default_rng(200 + frames)
    f = n_fft // 2 + 1
    s = (rng.standard_normal((f, frames)) ** 2).astype(np.float32)
    ph = np.exp(2j * np.pi * rng.random((f, frames))).astype(np.complex64)
    y_ref, sc_ref = O.griffinlim(s, n_iter=5, hop_length=hop, win_length=win, init_phase=ph, return_sc=True)
    y0, sc0 = emu_griffinlim(emu, to_btf(s), to_btf2(ph), [frames], n_fft, hop, win, 5, fpr)
    y1, sc1 = emu_griffinlim(emu, to_btf(s), to_btf2(ph), [frames], n_fft, hop, win, 5, fpr, recompute=1)
    assert rel_l2(y1[0], y0[0]) < 1e-6
    assert rel_l2(y1[0], y_ref) < 5e-6
    assert np.max(np.abs(sc1[0] - sc_ref) / sc_ref) < 1e-5


def test_device_phase_generator_statistics(emu):
    """The counter generator behind device-drawn initial phases: uniform on [0, 1), no visible
    correlation between neighbouring elements, between seeds, or across the 2^32 index boundary."""
    n = 1 << 18
    def draw(seed, idx0):
        u = np.empty(n, np.float32)
        emu.emu_counter_uniform(ctypes.c_uint64(seed), ctypes.c_uint64(idx0), n,
                                u.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
        return u.astype(np.float64)
    a = draw(0, 0)
    assert a.min() >= 0.0 and a.max() < 1.0
    assert abs(a.mean() - 0.5) < 3e-3 and abs(a.var() - 1 / 12) < 2e-3
    hist = np.histogram(a, bins=64, range=(0, 1))[0]
    assert np.abs(hist / (n / 64) - 1).max() < 0.1
    corr = lambda x, y: float(np.corrcoef(x, y)[0, 1])
    assert abs(corr(a[:-1], a[1:])) < 0.01
    assert abs(corr(a, draw(1, 0))) < 0.01
    assert abs(corr(a, draw(0, 1 << 32))) < 0.01
    assert abs(corr(a, draw(1 << 32, 0))) < 0.01
    # the phasor sum of n random phases is O(sqrt(n))
    assert abs(np.exp(2j * np.pi * a).sum()) < 5 * np.sqrt(n)


@pytest.mark.parametrize("n_fft,hop,win,lens", [(1024, 256, 1024, [23, 2, 9]), (2048, 256, 2048, [26, 3]),
                                                (2048, 512, 1200, [12]), (1024, 128, 1024, [30, 5]),
                                                (1024, 512, 1024, [7]), (2048, 1024, 2048, [6, 2])])
def test_one_frame_per_wave_mode_matches_oracle(emu, n_fft, hop, win, lens):
    """The small-batch form of the iteration (every frame its own wave, R partial-sum buffers
    instead of a running sum) is the same algorithm: ragged lengths, edge blocks, signed
    magnitudes, and buffers that start out holding garbage."""
    f = n_fft // 2 + 1
    t_max = max(lens)
    rng = np.random.default_rng(31 + t_max)
    s = np.zeros((len(lens), t_max, f), np.float32)
    p = np.zeros((len(lens), t_max, f, 2), np.float32)
    for b, t in enumerate(lens):
        s[b, :t] = rng.standard_normal((t, f)) ** 2
        s[b, :t, ::7] *= -0.3
        ang = 2 * np.pi * rng.random((t, f))
        p[b, :t, :, 0], p[b, :t, :, 1] = np.cos(ang), np.sin(ang)
    for k in (0, 4):
        y, sc = emu_griffinlim(emu, s, p, lens, n_fft, hop, win, k, 1, recompute=2)
        for b, t in enumerate(lens):
            n = hop * (t - 1)
            ref, sc_ref = O.griffinlim(np.ascontiguousarray(s[b, :t].T), n_iter=k, hop_length=hop, win_length=win,
                                       n_fft=n_fft, init_phase=from_btf2(p[b, :t]), return_sc=True)
            assert rel_l2(y[b, :n], ref) < 5e-6
            assert np.all(y[b, n:] == 0)
            if k:
                assert np.max(np.abs(sc[b] - sc_ref) / sc_ref) < 1e-5


@pytest.mark.parametrize("n_fft,hop,win,frames,fpr", SHAPES)
def test_time_domain_momentum_is_the_same_iteration(emu, n_fft, hop, win, frames, fpr):
    """rebuilt_k - c * rebuilt_{k-1} = STFT(y_k - c * y_{k-1}): the variant that forms the momentum
    term before the forward transform (no spectrum state in memory) gives the waveform and the
    convergence curve of librosa's loop."""
    rng = np.random.default_rng(100 + frames)
    f = n_fft // 2 + 1
    s = (rng.standard_normal((f, frames)) ** 2).astype(np.float32)
    s[::7] *= -0.3
    ph = np.exp(2j * np.pi * rng.random((f, frames))).astype(np.complex64)
    y_ref, sc_ref = O.griffinlim(s, n_iter=5, hop_length=hop, win_length=win, init_phase=ph, return_sc=True)
    y, sc = emu_griffinlim(emu, to_btf(s), to_btf2(ph), [frames], n_fft, hop, win, 5, fpr, recompute=3)
    assert rel_l2(y[0], y_ref) < 1e-5
    assert np.max(np.abs(sc[0] - sc_ref) / sc_ref) < 1e-5
    # the waveform does not depend on whether the curve is asked for
    y2, _ = emu_griffinlim(emu, to_btf(s), to_btf2(ph), [frames], n_fft, hop, win, 5, fpr, recompute=3,
                           want_sc=False)
    assert np.array_equal(y, y2)


@pytest.mark.parametrize("n_fft,hop,win,lens,fpr", [(1024, 256, 1024, [40, 23, 7, 64], 3), (1024, 256, 1024, [90], 5),
                                                    (1024, 256, 1024, [33, 3, 50], 9), (1024, 128, 1024, [60, 31], 7),
                                                    (1024, 512, 1024, [30, 12], 1), (1024, 256, 800, [45, 17], 4),
                                                    (2048, 256, 2048, [40, 20], 7)])
def test_run_boundaries_completed_by_the_producer(emu, n_fft, hop, win, lens, fpr):
    """Where two consecutive runs of an utterance are waves of one workgroup, the later run can
    hand its head sums to the earlier one, which writes the boundary blocks complete (modes 5 and
    6) instead of both writing halves that every reader adds (modes 3 and 4).  head + tail either
    way: the waveforms must be identical bit for bit - with all bins kept and with the upper
    quarter left out, for batches whose runs cross workgroup and utterance boundaries, for 0 and
    several iterations - and match the oracle.  (n_fft 2048 has no such kernels: the mode is
    accepted and changes nothing.)"""
    rng = np.random.default_rng(7 + len(lens) + fpr)
    f = n_fft // 2 + 1
    t_max = max(lens)
    s = np.zeros((len(lens), t_max, f), np.float32)
    ph = np.zeros((len(lens), t_max, f, 2), np.float32)
    for b, t in enumerate(lens):
        s[b, :t] = rng.standard_normal((t, f)) ** 2
        ang = 2 * np.pi * rng.random((t, f))
        ph[b, :t, :, 0], ph[b, :t, :, 1] = np.cos(ang), np.sin(ang)
    s_quarter = s.copy()
    s_quarter[:, :, (3 * (n_fft // 2)) // 4:] = 0.0          # what mode 4 / 6 may assume
    for k in (0, 1, 4):
        for two_halves, by_producer, mags in ((3, 5, s), (4, 6, s_quarter)):
            y_a, _ = emu_griffinlim(emu, mags, ph, lens, n_fft, hop, win, k, fpr, recompute=two_halves, want_sc=False)
            y_b, _ = emu_griffinlim(emu, mags, ph, lens, n_fft, hop, win, k, fpr, recompute=by_producer, want_sc=False)
            assert np.array_equal(y_a, y_b), (k, by_producer)
    for b, t in enumerate(lens):                             # y_b: mode 6, 4 iterations
        ref = O.griffinlim(np.ascontiguousarray(s_quarter[b, :t].T), n_iter=4, hop_length=hop, win_length=win,
                           init_phase=(ph[b, :t, :, 0] + 1j * ph[b, :t, :, 1]).T.astype(np.complex64))
        assert rel_l2(y_b[b, :hop * (t - 1)], ref) < 1e-5


def test_time_domain_momentum_drift_on_the_fixture(emu, lj001):
    """Same bound as the spectral-domain form: 16 iterations on 430 frames of the reference's mel
    stay two orders below the 1e-3 parity bound."""
    mel, mean, std = lj001
    s = O.mel_invert(mel[:430], mean, std, n_fft=1024)
    ph = np.exp(2j * np.pi * np.random.default_rng(0).random(s.shape)).astype(np.complex64)
    _, sc_ref, snaps = O.griffinlim(s, n_iter=16, hop_length=256, win_length=1024, init_phase=ph,
                                    return_sc=True, snapshots=[16])
    y, sc = emu_griffinlim(emu, to_btf(s), to_btf2(ph), [430], 1024, 256, 1024, 16, 14, recompute=3)
    assert rel_l2(y[0], snaps[16]) < 2e-4
    assert np.max(np.abs(sc[0] - sc_ref) / sc_ref) < 1e-5


def test_run_target_prefers_whole_workgroups_per_cu(emu):
    """The automatic run count on a 256-CU device with 4-wave workgroups: never more than the wave
    slots or than the minimum run length allows; a whole number of workgroups per CU where
    the measured launch times say so (8 utterances of 430 frames: 1024 runs, not the 1146 the
    minimum length allows; 16: 2048; 64: all 3072 slots); the longest table when that is cheaper
    (12 utterances)."""
    emu.emu_pick_run_target.restype = ctypes.c_int64
    pick = lambda total, bpc, min_len: emu.emu_pick_run_target(ctypes.c_int64(total), 256, 4, bpc, min_len)
    assert pick(430 * 8, 3, 3) == 1024
    assert pick(430 * 12, 3, 3) == 430 * 12 // 3
    assert pick(430 * 16, 3, 3) == 2048
    assert pick(430 * 64, 3, 3) == 3072
    assert pick(865 * 8, 2, 7) == 865 * 8 // 7
    assert pick(865 * 32, 2, 7) == 2048
    for total in (2, 7, 100, 3071, 3072, 3073, 50000, 196608):
        for bpc, min_len in ((3, 3), (2, 7), (1, 1)):
            t = pick(total, bpc, min_len)
            assert 1 <= t <= min(256 * 4 * bpc, max(1, total // min_len))


@pytest.mark.parametrize("lens,target,r", [([430] * 64, 3072, 4), ([430] * 64, 4096, 4), ([865] * 64, 2048, 8),
                                           ([97, 3, 640, 12, 2, 311], 100, 4), ([5, 5, 5], 64, 4),
                                           ([865], 2048, 8), ([2, 2], 1, 2)])
def test_run_table_with_a_given_number_of_runs(emu, lens, target, r):
    """Runs are dealt in proportion to the frames: exactly `target` of them when every utterance
    can be cut that often, never shorter than R-1 frames (unless the utterance is), contiguous
    and complete."""
    nf = np.asarray(lens, np.int32)
    out = np.zeros((max(target, len(lens)) * 2 + 8, 4), np.int32)
    n = emu.emu_runs_target(nf.ctypes.data_as(ip), len(lens), ctypes.c_int64(target), r,
                            out.ctypes.data_as(ip), len(out))
    runs = out[:n]
    most = sum(max(1, t // max(1, r - 1)) for t in lens)
    assert min(target, most) <= n <= min(most, target + len(lens))
    if most >= target >= len(lens) and len(set(lens)) == 1:
        assert n == target
    for b, t in enumerate(lens):
        mine = runs[runs[:, 0] == b]
        assert mine[0, 1] == 0 and mine[-1, 2] == t and np.all(mine[1:, 1] == mine[:-1, 2])
        assert np.all(mine[:, 3] == t)
        if len(mine) > 1:
            assert np.all(mine[:, 2] - mine[:, 1] >= r - 1)
    sizes = runs[:, 2] - runs[:, 1]
    if len(set(lens)) == 1 and n == target:
        assert sizes.max() - sizes.min() <= 1


def emu64_griffinlim(emu64, s_ft, ph_ft, n_fft, hop, n_iter, fpr, mode):
    """The double-precision build of the engine on one utterance; (F, T) float64 / complex128 in."""
    dp = ctypes.POINTER(ctypes.c_double)
    frames = s_ft.shape[1]
    s = np.ascontiguousarray(s_ft.T[None], np.float64)
    x = ph_ft.T
    p = np.ascontiguousarray(np.stack([x.real, x.imag], -1)[None], np.float64)
    stride = hop * (frames - 1)
    y = np.zeros((1, stride), np.float64)
    nf = np.array([frames], np.int32)
    rc = emu64.emu_griffinlim(n_fft, hop, n_fft, s.ctypes.data_as(dp), p.ctypes.data_as(dp), ctypes.c_uint64(0),
                              nf.ctypes.data_as(ip), 1, frames, n_iter, ctypes.c_double(0.99), fpr,
                              y.ctypes.data_as(dp), ctypes.c_int64(stride), None, mode)
    assert rc == 0
    return y[0]


@pytest.mark.parametrize("n_fft,frames,mode", [(1024, 430, 3), (1024, 430, 2), (2048, 200, 3), (2048, 200, 0)])
def test_double_precision_engine_is_librosas_iteration(emu64, lj001, n_fft, frames, mode):
    """The engine compiled in double precision against the oracle run in double precision, on
    the reference's mel, 60 iterations, momentum 0.99: the waveforms agree to 1e-9 - the
    engine's iteration IS librosa's (momentum in the time domain, mode 3, and one frame per
    wave, mode 2, included).  In fp32 the same pair differs by 1e-4 .. 1e-3 at 60 iterations, and
    so does the oracle from itself when only its precision changes (printed): that difference is
    rounding, amplified by the momentum iteration."""
    mel, mean, std = lj001
    s32 = O.mel_invert(mel[:frames], mean, std, n_fft=n_fft)
    ph = np.exp(2j * np.pi * np.random.default_rng(0).random(s32.shape))
    ks = [4, 60]
    _, want = O.griffinlim(s32.astype(np.float64), n_iter=60, hop_length=256, win_length=n_fft, init_phase=ph,
                           snapshots=ks)
    for k in ks:
        got = emu64_griffinlim(emu64, s32.astype(np.float64), ph, n_fft, 256, k, 9, mode)
        assert rel_l2(got, want[k]) < (1e-13 if k == 4 else 1e-9), k
    y32 = O.griffinlim(s32, n_iter=60, hop_length=256, win_length=n_fft, init_phase=ph.astype(np.complex64))
    print(f"oracle fp32 vs oracle fp64 at k=60, n_fft {n_fft}: {rel_l2(y32, want[60]):.2e}")


@pytest.mark.parametrize("n_fft,hop,win,frames,fpr", [(1024, 256, 1024, 23, 7), (2048, 256, 2048, 26, 9),
                                                      (2048, 512, 2048, 17, 5), (1024, 128, 1024, 30, 9),
                                                      (1024, 256, 1024, 2, 8)])
def test_zero_magnitude_bins_are_left_out(emu, n_fft, hop, win, frames, fpr):
    """Magnitudes that vanish from bin 3/4 * n_fft/2 on (what a mel basis ending below the Nyquist
    frequency gives the reference's pinv product): the iteration that neither fetches nor updates
    those bins (mode 4) is the oracle's iteration, waveform and convergence curve, and agrees
    with the full pair-order iteration (mode 3) to rounding."""
    rng = np.random.default_rng(300 + frames)
    f = n_fft // 2 + 1
    s = (rng.standard_normal((f, frames)) ** 2).astype(np.float32)
    s[::7] *= -0.3
    s[3 * (n_fft // 2) // 4:] = 0.0
    ph = np.exp(2j * np.pi * rng.random((f, frames))).astype(np.complex64)
    y_ref, sc_ref = O.griffinlim(s, n_iter=5, hop_length=hop, win_length=win, init_phase=ph, return_sc=True)
    y4, sc4 = emu_griffinlim(emu, to_btf(s), to_btf2(ph), [frames], n_fft, hop, win, 5, fpr, recompute=4)
    y3, sc3 = emu_griffinlim(emu, to_btf(s), to_btf2(ph), [frames], n_fft, hop, win, 5, fpr, recompute=3)
    assert rel_l2(y4[0], y_ref) < 1e-5
    assert rel_l2(y4[0], y3[0]) < 1e-6
    assert np.max(np.abs(sc4[0] - sc_ref) / sc_ref) < 1e-5
    y4n, _ = emu_griffinlim(emu, to_btf(s), to_btf2(ph), [frames], n_fft, hop, win, 5, fpr, recompute=4,
                            want_sc=False)
    assert np.array_equal(y4, y4n)


@pytest.mark.parametrize("n_fft,nzh", [(1024, 4), (1024, 2), (2048, 8), (2048, 4)])
def test_pair_order_of_the_magnitudes_is_a_layout(emu, n_fft, nzh):
    """Pair order gives every bin that can be non-zero exactly one slot inside the frame: the low
    registers where native order has them, the mirrors of the last nzh pairs behind them, the
    self-mirrored bin n_fft/4 next to the Nyquist bin; only registers from 3/4 up lose their slot
    when nzh is half the pairs (lane 0's mirrors sit one register higher than everybody else's)."""
    e, p = n_fft // 128, n_fft // 2
    slots = {}
    for reg in range(e):
        for lane in range(64):
            s = emu.emu_pair_s_slot(n_fft, nzh, reg, lane)
            low = (lane >> 3) + 8 * (lane & 7)
            if s < 0:
                assert nzh < e // 2 and reg >= e // 2 + nzh + (1 if low == 0 else 0)
                continue
            assert 0 <= s < p + 4 and s not in slots and s != p
            slots[s] = (reg, lane)
            if reg < e // 2:
                assert s == ((reg >> 2) * 64 + lane) * 4 + (reg & 3)
    assert slots[p + 1] == (e // 2, 0)
    # every native position when all mirrors are kept; otherwise the low half, nzh mirrors per
    # lane and the self-mirrored bin
    assert len(slots) == (64 * e if nzh == e // 2 else 32 * e + 64 * nzh + 1)
