// CPU emulation of the wave engine (tts_amd/csrc/engine.h): the 64 lanes of a
// wavefront become a loop, LDS becomes two alternating host buffers, a kernel
// launch becomes a loop over runs.  TEST INFRASTRUCTURE: it exists so the index
// algebra, the run/tail bookkeeping and the numerics of the exact code the GPU
// executes can be checked against the oracle on a machine without a GPU.  It is
// built by tests/conftest.py into tests/csrc/libglvemu.so and is not part of the
// product library.
#include <stdint.h>
#include <string.h>

#include <cmath>
#include <vector>

#include "../../tts_amd/csrc/elementwise.h"
#include "../../tts_amd/csrc/engine.h"
#include "../../tts_amd/csrc/host_tables.h"

using namespace glv;

namespace {

template <int A, int HB, int MODE, bool SC>
void emu_launch(const IterArgs& a) {
    using Eng = Engine<A, HB>;
    using G = Geo<A>;
    std::vector<cf> b0(G::XBUF), b1(G::XBUF);
    std::vector<Eng> e(64);
    // Run boundaries completed by the producer (IterArgs::deposit): on the device the later run of
    // a workgroup deposits its head sums in LDS early and the earlier run picks them up at its
    // end.  The runs of a launch are independent of each other, so here they are simply taken in
    // descending order: when a run ends, the deposit of the run after it is already there.
    constexpr bool kCanDeposit = (MODE == MODE_ITER_LIN || MODE == MODE_ITER_LIN_Z || MODE == MODE_ISTFT) && !SC && A == 1;
    const bool deposit = kCanDeposit && a.deposit != 0;
    std::vector<cf> dep_mine(Eng::kShareFloats / 2 + 1), dep_next(Eng::kShareFloats / 2 + 1);
    for (int ri = a.n_runs - 1; ri >= 0; --ri) {
        const RunDesc run = a.runs[ri];
        for (int l = 0; l < 64; ++l) e[l].init(a, a.tables, l, run);
        std::swap(dep_mine, dep_next);  // what the run after this one deposited
        if (deposit) {
            const bool out = (ri % 4) != 0 && a.runs[ri - 1].utt == run.utt;
            const bool in = (ri % 4) != 3 && ri + 1 < a.n_runs && a.runs[ri + 1].utt == run.utt;
            for (int l = 0; l < 64; ++l) {
                e[l].deposit_out = out ? dep_mine.data() : nullptr;
                e[l].deposit_in = in ? dep_next.data() : nullptr;
            }
        }
        if (MODE == MODE_ITER1 || MODE == MODE_ISTFT1) {
            // one frame per wave: no sliding window, no running sum
            const int t = run.t_begin;
            if (MODE == MODE_ITER1) {
                for (int l = 0; l < 64; ++l) e[l].template ph0_single<true>(a, t, b0.data());
                for (int l = 0; l < 64; ++l) e[l].ph1_read(b0.data());
                for (int l = 0; l < 64; ++l) e[l].ph1_write(b1.data());
                for (int l = 0; l < 64; ++l) e[l].ph2_read(b1.data());
                for (int l = 0; l < 64; ++l) e[l].ph2_write(b0.data());
                for (int l = 0; l < 64; ++l) e[l].template ph3_update<SC, 0, false>(a, t, b0.data());
            } else {
                for (int l = 0; l < 64; ++l) e[l].ph3_load(a, t);
            }
            for (int l = 0; l < 64; ++l) e[l].ph3_write(b1.data());
            for (int l = 0; l < 64; ++l) e[l].ph4(b1.data(), b0.data());
            for (int l = 0; l < 64; ++l) e[l].ph5_read(b0.data());
            for (int l = 0; l < 64; ++l) e[l].ph5_write(b1.data());
            for (int l = 0; l < 64; ++l) e[l].ph6_single(a, t, b1.data());
            if (SC && a.sc_partial) {
                float num = 0.f, den = 0.f;
                for (int l = 0; l < 64; ++l) {
                    num += e[l].sc_num;
                    den += e[l].sc_den;
                }
                a.sc_partial[2 * ri] = num;
                a.sc_partial[2 * ri + 1] = den;
            }
            continue;
        }
        constexpr bool kRc = MODE == MODE_ITER_RC || MODE == MODE_ITER_RC_FIRST;
        constexpr int kRcArg = MODE == MODE_ITER_RC ? 1 : (MODE == MODE_ITER_RC_FIRST ? 2 : 0);
        constexpr bool kLin = MODE == MODE_ITER_LIN || MODE == MODE_ITER_LIN_Z;
        constexpr int kNzh = MODE == MODE_ITER_LIN_Z ? Eng::NP / 2 : Eng::NP;
        // runs of one workgroup (4 consecutive runs) hand their preloaded boundary blocks to the
        // run before them, as the device kernel does through LDS (n_fft 1024, no metric)
        std::vector<cf> shared(Eng::kShareFloats / 2 + 1);
        bool take_next = false;
        if (kLin && !SC && A == 1 && (ri % 4) != 3 && ri + 1 < a.n_runs && a.runs[ri + 1].utt == run.utt && !a.raw_in) {
            std::vector<Eng> nx(64);
            for (int l = 0; l < 64; ++l) {
                nx[l].init(a, a.tables, l, a.runs[ri + 1]);
                nx[l].template preload_lin<SC>(a);
                if constexpr (A == 1) nx[l].publish_preload(shared.data());
            }
            take_next = true;
        }
        if (kLin)
            for (int l = 0; l < 64; ++l) {
                e[l].template preload_lin<SC>(a);
                if (take_next) e[l].share_next = shared.data();
                e[l].prefetch_block(a, run.t_begin);
                e[l].prefetch_block_prev(a, run.t_begin);
                e[l].template prefetch_s_pair<kNzh>(a, run.t_begin);
            }
        if (MODE != MODE_ISTFT && !kLin)
            for (int l = 0; l < 64; ++l) {
                e[l].preload(a);
                e[l].prefetch_block(a, run.t_begin);
            }
        if (MODE == MODE_ITER_RC)
            for (int l = 0; l < 64; ++l) {
                e[l].preload_prev(a);
                e[l].prefetch_block_prev(a, run.t_begin);
            }
        if (MODE == MODE_ITER)
            for (int l = 0; l < 64; ++l) e[l].prefetch_spec(a, run.t_begin);
        if (kRc)
            for (int l = 0; l < 64; ++l) e[l].prefetch_s(a, run.t_begin);
        for (int t = run.t_begin; t < run.t_end; ++t) {
            if (MODE == MODE_ITER_RC) {
                for (int l = 0; l < 64; ++l) e[l].ph0_prev(a, t, b0.data());
                for (int l = 0; l < 64; ++l) e[l].ph1_read(b0.data());
                for (int l = 0; l < 64; ++l) e[l].ph1_write(b1.data());
                for (int l = 0; l < 64; ++l) e[l].ph2_read(b1.data());
                for (int l = 0; l < 64; ++l) e[l].ph2_write(b0.data());
                for (int l = 0; l < 64; ++l) e[l].ph3_prev(b0.data());
            }
            if (kLin) {
                for (int l = 0; l < 64; ++l) e[l].template lin_advance<SC>(a, t);
                if (SC) {
                    for (int l = 0; l < 64; ++l) e[l].lin_window_signal(b0.data());
                    for (int l = 0; l < 64; ++l) e[l].ph1_read(b0.data());
                    for (int l = 0; l < 64; ++l) e[l].ph1_write(b1.data());
                    for (int l = 0; l < 64; ++l) e[l].ph2_read(b1.data());
                    for (int l = 0; l < 64; ++l) e[l].pair_store(b0.data());
                    for (int l = 0; l < 64; ++l) e[l].ph3_metric_pair(b0.data());
                }
                for (int l = 0; l < 64; ++l) {
                    e[l].lin_window_combined(b0.data());
                    e[l].template lin_shift<SC>();
                }
                for (int l = 0; l < 64; ++l) e[l].ph1_read(b0.data());
                for (int l = 0; l < 64; ++l) e[l].ph1_write(b1.data());
                for (int l = 0; l < 64; ++l) e[l].ph2_read(b1.data());
                for (int l = 0; l < 64; ++l) e[l].pair_store(b0.data());
                // (reads the partners from one buffer, sends them their results through the other)
                for (int l = 0; l < 64; ++l) e[l].template ph3_pair<kNzh>(a, t, b0.data(), b1.data());
                for (int l = 0; l < 64; ++l) e[l].ph4_pair(b1.data(), b0.data());
                for (int l = 0; l < 64; ++l) e[l].ph5_read(b0.data());
                for (int l = 0; l < 64; ++l) e[l].ph5_write(b1.data());
                for (int l = 0; l < 64; ++l) e[l].ph6(a, t, b1.data());
                continue;
            } else if (MODE != MODE_ISTFT) {
                for (int l = 0; l < 64; ++l) e[l].ph0(a, t, b0.data());
                for (int l = 0; l < 64; ++l) e[l].ph1_read(b0.data());
                for (int l = 0; l < 64; ++l) e[l].ph1_write(b1.data());
                for (int l = 0; l < 64; ++l) e[l].ph2_read(b1.data());
                for (int l = 0; l < 64; ++l) e[l].ph2_write(b0.data());
                if (MODE == MODE_STFT) {
                    for (int l = 0; l < 64; ++l) e[l].ph3_stft(a, t, b0.data());
                    continue;
                }
                for (int l = 0; l < 64; ++l) e[l].template ph3_update<SC, kRcArg>(a, t, b0.data());
            } else {
                for (int l = 0; l < 64; ++l) e[l].ph3_load(a, t);
            }
            for (int l = 0; l < 64; ++l) e[l].ph3_write(b1.data());
            for (int l = 0; l < 64; ++l) e[l].ph4(b1.data(), b0.data());
            for (int l = 0; l < 64; ++l) e[l].ph5_read(b0.data());
            for (int l = 0; l < 64; ++l) e[l].ph5_write(b1.data());
            for (int l = 0; l < 64; ++l) e[l].ph6(a, t, b1.data());
        }
        if (MODE != MODE_STFT)
            for (int l = 0; l < 64; ++l) e[l].flush_tails(a);
        if (SC && a.sc_partial) {
            float num = 0.f, den = 0.f;
            for (int l = 0; l < 64; ++l) {
                num += e[l].sc_num;
                den += e[l].sc_den;
            }
            a.sc_partial[2 * ri] = num;
            a.sc_partial[2 * ri + 1] = den;
        }
    }
}

struct Shape {
    int n_fft, hop, win_length;
};

template <int A, int HB>
struct Emu {
    using G = Geo<A>;
    static constexpr int R = 8 / HB, H = G::ROW * HB, P = G::P, F = G::P + 1;

    HostTables tab;
    std::vector<RunDesc> runs;
    std::vector<unsigned char> flags;
    std::vector<cf> spec;
    std::vector<float> S_native, ym[3], yt[3], sc_partial, zeros, trash;
    std::vector<int> nf;
    int batch, Ts, blocks;
    LayoutArgs L;
    IterArgs a;

    Emu(const Shape& s, const int* n_frames, int batch_, int frames_stride, int frames_per_run)
        : batch(batch_), Ts(frames_stride) {
        tab = build_tables<A>(s.hop, s.win_length);
        nf.assign(n_frames, n_frames + batch);
        runs = build_runs(n_frames, batch, frames_per_run, R);
        blocks = Ts + R;
        flags = build_tail_flags(runs, batch, blocks, R);
        spec.assign((size_t)batch * Ts * P, cf{0, 0});
        S_native.assign((size_t)batch * Ts * (P + 4), 0.f);
        for (int i = 0; i < 3; ++i) {
            ym[i].assign((size_t)batch * blocks * H, 0.f);
            yt[i].assign((size_t)batch * blocks * H, 0.f);
        }
        sc_partial.assign(2 * runs.size(), 0.f);
        L = LayoutArgs{s.n_fft, s.hop, F, P, R, Ts, (int64_t)blocks * H, nf.data(), 0, 0};
        memset(&a, 0, sizeof(a));
        a.runs = runs.data();
        a.n_runs = (int)runs.size();
        a.spec = spec.data();
        a.S = S_native.data();
        a.spec_utt_stride = (int64_t)Ts * P;
        a.s_utt_stride = (int64_t)Ts * (P + 4);
        a.y_utt_stride = (int64_t)blocks * H;
        a.w2 = tab.w2.data();
        a.iwss = tab.iwss.data();
        a.iwss_head = tab.iwss_head.data();
        a.iwss_tail = tab.iwss_tail.data();
        a.iwss_small = tab.iwss_small.data();
        zeros.assign(H, 0.f);
        trash.assign(H, 0.f);
        a.zeros = zeros.data();
        a.trash = trash.data();
        a.tables = tab.fft.data();
        a.inv_n = 1.0f / s.n_fft;
    }
    void io(int in, int out) {
        a.y_in = ym[in].data();
        a.y_in_tail = yt[in].data();
        a.y_out = ym[out].data();
        a.y_out_tail = yt[out].data();
    }
    void finalize(int buf, float* y, int64_t y_stride) {
        for (int b = 0; b < batch; ++b)
            for (int64_t n = 0; n < y_stride; ++n)
                finalize_sample(L, ym[buf].data(), yt[buf].data(), flags.data(), blocks, tab.w2.data(),
                                tab.iwss.data(), tab.iwss_head.data(), tab.iwss_tail.data(),
                                tab.iwss_small.data(), y, y_stride, b, n);
    }

    // one frame per wave: every frame is its own run, R partial-sum buffers per generation
    int griffinlim_single(const float* S, const float* phase0, uint64_t seed, int n_iter, float momentum,
                          float* y, int64_t y_stride, float* sc) {
        for (int b = 0; b < batch; ++b)
            for (int t = 0; t < nf[b]; ++t)
                for (int q = 0; q < P; ++q)
                    init_native_element<A>(L, S, reinterpret_cast<const cf*>(phase0), seed, a.inv_n,
                                           S_native.data(), spec.data(), b, t, q);
        std::vector<RunDesc> single;
        for (int b = 0; b < batch; ++b)
            for (int t = 0; t < nf[b]; ++t) single.push_back(RunDesc{b, t, t + 1, nf[b]});
        a.runs = single.data();
        a.n_runs = (int)single.size();
        const int64_t stride = (int64_t)batch * blocks * H;
        std::vector<float> parts[2];
        // garbage, not zeros: whatever a reader may use has been written by the launch before
        for (auto& p : parts) p.assign((size_t)R * stride, 1.0e30f);
        a.part_stride = stride;
        a.cmom = momentum / (1.0f + momentum);
        a.raw_in = 0;
        std::vector<float> scp(2 * single.size(), 0.f);
        a.part_out = parts[0].data();
        emu_launch<A, HB, MODE_ISTFT1, false>(a);
        int cur = 0;
        for (int k = 0; k < n_iter; ++k) {
            a.has_prev = k > 0;
            a.sc_partial = sc ? scp.data() : nullptr;
            a.part_in = parts[cur].data();
            a.part_out = parts[cur ^ 1].data();
            if (sc)
                emu_launch<A, HB, MODE_ITER1, true>(a);
            else
                emu_launch<A, HB, MODE_ITER1, false>(a);
            cur ^= 1;
            if (sc)
                for (int b = 0; b < batch; ++b) {
                    double num = 0, den = 0;
                    for (size_t r = 0; r < single.size(); ++r)
                        if (single[r].utt == b) {
                            num += scp[2 * r];
                            den += scp[2 * r + 1];
                        }
                    sc[(size_t)b * n_iter + k] = (float)std::sqrt(num / den);
                }
        }
        for (int b = 0; b < batch; ++b)
            for (int64_t n = 0; n < y_stride; ++n)
                finalize_sample_parts(L, parts[cur].data(), stride, tab.iwss.data(), tab.iwss_head.data(),
                                      tab.iwss_tail.data(), tab.iwss_small.data(), y, y_stride, b, n);
        a.runs = runs.data();
        a.n_runs = (int)runs.size();
        return 0;
    }

    int griffinlim(const float* S, const float* phase0, uint64_t seed, int n_iter, float momentum,
                   float* y, int64_t y_stride, float* sc, bool recompute, int linear = 0, bool deposit = false) {
        // deposit: run boundaries inside groups of 4 runs completed by the producer, where the
        // device kernels can do that (n_fft 1024, time-domain momentum, no convergence curve)
        deposit = deposit && A == 1 && linear != 0 && sc == nullptr;
        flags = build_tail_flags(runs, batch, blocks, R, deposit ? 4 : 0);
        a.deposit = deposit ? 1 : 0;
        // linear: 1 = momentum in the time domain (magnitudes in pair order), 2 = the same with
        // the upper quarter of every lane's bins known to be zero
        L.s_pair_nzh = linear == 1 ? 4 * A : (linear == 2 ? 2 * A : 0);
        for (int b = 0; b < batch; ++b)
            for (int t = 0; t < nf[b]; ++t)
                for (int q = 0; q < P; ++q)
                    init_native_element<A>(L, S, reinterpret_cast<const cf*>(phase0), seed, a.inv_n,
                                           S_native.data(), spec.data(), b, t, q);
        a.cmom = momentum / (1.0f + momentum);
        a.raw_in = 0;
        io(1, 0);
        emu_launch<A, HB, MODE_ISTFT, false>(a);
        int cur = 0;
        for (int k = 0; k < n_iter; ++k) {
            a.has_prev = k > 0;
            a.sc_partial = sc ? sc_partial.data() : nullptr;
            if (linear) {
                // three generations rotate; the first iteration has no older signal (c = 0,
                // the pointer only has to be readable)
                const int out = (cur + 1) % 3, prev = k == 0 ? cur : (cur + 2) % 3;
                io(cur, out);
                a.y_prev = ym[prev].data();
                a.y_prev_tail = yt[prev].data();
                if (linear == 2) {
                    if (sc) emu_launch<A, HB, MODE_ITER_LIN_Z, true>(a);
                    else emu_launch<A, HB, MODE_ITER_LIN_Z, false>(a);
                } else {
                    if (sc) emu_launch<A, HB, MODE_ITER_LIN, true>(a);
                    else emu_launch<A, HB, MODE_ITER_LIN, false>(a);
                }
                cur = out;
            } else if (recompute) {
                const int out = (cur + 1) % 3, prev = (cur + 2) % 3;
                io(cur, out);
                a.y_prev = ym[prev].data();
                a.y_prev_tail = yt[prev].data();
                if (k == 0) {
                    if (sc) emu_launch<A, HB, MODE_ITER_RC_FIRST, true>(a);
                    else emu_launch<A, HB, MODE_ITER_RC_FIRST, false>(a);
                } else {
                    if (sc) emu_launch<A, HB, MODE_ITER_RC, true>(a);
                    else emu_launch<A, HB, MODE_ITER_RC, false>(a);
                }
                cur = out;
            } else {
                io(cur, cur ^ 1);
                if (sc)
                    emu_launch<A, HB, MODE_ITER, true>(a);
                else
                    emu_launch<A, HB, MODE_ITER, false>(a);
                cur ^= 1;
            }
            if (sc) {
                for (int b = 0; b < batch; ++b) {
                    double num = 0, den = 0;
                    for (size_t r = 0; r < runs.size(); ++r)
                        if (runs[r].utt == b) {
                            num += sc_partial[2 * r];
                            den += sc_partial[2 * r + 1];
                        }
                    sc[(size_t)b * n_iter + k] = (float)std::sqrt(num / den);
                }
            }
        }
        finalize(cur, y, y_stride);
        return 0;
    }

    int stft(const float* y, int64_t y_stride, float* D) {
        for (int b = 0; b < batch; ++b) {
            const int64_t len = (int64_t)H * (nf[b] - 1);
            for (int64_t n = 0; n < len; ++n)
                ym[0][(size_t)b * a.y_utt_stride + s_nfft() / 2 + n] = y[(size_t)b * y_stride + n];
        }
        a.raw_in = 1;
        io(0, 1);
        emu_launch<A, HB, MODE_STFT, false>(a);
        for (int b = 0; b < batch; ++b)
            for (int t = 0; t < nf[b]; ++t)
                for (int q = 0; q < P; ++q)
                    spec_to_natural_element<A>(L, spec.data(), reinterpret_cast<cf*>(D), b, t, q);
        return 0;
    }
    int istft(const float* D, float* y, int64_t y_stride) {
        for (int b = 0; b < batch; ++b)
            for (int t = 0; t < nf[b]; ++t)
                for (int q = 0; q < P; ++q)
                    spec_to_native_element<A>(L, reinterpret_cast<const cf*>(D), a.inv_n, spec.data(), b, t, q);
        a.raw_in = 0;
        io(1, 0);
        emu_launch<A, HB, MODE_ISTFT, false>(a);
        finalize(0, y, y_stride);
        return 0;
    }
    static int s_nfft() { return G::N; }
};

template <typename Fn>
int dispatch(int n_fft, int hop, Fn&& fn) {
    const int row = n_fft / 8;  // samples per row of the engine's time layout
    if (n_fft <= 0 || hop % row != 0) return -2;
    const int hb = hop / row;
    if (n_fft == 1024 && hb == 1) return fn((Emu<1, 1>*)nullptr);
    if (n_fft == 1024 && hb == 2) return fn((Emu<1, 2>*)nullptr);
    if (n_fft == 1024 && hb == 4) return fn((Emu<1, 4>*)nullptr);
    if (n_fft == 2048 && hb == 1) return fn((Emu<2, 1>*)nullptr);
    if (n_fft == 2048 && hb == 2) return fn((Emu<2, 2>*)nullptr);
    if (n_fft == 2048 && hb == 4) return fn((Emu<2, 4>*)nullptr);
    return -2;
}

}  // namespace

extern "C" {

int emu_griffinlim(int n_fft, int hop, int win_length, const float* S, const float* phase0,
                   uint64_t seed, const int* n_frames, int batch, int frames_stride, int n_iter,
                   float momentum, int frames_per_run, float* y, int64_t y_stride, float* sc, int recompute) {
    // recompute: 0 = run-based iteration, 1 = its recompute-the-previous-spectrum variant,
    //            2 = one frame per wave, 3 = momentum in the time domain, 4 = the same with the
    //            magnitudes of the upper quarter of every lane's registers known to be zero,
    //            5 / 6 = 3 / 4 with the run boundaries inside a workgroup completed by the producer
    return dispatch(n_fft, hop, [&](auto* tag) {
        using E = std::remove_pointer_t<decltype(tag)>;
        E e(Shape{n_fft, hop, win_length}, n_frames, batch, frames_stride, frames_per_run);
        if (recompute == 2) return e.griffinlim_single(S, phase0, seed, n_iter, momentum, y, y_stride, sc);
        if (recompute == 3) return e.griffinlim(S, phase0, seed, n_iter, momentum, y, y_stride, sc, false, 1);
        if (recompute == 4) return e.griffinlim(S, phase0, seed, n_iter, momentum, y, y_stride, sc, false, 2);
        if (recompute == 5) return e.griffinlim(S, phase0, seed, n_iter, momentum, y, y_stride, sc, false, 1, true);
        if (recompute == 6) return e.griffinlim(S, phase0, seed, n_iter, momentum, y, y_stride, sc, false, 2, true);
        return e.griffinlim(S, phase0, seed, n_iter, momentum, y, y_stride, sc, recompute != 0);
    });
}

// slot of native position (reg, lane) in a frame of magnitudes stored in pair order
int emu_pair_s_slot(int n_fft, int nzh, int reg, int lane) {
    return n_fft == 1024 ? pair_s_slot<1>(nzh, reg, lane) : pair_s_slot<2>(nzh, reg, lane);
}

// u[i] = the generator's value for element idx0 + i (the device-side initial phase stream)
void emu_counter_uniform(uint64_t seed, uint64_t idx0, int n, float* u) {
    for (int i = 0; i < n; ++i) u[i] = counter_uniform(seed, idx0 + (uint64_t)i);
}

// the run table of build_runs_target: fills out[4*i .. 4*i+3] = utt, t_begin, t_end, n_frames;
// returns the number of runs (at most max_out are written)
int emu_runs_target(const int* n_frames, int batch, int64_t target, int R, int* out, int max_out) {
    const std::vector<RunDesc> runs = build_runs_target(n_frames, batch, target, R);
    for (size_t i = 0; i < runs.size() && (int)i < max_out; ++i) {
        out[4 * i] = runs[i].utt;
        out[4 * i + 1] = runs[i].t_begin;
        out[4 * i + 2] = runs[i].t_end;
        out[4 * i + 3] = runs[i].n_frames;
    }
    return (int)runs.size();
}

int emu_stft(int n_fft, int hop, int win_length, const float* y, int64_t y_stride, const int* n_frames,
             int batch, int frames_stride, int frames_per_run, float* D) {
    return dispatch(n_fft, hop, [&](auto* tag) {
        using E = std::remove_pointer_t<decltype(tag)>;
        E e(Shape{n_fft, hop, win_length}, n_frames, batch, frames_stride, frames_per_run);
        return e.stft(y, y_stride, D);
    });
}

int emu_istft(int n_fft, int hop, int win_length, const float* D, const int* n_frames, int batch,
              int frames_stride, int frames_per_run, float* y, int64_t y_stride) {
    return dispatch(n_fft, hop, [&](auto* tag) {
        using E = std::remove_pointer_t<decltype(tag)>;
        E e(Shape{n_fft, hop, win_length}, n_frames, batch, frames_stride, frames_per_run);
        return e.istft(D, y, y_stride);
    });
}

// the number of runs the automatic run table aims at (pick_run_target)
int64_t emu_pick_run_target(int64_t total_frames, int cus, int waves_per_block, int blocks_per_cu, int min_len) {
    return pick_run_target(total_frames, cus, waves_per_block, blocks_per_cu, min_len);
}
}
