// Host-side builders shared by libglvoc and the CPU emulation used in tests:
// window / twiddle tables, window-sum-square reciprocals and the run table that
// maps wavefronts to frame ranges.
#pragma once

#include <float.h>
#include <math.h>

#include <algorithm>
#include <utility>
#include <vector>

#include "engine.h"

namespace glv {

struct HostTables {
    std::vector<cf> fft;      // twA | twB | twC | twP | win   (Geo<A>::TABLES)
    std::vector<float> w2;    // squared window, N
    std::vector<float> iwss;  // reciprocal complete window-sum-square, H
    std::vector<float> iwss_head;  // [R-1][H] blocks 0..R-2: only frames 0..j exist
    std::vector<float> iwss_tail;  // [R-1][H] blocks T..T+R-2: frames past T-1 are missing
    std::vector<float> iwss_small;  // [R-1][2R][H] indexed [T][j]: utterances of T <= R-2 frames
};

// Periodic Hann window of win_length, zero-padded centrally to n_fft
// (scipy.signal.get_window('hann', win_length, fftbins=True) + librosa.util.pad_center).
inline std::vector<double> padded_hann(int n_fft, int win_length) {
    std::vector<double> w(n_fft, 0.0);
    const int lpad = (n_fft - win_length) / 2;
    const double pi = 3.14159265358979323846;
    for (int n = 0; n < win_length; ++n) w[lpad + n] = 0.5 - 0.5 * cos(2.0 * pi * n / win_length);
    return w;
}

template <int A>
HostTables build_tables(int hop, int win_length) {
    using G = Geo<A>;
    const double pi = 3.14159265358979323846;
    HostTables t;
    t.fft.resize(G::TABLES);
    cf* twA = t.fft.data();
    cf* twB = twA + G::TW_A;
    cf* twC = twB + G::TW_B;
    cf* twP = twC + G::TW_C;
    cf* win = twP + G::TW_P;
    for (int k0 = 1; k0 < 8; ++k0)
        for (int lane = 0; lane < 64; ++lane) {
            const double th = 2.0 * pi * (double)(k0 * lane) / 512.0;
            twA[G::pair(k0, lane)] = cf{(float)cos(th), (float)-sin(th)};
        }
    for (int k1 = 1; k1 < 8; ++k1)
        for (int n0 = 0; n0 < 8; ++n0) {
            const double th = 2.0 * pi * (double)(k1 * n0) / 64.0;
            twB[G::pair8(k1, n0)] = cf{(float)cos(th), (float)-sin(th)};
        }
    for (int i = 0; i < G::TW_C; ++i) {
        const double th = 2.0 * pi * (double)G::bin(i >> 6, i & 63) / G::P;
        twC[G::pair(i >> 6, i & 63)] = cf{(float)cos(th), (float)-sin(th)};
    }
    for (int reg = 0; reg < G::E; ++reg)
        for (int lane = 0; lane < 64; ++lane) {
            const double th = 2.0 * pi * (double)G::bin(reg, lane) / G::N;
            twP[G::pair(reg, lane)] = cf{(float)(0.5 * cos(th)), (float)(-0.5 * sin(th))};
        }
    const std::vector<double> w = padded_hann(G::N, win_length);
    for (int reg = 0; reg < G::E; ++reg)
        for (int lane = 0; lane < 64; ++lane) {
            const int s = G::sample(reg, lane);
            win[G::pair(reg, lane)] = cf{(float)w[s], (float)w[s + 1]};
        }
    t.w2.resize(G::N);
    for (int n = 0; n < G::N; ++n) t.w2[n] = (float)(w[n] * w[n]);
    // librosa.filters.window_sumsquare accumulates float64 squares into a float32
    // line, one frame after the other; a fully covered sample sees the frames in
    // ascending order, i.e. window offsets in descending order.
    const int R = G::N / hop;
    t.iwss.resize(hop);
    for (int s = 0; s < hop; ++s) {
        float acc = 0.f;
        for (int u = R - 1; u >= 0; --u) acc = (float)((double)acc + w[u * hop + s] * w[u * hop + s]);
        t.iwss[s] = acc > FLT_MIN ? 1.0f / acc : 1.0f;
    }
    // utterance edges: block j < R-1 is covered by frames 0..j only (offsets j..0);
    // block T-1+d, d = 1..R-1, by frames up to T-1 only (offsets R-1..d)
    t.iwss_head.assign((size_t)std::max(1, R - 1) * hop, 1.0f);
    t.iwss_tail.assign((size_t)std::max(1, R - 1) * hop, 1.0f);
    for (int j = 0; j < R - 1; ++j)
        for (int s = 0; s < hop; ++s) {
            float acc = 0.f;
            for (int u = j; u >= 0; --u) acc = (float)((double)acc + w[u * hop + s] * w[u * hop + s]);
            t.iwss_head[(size_t)j * hop + s] = acc > FLT_MIN ? 1.0f / acc : 1.0f;
        }
    for (int d = 1; d <= R - 1; ++d)
        for (int s = 0; s < hop; ++s) {
            float acc = 0.f;
            for (int u = R - 1; u >= d; --u) acc = (float)((double)acc + w[u * hop + s] * w[u * hop + s]);
            t.iwss_tail[(size_t)(d - 1) * hop + s] = acc > FLT_MIN ? 1.0f / acc : 1.0f;
        }
    // very short utterances (T <= R-2): a block can miss frames on both sides at once
    const int rows = std::max(1, R - 1);
    t.iwss_small.assign((size_t)rows * 2 * R * hop, 1.0f);
    for (int T = 1; T <= R - 2; ++T)
        for (int j = 0; j < 2 * R; ++j)
            for (int s = 0; s < hop; ++s) {
                float acc = 0.f;
                const int t0 = std::max(0, j - R + 1), t1 = std::min(T - 1, j);
                for (int tt = t0; tt <= t1; ++tt)
                    acc = (float)((double)acc + w[(j - tt) * hop + s] * w[(j - tt) * hop + s]);
                t.iwss_small[((size_t)T * 2 * R + j) * hop + s] = acc > FLT_MIN ? 1.0f / acc : 1.0f;
            }
    return t;
}

// Splits every utterance into runs of about frames_per_run consecutive frames.
// A run is never shorter than R-1 frames (unless it is the whole utterance), so
// the tail blocks of one run fall inside the head blocks of the next.
inline std::vector<RunDesc> build_runs(const int* n_frames, int batch, int frames_per_run, int R) {
    std::vector<RunDesc> runs;
    for (int b = 0; b < batch; ++b) {
        const int T = n_frames[b];
        if (T <= 0) continue;
        int nr = (T + frames_per_run / 2) / std::max(1, frames_per_run);
        const int min_len = std::max(1, R - 1);
        nr = std::max(1, std::min(nr, T / min_len));
        for (int i = 0; i < nr; ++i) {
            const int t0 = (int)((int64_t)i * T / nr), t1 = (int)((int64_t)(i + 1) * T / nr);
            runs.push_back(RunDesc{b, t0, t1, T});
        }
    }
    return runs;
}

// The same with the NUMBER of runs given: `target` runs are dealt to the utterances in
// proportion to their frames (largest remainders first), so that a launch can fill a whole
// number of wave slots.  An utterance still gets at most T / (R-1) runs and at least one.
inline std::vector<RunDesc> build_runs_target(const int* n_frames, int batch, int64_t target, int R) {
    int64_t total = 0;
    for (int b = 0; b < batch; ++b) total += std::max(0, n_frames[b]);
    const int min_len = std::max(1, R - 1);
    std::vector<int> nr(batch, 0);
    std::vector<std::pair<double, int>> rest;
    int64_t given = 0;
    for (int b = 0; b < batch; ++b) {
        const int T = n_frames[b];
        if (T <= 0) continue;
        const double share = total > 0 ? (double)target * T / (double)total : 1.0;
        const int cap = std::max(1, T / min_len);
        nr[b] = std::max(1, std::min((int)share, cap));
        given += nr[b];
        if (nr[b] < cap) rest.push_back({share - (int)share, b});
    }
    std::sort(rest.begin(), rest.end(), [](const std::pair<double, int>& x, const std::pair<double, int>& y) {
        return x.first != y.first ? x.first > y.first : x.second < y.second;
    });
    // one more run each, largest remainders first; again while utterances with room are left
    for (bool any = true; any && given < target;) {
        any = false;
        for (size_t i = 0; i < rest.size() && given < target; ++i) {
            const int b = rest[i].second;
            if (nr[b] >= std::max(1, n_frames[b] / min_len)) continue;
            nr[b]++;
            given++;
            any = true;
        }
    }
    std::vector<RunDesc> runs;
    for (int b = 0; b < batch; ++b) {
        const int T = n_frames[b];
        for (int i = 0; i < nr[b]; ++i) {
            const int t0 = (int)((int64_t)i * T / nr[b]), t1 = (int)((int64_t)(i + 1) * T / nr[b]);
            runs.push_back(RunDesc{b, t0, t1, T});
        }
    }
    return runs;
}

// How many runs the automatic run table should aim at when the batch fits the device's wave
// slots in one round (total frames <= 64 x slots).  Measured on MI355X (launch time of the
// iteration kernel against the target, batches of 8..64 utterances, both n_fft): a wave that
// walks L frames costs about (L + 4) frame times (preloading the blocks before its first
// frame, finishing the blocks after its last), and a frame time grows with the workgroups
// resident on a CU as 1 : 1.32 : 1.63 for 1 : 2 : 3.  Launches whose workgroups are a whole
// multiple of the CUs come out best; between multiples the slowest CU carries the next whole
// number while the others run lighter, which the average of the two figures describes within
// 1-2 % of the measurements.  The candidates are k x CUs workgroups for every k that leaves
// runs of at least min_len frames, and the largest table the minimum length allows.
inline int64_t pick_run_target(int64_t total_frames, int cus, int waves_per_block, int blocks_per_cu, int min_len) {
    const int64_t slots = (int64_t)cus * waves_per_block * blocks_per_cu;
    const int64_t cap = std::max<int64_t>(1, total_frames / std::max(1, min_len));
    auto frame_time = [](double k) { return 1.0 + 0.315 * (k - 1.0); };
    auto cost = [&](int64_t runs) {
        const double per_cu = (double)((runs + waves_per_block - 1) / waves_per_block) / cus;
        const double k = std::max(1.0, per_cu);
        const double t = 0.5 * (frame_time(std::ceil(k)) + frame_time(k));
        return ((double)total_frames / (double)runs + 4.0) * t;
    };
    int64_t best = std::min(cap, slots);
    double best_cost = cost(best);
    for (int k = 1; k <= blocks_per_cu; ++k) {
        const int64_t runs = (int64_t)k * cus * waves_per_block;
        if (runs > cap) break;
        const double c = cost(runs);
        if (c < best_cost) {
            best_cost = c;
            best = runs;
        }
    }
    return best;
}

// Run i completes its last R-1 blocks itself (Engine::deposit_in): the next run continues its
// utterance and is a wave of the same workgroup of `waves_per_block` consecutive runs.
inline bool run_completes_its_tail(const std::vector<RunDesc>& runs, size_t i, int waves_per_block) {
    return waves_per_block > 1 && (int)(i % (size_t)waves_per_block) + 1 < waves_per_block && i + 1 < runs.size() &&
           runs[i + 1].utt == runs[i].utt;
}

// tail_flag[b*blocks_stride + j] = 1 when block j of utterance b is completed by
// adding the tail buffer (first R-1 blocks after an interior run boundary, and
// the R-1 blocks past the last frame).  deposit_waves > 0: the boundaries that the producer
// completes inside workgroups of that many runs are not among them.
inline std::vector<unsigned char> build_tail_flags(const std::vector<RunDesc>& runs, int batch,
                                                   int blocks_stride, int R, int deposit_waves = 0) {
    std::vector<unsigned char> f((size_t)batch * blocks_stride, 0);
    for (size_t i = 0; i < runs.size(); ++i) {
        const RunDesc& r = runs[i];
        if (deposit_waves > 0 && run_completes_its_tail(runs, i, deposit_waves)) continue;
        for (int q = 0; q < R - 1; ++q) {
            const int j = r.t_end + q;
            if (j < blocks_stride) f[(size_t)r.utt * blocks_stride + j] = 1;
        }
    }
    return f;
}

}  // namespace glv
