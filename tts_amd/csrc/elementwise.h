// Element-wise stages around the wave engine, one output element per call.
// Device kernels (glvoc.hip) and the host emulation (tests/csrc) both loop over
// these functions.
#pragma once

#include "engine.h"

namespace glv {

struct LayoutArgs {
    int n_fft, hop, F, P, R;
    int frames_stride;        // frames per utterance row in caller arrays and in spec/S
    int64_t y_utt_stride;     // floats per utterance in the padded overlap-add buffers
    const int* n_frames;      // [batch] frames of the whole utterance
    int frames_local;         // > 0: only this many frames are held here (time shard)
    int64_t frame_origin;     // global index of local frame 0 (time shard), else 0
    const int64_t* utt_ids;   // [batch] identity of every utterance for the device-drawn phase, or
                              // null: its position in the batch
    int s_pair_nzh;           // 0: magnitudes in native order; else pair order (pair_s_slot) with
                              // that many possibly non-zero mirrors per lane
    int block_origin;         // time shard: the overlap-add buffers and tail flags start at this block
                              // of the utterance, and `out` at sample out_origin of the waveform
    int64_t out_origin;
};

// Device-drawn initial phase of (utterance b, frame t, bin k): the stream is keyed by the
// utterance's identity and indexed by (global frame, bin), so an utterance gets the same phase
// whatever batch it travels in, wherever it sits there and whatever the row stride is.
GLV_HD cf drawn_phase(const LayoutArgs& L, uint64_t seed, int b, int t, int k) {
    const uint64_t id = L.utt_ids ? (uint64_t)L.utt_ids[b] : (uint64_t)b;
    return counter_phasor(seed ^ (id * 0x9E3779B97F4A7C15ull), (uint64_t)((t + L.frame_origin) * L.F + k));
}

// --- natural (B, T, F) arrays -> native state for the iteration -------------
// One call per (utterance b, frame t, native position q).  angles = phase * S is
// stored pre-scaled by 1/n_fft (the inverse transform is unnormalised).
template <int A>
GLV_HD void init_native_element(const LayoutArgs& L, const float* S_nat, const cf* phase_nat,
                                uint64_t seed, float inv_n, float* S_native, cf* spec, int b, int t,
                                int q) {
    const int reg = q >> 6, lane = q & 63;
    const int k = Geo<A>::bin(reg, lane);
    const int64_t fr = (int64_t)b * L.frames_stride + t;
    const float* srow = S_nat + fr * L.F;
    float* snat = S_native + fr * (L.P + 4);
    const float s = srow[k];
    const int slot = L.s_pair_nzh ? pair_s_slot<A>(L.s_pair_nzh, reg, lane) : native_s_index(reg, lane);
    if (slot >= 0) snat[slot] = s;
    auto phase_at = [&](int kk) -> cf {
        if (phase_nat) return phase_nat[fr * L.F + kk];
        return drawn_phase(L, seed, b, t, kk);
    };
    if (q == 0) {
        snat[L.P] = srow[L.P];
        if (!L.s_pair_nzh) snat[L.P + 1] = 0.f;  // pair order keeps the bin P/2 there
        snat[L.P + 2] = snat[L.P + 3] = 0.f;
    }
    if (!spec) return;  // the first inverse transform draws the phase itself (Engine::ph3_draw)
    const cf p = phase_at(k);
    cf ang = cf{p.x * s * inv_n, p.y * s * inv_n};
    if (q == 0) {
        const cf pn = phase_at(L.P);
        ang = cf{p.x * s * inv_n, pn.x * srow[L.P] * inv_n};  // c2r ignores Im of DC / Nyquist
    }
    spec[fr * L.P + native_spec_index(reg, lane)] = ang;
}

// natural complex spectrum -> native (for glv_istft), scaled by 1/n_fft
template <int A>
GLV_HD void spec_to_native_element(const LayoutArgs& L, const cf* D_nat, float inv_n, cf* spec, int b,
                                   int t, int q) {
    const int reg = q >> 6, lane = q & 63;
    const int k = Geo<A>::bin(reg, lane);
    const int64_t fr = (int64_t)b * L.frames_stride + t;
    cf d = D_nat[fr * L.F + k];
    if (q == 0) d = cf{d.x, D_nat[fr * L.F + L.P].x};
    spec[fr * L.P + native_spec_index(reg, lane)] = cf{d.x * inv_n, d.y * inv_n};
}

// native -> natural complex spectrum (for glv_stft)
template <int A>
GLV_HD void spec_to_natural_element(const LayoutArgs& L, const cf* spec, cf* D_nat, int b, int t, int q) {
    const int reg = q >> 6, lane = q & 63;
    const int k = Geo<A>::bin(reg, lane);
    const int64_t fr = (int64_t)b * L.frames_stride + t;
    const cf d = spec[fr * L.P + native_spec_index(reg, lane)];
    if (q == 0) {
        D_nat[fr * L.F] = cf{d.x, 0.f};
        D_nat[fr * L.F + L.P] = cf{d.y, 0.f};
    } else {
        D_nat[fr * L.F + k] = d;
    }
}

// --- finished overlap-add sums -> output waveform ---------------------------
// out[b][n] for n < y_stride; returns |value| for the peak search.
GLV_HD float finalize_sample(const LayoutArgs& L, const float* y_main, const float* y_tail,
                             const unsigned char* tail_flag, int blocks_stride, const float* w2,
                             const float* iwss, const float* iwss_head, const float* iwss_tail,
                             const float* iwss_small, float* out, int64_t out_stride, int b, int64_t n) {
    const int T = L.n_frames[b];
    const int64_t len = (int64_t)L.hop * (T - 1);
    float val = 0.f;
    if (n < len) {
        const int64_t p = n + L.n_fft / 2;
        const int j = (int)(p / L.hop), s = (int)(p % L.hop);
        const int64_t at = (int64_t)b * L.y_utt_stride + p - (int64_t)L.block_origin * L.hop;
        if (j < T) val = y_main[at];
        if (tail_flag[(int64_t)b * blocks_stride + j - L.block_origin]) val += y_tail[at];
        const float iw = iwss_row(iwss, iwss_head, iwss_tail, iwss_small, L.hop, L.R, T, j)[s];
        val *= iw;
    }
    out[(int64_t)b * out_stride + n - L.out_origin] = val;
    return fabsf(val);
}

// the same for the one-frame-per-wave modes: the sample is the sum of the R partial-sum buffers
// (frames ascending), see Engine::load_frame_from_parts
GLV_HD float finalize_sample_parts(const LayoutArgs& L, const float* parts, int64_t part_stride,
                                   const float* iwss, const float* iwss_head, const float* iwss_tail,
                                   const float* iwss_small, float* out, int64_t out_stride, int b, int64_t n) {
    const int T = L.n_frames[b];
    const int64_t len = (int64_t)L.hop * (T - 1);
    float val = 0.f;
    if (n < len) {
        const int64_t p = n + L.n_fft / 2;
        const int j = (int)(p / L.hop), s = (int)(p % L.hop);
        const int64_t at = (int64_t)b * L.y_utt_stride + p;
        for (int d = L.R - 1; d >= 0; --d) {
            const int tc = j - d;
            if (tc >= 0 && tc < T) val += parts[(int64_t)(tc & (L.R - 1)) * part_stride + at];
        }
        val *= iwss_row(iwss, iwss_head, iwss_tail, iwss_small, L.hop, L.R, T, j)[s];
    }
    out[(int64_t)b * out_stride + n] = val;
    return fabsf(val);
}

}  // namespace glv
