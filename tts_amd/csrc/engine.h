// Per-wavefront Griffin-Lim engine: one wave walks a run of consecutive frames of
// one utterance and, per frame, performs
//     STFT frame -> momentum phase update -> inverse FFT -> overlap-add
// entirely in registers (plus the wave-private LDS exchanges of wave_fft.h).
//
// Overlap-add without atomics or barriers: consecutive frames overlap by
// N - H samples, so the wave keeps the N-sample running sum in registers
// (`acc`), adds each windowed frame, emits the H oldest samples (now complete)
// and shifts by H.  H = HB rows of 128*A samples (see wave_fft.h) is exactly HB registers
// of every sub-sequence per lane, so the shift is register renaming.  The same trick
// slides the STFT input (`raw`).
//
// Run boundaries: the first R-1 blocks a run emits lack the contributions of
// the previous run's last frames ("heads"); the R-1 blocks left in `acc`
// after a run's last frame lack those of the next run ("tails").  Heads go to
// the main buffer like complete blocks, tails to a second buffer, and whoever
// reads a boundary block adds the two.  Division by the window-sum-square
// happens at read time.  Buffers are indexed in padded coordinates:
// block j = samples [j*H, (j+1)*H) of the N + H*(T-1) sample overlap-add line;
// the signal proper is blocks [R/2, R/2 + T-1).
//
// Everything here compiles for the host as well (lanes become a loop in
// tests/csrc), see wave_fft.h.
#pragma once

#include <math.h>
#include <stdint.h>

#include "wave_fft.h"

// Build-time tuning switch: fetch the previous spectrum a whole frame ahead of its use
// (E more live registers) for n_fft 2048 as well, or only on demand.
#ifndef GLV_PREFETCH_TP_ALL
#define GLV_PREFETCH_TP_ALL 1
#endif
// Registers of the split / phase-update step handled between two scheduling fences
// (n_fft 2048 only): bounds how many mirror partners and previous-spectrum values are live.
#ifndef GLV_SPLIT_CHUNK
#define GLV_SPLIT_CHUNK 8
#endif
#ifndef GLV_SPLIT_CHUNK_A1
#define GLV_SPLIT_CHUNK_A1 8
#endif
// Keeps the compiler from moving instructions across this point (it would otherwise hoist
// every load of a phase to its top and hold all results in registers at once).
#if defined(__HIP_DEVICE_COMPILE__)
#define GLV_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// the same for memory operations at every level of the compiler: nothing that touches memory
// moves across (no instruction is emitted)
#define GLV_MEM_FENCE() asm volatile("" ::: "memory")
// ... and with a value that has to be in its register by then (pins the arithmetic producing
// it, and the loads that feeds, in front of the fence)
#define GLV_PIN(x) asm volatile("" : "+v"(x)::"memory")
#else
#define GLV_SCHED_FENCE() ((void)0)
#define GLV_MEM_FENCE() ((void)0)
#define GLV_PIN(x) ((void)0)
#endif
namespace glv {


struct RunDesc {
    int utt, t_begin, t_end, n_frames;
};

// MODE_TRAFFIC: the iteration kernel's loads and stores with the arithmetic removed - a
// measurement aid that gives the memory-system ceiling of this access pattern.
// MODE_ITER_RC / MODE_ITER_RC_FIRST: the iteration without the spectrum state in HBM - the
// previous rebuilt spectrum (the momentum term) is recomputed from the previous signal,
// which costs one more forward FFT per frame and saves reading and writing 8P bytes.
// MODE_ITER_LIN: the momentum term taken in the time domain.  The STFT is linear, so
// rebuilt_k - c * rebuilt_{k-1} = STFT(y_k - c * y_{k-1}): one forward transform of the combined
// signal replaces the spectrum state (read 8P, write 8P bytes per frame) by one more read of
// H samples.  Same arithmetic up to rounding; the convergence metric, which needs |rebuilt_k|
// itself, costs a second forward transform only when it is asked for.
// The split and the phase update of MODE_ITER_LIN work on mirror pairs of bins (wave_fft.h, "the
// split done on mirror PAIRS"), with the target magnitudes stored in PAIR ORDER (pair_s_slot).
// MODE_ITER_LIN_Z is the same for magnitudes known to vanish in the upper quarter of every
// lane's registers (a mel basis that ends below the Nyquist frequency makes its pseudo-inverse,
// and so the magnitudes, exactly zero there): those bins are neither fetched nor updated.
// MODE_ITER1 / MODE_ISTFT1: one frame per wave, for batches too small to fill the device with
// runs.  There is no running sum: a frame's windowed output goes, as R whole blocks, to
// partial-sum buffer (t mod R), and a reader adds the R buffers (see load_frame_from_parts).
enum Mode {
    MODE_ITER = 0,
    MODE_ISTFT = 1,
    MODE_STFT = 2,
    MODE_TRAFFIC = 3,
    MODE_ITER_RC = 4,
    MODE_ITER_RC_FIRST = 5,
    MODE_ITER1 = 6,
    MODE_ISTFT1 = 7,
    MODE_ITER_LIN = 8,
    MODE_ITER_LIN_Z = 9,
    // the memory operations of MODE_ITER_LIN / MODE_ITER_LIN_Z with the transforms removed (a
    // measurement aid like MODE_TRAFFIC: the memory-system ceiling of the default iteration)
    MODE_TRAFFIC_LIN = 10,
    MODE_TRAFFIC_LIN_Z = 11,
    // MODE_ITER_LIN / MODE_ITER_LIN_Z with ALL iterations of a call in one launch: every wave
    // keeps its run for the whole call and runs whose boundary blocks travel through memory wait
    // for their neighbours' progress words (see "iteration-spanning" in kernels.h)
    MODE_ITER_LIN_P = 12,
    MODE_ITER_LIN_Z_P = 13,
    // MODE_ITER1 with all iterations of a call in one launch: a wave keeps its frame and waits,
    // before every iteration, for the progress words of the 2(R-1) frames around it
    MODE_ITER1_P = 14
};

struct IterArgs {
    const RunDesc* runs;
    int n_runs;
    cf* spec;                 // native spectra: tprev / angles, [utt][frame][P]
    const float* S;           // native target magnitudes, [utt][frame][P+4] (Nyquist at [P])
    const float* y_in;        // previous overlap-add sums (or padded signal if raw_in)
    const float* y_in_tail;
    const float* y_prev;      // the signal one iteration older than y_in (MODE_ITER_RC)
    const float* y_prev_tail;
    float* y_out;
    float* y_out_tail;
    const float* part_in;     // R partial-sum buffers of the previous iteration (MODE_ITER1), laid
    float* part_out;          //   out like y_in, part_stride floats apart; and those being written
    int64_t part_stride;
    int64_t spec_utt_stride;  // cf units
    int64_t s_utt_stride;     // floats
    int64_t y_utt_stride;     // floats
    const float* w2;          // squared window, N floats
    const float* iwss;        // reciprocal of the complete window-sum-square, H floats
    const float* iwss_head;   // [R-1][H]: blocks 0..R-2 (frames before the first are missing)
    const float* iwss_tail;   // [R-1][H]: blocks T..T+R-2 (frames after the last are missing)
    const float* iwss_small;  // [R-1][2R][H]: utterances shorter than R-1 frames (both sides missing)
    const float* zeros;       // H zero floats: source of blocks that do not exist
    float* trash;             // H floats nobody reads: sink of blocks that do not exist
    const cf* tables;         // twA | twB | twC | twP | win
    float* sc_partial;        // [run][2] (numerator, denominator) or null
    float cmom;               // momentum / (1 + momentum)
    float inv_n;              // 1 / n_fft
    int has_prev;             // 0 on the first iteration (no momentum term)
    int raw_in;               // y_in holds the padded signal itself (glv_stft)
    int deposit;              // 1: run boundaries inside a workgroup are completed by the producer
                              //    (Engine::deposit_out / deposit_in); every generation of the call
                              //    and the tail flags follow the same convention
    // MODE_ISTFT / MODE_ISTFT1 without a stored initial spectrum: angles = S * phase drawn here
    int draw;                 // 1: `spec` holds nothing, the initial phase is drawn in the kernel
    int draw_nzh;             // layout of S: 0 native order, else pair order (pair_s_slot)
    uint64_t draw_seed;
    const int64_t* draw_ids;  // [batch] utterance identities of the phase stream, or null: the row
    int64_t draw_origin;      // global index of local frame 0
    // Time shard: runs, frames and blocks are numbered over the whole utterance, while spec, S and
    // the overlap-add buffers hold only the shard - index 0 there is frame / block t_origin.
    int t_origin;
    // MODE_ITER_LIN over a subset of the run table's workgroups: workgroup b of the launch takes
    // the runs of workgroup wg_first + b * (1 + wg_skip) of the table (0, 0: all of them in order)
    int wg_first, wg_skip;
    // Iteration-spanning kernels (MODE_ITER_LIN_P / _Z_P): iteration k reads generation k % 3 of
    // the overlap-add buffers (and (k + 2) % 3 as the older signal), writes (k + 1) % 3.
    int n_iter;
    float* gen_main[3];
    float* gen_tail[3];
    int* progress;  // [n_runs] words kProgressStride ints apart, zero at launch: iterations whose
                    //   boundary blocks the run has written through to memory
    int* status;    // one word, zero at launch: set when a wait ran out (the results are void then)
    int* host_status;  // the same in host memory the device can write; never cleared by a launch
    int census;     // 1: only count the workgroups that are resident together (nothing is computed)
};
// distance between two runs' progress words: a word is stored by one wave and polled by at most two
constexpr int kProgressStride = 32;
// offset of a run's second progress word: iterations whose first R-1 blocks (the half sums its
// predecessor reads) have been written through
constexpr int kProgressHeads = 16;

// Loads and stores of the iteration state.  Every byte of it is touched once per launch and not
// again before the whole working set has passed through the caches, so the accesses can be
// marked non-temporal (GLV_STREAM_NT bit 0: loads, bit 1: stores); plain accesses otherwise.
#ifndef GLV_PARTS_DEPTH
#define GLV_PARTS_DEPTH 2
#endif
#ifndef GLV_STREAM_NT
#define GLV_STREAM_NT 0
#endif
#if defined(__HIP_DEVICE_COMPILE__) && GLV_STREAM_NT
typedef float glv_v2f __attribute__((ext_vector_type(2)));
typedef float glv_v4f __attribute__((ext_vector_type(4)));
#endif
#if defined(__HIP_DEVICE_COMPILE__) && (GLV_STREAM_NT & 1)
__device__ __forceinline__ cf ld_stream(const cf* p) {
    const glv_v2f v = __builtin_nontemporal_load(reinterpret_cast<const glv_v2f*>(p));
    return cf{v.x, v.y};
}
__device__ __forceinline__ f4 ld_stream(const f4* p) {
    const glv_v4f v = __builtin_nontemporal_load(reinterpret_cast<const glv_v4f*>(p));
    return f4{v.x, v.y, v.z, v.w};
}
#else
GLV_HD cf ld_stream(const cf* p) { return *p; }
GLV_HD f4 ld_stream(const f4* p) { return *p; }
#endif
#if defined(__HIP_DEVICE_COMPILE__) && (GLV_STREAM_NT & 2)
__device__ __forceinline__ void st_stream(cf* p, cf v) {
    __builtin_nontemporal_store(glv_v2f{v.x, v.y}, reinterpret_cast<glv_v2f*>(p));
}
__device__ __forceinline__ void st_stream(f4* p, f4 v) {
    __builtin_nontemporal_store(glv_v4f{v.x, v.y, v.z, v.w}, reinterpret_cast<glv_v4f*>(p));
}
#else
GLV_HD void st_stream(cf* p, cf v) { *p = v; }
GLV_HD void st_stream(f4* p, f4 v) { *p = v; }
#endif

// Stores of blocks that waves of OTHER workgroups read later in the same launch (iteration-spanning
// kernels): written through to memory at device scope, 8 bytes per instruction.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void st_shared(cf* p, cf v) {
    const unsigned long long bits =
        ((unsigned long long)__float_as_uint(v.y) << 32) | (unsigned long long)__float_as_uint(v.x);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#else
GLV_HD void st_shared(cf* p, cf v) { *p = v; }
#endif

// 1 / (sqrt(x) + tiny) for x >= 0 without the square root: the hardware reciprocal square root
// (1 ulp), with the argument kept off zero so that a vanishing bin gives 0 * finite = 0 as
// 0 / (0 + tiny) does.  (Bins below 1e-19 in magnitude come out scaled down instead of
// normalised; they carry no signal.)
GLV_HD float inv_norm(float x) {
    const float xs = fmaxf(x, 1.17549435e-38f);
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rsqf(xs);
#else
    return 1.0f / sqrtf(xs);
#endif
}

GLV_HD float rcp_refined(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r = __builtin_amdgcn_rcpf(x);
    return fmaf(fmaf(-x, r, 1.0f), r, r);
#else
    return 1.0f / x;
#endif
}

// Row (H floats) of reciprocal window-sum-squares for block j of a T-frame utterance.
// Fully covered blocks share one row; blocks near an utterance edge, where some of the R
// covering frames do not exist, have rows of their own in host-built tables: `head` for
// j < R-1, `tail` for j > T-1, `small` when the utterance is so short that both ends are
// missing at once.  All arguments are wave-uniform, so this is scalar selection.
GLV_HD const float* iwss_row(const float* iwss, const float* iwss_head, const float* iwss_tail,
                             const float* iwss_small, int H, int R, int T, int j) {
    const bool after_head = j >= R - 1, before_tail = j <= T - 1;
    if (after_head && before_tail) return iwss;
    if (before_tail) return iwss_head + (int64_t)j * H;
    if (after_head) return iwss_tail + (int64_t)((j - T) < R - 1 ? (j - T) : R - 2) * H;
    const int jj = j < 2 * R ? j : 2 * R - 1;
    return iwss_small + ((int64_t)(T < R - 1 ? T : R - 2) * 2 * R + jj) * H;
}

// defined after the engine; used by its on-the-spot initial spectrum (ph3_draw)
GLV_HD int native_s_index(int reg, int lane);
template <int A>
GLV_HD int pair_s_slot(int nzh, int reg, int lane);
GLV_HD cf counter_phasor(uint64_t seed, uint64_t idx);
GLV_HD uint32_t counter_high_part(uint64_t seed, uint32_t idx_high);
GLV_HD float counter_uniform_low(uint32_t high_part, uint32_t idx_low);
GLV_HD cf phasor_of(float u);

// TWR = 2: only the first pass's (the second pass's 8 entries per lane are read from LDS; they
// are the same for lanes 8 apart, so those reads are broadcasts).
// TWR = 1: the twiddles of the first and second FFT pass stay in registers for the whole run as
// well (n_fft 1024 only; 26 more registers, 12 fewer 16-byte LDS reads per frame) - for the
// kernels whose register budget leaves room for them at three waves per SIMD.
// NT: the frame loop never reads a tail buffer (iteration-spanning kernels: the blocks whose
// second half sits there are all taken before the first frame, preload_lin / preload_next).
template <int A, int HB, int TWR = 0, bool NT = false>
struct Engine {
    using G = Geo<A>;
    using F = WaveFft<A>;
    static constexpr int E = G::E, P = G::P, N = G::N;
    static constexpr int ROW = G::ROW;   // samples per row
    static constexpr int H = ROW * HB;   // hop
    static constexpr int R = 8 / HB;     // frames covering one sample
    static constexpr int HV = HB * A;    // registers per hop
    static constexpr int CH = (A == 1) ? GLV_SPLIT_CHUNK_A1 : GLV_SPLIT_CHUNK;  // registers per split chunk
    static constexpr float kTiny = 1.17549435e-38f;
    // The previous spectrum is fetched a frame ahead (E more live registers) or on demand.
    static constexpr bool kPrefetchTp = GLV_PREFETCH_TP_ALL || (A == 1);

    cf raw[E];   // the window slides in registers: only the newest block is loaded per frame
    cf rawp[E];  // window over the previous signal (MODE_ITER_RC)
    cf acc[E];
    cf v[E];
    // loads issued one frame ahead of their use (software prefetch)
    cf tp_n[E];
    float s_n[E];
    float s_nyq_n;
    float s_mid_n;  // pair order: magnitude of the self-mirrored bin P/2 (used by lane 0)
    int mbase;
    int lane;
    RunDesc run;
    float sc_num, sc_den;
    Tables tab;

    // The window and the split twiddles are the same for every frame of a run: with registers
    // to spare (n_fft 1024) a lane keeps its entries instead of reading them from LDS twice per
    // frame - the LDS pipeline is the busiest unit of this kernel.
#ifndef GLV_TABLES_IN_REGS
#define GLV_TABLES_IN_REGS 1
#endif
    static constexpr bool kRegTab = (A == 1) && GLV_TABLES_IN_REGS;
    cf win_r[kRegTab ? E : 1];
    cf twp_r[kRegTab ? E : 1];
#ifndef GLV_TWIDDLES_IN_REGS
#define GLV_TWIDDLES_IN_REGS 0
#endif
#ifndef GLV_TWB_IN_REGS_A2
#define GLV_TWB_IN_REGS_A2 1
#endif
    // the FFT pass twiddles as well (TWR): 26 more registers, which the time-domain-momentum
    // kernels can afford at three waves per SIMD (38.7 against 39.9 us per launch) and the
    // others cannot (GLV_TWIDDLES_IN_REGS forces them on everywhere, for experiments)
    static constexpr bool kRegTw = kRegTab && (TWR != 0 || GLV_TWIDDLES_IN_REGS);
    // n_fft 2048: only the second pass's (one set serves both sub-sequences; 249 of the 256
    // registers two waves per SIMD can have, 148.3 -> 146.5 us per launch)
    static constexpr bool kRegTwB =
        (kRegTab && (TWR == 1 || GLV_TWIDDLES_IN_REGS)) || (A == 2 && TWR == 1 && GLV_TWB_IN_REGS_A2);
    cf twa_r[kRegTw ? 8 : 1];
    cf twb_r[kRegTwB ? 8 : 1];
    GLV_HD const cf* twa_regs() const { return kRegTw ? twa_r : nullptr; }
    GLV_HD const cf* twb_regs() const { return kRegTwB ? twb_r : nullptr; }
    GLV_HD const cf* twp_regs() const { return kRegTab ? twp_r : nullptr; }
    GLV_HD void load_window_lds(cf (&w)[E]) const {
#pragma unroll
        for (int kp = 0; kp < E / 2; ++kp) G::load2(tab.win, kp, lane, w[2 * kp], w[2 * kp + 1]);
    }
    // this lane's window values
    GLV_HD void load_window(cf (&w)[E]) const {
        if (kRegTab) {
#pragma unroll
            for (int q = 0; q < E; ++q) w[q] = win_r[q];
        } else {
            load_window_lds(w);
        }
    }
    // what the loads of the iteration state need (init sets it again, with the rest)
    GLV_HD void init_run(const IterArgs& a, int lane_, const RunDesc& r) {
        lane = lane_;
        run = r;
        block_extra = a.raw_in ? 1 : 0;
        share_next = nullptr;
        deposit_out = nullptr;
        deposit_in = nullptr;
        ring_prev = nullptr;
        heads_shared = false;
        tails_shared = false;
    }
    GLV_HD void init(const IterArgs& a, const cf* tables, int lane_, const RunDesc& r) {
        lane = lane_;
        run = r;
        block_extra = a.raw_in ? 1 : 0;
        share_next = nullptr;
        deposit_out = nullptr;
        deposit_in = nullptr;
        ring_prev = nullptr;
        heads_shared = false;
        tails_shared = false;
        tab.twA = tables;
        tab.twB = tab.twA + G::TW_A;
        tab.twC = tab.twB + G::TW_B;
        tab.twP = tab.twC + G::TW_C;
        tab.win = tab.twP + G::TW_P;
        mbase = G::mirror_base(lane);
        if (kRegTw) {
            cf ta[8];
            G::load8(tab.twA, 0, lane, ta);
#pragma unroll
            for (int k = 0; k < 8; ++k) twa_r[k % (kRegTw ? 8 : 1)] = ta[k];
        }
        if (kRegTwB) {
            cf tb[8];
#pragma unroll
            for (int kp = 0; kp < 4; ++kp) G::load2_8(tab.twB, kp, lane & 7, tb[2 * kp], tb[2 * kp + 1]);
#pragma unroll
            for (int k = 0; k < 8; ++k) twb_r[k % (kRegTwB ? 8 : 1)] = tb[k];
        }
        if (kRegTab) {
            cf w[E];
            load_window_lds(w);
#pragma unroll
            for (int q = 0; q < E; ++q) win_r[q] = w[q];
#pragma unroll
            for (int kp = 0; kp < E / 2; ++kp) {
                cf e0, e1;
                G::load2(tab.twP, kp, lane, e0, e1);
                twp_r[(2 * kp) % (kRegTab ? E : 1)] = e0;
                twp_r[(2 * kp + 1) % (kRegTab ? E : 1)] = e1;
            }
        }
#pragma unroll
        for (int r2 = 0; r2 < E; ++r2) acc[r2] = cf{0.f, 0.f};
        sc_num = 0.f;
        sc_den = 0.f;
    }
    // a further iteration over the same run (iteration-spanning kernels): the running sum restarts
    GLV_HD void begin_iteration() {
#pragma unroll
        for (int r2 = 0; r2 < E; ++r2) acc[r2] = cf{0.f, 0.f};
    }

    // Block j of the previous iteration's signal is (main + tail) / wss, and zero outside
    // the signal proper (the zero padding of a centred STFT).  Reading it is split in two so
    // the loads can be issued a whole frame ahead of their use, and it is branch-free:
    // blocks or parts that do not exist are read from a zero buffer, so the number of memory
    // operations per frame is constant and the compiler can count them (s_waitcnt vmcnt(N))
    // instead of draining the queue.
    struct BlockFetch {
        cf main[HV];
        cf tail[HV];
        cf w[HV];
    };
    // a lane's share of row i of a block: 2A contiguous floats -> one register per sub-sequence
    GLV_HD void load_row(const float* p, int i, cf* dst) const {
        if (A == 1) {
            dst[i] = ld_stream(reinterpret_cast<const cf*>(p + ROW * i + 2 * lane));
        } else {
            const f4 q = ld_stream(reinterpret_cast<const f4*>(p + ROW * i + 4 * lane));
            dst[i] = cf{q.x, q.y};
            dst[i + HB * (A - 1)] = cf{q.z, q.w};
        }
    }
    // the signal proper is blocks [R/2, R/2 + T-1); block_extra = 1 where the block after it holds
    // the up to H-1 samples of a signal that is not a whole number of hops long (forward STFT)
    int block_extra = 0;
    // Blocks t_end .. t_end+R-2, which this run's last frames need, are exactly the blocks the
    // NEXT run of the same utterance loaded, completed and combined before its first frame.  When
    // that run is a wave of the same workgroup it leaves them in LDS (share_next), and this run
    // takes them from there instead of fetching both halves of both signals again.
    const cf* share_next = nullptr;
    // Boundary blocks completed on the producer's side.  The R-1 blocks at a run boundary are the
    // sum of two partial sums: the earlier run's last frames ("tail") and the later run's first
    // frames ("head").  Normally each goes to a buffer of its own and whoever reads the blocks
    // adds the two.  Where both runs are waves of one workgroup, the later run leaves its head
    // sums in LDS as it emits them (deposit_out) and the earlier run, at its end and after a
    // workgroup barrier, adds its tail sums (deposit_in) and writes the COMPLETE blocks to the
    // main buffer: nothing goes to the tail buffer, and the next launch reads one half instead of
    // two.  head + tail in that order, as the reader adds them: the same bits.
    cf* deposit_out = nullptr;       // this run's first R-1 blocks go here, not to memory
    const cf* deposit_in = nullptr;  // the next run's deposit: completes this run's last R-1 blocks
    // Iteration-spanning kernels: the previous / next run reads this run's first R-1 blocks / its
    // tail sums from memory within the same launch, so those stores are written through (st_shared)
    bool heads_shared = false;
    bool tails_shared = false;
    // Iteration-spanning kernels, n_fft 1024: a run that completes its last R-1 blocks itself
    // (deposit_in) has, at the end of an iteration, everything the NEXT iteration's hand-over
    // between it and its successor consists of - the complete blocks of this iteration and of the
    // one before.  It leaves y_k - c y_{k-1} of those blocks in the LDS slot both runs read
    // (share_next; what the successor's preload would have made from memory) and keeps the
    // normalised blocks of this iteration in ring_prev for the next time.
    cf* ring_prev = nullptr;
    GLV_HD bool from_next(int j) const { return share_next != nullptr && j >= run.t_end; }
    GLV_HD bool block_valid(int j) const { return j >= R / 2 && j < R / 2 + run.n_frames - 1 + block_extra; }
    GLV_HD bool block_has_tail(const IterArgs& a, int j) const {
        if (a.raw_in) return false;
        if (j >= run.t_end) return deposit_in == nullptr;                  // else complete in main
        return run.t_begin > 0 && j < run.t_begin + R - 1 && deposit_out == nullptr;
    }
    GLV_HD void fetch_block(const IterArgs& a, int j, BlockFetch& f) const {
        fetch_block_from(a, a.y_in, a.y_in_tail, j, f);
    }
    GLV_HD void fetch_block_from(const IterArgs& a, const float* y_main, const float* y_tail, int j,
                                 BlockFetch& f) const {
        const bool valid = block_valid(j) && !from_next(j);
        const int64_t base = (int64_t)run.utt * a.y_utt_stride + (int64_t)(j - a.t_origin) * H;
        const float* pm = (valid && (a.raw_in || j < run.n_frames)) ? y_main + base : a.zeros;
        const float* pt = (valid && block_has_tail(a, j)) ? y_tail + base : a.zeros;
        const float* pw = iwss_row(a.iwss, a.iwss_head, a.iwss_tail, a.iwss_small, H, R, run.n_frames, j);
#pragma unroll
        for (int i = 0; i < HB; ++i) {
            load_row(pm, i, f.main);
            load_row(pt, i, f.tail);
            load_row(pw, i, f.w);
        }
    }
    // completes a fetched block into rows row0 .. row0+HB-1 of a time-side register array
    GLV_HD void finish_block(const IterArgs& a, const BlockFetch& f, cf (&dst)[E], int row0) const {
        finish_block_w(a, f, f.w, dst, row0);
    }
    // the same with the window-sum-square row taken from another fetch of the same block index
    GLV_HD void finish_block_w(const IterArgs& a, const BlockFetch& f, const cf (&w)[HV], cf (&dst)[E],
                               int row0) const {
#pragma unroll
        for (int s2 = 0; s2 < A; ++s2)
#pragma unroll
            for (int i = 0; i < HB; ++i) {
                const int k = i + HB * s2;
                const cf val = f.main[k] + f.tail[k];
                const float wx = a.raw_in ? 1.0f : w[k].x, wy = a.raw_in ? 1.0f : w[k].y;
                dst[row0 + i + 8 * s2] = cf{val.x * wx, val.y * wy};
            }
    }
    // the same for a block fetched without its tail half (frame loop of the NT engines)
    GLV_HD void finish_block_nt(const IterArgs& a, const BlockFetch& f, const cf (&w)[HV], cf (&dst)[E],
                                int row0) const {
#pragma unroll
        for (int s2 = 0; s2 < A; ++s2)
#pragma unroll
            for (int i = 0; i < HB; ++i) {
                const int k = i + HB * s2;
                dst[row0 + i + 8 * s2] = cf{f.main[k].x * w[k].x, f.main[k].y * w[k].y};
            }
    }
    GLV_HD void load_block(const IterArgs& a, const float* y_main, const float* y_tail, int j,
                           cf (&dst)[E], int row0) const {
        BlockFetch f;
        fetch_block_from(a, y_main, y_tail, j, f);
        finish_block(a, f, dst, row0);
    }
    // drops the oldest hop of a time-side register array
    static GLV_HD void shift_rows(cf (&x)[E]) {
#pragma unroll
        for (int s2 = 0; s2 < A; ++s2)
#pragma unroll
            for (int r = 0; r + HB < 8; ++r) x[r + 8 * s2] = x[r + HB + 8 * s2];
    }
    BlockFetch blk_n;
    BlockFetch blkp_n;  // same, previous signal (MODE_ITER_RC)

    // issue the loads frame t will need (t clamped into the run: the last frame re-requests
    // its own data rather than branching): its newest signal block ...
    GLV_HD void prefetch_block(const IterArgs& a, int t) {
        const int tt = t < run.t_end ? t : run.t_end - 1;
        fetch_block(a, tt + R - 1, blk_n);
    }
    GLV_HD void prefetch_block_prev(const IterArgs& a, int t) {
        const int tt = t < run.t_end ? t : run.t_end - 1;
        fetch_block_from(a, a.y_prev, a.y_prev_tail, tt + R - 1, blkp_n);
    }
    // both signals' newest block of frame t: one set of conditions, one window-sum-square row
    GLV_HD void prefetch_block_pair(const IterArgs& a, int t) {
        const int tt = t < run.t_end ? t : run.t_end - 1;
        const int j = tt + R - 1;
        const bool valid = block_valid(j) && !from_next(j);
        const bool has_main = valid && (a.raw_in || j < run.n_frames);
        const bool has_tail = valid && block_has_tail(a, j);
        const int64_t base = (int64_t)run.utt * a.y_utt_stride + (int64_t)(j - a.t_origin) * H;
        const float* pm = has_main ? a.y_in + base : a.zeros;
        const float* pmp = has_main ? a.y_prev + base : a.zeros;
        const float* pt = has_tail ? a.y_in_tail + base : a.zeros;
        const float* ptp = has_tail ? a.y_prev_tail + base : a.zeros;
        const float* pw = iwss_row(a.iwss, a.iwss_head, a.iwss_tail, a.iwss_small, H, R, run.n_frames, j);
#pragma unroll
        for (int i = 0; i < HB; ++i) {
            load_row(pm, i, blk_n.main);
            if (!NT) load_row(pt, i, blk_n.tail);
            load_row(pw, i, blk_n.w);
            load_row(pmp, i, blkp_n.main);
            if (!NT) load_row(ptp, i, blkp_n.tail);
        }
    }
    // ... and its previous spectrum and target magnitudes
    GLV_HD void prefetch_spec(const IterArgs& a, int t) {
        const int tt = t < run.t_end ? t : run.t_end - 1;
        if (kPrefetchTp) load_spec(a, tt, tp_n);
        prefetch_s(a, t);
    }
    GLV_HD void prefetch_s(const IterArgs& a, int t) {
        const int tt = t < run.t_end ? t : run.t_end - 1;
        const float* sf = a.S + (int64_t)run.utt * a.s_utt_stride + (int64_t)(tt - a.t_origin) * (P + 4);
#pragma unroll
        for (int q = 0; q < E / 4; ++q) {
            const f4 sq = ld_stream(reinterpret_cast<const f4*>(sf) + q * 64 + lane);
            s_n[4 * q] = sq.x;
            s_n[4 * q + 1] = sq.y;
            s_n[4 * q + 2] = sq.z;
            s_n[4 * q + 3] = sq.w;
        }
        s_nyq_n = sf[P];
    }

    // rows row0 .. row0+HB-1 of a time-side register array -> block j
    // (SHARED: another workgroup reads the block within this launch, see st_shared)
    template <bool SHARED = false>
    GLV_HD void store_block(const IterArgs& a, float* dst, int j, const cf (&src)[E], int row0) const {
        float* p = block_valid(j) ? dst + (int64_t)run.utt * a.y_utt_stride + (int64_t)(j - a.t_origin) * H : a.trash;
#pragma unroll
        for (int i = 0; i < HB; ++i) {
            if (A == 1) {
                cf* q = reinterpret_cast<cf*>(p + ROW * i + 2 * lane);
                if (SHARED)
                    st_shared(q, src[row0 + i]);
                else
                    st_stream(q, src[row0 + i]);
            } else {
                const cf lo = src[row0 + i], hi = src[row0 + i + 8 * (A - 1)];
                if (SHARED) {
                    cf* q = reinterpret_cast<cf*>(p + ROW * i + 4 * lane);
                    st_shared(q, lo);
                    st_shared(q + 1, hi);
                } else {
                    st_stream(reinterpret_cast<f4*>(p + ROW * i + 4 * lane), f4{lo.x, lo.y, hi.x, hi.y});
                }
            }
        }
    }

    // blocks t_begin .. t_begin+R-2 of the first frame
    GLV_HD void preload(const IterArgs& a) {
#pragma unroll
        for (int q = 0; q < R - 1; ++q) load_block(a, a.y_in, a.y_in_tail, run.t_begin + q, raw, q * HB);
    }
    GLV_HD void preload_prev(const IterArgs& a) {
#pragma unroll
        for (int q = 0; q < R - 1; ++q) load_block(a, a.y_prev, a.y_prev_tail, run.t_begin + q, rawp, q * HB);
    }
    // ---- phase 0 on the previous signal (MODE_ITER_RC): same frame, older iterate ----
    GLV_HD void ph0_prev(const IterArgs& a, int t, cf* w) {
        finish_block(a, blkp_n, rawp, (R - 1) * HB);
        prefetch_block_prev(a, t + 1);
        cf wtab[E];
        load_window(wtab);
#pragma unroll
        for (int r = 0; r < E; ++r) {
            const cf wn = wtab[r];
            v[r] = cf{rawp[r].x * wn.x, rawp[r].y * wn.y};
        }
        shift_rows(rawp);
        F::fwd_a(v, tab, w, lane, twa_regs());
    }
    // finishes the previous signal's frame: its spectrum is the momentum term's tprev
    GLV_HD void ph3_prev(const cf* r) {
        cf m[E];
        F::mirror_load(m, r, mbase);
        F::split_fwd(v, m, tab, lane, twp_regs());
#pragma unroll
        for (int reg = 0; reg < E; ++reg) tp_n[reg] = v[reg];
    }

    // ---- MODE_ITER_LIN: the sliding window holds y_k - c * y_{k-1}; with the metric wanted,
    // `rawp` holds y_k alone beside it ----
    template <bool SC>
    GLV_HD void preload_lin(const IterArgs& a) {
        const float c = a.has_prev ? a.cmom : 0.0f;
#pragma unroll
        for (int q = 0; q < R - 1; ++q) {
            load_block(a, a.y_in, a.y_in_tail, run.t_begin + q, raw, q * HB);
            load_block(a, a.y_prev, a.y_prev_tail, run.t_begin + q, v, q * HB);
        }
        combine_rows<SC, 0, (R - 1) * HB>(c);
        if (NT && share_next != nullptr) {
            // an utterance shorter than R-1 frames: some of these blocks lie behind the run's last
            // frame, were fetched as zeros and come from the LDS slot like the later ones
#pragma unroll
            for (int q = 0; q < R - 1; ++q)
                if (run.t_begin + q >= run.t_end) {
#pragma unroll
                    for (int i = 0; i < HB; ++i)
                        raw[q * HB + i] = raw[q * HB + i] + share_next[((run.t_begin + q - run.t_end) * HB + i) * (ROW / 2) + lane];
                }
        }
    }
    // rows [row0, row1) of every sub-sequence: raw = y_k - c * y_{k-1} (y_{k-1} sits in v)
    template <bool SC, int ROW0, int ROW1>
    GLV_HD void combine_rows(float c) {
#pragma unroll
        for (int s2 = 0; s2 < A; ++s2)
#pragma unroll
            for (int r = ROW0; r < ROW1; ++r) {
                const int k = r + 8 * s2;
                if (SC) rawp[k] = raw[k];
                raw[k] = cf{fmaf(-c, v[k].x, raw[k].x), fmaf(-c, v[k].y, raw[k].y)};
            }
    }
    // newest block of both signals into the windows, loads for the next frame issued
    template <bool SC>
    GLV_HD void lin_advance(const IterArgs& a, int t) {
        if (NT) {
            finish_block_nt(a, blk_n, blk_n.w, raw, (R - 1) * HB);
            finish_block_nt(a, blkp_n, blk_n.w, v, (R - 1) * HB);
        } else {
            finish_block(a, blk_n, raw, (R - 1) * HB);
            finish_block_w(a, blkp_n, blk_n.w, v, (R - 1) * HB);
        }
        prefetch_block_pair(a, t + 1);
        combine_rows<SC, (R - 1) * HB, R * HB>(a.has_prev ? a.cmom : 0.0f);
        if (from_next(t + R - 1)) take_shared(t + R - 1 - run.t_end);
    }
    // floats a run publishes: its R-1 preloaded blocks
    static constexpr int kShareFloats = (R - 1) * H;
    // rows [0, (R-1) HB) of the window after preload_lin -> LDS (n_fft 1024 form of the layout)
    GLV_HD void publish_preload(cf* mine) const {
        static_assert(A == 1, "only the n_fft 1024 kernels share boundary blocks");
#pragma unroll
        for (int row = 0; row < (R - 1) * HB; ++row) mine[row * (ROW / 2) + lane] = raw[row];
    }
    // The R-1 blocks after this run's last frame, combined like a preload (y_k - c y_{k-1}), left in
    // LDS where take_shared finds them: what the next run's publish_preload would have put there,
    // made by this run itself because that run is not a wave of this workgroup (or does not
    // exist).  Both halves of every block are fetched here, so the frame loop needs no tails.
    // Call before preload_lin (uses the same registers) and before share_next is set.
    GLV_HD void preload_next(const IterArgs& a, cf* slot) {
        static_assert(A == 1, "only the n_fft 1024 kernels share boundary blocks");
        const float c = a.has_prev ? a.cmom : 0.0f;
#pragma unroll
        for (int q = 0; q < R - 1; ++q) {
            load_block(a, a.y_in, a.y_in_tail, run.t_end + q, raw, q * HB);
            load_block(a, a.y_prev, a.y_prev_tail, run.t_end + q, v, q * HB);
        }
        combine_rows<false, 0, (R - 1) * HB>(c);
        publish_preload(slot);
    }
    // the window-sum-square rows of blocks t_end .. t_end+R-2, requested after the last frame into
    // the transform's work registers, idle until the next frame (their way from the cache passes
    // under flush_tails)
    // (hops of a quarter of the window and more: with seven boundary blocks the kernels would spill)
    static constexpr bool kRingShape = A == 1 && (R - 1) * HB <= E && R <= 4;
    GLV_HD void fetch_ring_rows(const IterArgs& a) {
#pragma unroll
        for (int q = 0; q < R - 1; ++q) {
            const float* pw = iwss_row(a.iwss, a.iwss_head, a.iwss_tail, a.iwss_small, H, R, run.n_frames, run.t_end + q);
#pragma unroll
            for (int i = 0; i < HB; ++i) load_row(pw, i, v + q * HB);
        }
    }
    // after flush_tails of a run with deposit_in (acc holds the complete sums of blocks t_end ..):
    // normalise them, combine with the blocks of the iteration before (`first`: those are what
    // the slot still holds from the successor's one preload from memory), leave the result in the
    // slot and the normalised blocks in ring_prev.  The same operations in the same order as
    // load_block + combine_rows of a preload from memory.
    GLV_HD void ring_hand_over(const IterArgs& a, cf* slot, bool first) {
        static_assert(A == 1, "only the n_fft 1024 kernels share boundary blocks");
#pragma unroll
        for (int q = 0; q < R - 1; ++q)
#pragma unroll
            for (int i = 0; i < HB; ++i) {
                const int row = q * HB + i, at = row * (ROW / 2) + lane;
                const cf cn = cf{acc[row].x * v[row].x, acc[row].y * v[row].y};
                const cf pn = first ? slot[at] : ring_prev[at];
                ring_prev[at] = cn;
                slot[at] = cf{fmaf(-a.cmom, pn.x, cn.x), fmaf(-a.cmom, pn.y, cn.y)};
            }
    }
    // a run whose predecessor handed over through the slot: the window's first R-1 blocks
    GLV_HD void take_ring_preload(const cf* mine) {
#pragma unroll
        for (int row = 0; row < (R - 1) * HB; ++row) raw[row] = mine[row * (ROW / 2) + lane];
    }
    // block q of the next run's preload = this frame's newest block (the fetch delivered zeros)
    GLV_HD void take_shared(int q) {
#pragma unroll
        for (int i = 0; i < HB; ++i) {
            const cf x = share_next[(q * HB + i) * (ROW / 2) + lane];
            raw[(R - 1) * HB + i] = raw[(R - 1) * HB + i] + x;
        }
    }
    // window one of the two arrays and start its forward transform
    GLV_HD void lin_window(const cf (&src)[E], cf* w) {
        cf wtab[E];
        load_window(wtab);
#pragma unroll
        for (int r = 0; r < E; ++r) {
            const cf wn = wtab[r];
            v[r] = cf{src[r].x * wn.x, src[r].y * wn.y};
        }
        F::fwd_a(v, tab, w, lane, twa_regs());
    }
    GLV_HD void lin_window_signal(cf* w) { lin_window(rawp, w); }
    GLV_HD void lin_window_combined(cf* w) { lin_window(raw, w); }
    template <bool SC>
    GLV_HD void lin_shift() {
        shift_rows(raw);
        if (SC) shift_rows(rawp);
    }
    // the spectrum of y_k alone exists only for the convergence metric
    GLV_HD void ph3_metric(const cf* r) {
        cf m[E];
        F::mirror_load(m, r, mbase);
        F::split_fwd(v, m, tab, lane, twp_regs());
#pragma unroll
        for (int reg = 0; reg < E; ++reg) {
            const cf x = v[reg];
            const bool packed = (reg == 0) && (lane == 0);
            const float sx = s_n[reg], sy = packed ? s_nyq_n : s_n[reg];
            const float mag = sqrtf(x.x * x.x + x.y * x.y);
            const float ex = (packed ? fabsf(x.x) : mag) - fabsf(sx);
            const float ey = packed ? fabsf(x.y) - fabsf(sy) : 0.f;
            sc_num += ex * ex + ey * ey;
            sc_den += sx * sx + (packed ? sy * sy : 0.f);
        }
    }

    // ---- phase 0: newest block, window, first DFT pass ----
    GLV_HD void ph0(const IterArgs& a, int t, cf* w) {
        finish_block(a, blk_n, raw, (R - 1) * HB);
        prefetch_block(a, t + 1);
        cf wtab[E];
        load_window(wtab);
#pragma unroll
        for (int r = 0; r < E; ++r) {
            const cf wn = wtab[r];
            v[r] = cf{raw[r].x * wn.x, raw[r].y * wn.y};
        }
        shift_rows(raw);
        F::fwd_a(v, tab, w, lane, twa_regs());
    }
    GLV_HD void ph1_read(const cf* r) { F::fwd_b(v, tab, r, lane, twb_regs()); }
    GLV_HD void ph1_write(cf* w) { F::fwd_b_store(v, w, lane); }
    GLV_HD void ph2_read(const cf* r) { F::fwd_c(v, tab, r, lane); }
    GLV_HD void ph2_write(cf* w) { F::mirror_store(v, w, lane); }

    GLV_HD cf* spec_frame(const IterArgs& a, int t) const {
        return a.spec + (int64_t)run.utt * a.spec_utt_stride + (int64_t)(t - a.t_origin) * P;
    }

    // ---- phase 3 (STFT only): finish the split and store the frame ----
    GLV_HD void ph3_stft(const IterArgs& a, int t, const cf* r) {
        cf m[E];
        F::mirror_load(m, r, mbase);
        F::split_fwd(v, m, tab, lane, twp_regs());
        store_spec(a, t);
    }
    // registers q0 .. q0+nq-1 (q0, nq even) of a native frame
    GLV_HD void store_spec(const IterArgs& a, int t, int q0 = 0, int nq = E) {
        f4* sp = reinterpret_cast<f4*>(spec_frame(a, t));
#pragma unroll
        for (int pq = q0 / 2; pq < (q0 + nq) / 2; ++pq)
            st_stream(sp + pq * 64 + lane, f4{v[2 * pq].x, v[2 * pq].y, v[2 * pq + 1].x, v[2 * pq + 1].y});
    }
    GLV_HD void load_spec(const IterArgs& a, int t, cf (&dst)[E], int q0 = 0, int nq = E) const {
        const f4* sp = reinterpret_cast<const f4*>(spec_frame(a, t));
#pragma unroll
        for (int pq = q0 / 2; pq < (q0 + nq) / 2; ++pq) {
            const f4 q = ld_stream(sp + pq * 64 + lane);
            dst[2 * pq] = cf{q.x, q.y};
            dst[2 * pq + 1] = cf{q.z, q.w};
        }
    }

    // ---- phase 3 (iteration): rebuilt spectrum, momentum phase update ----
    // librosa.griffinlim loop body after the stft: angles = rebuilt - c*tprev;
    // angles /= |angles| + tiny; angles *= S; tprev = rebuilt.
    // RC = 0: tprev lives in `spec` (read, then overwritten with rebuilt); 1: tprev was just
    // recomputed into tp_n and nothing is stored; 2: first iteration of that scheme (no tprev).
    template <bool SC, int RC = 0, bool PREFETCH_NEXT = true>
    GLV_HD void ph3_update(const IterArgs& a, int t, const cf* r) {
        // first iteration: no momentum term (tp_n then holds the initial angles, times zero)
        const float cmom = (RC == 2 || (RC == 0 && !a.has_prev)) ? 0.0f : a.cmom;
        if (RC == 2) {
#pragma unroll
            for (int reg = 0; reg < E; ++reg) tp_n[reg] = cf{0.f, 0.f};
        }
#pragma unroll
        for (int q0 = 0; q0 < E; q0 += CH) {
            cf m[CH];
            F::template mirror_load_part<CH>(m, r, mbase, q0);
            if (RC == 0 && !kPrefetchTp) load_spec(a, t, tp_n, q0, CH);
            F::template split_fwd_part<CH>(v, m, tab, lane, q0, twp_regs());  // v = rebuilt frame
            if (RC == 0) store_spec(a, t, q0, CH);
#pragma unroll
            for (int reg = q0; reg < q0 + CH; ++reg) {
                const cf x = v[reg];
                const bool packed = (reg == 0) && (lane == 0);
                const float sx = s_n[reg], sy = packed ? s_nyq_n : s_n[reg];
                if (SC) {
                    const float mag = sqrtf(x.x * x.x + x.y * x.y);
                    const float ex = (packed ? fabsf(x.x) : mag) - fabsf(sx);
                    const float ey = packed ? fabsf(x.y) - fabsf(sy) : 0.f;
                    sc_num += ex * ex + ey * ey;
                    sc_den += sx * sx + (packed ? sy * sy : 0.f);
                }
                const cf d = cf{x.x - cmom * tp_n[reg].x, x.y - cmom * tp_n[reg].y};
                // angles / (|angles| + tiny): one reciprocal square root per bin
                float ix = inv_norm(d.x * d.x + d.y * d.y) * a.inv_n, iy = ix;
                if (reg == 0) {
                    // lane 0 holds two real bins here
                    const float px = rcp_refined(fabsf(d.x) + kTiny) * a.inv_n;
                    const float py = rcp_refined(fabsf(d.y) + kTiny) * a.inv_n;
                    ix = packed ? px : ix;
                    iy = packed ? py : iy;
                }
                v[reg] = cf{d.x * ix * sx, d.y * iy * sy};
            }
            if (CH < E) GLV_SCHED_FENCE();
        }
        if (!PREFETCH_NEXT) return;
        if (RC == 0)
            prefetch_spec(a, t + 1);
        else
            prefetch_s(a, t + 1);
    }
    // ---- MODE_ITER_LIN / MODE_ITER_LIN_Z: split, phase update and inverse split on mirror pairs ----
    static constexpr int NP = E / 2;
    // magnitudes in pair order: s_n[i] = pair i's own bin, s_n[NP + i] = its mirror; NZH = how many
    // of the mirrors (the last NZH pairs) can be non-zero
    template <int NZH>
    GLV_HD void prefetch_s_pair(const IterArgs& a, int t) {
        const int tt = t < run.t_end ? t : run.t_end - 1;
        const float* sf = a.S + (int64_t)run.utt * a.s_utt_stride + (int64_t)(tt - a.t_origin) * (P + 4);
#pragma unroll
        for (int g = 0; g < NP / 4; ++g) {
            const f4 sq = ld_stream(reinterpret_cast<const f4*>(sf) + g * 64 + lane);
            s_n[4 * g] = sq.x;
            s_n[4 * g + 1] = sq.y;
            s_n[4 * g + 2] = sq.z;
            s_n[4 * g + 3] = sq.w;
        }
        const float* hi = sf + NP * 64;
        constexpr int first = NP + (NP - NZH);
        if (NZH % 4 == 0) {
#pragma unroll
            for (int g = 0; g < NZH / 4; ++g) {
                const f4 sq = ld_stream(reinterpret_cast<const f4*>(hi) + g * 64 + lane);
                s_n[first + 4 * g] = sq.x;
                s_n[first + 4 * g + 1] = sq.y;
                s_n[first + 4 * g + 2] = sq.z;
                s_n[first + 4 * g + 3] = sq.w;
            }
        } else {
            static_assert(NZH % 4 == 0 || NZH == 2, "the mirrors are stored four or two to a lane");
            const cf sq = ld_stream(reinterpret_cast<const cf*>(hi) + lane);
            s_n[first] = sq.x;
            s_n[first + 1] = sq.y;
        }
#pragma unroll
        for (int i = NP; i < first; ++i) s_n[i] = 0.f;
        s_nyq_n = sf[P];
        s_mid_n = sf[P + 1];
    }
    // angles / (|angles| + tiny) * S, pre-scaled by 1 / n_fft, of one complex bin
    GLV_HD cf unit_times(cf d, float sv, float inv_n) const {
        const float k = inv_norm(d.x * d.x + d.y * d.y) * inv_n * sv;
        return cf{d.x * k, d.y * k};
    }
    GLV_HD void sc_add(cf x, float sv) {
        const float e = sqrtf(x.x * x.x + x.y * x.y) - fabsf(sv);
        sc_num += e * e;
        sc_den += sv * sv;
    }
    // lane 0's two real bins, DC and Nyquist, packed as (re, im)
    GLV_HD void sc_add_packed(cf x, float sx, float sy) {
        const float ex = fabsf(x.x) - fabsf(sx), ey = fabsf(x.y) - fabsf(sy);
        sc_num += ex * ex + ey * ey;
        sc_den += sx * sx + sy * sy;
    }
    // the spectrum of the signal itself, for the convergence metric only (all bins count, also
    // those whose target is zero)
    GLV_HD void ph3_metric_pair(const cf* r) {
        const cf* twp = twp_regs();
#pragma unroll
        for (int i0 = 0; i0 < NP; i0 += CH / 2) {
            cf m[CH / 2], tw[CH / 2];
            F::template pair_partners<CH / 2>(m, r, mbase, i0);
            load_twp_pairs(tw, twp, i0);
#pragma unroll
            for (int j = 0; j < CH / 2; ++j) {
                const int i = i0 + j;
                const cf a = v[i];
                cf xk, xm;
                F::pair_fwd(a, m[j], tw[j], xk, xm);
                if (i == 0) {
                    if (lane == 0) {
                        sc_add_packed(cf{a.x + a.y, a.x - a.y}, s_n[0], s_nyq_n);
                        // the self-mirrored bin: X[P/2] = conj(Z[P/2])
                        sc_add(v[NP], s_mid_n);
                    } else {
                        sc_add(xk, s_n[0]);
                        sc_add(xm, s_n[NP]);
                    }
                } else {
                    sc_add(xk, s_n[i]);
                    sc_add(xm, s_n[NP + i]);
                }
            }
            if (CH < E) GLV_SCHED_FENCE();
        }
    }
    // this lane's twiddles 0.5 W_N^k of pairs i0 .. i0+NQ-1
    template <int NQ>
    GLV_HD void load_twp_pairs(cf (&tw)[NQ], const cf* twp, int i0) const {
        static_assert(NQ % 2 == 0, "table entries are fetched in pairs");
        if (twp) {
#pragma unroll
            for (int j = 0; j < NQ; ++j) tw[j] = twp[i0 + j];
        } else {
#pragma unroll
            for (int kp = 0; kp < NQ / 2; ++kp) G::load2(tab.twP, i0 / 2 + kp, lane, tw[2 * kp], tw[2 * kp + 1]);
        }
    }
    // v = forward transform of y_k - c y_{k-1}, partners published in r.  Leaves the low
    // registers of v holding the inverse split's result and sends the partners theirs through w
    // (the same buffer on the device: a wave's LDS reads complete before its later writes).
    template <int NZH>
    GLV_HD void ph3_pair(const IterArgs& a, int t, const cf* r, cf* w) {
        const cf* twp = twp_regs();
        // lane 0's self-mirrored bin P/2: X = conj(Z), Z' = 2 conj(X')
        const cf zmid = v[NP];
        const cf xmid = unit_times(cf{zmid.x, -zmid.y}, s_mid_n, a.inv_n);
        const cf zmid_new = cf{2.0f * xmid.x, -2.0f * xmid.y};
#pragma unroll
        for (int i0 = 0; i0 < NP; i0 += CH / 2) {
            cf m[CH / 2], tw[CH / 2];
            F::template pair_partners<CH / 2>(m, r, mbase, i0);
            load_twp_pairs(tw, twp, i0);
#pragma unroll
            for (int j = 0; j < CH / 2; ++j) {
                const int i = i0 + j;
                const bool has_hi = i >= NP - NZH;  // the mirror's magnitude can be non-zero
                const cf za = v[i];
                cf xk, xm = cf{0.f, 0.f};
                if (has_hi)
                    F::pair_fwd(za, m[j], tw[j], xk, xm);
                else
                    xk = F::pair_fwd_low(za, m[j], tw[j]);
                cf yk = unit_times(xk, s_n[i], a.inv_n), ym = cf{0.f, 0.f};
                if (has_hi) ym = unit_times(xm, s_n[NP + i], a.inv_n);
                if (i == 0) {
                    // lane 0 holds the two real bins DC and Nyquist here
                    const cf x0 = cf{za.x + za.y, za.x - za.y};
                    const float px = rcp_refined(fabsf(x0.x) + kTiny) * a.inv_n;
                    const float py = rcp_refined(fabsf(x0.y) + kTiny) * a.inv_n;
                    const cf y0 = cf{x0.x * px * s_n[0], x0.y * py * s_nyq_n};
                    const bool packed = lane == 0;
                    yk = packed ? y0 : yk;
                    if (has_hi) ym = packed ? cf{0.f, 0.f} : ym;
                }
                cf zk, zm;
                if (has_hi)
                    F::pair_inv(yk, ym, tw[j], zk, zm);
                else
                    F::pair_inv_low_only(yk, tw[j], zk, zm);
                if (i == 0) {
                    const bool packed = lane == 0;
                    zk = packed ? cf{yk.x + yk.y, yk.x - yk.y} : zk;
                }
                v[i] = zk;
                F::pair_send(zm, w, mbase, i);
            }
            if (CH < E) GLV_SCHED_FENCE();
        }
        // nobody else writes row 0 of lane 0's column (its own pairs send to rows 1 ..), so the
        // self-mirrored bin travels the same way as every other high register
        if (lane == 0) w[0] = zmid_new;
        prefetch_s_pair<NZH>(a, t + 1);
    }
    GLV_HD void pair_store(cf* w) { F::pair_store(v, w, lane); }
    // own high registers back from the partners, then the first inverse pass
    GLV_HD void ph4_pair(const cf* r, cf* w) {
        F::pair_receive(v, r, lane);
        F::inv_c(v, tab, w, lane);
    }

    // ---- MODE_TRAFFIC: one frame's memory operations, nothing else ----
    GLV_HD void traffic_frame(const IterArgs& a, int t) {
        cf nb[E];
        finish_block(a, blk_n, nb, 0);
        prefetch_block(a, t + 1);
        if (!kPrefetchTp) load_spec(a, t, tp_n);
#pragma unroll
        for (int r = 0; r < E; ++r) {
            const cf n = nb[(r % HB) + 8 * (r / 8)];
            v[r] = cf{n.x + tp_n[r].x * s_n[r], n.y + tp_n[r].y * s_nyq_n};
        }
        store_spec(a, t);
        prefetch_spec(a, t + 1);
#pragma unroll
        for (int q = 0; q < E; ++q) acc[q] = acc[q] + v[q];
        store_block(a, a.y_out, t, acc, 0);
        shift_rows(acc);
    }

    // ---- MODE_TRAFFIC_LIN(_Z): a frame of the time-domain-momentum iteration without its
    // arithmetic - magnitudes and the newest block of both signals in, one block out, boundary
    // blocks shared / deposited as the real kernel does; every loaded value is used once ----
    template <int NZH>
    GLV_HD void traffic_frame_lin(const IterArgs& a, int t, const cf* xb) {
        lin_advance<false>(a, t);
#pragma unroll
        for (int r = 0; r < E; ++r) v[r] = cf{raw[r].x + s_n[r], raw[r].y + s_nyq_n + s_mid_n};
        lin_shift<false>();
        prefetch_s_pair<NZH>(a, t + 1);
        ph6<false>(a, t, xb);
    }

    // ---- phase 3 (iSTFT only): the frame to invert is already in `spec` ----
    GLV_HD void ph3_load(const IterArgs& a, int t) {
        if (a.draw)
            ph3_draw(a, t);
        else
            load_spec(a, t, v);
    }
    // The initial spectrum of frame t made on the spot: magnitude (wherever the layout of S keeps
    // it; bins without a slot are zero) times the phase drawn for (utterance, frame, bin), scaled
    // by 1/n_fft like the stored form - the same products in the same order, so the same bits.
    GLV_HD void ph3_draw(const IterArgs& a, int t) {
        const uint64_t id = a.draw_ids ? (uint64_t)a.draw_ids[run.utt] : (uint64_t)run.utt;
        const uint64_t key = a.draw_seed ^ (id * 0x9E3779B97F4A7C15ull);
        const uint64_t first = (uint64_t)((t + a.draw_origin) * (P + 1));
        const float* sf = a.S + (int64_t)run.utt * a.s_utt_stride + (int64_t)(t - a.t_origin) * (P + 4);
        // the generator's seed-and-high-word part is the same for the whole frame unless the frame
        // straddles a multiple of 2^32 elements: computed once then (the same values either way)
        const bool one_high = (uint32_t)(first >> 32) == (uint32_t)((first + (uint64_t)P) >> 32);
        const uint32_t high_part = counter_high_part(key, (uint32_t)(first >> 32));
        auto phasor_at = [&](int k) -> cf {
            if (one_high) return phasor_of(counter_uniform_low(high_part, (uint32_t)first + (uint32_t)k));
            return counter_phasor(key, first + (uint64_t)k);
        };
#pragma unroll
        for (int reg = 0; reg < E; ++reg) {
            const int slot = a.draw_nzh ? pair_s_slot<A>(a.draw_nzh, reg, lane) : native_s_index(reg, lane);
            const float s = slot >= 0 ? sf[slot] : 0.0f;
            const cf ph = phasor_at(G::bin(reg, lane));
            v[reg] = cf{ph.x * s * a.inv_n, ph.y * s * a.inv_n};
        }
        if (lane == 0) {
            // DC and Nyquist are real and share native bin 0 (the inverse ignores their Im)
            const cf pn = phasor_at(P);
            v[0].y = pn.x * sf[P] * a.inv_n;
        }
    }

    GLV_HD void ph3_write(cf* w) { F::mirror_store(v, w, lane); }

    GLV_HD void ph4(const cf* r, cf* w) {
#pragma unroll
        for (int q0 = 0; q0 < E; q0 += CH) {
            cf m[CH];
            F::template mirror_load_part<CH>(m, r, mbase, q0);
            F::template split_inv_part<CH>(v, m, tab, lane, q0, twp_regs());
            if (CH < E) GLV_SCHED_FENCE();
        }
        F::inv_c(v, tab, w, lane);
    }
    GLV_HD void ph5_read(const cf* r) { F::inv_b(v, tab, r, lane, twb_regs()); }
    GLV_HD void ph5_write(cf* w) { F::inv_b_store(v, w, lane); }

    // ---- phase 6: last inverse pass, window, overlap-add, emit block t ----
    template <bool TRANSFORM = true>
    GLV_HD void ph6(const IterArgs& a, int t, const cf* r) {
        if (TRANSFORM) F::inv_a(v, tab, r, lane, twa_regs());
        cf wtab[E];
        load_window(wtab);
#pragma unroll
        for (int q = 0; q < E; ++q) {
            const cf wn = wtab[q];
            acc[q].x = fmaf(v[q].x, wn.x, acc[q].x);
            acc[q].y = fmaf(v[q].y, wn.y, acc[q].y);
        }
        if (deposit_out != nullptr && t < run.t_begin + R - 1)
            deposit_block(t - run.t_begin);
        else if (heads_shared && t < run.t_begin + R - 1)
            store_block<true>(a, a.y_out, t, acc, 0);
        else
            store_block(a, a.y_out, t, acc, 0);
        shift_rows(acc);
#pragma unroll
        for (int s2 = 0; s2 < A; ++s2)
#pragma unroll
            for (int r = 8 - HB; r < 8; ++r) acc[r + 8 * s2] = cf{0.f, 0.f};
    }
    // block q of this run's head sums (the oldest hop of the running sum) -> LDS, in the layout of
    // publish_preload
    GLV_HD void deposit_block(int q) {
        if constexpr (A == 1) {
#pragma unroll
            for (int i = 0; i < HB; ++i) deposit_out[(q * HB + i) * (ROW / 2) + lane] = acc[i];
        }
    }
    // ---- one frame per wave (MODE_ITER1 / MODE_ISTFT1) ----
    // Frame tc writes its windowed output, R whole blocks tc .. tc+R-1, to partial-sum buffer
    // (tc mod R): no two frames write the same place.  Block j of the signal is then the sum
    // over the frames j-R+1 .. j that exist, each found in its own buffer, added in ascending
    // frame order like the running sum of the run-based modes (and like librosa's overlap-add).
    GLV_HD void load_frame_from_parts(const IterArgs& a, int t) {
        const int64_t utt_off = (int64_t)run.utt * a.y_utt_stride;
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int j = t + q;
            const bool valid = block_valid(j);
            const float* pw = iwss_row(a.iwss, a.iwss_head, a.iwss_tail, a.iwss_small, H, R, run.n_frames, j);
            cf sum[HV];
#pragma unroll
            for (int k = 0; k < HV; ++k) sum[k] = cf{0.f, 0.f};
#pragma unroll
            for (int d = R - 1; d >= 0; --d) {
                const int tc = j - d;
                const bool ok = valid && tc >= 0 && tc < run.n_frames;
                const float* p = ok ? a.part_in + (int64_t)(tc & (R - 1)) * a.part_stride + utt_off + (int64_t)j * H
                                    : a.zeros;
                cf part[HV];
#pragma unroll
                for (int i = 0; i < HB; ++i) load_row(p, i, part);
#pragma unroll
                for (int k = 0; k < HV; ++k) sum[k] = sum[k] + part[k];
            }
            cf w[HV];
#pragma unroll
            for (int i = 0; i < HB; ++i) load_row(pw, i, w);
#pragma unroll
            for (int s2 = 0; s2 < A; ++s2)
#pragma unroll
                for (int i = 0; i < HB; ++i) {
                    const int k = i + HB * s2;
                    raw[q * HB + i + 8 * s2] = cf{sum[k].x * w[k].x, sum[k].y * w[k].y};
                }
            // one block's R + 1 loads at a time: with all R * R in flight the register
            // allocator spills and every load is waited for singly
            // Keep the loads of GLV_PARTS_DEPTH + 1 blocks in flight, no more: with all R * R
            // outstanding the register allocator spills and every load ends up waited for singly.
            if (A == 2 && q >= GLV_PARTS_DEPTH) {
                pin_block(q - GLV_PARTS_DEPTH);
                GLV_SCHED_FENCE();
            }
        }
    }
    // One block of the signal, summed from the partial-sum buffers and normalised exactly as
    // load_frame_from_parts does it (the value does not depend on which frame asks for it).
    GLV_HD void sum_block_from_parts(const IterArgs& a, int j, cf (&out)[HV]) const {
        const int64_t utt_off = (int64_t)run.utt * a.y_utt_stride;
        const bool valid = block_valid(j);
        const float* pw = iwss_row(a.iwss, a.iwss_head, a.iwss_tail, a.iwss_small, H, R, run.n_frames, j);
        cf sum[HV];
#pragma unroll
        for (int k = 0; k < HV; ++k) sum[k] = cf{0.f, 0.f};
#pragma unroll
        for (int d = R - 1; d >= 0; --d) {
            const int tc = j - d;
            const bool ok = valid && tc >= 0 && tc < run.n_frames;
            const float* p = ok ? a.part_in + (int64_t)(tc & (R - 1)) * a.part_stride + utt_off + (int64_t)j * H : a.zeros;
            cf part[HV];
#pragma unroll
            for (int i = 0; i < HB; ++i) load_row(p, i, part);
#pragma unroll
            for (int k = 0; k < HV; ++k) sum[k] = sum[k] + part[k];
        }
        cf w[HV];
#pragma unroll
        for (int i = 0; i < HB; ++i) load_row(pw, i, w);
#pragma unroll
        for (int k = 0; k < HV; ++k) out[k] = cf{sum[k].x * w[k].x, sum[k].y * w[k].y};
    }
    // The W consecutive frames t0 .. t0+W-1 of one utterance held by the W waves of a workgroup
    // need the R+W-1 blocks t0 .. t0+R+W-2 between them: wave `wave` sums every W-th of those
    // (sum_block_from_parts) into `area` ([block][HV][64] in LDS).  After a workgroup barrier
    // take_frame_from_area gives every wave its R blocks: R+W-1 blocks summed per workgroup
    // instead of W x R, the same values.
    template <int W>
    GLV_HD void sum_blocks_for_workgroup(const IterArgs& a, int t0, int wave, cf* area) const {
        constexpr int NB = R + W - 1;
        cf out[(NB + W - 1) / W][HV];
#pragma unroll
        for (int m = 0; m < (NB + W - 1) / W; ++m) {
            const int b = wave + W * m;
            // (a wave without an m-th block sums the last block once more, and does not store it)
            sum_block_from_parts(a, t0 + (b < NB ? b : NB - 1), out[m]);
        }
#pragma unroll
        for (int m = 0; m < (NB + W - 1) / W; ++m) {
            const int b = wave + W * m;
            if (b < NB) {
#pragma unroll
                for (int k = 0; k < HV; ++k) area[(b * HV + k) * 64 + lane] = out[m][k];
            }
        }
    }
    GLV_HD void take_frame_from_area(int t, int t0, const cf* area) {
#pragma unroll
        for (int q = 0; q < R; ++q)
#pragma unroll
            for (int s2 = 0; s2 < A; ++s2)
#pragma unroll
                for (int i = 0; i < HB; ++i)
                    raw[q * HB + i + 8 * s2] = area[((t - t0 + q) * HV + i + HB * s2) * 64 + lane];
    }
    // the sums of block q have to be in their registers here (and no memory operation moves across)
    GLV_HD void pin_block(int q) {
#pragma unroll
        for (int s2 = 0; s2 < A; ++s2)
#pragma unroll
            for (int i = 0; i < HB; ++i) {
                GLV_PIN(raw[q * HB + i + 8 * s2].x);
                GLV_PIN(raw[q * HB + i + 8 * s2].y);
            }
    }
    // phase 0 of a single-frame wave: the whole window is fetched, nothing slides
    // (FROM_AREA: the frame's blocks are in `raw` already, see sum_blocks_for_workgroup)
    template <bool WITH_SPEC, bool FROM_AREA = false>
    GLV_HD void ph0_single(const IterArgs& a, int t, cf* w) {
        if (!FROM_AREA) load_frame_from_parts(a, t);
        // the previous spectrum and the magnitudes are not needed before phase 3: their latency
        // hides behind the forward transform
        if (WITH_SPEC) prefetch_spec(a, t);
        cf wtab[E];
        load_window(wtab);
#pragma unroll
        for (int r = 0; r < E; ++r) {
            const cf wn = wtab[r];
            v[r] = cf{raw[r].x * wn.x, raw[r].y * wn.y};
        }
        F::fwd_a(v, tab, w, lane, twa_regs());
    }
    // phase 6 of a single-frame wave: last inverse pass, window, the frame's R blocks go out
    // (SHARED: the neighbouring frames' waves read these blocks within the same launch, st_shared)
    template <bool SHARED = false>
    GLV_HD void ph6_single(const IterArgs& a, int t, const cf* r) {
        F::inv_a(v, tab, r, lane, twa_regs());
        cf wtab[E];
        load_window(wtab);
#pragma unroll
        for (int q = 0; q < E; ++q) {
            const cf wn = wtab[q];
            v[q] = cf{v[q].x * wn.x, v[q].y * wn.y};
        }
        float* base = a.part_out + (int64_t)(t & (R - 1)) * a.part_stride + (int64_t)run.utt * a.y_utt_stride;
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int j = t + q;
            float* p = block_valid(j) ? base + (int64_t)j * H : a.trash;
#pragma unroll
            for (int i = 0; i < HB; ++i) {
                const int row = q * HB + i;
                if (A == 1) {
                    cf* dst = reinterpret_cast<cf*>(p + ROW * i + 2 * lane);
                    if (SHARED)
                        st_shared(dst, v[row]);
                    else
                        st_stream(dst, v[row]);
                } else {
                    const cf lo = v[row], hi = v[row + 8 * (A - 1)];
                    if (SHARED) {
                        cf* dst = reinterpret_cast<cf*>(p + ROW * i + 4 * lane);
                        st_shared(dst, lo);
                        st_shared(dst + 1, hi);
                    } else {
                        st_stream(reinterpret_cast<f4*>(p + ROW * i + 4 * lane), f4{lo.x, lo.y, hi.x, hi.y});
                    }
                }
            }
        }
    }

    // what is left after the last frame belongs to blocks t_end .. t_end+R-2
    // (store_complete = false: the complete blocks stay in `acc` for ring_hand_over and are not
    // written - inside an iteration-spanning launch nobody reads them from memory before the last
    // iteration has written them)
    GLV_HD void flush_tails(const IterArgs& a, bool store_complete = true) {
        if constexpr (A == 1) {
            if (deposit_in != nullptr) {
                // the next run's head sums + this run's tail sums = the complete blocks
#pragma unroll
                for (int row = 0; row < (R - 1) * HB; ++row) acc[row] = deposit_in[row * (ROW / 2) + lane] + acc[row];
                if (store_complete) {
#pragma unroll
                    for (int q = 0; q < R - 1; ++q) store_block(a, a.y_out, run.t_end + q, acc, q * HB);
                }
                return;
            }
        }
        if (tails_shared) {
#pragma unroll
            for (int q = 0; q < R - 1; ++q) store_block<true>(a, a.y_out_tail, run.t_end + q, acc, q * HB);
            return;
        }
#pragma unroll
        for (int q = 0; q < R - 1; ++q) store_block(a, a.y_out_tail, run.t_end + q, acc, q * HB);
    }
};

// ---------------------------------------------------------------------------
// Element-wise helpers shared by the device kernels and the host emulation.
// ---------------------------------------------------------------------------

// (position in native order) -> natural bin, for position q = reg*64 + lane
template <int A>
GLV_HD int native_bin(int q) {
    return Geo<A>::bin(q >> 6, q & 63);
}

// index of (reg, lane) inside a native spectrum frame (cf units) and S frame (floats)
GLV_HD int native_spec_index(int reg, int lane) { return ((reg >> 1) * 64 + lane) * 2 + (reg & 1); }
GLV_HD int native_s_index(int reg, int lane) { return ((reg >> 2) * 64 + lane) * 4 + (reg & 3); }

// PAIR ORDER of the target magnitudes (MODE_ITER_LIN*): where the magnitude of the bin at native
// position (reg, lane) sits inside a frame of P + 4 floats, or -1 if it has no slot because it is
// zero by construction.  A lane's own low registers come first, [reg / 4][lane][4] as in native
// order; then, for the pairs NP - nzh .. NP - 1 of every lane, the magnitude of the pair's mirror
// bin, [..][lane][4] (or [lane][2] when nzh is 2); the Nyquist bin at [P], the self-mirrored bin
// P/2 at [P + 1].
template <int A>
GLV_HD int pair_s_slot(int nzh, int reg, int lane) {
    constexpr int E = Geo<A>::E, NP = E / 2, P = Geo<A>::P;
    if (reg < NP) return native_s_index(reg, lane);
    const int low = (lane >> 3) + 8 * (lane & 7);
    int owner, i;
    if (low == 0) {
        if (reg == NP) return P + 1;
        owner = 0;
        i = E - reg;
    } else {
        const int lp = 64 - low;
        owner = 8 * (lp & 7) + (lp >> 3);
        i = E - 1 - reg;
    }
    const int h = i - (NP - nzh);
    if (h < 0) return -1;
    const int base = NP * 64;
    return (nzh % 4 == 0) ? base + ((h >> 2) * 64 + owner) * 4 + (h & 3) : base + owner * 2 + h;
}

// Initial phase drawn on the device: the stand-in for librosa's unseeded default_rng when the
// caller supplies no phase.  A counter-based generator (two rounds of a 32-bit
// multiply-xorshift mixer over seed and element index; not numpy's stream) gives u in [0,1),
// the phasor is (cos, sin)(2 pi u).  On the device the hardware sine / cosine take their
// argument in revolutions, so u goes in as it is.
GLV_HD uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
GLV_HD float counter_uniform(uint64_t seed, uint64_t idx) {
    const uint32_t hi = mix32((uint32_t)(idx >> 32) ^ (uint32_t)(seed >> 32) ^ 0x9E3779B9u);
    const uint32_t x = mix32((uint32_t)idx ^ mix32((uint32_t)seed + hi));
    return (float)(x >> 8) * (1.0f / 16777216.0f);
}
// The part of counter_uniform that depends on the seed and on the high word of the index only,
// and the rest: counter_uniform(seed, idx) == counter_uniform_low(counter_high_part(seed, idx >> 32), idx).
GLV_HD uint32_t counter_high_part(uint64_t seed, uint32_t idx_high) {
    return mix32((uint32_t)seed + mix32(idx_high ^ (uint32_t)(seed >> 32) ^ 0x9E3779B9u));
}
GLV_HD float counter_uniform_low(uint32_t high_part, uint32_t idx_low) {
    return (float)(mix32(idx_low ^ high_part) >> 8) * (1.0f / 16777216.0f);
}
GLV_HD cf phasor_of(float u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return cf{__builtin_amdgcn_cosf(u), __builtin_amdgcn_sinf(u)};
#else
    const float ph = 6.28318530717958647692 * u;
    return cf{cosf(ph), sinf(ph)};
#endif
}
GLV_HD cf counter_phasor(uint64_t seed, uint64_t idx) { return phasor_of(counter_uniform(seed, idx)); }

}  // namespace glv
