// Device kernels of libglvoc (gfx950), included once by glvoc.hip.
//
//   gl_engine_kernel    one wavefront per run of frames: STFT frame -> phase update -> inverse
//                       FFT -> register overlap-add (engine.h).  Modes: the iteration with the
//                       momentum term formed in the time domain (the default: per frame it reads
//                       the target magnitudes and one hop of each of the two newest signals and
//                       writes one hop; no spectrum is ever stored), the same in librosa's
//                       frequency-domain form (keeps the previous spectrum in memory), iSTFT only,
//                       STFT only, the one-frame-per-wave forms for small batches, and measurement
//                       variants.
//   mel_native_kernel   dB -> power -> f32 MFMA product with the inverse mel basis -> iteration
//                       state (magnitudes in native or pair order, initial angles), fused.
//   nnls_fused_kernel   all projected-gradient steps of the non-negative mel inversion in one
//                       launch, the solution resident in registers in MFMA accumulator layout.
//   gemm_nt_kernel      a plain "NT" GEMM on the f32 MFMA with selectable epilogues (mel
//                       inversion with the reference's step boundary, the NNLS start and its
//                       unfused steps, forward mel projection).
//   init / convert / pad / finalize / peak / sc / power kernels: element-wise stages
//                       (elementwise.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "elementwise.h"
#include "engine.h"

using namespace glv;

namespace {

#ifndef GLV_WAVES_PER_BLOCK
#define GLV_WAVES_PER_BLOCK 4
#endif
constexpr int kWavesPerBlock = GLV_WAVES_PER_BLOCK;

// Wave priority inside the frame loop.  A SIMD issues from its oldest wave first, so without
// help the waves of a SIMD finish one after the other and the last runs alone.  1: a wave's
// priority falls at the middle of its run, an eighth before its end and for its last frame,
// which lets the waves behind catch up and all of them end together.  0: priorities untouched.
#ifndef GLV_PRIO
#define GLV_PRIO 1
#endif
// 1: consecutive runs of one utterance that sit in one workgroup hand their boundary blocks over
// through LDS (n_fft 1024 iteration kernels); 0: every run fetches its own.
#ifndef GLV_SHARE_BOUNDARY
#define GLV_SHARE_BOUNDARY 1
#endif
// 1: the kernels that can (n_fft 1024: the first inverse transform and the time-domain-momentum
// iteration without the metric) complete run boundaries inside a workgroup on the producer's side
// when the call asks for it (IterArgs::deposit); 0: never.
#ifndef GLV_DEPOSIT
#define GLV_DEPOSIT 1
#endif

// LDS of a workgroup, in cf units: the table region - for n_fft 1024 long enough to hold, behind
// the pass twiddles (which some kernels keep reading), the deposits of kWavesPerBlock - 1 runs -
// then one exchange buffer per wave, then (n_fft 1024) the shared boundary blocks.
template <int A>
constexpr int deposit_offset() {
    return Geo<A>::TW_A + Geo<A>::TW_B;
}
template <int A, int HB>
constexpr int table_span() {
    using G = Geo<A>;
    constexpr int need = deposit_offset<A>() + (kWavesPerBlock - 1) * Engine<A, HB>::kShareFloats / 2;
    return A == 1 && need > G::TABLES ? need : G::TABLES;
}

__device__ __forceinline__ void wave_sync() {
    // Orders this wave's LDS writes before its later LDS reads (other lanes' data).
    // One wave's DS operations execute in issue order, so no counter wait is needed;
    // the fences only pin the compiler.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- iteration-spanning kernels: waiting for a neighbouring run ----
// A run's progress word holds the number of iterations whose boundary blocks it has written
// through to memory.  The poll is a device-scope load (served past this CU's L1), one per
// wave instruction; between polls the wave sleeps.  The wait is bounded: when it runs out (a
// workgroup of the launch is not resident, say because another process holds part of the
// device) the status word is raised, the wave stops waiting for good and finishes its loop on
// whatever it reads - the call's results are void, and the host sees the status word.
#ifndef GLV_SPIN_LIMIT
#define GLV_SPIN_LIMIT (1 << 17)
#endif
// 1: the loads an iteration's first frame waits for are issued at the end of the iteration before
// (under the wait for its stores) instead of after the preload (measured: no difference)
#ifndef GLV_EARLY_PREFETCH
#define GLV_EARLY_PREFETCH 0
#endif
__device__ __forceinline__ int poll_word(const int* w) {
    return __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// returns false when the wait ran out (or another wave's did)
__device__ __forceinline__ bool wait_progress(const int* word, int want, int* status, int* host_status) {
    for (int spins = 0; spins < GLV_SPIN_LIMIT; ++spins) {
        if (poll_word(word) >= want) return true;
        __builtin_amdgcn_s_sleep(8);
        if ((spins & 255) == 255 && poll_word(status) != 0) return false;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicOr(status, 1);
        __hip_atomic_fetch_or(host_status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return false;
}

// 1: the waves of a workgroup of an iteration-spanning kernel do not meet at barriers after the
// first iteration; what one hands to another through LDS (or through this CU's own stores) is
// announced by a flag word in LDS that the taker waits for where it needs the data.  A run then
// depends on its PREDECESSOR at the start of an iteration and on its successor only frames
// later, so the runs of an utterance fall into a staircase, each a hand-over behind the one
// before it.  0 (the default): three barriers per iteration.  Measured on MI355X at 64 x 430
// frames: same bits, but 35.0 against 33.0 us per iteration - the waves still wait for their
// predecessors every iteration (the time a wave needs for its frames varies with what its SIMD's
// other waves do, and a chain runs at the pace of its slowest link either way), and the staircase
// adds about 100 us to the start and the end of a call.
#ifndef GLV_PERSIST_FLAGS
#define GLV_PERSIST_FLAGS 0
#endif
// flag words in LDS: iterations whose data wave w has (0) published after its preload,
// (1) deposited (its first R-1 blocks' head sums), (2) stored and drained at its end
__device__ __forceinline__ void lds_raise(int* flag, int value, int lane) {
    asm volatile("" ::: "memory");  // the data this announces was written by earlier instructions
    if (lane == 0) *reinterpret_cast<volatile int*>(flag) = value;
}
__device__ __forceinline__ bool lds_wait(const int* flag, int want, int* status, int* host_status) {
    for (int spins = 0; spins < (GLV_SPIN_LIMIT << 3); ++spins) {
        if (*reinterpret_cast<const volatile int*>(flag) >= want) {
            asm volatile("" ::: "memory");
            return true;
        }
        __builtin_amdgcn_s_sleep(1);
        if ((spins & 4095) == 4095 && poll_word(status) != 0) return false;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicOr(status, 1);
        __hip_atomic_fetch_or(host_status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return false;
}

// iteration-spanning kernels: the buffers of iteration k, and (GLV_EARLY_PREFETCH) the loads its
// first frame waits for - they touch nothing a neighbour writes, so they can be issued at the end
// of iteration k - 1, under the wait for its stores
template <int NZH, typename EngT>
__device__ __forceinline__ void enter_iteration(IterArgs& a, EngT& e, int& g_cur, int k) {
    const int g_out = g_cur == 2 ? 0 : g_cur + 1, g_prev = k == 0 ? g_cur : (g_cur == 0 ? 2 : g_cur - 1);
    auto pick = [](float* const (&g)[3], int i) { return i == 0 ? g[0] : (i == 1 ? g[1] : g[2]); };
    a.y_in = pick(a.gen_main, g_cur);
    a.y_in_tail = pick(a.gen_tail, g_cur);
    a.y_prev = pick(a.gen_main, g_prev);
    a.y_prev_tail = pick(a.gen_tail, g_prev);
    a.y_out = pick(a.gen_main, g_out);
    a.y_out_tail = pick(a.gen_tail, g_out);
    a.has_prev = k > 0;
    g_cur = g_out;
#if GLV_EARLY_PREFETCH
    e.prefetch_block_pair(a, e.run.t_begin);
    e.template prefetch_s_pair<NZH>(a, e.run.t_begin);
#endif
}

#ifndef GLV_PERSIST_TWR
#define GLV_PERSIST_TWR 1
#endif
// 1: one frame per wave: the blocks the four frames of a workgroup need are summed once per
// workgroup (Engine::sum_blocks_for_workgroup)
#ifndef GLV_COOP_PARTS
#define GLV_COOP_PARTS 1
#endif
// 1 (with the ring): the blocks a run completes for the next run of its workgroup are written to
// memory by the last iteration only
#ifndef GLV_RING_SKIP_STORES
#define GLV_RING_SKIP_STORES 1
#endif
#ifndef GLV_EARLY_PRELOAD
#define GLV_EARLY_PRELOAD 1
#endif
// 1: iteration-spanning kernels, n_fft 1024: a run that completes its boundary with the next run of
// its workgroup also makes that boundary's hand-over for the next iteration, in LDS
// (Engine::ring_hand_over): after the first iteration no run loads blocks that a wave of its own
// workgroup wrote, and the barrier after the preload goes
#ifndef GLV_PERSIST_RING
#define GLV_PERSIST_RING 1
#endif
// second launch bound = waves per SIMD the register allocator must leave room for (the
// iteration-spanning run kernels of n_fft 1024 are held to three: they sit within a register or two
// of that limit, and a launch that holds a third fewer workgroups does not fit the usual batch)
#ifndef GLV_MIN_WAVES_A1
#define GLV_MIN_WAVES_A1 2
#endif
#ifndef GLV_MIN_WAVES_A2
#define GLV_MIN_WAVES_A2 2
#endif
#ifdef GLV_WAVE_TIMES
// measurement build: every wave of the iteration kernel leaves time stamps (100 MHz clock; slots 18 and
// 19: the shader clock at its start and end)
__device__ unsigned long long glv_wave_times[20 * 16384];
#define GLV_STAMP(k)                                                                          \
    if (stamp_on && (MODE == MODE_ITER_LIN || MODE == MODE_ITER_LIN_Z || MODE == MODE_ITER1 || MODE == MODE_ITER_LIN_P || MODE == MODE_ITER_LIN_Z_P) && (threadIdx.x & 63) == 0) {           \
        const int w_ = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);                      \
        if (w_ < 16384 && (k) < 18) glv_wave_times[20 * w_ + (k)] = wall_clock64();                        \
        if (w_ < 16384 && ((k) == 0 || (k) == 3)) glv_wave_times[20 * w_ + 18 + ((k) == 3)] = clock64(); \
    }
#else
#define GLV_STAMP(k)
#endif

template <int A, int HB, int MODE, bool SC>
__global__ __launch_bounds__(64 * kWavesPerBlock, (A == 1 ? ((MODE == MODE_ITER_LIN_P || MODE == MODE_ITER_LIN_Z_P) ? 3 : GLV_MIN_WAVES_A1) : GLV_MIN_WAVES_A2)) void
gl_engine_kernel(IterArgs a) {
    using G = Geo<A>;
    // Kernels that can complete run boundaries on the producer's side.  They keep every table in
    // registers (the default iteration kernels have the room anyway), so that after the set-up
    // the table region of LDS is free to hold the deposits.
    // all iterations of the call in this launch (the frame loop below runs once per iteration)
    constexpr bool kPersist1 = MODE == MODE_ITER1_P;  // ... of the one-frame-per-wave form
    constexpr bool kPersist = MODE == MODE_ITER_LIN_P || MODE == MODE_ITER_LIN_Z_P || kPersist1;
    constexpr bool kLin = MODE == MODE_ITER_LIN || MODE == MODE_ITER_LIN_Z || MODE == MODE_TRAFFIC_LIN ||
                          MODE == MODE_TRAFFIC_LIN_Z || (kPersist && !kPersist1);
    constexpr bool kCanDeposit = (kLin || MODE == MODE_ISTFT) && !SC && A == 1 && GLV_DEPOSIT && kWavesPerBlock > 1;
    // (GLV_PERSIST_TWR = 2: the iteration-spanning kernels read the second pass's twiddles from
    // LDS, which saves them 16 registers; 1: in registers like the other kernels)
    using Eng = Engine<A, HB, kPersist1 ? 0 : kPersist ? GLV_PERSIST_TWR : (((kLin && !SC) || kCanDeposit) ? 1 : 0),
                       kPersist && !kPersist1 && A == 1>;
    constexpr int kSpan = table_span<A, HB>();
    static_assert(!kCanDeposit || deposit_offset<A>() + (kWavesPerBlock - 1) * Eng::kShareFloats / 2 <= kSpan,
                  "the deposits of a workgroup must fit the table region");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf* tables = reinterpret_cast<cf*>(smem);
    if (kPersist) {
        // what these kernels are never launched with, made known to the compiler
        a.raw_in = 0;
        a.t_origin = 0;
        a.draw = 0;
        a.sc_partial = nullptr;
    }
#ifdef GLV_WAVE_TIMES
    // iteration-spanning kernels stamp their middle iteration: 0 its start, 1 neighbours waited
    // for, 2 preload issued, 4.. the frames, 3 its end
    bool stamp_on = !(MODE == MODE_ITER_LIN_P || MODE == MODE_ITER_LIN_Z_P);
#endif
    GLV_STAMP(0)
    // One frame per wave: a workgroup whose waves hold consecutive frames of one utterance sums
    // every block those frames need ONCE (R + 3 blocks instead of 4 x R, each from its R
    // partial-sum buffers) into the exchange buffers, which the transforms do not use yet - and
    // asks for them first thing, so that their way from memory passes under the staging of the
    // tables (the barrier behind the staging is also the one behind these sums).
    bool coop = false;
    int t_first = 0;
    Eng e;
    if constexpr (MODE == MODE_ITER1 && GLV_COOP_PARTS) {
        const int wave0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), first = (int)blockIdx.x * kWavesPerBlock;
        if (first + kWavesPerBlock <= a.n_runs) {
            const int u0 = __builtin_amdgcn_readfirstlane(a.runs[first].utt);
            const int u3 = __builtin_amdgcn_readfirstlane(a.runs[first + kWavesPerBlock - 1].utt);
            t_first = __builtin_amdgcn_readfirstlane(a.runs[first].t_begin);
            coop = u0 == u3;
        }
        if (coop) {
            RunDesc r0 = a.runs[first + wave0];
            r0.utt = __builtin_amdgcn_readfirstlane(r0.utt);
            r0.t_begin = __builtin_amdgcn_readfirstlane(r0.t_begin);
            r0.t_end = __builtin_amdgcn_readfirstlane(r0.t_end);
            r0.n_frames = __builtin_amdgcn_readfirstlane(r0.n_frames);
            e.init_run(a, threadIdx.x & 63, r0);
            e.template sum_blocks_for_workgroup<kWavesPerBlock>(a, t_first, wave0, tables + kSpan);
        }
    }
    // The run kernels that take one launch per iteration (n_fft 1024) likewise ask for their run's
    // first R-1 blocks before the tables are staged: which halves they read depends on the run
    // table alone.
    constexpr bool kEarly = GLV_EARLY_PRELOAD && !SC && A == 1 && GLV_SHARE_BOUNDARY &&
                            (MODE == MODE_ITER_LIN || MODE == MODE_ITER_LIN_Z);
    if constexpr (kEarly) {
        const int wave0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int wg0 = MODE == MODE_ITER_LIN ? a.wg_first + (int)blockIdx.x * (1 + a.wg_skip) : (int)blockIdx.x;
        const int ri0 = wg0 * kWavesPerBlock + wave0;
        if (ri0 < a.n_runs) {
            RunDesc r0 = a.runs[ri0];
            r0.utt = __builtin_amdgcn_readfirstlane(r0.utt);
            r0.t_begin = __builtin_amdgcn_readfirstlane(r0.t_begin);
            r0.t_end = __builtin_amdgcn_readfirstlane(r0.t_end);
            r0.n_frames = __builtin_amdgcn_readfirstlane(r0.n_frames);
            e.init_run(a, threadIdx.x & 63, r0);
            if (kCanDeposit && a.deposit != 0) {
                cf* area = tables + deposit_offset<A>();
                if (wave0 > 0 && __builtin_amdgcn_readfirstlane(a.runs[ri0 - 1].utt) == r0.utt)
                    e.deposit_out = area + (wave0 - 1) * (Eng::kShareFloats / 2);
                if (wave0 + 1 < kWavesPerBlock && ri0 + 1 < a.n_runs &&
                    __builtin_amdgcn_readfirstlane(a.runs[ri0 + 1].utt) == r0.utt)
                    e.deposit_in = area + wave0 * (Eng::kShareFloats / 2);
            }
            e.template preload_lin<SC>(a);
        }
    }
    {
        // stage the tables: all of a thread's loads are issued before the first is stored, so
        // the block pays one memory latency, not one per round
        static_assert(G::TABLES % 2 == 0, "tables are copied 16 bytes at a time");
        constexpr int kVec = G::TABLES / 2, kThreads = 64 * kWavesPerBlock;
        constexpr int kRounds = (kVec + kThreads - 1) / kThreads;
        const f4* src = reinterpret_cast<const f4*>(a.tables);
        f4* dst = reinterpret_cast<f4*>(smem);
        f4 tmp[kRounds];
#pragma unroll
        for (int k = 0; k < kRounds; ++k) {
            const int i = threadIdx.x + k * kThreads;
            tmp[k] = src[i < kVec ? i : kVec - 1];
        }
#pragma unroll
        for (int k = 0; k < kRounds; ++k) {
            const int i = threadIdx.x + k * kThreads;
            if (i < kVec) dst[i] = tmp[k];
        }
    }
    // the waves' flag words (iteration-spanning kernels), behind the shared boundary blocks
    int* lflag = nullptr;
    if constexpr (kPersist && !kPersist1 && A == 1 && GLV_PERSIST_FLAGS) {
        lflag = reinterpret_cast<int*>(tables + table_span<A, HB>() + kWavesPerBlock * G::XBUF +
                                       (kWavesPerBlock + 1) * (Engine<A, HB>::kShareFloats / 2));
        if (threadIdx.x < 3 * kWavesPerBlock) lflag[threadIdx.x] = 0;
    }
    __syncthreads();
    GLV_STAMP(1)

    // wave-uniform values are forced into scalar registers: the compiler cannot see that
    // threadIdx.x >> 6 (and everything loaded through it) is the same in all 64 lanes
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // (a time shard launches the workgroups at its two edges apart from those between them)
    const int wg = (MODE == MODE_ITER_LIN && !SC) ? a.wg_first + (int)blockIdx.x * (1 + a.wg_skip) : (int)blockIdx.x;
    const int ri = wg * kWavesPerBlock + wave;
    // kernels whose waves hand boundary blocks to each other through LDS meet at a barrier after
    // the preload: there a wave without a run has to stay until then
    constexpr bool kShare = kLin && !SC && A == 1 && GLV_SHARE_BOUNDARY;
    // iteration-spanning kernels whose waves exchange blocks through LDS or through this CU's own
    // stores: one more barrier separates an iteration's last stores from the next one's first loads
    constexpr bool kFlags = kPersist && !kPersist1 && A == 1 && GLV_PERSIST_FLAGS;
    // (n_fft 2048 too: there the waves of a workgroup exchange both halves of their boundary blocks
    // through this CU's own plain stores)
    constexpr bool kEndBarrier = kPersist && !kPersist1 && !kFlags;
    constexpr bool kRing = kPersist && !kPersist1 && Eng::kRingShape && GLV_PERSIST_RING && !GLV_PERSIST_FLAGS && GLV_PERSIST_TWR == 1;
    const bool live = ri < a.n_runs;
    const bool deposit = kCanDeposit && a.deposit != 0;  // the same in every wave of the launch
    const int n_outer = kPersist ? a.n_iter : 1;
    if (kPersist && a.census) {
        // how many workgroups of this launch are resident together: every workgroup arrives once
        // and waits (bounded) until all have; progress[0] counts arrivals, progress[1] the
        // workgroups that saw every other one arrive
        if (threadIdx.x == 0) {
            atomicAdd(a.progress, 1);
            bool all = false;
            for (int spins = 0; spins < (1 << 14) && !all; ++spins) {
                all = poll_word(a.progress) >= (int)gridDim.x;
                if (!all) __builtin_amdgcn_s_sleep(16);
            }
            if (all) atomicAdd(a.progress + 1, 1);
        }
        return;
    }
    if (!live) {
        // a wave without a run only keeps the workgroup's barriers company
        for (int k = 0; k < n_outer; ++k) {
            if ((kShare && (!kRing || k == 0)) || (deposit && k == 0)) __syncthreads();  // after the set-up
            if (kFlags) break;                                   // ... the only barrier there is
            if (deposit) __syncthreads();                        // before the tails are completed
            if (kEndBarrier) __syncthreads();
        }
        return;
    }
    cf* xb = tables + kSpan + wave * G::XBUF;
    cf* shared_blocks = tables + kSpan + kWavesPerBlock * G::XBUF;  // [wave][kShareFloats / 2]

    RunDesc rd = a.runs[live ? ri : 0];
    rd.utt = __builtin_amdgcn_readfirstlane(rd.utt);
    rd.t_begin = __builtin_amdgcn_readfirstlane(rd.t_begin);
    rd.t_end = __builtin_amdgcn_readfirstlane(rd.t_end);
    rd.n_frames = __builtin_amdgcn_readfirstlane(rd.n_frames);
    e.init(a, tables, lane, rd);
    if (kCanDeposit && deposit) {
        // wave w > 0 deposits for wave w - 1 when its run continues that wave's utterance
        // [wave - 1][kShareFloats / 2], over the tables that every wave by then holds in registers
        cf* area = tables + deposit_offset<A>();
        if (wave > 0 && __builtin_amdgcn_readfirstlane(a.runs[ri - 1].utt) == rd.utt)
            e.deposit_out = area + (wave - 1) * (Eng::kShareFloats / 2);
        if (wave + 1 < kWavesPerBlock && ri + 1 < a.n_runs &&
            __builtin_amdgcn_readfirstlane(a.runs[ri + 1].utt) == rd.utt)
            e.deposit_in = area + wave * (Eng::kShareFloats / 2);
    }
    if (!kShare && deposit) __syncthreads();  // every wave has taken its tables out of LDS
    // Iteration-spanning kernels: a boundary with the previous / next run of the utterance that
    // is not completed through LDS goes through memory, i.e. this run reads what that run wrote
    // in the previous iteration of the SAME launch (and the other way round).  Such blocks are
    // written through (Engine::heads_shared / tails_shared), the writer raises its progress word
    // once they have left, and the reader waits for that word before an iteration's first load,
    // then drops this CU's cached copies of anything older.
    bool wait_prev = false, wait_next = false;
    if (kPersist && !kPersist1) {
        wait_prev = rd.t_begin > 0 && e.deposit_out == nullptr;
        wait_next = rd.t_end < rd.n_frames && e.deposit_in == nullptr;
        if (!kShare) {
            // no boundary blocks in LDS (n_fft 2048): a neighbouring run that is a wave of this
            // workgroup still needs no progress word - its plain stores reach this CU's loads
            // once they have drained and the closing barrier of the iteration has been passed
            if (wave > 0 && __builtin_amdgcn_readfirstlane(a.runs[ri - 1].utt) == rd.utt) wait_prev = false;
            if (wave + 1 < kWavesPerBlock && ri + 1 < a.n_runs &&
                __builtin_amdgcn_readfirstlane(a.runs[ri + 1].utt) == rd.utt)
                wait_next = false;
        }
        e.heads_shared = wait_prev;
        e.tails_shared = wait_next;
    }
    bool waiting = true;  // false once a wait has run out
    int g_cur = 0;        // generation the coming iteration reads
    constexpr bool kRc = MODE == MODE_ITER_RC || MODE == MODE_ITER_RC_FIRST;
    constexpr bool kSingle = MODE == MODE_ITER1 || MODE == MODE_ISTFT1 || kPersist1;
    // mirrors of a lane's pairs whose magnitude can be non-zero: all, or the upper half
    constexpr int kNzh =
        (MODE == MODE_ITER_LIN_Z || MODE == MODE_TRAFFIC_LIN_Z || MODE == MODE_ITER_LIN_Z_P) ? Eng::NP / 2 : Eng::NP;
    // one frame per wave: the two generations of partial-sum buffers, swapped every iteration
    const float* parts_even = a.part_in;
    float* parts_odd = a.part_out;
    if constexpr (kPersist && !kPersist1) {
        // the blocks after the run's last frame always come from LDS: the next run's published
        // preload, or this run's own (preload_next); slot wave + 1 either way
        if (A == 1) e.share_next = shared_blocks + (wave + 1) * (Eng::kShareFloats / 2);
        if constexpr (kRing) {
            // the blocks of the iteration before, one set per boundary inside the workgroup: in the
            // slot no run publishes to (wave 0 has no predecessor here), over the two twiddle tables
            // (every wave holds them in registers) and behind the flag words
            if (e.deposit_in != nullptr)
                e.ring_prev = wave == 0   ? shared_blocks
                              : wave == 1 ? tables
                                          : shared_blocks + (kWavesPerBlock + 1) * (Eng::kShareFloats / 2) + 2 * kWavesPerBlock;
        }
        enter_iteration<kNzh>(a, e, g_cur, 0);
    }
    // (not a loop at all unless the kernel spans iterations: inside a loop the compiler hoists every
    // address computation of the body to its head, where they fight over the scalar registers)
    int k = 0;
    do {
    if (kPersist) {
        // the run descriptor passes through an opaque point every iteration, for the same reason:
        // what is derived from it is computed where it is used, not kept across the loop
        asm volatile("" : "+s"(e.run.utt), "+s"(e.run.t_begin), "+s"(e.run.t_end), "+s"(e.run.n_frames));
    }
#ifdef GLV_WAVE_TIMES
    if (kPersist) stamp_on = k == a.n_iter / 2;
    if (kPersist) { GLV_STAMP(0) }
#endif
    if constexpr (kPersist1) {
        a.part_in = (k & 1) ? parts_odd : parts_even;
        a.part_out = (k & 1) ? const_cast<float*>(parts_even) : parts_odd;
        a.has_prev = k > 0;
        if (k > 0 && waiting) {
            // lane l looks at frame t - (R-1) + l: every frame whose blocks overlap this one's must
            // have stored iteration k - 1 (and, with that, have read what this wave now overwrites)
            const int d = lane - (Eng::R - 1), tn = e.run.t_begin + d;
            const bool look = lane < 2 * Eng::R - 1 && d != 0 && tn >= 0 && tn < e.run.n_frames;
            const int* word = a.progress + (int64_t)(ri + (look ? d : 0)) * kProgressStride;
            int spins = 0;
            for (;; ++spins) {
                const bool ok = !look || poll_word(word) >= k;
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(4);
                if (spins >= GLV_SPIN_LIMIT || ((spins & 255) == 255 && poll_word(a.status) != 0)) {
                    if (spins >= GLV_SPIN_LIMIT && lane == 0) {
                        atomicOr(a.status, 1);
                        __hip_atomic_fetch_or(a.host_status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    waiting = false;
                    break;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    }
    if (kPersist && !kPersist1) {
        if (k > 0) {
            e.begin_iteration();
            // the run before this one has stored its part of the boundary blocks (iteration k - 1):
            // a flag word in LDS when it is a wave of this workgroup, else its progress word
            if (kFlags && e.deposit_out != nullptr && waiting)
                waiting = lds_wait(lflag + 2 * kWavesPerBlock + wave - 1, k, a.status, a.host_status);
            if (wait_prev || wait_next) {
                if (waiting && wait_prev) waiting = wait_progress(a.progress + (int64_t)(ri - 1) * kProgressStride, k, a.status, a.host_status);
                // ... and the run after it the blocks this run's last frames read.  With flags that is
                // its HEADS word, raised a few frames into its iteration: this run never waits for a
                // run that is behind it
                if (waiting && wait_next)
                    waiting = wait_progress(a.progress + (int64_t)(ri + 1) * kProgressStride + (kFlags ? kProgressHeads : 0), k,
                                            a.status, a.host_status);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
        }
#ifdef GLV_WAVE_TIMES
        GLV_STAMP(1)
#endif
    }
    if (kSingle) {
        // one frame per wave: the run is the frame
        const int t = e.run.t_begin;
        if (MODE == MODE_ITER1 || kPersist1) {
            if (coop) {
                // (summed before the tables were staged, see the kernel's start)
                e.take_frame_from_area(t, t_first, tables + kSpan);
                __syncthreads();  // every wave has its blocks: the buffers are the transforms' again
                e.template ph0_single<true, true>(a, t, xb);
            } else {
                e.template ph0_single<true>(a, t, xb);
            }
            GLV_STAMP(4)
            wave_sync();
            e.ph1_read(xb);
            wave_sync();
            e.ph1_write(xb);
            wave_sync();
            e.ph2_read(xb);
            wave_sync();
            e.ph2_write(xb);
            wave_sync();
            e.template ph3_update<SC, 0, false>(a, t, xb);
            GLV_STAMP(5)
        } else {
            e.ph3_load(a, t);
        }
        wave_sync();
        e.ph3_write(xb);
        wave_sync();
        e.ph4(xb, xb);
        wave_sync();
        e.ph5_read(xb);
        wave_sync();
        e.ph5_write(xb);
        wave_sync();
        GLV_STAMP(6)
        e.template ph6_single<kPersist1>(a, t, xb);
        if (!kPersist1) { GLV_STAMP(3) }
    }
    if (!kSingle && !kLin && MODE != MODE_ISTFT) {
        e.preload(a);
        e.prefetch_block(a, e.run.t_begin);
    }
    if (MODE == MODE_ITER_RC) {
        e.preload_prev(a);
        e.prefetch_block_prev(a, e.run.t_begin);
    }
    if constexpr (kShare) {
        cf* mine = shared_blocks + wave * (Eng::kShareFloats / 2);
        if constexpr (kPersist) {
            // a run whose successor is not an LDS neighbour makes the blocks after its last frame
            // itself (into the slot that neighbour would have published to), and only a run with
            // an LDS predecessor publishes: one writer per slot
            if (e.deposit_in == nullptr) {
                const cf* keep = e.share_next;
                e.share_next = nullptr;
                e.preload_next(a, mine + Eng::kShareFloats / 2);
                e.share_next = keep;
            }
            GLV_SCHED_FENCE();
            if (kRing && k > 0 && e.deposit_out != nullptr) {
                // the run before this one left these blocks in the slot at the end of its iteration
                e.take_ring_preload(mine);
            } else {
                e.template preload_lin<SC>(a);
                if (e.deposit_out != nullptr) {
                    e.publish_preload(mine);
                    if (kFlags) lds_raise(lflag + wave, k + 1, lane);
                }
            }
            GLV_SCHED_FENCE();
        } else if (live) {
            if (!kEarly) e.template preload_lin<SC>(a);  // (else done before the tables were staged)
            e.publish_preload(mine);
        }
        // (with flags: once, so that every wave holds its tables in registers before the first deposit)
        if ((!kFlags && !kRing) || k == 0) __syncthreads();
        // the next run, if it continues this utterance inside this workgroup
        if (!kPersist && wave + 1 < kWavesPerBlock && ri + 1 < a.n_runs && !a.raw_in) {
            const int next_utt = __builtin_amdgcn_readfirstlane(a.runs[ri + 1].utt);
            if (next_utt == e.run.utt) e.share_next = mine + Eng::kShareFloats / 2;
        }
    }
#if !GLV_EARLY_PREFETCH
    if (kPersist && !kPersist1) {
        // (n_fft 2048: no boundary blocks in LDS; every boundary goes through memory, both halves)
        if (!kShare) e.template preload_lin<SC>(a);
        e.prefetch_block_pair(a, e.run.t_begin);
        e.template prefetch_s_pair<kNzh>(a, e.run.t_begin);
    }
#endif
    if (kLin && !kPersist) {
        if (!kShare) e.template preload_lin<SC>(a);
        e.prefetch_block(a, e.run.t_begin);
        e.prefetch_block_prev(a, e.run.t_begin);
        e.template prefetch_s_pair<kNzh>(a, e.run.t_begin);
    }
    if (MODE == MODE_ITER || MODE == MODE_TRAFFIC) e.prefetch_spec(a, e.run.t_begin);
    if (kRc) e.prefetch_s(a, e.run.t_begin);
    GLV_STAMP(2)
    constexpr bool kPrio = GLV_PRIO == 1;
    const int prio_n = e.run.t_end - e.run.t_begin;
    const int prio_q1 = e.run.t_begin + (prio_n + 1) / 2, prio_q2 = e.run.t_end - 1 - (prio_n + 7) / 8,
              prio_q3 = e.run.t_end - 1;
    if (kPrio) __builtin_amdgcn_s_setprio(3);
    for (int t = e.run.t_begin; !kSingle && t < e.run.t_end; ++t) {
        if (kPrio) {
            if (t == prio_q1) __builtin_amdgcn_s_setprio(2);
            else if (t == prio_q2) __builtin_amdgcn_s_setprio(1);
            else if (t == prio_q3) __builtin_amdgcn_s_setprio(0);
        }
        GLV_STAMP(4 + t - e.run.t_begin)
        if constexpr (kFlags) {
            // the first frame that takes a block from the next run's published preload
            if (e.deposit_in != nullptr && waiting && (t == e.run.t_end - Eng::R + 1 || (t == e.run.t_begin && t > e.run.t_end - Eng::R + 1)))
                waiting = lds_wait(lflag + wave + 1, k + 1, a.status, a.host_status);
            // this run's first R-1 blocks left two frames ago: by now they have arrived, and the
            // wait costs nothing that the frame's own loads would not cost it anyway
            if (wait_prev && t == e.run.t_begin + Eng::R) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0)
                    __hip_atomic_store(a.progress + (int64_t)ri * kProgressStride + kProgressHeads, k + 1, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (MODE == MODE_TRAFFIC) {
            e.traffic_frame(a, t);
            continue;
        }
        if (MODE == MODE_TRAFFIC_LIN || MODE == MODE_TRAFFIC_LIN_Z) {
            e.template traffic_frame_lin<kNzh>(a, t, xb);
            continue;
        }
        if (MODE == MODE_ITER_RC) {
            // forward transform of the same frame of the previous signal -> tprev
            e.ph0_prev(a, t, xb);
            wave_sync();
            e.ph1_read(xb);
            wave_sync();
            e.ph1_write(xb);
            wave_sync();
            e.ph2_read(xb);
            wave_sync();
            e.ph2_write(xb);
            wave_sync();
            e.ph3_prev(xb);
            wave_sync();
        }
        if (kLin) {
            e.template lin_advance<SC>(a, t);
            if (SC) {
                // the metric needs the spectrum of the signal itself: a transform of its own
                e.lin_window_signal(xb);
                wave_sync();
                e.ph1_read(xb);
                wave_sync();
                e.ph1_write(xb);
                wave_sync();
                e.ph2_read(xb);
                wave_sync();
                e.pair_store(xb);
                wave_sync();
                e.ph3_metric_pair(xb);
                wave_sync();
            }
            e.lin_window_combined(xb);
            e.template lin_shift<SC>();
            wave_sync();
            e.ph1_read(xb);
            wave_sync();
            e.ph1_write(xb);
            wave_sync();
            e.ph2_read(xb);
            wave_sync();
            e.pair_store(xb);
            wave_sync();
            // v = STFT(y_k - c y_{k-1}): normalise, scale by S, start the way back - on mirror
            // pairs of bins; nothing to fetch or store
            e.template ph3_pair<kNzh>(a, t, xb, xb);
            wave_sync();
            e.ph4_pair(xb, xb);
            wave_sync();
            e.ph5_read(xb);
            wave_sync();
            e.ph5_write(xb);
            wave_sync();
            e.ph6(a, t, xb);
            wave_sync();
            if constexpr (kFlags) {
                if (e.deposit_out != nullptr && t == e.run.t_begin + Eng::R - 2)
                    lds_raise(lflag + kWavesPerBlock + wave, k + 1, lane);
            }
            continue;
        } else if (MODE != MODE_ISTFT) {
            e.ph0(a, t, xb);
            wave_sync();
            e.ph1_read(xb);
            wave_sync();
            e.ph1_write(xb);
            wave_sync();
            e.ph2_read(xb);
            wave_sync();
            e.ph2_write(xb);
            wave_sync();
            if (MODE == MODE_STFT) {
                e.ph3_stft(a, t, xb);
                wave_sync();
                continue;
            }
            e.template ph3_update<SC, (MODE == MODE_ITER_RC ? 1 : (MODE == MODE_ITER_RC_FIRST ? 2 : 0))>(a, t, xb);
        } else {
            e.ph3_load(a, t);
        }
        wave_sync();
        e.ph3_write(xb);
        wave_sync();
        e.ph4(xb, xb);
        wave_sync();
        e.ph5_read(xb);
        wave_sync();
        e.ph5_write(xb);
        wave_sync();
        e.ph6(a, t, xb);
        wave_sync();
    }
    if constexpr (kFlags) {
        // the deposits this wave's tails need were made by the next run frames ago
        if (e.deposit_in != nullptr && waiting)
            waiting = lds_wait(lflag + kWavesPerBlock + wave + 1, k + 1, a.status, a.host_status);
    } else {
        if (deposit) __syncthreads();  // the deposits this wave's tails need were made frames ago; now they are visible
    }
    if constexpr (kRing) {
        if (e.ring_prev != nullptr) e.fetch_ring_rows(a);
    }
    if (MODE != MODE_STFT && !kSingle) e.flush_tails(a, !(kRing && GLV_RING_SKIP_STORES) || e.ring_prev == nullptr || k + 1 == n_outer);
    if constexpr (kRing) {
        if (e.ring_prev != nullptr) e.ring_hand_over(a, const_cast<cf*>(e.share_next), k == 0);
    }
    if constexpr (kPersist1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0)
            __hip_atomic_store(a.progress + (int64_t)ri * kProgressStride, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if constexpr (kPersist && !kPersist1) {
        if (k + 1 < n_outer) enter_iteration<kNzh>(a, e, g_cur, k + 1);
        // every store of this iteration has left (the blocks other workgroups read went out
        // written through); only then do the run's progress words and its flag word rise
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if ((wait_prev || wait_next) && lane == 0) {
            __hip_atomic_store(a.progress + (int64_t)ri * kProgressStride, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // (a run too short to have raised its heads word inside the frame loop)
            if (kFlags && wait_prev)
                __hip_atomic_store(a.progress + (int64_t)ri * kProgressStride + kProgressHeads, k + 1, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        if (kFlags) lds_raise(lflag + 2 * kWavesPerBlock + wave, k + 1, lane);
        if (kEndBarrier) __syncthreads();
#ifdef GLV_WAVE_TIMES
        GLV_STAMP(3)
#endif
    }
    } while (kPersist && ++k < n_outer);  // iterations of an iteration-spanning kernel (one pass otherwise)
    if (!kPersist) { GLV_STAMP(3) }
    if (SC) {
        float num = e.sc_num, den = e.sc_den;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            num += __shfl_xor(num, off);
            den += __shfl_xor(den, off);
        }
        if (lane == 0 && a.sc_partial) {
            a.sc_partial[2 * ri] = num;
            a.sc_partial[2 * ri + 1] = den;
        }
    }
}

struct ElemArgs {
    LayoutArgs L;
    int batch;
};

template <int A>
__global__ void init_native_kernel(ElemArgs ea, const float* S_nat, const cf* phase_nat, uint64_t seed,
                                   float inv_n, float* S_native, cf* spec) {
    const int P = ea.L.P;
    const int64_t total = (int64_t)ea.batch * ea.L.frames_stride * P;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % P);
        const int64_t fr = i / P;
        const int t = (int)(fr % ea.L.frames_stride), b = (int)(fr / ea.L.frames_stride);
        if (t < (ea.L.frames_local > 0 ? ea.L.frames_local : ea.L.n_frames[b]))
            init_native_element<A>(ea.L, S_nat, phase_nat, seed, inv_n, S_native, spec, b, t, q);
    }
}

// the phasors a call without an initial phase draws, written out in caller order (B, T, F)
__global__ void drawn_phase_kernel(ElemArgs ea, uint64_t seed, cf* out) {
    const int F = ea.L.F;
    const int64_t total = (int64_t)ea.batch * ea.L.frames_stride * F;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % F);
        const int64_t fr = i / F;
        out[i] = drawn_phase(ea.L, seed, (int)(fr / ea.L.frames_stride), (int)(fr % ea.L.frames_stride), k);
    }
}

template <int A, bool TO_NATIVE>
__global__ void spec_convert_kernel(ElemArgs ea, const cf* src, float inv_n, cf* dst) {
    const int P = ea.L.P;
    const int64_t total = (int64_t)ea.batch * ea.L.frames_stride * P;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % P);
        const int64_t fr = i / P;
        const int t = (int)(fr % ea.L.frames_stride), b = (int)(fr / ea.L.frames_stride);
        if (t >= (ea.L.frames_local > 0 ? ea.L.frames_local : ea.L.n_frames[b])) continue;
        if (TO_NATIVE)
            spec_to_native_element<A>(ea.L, src, inv_n, dst, b, t, q);
        else
            spec_to_natural_element<A>(ea.L, src, dst, b, t, q);
    }
}

// caller signal (B, y_stride) -> padded overlap-add coordinates, offset n_fft/2.  Row b holds
// n_samples[b] samples (or hop * (n_frames[b] - 1) without n_samples); the rest of the block after
// the last whole hop is zero-filled, so that the transform sees the trailing partial hop.
__global__ void pad_signal_kernel(ElemArgs ea, const float* y, int64_t y_stride, const int* n_samples,
                                  float* ypad) {
    const int b = blockIdx.y;
    const int64_t whole = (int64_t)ea.L.hop * (ea.L.n_frames[b] - 1);
    const int64_t len = n_samples ? n_samples[b] : whole;
    for (int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; n < whole + ea.L.hop;
         n += (int64_t)gridDim.x * blockDim.x)
        ypad[(int64_t)b * ea.L.y_utt_stride + ea.L.n_fft / 2 + n] = n < len ? y[(int64_t)b * y_stride + n] : 0.f;
}

__global__ __launch_bounds__(256) void finalize_kernel(ElemArgs ea, const float* y_main, const float* y_tail,
                                                       const unsigned char* tail_flag, int blocks_stride,
                                                       const float* w2, const float* iwss,
                                                       const float* iwss_head, const float* iwss_tail,
                                                       const float* iwss_small, float* out, int64_t out_stride,
                                                       int64_t n_begin, unsigned* peak_bits) {
    // samples [n_begin, n_begin + out_stride) of every row (a time shard: its buffers start at
    // block L.block_origin, `out` at sample L.out_origin of the waveform).  With peak_bits the
    // row maximum of |y| is gathered on the way (one atomic per block, see peak_reduce_kernel).
    __shared__ float wave_max[4];
    const int b = blockIdx.y;
    float m = 0.f;
    for (int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; n < out_stride;
         n += (int64_t)gridDim.x * blockDim.x)
        m = fmaxf(m, finalize_sample(ea.L, y_main, y_tail, tail_flag, blocks_stride, w2, iwss, iwss_head,
                                     iwss_tail, iwss_small, out, out_stride, b, n_begin + n));
    if (!peak_bits) return;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(wave_max[0], wave_max[1]), fmaxf(wave_max[2], wave_max[3]));
        atomicMax(&peak_bits[b], __float_as_uint(m));
    }
}

// finalize_kernel for whole rows, four consecutive samples per thread (16-byte accesses; the hop
// is a power of two and a multiple of 4, so the four share their block and the index arithmetic
// is shifts).  The same operations per sample, hence the same bits.  Needs out and out_stride
// aligned to four floats.
__global__ __launch_bounds__(256) void finalize4_kernel(ElemArgs ea, const float* y_main, const float* y_tail,
                                                        const unsigned char* tail_flag, int blocks_stride,
                                                        const float* iwss, const float* iwss_head,
                                                        const float* iwss_tail, const float* iwss_small,
                                                        float* out, int64_t out_stride, int hop_shift,
                                                        unsigned* peak_bits) {
    __shared__ float wave_max[4];
    const int b = blockIdx.y;
    const LayoutArgs& L = ea.L;
    const int T = L.n_frames[b];
    const int64_t len = (int64_t)L.hop * (T - 1);
    float m = 0.f;
    for (int64_t n = 4 * (blockIdx.x * (int64_t)blockDim.x + threadIdx.x); n < out_stride;
         n += 4 * (int64_t)gridDim.x * blockDim.x) {
        f4 val = f4{0.f, 0.f, 0.f, 0.f};
        if (n < len) {
            const int64_t p = n + L.n_fft / 2;
            const int j = (int)(p >> hop_shift), s = (int)(p & (L.hop - 1));
            const int64_t at = (int64_t)b * L.y_utt_stride + p;
            if (j < T) val = *reinterpret_cast<const f4*>(y_main + at);
            if (tail_flag[(int64_t)b * blocks_stride + j]) {
                const f4 t = *reinterpret_cast<const f4*>(y_tail + at);
                val = f4{val.x + t.x, val.y + t.y, val.z + t.z, val.w + t.w};
            }
            const f4 iw = *reinterpret_cast<const f4*>(iwss_row(iwss, iwss_head, iwss_tail, iwss_small, L.hop, L.R, T, j) + s);
            val = f4{val.x * iw.x, val.y * iw.y, val.z * iw.z, val.w * iw.w};
        }
        *reinterpret_cast<f4*>(out + (int64_t)b * out_stride + n) = val;
        m = fmaxf(fmaxf(m, fmaxf(fabsf(val.x), fabsf(val.y))), fmaxf(fabsf(val.z), fabsf(val.w)));
    }
    if (!peak_bits) return;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(wave_max[0], wave_max[1]), fmaxf(wave_max[2], wave_max[3]));
        atomicMax(&peak_bits[b], __float_as_uint(m));
    }
}

// finalize_kernel for the one-frame-per-wave modes: the sample is the sum of R partial buffers
__global__ __launch_bounds__(256) void finalize_parts_kernel(ElemArgs ea, const float* parts, int64_t part_stride,
                                                             const float* iwss, const float* iwss_head,
                                                             const float* iwss_tail, const float* iwss_small,
                                                             float* out, int64_t out_stride, unsigned* peak_bits) {
    __shared__ float wave_max[4];
    const int b = blockIdx.y;
    float m = 0.f;
    for (int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; n < out_stride;
         n += (int64_t)gridDim.x * blockDim.x)
        m = fmaxf(m, finalize_sample_parts(ea.L, parts, part_stride, iwss, iwss_head, iwss_tail, iwss_small, out,
                                           out_stride, b, n));
    if (!peak_bits) return;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(wave_max[0], wave_max[1]), fmaxf(wave_max[2], wave_max[3]));
        atomicMax(&peak_bits[b], __float_as_uint(m));
    }
}

__global__ __launch_bounds__(256) void peak_reduce_kernel(const float* y, int64_t y_stride,
                                                          const int* n_samples, unsigned* peak_bits) {
    __shared__ float wave_max[4];
    const int b = blockIdx.y;
    const int64_t len = n_samples[b];
    const float* row = y + (int64_t)b * y_stride;
    float m = 0.f;
    for (int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; n < len;
         n += (int64_t)gridDim.x * blockDim.x)
        m = fmaxf(m, fabsf(row[n]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
    __syncthreads();
    // one atomic per block; non-negative floats order like their bit patterns
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(wave_max[0], wave_max[1]), fmaxf(wave_max[2], wave_max[3]));
        atomicMax(&peak_bits[b], __float_as_uint(m));
    }
}

// lengths: n_samples[b], or hop * (n_frames[b] - 1) when `hop` > 0 and `n_samples` holds frame counts
__global__ void peak_scale_kernel(float* y, int64_t y_stride, const int* n_samples,
                                  const unsigned* peak_bits, float eps, int hop) {
    const int b = blockIdx.y;
    const int64_t len = hop > 0 ? (int64_t)hop * (n_samples[b] - 1) : n_samples[b];
    const float denom = __uint_as_float(peak_bits[b]) + eps;
    for (int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; n < len;
         n += (int64_t)gridDim.x * blockDim.x) {
        float* p = y + (int64_t)b * y_stride + n;
        *p = *p / denom;
    }
}

// sc[b][k] = sqrt(sum_runs num / sum_runs den); partials are [k][run][2]
__global__ void sc_reduce_kernel(const float* partial, int n_runs, const int* run_off,
                                 const int* run_cnt, int batch, int n_iter, float* sc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch * n_iter) return;
    const int b = i / n_iter, k = i % n_iter;
    float num = 0.f, den = 0.f;
    const float* p = partial + ((int64_t)k * n_runs + run_off[b]) * 2;
    for (int r = 0; r < run_cnt[b]; ++r) {
        num += p[2 * r];
        den += p[2 * r + 1];
    }
    sc[(int64_t)b * n_iter + k] = sqrtf(num / den);
}

// C[row][col] = epilogue( sum_k load(X[row][k]) * W[col][k] ): an "NT" GEMM on the f32 MFMA
// (v_mfma_f32_32x32x2_f32: an exact fp32 fma chain).  Block = 4 waves, tile 64 rows x 64 cols,
// each wave one 32x32 accumulator; operands are staged through LDS in k-chunks of 32.
//   LOAD_DB    X holds normalised dB; it is converted to power while being staged:
//              10^(0.1*(x*std + mean)) with numpy's rounding sequence (mel inversion)
//   EPI_STORE  C = acc                     EPI_CLAMP  C = max(acc, 0)
//   EPI_SUB    C = acc - D                 EPI_STEP   C = max(D - step*acc, 0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kGemmKC = 32;
enum { EPI_STORE = 0, EPI_CLAMP = 1, EPI_SUB = 2, EPI_STEP = 3 };

template <bool LOAD_DB, int EPI>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const float* __restrict__ X,
                                                      const float* __restrict__ W, int64_t rows, int K,
                                                      int cols, float mean, float std_, float step,
                                                      const float* D, float* C) {
    __shared__ float sa[64][kGemmKC + 1];
    __shared__ float sb[64][kGemmKC + 1];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t row0 = (int64_t)blockIdx.y * 64;
    const int col0 = blockIdx.x * 64;
    const int wr = wave >> 1, wc = wave & 1;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int k0 = 0; k0 < K; k0 += kGemmKC) {
        for (int i = threadIdx.x; i < 64 * kGemmKC; i += 256) {
            const int r = i / kGemmKC, k = i % kGemmKC;
            float pa = 0.f, pb = 0.f;
            if (k0 + k < K) {
                if (row0 + r < rows) {
                    pa = X[(row0 + r) * K + k0 + k];
                    if (LOAD_DB) {
                        // same rounding sequence as numpy: (mel*std)+mean, 0.1*x, 10**e
                        const float x = __fadd_rn(__fmul_rn(pa, std_), mean);
                        pa = powf(10.0f, __fmul_rn(0.1f, x));
                    }
                }
                if (col0 + r < cols) pb = W[(int64_t)(col0 + r) * K + k0 + k];
            }
            sa[r][k] = pa;
            sb[r][k] = pb;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kGemmKC; k += 2) {
            const float av = sa[wr * 32 + (lane & 31)][k + (lane >> 5)];
            const float bv = sb[wc * 32 + (lane & 31)][k + (lane >> 5)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    const int col = col0 + wc * 32 + (lane & 31);
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int64_t row = row0 + wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        if (row < rows && col < cols) {
            const int64_t at = row * cols + col;
            float v = acc[reg];
            if (EPI == EPI_CLAMP) v = fmaxf(v, 0.f);
            if (EPI == EPI_SUB) v = v - D[at];
            if (EPI == EPI_STEP) v = fmaxf(D[at] - step * v, 0.f);
            C[at] = v;
        }
    }
}


// ---------------------------------------------------------------------------
// Non-negative least squares, all projected-gradient steps in ONE kernel.
//   X <- max(X - step * (X A^T - B) A, 0)      X: (rows, F) >= 0,  A: (M, F) mel basis,  B: (rows, M)
// A block takes 32 rows and keeps its X tile in REGISTERS for the whole solve, transposed and in
// the accumulator layout of v_mfma_f32_32x32x2_f32: wave w owns the 32-bin column tiles c = w,
// w + NW, ..., and X^T tile c is 16 registers per lane (bin 32c + (reg & 3) + 8 (reg >> 2) + 4 g,
// row l % 32; g = lane / 32).  Both products of a step then run on the matrix cores without
// moving X:
//   phase 1   R^T (M x 32)  = A (M x F) . X^T (F x 32): the accumulator registers of X^T ARE the
//             B operands (register s of lane group g is bin b0(s) + 4 g of the tile: the k-pairs
//             of the product are taken as (b0, b0 + 4), which the host-built A1 table matches);
//             every wave sums over its own bins, the partial R^T of the NW waves meet in LDS,
//             B is subtracted there;
//   phase 2   X^T tile -= step * A^T tile (32 x M) . R^T (M x 32): R^T from LDS, A2 table from L2.
// The mel basis is banded (a bin tile meets only a few mel filters), so each tile carries the
// range of mels it touches and both phases skip the all-zero blocks: the same sums, a quarter
// of the matrix instructions.  Rows stop on their own once ||X A^T - B|| <= tol ||B||.
//   A1[c][mt][s][lane] = A[32 mt + l % 32][32 c + b0(s) + 4 g]        (0 outside the matrix)
//   A2[c][j][lane]     = A[2 j + g][32 c + l % 32]
//   mel_lo / mel_hi[c]   mels [lo, hi) with a non-zero entry in bin tile c
// X0 (rows, F) in, X (rows, F) out, in place.
#ifndef GLV_NNLS_PIPE1
#define GLV_NNLS_PIPE1 1
#endif
#ifndef GLV_NNLS_PIPE2
#define GLV_NNLS_PIPE2 1
#endif
template <int NW, int TPW, int MT>
__global__ __launch_bounds__(64 * NW, (NW == 4 ? 2 : 1)) void nnls_fused_kernel(float* __restrict__ X, const float* __restrict__ Bmel,
                                                           const float* __restrict__ A1, const float* __restrict__ A2,
                                                           const int* __restrict__ mel_lo, const int* __restrict__ mel_hi,
                                                           int64_t rows, int F, int M, int CT, int n_steps, float step,
                                                           float tol, float* __restrict__ Xs, int accel) {
    constexpr int MP = 32 * MT;  // mels, padded to whole tiles
    extern __shared__ float nnls_smem[];
    float* part = nnls_smem;                 // [NW][MP][32] partial R^T of every wave
    float* Rl = part + NW * MP * 32;         // [MP][32] R^T = X A^T - B
    float* bnorm = Rl + MP * 32;             // [32] ||B||^2 per row
    float* rowstep = bnorm + 32;             // [32] step of every row, 0 once it has converged
    // "some row still moves", one flag per iteration modulo 3: the flag of iteration it + 1 is
    // cleared at the top of iteration it, two block-wide barriers after its last reader
    __shared__ int any_active[3];
    if (threadIdx.x < 3) any_active[threadIdx.x] = 0;
    // (wave-uniform values forced into scalar registers: the tile ranges steer branches)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, g = lane >> 5,
              i = lane & 31;
    const int64_t row0 = (int64_t)blockIdx.x * 32;
    const bool row_ok = row0 + i < rows;
    // B^T, the entries this thread subtracts in the reduction of every step
    constexpr int EPT = MP * 32 / (64 * NW);
    float bt[EPT];
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int e = threadIdx.x + q * 64 * NW, m = e >> 5, n = e & 31;
        bt[q] = (m < M && row0 + n < rows) ? Bmel[(row0 + n) * M + m] : 0.f;
    }
    if (threadIdx.x < 32) {
        float b2 = 0.f;
        if (row0 + threadIdx.x < rows)
            for (int m = 0; m < M; ++m) {
                const float bv = Bmel[(row0 + threadIdx.x) * M + m];
                b2 += bv * bv;
            }
        bnorm[threadIdx.x] = b2;
        rowstep[threadIdx.x] = step;
    }
    // this wave's tiles of X^T
    f32x16 xt[TPW];
    int lo[TPW], hi[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int c = wave + NW * t;
        lo[t] = __builtin_amdgcn_readfirstlane(c < CT ? mel_lo[c] : 0);
        hi[t] = __builtin_amdgcn_readfirstlane(c < CT ? mel_hi[c] : 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int bin = 32 * c + (r & 3) + 8 * (r >> 2) + 4 * g;
            xt[t][r] = (row_ok && bin < F) ? X[(row0 + i) * F + bin] : 0.f;
        }
    }
    // Accelerated form (accel): the registers hold the EXTRAPOLATED iterate y_k, the iterate
    // x_k itself lives in Xs, a scratch array in this kernel's own tile order
    // [block][tile][4][lane] x 16 bytes, which every step reads and rewrites once per tile (it
    // stays in the caches); beta_k = (t_k - 1) / t_{k+1} follows in double precision in every
    // thread.  Without accel beta is 0, y = x, and Xs is not touched.
    f4* xs = reinterpret_cast<f4*>(Xs) + (int64_t)blockIdx.x * CT * 256 + lane;
    auto xs_store = [&](int c, const f32x16& x) {
#pragma unroll
        for (int q = 0; q < 4; ++q) xs[(c * 4 + q) * 64] = f4{x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]};
    };
    if (accel) {
#pragma unroll
        for (int t = 0; t < TPW; ++t)
            if (wave + NW * t < CT) xs_store(wave + NW * t, xt[t]);
    }
    double tk = 1.0;
    __syncthreads();
    for (int it = 0; it < n_steps; ++it) {
        // ---- phase 1: this wave's share of R^T ----
        f32x16 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
        // (the table pointers are made opaque once per step: the operand loads are the same in
        // every step, and hoisted out of the step loop they would need a register each)
        const float* A1s = A1;
        const float* A2s = A2;
        asm volatile("" : "+s"(A1s), "+s"(A2s));
#if GLV_NNLS_PIPE1
        // The TPW * MT (tile, mel tile) blocks in a fixed order; a block is skipped when no filter
        // of the tile falls into the mel tile.  The operands of block n + 1 are requested before
        // the matrix instructions of block n are issued, so their way from L2 is hidden behind
        // 16 MFMAs; the fences keep the compiler from requesting all blocks at once.
        auto active = [&](int n) { return hi[n / MT] > 32 * (n % MT) && lo[n / MT] < 32 * (n % MT + 1); };
        auto request = [&](int n, float (&dst)[16]) {
            const float* base = A1s + ((wave + NW * (n / MT)) * MT + n % MT) * (16 * 64);
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) dst[s2] = base[lane + s2 * 64];
        };
        float av[2][16];
        if (active(0)) request(0, av[0]);
#pragma unroll
        for (int n = 0; n < TPW * MT; ++n) {
            if (n + 1 < TPW * MT && active(n + 1)) request(n + 1, av[(n + 1) & 1]);
            if (active(n)) {
#pragma unroll
                for (int s2 = 0; s2 < 16; ++s2)
                    acc[n % MT] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[n & 1][s2], xt[n / MT][s2], acc[n % MT], 0, 0, 0);
            }
            GLV_SCHED_FENCE();
        }
#else
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int c = wave + NW * t;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (hi[t] <= 32 * mt || lo[t] >= 32 * (mt + 1)) continue;  // no filter of this tile there
                const float* a1 = A1s + ((c * MT + mt) * 16) * 64 + lane;
                float av[16];
#pragma unroll
                for (int s2 = 0; s2 < 16; ++s2) av[s2] = a1[s2 * 64];
#pragma unroll
                for (int s2 = 0; s2 < 16; ++s2)
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s2], xt[t][s2], acc[mt], 0, 0, 0);
                GLV_SCHED_FENCE();
            }
        }
#endif
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                part[((wave * MP) + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * g) * 32 + i] = acc[mt][r];
        if (threadIdx.x == 0) any_active[(it + 1) % 3] = 0;
        __syncthreads();
        // ---- R^T = sum of the partials - B^T, in a fixed order ----
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int e = threadIdx.x + q * 64 * NW;
            float r = part[e];
#pragma unroll
            for (int w = 1; w < NW; ++w) r += part[w * MP * 32 + e];
            Rl[e] = r - bt[q];
        }
        __syncthreads();
        // a row stops for good once its residual passes the test: its step (and extrapolation
        // weight) is zero from then on, so this step leaves it at max(y, 0) and later ones there
        bool last_step = false;
        if (tol > 0.f && threadIdx.x < 32) {
            float r2 = 0.f;
            for (int m = 0; m < M; ++m) r2 += Rl[m * 32 + threadIdx.x] * Rl[m * 32 + threadIdx.x];
            const bool active = rowstep[threadIdx.x] != 0.f && r2 > tol * tol * bnorm[threadIdx.x];
            rowstep[threadIdx.x] = active ? step : 0.f;
            if (active) any_active[it % 3] = 1;
        }
        if (tol > 0.f) {
            __syncthreads();
            last_step = !any_active[it % 3];  // no row moves any more: this step only projects
        }
        // ---- phase 2: X^T tile -= step * A^T tile . R^T, clamp (and extrapolate) ----
        const float st = rowstep[i];
        const double tn = 0.5 * (1.0 + sqrt(1.0 + 4.0 * tk * tk));
        const float beta = (accel && st != 0.f) ? (float)((tk - 1.0) / tn) : 0.f;
        tk = tn;
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int c = wave + NW * t;
            if (hi[t] <= lo[t]) continue;
            if (last_step && !accel) continue;  // x = max(x, 0) changes nothing
            f32x16 gt;
#pragma unroll
            for (int r = 0; r < 16; ++r) gt[r] = 0.f;
            // x_k of this tile, requested before the products that hide its way from the cache
            f4 xo[4];
            if (accel) {
#pragma unroll
                for (int q = 0; q < 4; ++q) xo[q] = xs[(c * 4 + q) * 64];
            }
            const float* a2 = A2s + c * (MP / 2) * 64 + lane;
#if GLV_NNLS_PIPE2
            // four k-steps at a time; the operands of the next four are requested before the
            // matrix instructions of these four are issued (the table is zero outside the tile's
            // mels, so rounding the range outwards adds nothing)
            auto fetch = [&](int j, float (&av)[4], float (&bv)[4]) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    av[u] = a2[(j + u) * 64];
                    bv[u] = Rl[(2 * (j + u) + g) * 32 + i];
                }
            };
            const int jend = last_step ? 0 : (hi[t] + 1) >> 1;  // (every step is zero then: no products)
            float av[4], bv[4];
            fetch((lo[t] >> 1) & ~3, av, bv);
            for (int j = (lo[t] >> 1) & ~3; j < jend; j += 4) {
                float an[4], bn[4];
                fetch(j + 4 < MP / 2 ? j + 4 : j, an, bn);
#pragma unroll
                for (int u = 0; u < 4; ++u) gt = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], gt, 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    av[u] = an[u];
                    bv[u] = bn[u];
                }
            }
            GLV_SCHED_FENCE();
#else
            for (int j = (lo[t] >> 1) & ~3; j < (hi[t] + 1) >> 1; j += 4) {
                float av[4], bv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    av[u] = a2[(j + u) * 64];
                    bv[u] = Rl[(2 * (j + u) + g) * 32 + i];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) gt = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], gt, 0, 0, 0);
            }
            GLV_SCHED_FENCE();
#endif
            if (accel) {
                f32x16 xn;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float old = (r & 3) == 0 ? xo[r >> 2].x : (r & 3) == 1 ? xo[r >> 2].y
                                      : (r & 3) == 2 ? xo[r >> 2].z : xo[r >> 2].w;
                    xn[r] = fmaxf(xt[t][r] - st * gt[r], 0.f);
                    xt[t][r] = xn[r] + beta * (xn[r] - old);
                }
                xs_store(c, xn);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) xt[t][r] = fmaxf(xt[t][r] - st * gt[r], 0.f);
            }
        }
        if (last_step) break;
    }
    if (accel) {
        // the answer is x, not the extrapolated y: back from the scratch array (each thread reads
        // what it wrote itself)
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int c = wave + NW * t;
            if (c >= CT) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f4 v4 = xs[(c * 4 + q) * 64];
                xt[t][4 * q] = v4.x;
                xt[t][4 * q + 1] = v4.y;
                xt[t][4 * q + 2] = v4.z;
                xt[t][4 * q + 3] = v4.w;
            }
        }
    }
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int c = wave + NW * t;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int bin = 32 * c + (r & 3) + 8 * (r >> 2) + 4 * g;
            if (row_ok && c < CT && bin < F) X[(row0 + i) * F + bin] = xt[t][r];
        }
    }
}

// Mel inversion fused with the set-up of the iteration state: for a block of 32 frames the
// normalised-dB mel is turned into power once (LDS) and multiplied with the inverse basis on
// the f32 MFMA, 32 bins per wave and step (the same fma chain as gemm_nt_kernel<true,
// EPI_STORE>, so S is bit-identical).  The magnitudes of 512 bins x 32 frames are collected in
// LDS, then every wave writes whole native rows: 16-byte stores of S and of the initial
// angles phase * S / n_fft - what gemm_nt_kernel + init_native_kernel do through a (rows, F)
// array in HBM.
//   Wt  inverse basis transposed and zero-padded: [Mp][Fp], Mp a multiple of 16, Fp >= P + 32
constexpr int kMelWaves = 8;         // waves per block of mel_native_kernel
constexpr int kMelTileStride = 520;  // floats per frame of the LDS magnitude tile (512 + 8: the
                                     // accumulator rows 4 apart fall on different banks)
template <int A>
__global__ __launch_bounds__(64 * kMelWaves) void mel_native_kernel(ElemArgs ea, const float* __restrict__ mel_db,
                                                         const float* __restrict__ Wt, int M, int Mp, int Fp,
                                                         float mean, float std_, const cf* phase_nat,
                                                         uint64_t seed, float inv_n, float* S_native,
                                                         cf* spec, unsigned* peak_bits, int live_bins) {
    extern __shared__ float mel_smem[];
    // the per-utterance maxima the finalising kernel gathers start at zero (a memset of its
    // own would cost a gap in the stream)
    if (peak_bits && blockIdx.x == 0)
        for (int b = threadIdx.x; b < ea.batch; b += 64 * kMelWaves) peak_bits[b] = 0u;
    const int lds = Mp + 1;
    float* sa = mel_smem;                          // [32][Mp + 1] mel power of this block's frames
    float* tile = sa + 32 * lds;               // [32][kMelTileStride] magnitudes of one half
    float* nyq = tile + 32 * kMelTileStride;   // [32] magnitudes of the Nyquist bin
    const int P = ea.L.P, F = ea.L.F;
    constexpr int E = Geo<A>::E;
    const int nzh = ea.L.s_pair_nzh;
    const int64_t rows = (int64_t)ea.batch * ea.L.frames_stride;
    const int64_t row0 = (int64_t)blockIdx.x * 32;
    for (int i = threadIdx.x; i < 32 * Mp; i += 64 * kMelWaves) {
        const int r = i / Mp, k = i % Mp;
        float pa = 0.f;
        if (k < M && row0 + r < rows) {
            // same rounding sequence as numpy: (mel*std)+mean, 0.1*x, 10**e
            const float x = __fadd_rn(__fmul_rn(mel_db[(row0 + r) * M + k], std_), mean);
            pa = powf(10.0f, __fmul_rn(0.1f, x));
        }
        sa[r * lds + k] = pa;
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float* arow = sa + (lane & 31) * lds + (lane >> 5);
    auto product = [&](int col0) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        const float* bcol = Wt + (int64_t)(lane >> 5) * Fp + col0 + (lane & 31);
        // operands of 8 steps are fetched together so their latencies overlap (Mp is a multiple
        // of 16; the padding rows are zero and add nothing)
        for (int k0 = 0; k0 < Mp; k0 += 16) {
            float av[8], bv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                bv[i] = bcol[(int64_t)(k0 + 2 * i) * Fp];
                av[i] = arow[k0 + 2 * i];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[i], acc, 0, 0, 0);
        }
        return acc;
    };
    // frame (0..31) of accumulator register `reg` in this lane
    auto acc_frame = [&](int reg) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); };
    // bins from live_bins on have all-zero rows in the inverse basis: their products are exactly
    // zero and are not computed
    auto product_or_zero = [&](int col0) {
        if (col0 < live_bins) return product(col0);
        f32x16 z;
#pragma unroll
        for (int i = 0; i < 16; ++i) z[i] = 0.f;
        return z;
    };
    if (wave == kMelWaves - 1) {
        const f32x16 acc = product_or_zero(P);
        if ((lane & 31) == 0) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) nyq[acc_frame(reg)] = acc[reg];
        }
    }
    auto phase_at = [&](int64_t fr, int b, int t, int k) -> cf {
        if (phase_nat) return phase_nat[fr * F + k];
        return drawn_phase(ea.L, seed, b, t, k);
    };
    const int low = (lane >> 3) + 8 * (lane & 7);  // this lane's bin modulo 64 on the frequency side
    for (int h = 0; h < A; ++h) {
        for (int c = wave; c < 16; c += kMelWaves) {
            const f32x16 acc = product_or_zero(512 * h + 32 * c);
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) tile[acc_frame(reg) * kMelTileStride + 32 * c + (lane & 31)] = acc[reg];
        }
        __syncthreads();
        for (int f = wave * (32 / kMelWaves); f < (wave + 1) * (32 / kMelWaves); ++f) {
            const int64_t fr = row0 + f;
            if (fr >= rows) break;
            const int t = (int)(fr % ea.L.frames_stride), b = (int)(fr / ea.L.frames_stride);
            if (t >= (ea.L.frames_local > 0 ? ea.L.frames_local : ea.L.n_frames[b])) continue;
            float* snat = S_native + fr * (P + 4);
            f4* sp = reinterpret_cast<f4*>(spec + fr * P);
            const float* trow = tile + f * kMelTileStride + low;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                float s4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) s4[j] = trow[64 * (4 * g + j)];
                // native order keeps all registers; pair order only a lane's own low registers
                // (the first half of the registers), in the same place
                if (!nzh || 8 * h + 4 * g < E / 2)
                    reinterpret_cast<f4*>(snat)[(2 * h + g) * 64 + lane] = f4{s4[0], s4[1], s4[2], s4[3]};
                if (!spec) continue;  // the first inverse transform draws the phase itself (Engine::ph3_draw)
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    const int ka = 512 * h + low + 64 * (4 * g + 2 * pp);
                    const cf pa = phase_at(fr, b, t, ka), pb = phase_at(fr, b, t, ka + 64);
                    const float sa2 = s4[2 * pp], sb2 = s4[2 * pp + 1];
                    cf aa = cf{pa.x * sa2 * inv_n, pa.y * sa2 * inv_n};
                    const cf ab = cf{pb.x * sb2 * inv_n, pb.y * sb2 * inv_n};
                    if (ka == 0) {
                        // DC and Nyquist are real and share native bin 0 (c2r ignores their Im)
                        const cf pn = phase_at(fr, b, t, P);
                        aa.y = pn.x * nyq[f] * inv_n;
                    }
                    sp[(4 * h + 2 * g + pp) * 64 + lane] = f4{aa.x, aa.y, ab.x, ab.y};
                }
            }
            if (h == 0 && lane == 0) reinterpret_cast<f4*>(snat + P)[0] = f4{nyq[f], 0.f, 0.f, 0.f};
            if (nzh && h == A - 1) {
                // pair order: the magnitudes of the mirrors P - k of this lane's pairs i = NP - nzh
                // .. NP - 1 (all in this half of the spectrum), and the self-mirrored bin P/2
                const float* tr = tile + f * kMelTileStride;
                auto mirror_of = [&](int i) -> float {
                    if (low == 0) return i == 0 ? 0.f : tr[64 * (8 - i)];
                    return tr[(64 - low) + 64 * (7 - i)];
                };
                float* hi = snat + (E / 2) * 64;
                const int first = E / 2 - nzh;
                if (nzh % 4 == 0) {
                    for (int g = 0; g < nzh / 4; ++g)
                        reinterpret_cast<f4*>(hi)[g * 64 + lane] =
                            f4{mirror_of(first + 4 * g), mirror_of(first + 4 * g + 1), mirror_of(first + 4 * g + 2),
                               mirror_of(first + 4 * g + 3)};
                } else {
                    reinterpret_cast<cf*>(hi)[lane] = cf{mirror_of(first), mirror_of(first + 1)};
                }
                if (lane == 0) snat[P + 1] = tr[A == 1 ? 256 : 0];
            }
        }
        __syncthreads();
    }
}

// native spectrum -> natural-order power |D|^2, (rows, F) float32 (librosa's _spectrogram, power=2)
template <int A>
__global__ void spec_power_kernel(ElemArgs ea, const cf* spec, float* power) {
    const int P = ea.L.P, F = ea.L.F;
    const int64_t total = (int64_t)ea.batch * ea.L.frames_stride * P;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % P);
        const int64_t fr = i / P;
        const int t = (int)(fr % ea.L.frames_stride), b = (int)(fr / ea.L.frames_stride);
        const bool live = t < ea.L.n_frames[b];
        const int reg = q >> 6, lane = q & 63;
        const int k = Geo<A>::bin(reg, lane);
        const cf d = live ? spec[fr * P + native_spec_index(reg, lane)] : cf{0.f, 0.f};
        if (q == 0) {
            power[fr * F] = d.x * d.x;       // DC and Nyquist share native bin 0
            power[fr * F + P] = d.y * d.y;
        } else {
            power[fr * F + k] = d.x * d.x + d.y * d.y;
        }
    }
}

// librosa.power_to_db: 10*log10(max(amin, x)) - 10*log10(max(amin, ref)), floored at
// (its own maximum - top_db) when top_db >= 0.  ref is ref_dev[b] (per utterance) or ref_value.
__global__ void power_to_db_kernel(const float* x, int64_t stride, const int* n_valid, const float* ref_dev,
                                   float ref_value, float amin, float top_db, const float* max_dev,
                                   float* out) {
    const int b = blockIdx.y;
    const int64_t len = n_valid[b];
    const float ref = ref_dev ? ref_dev[b] : ref_value;
    const float ref_db = 10.0f * log10f(fmaxf(amin, ref));
    // the largest output is that of the largest input
    const float floor_db = 10.0f * log10f(fmaxf(amin, max_dev[b])) - ref_db - top_db;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < len;
         i += (int64_t)gridDim.x * blockDim.x) {
        float v = 10.0f * log10f(fmaxf(amin, x[b * stride + i])) - ref_db;
        if (top_db >= 0.f) v = fmaxf(v, floor_db);
        out[b * stride + i] = v;
    }
}

// x -> x^(1/power), in place (librosa.feature.inverse.mel_to_stft's last step)
__global__ void root_kernel(float* x, int64_t n, float inv_power) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        x[i] = (inv_power == 0.5f) ? sqrtf(x[i]) : powf(x[i], inv_power);
}

}  // namespace
