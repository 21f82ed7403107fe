// libglvoc: the plan (tables, workspace, run bookkeeping, graph cache) and the C ABI declared in
// include/glvoc.h.  The kernels are in kernels.h, the per-wavefront engine in engine.h and
// wave_fft.h.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/glvoc.h"
#include "elementwise.h"
#include "engine.h"
#include "host_tables.h"

using namespace glv;

// ---------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(GLV_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                               \
    } while (0)

extern "C" const char* glv_last_error(void) { return g_err; }
extern "C" int glv_version(void) { return 100; }

#include "kernels.h"

static int noalloc_warm_up_current_device();  // defined with glv_peak_normalize_noalloc


// ---------------------------------------------------------------------------
// plan
// ---------------------------------------------------------------------------
struct glv_plan {
    int n_fft, hop, win_length, n_mels, max_frames, max_batch, device;
    int A, HB, R, P, F, blocks;  // blocks = padded H-sample blocks per utterance row
    // options
    int frames_per_run = 0;  // 0 = choose from the batch and the device's wave slots
    int wave_slots = 2048;   // runs the device holds at once (CUs x resident blocks x waves)
    int run_target = 0;      // number of runs asked for when frames_per_run is 0 (0 = pick_run_target)
    int n_cus = 256, blocks_per_cu = 2;
    int use_graph = 1;
    int timing = 0;
    int traffic_only = 0;  // measurement aid: iteration launches move data but compute nothing
    int recompute = 0;     // iterate without the spectrum state in HBM (MODE_ITER_RC)
    int alias_utts = 0;    // measurement aid: all utterances share one utterance's memory
    int fuse_draw = 1;     // device-drawn phase made inside the first inverse transform (no stored spectrum)
    int zero_skip = 1;     // leave out the bins whose inverse-basis rows vanish (MODE_ITER_LIN_Z)
    int kmax_bin = 1 << 30;  // highest bin with a non-zero row in the inverse basis
    int cur_nzh = 0;       // pair-order parameter of the call in progress (0: native order)
    int linear = 1;        // momentum term formed in the time domain (MODE_ITER_LIN): no spectrum state
                           // in memory; 0 = in the frequency domain as librosa writes it (MODE_ITER)
    int single_frame = -1;  // one frame per wave: -1 = when the batch is too small to fill the device
                            // with runs, 0 = never, 1 = always
    bool cur_single = false;    // the run table in place is one run per frame
    // A batch too large for one iteration-spanning launch is cut into groups of consecutive
    // utterances, each with a run table of its own that fits such a launch (the tables lie one
    // after the other in d_runs): [r0, r1) of every group, empty when the batch is one piece.
    struct RunGroup {
        int r0, r1;
    };
    std::vector<RunGroup> groups;
    int64_t cur_group_frames = 0;  // the group size the table in place was cut for (0: one piece)
    int64_t group_frames = 0;      // option: frames per group (0 = automatic, -1 = never cut)
    int last_loop_groups = 1;
    // All iterations of a call in ONE launch (MODE_ITER_LIN_P / _Z_P) when the call's workgroups
    // are resident together: -1 = off for good (a launch timed out), 0 = off, 1 = on.
    int persistent = 1;         // (2: even when the launch cannot be resident as a whole - a test aid)
    int last_loop_launches = 0; // kernel launches the iteration loop of the last call took
    int persist_capacity[3] = {0, 0, 0};  // workgroups seen resident together, per kernel (run kernels with
                                          //   all bins / zero-skip, one frame per wave)
    int* d_progress = nullptr;         // status word (one stride) followed by the runs' progress words
    size_t progress_runs = 0;          // runs the progress words are laid out for
    int* h_status = nullptr;           // pinned, device-visible: raised by a wave whose wait ran out; sticky
    int* h_status_dev = nullptr;
    int deposit = 1;            // option: run boundaries inside a workgroup completed by the producer
    bool cur_deposit = false;   // the tail flags in place (and the call in progress) follow that convention
    float* d_parts = nullptr;   // 2 generations x R partial-sum buffers (one-frame-per-wave modes)
    size_t parts_cap = 0;       // floats
    int64_t parts_utt_stride = 0, parts_stride = 0;  // layout of the call in progress
    // device tables
    cf* d_tables = nullptr;
    float* d_w2 = nullptr;
    float* d_iwss = nullptr;
    float* d_iwss_head = nullptr;
    float* d_iwss_tail = nullptr;
    float* d_iwss_small = nullptr;
    float* d_zeros = nullptr;  // hop zeros followed by hop floats of write-only trash
    float* d_pinv = nullptr;
    float* d_pinv_t = nullptr;  // transposed, zero-padded to [Mp][Fp] (mel_native_kernel)
    int Mp = 0, Fp = 0;
    bool have_pinv = false;
    float* d_basis = nullptr;    // (n_mels, F) mel basis, NNLS mode
    float* d_basis_t = nullptr;  // (F, n_mels)
    float* d_resid = nullptr;    // (rows, n_mels) residual A X - B
    bool have_basis = false;
    // tables of the fused NNLS kernel (nnls_fused_kernel): the basis as MFMA operand streams of
    // the two products, and the mel range every 32-bin tile touches
    float* d_nnls_a1 = nullptr;
    float* d_nnls_a2 = nullptr;
    int* d_nnls_lo = nullptr;
    int* d_nnls_hi = nullptr;
    int nnls_mt = 0, nnls_ct = 0;  // mel tiles (0: shape not covered, unfused steps), bin tiles
    int nnls_fused = 1;            // option: 0 = one product per launch (the unfused form)
    int nnls_accel = 1;            // option: Nesterov / FISTA extrapolation of the fused solver's steps
    float* d_nnls_x = nullptr;     // the fused solver's iterate x_k in tile order (accelerated form)
    int64_t nnls_mfma_per_block_step = 0;
    // device workspace
    cf* d_spec = nullptr;
    float* d_S = nullptr;
    float* d_ym[3] = {nullptr, nullptr, nullptr};  // overlap-add generations (2 used unless "recompute")
    float* d_yt[3] = {nullptr, nullptr, nullptr};
    RunDesc* d_runs = nullptr;
    unsigned char* d_flags = nullptr;
    int* d_nframes = nullptr;
    int* d_nsamples = nullptr;
    int* d_run_off = nullptr;
    int* d_run_cnt = nullptr;
    unsigned* d_peak = nullptr;
    int64_t* d_utt_ids = nullptr;  // identities for the device-drawn phase (glv_plan_set_utterance_ids)
    bool have_ids = false;
    int n_ids = 0;  // identities uploaded by the last glv_plan_set_utterance_ids
    float* d_sc_partial = nullptr;
    size_t sc_partial_cap = 0;
    int64_t ws_bytes = 0;
    // Host mirrors of the run bookkeeping: pinned staging sets used in rotation.  A set is
    // rewritten only after the copies that read it have completed (its event), so a call with
    // new lengths does not have to drain the stream.
    struct Staging {
        RunDesc* runs = nullptr;
        unsigned char* flags = nullptr;
        int* ints = nullptr;  // nframes | run_off | run_cnt
        int64_t* ids = nullptr;
        hipEvent_t done = nullptr;
        bool in_flight = false;
    };
    static constexpr int kStagingSets = 3;
    Staging stage[kStagingSets];
    int stage_next = 0;
    size_t max_runs = 0;
    std::vector<int> cur_nframes;
    int cur_frames_stride = -1, cur_fpr = -1, cur_target = -1, n_runs = 0;
    // graph cache for the iteration loop: the most recently used shapes stay instantiated, so a
    // serving loop that alternates between a few batch shapes replays instead of re-capturing
    struct GraphKey {
        int n_runs, n_iter, with_sc, frames_stride, batch, single, nzh;
        float cmom;
        bool operator==(const GraphKey& o) const {
            return n_runs == o.n_runs && n_iter == o.n_iter && with_sc == o.with_sc &&
                   frames_stride == o.frames_stride && batch == o.batch && single == o.single &&
                   nzh == o.nzh && cmom == o.cmom;
        }
    };
    struct GraphEntry {
        GraphKey key;
        hipGraphExec_t exec;
        uint64_t last_use;
    };
    static constexpr size_t kGraphCache = 8;
    std::vector<GraphEntry> graphs;
    uint64_t graph_clock = 0;
    // time-shard state (glv_shard_*): frames [t0, t1) of a T-frame utterance live here
    bool shard_active = false;
    // the in-library halo exchange (glv_halo_attach): this process's place in the chain of
    // shards; null = the caller moves the halos
    glv_halo* halo = nullptr;  // not owned
    int halo_loopback = 0;     // test aid: both neighbours are this rank itself
    int shard_t0 = 0, shard_t1 = 0, shard_T = 0, shard_cur = 0, shard_k = 0;
    // glv_shard_run as one graph: the steps and swaps after the first swap, captured once per
    // (shard, iteration count, momentum) and replayed.  Form 2 launches the workgroups at the
    // shard's two edges apart from those between them, so that only the edge launch waits for the
    // swap and the swap only for the edge launch.
    struct ShardGraph {
        int t0, t1, T, n_runs, n_iter, cur, first, loopback, form;
        float momentum;
        const glv_halo* halo;
        int end_cur, end_k;
        hipGraphExec_t exec;
    };
    ShardGraph shard_graph = {};
    int shard_graph_form = -1;     // option "shard_graph": 0 = enqueue every step and swap, 1 = one
                                   // graph of steps and swaps in turn, 2 = one graph, edges apart,
                                   // -1 = 2 for shards of at least two workgroups per CU, else 1
    bool shard_graph_refused = false;  // a capture failed: this plan stays with the enqueued form
    int shard_last_form = 0;       // form the last glv_shard_run took
    hipStream_t shard_side = nullptr;  // capture streams and events of the graph's two branches
    hipStream_t shard_main = nullptr;
    hipEvent_t shard_ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    // timing
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool ev_valid = false;
    // kernel launchers for this shape
    hipError_t (*launch_engine)(int mode, bool sc, const IterArgs&, size_t grid, hipStream_t) = nullptr;
    int (*engine_blocks_per_cu)() = nullptr;
    int (*persist_blocks_per_cu)(int variant) = nullptr;
    void (*launch_init)(const ElemArgs&, const float*, const cf*, uint64_t, float, float*, cf*,
                        hipStream_t) = nullptr;
    void (*launch_convert)(bool to_native, const ElemArgs&, const cf*, float, cf*, hipStream_t) = nullptr;
    void (*launch_power)(const ElemArgs&, const cf*, float*, hipStream_t) = nullptr;
    hipError_t (*launch_mel_native)(const glv_plan*, const ElemArgs&, const float*, float, float, const cf*,
                                    uint64_t, float, cf*, unsigned*, hipStream_t) = nullptr;
};

namespace {

// LDS of one workgroup: tables, one exchange buffer per wave and (n_fft 1024) one set of shared
// boundary blocks per wave
template <int A, int HB>
constexpr size_t engine_lds_bytes() {
    using G = Geo<A>;
    return (size_t)(table_span<A, HB>() + kWavesPerBlock * G::XBUF) * sizeof(cf) +
           (A == 1 ? (size_t)(kWavesPerBlock + 1) * Engine<A, HB>::kShareFloats * sizeof(float) +
                         4 * kWavesPerBlock * sizeof(int) +  // flag words of the iteration-spanning kernels
                         Engine<A, HB>::kShareFloats * sizeof(float)  // ... and one set of boundary blocks
                   : 0);
}

template <int A, int HB>
hipError_t launch_engine_t(int mode, bool sc, const IterArgs& a, size_t grid, hipStream_t st) {
    const size_t lds = engine_lds_bytes<A, HB>();
    const dim3 g((unsigned)grid), b(64 * kWavesPerBlock);
    if (mode == MODE_ITER) {
        if (sc)
            hipLaunchKernelGGL((gl_engine_kernel<A, HB, MODE_ITER, true>), g, b, lds, st, a);
        else
            hipLaunchKernelGGL((gl_engine_kernel<A, HB, MODE_ITER, false>), g, b, lds, st, a);
    } else if (mode == MODE_ITER_RC) {
        if (sc)
            hipLaunchKernelGGL((gl_engine_kernel<A, HB, MODE_ITER_RC, true>), g, b, lds, st, a);
        else
            hipLaunchKernelGGL((gl_engine_kernel<A, HB, MODE_ITER_RC, false>), g, b, lds, st, a);
    } else if (mode == MODE_ITER_RC_FIRST) {
        if (sc)
            hipLaunchKernelGGL((gl_engine_kernel<A, HB, MODE_ITER_RC_FIRST, true>), g, b, lds, st, a);
        else
            hipLaunchKernelGGL((gl_engine_kernel<A, HB, MODE_ITER_RC_FIRST, false>), g, b, lds, st, a);
    } else if (mode == MODE_TRAFFIC) {
        hipLaunchKernelGGL((gl_engine_kernel<A, HB, MODE_TRAFFIC, false>), g, b, lds, st, a);
    } else if (mode == MODE_ISTFT) {
        hipLaunchKernelGGL((gl_engine_kernel<A, HB, MODE_ISTFT, false>), g, b, lds, st, a);
    } else if (mode == MODE_ITER1) {
        if (sc)
            hipLaunchKernelGGL((gl_engine_kernel<A, HB, MODE_ITER1, true>), g, b, lds, st, a);
        else
            hipLaunchKernelGGL((gl_engine_kernel<A, HB, MODE_ITER1, false>), g, b, lds, st, a);
    } else if (mode == MODE_ITER_LIN) {
        if (sc)
            hipLaunchKernelGGL((gl_engine_kernel<A, HB, MODE_ITER_LIN, true>), g, b, lds, st, a);
        else
            hipLaunchKernelGGL((gl_engine_kernel<A, HB, MODE_ITER_LIN, false>), g, b, lds, st, a);
    } else if (mode == MODE_ITER_LIN_Z) {
        if (sc)
            hipLaunchKernelGGL((gl_engine_kernel<A, HB, MODE_ITER_LIN_Z, true>), g, b, lds, st, a);
        else
            hipLaunchKernelGGL((gl_engine_kernel<A, HB, MODE_ITER_LIN_Z, false>), g, b, lds, st, a);
    } else if (mode == MODE_ITER1_P) {
        hipLaunchKernelGGL((gl_engine_kernel<A, HB, MODE_ITER1_P, false>), g, b, lds, st, a);
    } else if (mode == MODE_ITER_LIN_P) {
        hipLaunchKernelGGL((gl_engine_kernel<A, HB, MODE_ITER_LIN_P, false>), g, b, lds, st, a);

    } else if (mode == MODE_ITER_LIN_Z_P) {
        hipLaunchKernelGGL((gl_engine_kernel<A, HB, MODE_ITER_LIN_Z_P, false>), g, b, lds, st, a);
    } else if (mode == MODE_TRAFFIC_LIN) {
        hipLaunchKernelGGL((gl_engine_kernel<A, HB, MODE_TRAFFIC_LIN, false>), g, b, lds, st, a);
    } else if (mode == MODE_TRAFFIC_LIN_Z) {
        hipLaunchKernelGGL((gl_engine_kernel<A, HB, MODE_TRAFFIC_LIN_Z, false>), g, b, lds, st, a);
    } else if (mode == MODE_ISTFT1) {
        hipLaunchKernelGGL((gl_engine_kernel<A, HB, MODE_ISTFT1, false>), g, b, lds, st, a);
    } else {
        hipLaunchKernelGGL((gl_engine_kernel<A, HB, MODE_STFT, false>), g, b, lds, st, a);
    }
    return hipGetLastError();
}

template <int A, int HB>
int engine_blocks_per_cu_t() {
    const size_t lds = engine_lds_bytes<A, HB>();
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gl_engine_kernel<A, HB, MODE_ITER_LIN, false>,
                                                     64 * kWavesPerBlock, lds) != hipSuccess)
        n = 1;
    return n < 1 ? 1 : n;
}
// the same for the iteration-spanning kernels (variant 0: all bins, 1: zero-skip, 2: one frame
// per wave)
template <int A, int HB>
int persist_blocks_per_cu_t(int variant) {
    {
        const size_t lds = engine_lds_bytes<A, HB>();
        int n = 0;
        const hipError_t e =
            variant == 0
                ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gl_engine_kernel<A, HB, MODE_ITER_LIN_P, false>,
                                                               64 * kWavesPerBlock, lds)
            : variant == 1
                ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gl_engine_kernel<A, HB, MODE_ITER_LIN_Z_P, false>,
                                                               64 * kWavesPerBlock, lds)
                : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gl_engine_kernel<A, HB, MODE_ITER1_P, false>,
                                                               64 * kWavesPerBlock, lds);
        return e == hipSuccess && n > 0 ? n : 0;
    }
}

inline unsigned grid_for(int64_t total, int block) {
    int64_t g = (total + block - 1) / block;
    if (g > 8192) g = 8192;
    if (g < 1) g = 1;
    return (unsigned)g;
}

template <int A>
void launch_init_t(const ElemArgs& ea, const float* S_nat, const cf* phase, uint64_t seed, float inv_n,
                   float* S_native, cf* spec, hipStream_t st) {
    const int64_t total = (int64_t)ea.batch * ea.L.frames_stride * ea.L.P;
    hipLaunchKernelGGL((init_native_kernel<A>), dim3(grid_for(total, 256)), dim3(256), 0, st, ea, S_nat,
                       phase, seed, inv_n, S_native, spec);
}

template <int A>
void launch_convert_t(bool to_native, const ElemArgs& ea, const cf* src, float inv_n, cf* dst,
                      hipStream_t st) {
    const int64_t total = (int64_t)ea.batch * ea.L.frames_stride * ea.L.P;
    if (to_native)
        hipLaunchKernelGGL((spec_convert_kernel<A, true>), dim3(grid_for(total, 256)), dim3(256), 0, st,
                           ea, src, inv_n, dst);
    else
        hipLaunchKernelGGL((spec_convert_kernel<A, false>), dim3(grid_for(total, 256)), dim3(256), 0, st,
                           ea, src, inv_n, dst);
}

template <int A>
void launch_power_t(const ElemArgs& ea, const cf* spec, float* power, hipStream_t st) {
    const int64_t total = (int64_t)ea.batch * ea.L.frames_stride * ea.L.P;
    hipLaunchKernelGGL((spec_power_kernel<A>), dim3(grid_for(total, 256)), dim3(256), 0, st, ea, spec, power);
}

template <int A>
hipError_t launch_mel_native_t(const glv_plan* p, const ElemArgs& ea, const float* mel_db, float mean,
                               float std_, const cf* phase, uint64_t seed, float inv_n, cf* spec,
                               unsigned* peak_bits, hipStream_t st) {
    const int64_t rows = (int64_t)ea.batch * ea.L.frames_stride;
    const size_t lds = (size_t)(32 * (p->Mp + 1) + 32 * kMelTileStride + 32) * sizeof(float);
    hipLaunchKernelGGL((mel_native_kernel<A>), dim3((unsigned)((rows + 31) / 32)), dim3(64 * kMelWaves), lds, st, ea,
                       mel_db, p->d_pinv_t, p->n_mels, p->Mp, p->Fp, mean, std_, phase, seed, inv_n, p->d_S,
                       spec, peak_bits, (int)std::min<int64_t>((int64_t)p->kmax_bin + 1, p->F));
    return hipGetLastError();
}

template <int A, int HB>
void bind_launchers(glv_plan* p) {
    p->launch_mel_native = &launch_mel_native_t<A>;
    p->launch_power = &launch_power_t<A>;
    p->launch_engine = &launch_engine_t<A, HB>;
    p->engine_blocks_per_cu = &engine_blocks_per_cu_t<A, HB>;
    p->persist_blocks_per_cu = &persist_blocks_per_cu_t<A, HB>;
    p->launch_init = &launch_init_t<A>;
    p->launch_convert = &launch_convert_t<A>;
}

template <typename T>
hipError_t dmalloc(glv_plan* p, T** ptr, size_t count) {
    const size_t bytes = std::max<size_t>(count * sizeof(T), 16);
    hipError_t e = hipMalloc(reinterpret_cast<void**>(ptr), bytes);
    if (e == hipSuccess) p->ws_bytes += (int64_t)bytes;
    return e;
}

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

// kernel arguments baked into captured graphs went stale (an option or a buffer changed)
void drop_graphs(glv_plan* p) {
    for (auto& g : p->graphs)
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
    p->graphs.clear();
    if (p->shard_graph.exec) (void)hipGraphExecDestroy(p->shard_graph.exec);
    p->shard_graph.exec = nullptr;
}

void plan_free(glv_plan* p) {
    if (!p) return;
    DeviceGuard guard(p->device);
    drop_graphs(p);
    if (p->shard_side) (void)hipStreamDestroy(p->shard_side);
    if (p->shard_main) (void)hipStreamDestroy(p->shard_main);
    for (hipEvent_t e : p->shard_ev)
        if (e) (void)hipEventDestroy(e);
    if (p->ev0) (void)hipEventDestroy(p->ev0);
    if (p->ev1) (void)hipEventDestroy(p->ev1);
    if (p->h_status) (void)hipHostFree(p->h_status);
    void* dev[] = {p->d_progress, p->d_nnls_x, p->d_nnls_a1, p->d_nnls_a2, p->d_nnls_lo, p->d_nnls_hi, p->d_parts, p->d_pinv_t, p->d_iwss_small, p->d_zeros, p->d_basis, p->d_basis_t, p->d_resid, p->d_iwss_head, p->d_iwss_tail, p->d_tables, p->d_w2,    p->d_iwss,    p->d_pinv,     p->d_spec,    p->d_S,
                   p->d_ym[0],  p->d_ym[1], p->d_yt[0],   p->d_yt[1],    p->d_ym[2], p->d_yt[2], p->d_runs,    p->d_flags,
                   p->d_nframes, p->d_nsamples, p->d_run_off, p->d_run_cnt, p->d_peak, p->d_sc_partial, p->d_utt_ids};
    for (void* q : dev)
        if (q) (void)hipFree(q);
    for (auto& sg : p->stage) {
        if (sg.runs) (void)hipHostFree(sg.runs);
        if (sg.flags) (void)hipHostFree(sg.flags);
        if (sg.ints) (void)hipHostFree(sg.ints);
        if (sg.ids) (void)hipHostFree(sg.ids);
        if (sg.done) (void)hipEventDestroy(sg.done);
    }
    delete p;
}

ElemArgs elem_args(const glv_plan* p, int batch, int frames_stride) {
    ElemArgs ea;
    ea.L = LayoutArgs{p->n_fft, p->hop, p->F, p->P, p->R, frames_stride,
                      (int64_t)p->blocks * p->hop, p->d_nframes, 0, 0,
                      p->have_ids ? p->d_utt_ids : nullptr, 0, 0, 0};
    ea.batch = batch;
    return ea;
}

IterArgs iter_args(const glv_plan* p, int frames_stride) {
    IterArgs a;
    memset(&a, 0, sizeof(a));
    a.runs = p->d_runs;
    a.n_runs = p->n_runs;
    a.spec = p->d_spec;
    a.S = p->d_S;
    a.spec_utt_stride = (int64_t)frames_stride * p->P;
    a.s_utt_stride = (int64_t)frames_stride * (p->P + 4);
    a.y_utt_stride = (int64_t)p->blocks * p->hop;
    if (p->alias_utts) a.spec_utt_stride = a.s_utt_stride = a.y_utt_stride = 0;
    a.w2 = p->d_w2;
    a.iwss = p->d_iwss;
    a.iwss_head = p->d_iwss_head;
    a.iwss_tail = p->d_iwss_tail;
    a.iwss_small = p->d_iwss_small;
    a.zeros = p->d_zeros;
    a.trash = p->d_zeros + p->hop;
    a.tables = p->d_tables;
    a.inv_n = 1.0f / (float)p->n_fft;
    a.deposit = p->cur_deposit ? 1 : 0;
    return a;
}

void set_io(const glv_plan* p, IterArgs& a, int in, int out) {
    a.y_in = p->d_ym[in];
    a.y_in_tail = p->d_yt[in];
    a.y_out = p->d_ym[out];
    a.y_out_tail = p->d_yt[out];
}

// Next pinned staging set of the rotation.  The host waits only if the copies enqueued the
// last time this set was used (kStagingSets calls ago) have still not run.
int acquire_staging(glv_plan* p, glv_plan::Staging** out) {
    glv_plan::Staging* sg = &p->stage[p->stage_next];
    p->stage_next = (p->stage_next + 1) % glv_plan::kStagingSets;
    if (sg->in_flight) {
        HIP_TRY(hipEventSynchronize(sg->done));
        sg->in_flight = false;
    }
    *out = sg;
    return GLV_OK;
}
hipError_t release_staging(glv_plan::Staging* sg, hipStream_t st) {
    const hipError_t e = hipEventRecord(sg->done, st);
    sg->in_flight = e == hipSuccess;
    return e;
}

// Workspace that depends on how the plan is used, not only on its capacity: the partial-sum
// buffers of the one-frame-per-wave modes and the per-run partials of the convergence curve.
// glv_plan_create reserves what the automatic mode choice can ask for; growing later has to
// wait for the stream (the old buffer may be in use) and invalidates the captured graphs.
int ensure_parts(glv_plan* p, size_t need, hipStream_t st) {
    if (need <= p->parts_cap) return GLV_OK;
    if (p->d_parts) {
        HIP_TRY(hipStreamSynchronize(st));
        (void)hipFree(p->d_parts);
        p->ws_bytes -= (int64_t)(p->parts_cap * sizeof(float));
    }
    p->d_parts = nullptr;
    p->parts_cap = 0;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&p->d_parts), need * sizeof(float)));
    p->parts_cap = need;
    p->ws_bytes += (int64_t)(need * sizeof(float));
    drop_graphs(p);
    return GLV_OK;
}
int ensure_sc(glv_plan* p, size_t need, hipStream_t st) {
    if (need <= p->sc_partial_cap) return GLV_OK;
    if (p->d_sc_partial) {
        HIP_TRY(hipStreamSynchronize(st));
        (void)hipFree(p->d_sc_partial);
        p->ws_bytes -= (int64_t)(p->sc_partial_cap * sizeof(float));
    }
    p->d_sc_partial = nullptr;
    p->sc_partial_cap = 0;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&p->d_sc_partial), need * sizeof(float)));
    p->sc_partial_cap = need;
    p->ws_bytes += (int64_t)(need * sizeof(float));
    drop_graphs(p);
    return GLV_OK;
}
// runs a call of this shape can have at most, by mode
size_t runs_bound(const glv_plan* p, int64_t frames, bool single) {
    if (single) return (size_t)frames;
    if (p->frames_per_run > 0) return (size_t)(frames / std::max(1, std::min(p->frames_per_run, p->R - 1)) + 1);
    const int64_t rounds = std::max<int64_t>(1, (frames + (int64_t)p->wave_slots * 64 - 1) / ((int64_t)p->wave_slots * 64));
    return (size_t)(p->wave_slots * rounds + p->max_batch);
}

// (Re)builds the run table for these lengths and uploads it.  Cached: a call with
// the lengths of the previous call costs nothing.
int prepare_runs(glv_plan* p, const int* n_frames_host, int batch, int frames_stride, hipStream_t st,
                 bool single = false, bool deposit = false, int64_t group_frames = 0, int64_t group_runs = 0) {
    if (!n_frames_host) return fail(GLV_ERR_INVALID, "n_frames_host is null");
    if (batch < 1 || batch > p->max_batch)
        return fail(GLV_ERR_INVALID, "batch %d outside [1, %d]", batch, p->max_batch);
    if (frames_stride < 1 || frames_stride > p->max_frames)
        return fail(GLV_ERR_INVALID, "frames_stride %d outside [1, %d]", frames_stride, p->max_frames);
    for (int b = 0; b < batch; ++b)
        if (n_frames_host[b] < 2 || n_frames_host[b] > frames_stride)
            return fail(GLV_ERR_INVALID, "n_frames[%d] = %d outside [2, %d]", b, n_frames_host[b],
                        frames_stride);
    const bool same = (int)p->cur_nframes.size() == batch && p->cur_frames_stride == frames_stride &&
                      p->cur_fpr == p->frames_per_run && p->cur_target == p->run_target &&
                      p->cur_single == single && p->cur_deposit == deposit && p->cur_group_frames == group_frames &&
                      memcmp(p->cur_nframes.data(), n_frames_host, sizeof(int) * batch) == 0;
    if (same) return GLV_OK;
    std::vector<RunDesc> runs;
    std::vector<glv_plan::RunGroup> groups;
    std::vector<unsigned char> flags;
    if (group_frames > 0) {
        // groups of consecutive utterances, as few and as even as keeps each at about group_frames
        // frames or fewer, every one with the
        // run table a call of those utterances alone would get (at most group_runs runs); the tail
        // flags follow each group's own workgroups
        int64_t total = 0;
        for (int b = 0; b < batch; ++b) total += n_frames_host[b];
        const int n_groups = (int)std::max<int64_t>(1, (total + group_frames - 1) / group_frames);
        flags.assign((size_t)batch * p->blocks, 0);
        int b0 = 0;
        int64_t seen = 0;
        for (int g = 0; g < n_groups && b0 < batch; ++g) {
            // utterances up to this group's share of the frames (the last group takes the rest)
            const int64_t upto = g + 1 == n_groups ? total : (total * (g + 1)) / n_groups;
            int b1 = b0;
            int64_t mine = 0;
            while (b1 < batch && (b1 == b0 || g + 1 == n_groups || seen + mine + n_frames_host[b1] / 2 <= upto)) {
                mine += n_frames_host[b1];
                ++b1;
            }
            seen += mine;
            const int64_t target = std::min<int64_t>(
                group_runs, pick_run_target(mine, p->n_cus, kWavesPerBlock, p->blocks_per_cu, std::max(1, p->R - 1)));
            std::vector<RunDesc> part = p->frames_per_run > 0
                                            ? build_runs(n_frames_host + b0, b1 - b0, p->frames_per_run, p->R)
                                            : build_runs_target(n_frames_host + b0, b1 - b0, target, p->R);
            if ((int64_t)part.size() > group_runs) {
                // (a fixed run length that gives a group more runs than a launch holds: one piece)
                groups.clear();
                runs.clear();
                break;
            }
            for (RunDesc& r : part) r.utt += b0;
            const std::vector<unsigned char> f =
                build_tail_flags(part, batch, p->blocks, p->R, deposit ? kWavesPerBlock : 0);
            for (size_t i = (size_t)b0 * p->blocks; i < (size_t)b1 * p->blocks; ++i) flags[i] = f[i];
            groups.push_back(glv_plan::RunGroup{(int)runs.size(), (int)(runs.size() + part.size())});
            runs.insert(runs.end(), part.begin(), part.end());
            b0 = b1;
        }
    }
    if (!groups.empty()) {
        // (the groups' tables, one after the other)
    } else if (single) {
        for (int b = 0; b < batch; ++b)
            for (int t = 0; t < n_frames_host[b]; ++t) runs.push_back(RunDesc{b, t, t + 1, n_frames_host[b]});
    } else if (p->frames_per_run > 0) {
        runs = build_runs(n_frames_host, batch, p->frames_per_run, p->R);
    } else {
        // One wave per run.  Pick the run length so that the runs fill the device's wave
        // slots in a whole number of rounds: a handful of runs left over for an extra,
        // nearly empty round would cost a full round of time.
        int64_t total = 0;
        for (int b = 0; b < batch; ++b) total += n_frames_host[b];
        const int64_t slots = p->wave_slots;
        const int64_t rounds = std::max<int64_t>(1, (total + slots * 64 - 1) / (slots * 64));
        const int64_t target = p->run_target > 0 ? p->run_target
                               : rounds > 1      ? slots * rounds
                                                 : pick_run_target(total, p->n_cus, kWavesPerBlock,
                                                                   p->blocks_per_cu, std::max(1, p->R - 1));
        runs = build_runs_target(n_frames_host, batch, target, p->R);
    }
    if (runs.size() > p->max_runs) return fail(GLV_ERR_INVALID, "run table overflow");
    if (groups.size() < 2) {
        groups.clear();
        flags = build_tail_flags(runs, batch, p->blocks, p->R, deposit ? kWavesPerBlock : 0);
    }
    glv_plan::Staging* sg = nullptr;
    int rc = acquire_staging(p, &sg);
    if (rc != GLV_OK) return rc;
    memcpy(sg->runs, runs.data(), runs.size() * sizeof(RunDesc));
    memcpy(sg->flags, flags.data(), flags.size());
    int* nf = sg->ints;
    int* off = nf + p->max_batch;
    int* cnt = off + p->max_batch;
    for (int b = 0; b < batch; ++b) {
        nf[b] = n_frames_host[b];
        off[b] = 0;
        cnt[b] = 0;
    }
    for (size_t r = 0; r < runs.size(); ++r) {
        if (cnt[runs[r].utt] == 0) off[runs[r].utt] = (int)r;
        cnt[runs[r].utt]++;
    }
    HIP_TRY(hipMemcpyAsync(p->d_runs, sg->runs, runs.size() * sizeof(RunDesc), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(p->d_flags, sg->flags, flags.size(), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(p->d_nframes, nf, sizeof(int) * batch, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(p->d_run_off, off, sizeof(int) * batch, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(p->d_run_cnt, cnt, sizeof(int) * batch, hipMemcpyHostToDevice, st));
    HIP_TRY(release_staging(sg, st));
    p->cur_nframes.assign(n_frames_host, n_frames_host + batch);
    p->cur_frames_stride = frames_stride;
    p->cur_fpr = p->frames_per_run;
    p->cur_target = p->run_target;
    p->cur_single = single;
    p->cur_deposit = deposit;
    p->cur_group_frames = group_frames;
    p->groups = groups;
    p->n_runs = (int)runs.size();
    return GLV_OK;
}

size_t engine_grid(const glv_plan* p) { return (size_t)(p->n_runs + kWavesPerBlock - 1) / kWavesPerBlock; }

}  // namespace

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
namespace {
// How many workgroups of an iteration-spanning kernel (variant 0: all bins, 1: zero-skip, 2: one
// frame per wave) are resident together: launches of falling size in which
// every workgroup waits, bounded, to see all the others.  Synchronises the device.
hipError_t persist_census(glv_plan* p, int variant) {
    const int mode = variant == 0 ? MODE_ITER_LIN_P : variant == 1 ? MODE_ITER_LIN_Z_P : MODE_ITER1_P;
    hipError_t e = hipDeviceSynchronize();
    int per_cu = std::min(8, p->persist_blocks_per_cu(variant));
    for (; e == hipSuccess && per_cu > 0 && p->persist_capacity[variant] == 0; --per_cu) {
        const int grid = per_cu * p->n_cus;
        e = hipMemset(p->d_progress, 0, (size_t)2 * kProgressStride * sizeof(int));
        if (e != hipSuccess) break;
        IterArgs a = iter_args(p, 1);
        a.census = 1;
        a.n_iter = 0;
        a.progress = p->d_progress + kProgressStride;
        a.status = p->d_progress;
        a.host_status = p->h_status_dev;
        e = p->launch_engine(mode, false, a, (size_t)grid, nullptr);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        int seen[2] = {0, 0};
        if (e == hipSuccess) e = hipMemcpy(seen, p->d_progress + kProgressStride, sizeof(seen), hipMemcpyDeviceToHost);
        if (e == hipSuccess && seen[0] == grid && seen[1] == grid) p->persist_capacity[variant] = grid;
    }
    return e;
}
}  // namespace

extern "C" int glv_plan_create(glv_plan** out, int n_fft, int hop, int win_length, int n_mels,
                               int max_frames, int max_batch, int device) {
    if (!out) return fail(GLV_ERR_INVALID, "out is null");
    *out = nullptr;
    if (n_fft != 1024 && n_fft != 2048)
        return fail(GLV_ERR_UNSUPPORTED, "n_fft %d not supported (1024 or 2048)", n_fft);
    // the hop is 1, 2 or 4 rows of n_fft/8 samples (overlap factor 8, 4 or 2)
    if (hop != n_fft / 8 && hop != n_fft / 4 && hop != n_fft / 2)
        return fail(GLV_ERR_UNSUPPORTED, "hop %d not supported for n_fft %d (n_fft/8, n_fft/4 or n_fft/2)",
                    hop, n_fft);
    if (win_length < 2 || win_length > n_fft)
        return fail(GLV_ERR_INVALID, "win_length %d outside [2, n_fft]", win_length);
    if (n_mels < 1 || max_frames < 2 || max_batch < 1)
        return fail(GLV_ERR_INVALID, "n_mels/max_frames/max_batch must be positive (max_frames >= 2)");
    DeviceGuard guard(device);
    if (!guard.ok) return fail(GLV_ERR_HIP, "hipSetDevice(%d) failed", device);

    glv_plan* p = new glv_plan();
    p->n_fft = n_fft;
    p->hop = hop;
    p->win_length = win_length;
    p->n_mels = n_mels;
    p->max_frames = max_frames;
    p->max_batch = max_batch;
    p->device = device;
    p->A = n_fft / 1024;
    p->HB = hop / (n_fft / 8);
    p->R = n_fft / hop;
    p->P = n_fft / 2;
    p->F = p->P + 1;
    p->blocks = max_frames + p->R;

    HostTables tab;
    const int key = p->A * 10 + p->HB;
    switch (key) {
        case 11: tab = build_tables<1>(hop, win_length); bind_launchers<1, 1>(p); break;
        case 12: tab = build_tables<1>(hop, win_length); bind_launchers<1, 2>(p); break;
        case 14: tab = build_tables<1>(hop, win_length); bind_launchers<1, 4>(p); break;
        case 21: tab = build_tables<2>(hop, win_length); bind_launchers<2, 1>(p); break;
        case 22: tab = build_tables<2>(hop, win_length); bind_launchers<2, 2>(p); break;
        case 24: tab = build_tables<2>(hop, win_length); bind_launchers<2, 4>(p); break;
        default: delete p; return fail(GLV_ERR_UNSUPPORTED, "unsupported n_fft/hop pair");
    }

#define PLAN_TRY(expr)                                                                        \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            plan_free(p);                                                                     \
            return fail(GLV_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));          \
        }                                                                                     \
    } while (0)

    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) {
            p->n_cus = std::max(1, prop.multiProcessorCount);
            p->blocks_per_cu = p->engine_blocks_per_cu();
            p->wave_slots = p->n_cus * p->blocks_per_cu * kWavesPerBlock;
        }
    }
    const size_t frames = (size_t)max_batch * max_frames;
    const size_t ylen = (size_t)max_batch * p->blocks * hop;
    p->max_runs = frames;
    PLAN_TRY(dmalloc(p, &p->d_tables, tab.fft.size()));
    PLAN_TRY(dmalloc(p, &p->d_w2, tab.w2.size()));
    PLAN_TRY(dmalloc(p, &p->d_iwss, tab.iwss.size()));
    PLAN_TRY(dmalloc(p, &p->d_iwss_head, tab.iwss_head.size()));
    PLAN_TRY(dmalloc(p, &p->d_iwss_tail, tab.iwss_tail.size()));
    PLAN_TRY(dmalloc(p, &p->d_iwss_small, tab.iwss_small.size()));
    PLAN_TRY(dmalloc(p, &p->d_zeros, (size_t)2 * hop));
    PLAN_TRY(hipMemset(p->d_zeros, 0, sizeof(float) * 2 * hop));
    PLAN_TRY(hipMemcpy(p->d_iwss_small, tab.iwss_small.data(), tab.iwss_small.size() * sizeof(float),
                       hipMemcpyHostToDevice));
    PLAN_TRY(dmalloc(p, &p->d_pinv, (size_t)p->F * n_mels));
    p->Mp = (n_mels + 15) & ~15;
    p->Fp = (p->F + 63) & ~63;
    PLAN_TRY(dmalloc(p, &p->d_pinv_t, (size_t)p->Mp * p->Fp));
    PLAN_TRY(dmalloc(p, &p->d_basis, (size_t)p->F * n_mels));
    PLAN_TRY(dmalloc(p, &p->d_basis_t, (size_t)p->F * n_mels));
    PLAN_TRY(dmalloc(p, &p->d_resid, frames * n_mels));
    PLAN_TRY(dmalloc(p, &p->d_spec, frames * p->P));
    PLAN_TRY(dmalloc(p, &p->d_S, frames * (p->P + 4)));
    for (int i = 0; i < 3; ++i) {
        PLAN_TRY(dmalloc(p, &p->d_ym[i], ylen));
        PLAN_TRY(dmalloc(p, &p->d_yt[i], ylen));
    }
    PLAN_TRY(dmalloc(p, &p->d_runs, p->max_runs));
    PLAN_TRY(dmalloc(p, &p->d_flags, (size_t)max_batch * p->blocks));
    PLAN_TRY(dmalloc(p, &p->d_nframes, (size_t)max_batch));
    PLAN_TRY(dmalloc(p, &p->d_nsamples, (size_t)max_batch));
    PLAN_TRY(dmalloc(p, &p->d_run_off, (size_t)max_batch));
    PLAN_TRY(dmalloc(p, &p->d_run_cnt, (size_t)max_batch));
    PLAN_TRY(dmalloc(p, &p->d_peak, (size_t)max_batch));
    PLAN_TRY(dmalloc(p, &p->d_utt_ids, (size_t)max_batch));
    for (auto& sg : p->stage) {
        PLAN_TRY(hipHostMalloc(reinterpret_cast<void**>(&sg.runs), p->max_runs * sizeof(RunDesc)));
        PLAN_TRY(hipHostMalloc(reinterpret_cast<void**>(&sg.flags), (size_t)max_batch * p->blocks));
        PLAN_TRY(hipHostMalloc(reinterpret_cast<void**>(&sg.ints), sizeof(int) * 4 * max_batch));
        PLAN_TRY(hipHostMalloc(reinterpret_cast<void**>(&sg.ids), sizeof(int64_t) * max_batch));
        PLAN_TRY(hipEventCreateWithFlags(&sg.done, hipEventDisableTiming));
    }
    PLAN_TRY(hipMemcpy(p->d_tables, tab.fft.data(), tab.fft.size() * sizeof(cf), hipMemcpyHostToDevice));
    PLAN_TRY(hipMemcpy(p->d_w2, tab.w2.data(), tab.w2.size() * sizeof(float), hipMemcpyHostToDevice));
    PLAN_TRY(hipMemcpy(p->d_iwss, tab.iwss.data(), tab.iwss.size() * sizeof(float), hipMemcpyHostToDevice));
    PLAN_TRY(hipMemcpy(p->d_iwss_head, tab.iwss_head.data(), tab.iwss_head.size() * sizeof(float),
                       hipMemcpyHostToDevice));
    PLAN_TRY(hipMemcpy(p->d_iwss_tail, tab.iwss_tail.data(), tab.iwss_tail.size() * sizeof(float),
                       hipMemcpyHostToDevice));
    PLAN_TRY(hipEventCreate(&p->ev0));
    PLAN_TRY(hipEventCreate(&p->ev1));
    // iteration-spanning kernels: a status word the device can raise in host memory, progress
    // words for as many runs as the device can hold at once, and a census of how many of their
    // workgroups really are resident together (the occupancy query can overstate it)
    PLAN_TRY(hipHostMalloc(reinterpret_cast<void**>(&p->h_status), 64, hipHostMallocMapped));
    p->h_status[0] = 0;
    PLAN_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&p->h_status_dev), p->h_status, 0));
    if (const char* env = getenv("GLVOC_PERSISTENT")) p->persistent = atoi(env) != 0 ? 1 : 0;
    p->progress_runs = (size_t)p->n_cus * 8 * kWavesPerBlock;
    PLAN_TRY(dmalloc(p, &p->d_progress, (1 + p->progress_runs) * kProgressStride));
    PLAN_TRY(hipDeviceSynchronize());
    for (int variant = 0; variant < 3; ++variant) {
        // (one frame per wave at n_fft 2048: handing a frame's 8 KB to fourteen neighbours costs
        // more than the launch boundary it removes - 15.3 against 14.4 us per iteration for one
        // 865-frame utterance - so that form keeps its launch per iteration)
        if (variant == 2 && p->A != 1) continue;
        PLAN_TRY(persist_census(p, variant));
    }
#undef PLAN_TRY
    if (noalloc_warm_up_current_device() != GLV_OK) {
        plan_free(p);
        return GLV_ERR_HIP;
    }
    {
        // what the automatic choices can ask for: one frame per wave up to 4 x wave_slots padded
        // blocks, and a 64-iteration convergence curve over the automatic run table
        const int64_t pad_blocks = std::min<int64_t>((int64_t)max_batch * p->blocks, 4 * (int64_t)p->wave_slots);
        int rc = ensure_parts(p, (size_t)2 * p->R * pad_blocks * hop, nullptr);
        if (rc == GLV_OK)
            rc = ensure_sc(p, (size_t)64 * 2 * std::max(runs_bound(p, (int64_t)frames, false),
                                                        (size_t)std::min<int64_t>((int64_t)frames,
                                                                                 std::max<int64_t>(2 * (int64_t)p->wave_slots,
                                                                                                   (int64_t)(8 + 2 * p->A) * p->n_cus))),
                           nullptr);
        if (rc != GLV_OK) {
            plan_free(p);
            return rc;
        }
    }
    *out = p;
    return GLV_OK;
}

extern "C" void glv_plan_destroy(glv_plan* p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    (void)hipDeviceSynchronize();
    plan_free(p);
}

extern "C" int64_t glv_plan_workspace_bytes(const glv_plan* p) { return p ? p->ws_bytes : 0; }

extern "C" int glv_plan_set_option(glv_plan* p, const char* name, int value) {
    if (!p || !name) return fail(GLV_ERR_INVALID, "null plan or option name");
    if (!strcmp(name, "frames_per_run")) {
        if (value < 0) return fail(GLV_ERR_INVALID, "frames_per_run must be >= 0 (0 = automatic)");
        p->frames_per_run = value;
    } else if (!strcmp(name, "run_target")) {
        if (value < 0) return fail(GLV_ERR_INVALID, "run_target must be >= 0 (0 = automatic)");
        p->run_target = value;
    } else if (!strcmp(name, "linear")) {
        p->linear = value != 0;
        drop_graphs(p);
    } else if (!strcmp(name, "nnls_fused")) {
        p->nnls_fused = value != 0;
    } else if (!strcmp(name, "nnls_accel")) {
        p->nnls_accel = value != 0;
    } else if (!strcmp(name, "zero_skip")) {
        p->zero_skip = value != 0;
    } else if (!strcmp(name, "deposit")) {
        p->deposit = value != 0;
        drop_graphs(p);
    } else if (!strcmp(name, "persistent")) {
        if (value < 0 || value > 2) return fail(GLV_ERR_INVALID, "persistent must be 0, 1 or 2");
        if (p->persistent >= 0) p->persistent = value;
    } else if (!strcmp(name, "group_frames")) {
        if (value < -1) return fail(GLV_ERR_INVALID, "group_frames is -1 (never cut a batch into groups), 0 (automatic) or a number of frames");
        p->group_frames = value;
    } else if (!strcmp(name, "halo_loopback")) {
        p->halo_loopback = value != 0;
    } else if (!strcmp(name, "shard_graph")) {
        if (value < -1 || value > 2) return fail(GLV_ERR_INVALID, "shard_graph is -1 (automatic), 0, 1 or 2");
        p->shard_graph_form = value;
        p->shard_graph_refused = false;
    } else if (!strcmp(name, "fuse_draw")) {
        p->fuse_draw = value != 0;
    } else if (!strcmp(name, "single_frame")) {
        if (value < -1 || value > 1) return fail(GLV_ERR_INVALID, "single_frame must be -1 (automatic), 0 or 1");
        p->single_frame = value;
    } else if (!strcmp(name, "use_graph")) {
        p->use_graph = value != 0;
    } else if (!strcmp(name, "timing")) {
        p->timing = value != 0;
    } else if (!strcmp(name, "alias_utts")) {
        p->alias_utts = value != 0;
        drop_graphs(p);
    } else if (!strcmp(name, "recompute")) {
        p->recompute = value != 0;
        drop_graphs(p);
    } else if (!strcmp(name, "traffic_only")) {
        p->traffic_only = value != 0;
        drop_graphs(p);
    } else {
        return fail(GLV_ERR_INVALID, "unknown option '%s'", name);
    }
    return GLV_OK;
}

extern "C" int glv_plan_set_utterance_ids(glv_plan* p, const int64_t* ids_host, int batch, void* stream) {
    if (!p) return fail(GLV_ERR_INVALID, "null plan");
    if (!ids_host) {
        p->have_ids = false;
        p->n_ids = 0;
        return GLV_OK;
    }
    if (batch < 1 || batch > p->max_batch)
        return fail(GLV_ERR_INVALID, "batch %d outside [1, %d]", batch, p->max_batch);
    DeviceGuard guard(p->device);
    hipStream_t st = (hipStream_t)stream;
    glv_plan::Staging* sg = nullptr;
    int rc = acquire_staging(p, &sg);
    if (rc != GLV_OK) return rc;
    memcpy(sg->ids, ids_host, sizeof(int64_t) * batch);
    HIP_TRY(hipMemcpyAsync(p->d_utt_ids, sg->ids, sizeof(int64_t) * batch, hipMemcpyHostToDevice, st));
    HIP_TRY(release_staging(sg, st));
    p->have_ids = true;
    p->n_ids = batch;
    return GLV_OK;
}

extern "C" int glv_plan_drawn_phase(glv_plan* p, uint64_t seed, int batch, int frames_stride, int64_t frame_origin,
                                    float* phase_dev, void* stream) {
    if (!p || !phase_dev) return fail(GLV_ERR_INVALID, "null argument");
    if (batch < 1 || batch > p->max_batch || frames_stride < 1 || frame_origin < 0)
        return fail(GLV_ERR_INVALID, "batch %d outside [1, %d], or frames_stride / frame_origin not positive", batch,
                    p->max_batch);
    if (p->have_ids && batch > p->n_ids)
        return fail(GLV_ERR_INVALID, "batch %d but only %d utterance identities are set", batch, p->n_ids);
    DeviceGuard guard(p->device);
    ElemArgs ea = elem_args(p, batch, frames_stride);
    ea.L.frame_origin = frame_origin;
    const int64_t total = (int64_t)batch * frames_stride * p->F;
    hipLaunchKernelGGL(drawn_phase_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, ea, seed,
                       reinterpret_cast<cf*>(phase_dev));
    HIP_TRY(hipGetLastError());
    return GLV_OK;
}

extern "C" int glv_plan_reserve(glv_plan* p, int batch, int frames_stride, int n_iter, int single_frame,
                                void* stream) {
    if (!p) return fail(GLV_ERR_INVALID, "null plan");
    if (batch < 1 || batch > p->max_batch || frames_stride < 1 || frames_stride > p->max_frames || n_iter < 0)
        return fail(GLV_ERR_INVALID, "shape outside the plan's capacity");
    DeviceGuard guard(p->device);
    hipStream_t st = (hipStream_t)stream;
    const int64_t frames = (int64_t)batch * frames_stride;
    int rc = GLV_OK;
    if (single_frame)
        rc = ensure_parts(p, (size_t)2 * p->R * batch * (frames_stride + p->R) * p->hop, st);
    if (rc == GLV_OK && n_iter > 0)
        rc = ensure_sc(p, (size_t)n_iter * 2 * runs_bound(p, frames, single_frame != 0), st);
    return rc;
}

extern "C" int glv_plan_last_loop_ms(glv_plan* p, float* ms_out) {
    if (!p || !ms_out) return fail(GLV_ERR_INVALID, "null argument");
    if (!p->ev_valid) return fail(GLV_ERR_STATE, "no timed glv_griffinlim call yet (option 'timing')");
    DeviceGuard guard(p->device);
    HIP_TRY(hipEventSynchronize(p->ev1));
    HIP_TRY(hipEventElapsedTime(ms_out, p->ev0, p->ev1));
    return GLV_OK;
}

extern "C" int glv_plan_status(glv_plan* p) {
    if (!p) return fail(GLV_ERR_INVALID, "null plan");
    if (p->h_status && *(volatile int*)p->h_status != 0) {
        p->persistent = -1;
        return fail(GLV_ERR_STATE,
                    "an iteration-spanning launch waited in vain for a neighbouring workgroup (is the device shared "
                    "with other work?): the results of that call are void; later calls launch every iteration "
                    "separately");
    }
    return GLV_OK;
}

extern "C" int glv_plan_last_loop_launches(const glv_plan* p) { return p ? p->last_loop_launches : 0; }

extern "C" int glv_plan_persistent_capacity(const glv_plan* p, int variant) {
    return p && variant >= 0 && variant < 3 ? p->persist_capacity[variant] : 0;
}

extern "C" int glv_plan_set_inverse_basis(glv_plan* p, const float* pinv_host) {
    if (!p || !pinv_host) return fail(GLV_ERR_INVALID, "null argument");
    DeviceGuard guard(p->device);
    HIP_TRY(hipMemcpy(p->d_pinv, pinv_host, sizeof(float) * p->F * p->n_mels, hipMemcpyHostToDevice));
    std::vector<float> t((size_t)p->Mp * p->Fp, 0.f);
    for (int f = 0; f < p->F; ++f)
        for (int m = 0; m < p->n_mels; ++m) t[(size_t)m * p->Fp + f] = pinv_host[(size_t)f * p->n_mels + m];
    HIP_TRY(hipMemcpy(p->d_pinv_t, t.data(), sizeof(float) * t.size(), hipMemcpyHostToDevice));
    // A mel basis that ends below the Nyquist frequency has all-zero columns above its last
    // filter, its pseudo-inverse all-zero rows: the magnitudes of those bins are exactly zero
    // for every mel, and the iteration can leave them out.
    p->kmax_bin = 0;
    for (int f = 0; f < p->F; ++f)
        for (int m = 0; m < p->n_mels; ++m)
            if (pinv_host[(size_t)f * p->n_mels + m] != 0.f) p->kmax_bin = f;
    p->have_pinv = true;
    return GLV_OK;
}

extern "C" int glv_mel_invert(glv_plan* p, const float* mel_db_dev, int batch, int frames_stride,
                              float mean, float std_, float* S_dev, void* stream) {
    if (!p || !mel_db_dev || !S_dev) return fail(GLV_ERR_INVALID, "null argument");
    if (!p->have_pinv) return fail(GLV_ERR_STATE, "glv_plan_set_inverse_basis has not been called");
    if (batch < 1 || frames_stride < 1) return fail(GLV_ERR_INVALID, "batch and frames_stride must be >= 1");
    DeviceGuard guard(p->device);
    const int64_t rows = (int64_t)batch * frames_stride;
    const dim3 grid((p->F + 63) / 64, (unsigned)((rows + 63) / 64));
    hipLaunchKernelGGL((gemm_nt_kernel<true, EPI_STORE>), grid, dim3(256), 0, (hipStream_t)stream, mel_db_dev,
                       p->d_pinv, rows, p->n_mels, p->F, mean, std_, 0.f, (const float*)nullptr, S_dev);
    HIP_TRY(hipGetLastError());
    return GLV_OK;
}


extern "C" int glv_plan_set_mel_basis(glv_plan* p, const float* basis_host) {
    if (!p || !basis_host) return fail(GLV_ERR_INVALID, "null argument");
    DeviceGuard guard(p->device);
    std::vector<float> t((size_t)p->F * p->n_mels);
    for (int m = 0; m < p->n_mels; ++m)
        for (int f = 0; f < p->F; ++f) t[(size_t)f * p->n_mels + m] = basis_host[(size_t)m * p->F + f];
    HIP_TRY(hipMemcpy(p->d_basis, basis_host, sizeof(float) * p->F * p->n_mels, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(p->d_basis_t, t.data(), sizeof(float) * p->F * p->n_mels, hipMemcpyHostToDevice));
    {
        // operand streams of the fused solver, see nnls_fused_kernel
        const int M = p->n_mels, F = p->F, MT = (M + 31) / 32, CT = (F + 31) / 32, MP = 32 * MT;
        p->nnls_mt = MT <= 4 ? MT : 0;
        p->nnls_ct = CT;
        if (p->nnls_mt) {
            auto at = [&](int m, int f) -> float { return (m < M && f < F) ? basis_host[(size_t)m * F + f] : 0.f; };
            std::vector<float> a1((size_t)CT * MT * 16 * 64), a2((size_t)CT * (MP / 2) * 64);
            std::vector<int> lo(CT, 0), hi(CT, 0);
            int64_t mfma = 0;
            for (int c = 0; c < CT; ++c) {
                int l = MP, h = 0;
                for (int m = 0; m < M; ++m)
                    for (int f = 32 * c; f < std::min(F, 32 * c + 32); ++f)
                        if (basis_host[(size_t)m * F + f] != 0.f) {
                            l = std::min(l, m);
                            h = std::max(h, m + 1);
                        }
                if (h > l) {
                    lo[c] = l;
                    hi[c] = h;
                    for (int mt = 0; mt < MT; ++mt)
                        if (!(h <= 32 * mt || l >= 32 * (mt + 1))) mfma += 16;
                    mfma += (((h + 1) >> 1) - ((l >> 1) & ~3) + 3) / 4 * 4;
                }
                for (int lane = 0; lane < 64; ++lane) {
                    const int g = lane >> 5, i = lane & 31;
                    for (int mt = 0; mt < MT; ++mt)
                        for (int s2 = 0; s2 < 16; ++s2)
                            a1[(((size_t)c * MT + mt) * 16 + s2) * 64 + lane] =
                                at(32 * mt + i, 32 * c + (s2 & 3) + 8 * (s2 >> 2) + 4 * g);
                    for (int j = 0; j < MP / 2; ++j) a2[((size_t)c * (MP / 2) + j) * 64 + lane] = at(2 * j + g, 32 * c + i);
                }
            }
            p->nnls_mfma_per_block_step = mfma;
            for (void* q : {(void*)p->d_nnls_a1, (void*)p->d_nnls_a2, (void*)p->d_nnls_lo, (void*)p->d_nnls_hi,
                            (void*)p->d_nnls_x})
                if (q) (void)hipFree(q);
            p->d_nnls_a1 = p->d_nnls_a2 = nullptr;
            p->d_nnls_lo = p->d_nnls_hi = nullptr;
            p->d_nnls_x = nullptr;
            // x_k of every 32-frame block in tile order: [block][tile][4][64 lanes] x 16 bytes
            const size_t blocks32 = ((size_t)p->max_batch * p->max_frames + 31) / 32;
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&p->d_nnls_x), blocks32 * CT * 1024 * sizeof(float)));
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&p->d_nnls_a1), a1.size() * sizeof(float)));
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&p->d_nnls_a2), a2.size() * sizeof(float)));
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&p->d_nnls_lo), CT * sizeof(int)));
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&p->d_nnls_hi), CT * sizeof(int)));
            HIP_TRY(hipMemcpy(p->d_nnls_a1, a1.data(), a1.size() * sizeof(float), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(p->d_nnls_a2, a2.data(), a2.size() * sizeof(float), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(p->d_nnls_lo, lo.data(), CT * sizeof(int), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(p->d_nnls_hi, hi.data(), CT * sizeof(int), hipMemcpyHostToDevice));
        }
    }
    p->have_basis = true;
    return GLV_OK;
}

namespace {

template <int NW, int TPW>
hipError_t launch_nnls_fused(const glv_plan* p, float* X, const float* B, int64_t rows, int n_steps, float step,
                             float tol, hipStream_t st) {
    const int MP = 32 * p->nnls_mt;
    const size_t lds = ((size_t)(NW + 1) * MP * 32 + 64) * sizeof(float);
    const dim3 grid((unsigned)((rows + 31) / 32)), block(64 * NW);
#define GLV_NNLS_CASE(MT_)                                                                                      \
    case MT_:                                                                                                   \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nnls_fused_kernel<NW, TPW, MT_>),             \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                        \
        hipLaunchKernelGGL((nnls_fused_kernel<NW, TPW, MT_>), grid, block, lds, st, X, B, p->d_nnls_a1,         \
                           p->d_nnls_a2, p->d_nnls_lo, p->d_nnls_hi, rows, p->F, p->n_mels, p->nnls_ct, n_steps, \
                           step, tol, p->d_nnls_x, p->nnls_accel);                                              \
        break;
    switch (p->nnls_mt) {
        GLV_NNLS_CASE(1)
        GLV_NNLS_CASE(2)
        GLV_NNLS_CASE(3)
        GLV_NNLS_CASE(4)
        default: return hipErrorInvalidValue;
    }
#undef GLV_NNLS_CASE
    return hipGetLastError();
}

}  // namespace

extern "C" int glv_mel_invert_nnls(glv_plan* p, const float* mel_dev, int batch, int frames_stride,
                                   int n_steps, float step, float tol, float power, float* S_dev, void* stream) {
    if (!p || !mel_dev || !S_dev) return fail(GLV_ERR_INVALID, "null argument");
    if (!p->have_pinv || !p->have_basis)
        return fail(GLV_ERR_STATE, "glv_plan_set_inverse_basis and glv_plan_set_mel_basis must be called first");
    if (batch < 1 || frames_stride < 1 || (int64_t)batch * frames_stride > (int64_t)p->max_batch * p->max_frames)
        return fail(GLV_ERR_INVALID, "batch x frames_stride exceeds the plan's capacity");
    if (n_steps < 0 || !(step > 0.f) || !(power > 0.f) || !(tol >= 0.f))
        return fail(GLV_ERR_INVALID, "n_steps >= 0, step > 0, tol >= 0 and power > 0 are required");
    DeviceGuard guard(p->device);
    hipStream_t st = (hipStream_t)stream;
    const int64_t rows = (int64_t)batch * frames_stride;
    const int M = p->n_mels, F = p->F;
    const dim3 grid_f((F + 63) / 64, (unsigned)((rows + 63) / 64));
    const dim3 grid_m((M + 63) / 64, (unsigned)((rows + 63) / 64));
    // X0 = max(pinv @ B, 0): the deterministic start of librosa.util.nnls
    hipLaunchKernelGGL((gemm_nt_kernel<false, EPI_CLAMP>), grid_f, dim3(256), 0, st, mel_dev, p->d_pinv, rows, M, F,
                       0.f, 0.f, 0.f, (const float*)nullptr, S_dev);
    if (p->nnls_fused && p->nnls_mt && n_steps > 0) {
        // all steps in one launch, X resident in registers (nnls_fused_kernel)
        if (p->nnls_ct <= 20)
            HIP_TRY((launch_nnls_fused<4, 5>(p, S_dev, mel_dev, rows, n_steps, step, tol, st)));
        else if (p->nnls_ct <= 40)
            HIP_TRY((launch_nnls_fused<8, 5>(p, S_dev, mel_dev, rows, n_steps, step, tol, st)));
        else
            return fail(GLV_ERR_UNSUPPORTED, "n_fft too large for the fused solver");
    } else {
        for (int i = 0; i < n_steps; ++i) {
            // R = X A^T - B   (rows x M), then X = max(X - step * R A, 0)
            hipLaunchKernelGGL((gemm_nt_kernel<false, EPI_SUB>), grid_m, dim3(256), 0, st, S_dev, p->d_basis, rows, F,
                               M, 0.f, 0.f, 0.f, mel_dev, p->d_resid);
            hipLaunchKernelGGL((gemm_nt_kernel<false, EPI_STEP>), grid_f, dim3(256), 0, st, p->d_resid, p->d_basis_t,
                               rows, M, F, 0.f, 0.f, step, S_dev, S_dev);
        }
    }
    if (power != 1.0f)
        hipLaunchKernelGGL(root_kernel, dim3(grid_for(rows * F, 256)), dim3(256), 0, st, S_dev, rows * F,
                           1.0f / power);
    HIP_TRY(hipGetLastError());
    return GLV_OK;
}

namespace {

// Mirrors per lane whose magnitude can be non-zero (pair order, engine.h): all E/2 = 4A of them,
// or only the upper half when every bin from the first one of register 3E/4 on has an all-zero
// row in the inverse basis (and the magnitudes are this basis times a mel).
int pair_nzh(const glv_plan* p, bool magnitudes_from_basis) {
    const int E = 8 * p->A, q = E / 2 + E / 4;
    const int first_zero_bin = 64 * (q & 7) + 512 * (q >> 3);
    const bool quarter = magnitudes_from_basis && p->zero_skip && p->have_pinv && p->kmax_bin < first_zero_bin;
    return quarter ? E / 4 : E / 2;
}

// enqueues the n_iter iteration kernels (and nothing else) on st; *last receives the index
// of the overlap-add generation the final iteration wrote
// the loop rotates three signal generations (instead of two) in the variants that read the
// previous signal
// (the traffic-only stand-in of the time-domain-momentum iteration included; the older
// traffic-only kernel, "linear" off, moves the spectrum state between two generations)
bool lin_form(const glv_plan* p) { return p->linear && !p->recompute; }
bool three_generations(const glv_plan* p) { return lin_form(p) || (p->recompute && !p->traffic_only); }

// one-frame-per-wave modes: generation g of the R partial-sum buffers
void set_parts_io(const glv_plan* p, IterArgs& a, int in, int out) {
    a.y_utt_stride = p->parts_utt_stride;
    a.part_stride = p->parts_stride;
    a.part_in = p->d_parts + (size_t)in * p->R * p->parts_stride;
    a.part_out = p->d_parts + (size_t)out * p->R * p->parts_stride;
}

int enqueue_loop(glv_plan* p, int frames_stride, int n_iter, float cmom, bool with_sc, hipStream_t st,
                 int* last, bool single = false) {
    IterArgs a = iter_args(p, frames_stride);
    a.cmom = cmom;
    int cur = 0;  // the iSTFT of the initial angles wrote generation 0
    for (int k = 0; k < n_iter; ++k) {
        a.has_prev = k > 0;
        a.sc_partial = with_sc ? p->d_sc_partial + (size_t)k * p->n_runs * 2 : nullptr;
        if (single) {
            set_parts_io(p, a, cur, cur ^ 1);
            HIP_TRY(p->launch_engine(MODE_ITER1, with_sc, a, engine_grid(p), st));
            cur ^= 1;
        } else if (three_generations(p) && !p->recompute) {
            // momentum in the time domain: read this signal and the one before it, write the
            // next.  The first iteration has no older signal: its weight is zero and the
            // pointer only has to be readable.
            const int out = (cur + 1) % 3, prev = k == 0 ? cur : (cur + 2) % 3;
            set_io(p, a, cur, out);
            a.y_prev = p->d_ym[prev];
            a.y_prev_tail = p->d_yt[prev];
            const bool all_bins = p->cur_nzh == 4 * p->A;
            const int mode = p->traffic_only ? (all_bins ? MODE_TRAFFIC_LIN : MODE_TRAFFIC_LIN_Z)
                                             : (all_bins ? MODE_ITER_LIN : MODE_ITER_LIN_Z);
            HIP_TRY(p->launch_engine(mode, with_sc && !p->traffic_only, a, engine_grid(p), st));
            cur = out;
        } else if (p->recompute && !p->traffic_only) {
            // three generations rotate: read `cur` (and the one before it), write the next
            const int out = (cur + 1) % 3, prev = (cur + 2) % 3;
            set_io(p, a, cur, out);
            a.y_prev = p->d_ym[prev];
            a.y_prev_tail = p->d_yt[prev];
            HIP_TRY(p->launch_engine(k == 0 ? MODE_ITER_RC_FIRST : MODE_ITER_RC, with_sc, a, engine_grid(p), st));
            cur = out;
        } else {
            set_io(p, a, cur, cur ^ 1);
            HIP_TRY(p->launch_engine(p->traffic_only ? MODE_TRAFFIC : MODE_ITER, with_sc, a, engine_grid(p), st));
            cur ^= 1;
        }
    }
    if (last) *last = cur;
    return GLV_OK;
}

}  // namespace

namespace {

// where the target magnitudes of a Griffin-Lim call come from
struct MagnitudeSource {
    const float* S_dev = nullptr;     // (B, T, F) magnitudes, or
    const float* mel_db_dev = nullptr;  // (B, T, n_mels) normalised dB mel to invert on the way
    float mean = 0.f, std_ = 1.f;
};

// can this call's loop run as one iteration-spanning launch?  (variant: 0 all bins, 1 zero-skip,
// 2 one frame per wave)
bool persistent_fits(const glv_plan* p, bool lin_runs, bool with_sc, int n_iter, bool single = false) {
    if (single) {
        if (p->persistent <= 0 || with_sc || n_iter < 1 || (size_t)p->n_runs > p->progress_runs) return false;
        // (worth it only while every workgroup has a compute unit to itself: one 430-frame
        // utterance 6.3 -> 5.9 us per iteration, four of them 10.6 -> 14.7)
        return p->persist_capacity[2] > 0 &&
               (p->persistent == 2 || (int64_t)engine_grid(p) <= (int64_t)std::min(p->persist_capacity[2], p->n_cus));
    }
    if (p->persistent <= 0 || !lin_runs || with_sc || n_iter < 1) return false;
    if (p->traffic_only || p->alias_utts || p->recompute) return false;
    if (p->A == 1 && !p->cur_deposit) return false;  // (n_fft 1024: built on the producer-completed boundaries)
    const int variant = p->cur_nzh == 4 * p->A ? 0 : 1;
    if ((size_t)p->n_runs > p->progress_runs || p->persist_capacity[variant] <= 0) return false;
    return p->persistent == 2 || (int64_t)engine_grid(p) <= (int64_t)p->persist_capacity[variant];
}

// all n_iter iterations as ONE launch: progress words cleared, then the kernel
int enqueue_persistent(glv_plan* p, int frames_stride, int n_iter, float cmom, hipStream_t st, bool single = false,
                       int run_first = 0, int run_count = -1) {
    // (runs [run_first, run_first + run_count) of the table: one group of a batch cut into groups)
    if (run_count < 0) run_count = p->n_runs;
    HIP_TRY(hipMemsetAsync(p->d_progress, 0, (size_t)(1 + run_count) * kProgressStride * sizeof(int), st));
    IterArgs a = iter_args(p, frames_stride);
    a.runs = p->d_runs + run_first;
    a.n_runs = run_count;
    const size_t grid = (size_t)(run_count + kWavesPerBlock - 1) / kWavesPerBlock;
    a.cmom = cmom;
    a.n_iter = n_iter;
    if (single) {
        // one frame per wave: the kernel alternates between the two generations of partial sums
        set_parts_io(p, a, 0, 1);
        a.status = p->d_progress;
        a.progress = p->d_progress + kProgressStride;
        a.host_status = p->h_status_dev;
        HIP_TRY(p->launch_engine(MODE_ITER1_P, false, a, grid, st));
        return GLV_OK;
    }
    for (int g = 0; g < 3; ++g) {
        a.gen_main[g] = p->d_ym[g];
        a.gen_tail[g] = p->d_yt[g];
    }
    set_io(p, a, 0, 1);
    a.y_prev = p->d_ym[0];
    a.y_prev_tail = p->d_yt[0];
    a.status = p->d_progress;
    a.progress = p->d_progress + kProgressStride;
    a.host_status = p->h_status_dev;
    const bool all_bins = p->cur_nzh == 4 * p->A;
    HIP_TRY(p->launch_engine(all_bins ? MODE_ITER_LIN_P : MODE_ITER_LIN_Z_P, false, a, grid, st));
    return GLV_OK;
}

int griffinlim_impl(glv_plan* p, const MagnitudeSource& src, const float* phase0_dev, uint64_t seed,
                    const int* n_frames_host, int batch, int frames_stride, int n_iter, float momentum,
                    float* y_dev, int64_t y_stride, float* sc_curve_dev, unsigned* peak_bits, void* stream) {
    if (n_iter < 0) return fail(GLV_ERR_INVALID, "n_iter must be >= 0");
    if (momentum < 0.f) return fail(GLV_ERR_INVALID, "griffinlim() called with momentum=%g < 0", momentum);
    DeviceGuard guard(p->device);
    hipStream_t st = (hipStream_t)stream;
    if (!n_frames_host) return fail(GLV_ERR_INVALID, "n_frames_host is null");
    if (batch < 1 || batch > p->max_batch)
        return fail(GLV_ERR_INVALID, "batch %d outside [1, %d]", batch, p->max_batch);
    if (phase0_dev == nullptr && p->have_ids && batch > p->n_ids)
        return fail(GLV_ERR_INVALID, "batch %d but glv_plan_set_utterance_ids gave only %d identities", batch,
                    p->n_ids);
    // A batch of few frames cannot fill the device with runs of several frames: every frame then
    // gets a wave of its own (partial-sum buffers instead of a running sum).  Measured against the
    // best run table: faster up to about 10 frames per CU at n_fft 1024 and 12 at n_fft 2048 (6.2
    // against 10.8 us per launch for one utterance of 430 frames), slower beyond.  Not for layouts
    // so ragged that the padded buffers would dwarf the payload.
    int64_t total_frames = 0;
    for (int b = 0; b < batch; ++b) total_frames += std::max(0, n_frames_host[b]);
    const int64_t padded_blocks = (int64_t)batch * (frames_stride + p->R);
    const bool plain = !p->recompute && !p->traffic_only && !p->alias_utts;
    const bool single = plain && (p->single_frame == 1 ||
                                  (p->single_frame < 0 && p->frames_per_run == 0 &&
                                   total_frames <= (int64_t)(8 + 2 * p->A) * p->n_cus &&
                                   padded_blocks <= 4 * (int64_t)p->wave_slots));
    // Every kernel of a call must agree on who completes a run boundary: the producer does where
    // all of them can (n_fft 1024, the iteration that keeps no spectrum, no convergence curve).
    const bool deposit = p->deposit && p->A == 1 && kWavesPerBlock > 1 && !single && three_generations(p) &&
                         !p->recompute && !(sc_curve_dev != nullptr && n_iter > 0);
    // a wait of an earlier iteration-spanning launch ran out: that call's results were void
    // (glv_plan_status reports it); from now on the loop is launched one iteration at a time
    if (p->persistent > 0 && p->h_status && *(volatile int*)p->h_status != 0) p->persistent = -1;
    const bool with_sc = sc_curve_dev != nullptr && n_iter > 0;
    // The time-domain-momentum iteration keeps the magnitudes in pair order; with magnitudes that
    // come from this plan's inverse basis it also knows which bins are zero for good.
    const bool lin_runs = !single && three_generations(p) && !p->recompute;
    p->cur_nzh = lin_runs ? pair_nzh(p, src.mel_db_dev != nullptr) : 0;
    // A batch of more runs than one iteration-spanning launch holds is cut into groups of
    // consecutive utterances that each get such a launch - and so is one that would fit a single
    // launch but with more than 19 frames per wave slot: 1024 utterances of 430 frames in groups of
    // 10 / 14 / 16 / 18 / 20 / 24 frames per slot give 148.8 / 153.7 / 156.2 / 154.7 / 149.5 / 146.7 k
    // audio-s/s (beyond 19 a group's working set per iteration no longer fits the 256 MB of
    // memory-side cache), and a launch per iteration over everything does worse than any of them
    // (512 utterances: 139.7 k).  n_fft 2048, 865 frames: 64 / 128 utterances in
    // one launch 72.0 / 73.6 k, 128 as two groups of 64 72.7 k - there 54 frames per slot.  With a
    // fixed run length (option frames_per_run: an utterance's runs, and with them its bits, then do
    // not depend on who shares the call or the group) a group is as many utterances as nearly fill
    // a launch; 4096 utterances of N(560, 190) frames in runs of 32 / 24 / 18 / 14 frames: 138.6 /
    // 145.4 / 152.4 / 149.8 k in groups against 140.4 / 139.6 / 137.3 / 133.4 k in one piece.
    int64_t group_frames = 0, group_runs = 0;
    if (plain && lin_runs && !with_sc && n_iter > 0 && p->persistent > 0 && p->group_frames >= 0 &&
        p->run_target == 0 && (p->A != 1 || deposit)) {
        const int variant = p->cur_nzh == 4 * p->A ? 0 : 1;
        group_runs = (int64_t)p->persist_capacity[variant] * kWavesPerBlock;
        group_frames = p->group_frames > 0    ? p->group_frames
                       : p->frames_per_run > 0 ? (int64_t)p->frames_per_run * group_runs * 15 / 16
                                               : (p->A == 1 ? 19 : 60) * group_runs;
        // (up to an eighth over the limit two groups of half the size would be the slower choice:
        // 144 utterances 150.5 k in one piece against 148.8 k in two groups, 160: 148.3 against 151.1 k)
        if (group_runs <= 0 || (p->group_frames == 0 && p->frames_per_run == 0 && total_frames * 8 <= group_frames * 9) ||
            (total_frames + group_frames - 1) / group_frames < 2)
            group_frames = 0;
    }
    int rc = prepare_runs(p, n_frames_host, batch, frames_stride, st, single, deposit, group_frames, group_runs);
    if (rc != GLV_OK) return rc;
    const bool grouped = !p->groups.empty();
    if (single) {
        p->parts_utt_stride = (int64_t)(frames_stride + p->R) * p->hop;
        p->parts_stride = (int64_t)batch * p->parts_utt_stride;
        // beyond what the plan reserved (only when the mode is forced on a large batch) this
        // grows the buffer, which drains the stream: see glv_plan_reserve
        rc = ensure_parts(p, (size_t)2 * p->R * p->parts_stride, st);
        if (rc != GLV_OK) return rc;
    }
    int max_t = 0;
    for (int b = 0; b < batch; ++b) max_t = std::max(max_t, n_frames_host[b]);
    if (y_stride < (int64_t)p->hop * (max_t - 1))
        return fail(GLV_ERR_INVALID, "y_stride %lld shorter than hop*(n_frames-1) = %lld",
                    (long long)y_stride, (long long)p->hop * (max_t - 1));
    if (with_sc) {
        rc = ensure_sc(p, (size_t)n_iter * p->n_runs * 2, st);
        if (rc != GLV_OK) return rc;
    }
    ElemArgs ea = elem_args(p, batch, frames_stride);
    ea.L.s_pair_nzh = p->cur_nzh;
    const float inv_n = 1.0f / (float)p->n_fft;
    // With a device-drawn phase and an iteration that keeps no spectrum, the initial spectrum
    // S * phase is needed once, by the first inverse transform: that kernel draws the phase
    // itself and nothing is stored (8 P bytes per frame neither written nor read back).
    const bool fuse_draw = p->fuse_draw && lin_runs && phase0_dev == nullptr;
    cf* spec0 = fuse_draw ? nullptr : p->d_spec;
    if (src.S_dev) {
        p->launch_init(ea, src.S_dev, reinterpret_cast<const cf*>(phase0_dev), seed, inv_n, p->d_S, spec0, st);
        HIP_TRY(hipGetLastError());
    } else {
        HIP_TRY(p->launch_mel_native(p, ea, src.mel_db_dev, src.mean, src.std_,
                                     reinterpret_cast<const cf*>(phase0_dev), seed, inv_n, spec0, peak_bits, st));
    }

    // inverse = istft(angles) of the initial phase
    IterArgs a = iter_args(p, frames_stride);
    if (fuse_draw) {
        a.draw = 1;
        a.draw_nzh = p->cur_nzh;
        a.draw_seed = seed;
        a.draw_ids = ea.L.utt_ids;
        a.draw_origin = ea.L.frame_origin;
    }
    if (single) {
        set_parts_io(p, a, 1, 0);
        HIP_TRY(p->launch_engine(MODE_ISTFT1, false, a, engine_grid(p), st));
    } else if (grouped) {
        // (every group's runs form workgroups of their own: one launch per group)
        set_io(p, a, 1, 0);
        for (const glv_plan::RunGroup& g : p->groups) {
            a.runs = p->d_runs + g.r0;
            a.n_runs = g.r1 - g.r0;
            HIP_TRY(p->launch_engine(MODE_ISTFT, false, a, (size_t)(a.n_runs + kWavesPerBlock - 1) / kWavesPerBlock, st));
        }
        a.runs = p->d_runs;
        a.n_runs = p->n_runs;
    } else {
        set_io(p, a, 1, 0);
        HIP_TRY(p->launch_engine(MODE_ISTFT, false, a, engine_grid(p), st));
    }
    // the iteration launches do not look at these; cleared so that a captured graph holds no seed
    a.draw = a.draw_nzh = 0;
    a.draw_seed = 0;
    a.draw_ids = nullptr;
    a.draw_origin = 0;

    const float cmom = momentum / (1.0f + momentum);
    if (p->timing) HIP_TRY(hipEventRecord(p->ev0, st));
    p->last_loop_launches = n_iter;
    p->last_loop_groups = 1;
    if (grouped) {
        // one iteration-spanning launch per group, one after the other
        for (const glv_plan::RunGroup& g : p->groups) {
            rc = enqueue_persistent(p, frames_stride, n_iter, cmom, st, false, g.r0, g.r1 - g.r0);
            if (rc != GLV_OK) return rc;
        }
        p->last_loop_launches = 1;
        p->last_loop_groups = (int)p->groups.size();
    } else if (n_iter > 0 && persistent_fits(p, lin_runs, with_sc, n_iter, single)) {
        rc = enqueue_persistent(p, frames_stride, n_iter, cmom, st, single);
        if (rc != GLV_OK) return rc;
        p->last_loop_launches = 1;
    } else if (n_iter > 0) {
        if (p->use_graph) {
            const glv_plan::GraphKey key{p->n_runs, n_iter, with_sc ? 1 : 0, frames_stride, batch,
                                         single ? 1 : 0, p->cur_nzh, cmom};
            glv_plan::GraphEntry* hit = nullptr;
            for (auto& g : p->graphs)
                if (g.key == key) hit = &g;
            if (!hit) {
                if (p->graphs.size() >= glv_plan::kGraphCache) {
                    // evict the least recently used shape
                    size_t lru = 0;
                    for (size_t i = 1; i < p->graphs.size(); ++i)
                        if (p->graphs[i].last_use < p->graphs[lru].last_use) lru = i;
                    (void)hipGraphExecDestroy(p->graphs[lru].exec);
                    p->graphs.erase(p->graphs.begin() + lru);
                }
                hipGraphExec_t exec = nullptr;
                hipStream_t cap = nullptr;
                HIP_TRY(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
                hipGraph_t graph = nullptr;
                hipError_t e = hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal);
                int lrc = GLV_OK;
                if (e == hipSuccess) {
                    lrc = enqueue_loop(p, frames_stride, n_iter, cmom, with_sc, cap, nullptr, single);
                    e = hipStreamEndCapture(cap, &graph);
                }
                if (e == hipSuccess && lrc == GLV_OK) e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
                if (graph) (void)hipGraphDestroy(graph);
                (void)hipStreamDestroy(cap);
                if (lrc != GLV_OK) return lrc;
                if (e != hipSuccess)
                    return fail(GLV_ERR_HIP, "capturing the iteration loop failed: %s", hipGetErrorString(e));
                p->graphs.push_back(glv_plan::GraphEntry{key, exec, 0});
                hit = &p->graphs.back();
            }
            hit->last_use = ++p->graph_clock;
            HIP_TRY(hipGraphLaunch(hit->exec, st));
        } else {
            rc = enqueue_loop(p, frames_stride, n_iter, cmom, with_sc, st, nullptr, single);
            if (rc != GLV_OK) return rc;
        }
    }
    if (p->timing) {
        HIP_TRY(hipEventRecord(p->ev1, st));
        p->ev_valid = true;
    }
    const int last = three_generations(p) ? n_iter % 3 : (n_iter & 1);
    // with the peak gathered on the way every block ends in one atomic on its row's maximum:
    // about 2048 blocks in all then
    const unsigned fin_blocks =
        peak_bits ? std::min<unsigned>(grid_for(y_stride, 256), std::max(32, 2048 / batch)) : grid_for(y_stride, 256);
    if (single) {
        ea.L.y_utt_stride = p->parts_utt_stride;
        hipLaunchKernelGGL(finalize_parts_kernel, dim3(fin_blocks, batch), dim3(256), 0, st, ea,
                           p->d_parts + (size_t)(n_iter & 1) * p->R * p->parts_stride, p->parts_stride, p->d_iwss,
                           p->d_iwss_head, p->d_iwss_tail, p->d_iwss_small, y_dev, y_stride, peak_bits);
    } else if (y_stride % 4 == 0 && reinterpret_cast<uintptr_t>(y_dev) % 16 == 0) {
        int hop_shift = 0;
        while ((1 << hop_shift) < p->hop) ++hop_shift;
        const unsigned blocks4 = peak_bits ? std::min<unsigned>(grid_for(y_stride / 4, 256), std::max(32, 2048 / batch))
                                           : grid_for(y_stride / 4, 256);
        hipLaunchKernelGGL(finalize4_kernel, dim3(blocks4, batch), dim3(256), 0, st, ea, p->d_ym[last], p->d_yt[last],
                           p->d_flags, p->blocks, p->d_iwss, p->d_iwss_head, p->d_iwss_tail, p->d_iwss_small, y_dev,
                           y_stride, hop_shift, peak_bits);
    } else {
        hipLaunchKernelGGL(finalize_kernel, dim3(fin_blocks, batch), dim3(256), 0, st, ea,
                           p->d_ym[last], p->d_yt[last], p->d_flags, p->blocks, p->d_w2, p->d_iwss, p->d_iwss_head,
                           p->d_iwss_tail, p->d_iwss_small, y_dev, y_stride, (int64_t)0, peak_bits);
    }
    HIP_TRY(hipGetLastError());
    if (with_sc) {
        hipLaunchKernelGGL(sc_reduce_kernel, dim3((batch * n_iter + 127) / 128), dim3(128), 0, st,
                           p->d_sc_partial, p->n_runs, p->d_run_off, p->d_run_cnt, batch, n_iter,
                           sc_curve_dev);
        HIP_TRY(hipGetLastError());
    }
    return GLV_OK;
}

}  // namespace

extern "C" int glv_griffinlim(glv_plan* p, const float* S_dev, const float* phase0_dev, uint64_t seed,
                              const int* n_frames_host, int batch, int frames_stride, int n_iter,
                              float momentum, float* y_dev, int64_t y_stride, float* sc_curve_dev,
                              void* stream) {
    if (!p || !S_dev || !y_dev) return fail(GLV_ERR_INVALID, "null argument");
    MagnitudeSource src;
    src.S_dev = S_dev;
    return griffinlim_impl(p, src, phase0_dev, seed, n_frames_host, batch, frames_stride, n_iter, momentum,
                           y_dev, y_stride, sc_curve_dev, nullptr, stream);
}

extern "C" int glv_mel_to_audio(glv_plan* p, const float* mel_db_dev, float mean, float std_,
                                const float* phase0_dev, uint64_t seed, const int* n_frames_host, int batch,
                                int frames_stride, int n_iter, float momentum, float peak_eps, float* y_dev,
                                int64_t y_stride, float* sc_curve_dev, void* stream) {
    if (!p || !mel_db_dev || !y_dev || !n_frames_host) return fail(GLV_ERR_INVALID, "null argument");
    if (!p->have_pinv) return fail(GLV_ERR_STATE, "glv_plan_set_inverse_basis has not been called");
    if (batch < 1 || batch > p->max_batch) return fail(GLV_ERR_INVALID, "batch %d outside [1, %d]", batch, p->max_batch);
    DeviceGuard guard(p->device);
    hipStream_t st = (hipStream_t)stream;
    const bool normalise = peak_eps >= 0.f;
    MagnitudeSource src;
    src.mel_db_dev = mel_db_dev;
    src.mean = mean;
    src.std_ = std_;
    int rc = griffinlim_impl(p, src, phase0_dev, seed, n_frames_host, batch, frames_stride, n_iter, momentum,
                             y_dev, y_stride, sc_curve_dev, normalise ? p->d_peak : nullptr, stream);
    if (rc != GLV_OK || !normalise) return rc;
    // n_samples[b] = hop * (n_frames[b] - 1), from the frame counts already on the device
    int max_t = 0;
    for (int b = 0; b < batch; ++b) max_t = std::max(max_t, n_frames_host[b]);
    const int64_t max_len = std::max<int64_t>(1, (int64_t)p->hop * (max_t - 1));
    hipLaunchKernelGGL(peak_scale_kernel, dim3(std::min<unsigned>(grid_for(max_len, 256 * 8), 64), batch), dim3(256),
                       0, st, y_dev, y_stride, p->d_nframes, p->d_peak, peak_eps, p->hop);
    HIP_TRY(hipGetLastError());
    return GLV_OK;
}


// ---------------------------------------------------------------------------
// time sharding: one long utterance, frames [t0, t1) on this device
// ---------------------------------------------------------------------------
namespace {

// Engine arguments in GLOBAL frame / block coordinates: runs, frames and blocks are numbered over
// the whole utterance, and the engine subtracts the shard's origin where it addresses storage.
IterArgs shard_iter_args(const glv_plan* p) {
    IterArgs a = iter_args(p, p->max_frames);
    a.t_origin = p->shard_t0;
    return a;
}

int shard_check(const glv_plan* p) {
    if (!p) return fail(GLV_ERR_INVALID, "null plan");
    if (!p->shard_active) return fail(GLV_ERR_STATE, "glv_shard_begin has not been called");
    return GLV_OK;
}

}  // namespace

extern "C" int glv_shard_begin(glv_plan* p, const float* S_dev, const float* phase0_dev, uint64_t seed,
                               int frame_begin, int frames_local, int frames_global, void* stream) {
    if (!p || !S_dev) return fail(GLV_ERR_INVALID, "null argument");
    const int t0 = frame_begin, t1 = frame_begin + frames_local, T = frames_global;
    if (t0 < 0 || frames_local < 1 || t1 > T || T < 2)
        return fail(GLV_ERR_INVALID, "shard [%d, %d) does not lie inside [0, %d)", t0, t1, T);
    if (frames_local > p->max_frames)
        return fail(GLV_ERR_INVALID, "frames_local %d exceeds the plan's max_frames %d", frames_local,
                    p->max_frames);
    if (frames_local < 2 * (p->R - 1) && frames_local < T)
        return fail(GLV_ERR_INVALID, "a shard needs at least %d frames", 2 * (p->R - 1));
    DeviceGuard guard(p->device);
    hipStream_t st = (hipStream_t)stream;
    int fpr = p->frames_per_run;
    if (fpr <= 0) fpr = std::max(p->R - 1, (frames_local + p->wave_slots - 1) / p->wave_slots);
    std::vector<RunDesc> runs = build_runs(&frames_local, 1, fpr, p->R);
    for (RunDesc& r : runs) {
        r.t_begin += t0;
        r.t_end += t0;
        r.n_frames = T;
    }
    // blocks completed by the tail buffer, indexed from the shard's first block
    std::vector<unsigned char> flags((size_t)p->blocks, 0);
    auto mark = [&](int j) {
        if (j - t0 >= 0 && j - t0 < p->blocks) flags[j - t0] = 1;
    };
    // (inside the shard the producer completes the run boundaries it can, as in the batch path;
    // the shard's own edges always travel as two halves)
    const bool deposit = p->deposit && p->A == 1 && kWavesPerBlock > 1 && p->linear && !p->recompute && !p->traffic_only;
    for (size_t i = 0; i < runs.size(); ++i) {
        if (deposit && run_completes_its_tail(runs, i, kWavesPerBlock)) continue;
        for (int q = 0; q < p->R - 1; ++q) mark(runs[i].t_end + q);
    }
    if (t0 > 0)
        for (int q = 0; q < p->R - 1; ++q) mark(t0 + q);  // tails of the previous shard's last run
    glv_plan::Staging* sg = nullptr;
    int rc = acquire_staging(p, &sg);
    if (rc != GLV_OK) return rc;
    memcpy(sg->runs, runs.data(), runs.size() * sizeof(RunDesc));
    memcpy(sg->flags, flags.data(), flags.size());
    sg->ints[0] = T;
    HIP_TRY(hipMemcpyAsync(p->d_runs, sg->runs, runs.size() * sizeof(RunDesc), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(p->d_flags, sg->flags, flags.size(), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(p->d_nframes, sg->ints, sizeof(int), hipMemcpyHostToDevice, st));
    HIP_TRY(release_staging(sg, st));
    p->cur_nframes.clear();  // the batch path must rebuild its own tables afterwards
    p->groups.clear();
    p->cur_deposit = deposit;
    p->n_runs = (int)runs.size();
    p->shard_t0 = t0;
    p->shard_t1 = t1;
    p->shard_T = T;
    p->shard_cur = 0;
    p->shard_k = 0;
    p->shard_active = true;
    // halo regions nobody writes on the outermost shards must read as zero
    for (int i = 0; i < 3; ++i) {
        HIP_TRY(hipMemsetAsync(p->d_ym[i], 0, sizeof(float) * p->blocks * p->hop, st));
        HIP_TRY(hipMemsetAsync(p->d_yt[i], 0, sizeof(float) * p->blocks * p->hop, st));
    }
    ElemArgs ea = elem_args(p, 1, p->max_frames);
    ea.L.frames_local = frames_local;
    ea.L.frame_origin = t0;
    ea.L.utt_ids = nullptr;  // one utterance: its phase stream is keyed by the seed alone
    p->cur_nzh = p->linear ? 4 * p->A : 0;
    ea.L.s_pair_nzh = p->cur_nzh;
    p->launch_init(ea, S_dev, reinterpret_cast<const cf*>(phase0_dev), seed, 1.0f / (float)p->n_fft, p->d_S,
                   p->d_spec, st);
    HIP_TRY(hipGetLastError());
    IterArgs a = shard_iter_args(p);
    set_io(p, a, 1, 0);
    HIP_TRY(p->launch_engine(MODE_ISTFT, false, a, engine_grid(p), st));
    return GLV_OK;
}

namespace {

// One Griffin-Lim iteration over `grid` workgroups of the shard's run table, starting at
// workgroup wg_first and taking every (1 + wg_skip)-th; the plan's generation counters stay as
// they are (shard_advance moves them once all workgroups of the iteration are enqueued).
int shard_launch(glv_plan* p, float momentum, hipStream_t st, int wg_first, int wg_skip, size_t grid) {
    IterArgs a = shard_iter_args(p);
    a.cmom = momentum / (1.0f + momentum);
    a.has_prev = p->shard_k > 0;
    a.wg_first = wg_first;
    a.wg_skip = wg_skip;
    if (p->linear) {
        // momentum in the time domain: the halo swap has completed both older generations, so
        // the previous signal is as local as the current one
        const int cur = p->shard_cur, out = (cur + 1) % 3, prev = p->shard_k == 0 ? cur : (cur + 2) % 3;
        set_io(p, a, cur, out);
        a.y_prev = p->d_ym[prev];
        a.y_prev_tail = p->d_yt[prev];
        HIP_TRY(p->launch_engine(MODE_ITER_LIN, false, a, grid, st));
    } else {
        set_io(p, a, p->shard_cur, p->shard_cur ^ 1);
        HIP_TRY(p->launch_engine(MODE_ITER, false, a, grid, st));
    }
    return GLV_OK;
}

void shard_advance(glv_plan* p) {
    p->shard_cur = p->linear ? (p->shard_cur + 1) % 3 : p->shard_cur ^ 1;
    p->shard_k++;
}

}  // namespace

extern "C" int glv_shard_step(glv_plan* p, float momentum, void* stream) {
    int rc = shard_check(p);
    if (rc != GLV_OK) return rc;
    if (momentum < 0.f) return fail(GLV_ERR_INVALID, "griffinlim() called with momentum=%g < 0", momentum);
    DeviceGuard guard(p->device);
    rc = shard_launch(p, momentum, (hipStream_t)stream, 0, 0, engine_grid(p));
    if (rc == GLV_OK) shard_advance(p);
    return rc;
}

extern "C" int glv_shard_halo_samples(const glv_plan* p) { return p ? (p->R - 1) * p->hop : 0; }

extern "C" int glv_shard_halo_pack(glv_plan* p, float* to_prev_dev, float* to_next_dev, void* stream) {
    int rc = shard_check(p);
    if (rc != GLV_OK) return rc;
    DeviceGuard guard(p->device);
    hipStream_t st = (hipStream_t)stream;
    const size_t bytes = sizeof(float) * (p->R - 1) * p->hop;
    const int cur = p->shard_cur;
    // towards the previous shard: this shard's "head" partial sums of blocks [t0, t0+R-1)
    if (to_prev_dev && p->shard_t0 > 0)
        HIP_TRY(hipMemcpyAsync(to_prev_dev, p->d_ym[cur], bytes, hipMemcpyDeviceToDevice, st));
    // towards the next shard: this shard's "tail" partial sums of blocks [t1, t1+R-1)
    if (to_next_dev && p->shard_t1 < p->shard_T)
        HIP_TRY(hipMemcpyAsync(to_next_dev, p->d_yt[cur] + (int64_t)(p->shard_t1 - p->shard_t0) * p->hop, bytes,
                               hipMemcpyDeviceToDevice, st));
    return GLV_OK;
}

extern "C" int glv_shard_halo_unpack(glv_plan* p, const float* from_prev_dev, const float* from_next_dev,
                                     void* stream) {
    int rc = shard_check(p);
    if (rc != GLV_OK) return rc;
    DeviceGuard guard(p->device);
    hipStream_t st = (hipStream_t)stream;
    const size_t bytes = sizeof(float) * (p->R - 1) * p->hop;
    const int cur = p->shard_cur;
    if (from_prev_dev && p->shard_t0 > 0)
        HIP_TRY(hipMemcpyAsync(p->d_yt[cur], from_prev_dev, bytes, hipMemcpyDeviceToDevice, st));
    if (from_next_dev && p->shard_t1 < p->shard_T)
        HIP_TRY(hipMemcpyAsync(p->d_ym[cur] + (int64_t)(p->shard_t1 - p->shard_t0) * p->hop, from_next_dev, bytes,
                               hipMemcpyDeviceToDevice, st));
    return GLV_OK;
}

extern "C" int glv_shard_halo_views(glv_plan* p, float** send_prev, float** send_next, float** recv_prev,
                                    float** recv_next) {
    int rc = shard_check(p);
    if (rc != GLV_OK) return rc;
    if (!send_prev || !send_next || !recv_prev || !recv_next) return fail(GLV_ERR_INVALID, "null argument");
    // The halos are contiguous pieces of the newest overlap-add generation: this shard's head
    // blocks [t0, t0+R-1) and tail blocks [t1, t1+R-1) go out, the neighbours' tails and heads
    // of the same blocks come in beside them.
    const int cur = p->shard_cur;
    const int64_t end = (int64_t)(p->shard_t1 - p->shard_t0) * p->hop;
    *send_prev = p->shard_t0 > 0 ? p->d_ym[cur] : nullptr;
    *recv_prev = p->shard_t0 > 0 ? p->d_yt[cur] : nullptr;
    *send_next = p->shard_t1 < p->shard_T ? p->d_yt[cur] + end : nullptr;
    *recv_next = p->shard_t1 < p->shard_T ? p->d_ym[cur] + end : nullptr;
    return GLV_OK;
}

extern "C" int glv_shard_output_range(const glv_plan* p, int64_t* n_begin, int64_t* n_count) {
    int rc = shard_check(p);
    if (rc != GLV_OK) return rc;
    if (!n_begin || !n_count) return fail(GLV_ERR_INVALID, "null argument");
    // blocks [t0, t1) belong to this shard; the signal proper starts at block R/2, and the
    // last shard also owns the blocks past the last frame
    const int64_t half = p->R / 2;
    const int64_t b0 = std::max<int64_t>(p->shard_t0, half) - half;
    const int64_t b1 = (p->shard_t1 < p->shard_T) ? std::max<int64_t>(p->shard_t1, half) - half
                                                  : (int64_t)p->shard_T - 1;
    *n_begin = b0 * p->hop;
    *n_count = std::max<int64_t>(0, b1 - b0) * p->hop;
    return GLV_OK;
}

extern "C" int glv_shard_end(glv_plan* p, float* y_dev, void* stream) {
    int rc = shard_check(p);
    if (rc != GLV_OK) return rc;
    if (!y_dev) return fail(GLV_ERR_INVALID, "null argument");
    DeviceGuard guard(p->device);
    int64_t n_begin = 0, n_count = 0;
    glv_shard_output_range(p, &n_begin, &n_count);
    if (n_count > 0) {
        ElemArgs ea = elem_args(p, 1, p->max_frames);
        ea.L.block_origin = p->shard_t0;
        ea.L.out_origin = n_begin;
        const int cur = p->shard_cur;
        hipLaunchKernelGGL(finalize_kernel, dim3(grid_for(n_count, 256), 1), dim3(256), 0, (hipStream_t)stream,
                           ea, p->d_ym[cur], p->d_yt[cur], p->d_flags, p->blocks,
                           p->d_w2, p->d_iwss, p->d_iwss_head, p->d_iwss_tail, p->d_iwss_small, y_dev,
                           n_count, n_begin, (unsigned*)nullptr);
        HIP_TRY(hipGetLastError());
    }
    p->shard_active = false;
    return GLV_OK;
}

// ---------------------------------------------------------------------------
// Halo exchange inside the library: RCCL point-to-point on the caller's stream.
// RCCL is looked up at run time (the library has no link-time dependency on it): a process
// that already holds a copy (PyTorch ships one) gets that one, any other the system's.
// ---------------------------------------------------------------------------
namespace {

struct RcclId {
    char internal[128];  // ncclUniqueId
};
}  // namespace
struct glv_halo {
    void* comm;  // ncclComm_t
    int rank, world, device;
};
namespace {
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(RcclId*) = nullptr;
    int (*CommInitRank)(void**, int, RcclId, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
    std::string why;  // why it is not usable (kept: dlerror() clears itself once read)
};
constexpr int kRcclFloat32 = 7;  // ncclFloat32

RcclApi& rccl_state() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (api.lib) break;
            const char* err = dlerror();
            api.why = err ? err : "librccl.so.1 cannot be loaded";
        }
        if (!api.lib) return;
        auto sym = [&](const char* n) { return dlsym(api.lib, n); };
        api.GetUniqueId = reinterpret_cast<int (*)(RcclId*)>(sym("ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<int (*)(void**, int, RcclId, int)>(sym("ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<int (*)(void*)>(sym("ncclCommDestroy"));
        api.Send = reinterpret_cast<int (*)(const void*, size_t, int, int, void*, hipStream_t)>(sym("ncclSend"));
        api.Recv = reinterpret_cast<int (*)(void*, size_t, int, int, void*, hipStream_t)>(sym("ncclRecv"));
        api.GroupStart = reinterpret_cast<int (*)()>(sym("ncclGroupStart"));
        api.GroupEnd = reinterpret_cast<int (*)()>(sym("ncclGroupEnd"));
        api.GetErrorString = reinterpret_cast<const char* (*)(int)>(sym("ncclGetErrorString"));
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.Send && api.Recv &&
                 api.GroupStart && api.GroupEnd && api.GetErrorString;
        if (!api.ok) api.why = "librccl.so.1 lacks one of the ncclSend / ncclRecv / ncclGroup* entry points";
    });
    return api;
}
RcclApi* rccl_api() {
    RcclApi& api = rccl_state();
    return api.ok ? &api : nullptr;
}
// the reason rccl_api() returned null
const char* rccl_why() { return rccl_state().why.c_str(); }

#define RCCL_TRY(api, expr)                                                                    \
    do {                                                                                       \
        const int r_ = (expr);                                                                 \
        if (r_ != 0) return fail(GLV_ERR_HIP, "%s failed: %s", #expr, (api)->GetErrorString(r_)); \
    } while (0)

}  // namespace

extern "C" int glv_halo_unique_id(void* id_out) {
    if (!id_out) return fail(GLV_ERR_INVALID, "null argument");
    RcclApi* api = rccl_api();
    if (!api) return fail(GLV_ERR_UNSUPPORTED, "RCCL is not usable: %s", rccl_why());
    RcclId id;
    RCCL_TRY(api, api->GetUniqueId(&id));
    memcpy(id_out, id.internal, GLV_HALO_ID_BYTES);
    return GLV_OK;
}

extern "C" int glv_halo_create(glv_halo** out, const void* id, int rank, int world, int device) {
    if (!out || !id) return fail(GLV_ERR_INVALID, "null argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world)
        return fail(GLV_ERR_INVALID, "rank %d outside [0, %d)", rank, world);
    RcclApi* api = rccl_api();
    if (!api) return fail(GLV_ERR_UNSUPPORTED, "RCCL is not usable: %s", rccl_why());
    DeviceGuard guard(device);
    if (!guard.ok) return fail(GLV_ERR_HIP, "hipSetDevice(%d) failed", device);
    RcclId uid;
    memcpy(uid.internal, id, GLV_HALO_ID_BYTES);
    void* comm = nullptr;
    RCCL_TRY(api, api->CommInitRank(&comm, world, uid, rank));
    *out = new glv_halo{comm, rank, world, device};
    return GLV_OK;
}

extern "C" int glv_halo_destroy(glv_halo* h) {
    if (!h) return GLV_OK;
    RcclApi* api = rccl_api();
    DeviceGuard guard(h->device);
    void* comm = h->comm;
    delete h;
    if (api && comm) RCCL_TRY(api, api->CommDestroy(comm));
    return GLV_OK;
}

extern "C" int glv_halo_attach(glv_plan* p, glv_halo* h) {
    if (!p) return fail(GLV_ERR_INVALID, "null plan");
    if (h && h->device != p->device)
        return fail(GLV_ERR_INVALID, "the communicator lives on device %d, the plan on %d", h->device, p->device);
    p->halo = h;
    if (p->shard_graph.exec) {  // its transfers name the communicator that was attached before
        DeviceGuard guard(p->device);
        (void)hipGraphExecDestroy(p->shard_graph.exec);
        p->shard_graph.exec = nullptr;
    }
    return GLV_OK;
}

extern "C" int glv_shard_exchange(glv_plan* p, void* stream) {
    int rc = shard_check(p);
    if (rc != GLV_OK) return rc;
    if (!p->halo) return fail(GLV_ERR_STATE, "glv_halo_attach has not been called");
    const glv_halo* h = p->halo;
    RcclApi* api = rccl_api();
    if (!api) return fail(GLV_ERR_UNSUPPORTED, "RCCL is not usable: %s", rccl_why());
    float *send_prev, *send_next, *recv_prev, *recv_next;
    rc = glv_shard_halo_views(p, &send_prev, &send_next, &recv_prev, &recv_next);
    if (rc != GLV_OK) return rc;
    // the shard before this one is the rank before this one, and likewise after
    const int prev = p->halo_loopback ? h->rank : h->rank - 1;
    const int next = p->halo_loopback ? h->rank : h->rank + 1;
    if (send_prev && prev < 0) return fail(GLV_ERR_STATE, "shard starts at frame %d but rank 0 has no rank before it", p->shard_t0);
    if (send_next && next >= h->world)
        return fail(GLV_ERR_STATE, "shard ends at frame %d of %d but rank %d is the last", p->shard_t1, p->shard_T,
                    h->rank);
    if (!send_prev && !send_next) return GLV_OK;
    DeviceGuard guard(p->device);
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)(p->R - 1) * p->hop;
    // one group: all four transfers progress together, so neighbours that post them in the
    // same order cannot wait on each other
    RCCL_TRY(api, api->GroupStart());
    int e = 0;
    if (send_prev) {
        e = api->Send(send_prev, n, kRcclFloat32, prev, h->comm, st);
        if (!e) e = api->Recv(recv_prev, n, kRcclFloat32, prev, h->comm, st);
    }
    if (!e && send_next) {
        e = api->Send(send_next, n, kRcclFloat32, next, h->comm, st);
        if (!e) e = api->Recv(recv_next, n, kRcclFloat32, next, h->comm, st);
    }
    const int eg = api->GroupEnd();
    if (e) return fail(GLV_ERR_HIP, "ncclSend / ncclRecv failed: %s", api->GetErrorString(e));
    if (eg) return fail(GLV_ERR_HIP, "ncclGroupEnd failed: %s", api->GetErrorString(eg));
    return GLV_OK;
}

namespace {

// n_iter times { step; swap }, enqueued.  With `side` the workgroups at the shard's two edges
// (the only ones that read what a swap brings, and the only ones whose output a swap sends) are
// launched on `st`, followed by the swap, while the workgroups between them run on `side`: an
// iteration's interior launch waits for both launches of the iteration before, its edge launch
// for those and for the swap in between.  ev: interior done [2], edges done [2], fork, join.
// (The transfers stay on the stream the capture began on: RCCL 2.26 faults when asked to add
// them to a stream that joined the capture later.)
int enqueue_shard_chain(glv_plan* p, int n_iter, float momentum, hipStream_t st, hipStream_t side,
                        hipEvent_t* ev) {
    const size_t n_wg = engine_grid(p);
    if (!side || n_wg < 3 || !p->linear) {
        int rc = GLV_OK;
        for (int k = 0; k < n_iter && rc == GLV_OK; ++k) {
            rc = glv_shard_step(p, momentum, st);
            if (rc == GLV_OK) rc = glv_shard_exchange(p, st);
        }
        return rc;
    }
    HIP_TRY(hipEventRecord(ev[4], st));
    HIP_TRY(hipStreamWaitEvent(side, ev[4], 0));
    for (int k = 0; k < n_iter; ++k) {
        if (k > 0) {
            HIP_TRY(hipStreamWaitEvent(side, ev[2 + ((k - 1) & 1)], 0));
            HIP_TRY(hipStreamWaitEvent(st, ev[(k - 1) & 1], 0));
        }
        int rc = shard_launch(p, momentum, side, 1, 0, n_wg - 2);
        if (rc != GLV_OK) return rc;
        HIP_TRY(hipEventRecord(ev[k & 1], side));
        rc = shard_launch(p, momentum, st, 0, (int)n_wg - 2, 2);
        if (rc != GLV_OK) return rc;
        HIP_TRY(hipEventRecord(ev[2 + (k & 1)], st));
        shard_advance(p);
        rc = glv_shard_exchange(p, st);
        if (rc != GLV_OK) return rc;
    }
    HIP_TRY(hipEventRecord(ev[5], side));
    HIP_TRY(hipStreamWaitEvent(st, ev[5], 0));
    return GLV_OK;
}

}  // namespace

extern "C" int glv_shard_run(glv_plan* p, int n_iter, float momentum, void* stream) {
    int rc = shard_check(p);
    if (rc != GLV_OK) return rc;
    if (n_iter < 0) return fail(GLV_ERR_INVALID, "n_iter must be >= 0");
    if (momentum < 0.f) return fail(GLV_ERR_INVALID, "griffinlim() called with momentum=%g < 0", momentum);
    if (!p->halo) return fail(GLV_ERR_STATE, "glv_halo_attach has not been called");
    DeviceGuard guard(p->device);
    hipStream_t st = (hipStream_t)stream;
    // the swap that completes the initial inverse transform (enqueued as it is: the first transfer
    // between two ranks sets their connection up, which no capture can hold), then step and swap
    // n_iter times; the host never waits
    rc = glv_shard_exchange(p, st);
    p->shard_last_form = 0;
    if (rc != GLV_OK || n_iter == 0) return rc;
    int form = (p->use_graph && !p->shard_graph_refused) ? p->shard_graph_form : 0;
    // (two branches cost a few microseconds of waiting on each other per iteration: worth it once
    // the interior launch is long enough to hide a swap behind)
    if (form < 0) form = (int64_t)engine_grid(p) >= 2 * (int64_t)p->n_cus ? 2 : 1;
    if (form == 0) return enqueue_shard_chain(p, n_iter, momentum, st, nullptr, nullptr);

    glv_plan::ShardGraph& g = p->shard_graph;
    const bool hit = g.exec && g.t0 == p->shard_t0 && g.t1 == p->shard_t1 && g.T == p->shard_T &&
                     g.n_runs == p->n_runs && g.n_iter == n_iter && g.cur == p->shard_cur &&
                     g.first == (p->shard_k == 0) && g.loopback == p->halo_loopback && g.form == form &&
                     g.momentum == momentum && g.halo == p->halo;
    if (!hit) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        g.exec = nullptr;
        if (!p->shard_main) {
            HIP_TRY(hipStreamCreateWithFlags(&p->shard_main, hipStreamNonBlocking));
            HIP_TRY(hipStreamCreateWithFlags(&p->shard_side, hipStreamNonBlocking));
            for (hipEvent_t& e : p->shard_ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        const int cur0 = p->shard_cur, k0 = p->shard_k;
        hipGraph_t graph = nullptr;
        hipError_t e = hipStreamBeginCapture(p->shard_main, hipStreamCaptureModeRelaxed);
        int lrc = GLV_OK;
        if (e == hipSuccess) {
            lrc = enqueue_shard_chain(p, n_iter, momentum, p->shard_main, form == 2 ? p->shard_side : nullptr,
                                      p->shard_ev);
            e = hipStreamEndCapture(p->shard_main, &graph);
        }
        hipGraphExec_t exec = nullptr;
        if (e == hipSuccess && lrc == GLV_OK) e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (graph) (void)hipGraphDestroy(graph);
        const int end_cur = p->shard_cur, end_k = p->shard_k;
        p->shard_cur = cur0;
        p->shard_k = k0;
        if (e != hipSuccess || lrc != GLV_OK || !exec) {
            // (an RCCL build that cannot be captured, for one): nothing of the chain has run
            (void)hipGetLastError();
            p->shard_graph_refused = true;
            return enqueue_shard_chain(p, n_iter, momentum, st, nullptr, nullptr);
        }
        g = glv_plan::ShardGraph{p->shard_t0, p->shard_t1, p->shard_T, p->n_runs, n_iter, cur0, k0 == 0,
                                 p->halo_loopback, form, momentum, p->halo, end_cur, end_k, exec};
    }
    HIP_TRY(hipGraphLaunch(g.exec, st));
    p->shard_cur = g.end_cur;
    p->shard_k = g.end_k;
    p->shard_last_form = form;
    return GLV_OK;
}

extern "C" int glv_shard_last_run_form(const glv_plan* p) { return p ? p->shard_last_form : 0; }
extern "C" int glv_plan_last_loop_groups(const glv_plan* p) { return p ? p->last_loop_groups : 0; }

namespace {

int peak_prepare(glv_plan* p, const int* n_samples_host, int batch, int64_t y_stride, hipStream_t st,
                 dim3* grid) {
    if (batch < 1 || batch > p->max_batch)
        return fail(GLV_ERR_INVALID, "batch %d outside [1, %d]", batch, p->max_batch);
    int64_t max_len = 1;
    for (int b = 0; b < batch; ++b) {
        if (n_samples_host[b] < 0 || n_samples_host[b] > y_stride)
            return fail(GLV_ERR_INVALID, "n_samples[%d] = %d outside [0, y_stride]", b, n_samples_host[b]);
        max_len = std::max<int64_t>(max_len, n_samples_host[b]);
    }
    // pageable source: the copy is staged before the call returns
    HIP_TRY(hipMemcpyAsync(p->d_nsamples, n_samples_host, sizeof(int) * batch, hipMemcpyHostToDevice, st));
    *grid = dim3(std::min<unsigned>(grid_for(max_len, 256 * 8), 64), batch);
    return GLV_OK;
}

}  // namespace

extern "C" int glv_peak_reduce(glv_plan* p, const float* y_dev, const int* n_samples_host, int batch,
                               int64_t y_stride, float* peak_dev, void* stream) {
    if (!p || !y_dev || !n_samples_host || !peak_dev) return fail(GLV_ERR_INVALID, "null argument");
    DeviceGuard guard(p->device);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid;
    int rc = peak_prepare(p, n_samples_host, batch, y_stride, st, &grid);
    if (rc != GLV_OK) return rc;
    // max|y| >= 0, so the float's bit pattern orders like the value and 0 bits is the identity
    HIP_TRY(hipMemsetAsync(peak_dev, 0, sizeof(float) * batch, st));
    hipLaunchKernelGGL(peak_reduce_kernel, grid, dim3(256), 0, st, y_dev, y_stride, p->d_nsamples,
                       reinterpret_cast<unsigned*>(peak_dev));
    HIP_TRY(hipGetLastError());
    return GLV_OK;
}

extern "C" int glv_peak_scale(glv_plan* p, float* y_dev, const int* n_samples_host, int batch,
                              int64_t y_stride, const float* peak_dev, float eps, void* stream) {
    if (!p || !y_dev || !n_samples_host || !peak_dev) return fail(GLV_ERR_INVALID, "null argument");
    DeviceGuard guard(p->device);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid;
    int rc = peak_prepare(p, n_samples_host, batch, y_stride, st, &grid);
    if (rc != GLV_OK) return rc;
    hipLaunchKernelGGL(peak_scale_kernel, grid, dim3(256), 0, st, y_dev, y_stride, p->d_nsamples,
                       reinterpret_cast<const unsigned*>(peak_dev), eps, 0);
    HIP_TRY(hipGetLastError());
    return GLV_OK;
}

extern "C" int glv_peak_normalize(glv_plan* p, float* y_dev, const int* n_samples_host, int batch,
                                  int64_t y_stride, float eps, void* stream) {
    if (!p) return fail(GLV_ERR_INVALID, "null argument");
    int rc = glv_peak_reduce(p, y_dev, n_samples_host, batch, y_stride, reinterpret_cast<float*>(p->d_peak),
                             stream);
    if (rc != GLV_OK) return rc;
    return glv_peak_scale(p, y_dev, n_samples_host, batch, y_stride, reinterpret_cast<const float*>(p->d_peak),
                          eps, stream);
}

// ---------------------------------------------------------------------------
// peak normalisation for callers that cannot allocate
// ---------------------------------------------------------------------------
// Per-row maxima of glv_peak_normalize_noalloc: device memory that exists from the moment the
// library is loaded, one copy per GPU, so the call needs neither a plan nor an allocation.
#define GLV_NOALLOC_MAX_BATCH 32768
static __device__ __align__(256) unsigned scratch[GLV_NOALLOC_MAX_BATCH];

namespace {
constexpr int kMaxDevices = 64;
// device address of `scratch` on every GPU seen so far (written once per device)
std::atomic<unsigned*> g_scratch_ptr[kMaxDevices];
std::mutex g_scratch_mutex;

// The array's name is a host-side handle, not a device address: the address each GPU holds its
// copy at is looked up once per device, with that device current.  The look-up (which may load
// the code object) happens in glv_plan_create or in the first glv_peak_normalize_noalloc on the
// device; later calls only read the stored address and take no lock.
int scratch_for_current_device(unsigned** out) {
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return fail(GLV_ERR_UNSUPPORTED, "device index %d not supported", dev);
    unsigned* ptr = g_scratch_ptr[dev].load(std::memory_order_acquire);
    if (!ptr) {
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        ptr = g_scratch_ptr[dev].load(std::memory_order_relaxed);
        if (!ptr) {
            HIP_TRY(hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(scratch)));
            if (!ptr) return fail(GLV_ERR_HIP, "hipGetSymbolAddress returned a null address");
            g_scratch_ptr[dev].store(ptr, std::memory_order_release);
        }
    }
    *out = ptr;
    return GLV_OK;
}
}  // namespace

// glv_plan_create resolves the address for its device, so that a process that holds a plan can
// make even its first glv_peak_normalize_noalloc call under stream capture
static int noalloc_warm_up_current_device() {
    unsigned* unused = nullptr;
    return scratch_for_current_device(&unused);
}

extern "C" int glv_peak_normalize_noalloc(float* y_dev, const int* n_samples_dev, int batch, int64_t y_stride,
                                          float eps, void* stream) {
    if (!y_dev || !n_samples_dev) return fail(GLV_ERR_INVALID, "null argument");
    if (batch < 1 || batch > GLV_NOALLOC_MAX_BATCH)
        return fail(GLV_ERR_INVALID, "batch %d outside [1, %d]: glv_peak_normalize_noalloc keeps its per-row maxima "
                    "in a fixed device array; use glv_peak_normalize for larger batches", batch, GLV_NOALLOC_MAX_BATCH);
    if (y_stride < 1) return fail(GLV_ERR_INVALID, "y_stride must be >= 1");
    unsigned* peaks = nullptr;
    int rc = scratch_for_current_device(&peaks);
    if (rc != GLV_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    // the array keeps the previous call's maxima; 0 bits is the identity of the maximum
    HIP_TRY(hipMemsetAsync(peaks, 0, sizeof(unsigned) * batch, st));
    const dim3 grid(std::min<unsigned>(grid_for(y_stride, 256 * 8), 64), batch);
    hipLaunchKernelGGL(peak_reduce_kernel, grid, dim3(256), 0, st, y_dev, y_stride, n_samples_dev, peaks);
    hipLaunchKernelGGL(peak_scale_kernel, grid, dim3(256), 0, st, y_dev, y_stride, n_samples_dev, peaks, eps, 0);
    HIP_TRY(hipGetLastError());
    return GLV_OK;
}

namespace {

// librosa.stft of every row into the plan's native spectrum buffer: run table, centred zero
// padding (with the trailing partial hop of rows given by their sample counts), transform
int forward_transform(glv_plan* p, const float* y_dev, int64_t y_stride, const int* n_frames_host,
                      const int* n_samples_host, int batch, int frames_stride, hipStream_t st) {
    int rc = prepare_runs(p, n_frames_host, batch, frames_stride, st);
    if (rc != GLV_OK) return rc;
    int max_t = 0;
    for (int b = 0; b < batch; ++b) {
        max_t = std::max(max_t, n_frames_host[b]);
        const int64_t whole = (int64_t)p->hop * (n_frames_host[b] - 1);
        const int64_t len = n_samples_host ? n_samples_host[b] : whole;
        if (len < whole || len >= whole + p->hop)
            return fail(GLV_ERR_INVALID, "n_samples[%d] = %lld does not give n_frames[%d] = %d frames "
                        "(n_frames = 1 + n_samples / hop)", b, (long long)len, b, n_frames_host[b]);
        if (len > y_stride) return fail(GLV_ERR_INVALID, "y_stride shorter than row %d", b);
    }
    if (n_samples_host)  // pageable source: staged before the call returns
        HIP_TRY(hipMemcpyAsync(p->d_nsamples, n_samples_host, sizeof(int) * batch, hipMemcpyHostToDevice, st));
    const ElemArgs ea = elem_args(p, batch, frames_stride);
    hipLaunchKernelGGL(pad_signal_kernel, dim3(grid_for((int64_t)p->hop * max_t, 256), batch), dim3(256), 0, st, ea,
                       y_dev, y_stride, n_samples_host ? p->d_nsamples : (const int*)nullptr, p->d_ym[0]);
    IterArgs a = iter_args(p, frames_stride);
    a.raw_in = 1;
    set_io(p, a, 0, 1);
    HIP_TRY(p->launch_engine(MODE_STFT, false, a, engine_grid(p), st));
    return GLV_OK;
}

}  // namespace

extern "C" int glv_stft(glv_plan* p, const float* y_dev, int64_t y_stride, const int* n_frames_host,
                        const int* n_samples_host, int batch, int frames_stride, float* D_dev, void* stream) {
    if (!p || !y_dev || !D_dev) return fail(GLV_ERR_INVALID, "null argument");
    DeviceGuard guard(p->device);
    hipStream_t st = (hipStream_t)stream;
    int rc = forward_transform(p, y_dev, y_stride, n_frames_host, n_samples_host, batch, frames_stride, st);
    if (rc != GLV_OK) return rc;
    const ElemArgs ea = elem_args(p, batch, frames_stride);
    p->launch_convert(false, ea, p->d_spec, 1.0f, reinterpret_cast<cf*>(D_dev), st);
    HIP_TRY(hipGetLastError());
    return GLV_OK;
}


extern "C" int glv_melspectrogram(glv_plan* p, const float* y_dev, int64_t y_stride, const int* n_frames_host,
                                  const int* n_samples_host, int batch, int frames_stride, float* mel_dev,
                                  void* stream) {
    if (!p || !y_dev || !mel_dev) return fail(GLV_ERR_INVALID, "null argument");
    if (!p->have_basis) return fail(GLV_ERR_STATE, "glv_plan_set_mel_basis has not been called");
    DeviceGuard guard(p->device);
    hipStream_t st = (hipStream_t)stream;
    int rc = forward_transform(p, y_dev, y_stride, n_frames_host, n_samples_host, batch, frames_stride, st);
    if (rc != GLV_OK) return rc;
    const ElemArgs ea = elem_args(p, batch, frames_stride);
    // |D|^2 in natural order (the S workspace is idle here), then mel = power @ basis^T
    p->launch_power(ea, p->d_spec, p->d_S, st);
    const int64_t rows = (int64_t)batch * frames_stride;
    const dim3 grid((p->n_mels + 63) / 64, (unsigned)((rows + 63) / 64));
    hipLaunchKernelGGL((gemm_nt_kernel<false, EPI_STORE>), grid, dim3(256), 0, st, p->d_S, p->d_basis, rows, p->F,
                       p->n_mels, 0.f, 0.f, 0.f, (const float*)nullptr, mel_dev);
    HIP_TRY(hipGetLastError());
    return GLV_OK;
}

extern "C" int glv_power_to_db(glv_plan* p, const float* x_dev, const int* n_valid_host, int batch,
                               int64_t stride, int ref_is_max, float ref_value, float amin, float top_db,
                               float* out_dev, void* stream) {
    if (!p || !x_dev || !out_dev || !n_valid_host) return fail(GLV_ERR_INVALID, "null argument");
    if (!(amin > 0.f)) return fail(GLV_ERR_INVALID, "amin must be strictly positive");
    DeviceGuard guard(p->device);
    hipStream_t st = (hipStream_t)stream;
    // per-row maximum (power is non-negative): the reference level and/or the top_db floor
    int rc = glv_peak_reduce(p, x_dev, n_valid_host, batch, stride, reinterpret_cast<float*>(p->d_peak), stream);
    if (rc != GLV_OK) return rc;
    int64_t max_len = 1;
    for (int b = 0; b < batch; ++b) max_len = std::max<int64_t>(max_len, n_valid_host[b]);
    const dim3 grid(std::min<unsigned>(grid_for(max_len, 256 * 4), 256), batch);
    const float* maxes = reinterpret_cast<const float*>(p->d_peak);
    hipLaunchKernelGGL(power_to_db_kernel, grid, dim3(256), 0, st, x_dev, stride, p->d_nsamples,
                       ref_is_max ? maxes : (const float*)nullptr, ref_value, amin, top_db, maxes, out_dev);
    HIP_TRY(hipGetLastError());
    return GLV_OK;
}

extern "C" int glv_istft(glv_plan* p, const float* D_dev, const int* n_frames_host, int batch,
                         int frames_stride, float* y_dev, int64_t y_stride, void* stream) {
    if (!p || !y_dev || !D_dev) return fail(GLV_ERR_INVALID, "null argument");
    DeviceGuard guard(p->device);
    hipStream_t st = (hipStream_t)stream;
    int rc = prepare_runs(p, n_frames_host, batch, frames_stride, st);
    if (rc != GLV_OK) return rc;
    int max_t = 0;
    for (int b = 0; b < batch; ++b) max_t = std::max(max_t, n_frames_host[b]);
    if (y_stride < (int64_t)p->hop * (max_t - 1))
        return fail(GLV_ERR_INVALID, "y_stride shorter than hop*(n_frames-1)");
    const ElemArgs ea = elem_args(p, batch, frames_stride);
    p->launch_convert(true, ea, reinterpret_cast<const cf*>(D_dev), 1.0f / (float)p->n_fft, p->d_spec, st);
    IterArgs a = iter_args(p, frames_stride);
    set_io(p, a, 1, 0);
    HIP_TRY(p->launch_engine(MODE_ISTFT, false, a, engine_grid(p), st));
    hipLaunchKernelGGL(finalize_kernel, dim3(grid_for(y_stride, 256), batch), dim3(256), 0, st, ea,
                       p->d_ym[0], p->d_yt[0], p->d_flags, p->blocks, p->d_w2, p->d_iwss, p->d_iwss_head, p->d_iwss_tail,
                       p->d_iwss_small, y_dev, y_stride, (int64_t)0, (unsigned*)nullptr);
    HIP_TRY(hipGetLastError());
    return GLV_OK;
}

#ifdef GLV_WAVE_TIMES
// measurement build only: copies the time stamps of the last iteration launch to the host
extern "C" int glv_debug_wave_times(unsigned long long* out, int n_waves) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(glv_wave_times), sizeof(unsigned long long) * 20 * n_waves) == hipSuccess ? 0 : -1;
}
#endif
