/*
 * glvoc -- C ABI of the MI355X (gfx950) Griffin-Lim vocoder library, libglvoc.so.
 *
 * The reference has no FFI for this path: its boundary is the Python function
 *     mel_to_audio_griffin_lim(mel_db, mean, std, sr, n_fft, hop_length,
 *                              win_length, n_mels, fmax) -> ndarray
 * (reference vocoder/mel_audio.py:112-132, copies at vocoder/pra.py:65-80 and
 * test.py:143-158).  Each entry point below replaces one stage of that
 * function and cites the line it stands in for.  A binding needs nothing but
 * plain pointers and sizes: tts_amd/_lib.py binds these with ctypes, passing
 * torch device pointers (Tensor.data_ptr()) and the raw hipStream_t.
 *
 * Conventions
 *   - Every function returns GLV_OK (0) or a negative glv_status;
 *     glv_last_error() returns a thread-local message for the last failure.
 *   - Pointers named *_dev are device memory owned by the caller; pointers
 *     named *_host are host memory read before the call returns.
 *   - All work is enqueued on `stream` (a hipStream_t, NULL = default stream)
 *     and is asynchronous.  glv_plan_create / glv_plan_destroy synchronise the device, the
 *     glv_plan_set_*_basis uploads are blocking copies, glv_plan_last_loop_ms waits for its
 *     event.  A compute call waits on the host only in two cases: (a) it needs more of the
 *     usage-dependent workspace than glv_plan_create / glv_plan_reserve set aside (it then
 *     drains the stream, reallocates and drops its captured graphs), (b) the pinned staging
 *     set it is about to rewrite still feeds the upload of the third-last call with new
 *     lengths.  The first call with a new (lengths, n_iter, momentum) shape also captures a
 *     hipGraph of the iteration loop; the 8 most recently used shapes stay instantiated.
 *   - A plan is bound to one device and must be used from one stream at a time.
 *   - Layouts are frame-major: mel (B, T, n_mels), spectra (B, T, F) with
 *     F = n_fft/2+1 contiguous per frame, waveforms (B, y_stride).  `T` is the
 *     common row stride in frames (frames_stride); the true frame count of
 *     utterance b is n_frames_host[b] <= frames_stride <= max_frames.
 *   - Complex values are interleaved (re, im) float pairs.
 */
#ifndef GLVOC_H
#define GLVOC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct glv_plan glv_plan;

typedef enum glv_status {
    GLV_OK = 0,
    GLV_ERR_INVALID = -1,     /* bad argument (shape, size, null pointer)          */
    GLV_ERR_UNSUPPORTED = -2, /* n_fft / hop combination the kernels do not cover  */
    GLV_ERR_HIP = -3,         /* a HIP runtime call failed                         */
    GLV_ERR_STATE = -4        /* call order (e.g. inverse basis not set)           */
} glv_status;

/* Library version: major*10000 + minor*100 + patch. */
int glv_version(void);

/*
 * Creates the per-shape state: window / twiddle / window-sum-square tables and
 * the device workspace of the iteration loop.  Stands in for the keyword
 * defaults of the reference signature (vocoder/mel_audio.py:112).
 * Supported: n_fft in {1024, 2048}; hop in {n_fft/8, n_fft/4, n_fft/2}
 * (128, 256, 512 for n_fft 1024; 256, 512, 1024 for n_fft 2048);
 * win_length <= n_fft (periodic Hann, centred zero padding).
 */
int glv_plan_create(glv_plan** out, int n_fft, int hop, int win_length, int n_mels,
                    int max_frames, int max_batch, int device);

void glv_plan_destroy(glv_plan* plan);

/* Bytes of device memory the plan holds. */
int64_t glv_plan_workspace_bytes(const glv_plan* plan);

/*
 * Sets aside, ahead of time, the workspace whose size depends on how the plan is used rather
 * than on its capacity, for calls of up to (batch, frames_stride, n_iter): the per-run partial
 * sums of the convergence curve (sc_curve_dev != NULL) and, with single_frame != 0, the
 * partial-sum buffers of the one-frame-per-wave iteration when that mode is FORCED (option
 * "single_frame" = 1) on a batch larger than the automatic choice would take.
 * glv_plan_create already reserves what the automatic mode choice and a 64-iteration curve
 * over the automatic run table need, so most callers never call this.  No counterpart in the
 * reference (numpy allocates per call).
 */
int glv_plan_reserve(glv_plan* plan, int batch, int frames_stride, int n_iter, int single_frame,
                     void* stream);

/*
 * Identity of every utterance of the following calls, for the initial phase drawn on the device
 * (phase0_dev == NULL): utterance b draws the stream (seed, ids_host[b]) indexed by (frame, bin),
 * so its waveform does not depend on the batch it is in, on its row, or on frames_stride.
 * ids_host == NULL (the state after glv_plan_create) identifies an utterance by its row.
 * Stands in for the reference drawing a fresh default_rng per call of
 * librosa.griffinlim (vocoder/mel_audio.py:126): one independent stream per utterance.
 */
int glv_plan_set_utterance_ids(glv_plan* plan, const int64_t* ids_host, int batch, void* stream);

/*
 * Uploads the (F, n_mels) row-major float32 pseudo-inverse of the mel basis,
 * i.e. the result of np.linalg.pinv(librosa.filters.mel(...)) computed on the
 * host (reference vocoder/mel_audio.py:121-122).
 */
int glv_plan_set_inverse_basis(glv_plan* plan, const float* pinv_FxM_host);

/*
 * Denormalise, dB -> power, and multiply by the inverse basis
 * (reference vocoder/mel_audio.py:114, :118, :123):
 *     S[b, t, :] = pinv @ 10 ** (0.1 * (mel_db[b, t, :] * std + mean))
 * mel_db_dev: (batch, frames_stride, n_mels) float32.  S_dev: (batch,
 * frames_stride, F) float32.  All frames_stride rows are computed.
 */
int glv_mel_invert(glv_plan* plan, const float* mel_db_dev, int batch, int frames_stride,
                   float mean, float std, float* S_dev, void* stream);

/*
 * Non-negative inversion, the mel_to_stft step of librosa.feature.inverse.mel_to_audio
 * that the reference's older scripts call (test_cnn_model.py:117-119, test_1.py:52-54,
 * z_AR/test_AR_model.py:12-14):
 *     X = argmin_{X >= 0} 0.5 * || basis @ X - mel ||^2 ,  S = X ** (1 / power)
 * started from librosa's X0 = max(pinv @ mel, 0) and advanced by at most n_steps
 * projected-gradient steps X <- max(X - step * basis^T (basis X - mel), 0) (librosa runs
 * L-BFGS-B from the same start; F > n_mels, so the minimiser is not unique and only the
 * mel-domain residual is comparable between solvers).  step should be 1 / ||basis||_2^2.
 * With option "nnls_accel" (default 1) the steps are extrapolated the Nesterov / FISTA way -
 * the gradient is taken at y_k = x_k + beta_k (x_k - x_{k-1}), beta_k = (t_k - 1) / t_{k+1},
 * t_{k+1} = (1 + sqrt(1 + 4 t_k^2)) / 2 - which reaches the residual of 200 plain steps in
 * about 35; 0 = plain steps.  ("nnls_fused" = 0, one product per launch, always takes plain steps.)
 * A frame stops moving once its residual ||basis X - mel|| <= tol * ||mel|| (tol = 0: every
 * frame takes all n_steps steps); frames do not influence each other.
 *   glv_plan_set_mel_basis: (n_mels, F) row-major float32 librosa.filters.mel matrix.
 *   mel_dev: (batch, frames_stride, n_mels) mel POWER values (not dB); S_dev as glv_mel_invert.
 */
int glv_plan_set_mel_basis(glv_plan* plan, const float* basis_MxF_host);
int glv_mel_invert_nnls(glv_plan* plan, const float* mel_dev, int batch, int frames_stride,
                        int n_steps, float step, float tol, float power, float* S_dev, void* stream);

/*
 * librosa.griffinlim(S, n_iter, hop_length, win_length, momentum, init='random')
 * as called at reference vocoder/mel_audio.py:126 (window='hann', center=True,
 * pad_mode='constant').
 *   S_dev        (batch, frames_stride, F) float32 "magnitudes" (may be negative).
 *   phase0_dev   (batch, frames_stride, F) interleaved unit phasors, or NULL to
 *                draw them on the device from `seed` (counter-based generator;
 *                not numpy's stream).
 *   y_dev        (batch, y_stride) float32; the first hop*(n_frames[b]-1) samples
 *                of row b are written, the rest of the row is zero-filled.
 *   sc_curve_dev NULL, or (batch, n_iter) float32 receiving the spectral
 *                convergence || |stft(istft(angles_k))| - |S| || / ||S|| of every
 *                iteration.
 */
int glv_griffinlim(glv_plan* plan, const float* S_dev, const float* phase0_dev, uint64_t seed,
                   const int* n_frames_host, int batch, int frames_stride, int n_iter,
                   float momentum, float* y_dev, int64_t y_stride, float* sc_curve_dev,
                   void* stream);

/*
 * The whole of mel_to_audio_griffin_lim (reference vocoder/mel_audio.py:112-132) in one call:
 * glv_mel_invert + glv_griffinlim + glv_peak_normalize, with the same results bit for bit, but
 * the inverted magnitudes go straight from the matrix product into the iteration state
 * (no (batch, frames, F) array in between) and the peak is gathered while the waveform is
 * written.  Arguments as for the three calls; peak_eps is the reference's 1e-6, or negative
 * to leave the waveform un-normalised.
 */
int glv_mel_to_audio(glv_plan* plan, const float* mel_db_dev, float mean, float std,
                     const float* phase0_dev, uint64_t seed, const int* n_frames_host, int batch,
                     int frames_stride, int n_iter, float momentum, float peak_eps, float* y_dev,
                     int64_t y_stride, float* sc_curve_dev, void* stream);

/*
 * audio / (max|audio| + eps) per utterance (reference vocoder/mel_audio.py:129,
 * eps = 1e-6).  In place on y_dev rows of n_samples_host[b] samples.
 */
int glv_peak_normalize(glv_plan* plan, float* y_dev, const int* n_samples_host, int batch,
                       int64_t y_stride, float eps, void* stream);

/*
 * glv_peak_normalize for callers that cannot allocate device memory or hold a plan: code under
 * hipGraph capture, tight serving loops, harnesses that forbid hipMalloc.  It has no counterpart
 * in the reference.  Same arithmetic and the same bits as glv_peak_normalize; the differences:
 *   - no plan: it runs on the CURRENT device, on `stream`;
 *   - the row lengths are read on the device: n_samples_dev is device memory, (batch) int32,
 *     0 <= n_samples_dev[b] <= y_stride (not checked);
 *   - the per-row maxima live in a fixed device array inside the library (128 KiB per GPU, held
 *     until the process ends), so batch <= 32768; larger batches are refused with
 *     GLV_ERR_INVALID;
 *   - it calls no hipMalloc / hipFree and does not synchronise the device;
 *   - the device address of that array is looked up once per GPU (hipGetSymbolAddress, which may
 *     load the code object and is not a stream operation): by glv_plan_create for the plan's
 *     device, else by the first call on the device - make that first call OUTSIDE stream capture
 *     in a process that never created a plan there.  Later calls read the stored address without
 *     taking a lock.
 * Every call on a GPU uses that GPU's one array: calls that overlap in time on one GPU (two
 * streams, two threads) race on it - serialise them.
 */
int glv_peak_normalize_noalloc(float* y_dev, const int* n_samples_dev, int batch, int64_t y_stride,
                               float eps, void* stream);

/* The two halves of glv_peak_normalize, for callers that must combine the peak of
 * several pieces first (time-sharded synthesis takes the maximum over ranks in between):
 * peak_dev[b] = max|y[b, :n_samples[b]]|, then y[b] /= (peak_dev[b] + eps). */
int glv_peak_reduce(glv_plan* plan, const float* y_dev, const int* n_samples_host, int batch,
                    int64_t y_stride, float* peak_dev, void* stream);
int glv_peak_scale(glv_plan* plan, float* y_dev, const int* n_samples_host, int batch,
                   int64_t y_stride, const float* peak_dev, float eps, void* stream);

/*
 * librosa.stft / librosa.istft with the plan's n_fft, hop and window, exposed
 * for stage-wise checks of the two transforms inside librosa.griffinlim
 * (reference vocoder/mel_audio.py:126).
 *   glv_stft : y_dev (batch, y_stride); row b holds n_samples_host[b] samples and gives
 *              n_frames_host[b] = 1 + n_samples_host[b] / hop frames, as librosa frames a centred
 *              signal of any length (n_samples_host NULL: hop*(n_frames[b]-1) samples per row)
 *              -> D_dev (batch, frames_stride, F) interleaved complex64.
 *   glv_istft: D_dev -> y_dev (batch, y_stride), hop*(n_frames[b]-1) samples per row.
 */
int glv_stft(glv_plan* plan, const float* y_dev, int64_t y_stride, const int* n_frames_host,
             const int* n_samples_host, int batch, int frames_stride, float* D_dev, void* stream);
int glv_istft(glv_plan* plan, const float* D_dev, const int* n_frames_host, int batch,
              int frames_stride, float* y_dev, int64_t y_stride, void* stream);

/*
 * The forward direction, as the reference's preprocessing computes its training mels
 * (acoustic/dataset.py:75-84): librosa.feature.melspectrogram (|stft|^2 @ mel basis) and
 * librosa.power_to_db.  Same transform kernels as the vocoder; lets a wav -> mel -> wav round
 * trip stay on the device.
 *   glv_melspectrogram: y_dev (batch, y_stride), rows and frame counts as for glv_stft (signals of
 *                       any length: the reference feeds whole utterances) -> mel_dev (batch,
 *                       frames_stride, n_mels) power values; needs glv_plan_set_mel_basis.
 *   glv_power_to_db   : rows of n_valid_host[b] values -> 10*log10(max(amin, x)) -
 *                       10*log10(max(amin, ref)), ref = the row maximum (ref_is_max, i.e.
 *                       ref=np.max) or ref_value; then floored at (row maximum - top_db) unless
 *                       top_db < 0.  librosa defaults: amin 1e-10, top_db 80.
 */
int glv_melspectrogram(glv_plan* plan, const float* y_dev, int64_t y_stride, const int* n_frames_host,
                       const int* n_samples_host, int batch, int frames_stride, float* mel_dev, void* stream);
int glv_power_to_db(glv_plan* plan, const float* x_dev, const int* n_valid_host, int batch,
                    int64_t stride, int ref_is_max, float ref_value, float amin, float top_db,
                    float* out_dev, void* stream);

/*
 * Time-sharded librosa.griffinlim (reference vocoder/mel_audio.py:126) for one utterance
 * too long for, or wanted faster than, one GPU: this device holds frames
 * [frame_begin, frame_begin + frames_local) of frames_global.  The reference has no
 * counterpart (it runs one utterance on one CPU thread); the arithmetic per frame is that of
 * glv_griffinlim.  Per iteration each shard boundary needs one swap of
 * glv_shard_halo_samples() = n_fft - hop floats in each direction, which the caller moves
 * (RCCL point-to-point, or a device copy when both shards share a GPU):
 *
 *     glv_shard_begin(...);                 init + inverse STFT of the initial angles
 *     swap();                               pack -> send/recv -> unpack
 *     repeat n_iter times { glv_shard_step(); swap(); }
 *     glv_shard_output_range(); glv_shard_end(y);
 *
 *   swap() = glv_shard_halo_pack(to_prev, to_next), exchange with ranks -1 / +1,
 *            glv_shard_halo_unpack(from_prev, from_next).  The first shard has no previous
 *            and the last no next neighbour; those pointers are ignored (may be NULL).
 *   S_dev / phase0_dev hold only the local frames, (frames_local, F); a NULL phase is drawn
 *   from `seed` per global frame index, so the result does not depend on the sharding.
 *   glv_shard_end writes this shard's n_count samples, which start at sample n_begin of the
 *   whole waveform (glv_shard_output_range), un-normalised.
 *   glv_shard_halo_views is the swap without copies: it returns where, inside the plan's
 *   workspace, the halos of the NEWEST generation lie - two regions to send from and two to
 *   receive into, glv_shard_halo_samples() floats each (NULL on the side without a neighbour).
 *   A transport that takes device pointers (RCCL send / recv) moves them in place, ordered on the
 *   stream after glv_shard_begin / glv_shard_step and before the next step; the pointers change
 *   with every step (three generations rotate) and are valid until glv_shard_end.
 * A shard needs at least 2*(n_fft/hop - 1) frames.  One shard per plan at a time.
 */
int glv_shard_begin(glv_plan* plan, const float* S_dev, const float* phase0_dev, uint64_t seed,
                    int frame_begin, int frames_local, int frames_global, void* stream);
int glv_shard_step(glv_plan* plan, float momentum, void* stream);
int glv_shard_halo_samples(const glv_plan* plan);
int glv_shard_halo_pack(glv_plan* plan, float* to_prev_dev, float* to_next_dev, void* stream);
int glv_shard_halo_unpack(glv_plan* plan, const float* from_prev_dev, const float* from_next_dev,
                          void* stream);
int glv_shard_halo_views(glv_plan* plan, float** send_prev, float** send_next, float** recv_prev,
                         float** recv_next);
int glv_shard_output_range(const glv_plan* plan, int64_t* n_begin, int64_t* n_count);
int glv_shard_end(glv_plan* plan, float* y_dev, void* stream);

/*
 * The swap done by the library itself: RCCL point-to-point on the caller's stream, between
 * glv_shard_begin / glv_shard_step and the next step, with no host wait (the reference has no
 * counterpart: it synthesises one utterance on one CPU thread).  One process per GPU,
 * rank r holds shard r of `world` consecutive shards.  A glv_halo is this process's place in
 * that chain (an RCCL communicator); it is made once and serves any number of plans and calls:
 *
 *     rank 0: glv_halo_unique_id(id);  hand the GLV_HALO_ID_BYTES bytes to every rank
 *     all:    glv_halo_create(&halo, id, rank, world, device);   collective: returns when all joined
 *             glv_halo_attach(plan, halo);                        (NULL detaches; the plan does not own it)
 *             glv_shard_begin(...); glv_shard_exchange(plan, stream);
 *             repeat n_iter times { glv_shard_step(...); glv_shard_exchange(plan, stream); }
 *                                                 (or glv_shard_run for the three lines above)
 *             glv_shard_end(...);
 *             ... glv_halo_destroy(halo) when the stream has drained.
 *
 * glv_shard_exchange sends this shard's head blocks to rank - 1 and its tail blocks to rank + 1
 * and receives theirs in place (the regions glv_shard_halo_views names), all four transfers in
 * one RCCL group.  RCCL is loaded at run time (librccl.so.1; the copy the process already holds,
 * if any); without it these calls return GLV_ERR_UNSUPPORTED.  Option "halo_loopback" = 1 is a
 * test aid for a single rank: both neighbours are the rank itself, so each side receives what
 * it sent.
 */
#define GLV_HALO_ID_BYTES 128
typedef struct glv_halo glv_halo;
int glv_halo_unique_id(void* id_out);
int glv_halo_create(glv_halo** out, const void* id, int rank, int world, int device);
int glv_halo_destroy(glv_halo* halo);
int glv_halo_attach(glv_plan* plan, glv_halo* halo);
int glv_shard_exchange(glv_plan* plan, void* stream);
/* glv_shard_exchange, then n_iter times { glv_shard_step; glv_shard_exchange }: the whole loop
 * after glv_shard_begin enqueued by one call. */
int glv_shard_run(glv_plan* plan, int n_iter, float momentum, void* stream);
/* How the last glv_shard_run enqueued its steps and swaps (option "shard_graph" asks for a form;
 * default -1: form 2 for shards of at least two workgroups per compute unit, else form 1; a plan
 * whose capture fails stays with 0):
 *   0  every step and every swap enqueued by the host, one stream;
 *   1  the steps and swaps after the first swap replayed from one hipGraph;
 *   2  the same graph with two branches: the workgroups at the shard's two edges, which alone read
 *      what a swap brings and write what it sends, are launched apart from the workgroups between
 *      them, so a swap overlaps the interior of the iteration after it.
 * All three produce the same bits.  (No counterpart in the reference: vocoder/mel_audio.py:126
 * runs one utterance on one device.) */
int glv_shard_last_run_form(const glv_plan* plan);

/*
 * Tuning and measurement hooks (no counterpart in the reference).
 *   glv_plan_set_option: "frames_per_run" (target frames one wavefront walks; 0 = automatic),
 *                        "run_target" (number of runs the automatic run table aims at; 0 = chosen
 *                        from the batch and the device: whole workgroups per compute unit),
 *                        "linear" (1/0, default 1: form the momentum term in the time domain -
 *                        rebuilt_k - c*rebuilt_{k-1} = STFT(y_k - c*y_{k-1}) - so that no spectrum
 *                        state is kept in memory; 0 = subtract the stored previous spectrum as
 *                        librosa writes it; equal up to rounding),
 *                        "single_frame" (-1 automatic / 0 / 1: give every frame a wavefront of its
 *                        own - R partial-sum buffers instead of a running sum - which is what a
 *                        batch of only a few frames per compute unit needs; automatic only while
 *                        "frames_per_run" is 0),
 *                        "use_graph" (0/1: replay the iteration loop from a hipGraph),
 *                        "timing" (0/1: bracket the iteration loop with hipEvents),
 *                        "recompute" (0/1: iterate without the spectrum state in memory - the
 *                        momentum term's previous spectrum is recomputed from the previous
 *                        signal; one more forward FFT per frame, 2.4x fewer bytes),
 *                        "alias_utts" (0/1: measurement aid - every utterance of the batch uses the first
 *                        one's state, so the loop runs out of cache: compute-only time; output is
 *                        garbage),
 *                        "zero_skip" (1/0, default 1: the iteration leaves out the bins whose rows of the
 *                        inverse basis are all zero - their magnitudes are exactly zero),
 *                        "fuse_draw" (1/0, default 1: with no phase0 and "linear" on, the first inverse
 *                        transform draws the initial phase itself and S*phase is never stored;
 *                        0 = the set-up kernel stores it; same bits either way),
 *                        "deposit" (1/0, default 1: n_fft 1024, "linear" on, no convergence curve - where two
 *                        consecutive runs of an utterance are waves of one workgroup, the later one
 *                        hands its partial sums of the boundary blocks over through LDS and the
 *                        earlier one writes them complete, instead of both writing halves that
 *                        every reader adds; same bits either way),
 *                        "nnls_accel" (1/0, default 1: extrapolated steps in glv_mel_invert_nnls, see there),
 *                        "nnls_fused" (1/0, default 1: all steps of glv_mel_invert_nnls in one launch
 *                        with the solution resident in registers; 0 = two launches per step),
 *                        "traffic_only" (0/1: measurement aid - the iteration kernels issue
 *                        their loads and stores but skip the arithmetic; output is garbage),
 *                        "persistent" (1/0, default 1: with "linear" on and no convergence curve, a call
 *                        whose runs fit the device's wave slots at once runs ALL its iterations in one
 *                        launch - a wave keeps its run, and runs whose boundary blocks travel through
 *                        memory wait for their neighbours' progress words - instead of one launch per
 *                        iteration; same bits.  It needs the device to itself: if other work keeps
 *                        part of the launch from being resident the waits run out after a fraction
 *                        of a second, see glv_plan_status; set 0 where processes share a device - the
 *                        environment variable GLVOC_PERSISTENT=0 does so for every plan of a process.
 *                        2 = also for calls too large to be resident at once: a test aid that makes
 *                        the waits run out).
 *   glv_plan_last_loop_ms: with "timing" on, milliseconds the n_iter iteration
 *                        kernels of the last glv_griffinlim call took on the
 *                        stream (synchronises on the closing event).
 */
int glv_plan_set_option(glv_plan* plan, const char* name, int value);
int glv_plan_last_loop_ms(glv_plan* plan, float* ms_out);
/*
 * GLV_OK, or GLV_ERR_STATE once a wait inside a "persistent" launch has run out (the launch was
 * not resident as a whole): the results of that call are void, and the plan launches every
 * iteration separately from then on.  Reads a host word the device raises; does not synchronise,
 * so ask after the stream has been waited for.  No counterpart in the reference.
 *   glv_plan_persistent_capacity: workgroups of the iteration-spanning kernel (variant 0: all
 *   bins, 1: zero-for-good bins left out) that glv_plan_create saw resident together; calls with
 *   at most that many workgroups (4 runs each) use it.
 */
int glv_plan_status(glv_plan* plan);
int glv_plan_persistent_capacity(const glv_plan* plan, int variant);
/* Kernel launches the iteration loop of the last Griffin-Lim call took: 1 when it ran as one
 * iteration-spanning launch, n_iter otherwise. */
int glv_plan_last_loop_launches(const glv_plan* plan);
/* Groups of consecutive utterances the last Griffin-Lim call was cut into (1: one piece).  A batch
 * with more runs than one iteration-spanning launch holds runs as several such launches, one per
 * group, one after the other; glv_plan_last_loop_launches then counts the launches of ONE group's
 * loop.  A batch that would fit one launch only with very long runs is cut as well (a group's
 * working set per iteration should fit the memory-side cache).  Option "group_frames": most frames
 * per group (0 = automatic, -1 = never cut).  A row's bits are those of a call of its group alone;
 * with a fixed run length (option "frames_per_run") they depend neither on the group nor on the cut. */
int glv_plan_last_loop_groups(const glv_plan* plan);

/*
 * Writes the initial phase the library draws when a call passes phase0_dev == NULL: the unit
 * phasors of (seed, utterance identity, frame, bin) for `batch` utterances, as (batch,
 * frames_stride, F) interleaved pairs - exactly what glv_griffinlim / glv_mel_to_audio use for the
 * same seed and the identities in place (glv_plan_set_utterance_ids).  frame_origin is the global
 * index of row 0's first frame (0, or the shard's first frame for glv_shard_begin).  A checking aid:
 * it lets another implementation of librosa.griffinlim (reference vocoder/mel_audio.py:126) run on
 * the very phase a device-drawn call used.  No counterpart in the reference, whose phase is an
 * unseeded default_rng.
 */
int glv_plan_drawn_phase(glv_plan* plan, uint64_t seed, int batch, int frames_stride, int64_t frame_origin,
                         float* phase_dev, void* stream);

const char* glv_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* GLVOC_H */
