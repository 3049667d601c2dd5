/*
 * include/dss_hip.h -- C ABI of libdss_hip.so, the MI355X (gfx950) hot path of
 * cronelab/delayed-speech-synthesis: LPCNet vocoder + high-gamma (HGA) feature extractor.
 *
 * Plain C, plain pointers and sizes; no torch / C++ types.  Every entry point names the reference
 * interface it replaces (file:line relative to the reference repository).
 *
 * Part 1 exports every xiph/LPCNet symbol the reference's Cython wrapper binds (extensions/lpcnet/cLPCNet.pxd:10-19):
 * the four decoder entry points are implemented, the five feature-ENCODER entry points (corpus preparation, outside
 * the accelerated path) are present and fail cleanly (create returns NULL -> the wrapper's MemoryError,
 * LPCNet.pyx:53-56).  The reference's own LPCNet.pyx therefore compiles and links against this library with the two
 * shim headers under include/compat/ (tests/test_cpu_boundary.py does exactly that; INTEGRATION.md shows the recipe).
 * Part 2 is the batched form of the same operator (what AsynchronousSynthesisQueue's process pool,
 * local/training.py:165-207, and the north-star batch configs need).
 * Part 3 is the HGA operator (extensions/hga/hga_optimized.pyx and HighGammaExtractor,
 * local/units.py:97-161).
 *
 * Error convention: functions returning int return 0 on success and a negative DSS_E* code on failure;
 * dss_last_error() gives a thread-local message.  Creators return NULL on failure (the reference's
 * wrapper turns NULL into MemoryError, LPCNet.pyx:16-17).  There is NO CPU fallback: without a HIP
 * device every compute entry point fails with DSS_ENODEV.
 *
 * Threading (same contract as the reference): one state is used by one thread at a time;
 * different states may be used from different threads.  Do not fork() after the first call.
 */
#ifndef DSS_HIP_H
#define DSS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSS_OK        0
#define DSS_EINVAL   -1   /* bad argument / shape */
#define DSS_ENODEV   -2   /* no HIP device, or HIP runtime error (see dss_last_error) */
#define DSS_ENOMODEL -3   /* no LPCNet weight blob loaded */
#define DSS_ENOMEM   -4

const char *dss_last_error(void);
/* Library version and the device it runs on ("gfx950 ..."); never NULL. */
const char *dss_version(void);
int dss_device_count(void);
/* Select the HIP device used by objects created afterwards on this thread (default 0 / LOCAL_RANK). */
int dss_set_device(int device);
/* The device objects created next on this thread will live on (>= 0), or DSS_ENODEV.  Every object keeps the device
 * it was created on and runs its kernels there; device pointers handed to *_dev entry points must belong to it. */
int dss_current_device(void);

/* ------------------------------------------------------------------------------------------------
 * Part 0 -- streams, events and page-locked host memory as plain handles, for hosts that keep several calls of this
 * library in flight at once (the asynchronous segment synthesis of the gated streaming mode) without binding a HIP
 * runtime themselves.  A stream handle is what every *_dev entry point takes as `hip_stream` (a hipStream_t); streams
 * created here do not synchronise with the null stream.  Nothing in the reference corresponds to these: its one stream
 * blocks while it vocodes (local/units.py:531-538).
 * ---------------------------------------------------------------------------------------------- */
void *dss_stream_create(void);
void dss_stream_destroy(void *hip_stream);
int dss_stream_synchronize(void *hip_stream);
void *dss_event_create(void);
void dss_event_destroy(void *event);
int dss_event_record(void *event, void *hip_stream);
int dss_event_query(void *event);                 /* 1 = finished, 0 = not yet, < 0 = error */
int dss_event_synchronize(void *event);
int dss_stream_wait_event(void *hip_stream, void *event);
/* Page-locked host memory (results that arrive by asynchronous copy).  cached != 0: ordinary cacheable pages, valid to
 * read once the copy's event has completed; 0: coherent pages. */
void *dss_host_alloc(size_t bytes, int cached);
void dss_host_free(void *p);
/* Device -> page-locked host memory (from dss_host_alloc), ordered on hip_stream; both pointers 16-byte aligned.  Runs as a
 * small kernel that stores into the mapped host pages, NOT as a DMA copy: a DMA copy queued behind a long kernel holds up
 * every other copy of the process (the tick's packet upload) until that kernel has finished. */
int dss_memcpy_d2h_async(void *host_dst, const void *d_src, size_t bytes, void *hip_stream);

/* ------------------------------------------------------------------------------------------------
 * Part 1 -- xiph LPCNet decoder symbols, as bound by extensions/lpcnet/cLPCNet.pxd:10-13
 * ---------------------------------------------------------------------------------------------- */
typedef struct LPCNetState LPCNetState;

/* cLPCNet.pxd:10.  Allocates one decoder state (device resident) bound to the process-wide model.
 * The model is the blob given to dss_lpcnet_load_model*, else the file named by $DSS_LPCNET_WEIGHTS.
 * NULL if neither exists or no device is available. */
LPCNetState *lpcnet_create(void);
/* cLPCNet.pxd:11.  Zeroes the state, last_exc = ulaw(0), RNG re-seeded with "LPCNet". Returns 0. */
int lpcnet_init(LPCNetState *st);
/* cLPCNet.pxd:12 */
void lpcnet_destroy(LPCNetState *st);
/* cLPCNet.pxd:13.  One 10 ms frame: features[0..19] (host) -> output[0..N-1] (host), N must be 160
 * (LPCNet.pyx:10,39).  Runs the frame-rate network once, then N autoregressive sample steps. */
void lpcnet_synthesize(LPCNetState *st, const float *features, short *output, int N);
/* xiph lpcnet.h: size of the opaque state (reported for completeness; states live on the device). */
int lpcnet_get_size(void);
/* lpcnet_synthesize has no error channel (void).  It never aborts the process: on failure (N != 160, HIP error,
 * NULL argument) the frame is zero-filled, the reason is left in dss_last_error(), and this process-wide counter is
 * incremented (first failure and every 1000th are also printed to stderr). */
long dss_error_count(void);

/* cLPCNet.pxd:15-19 -- feature encoder (used only by prepare_corpus.py:72-73).  NOT provided: create returns NULL
 * (the reference's LPCFeatureEncoder.__cinit__ raises MemoryError, LPCNet.pyx:53-56), the others return -1 and
 * zero their output. */
typedef struct LPCNetEncState LPCNetEncState;
LPCNetEncState *lpcnet_encoder_create(void);
int lpcnet_encoder_init(LPCNetEncState *st);
void lpcnet_encoder_destroy(LPCNetEncState *st);
int lpcnet_compute_features(LPCNetEncState *st, const short *pcm, float features[4][36]);
int lpcnet_compute_single_frame_features(LPCNetEncState *st, const short *pcm, float features[36]);

/* Weights are data in this build (include/dss_lpcnet_blob.h); xiph compiles them in (nnet_data.c,
 * extensions/lpcnet/setup.py:34-36). */
int dss_lpcnet_load_model(const void *blob, size_t len);
int dss_lpcnet_load_model_file(const char *path);
/* SURVEY.md 8(d) algorithmic bytes per output sample for the loaded model (0 if none). */
double dss_lpcnet_bytes_per_sample(void);
/* Which sample-rate kernel the loaded model runs on, and why (any pointer may be NULL):
 *   fast_path     1 = CU-resident kernel, all weights in VGPRs/LDS; 2 = the same kernel with its extended paths (a model
 *                 with skewed sparsity: some row groups keep z/r blocks beyond their wave's register slots as LDS
 *                 records, or h lists longer than 28 -- a few per cent to tens of per cent slower, same results);
 *                 0 = generic kernel (GRU A blocks streamed from L2, several times slower) because the model exceeds
 *                 an outer capacity below;
 *   zr_slots_max  largest z- or r-gate block count of a row group (register slots: 12 on 16 groups, 8 on the other
 *                 32; outer capacity: 16 more per group);
 *   h_slots_max   largest h-gate block count of a row group (28, or 32 in the extended instantiation, with register-held column ids; outer capacity 64);
 *   h_lds_bytes   LDS image: h-gate blocks, z/r tail blocks and their tables (capacity 151 552 B);
 *   gru_a_order   dss_blob_header.gru_a_order of the model. */
int dss_lpcnet_model_info(int *fast_path, int *zr_slots_max, int *h_slots_max, int *h_lds_bytes, int *gru_a_order);

/* ------------------------------------------------------------------------------------------------
 * Part 2 -- batched decoder: B independent utterances / streams, one persistent workgroup each.
 * Replaces the one-process-per-file pool of local/training.py:165-207 and the per-row Python loop of
 * DelayedLPCNetVocoder.synthesize (local/units.py:531-538).  State persists across calls exactly like
 * a vector of LPCNetState (so it also serves 128 concurrent streams, 1 frame per call).
 * ---------------------------------------------------------------------------------------------- */
typedef struct dss_lpcnet_batch dss_lpcnet_batch;

dss_lpcnet_batch *dss_lpcnet_batch_create(int max_utts, int max_frames);
void dss_lpcnet_batch_destroy(dss_lpcnet_batch *b);
/* A lane: a second launch context on the decoder slots of `parent` (which must not itself be a lane).  It owns the per-call
 * scratch for max_rows x max_frames and nothing else; its calls are ragged calls whose slot list names the parent's slots.
 * Calls on DIFFERENT lanes (and on the parent) may be in flight on different streams at the same time as long as no slot
 * is in two of them at once -- the caller orders a slot's calls (an event between them).  That is the many-stream form
 * of the reference's one vocoder whose state carries from segment to segment (local/units.py:524,531-538): segments that
 * close on different streams are synthesised side by side, a stream's own segments one after the other.  Destroy with
 * dss_lpcnet_batch_destroy; a parent destroyed first is released with its last lane. */
dss_lpcnet_batch *dss_lpcnet_batch_create_lane(dss_lpcnet_batch *parent, int max_rows, int max_frames);
/* lpcnet_init() on every slot (or on slot `utt` only when utt >= 0). */
int dss_lpcnet_batch_reset(dss_lpcnet_batch *b, int utt);
/* Same, enqueued on `hip_stream` without waiting (for device-resident pipelines and timed loops). */
int dss_lpcnet_batch_reset_async(dss_lpcnet_batch *b, int utt, void *hip_stream);
/* Host buffers.  features: [n_utts][n_frames][feat_stride] float32 (feat_stride >= 20, first 20 used,
 * e.g. 36 for xiph .f32 feature files, LPCNet.pyx:97,115).  pcm: [n_utts][n_frames*160] int16. */
int dss_lpcnet_batch_synthesize(dss_lpcnet_batch *b, const float *features, int n_utts, int n_frames,
                                int feat_stride, short *pcm);
/* Device buffers (same shapes), asynchronous on `hip_stream` (a hipStream_t, or NULL for the default
 * stream).  Nothing is copied to or from the host. */
int dss_lpcnet_batch_synthesize_dev(dss_lpcnet_batch *b, const float *d_features, int n_utts, int n_frames,
                                    int feat_stride, short *d_pcm, void *hip_stream);
/* Ragged / slot-indexed form.  Row i of the call (features [n_utts][n_frames][feat_stride], pcm
 * [n_utts][n_frames*160]) continues the decoder state of slot slots[i] (NULL: slot i) and synthesizes
 * only its first counts[i] <= n_frames frames (NULL: n_frames); its workgroup then retires, pcm beyond
 * counts[i]*160 is left untouched and a count of 0 leaves the slot exactly as it was.  `slots` and
 * `counts` are HOST arrays of n_utts ints, validated (range; a slot may appear once per call, a decoder is
 * sequential) and uploaded on the stream.  This is the shape of the reference's real callers: .npy files
 * of different lengths with a fresh decoder each (local/training.py:182-198), and speech segments of
 * different lengths finishing on some of the 128 streams whose vocoder state carries across segments
 * (local/units.py:524,531-538).  With counts, workgroups take the rows by decreasing frame count whatever order the
 * caller used (the library sorts a dispatch list; results do not depend on it), so that short rows fill in behind the
 * long ones when n_utts exceeds the number of CUs.  Calls on one batch object must be issued on one stream at a time. */
int dss_lpcnet_batch_synthesize_ragged_dev(dss_lpcnet_batch *b, const float *d_features, const int *slots,
                                           const int *counts, int n_utts, int n_frames, int feat_stride,
                                           short *d_pcm, void *hip_stream);
int dss_lpcnet_batch_synthesize_ragged(dss_lpcnet_batch *b, const float *features, const int *slots,
                                       const int *counts, int n_utts, int n_frames, int feat_stride, short *pcm);
/* Kernel choice (uniform and ragged calls).  0 (default): one utterance per workgroup (csrc/lpcnet_sample.hip) while the
 * call has at most one row per CU, two utterances per workgroup -- carried as the two halves of packed fp32 instructions,
 * csrc/lpcnet_sample_pair.hip -- beyond (a uniform call is split: full rounds of two rows per CU on the pair kernel, a
 * remainder of at most one row per CU as one round of the one-utterance kernel); 1 or -1: always one per workgroup; 2:
 * always two (fails with DSS_EINVAL for a
 * model whose CU-resident layout leaves no room for the second utterance).  In a ragged call the two rows of a workgroup
 * are neighbours in the dispatch list (near-equal length); they run packed over the frames both have and the longer one
 * finishes alone.  Models on the extended / generic paths always run one utterance per workgroup.  Results are
 * bit-identical either way, and a decoder state written by one form is continued by the other. */
int dss_lpcnet_batch_set_multi(dss_lpcnet_batch *b, int utterances_per_workgroup);
/* Test taps (device -> host): frame-rate network outputs of the LAST call, per utterance and frame:
 * which = 0: gru_a_condition [n_frames][3*gru_a]; 1: gru_b_condition [n_frames][3*gru_b]; 2: lpc [n_frames][16].
 * which = 3: per-sample excitation index (uint8 stored as float) [n_frames*160]; 4: pre-de-emphasis pcm float.
 * (3 and 4 need dss_lpcnet_batch_enable_trace(b, 1) before the call.) */
int dss_lpcnet_batch_tap(dss_lpcnet_batch *b, int utt, int which, float *out, size_t n_floats);
/* on: 0 = off, 1 = trace on the kernel the model selects, 17 = trace on the generic kernel. */
int dss_lpcnet_batch_enable_trace(dss_lpcnet_batch *b, int on);
/* Teacher forcing (test instrument; needs trace enabled): in the following calls sample k of row u takes the excitation
 * index exc[u*n_frames*160 + k] (host array) instead of the sampled one -- the RNG advances as usual -- and tap 5
 * returns the pre-threshold logits of all 255 tree nodes per sample, [n_frames*160][256] ([.][0] unused).  The calls
 * must have this n_frames.  exc == NULL switches back to free running. */
int dss_lpcnet_batch_force_excitation(dss_lpcnet_batch *b, const unsigned char *exc, int n_utts, int n_frames);
/* Self-test of the only transcendental evaluated on the device on this path: out[i] = (float)(pow(10.0, x[i]) *
 * comp[i]), the expression of freq.c lpc_from_cepstrum (host buffers).  See DESIGN.md section 2. */
int dss_selftest_exp10(const float *x, const float *comp, float *out, long n);
/* Self-test of the device's lin2ulaw (xiph common.h; evaluated in a shorter instruction sequence):
 * out[i] = lin2ulaw(x) for the fp32 x whose bit pattern is start_bits + i * stride (wrapping), i < n (host buffer). */
int dss_selftest_lin2ulaw(unsigned start_bits, unsigned stride, long n, unsigned char *out);
/* Host-only (no GPU): lays the model out for the CU-resident sample kernel and walks every lane's z, r and h block lists
 * through that layout as the kernel indexes it.  info[8]: fast_path (0/1/2 as in dss_lpcnet_model_info), zr blocks max,
 * h blocks max, LDS bytes, register slots per gate on waves 4-5, tail blocks, mismatching rows, out-of-range reads. */
int dss_selftest_fast_layout(const void *blob, size_t len, int *info);
/* Average device time (ms) of the sample-rate kernel over the calls since the last query, measured with
 * HIP events on the stream the kernel was launched on; resets the accumulator.  Needs
 * dss_lpcnet_batch_enable_timing(b, 1). */
int dss_lpcnet_batch_enable_timing(dss_lpcnet_batch *b, int on);
double dss_lpcnet_batch_kernel_ms(dss_lpcnet_batch *b, int which /*0 = sample kernel, 1 = frame kernels*/);

/* ------------------------------------------------------------------------------------------------
 * Part 3 -- HGA: IIR cascade + warm-start frame buffer + log power, float64
 * ---------------------------------------------------------------------------------------------- */
/* compute_log_power_features(data, sr, window_length, window_shift)  hga_optimized.pyx:27-47.
 * data: host (T, C) float64 row-major.  out: host (W, C), W = dss_hga_num_windows(T, ...).
 * Windowed mean power runs on the device; the final log() is applied by the host libm while copying
 * out, which is what makes the result bit-identical to the reference on any host. */
int dss_hga_num_windows(int T, int sr, float window_length, float window_shift);
int dss_hga_log_power(const double *data, int T, int C, int sr, float window_length, float window_shift,
                      double *out);

/* Stateful extractor for n_streams independent streams of n_channels each: the GPU counterpart of
 * HighGammaExtractor (local/units.py:97-161) without the Python pre/post transforms.
 * sos_hg / sos_fh: (n_sections, 6) band-pass and band-stop second-order sections (units.py:124-126);
 * zi_hg / zi_fh: (n_sections, 2) scipy.signal.sosfilt_zi(sos), replicated over channels as
 * units.py:128-132 does.  Holds per-channel filter state and the last `overlap` filtered rows
 * (WarmStartFrameBuffer, hga_optimized.pyx:50-131) on the device. */
typedef struct dss_hga dss_hga;
dss_hga *dss_hga_create(int n_streams, int n_channels, int fs, float window_length, float window_shift,
                        int n_sections, const double *sos_hg, const double *sos_fh,
                        const double *zi_hg, const double *zi_fh);
void dss_hga_destroy(dss_hga *h);
int dss_hga_reset(dss_hga *h);
/* Frames the next extract call with n new samples per stream will emit. */
int dss_hga_frames_for(const dss_hga *h, int n);
/* extract_features (units.py:145-161): data host (n_streams, n, C) float64 -> out host (n_streams, W, C).
 * Returns W (>= 0) or a negative error.  All streams advance by the same n. */
int dss_hga_extract(dss_hga *h, const double *data, int n, double *out);
/* Fused front end: the pre-transforms decode_online.py:65-97 puts in front of the filters --
 * column reorder (local/common.py:31-32), per-grid common average referencing with excluded channels
 * (common.py:338-345) and channel selection (common.py:54-55) -- collapse to: output channel c =
 * raw[src_col[c]] - mean_g(c), where mean_g is the SEQUENTIAL sum (numpy reduces the fancy-indexed, Fortran-ordered
 * view one column at a time) of raw[comp_cols[comp_off[g] .. comp_off[g+1])] divided by their count, g = grid_of[c]
 * (-1: no referencing).  After this call dss_hga_extract_raw* take raw amplifier packets (n, c_raw). */
int dss_hga_set_frontend(dss_hga *h, int c_raw, const int *src_col, const int *grid_of, int n_grids,
                         const int *comp_cols, const int *comp_off);
int dss_hga_extract_raw(dss_hga *h, const double *raw, int n, double *out);
int dss_hga_extract_raw_dev(dss_hga *h, const double *d_raw, int n, double *d_out, int apply_log, void *hip_stream);
/* Device-resident form: d_data / d_out device pointers; log applied on the device (OCML log, <= 1 ulp
 * from the host's) when apply_log != 0, else out = mean power + 0.01. */
int dss_hga_extract_dev(dss_hga *h, const double *d_data, int n, double *d_out, int apply_log, void *hip_stream);
/* The same from payloads in WIRE format: d_payload (n_streams, c_in, n) float32, channel-major -- the body of the amplifier's
 * packets behind their 7-byte header (local/units.py:78-82: '=BBB HH' + float32[n_channels x n_samples];
 * development_amplifier.py:14-25), c_in = c_raw when a front end is configured, else n_channels.  One more small launch does on
 * the device what ZMQConnector.interpret_bytes does per packet on the host (reshape, transpose, astype(float64)); float32 ->
 * float64 is exact, the frames are those of dss_hga_extract_raw_dev / dss_hga_extract_dev on the converted rows, bit for bit.
 * Half the bytes cross the bus and no stream's packet is touched by the host. */
int dss_hga_extract_wire_dev(dss_hga *h, const float *d_payload, int n, double *d_out, int apply_log, void *hip_stream);
/* Optional last step of the reference's feature chain inside the extractor's launch: ZScoreNormalization,
 * (frame - means[c]) / stds[c] (local/common.py:367-376; decode_online.py:88-97 puts it behind HighGammaActivity as a
 * post-transform).  means / stds: host arrays of n_channels doubles, both NULL to clear.  The device-resident entry
 * points then return z-scored frames (hga_fused_kernel's epilogue, or hga_window_kernel's in the three-launch form); the
 * host-buffer ones apply it on the host after their host-libm log. */
int dss_hga_set_zscore(dss_hga *h, const double *means, const double *stds);
/* Tests and A/B timing only: which kernel form serves this extractor.  0 = choose (default: hga_fused_kernel; three
 * launches when its ring does not fit LDS), 1 = hga_fused_kernel, 2 = the three-launch form. */
int dss_selftest_hga_force_path(dss_hga *h, int path);

/* ------------------------------------------------------------------------------------------------
 * Part 4 -- speech-segment gate for n_streams streams: the two ring buffers the
 * reference chains behind its neural VAD in FilterSpeechSegments.process (local/units.py:432-447):
 * VoiceActivityDetectionSmoothing (local/common.py:106-153; window 2*smoothing_context+1 <= 64 labels,
 * a frame is speech when the proportion of raw speech labels in the window >= proportion_threshold) and
 * SpeechSegmentHistory (local/common.py:156-215; ring of buffer_size float32 frames, a segment is the
 * speech run plus `context` frames on both sides, emitted on the context-th non-speech frame after it).
 * The VAD network itself stays on PyTorch-ROCm (north_star); its per-frame decisions come in as int32.
 * decode_online.py:115-121 uses smoothing_context 5 (units.py:416), threshold 0.6, buffer 2000, context 50.
 * ---------------------------------------------------------------------------------------------- */
typedef struct dss_gate dss_gate;
dss_gate *dss_gate_create(int n_streams, int nb_features, int smoothing_context, double proportion_threshold,
                          int buffer_size, int context, int max_frames /* per push */);
void dss_gate_destroy(dss_gate *g);
int dss_gate_reset(dss_gate *g, int stream /* -1: all */);
/* Segments one stream can complete within one push of max_frames frames (E below). */
int dss_gate_max_events(const dss_gate *g);
/* All streams advance by n_frames: frames (n_streams, n_frames, nb_features) float64 (msg.data as the unit
 * receives it; stored as float32 like the numpy rings), labels (n_streams, n_frames) int32, nonzero = the
 * VAD's argmax said speech.  events (HOST out, n_streams x (2+E) ints): per stream [number of segments
 * completed in this push, number of frames the smoothing labelled speech in this push (units.py:445 needs
 * it for previous_frames), length of segment 0, ...].  Returns the total number of completed segments
 * (>= 0) or a negative error.  _dev: device pointers, enqueued on hip_stream, which is synchronised
 * before returning (the caller needs the event counts to go on). */
int dss_gate_push(dss_gate *g, const double *frames, const int *labels, int n_frames, int *events);
int dss_gate_push_dev(dss_gate *g, const double *d_frames, const int *d_labels, int n_frames, int *events,
                      void *hip_stream);
/* Copy segment `event` that `stream` completed in the LAST push: (length, nb_features) float32, at most
 * cap_frames rows.  Returns its length. */
int dss_gate_segment(dss_gate *g, int stream, int event, float *dst, int cap_frames);
int dss_gate_segment_dev(dss_gate *g, int stream, int event, float *d_dst, int cap_frames, void *hip_stream);
/* The same for n segments of the LAST push in one launch: segment (streams[i], events[i]) goes to row dst_rows[i] of d_dst, a
 * device buffer of row_frames frames per row ((rows, row_frames, nb_features) float32: a pool of segment buffers that outlive
 * the next push).  streams / events / dst_rows are HOST arrays.  Returns n. */
int dss_gate_collect_dev(dss_gate *g, int n, const int *streams, const int *events, const int *dst_rows, float *d_dst,
                         int row_frames, void *hip_stream);
/* Frames this stream has been pushed since the last reset (FilterSpeechSegments' frame_counter). */
int dss_gate_frames_seen(dss_gate *g, int stream);

/* ------------------------------------------------------------------------------------------------
 * Part 5 -- the neural voice-activity detector in front of the gate, for n_streams streams:
 * UnidirectionalVoiceActivityDetector (local/models.py:11-33: LSTM(n_inputs -> H) -> LSTM(H -> H) -> Linear(H -> 2)) as
 * FilterSpeechSegments.process calls it (local/units.py:432-434): every frame of a packet, (h, c) of both layers carried
 * across packets, label = argmax of the two logits.  One launch per call (csrc/vad_lstm.hip).  The reference's arithmetic
 * here is torch.nn.LSTM's: results agree with it to ~1e-6 on the logits (tested at 2e-5), not bit for bit.
 * decode_online.py:115-121 builds the model with 2 layers x 150 hidden units over 64 high-gamma features.
 * ---------------------------------------------------------------------------------------------- */
typedef struct dss_vad dss_vad;
dss_vad *dss_vad_create(int n_streams, int n_inputs, int hidden_units /* <= 160 */);
void dss_vad_destroy(dss_vad *v);
/* Host arrays in torch.nn.LSTM's own layout (state_dict of the reference class, gate order i, f, g, o):
 * lstm.weight_ih_l0 [4H][n_inputs], lstm.weight_hh_l0 [4H][H], lstm.bias_ih_l0 / bias_hh_l0 [4H], the same four for l1
 * ([4H][H]), classifier.weight [2][H], classifier.bias [2]. */
int dss_vad_load_weights(dss_vad *v, const float *w_ih0, const float *w_hh0, const float *b_ih0, const float *b_hh0,
                         const float *w_ih1, const float *w_hh1, const float *b_ih1, const float *b_hh1,
                         const float *cls_w, const float *cls_b);
/* Zero state (create_new_initial_state, models.py:22-24) of one stream, or of all (stream < 0).  dss_vad_reset runs on the
 * null stream and waits: it is ordered against steps issued on a blocking stream only.  dss_vad_reset_async is enqueued on
 * `hip_stream` -- pass the stream the steps run on. */
int dss_vad_reset(dss_vad *v, int stream);
int dss_vad_reset_async(dss_vad *v, int stream, void *hip_stream);
/* All streams advance by n_frames.  Device pointers, enqueued on hip_stream: d_frames (n_streams, n_frames, n_inputs)
 * float64 (frames_are_f64 != 0: as dss_hga_extract_dev returns them; cast to float32 like units.py:433) or float32;
 * d_labels (n_streams, n_frames) int32, 1 = speech -- what dss_gate_push_dev takes; d_logits (n_streams, n_frames, 2)
 * float32 or NULL. */
int dss_vad_step_dev(dss_vad *v, const void *d_frames, int frames_are_f64, int n_frames, int *d_labels, float *d_logits,
                     void *hip_stream);
/* Host copies of the recurrent state, [2 layers][n_streams][H] each, either may be NULL; set == 0 reads, else writes. */
int dss_vad_state(dss_vad *v, float *h, float *c, int set);

/* ------------------------------------------------------------------------------------------------
 * Part 6 -- the bidirectional recurrent decoder between the extractor and the vocoder:
 * BidirectionalSpeechSynthesisModel (local/models.py:36-58: LSTM(n_inputs -> H, 2 layers, bidirectional) ->
 * Linear(2H -> 20)) as DecodingModel.process calls it (local/units.py:499-508): all frames of a segment (or of a
 * packet, in the chunk-wise streaming mode), zero initial state per call, frames cast to float32.  Three launches per
 * call (csrc/bilstm_decoder.hip: one per layer with both directions side by side, one for the regressor) instead of
 * MIOpen's chain of ~20.  The reference's arithmetic here is torch.nn.LSTM's: results agree with it to ~1e-6 on the
 * features (tested at 2e-5 against the reference-generated golden vector), not bit for bit.  A decoder of another
 * architecture stays a PyTorch-ROCm module (dss_amd/pipeline.py falls back to it).
 * ---------------------------------------------------------------------------------------------- */
typedef struct dss_dec dss_dec;
dss_dec *dss_dec_create(int max_streams, int max_frames, int n_inputs /* <= 256 */, int hidden_units /* <= 128 */, int n_outputs /* <= 32 */);
void dss_dec_destroy(dss_dec *v);
/* w: 18 host arrays in torch.nn.LSTM's own layout (state_dict of the reference class, gate order i, f, g, o):
 * for layer l in (0, 1), for (forward, reverse): lstm.weight_ih_l{l}[_reverse] [4H][Cin], lstm.weight_hh_l{l}[_reverse] [4H][H],
 * lstm.bias_ih_l{l}[_reverse] [4H], lstm.bias_hh_l{l}[_reverse] [4H] (Cin = n_inputs for l = 0, 2H for l = 1); then
 * regressor.weight [n_outputs][2H], regressor.bias [n_outputs]. */
int dss_dec_load_weights(dss_dec *v, const float *const *w);
/* Device pointers, enqueued on hip_stream: d_frames (n_streams, n_frames, n_inputs) float64 (frames_are_f64 != 0: as
 * dss_hga_extract_dev returns them) or float32; d_feats (n_streams, n_frames, n_outputs) float32 -- what
 * dss_lpcnet_batch_synthesize_dev takes. */
int dss_dec_forward_dev(dss_dec *v, const void *d_frames, int frames_are_f64, int n_streams, int n_frames, float *d_feats,
                        void *hip_stream);
/* Ragged form -- the segments that closed on one tick, each decoded as a whole from a fresh state (units.py:499-508), in one
 * call: stream i has counts[i] <= n_frames frames, read from row in_rows[i] (NULL: i) of d_frames, a buffer of row_frames >=
 * n_frames frames per row (what dss_gate_collect_dev fills); its backward direction starts at its OWN last frame.  counts /
 * in_rows are HOST arrays.  d_feats is (n_streams, n_frames, n_outputs); rows beyond counts[i] are left untouched. */
int dss_dec_forward_rows_dev(dss_dec *v, const void *d_frames, int frames_are_f64, int row_frames, const int *in_rows,
                             const int *counts, int n_streams, int n_frames, float *d_feats, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* DSS_HIP_H */
