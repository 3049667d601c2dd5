// csrc/speech_gate.hip -- speech-segment gating for many streams on gfx950.
//
// Restates, for S independent streams advanced in lock-step, the two ring-buffer classes the reference chains
// behind its neural VAD (FilterSpeechSegments.process, local/units.py:432-447):
//   VoiceActivityDetectionSmoothing.insert      local/common.py:125-147   (majority vote over 2*ctx+1 raw labels,
//                                                                          frames leave 2*ctx inserts late)
//   SpeechSegmentHistory.insert                 local/common.py:184-215   (ring of frames; a segment = the speech run
//                                                                          plus `context` frames on both sides)
// All of it is byte/index work on float32 copies of the frames: results are bit-identical to the numpy classes,
// including their modulo arithmetic when a segment is longer than the ring.
//
// Mapping: one workgroup of four waves per stream.  Wave 0 runs the frame loop: a lane owns feature columns c, c+64, ...
// of every ring, so a lane only ever reads back what it wrote itself and the loop needs no barrier.  The scalar
// bookkeeping (pointers, counters, the label window as a 64-bit mask) is computed redundantly by every thread, so all four
// waves agree on when a segment closes; its rows are then copied out by all 256 threads, four loads in flight per thread
// (one wave copying row after row, each load waiting for the store before it, took ~0.3 ms for a 3.5-s segment -- on the
// tick's critical path).  Time is the serial axis (one tick has a handful of frames); streams x features are the parallel
// ones; every access is a coalesced row segment.
#include "dss_common.h"

__device__ __forceinline__ int gate_mod(int x, int n)       // Python's % for a positive modulus
{
    const int r = x % n;
    return r < 0 ? r + n : r;
}

#define GATE_THREADS 256

__global__ void __launch_bounds__(GATE_THREADS)
speech_gate_kernel(DssGateDev g, const double *__restrict__ frames, const int *__restrict__ labels, int W)
{
    const int s = blockIdx.x, tid = threadIdx.x;
    const bool w0 = tid < 64;                                 // the wave that owns the rings' columns
    const int C = g.C, SM = g.sm_size, N = g.hist_size;
    int *st = g.state + (size_t)s * DSS_GATE_STATE_INTS;
    int sm_w = st[0], sm_r = st[1], h_w = st[2], speech = st[3], future = st[4];
    unsigned long long mask = ((unsigned long long)(unsigned)st[6] << 32) | (unsigned)st[5];
    float *sm = g.sm_buf + (size_t)s * SM * C;
    float *hist = g.hist + (size_t)s * N * C;
    int n_events = 0, n_speech = 0;
    int *ev = g.events + (size_t)s * (2 + g.max_events);

    for (int i = 0; i < W; ++i) {
        // ---- smoothing: raw label and frame enter at the write pointer (common.py:130-136)
        const bool raw = labels[(size_t)s * W + i] != 0;
        if (raw) mask |= 1ull << sm_w; else mask &= ~(1ull << sm_w);
        const double *src = frames + ((size_t)s * W + i) * C;
        if (w0) for (int c = tid; c < C; c += 64) sm[(size_t)sm_w * C + c] = (float)src[c];   // float32 ring, as numpy casts
        // ratio >= threshold in float64, as Python evaluates it (common.py:139-140)
        const bool lab = ((double)__popcll(mask) / (double)SM) >= g.threshold;
        // ---- history: the frame leaving the smoothing window enters the segment ring (common.py:141, 196-199)
        if (w0) for (int c = tid; c < C; c += 64) hist[(size_t)h_w * C + c] = sm[(size_t)sm_r * C + c];
        sm_w = sm_w + 1 == SM ? 0 : sm_w + 1;
        sm_r = sm_r + 1 == SM ? 0 : sm_r + 1;
        h_w = h_w + 1 == N ? 0 : h_w + 1;
        if (lab) {
            ++speech;
            ++n_speech;
        } else if (speech > 0) {
            ++future;
            if (future >= g.hist_ctx) {                                                       // common.py:205-214
                const int stop = g.hist_ctx > 0 ? h_w : gate_mod(h_w - 1, N);
                const int start = gate_mod(stop - 2 * g.hist_ctx - speech, N);
                const int count = gate_mod(stop - start, N);
                if (n_events < g.max_events) {                                                // (uniform over the workgroup)
                    float *__restrict__ dst = g.seg_out + ((size_t)s * g.max_events + n_events) * (size_t)N * C;
                    __syncthreads();                          // wave 0's ring rows are visible to the copying waves
                    // element k of the segment = ring row (start + k / C) mod N, column k mod C; a thread takes k = tid,
                    // tid + 256, ...: (row, col) advance by (256 / C, 256 mod C) with a carry
                    const int dr = GATE_THREADS / C, dc = GATE_THREADS - dr * C;
                    int row = tid / C, col = tid - row * C;
                    const long total = (long)count * C;
                    for (long k = tid; k < total; k += 4 * GATE_THREADS) {
                        float v[4];
                        int r4[4], c4[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            r4[u] = row; c4[u] = col;
                            if (k + (long)u * GATE_THREADS < total) {
                                int p = start + row;
                                p = p >= N ? p - N : p;       // start < N, row < count <= N
                                v[u] = hist[(size_t)p * C + col];
                            }
                            col += dc; row += dr;
                            if (col >= C) { col -= C; ++row; }
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (k + (long)u * GATE_THREADS < total) dst[(size_t)r4[u] * C + c4[u]] = v[u];
                    }
                    __syncthreads();                          // the ring rows may be overwritten by the frames that follow
                    if (tid == 0) ev[2 + n_events] = count;
                }
                ++n_events;                                   // > max_events is reported, the host turns it into an error
                speech = 0;
                future = 0;
            }
        }
    }
    if (tid == 0) {
        st[0] = sm_w; st[1] = sm_r; st[2] = h_w; st[3] = speech; st[4] = future;
        st[5] = (int)(unsigned)(mask & 0xffffffffull); st[6] = (int)(unsigned)(mask >> 32);
        st[7] += W;                                           // frames seen so far (FilterSpeechSegments.frame_counter)
        ev[0] = n_events;
        ev[1] = n_speech;
    }
}

__global__ void speech_gate_reset_kernel(DssGateDev g, int only)
{
    const int s = only >= 0 ? only : blockIdx.x;
    const int tid = threadIdx.x;
    const size_t nsm = (size_t)g.sm_size * g.C, nh = (size_t)g.hist_size * g.C;
    for (size_t k = tid; k < nsm; k += blockDim.x) g.sm_buf[(size_t)s * nsm + k] = 0.f;
    for (size_t k = tid; k < nh; k += blockDim.x) g.hist[(size_t)s * nh + k] = 0.f;
    if (tid == 0) {
        int *st = g.state + (size_t)s * DSS_GATE_STATE_INTS;
        st[0] = 2 * g.sm_ctx;       // write pointer starts one full window ahead of the read pointer (common.py:122-123)
        for (int k = 1; k < DSS_GATE_STATE_INTS; ++k) st[k] = 0;
        int *ev = g.events + (size_t)s * (2 + g.max_events);
        for (int k = 0; k < 2 + g.max_events; ++k) ev[k] = 0;
    }
}

// Completed segments of the last push -> rows of a caller's buffer, one launch for up to DSS_GATE_COLLECT_MAX of them.
__global__ void __launch_bounds__(256)
gate_collect_kernel(DssGateDev g, DssGateCollect a, float *__restrict__ dst, long dst_row_floats)
{
    const int i = blockIdx.x;
    const int len = g.events[(size_t)a.stream[i] * (2 + g.max_events) + 2 + a.event[i]];
    const float *__restrict__ src = g.seg_out + ((size_t)a.stream[i] * g.max_events + a.event[i]) * (size_t)g.hist_size * g.C;
    float *__restrict__ out = dst + (size_t)a.dst_row[i] * dst_row_floats;
    const long total = (long)len * g.C;
    for (long k = (long)blockIdx.y * 256 + threadIdx.x; k < total; k += (long)gridDim.y * 256) out[k] = src[k];
}

int dss_launch_gate_collect(const DssGateDev &g, const DssGateCollect &a, int n, float *d_dst, long dst_row_floats, hipStream_t s)
{
    hipLaunchKernelGGL(gate_collect_kernel, dim3(n, 16), dim3(256), 0, s, g, a, d_dst, dst_row_floats);
    DSS_HIP_CHECK(hipGetLastError());
    return DSS_OK;
}

int dss_launch_gate(const DssGateDev &g, const double *d_frames, const int *d_labels, int W, hipStream_t s)
{
    hipLaunchKernelGGL(speech_gate_kernel, dim3(g.S), dim3(GATE_THREADS), 0, s, g, d_frames, d_labels, W);
    DSS_HIP_CHECK(hipGetLastError());
    return DSS_OK;
}

int dss_launch_gate_reset(const DssGateDev &g, int stream, hipStream_t s)
{
    hipLaunchKernelGGL(speech_gate_reset_kernel, dim3(stream >= 0 ? 1 : g.S), dim3(256), 0, s, g, stream);
    DSS_HIP_CHECK(hipGetLastError());
    return DSS_OK;
}
