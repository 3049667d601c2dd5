// csrc/bilstm_decoder.hip -- the bidirectional recurrent decoder of the online path for many streams per launch (gfx950).
//
// Restates, for S independent streams and the T high-gamma frames of a call (one amplifier packet = 4 frames in the streaming
// mode, a whole segment otherwise),
//   BidirectionalSpeechSynthesisModel.forward       local/models.py:36-58   LSTM(C -> H, 2 layers, bidirectional) -> Linear(2H -> 20)
//   DecodingModel.process, the model call           local/units.py:499-508  frames as float32, a fresh zero state per call
// The reference runs torch.nn.LSTM; its arithmetic is torch's, not a fixed C sequence, so parity here is tolerance-level like
// the detector's (vad_lstm.hip): the test states it (|feature - torch| <= 2e-5 on the reference-generated golden vector) and
// fused multiply-adds are allowed.  Gate order i, f, g, o (torch.nn.LSTM); a layer's input at frame t is
// [h_forward(t), h_backward(t)] of the layer below.
//
// Three launches per call instead of MIOpen's ~20: one per layer -- its two directions are independent and run as separate
// workgroups (blockIdx.y) -- and the regressor.  A 512-thread workgroup owns W streams (1, 2 or 4: as few as keeps every CU busy) and one direction for all T steps;
// thread t owns gate row t (4H <= 512 rows) and runs the row's dot product for the W streams at once: the weights
// (copies with four consecutive inputs of a row side by side: one 16-byte load per lane, 1 KB of consecutive bytes per wave)
// of the input half come from L2 once per workgroup and chunk of DEC_TP steps, the row of W_hh stays in the thread's registers
// for the whole call, the inputs come from LDS as broadcast reads; h lives in LDS, c in the registers of the thread that owns
// (stream, unit).  Time steps are sequential; streams x gate rows x directions are the parallel axes.
#include "dss_common.h"

#define DEC_THREADS 512           // >= 4 * H and >= W * H
#define DEC_MAXH DSS_DEC_MAXH              // (a multiple of 4)
#define DEC_MAXC DSS_DEC_MAXC              // inputs of a layer: n_inputs for layer 0, 2H above it
#define DEC_TP 4                  // steps whose input halves (W_ih x) are formed in one pass over W_ih

typedef float df4 __attribute__((ext_vector_type(4)));
template <int W> struct DecVec { typedef float type __attribute__((ext_vector_type(W))); };
template <> struct DecVec<1> { struct type { float v; __device__ float &operator[](int) { return v; } __device__ const float &operator[](int) const { return v; } }; };

__device__ __forceinline__ float dec_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

// Gate rows times inputs: wq is [n / 4][4H][4] (four consecutive inputs of a row side by side); n a multiple of 4 (the host pads with
// zero weights, the kernel keeps the padded inputs at zero); x is [input][W streams] in LDS.
// the input halves of DEC_TP steps' gate rows at once: one pass over W_ih (n / 4 sixteen-byte loads per thread) serves DEC_TP
// steps.  acc[tt] accumulates exactly the terms, in exactly the order, dec_dot would give step tt.
template <int W, typename V>
__device__ __forceinline__ void dec_dot_steps(V (&acc)[DEC_TP], const float *__restrict__ wq, int H4, int row, const V (*x)[DEC_MAXC], int n)
{
    const df4 *wr = reinterpret_cast<const df4 *>(wq) + row;
    int q = 0;
    for (; q + 4 <= n / 4; q += 4) {
        df4 w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = wr[(size_t)(q + u) * H4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int tt = 0; tt < DEC_TP; ++tt) {
                const V x0 = x[tt][4 * (q + u)], x1 = x[tt][4 * (q + u) + 1], x2 = x[tt][4 * (q + u) + 2], x3 = x[tt][4 * (q + u) + 3];
#pragma unroll
                for (int s = 0; s < W; ++s) {
                    acc[tt][s] = __builtin_fmaf(w[u].x, x0[s], acc[tt][s]);
                    acc[tt][s] = __builtin_fmaf(w[u].y, x1[s], acc[tt][s]);
                    acc[tt][s] = __builtin_fmaf(w[u].z, x2[s], acc[tt][s]);
                    acc[tt][s] = __builtin_fmaf(w[u].w, x3[s], acc[tt][s]);
                }
            }
        }
    }
    for (; q < n / 4; ++q) {
        const df4 w = wr[(size_t)q * H4];
#pragma unroll
        for (int tt = 0; tt < DEC_TP; ++tt) {
            const V x0 = x[tt][4 * q], x1 = x[tt][4 * q + 1], x2 = x[tt][4 * q + 2], x3 = x[tt][4 * q + 3];
#pragma unroll
            for (int s = 0; s < W; ++s) {
                acc[tt][s] = __builtin_fmaf(w.x, x0[s], acc[tt][s]);
                acc[tt][s] = __builtin_fmaf(w.y, x1[s], acc[tt][s]);
                acc[tt][s] = __builtin_fmaf(w.z, x2[s], acc[tt][s]);
                acc[tt][s] = __builtin_fmaf(w.w, x3[s], acc[tt][s]);
            }
        }
    }
}

// One layer, both directions (blockIdx.y): in (S, T, Cin) -> out (S, T, 2H), forward h in [0, H), backward in [H, 2H).
// Ragged form (segments of different lengths in one call, dss_dec_forward_rows_dev): stream s has counts[s] <= T frames (NULL:
// T) -- its backward direction starts at its OWN last frame, and nothing is computed or written beyond it -- and its input
// frames are row in_row[s] (NULL: s) of a buffer with Tin frames per row (a pool of segment buffers).
template <typename InT, int W>
__global__ void __launch_bounds__(DEC_THREADS)
bilstm_layer_kernel(const InT *__restrict__ in, int S, int T, int Cin, int H, const float *__restrict__ wT_f,
                    const float *__restrict__ wT_b, const float *__restrict__ b_f, const float *__restrict__ b_b,
                    float *__restrict__ out, const int *__restrict__ counts, const int *__restrict__ in_row, int Tin)
{
    typedef typename DecVec<W>::type V;
    __shared__ __attribute__((aligned(16))) V xin[DEC_TP][DEC_MAXC];      // [step of the chunk][input][stream of this workgroup]
    __shared__ __attribute__((aligned(16))) V hs[DEC_MAXH];               // [unit][stream]; units H .. Hp-1 stay zero
    __shared__ __attribute__((aligned(16))) V gates[4 * DEC_MAXH];        // [gate row][stream]
    const int tid = threadIdx.x, H4 = 4 * H, dir = blockIdx.y;
    const int Cp = (Cin + 3) & ~3, Hp = (H + 3) & ~3;      // the padded input counts the weight copies were built for
    const int s0 = blockIdx.x * W;
    const float *wT = dir ? wT_b : wT_f;
    const int cs = tid / H, cu = tid - cs * H;             // the (stream, unit) this thread owns in the cell updates
    const bool cell = tid < W * H && s0 + cs < S;
    const int Tc = cell ? (counts ? min(counts[s0 + cs], T) : T) : 0;      // frames of the stream this thread's cell belongs to
    __shared__ int Tsh[W], Rsh[W];                         // per stream of this workgroup: frames, input row
    if (tid < W) {
        const bool live = s0 + tid < S;
        Tsh[tid] = live ? (counts ? min(counts[s0 + tid], T) : T) : 0;
        Rsh[tid] = live ? (in_row ? in_row[s0 + tid] : s0 + tid) : 0;
    }
    float c = 0.f;                                         // create_new_initial_state: zeros (models.py:22-24)
    for (int k = tid; k < DEC_MAXH * W; k += DEC_THREADS) reinterpret_cast<float *>(hs)[k] = 0.f;
    for (int k = tid; k < DEC_TP * DEC_MAXC * W; k += DEC_THREADS) reinterpret_cast<float *>(xin)[k] = 0.f;
    const bool rowt = tid < H4;
    const float bias = rowt ? (dir ? b_b : b_f)[tid] : 0.f;
    // this thread's row of W_hh stays in its registers for all steps (DEC_MAXH values, zero beyond H): a step's recurrent half
    // then costs LDS reads of h and arithmetic only -- with the row streamed from L2 every step, L2 latency set the step time
    df4 whh[DEC_MAXH / 4];
    {
        const df4 *wr = reinterpret_cast<const df4 *>(wT + (size_t)Cp * H4) + (rowt ? tid : 0);
#pragma unroll
        for (int q = 0; q < DEC_MAXH / 4; ++q) whh[q] = (rowt && 4 * q < Hp) ? wr[(size_t)q * H4] : (df4){0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    int Tw = 0;                                            // the longest of this workgroup's streams
#pragma unroll
    for (int sl = 0; sl < W; ++sl) Tw = max(Tw, Tsh[sl]);
    // The recurrence is serial in time, the input halves of the gates are not: per chunk of DEC_TP steps one pass over W_ih forms
    // them for all its steps (each gets the same terms in the same order as a step on its own), and a step then adds only W_hh h.
    for (int step0 = 0; step0 < Tw; step0 += DEC_TP) {
        const int nst = min(DEC_TP, Tw - step0);
        for (int idx = tid; idx < nst * Cin * W; idx += DEC_THREADS) {
            const int tt = idx / (Cin * W), rem = idx - tt * (Cin * W);
            const int sl = rem / Cin, k = rem - sl * Cin;
            const int Ts = Tsh[sl], step = step0 + tt;
            const int t = dir ? Ts - 1 - step : step;
            reinterpret_cast<float *>(&xin[tt][k])[sl] = step < Ts ? (float)in[((size_t)Rsh[sl] * Tin + t) * Cin + k] : 0.f;
        }
        __syncthreads();
        V pre[DEC_TP];
#pragma unroll
        for (int tt = 0; tt < DEC_TP; ++tt)
#pragma unroll
            for (int s = 0; s < W; ++s) pre[tt][s] = 0.f;
        if (rowt) dec_dot_steps<W, V>(pre, wT, H4, tid, xin, Cp);               // (steps beyond nst: stale inputs, never used)
#pragma unroll
        for (int tt = 0; tt < DEC_TP; ++tt) {
            if (tt >= nst) break;
            const int step = step0 + tt;
            if (rowt) {                                    // gate pre-activations: W_ih x + W_hh h + (b_ih + b_hh)
                V acc = pre[tt];
#pragma unroll
                for (int q = 0; q < DEC_MAXH / 4; ++q) {
                    if (4 * q >= Hp) break;
                    const V x0 = hs[4 * q], x1 = hs[4 * q + 1], x2 = hs[4 * q + 2], x3 = hs[4 * q + 3];
#pragma unroll
                    for (int s = 0; s < W; ++s) {
                        acc[s] = __builtin_fmaf(whh[q].x, x0[s], acc[s]);
                        acc[s] = __builtin_fmaf(whh[q].y, x1[s], acc[s]);
                        acc[s] = __builtin_fmaf(whh[q].z, x2[s], acc[s]);
                        acc[s] = __builtin_fmaf(whh[q].w, x3[s], acc[s]);
                    }
                }
#pragma unroll
                for (int s = 0; s < W; ++s) acc[s] += bias;
                gates[tid] = acc;
            }
            __syncthreads();
            if (tid < W * H && step < Tc) {                // cell update of (stream cs, unit cu): c' = f c + i g, h' = o tanh(c')
                const float gi = reinterpret_cast<const float *>(&gates[cu])[cs];
                const float gf = reinterpret_cast<const float *>(&gates[H + cu])[cs];
                const float gg = reinterpret_cast<const float *>(&gates[2 * H + cu])[cs];
                const float go = reinterpret_cast<const float *>(&gates[3 * H + cu])[cs];
                c = dec_sigmoid(gf) * c + dec_sigmoid(gi) * tanhf(gg);
                const float h = dec_sigmoid(go) * tanhf(c);
                reinterpret_cast<float *>(&hs[cu])[cs] = h;
                out[((size_t)(s0 + cs) * T + (dir ? Tc - 1 - step : step)) * (2 * H) + dir * H + cu] = h;
            }
            __syncthreads();
        }
    }
}

// regressor (models.py:45,57): feats[row][o] = b[o] + sum_k w[o][k] top[row][k], row = (stream, frame).  A block takes
// DEC_RROWS rows: their inputs and the whole weight matrix go through LDS once (one thread per (row, output) reading both
// from global memory took 29 us for 512 rows: 200 strided loads in series per thread).
#define DEC_RROWS 8
__global__ void __launch_bounds__(256)
dec_regress_kernel(const float *__restrict__ top, long rows, int K, int O, const float *__restrict__ w, const float *__restrict__ b,
                   float *__restrict__ feats, const int *__restrict__ counts, int T)
{
    extern __shared__ __attribute__((aligned(16))) float rs[];             // [O][K + 1] weights, then [DEC_RROWS][K] inputs
    float *ws = rs, *xs = rs + (size_t)O * (K + 1);
    const int tid = threadIdx.x;
    const long r0 = (long)blockIdx.x * DEC_RROWS;
    for (int k = tid; k < O * K; k += 256) { const int o = k / K, j = k - o * K; ws[o * (K + 1) + j] = w[k]; }
    // ragged calls: row r = (stream r / T, frame r % T) exists only below its stream's frame count
    auto live = [&](long r) { return r < rows && (!counts || (int)(r % T) < counts[r / T]); };
    for (int k = tid; k < DEC_RROWS * K; k += 256) {
        const long r = r0 + k / K;
        xs[k] = live(r) ? top[r * K + (k % K)] : 0.f;
    }
    __syncthreads();
    for (int idx = tid; idx < DEC_RROWS * O; idx += 256) {
        const int rr = idx / O, o = idx - rr * O;
        if (!live(r0 + rr)) continue;
        const float *x = xs + rr * K, *wr = ws + o * (K + 1);
        float a = 0.f;
        for (int k = 0; k < K; ++k) a = __builtin_fmaf(wr[k], x[k], a);
        feats[(r0 + rr) * O + o] = a + b[o];
    }
}

int dss_launch_decoder(const DssDecDev &d, const void *d_frames, int frames_f64, int S, int T, float *d_feats,
                       const int *d_counts, const int *d_in_row, int Tin, hipStream_t st)
{
    if (Tin <= 0) Tin = T;                                 // plain calls: the input is (S, T, C)
    if (d.H < 1 || d.H > DEC_MAXH || 4 * d.H > DEC_THREADS || d.C < 1 || d.C > DEC_MAXC || 2 * d.H > DEC_MAXC) {
        dss_set_error("decoder kernel: hidden size %d / %d inputs out of range (<= %d / <= %d)", d.H, d.C, DEC_MAXH, DEC_MAXC);
        return DSS_EINVAL;
    }
    if (d.O < 1 || d.O > 32) { dss_set_error("decoder kernel: %d outputs out of range (<= 32)", d.O); return DSS_EINVAL; }
    if (S < 1 || S > d.S_max || T < 1 || T > d.T_max) {
        dss_set_error("decoder kernel: %d streams x %d frames exceed the handle's %d x %d", S, T, d.S_max, d.T_max);
        return DSS_EINVAL;
    }
    // streams per workgroup: a thread's work per step grows with W, the number of workgroups shrinks with it; 2 x S / W of
    // them should still cover the CUs (64 streams: 128 workgroups of one stream; 1024 streams: 512 of four)
    const int Wsel = 2 * S <= 256 ? 1 : (S <= 256 ? 2 : 4);
    const dim3 block(DEC_THREADS);
#define DEC_LAUNCH(INT, WV, IN, CIN, L, OUT)                                                                                   \
    hipLaunchKernelGGL((bilstm_layer_kernel<INT, WV>), dim3((S + WV - 1) / WV, 2), block, 0, st, (const INT *)(IN), S, T, CIN, d.H,  \
                       d.wT[L][0], d.wT[L][1], d.b[L][0], d.b[L][1], OUT, d_counts, (L) == 0 ? d_in_row : (const int *)nullptr,   \
                       (L) == 0 ? Tin : T)
#define DEC_LAUNCH_W(INT, IN, CIN, L, OUT)                                                                                      \
    do {                                                                                                                        \
        if (Wsel == 1) DEC_LAUNCH(INT, 1, IN, CIN, L, OUT);                                                                     \
        else if (Wsel == 2) DEC_LAUNCH(INT, 2, IN, CIN, L, OUT);                                                                \
        else DEC_LAUNCH(INT, 4, IN, CIN, L, OUT);                                                                               \
    } while (0)
    if (frames_f64) DEC_LAUNCH_W(double, d_frames, d.C, 0, d.mid);
    else DEC_LAUNCH_W(float, d_frames, d.C, 0, d.mid);
    DEC_LAUNCH_W(float, d.mid, 2 * d.H, 1, d.top);
#undef DEC_LAUNCH_W
#undef DEC_LAUNCH
    const long rows = (long)S * T;
    const size_t rlds = ((size_t)d.O * (2 * d.H + 1) + (size_t)DEC_RROWS * 2 * d.H) * sizeof(float);
    hipLaunchKernelGGL(dec_regress_kernel, dim3((unsigned)((rows + DEC_RROWS - 1) / DEC_RROWS)), dim3(256), rlds, st, d.top, rows, 2 * d.H,
                       d.O, d.wr, d.br, d_feats, d_counts, T);
    DSS_HIP_CHECK(hipGetLastError());
    return DSS_OK;
}
