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
// workgroups (blockIdx.y) -- and the regressor.  A 512-thread workgroup owns DEC_SPW streams and one direction for all T steps;
// thread t owns gate row t (4H <= 512 rows) and runs the row's dot product for the DEC_SPW streams at once: the weights
// (copies with four consecutive inputs of a row side by side: one 16-byte load per lane, 1 KB of consecutive bytes per wave)
// come from L2 once per workgroup and step, the inputs from LDS as broadcast reads; h lives in LDS, c in the registers of the
// thread that owns (stream, unit).  Time steps are sequential; streams x gate rows x directions are the parallel axes.
#include "dss_common.h"

#define DEC_SPW 4                 // streams per workgroup
#define DEC_THREADS 512           // >= 4 * H and >= DEC_SPW * H
#define DEC_MAXH 128              // (a multiple of 4)
#define DEC_MAXC 256              // inputs of a layer: n_inputs for layer 0, 2H above it
#ifndef DEC_INFLIGHT
#define DEC_INFLIGHT 8
#endif

typedef float df4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float dec_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

// one row of a gate matrix times [n inputs][DEC_SPW streams] from LDS.  wq: [n / 4][4H][4]; n a multiple of 4 (the host pads
// with zero weights, the kernel keeps the padded inputs at zero).
__device__ __forceinline__ void dec_dot(df4 &acc, const float *__restrict__ wq, int H4, int row, const df4 *x, int n)
{
    const df4 *wr = reinterpret_cast<const df4 *>(wq) + row;
    int q = 0;
    for (; q + DEC_INFLIGHT <= n / 4; q += DEC_INFLIGHT) { // DEC_INFLIGHT 16-byte loads in flight (L2 latency bounds a step)
        df4 w[DEC_INFLIGHT];
#pragma unroll
        for (int u = 0; u < DEC_INFLIGHT; ++u) w[u] = wr[(size_t)(q + u) * H4];
#pragma unroll
        for (int u = 0; u < DEC_INFLIGHT; ++u) {
            const df4 x0 = x[4 * (q + u)], x1 = x[4 * (q + u) + 1], x2 = x[4 * (q + u) + 2], x3 = x[4 * (q + u) + 3];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc[s] = __builtin_fmaf(w[u].x, x0[s], acc[s]);
                acc[s] = __builtin_fmaf(w[u].y, x1[s], acc[s]);
                acc[s] = __builtin_fmaf(w[u].z, x2[s], acc[s]);
                acc[s] = __builtin_fmaf(w[u].w, x3[s], acc[s]);
            }
        }
    }
    for (; q < n / 4; ++q) {
        const df4 w = wr[(size_t)q * H4];
        const df4 x0 = x[4 * q], x1 = x[4 * q + 1], x2 = x[4 * q + 2], x3 = x[4 * q + 3];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            acc[s] = __builtin_fmaf(w.x, x0[s], acc[s]);
            acc[s] = __builtin_fmaf(w.y, x1[s], acc[s]);
            acc[s] = __builtin_fmaf(w.z, x2[s], acc[s]);
            acc[s] = __builtin_fmaf(w.w, x3[s], acc[s]);
        }
    }
}

// One layer, both directions (blockIdx.y): in (S, T, Cin) -> out (S, T, 2H), forward h in [0, H), backward in [H, 2H).
template <typename InT>
__global__ void __launch_bounds__(DEC_THREADS)
bilstm_layer_kernel(const InT *__restrict__ in, int S, int T, int Cin, int H, const float *__restrict__ wT_f,
                    const float *__restrict__ wT_b, const float *__restrict__ b_f, const float *__restrict__ b_b,
                    float *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) df4 xin[DEC_MAXC];            // [input][stream of this workgroup]
    __shared__ __attribute__((aligned(16))) df4 hs[DEC_MAXH];             // [unit][stream]; units H .. Hp-1 stay zero
    __shared__ __attribute__((aligned(16))) df4 gates[4 * DEC_MAXH];      // [gate row][stream]
    const int tid = threadIdx.x, H4 = 4 * H, dir = blockIdx.y;
    const int Cp = (Cin + 3) & ~3, Hp = (H + 3) & ~3;      // the padded input counts the weight copies were built for
    const int s0 = blockIdx.x * DEC_SPW;
    const float *wT = dir ? wT_b : wT_f;
    const int cs = tid / H, cu = tid - cs * H;             // the (stream, unit) this thread owns in the cell updates
    const bool cell = tid < DEC_SPW * H && s0 + cs < S;
    float c = 0.f;                                         // create_new_initial_state: zeros (models.py:22-24)
    for (int k = tid; k < DEC_MAXH * DEC_SPW; k += DEC_THREADS) reinterpret_cast<float *>(hs)[k] = 0.f;
    for (int k = tid; k < DEC_MAXC * DEC_SPW; k += DEC_THREADS) reinterpret_cast<float *>(xin)[k] = 0.f;
    const bool rowt = tid < H4;
    const float bias = rowt ? (dir ? b_b : b_f)[tid] : 0.f;
    __syncthreads();
    for (int step = 0; step < T; ++step) {
        const int t = dir ? T - 1 - step : step;
        for (int idx = tid; idx < Cin * DEC_SPW; idx += DEC_THREADS) {
            const int sl = idx / Cin, k = idx - sl * Cin;
            reinterpret_cast<float *>(&xin[k])[sl] = (s0 + sl < S) ? (float)in[((size_t)(s0 + sl) * T + t) * Cin + k] : 0.f;
        }
        __syncthreads();
        if (rowt) {                                        // gate pre-activations: W_ih x + W_hh h + (b_ih + b_hh)
            df4 acc = {0.f, 0.f, 0.f, 0.f};
            dec_dot(acc, wT, H4, tid, xin, Cp);
            dec_dot(acc, wT + (size_t)Cp * H4, H4, tid, hs, Hp);
            acc += bias;
            gates[tid] = acc;
        }
        __syncthreads();
        if (tid < DEC_SPW * H) {                           // cell update of (stream cs, unit cu): c' = f c + i g, h' = o tanh(c')
            const float gi = reinterpret_cast<const float *>(&gates[cu])[cs];
            const float gf = reinterpret_cast<const float *>(&gates[H + cu])[cs];
            const float gg = reinterpret_cast<const float *>(&gates[2 * H + cu])[cs];
            const float go = reinterpret_cast<const float *>(&gates[3 * H + cu])[cs];
            c = dec_sigmoid(gf) * c + dec_sigmoid(gi) * tanhf(gg);
            const float h = dec_sigmoid(go) * tanhf(c);
            reinterpret_cast<float *>(&hs[cu])[cs] = h;
            if (cell) out[((size_t)(s0 + cs) * T + t) * (2 * H) + dir * H + cu] = h;
        }
        __syncthreads();
    }
}

// regressor (models.py:45,57): feats[row][o] = b[o] + sum_k w[o][k] top[row][k], row = (stream, frame).  A block takes
// DEC_RROWS rows: their inputs and the whole weight matrix go through LDS once (one thread per (row, output) reading both
// from global memory took 29 us for 512 rows: 200 strided loads in series per thread).
#define DEC_RROWS 8
__global__ void __launch_bounds__(256)
dec_regress_kernel(const float *__restrict__ top, long rows, int K, int O, const float *__restrict__ w, const float *__restrict__ b,
                   float *__restrict__ feats)
{
    extern __shared__ __attribute__((aligned(16))) float rs[];             // [O][K + 1] weights, then [DEC_RROWS][K] inputs
    float *ws = rs, *xs = rs + (size_t)O * (K + 1);
    const int tid = threadIdx.x;
    const long r0 = (long)blockIdx.x * DEC_RROWS;
    for (int k = tid; k < O * K; k += 256) { const int o = k / K, j = k - o * K; ws[o * (K + 1) + j] = w[k]; }
    for (int k = tid; k < DEC_RROWS * K; k += 256) {
        const long r = r0 + k / K;
        xs[k] = r < rows ? top[r * K + (k % K)] : 0.f;
    }
    __syncthreads();
    for (int idx = tid; idx < DEC_RROWS * O; idx += 256) {
        const int rr = idx / O, o = idx - rr * O;
        if (r0 + rr >= rows) continue;
        const float *x = xs + rr * K, *wr = ws + o * (K + 1);
        float a = 0.f;
        for (int k = 0; k < K; ++k) a = __builtin_fmaf(wr[k], x[k], a);
        feats[(r0 + rr) * O + o] = a + b[o];
    }
}

int dss_launch_decoder(const DssDecDev &d, const void *d_frames, int frames_f64, int S, int T, float *d_feats, hipStream_t st)
{
    if (d.H < 1 || d.H > DEC_MAXH || 4 * d.H > DEC_THREADS || d.C < 1 || d.C > DEC_MAXC || 2 * d.H > DEC_MAXC) {
        dss_set_error("decoder kernel: hidden size %d / %d inputs out of range (<= %d / <= %d)", d.H, d.C, DEC_MAXH, DEC_MAXC);
        return DSS_EINVAL;
    }
    if (S < 1 || S > d.S_max || T < 1 || T > d.T_max) {
        dss_set_error("decoder kernel: %d streams x %d frames exceed the handle's %d x %d", S, T, d.S_max, d.T_max);
        return DSS_EINVAL;
    }
    const dim3 grid((S + DEC_SPW - 1) / DEC_SPW, 2), block(DEC_THREADS);
    if (frames_f64)
        hipLaunchKernelGGL(bilstm_layer_kernel<double>, grid, block, 0, st, (const double *)d_frames, S, T, d.C, d.H, d.wT[0][0], d.wT[0][1],
                           d.b[0][0], d.b[0][1], d.mid);
    else
        hipLaunchKernelGGL(bilstm_layer_kernel<float>, grid, block, 0, st, (const float *)d_frames, S, T, d.C, d.H, d.wT[0][0], d.wT[0][1],
                           d.b[0][0], d.b[0][1], d.mid);
    hipLaunchKernelGGL(bilstm_layer_kernel<float>, grid, block, 0, st, (const float *)d.mid, S, T, 2 * d.H, d.H, d.wT[1][0], d.wT[1][1],
                       d.b[1][0], d.b[1][1], d.top);
    const long rows = (long)S * T;
    const size_t rlds = ((size_t)d.O * (2 * d.H + 1) + (size_t)DEC_RROWS * 2 * d.H) * sizeof(float);
    hipLaunchKernelGGL(dec_regress_kernel, dim3((unsigned)((rows + DEC_RROWS - 1) / DEC_RROWS)), dim3(256), rlds, st, d.top, rows, 2 * d.H,
                       d.O, d.wr, d.br, d_feats);
    DSS_HIP_CHECK(hipGetLastError());
    return DSS_OK;
}
