// csrc/lpcnet_frame.hip -- LPCNet frame-rate network for a batch of utterances (gfx950).
//
// Restates xiph/LPCNet src/lpcnet.c run_frame_network() as consumed through lpcnet_synthesize()
// (reference binding: extensions/lpcnet/cLPCNet.pxd:13; callers local/units.py:535, local/training.py:194):
//   pitch embedding -> conv1d(k=3) tanh -> conv1d(k=3) tanh -> dense tanh -> dense tanh
//   -> {gru_a_dense_feature, gru_b_dense_feature} (linear), and lpc_from_cepstrum with the 2-frame delay.
//
// In batch mode all frames' features are known up front, and the only temporal coupling is the two
// previous inputs of each conv (plus the saved conv memories / old_lpc of the persistent state), so
// every layer runs as one launch over all (utterance, frame) rows.  A row's dot products are
// accumulated one product at a time in ascending input order (xiph sgemv_accum: out[i] += w[j][i]*x[j]),
// one lane per output neuron, so results are bit-identical to the scalar C path.  Weights are input-major
// so a wave reads 256 contiguous bytes per input; a block reuses each weight across RT rows staged in LDS.
#include "dss_common.h"
#include "lpcnet_device.h"

#define FIN (DSS_NB_FEATURES + 64)    // 84: features + pitch embedding

// ---- stage 0: build the padded per-utterance input rows and snapshot frame_count -----------------------
// in_buf[b][0..1] = conv1_mem, in_buf[b][t+2] = [features(20) | embed_pitch(64)]
// c1_buf[b][0..1] = conv2_mem, lpc_buf[b][0..1] = old_lpc
__global__ void __launch_bounds__(128)
frame_prepare_kernel(DssModelDev m, DssBatchDev b, const float *__restrict__ feat, int n_frames, int feat_stride)
{
    const int utt = blockIdx.y, row = blockIdx.x, tid = threadIdx.x;      // row in [0, F+2)
    float *in = b.in_buf + ((size_t)utt * (n_frames + 2) + row) * FIN;
    if (row < 2) {
        const int slot = b.slot_of ? b.slot_of[utt] : utt;                // carried state of this row's decoder
        if (tid < FIN) in[tid] = b.conv1_mem[((size_t)slot * 2 + row) * FIN + tid];
        b.c1_buf[((size_t)utt * (n_frames + 2) + row) * 128 + tid] = b.conv2_mem[((size_t)slot * 2 + row) * 128 + tid];
        if (tid < 16) b.lpc_buf[((size_t)utt * (n_frames + 2) + row) * 16 + tid] = b.old_lpc[((size_t)slot * 2 + row) * 16 + tid];
        if (row == 0 && tid == 0) b.fc0[utt] = b.frame_count[slot];
        return;
    }
    const float *f = feat + ((size_t)utt * n_frames + (row - 2)) * feat_stride;
    if (tid < DSS_NB_FEATURES) {
        in[tid] = f[tid];
    } else if (tid < FIN) {
        // lpcnet.c: pitch = (int)floor(.1 + 50*features[NB_BANDS]+100), clamped to [33, 255]
        int pitch = (int)floor(.1 + (double)(50 * f[DSS_NB_BANDS]) + 100);
        pitch = pitch < 33 ? 33 : pitch;
        pitch = pitch > 255 ? 255 : pitch;
        in[tid] = m.embed_pitch[(size_t)pitch * 64 + (tid - DSS_NB_FEATURES)];
    }
}

// ---- generic dense layer over rows ----------------------------------------------------------------------
enum { ACT_LINEAR = 0, ACT_TANH = 1 };

template <int ACT, int RT>
__global__ void __launch_bounds__(128)
dense_rows_kernel(const float *__restrict__ x, long x_utt_stride, int x_row_stride,
                  const float *__restrict__ W, const float *__restrict__ bias, int M, int N,
                  float *__restrict__ out, long out_utt_stride, int out_row_stride, int out_col_off,
                  int rows_per_utt, int total_rows, int zero_below, const int *__restrict__ fc0,
                  const float *__restrict__ tansig)
{
    // the RT rows of the block, staged input-major ([input][row]): the products of one weight with two rows' inputs
    // are then one packed multiply (v_pk_mul_f32, the weight broadcast to both halves); the sums stay one per product
    extern __shared__ __attribute__((aligned(16))) float xs[];     // [M][RT]
    static_assert(RT == 8 || RT == 4 || RT == 2, "the packed products below take the rows in pairs");
    const int tid = threadIdx.x;
    const int i = blockIdx.y * 128 + tid;
    const int r0 = blockIdx.x * RT;
    for (int rr = 0; rr < RT; ++rr) {
        const int r = r0 + rr;
        if (r < total_rows) {
            const int ub = r / rows_per_utt, t = r - ub * rows_per_utt;
            const float *src = x + (size_t)ub * x_utt_stride + (size_t)t * x_row_stride;
            for (int j = tid; j < M; j += 128) xs[j * RT + rr] = src[j];
        } else {
            for (int j = tid; j < M; j += 128) xs[j * RT + rr] = 0.f;
        }
    }
    __syncthreads();
    if (i >= N) return;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 acc[RT / 2];
    const float bi = bias[i];
#pragma unroll
    for (int p = 0; p < RT / 2; ++p) acc[p] = (f32x2){bi, bi};
    // One input of all RT rows: the products first (a packed result needs a wait state before it can be read), then the sums:
    // per row still "acc += w[j]*x[j]" in ascending j, each product and each sum rounded on its own (xiph sgemv_accum)
#define DSS_DENSE_STEP2(J, W0, W1)                                                               \
    {                                                                                            \
        f32x2 pp[RT / 2], qq[RT / 2];                                                            \
        _Pragma("unroll") for (int p = 0; p < RT / 2; ++p) {                                     \
            pp[p] = (f32x2){W0, W0} * *reinterpret_cast<const f32x2 *>(&xs[(J) * RT + 2 * p]);   \
            qq[p] = (f32x2){W1, W1} * *reinterpret_cast<const f32x2 *>(&xs[((J) + 1) * RT + 2 * p]); \
        }                                                                                        \
        _Pragma("unroll") for (int p = 0; p < RT / 2; ++p) { acc[p].x += pp[p].x; acc[p].y += pp[p].y; } \
        _Pragma("unroll") for (int p = 0; p < RT / 2; ++p) { acc[p].x += qq[p].x; acc[p].y += qq[p].y; } \
    }
    // eight weights per trip, all loads issued before the first product: two at a time, every pair of inputs waited for L2
    // (18-26 us per launch for the 512 rows of a streaming tick)
    int j = 0;
    for (; j + 8 <= M; j += 8) {
        float wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) wv[u] = W[(size_t)(j + u) * N + i];
#pragma unroll
        for (int u = 0; u < 8; u += 2) DSS_DENSE_STEP2(j + u, wv[u], wv[u + 1])
    }
    for (; j < M; j += 2) {
        const float w0 = W[(size_t)(j + 0) * N + i];
        const float w1 = W[(size_t)(j + 1) * N + i];
        DSS_DENSE_STEP2(j, w0, w1)
    }
#undef DSS_DENSE_STEP2
#pragma unroll
    for (int rr = 0; rr < RT; ++rr) {
        const int r = r0 + rr;
        if (r < total_rows) {
            const int ub = r / rows_per_utt, t = r - ub * rows_per_utt;
            float v = (rr & 1) ? acc[rr / 2].y : acc[rr / 2].x;
            if (ACT == ACT_TANH) v = dss_tanh_approx(tansig, v);
            if (zero_below > 0 && fc0[ub] + t < zero_below) v = 0.f;   // lpcnet.c: RNN_CLEAR while frame_count < delay
            out[(size_t)ub * out_utt_stride + (size_t)t * out_row_stride + out_col_off + i] = v;
        }
    }
}

// ---- lpc_from_cepstrum (freq.c) for every (utterance, frame): 32 lanes per frame -----------------------------
// The chain of a frame -- inverse DCT of the cepstrum, band energies, band interpolation, the 17 autocorrelation lags as a
// direct inverse DFT (160 terms each, in ascending bin order), lag window, Levinson-Durbin -- with the reference's operation
// order inside every element; what is independent runs on different lanes: one band per lane, one bin per lane, one LAG per
// lane (its 160-term sum stays one sequential chain), and the recursion on one lane.  One lane per frame (rounds 1-3) took
// 60 us whatever the batch: 160 x 17 dependent pairs of instructions per lane behind scalar loads of the cosine table.
__constant__ float c_compensation[DSS_NB_BANDS] = {0.8f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 0.666667f, 0.5f, 0.5f, 0.5f,
                                                   0.333333f, 0.25f, 0.25f, 0.2f, 0.166667f, 0.173913f};
#define LPC_FPB 2                 // frames per 64-thread block

__global__ void __launch_bounds__(64)
frame_lpc_kernel(DssModelDev m, DssBatchDev b, const float *__restrict__ feat, int total, int n_frames, int feat_stride,
                 double idct_scale)
{
    __shared__ __attribute__((aligned(16))) float ck_lds[160][20];          // m.cos_kl ([bin][17 lags])
    __shared__ float cs[LPC_FPB][DSS_NB_BANDS + 2], exs[LPC_FPB][DSS_NB_BANDS + 2], xr[LPC_FPB][160], as[LPC_FPB][DSS_LPC_ORDER + 4];
    constexpr int eband5ms[DSS_NB_BANDS] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 34, 40};
    const int tid = threadIdx.x, sub = tid >> 5, l = tid & 31;
    int gid = blockIdx.x * LPC_FPB + sub;
    const bool valid = gid < total;
    if (!valid) gid = total - 1;
    const int utt = gid / n_frames, t = gid - utt * n_frames;
    for (int k = tid; k < 160 * (DSS_LPC_ORDER + 1); k += 64) ck_lds[k / (DSS_LPC_ORDER + 1)][k % (DSS_LPC_ORDER + 1)] = m.cos_kl[k];
    const float *cep = feat + (size_t)gid * feat_stride;
    if (l < DSS_NB_BANDS) cs[sub][l] = l == 0 ? cep[0] + 4 : cep[l];
    __syncthreads();
    if (l < DSS_NB_BANDS) {                                                 // band l: inverse DCT term by term, then the energy
        float sum = 0;
#pragma unroll
        for (int j = 0; j < DSS_NB_BANDS; ++j) sum += cs[sub][j] * m.dct_table[l * DSS_NB_BANDS + j];
        const float e = (float)((double)sum * idct_scale);                 // sum*sqrt(2./NB_BANDS)
        exs[sub][l] = (float)(pow(10.0, (double)e) * (double)c_compensation[l]);
    }
    __syncthreads();
    for (int k = l; k < 160; k += 32) {                                     // interp_band_gain: bin k lies in band i
        int i = 0;
#pragma unroll
        for (int q = 1; q < DSS_NB_BANDS - 1; ++q) i += (k >= eband5ms[q] * 4);
        xr[sub][k] = m.interp_a[k] * exs[sub][i] + m.interp_b[k] * exs[sub][i + 1];
    }
    __syncthreads();
    if (l <= DSS_LPC_ORDER) {                                               // lag l: direct inverse DFT, bins ascending
        float ac = xr[sub][0];
        for (int k = 1; k < 160; ++k) {
            const float x2 = 2.f * xr[sub][k];
            ac += x2 * ck_lds[k][l];
        }
        float a;
        if (l == 0) a = (float)((double)ac + ((double)ac * 1e-4 + 320 / 12 / 38.));
        else a = (float)((double)ac * m.lag_window[l]);
        as[sub][l] = a;
    }
    __syncthreads();
    if (l == 0) {                                                           // Levinson-Durbin (the recursion is serial)
        float a[DSS_LPC_ORDER + 1];
#pragma unroll
        for (int i = 0; i <= DSS_LPC_ORDER; ++i) a[i] = as[sub][i];
        float lpc[DSS_LPC_ORDER];
#pragma unroll
        for (int i = 0; i < DSS_LPC_ORDER; ++i) lpc[i] = 0.f;
        float error = a[0];
        bool live = a[0] != 0;
#pragma unroll
        for (int i = 0; i < DSS_LPC_ORDER; i++) {
            if (live) {
                float rr = 0;
#pragma unroll
                for (int j = 0; j < i; j++) rr += lpc[j] * a[i - j];
                rr += a[i + 1];
                const float r = -rr / error;
                lpc[i] = r;
#pragma unroll
                for (int j = 0; j < (i + 1) >> 1; j++) {
                    const float tmp1 = lpc[j], tmp2 = lpc[i - 1 - j];
                    lpc[j] = tmp1 + r * tmp2;
                    lpc[i - 1 - j] = tmp2 + r * tmp1;
                }
                error = error - (r * r) * error;
                if (error < .001f * a[0]) live = false;                      // the C loop's break
            }
        }
        if (valid) {
            float *dst = b.lpc_buf + ((size_t)utt * (n_frames + 2) + t + 2) * 16;
#pragma unroll
            for (int i = 0; i < DSS_LPC_ORDER; ++i) dst[i] = lpc[i];
        }
    }
}

// lpc_from_cepstrum's `pow(10.f, Ex[i]) * compensation[i]` (freq.c) is a double pow rounded to float after the
// multiplication; the LPC taps are bit-exact only if the device's pow and the host libm's agree after that rounding.
// tests/test_gpu_lpcnet.py sweeps the reachable exponent range through this kernel (same expression as line 141).
__global__ void exp10_selftest_kernel(const float *__restrict__ x, const float *__restrict__ comp, float *__restrict__ out, long n)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)(pow(10.0, (double)x[i]) * (double)comp[i]);
}

int dss_launch_exp10_selftest(const float *d_x, const float *d_comp, float *d_out, long n, hipStream_t s)
{
    hipLaunchKernelGGL(exp10_selftest_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_x, d_comp, d_out, n);
    DSS_HIP_CHECK(hipGetLastError());
    return DSS_OK;
}

// ---- self-test of lin2ulaw as the sample kernels evaluate it (lpcnet_device.h): out[i] = lin2ulaw of the fp32 value whose
// bit pattern is start + i * stride.  tests/test_gpu_lpcnet.py compares a strided sweep of all 2^32 patterns with the C form.
__global__ void lin2ulaw_selftest_kernel(unsigned start, unsigned stride, long n, unsigned char *__restrict__ out)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (unsigned char)dss_lin2ulaw(__uint_as_float(start + (unsigned)i * stride));
}

int dss_launch_lin2ulaw_selftest(unsigned start, unsigned stride, long n, unsigned char *d_out, hipStream_t s)
{
    hipLaunchKernelGGL(lin2ulaw_selftest_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, start, stride, n, d_out);
    DSS_HIP_CHECK(hipGetLastError());
    return DSS_OK;
}

// ---- final stage: delayed lpc into frame_out, persistent state update ------------------------------------
__global__ void __launch_bounds__(128)
frame_finish_kernel(DssBatchDev b, int n_frames)
{
    const int utt = blockIdx.y, t = blockIdx.x, tid = threadIdx.x;
    if (t < n_frames) {
        if (tid < 16)
            b.frame_out[((size_t)utt * n_frames + t) * DSS_COND_STRIDE + 3 * DSS_GRU_A + 3 * DSS_GRU_B + tid] =
                b.lpc_buf[((size_t)utt * (n_frames + 2) + t) * 16 + tid];       // lpc of frame t-2 (old_lpc chain)
        return;
    }
    // t == n_frames: one block per utterance updates the carried state from the last two rows it really consumed
    // (rows beyond a ragged utterance's own count are computed but never looked at)
    const int slot = b.slot_of ? b.slot_of[utt] : utt;
    const int nf = b.count_of ? min(b.count_of[utt], n_frames) : n_frames;
    for (int row = 0; row < 2; ++row) {
        const size_t src = (size_t)utt * (n_frames + 2) + nf + row;
        if (tid < FIN) b.conv1_mem[((size_t)slot * 2 + row) * FIN + tid] = b.in_buf[src * FIN + tid];
        b.conv2_mem[((size_t)slot * 2 + row) * 128 + tid] = b.c1_buf[src * 128 + tid];
        if (tid < 16) b.old_lpc[((size_t)slot * 2 + row) * 16 + tid] = b.lpc_buf[src * 16 + tid];
    }
    if (tid == 0) {
        int fc = b.fc0[utt] + nf;
        b.frame_count[slot] = fc > 1000 ? 1000 : fc;
    }
}

template <int ACT>
static int launch_dense(const float *x, long xus, int xrs, const float *W, const float *bias, int M, int N, float *out,
                        long ous, int ors, int ooff, int rows_per_utt, int total_rows, int zero_below, const int *fc0,
                        const float *tansig, hipStream_t s)
{
    // Rows per block: a thread's chain is RT sums per input, a block re-reads the layer's weights from L2.  Small calls (the
    // 512 rows of a streaming tick: 64 blocks of 8 rows leave most CUs idle behind long chains) take 4 rows per block.
    if (total_rows <= 1024) {
        constexpr int RT = 2;
        dim3 grid((total_rows + RT - 1) / RT, (N + 127) / 128);
        hipLaunchKernelGGL((dense_rows_kernel<ACT, RT>), grid, dim3(128), (size_t)RT * M * sizeof(float), s, x, xus, xrs, W, bias, M, N, out,
                           ous, ors, ooff, rows_per_utt, total_rows, zero_below, fc0, tansig);
    } else if (total_rows <= 2048) {
        constexpr int RT = 4;
        dim3 grid((total_rows + RT - 1) / RT, (N + 127) / 128);
        hipLaunchKernelGGL((dense_rows_kernel<ACT, RT>), grid, dim3(128), (size_t)RT * M * sizeof(float), s, x, xus, xrs, W, bias, M, N, out,
                           ous, ors, ooff, rows_per_utt, total_rows, zero_below, fc0, tansig);
    } else {
        constexpr int RT = 8;
        dim3 grid((total_rows + RT - 1) / RT, (N + 127) / 128);
        hipLaunchKernelGGL((dense_rows_kernel<ACT, RT>), grid, dim3(128), (size_t)RT * M * sizeof(float), s, x, xus, xrs, W, bias, M, N, out,
                           ous, ors, ooff, rows_per_utt, total_rows, zero_below, fc0, tansig);
    }
    DSS_HIP_CHECK(hipGetLastError());
    return DSS_OK;
}

int dss_launch_frame_network(const DssModelDev &m, DssBatchDev &b, const float *d_features, int B, int F, int feat_stride,
                             hipStream_t s)
{
    const int rows = B * F;
    hipLaunchKernelGGL(frame_prepare_kernel, dim3(F + 2, B), dim3(128), 0, s, m, b, d_features, F, feat_stride);
    DSS_HIP_CHECK(hipGetLastError());
    int rc;
    // conv1: window of 3 input rows (t, t+1, t+2 of the padded buffer) -> c1_buf[b][t+2]
    rc = launch_dense<ACT_TANH>(b.in_buf, (long)(F + 2) * FIN, FIN, m.conv1_w, m.conv1_b, 3 * FIN, 128, b.c1_buf + 2 * 128,
                                (long)(F + 2) * 128, 128, 0, F, rows, 1, b.fc0, m.tansig, s);
    if (rc) return rc;
    rc = launch_dense<ACT_TANH>(b.c1_buf, (long)(F + 2) * 128, 128, m.conv2_w, m.conv2_b, 3 * 128, 128, b.c2_buf,
                                (long)F * 128, 128, 0, F, rows, 2, b.fc0, m.tansig, s);
    if (rc) return rc;
    rc = launch_dense<ACT_TANH>(b.c2_buf, (long)F * 128, 128, m.dense1_w, m.dense1_b, 128, 128, b.d1_buf, (long)F * 128,
                                128, 0, F, rows, 0, b.fc0, m.tansig, s);
    if (rc) return rc;
    rc = launch_dense<ACT_TANH>(b.d1_buf, (long)F * 128, 128, m.dense2_w, m.dense2_b, 128, 128, b.cond_buf, (long)F * 128,
                                128, 0, F, rows, 0, b.fc0, m.tansig, s);
    if (rc) return rc;
    rc = launch_dense<ACT_LINEAR>(b.cond_buf, (long)F * 128, 128, m.gru_a_dense_w, m.gru_a_dense_b, 128, 3 * DSS_GRU_A,
                                  b.frame_out, (long)F * DSS_COND_STRIDE, DSS_COND_STRIDE, 0, F, rows, 0, b.fc0, m.tansig, s);
    if (rc) return rc;
    rc = launch_dense<ACT_LINEAR>(b.cond_buf, (long)F * 128, 128, m.gru_b_dense_w, m.gru_b_dense_b, 128, 3 * DSS_GRU_B,
                                  b.frame_out, (long)F * DSS_COND_STRIDE, DSS_COND_STRIDE, 3 * DSS_GRU_A, F, rows, 0, b.fc0,
                                  m.tansig, s);
    if (rc) return rc;
    hipLaunchKernelGGL(frame_lpc_kernel, dim3((rows + LPC_FPB - 1) / LPC_FPB), dim3(64), 0, s, m, b, d_features, rows, F, feat_stride,
                       sqrt(2. / DSS_NB_BANDS));
    DSS_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(frame_finish_kernel, dim3(F + 1, B), dim3(128), 0, s, b, F);
    DSS_HIP_CHECK(hipGetLastError());
    return DSS_OK;
}
