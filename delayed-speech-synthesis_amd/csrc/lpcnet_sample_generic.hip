// csrc/lpcnet_sample_generic.hip -- GENERIC (any sparsity pattern) form of the sample-rate kernel: GRU A
// block weights are streamed from L2 every sample.  Used only when a model's per-wave block counts exceed
// what the register-resident kernel (lpcnet_sample.hip) was compiled for; ~6x slower.  Also holds the
// lpcnet_init() reset kernel.
//
// Restates xiph/LPCNet src/lpcnet.c lpcnet_synthesize_tail_impl() + run_sample_network() and
// src/nnet.c compute_gru_a_input / compute_sparse_gru / compute_gruB / sample_mdense (generic float path
// of src/vec.h), as reached through the reference's binding extensions/lpcnet/cLPCNet.pxd:13.
//
// Work split inside the 512-thread workgroup (8 waves, 2 per SIMD), per output sample:
//   wave 7        : "scalar" recurrences -- order-16 LPC prediction, mu-law, de-emphasis, kiss99, tree walk
//   waves 0..5    : GRU A, lane = unit i of 384; each lane runs the z, r and h rows of its unit as three
//                   independent sequential chains over its 8x4 sparse blocks (same summation order as the C)
//   wave 6        : GRU B, lane = output row (48 rows), 384-term sequential chain from LDS-resident weights
//   waves 0..3    : dual-FC, lane = tree node n of 256: both channels' 16-term chains, weights in VGPRs;
//                   all 255 node logits are evaluated, the 8-level walk is then pure scalar bit tests
// Every floating-point expression keeps the C source's association and precision (-ffp-contract=off).
#include "dss_common.h"
#include "lpcnet_device.h"

#define NA DSS_GRU_A
#define NB DSS_GRU_B
#define NB3 (3 * DSS_GRU_B)

struct SampleLds {
    float state_a[2][NA];        // double-buffered GRU A state
    float gb_w[NA * NB3];        // GRU B input weights [384][48]
    float gb_wrec[NB * NB3];     // GRU B recurrent weights [16][48]
    float tansig[208];
    float ulaw2lin[256];
    float logit_table[256];
    float state_b[NB];
    float thr[8];
    unsigned bits[8];            // decision bit of every tree node (256 bits)
    int idx[4];                  // last_sig_ulaw, pred_ulaw, last_exc
    short pcm[DSS_FRAME_SIZE];
};

template <bool TRACE, bool STAMP>
__global__ void __launch_bounds__(512)
lpcnet_sample_generic_kernel(DssModelDev m, DssBatchDev b, int n_frames, short *__restrict__ pcm_out)
{
    __shared__ __attribute__((aligned(16))) SampleLds L;
    const int utt = blockIdx.x;
    const int slot = b.slot_of ? b.slot_of[utt] : utt;
    const int nf = b.count_of ? min(b.count_of[utt], n_frames) : n_frames;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;

    // ---------------- one-time staging ---------------------------------------------------------------
    for (int k = tid; k < NA * NB3; k += 512) L.gb_w[k] = m.gru_b_w_in[k];
    for (int k = tid; k < NB * NB3; k += 512) L.gb_wrec[k] = m.gru_b_w_rec[k];
    if (tid < 201) L.tansig[tid] = m.tansig[tid];
    if (tid < 256) { L.ulaw2lin[tid] = m.ulaw2lin[tid]; L.logit_table[tid] = m.logit_table[tid]; }
    if (tid < NA) L.state_a[0][tid] = b.gru_a_state[(size_t)slot * NA + tid];
    if (tid < NB) L.state_b[tid] = b.gru_b_state[(size_t)slot * NB + tid];

    // GRU A per-unit constants (waves 0..5)
    float rbz = 0, rbr = 0, rbh = 0, dgz = 0, dgr = 0, dgh = 0;
    if (tid < NA) {
        rbz = m.gru_a_rbias[tid]; rbr = m.gru_a_rbias[NA + tid]; rbh = m.gru_a_rbias[2 * NA + tid];
        dgz = m.gru_a_diag[tid];  dgr = m.gru_a_diag[NA + tid];  dgh = m.gru_a_diag[2 * NA + tid];
    }
    // dual-FC per-node constants (waves 0..3): node n = tid
    float fw0[NB], fw1[NB], fb0 = 0, fb1 = 0, ff0 = 0, ff1 = 0;
    if (tid < DSS_FC_OUT) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            fw0[j] = m.fc_w[(size_t)tid * 2 * NB + j];
            fw1[j] = m.fc_w[(size_t)tid * 2 * NB + NB + j];
        }
        fb0 = m.fc_bias[tid]; fb1 = m.fc_bias[DSS_FC_OUT + tid];
        ff0 = m.fc_factor[tid]; ff1 = m.fc_factor[DSS_FC_OUT + tid];
    } else {
#pragma unroll
        for (int j = 0; j < NB; ++j) { fw0[j] = 0; fw1[j] = 0; }
    }
    // GRU B per-row constants (wave 6)
    float gbb0 = 0, gbb1 = 0;
    if (wave == 6 && lane < NB3) { gbb0 = m.gru_b_bias[lane]; gbb1 = m.gru_b_bias[NB3 + lane]; }
    // scalar state (wave 7, replicated in every lane)
    float last_sig[DSS_LPC_ORDER], lpc[DSS_LPC_ORDER];
    float deemph = 0.f;
    int last_exc = 0;
    DssKiss99 rng = {0, 0, 0, 0};
    if (wave == 7) {
#pragma unroll
        for (int j = 0; j < DSS_LPC_ORDER; ++j) last_sig[j] = b.last_sig[(size_t)slot * DSS_LPC_ORDER + j];
        deemph = b.deemph[slot];
        last_exc = b.last_exc[slot];
        rng.z = b.rng[slot * 4 + 0]; rng.w = b.rng[slot * 4 + 1]; rng.jsr = b.rng[slot * 4 + 2]; rng.jcong = b.rng[slot * 4 + 3];
    } else {
#pragma unroll
        for (int j = 0; j < DSS_LPC_ORDER; ++j) { last_sig[j] = 0; lpc[j] = 0; }
    }
    const int fc0 = b.fc0[utt];
    const bool recur_first = m.h.gru_a_order == DSS_GRUA_RECUR_FIRST;      // association of the z/r pre-activations
    int cur = 0;
    unsigned long long stamp_acc[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long t_prev = 0;
    __syncthreads();

    for (int f = 0; f < nf; ++f) {
        short *pcm_frame = pcm_out + ((size_t)utt * n_frames + f) * DSS_FRAME_SIZE;
        if (fc0 + f < DSS_FEATURES_DELAY) {                 // lpcnet.c: frame_count <= FEATURES_DELAY -> silence
            if (tid < DSS_FRAME_SIZE / 2) reinterpret_cast<int *>(pcm_frame)[tid] = 0;
            if (TRACE && tid < DSS_FRAME_SIZE) {
                b.trace_exc[((size_t)utt * n_frames + f) * DSS_FRAME_SIZE + tid] = -1.f;
                b.trace_pcm[((size_t)utt * n_frames + f) * DSS_FRAME_SIZE + tid] = 0.f;
            }
            continue;
        }
        const float *fo = b.frame_out + ((size_t)utt * n_frames + f) * DSS_COND_STRIDE;
        float cz = 0, cr = 0, ch = 0, gbc = 0;
        if (tid < NA) { cz = fo[tid]; cr = fo[NA + tid]; ch = fo[2 * NA + tid]; }
        if (wave == 6 && lane < NB3) gbc = fo[3 * NA + lane];
        if (wave == 7) {
#pragma unroll
            for (int j = 0; j < DSS_LPC_ORDER; ++j) lpc[j] = fo[3 * NA + NB3 + j];
        }

        for (int i = 0; i < DSS_FRAME_SIZE; ++i) {
            // ---- P1 (wave 7): prediction, mu-law indices, sampling thresholds ------------------------
            float pred = 0;
            if (STAMP) t_prev = __builtin_readcyclecounter();
            if (wave == 7) {
#pragma unroll
                for (int j = 0; j < DSS_LPC_ORDER; ++j) pred -= last_sig[j] * lpc[j];
                const int su = dss_lin2ulaw(last_sig[0]);
                const int pu = dss_lin2ulaw(pred);
                const uint32_t r0 = dss_kiss99_rand(rng);
                const uint32_t r1 = dss_kiss99_rand(rng);
                if (lane == 0) { L.idx[0] = su; L.idx[1] = pu; L.idx[2] = last_exc; }
                if (lane < 8) {
                    const uint32_t r = lane < 4 ? r0 : r1;
                    L.thr[lane] = L.logit_table[(r >> (8 * (lane & 3))) & 0xFF];
                }
            }
            if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); stamp_acc[0] += t - t_prev; t_prev = t; }
            __syncthreads();                                                        // barrier A
            if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); stamp_acc[1] += t - t_prev; t_prev = t; }

            // ---- P2 (waves 0..5): GRU A ----------------------------------------------------------------
            if (tid < NA) {
                const int si = L.idx[0], pi = L.idx[1], ei = L.idx[2];
                const float *es = m.embed_sig + (size_t)si * 3 * NA + tid;
                const float *ep = m.embed_pred + (size_t)pi * 3 * NA + tid;
                const float *ee = m.embed_exc + (size_t)ei * 3 * NA + tid;
                const float gz = ((cz + es[0]) + ep[0]) + ee[0];                    // compute_gru_a_input
                const float gr = ((cr + es[NA]) + ep[NA]) + ee[NA];
                const float gh = ((ch + es[2 * NA]) + ep[2 * NA]) + ee[2 * NA];
                const float st = L.state_a[cur][tid];
                float az = rbz + dgz * st;                                          // compute_sparse_gru
                float ar = rbr + dgr * st;
                if (!recur_first) { az = az + gz; ar = ar + gr; }                  // nnet.c 2021: input before the blocks
                float ah = rbh + dgh * st;
                const char *xbase = reinterpret_cast<const char *>(L.state_a[cur]);
#define DSS_GATE(G, ACC)                                                                         \
                for (int sl = 0; sl < m.gate[G].slots; ++sl) {                                   \
                    const int p4 = m.gate[G].pos4[sl * NA + tid];                                \
                    const float *wp = m.gate[G].w + (size_t)sl * 4 * NA + tid;                   \
                    const float4 xv = *reinterpret_cast<const float4 *>(xbase + p4);             \
                    ACC += wp[0] * xv.x;                                                         \
                    ACC += wp[NA] * xv.y;                                                        \
                    ACC += wp[2 * NA] * xv.z;                                                    \
                    ACC += wp[3 * NA] * xv.w;                                                    \
                }
                DSS_GATE(0, az)
                DSS_GATE(1, ar)
                DSS_GATE(2, ah)
#undef DSS_GATE
                if (recur_first) { az = gz + az; ar = gr + ar; }                    // nnet.c 2019-20: zrh = input; zrh += recur
                const float z = dss_sigmoid_approx(L.tansig, az);
                const float r = dss_sigmoid_approx(L.tansig, ar);
                float h = ah * r + gh;
                h = dss_tanh_approx(L.tansig, h);
                L.state_a[cur ^ 1][tid] = z * st + (1 - z) * h;
            }
            __syncthreads();                                                        // barrier B
            if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); stamp_acc[2] += t - t_prev; t_prev = t; }

            // ---- P3 (wave 6): GRU B ----------------------------------------------------------------------
            if (wave == 6) {
                const int row = lane < NB3 ? lane : 0;
                const float *an = L.state_a[cur ^ 1];
                float acc = gbb0 + gbc;                                            // compute_gruB
                for (int j = 0; j < NA; j += 4) {
                    const float4 av = *reinterpret_cast<const float4 *>(an + j);
                    acc += L.gb_w[(j + 0) * NB3 + row] * av.x;
                    acc += L.gb_w[(j + 1) * NB3 + row] * av.y;
                    acc += L.gb_w[(j + 2) * NB3 + row] * av.z;
                    acc += L.gb_w[(j + 3) * NB3 + row] * av.w;
                }
                float rec = gbb1;
#pragma unroll
                for (int j = 0; j < NB; ++j) rec += L.gb_wrec[j * NB3 + row] * L.state_b[j];
                // lanes 0..15: z, 16..31: r, 32..47: h
                const float zr = dss_sigmoid_approx(L.tansig, acc + rec);
                const float r_for_h = __shfl(zr, lane - NB);                      // r_i for lane 32+i
                float hh = acc + rec * r_for_h;
                hh = dss_tanh_approx(L.tansig, hh);
                const float h_for_z = __shfl(hh, lane + 2 * NB);                  // h_i for lane i
                if (lane < NB) {
                    const float sb = L.state_b[lane];
                    L.state_b[lane] = zr * sb + (1 - zr) * h_for_z;
                }
            }
            __syncthreads();                                                        // barrier C
            if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); stamp_acc[3] += t - t_prev; t_prev = t; }

            // ---- P4 (waves 0..3): dual-FC logits of all tree nodes, decision bits ------------------------
            if (tid < DSS_FC_OUT) {
                float s1 = fb0, s2 = fb1;
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const float bj = L.state_b[j];
                    s1 += fw0[j] * bj;
                    s2 += fw1[j] * bj;
                }
                s1 = ff0 * dss_tanh_approx(L.tansig, s1);
                s2 = ff1 * dss_tanh_approx(L.tansig, s2);
                s1 += s2;
                const int level = 31 - __clz(tid | 1);                               // node = (1 << level) | prefix
                if (TRACE && b.trace_logits)
                    b.trace_logits[(((size_t)utt * n_frames + f) * DSS_FRAME_SIZE + i) * 256 + tid] = tid ? s1 : 0.f;
                const bool bit = L.thr[level] < s1;
                const unsigned long long mask = __ballot(bit);
                if (lane == 0) { L.bits[2 * wave] = (unsigned)mask; L.bits[2 * wave + 1] = (unsigned)(mask >> 32); }
            }
            __syncthreads();                                                        // barrier D
            if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); stamp_acc[4] += t - t_prev; t_prev = t; }

            // ---- P6 (wave 7): walk the tree, finish the sample ----------------------------------------------
            if (wave == 7) {
                int val = 0;
#pragma unroll
                for (int lv = 0; lv < 8; ++lv) {
                    const int node = (1 << lv) | val;
                    const unsigned wbits = L.bits[node >> 5];
                    val = (val << 1) | ((wbits >> (node & 31)) & 1);
                }
                int exc = val;
                if (TRACE && b.force_exc) exc = b.force_exc[((size_t)utt * n_frames + f) * DSS_FRAME_SIZE + i];
                float pcm = pred + L.ulaw2lin[exc];
                if (TRACE && lane == 0) {
                    const size_t o = ((size_t)utt * n_frames + f) * DSS_FRAME_SIZE + i;
                    b.trace_exc[o] = (float)exc;
                    b.trace_pcm[o] = pcm;
                }
#pragma unroll
                for (int j = DSS_LPC_ORDER - 1; j > 0; --j) last_sig[j] = last_sig[j - 1];
                last_sig[0] = pcm;
                last_exc = exc;
                pcm += 0.85f * deemph;
                deemph = pcm;
                if (pcm < -32767) pcm = -32767;
                if (pcm > 32767) pcm = 32767;
                if (lane == 0) L.pcm[i] = (short)(int)floor(.5 + (double)pcm);
            }
            if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); stamp_acc[5] += t - t_prev; t_prev = t; }
            cur ^= 1;
            // no barrier needed here: wave 7 alone touches idx/thr/pcm before barrier A of the next sample,
            // and bits[] is rewritten only after barriers A..C.
        }
        __syncthreads();
        if (tid < DSS_FRAME_SIZE / 2) reinterpret_cast<int *>(pcm_frame)[tid] = reinterpret_cast<const int *>(L.pcm)[tid];
    }

    // ---------------- write the persistent state back ------------------------------------------------------
    __syncthreads();
    if (STAMP && lane == 0 && b.trace_pcm) {        // diagnostic build only: per-wave phase cycle sums
        for (int k = 0; k < 6; ++k) b.trace_pcm[((size_t)utt * 8 + wave) * 6 + k] = (float)stamp_acc[k];
    }
    if (tid < NA) b.gru_a_state[(size_t)slot * NA + tid] = L.state_a[cur][tid];
    if (tid < NB) b.gru_b_state[(size_t)slot * NB + tid] = L.state_b[tid];
    if (wave == 7 && lane == 0) {
#pragma unroll
        for (int j = 0; j < DSS_LPC_ORDER; ++j) b.last_sig[(size_t)slot * DSS_LPC_ORDER + j] = last_sig[j];
        b.deemph[slot] = deemph;
        b.last_exc[slot] = last_exc;
        b.rng[slot * 4 + 0] = rng.z; b.rng[slot * 4 + 1] = rng.w; b.rng[slot * 4 + 2] = rng.jsr; b.rng[slot * 4 + 3] = rng.jcong;
    }
}

int dss_launch_sample_network_generic(const DssModelDev &m, DssBatchDev &b, int n_utts, int n_frames, short *d_pcm, int trace,
                              hipStream_t s)
{
    if (trace == 2)        // diagnostic: phase stamps written into trace_pcm (never used for timing claims)
        hipLaunchKernelGGL((lpcnet_sample_generic_kernel<false, true>), dim3(n_utts), dim3(512), 0, s, m, b, n_frames, d_pcm);
    else if (trace)
        hipLaunchKernelGGL((lpcnet_sample_generic_kernel<true, false>), dim3(n_utts), dim3(512), 0, s, m, b, n_frames, d_pcm);
    else
        hipLaunchKernelGGL((lpcnet_sample_generic_kernel<false, false>), dim3(n_utts), dim3(512), 0, s, m, b, n_frames, d_pcm);
    DSS_HIP_CHECK(hipGetLastError());
    return DSS_OK;
}

// lpcnet_init(): zero state, last_exc = lin2ulaw(0), kiss99_srand("LPCNet")
__global__ void lpcnet_reset_kernel(DssBatchDev b, int utt_only, int n_utts, DssKiss99 seed, int exc0)
{
    const int utt = utt_only >= 0 ? utt_only : blockIdx.x;
    if (utt >= n_utts) return;
    const int tid = threadIdx.x;
    for (int k = tid; k < NA; k += blockDim.x) b.gru_a_state[(size_t)utt * NA + k] = 0.f;
    for (int k = tid; k < 2 * 128; k += blockDim.x) b.conv2_mem[(size_t)utt * 2 * 128 + k] = 0.f;
    for (int k = tid; k < 2 * 84; k += blockDim.x) b.conv1_mem[(size_t)utt * 2 * 84 + k] = 0.f;
    if (tid < NB) b.gru_b_state[(size_t)utt * NB + tid] = 0.f;
    if (tid < DSS_LPC_ORDER) b.last_sig[(size_t)utt * DSS_LPC_ORDER + tid] = 0.f;
    if (tid < 2 * DSS_LPC_ORDER) b.old_lpc[(size_t)utt * 2 * DSS_LPC_ORDER + tid] = 0.f;
    if (tid == 0) {
        b.last_exc[utt] = exc0;
        b.deemph[utt] = 0.f;
        b.frame_count[utt] = 0;
        b.rng[utt * 4 + 0] = seed.z; b.rng[utt * 4 + 1] = seed.w; b.rng[utt * 4 + 2] = seed.jsr; b.rng[utt * 4 + 3] = seed.jcong;
    }
}

int dss_launch_lpcnet_reset(const DssModelDev &m, DssBatchDev &b, int utt, hipStream_t s)
{
    // kiss99_srand(&rng, "LPCNet", 6) evaluated on the host (integer only)
    DssKiss99 c = {362436069u, 521288629u, 123456789u, 380116160u};
    const unsigned char data[6] = {'L', 'P', 'C', 'N', 'e', 't'};
    const int n = 6;
    int i;
    for (i = 3; i < n; i += 4) {
        c.z ^= data[i - 3]; c.w ^= data[i - 2]; c.jsr ^= data[i - 1]; c.jcong ^= data[i];
        // one kiss99_rand step
        const uint32_t znew = 36969u * (c.z & 0xFFFF) + (c.z >> 16);
        const uint32_t wnew = 18000u * (c.w & 0xFFFF) + (c.w >> 16);
        uint32_t shr3 = c.jsr ^ (c.jsr << 17);
        shr3 ^= shr3 >> 13;
        shr3 ^= shr3 << 5;
        c.z = znew; c.w = wnew; c.jsr = shr3; c.jcong = 69069u * c.jcong + 1234567u;
    }
    if (i - 3 < n) c.z ^= data[i - 3];
    if (i - 2 < n) c.w ^= data[i - 2];
    if (i - 1 < n) c.jsr ^= data[i - 1];
    if (c.z == 0 || c.z == 0x9068FFFF) c.z++;
    if (c.w == 0 || c.w == 0x464FFFFF) c.w++;
    if (c.jsr == 0) c.jsr++;
    const int exc0 = 128;   // lin2ulaw(0.f): 128 + 128*log_approx(1)/LOG256 = 128.007 -> 128
    const int grid = utt >= 0 ? 1 : b.max_utts;
    hipLaunchKernelGGL(lpcnet_reset_kernel, dim3(grid), dim3(256), 0, s, b, utt, b.max_utts, c, exc0);
    DSS_HIP_CHECK(hipGetLastError());
    return DSS_OK;
}
