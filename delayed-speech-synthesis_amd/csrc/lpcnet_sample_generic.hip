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
//   wave 6        : GRU B, lane = output row (48 rows), 384-term sequential chain: 192 weights per row in VGPRs, the other
//                   192 streamed from LDS eight inputs ahead
//   waves 0..3    : dual-FC, lane = tree node n of 256: both channels' 16-term chains, weights in VGPRs;
//                   all 255 node logits are evaluated, the 8-level walk is then pure scalar bit tests
// Every floating-point expression keeps the C source's association and precision (-ffp-contract=off).
#define Z 0                               // (lpcnet_sample_common.h names a template parameter; unused by the macros taken here)
#include "lpcnet_sample_common.h"
#undef Z

#define GBR 192                           // GRU B inputs whose weights wave 6 keeps in VGPRs; the rest stream from LDS

struct SampleLds {
    float state_a[2][NA];        // double-buffered GRU A state
    float gb_w[(NA - GBR) * NB3];  // GRU B input weights of inputs GBR..383, [input][48] (0..GBR-1 live in wave 6's VGPRs)
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

// Roles are separate code paths (as in the CU-resident kernels), so that each gets the register file to itself: role A keeps
// a round of 3 x 8 blocks in flight, role B 192 GRU B weights; in one shared code path those live ranges add up and spill.
// Every role executes the same barrier sequence: per synthesised sample A B C D, per frame one more, then the final one.

// ---- waves 0..5: GRU A (lane = unit), waves 0..3 also the dual-FC (lane = tree node) -------------------------------------
template <bool TRACE, bool STAMP, bool HAS_FC>
__device__ __forceinline__ void generic_role_a(SampleLds &L, const DssModelDev &m, const DssBatchDev &b, int n_frames, int utt,
                                               int nf, int fc0, int tid, int wave, int lane, short *pcm_out)
{
    const float rbz = m.gru_a_rbias[tid], rbr = m.gru_a_rbias[NA + tid], rbh = m.gru_a_rbias[2 * NA + tid];
    const float dgz = m.gru_a_diag[tid], dgr = m.gru_a_diag[NA + tid], dgh = m.gru_a_diag[2 * NA + tid];
    float fw0[HAS_FC ? NB : 1], fw1[HAS_FC ? NB : 1], fb0 = 0, fb1 = 0, ff0 = 0, ff1 = 0;
    if constexpr (HAS_FC) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            fw0[j] = m.fc_w[(size_t)tid * 2 * NB + j];
            fw1[j] = m.fc_w[(size_t)tid * 2 * NB + NB + j];
        }
        fb0 = m.fc_bias[tid]; fb1 = m.fc_bias[DSS_FC_OUT + tid];
        ff0 = m.fc_factor[tid]; ff1 = m.fc_factor[DSS_FC_OUT + tid];
    }
    const bool recur_first = m.h.gru_a_order == DSS_GRUA_RECUR_FIRST;      // association of the z/r pre-activations
    const int nzr = m.gate[0].slots, nhh = m.gate[2].slots;      // z and r lists share one padded length (dss_capi.cpp)
    const int level = 31 - __clz(tid | 1);                                   // node = (1 << level) | prefix
    unsigned long long stamp_acc[6] = {0, 0, 0, 0, 0, 0}, t_prev = 0;
    int cur = 0;
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
        const float cz = fo[tid], cr = fo[NA + tid], ch = fo[2 * NA + tid];
        for (int i = 0; i < DSS_FRAME_SIZE; ++i) {
            if (STAMP) t_prev = __builtin_readcyclecounter();
            __syncthreads();                                                        // barrier A
            if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); stamp_acc[1] += t - t_prev; t_prev = t; }
            {
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
                // The block weights come from L2 every sample (that is what makes this kernel generic); they do not depend
                // on the state, so the loads of a whole ROUND -- sixteen blocks: eight of z and eight of r, then sixteen of h --
                // are in flight before the first product is formed: one L2 round trip per round instead of one per block.  Slots are padded with zero blocks up
                // to a multiple of 8; within a gate, products and sums stay one at a time in idx order.  (Masking the padded
                // slots' loads per lane was tried: the per-slot branches cost more than the traffic they save, 152 vs 144 ms.)
#define DSS_GATE_LOAD(G, P, W)                                                                   \
                _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                  \
                    P[u] = m.gate[G].pos4[(sl + u) * NA + tid];                                  \
                    const float *wp = m.gate[G].w + (size_t)(sl + u) * 4 * NA + tid;             \
                    W[u][0] = wp[0]; W[u][1] = wp[NA]; W[u][2] = wp[2 * NA]; W[u][3] = wp[3 * NA]; \
                }
#define DSS_GATE_MAC(ACC, P, W)                                                                  \
                _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                  \
                    const float4 xv = *reinterpret_cast<const float4 *>(xbase + P[u]);           \
                    ACC += W[u][0] * xv.x;                                                       \
                    ACC += W[u][1] * xv.y;                                                       \
                    ACC += W[u][2] * xv.z;                                                       \
                    ACC += W[u][3] * xv.w;                                                       \
                }
                // (no conditional loads: a conditionally defined array stays alive across loop iterations and spills;
                // dss_capi.cpp pads the z and r lists to one common multiple of 8 and the h list to a multiple of 16)
                for (int sl = 0; sl < nzr; sl += 8) {                               // z and r rounds: 16 blocks in flight
                    int pz[8], pr[8];
                    float wz[8][4], wr[8][4];
                    DSS_GATE_LOAD(0, pz, wz)
                    DSS_GATE_LOAD(1, pr, wr)
                    DSS_GATE_MAC(az, pz, wz)
                    DSS_GATE_MAC(ar, pr, wr)
                }
                for (int sl0 = 0; sl0 < nhh; sl0 += 16) {                           // h rounds: 16 blocks in flight
                    int pa[8], pb[8];
                    float wa[8][4], wb[8][4];
                    { const int sl = sl0; DSS_GATE_LOAD(2, pa, wa) }
                    { const int sl = sl0 + 8; DSS_GATE_LOAD(2, pb, wb) }
                    DSS_GATE_MAC(ah, pa, wa)
                    DSS_GATE_MAC(ah, pb, wb)
                }
#undef DSS_GATE_LOAD
#undef DSS_GATE_MAC
                if (recur_first) { az = gz + az; ar = gr + ar; }                    // nnet.c 2019-20: zrh = input; zrh += recur
                const float z = dss_sigmoid_approx(L.tansig, az);
                const float r = dss_sigmoid_approx(L.tansig, ar);
                float h = ah * r + gh;
                h = dss_tanh_approx(L.tansig, h);
                L.state_a[cur ^ 1][tid] = z * st + (1 - z) * h;
            }
            __syncthreads();                                                        // barrier B
            if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); stamp_acc[2] += t - t_prev; t_prev = t; }
            __syncthreads();                                                        // barrier C
            if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); stamp_acc[3] += t - t_prev; t_prev = t; }
            if constexpr (HAS_FC) {                                                 // sample_mdense, all nodes
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
                if (TRACE && b.trace_logits)
                    b.trace_logits[(((size_t)utt * n_frames + f) * DSS_FRAME_SIZE + i) * 256 + tid] = tid ? s1 : 0.f;
                const bool bit = L.thr[level] < s1;
                const unsigned long long mask = __ballot(bit);
                if (lane == 0) { L.bits[2 * wave] = (unsigned)mask; L.bits[2 * wave + 1] = (unsigned)(mask >> 32); }
            }
            __syncthreads();                                                        // barrier D
            if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); stamp_acc[4] += t - t_prev; t_prev = t; }
            cur ^= 1;
        }
        __syncthreads();                                                            // frame barrier
        if (tid < DSS_FRAME_SIZE / 2) reinterpret_cast<int *>(pcm_frame)[tid] = reinterpret_cast<const int *>(L.pcm)[tid];
    }
    __syncthreads();                                                                // final barrier
    if (STAMP && lane == 0 && b.trace_pcm)
        for (int k = 0; k < 6; ++k) b.trace_pcm[((size_t)utt * 8 + wave) * 6 + k] = (float)stamp_acc[k];
    b.gru_a_state[(size_t)(b.slot_of ? b.slot_of[utt] : utt) * NA + tid] = L.state_a[cur][tid];
}

template <bool TRACE, bool STAMP>
__global__ void __launch_bounds__(512)
lpcnet_sample_generic_kernel(DssModelDev m, DssBatchDev b, int n_frames, short *__restrict__ pcm_out)
{
    __shared__ __attribute__((aligned(16))) SampleLds L;
    const int utt = blockIdx.x;
    const int slot = b.slot_of ? b.slot_of[utt] : utt;
    const int nf = b.count_of ? min(b.count_of[utt], n_frames) : n_frames;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;

    // ---------------- one-time staging ---------------------------------------------------------------
    for (int k = tid; k < (NA - GBR) * NB3; k += 512) L.gb_w[k] = m.gru_b_w_in[(size_t)GBR * NB3 + k];
    for (int k = tid; k < NB * NB3; k += 512) L.gb_wrec[k] = m.gru_b_w_rec[k];
    if (tid < 201) L.tansig[tid] = m.tansig[tid];
    if (tid < 256) { L.ulaw2lin[tid] = m.ulaw2lin[tid]; L.logit_table[tid] = m.logit_table[tid]; }
    if (tid < NA) L.state_a[0][tid] = b.gru_a_state[(size_t)slot * NA + tid];
    if (tid < NB) L.state_b[tid] = b.gru_b_state[(size_t)slot * NB + tid];
    const int fc0 = b.fc0[utt];
    __syncthreads();

    if (wave < 4) { generic_role_a<TRACE, STAMP, true>(L, m, b, n_frames, utt, nf, fc0, tid, wave, lane, pcm_out); return; }
    if (wave < 6) { generic_role_a<TRACE, STAMP, false>(L, m, b, n_frames, utt, nf, fc0, tid, wave, lane, pcm_out); return; }
    unsigned long long stamp_acc[6] = {0, 0, 0, 0, 0, 0}, t_prev = 0;
    int cur = 0;
    if (wave == 6) {
        // ---- wave 6: GRU B, lane = output row (48 rows), one 384-term sequential chain per row -----------------------
        const int row = lane < NB3 ? lane : 0;
        const float gbb0 = m.gru_b_bias[row], gbb1 = m.gru_b_bias[NB3 + row];
        f32x2 WB[GBR / 2];                                   // inputs 0..GBR-1 as pairs, lane = row
#pragma unroll
        for (int j = 0; j < GBR / 2; ++j) {
            WB[j].x = m.gb_w_lane[(size_t)(2 * j) * 64 + lane];
            WB[j].y = m.gb_w_lane[(size_t)(2 * j + 1) * 64 + lane];
        }
        for (int f = 0; f < nf; ++f) {
            short *pcm_frame = pcm_out + ((size_t)utt * n_frames + f) * DSS_FRAME_SIZE;
            if (fc0 + f < DSS_FEATURES_DELAY) continue;
            const float gbc = b.frame_out[((size_t)utt * n_frames + f) * DSS_COND_STRIDE + 3 * NA + row];
            for (int i = 0; i < DSS_FRAME_SIZE; ++i) {
                if (STAMP) t_prev = __builtin_readcyclecounter();
                __syncthreads();                                                    // barrier A
                if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); stamp_acc[1] += t - t_prev; t_prev = t; }
                __syncthreads();                                                    // barrier B
                if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); stamp_acc[2] += t - t_prev; t_prev = t; }
                const float *an = L.state_a[cur ^ 1];
                float acc = gbb0 + gbc;                                            // compute_gruB
                DSS_GB_CHAIN(an, GBR)                                              // inputs 0..GBR-1: weights in VGPRs
                {   // inputs GBR..383: weights from LDS, the next eight fetched while the current eight are summed
                    float wA[8], wB[8];
                    f32x4 xA[2], xB[2];
#define DSS_GBS_LOAD(WQ, XQ, J)                                                                  \
                    {                                                                            \
                        _Pragma("unroll") for (int u = 0; u < 8; ++u) WQ[u] = L.gb_w[((J) - GBR + u) * NB3 + row]; \
                        XQ[0] = *reinterpret_cast<const f32x4 *>(an + (J));                      \
                        XQ[1] = *reinterpret_cast<const f32x4 *>(an + (J) + 4);                  \
                    }
#define DSS_GBS_MAC(WQ, XQ)                                                                      \
                    {                                                                            \
                        acc += WQ[0] * XQ[0].x; acc += WQ[1] * XQ[0].y; acc += WQ[2] * XQ[0].z; acc += WQ[3] * XQ[0].w; \
                        acc += WQ[4] * XQ[1].x; acc += WQ[5] * XQ[1].y; acc += WQ[6] * XQ[1].z; acc += WQ[7] * XQ[1].w; \
                    }
                    DSS_GBS_LOAD(wA, xA, GBR)
#pragma unroll 1
                    for (int j = GBR; j < NA; j += 16) {
                        DSS_GBS_LOAD(wB, xB, j + 8)
                        DSS_GBS_MAC(wA, xA)
                        if (j + 16 < NA) DSS_GBS_LOAD(wA, xA, j + 16)
                        DSS_GBS_MAC(wB, xB)
                    }
#undef DSS_GBS_LOAD
#undef DSS_GBS_MAC
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
                __syncthreads();                                                    // barrier C
                if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); stamp_acc[3] += t - t_prev; t_prev = t; }
                __syncthreads();                                                    // barrier D
                if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); stamp_acc[4] += t - t_prev; t_prev = t; }
                cur ^= 1;
            }
            __syncthreads();                                                        // frame barrier
            if (tid < DSS_FRAME_SIZE / 2) reinterpret_cast<int *>(pcm_frame)[tid] = reinterpret_cast<const int *>(L.pcm)[tid];
        }
        __syncthreads();                                                            // final barrier
        if (STAMP && lane == 0 && b.trace_pcm)
            for (int k = 0; k < 6; ++k) b.trace_pcm[((size_t)utt * 8 + wave) * 6 + k] = (float)stamp_acc[k];
        if (lane < NB) b.gru_b_state[(size_t)slot * NB + lane] = L.state_b[lane];
        return;
    }
    // ---- wave 7: "scalar" recurrences -- order-16 LPC prediction, mu-law, de-emphasis, kiss99, tree walk ------------------
    float last_sig[DSS_LPC_ORDER], lpc[DSS_LPC_ORDER];
#pragma unroll
    for (int j = 0; j < DSS_LPC_ORDER; ++j) { last_sig[j] = b.last_sig[(size_t)slot * DSS_LPC_ORDER + j]; lpc[j] = 0; }
    float deemph = b.deemph[slot];
    int last_exc = b.last_exc[slot];
    DssKiss99 rng = {b.rng[slot * 4 + 0], b.rng[slot * 4 + 1], b.rng[slot * 4 + 2], b.rng[slot * 4 + 3]};
    for (int f = 0; f < nf; ++f) {
        short *pcm_frame = pcm_out + ((size_t)utt * n_frames + f) * DSS_FRAME_SIZE;
        if (fc0 + f < DSS_FEATURES_DELAY) continue;
        const float *fo = b.frame_out + ((size_t)utt * n_frames + f) * DSS_COND_STRIDE;
#pragma unroll
        for (int j = 0; j < DSS_LPC_ORDER; ++j) lpc[j] = fo[3 * NA + NB3 + j];
        for (int i = 0; i < DSS_FRAME_SIZE; ++i) {
            // ---- P1: prediction, mu-law indices, sampling thresholds ------------------------------------
            if (STAMP) t_prev = __builtin_readcyclecounter();
            float pred = 0;
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
            if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); stamp_acc[0] += t - t_prev; t_prev = t; }
            __syncthreads();                                                        // barrier A
            __syncthreads();                                                        // barrier B
            __syncthreads();                                                        // barrier C
            __syncthreads();                                                        // barrier D
            if (STAMP) t_prev = __builtin_readcyclecounter();
            // ---- P6: walk the tree, finish the sample -----------------------------------------------------
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
            if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); stamp_acc[5] += t - t_prev; t_prev = t; }
        }
        __syncthreads();                                                            // frame barrier
        if (tid < DSS_FRAME_SIZE / 2) reinterpret_cast<int *>(pcm_frame)[tid] = reinterpret_cast<const int *>(L.pcm)[tid];
    }
    __syncthreads();                                                                // final barrier
    if (STAMP && lane == 0 && b.trace_pcm)
        for (int k = 0; k < 6; ++k) b.trace_pcm[((size_t)utt * 8 + wave) * 6 + k] = (float)stamp_acc[k];
    if (lane == 0) {
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
