/*
 * oracle/lpcnet_oracle.c -- CPU restatement of the LPCNet decoder the reference calls.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this file's library; the product path never links, imports or calls it.
 *
 * PARITY UNPINNED.  The reference does not contain this algorithm: extensions/lpcnet/LPCNet/ is an
 * empty, un-fetched submodule of xiph/LPCNet (.gitmodules:1-3, commit not recoverable; era: late
 * 2021 by the TU list at extensions/lpcnet/setup.py:34-36), its weights (src/nnet_data.c) come from
 * a network download, and the reference holds no test, golden waveform or feature file for it.
 * This file restates xiph/LPCNet's *published* decoder (src/lpcnet.c, src/nnet.c, src/vec.h generic
 * float path, src/kiss99.c, src/common.h, src/freq.c) as consumed at the reference's call sites:
 *     lpcnet_create/init/destroy/synthesize  <- extensions/lpcnet/cLPCNet.pxd:10-13,
 *                                               extensions/lpcnet/LPCNet.pyx:15,21,28,39
 *     one call per 10 ms frame, 20 floats in, 160 int16 out  <- local/units.py:531-538,
 *                                                               local/training.py:182-198
 * Known, deliberate deviations from the xiph text (documented in DESIGN.md section "Oracle"):
 *   - weights are read from a blob (include/dss_lpcnet_blob.h), not compiled in;
 *   - lpc_from_cepstrum's 320-point inverse FFT (kiss_fft) is restated as the direct real inverse
 *     DFT of the 17 lags Levinson needs -- equal in exact arithmetic, not bit-for-bit with kiss_fft;
 *   - tanh/sigmoid use the 201-entry table form (tansig_table[i] = tanh(0.04 i) to 6 decimals).
 * What pins this file today (tests/test_oracle_lpcnet_pins.py; DESIGN.md section 2 lists every assumption
 * with the test that covers it): Marsaglia's published KISS99 check values, the mu-law round trip over
 * all 256 codes, lpc_from_cepstrum against numpy's 320-point inverse FFT + scipy's Toeplitz solver,
 * Levinson on closed-form autocorrelations, the activation tables against libm, and an independent
 * NumPy restatement of one sample step written from the layer definitions (teacher-forced, all 255
 * node logits); plus oracle-vs-HIP bit equality and the committed self-generated golden vectors
 * (tests/golden/lpcnet_*.npz).  None of these is an output of xiph's own code: the +-1 LSB claim
 * against real xiph C stays open, and tests/golden/lpcnet_xiph.npz is the slot that closes it.
 *
 * The association order of compute_sparse_gru's z/r pre-activations differs between public xiph
 * revisions; both are implemented and the blob header selects one (dss_blob_header.gru_a_order).
 *
 * Float discipline: every operation below is IEEE binary32 (or binary64 where the xiph source
 * promotes to double), evaluated in source order, no FMA contraction (build: -O2 -ffp-contract=off,
 * generic x86-64, which is how the reference's extension is effectively compiled: setup.py:48-52
 * passes no -march; CFLAGS at :27-29 reach only ./configure).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/dss_lpcnet_blob.h"

#define LPC_ORDER 16
#define NB_BANDS 18
#define MAX_N 2048
#define FRAME_SIZE 160
#define WINDOW_SIZE 320
#define FREQ_SIZE 161
#define PREEMPH 0.85f
#define FEATURE_CONV1_DELAY 1
#define FEATURES_DELAY 2
#define LOG256 5.5451774445f

/* ------------------------------------------------------------------------------------------------
 * model = parsed view of a blob + the derived constant tables lpcnet_init()/freq.c build at run time
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    dss_blob_header h;
    void *storage;
    const float *embed_pitch;
    const float *conv1_w, *conv1_b, *conv2_w, *conv2_b;
    const float *dense1_w, *dense1_b, *dense2_w, *dense2_b;
    const float *gru_a_dense_w, *gru_a_dense_b, *gru_b_dense_w, *gru_b_dense_b;
    const float *embed_sig, *embed_pred, *embed_exc;
    const float *gru_a_rbias, *gru_a_diag;
    const int32_t *gru_a_idx;
    const float *gru_a_w;
    const float *gru_b_bias, *gru_b_w_in, *gru_b_w_rec;
    const float *dual_fc_bias, *dual_fc_w, *dual_fc_factor;
    /* derived tables */
    float tansig_table[201];
    float sampling_logit_table[256];   /* lpcnet_init(): -log((1-p)/p), p = .025+.95*i/255 */
    float ulaw2lin_table[256];         /* ulaw2lin(u) for integer u (only integers ever reach it) */
    float dct_table[NB_BANDS * NB_BANDS];
    float idct_scale;                  /* sqrt(2./NB_BANDS) applied as double in the source */
    float cos_table[WINDOW_SIZE];      /* cos(2 pi m / 320) */
} oracle_lpcnet_model;

static const float compensation[NB_BANDS] = {
    0.8f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 0.666667f, 0.5f, 0.5f, 0.5f, 0.333333f, 0.25f, 0.25f, 0.2f,
    0.166667f, 0.173913f};
static const int eband5ms[NB_BANDS] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 34, 40};

static float ulaw2lin(float u)       /* xiph common.h */
{
    float s;
    float scale_1 = 32768.f / 255.f;
    u = u - 128.f;
    s = (u < 0) ? -1.f : 1.f;
    u = fabsf(u);
    return s * scale_1 * (exp(u / 128. * LOG256) - 1);
}

static float log2_approx(float x)
{
    int integer;
    float frac;
    union { float f; int i; } in;
    in.f = x;
    integer = (in.i >> 23) - 127;
    in.i -= integer << 23;
    frac = in.f - 1.5f;
    frac = -0.41445418f + frac * (0.95909232f + frac * (-0.33951290f + frac * 0.16541097f));
    return 1 + integer + frac;
}
#define log_approx(x) (0.69315f * log2_approx(x))

static int lin2ulaw(float x)         /* xiph common.h */
{
    float u;
    float scale = 255.f / 32768.f;
    int s = (x < 0) ? -1 : 1;
    x = fabsf(x);
    u = (s * (128 * log_approx(1 + scale * x) / LOG256));
    u = 128 + u;
    if (u < 0) u = 0;
    if (u > 255) u = 255;
    return (int)floor(.5 + u);
}

/* exported for the known-answer tests */
int oracle_lin2ulaw(float x) { return lin2ulaw(x); }
/* lin2ulaw of the fp32 values with bit patterns start + i * stride (the sweep of the device self-test) */
void oracle_lin2ulaw_sweep(unsigned start, unsigned stride, long n, unsigned char *out)
{
    for (long i = 0; i < n; ++i) {
        union { unsigned u; float f; } v;
        v.u = start + (unsigned)i * stride;
        out[i] = (unsigned char)lin2ulaw(v.f);
    }
}
float oracle_ulaw2lin(float u) { return ulaw2lin(u); }

oracle_lpcnet_model *oracle_lpcnet_model_load(const void *blob, size_t len)
{
    if (len < sizeof(dss_blob_header)) return NULL;
    oracle_lpcnet_model *m = (oracle_lpcnet_model *)calloc(1, sizeof(*m));
    memcpy(&m->h, blob, sizeof(m->h));
    if (memcmp(m->h.magic, DSS_BLOB_MAGIC, 8) != 0 || m->h.version != 1) { free(m); return NULL; }
    const dss_blob_header *h = &m->h;
    if (h->nb_bands != NB_BANDS || h->lpc_order != LPC_ORDER || 3 * h->gru_a > MAX_N) { free(m); return NULL; }
    if (h->gru_a_order != DSS_GRUA_INPUT_FIRST && h->gru_a_order != DSS_GRUA_RECUR_FIRST) { free(m); return NULL; }
    m->storage = malloc(len);
    memcpy(m->storage, blob, len);
    const float *p = (const float *)((const char *)m->storage + sizeof(dss_blob_header));
    const int fin = h->nb_features + h->embed_pitch_dim;
    const int NA3 = 3 * h->gru_a, NB3 = 3 * h->gru_b;
#define TAKE(field, count) do { m->field = p; p += (size_t)(count); } while (0)
    TAKE(embed_pitch, (size_t)h->pitch_max * h->embed_pitch_dim);
    TAKE(conv1_w, (size_t)3 * fin * h->conv1_out);           TAKE(conv1_b, h->conv1_out);
    TAKE(conv2_w, (size_t)3 * h->conv1_out * h->conv2_out);  TAKE(conv2_b, h->conv2_out);
    TAKE(dense1_w, (size_t)h->conv2_out * h->dense1_out);    TAKE(dense1_b, h->dense1_out);
    TAKE(dense2_w, (size_t)h->dense1_out * h->dense2_out);   TAKE(dense2_b, h->dense2_out);
    TAKE(gru_a_dense_w, (size_t)h->dense2_out * NA3);        TAKE(gru_a_dense_b, NA3);
    TAKE(gru_b_dense_w, (size_t)h->dense2_out * NB3);        TAKE(gru_b_dense_b, NB3);
    TAKE(embed_sig, (size_t)256 * NA3);
    TAKE(embed_pred, (size_t)256 * NA3);
    TAKE(embed_exc, (size_t)256 * NA3);
    TAKE(gru_a_rbias, NA3);
    TAKE(gru_a_diag, NA3);
    m->gru_a_idx = (const int32_t *)p; p += h->sparse_idx_len;
    TAKE(gru_a_w, (size_t)h->sparse_nblocks * 32);
    TAKE(gru_b_bias, 2 * NB3);
    TAKE(gru_b_w_in, (size_t)h->gru_a * NB3);
    TAKE(gru_b_w_rec, (size_t)h->gru_b * NB3);
    TAKE(dual_fc_bias, 2 * h->dual_fc_out);
    TAKE(dual_fc_w, (size_t)h->dual_fc_out * 2 * h->gru_b);
    TAKE(dual_fc_factor, 2 * h->dual_fc_out);
#undef TAKE
    if ((size_t)((const char *)p - (const char *)m->storage) != len) { free(m->storage); free(m); return NULL; }

    for (int i = 0; i < 201; ++i)       /* tansig_table.h prints tanh(.04*i) with 6 decimals */
        m->tansig_table[i] = (float)(floor(tanh(.04 * i) * 1e6 + .5) / 1e6);
    for (int i = 0; i < 256; ++i) {     /* lpcnet.c lpcnet_init() */
        float prob = .025 + .95 * i / 255.;
        m->sampling_logit_table[i] = -log((1 - prob) / prob);
        m->ulaw2lin_table[i] = ulaw2lin((float)i);
    }
    for (int i = 0; i < NB_BANDS; ++i)  /* freq.c check_init() */
        for (int j = 0; j < NB_BANDS; ++j) {
            m->dct_table[i * NB_BANDS + j] = cos((i + .5) * j * M_PI / NB_BANDS);
            if (j == 0) m->dct_table[i * NB_BANDS + j] *= sqrt(.5);
        }
    for (int i = 0; i < WINDOW_SIZE; ++i) m->cos_table[i] = (float)cos(2. * M_PI * i / WINDOW_SIZE);
    return m;
}

void oracle_lpcnet_model_free(oracle_lpcnet_model *m)
{
    if (m) { free(m->storage); free(m); }
}

const float *oracle_lpcnet_table(const oracle_lpcnet_model *m, int which)
{
    switch (which) {
    case 0: return m->tansig_table;
    case 1: return m->sampling_logit_table;
    case 2: return m->ulaw2lin_table;
    case 3: return m->dct_table;
    case 4: return m->cos_table;
    }
    return NULL;
}

/* ------------------------------------------------------------------------------------------------
 * activations (vec.h generic path, table form)
 * ---------------------------------------------------------------------------------------------- */
static float tanh_approx(const oracle_lpcnet_model *m, float x)
{
    int i;
    float y, dy;
    float sign = 1;
    if (x < 0) { x = -x; sign = -1; }
    i = (int)floor(.5f + 25 * x);
    if (i < 0) i = 0;
    if (i > 200) i = 200;
    x -= .04f * i;
    y = m->tansig_table[i];
    dy = 1 - y * y;
    y = y + x * dy * (1 - y * x);
    return sign * y;
}

static float sigmoid_approx(const oracle_lpcnet_model *m, float x)
{
    return .5f + .5f * tanh_approx(m, .5f * x);
}

float oracle_tanh_approx(const oracle_lpcnet_model *m, float x) { return tanh_approx(m, x); }
float oracle_sigmoid_approx(const oracle_lpcnet_model *m, float x) { return sigmoid_approx(m, x); }

/* sgemv_accum (vec.h generic): out[i] += w[j*stride + i] * x[j], j ascending, one product at a time */
static void sgemv_accum(float *out, const float *w, int rows, int cols, int stride, const float *x)
{
    for (int i = 0; i < rows; ++i)
        for (int j = 0; j < cols; ++j)
            out[i] += w[(size_t)j * stride + i] * x[j];
}

enum { ACT_LINEAR = 0, ACT_TANH = 1, ACT_SIGMOID = 2 };

static void dense(const oracle_lpcnet_model *m, const float *w, const float *b, int n_in, int n_out, int act,
                  float *out, const float *in)
{
    for (int i = 0; i < n_out; ++i) out[i] = b[i];
    sgemv_accum(out, w, n_out, n_in, n_out, in);
    if (act == ACT_TANH) for (int i = 0; i < n_out; ++i) out[i] = tanh_approx(m, out[i]);
    else if (act == ACT_SIGMOID) for (int i = 0; i < n_out; ++i) out[i] = sigmoid_approx(m, out[i]);
}

/* compute_conv1d (nnet.c), kernel size 3: mem holds the previous two inputs */
static void conv1d(const oracle_lpcnet_model *m, const float *w, const float *b, int n_in, int n_out,
                   float *out, float *mem, const float *in)
{
    float tmp[3 * 512];
    memcpy(tmp, mem, sizeof(float) * 2 * n_in);
    memcpy(tmp + 2 * n_in, in, sizeof(float) * n_in);
    for (int i = 0; i < n_out; ++i) out[i] = b[i];
    sgemv_accum(out, w, n_out, 3 * n_in, n_out, tmp);
    for (int i = 0; i < n_out; ++i) out[i] = tanh_approx(m, out[i]);
    memcpy(mem, tmp + n_in, sizeof(float) * 2 * n_in);
}

/* ------------------------------------------------------------------------------------------------
 * kiss99 (src/kiss99.c): Marsaglia's KISS99 with the xiph seeding
 * ---------------------------------------------------------------------------------------------- */
typedef struct { uint32_t z, w, jsr, jcong; } kiss99_ctx;

static uint32_t kiss99_rand(kiss99_ctx *c)
{
    uint32_t znew = 36969 * (c->z & 0xFFFF) + (c->z >> 16);
    uint32_t wnew = 18000 * (c->w & 0xFFFF) + (c->w >> 16);
    uint32_t mwc = (znew << 16) + wnew;
    uint32_t shr3 = c->jsr ^ (c->jsr << 17);
    shr3 ^= shr3 >> 13;
    shr3 ^= shr3 << 5;
    uint32_t cong = 69069 * c->jcong + 1234567;
    c->z = znew; c->w = wnew; c->jsr = shr3; c->jcong = cong;
    return (mwc ^ cong) + shr3;
}

static void kiss99_srand(kiss99_ctx *c, const unsigned char *data, int n)
{
    int i;
    c->z = 362436069; c->w = 521288629; c->jsr = 123456789; c->jcong = 380116160;
    for (i = 3; i < n; i += 4) {
        c->z ^= data[i - 3]; c->w ^= data[i - 2]; c->jsr ^= data[i - 1]; c->jcong ^= data[i];
        kiss99_rand(c);
    }
    if (i - 3 < n) c->z ^= data[i - 3];
    if (i - 2 < n) c->w ^= data[i - 2];
    if (i - 1 < n) c->jsr ^= data[i - 1];
    if (c->z == 0 || c->z == 0x9068FFFF) c->z++;
    if (c->w == 0 || c->w == 0x464FFFFF) c->w++;
    if (c->jsr == 0) c->jsr++;
}

/* exported for the known-answer tests (Marsaglia's check values, the "LPCNet" seeding) */
void oracle_kiss99_seed(uint32_t *ctx4, uint32_t z, uint32_t w, uint32_t jsr, uint32_t jcong)
{
    ctx4[0] = z; ctx4[1] = w; ctx4[2] = jsr; ctx4[3] = jcong;
}
void oracle_kiss99_srand(uint32_t *ctx4, const unsigned char *data, int n)
{
    kiss99_ctx c;
    kiss99_srand(&c, data, n);
    ctx4[0] = c.z; ctx4[1] = c.w; ctx4[2] = c.jsr; ctx4[3] = c.jcong;
}
/* n draws; the last min(n, n_keep) values are stored in out (oldest first) */
void oracle_kiss99_draw(uint32_t *ctx4, long n, uint32_t *out, long n_keep)
{
    kiss99_ctx c = {ctx4[0], ctx4[1], ctx4[2], ctx4[3]};
    for (long i = 0; i < n; ++i) {
        uint32_t v = kiss99_rand(&c);
        if (out && i >= n - n_keep) out[i - (n - n_keep)] = v;
    }
    ctx4[0] = c.z; ctx4[1] = c.w; ctx4[2] = c.jsr; ctx4[3] = c.jcong;
}

/* ------------------------------------------------------------------------------------------------
 * decoder state (lpcnet_private.h LPCNetState)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const oracle_lpcnet_model *m;
    float conv1_mem[2 * 512];
    float conv2_mem[2 * 512];
    float gru_a_state[MAX_N / 3];
    float gru_b_state[64];
    float gru_a_condition[MAX_N];
    float gru_b_condition[3 * 64];
    float last_sig[LPC_ORDER];
    float lpc[LPC_ORDER];
    float old_lpc[FEATURES_DELAY][LPC_ORDER];
    int last_exc;
    int frame_count;
    float deemph_mem;
    kiss99_ctx rng;
    /* optional trace buffers (tests): per synthesized sample */
    unsigned char *trace_exc;
    float *trace_pcm;       /* pred + ulaw2lin(exc), before de-emphasis */
    long trace_pos, trace_cap;
    /* optional teacher forcing (tests): the excitation index of sample k is taken from forced_exc[k]
     * instead of the sampled one (the RNG still advances, so a free-running continuation stays aligned),
     * and the pre-threshold logits of ALL 255 tree nodes of sample k go to forced_logits[k*256 + node] */
    const unsigned char *forced_exc;
    float *forced_logits;
    long forced_pos, forced_cap;
} oracle_lpcnet_state;

int oracle_lpcnet_init(oracle_lpcnet_state *st)
{
    const oracle_lpcnet_model *m = st->m;
    unsigned char *te = st->trace_exc; float *tp = st->trace_pcm; long cap = st->trace_cap;
    const unsigned char *fe = st->forced_exc; float *fl = st->forced_logits; long fcap = st->forced_cap;
    memset(st, 0, sizeof(*st));
    st->m = m; st->trace_exc = te; st->trace_pcm = tp; st->trace_cap = cap;
    st->forced_exc = fe; st->forced_logits = fl; st->forced_cap = fcap;
    st->last_exc = lin2ulaw(0.f);
    kiss99_srand(&st->rng, (const unsigned char *)"LPCNet", 6);
    return 0;
}

oracle_lpcnet_state *oracle_lpcnet_create(const oracle_lpcnet_model *m)
{
    oracle_lpcnet_state *st = (oracle_lpcnet_state *)calloc(1, sizeof(*st));
    if (!st) return NULL;
    st->m = m;
    oracle_lpcnet_init(st);
    return st;
}

void oracle_lpcnet_destroy(oracle_lpcnet_state *st) { free(st); }

void oracle_lpcnet_set_trace(oracle_lpcnet_state *st, unsigned char *exc, float *pcm, long cap)
{
    st->trace_exc = exc; st->trace_pcm = pcm; st->trace_cap = cap; st->trace_pos = 0;
}

void oracle_lpcnet_set_forced(oracle_lpcnet_state *st, const unsigned char *exc, float *logits, long cap)
{
    st->forced_exc = exc; st->forced_logits = logits; st->forced_cap = cap; st->forced_pos = 0;
}

/* test hook: overwrite the recurrent state and the per-frame conditioning (teacher-forced single steps) */
void oracle_lpcnet_set_state(oracle_lpcnet_state *st, const float *gru_a_state, const float *gru_b_state,
                             const float *cond_a, const float *cond_b)
{
    const int N = st->m->h.gru_a, NB = st->m->h.gru_b;
    if (gru_a_state) memcpy(st->gru_a_state, gru_a_state, sizeof(float) * N);
    if (gru_b_state) memcpy(st->gru_b_state, gru_b_state, sizeof(float) * NB);
    if (cond_a) memcpy(st->gru_a_condition, cond_a, sizeof(float) * 3 * N);
    if (cond_b) memcpy(st->gru_b_condition, cond_b, sizeof(float) * 3 * NB);
}

/* ------------------------------------------------------------------------------------------------
 * freq.c: lpc_from_cepstrum
 * ---------------------------------------------------------------------------------------------- */
static float celt_lpc(float *lpc, const float *ac, int p)      /* celt_lpc.c _celt_lpc, float build */
{
    float r;
    float error = ac[0];
    memset(lpc, 0, sizeof(float) * p);
    if (ac[0] != 0) {
        for (int i = 0; i < p; i++) {
            float rr = 0;
            for (int j = 0; j < i; j++) rr += lpc[j] * ac[i - j];
            rr += ac[i + 1];
            r = -rr / error;
            lpc[i] = r;
            for (int j = 0; j < (i + 1) >> 1; j++) {
                float tmp1 = lpc[j], tmp2 = lpc[i - 1 - j];
                lpc[j] = tmp1 + r * tmp2;
                lpc[i - 1 - j] = tmp2 + r * tmp1;
            }
            error = error - (r * r) * error;
            if (error < .001f * ac[0]) break;
        }
    }
    return error;
}

float oracle_celt_lpc(float *lpc, const float *ac, int p) { return celt_lpc(lpc, ac, p); }

/* the host libm side of the device-pow self-test: freq.c `pow(10.f, Ex[i])*compensation[i]` stored to float */
void oracle_exp10_comp(const float *x, const float *comp, float *out, long n)
{
    for (long i = 0; i < n; ++i) out[i] = pow(10.f, x[i]) * comp[i];
}

void oracle_lpc_from_cepstrum(const oracle_lpcnet_model *m, float *lpc, const float *cepstrum)
{
    float tmp[NB_BANDS], Ex[NB_BANDS], Xr[FREQ_SIZE], ac[LPC_ORDER + 1];
    memcpy(tmp, cepstrum, sizeof(tmp));
    tmp[0] += 4;
    for (int i = 0; i < NB_BANDS; i++) {                         /* idct() */
        float sum = 0;
        for (int j = 0; j < NB_BANDS; j++) sum += tmp[j] * m->dct_table[i * NB_BANDS + j];
        Ex[i] = sum * sqrt(2. / NB_BANDS);
    }
    for (int i = 0; i < NB_BANDS; i++) Ex[i] = pow(10.f, Ex[i]) * compensation[i];
    memset(Xr, 0, sizeof(Xr));                                   /* interp_band_gain() */
    for (int i = 0; i < NB_BANDS - 1; i++) {
        int band_size = (eband5ms[i + 1] - eband5ms[i]) * 4;
        for (int j = 0; j < band_size; j++) {
            float frac = (float)j / band_size;
            Xr[eband5ms[i] * 4 + j] = (1 - frac) * Ex[i] + frac * Ex[i + 1];
        }
    }
    Xr[FREQ_SIZE - 1] = 0;
    /* inverse_transform() of the real, even spectrum, restated as a direct sum over the 17 lags used:
     * x[n] = X[0] + sum_{k=1}^{159} 2 X[k] cos(2 pi k n / 320)   (X[160] == 0) */
    for (int n = 0; n <= LPC_ORDER; n++) {
        float acc = Xr[0];
        for (int k = 1; k < FREQ_SIZE - 1; k++) acc += (2.f * Xr[k]) * m->cos_table[(k * n) % WINDOW_SIZE];
        ac[n] = acc;
    }
    ac[0] += ac[0] * 1e-4 + 320 / 12 / 38.;                      /* -40 dB noise floor */
    for (int i = 1; i < LPC_ORDER + 1; i++) ac[i] *= (1 - 6e-5 * i * i);   /* lag windowing */
    celt_lpc(lpc, ac, LPC_ORDER);
}

/* ------------------------------------------------------------------------------------------------
 * lpcnet.c: run_frame_network
 * ---------------------------------------------------------------------------------------------- */
void oracle_lpcnet_frame_network(oracle_lpcnet_state *st, const float *features)
{
    const oracle_lpcnet_model *m = st->m;
    const dss_blob_header *h = &m->h;
    float in[512], conv1_out[512], conv2_out[512], dense1_out[512], condition[512];
    int pitch = (int)floor(.1 + 50 * features[NB_BANDS] + 100);
    if (pitch < 33) pitch = 33;
    if (pitch > 255) pitch = 255;
    const int fin = h->nb_features + h->embed_pitch_dim;
    memcpy(in, features, sizeof(float) * h->nb_features);
    memcpy(in + h->nb_features, m->embed_pitch + (size_t)pitch * h->embed_pitch_dim,
           sizeof(float) * h->embed_pitch_dim);
    conv1d(m, m->conv1_w, m->conv1_b, fin, h->conv1_out, conv1_out, st->conv1_mem, in);
    if (st->frame_count < FEATURE_CONV1_DELAY) memset(conv1_out, 0, sizeof(float) * h->conv1_out);
    conv1d(m, m->conv2_w, m->conv2_b, h->conv1_out, h->conv2_out, conv2_out, st->conv2_mem, conv1_out);
    if (st->frame_count < FEATURES_DELAY) memset(conv2_out, 0, sizeof(float) * h->conv2_out);
    dense(m, m->dense1_w, m->dense1_b, h->conv2_out, h->dense1_out, ACT_TANH, dense1_out, conv2_out);
    dense(m, m->dense2_w, m->dense2_b, h->dense1_out, h->dense2_out, ACT_TANH, condition, dense1_out);
    dense(m, m->gru_a_dense_w, m->gru_a_dense_b, h->dense2_out, 3 * h->gru_a, ACT_LINEAR, st->gru_a_condition, condition);
    dense(m, m->gru_b_dense_w, m->gru_b_dense_b, h->dense2_out, 3 * h->gru_b, ACT_LINEAR, st->gru_b_condition, condition);
    memcpy(st->lpc, st->old_lpc[FEATURES_DELAY - 1], sizeof(st->lpc));
    memmove(st->old_lpc[1], st->old_lpc[0], (FEATURES_DELAY - 1) * sizeof(st->lpc));
    oracle_lpc_from_cepstrum(m, st->old_lpc[0], features);
    if (st->frame_count < 1000) st->frame_count++;
}

/* ------------------------------------------------------------------------------------------------
 * lpcnet.c: run_sample_network = compute_gru_a_input + compute_sparse_gru + compute_gruB + sample_mdense
 * ---------------------------------------------------------------------------------------------- */
static int run_sample_network(oracle_lpcnet_state *st, int last_exc, int last_sig, int pred)
{
    const oracle_lpcnet_model *m = st->m;
    const int N = m->h.gru_a, NB = m->h.gru_b;
    float gru_a_input[MAX_N], recur[MAX_N];

    /* compute_gru_a_input */
    for (int i = 0; i < 3 * N; i++)
        gru_a_input[i] = st->gru_a_condition[i] + m->embed_sig[(size_t)last_sig * 3 * N + i]
                         + m->embed_pred[(size_t)pred * 3 * N + i] + m->embed_exc[(size_t)last_exc * 3 * N + i];

    /* compute_sparse_gru */
    {
        float *state = st->gru_a_state;
        float *z = recur, *r = recur + N, *hh = recur + 2 * N;
        int k;
        const int recur_first = m->h.gru_a_order == DSS_GRUA_RECUR_FIRST;
        for (k = 0; k < 2; k++)
            for (int i = 0; i < N; i++) {
                recur[k * N + i] = m->gru_a_rbias[k * N + i] + m->gru_a_diag[k * N + i] * state[i];
                if (!recur_first) recur[k * N + i] = recur[k * N + i] + gru_a_input[k * N + i];   /* nnet.c 2021: input first */
            }
        for (; k < 3; k++)
            for (int i = 0; i < N; i++)
                recur[k * N + i] = m->gru_a_rbias[k * N + i] + m->gru_a_diag[k * N + i] * state[i];
        /* sparse_sgemv_accum8x4, float weights: blocks of 8 outputs x 4 inputs, input-major */
        const int32_t *idx = m->gru_a_idx;
        const float *w = m->gru_a_w;
        for (int i = 0; i < 3 * N; i += 8) {
            int cols = *idx++;
            for (int j = 0; j < cols; j++) {
                int pos = *idx++;
                float *y = &recur[i];
                for (int kk = 0; kk < 4; kk++) {
                    float xj = state[pos + kk];
                    for (int rr = 0; rr < 8; rr++) y[rr] += w[kk * 8 + rr] * xj;
                }
                w += 32;
            }
        }
        if (recur_first)                                     /* nnet.c 2019-2020: zrh = input; zrh += recur */
            for (int i = 0; i < 2 * N; i++) recur[i] = gru_a_input[i] + recur[i];
        for (int i = 0; i < 2 * N; i++) recur[i] = sigmoid_approx(m, recur[i]);
        for (int i = 0; i < N; i++) hh[i] = hh[i] * r[i] + gru_a_input[2 * N + i];
        for (int i = 0; i < N; i++) hh[i] = tanh_approx(m, hh[i]);
        for (int i = 0; i < N; i++) state[i] = z[i] * state[i] + (1 - z[i]) * hh[i];
    }

    /* compute_gruB: input = gru_a_state, condition added to the input bias */
    {
        float zrh[3 * 64], rec[3 * 64];
        float *state = st->gru_b_state;
        float *z = zrh, *r = zrh + NB, *hh = zrh + 2 * NB;
        for (int i = 0; i < 3 * NB; i++) zrh[i] = m->gru_b_bias[i] + st->gru_b_condition[i];
        sgemv_accum(zrh, m->gru_b_w_in, 3 * NB, N, 3 * NB, st->gru_a_state);
        for (int i = 0; i < 3 * NB; i++) rec[i] = m->gru_b_bias[3 * NB + i];
        sgemv_accum(rec, m->gru_b_w_rec, 3 * NB, NB, 3 * NB, state);
        for (int i = 0; i < 2 * NB; i++) zrh[i] += rec[i];
        for (int i = 0; i < 2 * NB; i++) zrh[i] = sigmoid_approx(m, zrh[i]);
        for (int i = 0; i < NB; i++) hh[i] += rec[2 * NB + i] * r[i];
        for (int i = 0; i < NB; i++) hh[i] = tanh_approx(m, hh[i]);
        for (int i = 0; i < NB; i++) hh[i] = z[i] * state[i] + (1 - z[i]) * hh[i];
        for (int i = 0; i < NB; i++) state[i] = hh[i];
    }

    /* sample_mdense: 8-level binary tree over the dual-FC outputs, thresholds from the RNG */
    {
        const int M = NB, Nout = m->h.dual_fc_out, stride = 2 * NB;
        float thresholds[8];
        int val = 0;
        for (int b = 0; b < 8; b += 4) {
            uint32_t r = kiss99_rand(&st->rng);
            thresholds[b] = m->sampling_logit_table[r & 0xFF];
            thresholds[b + 1] = m->sampling_logit_table[(r >> 8) & 0xFF];
            thresholds[b + 2] = m->sampling_logit_table[(r >> 16) & 0xFF];
            thresholds[b + 3] = m->sampling_logit_table[(r >> 24) & 0xFF];
        }
#define NODE_LOGIT(i, out) do {                                                             \
            float s1_ = m->dual_fc_bias[i];                                                \
            float s2_ = m->dual_fc_bias[(i) + Nout];                                       \
            for (int j = 0; j < M; j++) {                                                   \
                s1_ += m->dual_fc_w[(i) * stride + j] * st->gru_b_state[j];                \
                s2_ += m->dual_fc_w[(i) * stride + j + M] * st->gru_b_state[j];            \
            }                                                                               \
            s1_ = m->dual_fc_factor[i] * tanh_approx(m, s1_);                             \
            s2_ = m->dual_fc_factor[Nout + (i)] * tanh_approx(m, s2_);                    \
            s1_ += s2_;                                                                   \
            (out) = s1_;                                                                   \
        } while (0)
        for (int b = 0; b < 8; b++) {
            int i = (1 << b) | val;
            float sum1;
            NODE_LOGIT(i, sum1);
            int bit = thresholds[b] < sum1;
            val = (val << 1) | bit;
        }
        if (st->forced_exc && st->forced_pos < st->forced_cap) {     /* teacher forcing (tests only) */
            if (st->forced_logits) {
                float *lo = st->forced_logits + st->forced_pos * 256;
                lo[0] = 0.f;
                for (int i = 1; i < Nout; i++) NODE_LOGIT(i, lo[i]);
            }
            val = st->forced_exc[st->forced_pos++];
        }
#undef NODE_LOGIT
        return val;
    }
}

/* test hook: ONE run_sample_network step on the current state with the three embedding indices given;
 * returns the sampled excitation (the RNG advances as in synthesis) and, when logits256 != NULL, the
 * pre-threshold logits of all 255 tree nodes (index = node, [0] unused). */
int oracle_lpcnet_sample_step(oracle_lpcnet_state *st, int last_exc, int last_sig_ulaw, int pred_ulaw, float *logits256)
{
    const unsigned char *fe = st->forced_exc; float *fl = st->forced_logits; long fp = st->forced_pos, fc = st->forced_cap;
    unsigned char sink = 0;
    int exc;
    if (logits256) { st->forced_exc = &sink; st->forced_logits = logits256; st->forced_pos = 0; st->forced_cap = 1; }
    else st->forced_exc = NULL;
    /* with logits requested the walk's own result is replaced by `sink`; redo the walk from the logits */
    exc = run_sample_network(st, last_exc, last_sig_ulaw, pred_ulaw);
    st->forced_exc = fe; st->forced_logits = fl; st->forced_pos = fp; st->forced_cap = fc;
    return exc;
}

/* lpcnet.c: lpcnet_synthesize = run_frame_network + lpcnet_synthesize_tail_impl(preload = 0) */
void oracle_lpcnet_synthesize(oracle_lpcnet_state *st, const float *features, short *output, int N)
{
    const oracle_lpcnet_model *m = st->m;
    oracle_lpcnet_frame_network(st, features);
    if (st->frame_count <= FEATURES_DELAY) {
        memset(output, 0, sizeof(short) * N);
        return;
    }
    for (int i = 0; i < N; i++) {
        float pcm;
        int exc, last_sig_ulaw, pred_ulaw;
        float pred = 0;
        for (int j = 0; j < LPC_ORDER; j++) pred -= st->last_sig[j] * st->lpc[j];
        last_sig_ulaw = lin2ulaw(st->last_sig[0]);
        pred_ulaw = lin2ulaw(pred);
        exc = run_sample_network(st, st->last_exc, last_sig_ulaw, pred_ulaw);
        pcm = pred + m->ulaw2lin_table[exc];
        if (st->trace_exc && st->trace_pos < st->trace_cap) {
            st->trace_exc[st->trace_pos] = (unsigned char)exc;
            st->trace_pcm[st->trace_pos] = pcm;
            st->trace_pos++;
        }
        memmove(&st->last_sig[1], &st->last_sig[0], (LPC_ORDER - 1) * sizeof(float));
        st->last_sig[0] = pcm;
        st->last_exc = exc;
        pcm += PREEMPH * st->deemph_mem;
        st->deemph_mem = pcm;
        if (pcm < -32767) pcm = -32767;
        if (pcm > 32767) pcm = 32767;
        output[i] = (int)floor(.5 + pcm);
    }
}

/* Convenience for the cpu_baseline leg and tests: one fresh state, n_frames x 20 features in,
 * n_frames x 160 samples out (the loop of local/training.py:193-194). */
void oracle_lpcnet_synthesize_utterance(const oracle_lpcnet_model *m, const float *features, int n_frames,
                                        int feat_stride, short *pcm)
{
    oracle_lpcnet_state *st = oracle_lpcnet_create(m);
    for (int f = 0; f < n_frames; ++f)
        oracle_lpcnet_synthesize(st, features + (size_t)f * feat_stride, pcm + (size_t)f * FRAME_SIZE, FRAME_SIZE);
    oracle_lpcnet_destroy(st);
}

/* Debug taps used by parity tests of the frame-rate kernel. */
const float *oracle_lpcnet_tap(const oracle_lpcnet_state *st, int which)
{
    switch (which) {
    case 0: return st->gru_a_condition;
    case 1: return st->gru_b_condition;
    case 2: return st->lpc;
    case 3: return st->gru_a_state;
    case 4: return st->gru_b_state;
    }
    return NULL;
}
