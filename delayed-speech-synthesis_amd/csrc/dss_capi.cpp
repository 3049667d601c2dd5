// csrc/dss_capi.cpp -- host side of libdss_hip.so: the C ABI declared in include/dss_hip.h.
//
// Owns device memory, model upload and launch ordering; all arithmetic of the path runs in the HIP
// kernels (hga_kernels.hip, lpcnet_frame.hip, lpcnet_sample.hip).  The only numbers produced on the
// host are constant tables that xiph/LPCNet itself builds at run time with libm (lpcnet_init()'s
// sampling_logit_table, common.h's ulaw2lin over its 256 integer inputs, freq.c's dct table) and
// the final log() of dss_hga_log_power / dss_hga_extract's host-buffer form ("HGA log").
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <vector>

#include "dss_host.h"

// ------------------------------------------------------------------------------------------------------
// errors / device
// ------------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void dss_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *dss_last_error(void) { return g_err; }

static thread_local int g_device = -1;
// while upload_model() runs: the list every dev_upload() allocation is recorded in (so a model can be freed)
static thread_local std::vector<std::pair<int, void *>> *g_track = nullptr;

int dss_ensure_device(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        dss_set_error("no HIP device available (%s); libdss_hip has no CPU fallback",
                      e != hipSuccess ? hipGetErrorString(e) : "device count 0");
        return DSS_ENODEV;
    }
    if (g_device < 0) {
        const char *lr = getenv("LOCAL_RANK");
        int d = lr ? atoi(lr) : 0;
        g_device = (d >= 0 && d < n) ? d : 0;
    }
    DSS_HIP_CHECK(hipSetDevice(g_device));
    return DSS_OK;
}

extern "C" int dss_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int dss_set_device(int device)
{
    int n = dss_device_count();
    if (device < 0 || device >= n) { dss_set_error("device %d out of range (%d devices)", device, n); return DSS_EINVAL; }
    g_device = device;
    DSS_HIP_CHECK(hipSetDevice(device));
    return DSS_OK;
}

extern "C" int dss_current_device(void)
{
    if (dss_ensure_device()) return DSS_ENODEV;
    return g_device;
}

extern "C" const char *dss_version(void)
{
    static char buf[256];
    hipDeviceProp_t p;
    int n = dss_device_count();
    if (n > 0 && hipGetDeviceProperties(&p, g_device < 0 ? 0 : g_device) == hipSuccess)
        snprintf(buf, sizeof(buf), "libdss_hip 0.1 (gfx950 build) on %s %s, %d CUs", p.name, p.gcnArchName, p.multiProcessorCount);
    else
        snprintf(buf, sizeof(buf), "libdss_hip 0.1 (gfx950 build), no device");
    return buf;
}

template <typename T>
static int dev_upload(const T *host, size_t count, T **out)
{
    T *d = nullptr;
    DSS_HIP_CHECK(hipMalloc((void **)&d, count * sizeof(T) + 16));
    if (g_track) { int dv = 0; hipGetDevice(&dv); g_track->emplace_back(dv, (void *)d); }
    DSS_HIP_CHECK(hipMemcpy(d, host, count * sizeof(T), hipMemcpyHostToDevice));
    *out = d;
    return DSS_OK;
}

template <typename T>
static int dev_alloc(size_t count, T **out)
{
    T *d = nullptr;
    DSS_HIP_CHECK(hipMalloc((void **)&d, count * sizeof(T) + 16));
    DSS_HIP_CHECK(hipMemset(d, 0, count * sizeof(T)));
    *out = d;
    return DSS_OK;
}

// ------------------------------------------------------------------------------------------------------
// model: blob -> device
// ------------------------------------------------------------------------------------------------------
struct HostModel {
    std::vector<char> blob;
    dss_blob_header h;
    double bytes_per_sample = 0;
    int refs = 0;                      // decoder batches created from this model (guarded by g_model_mu)
    // per-device uploads
    std::vector<DssModelDev> dev;      // index = device id
    std::vector<char> dev_ready;
    std::vector<std::pair<int, void *>> dev_allocs;   // (device, pointer) of everything upload_model() allocated
};

static std::mutex g_model_mu;
static HostModel *g_model = nullptr;

// free a model's device memory and the host copy (caller holds g_model_mu; refs must be 0)
static void free_model(HostModel *hm)
{
    int cur = -1;
    hipGetDevice(&cur);
    for (auto &a : hm->dev_allocs) { hipSetDevice(a.first); hipFree(a.second); }
    if (cur >= 0) hipSetDevice(cur);
    delete hm;
}

static float host_ulaw2lin(float u)           // xiph common.h
{
    float s;
    float scale_1 = 32768.f / 255.f;
    u = u - 128.f;
    s = (u < 0) ? -1.f : 1.f;
    u = fabsf(u);
    return s * scale_1 * (exp(u / 128. * 5.5451774445f) - 1);
}

struct BlobView {
    const float *embed_pitch, *conv1_w, *conv1_b, *conv2_w, *conv2_b, *dense1_w, *dense1_b, *dense2_w, *dense2_b;
    const float *gru_a_dense_w, *gru_a_dense_b, *gru_b_dense_w, *gru_b_dense_b, *embed_sig, *embed_pred, *embed_exc;
    const float *gru_a_rbias, *gru_a_diag;
    const int32_t *gru_a_idx;
    const float *gru_a_w, *gru_b_bias, *gru_b_w_in, *gru_b_w_rec, *fc_bias, *fc_w, *fc_factor;
};

static int view_blob(const std::vector<char> &blob, const dss_blob_header &h, BlobView &v)
{
    const float *p = (const float *)(blob.data() + sizeof(dss_blob_header));
    const int fin = h.nb_features + h.embed_pitch_dim, NA3 = 3 * h.gru_a, NB3 = 3 * h.gru_b;
#define TAKE(f, c) do { v.f = p; p += (size_t)(c); } while (0)
    TAKE(embed_pitch, (size_t)h.pitch_max * h.embed_pitch_dim);
    TAKE(conv1_w, (size_t)3 * fin * h.conv1_out);          TAKE(conv1_b, h.conv1_out);
    TAKE(conv2_w, (size_t)3 * h.conv1_out * h.conv2_out);  TAKE(conv2_b, h.conv2_out);
    TAKE(dense1_w, (size_t)h.conv2_out * h.dense1_out);    TAKE(dense1_b, h.dense1_out);
    TAKE(dense2_w, (size_t)h.dense1_out * h.dense2_out);   TAKE(dense2_b, h.dense2_out);
    TAKE(gru_a_dense_w, (size_t)h.dense2_out * NA3);       TAKE(gru_a_dense_b, NA3);
    TAKE(gru_b_dense_w, (size_t)h.dense2_out * NB3);       TAKE(gru_b_dense_b, NB3);
    TAKE(embed_sig, (size_t)256 * NA3); TAKE(embed_pred, (size_t)256 * NA3); TAKE(embed_exc, (size_t)256 * NA3);
    TAKE(gru_a_rbias, NA3); TAKE(gru_a_diag, NA3);
    v.gru_a_idx = (const int32_t *)p; p += h.sparse_idx_len;
    TAKE(gru_a_w, (size_t)h.sparse_nblocks * 32);
    TAKE(gru_b_bias, 2 * NB3); TAKE(gru_b_w_in, (size_t)h.gru_a * NB3); TAKE(gru_b_w_rec, (size_t)h.gru_b * NB3);
    TAKE(fc_bias, 2 * h.dual_fc_out); TAKE(fc_w, (size_t)h.dual_fc_out * 2 * h.gru_b); TAKE(fc_factor, 2 * h.dual_fc_out);
#undef TAKE
    if ((size_t)((const char *)p - blob.data()) != blob.size()) {
        dss_set_error("weight blob length %zu does not match its header", blob.size());
        return DSS_EINVAL;
    }
    return DSS_OK;
}

static int check_header(const dss_blob_header &h)
{
    if (memcmp(h.magic, DSS_BLOB_MAGIC, 8) != 0 || h.version != 1) { dss_set_error("not a DSSLPCN1 v1 blob"); return DSS_EINVAL; }
    // the kernels are specialised for the published LPCNet dimensions
    if (h.nb_features != 20 || h.nb_bands != 18 || h.embed_pitch_dim != 64 || h.pitch_max != 256 || h.conv1_out != 128 ||
        h.conv2_out != 128 || h.dense1_out != 128 || h.dense2_out != 128 || h.gru_a != DSS_GRU_A || h.gru_b != DSS_GRU_B ||
        h.dual_fc_out != DSS_FC_OUT || h.lpc_order != DSS_LPC_ORDER) {
        dss_set_error("blob dimensions differ from the LPCNet architecture this build is specialised for "
                      "(features 20, conv/dense 128, GRU A 384, GRU B 16, dual FC 256)");
        return DSS_EINVAL;
    }
    if (h.gru_a_order != DSS_GRUA_INPUT_FIRST && h.gru_a_order != DSS_GRUA_RECUR_FIRST) {
        dss_set_error("blob header: gru_a_order %d is neither 0 (input first) nor 1 (recurrent first)", h.gru_a_order);
        return DSS_EINVAL;
    }
    return DSS_OK;
}

// parse + validate a blob into a new HostModel (no lock, no device work)
static int parse_blob(const void *blob, size_t len, HostModel **out)
{
    if (!blob || len < sizeof(dss_blob_header)) { dss_set_error("blob too short"); return DSS_EINVAL; }
    HostModel *hm = new HostModel;
    memcpy(&hm->h, blob, sizeof(hm->h));
    int rc = check_header(hm->h);
    if (rc) { delete hm; return rc; }
    hm->blob.assign((const char *)blob, (const char *)blob + len);
    BlobView v;
    rc = view_blob(hm->blob, hm->h, v);
    if (rc) { delete hm; return rc; }
    // validate the sparse index (host-side shape check before any kernel trusts it)
    {
        const int groups = 3 * hm->h.gru_a / 8;
        long pos = 0, blocks = 0;
        for (int g = 0; g < groups; ++g) {
            if (pos >= hm->h.sparse_idx_len) { dss_set_error("sparse idx truncated"); delete hm; return DSS_EINVAL; }
            int cnt = v.gru_a_idx[pos++];
            if (cnt < 0 || pos + cnt > hm->h.sparse_idx_len) { dss_set_error("sparse idx corrupt"); delete hm; return DSS_EINVAL; }
            for (int j = 0; j < cnt; ++j) {
                int p = v.gru_a_idx[pos++];
                if (p < 0 || p + 4 > hm->h.gru_a || (p & 3)) { dss_set_error("sparse idx position %d invalid", p); delete hm; return DSS_EINVAL; }
            }
            blocks += cnt;
        }
        if (pos != hm->h.sparse_idx_len || blocks != hm->h.sparse_nblocks) { dss_set_error("sparse idx/blocks mismatch"); delete hm; return DSS_EINVAL; }
    }
    const int na = hm->h.gru_a, nb = hm->h.gru_b;
    const double floats = 3.0 * (3 * na) + (double)hm->h.sparse_nblocks * 32 + 3 * na + 3.0 * nb * (na + nb) + 2.0 * nb * 8 + 16;
    hm->bytes_per_sample = 4.0 * floats + 2.0 + 80.0 / 160.0;
    *out = hm;
    return DSS_OK;
}

// caller holds g_model_mu
static int load_blob_locked(const void *blob, size_t len)
{
    HostModel *hm = nullptr;
    int rc = parse_blob(blob, len, &hm);
    if (rc) return rc;
    // an earlier model stays alive exactly as long as decoder batches created from it exist (they hold its device
    // pointers); with none left it is freed here, otherwise when its last batch is destroyed
    if (g_model && g_model->refs == 0) free_model(g_model);
    g_model = hm;
    return DSS_OK;
}

extern "C" int dss_lpcnet_load_model(const void *blob, size_t len)
{
    HostModel *hm = nullptr;
    int rc = parse_blob(blob, len, &hm);                    // the slow part outside the lock
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g_model_mu);
    if (g_model && g_model->refs == 0) free_model(g_model);
    g_model = hm;
    return DSS_OK;
}

static int read_file(const char *path, std::vector<char> &buf)
{
    FILE *f = fopen(path, "rb");
    if (!f) { dss_set_error("cannot open %s", path); return DSS_EINVAL; }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    buf.resize((size_t)(n > 0 ? n : 0));
    size_t got = n > 0 ? fread(buf.data(), 1, (size_t)n, f) : 0;
    fclose(f);
    if (n < 0 || got != (size_t)n) { dss_set_error("short read on %s", path); return DSS_EINVAL; }
    return DSS_OK;
}

extern "C" int dss_lpcnet_load_model_file(const char *path)
{
    std::vector<char> buf;
    int rc = read_file(path, buf);
    if (rc) return rc;
    return dss_lpcnet_load_model(buf.data(), buf.size());
}

extern "C" double dss_lpcnet_bytes_per_sample(void)
{
    std::lock_guard<std::mutex> lk(g_model_mu);
    return g_model ? g_model->bytes_per_sample : 0.0;
}

// ---- CU-resident layout of the sample-rate kernel (lpcnet_sample.hip): pure host work, no device calls -----------
struct FastLayout {
    int fast_ok = 0, zmax = 0, hmax = 0, zr_cap = 10, ext = 0, ext_tab = 0, hfloats = 0;
    std::vector<int> unit_of, unit_h, wave_nh, grp_hoff, wave_nzr, wave_nzt;
    std::vector<float> zr_w, hblk;
    std::vector<unsigned> zr_col, h_col;
};

static void build_fast_layout(const BlobView &v, int NA, FastLayout &F)
{
    const int G = NA / 8;                       // 48 row groups per gate
    std::vector<int> cnt(3 * G), start(3 * G), blk0(3 * G);
    long pos = 0, blk = 0;
    for (int g = 0; g < 3 * G; ++g) {
        cnt[g] = v.gru_a_idx[pos]; start[g] = (int)pos + 1; blk0[g] = (int)blk;
        pos += 1 + cnt[g]; blk += cnt[g];
    }
    int fast_ok = (NA == 384 && G == 48);
    int zmax = 0, hmax = 0;
    for (int g = 0; g < G; ++g) {
        zmax = std::max(zmax, std::max(cnt[g], cnt[G + g]));
        hmax = std::max(hmax, cnt[2 * G + g]);
    }
    if (hmax > DSS_HCX + DSS_HX) fast_ok = 0;
    // Two independent lane assignments, both "8 row groups per wave":
    //  * h-gate chains (LDS resident): groups sorted by h block count, so each wave's loop length (its
    //    largest group) is close to what all its groups need;
    //  * z/r chains (register resident): groups sorted by max(z, r) block count.  The 16 heaviest groups go to
    //    waves 4 and 5, whose code path carries no dual-FC weights and therefore has room for zr_cap register
    //    slots per gate; waves 0..3 run the 8-slot instantiation.  Blocks beyond a wave's register slots (models
    //    with skewed sparsity) stay in idx order behind them as "tail" records in LDS.
    // Both the h chain and the z/r block products run between barriers B and C, under the GRU B relay; waves 4
    // and 5 also run the speculation there, so they get the lightest h chains, and among waves 0..3 the heavier
    // z/r groups go with the lighter h chains.
    // The per-unit pre-activation of the h gate travels from its h lane to its z/r lane through LDS.
    std::vector<int> order_h(G), order_zr(G);
    for (int g = 0; g < G; ++g) order_h[g] = order_zr[g] = g;
    std::stable_sort(order_h.begin(), order_h.end(), [&](int a, int b2) { return cnt[2 * G + a] > cnt[2 * G + b2]; });
    std::stable_sort(order_zr.begin(), order_zr.end(), [&](int a, int b2) {
        return std::max(cnt[a], cnt[G + a]) > std::max(cnt[b2], cnt[G + b2]);
    });
    // Register slots per gate on waves 4, 5.  A model that fits the register slots as it is runs the 10-slot
    // instantiation (no spills) or, with 11 or 12 blocks in some group, the 12-slot one (22 spilled registers, ~4 %
    // slower).  Any other model runs the 10-slot layout with tails: measured faster than 12 slots + tails
    // (tools/model_fit.py), the spills cost more than the two extra tail blocks.
    const int zr17 = std::max(cnt[order_zr[16]], cnt[G + order_zr[16]]);      // heaviest group that lands on waves 0..3
    const bool plain = zmax <= DSS_ZRC && zr17 <= 8 && hmax <= DSS_HC;
    const int zr_cap = (plain && zmax > 10) ? DSS_ZRC : 10;
    int rank_wave_h[6] = {0, 1, 3, 2, 5, 4};
    static const int rank_wave_zr[6] = {4, 5, 2, 3, 1, 0};
    if (const char *e = getenv("DSS_RANK_WAVE_H")) {        // development switch (A/B timing of the h-chain assignment): a permutation of 0..5
        int p[6], seen = 0;
        if (sscanf(e, "%d,%d,%d,%d,%d,%d", &p[0], &p[1], &p[2], &p[3], &p[4], &p[5]) == 6) {
            for (int k = 0; k < 6; ++k) if (p[k] >= 0 && p[k] < 6) seen |= 1 << p[k];
            if (seen == 63) for (int k = 0; k < 6; ++k) rank_wave_h[k] = p[k];
        }
    }
    std::vector<int> grp_h(G, 0), grp_zr(G, 0);
    std::vector<int> &unit_of = F.unit_of, &unit_h = F.unit_h, &wave_nh = F.wave_nh, &grp_hoff = F.grp_hoff, &wave_nzr = F.wave_nzr, &wave_nzt = F.wave_nzt;
    unit_of.assign(NA, 0); unit_h.assign(NA, 0); wave_nh.assign(8, 0); grp_hoff.assign(G, 0); wave_nzr.assign(8, 0); wave_nzt.assign(8, 0);
    int hfloats = 0, ext = 0;
    for (int rk = 0; rk < 6 && fast_ok; ++rk) {
        int nh = 0, nzr = 0;
        for (int q = 0; q < 8; ++q) {
            grp_h[rank_wave_h[rk] * 8 + q] = order_h[rk * 8 + q];
            grp_zr[rank_wave_zr[rk] * 8 + q] = order_zr[rk * 8 + q];
            nh = std::max(nh, cnt[2 * G + order_h[rk * 8 + q]]);
            nzr = std::max(nzr, std::max(cnt[order_zr[rk * 8 + q]], cnt[G + order_zr[rk * 8 + q]]));
        }
        const int cap = rank_wave_zr[rk] < 4 ? 8 : zr_cap;
        wave_nh[rank_wave_h[rk]] = (nh + 1) & ~1;       // the kernel tests for the end of a list every 2 slots
        wave_nzr[rank_wave_zr[rk]] = std::min((nzr + 1) & ~1, cap);
        wave_nzt[rank_wave_zr[rk]] = std::max(0, nzr - cap);
        if (nzr - cap > DSS_ZR_TAIL) fast_ok = 0;
        if (nzr > cap || nh > DSS_HC) ext = 1;
    }
    // The h-gate image: every row group's own records back to back (128 bytes = [8 rows][4 inputs] per block), no
    // padding to the wave's longest list.  A wave still runs wave_nh slots on all its lanes: a lane whose group is
    // shorter reads on into the next group's records and multiplies them by "column 96", four zeros behind the
    // state vector, so the extra terms are +-0.  One spare record goes between two groups of a wave whenever
    // they would otherwise start an even number of records apart: 8-lane groups that start 32 banks apart keep
    // the wave's ds_read_b128 of its block records conflict-free.
    int hend = 0;
    for (int wv = 0; wv < 6 && fast_ok; ++wv)
        for (int q = 0; q < 8; ++q) {
            if (q && (((hfloats - grp_hoff[wv * 8 + q - 1]) / 32) & 1) == 0) hfloats += 32;
            grp_hoff[wv * 8 + q] = hfloats;
            hfloats += cnt[2 * G + grp_h[wv * 8 + q]] * 32;
            hend = std::max(hend, grp_hoff[wv * 8 + q] + (wave_nh[wv] + 4) * 32);      // + 4: the kernel fetches two chunks of two slots ahead
        }
    hfloats = std::max(hfloats, hend);                  // the last groups' over-reads stay inside the image
    // Extended paths (models with skewed sparsity only): behind the h records, the z and r tail lists of every
    // group of the z/r assignment (same over-read convention), then a table
    //   int   tail_off[48][2]                float offset of the group's z list and of its r list
    //   uint8 tail_col[48][2][DSS_ZR_TAIL]   block column of every tail slot (96 = unused)
    //   uint8 hx_col[48][DSS_HX]             block column of h slots DSS_HCX.. of the group of the h assignment
    const int ext_tab_floats = (G * 2 * 4 + G * 2 * DSS_ZR_TAIL + G * DSS_HX) / 4 + 4;     // + 16 B: the kernel reads columns one trip ahead
    std::vector<int> tail_off(G * 2, 0);
    int ext_tab = 0;
    if (fast_ok && ext) {
        int tend = hfloats;
        for (int wv = 0; wv < 6; ++wv)
            for (int q = 0; q < 8; ++q)
                for (int gate = 0; gate < 2; ++gate) {
                    const int cap = wv < 4 ? 8 : zr_cap, n = cnt[gate * G + grp_zr[wv * 8 + q]];
                    tail_off[(wv * 8 + q) * 2 + gate] = hfloats;
                    tend = std::max(tend, hfloats + wave_nzt[wv] * 32);
                    hfloats += std::max(0, n - cap) * 32;
                }
        hfloats = std::max(hfloats, tend);
        ext_tab = hfloats;
        hfloats += ext_tab_floats;
    }
    if ((size_t)hfloats * sizeof(float) > DSS_HBLK_BYTES) fast_ok = 0;
    std::vector<float> &zr_w = F.zr_w, &hblk = F.hblk;
    std::vector<unsigned> &zr_col = F.zr_col, &h_col = F.h_col;
    zr_w.assign((size_t)2 * DSS_ZRC * 4 * NA, 0.f); hblk.assign((size_t)std::max(hfloats, 4), 0.f);
    zr_col.assign((size_t)(2 * DSS_ZRC / 4) * NA, 0u); h_col.assign((size_t)(DSS_HCX / 4) * NA, 0u);
    if (fast_ok)
        for (int tid = 0; tid < NA; ++tid) {
            const int wv = tid / 64, l = tid & 63, q = l / 8, r = l & 7;
            {
                const int grp = grp_zr[wv * 8 + q];
                unit_of[tid] = grp * 8 + r;
                const int cap = wv < 4 ? 8 : zr_cap;
                for (int gate = 0; gate < 2; ++gate) {
                    const int g = gate * G + grp;
                    for (int sl = 0; sl < std::min(cnt[g], cap); ++sl) {
                        const int s2 = gate * DSS_ZRC + sl;
                        const float *wb = v.gru_a_w + (size_t)(blk0[g] + sl) * 32;
                        for (int k = 0; k < 4; ++k) zr_w[((size_t)s2 * 4 + k) * NA + tid] = wb[k * 8 + r];
                        zr_col[(size_t)(s2 >> 2) * NA + tid] |= (unsigned)(v.gru_a_idx[start[g] + sl] / 4) << (8 * (s2 & 3));
                    }
                    for (int sl = cap; sl < cnt[g]; ++sl) {              // tail: LDS records, columns in the table
                        const float *wb = v.gru_a_w + (size_t)(blk0[g] + sl) * 32;
                        float *rec = hblk.data() + tail_off[(wv * 8 + q) * 2 + gate] + (size_t)(sl - cap) * 32 + r * 4;
                        for (int k = 0; k < 4; ++k) rec[k] = wb[k * 8 + r];
                    }
                }
            }
            {
                const int grp = grp_h[wv * 8 + q], g = 2 * G + grp;
                unit_h[tid] = grp * 8 + r;
                for (int sl = 0; sl < cnt[g]; ++sl) {
                    const float *wb = v.gru_a_w + (size_t)(blk0[g] + sl) * 32;
                    float *rec = hblk.data() + grp_hoff[wv * 8 + q] + (size_t)sl * 32 + r * 4;
                    for (int k = 0; k < 4; ++k) rec[k] = wb[k * 8 + r];
                    if (sl < DSS_HCX) h_col[(size_t)(sl >> 2) * NA + tid] |= (unsigned)(v.gru_a_idx[start[g] + sl] / 4) << (8 * (sl & 3));
                }
                for (int sl = cnt[g]; sl < DSS_HCX; ++sl) h_col[(size_t)(sl >> 2) * NA + tid] |= 96u << (8 * (sl & 3));
            }
        }
    if (fast_ok && ext) {
        int *toff = reinterpret_cast<int *>(hblk.data() + ext_tab);
        unsigned char *tcol = reinterpret_cast<unsigned char *>(toff + G * 2);
        unsigned char *hxc = tcol + (size_t)G * 2 * DSS_ZR_TAIL;
        memset(tcol, 96, (size_t)G * 2 * DSS_ZR_TAIL + (size_t)G * DSS_HX);
        for (int wv = 0; wv < 6; ++wv)
            for (int q = 0; q < 8; ++q) {
                const int cap = wv < 4 ? 8 : zr_cap;
                for (int gate = 0; gate < 2; ++gate) {
                    const int g = gate * G + grp_zr[wv * 8 + q];
                    toff[(wv * 8 + q) * 2 + gate] = tail_off[(wv * 8 + q) * 2 + gate];
                    for (int sl = cap; sl < cnt[g]; ++sl)
                        tcol[((size_t)(wv * 8 + q) * 2 + gate) * DSS_ZR_TAIL + (sl - cap)] = (unsigned char)(v.gru_a_idx[start[g] + sl] / 4);
                }
                const int gh = 2 * G + grp_h[wv * 8 + q];
                for (int sl = DSS_HCX; sl < cnt[gh]; ++sl)
                    hxc[(size_t)(wv * 8 + q) * DSS_HX + (sl - DSS_HCX)] = (unsigned char)(v.gru_a_idx[start[gh] + sl] / 4);
            }
    }
    F.fast_ok = fast_ok; F.zmax = zmax; F.hmax = hmax; F.zr_cap = zr_cap; F.ext = fast_ok ? ext : 0; F.ext_tab = ext_tab; F.hfloats = hfloats;

}

static int upload_model(HostModel *hm, int device, DssModelDev &m)
{
    BlobView v;
    int rc = view_blob(hm->blob, hm->h, v);
    if (rc) return rc;
    const dss_blob_header &h = hm->h;
    memset(&m, 0, sizeof(m));
    m.h = h;
    const int fin = h.nb_features + h.embed_pitch_dim, NA = h.gru_a, NA3 = 3 * NA, NB3 = 3 * h.gru_b;
#define UP(field, src, count) do { float *d; rc = dev_upload<float>(src, (size_t)(count), &d); if (rc) return rc; m.field = d; } while (0)
    UP(embed_pitch, v.embed_pitch, (size_t)h.pitch_max * h.embed_pitch_dim);
    UP(conv1_w, v.conv1_w, (size_t)3 * fin * 128);   UP(conv1_b, v.conv1_b, 128);
    UP(conv2_w, v.conv2_w, (size_t)3 * 128 * 128);   UP(conv2_b, v.conv2_b, 128);
    UP(dense1_w, v.dense1_w, 128 * 128);             UP(dense1_b, v.dense1_b, 128);
    UP(dense2_w, v.dense2_w, 128 * 128);             UP(dense2_b, v.dense2_b, 128);
    UP(gru_a_dense_w, v.gru_a_dense_w, (size_t)128 * NA3);  UP(gru_a_dense_b, v.gru_a_dense_b, NA3);
    UP(gru_b_dense_w, v.gru_b_dense_w, (size_t)128 * NB3);  UP(gru_b_dense_b, v.gru_b_dense_b, NB3);
    UP(embed_sig, v.embed_sig, (size_t)256 * NA3);
    UP(embed_pred, v.embed_pred, (size_t)256 * NA3);
    UP(embed_exc, v.embed_exc, (size_t)256 * NA3);
    UP(gru_a_rbias, v.gru_a_rbias, NA3);
    UP(gru_a_diag, v.gru_a_diag, NA3);
    UP(gru_b_bias, v.gru_b_bias, 2 * NB3);
    UP(gru_b_w_in, v.gru_b_w_in, (size_t)NA * NB3);
    UP(gru_b_w_rec, v.gru_b_w_rec, (size_t)h.gru_b * NB3);
    UP(fc_bias, v.fc_bias, 2 * h.dual_fc_out);
    UP(fc_w, v.fc_w, (size_t)h.dual_fc_out * 2 * h.gru_b);
    UP(fc_factor, v.fc_factor, 2 * h.dual_fc_out);

    // ---- sparse GRU A: per gate, per unit, a padded list of (pos, 4 weights) slots in idx order ----------
    {
        const int groups_per_gate = NA / 8;
        std::vector<int> grp_start(3 * groups_per_gate), grp_cnt(3 * groups_per_gate), grp_blk(3 * groups_per_gate);
        long pos = 0, blk = 0;
        for (int g = 0; g < 3 * groups_per_gate; ++g) {
            grp_cnt[g] = v.gru_a_idx[pos];
            grp_start[g] = (int)pos + 1;
            grp_blk[g] = (int)blk;
            pos += 1 + grp_cnt[g];
            blk += grp_cnt[g];
        }
        int zr_slots = 0;
        for (int g = 0; g < 2 * groups_per_gate; ++g) zr_slots = std::max(zr_slots, grp_cnt[g]);
        for (int gate = 0; gate < 3; ++gate) {
            int slots = 0;
            for (int g = 0; g < groups_per_gate; ++g) slots = std::max(slots, grp_cnt[gate * groups_per_gate + g]);
            // the generic kernel keeps 16 blocks in flight, unconditionally: z and r lists share one length (multiple of 8),
            // the h list is a multiple of 16; the padding is zero blocks at input 0
            slots = gate < 2 ? ((std::max(zr_slots, 1) + 7) & ~7) : ((std::max(slots, 1) + 15) & ~15);
            std::vector<int> pos4((size_t)slots * NA, 0);
            std::vector<float> w((size_t)slots * 4 * NA, 0.f);
            for (int unit = 0; unit < NA; ++unit) {
                const int g = gate * groups_per_gate + unit / 8, r = unit & 7;
                for (int sl = 0; sl < grp_cnt[g]; ++sl) {
                    pos4[(size_t)sl * NA + unit] = v.gru_a_idx[grp_start[g] + sl] * 4;
                    const float *wb = v.gru_a_w + (size_t)(grp_blk[g] + sl) * 32;
                    for (int k = 0; k < 4; ++k) w[((size_t)sl * 4 + k) * NA + unit] = wb[k * 8 + r];
                }
            }
            int *dpos; float *dw;
            rc = dev_upload<int>(pos4.data(), pos4.size(), &dpos); if (rc) return rc;
            rc = dev_upload<float>(w.data(), w.size(), &dw); if (rc) return rc;
            m.gate[gate].slots = slots; m.gate[gate].pos4 = dpos; m.gate[gate].w = dw;
        }
    }
    // ---- CU-resident layout of the sample-rate kernel (lpcnet_sample.hip) ---------------------------------
    {
        FastLayout F;
        build_fast_layout(v, NA, F);
        const std::vector<int> &unit_of = F.unit_of, &unit_h = F.unit_h, &wave_nh = F.wave_nh, &grp_hoff = F.grp_hoff, &wave_nzr = F.wave_nzr, &wave_nzt = F.wave_nzt;
        const std::vector<float> &zr_w = F.zr_w, &hblk = F.hblk;
        const std::vector<unsigned> &zr_col = F.zr_col, &h_col = F.h_col;
        const int fast_ok = F.fast_ok, zmax = F.zmax, hmax = F.hmax, zr_cap = F.zr_cap, ext = F.ext, ext_tab = F.ext_tab, hfloats = F.hfloats;
        m.fast_ok = fast_ok;
        m.nzr_max = (zmax + 1) & ~1;
        m.zr_cap = zr_cap;
        m.hmax = hmax;
        m.ext = fast_ok ? ext : 0;
        m.ext_tab = ext_tab;
        m.hblk_floats = hfloats;
        int *di; float *df; unsigned *du;
        rc = dev_upload<int>(unit_of.data(), unit_of.size(), &di); if (rc) return rc; m.unit_of = di;
        {   // embedding rows permuted into lane order, the three gates of a lane's unit side by side
            const float *tabs[3] = {v.embed_sig, v.embed_pred, v.embed_exc};
            std::vector<float> perm((size_t)256 * NA * 3);
            for (int t = 0; t < 3; ++t) {
                for (int idx = 0; idx < 256; ++idx)
                    for (int tid = 0; tid < NA; ++tid)
                        for (int g = 0; g < 3; ++g)
                            perm[((size_t)idx * NA + tid) * 3 + g] = tabs[t][(size_t)idx * 3 * NA + (size_t)g * NA + unit_of[tid]];
                rc = dev_upload<float>(perm.data(), perm.size(), &df); if (rc) return rc; m.embed_lane[t] = df;
            }
        }
        rc = dev_upload<int>(unit_h.data(), unit_h.size(), &di); if (rc) return rc; m.unit_h = di;
        rc = dev_upload<int>(wave_nh.data(), wave_nh.size(), &di); if (rc) return rc; m.wave_nh = di;
        rc = dev_upload<int>(grp_hoff.data(), grp_hoff.size(), &di); if (rc) return rc; m.grp_hoff = di;
        rc = dev_upload<int>(wave_nzr.data(), wave_nzr.size(), &di); if (rc) return rc; m.wave_nzr = di;
        rc = dev_upload<int>(wave_nzt.data(), wave_nzt.size(), &di); if (rc) return rc; m.wave_nzt = di;
        rc = dev_upload<float>(zr_w.data(), zr_w.size(), &df); if (rc) return rc; m.zr_w = df;
        rc = dev_upload<unsigned>(zr_col.data(), zr_col.size(), &du); if (rc) return rc; m.zr_col = du;
        rc = dev_upload<unsigned>(h_col.data(), h_col.size(), &du); if (rc) return rc; m.h_col = du;
        rc = dev_upload<float>(hblk.data(), hblk.size(), &df); if (rc) return rc; m.hblk = df;
        // GRU B input weights for the two relay waves, j-major with lane = row: [384][64]
        std::vector<float> gbl((size_t)NA * 64, 0.f);
        for (int j = 0; j < NA; ++j)
            for (int row = 0; row < NB3; ++row) gbl[(size_t)j * 64 + row] = v.gru_b_w_in[(size_t)j * NB3 + row];
        rc = dev_upload<float>(gbl.data(), gbl.size(), &df); if (rc) return rc; m.gb_w_lane = df;
        // ... and four inputs of a lane side by side, for the pair kernel's relay waves, which stream them from L2
        std::vector<float> gbq((size_t)NA * 64, 0.f);
        for (int j = 0; j < NA; ++j)
            for (int row = 0; row < NB3; ++row) gbq[((size_t)(j / 4) * 64 + row) * 4 + (j & 3)] = v.gru_b_w_in[(size_t)j * NB3 + row];
        rc = dev_upload<float>(gbq.data(), gbq.size(), &df); if (rc) return rc; m.gb_w_quad = df;
        // dual-FC weights for the pair kernel, whose dual-FC waves load their node's 32 weights every sample instead of
        // holding them in registers: [k 8][node 256][4] = (layer 0, layer 1) weights of inputs 2k and 2k+1, so that load k
        // of a wave reads 1 KB of consecutive bytes
        std::vector<float> fcp((size_t)DSS_FC_OUT * DSS_GRU_B * 2, 0.f);
        for (int node = 0; node < DSS_FC_OUT; ++node)
            for (int j = 0; j < DSS_GRU_B; ++j) {
                const size_t o = ((size_t)(j / 2) * DSS_FC_OUT + node) * 4 + (j & 1) * 2;
                fcp[o + 0] = v.fc_w[(size_t)node * 2 * DSS_GRU_B + j];
                fcp[o + 1] = v.fc_w[(size_t)node * 2 * DSS_GRU_B + DSS_GRU_B + j];
            }
        rc = dev_upload<float>(fcp.data(), fcp.size(), &df); if (rc) return rc; m.fc_w_pair = df;
    }
    // ---- derived tables (host libm, exactly as xiph builds them at run time) ---------------------------
    {
        float tansig[201], logit[256], u2l[256], dct[18 * 18], costab[320], ia[160], ib[160];
        int iband[160];
        double lagw[17];
        for (int i = 0; i < 201; ++i) tansig[i] = (float)(floor(tanh(.04 * i) * 1e6 + .5) / 1e6);   // tansig_table.h
        for (int i = 0; i < 256; ++i) {                                                              // lpcnet_init()
            float prob = .025 + .95 * i / 255.;
            logit[i] = -log((1 - prob) / prob);
            u2l[i] = host_ulaw2lin((float)i);
        }
        for (int i = 0; i < 18; ++i)                                                                  // freq.c check_init()
            for (int j = 0; j < 18; ++j) {
                dct[i * 18 + j] = cos((i + .5) * j * M_PI / 18);
                if (j == 0) dct[i * 18 + j] *= sqrt(.5);
            }
        for (int i = 0; i < 320; ++i) costab[i] = (float)cos(2. * M_PI * i / 320);
        static const int eband5ms[18] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 34, 40};
        for (int i = 0; i < 17; ++i) {                                                                // interp_band_gain()
            const int band_size = (eband5ms[i + 1] - eband5ms[i]) * 4;
            for (int j = 0; j < band_size; ++j) {
                const float frac = (float)j / band_size;
                ia[eband5ms[i] * 4 + j] = 1 - frac;
                ib[eband5ms[i] * 4 + j] = frac;
                iband[eband5ms[i] * 4 + j] = i;
            }
        }
        for (int i = 0; i < 17; ++i) lagw[i] = (1 - 6e-5 * i * i);
        float *d; int *di; double *dd;
        rc = dev_upload<float>(tansig, 201, &d); if (rc) return rc; m.tansig = d;
        rc = dev_upload<float>(logit, 256, &d); if (rc) return rc; m.logit_table = d;
        rc = dev_upload<float>(u2l, 256, &d); if (rc) return rc; m.ulaw2lin = d;
        rc = dev_upload<float>(dct, 324, &d); if (rc) return rc; m.dct_table = d;
        rc = dev_upload<float>(costab, 320, &d); if (rc) return rc; m.cos_table = d;
        {
            std::vector<float> ckl((size_t)160 * 17);
            for (int k = 0; k < 160; ++k)
                for (int lag = 0; lag < 17; ++lag) ckl[(size_t)k * 17 + lag] = costab[(k * lag) % 320];
            rc = dev_upload<float>(ckl.data(), ckl.size(), &d); if (rc) return rc; m.cos_kl = d;
        }
        rc = dev_upload<float>(ia, 160, &d); if (rc) return rc; m.interp_a = d;
        rc = dev_upload<float>(ib, 160, &d); if (rc) return rc; m.interp_b = d;
        rc = dev_upload<int>(iband, 160, &di); if (rc) return rc; m.interp_band = di;
        rc = dev_upload<double>(lagw, 17, &dd); if (rc) return rc; m.lag_window = dd;
    }
#undef UP
    (void)device;
    return DSS_OK;
}

// Returns the current model and its copy on the calling thread's device.  With acquire set, the model's reference count
// is taken while g_model_mu is still held, so a concurrent dss_lpcnet_load_model() cannot free it in between; the caller
// then owns one reference (release_model()).
static int load_blob_locked(const void *blob, size_t len);

static int get_model(HostModel **out_hm, const DssModelDev **out, bool acquire)
{
    int rc = dss_ensure_device();
    if (rc) return rc;
    std::unique_lock<std::mutex> lk(g_model_mu);
    if (!g_model) {
        const char *path = getenv("DSS_LPCNET_WEIGHTS");
        if (!path) { dss_set_error("no LPCNet weights: call dss_lpcnet_load_model() or set DSS_LPCNET_WEIGHTS"); return DSS_ENOMODEL; }
        std::vector<char> buf;
        rc = read_file(path, buf);
        if (rc) return rc;
        rc = load_blob_locked(buf.data(), buf.size());
        if (rc) return rc;
    }
    HostModel *hm = g_model;
    const int ndev = dss_device_count();
    if ((int)hm->dev.size() < ndev) { hm->dev.resize(ndev); hm->dev_ready.resize(ndev, 0); }
    if (!hm->dev_ready[g_device]) {
        g_track = &hm->dev_allocs;
        rc = upload_model(hm, g_device, hm->dev[g_device]);
        g_track = nullptr;
        if (rc) return rc;
        hm->dev_ready[g_device] = 1;
    }
    if (acquire) hm->refs++;
    *out_hm = hm;
    *out = &hm->dev[g_device];
    return DSS_OK;
}

static void release_model(HostModel *hm)
{
    std::lock_guard<std::mutex> lk(g_model_mu);
    if (hm && --hm->refs == 0 && hm != g_model) free_model(hm);     // a superseded model dies with its last user
}

// ------------------------------------------------------------------------------------------------------
// batched decoder
// ------------------------------------------------------------------------------------------------------
struct dss_lpcnet_batch {
    int device;
    HostModel *host_model;        // keeps the model (and its device copy) alive
    const DssModelDev *model;
    DssBatchDev d;
    int last_utts = 0, last_frames = 0;
    int force_utts = 0, force_frames = 0;   // shape force_exc / trace_logits were sized for (dss_lpcnet_batch_force_excitation)
    int trace = 0, timing = 0;
    int pair = 0;                 // 0 auto, -1 never, 2 always: two utterances per workgroup (dss_lpcnet_batch_set_multi)
    int max_rows = 0;             // rows one call may carry (= scratch rows); d.max_utts = decoder slots (a lane: its parent's)
    dss_lpcnet_batch *parent = nullptr;   // a lane (dss_lpcnet_batch_create_lane): decoder state aliases the parent's arrays
    int lanes = 0;                // live lanes of this (parent) batch
    bool dead = false;            // destroyed by the caller while lanes were alive: freed with the last lane
    float *d_feat = nullptr;      // staging for the host-buffer entry point
    short *d_pcm = nullptr;
    int *d_slots = nullptr;       // [max_rows] slot list of a ragged call
    int *d_counts = nullptr;      // [max_rows] frame counts of a ragged call
    int *d_order = nullptr;       // [max_rows] dispatch order of a ragged call: rows by decreasing frame count
    DssPinnedRing meta;           // pinned staging of those three lists ([3][max_rows] ints per slot of the ring)
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    double ms_sum[2] = {0, 0};
    int ms_n = 0;
};

// per-call scratch (rows x frames), the staging buffers and the events: what a plain batch and a lane both own
static int batch_alloc_scratch(dss_lpcnet_batch *b, int max_rows, int max_frames)
{
    DssBatchDev &d = b->d;
    b->max_rows = max_rows; d.max_frames = max_frames;
    const size_t B = max_rows, F = max_frames;
    int rc = 0;
    rc |= dev_alloc<float>(B * (F + 2) * 84, &d.in_buf);
    rc |= dev_alloc<float>(B * (F + 2) * 128, &d.c1_buf);
    rc |= dev_alloc<float>(B * F * 128, &d.c2_buf);
    rc |= dev_alloc<float>(B * F * 128, &d.d1_buf);
    rc |= dev_alloc<float>(B * F * 128, &d.cond_buf);
    rc |= dev_alloc<float>(B * (F + 2) * 16, &d.lpc_buf);
    rc |= dev_alloc<float>(B * F * DSS_COND_STRIDE, &d.frame_out);
    rc |= dev_alloc<int>(B, &d.fc0);
    rc |= dev_alloc<float>(B * F * 20, &b->d_feat);
    rc |= dev_alloc<short>(B * F * DSS_FRAME_SIZE, &b->d_pcm);
    rc |= dev_alloc<int>(B, &b->d_slots);
    rc |= dev_alloc<int>(B, &b->d_counts);
    rc |= dev_alloc<int>(B, &b->d_order);
    rc |= b->meta.init(3 * B);
    for (int i = 0; i < 3; ++i) rc |= (hipEventCreate(&b->ev[i]) != hipSuccess);
    return rc;
}

extern "C" dss_lpcnet_batch *dss_lpcnet_batch_create(int max_utts, int max_frames)
{
    if (max_utts <= 0 || max_frames <= 0) { dss_set_error("batch dims must be positive"); return nullptr; }
    HostModel *hm; const DssModelDev *m;
    if (get_model(&hm, &m, true)) return nullptr;            // holds one reference from here on (dropped by destroy)
    dss_lpcnet_batch *b = new dss_lpcnet_batch;
    memset(&b->d, 0, sizeof(b->d));
    b->device = g_device;
    b->host_model = hm;
    b->model = m;
    DssBatchDev &d = b->d;
    d.max_utts = max_utts;
    const size_t B = max_utts;
    int rc = 0;
    rc |= dev_alloc<float>(B * DSS_GRU_A, &d.gru_a_state);
    rc |= dev_alloc<float>(B * DSS_GRU_B, &d.gru_b_state);
    rc |= dev_alloc<float>(B * 16, &d.last_sig);
    rc |= dev_alloc<int>(B, &d.last_exc);
    rc |= dev_alloc<float>(B, &d.deemph);
    rc |= dev_alloc<uint32_t>(B * 4, &d.rng);
    rc |= dev_alloc<int>(B, &d.frame_count);
    rc |= dev_alloc<float>(B * 2 * 84, &d.conv1_mem);
    rc |= dev_alloc<float>(B * 2 * 128, &d.conv2_mem);
    rc |= dev_alloc<float>(B * 2 * 16, &d.old_lpc);
    rc |= batch_alloc_scratch(b, max_utts, max_frames);
    if (rc) {
        dss_set_error("device allocation failed for batch %d x %d", max_utts, max_frames);
        dss_lpcnet_batch_destroy(b);
        return nullptr;
    }
    if (dss_launch_lpcnet_reset(*m, d, -1, 0) || hipDeviceSynchronize() != hipSuccess) { dss_lpcnet_batch_destroy(b); return nullptr; }
    return b;
}

// A lane: a second launch context on the decoder states of `parent`.  It owns scratch for max_rows x max_frames and nothing
// else; its rows name the parent's slots (ragged calls with a slot list).  Lanes exist so that calls touching DIFFERENT slots
// can be in flight on different streams at once (the asynchronous segment synthesis of the gated streaming mode).
extern "C" dss_lpcnet_batch *dss_lpcnet_batch_create_lane(dss_lpcnet_batch *parent, int max_rows, int max_frames)
{
    if (!parent || parent->parent || parent->dead || max_rows <= 0 || max_frames <= 0) {
        dss_set_error("dss_lpcnet_batch_create_lane: needs a live batch that is not itself a lane, and positive dims");
        return nullptr;
    }
    if (hipSetDevice(parent->device) != hipSuccess) { dss_set_error("hipSetDevice failed"); return nullptr; }
    dss_lpcnet_batch *b = new dss_lpcnet_batch;
    b->d = parent->d;                              // the state arrays (and max_utts = the slot count) are the parent's
    DssBatchDev &d = b->d;
    d.in_buf = d.c1_buf = d.c2_buf = d.d1_buf = d.cond_buf = d.lpc_buf = d.frame_out = nullptr;
    d.fc0 = nullptr; d.slot_of = d.count_of = d.row_of = nullptr; d.utt0 = 0;
    d.trace_exc = d.trace_pcm = d.trace_logits = nullptr; d.force_exc = nullptr;
    b->device = parent->device;
    b->host_model = parent->host_model;
    b->model = parent->model;
    b->pair = parent->pair;
    b->parent = parent;
    parent->lanes += 1;
    if (batch_alloc_scratch(b, max_rows, max_frames)) {
        dss_set_error("device allocation failed for lane %d x %d", max_rows, max_frames);
        dss_lpcnet_batch_destroy(b);
        return nullptr;
    }
    return b;
}

static void batch_free(dss_lpcnet_batch *b)
{
    DssBatchDev &d = b->d;
    if (!b->parent) {
        void *state[] = {d.gru_a_state, d.gru_b_state, d.last_sig, d.last_exc, d.deemph, d.rng, d.frame_count, d.conv1_mem,
                         d.conv2_mem, d.old_lpc};
        for (void *p : state) if (p) hipFree(p);
    }
    void *ptrs[] = {d.in_buf, d.c1_buf, d.c2_buf, d.d1_buf, d.cond_buf, d.lpc_buf, d.frame_out,
                    d.fc0, d.trace_exc, d.trace_pcm, d.trace_logits, (void *)d.force_exc, b->d_feat, b->d_pcm, b->d_slots, b->d_counts, b->d_order};
    for (void *p : ptrs) if (p) hipFree(p);
    b->meta.destroy();
    for (int i = 0; i < 3; ++i) if (b->ev[i]) hipEventDestroy(b->ev[i]);
    if (!b->parent) release_model(b->host_model);
    delete b;
}

extern "C" void dss_lpcnet_batch_destroy(dss_lpcnet_batch *b)
{
    if (!b) return;
    hipSetDevice(b->device);
    hipDeviceSynchronize();                        // nothing of this object may still be in flight on any stream
    if (b->parent) {
        dss_lpcnet_batch *p = b->parent;
        batch_free(b);
        if (--p->lanes == 0 && p->dead) batch_free(p);
        return;
    }
    if (b->lanes > 0) { b->dead = true; return; }  // its lanes still run on its state: freed with the last of them
    batch_free(b);
}

extern "C" int dss_lpcnet_batch_reset(dss_lpcnet_batch *b, int utt)
{
    if (!b || utt >= b->d.max_utts) { dss_set_error("bad batch/utt"); return DSS_EINVAL; }
    DSS_HIP_CHECK(hipSetDevice(b->device));
    int rc = dss_launch_lpcnet_reset(*b->model, b->d, utt, 0);
    if (rc) return rc;
    DSS_HIP_CHECK(hipStreamSynchronize(0));
    return DSS_OK;
}

extern "C" int dss_lpcnet_batch_reset_async(dss_lpcnet_batch *b, int utt, void *hip_stream)
{
    if (!b || utt >= b->d.max_utts) { dss_set_error("bad batch/utt"); return DSS_EINVAL; }
    DSS_HIP_CHECK(hipSetDevice(b->device));
    return dss_launch_lpcnet_reset(*b->model, b->d, utt, (hipStream_t)hip_stream);
}

extern "C" int dss_lpcnet_batch_enable_trace(dss_lpcnet_batch *b, int on)
{
    if (!b) return DSS_EINVAL;
    DSS_HIP_CHECK(hipSetDevice(b->device));
    if (on && !b->d.trace_exc) {
        const size_t n = (size_t)b->max_rows * b->d.max_frames * DSS_FRAME_SIZE;
        int rc = dev_alloc<float>(n, &b->d.trace_exc);
        rc |= dev_alloc<float>(n, &b->d.trace_pcm);
        if (rc) return DSS_ENOMEM;
    }
    b->trace = on;      // 1 = excitation/pcm trace, 2 = diagnostic phase stamps (development only)
    return DSS_OK;
}

extern "C" int dss_lpcnet_batch_force_excitation(dss_lpcnet_batch *b, const unsigned char *exc, int n_utts, int n_frames)
{
    if (!b) return DSS_EINVAL;
    DSS_HIP_CHECK(hipSetDevice(b->device));
    if (!exc) {                                   // back to free running
        if (b->d.force_exc) hipFree((void *)b->d.force_exc);
        if (b->d.trace_logits) hipFree(b->d.trace_logits);
        b->d.force_exc = nullptr; b->d.trace_logits = nullptr;
        b->force_utts = b->force_frames = 0;
        return DSS_OK;
    }
    if (n_utts <= 0 || n_utts > b->max_rows || n_frames <= 0 || n_frames > b->d.max_frames) {
        dss_set_error("forced excitation shape out of range"); return DSS_EINVAL;
    }
    if (!b->trace) { dss_set_error("teacher forcing needs dss_lpcnet_batch_enable_trace(b, 1 or 17) first"); return DSS_EINVAL; }
    const size_t n = (size_t)n_utts * n_frames * DSS_FRAME_SIZE;
    if (b->d.force_exc) hipFree((void *)b->d.force_exc);
    if (b->d.trace_logits) hipFree(b->d.trace_logits);
    b->d.force_exc = nullptr; b->d.trace_logits = nullptr;
    b->force_utts = b->force_frames = 0;
    unsigned char *de = nullptr;
    if (dev_upload<unsigned char>(exc, n, &de)) return DSS_ENOMEM;
    b->d.force_exc = de;
    if (dev_alloc<float>(n * 256, &b->d.trace_logits)) return DSS_ENOMEM;
    b->force_utts = n_utts; b->force_frames = n_frames;
    return DSS_OK;
}

extern "C" int dss_lpcnet_model_info(int *fast_path, int *zr_slots_max, int *h_slots_max, int *h_lds_bytes, int *gru_a_order)
{
    HostModel *hm; const DssModelDev *m;
    int rc = get_model(&hm, &m, true);
    if (rc) return rc;
    int hmax = 0;
    // recomputed from the blob (the device struct keeps only what the kernels need)
    {
        BlobView v;
        if (view_blob(hm->blob, hm->h, v)) { release_model(hm); return DSS_EINVAL; }
        const int G = hm->h.gru_a / 8;
        long pos = 0;
        for (int g = 0; g < 3 * G; ++g) { const int c = v.gru_a_idx[pos]; if (g >= 2 * G) hmax = std::max(hmax, c); pos += 1 + c; }
    }
    if (fast_path) *fast_path = m->fast_ok ? (m->ext ? 2 : 1) : 0;
    if (zr_slots_max) *zr_slots_max = m->nzr_max;
    if (h_slots_max) *h_slots_max = hmax;
    if (h_lds_bytes) *h_lds_bytes = m->hblk_floats * 4;
    if (gru_a_order) *gru_a_order = hm->h.gru_a_order;
    release_model(hm);
    return DSS_OK;
}

extern "C" int dss_lpcnet_batch_set_multi(dss_lpcnet_batch *b, int utterances_per_workgroup)
{
    if (!b) return DSS_EINVAL;
    const int u = utterances_per_workgroup;
    if (!(u == 0 || u == -1 || u == 1 || u == 2)) { dss_set_error("utterances per workgroup: 0 (auto), 1 or -1 (always one), 2 (always two)"); return DSS_EINVAL; }
    if (u == 2 && !dss_pair_fits(*b->model)) { dss_set_error("two utterances per workgroup do not fit beside this model in LDS (or it needs the extended paths)"); return DSS_EINVAL; }
    b->pair = u == 1 ? -1 : u;
    return DSS_OK;
}

extern "C" int dss_lpcnet_batch_enable_timing(dss_lpcnet_batch *b, int on)
{
    if (!b) return DSS_EINVAL;
    b->timing = on ? 1 : 0;
    b->ms_sum[0] = b->ms_sum[1] = 0; b->ms_n = 0;
    return DSS_OK;
}

extern "C" double dss_lpcnet_batch_kernel_ms(dss_lpcnet_batch *b, int which)
{
    if (!b || b->ms_n == 0 || which < 0 || which > 1) return 0.0;
    double v = b->ms_sum[which] / b->ms_n;
    if (which == 0) { /* keep accumulating until both have been read */ }
    return v;
}

static int check_batch_shape(dss_lpcnet_batch *b, int n_utts, int n_frames, int feat_stride)
{
    if (b->dead) { dss_set_error("this batch was destroyed (only its lanes are alive)"); return DSS_EINVAL; }
    if (n_utts <= 0 || n_utts > b->max_rows || n_frames <= 0 || n_frames > b->d.max_frames || feat_stride < DSS_NB_FEATURES) {
        dss_set_error("shape out of range: %d utts (max %d), %d frames (max %d), stride %d", n_utts, b->max_rows, n_frames,
                      b->d.max_frames, feat_stride);
        return DSS_EINVAL;
    }
    return DSS_OK;
}

// frame-rate network, then the persistent sample-rate kernel, on stream s; b->d.slot_of / count_of select the
// uniform (NULL) or the ragged form
static int run_batch(dss_lpcnet_batch *b, const float *d_features, int n_utts, int n_frames, int feat_stride, short *d_pcm,
                     hipStream_t s)
{
    // the kernels index the forced excitation and the logit trace with the CALL's shape: it must be the shape they were
    // sized for, and a uniform call (the trace build would launch a ragged one as if every row were full)
    if (b->d.force_exc || b->d.trace_logits) {
        if (n_utts != b->force_utts || n_frames != b->force_frames || b->d.slot_of || b->d.count_of) {
            dss_set_error("teacher forcing was set up for %d x %d frames, uniform calls only; this call is %d x %d%s",
                          b->force_utts, b->force_frames, n_utts, n_frames, (b->d.slot_of || b->d.count_of) ? " (ragged)" : "");
            return DSS_EINVAL;
        }
    }
    if (b->timing) DSS_HIP_CHECK(hipEventRecord(b->ev[0], s));
    int rc = dss_launch_frame_network(*b->model, b->d, d_features, n_utts, n_frames, feat_stride, s);
    if (rc) return rc;
    if (b->timing) DSS_HIP_CHECK(hipEventRecord(b->ev[1], s));
    rc = dss_launch_sample_network(*b->model, b->d, n_utts, n_frames, d_pcm, b->trace, b->pair, s);
    if (rc) return rc;
    if (b->timing) {
        DSS_HIP_CHECK(hipEventRecord(b->ev[2], s));
        DSS_HIP_CHECK(hipEventSynchronize(b->ev[2]));
        float fr = 0, sm = 0;
        DSS_HIP_CHECK(hipEventElapsedTime(&fr, b->ev[0], b->ev[1]));
        DSS_HIP_CHECK(hipEventElapsedTime(&sm, b->ev[1], b->ev[2]));
        b->ms_sum[0] += sm; b->ms_sum[1] += fr; b->ms_n += 1;
    }
    b->last_utts = n_utts; b->last_frames = n_frames;
    return DSS_OK;
}

extern "C" int dss_lpcnet_batch_synthesize_dev(dss_lpcnet_batch *b, const float *d_features, int n_utts, int n_frames,
                                               int feat_stride, short *d_pcm, void *hip_stream)
{
    if (!b || !d_features || !d_pcm) { dss_set_error("null argument"); return DSS_EINVAL; }
    int rc = check_batch_shape(b, n_utts, n_frames, feat_stride);
    if (rc) return rc;
    DSS_HIP_CHECK(hipSetDevice(b->device));
    b->d.slot_of = nullptr; b->d.count_of = nullptr; b->d.row_of = nullptr;
    return run_batch(b, d_features, n_utts, n_frames, feat_stride, d_pcm, (hipStream_t)hip_stream);
}

// Validate and upload the slot list / frame counts of a ragged call (either may be NULL).  The lists go through a pinned
// ring (dss_host.h): the upload neither waits for what is queued on `s` nor can a later call overwrite it before it has run.
static int stage_ragged(dss_lpcnet_batch *b, const int *slots, const int *counts, int n_utts, int n_frames, hipStream_t s)
{
    b->d.slot_of = nullptr; b->d.count_of = nullptr; b->d.row_of = nullptr;
    if (slots) {
        std::string seen((size_t)b->d.max_utts, 0);
        for (int i = 0; i < n_utts; ++i) {
            if (slots[i] < 0 || slots[i] >= b->d.max_utts) { dss_set_error("row %d: slot %d out of range (max %d)", i, slots[i], b->d.max_utts); return DSS_EINVAL; }
            if (seen[slots[i]]) { dss_set_error("row %d: slot %d appears twice in one call (a decoder is sequential)", i, slots[i]); return DSS_EINVAL; }
            seen[slots[i]] = 1;
        }
    } else if (n_utts > b->d.max_utts) {
        dss_set_error("%d rows without a slot list, %d decoder slots", n_utts, b->d.max_utts); return DSS_EINVAL;
    }
    if (counts)
        for (int i = 0; i < n_utts; ++i)
            if (counts[i] < 0 || counts[i] > n_frames) { dss_set_error("row %d: %d frames outside [0, %d]", i, counts[i], n_frames); return DSS_EINVAL; }
    if (!slots && !counts) return DSS_OK;
    int *h = b->meta.acquire();
    if (!h) { dss_set_error("pinned staging ring failed"); return DSS_ENODEV; }
    const size_t R = (size_t)b->max_rows;
    if (slots) {
        memcpy(h, slots, sizeof(int) * n_utts);
        DSS_HIP_CHECK(hipMemcpyAsync(b->d_slots, h, sizeof(int) * n_utts, hipMemcpyHostToDevice, s));
        b->d.slot_of = b->d_slots;
    }
    if (counts) {
        memcpy(h + R, counts, sizeof(int) * n_utts);
        DSS_HIP_CHECK(hipMemcpyAsync(b->d_counts, h + R, sizeof(int) * n_utts, hipMemcpyHostToDevice, s));
        b->d.count_of = b->d_counts;
        // Dispatch order: workgroups start in grid order, so the longest rows go first whatever order the caller used
        // (and the pair kernel's two rows of a workgroup are neighbours in length).  Results do not depend on it.
        int *order = h + 2 * R;
        for (int i = 0; i < n_utts; ++i) order[i] = i;
        std::stable_sort(order, order + n_utts, [&](int x, int y) { return counts[x] > counts[y]; });
        DSS_HIP_CHECK(hipMemcpyAsync(b->d_order, order, sizeof(int) * n_utts, hipMemcpyHostToDevice, s));
        b->d.row_of = b->d_order;
    }
    return b->meta.commit(s);
}

extern "C" int dss_lpcnet_batch_synthesize_ragged_dev(dss_lpcnet_batch *b, const float *d_features, const int *slots,
                                                      const int *counts, int n_utts, int n_frames, int feat_stride,
                                                      short *d_pcm, void *hip_stream)
{
    if (!b || !d_features || !d_pcm) { dss_set_error("null argument"); return DSS_EINVAL; }
    int rc = check_batch_shape(b, n_utts, n_frames, feat_stride);
    if (rc) return rc;
    DSS_HIP_CHECK(hipSetDevice(b->device));
    hipStream_t s = (hipStream_t)hip_stream;
    rc = stage_ragged(b, slots, counts, n_utts, n_frames, s);
    if (rc) return rc;
    return run_batch(b, d_features, n_utts, n_frames, feat_stride, d_pcm, s);
}

extern "C" int dss_lpcnet_batch_synthesize_ragged(dss_lpcnet_batch *b, const float *features, const int *slots,
                                                  const int *counts, int n_utts, int n_frames, int feat_stride, short *pcm)
{
    if (!b || !features || !pcm) { dss_set_error("null argument"); return DSS_EINVAL; }
    int rc = check_batch_shape(b, n_utts, n_frames, feat_stride);
    if (rc) return rc;
    DSS_HIP_CHECK(hipSetDevice(b->device));
    rc = stage_ragged(b, slots, counts, n_utts, n_frames, nullptr);
    if (rc) return rc;
    DSS_HIP_CHECK(hipMemcpy2D(b->d_feat, DSS_NB_FEATURES * sizeof(float), features, (size_t)feat_stride * sizeof(float),
                              DSS_NB_FEATURES * sizeof(float), (size_t)n_utts * n_frames, hipMemcpyHostToDevice));
    rc = run_batch(b, b->d_feat, n_utts, n_frames, DSS_NB_FEATURES, b->d_pcm, nullptr);
    if (rc) return rc;
    DSS_HIP_CHECK(hipMemcpy(pcm, b->d_pcm, (size_t)n_utts * n_frames * DSS_FRAME_SIZE * sizeof(short), hipMemcpyDeviceToHost));
    return DSS_OK;
}

extern "C" int dss_lpcnet_batch_synthesize(dss_lpcnet_batch *b, const float *features, int n_utts, int n_frames,
                                           int feat_stride, short *pcm)
{
    if (!b || !features || !pcm) { dss_set_error("null argument"); return DSS_EINVAL; }
    if (check_batch_shape(b, n_utts, n_frames, feat_stride)) return DSS_EINVAL;
    DSS_HIP_CHECK(hipSetDevice(b->device));
    // pack the first 20 floats of every row (feature files carry 36, LPCNet.pyx:97,115)
    DSS_HIP_CHECK(hipMemcpy2D(b->d_feat, DSS_NB_FEATURES * sizeof(float), features, (size_t)feat_stride * sizeof(float),
                              DSS_NB_FEATURES * sizeof(float), (size_t)n_utts * n_frames, hipMemcpyHostToDevice));
    int rc = dss_lpcnet_batch_synthesize_dev(b, b->d_feat, n_utts, n_frames, DSS_NB_FEATURES, b->d_pcm, nullptr);
    if (rc) return rc;
    DSS_HIP_CHECK(hipMemcpy(pcm, b->d_pcm, (size_t)n_utts * n_frames * DSS_FRAME_SIZE * sizeof(short), hipMemcpyDeviceToHost));
    return DSS_OK;
}

extern "C" int dss_lpcnet_batch_tap(dss_lpcnet_batch *b, int utt, int which, float *out, size_t n_floats)
{
    if (!b || !out || utt < 0 || utt >= b->last_utts) { dss_set_error("bad tap arguments"); return DSS_EINVAL; }
    DSS_HIP_CHECK(hipSetDevice(b->device));
    DSS_HIP_CHECK(hipDeviceSynchronize());
    const int F = b->last_frames;
    if (which >= 0 && which <= 2) {
        const int width = which == 0 ? 3 * DSS_GRU_A : which == 1 ? 3 * DSS_GRU_B : DSS_LPC_ORDER;
        const int off = which == 0 ? 0 : which == 1 ? 3 * DSS_GRU_A : 3 * DSS_GRU_A + 3 * DSS_GRU_B;
        if (n_floats < (size_t)F * width) { dss_set_error("tap buffer too small"); return DSS_EINVAL; }
        DSS_HIP_CHECK(hipMemcpy2D(out, width * sizeof(float), b->d.frame_out + (size_t)utt * F * DSS_COND_STRIDE + off,
                                  DSS_COND_STRIDE * sizeof(float), width * sizeof(float), F, hipMemcpyDeviceToHost));
        return DSS_OK;
    }
    if (which == 5 && b->d.trace_logits) {
        const size_t n = (size_t)F * DSS_FRAME_SIZE * 256;
        if (n_floats < n) { dss_set_error("tap buffer too small"); return DSS_EINVAL; }
        DSS_HIP_CHECK(hipMemcpy(out, b->d.trace_logits + (size_t)utt * n, n * sizeof(float), hipMemcpyDeviceToHost));
        return DSS_OK;
    }
    if ((which == 3 || which == 4) && b->d.trace_exc) {
        const size_t n = (size_t)F * DSS_FRAME_SIZE;
        if (n_floats < n) { dss_set_error("tap buffer too small"); return DSS_EINVAL; }
        const float *src = (which == 3 ? b->d.trace_exc : b->d.trace_pcm) + (size_t)utt * n;
        DSS_HIP_CHECK(hipMemcpy(out, src, n * sizeof(float), hipMemcpyDeviceToHost));
        return DSS_OK;
    }
    dss_set_error("unknown tap %d (or trace not enabled)", which);
    return DSS_EINVAL;
}

// ------------------------------------------------------------------------------------------------------
// xiph drop-in symbols: one state = a batch of one utterance, one frame per call
// ------------------------------------------------------------------------------------------------------
// One frame per call is a fixed sequence of ten small launches between two tiny copies (what an unchanged decode_online.py
// pays per 10 ms, local/units.py:531-538).  From its second call on, a state replays that sequence from a HIP graph captured
// once on a stream of its own: pinned staging for the 80 bytes in and the 320 bytes out, one hipGraphLaunch, one wait.
struct LPCNetState {
    dss_lpcnet_batch *b;
    hipStream_t stream = nullptr;
    hipGraphExec_t exec = nullptr;
    float *h_feat = nullptr;          // pinned
    short *h_pcm = nullptr;           // pinned
    long calls = 0;
    int graph_off = 0;                // capture failed once (or DSS_LEVEL1_EAGER is set): eager launches from then on
};

extern "C" LPCNetState *lpcnet_create(void)
{
    dss_lpcnet_batch *b = dss_lpcnet_batch_create(1, 1);
    if (!b) return nullptr;
    LPCNetState *st = new LPCNetState;
    st->b = b;
    st->graph_off = getenv("DSS_LEVEL1_EAGER") != nullptr;
    return st;
}

extern "C" int lpcnet_init(LPCNetState *st)
{
    if (!st) return -1;
    if (st->stream) hipStreamSynchronize(st->stream);
    return dss_lpcnet_batch_reset(st->b, -1);
}

extern "C" void lpcnet_destroy(LPCNetState *st)
{
    if (!st) return;
    hipSetDevice(st->b->device);
    if (st->stream) hipStreamSynchronize(st->stream);
    if (st->exec) hipGraphExecDestroy(st->exec);
    if (st->stream) hipStreamDestroy(st->stream);
    if (st->h_feat) hipHostFree(st->h_feat);
    if (st->h_pcm) hipHostFree(st->h_pcm);
    dss_lpcnet_batch_destroy(st->b);
    delete st;
}

// one frame through the captured graph; DSS_OK, or an error after which the caller falls back to the eager path for good
#define DSS_EINTERNAL_REPLAY (-1000)      // the replay itself failed (the frame was enqueued): see lpcnet_synthesize
static int level1_graph_frame(LPCNetState *st, const float *features, short *output)
{
    dss_lpcnet_batch *b = st->b;
    DSS_HIP_CHECK(hipSetDevice(b->device));
    if (!st->stream) {
        DSS_HIP_CHECK(hipStreamCreateWithFlags(&st->stream, hipStreamNonBlocking));
        DSS_HIP_CHECK(hipHostMalloc((void **)&st->h_feat, DSS_NB_FEATURES * sizeof(float), hipHostMallocDefault));
        DSS_HIP_CHECK(hipHostMalloc((void **)&st->h_pcm, DSS_FRAME_SIZE * sizeof(short), hipHostMallocDefault));
    }
    memcpy(st->h_feat, features, DSS_NB_FEATURES * sizeof(float));
    if (!st->exec) {
        hipGraph_t graph = nullptr;
        DSS_HIP_CHECK(hipStreamBeginCapture(st->stream, hipStreamCaptureModeThreadLocal));
        hipError_t e1 = hipMemcpyAsync(b->d_feat, st->h_feat, DSS_NB_FEATURES * sizeof(float), hipMemcpyHostToDevice, st->stream);
        b->d.slot_of = nullptr; b->d.count_of = nullptr; b->d.row_of = nullptr;
        const int rc = e1 == hipSuccess ? run_batch(b, b->d_feat, 1, 1, DSS_NB_FEATURES, b->d_pcm, st->stream) : DSS_ENODEV;
        hipError_t e2 = hipMemcpyAsync(st->h_pcm, b->d_pcm, DSS_FRAME_SIZE * sizeof(short), hipMemcpyDeviceToHost, st->stream);
        hipError_t e3 = hipStreamEndCapture(st->stream, &graph);             // always ends the capture
        if (rc || e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || !graph) {
            if (graph) hipGraphDestroy(graph);
            (void)hipGetLastError();
            dss_set_error("level-1 graph capture failed; staying on eager launches");
            return DSS_ENODEV;
        }
        hipError_t e4 = hipGraphInstantiate(&st->exec, graph, nullptr, nullptr, 0);
        hipGraphDestroy(graph);
        if (e4 != hipSuccess) { st->exec = nullptr; dss_set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e4)); return DSS_ENODEV; }
    }
    // From here on the frame IS enqueued: a failure must not be answered with an eager re-run (the decoder state may have
    // advanced already), so it gets its own code: the caller zero-fills this frame and stays eager afterwards.
    if (hipGraphLaunch(st->exec, st->stream) != hipSuccess || hipStreamSynchronize(st->stream) != hipSuccess) {
        dss_set_error("level-1 graph replay failed: %s", hipGetErrorString(hipGetLastError()));
        return DSS_EINTERNAL_REPLAY;
    }
    memcpy(output, st->h_pcm, DSS_FRAME_SIZE * sizeof(short));
    b->last_utts = 1; b->last_frames = 1;
    return DSS_OK;
}

// The xiph ABI gives this call no error channel (void; cLPCNet.pxd:13) and its caller is a live prosthesis loop
// (local/units.py:534-535): never abort the process.  On failure the frame is ZERO-FILLED (silence), the reason is
// kept in dss_last_error(), counted in dss_error_count(), and printed to stderr the first time and then every 1000th.
static std::atomic<long> g_synth_failures{0};

extern "C" long dss_error_count(void) { return g_synth_failures.load(); }

static void synth_failed(short *output, int N)
{
    if (output && N > 0) memset(output, 0, sizeof(short) * (size_t)N);
    const long k = g_synth_failures.fetch_add(1);
    if (k == 0 || k % 1000 == 0) fprintf(stderr, "libdss_hip: lpcnet_synthesize failed (%ld so far), frame zero-filled: %s\n", k + 1, g_err);
}

extern "C" void lpcnet_synthesize(LPCNetState *st, const float *features, short *output, int N)
{
    if (!st || !features || !output) { dss_set_error("lpcnet_synthesize: null argument"); synth_failed(output, N); return; }
    if (N != DSS_FRAME_SIZE) {           // the reference only ever asks for one 160-sample frame (LPCNet.pyx:39)
        dss_set_error("lpcnet_synthesize: N must be %d, got %d", DSS_FRAME_SIZE, N);
        synth_failed(output, N);
        return;
    }
    // first call of a state: eager (it also sets the kernels' attributes); traced or timed states stay eager
    if (st->calls++ > 0 && !st->graph_off && !st->b->trace && !st->b->timing) {
        const int grc = level1_graph_frame(st, features, output);
        if (grc == DSS_OK) return;
        st->graph_off = 1;               // eager launches from now on
        // a failed capture or instantiation enqueued nothing: this frame runs eagerly below.  A failed replay may have
        // advanced the decoder: silence for this one frame, no second pass over the same features.
        if (grc == DSS_EINTERNAL_REPLAY) { synth_failed(output, N); return; }
    }
    if (dss_lpcnet_batch_synthesize(st->b, features, 1, 1, DSS_NB_FEATURES, output)) synth_failed(output, N);
}

extern "C" int lpcnet_get_size(void) { return (int)sizeof(LPCNetState); }

// ---- encoder half of the bound ABI (cLPCNet.pxd:15-19; LPCNet.pyx:43-87) -----------------------------------------
// The feature ENCODER (pitch search, Bark cepstrum of a PCM frame) is corpus preparation (prepare_corpus.py:72-73),
// outside the accelerated path.  The symbols exist so that the reference's own LPCNet.pyx links against this library
// unchanged; lpcnet_encoder_create() returns NULL, which the reference's wrapper turns into MemoryError
// (LPCNet.pyx:53-56), so a caller finds out at construction time, not from wrong features.
struct LPCNetEncState;
extern "C" LPCNetEncState *lpcnet_encoder_create(void)
{
    dss_set_error("LPCNet feature encoder is not provided by libdss_hip (corpus preparation is outside the accelerated path)");
    return nullptr;
}
extern "C" int lpcnet_encoder_init(LPCNetEncState *) { return -1; }
extern "C" void lpcnet_encoder_destroy(LPCNetEncState *) {}
extern "C" int lpcnet_compute_features(LPCNetEncState *, const short *, float (*features)[36])
{
    if (features) memset(features, 0, sizeof(float) * 4 * 36);
    dss_set_error("lpcnet_compute_features: encoder not provided by libdss_hip");
    return -1;
}
extern "C" int lpcnet_compute_single_frame_features(LPCNetEncState *, const short *, float *features)
{
    if (features) memset(features, 0, sizeof(float) * 36);
    dss_set_error("lpcnet_compute_single_frame_features: encoder not provided by libdss_hip");
    return -1;
}
// cLPCNet.pxd:22-23 declares decode_packet inside a stray header block; nothing calls it

// Host-only check of build_fast_layout(): walks every lane's z, r and h lists through the arrays exactly as the kernel
// indexes them (register slots, tail records, long-list columns, over-reads) and compares the blocks it would multiply, in
// order, with the row's blocks in the blob.  Needs no GPU (tests/test_cpu_layout.py).
extern "C" int dss_selftest_fast_layout(const void *blob, size_t len, int *info)
{
    if (!blob || !info || len < sizeof(dss_blob_header)) { dss_set_error("dss_selftest_fast_layout: bad arguments"); return DSS_EINVAL; }
    dss_blob_header h;
    memcpy(&h, blob, sizeof(h));
    int rc = check_header(h);
    if (rc) return rc;
    std::vector<char> copy((const char *)blob, (const char *)blob + len);
    BlobView v;
    rc = view_blob(copy, h, v);
    if (rc) return rc;
    const int NA = h.gru_a, G = NA / 8;
    FastLayout F;
    build_fast_layout(v, NA, F);
    for (int k = 0; k < 8; ++k) info[k] = 0;
    info[0] = F.fast_ok ? (F.ext ? 2 : 1) : 0;
    info[1] = F.zmax; info[2] = F.hmax; info[3] = F.hfloats * 4; info[4] = F.zr_cap;
    if (!F.fast_ok) return DSS_OK;
    std::vector<int> cnt(3 * G), start(3 * G), blk0(3 * G);
    long pos = 0, blk = 0;
    for (int g = 0; g < 3 * G; ++g) { cnt[g] = v.gru_a_idx[pos]; start[g] = (int)pos + 1; blk0[g] = (int)blk; pos += 1 + cnt[g]; blk += cnt[g]; }
    int mismatches = 0, oob = 0, tails = 0;
    const int rec_end = F.ext ? F.ext_tab : F.hfloats;                 // records may be read up to here
    const int *toff = reinterpret_cast<const int *>(F.hblk.data() + F.ext_tab);
    const unsigned char *tcol = reinterpret_cast<const unsigned char *>(toff + G * 2);
    const unsigned char *hxc = tcol + (size_t)G * 2 * DSS_ZR_TAIL;
    std::vector<int> seen_zr(NA, 0), seen_h(NA, 0);
    struct Blk { int col; float w[4]; };
    auto compare = [&](const std::vector<Blk> &got, int g, int r) {
        // drop the terms that are +-0 by construction: zero column, or an all-zero padded register slot
        std::vector<Blk> eff;
        for (const Blk &b : got) {
            if (b.col == 96) continue;
            if (b.w[0] == 0.f && b.w[1] == 0.f && b.w[2] == 0.f && b.w[3] == 0.f) continue;
            eff.push_back(b);
        }
        if ((int)eff.size() != cnt[g]) { ++mismatches; return; }
        for (int sl = 0; sl < cnt[g]; ++sl) {
            const float *wb = v.gru_a_w + (size_t)(blk0[g] + sl) * 32;
            if (eff[sl].col != v.gru_a_idx[start[g] + sl] / 4) { ++mismatches; return; }
            for (int k = 0; k < 4; ++k) if (eff[sl].w[k] != wb[k * 8 + r]) { ++mismatches; return; }
        }
    };
    for (int tid = 0; tid < NA; ++tid) {
        const int wave = tid / 64, lane = tid & 63, grp2 = tid >> 3;
        {   // z and r lists of unit_of[tid]
            const int unit = F.unit_of[tid];
            if (unit < 0 || unit >= NA) { ++mismatches; continue; }
            ++seen_zr[unit];
            const int zreg = wave < 4 ? 8 : F.zr_cap, nzr = F.wave_nzr[wave], nzt = F.ext ? F.wave_nzt[wave] : 0;
            if (nzr > zreg || (!F.ext && F.wave_nzt[wave])) ++mismatches;
            for (int gate = 0; gate < 2; ++gate) {
                std::vector<Blk> got;
                for (int sl = 0; sl < nzr; ++sl) {
                    const int s2 = gate * DSS_ZRC + sl;
                    Blk b;
                    b.col = (F.zr_col[(size_t)(s2 >> 2) * NA + tid] >> (8 * (s2 & 3))) & 0xFF;
                    for (int k = 0; k < 4; ++k) b.w[k] = F.zr_w[((size_t)s2 * 4 + k) * NA + tid];
                    got.push_back(b);
                }
                for (int sl = 0; sl < nzt; ++sl) {
                    const int off = toff[grp2 * 2 + gate] + sl * 32 + (lane & 7) * 4;
                    if (off < 0 || off + 4 > rec_end) { ++oob; continue; }
                    Blk b;
                    b.col = tcol[((size_t)grp2 * 2 + gate) * DSS_ZR_TAIL + sl];
                    for (int k = 0; k < 4; ++k) b.w[k] = F.hblk[off + k];
                    if (b.col != 96) ++tails;
                    got.push_back(b);
                }
                compare(got, gate * G + unit / 8, unit & 7);
            }
        }
        {   // h list of unit_h[tid]
            const int unit = F.unit_h[tid];
            if (unit < 0 || unit >= NA) { ++mismatches; continue; }
            ++seen_h[unit];
            const int nh = F.wave_nh[wave], hreg = F.ext ? DSS_HCX : DSS_HC;
            if (!F.ext && nh > DSS_HC) ++mismatches;
            std::vector<Blk> got;
            for (int sl = 0; sl < nh; ++sl) {
                const int off = F.grp_hoff[grp2] + sl * 32 + (lane & 7) * 4;
                if (off < 0 || off + 4 > rec_end) { ++oob; continue; }
                Blk b;
                b.col = sl < hreg ? (int)((F.h_col[(size_t)(sl >> 2) * NA + tid] >> (8 * (sl & 3))) & 0xFF)
                                  : (int)hxc[(size_t)grp2 * DSS_HX + (sl - DSS_HCX)];
                for (int k = 0; k < 4; ++k) b.w[k] = F.hblk[off + k];
                got.push_back(b);
            }
            compare(got, 2 * G + unit / 8, unit & 7);
            // the kernel fetches two chunks of two slots ahead of the one it sums: those reads stay inside the image
            if (F.grp_hoff[grp2] + (nh + 4) * 32 > F.hfloats) ++oob;
        }
    }
    for (int u = 0; u < NA; ++u) if (seen_zr[u] != 1 || seen_h[u] != 1) ++mismatches;
    if ((size_t)F.hfloats * sizeof(float) > DSS_HBLK_BYTES) ++mismatches;
    info[5] = tails / 8;             // every tail block is seen by the 8 lanes of its row group
    info[6] = mismatches; info[7] = oob;
    return DSS_OK;
}

extern "C" int dss_selftest_exp10(const float *x, const float *comp, float *out, long n)
{
    if (!x || !comp || !out || n <= 0) { dss_set_error("bad arguments"); return DSS_EINVAL; }
    int rc = dss_ensure_device();
    if (rc) return rc;
    float *dx = nullptr, *dc = nullptr, *dout = nullptr;
    rc = dev_upload<float>(x, (size_t)n, &dx) | dev_upload<float>(comp, (size_t)n, &dc) | dev_alloc<float>((size_t)n, &dout);
    if (!rc) rc = dss_launch_exp10_selftest(dx, dc, dout, n, 0);
    if (!rc && hipMemcpy(out, dout, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) rc = DSS_ENODEV;
    hipFree(dx); hipFree(dc); hipFree(dout);
    return rc;
}

extern "C" int dss_selftest_lin2ulaw(unsigned start_bits, unsigned stride, long n, unsigned char *out)
{
    if (!out || n <= 0) { dss_set_error("bad arguments"); return DSS_EINVAL; }
    int rc = dss_ensure_device();
    if (rc) return rc;
    unsigned char *dout = nullptr;
    rc = dev_alloc<unsigned char>((size_t)n, &dout);
    if (!rc) rc = dss_launch_lin2ulaw_selftest(start_bits, stride, n, dout, 0);
    if (!rc && hipMemcpy(out, dout, (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) rc = DSS_ENODEV;
    hipFree(dout);
    return rc;
}

// ------------------------------------------------------------------------------------------------------
// HGA
// ------------------------------------------------------------------------------------------------------
extern "C" int dss_hga_num_windows(int T, int sr, float window_length, float window_shift)
{
    // hga_optimized.pyx:36 -- float32 products, C floor()
    return (int)floor((T - window_length * sr) / (window_shift * sr)) + 1;
}

extern "C" int dss_hga_log_power(const double *data, int T, int C, int sr, float wl, float ws, double *out)
{
    if (!data || !out || T <= 0 || C <= 0) { dss_set_error("bad arguments"); return DSS_EINVAL; }
    int rc = dss_ensure_device();
    if (rc) return rc;
    const int W = dss_hga_num_windows(T, sr, wl, ws);
    if (W <= 0) return DSS_OK;
    double *d_in = nullptr, *d_out = nullptr;
    DSS_HIP_CHECK(hipMalloc((void **)&d_in, sizeof(double) * (size_t)T * C));
    DSS_HIP_CHECK(hipMalloc((void **)&d_out, sizeof(double) * (size_t)W * C));
    DSS_HIP_CHECK(hipMemcpy(d_in, data, sizeof(double) * (size_t)T * C, hipMemcpyHostToDevice));
    rc = dss_launch_log_power(d_in, T, C, sr, wl, ws, W, d_out, 0, 0);
    if (!rc) {
        hipError_t e = hipMemcpy(out, d_out, sizeof(double) * (size_t)W * C, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { dss_set_error("copy back failed: %s", hipGetErrorString(e)); rc = DSS_ENODEV; }
    }
    hipFree(d_in); hipFree(d_out);
    if (rc) return rc;
    for (size_t k = 0; k < (size_t)W * C; ++k) out[k] = log(out[k]);       // pyx:46, host libm (DESIGN.md "HGA log")
    return DSS_OK;
}

struct dss_hga {
    int device;
    DssHgaDev d;
    int first_frame = 1;
    // optional fused front end
    int c_raw = 0, n_grids = 0;
    int *d_src_col = nullptr, *d_grid_of = nullptr, *d_comp_cols = nullptr, *d_comp_off = nullptr;
    double *d_pre = nullptr, *d_raw = nullptr, *d_wire = nullptr;
    size_t pre_cap = 0, raw_cap = 0, wire_cap = 0;
    double *d_zi0[2] = {nullptr, nullptr};
    double *d_in = nullptr, *d_out = nullptr;
    size_t in_cap = 0, out_cap = 0;
    double *d_zs[2] = {nullptr, nullptr};                  // z-score mean / std on the device ...
    std::vector<double> zs_host[2];                        // ... and on the host (host-buffer entry points); empty = no z-score
    std::vector<double> zs_dev[2];                         // the values the resident device copies hold (dss_hga_set_zscore)
};

static int hga_grow_rows(dss_hga *h, int need_rows)
{
    if (need_rows <= h->d.cap_rows) return DSS_OK;
    const int new_cap = need_rows + h->d.frame_length;
    double *nr = nullptr;
    DSS_HIP_CHECK(hipMalloc((void **)&nr, sizeof(double) * (size_t)h->d.S * new_cap * h->d.C));
    if (h->d.rows) {
        DSS_HIP_CHECK(hipMemcpy2D(nr, sizeof(double) * (size_t)new_cap * h->d.C, h->d.rows,
                                  sizeof(double) * (size_t)h->d.cap_rows * h->d.C,
                                  sizeof(double) * (size_t)h->d.overlap * h->d.C, h->d.S, hipMemcpyDeviceToDevice));
        hipFree(h->d.rows);
    }
    h->d.rows = nr;
    h->d.cap_rows = new_cap;
    return DSS_OK;
}

extern "C" dss_hga *dss_hga_create(int n_streams, int n_channels, int fs, float window_length, float window_shift,
                                   int n_sections, const double *sos_hg, const double *sos_fh, const double *zi_hg,
                                   const double *zi_fh)
{
    if (n_streams <= 0 || n_channels <= 0 || n_sections <= 0 || n_sections > 8 || !sos_hg || !sos_fh || !zi_hg || !zi_fh) {
        dss_set_error("bad HGA arguments (1..8 second-order sections supported)");
        return nullptr;
    }
    if (dss_ensure_device()) return nullptr;
    dss_hga *h = new dss_hga;
    h->device = g_device;
    DssHgaDev &d = h->d;
    memset(&d, 0, sizeof(d));
    d.S = n_streams; d.C = n_channels; d.fs = fs; d.nsec = n_sections; d.wl = window_length; d.ws = window_shift;
    // hga_optimized.pyx:72-74 (float32 products truncated to int)
    const int shift = (int)(window_shift * fs);
    d.frame_length = (int)(window_length * fs);
    d.overlap = d.frame_length - shift;
    for (int q = 0; q < n_sections; ++q)
        for (int k = 0; k < 6; ++k) { d.sos[0][q][k] = sos_hg[q * 6 + k]; d.sos[1][q][k] = sos_fh[q * 6 + k]; }
    int rc = dev_alloc<double>((size_t)n_streams * 2 * 8 * 2 * n_channels, &d.zi);
    rc |= dev_upload<double>(zi_hg, (size_t)n_sections * 2, &h->d_zi0[0]);
    rc |= dev_upload<double>(zi_fh, (size_t)n_sections * 2, &h->d_zi0[1]);
    if (!rc) rc = hga_grow_rows(h, d.overlap + 4 * d.frame_length);
    if (!rc) rc = dss_launch_hga_reset(d, h->d_zi0[0], h->d_zi0[1], 0);
    if (!rc && hipDeviceSynchronize() != hipSuccess) rc = DSS_ENODEV;
    if (rc) { dss_set_error("HGA device setup failed"); dss_hga_destroy(h); return nullptr; }
    return h;
}

extern "C" void dss_hga_destroy(dss_hga *h)
{
    if (!h) return;
    hipSetDevice(h->device);
    void *ptrs[] = {h->d.zi, h->d.rows, h->d_zi0[0], h->d_zi0[1], h->d_in, h->d_out, h->d_src_col, h->d_grid_of,
                    h->d_comp_cols, h->d_comp_off, h->d_pre, h->d_raw, h->d_wire, h->d_zs[0], h->d_zs[1]};
    for (void *p : ptrs) if (p) hipFree(p);
    delete h;
}

extern "C" int dss_hga_reset(dss_hga *h)
{
    if (!h) return DSS_EINVAL;
    DSS_HIP_CHECK(hipSetDevice(h->device));
    int rc = dss_launch_hga_reset(h->d, h->d_zi0[0], h->d_zi0[1], 0);
    if (rc) return rc;
    DSS_HIP_CHECK(hipStreamSynchronize(0));
    h->first_frame = 1;
    return DSS_OK;
}

// rows the frame buffer hands to the window stage for n new samples, and where the new rows start
static void hga_plan(const dss_hga *h, int n, int *row0, int *zero_rows, int *rows)
{
    const int fl = h->d.frame_length, ov = h->d.overlap;
    if (h->first_frame && n >= fl) { *row0 = 0; *zero_rows = 0; *rows = n; }                      // CASE 1, pyx:104-107
    else if (h->first_frame) { *row0 = fl - n; *zero_rows = fl - n; *rows = fl; }                  // CASE 2, pyx:111-122
    else { *row0 = ov; *zero_rows = 0; *rows = ov + n; }                                           // CASE 3, pyx:123-131
}

extern "C" int dss_hga_frames_for(const dss_hga *h, int n)
{
    if (!h || n <= 0) return 0;
    int row0, zr, rows;
    hga_plan(h, n, &row0, &zr, &rows);
    int W = dss_hga_num_windows(rows, h->d.fs, h->d.wl, h->d.ws);
    return W < 0 ? 0 : W;
}

// one call of the extractor on device-resident input: (S, n, C) rows, or with `fe` the raw amplifier rows
static int hga_run(dss_hga *h, const double *d_data, const DssHgaFrontDev *fe, int n, double *d_out, int apply_log, hipStream_t s)
{
    int row0, zr, rows;
    hga_plan(h, n, &row0, &zr, &rows);
    int rc = hga_grow_rows(h, rows);
    if (rc) return rc;
    int W = dss_hga_num_windows(rows, h->d.fs, h->d.wl, h->d.ws);
    if (W < 0) W = 0;
    rc = dss_launch_hga(h->d, d_data, fe, n, row0, zr, rows, W, d_out, apply_log, s);
    if (rc) return rc;
    h->first_frame = 0;
    return W;
}

extern "C" int dss_hga_extract_dev(dss_hga *h, const double *d_data, int n, double *d_out, int apply_log, void *hip_stream)
{
    if (!h || !d_data || !d_out || n <= 0) { dss_set_error("bad arguments"); return DSS_EINVAL; }
    DSS_HIP_CHECK(hipSetDevice(h->device));
    return hga_run(h, d_data, nullptr, n, d_out, apply_log, (hipStream_t)hip_stream);
}

/* Optional z-score of the frames, (x - mean[c]) / std[c] (ZScoreNormalization, local/common.py:367-376; the last step of
 * the reference's feature chain, decode_online.py:88-97), inside the extractor's launch.  NULL clears it.  The
 * host-buffer entry points apply it on the host after their host-libm log (same two IEEE operations). */
extern "C" int dss_hga_set_zscore(dss_hga *h, const double *means, const double *stds)
{
    if (!h || (!means) != (!stds)) { dss_set_error("z-score needs both means and stds (or neither)"); return DSS_EINVAL; }
    DSS_HIP_CHECK(hipSetDevice(h->device));
    // The device copies stay resident: clearing only drops the pointers the kernels see, and setting the values that are
    // already there only restores them -- a caller that toggles the epilogue per call (SegmentPipeline's intermediates
    // tap) pays no hipFree / hipMalloc / copy, i.e. no device-wide synchronisation, in its hot path.
    h->d.zs_mean = h->d.zs_std = nullptr;
    h->zs_host[0].clear(); h->zs_host[1].clear();
    if (!means) return DSS_OK;
    const size_t C = (size_t)h->d.C;
    const bool same = h->d_zs[0] && h->d_zs[1] && h->zs_dev[0].size() == C && !memcmp(h->zs_dev[0].data(), means, C * sizeof(double)) &&
                      !memcmp(h->zs_dev[1].data(), stds, C * sizeof(double));
    if (!same) {
        for (int k = 0; k < 2; ++k) { if (h->d_zs[k]) hipFree(h->d_zs[k]); h->d_zs[k] = nullptr; h->zs_dev[k].clear(); }
        if (dev_upload<double>(means, C, &h->d_zs[0]) || dev_upload<double>(stds, C, &h->d_zs[1])) return DSS_ENOMEM;
        h->zs_dev[0].assign(means, means + C);
        h->zs_dev[1].assign(stds, stds + C);
    }
    h->zs_host[0] = h->zs_dev[0];
    h->zs_host[1] = h->zs_dev[1];
    h->d.zs_mean = h->d_zs[0]; h->d.zs_std = h->d_zs[1];
    return DSS_OK;
}

/* Tests and A/B timing only: 0 = choose (default: hga_fused_kernel, three launches when its ring does not fit),
 * 1 = hga_fused_kernel, 2 = the three-launch form. */
extern "C" int dss_selftest_hga_force_path(dss_hga *h, int path)
{
    if (!h || path < 0 || path > 2) return DSS_EINVAL;
    h->d.force_path = path;
    return DSS_OK;
}

// host-side finish of the host-buffer entry points: glibc log (pyx:46; DESIGN.md "HGA log"), then the optional z-score
static void hga_host_finish(const dss_hga *h, double *out, size_t cnt)
{
    for (size_t k = 0; k < cnt; ++k) out[k] = log(out[k]);
    if (!h->zs_host[0].empty()) {
        const int C = h->d.C;
        for (size_t k = 0; k < cnt; ++k) out[k] = (out[k] - h->zs_host[0][k % C]) / h->zs_host[1][k % C];
    }
}

// the host-buffer entry points take the mean power from the device WITHOUT log and z-score (both are applied on the host)
struct HgaNoZs {
    dss_hga *h; const double *m, *sd;
    explicit HgaNoZs(dss_hga *hh) : h(hh), m(hh->d.zs_mean), sd(hh->d.zs_std) { h->d.zs_mean = h->d.zs_std = nullptr; }
    ~HgaNoZs() { h->d.zs_mean = m; h->d.zs_std = sd; }
};

extern "C" int dss_hga_extract(dss_hga *h, const double *data, int n, double *out)
{
    if (!h || !data || !out || n <= 0) { dss_set_error("bad arguments"); return DSS_EINVAL; }
    DSS_HIP_CHECK(hipSetDevice(h->device));
    const size_t in_n = (size_t)h->d.S * n * h->d.C;
    const int Wmax = dss_hga_frames_for(h, n);
    const size_t out_n = (size_t)h->d.S * (Wmax > 0 ? Wmax : 1) * h->d.C;
    if (in_n > h->in_cap) { if (h->d_in) hipFree(h->d_in); DSS_HIP_CHECK(hipMalloc((void **)&h->d_in, in_n * sizeof(double))); h->in_cap = in_n; }
    if (out_n > h->out_cap) { if (h->d_out) hipFree(h->d_out); DSS_HIP_CHECK(hipMalloc((void **)&h->d_out, out_n * sizeof(double))); h->out_cap = out_n; }
    DSS_HIP_CHECK(hipMemcpy(h->d_in, data, in_n * sizeof(double), hipMemcpyHostToDevice));
    int W;
    { HgaNoZs guard(h); W = dss_hga_extract_dev(h, h->d_in, n, h->d_out, 0, nullptr); }
    if (W < 0) return W;
    if (W == 0) { DSS_HIP_CHECK(hipDeviceSynchronize()); return 0; }
    const size_t cnt = (size_t)h->d.S * W * h->d.C;
    DSS_HIP_CHECK(hipMemcpy(out, h->d_out, cnt * sizeof(double), hipMemcpyDeviceToHost));
    hga_host_finish(h, out, cnt);
    return W;
}


extern "C" int dss_hga_set_frontend(dss_hga *h, int c_raw, const int *src_col, const int *grid_of, int n_grids,
                                    const int *comp_cols, const int *comp_off)
{
    if (!h || c_raw <= 0 || !src_col || !grid_of || n_grids < 0 || n_grids > 4 || (n_grids && (!comp_cols || !comp_off))) {
        dss_set_error("bad front-end description (at most 4 grids)");
        return DSS_EINVAL;
    }
    const int C = h->d.C;
    for (int c = 0; c < C; ++c)
        if (src_col[c] < 0 || src_col[c] >= c_raw || grid_of[c] >= n_grids) { dss_set_error("front-end column %d out of range", c); return DSS_EINVAL; }
    const int n_comp = n_grids ? comp_off[n_grids] : 0;
    if (n_comp > 4 * c_raw) { dss_set_error("front end: %d reference columns for %d raw columns", n_comp, c_raw); return DSS_EINVAL; }
    for (int g = 0; g < n_grids; ++g)
        if (comp_off[g + 1] <= comp_off[g]) { dss_set_error("grid %d has no reference channels", g); return DSS_EINVAL; }
    for (int k = 0; k < n_comp; ++k)
        if (comp_cols[k] < 0 || comp_cols[k] >= c_raw) { dss_set_error("reference column out of range"); return DSS_EINVAL; }
    DSS_HIP_CHECK(hipSetDevice(h->device));
    int rc = dev_upload<int>(src_col, C, &h->d_src_col);
    rc |= dev_upload<int>(grid_of, C, &h->d_grid_of);
    static const int zero2[2] = {0, 0};
    rc |= dev_upload<int>(n_comp ? comp_cols : zero2, n_comp ? n_comp : 1, &h->d_comp_cols);
    rc |= dev_upload<int>(n_grids ? comp_off : zero2, n_grids + 1, &h->d_comp_off);
    if (rc) return DSS_ENOMEM;
    h->c_raw = c_raw; h->n_grids = n_grids;
    return DSS_OK;
}

extern "C" int dss_hga_extract_raw_dev(dss_hga *h, const double *d_raw, int n, double *d_out, int apply_log, void *hip_stream)
{
    if (!h || !h->c_raw) { dss_set_error("no front end configured (dss_hga_set_frontend)"); return DSS_EINVAL; }
    if (!d_raw || !d_out || n <= 0) { dss_set_error("bad arguments"); return DSS_EINVAL; }
    DSS_HIP_CHECK(hipSetDevice(h->device));
    // two launches: the front end (HBM-bound), then the extractor (a one-launch form measured slower, profiles/r3_hga_experiment.md)
    const size_t need = (size_t)h->d.S * n * h->d.C;
    if (need > h->pre_cap) {
        if (h->d_pre) hipFree(h->d_pre);
        DSS_HIP_CHECK(hipMalloc((void **)&h->d_pre, need * sizeof(double)));
        h->pre_cap = need;
    }
    int rc = dss_launch_hga_frontend(d_raw, h->d_pre, h->d.S, n, h->c_raw, h->d.C, h->d_src_col, h->d_grid_of, h->n_grids,
                                     h->d_comp_cols, h->d_comp_off, (hipStream_t)hip_stream);
    if (rc) return rc;
    return dss_hga_extract_dev(h, h->d_pre, n, d_out, apply_log, hip_stream);
}

// Payloads in wire format (float32, [stream][channel][sample]: the body of the amplifier's packets) -> frames.  With a front end
// configured the payload carries its c_raw channels and goes through it, otherwise the extractor's own n_channels.
extern "C" int dss_hga_extract_wire_dev(dss_hga *h, const float *d_payload, int n, double *d_out, int apply_log, void *hip_stream)
{
    if (!h || !d_payload || !d_out || n <= 0) { dss_set_error("bad arguments"); return DSS_EINVAL; }
    DSS_HIP_CHECK(hipSetDevice(h->device));
    const int c_in = h->c_raw ? h->c_raw : h->d.C;
    const size_t need = (size_t)h->d.S * n * c_in;
    if (need > h->wire_cap) {
        if (h->d_wire) hipFree(h->d_wire);
        h->d_wire = nullptr; h->wire_cap = 0;
        DSS_HIP_CHECK(hipMalloc((void **)&h->d_wire, need * sizeof(double)));
        h->wire_cap = need;
    }
    int rc = dss_launch_hga_wire(d_payload, h->d_wire, h->d.S, c_in, n, (hipStream_t)hip_stream);
    if (rc) return rc;
    return h->c_raw ? dss_hga_extract_raw_dev(h, h->d_wire, n, d_out, apply_log, hip_stream)
                    : dss_hga_extract_dev(h, h->d_wire, n, d_out, apply_log, hip_stream);
}

extern "C" int dss_hga_extract_raw(dss_hga *h, const double *raw, int n, double *out)
{
    if (!h || !h->c_raw) { dss_set_error("no front end configured (dss_hga_set_frontend)"); return DSS_EINVAL; }
    if (!raw || !out || n <= 0) { dss_set_error("bad arguments"); return DSS_EINVAL; }
    DSS_HIP_CHECK(hipSetDevice(h->device));
    const size_t in_n = (size_t)h->d.S * n * h->c_raw;
    if (in_n > h->raw_cap) {
        if (h->d_raw) hipFree(h->d_raw);
        DSS_HIP_CHECK(hipMalloc((void **)&h->d_raw, in_n * sizeof(double)));
        h->raw_cap = in_n;
    }
    const int Wmax = dss_hga_frames_for(h, n);
    const size_t out_n = (size_t)h->d.S * (Wmax > 0 ? Wmax : 1) * h->d.C;
    if (out_n > h->out_cap) { if (h->d_out) hipFree(h->d_out); DSS_HIP_CHECK(hipMalloc((void **)&h->d_out, out_n * sizeof(double))); h->out_cap = out_n; }
    DSS_HIP_CHECK(hipMemcpy(h->d_raw, raw, in_n * sizeof(double), hipMemcpyHostToDevice));
    int W;
    { HgaNoZs guard(h); W = dss_hga_extract_raw_dev(h, h->d_raw, n, h->d_out, 0, nullptr); }
    if (W < 0) return W;
    if (W == 0) { DSS_HIP_CHECK(hipDeviceSynchronize()); return 0; }
    const size_t cnt = (size_t)h->d.S * W * h->d.C;
    DSS_HIP_CHECK(hipMemcpy(out, h->d_out, cnt * sizeof(double), hipMemcpyDeviceToHost));
    hga_host_finish(h, out, cnt);
    return W;
}

// ------------------------------------------------------------------------------------------------------
// speech-segment gate (Part 4 of include/dss_hip.h)
// ------------------------------------------------------------------------------------------------------
struct dss_gate {
    int device;
    int max_frames;
    DssGateDev d;
    double *d_frames = nullptr;   // staging of the host-buffer entry point
    int *d_labels = nullptr;
    std::vector<int> last_events; // host copy of the last push's event records
};

extern "C" dss_gate *dss_gate_create(int n_streams, int nb_features, int smoothing_context, double proportion_threshold,
                                     int buffer_size, int context, int max_frames)
{
    if (n_streams <= 0 || nb_features <= 0 || smoothing_context < 0 || 2 * smoothing_context + 1 > 64 || buffer_size <= 0 ||
        context < 0 || max_frames <= 0) {
        dss_set_error("bad gate arguments (smoothing window 2*ctx+1 must be <= 64)");
        return nullptr;
    }
    if (dss_ensure_device()) return nullptr;
    dss_gate *g = new dss_gate;
    g->device = g_device;
    g->max_frames = max_frames;
    DssGateDev &d = g->d;
    memset(&d, 0, sizeof(d));
    d.S = n_streams; d.C = nb_features; d.sm_ctx = smoothing_context; d.sm_size = 2 * smoothing_context + 1;
    d.hist_size = buffer_size; d.hist_ctx = context; d.threshold = proportion_threshold;
    // a segment closes on the context-th non-speech frame after >= 1 speech frame: at most one per (context + 1)
    // frames, or one per 2 frames without context
    d.max_events = context > 0 ? max_frames / (context + 1) + 1 : (max_frames + 1) / 2;
    const size_t S = n_streams, C = nb_features;
    int rc = dev_alloc<float>(S * d.sm_size * C, &d.sm_buf);
    rc |= dev_alloc<float>(S * d.hist_size * C, &d.hist);
    rc |= dev_alloc<float>(S * d.max_events * d.hist_size * C, &d.seg_out);
    rc |= dev_alloc<int>(S * DSS_GATE_STATE_INTS, &d.state);
    rc |= dev_alloc<int>(S * (2 + d.max_events), &d.events);
    rc |= dev_alloc<double>(S * max_frames * C, &g->d_frames);
    rc |= dev_alloc<int>(S * max_frames, &g->d_labels);
    if (!rc) rc = dss_launch_gate_reset(d, -1, 0);
    if (!rc && hipDeviceSynchronize() != hipSuccess) rc = DSS_ENODEV;
    if (rc) { dss_set_error("gate device setup failed"); dss_gate_destroy(g); return nullptr; }
    g->last_events.assign(S * (2 + d.max_events), 0);
    return g;
}

extern "C" void dss_gate_destroy(dss_gate *g)
{
    if (!g) return;
    hipSetDevice(g->device);
    void *ptrs[] = {g->d.sm_buf, g->d.hist, g->d.seg_out, g->d.state, g->d.events, g->d_frames, g->d_labels};
    for (void *p : ptrs) if (p) hipFree(p);
    delete g;
}

extern "C" int dss_gate_max_events(const dss_gate *g) { return g ? g->d.max_events : DSS_EINVAL; }

extern "C" int dss_gate_reset(dss_gate *g, int stream)
{
    if (!g || stream >= g->d.S) { dss_set_error("bad gate/stream"); return DSS_EINVAL; }
    DSS_HIP_CHECK(hipSetDevice(g->device));
    int rc = dss_launch_gate_reset(g->d, stream, 0);
    if (rc) return rc;
    DSS_HIP_CHECK(hipStreamSynchronize(0));
    return DSS_OK;
}

extern "C" int dss_gate_push_dev(dss_gate *g, const double *d_frames, const int *d_labels, int n_frames, int *events,
                                 void *hip_stream)
{
    if (!g || !d_frames || !d_labels || !events) { dss_set_error("null argument"); return DSS_EINVAL; }
    if (n_frames <= 0 || n_frames > g->max_frames) { dss_set_error("%d frames per push outside [1, %d]", n_frames, g->max_frames); return DSS_EINVAL; }
    DSS_HIP_CHECK(hipSetDevice(g->device));
    hipStream_t s = (hipStream_t)hip_stream;
    int rc = dss_launch_gate(g->d, d_frames, d_labels, n_frames, s);
    if (rc) return rc;
    const size_t n = (size_t)g->d.S * (2 + g->d.max_events);
    DSS_HIP_CHECK(hipMemcpyAsync(g->last_events.data(), g->d.events, n * sizeof(int), hipMemcpyDeviceToHost, s));
    DSS_HIP_CHECK(hipStreamSynchronize(s));
    int total = 0;
    for (int st = 0; st < g->d.S; ++st) {
        const int ne = g->last_events[(size_t)st * (2 + g->d.max_events)];
        if (ne > g->d.max_events) { dss_set_error("stream %d completed %d segments in one push (capacity %d)", st, ne, g->d.max_events); return DSS_EINVAL; }
        total += ne;
    }
    memcpy(events, g->last_events.data(), n * sizeof(int));
    return total;
}

extern "C" int dss_gate_push(dss_gate *g, const double *frames, const int *labels, int n_frames, int *events)
{
    if (!g || !frames || !labels || !events) { dss_set_error("null argument"); return DSS_EINVAL; }
    if (n_frames <= 0 || n_frames > g->max_frames) { dss_set_error("%d frames per push outside [1, %d]", n_frames, g->max_frames); return DSS_EINVAL; }
    DSS_HIP_CHECK(hipSetDevice(g->device));
    DSS_HIP_CHECK(hipMemcpy(g->d_frames, frames, sizeof(double) * g->d.S * n_frames * g->d.C, hipMemcpyHostToDevice));
    DSS_HIP_CHECK(hipMemcpy(g->d_labels, labels, sizeof(int) * g->d.S * n_frames, hipMemcpyHostToDevice));
    return dss_gate_push_dev(g, g->d_frames, g->d_labels, n_frames, events, nullptr);
}

static int gate_segment_src(dss_gate *g, int stream, int event, int cap_frames, const float **src, int *len)
{
    if (!g || stream < 0 || stream >= g->d.S || event < 0) { dss_set_error("bad gate/stream/event"); return DSS_EINVAL; }
    const int *ev = &g->last_events[(size_t)stream * (2 + g->d.max_events)];
    if (event >= ev[0]) { dss_set_error("stream %d completed %d segments in the last push, asked for #%d", stream, ev[0], event); return DSS_EINVAL; }
    *len = ev[2 + event];
    if (*len > cap_frames) { dss_set_error("segment has %d frames, buffer holds %d", *len, cap_frames); return DSS_EINVAL; }
    *src = g->d.seg_out + ((size_t)stream * g->d.max_events + event) * (size_t)g->d.hist_size * g->d.C;
    return DSS_OK;
}

extern "C" int dss_gate_segment_dev(dss_gate *g, int stream, int event, float *d_dst, int cap_frames, void *hip_stream)
{
    const float *src; int len;
    int rc = gate_segment_src(g, stream, event, cap_frames, &src, &len);
    if (rc) return rc;
    if (!d_dst) { dss_set_error("null argument"); return DSS_EINVAL; }
    DSS_HIP_CHECK(hipSetDevice(g->device));
    if (len) DSS_HIP_CHECK(hipMemcpyAsync(d_dst, src, sizeof(float) * (size_t)len * g->d.C, hipMemcpyDeviceToDevice, (hipStream_t)hip_stream));
    return len;
}

extern "C" int dss_gate_collect_dev(dss_gate *g, int n, const int *streams, const int *events, const int *dst_rows, float *d_dst,
                                    int row_frames, void *hip_stream)
{
    if (!g || n < 0 || (n && (!streams || !events || !dst_rows || !d_dst))) { dss_set_error("dss_gate_collect_dev: bad arguments"); return DSS_EINVAL; }
    for (int i = 0; i < n; ++i) {
        const float *src; int len;
        int rc = gate_segment_src(g, streams[i], events[i], row_frames, &src, &len);
        if (rc) return rc;
        if (dst_rows[i] < 0) { dss_set_error("segment %d: negative destination row", i); return DSS_EINVAL; }
    }
    DSS_HIP_CHECK(hipSetDevice(g->device));
    for (int i0 = 0; i0 < n; i0 += DSS_GATE_COLLECT_MAX) {
        const int m = std::min(DSS_GATE_COLLECT_MAX, n - i0);
        DssGateCollect a;
        memset(&a, 0, sizeof(a));
        for (int i = 0; i < m; ++i) { a.stream[i] = streams[i0 + i]; a.event[i] = events[i0 + i]; a.dst_row[i] = dst_rows[i0 + i]; }
        int rc = dss_launch_gate_collect(g->d, a, m, d_dst, (long)row_frames * g->d.C, (hipStream_t)hip_stream);
        if (rc) return rc;
    }
    return n;
}

extern "C" int dss_gate_segment(dss_gate *g, int stream, int event, float *dst, int cap_frames)
{
    const float *src; int len;
    int rc = gate_segment_src(g, stream, event, cap_frames, &src, &len);
    if (rc) return rc;
    if (!dst) { dss_set_error("null argument"); return DSS_EINVAL; }
    DSS_HIP_CHECK(hipSetDevice(g->device));
    if (len) DSS_HIP_CHECK(hipMemcpy(dst, src, sizeof(float) * (size_t)len * g->d.C, hipMemcpyDeviceToHost));
    return len;
}

extern "C" int dss_gate_frames_seen(dss_gate *g, int stream)
{
    if (!g || stream < 0 || stream >= g->d.S) { dss_set_error("bad gate/stream"); return DSS_EINVAL; }
    if (hipSetDevice(g->device) != hipSuccess) return DSS_ENODEV;
    int v = 0;
    if (hipMemcpy(&v, g->d.state + (size_t)stream * DSS_GATE_STATE_INTS + 7, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return DSS_ENODEV;
    return v;
}

// ------------------------------------------------------------------------------------------------------
// neural voice-activity detector (Part 5 of include/dss_hip.h; csrc/vad_lstm.hip)
// ------------------------------------------------------------------------------------------------------
struct dss_vad {
    int device;
    DssVadDev d;
    float *w[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // wT0, b0, wT1, b1, wc, bc
    bool loaded = false;
};

extern "C" dss_vad *dss_vad_create(int n_streams, int n_inputs, int hidden_units)
{
    if (n_streams <= 0 || n_inputs <= 0 || hidden_units <= 0) { dss_set_error("VAD dims must be positive"); return nullptr; }
    if (hidden_units > DSS_VAD_MAXH || n_inputs > DSS_VAD_MAXC) {
        dss_set_error("VAD kernel: %d hidden units / %d inputs out of range (<= %d / <= %d)", hidden_units, n_inputs, DSS_VAD_MAXH, DSS_VAD_MAXC);
        return nullptr;
    }
    if (dss_ensure_device()) return nullptr;
    dss_vad *v = new dss_vad;
    memset(&v->d, 0, sizeof(v->d));
    v->device = g_device;
    v->d.S = n_streams; v->d.C = n_inputs; v->d.H = hidden_units;
    const size_t n = (size_t)2 * n_streams * hidden_units;
    if (dev_alloc<float>(n, &v->d.h) || dev_alloc<float>(n, &v->d.c) || hipMemset(v->d.h, 0, n * sizeof(float)) != hipSuccess ||
        hipMemset(v->d.c, 0, n * sizeof(float)) != hipSuccess) {
        dss_set_error("device allocation failed for the VAD state");
        dss_vad_destroy(v);
        return nullptr;
    }
    return v;
}

extern "C" void dss_vad_destroy(dss_vad *v)
{
    if (!v) return;
    hipSetDevice(v->device);
    for (float *p : v->w) if (p) hipFree(p);
    if (v->d.h) hipFree(v->d.h);
    if (v->d.c) hipFree(v->d.c);
    delete v;
}

// torch.nn.LSTM parameter layout (host arrays): weight_ih_l0 [4H][C], weight_hh_l0 [4H][H], bias_ih_l0 / bias_hh_l0 [4H],
// weight_ih_l1 [4H][H], weight_hh_l1 [4H][H], bias_ih_l1 / bias_hh_l1 [4H], classifier weight [2][H] and bias [2]
extern "C" int dss_vad_load_weights(dss_vad *v, const float *w_ih0, const float *w_hh0, const float *b_ih0, const float *b_hh0,
                                    const float *w_ih1, const float *w_hh1, const float *b_ih1, const float *b_hh1,
                                    const float *cls_w, const float *cls_b)
{
    if (!v || !w_ih0 || !w_hh0 || !b_ih0 || !b_hh0 || !w_ih1 || !w_hh1 || !b_ih1 || !b_hh1 || !cls_w || !cls_b) {
        dss_set_error("dss_vad_load_weights: null argument"); return DSS_EINVAL;
    }
    DSS_HIP_CHECK(hipSetDevice(v->device));
    const int C = v->d.C, H = v->d.H, H4 = 4 * H, Cp = (C + 3) & ~3, Hp = (H + 3) & ~3;
    // the kernel's copies: [inputs / 4][4H rows][4 consecutive inputs], input counts padded to multiples of 4 with zero weights
    std::vector<float> t0((size_t)(Cp + Hp) * H4, 0.f), t1((size_t)2 * Hp * H4, 0.f), b0(H4), b1(H4);
    auto put = [&](std::vector<float> &t, int k, int r, float w) { t[((size_t)(k >> 2) * H4 + r) * 4 + (k & 3)] = w; };
    for (int r = 0; r < H4; ++r) {
        for (int k = 0; k < C; ++k) put(t0, k, r, w_ih0[(size_t)r * C + k]);
        for (int k = 0; k < H; ++k) put(t0, Cp + k, r, w_hh0[(size_t)r * H + k]);
        for (int k = 0; k < H; ++k) put(t1, k, r, w_ih1[(size_t)r * H + k]);
        for (int k = 0; k < H; ++k) put(t1, Hp + k, r, w_hh1[(size_t)r * H + k]);
        b0[r] = b_ih0[r] + b_hh0[r];
        b1[r] = b_ih1[r] + b_hh1[r];
    }
    // upload beside the weights in use and switch only when every array has arrived: a failed load leaves the detector
    // as it was (an earlier model keeps running; without one, `loaded` stays false)
    float *nw[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int rc = dev_upload<float>(t0.data(), t0.size(), &nw[0]);
    rc |= dev_upload<float>(b0.data(), b0.size(), &nw[1]);
    rc |= dev_upload<float>(t1.data(), t1.size(), &nw[2]);
    rc |= dev_upload<float>(b1.data(), b1.size(), &nw[3]);
    rc |= dev_upload<float>(cls_w, (size_t)2 * H, &nw[4]);
    rc |= dev_upload<float>(cls_b, 2, &nw[5]);
    if (rc) { for (float *p : nw) if (p) hipFree(p); return DSS_ENOMEM; }
    DSS_HIP_CHECK(hipDeviceSynchronize());                    // no launch may still read the arrays about to be freed
    for (int k = 0; k < 6; ++k) { if (v->w[k]) hipFree(v->w[k]); v->w[k] = nw[k]; }
    v->d.wT0 = v->w[0]; v->d.b0 = v->w[1]; v->d.wT1 = v->w[2]; v->d.b1 = v->w[3]; v->d.wc = v->w[4]; v->d.bc = v->w[5];
    v->loaded = true;
    return DSS_OK;
}

// zero the recurrent state of one stream (create_new_initial_state, models.py:22-24), or of all (stream < 0)
extern "C" int dss_vad_reset(dss_vad *v, int stream)
{
    if (!v || stream >= v->d.S) { dss_set_error("bad VAD / stream"); return DSS_EINVAL; }
    DSS_HIP_CHECK(hipSetDevice(v->device));
    const size_t SH = (size_t)v->d.S * v->d.H, H = v->d.H;
    for (float *p : {v->d.h, v->d.c}) {
        if (stream < 0) { DSS_HIP_CHECK(hipMemset(p, 0, 2 * SH * sizeof(float))); continue; }
        for (int layer = 0; layer < 2; ++layer) DSS_HIP_CHECK(hipMemset(p + layer * SH + (size_t)stream * H, 0, H * sizeof(float)));
    }
    return DSS_OK;
}

// the same, enqueued on the stream the steps run on (dss_vad_reset uses the null stream and waits: it is ordered against
// steps on a blocking stream only)
extern "C" int dss_vad_reset_async(dss_vad *v, int stream, void *hip_stream)
{
    if (!v || stream >= v->d.S) { dss_set_error("bad VAD / stream"); return DSS_EINVAL; }
    DSS_HIP_CHECK(hipSetDevice(v->device));
    hipStream_t st = (hipStream_t)hip_stream;
    const size_t SH = (size_t)v->d.S * v->d.H, H = v->d.H;
    for (float *p : {v->d.h, v->d.c}) {
        if (stream < 0) { DSS_HIP_CHECK(hipMemsetAsync(p, 0, 2 * SH * sizeof(float), st)); continue; }
        for (int layer = 0; layer < 2; ++layer) DSS_HIP_CHECK(hipMemsetAsync(p + layer * SH + (size_t)stream * H, 0, H * sizeof(float), st));
    }
    return DSS_OK;
}

// d_frames: (S, n_frames, C) float64 (frames_are_f64, as the extractor returns them) or float32; d_labels: (S, n_frames) int32;
// d_logits: (S, n_frames, 2) float32 or NULL.  All device pointers; asynchronous on hip_stream.
extern "C" int dss_vad_step_dev(dss_vad *v, const void *d_frames, int frames_are_f64, int n_frames, int *d_labels, float *d_logits,
                                void *hip_stream)
{
    if (!v || !d_frames || !d_labels || n_frames <= 0) { dss_set_error("dss_vad_step_dev: bad arguments"); return DSS_EINVAL; }
    if (!v->loaded) { dss_set_error("dss_vad_step_dev: no weights loaded (dss_vad_load_weights)"); return DSS_EINVAL; }
    DSS_HIP_CHECK(hipSetDevice(v->device));
    return dss_launch_vad(v->d, d_frames, frames_are_f64, n_frames, d_labels, d_logits, (hipStream_t)hip_stream);
}

// host copies of the recurrent state, [2 layers][S][H] each (either may be NULL); set == 0 reads, set != 0 writes
extern "C" int dss_vad_state(dss_vad *v, float *h, float *c, int set)
{
    if (!v) return DSS_EINVAL;
    DSS_HIP_CHECK(hipSetDevice(v->device));
    const size_t n = (size_t)2 * v->d.S * v->d.H * sizeof(float);
    DSS_HIP_CHECK(hipDeviceSynchronize());
    if (h) DSS_HIP_CHECK(set ? hipMemcpy(v->d.h, h, n, hipMemcpyHostToDevice) : hipMemcpy(h, v->d.h, n, hipMemcpyDeviceToHost));
    if (c) DSS_HIP_CHECK(set ? hipMemcpy(v->d.c, c, n, hipMemcpyHostToDevice) : hipMemcpy(c, v->d.c, n, hipMemcpyDeviceToHost));
    return DSS_OK;
}

// ------------------------------------------------------------------------------------------------------
// bidirectional recurrent decoder (Part 6 of include/dss_hip.h; csrc/bilstm_decoder.hip)
// ------------------------------------------------------------------------------------------------------
struct dss_dec {
    int device;
    DssDecDev d;
    float *w[10] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // wT[2][2], b[2][2], wr, br
    bool loaded = false;
    int *d_meta = nullptr;        // [2][S_max]: frame counts, input rows of a ragged call (dss_dec_forward_rows_dev)
    DssPinnedRing meta;           //   their pinned staging
};

extern "C" dss_dec *dss_dec_create(int max_streams, int max_frames, int n_inputs, int hidden_units, int n_outputs)
{
    if (max_streams <= 0 || max_frames <= 0 || n_inputs <= 0 || hidden_units <= 0 || n_outputs <= 0) {
        dss_set_error("decoder dims must be positive"); return nullptr;
    }
    if (hidden_units > DSS_DEC_MAXH || n_inputs > DSS_DEC_MAXC || n_outputs > DSS_DEC_MAXO) {
        dss_set_error("decoder kernel: %d hidden units / %d inputs / %d outputs out of range (<= %d / <= %d / <= %d)", hidden_units, n_inputs,
                      n_outputs, DSS_DEC_MAXH, DSS_DEC_MAXC, DSS_DEC_MAXO);
        return nullptr;
    }
    if (dss_ensure_device()) return nullptr;
    dss_dec *v = new dss_dec;
    memset(&v->d, 0, sizeof(v->d));
    v->device = g_device;
    v->d.S_max = max_streams; v->d.T_max = max_frames; v->d.C = n_inputs; v->d.H = hidden_units; v->d.O = n_outputs;
    const size_t n = (size_t)max_streams * max_frames * 2 * hidden_units;
    if (dev_alloc<float>(n, &v->d.mid) || dev_alloc<float>(n, &v->d.top) || dev_alloc<int>((size_t)2 * max_streams, &v->d_meta) ||
        v->meta.init((size_t)2 * max_streams)) {
        dss_set_error("device allocation failed for the decoder's layer outputs");
        dss_dec_destroy(v);
        return nullptr;
    }
    return v;
}

extern "C" void dss_dec_destroy(dss_dec *v)
{
    if (!v) return;
    hipSetDevice(v->device);
    for (float *p : v->w) if (p) hipFree(p);
    if (v->d.mid) hipFree(v->d.mid);
    if (v->d.top) hipFree(v->d.top);
    if (v->d_meta) hipFree(v->d_meta);
    v->meta.destroy();
    delete v;
}

// w: 18 host arrays in torch.nn.LSTM's own layout, in state_dict order of the reference class:
//   for layer in (0, 1): for direction in (forward, reverse): weight_ih [4H][Cin], weight_hh [4H][H], bias_ih [4H], bias_hh [4H]
//   (Cin = n_inputs for layer 0, 2H for layer 1), then regressor.weight [O][2H], regressor.bias [O]
extern "C" int dss_dec_load_weights(dss_dec *v, const float *const *w)
{
    if (!v || !w) { dss_set_error("dss_dec_load_weights: null argument"); return DSS_EINVAL; }
    for (int k = 0; k < 18; ++k) if (!w[k]) { dss_set_error("dss_dec_load_weights: null array %d", k); return DSS_EINVAL; }
    DSS_HIP_CHECK(hipSetDevice(v->device));
    const int H = v->d.H, H4 = 4 * H, Hp = (H + 3) & ~3;
    // upload beside the weights in use and switch only when every array has arrived (see dss_vad_load_weights)
    float *nw[10] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int rc = 0;
    for (int layer = 0; layer < 2; ++layer) {
        const int Cin = layer ? 2 * H : v->d.C, Cp = (Cin + 3) & ~3;
        for (int dir = 0; dir < 2; ++dir) {
            const float *w_ih = w[(layer * 2 + dir) * 4 + 0], *w_hh = w[(layer * 2 + dir) * 4 + 1];
            const float *b_ih = w[(layer * 2 + dir) * 4 + 2], *b_hh = w[(layer * 2 + dir) * 4 + 3];
            // the kernel's copy: [inputs / 4][4H rows][4 consecutive inputs], input counts padded to multiples of 4 with zero weights
            std::vector<float> t((size_t)(Cp + Hp) * H4, 0.f), b(H4);
            auto put = [&](int k, int r, float x) { t[((size_t)(k >> 2) * H4 + r) * 4 + (k & 3)] = x; };
            for (int r = 0; r < H4; ++r) {
                for (int k = 0; k < Cin; ++k) put(k, r, w_ih[(size_t)r * Cin + k]);
                for (int k = 0; k < H; ++k) put(Cp + k, r, w_hh[(size_t)r * H + k]);
                b[r] = b_ih[r] + b_hh[r];
            }
            rc |= dev_upload<float>(t.data(), t.size(), &nw[layer * 2 + dir]);
            rc |= dev_upload<float>(b.data(), b.size(), &nw[4 + layer * 2 + dir]);
        }
    }
    rc |= dev_upload<float>(w[16], (size_t)v->d.O * 2 * H, &nw[8]);
    rc |= dev_upload<float>(w[17], (size_t)v->d.O, &nw[9]);
    if (rc) { for (float *p : nw) if (p) hipFree(p); return DSS_ENOMEM; }
    DSS_HIP_CHECK(hipDeviceSynchronize());
    for (int k = 0; k < 10; ++k) { if (v->w[k]) hipFree(v->w[k]); v->w[k] = nw[k]; }
    for (int layer = 0; layer < 2; ++layer)
        for (int dir = 0; dir < 2; ++dir) { v->d.wT[layer][dir] = v->w[layer * 2 + dir]; v->d.b[layer][dir] = v->w[4 + layer * 2 + dir]; }
    v->d.wr = v->w[8]; v->d.br = v->w[9];
    v->loaded = true;
    return DSS_OK;
}

// d_frames: (n_streams, n_frames, n_inputs) float64 (frames_are_f64: as the extractor returns them; cast to float32 like
// units.py:503) or float32; d_feats: (n_streams, n_frames, n_outputs) float32.  Device pointers; asynchronous on hip_stream.
// Every call starts from the zero state (units.py:499-508: a fresh state per segment).
extern "C" int dss_dec_forward_dev(dss_dec *v, const void *d_frames, int frames_are_f64, int n_streams, int n_frames, float *d_feats,
                                   void *hip_stream)
{
    if (!v || !d_frames || !d_feats) { dss_set_error("dss_dec_forward_dev: bad arguments"); return DSS_EINVAL; }
    if (!v->loaded) { dss_set_error("dss_dec_forward_dev: no weights loaded (dss_dec_load_weights)"); return DSS_EINVAL; }
    DSS_HIP_CHECK(hipSetDevice(v->device));
    return dss_launch_decoder(v->d, d_frames, frames_are_f64, n_streams, n_frames, d_feats, nullptr, nullptr, 0, (hipStream_t)hip_stream);
}

// Ragged form (segments of different lengths closing on the same tick): stream i has counts[i] <= n_frames frames, read from
// row in_rows[i] (NULL: i) of d_frames, a buffer of row_frames frames per row; its backward direction starts at its own last
// frame.  counts / in_rows are HOST arrays.  d_feats is (n_streams, n_frames, n_outputs); rows beyond counts[i] stay untouched.
extern "C" int dss_dec_forward_rows_dev(dss_dec *v, const void *d_frames, int frames_are_f64, int row_frames, const int *in_rows,
                                        const int *counts, int n_streams, int n_frames, float *d_feats, void *hip_stream)
{
    if (!v || !d_frames || !d_feats || !counts) { dss_set_error("dss_dec_forward_rows_dev: bad arguments"); return DSS_EINVAL; }
    if (!v->loaded) { dss_set_error("dss_dec_forward_rows_dev: no weights loaded (dss_dec_load_weights)"); return DSS_EINVAL; }
    if (n_streams < 1 || n_streams > v->d.S_max || n_frames < 1 || n_frames > v->d.T_max || row_frames < n_frames) {
        dss_set_error("dss_dec_forward_rows_dev: %d streams x %d frames (rows of %d) exceed the handle's %d x %d", n_streams, n_frames,
                      row_frames, v->d.S_max, v->d.T_max);
        return DSS_EINVAL;
    }
    for (int i = 0; i < n_streams; ++i) {
        if (counts[i] < 0 || counts[i] > n_frames) { dss_set_error("stream %d: %d frames outside [0, %d]", i, counts[i], n_frames); return DSS_EINVAL; }
        if (in_rows && in_rows[i] < 0) { dss_set_error("stream %d: negative input row", i); return DSS_EINVAL; }
    }
    DSS_HIP_CHECK(hipSetDevice(v->device));
    hipStream_t s = (hipStream_t)hip_stream;
    int *h = v->meta.acquire();
    if (!h) { dss_set_error("pinned staging ring failed"); return DSS_ENODEV; }
    const size_t S = (size_t)v->d.S_max;
    memcpy(h, counts, sizeof(int) * n_streams);
    if (in_rows) memcpy(h + S, in_rows, sizeof(int) * n_streams);
    DSS_HIP_CHECK(hipMemcpyAsync(v->d_meta, h, sizeof(int) * n_streams, hipMemcpyHostToDevice, s));
    if (in_rows) DSS_HIP_CHECK(hipMemcpyAsync(v->d_meta + S, h + S, sizeof(int) * n_streams, hipMemcpyHostToDevice, s));
    int rc = v->meta.commit(s);
    if (rc) return rc;
    return dss_launch_decoder(v->d, d_frames, frames_are_f64, n_streams, n_frames, d_feats, v->d_meta, in_rows ? v->d_meta + S : nullptr,
                              row_frames, s);
}
