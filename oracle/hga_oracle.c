/*
 * oracle/hga_oracle.c -- CPU restatement of the reference's high-gamma (HGA) feature path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's library.
 * The product path (delayed-speech-synthesis_amd/) never links, imports or calls it.
 *
 * What is restated (all float64, no FMA contraction, same operation order as the reference):
 *   - scipy.signal.sosfilt(sos, x, axis=0, zi=...) as driven by HighGammaExtractor.extract_features
 *       reference: local/units.py:151-152 (two cascades: band-pass then band-stop, state carried)
 *       algorithm: scipy's direct-form-II-transposed biquad cascade, per sample / per section
 *           y  = b0*x + z0;  z0 = b1*x - a1*y + z1;  z1 = b2*x - a2*y
 *   - WarmStartFrameBuffer.insert      extensions/hga/hga_optimized.pyx:96-131 (three cases)
 *   - compute_log_power_features       extensions/hga/hga_optimized.pyx:27-47 (+ :9-22)
 *
 * Pinning: tests/test_oracle_hga.py checks this file bit-for-bit against tests/golden/hga_*.npz,
 * which were produced by the reference's own Cython module compiled from /root/reference
 * (oracle/Makefile target `ref`) plus scipy.signal.sosfilt, see oracle/make_golden.py.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (no -march, no fast-math), see oracle/Makefile.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- sosfilt: x is (n, C) row-major float64, filtered in place along axis 0 --------------------
 * sos: (n_sections, 6) rows [b0 b1 b2 a0 a1 a2] with a0 == 1 (scipy normalises);
 * zi : (n_sections, 2, C) as HighGammaExtractor keeps it (local/units.py:128-132). */
void oracle_sosfilt(const double *sos, int n_sections, double *x, int n, int C, double *zi)
{
    for (int c = 0; c < C; ++c) {
        for (int t = 0; t < n; ++t) {
            double v = x[(size_t)t * C + c];
            for (int s = 0; s < n_sections; ++s) {
                const double *k = sos + 6 * s;
                double *z0 = zi + ((size_t)s * 2 + 0) * C + c;
                double *z1 = zi + ((size_t)s * 2 + 1) * C + c;
                double y = k[0] * v + *z0;
                *z0 = k[1] * v - k[4] * y + *z1;
                *z1 = k[2] * v - k[5] * y;
                v = y;
            }
            x[(size_t)t * C + c] = v;
        }
    }
}

/* ---- WarmStartFrameBuffer (hga_optimized.pyx:50-131) ------------------------------------------- */
typedef struct {
    int frame_length;   /* <int>(frame_length * fs)  with float32 frame_length, pyx:73 */
    int overlap;        /* frame_length - <int>(frame_shift * fs), pyx:72-74 */
    int C;
    int first_frame;
    double *remainder;  /* (overlap, C) */
} oracle_framebuf;

oracle_framebuf *oracle_framebuf_create(float frame_length, float frame_shift, int fs, int C)
{
    oracle_framebuf *fb = (oracle_framebuf *)calloc(1, sizeof(*fb));
    int shift = (int)(frame_shift * fs);          /* float32 * int -> float32 product, truncated */
    fb->frame_length = (int)(frame_length * fs);
    fb->overlap = fb->frame_length - shift;
    fb->C = C;
    fb->first_frame = 1;
    fb->remainder = (double *)calloc((size_t)fb->overlap * C, sizeof(double));
    return fb;
}

void oracle_framebuf_destroy(oracle_framebuf *fb)
{
    if (fb) { free(fb->remainder); free(fb); }
}

void oracle_framebuf_reset(oracle_framebuf *fb)
{
    fb->first_frame = 1;
    memset(fb->remainder, 0, sizeof(double) * (size_t)fb->overlap * fb->C);
}

/* Returns the number of rows written to out (caller provides room for overlap + n + frame_length). */
int oracle_framebuf_insert(oracle_framebuf *fb, const double *data, int n, double *out)
{
    const int C = fb->C, ov = fb->overlap, fl = fb->frame_length;
    int rows;
    if (fb->first_frame && n >= fl) {                       /* CASE 1, pyx:104-107 */
        memcpy(out, data, sizeof(double) * (size_t)n * C);
        rows = n;
    } else if (fb->first_frame) {                           /* CASE 2, pyx:111-122 */
        int pre = fl - n;
        memset(out, 0, sizeof(double) * (size_t)pre * C);
        memcpy(out + (size_t)pre * C, data, sizeof(double) * (size_t)n * C);
        rows = fl;
    } else {                                                /* CASE 3, pyx:123-131 */
        memcpy(out, fb->remainder, sizeof(double) * (size_t)ov * C);
        memcpy(out + (size_t)ov * C, data, sizeof(double) * (size_t)n * C);
        rows = ov + n;
    }
    fb->first_frame = 0;
    memcpy(fb->remainder, out + (size_t)(rows - ov) * C, sizeof(double) * (size_t)ov * C);
    return rows;
}

/* ---- compute_log_power_features (hga_optimized.pyx:27-47) ---------------------------------------
 * window_length / window_shift arrive as C floats (pyx:27); every product with them is float32,
 * exactly as Cython generates it.  pow(x, 2) is x*x (gcc folds it; verified against the .pyx build). */
int oracle_num_windows(int T, int sr, float window_length, float window_shift)
{
    return (int)floor((T - window_length * sr) / (window_shift * sr)) + 1;     /* pyx:36 */
}

void oracle_log_power(const double *data, int T, int C, int sr, float window_length, float window_shift,
                      double *out /* (W, C) */)
{
    int W = oracle_num_windows(T, sr, window_length, window_shift);
    for (int win = 0; win < W; ++win) {
        int start = (int)round((win * window_shift) * sr);                     /* pyx:43 */
        int stop = (int)round(start + window_length * sr);                     /* pyx:44 */
        for (int c = 0; c < C; ++c) {
            double sum = 0.0;
            for (int r = start; r < stop; ++r) {                               /* pyx:20-21 */
                double v = data[(size_t)r * C + c];
                sum += v * v;
            }
            out[(size_t)win * C + c] = log(sum / (stop - start) + 0.01);       /* pyx:22,46 */
        }
    }
}

/* Same as above but stops before the log: mean power + 0.01.  Used by tests to check the device
 * kernel's pre-log value bit-for-bit. */
void oracle_mean_power(const double *data, int T, int C, int sr, float window_length, float window_shift,
                       double *out)
{
    int W = oracle_num_windows(T, sr, window_length, window_shift);
    for (int win = 0; win < W; ++win) {
        int start = (int)round((win * window_shift) * sr);
        int stop = (int)round(start + window_length * sr);
        for (int c = 0; c < C; ++c) {
            double sum = 0.0;
            for (int r = start; r < stop; ++r) {
                double v = data[(size_t)r * C + c];
                sum += v * v;
            }
            out[(size_t)win * C + c] = sum / (stop - start) + 0.01;
        }
    }
}

/* ---- whole extract_features step (local/units.py:145-161 without pre/post transforms) ----------
 * data (n, C) is consumed (filtered in place).  Returns number of frames written to out. */
int oracle_hga_extract(const double *sos_hg, const double *sos_fh, int n_sections,
                       double *zi_hg, double *zi_fh, oracle_framebuf *fb,
                       double *data, int n, int C, int fs, float window_length, float window_shift,
                       double *out)
{
    oracle_sosfilt(sos_hg, n_sections, data, n, C, zi_hg);
    oracle_sosfilt(sos_fh, n_sections, data, n, C, zi_fh);
    double *framed = (double *)malloc(sizeof(double) * (size_t)(fb->overlap + n + fb->frame_length) * C);
    int rows = oracle_framebuf_insert(fb, data, n, framed);
    int W = oracle_num_windows(rows, fs, window_length, window_shift);
    oracle_log_power(framed, rows, C, fs, window_length, window_shift, out);
    free(framed);
    return W;
}
