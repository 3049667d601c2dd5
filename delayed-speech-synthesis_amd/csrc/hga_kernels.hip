// csrc/hga_kernels.hip -- high-gamma feature extraction on gfx950 (float64, bit-exact operation order).
//
// Replaces, for many streams at once:
//   scipy.signal.sosfilt x2 with carried state      local/units.py:151-152
//   WarmStartFrameBuffer.insert                     extensions/hga/hga_optimized.pyx:96-131
//   compute_log_power_features                      extensions/hga/hga_optimized.pyx:27-47
//
// ONE launch per call (hga_fused_kernel): the IIR cascades with their sections spread over the 16 lanes of a DPP row (a
// column advances one sample per step instead of one per 16 dependent biquads); the filtered samples of a tile go into an
// LDS ring instead of HBM, every window that has become complete is summed from the ring (a 50-term sequential sum per
// (window, channel) lane, exactly array_sum_and_power's order), and the last `overlap` rows are left in the row buffer
// for the next call.  Only the input, the W x C frames and 40 rows per column touch HBM: at 1024 streams x 1.04 s that
// removes the 545 MB write + read of the filtered rows which the three-launch form (hga_filter_kernel +
// hga_window_kernel + hga_overlap_kernel, kept as the fallback for window shapes whose ring would not fit LDS) needed.
// (A one-launch form with the front end inside helper waves, hga_stream_kernel, was built in round 3, was exact and 1.4-2x
// slower -- profiles/r3_hga_experiment.md -- and left the tree in round 4.)
// Built with -ffp-contract=off: every product and sum rounds separately, as in scipy's C loop and in
// the reference's Cython kernel, which is what makes the mean power bit-identical.
#include <stdlib.h>

#include "dss_common.h"

struct HgaSos { double k[2][8][6]; };

__device__ __forceinline__ int hga_win_start(int win, float ws, int sr)
{
    // pyx:43  int(round((win * window_shift) * sr)) -- float32 products, C round()
    return (int)round((double)((win * ws) * sr));
}
__device__ __forceinline__ int hga_win_stop(int start, float wl, int sr)
{
    // pyx:44  int(round(start_eeg + window_length * sr))
    return (int)round((double)(start + wl * sr));
}

// ---- stage 1: the two IIR cascades, pipelined over their sections --------------------------------------------
// The 2*nsec (<= 16) second-order sections of one (stream, channel) column occupy the 16 lanes of one DPP row: at
// step k lane r filters sample k - r, taking its input from lane r-1's output of the previous step (row_shr:1), so
// the column advances one sample per step instead of one per 16 biquads.  Each section's arithmetic is exactly
// scipy's sosfilt inner loop (one product, one sum at a time).  A 256-thread block handles 16 consecutive columns
// (same stream, consecutive channels), whose input samples are staged through LDS in tiles of HGA_TT rows with
// 128-byte row segments; lane 2*nsec-1 writes the filtered sample to the column's row buffer at row0 + t.
#define HGA_TT 64
#define HGA_WTAB 256          // windows whose row ranges the fused kernel tabulates in LDS (2 KB)

// lanes 1..15 of every row receive their left neighbour's `y`; lane 0 (no neighbour) keeps `x`: the column's input
__device__ __forceinline__ double hga_shift_in(double x, double y)
{
    const unsigned long long ux = __builtin_bit_cast(unsigned long long, x), uy = __builtin_bit_cast(unsigned long long, y);
    const int lo = __builtin_amdgcn_update_dpp((int)(unsigned)ux, (int)(unsigned)uy, 0x111, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(unsigned)(ux >> 32), (int)(unsigned)(uy >> 32), 0x111, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

// one step of one section: scipy _sosfilt's inner statement order
#define HGA_BIQUAD(IN)                                                                           \
    {                                                                                            \
        y = b0 * (IN) + z0;                        /* x_c = b0*x_n + zi0        */               \
        z0 = b1 * (IN) - a1 * y + z1;              /* zi0 = b1*x_n - a1*x_c + zi1 */             \
        z1 = b2 * (IN) - a2 * y;                   /* zi1 = b2*x_n - a2*x_c     */               \
    }

__global__ void __launch_bounds__(256)
hga_filter_kernel(const double *__restrict__ data, double *__restrict__ zi, double *__restrict__ rowbuf,
                  HgaSos sos, int S, int C, int n, int nsec, int row0, int cap_rows, int zero_rows)
{
    __shared__ double xs[HGA_TT + 1][16];      // inputs of the tile's steps, one row segment of 16 columns per step (+1: prefetch)
    __shared__ double ys[HGA_TT][16];          // outputs produced at the tile's steps (sample k - (2*nsec-1) at step k)
    __shared__ double coef[16][5];
    __shared__ double dump[256];               // where the lanes that do not hold the last section "store" their y
    const int tid = threadIdx.x, r = tid & 15, pib = tid >> 4;
    const long total = (long)S * C;
    const long pair0 = (long)blockIdx.x * 16;
    long pair = pair0 + pib;
    const bool valid = pair < total;
    if (!valid) pair = total - 1;
    const int s = (int)(pair / C), c = (int)(pair - (long)s * C);
    const int nsec2 = 2 * nsec;
    const bool has_sec = r < nsec2;
    const bool is_last = r == nsec2 - 1;
    // branch-free output store of the steady-state loop: the last section's lane walks down its ys column, every other
    // lane rewrites its own dump word
    char *const ybase = is_last ? reinterpret_cast<char *>(&ys[0][pib]) : reinterpret_cast<char *>(&dump[tid]);
    const int ystride = is_last ? 16 * (int)sizeof(double) : 0;
    const int f = has_sec ? r / nsec : 0, q = has_sec ? r - f * nsec : 0;
    if (tid < 16) {           // section coefficients by lane (a kernel argument cannot be indexed per lane)
        const int ff = tid < nsec2 ? tid / nsec : 0, qq = tid < nsec2 ? tid - ff * nsec : 0;
        coef[tid][0] = sos.k[ff][qq][0]; coef[tid][1] = sos.k[ff][qq][1]; coef[tid][2] = sos.k[ff][qq][2];
        coef[tid][3] = sos.k[ff][qq][4]; coef[tid][4] = sos.k[ff][qq][5];
    }
    double *zp = zi + (size_t)s * 2 * 8 * 2 * C + c;
    double z0 = zp[((f * 8 + q) * 2 + 0) * (size_t)C], z1 = zp[((f * 8 + q) * 2 + 1) * (size_t)C];
    // CASE 2 of the frame buffer (first chunk shorter than a frame): left zero padding, pyx:116
    if (valid) {
        double *col = rowbuf + (size_t)s * cap_rows * C + c;
        for (int k = r; k < zero_rows; k += 16) col[(size_t)k * C] = 0.0;
    }
    __syncthreads();
    const double b0 = coef[r][0], b1 = coef[r][1], b2 = coef[r][2], a1 = coef[r][3], a2 = coef[r][4];
    const int steps = n + nsec2 - 1;
    double y = 0.0;
    for (int base = 0; base < steps; base += HGA_TT) {
        __syncthreads();                                   // previous tile fully consumed and written out
        for (int idx = tid; idx < HGA_TT * 16; idx += 256) {
            const int tt = idx >> 4, p = idx & 15;
            const long pp = pair0 + p;
            const int t = base + tt;
            double v = 0.0;
            if (t < n && pp < total) {
                const int sp = (int)(pp / C), cp = (int)(pp - (long)sp * C);
                v = data[((size_t)sp * n + t) * C + cp];
            }
            xs[tt][p] = v;
        }
        __syncthreads();
        const int kend = min(base + HGA_TT, steps);
        int k = base;
        // steps on which some lanes have no sample yet (pipeline filling) or no more (draining): predicated
        for (; k < kend && (k < nsec2 - 1 || k >= n); ++k) {
            const double in = hga_shift_in(xs[k - base][pib], y);
            const int t = k - r;
            if (has_sec && t >= 0 && t < n) {
                HGA_BIQUAD(in)
                if (is_last) ys[k - base][pib] = y;
            }
        }
        // steady state: every section has a sample
        const int ksteady = min(kend, n);
        {
            char *yp = ybase + (k - base) * ystride;
            const double *xp = &xs[k - base][pib];
            double xcur = *xp;
            for (; k < ksteady; ++k) {
                xp += 16;
                const double xnext = *xp;                  // next step's input: its LDS latency hides under this step
                const double in = hga_shift_in(xcur, y);
                HGA_BIQUAD(in)
                *reinterpret_cast<double *>(yp) = y;
                yp += ystride;
                xcur = xnext;
            }
        }
        for (; k < kend; ++k) {                            // draining steps at the end of the last tile(s)
            const double in = hga_shift_in(xs[k - base][pib], y);
            const int t = k - r;
            if (has_sec && t >= 0 && t < n) {
                HGA_BIQUAD(in)
                if (is_last) ys[k - base][pib] = y;
            }
        }
        __syncthreads();
        // the tile's finished samples go out as 128-byte row segments
        for (int idx = tid; idx < HGA_TT * 16; idx += 256) {
            const int tt = idx >> 4, p = idx & 15;
            const long pp = pair0 + p;
            const int t = base + tt - (nsec2 - 1);
            if (base + tt < kend && t >= 0 && t < n && pp < total) {
                const int sp = (int)(pp / C), cp = (int)(pp - (long)sp * C);
                rowbuf[((size_t)sp * cap_rows + row0 + t) * C + cp] = ys[tt][p];
            }
        }
    }
    if (valid && has_sec) {
        zp[((f * 8 + q) * 2 + 0) * (size_t)C] = z0;
        zp[((f * 8 + q) * 2 + 1) * (size_t)C] = z1;
    }
}

// ---- fused form: filter tile -> LDS ring -> completed windows -> overlap rows -------------------------------------------
// Same block shape and the same per-step code as hga_filter_kernel (16 columns x 16 section lanes).  `ring` holds the last
// R rows of the block's 16 columns (overlap / zero rows first, then the new samples at row0 + t), row i at position i mod R;
// R >= frame_length + window shift + TT + 2 so that no row a pending window still needs is overwritten by the next tile.
// Round 3: R carries no power-of-two padding and the lanes that do not hold the last section have no dummy store targets
// any more (18 KB of static LDS + 16 KB of ring instead of 37 KB).  (A 32-step-tile instantiation -- 22 KB, 7 blocks per CU
// instead of 5 -- was no faster at 1024 streams and slower on small calls, profiles/r3_hga_tile_ab_timing.txt, and is gone.)
#define HGF_WTAB 128
__global__ void __launch_bounds__(256, 4)
hga_fused_kernel(const double *__restrict__ data, double *__restrict__ zi, double *__restrict__ rowbuf, double *__restrict__ out,
                 HgaSos sos, int S, int C, int n, int nsec, int row0, int cap_rows, int zero_rows, int rows, int W, int overlap,
                 int sr, float wl, float ws, int apply_log, int R, const double *__restrict__ zs_mean,
                 const double *__restrict__ zs_std)
{
    __shared__ double xs[HGA_TT + 1][16];
    __shared__ double ys[HGA_TT][16];
    __shared__ double coef[16][5];
    __shared__ int wtab[2][HGF_WTAB];          // first and one-past-last row of the first HGF_WTAB windows (the float32 /
                                               //   round() arithmetic of pyx:43-44 once per block instead of once per tile)
    extern __shared__ __attribute__((aligned(16))) double ring[];         // [R][16]
    const int tid = threadIdx.x, r = tid & 15, pib = tid >> 4;
    const long total = (long)S * C;
    const long pair0 = (long)blockIdx.x * 16;
    long pair = pair0 + pib;
    const bool valid = pair < total;
    if (!valid) pair = total - 1;
    const int s = (int)(pair / C), c = (int)(pair - (long)s * C);
    const int nsec2 = 2 * nsec;
    const bool has_sec = r < nsec2;
    const bool is_last = r == nsec2 - 1;
    const int f = has_sec ? r / nsec : 0, q = has_sec ? r - f * nsec : 0;
    if (tid < 16) {
        const int ff = tid < nsec2 ? tid / nsec : 0, qq = tid < nsec2 ? tid - ff * nsec : 0;
        coef[tid][0] = sos.k[ff][qq][0]; coef[tid][1] = sos.k[ff][qq][1]; coef[tid][2] = sos.k[ff][qq][2];
        coef[tid][3] = sos.k[ff][qq][4]; coef[tid][4] = sos.k[ff][qq][5];
    }
    double *zp = zi + (size_t)s * 2 * 8 * 2 * C + c;
    double z0 = zp[((f * 8 + q) * 2 + 0) * (size_t)C], z1 = zp[((f * 8 + q) * 2 + 1) * (size_t)C];
    for (int w = tid; w < W && w < HGF_WTAB; w += 256) {
        const int st = hga_win_start(w, ws, sr);
        wtab[0][w] = st;
        wtab[1][w] = hga_win_stop(st, wl, sr);
    }
    // rows [0, row0) of the ring: zeros (CASE 2, pyx:116) or the overlap the previous call left (CASE 3, pyx:123-131)
    for (int idx = tid; idx < row0 * 16; idx += 256) {
        const int rr = idx >> 4, p = idx & 15;
        const long pp = pair0 + p;
        double v = 0.0;
        if (rr >= zero_rows && pp < total) {
            const int sp = (int)(pp / C), cp = (int)(pp - (long)sp * C);
            v = rowbuf[((size_t)sp * cap_rows + rr) * C + cp];
        }
        ring[(rr % R) * 16 + p] = v;
    }
    __syncthreads();
    const double b0 = coef[r][0], b1 = coef[r][1], b2 = coef[r][2], a1 = coef[r][3], a2 = coef[r][4];
    const int steps = n + nsec2 - 1;
    double y = 0.0;
    int w_next = 0;                                        // first window not yet written (block-uniform)
    // Input tiles: thread (tt0, p) owns column p of rows tt0, tt0 + 16, ... of every tile.  The next tile's samples are
    // fetched into registers before the current tile's steps run and go to LDS after them: the HBM latency hides under
    // the filter arithmetic instead of standing between two barriers.
    const int lp = tid & 15, ltt = tid >> 4;
    const long lpp = pair0 + lp;
    const bool lvalid = lpp < total;
    const double *lcol = data;
    if (lvalid) { const int sp = (int)(lpp / C), cp = (int)(lpp - (long)sp * C); lcol = data + (size_t)sp * n * C + cp; }
    double pre[HGA_TT / 16];
#define HGF_FETCH(BASE)                                                                          \
    _Pragma("unroll") for (int j = 0; j < HGA_TT / 16; ++j) {                                        \
        const int t = (BASE) + ltt + 16 * j;                                                     \
        pre[j] = (lvalid && t < n) ? lcol[(size_t)t * C] : 0.0;                                  \
    }
    HGF_FETCH(0)
    // ring position of row (row0 + t) for the first output of the current tile, kept modulo R (block-uniform)
    int tile_pos = (row0 + R * 16 - (nsec2 - 1)) % R;      // row0 + t for t = 0 - (nsec2 - 1): the row "before" the first output
    for (int base = 0; base < steps; base += HGA_TT) {
        __syncthreads();                                   // previous tile fully consumed
#pragma unroll
        for (int j = 0; j < HGA_TT / 16; ++j) xs[ltt + 16 * j][lp] = pre[j];
        if (base + HGA_TT < steps) { HGF_FETCH(base + HGA_TT) }
        __syncthreads();
        const int kend = min(base + HGA_TT, steps);
        int k = base;
        for (; k < kend && (k < nsec2 - 1 || k >= n); ++k) {
            const double in = hga_shift_in(xs[k - base][pib], y);
            const int t = k - r;
            if (has_sec && t >= 0 && t < n) {
                HGA_BIQUAD(in)
                if (is_last) ys[k - base][pib] = y;
            }
        }
        const int ksteady = min(kend, n);
        {
            const double *xp = &xs[k - base][pib];
            double xcur = *xp;
            // four steps per trip with constant LDS offsets (a DPP move is convergent: the compiler will not unroll a loop of
            // unknown length around it): per step 9 fp64 operations, 2 DPP moves, one LDS read, one LDS write
            double *const yl = &ys[k - base][pib];
            int kk = 0;
            for (; k + 4 <= ksteady; k += 4, kk += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double xnext = xp[16 * (kk + u + 1)];
                    const double in = hga_shift_in(xcur, y);
                    HGA_BIQUAD(in)
                    if (is_last) yl[16 * (kk + u)] = y;
                    xcur = xnext;
                }
            }
            for (; k < ksteady; ++k, ++kk) {
                const double xnext = xp[16 * (kk + 1)];
                const double in = hga_shift_in(xcur, y);
                HGA_BIQUAD(in)
                if (is_last) yl[16 * kk] = y;
                xcur = xnext;
            }
        }
        for (; k < kend; ++k) {
            const double in = hga_shift_in(xs[k - base][pib], y);
            const int t = k - r;
            if (has_sec && t >= 0 && t < n) {
                HGA_BIQUAD(in)
                if (is_last) ys[k - base][pib] = y;
            }
        }
        __syncthreads();
        // the tile's finished samples join the ring at their row (position row mod R, advanced from tile to tile)
        for (int idx = tid; idx < HGA_TT * 16; idx += 256) {
            const int tt = idx >> 4, p = idx & 15;
            const int t = base + tt - (nsec2 - 1);
            int pos = tile_pos + tt;
            if (pos >= R) pos -= R;
            if (base + tt < kend && t >= 0 && t < n) ring[pos * 16 + p] = ys[tt][p];
        }
        tile_pos += HGA_TT;
        if (tile_pos >= R) tile_pos -= R;
        __syncthreads();
        // every window that is complete now: lane = (window, column), a sequential sum over its rows (pyx:9-22, 42-46)
        int t_done = kend - (nsec2 - 1);
        t_done = t_done < 0 ? 0 : (t_done > n ? n : t_done);
        const int rows_done = row0 + t_done;
        {
            const int p = tid & 15, wi = tid >> 4;
            const long pp = pair0 + p;
            for (int w = w_next + wi; w < W; w += 16) {
                const int start = w < HGF_WTAB ? wtab[0][w] : hga_win_start(w, ws, sr);
                const int stop = w < HGF_WTAB ? wtab[1][w] : hga_win_stop(start, wl, sr);
                if (stop > rows_done) break;
                double sum = 0.0;
                // the window's rows sit at positions start mod R ... (wrapping once at most: stop - start <= R)
                int pos = start % R, left = stop - start;
                while (left > 0) {
                    const int run = min(left, R - pos);
                    const double *rp = ring + pos * 16 + p;
                    int i = 0;
                    for (; i + 8 <= run; i += 8) {             // eight ring reads in flight, then their terms in row order
                        double v[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) v[u] = rp[(i + u) * 16];
#pragma unroll
                        for (int u = 0; u < 8; ++u) sum += v[u] * v[u];
                    }
                    for (; i < run; ++i) {
                        const double v = rp[i * 16];
                        sum += v * v;
                    }
                    left -= run;
                    pos = 0;
                }
                if (pp < total) {
                    const int sp = (int)(pp / C), cp = (int)(pp - (long)sp * C);
                    const double pw = sum / (double)(stop - start) + 0.01;
                    double o = apply_log ? log(pw) : pw;
                    if (zs_mean) o = (o - zs_mean[cp]) / zs_std[cp];          // ZScoreNormalization (local/common.py:367-376)
                    out[((size_t)sp * W + w) * C + cp] = o;
                }
            }
            while (w_next < W && (w_next < HGF_WTAB ? wtab[1][w_next] : hga_win_stop(hga_win_start(w_next, ws, sr), wl, sr)) <= rows_done)
                ++w_next;
        }
    }
    __syncthreads();
    // the last `overlap` rows become rows 0..overlap-1 of the next call (WarmStartFrameBuffer.remainder_data)
    for (int idx = tid; idx < overlap * 16; idx += 256) {
        const int kk = idx >> 4, p = idx & 15;
        const long pp = pair0 + p;
        if (pp < total) {
            const int sp = (int)(pp / C), cp = (int)(pp - (long)sp * C);
            rowbuf[((size_t)sp * cap_rows + kk) * C + cp] = ring[((rows - overlap + kk) % R) * 16 + p];
        }
    }
    if (valid && has_sec) {
        zp[((f * 8 + q) * 2 + 0) * (size_t)C] = z0;
        zp[((f * 8 + q) * 2 + 1) * (size_t)C] = z1;
    }
#undef HGF_FETCH
}

// ---- stage 2: windowed mean power ---------------------------------------------------------------------------------
// A 256-thread block takes 8 consecutive windows x 32 consecutive channels of one stream: the rows those windows cover
// (50 + 7*10 for the reference's 50 ms / 10 ms at 1 kHz) are staged once through LDS in 256-byte row segments instead
// of being fetched by each of the five windows that overlap them; lane (window, channel) then runs its window's
// sequential sum over rows exactly as array_sum_and_power does (pyx:9-22).
#define HGA_WB 8
#define HGA_WC 32
#define HGA_WROWS 160            // LDS rows per block; a block whose windows span more falls back to global reads

__global__ void __launch_bounds__(256)
hga_window_kernel(const double *__restrict__ rowbuf, double *__restrict__ out, int S, int C, int W, int cap_rows, int sr,
                  float wl, float ws, int apply_log, const double *__restrict__ zs_mean, const double *__restrict__ zs_std)
{
    __shared__ double tile[HGA_WROWS][HGA_WC];
    const int tid = threadIdx.x, cl = tid & (HGA_WC - 1), wi = tid / HGA_WC;
    const int cgroups = (C + HGA_WC - 1) / HGA_WC, wchunks = (W + HGA_WB - 1) / HGA_WB;
    int bid = blockIdx.x;
    const int wc = bid % wchunks; bid /= wchunks;
    const int cg = bid % cgroups;
    const int s = bid / cgroups;
    const int w0 = wc * HGA_WB, wlast = min(w0 + HGA_WB, W) - 1;
    const int row_lo = hga_win_start(w0, ws, sr);
    const int row_hi = hga_win_stop(hga_win_start(wlast, ws, sr), wl, sr);
    const int nrows = row_hi - row_lo;
    const bool staged = nrows <= HGA_WROWS;
    const double *base = rowbuf + (size_t)s * cap_rows * C;
    const int c0 = cg * HGA_WC;
    if (staged) {
        for (int idx = tid; idx < nrows * HGA_WC; idx += 256) {
            const int rr = idx / HGA_WC, cc = idx - rr * HGA_WC;
            tile[rr][cc] = (c0 + cc < C) ? base[(size_t)(row_lo + rr) * C + c0 + cc] : 0.0;
        }
    }
    __syncthreads();
    const int win = w0 + wi, c = c0 + cl;
    if (win >= W || c >= C) return;
    const int start = hga_win_start(win, ws, sr);
    const int stop = hga_win_stop(start, wl, sr);
    double sum = 0.0;
    if (staged) {
        for (int rr = start; rr < stop; ++rr) {
            const double v = tile[rr - row_lo][cl];
            sum += v * v;
        }
    } else {
        for (int rr = start; rr < stop; ++rr) {
            const double v = base[(size_t)rr * C + c];
            sum += v * v;
        }
    }
    const double p = sum / (double)(stop - start) + 0.01;
    double o = apply_log ? log(p) : p;
    if (zs_mean) o = (o - zs_mean[c]) / zs_std[c];                    // ZScoreNormalization (local/common.py:367-376)
    out[((size_t)s * W + win) * C + c] = o;
}

// ---- stage 3: keep the last `overlap` rows (ascending copy: a source row is always ahead of its destination) ------
__global__ void __launch_bounds__(256)
hga_overlap_kernel(double *__restrict__ rowbuf, int S, int C, int cap_rows, int rows, int overlap)
{
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)S * C) return;
    const int s = (int)(gid / C), c = (int)(gid - (long)s * C);
    double *col = rowbuf + (size_t)s * cap_rows * C + c;
    for (int k = 0; k < overlap; ++k) col[(size_t)k * C] = col[(size_t)(rows - overlap + k) * C];
}

int dss_launch_hga(const DssHgaDev &h, const double *d_data, const DssHgaFrontDev *fe, int n, int row0, int zero_rows, int rows,
                   int W, double *d_out, int apply_log, hipStream_t st)
{
    HgaSos sos;
    memcpy(&sos, h.sos, sizeof(sos));
    const long pairs = (long)h.S * h.C;
    const int shift = h.frame_length - h.overlap;
    if (fe) { dss_set_error("HGA: raw amplifier rows go through dss_launch_hga_frontend first"); return DSS_EINVAL; }
    if (h.force_path != 2) {   // fused form when the ring (frame + shift + one tile of rows) fits beside the tiles
        const int TT = HGA_TT;
        int ring_rows = h.frame_length + shift + TT + 2;
        ring_rows = (ring_rows + 7) & ~7;
        const size_t ring_bytes = (size_t)ring_rows * 16 * sizeof(double);
        if (ring_bytes <= 40 * 1024 && rows - h.overlap + TT <= (1 << 30) && h.overlap <= ring_rows && row0 <= ring_rows) {
            hipLaunchKernelGGL(hga_fused_kernel, dim3((unsigned)((pairs + 15) / 16)), dim3(256), ring_bytes, st, d_data, h.zi,
                                   h.rows, d_out, sos, h.S, h.C, n, h.nsec, row0, h.cap_rows, zero_rows, rows, W, h.overlap, h.fs, h.wl,
                                   h.ws, apply_log, ring_rows, h.zs_mean, h.zs_std);
            DSS_HIP_CHECK(hipGetLastError());
            return DSS_OK;
        }
    }
    hipLaunchKernelGGL(hga_filter_kernel, dim3((unsigned)((pairs + 15) / 16)), dim3(256), 0, st, d_data, h.zi, h.rows, sos,
                       h.S, h.C, n, h.nsec, row0, h.cap_rows, zero_rows);
    DSS_HIP_CHECK(hipGetLastError());
    if (W > 0) {
        const long blocks = (long)h.S * ((h.C + HGA_WC - 1) / HGA_WC) * ((W + HGA_WB - 1) / HGA_WB);
        hipLaunchKernelGGL(hga_window_kernel, dim3((unsigned)blocks), dim3(256), 0, st, h.rows, d_out, h.S, h.C, W, h.cap_rows,
                           h.fs, h.wl, h.ws, apply_log, h.zs_mean, h.zs_std);
        DSS_HIP_CHECK(hipGetLastError());
    }
    hipLaunchKernelGGL(hga_overlap_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, st, h.rows, h.S, h.C,
                       h.cap_rows, rows, h.overlap);
    DSS_HIP_CHECK(hipGetLastError());
    return DSS_OK;
}

// state <- sosfilt_zi tiled over channels (units.py:128-132); zero the overlap rows (pyx:79-82)
__global__ void hga_reset_kernel(double *zi, double *rowbuf, const double *zi_hg, const double *zi_fh, int S, int C,
                                 int nsec, int cap_rows, int overlap)
{
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= S * C) return;
    const int s = gid / C, c = gid - s * C;
    double *zp = zi + (size_t)s * 2 * 8 * 2 * C + c;
    for (int f = 0; f < 2; ++f)
        for (int q = 0; q < 8; ++q)
            for (int k = 0; k < 2; ++k) {
                const double *src = f ? zi_fh : zi_hg;
                zp[((f * 8 + q) * 2 + k) * (size_t)C] = (q < nsec) ? src[q * 2 + k] : 0.0;
            }
    double *col = rowbuf + (size_t)s * cap_rows * C + c;
    for (int r = 0; r < overlap; ++r) col[(size_t)r * C] = 0.0;
}

int dss_launch_hga_reset(const DssHgaDev &h, const double *d_zi_hg, const double *d_zi_fh, hipStream_t st)
{
    const int total = h.S * h.C;
    hipLaunchKernelGGL(hga_reset_kernel, dim3((total + 255) / 256), dim3(256), 0, st, h.zi, h.rows, d_zi_hg, d_zi_fh, h.S,
                       h.C, h.nsec, h.cap_rows, h.overlap);
    DSS_HIP_CHECK(hipGetLastError());
    return DSS_OK;
}

// Stateless compute_log_power_features: one lane per (window, channel); 50-term sequential sum.
__global__ void __launch_bounds__(256)
hga_log_power_kernel(const double *__restrict__ data, double *__restrict__ out, int T, int C, int W, int sr, float wl,
                     float ws, int apply_log)
{
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= W * C) return;
    const int win = gid / C, c = gid - win * C;
    const int start = hga_win_start(win, ws, sr);
    const int stop = hga_win_stop(start, wl, sr);
    double sum = 0.0;
    for (int r = start; r < stop; ++r) {
        const double q = data[(size_t)r * C + c];
        sum += q * q;
    }
    const double p = sum / (double)(stop - start) + 0.01;
    out[gid] = apply_log ? log(p) : p;
}

int dss_launch_log_power(const double *d_data, int T, int C, int sr, float wl, float ws, int W, double *d_out,
                         int apply_log, hipStream_t st)
{
    const int total = W * C;
    if (total <= 0) return DSS_OK;
    hipLaunchKernelGGL(hga_log_power_kernel, dim3((total + 255) / 256), dim3(256), 0, st, d_data, d_out, T, C, W, sr, wl,
                       ws, apply_log);
    DSS_HIP_CHECK(hipGetLastError());
    return DSS_OK;
}


// Fused front end: column reorder + per-grid common average reference + channel selection.  A block stages FE_ROWS
// consecutive (stream, sample) rows of the raw packet through LDS with coalesced loads (lanes across the c_raw columns;
// the odd row length keeps column walks conflict-free), then one lane per (row, grid) runs the grid mean as the
// SEQUENTIAL sum the reference's numpy expression performs (np.mean over a Fortran-ordered fancy-index copy adds one
// column at a time) followed by one division, and finally lanes across the C output channels subtract and store
// coalesced.  Bit-identical to the numpy chain (local/common.py:16-58,308-345).
#define FE_ROWS 32
__global__ void __launch_bounds__(256)
hga_frontend_kernel(const double *__restrict__ raw, double *__restrict__ pre, int total, int c_raw, int C,
                    const int *__restrict__ src_col, const int *__restrict__ grid_of, int n_grids,
                    const int *__restrict__ comp_cols, const int *__restrict__ comp_off)
{
    extern __shared__ __attribute__((aligned(16))) double fe_tile[];     // [FE_ROWS][c_raw], [FE_ROWS][4] means, [<= 4 * c_raw] member columns
    double *means = fe_tile + (size_t)FE_ROWS * c_raw;
    int *cols = reinterpret_cast<int *>(means + FE_ROWS * 4);            // the grids' member columns, in summation order
    const int tid = threadIdx.x;
    const long row_base = (long)blockIdx.x * FE_ROWS;
    const int nrows = (int)min((long)FE_ROWS, (long)total - row_base);
    const double *src = raw + (size_t)row_base * c_raw;
    // rows are contiguous: fully coalesced; eight loads per thread in flight (a loop of unknown length is not unrolled,
    // and one load per trip would expose the HBM latency sixteen times per tile)
    const int n_el = nrows * c_raw;
    for (int base = 0; base < n_el; base += 8 * 256) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int idx = base + u * 256 + tid; v[u] = idx < n_el ? src[idx] : 0.0; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int idx = base + u * 256 + tid; if (idx < n_el) fe_tile[idx] = v[u]; }
    }
    for (int k = tid; k < comp_off[n_grids]; k += 256) cols[k] = comp_cols[k];
    __syncthreads();
    if (tid < nrows * n_grids) {
        const int rr = tid / n_grids, g = tid - rr * n_grids;
        const double *row = fe_tile + (size_t)rr * c_raw;
        const int a = comp_off[g], b2 = comp_off[g + 1];
        double sum = 0.0;
        int k = a;
        for (; k + 8 <= b2; k += 8) {              // eight members' reads in flight, then their terms in list order
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = row[cols[k + u]];
#pragma unroll
            for (int u = 0; u < 8; ++u) sum += v[u];
        }
        for (; k < b2; ++k) sum += row[cols[k]];
        means[rr * 4 + g] = sum / (double)(b2 - a);
    }
    __syncthreads();
    const int n_out = nrows * C;                    // eight independent elements per thread and trip, as for the loads
    for (int base = 0; base < n_out; base += 8 * 256) {
        double o[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * 256 + tid;
            o[u] = 0.0;
            if (idx < n_out) {
                const int rr = idx / C, c = idx - rr * C;
                const int g = grid_of[c];
                const double v = fe_tile[(size_t)rr * c_raw + src_col[c]];
                o[u] = g >= 0 ? v - means[rr * 4 + g] : v;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * 256 + tid;
            if (idx < n_out) pre[(size_t)row_base * C + idx] = o[u];
        }
    }
}

// Amplifier payloads as they arrive on the wire -- float32, channel-major: [stream][channel][sample] (the body of a BCI2000
// packet behind its 7-byte header, reference local/units.py:78-82 and development_amplifier.py:14-25) -- to the float64
// time-major rows [stream][sample][channel] every other entry point takes: what ZMQConnector.interpret_bytes does on the host
// with reshape + transpose + astype(float64), for all streams at once.  float32 -> float64 is exact, so results do not change.
// One workgroup per stream; the payload goes through LDS so that both the read and the write are linear.
__global__ void __launch_bounds__(256) hga_wire_kernel(const float *__restrict__ payload, double *__restrict__ rows, int C, int n)
{
    extern __shared__ float wtile[];                          // [channel][sample], as it arrived
    const int s = blockIdx.x, total = C * n;
    const float *src = payload + (size_t)s * total;
    for (int k = threadIdx.x; k < total; k += 256) wtile[k] = src[k];
    __syncthreads();
    double *dst = rows + (size_t)s * total;
    for (int k = threadIdx.x; k < total; k += 256) {
        const int t = k / C, c = k - t * C;
        dst[k] = (double)wtile[c * n + t];
    }
}

int dss_launch_hga_wire(const float *d_payload, double *d_rows, int S, int C, int n, hipStream_t s)
{
    const size_t lds = (size_t)C * n * sizeof(float);
    if (lds > 64 * 1024) { dss_set_error("wire packet of %d channels x %d samples exceeds the 64 KB tile", C, n); return DSS_EINVAL; }
    hipLaunchKernelGGL(hga_wire_kernel, dim3(S), dim3(256), lds, s, d_payload, d_rows, C, n);
    DSS_HIP_CHECK(hipGetLastError());
    return DSS_OK;
}

int dss_launch_hga_frontend(const double *d_raw, double *d_pre, int S, int n, int c_raw, int C, const int *src_col,
                            const int *grid_of, int n_grids, const int *comp_cols, const int *comp_off, hipStream_t st)
{
    const int total = S * n;
    const size_t lds = ((size_t)FE_ROWS * c_raw + FE_ROWS * 4) * sizeof(double) + (size_t)4 * c_raw * sizeof(int);   // member lists: at most 4 grids of c_raw columns
    if (lds > 64 * 1024 || n_grids > 4) { dss_set_error("front end: %d raw columns / %d grids exceed the tile", c_raw, n_grids); return DSS_EINVAL; }
    hipLaunchKernelGGL(hga_frontend_kernel, dim3((total + FE_ROWS - 1) / FE_ROWS), dim3(256), lds, st, d_raw, d_pre, total, c_raw, C,
                       src_col, grid_of, n_grids, comp_cols, comp_off);
    DSS_HIP_CHECK(hipGetLastError());
    return DSS_OK;
}
