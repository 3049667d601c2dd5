/* Development aid / proof: the device's short form of xiph's lin2ulaw() (csrc/lpcnet_device.h dss_lin2ulaw) returns the
 * same value as the reference form (oracle/lpcnet_oracle.c lin2ulaw, a restatement of xiph common.h) for EVERY fp32 input.
 *
 *   gcc -O2 -ffp-contract=off -fopenmp -o /tmp/l2u tools/verify/lin2ulaw_exhaustive.c -Loracle -loracle -lm
 *   LD_LIBRARY_PATH=oracle /tmp/l2u [stride]
 *
 * What differs in the short form: (1) u / 5.5451774445f as a product with the reciprocal and two fused corrections
 * (correctly rounded for every quotient that can occur; the loop below also checks the division alone over all floats),
 * (2) the clamp as max/min, (3) floor(.5 + (double)u) as floorf(u + .5f) (exact for 0 <= u <= 255), (4) the sign and the
 * factor 128 as one multiplication by +-128.
 * All 2^32 bit patterns are visited (stride 1); NaN inputs are skipped: (int) of a NaN is undefined in the C source. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int oracle_lin2ulaw(float x);

static float log2_approx_dev(float x)
{
    int32_t in;
    memcpy(&in, &x, 4);
    const int integer = (in >> 23) - 127;
    in -= integer << 23;
    float f;
    memcpy(&f, &in, 4);
    float frac = f - 1.5f;
    frac = -0.41445418f + frac * (0.95909232f + frac * (-0.33951290f + frac * 0.16541097f));
    return 1 + integer + frac;
}

#define DSS_LOG256 5.5451774445f
static inline float div_log256(float num)          /* == num / 5.5451774445f, see main() */
{
    const float rc = 1.0f / DSS_LOG256;
    const float q0 = num * rc;
    const float r = fmaf(-q0, DSS_LOG256, num);
    return fmaf(r, rc, q0);
}

static int lin2ulaw_dev(float x)
{
    const float scale = 255.f / 32768.f;
    const float s128 = (x < 0) ? -128.f : 128.f;
    x = fabsf(x);
    float u = s128 * (0.69315f * log2_approx_dev(1 + scale * x));
    u = div_log256(u);
    u = 128 + u;
    u = fminf(fmaxf(u, 0.f), 255.f);
    return (int)floorf(u + .5f);
}

int main(int argc, char **argv)
{
    const uint64_t stride = argc > 1 ? strtoull(argv[1], 0, 10) : 1;
    uint64_t bad = 0, bad_div = 0, seen = 0;
#pragma omp parallel for reduction(+ : bad, bad_div, seen) schedule(static)
    for (int64_t chunk = 0; chunk < 4096; ++chunk) {
        for (uint64_t k = (uint64_t)chunk << 20; k < ((uint64_t)chunk + 1) << 20; k += stride) {
            const uint32_t bits = (uint32_t)k;
            float x;
            memcpy(&x, &bits, 4);
            if (x != x) continue;
            ++seen;
            if (lin2ulaw_dev(x) != oracle_lin2ulaw(x)) {
                if (bad < 5) fprintf(stderr, "lin2ulaw differs at %a: %d vs %d\n", x, lin2ulaw_dev(x), oracle_lin2ulaw(x));
                ++bad;
            }
            /* the division by itself, over the magnitudes the quotient's numerator can take (|num| < 2^11) and beyond */
            const float q = x / DSS_LOG256, qd = div_log256(x);
            if (memcmp(&q, &qd, 4) != 0 && isfinite(x) && fabsf(x) < 0x1p100f && fabsf(x) > 0x1p-100f) {
                if (bad_div < 5) fprintf(stderr, "division differs at %a: %a vs %a\n", x, q, qd);
                ++bad_div;
            }
        }
    }
    printf("inputs visited: %llu (stride %llu), lin2ulaw mismatches: %llu, division mismatches for 2^-100 < |x| < 2^100: %llu\n",
           (unsigned long long)seen, (unsigned long long)stride, (unsigned long long)bad, (unsigned long long)bad_div);
    return bad || bad_div ? 1 : 0;
}
