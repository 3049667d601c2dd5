// csrc/lpcnet_device.h -- device-side scalar helpers of the LPCNet kernels.
//
// Each helper restates one inline function of xiph/LPCNet (src/vec.h generic path, src/common.h) in the
// exact operation order and precision of the C source: float where the C is float, double where C's
// usual arithmetic conversions promote to double.  The library is built with -ffp-contract=off so no
// product/sum pair is fused.  Tables (tansig, ulaw2lin, sampling logits) are computed by the host's
// libm when a model is loaded and passed in; the device never evaluates exp/log/tanh on this path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// vec.h tanh_approx (201-entry table form).  `tab` may live in LDS or global memory.
__device__ __forceinline__ float dss_tanh_approx(const float *tab, float x)
{
    float sign = 1.f;
    if (x < 0) { x = -x; sign = -1.f; }
    int i = (int)floorf(.5f + 25 * x);
    i = i < 0 ? 0 : i;
    i = i > 200 ? 200 : i;
    x -= .04f * i;
    const float y = tab[i];
    const float dy = 1 - y * y;
    const float r = y + x * dy * (1 - y * x);
    return sign * r;
}

// Two independent evaluations with both table reads issued before either is consumed (same arithmetic per
// element as dss_tanh_approx; only the instruction interleaving differs).
__device__ __forceinline__ void dss_tanh_approx2(const float *tab, float x1, float x2, float &o1, float &o2)
{
    float s1 = 1.f, s2 = 1.f;
    if (x1 < 0) { x1 = -x1; s1 = -1.f; }
    if (x2 < 0) { x2 = -x2; s2 = -1.f; }
    int i1 = (int)floorf(.5f + 25 * x1);
    int i2 = (int)floorf(.5f + 25 * x2);
    i1 = i1 < 0 ? 0 : i1; i1 = i1 > 200 ? 200 : i1;
    i2 = i2 < 0 ? 0 : i2; i2 = i2 > 200 ? 200 : i2;
    const float y1 = tab[i1];
    const float y2 = tab[i2];
    x1 -= .04f * i1;
    x2 -= .04f * i2;
    const float d1 = 1 - y1 * y1, d2 = 1 - y2 * y2;
    const float r1 = y1 + x1 * d1 * (1 - y1 * x1);
    const float r2 = y2 + x2 * d2 * (1 - y2 * x2);
    o1 = s1 * r1;
    o2 = s2 * r2;
}

__device__ __forceinline__ void dss_sigmoid_approx2(const float *tab, float x1, float x2, float &o1, float &o2)
{
    float t1, t2;
    dss_tanh_approx2(tab, .5f * x1, .5f * x2, t1, t2);
    o1 = .5f + .5f * t1;
    o2 = .5f + .5f * t2;
}

__device__ __forceinline__ float dss_sigmoid_approx(const float *tab, float x)
{
    return .5f + .5f * dss_tanh_approx(tab, .5f * x);
}

// common.h log2_approx / lin2ulaw
__device__ __forceinline__ float dss_log2_approx(float x)
{
    int in = __float_as_int(x);
    const int integer = (in >> 23) - 127;
    in -= integer << 23;
    float frac = __int_as_float(in) - 1.5f;
    frac = -0.41445418f + frac * (0.95909232f + frac * (-0.33951290f + frac * 0.16541097f));
    return 1 + integer + frac;
}

// xiph common.h lin2ulaw(), in a shorter instruction sequence that returns the same value for every fp32 input
// (tools/verify/lin2ulaw_exhaustive.c visits all 2^32 bit patterns against the test suite's restatement of the C source):
//   * u / 5.5451774445f as q0 = u * (1/c), r = fma(-q0, c, u), q = fma(r, 1/c, q0) -- the correctly rounded quotient for
//     every numerator that can occur (checked for all 2^-100 < |u| < 2^100), 3 instructions instead of the 11 of an IEEE
//     division.  The fused operations are internal to the division; they replace v_div_scale/v_rcp/v_fma.../v_div_fixup,
//     which are fused as well.
//   * the clamp to [0, 255] as max/min, floor(.5 + (double)u) as floorf(u + .5f): exact for 0 <= u <= 255.
// The speculation evaluates this 8 times per lane and sample (16 in the two-utterance kernel), between barriers B and
// C where the CU is issue-bound: 37 -> 21 VALU instructions each.
__device__ __forceinline__ int dss_lin2ulaw(float x)
{
    const float scale = 255.f / 32768.f;
    const float c = 5.5451774445f, rc = 1.0f / 5.5451774445f;
    const float s128 = (x < 0) ? -128.f : 128.f;         // s * (128 * t) == (s * 128) * t: both factors are exact scalings
    x = fabsf(x);
    float u = s128 * (0.69315f * dss_log2_approx(1 + scale * x));
    const float q0 = u * rc;
    const float r = __builtin_fmaf(-q0, c, u);
    u = __builtin_fmaf(r, rc, q0);
    u = 128 + u;
    u = __builtin_fminf(__builtin_fmaxf(u, 0.f), 255.f);
    return (int)floorf(u + .5f);            // (v_cvt_rpi_i32_f32 does this in one instruction and passes the sweep, but as inline asm it measured 0.3 % slower)
}

// kiss99.c
struct DssKiss99 { uint32_t z, w, jsr, jcong; };

__device__ __forceinline__ uint32_t dss_kiss99_rand(DssKiss99 &c)
{
    const uint32_t znew = 36969u * (c.z & 0xFFFF) + (c.z >> 16);
    const uint32_t wnew = 18000u * (c.w & 0xFFFF) + (c.w >> 16);
    const uint32_t mwc = (znew << 16) + wnew;
    uint32_t shr3 = c.jsr ^ (c.jsr << 17);
    shr3 ^= shr3 >> 13;
    shr3 ^= shr3 << 5;
    const uint32_t cong = 69069u * c.jcong + 1234567u;
    c.z = znew; c.w = wnew; c.jsr = shr3; c.jcong = cong;
    return (mwc ^ cong) + shr3;
}
