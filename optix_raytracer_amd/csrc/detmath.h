// detmath.h -- deterministic float math shared by the HIP kernels and the host side of libmipt.
//
// The reference evaluates cosf/sinf/powf with nvcc --use_fast_math (SDK/CMakeLists.txt:267), which is
// not reproducible off NVIDIA hardware.  A 1-ulp difference in a bounce direction changes which
// triangle the next ray hits, so this library fixes the arithmetic instead: every transcendental is a
// fixed polynomial evaluated with explicit fmaf in a fixed order, and everything else is IEEE
// + - * / sqrt (hipcc's default correctly-rounded div/sqrt, -ffp-contract=off, no fast-math).
// Specification: DESIGN.md "detmath".  The CPU oracle implements the same specification separately.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define MIPT_HD __host__ __device__ __forceinline__

namespace mipt {

MIPT_HD uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
MIPT_HD float    u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

// sin and cos of x (radians, |x| up to a few pi): quadrant = rint(x*2/pi), 3-term Cody-Waite
// reduction, degree-7 / degree-8 minimax polynomials on [-pi/4, pi/4].
MIPT_HD void det_sincos(float x, float& s, float& c)
{
    const float kf = __builtin_rintf(x * 0.636619772f);
    const int   k  = (int)kf;
    float r = __builtin_fmaf(kf, -1.5703125f, x);
    r       = __builtin_fmaf(kf, -4.837512969970703125e-4f, r);
    r       = __builtin_fmaf(kf, -7.54978995489188216e-8f, r);
    const float z = r * r;
    float sp = __builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    sp       = __builtin_fmaf(sp, z, -1.6666654611e-1f);
    sp       = __builtin_fmaf(sp, z * r, r);
    float cp = __builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    cp       = __builtin_fmaf(cp, z, 4.166664568298827e-2f);
    cp       = __builtin_fmaf(cp, z * z, __builtin_fmaf(-0.5f, z, 1.0f));
    const int q = k & 3;
    const float a = (q & 1) ? cp : sp;
    const float b = (q & 1) ? sp : cp;
    s = (q & 2) ? -a : a;
    c = (q == 1 || q == 2) ? -b : b;
}

// x^k, x a positive normal float (k = 1/2.4 for sRGB encoding, 2.2 for whitted::linearize).  log2 by the atanh series on the reduced mantissa,
// exp2 by a degree-7 polynomial on [-0.5, 0.5], exponent re-inserted with integer arithmetic.
MIPT_HD float det_powk(float x, float k)
{
    const uint32_t ix = f2u(x);
    int   e = (int)(ix >> 23) - 127;
    float m = u2f((ix & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421354f) { m *= 0.5f; e += 1; }
    const float f    = m - 1.0f;
    const float s    = f / (2.0f + f);
    const float z    = s * s;
    const float w    = z * z;
    const float t1   = w * __builtin_fmaf(w, 0.24279078841f, 0.40000972152f);
    const float t2   = z * __builtin_fmaf(w, 0.28498786688f, 0.66666662693f);
    const float R    = t2 + t1;
    const float hfsq = 0.5f * f * f;
    const float lnm  = f - (hfsq - s * (hfsq + R));
    const float lg2  = __builtin_fmaf(lnm, 1.44269502f, (float)e);
    const float y    = lg2 * k;
    const float n    = __builtin_rintf(y);
    const float g    = y - n;
    float p = __builtin_fmaf(1.52527338e-5f, g, 1.54035304e-4f);
    p       = __builtin_fmaf(p, g, 1.33335581e-3f);
    p       = __builtin_fmaf(p, g, 9.61812911e-3f);
    p       = __builtin_fmaf(p, g, 5.55041087e-2f);
    p       = __builtin_fmaf(p, g, 2.40226507e-1f);
    p       = __builtin_fmaf(p, g, 6.93147182e-1f);
    p       = __builtin_fmaf(p, g, 1.0f);
    return u2f(f2u(p) + ((uint32_t)(int)n << 23));
}

MIPT_HD float det_pow_inv24(float x) { return det_powk(x, 0.416666657f); }
// powf(c, 2.2f) of whitted::linearize (cuda/whitted_cuda.h:73-80) for texture values in [0, 1]
MIPT_HD float det_pow22(float x) { return x < 1e-10f ? 0.0f : det_powk(x, 2.2f); }

// ---- integer RNG, SDK/cuda/random.h:31-67 (tea<4>, lcg, rnd) ----
MIPT_HD uint32_t tea4(uint32_t v0, uint32_t v1)
{
    uint32_t s0 = 0;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        s0 += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v0;
}

MIPT_HD float rnd(uint32_t& state)
{
    state = 1664525u * state + 1013904223u;
    return (float)(state & 0x00FFFFFFu) / 16777216.0f;
}

// ---- float3 helpers in the operation order of SDK/sutil/vec_math.h:400-570 ----
struct f3 { float x, y, z; };
MIPT_HD f3    mk3(float x, float y, float z) { return f3{ x, y, z }; }
MIPT_HD f3    operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
MIPT_HD f3    operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
MIPT_HD f3    operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
MIPT_HD f3    operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
MIPT_HD f3    operator*(float s, f3 a) { return mk3(a.x * s, a.y * s, a.z * s); }
MIPT_HD f3    operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
MIPT_HD float dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
MIPT_HD f3    cross(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
MIPT_HD float length(f3 v) { return __builtin_sqrtf(dot(v, v)); }
MIPT_HD f3    normalize(f3 v) { const float inv = 1.0f / __builtin_sqrtf(dot(v, v)); return v * inv; }
MIPT_HD f3    faceforward(f3 n, f3 i, f3 nref) { return n * __builtin_copysignf(1.0f, dot(i, nref)); }
MIPT_HD f3    lerp3(f3 a, f3 b, float t) { return a + t * (b - a); }
MIPT_HD float clampf(float f, float a, float b) { return __builtin_fmaxf(a, __builtin_fminf(f, b)); }

// ---- sRGB + 8-bit quantisation, SDK/cuda/helpers.h:35-66 ----
MIPT_HD float srgb_channel(float c)
{
    return c < 0.0031308f ? 12.92f * c : 1.055f * det_pow_inv24(c) - 0.055f;
}
MIPT_HD uint32_t quantize8(float x)
{
    x = clampf(x, 0.0f, 1.0f);
    const uint32_t q = (uint32_t)(x * 256.0f);
    return q < 255u ? q : 255u;
}
// returns uchar4 packed little-endian: r | g<<8 | b<<16 | 255<<24
MIPT_HD uint32_t make_color(f3 c)
{
    const uint32_t r = quantize8(srgb_channel(clampf(c.x, 0.0f, 1.0f)));
    const uint32_t g = quantize8(srgb_channel(clampf(c.y, 0.0f, 1.0f)));
    const uint32_t b = quantize8(srgb_channel(clampf(c.z, 0.0f, 1.0f)));
    return r | (g << 8) | (b << 16) | 0xff000000u;
}

// ---- image-tile sharding, SDK/sutil/WorkDistribution.h:50-81 (8 x 4 pixel tiles) ----
MIPT_HD uint32_t shard_num_samples(uint32_t width, uint32_t height, uint32_t world)
{
    const uint32_t strip = 8u * world;
    return ((width + strip - 1u) / strip) * ((height + 3u) / 4u) * 32u;
}
MIPT_HD void shard_pixel(uint32_t width, uint32_t world, uint32_t rank, uint32_t sample, int& px, int& py)
{
    const uint32_t strip  = 8u * world;
    const uint32_t cols   = (width + strip - 1u) / strip;
    const uint32_t tile   = sample >> 5;
    const uint32_t within = sample & 31u;
    const uint32_t srow   = tile / cols;
    const uint32_t scol   = tile - srow * cols;
    const uint32_t rot    = (rank + srow % world) % world;
    px = (int)(scol * strip + rot * 8u + (within & 7u));
    py = (int)(srow * 4u + (within >> 3));
}

}  // namespace mipt
