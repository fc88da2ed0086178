// x87emu.h -- integer emulation of the three x87 80-bit ("long double") sub-expressions of
// the reference's sd_to_llr (src/mpdecode_core.c:584,592,594), for gfx950 which has no 80-bit
// floating point.  On x86-64 gcc evaluates
//
//     estEsN0 = 1.0/(2.0L * estvar + 1E-3);        // add + divide in 64-bit-significand precision,
//                                                  // result rounded AGAIN to double on assignment
//     llr[i]  = 4.0L * estEsN0 * sd[i];            // product rounded to 64 bits, then to float
//
// with a 64-bit significand and round-to-nearest-even at every step.  The functions below
// reproduce those roundings bit for bit (tests/test_host_numerics.py compares them with native
// long double on x86-64 over millions of random and adversarial operands).
//
// Usable from host and device code.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define WX_HD __host__ __device__ __forceinline__
#else
#define WX_HD static inline
#endif

struct wx_x87 {        // finite non-zero: value = (-1)^sign * mant * 2^(exp-63), mant has bit 63 set
    int sign;
    int exp;
    uint64_t mant;     // 0 => zero
};

WX_HD uint64_t wx_d2u(double d) { union { double d; uint64_t u; } c; c.d = d; return c.u; }
WX_HD double   wx_u2d(uint64_t u) { union { double d; uint64_t u; } c; c.u = u; return c.d; }
WX_HD uint32_t wx_f2u(float f) { union { float f; uint32_t u; } c; c.f = f; return c.u; }
WX_HD float    wx_u2f(uint32_t u) { union { float f; uint32_t u; } c; c.u = u; return c.f; }

WX_HD int wx_clz64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __clzll((long long)x);
#else
    return __builtin_clzll(x);
#endif
}

WX_HD void wx_mul64(uint64_t a, uint64_t b, uint64_t *hi, uint64_t *lo) {
#if defined(__HIP_DEVICE_COMPILE__)
    *lo = a * b;
    *hi = __umul64hi(a, b);
#else
    unsigned __int128 p = (unsigned __int128)a * b;
    *lo = (uint64_t)p;
    *hi = (uint64_t)(p >> 64);
#endif
}

WX_HD bool wx_finite(double d) { return ((wx_d2u(d) >> 52) & 0x7ff) != 0x7ff; }

// exact conversion of a FINITE double
WX_HD wx_x87 wx_from_double(double d) {
    wx_x87 r;
    uint64_t u = wx_d2u(d);
    int be = (int)((u >> 52) & 0x7ff);
    uint64_t frac = u & 0xfffffffffffffULL;
    r.sign = (int)(u >> 63);
    if (be == 0) {
        if (frac == 0) { r.exp = 0; r.mant = 0; return r; }
        int sh = wx_clz64(frac);               // subnormal: normalise
        r.mant = frac << sh;
        r.exp = -1022 - (sh - 11);
        return r;
    }
    r.mant = (frac | (1ULL << 52)) << 11;
    r.exp = be - 1023;
    return r;
}

// round-to-nearest-even of the 128-bit magnitude hi:lo (hi has bit 63 set) to 64 bits.
// Bits lost earlier are "jammed" into lo's LSB by the callers, so lo==2^63 is an exact tie.
WX_HD void wx_round64(wx_x87 *r, uint64_t hi, uint64_t lo) {
    const uint64_t half = 1ULL << 63;
    int up = (lo > half) || (lo == half && (hi & 1));
    hi += (uint64_t)up;
    if (up && hi == 0) { hi = half; r->exp += 1; }
    r->mant = hi;
}

// a + b with 64-bit significand, RNE (both finite)
WX_HD wx_x87 wx_add(wx_x87 a, wx_x87 b) {
    if (a.mant == 0) return b;
    if (b.mant == 0) return a;
    // make |a| >= |b|
    if (b.exp > a.exp || (b.exp == a.exp && b.mant > a.mant)) { wx_x87 t = a; a = b; b = t; }
    int d = a.exp - b.exp;
    uint64_t bh, bl;
    // b aligned under a as a 128-bit quantity; anything shifted out is jammed into bit 0
    if (d == 0) { bh = b.mant; bl = 0; }
    else if (d < 64) { bh = b.mant >> d; bl = b.mant << (64 - d); }
    else if (d == 64) { bh = 0; bl = b.mant; }
    else if (d < 128) { bh = 0; bl = (b.mant >> (d - 64)) | (uint64_t)((b.mant << (128 - d)) != 0); }
    else { bh = 0; bl = 1; }
    wx_x87 r; r.sign = a.sign; r.exp = a.exp;
    uint64_t hi, lo;
    if (a.sign == b.sign) {
        lo = bl; hi = a.mant + bh;
        if (hi < a.mant) {                       // carry out of bit 63: shift right one, keep sticky
            lo = (lo >> 1) | (lo & 1) | (hi << 63);
            hi = (hi >> 1) | (1ULL << 63);
            r.exp += 1;
        }
    } else {
        lo = 0 - bl;
        hi = a.mant - bh - (uint64_t)(bl != 0 ? 1 : 0);
        if (hi == 0 && lo == 0) { r.mant = 0; r.exp = 0; r.sign = 0; return r; }   // x - x = +0 (RNE)
        if (hi == 0) { hi = lo; lo = 0; r.exp -= 64; }
        int sh = wx_clz64(hi);
        if (sh) { hi = (hi << sh) | (lo >> (64 - sh)); lo <<= sh; r.exp -= sh; }
    }
    wx_round64(&r, hi, lo);
    return r;
}

// 1.0 / t with 64-bit significand, RNE (t finite, non-zero)
WX_HD wx_x87 wx_recip(wx_x87 t) {
    wx_x87 r; r.sign = t.sign;
    if (t.mant == (1ULL << 63)) { r.mant = t.mant; r.exp = -t.exp; return r; }   // power of two: exact
    // quotient of 2^127 / mant, mant in (2^63, 2^64): 64 significant bits, value in (2^63, 2^64)
    uint64_t rem = 1ULL << 63, q = 0;
    // rem < mant always holds before each step; one step = shift left, conditional subtract
    for (int i = 0; i < 64; i++) {
        int top = (int)(rem >> 63);
        rem <<= 1;
        q <<= 1;
        if (top || rem >= t.mant) { rem -= t.mant; q |= 1; }
    }
    r.exp = -t.exp - 1;
    // rounding: compare 2*rem with mant
    int top = (int)(rem >> 63);
    uint64_t twice = rem << 1;
    int gt = top || (twice > t.mant);
    int eq = (!top) && (twice == t.mant);
    int up = gt || (eq && (q & 1));
    q += (uint64_t)up;
    if (up && q == 0) { q = 1ULL << 63; r.exp += 1; }
    r.mant = q;
    return r;
}

// a * b with 64-bit significand, RNE
WX_HD wx_x87 wx_mul(wx_x87 a, wx_x87 b) {
    wx_x87 r; r.sign = a.sign ^ b.sign;
    if (a.mant == 0 || b.mant == 0) { r.mant = 0; r.exp = 0; return r; }
    uint64_t hi, lo;
    wx_mul64(a.mant, b.mant, &hi, &lo);
    r.exp = a.exp + b.exp + 1;
    if (!(hi >> 63)) { hi = (hi << 1) | (lo >> 63); lo <<= 1; r.exp -= 1; }
    wx_round64(&r, hi, lo);
    return r;
}

// round a 64-bit-significand value to double (RNE), including subnormals/overflow
WX_HD double wx_to_double(wx_x87 a) {
    uint64_t s = (uint64_t)a.sign << 63;
    if (a.mant == 0) return wx_u2d(s);
    int e = a.exp;
    if (e > 1023) return wx_u2d(s | 0x7ff0000000000000ULL);
    int drop = 11;                        // keep 53 bits
    if (e < -1022) drop += (-1022 - e);   // subnormal: fewer significant bits
    if (drop > 64) return wx_u2d(s);
    uint64_t keep, rest, half;
    if (drop == 64) { keep = 0; rest = a.mant; half = 1ULL << 63; }
    else { keep = a.mant >> drop; rest = a.mant & ((1ULL << drop) - 1); half = 1ULL << (drop - 1); }
    if (rest > half || (rest == half && (keep & 1))) keep++;
    if (e < -1022) return wx_u2d(s | keep);                   // keep==2^52 lands on the smallest normal
    if (keep >> 53) { keep >>= 1; e++; if (e > 1023) return wx_u2d(s | 0x7ff0000000000000ULL); }
    return wx_u2d(s | ((uint64_t)(e + 1023) << 52) | (keep & 0xfffffffffffffULL));
}

// round a 64-bit-significand value to float (RNE), including subnormals/overflow
WX_HD float wx_to_float(wx_x87 a) {
    uint32_t s = (uint32_t)a.sign << 31;
    if (a.mant == 0) return wx_u2f(s);
    int e = a.exp;
    if (e > 127) return wx_u2f(s | 0x7f800000u);
    int drop = 40;                       // keep 24 bits
    if (e < -126) drop += (-126 - e);    // subnormal: fewer significant bits
    if (drop > 64) return wx_u2f(s);     // below half of the smallest subnormal
    uint64_t keep, rest, half;
    if (drop == 64) { keep = 0; rest = a.mant; half = 1ULL << 63; }
    else { keep = a.mant >> drop; rest = a.mant & ((1ULL << drop) - 1); half = 1ULL << (drop - 1); }
    if (rest > half || (rest == half && (keep & 1))) keep++;
    if (e < -126) return wx_u2f(s | (uint32_t)keep);          // keep==2^23 lands on the smallest normal: correct
    if (keep >> 24) { keep >>= 1; e++; if (e > 127) return wx_u2f(s | 0x7f800000u); }
    return wx_u2f(s | ((uint32_t)(e + 127) << 23) | ((uint32_t)keep & 0x7fffffu));
}

// estEsN0 = 1.0/(2.0L*estvar + 1E-3)   (mpdecode_core.c:592), result as the double the reference stores
WX_HD double wx_est_esn0(double estvar) {
    if (!wx_finite(estvar)) return 1.0 / (2.0 * estvar + 1E-3);           // inf/nan: same in any precision
    wx_x87 v = wx_from_double(estvar);
    if (v.mant) v.exp += 1;                                               // 2.0L * estvar, exact
    wx_x87 t = wx_add(v, wx_from_double(1E-3));
    if (t.mant == 0) return 1.0 / 0.0;
    return wx_to_double(wx_recip(t));
}

// llr = (float)(4.0L * estEsN0 * sd)    (mpdecode_core.c:594)
WX_HD float wx_llr(double estEsN0, double sd) {
    if (!wx_finite(estEsN0) || !wx_finite(sd)) return (float)(4.0 * estEsN0 * sd);
    {
        // Fast path.  The x87 result is rnd24(rnd64(e)), e = the exact product.  hi = rnd53(e) (one double multiply; 4.0*estEsN0
        // is exact) rounds to the same float unless hi sits EXACTLY half-way between two floats: the half-way points are
        // doubles, rounding is monotone, so e, rnd64(e) and hi lie on the same side of every half-way point that hi does not
        // hit.  Outside the normal float range, and on a half-way point (2^-29 of the cases), the integer emulation decides.
        const double c4 = 4.0 * estEsN0;
        const double hi = c4 * sd;
        const uint64_t u = wx_d2u(hi);
        const int be = (int)((u >> 52) & 0x7ff);
        if (wx_finite(c4) && be >= 1023 - 126 && be <= 1023 + 126 && (u & 0x1fffffffULL) != 0x10000000ULL) return (float)hi;
    }
    wx_x87 c = wx_from_double(estEsN0);
    if (c.mant) c.exp += 2;                                               // 4.0L * estEsN0, exact
    return wx_to_float(wx_mul(c, wx_from_double(sd)));
}
