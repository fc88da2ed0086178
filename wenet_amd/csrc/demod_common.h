// demod_common.h -- device helpers shared by the demodulator kernels (demod_kernel.hip: one wavefront per
// capture; demod_pipe_kernel.hip: eight wavefronts per capture, pipelined).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "glibc_atan2f.h"
#include "wenet_internal.h"

#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {   // comp_prim.h:57-65
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it would wait
// for the global loads of the NEXT frame's samples that are deliberately left in flight (and for the
// soft-decision stores of the previous frame).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ void wave_sync() {     // LDS ordering inside ONE wavefront (it runs in lockstep)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// one sample of the channel's raw input -> COMP (fsk_demod.c:273-296)
__device__ __forceinline__ float2 load_sample(const void *raw, int fmt, long long idx) {
    if (fmt == WR_FMT_CU8) {
        const uchar2 v = ((const uchar2 *)raw)[idx];
        // ((float)u8 - 127.0)/128.0 is exact in float
        return make_float2(((float)v.x - 127.0f) / 128.0f, ((float)v.y - 127.0f) / 128.0f);
    } else if (fmt == WR_FMT_CS16) {
        const short2 v = ((const short2 *)raw)[idx];
        return make_float2((float)v.x / 1000.0f, (float)v.y / 1000.0f);   // FDMDV_SCALE
    } else if (fmt == WR_FMT_S16_REAL) {
        const short v = ((const short *)raw)[idx];
        return make_float2((float)v / 1000.0f, 0.0f);
    } else {
        return ((const float2 *)raw)[idx];
    }
}

// packed-f32 forms (v_pk_mul_f32 / v_pk_add_f32): two independent IEEE operations per instruction,
// no fusion -- a lone wavefront is instruction-issue bound, so halving the instruction count matters.
typedef float v2f __attribute__((ext_vector_type(2)));
// One NCO step with the real and imaginary parts of a phasor in neighbouring lanes (even lane: re, odd lane: im):
//   re' = re*d.x - im*d.y      im' = im*d.x + re*d.y     ==  own*k1 + partner*k2,  k1 = d.x,  k2 = -d.y (re lane) / +d.y (im lane)
// -- the same two products and one sum per component as cmul_pk, each rounded on its own (x - y == x + (-y) exactly).
// Three plain VALU operations, the partner's value read through a DPP operand; s_nop 0 fills the second wait state a
// DPP read needs after a VALU write of the same register.  Half the SIMD time of the packed form, but a longer dependent path.
__device__ __forceinline__ float nco_step_split(float own, float k1, float k2) {
    float r, t1, t2;
    asm("v_mul_f32 %1, %3, %4\n\t"
        "s_nop 0\n\t"
        "v_mul_f32_dpp %2, %3, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32 %0, %1, %2"
        : "=v"(r), "=&v"(t1), "=&v"(t2)
        : "v"(own), "v"(k1), "v"(k2));
    return r;
}
// Eight NCO steps in one asm block: between separate asm statements hipcc pads a wait state (s_nop) it cannot prove unnecessary,
// one issue slot in five on the chain wave.  The only hazard inside is the one handled above (DPP read after VALU write).
__device__ __forceinline__ float nco_step_split8(float own, float k1, float k2) {
    float t1, t2;
#define WR_NCO1 "v_mul_f32 %1, %0, %3\n\ts_nop 0\n\tv_mul_f32_dpp %2, %0, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_f32 %0, %1, %2\n\t"
    asm(WR_NCO1 WR_NCO1 WR_NCO1 WR_NCO1 WR_NCO1 WR_NCO1 WR_NCO1 WR_NCO1
        : "+v"(own), "=&v"(t1), "=&v"(t2)
        : "v"(k1), "v"(k2));
#undef WR_NCO1
    return own;
}
__device__ __forceinline__ v2f cmul_pk(v2f a, v2f b) {
    // (a.x*b.x - a.y*b.y, a.x*b.y + a.y*b.x), each product and each sum rounded separately (no FMA):
    //   t1 = (a.x*b.x, a.y*b.x)   t2 = (a.y*b.y, a.x*b.y)   r = (t1.x - t2.x, t1.y + t2.y)
    // hipcc needs 5 VALU + 2 nops for this shape; written out it is 3 packed instructions, back to back.
    // (hipcc pads a wait state after packed ops whose src0 has op_sel_hi set -- its dst_sel-forwarding rule keys on a
    // modifier bit that VOP3P reuses; VALU RAW dependencies are interlocked by the hardware.  Every parity test
    // runs millions of these dependent steps, the recurrence would expose a single wrong operand.)
#ifndef WR_CHAIN_SCALAR
    v2f r, t1, t2;
    asm("v_pk_mul_f32 %1, %3, %4 op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %2, %3, %4 op_sel:[1,1] op_sel_hi:[0,1]\n\t"
        "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,0]"
        : "=v"(r), "=&v"(t1), "=&v"(t2)
        : "v"(a), "v"(b));
#else
    // scalar f32 VALU: 4 independent multiplies, then the subtract and the add (dependent depth 2).
    // Written as asm so that the SLP vectoriser does not re-pack it into the slower v_pk_* forms.
    v2f r;
    float t1, t2, t3, t4, rx, ry;
    asm("v_mul_f32 %2, %6, %8\n\t"
        "v_mul_f32 %3, %7, %9\n\t"
        "v_mul_f32 %4, %6, %9\n\t"
        "v_mul_f32 %5, %7, %8\n\t"
        "v_sub_f32 %0, %2, %3\n\t"
        "v_add_f32 %1, %4, %5"
        : "=v"(rx), "=v"(ry), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4)
        : "v"(a.x), "v"(a.y), "v"(b.x), "v"(b.y));
    r.x = rx; r.y = ry;
#endif
    return r;
}

// raw (unconverted) sample fetch + later conversion: keeps the global loads of the next frame in flight
// (a load whose first use is the int->float conversion would be waited for on the spot)
#define WR_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ uint2 load_raw(const void *raw, int fmt, long long idx) {
    uint2 r = make_uint2(0u, 0u);
    if (fmt == WR_FMT_CU8 || fmt == WR_FMT_S16_REAL) r.x = ((const WR_GLOBAL unsigned short *)(uintptr_t)raw)[idx];
    else if (fmt == WR_FMT_CS16) r.x = ((const WR_GLOBAL unsigned int *)(uintptr_t)raw)[idx];
    else { const unsigned long long v = ((const WR_GLOBAL unsigned long long *)(uintptr_t)raw)[idx]; r.x = (unsigned)v; r.y = (unsigned)(v >> 32); }
    return r;
}
// the same load performed at AGENT scope (sc1: past this XCD's L2): samples that another kernel, on other compute units, is writing beside this one
// (live ticks, WrChan::arrive) -- no stale line of the caller's own L2 / vector cache can answer it, so the reader needs no cache invalidate behind the arrival word
__device__ __forceinline__ uint2 load_raw_agent(const void *raw, int fmt, long long idx) {
    uint2 r = make_uint2(0u, 0u);
    if (fmt == WR_FMT_CU8 || fmt == WR_FMT_S16_REAL) r.x = __hip_atomic_load(&((const unsigned short *)raw)[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (fmt == WR_FMT_CS16) r.x = __hip_atomic_load(&((const unsigned int *)raw)[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else { const unsigned long long v = __hip_atomic_load(&((const unsigned long long *)raw)[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); r.x = (unsigned)v; r.y = (unsigned)(v >> 32); }
    return r;
}
// KPRE raw samples per lane (sample index base + lane + 64k, clamped to `last`): the format switch is
// hoisted so that each arm is KPRE back-to-back loads with nothing waiting on them
template <int KPRE>
__device__ __forceinline__ void prefetch_raw(uint2 (&pre)[KPRE], const void *raw, int fmt, long long base, long long last, int tid, int nt) {
    long long idx[KPRE];
#pragma unroll
    for (int k = 0; k < KPRE; k++) { const long long i = base + tid + nt * k; idx[k] = i < last ? i : last; }
    if (fmt == WR_FMT_CU8 || fmt == WR_FMT_S16_REAL) {
        const WR_GLOBAL unsigned short *p = (const WR_GLOBAL unsigned short *)(uintptr_t)raw;
#pragma unroll
        for (int k = 0; k < KPRE; k++) pre[k].x = p[idx[k]];
    } else if (fmt == WR_FMT_CS16) {
        const WR_GLOBAL unsigned int *p = (const WR_GLOBAL unsigned int *)(uintptr_t)raw;
#pragma unroll
        for (int k = 0; k < KPRE; k++) pre[k].x = p[idx[k]];
    } else {
        const WR_GLOBAL unsigned long long *p = (const WR_GLOBAL unsigned long long *)(uintptr_t)raw;
#pragma unroll
        for (int k = 0; k < KPRE; k++) { const unsigned long long v = p[idx[k]]; pre[k].x = (unsigned)v; pre[k].y = (unsigned)(v >> 32); }
    }
}
__device__ __forceinline__ float2 convert_raw(uint2 r, int fmt) {       // fsk_demod.c:273-296
    if (fmt == WR_FMT_CU8)
        return make_float2(((float)(r.x & 0xffu) - 127.0f) / 128.0f, ((float)((r.x >> 8) & 0xffu) - 127.0f) / 128.0f);
    if (fmt == WR_FMT_CS16)
        return make_float2((float)(short)(r.x & 0xffffu) / 1000.0f, (float)(short)(r.x >> 16) / 1000.0f);
    if (fmt == WR_FMT_S16_REAL) return make_float2((float)(short)(r.x & 0xffffu) / 1000.0f, 0.0f);
    return make_float2(__uint_as_float(r.x), __uint_as_float(r.y));
}

struct BestBin { float v; int i; };

__device__ __forceinline__ BestBin better(BestBin a, BestBin b) {
    // first maximum wins: strictly greater value, or equal value at a lower bin
    if (b.v > a.v || (b.v == a.v && b.i < a.i)) return b;
    return a;
}

}  // namespace
