// ldpc_decode2.hip -- round 4: the (2580,2064) sum-product decoder with ONE-BYTE message cells.
//
// Reference map (file:line in /root/reference/src): SumProduct mpdecode_core.c:385-489 (check pass :414-436, variable pass :439-464, stop rules :466-483),
// phi0 phi0.c:13-218, sd_to_llr's product mpdecode_core.c:593-594, packing drs232_ldpc.c:234-239.
//
// Every message magnitude the algorithm ever stores is an OUTPUT of phi0, and phi0 has 103 outputs (a step function, phi0.c).  A message is therefore
// kept as one byte -- sign << 7 | index of the value in the list of phi0's outputs -- and turned back into its float by a 256-entry table (the
// sign applied there).  The float sums run exactly as before, in the reference's order: same bits, same iteration counts.  What the byte buys:
//   * the 14 messages of a check are 16 consecutive bytes: the check pass reads them with ONE ds_read_b128 and writes them with one ds_write_b128
//     (round 3: 14 + 14 LDS instructions); the parity of the signs and the new signs are formed on the packed words, four edges at a time;
//   * phi0 never looks its value up: the cell of the argument gives {threshold, index below it}; index = that + (argument >= threshold);
//   * LDS per packet falls from 28.9 KB to 8.3 KB.  The freed space holds the packet's LLRs (six registers per thread in round 3) and a SECOND buffer
//     into which the next packet's soft symbols land by LDS-DMA (global_load_lds_dword: no register, no wait) while this packet iterates -- round 3
//     paid ~10 k cycles per packet for the chain slot counter -> channel table -> deframer state -> start offset -> symbols, and another ~8 k
//     for packing the bits through a byte array; here the next slot's counter, its record and its symbols are fetched one, two and three barriers
//     ahead, and the decisions are OR-ed into 81 words that 65 threads store.
// One 512-thread workgroup per packet, four per CU, persistent, packets from a shared counter -- as in round 3.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "wenet_internal.h"
#include "x87emu.h"

#pragma clang fp contract(off)

namespace {

typedef __attribute__((address_space(3))) float lds_f32;
typedef const __attribute__((address_space(1))) float glb_cf32;

__device__ __forceinline__ long long uni64(long long v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}
// the whole of wx_llr (x87emu.h), integer emulation of the 80-bit product included, as a real call: needed for one symbol in 2^29
__device__ __attribute__((noinline)) float llr_exact2(double estEsN0, double sd) { return wx_llr(estEsN0, sd); }

// index of phi0(|x|) in the value list: the cell of |x|'s leading bits holds {threshold, index below it}.  `cellb` = the cell table's base minus the key
// bias (the LDS layout puts the table where that is a non-negative offset: it rides in the instruction's offset field).
__device__ __forceinline__ int2 phi0_cell(float x, const int2 *cellb) {
    const unsigned key = min(max((__float_as_uint(x) >> 18) & 0x1fffu, (unsigned)WR_PHI0_KEY_BIAS), (unsigned)(WR_PHI0_KEY_BIAS + WR_PHI0_LUT_ENTRIES - 1));   // v_bfe, v_med3
    return cellb[key];
}
__device__ __forceinline__ unsigned phi0_pick(float x, int2 c) {
    return (unsigned)c.y + (!(fabsf(x) < __int_as_float(c.x)) ? 1u : 0u);   // not-less-than: true for NaN (its cell is the last one: 10.0, as x86's cvttss2si overflow gives)
}
// byte k (0..3) of w, times four: the byte's entry in the value table -- one SDWA shift
template <int K>
__device__ __forceinline__ unsigned byte_x4(unsigned w) {
    unsigned r;
    if constexpr (K == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(2u), "v"(w));
    else if constexpr (K == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(2u), "v"(w));
    else if constexpr (K == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(2u), "v"(w));
    else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(2u), "v"(w));
    return r;
}
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (B < E) { f(std::integral_constant<int, B>{}); static_for<B + 1, E>(f); }
}
__device__ __forceinline__ unsigned word_of(const uint4 &w, int j) { return j == 0 ? w.x : (j == 1 ? w.y : (j == 2 ? w.z : w.w)); }

}  // namespace

#ifndef WR_D2_WAVES_PER_EU
#define WR_D2_WAVES_PER_EU 8            // 8: four workgroups per CU, 64 VGPRs; 6: three, 80 VGPRs
#endif
__global__ __launch_bounds__(WR_DEC_THREADS, WR_D2_WAVES_PER_EU) void wenet_decode2_kernel(WrDecodeArgs A) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint8_t  *MSG = smem + WR_D2_OFF_MSG;                               // [516][16] message bytes: check c's slots at 16 c + k
    int2     *CELL = (int2 *)(smem + WR_D2_OFF_CELL);
    const int2 *CELLB = (const int2 *)(smem + (WR_D2_OFF_CELL - 8 * WR_PHI0_KEY_BIAS));     // (table base minus the key bias: phi0_cell)
    const unsigned char *VTB = smem + WR_D2_OFF_VT;                       // the value table, addressed in bytes (byte_x4)
    float    *VT = (float *)(smem + WR_D2_OFF_VT);
    float    *LLB = (float *)(smem + WR_D2_OFF_LLB);                    // [2][6][512]: soft symbols as they land, LLRs after the prologue; position t of thread tid at [t][tid]
    unsigned *WORDS = (unsigned *)(smem + WR_D2_OFF_WORDS);             // the decided bits, MSB-first bytes in little-endian words = the packet's bytes
    int      *red = (int *)(smem + WR_D2_OFF_RED);
    int      *claim = (int *)(smem + WR_D2_OFF_CLAIM);                  // [2] slot decoded now / next (-1: none left)
    int      *REC = (int *)(smem + WR_D2_OFF_REC);                      // [2][4] their records {address lo, hi, estEsN0 lo, hi}
    constexpr int LLBW = WR_VARS_PER_THREAD * WR_DEC_THREADS;            // floats per buffer

    // ---- once per workgroup: tables into LDS; this thread's variables: edge addresses, where their symbols sit in a stored packet, scrambler signs ----
    for (int i = tid; i < WR_PHI0_LUT_ENTRIES; i += WR_DEC_THREADS) CELL[i] = A.d2_cells[i];
    for (int i = tid; i < 256; i += WR_DEC_THREADS) VT[i] = A.d2_vt[i];
    int ea[WR_VARS_PER_THREAD][3];
    unsigned soff[3] = {0u, 0u, 0u}, svar[3] = {0u, 0u, 0u}, sneg = 0u, svalid = 0u;
    int deg4 = 0, deg5 = 0;
    {
        const bool stream = A.input_kind == WR_DEC_IN_STREAM;
#pragma unroll
        for (int t = 0; t < WR_VARS_PER_THREAD; t++) {
            const int p = tid + t * WR_DEC_THREADS;
            const int v = p < WR_NCODE ? (int)A.vpos2[p] : WR_NCODE;
            int d = 0;
            ea[t][0] = ea[t][1] = ea[t][2] = 0;
            unsigned o = 0u;
            if (v < WR_NDATA) {
                d = 3;
#pragma unroll
                for (int k = 0; k < 3; k++) ea[t][k] = A.vedge2[v * 3 + k];
            } else if (v < WR_NCODE) {                                   // parity bit of check c = v - 2064: last slot of check c, slot 12 of check c + 1 (mpdecode_core.c:226-234,296-303)
                const int c = v - WR_NDATA;
                d = (v == WR_NCODE - 1) ? 1 : 2;
                ea[t][0] = c * 16 + (c == 0 ? 12 : 13);
                ea[t][1] = (c + 1) * 16 + 12;
            }
            if (v < WR_NCODE) {
                svalid |= 1u << t;
                o = (unsigned)v;
                if (stream && A.mode == 1) o = (unsigned)(10 * (v >> 3) + 8 - (v & 7));               // RS232 strip: out[8b+j] = in[10b + 8 - j] (drs232_ldpc.c:220-225)
                if (stream && A.mode == 2) {                                                          // v2: symbol * scramble_code[ind % 1000] (wenet_ldpc.c:207)
                    const int kb = v % 1000;
                    if ((A.scramble[kb >> 3] >> (7 - (kb & 7))) & 1) sneg |= 1u << t;
                }
            }
            soff[t >> 1] |= o << (16 * (t & 1));
            svar[t >> 1] |= (unsigned)(v < WR_NCODE ? v : 0) << (16 * (t & 1));
            if (t == 4) deg4 = d;
            if (t == 5) deg5 = d;
        }
    }
    for (int c = tid; c < WR_NPAR; c += WR_DEC_THREADS) { MSG[c * 16 + 14] = 0; MSG[c * 16 + 15] = 0; }     // bytes 14, 15 of a check's 16 stay zero for good: they take part in the word-wise sign parity
    const bool data4 = deg4 == 3;
    const long long nslots = (long long)A.nchan * A.max_pk;

    // (thread 0, synchronous) the record of slot s into cell c
    auto fetch_record = [&](unsigned s, int c) __attribute__((always_inline)) {
        unsigned long long b = 0ull, e = 0ull;
        if ((long long)s < nslots) {
            if (A.input_kind == WR_DEC_IN_LLR) {                        // dense LLR input: no statistics kernel has run
                const int chs = (int)(s / (unsigned)A.max_pk), pks = (int)(s - (unsigned)chs * (unsigned)A.max_pk);
                if (pks < A.npk_direct[chs]) b = (unsigned long long)(uintptr_t)(A.llr_in + (long long)s * WR_NCODE);
            } else { const WrSlotRec r = A.rec[s]; b = r.base; e = (unsigned long long)__double_as_longlong(r.esn0); }
        }
        claim[c] = (long long)s < nslots ? (int)s : -1;
        REC[c * 4 + 0] = (int)(unsigned)b; REC[c * 4 + 1] = (int)(unsigned)(b >> 32); REC[c * 4 + 2] = (int)(unsigned)e; REC[c * 4 + 3] = (int)(unsigned)(e >> 32);
    };
    // this thread's soft symbols of the packet at `base` -> buffer `buf`, by LDS-DMA: lane l of a wave's instruction t writes LLB[buf][t][64 wave + l]
    auto dma_symbols = [&](unsigned long long base, int buf) __attribute__((always_inline)) {
        glb_cf32 *sdp = (glb_cf32 *)base;
#pragma unroll
        for (int t = 0; t < WR_VARS_PER_THREAD; t++) {
            if ((svalid >> t) & 1u) {
                lds_f32 *dst = (lds_f32 *)(LLB + buf * LLBW + t * WR_DEC_THREADS + wave * 64);
                __builtin_amdgcn_global_load_lds(sdp + ((soff[t >> 1] >> (16 * (t & 1))) & 0xffffu), dst, 4, 0, 0);
            }
        }
    };

    if (tid == 0) fetch_record(atomicAdd(A.work, 1u), 0);                // the first packet: everything synchronously
    __syncthreads();
    int cur = 0;
    bool staged = false;                                                 // this packet's symbols are in (or on their way to) LLB[cur]

    for (;; cur ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this wave's DMA-ed symbols (and thread 0's record) have landed
        __syncthreads();
        const int slot_i = __builtin_amdgcn_readfirstlane(claim[cur]);
        if (slot_i < 0) break;
        const long long slot = slot_i;
        const unsigned long long base = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(REC[cur * 4 + 1]) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(REC[cur * 4 + 0]);
        const double estEsN0 = __longlong_as_double((long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(REC[cur * 4 + 3]) << 32) |
                                                                (unsigned)__builtin_amdgcn_readfirstlane(REC[cur * 4 + 2])));
        unsigned nxt = 0;
        if (tid == 0) nxt = atomicAdd(A.work, 1u);                        // the next packet's slot: the value is not waited for here
        int ahead = 0;                                                   // 0: counter read in flight; 1: claim + record in LDS (or landing); 2: symbols on their way
        if (base == 0ull) {                                              // nothing in this slot: move on (everything synchronously)
            if (tid == 0) fetch_record(nxt, cur ^ 1);
            staged = false;
            continue;
        }
        if (!staged) {                                                   // (the first packet, or the one after a short packet: fetch now)
            dma_symbols(base, cur);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        staged = false;

        // ---- LLRs in place: llr = (float)(4.0L * estEsN0 * sd), x87 rounding (mpdecode_core.c:593-594, x87emu.h); dense LLR input stays as it is ----
        float *LL = LLB + cur * LLBW;
        if (A.input_kind != WR_DEC_IN_LLR) {
            const double c4 = 4.0 * estEsN0;
            const bool c4ok = wx_finite(c4) && wx_finite(estEsN0);
            unsigned hard = 0u;
#pragma unroll
            for (int t = 0; t < WR_VARS_PER_THREAD; t++) {
                if (!((svalid >> t) & 1u)) continue;
                const float raw = LL[t * WR_DEC_THREADS + tid];
                const double sd = ((sneg >> t) & 1u) ? -(double)raw : (double)raw;
                const double hi = c4 * sd;                               // wx_llr's fast path: one double product decides the float unless it sits exactly half-way
                const unsigned long long u = wx_d2u(hi);                 // between two floats or outside the normal float range
                const int be = (int)((u >> 52) & 0x7ff);
                float l = (float)hi;
                if (!(c4ok && be >= 1023 - 126 && be <= 1023 + 126 && (u & 0x1fffffffULL) != 0x10000000ULL)) hard |= 1u << t;
                LL[t * WR_DEC_THREADS + tid] = l;
                if (A.llr_out && !((hard >> t) & 1u)) A.llr_out[slot * WR_NCODE + ((svar[t >> 1] >> (16 * (t & 1))) & 0xffffu)] = l;
            }
            if (hard) {                                                  // (rare: the integer emulation, from the symbol itself)
                glb_cf32 *sdp = (glb_cf32 *)base;
                for (int t = 0; t < WR_VARS_PER_THREAD; t++) {
                    if (!((hard >> t) & 1u)) continue;
                    const float raw = sdp[(soff[t >> 1] >> (16 * (t & 1))) & 0xffffu];
                    const float l = llr_exact2(estEsN0, ((sneg >> t) & 1u) ? -(double)raw : (double)raw);
                    LL[t * WR_DEC_THREADS + tid] = l;
                    if (A.llr_out) A.llr_out[slot * WR_NCODE + ((svar[t >> 1] >> (16 * (t & 1))) & 0xffffu)] = l;
                }
            }
        }
        // ---- initial variable -> check messages: phi0(|llr|), sign = llr < 0 (mpdecode_core.c:353-359); check 0's phantom 14th edge: +0.0 ----
#pragma unroll
        for (int t = 0; t < WR_VARS_PER_THREAD; t++) {
            const int d = t < WR_VARS_ALLDATA ? 3 : (t == 4 ? deg4 : deg5);
            if (d > 0) {
                const float l = LL[t * WR_DEC_THREADS + tid];
                const unsigned m0 = phi0_pick(l, phi0_cell(l, CELLB)) | (l < 0.f ? 0x80u : 0u);
#pragma unroll
                for (int k = 0; k < 3; k++) if (k < d) MSG[ea[t][k]] = (uint8_t)m0;
            }
        }
        if (tid == 0) MSG[13] = (uint8_t)(WR_PHI0_NVALS - 2);            // (value 102 = 0.0)
        if (tid < 4) red[tid] = 0;
        if (tid < 84) WORDS[tid] = 0u;
        __syncthreads();

        int result = A.max_iter, pcc = 0, pcc_written = 0;
        unsigned bits = 0;
        for (int iter = 0; iter < A.max_iter; iter++) {
            const int par = iter & 1;
            // ---- update r: thread = check (mpdecode_core.c:414-436) ----
            int ok = 0;
            {
                const uint4 w = *(const uint4 *)(MSG + tid * 16);
                float m[14];
                static_for<0, 14>([&](auto kc) __attribute__((always_inline)) { constexpr int k = kc; m[k] = *(const float *)(VTB + byte_x4<(k & 3)>(word_of(w, k >> 2))); });
                float phi_sum = fabsf(m[0]);
#pragma unroll
                for (int k = 1; k < 14; k++) phi_sum = phi_sum + fabsf(m[k]);
                unsigned x = w.x ^ w.y ^ w.z ^ w.w;                      // parity of the signs: bit 7 of the XOR of all bytes
                x ^= x >> 16; x ^= x >> 8;
                ok += (x & 0x80u) ? 0 : 1;
                const unsigned pm = __builtin_amdgcn_perm(x, x, 0u);     // byte 0 in all four bytes (only its bit 7 is used below)
                // fourteen evaluations, seven at a time: the differences, then their cells (independent LDS reads, in flight together), then the picks
                unsigned o[4] = {0u, 0u, 0u, 0u};
                static_for<0, 2>([&](auto hc) __attribute__((always_inline)) {
                    constexpr int k0 = 7 * hc;
                    float a[7];
                    int2 c[7];
#pragma unroll
                    for (int k = 0; k < 7; k++) { a[k] = phi_sum - fabsf(m[k0 + k]); asm("" : "+v"(a[k])); }     // (kept scalar: packed in pairs, the subtractions need their |m| formed first)
#pragma unroll
                    for (int k = 0; k < 7; k++) c[k] = phi0_cell(a[k], CELLB);
#pragma unroll
                    for (int k = 0; k < 7; k++) o[(k0 + k) >> 2] |= phi0_pick(a[k], c[k]) << (8 * ((k0 + k) & 3));
                });
                o[0] |= (w.x ^ pm) & 0x80808080u; o[1] |= (w.y ^ pm) & 0x80808080u; o[2] |= (w.z ^ pm) & 0x80808080u; o[3] |= (w.w ^ pm) & 0x00008080u;
                if (tid == 0) o[3] = (o[3] & 0xffff00ffu) | ((unsigned)(WR_PHI0_NVALS - 2) << 8);       // check 0 has 13 edges: its 14th slot stays a neutral +0.0
                *(uint4 *)(MSG + tid * 16) = make_uint4(o[0], o[1], o[2], o[3]);
            }
            // checks 512..515 edge-parallel on 56 lanes of the last wavefront: every lane of a check's group sums the 14 magnitudes itself, in order, and
            // updates only its own edge
            if (tid >= WR_DEC_THREADS - 64 && tid < WR_DEC_THREADS - 64 + (WR_NPAR - WR_DEC_THREADS) * 14) {
                const int l = tid - (WR_DEC_THREADS - 64), g = l / 14, k = l - g * 14, chk = WR_DEC_THREADS + g;
                const uint4 w = *(const uint4 *)(MSG + chk * 16);
                float phi_sum = 0.f, mine = 0.f;
#pragma unroll
                for (int kk = 0; kk < 14; kk++) {
                    const float mv = VT[(word_of(w, kk >> 2) >> (8 * (kk & 3))) & 0xffu];
                    phi_sum = (kk == 0) ? fabsf(mv) : phi_sum + fabsf(mv);
                    if (kk == k) mine = mv;
                }
                unsigned x = w.x ^ w.y ^ w.z ^ w.w;
                x ^= x >> 16; x ^= x >> 8;
                const unsigned pbit = x & 0x80u;
                if (k == 0) ok += pbit ? 0 : 1;
                const float am = phi_sum - fabsf(mine);
                const unsigned idx = phi0_pick(am, phi0_cell(am, CELLB));
                const unsigned sgn = ((__float_as_uint(mine) >> 24) ^ pbit) & 0x80u;
                // (the group's lanes run in lockstep and the byte below depends on the 16 read above: every lane has read them before any is rewritten)
                MSG[chk * 16 + k] = (uint8_t)(idx | sgn);
            }
            {
                const unsigned long long bal = __ballot(ok & 1), bal2 = __ballot(ok & 2);
                if (lane == 0 && (bal | bal2)) atomicAdd(&red[par * 2 + 0], __popcll(bal) + 2 * __popcll(bal2));
            }
            // one barrier ahead: the next slot's counter value has long returned -> its claim; its record on its way into LDS (thread 0)
            if (ahead == 0) {
                if (tid == 0) {
                    claim[cur ^ 1] = (long long)nxt < nslots ? (int)nxt : -1;
                    if (A.input_kind == WR_DEC_IN_LLR || (long long)nxt >= nslots) fetch_record(nxt, cur ^ 1);
                    else __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) unsigned *)(A.rec + nxt), (__attribute__((address_space(3))) unsigned *)(REC + (cur ^ 1) * 4), 16, 0, 0);
                }
                ahead = 1;
            } else if (ahead == 1) {
                // two barriers ahead: the record is in LDS (thread 0 waited for it before the last barrier) -> every thread sends for its symbols of the next packet
                const int ns = __builtin_amdgcn_readfirstlane(claim[cur ^ 1]);
                const unsigned long long nb = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(REC[(cur ^ 1) * 4 + 1]) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(REC[(cur ^ 1) * 4 + 0]);
                if (ns >= 0 && nb != 0ull) { dma_symbols(nb, cur ^ 1); staged = true; }
                ahead = 2;
            }
            __syncthreads();
            const int ssum = red[par * 2 + 0];
            if (tid == 0) { red[(par ^ 1) * 2 + 0] = 0; red[(par ^ 1) * 2 + 1] = 0; }
            // ---- update q: thread = variable (mpdecode_core.c:439-464) ----
            int any_data = 0;
            bits = 0;
            // Two variables at a time: their six message bytes, the six values, the two sums; then the six evaluations' cells (independent reads, in flight
            // together) and picks; the six stores last -- a byte store between two loads would hold the second load back (bytes may alias anything).
            static_for<0, WR_VARS_PER_THREAD / 2>([&](auto pc) __attribute__((always_inline)) {
                constexpr int t0 = 2 * pc;
                const int d0 = t0 < WR_VARS_ALLDATA ? 3 : (t0 == 4 ? deg4 : deg5), d1 = t0 + 1 < WR_VARS_ALLDATA ? 3 : (t0 + 1 == 4 ? deg4 : deg5);
                float cm[2][3], Qi[2];
                unsigned nb[2][3];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int d = u ? d1 : d0;
                    Qi[u] = LL[(t0 + u) * WR_DEC_THREADS + tid];
#pragma unroll
                    for (int k = 0; k < 3; k++) cm[u][k] = (k < d) ? VT[MSG[ea[t0 + u][k]]] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int d = u ? d1 : d0;
#pragma unroll
                    for (int k = 0; k < 3; k++) if (k < d) Qi[u] += cm[u][k];
                    const int b = (d > 0) && (Qi[u] < 0.f);
                    bits |= (unsigned)b << (t0 + u);
                    if (b && (t0 + u < WR_VARS_ALLDATA || (t0 + u == 4 && data4))) any_data = 1;
                    float ts[3];
                    int2 cc[3];
#pragma unroll
                    for (int k = 0; k < 3; k++) { ts[k] = Qi[u] - cm[u][k]; cc[k] = phi0_cell(ts[k], CELLB); }
#pragma unroll
                    for (int k = 0; k < 3; k++) nb[u][k] = phi0_pick(ts[k], cc[k]) | (!(ts[k] > 0.f) ? 0x80u : 0u);
                }
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int d = u ? d1 : d0;
#pragma unroll
                    for (int k = 0; k < 3; k++) if (k < d) MSG[ea[t0 + u][k]] = (uint8_t)nb[u][k];
                }
            });
            if (__ballot(any_data) && lane == 0) red[par * 2 + 1] = 1;
            if (ahead == 1 && tid == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the record has landed (it was sent for a whole variable pass ago)
            __syncthreads();
            const int any = red[par * 2 + 1];
            // ---- stop rules (mpdecode_core.c:466-483) ----
            if (!any) { result = iter + 1; break; }                     // "zero bit errors" against the all-zero data[]
            pcc = ssum; pcc_written = 1;
            if (ssum == WR_NPAR) { result = iter + 1; break; }
        }
        // what a short packet left undone of the look-ahead (synchronously; the next packet then fetches its symbols itself)
        if (ahead == 0) { if (tid == 0) fetch_record(nxt, cur ^ 1); }
        else if (ahead == 1 && tid == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

        // ---- the decisions as the packet's bytes, MSB first (drs232_ldpc.c:234-239): OR-ed into 81 words; CRC gate: wenet_crc_kernel ----
#pragma unroll
        for (int t = 0; t < WR_VARS_PER_THREAD; t++) {
            if ((svalid >> t) & 1u) {
                const unsigned v = (svar[t >> 1] >> (16 * (t & 1))) & 0xffffu;
                if (A.bits_out) A.bits_out[slot * WR_NCODE + v] = (uint8_t)((bits >> t) & 1u);
                if ((bits >> t) & 1u) atomicOr(&WORDS[v >> 5], 1u << ((((v >> 3) & 3u) << 3) + 7u - (v & 7u)));
            }
        }
        __syncthreads();
        WrPacketOut *out = A.out ? &A.out[slot] : nullptr;
        if (out) {
            if (tid < 65) {
                unsigned wv = WORDS[tid];
                if (tid == 64) wv &= 0xffffu;                            // bytes 256, 257; crc_ok and done (wenet_crc_kernel sets them) start as 0
                ((unsigned *)out->bytes)[tid] = wv;
            }
            if (tid == 65) { out->iter = result; out->pcc = pcc; out->pcc_written = pcc_written; }
        }
    }
}

extern "C" hipError_t wr_launch_decode2(const WrDecodeArgs *args, hipStream_t stream, int grid) {
    const int lds = WR_D2_LDS_BYTES;
    wr_attr_ok(hipFuncSetAttribute((const void *)wenet_decode2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(wenet_decode2_kernel, dim3(grid), dim3(WR_DEC_THREADS), lds, stream, *args);
    return hipGetLastError();
}
