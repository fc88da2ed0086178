// demod_chain_split.h -- the lane-split NCO chain wave of the pipelined kernels, ONE implementation for the one-capture kernel
// (demod_pipe_impl.h: CAPS = 1) and the three-captures-per-workgroup kernel (demod_tri_impl.h: CAPS = 3).  Included by both after their
// WP_* / CT_* definitions.
#pragma once

// NCO chain of one frame with the real / imaginary part of tone m in lanes 2m / 2m+1 (nco_step_split): the batch form of
// C(j) below -- same statements, half the SIMD time per step, a longer dependent path.  Out of line so that the kernel's
// register allocation (80 VGPRs in the three-captures-per-CU variant) is not disturbed by it.
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) int lds_i32;
template <int CAPS>
__device__ __forceinline__ void nco_chain_split(int j, int nin_j, int lane, int M, int N, int NH, int Nmem, int L, lds_i32 *CT, lds_f32 *PHE,
                                             lds_f32 *CKb, lds_f32 *CKD, const lds_f32 *dphi_t, const float *bin_freq, const float2 *backoff_tab,
                                             int capmask, int cap_stride_words) {
    // lanes [2M*c, 2M*(c+1)) carry capture c: tone m = pair index, part = re / im; pointers move to that capture's LDS block
    const int cap = lane / (2 * M);
    if (cap >= CAPS || !((capmask >> cap) & 1)) return;
    const int m = (lane - cap * 2 * M) >> 1, part = lane & 1;
    CT += cap * cap_stride_words; PHE += cap * cap_stride_words; CKb += cap * cap_stride_words; CKD += cap * cap_stride_words;
    const int nold = Nmem - nin_j;
    int bc = CT[CT_FBIN + (j & 3) * 4 + m];
    int bp = CT[CT_FBIN + ((j + 3) & 3) * 4 + m];
    const int bp0 = CT[CT_FBIN + ((j + 3) & 3) * 4 + 0];
    if (bin_freq[bp0] < 1.0f) bp = bc;                                   // first run (fsk.c:750-753)
    const int ncase = (nin_j < N) ? 0 : ((nin_j > N) ? 2 : 1);
    const float2 bo = backoff_tab[ncase * NH + bp];
    const lds_f32 *pc = PHE + (((j + 2) % 3) * 4 + m) * 2;
    const v2f phi0 = cmul_pk((v2f){bo.x, bo.y}, (v2f){pc[0], pc[1]});    // fsk.c:758-759 (both lanes of the pair)
    float own = part ? phi0.y : phi0.x;
    float dx = dphi_t[2 * bp], dy = dphi_t[2 * bp + 1];
    float k1 = dx, k2 = part ? dy : -dy;
    lds_f32 *ckA = CKb + ((((j & 1) * 2 + 0) * M + m) * WP_CKROW) * 2 + part;
    lds_f32 *ckB = CKb + ((((j & 1) * 2 + 1) * M + m) * WP_CKROW) * 2 + part;
    CKD[(((j & 1) * 2 + 0) * M + m) * 2 + part] = part ? dy : dx;
    int s = 0, c = 0;
    for (; s + WP_CK <= nold; s += WP_CK, c++) {
        ckA[2 * c] = own;
        static_assert(WP_CK == 8, "nco_step_split8"); own = nco_step_split8(own, k1, k2);
    }
    if (s < nold) { ckA[2 * c] = own; for (; s < nold; s++) own = nco_step_split(own, k1, k2); }
    {
        const float oth = __shfl_xor(own, 1, 64);
        const float re = part ? oth : own, im = part ? own : oth;
        const float av = sqrtf(re * re + im * im);                       // comp_normalize (fsk.c:787)
        own = own / av;
        dx = dphi_t[2 * bc]; dy = dphi_t[2 * bc + 1];
        k1 = dx; k2 = part ? dy : -dy;
    }
    CKD[(((j & 1) * 2 + 1) * M + m) * 2 + part] = part ? dy : dx;
    c = 0;
    for (; s + 4 * WP_CK <= L; s += 4 * WP_CK, c += 4) {                     // four checkpoints per trip: a taken branch costs ~16 cycles
#pragma unroll
        for (int k = 0; k < 4; k++) {
            ckB[2 * (c + k)] = own;
            own = nco_step_split8(own, k1, k2);
        }
    }
    for (; s + WP_CK <= L; s += WP_CK, c++) {
        ckB[2 * c] = own;
        static_assert(WP_CK == 8, "nco_step_split8"); own = nco_step_split8(own, k1, k2);
    }
    if (s < L) { ckB[2 * c] = own; for (; s < L; s++) own = nco_step_split(own, k1, k2); }
    PHE[((j % 3) * 4 + m) * 2 + part] = own;                             // un-normalised (fsk.c:846)
}
