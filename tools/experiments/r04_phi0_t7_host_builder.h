// Host side of the rejected one-read phi0 table (tools/experiments/README.md, r04_phi0_one_read_table.patch): goes into wenet_amd/csrc/ldpc_host_tables.h next to
// phi0_build_lut; LdpcTables::build (wenet_rx.hip) then calls phi0_build_t7 instead and uploads WR_PHI0_LDS_BYTES.  Verified here against the reference form on every float
// from 2^-17 to 32 (3 s on one core): 14 marked cells.
// blob = float table[2562] (padded to 16 bytes) | {threshold bits, value below, value at / above, 0} x 16
inline float phi0_t7_eval(const uint32_t *blob, float xf) {               // host twin of the kernel's lookup, the caller's >= 32768 rule included
    int32_t b; memcpy(&b, &xf, 4);
    uint32_t u;
    if (b >= WR_PHI0_BIG_BITS) u = f2u(10.0f);
    else {
        int k = b >> 16;
        k = (k < WR_PHI0_T7_KLO ? WR_PHI0_T7_KLO : (k > WR_PHI0_T7_KHI ? WR_PHI0_T7_KHI : k)) - WR_PHI0_T7_KLO;
        u = blob[k];
        if ((u & 0x7fffffffu) > 0x7f800000u) {                            // a marked cell
            const uint32_t *e = blob + WR_PHI0_T7_BYTES / 4 + 4 * (u & 15u);
            u = (b >= (int32_t)e[0]) ? e[2] : e[1];
        }
    }
    float f; memcpy(&f, &u, 4); return f;
}
inline bool phi0_build_t7(std::vector<uint32_t> &blob, bool exhaustive) {
    blob.assign((WR_PHI0_T7_BYTES + WR_PHI0_T7_SPECIALS * 16) / 4, 0);
    uint32_t *spec = blob.data() + WR_PHI0_T7_BYTES / 4;
    blob[0] = f2u(10.0f);
    blob[WR_PHI0_T7_ENTRIES - 1] = f2u(0.0f);
    int nspec = 0;
    std::vector<uint32_t> steps_at;
    for (int i = 1; i <= WR_PHI0_BINADES * WR_PHI0_T7_CELLS; i++) {
        const int p = (i - 1) / WR_PHI0_T7_CELLS, c = (i - 1) % WR_PHI0_T7_CELLS;
        const double lo = ldexp(1.0 + (double)c / WR_PHI0_T7_CELLS, p), hi = ldexp(1.0 + (double)(c + 1) / WR_PHI0_T7_CELLS, p);      // the cell in y = x * 65536
        const int x0 = (int)floor(lo), x1 = (int)ceil(hi) - 1;                  // integer parts met inside the cell
        blob[i] = f2u(phi0_linear_int(x0));
        int steps = 0, at = 0;
        for (int x = x0 + 1; x <= x1; x++) if (f2u(phi0_linear_int(x)) != f2u(phi0_linear_int(x - 1))) { steps++; at = x; }
        if (steps == 0) continue;
        if (steps > 1 || nspec >= WR_PHI0_T7_SPECIALS) return false;
        spec[4 * nspec + 0] = f2u((float)at) - 0x08000000u;                      // the step as bits of xf = y / 65536
        spec[4 * nspec + 1] = f2u(phi0_linear_int(at - 1));
        spec[4 * nspec + 2] = f2u(phi0_linear_int(at));
        steps_at.push_back(spec[4 * nspec]);
        blob[i] = WR_PHI0_T7_MARK | (uint32_t)nspec;
        nspec++;
    }
    auto bad = [&](uint32_t u) { float xf; memcpy(&xf, &u, 4); return f2u(phi0_t7_eval(blob.data(), xf)) != f2u(phi0_x86(xf)); };
    if (exhaustive) {
        for (uint32_t u = 0x37000000u; u < 0x42000000u; u++) if (bad(u)) return false;
        for (uint64_t u = 0; u < 0x100000000ull; u += 4099) if (bad((uint32_t)u)) return false;
    }
    for (uint32_t k = 0x3700u; k < 0x4200u; k++) for (uint32_t lo16 : {0x0000u, 0x0001u, 0xfffeu, 0xffffu}) if (bad((k << 16) | lo16)) return false;
    for (uint32_t t : steps_at) for (int d = -2; d <= 2; d++) if (bad(t + (uint32_t)d)) return false;
    return true;
}
