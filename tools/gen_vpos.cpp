// tools/gen_vpos.cpp -- writes wenet_amd/csrc/tables/ldpc_vpos.inc: the placement of the code's variables on the decoder's threads (position tid + 512 t),
// found by the local search of ldpc_host_tables.h (deterministic: fixed seed) so that the variable pass loads the LDS banks evenly.
//   g++ -O2 -std=c++17 -DWR_GEN_VPOS -Iwenet_amd/csrc tools/gen_vpos.cpp -o /tmp/gen_vpos && /tmp/gen_vpos > wenet_amd/csrc/tables/ldpc_vpos.inc
#include "ldpc_host_tables.h"

int main() {
    std::vector<uint16_t> vedge, vpos;
    if (!ldpc_build_vedge(vedge)) return 1;
    int c0 = 0, c1 = 0;
    place_variables(vpos, [&](int v, int k) { return vedge[v * 3 + k] & 31; }, &c0, &c1);
    if (!ldpc_vpos_valid(vpos.data())) { fprintf(stderr, "gen_vpos: invalid placement\n"); return 1; }
    fprintf(stderr, "gen_vpos: bank overload %d -> %d\n", c0, c1);
    printf("// variable at position p of the decoder's variable pass (thread p %% 512, its (p / 512)-th): tools/gen_vpos.cpp, bank overload %d -> %d\n", c0, c1);
    for (int p = 0; p < WR_NCODE; p++) printf("%d,%s", vpos[p], (p % 20 == 19) ? "\n" : " ");
    printf("\n");
    return 0;
}
