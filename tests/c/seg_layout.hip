// Host-side probe of the wave-per-chain kernels' compile-time layouts (csrc/common.hpp: SegB, seg_lo, seg_obs_share, make_lds_layout),
// compiled and run by tests/test_boundary.py::test_chain_kernel_layouts (no GPU needed: prints one JSON object).
#include "common.hpp"
#include <cstdio>
using namespace gusto;
template <int MODEL> static void model(const char* name, bool last) {
    printf("\"%s\": {", name);
    for (int N : {16, 33, 50, 63, 64}) {
        printf("\"%d\": [%d, %d, %d]%s", N, make_lds_layout<MODEL>(N, false, 0).total, make_lds_layout<MODEL>(N, false, 2).total,
               make_lds_layout<MODEL>(N, false, 4).total, N == 64 ? "" : ", ");
    }
    printf(", \"segB2\": %d, \"segB4\": %d, \"hlb2\": %d, \"hlb4\": %d, \"rp\": %d, \"npg\": %d}%s", SegB<MODEL, 2>::total, SegB<MODEL, 4>::total,
           SegB<MODEL, 2>::HLB, SegB<MODEL, 4>::HLB, SEG_RP, SegB<MODEL, 2>::NPG, last ? "" : ", ");
}
int main() {
    printf("{");
    model<2>("astrobee_se3", false);
    model<3>("astrobee_se3_manifold", false);
    printf("\"min_n\": %d, \"lo\": {", GUSTO_SEG_MIN_N);
    bool first = true;
    for (int nch : {2, 4})
        for (int N : {8, 16, 33, 50, 63, 64}) {
            printf("%s\"%d_%d\": [", first ? "" : ", ", nch, N);
            for (int c = 0; c <= nch; c++) printf("%d%s", seg_lo(c, N, nch), c == nch ? "" : ", ");
            printf("]");
            first = false;
        }
    printf("}, \"share\": {");
    first = true;
    for (int ns : {1, 2, 3, 4})
        for (int r = 0; r < ns; r++) {
            printf("%s\"%d_%d\": \"%llx\"", first ? "" : ", ", r, ns, seg_obs_share(r, ns));
            first = false;
        }
    printf("}}\n");
    return 0;
}
