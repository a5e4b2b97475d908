// orlg_inst_group.hip -- instantiations of the four-environments-per-wave step kernel (orlg_group_kernels.hip) for ONE word
// count, -DORLG_INST_W=<W> (see orlg_inst_wave.hip).
#include "orlg_host.h"
#include "orlg_group_kernels.hip"

#ifndef ORLG_INST_W
#error "compile with -DORLG_INST_W=<words per link>"
#endif
#define ORLG_CAT2(a, b) a##b
#define ORLG_CAT(a, b) ORLG_CAT2(a, b)

// stats = statistics level (0..2), + 4: the instantiation that leaves the release queue in HBM (launches of very few steps)
orlg_rmsa_kernel_t ORLG_CAT(orlg_group_kernel_W, ORLG_INST_W)(int stats) {
    constexpr int W = ORLG_INST_W;
    switch (stats) {
        case 0: return orlg_rmsa_group_kernel<W, 0>;
        case 1: return orlg_rmsa_group_kernel<W, 1>;
        case 2: return orlg_rmsa_group_kernel<W, 2>;
        case 4: return orlg_rmsa_group_kernel<W, 0, true>;
        case 5: return orlg_rmsa_group_kernel<W, 1, true>;
        case 6: return orlg_rmsa_group_kernel<W, 2, true>;
        case 10: return orlg_rmsa_group_kernel<W, 2, false, true>;   // + 8: full statistics with the link updates deferred (long launches)
        default: return nullptr;
    }
}
