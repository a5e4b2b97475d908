// orlg_inst_group.hip -- instantiations of the four-environments-per-wave step kernel (orlg_group_kernels.hip) for ONE word
// count, -DORLG_INST_W=<W> (see orlg_inst_wave.hip).
#include "orlg_host.h"
#include "orlg_group_kernels.hip"

#ifndef ORLG_INST_W
#error "compile with -DORLG_INST_W=<words per link>"
#endif
#define ORLG_CAT2(a, b) a##b
#define ORLG_CAT(a, b) ORLG_CAT2(a, b)

orlg_rmsa_kernel_t ORLG_CAT(orlg_group_kernel_W, ORLG_INST_W)(int stats) {
    constexpr int W = ORLG_INST_W;
    return stats == 0 ? orlg_rmsa_group_kernel<W, 0> : stats == 1 ? orlg_rmsa_group_kernel<W, 1> : orlg_rmsa_group_kernel<W, 2>;
}
