// orlg_inst_phy.hip -- instantiation of the QoT-aware step kernel (orlg_phy_kernels.hip) for ONE word count,
// -DORLG_INST_W=<W> (see orlg_inst_wave.hip).
#include "orlg_host.h"
#include "orlg_phy_kernels.hip"

#ifndef ORLG_INST_W
#error "compile with -DORLG_INST_W=<words per link>"
#endif
#define ORLG_CAT2(a, b) a##b
#define ORLG_CAT(a, b) ORLG_CAT2(a, b)

// variant 0: the step kernel proper; 1: + periodic defragmentation (and the node-degree vectors of its cut metric); 2: + the
// GN-model admission check
orlg_phy_kernel_t ORLG_CAT(orlg_phy_kernel_W, ORLG_INST_W)(int variant) {
    return variant == 2 ? orlg_phy_kernel<ORLG_INST_W, true, true>
                        : variant == 1 ? orlg_phy_kernel<ORLG_INST_W, true, false> : orlg_phy_kernel<ORLG_INST_W, false, false>;
}
