// orlg_inst_phy.hip -- instantiation of the QoT-aware step kernel (orlg_phy_kernels.hip) for ONE word count,
// -DORLG_INST_W=<W> (see orlg_inst_wave.hip).
#include "orlg_host.h"
#include "orlg_phy_kernels.hip"

#ifndef ORLG_INST_W
#error "compile with -DORLG_INST_W=<words per link>"
#endif
#define ORLG_CAT2(a, b) a##b
#define ORLG_CAT(a, b) ORLG_CAT2(a, b)

// variant bits 0..1 -- 0: the step kernel proper; 1: + periodic defragmentation; 2: + the GN-model admission check;
// bit 2: the policy sorts channels by the RSS metric (floating point) instead of an integer key
orlg_phy_kernel_t ORLG_CAT(orlg_phy_kernel_W, ORLG_INST_W)(int variant) {
    switch (variant) {
        case 0: return orlg_phy_kernel<ORLG_INST_W, false, false, false>;
        case 1: return orlg_phy_kernel<ORLG_INST_W, true, false, false>;
        case 2: return orlg_phy_kernel<ORLG_INST_W, true, true, false>;
        case 4: return orlg_phy_kernel<ORLG_INST_W, false, false, true>;
        case 5: return orlg_phy_kernel<ORLG_INST_W, true, false, true>;
        case 6: return orlg_phy_kernel<ORLG_INST_W, true, true, true>;
        default: return nullptr;
    }
}
