// orlg_inst_phy.hip -- instantiation of the QoT-aware step kernel (orlg_phy_kernels.hip) for ONE word count,
// -DORLG_INST_W=<W> (see orlg_inst_wave.hip).
#include "orlg_host.h"
#include "orlg_phy_kernels.hip"

#ifndef ORLG_INST_W
#error "compile with -DORLG_INST_W=<words per link>"
#endif
#define ORLG_CAT2(a, b) a##b
#define ORLG_CAT(a, b) ORLG_CAT2(a, b)

// variant = extra + 4 * (policy + 1); extra -- 0: the step kernel proper; 1: + periodic defragmentation; 2: + defragmentation
// and the GN-model admission check; 3: + the GN-model admission check alone (a handle without defrag_period does not carry the
// defragmentation's registers); policy = ORLG_PHY_POLICY_* (-1 external actions .. 6)
#define ORLG_PHY_POL_CASES(base, POL)                                                     \
    case base: return orlg_phy_kernel<ORLG_INST_W, false, false, POL>;                    \
    case base + 1: return orlg_phy_kernel<ORLG_INST_W, true, false, POL>;                 \
    case base + 2: return orlg_phy_kernel<ORLG_INST_W, true, true, POL>;                  \
    case base + 3: return orlg_phy_kernel<ORLG_INST_W, false, true, POL>;
orlg_phy_kernel_t ORLG_CAT(orlg_phy_kernel_W, ORLG_INST_W)(int variant) {
    switch (variant) {
        ORLG_PHY_POL_CASES(0, ORLG_PHY_POLICY_EXTERNAL)
        ORLG_PHY_POL_CASES(4, ORLG_PHY_POLICY_BMFA_CUT)
#ifndef ORLG_PHY_FEW_POLICIES   // (instrumented single-unit builds of tools/: external actions and bmfa only)
        ORLG_PHY_POL_CASES(8, ORLG_PHY_POLICY_BMFA_RSS_METRIC)
        ORLG_PHY_POL_CASES(12, ORLG_PHY_POLICY_SAPFF)
        ORLG_PHY_POL_CASES(16, ORLG_PHY_POLICY_BMFF)
        ORLG_PHY_POL_CASES(20, ORLG_PHY_POLICY_SAPBM)
        ORLG_PHY_POL_CASES(24, ORLG_PHY_POLICY_FAFF)
        ORLG_PHY_POL_CASES(28, ORLG_PHY_POLICY_FAFF_RSS)
#endif
        default: return nullptr;
    }
}
