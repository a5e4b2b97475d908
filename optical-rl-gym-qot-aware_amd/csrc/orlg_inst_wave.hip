// orlg_inst_wave.hip -- instantiations of the wave-per-environment kernels (orlg_kernels.hip) for ONE word count,
// -DORLG_INST_W=<W>: one object per W, so that the library builds in parallel (build.py).
#include "orlg_host.h"
#include "orlg_kernels.hip"

#ifndef ORLG_INST_W
#error "compile with -DORLG_INST_W=<words per link>"
#endif
#define ORLG_CAT2(a, b) a##b
#define ORLG_CAT(a, b) ORLG_CAT2(a, b)

template <int W>
static orlg_rmsa_kernel_t pick_stats(int kind, int stats) {
    switch (kind) {
        case ORLG_KIND_STEP:
            return stats == 0 ? orlg_rmsa_kernel<W, 0> : stats == 1 ? orlg_rmsa_kernel<W, 1> : orlg_rmsa_kernel<W, 2>;
        case ORLG_KIND_STEP_FF:  // first-fit policies only (k <= 8)
            return stats == 0 ? orlg_rmsa_kernel_ff<W, 0> : stats == 1 ? orlg_rmsa_kernel_ff<W, 1> : orlg_rmsa_kernel_ff<W, 2>;
        case ORLG_KIND_STEP_DF:
            return stats == 2 ? orlg_rmsa_kernel<W, 2, true> : nullptr;
        case ORLG_KIND_STEP_FF_DF:
            return stats == 2 ? orlg_rmsa_kernel_ff<W, 2, true> : nullptr;
        case ORLG_KIND_RESET:
            return stats == 0 ? orlg_rmsa_reset_kernel<W, 0> : stats == 1 ? orlg_rmsa_reset_kernel<W, 1> : orlg_rmsa_reset_kernel<W, 2>;
        case ORLG_KIND_OBS:
            return orlg_deeprmsa_obs_kernel<W>;
        default:
            return nullptr;
    }
}

orlg_rmsa_kernel_t ORLG_CAT(orlg_wave_kernel_W, ORLG_INST_W)(int kind, int stats) { return pick_stats<ORLG_INST_W>(kind, stats); }
orlg_masks_kernel_t ORLG_CAT(orlg_masks_kernel_W, ORLG_INST_W)() { return orlg_path_masks_kernel<ORLG_INST_W>; }
