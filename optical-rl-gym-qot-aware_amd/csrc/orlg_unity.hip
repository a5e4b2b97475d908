// orlg_unity.hip -- the whole library as ONE translation unit with the kernels of a single word count (default W = 5:
// NSFNET-320, US14-268).  Only for the instrumented builds of tools/ (-DORLG_SECTIONS keeps one orlg_sections[] array,
// -DORLG_SHAPE_ASSUME=...); the product is built from the separate units by build.py.
#ifndef ORLG_INST_W
#define ORLG_INST_W 5
#endif
#include "orlg_api.hip"
#include "orlg_phy_api.hip"
#include "orlg_osnr.hip"
#include "orlg_inst_wave.hip"
#include "orlg_inst_group.hip"
#include "orlg_inst_phy.hip"
