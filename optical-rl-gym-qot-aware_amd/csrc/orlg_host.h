// orlg_host.h -- host-side helpers shared by the translation units of liborlg.so (orlg_api.hip, orlg_phy_api.hip,
// orlg_osnr.hip) and the declarations of the per-shape kernel instantiation units (orlg_inst_*.hip, one object per word
// count W so that the library builds in parallel: build.py).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/orlg.h"
#include "orlg_device.h"

// ---------------------------------------------------------------------------------------- errors
// thread-local message of the last failure (orlg_last_error); defined in orlg_api.hip
int orlg_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
#define fail orlg_fail
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(ORLG_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// CPython _randommodule.c: random.Random(n) -> init_by_array(32-bit little-endian chunks of abs(n))
void orlg_mt_seed(uint32_t *mt, uint64_t seed);
int orlg_is_device_ptr(const void *ptr);
void *orlg_device_alias(const void *ptr);

// checkpoint / resume: the whole simulation state of a handle is a handful of flat device arrays
struct OrlgStatePart { void *ptr; size_t bytes; };
int orlg_state_copy(const std::vector<OrlgStatePart> &parts, void *buffer, bool save, int device, hipStream_t stream);

// Sticky error word of a handle in mapped host memory: a kernel that loses a release (queue overflow) stores 1 there, every
// entry point that waits for the stream looks at it afterwards and reports ORLG_ERR_QUEUE_FULL -- no extra copy, no extra wait.
struct OrlgErrWord {
    volatile int32_t *host;
    int32_t *dev;
};
int orlg_err_word_create(OrlgErrWord *w);
void orlg_err_word_destroy(OrlgErrWord *w);

// ---------------------------------------------------------------------------------------- kernel instantiation units
// Every unit exports one lookup per word count W; a W the library was not built for resolves to a null (weak) symbol.
typedef void (*orlg_rmsa_kernel_t)(const OrlgParams);
typedef void (*orlg_masks_kernel_t)(const OrlgParams, int, int, int, uint64_t *, int32_t *);
enum { ORLG_KIND_STEP = 0, ORLG_KIND_STEP_FF = 1, ORLG_KIND_RESET = 2, ORLG_KIND_OBS = 3, ORLG_KIND_GROUP = 4,
       ORLG_KIND_STEP_DF = 5, ORLG_KIND_STEP_FF_DF = 6 };   // _DF: full statistics with the links' float64 part deferred (link_replay)
#define ORLG_FOR_EACH_W(X) X(1) X(2) X(3) X(4) X(5) X(6) X(8)
#define ORLG_DECL_W(n)                                                                           \
    orlg_rmsa_kernel_t orlg_wave_kernel_W##n(int kind, int stats) __attribute__((weak));          \
    orlg_masks_kernel_t orlg_masks_kernel_W##n() __attribute__((weak));                           \
    orlg_rmsa_kernel_t orlg_group_kernel_W##n(int stats) __attribute__((weak));
ORLG_FOR_EACH_W(ORLG_DECL_W)
#undef ORLG_DECL_W

struct OrlgPhyParams;
typedef void (*orlg_phy_kernel_t)(const OrlgPhyParams);
#define ORLG_FOR_EACH_PHY_W(X) X(1) X(2) X(3) X(4) X(5)
#define ORLG_DECL_PHY_W(n) orlg_phy_kernel_t orlg_phy_kernel_W##n(int variant) __attribute__((weak));
ORLG_FOR_EACH_PHY_W(ORLG_DECL_PHY_W)
#undef ORLG_DECL_PHY_W
