// orlg_math.h -- double-precision helpers shared by the HIP kernels and the host side of liborlg.
//
// orlg_log(): natural logarithm used by the arrival process (the reference draws inter-arrival and
// holding times with random.expovariate = -log(1 - u) / lambda, rmsa_env.py:646-651).  Python takes
// log from the platform libm, which is not bit-reproducible across platforms; the device needs a
// routine whose result is a pure function of IEEE-754 +,-,*,/ so that the host build (orlg_host_log,
// used by the parity tests to drive the CPU oracle with the same arithmetic) and the gfx950 build
// agree bit for bit.  The algorithm is the classic argument-reduction + degree-14 minimax polynomial
// in s = f/(2+f) (error < 1 ulp); it must be compiled with FP contraction OFF on both sides.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define ORLG_HD __host__ __device__ inline
#else
#define ORLG_HD static inline
#endif

ORLG_HD uint64_t orlg_d2u(double d) {
    union { double d; uint64_t u; } c;
    c.d = d;
    return c.u;
}
ORLG_HD double orlg_u2d(uint64_t u) {
    union { double d; uint64_t u; } c;
    c.u = u;
    return c.d;
}

// a / b: on the device the gfx9 fdiv-f64 expansion without v_div_scale / v_div_fixup (identities for the normal-range
// operands used here), i.e. the same correctly rounded quotient as the host's IEEE division, in 8 instead of ~14
// instructions.  Host/device agreement is asserted by the GPU parity tests (every float bit-exact vs the oracle).
#if defined(__HIP_DEVICE_COMPILE__)
__device__ inline double orlg_fdiv(double a, double b) {
    double y = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    double q = a * y;
    double r = __builtin_fma(-b, q, a);
    return __builtin_fma(r, y, q);
}
#define ORLG_FDIV(a, b) orlg_fdiv((a), (b))
#else
#define ORLG_FDIV(a, b) ((a) / (b))
#endif

ORLG_HD double orlg_log(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    uint64_t ux = orlg_d2u(x);
    int32_t hx = (int32_t)(ux >> 32);
    uint32_t lx = (uint32_t)ux;
    int32_t k = 0;
    if (hx < 0x00100000) {  // x < 2^-1022, zero or negative
        if (((hx & 0x7fffffff) | lx) == 0) return orlg_u2d(0xfff0000000000000ull);  // log(0) = -inf
        if (hx < 0) return orlg_u2d(0x7ff8000000000000ull);                         // log(<0) = nan
        k -= 54;
        x *= 18014398509481984.0;  // 2^54
        hx = (int32_t)(orlg_d2u(x) >> 32);
    }
    if (hx >= 0x7ff00000) return x + x;
    k += (hx >> 20) - 1023;
    hx &= 0x000fffff;
    int32_t i = (hx + 0x95f64) & 0x100000;
    x = orlg_u2d((orlg_d2u(x) & 0xffffffffull) | ((uint64_t)(uint32_t)(hx | (i ^ 0x3ff00000)) << 32));
    k += (i >> 20);
    double f = x - 1.0;
    double dk = (double)k;
    if ((0x000fffff & (2 + hx)) < 3) {  // |f| < 2^-20
        if (f == 0.0) {
            if (k == 0) return 0.0;
            return dk * ln2_hi + dk * ln2_lo;
        }
        double R = f * f * (0.5 - 0.33333333333333333 * f);
        if (k == 0) return f - R;
        return dk * ln2_hi - ((R - dk * ln2_lo) - f);
    }
    double s = ORLG_FDIV(f, 2.0 + f);
    double z = s * s;
    i = hx - 0x6147a;
    double w = z * z;
    int32_t j = 0x6b851 - hx;
    double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    i |= j;
    double R = t2 + t1;
    if (i > 0) {
        double hfsq = 0.5 * f * f;
        if (k == 0) return f - (hfsq - s * (hfsq + R));
        return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    }
    if (k == 0) return f - s * (f - R);
    return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}
