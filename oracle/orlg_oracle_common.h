/*
 * oracle/orlg_oracle_common.h -- TEST INFRASTRUCTURE (see orlg_oracle.h): pieces shared by the RMSA and the
 * PhyRMSA restatements: CPython's random.Random (MT19937, random(), expovariate, choices) and numpy's
 * float64 pairwise summation.
 */
#ifndef ORLG_ORACLE_COMMON_H
#define ORLG_ORACLE_COMMON_H
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#if defined(__GNUC__)
#define ORC_UNUSED __attribute__((unused))
#else
#define ORC_UNUSED
#endif

/* ------------------------------------------------------------------ MT19937 / CPython random */
#define MT_N 624
#define MT_M 397
typedef struct { uint32_t mt[MT_N]; int idx; } py_rng;

ORC_UNUSED static void mt_init_genrand(py_rng *r, uint32_t s) {
    r->mt[0] = s;
    for (int i = 1; i < MT_N; i++)
        r->mt[i] = 1812433253u * (r->mt[i - 1] ^ (r->mt[i - 1] >> 30)) + (uint32_t)i;
    r->idx = MT_N;
}

ORC_UNUSED static void mt_init_by_array(py_rng *r, const uint32_t *key, int len) {
    mt_init_genrand(r, 19650218u);
    int i = 1, j = 0;
    int k = MT_N > len ? MT_N : len;
    for (; k; k--) {
        r->mt[i] = (r->mt[i] ^ ((r->mt[i - 1] ^ (r->mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        i++; j++;
        if (i >= MT_N) { r->mt[0] = r->mt[MT_N - 1]; i = 1; }
        if (j >= len) j = 0;
    }
    for (k = MT_N - 1; k; k--) {
        r->mt[i] = (r->mt[i] ^ ((r->mt[i - 1] ^ (r->mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        i++;
        if (i >= MT_N) { r->mt[0] = r->mt[MT_N - 1]; i = 1; }
    }
    r->mt[0] = 0x80000000u;
}

/* random.Random(int) : key = 32-bit little-endian chunks of abs(seed), at least one chunk */
ORC_UNUSED static void py_seed(py_rng *r, uint64_t seed) {
    uint32_t key[2] = { (uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32) };
    mt_init_by_array(r, key, key[1] ? 2 : 1);
}

ORC_UNUSED static uint32_t mt_genrand(py_rng *r) {
    static const uint32_t mag01[2] = { 0u, 0x9908b0dfu };
    uint32_t y;
    if (r->idx >= MT_N) {
        int kk;
        uint32_t *mt = r->mt;
        for (kk = 0; kk < MT_N - MT_M; kk++) {
            y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + MT_M] ^ (y >> 1) ^ mag01[y & 1u];
        }
        for (; kk < MT_N - 1; kk++) {
            y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + (MT_M - MT_N)] ^ (y >> 1) ^ mag01[y & 1u];
        }
        y = (mt[MT_N - 1] & 0x80000000u) | (mt[0] & 0x7fffffffu);
        mt[MT_N - 1] = mt[MT_M - 1] ^ (y >> 1) ^ mag01[y & 1u];
        r->idx = 0;
    }
    y = r->mt[r->idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

ORC_UNUSED static double py_random(py_rng *r) {
    uint32_t a = mt_genrand(r) >> 5, b = mt_genrand(r) >> 6;
    return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
}

/* Lib/random.py expovariate: -log(1.0 - random()) / lambd.  Python takes log from the platform libm;
 * orc_set_log_fn() lets a test substitute another log (the product's host build of its device log) so
 * that the oracle and the device can be compared bit for bit on every float as well. */
extern double (*orc_g_log)(double);
/* random.randrange / randint over `n` values: Random._randbelow_with_getrandbits (Lib/random.py): k = n.bit_length();
 * r = getrandbits(k) while r >= n; getrandbits(k <= 32) = genrand_uint32() >> (32 - k) (_randommodule.c) */
ORC_UNUSED static uint32_t py_randbelow(py_rng *r, uint32_t n) {
    int k = 0;
    for (uint32_t t = n; t; t >>= 1) k++;
    uint32_t v = mt_genrand(r) >> (32 - k);
    while (v >= n) v = mt_genrand(r) >> (32 - k);
    return v;
}
ORC_UNUSED static double py_expovariate(py_rng *r, double lambd) { return -orc_g_log(1.0 - py_random(r)) / lambd; }

/* Lib/random.py choices(k=1) with cumulative weights: bisect_right(cum, random()*total, 0, n-1) */
ORC_UNUSED static int py_choice_cum(py_rng *r, const double *cum, int n) {
    double total = cum[n - 1] + 0.0;
    double x = py_random(r) * total;
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        int mid = (lo + hi) / 2;
        if (x < cum[mid]) hi = mid; else lo = mid + 1;
    }
    return lo;
}


/* ------------------------------------------------------------------ numpy float64 sum / mean */
/* numpy/core/src/umath/loops_utils.h.src pairwise_sum (add.reduce starts from the identity 0) */
ORC_UNUSED static double np_pairwise(const double *a, int n) {
    if (n < 8) {
        double res = 0.;
        for (int i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8], res;
        int i;
        for (i = 0; i < 8; i++) r[i] = a[i];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int q = 0; q < 8; q++) r[q] += a[i + q];
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        int n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise(a, n2) + np_pairwise(a + n2, n - n2);
    }
}
ORC_UNUSED static double np_mean(const double *a, int n) { return np_pairwise(a, n) / (double)n; }


#endif
