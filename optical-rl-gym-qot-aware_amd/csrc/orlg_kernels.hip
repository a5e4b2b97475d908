// orlg_kernels.hip -- gfx950 (CDNA4, wave64) kernels of the batched RMSA / DeepRMSA step() path.
//
// Execution model: ONE WAVEFRONT PER ENVIRONMENT.  A workgroup carries up to eight environments (one
// per wave).  The only thing its waves share is a read-only copy of the topology tables staged into
// LDS at kernel start (path records = per-path link index sets, pair table, number-of-slots table,
// cumulative weight tables): one __syncthreads() after staging, none afterwards.  For the duration of a
// launch (n_steps steps) each wave keeps its environment on chip:
//     LDS (per wave)   link x slot free bitmap  E*W uint64        (occ)
//                      release queue            Q x (f64 time, u32 descriptor)
//                      MT19937 state            624 x u32
//                      per-link statistics      4 x E f64, per-link (span, gaps) cache, histograms,
//                      rarely touched counters  (OrlgWaveScalars)
//     SGPR/VGPR        wave-uniform scalars: clock, pending request, integer sums for the compactness
// and reads / writes the HBM copy exactly once, with lane-contiguous (coalesced) accesses.
// Within a step the 64 lanes are used as
//     (path, word) lanes  to AND the link bitmaps of the k candidate paths   (get_available_slots)
//     slot lanes          for the first-fit scan: lane l owns slot 64w+l, a __ballot gives the
//                         lowest fitting start                              (is_path_free loops)
//     (hop, word) lanes   to provision / release a slot window and to rebuild the link statistics
//     queue lanes         to find expired services with one ballot per 64 queue slots
//     node / rate lanes   for CPython's bisect in random.choices
// All fp64 arithmetic reproduces the reference's IEEE operations one for one (compile with
// -ffp-contract=off; the explicit fma() calls below are the hardware's own division sequence).
//
// Reference: optical_rl_gym/envs/rmsa_env.py (step :222-341, _provision_path :462-513, _release_path
// :515-535, _update_network_stats :537-560, _update_link_stats :562-641, _next_service :643-695,
// get_number_slots :708-719, is_path_free :721-734, get_available_slots :745-756, get_available_blocks
// :774-804, _get_network_compactness :806-851, heuristics :854-937), optical_network_env.py
// (_add_release :178-189, _get_node_pair :191-208), deeprmsa_env.py (step :48-58, observation :60-121).
#pragma once
#include <hip/hip_runtime.h>

#include "orlg_device.h"
#include "orlg_math.h"

typedef uint64_t u64;

#define DEV __device__ __forceinline__
#define ORLG_INF_BITS 0x7ff0000000000000ull

// Per-step output arrays: their addresses are kept in LDS (Tab::outs), and a pointer read from memory is a generic pointer --
// every store through it would be a flat instruction, which waits on both memory counters.  They are global memory.
typedef int orlg_v4i __attribute__((ext_vector_type(4)));
#define ORLG_GPTR(T, v) ((T __attribute__((address_space(1))) *)(v))   // an output array: global memory, not a generic pointer
// ---------------------------------------------------------------------------------------- wave helpers
DEV void wave_sync() {
    // LDS hand-off between lanes of ONE wave: hardware executes a wave's LDS operations in order, the
    // fences only stop the compiler from caching or reordering across the hand-off.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
DEV int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
DEV u64 readlane64(u64 v, int l) {
    uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
    return ((u64)hi << 32) | lo;
}
DEV double readlane_d(double v, int l) { return __longlong_as_double((long long)readlane64((u64)__double_as_longlong(v), l)); }
DEV u64 ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
DEV int ctz64(u64 v) { return __builtin_ctzll(v); }
DEV int clz64(u64 v) { return __builtin_clzll(v); }
DEV int popc64(u64 v) { return __builtin_popcountll(v); }

// valid slot bits of word w of a link's bitmap (slots >= S do not exist and are stored as 0 = not free)
DEV u64 valid_mask(int S, int w) {
    int nv = S - 64 * w;
    return nv >= 64 ? ~0ull : (nv <= 0 ? 0ull : ((1ull << nv) - 1ull));
}

// bits of the slot window [s, s+n) that fall in word w
DEV u64 window_mask(int s, int n, int w) {
    int lo = s - 64 * w, hi = s + n - 64 * w;
    lo = lo < 0 ? 0 : lo;
    hi = hi > 64 ? 64 : hi;
    if (hi <= lo) return 0ull;
    int len = hi - lo;
    u64 m = len >= 64 ? ~0ull : ((1ull << len) - 1ull);
    return m << lo;
}

// ---------------------------------------------------------------------------------------- fp64 division
// x / b for several numerators over ONE denominator.  This is the gfx9 fdiv-f64 expansion itself
// (v_rcp_f64, two Newton steps, quotient, residual, final fma) with the denominator-only part hoisted;
// v_div_scale / v_div_fixup are identities for the operand ranges here (simulation clock in
// (0, 1e12), numerators below 1e18), so every quotient is the correctly rounded IEEE quotient the
// reference computes.  Checked bit for bit against the oracle in tests/test_gpu_rmsa.py.
DEV double recip_refine(double b) {
    double y = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-b, y, 1.0);
    return __builtin_fma(y, e, y);
}
DEV double div_by(double a, double b, double y) {
    double q = a * y;
    double r = __builtin_fma(-b, q, a);
    return __builtin_fma(r, y, q);
}

// ---------------------------------------------------------------------------------------- contexts
struct Tab {  // topology tables staged in LDS (shared by the waves of a workgroup, read-only)
    const int32_t *pair_base;
    const OrlgPathRec *recs;
    const uint16_t *nslots;
    const int32_t *bit_rates;
    const double *br_cum, *src_cum, *dst_cum;
    const double *div_s, *inv_k;   // k / S and 1 / k tables (full statistics)
    const u64 *outs;
};

struct Wave {  // this wave's environment in LDS
    int lane;
    u64 *occ;
    double *qtime;
    uint32_t *qdesc;
    uint32_t *mt;
    double *lst;   // [4][E]
    int32_t *hist; // [4][NBR]
    int32_t *lint; // [E] span | gaps << 16
    uint32_t *scratch;
    OrlgWaveScalars *wsc;
    double *ring_iat, *ring_ht;  // [ORLG_RING] pre-generated arrivals
    uint32_t *ring_req;          // [ORLG_RING]
};

DEV Tab make_tab(unsigned char *smem, const OrlgParams &p) {
    Tab tb;
    tb.pair_base = reinterpret_cast<const int32_t *>(smem + p.t_pair);
    tb.recs = reinterpret_cast<const OrlgPathRec *>(smem + p.t_recs);
    tb.nslots = reinterpret_cast<const uint16_t *>(smem + p.t_nslots);
    tb.bit_rates = reinterpret_cast<const int32_t *>(smem + p.t_bitrates);
    tb.br_cum = reinterpret_cast<const double *>(smem + p.t_brcum);
    tb.src_cum = reinterpret_cast<const double *>(smem + p.t_srccum);
    tb.dst_cum = reinterpret_cast<const double *>(smem + p.t_dstcum);
    tb.div_s = reinterpret_cast<const double *>(smem + p.t_divs);
    tb.inv_k = reinterpret_cast<const double *>(smem + p.t_inv);
    tb.outs = reinterpret_cast<const u64 *>(smem + p.l_outs);
    return tb;
}

// stage the table blob (and the per-call output pointers) into LDS; every thread of the workgroup takes part
DEV void stage_tables(unsigned char *smem, const OrlgParams &p) {
    const uint4 *src = reinterpret_cast<const uint4 *>(p.tables);
    uint4 *dst = reinterpret_cast<uint4 *>(smem);
    const int n16 = p.tab_bytes >> 4;
    for (int i = threadIdx.x; i < n16; i += blockDim.x) dst[i] = src[i];
#pragma unroll
    for (int i = 0; i < ORLG_NUM_OUTS; ++i)
        if ((int)threadIdx.x == i) reinterpret_cast<u64 *>(smem + p.l_outs)[i] = reinterpret_cast<u64>(p.outs[i]);
    __syncthreads();
}

// bulk copies between an environment's HBM arrays and the wave's LDS region: 16 bytes per lane per instruction (both sides
// are 16-byte aligned: LDS offsets by construction, HBM per-env strides checked by the caller), 4-byte tail
DEV void copy_words(void *dst, const void *src, int bytes, int lane) {
    const int n16 = bytes >> 4;
    const uint4 *s16 = reinterpret_cast<const uint4 *>(src);
    uint4 *d16 = reinterpret_cast<uint4 *>(dst);
    for (int i = lane; i < n16; i += 64) d16[i] = s16[i];
    const uint32_t *s4 = reinterpret_cast<const uint32_t *>(src);
    uint32_t *d4 = reinterpret_cast<uint32_t *>(dst);
    for (int i = (n16 << 2) + lane; i < (bytes >> 2); i += 64) d4[i] = s4[i];
}

// ---------------------------------------------------------------------------------------- MT19937
// Regenerate all 624 words in place (CPython _randommodule.c genrand_uint32).  Sub-round r handles
// kk = 64r + lane; mt[kk+1] is still old (same or later sub-round), mt[kk+397] is old for kk < 227 and
// mt[kk-227] is already new for kk >= 227, exactly as in the sequential loop.
template <typename MT /* pointer to the 624 state words: generic or LDS-qualified */>
DEV void mt_regenerate(MT mt, int lane) {
    for (int r = 0; r < 10; ++r) {
        int kk = 64 * r + lane;
        uint32_t v = 0;
        if (kk < ORLG_MT_N) {
            int k1 = kk + 1 == ORLG_MT_N ? 0 : kk + 1;
            int ks = kk < ORLG_MT_N - ORLG_MT_M ? kk + ORLG_MT_M : kk - (ORLG_MT_N - ORLG_MT_M);
            uint32_t y = (mt[kk] & 0x80000000u) | (mt[k1] & 0x7fffffffu);
            v = mt[ks] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        wave_sync();
        if (kk < ORLG_MT_N) mt[kk] = v;
        wave_sync();
    }
}

// Five consecutive random.random() values, returned wave-uniform.
DEV void draw5(Wave &wv, int &idx, double (&u)[5]) {
    const int lane = wv.lane;
    int avail = ORLG_MT_N - idx;
    uint32_t w = 0;
    if (avail >= 10) {
        if (lane < 10) w = wv.mt[idx + lane];
        idx += 10;
    } else {
        if (lane < avail) w = wv.mt[idx + lane];
        mt_regenerate(wv.mt, lane);
        if (lane >= avail && lane < 10) w = wv.mt[lane - avail];
        idx = 10 - avail;
    }
    w ^= (w >> 11);
    w ^= (w << 7) & 0x9d2c5680u;
    w ^= (w << 15) & 0xefc60000u;
    w ^= (w >> 18);
    uint32_t nb = (uint32_t)__shfl_down((int)w, 1);
    double d = ((double)(w >> 5) * 67108864.0 + (double)(nb >> 6)) * (1.0 / 9007199254740992.0);
#pragma unroll
    for (int q = 0; q < 5; ++q) u[q] = readlane_d(d, 2 * q);
}

// random.choices(population, weights)[0] with cumulative weights: bisect_right(cum, u*total, 0, n-1)
// = number of cum[0..n-2] that are <= x (cum is non-decreasing).
DEV int choice_cum(const double *cum, int n, double u, int lane) {
    double total = cum[n - 1] + 0.0;
    double x = u * total;
    double c = lane < n - 1 ? cum[lane] : 0.0;
    return popc64(ballot(lane < n - 1 && c <= x));
}

// Pre-generate arrivals, one per lane (_next_service's five random() draws each: inter-arrival, holding time, source,
// destination, bit rate -- rmsa_env.py:646-659, optical_network_env.py:197-206).  The arrival process does not depend
// on the network state, so lane j produces request j of the RNG stream: words [idx + 10 j, idx + 10 j + 10).  At most
// one MT19937 regeneration happens inside a refill (n is capped accordingly), exactly where the sequential
// generator would do it.  Returns the number of requests written to the ring.
// (Out of line, so its pointers carry their address space in the signature: through generic pointers every access of the
// MT19937 state and the tables in LDS was a flat instruction.  RING_LDS: the ring lives in LDS (wave-per-environment kernel)
// or in HBM (the other two).  Returns count | new index << 8.)
typedef __attribute__((address_space(3))) uint32_t orlg_lds_u32;
typedef __attribute__((address_space(3))) double orlg_lds_f64;
typedef __attribute__((address_space(3))) const double orlg_lds_cf64;
typedef __attribute__((address_space(1))) uint32_t orlg_glb_u32;
typedef __attribute__((address_space(1))) double orlg_glb_f64;
template <bool RING_LDS>
__device__ __noinline__ int refill_requests_as(orlg_lds_u32 *mt, void *ring_iat_v, void *ring_ht_v, void *ring_req_v,
                                               orlg_lds_cf64 *src_cum, orlg_lds_cf64 *dst_cum, orlg_lds_cf64 *br_cum, int idx,
                                               int N, int NBR, double lam_arrival, double lam_holding, int env) {
    // out of line on purpose: it runs once per ~62 steps and must not add register pressure to the step loop
    const int lane = threadIdx.x & 63;
    const double ylam_arrival = recip_refine(lam_arrival), ylam_holding = recip_refine(lam_holding);
    int n = (2 * ORLG_MT_N - idx) / 10;
    n = n > ORLG_RING ? ORLG_RING : n;
    // A freshly seeded generator (idx == 624: every environment's first refill) hands out 62 - env % 56 requests instead of 62.
    // The request stream is the same whatever a refill's size; what changes is WHEN the environments run dry: batches stepped
    // one launch per step (agent-driven) otherwise refill all at once every 62nd launch, one after the other behind the
    // workgroup's staging-buffer lock.
#ifndef ORLG_EXP_NO_STAGGER   // (experiment: every environment refills in the same launch, the other 61 of 62 launches none)
    if (idx == ORLG_MT_N) { const int cap = 62 - env % 56; n = n > cap ? cap : n; }
#endif
    uint32_t w[10];
    const int g0 = idx + 10 * lane;
#pragma unroll
    for (int k = 0; k < 10; ++k) w[k] = (lane < n && g0 + k < ORLG_MT_N) ? mt[g0 + k] : 0u;
    if (idx + 10 * n > ORLG_MT_N) {
        mt_regenerate(mt, lane);
#pragma unroll
        for (int k = 0; k < 10; ++k)
            if (lane < n && g0 + k >= ORLG_MT_N) w[k] = mt[g0 + k - ORLG_MT_N];
        idx = idx + 10 * n - ORLG_MT_N;
    } else {
        idx += 10 * n;
    }
    double u[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        uint32_t a = w[2 * q], b = w[2 * q + 1];
        a ^= (a >> 11); a ^= (a << 7) & 0x9d2c5680u; a ^= (a << 15) & 0xefc60000u; a ^= (a >> 18);
        b ^= (b >> 11); b ^= (b << 7) & 0x9d2c5680u; b ^= (b << 15) & 0xefc60000u; b ^= (b >> 18);
        u[q] = ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
    }
    const double iat = div_by(-orlg_log(1.0 - u[0]), lam_arrival, ylam_arrival);
    const double ht = div_by(-orlg_log(1.0 - u[1]), lam_holding, ylam_holding);
    // random.choices: bisect_right(cum, u * total, 0, n - 1) = #{i < n - 1 : cum[i] <= x}
    int src = 0, dst = 0, bri = 0;
    {
        const double x = u[2] * (src_cum[N - 1] + 0.0);
        for (int i = 0; i < N - 1; ++i) src += src_cum[i] <= x ? 1 : 0;
    }
    {
        orlg_lds_cf64 *row = dst_cum + src * N;
        const double x = u[3] * (row[N - 1] + 0.0);
        for (int i = 0; i < N - 1; ++i) dst += row[i] <= x ? 1 : 0;
    }
    {
        const double x = u[4] * (br_cum[NBR - 1] + 0.0);
        for (int i = 0; i < NBR - 1; ++i) bri += br_cum[i] <= x ? 1 : 0;
    }
    // entries past n are dead; they are zeroed so that a snapshot of the state does not depend on what the ring held before
    const double o_iat = lane < n ? iat : 0.0, o_ht = lane < n ? ht : 0.0;
    const uint32_t o_rq = lane < n ? ((uint32_t)src | ((uint32_t)dst << 8) | ((uint32_t)bri << 16)) : 0u;
    if (RING_LDS) {
        ((orlg_lds_f64 *)ring_iat_v)[lane] = o_iat; ((orlg_lds_f64 *)ring_ht_v)[lane] = o_ht; ((orlg_lds_u32 *)ring_req_v)[lane] = o_rq;
    } else {
        ((orlg_glb_f64 *)ring_iat_v)[lane] = o_iat; ((orlg_glb_f64 *)ring_ht_v)[lane] = o_ht; ((orlg_glb_u32 *)ring_req_v)[lane] = o_rq;
    }
    wave_sync();
    return n | (idx << 8);
}
// the callers' form: generic pointers in, the new MT19937 index through idx_io
template <bool RING_LDS>
DEV int refill_requests_t(uint32_t *mt, double *ring_iat, double *ring_ht, uint32_t *ring_req, const double *src_cum,
                          const double *dst_cum, const double *br_cum, int *idx_io, int N, int NBR, double lam_arrival,
                          double lam_holding, int env) {
    const int r = refill_requests_as<RING_LDS>((orlg_lds_u32 *)mt, ring_iat, ring_ht, ring_req, (orlg_lds_cf64 *)src_cum,
                                               (orlg_lds_cf64 *)dst_cum, (orlg_lds_cf64 *)br_cum, *idx_io, N, NBR, lam_arrival,
                                               lam_holding, env);
    *idx_io = r >> 8;
    return r & 0xff;
}
DEV int refill_requests(uint32_t *mt, double *ring_iat, double *ring_ht, uint32_t *ring_req, const double *src_cum,
                        const double *dst_cum, const double *br_cum, int *idx_io, int N, int NBR, double lam_arrival,
                        double lam_holding, int env) {   // ring in HBM
    return refill_requests_t<false>(mt, ring_iat, ring_ht, ring_req, src_cum, dst_cum, br_cum, idx_io, N, NBR, lam_arrival, lam_holding, env);
}

// bit_rate_selection="continuous" (rmsa_env.py:95-101, 655-659): the bit rate is rng.randint(lower, higher) = lower +
// _randbelow(width), CPython's _randbelow_with_getrandbits: k = width.bit_length(); r = getrandbits(k) -- one MT19937 word
// shifted right by 32 - k -- until r < width.  A request then consumes eight words for its four random() values and a
// VARIABLE number for the bit rate, so request j no longer starts at a known word.  Two phases: (1) one walk over the word
// stream, wave-uniform, that only looks at the bit-rate words -- where every request starts and which r it accepts (~25
// instructions per request); (2) lane j computes request j from its eight words like the discrete generator.  A refill
// stays inside the state's current 624 words; the request that straddles a regeneration is generated alone, word by word.
// The ring entry holds r (the index into the table of the width bit rates lower .. higher).  Returns count | new index << 8.
template <bool RING_LDS>
__device__ __noinline__ int refill_requests_cont_as(orlg_lds_u32 *mt, void *ring_iat_v, void *ring_ht_v, void *ring_req_v,
                                                    orlg_lds_cf64 *src_cum, orlg_lds_cf64 *dst_cum, int idx, int N, int width,
                                                    double lam_arrival, double lam_holding) {
    const int lane = threadIdx.x & 63;
    const double ylam_arrival = recip_refine(lam_arrival), ylam_holding = recip_refine(lam_holding);
    const int sh = 32 - (32 - __builtin_clz((unsigned)width));   // 32 - k, k = width.bit_length()
    auto temper = [](uint32_t y) { y ^= (y >> 11); y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= (y >> 18); return y; };
    if (idx >= ORLG_MT_N) { mt_regenerate(mt, lane); idx = 0; }
    // phase 1: the requests that lie inside [idx, 624)
    int n = 0, my_off = 0, my_r = 0, off = idx;
    for (; n < ORLG_RING; ++n) {
        int w = off + 8;
        if (w >= ORLG_MT_N) break;
        int r = 0;
        bool got = false;
        while (w < ORLG_MT_N) {
            r = (int)(temper(mt[w]) >> sh);
            w += 1;
            if (r < width) { got = true; break; }
        }
        if (!got) break;          // its bit-rate draws run past the state's end
        if (lane == n) { my_off = off; my_r = r; }
        off = w;
    }
    double u[4];
    if (n == 0) {
        // the straddler: word by word through the regeneration, every lane the same values
        uint32_t wq[8];
        int r = 0;
        for (int k = 0;; ++k) {
            if (off >= ORLG_MT_N) { mt_regenerate(mt, lane); off = 0; }
            const uint32_t y = temper(mt[off]);
            off += 1;
            if (k < 8) { wq[k] = y; continue; }
            r = (int)(y >> sh);
            if (r < width) break;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) u[q] = ((double)(wq[2 * q] >> 5) * 67108864.0 + (double)(wq[2 * q + 1] >> 6)) * (1.0 / 9007199254740992.0);
        my_r = r;
        n = 1;
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t a = lane < n ? temper(mt[my_off + 2 * q]) : 0u, b = lane < n ? temper(mt[my_off + 2 * q + 1]) : 0u;
            u[q] = ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
        }
    }
    idx = off;
    const double iat = div_by(-orlg_log(1.0 - u[0]), lam_arrival, ylam_arrival);
    const double ht = div_by(-orlg_log(1.0 - u[1]), lam_holding, ylam_holding);
    int src = 0, dst = 0;
    {
        const double x = u[2] * (src_cum[N - 1] + 0.0);
        for (int i = 0; i < N - 1; ++i) src += src_cum[i] <= x ? 1 : 0;
    }
    {
        orlg_lds_cf64 *row = dst_cum + src * N;
        const double x = u[3] * (row[N - 1] + 0.0);
        for (int i = 0; i < N - 1; ++i) dst += row[i] <= x ? 1 : 0;
    }
    const double o_iat = lane < n ? iat : 0.0, o_ht = lane < n ? ht : 0.0;
    const uint32_t o_rq = lane < n ? ((uint32_t)src | ((uint32_t)dst << 8) | ((uint32_t)my_r << 16)) : 0u;
    if (RING_LDS) {
        ((orlg_lds_f64 *)ring_iat_v)[lane] = o_iat; ((orlg_lds_f64 *)ring_ht_v)[lane] = o_ht; ((orlg_lds_u32 *)ring_req_v)[lane] = o_rq;
    } else {
        ((orlg_glb_f64 *)ring_iat_v)[lane] = o_iat; ((orlg_glb_f64 *)ring_ht_v)[lane] = o_ht; ((orlg_glb_u32 *)ring_req_v)[lane] = o_rq;
    }
    wave_sync();
    return n | (idx << 8);
}
template <bool RING_LDS>
DEV int refill_requests_cont_t(uint32_t *mt, double *ring_iat, double *ring_ht, uint32_t *ring_req, const double *src_cum,
                               const double *dst_cum, int *idx_io, int N, int width, double lam_arrival, double lam_holding) {
    const int r = refill_requests_cont_as<RING_LDS>((orlg_lds_u32 *)mt, ring_iat, ring_ht, ring_req, (orlg_lds_cf64 *)src_cum,
                                                    (orlg_lds_cf64 *)dst_cum, *idx_io, N, width, lam_arrival, lam_holding);
    *idx_io = r >> 8;
    return r & 0xff;
}

// ---------------------------------------------------------------------------------------- first fit
// x[w]: wave-uniform free bitmap of one path (AND over its links).  Lane l of word w owns slot 64w+l
// and computes the length of the free run starting there.
template <int W>
DEV void ext_chain(const u64 (&x)[W], int (&ext)[W]) {
    ext[W - 1] = 0;
#pragma unroll
    for (int w = W - 2; w >= 0; --w) ext[w] = (x[w + 1] == ~0ull) ? 64 + ext[w + 1] : ctz64(~x[w + 1]);
}

// v_ffbl_b32 as the hardware defines it: index of the lowest set bit, 0xffffffff for 0 (__builtin_ctz is undefined there)
DEV uint32_t ffbl_hw(uint32_t v) {
    uint32_t r;
    asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
// length of the free run that starts at this lane's slot: t = (~word) >> lane has its lowest set bit at the first used slot at
// or after the lane; none left in the word (t == 0) -> the run reaches the word's end and goes on for `rest - (64 - lane)`
// slots in the next words.  ffbl(0) = 0xffffffff keeps the "none" case out of both minima without a select.
DEV int free_run_length(u64 t, int rest /* (64 - lane) + extension into the next words */) {
    const uint32_t a = ffbl_hw((uint32_t)t), b = ffbl_hw((uint32_t)(t >> 32)) | 32u;
    const uint32_t c = a < b ? a : b;
    return (int)(c < (uint32_t)rest ? c : (uint32_t)rest);
}

// smallest s in [0, limit) with slots [s, s+n) free, or -1 (rmsa_env.py:860-871, 908-913)
template <int W>
DEV int first_fit(const u64 (&x)[W], int n, int limit, int lane) {
    if (limit <= 0) return -1;
    int ext[W];
    ext_chain<W>(x, ext);
    const int to_end = 64 - lane;
#pragma unroll
    for (int w = 0; w < W; ++w) {
        if (x[w] != 0ull && 64 * w < limit) {
            const int len = free_run_length((~x[w]) >> lane, to_end + ext[w]);
            // start slots below the limit: a wave-uniform lane mask, no per-lane compare
            const int below = limit - 64 * w;
            const u64 ok = below >= 64 ? ~0ull : ((1ull << below) - 1ull);
            const u64 m = ballot(len >= n) & ok;
            if (m) return 64 * w + ctz64(m);
        }
    }
    return -1;
}

// b-th (0-based) free run with length >= n (rmsa_env.py:774-804); returns start or -1, *len_out = its length
template <int W>
DEV int find_block(const u64 (&x)[W], int n, int b, int lane, int *len_out) {
    int ext[W];
    ext_chain<W>(x, ext);
#pragma unroll
    for (int w = 0; w < W; ++w) {
        if (x[w] != 0ull) {
            u64 carry = w > 0 ? (x[w > 0 ? w - 1 : 0] >> 63) : 0ull;
            u64 starts = x[w] & ~((x[w] << 1) | carry);
            int len = free_run_length((~x[w]) >> lane, (64 - lane) + ext[w]);
            u64 m = ballot(len >= n) & starts;   // run starts: a wave-uniform lane mask
            int cnt = popc64(m);
            if (b < cnt) {
                for (int q = 0; q < b; ++q) m &= m - 1;
                int l = ctz64(m);
                *len_out = __builtin_amdgcn_readlane(len, l);
                return 64 * w + l;
            }
            b -= cnt;
        }
    }
    return -1;
}

// is_path_free (rmsa_env.py:721-734) on a path-wide mask
template <int W>
DEV bool window_free(const u64 (&x)[W], int s, int n, int S) {
    if (s + n > S) return false;
    bool ok = true;
#pragma unroll
    for (int w = 0; w < W; ++w) {
        u64 m = window_mask(s, n, w);
        ok = ok && ((x[w] & m) == m);
    }
    return ok;
}

// AND of the link bitmaps of path record `gid` for word w (get_available_slots, rmsa_env.py:745-756).
// `active` lanes hold a valid (gid, w); the hop loop is fully unrolled over the 16-byte record with
// compile-time byte positions (v_bfe_u32 + v_mad_u32_u24 per hop) and leaves as soon as no lane has hops left.
template <int W>
DEV u64 path_word(const u64 *occ, const OrlgPathRec *recs, int gid, int w, bool active) {
    uint4 r = make_uint4(0u, 0u, 0u, 0u);
    if (active) r = *reinterpret_cast<const uint4 *>(recs + gid);
    const uint32_t q[4] = {r.x, r.y, r.z, r.w};
    const int hops = (int)(r.x & 0xffu);  // 0 on inactive lanes
    u64 acc = active ? ~0ull : 0ull;
#pragma unroll
    for (int h = 0; h < ORLG_MAX_HOPS; ++h) {
        if (ballot(h < hops) == 0ull) break;
        const int link = (int)((q[(h + 2) >> 2] >> (8 * ((h + 2) & 3))) & 0xffu);
        if (h < hops) acc &= occ[__mul24(link, W) + w];
    }
    return acc;
}

// numpy's float64 add.reduce order (pairwise_sum in numpy/core/src/umath/loops_utils.h.src: 8 running
// accumulators, fixed combination tree, blocks of <= 128) so that np.mean(...) in the info dict is reproduced
// bit for bit; n <= 255 here (one link per element).
DEV double np_pairwise_block(const double *a, int n) {
    if (n < 8) {
        double res = 0.;
        for (int i = 0; i < n; i++) res += a[i];
        return res;
    }
    double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
        r0 += a[i]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3];
        r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7];
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; i++) res += a[i];
    return res;
}
DEV double np_pairwise_256(const double *a, int n) {  // n <= 256: at most one split
    if (n <= 128) return np_pairwise_block(a, n);
    int n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_block(a, n2) + np_pairwise_block(a + n2, n - n2);
}
DEV double np_mean(const double *a, int n) {  // n <= 512: at most two levels of splitting
    double s;
    if (n <= 128) {
        s = np_pairwise_block(a, n);
    } else {
        int n2 = n / 2;
        n2 -= n2 % 8;
        s = np_pairwise_256(a, n2) + np_pairwise_256(a + n2, n - n2);
    }
    return s / (double)n;
}

// _get_network_compactness (rmsa_env.py:844-851) from the maintained integer sums
DEV double network_compactness(int sum_span, int sum_slots_hops, int sum_gaps, int E) {
    if (sum_gaps > 0)
        return ORLG_FDIV((double)sum_span, (double)sum_slots_hops) * ORLG_FDIV((double)E, (double)sum_gaps);
    return 1.0;
}

// kernel parameters re-read from the kernarg segment through an opaque pointer: lets the compiler drop
// rarely used pointers from SGPRs across the step loop instead of spilling them
typedef const OrlgParams __attribute__((address_space(4))) *KernargParams;
DEV KernargParams kernarg_params() {
    auto k = __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(k));
    return (KernargParams)k;
}

// ---------------------------------------------------------------------------------------- link statistics
// Rebuild, for a list of links, the integer run statistics of the link's free bitmap and (LINKF) the
// time-weighted floats of _update_link_stats (rmsa_env.py:562-641).  Maintains the per-link (span, gaps)
// cache whose sums give _get_network_compactness (rmsa_env.py:806-851):
//     span = lambda_max - lambda_min, gaps = free runs inside the used span = used runs - 1
// for links with more than one used run, 0 otherwise.  links == nullptr means links 0..nlinks-1.
// GRAPH: after the last chunk also perform _update_network_stats (rmsa_env.py:537-560) -- its two
// time-weighted averages ride on lanes 62 / 63 of the same fp64 instruction stream as the links'.
// reductions over the 8 lanes of a link group (lane = link slot * 8 + word) with DPP lane permutations: xor 1 and xor 2
// inside a quad, then the mirrored half row brings in the other quad's total -- every lane ends with the group's result
DEV int dpp_xor1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false); }   // quad_perm [1,0,3,2]
DEV int dpp_xor2(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false); }   // quad_perm [2,3,0,1]
DEV int dpp_half_mirror(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false); }
// neighbours inside a row of 16 lanes: the previous lane (row_shr:1), the next lane (row_shl:1), K lanes ahead (row_shl:K);
// 0 where the row ends
DEV int lane_prev_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true); }
DEV int lane_next_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x101, 0xf, 0xf, true); }
template <int K>
DEV int lane_ahead_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x100 + K, 0xf, 0xf, true); }
template <int K>
DEV int lane_back_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x110 + K, 0xf, 0xf, true); }   // K lanes back (row_shr:K)
DEV u64 lane_prev_u64(u64 v) {
    const uint32_t lo = (uint32_t)lane_prev_i32((int)(uint32_t)v), hi = (uint32_t)lane_prev_i32((int)(uint32_t)(v >> 32));
    return ((u64)hi << 32) | lo;
}
DEV int group8_add(int v) { v += dpp_xor1(v); v += dpp_xor2(v); v += dpp_half_mirror(v); return v; }
DEV int group8_min(int v) {
    int o = dpp_xor1(v); v = o < v ? o : v;
    o = dpp_xor2(v); v = o < v ? o : v;
    o = dpp_half_mirror(v); return o < v ? o : v;
}
DEV int group8_max(int v) {
    int o = dpp_xor1(v); v = o > v ? o : v;
    o = dpp_xor2(v); v = o > v ? o : v;
    o = dpp_half_mirror(v); return o > v ? o : v;
}

// whole-wave reductions without LDS crossbar round trips: full-mask DPP permutations inside a row of 16 lanes (quad, half row,
// row: every lane of a row ends with the row's result), then the four row results are read with v_readlane and combined as
// wave-uniform values.  (The row-broadcast DPP modes with a partial row mask are avoided on purpose: whether the masked-off
// lanes keep the right value depends on how the compiler folds the move into the ALU op.)
DEV int dpp_row_mirror(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false); }
DEV int wave_max_i32(int v) {
    int o = dpp_xor1(v); v = o > v ? o : v;
    o = dpp_xor2(v); v = o > v ? o : v;
    o = dpp_half_mirror(v); v = o > v ? o : v;
    o = dpp_row_mirror(v); v = o > v ? o : v;
    const int r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
    const int r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
    const int a = r0 > r1 ? r0 : r1, b = r2 > r3 ? r2 : r3;
    return a > b ? a : b;
}
DEV int wave_add_i32(int v) {
    v += dpp_xor1(v); v += dpp_xor2(v); v += dpp_half_mirror(v); v += dpp_row_mirror(v);
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
           __builtin_amdgcn_readlane(v, 48);
}
#define ORLG_DPP_F64(fn, x) __hiloint2double(fn(__double2hiint(x)), fn(__double2loint(x)))
DEV double wave_max_f64(double v) {
    double o = ORLG_DPP_F64(dpp_xor1, v); v = o > v ? o : v;
    o = ORLG_DPP_F64(dpp_xor2, v); v = o > v ? o : v;
    o = ORLG_DPP_F64(dpp_half_mirror, v); v = o > v ? o : v;
    o = ORLG_DPP_F64(dpp_row_mirror, v); v = o > v ? o : v;
    const double r0 = readlane_d(v, 0), r1 = readlane_d(v, 16), r2 = readlane_d(v, 32), r3 = readlane_d(v, 48);
    const double a = r0 > r1 ? r0 : r1, b = r2 > r3 ? r2 : r3;
    return a > b ? a : b;
}
DEV double wave_add_f64(double v) {   // the association order differs from a sequential sum: only for tolerance-based results
    v += ORLG_DPP_F64(dpp_xor1, v); v += ORLG_DPP_F64(dpp_xor2, v);
    v += ORLG_DPP_F64(dpp_half_mirror, v); v += ORLG_DPP_F64(dpp_row_mirror, v);
    return (readlane_d(v, 0) + readlane_d(v, 16)) + (readlane_d(v, 32) + readlane_d(v, 48));
}

DEV u64 lane_next_u64(u64 v) {
    const uint32_t lo = (uint32_t)lane_next_i32((int)(uint32_t)v), hi = (uint32_t)lane_next_i32((int)(uint32_t)(v >> 32));
    return ((u64)hi << 32) | lo;
}
DEV int wave_min_i32(int v) {
    int o = dpp_xor1(v); v = o < v ? o : v;
    o = dpp_xor2(v); v = o < v ? o : v;
    o = dpp_half_mirror(v); v = o < v ? o : v;
    o = dpp_row_mirror(v); v = o < v ? o : v;
    const int r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
    const int r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
    const int a = r0 < r1 ? r0 : r1, b = r2 < r3 ? r2 : r3;
    return a < b ? a : b;
}

// path_word that also hands back the record's spectral efficiency and hop count (0 on inactive lanes)
template <int W>
DEV u64 path_word_rec(const u64 *occ, const OrlgPathRec *recs, int gid, int w, bool active, int &se, int &hops_out) {
    uint4 r = make_uint4(0u, 0u, 0u, 0u);
    if (active) r = *reinterpret_cast<const uint4 *>(recs + gid);
    const uint32_t q[4] = {r.x, r.y, r.z, r.w};
    const int hops = (int)(r.x & 0xffu);
    se = (int)((r.x >> 8) & 0xffu);
    hops_out = hops;
    u64 acc = active ? ~0ull : 0ull;
#pragma unroll
    for (int h = 0; h < ORLG_MAX_HOPS; ++h) {
        if (ballot(h < hops) == 0ull) break;
        const int link = (int)((q[(h + 2) >> 2] >> (8 * ((h + 2) & 3))) & 0xffu);
        if (h < hops) acc &= occ[__mul24(link, W) + w];
    }
    return acc;
}

// Bits b of word w such that slots [64 w + b, 64 w + b + n) are all free, for a bitmap whose W words sit on W consecutive lanes
// of a row (w = the lane's word; `x` = 0 on lanes that hold nothing).  r_m = AND of x >> 0 .. x >> (m - 1) is doubled:
// r_{m+k} = r_m & (r_m >> k) for k <= m; k <= 31, so that a shift is two 32-bit funnel shifts (v_alignbit_b32) fed by the next
// word's low half (one DPP read).  n may differ between the rows (and between the paths inside a row): the loop runs to the
// longest, a finished lane shifts by k = 0, which leaves it as it is -- no predication.
template <int W>
DEV u64 run_starts(u64 x, int n, int w) {
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    const uint32_t keep = w == W - 1 ? 0u : ~0u;  // nothing beyond the last word
    int have = 1;
    for (;;) {
        int k = n - have;
        k = k < have ? k : have;
        k = k < 31 ? k : 31;
        if (ballot(k > 0) == 0ull) break;
        const uint32_t nlo = (uint32_t)lane_next_i32((int)lo) & keep;
        const uint32_t slo = __builtin_amdgcn_alignbit(hi, lo, (uint32_t)k), shi = __builtin_amdgcn_alignbit(nlo, hi, (uint32_t)k);
        lo &= slo; hi &= shi;
        have += k;
    }
    return ((u64)hi << 32) | lo;
}

#define ORLG_LLOG_CAP 64      // logged updates per (environment, link) between two replays (6-bit count)
#define ORLG_LLOG_FLUSH 40    // a link that reaches this many asks for a replay
// The logged link updates (DEFER instantiations), worked off: lane gl of a row of GL lanes = link gl (+ GL, ...) of the row's
// environment (GL = 16: four environments per wave, GL = 64: one); every lane runs through its link's entries in their order with the link's four statistics in registers -- the
// float64 operations of _update_link_stats (rmsa_env.py:562-641) as link_stats_update / group_link_stats do them, one update after the other.
template <int GL>
DEV void link_replay(const int lane, double *lst, int32_t *lint, const Tab &tb, int S, int E, const uint4 *llog) {
    const int gl = lane & (GL - 1);
    // the entries other lanes of this wave logged: the stores only have to be complete (same CU)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    for (int l0 = 0; l0 < E; l0 += GL) {
        const int link = l0 + gl;
        const bool on = link < E;
        const int li = on ? lint[link] : 0;
        const int n = (int)((uint32_t)li >> 26);
        if (ballot(n > 0) == 0ull) continue;
        double s_util = 0.0, s_ef = 0.0, s_c = 0.0, s_lu = 0.0;
        if (on && n > 0) { s_util = lst[link]; s_ef = lst[E + link]; s_c = lst[2 * E + link]; s_lu = lst[3 * E + link]; }
        const uint4 *row = llog + __mul24(on ? link : 0, ORLG_LLOG_CAP);
        const int nmax = wave_max_i32(n);
        uint4 e_nx = make_uint4(0u, 0u, 0u, 0u);
        if (n > 0) e_nx = row[0];
        for (int k = 0; k < nmax; ++k) {
            const uint4 ev = e_nx;
            if (k + 1 < n) e_nx = row[k + 1];   // (the next entry is requested before this one is worked on)
            if (k < n) {
                const int freec = (int)(ev.x & 0x3ffu), max_empty = (int)((ev.x >> 10) & 0x3ffu), span = (int)(ev.x >> 20), U = (int)ev.y;
                const double now = __hiloint2double((int)ev.w, (int)ev.z);
                if (now > 0) {
                    const double ynow = recip_refine(now);
                    const double cur0 = tb.div_s[S - freec];  // (S - free) / S
                    double cur1 = 0.0, cur2 = 0.0;
                    if (freec > 0) {
                        cur1 = 1.0 - ORLG_FDIV((double)max_empty, (double)freec);
                        cur2 = U > 1 ? ORLG_FDIV((double)span, (double)(S - freec)) * tb.inv_k[U] : 1.0;
                    }
                    const double time_diff = now - s_lu;
                    s_util = div_by((s_util * s_lu) + (cur0 * time_diff), now, ynow);
                    s_ef = div_by((s_ef * s_lu) + (cur1 * time_diff), now, ynow);
                    s_c = div_by((s_c * s_lu) + (cur2 * time_diff), now, ynow);
                }
                s_lu = now;
            }
        }
        if (on && n > 0) {
            lst[link] = s_util; lst[E + link] = s_ef; lst[2 * E + link] = s_c; lst[3 * E + link] = s_lu;
            lint[link] = li & 0x03ffffff;
        }
    }
    wave_sync();
}

template <int W, bool LINKF, bool GRAPH, bool DEFER = false>
DEV void link_stats_update(Wave &wv, const Tab &tb, int S, int E, const uint8_t *links, int nlinks, double now,
                           int &sum_span, int &sum_gaps, double &comp_cur, int sum_sh, double cur_thr, uint4 *llog = nullptr) {
    // DEFER (see group_link_stats, orlg_group_kernels.hip): the links' float64 recurrences are not done here -- what they consume is
    // logged per link and link_replay works the logs off with one link per lane; the graph statistics stay (one chain per environment)
    static_assert(W <= 8, "a link group is 8 lanes");
    bool need_replay = false;
    constexpr int HPC = 8;  // links per chunk: lane = link slot * 8 + word
    const int lane = wv.lane;
    const int hl = lane >> 3, w = lane & 7;
    double ynow = 0.0;
    if ((LINKF || GRAPH) && now > 0) ynow = recip_refine(now);
    for (int h0 = 0; h0 < nlinks; h0 += HPC) {
        const int nl = nlinks - h0 < HPC ? nlinks - h0 : HPC;
        const bool last_chunk = h0 + HPC >= nlinks;
        // ---- per (link, word) lane: the word's run statistics ...
        int link = 0, packed = 0, lo = 0x7fff, hi = 0, ml = 0;
        if (hl < nl) link = links ? (int)links[h0 + hl] : h0 + hl;
        // the link's words sit on consecutive lanes of its group: the neighbouring words arrive by DPP, not by further LDS
        // reads (every DPP read stands outside any condition: a lane switched off by a branch is not a readable source)
        u64 x = 0ull;
        if (hl < nl && w < W) x = wv.occ[__mul24(link, W) + w];
        const u64 prev = lane_prev_u64(x);
        int e = 0;  // free slots that continue a run reaching this word's end into the next words
        if (LINKF) {
            const int lead = x == ~0ull ? 64 : ctz64(~x);  // free slots at the word's start
            const int nlead_raw = lane_next_i32(lead);
            const int nlead = w < W - 1 ? nlead_raw : 0;
            e = nlead;
#pragma unroll
            for (int i = 0; i < W - 2; ++i) {
                const int ne_raw = lane_next_i32(e);
                const int ne = w < W - 1 ? ne_raw : 0;
                e = nlead == 64 ? 64 + ne : nlead;
            }
        }
        const bool first_free = x & 1ull;  // meaningful on the link's first lane
        // slot S - 1 sits in word (S - 1) >> 6 -- not always the last of the W words (S = 400 runs on the 8-word layout)
        const int lw = (S - 1) >> 6;
        const int last_free_bit = w == lw ? (int)((x >> ((S - 1) & 63)) & 1ull) : 0;
        bool last_free;
        if (lw == W - 1) last_free = W == 1 ? last_free_bit != 0 : lane_ahead_i32<(W > 1 ? W - 1 : 1)>(last_free_bit) != 0;  // wave-uniform branch
        else last_free = group8_max(last_free_bit) != 0;
        if (hl < nl && w < W) {
            u64 u = ~x & valid_mask(S, w);
            u64 carry_f = w > 0 ? (prev >> 63) : 0ull;
            u64 carry_u = w > 0 ? ((~prev) >> 63) : 0ull;
            u64 fstarts = x & ~((x << 1) | carry_f);
            u64 ustarts = u & ~((u << 1) | carry_u);
            packed = popc64(x) | (popc64(fstarts) << 10) | (popc64(ustarts) << 20);  // free slots, free runs, used runs
            lo = u ? 64 * w + ctz64(u) : 0x7fff;
            hi = u ? 64 * w + 64 - clz64(u) : 0;
            if (LINKF) {
                u64 st = fstarts;
                while (st) {
                    int b = ctz64(st);
                    st &= st - 1;
                    int len = free_run_length((~x) >> b, 64 - b + e);
                    ml = len > ml ? len : ml;
                }
            }
        }
        // ---- ... combined over the link's words inside its 8-lane group (no LDS round trip)
        packed = group8_add(packed);
        const int lmin = group8_min(lo), lmax = group8_max(hi);
        if (LINKF) ml = group8_max(ml);
        const int freec = packed & 0x3ff, F = (packed >> 10) & 0x3ff, U = packed >> 20;
        const bool link_lane = hl < nl && w == 0;  // one lane per link carries on
        int dspan = 0, dgaps = 0;
        if (link_lane) {
            int nspan = U > 1 ? lmax - lmin : 0, ngaps = U > 1 ? U - 1 : 0;
            int old = wv.lint[link];
            int cnt = 0;
            if (LINKF && DEFER) {
                cnt = (int)((uint32_t)old >> 26);
                const int max_empty = (F > 1 && !(F == 2 && first_free && last_free)) ? ml : 0;
                if (cnt < ORLG_LLOG_CAP - 1) {
                    llog[__mul24(link, ORLG_LLOG_CAP) + cnt] =
                        make_uint4((uint32_t)freec | ((uint32_t)max_empty << 10) | ((uint32_t)(lmax - lmin) << 20), (uint32_t)U,
                                   (uint32_t)__double2loint(now), (uint32_t)__double2hiint(now));
                    cnt += 1;
                }
                if (cnt >= ORLG_LLOG_FLUSH) need_replay = true;
                old &= 0x03ffffff;
            }
            wv.lint[link] = nspan | (ngaps << 16) | (cnt << 26);
            dspan = nspan - (old & 0xffff);
            dgaps = ngaps - (old >> 16);
        }
        for (int q = 0; q < nl; ++q) {
            sum_span += __builtin_amdgcn_readlane(dspan, q * 8);
            sum_gaps += __builtin_amdgcn_readlane(dgaps, q * 8);
        }
        const bool graph_now = GRAPH && last_chunk;
        if (graph_now) comp_cur = network_compactness(sum_span, sum_sh, sum_gaps, E);
        // ---- floats: links on their group's first lane, graph throughput / compactness on lanes 62 / 63
        if (((LINKF && !DEFER) || graph_now) && now > 0) {
            const bool is_link = LINKF && !DEFER && link_lane;
            const bool is_graph = graph_now && lane >= 62;
            if (is_link || is_graph) {
                double *l_util = wv.lst, *l_ef = wv.lst + E, *l_c = wv.lst + 2 * E, *l_lu = wv.lst + 3 * E;
                double last_update, last0, cur0;
                double last1 = 0.0, last2 = 0.0, cur1 = 0.0, cur2 = 0.0;
                if (is_link) {
                    last_update = l_lu[link];
                    last0 = l_util[link]; last1 = l_ef[link]; last2 = l_c[link];
                    cur0 = tb.div_s[S - freec];  // (S - free) / S
                    if (freec > 0) {
                        int max_empty = (F > 1 && !(F == 2 && first_free && last_free)) ? ml : 0;
                        cur1 = 1.0 - ORLG_FDIV((double)max_empty, (double)freec);
                        cur2 = U > 1 ? ORLG_FDIV((double)(lmax - lmin), (double)(S - freec)) * tb.inv_k[U] : 1.0;
                    }
                } else {
                    last_update = wv.wsc->g_lu;
                    last0 = lane == 62 ? wv.wsc->g_thr : wv.wsc->g_comp;
                    cur0 = lane == 62 ? cur_thr : comp_cur;
                }
                double time_diff = now - last_update;
                double n0 = div_by((last0 * last_update) + (cur0 * time_diff), now, ynow);
                if (is_link) {
                    double n1 = div_by((last1 * last_update) + (cur1 * time_diff), now, ynow);
                    double n2 = div_by((last2 * last_update) + (cur2 * time_diff), now, ynow);
                    l_util[link] = n0; l_ef[link] = n1; l_c[link] = n2;
                } else if (lane == 62) {
                    wv.wsc->g_thr = n0;
                } else {
                    wv.wsc->g_comp = n0;
                }
            }
        }
        if (LINKF && !DEFER && link_lane) wv.lst[3 * E + link] = now;
        wave_sync();
        if (graph_now && lane == 0) wv.wsc->g_lu = now;
        wave_sync();
    }
    if (DEFER && ballot(need_replay) != 0ull) link_replay<64>(lane, wv.lst, wv.lint, tb, S, E, llog);
}

// set (release) or clear (provision) the window [s, s+n) on every link of a path
template <int W>
DEV void apply_window(Wave &wv, const uint8_t *links, int hops, int s, int n, bool set_free) {
    constexpr int HPC = 64 / W;
    const int hl = wv.lane / W, w = wv.lane - hl * W;
    u64 m = window_mask(s, n, w);
    for (int h0 = 0; h0 < hops; h0 += HPC) {
        int h = h0 + hl;
        if (hl < HPC && h < hops && m) {
            u64 *word = wv.occ + __mul24((int)links[h], W) + w;
            *word = set_free ? (*word | m) : (*word & ~m);
        }
    }
    wave_sync();
}

// ---------------------------------------------------------------------------------------- section timing (debug builds)
// -DORLG_SECTIONS: wall cycles each wave spends in the sections of a step, summed over all waves into orlg_sections[]
// (tools/section_profile.py).  Not compiled into the product library.
#ifdef ORLG_SECTIONS
__device__ unsigned long long orlg_sections[16];
#define SEC_DECL_G __shared__ unsigned long long sec_acc[16][16]; long long sec_t0 = 0; int sec_cur = 0; \
    if (lane < 16) sec_acc[wib][lane] = 0ull; wave_sync(); sec_t0 = __builtin_readcyclecounter();
#define SEC_DECL __shared__ unsigned long long sec_acc[ORLG_MAX_WAVES_PER_BLOCK][16]; long long sec_t0 = 0; int sec_cur = 0; \
    if (lane < 16) sec_acc[wib][lane] = 0ull; wave_sync(); sec_t0 = __builtin_readcyclecounter();
#define SEC(i) do { const long long sec_n = __builtin_readcyclecounter(); if (lane == 0) sec_acc[wib][sec_cur] += (unsigned long long)(sec_n - sec_t0); \
    sec_t0 = sec_n; sec_cur = (i); } while (0)
#define SEC_FLUSH do { SEC(0); wave_sync(); if (lane < 16) atomicAdd(&orlg_sections[lane], sec_acc[wib][lane]); } while (0)
// device functions that time their own sub-sections take the kernel's accumulators along
#define SEC_PARAMS , unsigned long long (*sec_acc)[16], long long &sec_t0, int &sec_cur, int wib
#define SEC_ARGS , sec_acc, sec_t0, sec_cur, wib
#else
#define SEC_PARAMS
#define SEC_ARGS
#define SEC_DECL
#define SEC_DECL_G
#define SEC(i) do { } while (0)
#define SEC_FLUSH do { } while (0)
#endif

// ---------------------------------------------------------------------------------------- the step kernel
// STEPK: the step kernel proper (mode STEP); the reset kernel (modes INIT / EPISODE_RESET: reset(), rmsa_env.py:343-457) is the
// same body without the policy / provisioning part, under its own name so that kernel statistics keep the two apart.
// FF: an instantiation that only knows the first-fit policies (shortest path / shortest available path, k <= 8): the other
// policies' code -- and the registers it pins -- is gone from the kernel the headline workload runs.
template <int W, int STATS, bool STEPK, bool FF = false, bool DEFER = false>
DEV void rmsa_body(const OrlgParams &p) {
    extern __shared__ __align__(16) unsigned char smem[];
#ifdef ORLG_SHAPE_ASSUME
    ORLG_SHAPE_ASSUME  // experiment (tools/shape_experiment.py): shape and layout fields as compile-time constants
#endif
    stage_tables(smem, p);
    const int lane = threadIdx.x & 63;
    const int wib = uni((int)(threadIdx.x >> 6));
    const Tab tb = make_tab(smem, p);
    unsigned char *wb = smem + p.l_shared_bytes + (size_t)wib * p.l_wave_bytes;
    Wave wv;
    wv.lane = lane;
    wv.occ = reinterpret_cast<u64 *>(wb + p.l_occ);
    wv.qtime = reinterpret_cast<double *>(wb + p.l_qtime);
    wv.qdesc = reinterpret_cast<uint32_t *>(wb + p.l_qdesc);
    wv.mt = reinterpret_cast<uint32_t *>(wb + p.l_mt);
    wv.lst = reinterpret_cast<double *>(wb + p.l_lstat);
    wv.hist = reinterpret_cast<int32_t *>(wb + p.l_hist);
    wv.lint = reinterpret_cast<int32_t *>(wb + p.l_lint);
    wv.scratch = reinterpret_cast<uint32_t *>(wb + p.l_scratch);
    wv.wsc = reinterpret_cast<OrlgWaveScalars *>(wb + p.l_wsc);
    wv.ring_iat = reinterpret_cast<double *>(wb + p.l_ring);
    wv.ring_ht = wv.ring_iat + ORLG_RING;
    wv.ring_req = reinterpret_cast<uint32_t *>(wv.ring_ht + ORLG_RING);

    const int E = p.E, S = p.S, K = p.K, N = p.N, NBR = p.NBR, Q = p.Q, NW = p.NW;
    const int LS = (FF || K <= 8) ? 8 : W;  // lanes from one candidate path to the next in the policy's (path, word) layout
    constexpr bool NET = STATS >= 1;
    constexpr bool FULL = STATS >= 2;
    SEC_DECL

    // ------------------------------------------------------------------ work queue: one environment at a time per wave
    // (a wave that finishes early takes the next environment instead of idling until its workgroup drains; every
    // wave leaves through the same exit: the first ticket at or beyond B)
    // The first environment of a wave is its own index (no atomic: all waves start at once); the remaining B - waves
    // environments are handed out by the ticket counter.
    const int n_static = (int)(gridDim.x * (blockDim.x >> 6)) < p.B ? (int)(gridDim.x * (blockDim.x >> 6)) : p.B;
    int env = (int)(blockIdx.x * (blockDim.x >> 6)) + wib;
    if (env >= p.B) return;
    // Long launches balance the load with the ticket counter (environments differ in work per step and a launch runs hundreds of
    // steps); short ones -- the agent-driven pattern, a launch per step -- stride statically over the environments: tens of
    // thousands of draws on one address would cost more than the steps themselves.  p.ticket_stride = 0 selects tickets.
    uint32_t nxt_tk = 0;  // lane 0: the ticket drawn for this wave's next environment
    const int n_waves = (int)(gridDim.x * (blockDim.x >> 6));
    for (bool first = true;; first = false) {
    if (!first) {
        if (p.ticket_stride) {
            env += n_waves;
            if (env >= p.B) break;
        } else {
            const uint32_t tk = (uint32_t)__builtin_amdgcn_readfirstlane((int)nxt_tk) - p.ticket_base;
            if (tk >= (uint32_t)(p.B - n_static)) break;
            env = n_static + (int)tk;
        }
    }
    SEC(1);  // state load
    // the next ticket is drawn now and looked at after this environment is done: its round trip is off the critical path
    if (!p.ticket_stride && lane == 0) nxt_tk = atomicAdd(kernarg_params()->ticket, 1u);
    // ------------------------------------------------------------------ HBM -> LDS (coalesced, 16 B per lane)
    const int n_iter = STEPK ? p.n_steps : 1;
    // The MT19937 state (2.5 KB) is only needed when the ring of pre-generated arrivals runs dry: it is fetched then.  A
    // launch of a few steps (the agent-driven pattern: one launch per step) does not stage the ring either: it reads the
    // entries it consumes straight from HBM -- the first one is requested as soon as the scalars are in -- and falls back to
    // the LDS ring after a refill.
    const bool direct_ring = n_iter <= ORLG_DIRECT_STEPS;
    bool mt_loaded = false, ring_dirty = false, ring_in_lds = !direct_ring;
    {
        // the scalar record and every array's first 1 KiB block are requested before anything is written to LDS: one HBM
        // round trip for the whole state, not one per array
        const uint4 *g_sc = reinterpret_cast<const uint4 *>(p.scal + env);
        const uint4 *g_occ = reinterpret_cast<const uint4 *>(p.occ + (size_t)env * NW);
        const uint4 *g_qt = reinterpret_cast<const uint4 *>(p.qtime + (size_t)env * Q);
        const uint4 *g_qd = reinterpret_cast<const uint4 *>(p.qdesc + (size_t)env * Q);
        const uint4 *g_ls = reinterpret_cast<const uint4 *>(p.lstat + (size_t)env * 4 * E);
        const uint4 *g_hi = reinterpret_cast<const uint4 *>(p.hist + (size_t)env * 4 * NBR);
        const uint4 *g_li = reinterpret_cast<const uint4 *>(p.lint + (size_t)env * p.lint_stride);
        const bool occ16 = (NW & 1) == 0;
        const int n_occ = occ16 ? NW >> 1 : 0, n_qt = Q >> 1, n_qd = Q >> 2, n_ls = FULL ? 2 * E : 0, n_hi = NBR;
        const int n_li = NET ? p.lint_stride >> 2 : 0;
        constexpr int n_sc = (int)sizeof(OrlgEnvScalars) / 16;
        uint4 a_sc = make_uint4(0, 0, 0, 0), a_occ = a_sc, a_qt = a_sc, a_qd = a_sc, a_ls = a_sc, a_hi = a_sc, a_li = a_sc;
        if (lane < n_sc) a_sc = g_sc[lane];
        if (lane < n_occ) a_occ = g_occ[lane];
        if (lane < n_qt) a_qt = g_qt[lane];
        if (lane < n_qd) a_qd = g_qd[lane];
        if (lane < n_ls) a_ls = g_ls[lane];
        if (lane < n_hi) a_hi = g_hi[lane];
        if (lane < n_li) a_li = g_li[lane];
        if (lane < n_sc) reinterpret_cast<uint4 *>(wv.scratch)[lane] = a_sc;
        if (lane < n_occ) reinterpret_cast<uint4 *>(wv.occ)[lane] = a_occ;
        if (lane < n_qt) reinterpret_cast<uint4 *>(wv.qtime)[lane] = a_qt;
        if (lane < n_qd) reinterpret_cast<uint4 *>(wv.qdesc)[lane] = a_qd;
        if (lane < n_ls) reinterpret_cast<uint4 *>(wv.lst)[lane] = a_ls;
        if (lane < n_hi) reinterpret_cast<uint4 *>(wv.hist)[lane] = a_hi;
        if (lane < n_li) reinterpret_cast<uint4 *>(wv.lint)[lane] = a_li;
        for (int i = lane + 64; i < n_li; i += 64) reinterpret_cast<uint4 *>(wv.lint)[i] = g_li[i];
        for (int i = lane + 64; i < n_occ; i += 64) reinterpret_cast<uint4 *>(wv.occ)[i] = g_occ[i];
        for (int i = lane + 64; i < n_qt; i += 64) reinterpret_cast<uint4 *>(wv.qtime)[i] = g_qt[i];
        for (int i = lane + 64; i < n_qd; i += 64) reinterpret_cast<uint4 *>(wv.qdesc)[i] = g_qd[i];
        for (int i = lane + 64; i < n_ls; i += 64) reinterpret_cast<uint4 *>(wv.lst)[i] = g_ls[i];
        for (int i = lane + 64; i < n_hi; i += 64) reinterpret_cast<uint4 *>(wv.hist)[i] = g_hi[i];
        if (!occ16) {
            const u64 *g = p.occ + (size_t)env * NW;
            for (int i = lane; i < NW; i += 64) wv.occ[i] = g[i];
        }
        if (!direct_ring) {
            wv.ring_iat[lane] = p.ring_iat[(size_t)env * ORLG_RING + lane];
            wv.ring_ht[lane] = p.ring_ht[(size_t)env * ORLG_RING + lane];
            wv.ring_req[lane] = p.ring_req[(size_t)env * ORLG_RING + lane];
        }
    }
    wave_sync();
    // wave-uniform working copies (the scalar record was staged through the scratch row)
    const OrlgEnvScalars *gs = reinterpret_cast<const OrlgEnvScalars *>(wv.scratch);
    int ring_pos = gs->ring_pos, ring_cnt = gs->ring_cnt;
    double pf_iat = 0.0, pf_ht = 0.0;   // direct mode: the ring entry of the launch's first arrival, requested early
    uint32_t pf_rq = 0;
    if (direct_ring && ring_cnt > 0) {
        KernargParams kq = kernarg_params();
        const size_t ro = (size_t)env * ORLG_RING + ring_pos;
        pf_iat = kq->ring_iat[ro]; pf_ht = kq->ring_ht[ro]; pf_rq = kq->ring_req[ro];
    }
    double current_time = gs->current_time;
    double comp_cur = 1.0;  // _get_network_compactness() of the current occupancy
    int sum_sh = gs->sum_slots_hops;
    const int gs_sum_span = gs->sum_span, gs_sum_gaps = gs->sum_gaps;
    int req_src = gs->req_src, req_dst = gs->req_dst, req_br = gs->req_br, req_sid = gs->req_sid;
    int mt_idx = gs->mt_idx, new_service = gs->new_service;
    // release queue: a time-sorted ring in LDS -- q_n entries from slot q_head on, ascending release time (OrlgParams::qtime)
    int q_head = gs->q_head, q_n = gs->n_running < Q ? gs->n_running : Q;   // (n_running also counts services an overflow lost)
    // release time at the ring's head; an empty ring's slots hold +inf, but a state from elsewhere (load_state) is not trusted on it
    double next_rel = q_n > 0 ? readlane_d(wv.qtime[q_head], 0) : __longlong_as_double((long long)ORLG_INF_BITS);
    int eproc = (int)gs->c[2];  // episode_services_processed, mirrored in a register for `done`
    if (lane < 8) wv.wsc->c[lane] = gs->c[lane];
    if (lane == 0) {
        wv.wsc->sum_bitrate_running = gs->sum_bitrate_running;
        wv.wsc->episodes_done = gs->episodes_done;
        wv.wsc->n_running = gs->n_running;
        wv.wsc->q_overflow = gs->q_overflow;
        wv.wsc->g_thr = gs->g_throughput; wv.wsc->g_comp = gs->g_compactness; wv.wsc->g_lu = gs->g_last_update;
        wv.wsc->req_arrival = gs->req_arrival; wv.wsc->req_holding = gs->req_holding;
    }
    wave_sync();

    // the per-link (span, gaps) cache and its sums travel with the state (they are a function of the occupancy)
    int sum_span = gs_sum_span, sum_gaps = gs_sum_gaps;
    if (NET) comp_cur = network_compactness(sum_span, sum_sh, sum_gaps, E);

    for (int t = 0; t < n_iter; ++t) {
        SEC(2);  // policy
        if (STEPK) {
            // ========================================================== policy: pick (path, slot)
            const int base = tb.pair_base[req_src * N + req_dst];
            // (path, word) lanes: AND over the links of candidate path pp.  With k <= 8 a path takes a group of 8 lanes (two per
            // row of 16, so that a path's words can exchange bits by DPP); otherwise the paths are packed W lanes apart
            const int pp = LS == 8 ? lane >> 3 : lane / W, pw = lane - pp * LS;
            int se_l, hops_l;
            const u64 acc = path_word_rec<W>(wv.occ, tb.recs, base + pp, pw, pp < K && pw < W, se_l, hops_l);
            int my_se = 0;
            if (lane < K) my_se = tb.recs[base + lane].se;
            int my_n = tb.nslots[req_br * ORLG_NSLOT_STRIDE + my_se];  // get_number_slots per candidate

            int a_path = K, a_slot = S;  // rejection (rmsa_env.py:871,913)
            const int policy = p.policy;
            if (FF) {
                const int kmax = policy == ORLG_POLICY_SP ? 1 : K;
                const bool on = pp < kmax && pw < W;
                int n_l = 1;
                if (on) n_l = tb.nslots[req_br * ORLG_NSLOT_STRIDE + se_l];
                u64 r = run_starts<W>(on ? acc : 0ull, n_l, pw);
                const int below = (S - n_l) - 64 * pw;
                r &= below >= 64 ? ~0ull : (below <= 0 ? 0ull : ((1ull << below) - 1ull));
                const int best = wave_min_i32(r ? (pp << 10) | (64 * pw + ctz64(r)) : 0x7fffffff);
                if (best != 0x7fffffff) { a_path = best >> 10; a_slot = best & 1023; }
            } else if (policy == ORLG_POLICY_EXT) {
                const int32_t *acts = kernarg_params()->actions;
                a_path = uni(acts[2 * env]);
                a_slot = uni(acts[2 * env + 1]);
            } else if (policy == ORLG_POLICY_PATH_EXT) {
                // PathOnlyFirstFitAction.action (rmsa_env.py:982-1005): the agent picks the path, first fit picks the slot
                int a = uni(kernarg_params()->actions[env]);
                if (a >= 0 && a < K) {
                    u64 x[W];
#pragma unroll
                    for (int w = 0; w < W; ++w) x[w] = readlane64(acc, a * LS + w);
                    int n = __builtin_amdgcn_readlane(my_n, a);
                    int s0 = first_fit<W>(x, n, S - n, lane);
                    if (s0 >= 0) { a_path = a; a_slot = s0; }
                }
            } else if (policy == ORLG_POLICY_DEEP_EXT) {
                int a = uni(kernarg_params()->actions[env]);
                if (a >= 0 && a < K * p.j) {
                    int route = a / p.j, blk = a - route * p.j;
                    u64 x[W];
#pragma unroll
                    for (int w = 0; w < W; ++w) x[w] = readlane64(acc, route * LS + w);
                    int n = __builtin_amdgcn_readlane(my_n, route), len;
                    int s0 = find_block<W>(x, n, blk, lane, &len);
                    if (s0 >= 0) { a_path = route; a_slot = s0; }
                }
            } else if (LS == 8 && (policy == ORLG_POLICY_SP || policy == ORLG_POLICY_SAP)) {
                // first fit on every candidate path at once: the starts of free runs of >= n slots by shift-and-AND doubling over
                // the path's words (run_starts), then the lowest (path, slot) of the wave
                const int kmax = policy == ORLG_POLICY_SP ? 1 : K;
                const bool on = pp < kmax && pw < W;
                int n_l = 1;
                if (on) n_l = tb.nslots[req_br * ORLG_NSLOT_STRIDE + se_l];
                u64 r = run_starts<W>(on ? acc : 0ull, n_l, pw);
                const int below = (S - n_l) - 64 * pw;  // start slots below S - n (exclusive: rmsa_env.py:860-871)
                r &= below >= 64 ? ~0ull : (below <= 0 ? 0ull : ((1ull << below) - 1ull));
                const int best = wave_min_i32(r ? (pp << 10) | (64 * pw + ctz64(r)) : 0x7fffffff);
                if (best != 0x7fffffff) { a_path = best >> 10; a_slot = best & 1023; }
            } else {
                int max_free = 0;
                const int kmax = (policy == ORLG_POLICY_SP || policy == ORLG_POLICY_DEEP_SP) ? 1 : K;
                for (int idp = 0; idp < kmax; ++idp) {
                    u64 x[W];
#pragma unroll
                    for (int w = 0; w < W; ++w) x[w] = readlane64(acc, idp * LS + w);
                    int n = __builtin_amdgcn_readlane(my_n, idp);
                    if (policy == ORLG_POLICY_DEEP_SP || policy == ORLG_POLICY_DEEP_SAP) {
                        int len;
                        int s0 = find_block<W>(x, n, 0, lane, &len);
                        if (s0 >= 0) { a_path = idp; a_slot = s0; break; }
                    } else {
                        int s0 = first_fit<W>(x, n, S - n, lane);  // NOTE exclusive bound S - n
                        if (s0 >= 0) {
                            if (policy == ORLG_POLICY_LLP) {
                                int fs = 0;
#pragma unroll
                                for (int w = 0; w < W; ++w) fs += popc64(x[w]);
                                if (fs > max_free) { a_path = idp; a_slot = s0; max_free = fs; }
                            } else {
                                a_path = idp; a_slot = s0;
                                break;
                            }
                        }
                    }
                }
            }

            // ========================================================== RMSAEnv.step (rmsa_env.py:222-341)
            SEC(3);  // validate + provision
            const double prev_compact = comp_cur;
            bool accepted = false;
            if (a_path >= 0 && a_path < K && a_slot >= 0 && a_slot < S) {
                const int n = __builtin_amdgcn_readlane(my_n, a_path);
                // the device policies only propose windows they found free; agent actions are checked (is_path_free)
                bool window_ok = true;
                if (!FF && (policy == ORLG_POLICY_EXT || policy == ORLG_POLICY_PATH_EXT || policy == ORLG_POLICY_DEEP_EXT)) {
                    u64 x[W];
#pragma unroll
                    for (int w = 0; w < W; ++w) x[w] = readlane64(acc, a_path * LS + w);
                    window_ok = window_free<W>(x, a_slot, n, S);
                }
                if (window_ok) {
                    // ---- _provision_path (rmsa_env.py:462-513)
                    const int gid = base + a_path;
                    const OrlgPathRec *rec = tb.recs + gid;
                    const int hops = rec->hops;
                    apply_window<W>(wv, rec->link, hops, a_slot, n, false);
                    sum_sh += n * hops;
                    const int br_val = tb.bit_rates[req_br];
                    double cur_thr = 0.0;
                    if (lane == 0) {
                        OrlgWaveScalars *ws = wv.wsc;
                        ws->n_running += 1;
                        ws->sum_bitrate_running += br_val;
                        ws->c[1] += 1; ws->c[3] += 1; ws->c[5] += br_val; ws->c[7] += br_val;
                        wv.hist[NBR + req_br] += 1;
                        wv.hist[3 * NBR + req_br] += 1;
                    }
                    SEC(4);  // statistics at provision
                    if (NET) {
                        wave_sync();
                        cur_thr = (double)wv.wsc->sum_bitrate_running;
                        // per-link stats of the path's links, then _update_network_stats (rmsa_env.py:494-499)
                        link_stats_update<W, FULL, true, DEFER>(wv, tb, S, E, rec->link, hops, current_time, sum_span, sum_gaps,
                                                                comp_cur, sum_sh, cur_thr, DEFER ? kernarg_params()->llog + (size_t)env * E * ORLG_LLOG_CAP : nullptr);
                    }
                    accepted = true;
                    SEC(5);  // queue insert
                    // ---- _add_release (optical_network_env.py:178-189): the entries that are released later move up one slot
                    // (from the top chunk down: a chunk's reads precede its writes), the new one takes the slot that opens
                    const double rel = wv.wsc->req_arrival + wv.wsc->req_holding;
                    if (q_n >= Q) {
                        if (lane == 0) wv.wsc->q_overflow = 1;
                    } else {
                        int r = 0;
                        for (int j0 = (q_n - 1) & ~63; j0 >= 0; j0 -= 64) {
                            const int j = j0 + lane;
                            const bool valid = j < q_n;
                            int pos = q_head + j;
                            pos -= pos >= Q ? Q : 0;
                            double tq = 0.0;
                            uint32_t dq = 0u;
                            if (valid) { tq = wv.qtime[pos]; dq = wv.qdesc[pos]; }
                            const bool later = valid && tq > rel;
                            const int pos1 = pos + 1 == Q ? 0 : pos + 1;
                            if (later) { wv.qtime[pos1] = tq; wv.qdesc[pos1] = dq; }
                            const u64 le = ballot(valid && !later);   // sorted: a prefix of the chunk
                            if (le) { r = j0 + popc64(le); break; }
                        }
                        int pr = q_head + r;
                        pr -= pr >= Q ? Q : 0;
                        if (lane == 0) {
                            wv.qtime[pr] = rel;
                            wv.qdesc[pr] = (uint32_t)gid | ((uint32_t)a_slot << 14) | ((uint32_t)req_br << 24);
                        }
                        q_n += 1;
                        if (r == 0) next_rel = rel;
                    }
                    wave_sync();
                }
            }

            SEC(6);  // outputs
            // per-step outputs (lane 0; consecutive envs are consecutive addresses)
            if (p.out_mask) {
                const size_t o = (size_t)t * p.B + env;
                if (lane == 0) {
                    const int om = p.out_mask;
                    if (om & (1 << ORLG_OUT_PATH)) ORLG_GPTR(int32_t, tb.outs[ORLG_OUT_PATH])[o] = a_path;
                    if (om & (1 << ORLG_OUT_SLOT)) ORLG_GPTR(int32_t, tb.outs[ORLG_OUT_SLOT])[o] = a_slot;
                    if (om & (1 << ORLG_OUT_ACCEPTED)) ORLG_GPTR(uint8_t, tb.outs[ORLG_OUT_ACCEPTED])[o] = accepted ? 1 : 0;
                    if (om & (1 << ORLG_OUT_REWARD))
                        ORLG_GPTR(double, tb.outs[ORLG_OUT_REWARD])[o] =
                            p.reward_mode == 1 ? (accepted ? 1.0 : -1.0) : (accepted ? 1.0 : 0.0);
                    if (om & (1 << ORLG_OUT_REQUEST))
                        ORLG_GPTR(orlg_v4i, tb.outs[ORLG_OUT_REQUEST])[o] =
                            orlg_v4i{req_sid, req_src, req_dst, tb.bit_rates[req_br]};
                    if (om & (1 << ORLG_OUT_ARRIVAL)) ORLG_GPTR(double, tb.outs[ORLG_OUT_ARRIVAL])[o] = wv.wsc->req_arrival;
                    if (om & (1 << ORLG_OUT_HOLDING)) ORLG_GPTR(double, tb.outs[ORLG_OUT_HOLDING])[o] = wv.wsc->req_holding;
                    if (om & (1 << ORLG_OUT_COMPACT)) ORLG_GPTR(double, tb.outs[ORLG_OUT_COMPACT])[o] = comp_cur;
                    if (om & (1 << ORLG_OUT_COMPACT_DIFF))
                        ORLG_GPTR(double, tb.outs[ORLG_OUT_COMPACT_DIFF])[o] = prev_compact - comp_cur;
                    // info["avg_link_compactness" | "avg_link_utilization"] (rmsa_env.py:311-322): np.mean over the
                    // links, taken here -- after the provisioning, before _next_service releases anything
                    if (FULL && (om & (1 << ORLG_OUT_AVG_LINK_COMPACT)))
                        ORLG_GPTR(double, tb.outs[ORLG_OUT_AVG_LINK_COMPACT])[o] = np_mean(wv.lst + 2 * E, E);
                    if (FULL && (om & (1 << ORLG_OUT_AVG_LINK_UTIL)))
                        ORLG_GPTR(double, tb.outs[ORLG_OUT_AVG_LINK_UTIL])[o] = np_mean(wv.lst, E);
                }
            }
            new_service = 0;
        } else if (p.mode == ORLG_MODE_EPISODE_RESET) {
            // reset(only_episode_counters=True) (rmsa_env.py:343-389)
            for (int i = lane; i < NBR; i += 64) { wv.hist[2 * NBR + i] = 0; wv.hist[3 * NBR + i] = 0; }
            wave_sync();
            eproc = 0;
            if (lane == 0) {
                OrlgWaveScalars *ws = wv.wsc;
                ws->c[2] = 0; ws->c[3] = 0; ws->c[6] = 0; ws->c[7] = 0;
                if (new_service) {
                    ws->c[2] = 1;
                    ws->c[6] = tb.bit_rates[req_br];
                    wv.hist[2 * NBR + req_br] += 1;
                }
            }
            if (new_service) eproc = 1;
            wave_sync();
        }

        // ============================================================== _next_service (rmsa_env.py:643-695)
        SEC(7);  // next arrival
        if ((STEPK || p.mode != ORLG_MODE_EPISODE_RESET) && !new_service) {
            if (ring_cnt == 0) {
                SEC(8);  // refill
                if (!mt_loaded) {
                    copy_words(wv.mt, kernarg_params()->mt + (size_t)env * ORLG_MT_N, ORLG_MT_N * 4, lane);
                    mt_loaded = true;
                    wave_sync();
                }
                ring_dirty = true;
                ring_in_lds = true;
                if (p.br_width > 0)   // bit_rate_selection="continuous"
                    ring_cnt = refill_requests_cont_t<true>(wv.mt, wv.ring_iat, wv.ring_ht, wv.ring_req, tb.src_cum, tb.dst_cum, &mt_idx, N,
                                                            p.br_width, p.arrival_lambda, p.holding_lambda);
                else
                    ring_cnt = refill_requests_t<true>(wv.mt, wv.ring_iat, wv.ring_ht, wv.ring_req, tb.src_cum, tb.dst_cum, tb.br_cum,
                                                       &mt_idx, N, NBR, p.arrival_lambda, p.holding_lambda, env);
                ring_pos = 0;
                SEC(7);
            }
            double r_iat, r_ht;
            uint32_t rq;
            if (ring_in_lds) {
                r_iat = wv.ring_iat[ring_pos]; r_ht = wv.ring_ht[ring_pos]; rq = wv.ring_req[ring_pos];
            } else if (t == 0) {
                r_iat = pf_iat; r_ht = pf_ht; rq = pf_rq;
            } else {
                KernargParams kq = kernarg_params();
                const size_t ro = (size_t)env * ORLG_RING + ring_pos;
                r_iat = kq->ring_iat[ro]; r_ht = kq->ring_ht[ro]; rq = kq->ring_req[ro];
            }
            const double at = current_time + r_iat;
            const double ht = r_ht;
            ring_pos += 1; ring_cnt -= 1;
            current_time = at;
            const int src = (int)(rq & 0xffu), dst = (int)((rq >> 8) & 0xffu), bri = (int)(rq >> 16);
            req_sid = eproc;
            req_src = src; req_dst = dst; req_br = bri;
            new_service = 1;
            eproc += 1;
            if (lane == 0) {
                const int br_val = tb.bit_rates[bri];
                OrlgWaveScalars *ws = wv.wsc;
                ws->req_arrival = at; ws->req_holding = ht;
                ws->c[0] += 1; ws->c[2] += 1; ws->c[4] += br_val; ws->c[6] += br_val;
                wv.hist[bri] += 1;
                wv.hist[2 * NBR + bri] += 1;
            }

            // ---- release every service with release time <= now, in time order (rmsa_env.py:689-695): the ring's head
            bool released = false;
            for (;;) {
                SEC(9);  // release scan
                if (!(next_rel <= current_time)) break;   // (a step without a due release touches no queue memory)
                SEC(10);  // release apply
                // ---- _release_path (rmsa_env.py:515-535)
                const uint32_t d = (uint32_t)uni((int)wv.qdesc[q_head]);
                const int gid = (int)(d & 0x3fff), s0 = (int)((d >> 14) & 0x3ff), bri2 = (int)(d >> 24);
                const OrlgPathRec *rec = tb.recs + gid;
                const int hops = rec->hops, se = rec->se;
                const int n = tb.nslots[bri2 * ORLG_NSLOT_STRIDE + se];
                if (lane == 0) {
                    wv.qtime[q_head] = __longlong_as_double((long long)ORLG_INF_BITS);
                    wv.qdesc[q_head] = 0u;
                    wv.wsc->n_running -= 1;
                    wv.wsc->sum_bitrate_running -= tb.bit_rates[bri2];
                }
                q_head = q_head + 1 == Q ? 0 : q_head + 1;
                q_n -= 1;
                apply_window<W>(wv, rec->link, hops, s0, n, true);   // (ends with a wave_sync)
                next_rel = q_n > 0 ? readlane_d(wv.qtime[q_head], 0) : __longlong_as_double((long long)ORLG_INF_BITS);   // the next entry
                sum_sh -= n * hops;
                SEC(11);  // statistics at release
                if (NET)
                    link_stats_update<W, FULL, false, DEFER>(wv, tb, S, E, rec->link, hops, current_time, sum_span, sum_gaps,
                                                             comp_cur, sum_sh, 0.0, DEFER ? kernarg_params()->llog + (size_t)env * E * ORLG_LLOG_CAP : nullptr);
                released = true;
            }
            if (NET && released) comp_cur = network_compactness(sum_span, sum_sh, sum_gaps, E);
        }

        SEC(12);  // done / episode reset
        if (STEPK) {
            const bool done = (eproc == p.episode_length);
            if (lane == 0 && (p.out_mask & (1 << ORLG_OUT_DONE)))
                ORLG_GPTR(uint8_t, tb.outs[ORLG_OUT_DONE])[(size_t)t * p.B + env] = done ? 1 : 0;
            if (done && p.auto_reset) {
                // reset(only_episode_counters=True) with a pending service (rmsa_env.py:343-389)
                for (int i = lane; i < NBR; i += 64) { wv.hist[2 * NBR + i] = 0; wv.hist[3 * NBR + i] = 0; }
                wave_sync();
                eproc = 1;
                if (lane == 0) {
                    OrlgWaveScalars *ws = wv.wsc;
                    ws->episodes_done += 1;
                    ws->c[2] = 1; ws->c[3] = 0; ws->c[6] = tb.bit_rates[req_br]; ws->c[7] = 0;
                    wv.hist[2 * NBR + req_br] = 1;
                }
                wave_sync();
            }
        }
    }

    // ------------------------------------------------------------------ LDS -> HBM (coalesced)
    SEC(13);  // state store
    wave_sync();
    // (the state that leaves carries no pending link updates)
    if (DEFER) link_replay<64>(lane, wv.lst, wv.lint, tb, S, E, kernarg_params()->llog + (size_t)env * E * ORLG_LLOG_CAP);
    {
        KernargParams kp = kernarg_params();
        if ((NW & 1) == 0) {
            copy_words(kp->occ + (size_t)env * NW, wv.occ, NW * 8, lane);
        } else {
            u64 *g = kp->occ + (size_t)env * NW;
            for (int i = lane; i < NW; i += 64) g[i] = wv.occ[i];
        }
        copy_words(kp->qtime + (size_t)env * Q, wv.qtime, Q * 8, lane);
        copy_words(kp->qdesc + (size_t)env * Q, wv.qdesc, Q * 4, lane);
        if (mt_loaded) copy_words(kp->mt + (size_t)env * ORLG_MT_N, wv.mt, ORLG_MT_N * 4, lane);
        if (FULL) copy_words(kp->lstat + (size_t)env * 4 * E, wv.lst, 4 * E * 8, lane);
        copy_words(kp->hist + (size_t)env * 4 * NBR, wv.hist, 4 * NBR * 4, lane);
        if (NET) copy_words(kp->lint + (size_t)env * kp->lint_stride, wv.lint, kp->lint_stride * 4, lane);
        if (ring_dirty) {
            kp->ring_iat[(size_t)env * ORLG_RING + lane] = wv.ring_iat[lane];
            kp->ring_ht[(size_t)env * ORLG_RING + lane] = wv.ring_ht[lane];
            kp->ring_req[(size_t)env * ORLG_RING + lane] = wv.ring_req[lane];
        }
        // the scalar record is assembled in the scratch area and leaves as one 192-byte row
        OrlgEnvScalars *go = reinterpret_cast<OrlgEnvScalars *>(wv.scratch);
        const OrlgWaveScalars *ws = wv.wsc;
        if (lane < 8) go->c[lane] = ws->c[lane];
        if (lane == 0) {
            go->current_time = current_time;
            go->req_arrival = ws->req_arrival; go->req_holding = ws->req_holding;
            go->g_throughput = ws->g_thr; go->g_compactness = ws->g_comp; go->g_last_update = ws->g_lu;
            go->sum_bitrate_running = ws->sum_bitrate_running;
            go->episodes_done = ws->episodes_done;
            go->sum_slots_hops = sum_sh; go->n_running = ws->n_running;
            go->req_src = req_src; go->req_dst = req_dst; go->req_br = req_br; go->req_sid = req_sid;
            go->mt_idx = mt_idx; go->new_service = new_service; go->q_overflow = ws->q_overflow;
            go->ring_pos = ring_pos; go->ring_cnt = ring_cnt;
            go->sum_span = sum_span; go->sum_gaps = sum_gaps; go->q_head = q_head;
            if (ws->q_overflow) *kp->err_flag = 1;   // reported by the next entry point that waits for the stream
        }
        wave_sync();
        if (lane < (int)sizeof(OrlgEnvScalars) / 16)
            reinterpret_cast<uint4 *>(kp->scal + env)[lane] = reinterpret_cast<const uint4 *>(wv.scratch)[lane];
    }
    wave_sync();
    SEC(0);
    }  // work queue
    SEC_FLUSH;
}

template <int W, int STATS, bool DEFER = false>
__global__ __launch_bounds__(ORLG_WAVE *ORLG_MAX_WAVES_PER_BLOCK, 4) void orlg_rmsa_kernel(const OrlgParams p) {
    rmsa_body<W, STATS, true, false, DEFER>(p);
}
template <int W, int STATS, bool DEFER = false>
__global__ __launch_bounds__(ORLG_WAVE *ORLG_MAX_WAVES_PER_BLOCK, 4) void orlg_rmsa_kernel_ff(const OrlgParams p) {
    rmsa_body<W, STATS, true, true, DEFER>(p);
}
template <int W, int STATS>
__global__ __launch_bounds__(ORLG_WAVE *ORLG_MAX_WAVES_PER_BLOCK, 4) void orlg_rmsa_reset_kernel(const OrlgParams p) {
    rmsa_body<W, STATS, false>(p);
}

// ---------------------------------------------------------------------------------------- queries
// For env `env_index`: the k path-wide free bitmaps of its pending request and get_number_slots per path
// (rmsa_env.py:708-719, 745-756).  One wave.
template <int W>
__global__ __launch_bounds__(ORLG_WAVE) void orlg_path_masks_kernel(const OrlgParams p, int env, int gid0, int count,
                                                                    u64 *masks, int32_t *nslots) {
    extern __shared__ __align__(16) unsigned char smem[];
    stage_tables(smem, p);
    const Tab tb = make_tab(smem, p);
    const int lane = threadIdx.x & 63;
    u64 *occ = reinterpret_cast<u64 *>(smem + p.l_shared_bytes);
    const u64 *g = p.occ + (size_t)env * p.NW;
    for (int i = lane; i < p.NW; i += 64) occ[i] = g[i];
    wave_sync();
    const OrlgEnvScalars *sc = p.scal + env;
    // gid0 < 0: the k candidate paths of the pending request; otherwise `count` records starting at gid0
    const int base = gid0 < 0 ? tb.pair_base[sc->req_src * p.N + sc->req_dst] : gid0;
    const int cnt = gid0 < 0 ? p.K : count;
    const int pp = lane / W, pw = lane - pp * W;
    {
        u64 m = path_word<W>(occ, tb.recs, base + pp, pw, pp < cnt);
        if (pp < cnt) masks[pp * W + pw] = m;
    }
    if (lane < cnt) nslots[lane] = tb.nslots[sc->req_br * ORLG_NSLOT_STRIDE + tb.recs[base + lane].se];
}

// DeepRMSAEnv.observation() (deeprmsa_env.py:60-121) for every env.  One wave per env at a time: the grid is sized to the
// device and strides over the environments (the topology tables are staged once per workgroup).  The per-path integers
// (block starts / lengths, slots needed, free slots, free runs) are found with wave-uniform scans and parked in LDS; then
// lane i evaluates element i of the vector -- one fp64 division sequence for all elements -- and the row leaves coalesced.
template <int W>
__global__ __launch_bounds__(ORLG_WAVE *ORLG_MAX_WAVES_PER_BLOCK) void orlg_deeprmsa_obs_kernel(const OrlgParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    stage_tables(smem, p);
    const Tab tb = make_tab(smem, p);
    const int lane = threadIdx.x & 63;
    const int wib = uni((int)(threadIdx.x >> 6));
    const int occ_bytes = (p.NW * 8 + 15) & ~15, obs_bytes = (p.obs_dim * 8 + 15) & ~15;
    unsigned char *wb = smem + p.l_shared_bytes + (size_t)wib * (occ_bytes + obs_bytes);
    u64 *occ = reinterpret_cast<u64 *>(wb);
    int *opa = reinterpret_cast<int *>(wb + occ_bytes);   // [obs_dim] integer operand of element i
    int *opb = opa + p.obs_dim;                            // [obs_dim] second operand (free runs) where needed
    const int N = p.N, K = p.K, S = p.S, J = p.j;
    const int PW = 2 * J + 3, head = 1 + 2 * N;
    const uint32_t pw_inv = (65536u + (uint32_t)PW - 1u) / (uint32_t)PW;
    const int n_waves = (int)(gridDim.x * (blockDim.x >> 6));
    const bool wide = (p.NW & 1) == 0;
    for (int env = blockIdx.x * (int)(blockDim.x >> 6) + wib; env < p.B; env += n_waves) {
        const OrlgEnvScalars *sc = p.scal + env;
        const int src = sc->req_src, dst = sc->req_dst, br = sc->req_br;
        if (wide) copy_words(occ, p.occ + (size_t)env * p.NW, p.NW * 8, lane);
        else {
            const u64 *g = p.occ + (size_t)env * p.NW;
            for (int i = lane; i < p.NW; i += 64) occ[i] = g[i];
        }
        wave_sync();
        const int mn = src < dst ? src : dst, mx = src < dst ? dst : src;
        const int base = tb.pair_base[src * N + dst];
        if (K <= 8) {
            // every candidate path at once: path g on the 8-lane group g (two groups per DPP row), one word per lane
            const int g8 = lane >> 3, w = lane & 7;
            const bool on = g8 < K && w < W;
            int se_l, hops_l;
            const u64 x = path_word_rec<W>(occ, tb.recs, base + g8, w, on, se_l, hops_l);
            int n = 1;
            if (on) n = tb.nslots[br * ORLG_NSLOT_STRIDE + se_l];
            const u64 xprev = lane_prev_u64(x);
            const u64 starts = x & ~((x << 1) | (w > 0 ? xprev >> 63 : 0ull));   // first slots of the free runs
            const u64 bs = starts & run_starts<W>(x, n, w);                        // ... of those with >= n slots: the blocks
            // free slots continuing a run that reaches this word's end (as in link_stats_update)
            const int lead = x == ~0ull ? 64 : ctz64(~x);
            const int nlead_raw = lane_next_i32(lead);
            const int nlead = w < W - 1 ? nlead_raw : 0;
            int e = nlead;
#pragma unroll
            for (int i = 0; i < W - 2; ++i) {
                const int ne_raw = lane_next_i32(e);
                const int ne = w < W - 1 ? ne_raw : 0;
                e = nlead == 64 ? 64 + ne : nlead;
            }
            // blocks in the words before this one (prefix sum over the group's lanes)
            const int cnt = popc64(bs);
            int incl = cnt, o;
            o = lane_back_i32<1>(incl); if (w >= 1) incl += o;
            o = lane_back_i32<2>(incl); if (w >= 2) incl += o;
            o = lane_back_i32<4>(incl); if (w >= 4) incl += o;
            const int n_blocks = group8_add(cnt), total = group8_add(popc64(x)), runs = group8_add(popc64(starts));
            int *row = opa + head + g8 * PW;
            for (int b = 0; b < J; ++b) {
                const int kth = b - (incl - cnt);   // which block of this word
                u64 m = bs;
                for (int q = 0; q < J; ++q)
                    if (q < kth) m &= m - 1;
                if (on && kth >= 0 && kth < cnt) {
                    const int sb = ctz64(m);
                    row[2 * b] = 64 * w + sb;
                    row[2 * b + 1] = free_run_length((~x) >> sb, 64 - sb + e);
                }
                if (on && w == 0 && b >= n_blocks) { row[2 * b] = -1; row[2 * b + 1] = -1; }
            }
            if (on && w == 0) {
                row[2 * J] = n; row[2 * J + 1] = total; row[2 * J + 2] = total;
                opb[head + g8 * PW + 2 * J + 2] = runs;
            }
        } else {
            const int pp = lane / W, pw = lane - pp * W;
            u64 acc = 0ull;
            acc = path_word<W>(occ, tb.recs, base + pp, pw, pp < K);
            int my_se = 0;
            if (lane < K) my_se = tb.recs[base + lane].se;
            int my_n = tb.nslots[br * ORLG_NSLOT_STRIDE + my_se];
            for (int idp = 0; idp < K; ++idp) {
                u64 x[W];
    #pragma unroll
                for (int w = 0; w < W; ++w) x[w] = readlane64(acc, idp * W + w);
                const int n = __builtin_amdgcn_readlane(my_n, idp);
                int *row = opa + head + idp * PW;
                for (int b = 0; b < J; ++b) {
                    int len = 0;
                    int s0 = find_block<W>(x, n, b, lane, &len);
                    if (lane == 0) { row[2 * b] = s0; row[2 * b + 1] = s0 >= 0 ? len : -1; }
                }
                int total = 0, runs = 0;
    #pragma unroll
                for (int w = 0; w < W; ++w) {
                    u64 carry = w > 0 ? (x[w > 0 ? w - 1 : 0] >> 63) : 0ull;
                    total += popc64(x[w]);
                    runs += popc64(x[w] & ~((x[w] << 1) | carry));
                }
                if (lane == 0) {
                    row[2 * J] = n; row[2 * J + 1] = total; row[2 * J + 2] = total;
                    opb[head + idp * PW + 2 * J + 2] = runs;
                }
            }
        }
        wave_sync();
        double *gout = p.o_obs + (size_t)env * p.obs_dim;
        float *gout32 = reinterpret_cast<float *>(p.o_obs) + (size_t)env * p.obs_dim;   // (obs_f32: the same vector rounded once)
        const int br_val = tb.bit_rates[br];
        for (int i = lane; i < p.obs_dim; i += 64) {
            // element i = num / den (one division for every kind of element), optionally followed by (q - 4) / 4
            double num = 0.0, den = 1.0, res;
            bool fixed = false, tail = false;
            double fixed_val = 0.0;
            if (i == 0) {
                num = (double)br_val; den = 100.0;                                   // bit_rate / 100
            } else if (i < head) {
                fixed = true; fixed_val = (i - 1 == mn || i - 1 == N + mx) ? 1.0 : 0.0;   // one-hot endpoints
            } else {
                // (r / PW by multiply-shift: exact for every r < 64 * PW, PW <= 35, checked exhaustively; r < K * PW with K <= 64 -- an integer division by a run-time value is ~30 instructions)
                const int r = i - head, c = r - (int)(((uint32_t)r * pw_inv) >> 16) * PW;
                const int v = opa[i];
                if (c < 2 * J) {
                    if (v < 0) { fixed = true; fixed_val = -1.0; }
                    else if ((c & 1) == 0) { num = 2 * ((double)v - 0.5 * S); den = (double)S; }   // 2 * (start - S/2) / S
                    else { num = (double)v - 8; den = 8.0; }                                        // (length - 8) / 8
                } else if (c == 2 * J) {
                    num = (double)v - 5.5; den = 3.5;                                               // (slots - 5.5) / 3.5
                } else if (c == 2 * J + 1) {
                    num = 2 * ((double)v - 0.5 * S); den = (double)S;                               // 2 * (free - S/2) / S
                } else {
                    const int runs = opb[i];
                    if (runs > 0) { num = (double)v; den = (double)runs; tail = true; }              // (free / runs - 4) / 4
                    else { fixed = true; fixed_val = -1.0; }
                }
            }
            res = num / den;
            if (tail) res = (res - 4) * 0.25;   // (x / 4 is x * 0.25 exactly)
            const double val = fixed ? fixed_val : res;
            if (p.obs_f32) gout32[i] = (float)val; else gout[i] = val;
        }
        wave_sync();
    }
}
