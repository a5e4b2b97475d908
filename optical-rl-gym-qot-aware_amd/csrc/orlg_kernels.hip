// orlg_kernels.hip -- gfx950 (CDNA4, wave64) kernels of the batched RMSA / DeepRMSA step() path.
//
// Execution model: ONE WAVEFRONT PER ENVIRONMENT.  A 256-thread workgroup carries four environments;
// nothing is shared between the waves of a workgroup, so there is no __syncthreads() anywhere and a
// wave whose environment index is out of range simply exits.  For the duration of a launch (n_steps
// steps) the wave keeps its environment on chip:
//     LDS (per wave)   link x slot free bitmap  E*W uint64        (occ)
//                      release queue            Q x (f64 time, u32 descriptor)
//                      MT19937 state            624 x u32
//                      per-link statistics      4 x E f64, per-link (span, gaps) cache, histograms
//     SGPR/VGPR        wave-uniform scalars: clock, pending request, counters, running sums
// and reads / writes the HBM copy exactly once, with lane-contiguous (coalesced) accesses.
// Within a step the 64 lanes are used as
//     (path, word) lanes  to AND the link bitmaps of the k candidate paths   (get_available_slots)
//     slot lanes          for the first-fit scan: lane l owns slot 64w+l, a __ballot gives the
//                         lowest fitting start                              (is_path_free loops)
//     (hop, word) lanes   to provision / release a slot window and to rebuild the link statistics
//     queue lanes         to find expired services with one ballot per 64 queue slots
//     node / rate lanes   for CPython's bisect in random.choices
// All fp64 arithmetic repeats the reference's operations one for one (compile with -ffp-contract=off).
//
// Reference: optical_rl_gym/envs/rmsa_env.py (step :222-341, _provision_path :462-513, _release_path
// :515-535, _update_network_stats :537-560, _update_link_stats :562-641, _next_service :643-695,
// get_number_slots :708-719, is_path_free :721-734, get_available_slots :745-756, get_available_blocks
// :774-804, _get_network_compactness :806-851, heuristics :854-937), optical_network_env.py
// (_add_release :178-189, _get_node_pair :191-208), deeprmsa_env.py (step :48-58, observation :60-121).
#include <hip/hip_runtime.h>

#include "orlg_device.h"
#include "orlg_math.h"

typedef uint64_t u64;

#define DEV __device__ __forceinline__
#define ORLG_INF_BITS 0x7ff0000000000000ull

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

// ---------------------------------------------------------------------------------------- wave context
struct Wave {
    int lane;
    u64 *occ;
    double *qtime;
    uint32_t *qdesc;
    uint32_t *mt;
    double *lst;   // [4][E]
    int32_t *hist; // [4][NBR]
    int32_t *lint; // [E] span | gaps << 16
    uint32_t *scratch;
};

// ---------------------------------------------------------------------------------------- MT19937
// Regenerate all 624 words in place (CPython _randommodule.c genrand_uint32).  Sub-round r handles
// kk = 64r + lane; mt[kk+1] is still old (same or later sub-round), mt[kk+397] is old for kk < 227 and
// mt[kk-227] is already new for kk >= 227, exactly as in the sequential loop.
DEV void mt_regenerate(uint32_t *mt, int lane) {
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

// ---------------------------------------------------------------------------------------- first fit
// x[w]: wave-uniform free bitmap of one path (AND over its links).  Lane l of word w owns slot 64w+l
// and computes the length of the free run starting there.
template <int W>
DEV void ext_chain(const u64 (&x)[W], int (&ext)[W]) {
    ext[W - 1] = 0;
#pragma unroll
    for (int w = W - 2; w >= 0; --w) ext[w] = (x[w + 1] == ~0ull) ? 64 + ext[w + 1] : ctz64(~x[w + 1]);
}

// smallest s in [0, limit) with slots [s, s+n) free, or -1 (rmsa_env.py:860-871, 908-913)
template <int W>
DEV int first_fit(const u64 (&x)[W], int n, int limit, int lane) {
    if (limit <= 0) return -1;
    int ext[W];
    ext_chain<W>(x, ext);
#pragma unroll
    for (int w = 0; w < W; ++w) {
        if (x[w] != 0ull && 64 * w < limit) {
            u64 t = (~x[w]) >> lane;
            int len = t ? ctz64(t) : (64 - lane) + ext[w];
            u64 m = ballot(len >= n && (64 * w + lane) < limit);
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
            u64 t = (~x[w]) >> lane;
            int len = t ? ctz64(t) : (64 - lane) + ext[w];
            u64 m = ballot(((starts >> lane) & 1ull) && len >= n);
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

DEV int rec_byte(u64 lo, u64 hi, int i) { return (int)(((i < 8 ? lo >> (8 * i) : hi >> (8 * (i - 8)))) & 0xffull); }

// AND of the link bitmaps of path record `gid` for word w (get_available_slots, rmsa_env.py:745-756)
template <int W>
DEV u64 path_word(const Wave &wv, const OrlgPathRec *recs, int gid, int w) {
    const u64 *rp = reinterpret_cast<const u64 *>(recs + gid);
    u64 lo = rp[0], hi = rp[1];
    int hops = (int)(lo & 0xff);
    u64 acc = ~0ull;
    for (int h = 0; h < hops; ++h) acc &= wv.occ[rec_byte(lo, hi, 2 + h) * W + w];
    return acc;
}

// ---------------------------------------------------------------------------------------- link statistics
// Rebuild, for a list of links, the integer run statistics of the link's free bitmap and (FULL) the
// time-weighted floats of _update_link_stats (rmsa_env.py:562-641).  Also maintains the per-link
// (span, gaps) cache whose sums give _get_network_compactness (rmsa_env.py:806-851):
//     span = lambda_max - lambda_min, gaps = free runs inside the used span = used runs - 1
// for links with more than one used run, 0 otherwise.  links == nullptr means links h0..h0+n-1.
template <int W, bool FLOATS>
DEV void link_stats_update(Wave &wv, const OrlgParams &p, const uint8_t *links, int first, int nlinks, double now,
                           int &sum_span, int &sum_gaps) {
    constexpr int HPC = 64 / W;  // links per chunk
    const int lane = wv.lane;
    const int S = p.S, E = p.E;
    const int hl = lane / W, w = lane - hl * W;
    for (int h0 = 0; h0 < nlinks; h0 += HPC) {
        int nl = nlinks - h0 < HPC ? nlinks - h0 : HPC;
        // ---- phase A: (link, word) lanes
        if (hl < nl) {
            int link = links ? (int)links[first + h0 + hl] : first + h0 + hl;
            const u64 *row = wv.occ + link * W;
            u64 x = row[w];
            u64 prev = w > 0 ? row[w - 1] : 0ull;
            u64 u = ~x & valid_mask(S, w);
            u64 carry_f = w > 0 ? (prev >> 63) : 0ull;
            u64 carry_u = w > 0 ? ((~prev) >> 63) : 0ull;
            u64 fstarts = x & ~((x << 1) | carry_f);
            u64 ustarts = u & ~((u << 1) | carry_u);
            int pf = popc64(x), nfs = popc64(fstarts), nus = popc64(ustarts);
            int lo = u ? 64 * w + ctz64(u) : 0x7fff;
            int hi = u ? 64 * w + 64 - clz64(u) : 0;
            int ml = 0;
            if (FLOATS) {
                int e = 0;
                if ((x >> 63) && w < W - 1) {
                    for (int w2 = w + 1; w2 < W; ++w2) {
                        u64 y = row[w2];
                        if (y == ~0ull) { e += 64; } else { e += ctz64(~y); break; }
                    }
                }
                u64 st = fstarts;
                while (st) {
                    int b = ctz64(st);
                    st &= st - 1;
                    u64 t = (~x) >> b;
                    int len = t ? ctz64(t) : 64 - b + e;
                    ml = len > ml ? len : ml;
                }
            }
            uint32_t *sc = wv.scratch + lane * 4;
            sc[0] = (uint32_t)pf | ((uint32_t)nfs << 16);
            sc[1] = (uint32_t)nus | ((uint32_t)ml << 16);
            sc[2] = (uint32_t)lo | ((uint32_t)hi << 16);
        }
        wave_sync();
        // ---- phase B: one lane per link
        int dspan = 0, dgaps = 0;
        if (lane < nl) {
            int link = links ? (int)links[first + h0 + lane] : first + h0 + lane;
            int freec = 0, F = 0, U = 0, ml = 0, lmin = 0x7fff, lmax = 0;
#pragma unroll
            for (int q = 0; q < W; ++q) {
                const uint32_t *sc = wv.scratch + (lane * W + q) * 4;
                uint32_t a = sc[0], b = sc[1], c = sc[2];
                freec += (int)(a & 0xffff);
                F += (int)(a >> 16);
                U += (int)(b & 0xffff);
                int m = (int)(b >> 16);
                ml = m > ml ? m : ml;
                int lo = (int)(c & 0xffff), hi = (int)(c >> 16);
                lmin = lo < lmin ? lo : lmin;
                lmax = hi > lmax ? hi : lmax;
            }
            int nspan = U > 1 ? lmax - lmin : 0, ngaps = U > 1 ? U - 1 : 0;
            int old = wv.lint[link];
            wv.lint[link] = nspan | (ngaps << 16);
            dspan = nspan - (old & 0xffff);
            dgaps = ngaps - (old >> 16);
            if (FLOATS) {
                double *l_util = wv.lst, *l_ef = wv.lst + E, *l_c = wv.lst + 2 * E, *l_lu = wv.lst + 3 * E;
                double last_update = l_lu[link];
                double time_diff = now - last_update;
                if (now > 0) {
                    const u64 *row = wv.occ + link * W;
                    bool first_free = row[0] & 1ull;
                    bool last_free = (row[(S - 1) >> 6] >> ((S - 1) & 63)) & 1ull;
                    double cur_util = (double)(S - freec) / (double)S;
                    l_util[link] = ((l_util[link] * last_update) + (cur_util * time_diff)) / now;
                    double cur_ef = 0.0, cur_c = 0.0;
                    if (freec > 0) {
                        int max_empty = (F > 1 && !(F == 2 && first_free && last_free)) ? ml : 0;
                        cur_ef = 1.0 - ((double)max_empty / (double)freec);
                        cur_c = U > 1 ? ((double)(lmax - lmin) / (double)(S - freec)) * (1.0 / (double)U) : 1.0;
                    }
                    l_ef[link] = ((l_ef[link] * last_update) + (cur_ef * time_diff)) / now;
                    l_c[link] = ((l_c[link] * last_update) + (cur_c * time_diff)) / now;
                }
                l_lu[link] = now;
            }
        }
        for (int q = 0; q < nl; ++q) {
            sum_span += __builtin_amdgcn_readlane(dspan, q);
            sum_gaps += __builtin_amdgcn_readlane(dgaps, q);
        }
        wave_sync();
    }
}

// _get_network_compactness (rmsa_env.py:844-851) from the maintained integer sums
DEV double network_compactness(int sum_span, int sum_slots_hops, int sum_gaps, int E) {
    if (sum_gaps > 0) return ((double)sum_span / (double)sum_slots_hops) * ((double)E / (double)sum_gaps);
    return 1.0;
}

// set (release) or clear (provision) the window [s, s+n) on every link of path record gid
template <int W>
DEV void apply_window(Wave &wv, const OrlgPathRec *recs, int gid, int hops, int s, int n, bool set_free) {
    constexpr int HPC = 64 / W;
    const uint8_t *rb = reinterpret_cast<const uint8_t *>(recs + gid);
    const int hl = wv.lane / W, w = wv.lane - hl * W;
    u64 m = window_mask(s, n, w);
    for (int h0 = 0; h0 < hops; h0 += HPC) {
        int h = h0 + hl;
        if (hl < HPC && h < hops && m) {
            u64 *word = wv.occ + (int)rb[2 + h] * W + w;
            *word = set_free ? (*word | m) : (*word & ~m);
        }
    }
    wave_sync();
}

// ---------------------------------------------------------------------------------------- the step kernel
template <int W, int STATS>
__global__ __launch_bounds__(ORLG_WAVE *ORLG_WAVES_PER_BLOCK) void orlg_rmsa_kernel(const OrlgParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wib = uni((int)(threadIdx.x >> 6));
    const int env = blockIdx.x * ORLG_WAVES_PER_BLOCK + wib;
    if (env >= p.B) return;
    unsigned char *wb = smem + (size_t)wib * p.l_wave_bytes;
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

    const int E = p.E, S = p.S, K = p.K, N = p.N, NBR = p.NBR, Q = p.Q, NW = p.NW;
    constexpr bool NET = STATS >= 1;
    constexpr bool FULL = STATS >= 2;

    // ------------------------------------------------------------------ HBM -> LDS (coalesced)
    {
        const u64 *g = p.occ + (size_t)env * NW;
        for (int i = lane; i < NW; i += 64) wv.occ[i] = g[i];
        const double *gq = p.qtime + (size_t)env * Q;
        const uint32_t *gd = p.qdesc + (size_t)env * Q;
        for (int i = lane; i < Q; i += 64) { wv.qtime[i] = gq[i]; wv.qdesc[i] = gd[i]; }
        const uint32_t *gm = p.mt + (size_t)env * ORLG_MT_N;
        for (int i = lane; i < ORLG_MT_N; i += 64) wv.mt[i] = gm[i];
        if (FULL) {
            const double *gl = p.lstat + (size_t)env * 4 * E;
            for (int i = lane; i < 4 * E; i += 64) wv.lst[i] = gl[i];
        }
        const int32_t *gh = p.hist + (size_t)env * 4 * NBR;
        for (int i = lane; i < 4 * NBR; i += 64) wv.hist[i] = gh[i];
        for (int i = lane; i < E; i += 64) wv.lint[i] = 0;
    }
    OrlgEnvScalars sc = p.scal[env];
    wave_sync();

    int sum_span = 0, sum_gaps = 0;
    if (NET) link_stats_update<W, false>(wv, p, nullptr, 0, E, 0.0, sum_span, sum_gaps);

    // wave-uniform working copies
    double current_time = sc.current_time;
    double req_arrival = sc.req_arrival, req_holding = sc.req_holding;
    double g_thr = sc.g_throughput, g_comp = sc.g_compactness, g_lu = sc.g_last_update;
    long long c_proc = sc.c[0], c_acc = sc.c[1], c_eproc = sc.c[2], c_eacc = sc.c[3];
    long long c_req = sc.c[4], c_prov = sc.c[5], c_ereq = sc.c[6], c_eprov = sc.c[7];
    long long sum_br = sc.sum_bitrate_running, episodes_done = sc.episodes_done;
    int sum_sh = sc.sum_slots_hops, n_running = sc.n_running;
    int req_src = sc.req_src, req_dst = sc.req_dst, req_br = sc.req_br, req_sid = sc.req_sid;
    int mt_idx = sc.mt_idx, new_service = sc.new_service, q_overflow = sc.q_overflow;

    const int n_iter = p.mode == ORLG_MODE_STEP ? p.n_steps : 1;
    for (int t = 0; t < n_iter; ++t) {
        bool done = false;
        if (p.mode == ORLG_MODE_STEP) {
            // ========================================================== policy: pick (path, slot)
            const int base = p.pair_base[req_src * N + req_dst];
            // (path, word) lanes: AND over the links of candidate path pp
            const int pp = lane / W, pw = lane - pp * W;
            u64 acc = 0ull;
            if (pp < K) acc = path_word<W>(wv, p.recs, base + pp, pw);
            int my_se = 0;
            if (lane < K) my_se = reinterpret_cast<const uint8_t *>(p.recs + base + lane)[1];
            int my_n = p.nslots_tab[req_br * ORLG_NSLOT_STRIDE + my_se];  // get_number_slots per candidate

            int a_path = K, a_slot = S;  // rejection (rmsa_env.py:871,913)
            const int policy = p.policy;
            if (policy == ORLG_POLICY_EXT) {
                a_path = uni(p.actions[2 * env]);
                a_slot = uni(p.actions[2 * env + 1]);
            } else if (policy == ORLG_POLICY_DEEP_EXT) {
                int a = uni(p.actions[env]);
                if (a >= 0 && a < K * p.j) {
                    int route = a / p.j, blk = a - route * p.j;
                    u64 x[W];
#pragma unroll
                    for (int w = 0; w < W; ++w) x[w] = readlane64(acc, route * W + w);
                    int n = __builtin_amdgcn_readlane(my_n, route), len;
                    int s0 = find_block<W>(x, n, blk, lane, &len);
                    if (s0 >= 0) { a_path = route; a_slot = s0; }
                }
            } else {
                long long max_free = 0;
                const int kmax = (policy == ORLG_POLICY_SP || policy == ORLG_POLICY_DEEP_SP) ? 1 : K;
                for (int idp = 0; idp < kmax; ++idp) {
                    u64 x[W];
#pragma unroll
                    for (int w = 0; w < W; ++w) x[w] = readlane64(acc, idp * W + w);
                    int n = __builtin_amdgcn_readlane(my_n, idp);
                    if (policy == ORLG_POLICY_DEEP_SP || policy == ORLG_POLICY_DEEP_SAP) {
                        int len;
                        int s0 = find_block<W>(x, n, 0, lane, &len);
                        if (s0 >= 0) { a_path = idp; a_slot = s0; break; }
                    } else {
                        int s0 = first_fit<W>(x, n, S - n, lane);  // NOTE exclusive bound S - n
                        if (s0 >= 0) {
                            if (policy == ORLG_POLICY_LLP) {
                                long long fs = 0;
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
            double prev_compact = 1.0, cur_compact = 1.0;
            if (NET) prev_compact = network_compactness(sum_span, sum_sh, sum_gaps, E);
            bool accepted = false;
            if (a_path >= 0 && a_path < K && a_slot >= 0 && a_slot < S) {
                u64 x[W];
#pragma unroll
                for (int w = 0; w < W; ++w) x[w] = readlane64(acc, a_path * W + w);
                const int n = __builtin_amdgcn_readlane(my_n, a_path);
                if (window_free<W>(x, a_slot, n, S)) {
                    // ---- _provision_path (rmsa_env.py:462-513)
                    const int gid = base + a_path;
                    const uint8_t *rb = reinterpret_cast<const uint8_t *>(p.recs + gid);
                    const int hops = rb[0];
                    apply_window<W>(wv, p.recs, gid, hops, a_slot, n, false);
                    n_running += 1;
                    sum_sh += n * hops;
                    const int br_val = p.bit_rates[req_br];
                    sum_br += br_val;
                    if (NET) {
                        if (FULL) link_stats_update<W, true>(wv, p, rb + 2, 0, hops, current_time, sum_span, sum_gaps);
                        else link_stats_update<W, false>(wv, p, rb + 2, 0, hops, current_time, sum_span, sum_gaps);
                        // _update_network_stats (rmsa_env.py:537-560)
                        double time_diff = current_time - g_lu;
                        if (current_time > 0) {
                            double cur_thr = (double)sum_br;
                            g_thr = ((g_thr * g_lu) + (cur_thr * time_diff)) / current_time;
                            double cc = network_compactness(sum_span, sum_sh, sum_gaps, E);
                            g_comp = ((g_comp * g_lu) + (cc * time_diff)) / current_time;
                        }
                        g_lu = current_time;
                    }
                    c_acc += 1; c_eacc += 1; c_prov += br_val; c_eprov += br_val;
                    if (lane == 0) { wv.hist[NBR + req_br] += 1; wv.hist[3 * NBR + req_br] += 1; }
                    accepted = true;
                    // ---- _add_release (optical_network_env.py:178-189): first empty queue slot
                    double rel = req_arrival + req_holding;
                    bool placed = false;
                    for (int q0 = 0; q0 < Q && !placed; q0 += 64) {
                        u64 m = ballot(__double_as_longlong(wv.qtime[q0 + lane]) == (long long)ORLG_INF_BITS);
                        if (m) {
                            int l = ctz64(m);
                            if (lane == l) {
                                wv.qtime[q0 + l] = rel;
                                wv.qdesc[q0 + l] = (uint32_t)gid | ((uint32_t)a_slot << 14) | ((uint32_t)req_br << 24);
                            }
                            placed = true;
                        }
                    }
                    if (!placed) q_overflow = 1;
                    wave_sync();
                }
            }
            if (NET) cur_compact = network_compactness(sum_span, sum_sh, sum_gaps, E);

            // per-step outputs (lane 0; consecutive envs are consecutive addresses)
            const size_t o = (size_t)t * p.B + env;
            if (lane == 0) {
                if (p.o_path) p.o_path[o] = a_path;
                if (p.o_slot) p.o_slot[o] = a_slot;
                if (p.o_accepted) p.o_accepted[o] = accepted ? 1 : 0;
                if (p.o_reward) p.o_reward[o] = p.reward_mode == 1 ? (accepted ? 1.0 : -1.0) : (accepted ? 1.0 : 0.0);
                if (p.o_request) {
                    int4 r = make_int4(req_sid, req_src, req_dst, p.bit_rates[req_br]);
                    reinterpret_cast<int4 *>(p.o_request)[o] = r;
                }
                if (p.o_arrival) p.o_arrival[o] = req_arrival;
                if (p.o_holding) p.o_holding[o] = req_holding;
                if (p.o_compact) p.o_compact[o] = cur_compact;
                if (p.o_compact_diff) p.o_compact_diff[o] = prev_compact - cur_compact;
            }
            new_service = 0;
        } else if (p.mode == ORLG_MODE_EPISODE_RESET) {
            // reset(only_episode_counters=True) (rmsa_env.py:343-389)
            c_eproc = 0; c_eacc = 0; c_ereq = 0; c_eprov = 0;
            for (int i = lane; i < NBR; i += 64) { wv.hist[2 * NBR + i] = 0; wv.hist[3 * NBR + i] = 0; }
            wave_sync();
            if (new_service) {
                c_eproc += 1;
                c_ereq += p.bit_rates[req_br];
                if (lane == 0) wv.hist[2 * NBR + req_br] += 1;
            }
        }

        // ============================================================== _next_service (rmsa_env.py:643-695)
        if (p.mode != ORLG_MODE_EPISODE_RESET && !new_service) {
            double u[5];
            draw5(wv, mt_idx, u);
            // expovariate: -log(1 - u) / lambd ; lane 0 does the inter-arrival, lane 2 the holding time
            double uu = lane == 2 ? u[1] : u[0];
            double lam = lane == 2 ? p.holding_lambda : p.arrival_lambda;
            double ex = -orlg_log(1.0 - uu) / lam;
            double at = current_time + readlane_d(ex, 0);
            double ht = readlane_d(ex, 2);
            current_time = at;
            int src = choice_cum(p.src_cum, N, u[2], lane);
            int dst = choice_cum(p.dst_cum + (size_t)src * N, N, u[3], lane);
            int bri = choice_cum(p.br_cum, NBR, u[4], lane);
            req_sid = (int)c_eproc;
            req_src = src; req_dst = dst; req_br = bri; req_arrival = at; req_holding = ht;
            new_service = 1;
            const int br_val = p.bit_rates[bri];
            c_proc += 1; c_eproc += 1; c_req += br_val; c_ereq += br_val;
            if (lane == 0) { wv.hist[bri] += 1; wv.hist[2 * NBR + bri] += 1; }

            // ---- release every service with release time <= now, in time order (rmsa_env.py:689-695)
            for (;;) {
                double best_t = 0.0;
                int best_q = -1;
                for (int q0 = 0; q0 < Q; q0 += 64) {
                    double tq = wv.qtime[q0 + lane];
                    u64 m = ballot(tq <= current_time);
                    while (m) {
                        int l = ctz64(m);
                        m &= m - 1;
                        double tt = readlane_d(tq, l);
                        if (best_q < 0 || tt < best_t) { best_t = tt; best_q = q0 + l; }
                    }
                }
                if (best_q < 0) break;
                // ---- _release_path (rmsa_env.py:515-535)
                const uint32_t d = wv.qdesc[best_q];
                const int gid = (int)(d & 0x3fff), s0 = (int)((d >> 14) & 0x3ff), bri2 = (int)(d >> 24);
                const uint8_t *rb = reinterpret_cast<const uint8_t *>(p.recs + gid);
                const int hops = rb[0], se = rb[1];
                const int n = p.nslots_tab[bri2 * ORLG_NSLOT_STRIDE + se];
                if (lane == 0) wv.qtime[best_q] = __longlong_as_double((long long)ORLG_INF_BITS);
                apply_window<W>(wv, p.recs, gid, hops, s0, n, true);
                n_running -= 1;
                sum_sh -= n * hops;
                sum_br -= p.bit_rates[bri2];
                if (NET) {
                    if (FULL) link_stats_update<W, true>(wv, p, rb + 2, 0, hops, current_time, sum_span, sum_gaps);
                    else link_stats_update<W, false>(wv, p, rb + 2, 0, hops, current_time, sum_span, sum_gaps);
                }
            }
        }

        if (p.mode == ORLG_MODE_STEP) {
            done = (c_eproc == (long long)p.episode_length);
            if (lane == 0 && p.o_done) p.o_done[(size_t)t * p.B + env] = done ? 1 : 0;
            if (done && p.auto_reset) {
                // reset(only_episode_counters=True) with a pending service (rmsa_env.py:343-389)
                episodes_done += 1;
                c_eacc = 0; c_eprov = 0;
                c_eproc = 1;
                c_ereq = p.bit_rates[req_br];
                for (int i = lane; i < NBR; i += 64) { wv.hist[2 * NBR + i] = 0; wv.hist[3 * NBR + i] = 0; }
                wave_sync();
                if (lane == 0) wv.hist[2 * NBR + req_br] = 1;
                wave_sync();
            }
        }
    }

    // ------------------------------------------------------------------ LDS -> HBM (coalesced)
    wave_sync();
    {
        u64 *g = p.occ + (size_t)env * NW;
        for (int i = lane; i < NW; i += 64) g[i] = wv.occ[i];
        double *gq = p.qtime + (size_t)env * Q;
        uint32_t *gd = p.qdesc + (size_t)env * Q;
        for (int i = lane; i < Q; i += 64) { gq[i] = wv.qtime[i]; gd[i] = wv.qdesc[i]; }
        uint32_t *gm = p.mt + (size_t)env * ORLG_MT_N;
        for (int i = lane; i < ORLG_MT_N; i += 64) gm[i] = wv.mt[i];
        if (FULL) {
            double *gl = p.lstat + (size_t)env * 4 * E;
            for (int i = lane; i < 4 * E; i += 64) gl[i] = wv.lst[i];
        }
        int32_t *gh = p.hist + (size_t)env * 4 * NBR;
        for (int i = lane; i < 4 * NBR; i += 64) gh[i] = wv.hist[i];
    }
    if (lane == 0) {
        sc.current_time = current_time;
        sc.req_arrival = req_arrival; sc.req_holding = req_holding;
        sc.g_throughput = g_thr; sc.g_compactness = g_comp; sc.g_last_update = g_lu;
        sc.c[0] = c_proc; sc.c[1] = c_acc; sc.c[2] = c_eproc; sc.c[3] = c_eacc;
        sc.c[4] = c_req; sc.c[5] = c_prov; sc.c[6] = c_ereq; sc.c[7] = c_eprov;
        sc.sum_bitrate_running = sum_br; sc.episodes_done = episodes_done;
        sc.sum_slots_hops = sum_sh; sc.n_running = n_running;
        sc.req_src = req_src; sc.req_dst = req_dst; sc.req_br = req_br; sc.req_sid = req_sid;
        sc.mt_idx = mt_idx; sc.new_service = new_service; sc.q_overflow = q_overflow;
        p.scal[env] = sc;
    }
}

// ---------------------------------------------------------------------------------------- queries
// For env `env_index`: the k path-wide free bitmaps of its pending request and get_number_slots per path
// (rmsa_env.py:708-719, 745-756).  One wave.
template <int W>
__global__ __launch_bounds__(ORLG_WAVE) void orlg_path_masks_kernel(const OrlgParams p, int env, u64 *masks,
                                                                    int32_t *nslots) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    Wave wv;
    wv.lane = lane;
    wv.occ = reinterpret_cast<u64 *>(smem);
    const u64 *g = p.occ + (size_t)env * p.NW;
    for (int i = lane; i < p.NW; i += 64) wv.occ[i] = g[i];
    wave_sync();
    const OrlgEnvScalars *sc = p.scal + env;
    const int base = p.pair_base[sc->req_src * p.N + sc->req_dst];
    const int pp = lane / W, pw = lane - pp * W;
    if (pp < p.K) masks[pp * W + pw] = path_word<W>(wv, p.recs, base + pp, pw);
    if (lane < p.K) {
        int se = reinterpret_cast<const uint8_t *>(p.recs + base + lane)[1];
        nslots[lane] = p.nslots_tab[sc->req_br * ORLG_NSLOT_STRIDE + se];
    }
}

// DeepRMSAEnv.observation() (deeprmsa_env.py:60-121) for every env; one wave per env.
template <int W>
__global__ __launch_bounds__(ORLG_WAVE *ORLG_WAVES_PER_BLOCK) void orlg_deeprmsa_obs_kernel(const OrlgParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wib = uni((int)(threadIdx.x >> 6));
    const int env = blockIdx.x * ORLG_WAVES_PER_BLOCK + wib;
    if (env >= p.B) return;
    Wave wv;
    wv.lane = lane;
    wv.occ = reinterpret_cast<u64 *>(smem + (size_t)wib * ((p.NW * 8 + 15) & ~15));
    const u64 *g = p.occ + (size_t)env * p.NW;
    for (int i = lane; i < p.NW; i += 64) wv.occ[i] = g[i];
    wave_sync();
    const OrlgEnvScalars *sc = p.scal + env;
    const int N = p.N, K = p.K, S = p.S, J = p.j;
    const int src = sc->req_src, dst = sc->req_dst, br = sc->req_br;
    double *out = p.o_obs + (size_t)env * p.obs_dim;
    const int mn = src < dst ? src : dst, mx = src < dst ? dst : src;
    // bit rate + one-hot endpoints
    if (lane == 0) out[0] = (double)p.bit_rates[br] / 100;
    for (int i = lane; i < 2 * N; i += 64) out[1 + i] = (i == mn || i == N + mx) ? 1.0 : 0.0;
    const int base = p.pair_base[src * N + dst];
    const int pp = lane / W, pw = lane - pp * W;
    u64 acc = 0ull;
    if (pp < K) acc = path_word<W>(wv, p.recs, base + pp, pw);
    int my_se = 0;
    if (lane < K) my_se = reinterpret_cast<const uint8_t *>(p.recs + base + lane)[1];
    int my_n = p.nslots_tab[br * ORLG_NSLOT_STRIDE + my_se];
    const int PW = 2 * J + 3;
    double *sp = out + 1 + 2 * N;
    for (int idp = 0; idp < K; ++idp) {
        u64 x[W];
#pragma unroll
        for (int w = 0; w < W; ++w) x[w] = readlane64(acc, idp * W + w);
        const int n = __builtin_amdgcn_readlane(my_n, idp);
        double *row = sp + idp * PW;
        for (int b = 0; b < J; ++b) {
            int len = 0;
            int s0 = find_block<W>(x, n, b, lane, &len);
            if (lane == 0) {
                row[2 * b] = s0 >= 0 ? 2 * ((double)s0 - 0.5 * S) / S : -1.0;
                row[2 * b + 1] = s0 >= 0 ? ((double)len - 8) / 8 : -1.0;
            }
        }
        int total = 0, runs = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            u64 carry = w > 0 ? (x[w > 0 ? w - 1 : 0] >> 63) : 0ull;
            total += popc64(x[w]);
            runs += popc64(x[w] & ~((x[w] << 1) | carry));
        }
        if (lane == 0) {
            row[2 * J] = ((double)n - 5.5) / 3.5;
            row[2 * J + 1] = 2 * ((double)total - 0.5 * S) / S;
            row[2 * J + 2] = runs > 0 ? ((double)total / (double)runs - 4) / 4 : -1.0;
        }
    }
}

// Sum of the counters of all envs (one workgroup; 64-bit integer adds, deterministic order per lane
// then a fixed tree): the vector the multi-GPU layer all-reduces.
__global__ __launch_bounds__(256) void orlg_reduce_counters_kernel(const OrlgEnvScalars *scal, int B, long long *out) {
    __shared__ long long part[256][10];
    long long acc[10];
    for (int q = 0; q < 10; ++q) acc[q] = 0;
    for (int i = threadIdx.x; i < B; i += 256) {
        for (int q = 0; q < 8; ++q) acc[q] += scal[i].c[q];
        acc[8] += scal[i].episodes_done;
        acc[9] += 1;
    }
    for (int q = 0; q < 10; ++q) part[threadIdx.x][q] = acc[q];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
            for (int q = 0; q < 10; ++q) part[threadIdx.x][q] += part[threadIdx.x + s][q];
        __syncthreads();
    }
    if (threadIdx.x < 16) out[threadIdx.x] = threadIdx.x < 10 ? part[0][threadIdx.x] : 0;
}
