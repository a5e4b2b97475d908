// orlg_group_kernels.hip -- RMSA step kernel, FOUR ENVIRONMENTS PER WAVEFRONT (one 16-lane DPP row each).
//
// The wave-per-environment kernel (orlg_kernels.hip) is bound by instruction issue, and most of its instructions are
// wave-uniform control work done for one environment.  Here a wave steps four environments in lockstep: every scalar of an
// environment (clock, pending request, counters, running sums) lives in VGPRs replicated over its row, so one instruction does
// that piece of bookkeeping for four environments, and the bitmap work maps onto the 16 lanes of a row:
//     (path, word) lanes   16 / W candidate paths per pass: AND of the link bitmaps; the first fit is found with a shift-and-AND
//                          doubling over the path's W words (the next word arrives by DPP row_shl:1), then one row minimum
//     (hop, word) lanes    provision / release of a slot window, 16 / W hops per pass
//     queue lanes          slot j * 16 + lane: lane-local minimum, then a row minimum of (time, slot)
//     (link, word) lanes   link statistics: 16 / W links per row and pass, reductions over a link's W lanes with DPP row_shl
// The rows diverge (accepted / blocked, number of releases, hops): every loop runs to the longest row and predicates the others.
// Reductions stay inside a row (full-mask DPP: quad_perm, row_half_mirror, row_mirror, row_shl), row-level votes come from
// one ballot shifted to the row's 16 bits.  The state format in HBM is the wave-per-environment kernel's: either kernel
// can continue a batch the other one stepped, and the reset kernel is shared.  The MT19937 state and the ring of pre-generated
// arrivals stay in HBM: a row reads its next arrival one step ahead, and a refill (every ~62 steps per environment) is done
// by the whole wave for one environment at a time through a per-wave LDS staging buffer (refill_requests, unchanged).
// The wave's LDS region is array-major (four occupancy bitmaps, then four link-statistics blocks, ...: OrlgParams::g_occ ...),
// as four consecutive environments lie in the HBM arrays: a quad's state moves as linear copies by all 64 lanes.
//
// Policies: the first-fit family (shortest path / shortest available path, path-only agent actions), the DeepRMSA block family
// (its two heuristics and the agent's (path, block) action), load balancing (llp_ff) and external (path, slot) actions.  Reference: the same lines of rmsa_env.py as orlg_kernels.hip cites.
#pragma once
#include "orlg_kernels.hip"

#define ORLG_GL 16  // lanes per environment (one DPP row)
#define ORLG_GE 4   // environments per wave

DEV int row_add_i32(int v) { v += dpp_xor1(v); v += dpp_xor2(v); v += dpp_half_mirror(v); v += dpp_row_mirror(v); return v; }
DEV int row_min_i32(int v) {
    int o = dpp_xor1(v); v = o < v ? o : v;
    o = dpp_xor2(v); v = o < v ? o : v;
    o = dpp_half_mirror(v); v = o < v ? o : v;
    o = dpp_row_mirror(v); return o < v ? o : v;
}
// votes of the lane's own row (16 bits)
DEV uint32_t row_ballot(bool p, int lane) { return (uint32_t)(ballot(p) >> (lane & 48)) & 0xffffu; }
// row minimum of (time, slot), ties to the lower slot: the minimum time first, then the lowest slot among the lanes that hold it
DEV void row_min_time_slot(double &t, int &q) {
    double m = t, o;
    o = ORLG_DPP_F64(dpp_xor1, m); m = o < m ? o : m;
    o = ORLG_DPP_F64(dpp_xor2, m); m = o < m ? o : m;
    o = ORLG_DPP_F64(dpp_half_mirror, m); m = o < m ? o : m;
    o = ORLG_DPP_F64(dpp_row_mirror, m); m = o < m ? o : m;
    q = row_min_i32(t == m ? q : 0x7fffffff);
    t = m;
}

// Reductions over the W consecutive lanes that hold one link's words (W is not a power of two in general: 16 / W links share a
// row): the link's FIRST lane ends with the link's result (the other lanes hold partial results nobody reads).  Sums and
// unsigned maxima only: a source past the row's end reads as 0 (bound_ctrl), their identity, so that every step is ONE
// instruction (v_add_u32_dpp / v_max_u32_dpp); a doubling tree: lanes i, i+1 -> i .. i+3 -> the rest.
#define ORLG_SEG_REDUCE(name, OP)                                                                   \
    template <int W>                                                                                \
    DEV uint32_t name(uint32_t v) {                                                                 \
        static_assert(W >= 1 && W <= 8 && W != 7, "words per link");                                \
        if (W == 1) return v;                                                                       \
        uint32_t o = (uint32_t)lane_ahead_i32<1>((int)v);                                           \
        const uint32_t s = OP(v, o);                                   /* lanes i, i+1 */           \
        if (W == 2) return s;                                                                       \
        o = (uint32_t)lane_ahead_i32<2>((int)(W == 3 ? v : s));                                     \
        const uint32_t t = OP(s, o);                                   /* lanes i .. i+2 / i+3 */   \
        if (W <= 4) return t;                                                                       \
        o = (uint32_t)lane_ahead_i32<4>((int)(W == 5 ? v : (W == 6 ? s : t)));                      \
        return OP(t, o);                                               /* lanes i .. i+W-1 */       \
    }
#define ORLG_OP_ADD(a, b) ((a) + (b))
#define ORLG_OP_MAX(a, b) ((a) > (b) ? (a) : (b))
ORLG_SEG_REDUCE(seg_add, ORLG_OP_ADD)
ORLG_SEG_REDUCE(seg_max, ORLG_OP_MAX)
#undef ORLG_SEG_REDUCE

// link statistics of up to ORLG_MAX_HOPS links per row: link_stats_update (orlg_kernels.hip) with 16 / W links per row and
// pass, one word per lane; nlinks = 0 for a row that does not take part.  links: the row's link indices (bytes, LDS).
// DEFER: the float64 part of _update_link_stats is not done here.  It is a recurrence PER LINK -- the time-weighted means of
// utilization, external fragmentation and compactness since the link's last update -- of ~100 instructions that every lane of
// the row executes for the one to three links a pass holds: 29 % of this kernel's instructions.  The deferred form logs what
// the recurrence consumes (the integers behind cur0..2 and the time: 16 bytes) per (environment, link) in HBM, counts the
// entries in the upper bits of the link's span cache, and group_link_replay works the logs off with ONE LINK PER LANE, sixteen
// links of an environment at a time, every lane through its own link's events in their order -- the same operations on the
// same values, so the same bits.
template <int W, bool LINKF, bool GRAPH, bool DEFER = false>
DEV void group_link_stats(const int lane, u64 *occ, double *lst, int32_t *lint, const Tab &tb, int S, int E, const uint8_t *links,
                          int nlinks, double now, int &sum_span, int &sum_gaps, double &comp_cur, int sum_sh, double cur_thr,
                          double &g_thr, double &g_comp, double &g_lu, uint4 *llog = nullptr, bool *need_replay = nullptr) {
    static_assert(W <= 8, "at least two links per row");
    constexpr int NS = ORLG_GL / W;  // links per row and pass
    const int gl = lane & 15;
    const int sl = gl / W, w = gl - sl * W;
    double ynow = 0.0;
    if ((LINKF || GRAPH) && nlinks > 0 && now > 0) ynow = recip_refine(now);
    for (int h0 = 0;; h0 += NS) {
        if (ballot(h0 < nlinks) == 0ull) break;
        const int h = h0 + sl;
        const bool on = sl < NS && h < nlinks;
        int link = 0;
        uint32_t packed = 0u, ilo = 0u, hi = 0u, ml = 0u;   // ilo = 0x7fff - first used slot (0: none): the minimum taken as a maximum
        if (on) link = (int)links[h];
        // the link's words sit on consecutive lanes: the neighbours' words arrive by DPP instead of further LDS reads
        u64 x = 0ull;
        if (on) x = occ[__mul24(link, W) + w];
        const u64 prev = lane_prev_u64(x);
        int e = 0;  // free slots that continue a run reaching this word's end into the next words
        if (LINKF) {
            const int lead = x == ~0ull ? 64 : ctz64(~x);  // free slots at the word's start
            // (the DPP reads stand outside any condition: a lane switched off by a branch is not a readable source)
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
        const bool first_free = x & 1ull;                                                     // meaningful on w == 0
        // slot S - 1 sits in word (S - 1) >> 6 -- not always the last of the W words (S = 400 runs on the 8-word layout)
        const uint32_t last_free_bit = w == ((S - 1) >> 6) ? (uint32_t)((x >> ((S - 1) & 63)) & 1ull) : 0u;
        // ... on the link's first lane: slot S - 1 sits in the link's last word or in the one before it (W = ceil(S / 64), 8 for 7)
        uint32_t lf = last_free_bit;
        if (W > 1) lf |= (uint32_t)lane_ahead_i32<(W > 1 ? W - 1 : 1)>((int)last_free_bit);
        if (W > 2) lf |= (uint32_t)lane_ahead_i32<(W > 2 ? W - 2 : 1)>((int)last_free_bit);
        const bool last_free = lf != 0u;
        if (on) {
            u64 u = ~x & valid_mask(S, w);
            u64 carry_f = w > 0 ? (prev >> 63) : 0ull;
            u64 carry_u = w > 0 ? ((~prev) >> 63) : 0ull;
            u64 fstarts = x & ~((x << 1) | carry_f);
            u64 ustarts = u & ~((u << 1) | carry_u);
            packed = (uint32_t)(popc64(x) | (popc64(fstarts) << 10) | (popc64(ustarts) << 20));  // free slots, free runs, used runs
            ilo = u ? (uint32_t)(0x7fff - (64 * w + ctz64(u))) : 0u;
            hi = u ? (uint32_t)(64 * w + 64 - clz64(u)) : 0u;
            if (LINKF) {
                u64 st = fstarts;
                while (st) {
                    int b = ctz64(st);
                    st &= st - 1;
                    const uint32_t len = (uint32_t)free_run_length((~x) >> b, 64 - b + e);
                    ml = len > ml ? len : ml;
                }
            }
        }
        packed = seg_add<W>(packed);
        const int lmin = 0x7fff - (int)seg_max<W>(ilo), lmax = (int)seg_max<W>(hi);
        if (LINKF) ml = seg_max<W>(ml);
        const int freec = (int)(packed & 0x3ff), F = (int)((packed >> 10) & 0x3ff), U = (int)(packed >> 20);
        const bool link_lane = on && w == 0;  // one lane per link carries on
        int dspan = 0, dgaps = 0;
        if (link_lane) {
            int nspan = U > 1 ? lmax - lmin : 0, ngaps = U > 1 ? U - 1 : 0;
            int old = lint[link];
            int cnt = 0;
            if (LINKF && DEFER) {
                // the update's inputs, for group_link_replay: free slots, max_empty, span, used runs | the time
                cnt = (int)((uint32_t)old >> 26);
                const int max_empty = (F > 1 && !(F == 2 && first_free && last_free)) ? (int)ml : 0;
                if (cnt < ORLG_LLOG_CAP - 1) {
                    llog[__mul24(link, ORLG_LLOG_CAP) + cnt] =
                        make_uint4((uint32_t)freec | ((uint32_t)max_empty << 10) | ((uint32_t)(lmax - lmin) << 20), (uint32_t)U,
                                   (uint32_t)__double2loint(now), (uint32_t)__double2hiint(now));
                    cnt += 1;
                }
                if (cnt >= ORLG_LLOG_FLUSH) *need_replay = true;
                old &= 0x03ffffff;
            }
            lint[link] = nspan | (ngaps << 16) | (cnt << 26);
            dspan = nspan - (old & 0xffff);
            dgaps = ngaps - (old >> 16);
        }
        sum_span += row_add_i32(dspan);
        sum_gaps += row_add_i32(dgaps);
        if (LINKF && !DEFER && link_lane && now > 0) {
            double *l_util = lst, *l_ef = lst + E, *l_c = lst + 2 * E, *l_lu = lst + 3 * E;
            const double last_update = l_lu[link];
            const double last0 = l_util[link], last1 = l_ef[link], last2 = l_c[link];
            const double cur0 = tb.div_s[S - freec];  // (S - free) / S
            double cur1 = 0.0, cur2 = 0.0;
            if (freec > 0) {
                int max_empty = (F > 1 && !(F == 2 && first_free && last_free)) ? (int)ml : 0;
                cur1 = 1.0 - ORLG_FDIV((double)max_empty, (double)freec);
                cur2 = U > 1 ? ORLG_FDIV((double)(lmax - lmin), (double)(S - freec)) * tb.inv_k[U] : 1.0;
            }
            const double time_diff = now - last_update;
            l_util[link] = div_by((last0 * last_update) + (cur0 * time_diff), now, ynow);
            l_ef[link] = div_by((last1 * last_update) + (cur1 * time_diff), now, ynow);
            l_c[link] = div_by((last2 * last_update) + (cur2 * time_diff), now, ynow);
        }
        if (LINKF && !DEFER && link_lane) lst[3 * E + link] = now;
        wave_sync();
    }
    if (GRAPH && nlinks > 0) {
        // _update_network_stats (rmsa_env.py:537-560), on every lane of the row
        comp_cur = network_compactness(sum_span, sum_sh, sum_gaps, E);
        if (now > 0) {
            const double time_diff = now - g_lu;
            g_thr = div_by((g_thr * g_lu) + (cur_thr * time_diff), now, ynow);
            g_comp = div_by((g_comp * g_lu) + (comp_cur * time_diff), now, ynow);
        }
        g_lu = now;
    }
}

DEV void group_link_replay(const int lane, double *lst, int32_t *lint, const Tab &tb, int S, int E, const uint4 *llog) {
    link_replay<ORLG_GL>(lane, lst, lint, tb, S, E, llog);
}

// set (release) or clear (provision) the window [s, s+n) on every link of a row's path; hops = 0: the row does not take part
template <int W>
DEV void group_apply_window(int lane, u64 *occ, const uint8_t *links, int hops, int s, int n, bool set_free) {
    constexpr int HPP = ORLG_GL / W;  // hops per pass
    const int gl = lane & 15;
    const int hs = gl / W, w = gl - hs * W;
    const u64 m = window_mask(s, n, w);
    for (int h0 = 0;; h0 += HPP) {
        if (ballot(h0 < hops) == 0ull) break;
        const int h = h0 + hs;
        if (hs < HPP && h < hops && m) {
            u64 *word = occ + __mul24((int)links[h], W) + w;
            *word = set_free ? (*word | m) : (*word & ~m);
        }
    }
    wave_sync();
}

// rows copy their environment's arrays between HBM and LDS: 16 lanes x 16 bytes per instruction and row
DEV void row_copy16(void *dst, const void *src, int bytes, int gl) {
    const int n16 = bytes >> 4;
    const uint4 *s16 = reinterpret_cast<const uint4 *>(src);
    uint4 *d16 = reinterpret_cast<uint4 *>(dst);
    for (int i = gl; i < n16; i += ORLG_GL) d16[i] = s16[i];
}
// A quad's slices of one array between HBM and LDS: `bytes` (a multiple of 8) from a wave-uniform source by all 64 lanes, 16
// bytes per lane and pass (the quad's first environment is a multiple of four: 32-byte aligned for every array copied this way)
DEV void quad_copy(void *dst, const void *src, int bytes, int lane) {
    const int n16 = bytes >> 4;
    const uint4 *s16 = reinterpret_cast<const uint4 *>(src);
    uint4 *d16 = reinterpret_cast<uint4 *>(dst);
    // four requests per lane before the first write: a quad's occupancy (3.5 KB) is one round trip, not four
    for (int i0 = 0; i0 < n16; i0 += 4 * ORLG_WAVE) {
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + k * ORLG_WAVE + lane;
            v[k] = make_uint4(0u, 0u, 0u, 0u);
            if (i < n16) v[k] = s16[i];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + k * ORLG_WAVE + lane;
            if (i < n16) d16[i] = v[k];
        }
    }
    if ((bytes & 8) && lane == 0) reinterpret_cast<u64 *>(dst)[2 * n16] = reinterpret_cast<const u64 *>(src)[2 * n16];
}
DEV void row_copy8(u64 *dst, const u64 *src, int n, int gl) {
    for (int i = gl; i < n; i += ORLG_GL) dst[i] = src[i];
}

// HBMQ: launches of very few steps (the agent-driven loop) leave the release queue where it is, in HBM: staged in LDS it is
// half of an environment's footprint there (its capacity, not its live part, sizes the region), and a one-step launch is
// bound by how many waves a CU keeps resident, not by the queue's latency.  The ring logic is the same code on
// global pointers; OrlgParams::g_wave_bytes of such a launch ends where the ring's LDS slices would begin.
template <int W, int STATS, bool HBMQ = false, bool DEFER = false>
__global__ __launch_bounds__(ORLG_WAVE *ORLG_GROUP_WAVES, (ORLG_GROUP_WAVES + 3) / 4) void orlg_rmsa_group_kernel(const OrlgParams p) {
    static_assert(!DEFER || (STATS >= 2 && !HBMQ), "the deferred link statistics belong to long launches with full statistics");
    extern __shared__ __align__(16) unsigned char smem[];
    stage_tables(smem, p);
    const int lane = threadIdx.x & 63;
    const int wib = uni((int)(threadIdx.x >> 6));
    const int g = lane >> 4, gl = lane & 15;
    const Tab tb = make_tab(smem, p);
    // one MT19937 staging buffer per workgroup (a refill happens every ~15 steps per wave and takes a fraction of a step),
    // handed from wave to wave with a lock word behind it: LDS per wave decides how many environments a CU keeps resident
    uint32_t *mt_lds = reinterpret_cast<uint32_t *>(smem + p.l_shared_bytes);
    int *mt_lock = reinterpret_cast<int *>(smem + p.l_shared_bytes + p.g_mt);
    if (threadIdx.x == 0) *mt_lock = 0;
    __syncthreads();
    unsigned char *wbase = smem + p.l_shared_bytes + p.g_mt + 16 + (size_t)wib * p.g_wave_bytes;
    // this row's environment: slice g of every array of the wave's region (array-major: OrlgParams::g_occ ...)
    u64 *occ = reinterpret_cast<u64 *>(wbase + p.g_occ) + g * p.NW;
    double *qtime = nullptr;
    uint32_t *qdesc = nullptr;
    if constexpr (!HBMQ) {
        qtime = reinterpret_cast<double *>(wbase + p.g_qtime) + g * p.Q;
        qdesc = reinterpret_cast<uint32_t *>(wbase + p.g_qdesc) + g * p.Q;
    }
    // (DEFER: the link statistics are touched by group_link_replay only, a few times per launch: they stay in HBM, and the 704
    // bytes per environment they took of the LDS buy a twelfth wave per CU)
    double *lst = DEFER ? nullptr : reinterpret_cast<double *>(wbase + p.g_lstat) + g * 4 * p.E;
    int32_t *lint = reinterpret_cast<int32_t *>(wbase + p.g_lint) + g * p.lint_stride;

    const int E = p.E, S = p.S, K = p.K, N = p.N, NBR = p.NBR, Q = p.Q, NW = p.NW;
    constexpr bool NET = STATS >= 1;
    constexpr bool FULL = STATS >= 2;
    const double INF = __longlong_as_double((long long)ORLG_INF_BITS);
    SEC_DECL_G

    // ------------------------------------------------------------------ work queue over quads of environments
    // quad q = environments 4q .. 4q+3; the first quad of a wave is its own index, the rest come from the ticket counter
    // (long launches) or by striding (short ones), as in the wave-per-environment kernel
    // A long launch hands its quads out in CHUNKS of steps (OrlgParams::n_chunks): with whole launches as tickets the last round
    // of a batch that is not a multiple of the resident waves runs at a fraction of the occupancy for a whole launch's time
    // (B = 65 536: 5.33 rounds); a chunk of a quad goes to whichever wave draws it, after the wave that ran the chunk before
    // has published the quad's state (progress[quad]; release / acquire at agent scope: another CU, maybe another XCD).
    const int n_quads = (p.B + ORLG_GE - 1) / ORLG_GE;
    const int n_chunks = p.n_chunks > 1 ? p.n_chunks : 1;
    const int n_tix = n_quads * n_chunks;
    const int n_waves = (int)(gridDim.x * (blockDim.x >> 6));
    // (with chunks EVERY ticket is drawn, a wave's first one too: a ticket that waits for its predecessor must be able to count on
    // a RUNNING wave holding it -- a statically assigned ticket of a workgroup that is not resident yet, because another kernel
    // shares the device, would be waited for by the very waves that keep that workgroup out)
    const int n_static = n_chunks > 1 ? 0 : (n_waves < n_tix ? n_waves : n_tix);
    int tix = (int)(blockIdx.x * (blockDim.x >> 6)) + wib;
    if (n_chunks == 1 && tix >= n_tix) return;
    uint32_t nxt_tk = 0;
    if (n_chunks > 1 && lane == 0) nxt_tk = atomicAdd(p.ticket, 1u);
    for (bool first = n_chunks == 1;; first = false) {
    if (!first) {
        if (p.ticket_stride) {
            tix += n_waves;
            if (tix >= n_tix) break;
        } else {
            const uint32_t tk = (uint32_t)__builtin_amdgcn_readfirstlane((int)nxt_tk) - p.ticket_base;
            if (tk >= (uint32_t)(n_tix - n_static)) break;
            tix = n_static + (int)tk;
        }
    }
    int chunk = 0, quad = tix;
    if (n_chunks > 1) { chunk = tix / n_quads; quad = tix - chunk * n_quads; }
    const int t0 = n_chunks > 1 ? chunk * p.chunk_steps : 0;   // first step of this ticket within the launch
    SEC(1);  // state load
    if (!p.ticket_stride && lane == 0) nxt_tk = atomicAdd(p.ticket, 1u);
    if (chunk > 0) {
        // the quad's state as the previous chunk left it: one relaxed poll, one acquire, the wait for its invalidate -- then
        // plain loads (MI355X_MICROARCH.md, inter-workgroup visibility)
        if (lane == 0)
            while (__hip_atomic_load(p.progress + quad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (uint32_t)chunk) __builtin_amdgcn_s_sleep(16);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const int env_raw = quad * ORLG_GE + g;
    const bool act = env_raw < p.B;          // rows past the batch's end idle (they load the last environment and store nothing)
    const int env = act ? env_raw : p.B - 1;

    // ------------------------------------------------------------------ HBM -> LDS
    // occupancy, link statistics, span caches of the quad's environments: linear copies (rows past the batch's end take a
    // copy of the last environment's slices afterwards)
    const int env0 = quad * ORLG_GE;
    const int nact = p.B - env0 < ORLG_GE ? p.B - env0 : ORLG_GE;
    quad_copy(wbase + p.g_occ, p.occ + (size_t)env0 * NW, nact * NW * 8, lane);
    if (FULL && !DEFER) quad_copy(wbase + p.g_lstat, p.lstat + (size_t)env0 * 4 * E, nact * 4 * E * 8, lane);
    // the bit-rate histograms are only ever incremented (and zeroed at an episode's end): they stay in HBM and take L2 atomics
    // without return -- 336 bytes of LDS per environment decide how many waves a CU keeps resident.  Every access
    // is an atomic, so that the updates of one address arrive at L2 in program order.
    int32_t *ghist = p.hist + (size_t)env * 4 * NBR;
    if (NET) quad_copy(wbase + p.g_lint, p.lint + (size_t)env0 * p.lint_stride, nact * p.lint_stride * 4, lane);
    if (nact < ORLG_GE) {   // the batch's last quad only
        wave_sync();
        if (!act) {
            const int gs_ = nact - 1;
            for (int i = gl; i < NW; i += ORLG_GL) occ[i] = (reinterpret_cast<u64 *>(wbase + p.g_occ) + gs_ * NW)[i];
            if (FULL && !DEFER) for (int i = gl; i < 4 * E; i += ORLG_GL) lst[i] = (reinterpret_cast<double *>(wbase + p.g_lstat) + gs_ * 4 * E)[i];
            if (NET) for (int i = gl; i < p.lint_stride; i += ORLG_GL) lint[i] = (reinterpret_cast<int32_t *>(wbase + p.g_lint) + gs_ * p.lint_stride)[i];
        }
    }
    // the link-update log of this row's environment (DEFER), and its link statistics where they live: HBM
    uint4 *llog = DEFER ? p.llog + (size_t)env * E * ORLG_LLOG_CAP : nullptr;
    if (DEFER) lst = p.lstat + (size_t)env * 4 * E;
    bool need_replay = false;
    const OrlgEnvScalars *gs = p.scal + env;
    double current_time = gs->current_time, req_arrival = gs->req_arrival, req_holding = gs->req_holding;
    double g_thr = gs->g_throughput, g_comp = gs->g_compactness, g_lu = gs->g_last_update;
    long long cnt = gs->c[gl & 7];  // counter gl & 7 (lanes 8..15 mirror lanes 0..7)
    long long sum_bitrate_running = gs->sum_bitrate_running, episodes_done = gs->episodes_done;
    int sum_sh = gs->sum_slots_hops, n_running = gs->n_running;
    int req_src = gs->req_src, req_dst = gs->req_dst, req_br = gs->req_br, req_sid = gs->req_sid;
    int mt_idx = gs->mt_idx, new_service = gs->new_service, q_overflow = gs->q_overflow;
    int ring_pos = gs->ring_pos, ring_cnt = gs->ring_cnt;
    int sum_span = gs->sum_span, sum_gaps = gs->sum_gaps;
    int eproc = (int)gs->c[2];
    wave_sync();
    double comp_cur = 1.0;
    if (NET) comp_cur = network_compactness(sum_span, sum_sh, sum_gaps, E);
    // release queue: a time-sorted ring in LDS (OrlgParams::qtime) -- q_n entries from slot q_head on; the row keeps the time of
    // its head in a register, so that a step without a due release touches no queue memory
    int q_head = gs->q_head, q_n = n_running < Q ? n_running : Q;   // (n_running also counts services an overflow lost)
    // only the live part of the ring moves between HBM and LDS (a launch of one step would otherwise spend most of its
    // traffic on empty slots); LDS slots outside it are never read
    const int q_head0 = q_head;
    int q_pops = 0;   // releases of this launch: the head may lap the ring (a 1000-step launch pops ~8 x Q entries)
    if constexpr (HBMQ) {
        qtime = p.qtime + (size_t)env * Q;
        qdesc = p.qdesc + (size_t)env * Q;
    } else {
        const double *gqt = p.qtime + (size_t)env * Q;
        const uint32_t *gqd = p.qdesc + (size_t)env * Q;
        for (int j = gl; j < q_n; j += 2 * ORLG_GL) {   // two slots per lane and pass: four requests in flight, then four writes
            int pos = q_head + j;
            pos -= pos >= Q ? Q : 0;
            int pos2 = pos + ORLG_GL;
            pos2 -= pos2 >= Q ? Q : 0;
            const bool two = j + ORLG_GL < q_n;
            const double t0 = gqt[pos];
            const uint32_t d0 = gqd[pos];
            double t1 = 0.0;
            uint32_t d1 = 0u;
            if (two) { t1 = gqt[pos2]; d1 = gqd[pos2]; }
            qtime[pos] = t0; qdesc[pos] = d0;
            if (two) { qtime[pos2] = t1; qdesc[pos2] = d1; }
        }
    }
    wave_sync();
    double next_rel = q_n > 0 ? qtime[q_head] : INF;
    const int cidx = gl & 7;
    int req_base = tb.pair_base[req_src * N + req_dst];  // first path record of the pending request's node pair

    const int n_iter = n_chunks > 1 ? (p.n_steps - t0 < p.chunk_steps ? p.n_steps - t0 : p.chunk_steps) : p.n_steps;
    const int policy = p.policy;
    for (int t = 0; t < n_iter; ++t) {
        SEC(2);  // policy
        // the arrival this step ends with is requested now (the ring entry is known unless a refill comes first)
        double pf_iat = 0.0, pf_ht = 0.0;
        uint32_t pf_rq = 0;
        const bool pf_ok = ring_cnt > 0;
        if (pf_ok) {
            const size_t ro = (size_t)env * ORLG_RING + ring_pos;
            pf_iat = p.ring_iat[ro]; pf_ht = p.ring_ht[ro]; pf_rq = p.ring_req[ro];
        }
        // ========================================================== policy: pick (path, slot)
        const int base = req_base;
        int a_path = K, a_slot = S;  // rejection (rmsa_env.py:871,913)
        int ff_n = 1, ff_hops = 0;
        if (policy == ORLG_POLICY_EXT) {
            a_path = p.actions[2 * env];
            a_slot = p.actions[2 * env + 1];
        } else {
            constexpr int PP = ORLG_GL / W;  // candidate paths per pass
            // The first-fit family: the lowest start of a free window of n slots below S - n on the first path that has one
            // (rmsa_env.py:854-913), over one given path for PathOnlyFirstFitAction (rmsa_env.py:982-1005).  The DeepRMSA family
            // (deeprmsa_env.py:48-58, rmsa_env.py:774-804): the start of the b-th free BLOCK (maximal free run) of >= n slots --
            // block 0 of the first path that has one for the heuristics, block a % j of path a / j for an agent action.
            const bool deep = policy == ORLG_POLICY_DEEP_SP || policy == ORLG_POLICY_DEEP_SAP || policy == ORLG_POLICY_DEEP_EXT;
            const bool given = policy == ORLG_POLICY_PATH_EXT || policy == ORLG_POLICY_DEEP_EXT;  // the agent names the path
            int path0 = 0, blk = 0;
            bool a_ok = true;
            if (given) {
                const int a = p.actions[env];
                if (policy == ORLG_POLICY_DEEP_EXT) {
                    a_ok = a >= 0 && a < K * p.j;
                    path0 = a_ok ? a / p.j : 0;
                    blk = a_ok ? a - path0 * p.j : 0;
                } else {
                    a_ok = a >= 0 && a < K;
                    path0 = a_ok ? a : 0;
                }
            }
            // Load balancing (least_loaded_path_first_fit, rmsa_env.py:893-937): of the paths that have a window, the one with the
            // most free slots on it (ties: the first), its first fit.
            const bool llp = policy == ORLG_POLICY_LLP;
            const int kmax = (given || policy == ORLG_POLICY_SP || policy == ORLG_POLICY_DEEP_SP) ? 1 : K;
            const int ps = gl / W, w = gl - ps * W;
            int found = 0x7fffffff, found_key = 0x7fffffff;
            for (int p0 = 0; p0 < kmax; p0 += PP) {
                if (!llp && ballot(act && a_ok && found == 0x7fffffff) == 0ull) break;
                const int pp = p0 + ps;
                const bool on = ps < PP && pp < kmax && a_ok;
                int se_pp, hops_pp;
                const u64 x = path_word_rec<W>(occ, tb.recs, base + path0 + pp, w, on, se_pp, hops_pp);
                int n = 1;
                if (on) n = tb.nslots[req_br * ORLG_NSLOT_STRIDE + se_pp];
                u64 r = run_starts<W>(x, n, w);
                const u64 xprev = lane_prev_u64(x);
                if (!deep) {
                    // start slots below S - n (exclusive: rmsa_env.py:860-871)
                    const int below = (S - n) - 64 * w;
                    r &= below >= 64 ? ~0ull : (below <= 0 ? 0ull : ((1ull << below) - 1ull));
                } else {
                    // block starts: free slots whose predecessor is not free
                    r &= x & ~((x << 1) | (w > 0 ? xprev >> 63 : 0ull));
                    if (policy == ORLG_POLICY_DEEP_EXT) {
                        // the blk-th block of the path (its words are the row's first W lanes): blocks in the words before this one
                        const int cntw = popc64(r);
                        int incl = cntw, o;
                        o = lane_back_i32<1>(incl); incl += o;
                        o = lane_back_i32<2>(incl); incl += o;
                        o = lane_back_i32<4>(incl); incl += o;
                        const int kth = blk - (incl - cntw);  // which block of this word
                        for (int q = 0; q < p.j; ++q)
                            if (q < kth) r &= r - 1;
                        if (kth < 0 || kth >= cntw) r = 0ull;
                    }
                }
                // key: (path, start slot) decide; the path's slot count and hops ride along in the low bits
                if (llp) {
                    // per path, on its first lane: free slots of the path-wide mask and its first fit (none: 0)
                    const uint32_t fs = seg_add<W>((uint32_t)popc64(x));
                    const uint32_t fit = seg_max<W>(r ? (uint32_t)(0x7fff - (64 * w + ctz64(r))) : 0u);
                    const bool head = on && w == 0 && fit != 0u;
                    const int key = head ? (int)(((1023u - fs) << 4) | (uint32_t)pp) : 0x7fffffff;   // most free slots, then lowest path
                    const int bk = row_min_i32(key);
                    const int payload = (head && key == bk) ? ((((pp << 10) | (0x7fff - (int)fit)) << 14) | (n << 4) | hops_pp) : 0x7fffffff;
                    const int bp = row_min_i32(payload);
                    if (bk < found_key) { found_key = bk; found = bp; }
                    continue;
                }
                const int cand = r ? (((((path0 + pp) << 10) | (64 * w + ctz64(r))) << 14) | (n << 4) | hops_pp) : 0x7fffffff;
                const int best = row_min_i32(cand);
                if (found == 0x7fffffff) found = best;
            }
            if (found != 0x7fffffff) { a_path = found >> 24; a_slot = (found >> 14) & 1023; ff_n = (found >> 4) & 1023; ff_hops = found & 15; }
        }

        // ========================================================== RMSAEnv.step (rmsa_env.py:222-341)
        SEC(3);  // validate + provision
        const double prev_compact = comp_cur;
        bool accepted = false;
        const bool in_range = act && a_path >= 0 && a_path < K && a_slot >= 0 && a_slot < S;
        const int gid = base + (in_range ? a_path : 0);
        const OrlgPathRec *rec = tb.recs + gid;
        int hops = ff_hops, n = ff_n;
        if (policy != ORLG_POLICY_EXT) {
            accepted = in_range;  // a first-fit result is a free window by construction
        } else {
            hops = rec->hops;
            n = tb.nslots[req_br * ORLG_NSLOT_STRIDE + rec->se];
            // is_path_free on the chosen window: word gl of the path on lane gl
            const bool on = in_range && gl < W;
            const u64 x = path_word<W>(occ, tb.recs, gid, gl < W ? gl : 0, on);
            const u64 m = on ? window_mask(a_slot, n, gl) : 0ull;
            const uint32_t bad = row_ballot((x & m) != m, lane);
            accepted = in_range && a_slot + n <= S && bad == 0u;
        }
        const int br_val = tb.bit_rates[req_br];
        // ---- _provision_path (rmsa_env.py:462-513)
        group_apply_window<W>(lane, occ, rec->link, accepted ? hops : 0, a_slot, n, false);
        if (accepted) {
            sum_sh += n * hops;
            n_running += 1;
            sum_bitrate_running += br_val;
            cnt += (cidx == 1 || cidx == 3) ? 1 : ((cidx == 5 || cidx == 7) ? br_val : 0);
            if (gl == 0) { atomicAdd(ghist + NBR + req_br, 1); atomicAdd(ghist + 3 * NBR + req_br, 1); }
        }
        SEC(4);  // statistics at provision
        if (NET)
            group_link_stats<W, FULL, true, DEFER>(lane, occ, lst, lint, tb, S, E, rec->link, accepted ? hops : 0, current_time, sum_span,
                                                   sum_gaps, comp_cur, sum_sh, (double)sum_bitrate_running, g_thr, g_comp, g_lu, llog, &need_replay);
        SEC(5);  // queue insert
        {
            // ---- _add_release (optical_network_env.py:178-189): the entries that are released later move up one slot (from the
            // top chunk of 16 down: a chunk's reads precede its writes), the new one takes the slot that opens -- each row its own
            const double rel = req_arrival + req_holding;
            bool ins = accepted && (act || !HBMQ);   // (a row past the batch's end must not touch the last environment's ring in HBM)
            if (ins && q_n >= Q) { q_overflow = 1; ins = false; }
            bool found = !ins;
            int r = 0, j0 = (q_n - 1) & ~(ORLG_GL - 1);   // q_n == 0: j0 < 0, nothing to move
            while (ballot(!found && j0 >= 0) != 0ull) {
                const bool scan = !found && j0 >= 0;
                const int j = j0 + gl;
                const bool valid = scan && j < q_n;
                int pos = q_head + j;
                pos -= pos >= Q ? Q : 0;
                double tq = 0.0;
                uint32_t dq = 0u;
                if (valid) { tq = qtime[pos]; dq = qdesc[pos]; }
                const bool later = valid && tq > rel;
                const int pos1 = pos + 1 == Q ? 0 : pos + 1;
                if (later) { qtime[pos1] = tq; qdesc[pos1] = dq; }
                const uint32_t le = row_ballot(valid && !later, lane);   // sorted: a prefix of the chunk
                if (scan) {
                    if (le) { r = j0 + __builtin_popcount(le); found = true; }
                    else j0 -= ORLG_GL;
                }
            }
            if (ins) {
                int pr = q_head + r;
                pr -= pr >= Q ? Q : 0;
                if (gl == 0) {
                    qtime[pr] = rel;
                    qdesc[pr] = (uint32_t)gid | ((uint32_t)a_slot << 14) | ((uint32_t)req_br << 24);
                }
                q_n += 1;
                next_rel = rel < next_rel ? rel : next_rel;
            }
            wave_sync();
        }

        SEC(6);  // outputs
        // per-step outputs (first lane of the row)
        if (p.out_mask && act && gl == 0) {
            const size_t o = (size_t)(t0 + t) * p.B + env;
            const int om = p.out_mask;
            if (om & (1 << ORLG_OUT_PATH)) ORLG_GPTR(int32_t, tb.outs[ORLG_OUT_PATH])[o] = a_path;
            if (om & (1 << ORLG_OUT_SLOT)) ORLG_GPTR(int32_t, tb.outs[ORLG_OUT_SLOT])[o] = a_slot;
            if (om & (1 << ORLG_OUT_ACCEPTED)) ORLG_GPTR(uint8_t, tb.outs[ORLG_OUT_ACCEPTED])[o] = accepted ? 1 : 0;
            if (om & (1 << ORLG_OUT_REWARD))
                ORLG_GPTR(double, tb.outs[ORLG_OUT_REWARD])[o] =
                    p.reward_mode == 1 ? (accepted ? 1.0 : -1.0) : (accepted ? 1.0 : 0.0);
            if (om & (1 << ORLG_OUT_REQUEST))
                ORLG_GPTR(orlg_v4i, tb.outs[ORLG_OUT_REQUEST])[o] = orlg_v4i{req_sid, req_src, req_dst, br_val};
            if (om & (1 << ORLG_OUT_ARRIVAL)) ORLG_GPTR(double, tb.outs[ORLG_OUT_ARRIVAL])[o] = req_arrival;
            if (om & (1 << ORLG_OUT_HOLDING)) ORLG_GPTR(double, tb.outs[ORLG_OUT_HOLDING])[o] = req_holding;
            if (om & (1 << ORLG_OUT_COMPACT)) ORLG_GPTR(double, tb.outs[ORLG_OUT_COMPACT])[o] = comp_cur;
            if (om & (1 << ORLG_OUT_COMPACT_DIFF))
                ORLG_GPTR(double, tb.outs[ORLG_OUT_COMPACT_DIFF])[o] = prev_compact - comp_cur;
            if (FULL && (om & (1 << ORLG_OUT_AVG_LINK_COMPACT)))
                ORLG_GPTR(double, tb.outs[ORLG_OUT_AVG_LINK_COMPACT])[o] = np_mean(lst + 2 * E, E);
            if (FULL && (om & (1 << ORLG_OUT_AVG_LINK_UTIL)))
                ORLG_GPTR(double, tb.outs[ORLG_OUT_AVG_LINK_UTIL])[o] = np_mean(lst, E);
        }
        new_service = 0;

        // ============================================================== _next_service (rmsa_env.py:643-695)
        SEC(7);  // next arrival
        {
            // a row whose ring ran dry: the whole wave generates the next ORLG_RING arrivals of that environment
            bool dry = act && ring_cnt == 0;
            for (u64 m = ballot(dry); m; m = ballot(dry)) {
                SEC(8);  // refill
                const int src_lane = ctz64(m) & 48;
                const int env_s = __builtin_amdgcn_readlane(env, src_lane);
                int idx_s = __builtin_amdgcn_readlane(mt_idx, src_lane);
                // the MT19937 state travels HBM -> registers -> (lock) LDS -> registers (unlock) -> HBM: the workgroup's staging
                // buffer is held for the regeneration and the draws only, not for the HBM round trips
                static_assert(ORLG_MT_N * 4 == 156 * 16, "MT19937 state = 156 rows of 16 bytes");
                const uint4 *g_mt = reinterpret_cast<const uint4 *>(p.mt + (size_t)env_s * ORLG_MT_N);
                uint4 *l_mt = reinterpret_cast<uint4 *>(mt_lds);
                uint4 m0 = g_mt[lane], m1 = g_mt[lane + 64], m2 = make_uint4(0u, 0u, 0u, 0u);
                if (lane < 156 - 128) m2 = g_mt[lane + 128];
                if (lane == 0) {
                    while (atomicCAS(mt_lock, 0, 1) != 0) __builtin_amdgcn_s_sleep(4);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                l_mt[lane] = m0; l_mt[lane + 64] = m1;
                if (lane < 156 - 128) l_mt[lane + 128] = m2;
                wave_sync();
                const int got = p.br_width > 0   // bit_rate_selection="continuous"
                    ? refill_requests_cont_t<false>(mt_lds, p.ring_iat + (size_t)env_s * ORLG_RING, p.ring_ht + (size_t)env_s * ORLG_RING,
                                                    p.ring_req + (size_t)env_s * ORLG_RING, tb.src_cum, tb.dst_cum, &idx_s, N, p.br_width,
                                                    p.arrival_lambda, p.holding_lambda)
                    : refill_requests(mt_lds, p.ring_iat + (size_t)env_s * ORLG_RING, p.ring_ht + (size_t)env_s * ORLG_RING,
                                      p.ring_req + (size_t)env_s * ORLG_RING, tb.src_cum, tb.dst_cum, tb.br_cum, &idx_s, N,
                                      NBR, p.arrival_lambda, p.holding_lambda, env_s);
                m0 = l_mt[lane]; m1 = l_mt[lane + 64];
                if (lane < 156 - 128) m2 = l_mt[lane + 128];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // this wave's reads of the buffer are done
                if (lane == 0) atomicExch(mt_lock, 0);
                uint4 *o_mt = reinterpret_cast<uint4 *>(p.mt + (size_t)env_s * ORLG_MT_N);
                o_mt[lane] = m0; o_mt[lane + 64] = m1;
                if (lane < 156 - 128) o_mt[lane + 128] = m2;
                // the ring entries written by other lanes are read back by this wave below: the stores have to be complete (same
                // CU: the vector cache is write-through and coherent for its own CU's stores, no L2 write-back / invalidate needed)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                wave_sync();
                if ((lane & 48) == src_lane) { ring_cnt = got; ring_pos = 0; mt_idx = idx_s; dry = false; }
            }
            SEC(7);
            double r_iat = pf_iat, r_ht = pf_ht;
            uint32_t rq = pf_rq;
            if (!pf_ok) {
                const size_t ro = (size_t)env * ORLG_RING + ring_pos;
                r_iat = p.ring_iat[ro]; r_ht = p.ring_ht[ro]; rq = p.ring_req[ro];
            }
            if (act) {
                const double at = current_time + r_iat;
                ring_pos += 1; ring_cnt -= 1;
                current_time = at;
                req_src = (int)(rq & 0xffu); req_dst = (int)((rq >> 8) & 0xffu); req_br = (int)(rq >> 16);
                req_base = tb.pair_base[req_src * N + req_dst];
                req_sid = eproc;
                new_service = 1;
                eproc += 1;
                req_arrival = at; req_holding = r_ht;
                const int bv = tb.bit_rates[req_br];
                cnt += (cidx == 0 || cidx == 2) ? 1 : ((cidx == 4 || cidx == 6) ? bv : 0);
                if (gl == 0) { atomicAdd(ghist + req_br, 1); atomicAdd(ghist + 2 * NBR + req_br, 1); }
            }

            // ---- release every service with release time <= now, in time order (rmsa_env.py:689-695): the ring's head
            bool released = false;
            for (;;) {
                SEC(9);  // release scan
                const bool rel_now = act && next_rel <= current_time;
                if (ballot(rel_now) == 0ull) break;
                SEC(10);  // release apply
                // ---- _release_path (rmsa_env.py:515-535)
                uint32_t d = 0;
                if (rel_now) d = qdesc[q_head];
                const int gid2 = (int)(d & 0x3fff), s0 = (int)((d >> 14) & 0x3ff), bri2 = (int)(d >> 24);
                const OrlgPathRec *rec2 = tb.recs + gid2;
                const int hops2 = rec2->hops;
                const int n2 = tb.nslots[bri2 * ORLG_NSLOT_STRIDE + rec2->se];
                if (rel_now) {
                    if (gl == 0) { qtime[q_head] = INF; qdesc[q_head] = 0u; }
                    q_head = q_head + 1 == Q ? 0 : q_head + 1;
                    q_pops += 1;
                    q_n -= 1;
                    n_running -= 1;
                    sum_bitrate_running -= tb.bit_rates[bri2];
                    sum_sh -= n2 * hops2;
                    released = true;
                }
                group_apply_window<W>(lane, occ, rec2->link, rel_now ? hops2 : 0, s0, n2, true);   // (ends with a wave_sync)
                if (rel_now) next_rel = q_n > 0 ? qtime[q_head] : INF;   // the next entry
                SEC(11);  // statistics at release
                if (NET)
                    group_link_stats<W, FULL, false, DEFER>(lane, occ, lst, lint, tb, S, E, rec2->link, rel_now ? hops2 : 0, current_time,
                                                            sum_span, sum_gaps, comp_cur, sum_sh, 0.0, g_thr, g_comp, g_lu, llog, &need_replay);
                if (DEFER && ballot(need_replay) != 0ull) {   // (a link's log never grows past ORLG_LLOG_FLUSH + 1 entries)
                    SEC(14);  // link replay
                    group_link_replay(lane, lst, lint, tb, S, E, llog);
                    need_replay = false;
                    SEC(11);
                }
            }
            if (NET && released) comp_cur = network_compactness(sum_span, sum_sh, sum_gaps, E);
        }

        if (DEFER && ballot(need_replay) != 0ull) {   // a link's log is filling up: every row works its logs off
            SEC(14);  // link replay
            group_link_replay(lane, lst, lint, tb, S, E, llog);
            need_replay = false;
        }
        // ============================================================== done / episode reset
        SEC(12);
        {
            const bool done = act && eproc == p.episode_length;
            if (act && gl == 0 && (p.out_mask & (1 << ORLG_OUT_DONE)))
                ORLG_GPTR(uint8_t, tb.outs[ORLG_OUT_DONE])[(size_t)(t0 + t) * p.B + env] = done ? 1 : 0;
            if (ballot(done && p.auto_reset)) {
                // reset(only_episode_counters=True) with a pending service (rmsa_env.py:343-389)
                if (done && p.auto_reset) {
                    for (int i = gl; i < NBR; i += ORLG_GL) { atomicExch(ghist + 2 * NBR + i, 0); atomicExch(ghist + 3 * NBR + i, 0); }
                    eproc = 1;
                    episodes_done += 1;
                    const int bv = tb.bit_rates[req_br];
                    if (cidx == 2) cnt = 1;
                    if (cidx == 3 || cidx == 7) cnt = 0;
                    if (cidx == 6) cnt = bv;
                }
                wave_sync();
                if (done && p.auto_reset && gl == 0) atomicExch(ghist + 2 * NBR + req_br, 1);
                wave_sync();
            }
        }
    }

    SEC(14);  // link replay
    if (DEFER) group_link_replay(lane, lst, lint, tb, S, E, llog);   // (the state that leaves carries no pending updates)
    // ------------------------------------------------------------------ LDS -> HBM
    SEC(13);  // state store
    wave_sync();
    {
        const int env0 = quad * ORLG_GE;
        const int nact = p.B - env0 < ORLG_GE ? p.B - env0 : ORLG_GE;
        quad_copy(p.occ + (size_t)env0 * NW, wbase + p.g_occ, nact * NW * 8, lane);
        if (FULL && !DEFER) quad_copy(p.lstat + (size_t)env0 * 4 * E, wbase + p.g_lstat, nact * 4 * E * 8, lane);
        if (NET) quad_copy(p.lint + (size_t)env0 * p.lint_stride, wbase + p.g_lint, nact * p.lint_stride * 4, lane);
    }
    if (act) {
        if constexpr (!HBMQ) {
            // the ring from where its head was at the start (slots popped since then hold (+inf, 0)) to its last entry; the
            // number of pops, not the head's distance modulo Q, says how far that is: a head that went round the ring has
            // emptied slots beyond (q_head - q_head0) % Q + q_n whose old entries HBM would otherwise keep
            double *gqt = p.qtime + (size_t)env * Q;
            uint32_t *gqd = p.qdesc + (size_t)env * Q;
            int span = q_pops + q_n;
            span = span > Q ? Q : span;
            for (int j = gl; j < span; j += ORLG_GL) {
                int pos = q_head0 + j;
                pos -= pos >= Q ? Q : 0;
                gqt[pos] = qtime[pos]; gqd[pos] = qdesc[pos];
            }
        }
        OrlgEnvScalars *go = p.scal + env;
        if (gl < 8) go->c[gl] = cnt;
        if (gl == 8) {
            go->current_time = current_time;
            go->req_arrival = req_arrival; go->req_holding = req_holding;
            go->g_throughput = g_thr; go->g_compactness = g_comp; go->g_last_update = g_lu;
            go->sum_bitrate_running = sum_bitrate_running;
            go->episodes_done = episodes_done;
            go->sum_slots_hops = sum_sh; go->n_running = n_running;
            go->req_src = req_src; go->req_dst = req_dst; go->req_br = req_br; go->req_sid = req_sid;
            go->mt_idx = mt_idx; go->new_service = new_service; go->q_overflow = q_overflow;
            go->ring_pos = ring_pos; go->ring_cnt = ring_cnt;
            go->sum_span = sum_span; go->sum_gaps = sum_gaps; go->q_head = q_head;
            if (q_overflow) *p.err_flag = 1;   // reported by the next entry point that waits for the stream
        }
    }
    wave_sync();
    if (n_chunks > 1) {
        // publish: this wave's stores complete, the XCD's L2 written back, then the flag (the explicit waits: the compiler may
        // drop the one behind the release)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(p.progress + quad, (uint32_t)(chunk + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    SEC(0);
    }  // work queue
    SEC_FLUSH;
}
