// orlg_phy_kernels.hip -- gfx950 kernels of the QoT-aware (PhyRMSA) step() path, physical layer.
//
// Reference: optical_rl_gym/envs/phy_rmsa_env.py -- step :272-351, _provision_path :544-623,
// _service_acceptance :767-778, _release_path :781-861, _next_service :969-1017, is_channel_free :1029-1035,
// calculate_r_cut (modified) :1123-1193, _calculate_total_cuts :1195-1203, calculate_total_r_spatial :1110-1121,
// phy_aware_bmfa_rmsa :1375-1438.  Scope = the reference's live experiment configuration
// (tests/test_rmsa_threads_us.py:133-148): grooming=False, no periodic defragmentation.
//
// Same execution model as orlg_kernels.hip: one wavefront per environment, the link x channel free bitmap
// (268 channels = 5 words per link), the release-time array and the MT19937 state live in LDS for the whole
// launch; the 32-byte service records (path, channel list) stay in HBM and are touched only on provision /
// release.  The QoT gate is the reference's: modulation_level[pair row][channel][k-path] (0 = unusable,
// capacity = level x 100 Gb/s) read from HBM in [row][k-path][channel] order (coalesced over channels).
// Lanes are channels: lane l of word w owns channel 64w + l.
#pragma once
#include "orlg_kernels.hip"

#define ORLG_PHY_MAX_CH 14
#define ORLG_PHY_MAX_K 5

struct __attribute__((aligned(16))) OrlgPhySvc {  // one running service (HBM)
    uint16_t gid;
    uint8_t nch, pad;
    uint16_t ch[ORLG_PHY_MAX_CH];
};
static_assert(sizeof(OrlgPhySvc) == 32, "OrlgPhySvc layout");

// per-env scalars in HBM (256 B)
struct __attribute__((aligned(16))) OrlgPhyScalars {
    double current_time, req_arrival, req_holding;
    double total_path_length, total_gsnr;        // per-episode sums (phy_rmsa_env.py:103-105)
    int64_t c[8];                                // orlg_counters order
    int64_t total_path_index, total_mod, channels_accepted, physical_accepted;
    int64_t episodes_done;
    int32_t n_running, req_src, req_dst, req_br, req_sid, mt_idx, new_service, q_overflow;
    int32_t pad[10];
};
static_assert(sizeof(OrlgPhyScalars) == 224, "OrlgPhyScalars layout");

enum { ORLG_PHY_POLICY_EXT = -1, ORLG_PHY_POLICY_BMFA = 0, ORLG_PHY_POLICY_BMFA_RSS = 1 };
enum { ORLG_PHY_OUT_PATH = 0, ORLG_PHY_OUT_NCH, ORLG_PHY_OUT_CHANNELS, ORLG_PHY_OUT_ACCEPTED, ORLG_PHY_OUT_DONE,
       ORLG_PHY_OUT_REQUEST, ORLG_PHY_OUT_ARRIVAL, ORLG_PHY_OUT_HOLDING, ORLG_PHY_OUT_CUTS, ORLG_PHY_OUT_RSS,
       ORLG_PHY_NUM_OUTS };

struct OrlgPhyParams {
    int32_t B, N, E, C, K, NBR, Q, NW;
    int32_t episode_length, n_steps, policy, auto_reset, mode, out_mask, num_rows, cpad;
    double arrival_lambda, holding_lambda;
    // per-env state in HBM
    uint64_t *occ;          // [B][E*W]
    double *qtime;          // [B][Q]   release times, compact: entries 0..n_running-1 are live
    OrlgPhySvc *qrec;       // [B][Q]
    uint32_t *mt;           // [B][624]
    OrlgPhyScalars *scal;   // [B]
    // shared tables
    const unsigned char *tables;   // blob staged into LDS
    int32_t tab_bytes, t_pair, t_recs, t_bitrates, t_brcum, t_srccum, t_dstcum, t_pairrow, t_adjoff, t_adj, t_sqrt,
        t_plen;
    const uint8_t *mod_t;   // [num_rows*K][cpad] modulation level per channel
    const double *gsnr_t;   // [num_rows*K][cpad]
    // per-call IO
    const int32_t *act_path;      // external actions: [B] path (-2 = blocked)
    const int16_t *act_channels;  // [B][ORLG_PHY_MAX_CH], -1 terminated
    void *outs[ORLG_PHY_NUM_OUTS];
    // per-wave LDS layout
    int32_t l_occ, l_qtime, l_mt, l_scratch, l_wsc, l_wave_bytes, l_shared_bytes, l_outs;
};

struct PhyWaveScalars {  // LDS
    int64_t c[8];
    int64_t total_path_index, total_mod, channels_accepted, physical_accepted, episodes_done;
    double total_path_length, total_gsnr, req_arrival, req_holding;
    int32_t q_overflow, pad;
};

struct PhyTab {
    const int32_t *pair_base;
    const OrlgPathRec *recs;
    const int32_t *bit_rates;
    const double *br_cum, *src_cum, *dst_cum;
    const int32_t *pair_row;
    const int32_t *adj_off;    // [num_paths+1]
    const uint16_t *adj;       // link | weight << 8
    const double *sqrt_tab;    // sqrt(k), k = 0..E*E
    const double *path_len;    // [num_paths]
    const uint64_t *outs;
};

DEV PhyTab make_phy_tab(unsigned char *smem, const OrlgPhyParams &p) {
    PhyTab tb;
    tb.pair_base = reinterpret_cast<const int32_t *>(smem + p.t_pair);
    tb.recs = reinterpret_cast<const OrlgPathRec *>(smem + p.t_recs);
    tb.bit_rates = reinterpret_cast<const int32_t *>(smem + p.t_bitrates);
    tb.br_cum = reinterpret_cast<const double *>(smem + p.t_brcum);
    tb.src_cum = reinterpret_cast<const double *>(smem + p.t_srccum);
    tb.dst_cum = reinterpret_cast<const double *>(smem + p.t_dstcum);
    tb.pair_row = reinterpret_cast<const int32_t *>(smem + p.t_pairrow);
    tb.adj_off = reinterpret_cast<const int32_t *>(smem + p.t_adjoff);
    tb.adj = reinterpret_cast<const uint16_t *>(smem + p.t_adj);
    tb.sqrt_tab = reinterpret_cast<const double *>(smem + p.t_sqrt);
    tb.path_len = reinterpret_cast<const double *>(smem + p.t_plen);
    tb.outs = reinterpret_cast<const uint64_t *>(smem + p.l_outs);
    return tb;
}

// _calculate_total_cuts (phy_rmsa_env.py:1195-1203) and calculate_total_r_spatial (:1110-1121): run-length statistics of
// every channel's column along the link axis.  Lane = channel; the link loop is wave-uniform.
template <int W>
DEV void phy_column_metrics(const u64 *occ, const double *sqrt_tab, int E, int C, int lane, double *scratch_d, bool want_cuts,
                            bool want_rss, double &cuts_out, double &rss_out) {
    int total_runs = 0;
    for (int w = 0; w < W; ++w) {
        const int ch = 64 * w + lane;
        int runs = 0, cur = 0, sumsq = 0, sum = 0;
        int prev = 0;
        for (int l = 0; l < E; ++l) {
            int b = (int)((occ[l * W + w] >> lane) & 1ull);
            runs += b & (prev ^ 1);
            if (want_rss) {
                if (b) {
                    cur += 1;
                } else {
                    sumsq += cur * cur; sum += cur; cur = 0;
                }
            }
            prev = b;
        }
        if (want_rss) {
            sumsq += cur * cur; sum += cur;
            double term = ch < C ? sqrt_tab[sumsq] / (double)(sum + 1) : 0.0;
            scratch_d[ch] = term;
        }
        if (ch >= C) runs = 0;
        // wave sum of the per-channel run counts (integers: order irrelevant)
        for (int off = 32; off > 0; off >>= 1) runs += __shfl_xor(runs, off);
        total_runs += runs;
    }
    cuts_out = (double)total_runs / (double)C;
    if (want_rss) {
        wave_sync();
        // the reference accumulates the per-channel terms in channel order in float64 (phy_rmsa_env.py:1117)
        double r = 0.0;
        for (int ch = 0; ch < C; ++ch) r += scratch_d[ch];
        rss_out = r / (double)C;
        wave_sync();
    }
    (void)want_cuts;
}

// Level and fragmentation metric of the lane's channel in every word of candidate path `idp` (level -1: not free).
//   cut (calculate_r_cut modified, phy_rmsa_env.py:1140-1193): for a channel free on the path the "cuts before minus
//   cuts after" against the links adjacent to the path's nodes reduce to  sum_j weight_j * (1 - 2 * available[link_j]);
//   rss (calculate_r_spatial, :1085-1108): sqrt(sum len^2) / (sum len + 1) over the free runs of the channel's column
//   along the link axis, after taking the channel on the path's links minus before.
template <int W>
DEV void phy_row_metrics(const u64 *occ, const PhyTab &tb, const OrlgPhyParams &p, u64 acc, int idp, int gid, int row, int lane,
                         bool rss, int (&lv)[W], double (&mt)[W]) {
    const int a0 = tb.adj_off[gid], a1 = tb.adj_off[gid + 1];
    const uint8_t *mrow = p.mod_t + (size_t)(row * p.K + idp) * p.cpad;
    // links of the path as a bit set (E <= 255: four words)
    const OrlgPathRec *rec = tb.recs + gid;
#pragma unroll
    for (int w = 0; w < W; ++w) {
        const u64 x = readlane64(acc, idp * W + w);
        lv[w] = -1; mt[w] = 0.0;
        if (x != 0ull) {
            const int ch = 64 * w + lane;
            const bool fr = ((x >> lane) & 1ull) && ch < p.C;
            const int level = (int)mrow[ch];
            double metric;
            if (!rss) {
                int m = 0;
                for (int j = a0; j < a1; ++j) {
                    const unsigned aw = tb.adj[j];
                    const int link = (int)(aw & 0xffu), wt = (int)(aw >> 8);
                    const int b = (int)((occ[__mul24(link, W) + w] >> lane) & 1ull);
                    m += wt * (1 - 2 * b);
                }
                metric = (double)m;
            } else {
                int cur0 = 0, sq0 = 0, sm0 = 0, cur1 = 0, sq1 = 0, sm1 = 0;
                u64 pm[4] = {0ull, 0ull, 0ull, 0ull};  // the path's links as a bit set (wave-uniform)
                for (int h = 0; h < rec->hops; ++h) {
                    const int pl = (int)rec->link[h];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if ((pl >> 6) == q) pm[q] |= 1ull << (pl & 63);
                }
                for (int l = 0; l < p.E; ++l) {
                    const int b = (int)((occ[__mul24(l, W) + w] >> lane) & 1ull);
                    const u64 pw = (l >> 6) == 0 ? pm[0] : (l >> 6) == 1 ? pm[1] : (l >> 6) == 2 ? pm[2] : pm[3];
                    const bool on_path = (pw >> (l & 63)) & 1ull;
                    const int b1 = on_path ? 0 : b;
                    if (b) { cur0 += 1; } else { sq0 += cur0 * cur0; sm0 += cur0; cur0 = 0; }
                    if (b1) { cur1 += 1; } else { sq1 += cur1 * cur1; sm1 += cur1; cur1 = 0; }
                }
                sq0 += cur0 * cur0; sm0 += cur0; sq1 += cur1 * cur1; sm1 += cur1;
                const double r0 = ORLG_FDIV(tb.sqrt_tab[sq0], (double)(sm0 + 1));
                const double r1 = ORLG_FDIV(tb.sqrt_tab[sq1], (double)(sm1 + 1));
                metric = r1 - r0;
            }
            if (fr) { lv[w] = level; mt[w] = metric; }
        }
    }
}

// Best remaining channel of a row in sorted order: max level, then max metric, then min channel (wave-wide).
template <int W>
DEV void phy_row_best(const int (&lv)[W], const double (&mt)[W], int lane, int &level, double &metric, int &channel) {
    int L = -1;
#pragma unroll
    for (int w = 0; w < W; ++w) L = lv[w] > L ? lv[w] : L;
    for (int off = 32; off > 0; off >>= 1) { int o = __shfl_xor(L, off); L = o > L ? o : L; }
    L = uni(L);
    level = L; metric = 0.0; channel = -1;
    if (L < 0) return;
    bool have = false;
    double M = 0.0;
#pragma unroll
    for (int w = 0; w < W; ++w)
        if (lv[w] == L && (!have || mt[w] > M)) { M = mt[w]; have = true; }
    for (int off = 32; off > 0; off >>= 1) {
        double om = __shfl_xor(M, off);
        int oh = __shfl_xor((int)have, off);
        if (oh && (!have || om > M)) { M = om; have = true; }
    }
    M = readlane_d(M, 0);
    int Cc = 0x7fffffff;
#pragma unroll
    for (int w = 0; w < W; ++w)
        if (lv[w] == L && mt[w] == M) { int c = 64 * w + lane; Cc = c < Cc ? c : Cc; }
    for (int off = 32; off > 0; off >>= 1) { int o = __shfl_xor(Cc, off); Cc = o < Cc ? o : Cc; }
    metric = M; channel = uni(Cc);
}

template <int W>
__global__ __launch_bounds__(ORLG_WAVE *ORLG_MAX_WAVES_PER_BLOCK, 2) void orlg_phy_kernel(const OrlgPhyParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(p.tables);
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        const int n16 = p.tab_bytes >> 4;
        for (int i = threadIdx.x; i < n16; i += blockDim.x) dst[i] = src[i];
#pragma unroll
        for (int i = 0; i < ORLG_PHY_NUM_OUTS; ++i)
            if ((int)threadIdx.x == i) reinterpret_cast<u64 *>(smem + p.l_outs)[i] = reinterpret_cast<u64>(p.outs[i]);
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const int wib = uni((int)(threadIdx.x >> 6));
    const int env = blockIdx.x * (int)(blockDim.x >> 6) + wib;
    if (env >= p.B) return;
    const PhyTab tb = make_phy_tab(smem, p);
    unsigned char *wb = smem + p.l_shared_bytes + (size_t)wib * p.l_wave_bytes;
    u64 *occ = reinterpret_cast<u64 *>(wb + p.l_occ);
    double *qtime = reinterpret_cast<double *>(wb + p.l_qtime);
    uint32_t *mt = reinterpret_cast<uint32_t *>(wb + p.l_mt);
    uint32_t *scratch = reinterpret_cast<uint32_t *>(wb + p.l_scratch);  // 3 KiB: scores / selection / rss terms
    PhyWaveScalars *ws = reinterpret_cast<PhyWaveScalars *>(wb + p.l_wsc);
    Wave wv;  // for draw5
    wv.lane = lane; wv.mt = mt;

    const int E = p.E, C = p.C, K = p.K, N = p.N, NBR = p.NBR, Q = p.Q, NW = p.NW;
    OrlgPhySvc *grec = p.qrec + (size_t)env * Q;

    // ------------------------------------------------------------------ HBM -> LDS
    const OrlgPhyScalars *gs = p.scal + env;
    int n_running = gs->n_running;
    {
        const u64 *g = p.occ + (size_t)env * NW;
        for (int i = lane; i < NW; i += 64) occ[i] = g[i];
        const double *gq = p.qtime + (size_t)env * Q;
        for (int i = lane; i < n_running; i += 64) qtime[i] = gq[i];
        const uint32_t *gm = p.mt + (size_t)env * ORLG_MT_N;
        for (int i = lane; i < ORLG_MT_N; i += 64) mt[i] = gm[i];
        if (lane < 8) ws->c[lane] = gs->c[lane];
        if (lane == 0) {
            ws->total_path_index = gs->total_path_index; ws->total_mod = gs->total_mod;
            ws->channels_accepted = gs->channels_accepted; ws->physical_accepted = gs->physical_accepted;
            ws->episodes_done = gs->episodes_done;
            ws->total_path_length = gs->total_path_length; ws->total_gsnr = gs->total_gsnr;
            ws->req_arrival = gs->req_arrival; ws->req_holding = gs->req_holding;
            ws->q_overflow = gs->q_overflow;
        }
    }
    double current_time = gs->current_time;
    int req_src = gs->req_src, req_dst = gs->req_dst, req_br = gs->req_br, req_sid = gs->req_sid;
    int mt_idx = gs->mt_idx, new_service = gs->new_service;
    int eproc = (int)gs->c[2];
    wave_sync();

    int *sel_ch = reinterpret_cast<int *>(scratch);            // [16] selected channels
    int *sel_mod = reinterpret_cast<int *>(scratch) + 16;      // [16] their modulation levels
    unsigned *maxword = reinterpret_cast<unsigned *>(scratch) + 32;
    double *scratch_d = reinterpret_cast<double *>(scratch + 64);  // [W*64] per-channel doubles

    const int n_iter = p.mode == ORLG_MODE_STEP ? p.n_steps : 1;
    for (int t = 0; t < n_iter; ++t) {
        if (p.mode == ORLG_MODE_STEP) {
            const int base = tb.pair_base[req_src * N + req_dst];
            const int row = tb.pair_row[req_src * N + req_dst];
            const int demand = tb.bit_rates[req_br];
            int a_path = -2, nsel = 0;

            if (p.policy == ORLG_PHY_POLICY_EXT) {
                const OrlgPhyParams __attribute__((address_space(4))) *kp =
                    (const OrlgPhyParams __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
                a_path = uni(kp->act_path[env]);
                const int16_t *ac = kp->act_channels + (size_t)env * ORLG_PHY_MAX_CH;
                if (lane < ORLG_PHY_MAX_CH) {
                    int c = ac[lane];
                    sel_ch[lane] = c;
                }
                wave_sync();
                u64 m = ballot(lane < ORLG_PHY_MAX_CH && sel_ch[lane] >= 0);
                nsel = popc64(m);  // channels are the leading non-negative entries
                if (a_path >= 0 && a_path < K && lane < nsel) {
                    int c = sel_ch[lane];
                    sel_mod[lane] = (c >= 0 && c < C) ? (int)p.mod_t[(size_t)(row * K + a_path) * p.cpad + c] : 0;
                }
                wave_sync();
            } else {
                // ---------------- phy_aware_bmfa_rmsa / phy_aware_bmfa_rss_rmsa (phy_rmsa_env.py:1375-1505), grooming off.
                // Per path ("row") the free channels are ordered by (level desc, fragmentation metric desc, channel asc)
                // = sorted(row, key=(-level, -metric)); the row with the best head (level, metric) is tried first,
                // its channels are taken in order until the bit rate is covered, otherwise the row is dropped.
                const bool rss = p.policy == ORLG_PHY_POLICY_BMFA_RSS;
                const int pp = lane / W, pw = lane - pp * W;
                const u64 acc = path_word<W>(occ, tb.recs, base + pp, pw, pp < K);
                // pass 1: head (level, metric) of every row
                int head_level[ORLG_PHY_MAX_K];
                double head_metric[ORLG_PHY_MAX_K];
                unsigned alive = 0;
#pragma unroll
                for (int idp = 0; idp < ORLG_PHY_MAX_K; ++idp) {
                    head_level[idp] = -1; head_metric[idp] = 0.0;
                    if (idp < K) {
                        int lv[W];
                        double mt[W];
                        phy_row_metrics<W>(occ, tb, p, acc, idp, base + idp, row, lane, rss, lv, mt);
                        int bl; double bm; int bc;
                        phy_row_best<W>(lv, mt, lane, bl, bm, bc);
                        if (bl >= 0) { head_level[idp] = bl; head_metric[idp] = bm; alive |= 1u << idp; }
                    }
                }
                for (;;) {
                    int best = -1, bl = -1;
                    double bm = 0.0;
#pragma unroll
                    for (int idp = 0; idp < ORLG_PHY_MAX_K; ++idp)
                        if (idp < K && ((alive >> idp) & 1u)) {
                            // row[0][0] > max_level or (row[0][0] == max_level and row[0][1] > max_metric)
                            if (best < 0 || head_level[idp] > bl || (head_level[idp] == bl && head_metric[idp] > bm)) {
                                best = idp; bl = head_level[idp]; bm = head_metric[idp];
                            }
                        }
                    if (best < 0) break;
                    // pass 2: the chosen row again, then greedy extraction in sorted order
                    int lv[W];
                    double mt[W];
                    phy_row_metrics<W>(occ, tb, p, acc, best, base + best, row, lane, rss, lv, mt);
                    int unassigned = demand;
                    nsel = 0;
                    bool covered = false;
                    while (nsel < ORLG_PHY_MAX_CH) {
                        int l0, c0;
                        double m0;
                        phy_row_best<W>(lv, mt, lane, l0, m0, c0);
                        if (l0 < 0) break;
#pragma unroll
                        for (int w = 0; w < W; ++w)
                            if (64 * w + lane == c0) lv[w] = -1;
                        if (lane == 0) { sel_ch[nsel] = c0; sel_mod[nsel] = l0; }
                        nsel += 1;
                        unassigned -= l0 * 100;
                        if (unassigned <= 0) { covered = true; break; }
                    }
                    if (covered) { a_path = best; break; }
                    alive &= ~(1u << best);  // sorted_free_channels.pop(row)
                    nsel = 0;
                }
                wave_sync();
            }

            // ========================================================== PhyRMSAEnv.step (phy_rmsa_env.py:272-351)
            bool accepted = false;
            if (a_path >= 0 && a_path < K && nsel > 0) {
                const int gid = base + a_path;
                const OrlgPathRec *rec = tb.recs + gid;
                const int hops = rec->hops;
                // is_path_free_on_channels (:1019-1027): lanes = (channel, hop) pairs
                bool bad = false;
                for (int i = lane; i < nsel * hops; i += 64) {
                    const int ci = i / hops, h = i - ci * hops;
                    const int ch = sel_ch[ci];
                    if (ch < 0 || ch >= C) { bad = true; } else {
                        bad = bad || !((occ[(int)rec->link[h] * W + (ch >> 6)] >> (ch & 63)) & 1ull);
                    }
                }
                if (ballot(bad) == 0ull) {
                    // _provision_path (:544-623): one lane per hop clears the channels on its link
                    if (lane < hops) {
                        u64 *rowp = occ + (int)rec->link[lane] * W;
                        for (int ci = 0; ci < nsel; ++ci) {
                            const int ch = sel_ch[ci];
                            rowp[ch >> 6] &= ~(1ull << (ch & 63));
                        }
                    }
                    // statistics, in channel order (the GSNR sum is a float64 accumulation)
                    if (lane == 0) {
                        const double *grow = p.gsnr_t + (size_t)(row * K + a_path) * p.cpad;
                        double tg = ws->total_gsnr;
                        long long tm = ws->total_mod;
                        for (int ci = 0; ci < nsel; ++ci) { tg += grow[sel_ch[ci]]; tm += sel_mod[ci]; }
                        ws->total_gsnr = tg; ws->total_mod = tm;
                        ws->channels_accepted += nsel;
                        // _service_acceptance(False) (:767-778)
                        ws->c[1] += 1; ws->c[3] += 1; ws->c[5] += demand; ws->c[7] += demand;
                        ws->total_path_length += tb.path_len[gid];
                        ws->total_path_index += a_path + 1;
                        ws->physical_accepted += 1;
                    }
                    accepted = true;
                    // _add_release: compact queue, append at n_running
                    if (n_running < Q) {
                        if (lane == 0) {
                            qtime[n_running] = ws->req_arrival + ws->req_holding;
                            OrlgPhySvc sv;
                            sv.gid = (uint16_t)gid; sv.nch = (uint8_t)nsel; sv.pad = 0;
                            for (int ci = 0; ci < ORLG_PHY_MAX_CH; ++ci) sv.ch[ci] = ci < nsel ? (uint16_t)sel_ch[ci] : 0xffffu;
                            grec[n_running] = sv;
                        }
                        n_running += 1;
                    } else if (lane == 0) {
                        ws->q_overflow = 1;
                    }
                    wave_sync();
                }
            }

            // per-step outputs
            if (p.out_mask) {
                const int om = p.out_mask;
                const size_t o = (size_t)t * p.B + env;
                double cuts = 0.0, rss = 0.0;
                const bool want_c = om & (1 << ORLG_PHY_OUT_CUTS), want_r = om & (1 << ORLG_PHY_OUT_RSS);
                if (want_c || want_r) phy_column_metrics<W>(occ, tb.sqrt_tab, E, C, lane, scratch_d, want_c, want_r, cuts, rss);
                if (om & (1 << ORLG_PHY_OUT_CHANNELS)) {
                    int16_t *oc = reinterpret_cast<int16_t *>(tb.outs[ORLG_PHY_OUT_CHANNELS]) + o * ORLG_PHY_MAX_CH;
                    if (lane < ORLG_PHY_MAX_CH) oc[lane] = lane < nsel ? (int16_t)sel_ch[lane] : (int16_t)-1;
                }
                if (lane == 0) {
                    if (om & (1 << ORLG_PHY_OUT_PATH)) reinterpret_cast<int32_t *>(tb.outs[ORLG_PHY_OUT_PATH])[o] = a_path;
                    if (om & (1 << ORLG_PHY_OUT_NCH)) reinterpret_cast<int32_t *>(tb.outs[ORLG_PHY_OUT_NCH])[o] = nsel;
                    if (om & (1 << ORLG_PHY_OUT_ACCEPTED)) reinterpret_cast<uint8_t *>(tb.outs[ORLG_PHY_OUT_ACCEPTED])[o] = accepted ? 1 : 0;
                    if (om & (1 << ORLG_PHY_OUT_REQUEST))
                        reinterpret_cast<int4 *>(tb.outs[ORLG_PHY_OUT_REQUEST])[o] = make_int4(req_sid, req_src, req_dst, demand);
                    if (om & (1 << ORLG_PHY_OUT_ARRIVAL)) reinterpret_cast<double *>(tb.outs[ORLG_PHY_OUT_ARRIVAL])[o] = ws->req_arrival;
                    if (om & (1 << ORLG_PHY_OUT_HOLDING)) reinterpret_cast<double *>(tb.outs[ORLG_PHY_OUT_HOLDING])[o] = ws->req_holding;
                    if (want_c) reinterpret_cast<double *>(tb.outs[ORLG_PHY_OUT_CUTS])[o] = cuts;
                    if (want_r) reinterpret_cast<double *>(tb.outs[ORLG_PHY_OUT_RSS])[o] = rss;
                }
            }
            new_service = 0;
        } else if (p.mode == ORLG_MODE_EPISODE_RESET) {
            // reset(only_episode_counters=True) (phy_rmsa_env.py:426-472)
            eproc = new_service ? 1 : 0;
            if (lane == 0) {
                ws->c[2] = new_service ? 1 : 0; ws->c[3] = 0; ws->c[6] = new_service ? tb.bit_rates[req_br] : 0; ws->c[7] = 0;
                ws->total_path_length = 0.0; ws->total_gsnr = 0.0; ws->total_path_index = 0; ws->total_mod = 0;
                ws->channels_accepted = 0; ws->physical_accepted = 0;
            }
            wave_sync();
        }

        // ============================================================== _next_service (phy_rmsa_env.py:969-1017)
        if (p.mode != ORLG_MODE_EPISODE_RESET && !new_service) {
            double u[5];
            draw5(wv, mt_idx, u);
            double uu = lane == 2 ? u[1] : u[0];
            double lam = lane == 2 ? p.holding_lambda : p.arrival_lambda;
            double ex = div_by(-orlg_log(1.0 - uu), lam, recip_refine(lam));
            double at = current_time + readlane_d(ex, 0);
            double ht = readlane_d(ex, 2);
            current_time = at;
            int src = choice_cum(tb.src_cum, N, u[2], lane);
            int dst = choice_cum(tb.dst_cum + src * N, N, u[3], lane);
            int bri = choice_cum(tb.br_cum, NBR, u[4], lane);
            req_sid = eproc;
            req_src = src; req_dst = dst; req_br = bri;
            new_service = 1;
            eproc += 1;
            if (lane == 0) {
                const int br_val = tb.bit_rates[bri];
                ws->c[0] += 1; ws->c[2] += 1; ws->c[4] += br_val; ws->c[6] += br_val;
                ws->req_arrival = at; ws->req_holding = ht;
            }
            // ---- release every service with release time <= now (:1009-1017, _release_path :781-861 with grooming off).
            // Channel frees commute, so the order of simultaneous releases is immaterial here.
            for (int q0 = 0; q0 < n_running;) {
                const int idx = q0 + lane;
                double tq = idx < n_running ? qtime[idx] : __longlong_as_double((long long)ORLG_INF_BITS);
                u64 m = ballot(tq <= current_time);
                if (!m) { q0 += 64; continue; }
                const int l = ctz64(m);
                const int victim = q0 + l;
                const OrlgPhySvc sv = grec[victim];
                const OrlgPathRec *rec = tb.recs + sv.gid;
                if (lane < rec->hops) {
                    u64 *rowp = occ + (int)rec->link[lane] * W;
                    for (int ci = 0; ci < sv.nch; ++ci) {
                        const int ch = sv.ch[ci];
                        rowp[ch >> 6] |= 1ull << (ch & 63);
                    }
                }
                // swap-remove: the last live entry takes the victim's place
                n_running -= 1;
                if (victim != n_running && lane == 0) {
                    qtime[victim] = qtime[n_running];
                    grec[victim] = grec[n_running];
                }
                wave_sync();
                // re-examine the same chunk (its slot `victim` now holds another service)
            }
        }

        if (p.mode == ORLG_MODE_STEP) {
            const bool done = (eproc == p.episode_length);
            if (lane == 0 && (p.out_mask & (1 << ORLG_PHY_OUT_DONE)))
                reinterpret_cast<uint8_t *>(tb.outs[ORLG_PHY_OUT_DONE])[(size_t)t * p.B + env] = done ? 1 : 0;
            if (done && p.auto_reset) {
                eproc = 1;
                if (lane == 0) {
                    ws->episodes_done += 1;
                    ws->c[2] = 1; ws->c[3] = 0; ws->c[6] = tb.bit_rates[req_br]; ws->c[7] = 0;
                    ws->total_path_length = 0.0; ws->total_gsnr = 0.0; ws->total_path_index = 0; ws->total_mod = 0;
                    ws->channels_accepted = 0; ws->physical_accepted = 0;
                }
                wave_sync();
            }
        }
    }

    // ------------------------------------------------------------------ LDS -> HBM
    wave_sync();
    {
        const OrlgPhyParams __attribute__((address_space(4))) *kp =
            (const OrlgPhyParams __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
        u64 *g = kp->occ + (size_t)env * NW;
        for (int i = lane; i < NW; i += 64) g[i] = occ[i];
        double *gq = kp->qtime + (size_t)env * Q;
        for (int i = lane; i < n_running; i += 64) gq[i] = qtime[i];
        uint32_t *gm = kp->mt + (size_t)env * ORLG_MT_N;
        for (int i = lane; i < ORLG_MT_N; i += 64) gm[i] = mt[i];
        OrlgPhyScalars *go = kp->scal + env;
        if (lane < 8) go->c[lane] = ws->c[lane];
        if (lane == 0) {
            go->current_time = current_time;
            go->req_arrival = ws->req_arrival; go->req_holding = ws->req_holding;
            go->total_path_length = ws->total_path_length; go->total_gsnr = ws->total_gsnr;
            go->total_path_index = ws->total_path_index; go->total_mod = ws->total_mod;
            go->channels_accepted = ws->channels_accepted; go->physical_accepted = ws->physical_accepted;
            go->episodes_done = ws->episodes_done;
            go->n_running = n_running;
            go->req_src = req_src; go->req_dst = req_dst; go->req_br = req_br; go->req_sid = req_sid;
            go->mt_idx = mt_idx; go->new_service = new_service; go->q_overflow = ws->q_overflow;
        }
    }
}
