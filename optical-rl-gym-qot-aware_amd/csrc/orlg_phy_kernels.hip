// orlg_phy_kernels.hip -- gfx950 kernels of the QoT-aware (PhyRMSA) step() path, physical layer.
//
// Reference: optical_rl_gym/envs/phy_rmsa_env.py -- step :272-351, _provision_path :544-623, _provision_virtual_path
// :625-659, _service_acceptance :767-778, _release_path :781-861, _next_service :969-1017, is_channel_free :1029-1035,
// calculate_r_cut (modified) :1123-1193, calculate_r_spatial :1085-1108, _calculate_total_cuts :1195-1203,
// calculate_total_r_spatial :1110-1121, heuristics phy_aware_sapbm_rmsa :1254, phy_aware_bmff_rmsa :1317,
// phy_aware_bmfa_rmsa :1375, phy_aware_bmfa_rss_rmsa :1441, use_existing_channels :1650, sapff_rmsa :1676.
// Periodic defragmentation (defrag_period, number_moves, metric): step :355-417, _move :662-697,
// _groom_defragmentation :703-733, _move_virtual :735-764 -- phy_defragmentation below.
//
// Same execution model as orlg_kernels.hip: one wavefront per environment, the link x channel free bitmap
// (268 channels = 5 words per link) and the MT19937 state live in LDS for the whole launch.  The release queue
// (loads of 1400-4000 = that many running services) stays in HBM -- release times and 48-byte service records,
// touched on provision / release -- and only the releases of the near future are kept in a small LDS buffer
// (NearBuffer below): the per-wave LDS footprint decides how many environments a CU keeps resident.  The QoT gate is the reference's: modulation_level[pair row][channel][k-path] (0 = unusable,
// capacity = level x 100 Gb/s) read from HBM in [row][k-path][channel] order (coalesced over channels).
// Lanes are channels: lane l of word w owns channel 64w + l.
#pragma once
#include "orlg_kernels.hip"

#define ORLG_PHY_MAX_CH 14
#define ORLG_PHY_MAX_K 5
#define ORLG_PHY_NB 128   // entries of the near-term release buffer (LDS)

struct __attribute__((aligned(16))) OrlgPhySvc {  // one running service (HBM)
    double arrival;                // service.arrival_time (age of a defragmentation candidate)
    uint32_t seq;                  // ascending seq = order of topology.graph["running_services"] (remove + append = new seq)
    uint16_t gid;
    uint8_t nch, flags;            // flags bit 0: served on the virtual layer; bit 1: source index > destination index
    uint16_t ch[ORLG_PHY_MAX_CH];  // service.channels in list order: channel | used << 9 | partial << 14  (partial: used != capacity)
};
// one entry of the per-env defragmentation work list (HBM): a candidate (diff, age, seq, idx, channel | position << 9)
// of the physical pass or a groom-eligible service (seq, idx) of the grooming pass
struct OrlgPhyCand { double diff, age; uint32_t seq; uint16_t idx, chj; uint16_t gid, pad0; uint32_t pad1; };
static_assert(sizeof(OrlgPhyCand) == 32, "OrlgPhyCand layout");
#define ORLG_CS_MAX 64             // entries per channel_state[src, dst, k-path] list: p.cs_len <= one wavefront
// one channel_state tuple (channel, used, free, capacity), 100 Gb/s units: ch | used << 9 | free << 14 | cap << 19 | 1 << 31
DEV uint32_t cs_pack(int ch, int used, int free_, int cap) {
    return (uint32_t)ch | ((uint32_t)used << 9) | ((uint32_t)free_ << 14) | ((uint32_t)cap << 19) | 0x80000000u;
}
DEV int cs_ch(uint32_t e) { return (int)(e & 0x1ffu); }
DEV int cs_used(uint32_t e) { return (int)((e >> 9) & 0x1fu); }
DEV int cs_free(uint32_t e) { return (int)((e >> 14) & 0x1fu); }
DEV int cs_cap(uint32_t e) { return (int)((e >> 19) & 0x1fu); }
static_assert(sizeof(OrlgPhySvc) == 48, "OrlgPhySvc layout");
// the head of a record in one 64-bit word (OrlgPhyParams::qsum): path, flags, channel count and the first two entries of
// service.channels (channel | used << 9 | partial << 14: 15 bits each; a service has 1.4 channels on average, the others are read
// from the record when nch > 2)
DEV u64 svc_summary(int gid, int flags, int nch, uint32_t hw0, uint32_t hw1) {
    return (u64)(uint32_t)gid | ((u64)(uint32_t)flags << 14) | ((u64)(uint32_t)nch << 16) | ((u64)(hw0 & 0x7fffu) << 20) | ((u64)(hw1 & 0x7fffu) << 35);
}
DEV int sum_gid(u64 s) { return (int)(s & 0x3fffu); }
DEV int sum_flags(u64 s) { return (int)((s >> 14) & 3u); }
DEV int sum_nch(u64 s) { return (int)((s >> 16) & 15u); }
DEV int sum_ch(u64 s, int j) { return (int)((s >> (20 + 15 * j)) & 0x7fffu); }   // j = 0, 1

// per-env scalars in HBM (256 B)
struct __attribute__((aligned(16))) OrlgPhyScalars {
    double current_time, req_arrival, req_holding;
    double total_path_length, total_gsnr;        // per-episode sums (phy_rmsa_env.py:103-105)
    int64_t c[8];                                // orlg_counters order
    int64_t total_path_index, total_mod, channels_accepted, physical_accepted;
    int64_t episodes_done;
    int32_t n_running, req_src, req_dst, req_br, req_sid, mt_idx, new_service, q_overflow;
    int32_t next_seq, counted_moves, counted_moves_groom, counted_defrag_cycles;  // phy_rmsa_env.py:110-112
    int32_t ring_pos, ring_cnt;                  // pre-generated arrivals: next entry, entries left (OrlgPhyParams::ring_*)
    int32_t pad[4];
};
static_assert(sizeof(OrlgPhyScalars) == 224, "OrlgPhyScalars layout");

// policies: ORLG_PHY_POLICY_* of include/orlg.h
enum { ORLG_PHY_OUT_PATH = 0, ORLG_PHY_OUT_NCH, ORLG_PHY_OUT_CHANNELS, ORLG_PHY_OUT_ACCEPTED, ORLG_PHY_OUT_DONE,
       ORLG_PHY_OUT_REQUEST, ORLG_PHY_OUT_ARRIVAL, ORLG_PHY_OUT_HOLDING, ORLG_PHY_OUT_CUTS, ORLG_PHY_OUT_RSS,
       ORLG_PHY_OUT_CH_USED, ORLG_PHY_OUT_DEFRAG, ORLG_PHY_OUT_GN, ORLG_PHY_NUM_OUTS };

struct OrlgPhyParams {
    int32_t B, N, E, C, K, NBR, Q, NW;
    int32_t episode_length, n_steps, policy, auto_reset, mode, out_mask, num_rows, cpad;
    int32_t grooming, cs_len;
    int32_t defrag_period, number_moves, defrag_metric /* 0 cut, 1 rss */, cand_cap;
    double arrival_lambda, holding_lambda;
    // per-env state in HBM
    uint64_t *occ;          // [B][E*W]
    double *qtime;          // [B][Q]   release times, compact: entries 0..n_running-1 are live
    OrlgPhySvc *qrec;       // [B][Q]
    uint32_t *mt;           // [B][624] MT19937 state: fetched only when an environment's arrival ring runs dry
    double *ring_iat, *ring_ht;   // [B][64] pre-generated inter-arrival / holding times, in RNG stream order (refill_requests)
    uint32_t *ring_req;           // [B][64] src | dst << 8 | bit-rate index << 16
    OrlgPhyScalars *scal;   // [B]
    uint32_t *cs;           // [B][N*N*K][cs_len] channel_state lists (virtual layer), list order = array order
    uint8_t *cs_n;          // [B][N*N*K] list lengths
    OrlgPhyCand *cand;      // [B][cand_cap] defragmentation work list (only with defrag_period > 0)
    // side arrays of the service records for the periodic defragmentation (only with defrag_period > 0, kept by the DF
    // instantiations at every site that writes a record): its scans walk 8 + 4 bytes per running service instead of 48
    uint64_t *qsum;         // [B][Q] svc_summary: gid | flags << 14 | nch << 16 | ch[0] << 20 | ch[1] << 35 (15-bit channel entries)
    uint32_t *qseq;         // [B][Q] the record's seq (list order of topology.graph["running_services"])
    const uint64_t *lvl_mask;   // [num_rows*K][32][W] channels of one modulation level on (table row, k-path), as bit masks
    uint32_t *ticket;       // work queue counter; environment = ticket - ticket_base
    uint32_t ticket_base, ticket_stride;
    // shared tables
    const unsigned char *tables;   // blob staged into LDS
    int32_t tab_bytes, t_pair, t_recs, t_bitrates, t_brcum, t_srccum, t_dstcum, t_pairrow, t_adjoff, t_adj, t_sqrt,
        t_plen, t_pathpair, t_masks;
    int32_t use_masks, pad_masks;   // E <= 32: link sets as 32-bit masks (OrlgPathMasks) instead of the adjacency CSR
    // cut metric through per-node free degrees (orlg_phy_config::path_node_weights), networks of at most 16 nodes of at most
    // 15 links each: D[channel] = 16 nibbles (nibble v = links at node v that are free on the channel) in the wave's LDS next
    // to the occupancy (l_nv), rebuilt from the occupancy at the start of every launch that evaluates the cut metric
    const uint4 *nvrec;     // [num_paths][2] node weights c (16 bytes: even nodes, then odd nodes) | wsum, cq (int16), chords
    int32_t use_nv;         // this launch keeps D (the handle has the tables and the launch's policy / defragmentation use the cut metric)
    int32_t l_nv, t_lnib, pad_nv;   // per-wave LDS offset of D; table: per link, 1 in the nibbles of its two end nodes
    // GN-model admission check of the chosen channels (include/orlg.h orlg_gn_gate), gn_on = 0: off
    int32_t gn_on, gn_nthr;
    double gn_pw, gn_bw, gn_att, gn_nf;
    const double *gn_cf;        // [C] centre frequencies
    const int32_t *gn_nspans;   // [E]
    const double *gn_spanlen;   // [E] km
    const double *gn_thr;       // [gn_nthr] dB, ascending
    // what the check evaluates that depends on the tables only, built once per handle ON THE DEVICE by orlg_gn_tables_kernel
    // with the very expressions gn_gsnr used to evaluate per check (same compiler, same libm routines: the same bits)
    const double *gn_A;         // [C][cpad] asinh(k (f_c - f_ch + bw/2)) - asinh(k (f_c - f_ch - bw/2)), 0 on the diagonal
    const double *gn_R;         // [C][cpad] bw / |f_c - f_ch|, 0 on the diagonal
    const double *gn_link;      // [E][4] l_eff, l_eff / span length, exp(2 att len) - 1, -; then [4E] = the self-channel asinh term
    // the channel-order sums of rss_total_metric, deferred (mc_flush): per env the terms at the start of a block of steps [cpad]
    // and the block's log of rewritten terms (value; channel | stamp << 16) [ORLG_RLOG_CAP each]
    double *rlog_t0, *rlog_val;
    uint32_t *rlog_key;
    double *cterm;          // [B][cpad] scratch: per-channel term of calculate_total_r_spatial while a launch keeps the per-step
                            // totals incrementally (not part of the state: rebuilt at the start of every launch that needs it)
    const uint8_t *mod_t;   // [num_rows*K][cpad] modulation level per channel
    const uint32_t *mod_k;  // [num_rows][cpad][2] the same, the levels of one channel on all K paths together (bytes 0..K-1)
    const double *gsnr_t;   // [num_rows*K][cpad]
    // per-call IO
    const int32_t *act_path;      // external actions: [B] path (-2 = blocked)
    const int16_t *act_channels;  // [B][ORLG_PHY_MAX_CH], -1 terminated; channel | used << 9 (used 0 = the full capacity)
    void *outs[ORLG_PHY_NUM_OUTS];
    int32_t *err_flag;            // the handle's sticky error word (mapped host memory): a queue / list overflow happened
    // per-wave LDS layout
    int32_t l_occ, l_nbt, l_nbi, l_scratch, l_wsc, l_wave_bytes, l_shared_bytes, l_outs;
    int32_t l_mtstage;      // the workgroup's MT19937 staging buffer (2496 B, then its lock word), after the tables
};

struct PhyWaveScalars {  // LDS
    int64_t c[8];
    int64_t total_path_index, total_mod, channels_accepted, physical_accepted, episodes_done;
    double total_path_length, total_gsnr, req_arrival, req_holding;
    int32_t q_overflow, counted_moves, counted_moves_groom, counted_defrag_cycles;
};

// the links of one path record as a bit mask over the link index (networks of at most 32 links: US14, NSFNET, JPN12):
// the RSS metric works on one channel's column along the link axis as a 32-bit vector
struct OrlgPathMasks { uint32_t path; };

struct PhyTab {
    const OrlgPathMasks *masks;
    const int32_t *pair_base;
    const OrlgPathRec *recs;
    const int32_t *bit_rates;
    const double *br_cum, *src_cum, *dst_cum;
    const int32_t *pair_row;
    const int32_t *adj_off;    // [num_paths+1]
    const uint16_t *adj;       // link | weight << 8
    const double *sqrt_tab;    // sqrt(k), k = 0..E*E
    const double *path_len;    // [num_paths]
    const uint16_t *path_pair; // [num_paths] a * N + b of the pair (a < b) the record belongs to
    const uint64_t *outs;
    const uint64_t *lnib;      // [E] 1 << 4 a | 1 << 4 b for a link a - b (only with OrlgPhyParams::use_nv)
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
    tb.path_pair = reinterpret_cast<const uint16_t *>(smem + p.t_pathpair);
    tb.masks = reinterpret_cast<const OrlgPathMasks *>(smem + p.t_masks);
    tb.outs = reinterpret_cast<const uint64_t *>(smem + p.l_outs);
    tb.lnib = reinterpret_cast<const uint64_t *>(smem + p.t_lnib);
    return tb;
}


// ---- columns as bit vectors (E <= 32).  The RSS metric and the per-step totals look at one channel's column along the LINK
// axis: bit l of col = available_channels[link l][channel].  Built once per word, a column serves every candidate path:
//   rss:  sqrt(sum len^2) / (sum len + 1) over the runs of ones of col (after: col & ~path, released: col | path)
//   cuts of a column (free runs) = popc(col & ~(col << 1))
// (the cut metric of a candidate only reads the few links adjacent to its path: it keeps its adjacency lists)
template <int W>
DEV uint32_t column_bits(const u64 *occ, int E, int w, int lane) {  // lane = channel within word w
    uint32_t col = 0u;
    for (int l = 0; l < E; ++l) col |= (uint32_t)((occ[__mul24(l, W) + w] >> lane) & 1ull) << l;
    return col;
}
DEV uint32_t lane_column_bits(const u64 *occ, int E, int W, int ch) {  // any channel, per lane
    uint32_t col = 0u;
    const int w = ch >> 6, b = ch & 63;
    for (int l = 0; l < E; ++l) col |= (uint32_t)((occ[__mul24(l, W) + w] >> b) & 1ull) << l;
    return col;
}
DEV double rss_of_column(uint32_t col, const double *sqrt_tab) {
    const int sm = __builtin_popcount(col);
    int sq = 0;
    while (col) {
        col >>= __builtin_ctz(col);
        const uint32_t inv = ~col;
        const int len = inv ? __builtin_ctz(inv) : 32;
        sq += len * len;
        col = len >= 32 ? 0u : col >> len;
    }
    return ORLG_FDIV(sqrt_tab[sq], (double)(sm + 1));
}

// float64 sum of n per-channel terms in channel order (the reference accumulates them one by one: phy_rmsa_env.py:1117) out
// of an LDS array whose entries from n up to the next multiple of 8 are zero: 8 terms per LDS round trip
DEV double ordered_sum_lds(const double *terms, int n) {
    double r = 0.0;
    const double2 *sd2 = reinterpret_cast<const double2 *>(terms);
    const int n8 = (n + 7) / 8;
    // the next eight terms are requested before the current eight are added: the additions (dependent, ~8 cycles each) hide
    // the round trip
    double2 a0 = sd2[0], a1 = sd2[1], a2 = sd2[2], a3 = sd2[3];
    for (int c8 = 1; c8 < n8; ++c8) {
        const double2 b0 = sd2[4 * c8], b1 = sd2[4 * c8 + 1], b2 = sd2[4 * c8 + 2], b3 = sd2[4 * c8 + 3];
        r += a0.x; r += a0.y; r += a1.x; r += a1.y; r += a2.x; r += a2.y; r += a3.x; r += a3.y;
        a0 = b0; a1 = b1; a2 = b2; a3 = b3;
    }
    r += a0.x; r += a0.y; r += a1.x; r += a1.y; r += a2.x; r += a2.y; r += a3.x; r += a3.y;
    return r;
}
// _calculate_total_cuts (phy_rmsa_env.py:1195-1203) and calculate_total_r_spatial (:1110-1121): run-length statistics of
// every channel's column along the link axis.  Lane = channel; the link loop is wave-uniform.
template <int W>
DEV void phy_column_metrics(const u64 *occ, const double *sqrt_tab, int E, int C, int lane, double *scratch_d, bool want_cuts,
                            bool want_rss, bool use_masks, double &cuts_out, double &rss_out, int &total_runs_out) {
    int total_runs = 0;
    for (int w = 0; w < W; ++w) {
        const int ch = 64 * w + lane;
        int runs = 0, cur = 0, sumsq = 0, sum = 0;
        int prev = 0;
        if (use_masks) {
            const uint32_t col = column_bits<W>(occ, E, w, lane);
            runs = __builtin_popcount(col & ~(col << 1));
            if (want_rss) scratch_d[ch] = ch < C ? rss_of_column(col, sqrt_tab) : 0.0;
            if (ch >= C) runs = 0;
            total_runs += wave_add_i32(runs);
            continue;
        }
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
    total_runs_out = total_runs;
    if (want_rss) {
        wave_sync();
        // the reference accumulates the per-channel terms in channel order in float64 (phy_rmsa_env.py:1117)
        rss_out = ordered_sum_lds(scratch_d, C) / (double)C;   // terms of channels >= C are zero (written above)
        wave_sync();
    }
    (void)want_cuts;
}

// ---- the cut metric through per-node free degrees (OrlgPhyParams::nv).  A path's record: c (16 node weights), wsum = sum of
// its adjacency weights, cq = c . (path links per node), its chords (links between two path nodes that are not path links).
struct NvRec { uint4 c; int wsum, cq, nchord; uint32_t cl_lo, cl_hi, cw_lo, cw_hi; };   // chord links / weights: bytes
DEV NvRec nv_unpack(const uint4 &a, const uint4 &b) {
    NvRec r;
    r.c = a;
    r.wsum = (int)(int16_t)(b.x & 0xffffu); r.cq = (int)(int16_t)(b.x >> 16);
    r.nchord = (int)(b.y & 0xffu);
    // bytes 21..25 chord links, 26..30 chord weights
    r.cl_lo = (b.y >> 8) | (b.z << 24); r.cl_hi = (b.z >> 8) & 0xffu;                 // links 0..3 | link 4
    r.cw_lo = (b.z >> 16) | (b.w << 16); r.cw_hi = (b.w >> 16) & 0xffu;               // weights 0..3 | weight 4
    return r;
}
DEV NvRec nv_load(const uint4 *nvrec, int gid) { return nv_unpack(nvrec[2 * gid], nvrec[2 * gid + 1]); }
// the record of candidate path i out of the lanes that fetched the pair's records together (lane 2 i, 2 i + 1)
DEV NvRec nv_from_lanes(const uint4 &q, int i) {
    uint4 a, b;
    a.x = (uint32_t)__builtin_amdgcn_readlane((int)q.x, 2 * i); a.y = (uint32_t)__builtin_amdgcn_readlane((int)q.y, 2 * i);
    a.z = (uint32_t)__builtin_amdgcn_readlane((int)q.z, 2 * i); a.w = (uint32_t)__builtin_amdgcn_readlane((int)q.w, 2 * i);
    b.x = (uint32_t)__builtin_amdgcn_readlane((int)q.x, 2 * i + 1); b.y = (uint32_t)__builtin_amdgcn_readlane((int)q.y, 2 * i + 1);
    b.z = (uint32_t)__builtin_amdgcn_readlane((int)q.z, 2 * i + 1); b.w = (uint32_t)__builtin_amdgcn_readlane((int)q.w, 2 * i + 1);
    return nv_unpack(a, b);
}
DEV int nv_dot(const uint4 &c, const uint4 &d) {
    uint32_t s = __builtin_amdgcn_udot4(c.x, d.x, 0u, false);
    s = __builtin_amdgcn_udot4(c.y, d.y, s, false);
    s = __builtin_amdgcn_udot4(c.z, d.z, s, false);
    return (int)__builtin_amdgcn_udot4(c.w, d.w, s, false);
}
// weighted free chords of the record on channel ch
DEV int nv_chords(const u64 *occ, const NvRec &r, int ch, int W) {
    int s = 0;
    for (int q = 0; q < r.nchord; ++q) {
        const int cl = (int)((q < 4 ? r.cl_lo >> (8 * q) : r.cl_hi) & 0xffu), cw = (int)((q < 4 ? r.cw_lo >> (8 * q) : r.cw_hi) & 0xffu);
        s += cw * (int)((occ[__mul24(cl, W) + (ch >> 6)] >> (ch & 63)) & 1ull);
    }
    return s;
}
// D of one channel as the LDS holds it (16 nibbles) -> the byte vectors the dot products take: x = nodes 0 2 4 6, y = nodes
// 8 10 12 14, z = nodes 1 3 5 7, w = nodes 9 11 13 15 (the records keep c in the same order)
DEV uint4 nv_split(u64 d) {
    const uint32_t lo = (uint32_t)d, hi = (uint32_t)(d >> 32);
    return make_uint4(lo & 0x0f0f0f0fu, hi & 0x0f0f0f0fu, (lo >> 4) & 0x0f0f0f0fu, (hi >> 4) & 0x0f0f0f0fu);
}
DEV u64 nv_nibbles(const uint4 &c) { return (u64)(c.x | (c.z << 4)) | ((u64)(c.y | (c.w << 4)) << 32); }
DEV uint4 nv_get(const u64 *dl, int ch, int C) { return nv_split(ch < C ? dl[ch] : 0ull); }
// D[ch] += c (the channel is returned on the path) or -= c (taken): nibbles never carry into their neighbours (a node has at
// least c[v] free / used links among the path's own), so one 64-bit LDS add without return does it
DEV void nv_update(u64 *dl, const uint4 &c, int ch, bool returned) {
    const u64 nb = nv_nibbles(c);
    atomicAdd(reinterpret_cast<unsigned long long *>(dl + ch), (unsigned long long)(returned ? nb : 0ull - nb));
}
// a wave reads D entries other lanes of it wrote: LDS operations of one wave complete in order
DEV void nv_fence() { wave_sync(); }
// D from the occupancy: every free link adds one to the nibbles of its two end nodes
template <int W>
DEV void nv_build(u64 *dl, const u64 *occ, const u64 *lnib, int E, int C, int lane) {
    u64 d[W];
#pragma unroll
    for (int w = 0; w < W; ++w) d[w] = 0ull;
    for (int l = 0; l < E; ++l) {
        const u64 nb = lnib[l];
        const u64 *rowp = occ + __mul24(l, W);
#pragma unroll
        for (int w = 0; w < W; ++w) d[w] += ((rowp[w] >> lane) & 1ull) ? nb : 0ull;
    }
#pragma unroll
    for (int w = 0; w < W; ++w)
        if (64 * w + lane < C) dl[64 * w + lane] = d[w];
    wave_sync();
}

// Level and fragmentation metric of the lane's channel in every word of candidate path `idp` (level -1: not free).
//   cut (calculate_r_cut modified, phy_rmsa_env.py:1140-1193): for a channel free on the path the "cuts before minus
//   cuts after" against the links adjacent to the path's nodes reduce to  sum_j weight_j * (1 - 2 * available[link_j]);
//   rss (calculate_r_spatial, :1085-1108): sqrt(sum len^2) / (sum len + 1) over the free runs of the channel's column
//   along the link axis, after taking the channel on the path's links minus before.
template <int W>
DEV void phy_row_metrics(const u64 *occ, const PhyTab &tb, const OrlgPhyParams &p, u64 acc, int idp, int gid, const uint8_t *mrow,
                         int lane, int metric_mode /* 0 cut, 1 rss, 2 none */, bool flat_level, int (&lv)[W], double (&mt)[W],
                         const uint32_t (&cols)[W], const double *r0w /* LDS [W][64]: RSS of the lane's columns as they are */,
                         const uint4 (&dv)[W] /* D of the lane's channels (cut metric with node-degree vectors) */) {
    if (p.use_masks && metric_mode == 1) {
        // the columns and their RSS as they are were built once for all candidate paths: phy_columns
        const uint32_t pmask = (uint32_t)uni((int)tb.masks[gid].path);
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const u64 x = readlane64(acc, idp * W + w);
            lv[w] = -1; mt[w] = 0.0;
            if (x != 0ull) {
                const int ch = 64 * w + lane;
                const bool fr = ((x >> lane) & 1ull) && ch < p.C;
                const double metric = rss_of_column(cols[w] & ~pmask, tb.sqrt_tab) - r0w[w * 64 + lane];
                if (fr) { lv[w] = flat_level ? 0 : (int)mrow[ch]; mt[w] = metric; }
            }
        }
        return;
    }
    if (metric_mode == 0 && p.use_nv) {
        // cut metric = wsum - 2 * (c . D[channel] - cq - free chords): four byte dot products per channel; the caller fetched D
        // of the lane's W channels once for all candidate paths
        const NvRec nr = nv_load(p.nvrec, gid);
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const u64 x = readlane64(acc, idp * W + w);
            const int ch = 64 * w + lane;
            const bool fr = ((x >> lane) & 1ull) && ch < p.C;
            lv[w] = -1; mt[w] = 0.0;
            int s = nv_dot(nr.c, dv[w]) - nr.cq;
            if (nr.nchord) s -= nv_chords(occ, nr, ch, W);
            if (fr) { lv[w] = flat_level ? 0 : (int)mrow[ch]; mt[w] = (double)(nr.wsum - 2 * s); }
        }
        return;
    }
    const int a0 = tb.adj_off[gid], a1 = tb.adj_off[gid + 1];
    if (metric_mode == 0) {
        // cut metric of every word at once: the adjacency entries sit on lanes (one LDS read), every entry then costs one
        // wave-uniform read of its link's W words -- the per-word loop of dependent LDS reads was the latency of this kernel
        int cutm[W];
#pragma unroll
        for (int w = 0; w < W; ++w) cutm[w] = 0;
        for (int j0 = a0; j0 < a1; j0 += 64) {
            const int cnt = a1 - j0 < 64 ? a1 - j0 : 64;
            const int adjv = lane < cnt ? (int)tb.adj[j0 + lane] : 0;
            for (int j = 0; j < cnt; ++j) {
                const int aw = __builtin_amdgcn_readlane(adjv, j);
                const int wt = aw >> 8;
                const u64 *rowp = occ + __mul24(aw & 0xff, W);
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    const int b = (int)((rowp[w] >> lane) & 1ull);
                    cutm[w] += wt * (1 - 2 * b);
                }
            }
        }
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const u64 x = readlane64(acc, idp * W + w);
            const int ch = 64 * w + lane;
            const bool fr = ((x >> lane) & 1ull) && ch < p.C;
            lv[w] = -1; mt[w] = 0.0;
            if (fr) { lv[w] = flat_level ? 0 : (int)mrow[ch]; mt[w] = (double)cutm[w]; }
        }
        return;
    }
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
            if (metric_mode == 2) {
                metric = 0.0;
            } else if (metric_mode == 0) {
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
            if (fr) { lv[w] = flat_level ? 0 : level; mt[w] = metric; }
        }
    }
}

// The same for the policies whose metric is an integer (cut: metric_mode 0) or absent (2): level, metric and channel of the
// lane's channel in one sortable key -- (level << 20) | (metric + 1024) << 9 | (511 - channel), -1 when the channel is not free
// on the path -- so that "best channel by (level desc, metric desc, channel asc)" is ONE integer maximum over the wave.
// |metric| <= sum of the adjacency weights < 1024 (checked at creation).
#define ORLG_PHY_KEY(level, metric, ch) (((level) << 20) | (((metric) + 1024) << 9) | (511 - (ch)))
// v_cndmask with a wave-uniform lane mask as the condition: lane l takes if_set when bit l of mask is set
DEV int select_by_lane_mask(u64 mask, int if_set, int if_clear) {
    int r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(mask));
    return r;
}
template <int W>
DEV void phy_row_keys(const u64 *occ, const PhyTab &tb, const OrlgPhyParams &p, u64 acc, int idp, int gid, const uint8_t *mrow,
                      int lane, int metric_mode /* 0 cut, 2 none */, bool flat_level, int (&key)[W],
                      const uint4 (&dv)[W] /* D of the lane's channels (cut metric with node-degree vectors) */,
                      const uint32_t (&lvk)[W] /* levels of the lane's channels on paths 0..3 (mod_k) */,
                      const uint32_t *mk_hi /* mod_k row of the lane's first channel, second word: paths 4.. */,
                      const uint4 &nvq /* lane 2 i, 2 i + 1: node record of candidate path i */) {
    // key = (level << 20) + (metric << 9) + kc, kc = (1024 << 9) | (511 - channel); bits of channels >= C are never set in the
    // occupancy (valid_mask), so "free on the path" (the lane's bit of the path's word) is the whole condition
    const int kc0 = (1024 << 9) + 511 - lane;
    const int lsh = 8 * (idp & 3);
    if (metric_mode == 0 && p.use_nv) {
        // cut metric = wsum - 2 * (c . D[channel] - cq - free chords): four byte dot products per channel
        const NvRec nr = nv_from_lanes(nvq, idp);
        const int kpath = kc0 + ((nr.wsum + 2 * nr.cq) << 9);
        int chs[W];   // weighted free chords of the lane's channels: per chord the link's W words in one go
#pragma unroll
        for (int w = 0; w < W; ++w) chs[w] = 0;
        for (int q = 0; q < nr.nchord; ++q) {
            const int cl = (int)((q < 4 ? nr.cl_lo >> (8 * q) : nr.cl_hi) & 0xffu), cw = (int)((q < 4 ? nr.cw_lo >> (8 * q) : nr.cw_hi) & 0xffu);
            const u64 *rowp = occ + __mul24(cl, W);
#pragma unroll
            for (int w = 0; w < W; ++w) {
                const u64 x = rowp[w];
                chs[w] += select_by_lane_mask(readlane64(x, 0), cw, 0);
            }
        }
#pragma unroll
        for (int w = 0; w < W; ++w) {
            int s = nv_dot(nr.c, dv[w]) - chs[w];
            const int lvl = flat_level ? 0 : (int)(((idp < 4 ? lvk[w] : mk_hi[128 * w]) >> lsh) & 0xffu);
            const int kk = (lvl << 20) + (kpath - 64 * w) - (s << 10);
            key[w] = select_by_lane_mask(readlane64(acc, idp * W + w), kk, -1);
        }
        return;
    }
    int m[W];
#pragma unroll
    for (int w = 0; w < W; ++w) m[w] = 0;
    if (metric_mode == 0) {
        // the adjacency entries sit on lanes (one LDS read), every entry then costs one wave-uniform read of its link's W words
        const int a0 = tb.adj_off[gid], a1 = tb.adj_off[gid + 1];
        for (int j0 = a0; j0 < a1; j0 += 64) {
            const int cnt = a1 - j0 < 64 ? a1 - j0 : 64;
            const int adjv = lane < cnt ? (int)tb.adj[j0 + lane] : 0;
            for (int j = 0; j < cnt; ++j) {
                const int aw = __builtin_amdgcn_readlane(adjv, j);
                const int wt = aw >> 8;
                const u64 *rowp = occ + __mul24(aw & 0xff, W);
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    const int b = (int)((rowp[w] >> lane) & 1ull);
                    m[w] += wt * (1 - 2 * b);
                }
            }
        }
    }
#pragma unroll
    for (int w = 0; w < W; ++w) {
        const int lvl = flat_level ? 0 : (int)(((idp < 4 ? lvk[w] : mk_hi[128 * w]) >> lsh) & 0xffu);
        const int kk = (lvl << 20) + (m[w] << 9) + (kc0 - 64 * w);
        key[w] = select_by_lane_mask(readlane64(acc, idp * W + w), kk, -1);
    }
}
template <int W>
DEV int phy_keys_best(const int (&key)[W]) {
    int h = key[0];
#pragma unroll
    for (int w = 1; w < W; ++w) h = key[w] > h ? key[w] : h;
    return wave_max_i32(h);
}

// the lane's channel columns of every word, built once per request for all candidate paths (mask mode only)
template <int W>
DEV void phy_columns(const u64 *occ, const PhyTab &tb, const OrlgPhyParams &p, int lane, int metric_mode, uint32_t (&cols)[W],
                     double *r0w /* LDS [W][64] */) {
#pragma unroll
    for (int w = 0; w < W; ++w) {
        cols[w] = 0u;
        if (p.use_masks && metric_mode == 1) {
            cols[w] = column_bits<W>(occ, p.E, w, lane);
            r0w[w * 64 + lane] = rss_of_column(cols[w], tb.sqrt_tab);   // read back by the same lane only
        }
    }
}

// Best remaining channel of a row in sorted order: max level, then max metric, then min channel (wave-wide).
template <int W>
DEV void phy_row_best(const int (&lv)[W], const double (&mt)[W], int lane, int &level, double &metric, int &channel) {
    int L = -1;
#pragma unroll
    for (int w = 0; w < W; ++w) L = lv[w] > L ? lv[w] : L;
    L = wave_max_i32(L);
    level = L; metric = 0.0; channel = -1;
    if (L < 0) return;
    double M = -__longlong_as_double((long long)ORLG_INF_BITS);  // lanes without a channel of that level stay at -inf
#pragma unroll
    for (int w = 0; w < W; ++w)
        if (lv[w] == L && mt[w] > M) M = mt[w];
    M = wave_max_f64(M);
    metric = M;
    // lowest channel among the ties: the first word with a match, its lowest lane
#pragma unroll
    for (int w = 0; w < W; ++w) {
        const u64 m = ballot(lv[w] == L && mt[w] == M);
        if (m) { channel = 64 * w + ctz64(m); return; }
    }
}

// ---- channel_state lists (virtual layer): one list = up to cs_len packed entries, entry i on lane i
struct CsList { uint32_t e; int n, cap; };
DEV CsList cs_load(const uint32_t *cs, const uint8_t *cs_n, int key, int lane, int cs_len) {
    CsList l;
    l.n = uni((int)cs_n[key]);
    l.cap = cs_len;
    l.e = lane < l.n ? cs[(size_t)key * cs_len + lane] : 0u;
    return l;
}
DEV void cs_store(uint32_t *cs, uint8_t *cs_n, int key, const CsList &l, int lane) {
    if (lane < l.n) cs[(size_t)key * l.cap + lane] = l.e;
    if (lane == 0) cs_n[key] = (uint8_t)l.n;
}
DEV int cs_find(const CsList &l, int ch, int lane) {  // first entry with this channel number, -1 if none
    u64 m = ballot(lane < l.n && cs_ch(l.e) == ch);
    return m ? ctz64(m) : -1;
}
DEV uint32_t cs_get(const CsList &l, int q) { return (uint32_t)__builtin_amdgcn_readlane((int)l.e, q); }
DEV void cs_remove(CsList &l, int q, int lane) {  // list.remove(entry q): later entries move up
    uint32_t nxt = (uint32_t)__shfl_down((int)l.e, 1);
    if (lane >= q) l.e = lane + 1 < l.n ? nxt : 0u;
    l.n -= 1;
}
DEV bool cs_append(CsList &l, uint32_t v, int lane) {  // list.append
    if (l.n >= l.cap) return false;
    if (lane == l.n) l.e = v;
    l.n += 1;
    return true;
}


// ---- near-term release buffer.  The release loop of _next_service (phy_rmsa_env.py:1009-1017) pops every event with
// time <= now; with ~load running services a scan of all release times per step would need them all in LDS.  Instead
// the LDS buffer holds (time, queue index) of every running service with release time <= horizon (it may hold a few
// later ones too); now <= horizon always holds when the loop looks for due services, so the buffer is all it has to
// read.  When the clock passes the horizon, or the buffer fills up, it is rebuilt from the HBM array with a horizon
// that is expected to catch half a buffer (exponential holding times: n_running * holding_lambda releases per unit time).
struct NearBuffer {
    double *t;        // [ORLG_PHY_NB] release times
    uint16_t *qi;     // [ORLG_PHY_NB] index of the service in the HBM queue
    int n;
    double horizon;
};
DEV int nb_collect(NearBuffer &nb, const double *gq, int n_running, double horizon, int lane) {
    int cnt = 0;
    for (int i0 = 0; i0 < n_running; i0 += 64) {
        const int i = i0 + lane;
        const double tq = i < n_running ? gq[i] : __longlong_as_double((long long)ORLG_INF_BITS);
        const bool in = tq <= horizon;
        const u64 m = ballot(in);
        if (m) {
            const int pos = cnt + popc64(m & ((1ull << lane) - 1ull));
            if (in && pos < ORLG_PHY_NB) { nb.t[pos] = tq; nb.qi[pos] = (uint16_t)i; }
            cnt += popc64(m);
        }
    }
    wave_sync();
    return cnt;
}
// returns false when even the services due right now do not fit (reported as a queue overflow)
DEV bool nb_rebuild(NearBuffer &nb, const double *gq, int n_running, double now, double holding_lambda, int lane) {
    double delta = (double)ORLG_PHY_NB / (2.0 * (double)(n_running > 0 ? n_running : 1) * holding_lambda);
    for (int it = 0; it < 48; ++it) {
        const double h = now + delta;
        const int cnt = nb_collect(nb, gq, n_running, h, lane);
        if (cnt <= ORLG_PHY_NB) { nb.n = cnt; nb.horizon = h; return true; }
        delta *= 0.5;
    }
    const int cnt = nb_collect(nb, gq, n_running, now, lane);
    nb.n = cnt <= ORLG_PHY_NB ? cnt : ORLG_PHY_NB;
    nb.horizon = now;
    return cnt <= ORLG_PHY_NB;
}

// channel_state list of a running service: (source, destination, k-path) key from its path record and direction flag
DEV int svc_key(const PhyTab &tb, int N, int K, int gid, int flags) {
    const int pair = tb.path_pair[gid];
    const int pa = pair / N, pb = pair - pa * N;
    const int s = (flags & 2) ? pb : pa, d = (flags & 2) ? pa : pb;
    return (s * N + d) * K + (gid - tb.pair_base[pair]);
}

// calculate_r_cut(modified=True) on ONE lane for channel `ch` of path `gid`: sum_j weight_j * (1 - 2 * available[link_j][ch])
// = cuts before minus after taking a free channel; the negative is the gain of releasing an occupied one (defrag_flag=True)
DEV int lane_cut_sum(const u64 *occ, const PhyTab &tb, int gid, int ch, int W) {
    int m = 0;
    const int w = ch >> 6, b = ch & 63;
    for (int j = tb.adj_off[gid]; j < tb.adj_off[gid + 1]; ++j) {
        const unsigned aw = tb.adj[j];
        const int bit = (int)((occ[__mul24((int)(aw & 0xffu), W) + w] >> b) & 1ull);
        m += (int)(aw >> 8) * (1 - 2 * bit);
    }
    return m;
}

// calculate_r_spatial on ONE lane (phy_rmsa_env.py:1085-1108): RSS of channel ch's column with the path's links forced
// to `force` (0: taken, 1: released = defrag_flag) minus the RSS of the column as it is
DEV double lane_rss_delta(const u64 *occ, const double *sqrt_tab, const OrlgPathRec *rec, int ch, int E, int W, int force,
                          const OrlgPathMasks *masks /* nullptr: no masks */) {
    if (masks) {
        const uint32_t col = lane_column_bits(occ, E, W, ch);
        return rss_of_column(force ? col | masks->path : col & ~masks->path, sqrt_tab) - rss_of_column(col, sqrt_tab);
    }
    u64 pm[4] = {0ull, 0ull, 0ull, 0ull};
    const int hops = rec->hops;
    for (int h = 0; h < hops; ++h) {
        const int pl = (int)rec->link[h];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if ((pl >> 6) == q) pm[q] |= 1ull << (pl & 63);
    }
    const int w = ch >> 6, bpos = ch & 63;
    int cur0 = 0, sq0 = 0, sm0 = 0, cur1 = 0, sq1 = 0, sm1 = 0;
    for (int l = 0; l < E; ++l) {
        const int b = (int)((occ[__mul24(l, W) + w] >> bpos) & 1ull);
        const u64 pw = (l >> 6) == 0 ? pm[0] : (l >> 6) == 1 ? pm[1] : (l >> 6) == 2 ? pm[2] : pm[3];
        const int b1 = ((pw >> (l & 63)) & 1ull) ? force : b;
        if (b) { cur0 += 1; } else { sq0 += cur0 * cur0; sm0 += cur0; cur0 = 0; }
        if (b1) { cur1 += 1; } else { sq1 += cur1 * cur1; sm1 += cur1; cur1 = 0; }
    }
    sq0 += cur0 * cur0; sm0 += cur0; sq1 += cur1 * cur1; sm1 += cur1;
    return ORLG_FDIV(sqrt_tab[sq1], (double)(sm1 + 1)) - ORLG_FDIV(sqrt_tab[sq0], (double)(sm0 + 1));
}

// smallest of the lanes' keys (sequence numbers: below 2^31); lanes without a key pass has = false; returns -1.0 when no lane has one
DEV double wave_min_key(uint32_t key, bool has) {
    const int m = wave_min_i32(has ? (int)key : 0x7fffffff);
    return m == 0x7fffffff ? -1.0 : (double)m;
}

// ---- per-step totals kept incrementally (networks of at most 32 links).  _calculate_total_cuts (phy_rmsa_env.py:1195-1203)
// is an integer count of free runs over all channel columns; calculate_total_r_spatial (:1110-1121) a float64 sum of one term
// per channel IN CHANNEL ORDER.  Both change only in the columns a provision / release / move touches: every such site
// subtracts the column's runs before it changes the occupancy and adds them back afterwards (mc_before / mc_after, wave
// uniform: the column = one ballot over link lanes), and rewrites the column's term; the per-step output is then the integer
// total and the ordered sum of the cached terms instead of a rebuild of all 268 columns.
#define ORLG_RLOG_CAP 384
struct MetricCache {
    bool on, want_rss;
    // the channel-order float64 sum of the RSS terms is a chain of C dependent additions per step -- a fifth of a step with the
    // metrics written.  A launch of many steps defers it: every rewritten term is logged (value, channel, the number
    // of output points passed in the block), and once per block of up to 64 steps the sums of ALL its steps are formed at once,
    // lane = step, every lane the same chain over ITS step's terms (mc_flush): C additions per block instead of per step.
    bool defer, log_overflow;
    int nlog, stamp, t0, env;   // (the log's arrays are addressed from the kernel arguments where they are used: OrlgPhyParams::rlog_*)
    int total_runs;
    double *cterm;          // HBM [cpad]
    double __attribute__((address_space(3))) *lterm;   // the same terms in LDS (mc_after<true>)
    const double *sqrt_tab;
    int E, W;
};
DEV uint32_t mc_column(const u64 *occ, const MetricCache &mc, int ch, int lane) {
    const bool bit = lane < mc.E && ((occ[__mul24(lane, mc.W) + (ch >> 6)] >> (ch & 63)) & 1ull);
    return (uint32_t)ballot(bit);
}
DEV void mc_before(const u64 *occ, MetricCache &mc, int ch, int lane) {
    if (!mc.on) return;
    const uint32_t col = mc_column(occ, mc, ch, lane);
    mc.total_runs -= __builtin_popcount(col & ~(col << 1));
}
// LT: the terms live in the wave's LDS (lterm) instead of the HBM scratch array -- the kernels whose steps leave the per-channel
// LDS scratch alone (no RSS-metric policy, no defragmentation): no HBM round trip per step for the channel-order sum
template <bool LT = false>
DEV void mc_after(const u64 *occ, MetricCache &mc, int ch, int lane) {
    if (!mc.on) return;
    const uint32_t col = mc_column(occ, mc, ch, lane);
    mc.total_runs += __builtin_popcount(col & ~(col << 1));
    if (mc.want_rss) {
        const double t = rss_of_column(col, mc.sqrt_tab);
        if (lane == 0) {
            if (LT) mc.lterm[ch] = t; else mc.cterm[ch] = t;
        }
        if (mc.defer) {
            if (mc.nlog < ORLG_RLOG_CAP) {
                if (lane == 0) {
                    const OrlgPhyParams __attribute__((address_space(4))) *kq =
                        (const OrlgPhyParams __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
                    const size_t at = (size_t)mc.env * ORLG_RLOG_CAP + mc.nlog;
                    kq->rlog_val[at] = t; kq->rlog_key[at] = (uint32_t)ch | ((uint32_t)mc.stamp << 16);
                }
                mc.nlog += 1;
            } else {
                mc.log_overflow = true;   // (reported: more than 160 terms rewritten in one step)
            }
        }
    }
}
// The deferred sums of a block: lane t = the block's step t.  Channel by channel in the reference's order (calculate_total_r_spatial
// adds the terms one by one, phy_rmsa_env.py:1117): the term as it was at the block's start, unless the log holds a rewrite
// the step has seen (stamp <= t; the last such).  Channels without a logged rewrite (a bit mask in LDS tells) cost one addition.
template <int W>
DEV void mc_flush(MetricCache &mc, const double *terms_now /* LDS [W*64] */, u64 *ormask /* LDS [W] */, int C, int cpad, int lane,
                  uint64_t out_rss, size_t B) {
    constexpr int SL = ORLG_RLOG_CAP / 64;
    const OrlgPhyParams __attribute__((address_space(4))) *kq =
        (const OrlgPhyParams __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
    const uint32_t *lkey = kq->rlog_key + (size_t)mc.env * ORLG_RLOG_CAP;
    const double *lval = kq->rlog_val + (size_t)mc.env * ORLG_RLOG_CAP;
    double *lt0 = kq->rlog_t0 + (size_t)mc.env * cpad;
    const int nb = mc.stamp, nlog = mc.nlog;
    uint32_t key[SL];
    double val[SL];
#pragma unroll
    for (int q = 0; q < SL; ++q) {
        const int i = lane + 64 * q;
        key[q] = 0xffffffffu; val[q] = 0.0;
        if (i < nlog) { key[q] = lkey[i]; val[q] = lval[i]; }
    }
    if (lane < W) ormask[lane] = 0ull;
    wave_sync();
#pragma unroll
    for (int q = 0; q < SL; ++q)
        if (lane + 64 * q < nlog) {
            const int ch = (int)(key[q] & 0xffffu);
            atomicOr(reinterpret_cast<unsigned long long *>(ormask + (ch >> 6)), 1ull << (ch & 63));
        }
    wave_sync();
    double S = 0.0;
    for (int w = 0; w < W; ++w) {
        const double t0v = lt0[64 * w + lane];   // (one round trip per word and block: a block is 64 steps)
        const u64 m = readlane64(ormask[w], 0);
        const int cmax = C - 64 * w < 64 ? C - 64 * w : 64;
        for (int c = 0; c < cmax; ++c) {
            double v = readlane_d(t0v, c);
            if ((m >> c) & 1ull) {
                const uint32_t chk = (uint32_t)(64 * w + c);
#pragma unroll
                for (int q = 0; q < SL; ++q) {
                    if (64 * q >= nlog) continue;
                    for (u64 mm = ballot((key[q] & 0xffffu) == chk); mm; mm &= mm - 1) {   // in log order: ascending lane, then slot
                        const int l = ctz64(mm);
                        const int st = (int)((uint32_t)__builtin_amdgcn_readlane((int)key[q], l) >> 16);
                        const double vv = readlane_d(val[q], l);
                        if (lane >= st) v = vv;
                    }
                }
            }
            S += v;
        }
    }
    if (lane < nb) ORLG_GPTR(double, out_rss)[(size_t)(mc.t0 + lane) * B + mc.env] = S / (double)C;
    // the next block starts from the terms as they are now (what the log held after the last output point is in them)
#pragma unroll
    for (int w = 0; w < W; ++w) lt0[64 * w + lane] = terms_now[64 * w + lane];
    mc.t0 += nb;
    mc.nlog = 0; mc.stamp = 0;
}

// The periodic defragmentation of PhyRMSAEnv.step (phy_rmsa_env.py:355-417), run when services_processed is a multiple of
// defrag_period, right after _next_service.  Two passes:
//  1. _groom_defragmentation (:703-733): a service that is the ONLY user of a partially used channel moves that share
//     onto another lit channel of its (source, destination, k-path) with enough residual capacity (_move_virtual); the
//     old channel goes dark.  The reference walks running_services / service.channels while it mutates them (remove +
//     append): the element after a moved one is skipped and the moved one is met again at the end.  Eligibility can only
//     be lost during the pass (residual capacities shrink, users are only added), so the services eligible at the start
//     -- found by all lanes in parallel -- plus the ones re-appended by a move are the only ones the walk can act on;
//     they are visited in list order (ascending seq) and re-checked exactly at their turn.
//  2. physical pass (:359-417): every channel a service fills, whose release would improve the metric, is a candidate
//     (metric gain, age); in (gain, age) order each candidate looks for a free channel of the same modulation level on
//     its path and moves there (_move, :662-697) when placing costs less than releasing gains.
template <int W>
DEV void phy_defragmentation(const OrlgPhyParams &p, const PhyTab &tb, u64 *occ, PhyWaveScalars *ws, OrlgPhySvc *grec, u64 *gsum,
                             uint32_t *gseq, uint32_t *gcs,
                             uint8_t *gcs_n, OrlgPhyCand *cand, int *lch /* LDS [16] */, double *r0w /* LDS [W][64] */, int n_running,
                             int &next_seq, double current_time, int req_src, int req_dst, int lane, u64 *gnv, MetricCache &mc SEC_PARAMS) {
    const int N = p.N, K = p.K, E = p.E;
    const bool rss = p.defrag_metric != 0;
    bool overflow = false;
    // ------------------------------------------------------------------ 1. grooming pass
    // Which services can the walk act on?  A service whose partially used channel has no other user (the list entry's `used` is
    // its own share) and whose channel_state list holds another entry with enough residual capacity.  Both are properties of the
    // LIST: (a) one pass over the lists of the environment (lane = list, coalesced) flags every entry (key, channel, used) that
    // has such a target in a small Bloom bitmap in LDS; (b) one pass over the 8-byte record summaries (lane = service) tests the
    // service's partial channels against the bitmap -- no gather per service; (c) the few that pass are resolved exactly against
    // their list (lane = service again).  The walk of round 2 read every 48-byte record and, per service with a partial channel
    // (three in four), its list: 190 KB per cycle where this reads 30.
    int n_el = 0;
    {
        // (the per-channel LDS scratch holds both: W x 512 bytes)
        constexpr int BM_WORDS = W >= 3 ? 128 : 16 * W;             // 4096 bits (512 / 1024 for one / two words of channels)
        constexpr int KU = W >= 3 ? 4 : 1;                          // summaries per lane requested at a time
        uint32_t *bm = reinterpret_cast<uint32_t *>(r0w);
        uint16_t *maybe = reinterpret_cast<uint16_t *>(bm + BM_WORDS);   // record indices that passed the bitmap
        constexpr int MAYBE_CAP = (W * 64 * 8 - BM_WORDS * 4) / 2 < 512 ? (W * 64 * 8 - BM_WORDS * 4) / 2 : 512;
        static_assert(MAYBE_CAP >= 2 * 64 * KU, "room for the services of two rounds that pass the bitmap");
        for (int q = lane; q < BM_WORDS; q += 64) bm[q] = 0u;
        wave_sync();
        auto bm_hash = [](int key, int ch, int used) { return (uint32_t)(key * 37 + ch * 11 + used * 1031) & (BM_WORDS * 32 - 1); };
        // (a) the lists
        const int n_lists = N * N * K;
        for (int k0 = 0; k0 < n_lists; k0 += 64) {
            const int key = k0 + lane;
            // (length and first eight entries requested together: the entries do not wait for the length)
            const uint32_t *lst = gcs + (size_t)(key < n_lists ? key : 0) * p.cs_len;
            const int n = key < n_lists ? (int)gcs_n[key] : 0;
            const uint4 e03 = reinterpret_cast<const uint4 *>(lst)[0], e47 = reinterpret_cast<const uint4 *>(lst)[1];
            if (n >= 2) {
                const uint32_t e8[8] = {e03.x, e03.y, e03.z, e03.w, e47.x, e47.y, e47.z, e47.w};
                if (n <= 8) {
                    // greatest and second greatest residual capacity: entry a has a target iff some OTHER entry's free >= used_a
                    int f1 = -1, f2 = -1, a1 = -1;
#pragma unroll
                    for (int t = 0; t < 8; ++t)
                        if (t < n) {
                            const int fr = cs_free(e8[t]);
                            if (fr > f1) { f2 = f1; f1 = fr; a1 = t; } else if (fr > f2) { f2 = fr; }
                        }
#pragma unroll
                    for (int t = 0; t < 8; ++t)
                        if (t < n) {
                            const int best_other = t == a1 ? f2 : f1;
                            if (best_other >= cs_used(e8[t])) {
                                const uint32_t h = bm_hash(key, cs_ch(e8[t]), cs_used(e8[t]));
                                atomicOr(bm + (h >> 5), 1u << (h & 31));
                            }
                        }
                } else {
                    int f1 = -1, f2 = -1, a1 = -1;
                    for (int t = 0; t < n; ++t) {
                        const int fr = cs_free(lst[t]);
                        if (fr > f1) { f2 = f1; f1 = fr; a1 = t; } else if (fr > f2) { f2 = fr; }
                    }
                    for (int t = 0; t < n; ++t) {
                        const uint32_t en = lst[t];
                        if ((t == a1 ? f2 : f1) >= cs_used(en)) {
                            const uint32_t h = bm_hash(key, cs_ch(en), cs_used(en));
                            atomicOr(bm + (h >> 5), 1u << (h & 31));
                        }
                    }
                }
            }
        }
        wave_sync();
        // (c) exact check of the services that passed, lane = service: as the reference's loop body up to the move
        int n_maybe = 0;
        auto resolve = [&]() {
            for (int m0 = 0; m0 < n_maybe; m0 += 64) {
                const bool on = m0 + lane < n_maybe;
                const int idx = on ? (int)maybe[m0 + lane] : 0;
                bool elig = false;
                uint32_t seq = 0u;
                int ekey = 0;
                if (on) {
                    const OrlgPhySvc *r = grec + idx;
                    // (path and direction from the summary: the list's address does not wait for the record)
                    const u64 sw = gsum[idx];
                    const int gid = sum_gid(sw), nch = sum_nch(sw), flags = sum_flags(sw);
                    seq = gseq[idx];
                    const int key = svc_key(tb, N, K, gid, flags);
                    ekey = key;
                    const uint32_t *lst = gcs + (size_t)key * p.cs_len;
                    const int n = gcs_n[key];
                    const uint4 e03 = reinterpret_cast<const uint4 *>(lst)[0], e47 = reinterpret_cast<const uint4 *>(lst)[1];
                    const uint32_t e8[8] = {e03.x, e03.y, e03.z, e03.w, e47.x, e47.y, e47.z, e47.w};
                    for (int j = 0; j < nch && !elig; ++j) {
                        const int raw = r->ch[j];
                        if (raw & (1 << 14)) {
                            const int ch = raw & 0x1ff, mine = (raw >> 9) & 0x1f;
                            bool sole = false, target = false;
#pragma unroll
                            for (int t = 0; t < 8; ++t)
                                if (t < n) {
                                    if (cs_ch(e8[t]) == ch) sole = sole || cs_used(e8[t]) == mine;
                                    else target = target || cs_free(e8[t]) >= mine;
                                }
                            for (int t = 8; t < n; t += 4) {  // four independent loads per round trip
                                uint32_t en[4];
#pragma unroll
                                for (int q = 0; q < 4; ++q) en[q] = t + q < n ? lst[t + q] : 0u;
#pragma unroll
                                for (int q = 0; q < 4; ++q)
                                    if (t + q < n) {
                                        if (cs_ch(en[q]) == ch) sole = sole || cs_used(en[q]) == mine;
                                        else target = target || cs_free(en[q]) >= mine;
                                    }
                            }
                            elig = sole && target;
                        }
                    }
                }
                const u64 m = ballot(elig);
                if (m) {
                    const int pos = n_el + popc64(m & ((1ull << lane) - 1ull));
                    if (elig && pos < p.cand_cap) { cand[pos].seq = seq; cand[pos].idx = (uint16_t)idx; cand[pos].gid = (uint16_t)ekey; }
                    n_el += popc64(m);
                }
            }
            n_maybe = 0;
            wave_sync();
        };
        // (b) the services: KU summaries per lane requested at a time
        for (int i0 = 0; i0 < n_running; i0 += 64 * KU) {
            u64 sv[KU];
#pragma unroll
            for (int k = 0; k < KU; ++k) {
                const int i = i0 + 64 * k + lane;
                sv[k] = 0ull;
                if (i < n_running) sv[k] = gsum[i];
            }
#pragma unroll
            for (int k = 0; k < KU; ++k) {
                const int i = i0 + 64 * k + lane;
                bool hit = false;
                if (i < n_running) {
                    const u64 sw = sv[k];
                    const int nch = sum_nch(sw);
                    const int h0 = sum_ch(sw, 0), h1 = nch > 1 ? sum_ch(sw, 1) : 0;
                    if (nch > 2) {
                        hit = true;     // (channels beyond the summary: looked at exactly)
                    } else if ((h0 | h1) & (1 << 14)) {
                        const int key = svc_key(tb, N, K, sum_gid(sw), sum_flags(sw));
                        if (h0 & (1 << 14)) { const uint32_t h = bm_hash(key, h0 & 0x1ff, (h0 >> 9) & 0x1f); hit = (bm[h >> 5] >> (h & 31)) & 1u; }
                        if (!hit && (h1 & (1 << 14))) { const uint32_t h = bm_hash(key, h1 & 0x1ff, (h1 >> 9) & 0x1f); hit = (bm[h >> 5] >> (h & 31)) & 1u; }
                    }
                }
                const u64 m = ballot(hit);
                if (m) {
                    if (hit) maybe[n_maybe + popc64(m & ((1ull << lane) - 1ull))] = (uint16_t)i;
                    n_maybe += popc64(m);
                }
            }
            wave_sync();
            if (n_maybe > MAYBE_CAP - 64 * KU) resolve();
        }
        if (n_maybe > 0) resolve();
    }
    if (n_el > p.cand_cap) { overflow = true; n_el = p.cand_cap; }
    int gmoves = 0;
    SEC(8);   // defragmentation: grooming walk
    {
        // The eligible services (seq, record index, list key) sit on lanes -- a cycle of the load-1400 workload has about a dozen,
        // moves re-append theirs -- and a visit requests the service's record and its channel_state list together: one HBM round
        // trip per visit.  More than a wavefront of them: the entries stay in the work list and every visit searches it.
        const bool ereg = n_el + p.number_moves <= 64;
        uint32_t eseq = 0u, ekx = 0u;    // lane e < n_el: entry e (seq; idx | key << 16)
        if (ereg && lane < n_el) { eseq = cand[lane].seq; ekx = (uint32_t)cand[lane].idx | ((uint32_t)cand[lane].gid << 16); }
        // A move makes the list iterator skip the service that FOLLOWED the moved one (it slides into its place): the walk needs
        // the successor in list order of every entry it moves -- the smallest seq above the entry's own among all running
        // services.  One pass over the dense seq array finds them all: the entries' seqs sorted in LDS, every service bisects
        // for the entry it follows and lowers that entry's successor (LDS atomic minimum).  (Round 2 searched all records after
        // every move: a third of the cycle's HBM traffic.)  Services that moved before an entry's turn lie below it in list
        // order; the re-appended ones take consecutive seqs from ns_first on and follow every original service.
        uint32_t esucc = 0xffffffffu;
        const int ns_first = next_seq;
        if (ereg && n_el > 0) {
            int erank = 0;
            for (int l2 = 0; l2 < n_el; ++l2) erank += ((uint32_t)__builtin_amdgcn_readlane((int)eseq, l2) < eseq) ? 1 : 0;
            uint32_t *ss = reinterpret_cast<uint32_t *>(r0w), *sx = ss + 64;   // [64] sorted seqs, [64] their successors
            if (lane < n_el) ss[erank] = eseq;
            sx[lane] = 0xffffffffu;
            wave_sync();
            const int steps = 32 - __builtin_clz((unsigned)n_el);   // bisection over 0 .. n_el
            for (int i0 = 0; i0 < n_running; i0 += 256) {
                uint32_t v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = i0 + 64 * k + lane;
                    v[k] = 0u;   // (below every entry: follows none)
                    if (i < n_running) v[k] = gseq[i];
                }
                int lo[4], hi[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) { lo[k] = 0; hi[k] = n_el; }
                for (int it = 0; it < steps; ++it) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {   // entries with a seq below v[k]: the first `lo` of the sorted ones
                        const int mid = (lo[k] + hi[k]) >> 1;
                        const bool open = lo[k] < hi[k];
                        const uint32_t sm_ = ss[open ? mid : 0];
                        if (open) { if (sm_ < v[k]) lo[k] = mid + 1; else hi[k] = mid; }
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (lo[k] > 0) atomicMin(sx + (lo[k] - 1), v[k]);
            }
            wave_sync();
            if (lane < n_el) esucc = sx[erank];
            wave_sync();
        }
        long long cursor = -1;
        bool stop = p.number_moves == 0;  // the reference returns at its first check
        for (int visit = 0; visit < 2 * p.cand_cap && !stop; ++visit) {  // every visit moves the cursor up the list
            uint32_t seq0;
            int idx, key;
            if (ereg) {
                const bool has = lane < n_el && (long long)eseq > cursor;
                const double kmin = wave_min_key(eseq, has);
                if (kmin < 0.0) break;
                seq0 = (uint32_t)kmin;
                const uint32_t kx = (uint32_t)__builtin_amdgcn_readlane((int)ekx, ctz64(ballot(has && eseq == seq0)));
                idx = (int)(kx & 0xffffu); key = (int)(kx >> 16);
            } else {
                uint32_t bs = 0u;
                int bi = -1;
                for (int c = lane; c < n_el; c += 64) {
                    const uint32_t sq = cand[c].seq;
                    if ((long long)sq > cursor && (bi < 0 || sq < bs)) { bs = sq; bi = (int)((uint32_t)cand[c].idx | ((uint32_t)cand[c].gid << 16)); }
                }
                const double kmin = wave_min_key(bs, bi >= 0);
                if (kmin < 0.0) break;
                seq0 = (uint32_t)kmin;
                const uint32_t kx = (uint32_t)__builtin_amdgcn_readlane(bi, ctz64(ballot(bi >= 0 && bs == seq0)));
                idx = (int)(kx & 0xffffu); key = (int)(kx >> 16);
            }
            const OrlgPhySvc *r = grec + idx;
            // record and list, requested together
            const uint32_t d3 = reinterpret_cast<const uint32_t *>(r)[3];   // gid | nch << 16 | flags << 24
            const int chl = lane < ORLG_PHY_MAX_CH ? (int)r->ch[lane] : 0xffff;
            CsList l = cs_load(gcs, gcs_n, key, lane, p.cs_len);
            const int gid = uni((int)(d3 & 0xffffu)), nch = uni((int)((d3 >> 16) & 0xffu)), flags = uni((int)(d3 >> 24));
            if (lane < ORLG_PHY_MAX_CH) lch[lane] = lane < nch ? chl : 0xffff;
            wave_sync();
            const OrlgPathRec *rec = tb.recs + gid;
            bool moved = false;
            for (int j = 0; j < nch; ++j) {  // the list keeps its length: every move is remove + append
                const int raw = lch[j];
                if (raw & (1 << 14)) {
                    const int ch = raw & 0x1ff, mine = (raw >> 9) & 0x1f;
                    const int q = cs_find(l, ch, lane);
                    if (q >= 0 && cs_used(cs_get(l, q)) == mine) {
                        const u64 tm = ballot(lane < l.n && cs_ch(l.e) != ch && cs_free(l.e) >= mine);
                        if (tm) {
                            const uint32_t tg = cs_get(l, ctz64(tm));
                            cs_remove(l, ctz64(tm), lane);
                            cs_remove(l, cs_find(l, ch, lane), lane);
                            cs_append(l, cs_pack(cs_ch(tg), cs_used(tg) + mine, cs_free(tg) - mine, cs_cap(tg)), lane);
                            cs_store(gcs, gcs_n, key, l, lane);   // (the list stays on lanes for the service's other channels)
                            // _move_virtual (:735-764): the old channel goes dark on the path, the list entry moves to the end
                            mc_before(occ, mc, ch, lane);
                            if (lane < rec->hops) occ[(int)rec->link[lane] * W + (ch >> 6)] |= 1ull << (ch & 63);
                            if (gnv && lane == 0) nv_update(gnv, p.nvrec[2 * gid], ch, true);
                            wave_sync();
                            mc_after(occ, mc, ch, lane);
                            const int nxt = (lane >= j && lane + 1 < nch) ? lch[lane + 1] : 0;
                            wave_sync();
                            if (lane >= j && lane + 1 < nch) lch[lane] = nxt;
                            if (lane == nch - 1) lch[lane] = cs_ch(tg) | (mine << 9) | (1 << 14);
                            wave_sync();
                            moved = true;
                            gmoves += 1;
                        }
                    }
                }
                if (gmoves == p.number_moves) { stop = true; break; }
            }
            if (moved) {
                const int ns = next_seq;
                next_seq += 1;
                if (lane < nch) grec[idx].ch[lane] = (uint16_t)lch[lane];
                if (lane == 0) {
                    grec[idx].seq = (uint32_t)ns;
                    gsum[idx] = svc_summary(gid, flags, nch, (uint32_t)lch[0], nch > 1 ? (uint32_t)lch[1] : 0u);
                    gseq[idx] = (uint32_t)ns;
                }
                // the list iterator skips the service that followed this one (it slid into its place): the smallest seq above
                // seq0, from the dense seq array -- eight coalesced requests per lane in flight (the strided reads of the 48-byte
                // records were a third of the defragmentation's HBM traffic)
                uint32_t sm = 0u;
                bool hs = false;
                if (ereg) {
                    // the entry's successor from the table; none: it was the list's last service (then the first re-appended one
                    // follows, or it follows itself), or a re-appended one (consecutive seqs)
                    uint32_t sc = 0xffffffffu;
                    if (seq0 < (uint32_t)ns_first) sc = (uint32_t)__builtin_amdgcn_readlane((int)esucc, ctz64(ballot(lane < n_el && eseq == seq0)));
                    if (sc == 0xffffffffu) sc = seq0 < (uint32_t)ns_first ? (uint32_t)ns_first : seq0 + 1u;   // (<= ns: ns is this service's own new seq)
                    sm = sc; hs = true;
                } else if (!stop) {   // (the walk is over with the last move: nobody asks for the cursor)
                    for (int i0 = 0; i0 < n_running; i0 += 512) {
                        uint32_t v[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const int i = i0 + 64 * k + lane;
                            v[k] = 0u;
                            if (i < n_running && i != idx) v[k] = gseq[i];
                        }
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const int i = i0 + 64 * k + lane;
                            const uint32_t sq = i == idx ? (uint32_t)ns : v[k];   // (this service's own new key: not read back)
                            if (i < n_running && sq > seq0 && (!hs || sq < sm)) { sm = sq; hs = true; }
                        }
                    }
                }
                {
                    const double nk = wave_min_key(sm, hs);   // this service itself carries a later key: never "none"
                    cursor = nk < 0.0 ? (long long)seq0 : (long long)nk;
                }
                if (ereg) {
                    if (lane == n_el) { eseq = (uint32_t)ns; ekx = (uint32_t)idx | ((uint32_t)key << 16); }
                    n_el += 1;
                } else if (n_el < p.cand_cap) {
                    if (lane == 0) { cand[n_el].seq = (uint32_t)ns; cand[n_el].idx = (uint16_t)idx; cand[n_el].gid = (uint16_t)key; }
                    n_el += 1;
                } else {
                    overflow = true;
                }
            } else {
                cursor = (long long)seq0;
            }
            wave_sync();
        }
    }
    int cmoves = 0, cycles = 0;
    // ------------------------------------------------------------------ 2. physical pass
    SEC(12);  // defragmentation: candidate scan
    if (gmoves <= p.number_moves) {
        int nc = 0;
        const int base_cur = tb.pair_base[req_src * N + req_dst];
        if (gnv) nv_fence();   // the grooming pass may have returned channels
        // lane = service, from the record summaries (8 bytes: path, channel count, the first two channels) and the dense seq array;
        // the record itself is read for the arrival time of a candidate and for the channels beyond the second -- one service in
        // ten has them: those are set aside (LDS list) and scored afterwards, a wavefront of them at a time, instead of making
        // every round of 64 services loop to the longest channel list among them.  The next 64 summaries are requested before the
        // current ones are scored.
        auto score = [&](int gid_, int ch_, const NvRec &nr_) -> double {
            if (rss) return lane_rss_delta(occ, tb.sqrt_tab, tb.recs + gid_, ch_, E, W, 1, p.use_masks ? tb.masks + gid_ : nullptr);
            if (gnv) {
                // the service holds the channel on its whole path: c . D[ch] counts the free links towards off-path nodes and the
                // free chords; gain of releasing = 2 * (that - chords) - wsum
                int sdot = nv_dot(nr_.c, nv_get(gnv, ch_, p.C));
                if (nr_.nchord) sdot -= nv_chords(occ, nr_, ch_, W);
                return (double)(2 * sdot - nr_.wsum);
            }
            return (double)(-lane_cut_sum(occ, tb, gid_, ch_, W));
        };
        auto emit = [&](bool is_c, double diff, int idx, int jpos, int ch, int gid_, uint32_t seq_) {
            const u64 m = ballot(is_c);
            if (m) {
                const int pos = nc + popc64(m & ((1ull << lane) - 1ull));
                if (is_c && pos < p.cand_cap) {
                    // (age, modulation level and table row are filled in when the candidates are ranked: cand_fill)
                    OrlgPhyCand c;
                    c.diff = diff; c.age = 0.0; c.seq = seq_; c.idx = (uint16_t)idx; c.chj = (uint16_t)(ch | (jpos << 9));
                    c.gid = (uint16_t)gid_; c.pad0 = 0; c.pad1 = 0u;
                    cand[pos] = c;
                }
                nc += popc64(m);
            }
        };
        uint16_t *more = reinterpret_cast<uint16_t *>(r0w);   // services with more than two channels
        constexpr int MORE_CAP = W * 64 * 8 / 2;
        int n_more = 0;
        auto score_more = [&]() {   // channels 2 .. of the services set aside: lane = service
            for (int m0 = 0; m0 < n_more; m0 += 64) {
                const bool act = m0 + lane < n_more;
                const int idx = act ? (int)more[m0 + lane] : 0;
                u64 sw = 0ull;
                uint32_t my_seq = 0u, x5 = 0u, x6 = 0u, x7 = 0u;   // ch[2..7] of the record
                if (act) {
                    const uint32_t *rr = reinterpret_cast<const uint32_t *>(grec + idx);
                    sw = gsum[idx]; my_seq = gseq[idx]; x5 = rr[5]; x6 = rr[6]; x7 = rr[7];
                }
                const int my_n = act ? sum_nch(sw) : 0, my_gid = sum_gid(sw);
                NvRec nr = nv_unpack(make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u));
                if (!rss && gnv && act) nr = nv_load(p.nvrec, my_gid);
                const int maxn = wave_max_i32(my_n);
                for (int jj = 2; jj < maxn; ++jj) {
                    bool is_c = false;
                    double diff = 0.0;
                    int ch = 0;
                    if (jj < my_n) {
                        const int raw = jj < 8 ? (int)(((jj < 4 ? x5 : jj < 6 ? x6 : x7) >> (16 * (jj & 1))) & 0xffffu) : (int)grec[idx].ch[jj];
                        if (!(raw & (1 << 14))) {  // only channels the service fills are reallocated
                            ch = raw & 0x1ff;
                            diff = score(my_gid, ch, nr);
                            is_c = diff > 0.0;
                        }
                    }
                    emit(is_c, diff, idx, jj, ch, my_gid, my_seq);
                }
            }
            n_more = 0;
            wave_sync();
        };
        u64 sw_n = 0ull;
        uint32_t seq_n = 0u;
        if (lane < n_running) { sw_n = gsum[lane]; seq_n = gseq[lane]; }
        for (int i0 = 0; i0 < n_running; i0 += 64) {
            const int idx = i0 + lane;
            const bool act = idx < n_running;
            const u64 sw = sw_n;
            const uint32_t my_seq = seq_n;
            if (idx + 64 < n_running) { sw_n = gsum[idx + 64]; seq_n = gseq[idx + 64]; }
            const int my_n = act ? sum_nch(sw) : 0, my_gid = sum_gid(sw);
            NvRec nr = nv_unpack(make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u));
            if (!rss && gnv && act) nr = nv_load(p.nvrec, my_gid);
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                bool is_c = false;
                double diff = 0.0;
                int ch = 0;
                if (jj < my_n) {
                    const int raw = sum_ch(sw, jj);
                    if (!(raw & (1 << 14))) {  // only channels the service fills are reallocated
                        ch = raw & 0x1ff;
                        diff = score(my_gid, ch, nr);
                        is_c = diff > 0.0;
                    }
                }
                emit(is_c, diff, idx, jj, ch, my_gid, my_seq);
            }
            const u64 mm = ballot(my_n > 2);
            if (mm) {
                if (my_n > 2) more[n_more + popc64(mm & ((1ull << lane) - 1ull))] = (uint16_t)idx;
                n_more += popc64(mm);
                wave_sync();
                if (n_more > MORE_CAP - 64) score_more();
            }
        }
        if (n_more > 0) score_more();
        if (nc > p.cand_cap) { overflow = true; nc = p.cand_cap; }
        wave_sync();
        SEC(14);  // defragmentation: candidate rounds
        // The rounds are sequential (a move changes what the next candidate sees), but the ORDER of the candidates is fixed once
        // they are scanned -- sorted(key=(-diff, -age)), stable -- and a candidate's table data (its path's node weights, the
        // modulation level of every channel on that path) do not depend on the moves either.  So: every candidate's rank in the
        // sorted order is counted once (all pairs, the keys are distinct; lane l holds candidates l, l + 64, ...: up to 256, the
        // load-1400 workload has 90-190 per cycle), each lane writes its candidates to their sorted position behind the work list,
        // and round q reads record q -- requested one round ahead, one dword per lane.  The service record is only read when a move
        // actually happens.  More candidates than that: the keys stay where they are and every round searches them.
        constexpr int RC = 4;                      // candidates per lane while the ranks are counted
        const bool sorted = nc <= 64 * RC && 2 * nc <= p.cand_cap;
        OrlgPhyCand *scand = cand + (p.cand_cap >> 1);
        // the rest of a candidate's record, one candidate per lane: its age (the service's arrival time: one gather) and what its
        // round will ask the QoT table -- the reference looks the candidate's path up among the k paths of the PENDING request
        // (:388-394): right when both serve the same node pair, otherwise its loop runs out and leaves k - 1 -- the level of its
        // channel on that column (:395) and the column itself
        auto cand_fill = [&](uint4 &a, uint4 &b) {
            const int idx_ = (int)(b.y & 0xffffu), ch_ = (int)((b.y >> 16) & 0x1ffu), gid_ = (int)(b.z & 0xffffu);
            const double age = current_time - grec[idx_].arrival;
            const int ridp = tb.pair_row[tb.path_pair[gid_]] * K + ((gid_ >= base_cur && gid_ < base_cur + K) ? gid_ - base_cur : K - 1);
            const uint32_t level = (uint32_t)p.mod_t[(size_t)ridp * p.cpad + ch_];
            a.z = (uint32_t)__double2loint(age); a.w = (uint32_t)__double2hiint(age);
            b.z = (uint32_t)gid_ | (level << 16); b.w = (uint32_t)ridp;
        };
        if (!sorted) {
            for (int c = lane; c < nc; c += 64) {
                uint4 a = reinterpret_cast<const uint4 *>(cand + c)[0], b = reinterpret_cast<const uint4 *>(cand + c)[1];
                cand_fill(a, b);
                reinterpret_cast<uint4 *>(cand + c)[0] = a; reinterpret_cast<uint4 *>(cand + c)[1] = b;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            wave_sync();
        }
        SEC(15);  // defragmentation: candidate ranks
        if (sorted) {
            uint4 v0[RC], v1[RC];                  // the lane's candidates: diff, age | seq, idx | chj << 16, gid, -
            int rank[RC];
#pragma unroll
            for (int s = 0; s < RC; ++s) {
                const int c = lane + 64 * s;
                v0[s] = make_uint4(0u, 0u, 0u, 0u); v1[s] = v0[s];
                rank[s] = 0;
                if (c < nc) { v0[s] = reinterpret_cast<const uint4 *>(cand + c)[0]; v1[s] = reinterpret_cast<const uint4 *>(cand + c)[1]; }
            }
#pragma unroll
            for (int s = 0; s < RC; ++s)
                if (lane + 64 * s < nc) cand_fill(v0[s], v1[s]);
            // All pairs, but on ONE 64-bit key per candidate that decides nearly every pair: the integer gain and the age rounded
            // to float32 (cut metric), the gain's bits (RSS metric).  A greater key precedes, a smaller one does not (rounding
            // is monotone); only equal keys -- the channels of one service, ages closer than 2^-24 -- take the exact three-part
            // comparison.  3 instead of 7 vector instructions per pair.
            u64 kf[RC];
#pragma unroll
            for (int s = 0; s < RC; ++s) {
                const double rd = __hiloint2double((int)v0[s].y, (int)v0[s].x), ra = __hiloint2double((int)v0[s].w, (int)v0[s].z);
                kf[s] = rss ? (u64)__double_as_longlong(rd) : (((u64)(uint32_t)(int)rd << 32) | (u64)__float_as_uint((float)ra));
                if (lane + 64 * s >= nc) kf[s] = 0ull;
            }
#pragma unroll
            for (int t = 0; t < RC; ++t) {
                const int cnt = nc - 64 * t < 64 ? nc - 64 * t : 64;
                for (int l = 0; l < cnt; ++l) {   // candidate (l, t) against every lane's own
                    const u64 jk = readlane64(kf[t], l);
                    int ties = 0;
#pragma unroll
                    for (int s = 0; s < RC; ++s) {
                        if (64 * s >= nc) continue;   // (wave-uniform: no candidate in this slot of any lane)
                        rank[s] += jk > kf[s] ? 1 : 0;
                        ties += popc64(ballot(jk == kf[s]));
                    }
                    if (ties > 1) {   // (itself is one)
                        const double jd = __hiloint2double(__builtin_amdgcn_readlane((int)v0[t].y, l), __builtin_amdgcn_readlane((int)v0[t].x, l));
                        const double ja = __hiloint2double(__builtin_amdgcn_readlane((int)v0[t].w, l), __builtin_amdgcn_readlane((int)v0[t].z, l));
                        const uint32_t jx = (uint32_t)__builtin_amdgcn_readlane((int)v1[t].x, l), jc = (uint32_t)__builtin_amdgcn_readlane((int)v1[t].y, l);
                        const u64 jo = ((u64)jx << 4) | (u64)(jc >> 25);   // order among equal (diff, age): running_services, then channel position
#pragma unroll
                        for (int s = 0; s < RC; ++s) {
                            const double rd = __hiloint2double((int)v0[s].y, (int)v0[s].x), ra = __hiloint2double((int)v0[s].w, (int)v0[s].z);
                            const u64 ro = ((u64)v1[s].x << 4) | (u64)(v1[s].y >> 25);
                            if (jk == kf[s]) rank[s] += (jd > rd || (jd == rd && (ja > ra || (ja == ra && jo < ro)))) ? 1 : 0;
                        }
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < RC; ++s) {
                const int c = lane + 64 * s;
                if (c < nc) { reinterpret_cast<uint4 *>(scand + rank[s])[0] = v0[s]; reinterpret_cast<uint4 *>(scand + rank[s])[1] = v1[s]; }
            }
            // other lanes of this wave read the sorted records back: the stores only have to be complete (same CU)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            wave_sync();
        }
        // The rounds, eight candidates at a time.  A group's records sit on the lanes (lane 8 c + d: dword d of candidate c, one
        // coalesced load, requested two groups ahead), and so does what the tables say about their paths (lane 8 c + w: the mask
        // of the channels of candidate c's modulation level in word w; lane 8 c + d: dword d of its path's node weights --
        // requested a group ahead): a round waits for no memory.  The free words of all eight paths come from ONE pass over
        // (candidate, word) lanes, as the policy reads its k candidate paths; then the candidates take their turns: free words
        // AND level mask off the lanes, D from LDS, the vote "does any channel beat -diff", the reduction to the best channel
        // only when it passes.  A move (one round in fifteen) changes what the later candidates of the group would see: the
        // next group starts right behind it.
        SEC(14);  // defragmentation: candidate rounds
        constexpr int GR = 8;
        uint32_t rec_a = 0u, rec_b = 0u, tn_a = 0u;
        u64 tm_a = 0ull;
        const int gc = lane >> 3, gd = lane & 7;   // candidate of the group, dword / word
        auto load_recs = [&](int q0) -> uint32_t {  // records q0 .. q0 + 7 of the sorted list
            uint32_t v = 0u;
            if (q0 + gc < nc) v = reinterpret_cast<const uint32_t *>(scand + q0)[lane];
            return v;
        };
        auto issue_tables = [&](uint32_t rv, int gsz, u64 &tm, uint32_t &tn) {
            const uint32_t x6 = (uint32_t)__shfl((int)rv, (lane & ~7) + 6), ridp = (uint32_t)__shfl((int)rv, (lane & ~7) + 7);
            tm = 0ull; tn = 0u;
            if (gc < gsz) {
                if (gd < W) tm = p.lvl_mask[((size_t)ridp * 32 + ((x6 >> 16) & 31u)) * W + gd];
                if (gnv && !rss) tn = reinterpret_cast<const uint32_t *>(p.nvrec + 2 * (x6 & 0xffffu))[gd];
            }
        };
        int q0 = 0;
        bool fresh = true;   // the group's inputs have to be fetched now (the first group, the group behind a move)
        while (q0 < nc) {
            uint32_t rv, tnq = 0u;
            u64 tmq = 0ull;
            int gsz;
            if (sorted) {
                gsz = nc - q0 < GR ? nc - q0 : GR;
                if (fresh) {
                    rec_a = load_recs(q0);
                    rec_b = load_recs(q0 + GR);
                    issue_tables(rec_a, gsz, tm_a, tn_a);
                    fresh = false;
                }
                rv = rec_a; tmq = tm_a; tnq = tn_a;
                rec_a = rec_b;
                rec_b = load_recs(q0 + 2 * GR);
                if (q0 + GR < nc) issue_tables(rec_a, nc - q0 - GR < GR ? nc - q0 - GR : GR, tm_a, tn_a);
            } else {
                // next candidate of sorted(key=(-diff, -age)) (stable: running_services order, then channel order): a group of one
                double bd = -1.0, ba = 0.0;
                u64 bo = ~0ull;
                int bc = -1;
                for (int c = lane; c < nc; c += 64) {
                    const double d = cand[c].diff, a = cand[c].age;
                    const u64 o = ((u64)cand[c].seq << 4) | (u64)(cand[c].chj >> 9);
                    if (d > 0.0 && (d > bd || (d == bd && (a > ba || (a == ba && o < bo))))) { bd = d; ba = a; bo = o; bc = c; }
                }
                // lexicographic maximum over the lanes' bests: greatest diff, then greatest age, then lowest order key (36 bits:
                // exact as a double); the lane that holds it hands out the candidate
                const double ninf = -__longlong_as_double((long long)ORLG_INF_BITS);
                const double D = wave_max_f64(bc >= 0 ? bd : ninf);
                if (!(D > 0.0)) break;
                const double A = wave_max_f64((bc >= 0 && bd == D) ? ba : ninf);
                const double O = -wave_max_f64((bc >= 0 && bd == D && ba == A) ? -(double)bo : ninf);
                const int wl = ctz64(ballot(bc >= 0 && bd == D && ba == A && (double)bo == O));
                const int cb = __builtin_amdgcn_readlane(bc, wl);
                rv = lane < 8 ? reinterpret_cast<const uint32_t *>(cand + cb)[lane] : 0u;
                wave_sync();
                if (lane == 0) cand[cb].diff = -1.0;
                gsz = 1;
                issue_tables(rv, 1, tmq, tnq);
            }
            // free on the path and of the candidate's modulation level: only those channels can take it over -- all candidates of
            // the group at once, lane = (candidate, word)
            u64 acc_g;
            {
                const int gid_l = (int)((uint32_t)__shfl((int)rv, (lane & ~7) + 6) & 0xffffu);
                const bool on = gc < gsz && gd < W;
                acc_g = path_word<W>(occ, tb.recs, gid_l, gd < W ? gd : 0, on) & (on ? tmq : 0ull);
            }
            bool moved_in_group = false;
            int c = 0;
            for (; c < gsz; ++c) {
                const int l8 = 8 * c;
                const double diff = __hiloint2double(__builtin_amdgcn_readlane((int)rv, l8 + 1), __builtin_amdgcn_readlane((int)rv, l8));
                const uint32_t x5 = (uint32_t)__builtin_amdgcn_readlane((int)rv, l8 + 5);
                const int idx = (int)(x5 & 0xffffu), ch = (int)((x5 >> 16) & 0x1ffu);
                const int gid = (int)((uint32_t)__builtin_amdgcn_readlane((int)rv, l8 + 6) & 0xffffu);
                const OrlgPhySvc *r = grec + idx;
                const OrlgPathRec *rec = tb.recs + gid;
                u64 xw[W];
                u64 any = 0ull;
#pragma unroll
                for (int w = 0; w < W; ++w) { xw[w] = readlane64(acc_g, l8 + w); any |= xw[w]; }
                int l0 = -1, c0 = -1;
                double m0 = 0.0;
                if (any != 0ull) {
                    if (gnv && !rss) {
                        // cut metric of the lane's channels from D (LDS) and the path's node weights: an integer; the best channel
                        // = greatest metric, then lowest channel number, as ONE key.  Four rounds in five find a free channel of
                        // that level, one in fifteen moves: the vote comes first, the reduction only when it passes.
                        uint4 qa, qb;
                        qa.x = (uint32_t)__builtin_amdgcn_readlane((int)tnq, l8 + 0); qa.y = (uint32_t)__builtin_amdgcn_readlane((int)tnq, l8 + 1);
                        qa.z = (uint32_t)__builtin_amdgcn_readlane((int)tnq, l8 + 2); qa.w = (uint32_t)__builtin_amdgcn_readlane((int)tnq, l8 + 3);
                        qb.x = (uint32_t)__builtin_amdgcn_readlane((int)tnq, l8 + 4); qb.y = (uint32_t)__builtin_amdgcn_readlane((int)tnq, l8 + 5);
                        qb.z = (uint32_t)__builtin_amdgcn_readlane((int)tnq, l8 + 6); qb.w = (uint32_t)__builtin_amdgcn_readlane((int)tnq, l8 + 7);
                        const NvRec nr = nv_unpack(qa, qb);
                        int key = -1;
                        // -metric < diff with an integer metric and an integer-valued diff: metric + 1024 > 1024 - diff
                        const int kthr = ((1024 - (int)diff) << 9) | 511;
#pragma unroll
                        for (int w = 0; w < W; ++w) {
                            const u64 x = xw[w];
                            if (x == 0ull) continue;   // (wave-uniform: no free channel of that level in this word)
                            const int cc = 64 * w + lane;
                            const bool fr = ((x >> lane) & 1ull) && cc < p.C;
                            int sdot = nv_dot(nr.c, nv_get(gnv, cc, p.C)) - nr.cq;
                            if (nr.nchord) sdot -= nv_chords(occ, nr, cc, W);
                            const int kk = ((nr.wsum - 2 * sdot + 1024) << 9) | (511 - cc);   // |metric| <= sum of the weights < 1024
                            if (fr && kk > key) key = kk;
                        }
                        if (ballot(key > kthr) != 0ull) {
                            key = wave_max_i32(key);
                            l0 = 0; c0 = 511 - (key & 511); m0 = (double)((key >> 9) - 1024);
                        }
                    } else {
                        int lv[W];
                        double mtr[W];
                        uint32_t cols[W];
                        uint4 dv0[W];
#pragma unroll
                        for (int w = 0; w < W; ++w) dv0[w] = make_uint4(0u, 0u, 0u, 0u);
                        u64 acc1 = 0ull;   // the candidate's free words as phy_row_metrics takes them: word w on lane w
#pragma unroll
                        for (int w = 0; w < W; ++w)
                            if (lane == w) acc1 = xw[w];
                        const uint8_t *mrow = p.mod_t + (size_t)__builtin_amdgcn_readlane((int)rv, l8 + 7) * p.cpad;   // (levels: not looked at, flat)
                        phy_columns<W>(occ, tb, p, lane, rss ? 1 : 0, cols, r0w);
                        phy_row_metrics<W>(occ, tb, p, acc1, 0, gid, mrow, lane, rss ? 1 : 0, true, lv, mtr, cols, r0w, dv0);
                        phy_row_best<W>(lv, mtr, lane, l0, m0, c0);  // sorted(key=(-metric, channel))[0]
                    }
                }
                if (l0 >= 0 && -1.0 * m0 < diff) {
                    // _move (:662-697): the service's channel list is read now -- the moved entry goes to its end
                    const uint32_t d3 = (uint32_t)uni((int)reinterpret_cast<const uint32_t *>(r)[3]);   // gid | nch << 16 | flags << 24
                    const int nch = (int)((d3 >> 16) & 0xffu), rflags = (int)(d3 >> 24);
                    const int mych = lane < nch ? (int)r->ch[lane] : 0xffff;
                    const u64 jm = ballot(lane < nch && (mych & 0x1ff) == ch && !(mych & (1 << 14)));
                    if (jm) {
                        const int jpos = ctz64(jm);
                        mc_before(occ, mc, c0, lane);
                        mc_before(occ, mc, ch, lane);
                        if (lane < rec->hops) {
                            u64 *rowp = occ + (int)rec->link[lane] * W;
                            rowp[c0 >> 6] &= ~(1ull << (c0 & 63));
                            rowp[ch >> 6] |= 1ull << (ch & 63);
                        }
                        wave_sync();
                        mc_after(occ, mc, c0, lane);
                        mc_after(occ, mc, ch, lane);
                        if (gnv) {
                            const uint4 cv4 = p.nvrec[2 * gid];
                            if (lane < 2) nv_update(gnv, cv4, lane == 0 ? c0 : ch, lane != 0);
                        }
                        const int nxtc = __shfl_down(mych, 1);
                        int nv2 = mych;
                        if (lane >= jpos && lane + 1 < nch) nv2 = nxtc;
                        if (lane == nch - 1) nv2 = c0 | (readlane64((u64)(uint32_t)mych, jpos) & 0xfe00u);
                        if (lane < nch) grec[idx].ch[lane] = (uint16_t)nv2;
                        {
                            const uint32_t h0 = (uint32_t)__builtin_amdgcn_readlane(nv2, 0), h1 = (uint32_t)__builtin_amdgcn_readlane(nv2, 1);
                            if (lane == 0) {
                                grec[idx].seq = (uint32_t)next_seq;
                                gsum[idx] = svc_summary(gid, rflags, nch, h0, nch > 1 ? h1 : 0u);
                                gseq[idx] = (uint32_t)next_seq;
                            }
                        }
                        next_seq += 1;
                        cmoves += 1;
                        moved_in_group = true;
                        wave_sync();
                    }
                }
                if (cmoves + gmoves > p.number_moves || moved_in_group) { c += 1; break; }
            }
            if (cmoves + gmoves > p.number_moves) break;
            q0 += c;                 // (the candidates behind a move see the new occupancy: their group is read again)
            if (moved_in_group) fresh = true;
        }
        cycles = cmoves != 0 ? 1 : 0;
    }
    if (lane == 0) {
        ws->counted_moves_groom = gmoves;
        ws->counted_moves += cmoves;
        ws->counted_defrag_cycles += cycles;
        if (overflow) ws->q_overflow |= 2;
    }
    wave_sync();
}

// GN-model GSNR [dB] of channel `ch` on the path `rec` against the live occupancy (include/orlg.h orlg_gn_gate): the
// arithmetic of examples/calculate_osnr.py:9-56 for a service that is not yet in the links' lists.  Wave-cooperative, result
// wave-uniform.  Lanes = channels: the two asinh terms and the modulation term of an interferer depend on the fibre only
// through its attenuation, uniform here, so they are evaluated once per channel (A, B) and summed per link over the channels
// the link has lit (the reference's per-interferer sum, re-associated: ~1e-15 relative); the spans of a link are equal, their
// contribution is added span by span like the reference does.
template <int W>
DEV double gn_gsnr(const OrlgPhyParams &p, const u64 *occ, const OrlgPathRec *rec, int mrow_off, int ch_v, int lane SEC_PARAMS) {
    SEC(8);   // (section profile of the check: table rows | 11 hop sums | 12 span powers | 14 logarithm)
    const double beta_2 = -21.3e-27, gamma = 1.3e-3, h_plank = 6.626e-34, pi = 3.141592653589793;
    // the channel and the table row are wave-uniform, and the compiler has to know it: as values of lanes (they come out of LDS)
    // every table address was a 64-bit register pair per word -- spilled, and each reload's wait also waited for the loads before it
    const int ch = uni(ch_v);
    const uint8_t *mrow = p.mod_t + (size_t)uni(mrow_off);
    const double bw = p.gn_bw, pw = p.gn_pw, nf = p.gn_nf;
    const double fc = p.gn_cf[ch];
    // the interferer terms of the lane's channels against channel ch: rows of the tables (coalesced over the lanes)
    const double *rowA = p.gn_A + (size_t)ch * p.cpad, *rowR = p.gn_R + (size_t)ch * p.cpad;
    // (every load of the check is issued before the first value is used, none of them under a condition: a load inside
    // `if (valid)` has to be waited for inside it -- one memory round trip per word, and they were most of the check's time)
    double A[W], B[W];
    int se_w[W];
#pragma unroll
    for (int w = 0; w < W; ++w) {
        const int c = 64 * w + lane;
        const int cc = c < p.C ? c : 0;   // a channel that exists: the value is dropped below
        se_w[w] = (int)mrow[cc]; A[w] = rowA[cc]; B[w] = rowR[cc];
    }
    const double base = p.gn_link[4 * p.E];
    const double r = pw / bw;
    double acc = 0.0;
    const int hops = rec->hops;
    // the links' constants (effective length, its ratio to the span length, exp(2 att len) - 1, spans) for every hop at once:
    // lane h = hop h, read back per hop by readlane -- one memory round trip per check instead of one per hop
    double lk0, lk1, lk2;
    int lkn;
    {
        const int lnk = (int)rec->link[lane < hops ? lane : 0];   // (lanes past the path's end read hop 0's constants and do not use them)
        lk0 = p.gn_link[4 * lnk]; lk1 = p.gn_link[4 * lnk + 1]; lk2 = p.gn_link[4 * lnk + 2];
        lkn = p.gn_nspans[lnk];
    }
#pragma unroll
    for (int w = 0; w < W; ++w) {
        const int c = 64 * w + lane;
        const bool valid = c < p.C && c != ch;
        int se = se_w[w];
        se = se < 1 ? 1 : (se > 6 ? 6 : se);
        const double pm = se <= 2 ? 1.0 : se == 3 ? 2.0 / 3 : se == 4 ? 17.0 / 25 : se == 5 ? 69.0 / 100 : 13.0 / 21;
        A[w] = valid ? A[w] : 0.0;
        B[w] = valid ? pm * B[w] * 5 / 3 : 0.0;
    }
    // per hop only the interferer sum over the link's lit channels is wave-wide work; what follows from it -- the span's NLI
    // and ASE power and its share of 1 / GSNR: ~100 instructions with two divisions -- is done for ALL hops at once, lane h =
    // hop h, and the spans are then added hop by hop, span by span, as the reference adds them
    double sp = 0.0;   // lane h: sum_phi of hop h
    SEC(11);
    // three hops at a time: their occupancy words are requested together and their wave sums -- chains of dependent DPP steps --
    // run interleaved (the sums themselves are formed as before, hop by hop)
    constexpr int HB = 3;
    for (int h0 = 0; h0 < hops; h0 += HB) {
        double sphi[HB];
#pragma unroll
        for (int j = 0; j < HB; ++j) {
            sphi[j] = 0.0;
            const int h = h0 + j < hops ? h0 + j : hops - 1;   // (a hop past the path's end repeats the last one; its sum is not used)
            const int link = (int)rec->link[h];
            const double ratio = readlane_d(lk1, h);
            // per interferer asinh(..) - asinh(..) - phi_mod (B / |df|) 5/3 l_eff / L, as calculate_osnr.py:33-45 sums them
#pragma unroll
            for (int w = 0; w < W; ++w) {
                const bool lit = !((occ[__mul24(link, W) + w] >> lane) & 1ull);   // (A, B are 0 on channels that do not exist)
                sphi[j] += lit ? (A[w] - (B[w] * ratio)) : 0.0;
            }
        }
#pragma unroll
        for (int j = 0; j < HB; ++j) {
            const double tot = base + wave_add_f64(sphi[j]);
            if (lane == h0 + j && h0 + j < hops) sp = tot;
        }
    }
    double gv = 0.0;
    SEC(12);
    {
        const double l_eff = lk0, e1 = lk2;
        const double power_nli_span = (r * r * r) * (8 / (27 * pi * fabs(beta_2))) * (gamma * gamma) * l_eff * sp * bw;
        const double power_ase = bw * h_plank * fc * e1 * nf;
        if (lane < hops) gv = 1 / (pw / (power_ase + power_nli_span));
    }
    for (int h = 0; h < hops; ++h) {
        const double g = readlane_d(gv, h);
        const int ns = __builtin_amdgcn_readlane(lkn, h);
#pragma unroll 4
        for (int sx = 0; sx < ns; ++sx) acc += g;
    }
    SEC(14);
    const double gsnr_db = 10 * log10(1 / acc);
    SEC(5);
    return gsnr_db;
}
// DF: the instantiation that carries the periodic defragmentation (and the node-degree vectors of its cut metric); handles
// without it run the other one, whose registers are not shared with code they never execute
// GN: ... and the one that also carries the GN-model admission check (orlg_gn_gate)
// ---- the release of the NEXT step, looked up ahead.  The arrival times come from the pre-generated ring, so the release
// loop's scan already knows the time of the following arrival and finds the service that will be released first then; its
// record is requested right away and is on lanes when the next step's release loop needs it.  That loop still finds its
// victims by itself: the record is used only when its first victim is the one looked up (anything else is a plain load).
// (Requesting the service's channel_state list, node weights and the queue's last record ahead as well was measured: no
// gain, four more registers held across the step.)
struct ReleaseAhead {
    int q;            // queue index of the looked-up service, -1: none
    uint32_t rec;     // lane < 12: dword `lane` of its record
};
DEV uint32_t rec_dword(const OrlgPhySvc *grec, int q, int lane) {
    return lane < 12 ? reinterpret_cast<const uint32_t *>(grec + q)[lane] : 0u;
}
static_assert(sizeof(OrlgPhySvc) == 48 && ORLG_PHY_MAX_CH == 14, "record = 12 dwords: arrival, seq, gid | nch | flags, 14 channels, pad");
// a new record written by lanes: lane i < nch holds channel i's halfword (0xffff beyond), dwords 4..10 pair them up
DEV uint32_t rec_store(OrlgPhySvc *dst, const double *arrival_lds, uint32_t seq, int gid, int nch, int flags, uint32_t hw, int lane) {
    const int j = lane >= 4 ? lane - 4 : 0;
    const uint32_t h0 = (uint32_t)__shfl((int)hw, 2 * j), h1 = (uint32_t)__shfl((int)hw, 2 * j + 1);
    uint32_t v = h0 | (h1 << 16);
    if (lane < 2) v = reinterpret_cast<const uint32_t *>(arrival_lds)[lane];
    if (lane == 2) v = seq;
    if (lane == 3) v = (uint32_t)gid | ((uint32_t)nch << 16) | ((uint32_t)flags << 24);
    if (lane == 11) v = 0u;
    if (lane < 12) reinterpret_cast<uint32_t *>(dst)[lane] = v;
    return v;
}
// the side arrays of a new record (OrlgPhyParams::qsum / qseq): hw = the halfword of channel `lane` as rec_store takes it
DEV void svc_side_store(u64 *gsum, uint32_t *gseq, int q, int gid, int flags, int nch, uint32_t hw, uint32_t seq, int lane) {
    const uint32_t h0 = (uint32_t)__builtin_amdgcn_readlane((int)hw, 0), h1 = (uint32_t)__builtin_amdgcn_readlane((int)hw, 1);
    if (lane == 0) {
        gsum[q] = svc_summary(gid, flags, nch, h0, nch > 1 ? h1 : 0u);
        gseq[q] = seq;
    }
}
// first service due at `time` among the near buffer's entries (earliest release, ties: lowest queue index)
DEV void nb_first_due(const NearBuffer &nb, double time, int lane, int &victim, int &vpos, double &best_t) {
    best_t = 0.0;
    victim = -1; vpos = -1;
    for (int c0 = 0; c0 < nb.n; c0 += 64) {
        const int c = c0 + lane;
        const double tq = c < nb.n ? nb.t[c] : __longlong_as_double((long long)ORLG_INF_BITS);
        const int qi = c < nb.n ? (int)nb.qi[c] : 0;
        u64 m = ballot(tq <= time);
        while (m) {
            const int l = ctz64(m);
            m &= m - 1;
            const double tt = readlane_d(tq, l);
            const int qq = __builtin_amdgcn_readlane(qi, l);
            if (victim < 0 || tt < best_t || (tt == best_t && qq < victim)) { best_t = tt; victim = qq; vpos = c0 + l; }
        }
    }
}

// POL: the policy of the launch (ORLG_PHY_POLICY_*; launches that do not step run the EXTERNAL instantiation).  A compile-time
// policy turns the per-policy choices inside the channel loops (level as a sort key or not, which metric, first row or best
// row) into straight-line code: a wave of this kernel is bound by its own instruction latency, and every wave-uniform branch
// inside an unrolled word loop is a fetch bubble paid W times per candidate path.
template <int W, bool DF, bool GN, int POL>
__global__ __launch_bounds__(ORLG_WAVE *ORLG_MAX_WAVES_PER_BLOCK, 4) void orlg_phy_kernel(const OrlgPhyParams p) {
    // the policy sorts channels by the RSS metric (floating point) instead of an integer key
    constexpr bool RSSP = POL == ORLG_PHY_POLICY_BMFA_RSS_METRIC || POL == ORLG_PHY_POLICY_FAFF_RSS;
    extern __shared__ __align__(16) unsigned char smem[];
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(p.tables);
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        const int n16 = p.tab_bytes >> 4;
        for (int i = threadIdx.x; i < n16; i += blockDim.x) dst[i] = src[i];
#pragma unroll
        for (int i = 0; i < ORLG_PHY_NUM_OUTS; ++i)
            if ((int)threadIdx.x == i) reinterpret_cast<u64 *>(smem + p.l_outs)[i] = reinterpret_cast<u64>(p.outs[i]);
        if (threadIdx.x == 0) *reinterpret_cast<int *>(smem + p.l_mtstage + ORLG_MT_N * 4) = 0;   // the staging buffer's lock
        __syncthreads();
    }
    // one MT19937 staging buffer per workgroup, handed from wave to wave with a lock word (as orlg_rmsa_group_kernel): the
    // arrivals of an environment are generated 64 at a time (refill_requests) into a ring in HBM
    uint32_t *mt_lds = reinterpret_cast<uint32_t *>(smem + p.l_mtstage);
    int *mt_lock = reinterpret_cast<int *>(smem + p.l_mtstage + ORLG_MT_N * 4);
    const int lane = threadIdx.x & 63;
    const int wib = uni((int)(threadIdx.x >> 6));
    const PhyTab tb = make_phy_tab(smem, p);
    unsigned char *wb = smem + p.l_shared_bytes + (size_t)wib * p.l_wave_bytes;
    u64 *occ = reinterpret_cast<u64 *>(wb + p.l_occ);
    NearBuffer nb;
    nb.t = reinterpret_cast<double *>(wb + p.l_nbt);
    nb.qi = reinterpret_cast<uint16_t *>(wb + p.l_nbi);
    uint32_t *scratch = reinterpret_cast<uint32_t *>(wb + p.l_scratch);  // selection lists + per-channel doubles
    PhyWaveScalars *ws = reinterpret_cast<PhyWaveScalars *>(wb + p.l_wsc);

    const int E = p.E, C = p.C, K = p.K, N = p.N, NBR = p.NBR, Q = p.Q, NW = p.NW;
    SEC_DECL
    // work queue (as orlg_rmsa_kernel): long launches draw environments from the ticket counter (the next ticket is drawn
    // while the current environment runs), short ones stride statically
    const int n_waves = (int)(gridDim.x * (blockDim.x >> 6));
    const int n_static = n_waves < p.B ? n_waves : p.B;
    int env = (int)(blockIdx.x * (blockDim.x >> 6)) + wib;
    if (env >= p.B) return;
    uint32_t nxt_tk = 0;
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
    if (!p.ticket_stride && lane == 0) {
        const OrlgPhyParams __attribute__((address_space(4))) *kq =
            (const OrlgPhyParams __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
        nxt_tk = atomicAdd(kq->ticket, 1u);
    }
    OrlgPhySvc *grec = p.qrec + (size_t)env * Q;
    double *gq = p.qtime + (size_t)env * Q;   // release times, compact: entries 0..n_running-1 are live
    u64 *gsum = DF && p.qsum ? p.qsum + (size_t)env * Q : nullptr;        // side arrays of the records (defragmentation scans)
    uint32_t *gseq = DF && p.qseq ? p.qseq + (size_t)env * Q : nullptr;
    nb.n = 0;
    nb.horizon = -__longlong_as_double((long long)ORLG_INF_BITS);  // the first look at the queue rebuilds the buffer
    uint32_t *gcs = p.cs + (size_t)env * N * N * K * p.cs_len;
    uint8_t *gcs_n = p.cs_n + (size_t)env * N * N * K;
    u64 *gnv = p.use_nv ? reinterpret_cast<u64 *>(wb + p.l_nv) : nullptr;   // node-degree vectors of the cut metric

    SEC(1);  // state load
    // ------------------------------------------------------------------ HBM -> LDS
    const OrlgPhyScalars *gs = p.scal + env;
    int n_running = gs->n_running;
    {
        const u64 *g = p.occ + (size_t)env * NW;
        for (int i = lane; i < NW; i += 64) occ[i] = g[i];
        if (lane < 8) ws->c[lane] = gs->c[lane];
        if (lane == 0) {
            ws->total_path_index = gs->total_path_index; ws->total_mod = gs->total_mod;
            ws->channels_accepted = gs->channels_accepted; ws->physical_accepted = gs->physical_accepted;
            ws->episodes_done = gs->episodes_done;
            ws->total_path_length = gs->total_path_length; ws->total_gsnr = gs->total_gsnr;
            ws->req_arrival = gs->req_arrival; ws->req_holding = gs->req_holding;
            ws->q_overflow = gs->q_overflow;
            ws->counted_moves = gs->counted_moves; ws->counted_moves_groom = gs->counted_moves_groom;
            ws->counted_defrag_cycles = gs->counted_defrag_cycles;
        }
    }
    int next_seq = gs->next_seq;
    OrlgPhyCand *gcand = p.cand ? p.cand + (size_t)env * p.cand_cap : nullptr;
    double current_time = gs->current_time;
    int req_src = gs->req_src, req_dst = gs->req_dst, req_br = gs->req_br, req_sid = gs->req_sid;
    int mt_idx = gs->mt_idx, new_service = gs->new_service;
    int ring_pos = gs->ring_pos, ring_cnt = gs->ring_cnt;
    int eproc = (int)gs->c[2];
    wave_sync();
    if (gnv && p.mode == ORLG_MODE_STEP) nv_build<W>(gnv, occ, tb.lnib, E, C, lane);

    int *sel_ch = reinterpret_cast<int *>(scratch);            // [16] selected channels
    int *sel_cap = reinterpret_cast<int *>(scratch) + 16;      // [16] their capacity (modulation level)
    int *sel_used = reinterpret_cast<int *>(scratch) + 32;     // [16] the share this service uses
    double *scratch_d = reinterpret_cast<double *>(scratch + 64);  // [W*64] per-channel doubles

    // per-step totals (number_cuts_total / rss_total_metric) kept incrementally over the launch when they are asked for
    MetricCache mc;
    mc.on = p.mode == ORLG_MODE_STEP && p.use_masks && p.cterm != nullptr &&
            (p.out_mask & ((1 << ORLG_PHY_OUT_CUTS) | (1 << ORLG_PHY_OUT_RSS))) != 0;
    mc.want_rss = (p.out_mask & (1 << ORLG_PHY_OUT_RSS)) != 0;
    mc.total_runs = 0;
    mc.cterm = p.cterm ? p.cterm + (size_t)env * p.cpad : nullptr;
    mc.lterm = (double __attribute__((address_space(3))) *)scratch_d;
    // the steps of this instantiation never touch scratch_d: the RSS terms stay there (a defragmentation cycle uses the scratch:
    // the terms go to the HBM array for its duration -- once in defrag_period steps instead of a round trip every step)
    constexpr bool LT = !RSSP;
    mc.sqrt_tab = tb.sqrt_tab; mc.E = E; mc.W = W;
    // (launches of few steps -- the gym views -- sum every step: a block of one step would cost more than its chain)
    mc.defer = LT && mc.on && mc.want_rss && p.rlog_t0 != nullptr && p.n_steps >= 16;
    mc.log_overflow = false; mc.nlog = 0; mc.stamp = 0; mc.t0 = 0; mc.env = env;
    if (mc.on) {
        double c0_unused, r0_unused;
        phy_column_metrics<W>(occ, tb.sqrt_tab, E, C, lane, scratch_d, true, mc.want_rss, true, c0_unused, r0_unused, mc.total_runs);
        if (mc.want_rss && !LT) {
            for (int ch = lane; ch < C; ch += 64) mc.cterm[ch] = scratch_d[ch];
            wave_sync();
        }
        if (mc.defer) {
            double *lt0 = p.rlog_t0 + (size_t)env * p.cpad;
#pragma unroll
            for (int w = 0; w < W; ++w) lt0[64 * w + lane] = scratch_d[64 * w + lane];
        }
    }

    // GN gate: threshold q of the modulation levels on lane q (+inf beyond the last), read once per environment
    double gn_thr_l = __longlong_as_double((long long)ORLG_INF_BITS);
    if (GN && p.gn_on && lane < p.gn_nthr) gn_thr_l = p.gn_thr[lane];
    ReleaseAhead ra;
    ra.q = -1; ra.rec = 0u;
    // the next ring entry, requested one step ahead: lanes 0, 1 inter-arrival time, lanes 2, 3 holding time, lane 4 the request
    // (lanes 8, 9: the inter-arrival time of the entry after it, for the look-ahead of the release loop)
    uint32_t pf_ring = 0u;
    bool pf_ring_ok = false, pf_next_ok = false;
    auto ring_fetch = [&]() {
        pf_ring_ok = ring_cnt > 0;
        pf_next_ok = ring_cnt > 1;
        if (pf_ring_ok) {
            const OrlgPhyParams __attribute__((address_space(4))) *kq =
                (const OrlgPhyParams __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
            const size_t ro = (size_t)env * ORLG_RING + ring_pos;
            const uint32_t *src = lane < 2 ? reinterpret_cast<const uint32_t *>(kq->ring_iat + ro) + lane
                                : lane < 4 ? reinterpret_cast<const uint32_t *>(kq->ring_ht + ro) + (lane - 2)
                                : lane < 8 ? kq->ring_req + ro : reinterpret_cast<const uint32_t *>(kq->ring_iat + ro + 1) + (lane - 8);
            if (lane < 5 || (pf_next_ok && (lane == 8 || lane == 9))) pf_ring = *src;
        }
    };
    if (p.mode == ORLG_MODE_STEP) ring_fetch();
    const int n_iter = p.mode == ORLG_MODE_STEP ? p.n_steps : 1;
    for (int t = 0; t < n_iter; ++t) {
        SEC(2);  // policy: virtual layer
        if (p.mode == ORLG_MODE_STEP) {
            // D of the lane's channels (cut metric): requested first, used after the virtual-layer check; it serves every candidate
            // path of the request.  (The fence: entries other lanes rewrote since the last look -- their stores are long done.)
            uint4 dv[W];
#pragma unroll
            for (int w = 0; w < W; ++w) dv[w] = make_uint4(0u, 0u, 0u, 0u);
            if (gnv && (POL == ORLG_PHY_POLICY_BMFA_CUT || POL == ORLG_PHY_POLICY_FAFF)) {
                nv_fence();
#pragma unroll
                for (int w = 0; w < W; ++w) dv[w] = nv_get(gnv, 64 * w + lane, C);
            }
            const int base = tb.pair_base[req_src * N + req_dst];
            const int row = tb.pair_row[req_src * N + req_dst];
            const int demand = tb.bit_rates[req_br];
            constexpr int policy = POL;
            int a_path = -2, nsel = 0;
            // requested now, used after the virtual-layer check as well: the modulation levels of the lane's channels on the K
            // candidate paths (one or two words per channel) and the paths' node records (lane 2 i, 2 i + 1: path i)
            uint32_t lvk[W];   // (a fifth path's levels are read where they are needed: K = 5 pays a wait, K <= 4 no registers)
            uint4 nvq = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int w = 0; w < W; ++w) lvk[w] = 0u;
            const uint32_t *mk_hi = p.mod_k + ((size_t)row * p.cpad + lane) * 2 + 1;
            if (!RSSP && policy != ORLG_PHY_POLICY_EXTERNAL) {
                const uint32_t *mk = p.mod_k + ((size_t)row * p.cpad + lane) * 2;
#pragma unroll
                for (int w = 0; w < W; ++w) lvk[w] = mk[128 * w];
                if (gnv && lane < 2 * K) nvq = p.nvrec[2 * base + lane];
            }

            if (policy == ORLG_PHY_POLICY_EXTERNAL) {
                const OrlgPhyParams __attribute__((address_space(4))) *kp =
                    (const OrlgPhyParams __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
                a_path = uni(kp->act_path[env]);
                const int16_t *ac = kp->act_channels + (size_t)env * ORLG_PHY_MAX_CH;
                int raw = lane < ORLG_PHY_MAX_CH ? (int)ac[lane] : -1;
                nsel = popc64(ballot(raw >= 0));  // channels are the leading non-negative entries
                const int idp = a_path > 10 ? a_path - 20 : a_path;
                if (lane < nsel) {
                    const int c = raw & 0x1ff, u = (raw >> 9) & 0x1f;
                    int cap = (idp >= 0 && idp < K && c < C) ? (int)p.mod_t[(size_t)(row * K + idp) * p.cpad + c] : 0;
                    sel_ch[lane] = c; sel_cap[lane] = cap; sel_used[lane] = u ? u : cap;
                }
                wave_sync();
            } else {
                // ---------------- the heuristics of phy_rmsa_env.py:1254-1737
                const bool faff = policy == ORLG_PHY_POLICY_FAFF || policy == ORLG_PHY_POLICY_FAFF_RSS;
                const bool bmfa = policy == ORLG_PHY_POLICY_BMFA_CUT || policy == ORLG_PHY_POLICY_BMFA_RSS_METRIC;
                const bool with_metric = bmfa || faff;
                const bool groom = bmfa ? (p.grooming != 0) : true;
                bool served = false;
                if (groom) {
                    // use_existing_channels (:1650-1673): residual capacity on channels this (src, dst, k-path) already lights
                    int unassigned = demand;
                    for (int idp = 0; idp < K && !served; ++idp) {
                        const CsList l = cs_load(gcs, gcs_n, (req_src * N + req_dst) * K + idp, lane, p.cs_len);
                        int fr = lane < l.n ? cs_free(l.e) : 0;
                        const int sum = wave_add_i32(fr);
                        if (sum * 100 >= unassigned) {
                            for (int i = 0; i < l.n && nsel < ORLG_PHY_MAX_CH; ++i) {
                                const uint32_t en = cs_get(l, i);
                                const int f = cs_free(en);
                                if (f > 0) {
                                    unassigned -= f * 100;
                                    int take = f;
                                    if (unassigned <= 0) take = f + unassigned / 100;  // unassigned is a multiple of 100
                                    if (lane == 0) { sel_ch[nsel] = cs_ch(en); sel_cap[nsel] = cs_cap(en); sel_used[nsel] = take; }
                                    nsel += 1;
                                    if (unassigned <= 0) { a_path = idp + 20; served = true; break; }
                                }
                            }
                        }
                    }
                    if (!served) nsel = 0;
                }
                SEC(3);  // policy: row metrics
                if (!served) {
                    // per path ("row") the free channels ordered by (level desc, metric desc, channel asc)
                    //   bmfa / bmfa_rss: sorted(row, key=(-level, -metric)), row with the best head (level, metric)
                    //   bmff:            sorted(row, key=(-level, channel)),  row with the best head level (ties: lower index)
                    //   sapbm:           sorted(row, key=(-level, channel)),  first non-empty row
                    //   sapff:           sorted(row, key=channel),            first non-empty row
                    //   faff / faff_rss: sorted(row, key=-metric),            row with the best head metric (ties: lower index)
                    const int metric_mode = RSSP ? 1 : (policy == ORLG_PHY_POLICY_BMFA_CUT || policy == ORLG_PHY_POLICY_FAFF) ? 0 : 2;
                    const bool flat = policy == ORLG_PHY_POLICY_SAPFF || faff;  // the level is not a sort key
                    const bool first_row = policy == ORLG_PHY_POLICY_SAPFF || policy == ORLG_PHY_POLICY_SAPBM;
                    const int pp = lane / W, pw = lane - pp * W;
                    const u64 acc = path_word<W>(occ, tb.recs, base + pp, pw, pp < K);
                    if constexpr (!RSSP) {
                        // integer metric: one sortable key per channel (phy_row_keys)
                        int head_key[ORLG_PHY_MAX_K];
                        unsigned alive = 0;
                        // the row the first pick below will choose keeps its per-channel keys: no second pass
                        int keep_idp = -1, keep_key[W], keep_h = -1;
#pragma unroll
                        for (int w = 0; w < W; ++w) keep_key[w] = -1;
#pragma unroll
                        for (int idp = 0; idp < ORLG_PHY_MAX_K; ++idp) {
                            head_key[idp] = -1;
                            if (idp < K) {
                                int key[W];
                                phy_row_keys<W>(occ, tb, p, acc, idp, base + idp, p.mod_t + (size_t)(row * K + idp) * p.cpad, lane, metric_mode, flat, key, dv, lvk, mk_hi, nvq);
                                const int h = phy_keys_best<W>(key);
                                if (h >= 0) {
                                    head_key[idp] = h; alive |= 1u << idp;
                                    // the head's (level, metric): the key without its channel bits
                                    if (keep_idp < 0 || (!first_row && (h >> 9) > (keep_h >> 9))) {
                                        keep_idp = idp; keep_h = h;
#pragma unroll
                                        for (int w = 0; w < W; ++w) keep_key[w] = key[w];
                                    }
                                }
                            }
                        }
                        SEC(4);  // policy: channel selection
                        for (;;) {
                            int best = -1, bh = -1;
#pragma unroll
                            for (int idp = 0; idp < ORLG_PHY_MAX_K; ++idp)
                                if (idp < K && ((alive >> idp) & 1u)) {
                                    if (best < 0 || (!first_row && (head_key[idp] >> 9) > (bh >> 9))) { best = idp; bh = head_key[idp]; }
                                }
                            if (best < 0) break;
                            int key[W];
                            const uint8_t *mrow = p.mod_t + (size_t)(row * K + best) * p.cpad;
                            if (best == keep_idp) {
#pragma unroll
                                for (int w = 0; w < W; ++w) key[w] = keep_key[w];
                                keep_idp = -1;
                            } else {
                                phy_row_keys<W>(occ, tb, p, acc, best, base + best, mrow, lane, metric_mode, flat, key, dv, lvk, mk_hi, nvq);
                            }
                            int unassigned = demand;
                            nsel = 0;
                            bool covered = false;
                            int h = bh;   // the row's head is its first channel
                            while (nsel < ORLG_PHY_MAX_CH) {
                                if (h < 0) break;
                                const int c0 = 511 - (h & 511);
#pragma unroll
                                for (int w = 0; w < W; ++w) key[w] = key[w] == h ? -1 : key[w];
                                const int level = flat ? (int)mrow[c0] : (h >> 20);
                                unassigned -= level * 100;
                                const int used = unassigned <= 0 ? level + unassigned / 100 : level;
                                if (lane == 0) { sel_ch[nsel] = c0; sel_cap[nsel] = level; sel_used[nsel] = used; }
                                nsel += 1;
                                if (unassigned <= 0) { covered = true; break; }
                                h = phy_keys_best<W>(key);
                            }
                            if (covered) { a_path = best; break; }
                            alive &= ~(1u << best);  // sorted_free_channels.pop(row)
                            nsel = 0;
                        }
                    } else {
                        uint32_t cols[W];
                        double *r0w = scratch_d;   // free until the per-step outputs
                        phy_columns<W>(occ, tb, p, lane, metric_mode, cols, r0w);
                        int head_level[ORLG_PHY_MAX_K];
                        double head_metric[ORLG_PHY_MAX_K];
                        unsigned alive = 0;
                        // the row the first pick below will choose keeps its per-channel (level, metric) values: no second pass
                        int keep_idp = -1, keep_lv[W], keep_l = -1;
                        double keep_mt[W], keep_m = 0.0;
#pragma unroll
                        for (int w = 0; w < W; ++w) { keep_lv[w] = -1; keep_mt[w] = 0.0; }
#pragma unroll
                        for (int idp = 0; idp < ORLG_PHY_MAX_K; ++idp) {
                            head_level[idp] = -1; head_metric[idp] = 0.0;
                            if (idp < K) {
                                int lv[W];
                                double mtr[W];
                                phy_row_metrics<W>(occ, tb, p, acc, idp, base + idp, p.mod_t + (size_t)(row * K + idp) * p.cpad, lane, metric_mode, flat, lv, mtr, cols, r0w, dv);
                                int bl, bc;
                                double bm;
                                phy_row_best<W>(lv, mtr, lane, bl, bm, bc);
                                if (bl >= 0) {
                                    head_level[idp] = bl; head_metric[idp] = bm; alive |= 1u << idp;
                                    if (keep_idp < 0 || (!first_row && (bl > keep_l || (with_metric && bl == keep_l && bm > keep_m)))) {
                                        keep_idp = idp; keep_l = bl; keep_m = bm;
#pragma unroll
                                        for (int w = 0; w < W; ++w) { keep_lv[w] = lv[w]; keep_mt[w] = mtr[w]; }
                                    }
                                }
                            }
                        }
                        SEC(4);  // policy: channel selection
                        for (;;) {
                            int best = -1, bl = -1;
                            double bm = 0.0;
#pragma unroll
                            for (int idp = 0; idp < ORLG_PHY_MAX_K; ++idp)
                                if (idp < K && ((alive >> idp) & 1u)) {
                                    if (best < 0 || (!first_row && (head_level[idp] > bl ||
                                                                    (with_metric && head_level[idp] == bl && head_metric[idp] > bm)))) {
                                        best = idp; bl = head_level[idp]; bm = head_metric[idp];
                                    }
                                }
                            if (best < 0) break;
                            int lv[W];
                            double mtr[W];
                            const uint8_t *mrow = p.mod_t + (size_t)(row * K + best) * p.cpad;
                            if (best == keep_idp) {
#pragma unroll
                                for (int w = 0; w < W; ++w) { lv[w] = keep_lv[w]; mtr[w] = keep_mt[w]; }
                                keep_idp = -1;
                            } else {
                                phy_row_metrics<W>(occ, tb, p, acc, best, base + best, mrow, lane, metric_mode, flat, lv, mtr, cols, r0w, dv);
                            }
                            int unassigned = demand;
                            nsel = 0;
                            bool covered = false;
                            while (nsel < ORLG_PHY_MAX_CH) {
                                int l0, c0;
                                double m0;
                                phy_row_best<W>(lv, mtr, lane, l0, m0, c0);
                                if (l0 < 0) break;
#pragma unroll
                                for (int w = 0; w < W; ++w)
                                    if (64 * w + lane == c0) lv[w] = -1;
                                const int level = flat ? (int)mrow[c0] : l0;
                                unassigned -= level * 100;
                                const int used = unassigned <= 0 ? level + unassigned / 100 : level;
                                if (lane == 0) { sel_ch[nsel] = c0; sel_cap[nsel] = level; sel_used[nsel] = used; }
                                nsel += 1;
                                if (unassigned <= 0) { covered = true; break; }
                            }
                            if (covered) { a_path = best; break; }
                            alive &= ~(1u << best);  // sorted_free_channels.pop(row)
                            nsel = 0;
                        }
                
                    }
                }
                wave_sync();
            }

            SEC(5);  // provision
            bool accepted = false;
            double gn_last = __longlong_as_double(0x7ff8000000000000ll);   // NaN: no GN check in this step
            const bool dirbit = req_src > req_dst;
            if (a_path > 10 && a_path - 20 < K && nsel > 0) {
                // ---- virtual layer: _service_acceptance(True), _provision_virtual_path (:280-288, 625-659)
                const int idp = a_path - 20, gid = base + idp;
                const int key = (req_src * N + req_dst) * K + idp;
                CsList l = cs_load(gcs, gcs_n, key, lane, p.cs_len);
                bool ok = true;
                for (int ci = 0; ci < nsel && ok; ++ci) {
                    const int q = cs_find(l, sel_ch[ci], lane);
                    if (q < 0) { ok = false; break; }
                    const uint32_t en = cs_get(l, q);
                    const int take = sel_used[ci];
                    if (cs_free(en) < take) { ok = false; break; }  // the reference raises here
                    cs_remove(l, q, lane);
                    cs_append(l, cs_pack(cs_ch(en), cs_used(en) + take, cs_free(en) - take, cs_cap(en)), lane);
                }
                if (ok) {
                    cs_store(gcs, gcs_n, key, l, lane);
                    if (lane == 0) { ws->c[1] += 1; ws->c[3] += 1; ws->c[5] += demand; ws->c[7] += demand; }
                    accepted = true;
                    if (n_running < Q) {
                        if (lane == 0) gq[n_running] = ws->req_arrival + ws->req_holding;
                        {
                            const uint32_t hw_v = lane < nsel ? (uint32_t)(sel_ch[lane] | (sel_used[lane] << 9) | (1 << 14)) : 0xffffu;
                            rec_store(grec + n_running, &ws->req_arrival, (uint32_t)next_seq, gid, nsel, 1 | (dirbit ? 2 : 0), hw_v, lane);
                            if (DF && gsum) svc_side_store(gsum, gseq, n_running, gid, 1 | (dirbit ? 2 : 0), nsel, hw_v, (uint32_t)next_seq, lane);
                        }
                        {   // _add_release: the release joins the near-term buffer when it falls before the horizon
                            const double rel = readlane_d(ws->req_arrival + ws->req_holding, 0);
                            const int qidx = n_running;
                            n_running += 1;
                            if (rel <= nb.horizon) {
                                if (nb.n < ORLG_PHY_NB) {
                                    if (lane == 0) { nb.t[nb.n] = rel; nb.qi[nb.n] = (uint16_t)qidx; }
                                    nb.n += 1;
                                } else {
                                    wave_sync();
                                    if (!nb_rebuild(nb, gq, n_running, current_time, p.holding_lambda, lane) && lane == 0) ws->q_overflow |= 8;
                                }
                            }
                        }
                        next_seq += 1;
                    } else if (lane == 0) {
                        ws->q_overflow |= 1;
                    }
                    wave_sync();
                }
            } else if (a_path >= 0 && a_path < K && nsel > 0) {
                const int gid = base + a_path;
                const OrlgPathRec *rec = tb.recs + gid;
                const int hops = rec->hops;
                // requested now, used after the checks: the GSNR of the chosen channels (lane i: channel i) and, when a channel is
                // only partly used, the channel_state list it will join
                const int cs_key_p = (req_src * N + req_dst) * K + a_path;
                const int my_ch = lane < nsel ? sel_ch[lane] : 0;
                const bool ch_ok = lane < nsel && my_ch >= 0 && my_ch < C;
                const double my_gsnr = ch_ok ? p.gsnr_t[(size_t)(row * K + a_path) * p.cpad + my_ch] : 0.0;
                const int my_used = lane < nsel ? sel_used[lane] : 0, my_cap = lane < nsel ? sel_cap[lane] : 0;
                const bool any_partial_p = ballot(lane < nsel && my_used != my_cap) != 0ull;
                CsList csl;
                csl.e = 0u; csl.n = 0; csl.cap = p.cs_len;
                if (any_partial_p) csl = cs_load(gcs, gcs_n, cs_key_p, lane, p.cs_len);
                // is_path_free_on_channels (:1019-1027): lanes = (channel, hop) pairs
                // (the heuristics pick among the channels that are free on the path right now: only external actions need the look)
                bool pass = true;
                if (policy == ORLG_PHY_POLICY_EXTERNAL) {
                    bool bad = false;
                    for (int i = lane; i < nsel * hops; i += 64) {
                        const int ci = i / hops, h = i - ci * hops;
                        const int ch = sel_ch[ci];
                        if (ch < 0 || ch >= C) { bad = true; } else {
                            bad = bad || !((occ[(int)rec->link[h] * W + (ch >> 6)] >> (ch & 63)) & 1ull);
                        }
                    }
                    pass = ballot(bad) == 0ull;
                }
                if (GN && p.gn_on && pass) {
                    // GN gate (not in the reference): every chosen channel must reach the level the table promised
                    const int mrow_g = (row * K + a_path) * p.cpad;
                    for (int ci = 0; ci < nsel && pass; ++ci) {
                        const double gdb = gn_gsnr<W>(p, occ, rec, mrow_g, sel_ch[ci], lane SEC_ARGS);
                        const int level = popc64(ballot(gdb >= gn_thr_l));   // thresholds reached (lane q: threshold q)
                        gn_last = gdb;
                        if (level < sel_cap[ci]) pass = false;
                    }
                }
                if (pass) {
                    // _provision_path (:544-623): one lane per hop clears the channels on its link
                    for (int ci = 0; ci < nsel; ++ci) mc_before(occ, mc, sel_ch[ci], lane);
                    {
                        u64 clr[W];   // the chosen channels as masks (wave-uniform): one read-modify-write per word and link
#pragma unroll
                        for (int w = 0; w < W; ++w) clr[w] = 0ull;
                        for (int ci = 0; ci < nsel; ++ci) {
                            const int ch = __builtin_amdgcn_readlane(my_ch, ci);
#pragma unroll
                            for (int w = 0; w < W; ++w)
                                if ((ch >> 6) == w) clr[w] |= 1ull << (ch & 63);
                        }
                        if (lane < hops) {
                            u64 *rowp = occ + (int)rec->link[lane] * W;
#pragma unroll
                            for (int w = 0; w < W; ++w)
                                if (clr[w] != 0ull) rowp[w] &= ~clr[w];
                        }
                    }
                    if (mc.on) {
                        wave_sync();
                        for (int ci = 0; ci < nsel; ++ci) mc_after<LT>(occ, mc, sel_ch[ci], lane);
                    }
                    if (gnv) {   // the nodes of the path lose free links on these channels
                        uint4 cv;
                        if (!RSSP && policy != ORLG_PHY_POLICY_EXTERNAL) {   // the pair's records are on lanes (nvq)
                            cv.x = (uint32_t)__builtin_amdgcn_readlane((int)nvq.x, 2 * a_path); cv.y = (uint32_t)__builtin_amdgcn_readlane((int)nvq.y, 2 * a_path);
                            cv.z = (uint32_t)__builtin_amdgcn_readlane((int)nvq.z, 2 * a_path); cv.w = (uint32_t)__builtin_amdgcn_readlane((int)nvq.w, 2 * a_path);
                        } else {
                            cv = p.nvrec[2 * gid];
                        }
                        if (lane < nsel) nv_update(gnv, cv, my_ch, false);
                    }
                    // statistics, in channel order (the GSNR sum is a float64 accumulation)
                    double tg = ws->total_gsnr;
                    for (int ci = 0; ci < nsel; ++ci) tg += readlane_d(my_gsnr, ci);
                    if (lane == 0) {
                        long long tm = ws->total_mod;
                        for (int ci = 0; ci < nsel; ++ci) tm += sel_cap[ci];
                        ws->total_gsnr = tg; ws->total_mod = tm;
                        ws->channels_accepted += nsel;
                        // _service_acceptance(False) (:767-778)
                        ws->c[1] += 1; ws->c[3] += 1; ws->c[5] += demand; ws->c[7] += demand;
                        ws->total_path_length += tb.path_len[gid];
                        ws->total_path_index += a_path + 1;
                        ws->physical_accepted += 1;
                    }
                    // (the loads above are consumed: from here on stores)
                    // partially used channels enter channel_state (:600-602)
                    if (any_partial_p) {
                        bool overflow = false;
                        for (int ci = 0; ci < nsel; ++ci) {
                            const int cap = sel_cap[ci], used = sel_used[ci];
                            if (used != cap) {
                                if (!cs_append(csl, cs_pack(sel_ch[ci], used, cap - used, cap), lane)) overflow = true;
                            }
                        }
                        cs_store(gcs, gcs_n, cs_key_p, csl, lane);
                        if (overflow && lane == 0) ws->q_overflow |= 4;
                    }
                    accepted = true;
                    // _add_release: compact queue, append at n_running
                    if (n_running < Q) {
                        if (lane == 0) gq[n_running] = ws->req_arrival + ws->req_holding;
                        {
                            const uint32_t hw_p = lane < nsel ? (uint32_t)(my_ch | (my_used << 9) | ((my_used != my_cap ? 1 : 0) << 14)) : 0xffffu;
                            rec_store(grec + n_running, &ws->req_arrival, (uint32_t)next_seq, gid, nsel, dirbit ? 2 : 0, hw_p, lane);
                            if (DF && gsum) svc_side_store(gsum, gseq, n_running, gid, dirbit ? 2 : 0, nsel, hw_p, (uint32_t)next_seq, lane);
                        }
                        {   // _add_release: the release joins the near-term buffer when it falls before the horizon
                            const double rel = readlane_d(ws->req_arrival + ws->req_holding, 0);
                            const int qidx = n_running;
                            n_running += 1;
                            if (rel <= nb.horizon) {
                                if (nb.n < ORLG_PHY_NB) {
                                    if (lane == 0) { nb.t[nb.n] = rel; nb.qi[nb.n] = (uint16_t)qidx; }
                                    nb.n += 1;
                                } else {
                                    wave_sync();
                                    if (!nb_rebuild(nb, gq, n_running, current_time, p.holding_lambda, lane) && lane == 0) ws->q_overflow |= 8;
                                }
                            }
                        }
                        next_seq += 1;
                    } else if (lane == 0) {
                        ws->q_overflow |= 1;
                    }
                    wave_sync();
                }
            }

            SEC(6);  // outputs
            // per-step outputs
            if (p.out_mask) {
                const int om = p.out_mask;
                const size_t o = (size_t)t * p.B + env;
                double cuts = 0.0, rss = 0.0;
                const bool want_c = om & (1 << ORLG_PHY_OUT_CUTS), want_r = om & (1 << ORLG_PHY_OUT_RSS);
                if (mc.on) {
                    cuts = (double)mc.total_runs / (double)C;
                    if (want_r && mc.defer) {
                        mc.stamp += 1;   // this step's output point: its sum is formed with the block's (mc_flush, below)
                    } else if (want_r) {
                        // the terms lane 0 rewrote in this step are complete (same wave: in order).  Through LDS, then the
                        // reference's channel-order float64 sum: 8 terms per LDS round trip (one trip per term made this sum
                        // half of the step), the zero terms past C leave the sum as it is
                        nv_fence();
                        if (!LT) {
#pragma unroll
                            for (int w = 0; w < W; ++w) scratch_d[64 * w + lane] = 64 * w + lane < C ? mc.cterm[64 * w + lane] : 0.0;
                            wave_sync();
                        }
                        const double r = ordered_sum_lds(scratch_d, C);
                        wave_sync();
                        rss = r / (double)C;
                    }
                } else if (want_c || want_r) {
                    int tr_unused;
                    phy_column_metrics<W>(occ, tb.sqrt_tab, E, C, lane, scratch_d, want_c, want_r, p.use_masks != 0, cuts, rss, tr_unused);
                }
                if (om & (1 << ORLG_PHY_OUT_CHANNELS)) {
                    auto oc = ORLG_GPTR(int16_t, tb.outs[ORLG_PHY_OUT_CHANNELS]) + o * ORLG_PHY_MAX_CH;
                    if (lane < ORLG_PHY_MAX_CH) oc[lane] = lane < nsel ? (int16_t)sel_ch[lane] : (int16_t)-1;
                }
                if (om & (1 << ORLG_PHY_OUT_CH_USED)) {
                    auto oc = ORLG_GPTR(int16_t, tb.outs[ORLG_PHY_OUT_CH_USED]) + o * ORLG_PHY_MAX_CH;
                    if (lane < ORLG_PHY_MAX_CH) oc[lane] = lane < nsel ? (int16_t)sel_used[lane] : (int16_t)0;
                }
                if (lane == 0) {
                    if (om & (1 << ORLG_PHY_OUT_PATH)) ORLG_GPTR(int32_t, tb.outs[ORLG_PHY_OUT_PATH])[o] = a_path;
                    if (om & (1 << ORLG_PHY_OUT_NCH)) ORLG_GPTR(int32_t, tb.outs[ORLG_PHY_OUT_NCH])[o] = nsel;
                    if (om & (1 << ORLG_PHY_OUT_ACCEPTED)) ORLG_GPTR(uint8_t, tb.outs[ORLG_PHY_OUT_ACCEPTED])[o] = accepted ? 1 : 0;
                    if (om & (1 << ORLG_PHY_OUT_REQUEST))
                        ORLG_GPTR(orlg_v4i, tb.outs[ORLG_PHY_OUT_REQUEST])[o] = orlg_v4i{req_sid, req_src, req_dst, demand};
                    if (om & (1 << ORLG_PHY_OUT_ARRIVAL)) ORLG_GPTR(double, tb.outs[ORLG_PHY_OUT_ARRIVAL])[o] = ws->req_arrival;
                    if (om & (1 << ORLG_PHY_OUT_HOLDING)) ORLG_GPTR(double, tb.outs[ORLG_PHY_OUT_HOLDING])[o] = ws->req_holding;
                    if (om & (1 << ORLG_PHY_OUT_DEFRAG)) {  // the counters as the info dict sees them: before this step's defragmentation
                        auto od = ORLG_GPTR(int32_t, tb.outs[ORLG_PHY_OUT_DEFRAG]) + o * 3;
                        od[0] = ws->counted_moves; od[1] = ws->counted_moves_groom; od[2] = ws->counted_defrag_cycles;
                    }
                    if (om & (1 << ORLG_PHY_OUT_GN)) ORLG_GPTR(double, tb.outs[ORLG_PHY_OUT_GN])[o] = gn_last;
                    if (want_c) ORLG_GPTR(double, tb.outs[ORLG_PHY_OUT_CUTS])[o] = cuts;
                    if (want_r && !mc.defer) ORLG_GPTR(double, tb.outs[ORLG_PHY_OUT_RSS])[o] = rss;
                }
                // a block of deferred sums is due: 64 steps, or a log that the next step's rewrites might overrun
                if (mc.defer && (mc.stamp == 64 || mc.nlog > ORLG_RLOG_CAP - 160)) {
                    nv_fence();
                    mc_flush<W>(mc, scratch_d, reinterpret_cast<u64 *>(scratch), C, p.cpad, lane, tb.outs[ORLG_PHY_OUT_RSS], (size_t)p.B);
                    wave_sync();
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
                ws->counted_moves = 0; ws->counted_moves_groom = 0; ws->counted_defrag_cycles = 0;
            }
            wave_sync();
        }

        bool defrag_now = false;   // this step ends with a defragmentation cycle (services_processed % defrag_period == 0)
        SEC(7);  // next arrival
        // ============================================================== _next_service (phy_rmsa_env.py:969-1017)
        if (p.mode != ORLG_MODE_EPISODE_RESET && !new_service) {
            // the arrival process does not depend on the network state: requests come from the ring of pre-generated arrivals
            // (five random() draws each, rmsa-style: phy_rmsa_env.py:971-986), refilled 64 at a time when it runs dry
            if (ring_cnt == 0) {
                SEC(8);  // refill
                static_assert(ORLG_MT_N * 4 == 156 * 16, "MT19937 state = 156 rows of 16 bytes");
                const OrlgPhyParams __attribute__((address_space(4))) *kq =
                    (const OrlgPhyParams __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
                const uint4 *g_mt = reinterpret_cast<const uint4 *>(kq->mt + (size_t)env * ORLG_MT_N);
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
                int idx_s = mt_idx;
                const int got = refill_requests(mt_lds, kq->ring_iat + (size_t)env * ORLG_RING, kq->ring_ht + (size_t)env * ORLG_RING,
                                                kq->ring_req + (size_t)env * ORLG_RING, tb.src_cum, tb.dst_cum, tb.br_cum, &idx_s, N, NBR,
                                                p.arrival_lambda, p.holding_lambda, env);
                m0 = l_mt[lane]; m1 = l_mt[lane + 64];
                if (lane < 156 - 128) m2 = l_mt[lane + 128];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // this wave's reads of the buffer are done
                if (lane == 0) atomicExch(mt_lock, 0);
                uint4 *o_mt = reinterpret_cast<uint4 *>(kq->mt + (size_t)env * ORLG_MT_N);
                o_mt[lane] = m0; o_mt[lane + 64] = m1;
                if (lane < 156 - 128) o_mt[lane + 128] = m2;
                // the ring entries other lanes wrote are read back below: same CU, the stores only have to be complete
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                wave_sync();
                mt_idx = idx_s; ring_cnt = got; ring_pos = 0;
                SEC(7);
            }
            if (!pf_ring_ok) ring_fetch();   // the ring was empty when this step began (or the launch does not step)
            const double r_iat = __hiloint2double(__builtin_amdgcn_readlane((int)pf_ring, 1), __builtin_amdgcn_readlane((int)pf_ring, 0));
            const double ht = __hiloint2double(__builtin_amdgcn_readlane((int)pf_ring, 3), __builtin_amdgcn_readlane((int)pf_ring, 2));
            const uint32_t rq = (uint32_t)__builtin_amdgcn_readlane((int)pf_ring, 4);
            const bool have_next = pf_next_ok;
            const double next_iat = __hiloint2double(__builtin_amdgcn_readlane((int)pf_ring, 9), __builtin_amdgcn_readlane((int)pf_ring, 8));
            ring_pos += 1; ring_cnt -= 1;
            if (p.mode == ORLG_MODE_STEP) ring_fetch();   // the entry of the next step
            else pf_ring_ok = false;
            const double at = current_time + r_iat;
            current_time = at;
            const int src = (int)(rq & 0xffu), dst = (int)((rq >> 8) & 0xffu), bri = (int)(rq >> 16);
            req_sid = eproc;
            req_src = src; req_dst = dst; req_br = bri;
            new_service = 1;
            eproc += 1;
            if (lane == 0) {
                const int br_val = tb.bit_rates[bri];
                ws->c[0] += 1; ws->c[2] += 1; ws->c[4] += br_val; ws->c[6] += br_val;
                ws->req_arrival = at; ws->req_holding = ht;
            }
            SEC(9);  // release: buffer / rebuild
            // ---- release every service with release time <= now in time order (:1009-1017, _release_path :781-861):
            // with the virtual layer the order of simultaneous releases decides who frees a shared channel
            wave_sync();
            if (current_time > nb.horizon) {
                if (!nb_rebuild(nb, gq, n_running, current_time, p.holding_lambda, lane) && lane == 0) ws->q_overflow |= 8;
            }
            // the scan looks one arrival ahead: the earliest release up to the NEXT arrival's time is this step's victim when it
            // is due now, and otherwise the service the next step will release first -- its record is requested right away
            // (ReleaseAhead; a defragmentation cycle in between drops the look-ahead)
            const double look_time = have_next ? current_time + next_iat : current_time;
            int ahead_q = -1;
            for (;;) {
                int victim, vpos;
                double vt;
                nb_first_due(nb, look_time, lane, victim, vpos, vt);
                if (victim >= 0 && vt > current_time) { ahead_q = victim; victim = -1; }
                if (victim < 0) break;
                SEC(10);  // release apply
                const bool ahead = victim == ra.q;   // the service looked up at the start of the step: its data is here
                // the record stays on lanes (lane i < 12: dword i): no array of channels in private memory
                const uint32_t rv = ahead ? ra.rec : rec_dword(grec, victim, lane);
                const uint32_t d3 = (uint32_t)__builtin_amdgcn_readlane((int)rv, 3);
                const int sv_gid = (int)(d3 & 0xffffu), sv_nch = (int)((d3 >> 16) & 0xffu), sv_flags = (int)(d3 >> 24);
                // lane i < nch: channel i of the service (channel | used << 9 | partial << 14)
                const uint32_t pair_dw = (uint32_t)__shfl((int)rv, 4 + (lane >> 1));
                const int raw_l = lane < sv_nch ? (int)((pair_dw >> (16 * (lane & 1))) & 0xffffu) : 0;
                const OrlgPathRec *rec = tb.recs + sv_gid;
                const int pair = tb.path_pair[sv_gid];
                const int pa = pair / N, pb = pair - pa * N;
                const int ssrc = (sv_flags & 2) ? pb : pa, sdst = (sv_flags & 2) ? pa : pb;
                const int idp = sv_gid - tb.pair_base[pair];
                const int key = (ssrc * N + sdst) * K + idp;
                u64 freemask[W];  // channels to return on every link of the path
                const bool any_partial = ballot(lane < sv_nch && (raw_l & (1 << 14))) != 0ull;
#pragma unroll
                for (int w = 0; w < W; ++w) freemask[w] = 0ull;
                for (int ci = 0; ci < sv_nch; ++ci) {
                    const int raw = __builtin_amdgcn_readlane(raw_l, ci);
                    if (!(raw & (1 << 14))) {
                        const int ch = raw & 0x1ff;
#pragma unroll
                        for (int w = 0; w < W; ++w)
                            if ((ch >> 6) == w) freemask[w] |= 1ull << (ch & 63);
                    }
                }
                // every load of this release first (what was not requested ahead), every store last
                const bool move_last = victim != n_running - 1;
                uint32_t lastv = 0u;   // the queue's last record (lanes 0..11) and release time (12, 13): it takes the victim's place
                if (move_last) {
                    lastv = lane < 12 ? reinterpret_cast<const uint32_t *>(grec + (n_running - 1))[lane]
                                      : lane < 14 ? reinterpret_cast<const uint32_t *>(gq + (n_running - 1))[lane - 12] : 0u;
                    if (DF && gsum) {   // its side-array entries travel with it (lanes 14, 15: summary; 16: seq)
                        if (lane == 14 || lane == 15) lastv = reinterpret_cast<const uint32_t *>(gsum + (n_running - 1))[lane - 14];
                        if (lane == 16) lastv = gseq[n_running - 1];
                    }
                }
                uint32_t cvl = 0u;
                if (gnv && lane < 4) cvl = reinterpret_cast<const uint32_t *>(p.nvrec + 2 * sv_gid)[lane];
                CsList l;
                l.e = 0u; l.n = 0; l.cap = p.cs_len;
                if (any_partial) {
                    l = cs_load(gcs, gcs_n, key, lane, p.cs_len);
                    for (int ci = 0; ci < sv_nch; ++ci) {
                        const int raw = __builtin_amdgcn_readlane(raw_l, ci);
                        if (!(raw & (1 << 14))) continue;
                        const int ch = raw & 0x1ff, mine = (raw >> 9) & 0x1f;
                        const int q = cs_find(l, ch, lane);
                        if (q < 0) continue;  // cannot happen for states produced by this kernel
                        const uint32_t en = cs_get(l, q);
                        cs_remove(l, q, lane);
                        if (cs_used(en) == mine) {  // last user of the channel: it goes dark
#pragma unroll
                            for (int w = 0; w < W; ++w)
                                if ((ch >> 6) == w) freemask[w] |= 1ull << (ch & 63);
                        } else {
                            cs_append(l, cs_pack(ch, cs_used(en) - mine, cs_free(en) + mine, cs_cap(en)), lane);
                        }
                    }
                }
                if (mc.on) {
#pragma unroll
                    for (int w = 0; w < W; ++w)
                        for (u64 m = readlane64(freemask[w], 0); m; m &= m - 1) mc_before(occ, mc, 64 * w + ctz64(m), lane);
                }
                if (lane < rec->hops) {
                    u64 *rowp = occ + (int)rec->link[lane] * W;
#pragma unroll
                    for (int w = 0; w < W; ++w) rowp[w] |= freemask[w];
                }
                if (mc.on) {
                    wave_sync();
#pragma unroll
                    for (int w = 0; w < W; ++w)
                        for (u64 m = readlane64(freemask[w], 0); m; m &= m - 1) mc_after<LT>(occ, mc, 64 * w + ctz64(m), lane);
                }
                if (gnv) {   // the returned channels: lane = channel of word w, the nodes of the path gain free links
                    uint4 cv_rel;
                    cv_rel.x = (uint32_t)__builtin_amdgcn_readlane((int)cvl, 0); cv_rel.y = (uint32_t)__builtin_amdgcn_readlane((int)cvl, 1);
                    cv_rel.z = (uint32_t)__builtin_amdgcn_readlane((int)cvl, 2); cv_rel.w = (uint32_t)__builtin_amdgcn_readlane((int)cvl, 3);
#pragma unroll
                    for (int w = 0; w < W; ++w)
                        if (freemask[w] != 0ull && ((freemask[w] >> lane) & 1ull)) nv_update(gnv, cv_rel, 64 * w + lane, true);
                }
                // the stores: the rewritten channel_state list ...
                if (any_partial) cs_store(gcs, gcs_n, key, l, lane);
                // ... and the swap-remove: the last live entry takes the victim's place (its near-buffer entry follows it) ...
                n_running -= 1;
                if (move_last) {
                    if (lane < 12) reinterpret_cast<uint32_t *>(grec + victim)[lane] = lastv;
                    else if (lane < 14) reinterpret_cast<uint32_t *>(gq + victim)[lane - 12] = lastv;
                    else if (DF && gsum && lane < 16) reinterpret_cast<uint32_t *>(gsum + victim)[lane - 14] = lastv;
                    else if (DF && gsum && lane == 16) gseq[victim] = lastv;
                    for (int c0 = 0; c0 < nb.n; c0 += 64) {
                        const int c = c0 + lane;
                        if (c < nb.n && (int)nb.qi[c] == n_running) nb.qi[c] = (uint16_t)victim;
                    }
                }
                wave_sync();
                // ... and the victim's own buffer entry is replaced by the buffer's last one
                nb.n -= 1;
                if (vpos != nb.n && lane == 0) { nb.t[vpos] = nb.t[nb.n]; nb.qi[vpos] = nb.qi[nb.n]; }
                wave_sync();
                ra.q = -1;   // the queue has changed: what was looked up ahead is stale
            }
            ra.q = -1;
            if (DF && p.mode == ORLG_MODE_STEP && p.defrag_period > 0) defrag_now = uni((int)(ws->c[0] % p.defrag_period)) == 0;
            if (ahead_q >= 0 && !defrag_now && p.mode == ORLG_MODE_STEP) { ra.q = ahead_q; ra.rec = rec_dword(grec, ahead_q, lane); }
        }

        if (DF && p.mode == ORLG_MODE_STEP && p.defrag_period > 0) {
        SEC(11);  // defragmentation
            // periodic defragmentation (phy_rmsa_env.py:355-417): services_processed % defrag_period == 0
            wave_sync();
            if (__builtin_expect(defrag_now, 1)) {   // (most of the time of such a launch is spent in here: its loops get the registers)
                const bool park = LT && mc.on && mc.want_rss;
                if (park) {
#pragma unroll
                    for (int w = 0; w < W; ++w)
                        if (64 * w + lane < C) mc.cterm[64 * w + lane] = scratch_d[64 * w + lane];
                    wave_sync();
                }
                phy_defragmentation<W>(p, tb, occ, ws, grec, gsum, gseq, gcs, gcs_n, gcand, sel_ch, scratch_d, n_running, next_seq, current_time, req_src, req_dst, lane, gnv, mc SEC_ARGS);
                if (park) {   // (the cycle's own updates went to the HBM array; the terms past C are zero: ordered_sum_lds)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
                    for (int w = 0; w < W; ++w) scratch_d[64 * w + lane] = 64 * w + lane < C ? mc.cterm[64 * w + lane] : 0.0;
                    wave_sync();
                }
            }
        }

        if (p.mode == ORLG_MODE_STEP) {
            const bool done = (eproc == p.episode_length);
            if (lane == 0 && (p.out_mask & (1 << ORLG_PHY_OUT_DONE)))
                ORLG_GPTR(uint8_t, tb.outs[ORLG_PHY_OUT_DONE])[(size_t)t * p.B + env] = done ? 1 : 0;
            if (done && p.auto_reset) {
                eproc = 1;
                if (lane == 0) {
                    ws->episodes_done += 1;
                    ws->c[2] = 1; ws->c[3] = 0; ws->c[6] = tb.bit_rates[req_br]; ws->c[7] = 0;
                    ws->total_path_length = 0.0; ws->total_gsnr = 0.0; ws->total_path_index = 0; ws->total_mod = 0;
                    ws->channels_accepted = 0; ws->physical_accepted = 0;
                    ws->counted_moves = 0; ws->counted_moves_groom = 0; ws->counted_defrag_cycles = 0;
                }
                wave_sync();
            }
        }
    }

    if (mc.defer) {   // the last block's sums
        wave_sync();
        if (mc.stamp > 0) mc_flush<W>(mc, scratch_d, reinterpret_cast<u64 *>(scratch), C, p.cpad, lane, tb.outs[ORLG_PHY_OUT_RSS], (size_t)p.B);
        if (mc.log_overflow && lane == 0) ws->q_overflow |= 16;
        wave_sync();
    }
    SEC(13);  // state store
    // ------------------------------------------------------------------ LDS -> HBM
    wave_sync();
    {
        const OrlgPhyParams __attribute__((address_space(4))) *kp =
            (const OrlgPhyParams __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
        u64 *g = kp->occ + (size_t)env * NW;
        for (int i = lane; i < NW; i += 64) g[i] = occ[i];
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
            go->ring_pos = ring_pos; go->ring_cnt = ring_cnt;
            if (ws->q_overflow) *kp->err_flag = 1;   // reported by the next entry point that waits for the stream
            go->next_seq = next_seq; go->counted_moves = ws->counted_moves; go->counted_moves_groom = ws->counted_moves_groom;
            go->counted_defrag_cycles = ws->counted_defrag_cycles;
        }
    }
    wave_sync();
    SEC(0);
    }  // work queue
    SEC_FLUSH;
}
