// orlg_device.h -- device-side data layout shared by the kernels (orlg_kernels.hip) and the host API
// (orlg_api.hip).  Names follow the reference's domain: links, slots, paths, services, release queue.
#pragma once
#include <stdint.h>

#define ORLG_WAVE 64
#define ORLG_MAX_WAVES_PER_BLOCK 8
#define ORLG_MT_N 624
#define ORLG_MT_M 397
#define ORLG_MAX_W 8          // 64-bit words per link: S <= 512
#define ORLG_MAX_HOPS 14
#define ORLG_NSLOT_STRIDE 8   // nslots table: [bit-rate index][spectral efficiency 0..7]
#define ORLG_NUM_OUTS 12
#define ORLG_RING 64           // arrivals generated per refill (one per lane)
#ifndef ORLG_GROUP_WAVES
#define ORLG_GROUP_WAVES 12    // four-environments-per-wave kernel: waves per workgroup at most (LDS decides how many fit): up to 3 per SIMD
#endif
#define ORLG_DIRECT_STEPS 4    // launches of at most this many steps read their ring entries straight from HBM

// One k-shortest-path record (16 B): Path.hops, Path.best_modulation.spectral_efficiency and the link
// "index" of every hop (utils.py:27-36, rmsa_env.py:479-483).
struct __attribute__((aligned(16))) OrlgPathRec {
    uint8_t hops, se;
    uint8_t link[ORLG_MAX_HOPS];
};

// Per-environment scalars kept in HBM between launches (one record per env, 192 B).
struct __attribute__((aligned(16))) OrlgEnvScalars {
    double current_time;                                   // optical_network_env.py:33
    double req_arrival, req_holding;                       // current_service.arrival_time / holding_time
    double g_throughput, g_compactness, g_last_update;     // topology.graph[...] rmsa_env.py:537-560
    int64_t c[8];                                          // orlg_counters order
    int64_t sum_bitrate_running;                           // sum of bit_rate over running services
    int64_t episodes_done;
    int32_t sum_slots_hops;                                // sum of number_slots * hops over running services
    int32_t n_running;
    int32_t req_src, req_dst, req_br, req_sid;             // pending request (br = bit-rate INDEX)
    int32_t mt_idx;                                        // MT19937 position (0..624)
    int32_t new_service;                                   // self._new_service
    int32_t q_overflow;                                    // release queue overflowed (error)
    int32_t ring_pos, ring_cnt;                            // pre-generated arrivals: next entry, entries left
    int32_t sum_span, sum_gaps;                            // sums over the per-link (span, gaps) cache (_get_network_compactness)
    int32_t q_head;                                        // release queue: slot of the earliest entry (time-sorted ring, see OrlgParams::qtime)
};
static_assert(sizeof(OrlgEnvScalars) == 192, "OrlgEnvScalars layout");

// Wave scalars that live in LDS during a launch (rarely touched counters; lane 0 updates them).
struct OrlgWaveScalars {
    int64_t c[8];
    int64_t sum_bitrate_running;
    int64_t episodes_done;
    double g_thr, g_comp, g_lu;        // topology.graph["throughput" | "compactness" | "last_update"]
    double req_arrival, req_holding;   // pending request times
    int32_t n_running, q_overflow;
};
static_assert(sizeof(OrlgWaveScalars) == 128, "OrlgWaveScalars layout");

enum { ORLG_MODE_STEP = 0, ORLG_MODE_INIT = 1, ORLG_MODE_EPISODE_RESET = 2 };
// device-side policy ids (== include/orlg.h ORLG_POLICY_*)
enum { ORLG_POLICY_EXT = -1, ORLG_POLICY_SP = 0, ORLG_POLICY_SAP = 1, ORLG_POLICY_LLP = 2, ORLG_POLICY_DEEP_SP = 3,
       ORLG_POLICY_DEEP_SAP = 4, ORLG_POLICY_DEEP_EXT = 5, ORLG_POLICY_PATH_EXT = 6 };
// per-step output slots (OrlgParams::outs)
enum { ORLG_OUT_PATH = 0, ORLG_OUT_SLOT, ORLG_OUT_ACCEPTED, ORLG_OUT_DONE, ORLG_OUT_REWARD, ORLG_OUT_REQUEST,
       ORLG_OUT_ARRIVAL, ORLG_OUT_HOLDING, ORLG_OUT_COMPACT, ORLG_OUT_COMPACT_DIFF, ORLG_OUT_AVG_LINK_COMPACT,
       ORLG_OUT_AVG_LINK_UTIL };

// Kernel parameters (passed by value).
struct OrlgParams {
    // sizes
    int32_t B, N, E, S, K, NBR, Q, NW;       // NW = E*W words of occupancy per env
    int32_t episode_length, n_steps, policy, auto_reset, mode, reward_mode, stats_level, j;
    int32_t obs_dim, out_mask;               // out_mask: bit i set <=> outs[i] != nullptr
    double arrival_lambda, holding_lambda;
    // per-env state in HBM
    uint64_t *occ;            // [B][NW]   free-slot bitmap, word (link*W + w)
    // release queue (the reference's heapq of (release time, service), optical_network_env.py:178-189): a time-sorted ring.  The
    // n_running entries sit at slots (q_head + rank) % Q in ascending release time; every other slot holds (+inf, 0).  A
    // release pops the head, an insert moves the later entries up by one slot.
    double *qtime;            // [B][Q]    release time, +inf = empty slot
    uint32_t *qdesc;          // [B][Q]    path gid | start << 14 | bit-rate index << 24
    uint32_t *mt;             // [B][624]
    OrlgEnvScalars *scal;     // [B]
    int32_t *hist;            // [B][4][NBR] requested, provisioned, episode requested, episode provisioned
    double *lstat;            // [B][4][E] utilization, external_fragmentation, compactness, last_update
    int32_t *lint;            // [B][lint_stride] per-link span | gaps << 16 (the cache behind _get_network_compactness)
    int32_t lint_stride;
    int32_t obs_f32;          // orlg_deeprmsa_obs_kernel writes float32 (o_obs then points to floats)
    uint4 *llog;              // [B][E][ORLG_LLOG_CAP] logged link-statistics updates (orlg_rmsa_group_kernel<.., DEFER>), scratch
    int32_t br_width, pad_br;  // bit_rate_selection="continuous": number of bit rates lower .. higher (the table bit_rates holds them), 0 = discrete
    double *ring_iat, *ring_ht;   // [B][64] pre-generated inter-arrival / holding times (in RNG stream order)
    uint32_t *ring_req;           // [B][64] src | dst << 8 | bit-rate index << 16
    // read-only tables: ONE blob in HBM that every workgroup stages into LDS (byte offsets t_*, 16-B aligned)
    const unsigned char *tables;
    int32_t tab_bytes;
    int32_t t_pair;           // int32  [N*N]        first path record of pair (src, dst)
    int32_t t_recs;           // PathRec[num_paths]
    int32_t t_nslots;         // uint16 [NBR][8]     get_number_slots per (bit rate, spectral efficiency)
    int32_t t_bitrates;       // int32  [NBR]
    int32_t t_brcum;          // double [NBR]
    int32_t t_srccum;         // double [N]
    int32_t t_dstcum;         // double [N*N]
    int32_t t_divs;           // double [S+1]        k / S   (link utilization: exact IEEE quotients, built on the host)
    int32_t t_inv;            // double [S/2+2]      1 / k   (link compactness)
    // per-call IO
    const int32_t *actions;
    void *outs[ORLG_NUM_OUTS];
    double *o_obs;
    int32_t *err_flag;        // the handle's sticky error word (mapped host memory): 1 = a release queue overflowed
    // work queue: every wave draws environments from *ticket until the launch's B are taken (ticket - ticket_base
    // is the environment index; the host advances ticket_base by B + launched waves per launch, no memset needed)
    uint32_t *ticket;
    uint32_t ticket_base;
    uint32_t ticket_stride;   // 1: static striding (env = wave, wave + waves, ...) instead of tickets: short launches
    // long launches of the four-environments-per-wave kernel, tickets in CHUNKS of steps (n_chunks > 1): ticket t = chunk t / quads
    // of quad t % quads -- a quad's launch is cut into n_chunks pieces that different waves may run, so the last round of
    // tickets is short; progress[quad] = chunks of the quad completed in this launch (the hand-off between the waves)
    int32_t n_chunks, chunk_steps;
    uint32_t *progress;
    // per-wave LDS layout (byte offsets from the wave's base) and size
    int32_t l_occ, l_qtime, l_qdesc, l_mt, l_lstat, l_hist, l_lint, l_scratch, l_wsc, l_ring, l_wave_bytes;
    int32_t l_shared_bytes;   // tables + output pointer block, in front of the per-wave regions
    int32_t l_outs;           // byte offset of the staged outs[] array
    // four-environments-per-wave kernel (orlg_group_kernels.hip): byte offsets inside one environment's LDS region, the region's
    // size (a wave's region = 4 environments), and g_mt = size of the workgroup's MT19937 staging buffer, which sits with its
    // lock word between the tables and the waves' regions
    int32_t g_occ, g_qtime, g_qdesc, g_lstat, g_hist, g_lint, g_env_bytes, g_mt, g_wave_bytes;
};
