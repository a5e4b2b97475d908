/*
 * oracle/orlg_oracle_phy.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  CPU restatement of the QoT-aware
 * environment core (optical_rl_gym/envs/phy_rmsa_env.py); see orlg_oracle_phy.c for scope and pinning.
 */
#ifndef ORLG_ORACLE_PHY_H
#define ORLG_ORACLE_PHY_H
#include "orlg_oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_PHY_MAX_CH 12

typedef struct orc_phy_config {
    int32_t num_channels;   /* L + C + S bands = 80 + 80 + 108 (optical_network_env.py:78-84) */
    int32_t episode_length;
    int32_t num_bit_rates;
    int32_t k_table;        /* number of k-path columns in the QoT tables */
    int32_t grooming;       /* env.grooming (phy_rmsa_env.py:57): consulted by bmfa / bmfa_rss only */
    int32_t defrag_period;  /* phy_rmsa_env.py:54: 0 = no periodic defragmentation */
    int32_t number_moves;   /* phy_rmsa_env.py:55 */
    int32_t defrag_metric;  /* env.metric (phy_rmsa_env.py:56): 0 = 'cut', 1 = anything else (RSS) */
    double arrival_lambda, holding_lambda;
    const int32_t *bit_rates;
    const double *bit_rate_cum, *src_cum, *dst_cum;
    const int32_t *pair_table_row;   /* [N*N] row of connections_detail matching (src, dst) in either order (phy_rmsa_env.py:562-565) */
    const uint8_t *modulation_level; /* [rows][channels][k_table] */
    const double *gsnr;              /* [rows][channels][k_table] */
    const int32_t *link_ends;        /* [E][2] node ids of every link */
    const int32_t *path_node_off;    /* [num_paths+1] CSR into path_nodes */
    const int32_t *path_nodes;       /* node ids along every path */
    /* GN-model admission check of the chosen channels (north_star: "GN-model OSNR admission check").  NOT in the reference,
     * which gates by table only (phy_rmsa_env.py:596): PARITY UNPINNED.  The arithmetic is the restatement of
     * examples/calculate_osnr.py (orlg_oracle_osnr.c) applied to the LIVE occupancy: a channel chosen on the physical layer is
     * checked with every lit channel of every link of the path as an interferer (centre frequencies by channel index, the
     * interferer's modulation taken from the QoT table row of the candidate path), spans = int(length // 80 km) + 1 equal
     * spans per link (examples/create_topology_gn.py:124); level = number of thresholds met; the service is blocked when a
     * chosen channel's level falls short of the capacity level the table promised. */
    int32_t gn_on, gn_num_thresholds;
    double gn_launch_power_w, gn_channel_bandwidth_hz, gn_attenuation, gn_noise_figure;
    const double *gn_center_frequency_hz;   /* [channels] */
    const int32_t *gn_link_num_spans;       /* [E] */
    const double *gn_link_span_length_km;   /* [E] */
    const double *gn_thresholds_db;         /* [gn_num_thresholds] ascending */
} orc_phy_config;

enum { ORC_PHY_POLICY_BMFA = 0, ORC_PHY_POLICY_BMFA_RSS = 1, ORC_PHY_POLICY_SAPFF = 2, ORC_PHY_POLICY_BMFF = 3,
       ORC_PHY_POLICY_SAPBM = 4, ORC_PHY_POLICY_FAFF = 5, ORC_PHY_POLICY_FAFF_RSS = 6 };

typedef struct orc_phy_action {
    int32_t path;   /* -2 = blocked; 20 + idp = served on the virtual layer (phy_rmsa_env.py:280-288) */
    int32_t n;
    int32_t ch[ORC_PHY_MAX_CH], cap[ORC_PHY_MAX_CH];
    double used[ORC_PHY_MAX_CH], free_[ORC_PHY_MAX_CH];
} orc_phy_action;

typedef struct orc_phy_result {
    double reward;
    int32_t done, accepted;
    double number_cuts_total, rss_total_metric, total_path_length, avrage_gsnr, average_path_index;
    double service_blocking_rate, episode_service_blocking_rate, bit_rate_blocking_rate,
        episode_bit_rate_blocking_rate;
    int64_t total_modulation_level, channels_accepted, path_index, physical_paths;
    double num_moves;       /* counted_moves / 2 + counted_moves_groom (phy_rmsa_env.py:340) */
    int64_t num_moves_groom, num_defrag_cycle;
    double gn_gsnr_db;      /* GN gate: GSNR of the last channel checked in this step (NaN: no check) */
} orc_phy_result;

typedef struct orc_phy_trace {
    int32_t *service_id, *src, *dst, *bit_rate, *act_path, *n_channels, *channels, *ch_cap;
    double *arrival, *holding, *ch_used;
    uint8_t *accepted, *done;
    int64_t *services_accepted, *total_modulation_level, *channels_accepted, *path_index, *physical_paths,
        *n_running, *free_total;
    double *number_cuts_total, *rss_total_metric, *total_path_length, *avrage_gsnr, *average_path_index,
        *episode_service_blocking_rate, *bit_rate_blocking_rate, *current_time;
    double *num_moves;
    int64_t *num_moves_groom, *num_defrag_cycle;
    double *gn_gsnr_db;
} orc_phy_trace;

typedef struct orc_phy_env orc_phy_env;

orc_phy_env *orc_phy_create(const orc_topology *topo, const orc_phy_config *cfg, uint64_t seed);
void orc_phy_seed(orc_phy_env *e, uint64_t seed);
void orc_phy_reseed(orc_phy_env *e, uint64_t seed);
void orc_phy_destroy(orc_phy_env *e);
void orc_phy_reset(orc_phy_env *e, int only_episode_counters);
void orc_phy_get_request(const orc_phy_env *e, orc_request *out);
void orc_phy_policy(orc_phy_env *e, int policy, orc_phy_action *act);
void orc_phy_step(orc_phy_env *e, const orc_phy_action *act, orc_phy_result *out);
void orc_phy_get_counters(const orc_phy_env *e, orc_counters *out);
double orc_phy_current_time(const orc_phy_env *e);
void orc_phy_get_available_channels(const orc_phy_env *e, uint8_t *out);
int orc_phy_num_running(const orc_phy_env *e);
int orc_phy_channel_state(orc_phy_env *e, int src, int dst, int idp, int32_t *out /* [max_entries][4] */, int max_entries);
void orc_phy_run(orc_phy_env *e, int policy, int64_t n_steps, int reset_on_done, orc_phy_trace *tr);

#ifdef __cplusplus
}
#endif
#endif
