/*
 * include/orlg.h -- C ABI of liborlg.so, the MI355X (gfx950) batched RMSA / DeepRMSA step() path.
 *
 * The reference (ehsan5890/optical-rl-gym-qot-aware) is pure Python and has NO native / FFI boundary
 * (SURVEY.md section 0.1, 8b): the contract it defines is the gym.Env object surface.  This header is
 * therefore the boundary a maintainer of the reference would bind with ctypes (INTEGRATION.md shows
 * the stub); every entry point names the reference method(s) it replaces.  All citations are relative
 * to the reference repository root.
 *
 * Conventions
 *   - plain C types only; every function returns ORLG_OK (0) or a negative error code and never
 *     throws; orlg_last_error() returns a thread-local message for the last failure.
 *   - one handle = B independent environments on one HIP device and one stream; a handle is not
 *     thread-safe; use one handle (one process) per GPU.
 *   - the library owns all device state.  Array arguments are caller-owned and may be HOST or
 *     DEVICE pointers (copies use hipMemcpyDefault on the handle's stream); tables passed to
 *     orlg_create() are copied before it returns.
 *   - there is no CPU fallback: without a HIP device orlg_create() fails with ORLG_ERR_NO_DEVICE.
 */
#ifndef ORLG_H
#define ORLG_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORLG_ABI_VERSION 3

enum {
    ORLG_OK = 0,
    ORLG_ERR_INVALID = -1,     /* bad argument / unsupported size */
    ORLG_ERR_NO_DEVICE = -2,   /* no HIP device or HIP runtime failure */
    ORLG_ERR_HIP = -3,         /* a HIP call failed (message in orlg_last_error) */
    ORLG_ERR_QUEUE_FULL = -4,  /* an environment's release queue (PhyRMSA: also a channel_state list or the defragmentation
                                * work list) overflowed: the simulation of that environment is no longer the reference's.
                                * Sticky: reported by every orlg_step / orlg_synchronize / orlg_reduce_counters that waits
                                * for the stream after the launch that overflowed, until a full reset or the load of a
                                * checkpoint taken before the overflow.  Raise queue_capacity. */
};

/* Frozen topology: topology.graph["ksp"|"k_paths"|"node_indices"] + edge attr "index"
 * (examples/create_topology.py:96-147, examples/graph_utils.py:89-116, optical_rl_gym/utils.py:27-36). */
typedef struct orlg_topology {
    int32_t num_nodes, num_links, k_paths, num_paths;
    const int32_t *pair_path_base;  /* [N*N] first path record of ordered pair (src,dst); -1 on the diagonal */
    const int32_t *pair_path_count; /* [N*N] must equal k_paths off the diagonal */
    const int32_t *path_hops;       /* [num_paths] Path.hops */
    const int32_t *path_se;         /* [num_paths] Path.best_modulation.spectral_efficiency */
    const double *path_length;      /* [num_paths] Path.length (km) */
    const int32_t *path_link_off;   /* [num_paths+1] CSR offsets into path_links */
    const int32_t *path_links;      /* link "index" of every hop */
} orlg_topology;

/* Constructor kwargs of RMSAEnv / DeepRMSAEnv (optical_rl_gym/envs/rmsa_env.py:29-53,
 * deeprmsa_env.py:10-32) after the host has turned weights into the cumulative tables CPython's
 * random.choices builds (optical_network_env.py:197-206).
 * bit_rate_selection = "continuous" (rmsa_env.py:95-104, 655-659): bit_rate_cum = NULL, bit_rates = the integers lower ..
 * higher (at most 256).  The reference draws the rate with rng.randint, i.e. CPython's _randbelow -- getrandbits(k) repeated
 * until the value is below the width -- which consumes a data-dependent number of MT19937 words per request; the arrival
 * generator then walks the word stream once to find where every request starts before the lanes compute their requests.
 * The bit-rate histograms are kept per integer rate (the reference keeps none in this mode). */
typedef struct orlg_rmsa_config {
    int32_t num_slots;        /* num_spectrum_resources, 1..512 */
    int32_t episode_length;
    int32_t num_bit_rates;    /* 1..64 (continuous: 1..256) */
    int32_t j;                /* DeepRMSA: blocks per path in action / observation (deeprmsa_env.py:34) */
    int32_t reward_mode;      /* 0: 1/0 (optical_network_env.py:213-214); 1: +1/-1 (deeprmsa_env.py:123-124) */
    int32_t queue_capacity;   /* release-queue slots per env (rounded up to a multiple of 16, at least 64); 0 = pick from the
                               * load: mean + 10 sigma of the M/M/inf occupancy */
    int32_t stats_level;      /* ORLG_STATS_* */
    int32_t step_kernel;      /* ORLG_KERNEL_*: which step kernel serves the first-fit policies and external actions */
    double arrival_lambda;    /* 1 / mean_service_inter_arrival_time (rmsa_env.py:646-648) */
    double holding_lambda;    /* 1 / mean_service_holding_time (rmsa_env.py:651) */
    double channel_width;     /* GHz per slot, 12.5 (rmsa_env.py:46,708-719) */
    const int32_t *bit_rates;   /* [num_bit_rates] */
    const double *bit_rate_cum; /* [num_bit_rates]; NULL: bit_rate_selection="continuous" */
    const double *src_cum;      /* [N] */
    const double *dst_cum;      /* [N*N] row s = destination table given source s */
} orlg_rmsa_config;

/* Two step kernels share one state format.  WAVE: one wavefront per environment (every policy).  GROUP: four environments
 * per wavefront, 16 lanes each (every policy too).  AUTO picks GROUP for a batch larger
 * than the WAVE kernel's resident wavefronts (4096 on MI355X), WAVE otherwise; launches of at most four steps (the
 * agent-driven loop: one orlg_step per action) run an instantiation of GROUP that leaves the release queue in HBM.
 * Results are identical bit for bit, and so is the saved state. */
enum { ORLG_KERNEL_AUTO = 0, ORLG_KERNEL_WAVE = 1, ORLG_KERNEL_GROUP = 2 };

enum {
    ORLG_STATS_COUNTERS = 0, /* occupancy, queue, counters, histograms only */
    ORLG_STATS_NETWORK = 1,  /* + network spectrum compactness per step and time-weighted graph
                                throughput / compactness (rmsa_env.py:537-560, 806-851) */
    ORLG_STATS_FULL = 2,     /* + per-link time-weighted utilization / external fragmentation /
                                compactness (rmsa_env.py:562-641): everything RMSAEnv.step computes */
};

/* device-side policies: the reference's heuristic callbacks (rmsa_env.py:854-937, deeprmsa_env.py:135-155) */
enum {
    ORLG_POLICY_EXTERNAL = -1,        /* actions supplied by the caller */
    ORLG_POLICY_SP_FF = 0,            /* shortest_path_first_fit */
    ORLG_POLICY_SAP_FF = 1,           /* shortest_available_path_first_fit */
    ORLG_POLICY_LLP_FF = 2,           /* least_loaded_path_first_fit */
    ORLG_POLICY_DEEPRMSA_SP_FF = 3,   /* deeprmsa_env.shortest_path_first_fit (allow_rejection=False) */
    ORLG_POLICY_DEEPRMSA_SAP_FF = 4,  /* deeprmsa_env.shortest_available_path_first_fit */
    ORLG_POLICY_DEEPRMSA_EXTERNAL = 5, /* caller supplies Discrete(k*j) actions (deeprmsa_env.py:48-58) */
    ORLG_POLICY_PATH_FF_EXTERNAL = 6   /* caller supplies the path, first fit the slot: PathOnlyFirstFitAction
                                          (rmsa_env.py:974-1008); actions is [B] int32 */
};

/* Optional per-step outputs of orlg_step(); any pointer may be NULL.  Arrays are [n_steps][B]
 * (step-major) unless noted; with n_steps == 1 they are the usual per-env vectors. */
typedef struct orlg_step_io {
    int32_t *act_path;   /* action taken, path component (k = rejection) */
    int32_t *act_slot;   /* action taken, initial slot (S = rejection) */
    uint8_t *accepted;   /* current_service.accepted */
    uint8_t *done;       /* episode_services_processed == episode_length (rmsa_env.py:339) */
    double *reward;      /* reward() */
    int32_t *request;    /* [n_steps][B][4] service_id, source_id, destination_id, bit_rate of the request served */
    double *arrival;     /* its arrival_time */
    double *holding;     /* its holding_time */
    double *network_compactness;            /* info["network_compactness"] */
    double *network_compactness_difference; /* info["network_compactness_difference"] */
    double *avg_link_compactness;           /* info["avg_link_compactness"]  (stats level FULL) */
    double *avg_link_utilization;           /* info["avg_link_utilization"]  (stats level FULL) */
} orlg_step_io;

/* counters of one env, rmsa_env.py:84-87 + optical_network_env.py:35-38 */
typedef struct orlg_counters {
    int64_t services_processed, services_accepted;
    int64_t episode_services_processed, episode_services_accepted;
    int64_t bit_rate_requested, bit_rate_provisioned;
    int64_t episode_bit_rate_requested, episode_bit_rate_provisioned;
} orlg_counters;

typedef struct orlg_request {
    int32_t service_id, src, dst, bit_rate;
    double arrival_time, holding_time;
} orlg_request;

typedef struct orlg_env orlg_env;

int orlg_abi_version(void);
const char *orlg_last_error(void);
int orlg_device_count(void);

/* RMSAEnv.__init__ + reset(only_episode_counters=False) (rmsa_env.py:29-220) for B envs;
 * env i is seeded like random.Random(seeds ? seeds[i] : base_seed + i) (optical_network_env.py:266-271). */
int orlg_create(const orlg_topology *topo, const orlg_rmsa_config *cfg, int32_t batch, const uint64_t *seeds,
                uint64_t base_seed, int32_t device, orlg_env **out);
int orlg_destroy(orlg_env *env);
/* use an existing HIP stream (e.g. torch.cuda.current_stream().cuda_stream); NULL = the handle's own */
int orlg_set_stream(orlg_env *env, void *hip_stream);
int orlg_synchronize(orlg_env *env);
/* launch geometry of the step kernel: out[0] environments (waves) per workgroup, out[1] LDS bytes per workgroup,
 * out[2] resident workgroups per CU (occupancy query), out[3] 64-bit words per link */
int orlg_launch_info(orlg_env *env, int32_t *out /* [4] */);

/* name, template arguments and launch shape of the kernel that served the last orlg_step / orlg_reset of this handle, e.g.
 * "orlg_rmsa_group_kernel<5,2> grid=256 block=704 lds=152064" (for benchmarks and profiles: which of the two step kernels
 * AUTO picked) */
int orlg_last_kernel(orlg_env *env, char *buf, int32_t capacity);

/* RMSAEnv.reset(only_episode_counters) (rmsa_env.py:343-457), all envs */
int orlg_reset(orlg_env *env, int32_t only_episode_counters);

/* n_steps x { action = policy(env); env.step(action); optionally env.reset() when done }
 * (utils.py:134-149, rmsa_env.py:222-341).  policy EXTERNAL: actions is [B][2] int32 (path, initial_slot),
 * n_steps must be 1; DEEPRMSA_EXTERNAL: actions is [B] int32.  auto_reset != 0 applies
 * reset(only_episode_counters=True) to every env whose step returned done. */
int orlg_step(orlg_env *env, int32_t policy, int32_t n_steps, const int32_t *actions, int32_t auto_reset,
              const orlg_step_io *io);
/* The launch is asynchronous on the handle's stream.  When host output arrays are given the call waits for them and then
 * also returns ORLG_ERR_QUEUE_FULL if an environment lost a release; otherwise the next orlg_synchronize /
 * orlg_reduce_counters reports it. */

/* state read-back (all arrays [B] or [B][...] env-major) */
int orlg_get_requests(orlg_env *env, orlg_request *out /* [B] */);
int orlg_get_counters(orlg_env *env, orlg_counters *out /* [B] */);
int orlg_get_current_time(orlg_env *env, double *out /* [B] */);
/* topology.graph["available_slots"] as a bitmap: [B][E][W] uint64, bit s of word w = slot 64w+s free */
int orlg_get_occupancy(orlg_env *env, uint64_t *out);
int orlg_words_per_link(orlg_env *env);
/* per-link time-weighted stats [B][E] each (rmsa_env.py:562-641); any pointer may be NULL */
int orlg_get_link_stats(orlg_env *env, double *utilization, double *external_fragmentation, double *compactness,
                        double *last_update);
/* topology.graph["throughput"|"compactness"|"last_update"] [B] each */
int orlg_get_graph_stats(orlg_env *env, double *throughput, double *compactness, double *last_update);
/* bit-rate histograms [B][num_bit_rates] int64 each: requested, provisioned, episode requested, episode provisioned */
int orlg_get_bit_rate_hist(orlg_env *env, int64_t *req, int64_t *prov, int64_t *ereq, int64_t *eprov);
int orlg_get_num_running(orlg_env *env, int32_t *out /* [B] */);
int orlg_get_episodes_done(orlg_env *env, int64_t *out /* [B] */);

/* heuristic-callback queries for ONE env (rmsa_env.py:721-756, 774-804): for each of the k candidate
 * paths of env_index's pending request, the AND of the free-slot bitmaps of its links.
 * masks: [k][W] uint64 (host or device), nslots: [k] int32 = get_number_slots(path). */
int orlg_query_path_masks(orlg_env *env, int32_t env_index, uint64_t *masks, int32_t *nslots);
/* the same for ONE arbitrary path record (any Path object a heuristic passes to is_path_free /
 * get_available_slots): mask [W] uint64, nslots [1] = get_number_slots for the pending bit rate */
int orlg_query_path_mask(orlg_env *env, int32_t env_index, int32_t path_gid, uint64_t *mask, int32_t *nslots);

/* DeepRMSAEnv.observation() (deeprmsa_env.py:60-121) for every env: [B][1 + 2N + (2j+3)k] float64 */
int orlg_deeprmsa_observation(orlg_env *env, double *out);
/* the same vectors as float32 -- every element the float64 value rounded once to nearest, i.e. numpy's
 * obs.astype(float32), which is what a stable-baselines agent does with the reference's Box(float64) observation: half the
 * bytes written and, for a host buffer, copied back */
int orlg_deeprmsa_observation_f32(orlg_env *env, float *out);
/* Both observation entries write `out` in place when it is device memory OR pinned host memory (hipHostMalloc, torch's
 * pin_memory()): the kernel's stores then cross the bus themselves and the call returns without waiting, as for a device
 * buffer (synchronise the stream or an event before reading).  Pageable host memory is staged and copied, and the call waits.
 * The actions of orlg_step may likewise lie in pinned host memory. */
int orlg_deeprmsa_obs_dim(orlg_env *env);

/* SimpleMatrixObservation.observation() (rmsa_env.py:940-971) for every env: [B][2N + E*S] uint8 */
int orlg_simple_matrix_observation(orlg_env *env, uint8_t *out);
int orlg_simple_matrix_obs_dim(orlg_env *env);

/* Every environment gets a fresh random.Random -- seeds[i] if given, else base_seed + i -- and nothing else changes: the
 * pending request stays, the next arrival is the new generator's first draw (arrivals pre-generated from the old generator
 * are dropped).  This is NOT OpticalNetworkEnv.seed (optical_network_env.py:266-271) called after construction: the
 * reference's bit-rate draw stays bound to the generator object of construction time (functools.partial(self.rng.choices,
 * ...), rmsa_env.py:109-111, phy_rmsa_env.py:130-132), so there a request then takes four draws from the new generator and
 * the bit rate from the old one (tests/golden/seed_rmsa_nsfnet_s10.npz pins it; the oracle restates it).  Here all five
 * come from the new generator; the gym views refuse seed() for that reason. */
int orlg_reseed(orlg_env *env, const uint64_t *seeds, uint64_t base_seed);

/* checkpoint / resume (the reference has none for environment state): the complete simulation state
 * of the handle -- occupancy, release queue, RNG, pending requests, counters, statistics -- as one flat blob of
 * orlg_state_size() bytes (host or device buffer).  A blob only fits a handle created with the same arguments.
 * orlg_load_state recomputes the sticky error word from the loaded state: a clean checkpoint clears a reported
 * ORLG_ERR_QUEUE_FULL, the checkpoint of an overflowed batch reports it again. */
int64_t orlg_state_size(orlg_env *env);
int orlg_save_state(orlg_env *env, void *buffer);
int orlg_load_state(orlg_env *env, const void *buffer);

/* sum of all B counter records + accepted/processed, for multi-GPU statistics: out[0..7] = orlg_counters summed,
 * out[8] = total episodes done, out[9] = B.  The caller all-reduces this vector (RCCL, SUM). */
int orlg_reduce_counters(orlg_env *env, int64_t *out /* [16], host or device */);

/* ------------------------------------------------------------------------------------------------
 * QoT-aware environment: PhyRMSAEnv (optical_rl_gym/envs/phy_rmsa_env.py), physical layer + virtual ("grooming")
 * layer + periodic defragmentation.  Allocation is per CHANNEL (L+C+S bands = 268
 * channels, non-contiguous); the QoT gate is the table modulation_level[pair row][channel][k-path] (capacity = level
 * x 100 Gb/s); GSNR[...] feeds the statistics.  A service that does not use the whole capacity of its last channel
 * leaves the rest in channel_state[src, dst, k-path] (phy_rmsa_env.py:600-602), where use_existing_channels (:1650-1673)
 * finds it for later requests of the same (source, destination, k-path): action path = 20 + k-path (:280-288).
 */
struct orlg_gn_gate;
typedef struct orlg_phy_config {
    int32_t num_channels;     /* 2*number_spectrum_channels + number_spectrum_channels_s_band (optical_network_env.py:78-84) */
    int32_t episode_length;
    int32_t num_bit_rates;
    int32_t k_table;          /* k-path columns of the tables (>= topology k_paths) */
    int32_t num_table_rows;
    int32_t queue_capacity;   /* running services per env, multiple of 64; 0 = from the load */
    int32_t grooming;         /* env.grooming (phy_rmsa_env.py:57): bmfa / bmfa_rss consult the virtual layer only when set;
                               * sapff / bmff / sapbm always do (:1256, 1321, 1678) */
    int32_t channel_state_capacity; /* entries per channel_state[src, dst, k-path] list (8..64); 0 = from the load */
    /* periodic defragmentation (phy_rmsa_env.py:54-56, 355-417): every defrag_period processed services, at most
     * number_moves (+1) reallocations ranked by the cut (0) or RSS (1) metric; defrag_period 0 = off */
    int32_t defrag_period, number_moves, defrag_metric;
    int32_t defrag_capacity;  /* entries of the per-env defragmentation work list; 0 = 2 x queue capacity */
    double arrival_lambda, holding_lambda;
    const int32_t *bit_rates;           /* [num_bit_rates], default 100..600 (phy_rmsa_env.py:38) */
    const double *bit_rate_cum, *src_cum, *dst_cum;
    const int32_t *pair_table_row;      /* [N*N] table row of (src, dst) in either order (phy_rmsa_env.py:562-565) */
    const uint8_t *modulation_level;    /* [rows][channels][k_table], every used entry >= 1 */
    const double *gsnr;                 /* [rows][channels][k_table] */
    /* calculate_r_cut(modified=True) (phy_rmsa_env.py:1140-1193): per path the links adjacent to its nodes that are
     * not on the path, weight 1 at the end nodes and 2 at interior nodes (CSR over path records) */
    const int32_t *adj_off, *adj_link, *adj_weight;
    /* Optional, networks of at most 16 nodes of at most 15 links each: the same metric through per-node free degrees.  With
     * D[v][channel] = number of links at node v that are free on the channel, sum_j weight_j * available[link_j] =
     * c . D[:, channel] - (the path's own links, if free) - (chords, if free), c[v] = 1 / 2 / 0 for an end / interior /
     * off-path node; the library keeps D next to the occupancy on chip (it changes by exactly c when a channel is taken or
     * returned on a path; rebuilt from the occupancy at the start of a launch, so it is not part of the saved state) and
     * evaluates the metric with byte dot products.  path_node_weights: [num_paths][32] records -- bytes 0..15 c, 16..17 sum of
     * the adjacency weights (int16), 18..19 c . (path links per node) (int16), 20 number of chords (<= 5), 21..25 chord
     * links, 26..30 chord weights; node_degree: [16] links per node; link_ends: [num_links][2] the two nodes of every link.
     * Any of them NULL = adjacency lists only (identical results, slower). */
    const uint8_t *path_node_weights, *node_degree;
    const int32_t *link_ends;
    const struct orlg_gn_gate *gn_gate; /* GN-model admission check of the chosen channels; NULL = off (the reference) */
} orlg_phy_config;

/* GN-model admission check inside the QoT-aware step (north_star: "GN-model OSNR admission check").  The reference gates by
 * table only (phy_rmsa_env.py:596, 1279, 1341, 1398): this mode has NO reference behaviour -- PARITY UNPINNED; it is pinned
 * to the oracle's restatement (oracle/orlg_oracle_phy.c gn_gsnr_db), which feeds the restatement of
 * examples/calculate_osnr.py:9-56 with the LIVE occupancy.  After the policy (or the caller) has chosen (path, channels) on
 * the physical layer and they are free, every chosen channel is checked: GSNR of a channel of channel_bandwidth_hz at
 * channel_center_frequency_hz[channel] over the path's links -- link l = link_num_spans[l] equal spans of
 * link_span_length_km[l] (examples/create_topology_gn.py:122-125), every lit channel of the link an interferer of the same
 * bandwidth at its own centre frequency whose modulation is the QoT table's level for that channel on the candidate path.
 * level = number of thresholds_db met; if it is below the capacity level the table promised for the channel, the service
 * is blocked (not accepted).  Services served on the virtual layer light nothing new and are not checked. */
typedef struct orlg_gn_gate {
    double launch_power_w, channel_bandwidth_hz;
    double attenuation_normalized;  /* 1/m */
    double noise_figure;            /* linear */
    const double *channel_center_frequency_hz; /* [num_channels] */
    const int32_t *link_num_spans;              /* [num_links] */
    const double *link_span_length_km;          /* [num_links] */
    const double *thresholds_db;                /* [num_thresholds] ascending */
    int32_t num_thresholds, pad;
} orlg_gn_gate;

enum {
    ORLG_PHY_POLICY_EXTERNAL = -1, /* caller supplies (path, channels) per env */
    ORLG_PHY_POLICY_BMFA_CUT = 0,  /* phy_aware_bmfa_rmsa (phy_rmsa_env.py:1375-1438) */
    ORLG_PHY_POLICY_BMFA_RSS_METRIC = 1, /* phy_aware_bmfa_rss_rmsa (phy_rmsa_env.py:1441-1505) */
    ORLG_PHY_POLICY_SAPFF = 2,     /* sapff_rmsa (phy_rmsa_env.py:1676-1737) */
    ORLG_PHY_POLICY_BMFF = 3,      /* phy_aware_bmff_rmsa (phy_rmsa_env.py:1317-1372) */
    ORLG_PHY_POLICY_SAPBM = 4,     /* phy_aware_sapbm_rmsa (phy_rmsa_env.py:1254-1314) */
    ORLG_PHY_POLICY_FAFF = 5,      /* phy_aware_faff_rmsa (phy_rmsa_env.py:1508-1569), cut metric */
    ORLG_PHY_POLICY_FAFF_RSS = 6,  /* phy_aware_faff_rss_rmsa (phy_rmsa_env.py:1572-1647), RSS metric */
};
#define ORLG_PHY_MAX_CHANNELS 14 /* channels per service */

typedef struct orlg_phy_step_io { /* optional per-step outputs, [n_steps][B] each */
    int32_t *act_path;          /* -2 = blocked */
    int32_t *n_channels;
    int16_t *channels;          /* [n_steps][B][ORLG_PHY_MAX_CHANNELS], -1 padded, in allocation order */
    uint8_t *accepted, *done;
    int32_t *request;           /* [n_steps][B][4] service_id, source_id, destination_id, bit_rate */
    double *arrival, *holding;
    double *number_cuts_total;  /* info["number_cuts_total"] (_calculate_total_cuts, phy_rmsa_env.py:1195-1203) */
    double *rss_total_metric;   /* info["rss_total_metric"] (calculate_total_r_spatial, :1110-1121) */
    int16_t *channels_used;     /* [n_steps][B][ORLG_PHY_MAX_CHANNELS] share of each channel the service uses, 100 Gb/s units */
    int32_t *defrag_counters;   /* [n_steps][B][3] counted_moves, counted_moves_groom, counted_defrag_cycles as the step's info
                                 * dict sees them (phy_rmsa_env.py:340-342: before the step's own defragmentation) */
    double *gn_gsnr_db;         /* GN gate: GSNR [dB] of the last channel the step checked, NaN when it checked none */
} orlg_phy_step_io;

typedef struct orlg_phy_episode_stats { /* per-episode sums behind the info dict (phy_rmsa_env.py:339-347) */
    double total_path_length, total_gsnr;
    int64_t total_path_index, total_modulation_level, channels_accepted, physical_services_accepted;
    int64_t episodes_done, queue_overflow;
    int64_t counted_moves, counted_moves_groom, counted_defrag_cycles; /* phy_rmsa_env.py:110-112 */
} orlg_phy_episode_stats;

typedef struct orlg_phy_env orlg_phy_env;

/* PhyRMSAEnv.__init__ + reset(only_episode_counters=False) (phy_rmsa_env.py:30-270, 426-539) for B envs */
int orlg_phy_create(const orlg_topology *topo, const orlg_phy_config *cfg, int32_t batch, const uint64_t *seeds,
                    uint64_t base_seed, int32_t device, orlg_phy_env **out);
int orlg_phy_destroy(orlg_phy_env *env);
int orlg_phy_set_stream(orlg_phy_env *env, void *hip_stream);
int orlg_phy_synchronize(orlg_phy_env *env);
int orlg_phy_reset(orlg_phy_env *env, int32_t only_episode_counters);
/* n_steps x { action = policy(env); env.step(action) } (phy_rmsa_env.py:272-351).  EXTERNAL: n_steps == 1, act_path
 * [B] (-2 = blocked, 0..k-1 = physical path, 20 + k-path = virtual layer) and act_channels [B][ORLG_PHY_MAX_CHANNELS]
 * (-1 padded), entry = channel | used << 9 with `used` the share of the channel the service takes in 100 Gb/s units
 * (the reference's selected_channels tuple fields 0 and 1; used == 0 means the channel's whole capacity). */
int orlg_phy_step(orlg_phy_env *env, int32_t policy, int32_t n_steps, const int32_t *act_path,
                  const int16_t *act_channels, int32_t auto_reset, const orlg_phy_step_io *io);
int orlg_phy_last_kernel(orlg_phy_env *env, char *buf, int32_t capacity); /* as orlg_last_kernel */
int orlg_phy_words_per_link(orlg_phy_env *env);
/* 1 when the handle evaluates the cut metric through the node-degree vectors (orlg_phy_config::path_node_weights), 0 when
 * through the adjacency lists (tables missing, more than 16 nodes or 15 links at a node, no room on chip) */
int orlg_phy_node_vectors(orlg_phy_env *env);
int orlg_phy_get_requests(orlg_phy_env *env, orlg_request *out /* [B] */);
int orlg_phy_get_counters(orlg_phy_env *env, orlg_counters *out /* [B] */);
int orlg_phy_get_current_time(orlg_phy_env *env, double *out /* [B] */);
int orlg_phy_get_num_running(orlg_phy_env *env, int32_t *out /* [B] */);
int orlg_phy_get_episode_stats(orlg_phy_env *env, orlg_phy_episode_stats *out /* [B] */);
/* topology.graph["available_channels"] as a bitmap [B][E][W] uint64 */
int orlg_phy_get_occupancy(orlg_phy_env *env, uint64_t *out);
int orlg_phy_reduce_counters(orlg_phy_env *env, int64_t *out /* [16] as orlg_reduce_counters */);
/* as orlg_reseed */
int orlg_phy_reseed(orlg_phy_env *env, const uint64_t *seeds, uint64_t base_seed);
/* checkpoint / resume, as orlg_save_state */
int64_t orlg_phy_state_size(orlg_phy_env *env);
int orlg_phy_save_state(orlg_phy_env *env, void *buffer);
int orlg_phy_load_state(orlg_phy_env *env, const void *buffer);
/* env.channel_state[src_id, dst_id, k-path] of ONE env (phy_rmsa_env.py:117-125): entries [N*N*K][capacity] packed
 * channel | used << 9 | free << 14 | capacity << 19 (100 Gb/s units) in list order, lengths [N*N*K]; returns capacity */
int orlg_phy_get_channel_state(orlg_phy_env *env, int32_t env_index, uint32_t *entries, uint8_t *lengths);
int orlg_phy_channel_state_capacity(orlg_phy_env *env);

/* ------------------------------------------------------------------------------------------------
 * GN-model GSNR admission check: calculate_osnr (examples/calculate_osnr.py:9-56) for a flattened batch of checks.
 * Check m walks links check_link_off[m]..check_link_off[m+1]; link l has spans link_span_off[l].. and the services
 * running on it link_svc_off[l].. in list order (svc_is_self marks the entry that is the current service itself: the
 * reference adds a stale phi for it, calculate_osnr.py:31-46).  Units: Hz, W, km, 1/m (attenuation_normalized),
 * linear noise figure.  Result: GSNR in dB.  Arrays may be host or device pointers.  Orphaned in the reference
 * (no caller / test): parity is pinned only by tests/golden/osnr_grid.npz.
 */
typedef struct orlg_osnr_batch {
    int32_t num_checks, num_links, num_spans, num_services;
    const int32_t *check_link_off, *link_span_off, *link_svc_off;
    const double *bandwidth, *center_frequency, *launch_power;            /* [num_checks] */
    const double *span_length_km, *span_attenuation, *span_noise_figure;  /* [num_spans] */
    const double *svc_bandwidth, *svc_center_frequency;                   /* [num_services] */
    const int32_t *svc_se;                                                /* [num_services] 1..6 */
    const uint8_t *svc_is_self;                                           /* [num_services] */
} orlg_osnr_batch;
int orlg_gn_osnr(const orlg_osnr_batch *batch, double *gsnr_db /* [num_checks] */, int32_t device, void *hip_stream);

/* host build of the device's natural-log routine (bit-identical algorithm; see csrc/orlg_math.h) */
double orlg_host_log(double x);

#ifdef __cplusplus
}
#endif
#endif
