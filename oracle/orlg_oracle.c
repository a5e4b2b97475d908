/*
 * oracle/orlg_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see orlg_oracle.h).
 *
 * CPU restatement of the reference's RMSA / DeepRMSA step() path.  All file:line citations are
 * relative to /root/reference.  Build with -ffp-contract=off so that every double operation is
 * the single IEEE operation Python performs.
 *
 * Third-party algorithm restated here because the reference takes it from its interpreter and
 * not from its own tree: CPython 3.10 `random.Random` = MT19937 (Matsumoto & Nishimura 1998,
 * init_genrand / init_by_array / genrand_int32) with CPython's 53-bit `random()`,
 * `expovariate` and `choices` (Lib/random.py, Modules/_randommodule.c).  It is pinned by the
 * known answer Random(10).random() -> 0.5714025946899135, 0.4288890546751146,
 * 0.5780913011344704 and by every golden trace's request stream.
 */
#include "orlg_oracle.h"

#include "orlg_oracle_common.h"

double (*orc_g_log)(double) = log;
void orc_set_log_fn(double (*fn)(double)) { orc_g_log = fn ? fn : log; }

void orc_py_random_stream(uint64_t seed, int n, double *out) {
    py_rng r;
    py_seed(&r, seed);
    for (int i = 0; i < n; i++) out[i] = py_random(&r);
}

/* ------------------------------------------------------------------ env state */
typedef struct service {
    int32_t service_id, src, dst, bit_rate, br_index;
    double arrival_time, holding_time;
    int32_t path_gid, initial_slot, number_slots, hops;
    int32_t accepted;
    int64_t seq; /* insertion order, only to make heap ties deterministic (reference would raise) */
} service;

typedef struct { double time; service *svc; } event;

struct orc_env {
    orc_topology topo;
    orc_config cfg;
    int N, E, S, K;
    py_rng rng;
    py_rng rng_br;   /* after seed(): the generator the bit-rate partial stays bound to (see orc_seed) */
    int split;
    uint8_t *available;  /* topology.graph["available_slots"], E*S, 1 = free */
    /* per link (rmsa_env.py:562-641, optical_network_env.py:260-264) */
    double *l_util, *l_extfrag, *l_compact, *l_last_update;
    /* graph level (rmsa_env.py:537-560) */
    double g_throughput, g_compactness, g_last_update;
    double current_time;
    orc_counters c;
    int64_t h_req[ORC_MAX_BIT_RATES], h_prov[ORC_MAX_BIT_RATES], h_ereq[ORC_MAX_BIT_RATES],
        h_eprov[ORC_MAX_BIT_RATES];
    service *current;
    int new_service;
    /* topology.graph["running_services"] */
    service **running;
    int n_running, cap_running;
    /* self._events (heapq) */
    event *heap;
    int n_heap, cap_heap;
    int64_t seq;
    int last_path, last_slot; /* the [path, initial_slot] RMSAEnv.step last received */
    /* scratch for rle */
    int *r_start, *r_len;
    uint8_t *r_val;
    uint8_t *tmp_slots;
};

static int path_gid(const orc_env *e, int src, int dst, int idp) {
    return e->topo.pair_path_base[src * e->N + dst] + idp;
}
static int pair_count(const orc_env *e, int src, int dst) { return e->topo.pair_path_count[src * e->N + dst]; }

/* rmsa_env.py:758-772 rle(): returns run count; starts / values / lengths per run */
static int rle(const uint8_t *a, int n, int *starts, uint8_t *values, int *lengths) {
    if (n == 0) return 0;
    int runs = 0, s = 0;
    for (int i = 1; i <= n; i++) {
        if (i == n || a[i] != a[i - 1]) {
            starts[runs] = s; values[runs] = a[s]; lengths[runs] = i - s;
            runs++; s = i;
        }
    }
    return runs;
}

/* rmsa_env.py:708-719 */
static int number_slots_gid(const orc_env *e, int gid) {
    double q = (double)e->current->bit_rate / ((double)e->topo.path_se[gid] * e->cfg.channel_width);
    return (int)ceil(q) + 1;
}
int orc_get_number_slots(const orc_env *e, int idp) {
    return number_slots_gid(e, path_gid(e, e->current->src, e->current->dst, idp));
}

/* rmsa_env.py:721-734 */
static int is_path_free_gid(const orc_env *e, int gid, int initial_slot, int number_slots) {
    if (initial_slot + number_slots > e->S) return 0;
    for (int h = e->topo.path_link_off[gid]; h < e->topo.path_link_off[gid + 1]; h++) {
        const uint8_t *row = e->available + (size_t)e->topo.path_links[h] * e->S;
        for (int s = initial_slot; s < initial_slot + number_slots; s++)
            if (row[s] == 0) return 0;
    }
    return 1;
}
int orc_is_path_free(const orc_env *e, int idp, int initial_slot, int number_slots) {
    return is_path_free_gid(e, path_gid(e, e->current->src, e->current->dst, idp), initial_slot, number_slots);
}

/* rmsa_env.py:745-756 (elementwise product over the path's links; "id" == "index" for txt topologies) */
static void available_slots_gid(const orc_env *e, int gid, uint8_t *out) {
    memset(out, 1, e->S);
    for (int h = e->topo.path_link_off[gid]; h < e->topo.path_link_off[gid + 1]; h++) {
        const uint8_t *row = e->available + (size_t)e->topo.path_links[h] * e->S;
        for (int s = 0; s < e->S; s++) out[s] &= row[s];
    }
}

/* rmsa_env.py:806-851 */
static double network_compactness(orc_env *e) {
    int64_t sum_slots_paths = 0, sum_occupied = 0, sum_unused_blocks = 0;
    for (int i = 0; i < e->n_running; i++)
        sum_slots_paths += (int64_t)e->running[i]->number_slots * e->running[i]->hops;
    for (int l = 0; l < e->E; l++) {
        const uint8_t *row = e->available + (size_t)l * e->S;
        int runs = rle(row, e->S, e->r_start, e->r_val, e->r_len);
        int first_used = -1, last_used = -1, n_used = 0;
        for (int i = 0; i < runs; i++)
            if (e->r_val[i] == 0) { if (first_used < 0) first_used = i; last_used = i; n_used++; }
        if (n_used > 1) {
            int lambda_min = e->r_start[first_used];
            int lambda_max = e->r_start[last_used] + e->r_len[last_used];
            sum_occupied += lambda_max - lambda_min;
            /* second rle over [lambda_min, lambda_max): sum of run values = number of free runs inside */
            int iruns = rle(row + lambda_min, lambda_max - lambda_min, e->r_start, e->r_val, e->r_len);
            for (int i = 0; i < iruns; i++) sum_unused_blocks += e->r_val[i];
        }
    }
    if (sum_unused_blocks > 0)
        return ((double)sum_occupied / (double)sum_slots_paths) * ((double)e->E / (double)sum_unused_blocks);
    return 1.0;
}

/* rmsa_env.py:562-641 */
static void update_link_stats(orc_env *e, int l) {
    double last_update = e->l_last_update[l];
    double time_diff = e->current_time - e->l_last_update[l];
    if (e->current_time > 0) {
        const uint8_t *row = e->available + (size_t)l * e->S;
        int64_t sum_free = 0;
        for (int s = 0; s < e->S; s++) sum_free += row[s];
        double last_util = e->l_util[l];
        double cur_util = (double)(e->S - sum_free) / (double)e->S;
        e->l_util[l] = ((last_util * last_update) + (cur_util * time_diff)) / e->current_time;

        double last_ef = e->l_extfrag[l], last_c = e->l_compact[l];
        double cur_ef = 0.0, cur_c = 0.0;
        if (sum_free > 0) {
            int runs = rle(row, e->S, e->r_start, e->r_val, e->r_len);
            /* external fragmentation (:596-602) */
            int n_unused = 0, first_unused = -1, last_unused = -1, max_empty = 0;
            for (int i = 0; i < runs; i++)
                if (e->r_val[i] == 1) { if (first_unused < 0) first_unused = i; last_unused = i; n_unused++; }
            /* len(unused_blocks) > 1 and unused_blocks != [0, len(values) - 1] */
            if (n_unused > 1 && !(n_unused == 2 && first_unused == 0 && last_unused == runs - 1)) {
                for (int i = 0; i < runs; i++)
                    if (e->r_val[i] == 1 && e->r_len[i] > max_empty) max_empty = e->r_len[i];
            }
            cur_ef = 1.0 - ((double)max_empty / (double)sum_free);
            /* link spectrum compactness (:605-626) */
            int n_used = 0, first_used = -1, last_used = -1;
            for (int i = 0; i < runs; i++)
                if (e->r_val[i] == 0) { if (first_used < 0) first_used = i; last_used = i; n_used++; }
            if (n_used > 1) {
                int lambda_min = e->r_start[first_used];
                int lambda_max = e->r_start[last_used] + e->r_len[last_used];
                int iruns = rle(row + lambda_min, lambda_max - lambda_min, e->r_start, e->r_val, e->r_len);
                int64_t unused_spectrum_slots = 0; /* np.sum(1 - internal_values): number of USED runs inside */
                for (int i = 0; i < iruns; i++) unused_spectrum_slots += 1 - e->r_val[i];
                if (unused_spectrum_slots > 0) {
                    int64_t sum_used = e->S - sum_free; /* np.sum(1 - slot_allocation) */
                    cur_c = ((double)(lambda_max - lambda_min) / (double)sum_used) *
                            (1.0 / (double)unused_spectrum_slots);
                } else {
                    cur_c = 1.0;
                }
            } else {
                cur_c = 1.0;
            }
        }
        e->l_extfrag[l] = ((last_ef * last_update) + (cur_ef * time_diff)) / e->current_time;
        e->l_compact[l] = ((last_c * last_update) + (cur_c * time_diff)) / e->current_time;
    }
    e->l_last_update[l] = e->current_time;
}

/* rmsa_env.py:537-560 */
static void update_network_stats(orc_env *e) {
    double last_update = e->g_last_update;
    double time_diff = e->current_time - last_update;
    if (e->current_time > 0) {
        double cur_throughput = 0.0;
        for (int i = 0; i < e->n_running; i++) cur_throughput += (double)e->running[i]->bit_rate;
        e->g_throughput = ((e->g_throughput * last_update) + (cur_throughput * time_diff)) / e->current_time;
        e->g_compactness =
            ((e->g_compactness * last_update) + (network_compactness(e) * time_diff)) / e->current_time;
    }
    e->g_last_update = e->current_time;
}

/* heapq on (time, service) */
static int ev_less(const event *a, const event *b) {
    if (a->time != b->time) return a->time < b->time;
    return a->svc->seq < b->svc->seq;
}
static void heap_push(orc_env *e, double t, service *s) {
    if (e->n_heap == e->cap_heap) {
        e->cap_heap = e->cap_heap ? 2 * e->cap_heap : 256;
        e->heap = (event *)realloc(e->heap, sizeof(event) * e->cap_heap);
    }
    int i = e->n_heap++;
    e->heap[i].time = t; e->heap[i].svc = s;
    while (i > 0) {
        int p = (i - 1) / 2;
        if (!ev_less(&e->heap[i], &e->heap[p])) break;
        event tmp = e->heap[i]; e->heap[i] = e->heap[p]; e->heap[p] = tmp;
        i = p;
    }
}
static event heap_pop(orc_env *e) {
    event top = e->heap[0];
    e->heap[0] = e->heap[--e->n_heap];
    int i = 0;
    for (;;) {
        int l = 2 * i + 1, r = l + 1, m = i;
        if (l < e->n_heap && ev_less(&e->heap[l], &e->heap[m])) m = l;
        if (r < e->n_heap && ev_less(&e->heap[r], &e->heap[m])) m = r;
        if (m == i) break;
        event tmp = e->heap[i]; e->heap[i] = e->heap[m]; e->heap[m] = tmp;
        i = m;
    }
    return top;
}

static void running_add(orc_env *e, service *s) {
    if (e->n_running == e->cap_running) {
        e->cap_running = e->cap_running ? 2 * e->cap_running : 256;
        e->running = (service **)realloc(e->running, sizeof(service *) * e->cap_running);
    }
    e->running[e->n_running++] = s;
}
static void running_remove(orc_env *e, service *s) {
    for (int i = 0; i < e->n_running; i++)
        if (e->running[i] == s) {
            memmove(&e->running[i], &e->running[i + 1], sizeof(service *) * (e->n_running - i - 1));
            e->n_running--;
            return;
        }
}

/* rmsa_env.py:462-513 */
static void provision_path(orc_env *e, int gid, int initial_slot, int number_slots) {
    service *s = e->current;
    s->path_gid = gid; s->initial_slot = initial_slot; s->number_slots = number_slots;
    s->hops = e->topo.path_hops[gid];
    for (int h = e->topo.path_link_off[gid]; h < e->topo.path_link_off[gid + 1]; h++) {
        int l = e->topo.path_links[h];
        memset(e->available + (size_t)l * e->S + initial_slot, 0, number_slots);
        update_link_stats(e, l);
    }
    running_add(e, s);
    update_network_stats(e);
    e->c.services_accepted++;
    e->c.episode_services_accepted++;
    e->c.bit_rate_provisioned += s->bit_rate;
    e->c.episode_bit_rate_provisioned += s->bit_rate;
    e->h_prov[s->br_index]++;
    e->h_eprov[s->br_index]++;
}

/* rmsa_env.py:515-535 */
static void release_path(orc_env *e, service *s) {
    int gid = s->path_gid;
    for (int h = e->topo.path_link_off[gid]; h < e->topo.path_link_off[gid + 1]; h++) {
        int l = e->topo.path_links[h];
        memset(e->available + (size_t)l * e->S + s->initial_slot, 1, s->number_slots);
        update_link_stats(e, l);
    }
    running_remove(e, s);
}

/* rmsa_env.py:643-695 + optical_network_env.py:191-208 */
static void next_service(orc_env *e) {
    if (e->new_service) return;
    double at = e->current_time + py_expovariate(&e->rng, e->cfg.arrival_lambda);
    e->current_time = at;
    double ht = py_expovariate(&e->rng, e->cfg.holding_lambda);
    int src = py_choice_cum(&e->rng, e->cfg.src_cum, e->N);
    int dst = py_choice_cum(&e->rng, e->cfg.dst_cum + (size_t)src * e->N, e->N);
    /* rmsa_env.py:655-659: continuous -> rng.randint(lower, higher) (the partial is bound like the discrete one: orc_seed) */
    int bri = e->cfg.bit_rate_cum ? py_choice_cum(e->split ? &e->rng_br : &e->rng, e->cfg.bit_rate_cum, e->cfg.num_bit_rates)
                                  : (int)py_randbelow(e->split ? &e->rng_br : &e->rng, (uint32_t)e->cfg.num_bit_rates);

    service *s = (service *)calloc(1, sizeof(service));
    s->service_id = (int32_t)e->c.episode_services_processed;
    s->src = src; s->dst = dst; s->br_index = bri; s->bit_rate = e->cfg.bit_rates[bri];
    s->arrival_time = at; s->holding_time = ht;
    s->seq = e->seq++;
    /* a rejected service is only referenced from topology.graph["services"]; free the old one here */
    if (e->current && !e->current->accepted) free(e->current);
    e->current = s;
    e->new_service = 1;

    e->c.services_processed++;
    e->c.episode_services_processed++;
    e->c.bit_rate_requested += s->bit_rate;
    e->c.episode_bit_rate_requested += s->bit_rate;
    e->h_req[bri]++;
    e->h_ereq[bri]++;

    while (e->n_heap > 0) {
        event ev = heap_pop(e);
        if (ev.time <= e->current_time) {
            release_path(e, ev.svc);
            free(ev.svc);
        } else {
            heap_push(e, ev.time, ev.svc);
            break;
        }
    }
}

static void full_reset(orc_env *e) {
    /* optical_network_env.py:216-264 + rmsa_env.py:391-457 */
    for (int i = 0; i < e->n_heap; i++) free(e->heap[i].svc);
    e->n_heap = 0;
    e->n_running = 0;
    if (e->current && !e->current->accepted) free(e->current);
    e->current = NULL;
    e->current_time = 0;
    memset(&e->c, 0, sizeof(e->c));
    memset(e->available, 1, (size_t)e->E * e->S);
    memset(e->h_req, 0, sizeof(e->h_req));
    memset(e->h_prov, 0, sizeof(e->h_prov));
    for (int l = 0; l < e->E; l++)
        e->l_util[l] = e->l_extfrag[l] = e->l_compact[l] = e->l_last_update[l] = 0.0;
    e->g_throughput = e->g_compactness = e->g_last_update = 0.0;
    e->new_service = 0;
    next_service(e);
}

/* rmsa_env.py:343-457 */
void orc_reset(orc_env *e, int only_episode_counters) {
    e->c.episode_bit_rate_requested = 0;
    e->c.episode_bit_rate_provisioned = 0;
    e->c.episode_services_processed = 0;
    e->c.episode_services_accepted = 0;
    memset(e->h_ereq, 0, sizeof(e->h_ereq));
    memset(e->h_eprov, 0, sizeof(e->h_eprov));
    if (only_episode_counters) {
        if (e->new_service) {
            e->c.episode_services_processed += 1;
            e->c.episode_bit_rate_requested += e->current->bit_rate;
            e->h_ereq[e->current->br_index] += 1;
        }
        return;
    }
    full_reset(e);
}

/* OpticalNetworkEnv.seed (optical_network_env.py:266-271): self.rng = random.Random(seed), nothing else changes.  A quirk
 * comes with it: the constructors bind the bit-rate draw to the generator OBJECT of that moment -- functools.partial(
 * self.rng.choices, ...) (rmsa_env.py:109-111, phy_rmsa_env.py:130-132) -- so after a later seed() inter-arrival time, holding
 * time, source and destination come from the new generator and the bit rate still from the old one, which carries on where
 * it was. */
void orc_seed(orc_env *e, uint64_t seed) {
    if (!e->split) { e->rng_br = e->rng; e->split = 1; }
    py_seed(&e->rng, seed);
}
/* NOT the reference: a fresh generator for all five draws of a request (what the device's orlg_reseed does) */
void orc_reseed(orc_env *e, uint64_t seed) { py_seed(&e->rng, seed); e->split = 0; }

orc_env *orc_create(const orc_topology *topo, const orc_config *cfg, uint64_t seed) {
    orc_env *e = (orc_env *)calloc(1, sizeof(orc_env));
    e->topo = *topo; e->cfg = *cfg;
    e->N = topo->num_nodes; e->E = topo->num_links; e->S = cfg->num_slots; e->K = topo->k_paths;
    py_seed(&e->rng, seed);
    e->available = (uint8_t *)malloc((size_t)e->E * e->S);
    e->l_util = (double *)calloc(e->E, sizeof(double));
    e->l_extfrag = (double *)calloc(e->E, sizeof(double));
    e->l_compact = (double *)calloc(e->E, sizeof(double));
    e->l_last_update = (double *)calloc(e->E, sizeof(double));
    e->r_start = (int *)malloc(sizeof(int) * (e->S + 1));
    e->r_len = (int *)malloc(sizeof(int) * (e->S + 1));
    e->r_val = (uint8_t *)malloc(e->S + 1);
    e->tmp_slots = (uint8_t *)malloc(e->S);
    /* rmsa_env.py:219-220: the constructor does reset(only_episode_counters=False) */
    orc_reset(e, 0);
    return e;
}

void orc_destroy(orc_env *e) {
    if (!e) return;
    for (int i = 0; i < e->n_heap; i++) free(e->heap[i].svc);
    if (e->current && !e->current->accepted) free(e->current);
    free(e->heap); free(e->running); free(e->available);
    free(e->l_util); free(e->l_extfrag); free(e->l_compact); free(e->l_last_update);
    free(e->r_start); free(e->r_len); free(e->r_val); free(e->tmp_slots);
    free(e);
}

void orc_get_request(const orc_env *e, orc_request *o) {
    o->service_id = e->current->service_id; o->src = e->current->src; o->dst = e->current->dst;
    o->bit_rate = e->current->bit_rate; o->arrival_time = e->current->arrival_time;
    o->holding_time = e->current->holding_time;
}

/* rmsa_env.py:222-341 */
void orc_step(orc_env *e, int path, int initial_slot, orc_step_result *out) {
    service *s = e->current;
    double previous = network_compactness(e);
    e->last_path = path; e->last_slot = initial_slot;
    s->accepted = 0;
    if (path < e->K && initial_slot < e->S && path >= 0 && initial_slot >= 0) {
        int gid = path_gid(e, s->src, s->dst, path);
        int slots = number_slots_gid(e, gid);
        if (is_path_free_gid(e, gid, initial_slot, slots)) {
            provision_path(e, gid, initial_slot, slots);
            s->accepted = 1;
            heap_push(e, s->arrival_time + s->holding_time, s); /* optical_network_env.py:178-189 */
        }
    }
    int nb = e->cfg.num_bit_rates;
    double bmax = 0, bmin = 0;
    for (int b = 0; b < nb; b++) {
        double v = 0.0;
        if (e->h_req[b] > 0) v = (double)(e->h_req[b] - e->h_prov[b]) / (double)e->h_req[b];
        if (out) out->bit_rate_blocking[b] = v;
        if (b == 0 || v > bmax) bmax = v;
        if (b == 0 || v < bmin) bmin = v;
    }
    double cur = network_compactness(e);
    if (out) {
        out->reward = e->cfg.reward_mode == 1 ? (s->accepted ? 1.0 : -1.0) : (s->accepted ? 1.0 : 0.0);
        out->accepted = s->accepted;
        out->service_blocking_rate =
            (double)(e->c.services_processed - e->c.services_accepted) / (double)e->c.services_processed;
        out->episode_service_blocking_rate =
            (double)(e->c.episode_services_processed - e->c.episode_services_accepted) /
            (double)e->c.episode_services_processed;
        out->bit_rate_blocking_rate =
            (double)(e->c.bit_rate_requested - e->c.bit_rate_provisioned) / (double)e->c.bit_rate_requested;
        out->episode_bit_rate_blocking_rate =
            (double)(e->c.episode_bit_rate_requested - e->c.episode_bit_rate_provisioned) /
            (double)e->c.episode_bit_rate_requested;
        out->network_compactness = cur;
        out->network_compactness_difference = previous - cur;
        out->avg_link_compactness = np_mean(e->l_compact, e->E);
        out->avg_link_utilization = np_mean(e->l_util, e->E);
        out->fairness = bmax - bmin;
    }
    e->new_service = 0;
    next_service(e);
    if (out) out->done = (e->c.episode_services_processed == e->cfg.episode_length);
}

/* rmsa_env.py:774-804 : first j free runs of the path-wide AND with length >= slots */
static int available_blocks_gid(orc_env *e, int gid, int *starts, int *lengths) {
    available_slots_gid(e, gid, e->tmp_slots);
    int slots = number_slots_gid(e, gid);
    int runs = rle(e->tmp_slots, e->S, e->r_start, e->r_val, e->r_len);
    int n = 0;
    for (int i = 0; i < runs && n < e->cfg.j; i++)
        if (e->r_val[i] == 1 && e->r_len[i] >= slots) { starts[n] = e->r_start[i]; lengths[n] = e->r_len[i]; n++; }
    return n;
}
int orc_get_available_blocks(orc_env *e, int idp, int *starts, int *lengths) {
    return available_blocks_gid(e, path_gid(e, e->current->src, e->current->dst, idp), starts, lengths);
}

/* deeprmsa_env.py:48-58 */
void orc_step_deeprmsa(orc_env *e, int action, orc_step_result *out) {
    int j = e->cfg.j;
    if (action >= 0 && action < e->K * j) {
        int route = action / j, block = action % j;
        int starts[64], lengths[64];
        int n = orc_get_available_blocks(e, route, starts, lengths);
        if (block < n) { orc_step(e, route, starts[block], out); return; }
    }
    orc_step(e, e->K, e->S, out);
}

/* heuristics: rmsa_env.py:854-871 (SP-FF), :901-913 (SAP-FF), :916-937 (LLP-FF);
 * deeprmsa_env.py:135-143 (SP-FF), :146-155 (SAP-FF) */
void orc_policy(orc_env *e, int policy, int *path, int *slot) {
    const service *s = e->current;
    int np = pair_count(e, s->src, s->dst);
    *path = e->K; *slot = e->S;
    if (policy == ORC_POLICY_SP_FF) {
        int gid = path_gid(e, s->src, s->dst, 0);
        int n = number_slots_gid(e, gid);
        for (int i = 0; i < e->S - n; i++) /* NOTE exclusive bound, rmsa_env.py:860-862 */
            if (is_path_free_gid(e, gid, i, n)) { *path = 0; *slot = i; return; }
    } else if (policy == ORC_POLICY_SAP_FF) {
        for (int idp = 0; idp < np; idp++) {
            int gid = path_gid(e, s->src, s->dst, idp);
            int n = number_slots_gid(e, gid);
            for (int i = 0; i < e->S - n; i++)
                if (is_path_free_gid(e, gid, i, n)) { *path = idp; *slot = i; return; }
        }
    } else if (policy == ORC_POLICY_LLP_FF) {
        int64_t max_free = 0;
        for (int idp = 0; idp < np; idp++) {
            int gid = path_gid(e, s->src, s->dst, idp);
            int n = number_slots_gid(e, gid);
            for (int i = 0; i < e->S - n; i++)
                if (is_path_free_gid(e, gid, i, n)) {
                    available_slots_gid(e, gid, e->tmp_slots);
                    int64_t fs = 0;
                    for (int q = 0; q < e->S; q++) fs += e->tmp_slots[q];
                    if (fs > max_free) { *path = idp; *slot = i; max_free = fs; }
                    break;
                }
        }
    } else if (policy == ORC_POLICY_DEEPRMSA_SP_FF) {
        /* allow_rejection is False in every shipped config -> action 0 (deeprmsa_env.py:136-137) */
        *path = 0; *slot = 0;
    } else if (policy == ORC_POLICY_DEEPRMSA_SAP_FF) {
        int starts[64], lengths[64];
        *path = e->K * e->cfg.j; *slot = 0;
        for (int idp = 0; idp < np; idp++)
            if (orc_get_available_blocks(e, idp, starts, lengths) > 0) { *path = idp * e->cfg.j; return; }
    }
}

/* deeprmsa_env.py:60-121 */
void orc_deeprmsa_observation(orc_env *e, double *out) {
    const service *s = e->current;
    int N = e->N, K = e->K, j = e->cfg.j, S = e->S;
    int W = 2 * j + 3;
    out[0] = (double)s->bit_rate / 100;
    for (int i = 0; i < 2 * N; i++) out[1 + i] = 0.0;
    int mn = s->src < s->dst ? s->src : s->dst, mx = s->src < s->dst ? s->dst : s->src;
    out[1 + mn] = 1.0;
    out[1 + N + mx] = 1.0;
    double *sp = out + 1 + 2 * N;
    for (int i = 0; i < K * W; i++) sp[i] = -1.0;
    int np = pair_count(e, s->src, s->dst);
    for (int idp = 0; idp < np; idp++) {
        int gid = path_gid(e, s->src, s->dst, idp);
        int num_slots = number_slots_gid(e, gid);
        int starts[64], lengths[64];
        int nb = available_blocks_gid(e, gid, starts, lengths);
        for (int b = 0; b < nb; b++) {
            sp[idp * W + 2 * b] = 2 * ((double)starts[b] - 0.5 * S) / S;
            sp[idp * W + 2 * b + 1] = ((double)lengths[b] - 8) / 8;
        }
        sp[idp * W + 2 * j] = (num_slots - 5.5) / 3.5;
        available_slots_gid(e, gid, e->tmp_slots);
        int runs = rle(e->tmp_slots, S, e->r_start, e->r_val, e->r_len);
        int64_t total = 0, nfree = 0, sumlen = 0;
        for (int q = 0; q < S; q++) total += e->tmp_slots[q];
        sp[idp * W + 2 * j + 1] = 2 * ((double)total - 0.5 * S) / S;
        for (int i = 0; i < runs; i++)
            if (e->r_val[i] == 1) { nfree++; sumlen += e->r_len[i]; }
        if (nfree > 0) sp[idp * W + 2 * j + 2] = ((double)sumlen / (double)nfree - 4) / 4;
    }
}

void orc_simple_matrix_observation(const orc_env *e, double *out) {
    const service *s = e->current;
    int mn = s->src < s->dst ? s->src : s->dst, mx = s->src < s->dst ? s->dst : s->src;
    for (int i = 0; i < 2 * e->N; i++) out[i] = 0.0;
    out[mn] = 1.0;
    out[e->N + mx] = 1.0;
    for (size_t q = 0; q < (size_t)e->E * e->S; q++) out[2 * e->N + q] = (double)e->available[q];
}

void orc_get_counters(const orc_env *e, orc_counters *out) { *out = e->c; }
double orc_current_time(const orc_env *e) { return e->current_time; }
void orc_get_available_slots(const orc_env *e, uint8_t *out) { memcpy(out, e->available, (size_t)e->E * e->S); }
void orc_get_link_stats(const orc_env *e, double *u, double *f, double *c, double *t) {
    for (int l = 0; l < e->E; l++) {
        if (u) u[l] = e->l_util[l];
        if (f) f[l] = e->l_extfrag[l];
        if (c) c[l] = e->l_compact[l];
        if (t) t[l] = e->l_last_update[l];
    }
}
void orc_get_graph_stats(const orc_env *e, double *thr, double *comp, double *lu) {
    if (thr) *thr = e->g_throughput;
    if (comp) *comp = e->g_compactness;
    if (lu) *lu = e->g_last_update;
}
void orc_get_bit_rate_hist(const orc_env *e, int64_t *req, int64_t *prov, int64_t *ereq, int64_t *eprov) {
    for (int b = 0; b < e->cfg.num_bit_rates; b++) {
        if (req) req[b] = e->h_req[b];
        if (prov) prov[b] = e->h_prov[b];
        if (ereq) ereq[b] = e->h_ereq[b];
        if (eprov) eprov[b] = e->h_eprov[b];
    }
}
int orc_num_running(const orc_env *e) { return e->n_running; }

void orc_run(orc_env *e, int policy, int64_t n_steps, int reset_on_done, const int32_t *actions_in,
             orc_trace *tr) {
    orc_step_result r;
    int obs_dim = 1 + 2 * e->N + (2 * e->cfg.j + 3) * e->K;
    for (int64_t i = 0; i < n_steps; i++) {
        int path, slot;
        const service *s = e->current;
        if (tr) {
            if (tr->service_id) tr->service_id[i] = s->service_id;
            if (tr->src) tr->src[i] = s->src;
            if (tr->dst) tr->dst[i] = s->dst;
            if (tr->bit_rate) tr->bit_rate[i] = s->bit_rate;
            if (tr->arrival) tr->arrival[i] = s->arrival_time;
            if (tr->holding) tr->holding[i] = s->holding_time;
        }
        if (policy == ORC_POLICY_DEEPRMSA_EXTERNAL) { path = actions_in[i]; slot = 0; }
        else if (policy == ORC_POLICY_PATH_FF_EXTERNAL) {
            /* PathOnlyFirstFitAction.action (rmsa_env.py:982-1005) */
            int a = actions_in[i];
            path = e->K; slot = e->S;
            if (a >= 0 && a < e->K) {
                int gid = path_gid(e, s->src, s->dst, a);
                int n = number_slots_gid(e, gid);
                for (int q = 0; q < e->S - n; q++)
                    if (is_path_free_gid(e, gid, q, n)) { path = a; slot = q; break; }
            }
        }
        else if (policy < 0) { path = actions_in[2 * i]; slot = actions_in[2 * i + 1]; }
        else orc_policy(e, policy, &path, &slot);
        if (policy == ORC_POLICY_DEEPRMSA_SP_FF || policy == ORC_POLICY_DEEPRMSA_SAP_FF || policy == ORC_POLICY_DEEPRMSA_EXTERNAL) {
            orc_step_deeprmsa(e, path, &r);
            path = e->last_path; slot = e->last_slot; /* record the RMSA-level action */
        } else {
            orc_step(e, path, slot, &r);
        }
        if (tr) {
            if (tr->act_path) tr->act_path[i] = path;
            if (tr->act_slot) tr->act_slot[i] = slot;
            if (tr->accepted) tr->accepted[i] = (uint8_t)r.accepted;
            if (tr->done) tr->done[i] = (uint8_t)r.done;
            if (tr->reward) tr->reward[i] = r.reward;
            if (tr->services_processed) tr->services_processed[i] = e->c.services_processed;
            if (tr->services_accepted) tr->services_accepted[i] = e->c.services_accepted;
            if (tr->episode_services_processed) tr->episode_services_processed[i] = e->c.episode_services_processed;
            if (tr->episode_services_accepted) tr->episode_services_accepted[i] = e->c.episode_services_accepted;
            if (tr->bit_rate_requested) tr->bit_rate_requested[i] = e->c.bit_rate_requested;
            if (tr->bit_rate_provisioned) tr->bit_rate_provisioned[i] = e->c.bit_rate_provisioned;
            if (tr->episode_bit_rate_requested) tr->episode_bit_rate_requested[i] = e->c.episode_bit_rate_requested;
            if (tr->episode_bit_rate_provisioned) tr->episode_bit_rate_provisioned[i] = e->c.episode_bit_rate_provisioned;
            if (tr->network_compactness) tr->network_compactness[i] = r.network_compactness;
            if (tr->network_compactness_difference) tr->network_compactness_difference[i] = r.network_compactness_difference;
            if (tr->avg_link_compactness) tr->avg_link_compactness[i] = r.avg_link_compactness;
            if (tr->avg_link_utilization) tr->avg_link_utilization[i] = r.avg_link_utilization;
            if (tr->fairness) tr->fairness[i] = r.fairness;
            if (tr->current_time) tr->current_time[i] = e->current_time;
            if (tr->graph_throughput) tr->graph_throughput[i] = e->g_throughput;
            if (tr->graph_compactness) tr->graph_compactness[i] = e->g_compactness;
            if (tr->free_total) {
                int64_t f = 0;
                for (size_t q = 0; q < (size_t)e->E * e->S; q++) f += e->available[q];
                tr->free_total[i] = f;
            }
            if (tr->obs) orc_deeprmsa_observation(e, tr->obs + (size_t)i * obs_dim);
        }
        if (r.done && reset_on_done) orc_reset(e, 1);
    }
}
