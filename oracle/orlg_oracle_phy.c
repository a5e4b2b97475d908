/*
 * oracle/orlg_oracle_phy.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see orlg_oracle.h).
 *
 * CPU restatement of the QoT-aware environment's core, optical_rl_gym/envs/phy_rmsa_env.py (PhyRMSAEnv),
 * in the reference's own representation (one byte per (link, channel), run-length encoding along the
 * link axis for the cut / RSS fragmentation metrics, a binary heap of release events, ordered
 * residual-capacity lists channel_state[src, dst, k-path] for the virtual "grooming" layer).  Heuristics:
 * phy_aware_bmfa_rmsa / phy_aware_bmfa_rss_rmsa (which honour env.grooming) and sapff_rmsa /
 * phy_aware_bmff_rmsa / phy_aware_sapbm_rmsa / phy_aware_faff_rmsa / phy_aware_faff_rss_rmsa (which ALWAYS try
 * use_existing_channels first, whatever env.grooming says: phy_rmsa_env.py:1256,1321,1510,1574,1678).  NOT restated: the periodic defragmentation
 * (defrag_period, phy_rmsa_env.py:355-417, 662-764).
 *
 * Pinned bit for bit (floats included) against tests/golden/phy_*.npz recorded from the reference.
 */
#include "orlg_oracle_phy.h"
#include "orlg_oracle_osnr.h"

#include "orlg_oracle_common.h"

typedef struct pservice {
    int32_t service_id, src, dst, bit_rate, br_index;
    double arrival_time, holding_time;
    int32_t path_gid, idp, nch;
    int32_t ch[ORC_PHY_MAX_CH], used[ORC_PHY_MAX_CH], cap[ORC_PHY_MAX_CH];
    int32_t accepted, virtual_layer;
    int64_t seq;
} pservice;

typedef struct { double time; pservice *svc; } pevent;

/* one tuple of channel_state[src, dst, idp]: (channel number, used, free, capacity) in 100 Gb/s units */
typedef struct { int ch, used, free_, cap; } cs_entry;
typedef struct { cs_entry *e; int n, capn; } cs_list;

struct orc_phy_env {
    orc_topology topo;
    orc_phy_config cfg;
    int N, E, C, K;
    py_rng rng;
    py_rng rng_br;   /* after seed(): the generator the bit-rate partial stays bound to (see orc_seed) */
    int split;
    uint8_t *avail;     /* topology.graph["available_channels"], E*C, 1 = free */
    int *link_of;       /* [N*N] link index of edge (a,b), -1 if none */
    double current_time;
    orc_counters c;
    /* per-episode statistics of phy_rmsa_env.py:101-112 */
    double total_path_length_episode, total_gsnr_episode;
    int64_t total_path_index_episode, total_modulation_level_episode, channels_accepted_episode,
        physical_services_accepted_episode;
    pservice *current;
    int new_service;
    pservice **running;
    int n_running, cap_running;
    pevent *heap;
    int n_heap, cap_heap;
    int64_t seq;
    int64_t counted_moves, counted_moves_groom, counted_defrag_cycles;  /* phy_rmsa_env.py:110-112 */
    cs_list *cs;        /* [N*N*K] */
    uint8_t *col, *col2;
    int *r_start, *r_len;
    uint8_t *r_val;
};

static int ppath_gid(const orc_phy_env *e, int s, int d, int idp) { return e->topo.pair_path_base[s * e->N + d] + idp; }

static cs_list *cs_of(orc_phy_env *e, int src, int dst, int idp) { return &e->cs[((size_t)src * e->N + dst) * e->K + idp]; }
static void cs_append(cs_list *l, cs_entry v) {
    if (l->n == l->capn) { l->capn = l->capn ? 2 * l->capn : 8; l->e = (cs_entry *)realloc(l->e, sizeof(cs_entry) * l->capn); }
    l->e[l->n++] = v;
}
static int cs_find(const cs_list *l, int ch) {
    for (int i = 0; i < l->n; i++) if (l->e[i].ch == ch) return i;
    return -1;
}
static void cs_remove(cs_list *l, int i) { memmove(&l->e[i], &l->e[i + 1], sizeof(cs_entry) * (l->n - i - 1)); l->n--; }

static int prle(const uint8_t *a, int n, int *starts, uint8_t *values, int *lengths) {
    if (n == 0) return 0;
    int runs = 0, s = 0;
    for (int i = 1; i <= n; i++)
        if (i == n || a[i] != a[i - 1]) { starts[runs] = s; values[runs] = a[s]; lengths[runs] = i - s; runs++; s = i; }
    return runs;
}

static void column(const orc_phy_env *e, int ch, uint8_t *out) {
    for (int l = 0; l < e->E; l++) out[l] = e->avail[(size_t)l * e->C + ch];
}

/* phy_rmsa_env.py:1029-1035 */
static int is_channel_free(const orc_phy_env *e, int gid, int ch) {
    for (int h = e->topo.path_link_off[gid]; h < e->topo.path_link_off[gid + 1]; h++)
        if (e->avail[(size_t)e->topo.path_links[h] * e->C + ch] == 0) return 0;
    return 1;
}

/* phy_rmsa_env.py:1123-1193, modified=True, defrag_flag=False: cuts against the links adjacent to the
 * path's nodes before minus after taking the channel on the path's links */
static int r_cut_modified_flag(orc_phy_env *e, int gid, int ch, int defrag_flag) {
    const int32_t *nodes = e->cfg.path_nodes + e->cfg.path_node_off[gid];
    int nn = e->cfg.path_node_off[gid + 1] - e->cfg.path_node_off[gid];
    int before = 0, after = 0;
    for (int pass = 0; pass < 2; pass++) {
        column(e, ch, e->col);
        if (pass == 1)
            for (int h = e->topo.path_link_off[gid]; h < e->topo.path_link_off[gid + 1]; h++)
                e->col[e->topo.path_links[h]] = defrag_flag ? 1 : 0;
        int acc = 0;
        for (int i = 0; i < nn; i++) {
            for (int nk = 0; nk < e->N; nk++) {
                int l = e->link_of[nodes[i] * e->N + nk];
                if (l < 0) continue;
                int on_path = 0;
                for (int q = 0; q < nn; q++) on_path |= (nodes[q] == nk);
                if (on_path) continue;
                if (i == nn - 1) {
                    acc += abs((int)e->col[e->link_of[nodes[i - 1] * e->N + nodes[i]]] - (int)e->col[l]);
                } else if (i == 0) {
                    acc += abs((int)e->col[e->link_of[nodes[0] * e->N + nodes[1]]] - (int)e->col[l]);
                } else {
                    acc += abs((int)e->col[e->link_of[nodes[i] * e->N + nodes[i + 1]]] - (int)e->col[l]) +
                           abs((int)e->col[e->link_of[nodes[i - 1] * e->N + nodes[i]]] - (int)e->col[l]);
                }
            }
        }
        if (pass == 0) before = acc; else after = acc;
    }
    return before - after;
}

static int r_cut_modified(orc_phy_env *e, int gid, int ch) { return r_cut_modified_flag(e, gid, ch, 0); }

static double rss_of_column(orc_phy_env *e, const uint8_t *col) {
    int runs = prle(col, e->E, e->r_start, e->r_val, e->r_len);
    int64_t sq = 0, sm = 0;
    for (int i = 0; i < runs; i++)
        if (e->r_val[i] == 1) { sq += (int64_t)e->r_len[i] * e->r_len[i]; sm += e->r_len[i]; }
    return sqrt((double)sq) / (double)(sm + 1);
}

/* phy_rmsa_env.py:1085-1108 */
static double r_spatial_flag(orc_phy_env *e, int gid, int ch, int defrag_flag) {
    column(e, ch, e->col);
    double r0 = 0 + rss_of_column(e, e->col);
    memcpy(e->col2, e->col, e->E);
    for (int h = e->topo.path_link_off[gid]; h < e->topo.path_link_off[gid + 1]; h++)
        e->col2[e->topo.path_links[h]] = defrag_flag ? 1 : 0;
    double r1 = 0 + rss_of_column(e, e->col2);
    return r1 - r0;
}
static double r_spatial(orc_phy_env *e, int gid, int ch) { return r_spatial_flag(e, gid, ch, 0); }

/* phy_rmsa_env.py:1195-1203 */
static double total_cuts(orc_phy_env *e) {
    int64_t n = 0;
    for (int ch = 0; ch < e->C; ch++) {
        column(e, ch, e->col);
        int runs = prle(e->col, e->E, e->r_start, e->r_val, e->r_len);
        for (int i = 0; i < runs; i++) n += e->r_val[i];
    }
    return (double)n / (double)e->C;
}

/* phy_rmsa_env.py:1110-1121 */
static double total_r_spatial(orc_phy_env *e) {
    double r = 0;
    for (int ch = 0; ch < e->C; ch++) {
        column(e, ch, e->col);
        r += rss_of_column(e, e->col);
    }
    return r / (double)e->C;
}

/* heap */
static int pev_less(const pevent *a, const pevent *b) {
    if (a->time != b->time) return a->time < b->time;
    return a->svc->seq < b->svc->seq;
}
static void pheap_push(orc_phy_env *e, double t, pservice *s) {
    if (e->n_heap == e->cap_heap) {
        e->cap_heap = e->cap_heap ? 2 * e->cap_heap : 1024;
        e->heap = (pevent *)realloc(e->heap, sizeof(pevent) * e->cap_heap);
    }
    int i = e->n_heap++;
    e->heap[i].time = t; e->heap[i].svc = s;
    while (i > 0) {
        int p = (i - 1) / 2;
        if (!pev_less(&e->heap[i], &e->heap[p])) break;
        pevent tmp = e->heap[i]; e->heap[i] = e->heap[p]; e->heap[p] = tmp;
        i = p;
    }
}
static pevent pheap_pop(orc_phy_env *e) {
    pevent top = e->heap[0];
    e->heap[0] = e->heap[--e->n_heap];
    int i = 0;
    for (;;) {
        int l = 2 * i + 1, r = l + 1, m = i;
        if (l < e->n_heap && pev_less(&e->heap[l], &e->heap[m])) m = l;
        if (r < e->n_heap && pev_less(&e->heap[r], &e->heap[m])) m = r;
        if (m == i) break;
        pevent tmp = e->heap[i]; e->heap[i] = e->heap[m]; e->heap[m] = tmp;
        i = m;
    }
    return top;
}

/* phy_rmsa_env.py:781-861 */
static void free_channel_on_path(orc_phy_env *e, int gid, int ch) {
    for (int h = e->topo.path_link_off[gid]; h < e->topo.path_link_off[gid + 1]; h++)
        e->avail[(size_t)e->topo.path_links[h] * e->C + ch] = 1;
}
static void release_path(orc_phy_env *e, pservice *s) {
    for (int i = 0; i < s->nch; i++)
        if (s->used[i] == s->cap[i]) free_channel_on_path(e, s->path_gid, s->ch[i]);
    cs_list *l = cs_of(e, s->src, s->dst, s->idp);
    for (int i = 0; i < s->nch; i++)
        if (s->used[i] != s->cap[i]) {
            int q = cs_find(l, s->ch[i]);
            cs_entry r = l->e[q];
            cs_remove(l, q);
            if (r.used == s->used[i]) {  /* this service was the channel's last user */
                free_channel_on_path(e, s->path_gid, s->ch[i]);
            } else {
                cs_entry u = { r.ch, r.used - s->used[i], r.free_ + s->used[i], r.cap };
                cs_append(l, u);
            }
        }
    for (int i = 0; i < e->n_running; i++)
        if (e->running[i] == s) {
            memmove(&e->running[i], &e->running[i + 1], sizeof(pservice *) * (e->n_running - i - 1));
            e->n_running--;
            break;
        }
}

/* phy_rmsa_env.py:969-1017 */
static void next_service(orc_phy_env *e) {
    if (e->new_service) return;
    double at = e->current_time + py_expovariate(&e->rng, e->cfg.arrival_lambda);
    e->current_time = at;
    double ht = py_expovariate(&e->rng, e->cfg.holding_lambda);
    int src = py_choice_cum(&e->rng, e->cfg.src_cum, e->N);
    int dst = py_choice_cum(&e->rng, e->cfg.dst_cum + (size_t)src * e->N, e->N);
    int bri = py_choice_cum(e->split ? &e->rng_br : &e->rng, e->cfg.bit_rate_cum, e->cfg.num_bit_rates);
    pservice *s = (pservice *)calloc(1, sizeof(pservice));
    s->service_id = (int32_t)e->c.episode_services_processed;
    s->src = src; s->dst = dst; s->br_index = bri; s->bit_rate = e->cfg.bit_rates[bri];
    s->arrival_time = at; s->holding_time = ht; s->seq = e->seq++;
    if (e->current && !e->current->accepted) free(e->current);
    e->current = s;
    e->new_service = 1;
    e->c.services_processed++; e->c.episode_services_processed++;
    e->c.bit_rate_requested += s->bit_rate; e->c.episode_bit_rate_requested += s->bit_rate;
    while (e->n_heap > 0) {
        pevent ev = pheap_pop(e);
        if (ev.time <= e->current_time) {
            release_path(e, ev.svc);
            free(ev.svc);
        } else {
            pheap_push(e, ev.time, ev.svc);
            break;
        }
    }
}

/* phy_rmsa_env.py:426-539 */
void orc_phy_reset(orc_phy_env *e, int only_episode_counters) {
    e->c.episode_bit_rate_requested = 0; e->c.episode_bit_rate_provisioned = 0;
    e->c.episode_services_processed = 0; e->c.episode_services_accepted = 0;
    e->total_path_length_episode = 0; e->total_path_index_episode = 0; e->total_gsnr_episode = 0;
    e->total_modulation_level_episode = 0; e->channels_accepted_episode = 0;
    e->physical_services_accepted_episode = 0;
    e->counted_defrag_cycles = 0; e->counted_moves = 0; e->counted_moves_groom = 0;
    if (only_episode_counters) {
        if (e->new_service) {
            e->c.episode_services_processed += 1;
            e->c.episode_bit_rate_requested += e->current->bit_rate;
        }
        return;
    }
    for (int i = 0; i < e->n_heap; i++) free(e->heap[i].svc);
    e->n_heap = 0; e->n_running = 0;
    if (e->current && !e->current->accepted) free(e->current);
    e->current = NULL;
    e->current_time = 0;
    memset(&e->c, 0, sizeof(e->c));
    memset(e->avail, 1, (size_t)e->E * e->C);
    for (size_t i = 0; i < (size_t)e->N * e->N * e->K; i++) e->cs[i].n = 0;
    e->new_service = 0;
    next_service(e);
}

/* OpticalNetworkEnv.seed (optical_network_env.py:266-271): self.rng = random.Random(seed), nothing else changes.  A quirk
 * comes with it: the constructors bind the bit-rate draw to the generator OBJECT of that moment -- functools.partial(
 * self.rng.choices, ...) (rmsa_env.py:109-111, phy_rmsa_env.py:130-132) -- so after a later seed() inter-arrival time, holding
 * time, source and destination come from the new generator and the bit rate still from the old one, which carries on where
 * it was. */
void orc_phy_seed(orc_phy_env *e, uint64_t seed) {
    if (!e->split) { e->rng_br = e->rng; e->split = 1; }
    py_seed(&e->rng, seed);
}
/* NOT the reference: a fresh generator for all five draws of a request (what the device's orlg_reseed does) */
void orc_phy_reseed(orc_phy_env *e, uint64_t seed) { py_seed(&e->rng, seed); e->split = 0; }

orc_phy_env *orc_phy_create(const orc_topology *topo, const orc_phy_config *cfg, uint64_t seed) {
    orc_phy_env *e = (orc_phy_env *)calloc(1, sizeof(orc_phy_env));
    e->topo = *topo; e->cfg = *cfg;
    e->N = topo->num_nodes; e->E = topo->num_links; e->C = cfg->num_channels; e->K = topo->k_paths;
    py_seed(&e->rng, seed);
    e->avail = (uint8_t *)malloc((size_t)e->E * e->C);
    e->link_of = (int *)malloc(sizeof(int) * e->N * e->N);
    for (int i = 0; i < e->N * e->N; i++) e->link_of[i] = -1;
    for (int l = 0; l < e->E; l++) {
        int a = cfg->link_ends[2 * l], b = cfg->link_ends[2 * l + 1];
        e->link_of[a * e->N + b] = l;
        e->link_of[b * e->N + a] = l;
    }
    e->cs = (cs_list *)calloc((size_t)e->N * e->N * e->K, sizeof(cs_list));
    e->col = (uint8_t *)malloc(e->E); e->col2 = (uint8_t *)malloc(e->E);
    e->r_start = (int *)malloc(sizeof(int) * (e->E + 1));
    e->r_len = (int *)malloc(sizeof(int) * (e->E + 1));
    e->r_val = (uint8_t *)malloc(e->E + 1);
    orc_phy_reset(e, 0);
    return e;
}

void orc_phy_destroy(orc_phy_env *e) {
    if (!e) return;
    for (int i = 0; i < e->n_heap; i++) free(e->heap[i].svc);
    if (e->current && !e->current->accepted) free(e->current);
    for (size_t i = 0; i < (size_t)e->N * e->N * e->K; i++) free(e->cs[i].e);
    free(e->cs);
    free(e->heap); free(e->running); free(e->avail); free(e->link_of); free(e->col); free(e->col2);
    free(e->r_start); free(e->r_len); free(e->r_val);
    free(e);
}

void orc_phy_get_request(const orc_phy_env *e, orc_request *o) {
    o->service_id = e->current->service_id; o->src = e->current->src; o->dst = e->current->dst;
    o->bit_rate = e->current->bit_rate; o->arrival_time = e->current->arrival_time;
    o->holding_time = e->current->holding_time;
}

/* the heuristics of phy_rmsa_env.py:1254-1737.  All build, per candidate path ("row"), the list of free channels
 * (level, [metric,] channel, idp), sort it, pick a row and take its channels in order until the bit rate is covered. */
typedef struct { int mod; double frag; int ch, idp, pos; uint8_t key0; } cand;
static int cmp_level_frag(const void *a, const void *b) { /* key (-x[0], -x[1]); -np.uint8 wraps (key0); stable */
    const cand *x = (const cand *)a, *y = (const cand *)b;
    if (x->key0 != y->key0) return x->key0 < y->key0 ? -1 : 1;
    if (x->frag != y->frag) return x->frag > y->frag ? -1 : 1;
    return x->pos - y->pos;
}
static int cmp_frag(const void *a, const void *b) { /* key (-x[1]); stable (phy_rmsa_env.py:1538, 1601) */
    const cand *x = (const cand *)a, *y = (const cand *)b;
    if (x->frag != y->frag) return x->frag > y->frag ? -1 : 1;
    return x->pos - y->pos;
}
static int cmp_level_ch(const void *a, const void *b) { /* key (-x[0], x[1]) */
    const cand *x = (const cand *)a, *y = (const cand *)b;
    if (x->key0 != y->key0) return x->key0 < y->key0 ? -1 : 1;
    return x->ch - y->ch;
}

static void running_add(orc_phy_env *e, pservice *s);

/* use_existing_channels (phy_rmsa_env.py:1650-1673): residual capacity of partially used channels between the
 * same (source, destination, k-path); returns 1 and fills act (path = idp + 20) when the request fits */
static int use_existing_channels(orc_phy_env *e, orc_phy_action *act) {
    const pservice *s = e->current;
    double unassigned = s->bit_rate;
    int n = 0;
    for (int idp = 0; idp < e->K; idp++) {
        const cs_list *l = cs_of(e, s->src, s->dst, idp);
        int sum = 0;
        for (int i = 0; i < l->n; i++) sum += l->e[i].free_;
        if ((double)sum >= unassigned / 100) {
            for (int i = 0; i < l->n && n < ORC_PHY_MAX_CH; i++) {
                const cs_entry *c = &l->e[i];
                if (c->free_ > 0) {
                    unassigned -= c->free_ * 100;
                    act->ch[n] = c->ch; act->cap[n] = c->cap;
                    if (unassigned <= 0) {
                        act->used[n] = c->free_ + unassigned / 100; act->free_[n] = unassigned / -100;
                        act->path = idp + 20; act->n = n + 1;
                        return 1;
                    }
                    act->used[n] = c->free_; act->free_[n] = 0;
                    n++;
                }
            }
        }
    }
    return 0;
}

void orc_phy_policy(orc_phy_env *e, int policy, orc_phy_action *act) {
    const pservice *s = e->current;
    int K = e->K, C = e->C;
    act->path = -2; act->n = 0;
    /* bmfa / bmfa_rss consult the virtual layer only when env.grooming; the other three always do */
    const int groom = (policy == ORC_PHY_POLICY_BMFA || policy == ORC_PHY_POLICY_BMFA_RSS) ? e->cfg.grooming : 1;
    if (groom && use_existing_channels(e, act)) return;
    act->path = -2; act->n = 0;
    int row = e->cfg.pair_table_row[s->src * e->N + s->dst];
    cand *rows = (cand *)malloc(sizeof(cand) * (size_t)K * C);
    int *cnt = (int *)calloc(K, sizeof(int));
    int *alive = (int *)malloc(sizeof(int) * K);
    const int faff = policy == ORC_PHY_POLICY_FAFF || policy == ORC_PHY_POLICY_FAFF_RSS;
    const int rss = policy == ORC_PHY_POLICY_BMFA_RSS || policy == ORC_PHY_POLICY_FAFF_RSS;
    const int with_metric = policy == ORC_PHY_POLICY_BMFA || policy == ORC_PHY_POLICY_BMFA_RSS || faff;
    for (int idp = 0; idp < K; idp++) {
        int gid = ppath_gid(e, s->src, s->dst, idp);
        alive[idp] = 1;
        for (int ch = 0; ch < C; ch++)
            if (is_channel_free(e, gid, ch)) {
                cand *c = &rows[(size_t)idp * C + cnt[idp]];
                c->mod = e->cfg.modulation_level[((size_t)row * C + ch) * e->cfg.k_table + idp];
                c->frag = !with_metric ? 0.0 : rss ? r_spatial(e, gid, ch) : (double)r_cut_modified(e, gid, ch);
                c->ch = ch; c->idp = idp; c->pos = cnt[idp]; c->key0 = (uint8_t)(-c->mod);
                cnt[idp]++;
            }
        if (faff) qsort(&rows[(size_t)idp * C], cnt[idp], sizeof(cand), cmp_frag);
        else if (with_metric) qsort(&rows[(size_t)idp * C], cnt[idp], sizeof(cand), cmp_level_frag);
        else if (policy != ORC_PHY_POLICY_SAPFF) qsort(&rows[(size_t)idp * C], cnt[idp], sizeof(cand), cmp_level_ch);
        /* sapff: key x[1] = channel, already ascending */
    }
    for (;;) {
        int best = -1;
        if (policy == ORC_PHY_POLICY_SAPFF || policy == ORC_PHY_POLICY_SAPBM) {
            /* empty rows are dropped, the first remaining row is used (phy_rmsa_env.py:1289-1297, 1711-1719) */
            for (int i = 0; i < K && best < 0; i++)
                if (alive[i] && cnt[i] > 0) best = i;
        } else if (faff) {
            /* the row whose head has the best metric, ties keep the lower row (phy_rmsa_env.py:1546-1551) */
            double max_frag = -INFINITY;
            for (int i = 0; i < K; i++) {
                if (!alive[i] || cnt[i] == 0) continue;
                if (rows[(size_t)i * C].frag > max_frag) { max_frag = rows[(size_t)i * C].frag; best = i; }
            }
        } else {
            double max_mod = -INFINITY, max_frag = -INFINITY;
            for (int i = 0; i < K; i++) {
                if (!alive[i] || cnt[i] == 0) continue;
                const cand *h = &rows[(size_t)i * C];
                /* bmff: ties keep the lower row index (:1352-1355); bmfa: ties broken by the metric (:1416-1421) */
                if ((double)h->mod > max_mod || (with_metric && (double)h->mod == max_mod && h->frag > max_frag)) {
                    max_mod = h->mod; max_frag = h->frag; best = i;
                }
            }
        }
        if (best < 0) break;
        double unassigned = s->bit_rate;
        int n = 0, covered = 0;
        for (int q = 0; q < cnt[best] && n < ORC_PHY_MAX_CH; q++) {
            const cand *c = &rows[(size_t)best * C + q];
            unassigned -= c->mod * 100;
            act->ch[n] = c->ch; act->cap[n] = c->mod;
            if (unassigned <= 0) {
                act->used[n] = c->mod + unassigned / 100; act->free_[n] = unassigned / -100;
                n++; covered = 1;
                break;
            }
            act->used[n] = c->mod; act->free_[n] = 0;
            n++;
        }
        if (covered) { act->path = best; act->n = n; break; }
        alive[best] = 0;
    }
    free(rows); free(cnt); free(alive);
}

static void running_add(orc_phy_env *e, pservice *s) {
    if (e->n_running == e->cap_running) {
        e->cap_running = e->cap_running ? 2 * e->cap_running : 1024;
        e->running = (pservice **)realloc(e->running, sizeof(pservice *) * e->cap_running);
    }
    e->running[e->n_running++] = s;
}

static void running_remove(orc_phy_env *e, pservice *s) {
    for (int i = 0; i < e->n_running; i++)
        if (e->running[i] == s) {
            memmove(&e->running[i], &e->running[i + 1], sizeof(pservice *) * (e->n_running - i - 1));
            e->n_running--;
            return;
        }
}
/* service.channels.remove(channel j); service.channels.append((ch, used, ..., cap, ...)) */
static void channels_move(pservice *s, int j, int ch, int used, int cap) {
    for (int q = j; q + 1 < s->nch; q++) { s->ch[q] = s->ch[q + 1]; s->used[q] = s->used[q + 1]; s->cap[q] = s->cap[q + 1]; }
    s->ch[s->nch - 1] = ch; s->used[s->nch - 1] = used; s->cap[s->nch - 1] = cap;
}

/* _groom_defragmentation (phy_rmsa_env.py:703-733) + _move_virtual (:735-764).  The two for loops run over Python
 * lists that the body mutates (remove + append): an index-based walk over the live arrays is exactly the list
 * iterator's behaviour (the element after a moved one is skipped, the moved one is met again at the end). */
static int groom_defragmentation(orc_phy_env *e) {
    int moves = 0;
    for (int i = 0; i < e->n_running; i++) {
        pservice *s = e->running[i];
        for (int j = 0; j < s->nch; j++) {
            if (s->used[j] != s->cap[j]) {
                cs_list *l = cs_of(e, s->src, s->dst, s->idp);
                const cs_entry corr = l->e[cs_find(l, s->ch[j])];
                if (corr.used == s->used[j]) {  /* this service is the channel's only user */
                    for (int t = 0; t < l->n; t++) {
                        const cs_entry tg = l->e[t];
                        if (tg.ch != corr.ch && tg.free_ >= s->used[j]) {
                            const int share = s->used[j];
                            cs_remove(l, t);
                            cs_remove(l, cs_find(l, corr.ch));
                            cs_entry up = { tg.ch, tg.used + share, tg.free_ - share, tg.cap };
                            cs_append(l, up);
                            free_channel_on_path(e, s->path_gid, s->ch[j]);
                            running_remove(e, s);
                            channels_move(s, j, tg.ch, share, tg.cap);
                            running_add(e, s);
                            moves++;
                            break;
                        }
                    }
                }
            }
            if (moves == e->cfg.number_moves) return moves;
        }
    }
    return moves;
}

typedef struct { double diff, age; pservice *svc; int ch, order; } dcand;
static int cmp_dcand(const void *a, const void *b) { /* key (-x[0], -x[1]); stable */
    const dcand *x = (const dcand *)a, *y = (const dcand *)b;
    if (x->diff != y->diff) return x->diff > y->diff ? -1 : 1;
    if (x->age != y->age) return x->age > y->age ? -1 : 1;
    return x->order - y->order;
}

/* the periodic defragmentation block of step (phy_rmsa_env.py:355-417) + _move (:662-697) */
static void periodic_defragmentation(orc_phy_env *e) {
    const int rss = e->cfg.defrag_metric == 1;
    e->counted_moves_groom = groom_defragmentation(e);
    if (e->counted_moves_groom > e->cfg.number_moves) return;
    int total = 0;
    for (int i = 0; i < e->n_running; i++) total += e->running[i]->nch;
    dcand *cands = (dcand *)malloc(sizeof(dcand) * (size_t)(total + 1));
    int nc = 0;
    for (int i = 0; i < e->n_running; i++) {
        pservice *s = e->running[i];
        for (int j = 0; j < s->nch; j++)
            if (s->used[j] == s->cap[j]) {  /* only channels the service fills are reallocated */
                double diff = rss ? r_spatial_flag(e, s->path_gid, s->ch[j], 1) : (double)r_cut_modified_flag(e, s->path_gid, s->ch[j], 1);
                if (diff > 0) {
                    dcand c = { diff, e->current_time - s->arrival_time, s, s->ch[j], nc };
                    cands[nc++] = c;
                }
            }
    }
    qsort(cands, nc, sizeof(dcand), cmp_dcand);
    int num_moves = 0;
    const pservice *cur = e->current;
    for (int q = 0; q < nc; q++) {
        pservice *s = cands[q].svc;
        const int row = e->cfg.pair_table_row[s->src * e->N + s->dst];
        /* the reference looks the candidate's path up among the k paths of the PENDING request (:388-394): the
         * index is right when both serve the same node pair, otherwise the loop runs out and leaves k - 1 */
        int idp = e->K - 1;
        for (int k = 0; k < e->K; k++)
            if (ppath_gid(e, cur->src, cur->dst, k) == s->path_gid) { idp = k; break; }
        const int level = e->cfg.modulation_level[((size_t)row * e->C + cands[q].ch) * e->cfg.k_table + idp];
        int have = 0, best_ch = -1;
        double best_m = 0;
        for (int ch = 0; ch < e->C; ch++)
            if (is_channel_free(e, s->path_gid, ch) &&
                e->cfg.modulation_level[((size_t)row * e->C + ch) * e->cfg.k_table + idp] == level) {
                double m = rss ? r_spatial(e, s->path_gid, ch) : (double)r_cut_modified(e, s->path_gid, ch);
                if (!have || m > best_m) { have = 1; best_m = m; best_ch = ch; }  /* sorted by (-metric, channel) */
            }
        if (have && -1 * best_m < cands[q].diff) {
            /* _move: the channel list entry is found by value (channel numbers are unique within a service) */
            int j = 0;
            while (j < s->nch && s->ch[j] != cands[q].ch) j++;
            for (int h = e->topo.path_link_off[s->path_gid]; h < e->topo.path_link_off[s->path_gid + 1]; h++) {
                e->avail[(size_t)e->topo.path_links[h] * e->C + best_ch] = 0;
                e->avail[(size_t)e->topo.path_links[h] * e->C + cands[q].ch] = 1;
            }
            running_remove(e, s);
            channels_move(s, j, best_ch, s->used[j], s->cap[j]);
            running_add(e, s);
            num_moves++;
            e->counted_moves++;
        }
        if (num_moves + e->counted_moves_groom > e->cfg.number_moves) break;
    }
    if (num_moves != 0) e->counted_defrag_cycles++;
    free(cands);
}

/* GN-model GSNR of channel `ch` on path `gid` against the live occupancy (see orc_phy_config): the flattened batch of
 * orc_gn_osnr -- the restatement of examples/calculate_osnr.py -- is built for ONE check: links = the path's links in path
 * order, per link its equal spans, per link the lit channels other than `ch` in channel order as interferers. */
static double gn_gsnr_db(orc_phy_env *e, int gid, int idp, int row, int ch) {
    const orc_phy_config *g = &e->cfg;
    const int h0 = e->topo.path_link_off[gid], h1 = e->topo.path_link_off[gid + 1], nl = h1 - h0;
    int nspans = 0, nsvc = 0;
    for (int h = h0; h < h1; h++) {
        const int l = e->topo.path_links[h];
        nspans += g->gn_link_num_spans[l];
        for (int c = 0; c < e->C; c++) nsvc += (c != ch && !e->avail[(size_t)l * e->C + c]);
    }
    int32_t *link_span_off = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nl + 1)), *link_svc_off = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nl + 1));
    double *slen = (double *)malloc(sizeof(double) * (size_t)(nspans + 1)), *satt = (double *)malloc(sizeof(double) * (size_t)(nspans + 1)),
           *snf = (double *)malloc(sizeof(double) * (size_t)(nspans + 1));
    double *vbw = (double *)malloc(sizeof(double) * (size_t)(nsvc + 1)), *vcf = (double *)malloc(sizeof(double) * (size_t)(nsvc + 1));
    int32_t *vse = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nsvc + 1));
    uint8_t *vself = (uint8_t *)calloc((size_t)(nsvc + 1), 1);
    int si = 0, vi = 0;
    for (int h = h0; h < h1; h++) {
        const int l = e->topo.path_links[h];
        link_span_off[h - h0] = si; link_svc_off[h - h0] = vi;
        for (int s = 0; s < g->gn_link_num_spans[l]; s++) {
            slen[si] = g->gn_link_span_length_km[l]; satt[si] = g->gn_attenuation; snf[si] = g->gn_noise_figure; si++;
        }
        for (int c = 0; c < e->C; c++)
            if (c != ch && !e->avail[(size_t)l * e->C + c]) {
                int se = g->modulation_level[((size_t)row * e->C + c) * g->k_table + idp];
                vbw[vi] = g->gn_channel_bandwidth_hz; vcf[vi] = g->gn_center_frequency_hz[c];
                vse[vi] = se < 1 ? 1 : (se > 6 ? 6 : se); vi++;
            }
    }
    link_span_off[nl] = si; link_svc_off[nl] = vi;
    int32_t check_link_off[2] = {0, nl};
    double bw = g->gn_channel_bandwidth_hz, fc = g->gn_center_frequency_hz[ch], pw = g->gn_launch_power_w, out = 0;
    orc_osnr_batch b = {1, nl, si, vi, check_link_off, link_span_off, link_svc_off, &bw, &fc, &pw, slen, satt, snf, vbw, vcf, vse, vself};
    orc_gn_osnr(&b, &out);
    free(link_span_off); free(link_svc_off); free(slen); free(satt); free(snf); free(vbw); free(vcf); free(vse); free(vself);
    return out;
}

/* PhyRMSAEnv.step((path, channels)) (phy_rmsa_env.py:272-424) */
void orc_phy_step(orc_phy_env *e, const orc_phy_action *act, orc_phy_result *out) {
    pservice *s = e->current;
    double gn_last = NAN;
    s->accepted = 0; s->virtual_layer = 0;
    if (act->path != -2) {
        if (act->path > 10) {
            /* virtual layer: _service_acceptance(True) then _provision_virtual_path (:280-288, 625-659) */
            int idp = act->path - 20;
            s->virtual_layer = 1; s->accepted = 1;
            e->c.services_accepted++; e->c.episode_services_accepted++;
            e->c.bit_rate_provisioned += s->bit_rate; e->c.episode_bit_rate_provisioned += s->bit_rate;
            cs_list *l = cs_of(e, s->src, s->dst, idp);
            for (int i = 0; i < act->n; i++) {
                int q = cs_find(l, act->ch[i]);
                cs_entry t = l->e[q];
                int taken = (int)act->used[i];
                cs_remove(l, q);
                cs_entry u = { t.ch, t.used + taken, t.free_ - taken, t.cap };
                cs_append(l, u);
                s->ch[i] = act->ch[i]; s->used[i] = taken; s->cap[i] = act->cap[i];
            }
            s->nch = act->n; s->idp = idp; s->path_gid = ppath_gid(e, s->src, s->dst, idp);
            running_add(e, s);
            pheap_push(e, s->arrival_time + s->holding_time, s);
        } else if (act->path >= 0 && act->path < e->K) {
            int gid = ppath_gid(e, s->src, s->dst, act->path);
            int free_flag = 1; /* is_path_free_on_channels :1019-1027 */
            for (int i = 0; i < act->n; i++) free_flag &= is_channel_free(e, gid, act->ch[i]);
            if (free_flag && e->cfg.gn_on) {
                /* GN gate (not in the reference): every chosen channel must reach the level the table promised */
                const int row0 = e->cfg.pair_table_row[s->src * e->N + s->dst];
                for (int i = 0; i < act->n; i++) {
                    const double gdb = gn_gsnr_db(e, gid, act->path, row0, act->ch[i]);
                    int level = 0;
                    for (int t = 0; t < e->cfg.gn_num_thresholds; t++) level += gdb >= e->cfg.gn_thresholds_db[t];
                    gn_last = gdb;
                    if (level < act->cap[i]) { free_flag = 0; break; }
                }
            }
            if (free_flag) {
                /* _provision_path :544-623 */
                int row = e->cfg.pair_table_row[s->src * e->N + s->dst];
                for (int h = e->topo.path_link_off[gid]; h < e->topo.path_link_off[gid + 1]; h++)
                    for (int i = 0; i < act->n; i++) e->avail[(size_t)e->topo.path_links[h] * e->C + act->ch[i]] = 0;
                cs_list *l = cs_of(e, s->src, s->dst, act->path);
                for (int i = 0; i < act->n; i++) {
                    double g = e->cfg.gsnr[((size_t)row * e->C + act->ch[i]) * e->cfg.k_table + act->path];
                    e->total_gsnr_episode += g;
                    e->total_modulation_level_episode += act->cap[i];
                    e->channels_accepted_episode += 1;
                    s->ch[i] = act->ch[i]; s->used[i] = (int)act->used[i]; s->cap[i] = act->cap[i];
                    if (act->free_[i] != 0) {
                        cs_entry v = { act->ch[i], (int)act->used[i], (int)act->free_[i], act->cap[i] };
                        cs_append(l, v);
                    }
                }
                s->nch = act->n; s->path_gid = gid; s->idp = act->path;
                running_add(e, s);
                /* _service_acceptance(False) :767-778 */
                s->accepted = 1;
                e->c.services_accepted++; e->c.episode_services_accepted++;
                e->total_path_length_episode += e->topo.path_length[gid];
                e->total_path_index_episode += act->path + 1;
                e->physical_services_accepted_episode += 1;
                e->c.bit_rate_provisioned += s->bit_rate; e->c.episode_bit_rate_provisioned += s->bit_rate;
                pheap_push(e, s->arrival_time + s->holding_time, s);
            }
        }
    }
    if (out) {
        out->accepted = s->accepted;
        out->reward = s->accepted ? 1.0 : 0.0;
        out->number_cuts_total = total_cuts(e);
        out->rss_total_metric = total_r_spatial(e);
        out->service_blocking_rate = (double)(e->c.services_processed - e->c.services_accepted) / (double)e->c.services_processed;
        out->episode_service_blocking_rate = (double)(e->c.episode_services_processed - e->c.episode_services_accepted) /
                                             (double)e->c.episode_services_processed;
        out->bit_rate_blocking_rate = (double)(e->c.bit_rate_requested - e->c.bit_rate_provisioned) / (double)e->c.bit_rate_requested;
        out->episode_bit_rate_blocking_rate = (double)(e->c.episode_bit_rate_requested - e->c.episode_bit_rate_provisioned) /
                                              (double)e->c.episode_bit_rate_requested;
        out->total_path_length = e->total_path_length_episode / (double)(e->physical_services_accepted_episode + 1);
        out->avrage_gsnr = e->total_gsnr_episode / (double)(e->channels_accepted_episode + 1);
        out->total_modulation_level = e->total_modulation_level_episode;
        out->channels_accepted = e->channels_accepted_episode;
        out->average_path_index = (double)e->total_path_index_episode / (double)(e->physical_services_accepted_episode + 1);
        out->path_index = e->total_path_index_episode;
        out->physical_paths = e->physical_services_accepted_episode;
        out->num_moves = (double)e->counted_moves / 2 + (double)e->counted_moves_groom;
        out->num_moves_groom = e->counted_moves_groom;
        out->num_defrag_cycle = e->counted_defrag_cycles;
        out->gn_gsnr_db = gn_last;
    }
    e->new_service = 0;
    next_service(e);
    if (e->cfg.defrag_period > 0 && e->c.services_processed % e->cfg.defrag_period == 0) periodic_defragmentation(e);
    if (out) out->done = (e->c.episode_services_processed == e->cfg.episode_length);
}

void orc_phy_get_counters(const orc_phy_env *e, orc_counters *out) { *out = e->c; }
double orc_phy_current_time(const orc_phy_env *e) { return e->current_time; }
void orc_phy_get_available_channels(const orc_phy_env *e, uint8_t *out) { memcpy(out, e->avail, (size_t)e->E * e->C); }
int orc_phy_num_running(const orc_phy_env *e) { return e->n_running; }
/* channel_state[src, dst, idp] as (channel, used, free, capacity) rows in list order; returns the list length */
int orc_phy_channel_state(orc_phy_env *e, int src, int dst, int idp, int32_t *out, int max_entries) {
    const cs_list *l = cs_of(e, src, dst, idp);
    for (int i = 0; i < l->n && i < max_entries; i++) {
        out[4 * i] = l->e[i].ch; out[4 * i + 1] = l->e[i].used; out[4 * i + 2] = l->e[i].free_; out[4 * i + 3] = l->e[i].cap;
    }
    return l->n;
}

void orc_phy_run(orc_phy_env *e, int policy, int64_t n_steps, int reset_on_done, orc_phy_trace *tr) {
    orc_phy_action act;
    orc_phy_result r;
    for (int64_t i = 0; i < n_steps; i++) {
        const pservice *s = e->current;
        if (tr) {
            if (tr->service_id) tr->service_id[i] = s->service_id;
            if (tr->src) tr->src[i] = s->src;
            if (tr->dst) tr->dst[i] = s->dst;
            if (tr->bit_rate) tr->bit_rate[i] = s->bit_rate;
            if (tr->arrival) tr->arrival[i] = s->arrival_time;
            if (tr->holding) tr->holding[i] = s->holding_time;
        }
        orc_phy_policy(e, policy, &act);
        orc_phy_step(e, &act, &r);
        if (tr) {
            if (tr->act_path) tr->act_path[i] = act.path;
            if (tr->n_channels) tr->n_channels[i] = act.n;
            if (tr->channels)
                for (int q = 0; q < ORC_PHY_MAX_CH; q++) tr->channels[i * ORC_PHY_MAX_CH + q] = q < act.n ? act.ch[q] : -1;
            if (tr->ch_used)
                for (int q = 0; q < ORC_PHY_MAX_CH; q++) tr->ch_used[i * ORC_PHY_MAX_CH + q] = q < act.n ? act.used[q] : 0.0;
            if (tr->ch_cap)
                for (int q = 0; q < ORC_PHY_MAX_CH; q++) tr->ch_cap[i * ORC_PHY_MAX_CH + q] = q < act.n ? act.cap[q] : 0;
            if (tr->accepted) tr->accepted[i] = (uint8_t)r.accepted;
            if (tr->done) tr->done[i] = (uint8_t)r.done;
            if (tr->services_accepted) tr->services_accepted[i] = e->c.services_accepted;
            if (tr->number_cuts_total) tr->number_cuts_total[i] = r.number_cuts_total;
            if (tr->rss_total_metric) tr->rss_total_metric[i] = r.rss_total_metric;
            if (tr->total_path_length) tr->total_path_length[i] = r.total_path_length;
            if (tr->avrage_gsnr) tr->avrage_gsnr[i] = r.avrage_gsnr;
            if (tr->total_modulation_level) tr->total_modulation_level[i] = r.total_modulation_level;
            if (tr->channels_accepted) tr->channels_accepted[i] = r.channels_accepted;
            if (tr->average_path_index) tr->average_path_index[i] = r.average_path_index;
            if (tr->path_index) tr->path_index[i] = r.path_index;
            if (tr->physical_paths) tr->physical_paths[i] = r.physical_paths;
            if (tr->episode_service_blocking_rate) tr->episode_service_blocking_rate[i] = r.episode_service_blocking_rate;
            if (tr->bit_rate_blocking_rate) tr->bit_rate_blocking_rate[i] = r.bit_rate_blocking_rate;
            if (tr->current_time) tr->current_time[i] = e->current_time;
            if (tr->num_moves) tr->num_moves[i] = r.num_moves;
            if (tr->num_moves_groom) tr->num_moves_groom[i] = r.num_moves_groom;
            if (tr->num_defrag_cycle) tr->num_defrag_cycle[i] = r.num_defrag_cycle;
            if (tr->gn_gsnr_db) tr->gn_gsnr_db[i] = r.gn_gsnr_db;
            if (tr->n_running) tr->n_running[i] = e->n_running;
            if (tr->free_total) {
                int64_t f = 0;
                for (size_t q = 0; q < (size_t)e->E * e->C; q++) f += e->avail[q];
                tr->free_total[i] = f;
            }
        }
        if (r.done && reset_on_done) orc_phy_reset(e, 1);
    }
}
