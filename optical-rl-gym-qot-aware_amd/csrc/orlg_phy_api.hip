// orlg_phy_api.hip -- host side of the QoT-aware (PhyRMSA) path: orlg_phy_* entry points of include/orlg.h.
// Its own translation unit; the step kernel is instantiated per word count in orlg_inst_phy.hip.
#include "orlg_host.h"
#include "orlg_phy_kernels.hip"   // data layout + device helpers

struct orlg_phy_env {
    OrlgPhyParams p;
    int W, device, waves_per_block, num_paths, num_cu;
    int resident_blocks[32];   // per kernel variant (phy_launch)
    uint32_t ticket_base;
    hipStream_t stream;
    bool own_stream;
    size_t lds_block_bytes;
    std::vector<void *> bufs;
    unsigned char *staging;
    size_t staging_bytes;
    void *io_buf[ORLG_PHY_NUM_OUTS];
    size_t io_cap[ORLG_PHY_NUM_OUTS];
    int32_t *d_act_path;
    int16_t *d_act_ch;
    OrlgErrWord err;         // sticky error word the kernel sets when a queue / channel_state list / work list overflows
    char last_kernel[96];    // name and shape of the kernel behind the last launch (orlg_phy_last_kernel)
};

static int phy_sync_check(orlg_phy_env *e) {
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (e->err.host && *e->err.host)
        return fail(ORLG_ERR_QUEUE_FULL, "a release queue, channel_state list or defragmentation work list overflowed: raise "
                                         "queue_capacity / channel_state_capacity / defrag_capacity");
    return ORLG_OK;
}

typedef orlg_phy_kernel_t phy_kernel_t;
static phy_kernel_t pick_phy(int W, int variant) {
    switch (W) {
#define X(n) case n: return orlg_phy_kernel_W##n ? orlg_phy_kernel_W##n(variant) : nullptr;
        ORLG_FOR_EACH_PHY_W(X)
#undef X
        default: return nullptr;
    }
}

__global__ void orlg_phy_clear_kernel(OrlgPhyParams p, int W, int keep_rng) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
    for (size_t i = tid; i < (size_t)p.B * p.NW; i += nth) p.occ[i] = valid_mask(p.C, (int)(i % W));
    for (size_t i = tid; i < (size_t)p.B * p.N * p.N * p.K; i += nth) p.cs_n[i] = 0;
    for (size_t i = tid; i < (size_t)p.B; i += nth) {
        OrlgPhyScalars s;
        memset(&s, 0, sizeof(s));
        s.mt_idx = keep_rng ? p.scal[i].mt_idx : ORLG_MT_N;
        // pre-generated arrivals are part of the RNG stream: a full reset keeps them
        s.ring_pos = keep_rng ? p.scal[i].ring_pos : 0;
        s.ring_cnt = keep_rng ? p.scal[i].ring_cnt : 0;
        p.scal[i] = s;
    }
}

enum { PX_REQUEST, PX_COUNTERS, PX_TIME, PX_RUNNING, PX_EPISODE };
__global__ void orlg_phy_extract_kernel(OrlgPhyParams p, int what, unsigned char *out) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
    const size_t B = p.B;
    if (what == PX_REQUEST) {
        orlg_request *o = reinterpret_cast<orlg_request *>(out);
        for (size_t i = tid; i < B; i += nth) {
            const OrlgPhyScalars &s = p.scal[i];
            o[i].service_id = s.req_sid; o[i].src = s.req_src; o[i].dst = s.req_dst;
            o[i].bit_rate = reinterpret_cast<const int32_t *>(p.tables + p.t_bitrates)[s.req_br];
            o[i].arrival_time = s.req_arrival; o[i].holding_time = s.req_holding;
        }
    } else if (what == PX_COUNTERS) {
        int64_t *o = reinterpret_cast<int64_t *>(out);
        for (size_t i = tid; i < B * 8; i += nth) o[i] = p.scal[i / 8].c[i % 8];
    } else if (what == PX_TIME) {
        double *o = reinterpret_cast<double *>(out);
        for (size_t i = tid; i < B; i += nth) o[i] = p.scal[i].current_time;
    } else if (what == PX_RUNNING) {
        int32_t *o = reinterpret_cast<int32_t *>(out);
        for (size_t i = tid; i < B; i += nth) o[i] = p.scal[i].n_running;
    } else if (what == PX_EPISODE) {
        orlg_phy_episode_stats *o = reinterpret_cast<orlg_phy_episode_stats *>(out);
        for (size_t i = tid; i < B; i += nth) {
            const OrlgPhyScalars &s = p.scal[i];
            o[i].total_path_length = s.total_path_length; o[i].total_gsnr = s.total_gsnr;
            o[i].total_path_index = s.total_path_index; o[i].total_modulation_level = s.total_mod;
            o[i].channels_accepted = s.channels_accepted; o[i].physical_services_accepted = s.physical_accepted;
            o[i].episodes_done = s.episodes_done; o[i].queue_overflow = s.q_overflow;
            o[i].counted_moves = s.counted_moves; o[i].counted_moves_groom = s.counted_moves_groom;
            o[i].counted_defrag_cycles = s.counted_defrag_cycles;
        }
    }
}

__global__ __launch_bounds__(256) void orlg_phy_reduce_kernel(const OrlgPhyScalars *scal, int B, long long *out) {
    __shared__ long long part[256][11];
    long long acc[11];
    for (int q = 0; q < 11; ++q) acc[q] = 0;
    for (int i = threadIdx.x; i < B; i += 256) {
        for (int q = 0; q < 8; ++q) acc[q] += scal[i].c[q];
        acc[8] += scal[i].episodes_done;
        acc[9] += 1;
        acc[10] += scal[i].q_overflow;
    }
    for (int q = 0; q < 11; ++q) part[threadIdx.x][q] = acc[q];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
            for (int q = 0; q < 11; ++q) part[threadIdx.x][q] += part[threadIdx.x + s][q];
        __syncthreads();
    }
    if (threadIdx.x < 16) out[threadIdx.x] = threadIdx.x < 11 ? part[0][threadIdx.x] : 0;
}

// the tables of gn_gsnr: one thread per (channel, interferer) pair / per link.  The expressions are examples/calculate_osnr.py's
// (:22-48), as gn_gsnr evaluated them per check until round 3.
__global__ void orlg_gn_tables_kernel(const OrlgPhyParams p, double *A, double *R, double *L) {
    const double beta_2 = -21.3e-27, pi = 3.141592653589793;
    const double bw = p.gn_bw, att = p.gn_att;
    const double l_eff_a = 1 / (2 * att);
    const int n = p.C * p.cpad;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int ch = i / p.cpad, c = i - ch * p.cpad;
        double a = 0.0, rr = 0.0;
        if (c < p.C && c != ch) {
            const double fc = p.gn_cf[ch], sf = p.gn_cf[c];
            a = asinh(pi * pi * fabs(beta_2) * l_eff_a * bw * (sf - fc + (bw / 2))) -
                asinh(pi * pi * fabs(beta_2) * l_eff_a * bw * (sf - fc - (bw / 2)));
            rr = bw / fabs(sf - fc);
        }
        A[i] = a; R[i] = rr;
    }
    for (int l = blockIdx.x * blockDim.x + threadIdx.x; l < p.E; l += gridDim.x * blockDim.x) {
        const double len = p.gn_spanlen[l];
        const double l_eff = (1 - exp(-2 * att * len * 1e3)) / (2 * att);
        L[4 * l] = l_eff;
        L[4 * l + 1] = l_eff / (len * 1e3);
        L[4 * l + 2] = exp(2 * att * len * 1e3) - 1;
        L[4 * l + 3] = 0.0;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) L[4 * p.E] = asinh(pi * pi * fabs(beta_2) * (bw * bw) / (4 * att));
}

__global__ void orlg_phy_reseed_kernel(OrlgPhyScalars *scal, int B) {   // as orlg_reseed_kernel
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B; i += gridDim.x * blockDim.x) {
        scal[i].mt_idx = ORLG_MT_N;
        scal[i].ring_pos = 0; scal[i].ring_cnt = 0;
    }
}
// the sticky error word recomputed from the scalars (orlg_phy_load_state; mapped host memory: a plain store)
__global__ void orlg_phy_overflow_store_kernel(const OrlgPhyScalars *scal, int B, int *err_flag) {
    int any = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B; i += gridDim.x * blockDim.x) any |= scal[i].q_overflow;
    if (any) *err_flag = 1;
}

static int phy_launch(orlg_phy_env *e, const OrlgPhyParams &p) {
    // kernel variant (orlg_inst_phy.hip): 0 the step proper, 1 + periodic defragmentation, 2 + defragmentation and GN-model
    // admission check, 3 + GN-model admission check alone
    const int df = p.gn_on ? (p.defrag_period > 0 ? 2 : 3) : p.defrag_period > 0 ? 1 : 0;
    const int pol = p.mode == ORLG_MODE_STEP ? p.policy : ORLG_PHY_POLICY_EXTERNAL;   // one instantiation per policy
    const int variant = df + 4 * (pol + 1);
    phy_kernel_t k = pick_phy(e->W, variant);
    if (!k) return fail(ORLG_ERR_INVALID, "no PhyRMSA kernel for W=%d", e->W);
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)e->lds_block_bytes));
    const int wpb = e->waves_per_block;
    // the instantiations differ in registers and scratch: the resident workgroups (= the grid of the work queue) per variant
    if (e->resident_blocks[variant] <= 0) {
        int nb = 0;
        if (e->num_cu <= 0) {
            hipDeviceProp_t prop;
            HIP_TRY(hipGetDeviceProperties(&prop, e->device));
            e->num_cu = prop.multiProcessorCount;
        }
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(k), ORLG_WAVE * wpb, e->lds_block_bytes));
        e->resident_blocks[variant] = (nb > 0 ? nb : 1) * e->num_cu;
    }
    int nblocks = (p.B + wpb - 1) / wpb;
    if (nblocks > e->resident_blocks[variant]) nblocks = e->resident_blocks[variant];
    OrlgPhyParams q = p;
    // the node-degree vectors are rebuilt from the occupancy by every launch that evaluates the cut metric, and only by those
    q.use_nv = (p.use_nv && p.mode == ORLG_MODE_STEP &&
                (p.policy == ORLG_PHY_POLICY_BMFA_CUT || p.policy == ORLG_PHY_POLICY_FAFF || (p.defrag_period > 0 && p.defrag_metric == 0)))
                   ? 1 : 0;
    q.ticket_base = e->ticket_base;
    q.ticket_stride = (p.mode != ORLG_MODE_STEP || p.n_steps <= 16) ? 1u : 0u;
    if (!q.ticket_stride) e->ticket_base += (uint32_t)p.B;
    dim3 grid(nblocks), block(ORLG_WAVE * wpb);
    hipLaunchKernelGGL(k, grid, block, e->lds_block_bytes, e->stream, q);
    HIP_TRY(hipGetLastError());
    snprintf(e->last_kernel, sizeof(e->last_kernel), "orlg_phy_kernel<%d,%s,%d> grid=%d block=%d lds=%zu", e->W,
             df == 2 ? "true,true" : df == 1 ? "true,false" : df == 3 ? "false,true" : "false,false", pol, nblocks, ORLG_WAVE * wpb, e->lds_block_bytes);
    return ORLG_OK;
}

static int phy_extract(orlg_phy_env *e, int what, void *out, size_t bytes) {
    if (bytes > e->staging_bytes) {
        if (e->staging) HIP_TRY(hipFree(e->staging));
        e->staging = nullptr;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&e->staging), bytes));
        e->staging_bytes = bytes;
    }
    hipLaunchKernelGGL(orlg_phy_extract_kernel, dim3(256), dim3(256), 0, e->stream, e->p, what, e->staging);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, e->staging, bytes, hipMemcpyDefault, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return ORLG_OK;
}

extern "C" {

int orlg_phy_destroy(orlg_phy_env *e) {
    if (!e) return ORLG_OK;
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
    for (void *b : e->bufs) (void)hipFree(b);
    if (e->staging) (void)hipFree(e->staging);
    for (int i = 0; i < ORLG_PHY_NUM_OUTS; i++)
        if (e->io_buf[i]) (void)hipFree(e->io_buf[i]);
    if (e->d_act_path) (void)hipFree(e->d_act_path);
    if (e->d_act_ch) (void)hipFree(e->d_act_ch);
    orlg_err_word_destroy(&e->err);
    if (e->own_stream) (void)hipStreamDestroy(e->stream);
    delete e;
    return ORLG_OK;
}

int orlg_phy_create(const orlg_topology *t, const orlg_phy_config *c, int32_t batch, const uint64_t *seeds,
                    uint64_t base_seed, int32_t device, orlg_phy_env **out) {
    if (!t || !c || !out) return fail(ORLG_ERR_INVALID, "null argument");
    *out = nullptr;
    const int N = t->num_nodes, E = t->num_links, K = t->k_paths, C = c->num_channels, NBR = c->num_bit_rates;
    if (batch < 1) return fail(ORLG_ERR_INVALID, "batch must be >= 1");
    if (N < 2 || N > 64) return fail(ORLG_ERR_INVALID, "num_nodes %d not in 2..64", N);
    if (E < 1 || E > 255) return fail(ORLG_ERR_INVALID, "num_links %d not in 1..255", E);
    if (C < 1 || C > 320) return fail(ORLG_ERR_INVALID, "num_channels %d not in 1..320", C);
    if (NBR < 1 || NBR > 64) return fail(ORLG_ERR_INVALID, "num_bit_rates %d not in 1..64", NBR);
    if (K < 1 || K > ORLG_PHY_MAX_K) return fail(ORLG_ERR_INVALID, "k_paths %d not in 1..%d", K, ORLG_PHY_MAX_K);
    if (c->k_table < K) return fail(ORLG_ERR_INVALID, "QoT tables have %d k-path columns, topology has k=%d", c->k_table, K);
    if (t->num_paths < 1 || t->num_paths >= 65536) return fail(ORLG_ERR_INVALID, "num_paths out of range");
    const int W = (C + 63) / 64;
    if (K * W > 64) return fail(ORLG_ERR_INVALID, "k_paths * words_per_link exceeds one wavefront");
    if (!(c->arrival_lambda > 0) || !(c->holding_lambda > 0)) return fail(ORLG_ERR_INVALID, "lambdas must be positive");
    for (int i = 0; i < N * N; i++) {
        int s = i / N, d = i % N;
        if (s == d) continue;
        if (t->pair_path_count[i] != K || t->pair_path_base[i] < 0) return fail(ORLG_ERR_INVALID, "pair (%d,%d) lacks k path records", s, d);
        int row = c->pair_table_row[i];
        if (row < 0 || row >= c->num_table_rows) return fail(ORLG_ERR_INVALID, "pair (%d,%d) has no QoT table row", s, d);
        // the greedy channel order below relies on level >= 1 (a level-0 entry would sort FIRST in the reference
        // because -np.uint8(0) == 0, SURVEY 8c caveat 3): refuse such tables instead of silently diverging
        for (int ch = 0; ch < C; ch++)
            for (int k = 0; k < K; k++) {
                int lv = c->modulation_level[((size_t)row * C + ch) * c->k_table + k];
                if (lv < 1 || lv > 31) return fail(ORLG_ERR_INVALID, "modulation level %d at row %d channel %d path %d not in 1..31", lv, row, ch, k);
            }
    }
    int ndev = orlg_device_count();
    if (ndev < 1) return fail(ORLG_ERR_NO_DEVICE, "no HIP device visible: liborlg has no CPU path");
    if (device < 0 || device >= ndev) return fail(ORLG_ERR_INVALID, "device %d out of range (have %d)", device, ndev);
    HIP_TRY(hipSetDevice(device));

    orlg_phy_env *e = new orlg_phy_env();
    memset(&e->p, 0, sizeof(e->p));
    e->W = W; e->device = device; e->own_stream = true; e->staging = nullptr; e->staging_bytes = 0;
    e->d_act_path = nullptr; e->d_act_ch = nullptr; e->num_paths = t->num_paths;
    e->err.host = nullptr; e->err.dev = nullptr; e->last_kernel[0] = 0;
    memset(e->resident_blocks, 0, sizeof(e->resident_blocks)); e->num_cu = 0; e->ticket_base = 0;
    for (int i = 0; i < ORLG_PHY_NUM_OUTS; i++) { e->io_buf[i] = nullptr; e->io_cap[i] = 0; }
    hipError_t he = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
    if (he != hipSuccess) { delete e; return fail(ORLG_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(he)); }
    {
        int rc0 = orlg_err_word_create(&e->err);
        if (rc0) { orlg_phy_destroy(e); return rc0; }
    }
    OrlgPhyParams &p = e->p;
    p.err_flag = e->err.dev;
    p.B = batch; p.N = N; p.E = E; p.C = C; p.K = K; p.NBR = NBR; p.NW = E * W;
    p.episode_length = c->episode_length; p.num_rows = c->num_table_rows; p.cpad = W * 64;
    p.arrival_lambda = c->arrival_lambda; p.holding_lambda = c->holding_lambda;
    p.grooming = c->grooming ? 1 : 0;
    p.defrag_period = c->defrag_period > 0 ? c->defrag_period : 0;
    p.number_moves = c->number_moves; p.defrag_metric = c->defrag_metric ? 1 : 0;
    if (p.defrag_period > 0 && p.number_moves < 0) { orlg_phy_destroy(e); return fail(ORLG_ERR_INVALID, "number_moves must be >= 0"); }
    {
        // channel_state[src, dst, k-path] lists (virtual layer): one entry per lit, partially used channel of the
        // (pair, path); sized from the mean number of services per ordered pair, overflow is reported, never dropped
        int cl = c->channel_state_capacity;
        if (cl <= 0) {
            double m = (c->arrival_lambda / c->holding_lambda) / ((double)N * (N - 1));
            cl = (int)(m + 6.0 * std::sqrt(m) + 8.0);
        }
        int pw2 = 8;
        while (pw2 < cl) pw2 <<= 1;
        if (pw2 > ORLG_CS_MAX) {
            if (c->channel_state_capacity > ORLG_CS_MAX) { orlg_phy_destroy(e); return fail(ORLG_ERR_INVALID, "channel_state_capacity %d exceeds %d", cl, ORLG_CS_MAX); }
            pw2 = ORLG_CS_MAX;
        }
        p.cs_len = pw2;
    }
    // release queue: at most Poisson(load) services in progress, and never more than the channel-links can hold
    int Q = c->queue_capacity;
    if (Q <= 0) {
        double load = c->arrival_lambda / c->holding_lambda;
        double q = load + 8.0 * std::sqrt(load) + 32.0;
        double cap = (double)E * C / 2.0 + 64.0;
        Q = (int)(q < cap ? q : cap);
    }
    Q = ((Q + 63) / 64) * 64;
    if (Q > 8192) { orlg_phy_destroy(e); return fail(ORLG_ERR_INVALID, "queue_capacity %d too large (max 8192)", Q); }
    p.Q = Q;
    auto up16 = [](int v) { return (v + 15) & ~15; };
    int off = 0;
    p.l_occ = off; off = up16(off + p.NW * 8);
    p.l_nbt = off; off = up16(off + ORLG_PHY_NB * 8);
    p.l_nbi = off; off = up16(off + ORLG_PHY_NB * 2);
    p.l_scratch = off; off = up16(off + 256 + W * 64 * 8);
    p.l_wsc = off; off = up16(off + (int)sizeof(PhyWaveScalars));
    p.l_wave_bytes = off;

    int rc = ORLG_OK;
#define TRY(x) do { rc = (x); if (rc) { orlg_phy_destroy(e); return rc; } } while (0)
    {
        std::vector<unsigned char> blob;
        auto put = [&](const void *src, size_t bytes) {
            size_t at = blob.size();
            blob.resize((at + bytes + 15) & ~(size_t)15, 0);
            memcpy(blob.data() + at, src, bytes);
            return (int32_t)at;
        };
        p.t_pair = put(t->pair_path_base, (size_t)N * N * 4);
        std::vector<OrlgPathRec> recs(t->num_paths);
        for (int g = 0; g < t->num_paths; g++) {
            memset(&recs[g], 0, sizeof(OrlgPathRec));
            int h = t->path_hops[g];
            if (h < 1 || h > ORLG_MAX_HOPS || t->path_link_off[g + 1] - t->path_link_off[g] != h) {
                orlg_phy_destroy(e);
                return fail(ORLG_ERR_INVALID, "path %d: hops %d not in 1..%d or CSR mismatch", g, h, ORLG_MAX_HOPS);
            }
            recs[g].hops = (uint8_t)h; recs[g].se = (uint8_t)t->path_se[g];
            for (int i = 0; i < h; i++) {
                int l = t->path_links[t->path_link_off[g] + i];
                if (l < 0 || l >= E) { orlg_phy_destroy(e); return fail(ORLG_ERR_INVALID, "path %d: link out of range", g); }
                recs[g].link[i] = (uint8_t)l;
            }
        }
        p.t_recs = put(recs.data(), recs.size() * sizeof(OrlgPathRec));
        p.t_bitrates = put(c->bit_rates, (size_t)NBR * 4);
        p.t_brcum = put(c->bit_rate_cum, (size_t)NBR * 8);
        p.t_srccum = put(c->src_cum, (size_t)N * 8);
        p.t_dstcum = put(c->dst_cum, (size_t)N * N * 8);
        p.t_pairrow = put(c->pair_table_row, (size_t)N * N * 4);
        p.use_masks = E <= 32 ? 1 : 0;
        const int nadj = c->adj_off[t->num_paths];
        std::vector<uint16_t> adj(nadj > 0 ? nadj : 1, 0);
        int wsum_max = 0;
        for (int g = 0; g < t->num_paths; g++) {
            int ws = 0;
            for (int j = c->adj_off[g]; j < c->adj_off[g + 1]; j++) {
                int l = c->adj_link[j], w = c->adj_weight[j];
                if (l < 0 || l >= E || w < 1 || w > 2) { orlg_phy_destroy(e); return fail(ORLG_ERR_INVALID, "bad cut adjacency entry %d", j); }
                adj[j] = (uint16_t)(l | (w << 8));
                ws += w;
            }
            wsum_max = ws > wsum_max ? ws : wsum_max;
        }
        if (wsum_max > 127) { orlg_phy_destroy(e); return fail(ORLG_ERR_INVALID, "cut metric range %d exceeds the 8-bit score field", wsum_max); }
        p.t_adjoff = put(c->adj_off, (size_t)(t->num_paths + 1) * 4);
        p.t_adj = put(adj.data(), adj.size() * 2);
        p.t_masks = 0;
        if (p.use_masks) {
            std::vector<OrlgPathMasks> masks(t->num_paths);
            for (int g = 0; g < t->num_paths; g++) {
                masks[g].path = 0u;
                for (int i = 0; i < recs[g].hops; i++) masks[g].path |= 1u << recs[g].link[i];
            }
            p.t_masks = put(masks.data(), masks.size() * sizeof(OrlgPathMasks));
        }
        std::vector<double> sq((size_t)E * E + 1);
        for (size_t k = 0; k < sq.size(); k++) sq[k] = std::sqrt((double)k);
        p.t_sqrt = put(sq.data(), sq.size() * 8);
        p.t_plen = put(t->path_length, (size_t)t->num_paths * 8);
        {
            // record -> the unordered node pair it serves (ksp[a, b] and ksp[b, a] are the same records)
            std::vector<uint16_t> pp(t->num_paths, 0);
            for (int a = 0; a < N; a++)
                for (int b = a + 1; b < N; b++) {
                    if (t->pair_path_base[a * N + b] != t->pair_path_base[b * N + a]) {
                        orlg_phy_destroy(e);
                        return fail(ORLG_ERR_INVALID, "pair (%d,%d) and (%d,%d) must share their path records", a, b, b, a);
                    }
                    for (int k = 0; k < K; k++) pp[t->pair_path_base[a * N + b] + k] = (uint16_t)(a * N + b);
                }
            p.t_pathpair = put(pp.data(), pp.size() * 2);
        }
        // node-degree vectors of the cut metric (include/orlg.h, orlg_phy_config::path_node_weights): D[channel] = 16 nibbles,
        // links free at every node, in the wave's LDS; ORLG_PHY_NODEVEC=0 keeps the adjacency lists (tests)
        const char *nvm = getenv("ORLG_PHY_NODEVEC");
        p.use_nv = (c->path_node_weights && c->node_degree && c->link_ends && N <= 16 && !(nvm && nvm[0] == '0')) ? 1 : 0;
        if (p.use_nv) {
            std::vector<uint64_t> lnib(E, 0);
            int deg[16] = {0};
            for (int l = 0; l < E && p.use_nv; l++) {
                const int a = c->link_ends[2 * l], b = c->link_ends[2 * l + 1];
                if (a < 0 || a >= N || b < 0 || b >= N || a == b) { orlg_phy_destroy(e); return fail(ORLG_ERR_INVALID, "link_ends[%d] = (%d, %d)", l, a, b); }
                lnib[l] = (1ull << (4 * a)) | (1ull << (4 * b));
                deg[a]++; deg[b]++;
            }
            for (int v = 0; v < N; v++) {
                if (deg[v] != (int)c->node_degree[v]) { orlg_phy_destroy(e); return fail(ORLG_ERR_INVALID, "node_degree[%d] = %d, link_ends say %d", v, (int)c->node_degree[v], deg[v]); }
                if (deg[v] > 15) p.use_nv = 0;   // a nibble per node
            }
            if (p.use_nv) p.t_lnib = put(lnib.data(), lnib.size() * 8);
        }
        p.tab_bytes = (int32_t)blob.size();
        p.l_outs = p.tab_bytes;
        p.l_mtstage = p.tab_bytes + up16(ORLG_PHY_NUM_OUTS * 8);
        p.l_shared_bytes = p.l_mtstage + up16(ORLG_MT_N * 4 + 4);
        unsigned char *d_blob;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_blob), blob.size()));
        e->bufs.push_back(d_blob);
        HIP_TRY(hipMemcpy(d_blob, blob.data(), blob.size(), hipMemcpyHostToDevice));
        p.tables = d_blob;
        // QoT tables re-laid [row][k-path][channel] so that the 64 lanes of a word read 64 consecutive bytes
        std::vector<uint8_t> mt((size_t)c->num_table_rows * K * p.cpad, 0);
        std::vector<double> gt((size_t)c->num_table_rows * K * p.cpad, 0.0);
        for (int r = 0; r < c->num_table_rows; r++)
            for (int ch = 0; ch < C; ch++)
                for (int k = 0; k < K; k++) {
                    size_t src = ((size_t)r * C + ch) * c->k_table + k, dst = ((size_t)r * K + k) * p.cpad + ch;
                    mt[dst] = c->modulation_level[src];
                    gt[dst] = c->gsnr[src];
                }
        uint8_t *d_m; double *d_g;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_m), mt.size())); e->bufs.push_back(d_m);
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_g), gt.size() * 8)); e->bufs.push_back(d_g);
        HIP_TRY(hipMemcpy(d_m, mt.data(), mt.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_g, gt.data(), gt.size() * 8, hipMemcpyHostToDevice));
        p.mod_t = d_m; p.gsnr_t = d_g;
        if (p.defrag_period > 0) {
            // defragmentation rounds: the channels of one modulation level on a (table row, k-path) as W 64-bit masks -- "a free
            // channel of the candidate's level" (phy_rmsa_env.py:395) is then the path's free word AND one mask word
            std::vector<uint64_t> lm((size_t)c->num_table_rows * K * 32 * W, 0ull);
            for (int r = 0; r < c->num_table_rows; r++)
                for (int k = 0; k < K; k++)
                    for (int ch = 0; ch < C; ch++) {
                        const int lv = mt[((size_t)r * K + k) * p.cpad + ch] & 31;
                        lm[(((size_t)r * K + k) * 32 + lv) * W + (ch >> 6)] |= 1ull << (ch & 63);
                    }
            uint64_t *d_lm;
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_lm), lm.size() * 8)); e->bufs.push_back(d_lm);
            HIP_TRY(hipMemcpy(d_lm, lm.data(), lm.size() * 8, hipMemcpyHostToDevice));
            p.lvl_mask = d_lm;
        }
        // the levels of one channel on all K paths next to each other (8 bytes per channel): one load per channel and step
        std::vector<uint8_t> mk((size_t)c->num_table_rows * p.cpad * 8, 0);
        for (int r = 0; r < c->num_table_rows; r++)
            for (int k = 0; k < K; k++)
                for (int ch = 0; ch < C; ch++) mk[((size_t)r * p.cpad + ch) * 8 + k] = mt[((size_t)r * K + k) * p.cpad + ch];
        uint8_t *d_k;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_k), mk.size())); e->bufs.push_back(d_k);
        HIP_TRY(hipMemcpy(d_k, mk.data(), mk.size(), hipMemcpyHostToDevice));
        p.mod_k = reinterpret_cast<const uint32_t *>(d_k);
    }
    if (p.use_nv) {
        // D takes 8 bytes per channel and wave; it must not cost a resident workgroup (two of ORLG_MAX_WAVES_PER_BLOCK waves
        // fill the 128-VGPR budget of a CU) -- otherwise the adjacency lists stay
        const int nv_bytes = up16(C * 8);
        if (2 * ((size_t)p.l_shared_bytes + (size_t)ORLG_MAX_WAVES_PER_BLOCK * (p.l_wave_bytes + nv_bytes)) <= 160 * 1024) {
            p.l_nv = p.l_wave_bytes;
            p.l_wave_bytes += nv_bytes;
        } else {
            p.use_nv = 0;
        }
    }
    {
        int wpb = ORLG_MAX_WAVES_PER_BLOCK;
        while (wpb > 1 && (size_t)p.l_shared_bytes + (size_t)wpb * p.l_wave_bytes > 160 * 1024) wpb >>= 1;
        if ((size_t)p.l_shared_bytes + (size_t)wpb * p.l_wave_bytes > 160 * 1024) {
            orlg_phy_destroy(e);
            return fail(ORLG_ERR_INVALID, "tables + one environment exceed the 160 KiB LDS");
        }
        e->waves_per_block = wpb;
        e->lds_block_bytes = (size_t)p.l_shared_bytes + (size_t)wpb * p.l_wave_bytes;
    }
    auto alloc = [&](void **ptr, size_t bytes) -> int {
        HIP_TRY(hipMalloc(ptr, bytes ? bytes : 16));
        e->bufs.push_back(*ptr);
        return ORLG_OK;
    };
    TRY(alloc(reinterpret_cast<void **>(&p.occ), (size_t)batch * p.NW * 8));
    TRY(alloc(reinterpret_cast<void **>(&p.qtime), (size_t)batch * Q * 8));
    TRY(alloc(reinterpret_cast<void **>(&p.qrec), (size_t)batch * Q * sizeof(OrlgPhySvc)));
    TRY(alloc(reinterpret_cast<void **>(&p.mt), (size_t)batch * ORLG_MT_N * 4));
    TRY(alloc(reinterpret_cast<void **>(&p.ring_iat), (size_t)batch * ORLG_RING * 8));
    TRY(alloc(reinterpret_cast<void **>(&p.ring_ht), (size_t)batch * ORLG_RING * 8));
    TRY(alloc(reinterpret_cast<void **>(&p.ring_req), (size_t)batch * ORLG_RING * 4));
    {
        hipError_t er = hipMemset(p.ring_iat, 0, (size_t)batch * ORLG_RING * 8);
        if (er == hipSuccess) er = hipMemset(p.ring_ht, 0, (size_t)batch * ORLG_RING * 8);
        if (er == hipSuccess) er = hipMemset(p.ring_req, 0, (size_t)batch * ORLG_RING * 4);
        if (er != hipSuccess) { orlg_phy_destroy(e); return fail(ORLG_ERR_HIP, "hipMemset: %s", hipGetErrorString(er)); }
    }
    TRY(alloc(reinterpret_cast<void **>(&p.scal), (size_t)batch * sizeof(OrlgPhyScalars)));
    TRY(alloc(reinterpret_cast<void **>(&p.cs), (size_t)batch * N * N * K * p.cs_len * 4));
    TRY(alloc(reinterpret_cast<void **>(&p.cs_n), (size_t)batch * N * N * K));
    TRY(alloc(reinterpret_cast<void **>(&p.ticket), 16));
    {
        hipError_t er = hipMemset(p.ticket, 0, 16);
        if (er != hipSuccess) { orlg_phy_destroy(e); return fail(ORLG_ERR_HIP, "hipMemset: %s", hipGetErrorString(er)); }
    }
    if (p.use_nv) {
        // the records with c in the order the byte dot products take it: even nodes, then odd nodes (nv_split)
        std::vector<uint8_t> rec((size_t)t->num_paths * 32);
        memcpy(rec.data(), c->path_node_weights, rec.size());
        for (int g = 0; g < t->num_paths; g++) {
            const uint8_t *src = c->path_node_weights + (size_t)g * 32;
            for (int v = 0; v < 16; v++) {
                if (src[v] > 2) { orlg_phy_destroy(e); return fail(ORLG_ERR_INVALID, "path_node_weights[%d][%d] = %d", g, v, (int)src[v]); }
                rec[(size_t)g * 32 + (v & 1) * 8 + (v >> 1)] = src[v];
            }
        }
        uint4 *d_rec = nullptr;
        TRY(alloc(reinterpret_cast<void **>(&d_rec), rec.size()));
        hipError_t er = hipMemcpy(d_rec, rec.data(), rec.size(), hipMemcpyHostToDevice);
        if (er != hipSuccess) { orlg_phy_destroy(e); return fail(ORLG_ERR_HIP, "upload of the node weight records: %s", hipGetErrorString(er)); }
        p.nvrec = d_rec;
    }
    if (c->gn_gate) {
        const orlg_gn_gate *g = c->gn_gate;
        if (!g->channel_center_frequency_hz || !g->link_num_spans || !g->link_span_length_km || !g->thresholds_db || g->num_thresholds < 1 ||
            !(g->launch_power_w > 0) || !(g->channel_bandwidth_hz > 0) || !(g->attenuation_normalized > 0) || !(g->noise_figure > 0)) {
            orlg_phy_destroy(e);
            return fail(ORLG_ERR_INVALID, "gn_gate: missing array or non-positive physical parameter");
        }
        for (int l = 0; l < E; l++)
            if (g->link_num_spans[l] < 1 || !(g->link_span_length_km[l] > 0)) { orlg_phy_destroy(e); return fail(ORLG_ERR_INVALID, "gn_gate: link %d has no spans", l); }
        double *d_cf = nullptr, *d_sl = nullptr, *d_thr = nullptr;
        int32_t *d_ns = nullptr;
        TRY(alloc(reinterpret_cast<void **>(&d_cf), (size_t)C * 8));
        TRY(alloc(reinterpret_cast<void **>(&d_sl), (size_t)E * 8));
        TRY(alloc(reinterpret_cast<void **>(&d_thr), (size_t)g->num_thresholds * 8));
        TRY(alloc(reinterpret_cast<void **>(&d_ns), (size_t)E * 4));
        hipError_t er = hipMemcpy(d_cf, g->channel_center_frequency_hz, (size_t)C * 8, hipMemcpyHostToDevice);
        if (er == hipSuccess) er = hipMemcpy(d_sl, g->link_span_length_km, (size_t)E * 8, hipMemcpyHostToDevice);
        if (er == hipSuccess) er = hipMemcpy(d_thr, g->thresholds_db, (size_t)g->num_thresholds * 8, hipMemcpyHostToDevice);
        if (er == hipSuccess) er = hipMemcpy(d_ns, g->link_num_spans, (size_t)E * 4, hipMemcpyHostToDevice);
        if (er != hipSuccess) { orlg_phy_destroy(e); return fail(ORLG_ERR_HIP, "upload of the GN gate tables: %s", hipGetErrorString(er)); }
        p.gn_on = 1; p.gn_nthr = g->num_thresholds;
        p.gn_pw = g->launch_power_w; p.gn_bw = g->channel_bandwidth_hz; p.gn_att = g->attenuation_normalized; p.gn_nf = g->noise_figure;
        p.gn_cf = d_cf; p.gn_nspans = d_ns; p.gn_spanlen = d_sl; p.gn_thr = d_thr;
        // the table-only part of a check (two asinh per channel pair, two exp per link), evaluated once on the device
        double *d_A = nullptr, *d_R = nullptr, *d_L = nullptr;
        TRY(alloc(reinterpret_cast<void **>(&d_A), (size_t)C * p.cpad * 8));
        TRY(alloc(reinterpret_cast<void **>(&d_R), (size_t)C * p.cpad * 8));
        TRY(alloc(reinterpret_cast<void **>(&d_L), (size_t)(4 * E + 4) * 8));
        hipLaunchKernelGGL(orlg_gn_tables_kernel, dim3(256), dim3(256), 0, e->stream, p, d_A, d_R, d_L);
        er = hipGetLastError();
        if (er == hipSuccess) er = hipStreamSynchronize(e->stream);
        if (er != hipSuccess) { orlg_phy_destroy(e); return fail(ORLG_ERR_HIP, "GN gate tables: %s", hipGetErrorString(er)); }
        p.gn_A = d_A; p.gn_R = d_R; p.gn_link = d_L;
    }
    if (p.use_masks) TRY(alloc(reinterpret_cast<void **>(&p.cterm), (size_t)batch * p.cpad * sizeof(double)));   // (scratch, see OrlgPhyParams)
    if (p.use_masks) {   // the deferred channel-order sums (MetricCache): scratch too
        TRY(alloc(reinterpret_cast<void **>(&p.rlog_t0), (size_t)batch * p.cpad * sizeof(double)));
        TRY(alloc(reinterpret_cast<void **>(&p.rlog_val), (size_t)batch * ORLG_RLOG_CAP * sizeof(double)));
        TRY(alloc(reinterpret_cast<void **>(&p.rlog_key), (size_t)batch * ORLG_RLOG_CAP * sizeof(uint32_t)));
    }
    if (p.defrag_period > 0) {
        // defragmentation work list: one entry per channel in use that a service fills (candidates of the physical pass)
        p.cand_cap = c->defrag_capacity > 0 ? c->defrag_capacity : 2 * Q;
        if (Q > 65535) { orlg_phy_destroy(e); return fail(ORLG_ERR_INVALID, "defragmentation needs queue_capacity < 65536"); }
        TRY(alloc(reinterpret_cast<void **>(&p.cand), (size_t)batch * p.cand_cap * sizeof(OrlgPhyCand)));
        // side arrays of the service records (the scans of the defragmentation walk these): part of the state
        if (t->num_paths >= 16384) { orlg_phy_destroy(e); return fail(ORLG_ERR_INVALID, "defragmentation needs fewer than 16384 path records"); }
        TRY(alloc(reinterpret_cast<void **>(&p.qsum), (size_t)batch * Q * sizeof(uint64_t)));
        TRY(alloc(reinterpret_cast<void **>(&p.qseq), (size_t)batch * Q * sizeof(uint32_t)));
        hipError_t em = hipMemset(p.qsum, 0, (size_t)batch * Q * sizeof(uint64_t));
        if (em == hipSuccess) em = hipMemset(p.qseq, 0, (size_t)batch * Q * sizeof(uint32_t));
        if (em != hipSuccess) { orlg_phy_destroy(e); return fail(ORLG_ERR_HIP, "hipMemset: %s", hipGetErrorString(em)); }
    }
    {
        std::vector<uint32_t> mt((size_t)batch * ORLG_MT_N);
        for (int i = 0; i < batch; i++) orlg_mt_seed(&mt[(size_t)i * ORLG_MT_N], seeds ? seeds[i] : base_seed + (uint64_t)i);
        hipError_t er = hipMemcpy(p.mt, mt.data(), mt.size() * 4, hipMemcpyHostToDevice);
        if (er != hipSuccess) { orlg_phy_destroy(e); return fail(ORLG_ERR_HIP, "upload of MT19937 states: %s", hipGetErrorString(er)); }
    }
    hipLaunchKernelGGL(orlg_phy_clear_kernel, dim3(512), dim3(256), 0, e->stream, p, W, 0);
    OrlgPhyParams pi = p;
    pi.mode = ORLG_MODE_INIT; pi.n_steps = 1;
    TRY(phy_launch(e, pi));
    {
        hipError_t er = hipStreamSynchronize(e->stream);
        if (er != hipSuccess) { orlg_phy_destroy(e); return fail(ORLG_ERR_HIP, "initial reset: %s", hipGetErrorString(er)); }
    }
#undef TRY
    *out = e;
    return ORLG_OK;
}

int orlg_phy_set_stream(orlg_phy_env *e, void *hip_stream) {
    if (!e) return fail(ORLG_ERR_INVALID, "null handle");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (e->own_stream) { HIP_TRY(hipStreamDestroy(e->stream)); e->own_stream = false; }
    if (hip_stream) {
        e->stream = static_cast<hipStream_t>(hip_stream);
    } else {
        HIP_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
        e->own_stream = true;
    }
    return ORLG_OK;
}

int orlg_phy_reseed(orlg_phy_env *e, const uint64_t *seeds, uint64_t base_seed) {
    if (!e) return fail(ORLG_ERR_INVALID, "null handle");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    const int B = e->p.B;
    std::vector<uint32_t> mt((size_t)B * ORLG_MT_N);
    for (int i = 0; i < B; i++) orlg_mt_seed(&mt[(size_t)i * ORLG_MT_N], seeds ? seeds[i] : base_seed + (uint64_t)i);
    HIP_TRY(hipMemcpyAsync(e->p.mt, mt.data(), mt.size() * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(orlg_phy_reseed_kernel, dim3(64), dim3(256), 0, e->stream, e->p.scal, B);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(e->stream));
    return ORLG_OK;
}

int orlg_phy_reset(orlg_phy_env *e, int32_t only_episode_counters) {
    if (!e) return fail(ORLG_ERR_INVALID, "null handle");
    HIP_TRY(hipSetDevice(e->device));
    OrlgPhyParams p = e->p;
    p.n_steps = 1;
    if (only_episode_counters) {
        p.mode = ORLG_MODE_EPISODE_RESET;
    } else {
        HIP_TRY(hipStreamSynchronize(e->stream));   // a full reset also clears a reported overflow
        *e->err.host = 0;
        hipLaunchKernelGGL(orlg_phy_clear_kernel, dim3(512), dim3(256), 0, e->stream, e->p, e->W, 1);
        HIP_TRY(hipGetLastError());
        p.mode = ORLG_MODE_INIT;
    }
    return phy_launch(e, p);
}

int orlg_phy_step(orlg_phy_env *e, int32_t policy, int32_t n_steps, const int32_t *act_path, const int16_t *act_channels,
                  int32_t auto_reset, const orlg_phy_step_io *io) {
    if (!e) return fail(ORLG_ERR_INVALID, "null handle");
    if (n_steps < 1) return fail(ORLG_ERR_INVALID, "n_steps must be >= 1");
    if (policy < ORLG_PHY_POLICY_EXTERNAL || policy > ORLG_PHY_POLICY_FAFF_RSS) return fail(ORLG_ERR_INVALID, "unknown PhyRMSA policy %d", policy);
    if (policy == ORLG_PHY_POLICY_EXTERNAL && (!act_path || !act_channels || n_steps != 1))
        return fail(ORLG_ERR_INVALID, "external actions need path and channel arrays and n_steps == 1");
    HIP_TRY(hipSetDevice(e->device));
    OrlgPhyParams p = e->p;
    p.mode = ORLG_MODE_STEP; p.n_steps = n_steps; p.policy = policy; p.auto_reset = auto_reset;
    if (policy == ORLG_PHY_POLICY_EXTERNAL) {
        if (orlg_is_device_ptr(act_path) && orlg_is_device_ptr(act_channels)) {
            p.act_path = act_path; p.act_channels = act_channels;
        } else {
            if (!e->d_act_path) {
                HIP_TRY(hipMalloc(reinterpret_cast<void **>(&e->d_act_path), (size_t)p.B * 4));
                HIP_TRY(hipMalloc(reinterpret_cast<void **>(&e->d_act_ch), (size_t)p.B * ORLG_PHY_MAX_CH * 2));
            }
            HIP_TRY(hipMemcpyAsync(e->d_act_path, act_path, (size_t)p.B * 4, hipMemcpyDefault, e->stream));
            HIP_TRY(hipMemcpyAsync(e->d_act_ch, act_channels, (size_t)p.B * ORLG_PHY_MAX_CH * 2, hipMemcpyDefault, e->stream));
            p.act_path = e->d_act_path; p.act_channels = e->d_act_ch;
        }
    }
    struct Slot { void *user; size_t elem; };
    const size_t cnt = (size_t)n_steps * p.B;
    Slot slots[ORLG_PHY_NUM_OUTS] = {
        {io ? io->act_path : nullptr, 4},  {io ? io->n_channels : nullptr, 4}, {io ? io->channels : nullptr, 2 * ORLG_PHY_MAX_CH},
        {io ? io->accepted : nullptr, 1},  {io ? io->done : nullptr, 1},       {io ? io->request : nullptr, 16},
        {io ? io->arrival : nullptr, 8},   {io ? io->holding : nullptr, 8},    {io ? io->number_cuts_total : nullptr, 8},
        {io ? io->rss_total_metric : nullptr, 8}, {io ? io->channels_used : nullptr, 2 * ORLG_PHY_MAX_CH},
        {io ? io->defrag_counters : nullptr, 12}, {io ? io->gn_gsnr_db : nullptr, 8}};
    bool staged[ORLG_PHY_NUM_OUTS] = {false};
    p.out_mask = 0;
    for (int i = 0; i < ORLG_PHY_NUM_OUTS; i++) {
        p.outs[i] = nullptr;
        if (!slots[i].user) continue;
        p.out_mask |= 1 << i;
        if (orlg_is_device_ptr(slots[i].user)) {
            p.outs[i] = slots[i].user;
        } else {
            size_t bytes = cnt * slots[i].elem;
            if (bytes > e->io_cap[i]) {
                if (e->io_buf[i]) HIP_TRY(hipFree(e->io_buf[i]));
                e->io_buf[i] = nullptr; e->io_cap[i] = 0;
                HIP_TRY(hipMalloc(&e->io_buf[i], bytes));
                e->io_cap[i] = bytes;
            }
            p.outs[i] = e->io_buf[i];
            staged[i] = true;
        }
    }
    int rc = phy_launch(e, p);
    if (rc) return rc;
    bool any = false;
    for (int i = 0; i < ORLG_PHY_NUM_OUTS; i++)
        if (staged[i]) {
            HIP_TRY(hipMemcpyAsync(slots[i].user, e->io_buf[i], cnt * slots[i].elem, hipMemcpyDeviceToHost, e->stream));
            any = true;
        }
    if (any) return phy_sync_check(e);
    return ORLG_OK;
}

int orlg_phy_synchronize(orlg_phy_env *e) {
    if (!e) return fail(ORLG_ERR_INVALID, "null handle");
    HIP_TRY(hipSetDevice(e->device));
    return phy_sync_check(e);
}

int orlg_phy_last_kernel(orlg_phy_env *e, char *buf, int32_t cap) {
    if (!e || !buf || cap < 1) return fail(ORLG_ERR_INVALID, "null argument");
    snprintf(buf, (size_t)cap, "%s", e->last_kernel);
    return ORLG_OK;
}
int orlg_phy_words_per_link(orlg_phy_env *e) { return e ? e->W : ORLG_ERR_INVALID; }
int orlg_phy_node_vectors(orlg_phy_env *e) { return e ? e->p.use_nv : ORLG_ERR_INVALID; }
int orlg_phy_get_requests(orlg_phy_env *e, orlg_request *out) {
    if (!e || !out) return fail(ORLG_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    return phy_extract(e, PX_REQUEST, out, (size_t)e->p.B * sizeof(orlg_request));
}
int orlg_phy_get_counters(orlg_phy_env *e, orlg_counters *out) {
    if (!e || !out) return fail(ORLG_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    return phy_extract(e, PX_COUNTERS, out, (size_t)e->p.B * sizeof(orlg_counters));
}
int orlg_phy_get_current_time(orlg_phy_env *e, double *out) {
    if (!e || !out) return fail(ORLG_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    return phy_extract(e, PX_TIME, out, (size_t)e->p.B * 8);
}
int orlg_phy_get_num_running(orlg_phy_env *e, int32_t *out) {
    if (!e || !out) return fail(ORLG_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    return phy_extract(e, PX_RUNNING, out, (size_t)e->p.B * 4);
}
int orlg_phy_get_episode_stats(orlg_phy_env *e, orlg_phy_episode_stats *out) {
    if (!e || !out) return fail(ORLG_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    return phy_extract(e, PX_EPISODE, out, (size_t)e->p.B * sizeof(orlg_phy_episode_stats));
}
int orlg_phy_get_occupancy(orlg_phy_env *e, uint64_t *out) {
    if (!e || !out) return fail(ORLG_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipMemcpyAsync(out, e->p.occ, (size_t)e->p.B * e->p.NW * 8, hipMemcpyDefault, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return ORLG_OK;
}
typedef OrlgStatePart StatePart;
static std::vector<StatePart> phy_state_parts(orlg_phy_env *e) {
    const OrlgPhyParams &p = e->p;
    const size_t B = p.B, lists = (size_t)p.N * p.N * p.K;
    std::vector<StatePart> parts = {{p.occ, B * p.NW * 8}, {p.qtime, B * p.Q * 8}, {p.qrec, B * p.Q * sizeof(OrlgPhySvc)}, {p.mt, B * ORLG_MT_N * 4},
            {p.scal, B * sizeof(OrlgPhyScalars)}, {p.cs, B * lists * p.cs_len * 4}, {p.cs_n, B * lists},
            {p.ring_iat, B * ORLG_RING * 8}, {p.ring_ht, B * ORLG_RING * 8}, {p.ring_req, B * ORLG_RING * 4}};
    if (p.qsum) { parts.push_back({p.qsum, B * p.Q * sizeof(uint64_t)}); parts.push_back({p.qseq, B * p.Q * sizeof(uint32_t)}); }
    return parts;
}
int64_t orlg_phy_state_size(orlg_phy_env *e) {
    if (!e) return fail(ORLG_ERR_INVALID, "null handle");
    int64_t n = 0;
    for (const StatePart &sp : phy_state_parts(e)) n += (int64_t)sp.bytes;
    return n;
}
int orlg_phy_save_state(orlg_phy_env *e, void *buffer) {
    if (!e || !buffer) return fail(ORLG_ERR_INVALID, "null argument");
    return orlg_state_copy(phy_state_parts(e), buffer, true, e->device, e->stream);
}
int orlg_phy_load_state(orlg_phy_env *e, const void *buffer) {
    if (!e || !buffer) return fail(ORLG_ERR_INVALID, "null argument");
    int rc = orlg_state_copy(phy_state_parts(e), const_cast<void *>(buffer), false, e->device, e->stream);
    if (rc) return rc;
    *e->err.host = 0;   // as orlg_load_state: the error word follows the loaded state
    hipLaunchKernelGGL(orlg_phy_overflow_store_kernel, dim3(64), dim3(256), 0, e->stream, e->p.scal, e->p.B, e->err.dev);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(e->stream));
    return ORLG_OK;
}
int orlg_phy_channel_state_capacity(orlg_phy_env *e) { return e ? e->p.cs_len : ORLG_ERR_INVALID; }
int orlg_phy_get_channel_state(orlg_phy_env *e, int32_t env_index, uint32_t *entries, uint8_t *lengths) {
    if (!e || !entries || !lengths) return fail(ORLG_ERR_INVALID, "null argument");
    if (env_index < 0 || env_index >= e->p.B) return fail(ORLG_ERR_INVALID, "env_index %d out of range", env_index);
    HIP_TRY(hipSetDevice(e->device));
    const size_t lists = (size_t)e->p.N * e->p.N * e->p.K;
    HIP_TRY(hipMemcpyAsync(entries, e->p.cs + (size_t)env_index * lists * e->p.cs_len, lists * e->p.cs_len * 4, hipMemcpyDefault, e->stream));
    HIP_TRY(hipMemcpyAsync(lengths, e->p.cs_n + (size_t)env_index * lists, lists, hipMemcpyDefault, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return e->p.cs_len;
}
int orlg_phy_reduce_counters(orlg_phy_env *e, int64_t *out) {
    if (!e || !out) return fail(ORLG_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    if (e->staging_bytes < 16 * 8) {
        if (e->staging) HIP_TRY(hipFree(e->staging));
        e->staging = nullptr;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&e->staging), 1024));
        e->staging_bytes = 1024;
    }
    long long *d = reinterpret_cast<long long *>(e->staging);
    hipLaunchKernelGGL(orlg_phy_reduce_kernel, dim3(1), dim3(256), 0, e->stream, e->p.scal, e->p.B, d);
    HIP_TRY(hipGetLastError());
    long long host[16];
    HIP_TRY(hipMemcpyAsync(host, d, sizeof(host), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(out, d, 16 * 8, hipMemcpyDefault, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (host[10]) return fail(ORLG_ERR_QUEUE_FULL, "a release queue (capacity %d), channel_state list (capacity %d) or defragmentation work list (capacity %d) overflowed: raise queue_capacity / channel_state_capacity / defrag_capacity", e->p.Q, e->p.cs_len, e->p.cand_cap);
    return ORLG_OK;
}

}  // extern "C"
