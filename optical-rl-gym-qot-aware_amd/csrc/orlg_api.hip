// orlg_api.hip -- host side of liborlg.so: the C ABI declared in include/orlg.h.
// Owns the device state of B environments, builds the read-only tables, seeds MT19937 the way
// random.Random(int) does and launches the kernels of orlg_kernels.hip.  No CPU compute path exists:
// every entry point that touches environments needs a HIP device.
#include "orlg_host.h"
#include "orlg_kernels.hip"   // data layout + device helpers; the step kernels are instantiated in orlg_inst_*.hip

// ---------------------------------------------------------------------------------------- errors
static thread_local std::string g_err;
int orlg_fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

int orlg_err_word_create(OrlgErrWord *w) {
    void *h = nullptr, *d = nullptr;
    w->host = nullptr; w->dev = nullptr;
    HIP_TRY(hipHostMalloc(&h, 64, hipHostMallocMapped));
    memset(h, 0, 64);
    hipError_t er = hipHostGetDevicePointer(&d, h, 0);
    if (er != hipSuccess) { (void)hipHostFree(h); return fail(ORLG_ERR_HIP, "hipHostGetDevicePointer: %s", hipGetErrorString(er)); }
    w->host = static_cast<volatile int32_t *>(h);
    w->dev = static_cast<int32_t *>(d);
    return ORLG_OK;
}
void orlg_err_word_destroy(OrlgErrWord *w) {
    if (w->host) (void)hipHostFree(const_cast<int32_t *>(w->host));
    w->host = nullptr; w->dev = nullptr;
}

// ---------------------------------------------------------------------------------------- handle
struct orlg_env {
    OrlgParams p;
    int W;
    int device;
    hipStream_t stream;
    bool own_stream;
    size_t lds_block_bytes;
    int waves_per_block;
    int resident_blocks;   // workgroups of the step kernel the device keeps resident (grid size of the work queue)
    // four-environments-per-wave step kernel: workgroup shape, LDS bytes, resident workgroups; group_mode = ORLG_KERNEL_*
    // (AUTO falls back to WAVE when the shape does not fit the kernel's LDS budget)
    int group_mode, group_wpb;   // group_wpb: the most waves per workgroup the LDS holds
    int group_resident[ORLG_GROUP_WAVES + 1];   // resident workgroups by waves per workgroup (0 = not asked yet)
    int group_wpb_hq, group_wave_bytes_hq;      // the same for launches that leave the release queue in HBM (orlg_rmsa_group_kernel<.., true>)
    int group_resident_hq[ORLG_GROUP_WAVES + 1];
    int group_resident_df[ORLG_GROUP_WAVES + 1];   // ... of the instantiation with the link statistics deferred
    int group_df_lint, group_df_qtime, group_df_qdesc, group_df_wave_bytes, group_df_wpb;   // its LDS layout (no link-statistics slices)
    uint4 *llog;             // its log of link updates [B][E][64] (allocated with the first such launch)
    uint32_t *progress;      // chunked tickets: chunks completed per quad in the current launch (allocated with the first such launch)
    size_t group_lds_bytes;
    int num_cu;
    uint32_t ticket_base;
    int num_paths;
    // owned device buffers
    std::vector<void *> bufs;
    unsigned char *staging;
    size_t staging_bytes;
    // lazily grown io buffers
    void *io_buf[12];
    size_t io_cap[12];
    int32_t *d_actions;
    size_t d_actions_cap;
    OrlgErrWord err;         // sticky error word the kernels set on a release-queue overflow
    char last_kernel[96];    // name and shape of the kernel behind the last step / reset launch (orlg_last_kernel)
};

// wait for the handle's stream, then report an overflow a kernel flagged (ORLG_ERR_QUEUE_FULL is sticky until a full reset)
static int sync_check(orlg_env *e) {
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (e->err.host && *e->err.host)
        return fail(ORLG_ERR_QUEUE_FULL, "a release queue overflowed (capacity %d): the service stays provisioned and is never "
                                         "released -- raise queue_capacity", e->p.Q);
    return ORLG_OK;
}
#define SYNC_CHECK(e) do { int rc_ = sync_check(e); if (rc_) return rc_; } while (0)

template <typename T>
static int dev_alloc(orlg_env *e, T **out, size_t count) {
    void *ptr = nullptr;
    HIP_TRY(hipMalloc(&ptr, count * sizeof(T) > 0 ? count * sizeof(T) : 16));
    e->bufs.push_back(ptr);
    *out = static_cast<T *>(ptr);
    return ORLG_OK;
}
template <typename T>
static int dev_upload(orlg_env *e, T **out, const T *host, size_t count) {
    int rc = dev_alloc(e, out, count);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(*out, host, count * sizeof(T), hipMemcpyHostToDevice));
    return ORLG_OK;
}

// ---------------------------------------------------------------------------------------- MT19937 seeding
// CPython _randommodule.c: random.Random(n) -> init_by_array(32-bit little-endian chunks of abs(n)).
void orlg_mt_seed(uint32_t *mt, uint64_t seed) {
    uint32_t key[2] = {(uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32)};
    int len = key[1] ? 2 : 1;
    mt[0] = 19650218u;
    for (int i = 1; i < ORLG_MT_N; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    int i = 1, j = 0;
    for (int k = ORLG_MT_N > len ? ORLG_MT_N : len; k; k--) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        i++; j++;
        if (i >= ORLG_MT_N) { mt[0] = mt[ORLG_MT_N - 1]; i = 1; }
        if (j >= len) j = 0;
    }
    for (int k = ORLG_MT_N - 1; k; k--) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        i++;
        if (i >= ORLG_MT_N) { mt[0] = mt[ORLG_MT_N - 1]; i = 1; }
    }
    mt[0] = 0x80000000u;
}

// ---------------------------------------------------------------------------------------- small kernels
// reset(only_episode_counters=False) state: every slot free, empty queue, zero counters and statistics;
// the RNG state (mt, mt_idx) is deliberately kept (optical_network_env.py:216-264, rmsa_env.py:391-455).
__global__ void orlg_clear_state_kernel(OrlgParams p, int W, int keep_rng) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
    for (size_t i = tid; i < (size_t)p.B * p.NW; i += nth) p.occ[i] = valid_mask(p.S, (int)(i % W));
    for (size_t i = tid; i < (size_t)p.B * p.Q; i += nth) {
        p.qtime[i] = __longlong_as_double((long long)ORLG_INF_BITS);
        p.qdesc[i] = 0;
    }
    for (size_t i = tid; i < (size_t)p.B * 4 * p.NBR; i += nth) p.hist[i] = 0;
    for (size_t i = tid; i < (size_t)p.B * 4 * p.E; i += nth) p.lstat[i] = 0.0;
    for (size_t i = tid; i < (size_t)p.B * p.lint_stride; i += nth) p.lint[i] = 0;
    for (size_t i = tid; i < (size_t)p.B; i += nth) {
        OrlgEnvScalars s;
        memset(&s, 0, sizeof(s));
        s.mt_idx = keep_rng ? p.scal[i].mt_idx : ORLG_MT_N;
        // pre-generated arrivals are part of the RNG stream: a full reset keeps them
        s.ring_pos = keep_rng ? p.scal[i].ring_pos : 0;
        s.ring_cnt = keep_rng ? p.scal[i].ring_cnt : 0;
        p.scal[i] = s;
    }
}

// SimpleMatrixObservation.observation (rmsa_env.py:952-971) for every env: [B][2N + E*S] uint8 = one-hot of the lower
// and of the higher endpoint index, then the free-slot flags link-major (the bitmap unpacked).
__global__ __launch_bounds__(256) void orlg_simple_matrix_obs_kernel(const OrlgParams p, int W, uint8_t *out) {
    const size_t dim = (size_t)2 * p.N + (size_t)p.E * p.S;
    const size_t total = dim * p.B;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / dim, q = i - b * dim;
        uint8_t v;
        if (q < (size_t)2 * p.N) {
            const OrlgEnvScalars *sc = p.scal + b;
            const int mn = sc->req_src < sc->req_dst ? sc->req_src : sc->req_dst;
            const int mx = sc->req_src < sc->req_dst ? sc->req_dst : sc->req_src;
            v = ((int)q == mn || (int)q == p.N + mx) ? 1 : 0;
        } else {
            const size_t r = q - 2 * p.N;
            const int link = (int)(r / p.S), s = (int)(r - (size_t)link * p.S);
            v = (uint8_t)((p.occ[b * p.NW + (size_t)link * W + (s >> 6)] >> (s & 63)) & 1ull);
        }
        out[i] = v;
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

enum { EX_REQUEST, EX_COUNTERS, EX_TIME, EX_GRAPH, EX_RUNNING, EX_EPISODES, EX_HIST, EX_LSTAT };
__global__ void orlg_extract_kernel(OrlgParams p, int what, unsigned char *out) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
    const size_t B = p.B;
    if (what == EX_REQUEST) {
        orlg_request *o = reinterpret_cast<orlg_request *>(out);
        for (size_t i = tid; i < B; i += nth) {
            const OrlgEnvScalars &s = p.scal[i];
            o[i].service_id = s.req_sid; o[i].src = s.req_src; o[i].dst = s.req_dst;
            o[i].bit_rate = reinterpret_cast<const int32_t *>(p.tables + p.t_bitrates)[s.req_br];
            o[i].arrival_time = s.req_arrival; o[i].holding_time = s.req_holding;
        }
    } else if (what == EX_COUNTERS) {
        int64_t *o = reinterpret_cast<int64_t *>(out);
        for (size_t i = tid; i < B * 8; i += nth) o[i] = p.scal[i / 8].c[i % 8];
    } else if (what == EX_TIME) {
        double *o = reinterpret_cast<double *>(out);
        for (size_t i = tid; i < B; i += nth) o[i] = p.scal[i].current_time;
    } else if (what == EX_GRAPH) {
        double *o = reinterpret_cast<double *>(out);
        for (size_t i = tid; i < B; i += nth) {
            o[i] = p.scal[i].g_throughput; o[B + i] = p.scal[i].g_compactness; o[2 * B + i] = p.scal[i].g_last_update;
        }
    } else if (what == EX_RUNNING) {
        int32_t *o = reinterpret_cast<int32_t *>(out);
        for (size_t i = tid; i < B; i += nth) o[i] = p.scal[i].n_running;
    } else if (what == EX_EPISODES) {
        int64_t *o = reinterpret_cast<int64_t *>(out);
        for (size_t i = tid; i < B; i += nth) o[i] = p.scal[i].episodes_done;
    } else if (what == EX_HIST) {  // [4][B][NBR] int64 from [B][4][NBR] int32
        int64_t *o = reinterpret_cast<int64_t *>(out);
        const size_t n = (size_t)p.NBR;
        for (size_t i = tid; i < 4 * B * n; i += nth) {
            size_t k = i / (B * n), r = i % (B * n), b = r / n, q = r % n;
            o[i] = p.hist[(b * 4 + k) * n + q];
        }
    } else if (what == EX_LSTAT) {  // [4][B][E] from [B][4][E]
        double *o = reinterpret_cast<double *>(out);
        const size_t n = (size_t)p.E;
        for (size_t i = tid; i < 4 * B * n; i += nth) {
            size_t k = i / (B * n), r = i % (B * n), b = r / n, q = r % n;
            o[i] = p.lstat[(b * 4 + k) * n + q];
        }
    }
}

__global__ void orlg_overflow_kernel(const OrlgEnvScalars *scal, int B, int *out) {
    int any = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B; i += gridDim.x * blockDim.x) any |= scal[i].q_overflow;
    if (any) atomicOr(out, 1);
}

// the sticky error word (mapped host memory: a plain store, no atomic across the bus) recomputed from the scalars
__global__ void orlg_overflow_store_kernel(const OrlgEnvScalars *scal, int B, int *err_flag) {
    int any = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B; i += gridDim.x * blockDim.x) any |= scal[i].q_overflow;
    if (any) *err_flag = 1;
}

// ---------------------------------------------------------------------------------------- kernel dispatch
// the kernels live in per-W objects (orlg_inst_wave.hip / orlg_inst_group.hip); a W the library was not built for is a null symbol
typedef orlg_rmsa_kernel_t rmsa_kernel_t;
typedef orlg_masks_kernel_t masks_kernel_t;
static rmsa_kernel_t pick_wave(int W, int kind, int stats) {
    switch (W) {
#define X(n) case n: return orlg_wave_kernel_W##n ? orlg_wave_kernel_W##n(kind, stats) : nullptr;
        ORLG_FOR_EACH_W(X)
#undef X
        default: return nullptr;
    }
}
// the wave-per-environment step kernel that only carries the first-fit policies (k <= 8)
static rmsa_kernel_t pick_rmsa_ff(int W, int stats) { return pick_wave(W, ORLG_KIND_STEP_FF, stats); }
static rmsa_kernel_t pick_rmsa(int W, int stats, bool step = true) { return pick_wave(W, step ? ORLG_KIND_STEP : ORLG_KIND_RESET, stats); }
static rmsa_kernel_t pick_obs(int W) { return pick_wave(W, ORLG_KIND_OBS, 0); }
static rmsa_kernel_t pick_group(int W, int stats) {
    switch (W) {
#define X(n) case n: return orlg_group_kernel_W##n ? orlg_group_kernel_W##n(stats) : nullptr;
        ORLG_FOR_EACH_W(X)
#undef X
        default: return nullptr;
    }
}
static masks_kernel_t pick_masks(int W) {
    switch (W) {
#define X(n) case n: return orlg_masks_kernel_W##n ? orlg_masks_kernel_W##n() : nullptr;
        ORLG_FOR_EACH_W(X)
#undef X
        default: return nullptr;
    }
}

// the step kernel with four environments per wave: first-fit policies and external (path, slot) actions
static int launch_rmsa_group(orlg_env *e, const OrlgParams &p) {
    // launches of very few steps leave the release queue in HBM (the kernel's HBMQ instantiation): without the queue's slices an
    // environment takes half the LDS, and such a launch is bound by the waves a CU keeps resident
    const bool hq = p.n_steps <= ORLG_DIRECT_STEPS && e->group_wpb_hq > e->group_wpb;
    const bool df_ok = !hq && p.stats_level >= 2 && p.n_steps >= 16 && !getenv("ORLG_NO_DEFER") && e->group_df_wpb >= 1 &&
                       !(p.out_mask & ((1 << ORLG_OUT_AVG_LINK_COMPACT) | (1 << ORLG_OUT_AVG_LINK_UTIL)));
    const int wave_bytes = hq ? e->group_wave_bytes_hq : df_ok ? e->group_df_wave_bytes : p.g_wave_bytes;
    const int wpb_max = hq ? e->group_wpb_hq : df_ok ? e->group_df_wpb : e->group_wpb;
    // long launches with full statistics whose outputs do not read the link statistics step by step: the instantiation that
    // logs the links' updates and works them off one link per lane (group_link_replay)
    const bool df = df_ok;
    int *resident = hq ? e->group_resident_hq : df ? e->group_resident_df : e->group_resident;
    rmsa_kernel_t k = pick_group(e->W, p.stats_level + (hq ? 4 : df ? 8 : 0));
    if (!k) return fail(ORLG_ERR_INVALID, "no kernel for W=%d", e->W);
    if (df && !e->llog) {
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&e->llog), (size_t)p.B * p.E * 64 * sizeof(uint4)));
        e->bufs.push_back(e->llog);
    }
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)((size_t)p.l_shared_bytes + p.g_mt + 16 + (size_t)wpb_max * wave_bytes)));
    const int n_quads = (p.B + 3) / 4;
    // Waves per workgroup: as many as the LDS holds when the batch keeps every CU busy for several rounds (more resident waves
    // per SIMD hide more latency); fewer when that would leave CUs idle or the last round mostly empty.  A round of w waves per
    // CU costs about w + 1.5 (measured: 10 waves per CU step 3 % more environments per second than 8); few rounds count whole.
    int wpb = wpb_max;
    // (long launches of batches beyond one round of the full workgroup keep it: their rounds are evened out by tickets in chunks
    // of steps, below -- the model here would trade resident waves for whole rounds)
    const bool long_rounds = p.n_steps >= 256 && !hq && n_quads >= wpb_max * e->num_cu && !getenv("ORLG_NO_CHUNKS");
    if (!long_rounds) {
        double best = 1e300;
        for (int w = wpb_max; w >= 1; --w) {
            const double rounds = (double)n_quads / ((double)e->num_cu * w);
            // (short launches stride statically over the quads: whole rounds; long ones draw tickets: the last round is partial)
            const double cost = ((rounds < 3.0 || p.n_steps <= 16) ? std::ceil(rounds) : rounds + 0.5) * (w + 1.5);
            if (cost < best - 1e-9) { best = cost; wpb = w; }
        }
    }
    if (const char *ov = getenv("ORLG_GROUP_WPB")) {   // tooling override: waves per workgroup
        const int v = atoi(ov);
        if (v >= 1 && v <= wpb_max) wpb = v;
    }
    const size_t lds_bytes = (size_t)p.l_shared_bytes + p.g_mt + 16 + (size_t)wpb * wave_bytes;
    if (resident[wpb] <= 0) {
        int nb = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(k), ORLG_WAVE * wpb, lds_bytes));
        resident[wpb] = (nb > 0 ? nb : 1) * e->num_cu;
    }
    int nblocks = (n_quads + wpb - 1) / wpb;
    if (nblocks > resident[wpb]) nblocks = resident[wpb];
    OrlgParams q = p;
    q.llog = df ? e->llog : nullptr;
    if (df) { q.g_lint = e->group_df_lint; q.g_qtime = e->group_df_qtime; q.g_qdesc = e->group_df_qdesc; }
    q.g_wave_bytes = wave_bytes;
    q.ticket_base = e->ticket_base;
    q.ticket_stride = p.n_steps <= 16 ? 1u : 0u;
    // Tickets in chunks of steps (orlg_rmsa_group_kernel, work queue): when the batch is not a multiple of the resident waves, a
    // launch's last round of whole-launch tickets runs at a fraction of the occupancy for a whole launch's time (B = 65 536 on 3072
    // wave slots: 5.33 rounds, the last one 1/3 full and nearly as long as a full one).  With k chunks per quad the tail is one
    // chunk long; a hand-off between waves costs a few microseconds (agent-scope release + acquire) against milliseconds of steps.
    q.n_chunks = 1; q.chunk_steps = p.n_steps; q.progress = nullptr;
    {
        const int slots = nblocks * wpb;
        int k = 1;
        if (long_rounds && !q.ticket_stride && n_quads > slots) {   // (exactly one round: 1 116 with chunks against 1 131-1 140 M)
            // Measured (NSFNET-320, 1000-step launches, M env-steps/s by chunks k = 1 / 2 / 3 / 4; r = quads / slots rounds):
            //   B = 16 384 (r = 1.33):   853 / 1 090 / 1 131 / 1 178      B = 49 152 (r = 4):    1 151 / 1 250 / 1 212 / 1 240
            //   B = 24 576 (r = 2):    1 139 / 1 136 / 1 234 / 1 234      B = 65 536 (r = 5.33): 1 206 / 1 259 / 1 249 / 1 235
            //   B = 131 072 (r = 10.7): 1 282 with k = 1, 1 257 with k = 3: after many rounds the waves' finishing times have
            //   spread and the last round is short by itself.
            // A chunk boundary costs a quad ~0.55 % of a 1000-step launch (state store + load, release + acquire); whole rounds
            // (r = 2, 4) gain as well: waves that start together stay in step -- all in the same refill at the same time -- and
            // chunks of different quads break that up.  About eight rounds of tickets are enough:
            k = (int)std::floor(8.0 * slots / n_quads + 0.5);
            k = k < 1 ? 1 : (k > 4 ? 4 : k);
            while (k > 1 && p.n_steps / k < 128) --k;   // (a boundary costs the same whatever the chunk's length)
        }
        if (const char *ov = getenv("ORLG_GROUP_CHUNKS")) {   // tooling / tests: force the number of chunks (any batch)
            const int v = atoi(ov);
            if (v >= 1 && v <= 64 && !q.ticket_stride && !hq) k = v < p.n_steps ? v : p.n_steps;
        }
        if (k > 1) {
            if (!e->progress) {
                HIP_TRY(hipMalloc(reinterpret_cast<void **>(&e->progress), ((size_t)p.B / 4 + 1) * sizeof(uint32_t)));
                e->bufs.push_back(e->progress);
            }
            HIP_TRY(hipMemsetAsync(e->progress, 0, (size_t)n_quads * sizeof(uint32_t), e->stream));
            q.chunk_steps = (p.n_steps + k - 1) / k;
            q.n_chunks = (p.n_steps + q.chunk_steps - 1) / q.chunk_steps;
            q.progress = e->progress;
        }
    }
    // one draw per ticket a wave takes on (the next one is drawn when a ticket is taken up); with chunks every wave draws its first
    // ticket as well
    if (!q.ticket_stride) e->ticket_base += (uint32_t)n_quads * (uint32_t)q.n_chunks + (q.n_chunks > 1 ? (uint32_t)(nblocks * wpb) : 0u);
    dim3 grid(nblocks), block(ORLG_WAVE * wpb);
    hipLaunchKernelGGL(k, grid, block, lds_bytes, e->stream, q);
    HIP_TRY(hipGetLastError());
    snprintf(e->last_kernel, sizeof(e->last_kernel), "orlg_rmsa_group_kernel<%d,%d%s> grid=%d block=%d lds=%zu chunks=%d", e->W, p.stats_level,
             hq ? ",true" : df ? ",false,true" : "", nblocks, ORLG_WAVE * wpb, lds_bytes, q.n_chunks);
    return ORLG_OK;
}

static bool group_kernel_serves(const orlg_env *e, const OrlgParams &p) {
    if (e->group_mode == ORLG_KERNEL_WAVE || p.mode != ORLG_MODE_STEP) return false;
    if (e->group_mode == ORLG_KERNEL_GROUP) return true;
    // AUTO: this kernel once the batch exceeds what the wave-per-environment kernel keeps resident (4096 environments on
    // MI355X) -- long launches (B = 65 536: 1140 vs 630 M env-steps/s) and launches of one step alike (B = 32 768, DeepRMSA
    // shape: 0.145 vs 0.161 ms per step + observation); at or below it the wave-per-environment kernel has the shorter step
    // (B = 4096: 630 vs 440 M).
    return p.B > e->resident_blocks * e->waves_per_block;
}

static int launch_rmsa(orlg_env *e, const OrlgParams &p) {
    const bool ff = p.mode == ORLG_MODE_STEP && p.K <= 8 && (p.policy == ORLG_POLICY_SP || p.policy == ORLG_POLICY_SAP);
    // long launches with full statistics whose outputs do not read the link statistics step by step: the instantiation that logs the
    // links' updates and works them off one link per lane (link_replay) -- as launch_rmsa_group
    const bool df = p.mode == ORLG_MODE_STEP && p.stats_level >= 2 && p.n_steps >= 16 && !getenv("ORLG_NO_DEFER") &&
                    !(p.out_mask & ((1 << ORLG_OUT_AVG_LINK_COMPACT) | (1 << ORLG_OUT_AVG_LINK_UTIL)));
    rmsa_kernel_t k = df ? pick_wave(e->W, ff ? ORLG_KIND_STEP_FF_DF : ORLG_KIND_STEP_DF, p.stats_level)
                         : ff ? pick_rmsa_ff(e->W, p.stats_level) : pick_rmsa(e->W, p.stats_level, p.mode == ORLG_MODE_STEP);
    if (!k) return fail(ORLG_ERR_INVALID, "no kernel for W=%d", e->W);
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)e->lds_block_bytes));
    const int wpb = e->waves_per_block;
    if (e->resident_blocks <= 0) {
        int nb = 0;
        rmsa_kernel_t ks = pick_rmsa(e->W, p.stats_level, true);  // the grid is sized for the step kernel (any grid is correct)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(ks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->lds_block_bytes));
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(ks), ORLG_WAVE * wpb, e->lds_block_bytes));
        e->resident_blocks = (nb > 0 ? nb : 1) * e->num_cu;
    }
    if (group_kernel_serves(e, p)) return launch_rmsa_group(e, p);
    int nblocks = (p.B + wpb - 1) / wpb;
    if (nblocks > e->resident_blocks) nblocks = e->resident_blocks;
    if (df && !e->llog) {
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&e->llog), (size_t)p.B * p.E * 64 * sizeof(uint4)));
        e->bufs.push_back(e->llog);
    }
    OrlgParams q = p;
    q.llog = df ? e->llog : nullptr;
    q.ticket_base = e->ticket_base;
    q.ticket_stride = (p.mode != ORLG_MODE_STEP || p.n_steps <= 16) ? 1u : 0u;
    if (!q.ticket_stride) e->ticket_base += (uint32_t)p.B;  // waves * 1 static environment + (B - waves) tickets + one failing draw per wave
    dim3 grid(nblocks), block(ORLG_WAVE * wpb);
    hipLaunchKernelGGL(k, grid, block, e->lds_block_bytes, e->stream, q);
    HIP_TRY(hipGetLastError());
    snprintf(e->last_kernel, sizeof(e->last_kernel), "%s<%d,%d%s> grid=%d block=%d lds=%zu",
             ff ? "orlg_rmsa_kernel_ff" : (p.mode == ORLG_MODE_STEP ? "orlg_rmsa_kernel" : "orlg_rmsa_reset_kernel"), e->W, p.stats_level,
             df ? ",true" : "", nblocks, ORLG_WAVE * wpb, e->lds_block_bytes);
    return ORLG_OK;
}

int orlg_is_device_ptr(const void *ptr) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, ptr) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

// A pointer the kernels can use as it is: device / managed memory, or PINNED host memory (hipHostMalloc, torch's pin_memory),
// which the device reaches over the bus -- returns the device-side alias, nullptr for pageable host memory
void *orlg_device_alias(const void *ptr) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, ptr) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    if (a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged) return const_cast<void *>(ptr);
    if (a.type == hipMemoryTypeHost && a.devicePointer) return a.devicePointer;
    return nullptr;
}

// copy `bytes` from a device buffer to a caller pointer (host or device) and wait
static int copy_out(orlg_env *e, void *dst, const void *src, size_t bytes) {
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return ORLG_OK;
}

static int ensure_staging(orlg_env *e, size_t bytes) {
    if (bytes <= e->staging_bytes) return ORLG_OK;
    if (e->staging) HIP_TRY(hipFree(e->staging));
    e->staging = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&e->staging), bytes));
    e->staging_bytes = bytes;
    return ORLG_OK;
}

static int extract(orlg_env *e, int what, size_t bytes) {
    int rc = ensure_staging(e, bytes);
    if (rc) return rc;
    hipLaunchKernelGGL(orlg_extract_kernel, dim3(256), dim3(256), 0, e->stream, e->p, what, e->staging);
    HIP_TRY(hipGetLastError());
    return ORLG_OK;
}

// ---------------------------------------------------------------------------------------- C ABI
extern "C" {

int orlg_abi_version(void) { return ORLG_ABI_VERSION; }
const char *orlg_last_error(void) { return g_err.c_str(); }
double orlg_host_log(double x) { return orlg_log(x); }

int orlg_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int orlg_destroy(orlg_env *e) {
    if (!e) return ORLG_OK;
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
    for (void *b : e->bufs) (void)hipFree(b);
    if (e->staging) (void)hipFree(e->staging);
    for (int i = 0; i < 12; i++)
        if (e->io_buf[i]) (void)hipFree(e->io_buf[i]);
    if (e->d_actions) (void)hipFree(e->d_actions);
    orlg_err_word_destroy(&e->err);
    if (e->own_stream) (void)hipStreamDestroy(e->stream);
    delete e;
    return ORLG_OK;
}

int orlg_create(const orlg_topology *t, const orlg_rmsa_config *c, int32_t batch, const uint64_t *seeds,
                uint64_t base_seed, int32_t device, orlg_env **out) {
    if (!t || !c || !out) return fail(ORLG_ERR_INVALID, "null argument");
    *out = nullptr;
    const int N = t->num_nodes, E = t->num_links, K = t->k_paths, S = c->num_slots, NBR = c->num_bit_rates;
    if (batch < 1) return fail(ORLG_ERR_INVALID, "batch must be >= 1");
    if (N < 2 || N > 64) return fail(ORLG_ERR_INVALID, "num_nodes %d not in 2..64", N);
    if (E < 1 || E > 255) return fail(ORLG_ERR_INVALID, "num_links %d not in 1..255", E);
    if (S < 1 || S > 512) return fail(ORLG_ERR_INVALID, "num_slots %d not in 1..512", S);
    // bit_rate_cum == NULL: bit_rate_selection="continuous" -- bit_rates = lower .. higher, one apart, drawn with rng.randint
    const bool cont = c->bit_rate_cum == nullptr;
    if (NBR < 1 || NBR > (cont ? 256 : 64)) return fail(ORLG_ERR_INVALID, "num_bit_rates %d not in 1..%d", NBR, cont ? 256 : 64);
    if (cont) {
        if (!c->bit_rates) return fail(ORLG_ERR_INVALID, "null argument");
        for (int b = 1; b < NBR; b++)
            if (c->bit_rates[b] != c->bit_rates[0] + b) return fail(ORLG_ERR_INVALID, "continuous bit rates must be lower .. higher, one apart");
    }
    if (t->num_paths < 1 || t->num_paths >= (1 << 14)) return fail(ORLG_ERR_INVALID, "num_paths %d not in 1..16383", t->num_paths);
    int W = (S + 63) / 64;
    if (W == 7) W = 8;  // no W=7 instantiation: pad to 8 words (top word all invalid)
    if (K < 1 || K * W > 64) return fail(ORLG_ERR_INVALID, "k_paths * words_per_link = %d exceeds one wavefront", K * W);
    if (c->j < 1 || c->j > 16) return fail(ORLG_ERR_INVALID, "j %d not in 1..16", c->j);
    if (!(c->arrival_lambda > 0) || !(c->holding_lambda > 0) || !(c->channel_width > 0))
        return fail(ORLG_ERR_INVALID, "arrival_lambda, holding_lambda and channel_width must be positive");
    if (c->stats_level < 0 || c->stats_level > 2) return fail(ORLG_ERR_INVALID, "bad stats_level");
    for (int i = 0; i < N * N; i++) {
        int s = i / N, d = i % N;
        if (s != d && (t->pair_path_count[i] != K || t->pair_path_base[i] < 0 || t->pair_path_base[i] + K > t->num_paths))
            return fail(ORLG_ERR_INVALID, "pair (%d,%d) does not have k=%d path records", s, d, K);
    }
    int ndev = orlg_device_count();
    if (ndev < 1) return fail(ORLG_ERR_NO_DEVICE, "no HIP device visible: liborlg has no CPU path");
    if (device < 0 || device >= ndev) return fail(ORLG_ERR_INVALID, "device %d out of range (have %d)", device, ndev);
    HIP_TRY(hipSetDevice(device));

    orlg_env *e = new orlg_env();
    memset(&e->p, 0, sizeof(e->p));
    e->W = W; e->device = device; e->own_stream = true; e->staging = nullptr; e->staging_bytes = 0;
    e->resident_blocks = 0; e->ticket_base = 0;
    {
        hipDeviceProp_t prop;
        hipError_t er = hipGetDeviceProperties(&prop, device);
        e->num_cu = er == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    e->d_actions = nullptr; e->d_actions_cap = 0; e->num_paths = t->num_paths;
    e->err.host = nullptr; e->err.dev = nullptr; e->last_kernel[0] = 0;
    for (int i = 0; i < 12; i++) { e->io_buf[i] = nullptr; e->io_cap[i] = 0; }
    hipError_t he = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
    if (he != hipSuccess) { delete e; return fail(ORLG_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(he)); }

    {
        int rc0 = orlg_err_word_create(&e->err);
        if (rc0) { orlg_destroy(e); return rc0; }
    }
    OrlgParams &p = e->p;
    p.err_flag = e->err.dev;
    p.B = batch; p.N = N; p.E = E; p.S = S; p.K = K; p.NBR = NBR; p.NW = E * W;
    p.episode_length = c->episode_length; p.reward_mode = c->reward_mode; p.stats_level = c->stats_level; p.j = c->j;
    p.obs_dim = 1 + 2 * N + (2 * c->j + 3) * K;
    p.arrival_lambda = c->arrival_lambda; p.holding_lambda = c->holding_lambda;
    // release-queue capacity: offered load in Erlang = arrival_lambda / holding_lambda; the number of
    // services in progress is at most Poisson(load) distributed -> mean + 10 sigma, whole waves (an overflow is
    // detected and reported, never silent)
    int Q = c->queue_capacity;
    if (Q <= 0) {
        double load = c->arrival_lambda / c->holding_lambda;
        Q = (int)(load + 10.0 * std::sqrt(load));
    }
    Q = ((Q + 15) / 16) * 16;   // 16 slots = one row of the four-environments-per-wave kernel, 128 bytes of times
    if (Q < 64) Q = 64;
    if (Q > 4096) { orlg_destroy(e); return fail(ORLG_ERR_INVALID, "queue_capacity %d too large for LDS (max 4096)", Q); }
    p.Q = Q;

    // per-wave LDS layout
    auto up16 = [](int v) { return (v + 15) & ~15; };
    int off = 0;
    p.l_occ = off; off = up16(off + p.NW * 8);
    p.l_qtime = off; off = up16(off + Q * 8);
    p.l_qdesc = off; off = up16(off + Q * 4);
    p.l_mt = off; off = up16(off + ORLG_MT_N * 4);
    p.l_lstat = off; off = up16(off + 4 * E * 8);
    p.l_hist = off; off = up16(off + 4 * NBR * 4);
    p.lint_stride = (E + 3) & ~3;
    p.l_lint = off; off = up16(off + p.lint_stride * 4);
    p.l_scratch = off; off = up16(off + (int)sizeof(OrlgEnvScalars));  // staging row of the scalar record
    p.l_wsc = off; off = up16(off + (int)sizeof(OrlgWaveScalars));
    p.l_ring = off; off = up16(off + ORLG_RING * (8 + 8 + 4));
    p.l_wave_bytes = off;

    int rc = ORLG_OK;
#define TRY(x) do { rc = (x); if (rc) { orlg_destroy(e); return rc; } } while (0)
    // read-only tables: one blob, staged into LDS by every workgroup (layout = OrlgParams::t_*)
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
            int h = t->path_hops[g], se = t->path_se[g];
            if (h < 1 || h > ORLG_MAX_HOPS || t->path_link_off[g + 1] - t->path_link_off[g] != h) {
                orlg_destroy(e);
                return fail(ORLG_ERR_INVALID, "path %d: hops %d not in 1..%d or CSR mismatch", g, h, ORLG_MAX_HOPS);
            }
            if (se < 1 || se >= ORLG_NSLOT_STRIDE) { orlg_destroy(e); return fail(ORLG_ERR_INVALID, "path %d: spectral efficiency %d not in 1..7", g, se); }
            recs[g].hops = (uint8_t)h; recs[g].se = (uint8_t)se;
            for (int i = 0; i < h; i++) {
                int l = t->path_links[t->path_link_off[g] + i];
                if (l < 0 || l >= E) { orlg_destroy(e); return fail(ORLG_ERR_INVALID, "path %d: link %d out of range", g, l); }
                recs[g].link[i] = (uint8_t)l;
            }
        }
        p.t_recs = put(recs.data(), recs.size() * sizeof(OrlgPathRec));
        // get_number_slots (rmsa_env.py:708-719): ceil(bit_rate / (SE * channel_width)) + 1
        std::vector<uint16_t> ns((size_t)NBR * ORLG_NSLOT_STRIDE, 0);
        for (int b = 0; b < NBR; b++)
            for (int se = 1; se < ORLG_NSLOT_STRIDE; se++) {
                double q = (double)c->bit_rates[b] / ((double)se * c->channel_width);
                double n = std::ceil(q) + 1;
                ns[(size_t)b * ORLG_NSLOT_STRIDE + se] = (uint16_t)(n > 65535 ? 65535 : n);
            }
        p.t_nslots = put(ns.data(), ns.size() * 2);
        p.t_bitrates = put(c->bit_rates, (size_t)NBR * 4);
        {
            std::vector<double> zc((size_t)NBR, 0.0);   // (continuous: no cumulative weights, the table stays in place)
            p.t_brcum = put(cont ? zc.data() : c->bit_rate_cum, (size_t)NBR * 8);
        }
        p.br_width = cont ? NBR : 0;
        p.t_srccum = put(c->src_cum, (size_t)N * 8);
        p.t_dstcum = put(c->dst_cum, (size_t)N * N * 8);
        if (c->stats_level >= ORLG_STATS_FULL) {
            // quotients of _update_link_stats with a small integer range (rmsa_env.py:575-600): the host's IEEE division
            std::vector<double> divs(S + 1), inv(S / 2 + 2);
            for (int k = 0; k <= S; k++) divs[k] = (double)k / (double)S;
            inv[0] = 0.0;
            for (size_t k = 1; k < inv.size(); k++) inv[k] = 1.0 / (double)k;
            p.t_divs = put(divs.data(), divs.size() * 8);
            p.t_inv = put(inv.data(), inv.size() * 8);
        }
        p.tab_bytes = (int32_t)blob.size();
        p.l_outs = p.tab_bytes;
        p.l_shared_bytes = p.tab_bytes + up16(ORLG_NUM_OUTS * 8);
        unsigned char *d_blob; TRY(dev_upload(e, &d_blob, blob.data(), blob.size())); p.tables = d_blob;
    }
    // environments (waves) per workgroup: as many as fit in the 160 KiB of LDS next to the staged tables, so
    // that two workgroups still fit on a CU when possible
    {
        // pick the workgroup size that keeps the most environments (waves) resident per CU: 160 KiB of LDS, at most
        // 16 waves per CU wanted (4 per SIMD, the kernel's register budget); ties go to the larger workgroup
        const size_t lds_cu = 160 * 1024;
        int wpb = 0, best_waves = 0;
        const int wave_cap = 16;
        for (int cand = ORLG_MAX_WAVES_PER_BLOCK; cand >= 1; cand >>= 1) {
            size_t blk = (size_t)p.l_shared_bytes + (size_t)cand * p.l_wave_bytes;
            if (blk > lds_cu) continue;
            int waves = (int)(lds_cu / blk) * cand;
            if (waves > wave_cap) waves = wave_cap;
            if (waves > best_waves) { best_waves = waves; wpb = cand; }
        }
        if (wpb == 0) {
            size_t need = (size_t)p.l_shared_bytes + (size_t)p.l_wave_bytes;
            orlg_destroy(e);
            return fail(ORLG_ERR_INVALID, "tables + one environment (%zu B) exceed the 160 KiB LDS", need);
        }
        e->waves_per_block = wpb;
        e->lds_block_bytes = (size_t)p.l_shared_bytes + (size_t)wpb * p.l_wave_bytes;
    }
    // the four-environments-per-wave kernel: one environment's LDS region (no MT19937 state, no arrival ring, scalars in
    // registers), four per wave + the wave's MT19937 staging buffer; as many waves per workgroup as the LDS holds
    {
        // a wave's region is array-major: the four environments' occupancy bitmaps behind each other, then their link statistics,
        // span caches, release times, descriptors -- exactly as four consecutive environments lie in the HBM arrays, so a quad's
        // occupancy / statistics / span cache move as ONE linear copy by all 64 lanes (uniform base + lane offset); g_* = offset of
        // the array in the wave's region, row g's slice starts g * (slice bytes) further
        int go = 0;
        p.g_occ = go; go = up16(go + 4 * p.NW * 8);
        p.g_lstat = go; if (c->stats_level >= ORLG_STATS_FULL) go = up16(go + 4 * 4 * E * 8);
        p.g_hist = go;   // (unused: the four-environments-per-wave kernel updates the histograms in HBM)
        p.g_lint = go; go = up16(go + 4 * p.lint_stride * 4);
        p.g_qtime = go; go = up16(go + 4 * Q * 8);
        p.g_qdesc = go; go = up16(go + 4 * Q * 4);
        p.g_env_bytes = (go + 3) / 4;   // (per environment, for messages)
        p.g_mt = up16(ORLG_MT_N * 4);   // the workgroup's MT19937 staging buffer (then its lock word), in front of the waves' regions
        p.g_wave_bytes = go;
        e->group_wpb = 0;
        for (int cand = ORLG_GROUP_WAVES; cand >= 1 && !e->group_wpb; cand--)
            if ((size_t)p.l_shared_bytes + p.g_mt + 16 + (size_t)cand * p.g_wave_bytes <= 160 * 1024) e->group_wpb = cand;
        e->group_lds_bytes = (size_t)p.l_shared_bytes + p.g_mt + 16 + (size_t)e->group_wpb * p.g_wave_bytes;
        for (int w = 0; w <= ORLG_GROUP_WAVES; w++) e->group_resident[w] = e->group_resident_hq[w] = e->group_resident_df[w] = 0;
        e->llog = nullptr;
        e->progress = nullptr;
        {   // the instantiation that defers the link statistics keeps them in HBM: the same arrays without their slices
            int gd = up16(p.g_occ + 4 * p.NW * 8);
            e->group_df_lint = gd; gd = up16(gd + 4 * p.lint_stride * 4);
            e->group_df_qtime = gd; gd = up16(gd + 4 * Q * 8);
            e->group_df_qdesc = gd; gd = up16(gd + 4 * Q * 4);
            e->group_df_wave_bytes = gd;
            e->group_df_wpb = 0;
            for (int cand = ORLG_GROUP_WAVES; cand >= 1 && !e->group_df_wpb; cand--)
                if ((size_t)p.l_shared_bytes + p.g_mt + 16 + (size_t)cand * gd <= 160 * 1024) e->group_df_wpb = cand;
        }
        e->group_wave_bytes_hq = p.g_qtime;   // the region ends where the ring's slices would begin (they are the last arrays)
        e->group_wpb_hq = 0;
        for (int cand = ORLG_GROUP_WAVES; cand >= 1 && !e->group_wpb_hq; cand--)
            if ((size_t)p.l_shared_bytes + p.g_mt + 16 + (size_t)cand * e->group_wave_bytes_hq <= 160 * 1024) e->group_wpb_hq = cand;
        e->group_mode = c->step_kernel;
        const char *gm = getenv("ORLG_GROUP_KERNEL");  // tooling override: 0 = WAVE, 1 = GROUP
        if (gm && (gm[0] == '0' || gm[0] == '1')) e->group_mode = gm[0] == '1' ? ORLG_KERNEL_GROUP : ORLG_KERNEL_WAVE;
        if (e->group_mode < ORLG_KERNEL_AUTO || e->group_mode > ORLG_KERNEL_GROUP) {
            orlg_destroy(e);
            return fail(ORLG_ERR_INVALID, "step_kernel %d: not one of ORLG_KERNEL_AUTO / WAVE / GROUP", c->step_kernel);
        }
        if (e->group_wpb < 1) {
            if (e->group_mode == ORLG_KERNEL_GROUP) {
                orlg_destroy(e);
                return fail(ORLG_ERR_INVALID, "step_kernel GROUP: four environments (%d B each) do not fit the LDS", p.g_env_bytes);
            }
            e->group_mode = ORLG_KERNEL_WAVE;
        }
    }
    // per-env state
    TRY(dev_alloc(e, &p.occ, (size_t)batch * p.NW));
    TRY(dev_alloc(e, &p.qtime, (size_t)batch * Q));
    TRY(dev_alloc(e, &p.qdesc, (size_t)batch * Q));
    TRY(dev_alloc(e, &p.mt, (size_t)batch * ORLG_MT_N));
    TRY(dev_alloc(e, &p.scal, (size_t)batch));
    TRY(dev_alloc(e, &p.lint, (size_t)batch * p.lint_stride));
    TRY(dev_alloc(e, &p.ticket, (size_t)4));
    {
        hipError_t er = hipMemset(p.ticket, 0, 16);
        if (er != hipSuccess) { orlg_destroy(e); return fail(ORLG_ERR_HIP, "hipMemset: %s", hipGetErrorString(er)); }
    }
    TRY(dev_alloc(e, &p.hist, (size_t)batch * 4 * NBR));
    TRY(dev_alloc(e, &p.lstat, (size_t)batch * 4 * E));
    TRY(dev_alloc(e, &p.ring_iat, (size_t)batch * ORLG_RING));
    TRY(dev_alloc(e, &p.ring_ht, (size_t)batch * ORLG_RING));
    TRY(dev_alloc(e, &p.ring_req, (size_t)batch * ORLG_RING));
    {
        std::vector<uint32_t> mt((size_t)batch * ORLG_MT_N);
        for (int i = 0; i < batch; i++) orlg_mt_seed(&mt[(size_t)i * ORLG_MT_N], seeds ? seeds[i] : base_seed + (uint64_t)i);
        hipError_t er = hipMemcpy(p.mt, mt.data(), mt.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
        if (er != hipSuccess) { orlg_destroy(e); return fail(ORLG_ERR_HIP, "upload of MT19937 states: %s", hipGetErrorString(er)); }
    }
    hipLaunchKernelGGL(orlg_clear_state_kernel, dim3(512), dim3(256), 0, e->stream, p, W, 0);
    OrlgParams pi = p;
    pi.mode = ORLG_MODE_INIT; pi.n_steps = 1;
    TRY(launch_rmsa(e, pi));
    {
        hipError_t er = hipStreamSynchronize(e->stream);
        if (er != hipSuccess) { orlg_destroy(e); return fail(ORLG_ERR_HIP, "initial reset: %s", hipGetErrorString(er)); }
    }
#undef TRY
    *out = e;
    return ORLG_OK;
}

/* launch geometry of the step kernel: out[0] waves (envs) per workgroup, out[1] LDS bytes per workgroup,
 * out[2] workgroups resident per CU according to hipOccupancyMaxActiveBlocksPerMultiprocessor, out[3] W */
int orlg_launch_info(orlg_env *e, int32_t *out) {
    if (!e || !out) return fail(ORLG_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    rmsa_kernel_t k = pick_rmsa(e->W, e->p.stats_level);
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)e->lds_block_bytes));
    int nb = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(k), ORLG_WAVE * e->waves_per_block,
                                                         e->lds_block_bytes));
    out[0] = e->waves_per_block; out[1] = (int32_t)e->lds_block_bytes; out[2] = nb; out[3] = e->W;
    return ORLG_OK;
}

int orlg_last_kernel(orlg_env *e, char *buf, int32_t cap) {
    if (!e || !buf || cap < 1) return fail(ORLG_ERR_INVALID, "null argument");
    snprintf(buf, (size_t)cap, "%s", e->last_kernel);
    return ORLG_OK;
}

int orlg_set_stream(orlg_env *e, void *hip_stream) {
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

int orlg_synchronize(orlg_env *e) {
    if (!e) return fail(ORLG_ERR_INVALID, "null handle");
    HIP_TRY(hipSetDevice(e->device));
    SYNC_CHECK(e);
    return ORLG_OK;
}

int orlg_reset(orlg_env *e, int32_t only_episode_counters) {
    if (!e) return fail(ORLG_ERR_INVALID, "null handle");
    HIP_TRY(hipSetDevice(e->device));
    OrlgParams p = e->p;
    p.n_steps = 1;
    if (only_episode_counters) {
        p.mode = ORLG_MODE_EPISODE_RESET;
    } else {
        HIP_TRY(hipStreamSynchronize(e->stream));   // a full reset also clears a reported queue overflow
        *e->err.host = 0;
        hipLaunchKernelGGL(orlg_clear_state_kernel, dim3(512), dim3(256), 0, e->stream, e->p, e->W, 1);
        HIP_TRY(hipGetLastError());
        p.mode = ORLG_MODE_INIT;
    }
    return launch_rmsa(e, p);
}

__global__ void orlg_reseed_kernel(OrlgEnvScalars *scal, int B) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B; i += gridDim.x * blockDim.x) {
        scal[i].mt_idx = ORLG_MT_N;               // a freshly seeded generator: the first draw regenerates the state
        scal[i].ring_pos = 0; scal[i].ring_cnt = 0;   // arrivals pre-generated from the old generator are dropped
    }
}
int orlg_reseed(orlg_env *e, const uint64_t *seeds, uint64_t base_seed) {
    if (!e) return fail(ORLG_ERR_INVALID, "null handle");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    const int B = e->p.B;
    std::vector<uint32_t> mt((size_t)B * ORLG_MT_N);
    for (int i = 0; i < B; i++) orlg_mt_seed(&mt[(size_t)i * ORLG_MT_N], seeds ? seeds[i] : base_seed + (uint64_t)i);
    HIP_TRY(hipMemcpyAsync(e->p.mt, mt.data(), mt.size() * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(orlg_reseed_kernel, dim3(64), dim3(256), 0, e->stream, e->p.scal, B);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(e->stream));
    return ORLG_OK;
}

// io slot ids
enum { IO_PATH, IO_SLOT, IO_ACC, IO_DONE, IO_REWARD, IO_REQ, IO_ARR, IO_HOLD, IO_COMP, IO_CDIFF };

int orlg_step(orlg_env *e, int32_t policy, int32_t n_steps, const int32_t *actions, int32_t auto_reset,
              const orlg_step_io *io) {
    if (!e) return fail(ORLG_ERR_INVALID, "null handle");
    if (n_steps < 1) return fail(ORLG_ERR_INVALID, "n_steps must be >= 1");
    if (policy < ORLG_POLICY_EXTERNAL || policy > ORLG_POLICY_PATH_FF_EXTERNAL) return fail(ORLG_ERR_INVALID, "unknown policy %d", policy);
    const bool ext = policy == ORLG_POLICY_EXTERNAL || policy == ORLG_POLICY_DEEPRMSA_EXTERNAL || policy == ORLG_POLICY_PATH_FF_EXTERNAL;
    if (ext && (!actions || n_steps != 1)) return fail(ORLG_ERR_INVALID, "external actions need an action array and n_steps == 1");
    HIP_TRY(hipSetDevice(e->device));
    OrlgParams p = e->p;
    p.mode = ORLG_MODE_STEP; p.n_steps = n_steps; p.policy = policy; p.auto_reset = auto_reset;
    if (ext) {
        size_t n = (size_t)p.B * (policy == ORLG_POLICY_EXTERNAL ? 2 : 1);
        if (const void *da = orlg_device_alias(actions)) {   // device memory, or pinned host memory read over the bus
            p.actions = static_cast<const int32_t *>(da);
        } else {
            if (n > e->d_actions_cap) {
                if (e->d_actions) HIP_TRY(hipFree(e->d_actions));
                e->d_actions = nullptr;
                HIP_TRY(hipMalloc(reinterpret_cast<void **>(&e->d_actions), n * sizeof(int32_t)));
                e->d_actions_cap = n;
            }
            HIP_TRY(hipMemcpyAsync(e->d_actions, actions, n * sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
            p.actions = e->d_actions;
        }
    }
    // outputs: write straight into device pointers, stage host pointers
    struct Slot { void *user; size_t elem; };
    const size_t cnt = (size_t)n_steps * p.B;
    Slot slots[ORLG_NUM_OUTS] = {
        {io ? io->act_path : nullptr, 4},   {io ? io->act_slot : nullptr, 4},  {io ? io->accepted : nullptr, 1},
        {io ? io->done : nullptr, 1},       {io ? io->reward : nullptr, 8},    {io ? io->request : nullptr, 16},
        {io ? io->arrival : nullptr, 8},    {io ? io->holding : nullptr, 8},   {io ? io->network_compactness : nullptr, 8},
        {io ? io->network_compactness_difference : nullptr, 8},
        {io ? io->avg_link_compactness : nullptr, 8},   {io ? io->avg_link_utilization : nullptr, 8}};
    bool staged[ORLG_NUM_OUTS] = {false};
    p.out_mask = 0;
    for (int i = 0; i < ORLG_NUM_OUTS; i++) {
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
    int rc = launch_rmsa(e, p);
    if (rc) return rc;
    bool any = false;
    for (int i = 0; i < ORLG_NUM_OUTS; i++)
        if (staged[i]) {
            HIP_TRY(hipMemcpyAsync(slots[i].user, e->io_buf[i], cnt * slots[i].elem, hipMemcpyDeviceToHost, e->stream));
            any = true;
        }
    if (any) SYNC_CHECK(e);
    return ORLG_OK;
}

int orlg_get_requests(orlg_env *e, orlg_request *out) {
    if (!e || !out) return fail(ORLG_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    size_t bytes = (size_t)e->p.B * sizeof(orlg_request);
    int rc = extract(e, EX_REQUEST, bytes);
    return rc ? rc : copy_out(e, out, e->staging, bytes);
}
int orlg_get_counters(orlg_env *e, orlg_counters *out) {
    if (!e || !out) return fail(ORLG_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    size_t bytes = (size_t)e->p.B * sizeof(orlg_counters);
    int rc = extract(e, EX_COUNTERS, bytes);
    return rc ? rc : copy_out(e, out, e->staging, bytes);
}
int orlg_get_current_time(orlg_env *e, double *out) {
    if (!e || !out) return fail(ORLG_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    size_t bytes = (size_t)e->p.B * 8;
    int rc = extract(e, EX_TIME, bytes);
    return rc ? rc : copy_out(e, out, e->staging, bytes);
}
int orlg_words_per_link(orlg_env *e) { return e ? e->W : ORLG_ERR_INVALID; }
int orlg_get_occupancy(orlg_env *e, uint64_t *out) {
    if (!e || !out) return fail(ORLG_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    return copy_out(e, out, e->p.occ, (size_t)e->p.B * e->p.NW * 8);
}
int orlg_get_link_stats(orlg_env *e, double *u, double *f, double *c, double *t) {
    if (!e) return fail(ORLG_ERR_INVALID, "null handle");
    HIP_TRY(hipSetDevice(e->device));
    size_t one = (size_t)e->p.B * e->p.E * 8;
    int rc = extract(e, EX_LSTAT, 4 * one);
    if (rc) return rc;
    double *outs[4] = {u, f, c, t};
    for (int k = 0; k < 4; k++)
        if (outs[k]) HIP_TRY(hipMemcpyAsync(outs[k], e->staging + k * one, one, hipMemcpyDefault, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return ORLG_OK;
}
int orlg_get_graph_stats(orlg_env *e, double *thr, double *comp, double *lu) {
    if (!e) return fail(ORLG_ERR_INVALID, "null handle");
    HIP_TRY(hipSetDevice(e->device));
    size_t one = (size_t)e->p.B * 8;
    int rc = extract(e, EX_GRAPH, 3 * one);
    if (rc) return rc;
    double *outs[3] = {thr, comp, lu};
    for (int k = 0; k < 3; k++)
        if (outs[k]) HIP_TRY(hipMemcpyAsync(outs[k], e->staging + k * one, one, hipMemcpyDefault, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return ORLG_OK;
}
int orlg_get_bit_rate_hist(orlg_env *e, int64_t *req, int64_t *prov, int64_t *ereq, int64_t *eprov) {
    if (!e) return fail(ORLG_ERR_INVALID, "null handle");
    HIP_TRY(hipSetDevice(e->device));
    size_t one = (size_t)e->p.B * e->p.NBR * 8;
    int rc = extract(e, EX_HIST, 4 * one);
    if (rc) return rc;
    int64_t *outs[4] = {req, prov, ereq, eprov};
    for (int k = 0; k < 4; k++)
        if (outs[k]) HIP_TRY(hipMemcpyAsync(outs[k], e->staging + k * one, one, hipMemcpyDefault, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return ORLG_OK;
}
int orlg_get_num_running(orlg_env *e, int32_t *out) {
    if (!e || !out) return fail(ORLG_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    size_t bytes = (size_t)e->p.B * 4;
    int rc = extract(e, EX_RUNNING, bytes);
    return rc ? rc : copy_out(e, out, e->staging, bytes);
}
int orlg_get_episodes_done(orlg_env *e, int64_t *out) {
    if (!e || !out) return fail(ORLG_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    size_t bytes = (size_t)e->p.B * 8;
    int rc = extract(e, EX_EPISODES, bytes);
    return rc ? rc : copy_out(e, out, e->staging, bytes);
}

static int query_masks(orlg_env *e, int32_t env_index, int gid0, int count, uint64_t *masks, int32_t *nslots) {
    if (!e || !masks || !nslots) return fail(ORLG_ERR_INVALID, "null argument");
    if (env_index < 0 || env_index >= e->p.B) return fail(ORLG_ERR_INVALID, "env_index out of range");
    HIP_TRY(hipSetDevice(e->device));
    size_t mbytes = (size_t)count * e->W * 8, nbytes = (size_t)count * 4;
    int rc = ensure_staging(e, mbytes + nbytes + 64);
    if (rc) return rc;
    masks_kernel_t k = pick_masks(e->W);
    u64 *dm = reinterpret_cast<u64 *>(e->staging);
    int32_t *dn = reinterpret_cast<int32_t *>(e->staging + ((mbytes + 15) & ~(size_t)15));
    size_t lds = (size_t)e->p.l_shared_bytes + (size_t)e->p.NW * 8;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3(1), dim3(ORLG_WAVE), lds, e->stream, e->p, env_index, gid0, count, dm, dn);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(masks, dm, mbytes, hipMemcpyDefault, e->stream));
    HIP_TRY(hipMemcpyAsync(nslots, dn, nbytes, hipMemcpyDefault, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return ORLG_OK;
}

int orlg_query_path_masks(orlg_env *e, int32_t env_index, uint64_t *masks, int32_t *nslots) {
    if (!e) return fail(ORLG_ERR_INVALID, "null handle");
    return query_masks(e, env_index, -1, e->p.K, masks, nslots);
}

int orlg_query_path_mask(orlg_env *e, int32_t env_index, int32_t path_gid, uint64_t *mask, int32_t *nslots) {
    if (!e) return fail(ORLG_ERR_INVALID, "null handle");
    if (path_gid < 0 || path_gid >= e->num_paths) return fail(ORLG_ERR_INVALID, "path_gid out of range");
    return query_masks(e, env_index, path_gid, 1, mask, nslots);
}

int orlg_deeprmsa_obs_dim(orlg_env *e) { return e ? e->p.obs_dim : ORLG_ERR_INVALID; }
static int deeprmsa_observation(orlg_env *e, void *out, bool f32) {
    if (!e || !out) return fail(ORLG_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    OrlgParams p = e->p;
    size_t bytes = (size_t)p.B * p.obs_dim * (f32 ? 4 : 8);
    // device memory -- or pinned host memory, which the kernel then writes over the bus, asynchronously like a device buffer
    // (no staging copy, no wait: the caller synchronises the stream or an event) -- is written in place
    void *alias = orlg_device_alias(out);
    const bool dev = alias != nullptr;
    p.obs_f32 = f32 ? 1 : 0;
    if (!dev) {
        int rc = ensure_staging(e, bytes);
        if (rc) return rc;
        p.o_obs = reinterpret_cast<double *>(e->staging);
    } else {
        p.o_obs = reinterpret_cast<double *>(alias);
    }
    rmsa_kernel_t k = pick_obs(e->W);
    const int wpb = e->waves_per_block;
    size_t lds = (size_t)p.l_shared_bytes + (size_t)(((p.NW * 8 + 15) & ~15) + ((p.obs_dim * 8 + 15) & ~15)) * wpb;
    if (lds > 160 * 1024) return fail(ORLG_ERR_INVALID, "observation of %d values does not fit the LDS next to the tables", p.obs_dim);
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int nblocks = (p.B + wpb - 1) / wpb;
    if (nblocks > 4 * e->num_cu) nblocks = 4 * e->num_cu;   // a few workgroups per CU, each wave striding over its environments
    dim3 grid(nblocks), block(ORLG_WAVE * wpb);
    hipLaunchKernelGGL(k, grid, block, lds, e->stream, p);
    HIP_TRY(hipGetLastError());
    if (!dev) return copy_out(e, out, e->staging, bytes);
    return ORLG_OK;
}
int orlg_deeprmsa_observation(orlg_env *e, double *out) { return deeprmsa_observation(e, out, false); }
int orlg_deeprmsa_observation_f32(orlg_env *e, float *out) { return deeprmsa_observation(e, out, true); }

int orlg_simple_matrix_obs_dim(orlg_env *e) { return e ? 2 * e->p.N + e->p.E * e->p.S : ORLG_ERR_INVALID; }
int orlg_simple_matrix_observation(orlg_env *e, uint8_t *out) {
    if (!e || !out) return fail(ORLG_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    const size_t bytes = (size_t)e->p.B * (size_t)(2 * e->p.N + e->p.E * e->p.S);
    const bool dev = orlg_is_device_ptr(out);
    uint8_t *d = out;
    if (!dev) {
        int rc = ensure_staging(e, bytes);
        if (rc) return rc;
        d = e->staging;
    }
    hipLaunchKernelGGL(orlg_simple_matrix_obs_kernel, dim3(2048), dim3(256), 0, e->stream, e->p, e->W, d);
    HIP_TRY(hipGetLastError());
    if (!dev) return copy_out(e, out, e->staging, bytes);
    return ORLG_OK;
}

int orlg_reduce_counters(orlg_env *e, int64_t *out) {
    if (!e || !out) return fail(ORLG_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    int rc = ensure_staging(e, 16 * 8 + 16);
    if (rc) return rc;
    long long *d = reinterpret_cast<long long *>(e->staging);
    int *flag = reinterpret_cast<int *>(e->staging + 16 * 8);
    HIP_TRY(hipMemsetAsync(flag, 0, 4, e->stream));
    hipLaunchKernelGGL(orlg_reduce_counters_kernel, dim3(1), dim3(256), 0, e->stream, e->p.scal, e->p.B, d);
    hipLaunchKernelGGL(orlg_overflow_kernel, dim3(64), dim3(256), 0, e->stream, e->p.scal, e->p.B, flag);
    HIP_TRY(hipGetLastError());
    int hflag = 0;
    HIP_TRY(hipMemcpyAsync(&hflag, flag, 4, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(out, d, 16 * 8, hipMemcpyDefault, e->stream));
    SYNC_CHECK(e);
    if (hflag) return fail(ORLG_ERR_QUEUE_FULL, "a release queue overflowed (capacity %d): raise queue_capacity", e->p.Q);
    return ORLG_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------- checkpoint / resume
// The whole simulation state of a handle is a handful of flat device arrays: a snapshot is their concatenation.
typedef OrlgStatePart StatePart;
static std::vector<StatePart> rmsa_state_parts(orlg_env *e) {
    const OrlgParams &p = e->p;
    const size_t B = p.B;
    return {{p.occ, B * p.NW * 8}, {p.qtime, B * p.Q * 8}, {p.qdesc, B * p.Q * 4}, {p.mt, B * ORLG_MT_N * 4},
            {p.scal, B * sizeof(OrlgEnvScalars)}, {p.lint, B * p.lint_stride * 4}, {p.hist, B * 4 * p.NBR * 4},
            {p.lstat, B * 4 * p.E * 8}, {p.ring_iat, B * ORLG_RING * 8}, {p.ring_ht, B * ORLG_RING * 8},
            {p.ring_req, B * ORLG_RING * 4}};
}
int orlg_state_copy(const std::vector<OrlgStatePart> &parts, void *buffer, bool save, int device, hipStream_t stream) {
    HIP_TRY(hipSetDevice(device));
    unsigned char *b = static_cast<unsigned char *>(buffer);
    for (const StatePart &sp : parts) {
        if (save) HIP_TRY(hipMemcpyAsync(b, sp.ptr, sp.bytes, hipMemcpyDefault, stream));
        else HIP_TRY(hipMemcpyAsync(sp.ptr, b, sp.bytes, hipMemcpyDefault, stream));
        b += sp.bytes;
    }
    HIP_TRY(hipStreamSynchronize(stream));
    return ORLG_OK;
}
extern "C" {
int64_t orlg_state_size(orlg_env *e) {
    if (!e) return fail(ORLG_ERR_INVALID, "null handle");
    int64_t n = 0;
    for (const StatePart &sp : rmsa_state_parts(e)) n += (int64_t)sp.bytes;
    return n;
}
int orlg_save_state(orlg_env *e, void *buffer) {
    if (!e || !buffer) return fail(ORLG_ERR_INVALID, "null argument");
    return orlg_state_copy(rmsa_state_parts(e), buffer, true, e->device, e->stream);
}
int orlg_load_state(orlg_env *e, const void *buffer) {
    if (!e || !buffer) return fail(ORLG_ERR_INVALID, "null argument");
    int rc = orlg_state_copy(rmsa_state_parts(e), const_cast<void *>(buffer), false, e->device, e->stream);
    if (rc) return rc;
    // the sticky error word describes the state the handle holds: recomputed from the loaded scalars (a clean checkpoint
    // clears a reported ORLG_ERR_QUEUE_FULL, a checkpoint of an overflowed batch brings it back)
    *e->err.host = 0;
    hipLaunchKernelGGL(orlg_overflow_store_kernel, dim3(64), dim3(256), 0, e->stream, e->p.scal, e->p.B, e->err.dev);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(e->stream));
    return ORLG_OK;
}
}

// the int32 shape / layout fields of an environment batch, in ORLG_SHAPE_FIELDS order (tools/shape_experiment.py)
#define ORLG_SHAPE_FIELDS(X) X(N) X(E) X(S) X(K) X(NBR) X(Q) X(NW) X(lint_stride) X(tab_bytes) X(t_pair) X(t_recs) X(t_nslots) \
    X(t_bitrates) X(t_brcum) X(t_srccum) X(t_dstcum) X(t_divs) X(t_inv) X(l_occ) X(l_qtime) X(l_qdesc) X(l_mt) X(l_lstat) X(l_hist) \
    X(l_lint) X(l_scratch) X(l_wsc) X(l_ring) X(l_wave_bytes) X(l_shared_bytes) X(l_outs)
extern "C" int orlg_debug_layout(const orlg_env *e, int32_t *out, int n) {
    int i = 0;
#define X(f) if (i < n) out[i] = e->p.f; i++;
    ORLG_SHAPE_FIELDS(X)
#undef X
    return i;
}
#ifdef ORLG_SECTIONS
extern "C" int orlg_debug_sections(unsigned long long *out, int reset) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(orlg_sections), 16 * 8));
    if (reset) { unsigned long long z[16] = {0}; HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(orlg_sections), z, 16 * 8)); }
    return ORLG_OK;
}
#endif
