// dense.hip -- aread_forward / aread_backward: launch sequence of the dense path.
// No allocation, no synchronisation, no host<->device copy: the whole sequence can be captured
// into a hipGraph by the caller.
#include <cstring>
#include <cstdlib>
#include "dense_bwd_kernels.h"
#include "gemm_wide.h"
#include "tower_fused.h"
#include "tower_fused_bwd.h"


struct Ctx {
    const aread_model* m;
    const aread_call* c;
    WsLayout w;
    float* ws;
    RowsP r;
    ModeP mp;
    hipStream_t st;
    int64_t rows;
    int n_tiles;
    uint32_t thr;
    float keep_scale;
    const float* params;
    SplitKAllP splitk;      // wgrads waiting for the batched split-K reduction
    BiasAllP bias;          // layers waiting for the batched bias-gradient reduction
    hipStream_t side;       // fork-join side stream (wgrads, gate logits): work off the critical path
    int ev_next;
    // weight-gradient GEMMs waiting for the next batch launch on the side stream (flush_wgrads).  A cross-stream fork costs
    // the host and the command processor far more than a launch (measured: ~20 us per fork-launch pair against ~4 us per
    // launch, tools/event_cost.py), so the backward forks three times, not once per layer.
    GemmP pend[MAX_WGRADS];
    int n_pend;
};

// side stream waits for everything issued so far on the main stream
static int fork_side(Ctx& x) {
    hipEvent_t e = x.m->ev[x.ev_next++ % (x.m->n_ev - 1)];
    AR_HIP(hipEventRecord(e, x.st));
    AR_HIP(hipStreamWaitEvent(x.side, e, 0));
    return AREAD_OK;
}
// main stream waits for everything issued so far on the side stream
static int join_side(Ctx& x) {
    hipEvent_t e = x.m->ev[x.ev_next++ % (x.m->n_ev - 1)];
    AR_HIP(hipEventRecord(e, x.side));
    AR_HIP(hipStreamWaitEvent(x.st, e, 0));
    return AREAD_OK;
}

#define TRY0(expr) do { int _s = (expr); if (_s != AREAD_OK) return _s; } while (0)
static int make_ctx(const aread_model* m, const aread_call* c, void* stream, Ctx* x) {
    AR_CHECK_ARG(m && c, "aread: null model/call");
    AR_CHECK_ARG(c->B > 0 && c->plan && c->params && c->ws, "aread: null plan/params/ws or B <= 0");
    AR_CHECK_ARG(c->n_seg == 1 || c->n_seg == m->cfg.n_domain, "aread: n_seg=%d must be 1 or n_domain=%d", c->n_seg, m->cfg.n_domain);
    AR_CHECK_ARG(c->mode == 0 || c->mode == 1, "aread: mode=%d", c->mode);
    AR_CHECK_ARG(c->mode == 1 || c->masks, "aread: masked mode needs masks");
    AR_CHECK_ARG(c->mode == 0 || c->n_seg == 1, "aread: wo_mask runs as one segment");
    AR_CHECK_ARG(((uintptr_t)c->ws & 255) == 0 && ((uintptr_t)c->params & 15) == 0, "aread: ws must be 256-byte, params 16-byte aligned");
    x->m = m; x->c = c; x->st = (hipStream_t)stream;
    ws_layout(m, c->B, c->n_seg, &x->w);
    x->ws = (float*)c->ws;
    x->rows = x->w.max_rows;
    x->n_tiles = (int)x->w.n_tiles;
    const PlanView pv = plan_view(c->plan, c->B, c->n_seg);
    x->r.tile_seg = pv.tile_seg; x->r.tile_valid = pv.tile_valid; x->r.row_sample = pv.row_sample;
    x->r.seg_count = pv.seg_count; x->r.seg_start = pv.seg_start; x->r.hdr = pv.hdr; x->r.n_tiles = x->n_tiles; x->r.n_seg = c->n_seg;
    int32_t* ints = (int32_t*)(x->ws + x->w.kact);
    x->mp.active = (const uint8_t*)(x->ws + x->w.active);
    x->mp.kact = ints; x->mp.n0act = ints + MAX_SEG; x->mp.seg_dom = (int32_t*)(x->ws + x->w.seg_dom);
    x->mp.masks = c->mode == 0 ? c->masks : nullptr;
    x->mp.edge_count = m->edge_count; x->mp.mode = c->mode;
    const bool drop = c->train && m->cfg.dropout > 0.f;
    x->thr = drop ? drop_threshold(m->cfg.dropout) : 0u;
    x->keep_scale = drop ? 1.0f / (1.0f - m->cfg.dropout) : 1.f;
    x->params = c->params;
    x->splitk.n = 0;
    x->bias.n = 0;
    x->n_pend = 0;
    TRY0(model_streams_init(m));
    x->side = m->side;
    x->ev_next = 0;
    return AREAD_OK;
}

#define LAUNCH(kernel, grid, block, ...)                                       \
    do {                                                                       \
        hipLaunchKernelGGL(kernel, grid, block, 0, x.st, __VA_ARGS__);         \
        AR_LAUNCH_CHECK();                                                     \
    } while (0)
#define TRY(expr) do { int _s = (expr); if (_s != AREAD_OK) return _s; } while (0)

static const uint8_t* level_active(const Ctx& x, int level) {
    return level >= 0 ? x.mp.active + (size_t)level * MAX_SEG * MAX_TOWER : nullptr;
}

// ---- pre-tiled split-bf16 weight images of the tower layers (gemm_wide.h): read by the fused tower kernels ---------------
static bool layer_one_group(const LayerL& L) { return (L.in_gs == 0 && L.G > 1) || L.G == 1; }
// which == 0: forward images, 1: dgrad images.  One launch for every Linear of the tower stacks, on x.st.
static int prepare_wimg(Ctx& x, int which) {
    const aread_model* m = x.m;
    WPrepAllP a = {};
    auto add = [&](const LayerL& L, const LayerWs& lw) {
        const bool one = layer_one_group(L);
        const int out = one ? L.ncols : L.out_dim;
        WPrepOne& d = a.d[a.n++];
        d.W = x.params + L.w; d.G = one ? 1 : L.G; d.gs = (int64_t)L.out_dim * L.in_dim;
        if (which == 0) { d.N = out; d.K = L.in_dim; d.sn = L.in_dim; d.sk = 1; d.img = (__bf16*)(x.ws + lw.wimg_f); }
        else { d.N = L.in_dim; d.K = out; d.sn = 1; d.sk = L.in_dim; d.img = (__bf16*)(x.ws + lw.wimg_d); }
        d.NF = wide_nf(d.N); d.NT = (d.N + 32 * d.NF - 1) / (32 * d.NF); d.KS = (d.K + 31) / 32;
    };
    if (!m->is_mlp)
        for (int l = 0; l < m->cfg.n_level; ++l)
            for (int j = 0; j < m->towers[l].n_layers; ++j) add(m->towers[l].L[j], x.w.tw[l][j]);
    AR_CHECK_ARG(a.n <= WPREP_MAX, "too many layers for one weight-image launch");
    return launch_prep_wimg(a, x.st);
}
// ---- one MLP layer forward: H = in W^T + b (statistics in the epilogue) ; finalize ; BN+ReLU+dropout ----
static int layer_fwd(Ctx& x, const LayerL& L, const LayerWs& lw, const float* in, int level) {
    const bool shared = L.in_gs == 0 && L.G > 1;
    GemmP g = {};
    g.A = in; g.lda = L.in_ld; g.a_gs = L.in_gs;
    g.B = x.params + L.w; g.ldb = L.in_dim; g.b_gs = (int64_t)L.out_dim * L.in_dim;
    g.C = x.ws + lw.H; g.ldc = L.ncols; g.c_gs = L.out_dim;
    g.bias = x.params + L.b; g.bias_gs = L.out_dim;
    g.M = (int)x.rows; g.N = L.out_dim; g.K = L.in_dim; g.G = L.G;
    if (shared || L.G == 1) { g.N = L.ncols; g.G = 1; g.a_gs = 0; g.b_gs = 0; g.c_gs = 0; g.bias_gs = 0; }
    g.gate_axis = 1; g.tile_seg = x.r.tile_seg; g.tile_valid = x.r.tile_valid;
    g.active = (level >= 0 && g.G > 1) ? level_active(x, level) : nullptr; g.active_ld = MAX_TOWER;
    if (x.c->train) { g.stat_part = x.ws + lw.part; g.stat_ld = L.ncols; }
    // inference (eval mode, no backward to follow): BatchNorm with the running statistics + ReLU in the GEMM's epilogue, the
    // activation goes straight to Act -- no k_bn_act launch, no H round trip (expert layers; the fused tower kernel has its own)
    const bool fold = !x.c->train && x.c->inference && level < 0;
    if (fold) {
        g.C = x.ws + lw.Act;
        g.ep_rmean = x.c->stats + L.rmean; g.ep_rvar = x.c->stats + L.rvar;
        g.ep_gamma = x.params + L.gamma; g.ep_beta = x.params + L.beta;
        g.ep_seg_count = x.r.seg_count;
    }
    if (x.m->cfg.precision == 1) TRY(launch_gemm_bf3(g, x.st));
    else TRY(launch_gemm(g, true, true, x.st));
    if (fold) return AREAD_OK;
    BnActP a = {};
    a.H = x.ws + lw.H; a.Act = x.ws + lw.Act; a.part = x.ws + lw.part;
    a.mean = x.ws + lw.mean; a.rstd = x.ws + lw.rstd; a.var = x.ws + lw.var;
    a.rmean = x.c->stats + L.rmean; a.rvar = x.c->stats + L.rvar;
    a.gamma = x.params + L.gamma; a.beta = x.params + L.beta;
    a.ncols = L.ncols; a.h = L.out_dim; a.level = level; a.stack = L.stack; a.layer = L.layer; a.train = x.c->train;
    a.seed = x.c->drop_seed; a.seed_dev = x.c->drop_seed_dev; a.thr = x.thr; a.keep_scale = x.keep_scale; a.r = x.r; a.mp = x.mp;
    LAUNCH(k_bn_act, dim3(x.n_tiles, cdiv(L.ncols, 64)), dim3(256), a);
    return AREAD_OK;
}

static int stack_fwd(Ctx& x, const StackL& S, const LayerWs* lw, const float* in, int level) {
    for (int j = 0; j < S.n_layers; ++j) {
        TRY(layer_fwd(x, S.L[j], lw[j], in, level));
        in = x.ws + lw[j].Act;
    }
    return AREAD_OK;
}

static int simple_gemm(Ctx& x, const float* A, int64_t lda, bool a_kc, const float* B, int64_t ldb, bool b_kc, float* C,
                       int64_t ldc, const float* bias, int M, int N, int K, int accumulate, int gate_axis) {
    GemmP g = {};
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc; g.bias = bias;
    g.M = M; g.N = N; g.K = K; g.G = 1; g.accumulate = accumulate;
    g.gate_axis = gate_axis; g.tile_seg = x.r.tile_seg; g.tile_valid = x.r.tile_valid;
    return launch_gemm(g, a_kc, b_kc, x.st);
}

// transposed weight copies [g][in][out] for the split-bf16 dgrad of the backward (it wants k-contiguous operands);
// launched on x.st for every Linear of the model's stacks
static int transpose_weights(Ctx& x) {
    const aread_model* m = x.m;
    TransAllP ta = {};
    int64_t mx = 0;
    auto add = [&](const LayerL& L, const LayerWs& lw) {
        const bool sh = (L.in_gs == 0 && L.G > 1) || L.G == 1;
        TransOne& d = ta.d[ta.n++];
        d.W = x.params + L.w; d.WT = x.ws + lw.wT; d.G = sh ? 1 : L.G; d.out = sh ? L.ncols : L.out_dim; d.in = L.in_dim;
        if ((int64_t)L.ncols * L.in_dim > mx) mx = (int64_t)L.ncols * L.in_dim;
    };
    for (int j = 0; j < m->experts.n_layers; ++j) add(m->experts.L[j], x.w.ex[j]);
    if (!m->is_mlp)
        for (int l = 0; l < m->cfg.n_level; ++l)
            for (int j = 0; j < m->towers[l].n_layers; ++j) add(m->towers[l].L[j], x.w.tw[l][j]);
    int bx = cdiv(mx, 256);
    if (bx > 256) bx = 256;
    LAUNCH(k_transpose_weights, dim3(bx, ta.n), dim3(256), ta);
    return AREAD_OK;
}

template <int MAXV>
static void launch_rowwise_fwd(Ctx& x, const RowwiseP& p) {
    hipLaunchKernelGGL((k_rowwise_fwd<MAXV>), dim3(x.n_tiles * SUB), dim3(256), 0, x.st, p);
}

// ---- fused tower pyramid (tower_fused.h) -----------------------------------------------------------------------------
// ---- diagnostics: GPU time stamps of the phases of the main stream (aread_debug_set("phase_events", 1), tools/phase_times.py) ----
#define N_PHASE_EV 16
static int g_phase_on = 0;
static hipEvent_t g_phase_ev[N_PHASE_EV];
static bool g_phase_init = false, g_phase_rec[N_PHASE_EV] = {};
static void phase_mark(hipStream_t st, int i) {
    if (!g_phase_on) return;
    if (!g_phase_init) { for (int k = 0; k < N_PHASE_EV; ++k) (void)hipEventCreate(&g_phase_ev[k]); g_phase_init = true; }
    (void)hipEventRecord(g_phase_ev[i], st);
    g_phase_rec[i] = true;
}
extern "C" int aread_debug_phase_times(float* out_ms, int n) {
    AR_CHECK_ARG(out_ms && g_phase_init, "aread_debug_phase_times: phase events are not enabled");
    for (int i = 0; i + 1 < N_PHASE_EV && i < n; ++i) {
        out_ms[i] = -1.f;
        if (g_phase_rec[i] && g_phase_rec[i + 1] && hipEventQuery(g_phase_ev[i]) == hipSuccess && hipEventQuery(g_phase_ev[i + 1]) == hipSuccess)
            (void)hipEventElapsedTime(&out_ms[i], g_phase_ev[i], g_phase_ev[i + 1]);
    }
    (void)hipGetLastError();
    return AREAD_OK;
}

static int g_fused_mode = -1;        // AREAD_FUSED_TOWERS: 0 = layer-by-layer launches (default: measured faster), 1 = fused tower forward
extern int g_plan_single;          // plan.hip
static int g_two_hop_nt = -1;      // AREAD_TWO_HOP_NT / aread_debug_set("two_hop_nt", v): segments of more tiles merge their BatchNorm statistics in two hops
static int two_hop_nt() {
    if (g_two_hop_nt < 0) { const char* e = getenv("AREAD_TWO_HOP_NT"); g_two_hop_nt = e ? atoi(e) : TF_TWO_HOP_NT; }
    return g_two_hop_nt;
}
static int g_two_hop_nt_ab = -1;   // aread_debug_set("two_hop_nt_act_bn", v): the threshold of k_act_bn_bwd alone (tests: the towers' merge held fixed)
static int g_fused_act_bn = -1;    // AREAD_FUSED_ACT_BN / aread_debug_set("fused_act_bn", v): k_act_bn_bwd for the expert layers (A/B)
static int g_n_cu = 0;
static long long g_fused_fwd_calls = 0, g_fused_bwd_calls = 0;   // aread_debug_get: the tests check that the fused kernels really ran
static int g_tf_stamps = 0;     // aread_debug_set("tf_stamps", 1): phase time stamps of k_tower_fwd into the workspace (tools/tf_stamps.py)
static size_t tower_fwd_lds(const aread_model* m, TFwdP* p) {
    const aread_model_cfg& c = m->cfg;
    int max_blk_bytes = 0, max_cols = 0, max_ngate = 0;
    for (int l = 0; l < c.n_level; ++l) {
        const int n_src = l == 0 ? c.n_expert : c.n_tower[l - 1];
        if (c.n_tower[l] * n_src > max_ngate) max_ngate = c.n_tower[l] * n_src;
        for (int j = 0; j < m->towers[l].n_layers; ++j) {
            const LayerL& L = m->towers[l].L[j];
            const int pk = L.in_dim >= 32 ? 4 : L.in_dim / 8, ks = (L.in_dim + 31) / 32;
            const int bytes = L.G * ks * 2 * pk * 512 * 2;
            if (bytes > max_blk_bytes) max_blk_bytes = bytes;
            if (L.ncols > max_cols) max_cols = L.ncols;
        }
    }
    auto up = [](size_t v) { return (v + 1023) & ~(size_t)1023; };
    size_t o = 0;
    if (p) p->lds_aimg = (int)o;
    o = up(o + max_blk_bytes);
    if (p) p->lds_actf = (int)o;
    o = up(o + (size_t)TILE_M * max_cols * 4);
    if (p) { p->lds_gate = (int)o; p->max_ngate = max_ngate; }
    o = up(o + (size_t)2 * TILE_M * max_ngate * 4);
    return o;
}
static bool tower_fused_ok(const Ctx& x) {
    const aread_model* m = x.m;
    const aread_model_cfg& c = m->cfg;
    if (g_fused_mode < 0) {
        const char* e = getenv("AREAD_FUSED_TOWERS");
        g_fused_mode = e ? atoi(e) : 1;                        // default on: 103 us in one launch instead of 19 launches / ~150 us
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&g_n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            g_n_cu = 0;
    }
    if (!g_fused_mode || c.precision != 1 || m->is_mlp) return false;
    // every workgroup of a segment must be resident (one per CU, ~150 KB of LDS each); keep a margin of CUs for whatever else
    // holds LDS at the same time (RCCL's collective kernels in the multi-GPU step)
    static int margin = -1;
    if (margin < 0) { const char* e = getenv("AREAD_FUSED_CU_MARGIN"); margin = e ? atoi(e) : 32; }
    if (x.n_tiles > g_n_cu - margin) return false;
    if (m->n_heads > MAX_TOWER || m->ld_h > 64) return false;
    int prev_w = m->experts.L[m->experts.n_layers - 1].out_dim;
    for (int l = 0; l < c.n_level; ++l) {
        const int n_src = l == 0 ? c.n_expert : c.n_tower[l - 1];
        if (c.n_tower[l] > MAX_TOWER || n_src > MAX_TOWER) return false;
        for (int j = 0; j < m->towers[l].n_layers; ++j) {
            const LayerL& L = m->towers[l].L[j];
            if (L.in_dim != prev_w || L.in_dim % 8 || L.out_dim % 4 || L.out_dim > 64 || L.ncols > 256) return false;
            if (L.G * ((L.out_dim + 15) / 16) > TF_WAVES * TF_MAX_UNITS) return false;
            prev_w = L.out_dim;
        }
    }
    if (prev_w != m->h_last) return false;
    return tower_fwd_lds(m, nullptr) + 4096 <= 160 * 1024;
}

static int tower_fused_fwd(Ctx& x, bool want_gates) {
    const aread_model* m = x.m;
    const aread_call* c = x.c;
    const aread_model_cfg& cfg = m->cfg;
    float* ws = x.ws;
    const float* P = x.params;
    TFwdP p = {};
    p.n_level = cfg.n_level; p.n_layers = m->towers[0].n_layers; p.train = c->train; p.mode = c->mode;
    p.two_hop_nt = two_hop_nt();
    // the latency-bound tower kernel raises its waves' issue priority over the streaming table-L2 sweep that shares its CUs (-3..10 us per step)
    { static int pr = -1; if (pr < 0) { const char* e = getenv("AREAD_TOWER_PRIO"); pr = e ? atoi(e) : 1; } p.prio = pr; }
    p.seed = c->drop_seed; p.seed_dev = c->drop_seed_dev; p.thr = x.thr; p.keep_scale = x.keep_scale;
    for (int l = 0; l < cfg.n_level; ++l) {
        p.n_t[l] = cfg.n_tower[l]; p.mask_off[l] = m->mask_off[l]; p.gate_off[l] = m->gate_off[l];
        p.In[l] = ws + x.w.In[l];
        for (int j = 0; j < m->towers[l].n_layers; ++j) {
            const LayerL& L = m->towers[l].L[j];
            const LayerWs& lw = x.w.tw[l][j];
            TFLayer& T = p.L[l][j];
            T.n_t = L.G; T.in_w = L.in_dim; T.out_w = L.out_dim; T.ncols = L.ncols;
            T.ks = (L.in_dim + 31) / 32; T.nfr = (L.out_dim + 15) / 16; T.pk = L.in_dim >= 32 ? 4 : L.in_dim / 8;
            T.stack = L.stack; T.layer = L.layer;
            T.wimg = (const __bf16*)(ws + lw.wimg_f);
            T.bias = P + L.b; T.gamma = P + L.gamma; T.beta = P + L.beta;
            T.rmean = c->stats + L.rmean; T.rvar = c->stats + L.rvar;
            T.H = ws + lw.H; T.Act = ws + lw.Act; T.part = ws + lw.part; T.mean = ws + lw.mean; T.rstd = ws + lw.rstd; T.var = ws + lw.var;
            T.tags = (tf_u64*)(ws + lw.tag_f);
            T.fin = (tf_u64*)(ws + lw.fin_f);
        }
    }
    const int nle = m->experts.n_layers;
    p.X = ws + x.w.ex[nle - 1].Act; p.n_exp = cfg.n_expert; p.xw = m->experts.L[nle - 1].out_dim;
    p.glogE = ws + x.w.glogE; p.ld_ge = m->ld_ge; p.glogT = ws + x.w.glogT; p.ld_gt = m->ld_gt;
    p.gate_part = want_gates ? ws + x.w.gate_part : nullptr;
    p.hc = ws + x.w.hc; p.lin = ws + x.w.lin; p.head_w = P + m->head_w; p.head_ld = m->head_ld; p.D = m->D; p.n_heads = m->n_heads;
    p.ld_h = m->ld_h; p.h_last = m->h_last;
    p.z = ws + x.w.z; p.prob = ws + x.w.prob; p.dz = ws + x.w.dz; p.probs_out = c->probs; p.B = c->B;
    p.y = c->y; p.seg_weight = c->seg_weight; p.loss_part = (c->y && c->loss_out) ? ws + x.w.loss_part : nullptr;
    p.err = (unsigned*)(ws + x.w.tf_sync) + AREAD_MAX_LEVEL * AREAD_MAX_LAYER * MAX_SEG;
    p.stamps = g_tf_stamps ? (unsigned long long*)(ws + x.w.misc_part) : nullptr;   // diagnostics: misc_part is free during the forward
    p.r = x.r; p.mp = x.mp;
    const size_t lds = tower_fwd_lds(m, &p);
    static size_t lds_set = 0;
    if (lds > lds_set) {
        AR_HIP(hipFuncSetAttribute((const void*)k_tower_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        lds_set = lds;
    }
    hipLaunchKernelGGL(k_tower_fwd, dim3(x.n_tiles), dim3(TF_THREADS), lds, x.st, p);
    ++g_fused_fwd_calls;
    AR_LAUNCH_CHECK();
    return AREAD_OK;
}


// ---- fused tower pyramid, backward (tower_fused_bwd.h) ------------------------------------------------------------------
static int g_fused_bwd = -1;     // AREAD_FUSED_TOWERS_BWD (default 1); needs the forward conditions (tower_fused_ok) as well
static size_t tower_bwd_lds(const aread_model* m, TBwdP* p) {
    const aread_model_cfg& c = m->cfg;
    int max_blk_bytes = 0, max_cols = 0, max_ngate = 0;
    for (int l = 0; l < c.n_level; ++l) {
        const int n_src = l == 0 ? c.n_expert : c.n_tower[l - 1];
        if (l > 0 && c.n_tower[l] * n_src > max_ngate) max_ngate = c.n_tower[l] * n_src;
        for (int j = 0; j < m->towers[l].n_layers; ++j) {
            const LayerL& L = m->towers[l].L[j];
            const int pk = L.out_dim >= 32 ? 4 : L.out_dim / 8, ks = (L.out_dim + 31) / 32;
            const int bytes = L.G * ks * 2 * pk * 512 * 2;
            if (bytes > max_blk_bytes) max_blk_bytes = bytes;
            if (L.ncols > max_cols) max_cols = L.ncols;
            if (L.G * L.in_dim > max_cols) max_cols = L.G * L.in_dim;
        }
    }
    auto up = [](size_t v) { return (v + 1023) & ~(size_t)1023; };
    const int ldd = max_cols + 4;
    const size_t dbytes = up((size_t)TILE_M * ldd * 4);
    size_t scratch = (size_t)8 * 256 * 4;                                   // merge scratch
    if ((size_t)(2 * TILE_M + 16) * max_ngate * 4 > scratch) scratch = (size_t)(2 * TILE_M + 16) * max_ngate * 4;   // mixing weights + dot products + lane table
    if ((size_t)TILE_M * m->n_heads * 4 > scratch) scratch = (size_t)TILE_M * m->n_heads * 4;
    size_t abytes = up(max_blk_bytes > (int)scratch ? (size_t)max_blk_bytes : scratch);
    // the expert-output tile of the MMoE mix backward spans D1 + the A image region
    const int nle = m->experts.n_layers;
    const int wx = c.n_expert * m->experts.L[nle - 1].out_dim, ldx = wx + 4;
    const size_t xbytes = (size_t)TILE_M * ldx * 4 + (size_t)(2 * TILE_M + 16) * c.n_tower[0] * c.n_expert * 4;
    if (xbytes > dbytes + abytes) abytes = up(xbytes - dbytes);
    if (p) { p->lds_d0 = 0; p->lds_d1 = (int)dbytes; p->lds_aimg = (int)(2 * dbytes); p->ldd = ldd; p->ldx = ldx; }
    return 2 * dbytes + abytes;
}
static bool tower_fused_bwd_ok(const Ctx& x) {
    if (g_fused_bwd < 0) {
        const char* e = getenv("AREAD_FUSED_TOWERS_BWD");
        g_fused_bwd = e ? atoi(e) : 1;
    }
    if (!g_fused_bwd || !tower_fused_ok(x)) return false;
    const aread_model* m = x.m;
    const aread_model_cfg& c = m->cfg;
    if (c.n_expert > 8) return false;
    for (int l = 0; l < c.n_level; ++l)
        for (int j = 0; j < m->towers[l].n_layers; ++j) {
            const LayerL& L = m->towers[l].L[j];
            if (L.out_dim % 8 || L.in_dim % 4 || L.in_dim > 64 || L.G * L.in_dim > 256) return false;
            if (L.G * ((L.in_dim + 15) / 16) > TF_WAVES * TF_MAX_UNITS) return false;
        }
    return tower_bwd_lds(m, nullptr) + 4096 <= 160 * 1024;
}

static int wgrad(Ctx& x, const float* dY, int64_t ld_dy, int64_t dy_gs, const float* X, int64_t ldx, int64_t x_gs, int G,
                 int M, int N, float* out, int64_t ldo, int64_t o_gs, const uint8_t* active, int64_t slab_off);
static const uint8_t* level_active(const Ctx& x, int level);

// heads backward, the tower pyramid top down, gate-mix and MMoE-mix backward in one launch; queues the same weight-gradient
// GEMMs and bias reductions as layer_bwd
static int tower_fused_bwd(Ctx& x, float* grads) {
    const aread_model* m = x.m;
    const aread_call* c = x.c;
    const aread_model_cfg& cfg = m->cfg;
    float* ws = x.ws;
    const float* P = x.params;
    const int LL = cfg.n_level - 1, nle = m->experts.n_layers;
    TBwdP p = {};
    p.n_level = cfg.n_level; p.n_layers = m->towers[0].n_layers; p.train = c->train; p.mode = c->mode;
    p.two_hop_nt = two_hop_nt();
    // the latency-bound tower kernel raises its waves' issue priority over the streaming table-L2 sweep that shares its CUs (-3..10 us per step)
    { static int pr = -1; if (pr < 0) { const char* e = getenv("AREAD_TOWER_PRIO"); pr = e ? atoi(e) : 1; } p.prio = pr; }
    p.seed = c->drop_seed; p.seed_dev = c->drop_seed_dev; p.thr = x.thr; p.keep_scale = x.keep_scale;
    for (int l = 0; l < cfg.n_level; ++l) {
        p.n_t[l] = cfg.n_tower[l]; p.mask_off[l] = m->mask_off[l]; p.gate_off[l] = m->gate_off[l];
        p.prevAct[l] = l > 0 ? ws + x.w.tw[l - 1][m->towers[l - 1].n_layers - 1].Act : nullptr;
        for (int j = 0; j < m->towers[l].n_layers; ++j) {
            const LayerL& L = m->towers[l].L[j];
            const LayerWs& lw = x.w.tw[l][j];
            TBLayer& T = p.L[l][j];
            T.n_t = L.G; T.in_w = L.in_dim; T.out_w = L.out_dim; T.ncols = L.ncols;
            T.ks = (L.out_dim + 31) / 32; T.nfr = (L.in_dim + 15) / 16; T.pk = L.out_dim >= 32 ? 4 : L.out_dim / 8;
            T.stack = L.stack; T.layer = L.layer;
            T.wimg = (const __bf16*)(ws + lw.wimg_d);
            T.gamma = P + L.gamma; T.beta = P + L.beta;
            T.H = ws + lw.H; T.mean = ws + lw.mean; T.rstd = ws + lw.rstd;
            T.dH = ws + lw.dAct; T.bpart = ws + lw.bpart; T.cpart = ws + lw.cpart;
            T.tags = (tf_u64*)(ws + lw.tag_b);
            T.fin = (tf_u64*)(ws + lw.fin_b);
        }
    }
    p.X = ws + x.w.ex[nle - 1].Act; p.n_exp = cfg.n_expert; p.xw = m->experts.L[nle - 1].out_dim; p.dX = ws + x.w.ex[nle - 1].dAct;
    p.glogE = ws + x.w.glogE; p.dglogE = ws + x.w.dglogE; p.ld_ge = m->ld_ge;
    p.glogT = ws + x.w.glogT; p.dglogT = ws + x.w.dglogT; p.ld_gt = m->ld_gt;
    p.dz = ws + x.w.dz; p.actLast = ws + x.w.tw[LL][m->towers[LL].n_layers - 1].Act;
    p.head_w = P + m->head_w; p.head_ld = m->head_ld; p.D = m->D; p.n_heads = m->n_heads; p.ld_h = m->ld_h; p.h_last = m->h_last;
    p.dlin = ws + x.w.dlin; p.head_part = ws + x.w.misc_part; p.ld_hp = 1024;
    p.err = (unsigned*)(ws + x.w.tf_sync) + AREAD_MAX_LEVEL * AREAD_MAX_LAYER * MAX_SEG;     // the forward's error word
    p.stamps = g_tf_stamps == 2 ? (unsigned long long*)(ws + x.w.gate_part) : nullptr;        // diagnostics only
    p.r = x.r; p.mp = x.mp;
    const int swaps = cfg.n_level * p.n_layers + (cfg.n_level - 1);
    p.first_buf = swaps & 1;                                 // the level-0 input gradient must end up in D0 (the X tile spans D1)
    const size_t lds = tower_bwd_lds(m, &p);
    static size_t lds_set = 0;
    if (lds > lds_set) {
        AR_HIP(hipFuncSetAttribute((const void*)k_tower_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        lds_set = lds;
    }
    if (!m->bwd_tags_clean)                                  // a second backward on one forward: the tags of the first are stale
        AR_HIP(hipMemsetAsync(ws + x.w.tf_tags, 0, (size_t)x.w.tf_tags_floats * sizeof(float), x.st));
    m->bwd_tags_clean = false;
    hipLaunchKernelGGL(k_tower_bwd, dim3(x.n_tiles), dim3(TF_THREADS), lds, x.st, p);
    ++g_fused_bwd_calls;
    AR_LAUNCH_CHECK();
    // the side-stream consumers of what the kernel wrote: bias / gamma / beta reductions and weight gradients, tower by tower
    for (int l = LL; l >= 0; --l) {
        const StackL& S = m->towers[l];
        for (int j = S.n_layers - 1; j >= 0; --j) {
            const LayerL& L = S.L[j];
            const LayerWs& lw = x.w.tw[l][j];
            const float* in = j == 0 ? ws + x.w.In[l] : ws + x.w.tw[l][j - 1].Act;
            AR_CHECK_ARG(x.bias.n < MAX_BN_LAYERS_DECL, "too many layers");
            BiasOne& bo = x.bias.d[x.bias.n++];
            bo.cpart = ws + lw.cpart; bo.bpart = ws + lw.bpart; bo.db = grads + L.b; bo.dgamma = grads + L.gamma; bo.dbeta = grads + L.beta;
            bo.ncols = L.ncols; bo.h = L.out_dim; bo.level = l;
            const uint8_t* act = L.G > 1 ? level_active(x, l) : nullptr;
            TRY(wgrad(x, ws + lw.dAct, L.ncols, L.out_dim, in, L.in_ld, L.in_gs, L.G, L.out_dim, L.in_dim, grads + L.w, L.in_dim,
                      (int64_t)L.out_dim * L.in_dim, act, x.w.slab_tw[l][j]));
        }
    }
    return AREAD_OK;
}

// ---- the forward's tail on the side stream: loss value from the per-tile partials, running statistics in domain order ----
static int forward_tail(Ctx& x, bool fused_towers) {
    const aread_model* m = x.m;
    const aread_call* c = x.c;
    const aread_model_cfg& cfg = m->cfg;
    float* ws = x.ws;
    if (c->y && c->loss_out) {
        hipLaunchKernelGGL(k_loss_finish, dim3(1), dim3(64), 0, x.side, ws + x.w.loss_part, c->seg_weight, c->loss_out, x.r,
                           fused_towers ? (const unsigned*)(ws + x.w.tf_sync) + AREAD_MAX_LEVEL * AREAD_MAX_LAYER * MAX_SEG : nullptr);
        AR_LAUNCH_CHECK();
    }
    if (c->train && c->update_running) {
        BnRunAllP all = {};
        int max_cols = 0;
        auto add = [&](const LayerL& L, const LayerWs& lw, int level) {
            BnRunP& p = all.L[all.n_layers++];
            p.mean = ws + lw.mean; p.var = ws + lw.var; p.rmean = c->stats + L.rmean; p.rvar = c->stats + L.rvar;
            p.nbt = c->nbt + L.nbt0; p.ncols = L.ncols; p.h = L.out_dim; p.level = level;
            if (L.ncols > max_cols) max_cols = L.ncols;
        };
        for (int j = 0; j < m->experts.n_layers; ++j) add(m->experts.L[j], x.w.ex[j], -1);
        for (int l = 0; l < cfg.n_level; ++l)
            for (int j = 0; j < m->towers[l].n_layers; ++j) add(m->towers[l].L[j], x.w.tw[l][j], l);
        all.r = x.r; all.mp = x.mp;
        // off the critical path: the statistics buffers are not touched again before the next forward
        hipLaunchKernelGGL(k_bn_running, dim3(cdiv(max_cols, 256), all.n_layers), dim3(256), 0, x.side, all);
        AR_LAUNCH_CHECK();
    }
    return AREAD_OK;
}

// ---- the part of the forward's side work that depends on the parameters, the masks and the call's scalars only (not on
// the row plan, not on the embedding output): mask tables + group embedding, pre-tiled weight images, hand-off tag memsets,
// the probs memset; in training also what only the backward reads (transposed weights, dgrad weight images, behind
// ev_prep).  Launched on x.st (the side stream).  aread_prepare() issues it at the very start of a step -- before the row
// plan and the gather -- so that only the row-wise trunk and the three gate / head GEMMs remain between the gather and the
// tower kernel on the side stream; without aread_prepare, aread_forward runs it in place.
static int forward_prep(Ctx& x, bool with_plan_independent_tail) {
    const aread_model* m = x.m;
    const aread_call* c = x.c;
    const aread_model_cfg& cfg = m->cfg;
    float* ws = x.ws;
    const float* P = x.params;
    MaskPrepP mp = {};
    mp.masks = x.mp.masks; mp.n_seg = c->n_seg; mp.domain = c->domain; mp.mode = c->mode; mp.n_level = cfg.n_level;
    mp.n_domain = cfg.n_domain; mp.edge_count = m->edge_count; mp.E = m->E;
    for (int l = 0; l < cfg.n_level; ++l) mp.n_tower[l] = cfg.n_tower[l];
    for (int l = 0; l <= cfg.n_level; ++l) mp.mask_off[l] = m->mask_off[l];
    mp.group_emb = P + m->group_emb;
    mp.active = (uint8_t*)(ws + x.w.active); mp.kact = (int32_t*)(ws + x.w.kact); mp.n0act = mp.kact + MAX_SEG;
    mp.seg_dom = (int32_t*)(ws + x.w.seg_dom); mp.grp = ws + x.w.grp;
    const bool fused_towers = tower_fused_ok(x);
    LAUNCH(k_mask_prep, dim3(1), dim3(256), mp);
    if (cfg.precision == 1 && fused_towers) TRY(prepare_wimg(x, 0));   // (the fused tower kernel reads them after the join)
    if (g_fused_act_bn < 0) { const char* e = getenv("AREAD_FUSED_ACT_BN"); g_fused_act_bn = e ? atoi(e) : 1; }
    m->ab_tags_clean = false;
    if (fused_towers || g_fused_act_bn > 0) {
        // hand-off granules of the fused tower kernels (forward AND backward of this step) and the error word: zero tags
        const int64_t n_tag = x.w.tf_tags_floats + (g_fused_act_bn > 0 ? x.w.ab_tags_floats : 0);
        m->ab_tags_clean = g_fused_act_bn > 0;
        AR_HIP(hipMemsetAsync(ws + x.w.tf_tags, 0, (size_t)n_tag * sizeof(float), x.st));
        AR_HIP(hipMemsetAsync((unsigned*)(ws + x.w.tf_sync) + AREAD_MAX_LEVEL * AREAD_MAX_LAYER * MAX_SEG, 0, 16 * sizeof(unsigned), x.st));
        m->bwd_tags_clean = true;
    }
    if (c->probs) AR_HIP(hipMemsetAsync(c->probs, 0, (size_t)m->n_heads * c->B * sizeof(float), x.st));
    if (with_plan_independent_tail) {
        // (only the backward reads these; aread_backward waits for ev_prep)
        m->prep_pending = false;
        if (c->train && cfg.precision == 1) {
            TRY(transpose_weights(x));
            if (tower_fused_bwd_ok(x)) TRY(prepare_wimg(x, 1));   // dgrad images of the tower layers
            AR_HIP(hipEventRecord(m->ev_prep, x.st));
            m->prep_pending = true;
        }
    }
    return AREAD_OK;
}

// make_ctx without a row plan (aread_prepare runs before the plan exists)
static int make_ctx_noplan(const aread_model* m, const aread_call* c, void* stream, Ctx* x) {
    AR_CHECK_ARG(m && c, "aread: null model/call");
    AR_CHECK_ARG(c->B > 0 && c->params && c->ws, "aread: null params/ws or B <= 0");
    AR_CHECK_ARG(c->n_seg == 1 || c->n_seg == m->cfg.n_domain, "aread: n_seg=%d must be 1 or n_domain=%d", c->n_seg, m->cfg.n_domain);
    AR_CHECK_ARG(c->mode == 0 || c->mode == 1, "aread: mode=%d", c->mode);
    AR_CHECK_ARG(c->mode == 1 || c->masks, "aread: masked mode needs masks");
    AR_CHECK_ARG(c->mode == 0 || c->n_seg == 1, "aread: wo_mask runs as one segment");
    AR_CHECK_ARG(((uintptr_t)c->ws & 255) == 0 && ((uintptr_t)c->params & 15) == 0, "aread: ws must be 256-byte, params 16-byte aligned");
    aread_call tmp = *c;
    static const int32_t dummy_plan[4] = {0, 0, 0, 0};
    tmp.plan = dummy_plan;                                   // never dereferenced: the kernels of forward_prep take no RowsP
    TRY0(make_ctx(m, &tmp, stream, x));
    x->c = c;
    x->r = RowsP{};
    return AREAD_OK;
}

extern "C" int aread_prepare(const aread_model* m, const aread_call* c, void* stream) {
    Ctx x;
    TRY(make_ctx_noplan(m, c, stream, &x));
    AR_CHECK_ARG(!m->is_mlp, "aread_prepare: not an AREAD handle");
    // the side stream forks HERE (everything queued on `stream` so far -- the previous step included -- precedes it)
    TRY(fork_side(x));
    const hipStream_t main_st = x.st;
    x.st = x.side;
    const int rc = forward_prep(x, true);
    // (everything above precedes the event the forward's main stream waits for before the tower kernel: the backward needs no
    // separate wait for the transposed weights / dgrad images)
    m->prep_pending = false;
    if (rc == AREAD_OK && c->init_grads) {
        // dense half of get_regularization_loss at the HEAD of the step: grads = 2*coef*w, reg = sum coef*w^2 (k_l2_dense, init
        // mode); the backward's reductions run on this stream and its successor (side2 waits on it) and ADD onto the buffer
        if (!(c->l2_dense_coef && c->init_reg_out)) { x.st = main_st; AR_CHECK_ARG(false, "aread_prepare: init_grads needs l2_dense_coef and init_reg_out"); }
        hipLaunchKernelGGL(k_l2_dense, dim3(256), dim3(256), 0, x.side, x.params, c->l2_dense_coef, m->n_params, c->init_grads,
                           c->init_reg_out + 1, c->init_reg_out, 0, (const float*)nullptr, (float*)nullptr, 1);
        if (hipGetLastError() != hipSuccess) { x.st = main_st; AR_CHECK_ARG(false, "aread_prepare: k_l2_dense launch failed"); }
    }
    x.st = main_st;
    if (rc != AREAD_OK) return rc;
    m->fwd_prepared = c->ws;                                 // consumed by the next aread_forward on this workspace
    return AREAD_OK;
}

extern "C" int aread_forward(const aread_model* m, const aread_call* c, const float* e_in, void* stream) {
    Ctx x;
    TRY(make_ctx(m, c, stream, &x));
    AR_CHECK_ARG(e_in && c->stats, "aread_forward: null e_in/stats");
    AR_CHECK_ARG(!(c->train && c->update_running) || c->nbt, "aread_forward: update_running needs nbt");
    const aread_model_cfg& cfg = m->cfg;
    const int D = m->D, E = m->E;
    float* ws = x.ws;
    const float* P = x.params;
    const bool prepared = m->fwd_prepared == c->ws && c->ws != nullptr;      // aread_prepare already queued the plan-independent side work
    m->fwd_prepared = nullptr;
    const bool fused_towers = tower_fused_ok(x);
    // 4. experts FIRST on the main stream (issue order = the order a captured graph schedules independent branches): the
    // side work below forks from the point before them
    hipEvent_t ev_f0 = x.m->ev[x.ev_next++ % (x.m->n_ev - 1)];
    AR_HIP(hipEventRecord(ev_f0, x.st));
    phase_mark(x.st, 0);
    TRY(stack_fwd(x, m->experts, x.w.ex, e_in, -1));
    phase_mark(x.st, 1);
    // 1.-3. (side stream, joined before the MMoE mix: the expert layers need none of it) mask tables, row-wise trunk,
    // gate logits, cross-network part of the heads
    AR_HIP(hipStreamWaitEvent(x.side, ev_f0, 0));
    hipEvent_t ev_towers = nullptr;
    {
        const hipStream_t main_st = x.st;
        x.st = x.side;
        if (!prepared) TRY(forward_prep(x, false));
        RowwiseP rw = {};
        rw.e = e_in; rw.cn = ws + x.w.cn; rw.lin = ws + x.w.lin; rw.xw = ws + x.w.xw; rw.q = ws + x.w.q; rw.grp = ws + x.w.grp;
        rw.lin_w = P + m->lin_w; rw.lin_b = P + m->lin_b; rw.cn_w = P + m->cn_w; rw.cn_b = P + m->cn_b;
        rw.D = D; rw.E = E; rw.n_cross = cfg.n_cross; rw.dom_field = cfg.domain_field; rw.rows = x.rows; rw.r = x.r;
        if (D <= 256) launch_rowwise_fwd<1>(x, rw); else if (D <= 512) launch_rowwise_fwd<2>(x, rw); else launch_rowwise_fwd<4>(x, rw);
        AR_LAUNCH_CHECK();
        const int n_ge = cfg.n_tower[0] * cfg.n_expert;
        TRY(simple_gemm(x, e_in, D, true, P + m->gate_w, D, true, ws + x.w.glogE, m->ld_ge, P + m->gate_b, (int)x.rows, n_ge, D, 0, 1));
        if (m->gate_rows > 0)
            TRY(simple_gemm(x, ws + x.w.q, 2 * E, true, P + m->tgate_w, 2 * E, true, ws + x.w.glogT, m->ld_gt, P + m->tgate_b,
                            (int)x.rows, m->gate_rows, 2 * E, 0, 1));
        TRY(simple_gemm(x, ws + x.w.cn, D, true, P + m->head_w, m->head_ld, true, ws + x.w.hc, m->ld_h, nullptr, (int)x.rows,
                        m->n_heads, D, 0, 1));
        // everything the tower pyramid needs is queued: the main stream joins HERE; what only the backward reads (transposed
        // weights, dgrad weight images) follows on the side stream beside the tower kernel and is awaited by aread_backward
        ev_towers = x.m->ev[x.ev_next++ % (x.m->n_ev - 1)];
        AR_HIP(hipEventRecord(ev_towers, x.side));
        if (!prepared) {
            m->prep_pending = false;
            if (c->train && cfg.precision == 1) {
                TRY(transpose_weights(x));
                if (tower_fused_bwd_ok(x)) TRY(prepare_wimg(x, 1));   // dgrad images of the tower layers
                AR_HIP(hipEventRecord(m->ev_prep, x.side));
                m->prep_pending = true;
            }
        }
        x.st = main_st;
    }
    // 5.-7. MMoE mix, tower pyramid, heads + fused bagging loss: one launch when the configuration allows it
    AR_HIP(hipStreamWaitEvent(x.st, ev_towers, 0));
    const bool want_gates = c->gate_stats != nullptr && m->gate_rows > 0;
    const bool have_loss = c->y && c->loss_out;
    if (fused_towers) {
        TRY(tower_fused_fwd(x, want_gates));
        if (want_gates) LAUNCH(k_gate_stats, dim3(c->n_seg), dim3(256), ws + x.w.gate_part, m->ld_gt, m->gate_rows, c->gate_stats, x.r);
    } else {
    const LayerL& EL = m->experts.L[m->experts.n_layers - 1];
    Mix0P m0 = {};
    m0.glog = ws + x.w.glogE; m0.ld_g = m->ld_ge; m0.X = ws + x.w.ex[m->experts.n_layers - 1].Act; m0.In0 = ws + x.w.In[0];
    m0.n_t = cfg.n_tower[0]; m0.n_exp = cfg.n_expert; m0.h = EL.out_dim; m0.r = x.r; m0.mp = x.mp;
    LAUNCH(k_mix0, dim3(cdiv(x.rows * m0.n_t * (m0.h / 4), 256)), dim3(256), m0);
    // 6. tower pyramid
    for (int l = 0; l < cfg.n_level; ++l) {
        const StackL& S = m->towers[l];
        if (l > 0) {
            const StackL& Sp = m->towers[l - 1];
            MixLP ml = {};
            ml.glog = ws + x.w.glogT; ml.ld_g = m->ld_gt; ml.goff = m->gate_off[l];
            ml.prev = ws + x.w.tw[l - 1][Sp.n_layers - 1].Act; ml.In = ws + x.w.In[l];
            ml.n_src = cfg.n_tower[l - 1]; ml.n_t = cfg.n_tower[l]; ml.w = S.L[0].in_dim; ml.level = l; ml.mask_off = m->mask_off[l];
            ml.gate_part = want_gates ? ws + x.w.gate_part : nullptr; ml.r = x.r; ml.mp = x.mp;
            LAUNCH(k_mixl, dim3(x.n_tiles * SUB), dim3(256), ml);
        }
        TRY(stack_fwd(x, S, x.w.tw[l], ws + x.w.In[l], l));
    }
    if (want_gates) LAUNCH(k_gate_stats, dim3(c->n_seg), dim3(256), ws + x.w.gate_part, m->ld_gt, m->gate_rows, c->gate_stats, x.r);
    // 7. heads + fused bagging loss
    const int LL = cfg.n_level - 1;
    HeadsP hp = {};
    hp.hc = ws + x.w.hc; hp.lin = ws + x.w.lin; hp.act = ws + x.w.tw[LL][m->towers[LL].n_layers - 1].Act;
    hp.head_w = P + m->head_w; hp.head_ld = m->head_ld; hp.D = D; hp.h = m->h_last; hp.n_heads = m->n_heads; hp.ld_h = m->ld_h;
    hp.z = ws + x.w.z; hp.prob = ws + x.w.prob; hp.dz = ws + x.w.dz; hp.probs_out = c->probs; hp.B = c->B;
    hp.y = c->y; hp.seg_weight = c->seg_weight; hp.dprobs = nullptr;
    hp.loss_part = (c->y && c->loss_out) ? ws + x.w.loss_part : nullptr; hp.level = LL; hp.r = x.r; hp.mp = x.mp;
    LAUNCH(k_heads_fwd, dim3(x.n_tiles * SUB), dim3(256), hp);
    }
    phase_mark(x.st, 2);
    const bool side_tail = have_loss || (c->train && c->update_running);
    m->fwd_tail_deferred = false;
    if (side_tail) {
        if (c->async_tail & 2) {
            // fused step: aread_backward follows on this stream with the same call.  The forward's tail (loss value, running
            // statistics: nothing on the main stream reads them) is queued by the backward on the side stream behind its own
            // first fork -- one event record less on the main stream's dependent chain (each costs it 3-9 us).
            m->fwd_tail_deferred = true;
        } else {
            TRY(fork_side(x));
            TRY(forward_tail(x, fused_towers));
            TRY(join_side(x));
        }
    }
    return AREAD_OK;
}

// ================================================================================================
// backward
// ================================================================================================
static int wgrad(Ctx& x, const float* dY, int64_t ld_dy, int64_t dy_gs, const float* X, int64_t ldx, int64_t x_gs, int G,
                 int M, int N, float* out, int64_t ldo, int64_t o_gs, const uint8_t* active, int64_t slab_off) {
    // dW[M][N] = dY^T X.  A skinny M (a handful of gate / head rows) would waste most of the 64-row MFMA tile:
    // compute the transposed product X^T dY instead (M plays the 16-wide N role) and let the reduction write it back
    // transposed.
    const bool swap = M <= 16 && N > 16 && G == 1;
    const int Mg = swap ? N : M, Ng = swap ? M : N;
    const KSplit ks = wgrad_ksplit(x.rows, G, Mg, Ng);
    GemmP g = {};
    g.A = swap ? X : dY; g.lda = swap ? ldx : ld_dy; g.a_gs = swap ? x_gs : dy_gs;
    g.B = swap ? dY : X; g.ldb = swap ? ld_dy : ldx; g.b_gs = swap ? dy_gs : x_gs;
    g.C = x.ws + slab_off; g.ldc = Ng; g.c_gs = (int64_t)Mg * Ng; g.c_ks = (int64_t)G * Mg * Ng;
    g.M = Mg; g.N = Ng; g.K = (int)x.rows; g.G = G;
    g.k_split = ks.k_split; g.k_chunk = ks.k_chunk;
    g.gate_axis = 2; g.tile_seg = x.r.tile_seg; g.tile_valid = x.r.tile_valid; g.active = active; g.active_ld = MAX_TOWER;
    AR_CHECK_ARG(x.splitk.n < MAX_WGRADS && x.n_pend < MAX_WGRADS, "too many wgrads");
    x.pend[x.n_pend++] = g;                              // launched by the next flush_wgrads (its operands stay in the workspace)
    SplitKOne& d = x.splitk.d[x.splitk.n++];
    d.slab = x.ws + slab_off; d.out = out; d.k_split = ks.k_split; d.G = G; d.M = Mg; d.N = Ng; d.ldo = ldo; d.o_gs = o_gs;
    d.transposed = swap ? 1 : 0;
    return AREAD_OK;
}

// fork once, then every pending weight-gradient GEMM on the side stream.  split-bf16 mode: the row-contiguous operands go
// through the transposing LDS reads (k_gemm_bf3_rc), 3 bf16 MFMA products instead of the fp32 MFMA.
// split-bf16 mode: ONE launch for the whole range (k_gemm_bf3_rc_multi) -- as separate launches the dozen tower / gate / head
// weight gradients were a ~190 us serial chain of 8-22 us kernels on the side stream and the step's tail waited for it
static int g_wgrad_multi = -1;     // AREAD_WGRAD_MULTI=0: one launch per weight gradient (A/B)
static int flush_wgrads_range(Ctx& x, int lo, int hi) {
    if (g_wgrad_multi < 0) { const char* e = getenv("AREAD_WGRAD_MULTI"); g_wgrad_multi = e ? atoi(e) : 1; }
    if (x.m->cfg.precision == 1 && g_wgrad_multi) return launch_gemm_bf3_rc_multi(x.pend + lo, hi - lo, x.side);
    for (int i = lo; i < hi; ++i) {
        if (x.m->cfg.precision == 1) TRY(launch_gemm_bf3_rc(x.pend[i], x.side));
        else TRY(launch_gemm(x.pend[i], false, false, x.side));
    }
    return AREAD_OK;
}
static int flush_wgrads(Ctx& x, bool fork) {
    if (fork) TRY(fork_side(x));
    TRY(flush_wgrads_range(x, 0, x.n_pend));
    x.n_pend = 0;
    return AREAD_OK;
}

static int flush_reductions(Ctx& x) {
    static int merge = -1;                                   // AREAD_REDUCE_MERGE=0: two launches (A/B)
    if (merge < 0) { const char* e = getenv("AREAD_REDUCE_MERGE"); merge = e ? atoi(e) : 1; }
    if (merge && x.splitk.n > 0 && x.bias.n > 0) {           // both pending: one launch
        int64_t mx = 0;
        for (int i = 0; i < x.splitk.n; ++i) {
            const int64_t e = (int64_t)x.splitk.d[i].G * x.splitk.d[i].M * x.splitk.d[i].N;
            if (e > mx) mx = e;
        }
        int bx = cdiv(mx, 128);
        if (bx > 2304) bx = 2304;
        int mc = 0;
        for (int i = 0; i < x.bias.n; ++i) if (x.bias.d[i].ncols > mc) mc = x.bias.d[i].ncols;
        if (cdiv(mc, 16) > bx) bx = cdiv(mc, 16);
        x.bias.r = x.r; x.bias.mp = x.mp;
        LAUNCH(k_reduce_tail_all, dim3(bx, x.splitk.n + x.bias.n), dim3(256), x.splitk, x.bias);
        x.splitk.n = 0; x.bias.n = 0;
        return AREAD_OK;
    }
    if (x.splitk.n > 0) {
        int64_t mx = 0;
        for (int i = 0; i < x.splitk.n; ++i) {
            const int64_t e = (int64_t)x.splitk.d[i].G * x.splitk.d[i].M * x.splitk.d[i].N;
            if (e > mx) mx = e;
        }
        int bx = cdiv(mx, 128);
        if (bx > 2304) bx = 2304;
        LAUNCH(k_splitk_reduce_all, dim3(bx, x.splitk.n), dim3(256), x.splitk);
        x.splitk.n = 0;
    }
    if (x.bias.n > 0) {
        int mx = 0;
        for (int i = 0; i < x.bias.n; ++i) if (x.bias.d[i].ncols > mx) mx = x.bias.d[i].ncols;
        x.bias.r = x.r; x.bias.mp = x.mp;
        LAUNCH(k_bias_reduce_all, dim3(cdiv(mx, 16), x.bias.n), dim3(256), x.bias);
        x.bias.n = 0;
    }
    return AREAD_OK;
}

// one MLP layer backward.  d = dL/dAct on entry (in lw.dAct); on exit it holds dL/dH.
static int layer_bwd(Ctx& x, const LayerL& L, const LayerWs& lw, const float* in, float* d_in, int accumulate_d_in,
                     float* grads, int level, int64_t slab_off, hipEvent_t before_dgrad = nullptr, bool fork_after_act = false) {
    float* ws = x.ws;
    float* d = ws + lw.dAct;
    ActBwdP a = {};
    a.d = d; a.H = ws + lw.H; a.mean = ws + lw.mean; a.rstd = ws + lw.rstd;
    a.gamma = x.params + L.gamma; a.beta = x.params + L.beta; a.bpart = ws + lw.bpart;
    a.ncols = L.ncols; a.h = L.out_dim; a.level = level; a.stack = L.stack; a.layer = L.layer; a.train = x.c->train;
    a.seed = x.c->drop_seed; a.seed_dev = x.c->drop_seed_dev; a.thr = x.thr; a.keep_scale = x.keep_scale; a.r = x.r; a.mp = x.mp;
    // wide layers: both passes in one launch with the segment sums handed off inside the kernel (k_act_bn_bwd); the
    // workgroups that wait for each other sit next to each other in dispatch order, 2 x n_tiles resident ones suffice
    // With the drain + counter + poll hand-off this was +40 us per step (2432 workgroups stalling ~5 us each); with data-tagged
    // granules it is -12 us: on by default (AREAD_FUSED_ACT_BN=0 / aread_debug_set("fused_act_bn", 0) for the two-pass path)
    int& fused_ab = g_fused_act_bn;
    if (fused_ab < 0) { const char* e = getenv("AREAD_FUSED_ACT_BN"); fused_ab = e ? atoi(e) : 1; }
    const int n_chunks = cdiv(L.ncols, 64);
    if (fused_ab && lw.tag_b >= 0 && x.w.tf_sync >= 0 && level < 0 && x.n_tiles <= 700 && x.m->ab_tags_clean) {
        ActBnBwdP q = {};
        q.a = a; q.cpart = ws + lw.cpart;
        q.tags = (tf_u64*)(ws + lw.tag_b);
        q.fin = (tf_u64*)(ws + lw.fin_b);
        q.two_hop_nt = g_two_hop_nt_ab >= 0 ? g_two_hop_nt_ab : two_hop_nt();
        q.err = (unsigned*)(ws + x.w.tf_sync) + AREAD_MAX_LEVEL * AREAD_MAX_LAYER * MAX_SEG;
        LAUNCH(k_act_bn_bwd, dim3(x.n_tiles, n_chunks), dim3(256), q);
    } else {
    LAUNCH(k_act_bwd, dim3(x.n_tiles, cdiv(L.ncols, 64)), dim3(256), a);
    BnBwdApplyP b = {};
    b.d = d; b.H = ws + lw.H; b.mean = ws + lw.mean; b.rstd = ws + lw.rstd; b.gamma = x.params + L.gamma; b.bpart = ws + lw.bpart;
    b.cpart = ws + lw.cpart; b.ncols = L.ncols; b.h = L.out_dim; b.level = level; b.train = x.c->train; b.r = x.r; b.mp = x.mp;
    LAUNCH(k_bn_bwd_apply, dim3(x.n_tiles, cdiv(L.ncols, 64)), dim3(256), b);
    }
    AR_CHECK_ARG(x.bias.n < MAX_BN_LAYERS_DECL, "too many layers");
    BiasOne& bo = x.bias.d[x.bias.n++];
    bo.cpart = ws + lw.cpart; bo.bpart = ws + lw.bpart; bo.db = grads + L.b; bo.dgamma = grads + L.gamma; bo.dbeta = grads + L.beta;
    bo.ncols = L.ncols; bo.h = L.out_dim; bo.level = level;
    const bool shared = L.in_gs == 0 && L.G > 1;
    const uint8_t* act = (level >= 0 && L.G > 1) ? level_active(x, level) : nullptr;
    if (fork_after_act) TRY(fork_side(x));         // dH is final: the side stream's weight gradients may start beside the dgrad
    // dgrad: d_in = dH W
    if (d_in && before_dgrad) AR_HIP(hipStreamWaitEvent(x.st, before_dgrad, 0));   // (whoever initialises d_in for an accumulating dgrad)
    if (d_in) {
        GemmP g = {};
        g.A = d; g.lda = L.ncols; g.a_gs = L.out_dim;
        g.B = x.params + L.w; g.ldb = L.in_dim; g.b_gs = (int64_t)L.out_dim * L.in_dim;
        g.C = d_in; g.ldc = L.in_ld; g.c_gs = L.in_gs;
        g.M = (int)x.rows; g.N = L.in_dim; g.K = L.out_dim; g.G = L.G; g.accumulate = accumulate_d_in;
        if (shared || L.G == 1) { g.K = L.ncols; g.G = 1; g.a_gs = 0; g.b_gs = 0; g.c_gs = 0; }
        g.gate_axis = 1; g.tile_seg = x.r.tile_seg; g.tile_valid = x.r.tile_valid;
        if (x.m->cfg.precision == 1) {         // k-contiguous B from the transposed weight copy [g][in][out]
            g.B = x.ws + lw.wT;
            g.ldb = (shared || L.G == 1) ? L.ncols : L.out_dim;
            g.b_gs = (shared || L.G == 1) ? 0 : (int64_t)L.in_dim * L.out_dim;
                    TRY(launch_gemm_bf3(g, x.st));
        } else TRY(launch_gemm(g, true, false, x.st));   // inactive towers contribute dH = 0
    }
    // wgrad: dW = dH^T in
    if (shared) TRY(wgrad(x, d, L.ncols, 0, in, L.in_ld, 0, 1, L.ncols, L.in_dim, grads + L.w, L.in_dim, 0, nullptr, slab_off));
    else TRY(wgrad(x, d, L.ncols, L.out_dim, in, L.in_ld, L.in_gs, L.G, L.out_dim, L.in_dim, grads + L.w, L.in_dim,
                   (int64_t)L.out_dim * L.in_dim, act, slab_off));
    return AREAD_OK;
}

template <int NC, int NV>
static void launch_rowwise_bwd2(Ctx& x, const RowwiseBwdP& p) {
    hipLaunchKernelGGL((k_rowwise_bwd<NC, NV>), dim3(x.n_tiles * RWB_SUB), dim3(256), 0, x.st, p);
}
template <int NV>
static void launch_rowwise_bwd(Ctx& x, const RowwiseBwdP& p) {
    switch (p.n_cross) {
        case 0: launch_rowwise_bwd2<0, NV>(x, p); break;
        case 1: launch_rowwise_bwd2<1, NV>(x, p); break;
        case 2: launch_rowwise_bwd2<2, NV>(x, p); break;
        case 3: launch_rowwise_bwd2<3, NV>(x, p); break;
        default: launch_rowwise_bwd2<4, NV>(x, p); break;
    }
}

extern "C" int aread_backward(const aread_model* m, const aread_call* c, const float* e_in, const float* dprobs,
                              float* grads, float* de_out, void* stream) {
    Ctx x;
    TRY(make_ctx(m, c, stream, &x));
    AR_CHECK_ARG(e_in && grads && de_out, "aread_backward: null e_in/grads/de_out");
    AR_CHECK_ARG(dprobs || c->y, "aread_backward: need dprobs or labels");
    AR_CHECK_ARG(((uintptr_t)grads & 15) == 0 && ((uintptr_t)de_out & 15) == 0 && ((uintptr_t)c->de_rw & 15) == 0, "aread_backward: alignment");
    const aread_model_cfg& cfg = m->cfg;
    const int D = m->D, E = m->E, LL = cfg.n_level - 1;
    float* ws = x.ws;
    const float* P = x.params;
    phase_mark(x.st, 3);
    if (g_fused_act_bn < 0) { const char* e = getenv("AREAD_FUSED_ACT_BN"); g_fused_act_bn = e ? atoi(e) : 1; }
    // Stream plan.  Every event record / cross-stream wait on the MAIN stream costs its dependent chain 3-9 us
    // (profiles/r02_event_cost.txt, r03_step_anatomy.txt), so the main stream records exactly two events here (after the
    // tower backward; after the first expert layer's act/BN backward) and waits for one (the row-wise backward):
    //   main  : [dz] -> tower backward -> R(ev_b1) -> experts L(n-1)..L1 -> L0 act/BN backward -> R(fork) -> L0 dgrad -> W(ev_rw)
    //   side  : [forward's deferred tail] -> grads init -> dcn GEMM -> gate-input GEMMs -> R'(ev_gates) -> weight gradients A (one
    //           launch) | after the fork: weight gradients B (one launch) -> reductions -> waits for side2's last kernel
    //   side2 : table L2 sweep (behind the side stream's position at entry) | W'(ev_gates) -> row-wise backward -> R'(ev_rw) ->
    //           its parameter-gradient finish, group-embedding and gate-bias reductions
    // Issue order = the order in which a captured graph schedules independent branches (and the order a just-in-time host
    // feeds the queues): the main stream's dependent chain is issued FIRST in every section.
    auto mark_main = [&](hipEvent_t* ev) -> int {
        *ev = x.m->ev[x.ev_next++ % (x.m->n_ev - 1)];
        AR_HIP(hipEventRecord(*ev, x.st));
        return AREAD_OK;
    };
    const hipStream_t main_st = x.st;
    if (m->prep_pending) {                                   // transposed weights / dgrad images of this step's forward (side stream)
        AR_HIP(hipStreamWaitEvent(x.st, m->ev_prep, 0));
        m->prep_pending = false;
    }
    // 1. dz
    HeadsP hp = {};
    hp.prob = ws + x.w.prob; hp.dz = ws + x.w.dz; hp.n_heads = m->n_heads; hp.ld_h = m->ld_h; hp.B = c->B;
    hp.y = c->y; hp.seg_weight = c->seg_weight; hp.dprobs = dprobs; hp.level = LL; hp.r = x.r; hp.mp = x.mp;
    if (dprobs || !c->y) LAUNCH(k_heads_dz, dim3(x.n_tiles * SUB), dim3(256), hp);   // else: fused into k_heads_fwd / k_tower_fwd
    const bool tail_deferred = m->fwd_tail_deferred;         // fused step: the forward left its loss / running-statistics tail to us
    m->fwd_tail_deferred = false;
    const bool fused_bwd = c->train && tower_fused_bwd_ok(x);
    const int nle = m->experts.n_layers;
    AR_CHECK_ARG(m->n_heads * m->h_last <= 1024, "aread_backward: n_heads*h_last too large");
    // the caller's embedding-table L2 sweep, released now on the second side stream behind the side stream's position at
    // entry (the forward's side prologue): it runs beside the tower kernels (chains of latency-bound phases that leave the
    // HBM idle) and is complete on `stream` when this call returns (W(ev_rw) below covers it)
    if (c->l2_table) {
        hipEvent_t ev_s0 = x.m->ev[x.ev_next++ % (x.m->n_ev - 1)];
        AR_HIP(hipEventRecord(ev_s0, x.side));
        AR_HIP(hipStreamWaitEvent(m->side2, ev_s0, 0));
        TRY(aread_l2_table_throttled(c->l2_table, c->l2_n, c->l2_coef, 1.0f, c->l2_grad, c->l2_partial, c->l2_workgroups, m->side2));
        TRY(aread_l2_finish(c->l2_partial, aread_l2_partials(), c->l2_coef, c->l2_reg_out, 0, m->side2));
    }
    // dV[:, :D] = dz^T cn: queued for the side stream
    TRY(wgrad(x, ws + x.w.dz, m->ld_h, 0, ws + x.w.cn, D, 0, 1, m->n_heads, D, grads + m->head_w, m->head_ld, 0, nullptr, x.w.slab_head));
    phase_mark(x.st, 4);
    if (fused_bwd) {
        // 2.-4. heads backward, tower pyramid, gate-mix and MMoE-mix backward: one launch
        TRY(tower_fused_bwd(x, grads));
    } else {
    // 2. heads backward
    const LayerWs& last = x.w.tw[LL][m->towers[LL].n_layers - 1];
    HeadsBwdP hb = {};
    hb.dz = ws + x.w.dz; hb.act = ws + last.Act; hb.head_w = P + m->head_w; hb.head_ld = m->head_ld; hb.D = D; hb.h = m->h_last;
    hb.n_heads = m->n_heads; hb.ld_h = m->ld_h; hb.dact = ws + last.dAct; hb.dlin = ws + x.w.dlin;
    hb.part = ws + x.w.misc_part; hb.ldp = 1024; hb.r = x.r;
    LAUNCH(k_heads_bwd, dim3(x.n_tiles * SUB), dim3(256), hb);
    // 3. tower pyramid, top down (the weight gradients queue up)
    for (int l = LL; l >= 0; --l) {
        const StackL& S = m->towers[l];
        for (int j = S.n_layers - 1; j >= 0; --j) {
            const float* in = j == 0 ? ws + x.w.In[l] : ws + x.w.tw[l][j - 1].Act;
            float* d_in = j == 0 ? ws + x.w.dIn[l] : ws + x.w.tw[l][j - 1].dAct;
            TRY(layer_bwd(x, S.L[j], x.w.tw[l][j], in, d_in, 0, grads, l, x.w.slab_tw[l][j]));
        }
        if (l > 0) {
            const StackL& Sp = m->towers[l - 1];
            MixLBwdP mb = {};
            mb.glog = ws + x.w.glogT; mb.dglog = ws + x.w.dglogT; mb.ld_g = m->ld_gt; mb.goff = m->gate_off[l];
            mb.prev = ws + x.w.tw[l - 1][Sp.n_layers - 1].Act; mb.dIn = ws + x.w.dIn[l];
            mb.dprev = ws + x.w.tw[l - 1][Sp.n_layers - 1].dAct;
            mb.n_src = cfg.n_tower[l - 1]; mb.n_t = cfg.n_tower[l]; mb.w = S.L[0].in_dim; mb.level = l; mb.mask_off = m->mask_off[l];
            mb.r = x.r; mb.mp = x.mp;
            LAUNCH(k_mixl_bwd, dim3(x.n_tiles * SUB), dim3(256), mb);
        }
    }
    // 4. MMoE mix backward
    const LayerL& EL = m->experts.L[nle - 1];
    Mix0BwdP m0 = {};
    m0.glog = ws + x.w.glogE; m0.dglog = ws + x.w.dglogE; m0.ld_g = m->ld_ge; m0.X = ws + x.w.ex[nle - 1].Act;
    m0.dU = ws + x.w.dIn[0]; m0.dX = ws + x.w.ex[nle - 1].dAct; m0.n_t = cfg.n_tower[0]; m0.n_exp = cfg.n_expert; m0.h = EL.out_dim;
    m0.r = x.r; m0.mp = x.mp;
    LAUNCH(k_mix0_bwd, dim3(x.n_tiles * SUB), dim3(256), m0);
    }
    phase_mark(x.st, 5);
    // gate weight gradients (queued), then the mark the side stream's first batch waits for
    const int n_ge = cfg.n_tower[0] * cfg.n_expert;
    AR_CHECK_ARG(m->gate_rows <= 1024 && n_ge <= 1024, "aread_backward: too many gate rows");
    TRY(wgrad(x, ws + x.w.dglogE, m->ld_ge, 0, e_in, D, 0, 1, n_ge, D, grads + m->gate_w, D, 0, nullptr, x.w.slab_gate));
    if (m->gate_rows > 0)
        TRY(wgrad(x, ws + x.w.dglogT, m->ld_gt, 0, ws + x.w.q, 2 * E, 0, 1, m->gate_rows, 2 * E, grads + m->tgate_w, 2 * E, 0, nullptr,
                  x.w.slab_tgate));
    const int n_pend_a = x.n_pend;                           // weight gradients whose operands are final at ev_b1
    hipEvent_t ev_b1;
    TRY(mark_main(&ev_b1));
    phase_mark(x.st, 6);
    // 5a. experts, all layers but the first: main stream, issued before any side work
    for (int j = nle - 1; j >= 1; --j)
        TRY(layer_bwd(x, m->experts.L[j], x.w.ex[j], ws + x.w.ex[j - 1].Act, ws + x.w.ex[j - 1].dAct, 0, grads, -1, x.w.slab_ex[j]));
    const int n_pend_b = x.n_pend;                           // ... and the deeper expert layers' (their dH is final here)
    hipEvent_t ev_b2 = nullptr;
    if (n_pend_b > n_pend_a) TRY(mark_main(&ev_b2));         // (one record: their weight gradients leave the side stream's tail)
    // ---- side stream, batch A ------------------------------------------------------------------------------------------------
    x.st = x.side;
    AR_HIP(hipStreamWaitEvent(x.side, ev_b1, 0));      // (the side stream's first wait of this call: dz, dglog*, dH of the towers are final)
    if (!c->grads_init) AR_HIP(hipMemsetAsync(grads, 0, (size_t)m->n_params * sizeof(float), x.st));   // every reduction below ADDS to grads
    hipEvent_t ev_gates = x.m->ev[x.ev_next++ % (x.m->n_ev - 1)];   // side2's cue: behind ev_b1 and behind the gradient buffer's initialisation
    AR_HIP(hipEventRecord(ev_gates, x.side));
    TRY(flush_wgrads_range(x, 0, n_pend_a));             // head, tower and gate weight gradients: one launch
    if (ev_b2) {
        AR_HIP(hipStreamWaitEvent(x.side, ev_b2, 0));
        TRY(flush_wgrads_range(x, n_pend_a, n_pend_b));  // expert layers n-1 .. 1: one launch, beside the first layer's act/BN backward
    }
    x.st = main_st;
    // ---- side2: row-wise trunk backward beside the expert backward.  It needs dcn / dlin and dq / deg (all ordered behind
    // ev_gates on the side stream) and WRITES its share of dL/de: into c->de_rw when the caller gave one (the embedding
    // backward then reads de_out + de_rw and the first expert layer's dgrad does not depend on this chain at all), else into
    // de_out, onto which that dgrad accumulates behind ev_rw.
    // (No stream ever waits on its own event and no two forked streams wait on each other: hipStreamEndCapture walks the
    // fork relation recursively and does not terminate on such a cycle -- side2 waits on side, side waits on side2 only at
    // the very end, after side2's last wait on side.)
    hipEvent_t ev_rw = x.m->ev[x.ev_next++ % (x.m->n_ev - 1)];
    const hipStream_t rw_stream = m->side2;
    float* de_rw = c->de_rw ? c->de_rw : de_out;
    {
        AR_HIP(hipStreamWaitEvent(rw_stream, ev_gates, 0));
        x.st = rw_stream;
        // dcn = dz V[:, :D]; gate-input gradients dq = dglogT Tw (tower gates) and deg = dglogE Gw (MMoE gates): the row-wise
        // backward's inputs, on ITS stream (the side stream carries the weight gradients at the same time)
        TRY(simple_gemm(x, ws + x.w.dz, m->ld_h, true, P + m->head_w, m->head_ld, false, ws + x.w.dcn, D, nullptr, (int)x.rows, D,
                        m->n_heads, 0, 1));
        if (m->gate_rows == 0) AR_HIP(hipMemsetAsync(ws + x.w.dq, 0, (size_t)x.rows * 2 * E * sizeof(float), x.st));
        if (m->gate_rows > 0)
            TRY(simple_gemm(x, ws + x.w.dglogT, m->ld_gt, true, P + m->tgate_w, 2 * E, false, ws + x.w.dq, 2 * E, nullptr, (int)x.rows,
                            2 * E, m->gate_rows, 0, 1));
        TRY(simple_gemm(x, ws + x.w.dglogE, m->ld_ge, true, P + m->gate_w, D, false, ws + x.w.deg, D, nullptr, (int)x.rows, D,
                        n_ge, 0, 1));
        RowwiseBwdP rb = {};
        rb.e = e_in; rb.xw = ws + x.w.xw; rb.dcn = ws + x.w.dcn; rb.dlin = ws + x.w.dlin; rb.dq = ws + x.w.dq; rb.deg = ws + x.w.deg;
        rb.lin_w = P + m->lin_w; rb.cn_w = P + m->cn_w; rb.cn_b = P + m->cn_b;
        rb.de = de_rw; rb.part = ws + x.w.rw_part; rb.part_ld = (int64_t)(cfg.n_cross + 2) * D + 8; rb.dgrp_part = ws + x.w.dgrp_part;
        rb.D = D; rb.E = E; rb.n_cross = cfg.n_cross; rb.dom_field = cfg.domain_field; rb.rows = x.rows; rb.de_init = 1; rb.r = x.r;
        AR_CHECK_ARG(D <= 1024 && cfg.n_cross <= MAX_CROSS, "aread_backward: D=%d > 1024 or too many cross layers", D);
        if (D <= 384) launch_rowwise_bwd<3>(x, rb); else if (D <= 768) launch_rowwise_bwd<6>(x, rb); else launch_rowwise_bwd<8>(x, rb);
        AR_LAUNCH_CHECK();
        AR_HIP(hipEventRecord(ev_rw, rw_stream));
        RowwiseFinP rf = {};
        rf.part = ws + x.w.rw_part; rf.ld = rb.part_ld; rf.cn_w = P + m->cn_w; rf.cn_b = P + m->cn_b;
        rf.g_cn_w = grads + m->cn_w; rf.g_cn_b = grads + m->cn_b; rf.g_lin_w = grads + m->lin_w; rf.g_lin_b = grads + m->lin_b;
        rf.D = D; rf.n_cross = cfg.n_cross; rf.r = x.r;
        switch (cfg.n_cross) {
            case 0: LAUNCH(k_rowwise_finish<0>, dim3(cdiv(D, 8)), dim3(256), rf); break;
            case 1: LAUNCH(k_rowwise_finish<1>, dim3(cdiv(D, 8)), dim3(256), rf); break;
            case 2: LAUNCH(k_rowwise_finish<2>, dim3(cdiv(D, 8)), dim3(256), rf); break;
            case 3: LAUNCH(k_rowwise_finish<3>, dim3(cdiv(D, 8)), dim3(256), rf); break;
            default: LAUNCH(k_rowwise_finish<4>, dim3(cdiv(D, 8)), dim3(256), rf); break;
        }
        LAUNCH(k_seg_reduce, dim3(c->n_seg, cdiv(E, 16)), dim3(256), ws + x.w.dgrp_part, (int64_t)E, E, ws + x.w.grp, RWB_SUB, x.r);
        LAUNCH(k_grp_bwd, dim3(1), dim3(256), ws + x.w.grp, grads + m->group_emb, cfg.n_tower[0], E, x.r, x.mp);
        // head tails (partials of the heads backward) and the gate biases (column sums of the gate-logit gradients): two
        // launches here, behind the row-wise chain (nothing waits for them before the end of the call), instead of five
        // serial ones in front of the weight gradients on the side stream
        {
            ColsumAllP ca = {};
            ca.d[ca.n++] = ColsumOne{ws + x.w.dglogE, (int64_t)m->ld_ge, n_ge, ws + x.w.cs_part_e, (int64_t)m->ld_ge};
            if (m->gate_rows > 0) ca.d[ca.n++] = ColsumOne{ws + x.w.dglogT, (int64_t)m->ld_gt, m->gate_rows, ws + x.w.cs_part_t, (int64_t)m->ld_gt};
            ca.r = x.r;
            LAUNCH(k_colsum_all, dim3(x.n_tiles, ca.n), dim3(256), ca);
            ReduceAllP ra = {};
            ra.d[ra.n++] = ReduceOne{ws + x.w.misc_part, (int64_t)1024, m->n_heads * m->h_last, grads + m->head_w + D, m->h_last, (int64_t)m->head_ld, SUB};
            ra.d[ra.n++] = ReduceOne{ws + x.w.cs_part_e, (int64_t)m->ld_ge, n_ge, grads + m->gate_b, n_ge, (int64_t)0, 1};
            if (m->gate_rows > 0) ra.d[ra.n++] = ReduceOne{ws + x.w.cs_part_t, (int64_t)m->ld_gt, m->gate_rows, grads + m->tgate_b, m->gate_rows, (int64_t)0, 1};
            ra.r = x.r;
            int mxc = m->n_heads * m->h_last;
            if (n_ge > mxc) mxc = n_ge;
            if (m->gate_rows > mxc) mxc = m->gate_rows;
            LAUNCH(k_reduce_tiles_all, dim3(cdiv(mxc, 32), ra.n), dim3(256), ra);
        }
        // the forward's deferred tail (loss value, running statistics): this stream has the slack (the side stream's last
        // reductions decide when the step's parameter gradients are complete)
        if (tail_deferred) {
            const hipStream_t keep_side = x.side;
            x.side = rw_stream;
            const int rc = forward_tail(x, tower_fused_ok(x));
            x.side = keep_side;
            if (rc != AREAD_OK) return rc;
        }
        x.st = main_st;
    }
    // 5b. the first expert layer: its input gradient goes to de_out; without a separate de_rw it ADDS onto what the row-wise
    // backward (side2) has written there
    TRY(layer_bwd(x, m->experts.L[0], x.w.ex[0], e_in, de_out, c->de_rw ? 0 : 1, grads, -1, x.w.slab_ex[0], c->de_rw ? nullptr : ev_rw,
                  true));
    phase_mark(x.st, 7);
    if (c->de_rw) AR_HIP(hipStreamWaitEvent(x.st, ev_rw, 0));   // (long finished: de_rw and the table L2 sweep are complete on `stream`)
    // ---- side stream, batch B (its fork was recorded inside layer_bwd, right behind the act/BN backward and BEFORE the dgrad):
    // every expert layer's weight gradient in one launch, then the reductions of everything
    x.st = x.side;
    TRY(flush_wgrads_range(x, ev_b2 ? n_pend_b : n_pend_a, x.n_pend));   // the first expert layer's weight gradient
    x.n_pend = 0;
    TRY(flush_reductions(x));
    x.st = main_st;
    // de_out (and de_rw) are complete on the main stream here; the parameter gradients complete on the two side streams
    // (side: weight / bias / BatchNorm gradients; side2: row-wise trunk, group embedding, gate biases, head tails).
    // async_tail: the caller overlaps its own work (embedding scatter) and calls aread_join() afterwards.
    // (side never waits on side2 -- side2 has waited on side, and hipStreamEndCapture does not terminate on such a cycle:
    // tests/test_gpu_graph.py -- so the join makes the main stream wait for both.)
    m->side2_pending = true;
    if (!(c->async_tail & 1)) TRY(aread_join(m, stream));
    return AREAD_OK;
}

extern "C" int aread_debug_set(const char* key, int value) {
    AR_CHECK_ARG(key != nullptr, "aread_debug_set: null key");
    if (!strcmp(key, "fused_towers")) g_fused_mode = value;
    else if (!strcmp(key, "fused_towers_bwd")) g_fused_bwd = value;
    else if (!strcmp(key, "fused_act_bn")) g_fused_act_bn = value;
    else if (!strcmp(key, "two_hop_nt")) g_two_hop_nt = value;
    else if (!strcmp(key, "two_hop_nt_act_bn")) g_two_hop_nt_ab = value;
    else if (!strcmp(key, "plan_single")) g_plan_single = value;
    else if (!strcmp(key, "tf_stamps")) g_tf_stamps = value;
    else if (!strcmp(key, "phase_events")) g_phase_on = value;
    else AR_CHECK_ARG(false, "aread_debug_set: unknown key %s", key);
    if (g_n_cu == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&g_n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) g_n_cu = 0;
    }
    return AREAD_OK;
}

extern "C" long long aread_debug_get(const char* key) {
    if (!key) return -1;
    if (!strcmp(key, "fused_fwd_calls")) return g_fused_fwd_calls;
    if (!strcmp(key, "fused_bwd_calls")) return g_fused_bwd_calls;
    return -1;
}

extern "C" int aread_join(const aread_model* m, void* stream) {
    AR_CHECK_ARG(m != nullptr, "aread_join: null model");
    if (!m->side) return AREAD_OK;
    hipEvent_t e = m->ev[m->n_ev - 1];
    AR_HIP(hipEventRecord(e, m->side));
    AR_HIP(hipStreamWaitEvent((hipStream_t)stream, e, 0));
    if (m->side2_pending) {
        AR_HIP(hipEventRecord(m->ev_join2, m->side2));
        AR_HIP(hipStreamWaitEvent((hipStream_t)stream, m->ev_join2, 0));
        m->side2_pending = false;
    }
    return AREAD_OK;
}

// ================================================================================================
// stand-alone MLP block (MultiLayerPerceptron, layer.py:203-229) on the same layer kernels
// ================================================================================================
// dst[r][0..cols) = r < B ? src[r][0..cols) : 0     (ld_dst = ld_src = cols unless given)
__global__ __launch_bounds__(256) void k_pad_rows(const float* src, int64_t ld_src, float* dst, int64_t ld_dst, int cols, int64_t B,
                                                  int64_t rows) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * cols) return;
    const int64_t r = idx / cols;
    const int c = (int)(idx - r * cols);
    dst[r * ld_dst + c] = r < B ? src[r * ld_src + c] : 0.f;
}

static int make_mlp_ctx(const aread_model* m, const aread_mlp_call* c, void* stream, Ctx* x, aread_call* fake) {
    AR_CHECK_ARG(m && m->is_mlp && c, "aread_mlp: not an MLP handle");
    AR_CHECK_ARG(c->B > 0 && c->plan && c->params && c->ws && c->stats, "aread_mlp: null plan/params/stats/ws or B <= 0");
    AR_CHECK_ARG(((uintptr_t)c->ws & 255) == 0 && ((uintptr_t)c->params & 15) == 0, "aread_mlp: alignment");
    memset(fake, 0, sizeof(*fake));
    fake->B = c->B; fake->n_seg = 1; fake->mode = 1; fake->train = c->train; fake->update_running = c->update_running;
    fake->drop_seed = c->drop_seed; fake->plan = c->plan; fake->params = c->params; fake->stats = c->stats; fake->nbt = c->nbt;
    fake->ws = c->ws;
    x->m = m; x->c = fake; x->st = (hipStream_t)stream;
    mlp_ws_layout(m, c->B, &x->w);
    x->ws = (float*)c->ws;
    x->rows = x->w.max_rows;
    x->n_tiles = (int)x->w.n_tiles;
    const PlanView pv = plan_view(c->plan, c->B, 1);
    x->r.tile_seg = pv.tile_seg; x->r.tile_valid = pv.tile_valid; x->r.row_sample = pv.row_sample;
    x->r.seg_count = pv.seg_count; x->r.seg_start = pv.seg_start; x->r.hdr = pv.hdr; x->r.n_tiles = x->n_tiles; x->r.n_seg = 1;
    int32_t* ints = (int32_t*)(x->ws + x->w.kact);
    x->mp.active = (const uint8_t*)(x->ws + x->w.active);
    x->mp.kact = ints; x->mp.n0act = ints + MAX_SEG; x->mp.seg_dom = (int32_t*)(x->ws + x->w.seg_dom);
    x->mp.masks = nullptr; x->mp.edge_count = 0; x->mp.mode = 1;
    const bool drop = c->train && m->cfg.dropout > 0.f;
    x->thr = drop ? drop_threshold(m->cfg.dropout) : 0u;
    x->keep_scale = drop ? 1.0f / (1.0f - m->cfg.dropout) : 1.f;
    x->params = c->params;
    x->splitk.n = 0; x->bias.n = 0; x->n_pend = 0;
    TRY0(model_streams_init(m));
    x->side = m->side; x->ev_next = 0;
    return AREAD_OK;
}

extern "C" int aread_mlp_forward(const aread_model* m, const aread_mlp_call* c, const float* xin, float* out, void* stream) {
    Ctx x; aread_call fake;
    TRY(make_mlp_ctx(m, c, stream, &x, &fake));
    AR_CHECK_ARG(xin && out, "aread_mlp_forward: null x/out");
    AR_CHECK_ARG(!(c->train && c->update_running) || c->nbt, "aread_mlp_forward: update_running needs nbt");
    float* ws = x.ws;
    const int in = m->mlp_in, nl = m->experts.n_layers, last = m->h_last;
    LAUNCH(k_pad_rows, dim3(cdiv(x.rows * in, 256)), dim3(256), xin, (int64_t)in, ws + x.w.In[0], (int64_t)in, in, c->B, x.rows);
    if (c->train && m->cfg.precision == 1) TRY(transpose_weights(x));   // for the dgrad of aread_mlp_backward
    TRY(stack_fwd(x, m->experts, x.w.ex, ws + x.w.In[0], -1));
    const float* act = ws + x.w.ex[nl - 1].Act;
    if (m->mlp_out_layer) {
        TRY(simple_gemm(x, act, last, true, x.params + m->out_w, last, true, ws + x.w.hc, 4, x.params + m->out_b, (int)x.rows, 1, last, 0, 1));
        LAUNCH(k_pad_rows, dim3(cdiv(c->B, 256)), dim3(256), ws + x.w.hc, (int64_t)4, out, (int64_t)1, 1, c->B, c->B);
    } else {
        LAUNCH(k_pad_rows, dim3(cdiv(c->B * last, 256)), dim3(256), act, (int64_t)last, out, (int64_t)last, last, c->B, c->B);
    }
    if (c->train && c->update_running) {
        BnRunAllP all = {};
        int max_cols = 0;
        for (int j = 0; j < nl; ++j) {
            const LayerL& L = m->experts.L[j];
            BnRunP& p = all.L[all.n_layers++];
            p.mean = ws + x.w.ex[j].mean; p.var = ws + x.w.ex[j].var; p.rmean = c->stats + L.rmean; p.rvar = c->stats + L.rvar;
            p.nbt = c->nbt + L.nbt0; p.ncols = L.ncols; p.h = L.out_dim; p.level = -1;
            if (L.ncols > max_cols) max_cols = L.ncols;
        }
        all.r = x.r; all.mp = x.mp;
        LAUNCH(k_bn_running, dim3(cdiv(max_cols, 256), all.n_layers), dim3(256), all);
    }
    return AREAD_OK;
}

extern "C" int aread_mlp_backward(const aread_model* m, const aread_mlp_call* c, const float* xin, const float* dout, float* grads,
                                  float* dx, void* stream) {
    Ctx x; aread_call fake;
    TRY(make_mlp_ctx(m, c, stream, &x, &fake));
    AR_CHECK_ARG(xin && dout && grads, "aread_mlp_backward: null x/dout/grads");
    float* ws = x.ws;
    const int in = m->mlp_in, nl = m->experts.n_layers, last = m->h_last;
    AR_HIP(hipMemsetAsync(grads, 0, (size_t)m->n_params * sizeof(float), x.st));
    float* dact_last = ws + x.w.ex[nl - 1].dAct;
    if (m->mlp_out_layer) {
        // dz [rows, 1] padded; dAct = dz * w_out ; dW_out = dz^T Act ; db_out = sum dz
        LAUNCH(k_pad_rows, dim3(cdiv(x.rows, 256)), dim3(256), dout, (int64_t)1, ws + x.w.dz, (int64_t)4, 1, c->B, x.rows);
        TRY(simple_gemm(x, ws + x.w.dz, 4, true, x.params + m->out_w, last, false, dact_last, last, nullptr, (int)x.rows, last, 1, 0, 1));
        TRY(wgrad(x, ws + x.w.dz, 4, 0, ws + x.w.ex[nl - 1].Act, last, 0, 1, 1, last, grads + m->out_w, last, 0, nullptr, x.w.slab_head));
        LAUNCH(k_colsum, dim3(x.n_tiles), dim3(256), ws + x.w.dz, (int64_t)4, 1, ws + x.w.misc_part, (int64_t)1024, x.r);
        LAUNCH(k_reduce_tiles, dim3(1), dim3(256), ws + x.w.misc_part, (int64_t)1024, 1, grads + m->out_b, 1, (int64_t)0, 0, 1, x.r);
    } else {
        LAUNCH(k_pad_rows, dim3(cdiv(x.rows * last, 256)), dim3(256), dout, (int64_t)last, dact_last, (int64_t)last, last, c->B, x.rows);
    }
    for (int j = nl - 1; j >= 0; --j) {
        const float* inp = j == 0 ? ws + x.w.In[0] : ws + x.w.ex[j - 1].Act;
        float* d_in = j == 0 ? (dx ? ws + x.w.dIn[0] : nullptr) : ws + x.w.ex[j - 1].dAct;
        TRY(layer_bwd(x, m->experts.L[j], x.w.ex[j], inp, d_in, 0, grads, -1, x.w.slab_ex[j]));
    }
    if (dx) LAUNCH(k_pad_rows, dim3(cdiv(c->B * in, 256)), dim3(256), ws + x.w.dIn[0], (int64_t)in, dx, (int64_t)in, in, c->B, c->B);
    const hipStream_t main_st = x.st;
    TRY(flush_wgrads(x, true));
    x.st = x.side;
    TRY(flush_reductions(x));
    x.st = main_st;
    TRY(join_side(x));
    return AREAD_OK;
}

extern "C" int aread_l2_dense_total(const float* params, const float* coef, int64_t n, float* grads, float* loss_out,
                                    int accumulate, const float* loss_in, float* total_out, void* stream) {
    AR_CHECK_ARG(params && coef && loss_out && n > 0, "aread_l2_dense: bad arguments");
    // the block partials live at the tail of loss_out's caller-provided scratch: loss_out[1..256]; the last block finishes
    hipLaunchKernelGGL(k_l2_dense, dim3(256), dim3(256), 0, (hipStream_t)stream, params, coef, n, grads, loss_out + 1, loss_out,
                       accumulate, loss_in, total_out, 0);
    AR_LAUNCH_CHECK();
    return AREAD_OK;
}
extern "C" int aread_l2_dense_init(const float* params, const float* coef, int64_t n, float* grads, float* reg_out, void* stream) {
    AR_CHECK_ARG(params && coef && grads && reg_out && n > 0, "aread_l2_dense_init: bad arguments");
    hipLaunchKernelGGL(k_l2_dense, dim3(256), dim3(256), 0, (hipStream_t)stream, params, coef, n, grads, reg_out + 1, reg_out,
                       0, (const float*)nullptr, (float*)nullptr, 1);
    AR_LAUNCH_CHECK();
    return AREAD_OK;
}
extern "C" int aread_step_total(const float* loss, const float* reg_dense, float* reg, float* total, void* stream) {
    AR_CHECK_ARG(loss && reg_dense && reg && total, "aread_step_total: null pointer");
    hipLaunchKernelGGL(k_step_total, dim3(1), dim3(64), 0, (hipStream_t)stream, loss, reg_dense, reg, total);
    AR_LAUNCH_CHECK();
    return AREAD_OK;
}
extern "C" int aread_l2_dense(const float* params, const float* coef, int64_t n, float* grads, float* loss_out,
                              int accumulate, void* stream) {
    return aread_l2_dense_total(params, coef, n, grads, loss_out, accumulate, nullptr, nullptr, stream);
}
