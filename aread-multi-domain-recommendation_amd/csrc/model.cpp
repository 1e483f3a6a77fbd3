// model.cpp -- layouts of the dense model (flat parameter buffer, statistics buffer, workspace) and the
// table that maps the reference's state_dict keys onto them.
#include "model.h"
#include <string.h>

static int64_t pad4(int64_t v) { return (v + 3) & ~3ll; }

static void add_tensor(aread_model* m, const std::string& name, int kind, int64_t off, int ndim, int64_t s0, int64_t s1,
                       float l2) {
    aread_tensor_desc d;
    memset(&d, 0, sizeof(d));
    snprintf(d.name, sizeof(d.name), "%s", name.c_str());
    d.kind = kind; d.ndim = ndim; d.offset = off; d.shape[0] = s0; d.shape[1] = s1; d.l2 = l2;
    m->tensors.push_back(d);
}

// lay out one MLP stack; `prefix(g)` is the reference module path of group g
template <class PrefixFn>
static void layout_stack(aread_model* m, StackL* st, int G, int in_dim, const int32_t* dims, int n_layers, int stack_id,
                         bool shared_first_input, int first_in_ld, int64_t first_in_gs, PrefixFn prefix, int64_t* po,
                         int64_t* so, int* nbt) {
    st->n_layers = n_layers;
    int prev = in_dim;
    for (int j = 0; j < n_layers; ++j) {
        LayerL& L = st->L[j];
        L.G = G; L.in_dim = prev; L.out_dim = dims[j]; L.ncols = G * dims[j];
        L.stack = stack_id; L.layer = j;
        if (j == 0) { L.in_ld = first_in_ld; L.in_gs = shared_first_input ? 0 : first_in_gs; }
        else { L.in_ld = G * prev; L.in_gs = prev; }
        L.w = *po; *po += pad4((int64_t)L.ncols * prev);
        L.b = *po; *po += pad4(L.ncols);
        L.gamma = *po; *po += pad4(L.ncols);
        L.beta = *po; *po += pad4(L.ncols);
        L.rmean = *so; *so += pad4(L.ncols);
        L.rvar = *so; *so += pad4(L.ncols);
        L.nbt0 = *nbt; *nbt += G;
        for (int g = 0; g < G; ++g) {
            const std::string p = prefix(g) + ".layers.";
            const std::string lin = p + std::to_string(4 * j), bn = p + std::to_string(4 * j + 1);
            add_tensor(m, lin + ".weight", 0, L.w + (int64_t)g * dims[j] * prev, 2, dims[j], prev, m->cfg.l2_dnn);
            add_tensor(m, lin + ".bias", 0, L.b + (int64_t)g * dims[j], 1, dims[j], 0, 0.f);
            add_tensor(m, bn + ".weight", 0, L.gamma + (int64_t)g * dims[j], 1, dims[j], 0, m->cfg.l2_dnn);
            add_tensor(m, bn + ".bias", 0, L.beta + (int64_t)g * dims[j], 1, dims[j], 0, 0.f);
            add_tensor(m, bn + ".running_mean", 1, L.rmean + (int64_t)g * dims[j], 1, dims[j], 0, 0.f);
            add_tensor(m, bn + ".running_var", 1, L.rvar + (int64_t)g * dims[j], 1, dims[j], 0, 0.f);
            add_tensor(m, bn + ".num_batches_tracked", 2, L.nbt0 + g, 0, 0, 0, 0.f);
        }
        prev = dims[j];
    }
}

extern "C" int aread_model_create(const aread_model_cfg* c, aread_model** out) {
    AR_CHECK_ARG(c && out, "aread_model_create: null argument");
    AR_CHECK_ARG(c->embed_dim > 0 && c->embed_dim % 4 == 0, "embed_dim=%d must be a positive multiple of 4", c->embed_dim);
    AR_CHECK_ARG(c->f_out > 0 && c->domain_field >= 0 && c->domain_field < c->f_out, "bad f_out/domain_field");
    AR_CHECK_ARG((int64_t)c->f_out * c->embed_dim <= 1024, "D = f_out*embed_dim = %d exceeds 1024", c->f_out * c->embed_dim);
    AR_CHECK_ARG(c->n_expert >= 1 && c->n_expert <= 8, "n_expert=%d not in [1,8]", c->n_expert);
    AR_CHECK_ARG(c->n_expert_layers >= 1 && c->n_expert_layers <= AREAD_MAX_LAYER, "bad n_expert_layers");
    AR_CHECK_ARG(c->n_level >= 1 && c->n_level <= AREAD_MAX_LEVEL, "bad n_level");
    AR_CHECK_ARG(c->n_tower_layers >= 1 && c->n_tower_layers <= AREAD_MAX_LAYER, "bad n_tower_layers");
    AR_CHECK_ARG(c->n_cross >= 0 && c->n_cross <= MAX_CROSS, "n_cross=%d not in [0,%d]", c->n_cross, MAX_CROSS);
    AR_CHECK_ARG(c->n_domain >= 1 && c->n_domain <= MAX_SEG, "n_domain=%d not in [1,%d]", c->n_domain, MAX_SEG);
    AR_CHECK_ARG(c->dropout >= 0.f && c->dropout < 1.f, "dropout=%f not in [0,1)", c->dropout);
    AR_CHECK_ARG(c->precision == 0 || c->precision == 1, "precision=%d must be 0 (fp32) or 1 (split-bf16)", c->precision);
    for (int j = 0; j < c->n_expert_layers; ++j)
        AR_CHECK_ARG(c->expert_dims[j] > 0 && c->expert_dims[j] % 4 == 0 && c->expert_dims[j] <= 4096,
                     "expert_dims[%d]=%d must be a positive multiple of 4", j, c->expert_dims[j]);
    for (int l = 0; l < c->n_level; ++l) {
        AR_CHECK_ARG(c->n_tower[l] >= 1 && c->n_tower[l] <= MAX_TOWER, "n_tower[%d]=%d not in [1,%d]", l, c->n_tower[l], MAX_TOWER);
        for (int j = 0; j < c->n_tower_layers; ++j)
            AR_CHECK_ARG(c->tower_dims[l][j] > 0 && c->tower_dims[l][j] % 4 == 0, "tower_dims[%d][%d]=%d must be a positive multiple of 4",
                         l, j, c->tower_dims[l][j]);
    }
    for (int l = 0; l < c->n_level; ++l) {
        // row-local mix kernels stage 16 rows x (towers x width) floats per operand in LDS
        const int w_in = l == 0 ? c->expert_dims[c->n_expert_layers - 1] : c->tower_dims[l - 1][c->n_tower_layers - 1];
        AR_CHECK_ARG(16 * c->n_tower[l] * w_in <= 4096, "level %d: n_tower*input_width = %d exceeds 256", l, c->n_tower[l] * w_in);
        if (l > 0) AR_CHECK_ARG(c->n_tower[l] * c->n_tower[l - 1] <= 128, "level %d: too many gate edges", l);
    }
    AR_CHECK_ARG(16 * c->n_expert * c->expert_dims[c->n_expert_layers - 1] <= 8192, "n_expert*expert_out exceeds 512");
    aread_model* m = new aread_model();
    m->cfg = *c;
    m->E = c->embed_dim;
    m->D = c->f_out * c->embed_dim;
    m->n_heads = c->n_tower[c->n_level - 1];
    m->h_last = c->tower_dims[c->n_level - 1][c->n_tower_layers - 1];
    m->head_ld = m->D + m->h_last;
    const int D = m->D, E = m->E;
    int64_t po = 0, so = 0;
    int nbt = 0;
    // --- small dense tensors -----------------------------------------------------------------
    m->lin_w = po; po += pad4(D);
    add_tensor(m, "linear.fc.weight", 0, m->lin_w, 2, 1, D, c->l2_linear);
    m->lin_b = po; po += 4;
    add_tensor(m, "linear.fc.bias", 0, m->lin_b, 1, 1, 0, 0.f);
    m->cn_w = po; po += pad4((int64_t)c->n_cross * D);
    m->cn_b = po; po += pad4((int64_t)c->n_cross * D);
    for (int i = 0; i < c->n_cross; ++i) {
        add_tensor(m, "cn.w." + std::to_string(i) + ".weight", 0, m->cn_w + (int64_t)i * D, 2, 1, D, c->l2_cross);
        add_tensor(m, "cn.b." + std::to_string(i), 0, m->cn_b + (int64_t)i * D, 1, D, 0, 0.f);
    }
    const int n0 = c->n_tower[0];
    m->gate_w = po; po += pad4((int64_t)n0 * c->n_expert * D);
    m->gate_b = po; po += pad4(n0 * c->n_expert);
    for (int t = 0; t < n0; ++t) {
        add_tensor(m, "mmoe_gates." + std::to_string(t) + ".0.weight", 0, m->gate_w + (int64_t)t * c->n_expert * D, 2, c->n_expert, D, 0.f);
        add_tensor(m, "mmoe_gates." + std::to_string(t) + ".0.bias", 0, m->gate_b + (int64_t)t * c->n_expert, 1, c->n_expert, 0, 0.f);
    }
    m->group_emb = po; po += pad4((int64_t)n0 * E);
    add_tensor(m, "group_embedding.weight", 0, m->group_emb, 2, n0, E, 0.f);
    // tower gates: rows ordered (level, tower t, source s)
    m->gate_rows = 0;
    for (int l = 1; l < c->n_level; ++l) { m->gate_off[l] = m->gate_rows; m->gate_rows += c->n_tower[l] * c->n_tower[l - 1]; }
    m->gate_off[0] = 0;
    m->tgate_w = po; po += pad4((int64_t)(m->gate_rows > 0 ? m->gate_rows : 1) * 2 * E);
    m->tgate_b = po; po += pad4(m->gate_rows > 0 ? m->gate_rows : 1);
    for (int l = 1; l < c->n_level; ++l)
        for (int t = 0; t < c->n_tower[l]; ++t) {
            const int r0 = m->gate_off[l] + t * c->n_tower[l - 1];
            const std::string p = "tower_gates." + std::to_string(l - 1) + "." + std::to_string(t) + ".0.";
            add_tensor(m, p + "weight", 0, m->tgate_w + (int64_t)r0 * 2 * E, 2, c->n_tower[l - 1], 2 * E, 0.f);
            add_tensor(m, p + "bias", 0, m->tgate_b + r0, 1, c->n_tower[l - 1], 0, 0.f);
        }
    m->head_w = po; po += pad4((int64_t)m->n_heads * m->head_ld);
    for (int i = 0; i < m->n_heads; ++i)
        add_tensor(m, "towers_linear." + std::to_string(i) + ".weight", 0, m->head_w + (int64_t)i * m->head_ld, 2, 1, m->head_ld, 0.f);
    // --- MLP stacks ----------------------------------------------------------------------------
    layout_stack(m, &m->experts, c->n_expert, D, c->expert_dims, c->n_expert_layers, 0, true, D, 0,
                 [](int g) { return "mmoe_experts." + std::to_string(g); }, &po, &so, &nbt);
    int tin = c->expert_dims[c->n_expert_layers - 1];
    for (int l = 0; l < c->n_level; ++l) {
        layout_stack(m, &m->towers[l], c->n_tower[l], tin, c->tower_dims[l], c->n_tower_layers, 1 + l, false,
                     c->n_tower[l] * tin, tin, [l](int g) { return "towers." + std::to_string(l) + "." + std::to_string(g); },
                     &po, &so, &nbt);
        tin = c->tower_dims[l][c->n_tower_layers - 1];
    }
    m->n_params = po;
    m->n_stats = so;
    m->n_bn = nbt;
    // --- masks ---------------------------------------------------------------------------------
    int eo = 0;
    m->mask_off[0] = 0; eo += c->n_tower[0];
    for (int l = 1; l < c->n_level; ++l) { m->mask_off[l] = eo; eo += c->n_tower[l - 1] * c->n_tower[l]; }
    m->mask_off[c->n_level] = eo; eo += m->n_heads;
    m->edge_count = eo;
    m->ld_ge = (int)pad4(n0 * c->n_expert);
    m->ld_gt = (int)pad4(m->gate_rows > 0 ? m->gate_rows : 1);
    m->ld_h = (int)pad4(m->n_heads);
    *out = m;
    return AREAD_OK;
}

int model_streams_init(const aread_model* m) {
    if (m->side) return AREAD_OK;
    AR_HIP(hipStreamCreateWithFlags(&m->side, hipStreamNonBlocking));
    AR_HIP(hipStreamCreateWithFlags(&m->side2, hipStreamNonBlocking));
    for (int i = 0; i < 64; ++i) AR_HIP(hipEventCreateWithFlags(&m->ev[i], hipEventDisableTiming));
    AR_HIP(hipEventCreateWithFlags(&m->ev_prep, hipEventDisableTiming));
    AR_HIP(hipEventCreateWithFlags(&m->ev_join2, hipEventDisableTiming));
    m->n_ev = 64;
    return AREAD_OK;
}

extern "C" void aread_model_destroy(aread_model* m) {
    if (!m) return;
    if (m->side) {
        for (int i = 0; i < m->n_ev; ++i) (void)hipEventDestroy(m->ev[i]);
        if (m->ev_prep) (void)hipEventDestroy(m->ev_prep);
        if (m->ev_join2) (void)hipEventDestroy(m->ev_join2);
        (void)hipStreamDestroy(m->side);
        if (m->side2) (void)hipStreamDestroy(m->side2);
    }
    delete m;
}
extern "C" int64_t aread_model_param_floats(const aread_model* m) { return m ? m->n_params : -1; }
extern "C" int64_t aread_model_stat_floats(const aread_model* m) { return m ? m->n_stats : -1; }
extern "C" int aread_model_n_bn(const aread_model* m) { return m ? m->n_bn : -1; }
extern "C" int aread_model_n_tensors(const aread_model* m) { return m ? (int)m->tensors.size() : -1; }
extern "C" int aread_model_edge_count(const aread_model* m) { return m ? m->edge_count : -1; }
extern "C" int aread_model_gate_rows(const aread_model* m) { return m ? m->gate_rows : -1; }
extern "C" int aread_model_tensor(const aread_model* m, int i, aread_tensor_desc* out) {
    AR_CHECK_ARG(m && out && i >= 0 && i < (int)m->tensors.size(), "aread_model_tensor: bad index %d", i);
    *out = m->tensors[i];
    return AREAD_OK;
}
extern "C" int aread_model_l2_coef(const aread_model* m, float* coef) {
    AR_CHECK_ARG(m && coef, "aread_model_l2_coef: null argument");
    for (int64_t i = 0; i < m->n_params; ++i) coef[i] = 0.f;
    for (const auto& t : m->tensors) {
        if (t.kind != 0 || t.l2 == 0.f) continue;
        int64_t n = 1;
        for (int d = 0; d < t.ndim; ++d) n *= t.shape[d];
        for (int64_t i = 0; i < n; ++i) coef[t.offset + i] = t.l2;
    }
    return AREAD_OK;
}

// ------------------------------------------------------------------------------------------------
// workspace
// ------------------------------------------------------------------------------------------------

static int64_t take(int64_t* o, int64_t n) {
    int64_t r = *o;
    *o += (n + 63) & ~63ll;          // 256-byte granules
    return r;
}

// floats taken by the pre-tiled split-bf16 image of a weight operand (gemm_wide.h: 32*NF-row tiles, 32-wide k-steps, hi + lo)
static int64_t wimg_floats(int G, int N, int K) {
    const int nf = N <= 64 ? 2 : (((N + 95) / 96 * 96 < (N + 127) / 128 * 128) ? 3 : 4);
    const int64_t tn = 32 * nf;
    return (int64_t)G * ((N + tn - 1) / tn) * ((K + 31) / 32) * 2 * tn * 32 / 2;
}

static void layer_ws(const LayerL& L, int64_t rows, int64_t tiles, int n_seg, LayerWs* w, int64_t* o) {
    w->H = take(o, rows * L.ncols);
    w->Act = take(o, rows * L.ncols);
    w->dAct = take(o, rows * L.ncols);
    w->part = take(o, tiles * L.ncols * 2);
    w->mean = take(o, (int64_t)n_seg * L.ncols);
    w->rstd = take(o, (int64_t)n_seg * L.ncols);
    w->var = take(o, (int64_t)n_seg * L.ncols);
    w->bpart = take(o, tiles * L.ncols * 2);
    w->s12 = take(o, (int64_t)n_seg * L.ncols * 2);
    w->cpart = take(o, tiles * L.ncols);
    w->wT = take(o, (int64_t)L.ncols * L.in_dim);
    const bool one = (L.in_gs == 0 && L.G > 1) || L.G == 1;          // the groups share their input: one GEMM over all columns
    const int G = one ? 1 : L.G, N = one ? L.ncols : L.out_dim;
    w->wimg_f = take(o, wimg_floats(G, N, L.in_dim));
    w->wimg_d = take(o, wimg_floats(G, L.in_dim, N));
}

void ws_layout(const aread_model* m, int64_t B, int n_seg, WsLayout* w) {
    const aread_model_cfg& c = m->cfg;
    const int64_t rows = plan_max_rows(B, n_seg), tiles = rows / TILE_M;
    const int D = m->D, E = m->E;
    int64_t o = 0;
    w->max_rows = rows; w->n_tiles = tiles;
    w->cn = take(&o, rows * D);
    w->dcn = take(&o, rows * D);
    w->deg = take(&o, rows * D);
    w->lin = take(&o, rows);
    w->dlin = take(&o, rows);
    w->xw = take(&o, (int64_t)MAX_CROSS * rows);
    w->q = take(&o, rows * 2 * E);
    w->dq = take(&o, rows * 2 * E);
    w->glogE = take(&o, rows * m->ld_ge);
    w->dglogE = take(&o, rows * m->ld_ge);
    w->glogT = take(&o, rows * m->ld_gt);
    w->dglogT = take(&o, rows * m->ld_gt);
    w->hc = take(&o, rows * m->ld_h);
    w->z = take(&o, rows * m->ld_h);
    w->prob = take(&o, rows * m->ld_h);
    w->dz = take(&o, rows * m->ld_h);
    w->grp = take(&o, (int64_t)MAX_SEG * E);
    w->dgrp_part = take(&o, tiles * 4 * E);          // SUB slots per tile
    for (int l = 0; l < c.n_level; ++l) {
        const LayerL& L0 = m->towers[l].L[0];
        w->In[l] = take(&o, rows * L0.G * L0.in_dim);
        w->dIn[l] = take(&o, rows * L0.G * L0.in_dim);
    }
    for (int j = 0; j < m->experts.n_layers; ++j) layer_ws(m->experts.L[j], rows, tiles, n_seg, &w->ex[j], &o);
    for (int l = 0; l < c.n_level; ++l)
        for (int j = 0; j < m->towers[l].n_layers; ++j) layer_ws(m->towers[l].L[j], rows, tiles, n_seg, &w->tw[l][j], &o);
    w->active = take(&o, (int64_t)AREAD_MAX_LEVEL * MAX_SEG * MAX_TOWER / 4);
    w->kact = take(&o, 2 * MAX_SEG);      // kact[MAX_SEG] then n0act[MAX_SEG]
    w->seg_dom = take(&o, MAX_SEG);
    w->loss_part = take(&o, tiles * 4);
    w->gate_part = take(&o, tiles * 4 * m->ld_gt);
    // split-K slabs, one region per wgrad so that all of them can be reduced by one launch at the end
    auto slab = [&](int G, int M, int N) {       // dense.hip::wgrad swaps the roles of a skinny M (<= 16) and N
        const bool swap = M <= 16 && N > 16 && G == 1;
        return take(&o, (int64_t)G * M * N * wgrad_ksplit(rows, G, swap ? N : M, swap ? M : N).k_split);
    };
    for (int j = 0; j < m->experts.n_layers; ++j) {
        const LayerL& L = m->experts.L[j];
        w->slab_ex[j] = (L.in_gs == 0 && L.G > 1) ? slab(1, L.ncols, L.in_dim) : slab(L.G, L.out_dim, L.in_dim);
    }
    for (int l = 0; l < c.n_level; ++l)
        for (int j = 0; j < m->towers[l].n_layers; ++j) {
            const LayerL& L = m->towers[l].L[j];
            w->slab_tw[l][j] = slab(L.G, L.out_dim, L.in_dim);
        }
    w->slab_head = slab(1, m->n_heads, D);
    w->slab_gate = slab(1, c.n_tower[0] * c.n_expert, D);
    w->slab_tgate = slab(1, m->gate_rows > 0 ? m->gate_rows : 1, 2 * E);
    w->rw_part = take(&o, tiles * 4 * ((int64_t)(2 * MAX_CROSS + 1) * D + 4));
    w->misc_part = take(&o, tiles * 4 * 1024);
    w->cs_part_e = take(&o, tiles * m->ld_ge);
    w->cs_part_t = take(&o, tiles * m->ld_gt);
    w->tf_sync = take(&o, 2 * (AREAD_MAX_LEVEL * AREAD_MAX_LAYER * MAX_SEG + 64));
    w->ab_sync = take(&o, AREAD_MAX_LAYER * 16 * MAX_SEG);      // arrival counters of k_act_bn_bwd: [layer][<= 16 column chunks][segment]
    w->tf_tags = o;
    for (int l = 0; l < c.n_level; ++l)
        for (int j = 0; j < m->towers[l].n_layers; ++j) {
            const LayerL& L = m->towers[l].L[j];
            w->tw[l][j].tag_f = take(&o, tiles * L.ncols * 4);
            w->tw[l][j].tag_b = take(&o, tiles * L.ncols * 4);
            w->tw[l][j].fin_f = take(&o, (int64_t)MAX_SEG * L.ncols * 4);
            w->tw[l][j].fin_b = take(&o, (int64_t)MAX_SEG * L.ncols * 4);
        }
    w->tf_tags_floats = o - w->tf_tags;
    for (int j = 0; j < m->experts.n_layers; ++j)            // k_act_bn_bwd (A/B only): right behind, one memset can cover both
        w->ex[j].tag_b = take(&o, tiles * m->experts.L[j].ncols * 4);
    for (int j = 0; j < m->experts.n_layers; ++j)
        w->ex[j].fin_b = take(&o, (int64_t)MAX_SEG * m->experts.L[j].ncols * 4);
    w->ab_tags_floats = o - w->tf_tags - w->tf_tags_floats;
    w->total = o;
}

extern "C" int64_t aread_model_workspace_bytes(const aread_model* m, int64_t B, int n_seg) {
    if (!m || B <= 0 || n_seg < 1 || n_seg > MAX_SEG) return -1;
    WsLayout w;
    ws_layout(m, B, n_seg, &w);
    return w.total * 4;
}

// test/debug introspection: float offset of a named workspace buffer ("cn", "lin", "q", "glogE", "glogT", "hc",
// "z", "prob", "dz", "In<l>", "dIn<l>", "ex<j>.<field>", "tw<l>.<j>.<field>" with field in H, Act, dAct, mean, rstd, var)
static int64_t layer_field(const LayerWs& w, const char* f) {
    if (!strcmp(f, "H")) return w.H;
    if (!strcmp(f, "Act")) return w.Act;
    if (!strcmp(f, "dAct")) return w.dAct;
    if (!strcmp(f, "mean")) return w.mean;
    if (!strcmp(f, "rstd")) return w.rstd;
    if (!strcmp(f, "var")) return w.var;
    return -1;
}
extern "C" int64_t aread_debug_ws_offset(const aread_model* m, int64_t B, int n_seg, const char* name) {
    if (!m || !name) return -1;
    WsLayout w;
    ws_layout(m, B, n_seg, &w);
    int l, j;
    char f[16];
    if (!strcmp(name, "misc_part")) return w.misc_part;
    if (!strcmp(name, "gate_part")) return w.gate_part;
    if (!strcmp(name, "tf_err")) return w.tf_sync + AREAD_MAX_LEVEL * AREAD_MAX_LAYER * MAX_SEG;
    if (!strcmp(name, "cn")) return w.cn;
    if (!strcmp(name, "lin")) return w.lin;
    if (!strcmp(name, "xw")) return w.xw;
    if (!strcmp(name, "q")) return w.q;
    if (!strcmp(name, "dq")) return w.dq;
    if (!strcmp(name, "glogE")) return w.glogE;
    if (!strcmp(name, "glogT")) return w.glogT;
    if (!strcmp(name, "dglogE")) return w.dglogE;
    if (!strcmp(name, "dglogT")) return w.dglogT;
    if (!strcmp(name, "hc")) return w.hc;
    if (!strcmp(name, "z")) return w.z;
    if (!strcmp(name, "prob")) return w.prob;
    if (!strcmp(name, "dz")) return w.dz;
    if (!strcmp(name, "dcn")) return w.dcn;
    if (sscanf(name, "In%d", &l) == 1 && l >= 0 && l < m->cfg.n_level) return w.In[l];
    if (sscanf(name, "dIn%d", &l) == 1 && l >= 0 && l < m->cfg.n_level) return w.dIn[l];
    if (sscanf(name, "ex%d.%15s", &j, f) == 2 && j >= 0 && j < m->experts.n_layers) return layer_field(w.ex[j], f);
    if (sscanf(name, "tw%d.%d.%15s", &l, &j, f) == 3 && l >= 0 && l < m->cfg.n_level && j >= 0 && j < m->cfg.n_tower_layers)
        return layer_field(w.tw[l][j], f);
    return -1;
}

// ------------------------------------------------------------------------------------------------
// stand-alone MLP (MultiLayerPerceptron, layer.py:203-229)
// ------------------------------------------------------------------------------------------------
extern "C" int aread_mlp_create(const aread_mlp_cfg* c, aread_model** out) {
    AR_CHECK_ARG(c && out, "aread_mlp_create: null argument");
    AR_CHECK_ARG(c->in_dim > 0 && c->in_dim % 4 == 0, "in_dim=%d must be a positive multiple of 4", c->in_dim);
    AR_CHECK_ARG(c->n_layers >= 1 && c->n_layers <= AREAD_MAX_LAYER, "n_layers=%d not in [1,%d]", c->n_layers, AREAD_MAX_LAYER);
    AR_CHECK_ARG(c->dropout >= 0.f && c->dropout < 1.f, "dropout=%f not in [0,1)", c->dropout);
    AR_CHECK_ARG(c->precision == 0 || c->precision == 1, "precision=%d", c->precision);
    for (int j = 0; j < c->n_layers; ++j)
        AR_CHECK_ARG(c->dims[j] > 0 && c->dims[j] % 4 == 0, "dims[%d]=%d must be a positive multiple of 4", j, c->dims[j]);
    aread_model* m = new aread_model();
    memset(&m->cfg, 0, sizeof(m->cfg));
    m->cfg.n_expert = 1; m->cfg.n_expert_layers = c->n_layers; m->cfg.n_level = 0; m->cfg.n_domain = 1;
    m->cfg.dropout = c->dropout; m->cfg.precision = c->precision; m->cfg.l2_dnn = 0.f;
    for (int j = 0; j < c->n_layers; ++j) m->cfg.expert_dims[j] = c->dims[j];
    m->is_mlp = true; m->mlp_in = c->in_dim; m->mlp_out_layer = c->output_layer;
    m->D = c->in_dim; m->E = 0; m->n_heads = 0; m->h_last = c->dims[c->n_layers - 1]; m->head_ld = 0;
    int64_t po = 0, so = 0;
    int nbt = 0;
    layout_stack(m, &m->experts, 1, c->in_dim, c->dims, c->n_layers, 0, true, c->in_dim, 0,
                 [](int) { return std::string(""); }, &po, &so, &nbt);
    for (auto& t : m->tensors) {                       // ".layers.N.x" -> "layers.N.x"
        std::string n = t.name;
        if (!n.empty() && n[0] == '.') n = n.substr(1);
        snprintf(t.name, sizeof(t.name), "%s", n.c_str());
    }
    if (c->output_layer) {
        const int last = m->h_last;
        m->out_w = po; po += pad4(last);
        m->out_b = po; po += 4;
        const std::string p = "layers." + std::to_string(4 * c->n_layers);
        add_tensor(m, p + ".weight", 0, m->out_w, 2, 1, last, 0.f);
        add_tensor(m, p + ".bias", 0, m->out_b, 1, 1, 0, 0.f);
    }
    m->n_params = po; m->n_stats = so; m->n_bn = nbt;
    m->edge_count = 0; m->gate_rows = 0; m->ld_ge = m->ld_gt = m->ld_h = 4;
    *out = m;
    return AREAD_OK;
}

void mlp_ws_layout(const aread_model* m, int64_t B, WsLayout* w) {
    const int64_t rows = plan_max_rows(B, 1), tiles = rows / TILE_M;
    int64_t o = 0;
    memset(w, 0, sizeof(*w));
    w->ab_sync = -1; w->tf_sync = -1;                // no fused hand-off kernels in the stand-alone MLP
    w->max_rows = rows; w->n_tiles = tiles;
    w->In[0] = take(&o, rows * m->mlp_in);           // padded copy of x
    w->dIn[0] = take(&o, rows * m->mlp_in);          // padded dx
    w->hc = take(&o, rows * 4);                      // output layer: padded [rows, 1(+3)]
    w->dz = take(&o, rows * 4);
    for (int j = 0; j < m->experts.n_layers; ++j) {
        layer_ws(m->experts.L[j], rows, tiles, 1, &w->ex[j], &o);
        const LayerL& L = m->experts.L[j];
        w->slab_ex[j] = take(&o, (int64_t)L.out_dim * L.in_dim * wgrad_ksplit(rows, 1, L.out_dim, L.in_dim).k_split);
    }
    w->slab_head = take(&o, (int64_t)m->h_last * wgrad_ksplit(rows, 1, m->h_last > 16 ? m->h_last : 1, m->h_last > 16 ? 1 : m->h_last).k_split + 64);
    w->active = take(&o, (int64_t)AREAD_MAX_LEVEL * MAX_SEG * MAX_TOWER / 4);
    w->kact = take(&o, 2 * MAX_SEG);
    w->seg_dom = take(&o, MAX_SEG);
    w->misc_part = take(&o, tiles * 1024);
    w->total = o;
}

extern "C" int64_t aread_mlp_workspace_bytes(const aread_model* m, int64_t B) {
    if (!m || !m->is_mlp || B <= 0) return -1;
    WsLayout w;
    mlp_ws_layout(m, B, &w);
    return w.total * 4;
}
