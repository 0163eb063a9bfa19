// Host side of libflowmol_hip.so: the C ABI of include/flowmol_hip.h.
//  - fm_create    : looks the reference's state-dict tensors up by name, repacks them into MFMA
//                   B-fragment order (fm_device.h) with the algebraic hoists of SURVEY.md §7, uploads once
//  - fm_batch_bind: carves the caller's workspace, builds the destination-sorted edge layout on device
//  - fm_forward / fm_ctmc_step / fm_integrate: enqueue the kernel sequence on the caller's stream
// Compiled as HIP for gfx950 (flowmol_amd/build.py).  No host synchronisation on the hot path.
#include "fm_host.h"

namespace fmh { thread_local std::string g_create_error; }

namespace {

// ---------------------------------------------------------------------------------------- weight blob access
struct Blob {
    const float* base; const fm_tensor_desc* t; int n;
    std::string err;
    const float* get(const std::string& name, int64_t d0, int64_t d1 = -1) {
        for (int i = 0; i < n; ++i)
            if (name == t[i].name) {
                const bool ok = (d1 < 0) ? (t[i].ndim == 1 && t[i].shape[0] == d0)
                                         : (t[i].ndim == 2 && t[i].shape[0] == d0 && t[i].shape[1] == d1);
                if (!ok) {
                    char b[256];
                    snprintf(b, sizeof b, "tensor %s: shape (%lld,%lld) != expected (%lld,%lld)", name.c_str(),
                             (long long)t[i].shape[0], (long long)(t[i].ndim > 1 ? t[i].shape[1] : -1), (long long)d0, (long long)d1);
                    if (err.empty()) err = b;
                    return nullptr;
                }
                return base + t[i].offset;
            }
        if (err.empty()) err = "missing tensor " + name;
        return nullptr;
    }
};

// host-side staging arena; device pointers are offsets into the final device arena
struct Arena {
    std::vector<float> h;
    size_t add(const std::vector<float>& v) {
        size_t off = align_up(h.size(), 64);
        h.resize(off, 0.f);
        h.insert(h.end(), v.begin(), v.end());
        return off;
    }
    size_t add_raw(const float* p, size_t n) { return add(std::vector<float>(p, p + n)); }
};

// pack W_logical[k][n] (K x N, K%8==0, N%16==0) into fragment order
std::vector<float> pack(int K, int N, const std::function<float(int, int)>& w) {
    const int K8 = K / 8, NT = N / 16;
    std::vector<float> out((size_t)K8 * NT * 64 * 2);
    for (int ks = 0; ks < K8; ++ks)
        for (int nt = 0; nt < NT; ++nt)
            for (int lane = 0; lane < 64; ++lane) {
                const int j = lane & 15, h = lane >> 4;
                const size_t o = (((size_t)ks * NT + nt) * 64 + lane) * 2;
                out[o] = w(8 * ks + 2 * h, 16 * nt + j);
                out[o + 1] = w(8 * ks + 2 * h + 1, 16 * nt + j);
            }
    return out;
}

// quad-row packing for fm_wave_gemm4 (4-row tiles on v_mfma_f32_4x4x1_16B_f32): W_logical[k][n] (K x 256, K%4==0); entry (kq, g, lane) = the four
// weights W[4kq .. 4kq+3][64g + lane] -- one 1-KB buffer_load_dwordx4 per quad step and wave
std::vector<float> pack4(int K, const std::function<float(int, int)>& w, int G = 4) {      // G column groups of 64: N = 64 G
    const int KQ = K / 4;
    std::vector<float> out((size_t)KQ * G * 64 * 4);
    for (int kq = 0; kq < KQ; ++kq)
        for (int g = 0; g < G; ++g)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 4; ++i) out[(((size_t)kq * G + g) * 64 + lane) * 4 + i] = w(4 * kq + i, 64 * g + lane);
    return out;
}

// split-precision packing (fm_device.h "bf16x3"): W_logical[k][n] (K x N, K%32==0, N%16==0) as hi/lo bf16 planes in
// v_mfma_f32_16x16x32_bf16 B-fragment order: entry (kb, nt, plane, lane) = 8 bf16 = W[32kb + 8(lane>>4) + q][16nt + (lane&15)], q = 0..7
inline uint16_t bf16_rne(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);       // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline float bf16_f32(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
// 16-bit plane formats of the split modes (fm_device.h): 0 = bf16, 1 = IEEE half (round to nearest even, subnormals kept, clamped to +-65504)
inline uint16_t f16_rne(float f) { f = std::fmin(std::fmax(f, -65504.f), 65504.f); const _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }
inline float f16_f32(uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }
std::vector<float> pack_sp(int K, int N, const std::function<float(int, int)>& w, int npl = 2, int fmt = 0) {      // npl planes: hi, lo (| hi, mid, lo of the three-term mode)
    const int KB = K / 32, NT = N / 16;
    std::vector<uint16_t> out((size_t)KB * NT * npl * 64 * 8);
    for (int kb = 0; kb < KB; ++kb)
        for (int nt = 0; nt < NT; ++nt)
            for (int lane = 0; lane < 64; ++lane)
                for (int q = 0; q < 8; ++q) {
                    float r = w(32 * kb + 8 * (lane >> 4) + q, 16 * nt + (lane & 15)) * (fmt == 1 ? FM_F16_WSCALE : 1.0f);      // half planes: weights times 2^6 (fm_device.h)
                    const size_t e = (((size_t)kb * NT + nt) * npl) * 64 * 8;
                    for (int p_ = 0; p_ < npl; ++p_) {
                        const uint16_t h = fmt == 1 ? f16_rne(r) : bf16_rne(r);
                        out[e + (size_t)p_ * 64 * 8 + (size_t)lane * 8 + q] = h;
                        r -= fmt == 1 ? f16_f32(h) : bf16_f32(h);          // exact in f32
                    }
                }
    std::vector<float> f(out.size() / 2);
    memcpy(f.data(), out.data(), out.size() * 2);
    return f;
}

struct Fix { const void** slot; size_t off; };

struct Builder {
    Arena A; std::vector<Fix> fix;
    int sp_fmt = 0;          // plane format of the split-precision copies this builder packs: 0 = bf16, 1 = IEEE half (weights times 2^6); set once by fm_create
    template <class T> void put(const T*& slot, const std::vector<float>& v) { fix.push_back({(const void**)&slot, A.add(v)}); }
    void putv(const void*& slot, const std::vector<float>& v) { fix.push_back({&slot, A.add(v)}); }
};

// Linear weight W (out,in) row-major -> packed with K = Kp (logical input index remapped by kmap), N = Np
void pack_linear(Builder& B, const float2*& slot, const float* W, int out, int in, int Kp, int Np,
                 const std::function<int(int)>& kmap) {
    B.put(slot, pack(Kp, Np, [&](int k, int n) -> float {
        if (n >= out) return 0.f;
        const int kk = kmap(k);
        return (kk >= 0 && kk < in) ? W[(size_t)n * in + kk] : 0.f;
    }));
}
// the same logical matrix as pack_linear, K padded to a multiple of 32, as split-precision planes
void pack_linear_sp(Builder& B, const void*& slot, const float* W, int out, int in, int Kp, int Np, const std::function<int(int)>& kmap, int npl = 2) {
    const int K32 = (Kp + 31) / 32 * 32;
    B.putv(slot, pack_sp(K32, Np, [&](int k, int n) -> float {
        if (n >= out || k >= Kp) return 0.f;
        const int kk = kmap(k);
        return (kk >= 0 && kk < in) ? W[(size_t)n * in + kk] : 0.f;
    }, npl, B.sp_fmt));
}
void pack_linear4(Builder& B, const void*& slot, const float* W, int out, int in, int Kp, const std::function<int(int)>& kmap, int G = 4) {
    B.putv(slot, pack4(Kp, [&](int k, int n) -> float {
        if (n >= out) return 0.f;
        const int kk = kmap(k);
        return (kk >= 0 && kk < in) ? W[(size_t)n * in + kk] : 0.f;
    }, G));
}
void pad_vec(Builder& B, const float*& slot, const float* v, int n, int np) {
    std::vector<float> t(np, 0.f);
    for (int i = 0; i < n; ++i) t[i] = v[i];
    B.put(slot, t);
}

// one non-first GVP (vin = V, hidden = V, S real scalar channels in a 256-wide tile): reference gvp.py:30-88 parameter shapes
bool pack_gvp(Builder& B, Blob& bl, const std::string& key, int V, int S, int vout, FmGvpW& g, int sp = 0 /* bf16 planes of the split-precision copies: 0 | 2 | 3 */, bool rows4 = false) {
    const int vop = vout < 16 ? 16 : vout;
    const float* Wh = bl.get(key + ".Wh", V, V);
    const float* Wcp = bl.get(key + ".Wcp", V, 8);
    const float* Wu = bl.get(key + ".Wu", V + 4, vout);
    const float* Ws = bl.get(key + ".to_feats_out.0.weight", S, V + 4 + S);
    const float* bs = bl.get(key + ".to_feats_out.0.bias", S);
    const float* Wg = bl.get(key + ".scalar_to_vector_gates.weight", vout, S);
    const float* bg = bl.get(key + ".scalar_to_vector_gates.bias", vout);
    if (!Wh || !Wcp || !Wu || !Ws || !bs || !Wg || !bg) return false;
    B.put(g.Wv1, pack(V, V + 16, [&](int k, int n) -> float {
        if (n < V) return Wh[k * V + n];
        if (n < V + 8) return Wcp[k * 8 + (n - V)];
        return 0.f; }));
    B.put(g.Wu, pack(V + 8, vop, [&](int k, int n) -> float { return (k < V + 4 && n < vout) ? Wu[k * vout + n] : 0.f; }));
    // tile K order [s (256 columns, S real) | norms of the V hidden channels | cp norm 0, 0, cp norm 1, 0, cp norm 2, 0, cp norm 3, 0]: the four
    // cross-product norms sit on the even k-slots of the last k-superstep (fm_gvp_core: its all-zero second MFMA pass is skipped);
    // reference order [s (S) | sh (V+4)]
    auto kmap_s = [&](int k) { if (k < 256) return k < S ? k : -1;
                               const int o = k - 256; if (o < V) return S + o;
                               return ((o - V) & 1) ? -1 : S + V + (o - V) / 2; };
    pack_linear(B, g.Ws, Ws, S, V + 4 + S, 256 + V + 8, 256, kmap_s);
    if (rows4) pack_linear4(B, g.Ws4, Ws, S, V + 4 + S, 256 + V + 8, kmap_s);      // node-side GVPs: second copy for the 4-node tiles (fm_wave_gemm4)
    pad_vec(B, g.bs, bs, S, 256);
    pack_linear(B, g.Wg, Wg, vout, S, 256, vop, [&](int k) { return k < S ? k : -1; });
    pad_vec(B, g.bg, bg, vout, vop);
    if (sp) {
        pack_linear_sp(B, g.Ws_sp, Ws, S, V + 4 + S, 256 + V + 8, 256, kmap_s, sp);
        pack_linear_sp(B, g.Wg_sp, Wg, vout, S, 256, vop, [&](int k) { return k < S ? k : -1; }, sp);
    }
    return true;
}

void fill_mlp(FmMlpArgs& a, const MlpW& w, int rows) {
    a.rows = rows; a.K1p = w.K1p; a.H = w.H; a.O = w.O;
    a.W1 = w.W1; a.b1 = w.b1; a.W2 = w.W2; a.b2 = w.b2;
    a.ldx = ld_for(w.K1p > w.O ? w.K1p : w.O); a.ldh = ld_for(w.H);
    if (a.slabQ0 && a.ldh < 164) a.ldh = 164;      // SC_EDGE with the fused pair slab: the hidden tile later holds the K = 160 rows [rbf | ef]
}
template <int MODE>
void launch_mlp(Launch& L, const char* name, FmMlpArgs a, const MlpW& w, int rows, bool small_tiles = false) {
    fill_mlp(a, w, rows);
    if (small_tiles) L(name, fm_k_mlp2<MODE, 16>, dim3((rows + 15) / 16), dim3(FM_THREADS), lds_mlp(a.ldx, a.ldh, 16), a);
    else if (MODE == FM_MLP_SC_EDGE && a.slabQ0)
        // with the pair slab the kernel is matrix-pipe work followed by 3 KB of row stores per pair: 32-row tiles (38 KB of LDS) put four
        // independent workgroups on a CU instead of two, so that one's stores overlap the others' GEMMs
        L(name, fm_k_mlp2<FM_MLP_SC_EDGE, 32>, dim3((rows + 31) / 32), dim3(FM_THREADS), lds_mlp(a.ldx, a.ldh, 32), a);
    else if (MODE == FM_MLP_EDGE_HEAD && (rows + 31) / 32 >= 16 * L.c->n_cus)
        // large batches: the pair head is a gather of two 512-byte rows per pair in front of 17 k MAC -- latency / HBM work; 32-row tiles (34 KB of LDS)
        // put four workgroups on a CU instead of two
        L(name, fm_k_mlp2<FM_MLP_EDGE_HEAD, 32>, dim3((rows + 31) / 32), dim3(FM_THREADS), lds_mlp(a.ldx, a.ldh, 32), a);
    else L(name, fm_k_mlp2<MODE>, dim3((rows + FM_TM - 1) / FM_TM), dim3(FM_THREADS), lds_mlp(a.ldx, a.ldh), a);
}
template <int MODE_A, int MODE_B>
void launch_mlp_pair(Launch& L, const char* name, FmMlpArgs a, const MlpW& wa, int rows_a, FmMlpArgs b, const MlpW& wb, int rows_b, bool small_tiles = false) {
    fill_mlp(a, wa, rows_a); fill_mlp(b, wb, rows_b);
    const int tm = small_tiles ? 16 : FM_TM;
    const int ta = (rows_a + tm - 1) / tm, tb = (rows_b + tm - 1) / tm;
    const size_t lds = std::max(lds_mlp(a.ldx, a.ldh, tm), lds_mlp(b.ldx, b.ldh, tm));
    if (small_tiles) L(name, fm_k_mlp2_pair<MODE_A, MODE_B, 16>, dim3(ta + tb), dim3(FM_THREADS), lds, a, b, ta);
    else L(name, fm_k_mlp2_pair<MODE_A, MODE_B>, dim3(ta + tb), dim3(FM_THREADS), lds, a, b, ta);
}

// Pair-slab convolutions of a bound batch (fm_ctx::n_pq of them at most): the hoist is on for batches with at least four rounds of 32-row pair tiles
// (measured neutral below: the table costs a kernel phase, the saving is matrix-pipe time small batches are not bound by) or when fm_config.pair_slab
// forces it.  ONE predicate for the workspace layout (the Q tables, U KB each, exist only when it holds) and for every evaluation.
inline int pq_convs(const fm_ctx* c, long long U) {
    if (c->n_pq == 0 || U <= 0 || ld_for(c->sc_edge.H) > 164) return 0;      // the slab GEMM reads [rbf | ef] rows at the pitch 164 of a 128-wide hidden tile
    return (c->pq_forced || c->canonical || (U + 31) / 32 >= 16LL * c->n_cus) ? c->n_pq : 0;      // canonical: a rule that does not read the batch size
}

// ---------------------------------------------------------------------------------------- one network evaluation
// node_proj instances (small kernels of this unit): V = 16 | 32 vector channels, 64- or 16-row tiles
void launch_node_proj(Launch& L, const char* name, int V, bool small, int N, const FmProjArgs& pa) {
    const dim3 blk(FM_THREADS);
    if (V == 32) { if (small) L(name, fm_k_node_proj<32, 16>, dim3((N + 15) / 16), blk, lds_proj(32, 16), pa); else L(name, fm_k_node_proj<32>, dim3((N + FM_TM - 1) / FM_TM), blk, lds_proj(32), pa); }
    else { if (small) L(name, fm_k_node_proj<16, 16>, dim3((N + 15) / 16), blk, lds_proj(16, 16), pa); else L(name, fm_k_node_proj<16>, dim3((N + FM_TM - 1) / FM_TM), blk, lds_proj(16), pa); }
}

// One network evaluation of the bound batch.  The kernel instances are selected from the context's run-time parameters (vector channels V, edge / node tile heights
// TE / TN chosen at fm_batch_bind, destination-feature vectors HX); the heavy families live in translation units of their own (fm_host.h).
int evaluate(fm_ctx* c, hipStream_t st, const fm_state* state, const fm_dst* prev, int remove_com, const fm_dst* out,
             bool taps_on, const fm_dense_state* dense = nullptr, const float* temb = nullptr) {
    Launch L{c, st};
    const int V = c->V, TE = c->tm_edge, TN = c->tm_node, HX = c->HX;
    const FmBatch& b = c->b;
    const int N = b.N, E = b.E, U = b.U;
    const fm_config& cf = c->cfg;
    const int nc1 = c->nc + 1;
    auto tap = [&](const std::string& n, const void* p, size_t bytes) { if (taps_on) L.tap(n, p, bytes); };

    // Node- and pair-side MLPs with the same inputs share one launch while the batch is small (two launches less per step where launches
    // are what a step costs).  The shared launch allocates the LARGER tile's LDS (node tiles: 147 KB -> one workgroup per CU) for every
    // workgroup, so once the pair tiles alone fill the chip they run as a launch of their own at two workgroups per CU
    // (profiles/r02n: 1.16 -> 0.94 ms and 0.83 -> 0.68 ms per step at 1024 molecules).  fm_config.pair_mlps = 1 | -1 forces either (0 = this rule).
    const int mlp_tiles = (N + FM_TM - 1) / FM_TM + (U + FM_TM - 1) / FM_TM;
    const bool pair_mlps = c->fuse_node && (c->pair_mlps_forced >= 0 ? c->pair_mlps_forced != 0 : mlp_tiles <= c->n_cus);
    // 16-row MLP tiles while even those do not fill the chip (4 per CU fit in LDS): a tile's two dependent GEMMs are matrix-pipe time on one CU,
    // so a quarter of the rows is a quarter of the latency (fm_config.mlp_small_tiles = 1 | -1 forces either, 0 = this rule)
    const bool small_node = c->small_mlp_forced >= 0 ? c->small_mlp_forced != 0 : (N + 15) / 16 <= 4 * c->n_cus;      // decided per side: the node side
    const bool small_pair = c->small_mlp_forced >= 0 ? c->small_mlp_forced != 0 : (U + 15) / 16 <= 4 * c->n_cus;      // stays small ~25x longer than the pair side
    const bool small_mlp = small_node && small_pair;                                                                   // shared launches
    // node-side MLPs on 4-row tiles (fm_k_mlp4) while such tiles fit one per CU: a 16-row tile's two 256-wide layers are ~7 us of matrix time on one CU
    // whatever the batch, four rows on v_mfma_f32_4x4x1 are the layers' weight stream; fm_config.mlp_small_tiles = 2 forces it (1 / -1: never)
    const bool mlp4 = c->node_head_W1q && !dense && (c->mlp4_forced >= 0 ? c->mlp4_forced != 0 : (N + 3) / 4 <= c->n_cus);      // the regular tiles' bits (fm_rows4_linear): the choice may follow the batch size in canonical mode
    const int tiles4 = (N + 3) / 4;
    // pair-slab convolutions of this evaluation (first pass only; FmMlpArgs::slabQ0): self-conditioned evaluations with at least four rounds of
    // 32-row pair tiles (measured neutral on small batches: the table costs a kernel phase, the saving is matrix-pipe time they are not bound by)
    const int n_pq = (HX == 0 && prev && !dense) ? pq_convs(c, U) : 0;
    FmMlpArgs ma{};
    ma.na = c->na; ma.nc = c->nc; ma.ne = c->ne;
    ma.rbf_mu_step = c->rbf_mu_step; ma.rbf_inv_sigma = c->rbf_inv_sigma;
    const float* x_t = dense ? dense->x_t : state->x_t;
    if (dense) {
        // endpoint-parameterised model: the categorical inputs are continuous vectors, so the embeddings are real MLPs over the N
        // node rows [a_t | c_t | temb] and the U pair rows e_t (vector_field.py:226-261) -- the same two-layer kernel the token
        // tables use, fed with dense rows; pair rows are written to both directed edges.  Ps is free scratch until the first node_proj.
        const int kp = c->node_embed.K1p;
        L("dense_in", fm_k_dense_node_in, dim3(std::min(2048, (int)(((size_t)N * kp + 255) / 256))), dim3(256), 0, c->Ps, kp, N,
          (const float*)dense->a_t, c->na, (const float*)dense->c_t, c->nc, temb, cf.time_embedding_dim);
        FmMlpArgs a{};
        a.in = c->Ps; a.in_ld = kp; a.out = c->s; a.out_ld = 256; a.ln_g = c->node_ln_g; a.ln_b = c->node_ln_b; a.ln_n = c->S;
        launch_mlp<FM_MLP_TABLE>(L, "embed_nodes", a, c->node_embed, N);
        FmMlpArgs e{};
        e.in = dense->e_t; e.in_ld = c->ne; e.in_w = c->ne; e.out = c->ef; e.out_ld = 128; e.ln_g = c->edge_ln_g; e.ln_b = c->edge_ln_b; e.ln_n = c->F;
        e.p_e0 = b.p_e0; e.p_e1 = b.p_e1;
        launch_mlp<FM_MLP_TABLE>(L, "embed_pairs", e, c->edge_embed, U);
        tap("embed.s", c->s, (size_t)N * 256 * 4);
        tap("embed.ef", c->ef, (size_t)E * 128 * 4);
    } else if (prev) {
        FmMlpArgs a = ma;
        a.s_tab = c->s_tab; a.tok_a = state->a_t; a.tok_c = state->c_t; a.n_c1 = nc1;
        a.prev_a = prev->a; a.prev_c = prev->c; a.prev_x = prev->x; a.x_t = state->x_t;
        a.out = c->s;
        FmMlpArgs e = ma;
        e.e_src = b.e_src; e.e_dst = b.e_dst; e.e_pair = b.e_pair; e.tok_e = state->e_t;
        e.prev_e = prev->e; e.prev_x = prev->x; e.x_t = state->x_t; e.T1 = c->T1; e.ef_tab = c->ef_tab;
        e.out = c->ef;
        e.p_e0 = b.p_e0; e.p_e1 = b.p_e1;
        if (n_pq > 0) {      // the self-conditioning layer produces the edge features the first convolutions see: their pair slab in the same kernel
            e.slabW0 = c->conv[0].Ws_slab; e.slabQ0 = c->Q[0];
            if (n_pq > 1) { e.slabW1 = c->conv[1].Ws_slab; e.slabQ1 = c->Q[1]; }
        }
        FmMlp4Args a4{};
        a4.N = N; a4.W1q = c->sc_node_W1q; a4.b1 = c->sc_node.b1; a4.W2q = c->sc_node_W2q; a4.b2 = c->sc_node.b2;
        a4.s_tab = a.s_tab; a4.tok_a = a.tok_a; a4.tok_c = a.tok_c; a4.n_c1 = nc1; a4.prev_a = a.prev_a; a4.prev_c = a.prev_c; a4.prev_x = a.prev_x; a4.x_t = a.x_t;
        a4.na = c->na; a4.nc = c->nc; a4.rbf_mu_step = c->rbf_mu_step; a4.rbf_inv_sigma = c->rbf_inv_sigma; a4.out = c->s;
        // one row per unordered pair, written to both directed edges; node and pair tiles share one launch
        if (pair_mlps && mlp4 && small_mlp && c->sc_node_W1q) {
            fill_mlp(e, c->sc_edge, U);
            const int tb = (U + 15) / 16;
            L("sc", fm_k_mlp4_pair<FM_MLP4_SC_NODE, FM_MLP_SC_EDGE>, dim3(tiles4 + tb), dim3(FM_THREADS), std::max((size_t)FM_MLP4_LDS_BYTES, lds_mlp(e.ldx, e.ldh, 16)), a4, e, tiles4);
        } else if (pair_mlps) launch_mlp_pair<FM_MLP_SC_NODE, FM_MLP_SC_EDGE>(L, "sc", a, c->sc_node, N, e, c->sc_edge, U, small_mlp);
        else {
            if (mlp4 && c->sc_node_W1q) L("sc_node", fm_k_mlp4<FM_MLP4_SC_NODE>, dim3(tiles4), dim3(FM_THREADS), (size_t)FM_MLP4_LDS_BYTES, a4);
            else launch_mlp<FM_MLP_SC_NODE>(L, "sc_node", a, c->sc_node, N, small_node);
            launch_mlp<FM_MLP_SC_EDGE>(L, "sc_edge", e, c->sc_edge, U, small_pair);
        }
        tap("sc.s", c->s, (size_t)N * 256 * 4);
        tap("sc.ef", c->ef, (size_t)E * 128 * 4);
    } else {
        L("gather_s", fm_k_gather_rows, dim3(std::min(2048, (N * 64 + 255) / 256)), dim3(256), 0, c->s, (const float*)c->s_tab, 256, N,
          (const int*)state->a_t, (const int*)state->c_t, nc1, (const int*)nullptr);
        L("gather_ef", fm_k_gather_rows, dim3(std::min(4096, (int)(((size_t)E * 32 + 255) / 256))), dim3(256), 0, c->ef, c->ef_tab, 128, E,
          (const int*)state->e_t, (const int*)nullptr, 0, (const int*)b.e_pair);
        tap("embed.s", c->s, (size_t)N * 256 * 4);
        tap("embed.ef", c->ef, (size_t)E * 128 * 4);
    }

    const dim3 blk(FM_THREADS);
    const dim3 gn((N + FM_TM - 1) / FM_TM), ge((E + FM_TM - 1) / FM_TM);     // 64-row kernels
    const dim3 gnt((N + TN - 1) / TN), get(c->n_tiles_msg);                 // GVP kernels (edge tiles: molecule-aligned, FmBatch::tile_desc)
    // fm_config.fuse_node = -1 keeps round 1's launch sequence: node_proj / pos_update / node_proj_asd as kernels of their own (0 / 1 = fused)
    const bool fuse = c->fuse_node != 0 && HX == 0;      // destination-feature models keep the unfused node sequence (their projection GVP reuses the tile)
    const int n_pass = cf.n_convs * (cf.n_recycles > 1 ? cf.n_recycles : 1);       // vector_field.py:307: the whole stack again, same weights
    bool head_done = false;      // the edge head ran as the epilogue of the last EdgeUpdate
    for (int it = 0; it < n_pass; ++it) {
        const int i = it % cf.n_convs;
        const ConvW& cw = c->conv[i];
        if (it == 0 || !fuse) {      // later convs: projected in the previous conv's node_update
            FmProjArgs pa{};
            pa.N = N; pa.s = c->s; pa.v = c->v; pa.Wps = cw.Wps; pa.Ps = c->Ps; pa.Wpv = cw.Wpv; pa.PV = c->PV; pa.pv_w = c->PVW;
            if (it == 0) { pa.v_init = c->v; pa.x_src = x_t; pa.x_dst = c->xw; }     // v = 0, working copy of x (no memset / memcpy nodes)
            if (it == 0 && mlp4 && cw.Wps4) {
                FmMlp4Args p4{};
                p4.N = N; p4.in = c->s; p4.Wps4 = cw.Wps4; p4.Ps = c->Ps; p4.PV = c->PV; p4.pv_w = c->PVW; p4.v_init = c->v; p4.V = V; p4.x_src = x_t; p4.x_dst = c->xw;
                L("node_proj", fm_k_mlp4<FM_MLP4_PROJ0>, dim3(tiles4), blk, (size_t)FM_MLP4_LDS_BYTES, p4);
            }
            else launch_node_proj(L, "node_proj", V, small_node, N, pa);
        }
        FmMsgArgs m{};
        m.b = b; m.x = c->xw; m.ef = c->ef; m.Ps = c->Ps; m.PV = c->PV; m.w0 = cw.w0;
        if (HX > 0) {       // use_dst_feats: projection GVP of the conv's input features + its per-node hoists
            FmDstProjArgs dp{};
            dp.N = N; dp.s = c->s; dp.v = c->v; dp.g = cw.dproj; dp.Wsd = cw.Wsd; dp.Psd = c->Psd; dp.Wpvd = cw.Wpvd; dp.PVd = c->PVd; dp.pv_w = c->PVW;
            fm_launch_dst_proj(L, V, TN, HX, gnt, dp);
            m.Psd = c->Psd; m.PVd = c->PVd;
        }
        m.g0 = cw.msg[0]; m.g1 = cw.msg[1]; m.g2 = cw.msg[2];
        m.part_s = c->part_s; m.part_v = c->part_v;
        m.rbf_mu_step = c->rbf_mu_step; m.rbf_inv_sigma = c->rbf_inv_sigma;
        // per-edge message taps are written straight into the caller's buffers (both must be registered)
        const bool dbg = taps_on && it == 0 && c->taps.count("conv0.msg.s") && c->taps.count("conv0.msg.v");
        m.dbg_s = dbg ? (float*)c->taps["conv0.msg.s"] : nullptr; m.dbg_v = dbg ? (float*)c->taps["conv0.msg.v"] : nullptr;
        dim3 gmsg = get;
        if (c->xcd_swizzle) { m.xcd_chunk = ((int)get.x + 7) / 8; gmsg = dim3(8 * m.xcd_chunk); }
        const bool pq = HX == 0 && cf.precision == FM_PREC_F32 && it < n_pq;
        if (pq) { m.Q = c->Q[it]; m.g0.Ws = cw.Ws_sh; }          // GVP0's scalar GEMM: K = KU0 (hidden-vector norms); the rest arrives through Q
        fm_launch_edge_message(L, V, TE, HX, cf.precision, pq, gmsg, m);
        const int u = cf.update_after[i];
        FmNodeUpdArgs nu{};
        nu.b = b; nu.s = c->s; nu.v = c->v; nu.part_s = c->part_s; nu.part_v = c->part_v; nu.inv_z = cf.msg_z < 0.f ? -1.0f : 1.0f / cf.msg_z;
        nu.g0 = cw.upd[0]; nu.g1 = cw.upd[1]; nu.g2 = cw.upd[2];
        nu.ln1_g = cw.ln1_g; nu.ln1_b = cw.ln1_b; nu.ln2_g = cw.ln2_g; nu.ln2_b = cw.ln2_b;
        const std::string ci = "conv" + std::to_string(i);
        const bool tagg = taps_on && (c->taps.count(ci + ".agg.s") || c->taps.count(ci + ".agg.v"));
        // aggregated-message taps go through scratch of their own (Ps / PV, which served in round 1, are outputs of the fused kernel)
        nu.agg_s = tagg ? c->tap_s : nullptr;
        nu.agg_v = tagg ? c->tap_v : nullptr;
        if (fuse) {
            if (it + 1 < n_pass) { const ConvW& nx = c->conv[(i + 1) % cf.n_convs]; nu.Wps = nx.Wps; nu.Wps4 = nx.Wps4; nu.Ps = c->Ps; nu.Wpv = nx.Wpv; nu.PV = c->PV; }
            if (u >= 0) {
                const UpdW& uw = c->upd[u];
                nu.Wasd = uw.Wasd; nu.Wasd4 = uw.Wasd4; nu.Asd = c->Asd; nu.p0 = uw.pos[0]; nu.p1 = uw.pos[1]; nu.p2 = uw.pos[2]; nu.x = c->xw;
            }
        }
        nu.s_real = c->S;
        {   // instance of the node kernel: split precision (fused sequence only), 4 rg nodes per workgroup (small batches), narrow, or the regular one
            const bool spn = HX == 0 && TN <= 32 && prec_two_plane(cf.precision) && fuse;
            const int rg = (!spn && HX == 0 && (TN == 16 || TN == 32) && c->node_rg && fuse && c->S == 256 && cf.precision == FM_PREC_F32) ? c->node_rg : 0;
            if (spn) {
                if (it + 1 < n_pass) nu.Wps_sp = c->conv[(i + 1) % cf.n_convs].Wps_sp;
                if (u >= 0) nu.Wasd_sp = c->upd[u].Wasd_sp;
                fm_launch_node_update(L, V, TN, true, cf.precision == FM_PREC_F16X3 ? 3 : 1, 0, gnt, lds_gvp_sp(V, TN) - (size_t)TN * 9 * 4, nu);
            } else if (rg) {        // one tile per CU
                fm_launch_node_update(L, V, TN, false, 0, rg, dim3((N + 4 * rg - 1) / (4 * rg)), lds_gvp(V, TN, false), nu);
            } else {
                fm_launch_node_update(L, V, TN, c->S != 256, 0, 0, gnt, lds_gvp(V, TN, false), nu);
            }
        }
        if (tagg) { tap(ci + ".agg.s", c->tap_s, (size_t)N * 256 * 4); tap(ci + ".agg.v", c->tap_v, (size_t)N * 3 * V * 4); }
        tap(ci + ".s", c->s, (size_t)N * 256 * 4);
        tap(ci + ".v", c->v, (size_t)N * 3 * V * 4);
        if (u >= 0) {
            const UpdW& uw = c->upd[u];
            if (!fuse) {
                FmPosArgs pp{};
                pp.N = N; pp.s = c->s; pp.v = c->v; pp.x = c->xw; pp.g0 = uw.pos[0]; pp.g1 = uw.pos[1]; pp.g2 = uw.pos[2];
                fm_launch_pos_update(L, V, TN, gnt, pp);
                FmProjArgs pa2{};
                pa2.N = N; pa2.s = c->s; pa2.v = c->v; pa2.Wasd = uw.Wasd; pa2.Asd = c->Asd;
                launch_node_proj(L, "node_proj_asd", V, small_node, N, pa2);
            }
            FmEdgeUpdArgs eu{};
            eu.b = b; eu.x = c->xw; eu.Asd = c->Asd; eu.ef = c->ef; eu.W1 = uw.W1; eu.b1 = uw.b1; eu.W2 = uw.W2; eu.b2 = uw.b2;
            eu.ln_g = uw.ln_g; eu.ln_b = uw.ln_b; eu.rbf_mu_step = c->rbf_mu_step; eu.rbf_inv_sigma = c->rbf_inv_sigma;
            eu.f_real = c->F;
            if (prec_two_plane(cf.precision)) {
                FmEdgeUpdSpW sw{uw.W1_sp, uw.W2_sp};
                if (cf.precision == FM_PREC_F16X3) L("edge_update", fm_k_edge_update_sp<32, 1>, dim3((E + 31) / 32), blk, lds_edge_upd_sp(32), eu, sw);
                else L("edge_update", fm_k_edge_update_sp<32>, dim3((E + 31) / 32), blk, lds_edge_upd_sp(32), eu, sw);
            } else if (it == n_pass - 1 && c->fuse_head && c->F == 128 && c->tm_eupd == 32 && (long long)c->nmax * (c->nmax - 1) * 512 < 0x7ffffe00LL      // the tile's pair rows are
                       && !(taps_on && c->taps.count("upd" + std::to_string(i) + ".ef"))) {      // gathered with 31-bit offsets inside ONE molecule's edge rows (n < 2048 atoms); larger: separate head
                // the evaluation's last EdgeUpdate: its rows feed the edge head and nothing else -- tiles of 16 pairs, the head as the epilogue, no ef store
                eu.hW1 = c->edge_head.W1; eu.hb1 = c->edge_head.b1; eu.hW2 = c->edge_head.W2; eu.hb2 = c->edge_head.b2; eu.out_e = out->e; eu.ne = c->ne;
                L("edge_update_head", fm_k_edge_update<32, false, true>, dim3((U + 15) / 16), blk, lds_edge_upd(32), eu);
                head_done = true;
            } else if (c->F != 128) L("edge_update", fm_k_edge_update<32, true>, dim3((E + 31) / 32), blk, lds_edge_upd(32), eu);
            else if (c->tm_eupd == 32) L("edge_update", fm_k_edge_update<32, false>, dim3((E + 31) / 32), blk, lds_edge_upd(32), eu);
            else L("edge_update", fm_k_edge_update<64, false>, dim3((E + 63) / 64), blk, lds_edge_upd(64), eu);
            const std::string ui = "upd" + std::to_string(i);
            tap(ui + ".x", c->xw, (size_t)N * 3 * 4);
            if (!head_done) tap(ui + ".ef", c->ef, (size_t)E * 128 * 4);
        }
    }
    {
        FmMlpArgs a = ma;
        a.in = c->s; a.out = out->a; a.out2 = out->c;
        FmMlpArgs e = ma;
        e.ef = c->ef; e.p_e0 = b.p_e0; e.p_e1 = b.p_e1; e.out = out->e;
        FmMlp4Args a4{};
        a4.N = N; a4.W1q = c->node_head_W1q; a4.b1 = c->node_head.b1; a4.W2q = c->node_head_W2q; a4.b2 = c->node_head.b2;
        a4.na = c->na; a4.nc = c->nc; a4.in = c->s; a4.out = out->a; a4.out2 = out->c;
        if (head_done) {          // only the node head is left
            if (mlp4) L("node_head", fm_k_mlp4<FM_MLP4_NODE_HEAD>, dim3(tiles4), dim3(FM_THREADS), (size_t)FM_MLP4_LDS_BYTES, a4);
            else launch_mlp<FM_MLP_NODE_HEAD>(L, "node_head", a, c->node_head, N, small_node);
        } else if (pair_mlps && mlp4 && small_mlp) {
            fill_mlp(e, c->edge_head, U);
            const int tb = (U + 15) / 16;
            L("heads", fm_k_mlp4_pair<FM_MLP4_NODE_HEAD, FM_MLP_EDGE_HEAD>, dim3(tiles4 + tb), dim3(FM_THREADS), std::max((size_t)FM_MLP4_LDS_BYTES, lds_mlp(e.ldx, e.ldh, 16)), a4, e, tiles4);
        } else if (pair_mlps) launch_mlp_pair<FM_MLP_NODE_HEAD, FM_MLP_EDGE_HEAD>(L, "heads", a, c->node_head, N, e, c->edge_head, U, small_mlp);
        else {
            if (mlp4) L("node_head", fm_k_mlp4<FM_MLP4_NODE_HEAD>, dim3(tiles4), dim3(FM_THREADS), (size_t)FM_MLP4_LDS_BYTES, a4);
            else launch_mlp<FM_MLP_NODE_HEAD>(L, "node_head", a, c->node_head, N, small_node);
            launch_mlp<FM_MLP_EDGE_HEAD>(L, "edge_head", e, c->edge_head, U, small_pair);
        }
    }
    if (remove_com != 2) {       // 2: the caller's fused CTMC kernel centres the raw positions (c->xw) and writes out->x itself
        L.copy(out->x, c->xw, (size_t)N * 3 * 4);
        if (remove_com) L("remove_com", fm_k_remove_com, dim3(b.B), dim3(64), 0, out->x, (const int*)b.mol_node_off);
    }
    return L.rc;
}

int evaluate_dispatch(fm_ctx* c, hipStream_t st, const fm_state* state, const fm_dst* prev, int remove_com, const fm_dst* out, bool taps_on) {
    return evaluate(c, st, state, prev, remove_com, out, taps_on);
}

// The (atom type, charge) embedding table(s) of `n_tables` consecutive time points (temb: n_tables x time_embedding_dim) into the
// workspace's table slots 0..n_tables-1, ONE launch: the table depends on the time only, so fm_integrate builds the tables of a
// whole chunk of steps up front (a launch that fills the chip) instead of one 2-tile launch on every step's critical path.
int embed_table(fm_ctx* c, hipStream_t st, const float* temb, int n_tables = 1) {
    Launch L{c, st};
    const fm_config& cf = c->cfg;
    FmMlpArgs a{};
    a.in = nullptr; a.n_c1 = c->nc + 1;          // rows = (a,c) token pairs; the input row is built in the kernel's prologue
    a.emb_a = c->emb_a; a.emb_c = c->emb_c; a.temb = temb;
    a.ta = cf.a_token_dim ? cf.a_token_dim : c->na + 1; a.tc = cf.c_token_dim ? cf.c_token_dim : c->nc + 1; a.tt = cf.time_embedding_dim;
    a.out = c->s_tab_base; a.out_ld = 256; a.ln_g = c->node_ln_g; a.ln_b = c->node_ln_b; a.ln_n = c->S;
    fill_mlp(a, c->node_embed, c->tab_rows);
    a.tab_tiles = (c->tab_rows + FM_TM - 1) / FM_TM; a.tab_stride = a.tab_tiles * FM_TM * 256;
    L("embed_table", fm_k_mlp2<FM_MLP_TABLE>, dim3(a.tab_tiles * n_tables), dim3(FM_THREADS), lds_mlp(a.ldx, a.ldh), a);
    return L.rc;
}

// tables_ready: the caller (fm_integrate) has already built this step's table into slot `table_slot`
int forward_impl(fm_ctx* c, hipStream_t st, const fm_state* state, const float* temb, const fm_dst* prev, int bootstrap,
                 int remove_com, const fm_dst* out, int table_slot = -1) {
    int rc = 0;
    if (table_slot < 0) { table_slot = 0; rc = embed_table(c, st, temb); }
    if (rc) return rc;
    c->s_tab = c->s_tab_base + (size_t)table_slot * c->tab_slot_floats;
    const bool sc = c->cfg.self_conditioning != 0;
    if (sc && !prev && bootstrap) {
        rc = evaluate_dispatch(c, st, state, nullptr, 0, &c->boot, false);
        if (rc) return rc;
        if (c->taps.count("boot.x")) { Launch L{c, st}; L.tap("boot.x", c->boot.x, (size_t)c->b.N * 12); L.tap("boot.a", c->boot.a, (size_t)c->b.N * c->na * 4);
            L.tap("boot.c", c->boot.c, (size_t)c->b.N * c->nc * 4); L.tap("boot.e", c->boot.e, (size_t)c->b.U * c->ne * 4); if (L.rc) return L.rc; }
        prev = &c->boot;
    }
    if (!sc) prev = nullptr;
    return evaluate_dispatch(c, st, state, prev, remove_com, out, true);
}

__global__ void fm_k_noop() {}

// frame: this step's slice of the trajectory sink (x, a, c, e, x1 used; a1 / c1 / e1 travel in `smp`); campbell steps write it inside the fused kernel
int ctmc_impl(fm_ctx* c, hipStream_t st, const fm_state* state, const fm_dst* dst, const fm_step_noise* nz,
              const fm_step_scalars* sc, const fm_sampled* smp, const float* x_raw = nullptr, const fm_traj_sink* frame = nullptr) {
    Launch L{c, st};
    // profiling only: an event pair around an empty kernel, once per step.  Its elapsed time is what a pair adds to every profiled launch
    // (event signalling + the dispatch that cannot overlap the previous kernel's tail); fm_profile_get("event_overhead") lets the caller
    // subtract it, which matters for the sub-millisecond kernels of small batches (HIP events vs rocprofv3: +12 % on a 350 us kernel).
    if (c->prof) L("event_overhead", fm_k_noop, dim3(1), dim3(64), 0);
    const FmBatch& b = c->b;
    struct Mod { int rows, K; const float* p; const int* mol; int* xt; int* x1; const float *q, *u1, *u2; };
    static const fm_step_noise no_noise{};
    if (!nz) {                    // Philox steps draw inside the kernel
        if (sc->noise_mode != FM_NOISE_PHILOX) return fail(c, FM_ERR_INVALID, "fm_ctmc_step: noise tensors missing (noise_mode FM_NOISE_TENSORS)");
        nz = &no_noise;
    }
    Mod mods[3] = {
        {b.N, c->na, dst->a, b.node_mol, state->a_t, (smp && smp->a1) ? smp->a1 : c->sa1, nz->q_a, nz->u1_a, nz->u2_a},
        {b.N, c->nc, dst->c, b.node_mol, state->c_t, (smp && smp->c1) ? smp->c1 : c->sc1, nz->q_c, nz->u1_c, nz->u2_c},
        {b.U, c->ne, dst->e, b.pair_mol, state->e_t, (smp && smp->e1) ? smp->e1 : c->se1, nz->q_e, nz->u1_e, nz->u2_e},
    };
    if (sc->dfm_type == FM_DFM_GAT && (sc->noise_mode == FM_NOISE_PHILOX || !nz->q_a))
        return fail(c, FM_ERR_INVALID, "fm_ctmc_step: dfm_type 'gat' needs the caller's noise tensors");
    if (sc->dfm_type == FM_DFM_GAT) {
        L("x_step", fm_k_x_step, dim3((b.N * 3 + 255) / 256), dim3(256), 0, state->x_t, (const float*)dst->x, sc->x_coef, sc->dt, sc->x_scale, b.N * 3);
        for (int m = 0; m < 3; ++m) {
            if (mods[m].rows == 0) continue;
            FmGatArgs a{};
            a.rows = mods[m].rows; a.K = mods[m].K; a.p = mods[m].p; a.xt = mods[m].xt; a.x1 = mods[m].x1; a.q = mods[m].q;
            a.temp = sc->cat_temperature; a.cf = sc->gat_cf[m]; a.cb = sc->gat_cb[m]; a.fw = sc->gat_fw; a.bw = sc->gat_bw; a.dt = sc->dt;
            L("ctmc_gat", fm_k_ctmc_gat, dim3((a.rows + 255) / 256), dim3(256), 0, a);
        }
        return L.rc;
    }
    if (sc->dfm_type != FM_DFM_CAMPBELL) return fail(c, FM_ERR_INVALID, "fm_ctmc_step: unknown dfm_type %d", sc->dfm_type);
    FmCtmcFusedArgs f{};
    const int* offs[3] = {b.mol_node_off, b.mol_node_off, b.mol_pair_off};
    for (int m = 0; m < 3; ++m) {
        FmCtmcMod& d = f.mod[m];
        d.K = mods[m].K; d.p = mods[m].p; d.xt = mods[m].xt; d.x1 = mods[m].x1; d.q = mods[m].q; d.u1 = mods[m].u1; d.u2 = mods[m].u2;
        d.off = offs[m]; d.unmask_prob = sc->unmask_prob[m]; d.mask_prob = sc->mask_prob[m];
        d.sink_t = frame ? (m == 0 ? frame->a : m == 1 ? frame->c : frame->e) : nullptr;
    }
    f.temp = sc->cat_temperature; f.hc_thresh = sc->hc_thresh; f.last_step = sc->last_step;
    f.x_t = state->x_t; f.x1 = dst->x; f.node_off = b.mol_node_off; f.coef = sc->x_coef; f.dt = sc->dt; f.scale = sc->x_scale;
    f.x_raw = x_raw; f.x1_out = dst->x;
    if (frame) { f.sink_x = frame->x; f.sink_x1 = frame->x1; }
    if (sc->noise_mode == FM_NOISE_PHILOX) { f.philox = 1; f.seed_lo = sc->philox_seed_lo; f.seed_hi = sc->philox_seed_hi; f.step = sc->step_index; f.mol_gid = c->mol_gid; }
    else if (!nz->q_a || !nz->u1_a || !nz->q_e) return fail(c, FM_ERR_INVALID, "fm_ctmc_step: noise tensors missing (noise_mode FM_NOISE_TENSORS)");
    // a few molecules: 1024-thread workgroups (one per molecule and modality is all the parallelism there is) -- and batches whose LARGEST molecule has more
    // than 4096 pairs (n >= 92): its pair rows are one workgroup's serial loop, 35 rounds of 256 threads at 134 atoms (GEOM size distribution: 135 us of a
    // 67.8-ms step against 39 us at 1024 x 47 atoms); else 256.  Same arithmetic either way (integer counts, per-row decisions).
    if ((b.B * 4 <= c->n_cus && c->nmax > 23) || c->nmax >= 92) L("ctmc", fm_k_ctmc_fused<1024>, dim3(b.B, 4), dim3(1024), 0, f);
    else L("ctmc", fm_k_ctmc_fused<256>, dim3(b.B, 4), dim3(256), 0, f);
    return L.rc;
}

}  // namespace

// ================================================================================================= C ABI
extern "C" {

const char* fm_last_error(const fm_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }
int fm_abi_version(void) { return FM_ABI_VERSION; }

int fm_create(const fm_config* cfg, const fm_tensor_desc* tensors, int n_tensors, const float* host_blob, fm_ctx** out) {
    if (!cfg || !tensors || !host_blob || !out) return fail(nullptr, FM_ERR_INVALID, "fm_create: null argument");
    if (cfg->abi_version != FM_ABI_VERSION) return fail(nullptr, FM_ERR_INVALID, "fm_create: ABI version %d != %d", cfg->abi_version, FM_ABI_VERSION);
    // The kernels' tiles are 256 scalar / 128 edge-feature columns wide.  Narrower models (configs/dev.yml: 64 / 64) run on the
    // same tiles: weights, biases and LayerNorm affine parameters are zero-padded when they are repacked, so the extra columns
    // stay exactly 0 through every Linear / SiLU / residual, and LayerNorm takes its statistics over the REAL width only.
    auto pow2 = [](int v) { return v >= 8 && (v & (v - 1)) == 0; };      // LayerNorm mean = sum * (1/n): exact division only for a power-of-two width
    if (!pow2(cfg->n_hidden_scalars) || cfg->n_hidden_scalars > 256 || !pow2(cfg->n_hidden_edge_feats) || cfg->n_hidden_edge_feats > 128 || cfg->rbf_dim != 32)
        return fail(nullptr, FM_ERR_INVALID, "fm_create: need power-of-two widths 8 <= n_hidden_scalars <= 256, 8 <= n_hidden_edge_feats <= 128, and rbf_dim == 32");
    if (cfg->n_vec_channels != 16 && cfg->n_vec_channels != 32) return fail(nullptr, FM_ERR_INVALID, "fm_create: n_vec_channels must be 16 or 32");
    if (cfg->n_convs < 1 || cfg->n_convs > FM_MAX_CONVS) return fail(nullptr, FM_ERR_INVALID, "fm_create: bad n_convs");
    if (cfg->n_recycles < 0 || cfg->n_recycles > 64) return fail(nullptr, FM_ERR_INVALID, "fm_create: n_recycles must be 0..64");
    if (cfg->msg_z == 0.f) return fail(nullptr, FM_ERR_INVALID, "fm_create: msg_z must be > 0 (divisor) or < 0 (mean over the in-edges)");
    if (cfg->n_atom_types + 1 > 16 || cfg->n_charges + 1 > 16 || cfg->n_bond_types + 1 > 16 || cfg->n_atom_types + cfg->n_charges > 32)
        return fail(nullptr, FM_ERR_INVALID, "fm_create: categorical widths exceed kernel limits");
    const bool tok = cfg->a_token_dim > 0;
    const int mk = cfg->has_mask ? 1 : 0;       // CTMC models: one mask category per categorical input
    if (!mk && (tok || cfg->self_conditioning)) return fail(nullptr, FM_ERR_INVALID, "fm_create: endpoint-parameterised models (has_mask = 0) take raw categorical vectors (token dims 0) and no self-conditioning");
    if ((cfg->c_token_dim > 0) != tok || (cfg->e_token_dim > 0) != tok) return fail(nullptr, FM_ERR_INVALID, "fm_create: token dims must be all zero or all non-zero");

    fm_ctx* c = new fm_ctx();
    c->cfg = *cfg;
    const int V = c->V = cfg->n_vec_channels;
    const int S = c->S = cfg->n_hidden_scalars, F = c->F = cfg->n_hidden_edge_feats;
    const int HX = c->HX = cfg->v_dst_feats, SD = c->SD = cfg->s_dst_feats;
    if ((HX > 0) != (SD > 0) || HX < 0 || HX > 8 || SD > 256 || (HX > 0 && HX != V / 4))
        { delete c; return fail(nullptr, FM_ERR_INVALID, "fm_create: destination-feature widths must be both 0 or v = n_vec_channels/4 (<= 8), s <= 256"); }
    const int H0 = V + 1 + HX, KU0 = pad8(H0 + 4), PVW = c->PVW = pvw_of(V, HX);
    if (cfg->precision != FM_PREC_F32 && cfg->precision != FM_PREC_BF16X3 && cfg->precision != FM_PREC_BF16X6 && cfg->precision != FM_PREC_F16X3) { delete c; return fail(nullptr, FM_ERR_INVALID, "fm_create: unknown precision %d", cfg->precision); }
    if (cfg->precision != FM_PREC_F32 && HX > 0) { delete c; return fail(nullptr, FM_ERR_INVALID, "fm_create: split precision is built for models without destination features"); }
    const int na = c->na = cfg->n_atom_types, nc = c->nc = cfg->n_charges, ne = c->ne = cfg->n_bond_types;
    c->rbf_mu_step = cfg->rbf_dmax / (float)(cfg->rbf_dim - 1);
    c->rbf_inv_sigma = 1.0f / (cfg->rbf_dmax / (float)cfg->rbf_dim);
    Blob bl{host_blob, tensors, n_tensors, {}};
    Builder B;
    B.sp_fmt = cfg->precision == FM_PREC_F16X3 ? 1 : 0;
    auto bail = [&](const std::string& m) { std::string mm = m; delete c; return fail(nullptr, FM_ERR_WEIGHTS, "fm_create: %s", mm.c_str()); };
    auto ident = [](int k) { return k; };

    // ---- input embeddings
    const int ta = tok ? cfg->a_token_dim : na + mk, tc = tok ? cfg->c_token_dim : nc + mk, te = tok ? cfg->e_token_dim : ne + mk;
    const int tt = cfg->time_embedding_dim;
    const float* emb_e = nullptr;
    if (tok) {
        const float* ea = bl.get("token_embeddings.a.weight", na + 1, ta);
        const float* ec = bl.get("token_embeddings.c.weight", nc + 1, tc);
        emb_e = bl.get("token_embeddings.e.weight", ne + 1, te);
        if (!ea || !ec || !emb_e) return bail(bl.err);
        B.put(c->emb_a, std::vector<float>(ea, ea + (na + 1) * ta));
        B.put(c->emb_c, std::vector<float>(ec, ec + (nc + 1) * tc));
    }
    {
        const int kin = ta + tc + tt;
        const float* W1 = bl.get("scalar_embedding.0.weight", S, kin); const float* b1 = bl.get("scalar_embedding.0.bias", S);
        const float* W2 = bl.get("scalar_embedding.2.weight", S, S); const float* b2 = bl.get("scalar_embedding.2.bias", S);
        const float* g = bl.get("scalar_embedding.4.weight", S); const float* be = bl.get("scalar_embedding.4.bias", S);
        if (!W1 || !b1 || !W2 || !b2 || !g || !be) return bail(bl.err);
        c->node_embed.K1p = pad8(kin); c->node_embed.H = 256; c->node_embed.O = 256;
        pack_linear(B, c->node_embed.W1, W1, S, kin, pad8(kin), 256, ident);
        pad_vec(B, c->node_embed.b1, b1, S, 256);
        pack_linear(B, c->node_embed.W2, W2, S, S, 256, 256, ident);
        pad_vec(B, c->node_embed.b2, b2, S, 256);
        pad_vec(B, c->node_ln_g, g, S, 256); pad_vec(B, c->node_ln_b, be, S, 256);
        c->tab_rows = (na + 1) * (nc + 1); c->tab_kp = pad8(kin);
    }
    const float *ee_W1, *ee_b1, *ee_W2, *ee_b2, *ee_g, *ee_b;
    {
        ee_W1 = bl.get("edge_embedding.0.weight", F, te); ee_b1 = bl.get("edge_embedding.0.bias", F);
        ee_W2 = bl.get("edge_embedding.2.weight", F, F); ee_b2 = bl.get("edge_embedding.2.bias", F);
        ee_g = bl.get("edge_embedding.4.weight", F); ee_b = bl.get("edge_embedding.4.bias", F);
        if (!ee_W1 || !ee_b1 || !ee_W2 || !ee_b2 || !ee_g || !ee_b) return bail(bl.err);
    }
    // edge-embedding table (ne+1 rows): the edge embedding has only ne+1 distinct inputs;
    // evaluated once here on the host in f32 (same op order as a row of the device MLP is not required: 1e-7 class)
    std::vector<float> ef_tab((size_t)(ne + 1) * 128), T1((size_t)(ne + 1) * 128, 0.f);
    for (int t = 0; t <= ne; ++t) {
        std::vector<float> in(te, 0.f), h1(F), h2(F);
        if (tok) for (int k = 0; k < te; ++k) in[k] = emb_e[t * te + k]; else if (t < te) in[t] = 1.f;
        for (int n = 0; n < F; ++n) { float acc = ee_b1[n]; for (int k = 0; k < te; ++k) acc = fmaf(ee_W1[n * te + k], in[k], acc); h1[n] = acc / (1.0f + expf(-acc)); }
        for (int n = 0; n < F; ++n) { float acc = ee_b2[n]; for (int k = 0; k < F; ++k) acc = fmaf(ee_W2[n * F + k], h1[k], acc); h2[n] = acc / (1.0f + expf(-acc)); }
        double mean = 0; for (float x : h2) mean += x; mean /= F;
        double var = 0; for (float x : h2) var += (x - mean) * (x - mean); var /= F;
        const float rstd = (float)(1.0 / std::sqrt(var + 1e-5));
        for (int n = 0; n < F; ++n) ef_tab[(size_t)t * 128 + n] = (h2[n] - (float)mean) * rstd * ee_g[n] + ee_b[n];      // columns F..127 stay 0
    }
    if (!mk) {      // dense edge embedding of endpoint models: the same MLP as a device kernel over the pair rows
        c->edge_embed.K1p = pad8(te); c->edge_embed.H = 128; c->edge_embed.O = 128;
        pack_linear(B, c->edge_embed.W1, ee_W1, F, te, pad8(te), 128, ident); pad_vec(B, c->edge_embed.b1, ee_b1, F, 128);
        pack_linear(B, c->edge_embed.W2, ee_W2, F, F, 128, 128, ident); pad_vec(B, c->edge_embed.b2, ee_b2, F, 128);
        pad_vec(B, c->edge_ln_g, ee_g, F, 128); pad_vec(B, c->edge_ln_b, ee_b, F, 128);
    }
    // ---- self-conditioning
    if (cfg->self_conditioning) {
        const std::string p = "self_conditioning_residual_layer.";
        const int kin = S + na + nc + 32, kinp = 256 + na + nc + 32;       // reference / tile input widths: [s | p_a | p_c | rbf]
        const float* W1 = bl.get(p + "node_residual_mlp.0.weight", S, kin); const float* b1 = bl.get(p + "node_residual_mlp.0.bias", S);
        const float* W2 = bl.get(p + "node_residual_mlp.2.weight", S, S); const float* b2 = bl.get(p + "node_residual_mlp.2.bias", S);
        const int kie = F + ne + 32;
        const float* E1 = bl.get(p + "edge_residual_mlp.0.weight", F, kie); const float* eb1 = bl.get(p + "edge_residual_mlp.0.bias", F);
        const float* E2 = bl.get(p + "edge_residual_mlp.2.weight", F, F); const float* eb2 = bl.get(p + "edge_residual_mlp.2.bias", F);
        if (!W1 || !b1 || !W2 || !b2 || !E1 || !eb1 || !E2 || !eb2) return bail(bl.err);
        c->sc_node.K1p = pad8(kinp); c->sc_node.H = 256; c->sc_node.O = 256;
        pack_linear(B, c->sc_node.W1, W1, S, kin, pad8(kinp), 256, [&](int k) { return k < 256 ? (k < S ? k : -1) : S + (k - 256); });
        if (S == 256 && HX == 0 && !prec_two_plane(cfg->precision)) {      // 4-row node tiles of small batches (fm_k_mlp4): K padded to 320, quad-row packed
            pack_linear4(B, c->sc_node_W1q, W1, S, kin, 320, [&](int k) { return k < 256 ? k : (k - 256 < na + nc + 32 ? S + (k - 256) : -1); });
            pack_linear4(B, c->sc_node_W2q, W2, S, S, 256, ident);
        }
        pad_vec(B, c->sc_node.b1, b1, S, 256);
        pack_linear(B, c->sc_node.W2, W2, S, S, 256, 256, ident);
        pad_vec(B, c->sc_node.b2, b2, S, 256);
        c->sc_edge.K1p = pad8(ne + 32); c->sc_edge.H = 128; c->sc_edge.O = 128;
        pack_linear(B, c->sc_edge.W1, E1, F, kie, pad8(ne + 32), 128, [&](int k) { return k < ne + 32 ? F + k : -1; });
        pad_vec(B, c->sc_edge.b1, eb1, F, 128);
        pack_linear(B, c->sc_edge.W2, E2, F, F, 128, 128, ident);
        pad_vec(B, c->sc_edge.b2, eb2, F, 128);
        for (int t = 0; t <= ne; ++t)
            for (int n = 0; n < F; ++n) {
                float acc = eb1[n];
                for (int k = 0; k < F; ++k) acc = fmaf(E1[(size_t)n * kie + k], ef_tab[(size_t)t * 128 + k], acc);
                T1[(size_t)t * 128 + n] = acc;
            }
    }
    B.put(c->ef_tab, ef_tab);
    B.put(c->T1, T1);
    // ---- convolutions
    c->conv.resize(cfg->n_convs);
    for (int i = 0; i < cfg->n_convs; ++i) {
        ConvW& cw = c->conv[i];
        const std::string p = "conv_layers." + std::to_string(i) + ".";
        const std::string k0 = p + "edge_message.0";
        const int kin0 = S + 32 + F + SD + H0 + 4;
        const float* Wh = bl.get(k0 + ".Wh", H0, H0);        // input vectors [x_diff | v_src (V) | v_dst_msg (HX)], hidden = max(in, out) = H0
        const float* Wcp = bl.get(k0 + ".Wcp", H0, 8);
        const float* Wu = bl.get(k0 + ".Wu", H0 + 4, V);
        const float* Ws = bl.get(k0 + ".to_feats_out.0.weight", S, kin0);
        const float* bs = bl.get(k0 + ".to_feats_out.0.bias", S);
        const float* Wg = bl.get(k0 + ".scalar_to_vector_gates.weight", V, S);
        const float* bg = bl.get(k0 + ".scalar_to_vector_gates.bias", V);
        if (!Wh || !Wcp || !Wu || !Ws || !bs || !Wg || !bg) return bail(bl.err);
        // hoisted per-node parts (input vector 0 is the displacement; 1.. are v_src)
        pack_linear(B, cw.Wps, Ws, S, kin0, 256, 256, [&](int k) { return k < S ? k : -1; });
        // hidden-vector row layout (FmGvpTile): [hidden (H0) | cp (4, filled by the kernel) | 0 .. KU0) | Vcp sources (8) | 0 .. PVW)
        auto hrow = [&](int vin_row, int n) -> float {
            if (n < H0) return Wh[(size_t)vin_row * H0 + n];
            if (n >= KU0 && n < KU0 + 8) return Wcp[(size_t)vin_row * 8 + (n - KU0)];
            return 0.f; };
        B.put(cw.Wpv, pack(V, PVW, [&](int k, int n) -> float { return hrow(1 + k, n); }));
        {
            std::vector<float> w0(PVW, 0.f);
            for (int n = 0; n < PVW; ++n) w0[n] = hrow(0, n);
            B.put(cw.w0, w0);
        }
        FmGvpW& g0 = cw.msg[0];
        g0.Wv1 = nullptr;
        B.put(g0.Wu, pack(KU0, V, [&](int k, int n) -> float { return k < H0 + 4 ? Wu[k * V + n] : 0.f; }));
        // K order of the first scalar linear: [rbf(32) | ef(128 columns, F real) | sh(V+5) | 0]; reference column order
        // [s_src(S) | rbf(32) | ef(F) | sh(V+5)] (gvp.py:532-539,118)
        pack_linear(B, g0.Ws, Ws, S, kin0, 160 + KU0, 256, [&](int k) {
            if (k < 32) return S + k;
            if (k < 160) return k - 32 < F ? S + 32 + (k - 32) : -1;
            return k < 160 + H0 + 4 ? S + 32 + F + SD + (k - 160) : -1; });
        pack_linear(B, cw.Ws_slab, Ws, S, kin0, 160, 256, [&](int k) { return k < 32 ? S + k : (k - 32 < F ? S + 32 + (k - 32) : -1); });
        pack_linear(B, cw.Ws_sh, Ws, S, kin0, KU0, 256, [&](int k) { return k < H0 + 4 ? S + 32 + F + SD + k : -1; });
        pad_vec(B, g0.bs, bs, S, 256);
        pack_linear(B, g0.Wg, Wg, V, S, 256, V, [&](int k) { return k < S ? k : -1; });
        pad_vec(B, g0.bg, bg, V, V);
        const bool sp = prec_two_plane(cfg->precision);                                       // split-precision NODE / EdgeUpdate kernels: the two-plane modes only
        const int sp_msg = prec_two_plane(cfg->precision) ? 2 : cfg->precision == FM_PREC_BF16X6 ? 3 : 0;      // 16-bit planes of the edge-message GEMM operands
        if (sp_msg) {
            pack_linear_sp(B, g0.Ws_sp, Ws, S, kin0, 160 + KU0, 256, [&](int k) {
                if (k < 32) return S + k;
                if (k < 160) return k - 32 < F ? S + 32 + (k - 32) : -1;
                return k < 160 + H0 + 4 ? S + 32 + F + SD + (k - 160) : -1; }, sp_msg);
            pack_linear_sp(B, g0.Wg_sp, Wg, V, S, 256, V, [&](int k) { return k < S ? k : -1; }, sp_msg);
        }
        if (HX > 0) {
            // destination-node terms of the first edge GVP, hoisted per node: vectors through [Wh | Wcp] rows V+1.., scalars through Ws columns S+32+F..
            B.put(cw.Wpvd, pack(8, PVW, [&](int k, int n) -> float { return k < HX ? hrow(V + 1 + k, n) : 0.f; }));
            pack_linear(B, cw.Wsd, Ws, S, kin0, 256, 256, [&](int k) { return k < SD ? S + 32 + F + k : -1; });
            // the projection GVP itself (gvp.py:304-311): V -> HX vectors, S -> SD scalars, hidden V, no cross-product features
            const std::string kp = p + "dst_feat_msg_projection";
            const float* pWh = bl.get(kp + ".Wh", V, V); const float* pWu = bl.get(kp + ".Wu", V, HX);
            const float* pWs = bl.get(kp + ".to_feats_out.0.weight", SD, V + S); const float* pbs = bl.get(kp + ".to_feats_out.0.bias", SD);
            const float* pWg = bl.get(kp + ".scalar_to_vector_gates.weight", HX, SD); const float* pbg = bl.get(kp + ".scalar_to_vector_gates.bias", HX);
            if (!pWh || !pWu || !pWs || !pbs || !pWg || !pbg) return bail(bl.err);
            FmGvpW& dp = cw.dproj;
            B.put(dp.Wv1, pack(V, V + 16, [&](int k, int n) -> float { return n < V ? pWh[k * V + n] : 0.f; }));         // Wcp = 0
            B.put(dp.Wu, pack(V + 8, 16, [&](int k, int n) -> float { return (k < V && n < HX) ? pWu[k * HX + n] : 0.f; }));
            pack_linear(B, dp.Ws, pWs, SD, V + S, 256 + V + 8, 256, [&](int k) { return k < 256 ? (k < S ? k : -1) : (k < 256 + V ? S + (k - 256) : -1); });
            pad_vec(B, dp.bs, pbs, SD, 256);
            pack_linear(B, dp.Wg, pWg, HX, SD, 256, 16, [&](int k) { return k < SD ? k : -1; });
            pad_vec(B, dp.bg, pbg, HX, 16);
        }
        for (int g = 1; g < 3; ++g) if (!pack_gvp(B, bl, p + "edge_message." + std::to_string(g), V, S, V, cw.msg[g], sp_msg)) return bail(bl.err);
        const bool rows4 = S == 256 && HX == 0 && !sp;       // 4-node tiles exist for full-width f32 models on the fused node sequence
        for (int g = 0; g < 3; ++g) if (!pack_gvp(B, bl, p + "node_update." + std::to_string(g), V, S, V, cw.upd[g], sp ? 2 : 0, rows4)) return bail(bl.err);
        if (rows4) pack_linear4(B, cw.Wps4, Ws, S, kin0, 256, [&](int k) { return k < S ? k : -1; });
        if (sp) pack_linear_sp(B, cw.Wps_sp, Ws, S, kin0, 256, 256, [&](int k) { return k < S ? k : -1; });
        const float* l1g = bl.get(p + "message_layer_norm.feat_norm.weight", S); const float* l1b = bl.get(p + "message_layer_norm.feat_norm.bias", S);
        const float* l2g = bl.get(p + "update_layer_norm.feat_norm.weight", S); const float* l2b = bl.get(p + "update_layer_norm.feat_norm.bias", S);
        if (!l1g || !l1b || !l2g || !l2b) return bail(bl.err);
        pad_vec(B, cw.ln1_g, l1g, S, 256); pad_vec(B, cw.ln1_b, l1b, S, 256);
        pad_vec(B, cw.ln2_g, l2g, S, 256); pad_vec(B, cw.ln2_b, l2b, S, 256);
    }
    // ---- molecule updaters (only those the schedule uses; index 0 is dead when convs_per_update == 1)
    c->upd.resize(cfg->n_updaters);
    for (int u = 0; u < cfg->n_updaters; ++u) {
        bool used = false;
        for (int i = 0; i < cfg->n_convs; ++i) used |= cfg->update_after[i] == u;
        if (!used) continue;
        UpdW& uw = c->upd[u];
        const std::string p = "node_position_updaters." + std::to_string(u) + ".gvps.";
        const bool spu = prec_two_plane(cfg->precision);
        const bool rows4u = S == 256 && HX == 0 && !spu;
        if (!pack_gvp(B, bl, p + "0", V, S, V, uw.pos[0], spu ? 2 : 0, rows4u) || !pack_gvp(B, bl, p + "1", V, S, V, uw.pos[1], spu ? 2 : 0, rows4u) || !pack_gvp(B, bl, p + "2", V, S, 1, uw.pos[2], spu ? 2 : 0, rows4u))
            return bail(bl.err);
        const std::string q = "edge_updaters." + std::to_string(u) + ".";
        const bool with_d = !cfg->edge_update_no_distance;
        const int kin = 2 * S + F + (with_d ? 32 : 0);
        const float* W1 = bl.get(q + "edge_update_fn.0.weight", F, kin); const float* b1 = bl.get(q + "edge_update_fn.0.bias", F);
        const float* W2 = bl.get(q + "edge_update_fn.2.weight", F, F); const float* b2 = bl.get(q + "edge_update_fn.2.bias", F);
        const float* g = bl.get(q + "edge_norm.weight", F); const float* be = bl.get(q + "edge_norm.bias", F);
        if (!W1 || !b1 || !W2 || !b2 || !g || !be) return bail(bl.err);
        // input order [s_src(S) | s_dst(S) | ef(F) | d(32)] (vector_field.py:870-877); tile: Asd = [W1_src s | W1_dst s] (2 x 128 columns)
        B.put(uw.Wasd, pack(256, 256, [&](int k, int n) -> float {
            const int o = n < 128 ? n : n - 128;
            if (k >= S || o >= F) return 0.f;
            return W1[(size_t)o * kin + (n < 128 ? 0 : S) + k]; }));
        if (rows4u) B.putv(uw.Wasd4, pack4(256, [&](int k, int n) -> float {
            const int o = n < 128 ? n : n - 128;
            if (k >= S || o >= F) return 0.f;
            return W1[(size_t)o * kin + (n < 128 ? 0 : S) + k]; }));
        // without update_edge_w_distance the rbf(d) block of the tile meets zero weights
        pack_linear(B, uw.W1, W1, F, kin, 160, 128, [&](int k) { return k < 128 ? (k < F ? 2 * S + k : -1) : (with_d ? 2 * S + F + (k - 128) : -1); });
        pad_vec(B, uw.b1, b1, F, 128);
        pack_linear(B, uw.W2, W2, F, F, 128, 128, ident);
        pad_vec(B, uw.b2, b2, F, 128);
        if (prec_two_plane(cfg->precision)) {
            B.putv(uw.Wasd_sp, pack_sp(256, 256, [&](int k, int n) -> float {
                const int o = n < 128 ? n : n - 128;
                if (k >= S || o >= F) return 0.f;
                return W1[(size_t)o * kin + (n < 128 ? 0 : S) + k]; }, 2, B.sp_fmt));
            pack_linear_sp(B, uw.W1_sp, W1, F, kin, 160, 128, [&](int k) { return k < 128 ? (k < F ? 2 * S + k : -1) : (with_d ? 2 * S + F + (k - 128) : -1); });
            pack_linear_sp(B, uw.W2_sp, W2, F, F, 128, 128, ident);
        }
        pad_vec(B, uw.ln_g, g, F, 128); pad_vec(B, uw.ln_b, be, F, 128);
    }
    // ---- output heads
    {
        const float* W1 = bl.get("node_output_head.0.weight", S, S); const float* b1 = bl.get("node_output_head.0.bias", S);
        const float* W2 = bl.get("node_output_head.2.weight", na + nc, S); const float* b2 = bl.get("node_output_head.2.bias", na + nc);
        const float* E1 = bl.get("to_edge_logits.0.weight", F, F); const float* eb1 = bl.get("to_edge_logits.0.bias", F);
        const float* E2 = bl.get("to_edge_logits.2.weight", ne, F); const float* eb2 = bl.get("to_edge_logits.2.bias", ne);
        if (!W1 || !b1 || !W2 || !b2 || !E1 || !eb1 || !E2 || !eb2) return bail(bl.err);
        c->node_head.K1p = 256; c->node_head.H = 256; c->node_head.O = pad16(na + nc);
        pack_linear(B, c->node_head.W1, W1, S, S, 256, 256, ident); pad_vec(B, c->node_head.b1, b1, S, 256);
        pack_linear(B, c->node_head.W2, W2, na + nc, S, 256, pad16(na + nc), ident); pad_vec(B, c->node_head.b2, b2, na + nc, pad16(na + nc));
        if (S == 256 && HX == 0 && !prec_two_plane(cfg->precision)) {
            pack_linear4(B, c->node_head_W1q, W1, S, S, 256, ident);
            pack_linear4(B, c->node_head_W2q, W2, na + nc, S, 256, ident, 1);      // N = 64 (na + nc <= 32 real columns): one column group, K over all eight waves
        }
        c->edge_head.K1p = 128; c->edge_head.H = 128; c->edge_head.O = 16;
        pack_linear(B, c->edge_head.W1, E1, F, F, 128, 128, ident); pad_vec(B, c->edge_head.b1, eb1, F, 128);
        pack_linear(B, c->edge_head.W2, E2, ne, F, 128, 16, ident); pad_vec(B, c->edge_head.b2, eb2, ne, 16);
    }
    // ---- upload
    c->arena_bytes = B.A.h.size() * sizeof(float);
    hipError_t e = hipMalloc((void**)&c->arena, c->arena_bytes);
    if (e != hipSuccess) { delete c; return fail(nullptr, FM_ERR_NOMEM, "fm_create: hipMalloc(%zu) failed: %s", B.A.h.size() * 4, hipGetErrorString(e)); }
    e = hipMemcpy(c->arena, B.A.h.data(), c->arena_bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(c->arena); delete c; return fail(nullptr, FM_ERR_HIP, "fm_create: weight upload failed: %s", hipGetErrorString(e)); }
    for (const Fix& f : B.fix) *f.slot = c->arena + f.off * sizeof(float);
    // ---- dynamic LDS opt-in (up to 160 KiB per workgroup on gfx950)
    // launch-tuning overrides (fm_config, ABI 5; 0 = automatic everywhere)
    c->tm_edge_forced = cfg->tile_edge; c->tm_node_forced = cfg->tile_node;
    c->tm_eupd = cfg->tile_edge_update == 64 ? 64 : 32;
    c->xcd_swizzle = cfg->xcd_swizzle >= 0; c->fuse_node = cfg->fuse_node >= 0; c->fuse_head = cfg->fuse_node == 0 || cfg->fuse_node == 1;
    c->pair_mlps_forced = cfg->pair_mlps == 0 ? -1 : (cfg->pair_mlps > 0);
    c->small_mlp_forced = cfg->mlp_small_tiles == 0 ? -1 : (cfg->mlp_small_tiles > 0);
    c->mlp4_forced = cfg->mlp_small_tiles == 0 ? -1 : (cfg->mlp_small_tiles == 2);
    // pair-slab hoist: the convolutions that run before any molecule update see pair-symmetric edge features and distances
    c->n_pq = 0;
    c->pq_forced = cfg->pair_slab > 0;
    c->canonical = cfg->canonical >= 0;
    if (cfg->pair_slab >= 0 && HX == 0 && cfg->precision == FM_PREC_F32 && cfg->self_conditioning)
        for (int i = 0; i < cfg->n_convs && i < 2; ++i) {
            bool clean = true;
            for (int j = 0; j < i; ++j) clean &= cfg->update_after[j] < 0;
            if (!clean) break;
            c->n_pq = i + 1;
        }
    auto tile_ok = [](int t) { return t == 0 || t == 16 || t == 32 || t == 64; };
    auto rg_tile = [](int t) { return t == 4 || t == 8 || t == 12 || t == 20; };
    if (!tile_ok(c->tm_edge_forced) || !(tile_ok(c->tm_node_forced) || rg_tile(c->tm_node_forced))) {
        (void)hipFree(c->arena); delete c;
        return fail(nullptr, FM_ERR_INVALID, "fm_create: fm_config.tile_edge must be 0 (automatic), 16, 32 or 64; tile_node additionally 4, 8, 12 or 20");
    }
    {
        int dev = 0; hipDeviceProp_t prop{};
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            c->n_cus = prop.multiProcessorCount;
    }
    fm_set_lds_msg_v32(); fm_set_lds_msg_v16(); fm_set_lds_node();      // the heavy families' instances, in their own translation units
    set_lds(fm_k_node_proj<32>, lds_proj(32)); set_lds(fm_k_node_proj<16>, lds_proj(16));
    set_lds(fm_k_node_proj<32, 16>, lds_proj(32, 16)); set_lds(fm_k_node_proj<16, 16>, lds_proj(16, 16));
    set_lds(fm_k_edge_update_sp<32>, lds_edge_upd_sp(32)); set_lds(fm_k_edge_update_sp<32, 1>, lds_edge_upd_sp(32));
    set_lds(fm_k_edge_update<32, false>, lds_edge_upd(32)); set_lds(fm_k_edge_update<64, false>, lds_edge_upd(64)); set_lds(fm_k_edge_update<32, true>, lds_edge_upd(32)); set_lds(fm_k_edge_update<32, false, true>, lds_edge_upd(32));
    const size_t mlp_max = lds_mlp(ld_for(pad8(256 + 16 + 16 + 32)), 260);
    set_lds(fm_k_mlp2<FM_MLP_TABLE>, mlp_max); set_lds(fm_k_mlp2<FM_MLP_SC_NODE>, mlp_max); set_lds(fm_k_mlp2<FM_MLP_NODE_HEAD>, mlp_max);
    set_lds(fm_k_mlp2<FM_MLP_EDGE_HEAD>, mlp_max); set_lds(fm_k_mlp2<FM_MLP_SC_EDGE>, mlp_max); set_lds(fm_k_mlp2<FM_MLP_SC_EDGE, 32>, mlp_max); set_lds(fm_k_mlp2<FM_MLP_EDGE_HEAD, 32>, mlp_max);
    set_lds(fm_k_mlp2_pair<FM_MLP_SC_NODE, FM_MLP_SC_EDGE>, mlp_max); set_lds(fm_k_mlp2_pair<FM_MLP_NODE_HEAD, FM_MLP_EDGE_HEAD>, mlp_max);
    const size_t mlp_small = lds_mlp(ld_for(pad8(256 + 16 + 16 + 32)), 260, 16);
    set_lds(fm_k_mlp4<FM_MLP4_SC_NODE>, FM_MLP4_LDS_BYTES); set_lds(fm_k_mlp4<FM_MLP4_NODE_HEAD>, FM_MLP4_LDS_BYTES); set_lds(fm_k_mlp4<FM_MLP4_PROJ0>, FM_MLP4_LDS_BYTES);
    set_lds(fm_k_mlp4_pair<FM_MLP4_SC_NODE, FM_MLP_SC_EDGE>, mlp_small); set_lds(fm_k_mlp4_pair<FM_MLP4_NODE_HEAD, FM_MLP_EDGE_HEAD>, mlp_small);
    set_lds(fm_k_mlp2<FM_MLP_SC_NODE, 16>, mlp_small); set_lds(fm_k_mlp2<FM_MLP_NODE_HEAD, 16>, mlp_small); set_lds(fm_k_mlp2<FM_MLP_TABLE, 16>, mlp_small);
    set_lds(fm_k_mlp2<FM_MLP_EDGE_HEAD, 16>, mlp_small); set_lds(fm_k_mlp2<FM_MLP_SC_EDGE, 16>, mlp_small);
    set_lds(fm_k_mlp2_pair<FM_MLP_SC_NODE, FM_MLP_SC_EDGE, 16>, mlp_small); set_lds(fm_k_mlp2_pair<FM_MLP_NODE_HEAD, FM_MLP_EDGE_HEAD, 16>, mlp_small);
    *out = c;
    return FM_OK;
}

int fm_destroy(fm_ctx* c) {
    if (!c) return FM_OK;
    for (auto& pe : c->prof_events) { (void)hipEventDestroy(pe.a); (void)hipEventDestroy(pe.b); }
    for (auto& e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->arena) (void)hipFree(c->arena);
    if (c->stage_ev) { (void)hipEventSynchronize(c->stage_ev); (void)hipEventDestroy(c->stage_ev); }
    if (c->stage) (void)hipHostFree(c->stage);
    delete c;
    return FM_OK;
}

// ---------------------------------------------------------------------------------------- workspace
struct WsLayout {
    int B, N, E, U, P, nmax, tab_rows, tab_kp, tm_edge, tm_node, node_rg, n_tiles_msg;
    size_t off_mol_tile, off_tile_desc, off_mol_node, off_mol_edge, off_mol_pair, off_node_mol, off_first_edge, off_esrc, off_edst, off_epair, off_pe0, off_pe1,
        off_pair_mol, off_s, off_v, off_xw, off_ef, off_Ps, off_Asd, off_PV, off_part_s, off_part_v, off_Psd, off_PVd, off_stab, off_bx, off_ba,
        off_bc, off_be, off_tap_s, off_tap_v, off_gid, off_sa1, off_sc1, off_se1, off_Q0, off_Q1, total;
};

static int ws_layout(fm_ctx* c, const int32_t* n_atoms, int B, WsLayout& w) {
    if (B <= 0) return fail(c, FM_ERR_INVALID, "batch of %d molecules", B);
    long long N = 0, E = 0;
    int nmax = 0;
    for (int i = 0; i < B; ++i) {
        const int n = n_atoms[i];
        if (n < 1) return fail(c, FM_ERR_INVALID, "molecule %d has %d atoms", i, n);      // a 1-atom molecule has no edges: its node rows simply receive no messages
        N += n; E += (long long)n * (n - 1); nmax = n > nmax ? n : nmax;
    }
    if (E > 0x7fffffffLL / 4) return fail(c, FM_ERR_INVALID, "batch too large for int32 edge indexing (%lld edges)", E);
    // per-node tables (Ps, Asd: 1 KiB rows) are gathered through buffer descriptors with 31-bit byte offsets
    if (N > 0x7fffffffLL / 1024) return fail(c, FM_ERR_INVALID, "batch too large: %lld nodes (limit %lld per bind; split the batch)", N, 0x7fffffffLL / 1024);
    const int V = c->V;
    w.B = B; w.N = (int)N; w.E = (int)E; w.U = (int)(E / 2);
    // tile sizes of this batch.  The edge-message kernel is bound by a CU's matrix pipe, so a launch takes (tiles per CU, rounded up) rounds of one tile time plus the
    // first tile's latency: measured on the MI355X (flowmol3 model, 1 .. 64 molecules x 47 atoms, profiles/r06t_*, r06u_*) 11 + 18 r16 us with 16-row tiles and
    // 13.5 + 31 r32 us with 32-row tiles, r = ceil(tiles / CUs) -- 32-row tiles do 16 rows in 15.5 instead of 18 us, 16-row tiles quantise in half the step.  The
    // cheaper of the two by that model (one molecule: 16 rows; 4, 5, 8, 9 molecules: 16; everything from 14 molecules on: 32).  Both give the same bits (canonical
    // arithmetic), so the choice is free to follow the batch size.
    {
        long long t16 = 0, t32 = 0;
        for (int i = 0; i < B; ++i) { const long long e = (long long)n_atoms[i] * (n_atoms[i] - 1); t16 += (e + 15) / 16; t32 += (e + 31) / 32; }
        const long long r16 = (t16 + c->n_cus - 1) / c->n_cus, r32 = (t32 + c->n_cus - 1) / c->n_cus;
        w.tm_edge = c->tm_edge_forced ? c->tm_edge_forced : (36 * r16 + 22 < 62 * r32 + 27 ? 16 : 32);
    }
    if ((c->HX || !c->cfg.has_mask) && (w.tm_edge > 32)) w.tm_edge = 32;
    // node tiles: 32 rows once the chip is full, 16 while 32-row tiles would leave CUs idle -- and, for full-width f32 models on the fused node
    // sequence, tiles of 4 / 8 / 12 nodes in the 16-row frame or 20 nodes in the 32-row frame (RG instances of fm_k_node_update) whenever such
    // tiles fit ONE per CU: the node kernel is a serial chain per tile whose scalar GEMMs scale with the tile height, so the smallest tile that
    // still gives every tile a CU of its own is the fastest (beyond one tile per CU the small tiles lose: each streams the full weights)
    const bool rg_ok = c->S == 256 && c->HX == 0 && c->cfg.precision == FM_PREC_F32 && c->fuse_node;
    int tn = c->tm_node_forced;
    if (!tn) {
        tn = (N + 31) / 32 <= c->n_cus ? 16 : 32;
        if (rg_ok) {      // the 4 RG-node instances keep the regular tiles' summation order (fm_wave_gemm4): the choice may follow the batch size in canonical mode
            static const int cand[] = {4, 8, 12, 16, 20};
            for (int r : cand) if ((N + r - 1) / r <= c->n_cus) { tn = r; break; }
        }
    }
    w.node_rg = 0;
    if (tn == 4 || tn == 8 || tn == 12 || tn == 20) {
        if (rg_ok) { w.node_rg = tn / 4; tn = tn == 20 ? 32 : 16; }
        else tn = tn == 20 ? 32 : 16;          // models the instances do not exist for take the frame's regular tile
    }
    w.tm_node = tn;
    if ((c->HX || !c->cfg.has_mask) && (w.tm_node > 32)) w.tm_node = 32;
    w.P = nmax > 1 ? (nmax - 2) / FM_CHUNK_E + 2 : 1; w.nmax = nmax;      // chunks of FM_CHUNK_E rows a destination's n - 1 in-edges can touch
    long long nt = 0;
    for (int i = 0; i < B; ++i) nt += ((long long)n_atoms[i] * (n_atoms[i] - 1) + w.tm_edge - 1) / w.tm_edge;      // every molecule starts a tile
    w.n_tiles_msg = (int)nt;
    w.tab_rows = c->tab_rows; w.tab_kp = c->tab_kp;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes, 256); return r; };
    w.off_mol_tile = take((size_t)(B + 1) * 4); w.off_tile_desc = take((size_t)w.n_tiles_msg * 16);
    w.off_mol_node = take((size_t)(B + 1) * 4); w.off_mol_edge = take((size_t)(B + 1) * 4); w.off_mol_pair = take((size_t)(B + 1) * 4);
    w.off_node_mol = take((size_t)N * 4); w.off_first_edge = take((size_t)N * 4);
    w.off_esrc = take((size_t)E * 4); w.off_edst = take((size_t)E * 4); w.off_epair = take((size_t)E * 4);
    w.off_pe0 = take((size_t)w.U * 4); w.off_pe1 = take((size_t)w.U * 4); w.off_pair_mol = take((size_t)w.U * 4);
    w.off_s = take((size_t)N * 256 * 4); w.off_v = take((size_t)N * 3 * V * 4); w.off_xw = take((size_t)N * 3 * 4);
    w.off_ef = take((size_t)E * 128 * 4);
    w.off_Ps = take((size_t)N * 256 * 4); w.off_Asd = take((size_t)N * 256 * 4); w.off_PV = take((size_t)N * 3 * c->PVW * 4);
    w.off_Psd = take(c->HX ? (size_t)N * 256 * 4 : 0); w.off_PVd = take(c->HX ? (size_t)N * 3 * c->PVW * 4 : 0);
    w.off_part_s = take((size_t)N * w.P * 256 * 4); w.off_part_v = take((size_t)N * w.P * 3 * V * 4);
    w.off_stab = take((size_t)align_up(w.tab_rows, FM_TM) * 256 * 4 * FM_TAB_SLOTS);
    w.off_bx = take((size_t)N * 3 * 4); w.off_ba = take((size_t)N * c->na * 4); w.off_bc = take((size_t)N * c->nc * 4); w.off_be = take((size_t)w.U * c->ne * 4);
    w.off_tap_s = take((size_t)N * 256 * 4); w.off_tap_v = take((size_t)N * 3 * V * 4);
    w.off_gid = take((size_t)B * 4);
    w.off_sa1 = take((size_t)N * 4); w.off_sc1 = take((size_t)N * 4); w.off_se1 = take((size_t)w.U * 4);
    const int npq = pq_convs(c, w.U);      // the pair-slab tables are as large as `ef` each: only batches that use them pay for them (8192 x 47 atoms: 13.3 GB without, 31.5 GB with)
    w.off_Q0 = take(npq > 0 ? (size_t)w.U * 256 * 4 : 0); w.off_Q1 = take(npq > 1 ? (size_t)w.U * 256 * 4 : 0);
    w.total = o;
    return FM_OK;
}

int fm_workspace_bytes(fm_ctx* c, const int32_t* n_atoms, int B, size_t* bytes) {
    if (!c || !n_atoms || !bytes) return fail(c, FM_ERR_INVALID, "fm_workspace_bytes: null argument");
    WsLayout w;
    int rc = ws_layout(c, n_atoms, B, w);
    if (rc) return rc;
    *bytes = w.total;
    return FM_OK;
}

// Pinned staging for `n_ints` int32 about to be copied to the device on a stream: waits (host-side) only for the copies of the PREVIOUS
// use of the staging buffer, never for the stream.
static int stage_acquire(fm_ctx* c, hipStream_t st, size_t n_ints) {
    {   // setup calls wait on an event and (re)allocate pinned memory: both are illegal while the stream is being captured
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
            return fail(c, FM_ERR_STATE, "fm_batch_bind / fm_set_molecule_ids are setup calls and must stay outside a stream capture");
    }
    if (!c->stage_ev) FM_HIP(c, hipEventCreateWithFlags(&c->stage_ev, hipEventDisableTiming));
    if (c->stage_busy) { FM_HIP(c, hipEventSynchronize(c->stage_ev)); c->stage_busy = false; }
    if (c->stage_cap < n_ints) {
        if (c->stage) { FM_HIP(c, hipHostFree(c->stage)); c->stage = nullptr; c->stage_cap = 0; }
        const size_t cap = n_ints < 4096 ? 4096 : n_ints + n_ints / 2;
        FM_HIP(c, hipHostMalloc((void**)&c->stage, cap * sizeof(int32_t), hipHostMallocDefault));
        c->stage_cap = cap;
    }
    return FM_OK;
}

static int stage_release(fm_ctx* c, hipStream_t st) {
    FM_HIP(c, hipEventRecord(c->stage_ev, st));
    c->stage_busy = true;
    return FM_OK;
}

int fm_batch_bind(fm_ctx* c, void* stream, const int32_t* n_atoms, int B, void* workspace, size_t bytes) {
    if (!c || !n_atoms || !workspace) return fail(c, FM_ERR_INVALID, "fm_batch_bind: null argument");
    WsLayout w;
    int rc = ws_layout(c, n_atoms, B, w);
    if (rc) return rc;
    if (bytes < w.total) return fail(c, FM_ERR_INVALID, "fm_batch_bind: workspace of %zu bytes < required %zu", bytes, w.total);
    if ((uintptr_t)workspace % 256) return fail(c, FM_ERR_INVALID, "fm_batch_bind: workspace must be 256-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    char* base = (char*)workspace;
    rc = stage_acquire(c, st, 4 * (size_t)(B + 1) + (size_t)B);
    if (rc) return rc;
    int32_t* no = c->stage; int32_t* eo = no + (B + 1); int32_t* po = eo + (B + 1); int32_t* to = po + (B + 1); int32_t* ids = to + (B + 1);
    no[0] = eo[0] = po[0] = to[0] = 0;
    for (int i = 0; i < B; ++i) {
        const int n = n_atoms[i];
        no[i + 1] = no[i] + n; eo[i + 1] = eo[i] + n * (n - 1); po[i + 1] = po[i] + n * (n - 1) / 2; ids[i] = i;
        to[i + 1] = to[i] + (n * (n - 1) + w.tm_edge - 1) / w.tm_edge;
    }
    {   // a copy that fails after earlier ones were enqueued must not leave the staging buffer unguarded: the event is recorded either way
        hipError_t e_ = hipMemcpyAsync(base + w.off_mol_node, no, (size_t)(B + 1) * 4, hipMemcpyHostToDevice, st);
        if (e_ == hipSuccess) e_ = hipMemcpyAsync(base + w.off_mol_edge, eo, (size_t)(B + 1) * 4, hipMemcpyHostToDevice, st);
        if (e_ == hipSuccess) e_ = hipMemcpyAsync(base + w.off_mol_pair, po, (size_t)(B + 1) * 4, hipMemcpyHostToDevice, st);
        if (e_ == hipSuccess) e_ = hipMemcpyAsync(base + w.off_mol_tile, to, (size_t)(B + 1) * 4, hipMemcpyHostToDevice, st);
        if (e_ == hipSuccess) e_ = hipMemcpyAsync(base + w.off_gid, ids, (size_t)B * 4, hipMemcpyHostToDevice, st);
        rc = stage_release(c, st);
        if (e_ != hipSuccess) return fail(c, FM_ERR_HIP, "fm_batch_bind: descriptor copy failed: %s", hipGetErrorString(e_));
        if (rc) return rc;
    }
    FmBatch& b = c->b;
    b.B = B; b.N = w.N; b.E = w.E; b.U = w.U; b.P = w.P;
    b.n_tiles = w.n_tiles_msg; b.tile_rows = w.tm_edge;
    b.mol_tile_off = (const int*)(base + w.off_mol_tile); b.tile_desc = (int4*)(base + w.off_tile_desc);
    b.mol_node_off = (const int*)(base + w.off_mol_node); b.mol_edge_off = (const int*)(base + w.off_mol_edge); b.mol_pair_off = (const int*)(base + w.off_mol_pair);
    b.node_mol = (int*)(base + w.off_node_mol); b.node_first_edge = (int*)(base + w.off_first_edge);
    b.e_src = (int*)(base + w.off_esrc); b.e_dst = (int*)(base + w.off_edst); b.e_pair = (int*)(base + w.off_epair);
    b.p_e0 = (int*)(base + w.off_pe0); b.p_e1 = (int*)(base + w.off_pe1); b.pair_mol = (int*)(base + w.off_pair_mol);
    c->s = (float*)(base + w.off_s); c->v = (float*)(base + w.off_v); c->xw = (float*)(base + w.off_xw); c->ef = (float*)(base + w.off_ef);
    c->Ps = (float*)(base + w.off_Ps); c->Asd = (float*)(base + w.off_Asd); c->PV = (float*)(base + w.off_PV);
    c->part_s = (float*)(base + w.off_part_s); c->part_v = (float*)(base + w.off_part_v);
    c->Psd = (float*)(base + w.off_Psd); c->PVd = (float*)(base + w.off_PVd);
    c->s_tab = c->s_tab_base = (float*)(base + w.off_stab);
    c->tab_slot_floats = (size_t)align_up(w.tab_rows, FM_TM) * 256;
    c->boot.x = (float*)(base + w.off_bx); c->boot.a = (float*)(base + w.off_ba); c->boot.c = (float*)(base + w.off_bc); c->boot.e = (float*)(base + w.off_be);
    c->tap_s = (float*)(base + w.off_tap_s); c->tap_v = (float*)(base + w.off_tap_v);
    c->mol_gid = (int*)(base + w.off_gid);
    c->sa1 = (int32_t*)(base + w.off_sa1); c->sc1 = (int32_t*)(base + w.off_sc1); c->se1 = (int32_t*)(base + w.off_se1);
    c->Q[0] = (float*)(base + w.off_Q0); c->Q[1] = (float*)(base + w.off_Q1);
    c->n_tiles_e = (w.E + FM_TM - 1) / FM_TM; c->n_tiles_n = (w.N + FM_TM - 1) / FM_TM; c->n_tiles_u = (w.U + FM_TM - 1) / FM_TM;
    Launch L{c, st};
    const int work = w.E > w.N ? w.E : w.N;
    L("batch_setup", fm_k_batch_setup, dim3((work + 255) / 256), dim3(256), 0, b);
    if (L.rc) return L.rc;
    c->bound = true; c->nmax = w.nmax; c->tm_edge = w.tm_edge; c->tm_node = w.tm_node; c->node_rg = w.node_rg; c->n_tiles_msg = w.n_tiles_msg;
    return FM_OK;
}

int fm_remove_com(fm_ctx* c, void* stream, float* x) {
    if (!c || !x) return fail(c, FM_ERR_INVALID, "fm_remove_com: null argument");
    if (!c->bound) return fail(c, FM_ERR_STATE, "fm_remove_com: no batch bound");
    Launch L{c, (hipStream_t)stream};
    L("remove_com", fm_k_remove_com, dim3(c->b.B), dim3(64), 0, x, (const int*)c->b.mol_node_off);
    return L.rc;
}

int fm_set_molecule_ids(fm_ctx* c, void* stream, const int32_t* ids_host) {
    if (!c) return fail(c, FM_ERR_INVALID, "fm_set_molecule_ids: null context");
    if (!c->bound) return fail(c, FM_ERR_STATE, "fm_set_molecule_ids: no batch bound");
    int rc = stage_acquire(c, (hipStream_t)stream, (size_t)c->b.B);
    if (rc) return rc;
    for (int i = 0; i < c->b.B; ++i) c->stage[i] = ids_host ? ids_host[i] : i;
    const hipError_t e_ = hipMemcpyAsync(c->mol_gid, c->stage, (size_t)c->b.B * 4, hipMemcpyHostToDevice, (hipStream_t)stream);
    rc = stage_release(c, (hipStream_t)stream);
    if (e_ != hipSuccess) return fail(c, FM_ERR_HIP, "fm_set_molecule_ids: copy failed: %s", hipGetErrorString(e_));
    return rc;
}

int fm_prior_philox(fm_ctx* c, void* stream, uint64_t seed, float* x0) {
    if (!c || !x0) return fail(c, FM_ERR_INVALID, "fm_prior_philox: null argument");
    if (!c->bound) return fail(c, FM_ERR_STATE, "fm_prior_philox: no batch bound");
    Launch L{c, (hipStream_t)stream};
    L("prior_philox", fm_k_prior_philox, dim3(c->b.B), dim3(64), 0, x0, (const int*)c->b.mol_node_off, (const int*)c->mol_gid,
      (unsigned)(seed & 0xffffffffu), (unsigned)(seed >> 32));
    return L.rc;
}

int fm_forward(fm_ctx* c, void* stream, const fm_state* state, const float* temb, const fm_dst* prev, int bootstrap, int remove_com,
               const fm_dst* out) {
    if (!c || !state || !temb || !out) return fail(c, FM_ERR_INVALID, "fm_forward: null argument");
    if (!c->bound) return fail(c, FM_ERR_STATE, "fm_forward: no batch bound");
    if (!c->cfg.has_mask) return fail(c, FM_ERR_INVALID, "fm_forward: endpoint-parameterised model (continuous inputs): use fm_forward_dense");
    return forward_impl(c, (hipStream_t)stream, state, temb, prev, bootstrap, remove_com ? 1 : 0, out);
}

int fm_forward_dense(fm_ctx* c, void* stream, const fm_dense_state* state, const float* temb, int remove_com, const fm_dst* out) {
    if (!c || !state || !temb || !out) return fail(c, FM_ERR_INVALID, "fm_forward_dense: null argument");
    if (!c->bound) return fail(c, FM_ERR_STATE, "fm_forward_dense: no batch bound");
    if (c->cfg.has_mask) return fail(c, FM_ERR_INVALID, "fm_forward_dense: this is a CTMC model (token inputs): use fm_forward");
    return evaluate(c, (hipStream_t)stream, nullptr, nullptr, remove_com ? 1 : 0, out, true, state, temb);
}

int fm_endpoint_step(fm_ctx* c, void* stream, const fm_dense_state* state, const fm_dst* dst, const fm_endpoint_scalars* sc) {
    if (!c || !state || !dst || !sc) return fail(c, FM_ERR_INVALID, "fm_endpoint_step: null argument");
    if (!c->bound) return fail(c, FM_ERR_STATE, "fm_endpoint_step: no batch bound");
    const FmBatch& b = c->b;
    FmEndpointStepArgs a{};
    float* xt[4] = {state->x_t, state->a_t, state->c_t, state->e_t};
    const float* x1[4] = {dst->x, dst->a, dst->c, dst->e};
    const int n[4] = {b.N * 3, b.N * c->na, b.N * c->nc, b.U * c->ne};
    int nmax = 0;
    for (int f = 0; f < 4; ++f) { a.xt[f] = xt[f]; a.x1[f] = x1[f]; a.n[f] = n[f]; a.coef[f] = sc->coef[f]; nmax = std::max(nmax, n[f]); }
    a.scale = sc->scale; a.dt = sc->dt;
    Launch L{c, (hipStream_t)stream};
    L("endpoint_step", fm_k_endpoint_step, dim3(std::min(4096, (nmax + 255) / 256), 4), dim3(256), 0, a);
    return L.rc;
}

int fm_ctmc_step(fm_ctx* c, void* stream, const fm_state* state, const fm_dst* dst, const fm_step_noise* noise, const fm_step_scalars* sc,
                 const fm_sampled* sampled) {
    if (!c || !state || !dst || !sc) return fail(c, FM_ERR_INVALID, "fm_ctmc_step: null argument");
    if (!c->bound) return fail(c, FM_ERR_STATE, "fm_ctmc_step: no batch bound");
    return ctmc_impl(c, (hipStream_t)stream, state, dst, noise, sc, sampled);
}

int fm_integrate(fm_ctx* c, void* stream, const fm_state* state, int n_steps, const fm_step_scalars* steps, const float* temb,
                 const fm_step_noise* noise, const fm_dst* prev0, const fm_dst* dst_a, const fm_dst* dst_b, const fm_traj_sink* sink,
                 int* final_dst) {
    if (!c || !state || !steps || !temb || !dst_a || !dst_b) return fail(c, FM_ERR_INVALID, "fm_integrate: null argument");
    if (!c->bound) return fail(c, FM_ERR_STATE, "fm_integrate: no batch bound");
    hipStream_t st = (hipStream_t)stream;
    const int tt = c->cfg.time_embedding_dim;
    const FmBatch& b = c->b;
    const fm_dst* prev = prev0;
    int cur = (prev0 && prev0->x == dst_a->x) ? 1 : 0;
    for (int i = 0; i < n_steps; ++i) {
        const fm_dst* out = cur == 0 ? dst_a : dst_b;
        const int boot = (!prev && steps[i].t == 0.0f) ? 1 : 0;      // prev is None and (t == 0).all(), vector_field.py:269-272
        // campbell steps: the COM removal of the endpoint positions runs inside the fused CTMC kernel (one launch and one copy less)
        const bool defer_com = steps[i].dfm_type == FM_DFM_CAMPBELL;
        // embedding tables of the next FM_TAB_SLOTS steps in one launch
        if (i % FM_TAB_SLOTS == 0) {
            const int nt = n_steps - i < FM_TAB_SLOTS ? n_steps - i : FM_TAB_SLOTS;
            const int rc0 = embed_table(c, st, temb + (size_t)i * tt, nt);
            if (rc0) return rc0;
        }
        int rc = forward_impl(c, st, state, temb + (size_t)i * tt, prev, boot, defer_com ? 2 : 1, out, i % FM_TAB_SLOTS);
        if (rc) return rc;
        fm_sampled smp{};
        fm_traj_sink frame{};      // step i's frames: the fused CTMC kernel writes them next to the state (no copy nodes per step)
        if (sink) {
            smp.a1 = sink->a1 ? sink->a1 + (size_t)i * b.N : nullptr;
            smp.c1 = sink->c1 ? sink->c1 + (size_t)i * b.N : nullptr;
            smp.e1 = sink->e1 ? sink->e1 + (size_t)i * b.U : nullptr;
            frame.x = sink->x ? sink->x + (size_t)i * b.N * 3 : nullptr;
            frame.a = sink->a ? sink->a + (size_t)i * b.N : nullptr;
            frame.c = sink->c ? sink->c + (size_t)i * b.N : nullptr;
            frame.e = sink->e ? sink->e + (size_t)i * b.U : nullptr;
            frame.x1 = sink->x1 ? sink->x1 + (size_t)i * b.N * 3 : nullptr;
        }
        const bool in_kernel = sink && steps[i].dfm_type == FM_DFM_CAMPBELL;
        rc = ctmc_impl(c, st, state, out, noise ? &noise[i] : nullptr, &steps[i], &smp, defer_com ? c->xw : nullptr, in_kernel ? &frame : nullptr);
        if (rc) return rc;
        if (sink && !in_kernel) {      // 'gat' steps (three small kernels): frames by copy
            Launch L{c, st};
            L.copy(frame.x, state->x_t, frame.x ? (size_t)b.N * 12 : 0);
            L.copy(frame.a, state->a_t, frame.a ? (size_t)b.N * 4 : 0);
            L.copy(frame.c, state->c_t, frame.c ? (size_t)b.N * 4 : 0);
            L.copy(frame.e, state->e_t, frame.e ? (size_t)b.U * 4 : 0);
            L.copy(frame.x1, out->x, frame.x1 ? (size_t)b.N * 12 : 0);
            if (L.rc) return L.rc;
        }
        prev = out;
        cur ^= 1;
    }
    if (final_dst) *final_dst = cur ^ 1;
    return FM_OK;
}

int fm_set_tap(fm_ctx* c, const char* name, void* dst) {
    if (!c || !name) return fail(c, FM_ERR_INVALID, "fm_set_tap: null argument");
    if (dst) c->taps[name] = dst; else c->taps.erase(name);
    return FM_OK;
}
int fm_clear_taps(fm_ctx* c) { if (c) c->taps.clear(); return FM_OK; }

int fm_batch_query(fm_ctx* c, void* stream, const char* name, int32_t* dst) {
    if (!c || !name || !dst) return fail(c, FM_ERR_INVALID, "fm_batch_query: null argument");
    if (!c->bound) return fail(c, FM_ERR_STATE, "fm_batch_query: no batch bound");
    const FmBatch& b = c->b;
    const std::string n = name;
    const int* src = nullptr; size_t cnt = 0;
    if (n == "e_src") { src = b.e_src; cnt = b.E; } else if (n == "e_dst") { src = b.e_dst; cnt = b.E; }
    else if (n == "e_pair") { src = b.e_pair; cnt = b.E; } else if (n == "p_e0") { src = b.p_e0; cnt = b.U; }
    else if (n == "p_e1") { src = b.p_e1; cnt = b.U; } else if (n == "node_mol") { src = b.node_mol; cnt = b.N; }
    else if (n == "pair_mol") { src = b.pair_mol; cnt = b.U; }
    else return fail(c, FM_ERR_INVALID, "fm_batch_query: unknown array %s", name);
    FM_HIP(c, hipMemcpyAsync(dst, src, cnt * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return FM_OK;
}

#ifdef FM_TRACE
int fm_trace_read(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(fm_trace), 16384 * 16 * 8); return 0; }
#endif
#ifdef FM_PHASE_TIMING
// dev-only (not part of the ABI header): read / reset the phase-cycle accumulators of a -DFM_PHASE_TIMING build
int fm_tlog_read(unsigned long long* out, int reset) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(fm_tlog), 64 * 8);
    if (reset) { unsigned long long z[64] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(fm_tlog), z, 64 * 8); }
    return 0;
}
#endif

int fm_stability(fm_ctx* c, void* stream, const fm_state* state, const uint32_t* table, int n_types, int fake_atom_token,
                 int explicit_aromaticity, int32_t* out) {
    if (!c || !state || !table || !out) return fail(c, FM_ERR_INVALID, "fm_stability: null argument");
    if (!c->bound) return fail(c, FM_ERR_STATE, "fm_stability: no batch bound");
    if (!state->a_t || !state->c_t || !state->e_t) return fail(c, FM_ERR_INVALID, "fm_stability: state tokens missing");
    Launch L{c, (hipStream_t)stream};
    FmStabArgs a{};
    a.b = c->b; a.a = state->a_t; a.c = state->c_t; a.e = state->e_t; a.table = table; a.n_types = n_types; a.n_charges = c->nc;
    a.fake_tok = fake_atom_token; a.ne = c->ne; a.arom = explicit_aromaticity; a.out = out;
    L("stability", fm_k_stability, dim3(c->b.B), dim3(64), (size_t)c->nmax * 8, a);
    return L.rc;
}

int fm_profile_enable(fm_ctx* c, int on) {
    if (!c) return FM_ERR_INVALID;
    c->prof = on != 0;
    if (on) {
        for (auto& pe : c->prof_events) { c->ev_pool.push_back(pe.a); c->ev_pool.push_back(pe.b); }
        c->prof_events.clear(); c->prof_acc.clear();
    }
    return FM_OK;
}

int fm_profile_get(fm_ctx* c, const char* kernel, double* total_ms, int64_t* launches) {
    if (!c || !kernel || !total_ms || !launches) return fail(c, FM_ERR_INVALID, "fm_profile_get: null argument");
    for (auto& pe : c->prof_events) {       // fold finished events into the accumulators
        FM_HIP(c, hipEventSynchronize(pe.b));
        float ms = 0.f;
        FM_HIP(c, hipEventElapsedTime(&ms, pe.a, pe.b));
        auto& acc = c->prof_acc[c->prof_names[pe.kid]];
        acc.first += ms; acc.second += 1;
        c->ev_pool.push_back(pe.a); c->ev_pool.push_back(pe.b);
    }
    c->prof_events.clear();
    auto it = c->prof_acc.find(kernel);
    if (it == c->prof_acc.end()) { *total_ms = 0; *launches = 0; return FM_OK; }
    *total_ms = it->second.first; *launches = it->second.second;
    return FM_OK;
}

}  // extern "C"
