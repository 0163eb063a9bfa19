// __global__ kernels of the FlowMol3 sampling hot path for gfx950.  See DESIGN.md for the map
// kernel -> reference function and the HBM data layout.  All kernels use 512-thread workgroups over
// row tiles whose height TM is a template parameter (fm_device.h: 32 / 16 rows for the GVP kernels,
// four nodes in a 16-row frame for the node kernel of a few molecules, 64 / 32 / 16 rows for the MLPs);
// weights come pre-packed in MFMA fragment order.
// The header is included by several translation units (fm_host.h): the non-template kernels are `static` (only the engine unit launches them; the others drop them).
#pragma once
#include <type_traits>

#include "fm_device.h"

// ------------------------------------------------------------------------------------------------
// batch descriptor (all device pointers; built by fm_k_batch_setup from the per-molecule offsets)
// ------------------------------------------------------------------------------------------------
struct FmBatch {
    int B, N, E, U;          // molecules, nodes, directed edges, unordered pairs (E = 2U)
    int P;                   // max number of FM_CHUNK_E-row chunks one destination's in-edges can span
    int n_tiles, tile_rows;  // edge-message tiles of the bound batch: tile_rows (16 | 32 | 64) rows each, every molecule's first edge row starts a tile
    const int* mol_node_off; // [B+1]
    const int* mol_edge_off; // [B+1]  (internal, dst-major directed edges)
    const int* mol_pair_off; // [B+1]  (reference upper-triangle order)
    int* node_mol;           // [N]
    int* node_first_edge;    // [N]  internal index of the first in-edge of the node
    int* e_src;              // [E]  global node id of the source
    int* e_dst;              // [E]
    int* e_pair;             // [E]  global pair id (reference order) of the unordered pair
    int* p_e0;               // [U]  internal edge id of (src=a -> dst=b), a<b   ("upper" edge)
    int* p_e1;               // [U]  internal edge id of (src=b -> dst=a)        ("lower" edge)
    int* pair_mol;           // [U]
    const int* mol_tile_off; // [B+1] first edge-message tile of every molecule (host-computed: sum of ceil(n (n - 1) / tile_rows))
    int4* tile_desc;         // [n_tiles] {first edge row, rows that exist (1 .. tile_rows), the molecule's first edge row, molecule}
};

// CANONICAL AGGREGATION ORDER (round 6).  In the reference a molecule's result is a function of the molecule and its noise rows only: every reduction is per
// molecule (gvp.py:491-492 message sum, ctmc_utils.py:11-20 purity counts, vector_field.py:347-350 centring).  Up to round 5 the f32 ORDER of the message sum
// depended on where the molecule sat in the batch: tiles were cut from the batch-global edge list, so a destination's in-edges met tile boundaries at
// offset-dependent places.  Now every molecule's directed-edge rows start at a tile boundary (the last tile of a molecule is ragged; the edge arrays stay
// compact -- only the tile -> row map changes) and a destination's rows are summed in chunks of FM_CHUNK_E rows counted from the MOLECULE's first row:
// rows of a chunk in order, chunks in order.  16 divides every tile height, so 16-, 32- and 64-row tiles produce the same partial sums, and a molecule
// alone, in a 1024-batch or in any shard of it gives bit-identical messages.  Cost at 1024 x 47 atoms: 68 tiles of 32 rows for 2162 rows = 0.65 % padding.
#define FM_CHUNK_E 16

// Internal edge order: per molecule, destination-major: edge (dst=i, src=j), j != i, sits at
// off + i*(n-1) + (j - (j>i)).  The reference's order (upper triangle row-major, then the same pairs
// swapped; flowmol/data_processing/utils.py:4-17) only matters at the boundary, where edge state is
// exchanged per unordered pair p(a,b) = a*(2n-a-1)/2 + (b-a-1), a<b.
static __global__ void __launch_bounds__(256) fm_k_batch_setup(FmBatch b) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid < b.N) {
        int lo = 0, hi = b.B;               // largest m with node_off[m] <= gid
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (b.mol_node_off[mid] <= gid) lo = mid; else hi = mid; }
        b.node_mol[gid] = lo;
        const int n = b.mol_node_off[lo + 1] - b.mol_node_off[lo];
        b.node_first_edge[gid] = b.mol_edge_off[lo] + (gid - b.mol_node_off[lo]) * (n - 1);
    }
    if (gid < b.E) {
        int lo = 0, hi = b.B;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (b.mol_edge_off[mid] <= gid) lo = mid; else hi = mid; }
        const int noff = b.mol_node_off[lo];
        const int n = b.mol_node_off[lo + 1] - noff;
        const int local = gid - b.mol_edge_off[lo];
        const int i = local / (n - 1), k = local % (n - 1);
        const int j = k + (k >= i ? 1 : 0);
        b.e_dst[gid] = noff + i;
        b.e_src[gid] = noff + j;
        const int a = i < j ? i : j, c = i < j ? j : i;
        const int pair = b.mol_pair_off[lo] + a * (2 * n - a - 1) / 2 + (c - a - 1);
        b.e_pair[gid] = pair;
        if (j < i) { b.p_e0[pair] = gid; b.pair_mol[pair] = lo; }   // src=a<b=dst : the reference's upper edge
        else b.p_e1[pair] = gid;
    }
    if (gid < b.n_tiles) {              // n_tiles <= E: every tile holds at least one row
        int lo = 0, hi = b.B;           // largest m with mol_tile_off[m] <= gid (molecules without edges own no tile: equal offsets, the search lands on the owner)
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (b.mol_tile_off[mid] <= gid) lo = mid; else hi = mid; }
        const int moff = b.mol_edge_off[lo];
        const int e0 = moff + (gid - b.mol_tile_off[lo]) * b.tile_rows;
        const int left = b.mol_edge_off[lo + 1] - e0;
        b.tile_desc[gid] = make_int4(e0, left < b.tile_rows ? left : b.tile_rows, moff, lo);
    }
}

// ------------------------------------------------------------------------------------------------
// generic two-layer MLP on 64-row tiles:  Y = post( act2( W2 * silu(W1 * x + b1) + b2 ) )
// The prologue that builds x and the epilogue are selected by MODE.
// ------------------------------------------------------------------------------------------------
enum FmMlpMode {
    FM_MLP_TABLE = 0,     // x = rows of a dense matrix; out = LayerNorm(silu(.))          (input embeddings)
    FM_MLP_SC_NODE = 1,   // x = [s_tab | p_a | p_c | rbf(|x_t - x1_prev|)]; out = s_tab + silu(.)
    FM_MLP_NODE_HEAD = 2, // x = s; out = softmax over [0,na) and [na,na+nc) of the logits
    FM_MLP_EDGE_HEAD = 3, // x = ef[e0]+ef[e1] per pair; out = softmax over ne logits
    FM_MLP_SC_EDGE = 4,   // x = [p_e | rbf(d(x1_prev)) - rbf(d(x_t))]; layer-1 adds T1[token]; out = ef_tab[token] + silu(.)
};

struct FmMlpArgs {
    int rows;                 // number of valid rows
    int K1p;                  // padded input width (multiple of 8)
    int H;                    // hidden width (multiple of 16, <= 256)
    int O;                    // padded output width (multiple of 16, <= 256)
    int ldx, ldh;             // LDS leading dims ((ld/4) odd, ldx >= max(K1p, O), ldh >= H)
    const float2* W1; const float* b1;
    const float2* W2; const float* b2;
    const float* ln_g; const float* ln_b; int ln_n;     // TABLE: LayerNorm affine + the REAL width its statistics run over (<= O)
    // generic sources / destinations
    const float* in;  int in_ld;              // TABLE: dense input
    int in_w;                                 // TABLE: valid input columns (0 = K1p); columns beyond read 0
    float* out; int out_ld;                   // TABLE / SC_NODE / heads
    // node-side
    const float* s_tab; const int* tok_a; const int* tok_c; int n_c1;   // s_tab row = tok_a*(n_c1)+tok_c
    const float* prev_a; const float* prev_c; const float* prev_x; const float* x_t;
    int na, nc, ne;
    float rbf_mu_step, rbf_inv_sigma;
    float* out2;                              // NODE_HEAD: charges
    // edge-side
    const float* ef; const int* p_e0; const int* p_e1;                   // EDGE_HEAD
    const int* e_src; const int* e_dst; const int* e_pair; const int* tok_e;   // SC_EDGE
    const float* prev_e; const float* T1; const float* ef_tab;
    // TABLE with in == null: rows are the (a,c) token pairs, x = [emb_a[a] | emb_c[c] | temb] (or one-hots when emb_* are null)
    const float* emb_a; const float* emb_c; const float* temb; int ta, tc, tt;
    // SC_EDGE with the pair-slab hoist.  The convolutions that run BEFORE the first EdgeUpdate / NodePositionUpdate (vector_field.py:320-326:
    // convs 0 and 1 of every shipped schedule) see edge features that are identical for the two directed edges of a pair (this layer writes its
    // output to both triangles, self_conditioning.py:78-81) and the same rbf(d), so the K = 160 slab [rbf(d) | ef] x Ws[rbf|ef rows] of their
    // first scalar GEMM is the same vector for both directions.  It is computed HERE, once per unordered pair, while the new edge features are
    // still on chip: the tile's rows [rbf(d(x_t)) | ef] are rebuilt in Hb (ldh >= 164) after the epilogue and multiplied with the slabs of the
    // first one or two convolutions -> Q0 / Q1 (U,256), which the PQ instances of fm_k_edge_message gather like the hoisted Ps[src]:
    // 40,960 of 248,064 executed MAC per edge leave the edge kernel for 20,480 per edge here.  null = off
    const float2* slabW0; float* slabQ0; const float2* slabW1; float* slabQ1;
    // TABLE, several tables in one launch (the embedding tables of a whole chunk of integration steps): workgroup b builds tile
    // b % tab_tiles of table b / tab_tiles, whose time embedding is temb + table * tt and whose rows start at out + table * tab_stride
    int tab_tiles; int tab_stride;
};

// TM = rows per tile: 64 (throughput: every weight fragment serves four row tiles) or 16 (small batches: a tile's two dependent GEMMs are
// matrix-pipe time on ONE CU, so a quarter of the rows is a quarter of the latency, spread over four times as many CUs).
template <int MODE, int TM>
__device__ __forceinline__ void fm_mlp2_tile(const FmMlpArgs& a, int tile, float* lds) {
    float* X = lds;                         // [64][ldx]
    float* Hb = lds + TM * a.ldx;        // [64][ldh]
    int* meta = reinterpret_cast<int*>(Hb + TM * a.ldh);   // [64] token / row ids
    const int tid = threadIdx.x;
    const int row0 = tile * TM;

    // ---------------- prologue: fill X[:, 0..K1p)
    if (MODE == FM_MLP_TABLE) {
        for (int idx = tid; idx < TM * a.K1p; idx += FM_THREADS) {
            const int r = idx / a.K1p, c = idx % a.K1p, row = row0 + r;
            float v = 0.f;
            if (row < a.rows) {
                if (a.in) v = (a.in_w == 0 || c < a.in_w) ? a.in[(size_t)row * a.in_ld + c] : 0.f;
                else {      // input row of the (a,c) embedding table, built in place (vector_field.py:228-243)
                    const int ia = row / a.n_c1, ic = row % a.n_c1;
                    if (c < a.ta) v = a.emb_a ? a.emb_a[ia * a.ta + c] : (c == ia ? 1.f : 0.f);
                    else if (c < a.ta + a.tc) v = a.emb_c ? a.emb_c[ic * a.tc + (c - a.ta)] : ((c - a.ta) == ic ? 1.f : 0.f);
                    else if (c < a.ta + a.tc + a.tt) v = a.temb[c - a.ta - a.tc];
                }
            }
            X[r * a.ldx + c] = v;
        }
    } else if (MODE == FM_MLP_SC_NODE) {
        float* dd = reinterpret_cast<float*>(meta + TM);     // [64] |x_t - x1_prev| per row
        if (tid < TM) {
            const int n = row0 + tid;
            int tok = -1; float d = 0.f;
            if (n < a.rows) {
                tok = a.tok_a[n] * a.n_c1 + a.tok_c[n];
                d = fm_norm3(a.x_t[n * 3 + 0] - a.prev_x[n * 3 + 0], a.x_t[n * 3 + 1] - a.prev_x[n * 3 + 1], a.x_t[n * 3 + 2] - a.prev_x[n * 3 + 2]);
            }
            meta[tid] = tok; dd[tid] = d;
        }
        __syncthreads();
        {   // columns 0..255: the (a,c)-token embedding row, 16-byte loads (rows without a node read 0 via the range check)
            const auto rs = fm_buf(a.s_tab, 0x7fffffffu);
            constexpr int NQ = TM * 64 / FM_THREADS;
            float4 q[NQ];
#pragma unroll
            for (int k = 0; k < NQ; ++k) {
                const int idx = tid + k * FM_THREADS, r = idx >> 6, c4 = idx & 63;
                const int tok = meta[r];
                q[k] = fm_buf_f32x4(rs, tok >= 0 ? tok * 1024 + c4 * 16 : FM_BUF_OOB, 0);
            }
#pragma unroll
            for (int k = 0; k < NQ; ++k) {
                const int idx = tid + k * FM_THREADS, r = idx >> 6, c4 = idx & 63;
                *reinterpret_cast<float4*>(X + r * a.ldx + c4 * 4) = q[k];
            }
        }
        const int kin = a.na + a.nc + 32, kw = a.K1p - 256;     // [p_a | p_c | rbf | zero padding]
        for (int idx = tid; idx < TM * kw; idx += FM_THREADS) {
            const int r = idx / kw, c = idx % kw;
            const int n = row0 + r;
            float v = 0.f;
            if (meta[r] >= 0 && c < kin) {
                if (c < a.na) v = a.prev_a[(size_t)n * a.na + c];
                else if (c < a.na + a.nc) v = a.prev_c[(size_t)n * a.nc + (c - a.na)];
                else v = fm_rbf(dd[r], c - a.na - a.nc, a.rbf_mu_step, a.rbf_inv_sigma);
            }
            X[r * a.ldx + 256 + c] = v;
        }
    } else if (MODE == FM_MLP_NODE_HEAD) {
        const int left = a.rows - row0;
        const auto rs = fm_buf(a.in + (size_t)row0 * 256, (unsigned)(left < TM ? left : TM) * 1024u);
        constexpr int NQ = TM * 64 / FM_THREADS;
        float4 q[NQ];
#pragma unroll
        for (int k = 0; k < NQ; ++k) q[k] = fm_buf_f32x4(rs, (tid + k * FM_THREADS) * 16, 0);
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            const int idx = tid + k * FM_THREADS, r = idx >> 6, c4 = idx & 63;
            *reinterpret_cast<float4*>(X + r * a.ldx + c4 * 4) = q[k];
        }
    } else if (MODE == FM_MLP_EDGE_HEAD) {
        int* pr = meta + 3 * TM;                              // [64] second edge of the pair
        if (tid < TM) {
            const int p = row0 + tid;
            meta[tid] = (p < a.rows) ? a.p_e0[p] : -1;
            pr[tid] = (p < a.rows) ? a.p_e1[p] : -1;
        }
        __syncthreads();
        // x = ef[e0] + ef[e1]: 16-byte loads, all of a thread's requests issued before the first use
        constexpr int NQ = TM * 32 / FM_THREADS;
        float4 u0[NQ], u1[NQ];
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            const int idx = tid + k * FM_THREADS, r = idx >> 5, c4 = idx & 31;
            const int ea = meta[r], eb = pr[r];
            u0[k] = reinterpret_cast<const float4*>(a.ef)[(size_t)(ea < 0 ? 0 : ea) * 32 + c4];
            u1[k] = reinterpret_cast<const float4*>(a.ef)[(size_t)(eb < 0 ? 0 : eb) * 32 + c4];
        }
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            const int idx = tid + k * FM_THREADS, r = idx >> 5, c4 = idx & 31;
            const bool ok = meta[r] >= 0;
            float4 v;
            v.x = ok ? u0[k].x + u1[k].x : 0.f; v.y = ok ? u0[k].y + u1[k].y : 0.f;
            v.z = ok ? u0[k].z + u1[k].z : 0.f; v.w = ok ? u0[k].w + u1[k].w : 0.f;
            *reinterpret_cast<float4*>(X + r * a.ldx + 4 * c4) = v;
        }
    } else {   // FM_MLP_SC_EDGE: per unordered pair (input, token and output are the same for both directions)
        float* dd = reinterpret_cast<float*>(meta + TM);     // [64][2]: d(x_t), d(x1_prev)
        int* pr = meta + 3 * TM;                              // [64] second edge of the pair
        if (tid < TM) {
            const int p = row0 + tid;
            int tok = -1, ea = -1, eb = -1;
            float dt_ = 0.f, d1_ = 0.f;
            if (p < a.rows) {
                ea = a.p_e0[p]; eb = a.p_e1[p];
                const int i = a.e_src[ea], j = a.e_dst[ea];
                tok = a.tok_e[p];
                dt_ = fm_norm3(a.x_t[i * 3] - a.x_t[j * 3], a.x_t[i * 3 + 1] - a.x_t[j * 3 + 1], a.x_t[i * 3 + 2] - a.x_t[j * 3 + 2]) + 1e-8f;
                d1_ = fm_norm3(a.prev_x[i * 3] - a.prev_x[j * 3], a.prev_x[i * 3 + 1] - a.prev_x[j * 3 + 1],
                               a.prev_x[i * 3 + 2] - a.prev_x[j * 3 + 2]) + 1e-8f;
            }
            meta[tid] = tok; pr[tid] = eb; dd[2 * tid] = dt_; dd[2 * tid + 1] = d1_;
            meta[4 * TM + tid] = ea;       // [64] first edge of the pair
        }
        __syncthreads();
        for (int idx = tid; idx < TM * a.K1p; idx += FM_THREADS) {
            const int r = idx / a.K1p, c = idx % a.K1p;
            float v = 0.f;
            if (meta[r] >= 0) {
                if (c < a.ne) v = a.prev_e[(size_t)(row0 + r) * a.ne + c];
                else if (c < a.ne + 32)
                    v = fm_rbf(dd[2 * r + 1], c - a.ne, a.rbf_mu_step, a.rbf_inv_sigma) -
                        fm_rbf(dd[2 * r], c - a.ne, a.rbf_mu_step, a.rbf_inv_sigma);
            }
            X[r * a.ldx + c] = v;
        }
    }
    __syncthreads();

    // ---------------- layer 1 -> Hb
    fm_block_gemm<TM / 16, 1>(X, a.ldx, TM / 16, a.K1p / 8, a.W1, a.H / 16, [&](int row, int col, float v) {
        if (MODE == FM_MLP_SC_EDGE) { const int t = meta[row]; v += (t >= 0) ? a.T1[t * 128 + col] : 0.f; }
        else v += a.b1[col];
        Hb[row * a.ldh + col] = fm_silu(v);
    });
    __syncthreads();
    // ---------------- layer 2 -> X[:, 0..O)
    auto epi2 = [&](int row, int col, float v) {
        v += a.b2[col];
        if (MODE == FM_MLP_TABLE || MODE == FM_MLP_SC_NODE || MODE == FM_MLP_SC_EDGE) v = fm_silu(v);
        X[row * a.ldx + col] = v;
    };
    // the heads have 1-2 output column tiles: single-tile jobs keep 4-8 waves busy instead of 1-2
    if (MODE == FM_MLP_NODE_HEAD || MODE == FM_MLP_EDGE_HEAD) fm_block_gemm<1, 1>(Hb, a.ldh, TM / 16, a.H / 8, a.W2, a.O / 16, epi2);
    else fm_block_gemm<TM / 16, 1>(Hb, a.ldh, TM / 16, a.H / 8, a.W2, a.O / 16, epi2);
    __syncthreads();

    // ---------------- epilogue
    constexpr int LPR = FM_THREADS / TM;            // lanes per row: 8 (64-row tiles) or 32
    const int r = tid / LPR, sub = tid % LPR;
    const int grow = row0 + r;
    if (MODE == FM_MLP_TABLE) {
        float mean, rstd;
        fm_row_stats<LPR>(X + r * a.ldx, a.ln_n, sub, mean, rstd);
        if (grow < a.rows) {
            // rows of a per-pair table (dense edge embedding of endpoint models) go to both directed edges of the pair
            float* o0 = a.out + (size_t)(a.p_e0 ? a.p_e0[grow] : grow) * a.out_ld;
            float* o1 = a.p_e1 ? a.out + (size_t)a.p_e1[grow] * a.out_ld : nullptr;
            for (int c = sub; c < a.O; c += LPR) {
                const float y = fm_fma((X[r * a.ldx + c] - mean) * rstd, a.ln_g[c], a.ln_b[c]);
                o0[c] = y;
                if (o1) o1[c] = y;
            }
        }
    } else if (MODE == FM_MLP_SC_NODE) {
        if (grow < a.rows) {
            const float* trow = a.s_tab + (size_t)meta[r] * 256;
            for (int c = sub * 4; c < 256; c += LPR * 4) {
                const float4 t = *reinterpret_cast<const float4*>(trow + c);
                const float4 x = *reinterpret_cast<const float4*>(X + r * a.ldx + c);
                *reinterpret_cast<float4*>(a.out + (size_t)grow * 256 + c) = make_float4(t.x + x.x, t.y + x.y, t.z + x.z, t.w + x.w);
            }
        }
    } else if (MODE == FM_MLP_SC_EDGE) {
        const bool slab = a.slabQ0 != nullptr;          // uniform
        if (grow < a.rows) {
            const int ea = meta[4 * TM + r], eb = meta[3 * TM + r], tok = meta[r];
            for (int c = sub * 4; c < 128; c += LPR * 4) {
                const float4 t = *reinterpret_cast<const float4*>(a.ef_tab + tok * 128 + c);
                const float4 x = *reinterpret_cast<const float4*>(X + r * a.ldx + c);
                const float4 o = make_float4(t.x + x.x, t.y + x.y, t.z + x.z, t.w + x.w);
                *reinterpret_cast<float4*>(a.out + (size_t)ea * 128 + c) = o;
                *reinterpret_cast<float4*>(a.out + (size_t)eb * 128 + c) = o;
                if (slab) *reinterpret_cast<float4*>(Hb + r * a.ldh + 32 + c) = o;      // the hidden layer is dead: its tile receives [rbf | ef]
            }
        }
        if (slab) {
            // the new edge features are still on chip: the pair-symmetric slab of the first convolutions' scalar linear right here,
            // instead of a kernel of its own that would gather the rows back from HBM
            const float* dd = reinterpret_cast<const float*>(meta + TM);
            for (int idx = tid; idx < TM * 32; idx += FM_THREADS) {
                const int rr = idx >> 5, k = idx & 31;
                Hb[rr * a.ldh + k] = fm_rbf(dd[2 * rr], k, a.rbf_mu_step, a.rbf_inv_sigma);      // rows beyond the batch: finite, never stored
            }
            __syncthreads();
            // Q rows are stored in ACCUMULATOR order: [wave w (8)][column i within a tile (16)][j (2)] = slab column 16 (2w + j) + i -- what lane i of
            // wave w holds for its two column tiles here, and what lane i of wave w needs for ITS two column tiles in fm_k_edge_message's scalar GEMM
            // (same ownership: wave w <-> column tiles 2w, 2w+1).  So a row is written with one 8-byte store per lane and later gathered with one
            // 8-byte load per lane, through tile-relative buffer descriptors (no 64-bit address arithmetic; rows beyond the batch are dropped).
            const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
            const int left = a.rows - row0;
            auto slab_gemm = [&](const float2* __restrict__ W, float* Q) {
                constexpr int MT = TM / 16;
                f32x4 acc[MT][2];
#pragma unroll
                for (int i = 0; i < MT; ++i) { acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                fm_wave_gemm<MT, 2>(acc, Hb, 164, 20, W, 16, 2 * wave, lane);
                const auto rs = fm_buf(Q + (size_t)row0 * 256, (unsigned)(left < TM ? left : TM) * 1024u);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        fm_buf_store_f32x2(rs, (i * 16 + 4 * (lane >> 4) + q) * 1024 + (lane & 15) * 8, wave * 128, acc[i][0][q], acc[i][1][q]);
            };
            slab_gemm(a.slabW0, a.slabQ0);
            if (a.slabQ1) slab_gemm(a.slabW1, a.slabQ1);
        }
    } else {
        // softmax heads (vector_field.py:336-344,364-367): one lane per (row, head)
        if (sub == 0 && grow < a.rows) {
            const float* lg = X + r * a.ldx;
            if (MODE == FM_MLP_NODE_HEAD) {
                float m = lg[0];
                for (int c = 1; c < a.na; ++c) m = fmaxf(m, lg[c]);
                float s = 0.f;
                for (int c = 0; c < a.na; ++c) s += expf(lg[c] - m);
                for (int c = 0; c < a.na; ++c) a.out[(size_t)grow * a.na + c] = expf(lg[c] - m) / s;
                m = lg[a.na];
                for (int c = 1; c < a.nc; ++c) m = fmaxf(m, lg[a.na + c]);
                s = 0.f;
                for (int c = 0; c < a.nc; ++c) s += expf(lg[a.na + c] - m);
                for (int c = 0; c < a.nc; ++c) a.out2[(size_t)grow * a.nc + c] = expf(lg[a.na + c] - m) / s;
            } else {
                float m = lg[0];
                for (int c = 1; c < a.ne; ++c) m = fmaxf(m, lg[c]);
                float s = 0.f;
                for (int c = 0; c < a.ne; ++c) s += expf(lg[c] - m);
                for (int c = 0; c < a.ne; ++c) a.out[(size_t)grow * a.ne + c] = expf(lg[c] - m) / s;
            }
        }
    }
}

template <int MODE, int TM = FM_TM>
__global__ void __launch_bounds__(FM_THREADS) fm_k_mlp2(FmMlpArgs a) {
    HIP_DYNAMIC_SHARED(float, lds)
    int tile = blockIdx.x;
    if (MODE == FM_MLP_TABLE && a.tab_tiles > 0) {
        const int tab = tile / a.tab_tiles;
        tile -= tab * a.tab_tiles;
        a.temb += tab * a.tt;
        a.out += (size_t)tab * a.tab_stride;
    }
    fm_mlp2_tile<MODE, TM>(a, tile, lds);
}

// Two independent MLP passes in ONE launch (node-side tiles first, then the pair-side tiles): the self-conditioning layers
// (SC_NODE + SC_EDGE) and the output heads (NODE_HEAD + EDGE_HEAD) each depend on the same inputs only, so one kernel
// boundary per pair is pure latency on the per-step critical path.
template <int MODE_A, int MODE_B, int TM = FM_TM>
__global__ void __launch_bounds__(FM_THREADS) fm_k_mlp2_pair(FmMlpArgs a, FmMlpArgs b, int tiles_a) {
    HIP_DYNAMIC_SHARED(float, lds)
    if ((int)blockIdx.x < tiles_a) fm_mlp2_tile<MODE_A, TM>(a, blockIdx.x, lds);
    else fm_mlp2_tile<MODE_B, TM>(b, (int)blockIdx.x - tiles_a, lds);
}

// ------------------------------------------------------------------------------------------------
// Node-side MLPs of SMALL batches on 4-row tiles (one molecule .. ~21 molecules of 47 atoms: while (N + 3) / 4 tiles fit one per CU).  A 16-row tile's
// two dependent 256-wide layers are 4 + 3.4 us of matrix time on ONE CU whatever the batch (v_mfma_f32_16x16x4 multiplies 16 rows or none); on four rows
// with v_mfma_f32_4x4x1_16B_f32 a layer is its 0.3 MB weight stream at a CU's 150 GB/s, and a 47-atom molecule spreads over 12 CUs instead of 3 -- the
// treatment round 4 gave fm_k_node_update (fm_wave_gemm4), here for the self-conditioning node layer (self_conditioning.py:49-58), the node output
// head (vector_field.py:336-339) and the first convolution's hoisted projection.  Arithmetic per element AND summation order as in fm_mlp2_tile / fm_k_node_proj
// (round 6: fm_rows4_linear runs the regular tiles' fma chains, one column group of 64 per wave; rounds 4-5 split K over the eight waves -- another order):
// bit-identical to the 16- / 64-row MLP tiles, so the choice follows the batch size in canonical mode too.
// ------------------------------------------------------------------------------------------------
enum FmMlp4Mode { FM_MLP4_SC_NODE = 0, FM_MLP4_NODE_HEAD = 1, FM_MLP4_PROJ0 = 2 };
struct FmMlp4Args {
    int N;
    const void* W1q; const float* b1;          // quad-row packed (fm_wave_gemm4): SC_NODE K = 320 ([s_tab (256) | p_a | p_c | rbf | 0]), NODE_HEAD K = 256; N = 256
    const void* W2q; const float* b2;          // SC_NODE: K = 256, N = 256 (G = 4); NODE_HEAD: K = 256, N = 64 (G = 1; na + nc real columns)
    // SC_NODE
    const float* s_tab; const int* tok_a; const int* tok_c; int n_c1;
    const float* prev_a; const float* prev_c; const float* prev_x; const float* x_t;
    int na, nc; float rbf_mu_step, rbf_inv_sigma;
    float* out; float* out2;                   // SC_NODE: s (N,256); NODE_HEAD: atom-type / charge probabilities
    const float* in;                           // NODE_HEAD / PROJ0: s (N,256)
    // PROJ0: the first convolution's Ps = s * Ws_src; v = 0 (vector_field.py:246), so PV = 0 as well; working copy of the positions
    const void* Wps4; float* Ps; float* PV; int pv_w; float* v_init; int V; const float* x_src; float* x_dst;
};
#define FM_MLP4_LDX 324          // >= 320 columns, 16-byte rows
#define FM_MLP4_LDH 260
#define FM_MLP4_LDS_BYTES ((4 * FM_MLP4_LDX + 4 * FM_MLP4_LDH + 16) * 4)

template <int MODE>
__device__ __forceinline__ void fm_mlp4_tile(const FmMlp4Args& a, int tile, float* lds) {
    constexpr int LDX = FM_MLP4_LDX, LDH = FM_MLP4_LDH;
    float* X = lds;                               // [4][324]
    float* Hb = X + 4 * LDX;                      // [4][260]
    int* meta = reinterpret_cast<int*>(Hb + 4 * LDH); // [4] token row
    float* dd = reinterpret_cast<float*>(meta + 4);
    const int tid = threadIdx.x, row0 = tile * 4;
    const int rows = a.N - row0 < 4 ? a.N - row0 : 4;
    if (MODE == FM_MLP4_SC_NODE) {
        if (tid < 4) {
            const int n = row0 + tid;
            int tok = -1; float d = 0.f;
            if (n < a.N) {
                tok = a.tok_a[n] * a.n_c1 + a.tok_c[n];
                d = fm_norm3(a.x_t[n * 3 + 0] - a.prev_x[n * 3 + 0], a.x_t[n * 3 + 1] - a.prev_x[n * 3 + 1], a.x_t[n * 3 + 2] - a.prev_x[n * 3 + 2]);
            }
            meta[tid] = tok; dd[tid] = d;
        }
        __syncthreads();
        if (tid < 256) {         // columns 0..255: the (a,c)-token embedding row (rows without a node read 0 through the range check)
            const int r = tid >> 6, c4 = tid & 63, tok = meta[r];
            *reinterpret_cast<float4*>(X + r * LDX + c4 * 4) = fm_buf_f32x4(fm_buf(a.s_tab), tok >= 0 ? tok * 1024 + c4 * 16 : FM_BUF_OOB, 0);
        } else {                 // columns 256..319: [p_a | p_c | rbf(|x_t - x1_prev|) | 0]
            const int r = (tid - 256) >> 6, c = (tid - 256) & 63, n = row0 + r;
            float v = 0.f;
            if (meta[r] >= 0 && c < a.na + a.nc + 32) {
                if (c < a.na) v = a.prev_a[(size_t)n * a.na + c];
                else if (c < a.na + a.nc) v = a.prev_c[(size_t)n * a.nc + (c - a.na)];
                else v = fm_rbf(dd[r], c - a.na - a.nc, a.rbf_mu_step, a.rbf_inv_sigma);
            }
            X[r * LDX + 256 + c] = v;
        }
    } else {
        if (tid < 256)           // the tile's rows of s, 16-byte loads; rows beyond N read 0
            *reinterpret_cast<float4*>(X + (tid >> 6) * LDX + (tid & 63) * 4) = fm_buf_f32x4(fm_buf(a.in + (size_t)row0 * 256, (unsigned)rows * 1024u), tid * 16, 0);
        if (MODE == FM_MLP4_PROJ0) {
            for (int i = tid; i < rows * 3; i += FM_THREADS) a.x_dst[row0 * 3 + i] = a.x_src[row0 * 3 + i];
            for (int i = tid; i < rows * 3 * a.V; i += FM_THREADS) a.v_init[(size_t)row0 * 3 * a.V + i] = 0.f;
            for (int i = tid; i < rows * 3 * a.pv_w; i += FM_THREADS) a.PV[(size_t)row0 * 3 * a.pv_w + i] = 0.f;      // v = 0: the hoisted hidden-vector rows are 0 * W
        }
    }
    __syncthreads();
    if (MODE == FM_MLP4_PROJ0) {
        fm_rows4_linear<64, 4>(X, LDX, a.Wps4, [&](int r, int c, float v) { if (r < rows) a.Ps[(size_t)(row0 + r) * 256 + c] = v; });
        return;
    }
    if (MODE == FM_MLP4_SC_NODE)
        fm_rows4_linear<80, 4>(X, LDX, a.W1q, [&](int r, int c, float v) { Hb[r * LDH + c] = fm_silu(v + a.b1[c]); });
    else
        fm_rows4_linear<64, 4>(X, LDX, a.W1q, [&](int r, int c, float v) { Hb[r * LDH + c] = fm_silu(v + a.b1[c]); });
    if (MODE == FM_MLP4_SC_NODE) {
        fm_rows4_linear<64, 4>(Hb, LDH, a.W2q, [&](int r, int c, float v) {
            if (r < rows) a.out[(size_t)(row0 + r) * 256 + c] = a.s_tab[(size_t)meta[r] * 256 + c] + fm_silu(v + a.b2[c]);
        });
    } else {
        const int no = a.na + a.nc;
        fm_rows4_linear<64, 1>(Hb, LDH, a.W2q, [&](int r, int c, float v) { if (c < no) X[r * LDX + c] = v + a.b2[c]; });
        if (tid < rows) {        // softmax heads (vector_field.py:336-339,364-367): one lane per row, the arithmetic of fm_mlp2_tile
            const float* lg = X + tid * LDX;
            const int n = row0 + tid;
            float m = lg[0];
            for (int c = 1; c < a.na; ++c) m = fmaxf(m, lg[c]);
            float sum = 0.f;
            for (int c = 0; c < a.na; ++c) sum += expf(lg[c] - m);
            for (int c = 0; c < a.na; ++c) a.out[(size_t)n * a.na + c] = expf(lg[c] - m) / sum;
            m = lg[a.na];
            for (int c = 1; c < a.nc; ++c) m = fmaxf(m, lg[a.na + c]);
            sum = 0.f;
            for (int c = 0; c < a.nc; ++c) sum += expf(lg[a.na + c] - m);
            for (int c = 0; c < a.nc; ++c) a.out2[(size_t)n * a.nc + c] = expf(lg[a.na + c] - m) / sum;
        }
    }
}

template <int MODE>
__global__ void __launch_bounds__(FM_THREADS) fm_k_mlp4(FmMlp4Args a) {
    HIP_DYNAMIC_SHARED(float, lds)
    fm_mlp4_tile<MODE>(a, blockIdx.x, lds);
}

// 4-row node tiles and 16-row pair tiles of the same stage in ONE launch (the small-batch counterpart of fm_k_mlp2_pair)
template <int MODE4, int MODE_B>
__global__ void __launch_bounds__(FM_THREADS) fm_k_mlp4_pair(FmMlp4Args a, FmMlpArgs b, int tiles_a) {
    HIP_DYNAMIC_SHARED(float, lds)
    if ((int)blockIdx.x < tiles_a) fm_mlp4_tile<MODE4>(a, blockIdx.x, lds);
    else fm_mlp2_tile<MODE_B, 16>(b, (int)blockIdx.x - tiles_a, lds);
}

// gather-only initialisation when there is no self-conditioning input (bootstrap pass / non-SC models)
static __global__ void __launch_bounds__(256) fm_k_gather_rows(float* __restrict__ out, const float* __restrict__ tab, int width,
                                                         int rows, const int* __restrict__ tok_a, const int* __restrict__ tok_c,
                                                         int n_c1, const int* __restrict__ e_pair) {
    // node rows: tok = tok_a*n_c1+tok_c ; edge rows (e_pair != null): tok = tok_a[e_pair[row]]
    const int w4 = width / 4;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < (size_t)rows * w4; idx += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(idx / w4), c4 = (int)(idx % w4);
        const int tok = e_pair ? tok_a[e_pair[row]] : tok_a[row] * n_c1 + tok_c[row];
        reinterpret_cast<float4*>(out)[idx] = reinterpret_cast<const float4*>(tab)[(size_t)tok * w4 + c4];
    }
}

// ------------------------------------------------------------------------------------------------
// per-node projections hoisted out of the per-edge linear layers (SURVEY.md §7 "hard parts"):
//   Ps  = s * Ws_src          (256)   first edge-message GVP, scalar part of s[src]
//   Asd = s * [W1_src|W1_dst] (256)   EdgeUpdate first layer, node parts
//   PV  = v * [Wh[1:] | 0 | Wcp[1:]]  (per xyz: V+16)   first edge-message GVP, vector part of v[src]
// ------------------------------------------------------------------------------------------------
struct FmProjArgs {
    int N;
    const float* s; const float* v;
    const float2* Wps; float* Ps;         // null -> skip
    const float2* Wasd; float* Asd;       // null -> skip
    const float2* Wpv; float* PV;         // null -> skip
    int pv_w;                             // PV row width (FmGvpTile::PVW of the edge-message instance: V+16 without destination features)
    float* v_init;                        // non-null: the vector features start at zero (vector_field.py:246): use zeros AND write them
    const float* x_src; float* x_dst;     // non-null: copy the tile's positions (working copy updated by NodePositionUpdate)
};

// TM = rows per tile: 64, or 16 while even 16-row tiles do not fill the chip (latency of a small batch: see fm_mlp2_tile)
template <int V, int TM = FM_TM>
__global__ void __launch_bounds__(FM_THREADS) fm_k_node_proj(FmProjArgs a) {
    HIP_DYNAMIC_SHARED(float, lds)
    constexpr int LDS_ = 260;                 // 260/4 = 65 odd
    constexpr int LDV = V + 4;
    float* X = lds;                            // [TM][260]
    float* Vt = lds + TM * LDS_;               // [3*TM][V+4]
    const int tid = threadIdx.x, row0 = blockIdx.x * TM;
    for (int idx = tid; idx < TM * 64; idx += FM_THREADS) {
        const int r = idx >> 6, c4 = idx & 63;
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + r < a.N) val = reinterpret_cast<const float4*>(a.s)[(size_t)(row0 + r) * 64 + c4];
        float* d = X + r * LDS_ + 4 * c4;
        *reinterpret_cast<float4*>(d) = val;        // one ds_write_b128 (16-B aligned: row pitch and column offset are multiples of 16 B)
    }
    if (a.x_dst) {
        const int i0 = row0 * 3, i1 = (row0 + TM < a.N ? row0 + TM : a.N) * 3;
        for (int i = i0 + tid; i < i1; i += FM_THREADS) a.x_dst[i] = a.x_src[i];
    }
    if (a.PV) {
        const int rows = a.N - row0 < TM ? a.N - row0 : TM;
        const auto rs_v = fm_buf(a.v + (size_t)row0 * 3 * V, (unsigned)rows * (3 * V * 4));
        constexpr int V4 = V / 4, NCHK = TM * 3 * V4, NV = (NCHK + FM_THREADS - 1) / FM_THREADS;
        float4 qv[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int idx = tid + k * FM_THREADS;
            if (a.v_init) {
                qv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (idx < rows * 3 * V4) reinterpret_cast<float4*>(a.v_init + (size_t)row0 * 3 * V)[idx] = qv[k];
            } else {
                qv[k] = fm_buf_f32x4(rs_v, idx < NCHK ? idx * 16 : FM_BUF_OOB, 0);
            }
        }
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int idx = tid + k * FM_THREADS;
            if (idx < NCHK) {
                const int r = idx / (3 * V4), rem = idx % (3 * V4), c = rem / V4, u4 = rem % V4;
                *reinterpret_cast<float4*>(Vt + (c * TM + r) * LDV + u4 * 4) = qv[k];
            }
        }
    }
    __syncthreads();
    if (a.Ps)
        fm_block_gemm<TM / 16, 2>(X, LDS_, TM / 16, 32, a.Wps, 16, [&](int row, int col, float v) {
            if (row0 + row < a.N) a.Ps[(size_t)(row0 + row) * 256 + col] = v;
        });
    if (a.Asd)
        fm_block_gemm<TM / 16, 2>(X, LDS_, TM / 16, 32, a.Wasd, 16, [&](int row, int col, float v) {
            if (row0 + row < a.N) a.Asd[(size_t)(row0 + row) * 256 + col] = v;
        });
    if (a.PV)
        fm_block_gemm<1, 1>(Vt, LDV, 3 * TM / 16, V / 8, a.Wpv, a.pv_w / 16, [&](int row, int col, float v) {
            const int c = row / TM, r = row % TM;
            if (row0 + r < a.N) a.PV[((size_t)(row0 + r) * 3 + c) * a.pv_w + col] = v;
        });
}

// ------------------------------------------------------------------------------------------------
// use_dst_feats (gvp.py:300-316, 472-473): per node, the projection GVP (V -> VD vectors, S -> S/r scalars, no cross
// products) of the conv's input features, followed at once by the per-node hoists of the first edge GVP's destination terms:
//   Psd = s_dst_msg * Ws[:, dst columns]   (256)        added to the scalar-linear accumulators like Ps[src]
//   PVd = v_dst_msg * [Wh | Wcp][dst rows] (per xyz: PVW) added to the hidden vectors like PV[src]
// The GVP is fm_gvp_core with zero Wcp: its cross products are exactly 0 and their sh entries meet zero weights.
// ------------------------------------------------------------------------------------------------
struct FmDstProjArgs {
    int N;
    const float* s; const float* v;
    FmGvpW g;
    const float2* Wsd; float* Psd;
    const float2* Wpvd; float* PVd; int pv_w;
};

template <int V, int TM, int VD>
__global__ void __launch_bounds__(FM_THREADS) fm_k_dst_proj(FmDstProjArgs a) {
    typedef FmGvpTile<V, TM> T;
    HIP_DYNAMIC_SHARED(float, lds)
    float* X = lds;
    float* Vin = X + T::X_FLOATS;
    float* Vh = Vin + T::VIN_FLOATS;
    float* G = Vh + T::VH_FLOATS;
    const int tid = threadIdx.x, row0 = blockIdx.x * TM;
    {
        const int rows = a.N - row0 < TM ? a.N - row0 : TM;
        const auto rs_s = fm_buf(a.s + (size_t)row0 * 256, (unsigned)rows * 1024u);
        const auto rs_v = fm_buf(a.v + (size_t)row0 * 3 * V, (unsigned)rows * (3 * V * 4));
        constexpr int V4 = V / 4, NCHK = TM * 3 * V4;
        for (int idx = tid; idx < TM * 64; idx += FM_THREADS)
            *reinterpret_cast<float4*>(X + (idx >> 6) * FM_LDX + (idx & 63) * 4) = fm_buf_f32x4(rs_s, idx * 16, 0);
        for (int idx = tid; idx < NCHK; idx += FM_THREADS) {
            const int r = idx / (3 * V4), rem = idx % (3 * V4), c = rem / V4, u4 = rem % V4;
            *reinterpret_cast<float4*>(Vin + (c * TM + r) * T::LDVI + u4 * 4) = fm_buf_f32x4(rs_v, idx * 16, 0);
        }
    }
    __syncthreads();
    {
        FM_MARK_DECL
        float pre[TM / 16][2][4];
        fm_gvp_core<V, VD, false, true, TM, FM_THREADS>(X, Vin, Vh, G, a.g, pre FM_MARK_PASS(50));
    }
    constexpr int MT = TM / 16;
    fm_block_gemm<MT, 2>(X, FM_LDX, MT, 32, a.Wsd, 16, [&](int row, int col, float val) {
        if (row0 + row < a.N) a.Psd[(size_t)(row0 + row) * 256 + col] = val;
    });
    fm_block_gemm<1, 1>(Vin, T::LDVI, 3 * MT, 1, a.Wpvd, a.pv_w / 16, [&](int row, int col, float val) {      // K = 8 >= VD (columns VD.. of Vin are 0)
        const int c = row / TM, r = row % TM;
        if (row0 + r < a.N) a.PVd[((size_t)(row0 + r) * 3 + c) * a.pv_w + col] = val;
    });
}

// ------------------------------------------------------------------------------------------------
// fused GVPConv edge message + aggregation  (reference gvp.py:476-492, 523-543)
//   per directed edge j->i: GVP0([s_j|rbf|ef], [xhat_ji|v_j]) -> GVP1 -> GVP2, summed over j per destination i.
//   Output: per (destination, piece) partial sums; a destination's in-edges are contiguous in the
//   internal order and span at most P tiles, piece = tile - first_tile(dst).  No atomics -> deterministic.
// ------------------------------------------------------------------------------------------------
struct FmMsgArgs {
    FmBatch b;
    const float* x;           // (N,3) positions used for distances
    const float* ef;          // (E,128)
    const float* Ps;          // (N,256)
    const float* PV;          // (N,3,PVW)
    const float* w0;          // (PVW): [Wh[0,:] | 0.. | Wcp[0,:]]  row of the displacement vector
    const float* Psd;         // (N,256)   use_dst_feats: W_s * s_dst_msg[dst], added like Ps[src]        (null otherwise)
    const float* PVd;         // (N,3,PVW) use_dst_feats: hidden-vector contribution of v_dst_msg[dst]    (null otherwise)
    FmGvpW g0, g1, g2;
    float* part_s;            // (N, P, 256)
    float* part_v;            // (N, P, 3, V)
    float rbf_mu_step, rbf_inv_sigma;
    float* dbg_s; float* dbg_v;   // optional: per-edge messages (E,256),(E,3,V) for debugging, else null
    const float* Q;           // (U,256) PQ instances: the pair-symmetric [rbf | ef] slab of GVP0's scalar linear (FmMlpArgs::slabQ0, written by the SC_EDGE kernel
                              // in accumulator order: row[w * 32 + i * 2 + j] = column 16 (2w + j) + i)
    int xcd_chunk;            // > 0: workgroup b handles tile (b % 8) * xcd_chunk + b / 8 (grid = 8 * xcd_chunk); 0: tile = b
};

// SP = 1: opt-in split-precision instance (fm_device.h "bf16x3"): the scalar tile is two bf16 planes (TM * FM_LDP * 4 bytes, 4.6 KB more
// than the f32 tile) and the gate buffer lives inside Vh (dead whenever gates exist), so that two workgroups still share a CU.
// PQ = 1: instance for the convolutions before the first molecule update: the [rbf | ef] slab of GVP0's scalar linear comes from the per-pair
// table Q (written by the self-conditioning edge kernel, FmMlpArgs::slabQ0) like the hoisted Ps[src]; the tile neither loads ef nor evaluates the 32 radial basis functions, and GVP0's scalar
// GEMM shrinks from K = 200 to K = 40 (the hidden-vector norms).
template <int V, int TM, int NTH, int HX, int SP, int PQ = 0>
__global__ void __launch_bounds__(NTH) fm_k_edge_message(FmMsgArgs a) {
    static_assert(!(PQ && (SP || HX)), "the pair-slab instance exists for f32 models without destination features");
    typedef FmGvpTile<V, TM, HX> T;
    HIP_DYNAMIC_SHARED(float, lds)
    float* X = lds;
    constexpr int FMT = SP == 3 ? 1 : 0, NPL = SP == 2 ? 3 : 2;                 // SP: 1 = two bf16 planes, 2 = three bf16 planes, 3 = two half planes (fm_device.h)
    float* Vin = X + (SP ? TM * FM_LDP * NPL / 2 : T::X_FLOATS);
    float* Vh = Vin + T::VIN_FLOATS;
    float* G = SP ? Vh + TM * FM_LDG : Vh + T::VH_FLOATS;
    int* m_src = reinterpret_cast<int*>(SP ? Vh + T::VH_FLOATS : G + T::G_FLOATS);   // [64]
    int* m_dst = m_src + TM;                              // [64]
    float* m_geo = reinterpret_cast<float*>(m_dst + TM);  // [TM][4]: xhat(3), dist
    int* m_piece = reinterpret_cast<int*>(m_geo + 4 * TM); // [TM] which partial-sum slot of its destination the row adds to
    int* m_soff = m_piece + TM;                            // [TM] byte offset of the source's row in the (N,256) tables, FM_BUF_OOB if none
    int* m_doff = m_soff + TM;                             // [TM] same for the destination (use_dst_feats)
    const int tid = threadIdx.x;
    // Workgroups are dealt round-robin to the 8 XCDs (each with its own L2).  With the chunked mapping every XCD walks
    // one contiguous range of tiles, so a molecule's Ps / PV rows (shared by its ~n^2/TM consecutive tiles) are filled
    // into one L2 instead of all eight.
    const int tile = a.xcd_chunk > 0 ? (int)(blockIdx.x & 7) * a.xcd_chunk + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (tile >= a.b.n_tiles) return;  // padding workgroups of the chunked grid (uniform exit, before any barrier)
    const int4 td = a.b.tile_desc[tile];      // wave-uniform: one scalar load
    const int e0 = td.x, left = td.y, moff = td.z;      // first row, rows that exist, the molecule's first row (e0 - moff is a multiple of TM)
    FM_MARK_DECL
    // (A) the edge-feature rows depend only on the tile index: request them before anything else (HBM latency)
    constexpr int NEF = PQ ? 1 : TM * 32 / NTH;
    float4 efv[NEF];
    if constexpr (!PQ) {
        const auto rs = fm_buf(a.ef + (size_t)e0 * 128, (unsigned)left * 512u);      // ragged last tile of a molecule: the range check zero-fills
#pragma unroll
        for (int k = 0; k < NEF; ++k) efv[k] = (FM_ABLATE & 16) ? make_float4(0.1f, 0.2f, 0.3f, 0.4f) : fm_buf_f32x4(rs, tid * 16 + k * NTH * 16, 0);
    }
    // (B) endpoints and geometry of the tile's edges
    if (PQ && tid < 64) {
        // PQ: byte offsets of the rows' pairs in Q, relative to the tile's smallest pair id (a wave-wide min: the tile's pairs lie within a few
        // molecules, so the relative offsets stay far below the 31-bit limit of a buffer offset however large the batch's Q table is)
        const int e = e0 + tid;
        const int pr = tid < left ? a.b.e_pair[e] : 0x7fffffff;
        int mn = pr;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_xor(mn, o); mn = t < mn ? t : mn; }
        if (tid < TM) m_doff[tid] = pr != 0x7fffffff ? (pr - mn) * 1024 : FM_BUF_OOB;
        if (tid == 0) m_doff[TM] = mn;
    }
    if (tid < TM) {
        const int e = e0 + tid;
        int s = -1, d = -1, piece = 0;
        float gx = 0.f, gy = 0.f, gz = 0.f, dist = 0.f;
        if (tid < left) {
            s = a.b.e_src[e]; d = a.b.e_dst[e];
            piece = (e - moff) / FM_CHUNK_E - (a.b.node_first_edge[d] - moff) / FM_CHUNK_E;      // chunk of the row minus the first chunk of its destination, both counted from the molecule's first row
            // x_diff = x[src] - x[dst]; d = sqrt(max(|.|^2,1e-8)) + 1e-8; xhat = x_diff / d  (vector_field.py:381-383)
            const float dx = a.x[s * 3] - a.x[d * 3], dy = a.x[s * 3 + 1] - a.x[d * 3 + 1], dz = a.x[s * 3 + 2] - a.x[d * 3 + 2];
            dist = fm_norm3(dx, dy, dz) + 1e-8f;
            const float inv_d = __builtin_amdgcn_rcpf(dist);       // one v_rcp_f32 instead of three IEEE divisions on the tile's start-up path
            gx = dx * inv_d; gy = dy * inv_d; gz = dz * inv_d;
        }
        m_src[tid] = s; m_dst[tid] = d; m_piece[tid] = piece;
        m_soff[tid] = s >= 0 ? s * 1024 : FM_BUF_OOB;
        if (!PQ) m_doff[tid] = d >= 0 ? d * 1024 : FM_BUF_OOB;
        m_geo[4 * tid] = gx; m_geo[4 * tid + 1] = gy; m_geo[4 * tid + 2] = gz; m_geo[4 * tid + 3] = dist;
    }
    __syncthreads();
    FM_MARK(0);
    // (C) the hoisted per-source scalar term of GVP0 is only consumed after its scalar GEMM: request it now so that its
    //     L2 latency overlaps the fill below and the first phases of the GVP (VMEM returns in order, so a request
    //     placed right before the GEMM would stall the GEMM's first weight fragments behind it)
    float pre[TM / 16][1024 / NTH][4];
    if (!(FM_ABLATE & 16)) fm_gather_pre<TM, NTH, false>(pre, a.Ps, a.b.N, m_soff);
    else { for (auto& p1 : pre) for (auto& p2 : p1) for (auto& p3 : p2) p3 = 0.25f; }
    if (HX > 0) fm_gather_pre<TM, NTH, true>(pre, a.Psd, a.b.N, m_doff);      // + the destination node's hoisted scalar term
    float preq[PQ ? TM / 16 : 1][PQ ? 1024 / NTH : 1][4];
    if constexpr (PQ) {      // the pair's slab row, requested right behind Ps[src]; summed into `pre` after the fill below (VMEM returns in order:
                             // once the fill's own gathers have arrived these have too, so the sum waits for nothing)
        static_assert(!PQ || NTH == 512, "Q rows are laid out for 8 waves x 2 column tiles");
        const int pmin = __builtin_amdgcn_readfirstlane(m_doff[TM]);
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const auto rs = fm_buf(a.Q + (size_t)pmin * 256, 0x7ffffc00u);
#pragma unroll
        for (int i = 0; i < TM / 16; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {       // one 8-byte load per row: the row's values for this lane's two column tiles sit side by side (accumulator order)
                const float2 t = fm_buf_f32x2(rs, m_doff[i * 16 + 4 * (lane >> 4) + r] + (lane & 15) * 8, wave * 128);
                preq[i][0][r] = t.x; preq[i][1][r] = t.y;
            }
    }
    // (D) X[:, 0..31] = rbf(d), X[:, 32..159] = ef; hidden vectors of GVP0: Vh[c*TM+r][:] = PV[src][c][:] + xhat[r][c]*w0[:]
    //     All gathers of a thread are issued back to back (unconditional loads from a clamped index, select afterwards):
    //     a branchy load-use-store loop serialises one L2 round trip per iteration (profiles/r01d: 37k cycles here).
    {
        // thread -> (row chunk q, column j of a 16-wide block); chunk = (xyz c, block cb, row r).  With TM*16 == NTH the
        // chunk's (c, cb) is the unrolled loop index and r = q, so every address is one VGPR + an immediate.
        constexpr int PVW = T::PVW, CB = PVW / 16, NCH = 3 * CB * TM, QN = NTH / 16, NP = (NCH + QN - 1) / QN;
        const int q = tid >> 4, j = tid & 15;
        const auto rs = fm_buf(a.PV, (unsigned)a.b.N * (unsigned)(3 * PVW * 4));
        const auto rsd = fm_buf(HX > 0 ? a.PVd : a.PV, (unsigned)a.b.N * (unsigned)(3 * PVW * 4));
        float pv[NP], w0v[NP];
#pragma unroll
        for (int p_ = 0; p_ < NP; ++p_) {
            const int ch = q + p_ * QN, r = ch % TM, ccb = ch / TM, c = ccb / CB, cb = ccb % CB;
            const int sidx = (NCH % QN == 0 || ch < NCH) ? m_src[r] : -1;
            pv[p_] = (FM_ABLATE & 16) ? 0.5f : fm_buf_f32(rs, sidx >= 0 ? sidx * (3 * PVW * 4) + j * 4 : FM_BUF_OOB, (c * PVW + cb * 16) * 4);
            if (HX > 0) {        // v_dst_msg[dst] * Wh / Wcp rows of the destination vectors, hoisted per node like PV
                const int didx = (NCH % QN == 0 || ch < NCH) ? m_dst[r] : -1;
                pv[p_] += fm_buf_f32(rsd, didx >= 0 ? didx * (3 * PVW * 4) + j * 4 : FM_BUF_OOB, (c * PVW + cb * 16) * 4);
            }
            w0v[p_] = a.w0[cb * 16 + j];
        }
#pragma unroll
        for (int p_ = 0; p_ < NP; ++p_) {
            const int ch = q + p_ * QN, r = ch % TM, ccb = ch / TM, c = ccb / CB, cb = ccb % CB;
            if (NCH % QN == 0 || ch < NCH)
                Vh[(c * TM + r) * T::LDVH + cb * 16 + j] = fm_fma(m_geo[4 * r + c], w0v[p_], pv[p_]);      // rows without an edge: 0 + 0 * w0 (range-checked gather, zero geometry)
        }
    }
    if (SP) {
        unsigned short* XH = reinterpret_cast<unsigned short*>(X);
        unsigned short* XL = XH + TM * FM_LDP;
        // K padding of the two scalar-GEMM layouts ([rbf | ef | sh] -> 7 k32 blocks, [s | sh] -> 10): written once, never touched again
        constexpr int Z0 = 160 + T::KU0, Z1 = 256 + T::KU;
        for (int idx = tid; idx < TM * 24; idx += NTH) {
            const int r = idx / 24, c = idx % 24;
            if (Z0 + c < (Z0 + 31) / 32 * 32) { XH[r * FM_LDP + Z0 + c] = 0; XL[r * FM_LDP + Z0 + c] = 0; if (SP == 2) XL[TM * FM_LDP + r * FM_LDP + Z0 + c] = 0; }
            if (Z1 + c < 320) { XH[r * FM_LDP + Z1 + c] = 0; XL[r * FM_LDP + Z1 + c] = 0; if (SP == 2) XL[TM * FM_LDP + r * FM_LDP + Z1 + c] = 0; }
        }
        for (int idx = tid; idx < TM * 32; idx += NTH) {
            const int r = idx >> 5, k = idx & 31;
            if (SP == 2) fm_split3_store(XH, XL, r, k, fm_rbf(m_geo[4 * r + 3], k, a.rbf_mu_step, a.rbf_inv_sigma));
            else fm_split_store<FM_LDP, FMT>(XH, XL, r, k, fm_rbf(m_geo[4 * r + 3], k, a.rbf_mu_step, a.rbf_inv_sigma));
        }
#pragma unroll
        for (int k = 0; k < NEF; ++k) {
            const int idx = tid + k * NTH, r = idx >> 5, c4 = idx & 31;
            if (SP == 2) {
                fm_split3_store(XH, XL, r, 32 + 4 * c4 + 0, efv[k].x); fm_split3_store(XH, XL, r, 32 + 4 * c4 + 1, efv[k].y);
                fm_split3_store(XH, XL, r, 32 + 4 * c4 + 2, efv[k].z); fm_split3_store(XH, XL, r, 32 + 4 * c4 + 3, efv[k].w);
            } else {
                fm_split_store<FM_LDP, FMT>(XH, XL, r, 32 + 4 * c4 + 0, efv[k].x); fm_split_store<FM_LDP, FMT>(XH, XL, r, 32 + 4 * c4 + 1, efv[k].y);
                fm_split_store<FM_LDP, FMT>(XH, XL, r, 32 + 4 * c4 + 2, efv[k].z); fm_split_store<FM_LDP, FMT>(XH, XL, r, 32 + 4 * c4 + 3, efv[k].w);
            }
        }
    } else if constexpr (!PQ) {
    for (int idx = tid; idx < TM * 32; idx += NTH) {
        const int r = idx >> 5, k = idx & 31;
        X[r * FM_LDX + k] = fm_rbf(m_geo[4 * r + 3], k, a.rbf_mu_step, a.rbf_inv_sigma);      // rows without an edge hold finite junk that is never aggregated
    }
#pragma unroll
    for (int k = 0; k < NEF; ++k) {
        const int idx = tid + k * NTH, r = idx >> 5, c4 = idx & 31;
        *reinterpret_cast<float4*>(X + r * FM_LDX + 32 + 4 * c4) = efv[k];   // ds_write_b128 (16-B aligned)
    }
    }
    if constexpr (PQ) {
#pragma unroll
        for (int i = 0; i < TM / 16; ++i)
#pragma unroll
            for (int j = 0; j < 1024 / NTH; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) pre[i][j][r] += preq[i][j][r];
    }
    __syncthreads();
    FM_MARK(1);
    fm_gvp_core<V, V, true, true, TM, NTH, HX, SP, false, PQ != 0>(X, Vin, Vh, G, a.g0, pre FM_MARK_PASS(10));
    fm_gvp_core<V, V, false, true, TM, NTH, HX, SP, false>(X, Vin, Vh, G, a.g1, pre FM_MARK_PASS(20));
    fm_gvp_core<V, V, false, true, TM, NTH, HX, SP, true>(X, Vin, Vh, G, a.g2, pre FM_MARK_PASS(30));

    if (a.dbg_s) {
        for (int idx = tid; idx < TM * 256; idx += NTH) {
            const int r = idx >> 8, c = idx & 255;
            if (m_src[r] >= 0) a.dbg_s[(size_t)(e0 + r) * 256 + c] = X[r * FM_LDX + c];
        }
        for (int idx = tid; idx < 3 * TM * V; idx += NTH) {
            const int row = idx / V, u = idx % V, c = row / TM, r = row % TM;
            if (m_src[r] >= 0) a.dbg_v[((size_t)(e0 + r) * 3 + c) * V + u] = Vin[row * T::LDVI + u];
        }
    }
    FM_MARK(40);
    // segmented sum over the rows of each destination (rows are dst-sorted), cut at the molecule-relative chunk boundaries (FM_CHUNK_E: tile rows 15, 31, 47 --
    // tiles start at multiples of TM from the molecule's first row).  One thread per output column; the column's TM
    // values are pulled into registers with independent LDS reads and summed there (the first version walked the rows with
    // dependent LDS reads: 15k cycles).  Where a destination's rows end inside the tile is ONE wave-uniform bit mask (a ballot of
    // dst[r] != dst[r+1]), so the segment logic is a scalar bit test per row; destination id and piece are fetched (LDS
    // broadcast + v_readfirstlane) only at the 2-5 segment ends of a tile.  Round 1 made all 2*TM ids wave-uniform with
    // v_readlane: 64 VALU per wave and tile, each of which costs matrix-pipe time (profiles/r02a ablation).  Rows past the end
    // of the edge list (ragged last tile) come after the last set bit, so whatever they add to `run` is never stored.
    static_assert(NTH >= 512 && 3 * V <= 128 && TM <= 64, "waves 0..3 own the 256 scalar columns, waves 4..5 the 3*V vector columns (the others idle here; 1024-thread instances were measured in round 6: profiles/r06d_*)");
    if (!(FM_ABLATE & 8)) {
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
        const int lr = lane < TM ? lane : TM - 1;
        const int myd = m_dst[lr];
        const int nxd = m_dst[lr + 1 < TM ? lr + 1 : TM - 1];
        const unsigned long long ends = __ballot(lane < TM && myd >= 0 && (lr + 1 == TM || nxd != myd || (lr & (FM_CHUNK_E - 1)) == FM_CHUNK_E - 1));      // a destination's rows end, or a chunk does
        const bool is_s = wave < 4;                                       // columns 0..255: scalars
        const int cv = tid - 256;                                         // vector column = xyz*V + channel
        // one instance per tile kind, so that the row stride is a compile-time constant and the TM LDS reads of a column are
        // one base register + immediates (a run-time stride costs one VALU address computation per row)
        auto column_sums = [&](const float* vp, auto stride_c, float* out_base, unsigned row_bytes, int col) {
            constexpr int STRIDE = decltype(stride_c)::value;
            float val[TM];
#pragma unroll
            for (int r = 0; r < TM; ++r) val[r] = vp[r * STRIDE];
            float run = 0.f;
#pragma unroll
            for (int r = 0; r < TM; ++r) {
                run += val[r];
                if ((ends >> r) & 1ull) {
                    const int d = __builtin_amdgcn_readfirstlane(m_dst[r]), pcr = __builtin_amdgcn_readfirstlane(m_piece[r]);
                    const size_t slot = (size_t)d * a.b.P + pcr;
                    fm_buf_store_f32(fm_buf(reinterpret_cast<const char*>(out_base) + slot * row_bytes, row_bytes), col * 4, 0, run);
                    run = 0.f;
                }
            }
        };
        if (is_s) column_sums(X + tid, std::integral_constant<int, FM_LDX>{}, a.part_s, 1024u, tid);
        else if (cv < 3 * V) column_sums(Vin + (cv / V) * TM * T::LDVI + (cv % V), std::integral_constant<int, T::LDVI>{}, a.part_v, 3 * V * 4u, cv);
    }
    FM_MARK(41);
}

// ------------------------------------------------------------------------------------------------
// node update of a GVPConv (gvp.py:494-519): sum pieces, /z, residual + GVPLayerNorm, 3 node GVPs,
// residual + GVPLayerNorm.
// ------------------------------------------------------------------------------------------------
struct FmNodeUpdArgs {
    FmBatch b;
    float* s; float* v;              // (N,256), (N,3,V) updated in place
    const float* part_s; const float* part_v;
    float inv_z;                     // 1 / z; < 0: divide by the node's in-degree (message_norm 'mean')
    FmGvpW g0, g1, g2;
    const float* ln1_g; const float* ln1_b; const float* ln2_g; const float* ln2_b;
    float* agg_s; float* agg_v;      // optional debug taps of the aggregated messages, else null
    // Fused tail, every part optional (null = skip).  The updated (s, v) tile is still in LDS, so everything that depends only
    // on it runs here instead of in launches of its own that would re-read it: the hoisted per-node projections of the NEXT
    // conv's edge messages (Ps, PV) and of this conv's EdgeUpdate (Asd), and NodePositionUpdate (vector_field.py:813-842).
    const float2* Wps; float* Ps;
    const float2* Wpv; float* PV;
    const float2* Wasd; float* Asd;
    FmGvpW p0, p1, p2; float* x;     // x != null: x += GVP3(GVP2(GVP1(s, v))).v[:, 0]
    const void* Wps_sp; const void* Wasd_sp;      // split-precision instance: Wps / Wasd as bf16 hi/lo planes
    const void* Wps4; const void* Wasd4;          // RG instances: Wps / Wasd quad-row packed (fm_wave_gemm4)
    int s_real;                      // NARROW instances: real scalar width (< 256); the LayerNorm statistics run over it
};

// GVPLayerNorm of a tile held in X[:, 0..255] / Vin (gvp.py:169-184); result written to LDS in place and,
// when out_s/out_v are given, to HBM.
// SP = 1 (split precision): the normalised scalars leave as the bf16 hi/lo planes the following GVPs / projections read (the planes
// overlay the f32 tile, so every thread first finishes reading its row -- values in registers, one barrier -- then writes).
template <int V, int TM, int SP = 0>
__device__ __forceinline__ void fm_gvp_layernorm_tile(float* X, float* Vin, const float* g, const float* b_,
                                                      int row0, int nrows, float* out_s, float* out_v, int s_width = 256, int rows_valid = TM) {
    typedef FmGvpTile<V, TM> T;
    constexpr int LPR = FM_THREADS / TM;          // lanes per row
    const int tid = threadIdx.x, r = tid / LPR, sub = tid % LPR;
    float mean, rstd;
    fm_row_stats<LPR>(X + r * FM_LDX, s_width, sub, mean, rstd);      // zero-padded columns beyond s_width stay 0: gain and bias are padded with 0
    // vector norm: vn = sqrt(mean_c max(|v_c|^2, 1e-8) + eps) + eps
    float q = 0.f;
    for (int u = fm_ln_sub<LPR>(sub); u < V; u += FmLnLanes<LPR>::value) {      // canonical order (fm_row_stats): 16 lanes per row whatever the tile height
        const float vx = Vin[(0 * TM + r) * T::LDVI + u], vy = Vin[(1 * TM + r) * T::LDVI + u], vz = Vin[(2 * TM + r) * T::LDVI + u];
        q += fmaxf(fm_fma(vz, vz, fm_fma(vy, vy, vx * vx)), 1e-8f);
    }
    q = fm_group_sum<FmLnLanes<LPR>::value>(q);
    const float vn = __builtin_amdgcn_sqrtf(fm_fma(q, 1.0f / (float)V, 1e-5f)) + 1e-5f;
    const float inv_vn = __builtin_amdgcn_rcpf(vn);          // hardware sqrt / rcp (~1 ulp each)
    const bool valid = r < rows_valid && row0 + r < nrows;       // rows_valid < TM: the RG instances' tiles hold 4 RG nodes in a TM-row frame
    if constexpr (SP) {
        float ys[256 / LPR];
#pragma unroll
        for (int k = 0; k < 256 / LPR; ++k) {
            const int c = sub + k * LPR;
            ys[k] = fm_fma((X[r * FM_LDX + c] - mean) * rstd, g[c], b_[c]);
            if (out_s && valid) out_s[(size_t)(row0 + r) * 256 + c] = ys[k];
        }
        __syncthreads();
        unsigned short* XH = reinterpret_cast<unsigned short*>(X);
        unsigned short* XL = XH + TM * FM_LDP;
#pragma unroll
        for (int k = 0; k < 256 / LPR; ++k) fm_split_store<FM_LDP, (SP == 3 ? 1 : 0)>(XH, XL, r, sub + k * LPR, ys[k]);
        for (int c = 256 + V + 8 + sub; c < 320; c += LPR) { XH[r * FM_LDP + c] = 0; XL[r * FM_LDP + c] = 0; }      // K padding of the [s | sh] layout
    } else {
    for (int c = sub; c < 256; c += LPR) {
        const float y = fm_fma((X[r * FM_LDX + c] - mean) * rstd, g[c], b_[c]);
        X[r * FM_LDX + c] = y;
        if (out_s && valid) out_s[(size_t)(row0 + r) * 256 + c] = y;
    }
    }
    for (int u = sub; u < V; u += LPR)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float y = Vin[(c * TM + r) * T::LDVI + u] * inv_vn;
            Vin[(c * TM + r) * T::LDVI + u] = y;
            if (out_v && valid) out_v[((size_t)(row0 + r) * 3 + c) * V + u] = y;
        }
    __syncthreads();
}

// SP = 1: opt-in split precision (fm_device.h "bf16x3"): the scalar / gate GEMMs of its six GVPs and the two 256 x 256 projections on the
// bf16 matrix cores; LayerNorm, residuals, the vector path and the hidden-vector projection stay f32.
// RG > 0 (small batches): a workgroup owns 4 RG nodes in a TM-row frame (RG = 1, 2, 3 in the 16-row frame, 5 in the 32-row frame), chosen so
// that the tiles fit one per CU: a 47-atom molecule spreads over 12 CUs instead of 3, and 256 x 18 atoms (C2) over 231 CUs with one tile each
// instead of 288 sixteen-row tiles on 256 CUs.  The scalar GEMMs of its six GVPs and the two 256 x 256 projections, whose cost scales with the
// tile height, run on the real rows with v_mfma_f32_4x4x1_16B_f32 (fm_wave_gemm4: every 1-KB weight fragment is loaded once and multiplied with
// RG row groups); the vector-side GEMMs, LayerNorms and gates are the TM-row code over the frame (the other rows carry zeros / finite junk
// that is never stored).
template <int V, int TM, bool NARROW, int SP, int RG = 0>
__global__ void __launch_bounds__(FM_THREADS) fm_k_node_update(FmNodeUpdArgs a) {
    static_assert(RG == 0 || (4 * RG <= TM && !NARROW && !SP), "the 4 RG-node instances exist for f32 full-width models");
    typedef FmGvpTile<V, TM> T;
    constexpr int RV = RG ? 4 * RG : TM;                           // rows of the frame that are nodes
    HIP_DYNAMIC_SHARED(float, lds)
    float* X = lds;
    float* Vin = X + (SP ? TM * FM_LDP : T::X_FLOATS);
    float* Vh = Vin + T::VIN_FLOATS;
    float* G = SP ? Vh + TM * FM_LDG : Vh + T::VH_FLOATS;
    const int tid = threadIdx.x, row0 = blockIdx.x * RV;
    const int N = a.b.N, P = a.b.P;
    const int rows = N - row0 < RV ? N - row0 : RV;                // rows of this tile that exist
    // Number of partial-sum pieces of each row, once per row (G is free until the first gate GEMM).  Everything after
    // it is 16-byte buffer loads through tile-relative descriptors, all of a thread's requests issued back to back:
    // rows beyond N and pieces beyond a row's count read 0 through the range check (a per-element version of this
    // prologue with its dependent index loads was the latency of the whole kernel at small batch sizes).
    int* r_np = reinterpret_cast<int*>(G);
    float* r_iz = G + TM;            // 1 / z per row: a.inv_z, or 1 / in-degree for message_norm 'mean' (a.inv_z < 0)
    if (tid < TM) {
        const int n = row0 + tid;
        int np = 0;
        float iz = a.inv_z;
        if (n < N && tid < RV) {
            const int m = a.b.node_mol[n];
            const int deg = a.b.mol_node_off[m + 1] - a.b.mol_node_off[m] - 1;
            if (deg > 0) {
                const int fe = a.b.node_first_edge[n] - a.b.mol_edge_off[m];       // counted from the molecule's first row, like the chunks
                np = (fe + deg - 1) / FM_CHUNK_E - fe / FM_CHUNK_E + 1;
            }
            if (iz < 0.f) iz = deg > 0 ? 1.0f / (float)deg : 0.f;
        }
        r_np[tid] = np; r_iz[tid] = iz;
    }
    __syncthreads();
    const auto rs_s = fm_buf(a.s + (size_t)row0 * 256, (unsigned)rows * 1024u);
    const auto rs_v = fm_buf(a.v + (size_t)row0 * 3 * V, (unsigned)rows * (3 * V * 4));
    {
        // s + sum(pieces)/z
        const auto rs_p = fm_buf(a.part_s + (size_t)row0 * P * 256, (unsigned)rows * (unsigned)P * 1024u);
        constexpr int NQ = TM * 64 / FM_THREADS;
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            const int idx = tid + k * FM_THREADS, r = idx >> 6, c4 = idx & 63;
            const int np = r_np[r];
            const float4 sv = fm_buf_f32x4(rs_s, (r * 256 + c4 * 4) * 4, 0);
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int p0 = 0; p0 < P; p0 += 4) {
                float4 q[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) q[j] = fm_buf_f32x4(rs_p, p0 + j < np ? ((r * P + p0 + j) * 256 + c4 * 4) * 4 : FM_BUF_OOB, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) { acc.x += q[j].x; acc.y += q[j].y; acc.z += q[j].z; acc.w += q[j].w; }
            }
            const float iz = r_iz[r];
            acc.x *= iz; acc.y *= iz; acc.z *= iz; acc.w *= iz;      // the aggregated message M / z (tapped), then the residual: two roundings, as the reference's ops
            if (a.agg_s && r < rows) *reinterpret_cast<float4*>(a.agg_s + (size_t)(row0 + r) * 256 + c4 * 4) = acc;
            *reinterpret_cast<float4*>(X + r * FM_LDX + c4 * 4) = make_float4(sv.x + acc.x, sv.y + acc.y, sv.z + acc.z, sv.w + acc.w);
        }
    }
    {
        const auto rs_p = fm_buf(a.part_v + (size_t)row0 * P * 3 * V, (unsigned)rows * (unsigned)P * (3 * V * 4));
        constexpr int V4 = V / 4, NCHK = TM * 3 * V4;
        for (int idx = tid; idx < NCHK; idx += FM_THREADS) {
            const int r = idx / (3 * V4), rem = idx % (3 * V4), c = rem / V4, u4 = rem % V4;
            const int np = r_np[r];
            const float4 vv = fm_buf_f32x4(rs_v, ((r * 3 + c) * V + u4 * 4) * 4, 0);
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int p0 = 0; p0 < P; p0 += 4) {
                float4 q[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) q[j] = fm_buf_f32x4(rs_p, p0 + j < np ? (((r * P + p0 + j) * 3 + c) * V + u4 * 4) * 4 : FM_BUF_OOB, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) { acc.x += q[j].x; acc.y += q[j].y; acc.z += q[j].z; acc.w += q[j].w; }
            }
            const float iz = r_iz[r];
            acc.x *= iz; acc.y *= iz; acc.z *= iz; acc.w *= iz;
            if (a.agg_v && r < rows) *reinterpret_cast<float4*>(a.agg_v + ((size_t)(row0 + r) * 3 + c) * V + u4 * 4) = acc;
            *reinterpret_cast<float4*>(Vin + (c * TM + r) * T::LDVI + u4 * 4) = make_float4(vv.x + acc.x, vv.y + acc.y, vv.z + acc.z, vv.w + acc.w);
        }
    }
    __syncthreads();
    const int s_width = NARROW ? a.s_real : 256;
    fm_gvp_layernorm_tile<V, TM, SP>(X, Vin, a.ln1_g, a.ln1_b, row0, N, a.s, a.v, s_width, RV);     // s1, v1 -> HBM (needed for the residual)
    {
        FM_MARK_DECL
        float pre[TM / 16][2][4];
        fm_gvp_core<V, V, false, true, TM, FM_THREADS, 0, SP, false, false, RG>(X, Vin, Vh, G, a.g0, pre FM_MARK_PASS(50));
        fm_gvp_core<V, V, false, true, TM, FM_THREADS, 0, SP, false, false, RG>(X, Vin, Vh, G, a.g1, pre FM_MARK_PASS(50));
        fm_gvp_core<V, V, false, true, TM, FM_THREADS, 0, SP, true, false, RG>(X, Vin, Vh, G, a.g2, pre FM_MARK_PASS(50));      // SP: leaves the f32 tile for the residual
    }
    {   // residual: + (s1, v1), re-read from HBM with 16-byte loads (rows beyond N read 0)
        constexpr int NQ = TM * 64 / FM_THREADS;
        float4 sv[NQ];
#pragma unroll
        for (int k = 0; k < NQ; ++k) { const int idx = tid + k * FM_THREADS; sv[k] = fm_buf_f32x4(rs_s, ((idx >> 6) * 256 + (idx & 63) * 4) * 4, 0); }
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            const int idx = tid + k * FM_THREADS;
            float4* xp = reinterpret_cast<float4*>(X + (idx >> 6) * FM_LDX + (idx & 63) * 4);
            float4 x = *xp;
            x.x += sv[k].x; x.y += sv[k].y; x.z += sv[k].z; x.w += sv[k].w;
            *xp = x;
        }
        constexpr int V4 = V / 4, NCHK = TM * 3 * V4;
        for (int idx = tid; idx < NCHK; idx += FM_THREADS) {
            const int r = idx / (3 * V4), rem = idx % (3 * V4), c = rem / V4, u4 = rem % V4;
            const float4 vv = fm_buf_f32x4(rs_v, ((r * 3 + c) * V + u4 * 4) * 4, 0);
            float4* vp = reinterpret_cast<float4*>(Vin + (c * TM + r) * T::LDVI + u4 * 4);
            float4 x = *vp;
            x.x += vv.x; x.y += vv.y; x.z += vv.z; x.w += vv.w;
            *vp = x;
        }
    }
    __syncthreads();
    fm_gvp_layernorm_tile<V, TM, SP>(X, Vin, a.ln2_g, a.ln2_b, row0, N, a.s, a.v, s_width, RV);       // ends with a barrier: X = s (planes if SP), Vin = v
    // ---- fused tail: projections first (they only read the tile), then the position GVPs (which overwrite it)
    constexpr int MT = TM / 16;
    if constexpr (SP) {
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const unsigned short* XH = reinterpret_cast<const unsigned short*>(X);
        const unsigned short* XL = XH + TM * FM_LDP;
        auto project = [&](const void* wsp, float* out) {       // (TM x 256) x (256 x 256): wave w owns column tiles 2w, 2w+1 for all row tiles
            f32x4 acc[MT][2];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            fm_wave_gemm_sp<MT, 2, FM_LDP, (SP == 3 ? 1 : 0)>(acc, XH, XL, 0, 8, wsp, 16, 2 * wave, lane);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = i * 16 + 4 * (lane >> 4) + r;
                        if (row0 + row < N) out[(size_t)(row0 + row) * 256 + (2 * wave + j) * 16 + (lane & 15)] = acc[i][j][r] * (1.0f / fm_sp_wscale<(SP == 3 ? 1 : 0)>());
                    }
        };
        if (a.Ps) project(a.Wps_sp, a.Ps);
        if (a.Asd) project(a.Wasd_sp, a.Asd);
    } else if constexpr (RG > 0) {
        // the two 256 x 256 projections side by side: waves 0..3 the next convolution's Ps, waves 4..7 EdgeUpdate's Asd, 64 columns per wave
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const void* wq = wave < 4 ? a.Wps4 : a.Wasd4;
        float* out = wave < 4 ? a.Ps : a.Asd;
        if (out) {
            f32x4 acc[RG];
#pragma unroll
            for (int rg = 0; rg < RG; ++rg) acc[rg] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (!(FM_ABLATE & 32)) fm_wave_gemm4<64, RG>(acc, X, FM_LDX, wq, wave & 3, lane);
#pragma unroll
            for (int rg = 0; rg < RG; ++rg)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * rg + r < rows) out[(size_t)(row0 + 4 * rg + r) * 256 + 64 * (wave & 3) + lane] = acc[rg][r];
        }
    } else {
    if (a.Ps)
        fm_block_gemm<MT, 2>(X, FM_LDX, MT, 32, a.Wps, 16, [&](int row, int col, float val) {
            if (row0 + row < N) a.Ps[(size_t)(row0 + row) * 256 + col] = val;
        });
    if (a.Asd)
        fm_block_gemm<MT, 2>(X, FM_LDX, MT, 32, a.Wasd, 16, [&](int row, int col, float val) {
            if (row0 + row < N) a.Asd[(size_t)(row0 + row) * 256 + col] = val;
        });
    }
    if (a.PV)
        fm_block_gemm<1, 1>(Vin, T::LDVI, 3 * MT, V / 8, a.Wpv, (V + 16) / 16, [&](int row, int col, float val) {
            const int c = row / TM, r = row % TM;
            if (r < rows) a.PV[((size_t)(row0 + r) * 3 + c) * (V + 16) + col] = val;
        });
    if (a.x) {
        __syncthreads();                     // every wave has read the tile for the projections
        FM_MARK_DECL
        float pre[TM / 16][2][4];
        fm_gvp_core<V, V, false, true, TM, FM_THREADS, 0, SP, false, false, RG>(X, Vin, Vh, G, a.p0, pre FM_MARK_PASS(50));
        fm_gvp_core<V, V, false, true, TM, FM_THREADS, 0, SP, false, false, RG>(X, Vin, Vh, G, a.p1, pre FM_MARK_PASS(50));
        fm_gvp_core<V, 1, false, false, TM, FM_THREADS, 0, SP, false, false, RG>(X, Vin, Vh, G, a.p2, pre FM_MARK_PASS(50));
        if (tid < TM * 3) {
            const int r = tid / 3, c = tid % 3, n = row0 + r;
            if (r < rows) a.x[n * 3 + c] += Vin[(c * TM + r) * T::LDVI];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// NodePositionUpdate (vector_field.py:813-842): x += GVP3(GVP2(GVP1(s, v))).v[:, 0]
// ------------------------------------------------------------------------------------------------
struct FmPosArgs {
    int N;
    const float* s; const float* v;
    float* x;                   // (N,3) updated in place
    FmGvpW g0, g1, g2;
};

template <int V, int TM>
__global__ void __launch_bounds__(FM_THREADS) fm_k_pos_update(FmPosArgs a) {
    typedef FmGvpTile<V, TM> T;
    HIP_DYNAMIC_SHARED(float, lds)
    float* X = lds;
    float* Vin = X + T::X_FLOATS;
    float* Vh = Vin + T::VIN_FLOATS;
    float* G = Vh + T::VH_FLOATS;
    const int tid = threadIdx.x, row0 = blockIdx.x * TM;
    {   // tile of (s, v): 16-byte buffer loads, all requests of a thread issued before the LDS stores; rows beyond N read 0
        const int rows = a.N - row0 < TM ? a.N - row0 : TM;
        const auto rs_s = fm_buf(a.s + (size_t)row0 * 256, (unsigned)rows * 1024u);
        const auto rs_v = fm_buf(a.v + (size_t)row0 * 3 * V, (unsigned)rows * (3 * V * 4));
        constexpr int NQ = TM * 64 / FM_THREADS, V4 = V / 4, NCHK = TM * 3 * V4, NV = (NCHK + FM_THREADS - 1) / FM_THREADS;
        float4 q[NQ], qv[NV];
#pragma unroll
        for (int k = 0; k < NQ; ++k) q[k] = fm_buf_f32x4(rs_s, (tid + k * FM_THREADS) * 16, 0);
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int idx = tid + k * FM_THREADS;
            qv[k] = fm_buf_f32x4(rs_v, idx < NCHK ? idx * 16 : FM_BUF_OOB, 0);       // (r, c, u4) is the memory order of v
        }
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            const int idx = tid + k * FM_THREADS;
            *reinterpret_cast<float4*>(X + (idx >> 6) * FM_LDX + (idx & 63) * 4) = q[k];
        }
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int idx = tid + k * FM_THREADS;
            if (idx < NCHK) {
                const int r = idx / (3 * V4), rem = idx % (3 * V4), c = rem / V4, u4 = rem % V4;
                *reinterpret_cast<float4*>(Vin + (c * TM + r) * T::LDVI + u4 * 4) = qv[k];
            }
        }
    }
    __syncthreads();
    {
        FM_MARK_DECL
        float pre[TM / 16][2][4];
        fm_gvp_core<V, V, false, true, TM, FM_THREADS>(X, Vin, Vh, G, a.g0, pre FM_MARK_PASS(50));
        fm_gvp_core<V, V, false, true, TM, FM_THREADS>(X, Vin, Vh, G, a.g1, pre FM_MARK_PASS(50));
        fm_gvp_core<V, 1, false, false, TM, FM_THREADS>(X, Vin, Vh, G, a.g2, pre FM_MARK_PASS(50));
    }
    if (tid < TM * 3) {
        const int r = tid / 3, c = tid % 3, n = row0 + r;
        if (n < a.N) a.x[n * 3 + c] += Vin[(c * TM + r) * T::LDVI];
    }
}

// ------------------------------------------------------------------------------------------------
// EdgeUpdate (vector_field.py:864-880): ef = LN(ef + silu(W2 silu(W1 [s_src|s_dst|ef|rbf(d)] + b1) + b2))
// with the node parts of W1 hoisted into Asd (fm_k_node_proj).
// ------------------------------------------------------------------------------------------------
struct FmEdgeUpdArgs {
    FmBatch b;
    const float* x;            // updated positions
    const float* Asd;          // (N,256): [:128] = W1_src s, [128:] = W1_dst s
    float* ef;                 // (E,128) in place
    const float2* W1; const float* b1;      // K = 160 ([ef|rbf]), N = 128
    const float2* W2; const float* b2;      // K = 128, N = 128
    const float* ln_g; const float* ln_b;
    float rbf_mu_step, rbf_inv_sigma;
    int f_real;                // NARROW instances: real edge-feature width (< 128) of the LayerNorm statistics
    // HEAD instances (the evaluation's LAST EdgeUpdate): the edge output head (vector_field.py:340-344,364-367) on the tile's pairs -- the updated features
    // of a pair's two directed edges are on chip, nobody else reads them, so they are summed, sent through to_edge_logits and the softmax right here
    // and never written to HBM
    const float2* hW1; const float* hb1;    // K = 128, N = 128
    const float2* hW2; const float* hb2;    // K = 128, N = 16 (ne real columns)
    float* out_e; int ne;                   // (U, ne) bond-order probabilities
};

// HEAD = false: a tile is TM consecutive directed edges (internal order).  HEAD = true (TM = 32): a tile is 16 consecutive unordered pairs, rows 2k / 2k + 1 =
// the pair's two directed edges (src < dst first, the reference's upper edge): same per-row arithmetic, rows gathered instead of streamed, and the epilogue
// is the edge head of those 16 pairs instead of the store of the rows.
template <int TM, bool NARROW, bool HEAD = false>
__global__ void __launch_bounds__(FM_THREADS) fm_k_edge_update(FmEdgeUpdArgs a) {
    static_assert(!HEAD || (TM == 32 && !NARROW), "the fused edge head exists for 32-row tiles of full-width models");
    HIP_DYNAMIC_SHARED(float, lds)
    constexpr int LDX = 164, LDH = 132, MT = TM / 16, LPR = FM_THREADS / TM;
    float* X = lds;                       // [TM][164]: ef(128) | rbf(32)
    float* Hb = lds + TM * LDX;           // [TM][132]
    int* m_soff = reinterpret_cast<int*>(Hb + TM * LDH);      // [TM] byte offset of the source's row in Asd (FM_BUF_OOB for rows past the edge list:
    int* m_doff = m_soff + TM;                                // the range check returns 0 whatever column offset is added), same for the destination
    float* m_d = reinterpret_cast<float*>(m_doff + TM);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), e0 = blockIdx.x * TM;
    constexpr int NEF = TM * 32 / FM_THREADS;
    float4 efv[NEF];
    // the tile's ef rows depend only on the tile index: request them first.  Tile-relative descriptor; its range check
    // zero-fills the rows of a ragged last tile on load and drops them on the final store.
    const int left = a.b.E - e0;
    const auto rs_ef = fm_buf(a.ef + (size_t)(HEAD ? 0 : e0) * 128, HEAD ? 0u : (unsigned)(left < TM ? left : TM) * 512u);
    if constexpr (!HEAD) {
#pragma unroll
        for (int k = 0; k < NEF; ++k) efv[k] = fm_buf_f32x4(rs_ef, tid * 16 + k * FM_THREADS * 16, 0);
    }
    int* m_eoff = reinterpret_cast<int*>(m_d + TM);           // HEAD: [TM] byte offset of the row's edge in ef relative to the tile's smallest edge id, [TM] = that id
    if (HEAD && tid < 64) {
        const int p = blockIdx.x * (TM / 2) + (tid >> 1);
        const int e = (tid < TM && p < a.b.U) ? ((tid & 1) ? a.b.p_e1[p] : a.b.p_e0[p]) : 0x7fffffff;
        int mn = e;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_xor(mn, o); mn = t < mn ? t : mn; }      // the tile's pairs lie within a few molecules: relative offsets stay below 2^31
        if (tid < TM) {
            int s = -1, d = -1; float dist = 0.f;
            if (e != 0x7fffffff) {
                s = a.b.e_src[e]; d = a.b.e_dst[e];
                dist = fm_norm3(a.x[s * 3] - a.x[d * 3], a.x[s * 3 + 1] - a.x[d * 3 + 1], a.x[s * 3 + 2] - a.x[d * 3 + 2]) + 1e-8f;
            }
            m_soff[tid] = s >= 0 ? s * 1024 : FM_BUF_OOB; m_doff[tid] = s >= 0 ? d * 1024 : FM_BUF_OOB; m_d[tid] = dist;
            m_eoff[tid] = s >= 0 ? (e - mn) * 512 : FM_BUF_OOB;
        }
        if (tid == 0) m_eoff[TM] = mn;
    }
    if (!HEAD && tid < TM) {
        const int e = e0 + tid;
        int s = -1, d = -1; float dist = 0.f;
        if (e < a.b.E) {
            s = a.b.e_src[e]; d = a.b.e_dst[e];
            dist = fm_norm3(a.x[s * 3] - a.x[d * 3], a.x[s * 3 + 1] - a.x[d * 3 + 1], a.x[s * 3 + 2] - a.x[d * 3 + 2]) + 1e-8f;
        }
        m_soff[tid] = s >= 0 ? s * 1024 : FM_BUF_OOB; m_doff[tid] = s >= 0 ? d * 1024 : FM_BUF_OOB; m_d[tid] = dist;
    }
    __syncthreads();
    if constexpr (HEAD) {       // the rows of the tile's edges, gathered: 16-byte loads, all of a thread's requests back to back (rows without a pair read 0)
        const int emin = __builtin_amdgcn_readfirstlane(m_eoff[TM]);
        const auto rs_g = fm_buf(a.ef + (size_t)(emin == 0x7fffffff ? 0 : emin) * 128, 0x7ffffe00u);
#pragma unroll
        for (int k = 0; k < NEF; ++k) {
            const int idx = tid + k * FM_THREADS;
            efv[k] = fm_buf_f32x4(rs_g, m_eoff[idx >> 5] + (idx & 31) * 16, 0);
        }
    }
    // layer 1: wave w owns column tile w for all MT row tiles.  The hoisted node terms W1_src*s[src] + W1_dst*s[dst]
    // (+ bias) are requested before the GEMM so their L2 latency hides behind the fill and the MFMAs; rows without an
    // edge read 0 through the range check.
    const int col = wave * 16 + (lane & 15);
    float pre_s[MT][4], pre_d[MT][4];
    const float b1 = a.b1[col], b2 = a.b2[col];
    {
        const auto rs = fm_buf(a.Asd, (unsigned)a.b.N * 1024u);
#pragma unroll
        for (int i = 0; i < MT; ++i) {       // the lane's four rows of a row tile: one ds_read_b128 per table, one v_add per gathered row (fm_gather_pre's addressing)
            const int4 so = *reinterpret_cast<const int4*>(m_soff + i * 16 + 4 * (lane >> 4)), dof = *reinterpret_cast<const int4*>(m_doff + i * 16 + 4 * (lane >> 4));
            const int so_[4] = {so.x, so.y, so.z, so.w}, do_[4] = {dof.x, dof.y, dof.z, dof.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pre_s[i][r] = fm_buf_f32(rs, so_[r] + (lane & 15) * 4, wave * 64);
                pre_d[i][r] = fm_buf_f32(rs, do_[r] + (lane & 15) * 4, 512 + wave * 64);
            }
        }
    }
    {   // thread -> (row fr + 16 k, float4 column c4): one address per thread, the passes are immediate offsets
        constexpr int RP = FM_THREADS / 32;
        const int fr = tid >> 5, c4 = tid & 31;
        float* xe = X + fr * LDX + 4 * c4;
        float* xr = X + fr * LDX + 128 + c4;
#pragma unroll
        for (int k = 0; k < NEF; ++k) {
            *reinterpret_cast<float4*>(xe + k * RP * LDX) = efv[k];       // one ds_write_b128 (row pitch 656 B and column offset are multiples of 16 B)
            xr[k * RP * LDX] = fm_rbf(m_d[fr + k * RP], c4, a.rbf_mu_step, a.rbf_inv_sigma);      // rows past the edge list (d = 0): finite junk, dropped by the final store's range check
        }
    }
    __syncthreads();
    float* ho = Hb + (4 * (lane >> 4)) * LDH + col;
    float* xo = X + (4 * (lane >> 4)) * LDX + col;
    {
        f32x4 acc[MT][1];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][0][r] = (pre_s[i][r] + pre_d[i][r]) + b1;
        fm_wave_gemm<MT, 1>(acc, X, LDX, 160 / 8, a.W1, 8, wave, lane);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) ho[(i * 16 + r) * LDH] = fm_silu(acc[i][0][r]);
    }
    __syncthreads();
    {
        f32x4 acc[MT][1];
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[i][0] = f32x4{b2, b2, b2, b2};
        fm_wave_gemm<MT, 1>(acc, Hb, LDH, 128 / 8, a.W2, 8, wave, lane);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) xo[(i * 16 + r) * LDX] = fm_fma(acc[i][0][r], fm_sigmoid(acc[i][0][r]), xo[(i * 16 + r) * LDX]);   // own element: ef + silu(.), one fma
    }
    __syncthreads();
    const int r = tid / LPR, sub = tid % LPR;
    float mean, rstd;
    fm_row_stats<LPR>(X + r * LDX, NARROW ? a.f_real : 128, sub, mean, rstd);
#pragma unroll
    for (int j = 0; j < 32 / LPR; ++j) {
        const int c = (j * LPR + sub) * 4;
        const float4 xv = *reinterpret_cast<const float4*>(X + r * LDX + c);
        const float4 g = reinterpret_cast<const float4*>(a.ln_g)[c >> 2], bb = reinterpret_cast<const float4*>(a.ln_b)[c >> 2];
        float4 o;
        o.x = fm_fma((xv.x - mean) * rstd, g.x, bb.x);
        o.y = fm_fma((xv.y - mean) * rstd, g.y, bb.y);
        o.z = fm_fma((xv.z - mean) * rstd, g.z, bb.z);
        o.w = fm_fma((xv.w - mean) * rstd, g.w, bb.w);
        if constexpr (HEAD) *reinterpret_cast<float4*>(X + r * LDX + c) = o;       // stays on chip: own elements, in place
        else fm_buf_store_f32x4(rs_ef, r * 512 + c * 4, 0, o);
    }
    if constexpr (HEAD) {
        // ---- edge output head on the tile's 16 pairs: x = ef[upper] + ef[lower] (vector_field.py:342), Linear(128,128) . SiLU . Linear(128,ne), softmax
        constexpr int NP = TM / 2;
        float* P = Hb;                          // [16][132]: pair inputs (the hidden tile is dead)
        float* part = Hb + NP * LDH;            // [8][16][16]: K-slice partial logits (2048 floats <= the other half of Hb)
        static_assert(NP * LDH + 2048 <= TM * LDH, "partial logits must fit into the hidden tile");
        __syncthreads();
        for (int idx = tid; idx < NP * 32; idx += FM_THREADS) {
            const int k = idx >> 5, c4 = idx & 31;
            const float4 u = *reinterpret_cast<const float4*>(X + (2 * k) * LDX + 4 * c4), l = *reinterpret_cast<const float4*>(X + (2 * k + 1) * LDX + 4 * c4);
            *reinterpret_cast<float4*>(P + k * LDH + 4 * c4) = make_float4(u.x + l.x, u.y + l.y, u.z + l.z, u.w + l.w);
        }
        __syncthreads();
        {   // layer 1: 16 x 128 -> 128, wave w owns column tile w; result (SiLU) into rows 0..15 of X (dead: every pair sum has been formed)
            f32x4 acc[1][1];
            const float hb1 = a.hb1[col];
            acc[0][0] = f32x4{hb1, hb1, hb1, hb1};
            fm_wave_gemm<1, 1>(acc, P, LDH, 128 / 8, a.hW1, 8, wave, lane);
#pragma unroll
            for (int q = 0; q < 4; ++q) xo[q * LDX] = fm_silu(acc[0][0][q]);
        }
        __syncthreads();
        {   // layer 2: 16 x 128 -> 16 (ne real columns): ONE column tile, its K = 128 split over the eight waves (two k-steps each)
            f32x4 acc[1][1] = {{f32x4{0.f, 0.f, 0.f, 0.f}}};
            fm_wave_gemm<1, 1>(acc, X + 16 * wave, LDX, 2, a.hW2 + (size_t)(2 * wave) * 64, 1, 0, lane);
#pragma unroll
            for (int q = 0; q < 4; ++q) part[(wave * 16 + 4 * (lane >> 4) + q) * 16 + (lane & 15)] = acc[0][0][q];
        }
        __syncthreads();
        if (tid < NP) {
            const int p = blockIdx.x * NP + tid;
            if (p < a.b.U) {
                float lg[16];
                for (int c = 0; c < a.ne; ++c) {
                    float v = part[tid * 16 + c];
#pragma unroll
                    for (int w = 1; w < 8; ++w) v += part[(w * 16 + tid) * 16 + c];
                    lg[c] = v + a.hb2[c];
                }
                float m = lg[0];
                for (int c = 1; c < a.ne; ++c) m = fmaxf(m, lg[c]);
                float sum = 0.f;
                for (int c = 0; c < a.ne; ++c) sum += expf(lg[c] - m);
                for (int c = 0; c < a.ne; ++c) a.out_e[(size_t)p * a.ne + c] = expf(lg[c] - m) / sum;
            }
        }
    }
}

// EdgeUpdate in the opt-in split precision (fm_device.h "bf16x3"): both linear layers on v_mfma_f32_16x16x32_bf16 with hi/lo planes;
// the f32 copy of ef stays in LDS for the residual and the LayerNorm.  With the matrix work off the f32 ALU the kernel is bound by
// its 1 KB per edge of HBM traffic and by VALU (SiLU, LayerNorm, the splits).
struct FmEdgeUpdSpW { const void* W1; const void* W2; };
template <int TM, int FMT = 0>      // FMT: 0 = bf16 planes, 1 = half planes
__global__ void __launch_bounds__(FM_THREADS) fm_k_edge_update_sp(FmEdgeUpdArgs a, FmEdgeUpdSpW w) {
    HIP_DYNAMIC_SHARED(float, lds)
    constexpr int LDF = 132, LD1 = 176, LD2 = 144, MT = TM / 16, LPR = FM_THREADS / TM;     // f32 ef tile; planes of [ef | rbf] (K = 160) and of the hidden layer (K = 128)
    float* Xf = lds;                                                         // [TM][132] ef (f32), later ef + update
    unsigned short* P1H = reinterpret_cast<unsigned short*>(Xf + TM * LDF);  // [TM][176] x 2
    unsigned short* P1L = P1H + TM * LD1;
    unsigned short* P2H = P1H;                                               // [TM][144] x 2, over the (then dead) layer-1 planes: 39.8 KB per
    unsigned short* P2L = P2H + TM * LD2;                                    // workgroup, four workgroups per CU hide the HBM latency of the ef tile
    int* m_src = reinterpret_cast<int*>(P1L + TM * LD1);
    int* m_dst = m_src + TM;
    float* m_d = reinterpret_cast<float*>(m_dst + TM);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), e0 = blockIdx.x * TM;
    const int left = a.b.E - e0;
    const auto rs_ef = fm_buf(a.ef + (size_t)e0 * 128, (unsigned)(left < TM ? left : TM) * 512u);
    constexpr int NEF = TM * 32 / FM_THREADS;
    float4 efv[NEF];
#pragma unroll
    for (int k = 0; k < NEF; ++k) efv[k] = fm_buf_f32x4(rs_ef, tid * 16 + k * FM_THREADS * 16, 0);
    if (tid < TM) {
        const int e = e0 + tid;
        int s = -1, d = -1; float dist = 0.f;
        if (e < a.b.E) {
            s = a.b.e_src[e]; d = a.b.e_dst[e];
            dist = fm_norm3(a.x[s * 3] - a.x[d * 3], a.x[s * 3 + 1] - a.x[d * 3 + 1], a.x[s * 3 + 2] - a.x[d * 3 + 2]) + 1e-8f;
        }
        m_src[tid] = s; m_dst[tid] = d; m_d[tid] = dist;
    }
    __syncthreads();
    const int col = wave * 16 + (lane & 15);
    float pre_s[MT][4], pre_d[MT][4];
    const float b1 = a.b1[col], b2 = a.b2[col];
    {
        const auto rs = fm_buf(a.Asd, (unsigned)a.b.N * 1024u);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i * 16 + 4 * (lane >> 4) + r;
                const int sidx = m_src[row], didx = m_dst[row];
                pre_s[i][r] = fm_buf_f32(rs, sidx >= 0 ? sidx * 1024 + (lane & 15) * 4 : FM_BUF_OOB, wave * 64);
                pre_d[i][r] = fm_buf_f32(rs, sidx >= 0 ? didx * 1024 + (lane & 15) * 4 : FM_BUF_OOB, 512 + wave * 64);
            }
    }
#pragma unroll
    for (int k = 0; k < NEF; ++k) {
        const int idx = tid + k * FM_THREADS, r = idx >> 5, c4 = idx & 31;
        *reinterpret_cast<float4*>(Xf + r * LDF + 4 * c4) = efv[k];
        fm_split_store<LD1, FMT>(P1H, P1L, r, 4 * c4 + 0, efv[k].x); fm_split_store<LD1, FMT>(P1H, P1L, r, 4 * c4 + 1, efv[k].y);
        fm_split_store<LD1, FMT>(P1H, P1L, r, 4 * c4 + 2, efv[k].z); fm_split_store<LD1, FMT>(P1H, P1L, r, 4 * c4 + 3, efv[k].w);
        fm_split_store<LD1, FMT>(P1H, P1L, r, 128 + c4, fm_rbf(m_d[r], c4, a.rbf_mu_step, a.rbf_inv_sigma));
    }
    __syncthreads();
    {
        f32x4 acc[MT][1];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][0][r] = ((pre_s[i][r] + pre_d[i][r]) + b1) * fm_sp_wscale<FMT>();
        fm_wave_gemm_sp<MT, 1, LD1, FMT>(acc, P1H, P1L, 0, 5, w.W1, 8, wave, lane);
        __syncthreads();                     // every wave has read the layer-1 planes: the hidden layer's planes take their place
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) fm_split_store<LD2, FMT>(P2H, P2L, i * 16 + 4 * (lane >> 4) + r, col, fm_silu(acc[i][0][r] * (1.0f / fm_sp_wscale<FMT>())));
    }
    __syncthreads();
    {
        f32x4 acc[MT][1];
#pragma unroll
        for (int i = 0; i < MT; ++i) { const float b2s = b2 * fm_sp_wscale<FMT>(); acc[i][0] = f32x4{b2s, b2s, b2s, b2s}; }
        fm_wave_gemm_sp<MT, 1, LD2, FMT>(acc, P2H, P2L, 0, 4, w.W2, 8, wave, lane);
        float* xo = Xf + (4 * (lane >> 4)) * LDF + col;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) xo[(i * 16 + r) * LDF] += fm_silu(acc[i][0][r] * (1.0f / fm_sp_wscale<FMT>()));   // own element: ef + update
    }
    __syncthreads();
    const int r = tid / LPR, sub = tid % LPR;
    float mean, rstd;
    fm_row_stats<LPR>(Xf + r * LDF, a.f_real, sub, mean, rstd);
#pragma unroll
    for (int j = 0; j < 32 / LPR; ++j) {
        const int c = (j * LPR + sub) * 4;
        const float4 xv = *reinterpret_cast<const float4*>(Xf + r * LDF + c);
        const float4 g = reinterpret_cast<const float4*>(a.ln_g)[c >> 2], bb = reinterpret_cast<const float4*>(a.ln_b)[c >> 2];
        float4 o;
        o.x = fm_fma((xv.x - mean) * rstd, g.x, bb.x);
        o.y = fm_fma((xv.y - mean) * rstd, g.y, bb.y);
        o.z = fm_fma((xv.z - mean) * rstd, g.z, bb.z);
        o.w = fm_fma((xv.w - mean) * rstd, g.w, bb.w);
        fm_buf_store_f32x4(rs_ef, r * 512 + c * 4, 0, o);
    }
}

// ------------------------------------------------------------------------------------------------
// small element-wise kernels
// ------------------------------------------------------------------------------------------------
// endpoint-parameterised models: node rows of the scalar embedding's input [a_t | c_t | temb | 0] (vector_field.py:228-243)
static __global__ void __launch_bounds__(256) fm_k_dense_node_in(float* __restrict__ out, int ld, int N, const float* __restrict__ a, int na,
                                                           const float* __restrict__ c, int nc, const float* __restrict__ temb, int tt) {
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < (size_t)N * ld; idx += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(idx / ld), k = (int)(idx % ld);
        float v = 0.f;
        if (k < na) v = a[(size_t)n * na + k];
        else if (k < na + nc) v = c[(size_t)n * nc + (k - na)];
        else if (k < na + nc + tt) v = temb[k - na - nc];
        out[idx] = v;
    }
}

// EndpointVectorField.step (vector_field.py:528-543) for one flat feature array per blockIdx.y:
//   vf = coef * (x_1 - x_t);  vf = vf * scale;  x_s = x_t + vf * dt      (never-contracted, as torch's separately rounded ops)
struct FmEndpointStepArgs { float* xt[4]; const float* x1[4]; int n[4]; float coef[4]; float scale, dt; };
static __global__ void __launch_bounds__(256) fm_k_endpoint_step(FmEndpointStepArgs a) {
    const int f = blockIdx.y;
    float* xt = a.xt[f]; const float* x1 = a.x1[f];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n[f]; i += gridDim.x * blockDim.x) {
        const float vf = fm_mul_rn(fm_mul_rn(a.coef[f], fm_sub_rn(x1[i], xt[i])), a.scale);
        xt[i] = fm_add_rn(xt[i], fm_mul_rn(vf, a.dt));
    }
}

// x -= per-molecule mean (vector_field.py:347-350); one 64-lane workgroup per molecule
static __global__ void __launch_bounds__(64) fm_k_remove_com(float* __restrict__ x, const int* __restrict__ mol_node_off) {
    const int m = blockIdx.x, lane = threadIdx.x;
    const int n0 = mol_node_off[m], n1 = mol_node_off[m + 1];
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int n = n0 + lane; n < n1; n += 64) { sx += x[n * 3]; sy += x[n * 3 + 1]; sz += x[n * 3 + 2]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sx += __shfl_xor(sx, o); sy += __shfl_xor(sy, o); sz += __shfl_xor(sz, o); }
    const float inv = 1.0f / (float)(n1 - n0);
    sx *= inv; sy *= inv; sz *= inv;
    for (int n = n0 + lane; n < n1; n += 64) { x[n * 3] -= sx; x[n * 3 + 1] -= sy; x[n * 3 + 2] -= sz; }
}

// Euler step for positions (ctmc_vector_field.py:331-334): x_t += dt * (coef * (x1 - x_t)), coef = alpha'/(1-alpha)
//   scale = inv_temp_func(t_i), 1 by default (multiplying by 1.0f is exact)
static __global__ void __launch_bounds__(256) fm_k_x_step(float* __restrict__ x_t, const float* __restrict__ x1, float coef, float dt, float scale, int n3) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n3) {
        const float vf = fm_mul_rn(coef, fm_sub_rn(x1[i], x_t[i]));
        x_t[i] = fm_add_rn(x_t[i], fm_mul_rn(fm_mul_rn(dt, vf), scale));
    }
}

// Position prior of the Philox mode: x0 ~ N(0, I) per atom from the molecule's own stream (Box-Muller on draw block
// (atom, 0xFFFFFFFF, 0)), then minus the molecule's mean (priors.py:27-35) -- one 64-lane workgroup per molecule.
static __global__ void __launch_bounds__(64) fm_k_prior_philox(float* __restrict__ x, const int* __restrict__ mol_node_off,
                                                         const int* __restrict__ mol_gid, unsigned seed_lo, unsigned seed_hi) {
    const int m = blockIdx.x, lane = threadIdx.x;
    const int n0 = mol_node_off[m], n1 = mol_node_off[m + 1];
    const unsigned gid = (unsigned)mol_gid[m];
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int n = n0 + lane; n < n1; n += 64) {
        const FmPhilox4 rn = fm_philox4x32(gid, (unsigned)(n - n0), 0xFFFFFFFFu, 0u, seed_lo, seed_hi);
        const float r0 = sqrtf(2.0f * fm_exp1(rn.v[0])), r1 = sqrtf(2.0f * fm_exp1(rn.v[2]));
        const float t0 = 6.283185307179586f * fm_u01(rn.v[1]), t1 = 6.283185307179586f * fm_u01(rn.v[3]);
        const float gx = r0 * cosf(t0), gy = r0 * sinf(t0), gz = r1 * cosf(t1);
        x[n * 3] = gx; x[n * 3 + 1] = gy; x[n * 3 + 2] = gz;
        sx += gx; sy += gy; sz += gz;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sx += __shfl_xor(sx, o); sy += __shfl_xor(sy, o); sz += __shfl_xor(sz, o); }
    const float inv = 1.0f / (float)(n1 - n0);
    sx *= inv; sy *= inv; sz *= inv;
    for (int n = n0 + lane; n < n1; n += 64) { x[n * 3] -= sx; x[n * 3 + 1] -= sy; x[n * 3 + 2] -= sz; }
}

// ------------------------------------------------------------------------------------------------
// CTMC categorical update (ctmc_vector_field.py:349-357,414-461; ctmc_utils.py:4-34) + the Euler step of the positions
// (:331-334): the whole campbell update of one integration step in ONE launch.  Grid (B, 4): one workgroup per
// (molecule, job), job 0..2 = modality a, c, e, job 3 = the molecule's positions.
//   pass 1: p~ = softmax(log p / T); x1 = argmax((p~/sum p~)/q) (== torch.multinomial n=1 fast path); counts of the
//           molecule's masked rows (m) and high-confidence masked rows (h)
//   pass 2: per-row unmask / re-mask decisions and the new token.
// A molecule's rows are contiguous (atoms: mol_node_off, unordered pairs: mol_pair_off), so m / h are a workgroup
// reduction (registers -> wave shuffle -> LDS) instead of one global atomic per masked row (round 1: 1.1 M atomics onto
// 1024 addresses at the bench size, 0.3 ms per modality), no counter-zeroing pass exists, and pass 2 re-reads only what
// the same thread wrote in pass 1 (x1 with the high-confidence flag parked in bit 8).  Every op that must equal torch's
// separately rounded f32 op is never-contracted (fm_*_rn); logf / expf / IEEE division as in round 1 (bit-exact vs the
// CPU reference path on every fixture).
// ------------------------------------------------------------------------------------------------
struct FmCtmcMod {
    int K;                          // real categories; mask token = K
    const float* p;                 // (rows,K)
    int* xt; int* x1;               // (rows)
    const float* q; const float* u1; const float* u2;
    const int* off;                 // [B+1] first row of every molecule
    float unmask_prob, mask_prob;
    int* sink_t;                    // trajectory sink: this step's frame of the new state tokens (rows), null = off
};
struct FmCtmcFusedArgs {
    FmCtmcMod mod[3];
    float temp, hc_thresh;
    int last_step;
    float* x_t; const float* x1; const int* node_off; float coef, dt, scale;     // Euler step (fm_k_x_step)
    int philox; unsigned seed_lo, seed_hi; int step; const int* mol_gid;      // philox != 0: q / u1 / u2 come from fm_philox4x32, not from memory
    const float* x_raw; float* x1_out;   // x_raw != null: the network's raw endpoint positions; x1_out = x_raw - per-molecule mean
                                         // (vector_field.py:347-350, arithmetic of fm_k_remove_com) is written first and used as x1
    // trajectory sink (ctmc_vector_field.py:235-255), null = off: this step's frames of the new positions and of the endpoint positions.  The kernel holds
    // every value of a frame in registers when it writes the state, so the frames cost one more store each -- not five copy nodes per step
    float* sink_x; float* sink_x1;
};

// NT threads per workgroup: 256, or 1024 for batches of a few molecules -- there one workgroup per (molecule, modality) is all the parallelism the kernel has,
// and a 47-atom molecule's 1081 pair rows are five dependent load-compute rounds of 256 threads but two of 1024 (one molecule: 13.9 -> 8 us per step)
template <int NT>
__global__ void __launch_bounds__(NT) fm_k_ctmc_fused(FmCtmcFusedArgs a) {
    __shared__ int red[2][NT / 64];
    const int mol = blockIdx.x, job = blockIdx.y, tid = threadIdx.x;
    if (job == 3) {
        const int n0 = a.node_off[mol], n1 = a.node_off[mol + 1];
        if (a.x_raw) {
            __shared__ float com[3];
            if (tid < 64) {          // the 64-lane strided sum + xor tree of fm_k_remove_com: identical rounding
                float sx = 0.f, sy = 0.f, sz = 0.f;
                for (int n = n0 + tid; n < n1; n += 64) { sx += a.x_raw[n * 3]; sy += a.x_raw[n * 3 + 1]; sz += a.x_raw[n * 3 + 2]; }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { sx += __shfl_xor(sx, o); sy += __shfl_xor(sy, o); sz += __shfl_xor(sz, o); }
                const float inv = 1.0f / (float)(n1 - n0);
                if (tid == 0) { com[0] = sx * inv; com[1] = sy * inv; com[2] = sz * inv; }
            }
            __syncthreads();
            for (int i = n0 * 3 + tid; i < n1 * 3; i += NT) {
                const float x1 = a.x_raw[i] - com[(i - n0 * 3) % 3];
                a.x1_out[i] = x1;
                const float vf = fm_mul_rn(a.coef, fm_sub_rn(x1, a.x_t[i]));
                const float xn = fm_add_rn(a.x_t[i], fm_mul_rn(fm_mul_rn(a.dt, vf), a.scale));
                a.x_t[i] = xn;
                if (a.sink_x) a.sink_x[i] = xn;
                if (a.sink_x1) a.sink_x1[i] = x1;
            }
            return;
        }
        for (int i = n0 * 3 + tid; i < n1 * 3; i += NT) {
            const float x1 = a.x1[i];
            const float vf = fm_mul_rn(a.coef, fm_sub_rn(x1, a.x_t[i]));
            const float xn = fm_add_rn(a.x_t[i], fm_mul_rn(fm_mul_rn(a.dt, vf), a.scale));
            a.x_t[i] = xn;
            if (a.sink_x) a.sink_x[i] = xn;
            if (a.sink_x1) a.sink_x1[i] = x1;
        }
        return;
    }
    const FmCtmcMod md = a.mod[job];
    const int r0 = md.off[mol], r1 = md.off[mol + 1], K = md.K;
    const unsigned gid = a.philox ? (unsigned)a.mol_gid[mol] : 0u, ctr2 = (unsigned)a.step * 4u + (unsigned)job;
    int cm = 0, ch = 0;
    for (int i = r0 + tid; i < r1; i += NT) {
        float lp[16], qv[16];
        if (a.philox) {            // draws 0..K-1 of this row's stream: Exp(1) for the categorical sample
            for (int blk = 0; blk * 4 < K; ++blk) {
                const FmPhilox4 rn = fm_philox4x32(gid, (unsigned)(i - r0), ctr2, (unsigned)blk, a.seed_lo, a.seed_hi);
#pragma unroll
                for (int j = 0; j < 4; ++j) qv[(blk * 4 + j) & 15] = fm_exp1(rn.v[j]);
            }
        } else {
            for (int k = 0; k < K; ++k) qv[k] = md.q[(size_t)i * K + k];
        }
        float mx = -INFINITY;
        for (int k = 0; k < K; ++k) { lp[k] = fm_div_rn(logf(md.p[(size_t)i * K + k]), a.temp); mx = fmaxf(mx, lp[k]); }
        float sum = 0.f;
        for (int k = 0; k < K; ++k) { lp[k] = expf(fm_sub_rn(lp[k], mx)); sum = fm_add_rn(sum, lp[k]); }
        float purity = 0.f, psum = 0.f;
        for (int k = 0; k < K; ++k) { lp[k] = fm_div_rn(lp[k], sum); purity = fmaxf(purity, lp[k]); psum = fm_add_rn(psum, lp[k]); }
        int best = 0; float bestv = -1.f;
        for (int k = 0; k < K; ++k) {
            const float v = fm_div_rn(fm_div_rn(lp[k], psum), qv[k]);
            if (v > bestv) { bestv = v; best = k; }
        }
        const bool masked = md.xt[i] == K;
        const bool hc = masked && (purity >= a.hc_thresh);
        md.x1[i] = best | (hc ? 256 : 0);
        cm += masked ? 1 : 0; ch += hc ? 1 : 0;
    }
    if (a.hc_thresh > 0.f) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { cm += __shfl_xor(cm, o); ch += __shfl_xor(ch, o); }
        if ((tid & 63) == 0) { red[0][tid >> 6] = cm; red[1][tid >> 6] = ch; }
        __syncthreads();
        cm = ch = 0;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) { cm += red[0][w]; ch += red[1][w]; }
    }
    // per-molecule probabilities: ph = min(unmask_prob*m/h, 1) (inf when h == 0); pl = (unmask_prob*m - ph*h)/(m-h)
    const float m = (float)cm, h = (float)ch;
    const float um = fm_mul_rn(md.unmask_prob, m);
    float ph = (ch == 0) ? INFINITY : fm_div_rn(um, h);
    ph = fminf(ph, 1.0f);
    const float pl = fm_div_rn(fm_sub_rn(um, fm_mul_rn(ph, h)), fm_sub_rn(m, h));
    for (int i = r0 + tid; i < r1; i += NT) {
        const int packed = md.x1[i];
        const int x1 = packed & 255;
        const int tok = md.xt[i];
        const bool masked = tok == K;
        float u1, u2 = 1.f;
        if (a.philox) {            // draws 16, 17 of the row's stream (block 4), clear of the <= 16 categorical draws
            const FmPhilox4 rn = fm_philox4x32(gid, (unsigned)(i - r0), ctr2, 4u, a.seed_lo, a.seed_hi);
            u1 = fm_u01(rn.v[0]); u2 = fm_u01(rn.v[1]);
        } else {
            u1 = md.u1[i];
            if (!a.last_step) u2 = md.u2[i];
        }
        bool will_unmask;
        if (a.hc_thresh > 0.f) {
            const float prob = masked ? ((packed & 256) ? ph : pl) : 0.f;
            will_unmask = u1 < prob;           // comparisons against NaN are false, as in torch
        } else {
            will_unmask = (u1 < md.unmask_prob) && masked;
        }
        int nt = tok;
        if (!a.last_step) { if ((u2 < md.mask_prob) && !masked) nt = K; }
        if (will_unmask) nt = x1;
        md.xt[i] = nt;
        md.x1[i] = x1;
        if (md.sink_t) md.sink_t[i] = nt;
    }
}

// ------------------------------------------------------------------------------------------------
// dfm_type 'gat' (ctmc_vector_field.py:373-388, 463-510): one draw from the transition distribution
//   p_step = clamp(delta_xt + dt*(fw*u_f - bw*u_b), 1e-9, 1) over K+1 classes (mask = class K),
//   u_f = cf*([p~, 0] - delta_xt), u_b = cb*(delta_xt - delta_mask); x_{t+dt} = argmax((p_step/sum)/q).
// x1 receives argmax(p~) (the reference records the tempered probabilities as the "endpoint" frame).
// ------------------------------------------------------------------------------------------------
struct FmGatArgs {
    int rows, K;
    const float* p;                 // (rows,K) endpoint probabilities (un-tempered softmax)
    int* xt;                        // (rows) tokens, updated in place
    int* x1;                        // (rows) argmax of the tempered endpoint distribution
    const float* q;                 // (rows,K+1) Exp(1) noise
    float temp, cf, cb, fw, bw, dt;
};

static __global__ void __launch_bounds__(256) fm_k_ctmc_gat(FmGatArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.rows) return;
    float lp[17];
    float mx = -INFINITY;
    for (int k = 0; k < a.K; ++k) { lp[k] = fm_div_rn(logf(a.p[(size_t)i * a.K + k]), a.temp); mx = fmaxf(mx, lp[k]); }
    float sum = 0.f;
    for (int k = 0; k < a.K; ++k) { lp[k] = expf(fm_sub_rn(lp[k], mx)); sum = fm_add_rn(sum, lp[k]); }
    int amax = 0; float amaxv = -1.f;
    for (int k = 0; k < a.K; ++k) { lp[k] = fm_div_rn(lp[k], sum); if (lp[k] > amaxv) { amaxv = lp[k]; amax = k; } }
    lp[a.K] = 0.f;
    const int tok = a.xt[i];
    float psum = 0.f;
    for (int k = 0; k <= a.K; ++k) {
        const float dx = (k == tok) ? 1.f : 0.f, dm = (k == a.K) ? 1.f : 0.f;
        const float uf = fm_mul_rn(a.cf, fm_sub_rn(lp[k], dx));
        const float ub = fm_mul_rn(a.cb, fm_sub_rn(dx, dm));
        const float pv = fm_sub_rn(fm_mul_rn(a.fw, uf), fm_mul_rn(a.bw, ub));
        float ps = fm_add_rn(dx, fm_mul_rn(a.dt, pv));
        ps = fminf(fmaxf(ps, 1.0e-9f), 1.0f);
        lp[k] = ps;
        psum = fm_add_rn(psum, ps);
    }
    int best = 0; float bestv = -1.f;
    for (int k = 0; k <= a.K; ++k) {
        const float v = fm_div_rn(fm_div_rn(lp[k], psum), a.q[(size_t)i * (a.K + 1) + k]);
        if (v > bestv) { bestv = v; best = k; }
    }
    a.xt[i] = best;
    a.x1[i] = amax;
}

// ------------------------------------------------------------------------------------------------
// Valence stability + connectivity of the sampled molecules straight from the final tokens (reference
// flowmol/analysis/metrics.py:96-117,333-363 with molecule_builder.py:138-157,217-265: fake atoms removed,
// mask bond token = no bond, valency = sum of bond orders with aromatic = 1.5).  One workgroup per molecule.
//   table[a * n_charges + c]: bit v set <=> valency v is valid for atom token a with charge token c
//   (explicit aromaticity: bit n_arom*8 + v for the (n_arom, v) pairs of the aromatic tables); 0 = charge not listed.
//   out[m] = {stable atoms, real atoms, connected components of the real atoms, size of the largest component}
// ------------------------------------------------------------------------------------------------
struct FmStabArgs {
    FmBatch b;
    const int* a; const int* c; const int* e;     // tokens: (N), (N), (U)
    const unsigned* table; int n_types, n_charges;
    int fake_tok;                                  // atom token of the fake atom ('Sn'), -1 if the model has none
    int ne;                                        // number of bond types; token ne = mask = no bond
    int arom;                                      // explicit aromaticity: bond token 4 = aromatic (1.5)
    int* out;                                      // (B,4)
};

static __global__ void __launch_bounds__(64) fm_k_stability(FmStabArgs s) {
    HIP_DYNAMIC_SHARED(int, sm)
    const int m = blockIdx.x, tid = threadIdx.x;
    const int n0 = s.b.mol_node_off[m], n = s.b.mol_node_off[m + 1] - n0, p0 = s.b.mol_pair_off[m];
    int* label = sm;            // [n]
    int* cnt = sm + n;          // [n]
    __shared__ int acc[4];
    __shared__ int changed;
    if (tid < 4) acc[tid] = 0;
    __syncthreads();
    auto bond = [&](int i, int j) {       // bond token of the unordered pair, mask -> 0
        const int lo = i < j ? i : j, hi = i < j ? j : i;
        const int t = s.e[p0 + lo * (2 * n - lo - 1) / 2 + (hi - lo - 1)];
        return t == s.ne ? 0 : t;
    };
    for (int i = tid; i < n; i += 64) {
        const int at = s.a[n0 + i];
        const bool real = at != s.fake_tok;
        label[i] = real ? i : -1;
        cnt[i] = 0;
        if (!real) continue;
        int val2 = 0, narom = 0;           // valency in half-bond units
        for (int j = 0; j < n; ++j) {
            if (j == i || s.a[n0 + j] == s.fake_tok) continue;
            const int t = bond(i, j);
            if (s.arom && t == 4) { val2 += 3; ++narom; }
            else val2 += 2 * t;
        }
        const int v = s.arom ? (val2 - 3 * narom) / 2 : val2 / 2;
        const int bit = s.arom ? narom * 8 + v : v;
        const int ct = s.c[n0 + i];
        bool stable = false;
        if (at < s.n_types && ct < s.n_charges && bit < 32 && (!s.arom || v < 8))
            stable = (s.table[at * s.n_charges + ct] >> bit) & 1u;
        atomicAdd(&acc[1], 1);
        if (stable) atomicAdd(&acc[0], 1);
    }
    __syncthreads();
    // connected components by min-label propagation (monotone, so in-place updates are safe)
    for (int it = 0; it < n; ++it) {
        if (tid == 0) changed = 0;
        __syncthreads();
        for (int i = tid; i < n; i += 64) {
            int l = label[i];
            if (l < 0) continue;
            for (int j = 0; j < n; ++j) {
                const int lj = label[j];
                if (j != i && lj >= 0 && lj < l && bond(i, j) != 0) l = lj;
            }
            if (l != label[i]) { label[i] = l; changed = 1; }
        }
        __syncthreads();
        const int ch = changed;
        __syncthreads();
        if (!ch) break;
    }
    for (int i = tid; i < n; i += 64) if (label[i] >= 0) atomicAdd(&cnt[label[i]], 1);
    __syncthreads();
    for (int i = tid; i < n; i += 64) {
        if (label[i] == i) atomicAdd(&acc[2], 1);
        if (cnt[i] > 0) atomicMax(&acc[3], cnt[i]);
    }
    __syncthreads();
    if (tid < 4) s.out[m * 4 + tid] = acc[tid];
}
