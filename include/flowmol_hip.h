/* C ABI of libflowmol_hip.so -- the MI355X-native replacement for the FlowMol3 sampling hot path.
 *
 * The reference (Dunni3/FlowMol) is pure Python and has NO FFI; this header declares the boundary a
 * maintainer would bind with ctypes (INTEGRATION.md shows the stub).  Each entry point cites the
 * reference code it replaces (paths are into the reference tree):
 *
 *   fm_create           weights of flowmol/models/ctmc_vector_field.py:CTMCVectorField (state-dict keys
 *                       "vector_field.*"), repacked for MFMA
 *   fm_batch_bind       graph batching: flowmol/models/flowmol.py:509-529 (dgl.graph/dgl.batch),
 *                       flowmol/data_processing/utils.py:4-46 (edge list, upper-edge mask, batch idxs)
 *   fm_forward          EndpointVectorField.forward, flowmol/models/vector_field.py:212-369
 *                       (incl. GVPConv gvp.py:435-543, NodePositionUpdate/EdgeUpdate vector_field.py:813-880,
 *                       SelfConditioningResidualLayer self_conditioning.py:37-85)
 *   fm_ctmc_step        the part of CTMCVectorField.step after the network evaluation,
 *                       ctmc_vector_field.py:328-411 + campbell_step :414-461 + purity_sampling ctmc_utils.py:4-34
 *   fm_integrate        CTMCVectorField.integrate, ctmc_vector_field.py:145-285
 *
 * Conventions
 *   - every function returns 0 on success or a negative fm_status; the message is available from
 *     fm_last_error(ctx) (ctx may be NULL for errors of fm_create).  Nothing throws across the ABI.
 *   - all tensor arguments are DEVICE pointers to caller-owned memory (torch tensors passed as
 *     data_ptr()); the library never frees caller memory.  After fm_create the library allocates
 *     no DEVICE memory: per-batch scratch lives in the caller-provided workspace of fm_batch_bind.
 *     (Host side: one pinned staging buffer of 16 bytes per molecule for the descriptor arrays of
 *     fm_batch_bind / fm_set_molecule_ids, grown on demand.)
 *   - kernels are enqueued on the hipStream_t passed in (as void*); no call synchronises the stream
 *     or the device.  fm_batch_bind / fm_set_molecule_ids copy their host arrays through the pinned
 *     staging buffer and wait (on an event) only for the copies of the PREVIOUS such call, which
 *     have completed long before in any realistic call sequence; they are setup calls and, because
 *     of that event wait, must stay outside a stream capture.
 *   - the library reads no environment variables; every switch is a field of fm_config.
 *   - a context is bound to the device current at fm_create and is not thread-safe.
 *   - categorical state is exchanged as int32 token indices (mask token = number of real categories);
 *     edge state is per UNORDERED pair in the reference's upper-triangle order
 *     (torch.triu_indices(n,n,1) row-major per molecule, molecules concatenated) -- both directed
 *     edges of a pair always carry the same token in the reference (ctmc_vector_field.py:397-409).
 *   - probabilities ("dst") are float32: x (N,3), a (N,n_atom_types), c (N,n_charges), e (U,n_bond_types).
 */
#ifndef FLOWMOL_HIP_H
#define FLOWMOL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FM_ABI_VERSION 7
#define FM_MAX_CONVS 16

typedef enum fm_status {
    FM_OK = 0,
    FM_ERR_INVALID = -1,      /* bad argument / unsupported configuration */
    FM_ERR_WEIGHTS = -2,      /* missing tensor or wrong shape */
    FM_ERR_HIP = -3,          /* a HIP runtime call or kernel launch failed */
    FM_ERR_STATE = -4,        /* call order (no batch bound, ...) */
    FM_ERR_NOMEM = -5
} fm_status;

typedef struct fm_ctx fm_ctx;

/* mirrors flowmol_amd.config.VFConfig (the reference's vector_field: YAML block) */
typedef struct fm_config {
    int32_t abi_version;          /* FM_ABI_VERSION */
    int32_t n_atom_types;         /* real atom categories incl. the fake-atom type, excl. mask */
    int32_t n_charges;
    int32_t n_bond_types;
    int32_t n_vec_channels;       /* 16 or 32 */
    int32_t n_hidden_scalars;     /* power of two, 8..256 (flowmol3: 256, configs/dev.yml: 64); narrower models run on zero-padded 256-column tiles */
    int32_t n_hidden_edge_feats;  /* power of two, 8..128 (flowmol3: 128, dev.yml: 64) */
    int32_t rbf_dim;              /* 32 */
    int32_t n_convs;
    int32_t n_updaters;
    int32_t update_after[FM_MAX_CONVS];   /* updater index run after conv i, or -1 (vector_field.py:320-326) */
    int32_t self_conditioning;
    int32_t time_embedding_dim;   /* 1 = raw t */
    int32_t a_token_dim, c_token_dim, e_token_dim;   /* 0 = one-hot input */
    float rbf_dmax;
    float msg_z;                  /* divisor of the aggregated messages (1 for message_norm 'sum'); < 0: message_norm 'mean' -- every node's
                                   * sum is divided by its in-degree n_i - 1 (DGL fn.mean, gvp.py:401-404; 0 for a 1-atom molecule) */
    /* --- ABI 3: use_dst_feats (flowmol/models/gvp.py:300-316,472-473,527-537): widths of the projected destination-node
     * features that join every message's inputs; 0 / 0 = off */
    int32_t s_dst_feats;          /* int(n_hidden_scalars / dst_feat_msg_reduction_factor) */
    int32_t v_dst_feats;          /* int(n_vec_channels / dst_feat_msg_reduction_factor), <= 8 */
    /* --- ABI 3: 1 = CTMC model (categorical inputs are tokens with a mask state, CTMCVectorField); 0 = endpoint-parameterised model
     * (EndpointVectorField, flowmol/models/vector_field.py:15-211: categorical inputs are continuous (rows, n) vectors, token dims 0) */
    int32_t has_mask;
    /* --- ABI 3: arithmetic of the edge-message GEMMs.  FM_PREC_F32 (default): exact f32 MFMAs, the reference's arithmetic.
     * FM_PREC_BF16X3: OPT-IN split precision -- the scalar and gate GEMMs of GVPConv.message run on the bf16 matrix cores with every
     * f32 operand carried as hi + lo bf16 and three products per term (~2^-17 relative per product instead of 2^-24).  Not f32
     * arithmetic: per-stage errors are ~10x larger; meant for throughput runs, never for parity claims.
     * FM_PREC_BF16X6 (round 5, OPT-IN, edge-message kernel only; node kernels and EdgeUpdate stay f32): three-term split -- hi + mid + lo bf16 = all 24
     * mantissa bits, six products per term, dropped products <= 3 * 2^-24 relative: f32-class accuracy on the bf16 matrix cores at the price of 6-byte
     * operands (weight stream, LDS).  Measured (profiles/r05c_*): every stage within 1.32x the f32 kernels' error against float64,
     * 71 molecules/s at C3 (f32: 63): three planes leave room for ONE workgroup per CU.  A separately reported mode like FM_PREC_BF16X3.
     * FM_PREC_F16X3 (round 5, OPT-IN): FM_PREC_BF16X3's kernels with IEEE-half planes -- hi + lo half = 22 of f32's 24 mantissa bits in the same 4 bytes per
     * element and the same three products (v_mfma_f32_16x16x32_f16).  RANGE is the price: |activations| beyond 65504 are clamped (a finite, wrong result). */
    int32_t precision;
    /* --- ABI 4: remaining architecture switches of EndpointVectorField.__init__ that no shipped YAML enables */
    int32_t n_recycles;           /* vector_field.py:307: the conv / update stack runs n_recycles times over the same weights (0 or 1 = once) */
    int32_t edge_update_no_distance;   /* 1 = update_edge_w_distance False: EdgeUpdate's first Linear has no rbf(d) columns (vector_field.py:851-853,876-877) */
    /* --- ABI 5: launch-tuning overrides -- 0 = automatic for every one of them (what the product always passes).  They exist for the
     * A/B measurements under profiles/ and for the parity tests that run every tile size; up to ABI 4 they were environment variables
     * read inside fm_create, which hid them from the interface.  The library reads NO environment variable. */
    int32_t tile_edge;            /* rows per workgroup tile of the edge kernels: 0 = per batch (the cheaper of 16 / 32 by a rounds-per-CU model: 16 for a few molecules, 32 for large batches; same bits) | 16 | 32 | 64 */
    int32_t tile_node;            /* same for the node kernels; additionally 4 | 8 | 12 (nodes per workgroup in the 16-row frame) | 20 (in the 32-row frame), scalar GEMMs
                                   * on v_mfma_f32_4x4x1: the automatic choice is the smallest of 4 / 8 / 12 / 16 / 20 whose tiles fit one per CU */
    int32_t tile_edge_update;     /* EdgeUpdate tile: 0 = 32 | 32 | 64 */
    int32_t xcd_swizzle;          /* edge-message tile -> workgroup map: 0 / 1 = one contiguous tile range per XCD | -1 = identity */
    int32_t fuse_node;            /* 0 / 1 = node_update also runs the next conv's projections + NodePositionUpdate, and the last EdgeUpdate the edge output head
                                     | 2 = the node fusion only (edge head as a kernel of its own) | -1 = separate launches */
    int32_t pair_mlps;            /* node-side and pair-side MLPs of a stage in ONE launch: 0 = while the pair tiles do not fill the chip | 1 always | -1 never */
    int32_t mlp_small_tiles;      /* 16-row tiles for the MLP kernels: 0 = while they do not fill the chip (and 4-row NODE tiles, fm_k_mlp4, while those fit one per CU)
                                     | 1 always 16 rows | -1 never | 2 = 16-row pair tiles + 4-row node tiles whatever the batch */
    /* --- ABI 6 */
    int32_t pair_slab;            /* [rbf | ef] slab of the first edge GVP once per unordered pair for the convolutions before the first molecule
                                   * update (pair-symmetric inputs; self-conditioned f32 models without destination features): 0 = in evaluations
                                   * whose pair tiles fill the chip (canonical mode: in every one) | 1 = in every self-conditioned evaluation | -1 = off */
    /* --- ABI 7: canonical arithmetic.  In the reference a molecule's result is a function of the molecule and its noise rows only -- every reduction is per
     * molecule (flowmol/models/gvp.py:491-492, flowmol/utils/ctmc_utils.py:11-20, flowmol/models/vector_field.py:347-350).  0 / 1 (default): the library gives the
     * same guarantee BIT FOR BIT: the f32 summation order of everything computed for a molecule depends on the molecule alone -- edge-message tiles start at the
     * molecule's first edge row and in-edges are summed in 16-row chunks counted from it, LayerNorm statistics and gate sums have one order for every tile
     * height, the 4-row instances of the node kernels (small batches) run the regular tiles' fma chains, and the one launch choice that selects another order
     * (pair slab on / off) is fixed instead of following the batch size.  A molecule alone, inside a 1024-batch, in any shard of it and on 1 or 8 GPUs gives
     * identical coordinates and tokens for identical noise.  Holds for every automatic tile height and across explicit tile_edge 16 | 32, tile_node 4 .. 32,
     * mlp_small_tiles; 64-row tiles, pair_slab -1 and fuse_node 2 | -1 (the edge head as a kernel of its own, also taken by batches with a molecule of >= 2048
     * atoms) are other (self-consistent) orders.
     * -1: the pair slab follows the batch size (round 5's rule); results then agree between differently composed batches to f32 summation order only.  (Up to
     * round 5 this was a "latency mode" with its own 4-row kernels; since the 4-row kernels are canonical it gains nothing measurable: one molecule 0.55 vs 0.56 ms.) */
    int32_t canonical;
} fm_config;

enum fm_precision { FM_PREC_F32 = 0, FM_PREC_BF16X3 = 1, FM_PREC_BF16X6 = 2, FM_PREC_F16X3 = 3 };

/* one tensor of the reference state dict inside the host weight blob */
typedef struct fm_tensor_desc {
    const char* name;             /* reference state-dict key without the "vector_field." prefix */
    int64_t offset;               /* in floats from the start of the blob */
    int32_t ndim;
    int64_t shape[2];
} fm_tensor_desc;

typedef struct fm_dst {           /* endpoint prediction ("dst_dict" of the reference) */
    float* x;                     /* (N,3) */
    float* a;                     /* (N,n_atom_types) probabilities */
    float* c;                     /* (N,n_charges) */
    float* e;                     /* (U,n_bond_types) */
} fm_dst;

typedef struct fm_state {         /* g.ndata['x_t','a_t','c_t'], g.edata['e_t'] of the reference */
    float* x_t;                   /* (N,3) */
    int32_t* a_t;                 /* (N) */
    int32_t* c_t;                 /* (N) */
    int32_t* e_t;                 /* (U) */
} fm_state;

typedef struct fm_dense_state {   /* endpoint-parameterised models: g.ndata['x_t','a_t','c_t'], g.edata['e_t'][upper] as float features */
    float* x_t;                   /* (N,3) */
    float* a_t;                   /* (N,n_atom_types) */
    float* c_t;                   /* (N,n_charges) */
    float* e_t;                   /* (U,n_bond_types), per unordered pair */
} fm_dense_state;

typedef struct fm_endpoint_scalars {   /* EndpointVectorField.step, vector_field.py:501-564, host-computed in float32 */
    float dt;                     /* s_i - t_i */
    float coef[4];                /* x, a, c, e: alpha'/(1 - alpha) */
    float scale;                  /* inv_temp_func(t_i): 1, or continuous_inv_temp_max * (1 - t_i) */
} fm_endpoint_scalars;

typedef struct fm_step_noise {    /* draws of one CTMC step, in the reference's order and shapes */
    const float* q_a;  const float* u1_a;  const float* u2_a;    /* (N,na) Exp(1), (N) U, (N) U */
    const float* q_c;  const float* u1_c;  const float* u2_c;
    const float* q_e;  const float* u1_e;  const float* u2_e;    /* (U,ne), (U), (U) */
    /* dfm_type FM_DFM_GAT: one Categorical draw per modality over K+1 classes (mask included), so q_* are
     * (rows, K+1) and u1_*, u2_* are unused (may be NULL) */
} fm_step_noise;

enum fm_dfm_type { FM_DFM_CAMPBELL = 0, FM_DFM_GAT = 1 };   /* reference ctmc_vector_field.py:357 / :373 */

typedef struct fm_step_scalars {  /* host-computed with the reference's float32 arithmetic */
    float t;                      /* t_i */
    float dt;                     /* s_i - t_i */
    float x_coef;                 /* alpha'_x / (1 - alpha_x) */
    float unmask_prob[3];         /* a, c, e: clamp(dt*(alpha' + eta*alpha)/(1-alpha), 0, 1) */
    float mask_prob[3];           /* clamp(dt*eta, 0, 1) */
    float hc_thresh;              /* purity threshold; 0 = uniform unmasking branch */
    float cat_temperature;        /* cat_temp_func(t_i): 0.05 by default, or the 'decay' schedule */
    int32_t last_step;
    /* --- ABI 2 */
    float x_scale;                /* inv_temp_func(t_i) of the reference's step(): x_t += (dt*vf)*x_scale; 1 by default */
    int32_t dfm_type;             /* fm_dfm_type */
    float gat_cf[3];              /* a, c, e: alpha'/(1 - alpha)        (gat_step, ctmc_vector_field.py:481) */
    float gat_cb[3];              /* a, c, e: alpha'/(alpha + 1e-8)     (:488) */
    float gat_fw, gat_bw;         /* forward_weight_func(t_i) and forward_weight - 1 (:491-492) */
    /* --- ABI 3: noise source of this step (campbell only).  FM_NOISE_TENSORS: the caller's fm_step_noise (the reference's draws,
     * bit-exact parity mode).  FM_NOISE_PHILOX: drawn inside the kernel from Philox4x32-10 streams keyed by (philox_seed, global
     * molecule id, step_index, modality, row) -- no noise tensors, and a molecule's trajectory does not depend on how the
     * batch is sharded; fm_step_noise may then be NULL */
    int32_t noise_mode;
    int32_t step_index;
    uint32_t philox_seed_lo, philox_seed_hi;
} fm_step_scalars;

enum fm_noise_mode { FM_NOISE_TENSORS = 0, FM_NOISE_PHILOX = 1 };

typedef struct fm_sampled {       /* sampled endpoint tokens of a step ("*_1_pred"), optional (may be NULL) */
    int32_t* a1; int32_t* c1; int32_t* e1;
} fm_sampled;

typedef struct fm_traj_sink {     /* optional per-step frames (xt_traj / ep_traj); any pointer may be NULL */
    float* x;  int32_t* a;  int32_t* c;  int32_t* e;             /* (n_steps, ...) state after each step */
    float* x1; int32_t* a1; int32_t* c1; int32_t* e1;            /* (n_steps, ...) endpoint predictions */
} fm_traj_sink;

const char* fm_last_error(const fm_ctx* ctx);
int fm_abi_version(void);

int fm_create(const fm_config* cfg, const fm_tensor_desc* tensors, int n_tensors, const float* host_blob,
              fm_ctx** out);
int fm_destroy(fm_ctx* ctx);

/* bytes of caller-provided device workspace needed for a batch of molecules with the given sizes */
int fm_workspace_bytes(fm_ctx* ctx, const int32_t* n_atoms_host, int n_mols, size_t* bytes);
/* bind a batch: builds the internal destination-sorted edge layout inside `workspace` */
int fm_batch_bind(fm_ctx* ctx, void* stream, const int32_t* n_atoms_host, int n_mols, void* workspace, size_t bytes);

/* x -= per-molecule mean, in place, for the bound batch: the centring step of the position prior
 * (centered_normal_prior_batched_graph, flowmol/data_processing/priors.py:27-35) */
int fm_remove_com(fm_ctx* ctx, void* stream, float* x);

/* Philox mode: global ids of the bound batch's molecules (host array of n_mols; NULL = 0..n_mols-1, the default after
 * fm_batch_bind), and the position prior x0 ~ N(0, I) - per-molecule mean (priors.py:27-35) drawn from the same streams */
int fm_set_molecule_ids(fm_ctx* ctx, void* stream, const int32_t* ids_host);
int fm_prior_philox(fm_ctx* ctx, void* stream, uint64_t seed, float* x0);

/* one network evaluation.  temb: device (time_embedding_dim) floats (raw t when dim == 1).
 * prev: previous endpoint for self-conditioning or NULL.  bootstrap != 0 reproduces the reference's
 * first-step behaviour (vector_field.py:269-282): an extra evaluation with remove_com=False whose
 * result is used as `prev`.  out.a/c/e receive softmax probabilities, out.x is COM-free iff remove_com. */
int fm_forward(fm_ctx* ctx, void* stream, const fm_state* state, const float* temb, const fm_dst* prev,
               int bootstrap, int remove_com, const fm_dst* out);

/* Endpoint-parameterised models (has_mask == 0): one network evaluation on continuous categorical features (no self-conditioning),
 * and the Euler step of all four modalities  x_s = x_t + ((coef * (x_1 - x_t)) * scale) * dt  given the endpoint prediction `dst`
 * (EndpointVectorField.forward / .step, flowmol/models/vector_field.py:212-293, 501-564). */
int fm_forward_dense(fm_ctx* ctx, void* stream, const fm_dense_state* state, const float* temb, int remove_com, const fm_dst* out);
int fm_endpoint_step(fm_ctx* ctx, void* stream, const fm_dense_state* state, const fm_dst* dst, const fm_endpoint_scalars* sc);

/* Euler step for x and the CTMC update of a, c, e given the endpoint prediction `dst` */
int fm_ctmc_step(fm_ctx* ctx, void* stream, const fm_state* state, const fm_dst* dst,
                 const fm_step_noise* noise, const fm_step_scalars* sc, const fm_sampled* sampled);

/* n_steps Euler/CTMC steps: per step fm_forward (+bootstrap on step 0 for self-conditioned models when
 * steps[0].t == 0) then fm_ctmc_step.  temb: device (n_steps, time_embedding_dim); noise: host array of
 * n_steps fm_step_noise (device pointers); prev0: endpoint of the step before the first one (NULL at the
 * start of a trajectory; dst_a or dst_b when a trajectory is integrated in several calls);
 * dst_a/dst_b: two caller-provided endpoint buffers used alternately (the one holding the final
 * prediction is returned in *final_dst, 0 or 1). */
int fm_integrate(fm_ctx* ctx, void* stream, const fm_state* state, int n_steps, const fm_step_scalars* steps,
                 const float* temb, const fm_step_noise* noise, const fm_dst* prev0, const fm_dst* dst_a,
                 const fm_dst* dst_b, const fm_traj_sink* sink, int* final_dst);

/* debugging / parity taps: copy an internal buffer to `dst` (device) after the next fm_forward stages.
 * name: "embed.s", "sc.s", "sc.ef", "conv<i>.s", "conv<i>.v", "conv<i>.agg.s", "conv<i>.agg.v",
 * "upd<i>.x", "upd<i>.ef", "conv0.msg.s", "conv0.msg.v"; internal layouts: v (N,3,V), ef (E,128) in the
 * internal edge order.  fm_batch_query copies an int32 descriptor array ("e_src","e_dst","e_pair",
 * "p_e0","p_e1","node_mol") to `dst`. */
int fm_set_tap(fm_ctx* ctx, const char* name, void* dst);
int fm_clear_taps(fm_ctx* ctx);
int fm_batch_query(fm_ctx* ctx, void* stream, const char* name, int32_t* dst);

/* Valence stability and connectivity of the molecules in `state` (tokens), computed on the device
 * (reference flowmol/analysis/metrics.py:96-117 + check_stability :333-363; SampledMolecule.compute_valencies,
 * molecule_builder.py:138-157).  table: device array of n_types * n_charges uint32 bit masks (bit v = valency v valid;
 * explicit aromaticity: bit n_arom*8+v); fake_atom_token: token of the fake atom or -1; out: device (n_mols,4) int32 =
 * {stable atoms, real atoms, connected components, largest component}. */
int fm_stability(fm_ctx* ctx, void* stream, const fm_state* state, const uint32_t* table, int n_types,
                 int fake_atom_token, int explicit_aromaticity, int32_t* out);

/* per-kernel timing of the last fm_forward / fm_integrate when enabled (HIP events on `stream`):
 * fm_profile_enable(ctx, 1); ... ; fm_profile_get(ctx, "edge_message", &total_ms, &launches).
 * While profiling, every CTMC step also times an EMPTY kernel under the name "event_overhead": what an event pair adds to a launch
 * (subtract its average from the other kernels' averages to compare with rocprofv3 kernel durations). */
int fm_profile_enable(fm_ctx* ctx, int on);
int fm_profile_get(fm_ctx* ctx, const char* kernel, double* total_ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* FLOWMOL_HIP_H */
