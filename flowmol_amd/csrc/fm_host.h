// Host-side declarations shared by the translation units of libflowmol_hip.so (round 6: the kernel instances are compiled in several units in parallel --
// fm_engine.cpp: C ABI, weight packing, workspace, the launch sequence and the small kernels; fm_tu_msg32.cpp / fm_tu_msg16.cpp: the edge-message instances;
// fm_tu_node.cpp: the node-kernel instances -- instead of one 78-second unit; flowmol_amd/build.py).  The engine reaches the heavy kernels through the
// plain launcher functions declared at the end of this file, so that no unit instantiates another unit's kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "../../include/flowmol_hip.h"
#include "fm_kernels.h"

namespace fmh {

constexpr int FM_TAB_SLOTS = 32;      // embedding tables kept per bound batch: fm_integrate builds those of up to 32 steps in one launch

extern thread_local std::string g_create_error;      // defined in fm_engine.cpp (fm_last_error(NULL))

struct ProfEvent { int kid; hipEvent_t a, b; };

struct MlpW { const float2* W1; const float* b1; const float2* W2; const float* b2; int K1p, H, O; };

struct ConvW {
    const float2* Wps; const float2* Wpv; const float* w0;
    const float2* Ws_slab = nullptr;   // pair-slab convolutions: [rbf | ef] rows of GVP0's scalar linear (K = 160), multiplied per pair in the SC_EDGE kernel
    const float2* Ws_sh = nullptr;     //                         and its remaining rows, the hidden-vector norms (K = KU0)
    const void* Wps_sp = nullptr;      // split precision
    const void* Wps4 = nullptr;        // quad-row packed (4-node tiles)
    FmGvpW dproj{}; const float2* Wsd = nullptr; const float2* Wpvd = nullptr;     // use_dst_feats: projection GVP + hoisted destination terms
    FmGvpW msg[3]; FmGvpW upd[3];
    const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
};
struct UpdW {
    FmGvpW pos[3];
    const float2* Wasd; const float2* W1; const float* b1; const float2* W2; const float* b2;
    const float *ln_g, *ln_b;
    const void *W1_sp = nullptr, *W2_sp = nullptr, *Wasd_sp = nullptr;      // split-precision builds
    const void* Wasd4 = nullptr;       // quad-row packed (4-node tiles)
};

}  // namespace fmh
using namespace fmh;

struct fm_ctx {
    fm_config cfg{};
    std::string err;
    int V = 32, S = 256, F = 128, na = 0, nc = 0, ne = 0;
    int HX = 0, SD = 0, PVW = 48;     // use_dst_feats: destination vectors / scalars per message; width of the hoisted hidden-vector rows
    // Rows per workgroup tile of the GVP kernels, chosen per bound batch (ws_layout): 32 once the chip is full, 16 while
    // the 32-row tiling would leave CUs idle (fewer tiles than CUs) - half the work per tile, i.e. lower step latency
    // for small batches.  fm_config.tile_edge / tile_node (16|32|64) force a size; tile_edge_update (32|64) for EdgeUpdate.
    int tm_edge = 32, tm_node = 32, tm_eupd = 32;
    int tm_edge_forced = 0, tm_node_forced = 0;
    int n_cus = 256;
    int pair_mlps_forced = -1;      // fm_config.pair_mlps
    int small_mlp_forced = -1;      // fm_config.mlp_small_tiles
    int mlp4_forced = -1;           // fm_config.mlp_small_tiles = 2: the node-side MLPs on 4-row tiles (fm_k_mlp4) whatever the batch; 1 / -1: never
    const void *sc_node_W1q = nullptr, *sc_node_W2q = nullptr, *node_head_W1q = nullptr, *node_head_W2q = nullptr;      // quad-row packed copies for fm_k_mlp4
    int fuse_head = 1;        // the evaluation's last EdgeUpdate also runs the edge output head on its pairs (fm_k_edge_update<32, false, true>; fm_config.fuse_node = 2 | -1: separate)
    int fuse_node = 1;        // node_update also runs the next conv's projections, EdgeUpdate's node terms and NodePositionUpdate (fm_config.fuse_node = -1: separate launches)
    int n_pq = 0;             // leading convolutions (0..2) whose [rbf | ef] slab is computed per unordered pair (self-conditioned models; fm_config.pair_slab = -1: 0)
    int node_rg = 0;          // this batch runs the node kernel on tiles of 4 * node_rg nodes (RG instances; 1, 2, 3 in the 16-row frame, 5 in the 32-row
                              // frame): chosen per bound batch, fm_config.tile_node = 4 / 8 / 12 / 20 forces it
    int pq_forced = 0;        // fm_config.pair_slab = 1: also for batches whose pair tiles do not fill the chip
    // fm_config.canonical >= 0 (default): the ONE launch choice that selects another f32 summation order -- the pair slab (slab + K = 40 chain instead of one K = 200
    // chain) -- is FIXED: computed in every evaluation that can use it, so that a molecule's result does not depend on the size or composition of its batch (see
    // FM_CHUNK_E in fm_kernels.h for the aggregation order).  Tile heights -- incl. the 4 RG-node instances and the 4-row node MLPs, whose GEMMs keep the regular
    // tiles' order since round 6 (fm_wave_gemm4) -- follow the batch size in both modes.  -1: the pair slab follows the batch size too (round 5's rule).
    bool canonical = true;
    float* Q[2] = {nullptr, nullptr};      // (U,256) each, in the workspace
    int xcd_swizzle = 1;      // edge-message tile -> workgroup mapping: contiguous tile range per XCD (fm_config.xcd_swizzle = -1 disables)
    float rbf_mu_step = 0.f, rbf_inv_sigma = 0.f;
    // ---- weights (one device arena)
    char* arena = nullptr; size_t arena_bytes = 0;
    const float *emb_a = nullptr, *emb_c = nullptr;
    MlpW node_embed{}, edge_embed{}, sc_node{}, sc_edge{}, node_head{}, edge_head{};
    const float *node_ln_g = nullptr, *node_ln_b = nullptr, *edge_ln_g = nullptr, *edge_ln_b = nullptr;
    const float *ef_tab = nullptr, *T1 = nullptr;          // (ne+1,128) each
    std::vector<ConvW> conv;
    std::vector<UpdW> upd;
    int tab_rows = 0, tab_kp = 0;
    // ---- batch binding
    bool bound = false;
    int nmax = 0;             // atoms of the largest molecule of the bound batch
    FmBatch b{};
    int n_tiles_e = 0, n_tiles_n = 0, n_tiles_u = 0;
    int n_tiles_msg = 0;      // molecule-aligned edge-message tiles of the bound batch (FmBatch::n_tiles)
    float *s = nullptr, *v = nullptr, *xw = nullptr, *ef = nullptr, *Ps = nullptr, *Asd = nullptr, *PV = nullptr;
    float *part_s = nullptr, *part_v = nullptr, *s_tab = nullptr, *Psd = nullptr, *PVd = nullptr;
    float* s_tab_base = nullptr; size_t tab_slot_floats = 0;      // FM_TAB_SLOTS embedding tables (one per step of a chunk); s_tab = the current step's
    float *tap_s = nullptr, *tap_v = nullptr;    // scratch of the aggregated-message taps (parity runs only)
    fm_dst boot{};
    int32_t *sa1 = nullptr, *sc1 = nullptr, *se1 = nullptr;
    int* mol_gid = nullptr;   // [B] global molecule ids of the Philox noise streams
    // ---- pinned host staging of the per-molecule descriptor arrays (fm_batch_bind / fm_set_molecule_ids: 16 B per molecule).  The copies
    // read it asynchronously; `stage_ev` marks their completion, so the next writer waits for THAT event only (long complete by then) and
    // no entry point ever synchronises the stream.
    int32_t* stage = nullptr; size_t stage_cap = 0; hipEvent_t stage_ev = nullptr; bool stage_busy = false;
    // ---- taps / profiling
    std::map<std::string, void*> taps;
    bool prof = false;
    std::vector<hipEvent_t> ev_pool;          // recycled timing events: creating a pair per launch made the host the bottleneck of a profiled step
    std::vector<ProfEvent> prof_events;
    std::vector<std::string> prof_names;
    std::map<std::string, std::pair<double, int64_t>> prof_acc;
};

namespace fmh {

inline int fail(fm_ctx* c, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

#define FM_HIP(c, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) \
    return fail((c), FM_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

inline bool prec_two_plane(int p) { return p == FM_PREC_BF16X3 || p == FM_PREC_F16X3; }      // the modes whose node / EdgeUpdate kernels run split precision too
inline int pad8(int k) { return (k + 7) / 8 * 8; }
inline int pad16(int k) { return (k + 15) / 16 * 16; }
inline int ld_for(int k) { int ld = (k + 3) / 4 * 4; while (((ld / 4) & 1) == 0) ld += 4; return ld; }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

template <class F> void set_lds(F f, size_t bytes) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(f), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); }

inline int pvw_of(int V, int HX) { return (pad8(V + 1 + HX + 4) + 8 + 15) / 16 * 16; }     // FmGvpTile::PVW
inline size_t lds_gvp_sp(int V, int TM, int npl = 2) {       // split-precision edge message: npl bf16 planes instead of the f32 scalar tile, gates inside Vh
    size_t fl = (size_t)TM * FM_LDP * npl / 2 + 3 * TM * (V + 4) + 3 * TM * (pvw_of(V, 0) + 4);
    return fl * 4 + (size_t)TM * 9 * 4;
}
inline size_t lds_gvp(int V, int TM, bool with_meta, int HX = 0) {
    size_t fl = (size_t)TM * FM_LDX + 3 * TM * (V + 4) + 3 * TM * (pvw_of(V, HX) + 4) + TM * FM_LDG;
    return fl * 4 + (with_meta ? (size_t)TM * 9 * 4 + 64 : 0);      // + one slot for the tile's smallest pair id (PQ instances)
}
inline size_t lds_mlp(int ldx, int ldh, int tm = FM_TM) { return ((size_t)tm * ldx + (size_t)tm * ldh) * 4 + 5 * (size_t)tm * 4; }
inline size_t lds_proj(int V, int tm = FM_TM) { return ((size_t)tm * 260 + 3 * (size_t)tm * (V + 4)) * 4; }
inline size_t lds_edge_upd(int TM) { return ((size_t)TM * 164 + TM * 132) * 4 + TM * 4 * 4 + 16; }
inline size_t lds_edge_upd_sp(int TM) { return (size_t)TM * 132 * 4 + (size_t)TM * 176 * 2 * 2 + TM * 3 * 4; }

// ---------------------------------------------------------------------------------------- launch helper
inline int kid_of(fm_ctx* c, const char* name) {
    for (size_t i = 0; i < c->prof_names.size(); ++i) if (c->prof_names[i] == name) return (int)i;
    c->prof_names.push_back(name);
    return (int)c->prof_names.size() - 1;
}

struct Launch {
    fm_ctx* c; hipStream_t st; int rc = FM_OK;
    template <class K, class... Args>
    void operator()(const char* name, K kernel, dim3 grid, dim3 block, size_t shmem, Args... args) {
        if (rc != FM_OK || grid.x == 0) return;
        ProfEvent pe{};
        if (c->prof) {
            pe.kid = kid_of(c, name);
            auto take = [&](hipEvent_t& e) { if (!c->ev_pool.empty()) { e = c->ev_pool.back(); c->ev_pool.pop_back(); } else (void)hipEventCreate(&e); };
            take(pe.a); take(pe.b);
            (void)hipEventRecord(pe.a, st);
        }
        hipLaunchKernelGGL(kernel, grid, block, shmem, st, args...);
        hipError_t e = hipGetLastError();
        if (c->prof) { (void)hipEventRecord(pe.b, st); c->prof_events.push_back(pe); }
        if (e != hipSuccess) rc = fail(c, FM_ERR_HIP, "launch of %s failed: %s", name, hipGetErrorString(e));
    }
    void copy(void* dst, const void* src, size_t bytes) {
        if (rc != FM_OK || bytes == 0) return;
        hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) rc = fail(c, FM_ERR_HIP, "hipMemcpyAsync failed: %s", hipGetErrorString(e));
    }
    void zero(void* dst, size_t bytes) {
        if (rc != FM_OK || bytes == 0) return;
        hipError_t e = hipMemsetAsync(dst, 0, bytes, st);
        if (e != hipSuccess) rc = fail(c, FM_ERR_HIP, "hipMemsetAsync failed: %s", hipGetErrorString(e));
    }
    void tap(const std::string& name, const void* src, size_t bytes) {
        auto it = c->taps.find(name);
        if (it != c->taps.end()) copy(it->second, src, bytes);
    }
};

// ---------------------------------------------------------------------------------------- launchers of the heavy kernel families (one translation unit each)
// Every launcher selects the instance from run-time parameters and reports an unsupported combination through L.rc; fm_set_lds_* opt the unit's instances
// into their dynamic LDS sizes (called once by fm_create).
void fm_launch_edge_message(Launch& L, int V, int TE, int HX, int precision, bool pq, dim3 grid, const FmMsgArgs& m);      // fm_tu_msg32.cpp / fm_tu_msg16.cpp
void fm_launch_edge_message_v32(Launch& L, int TE, int HX, int precision, bool pq, dim3 grid, const FmMsgArgs& m);
void fm_launch_edge_message_v16(Launch& L, int TE, int HX, int precision, bool pq, dim3 grid, const FmMsgArgs& m);
void fm_set_lds_msg_v32(); void fm_set_lds_msg_v16();
// node kernels (fm_tu_node.cpp): narrow = LayerNorm statistics over a real width < 256; sp = 0 | 1 (bf16x3) | 3 (f16x3); rg = 0 | 1 | 2 | 3 | 5 (4 rg nodes per tile)
void fm_launch_node_update(Launch& L, int V, int TN, bool narrow, int sp, int rg, dim3 grid, size_t lds, const FmNodeUpdArgs& nu);
void fm_launch_pos_update(Launch& L, int V, int TN, dim3 grid, const FmPosArgs& pp);
void fm_launch_dst_proj(Launch& L, int V, int TN, int HX, dim3 grid, const FmDstProjArgs& dp);
void fm_set_lds_node();

}  // namespace fmh
