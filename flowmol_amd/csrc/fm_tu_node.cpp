// Translation unit of the node-kernel instances: fm_k_node_update<V, TN, NARROW, SP, RG> (V = 16 | 32; TN = 16 | 32 | 64 rows; NARROW: LayerNorm statistics over
// a real width < 256; SP = 1 | 3: the split-precision instances (16 / 32 rows); RG = 1 | 2 | 3 (16-row frame) | 5 (32-row frame): 4 RG nodes per workgroup),
// fm_k_pos_update<V, TN> (unfused sequence), fm_k_dst_proj<V, TN, V / 4> (use_dst_feats).  Compiled in parallel with the other units (fm_host.h).
#include "fm_host.h"

namespace fmh {

template <int V>
static void node_update_v(Launch& L, int TN, bool narrow, int sp, int rg, dim3 grid, size_t lds, const FmNodeUpdArgs& nu) {
    const dim3 blk(FM_THREADS);
    const char* nm = "node_update";
    if (sp) {
#define FM_NODE_SP(TN_) if (TN == TN_) { if (sp == 3) L(nm, fm_k_node_update<V, TN_, true, 3>, grid, blk, lds, nu); else L(nm, fm_k_node_update<V, TN_, true, 1>, grid, blk, lds, nu); return; }
        FM_NODE_SP(16) FM_NODE_SP(32)
#undef FM_NODE_SP
    } else if (rg) {      // 4 RG-node instances (fm_wave_gemm4: the regular tiles' summation order)
        if (TN == 16 && rg == 1) { L(nm, fm_k_node_update<V, 16, false, 0, 1>, grid, blk, lds, nu); return; }
        if (TN == 16 && rg == 2) { L(nm, fm_k_node_update<V, 16, false, 0, 2>, grid, blk, lds, nu); return; }
        if (TN == 16 && rg == 3) { L(nm, fm_k_node_update<V, 16, false, 0, 3>, grid, blk, lds, nu); return; }
        if (TN == 32 && rg == 5) { L(nm, fm_k_node_update<V, 32, false, 0, 5>, grid, blk, lds, nu); return; }
    } else {
#define FM_NODE_TN(TN_) if (TN == TN_) { if (narrow) L(nm, fm_k_node_update<V, TN_, true, 0>, grid, blk, lds, nu); else L(nm, fm_k_node_update<V, TN_, false, 0>, grid, blk, lds, nu); return; }
        FM_NODE_TN(16) FM_NODE_TN(32) FM_NODE_TN(64)
#undef FM_NODE_TN
    }
    L.rc = fail(L.c, FM_ERR_INVALID, "no node-update instance for V=%d tile_node=%d split=%d nodes_per_tile=%d", V, TN, sp, 4 * rg);
}

void fm_launch_node_update(Launch& L, int V, int TN, bool narrow, int sp, int rg, dim3 grid, size_t lds, const FmNodeUpdArgs& nu) {
    if (L.rc != FM_OK) return;
    if (V == 32) node_update_v<32>(L, TN, narrow, sp, rg, grid, lds, nu);
    else if (V == 16) node_update_v<16>(L, TN, narrow, sp, rg, grid, lds, nu);
    else L.rc = fail(L.c, FM_ERR_INVALID, "no node-update instance for %d vector channels", V);
}

void fm_launch_pos_update(Launch& L, int V, int TN, dim3 grid, const FmPosArgs& pp) {
    if (L.rc != FM_OK) return;
    const dim3 blk(FM_THREADS);
#define FM_POS(V_, TN_) if (V == V_ && TN == TN_) { L("pos_update", fm_k_pos_update<V_, TN_>, grid, blk, lds_gvp(V_, TN_, false), pp); return; }
    FM_POS(32, 16) FM_POS(32, 32) FM_POS(32, 64) FM_POS(16, 16) FM_POS(16, 32) FM_POS(16, 64)
#undef FM_POS
    L.rc = fail(L.c, FM_ERR_INVALID, "no pos-update instance for V=%d tile_node=%d", V, TN);
}

void fm_launch_dst_proj(Launch& L, int V, int TN, int HX, dim3 grid, const FmDstProjArgs& dp) {
    if (L.rc != FM_OK) return;
    const dim3 blk(FM_THREADS);
#define FM_DSTP(V_, TN_, H_) if (V == V_ && TN == TN_ && HX == H_) { L("dst_proj", fm_k_dst_proj<V_, TN_, H_>, grid, blk, lds_gvp(V_, TN_, false), dp); return; }
    FM_DSTP(16, 16, 4) FM_DSTP(16, 32, 4) FM_DSTP(32, 16, 8) FM_DSTP(32, 32, 8)
#undef FM_DSTP
    L.rc = fail(L.c, FM_ERR_INVALID, "no dst-proj instance for V=%d tile_node=%d dst_vectors=%d", V, TN, HX);
}

void fm_set_lds_node() {
#define FM_SET(V_, T_) set_lds(fm_k_node_update<V_, T_, false, 0>, lds_gvp(V_, T_, false)); set_lds(fm_k_node_update<V_, T_, true, 0>, lds_gvp(V_, T_, false)); set_lds(fm_k_pos_update<V_, T_>, lds_gvp(V_, T_, false));
    FM_SET(32, 16) FM_SET(32, 32) FM_SET(32, 64) FM_SET(16, 16) FM_SET(16, 32) FM_SET(16, 64)
#undef FM_SET
#define FM_SET_SP(V_, T_) set_lds(fm_k_node_update<V_, T_, true, 1>, lds_gvp_sp(V_, T_)); set_lds(fm_k_node_update<V_, T_, true, 3>, lds_gvp_sp(V_, T_));
    FM_SET_SP(32, 16) FM_SET_SP(32, 32) FM_SET_SP(16, 16) FM_SET_SP(16, 32)
#undef FM_SET_SP
#define FM_SET_RG(V_) set_lds(fm_k_node_update<V_, 16, false, 0, 1>, lds_gvp(V_, 16, false)); set_lds(fm_k_node_update<V_, 16, false, 0, 2>, lds_gvp(V_, 16, false)); \
    set_lds(fm_k_node_update<V_, 16, false, 0, 3>, lds_gvp(V_, 16, false)); set_lds(fm_k_node_update<V_, 32, false, 0, 5>, lds_gvp(V_, 32, false));
    FM_SET_RG(32) FM_SET_RG(16)
#undef FM_SET_RG
    set_lds(fm_k_dst_proj<16, 16, 4>, lds_gvp(16, 16, false)); set_lds(fm_k_dst_proj<16, 32, 4>, lds_gvp(16, 32, false));
    set_lds(fm_k_dst_proj<32, 16, 8>, lds_gvp(32, 16, false)); set_lds(fm_k_dst_proj<32, 32, 8>, lds_gvp(32, 32, false));
}

}  // namespace fmh
