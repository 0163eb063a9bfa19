// The edge-message instances of ONE vector width (FM_TU_V = 32 | 16): included by fm_tu_msg32.cpp / fm_tu_msg16.cpp, which are compiled in parallel with the
// other translation units of the library (fm_host.h).  fm_k_edge_message<V, TE, 512, HX, SP, PQ>: TE = 16 | 32 | 64 rows; SP = 0 f32, 1 bf16x3, 2 bf16x6
// (16 / 32 rows), 3 f16x3; PQ = 1 the pair-slab instance (f32); HX = V / 4 destination-feature vectors (f32, 16 / 32 rows).
#pragma once
#include "fm_host.h"

namespace fmh {

template <int V>
void fm_launch_edge_message_v(Launch& L, int TE, int HX, int precision, bool pq, dim3 grid, const FmMsgArgs& m) {
    constexpr int HXV = V / 4;
    const dim3 blk(512);
#define FM_MSG_TE(TE_)                                                                                                                                          \
    if (TE == TE_) {                                                                                                                                            \
        if (HX == 0) {                                                                                                                                          \
            if (precision == FM_PREC_BF16X3) { L("edge_message", fm_k_edge_message<V, TE_, 512, 0, 1>, grid, blk, lds_gvp_sp(V, TE_), m); return; }              \
            if (precision == FM_PREC_F16X3) { L("edge_message", fm_k_edge_message<V, TE_, 512, 0, 3>, grid, blk, lds_gvp_sp(V, TE_), m); return; }               \
            if (precision == FM_PREC_BF16X6) {                                                                                                                  \
                if constexpr (TE_ <= 32) { L("edge_message", fm_k_edge_message<V, TE_, 512, 0, 2>, grid, blk, lds_gvp_sp(V, TE_, 3), m); return; }               \
                else { L.rc = fail(L.c, FM_ERR_INVALID, "the three-term split precision runs 16- or 32-row edge tiles (three planes of a 64-row tile exceed the LDS)"); return; } \
            }                                                                                                                                                   \
            if (pq) { L("edge_message_pq", fm_k_edge_message<V, TE_, 512, 0, 0, 1>, grid, blk, lds_gvp(V, TE_, true, 0), m); return; }                           \
            L("edge_message", fm_k_edge_message<V, TE_, 512, 0, 0>, grid, blk, lds_gvp(V, TE_, true, 0), m); return;                                            \
        }                                                                                                                                                       \
        if constexpr (TE_ <= 32) { if (HX == HXV && precision == FM_PREC_F32) { L("edge_message", fm_k_edge_message<V, TE_, 512, HXV, 0>, grid, blk, lds_gvp(V, TE_, true, HXV), m); return; } } \
    }
    FM_MSG_TE(16) FM_MSG_TE(32) FM_MSG_TE(64)
#undef FM_MSG_TE
    if (L.rc == FM_OK) L.rc = fail(L.c, FM_ERR_INVALID, "no edge-message instance for V=%d tile_edge=%d dst_vectors=%d precision=%d", V, TE, HX, precision);
}

template <int V>
void fm_set_lds_msg_v() {
    constexpr int HXV = V / 4;
#define FM_MSG_SET(T_) set_lds(fm_k_edge_message<V, T_, 512, 0, 0>, lds_gvp(V, T_, true)); set_lds(fm_k_edge_message<V, T_, 512, 0, 0, 1>, lds_gvp(V, T_, true)); \
    set_lds(fm_k_edge_message<V, T_, 512, 0, 1>, lds_gvp_sp(V, T_)); set_lds(fm_k_edge_message<V, T_, 512, 0, 3>, lds_gvp_sp(V, T_));
    FM_MSG_SET(16) FM_MSG_SET(32) FM_MSG_SET(64)
#undef FM_MSG_SET
    set_lds(fm_k_edge_message<V, 16, 512, 0, 2>, lds_gvp_sp(V, 16, 3)); set_lds(fm_k_edge_message<V, 32, 512, 0, 2>, lds_gvp_sp(V, 32, 3));
    set_lds(fm_k_edge_message<V, 16, 512, HXV, 0>, lds_gvp(V, 16, true, HXV)); set_lds(fm_k_edge_message<V, 32, 512, HXV, 0>, lds_gvp(V, 32, true, HXV));
}

}  // namespace fmh
