// Translation unit of the edge-message instances with 16 vector channels (configs/dev.yml family) + the V dispatch; see fm_tu_msg.h.
#include "fm_tu_msg.h"

namespace fmh {
void fm_launch_edge_message_v16(Launch& L, int TE, int HX, int precision, bool pq, dim3 grid, const FmMsgArgs& m) { fm_launch_edge_message_v<16>(L, TE, HX, precision, pq, grid, m); }
void fm_set_lds_msg_v16() { fm_set_lds_msg_v<16>(); }
void fm_launch_edge_message(Launch& L, int V, int TE, int HX, int precision, bool pq, dim3 grid, const FmMsgArgs& m) {
    if (L.rc != FM_OK) return;
    if (V == 32) fm_launch_edge_message_v32(L, TE, HX, precision, pq, grid, m);
    else if (V == 16) fm_launch_edge_message_v16(L, TE, HX, precision, pq, grid, m);
    else L.rc = fail(L.c, FM_ERR_INVALID, "no edge-message instance for %d vector channels", V);
}
}  // namespace fmh
