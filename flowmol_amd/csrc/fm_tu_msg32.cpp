// Translation unit of the edge-message instances with 32 vector channels (flowmol3, geom, qm9 models); see fm_tu_msg.h.
#include "fm_tu_msg.h"

namespace fmh {
void fm_launch_edge_message_v32(Launch& L, int TE, int HX, int precision, bool pq, dim3 grid, const FmMsgArgs& m) { fm_launch_edge_message_v<32>(L, TE, HX, precision, pq, grid, m); }
void fm_set_lds_msg_v32() { fm_set_lds_msg_v<32>(); }
}  // namespace fmh
