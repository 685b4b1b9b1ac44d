// Host-side launch sequencer (no device code): runs a list of this library's grouped launches on one stream in
// one call.  The NMN trunk is ~60 grouped launches forward and ~100 backward per step; issued one by one
// through the Python binding they cost 5-8 us of host time each, and at 128 questions per GPU the step is
// bound by exactly that host time.  The list is built by the scheduler (probnmn/runtime/engine.py) from the
// same work-item records; each entry is one call of the entry point named by `op` with the arguments it
// would have received -- nothing else changes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/probnmn_hip.h"

extern "C" int pnmn_run_launches(const pnmn_launch* list, int n, void* stream) {
    if (n <= 0) return 0;
    if (!list) return PNMN_EINVAL;
    for (int i = 0; i < n; ++i) {
        const pnmn_launch& l = list[i];
        const int32_t* p = l.p;
        int rc;
        switch (l.op) {
            case PNMN_OP_CONV:
                rc = pnmn_conv_nhwc_cus(static_cast<const pnmn_conv_item*>(l.a), l.n, p[0], p[1], p[2], p[3], p[4], p[5], p[6],
                                        p[7], (int)reinterpret_cast<uintptr_t>(l.c), stream);
                break;
            case PNMN_OP_WGRAD:
                rc = pnmn_conv_wgrad_cus(static_cast<const pnmn_wgrad_item*>(l.a), static_cast<const pnmn_wgrad_job*>(l.b), l.n,
                                         p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], stream);
                break;
            case PNMN_OP_TRANSPOSE_WEIGHTS:
                rc = pnmn_transpose_weights(static_cast<const pnmn_wtrans_item*>(l.a), l.n, stream);
                break;
            case PNMN_OP_DOT_FWD:
                rc = pnmn_dot1_sigmoid_fwd(static_cast<const pnmn_dot1_item*>(l.a), l.n, p[0], stream);
                break;
            case PNMN_OP_DOT_BWD:
                rc = pnmn_dot1_sigmoid_bwd(static_cast<const pnmn_dot1_item*>(l.a), l.n, p[0], stream);
                break;
            case PNMN_OP_SAME_FWD:
                rc = pnmn_same_fwd(static_cast<const pnmn_same_item*>(l.a), l.n, p[0], stream);
                break;
            case PNMN_OP_SAME_BWD:
                rc = pnmn_same_bwd(static_cast<const pnmn_same_item*>(l.a), l.n, p[0], stream);
                break;
            case PNMN_OP_MINMAX_FWD:
                rc = pnmn_minmax_fwd(static_cast<const pnmn_minmax_item*>(l.a), l.n, p[0], p[1], stream);
                break;
            case PNMN_OP_MINMAX_BWD:
                rc = pnmn_minmax_bwd(static_cast<const pnmn_minmax_item*>(l.a), l.n, p[0], p[1], stream);
                break;
            case PNMN_OP_MASK_BWD:
                rc = pnmn_mask_bwd(static_cast<const pnmn_maskbwd_item*>(l.a), l.n, p[0], stream);
                break;
            case PNMN_OP_MAXPOOL_FWD:
                rc = pnmn_maxpool2_flatten_fwd(static_cast<const float*>(l.a), static_cast<float*>(const_cast<void*>(l.b)), l.n,
                                               p[0], p[1], p[2], stream);
                break;
            case PNMN_OP_MAXPOOL_BWD:
                rc = pnmn_maxpool2_flatten_bwd(static_cast<const float*>(l.a), static_cast<const float*>(l.b),
                                               static_cast<float*>(const_cast<void*>(l.c)), l.n, p[0], p[1], p[2], stream);
                break;
            case PNMN_OP_NCHW_TO_NHWC:
                rc = l.c ? pnmn_nchw_to_nhwc_rows(static_cast<const float*>(l.a), static_cast<float*>(const_cast<void*>(l.b)),
                                                  static_cast<const int64_t*>(l.c), l.n, p[0], p[1], stream)
                         : pnmn_nchw_to_nhwc(static_cast<const float*>(l.a), static_cast<float*>(const_cast<void*>(l.b)), l.n,
                                             p[0], p[1], stream);
                break;
            case PNMN_OP_SET_ROWS:
                rc = pnmn_set_rows(static_cast<const pnmn_axpy_item*>(l.a), l.n, stream);
                break;
            case PNMN_OP_ACCUMULATE:
                rc = pnmn_accumulate(static_cast<const pnmn_axpy_item*>(l.a), l.n, stream);
                break;
            case PNMN_OP_FEAT_GATHER:
                rc = pnmn_feat_grad_gather(static_cast<const pnmn_maskbwd_item*>(l.a),
                                           static_cast<float*>(const_cast<void*>(l.b)), l.n, p[0], p[1], stream);
                break;

            case PNMN_OP_ZERO:
                rc = (int)hipMemsetAsync(const_cast<void*>(l.a), 0, (size_t)reinterpret_cast<uintptr_t>(l.b),
                                         static_cast<hipStream_t>(stream));
                break;
            default:
                return PNMN_EINVAL;
        }
        if (rc != 0) return rc;
    }
    return 0;
}
