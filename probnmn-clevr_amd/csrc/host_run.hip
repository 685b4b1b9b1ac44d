// Host-side launch sequencer (no device code): runs a list of this library's grouped launches on one stream in
// one call.  The NMN trunk is ~60 grouped launches forward and ~100 backward per step; issued one by one
// through the Python binding they cost 5-8 us of host time each, and at 128 questions per GPU the step is
// bound by exactly that host time.  The list is built by the scheduler (probnmn/runtime/engine.py) from the
// same work-item records; each entry is one call of the entry point named by `op` with the arguments it
// would have received -- nothing else changes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "../../include/probnmn_hip.h"
#include "conv_plan.h"

// ---- launch trace (pnmn_launch_trace_begin / _end): events around the CONV / WGRAD rows of every list run in between ----
namespace {

struct Traced {
    pnmn_launch row;
    hipEvent_t e0, e1;
    void* records;  // page-locked copy of the call's items (CONV) / jobs (WGRAD), taken in stream order in front of the launch:
                    // by the time the trace is collected the device buffer may hold the next upload
};
struct Trace {
    std::mutex mu;
    std::atomic<bool> on{false};  // (read by pnmn_run_launches without the mutex)
    std::vector<Traced> rows;
    std::vector<hipEvent_t> pool;  // events are kept from trace to trace
    // page-locked arena the record copies are cut from (one allocation, kept: a hipHostMalloc per traced launch cost the
    // host ~0.1 ms each and the launches behind it started on an idling chip)
    char* arena = nullptr;
    size_t arena_size = 0, arena_used = 0;
    void* cut(size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        if (!arena) {
            arena_size = (size_t)64 << 20;
            if (hipHostMalloc(reinterpret_cast<void**>(&arena), arena_size, hipHostMallocDefault) != hipSuccess) arena = nullptr;
        }
        if (!arena || arena_used + bytes > arena_size) return nullptr;
        void* p = arena + arena_used;
        arena_used += bytes;
        return p;
    }
    hipEvent_t event() {
        if (!pool.empty()) {
            hipEvent_t e = pool.back();
            pool.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        return hipEventCreate(&e) == hipSuccess ? e : nullptr;
    }
};
Trace& trace() {
    static Trace t;
    return t;
}

// (tap, m-tile) pairs of the 117 a 3x3 convolution on a 14x14 map does not contract (conv_stream.h, KIND 1): work that
// is not done is not counted as done
inline double skipped_tap_tiles(int dilation) { return dilation == 8 ? 39.0 : 0.0; }

// Algorithmic work of one grouped convolution call: FLOPs = 2 * pixels * Cout * taps * Cin per item,
// less the skipped tap rows; bytes = every map an item must read (input chunks, the ReLU gate of a data gradient, the
// attention mask, the forward features and the accumulated gradient of the fused mask backward, the previous contents of
// an accumulating output) and write once, plus ONE pass over each distinct weight of the call.
int conv_work(const pnmn_launch& l, const void* records, pnmn_launch_timing* t) {
    const int H = l.p[0], W = l.p[1], cin_chunks = l.p[2], ntaps = l.p[3], cout_blocks = l.p[6];
    const double HW = (double)H * W, C = 128.0;
    const pnmn_conv_item* first = static_cast<const pnmn_conv_item*>(records);
    const std::vector<pnmn_conv_item> items(first, first + l.n);
    const double full = 2.0 * HW * cout_blocks * C * ntaps * cin_chunks * C;
    const double map = HW * C * 4.0, wbytes = cout_blocks * C * ntaps * cin_chunks * C * 4.0;
    double flops = 0.0, maps = 0.0, extra = 0.0;
    std::vector<const float*> weights;
    // the kernel skips the out-of-map tap rows of dilation 8 only in its split-1 / split-2 bodies (conv_stream.h, KIND 1):
    // the launch is planned again here, as conv_nhwc.hip plans it, to know which items those are
    const pnmn::LaunchPlan lp = pnmn::plan_launch(l.n, cout_blocks, cin_chunks, ntaps, (int)reinterpret_cast<uintptr_t>(l.c));
    size_t index = 0;
    for (const pnmn_conv_item& it : items) {
        int split = 8;
        {
            size_t at = 0;
            for (int k = 0; k < lp.n_seg; ++k) {
                if (index < at + (size_t)lp.count[k]) {
                    split = lp.split[k];
                    break;
                }
                at += (size_t)lp.count[k];
            }
        }
        ++index;
        const bool skips = ntaps == 9 && H == 14 && W == 14 && split <= 2;
        flops += full * (skips ? 1.0 - skipped_tap_tiles(it.dilation) / 117.0 : 1.0);
        const bool mb = (it.flags & PNMN_CONV_MASKBWD) != 0, da = (it.flags & PNMN_CONV_DATTN) != 0;
        maps += cin_chunks + cout_blocks;
        if (it.gate) maps += cin_chunks;
        if (it.flags & PNMN_CONV_ACCUMULATE) maps += cout_blocks;
        if (mb) maps += 1.0 + (it.mb_attn ? 1.0 : 0.0);
        if (da) maps += 1.0;
        extra += HW * 4.0 * ((it.mask ? 1.0 : 0.0) + (((mb || da) && it.mb_attn) ? 2.0 : 0.0));
        bool seen = false;
        for (const float* w : weights) seen = seen || w == it.weight;
        if (!seen) weights.push_back(it.weight);
    }
    t->flops = flops;
    t->bytes = maps * map + extra + (double)weights.size() * wbytes;
    t->n_items = l.n;
    return 0;
}

int wgrad_work(const pnmn_launch& l, const void* records, pnmn_launch_timing* t) {
    const int H = l.p[0], W = l.p[1], ntaps = l.p[2], cin_blocks = l.p[3], cout_blocks = l.p[4];
    const double HW = (double)H * W, C = 128.0;
    const pnmn_wgrad_job* first = static_cast<const pnmn_wgrad_job*>(records);
    const std::vector<pnmn_wgrad_job> jobs(first, first + l.n);
    long n_items = 0;
    for (const pnmn_wgrad_job& j : jobs) n_items += j.item_end - j.item_begin;
    t->n_items = (int32_t)n_items;
    t->flops = 2.0 * n_items * HW * cout_blocks * C * ntaps * cin_blocks * C;
    t->bytes = 4.0 * (n_items * HW * (cin_blocks + cout_blocks) * C + cout_blocks * C * ntaps * cin_blocks * C);
    return 0;
}

}  // namespace

extern "C" int pnmn_launch_trace_begin(void) {
    Trace& T = trace();
    std::lock_guard<std::mutex> g(T.mu);
    for (Traced& r : T.rows) T.pool.push_back(r.e0), T.pool.push_back(r.e1);  // (a trace that was never collected)
    T.arena_used = 0;
    T.rows.clear();
    T.on = true;
    return 0;
}

extern "C" int pnmn_launch_trace_end(pnmn_launch_timing* out, int capacity, int* n_out) {
    Trace& T = trace();
    std::lock_guard<std::mutex> g(T.mu);
    T.on = false;
    if (!n_out || (capacity > 0 && !out)) return PNMN_EINVAL;
    *n_out = (int)T.rows.size();
    if (*n_out > capacity) return PNMN_EAGAIN;  // (nothing is dropped: call again with room for *n_out rows)
    int rc = 0;
    for (size_t i = 0; i < T.rows.size(); ++i) {
        Traced& r = T.rows[i];
        if (rc == 0 && (int)i < capacity) {
            pnmn_launch_timing* t = out + i;
            *t = pnmn_launch_timing{};
            t->op = r.row.op, t->n = r.row.n;
            for (int k = 0; k < 8; ++k) t->p[k] = r.row.p[k];
            hipError_t e = hipEventSynchronize(r.e1);
            if (e == hipSuccess) e = hipEventElapsedTime(&t->ms, r.e0, r.e1);
            rc = (int)e;
            if (rc == 0) rc = r.row.op == PNMN_OP_CONV ? conv_work(r.row, r.records, t) : wgrad_work(r.row, r.records, t);
        }
        T.pool.push_back(r.e0), T.pool.push_back(r.e1);
    }
    T.rows.clear();
    T.arena_used = 0;
    return rc;
}

extern "C" int pnmn_run_launches(const pnmn_launch* list, int n, void* stream) {
    if (n <= 0) return 0;
    if (!list) return PNMN_EINVAL;
    Trace& T = trace();
    const bool tracing = T.on;  // (set and cleared by the thread that runs the instrumented step)
    for (int i = 0; i < n; ++i) {
        const pnmn_launch& l = list[i];
        const int32_t* p = l.p;
        int rc;
        Traced tr{};
        const bool timed = tracing && (l.op == PNMN_OP_CONV || l.op == PNMN_OP_WGRAD) && l.n > 0;
        if (timed) {
            std::lock_guard<std::mutex> g(T.mu);
            tr.row = l, tr.e0 = T.event(), tr.e1 = T.event();
            auto give_back = [&](int code) {  // (an error path keeps the pool whole)
                if (tr.e0) T.pool.push_back(tr.e0);
                if (tr.e1) T.pool.push_back(tr.e1);
                return code;
            };
            if (!tr.e0 || !tr.e1) return give_back(PNMN_EINVAL);
            const bool conv = l.op == PNMN_OP_CONV;
            const size_t bytes = (size_t)l.n * (conv ? sizeof(pnmn_conv_item) : sizeof(pnmn_wgrad_job));
            tr.records = T.cut(bytes);
            if (!tr.records) return give_back(PNMN_EAGAIN);  // (more than 64 MB of records in one trace: collect it more often)
            if (hipMemcpyAsync(tr.records, conv ? l.a : l.b, bytes, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)) != hipSuccess)
                return give_back(PNMN_EINVAL);
            if (hipEventRecord(tr.e0, static_cast<hipStream_t>(stream)) != hipSuccess) return give_back(PNMN_EINVAL);
        }
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
        if (timed) {
            std::lock_guard<std::mutex> g(T.mu);
            if (hipEventRecord(tr.e1, static_cast<hipStream_t>(stream)) != hipSuccess) {
                T.pool.push_back(tr.e0), T.pool.push_back(tr.e1);
                return PNMN_EINVAL;
            }
            T.rows.push_back(tr);
        }
        if (rc != 0) return rc;
    }
    return 0;
}
