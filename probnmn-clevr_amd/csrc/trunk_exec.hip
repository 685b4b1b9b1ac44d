// Trunk executor (gfx950): the module programs of a batch, forward or data-gradient, as ONE launch.
//
// The grouped launches of conv_nhwc.hip / pointwise.hip are level-synchronous: a kernel boundary sits between two
// levels of the module programs (reference nmn.py:197-238 runs them one example and one module at a time), although
// an example's next module only needs that example's previous one.  At 64 sampled rows that is ~45 dependent
// launches forward and ~60 backward of 1-65 items each; at 1024 rows every level ends in a partly filled round.
//
// Here persistent workgroups walk the same work-item records as UNITS (include/probnmn_hip.h: pnmn_exec_unit,
// built by host_trunk.hip).  Round 1 had a first executor (one global queue, exact producer lists, agent-scope
// release / acquire around every task, one K-split for the whole launch: +3 %, removed in round 2).  What is
// different now:
//   * every example is pinned to ONE XCD; its units are taken only by workgroups running on that XCD (queue index =
//     HW_REG_XCC_ID, not blockIdx: correctness does not rest on how the dispatcher deals workgroups).  A producer and
//     its consumer therefore share an L2, and a hand-off is "my stores have reached L2" + "drop this CU's L1" --
//     not the L2 write-back + invalidate of an agent-scope release / acquire (6-9 us with 256 workgroups resident,
//     cluster.h);
//   * a unit waits on ONE counter, its example's count of completed units: `need` = the example's units in all
//     earlier launches of the level order, i.e. an example sees exactly the launch order the grouped path gives it
//     (read-modify-write accumulations keep their order; units of one launch still run side by side);
//   * the split of a convolution is chosen per (launch, XCD) by the launch planner of conv_plan.h, as the grouped
//     path does per launch.
// Progress: a queue is in launch order and taken in order, so a unit only waits for units that were taken earlier --
// by workgroups that are running.  No co-residency of the grid is assumed; every wait is bounded by a trap.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "conv_body.h"
#include "pointwise_body.h"

namespace {

using pnmn::CB;
constexpr int C = PNMN_CHANNELS;
constexpr int NTHREADS = 512;
constexpr int NHW = NTHREADS / 32;  // half-waves per workgroup

// One convolution unit: the body of conv_nhwc.hip's kernel with the unit's own split.  A real call, not inlined: the five
// instantiations of the body each fill the 256-register budget, and inlined into the loop below the loop's own state
// pushed them into scratch (302 spilled registers); as a callee the body is allocated as it is in conv_nhwc_kernel.
// Arguments of a non-kernel function arrive in vector registers -- readfirstlane puts the (uniform) record address and
// parameters back into scalar registers, so that the record's pointers are scalar loads as they are there.
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

template <int H, int W, int TH>
__device__ __noinline__ void conv_unit(const pnmn_conv_item* item, int packed /* split | sub << 8 | band << 16 | kind << 24 */) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* lds = reinterpret_cast<float*>(smem_raw);
    const uint64_t addr = (uint64_t)(uint32_t)uniform((int)(reinterpret_cast<uint64_t>(item) & 0xffffffffu)) |
                          ((uint64_t)(uint32_t)uniform((int)(reinterpret_cast<uint64_t>(item) >> 32)) << 32);
    packed = uniform(packed);
    const pnmn_conv_item it = *reinterpret_cast<const pnmn_conv_item*>(addr);
    const int split = packed & 255, sub = (packed >> 8) & 255, band = (packed >> 16) & 255, kind = (packed >> 24) & 255;
    const int cin_chunks = kind == PNMN_EXEC_PROJ ? 2 : 1;
    const int ntaps = (kind == PNMN_EXEC_CONV || kind == PNMN_EXEC_DGRAD) ? 9 : 1;
    const int relu = kind <= PNMN_EXEC_PROJ ? 1 : 0;
    const pnmn::MaskBwd mb{it.mb_feats, it.mb_attn, it.mb_dfeats, it.mb_dattn};
    const pnmn::MaskBwd* mbp = (it.flags & (PNMN_CONV_MASKBWD | PNMN_CONV_DATTN)) ? &mb : nullptr;
    switch (split) {  // (uniform over the workgroup)
        case 1:
            pnmn::conv_body<H, W, TH, 1>(it, band, 0, 0, cin_chunks, ntaps, C, C, relu, lds, mbp);
            break;
        case 2:
            pnmn::conv_body<H, W, TH, 2>(it, band, sub, 0, cin_chunks, ntaps, C, C, relu, lds, mbp);
            break;
        case 4:
            pnmn::conv_body<H, W, TH, 4>(it, band, sub, 0, cin_chunks, ntaps, C, C, relu, lds, mbp);
            break;
        case 8:
            pnmn::conv_body<H, W, TH, 8>(it, band, sub, 0, cin_chunks, ntaps, C, C, relu, lds, mbp);
            break;
        default:  // 16
            pnmn::conv_body<H, W, TH, 8, 2>(it, band, sub % 8, 0, cin_chunks, ntaps, C, C, relu, lds, mbp, sub / 8);
            break;
    }
}

template <int H, int W, int TH>
__global__ __launch_bounds__(NTHREADS, 2) void trunk_exec_kernel(const pnmn_exec_program* __restrict__ prog) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* lds = reinterpret_cast<float*>(smem_raw);
    __shared__ int taken;
    constexpr int HW = H * W;
    const int tid = threadIdx.x;
    // the XCD this workgroup RUNS on picks its queue (HW_REG_XCC_ID[3:0], as cluster.h reads it)
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;
    if (xcc >= 8) __builtin_trap();
    const pnmn_exec_unit* __restrict__ units = prog->units[xcc];
    const int n_units = prog->n_units[xcc];
    int* head = prog->heads + xcc * PNMN_EXEC_COUNTER_STRIDE;
    int* progress = prog->progress;
    int* dbg = prog->debug ? prog->debug + blockIdx.x * 16 : nullptr;  // (host-visible trace, hang debugging)
    auto trace = [&](int slot, int v) {
        if (dbg && tid == 0) __hip_atomic_store(dbg + slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    };
    trace(5, (int)xcc), trace(7, n_units), trace(0, 1);
    int units_run = 0;
    // (debugging only: where this workgroup's time goes, in units of 1024 cycles of the 100 MHz constant clock)
    long long t_begin = dbg ? (long long)__builtin_amdgcn_s_memrealtime() : 0, t_wait = 0, t_work = 0, t0 = t_begin;

    for (;;) {
        if (tid == 0) taken = __hip_atomic_fetch_add(head, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int ui = __builtin_amdgcn_readfirstlane(taken);
        trace(1, ui), trace(0, 2);
        if (ui >= n_units) break;
        const pnmn_exec_unit u = units[ui];
        trace(2, u.owner), trace(3, u.need), trace(8, u.kind | (u.split << 8) | (u.sub << 16)), trace(9, u.record), trace(0, 3);
        int* done = progress + (size_t)u.owner * PNMN_EXEC_COUNTER_STRIDE;
        if (tid == 0 && u.need > 0) {
            int spins = 0;
            int seen;
            while ((seen = __hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < u.need) {
                trace(4, seen);
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 22)) __builtin_trap();  // (a bug in the unit order, not a schedule: see the header)
            }
            // the producers ran on this XCD: their stores are in its L2; drop this CU's vector L1 (shared by all waves)
            asm volatile("buffer_inv sc0\n\ts_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();  // (also: everyone has read `taken`)
        trace(0, 4);
        if (dbg) {
            const long long t = (long long)__builtin_amdgcn_s_memrealtime();
            t_wait += t - t0, t0 = t;
        }

        if (u.kind <= PNMN_EXEC_PDGRAD) {
            conv_unit<H, W, TH>(static_cast<const pnmn_conv_item*>(prog->records[u.kind]) + u.record,
                                (int)u.split | ((int)u.sub << 8) | ((int)u.band << 16) | ((int)u.kind << 24));
        } else
        switch (u.kind) {
            case PNMN_EXEC_DOT_FWD:
                pnmn::pointwise::dot1_fwd<NTHREADS>(static_cast<const pnmn_dot1_item*>(prog->records[PNMN_EXEC_DOT_FWD])[u.record], HW);
                break;
            case PNMN_EXEC_DOT_BWD:
                pnmn::pointwise::dot1_bwd<NTHREADS>(static_cast<const pnmn_dot1_item*>(prog->records[PNMN_EXEC_DOT_BWD])[u.record], HW, lds);
                break;
            case PNMN_EXEC_SAME_FWD:
                pnmn::pointwise::same_fwd<NTHREADS>(static_cast<const pnmn_same_item*>(prog->records[PNMN_EXEC_SAME_FWD])[u.record], HW, lds);
                break;
            case PNMN_EXEC_SAME_BWD:
                pnmn::pointwise::same_bwd<NTHREADS>(static_cast<const pnmn_same_item*>(prog->records[PNMN_EXEC_SAME_BWD])[u.record], HW, lds);
                break;
            case PNMN_EXEC_MINMAX_FWD:
                pnmn::pointwise::minmax_fwd<NTHREADS>(static_cast<const pnmn_minmax_item*>(prog->records[PNMN_EXEC_MINMAX_FWD])[u.record], HW);
                break;
            case PNMN_EXEC_MINMAX_BWD:
                pnmn::pointwise::minmax_bwd<NTHREADS>(static_cast<const pnmn_minmax_item*>(prog->records[PNMN_EXEC_MINMAX_BWD])[u.record], HW);
                break;
            default:
                __builtin_trap();
        }

        // publish: every wave's stores and atomics have reached L2, then one lane counts the unit
        trace(0, 5);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // (also: nobody reads LDS or `taken` any more)
        if (tid == 0) __hip_atomic_fetch_add(done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        trace(6, ++units_run), trace(0, 6);
        if (dbg) {
            const long long t = (long long)__builtin_amdgcn_s_memrealtime();
            t_work += t - t0, t0 = t;
            trace(10, (int)t_wait), trace(11, (int)t_work), trace(12, (int)(t - t_begin));
        }
    }
    trace(0, 7);
}

template <int H, int W, int TH>
int launch_exec(const pnmn_exec_program* program, int workgroups, hipStream_t stream) {
    constexpr size_t lds_bytes = (size_t)pnmn::lds_rows<H, W, TH>() * CB * sizeof(float);
    static_assert(lds_bytes <= 160 * 1024 - 64, "the staged region (+ the queue word) must fit the CU's LDS");
    static_assert((size_t)NHW * (2 * C + 2) * 4 <= lds_bytes && (size_t)2 * NTHREADS * 4 <= lds_bytes, "point-wise scratch");
    static bool configured = false;
    auto kern = trunk_exec_kernel<H, W, TH>;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    hipLaunchKernelGGL(kern, dim3(workgroups), dim3(NTHREADS), lds_bytes, stream, program);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" int pnmn_trunk_exec_debug_block(int32_t** out) {
    if (!out) return PNMN_EINVAL;
    static int32_t* block = [] {
        void* p = nullptr;
        if (!getenv("PNMN_EXEC_DEBUG")) return static_cast<int32_t*>(nullptr);
        if (hipHostMalloc(&p, 256 * 16 * sizeof(int32_t), hipHostMallocMapped) != hipSuccess) return static_cast<int32_t*>(nullptr);
        memset(p, 0, 256 * 16 * sizeof(int32_t));
        return static_cast<int32_t*>(p);
    }();
    *out = block;
    return 0;
}

extern "C" int pnmn_trunk_exec(const pnmn_exec_program* program, int workgroups, int H, int W, void* stream) {
    if (!program || workgroups < 8 || workgroups > 256 || (workgroups & 7)) return PNMN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (H == 14 && W == 14) return launch_exec<14, 14, 14>(program, workgroups, s);
    if (H == 28 && W == 28) return launch_exec<28, 28, 7>(program, workgroups, s);
    return PNMN_ESHAPE;
}
