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

namespace {

using pnmn::CB;
constexpr int C = PNMN_CHANNELS;
constexpr int NTHREADS = 512;
constexpr int NHW = NTHREADS / 32;  // half-waves per workgroup

__device__ __forceinline__ float half_wave_sum(float v) {
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 1);
    return v;
}
__device__ __forceinline__ float dot4(f32x4 a, f32x4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ float sigmoidf_(float z) { return 1.f / (1.f + expf(-z)); }

// The point-wise modules below restate pointwise.hip's 256-thread kernels for the executor's 512-thread workgroup
// (sixteen half-waves instead of eight; the reductions through LDS have sixteen rows).  tests/test_nmn_gpu.py holds
// the two paths to each other.

// ---- one-channel head: conv1x1 (128 -> 1) + sigmoid (nmn_modules.py:86,167) ------------------------------------
__device__ void dot1_fwd(const pnmn_dot1_item& it, int HW) {
    constexpr int NB = 13;  // pixel rows requested before the first reduction (one round trip for a 14x14 map)
    const int h = threadIdx.x & 31, hw = threadIdx.x >> 5;
    const pnmn::gfloat* in = pnmn::as_global(it.in);
    pnmn::gfloat* out = pnmn::as_global(it.out);
    const f32x4 w = pnmn::load4(pnmn::as_global(it.w) + 4 * h);
    const float b = pnmn::as_global(it.b)[0];
    for (int p0 = hw; p0 < HW; p0 += NHW * NB) {
        f32x4 x[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int p = p0 + NHW * k;
            x[k] = p < HW ? pnmn::load4(in + (size_t)p * C + 4 * h) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int p = p0 + NHW * k;
            const float s = half_wave_sum(dot4(x[k], w));
            if (h == 0 && p < HW) out[p] = sigmoidf_(s + b);
        }
    }
}

__device__ void dot1_bwd(const pnmn_dot1_item& it, int HW, float* red /* [NHW][C + 1] */) {
    constexpr int NB = 7;
    const int h = threadIdx.x & 31, hw = threadIdx.x >> 5;
    const pnmn::gfloat* in = pnmn::as_global(it.in);
    const pnmn::gfloat* outv = pnmn::as_global(it.out);
    const pnmn::gfloat* dout = pnmn::as_global(it.dout);
    pnmn::gfloat* din = pnmn::as_global(it.din);
    const f32x4 w = pnmn::load4(pnmn::as_global(it.w) + 4 * h);
    f32x4 dw = f32x4{0.f, 0.f, 0.f, 0.f};
    float db = 0.f;
    for (int p0 = hw; p0 < HW; p0 += NHW * NB) {
        f32x4 x[NB];
        float o[NB], g[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int p = p0 + NHW * k;
            const bool in_range = p < HW;
            x[k] = in_range ? pnmn::load4(in + (size_t)p * C + 4 * h) : f32x4{0.f, 0.f, 0.f, 0.f};
            o[k] = in_range ? outv[p] : 0.f;
            g[k] = in_range ? dout[p] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int p = p0 + NHW * k;
            const float dz = g[k] * o[k] * (1.f - o[k]);
            dw += x[k] * dz;
            db += dz;
            if (p < HW) pnmn::store4(din + (size_t)p * C + 4 * h, w * dz);
        }
    }
    float* r = red + hw * (C + 1);
    r[4 * h + 0] = dw.x;
    r[4 * h + 1] = dw.y;
    r[4 * h + 2] = dw.z;
    r[4 * h + 3] = dw.w;
    if (h == 0) r[C] = db;
    __syncthreads();
    if (threadIdx.x <= C) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NHW; ++k) s += red[k * (C + 1) + threadIdx.x];
        unsafeAtomicAdd(threadIdx.x < C ? it.dw + threadIdx.x : it.db, s);
    }
}

// ---- SameModule (nmn_modules.py:200-208) -------------------------------------------------------------------------
__device__ int first_argmax(const float* __restrict__ attn, int HW, float* sval, int* sidx) {
    // first maximum in scan order (what max_pool2d(return_indices=True) reports)
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int p = threadIdx.x; p < HW; p += NTHREADS) {
        const float v = attn[p];
        if (v > best || (v != v && best == best)) {
            best = v;
            bi = p;
        }
    }
    sval[threadIdx.x] = best;
    sidx[threadIdx.x] = bi;
    __syncthreads();
    for (int s = NTHREADS / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            const float ov = sval[threadIdx.x + s], mv = sval[threadIdx.x];
            const int oi = sidx[threadIdx.x + s], mi = sidx[threadIdx.x];
            if (ov > mv || (ov == mv && oi < mi)) {
                sval[threadIdx.x] = ov;
                sidx[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    const int r = sidx[0];
    __syncthreads();
    return r == 0x7fffffff ? 0 : r;
}

__device__ void same_fwd(const pnmn_same_item& it, int HW, float* scratch) {
    const int j = first_argmax(it.attn, HW, scratch, reinterpret_cast<int*>(scratch + NTHREADS));
    const int h = threadIdx.x & 31, hw = threadIdx.x >> 5;
    const f32x4 v = *reinterpret_cast<const f32x4*>(it.feats + (size_t)j * C + 4 * h);
    const f32x4 wv = *reinterpret_cast<const f32x4*>(it.w + 4 * h) * v;
    const float wa = it.w[C];
    const float b = it.b[0];
    for (int p = hw; p < HW; p += NHW) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(it.feats + (size_t)p * C + 4 * h);
        const float s = half_wave_sum(dot4(x, wv));
        if (h == 0) it.out[p] = sigmoidf_(s + wa * it.attn[p] + b);
    }
}

__device__ void same_bwd(const pnmn_same_item& it, int HW, float* scratch) {
    const int j = first_argmax(it.attn, HW, scratch, reinterpret_cast<int*>(scratch + NTHREADS));
    float* red = scratch;  // [NHW][2C + 2] (the arg-max scratch is dead: first_argmax ends with a barrier)
    const int h = threadIdx.x & 31, hw = threadIdx.x >> 5;
    const f32x4 v = *reinterpret_cast<const f32x4*>(it.feats + (size_t)j * C + 4 * h);
    const f32x4 w = *reinterpret_cast<const f32x4*>(it.w + 4 * h);
    const f32x4 wv = w * v;
    const float wa = it.w[C];
    f32x4 sfx = f32x4{0.f, 0.f, 0.f, 0.f};  // sum_p dz[p] * feats[p][c]
    float dwa = 0.f, db = 0.f;
    for (int p = hw; p < HW; p += NHW) {
        const float o = it.out[p];
        const float dz = it.dout[p] * o * (1.f - o);
        const f32x4 x = *reinterpret_cast<const f32x4*>(it.feats + (size_t)p * C + 4 * h);
        sfx += x * dz;
        if (h == 0) {
            dwa += dz * it.attn[p];
            db += dz;
            if (it.dattn) unsafeAtomicAdd(it.dattn + p, dz * wa);
        }
        const f32x4 df = wv * dz;  // through x = feats * v, wrt feats[p]
        float* d = it.dfeats + (size_t)p * C + 4 * h;
        unsafeAtomicAdd(d + 0, df.x);
        unsafeAtomicAdd(d + 1, df.y);
        unsafeAtomicAdd(d + 2, df.z);
        unsafeAtomicAdd(d + 3, df.w);
    }
    float* r = red + hw * (2 * C + 2);
    r[4 * h + 0] = sfx.x;
    r[4 * h + 1] = sfx.y;
    r[4 * h + 2] = sfx.z;
    r[4 * h + 3] = sfx.w;
    if (h == 0) {
        r[2 * C] = dwa;
        r[2 * C + 1] = db;
    }
    __syncthreads();
    if (threadIdx.x < C) {
        const int c = threadIdx.x;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NHW; ++k) s += red[k * (2 * C + 2) + c];
        unsafeAtomicAdd(it.dw + c, s * it.feats[(size_t)j * C + c]);      // d/dw[c]
        unsafeAtomicAdd(it.dfeats + (size_t)j * C + c, s * it.w[c]);      // through v = feats[j]
    } else if (threadIdx.x == C || threadIdx.x == C + 1) {
        const int k2 = 2 * C + (threadIdx.x - C);
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NHW; ++k) s += red[k * (2 * C + 2) + k2];
        unsafeAtomicAdd(threadIdx.x == C ? it.dw + C : it.db, s);
    }
}

// ---- And / Or (nmn_modules.py:25-27,43-45) -----------------------------------------------------------------------
__device__ void minmax_fwd(const pnmn_minmax_item& it, int HW) {
    const int oc = it.a_channels > it.b_channels ? it.a_channels : it.b_channels;
    const int n = HW * oc;
    for (int i = threadIdx.x; i < n; i += NTHREADS) {
        const int p = i / oc;
        const int c = i - p * oc;
        const float a = it.a[it.a_channels == 1 ? p : p * C + c];
        const float b = it.b[it.b_channels == 1 ? p : p * C + c];
        // torch.min/max propagate NaN; fminf/fmaxf would not
        it.out[i] = (a != a || b != b) ? NAN : (it.is_max ? (a > b ? a : b) : (a < b ? a : b));
    }
}

__device__ void minmax_bwd(const pnmn_minmax_item& it, int HW) {
    const int oc = it.a_channels > it.b_channels ? it.a_channels : it.b_channels;
    const int h = threadIdx.x & 31, hw = threadIdx.x >> 5;
    if (oc == 1) {
        for (int p = threadIdx.x; p < HW; p += NTHREADS) {
            const float a = it.a[p], b = it.b[p], g = it.dout[p];
            const bool a_wins = it.is_max ? (a > b) : (a < b);
            const float ga = (a == b) ? 0.5f * g : (a_wins ? g : 0.f);
            const float gb = (a == b) ? 0.5f * g : (a_wins ? 0.f : g);
            if (it.da) unsafeAtomicAdd(it.da + p, ga);
            if (it.db) unsafeAtomicAdd(it.db + p, gb);
        }
        return;
    }
    // oc == C == 128: half-wave per pixel, 4 channels per lane
    for (int p = hw; p < HW; p += NHW) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(it.dout + (size_t)p * C + 4 * h);
        f32x4 a, b;
        if (it.a_channels == 1) {
            const float s = it.a[p];
            a = f32x4{s, s, s, s};
        } else {
            a = *reinterpret_cast<const f32x4*>(it.a + (size_t)p * C + 4 * h);
        }
        if (it.b_channels == 1) {
            const float s = it.b[p];
            b = f32x4{s, s, s, s};
        } else {
            b = *reinterpret_cast<const f32x4*>(it.b + (size_t)p * C + 4 * h);
        }
        f32x4 ga, gb;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool a_wins = it.is_max ? (a[k] > b[k]) : (a[k] < b[k]);
            ga[k] = (a[k] == b[k]) ? 0.5f * g[k] : (a_wins ? g[k] : 0.f);
            gb[k] = (a[k] == b[k]) ? 0.5f * g[k] : (a_wins ? 0.f : g[k]);
        }
        if (it.da) {
            if (it.a_channels == 1) {
                const float s = half_wave_sum(ga.x + ga.y + ga.z + ga.w);
                if (h == 0) unsafeAtomicAdd(it.da + p, s);
            } else {
                float* d = it.da + (size_t)p * C + 4 * h;
                unsafeAtomicAdd(d + 0, ga.x);
                unsafeAtomicAdd(d + 1, ga.y);
                unsafeAtomicAdd(d + 2, ga.z);
                unsafeAtomicAdd(d + 3, ga.w);
            }
        }
        if (it.db) {
            if (it.b_channels == 1) {
                const float s = half_wave_sum(gb.x + gb.y + gb.z + gb.w);
                if (h == 0) unsafeAtomicAdd(it.db + p, s);
            } else {
                float* d = it.db + (size_t)p * C + 4 * h;
                unsafeAtomicAdd(d + 0, gb.x);
                unsafeAtomicAdd(d + 1, gb.y);
                unsafeAtomicAdd(d + 2, gb.z);
                unsafeAtomicAdd(d + 3, gb.w);
            }
        }
    }
}

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
                dot1_fwd(static_cast<const pnmn_dot1_item*>(prog->records[PNMN_EXEC_DOT_FWD])[u.record], HW);
                break;
            case PNMN_EXEC_DOT_BWD:
                dot1_bwd(static_cast<const pnmn_dot1_item*>(prog->records[PNMN_EXEC_DOT_BWD])[u.record], HW, lds);
                break;
            case PNMN_EXEC_SAME_FWD:
                same_fwd(static_cast<const pnmn_same_item*>(prog->records[PNMN_EXEC_SAME_FWD])[u.record], HW, lds);
                break;
            case PNMN_EXEC_SAME_BWD:
                same_bwd(static_cast<const pnmn_same_item*>(prog->records[PNMN_EXEC_SAME_BWD])[u.record], HW, lds);
                break;
            case PNMN_EXEC_MINMAX_FWD:
                minmax_fwd(static_cast<const pnmn_minmax_item*>(prog->records[PNMN_EXEC_MINMAX_FWD])[u.record], HW);
                break;
            case PNMN_EXEC_MINMAX_BWD:
                minmax_bwd(static_cast<const pnmn_minmax_item*>(prog->records[PNMN_EXEC_MINMAX_BWD])[u.record], HW);
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
