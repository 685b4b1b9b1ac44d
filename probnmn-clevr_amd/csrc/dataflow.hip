// Persistent dataflow executor for the module programs of one step (gfx950).
//
// The per-level grouped launches (conv_nhwc.hip, pointwise.hip) leave CUs idle whenever a level has
// fewer ready items than the chip has CUs -- and programs differ in length, so most levels do.
// Here the whole forward (or backward) of all module programs of the batch is ONE launch: a task
// list in topological (level) order, one persistent workgroup per CU that repeatedly
//     takes the next task index from a global counter,
//     waits until the tasks that produce its inputs have signalled completion,
//     runs it (a slice of a convolution, a 1-channel head, Same, And/Or -- forward or backward),
//     publishes its outputs (agent-scope release) and bumps its completion counter.
// Different examples' chains advance independently, so the chip stays full until the last few
// tasks; a convolution is split into KSPLIT sub-tasks of 128/KSPLIT output channels so that the
// critical path of the longest program (25 convolutions deep) stays below the total work time.
//
// Progress guarantee: a task only ever waits for tasks with a smaller index, all of which have
// already been claimed by a workgroup that is running (or has finished) -- no co-residency of
// the whole grid is assumed, the grid size is a speed choice.  Every wait is bounded; on a
// time-out the kernel raises ctrl[1] and all workgroups drain.
// Inter-workgroup visibility follows the agent-scope release/acquire recipe of the CDNA guide:
// producer: every wave drains its stores -> barrier -> one lane: release fence, drain, relaxed
// agent-scope counter increment; consumer: one lane polls relaxed, one acquire fence, barrier.
#include "conv_body.h"

#include <math.h>

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

template <typename T>
__device__ __forceinline__ T* P(const pnmn_task& t, int i) {
    return reinterpret_cast<T*>(t.p[i]);
}

// ---- 1-channel head ---------------------------------------------------------------------------------
__device__ void dot1_fwd(const pnmn_task& t, int HW) {
    const float* in = P<const float>(t, 0);
    const int h = threadIdx.x & 31, hw = threadIdx.x >> 5;
    const f32x4 w = *reinterpret_cast<const f32x4*>(P<const float>(t, 1) + 4 * h);
    const float b = P<const float>(t, 2)[0];
    float* out = P<float>(t, 3);
    for (int p = hw; p < HW; p += NHW) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(in + (size_t)p * C + 4 * h);
        const float s = half_wave_sum(dot4(x, w));
        if (h == 0) out[p] = sigmoidf_(s + b);
    }
}

__device__ void dot1_bwd(const pnmn_task& t, int HW, float* red /* [NHW][C+1] */) {
    const float* in = P<const float>(t, 0);
    const float* out = P<const float>(t, 3);
    const float* dout = P<const float>(t, 4);
    float* din = P<float>(t, 5);
    const int h = threadIdx.x & 31, hw = threadIdx.x >> 5;
    const f32x4 w = *reinterpret_cast<const f32x4*>(P<const float>(t, 1) + 4 * h);
    f32x4 dw = f32x4{0.f, 0.f, 0.f, 0.f};
    float db = 0.f;
    for (int p = hw; p < HW; p += NHW) {
        const float o = out[p];
        const float dz = dout[p] * o * (1.f - o);
        const f32x4 x = *reinterpret_cast<const f32x4*>(in + (size_t)p * C + 4 * h);
        dw += x * dz;
        db += dz;
        *reinterpret_cast<f32x4*>(din + (size_t)p * C + 4 * h) = w * dz;
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
        unsafeAtomicAdd(threadIdx.x < C ? P<float>(t, 6) + threadIdx.x : P<float>(t, 7), s);
    }
}

// ---- SameModule -------------------------------------------------------------------------------------
__device__ int first_argmax(const float* __restrict__ attn, int HW, float* sval, int* sidx) {
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int p = threadIdx.x; p < HW; p += NTHREADS) {
        const float v = attn[p];
        if (v > best) {
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

__device__ void same_fwd(const pnmn_task& t, int HW, float* scratch) {
    const float* feats = P<const float>(t, 0);
    const float* attn = P<const float>(t, 1);
    const float* w = P<const float>(t, 2);
    float* out = P<float>(t, 4);
    const int j = first_argmax(attn, HW, scratch, reinterpret_cast<int*>(scratch + NTHREADS));
    const int h = threadIdx.x & 31, hw = threadIdx.x >> 5;
    const f32x4 v = *reinterpret_cast<const f32x4*>(feats + (size_t)j * C + 4 * h);
    const f32x4 wv = *reinterpret_cast<const f32x4*>(w + 4 * h) * v;
    const float wa = w[C];
    const float b = P<const float>(t, 3)[0];
    for (int p = hw; p < HW; p += NHW) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(feats + (size_t)p * C + 4 * h);
        const float s = half_wave_sum(dot4(x, wv));
        if (h == 0) out[p] = sigmoidf_(s + wa * attn[p] + b);
    }
}

__device__ void same_bwd(const pnmn_task& t, int HW, float* scratch) {
    const float* feats = P<const float>(t, 0);
    const float* attn = P<const float>(t, 1);
    const float* w = P<const float>(t, 2);
    const float* out = P<const float>(t, 4);
    const float* dout = P<const float>(t, 5);
    float* dfeats = P<float>(t, 6);
    float* dattn = P<float>(t, 7);
    float* dwp = P<float>(t, 8);
    float* dbp = P<float>(t, 9);
    const int j = first_argmax(attn, HW, scratch, reinterpret_cast<int*>(scratch + NTHREADS));
    float* red = scratch;  // [NHW][2C+2] (the arg-max scratch is dead)
    const int h = threadIdx.x & 31, hw = threadIdx.x >> 5;
    const f32x4 v = *reinterpret_cast<const f32x4*>(feats + (size_t)j * C + 4 * h);
    const f32x4 wv = *reinterpret_cast<const f32x4*>(w + 4 * h) * v;
    const float wa = w[C];
    f32x4 sfx = f32x4{0.f, 0.f, 0.f, 0.f};
    float dwa = 0.f, db = 0.f;
    for (int p = hw; p < HW; p += NHW) {
        const float o = out[p];
        const float dz = dout[p] * o * (1.f - o);
        const f32x4 x = *reinterpret_cast<const f32x4*>(feats + (size_t)p * C + 4 * h);
        sfx += x * dz;
        if (h == 0) {
            dwa += dz * attn[p];
            db += dz;
            if (dattn) unsafeAtomicAdd(dattn + p, dz * wa);
        }
        const f32x4 df = wv * dz;
        float* d = dfeats + (size_t)p * C + 4 * h;
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
        unsafeAtomicAdd(dwp + c, s * feats[(size_t)j * C + c]);
        unsafeAtomicAdd(dfeats + (size_t)j * C + c, s * w[c]);
    } else if (threadIdx.x == C || threadIdx.x == C + 1) {
        const int k2 = 2 * C + (threadIdx.x - C);
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NHW; ++k) s += red[k * (2 * C + 2) + k2];
        unsafeAtomicAdd(threadIdx.x == C ? dwp + C : dbp, s);
    }
}

// ---- And / Or ---------------------------------------------------------------------------------------
__device__ void minmax_fwd(const pnmn_task& t, int HW) {
    const float* a = P<const float>(t, 0);
    const float* b = P<const float>(t, 1);
    float* out = P<float>(t, 2);
    const int ac = (t.flags & 1) ? C : 1, bc = (t.flags & 2) ? C : 1, is_max = (t.flags >> 2) & 1;
    const int oc = ac > bc ? ac : bc;
    for (int i = threadIdx.x; i < HW * oc; i += NTHREADS) {
        const int p = i / oc, c = i - p * oc;
        const float x = a[ac == 1 ? p : p * C + c];
        const float y = b[bc == 1 ? p : p * C + c];
        out[i] = (x != x || y != y) ? NAN : (is_max ? (x > y ? x : y) : (x < y ? x : y));
    }
}

__device__ void minmax_bwd(const pnmn_task& t, int HW) {
    const float* a = P<const float>(t, 0);
    const float* b = P<const float>(t, 1);
    const float* dout = P<const float>(t, 3);
    float* da = P<float>(t, 4);
    float* db = P<float>(t, 5);
    const int ac = (t.flags & 1) ? C : 1, bc = (t.flags & 2) ? C : 1, is_max = (t.flags >> 2) & 1;
    const int oc = ac > bc ? ac : bc;
    if (oc == 1) {
        for (int p = threadIdx.x; p < HW; p += NTHREADS) {
            const float x = a[p], y = b[p], g = dout[p];
            const bool aw = is_max ? (x > y) : (x < y);
            if (da) unsafeAtomicAdd(da + p, x == y ? 0.5f * g : (aw ? g : 0.f));
            if (db) unsafeAtomicAdd(db + p, x == y ? 0.5f * g : (aw ? 0.f : g));
        }
        return;
    }
    const int h = threadIdx.x & 31, hw = threadIdx.x >> 5;
    for (int p = hw; p < HW; p += NHW) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(dout + (size_t)p * C + 4 * h);
        f32x4 x, y;
        if (ac == 1) {
            const float s = a[p];
            x = f32x4{s, s, s, s};
        } else {
            x = *reinterpret_cast<const f32x4*>(a + (size_t)p * C + 4 * h);
        }
        if (bc == 1) {
            const float s = b[p];
            y = f32x4{s, s, s, s};
        } else {
            y = *reinterpret_cast<const f32x4*>(b + (size_t)p * C + 4 * h);
        }
        f32x4 ga, gb;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool aw = is_max ? (x[k] > y[k]) : (x[k] < y[k]);
            ga[k] = (x[k] == y[k]) ? 0.5f * g[k] : (aw ? g[k] : 0.f);
            gb[k] = (x[k] == y[k]) ? 0.5f * g[k] : (aw ? 0.f : g[k]);
        }
        if (da) {
            if (ac == 1) {
                const float s = half_wave_sum(ga.x + ga.y + ga.z + ga.w);
                if (h == 0) unsafeAtomicAdd(da + p, s);
            } else {
                float* d = da + (size_t)p * C + 4 * h;
                unsafeAtomicAdd(d + 0, ga.x);
                unsafeAtomicAdd(d + 1, ga.y);
                unsafeAtomicAdd(d + 2, ga.z);
                unsafeAtomicAdd(d + 3, ga.w);
            }
        }
        if (db) {
            if (bc == 1) {
                const float s = half_wave_sum(gb.x + gb.y + gb.z + gb.w);
                if (h == 0) unsafeAtomicAdd(db + p, s);
            } else {
                float* d = db + (size_t)p * C + 4 * h;
                unsafeAtomicAdd(d + 0, gb.x);
                unsafeAtomicAdd(d + 1, gb.y);
                unsafeAtomicAdd(d + 2, gb.z);
                unsafeAtomicAdd(d + 3, gb.w);
            }
        }
    }
}

// ---- the persistent kernel ----------------------------------------------------------------------------
template <int H, int W, int KSPLIT>
__global__ __launch_bounds__(NTHREADS) void dataflow_kernel(const pnmn_task* __restrict__ tasks, int n_tasks,
                                                            int* __restrict__ ctrl, int* __restrict__ done,
                                                            unsigned spin_limit) {
    constexpr int HW = H * W;
    constexpr int LDS_FLOATS = pnmn::lds_rows(HW) * CB;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* lds = reinterpret_cast<float*>(smem_raw);
    int* sh = reinterpret_cast<int*>(lds + LDS_FLOATS);  // [0] task index, [1] abort
    const int tid = threadIdx.x;

    for (;;) {
        __syncthreads();  // the previous task is completely done with LDS and sh
        if (tid == 0) {
            sh[0] = atomicAdd(&ctrl[0], 1);
            sh[1] = __hip_atomic_load(&ctrl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        const int ti = __builtin_amdgcn_readfirstlane(sh[0]);
        if (ti >= n_tasks || sh[1] != 0) break;
        const pnmn_task& t = tasks[ti];  // uniform index: fields are fetched with scalar loads as needed

        // ---- wait for the producers of this task's inputs ----
        if (tid < 3) {
            const int d = t.dep[tid];
            if (d >= 0) {
                const int need = t.need[tid];
                unsigned spins = 0;
                while (__hip_atomic_load(&done[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > spin_limit) {
                        atomicExch(&ctrl[1], ti + 1);  // report and let everyone drain
                        sh[1] = 1;
                        break;
                    }
                }
            }
        }
        __syncthreads();
        if (sh[1] != 0) break;
        if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads();

        switch (t.type) {
            case PNMN_TASK_CONV: {
                pnmn_conv_item it;
                it.in = P<const float>(t, 0);
                it.in2 = P<const float>(t, 1);
                it.mask = P<const float>(t, 2);
                it.gate = P<const float>(t, 3);
                it.weight = P<const float>(t, 4);
                it.bias = P<const float>(t, 5);
                it.out = P<float>(t, 6);
                it.dilation = t.dilation;
                it.flags = t.flags & (PNMN_CONV_ACCUMULATE | PNMN_CONV_ATOMIC);
                it.mb_feats = nullptr;
                it.mb_attn = nullptr;
                it.mb_dfeats = nullptr;
                it.mb_dattn = nullptr;
                const int relu = (t.flags >> 4) & 1;
                const int ntaps = (t.flags & 32) ? 1 : 9;
                const int chunks = (t.flags & 64) ? 2 : 1;
                const pnmn::MaskBwd mb{P<const float>(t, 7), P<const float>(t, 8), P<float>(t, 9), P<float>(t, 10)};
                pnmn::conv_body<H, W, KSPLIT>(it, t.sub, 0, chunks, ntaps, C, C, relu, lds,
                                              (t.flags & 128) ? &mb : nullptr);
                break;
            }
            case PNMN_TASK_DOT_FWD: dot1_fwd(t, HW); break;
            case PNMN_TASK_DOT_BWD: dot1_bwd(t, HW, lds); break;
            case PNMN_TASK_SAME_FWD: same_fwd(t, HW, lds); break;
            case PNMN_TASK_SAME_BWD: same_bwd(t, HW, lds); break;
            case PNMN_TASK_MINMAX_FWD: minmax_fwd(t, HW); break;
            case PNMN_TASK_MINMAX_BWD: minmax_bwd(t, HW); break;
            default: break;
        }

        // ---- publish: every wave drains its stores/atomics, then one lane releases and signals ----
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(&done[tasks[ti].slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int H, int W, int KSPLIT>
int launch_dataflow(const pnmn_task* tasks, int n_tasks, int* ctrl, int* done, int n_workgroups, hipStream_t stream) {
    constexpr size_t lds_bytes = (size_t)pnmn::lds_rows(H * W) * CB * sizeof(float) + 64;
    static bool configured = false;
    auto kern = dataflow_kernel<H, W, KSPLIT>;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    const int grid = n_tasks < n_workgroups ? n_tasks : n_workgroups;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHREADS), lds_bytes, stream, tasks, n_tasks, ctrl, done,
                       4000000u /* ~1 s of s_sleep(8) polls */);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" int pnmn_dataflow(const pnmn_task* tasks, int n_tasks, int* ctrl, int* done, int H, int W, int ksplit,
                             int n_workgroups, void* stream) {
    if (n_tasks <= 0) return 0;
    if (!tasks || !ctrl || !done || n_workgroups < 1) return PNMN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (H == 14 && W == 14) {
        switch (ksplit) {
            case 1: return launch_dataflow<14, 14, 1>(tasks, n_tasks, ctrl, done, n_workgroups, s);
            case 2: return launch_dataflow<14, 14, 2>(tasks, n_tasks, ctrl, done, n_workgroups, s);
            case 4: return launch_dataflow<14, 14, 4>(tasks, n_tasks, ctrl, done, n_workgroups, s);
            default: return PNMN_EINVAL;
        }
    }
    return PNMN_ESHAPE;
}
