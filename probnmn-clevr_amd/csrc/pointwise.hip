// Bandwidth-bound kernels of the NMN path for gfx950: the single-channel heads (conv1x1->1 +
// sigmoid, SameModule), And/Or compose, attention-mask backward, layout changes, the
// classifier's max-pool/flatten, the answer loss and the fused clamp+Adam update.
//
// Conventions: feature maps are NHWC [HW][128]; a "half-wave" (32 lanes x float4 = one 512-byte
// pixel row) is the unit that walks pixels, so every global access is a full coalesced row and
// channel reductions are five __shfl_xor steps inside the half-wave.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/probnmn_hip.h"
#include "lds_optin.h"
#include "global_ptr.h"
#include "pointwise_body.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int C = PNMN_CHANNELS;  // 128

using pnmn::pointwise::dot4;
using pnmn::pointwise::half_wave_sum;

// One workgroup of 256 threads per item (bodies: pointwise_body.h).
constexpr int NT = 256;

// conv1x1 (128 -> 1) + sigmoid
__global__ __launch_bounds__(NT) void dot1_sigmoid_fwd_kernel(const pnmn_dot1_item* __restrict__ items, int HW) {
    const pnmn_dot1_item it = items[blockIdx.x];
    pnmn::pointwise::dot1_fwd<NT>(it, HW);
}

__global__ __launch_bounds__(NT) void dot1_sigmoid_bwd_kernel(const pnmn_dot1_item* __restrict__ items, int HW) {
    __shared__ float scratch[pnmn::pointwise::scratch_floats<NT>()];
    const pnmn_dot1_item it = items[blockIdx.x];
    pnmn::pointwise::dot1_bwd<NT>(it, HW, scratch);
}

// SameModule
__global__ __launch_bounds__(NT) void same_fwd_kernel(const pnmn_same_item* __restrict__ items, int HW) {
    __shared__ float scratch[pnmn::pointwise::scratch_floats<NT>()];
    const pnmn_same_item it = items[blockIdx.x];
    pnmn::pointwise::same_fwd<NT>(it, HW, scratch);
}

__global__ __launch_bounds__(NT) void same_bwd_kernel(const pnmn_same_item* __restrict__ items, int HW) {
    __shared__ float scratch[pnmn::pointwise::scratch_floats<NT>()];
    const pnmn_same_item it = items[blockIdx.x];
    pnmn::pointwise::same_bwd<NT>(it, HW, scratch);
}

// And / Or
constexpr int MM_NT = 448;  // (seven waves: a 128-channel map of 14x14 / 28x28 is a whole number of rounds, pointwise_body.h)
__global__ __launch_bounds__(MM_NT) void minmax_fwd_kernel(const pnmn_minmax_item* __restrict__ items, int HW, int Cn) {
    const pnmn_minmax_item it = items[blockIdx.x];
    pnmn::pointwise::minmax_fwd<MM_NT>(it, HW);
}

__global__ __launch_bounds__(NT) void minmax_bwd_kernel(const pnmn_minmax_item* __restrict__ items, int HW, int Cn) {
    const pnmn_minmax_item it = items[blockIdx.x];
    pnmn::pointwise::minmax_bwd<NT>(it, HW);
}

// ------------------------------------------------------------------------------------------------
// backward of feats * attn (broadcast)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mask_bwd_kernel(const pnmn_maskbwd_item* __restrict__ items, int HW) {
    const pnmn_maskbwd_item it = items[blockIdx.x];
    const int h = threadIdx.x & 31;
    const int hw = threadIdx.x >> 5;
    for (int p = hw; p < HW; p += 8) {
        const f32x4 dx = *reinterpret_cast<const f32x4*>(it.dx + (size_t)p * C + 4 * h);
        float m = 1.f;
        if (it.attn) {
            m = it.attn[p];
            const f32x4 f = *reinterpret_cast<const f32x4*>(it.feats + (size_t)p * C + 4 * h);
            const float s = half_wave_sum(dot4(dx, f));
            if (h == 0) unsafeAtomicAdd(it.dattn + p, s);
        }
        float* d = it.dfeats + (size_t)p * C + 4 * h;
        unsafeAtomicAdd(d + 0, dx.x * m);
        unsafeAtomicAdd(d + 1, dx.y * m);
        unsafeAtomicAdd(d + 2, dx.z * m);
        unsafeAtomicAdd(d + 3, dx.w * m);
    }
}

// deferred d(feats) of the masked convolutions: see pnmn_feat_grad_gather in the header
__global__ __launch_bounds__(256) void feat_grad_gather_kernel(const pnmn_maskbwd_item* __restrict__ items, int n_items,
                                                               float* __restrict__ gfeat, int HW) {
    const int e = blockIdx.x;
    float* target = gfeat + (size_t)e * HW * C;
    // items are sorted by dfeats: [lo, hi) = those of this example
    int lo = 0, hi = n_items;
    {
        int a = 0, b = n_items;
        while (a < b) {
            const int mid = (a + b) >> 1;
            if (items[mid].dfeats < target) a = mid + 1; else b = mid;
        }
        lo = a;
        b = n_items;
        while (a < b) {
            const int mid = (a + b) >> 1;
            if (items[mid].dfeats <= target) a = mid + 1; else b = mid;
        }
        hi = a;
    }
    if (lo == hi) return;
    const int h = threadIdx.x & 31;    // four channels
    const int row = threadIdx.x >> 5;  // 8 pixels per pass
    const int p0 = blockIdx.y * (HW / gridDim.y), p1 = (blockIdx.y + 1 == gridDim.y) ? HW : p0 + HW / gridDim.y;
    for (int p = p0 + row; p < p1; p += 8) {
        f32x4 acc = *reinterpret_cast<const f32x4*>(target + (size_t)p * C + 4 * h);
        int i = lo;
        for (; i + 4 <= hi; i += 4) {  // four maps in flight
            f32x4 dx[4];
            float m[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                dx[k] = *reinterpret_cast<const f32x4*>(items[i + k].dx + (size_t)p * C + 4 * h);
                m[k] = items[i + k].attn ? items[i + k].attn[p] : 1.f;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) acc += dx[k] * m[k];
        }
        for (; i < hi; ++i) {
            const f32x4 dx = *reinterpret_cast<const f32x4*>(items[i].dx + (size_t)p * C + 4 * h);
            acc += dx * (items[i].attn ? items[i].attn[p] : 1.f);
        }
        *reinterpret_cast<f32x4*>(target + (size_t)p * C + 4 * h) = acc;
    }
}

__global__ __launch_bounds__(256) void accumulate_kernel(const pnmn_axpy_item* __restrict__ items) {
    const pnmn_axpy_item it = items[blockIdx.x];
    const int64_t n4 = it.n >> 2;
    for (int64_t i = threadIdx.x; i < n4; i += blockDim.x) {
        f32x4 d = reinterpret_cast<f32x4*>(it.dst)[i];
        d += reinterpret_cast<const f32x4*>(it.src)[i];
        reinterpret_cast<f32x4*>(it.dst)[i] = d;
    }
    for (int64_t i = (n4 << 2) + threadIdx.x; i < it.n; i += blockDim.x) it.dst[i] += it.src[i];
}

// dst = src (or zeros): a few 100 KB rows per launch, eight workgroups per row
__global__ __launch_bounds__(256) void set_rows_kernel(const pnmn_axpy_item* __restrict__ items) {
    const pnmn_axpy_item it = items[blockIdx.x];
    const int64_t n4 = it.n >> 2;
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int64_t i = blockIdx.y * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.y * blockDim.x)
        reinterpret_cast<f32x4*>(it.dst)[i] = it.src ? reinterpret_cast<const f32x4*>(it.src)[i] : zero;
    if (blockIdx.y == 0)
        for (int64_t i = (n4 << 2) + threadIdx.x; i < it.n; i += blockDim.x) it.dst[i] = it.src ? it.src[i] : 0.f;
}

// ------------------------------------------------------------------------------------------------
// weight transposition for dgrad: [Cout][T][Cin] -> [Cin][T-1-t][Cout]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_weights_kernel(const pnmn_wtrans_item* __restrict__ items) {
    __shared__ float tile[32][33];
    const pnmn_wtrans_item it = items[blockIdx.z];
    const int tiles_ci = (it.cin + 31) / 32;
    const int tiles_co = (it.cout + 31) / 32;
    const int per_tap = tiles_ci * tiles_co;
    // blockIdx.x enumerates (tap, co tile, ci tile); items may have fewer tiles than the grid
    if ((int)blockIdx.x >= per_tap * it.ntaps) return;
    const int tap = blockIdx.x / per_tap;
    const int rem = blockIdx.x % per_tap;
    const int co0 = (rem / tiles_ci) * 32;
    const int ci0 = (rem % tiles_ci) * 32;
    const int tx = threadIdx.x & 31;
    const int ty = threadIdx.x >> 5;  // 0..7
    for (int r = ty; r < 32; r += 8) {
        const int co = co0 + r, ci = ci0 + tx;
        if (co < it.cout && ci < it.cin)
            tile[r][tx] = it.src[((size_t)co * it.ntaps + tap) * it.cin + ci];
    }
    __syncthreads();
    const int rt = it.ntaps - 1 - tap;
    for (int r = ty; r < 32; r += 8) {
        const int ci = ci0 + r, co = co0 + tx;
        if (co < it.cout && ci < it.cin)
            it.dst[((size_t)ci * it.ntaps + rt) * it.cout + co] = tile[tx][r];
    }
}

// global -> LDS copy loop with eight loads in flight per thread (the natural loop waits for each
// 4-byte load before issuing the next: the compiler does not move loads across the LDS stores)
template <typename Load, typename Store>
__device__ __forceinline__ void copy_batched(int n, Load load, Store store) {
    constexpr int NB = 8;
    const int step = blockDim.x;
    for (int i0 = threadIdx.x; i0 < n; i0 += step * NB) {
        float v[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int i = i0 + k * step;
            v[k] = i < n ? load(i) : 0.f;
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int i = i0 + k * step;
            if (i < n) store(i, v[k]);
        }
    }
}

// the same with 16-byte pieces (channel-contiguous sides of the NHWC kernels: a fourth of the address arithmetic
// and of the memory instructions per byte)
template <typename Load, typename Store>
__device__ __forceinline__ void copy_batched4(int n, Load load, Store store) {
    constexpr int NB = 8;
    const int step = blockDim.x;
    for (int i0 = threadIdx.x; i0 < n; i0 += step * NB) {
        float4 v[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int i = i0 + k * step;
            v[k] = i < n ? load(i) : float4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int i = i0 + k * step;
            if (i < n) store(i, v[k]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// NCHW <-> NHWC
// ------------------------------------------------------------------------------------------------
template <bool TO_NHWC>
__global__ __launch_bounds__(256) void layout_kernel(const float* __restrict__ src_base, float* __restrict__ dst,
                                                     int Cn, int HW, int PT, const int64_t* __restrict__ rows) {
    extern __shared__ float tile[];  // [64][PT+1]: 64 channels x one chunk of PT pixels (blockIdx.z)
    const int n = blockIdx.y;
    // (rows: example n of the output is example rows[n] of the source -- a batch subset without a gathered copy)
    const float* src = rows ? src_base + ((size_t)rows[n] - (size_t)n) * Cn * HW : src_base;
    const int c0 = blockIdx.x * 64;
    const int p0 = blockIdx.z * PT;
    const int np = (HW - p0) < PT ? (HW - p0) : PT;
    const int ld = PT + 1;
    const int cw = (Cn - c0) < 64 ? (Cn - c0) : 64;
    if (TO_NHWC) {
        const float* s0 = src + ((size_t)n * Cn + c0) * HW + p0;
        if (((HW | p0 | np) & 3) == 0) {
            // 16-byte loads on the pixel-contiguous side too (a channel's row is HW * 4 bytes: 16-byte aligned when HW is a
            // multiple of 4): with 4-byte loads the three workgroups a CU holds (50 KB of LDS each) keep 24 KB in flight,
            // a third of what the memory system needs per CU to stream at its rate
            const int q = np >> 2;
            copy_batched4(cw * q, [&](int i) { const int c = i / q; return *reinterpret_cast<const float4*>(s0 + (size_t)c * HW + 4 * (i - c * q)); },
                          [&](int i, float4 v) {
                              const int c = i / q;
                              float* t = tile + c * ld + 4 * (i - c * q);
                              t[0] = v.x, t[1] = v.y, t[2] = v.z, t[3] = v.w;
                          });
        } else {
            copy_batched(cw * np, [&](int i) { const int c = i / np; return s0[(size_t)c * HW + (i - c * np)]; },
                         [&](int i, float v) { const int c = i / np; tile[c * ld + (i - c * np)] = v; });
        }
        __syncthreads();
        if (cw == 64 && (Cn & 3) == 0) {
#pragma unroll 8
            for (int i = threadIdx.x; i < 16 * np; i += blockDim.x) {
                const int p = i >> 4, c = 4 * (i & 15);
                const float* t = tile + c * ld + p;
                *reinterpret_cast<float4*>(dst + ((size_t)n * HW + p0 + p) * Cn + c0 + c) = float4{t[0], t[ld], t[2 * ld], t[3 * ld]};
            }
        } else {
#pragma unroll 8
            for (int i = threadIdx.x; i < cw * np; i += blockDim.x) {
                const int p = i / cw, c = i - p * cw;
                dst[((size_t)n * HW + p0 + p) * Cn + c0 + c] = tile[c * ld + p];
            }
        }
    } else {
        if (cw == 64 && (Cn & 3) == 0)
            copy_batched4(16 * np, [&](int i) { return *reinterpret_cast<const float4*>(src + ((size_t)n * HW + p0 + (i >> 4)) * Cn + c0 + 4 * (i & 15)); },
                          [&](int i, float4 v) {
                              float* t = tile + 4 * (i & 15) * ld + (i >> 4);
                              t[0] = v.x, t[ld] = v.y, t[2 * ld] = v.z, t[3 * ld] = v.w;
                          });
        else
            copy_batched(cw * np, [&](int i) { const int p = i / cw; return src[((size_t)n * HW + p0 + p) * Cn + c0 + (i - p * cw)]; },
                         [&](int i, float v) { const int p = i / cw; tile[(i - p * cw) * ld + p] = v; });
        __syncthreads();
#pragma unroll 8
        for (int i = threadIdx.x; i < cw * np; i += blockDim.x) {
            const int c = i / np, p = i - c * np;
            dst[((size_t)n * Cn + c0 + c) * HW + p0 + p] = tile[c * ld + p];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// MaxPool2d(2,2) + Flatten over a ReLU'd NHWC map, output in NCHW-flatten order.  One workgroup = 64
// channels x one band of RB (even) image rows (blockIdx.z): 14x14 maps are one band, 28x28 two.
// ------------------------------------------------------------------------------------------------
template <bool BWD>
__global__ __launch_bounds__(256) void maxpool_kernel(const float* __restrict__ in, const float* __restrict__ dout,
                                                      float* __restrict__ out, int H, int W, int Cn, int RB) {
    extern __shared__ float tile[];  // [RB*W][65]
    const int n = blockIdx.y;
    const int c0 = blockIdx.x * 64;
    const int y0 = blockIdx.z * RB;
    const int rows = (H - y0) < RB ? (H - y0) : RB;
    const int HW = H * W;
    const int NP = rows * W;  // pixels of this band
    const int PH = H / 2, PW = W / 2, PS = PH * PW;
    const int pr0 = y0 / 2;                                   // first pooled row of the band
    const int prn = ((y0 + rows) / 2 < PH ? (y0 + rows) / 2 : PH) - pr0;  // pooled rows of the band
    const int BS = prn * PW;                                  // pooled positions of the band
    const float* src = in + ((size_t)n * HW + (size_t)y0 * W) * Cn;
    copy_batched4(NP * 16, [&](int i) { return *reinterpret_cast<const float4*>(src + (size_t)(i >> 4) * Cn + c0 + 4 * (i & 15)); },
                  [&](int i, float4 v) {
                      float* t = tile + (i >> 4) * 65 + 4 * (i & 15);
                      t[0] = v.x, t[1] = v.y, t[2] = v.z, t[3] = v.w;
                  });
    __syncthreads();
    if (!BWD) {
        float* o = out + (size_t)n * Cn * PS + (size_t)c0 * PS + pr0 * PW;
#pragma unroll 4
        for (int i = threadIdx.x; i < 64 * BS; i += blockDim.x) {
            const int c = i / BS, s = i - c * BS;
            const int y = (s / PW) * 2, x = (s % PW) * 2;
            const float a = tile[(y * W + x) * 65 + c];
            const float b = tile[(y * W + x + 1) * 65 + c];
            const float d = tile[((y + 1) * W + x) * 65 + c];
            const float e = tile[((y + 1) * W + x + 1) * 65 + c];
            o[(size_t)c * PS + s] = fmaxf(fmaxf(a, b), fmaxf(d, e));
        }
    } else {
        // pass 1: pooled gradient -> argmax position, written back into the tile (in place)
        const float* g = dout + (size_t)n * Cn * PS + (size_t)c0 * PS + pr0 * PW;
        for (int i = threadIdx.x; i < 64 * BS; i += blockDim.x) {
            const int c = i / BS, s = i - c * BS;
            const int y = (s / PW) * 2, x = (s % PW) * 2;
            const int q[4] = {y * W + x, y * W + x + 1, (y + 1) * W + x, (y + 1) * W + x + 1};
            float best = tile[q[0] * 65 + c];
            int bi = 0;
#pragma unroll
            for (int k = 1; k < 4; ++k) {
                const float v = tile[q[k] * 65 + c];
                if (v > best) {
                    best = v;
                    bi = k;
                }
            }
            const float gv = best > 0.f ? g[(size_t)c * PS + s] : 0.f;  // ReLU gate of the conv output
#pragma unroll
            for (int k = 0; k < 4; ++k) tile[q[k] * 65 + c] = (k == bi) ? gv : 0.f;
        }
        __syncthreads();
        // rows/cols beyond the pooled area (odd H or W) receive no gradient
        float* d = out + ((size_t)n * HW + (size_t)y0 * W) * Cn;
#pragma unroll 8
        for (int i = threadIdx.x; i < NP * 16; i += blockDim.x) {
            const int p = i >> 4, c = 4 * (i & 15);
            const int y = p / W, x = p - y * W;
            const bool covered = (y0 + y < PH * 2) && (x < PW * 2);
            const float* t = tile + p * 65 + c;
            *reinterpret_cast<float4*>(d + (size_t)p * Cn + c0 + c) =
                covered ? float4{t[0], t[1], t[2], t[3]} : float4{0.f, 0.f, 0.f, 0.f};
        }
    }
}

// ------------------------------------------------------------------------------------------------
// answer loss
// ------------------------------------------------------------------------------------------------
__global__ void answer_loss_kernel(const float* __restrict__ logits, const int64_t* __restrict__ answers,
                                   const int32_t* __restrict__ valid, int64_t* __restrict__ predictions,
                                   float* __restrict__ loss, float* __restrict__ dlogits, int n, int A,
                                   int unknown_index, float scale) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n) return;
    const float* z = logits + (size_t)b * A;
    float mx = z[0];
    int am = 0;
    for (int k = 1; k < A; ++k)
        if (z[k] > mx) {
            mx = z[k];
            am = k;
        }
    float se = 0.f;
    for (int k = 0; k < A; ++k) se += expf(z[k] - mx);
    const float lse = mx + logf(se);
    const bool ok = valid == nullptr || valid[b] != 0;
    if (predictions) predictions[b] = ok ? (int64_t)am : (int64_t)unknown_index;
    float l;
    if (answers)
        l = lse - z[answers[b]];
    else
        l = lse - mx;
    loss[b] = ok ? l : 3.33f;
    if (dlogits) {
        for (int k = 0; k < A; ++k) {
            float gk = 0.f;
            if (ok && answers) gk = (expf(z[k] - lse) - (k == (int)answers[b] ? 1.f : 0.f)) * scale;
            dlogits[(size_t)b * A + k] = gk;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// clamp + Adam
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void clamp_adam_kernel(const pnmn_adam_item* __restrict__ items, float lr,
                                                         float beta1, float beta2, float eps, float wd,
                                                         float clampv) {
    const pnmn_adam_item it = items[blockIdx.y];
    const float bc1 = it.bc1, bc2_sqrt = it.bc2_sqrt;
    const int64_t n4 = it.n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const float step_size = lr / bc1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4 p = reinterpret_cast<f32x4*>(it.param)[i];
        f32x4 g = reinterpret_cast<const f32x4*>(it.grad)[i];
        f32x4 m = reinterpret_cast<f32x4*>(it.exp_avg)[i];
        f32x4 v = reinterpret_cast<f32x4*>(it.exp_avg_sq)[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float gk = g[k];
            if (clampv > 0.f) gk = fminf(fmaxf(gk, -clampv), clampv);
            gk += wd * p[k];
            m[k] = m[k] + (gk - m[k]) * (1.f - beta1);  // lerp, as torch
            v[k] = v[k] * beta2 + (1.f - beta2) * gk * gk;
            const float denom = sqrtf(v[k]) / bc2_sqrt + eps;
            p[k] = p[k] - step_size * (m[k] / denom);
        }
        reinterpret_cast<f32x4*>(it.param)[i] = p;
        reinterpret_cast<f32x4*>(it.exp_avg)[i] = m;
        reinterpret_cast<f32x4*>(it.exp_avg_sq)[i] = v;
    }
    // tail
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < it.n; i += stride) {
        float gk = it.grad[i];
        if (clampv > 0.f) gk = fminf(fmaxf(gk, -clampv), clampv);
        gk += wd * it.param[i];
        const float m = it.exp_avg[i] + (gk - it.exp_avg[i]) * (1.f - beta1);
        const float v = it.exp_avg_sq[i] * beta2 + (1.f - beta2) * gk * gk;
        it.exp_avg[i] = m;
        it.exp_avg_sq[i] = v;
        it.param[i] -= step_size * (m / (sqrtf(v) / bc2_sqrt + eps));
    }
}

inline int last_error() { return (int)hipGetLastError(); }

}  // namespace

#define STREAM(s) static_cast<hipStream_t>(s)

// pixels per workgroup of the layout kernels: the whole map when 64 channels of it fit the LDS
// budget (14x14), otherwise the smallest number of equal chunks that do (28x28: two of 392)
static int layout_chunk(int HW) {
    int parts = 1;
    while ((size_t)64 * ((HW + parts - 1) / parts + 1) * sizeof(float) > 112 * 1024) ++parts;
    return (HW + parts - 1) / parts;
}

template <bool TO_NHWC>
static int launch_layout(const float* src, float* dst, int n, int Cn, int HW, void* stream, const int64_t* rows) {
    if (n <= 0) return 0;
    if (!src || !dst || Cn <= 0 || HW <= 0) return PNMN_EINVAL;
    const int PT = layout_chunk(HW);
    const size_t lds = (size_t)64 * (PT + 1) * sizeof(float);
    static std::atomic<uint64_t> cfg{0};  // (per device: lds_optin.h)
    if (const int e = pnmn::opt_in_lds(reinterpret_cast<const void*>(layout_kernel<TO_NHWC>), 160 * 1024, cfg)) return e;
    hipLaunchKernelGGL(layout_kernel<TO_NHWC>, dim3((Cn + 63) / 64, n, (HW + PT - 1) / PT), dim3(256), lds,
                       STREAM(stream), src, dst, Cn, HW, PT, rows);
    return last_error();
}

// image rows per workgroup of the max-pool kernels (even; [rows*W][65] floats within the LDS budget)
static int pool_band(int H, int W) {
    int rb = (H + 1) & ~1;
    while (rb > 2 && (size_t)rb * W * 65 * sizeof(float) > 112 * 1024) rb = ((rb / 2) + 1) & ~1;
    return rb;
}

// Backward of the ReLU-gated 2x2 max-pool for EVEN maps (14x14, 28x28 -- every shape the engine runs): only the pooled
// gradient -- a quarter of the bytes, and the one operand whose layout ([channel][position], NCHW-flatten order) does not
// match the NHWC maps -- goes through LDS; a thread takes one 2x2 window x 4 channels, reads its four input pieces
// straight from global memory (16 lanes = 256 contiguous bytes of a pixel), picks the first maximum per channel and
// writes the four gradient pieces.  (The kernel above staged the INPUT: ~3 scalar LDS accesses per element, 0.50 of the
// HBM peak; bench.py: hbm_bound_kernels.)
__global__ __launch_bounds__(256) void maxpool_bwd_even_kernel(const float* __restrict__ in, const float* __restrict__ dout,
                                                              float* __restrict__ din, int H, int W, int Cn, int PRB) {
    extern __shared__ float gt[];  // [pooled positions of the band][65]
    const int n = blockIdx.y, c0 = blockIdx.x * 64;
    const int PH = H / 2, PW = W / 2, PS = PH * PW;
    const int pr0 = blockIdx.z * PRB;                              // first pooled row of the band
    const int prn = PH - pr0 < PRB ? PH - pr0 : PRB;
    const int BS = prn * PW;
    const pnmn::gfloat* g = pnmn::as_global(dout) + ((size_t)n * Cn + c0) * PS + (size_t)pr0 * PW;
    copy_batched(64 * BS, [&](int i) { const int c = i / BS; return g[(size_t)c * PS + (i - c * BS)]; },
                 [&](int i, float v) { const int c = i / BS; gt[(i - c * BS) * 65 + c] = v; });
    __syncthreads();
    const pnmn::gfloat* src = pnmn::as_global(in) + (size_t)n * H * W * Cn + c0;
    pnmn::gfloat* dst = pnmn::as_global(din) + (size_t)n * H * W * Cn + c0;
    constexpr int NB = 2;  // windows in flight per thread (8 pieces of 16 bytes)
    for (int i0 = threadIdx.x; i0 < BS * 16; i0 += 256 * NB) {
        f32x4 v[NB][4];
        size_t at[NB][4];
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int i = i0 + k * 256 < BS * 16 ? i0 + k * 256 : BS * 16 - 1;
            const int s = i >> 4, c = 4 * (i & 15);
            const int y = (pr0 + s / PW) * 2, x = (s % PW) * 2;
            at[k][0] = ((size_t)y * W + x) * Cn + c, at[k][1] = at[k][0] + Cn;
            at[k][2] = at[k][0] + (size_t)W * Cn, at[k][3] = at[k][2] + Cn;
#pragma unroll
            for (int q = 0; q < 4; ++q) v[k][q] = pnmn::load4(src + at[k][q]);
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int i = i0 + k * 256;
            if (i >= BS * 16) continue;
            const int s = i >> 4, c = 4 * (i & 15);
            f32x4 o[4] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float best = v[k][0][e];
                int bi = 0;
#pragma unroll
                for (int q = 1; q < 4; ++q)
                    if (v[k][q][e] > best) best = v[k][q][e], bi = q;
                const float gv = best > 0.f ? gt[s * 65 + c + e] : 0.f;  // ReLU gate of the conv output
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q][e] = (q == bi) ? gv : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) pnmn::store4(dst + at[k][q], o[q]);
        }
    }
}

template <bool BWD>
static int launch_pool(const float* in, const float* dout, float* out, int n, int H, int W, int Cn, void* stream) {
    if (n <= 0) return 0;
    if (!in || !out || (BWD && !dout) || (Cn % 64) != 0 || H < 2 || W < 2) return PNMN_EINVAL;
    if (BWD && (H % 2) == 0 && (W % 2) == 0) {
        const int PRB = (H / 2) * (W / 2) <= 64 ? H / 2 : 7;  // pooled rows per workgroup: 14x14 all 7, 28x28 two bands of 7
        hipLaunchKernelGGL(maxpool_bwd_even_kernel, dim3(Cn / 64, n, (H / 2 + PRB - 1) / PRB), dim3(256),
                           (size_t)PRB * (W / 2) * 65 * sizeof(float), STREAM(stream), in, dout, out, H, W, Cn, PRB);
        return last_error();
    }
    const int RB = pool_band(H, W);
    const size_t lds = (size_t)RB * W * 65 * sizeof(float);
    if (lds > 160 * 1024) return PNMN_ESHAPE;
    static std::atomic<uint64_t> cfg{0};  // (per device: lds_optin.h)
    if (const int e = pnmn::opt_in_lds(reinterpret_cast<const void*>(maxpool_kernel<BWD>), 160 * 1024, cfg)) return e;
    hipLaunchKernelGGL(maxpool_kernel<BWD>, dim3(Cn / 64, n, (H + RB - 1) / RB), dim3(256), lds, STREAM(stream), in,
                       dout, out, H, W, Cn, RB);
    return last_error();
}

extern "C" {

int pnmn_abi_version(void) { return 12; }

int pnmn_dot1_sigmoid_fwd(const pnmn_dot1_item* items, int n_items, int HW, void* stream) {
    if (n_items <= 0) return 0;
    if (!items || HW <= 0) return PNMN_EINVAL;
    hipLaunchKernelGGL(dot1_sigmoid_fwd_kernel, dim3(n_items), dim3(256), 0, STREAM(stream), items, HW);
    return last_error();
}

int pnmn_dot1_sigmoid_bwd(const pnmn_dot1_item* items, int n_items, int HW, void* stream) {
    if (n_items <= 0) return 0;
    if (!items || HW <= 0) return PNMN_EINVAL;
    hipLaunchKernelGGL(dot1_sigmoid_bwd_kernel, dim3(n_items), dim3(256), 0, STREAM(stream), items, HW);
    return last_error();
}

int pnmn_same_fwd(const pnmn_same_item* items, int n_items, int HW, void* stream) {
    if (n_items <= 0) return 0;
    if (!items || HW <= 0) return PNMN_EINVAL;
    hipLaunchKernelGGL(same_fwd_kernel, dim3(n_items), dim3(256), 0, STREAM(stream), items, HW);
    return last_error();
}

int pnmn_same_bwd(const pnmn_same_item* items, int n_items, int HW, void* stream) {
    if (n_items <= 0) return 0;
    if (!items || HW <= 0) return PNMN_EINVAL;
    hipLaunchKernelGGL(same_bwd_kernel, dim3(n_items), dim3(256), 0, STREAM(stream), items, HW);
    return last_error();
}

int pnmn_minmax_fwd(const pnmn_minmax_item* items, int n_items, int HW, int Cn, void* stream) {
    if (n_items <= 0) return 0;
    if (!items || HW <= 0 || Cn != C) return PNMN_EINVAL;
    hipLaunchKernelGGL(minmax_fwd_kernel, dim3(n_items), dim3(MM_NT), 0, STREAM(stream), items, HW, Cn);
    return last_error();
}

int pnmn_minmax_bwd(const pnmn_minmax_item* items, int n_items, int HW, int Cn, void* stream) {
    if (n_items <= 0) return 0;
    if (!items || HW <= 0 || Cn != C) return PNMN_EINVAL;
    hipLaunchKernelGGL(minmax_bwd_kernel, dim3(n_items), dim3(256), 0, STREAM(stream), items, HW, Cn);
    return last_error();
}

int pnmn_mask_bwd(const pnmn_maskbwd_item* items, int n_items, int HW, void* stream) {
    if (n_items <= 0) return 0;
    if (!items || HW <= 0) return PNMN_EINVAL;
    hipLaunchKernelGGL(mask_bwd_kernel, dim3(n_items), dim3(256), 0, STREAM(stream), items, HW);
    return last_error();
}

int pnmn_feat_grad_gather(const pnmn_maskbwd_item* items, float* gfeat, int n_items, int n_examples, int HW, void* stream) {
    if (n_items <= 0 || n_examples <= 0) return 0;
    if (!items || !gfeat || HW <= 0) return PNMN_EINVAL;
    // enough workgroups to fill the chip at small batches: pixel ranges of >= 14 rows' worth
    int parts = 1;
    while (n_examples * parts < 512 && parts < 8 && HW / (parts * 2) >= 24) parts *= 2;
    hipLaunchKernelGGL(feat_grad_gather_kernel, dim3(n_examples, parts), dim3(256), 0, STREAM(stream), items, n_items, gfeat, HW);
    return last_error();
}

int pnmn_accumulate(const pnmn_axpy_item* items, int n_items, void* stream) {
    if (n_items <= 0) return 0;
    if (!items) return PNMN_EINVAL;
    hipLaunchKernelGGL(accumulate_kernel, dim3(n_items), dim3(256), 0, STREAM(stream), items);
    return last_error();
}

int pnmn_set_rows(const pnmn_axpy_item* items, int n_items, void* stream) {
    if (n_items <= 0) return 0;
    if (!items) return PNMN_EINVAL;
    hipLaunchKernelGGL(set_rows_kernel, dim3(n_items, 8), dim3(256), 0, STREAM(stream), items);
    return last_error();
}

int pnmn_transpose_weights(const pnmn_wtrans_item* items, int n_items, void* stream) {
    if (n_items <= 0) return 0;
    if (!items) return PNMN_EINVAL;
    // grid.x covers the largest supported weight: 9 taps x (1024/32) x (1024/32) tiles would be
    // wasteful; the caller's weights are at most [1024][1][128] / [128][9][128] / [128][1][256].
    const int max_tiles = 9 * 4 * 4 > 32 * 8 ? 9 * 4 * 4 : 32 * 8;
    hipLaunchKernelGGL(transpose_weights_kernel, dim3(max_tiles, 1, n_items), dim3(256), 0,
                       STREAM(stream), items);
    return last_error();
}

int pnmn_nchw_to_nhwc(const float* src, float* dst, int n, int Cn, int HW, void* stream) {
    return launch_layout<true>(src, dst, n, Cn, HW, stream, nullptr);
}

int pnmn_nchw_to_nhwc_rows(const float* src, float* dst, const int64_t* rows, int n, int Cn, int HW, void* stream) {
    if (!rows) return PNMN_EINVAL;
    return launch_layout<true>(src, dst, n, Cn, HW, stream, rows);
}

int pnmn_nhwc_to_nchw(const float* src, float* dst, int n, int Cn, int HW, void* stream) {
    return launch_layout<false>(src, dst, n, Cn, HW, stream, nullptr);
}

int pnmn_maxpool2_flatten_fwd(const float* in, float* out, int n, int H, int W, int Cn, void* stream) {
    return launch_pool<false>(in, nullptr, out, n, H, W, Cn, stream);
}

int pnmn_maxpool2_flatten_bwd(const float* in, const float* dout, float* din, int n, int H, int W, int Cn,
                              void* stream) {
    return launch_pool<true>(in, dout, din, n, H, W, Cn, stream);
}

int pnmn_answer_loss(const float* logits, const int64_t* answers, const int32_t* valid, int64_t* predictions,
                     float* loss, float* dlogits, int n, int num_answers, int unknown_index, float scale,
                     void* stream) {
    if (n <= 0) return 0;
    if (!logits || !loss || num_answers <= 0) return PNMN_EINVAL;
    hipLaunchKernelGGL(answer_loss_kernel, dim3((n + 63) / 64), dim3(64), 0, STREAM(stream), logits, answers,
                       valid, predictions, loss, dlogits, n, num_answers, unknown_index, scale);
    return last_error();
}

int pnmn_clamp_adam(const pnmn_adam_item* items, int n_items, double lr, double beta1, double beta2, double eps,
                    double weight_decay, double clamp, void* stream) {
    if (n_items <= 0) return 0;
    if (!items) return PNMN_EINVAL;
    // (the bias corrections travel in the items: computed in double by the caller, as torch.optim.Adam does on the host)
    return pnmn_clamp_adam_blocks(items, n_items, lr, beta1, beta2, eps, weight_decay, clamp, 2048, stream);
}

int pnmn_clamp_adam_blocks(const pnmn_adam_item* items, int n_items, double lr, double beta1, double beta2, double eps,
                           double weight_decay, double clamp, int blocks_per_item, void* stream) {
    if (n_items <= 0) return 0;
    if (!items || blocks_per_item <= 0) return PNMN_EINVAL;
    hipLaunchKernelGGL(clamp_adam_kernel, dim3(blocks_per_item, n_items), dim3(256), 0, STREAM(stream), items, (float)lr,
                       (float)beta1, (float)beta2, (float)eps, (float)weight_decay, (float)clamp);
    return last_error();
}

}  // extern "C"
