// Bodies of the point-wise module kernels (one-channel head, SameModule, And / Or; forward and backward), for a
// workgroup of NT threads working on ONE item (pointwise.hip launches them one workgroup of 256 threads per item).
// NT / 32 half-waves walk the pixels (a half-wave = 32 lanes x float4 = one 512-byte pixel row); reductions over the
// pixels go through `scratch` (LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#include "../../include/probnmn_hip.h"
#include "global_ptr.h"

namespace pnmn {
namespace pointwise {

constexpr int C = PNMN_CHANNELS;
// floats of LDS scratch the bodies need: the arg-max of Same (two words per thread) or the partial sums of a backward
// ([NT / 32][2 C + 2])
template <int NT>
constexpr int scratch_floats() { return (2 * NT > (NT / 32) * (2 * C + 2)) ? 2 * NT : (NT / 32) * (2 * C + 2); }

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

// ---- one-channel head: conv1x1 (128 -> 1) + sigmoid (nmn_modules.py:86,167) ------------------------------------
template <int NT>
__device__ __forceinline__ void dot1_fwd(const pnmn_dot1_item& it, int HW) {
    constexpr int NHW = NT / 32;  // half-waves
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

template <int NT>
__device__ __forceinline__ void dot1_bwd(const pnmn_dot1_item& it, int HW, float* red /* [NHW][C + 1] */) {
    constexpr int NHW = NT / 32;  // half-waves
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
template <int NT>
__device__ __forceinline__ int first_argmax(const float* __restrict__ attn, int HW, float* sval, int* sidx) {
    // first maximum in scan order (what max_pool2d(return_indices=True) reports)
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int p = threadIdx.x; p < HW; p += NT) {
        const float v = attn[p];
        if (v > best || (v != v && best == best)) {
            best = v;
            bi = p;
        }
    }
    sval[threadIdx.x] = best;
    sidx[threadIdx.x] = bi;
    __syncthreads();
    for (int s = NT / 2; s > 0; s >>= 1) {
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

template <int NT>
__device__ __forceinline__ void same_fwd(const pnmn_same_item& it, int HW, float* scratch) {
    constexpr int NHW = NT / 32;  // half-waves
    const int j = first_argmax<NT>(it.attn, HW, scratch, reinterpret_cast<int*>(scratch + NT));
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

template <int NT>
__device__ __forceinline__ void same_bwd(const pnmn_same_item& it, int HW, float* scratch) {
    constexpr int NHW = NT / 32;  // half-waves
    const int j = first_argmax<NT>(it.attn, HW, scratch, reinterpret_cast<int*>(scratch + NT));
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
template <int NT>
__device__ __forceinline__ void minmax_fwd(const pnmn_minmax_item& it, int HW) {
    const int oc = it.a_channels > it.b_channels ? it.a_channels : it.b_channels;
    // torch.min/max propagate NaN; fminf/fmaxf would not
    auto pick = [&](float a, float b) { return (a != a || b != b) ? NAN : (it.is_max ? (a > b ? a : b) : (a < b ? a : b)); };
    if (oc == C) {
        // a 128-channel result: 16 bytes per thread, eight of them requested before the first is used (the first version
        // -- one float and one integer division per element -- ran at half the HBM rate)
        constexpr int NB = 7;  // (NT = 448: 196 pixels x 32 pieces = 6 272 = 448 x 7 x 2 -- two full rounds per workgroup, and
                               // 28x28: eight; with 256 threads it was 3.5 rounds, the last one half empty)
        const int n4 = HW * (C / 4);
        const gfloat* ga = as_global(it.a);
        const gfloat* gb = as_global(it.b);
        for (int i0 = threadIdx.x; i0 < n4; i0 += NT * NB) {
            f32x4 a[NB], b[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int i = i0 + k * NT;
                if (i >= n4) continue;
                const int p = i >> 5, c = (i & 31) * 4;
                if (it.a_channels == 1) {
                    const float v = ga[p];
                    a[k] = f32x4{v, v, v, v};
                } else {
                    a[k] = load4(ga + (size_t)p * C + c);
                }
                if (it.b_channels == 1) {
                    const float v = gb[p];
                    b[k] = f32x4{v, v, v, v};
                } else {
                    b[k] = load4(gb + (size_t)p * C + c);
                }
            }
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int i = i0 + k * NT;
                if (i < n4)
                    store4(as_global(it.out) + (size_t)i * 4,
                           f32x4{pick(a[k].x, b[k].x), pick(a[k].y, b[k].y), pick(a[k].z, b[k].z), pick(a[k].w, b[k].w)});
            }
        }
        return;
    }
    const int n = HW * oc;
    for (int i = threadIdx.x; i < n; i += NT) {
        const int p = i / oc;
        const int c = i - p * oc;
        const float a = it.a[it.a_channels == 1 ? p : p * C + c];
        const float b = it.b[it.b_channels == 1 ? p : p * C + c];
        it.out[i] = pick(a, b);
    }
}

template <int NT>
__device__ __forceinline__ void minmax_bwd(const pnmn_minmax_item& it, int HW) {
    constexpr int NHW = NT / 32;  // half-waves
    const int oc = it.a_channels > it.b_channels ? it.a_channels : it.b_channels;
    const int h = threadIdx.x & 31, hw = threadIdx.x >> 5;
    if (oc == 1) {
        for (int p = threadIdx.x; p < HW; p += NT) {
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

}  // namespace pointwise
}  // namespace pnmn
