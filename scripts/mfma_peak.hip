// Calibration: which part of the conv kernel's contraction loop costs the distance to the fp32 MFMA
// peak?  512 threads per workgroup (2 waves per SIMD), 13 accumulators per wave, built up step by
// step towards conv_body.h:  V0 operands in registers; V1 A fragments read from LDS right before use;
// V2 the half-step software pipeline of conv_body.h; V3 + the weight fragment from global memory;
// V4 + the per-tap row-offset arithmetic.      hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int MT = 13, MH = 7;

#define MFMA(a, b, c) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)

template <int V>
__global__ __launch_bounds__(512) void k(float* out, const float* __restrict__ w, int steps) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    for (int i = tid; i < 204 * 128; i += 512) lds[i] = (float)(i & 7) * 1e-3f;
    __syncthreads();
    f32x4 acc[MT];
    for (int i = 0; i < MT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    int rb[MT];
    for (int mt = 0; mt < MT; ++mt) {
        const int q = (mt * 16 + li) % 196;
        rb[mt] = q * 128 + ((g & 1) << 5 | (g >> 1) << 6) + (q & 7);
    }
    const float* wrow = w + (size_t)(blockIdx.x & 127) * 9 * 128 + 4 * g + li * 1152;
    f32x4 afrag[MT];
    f32x4 bw = *reinterpret_cast<const f32x4*>(wrow);
    auto load = [&](int lo, int hi, int kb) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            if (mt >= lo && mt < hi) afrag[mt] = *reinterpret_cast<const f32x4*>(lds + (rb[mt] & ~7) + (((kb ^ rb[mt]) & 7) << 2));
    };
    auto mma = [&](int lo, int hi, const f32x4 b) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) if (mt >= lo && mt < hi) MFMA(b.x, afrag[mt].x, acc[mt]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) if (mt >= lo && mt < hi) MFMA(b.y, afrag[mt].y, acc[mt]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) if (mt >= lo && mt < hi) MFMA(b.z, afrag[mt].z, acc[mt]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) if (mt >= lo && mt < hi) MFMA(b.w, afrag[mt].w, acc[mt]);
    };
    if (V == 0) {
        for (int mt = 0; mt < MT; ++mt) afrag[mt] = f32x4{1.f + li, 2.f, 3.f, 4.f};
        for (int s = 0; s < steps; ++s) { mma(0, MT, bw); bw.x += 1e-9f; }
    } else if (V == 1) {
        for (int s = 0; s < steps; ++s) { load(0, MT, s & 7); mma(0, MT, bw); }
    } else {
        load(0, MT, 0);
        f32x4 bnext = bw;
        for (int tap = 0; tap < steps / 8; ++tap) {
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                const int kn = (kb + 1) & 7;
                if (V >= 3) bnext = *reinterpret_cast<const f32x4*>(wrow + ((tap % 9) * 128 + kn * 16));
                const f32x4 b = bw;
                mma(0, MH, b);
                __builtin_amdgcn_sched_barrier(0);
                if (V >= 4 && kb == 7) {
                    const int dy = ((tap + 1) % 9 / 3 - 1), dx = ((tap + 1) % 9 % 3 - 1);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const int p = mt * 16 + li, yy = p / 14 + dy, xx = p % 14 + dx;
                        const bool ok = ((unsigned)yy < 14u) && ((unsigned)xx < 14u);
                        const int qv = yy * 14 + xx, q = ok ? qv : 200 + (qv & 7);
                        rb[mt] = q * 128 + ((g & 1) << 5 | (g >> 1) << 6) + (q & 7);
                    }
                }
                load(0, MH, kn);
                __builtin_amdgcn_sched_barrier(0);
                mma(MH, MT, b);
                __builtin_amdgcn_sched_barrier(0);
                load(MH, MT, kn);
                __builtin_amdgcn_sched_barrier(0);
                bw = (V >= 3) ? bnext : b;
            }
        }
    }
    f32x4 s = acc[0];
    for (int i = 1; i < MT; ++i) s += acc[i];
    out[blockIdx.x * 512 + tid] = s.x + s.y + s.z + s.w;
}

template <int V>
void run() {
    float *out, *w;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&w, 128 * 1152 * 16 * 4);
    hipMemset(w, 0, 128 * 1152 * 16 * 4);
    const int steps = 9 * 8 * 8;  // one stem-conv item
    const size_t lds = 204 * 128 * 4;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<V>, dim3(256), dim3(512), lds, 0, out, w, 16);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<V>, dim3(256), dim3(512), lds, 0, out, w, steps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * 8 * steps * 52 * 2048.0;
    printf("V%d: %.3f ms  %.1f TFLOP/s\n", V, ms, flops / ms / 1e9);
    hipFree(out);
    hipFree(w);
}

int main() {
    run<0>();
    run<1>();
    run<2>();
    run<3>();
    run<4>();
    return 0;
}
