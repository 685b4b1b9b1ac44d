// Grouped convolution weight gradient for gfx950.
//
//   dW[n][tap][c] += sum over the job's items, over pixels p:
//                    dy[p][n] * (gate[p][n] > 0) * (x * xmask)[shift(p, tap, dil)][c]
//
// GEMM view per tap: M = 128 output channels, N = 128 input channels, K = pixels (196 per item,
// concatenated over every item of the job -- the items of a job share one weight).  One
// workgroup (4 waves, one per SIMD) owns a [128 x TAPS x 128] slab of dW for one job: wave w owns
// the 64x64 quadrant (w>>1, w&1) for all TAPS taps = TAPS*16 accumulators of 16x16, which stay in
// registers across all of the job's items; at the end the slab is added into dW with fp32 atomics
// (jobs that split one weight's items over several workgroups meet there).
//
// Both operands are read from NHWC LDS tiles with one ds_read_b128 per lane and k-step: lane
// (li, g) takes pixel k = 4*step + g and channels 4*li..4*li+3, i.e. value j of the load belongs
// to the "interleaved" sub-tile j (channels {4r + j}).  That is conflict-free without swizzling:
// a lane group reads 16 consecutive 16-byte slots of (at most two) pixel rows.
// LDS: x tile [(HW+1)][128] (row HW = zeros for out-of-image taps) + dy in two pixel-halves.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/probnmn_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int CB = 128;

template <int H, int W, int TAPS>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(
    const pnmn_wgrad_item* __restrict__ items, const pnmn_wgrad_job* __restrict__ jobs, int ntaps,
    int cin_blocks, int x_stride, int dy_stride) {
    constexpr int HW = H * W;
    constexpr int KSTEPS = HW / 4;               // HW % 4 == 0 for 14x14 and 28x28
    constexpr int HALF0 = ((KSTEPS + 1) / 2) * 4;  // pixels in the first dy half
    static_assert(HW % 4 == 0, "pixel count must be a multiple of the MFMA k (4)");

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* xl = reinterpret_cast<float*>(smem_raw);  // [(HW+1)][128]
    float* dl = xl + (HW + 1) * CB;                  // [HALF0][128]

    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    const int li = lane & 15;
    const int g = lane >> 4;
    const int co_half = wave >> 1;
    const int ci_half = wave & 1;

    // blockIdx.y enumerates (cout block, cin block, tap group)
    const int tap_groups = ntaps / TAPS;
    int by = blockIdx.y;
    const int tg = by % tap_groups;
    by /= tap_groups;
    const int cib = by % cin_blocks;
    const int cob = by / cin_blocks;
    const int tap0 = tg * TAPS;
    const int cin_total = cin_blocks * CB;

    const pnmn_wgrad_job job = jobs[blockIdx.x];

    f32x4 acc[TAPS][4][4];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bias_acc = 0.f;  // per-thread partial of dbias for channel (tid & 127), pixels split in 2

    if (tid < 32) reinterpret_cast<f32x4*>(xl + HW * CB)[tid] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int ii = job.item_begin; ii < job.item_end; ++ii) {
        const pnmn_wgrad_item it = items[ii];
        const int dil = it.dilation;
        const float* xsrc = (it.x2 != nullptr && cib > 0) ? it.x2 : it.x + cib * CB;
        __syncthreads();  // previous item fully consumed
        for (int idx = tid; idx < HW * 32; idx += 256) {
            const int p = idx >> 5;
            const int s = idx & 31;
            f32x4 v = *reinterpret_cast<const f32x4*>(xsrc + (size_t)p * x_stride + s * 4);
            if (it.xmask) v *= it.xmask[p];
            *reinterpret_cast<f32x4*>(xl + p * CB + s * 4) = v;
        }
        for (int half = 0; half < 2; ++half) {
            const int pbeg = half ? HALF0 : 0;
            const int pend = half ? HW : HALF0;
            if (half) __syncthreads();  // first half consumed before overwriting dl
            for (int idx = tid; idx < (pend - pbeg) * 32; idx += 256) {
                const int p = pbeg + (idx >> 5);
                const int s = idx & 31;
                const size_t o = (size_t)p * dy_stride + cob * CB + s * 4;
                f32x4 v = *reinterpret_cast<const f32x4*>(it.dy + o);
                if (it.gate) {
                    const f32x4 gt = *reinterpret_cast<const f32x4*>(it.gate + o);
                    v.x = gt.x > 0.f ? v.x : 0.f;
                    v.y = gt.y > 0.f ? v.y : 0.f;
                    v.z = gt.z > 0.f ? v.z : 0.f;
                    v.w = gt.w > 0.f ? v.w : 0.f;
                }
                *reinterpret_cast<f32x4*>(dl + (p - pbeg) * CB + s * 4) = v;
            }
            __syncthreads();

            // bias gradient: thread t sums channel t&127 over alternating pixels
            if (job.dbias != nullptr && cib == 0 && tg == 0) {
                const int c = tid & 127;
                for (int p = (tid >> 7); p < pend - pbeg; p += 2) bias_acc += dl[p * CB + c];
            }

            for (int p0 = pbeg; p0 < pend; p0 += 4) {
                const int p = p0 + g;  // this lane's k (pixel)
                const f32x4 a = *reinterpret_cast<const f32x4*>(dl + (p - pbeg) * CB + co_half * 64 + li * 4);
                const int y = p / W;
                const int x = p % W;
#pragma unroll
                for (int t = 0; t < TAPS; ++t) {
                    int q = p;
                    if (TAPS == 3) {
                        const int tap = tap0 + t;
                        const int yy = y + (tap / 3 - 1) * dil;
                        const int xx = x + (tap % 3 - 1) * dil;
                        const bool ok = ((unsigned)yy < (unsigned)H) && ((unsigned)xx < (unsigned)W);
                        q = ok ? yy * W + xx : HW;
                    }
                    const f32x4 b = *reinterpret_cast<const f32x4*>(xl + q * CB + ci_half * 64 + li * 4);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc[t][i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b.x, acc[t][i][0], 0, 0, 0);
                        acc[t][i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b.y, acc[t][i][1], 0, 0, 0);
                        acc[t][i][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b.z, acc[t][i][2], 0, 0, 0);
                        acc[t][i][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b.w, acc[t][i][3], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- write-out: acc[t][i][j][r] = dW[cout = cob*128 + co_half*64 + 4*(4g+r) + i]
    //                                     [tap0+t][cin = cib*128 + ci_half*64 + 4*li + j]
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cout = cob * CB + co_half * 64 + 4 * (4 * g + r) + i;
                float* dst = job.dw + ((size_t)cout * ntaps + tap0 + t) * cin_total + cib * CB +
                             ci_half * 64 + 4 * li;
                unsafeAtomicAdd(dst + 0, acc[t][i][0][r]);
                unsafeAtomicAdd(dst + 1, acc[t][i][1][r]);
                unsafeAtomicAdd(dst + 2, acc[t][i][2][r]);
                unsafeAtomicAdd(dst + 3, acc[t][i][3][r]);
            }
        }
    }
    if (job.dbias != nullptr && cib == 0 && tg == 0) {
        unsafeAtomicAdd(job.dbias + cob * CB + (tid & 127), bias_acc);
    }
}

template <int H, int W, int TAPS>
int launch_wgrad(const pnmn_wgrad_item* items, const pnmn_wgrad_job* jobs, int n_jobs, int ntaps,
                 int cin_blocks, int cout_blocks, int x_stride, int dy_stride, hipStream_t stream) {
    constexpr int HW = H * W;
    constexpr int HALF0 = (((HW / 4) + 1) / 2) * 4;
    constexpr size_t lds_bytes = (size_t)(HW + 1 + HALF0) * CB * sizeof(float);
    static bool configured = false;
    auto kern = conv_wgrad_kernel<H, W, TAPS>;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    dim3 grid(n_jobs, cout_blocks * cin_blocks * (ntaps / TAPS));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, stream, items, jobs, ntaps, cin_blocks,
                       x_stride, dy_stride);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" int pnmn_conv_wgrad(const pnmn_wgrad_item* items, const pnmn_wgrad_job* jobs, int n_jobs,
                               int H, int W, int ntaps, int cin_blocks, int cout_blocks,
                               int x_stride, int dy_stride, void* stream) {
    if (n_jobs <= 0) return 0;
    if (!items || !jobs || cin_blocks < 1 || cout_blocks < 1 || (ntaps != 9 && ntaps != 1))
        return PNMN_EINVAL;
    if ((x_stride & 3) || (dy_stride & 3)) return PNMN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (H == 14 && W == 14) {
        if (ntaps == 9)
            return launch_wgrad<14, 14, 3>(items, jobs, n_jobs, ntaps, cin_blocks, cout_blocks,
                                           x_stride, dy_stride, s);
        return launch_wgrad<14, 14, 1>(items, jobs, n_jobs, ntaps, cin_blocks, cout_blocks, x_stride,
                                       dy_stride, s);
    }
    return PNMN_ESHAPE;
}
