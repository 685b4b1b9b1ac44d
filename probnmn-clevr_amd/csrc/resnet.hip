// Feature extractor kernels for gfx950: a general NHWC convolution as an implicit GEMM on the fp32 matrix cores with the
// folded batch norm, the residual add and the ReLU of a ResNet block in its epilogue, and the stem's 3x3 / 2 max pool.
//
// The reference extracts the NMN's input features OFFLINE with torchvision's ResNet-101 cut after stage 3
// (/root/reference/scripts/preprocess/extract_features.py:98-105: resnet101(pretrained=True), layer4 / avgpool / fc = Identity,
// eval mode; :124-131 the forward under no_grad).  That network is 1 + 3 x (3 + 4 + 23) + 3 = 94 convolutions of eleven
// different shapes (7x7 / 2 on 3 channels; 1x1 and 3x3, stride 1 and 2, 64-1024 channels, 56x56 to 14x14 maps), so unlike
// the NMN's own convolutions (conv_stream.h: two map sizes, 128-channel blocks) this kernel takes the shape as data:
//
//   GEMM view   out[pixel][cout] = sum_k x_gathered[pixel][k] * w[cout][k],  k = (ky, kx, cin) flattened, cin fastest
//   workgroup   64 output pixels (flattened over the batch) x 64 output channels, 4 waves of 32 x 32 (2 x 2 tiles of
//               v_mfma_f32_16x16x4_f32, weights as the A operand: a lane ends up with 4 consecutive channels of one
//               pixel -> 16-byte NHWC stores)
//   K loop      blocks of 32: every thread gathers two 16-byte pieces of the input tile (4 consecutive channels of one
//               tap; out-of-image taps and the K padding read zeros) and two of the weight tile into registers while
//               the previous block is contracted out of LDS (double buffered: one barrier per block)
//   epilogue    y = acc * scale[c] + shift[c] (+ residual) (ReLU)      -- eval-mode batch norm folded by the host
//
// fp32 throughout (the reference runs the extractor in fp32).  Offline preprocessing, not the training step's hot path:
// built for correctness and a sane fraction of the matrix rate, not tuned per shape.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/probnmn_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BM = 64, BN = 64, BK = 32;
constexpr int LDK = BK + 4;  // floats per LDS row: 144 bytes, so the 16 rows of an operand read spread over the banks

struct ConvArgs {
    const float* x;
    const float* w;  // [Cout][Kpad], k = (ky * kw + kx) * Cin + c, zero beyond K
    const float* scale;
    const float* shift;
    const float* residual;
    float* y;
    int M, H, W, Cin, Ho, Wo, Cout, kw, ntaps, stride, pad, Kpad, relu;
};

__global__ __launch_bounds__(256) void conv2d_nhwc_kernel(const ConvArgs a) {
    __shared__ __attribute__((aligned(16))) float Xs[2][BM][LDK];
    __shared__ __attribute__((aligned(16))) float Ws[2][BN][LDK];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int wm = wave & 1, wn = wave >> 1;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;

    // staging: thread -> (tile row, two of the row's eight 16-byte pieces)
    const int srow = tid >> 2, sq = tid & 3;
    const int p = m0 + srow;
    const bool pvalid = p < a.M;
    const int pc = pvalid ? p : 0;
    const int img = pc / (a.Ho * a.Wo), rem = pc - img * (a.Ho * a.Wo);
    const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
    const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
    const float* ximg = a.x + (size_t)img * a.H * a.W * a.Cin;
    const float* wrow = a.w + (size_t)(n0 + srow) * a.Kpad;

    f32x4 xr[2], wr[2];
    auto fetch = [&](int kb) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int kk = kb * BK + 4 * (sq + 4 * h);
            const int tap = kk / a.Cin, c = kk - tap * a.Cin;
            const int ky = tap / a.kw, kx = tap - ky * a.kw;
            const int iy = iy0 + ky, ix = ix0 + kx;
            const bool ok = pvalid && tap < a.ntaps && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            xr[h] = ok ? *reinterpret_cast<const f32x4*>(ximg + ((size_t)iy * a.W + ix) * a.Cin + c) : f32x4{0.f, 0.f, 0.f, 0.f};
            wr[h] = *reinterpret_cast<const f32x4*>(wrow + kk);
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            *reinterpret_cast<f32x4*>(&Xs[buf][srow][4 * (sq + 4 * h)]) = xr[h];
            *reinterpret_cast<f32x4*>(&Ws[buf][srow][4 * (sq + 4 * h)]) = wr[h];
        }
    };

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nkb = a.Kpad / BK;
    fetch(0);
    stash(0);
    __syncthreads();
    for (int kb = 0; kb < nkb; ++kb) {
        const int buf = kb & 1;
        if (kb + 1 < nkb) fetch(kb + 1);
#pragma unroll
        for (int k16 = 0; k16 < BK / 16; ++k16) {
            // K in a permuted order: lane group g takes k = 16 k16 + 4 g + j in MFMA j -- both operands are one 16-byte read
            f32x4 wa[2], xb[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                wa[t] = *reinterpret_cast<const f32x4*>(&Ws[buf][wn * 32 + t * 16 + li][16 * k16 + 4 * g]);
                xb[t] = *reinterpret_cast<const f32x4*>(&Xs[buf][wm * 32 + t * 16 + li][16 * k16 + 4 * g]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
                        acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[tn][j], xb[tm][j], acc[tn][tm], 0, 0, 0);
        }
        if (kb + 1 < nkb) stash(buf ^ 1);
        __syncthreads();
    }

#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        const int n = n0 + wn * 32 + tn * 16 + 4 * g;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + n), sh = *reinterpret_cast<const f32x4*>(a.shift + n);
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
            const int q = m0 + wm * 32 + tm * 16 + li;
            if (q >= a.M) continue;
            f32x4 v = acc[tn][tm] * sc + sh;
            if (a.residual) v += *reinterpret_cast<const f32x4*>(a.residual + (size_t)q * a.Cout + n);
            if (a.relu) {
                v.x = v.x > 0.f ? v.x : 0.f;
                v.y = v.y > 0.f ? v.y : 0.f;
                v.z = v.z > 0.f ? v.z : 0.f;
                v.w = v.w > 0.f ? v.w : 0.f;
            }
            *reinterpret_cast<f32x4*>(a.y + (size_t)q * a.Cout + n) = v;
        }
    }
}

// nn.MaxPool2d(kernel_size=3, stride=2, padding=1) on NHWC: thread -> (output pixel, 4 channels)
__global__ __launch_bounds__(256) void maxpool3x3s2_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W,
                                                                int C, int Ho, int Wo) {
    const size_t total = (size_t)N * Ho * Wo * (C / 4);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % (C / 4));
        const size_t q = i / (C / 4);
        const int ox = (int)(q % Wo), oy = (int)((q / Wo) % Ho), img = (int)(q / ((size_t)Wo * Ho));
        f32x4 m = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = 2 * oy - 1 + ky, ix = 2 * ox - 1 + kx;
                if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) continue;
                const f32x4 v = *reinterpret_cast<const f32x4*>(x + (((size_t)img * H + iy) * W + ix) * C + 4 * c4);
                m.x = v.x > m.x ? v.x : m.x;
                m.y = v.y > m.y ? v.y : m.y;
                m.z = v.z > m.z ? v.z : m.z;
                m.w = v.w > m.w ? v.w : m.w;
            }
        *reinterpret_cast<f32x4*>(y + q * C + 4 * c4) = m;
    }
}

}  // namespace

extern "C" int pnmn_conv2d_nhwc(const pnmn_conv2d_desc* d, void* stream) {
    if (!d || !d->x || !d->w || !d->scale || !d->shift || !d->y) return PNMN_EINVAL;
    if (d->N <= 0) return 0;
    if (d->Cin <= 0 || (d->Cin & 3) || d->Cout <= 0 || (d->Cout % BN) || d->kh < 1 || d->kw < 1 || d->stride < 1 || d->pad < 0)
        return PNMN_ESHAPE;
    const int Ho = (d->H + 2 * d->pad - d->kh) / d->stride + 1, Wo = (d->W + 2 * d->pad - d->kw) / d->stride + 1;
    if (Ho != d->Ho || Wo != d->Wo || Ho <= 0 || Wo <= 0) return PNMN_ESHAPE;
    const long M = (long)d->N * Ho * Wo;
    if (M > (1L << 30)) return PNMN_ESHAPE;
    const int K = d->kh * d->kw * d->Cin;
    ConvArgs a{d->x, d->w, d->scale, d->shift, d->residual, d->y, (int)M, d->H, d->W, d->Cin, Ho, Wo, d->Cout, d->kw, d->kh * d->kw,
               d->stride, d->pad, (K + BK - 1) / BK * BK, d->relu};
    hipLaunchKernelGGL(conv2d_nhwc_kernel, dim3((unsigned)((M + BM - 1) / BM), d->Cout / BN), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}

extern "C" int pnmn_conv2d_weight_floats(int Cout, int Cin, int kh, int kw) {
    if (Cout <= 0 || Cin <= 0 || kh < 1 || kw < 1) return PNMN_EINVAL;
    const long K = (long)kh * kw * Cin, total = (long)Cout * ((K + BK - 1) / BK * BK);
    return total > 0x7fffffffL ? PNMN_ESHAPE : (int)total;
}

extern "C" int pnmn_maxpool3x3s2_nhwc(const float* x, float* y, int N, int H, int W, int C, void* stream) {
    if (N <= 0) return 0;
    if (!x || !y) return PNMN_EINVAL;
    if (H < 1 || W < 1 || C <= 0 || (C & 3)) return PNMN_ESHAPE;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const size_t total = (size_t)N * Ho * Wo * (C / 4);
    const unsigned grid = (unsigned)((total + 255) / 256 < 256 * 16 ? (total + 255) / 256 : 256 * 16);
    hipLaunchKernelGGL(maxpool3x3s2_nhwc_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, N, H, W, C, Ho, Wo);
    return (int)hipGetLastError();
}
