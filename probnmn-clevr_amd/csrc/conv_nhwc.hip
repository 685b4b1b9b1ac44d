// Grouped NHWC convolution (3x3 with dilation, or 1x1) for gfx950 -- forward and data-gradient.
//
// One workgroup = one (example, convolution) item x one block of 128 output channels.
// The whole H*W x 128-channel input tile of the example is staged ONCE into LDS (14x14: 98 KiB of
// the CU's 160 KiB) with the fused prologue (attention mask multiply, or ReLU-backward gate), and
// all nine taps read their A operands straight out of that image at shifted pixel rows -- no
// im2col, no halo copy: out-of-image taps point at one extra all-zero row.  Weights are never
// staged: each wave owns 16 output channels and streams its own [16][tap][cin] slice from L2
// into registers one tap ahead of use (nobody else in the workgroup needs that slice, so an LDS
// round trip would be pure overhead).
//
// Math is exact fp32 on the matrix cores: v_mfma_f32_16x16x4_f32 (weights as the A operand so
// that each lane ends up with 4 consecutive output channels of one pixel -> 16-byte stores).
// 16x16 tiles because H*W = 196 = 12.25 x 16: 13 m-tiles waste 5.8 % (32x32 tiles would waste
// 12.5 %).  K is consumed in a permuted order (lane group g takes channels 4g..4g+3 of each
// 16-channel block, one per MFMA) so that both operands are single 16-byte loads.
//
// Load balance: a launch rarely has exactly k x 256 items (programs differ in length, so the
// later levels of a step have few active examples).  The K-split variants cut one item into
// KSPLIT workgroups of 16*8/KSPLIT output channels each; inside a workgroup the 8 waves then
// split the 128 input channels of every tap KSPLIT ways and are summed through LDS at the end.
// Each workgroup still stages the whole input tile, so the launcher picks the smallest split
// that fills the chip (see pick_ksplit).
//
// LDS image: row p (pixel) = 128 floats = 32 slots of 16 B; slot s is stored at s ^ (p & 15), which
// makes the 16 pixel rows a ds_read_b128 lane-group touches land on 16 different bank slots.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/probnmn_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int CB = 128;  // channels per block (input chunk and output block)

template <int H, int W, int KSPLIT>
__global__ __launch_bounds__(512) void conv_nhwc_kernel(
    const pnmn_conv_item* __restrict__ items, int cin_chunks, int ntaps, int in_stride,
    int out_stride, int relu) {
    constexpr int HW = H * W;
    constexpr int MT = (HW + 15) / 16;
    constexpr int NT = 8 / KSPLIT;   // 16-channel output tiles per workgroup
    constexpr int KB = 8 / KSPLIT;   // 16-channel input blocks per wave and tap
    constexpr int NTHREADS = 512;

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* lds = reinterpret_cast<float*>(smem_raw);  // [(HW+1)][128], row HW is zero

    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    const int li = lane & 15;
    const int g = lane >> 4;

    const int item_id = blockIdx.x / KSPLIT;
    const int nsub = blockIdx.x % KSPLIT;
    const int nt = wave % NT;
    const int ks = wave / NT;  // which slice of the input channels this wave contracts
    const pnmn_conv_item it = items[item_id];
    const int n0 = blockIdx.y * CB + (nsub * NT + nt) * 16;  // this wave's 16 out channels
    const int cin_total = cin_chunks * CB;
    const int dil = it.dilation;

    // pixel handled by this lane in each m-tile (as the MFMA "column" index)
    int py[MT], px[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int p = mt * 16 + li;
        py[mt] = (p < HW) ? p / W : -100000;
        px[mt] = p % W;
    }

    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // weight row of this lane: output channel n0 + li, channels 4g.. of each 16-block
    const float* wrow = it.weight + (size_t)(n0 + li) * ntaps * cin_total + 4 * g + ks * KB * 16;

    if (tid < 32) reinterpret_cast<f32x4*>(lds + HW * CB)[tid] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int chunk = 0; chunk < cin_chunks; ++chunk) {
        // ---- stage this 128-channel chunk of the input into LDS (fused prologue) ----
        const float* src = (it.in2 != nullptr && chunk > 0) ? it.in2 : it.in + chunk * CB;
        const float* gsrc = it.gate ? it.gate + chunk * CB : nullptr;
        if (chunk > 0) __syncthreads();  // everyone done reading the previous chunk
        for (int idx = tid; idx < HW * 32; idx += NTHREADS) {
            const int p = idx >> 5;
            const int s = idx & 31;
            f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)p * in_stride + s * 4);
            if (it.mask) {
                const float m = it.mask[p];
                v *= m;
            }
            if (gsrc) {
                const f32x4 gt = *reinterpret_cast<const f32x4*>(gsrc + (size_t)p * in_stride + s * 4);
                v.x = gt.x > 0.f ? v.x : 0.f;
                v.y = gt.y > 0.f ? v.y : 0.f;
                v.z = gt.z > 0.f ? v.z : 0.f;
                v.w = gt.w > 0.f ? v.w : 0.f;
            }
            *reinterpret_cast<f32x4*>(lds + p * CB + ((s ^ (p & 15)) << 2)) = v;
        }
        __syncthreads();

        const float* wchunk = wrow + chunk * CB;
        // prefetch tap 0 weights
        f32x4 bcur[KB];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
            bcur[kb] = *reinterpret_cast<const f32x4*>(wchunk + kb * 16);

        for (int tap = 0; tap < ntaps; ++tap) {
            // prefetch next tap's weights (clamped: the last iteration re-reads its own)
            const int tnext = (tap + 1 < ntaps) ? tap + 1 : tap;
            f32x4 bnext[KB];
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
                bnext[kb] = *reinterpret_cast<const f32x4*>(wchunk + (size_t)tnext * cin_total + kb * 16);

            int dy = 0, dx = 0;
            if (ntaps == 9) {
                dy = (tap / 3 - 1) * dil;
                dx = (tap % 3 - 1) * dil;
            }
            // LDS float offset of the shifted pixel row for each m-tile (+ swizzle bits)
            int rowbase[MT];  // q*128 + ((g ^ (q&3)) << 2), low 2 bits carry (q>>2)&3
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int yy = py[mt] + dy;
                const int xx = px[mt] + dx;
                const bool ok = ((unsigned)yy < (unsigned)H) && ((unsigned)xx < (unsigned)W);
                const int q = ok ? yy * W + xx : HW;
                rowbase[mt] = q * CB + ((g ^ (q & 3)) << 2) + ((q >> 2) & 3);
            }
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const f32x4 b = bcur[kb];
                const int kbg = ks * KB + kb;  // 16-channel block index inside the 128-channel chunk
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int rb = rowbase[mt];
                    const int off = (rb & ~3) + ((kbg ^ (rb & 3)) << 4);
                    const f32x4 a = *reinterpret_cast<const f32x4*>(lds + off);
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.x, a.x, acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.y, a.y, acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.z, a.z, acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.w, a.w, acc[mt], 0, 0, 0);
                }
            }
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) bcur[kb] = bnext[kb];
        }
    }

    if (KSPLIT > 1) {
        // sum the KSPLIT partial accumulators of each output tile through LDS (input image is dead)
        __syncthreads();
        f32x4* red = reinterpret_cast<f32x4*>(lds);
        if (ks > 0) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) red[(((ks - 1) * NT + nt) * MT + mt) * 64 + lane] = acc[mt];
        }
        __syncthreads();
        if (ks > 0) return;
#pragma unroll 1
        for (int k2 = 1; k2 < KSPLIT; ++k2) {
            const f32x4* src = red + (((k2 - 1) * NT + nt) * MT) * 64 + lane;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] += src[mt * 64];
        }
    }

    // ---- epilogue: lane holds out channels n0+4g..+3 of pixel mt*16+li ----
    f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (it.bias) bias4 = *reinterpret_cast<const f32x4*>(it.bias + n0 + 4 * g);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int p = mt * 16 + li;
        if (p < HW) {
            f32x4 v = acc[mt] + bias4;
            if (relu) {
                v.x = fmaxf(v.x, 0.f);
                v.y = fmaxf(v.y, 0.f);
                v.z = fmaxf(v.z, 0.f);
                v.w = fmaxf(v.w, 0.f);
            }
            f32x4* dst = reinterpret_cast<f32x4*>(it.out + (size_t)p * out_stride + n0 + 4 * g);
            if (it.flags & PNMN_CONV_ACCUMULATE) v += *dst;
            *dst = v;
        }
    }
}

template <int H, int W, int KSPLIT>
int launch_conv_k(const pnmn_conv_item* items, int n_items, int cin_chunks, int ntaps, int in_stride,
                  int out_stride, int cout_blocks, int relu, hipStream_t stream) {
    constexpr size_t lds_bytes = (size_t)(H * W + 1) * CB * sizeof(float);
    static_assert((size_t)(KSPLIT - 1) * (8 / KSPLIT) * ((H * W + 15) / 16) * 64 * 16 <= lds_bytes,
                  "reduction scratch must fit in the input image");
    static bool configured = false;
    auto kern = conv_nhwc_kernel<H, W, KSPLIT>;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    dim3 grid(n_items * KSPLIT, cout_blocks);
    hipLaunchKernelGGL(kern, grid, dim3(512), lds_bytes, stream, items, cin_chunks, ntaps, in_stride,
                       out_stride, relu);
    return (int)hipGetLastError();
}

// Smallest makespan of ceil(workgroups / CUs) rounds, each costing (contraction / split + staging).
// Relative costs only: one tap of one 128-channel chunk = 1 unit; staging a chunk ~ 0.5 unit.
inline int pick_ksplit(int n_items, int cout_blocks, int cin_chunks, int ntaps) {
    const double work = (double)ntaps * cin_chunks;
    const double overhead = 0.5 * cin_chunks + 0.25;
    int best = 1;
    double best_t = 1e30;
    for (int s = 1; s <= 8; s *= 2) {
        const long wgs = (long)n_items * cout_blocks * s;
        const long rounds = (wgs + 255) / 256;
        const double t = rounds * (work / s + overhead);
        if (t < best_t * 0.97) {  // prefer the smaller split unless the gain is real
            best_t = t;
            best = s;
        }
    }
    return best;
}

template <int H, int W>
int launch_conv(const pnmn_conv_item* items, int n_items, int cin_chunks, int ntaps, int in_stride,
                int out_stride, int cout_blocks, int relu, hipStream_t stream) {
    switch (pick_ksplit(n_items, cout_blocks, cin_chunks, ntaps)) {
        case 8:
            return launch_conv_k<H, W, 8>(items, n_items, cin_chunks, ntaps, in_stride, out_stride,
                                          cout_blocks, relu, stream);
        case 4:
            return launch_conv_k<H, W, 4>(items, n_items, cin_chunks, ntaps, in_stride, out_stride,
                                          cout_blocks, relu, stream);
        case 2:
            return launch_conv_k<H, W, 2>(items, n_items, cin_chunks, ntaps, in_stride, out_stride,
                                          cout_blocks, relu, stream);
        default:
            return launch_conv_k<H, W, 1>(items, n_items, cin_chunks, ntaps, in_stride, out_stride,
                                          cout_blocks, relu, stream);
    }
}

}  // namespace

extern "C" int pnmn_conv_nhwc(const pnmn_conv_item* items, int n_items, int H, int W,
                              int cin_chunks, int ntaps, int in_stride, int out_stride,
                              int cout_blocks, int relu, void* stream) {
    if (n_items <= 0) return 0;
    if (!items || cin_chunks < 1 || cout_blocks < 1 || (ntaps != 9 && ntaps != 1)) return PNMN_EINVAL;
    if ((in_stride & 3) || (out_stride & 3)) return PNMN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (H == 14 && W == 14)
        return launch_conv<14, 14>(items, n_items, cin_chunks, ntaps, in_stride, out_stride,
                                   cout_blocks, relu, s);
    return PNMN_ESHAPE;
}
