// Grouped NHWC convolution (3x3 with dilation, or 1x1) for gfx950 -- forward and data-gradient.
//
// One workgroup = one (example, convolution) item x one band of 196 output pixels x one block of 128
// output channels.  14x14 maps are one band: the whole H*W x 128-channel input tile of the example is
// staged ONCE into LDS (98 KiB of the CU's 160 KiB) with the fused prologue (attention mask multiply, or
// ReLU-backward gate), and all nine taps read their A operands straight out of that image at shifted
// pixel rows -- no im2col, no halo copy: out-of-image taps point at one extra all-zero row.  28x28 maps
// (BASELINE config 5) are four bands of seven full-width rows, each staging only the image rows its
// taps touch (conv_body.h: one pass with a one-row halo for dilation 1, one pass per tap row for
// dilations 2 / 4 / 8), so the same 13 m-tiles, the same LDS layout and the same contraction loop
// serve both shapes.  Weights are never staged: each wave owns 16 output channels and streams its own
// [16][tap][cin] slice from L2 into registers one tap ahead of use (nobody else in the workgroup needs
// that slice, so an LDS round trip would be pure overhead).
//
// Math is exact fp32 on the matrix cores: v_mfma_f32_16x16x4_f32 (weights as the A operand so
// that each lane ends up with 4 consecutive output channels of one pixel -> 16-byte stores).
// 16x16 tiles because 196 = 12.25 x 16: 13 m-tiles waste 5.8 % (32x32 tiles would waste
// 12.5 %).  K is consumed in a permuted order (lane group g takes channels 4g..4g+3 of each
// 16-channel block, one per MFMA) so that both operands are single 16-byte loads.
//
// Load balance: a launch rarely has exactly k x 256 units (programs differ in length, so the
// later levels of a step have few active examples).  The K-split variants cut one unit into
// KSPLIT workgroups of 16*8/KSPLIT output channels each; inside a workgroup the 8 waves then
// split the 128 input channels of every tap KSPLIT ways and are summed through LDS at the end.
// Each workgroup still stages the whole input region, so the launcher picks the smallest split
// that fills the chip (see plan_launch).
//
// LDS image: conv_body.h (row = pixel, 32 slots of 16 bytes, swizzled to the lane groups ds_read_b128 /
// ds_write_b128 are served in).
#include <stdlib.h>

#include "conv_body.h"
#include "conv_plan.h"
#include "conv_stream.h"

namespace {

using pnmn::CB;

// A unit = one band of one item (14x14: unit = item; 28x28: four units per item).
//
// One launch holds up to three SEGMENTS of units, each with its own split (plan_launch below): workgroups are
// dispatched in the order of their ids, so the short workgroups of the larger splits fill the chip behind the last
// whole round of the first split without the drain + launch gap a kernel boundary costs (round 3: the segments used to
// be separate launches).  A segment's workgroup count is a multiple of 8, so (blockIdx.x & 7) -- the XCD -- means the
// same inside every segment.
struct Segments {
    int n;
    int wg_begin[3];  // first blockIdx.x of the segment
    int split[3];     // 1, 2, 4, 8 (K-split) or 16 (K-split 8 x two m-halves)
    int unit0[3], n_units[3], per_xcd[3];
};

template <int H, int W, int TH>
__global__ __launch_bounds__(512, 2) void conv_nhwc_kernel(
    const pnmn_conv_item* __restrict__ items, const Segments sg, int cin_chunks, int ntaps, int in_stride,
    int out_stride, int relu) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* lds = reinterpret_cast<float*>(smem_raw);  // [lds_rows][128], the last rows are zero
    constexpr int NB = H / TH;
    int wg_begin = sg.wg_begin[0], split = sg.split[0], unit0 = sg.unit0[0], n_units = sg.n_units[0], per_xcd = sg.per_xcd[0];
    if (sg.n > 1 && (int)blockIdx.x >= sg.wg_begin[1])
        wg_begin = sg.wg_begin[1], split = sg.split[1], unit0 = sg.unit0[1], n_units = sg.n_units[1], per_xcd = sg.per_xcd[1];
    if (sg.n > 2 && (int)blockIdx.x >= sg.wg_begin[2])
        wg_begin = sg.wg_begin[2], split = sg.split[2], unit0 = sg.unit0[2], n_units = sg.n_units[2], per_xcd = sg.per_xcd[2];
    // XCD-aware mapping: the hardware deals workgroups round-robin over the 8 XCDs (XCD = linear id % 8), each
    // with its own L2.  XCD x takes the CONTIGUOUS range [x per_xcd, (x+1) per_xcd) of the segment's units: the
    // host sorts a launch's items by weight, so an XCD streams one or two 590 KB weights through its 4 MB L2
    // instead of all ~15 of the level (dealing units round-robin fetched 380-700 KB per item, ranges 130-160;
    // scripts/pmc_conv.sh).  The `split` workgroups of a unit all stage the same input region and get ids that
    // are congruent mod 8: the region is fetched from HBM once and hit in that XCD's L2 split-1 times.
    const int local = (int)blockIdx.x - wg_begin;
    const int slot = local >> 3;
    const int j = slot / split;
    const int sub = slot % split;
    const int unit = per_xcd ? (local & 7) * per_xcd + j : j * 8 + (local & 7);
    if (unit >= n_units || (per_xcd && j >= per_xcd)) return;
    const int u = unit0 + unit;
    const pnmn_conv_item it = items[u / NB];
    const pnmn::MaskBwd mb{it.mb_feats, it.mb_attn, it.mb_dfeats, it.mb_dattn};
    const pnmn::MaskBwd* mbp = (it.flags & (PNMN_CONV_MASKBWD | PNMN_CONV_DATTN)) ? &mb : nullptr;
    const int band = u % NB, cb = blockIdx.y;
    switch (split) {  // (uniform over the workgroup)
        case 1:
            pnmn::conv_body<H, W, TH, 1>(it, band, 0, cb, cin_chunks, ntaps, in_stride, out_stride, relu, lds, mbp);
            break;
        case 2:
            pnmn::conv_body<H, W, TH, 2>(it, band, sub, cb, cin_chunks, ntaps, in_stride, out_stride, relu, lds, mbp);
            break;
        case 4:
            pnmn::conv_body<H, W, TH, 4>(it, band, sub, cb, cin_chunks, ntaps, in_stride, out_stride, relu, lds, mbp);
            break;
        case 8:
            pnmn::conv_body<H, W, TH, 8>(it, band, sub, cb, cin_chunks, ntaps, in_stride, out_stride, relu, lds, mbp);
            break;
        default:  // 16
            pnmn::conv_body<H, W, TH, 8, 2>(it, band, sub % 8, cb, cin_chunks, ntaps, in_stride, out_stride, relu, lds, mbp,
                                            sub / 8);
            break;
    }
}

using pnmn::LaunchPlan;
using pnmn::plan_launch;

template <int H, int W, int TH>
int launch_segments(const pnmn_conv_item* items, const LaunchPlan& lp, int first, int last, int unit_at, int cin_chunks,
                    int ntaps, int in_stride, int out_stride, int cout_blocks, int relu, hipStream_t stream) {
    constexpr size_t lds_bytes = (size_t)pnmn::lds_rows<H, W, TH>() * CB * sizeof(float);
    static_assert(lds_bytes <= 160 * 1024, "the staged region must fit the CU's LDS");
    static_assert((size_t)7 * ((TH * W + 15) / 16) * 64 * 16 <= lds_bytes, "reduction scratch must fit in the input image");
    static bool configured = false;
    auto kern = conv_nhwc_kernel<H, W, TH>;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    Segments sg{};
    int wgs = 0;
    for (int k = first; k < last; ++k) {
        if (lp.count[k] <= 0) continue;
        const int i = sg.n++;
        sg.wg_begin[i] = wgs;
        sg.split[i] = lp.split[k];
        sg.unit0[i] = unit_at;
        sg.n_units[i] = lp.count[k];
        sg.per_xcd[i] = (lp.count[k] + 7) / 8;
        wgs += ((lp.count[k] + 7) / 8) * 8 * lp.split[k];
        unit_at += lp.count[k];
    }
    if (sg.n == 0) return 0;
    hipLaunchKernelGGL(kern, dim3(wgs, cout_blocks), dim3(512), lds_bytes, stream, items, sg, cin_chunks, ntaps, in_stride,
                       out_stride, relu);
    return (int)hipGetLastError();
}

// the convolutions run on the streamed kernel (conv_stream.h); PNMN_CONV_STREAM=1: the 3x3 ones only, 0: everything on
// the kernel above
inline int stream_level() {
    static const int v = [] {
        const char* e = getenv("PNMN_CONV_STREAM");
        return e ? atoi(e) : 2;
    }();
    return v;
}
inline bool streamed() { return stream_level() != 0; }


template <int H, int W, int TH>
int launch_conv(const pnmn_conv_item* items, int n_items, int cin_chunks, int ntaps, int in_stride,
                int out_stride, int cout_blocks, int relu, int cus, hipStream_t stream) {
    const int n_units = n_items * (H / TH);
    // (the forced split pins the STREAMED kernel's split, whose results do not depend on it; this kernel's K-split
    // changes the summation order, so while the streamed kernel runs the 3x3 convolutions the 1x1 ones keep the plan)
    const LaunchPlan lp = plan_launch(n_units, cout_blocks, cin_chunks, ntaps, cus, false, !streamed());
    return launch_segments<H, W, TH>(items, lp, 0, lp.n_seg, 0, cin_chunks, ntaps, in_stride, out_stride, cout_blocks, relu,
                                     stream);
}

// ---- streamed kernel (conv_stream.h): persistent workgroups of 8 contraction waves + 1 loader wave ----
template <int H, int W, int TH, int TAPS>
__global__ __launch_bounds__(pnmn::stream::NTHREADS, 1) void conv_stream_kernel(const pnmn_conv_item* __restrict__ items,
                                                                                  const pnmn::stream::Launch L) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    pnmn::stream::conv_stream<H, W, TH, TAPS>(L, items, smem_raw);
}


template <int H, int W, int TH, int TAPS>
int launch_stream(const pnmn_conv_item* items, int n_items, int cin_chunks, int ntaps, int in_stride, int out_stride,
                  int cout_blocks, int relu, int cus, hipStream_t stream) {
    using G = pnmn::stream::Geom<H, W, TH>;
    static bool configured = false;
    auto kern = conv_stream_kernel<H, W, TH, TAPS>;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)G::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    const int n_units = n_items * (H / TH);
    const LaunchPlan lp = plan_launch(n_units, cout_blocks, cin_chunks, ntaps, cus, /*streamed=*/true);
    pnmn::stream::Launch L{};
    int wgs = 0, unit_at = 0;
    for (int k = 0; k < lp.n_seg; ++k) {
        if (lp.count[k] <= 0) continue;
        const int i = L.n_seg++;
        L.wg_begin[i] = wgs;
        L.split[i] = lp.split[k];
        L.unit0[i] = unit_at;
        L.n_units[i] = lp.count[k];
        L.per_xcd[i] = (lp.count[k] + 7) / 8;
        wgs += L.per_xcd[i] * 8 * L.split[i];
        unit_at += lp.count[k];
    }
    if (L.n_seg == 0) return 0;
    L.wgs_x = wgs;
    L.total = wgs * cout_blocks;
    L.cin_chunks = cin_chunks, L.ntaps = ntaps, L.in_stride = in_stride, L.out_stride = out_stride, L.relu = relu;
#ifdef PNMN_STREAM_CYCLES  // (cycle accounting build: make cycles, scripts/r04_cycles.py)
    {
        const char* e = getenv("PNMN_CONV_DBGPTR");
        L.dbg = e ? reinterpret_cast<unsigned long long*>(strtoull(e, nullptr, 0)) : nullptr;
    }
#endif
    int grid = (cus >= 8 && cus <= 256) ? (cus & ~7) : pnmn::default_conv_cus();
    if (grid > L.total) grid = L.total;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(pnmn::stream::NTHREADS), G::LDS_BYTES, stream, items, L);
    return (int)hipGetLastError();
}

// bands per item of the shapes the kernels are built for (0: unsupported)
inline int bands_of(int H, int W) { return (H == 14 && W == 14) ? 1 : (H == 28 && W == 28) ? 4 : 0; }

}  // namespace

extern "C" int pnmn_conv_force_split(int split) {
    if (split != 0 && split != 1 && split != 2 && split != 4 && split != 8 && split != 16) return PNMN_EINVAL;
    pnmn::forced_split() = split;
    return 0;
}

extern "C" int pnmn_conv_nhwc_launches(int n_items, int H, int W, int cin_chunks, int ntaps, int cout_blocks) {
    const int nb = bands_of(H, W);
    if (n_items <= 0 || nb == 0) return 0;
    return 1;  // (a call's segments go out as one launch, on either kernel)
}

extern "C" int pnmn_conv_nhwc(const pnmn_conv_item* items, int n_items, int H, int W,
                              int cin_chunks, int ntaps, int in_stride, int out_stride,
                              int cout_blocks, int relu, void* stream) {
    return pnmn_conv_nhwc_cus(items, n_items, H, W, cin_chunks, ntaps, in_stride, out_stride, cout_blocks, relu, 0, stream);
}

extern "C" int pnmn_conv_nhwc_cus(const pnmn_conv_item* items, int n_items, int H, int W,
                                  int cin_chunks, int ntaps, int in_stride, int out_stride,
                                  int cout_blocks, int relu, int cus, void* stream) {
    if (n_items <= 0) return 0;
    if (!items || cin_chunks < 1 || cout_blocks < 1 || (ntaps != 9 && ntaps != 1)) return PNMN_EINVAL;
    if ((in_stride & 3) || (out_stride & 3)) return PNMN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (stream_level() >= 1 && ntaps == 9) {
        if (H == 14 && W == 14)
            return launch_stream<14, 14, 14, 9>(items, n_items, cin_chunks, ntaps, in_stride, out_stride, cout_blocks, relu, cus, s);
        if (H == 28 && W == 28)
            return launch_stream<28, 28, 7, 9>(items, n_items, cin_chunks, ntaps, in_stride, out_stride, cout_blocks, relu, cus, s);
        return PNMN_ESHAPE;
    }
    if (stream_level() >= 2 && ntaps == 1) {
        if (H == 14 && W == 14)
            return launch_stream<14, 14, 14, 1>(items, n_items, cin_chunks, ntaps, in_stride, out_stride, cout_blocks, relu, cus, s);
        if (H == 28 && W == 28)
            return launch_stream<28, 28, 7, 1>(items, n_items, cin_chunks, ntaps, in_stride, out_stride, cout_blocks, relu, cus, s);
        return PNMN_ESHAPE;
    }
    if (H == 14 && W == 14)
        return launch_conv<14, 14, 14>(items, n_items, cin_chunks, ntaps, in_stride, out_stride, cout_blocks, relu, cus, s);
    if (H == 28 && W == 28)
        return launch_conv<28, 28, 7>(items, n_items, cin_chunks, ntaps, in_stride, out_stride, cout_blocks, relu, cus, s);
    return PNMN_ESHAPE;
}
