// Grouped NHWC convolution (3x3 with dilation, or 1x1) for gfx950 -- forward and data-gradient: the launch side.
//
// The kernel body is conv_stream.h: persistent, wave-specialised workgroups (4 contraction + 4 loader waves) that walk
// the launch's units -- unit = (item, band of 196 output pixels, block of 128 / split output channels) -- with the
// input streamed through a ring of LDS slots by direct-to-LDS loads.  14x14 maps are one band, 28x28 maps (BASELINE
// config 5) four bands of seven full-width rows.  Weights are never staged: each contraction wave owns 16 or 32 output
// channels and streams its own [16][tap][cin] slice from L2 into registers ahead of use.
//
// Math is exact fp32 on the matrix cores: v_mfma_f32_16x16x4_f32 (weights as the A operand so that each lane ends up
// with 4 consecutive output channels of one pixel -> 16-byte stores).  16x16 tiles because 196 = 12.25 x 16: 13 m-tiles
// waste 5.8 % (32x32 tiles would waste 12.5 %).  K is consumed in a permuted order (lane group g takes channels
// 4g..4g+3 of each 16-channel block, one per MFMA) so that both operands are single 16-byte loads.
//
// Load balance: a launch rarely has exactly k x 256 units (programs differ in length, so the later levels of a step
// have few active examples).  One launch holds up to three SEGMENTS of units, each with its own split (conv_plan.h):
// workgroups take virtual ids in order, so the short workgroups of the larger splits fill the chip behind the last
// whole round of the first split.  A segment's workgroup count is a multiple of 8, so (id & 7) -- the XCD -- means the
// same inside every segment; XCD x takes a CONTIGUOUS range of a segment's units: the host sorts a launch's items by
// weight, so an XCD streams one or two 590 KB weights through its 4 MB L2 instead of all ~15 of the level, and the
// `split` workgroups of a unit -- which all stream the same input -- share that L2.
//
// (Rounds 1-3 ran a second kernel -- one workgroup per unit staging the whole 98 KiB input tile before contracting it,
// with K-split partial sums exchanged through LDS; round 4 moved the 3x3 and then the 1x1 convolutions here and
// deleted it: profiles/ab/round4_conv_stream_ab.txt, round4_conv1x1_ab.txt.)
#include <stdlib.h>

#include "conv_plan.h"
#include "conv_stream.h"
#include "lds_optin.h"

namespace {

using pnmn::LaunchPlan;
using pnmn::plan_launch;

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
    static std::atomic<uint64_t> configured{0};  // (per device: lds_optin.h)
    auto kern = conv_stream_kernel<H, W, TH, TAPS>;
    if (const int e = pnmn::opt_in_lds(reinterpret_cast<const void*>(kern), G::LDS_BYTES, configured)) return e;
    const int n_units = n_items * (H / TH);
    const LaunchPlan lp = plan_launch(n_units, cout_blocks, cin_chunks, ntaps, cus);
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
    if (split != 0 && split != 1 && split != 2 && split != 4 && split != 6 && split != 8 && split != 14 && split != 26) return PNMN_EINVAL;
    pnmn::forced_split() = split;
    return 0;
}

extern "C" int pnmn_conv_nhwc_launches(int n_items, int H, int W, int cin_chunks, int ntaps, int cout_blocks) {
    const int nb = bands_of(H, W);
    if (n_items <= 0 || nb == 0) return 0;
    return 1;  // (a call's segments go out as one launch)
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
    if (H == 14 && W == 14)
        return ntaps == 9 ? launch_stream<14, 14, 14, 9>(items, n_items, cin_chunks, ntaps, in_stride, out_stride, cout_blocks, relu, cus, s)
                          : launch_stream<14, 14, 14, 1>(items, n_items, cin_chunks, ntaps, in_stride, out_stride, cout_blocks, relu, cus, s);
    if (H == 28 && W == 28)
        return ntaps == 9 ? launch_stream<28, 28, 7, 9>(items, n_items, cin_chunks, ntaps, in_stride, out_stride, cout_blocks, relu, cus, s)
                          : launch_stream<28, 28, 7, 1>(items, n_items, cin_chunks, ntaps, in_stride, out_stride, cout_blocks, relu, cus, s);
    return PNMN_ESHAPE;
}
