// Grouped convolution weight gradient for gfx950.
//
//   dW[n][tap][c] += sum over the job's items, over pixels p:
//                    dy[p][n] * (gate[p][n] > 0) * (x * xmask)[shift(p, tap, dil)][c]
//
// GEMM view per tap: M = output channels, N = input channels, K = pixels (196 per item,
// concatenated over every item of the job -- the items of a job share one weight).
//
// One workgroup (8 waves, two per SIMD) owns the slab dW[64 output channels][all taps][128 input
// channels] of one job.  Wave w owns the 16 input channels [16w, 16w+16) for every tap and all 64
// output channels: TAPS x 4 accumulators of 16x16 (144 VGPRs for 3x3), resident across all of the
// job's items; at the end the slab is added into dW with fp32 atomics (jobs that split one weight's
// items over several workgroups meet there).  Per item the workgroup stages the example's full
// input map (x, with the attention mask fused) and its half of dy (with the ReLU gate fused) ONCE
// for all nine taps: 150 KiB of LDS, 300 KB of traffic per item per weight -- the previous
// tap-row tiling read 600 KB.
//
// Operands per k-step (4 pixels; lane (li, g) takes pixel 4*step + g):
//   A = dy[pixel][64 channels]   one ds_read_b128 per lane: channels 4*li..4*li+3, i.e. value i of
//       the load is row li of the interleaved sub-tile i (output channels {4r + i}) -- shared by
//       all taps;
//   B = x[shift(pixel, tap)][16w + li]   one ds_read_b32 per lane and tap (out-of-image taps read
//       an all-zero row).
// Both patterns touch 16 consecutive slots/words of at most two pixel rows per lane group:
// conflict-free without swizzling.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/probnmn_hip.h"
#include "conv_plan.h"
#include "conv_wgrad_stream.h"
#include "global_ptr.h"
#include "lds_optin.h"


namespace {

constexpr int CB = 128;
constexpr int CH = 64;  // output channels per workgroup

template <int H, int W, int TAPS>
__global__ __launch_bounds__(512) void conv_wgrad_kernel(
    const pnmn_wgrad_item* __restrict__ items, const pnmn_wgrad_job* __restrict__ jobs, int cin_blocks,
    int x_stride, int dy_stride, int n_jobs, int ny) {
    constexpr int HW = H * W;
    static_assert(HW % 4 == 0, "pixel count must be a multiple of the MFMA k (4)");

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    // x image: row p = 128 channels with channel bit 4 flipped on odd rows.  The B operand is read with
    // ds_read_b32 (served 32 lanes at a time over 32 banks): lanes of lane-group g and g + 1 read the
    // same 16 channels of two ADJACENT pixels, i.e. the same banks at different addresses unless the
    // rows are skewed -- SQ_LDS_BANK_CONFLICT was 38 % of the LDS cycles without the flip.  Two zero
    // rows (one per parity) stand in for taps outside the image.
    float* xl = reinterpret_cast<float*>(smem_raw);  // [(HW+2)][128], rows HW, HW+1 = zeros
    float* dl = xl + (HW + 2) * CB;                  // [HW][64]
    // [TAPS][HW]: float offset of the x row a tap reads at each pixel, row parity << 4 folded in.  Built
    // once per item (the dilation is per item); the contraction loop then spends one LDS read + two
    // integer ops per tap instead of the dozen it takes to derive the shifted, bounds-checked, skewed
    // address -- that loop was limited by those integer ops, not by the matrix cores.
    int* qtab = reinterpret_cast<int*>(dl + HW * CH);
    static_assert(HW % 2 == 0, "zero rows keep the parity of the pixel they replace");

    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    const int li = lane & 15;
    const int g = lane >> 4;

    const int cin_total = cin_blocks * CB;
    // Units = (job, slab) pairs, job fastest.  The grid normally holds one workgroup per unit; a launch that shares
    // the chip with other streams' kernels (pnmn_conv_wgrad_cus) holds only as many workgroups as it may count on CUs,
    // and each walks its units -- these workgroups own a CU (150 KB of LDS) for hundreds of microseconds, and one per
    // CU across the whole chip keeps the multi-CU recurrent kernels of the other stream waiting for that long.
    for (int unit = blockIdx.x; unit < n_jobs * ny; unit += gridDim.x) {
    const int slab = unit / n_jobs;
    // `slab` enumerates (64-channel output half-block, cin block)
    const int cib = slab % cin_blocks;
    const int coh = slab / cin_blocks;  // output channels [64*coh, 64*coh + 64)

    const pnmn_wgrad_job job = jobs[unit % n_jobs];

    f32x4 acc[TAPS][4];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bias_acc = 0.f;  // partial of dbias for channel (tid & 63), pixels strided by 8

    if (tid < 64) reinterpret_cast<f32x4*>(xl + HW * CB)[tid] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int ii = job.item_begin; ii < job.item_end; ++ii) {
        const pnmn_wgrad_item it = items[ii];
        const int dil = it.dilation;
        // (global, not flat, loads: see global_ptr.h)
        const pnmn::gfloat* xsrc = pnmn::as_global((it.x2 != nullptr && cib > 0) ? it.x2 : it.x + cib * CB);
        const pnmn::gfloat* xmask = pnmn::as_global(it.xmask);
        const pnmn::gfloat* dysrc = pnmn::as_global(it.dy);
        const pnmn::gfloat* gatesrc = pnmn::as_global(it.gate);
        __syncthreads();  // previous item fully consumed
        for (int idx = tid; idx < TAPS * HW; idx += 512) {
            const int t = idx / HW, p = idx - t * HW;
            int q = p;
            if (TAPS == 9) {
                const int yy = p / W + (t / 3 - 1) * dil;
                const int xx = p % W + (t % 3 - 1) * dil;
                const bool ok = ((unsigned)yy < (unsigned)H) && ((unsigned)xx < (unsigned)W);
                const int qv = yy * W + xx;
                q = ok ? qv : HW + (qv & 1);
            }
            qtab[idx] = q * CB + ((q & 1) << 4);
        }
        // Both tiles are fetched in batches of seven (x) / four (dy + gate) 16-byte pieces per thread that are all requested
        // before the first is used (a rolled loop pays one memory round trip per piece: ~18 in a row per
        // item, a third of the item's matrix time).
        constexpr int NX = (HW * 32 + 511) / 512, ND = (HW * 16 + 511) / 512, BATCH = 7, BATCH_DY = 4;
#pragma unroll 1
        for (int i0 = 0; i0 < NX; i0 += BATCH) {
            f32x4 v[BATCH];
            float mk[BATCH];
#pragma unroll
            for (int i = 0; i < BATCH; ++i) {
                const int idx = tid + (i0 + i) * 512;
                const bool in = (i0 + i < NX) && idx < HW * 32;
                v[i] = in ? pnmn::load4(xsrc + (size_t)(idx >> 5) * x_stride + (idx & 31) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
                mk[i] = (in && xmask) ? xmask[idx >> 5] : 1.f;
            }
#pragma unroll
            for (int i = 0; i < BATCH; ++i) {
                const int idx = tid + (i0 + i) * 512;
                const int p = idx >> 5, sl = idx & 31;
                if ((i0 + i < NX) && idx < HW * 32)
                    *reinterpret_cast<f32x4*>(xl + p * CB + ((sl * 4) ^ ((p & 1) << 4))) = v[i] * mk[i];
            }
        }
#pragma unroll 1
        for (int i0 = 0; i0 < ND; i0 += BATCH_DY) {
            f32x4 v[BATCH_DY], gt[BATCH_DY];
#pragma unroll
            for (int i = 0; i < BATCH_DY; ++i) {
                const int idx = tid + (i0 + i) * 512;
                const bool in = (i0 + i < ND) && idx < HW * 16;
                const size_t o = (size_t)(idx >> 4) * dy_stride + coh * CH + (idx & 15) * 4;
                v[i] = in ? pnmn::load4(dysrc + o) : f32x4{0.f, 0.f, 0.f, 0.f};
                gt[i] = (in && gatesrc) ? pnmn::load4(gatesrc + o) : f32x4{1.f, 1.f, 1.f, 1.f};
            }
#pragma unroll
            for (int i = 0; i < BATCH_DY; ++i) {
                const int idx = tid + (i0 + i) * 512;
                if ((i0 + i < ND) && idx < HW * 16) {
                    f32x4 w = v[i];
                    w.x = gt[i].x > 0.f ? w.x : 0.f;
                    w.y = gt[i].y > 0.f ? w.y : 0.f;
                    w.z = gt[i].z > 0.f ? w.z : 0.f;
                    w.w = gt[i].w > 0.f ? w.w : 0.f;
                    *reinterpret_cast<f32x4*>(dl + (idx >> 4) * CH + (idx & 15) * 4) = w;
                }
            }
        }
        __syncthreads();

        if (job.dbias != nullptr && cib == 0) {
            const int c = tid & 63;
            for (int p = (tid >> 6); p < HW; p += 8) bias_acc += dl[p * CH + c];
        }

        const int col_lo = (wave * 16 + li) & ~16, col_bit = (wave * 16 + li) & 16;
        for (int p0 = 0; p0 < HW; p0 += 4) {
            const int p = p0 + g;  // this lane's k (pixel)
            const f32x4 a = *reinterpret_cast<const f32x4*>(dl + p * CH + li * 4);
#pragma unroll
            for (int t = 0; t < TAPS; ++t) {
                const float b = xl[(qtab[t * HW + p] ^ col_bit) + col_lo];
                acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b, acc[t][0], 0, 0, 0);
                acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b, acc[t][1], 0, 0, 0);
                acc[t][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b, acc[t][2], 0, 0, 0);
                acc[t][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b, acc[t][3], 0, 0, 0);
            }
        }
    }

    // ---- write-out: acc[t][i][r] = dW[cout = 64*coh + 4*(4g + r) + i][t][cin = 128*cib + 16w + li]
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cout = coh * CH + 4 * (4 * g + r) + i;
                float* dst = job.dw + ((size_t)cout * TAPS + t) * cin_total + cib * CB + wave * 16 + li;
                unsafeAtomicAdd(dst, acc[t][i][r]);
            }
        }
    }
    if (job.dbias != nullptr && cib == 0) {
        unsafeAtomicAdd(job.dbias + coh * CH + (tid & 63), bias_acc);
    }
    __syncthreads();  // (the next unit rebuilds the tiles)
    }
}

// ------------------------------------------------------------------------------------------------
// 28x28 maps (BASELINE config 5): the pixel dimension (the GEMM's K) is walked in bands of TH = 7
// full-width rows (196 pixels, as one 14x14 map), and the workgroup's slab shrinks to 64 output x 64
// input channels so that a band's x region WITH the rows its taps reach still fits beside the dy band:
//   x region  [(TH + 2) * W + 2][64]   dilation 1: rows [y0 - 1, y0 + TH + 1), read by all nine taps;
//                                      dilation d > 1: one staging per tap row ky, rows
//                                      [y0 + ky d, y0 + TH + ky d) clipped to the image (a 2 d-row halo
//                                      does not fit for d = 4, 8); rows outside the image -> zero rows
//   dy band   [TH * W][64]             odd pixels stored with their 32-channel halves swapped: the A
//                                      operand is one ds_read_b64 per lane (32 lanes at a time over 64
//                                      banks), and lane groups g, g + 1 read ADJACENT pixels
//   qtab      [TAPS][TH * W]
// Wave w owns input channels [16 (w & 3), +16) x output channels [32 (w >> 2), +32) of the slab for
// every tap: TAPS x 2 accumulators (72 VGPRs for 3x3).
// ------------------------------------------------------------------------------------------------
constexpr int CK = 64;  // slab edge (input and output channels) of the banded kernel

template <int H, int W, int TH, int TAPS>
__global__ __launch_bounds__(512) void conv_wgrad_band_kernel(
    const pnmn_wgrad_item* __restrict__ items, const pnmn_wgrad_job* __restrict__ jobs, int cin_blocks,
    int x_stride, int dy_stride) {
    constexpr int HWB = TH * W;            // pixels per band
    constexpr int NB = H / TH;
    constexpr int NRZ = (TH + 2) * W;      // first zero row of the x image (even: keeps parity)
    static_assert(HWB % 4 == 0 && NRZ % 2 == 0 && H % TH == 0, "band shape");

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* xl = reinterpret_cast<float*>(smem_raw);  // [NRZ + 2][64], channel bit 4 flipped on odd rows
    float* dl = xl + (NRZ + 2) * CK;                 // [HWB][64], halves swapped on odd pixels
    int* qtab = reinterpret_cast<int*>(dl + HWB * CK);

    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    const int li = lane & 15;
    const int g = lane >> 4;
    const int ig = wave & 3;   // 16-input-channel group of the slab
    const int oh = wave >> 2;  // 32-output-channel half of the slab

    // blockIdx.y enumerates (64-channel output block, 64-channel input block)
    const int cin64 = cin_blocks * 2;
    const int cib = blockIdx.y % cin64;
    const int cob = blockIdx.y / cin64;
    const int cin_total = cin_blocks * CB;

    const pnmn_wgrad_job job = jobs[blockIdx.x];

    f32x4 acc[TAPS][2];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bias_acc = 0.f;

    if (tid < 32) reinterpret_cast<f32x4*>(xl + NRZ * CK)[tid] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int col = ig * 16 + li;
    const int col_lo = col & ~16, col_bit = col & 16;

    // stage x rows [rs, re) of the item (masked) and the tap table of taps [t0, t0 + nt)
    auto stage_x = [&](const pnmn_wgrad_item& it, int y0, int rs, int re, int t0, int nt) {
        const int dil = it.dilation;
        const int NR = (re - rs) * W;
        const int r_img = rs * W;
        const pnmn::gfloat* xsrc = pnmn::as_global((it.x2 != nullptr && cib >= 2) ? it.x2 + (cib & 1) * CK
                                                                                  : it.x + cib * CK) + (size_t)r_img * x_stride;
        const pnmn::gfloat* xmask = it.xmask ? pnmn::as_global(it.xmask) + r_img : nullptr;
        for (int idx = tid; idx < nt * HWB; idx += 512) {
            const int tr = idx / HWB, p = idx - tr * HWB;
            const int t = t0 + tr;
            int q = p + (y0 - rs) * W;
            if (TAPS == 9) {
                const int yy = y0 + p / W + (t / 3 - 1) * dil;
                const int xx = p % W + (t % 3 - 1) * dil;
                const bool ok = (yy >= rs) && (yy < re) && ((unsigned)xx < (unsigned)W);
                const int qv = (yy - rs) * W + xx;
                q = ok ? qv : NRZ + (qv & 1);
            }
            qtab[t * HWB + p] = q * CK + ((q & 1) << 4);
        }
        constexpr int BATCH = 4;  // (8 in flight per thread spill the accumulators)
        const int NX = (NR * 16 + 511) / 512;
#pragma unroll 1
        for (int i0 = 0; i0 < NX; i0 += BATCH) {
            f32x4 v[BATCH];
            float mk[BATCH];
#pragma unroll
            for (int i = 0; i < BATCH; ++i) {
                const int idx = tid + (i0 + i) * 512;
                const bool in = idx < NR * 16;
                v[i] = in ? pnmn::load4(xsrc + (size_t)(idx >> 4) * x_stride + (idx & 15) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
                mk[i] = (in && xmask) ? xmask[idx >> 4] : 1.f;
            }
#pragma unroll
            for (int i = 0; i < BATCH; ++i) {
                const int idx = tid + (i0 + i) * 512;
                const int p = idx >> 4, sl = idx & 15;
                if (idx < NR * 16)
                    *reinterpret_cast<f32x4*>(xl + p * CK + ((sl * 4) ^ ((p & 1) << 4))) = v[i] * mk[i];
            }
        }
    };

    // contraction of the staged band over taps [T0, T0 + NT)
    auto contract = [&](auto T0c, auto NTc) {
        constexpr int T0 = decltype(T0c)::value, NT = decltype(NTc)::value;
#pragma unroll 1
        for (int p0 = 0; p0 < HWB; p0 += 4) {
            const int p = p0 + g;  // this lane's k (pixel)
            const float2 a = *reinterpret_cast<const float2*>(dl + p * CK + ((32 * oh) ^ ((p & 1) << 5)) + 2 * li);
#pragma unroll
            for (int t = T0; t < T0 + NT; ++t) {
                const float b = xl[(qtab[t * HWB + p] ^ col_bit) + col_lo];
                acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b, acc[t][0], 0, 0, 0);
                acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b, acc[t][1], 0, 0, 0);
            }
        }
    };

    for (int ii = job.item_begin; ii < job.item_end; ++ii) {
        const pnmn_wgrad_item it = items[ii];
        const int dil = it.dilation;
        for (int band = 0; band < NB; ++band) {
            const int y0 = band * TH;
            const pnmn::gfloat* dysrc = pnmn::as_global(it.dy) + (size_t)y0 * W * dy_stride + cob * CK;
            const pnmn::gfloat* gatesrc = it.gate ? pnmn::as_global(it.gate) + (size_t)y0 * W * dy_stride + cob * CK : nullptr;
            __syncthreads();  // previous band fully consumed
            constexpr int ND = (HWB * 16 + 511) / 512, BATCH_DY = 4;
#pragma unroll 1
            for (int i0 = 0; i0 < ND; i0 += BATCH_DY) {
                f32x4 v[BATCH_DY], gt[BATCH_DY];
#pragma unroll
                for (int i = 0; i < BATCH_DY; ++i) {
                    const int idx = tid + (i0 + i) * 512;
                    const bool in = (i0 + i < ND) && idx < HWB * 16;
                    const size_t o = (size_t)(idx >> 4) * dy_stride + (idx & 15) * 4;
                    v[i] = in ? pnmn::load4(dysrc + o) : f32x4{0.f, 0.f, 0.f, 0.f};
                    gt[i] = (in && gatesrc) ? pnmn::load4(gatesrc + o) : f32x4{1.f, 1.f, 1.f, 1.f};
                }
#pragma unroll
                for (int i = 0; i < BATCH_DY; ++i) {
                    const int idx = tid + (i0 + i) * 512;
                    if ((i0 + i < ND) && idx < HWB * 16) {
                        f32x4 w = v[i];
                        w.x = gt[i].x > 0.f ? w.x : 0.f;
                        w.y = gt[i].y > 0.f ? w.y : 0.f;
                        w.z = gt[i].z > 0.f ? w.z : 0.f;
                        w.w = gt[i].w > 0.f ? w.w : 0.f;
                        const int p = idx >> 4;
                        *reinterpret_cast<f32x4*>(dl + p * CK + (((idx & 15) * 4) ^ ((p & 1) << 5))) = w;
                    }
                }
            }
            if (TAPS == 1 || dil == 1) {
                const int halo = TAPS == 1 ? 0 : 1;
                const int rs = y0 - halo < 0 ? 0 : y0 - halo;
                const int re = y0 + TH + halo > H ? H : y0 + TH + halo;
                stage_x(it, y0, rs, re, 0, TAPS);
                __syncthreads();
                if (job.dbias != nullptr && cib == 0) {
                    const int c = tid & 63;
                    for (int p = (tid >> 6); p < HWB; p += 8) bias_acc += dl[p * CK + (c ^ ((p & 1) << 5))];
                }
                contract(std::integral_constant<int, 0>{}, std::integral_constant<int, TAPS>{});
            } else if constexpr (TAPS == 9) {
                bool first = true;
                auto pass = [&](auto kyc) {
                    constexpr int KY = decltype(kyc)::value;
                    const int a = y0 + (KY - 1) * dil;
                    const int rs = a < 0 ? 0 : (a > H ? H : a);
                    const int re = a + TH < 0 ? 0 : (a + TH > H ? H : a + TH);
                    if (re <= rs) return;  // this tap row reads nothing but padding
                    if (!first) __syncthreads();  // previous region fully consumed
                    stage_x(it, y0, rs, re, 3 * KY, 3);
                    __syncthreads();
                    if (first && job.dbias != nullptr && cib == 0) {
                        const int c = tid & 63;
                        for (int p = (tid >> 6); p < HWB; p += 8) bias_acc += dl[p * CK + (c ^ ((p & 1) << 5))];
                    }
                    first = false;
                    contract(std::integral_constant<int, 3 * KY>{}, std::integral_constant<int, 3>{});
                };
                pass(std::integral_constant<int, 0>{});
                pass(std::integral_constant<int, 1>{});
                pass(std::integral_constant<int, 2>{});
            }
        }
    }

    // ---- write-out: acc[t][i][r] = dW[cout = 64 cob + 32 oh + 2 (4g + r) + i][t][cin = 64 cib + 16 ig + li]
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cout = cob * CK + 32 * oh + 2 * (4 * g + r) + i;
                float* dst = job.dw + ((size_t)cout * TAPS + t) * cin_total + cib * CK + ig * 16 + li;
                unsafeAtomicAdd(dst, acc[t][i][r]);
            }
        }
    }
    if (job.dbias != nullptr && cib == 0) {
        unsafeAtomicAdd(job.dbias + cob * CK + (tid & 63), bias_acc);
    }
}

template <int H, int W, int TH, int TAPS>
int launch_wgrad_band(const pnmn_wgrad_item* items, const pnmn_wgrad_job* jobs, int n_jobs, int cin_blocks,
                      int cout_blocks, int x_stride, int dy_stride, hipStream_t stream) {
    constexpr size_t lds_bytes = ((size_t)((TH + 2) * W + 2) * CK + (size_t)TH * W * CK + (size_t)TAPS * TH * W) * sizeof(float);
    static_assert(lds_bytes <= 160 * 1024, "tiles must fit the CU's LDS");
    static std::atomic<uint64_t> configured{0};  // (per device: lds_optin.h)
    auto kern = conv_wgrad_band_kernel<H, W, TH, TAPS>;
    if (const int e = pnmn::opt_in_lds(reinterpret_cast<const void*>(kern), lds_bytes, configured)) return e;
    dim3 grid(n_jobs, cout_blocks * 2 * cin_blocks * 2);
    hipLaunchKernelGGL(kern, grid, dim3(512), lds_bytes, stream, items, jobs, cin_blocks, x_stride, dy_stride);
    return (int)hipGetLastError();
}

// 3x3 on 14x14 maps: the streamed kernel (conv_wgrad_stream.h)
__global__ __launch_bounds__(pnmn::stream::NTHREADS, 1) void conv_wgrad_stream_kernel(const pnmn_wgrad_item* __restrict__ items,
                                                                                     const pnmn_wgrad_job* __restrict__ jobs,
                                                                                     const pnmn::wstream::Launch L) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    pnmn::wstream::wgrad_stream(L, items, jobs, smem_raw);
}

int launch_wgrad_stream(const pnmn_wgrad_item* items, const pnmn_wgrad_job* jobs, int n_jobs, int cin_blocks, int cout_blocks,
                        int x_stride, int dy_stride, int cus, hipStream_t stream) {
    using G = pnmn::wstream::G;
    auto kern = conv_wgrad_stream_kernel;
    static std::atomic<uint64_t> configured{0};  // (per device: lds_optin.h)
    if (const int e = pnmn::opt_in_lds(reinterpret_cast<const void*>(kern), G::LDS_BYTES, configured)) return e;
    pnmn::wstream::Launch L{};
    L.n_jobs = n_jobs, L.cin64 = cin_blocks * 2, L.ny = cout_blocks * 2 * L.cin64;
    L.total = ((n_jobs + 7) / 8) * 8 * L.ny;
    L.x_stride = x_stride, L.dy_stride = dy_stride, L.cin_total = cin_blocks * CB;
    int grid = (cus >= 1 && cus <= 256) ? cus : pnmn::default_conv_cus();
    if (grid > L.total) grid = L.total;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(pnmn::stream::NTHREADS), G::LDS_BYTES, stream, items, jobs, L);
    return (int)hipGetLastError();
}

template <int H, int W, int TAPS>
int launch_wgrad(const pnmn_wgrad_item* items, const pnmn_wgrad_job* jobs, int n_jobs, int cin_blocks,
                 int cout_blocks, int x_stride, int dy_stride, int cus, hipStream_t stream) {
    constexpr int HW = H * W;
    constexpr size_t lds_bytes = ((size_t)(HW + 2) * CB + (size_t)HW * CH + (size_t)TAPS * HW) * sizeof(float);
    static_assert(lds_bytes <= 160 * 1024, "tiles must fit the CU's LDS");
    static std::atomic<uint64_t> configured{0};  // (per device: lds_optin.h)
    auto kern = conv_wgrad_kernel<H, W, TAPS>;
    if (const int e = pnmn::opt_in_lds(reinterpret_cast<const void*>(kern), lds_bytes, configured)) return e;
    const int ny = cout_blocks * 2 * cin_blocks;
    const long units = (long)n_jobs * ny;
    const long wgs = (cus >= 1 && cus < units) ? cus : units;
    hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(512), lds_bytes, stream, items, jobs, cin_blocks, x_stride,
                       dy_stride, n_jobs, ny);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" int pnmn_conv_wgrad(const pnmn_wgrad_item* items, const pnmn_wgrad_job* jobs, int n_jobs,
                               int H, int W, int ntaps, int cin_blocks, int cout_blocks,
                               int x_stride, int dy_stride, void* stream) {
    return pnmn_conv_wgrad_cus(items, jobs, n_jobs, H, W, ntaps, cin_blocks, cout_blocks, x_stride, dy_stride, 0, stream);
}

extern "C" int pnmn_conv_wgrad_cus(const pnmn_wgrad_item* items, const pnmn_wgrad_job* jobs, int n_jobs,
                                   int H, int W, int ntaps, int cin_blocks, int cout_blocks,
                                   int x_stride, int dy_stride, int cus, void* stream) {
    if (n_jobs <= 0) return 0;
    if (!items || !jobs || cin_blocks < 1 || cout_blocks < 1 || (ntaps != 9 && ntaps != 1))
        return PNMN_EINVAL;
    if ((x_stride & 3) || (dy_stride & 3)) return PNMN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (H == 14 && W == 14) {
        if (ntaps == 9) {
            static const bool old_kernel = getenv("PNMN_WGRAD_OLD") != nullptr;  // (A/B hook of round 5; goes with the old kernel)
            if (old_kernel) return launch_wgrad<14, 14, 9>(items, jobs, n_jobs, cin_blocks, cout_blocks, x_stride, dy_stride, cus, s);
            return launch_wgrad_stream(items, jobs, n_jobs, cin_blocks, cout_blocks, x_stride, dy_stride, cus, s);
        }
        return launch_wgrad<14, 14, 1>(items, jobs, n_jobs, cin_blocks, cout_blocks, x_stride,
                                       dy_stride, cus, s);
    }
    if (H == 28 && W == 28) {
        if (ntaps == 9)
            return launch_wgrad_band<28, 28, 7, 9>(items, jobs, n_jobs, cin_blocks, cout_blocks, x_stride,
                                                   dy_stride, s);
        return launch_wgrad_band<28, 28, 7, 1>(items, jobs, n_jobs, cin_blocks, cout_blocks, x_stride,
                                               dy_stride, s);
    }
    return PNMN_ESHAPE;
}
