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

// ------------------------------------------------------------------------------------------------
// 1x1 weight gradient over ONE 128-channel input block (the classifier conv, nmn.py:67-78: 128 -> 1024 channels): no taps,
// no halo -- a plain GEMM  dW[cout][cin] += sum over the job's pixels k of dy[k][cout] (gate[k][cout] > 0) x[k][cin] xmask[k]
// with a 100 000-long reduction (512 items x 196 pixels).  conv_wgrad_kernel<.., 1> staged an item's whole x map and 64
// output channels of dy, then contracted them with 4 MFMAs per two LDS reads: 65 TFLOP/s.  Here a workgroup (4 waves, one
// per SIMD) owns 256 output x 128 input channels of a job -- 32 accumulators of 16x16 per wave (64 output x 128 input
// channels), resident across the job -- and streams BOTH operands in stages of 28 pixels (196 = 7 stages, 784 = 28) through
// two LDS buffers: the next stage's 11 pieces of 16 bytes per thread are requested before the current stage is contracted
// (7 k-steps x 32 MFMAs per wave against 12 four-byte operand reads each) and stored behind it.  Rows are padded by 16
// floats: the four pixel groups of an operand read then fall on two disjoint halves of the banks.
// ------------------------------------------------------------------------------------------------
constexpr int G1_CO = 256;            // output channels of a workgroup
constexpr int G1_PX = 28;             // pixels of a stage
constexpr int G1_DLD = G1_CO + 16;    // floats per staged dy row
constexpr int G1_XLD = CB + 16;       // ... per staged x row
constexpr int G1_STAGE = G1_PX * (G1_DLD + G1_XLD);  // floats per buffer
constexpr size_t G1_LDS = (size_t)G1_STAGE * sizeof(float);  // ONE buffer: two workgroups share a CU and fill each other's gaps

// Work split: the (job, item, stage) sequence of an output block is ONE reduction -- every job of a classifier launch adds
// into the same weight -- so it is cut into equal stage ranges, one per workgroup (a quarter of the grid per output block):
// no round quantisation (65 jobs x 4 blocks on 256 CUs would run 260 units in two rounds) and one atomic flush per
// workgroup; a range that crosses into a job with another weight flushes there.
constexpr int G1_MAX_JOBS = 512;      // job records live in LDS (24 B each) ...
constexpr int G1_MAX_ITEMS = 256;     // ... and the item records of a workgroup's range (48 B each; longer ranges go in pieces)
constexpr size_t G1_JOBS_OFF = G1_LDS, G1_JOFF_OFF = G1_JOBS_OFF + (size_t)G1_MAX_JOBS * sizeof(pnmn_wgrad_job);
constexpr size_t G1_ITEMS_OFF = G1_JOFF_OFF + (size_t)G1_MAX_JOBS * sizeof(int);
constexpr size_t G1_LDS_ALL = G1_ITEMS_OFF + (size_t)G1_MAX_ITEMS * sizeof(pnmn_wgrad_item);
static_assert(2 * G1_LDS_ALL <= 160 * 1024, "two workgroups per CU");

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_wgrad_1x1_gemm_kernel(const pnmn_wgrad_item* __restrict__ items,
                                                                  const pnmn_wgrad_job* __restrict__ jobs, int x_stride,
                                                                  int dy_stride, int HW, int n_jobs, int n_cob) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* const lds = reinterpret_cast<float*>(smem_raw);
    // The job records and the item records of this workgroup's range live in LDS: a stage's loads must not begin with a
    // chase job -> item -> operands through global memory (two dependent round trips in front of every stage's MFMAs, and
    // the in-order memory counter then also holds the operand loads back: the first version ran 54 TFLOP/s that way).
    pnmn_wgrad_job* const jl = reinterpret_cast<pnmn_wgrad_job*>(smem_raw + G1_JOBS_OFF);
    int* const joff = reinterpret_cast<int*>(smem_raw + G1_JOFF_OFF);   // slot of a job's first item of the range in `il`
    pnmn_wgrad_item* const il = reinterpret_cast<pnmn_wgrad_item*>(smem_raw + G1_ITEMS_OFF);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
    constexpr int ND = G1_PX * G1_CO / 4 / 256;                 // 7 pieces of dy per thread and stage
    constexpr int NX = (G1_PX * CB / 4 + 255) / 256;            // 4 (the last one for the first 128 threads)
    const int spi = HW / G1_PX;                                 // stages per item
    const int cob = blockIdx.x % n_cob, w = blockIdx.x / n_cob, wpc = gridDim.x / n_cob;
    for (int j = tid; j < n_jobs; j += 256) jl[j] = jobs[j];
    __syncthreads();
    if (w >= wpc) return;
    // this workgroup's stage range [t0, t1) of the block's reduction, and the (job, item, stage) it starts at
    long total = 0;
    for (int j = 0; j < n_jobs; ++j) total += (long)(jl[j].item_end - jl[j].item_begin) * spi;
    const long r0 = total * w / wpc, r1 = total * (w + 1) / wpc;
    struct Cursor {
        int job, item, st;  // job, absolute item index, stage within the item
    };
    // (a range of more items than the LDS holds records of goes in pieces: one more flush per 511 items)
    for (long t0 = r0; t0 < r1;) {
    const long t1 = r1 - t0 > (long)(G1_MAX_ITEMS - 1) * spi ? t0 + (long)(G1_MAX_ITEMS - 1) * spi : r1;
    __syncthreads();  // (the previous piece's records and stage buffers are no longer read)
    Cursor cur{0, 0, 0};
    {
        long base = 0;
        for (; cur.job < n_jobs; ++cur.job) {
            const long n = (long)(jl[cur.job].item_end - jl[cur.job].item_begin) * spi;
            if (t0 < base + n) break;
            base += n;
        }
        const int local = (int)(t0 - base);
        cur.item = jl[cur.job].item_begin + local / spi, cur.st = local % spi;
    }
    // the items of the range, job by job, into LDS (slot of item i of job j: joff[j] + i)
    {
        const long n_items_mine = (cur.st + (t1 - t0) + spi - 1) / spi;
        long left = n_items_mine;
        int slot = 0;
        for (int j = cur.job; j < n_jobs && left > 0; ++j) {
            const int lo = j == cur.job ? cur.item : jl[j].item_begin, hi = jl[j].item_end;
            const int n = (int)((long)(hi - lo) < left ? hi - lo : left);
            if (tid == 0) joff[j] = slot - lo;
            for (int k = tid; k < n; k += 256) il[slot + k] = items[lo + k];
            slot += n > 0 ? n : 0;
            left -= n > 0 ? n : 0;
        }
        __syncthreads();
    }
    // the item behind item `c.item` of job `c.job`; false: none
    auto item_after = [&](const Cursor& c, Cursor& n) {
        n.job = c.job, n.item = c.item + 1, n.st = 0;
        if (n.item < jl[n.job].item_end) return true;
        for (++n.job; n.job < n_jobs; ++n.job)
            if (jl[n.job].item_end > jl[n.job].item_begin) {
                n.item = jl[n.job].item_begin;
                return true;
            }
        return false;
    };

    f32x4 acc[4][8];
    // bias gradient of output channels 4 (tid & 63) .. + 3 over the pixels this thread stages: `bias_staged` = of the stage in
    // LDS (taken over when that stage is contracted: the stage behind a change of weight is staged before the flush)
    f32x4 bias_acc = f32x4{0.f, 0.f, 0.f, 0.f}, bias_staged = bias_acc;
    auto zero = [&] {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) acc[ct][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        bias_acc = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    // acc[ct][nt][r] = dW[cout = 256 cob + 64 wave + 16 ct + 4 g + r][cin = 16 nt + li]
    auto flush = [&](float* dw, float* dbias) {
        float* const out = dw + (size_t)(cob * G1_CO + wave * 64 + 4 * g) * CB + li;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int nt = 0; nt < 8; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) unsafeAtomicAdd(out + (size_t)(ct * 16 + r) * CB + nt * 16, acc[ct][nt][r]);
        if (dbias != nullptr) {
            float* const b = dbias + cob * G1_CO + 4 * (tid & 63);
            unsafeAtomicAdd(b + 0, bias_acc.x);
            unsafeAtomicAdd(b + 1, bias_acc.y);
            unsafeAtomicAdd(b + 2, bias_acc.z);
            unsafeAtomicAdd(b + 3, bias_acc.w);
        }
    };
    f32x4 dv[ND], xv[NX];
    float mk[NX];
    const pnmn::gfloat* gate_at = nullptr;  // the staged pieces' ReLU gate map (applied where they are stored), or null
    auto fetch = [&](const pnmn_wgrad_item& it, int st) {
        const int p0 = st * G1_PX;
        const pnmn::gfloat* d = pnmn::as_global(it.dy) + (size_t)p0 * dy_stride + cob * G1_CO;
        const pnmn::gfloat* x = pnmn::as_global(it.x) + (size_t)p0 * x_stride;
        gate_at = it.gate ? pnmn::as_global(it.gate) + (size_t)p0 * dy_stride + cob * G1_CO : nullptr;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int idx = tid + 256 * i, px = idx >> 6, c4 = idx & 63;
            dv[i] = pnmn::load4(d + (size_t)px * dy_stride + 4 * c4);
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int idx = tid + 256 * i, px = idx >> 5, c4 = idx & 31;
            const int pxc = idx < G1_PX * CB / 4 ? px : 0;  // (the fourth piece exists for half the threads: the others re-read row 0)
            xv[i] = pnmn::load4(x + (size_t)pxc * x_stride + 4 * c4);
            mk[i] = 1.f;
        }
        if (it.xmask != nullptr) {
            const pnmn::gfloat* xm = pnmn::as_global(it.xmask) + p0;
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const int idx = tid + 256 * i, px = idx >> 5;
                mk[i] = xm[idx < G1_PX * CB / 4 ? px : 0];
            }
        }
    };
    auto stash = [&] {
        float* dl = lds;
        float* xl = dl + G1_PX * G1_DLD;
        if (gate_at != nullptr) {  // (a gated weight gradient is not the classifier's: its map is fetched here, not a stage ahead)
#pragma unroll
            for (int i = 0; i < ND; ++i) {
                const int idx = tid + 256 * i, px = idx >> 6, c4 = idx & 63;
                const f32x4 gt = pnmn::load4(gate_at + (size_t)px * dy_stride + 4 * c4);
                dv[i].x = gt.x > 0.f ? dv[i].x : 0.f;
                dv[i].y = gt.y > 0.f ? dv[i].y : 0.f;
                dv[i].z = gt.z > 0.f ? dv[i].z : 0.f;
                dv[i].w = gt.w > 0.f ? dv[i].w : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int idx = tid + 256 * i, px = idx >> 6, c4 = idx & 63;
            *reinterpret_cast<f32x4*>(dl + px * G1_DLD + 4 * c4) = dv[i];
            bias_staged += dv[i];  // (the gated values, as they are contracted)
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int idx = tid + 256 * i, px = idx >> 5, c4 = idx & 31;
            if (idx < G1_PX * CB / 4) *reinterpret_cast<f32x4*>(xl + px * G1_XLD + 4 * c4) = xv[i] * mk[i];
        }
    };
    fetch(il[joff[cur.job] + cur.item], cur.st);
    stash();
    __syncthreads();
    const int n_mine = (int)(t1 - t0);
    const float* const dl = lds;
    const float* const xl = dl + G1_PX * G1_DLD;
    const float* const ap = dl + g * G1_DLD + wave * 64 + li;   // A[m = li][k = g]: dy[pixel 4 kk + g][output channel]
    const float* const bp = xl + g * G1_XLD + li;               // B[k = g][n = li]: x[pixel 4 kk + g][input channel]
    for (int s = 0; s < n_mine;) {
        // a run of stages that add into one weight: the accumulators stay put until its end (a flush inside the stage
        // loop made the compiler carry them in other registers round the loop and copy all 128 every stage)
        const pnmn_wgrad_job job = jl[cur.job];
        zero();
        for (;;) {
            const bool more = s + 1 < n_mine;
            Cursor nxt{cur.job, cur.item, cur.st + 1};
            if (nxt.st == spi && !item_after(cur, nxt)) nxt = cur;  // (only behind the last stage of all: never fetched)
            const bool go_on = more && jl[nxt.job].dw == job.dw && jl[nxt.job].dbias == job.dbias;
            if (more) fetch(il[joff[nxt.job] + nxt.item], nxt.st);
            bias_acc += bias_staged;
            bias_staged = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < G1_PX / 4; ++kk) {
                float a[4], b[8];
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) a[ct] = ap[kk * 4 * G1_DLD + ct * 16];
#pragma unroll
                for (int nt = 0; nt < 8; ++nt) b[nt] = bp[kk * 4 * G1_XLD + nt * 16];
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                    for (int nt = 0; nt < 8; ++nt) acc[ct][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ct], b[nt], acc[ct][nt], 0, 0, 0);
            }
            __syncthreads();  // (everyone has read the buffer)
            if (more) stash();
            __syncthreads();
            cur = nxt, ++s;
            if (!go_on) break;
        }
        flush(job.dw, job.dbias);
    }
    t0 = t1;
    }
}

int launch_wgrad_1x1_gemm(const pnmn_wgrad_item* items, const pnmn_wgrad_job* jobs, int n_jobs, int HW, int cout_blocks,
                          int x_stride, int dy_stride, int cus, hipStream_t stream) {
    static std::atomic<uint64_t> configured{0};  // (per device: lds_optin.h)
    if (const int e = pnmn::opt_in_lds(reinterpret_cast<const void*>(conv_wgrad_1x1_gemm_kernel), G1_LDS_ALL, configured)) return e;
    const int n_cob = cout_blocks / 2;
    // workgroups per output block: two per CU the launch may use, shared out among the blocks -- but a range of at least
    // ~2 stages each (n_jobs x HW / 28 stages is a lower bound that needs no look at the device records: every job holds
    // at least one item)
    const int budget = (cus >= 1 && cus <= 256) ? cus : pnmn::default_conv_cus();
    int wpc = 2 * budget / n_cob;
    const long least = (long)n_jobs * (HW / G1_PX);
    if (wpc > least / 2) wpc = (int)(least / 2);
    if (wpc < 1) wpc = 1;
    hipLaunchKernelGGL(conv_wgrad_1x1_gemm_kernel, dim3((unsigned)(wpc * n_cob)), dim3(256), G1_LDS_ALL, stream, items, jobs, x_stride,
                       dy_stride, HW, n_jobs, n_cob);
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
    // a 1x1 weight gradient over one 128-channel input block and whole 256-channel output blocks (the classifier conv,
    // either map size): the GEMM kernel
    if (ntaps == 1 && cin_blocks == 1 && (cout_blocks & 1) == 0 && (H * W) % G1_PX == 0 && x_stride >= CB && n_jobs <= G1_MAX_JOBS)
        return launch_wgrad_1x1_gemm(items, jobs, n_jobs, H * W, cout_blocks, x_stride, dy_stride, cus, s);
    if (H == 14 && W == 14) {
        if (ntaps == 9) return launch_wgrad_stream(items, jobs, n_jobs, cin_blocks, cout_blocks, x_stride, dy_stride, cus, s);
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
