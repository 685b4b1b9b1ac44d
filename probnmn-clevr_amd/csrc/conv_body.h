// Body of the grouped NHWC convolution (see conv_nhwc.hip for the design notes).
//
// conv_body<H, W, TH, KSPLIT>: the calling workgroup (512 threads) computes output channels
// [cout_block*128 + nsub*128/KSPLIT, ... + 128/KSPLIT) of the TH x W output rows [band*TH, band*TH + TH)
// of one item; `lds` is lds_rows<W, TH, H>() * 128 floats.  TH * W is 196 output pixels (13 m-tiles) in
// both shipped shapes: 14x14 maps are one band (TH = H, the whole map is staged once per 128-channel
// chunk and all nine taps read it), 28x28 maps are four bands of 7 full-width rows.  A band stages only
// the image rows its taps touch, in PASSES:
//     dilation 1 (and 1x1):  one pass per chunk -- rows [y0 - 1, y0 + TH + 1), all taps   (252 pixels)
//     dilation d > 1:        three passes per chunk, one per tap row ky -- rows
//                            [y0 + ky d, y0 + TH + ky d) clipped to the image, taps (ky, -1..1)
// so the LDS image never exceeds (TH + 2) * W pixel rows whatever the dilation (a halo of 2 d rows would
// not fit for d = 4, 8), and full-width rows keep the LDS row index LINEAR in the pixel index, which is
// what the conflict-free slot swizzle below relies on.  Rows outside the image read the zero rows.
// All 512 threads must call it; it ends with the epilogue executed by the waves that own the
// reduced accumulators and contains no barrier after the point where the other waves return.
//
// Optional fused epilogue (MaskBwd != nullptr; data-gradient of a conv whose forward input was
// feats * attn): instead of storing dx, adds dx * attn into dfeats and sum_c dx*feats into dattn
// (fp32 atomics; the same buffers are shared by every consumer of the example's stem output).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/probnmn_hip.h"
#include "global_ptr.h"

namespace pnmn {

constexpr int CB = 128;  // channels per block (input chunk and output block)
constexpr int ZERO_ROWS = 8;  // all-zero pixel rows behind the image, for taps that fall outside it
// first zero row: the image rounded up to a multiple of 8 rows, so that (row & 7) of a zero row is the
// (position & 7) it stands in for
__host__ __device__ constexpr int zero_base(int hw) { return (hw + 7) & ~7; }
// pixel rows of the largest region a band stages (the whole map when the band is the map)
template <int H, int W, int TH>
__host__ __device__ constexpr int region_pixels() { return (TH == H ? H : TH + 2) * W; }
template <int H, int W, int TH>
__host__ __device__ constexpr int lds_rows() { return zero_base(region_pixels<H, W, TH>()) + ZERO_ROWS; }

// LDS image of one 128-channel chunk: pixel row q = 32 slots of 16 bytes (4 channels each).  The slot
// of channels [16 kb + 4 g, +4) of row q is
//     ((kb ^ q) & 7)  |  (g & 1) << 3  |  (g >> 1) << 4
// ds_read_b128 is served in four groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the
// same in the upper half -- over 64 banks of 4 bytes (16 slots): a group holds eight lanes of lane-group
// g (pixels li in {0-3, 12-15} or {4-11}) and eight of g ^ 1 (the other pixels).  Bit 3 of the slot
// separates g from g ^ 1, and within one g the eight pixels differ in q & 7 (a tap shifts all pixels of
// an m-tile by the same amount), so every group touches 16 different slots: no bank conflicts for any
// tap or dilation.  A tap outside the image reads zero row zero_base(HW) + (q & 7) of the position it would have
// had, which keeps that property.  (The first layout XOR-ed the slot with q & 15: conflict-free over
// lanes 0-15, which is not how the hardware groups lanes -- SQ_LDS_BANK_CONFLICT was 36 % of the LDS
// cycles.)
__device__ __forceinline__ int lds_slot(int kb, int g, int q) { return ((kb ^ q) & 7) | ((g & 1) << 3) | ((g >> 1) << 4); }

struct MaskBwd {
    const float* feats;  // [HW][128] forward features (stem output)
    const float* attn;   // [HW] or nullptr for the all-ones attention
    float* dfeats;       // [HW][128], +=
    float* dattn;        // [HW], += (ignored when attn == nullptr)
};

// MSPLIT = 2 (launches of a handful of items only): the band's 13 m-tiles are shared by two workgroups, `msub`
// 0 / 1 taking m-tiles [0, 7) / [7, 14) -- the 14th lies outside the band and is masked.  Both stage the whole
// region (the halo rows are needed anyway); what halves is the matrix time of a launch that is pure latency.
template <int H, int W, int TH, int KSPLIT, int MSPLIT = 1>
__device__ __forceinline__ void conv_body(const pnmn_conv_item& it, int band, int nsub, int cout_block, int cin_chunks,
                                          int ntaps, int in_stride, int out_stride, int relu, float* lds,
                                          const MaskBwd* mb, int msub = 0) {
    constexpr bool WHOLE = (TH == H);   // the band is the whole map: one pass per chunk, compile-time region
    constexpr int HW = TH * W;          // output pixels of this workgroup
    constexpr int ZB = zero_base(region_pixels<H, W, TH>());  // first zero row of the LDS image
    constexpr int MT = ((HW + 15) / 16 + MSPLIT - 1) / MSPLIT;  // m-tiles of this workgroup
    const int mbase = (MSPLIT == 1) ? 0 : msub * MT;             // its first m-tile
    constexpr int NT = 8 / KSPLIT;   // 16-channel output tiles per workgroup
    constexpr int KB = 8 / KSPLIT;   // 16-channel input blocks per wave and tap
    constexpr int NTHREADS = 512;


    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    const int li = lane & 15;
    const int g = lane >> 4;

    // (Measured and rejected: a static s_setprio 1 for waves 4-7 -- MI355X_MICROARCH.md, two waves per SIMD,
    // item 4 -- changes no launch size by more than 1 %.  Also rejected: requesting the NEXT stage's region
    // into registers at the start of a stage's taps, to hide the staging round trips of multi-chunk and
    // three-pass convolutions behind the MFMAs -- 13-16 more live 16-byte registers push the kernel from 128
    // VGPRs to 256 + 60-90 spilled, and every launch size lost 5-10 %: full launches 115 -> 108 TFLOP/s,
    // stem conv1 106 -> 104.)
    const int nt = wave % NT;
    const int ks = wave / NT;  // which slice of the input channels this wave contracts
    const int n0 = cout_block * CB + (nsub * NT + nt) * 16;  // this wave's 16 out channels
    const int cin_total = cin_chunks * CB;
    const int dil = it.dilation;
    const int y0 = WHOLE ? 0 : band * TH;  // first output row of the band
    const int p_img = y0 * W;              // image pixel index of the band's pixel 0

    // pixel handled by this lane in each m-tile (as the MFMA "column" index)
    int py[MT], px[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int p = (mbase + mt) * 16 + li;
        py[mt] = (p < HW) ? y0 + p / W : -100000;
        px[mt] = p % W;
    }

    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // weight row of this lane: output channel n0 + li, channels 4g.. of each 16-block
    const gfloat* wrow = as_global(it.weight) + (size_t)(n0 + li) * ntaps * cin_total + 4 * g + ks * KB * 16;

    if (tid < ZERO_ROWS * 32) reinterpret_cast<f32x4*>(lds + ZB * CB)[tid] = f32x4{0.f, 0.f, 0.f, 0.f};

    // passes per chunk (see the header): uniform over the workgroup
    const int npass = (WHOLE || ntaps == 1 || dil == 1) ? 1 : 3;
    bool first = true;
    for (int chunk = 0; chunk < cin_chunks; ++chunk)
    for (int pass = 0; pass < npass; ++pass) {
        // image rows [rs, re) staged by this pass and the taps [t0, t1) that read them
        int rs = 0, re = H, t0 = 0, t1 = ntaps;
        if (!WHOLE) {
            if (npass == 1) {
                const int halo = (ntaps == 1) ? 0 : 1;
                rs = y0 - halo < 0 ? 0 : y0 - halo;
                re = y0 + TH + halo > H ? H : y0 + TH + halo;
            } else {
                const int a = y0 + (pass - 1) * dil;
                rs = a < 0 ? 0 : (a > H ? H : a);
                re = a + TH < 0 ? 0 : (a + TH > H ? H : a + TH);
                t0 = 3 * pass;
                t1 = t0 + 3;
                if (re <= rs) continue;  // the whole tap row lies outside the image: contributes nothing
            }
        }
        const int NR = WHOLE ? H * W : (re - rs) * W;  // pixel rows of the region
        const int r_img = rs * W;                      // image pixel index of region row 0
        // ---- stage this 128-channel chunk of the region into LDS (fused prologue) ----
        const gfloat* src = as_global((it.in2 != nullptr && chunk > 0) ? it.in2 : it.in + chunk * CB) + (size_t)r_img * in_stride;
        const gfloat* gsrc = it.gate ? as_global(it.gate + chunk * CB) + (size_t)r_img * in_stride : nullptr;
        const gfloat* msrc = it.mask ? as_global(it.mask) + r_img : nullptr;
        if (!first) __syncthreads();  // everyone done reading the previous region
        first = false;
        // This thread's 16-byte pieces of the tile (and of the mask / gate) are requested in two batches
        // of seven before any is used: a rolled loop waits out one full memory round trip per piece --
        // 13 in a row, 12-25 us per chunk with nothing to overlap at one workgroup per CU.  (One batch of
        // 13 would need more registers than the accumulators leave.)
        const int NST = (NR * 32 + NTHREADS - 1) / NTHREADS;
        // (half-map tiles run two workgroups per CU on a 128-register budget: smaller batches)
        constexpr int BATCH = (MT > 8) ? 7 : 4;
#pragma unroll 1
        for (int i0 = 0; i0 < NST; i0 += BATCH) {
            f32x4 sv[BATCH];
#pragma unroll
            for (int i = 0; i < BATCH; ++i) {
                const int idx = tid + (i0 + i) * NTHREADS;
                // eight consecutive lanes take the eight k-blocks of one g: their slots differ in the low
                // three bits, which is what ds_write_b128 (served 8 lanes at a time over 32 banks) needs;
                // the wave as a whole still reads whole 512-byte pixel rows from memory
                const int c4 = ((idx & 7) * 4 + ((idx >> 3) & 3)) * 4;  // first of this thread's four channels
                sv[i] = idx < NR * 32 ? load4(src + (size_t)(idx >> 5) * in_stride + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (msrc) {
                float mk[BATCH];
#pragma unroll
                for (int i = 0; i < BATCH; ++i) {
                    const int idx = tid + (i0 + i) * NTHREADS;
                    mk[i] = idx < NR * 32 ? msrc[idx >> 5] : 0.f;
                }
#pragma unroll
                for (int i = 0; i < BATCH; ++i) sv[i] *= mk[i];
            }
            if (gsrc) {
                f32x4 gt[BATCH];
#pragma unroll
                for (int i = 0; i < BATCH; ++i) {
                    const int idx = tid + (i0 + i) * NTHREADS;
                    const int c4 = ((idx & 7) * 4 + ((idx >> 3) & 3)) * 4;
                    gt[i] = idx < NR * 32 ? load4(gsrc + (size_t)(idx >> 5) * in_stride + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int i = 0; i < BATCH; ++i) {
                    sv[i].x = gt[i].x > 0.f ? sv[i].x : 0.f;
                    sv[i].y = gt[i].y > 0.f ? sv[i].y : 0.f;
                    sv[i].z = gt[i].z > 0.f ? sv[i].z : 0.f;
                    sv[i].w = gt[i].w > 0.f ? sv[i].w : 0.f;
                }
            }
#pragma unroll
            for (int i = 0; i < BATCH; ++i) {
                const int idx = tid + (i0 + i) * NTHREADS;
                const int p = idx >> 5;
                if (idx < NR * 32)
                    *reinterpret_cast<f32x4*>(lds + p * CB + (lds_slot(idx & 7, (idx >> 3) & 3, p) << 2)) = sv[i];
            }
        }
        __syncthreads();

        const gfloat* wchunk = wrow + chunk * CB;

        // LDS float offset of the (tap-shifted) pixel row of every m-tile with this lane's g bits of the
        // slot folded in; the low three bits carry q & 7 for the k-block part of the slot
        auto rowbases = [&](int tap, int (&rb)[MT]) {
            int dy = 0, dx = 0;
            if (ntaps == 9) {
                dy = (tap / 3 - 1) * dil;
                dx = (tap % 3 - 1) * dil;
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int yy = py[mt] + dy;
                const int xx = px[mt] + dx;
                const bool ok = (yy >= rs) && (yy < re) && ((unsigned)xx < (unsigned)W);
                const int qv = (yy - rs) * W + xx;        // region row the tap would have (any sign)
                const int q = ok ? qv : ZB + (qv & 7);
                rb[mt] = q * CB + (((g & 1) << 3 | (g >> 1) << 4) << 2) + (q & 7);
            }
        };
        // Software pipeline, half a step deep, on ONE set of fragment registers: a step (one tap, one
        // 16-channel block) is computed in two halves of m-tiles; as soon as the MFMAs of a half
        // have been issued, the same registers are re-loaded with that half's fragments of the NEXT
        // step, which then have the other half's 24-28 MFMAs (~800 cycles) to arrive -- LDS and L2
        // latency never sit between MFMAs, at no extra register cost.  Inside a half the MFMAs walk
        // its 6-7 accumulators round-robin (no dependent-issue stall).
        constexpr int MH = (MT + 1) / 2;
        int rb[MT];
        rowbases(t0, rb);
        f32x4 afrag[MT];
        f32x4 bfrag[2];
        auto load_half = [&](int lo, int hi, int kbg) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                if (mt >= lo && mt < hi)
                    afrag[mt] = *reinterpret_cast<const f32x4*>(lds + (rb[mt] & ~7) + (((kbg ^ rb[mt]) & 7) << 2));
        };
        auto mfma_half = [&](int lo, int hi, const f32x4 bw) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                if (mt >= lo && mt < hi) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw.x, afrag[mt].x, acc[mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                if (mt >= lo && mt < hi) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw.y, afrag[mt].y, acc[mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                if (mt >= lo && mt < hi) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw.z, afrag[mt].z, acc[mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                if (mt >= lo && mt < hi) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw.w, afrag[mt].w, acc[mt], 0, 0, 0);
        };
        load_half(0, MT, ks * KB);
        bfrag[0] = load4(wchunk + (size_t)t0 * cin_total);

        for (int tap = t0; tap < t1; ++tap) {
            const int tn = (tap + 1 < t1) ? tap + 1 : tap;  // (the last tap re-requests its own data)
            const gfloat* wtap = wchunk + (size_t)tap * cin_total;
            const gfloat* wtn = wchunk + (size_t)tn * cin_total;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const bool last = (kb + 1 == KB);
                const int kbg_next = last ? ks * KB : ks * KB + kb + 1;
                bfrag[1] = load4(last ? wtn : wtap + (kb + 1) * 16);
                const f32x4 bw = bfrag[0];
                mfma_half(0, MH, bw);
                __builtin_amdgcn_sched_barrier(0);
                if (last) rowbases(tn, rb);  // rows of the next tap (every fragment of this tap is already in flight or in registers)
                load_half(0, MH, kbg_next);
                __builtin_amdgcn_sched_barrier(0);
                mfma_half(MH, MT, bw);
                __builtin_amdgcn_sched_barrier(0);
                load_half(MH, MT, kbg_next);
                __builtin_amdgcn_sched_barrier(0);
                bfrag[0] = bfrag[1];
            }
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
        if (ks == 0) {
#pragma unroll 1
            for (int k2 = 1; k2 < KSPLIT; ++k2) {
                const f32x4* src = red + (((k2 - 1) * NT + nt) * MT) * 64 + lane;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] += src[mt * 64];
            }
        }
    }
    if (ks != 0) return;  // (no barrier follows inside this function)

    // ---- epilogue: lane holds out channels n0+4g..+3 of pixel mt*16+li ----
    f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (it.bias) bias4 = load4(as_global(it.bias) + n0 + 4 * g);
    // Every load of the epilogue (previous contents for accumulation, forward features, attention) is
    // requested for a whole group of m-tiles before the first use: issued one m-tile at a time they cost
    // one memory round trip each, in a row, with the matrix cores idle.
    // EG m-tiles at a time: all of them where the register budget allows (one workgroup per CU), four where
    // two workgroups share a CU (the other workgroup's MFMAs cover the extra round trips)
    constexpr int EG = (MT > 8) ? MT : 4;
    if (mb == nullptr && !(it.flags & (PNMN_CONV_ACCUMULATE | PNMN_CONV_ATOMIC))) {
        // plain store, on a path of its own: on gfx9 stores count on vmcnt like loads, and the shared body below -- "add
        // the previous contents, which may or may not have been loaded" -- makes every m-tile's store wait for the store
        // of the tile before it (found in the streamed kernel's epilogue, round 4)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int p = (mbase + mt) * 16 + li;
            if (p < HW) {
                f32x4 v = acc[mt] + bias4;
                if (relu) {
                    v.x = fmaxf(v.x, 0.f);
                    v.y = fmaxf(v.y, 0.f);
                    v.z = fmaxf(v.z, 0.f);
                    v.w = fmaxf(v.w, 0.f);
                }
                store4(as_global(it.out) + (size_t)(p_img + p) * out_stride + n0 + 4 * g, v);
            }
        }
        return;
    }
    if (mb == nullptr || (it.flags & PNMN_CONV_DATTN)) {
        const bool accumulate = (it.flags & PNMN_CONV_ACCUMULATE) && !(it.flags & PNMN_CONV_ATOMIC);
        // PNMN_CONV_DATTN: besides the plain store of dx, d(attention)[p] += sum_c dx[p][c] * feats[p][c] (this wave's
        // 16 channels: four lanes groups of four, then one atomic per pixel and wave)
        const bool dattn = mb != nullptr && mb->attn != nullptr;
#pragma unroll
        for (int m0 = 0; m0 < MT; m0 += EG) {
            f32x4 old[EG];
#pragma unroll
            for (int j = 0; j < EG; ++j) {
                const int mt = m0 + j;
                const int p = (mbase + mt) * 16 + li;
                // (an accumulating output and a d(attention) item never coincide: `old` carries the forward features
                // for the latter)
                old[j] = (mt < MT && accumulate && p < HW)
                             ? load4(as_global(it.out) + (size_t)(p_img + p) * out_stride + n0 + 4 * g)
                             : (mt < MT && dattn && p < HW)
                                   ? load4(as_global(mb->feats) + (size_t)(p_img + p) * CB + n0 + 4 * g)
                                   : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int j = 0; j < EG; ++j) {
                const int mt = m0 + j;
                const int p = (mbase + mt) * 16 + li;
                if (mt < MT && p < HW) {
                    f32x4 v = acc[mt < MT ? mt : 0] + bias4;
                    if (relu) {
                        v.x = fmaxf(v.x, 0.f);
                        v.y = fmaxf(v.y, 0.f);
                        v.z = fmaxf(v.z, 0.f);
                        v.w = fmaxf(v.w, 0.f);
                    }
                    float* dstf = it.out + (size_t)(p_img + p) * out_stride + n0 + 4 * g;
                    if (it.flags & PNMN_CONV_ATOMIC) {
                        unsafeAtomicAdd(dstf + 0, v.x);
                        unsafeAtomicAdd(dstf + 1, v.y);
                        unsafeAtomicAdd(dstf + 2, v.z);
                        unsafeAtomicAdd(dstf + 3, v.w);
                    } else if (dattn) {
                        store4(as_global(dstf), v);
                    } else {
                        store4(as_global(dstf), v + old[j]);
                    }
                }
                if (dattn && mt < MT) {
                    const f32x4 v = acc[mt < MT ? mt : 0];
                    float part = (p < HW) ? v.x * old[j].x + v.y * old[j].y + v.z * old[j].z + v.w * old[j].w : 0.f;
                    part += __shfl_xor(part, 16);  // sum the four channel groups g = 0..3 of this pixel
                    part += __shfl_xor(part, 32);
                    if (p < HW && g == 0) unsafeAtomicAdd(mb->dattn + p_img + p, part);
                }
            }
        }
    } else {
        // fused backward of (feats * attn): this wave owns channels n0..n0+15 of every pixel
        const bool sole = it.flags & PNMN_CONV_MB_SOLE;
        const gfloat* attn = as_global(mb->attn);
#pragma unroll
        for (int m0 = 0; m0 < MT; m0 += EG) {
            float am[EG];
            f32x4 fv[EG], dold[EG];
#pragma unroll
            for (int j = 0; j < EG; ++j) {
                const int mt = m0 + j;
                const int p = (mbase + mt) * 16 + li;
                const bool ok = mt < MT && p < HW;
                am[j] = (ok && attn) ? attn[p_img + p] : 1.f;
                fv[j] = (ok && attn) ? load4(as_global(mb->feats) + (size_t)(p_img + p) * CB + n0 + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
                dold[j] = (ok && sole) ? load4(as_global(mb->dfeats) + (size_t)(p_img + p) * CB + n0 + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int j = 0; j < EG; ++j) {
                const int mt = m0 + j;
                if (mt >= MT) continue;
                const int p = (mbase + mt) * 16 + li;
                const bool ok = p < HW;
                const f32x4 v = acc[mt < MT ? mt : 0];
                float part = v.x * fv[j].x + v.y * fv[j].y + v.z * fv[j].z + v.w * fv[j].w;
                if (ok) {
                    float* d = mb->dfeats + (size_t)(p_img + p) * CB + n0 + 4 * g;
                    if (sole) {  // only this workgroup touches these 4 channels of pixel p
                        store4(as_global(d), dold[j] + v * am[j]);
                    } else {
                        unsafeAtomicAdd(d + 0, v.x * am[j]);
                        unsafeAtomicAdd(d + 1, v.y * am[j]);
                        unsafeAtomicAdd(d + 2, v.z * am[j]);
                        unsafeAtomicAdd(d + 3, v.w * am[j]);
                    }
                }
                if (attn) {
                    part += __shfl_xor(part, 16);  // sum the four channel groups g = 0..3 of this pixel
                    part += __shfl_xor(part, 32);
                    if (ok && g == 0) unsafeAtomicAdd(mb->dattn + p_img + p, part);
                }
            }
        }
    }
}

}  // namespace pnmn
