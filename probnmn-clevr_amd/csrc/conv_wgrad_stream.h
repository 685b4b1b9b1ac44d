// Streamed grouped 3x3 weight gradient for 14x14 maps on gfx950 (launch side: conv_wgrad.hip).
//
//   dW[n][tap][c] += sum over the job's items, over pixels p:
//                    dy[p][n] * (gate[p][n] > 0) * (x * xmask)[shift(p, tap, dil)][c]
//
// (autograd's weight gradient of nn.Conv2d(.., 3, padding=d, dilation=d) + ReLU at
// /root/reference/probnmn/modules/nmn_modules.py:83-86,161-167,241-243 and models/nmn.py:67-78.)
//
// GEMM view per tap: M = output channels, N = input channels, K = pixels, concatenated over the items of a job (the
// items of a job share one weight).  Rounds 1-4 staged an item's whole x and dy tiles (150 KB), then contracted them
// with two waves per SIMD: nothing overlapped the staging and the pairs ran the matrix pipe at ~64 % (PMC).  This is
// the design of the streamed convolution (conv_stream.h) carried over:
//
//   * PERSISTENT, WAVE-SPECIALISED workgroups: 4 contraction waves (one per SIMD) + 4 loader waves.  A workgroup walks
//     UNITS = (job, slab of 64 output x 64 input channels x 9 taps); contraction wave w owns the slab's input channels
//     [16 w, 16 w + 16) x 9 taps x 64 output channels = 36 accumulators of 16x16, resident across the job's items and
//     added into dW with fp32 atomics at the unit's end (jobs that split one weight's items meet there).
//   * x of item i + 1 (64 channels, all 196 pixels: the taps of dilation 8 reach across the whole map) travels by
//     direct-to-LDS loads into the second of two x buffers while item i is contracted -- conv_stream's slot format
//     (two 16-channel sub-slots per 32-channel slot, eight zero rows behind each, pieces permuted), its issue_rows(),
//     its in-place mask fix-up and its row table [tap][pixel] -> LDS row are used as they are.
//   * dy is K-major and needs no halo, so it is streamed in HALVES of an item (25 + 24 k-steps of 4 pixels) through two
//     25 KB slots: the loaders fetch dy and the ReLU gate map into registers, apply the gate and write [pixel][64] rows
//     (one 1 KB wave-store per k-step); the bias gradient falls out of the same registers.
//   * One s_barrier per half item hands a stage over; the loaders then refill what the barrier freed.  Per k-step a
//     contraction wave issues 36 MFMAs (v_mfma_f32_16x16x4_f32, A = dy: one ds_read_b128 serves the four output-channel
//     sub-tiles; B = x: one ds_read_u16 of the row table, one v_xad, one ds_read_b32 per tap) with the next step's
//     operands requested one step ahead and the row entries two steps ahead.
//
// LDS: 4 x-slots (106 496 B) + 2 dy half slots (51 200 B) + row table at conv_stream's offset = Geom::LDS_BYTES.
#pragma once
#include "conv_stream.h"

namespace pnmn {
namespace wstream {

using namespace pnmn::stream;

using G = Geom<14, 14, 14>;
constexpr int KSTEPS = 49;                       // 196 pixels / 4 (the MFMA's k)
constexpr int K_HALF0 = 25;                      // k-steps of the first half of an item (second: 24)
constexpr int XBUF_BYTES = 2 * G::SLOT_BYTES;    // 64 channels
constexpr int DY_OFF = 2 * XBUF_BYTES;
constexpr int DY_SLOT = K_HALF0 * 1024;          // a k-step = 4 pixels x 64 channels x 4 B
constexpr int DY_PIECES = (K_HALF0 + NLOAD - 1) / NLOAD;  // wave-loads of one loader per half (7)
static_assert(DY_OFF + 2 * DY_SLOT <= G::TAB_OFF, "x buffers + dy slots end in front of the row table");

struct Launch {
    int n_jobs, ny, cin64, total;   // ny = 64x64 slabs per weight; total = virtual unit ids
    int x_stride, dy_stride, cin_total;
};

// The (unit, item, half) sequence of this workgroup; every wave walks it with the same arithmetic.  Virtual unit id v:
// XCD v & 7 (workgroups are dealt round-robin over the XCDs) walks the slabs of ITS jobs one behind the other, so the
// workgroups that read one job's maps at about the same time sit behind one L2.
struct Walker {
    const pnmn_wgrad_job* jobs;
    int vid, total, n_jobs, ny;
    int slab, item, item_end, half;
    float *dw, *dbias;

    __device__ __forceinline__ bool valid() const { return vid < total; }
    __device__ __forceinline__ void open() {
        for (; vid < total; vid += (int)gridDim.x) {
            const int idx = vid >> 3;
            const int job = (idx / ny) * 8 + (vid & 7);
            if (job >= n_jobs) continue;
            const pnmn_wgrad_job j = jobs[job];
            if (j.item_end <= j.item_begin) continue;
            slab = idx % ny, item = j.item_begin, item_end = j.item_end, half = 0, dw = j.dw, dbias = j.dbias;
            return;
        }
    }
    __device__ __forceinline__ void start(const Launch& L, const pnmn_wgrad_job* js) {
        jobs = js, vid = (int)blockIdx.x, total = L.total, n_jobs = L.n_jobs, ny = L.ny;
        open();
    }
    // next stage of the same unit; false: the unit is finished
    __device__ __forceinline__ bool next_stage() {
        half ^= 1;
        if (half) return true;
        return ++item < item_end;
    }
    __device__ __forceinline__ void next_unit() {
        vid += (int)gridDim.x;
        open();
    }
    __device__ __forceinline__ bool last_stage() const { return half == 1 && item + 1 == item_end; }
};

// zero rows of the four x slots (no load ever writes them) and the row table of the first unit; ends at the
// workgroup's first barrier
__device__ __forceinline__ void start_up(char* lds, int dil, bool any, int wave, int lane) {
    for (int t = wave * 64 + lane; t < 4 * 64; t += NTHREADS) {
        const int sl = t >> 6, kb = (t >> 5) & 1;
        *reinterpret_cast<f32x4*>(lds + sl * G::SLOT_BYTES + kb * G::SUB_BYTES + G::Z0 * 64 + (t & 31) * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (any) fill_table<14, 14, 14>(lds, 0, dil, 1, 9, lane, wave, NTHREADS / 64);
    lds_barrier();
}

// ------------------------------------------------------------------------------------------------------------------
// loader wave LW
// ------------------------------------------------------------------------------------------------------------------
template <int LW>
__device__ __forceinline__ void loader(const Launch& L, const pnmn_wgrad_item* items, const pnmn_wgrad_job* jobs, char* lds, int lane) {
    lchar* ring = (lchar*)lds;
    __builtin_amdgcn_s_setprio(3);  // (shares its SIMD with a wave that saturates the matrix pipe: conv_stream.h)
    Walker C;
    C.start(L, jobs);

    f32x4 dv[DY_PIECES], gv[DY_PIECES];
    f32x4 bias_acc = f32x4{0.f, 0.f, 0.f, 0.f};

    // x of the stage's item: 64 channels of all rows into x buffer `buf`
    auto issue_x = [&](const Walker& S, int buf) {
        const pnmn_wgrad_item* it = items + S.item;
        const int cib = S.slab % L.cin64;
        const float* src = (it->x2 != nullptr && cib >= 2) ? it->x2 + (cib & 1) * 64 : it->x + cib * 64;
        issue_rows<14, 14, 14, LW>(src, L.x_stride, 0, 14, ring + buf * XBUF_BYTES, lane);
        issue_rows<14, 14, 14, LW>(src + QC, L.x_stride, 0, 14, ring + buf * XBUF_BYTES + G::SLOT_BYTES, lane);
    };
    // dy and gate of the stage (this wave's k-steps: every fourth) into registers
    auto load_dy = [&](const Walker& S) {
        const pnmn_wgrad_item* it = items + S.item;
        const int coh = S.slab / L.cin64;
        const int k0 = S.half ? K_HALF0 : 0, nk = S.half ? KSTEPS - K_HALF0 : K_HALF0;
        const gfloat* d = as_global(it->dy) + coh * 64 + (lane & 15) * 4;
        const gfloat* gt = it->gate ? as_global(it->gate) + coh * 64 + (lane & 15) * 4 : nullptr;
#pragma unroll
        for (int i = 0; i < DY_PIECES; ++i) {
            const int j = LW + NLOAD * i;
            if (j < nk) {
                const size_t o = (size_t)((k0 + j) * 4 + (lane >> 4)) * L.dy_stride;
                dv[i] = load4(d + o);
                gv[i] = gt ? load4(gt + o) : f32x4{1.f, 1.f, 1.f, 1.f};
            }
        }
    };
    // ... gated, into the stage's dy slot; the bias gradient of the slab's output channels on the way
    auto write_dy = [&](const Walker& S) {
        const int nk = S.half ? KSTEPS - K_HALF0 : K_HALF0;
        const bool want_bias = S.dbias != nullptr && (S.slab % L.cin64) == 0;
        char* slot = lds + DY_OFF + S.half * DY_SLOT + lane * 16;
#pragma unroll
        for (int i = 0; i < DY_PIECES; ++i) {
            const int j = LW + NLOAD * i;
            if (j < nk) {
                f32x4 w = dv[i];
                w.x = gv[i].x > 0.f ? w.x : 0.f;
                w.y = gv[i].y > 0.f ? w.y : 0.f;
                w.z = gv[i].z > 0.f ? w.z : 0.f;
                w.w = gv[i].w > 0.f ? w.w : 0.f;
                *reinterpret_cast<f32x4*>(slot + j * 1024) = w;
                if (want_bias) bias_acc += w;
            }
        }
        if (S.last_stage()) {
            if (want_bias) {
                f32x4 v = bias_acc;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    v[c] += __shfl_xor(v[c], 16);
                    v[c] += __shfl_xor(v[c], 32);
                }
                if (lane < 16) {
                    float* dst = S.dbias + (S.slab / L.cin64) * 64 + lane * 4;
                    unsafeAtomicAdd(dst + 0, v.x);
                    unsafeAtomicAdd(dst + 1, v.y);
                    unsafeAtomicAdd(dst + 2, v.z);
                    unsafeAtomicAdd(dst + 3, v.w);
                }
            }
            bias_acc = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    // the stage is ready for its hand-over: dy written; on an item's first half its x has landed and is masked
    auto prepare = [&](const Walker& S, int buf) {
        write_dy(S);
        if (S.half == 0) {
            wait_vm<0>();
            const pnmn_wgrad_item* it = items + S.item;
            if (it->xmask != nullptr) {
                fixup<14, 14, 14, LW>(lds, buf * XBUF_BYTES, 0, it->xmask, 0, 14, false, lane);
                fixup<14, 14, 14, LW>(lds, buf * XBUF_BYTES + G::SLOT_BYTES, 0, it->xmask, 0, 14, false, lane);
            }
        }
    };

    int cnt = 0;  // items handed over so far: item k of the walk lives in x buffer k & 1
    int tab_dil = -1;
    if (C.valid()) {
        load_dy(C);
        issue_x(C, 0);
        tab_dil = items[C.item].dilation;
    }
    start_up(lds, tab_dil, C.valid(), LOADER_WAVE + LW, lane);
    if (C.valid()) prepare(C, 0);
    bool unit_end = false;
    while (C.valid()) {
        if (unit_end) {  // the previous unit's contraction is over: its table may go
            lds_barrier();
            unit_end = false;
            const int dil = items[C.item].dilation;
            if (dil != tab_dil) {
                fill_table<14, 14, 14>(lds, 0, dil, 1, 9, lane, LW, NLOAD);
                tab_dil = dil;
            }
        }
        lds_barrier();  // hand-over: the stage is the contraction waves'; every stage before it is finished
        const int half = C.half;
        Walker N = C;   // the next stage
        if (!N.next_stage()) {
            N.next_unit();
            unit_end = true;
        }
        if (N.valid()) load_dy(N);
        if (half == 0) {
            // the x buffer of the item before this one is free: the item behind this one goes there
            Walker X = N;  // (N is this item's second half)
            if (!X.next_stage()) X.next_unit();
            if (X.valid()) issue_x(X, (cnt + 1) & 1);
        } else {
            ++cnt;
        }
        if (N.valid()) prepare(N, cnt & 1);
        C = N;
    }
    if (unit_end) lds_barrier();
}

// ------------------------------------------------------------------------------------------------------------------
// contraction wave
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void contraction(const Launch& L, const pnmn_wgrad_item* items, const pnmn_wgrad_job* jobs, char* lds,
                                            int wave, int lane) {
    using lf32x4 = __attribute__((address_space(3))) f32x4;
    using lfloat = __attribute__((address_space(3))) float;
    const int li = lane & 15, g = lane >> 4;
    Walker Wk;
    Wk.start(L, jobs);
    start_up(lds, Wk.valid() ? items[Wk.item].dilation : 1, Wk.valid(), wave, lane);

    const uint32_t lds0 = (uint32_t)(uintptr_t)(lchar*)lds;
    // this lane's channel li of a 16-channel sub-slot row: piece (li >> 2) at its permuted position (conv_stream.h),
    // the row-dependent bit of which is in the table entry
    const uint32_t lconst = (uint32_t)(((li & 3) << 2) | (((li >> 2) & 1) << 5) | (((li >> 3) & 1) << 4));
    const uint32_t xsub = lds0 + (uint32_t)((wave >> 1) * G::SLOT_BYTES + (wave & 1) * G::SUB_BYTES);
    const uint32_t dylane = lds0 + (uint32_t)(DY_OFF + g * 256 + li * 16);
    const char* tab_lane = lds + G::TAB_OFF + g * 2;  // entry of (tap, k-step k): + tap * 416 + k * 8

    f32x4 acc[9][4];
    int cnt = 0;
    while (Wk.valid()) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int slab = Wk.slab;
        float* const dw = Wk.dw;
        bool more = true;
        while (more) {
            const int half = Wk.half;
            const uint32_t xbase = xsub + (uint32_t)((cnt & 1) * XBUF_BYTES);
            const int nk = half ? KSTEPS - K_HALF0 : K_HALF0;
            uint32_t dyp = dylane + (uint32_t)(half * DY_SLOT);
            const char* tp = tab_lane + (half ? K_HALF0 * 8 : 0);
            lds_barrier();  // the stage is resident (and the table, on a unit's first stage)

            uint32_t rraw[9], radd[9];
            f32x4 a[2];
            float b[2][9];
            // row address = (table entry ^ lane's channel bits) + x buffer.  As an opaque instruction: left to the compiler,
            // the second step's nine addresses are carried round the loop as 16-bit entries and formed (v_and + v_xad)
            // in front of the next iteration's first MFMA.
            auto xad = [&](uint32_t& dst, uint32_t entry) {
                asm volatile("v_xad_u32 %0, %1, %2, %3" : "=v"(dst) : "v"(entry), "v"(lconst), "s"(xbase));
            };
#pragma unroll
            for (int t = 0; t < 9; ++t) rraw[t] = *reinterpret_cast<const uint16_t*>(tp + t * (TAB_ROWS * 2));
#pragma unroll
            for (int t = 0; t < 9; ++t) radd[t] = (rraw[t] ^ lconst) + xbase;
            a[0] = *reinterpret_cast<const lf32x4*>((uintptr_t)dyp);
#pragma unroll
            for (int t = 0; t < 9; ++t) b[0][t] = *reinterpret_cast<const lfloat*>((uintptr_t)radd[t]);
#pragma unroll
            for (int t = 0; t < 9; ++t) rraw[t] = *reinterpret_cast<const uint16_t*>(tp + t * (TAB_ROWS * 2) + 8);
#pragma unroll
            for (int t = 0; t < 9; ++t) radd[t] = (rraw[t] ^ lconst) + xbase;
            tp += 16;  // (the rows two steps ahead)

            // One k-step: 36 MFMAs on the operands of buffer P, one filler behind each of the first 28 -- the B values
            // and the A fragment of the NEXT step into buffer P ^ 1 (through the row addresses formed a step ago), the
            // table entries of the step after it, and their addresses.  Reads past the half's end (its last two steps
            // run ahead) stay inside the dy slots / the table's 208 rows and are never used.
            auto step = [&](auto PAR) {
                constexpr int P = decltype(PAR)::value;
                dyp += 1024;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        mfma(acc[t][i], a[P][i], b[P][t]);
                        const int n = t * 4 + i;
                        // (every LDS request in the first half of the step, every use of a table entry in the last
                        // quarter: wherever the compiler decides to wait for ALL outstanding LDS reads instead of the one
                        // it needs -- it does, behind the last entry fetch -- they are nine MFMAs old)
                        if (n < 9) rraw[n] = *reinterpret_cast<const uint16_t*>(tp + n * (TAB_ROWS * 2));
                        else if (n == 9) a[P ^ 1] = *reinterpret_cast<const lf32x4*>((uintptr_t)dyp);
                        else if (n < 19) b[P ^ 1][n - 10] = *reinterpret_cast<const lfloat*>((uintptr_t)radd[n - 10]);
                        else if (n >= 27) xad(radd[n - 27], rraw[n - 27]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                tp += 8;
            };
            int kk = 0;
            for (; kk + 1 < nk; kk += 2) {
                step(std::integral_constant<int, 0>{});
                step(std::integral_constant<int, 1>{});
            }
            if (kk < nk) step(std::integral_constant<int, 0>{});

            more = Wk.next_stage();
            if (half == 1) ++cnt;
        }
        lds_barrier();  // end of the unit's contraction: the loaders may rewrite the row table

        // ---- flush: acc[t][i][r] = dW[cout = 64 coh + 4 (4 g + r) + i][t][cin = 64 cib + 16 wave + li]
        // (the 144 addresses depend on nothing the contraction computes: left visible, the compiler forms them before the
        // stage loop and spills them across it -- 218 registers to scratch per unit.  An opaque lane id keeps them here.)
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const int li_o = lane_o & 15, g_o = lane_o >> 4;
        const int cib = slab % L.cin64, coh = slab / L.cin64;
        float* const base = dw + (size_t)(coh * 64 + 16 * g_o) * 9 * L.cin_total + cib * 64 + wave * 16 + li_o;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int cout = 4 * r + i;  // (+ 16 g: in the base)
                    unsafeAtomicAdd(base + (size_t)(cout * 9 + t) * L.cin_total, acc[t][i][r]);
                }
            }
        }
        Wk.next_unit();
    }
}

__device__ __forceinline__ void wgrad_stream(const Launch& L, const pnmn_wgrad_item* items, const pnmn_wgrad_job* jobs, char* lds) {
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    if (wave >= LOADER_WAVE) {
        switch (wave - LOADER_WAVE) {
            case 0: loader<0>(L, items, jobs, lds, lane); break;
            case 1: loader<1>(L, items, jobs, lds, lane); break;
            case 2: loader<2>(L, items, jobs, lds, lane); break;
            default: loader<3>(L, items, jobs, lds, lane); break;
        }
        return;
    }
    contraction(L, items, jobs, lds, wave, lane);
}

}  // namespace wstream
}  // namespace pnmn
