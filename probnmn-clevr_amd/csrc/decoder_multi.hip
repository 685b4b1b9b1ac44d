// Multi-CU attention-LSTM decoder for gfx950: eight workgroups per 16-row tile.
//
// decoder.hip gives one workgroup 16 rows for all steps; a step then costs ~27 us of fp32 MFMA on ONE
// CU plus ~40 us of attention (two rows per wave, encoder rows fetched from L2 twice), while the CUs
// that hold no tile idle -- 8 of 256 busy at 128 rows, 32 at 512.  Here a tile is shared by EIGHT
// workgroups that step in lockstep (cluster.h), member p owning
//     attention   rows 2p, 2p+1 of the tile: their encoder outputs (S x 256 floats each) stay in LDS
//                 for the whole sequence, four waves per row split the source positions
//     gates/cell  hidden units [32p, 32p+32) of all four gates: its slices of W_c and W_hh (2 x 16
//                 fragments = 128 VGPRs per lane) stay in registers for the whole sequence
// and two hand-offs per step: the context rows (all-gather through the saved `ctx` tensor), then the
// new hidden state (all-gather through the saved `hs` tensor) -- both tensors are outputs anyway, so
// the exchange costs no extra traffic.  In sampling mode every member computes the 16 x V logits tile
// from the gathered h_t and draws all 16 tokens itself (the Philox stream is keyed by the global row
// and step), so tokens need no exchange.
//
// Backward: member p owns the same units for the cell backward and produces, from its 128 gate
// columns, 16 x 256 PARTIAL sums of dctx = dgates W_c and dh = dgates W_hh (K split eight ways);
// hand-off A; the row owners add the eight partials of their two rows, run the attention backward
// against the LDS-resident encoder rows and publish the rows' complete dh_{t-1}; hand-off B.
// The encoder-output gradient is NOT accumulated here: the kernel emits dctx [B,T,H] and dscore
// [B,T,S] and the caller forms denc = w^T dctx + dscore^T h_prev as two batched GEMMs over time
// (the one-workgroup kernel's per-step read-modify-write of denc is the largest part of its step).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/probnmn_hip.h"
#include "cluster.h"
#include "lds_optin.h"
#include "sampling.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

using pnmn::sigm;
using pnmn::wmax;
using pnmn::wsum;

constexpr int H = 256;
constexpr int G4 = 4 * H;
constexpr int ROWS = 16;
constexpr int MEMBERS = 8;
constexpr int RW = ROWS / MEMBERS;   // attention rows per member
constexpr int UW = H / MEMBERS;      // hidden units per member
constexpr int GLD = UW + 4;
constexpr int MAXS = 64;
constexpr int MAXV = 128;

__device__ __forceinline__ float dot4(const f32x4 a, const f32x4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

struct MFwdArgs {
    const float* xe;
    const float* etable;
    const float* enc;
    const float* mask;
    const float* h0;
    const float* w_c;     // fragment-packed [4H/16][H/16][64][4]
    const float* w_hh;
    const float* w_p;     // [V][H] row major
    const float* b_p;
    float* hs;
    float* cs;
    float* act;
    float* ctx;
    float* probs;
    int64_t* tokens;
    const int64_t* in_tokens;  // teacher forcing without xe: step t's input is row in_tokens[row][t] of etable
    long in_stride;
    int* sync;
    int B, T, S, V, tiles;
    int sample;
    int pad, unk, start;
    uint64_t seed, row_offset;
};

constexpr size_t FWD_FIXED_LDS = sizeof(float) * (4 * ROWS * GLD + RW * 4 * H + ROWS * MAXV + RW * MAXS) + sizeof(int) * ROWS;
constexpr size_t FWD_OVERLAP_LDS = sizeof(float) * (ROWS * H + 64);  // (OVERLAP) the staged h_{t-1} tile + a sink for predicated stores

// OVERLAP (the launch picks it when S leaves 16.6 KB of LDS free: S <= 59): the step's matrix work is two products,
// W_hh h_{t-1} and W_c ctx_t, and only the second needs the attention's result.  So h_{t-1} is staged into LDS at the TOP
// of the step and the 64 MFMAs of its product are issued in four groups of 16 inside the attention phase -- beside the
// scores' LDS reads and DPP reductions, the softmax and the context sums, which are latency, not issue slots -- and the
// gates phase behind the context hand-off stages and contracts the context alone: half the matrix time of a step leaves
// its critical path (cycle stamps before: gate MFMAs 42 % of a step, attention 23 %).  Each partial sum keeps its order
// (k-blocks ascending, x y z w), so the result is bit-identical to the other path's.  Without OVERLAP (S > 59): both
// products behind the hand-off, their A operands double buffered in the 16 KB that `cpart` and `logl` leave idle there.
// SAMPLE: the free-running modes (the kernel picks each step's token); a teacher-forced pass compiles without the token
// phase -- its logits tile, Philox draw and the pointers they keep live are what the register allocator spills.
template <bool OVERLAP, bool SAMPLE>
__device__ __forceinline__ void attn_lstm_fwd_multi_body(const MFwdArgs& a, const int tile, const int part) {
    extern __shared__ __attribute__((aligned(16))) char raw[];
    const int T = a.T, S = a.S;
    float* encl = reinterpret_cast<float*>(raw);                                       // [RW][S][H]
    float (*gl)[ROWS][GLD] = reinterpret_cast<float (*)[ROWS][GLD]>(encl + RW * S * H);  // [4][16][36]
    float (*scl)[MAXS] = reinterpret_cast<float (*)[MAXS]>(&gl[4][0][0]);              // [RW][64]
    int* tokl = reinterpret_cast<int*>(&scl[RW][0]);                                   // [16]
    float (*cpart)[4][H] = reinterpret_cast<float (*)[4][H]>(tokl + ROWS);             // [RW][4][H]
    float (*logl)[MAXV] = reinterpret_cast<float (*)[MAXV]>(&cpart[RW][0][0]);         // [16][128]
    float* hst = &logl[ROWS][0];                                                       // (OVERLAP) [4 parts][16 rows][64]
    float* sink = hst + ROWS * H;                                                      // (OVERLAP) [64]

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int row0 = tile * ROWS, myrow0 = row0 + RW * part, u0 = UW * part;
    pnmn::Cluster cl;
    cl.start(a.sync + tile * pnmn::CLUSTER_COUNTER_STRIDE, MEMBERS);

    for (int i = tid; i < RW * S * (H / 4); i += 512) {
        const int rl = i / (S * (H / 4)), rem = i - rl * S * (H / 4);
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (myrow0 + rl < a.B) v = *reinterpret_cast<const f32x4*>(a.enc + (size_t)(myrow0 + rl) * S * H + 4 * rem);
        *reinterpret_cast<f32x4*>(encl + (size_t)rl * S * H + 4 * rem) = v;
    }
    // this wave's gate tile: gate q = wave / 2, units u0 + 16 * (wave % 2) .. + 15
    const int gate = wave >> 1, ub = wave & 1;
    f32x4 wc[H / 16], wh[H / 16];
    {
        const int ntile = gate * (H / 16) + u0 / 16 + ub;
#pragma unroll
        for (int kb = 0; kb < H / 16; ++kb) {
            const size_t fo = ((size_t)(ntile * (H / 16) + kb) * 64 + lane) * 4;
            wc[kb] = *reinterpret_cast<const f32x4*>(a.w_c + fo);
            wh[kb] = *reinterpret_cast<const f32x4*>(a.w_hh + fo);
        }
    }
    if (tid < ROWS) tokl[tid] = a.start;
    // source mask of the row this wave attends for (constant over the steps)
    const float row_mask = lane < S ? a.mask[(size_t)min(myrow0 + (wave >> 2), a.B - 1) * S + lane] : 0.f;
    float creg = 0.f;                       // cell state of (row tid / 32, unit u0 + tid % 32)
    const int arow = min(row0 + li, a.B - 1);  // A-operand row (padding rows read a real row; results dropped)
    // teacher forcing from the table: tokl holds the step's input tokens as in the sampling modes; thread r < 16
    // fetches row r's next one a step ahead
    const int64_t* tf_row = (a.in_tokens && tid < ROWS) ? a.in_tokens + (size_t)min(row0 + tid, a.B - 1) * a.in_stride : nullptr;
    if (tf_row) tokl[tid] = (int)tf_row[0];

    // A-operand tiles (16 rows x 256 columns of h_{t-1} or ctx_t) in LDS: four parts of 64 columns, slot s (16 bytes) of
    // row r at slot s ^ r -- the 16 lanes of an MFMA operand read hit 16 bank groups.  A thread stages two pieces of a
    // row, 128 columns apart.
    constexpr int PARTF = ROWS * 64;  // floats of a part
    const int s_row = tid >> 5, s_q = tid & 31;
    const int s_rowc = min(row0 + s_row, a.B - 1);
    const int s_off = (s_q >> 4) * PARTF + s_row * 64 + 4 * ((s_q & 15) ^ s_row);
    // the four k-blocks of part p of a staged tile as this lane's MFMA fragments
    auto frags = [&](const float* tile_lds, const int p, f32x4 (&f)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) f[j] = *reinterpret_cast<const f32x4*>(tile_lds + p * PARTF + li * 64 + 4 * ((4 * j + g) ^ li));
    };
    // 16 MFMAs: part p of a 256-long product into two partial sums (even / odd k-block), k-blocks ascending, x y z w
    auto mfma16 = [&](const f32x4 (&f)[4], const f32x4 (&w)[H / 16], const int p, f32x4& even, f32x4& odd) {
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
            const int kb = p * 4 + j;
            even = __builtin_amdgcn_mfma_f32_16x16x4f32(f[j].x, w[kb].x, even, 0, 0, 0);
            odd = __builtin_amdgcn_mfma_f32_16x16x4f32(f[j + 1].x, w[kb + 1].x, odd, 0, 0, 0);
            even = __builtin_amdgcn_mfma_f32_16x16x4f32(f[j].y, w[kb].y, even, 0, 0, 0);
            odd = __builtin_amdgcn_mfma_f32_16x16x4f32(f[j + 1].y, w[kb + 1].y, odd, 0, 0, 0);
            even = __builtin_amdgcn_mfma_f32_16x16x4f32(f[j].z, w[kb].z, even, 0, 0, 0);
            odd = __builtin_amdgcn_mfma_f32_16x16x4f32(f[j + 1].z, w[kb + 1].z, odd, 0, 0, 0);
            even = __builtin_amdgcn_mfma_f32_16x16x4f32(f[j].w, w[kb].w, even, 0, 0, 0);
            odd = __builtin_amdgcn_mfma_f32_16x16x4f32(f[j + 1].w, w[kb + 1].w, odd, 0, 0, 0);
        }
    };

    // (OVERLAP) the tile's recurrent vector into LDS: h_0 here, h_t right behind the hand-off that completes it
    auto stage_h = [&](const float* rows, const size_t row_stride) {
        const float* hsrc = rows + (size_t)s_rowc * row_stride + 4 * s_q;
        const f32x4 h_lo = *reinterpret_cast<const f32x4*>(hsrc), h_hi = *reinterpret_cast<const f32x4*>(hsrc + 128);
        *reinterpret_cast<f32x4*>(hst + s_off) = h_lo;
        *reinterpret_cast<f32x4*>(hst + s_off + 2 * PARTF) = h_hi;
    };
    if constexpr (OVERLAP) stage_h(a.h0, H);
    __syncthreads();

    for (int t = 0; t < T; ++t) {
        // An opaque zero, added to the row indices below: the per-lane 64-bit addresses of a dozen tensors are then formed
        // where they are used (a few VALU each) instead of being hoisted out of the step loop, where they cost 2 VGPRs
        // each for the whole sequence -- beside the 128 weight registers that is what the allocator spilled.
        int zt = 0;
        asm volatile("" : "+v"(zt));
        f32x4 acc;                                             // the step's input projection rows of this wave's gate tile
        f32x4 ph0 = f32x4{0.f, 0.f, 0.f, 0.f}, ph1 = ph0;      // W_hh h_{t-1}, even / odd k-blocks
        auto load_inputs = [&] {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rl = 4 * g + r, row = row0 + rl + zt;
                const int n = gate * H + u0 + 16 * ub + li;
                float v = 0.f;
                if (row < a.B) v = a.xe ? a.xe[((size_t)row * T + t) * G4 + n] : a.etable[(size_t)tokl[rl] * G4 + n];
                acc[r] = v;
            }
        };
        // ---------------- attention for my rows: four waves per row split the source positions ----------------
        {
            const int rl = wave >> 2, q = wave & 3;
            const int row = myrow0 + rl + zt, rowc = min(row, a.B - 1);
            f32x4 hv;
            if constexpr (OVERLAP) {
                const int r = RW * part + rl;  // my row within the tile; columns 4 lane .. + 3
                hv = *reinterpret_cast<const f32x4*>(hst + (lane >> 4) * PARTF + r * 64 + 4 * ((lane & 15) ^ r));
            } else {
                const float* hp = t > 0 ? a.hs + ((size_t)rowc * T + (t - 1)) * H : a.h0 + (size_t)rowc * H;
                hv = *reinterpret_cast<const f32x4*>(hp + 4 * lane);
            }
            const float* er = encl + (size_t)rl * S * H + 4 * lane;
            constexpr int NP = 4;  // (more in flight would spill beside the weight registers)
            int k_first = 0;
            if constexpr (OVERLAP) {
                // the first eight positions of this wave as straight-line code (positions past S re-read the last one;
                // their sums go to the sink), 16 MFMAs of W_hh h_{t-1} inside each batch of four
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    f32x4 ev[NP], hf[4];
                    float pk[NP];
#pragma unroll
                    for (int k = 0; k < NP; ++k) {
                        const int s = q + 4 * (NP * b + k);
                        ev[k] = *reinterpret_cast<const f32x4*>(er + (size_t)(s < S ? s : S - 1) * H);
                    }
                    frags(hst, b, hf);
                    mfma16(hf, wh, b, ph0, ph1);
#pragma unroll
                    for (int k = 0; k < NP; ++k) pk[k] = dot4(ev[k], hv);
#pragma unroll
                    for (int k = 0; k < NP; ++k) pk[k] = wsum(pk[k]);
                    // (branch-free: lane 0 stores a sum that exists into its place, every other lane into the sink --
                    // a conditional store would cut the block the MFMAs are scheduled in)
#pragma unroll
                    for (int k = 0; k < NP; ++k) {
                        const int s = q + 4 * (NP * b + k);
                        float* dst = (lane == 0 && s < S) ? &scl[rl][s] : sink + lane;
                        *dst = pk[k];
                    }
                }
                k_first = 2 * NP;
            }
            {   // this wave's (remaining) positions q + 4 k: all partial dot products first, then the six-step wave
                // reductions of all of them interleaved (one at a time they are 6 dependent shuffles each)
                for (int k0 = k_first; q + 4 * k0 < S; k0 += NP) {
                    float part[NP];
#pragma unroll
                    for (int k = 0; k < NP; ++k) {
                        const int s = q + 4 * (k0 + k);
                        part[k] = s < S ? dot4(*reinterpret_cast<const f32x4*>(er + (size_t)s * H), hv) : 0.f;
                    }
#pragma unroll
                    for (int k = 0; k < NP; ++k) part[k] = wsum(part[k]);  // (DPP chains: independent, the scheduler interleaves them)
#pragma unroll
                    for (int k = 0; k < NP; ++k)
                        if (lane == 0 && q + 4 * (k0 + k) < S) scl[rl][q + 4 * (k0 + k)] = part[k];
                }
            }
            __syncthreads();
            auto softmax = [&] {
                const float m = row_mask;
                const float v = (lane < S ? scl[rl][lane] : 0.f) * m;  // allennlp masked_softmax: softmax(vector * mask) ...
                const float mx = wmax(lane < S ? v : -INFINITY);
                const float ex = lane < S ? expf(v - mx) : 0.f;
                const float p = ex / wsum(ex);
                const float qv = p * m;                                 // ... * mask, renormalised with 1e-13
                const float wgt = qv / (wsum(qv) + 1e-13f);
                if (lane < S) {
                    scl[rl][lane] = wgt;
                    if (row < a.B) a.probs[((size_t)row * T + t) * S + lane] = p;
                }
            };
            if constexpr (OVERLAP) {
                f32x4 hf[4];
                frags(hst, 2, hf);
                if (q == 0) {  // (the MFMAs in both branches: beside the softmax's chain of reductions where there is one)
                    mfma16(hf, wh, 2, ph0, ph1);
                    softmax();
                } else {
                    mfma16(hf, wh, 2, ph0, ph1);
                }
            } else {
                if (q == 0) softmax();
            }
            __syncthreads();
            f32x4 c4 = f32x4{0.f, 0.f, 0.f, 0.f};
            int s_first = q;
            if constexpr (OVERLAP) {
                // the first four positions straight-line (weight 0 past S), the last 16 MFMAs among them
                constexpr int NC = 4;
                f32x4 ev[NC], hf[4];
                float wv[NC];
#pragma unroll
                for (int k = 0; k < NC; ++k) {
                    const int s = q + 4 * k, sc = s < S ? s : S - 1;
                    ev[k] = *reinterpret_cast<const f32x4*>(er + (size_t)sc * H);
                    const float w = scl[rl][sc];
                    wv[k] = s < S ? w : 0.f;
                }
                frags(hst, 3, hf);
                mfma16(hf, wh, 3, ph0, ph1);
#pragma unroll
                for (int k = 0; k < NC; ++k) c4 += ev[k] * wv[k];
                s_first = q + 4 * NC;
            }
            for (int s = s_first; s < S; s += 4) c4 += *reinterpret_cast<const f32x4*>(er + (size_t)s * H) * scl[rl][s];
            *reinterpret_cast<f32x4*>(&cpart[rl][q][4 * lane]) = c4;
            __syncthreads();
            const int rl2 = tid >> 8, k = tid & 255;
            const float cv = (cpart[rl2][0][k] + cpart[rl2][1][k]) + (cpart[rl2][2][k] + cpart[rl2][3][k]);
            if (myrow0 + rl2 < a.B) a.ctx[((size_t)(myrow0 + rl2 + zt) * T + t) * H + k] = cv;
        }
        cl.signal();
        cl.wait();

        // ---------------- gates of my units on the matrix cores ----------------
        const int tf_next = tf_row ? (int)tf_row[min(t + 1, T - 1)] : 0;  // (stored into tokl behind the cell phase)
        if constexpr (OVERLAP) {
            // the context tile through LDS (into the 16 KB of `cpart` + `logl`, idle here), then its 64 MFMAs
            float* cst = &cpart[0][0][0];
            const float* csrc = a.ctx + ((size_t)(s_rowc + zt) * T + t) * H + 4 * s_q;
            const f32x4 c_lo = *reinterpret_cast<const f32x4*>(csrc), c_hi = *reinterpret_cast<const f32x4*>(csrc + 128);
            load_inputs();  // (requested with the context: kept live across the attention they cost spills)
            *reinterpret_cast<f32x4*>(cst + s_off) = c_lo;
            *reinterpret_cast<f32x4*>(cst + s_off + 2 * PARTF) = c_hi;
            __syncthreads();
            f32x4 pc0 = f32x4{0.f, 0.f, 0.f, 0.f}, pc1 = pc0;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                f32x4 cf[4];
                frags(cst, p, cf);
                mfma16(cf, wc, p, pc0, pc1);
            }
            acc += (pc0 + pc1) + (ph0 + ph1);
        } else {
            load_inputs();
            // A operands (the tile's 16 x 256 context and hidden vectors) go through LDS in four parts of 64
            // columns, double buffered in the 16 KB that the attention phase's partial contexts and the token phase's
            // logits leave idle here: part p + 1 requested while part p's MFMAs run (holding all four in registers
            // instead spills -- kernels with scratch memory were seen to cost the HOST milliseconds per step once a
            // second stream runs beside them).  This wave owns ONE output tile, so a single accumulator would make its
            // 128 MFMAs one dependent chain: four partial sums -- context / hidden state, even / odd k-block -- keep
            // four chains in flight.
            constexpr int NB = 4, NPARTS = (H / 16) / NB, PART = 2 * ROWS * 64;
            float* stage = &cpart[0][0][0];  // [2 buffers][2 tensors][16 rows][64]
            const int s_tensor = tid >> 8, o_row = (tid >> 4) & 15, o_slot = tid & 15;
            const int o_rowc = min(row0 + o_row, a.B - 1);
            const float* s_src = (s_tensor == 0 ? a.ctx + ((size_t)o_rowc * T + t) * H
                                                : (t > 0 ? a.hs + ((size_t)o_rowc * T + (t - 1)) * H : a.h0 + (size_t)o_rowc * H)) + 4 * o_slot;
            float* s_dst = stage + (s_tensor * ROWS + o_row) * 64 + 4 * (o_slot ^ o_row);
            f32x4 sv = *reinterpret_cast<const f32x4*>(s_src);
            *reinterpret_cast<f32x4*>(s_dst) = sv;
            __syncthreads();
            f32x4 pc0 = f32x4{0.f, 0.f, 0.f, 0.f}, pc1 = pc0;
#pragma unroll
            for (int p = 0; p < NPARTS; ++p) {
                if (p + 1 < NPARTS) sv = *reinterpret_cast<const f32x4*>(s_src + (p + 1) * 64);
                const float* buf = stage + (p & 1) * PART;
                f32x4 ac[NB], ah[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int slot = (4 * j + g) ^ li;
                    ac[j] = *reinterpret_cast<const f32x4*>(buf + li * 64 + 4 * slot);
                    ah[j] = *reinterpret_cast<const f32x4*>(buf + (ROWS + li) * 64 + 4 * slot);
                }
                mfma16(ac, wc, p, pc0, pc1);
                mfma16(ah, wh, p, ph0, ph1);
                if (p + 1 < NPARTS) {
                    *reinterpret_cast<f32x4*>(s_dst + ((p + 1) & 1) * PART) = sv;
                    __syncthreads();
                }
            }
            acc += (pc0 + pc1) + (ph0 + ph1);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) gl[gate][4 * g + r][16 * ub + li] = acc[r];
        __syncthreads();
        // ---------------- cell: thread -> (row, unit) ----------------
        {
            const int rl = tid >> 5, ul = tid & 31;
            const int row = row0 + rl + zt, u = u0 + ul;
            const float ig = sigm(gl[0][rl][ul]), fg = sigm(gl[1][rl][ul]);
            const float gg = tanhf(gl[2][rl][ul]), og = sigm(gl[3][rl][ul]);
            const float c = fg * creg + ig * gg;
            const float h = og * tanhf(c);
            creg = c;
            if (row < a.B) {
                const size_t o = ((size_t)row * T + t) * H + u;
                a.hs[o] = h;
                a.cs[o] = c;
                float* ar = a.act + ((size_t)row * T + t) * G4;
                ar[u] = ig;
                ar[H + u] = fg;
                ar[2 * H + u] = gg;
                ar[3 * H + u] = og;
            }
        }
        if (tf_row) tokl[tid] = tf_next;       // (read again behind the hand-off's barriers)
        if (t + 1 == T && !SAMPLE) break;  // nobody needs h_T
        cl.signal();
        cl.wait();
        if constexpr (OVERLAP) {  // h_t of the whole tile: the next step's attention and W_hh product, the logits below
            stage_h(a.hs + ((size_t)zt * T + t) * H, (size_t)T * H);
            __syncthreads();
        }

        // ---------------- token choice for the next step (every member, all 16 rows) ----------------
        if constexpr (SAMPLE) {
            const int V = a.V;
            if (16 * wave < V) {
                f32x4 lacc = f32x4{0.f, 0.f, 0.f, 0.f}, lodd = lacc;
                const int vn = 16 * wave + li;
                const bool vok = vn < V;
                const float* hp = a.hs + ((size_t)arow * T + t) * H + 4 * g;
                constexpr int NB = 4;
#pragma unroll
                for (int part = 0; part < (H / 16) / NB; ++part) {
                    f32x4 ah[NB], bp[NB];
                    if constexpr (OVERLAP) frags(hst, part, ah);
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        const int kb = part * NB + j;
                        if constexpr (!OVERLAP) ah[j] = *reinterpret_cast<const f32x4*>(hp + kb * 16);
                        bp[j] = vok ? *reinterpret_cast<const f32x4*>(a.w_p + (size_t)vn * H + kb * 16 + 4 * g)
                                    : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                    for (int j = 0; j < NB; j += 2) {  // two independent chains (even / odd k-block)
                        lacc = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[j].x, bp[j].x, lacc, 0, 0, 0);
                        lodd = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[j + 1].x, bp[j + 1].x, lodd, 0, 0, 0);
                        lacc = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[j].y, bp[j].y, lacc, 0, 0, 0);
                        lodd = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[j + 1].y, bp[j + 1].y, lodd, 0, 0, 0);
                        lacc = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[j].z, bp[j].z, lacc, 0, 0, 0);
                        lodd = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[j + 1].z, bp[j + 1].z, lodd, 0, 0, 0);
                        lacc = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[j].w, bp[j].w, lacc, 0, 0, 0);
                        lodd = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[j + 1].w, bp[j + 1].w, lodd, 0, 0, 0);
                    }
                }
                lacc += lodd;
                const float bias = vok ? a.b_p[vn] : 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) logl[4 * g + r][vn < MAXV ? vn : 0] = vok ? lacc[r] + bias : -INFINITY;
            }
            __syncthreads();
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int rl = 2 * wave + rr;
                const int row = row0 + rl;
                if (row >= a.B) continue;
                const int choice = pnmn::choose_row_token(logl[rl], V, a.sample, a.pad, a.unk, a.start, a.seed,
                                                          a.row_offset + (uint64_t)row, (uint32_t)t);
                if (lane == 0) {
                    tokl[rl] = choice;
                    if (part == 0) a.tokens[(size_t)row * T + t] = choice;
                }
            }
            __syncthreads();
        }
    }
    cl.finish();
}

template <bool OVERLAP, bool SAMPLE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_lstm_fwd_multi_kernel(const MFwdArgs a) {
    int tile, part;
    pnmn::cluster_coords<MEMBERS>(tile, part);
    if (tile >= a.tiles) return;
    attn_lstm_fwd_multi_body<OVERLAP, SAMPLE>(a, tile, part);
}

// TWO independent decoder passes in one launch (the reconstructor's teacher-forced decode and the generator's
// supervised decode of a training iteration: different weights, step counts, source lengths): tiles [0, tiles0) run
// pass a0, the tiles behind them pass a1.  These kernels are bound by their per-step hand-off latency, not by the
// chip: at 128 questions per GPU a pass occupies 32-64 of the 256 CUs for 0.3-0.5 ms, and back to back the passes ADD
// their step counts on the iteration's critical chain; side by side the launch takes as long as the longer one.
// (Two launches on two streams would do the same but put two kernels that wait for their own workgroups on the chip
// next to a third -- DESIGN 6; one grid is resident as a whole or not at all.)
template <bool OVERLAP, bool SAMPLE0, bool SAMPLE1>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_lstm_fwd_pair_kernel(const MFwdArgs a0,
                                                                                                           const MFwdArgs a1,
                                                                                                           const int tiles0) {
    int tile, part;
    pnmn::cluster_coords<MEMBERS>(tile, part);
    if (tile < tiles0) {
        if (tile >= a0.tiles) return;
        attn_lstm_fwd_multi_body<OVERLAP, SAMPLE0>(a0, tile, part);
    } else {
        tile -= tiles0;
        if (tile >= a1.tiles) return;
        attn_lstm_fwd_multi_body<OVERLAP, SAMPLE1>(a1, tile, part);
    }
}

struct MBwdArgs {
    const float* dhs;
    const float* act;
    const float* cs;
    const float* hs;
    const float* probs;
    const float* enc;
    const float* mask;
    const float* h0;
    const float* w_c_t;   // fragment-packed [H/16][4H/16][64][4]
    const float* w_hh_t;
    float* dgates;        // [B][T][4H]
    float* dctx;          // [B][T][H]
    float* dscore;        // [B][T][S]
    float* weights;       // [B][T][S]  masked, renormalised attention weights w (what the caller's dctx GEMM needs)
    float* dh0;           // [B][H]
    float* x1;            // [tiles][2][MEMBERS][2][16][H]   partial dctx / dh of every member
    float* x2;            // [tiles][2][16][H]               complete dh_{t-1} rows
    int* sync;
    int B, T, S, tiles;
};

constexpr int DLD = 4 * UW + 4;
constexpr size_t BWD_FIXED_LDS = sizeof(float) * (ROWS * DLD + 2 * RW * H + 2 * RW * MAXS + RW * 4 * H + 64);  // (+ a sink for predicated stores)

__device__ __forceinline__ void attn_lstm_bwd_multi_body(const MBwdArgs& a, const int tile, const int part) {
    extern __shared__ __attribute__((aligned(16))) char raw[];
    const int T = a.T, S = a.S;
    float* encl = reinterpret_cast<float*>(raw);                                        // [RW][S][H]
    float (*dgl)[DLD] = reinterpret_cast<float (*)[DLD]>(encl + RW * S * H);             // [16][132]
    float (*dctxl)[H] = reinterpret_cast<float (*)[H]>(&dgl[ROWS][0]);                   // [RW][H]
    float (*dhl)[H] = dctxl + RW;                                                        // [RW][H] (unused since round 5)
    float (*dwl)[MAXS] = reinterpret_cast<float (*)[MAXS]>(&dhl[RW][0]);                 // [RW][64] d weights -> d scores
    float (*dhpart)[4][H] = reinterpret_cast<float (*)[4][H]>(&dwl[2 * RW][0]);          // [RW][4][H]
    float* sink = &dhpart[RW][0][0];                                                     // [64]

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int row0 = tile * ROWS, myrow0 = row0 + RW * part, u0 = UW * part;
    pnmn::Cluster cl;
    cl.start(a.sync + tile * pnmn::CLUSTER_COUNTER_STRIDE, MEMBERS);
    float* x1t = a.x1 + (size_t)tile * 2 * MEMBERS * 2 * ROWS * H;
    float* x2t = a.x2 + (size_t)tile * 2 * ROWS * H;

    for (int i = tid; i < RW * S * (H / 4); i += 512) {
        const int rl = i / (S * (H / 4)), rem = i - rl * S * (H / 4);
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (myrow0 + rl < a.B) v = *reinterpret_cast<const f32x4*>(a.enc + (size_t)(myrow0 + rl) * S * H + 4 * rem);
        *reinterpret_cast<f32x4*>(encl + (size_t)rl * S * H + 4 * rem) = v;
    }
    // this wave's output tiles (units 32 * wave .. + 31 of all 256) x my 128 gate columns, both matrices
    constexpr int KB = 4 * UW / 16;  // 8
    f32x4 wc[2][KB], wh[2][KB];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int kl = 0; kl < KB; ++kl) {
            const int q = kl / (UW / 16), jj = kl % (UW / 16);
            const int kb = (q * H + u0) / 16 + jj;
            const size_t fo = ((size_t)((2 * wave + nt) * (G4 / 16) + kb) * 64 + lane) * 4;
            wc[nt][kl] = *reinterpret_cast<const f32x4*>(a.w_c_t + fo);
            wh[nt][kl] = *reinterpret_cast<const f32x4*>(a.w_hh_t + fo);
        }
    float dc_rec = 0.f;
    const float row_mask = lane < S ? a.mask[(size_t)min(myrow0 + (wave >> 2), a.B - 1) * S + lane] : 0.f;
    __syncthreads();

    // What a step reads of the forward pass (activated gates, cell states, the incoming d h, the attention
    // probabilities) does not depend on the recurrence: step t - 1's values are requested before the last
    // hand-off of step t and arrive while it completes, instead of costing an HBM round trip at the top of
    // every step.
    struct Saved {
        float ig, fg, gg, og, c, cp, dhs, p;
    };
    const int c_rl = tid >> 5, c_ul = tid & 31;
    const int c_row = row0 + c_rl, c_u = u0 + c_ul;
    const int p_row = min(myrow0 + (wave >> 2), a.B - 1);
    auto load_saved = [&](int t) {
        Saved v{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (c_row < a.B) {
            const size_t o = ((size_t)c_row * T + t) * H + c_u;
            const float* ar = a.act + ((size_t)c_row * T + t) * G4;
            v.ig = ar[c_u], v.fg = ar[H + c_u], v.gg = ar[2 * H + c_u], v.og = ar[3 * H + c_u];
            v.c = a.cs[o];
            v.cp = t > 0 ? a.cs[o - H] : 0.f;
            v.dhs = a.dhs[o];
        }
        if ((wave & 3) == 0 && lane < S) v.p = a.probs[((size_t)p_row * T + t) * S + lane];
        return v;
    };
    Saved sv = load_saved(T - 1);

    // this wave's A fragments of the step's gate gradients (16 rows x my 128 gate columns, K block kl)
    auto dg_frag = [&](const int kl) { return *reinterpret_cast<const f32x4*>(&dgl[li][kl * 16 + 4 * g]); };
    // 16 MFMAs of one product (weights as the A operand: the product comes out transposed, each lane holding FOUR
    // CONSECUTIVE UNITS of one row -- one 16-byte store per tile instead of four 4-byte ones): K blocks kl0, kl0 + 1 of both
    // output tiles; every accumulator sees its K blocks ascending, x y z w
    auto mfma16 = [&](const f32x4 (&w)[2][KB], const int kl0, f32x4 (&acc)[2]) {
#pragma unroll
        for (int kl = kl0; kl < kl0 + 2; ++kl) {
            const f32x4 av = dg_frag(kl);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[nt][kl].x, av.x, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[nt][kl].y, av.y, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[nt][kl].z, av.z, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[nt][kl].w, av.w, acc[nt], 0, 0, 0);
            }
        }
    };

    // The two products of a step's gate gradients are not equally urgent: the partial dctx = dgates W_c is what the
    // attention backward waits for (hand-off A), the partial dh = dgates W_hh only enters dh_{t-1}, which nobody reads
    // before the NEXT step's cell backward.  So only the dctx product runs in front of hand-off A; the 64 MFMAs of the dh
    // product are issued inside the attention backward (its scores, softmax backward and d h sums are latency, not issue
    // slots -- as in the forward kernel), every member publishes its partial with hand-off B, and the next step's cell
    // backward adds the eight partials (same order as the row owners added them before) to the row owner's attention part.
    for (int t = T - 1; t >= 0; --t) {
        const Saved cur = sv;
        int zt = 0;  // (an opaque zero in the row indices: addresses formed where they are used, see the forward kernel)
        asm volatile("" : "+v"(zt));
        // ---------------- cell backward of my units: thread -> (row, unit) ----------------
        {
            const int rl = c_rl, ul = c_ul;
            const int row = c_row + zt, u = c_u;
            float di = 0.f, df = 0.f, dg = 0.f, dout = 0.f, dcp = 0.f;
            if (row < a.B) {
                const float ig = cur.ig, fg = cur.fg, gg = cur.gg, og = cur.og;
                const float c = cur.c;
                const float cp = cur.cp;
                const float tc = tanhf(c);
                float dhp = 0.f;
                if (t < T - 1) {
                    // d h_t from step t + 1: the members' partial products, then the row owner's attention part
                    const float* ph = x1t + (((size_t)((t + 1) & 1) * MEMBERS) * 2 + 1) * ROWS * H + (size_t)(rl + zt) * H + u;
                    float pv[MEMBERS];
#pragma unroll
                    for (int m = 0; m < MEMBERS; ++m) pv[m] = ph[(size_t)(2 * m) * ROWS * H];
                    const float att = x2t[((size_t)((t + 1) & 1) * ROWS + rl) * H + u];
                    float sh = 0.f;
#pragma unroll
                    for (int m = 0; m < MEMBERS; ++m) sh += pv[m];
                    dhp = sh + att;
                }
                const float dh = cur.dhs + dhp;
                const float dc = dc_rec + dh * og * (1.f - tc * tc);
                di = dc * gg * ig * (1.f - ig);
                df = dc * cp * fg * (1.f - fg);
                dg = dc * ig * (1.f - gg * gg);
                dout = dh * tc * og * (1.f - og);
                dcp = dc * fg;
                float* dr = a.dgates + ((size_t)row * T + t) * G4;
                dr[u] = di;
                dr[H + u] = df;
                dr[2 * H + u] = dg;
                dr[3 * H + u] = dout;
            }
            dc_rec = dcp;
            dgl[rl][ul] = di;
            dgl[rl][UW + ul] = df;
            dgl[rl][2 * UW + ul] = dg;
            dgl[rl][3 * UW + ul] = dout;
        }
        __syncthreads();
        // ---------------- partial dctx from my gate columns ----------------
        float* const pc = x1t + (((size_t)(t & 1) * MEMBERS + part) * 2 + 0) * ROWS * H;
        float* const ph = pc + ROWS * H;
        {
            f32x4 accc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int kl = 0; kl < KB; kl += 2) mfma16(wc, kl, accc);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const size_t o = (size_t)(li + zt) * H + 16 * (2 * wave + nt) + 4 * g;  // row li, units 16 (2 wave + nt) + 4 g ..
                *reinterpret_cast<f32x4*>(pc + o) = accc[nt];
            }
        }
        cl.signal();
        cl.wait();

        // ---------------- my rows: gather the dctx partials, attention backward (+ the dh product) ----------------
        f32x4 acch[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        {
            const int rl2 = tid >> 8, k = tid & 255;
            const float* src = x1t + (size_t)(t & 1) * MEMBERS * 2 * ROWS * H + (size_t)(RW * part + rl2 + zt) * H + k;
            float pv[MEMBERS];
#pragma unroll
            for (int s = 0; s < MEMBERS; ++s) pv[s] = src[(size_t)(2 * s) * ROWS * H];
            float sc = 0.f;
#pragma unroll
            for (int s = 0; s < MEMBERS; ++s) sc += pv[s];
            dctxl[rl2][k] = sc;
            if (myrow0 + rl2 < a.B) a.dctx[((size_t)(myrow0 + rl2 + zt) * T + t) * H + k] = sc;
        }
        __syncthreads();
        {
            const int rl = wave >> 2, q = wave & 3;
            const int row = myrow0 + rl + zt;
            const float* er = encl + (size_t)rl * S * H + 4 * lane;
            const f32x4 dc4 = *reinterpret_cast<const f32x4*>(&dctxl[rl][4 * lane]);
            constexpr int NP = 4;  // (more in flight would spill beside the weight registers)
            // the first eight positions of this wave as straight-line code (positions past S re-read the last one; their
            // sums go to the sink), 16 MFMAs of the dh product inside each batch of four
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                f32x4 ev[NP];
                float pk[NP];
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    const int s = q + 4 * (NP * b + k);
                    ev[k] = *reinterpret_cast<const f32x4*>(er + (size_t)(s < S ? s : S - 1) * H);
                }
                mfma16(wh, 2 * b, acch);
#pragma unroll
                for (int k = 0; k < NP; ++k) pk[k] = dot4(ev[k], dc4);
#pragma unroll
                for (int k = 0; k < NP; ++k) pk[k] = wsum(pk[k]);
#pragma unroll
                for (int k = 0; k < NP; ++k) {  // (branch-free: lane 0 stores a sum that exists, every other lane into the sink)
                    const int s = q + 4 * (NP * b + k);
                    float* dst = (lane == 0 && s < S) ? &dwl[rl][s] : sink + lane;
                    *dst = pk[k];
                }
            }
            for (int k0 = 2 * NP; q + 4 * k0 < S; k0 += NP) {  // (the remaining positions, S > 32)
                float part[NP];
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    const int s = q + 4 * (k0 + k);
                    part[k] = s < S ? dot4(*reinterpret_cast<const f32x4*>(er + (size_t)s * H), dc4) : 0.f;
                }
#pragma unroll
                for (int k = 0; k < NP; ++k) part[k] = wsum(part[k]);  // (DPP chains: independent, the scheduler interleaves them)
#pragma unroll
                for (int k = 0; k < NP; ++k)
                    if (lane == 0 && q + 4 * (k0 + k) < S) dwl[rl][q + 4 * (k0 + k)] = part[k];
            }
            __syncthreads();
            auto softmax_bwd = [&] {
                // forward quantities of this (row, step): p (softmax before masking), mask, q, Z
                const float m = row_mask;
                const float p = lane < S ? cur.p : 0.f;
                const float qv = p * m;
                const float Z = wsum(qv) + 1e-13f;
                const float dw = lane < S ? dwl[rl][lane] : 0.f;
                // w = q / Z ; q = p * mask ; p = softmax(score * mask)
                const float dq = dw / Z - wsum(dw * qv) / (Z * Z);
                const float dp = dq * m;
                const float dv = p * (dp - wsum(dp * p));
                const float dscore = dv * m;
                if (lane < S) {
                    dwl[rl][lane] = dscore;
                    if (row < a.B) {
                        a.dscore[((size_t)row * T + t) * S + lane] = dscore;
                        a.weights[((size_t)row * T + t) * S + lane] = qv / Z;
                    }
                }
            };
            if (q == 0) {  // (the MFMAs in both branches: beside the chain of reductions where there is one)
                mfma16(wh, 4, acch);
                softmax_bwd();
            } else {
                mfma16(wh, 4, acch);
            }
            __syncthreads();
            f32x4 dh4 = f32x4{0.f, 0.f, 0.f, 0.f};
            {
                // the first four positions straight-line (weight 0 past S), the last 16 MFMAs among them
                constexpr int NC = 4;
                f32x4 ev[NC];
                float wv[NC];
#pragma unroll
                for (int k = 0; k < NC; ++k) {
                    const int s = q + 4 * k, sc = s < S ? s : S - 1;
                    ev[k] = *reinterpret_cast<const f32x4*>(er + (size_t)sc * H);
                    const float w = dwl[rl][sc];
                    wv[k] = s < S ? w : 0.f;
                }
                mfma16(wh, 6, acch);
#pragma unroll
                for (int k = 0; k < NC; ++k) dh4 += ev[k] * wv[k];
            }
            for (int s = q + 16; s < S; s += 4) dh4 += *reinterpret_cast<const f32x4*>(er + (size_t)s * H) * dwl[rl][s];
            *reinterpret_cast<f32x4*>(&dhpart[rl][q][4 * lane]) = dh4;
        }
        // this member's partial dh: published by hand-off B (the final step's: by the one after the loop)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const size_t o = (size_t)(li + zt) * H + 16 * (2 * wave + nt) + 4 * g;
            *reinterpret_cast<f32x4*>(ph + o) = acch[nt];
        }
        __syncthreads();
        {
            // the attention's part of d h_{t-1} for my rows
            const int rl2 = tid >> 8, k = tid & 255;
            const float v = (dhpart[rl2][0][k] + dhpart[rl2][1][k]) + (dhpart[rl2][2][k] + dhpart[rl2][3][k]);
            x2t[((size_t)(t & 1) * ROWS + RW * part + rl2 + zt) * H + k] = v;
        }
        if (t > 0) sv = load_saved(t - 1);
        cl.signal();
        cl.wait();
    }
    {   // d h_0 of my rows: the members' partials of the last step (t = 0), then the attention part (as the cell backward adds them)
        const int rl2 = tid >> 8, k = tid & 255;
        const float* ph0 = x1t + (size_t)1 * ROWS * H + (size_t)(RW * part + rl2) * H + k;  // buffer 0 (t = 0), tensor 1
        float pv[MEMBERS];
#pragma unroll
        for (int m = 0; m < MEMBERS; ++m) pv[m] = ph0[(size_t)(2 * m) * ROWS * H];
        const float att = x2t[((size_t)0 * ROWS + RW * part + rl2) * H + k];
        float sh = 0.f;
#pragma unroll
        for (int m = 0; m < MEMBERS; ++m) sh += pv[m];
        if (myrow0 + rl2 < a.B) a.dh0[(size_t)(myrow0 + rl2) * H + k] = sh + att;
    }
    cl.finish();
}

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_lstm_bwd_multi_kernel(const MBwdArgs a) {
    int tile, part;
    pnmn::cluster_coords<MEMBERS>(tile, part);
    if (tile >= a.tiles) return;
    attn_lstm_bwd_multi_body(a, tile, part);
}

// the backward passes of two decoders in one launch (see attn_lstm_fwd_pair_kernel)
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_lstm_bwd_pair_kernel(const MBwdArgs a0,
                                                                                                           const MBwdArgs a1,
                                                                                                           const int tiles0) {
    int tile, part;
    pnmn::cluster_coords<MEMBERS>(tile, part);
    if (tile < tiles0) {
        if (tile >= a0.tiles) return;
        attn_lstm_bwd_multi_body(a0, tile, part);
    } else {
        tile -= tiles0;
        if (tile >= a1.tiles) return;
        attn_lstm_bwd_multi_body(a1, tile, part);
    }
}

// ... of THREE (round 5): the generator's two decodes and the reconstructor's.  In the backward pass of a training iteration
// the three are independent (the samples are discrete: nothing flows from the reconstructor into the generator), and on one
// stream they would add their step counts on the iteration's critical chain -- at 128 questions per GPU the seq2seq backward
// is that chain (scripts/step_timeline.py).
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_lstm_bwd_group3_kernel(const MBwdArgs a0,
                                                                                                             const MBwdArgs a1,
                                                                                                             const MBwdArgs a2,
                                                                                                             const int tiles0,
                                                                                                             const int tiles01) {
    int tile, part;
    pnmn::cluster_coords<MEMBERS>(tile, part);
    if (tile < tiles0) {
        if (tile >= a0.tiles) return;
        attn_lstm_bwd_multi_body(a0, tile, part);
    } else if (tile < tiles01) {
        tile -= tiles0;
        if (tile >= a1.tiles) return;
        attn_lstm_bwd_multi_body(a1, tile, part);
    } else {
        tile -= tiles01;
        if (tile >= a2.tiles) return;
        attn_lstm_bwd_multi_body(a2, tile, part);
    }
}

// the compiled variants of the forward kernels
const void* fwd_multi_variant(bool overlap, bool sample) {
    if (overlap) return sample ? reinterpret_cast<const void*>(attn_lstm_fwd_multi_kernel<true, true>)
                               : reinterpret_cast<const void*>(attn_lstm_fwd_multi_kernel<true, false>);
    return sample ? reinterpret_cast<const void*>(attn_lstm_fwd_multi_kernel<false, true>)
                  : reinterpret_cast<const void*>(attn_lstm_fwd_multi_kernel<false, false>);
}
template <bool OVERLAP>
const void* fwd_pair_variant_of(bool s0, bool s1) {
    if (s0) return s1 ? reinterpret_cast<const void*>(attn_lstm_fwd_pair_kernel<OVERLAP, true, true>)
                      : reinterpret_cast<const void*>(attn_lstm_fwd_pair_kernel<OVERLAP, true, false>);
    return s1 ? reinterpret_cast<const void*>(attn_lstm_fwd_pair_kernel<OVERLAP, false, true>)
              : reinterpret_cast<const void*>(attn_lstm_fwd_pair_kernel<OVERLAP, false, false>);
}
const void* fwd_pair_variant(bool overlap, bool s0, bool s1) {
    return overlap ? fwd_pair_variant_of<true>(s0, s1) : fwd_pair_variant_of<false>(s0, s1);
}

// rows one launch can take: all tiles x 8 members resident, one workgroup per CU
int rows_per_launch() {
    const int cus = pnmn::device_cus();
    return cus >= 8 * MEMBERS ? (cus / (8 * MEMBERS)) * 8 * ROWS : 0;
}

}  // namespace

extern "C" {

int64_t pnmn_attn_lstm_multi_workspace_bytes(int B, int backward) {
    const int chunk = rows_per_launch();
    if (chunk <= 0 || B <= 0) return 0;
    const int tiles = ((B < chunk ? B : chunk) + ROWS - 1) / ROWS;
    int64_t n = (int64_t)pnmn::CLUSTER_SYNC_BYTES;
    if (backward) n += (int64_t)tiles * (2 * MEMBERS * 2 + 2) * ROWS * H * sizeof(float);
    return n;
}

int pnmn_attn_lstm_fwd_multi(const float* xe, const float* etable, const float* enc, const float* mask, const float* h0,
                             const float* w_c, const float* w_hh, const float* w_p, const float* b_p, float* hs, float* cs,
                             float* act, float* ctx, float* probs, int64_t* tokens, int B, int T, int S, int V, int hidden,
                             int sample, int pad_index, int unk_index, int start_index, uint64_t seed, uint64_t row_offset,
                             const int64_t* in_tokens, int64_t in_token_stride, void* workspace, void* stream) {
    if (B <= 0 || T <= 0) return 0;
    if (!enc || !mask || !h0 || !w_c || !w_hh || !hs || !cs || !act || !ctx || !probs || !workspace) return PNMN_EINVAL;
    if (sample ? (!etable || !w_p || !b_p || !tokens) : (!xe && !(etable && in_tokens))) return PNMN_EINVAL;
    if (hidden != H || S < 1 || S > MAXS || (sample && (V < 1 || V > MAXV))) return PNMN_ESHAPE;
    const int chunk = rows_per_launch();
    if (chunk <= 0) return PNMN_ESHAPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    constexpr size_t LDS_LIMIT = 160 * 1024;
    const bool overlap = FWD_FIXED_LDS + sizeof(float) * RW * S * H + FWD_OVERLAP_LDS <= LDS_LIMIT;
    const size_t lds = FWD_FIXED_LDS + sizeof(float) * RW * S * H + (overlap ? FWD_OVERLAP_LDS : 0);
    {   // (the opt-in is per device and per kernel: lds_optin.h; the limit itself, whatever this launch uses)
        static std::atomic<uint64_t> cfg[4] = {{0}, {0}, {0}, {0}};
        if (const int e = pnmn::opt_in_lds(fwd_multi_variant(overlap, sample != 0), LDS_LIMIT, cfg[2 * overlap + (sample != 0)])) return e;
    }
    const auto kernel = reinterpret_cast<void (*)(const MFwdArgs)>(const_cast<void*>(fwd_multi_variant(overlap, sample != 0)));
    for (int r0 = 0; r0 < B; r0 += chunk) {
        const int rows = B - r0 < chunk ? B - r0 : chunk;
        const int tiles = (rows + ROWS - 1) / ROWS;
        int* sync = nullptr;
        hipError_t e = pnmn::cluster_sync_block(workspace, st, &sync);
        if (e != hipSuccess) return (int)e;
        const size_t r = (size_t)r0;
        MFwdArgs a{xe ? xe + r * T * G4 : nullptr, etable, enc + r * S * H, mask + r * S, h0 + r * H, w_c, w_hh, w_p, b_p,
                   hs + r * T * H, cs + r * T * H, act + r * T * G4, ctx + r * T * H, probs + r * T * S,
                   tokens ? tokens + r * T : nullptr, (!sample && !xe) ? in_tokens + r * in_token_stride : nullptr,
                   (long)in_token_stride, sync, rows, T, S, V, tiles, sample,
                   pad_index, unk_index, start_index, seed, row_offset + r};
        hipLaunchKernelGGL(kernel, dim3(8 * MEMBERS * ((tiles + 7) / 8)), dim3(512), lds, st, a);
        e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

int pnmn_attn_lstm_bwd_multi(const float* dhs, const float* act, const float* cs, const float* hs, const float* probs,
                             const float* enc, const float* mask, const float* h0, const float* w_c_t,
                             const float* w_hh_t, float* dgates, float* dctx, float* dscore, float* weights, float* dh0, int B,
                             int T,
                             int S, int hidden, void* workspace, void* stream) {
    if (B <= 0 || T <= 0) return 0;
    if (!dhs || !act || !cs || !hs || !probs || !enc || !mask || !h0 || !w_c_t || !w_hh_t || !dgates || !dctx ||
        !dscore || !weights || !dh0 || !workspace)
        return PNMN_EINVAL;
    if (hidden != H || S < 1 || S > MAXS) return PNMN_ESHAPE;
    const int chunk = rows_per_launch();
    if (chunk <= 0) return PNMN_ESHAPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t lds = BWD_FIXED_LDS + sizeof(float) * RW * S * H;
    {
        static std::atomic<uint64_t> cfg{0};  // (per device: lds_optin.h)
        if (const int e = pnmn::opt_in_lds(reinterpret_cast<const void*>(attn_lstm_bwd_multi_kernel), 160 * 1024, cfg)) return e;
    }
    char* ws = static_cast<char*>(workspace);
    float* x1 = reinterpret_cast<float*>(ws + pnmn::CLUSTER_SYNC_BYTES);
    const int max_tiles = chunk / ROWS;
    float* x2 = x1 + (size_t)((B < chunk ? (B + ROWS - 1) / ROWS : max_tiles)) * 2 * MEMBERS * 2 * ROWS * H;
    for (int r0 = 0; r0 < B; r0 += chunk) {
        const int rows = B - r0 < chunk ? B - r0 : chunk;
        const int tiles = (rows + ROWS - 1) / ROWS;
        int* sync = nullptr;
        hipError_t e = pnmn::cluster_sync_block(workspace, st, &sync);
        if (e != hipSuccess) return (int)e;
        const size_t r = (size_t)r0;
        MBwdArgs a{dhs + r * T * H, act + r * T * G4, cs + r * T * H, hs + r * T * H, probs + r * T * S, enc + r * S * H,
                   mask + r * S, h0 + r * H, w_c_t, w_hh_t, dgates + r * T * G4, dctx + r * T * H, dscore + r * T * S,
                   weights + r * T * S,
                   dh0 + r * H, x1, x2, sync, rows, T, S, tiles};
        hipLaunchKernelGGL(attn_lstm_bwd_multi_kernel, dim3(8 * MEMBERS * ((tiles + 7) / 8)), dim3(512), lds, st, a);
        e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}


// ---- two passes side by side -----------------------------------------------------------------------------------
static inline int padded_tiles(int B) { return 8 * ((((B + ROWS - 1) / ROWS) + 7) / 8); }  // (whole groups of 8 tiles)

// both passes in one launch?  (their tile groups must be resident together)
static bool pair_fits(int Ba, int Bb) {
    const int chunk = rows_per_launch();
    return chunk > 0 && Ba > 0 && Bb > 0 && (padded_tiles(Ba) + padded_tiles(Bb)) * ROWS <= chunk + 0 &&
           (padded_tiles(Ba) + padded_tiles(Bb)) <= 128;
}

int64_t pnmn_attn_lstm_pair_workspace_bytes(int Ba, int Bb, int backward) {
    if (!pair_fits(Ba, Bb)) {
        const int64_t a = pnmn_attn_lstm_multi_workspace_bytes(Ba, backward), b = pnmn_attn_lstm_multi_workspace_bytes(Bb, backward);
        return a > b ? a : b;
    }
    int64_t n = (int64_t)pnmn::CLUSTER_SYNC_BYTES;
    if (backward) n += (int64_t)(padded_tiles(Ba) + padded_tiles(Bb)) * (2 * MEMBERS * 2 + 2) * ROWS * H * sizeof(float);
    return n;
}

int pnmn_attn_lstm_fwd_multi_pair(const pnmn_decoder_fwd_job* ja, const pnmn_decoder_fwd_job* jb, int hidden, void* workspace,
                                  void* stream) {
    if (!ja || !jb || !workspace) return PNMN_EINVAL;
    auto single = [&](const pnmn_decoder_fwd_job* j) {
        return pnmn_attn_lstm_fwd_multi(j->xe, j->etable, j->enc, j->mask, j->h0, j->w_c, j->w_hh, j->w_p, j->b_p, j->hs, j->cs,
                                        j->act, j->ctx, j->probs, j->tokens, j->B, j->T, j->S, j->V, hidden, j->sample, j->pad_index,
                                        j->unk_index, j->start_index, j->seed, j->row_offset, j->in_tokens, j->in_token_stride,
                                        workspace, stream);
    };
    if (!pair_fits(ja->B, jb->B) || ja->T <= 0 || jb->T <= 0) {  // (one after the other: same results)
        const int rc = single(ja);
        return rc != 0 ? rc : single(jb);
    }
    const pnmn_decoder_fwd_job* jobs[2] = {ja, jb};
    for (const pnmn_decoder_fwd_job* j : jobs) {
        if (!j->enc || !j->mask || !j->h0 || !j->w_c || !j->w_hh || !j->hs || !j->cs || !j->act || !j->ctx || !j->probs) return PNMN_EINVAL;
        if (j->sample ? (!j->etable || !j->w_p || !j->b_p || !j->tokens) : (!j->xe && !(j->etable && j->in_tokens))) return PNMN_EINVAL;
        if (hidden != H || j->S < 1 || j->S > MAXS || (j->sample && (j->V < 1 || j->V > MAXV))) return PNMN_ESHAPE;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    constexpr size_t LDS_LIMIT = 160 * 1024;
    const int smax = ja->S > jb->S ? ja->S : jb->S;
    const bool overlap = FWD_FIXED_LDS + sizeof(float) * RW * smax * H + FWD_OVERLAP_LDS <= LDS_LIMIT;
    const size_t lds = FWD_FIXED_LDS + sizeof(float) * RW * smax * H + (overlap ? FWD_OVERLAP_LDS : 0);
    {   // (the opt-in is per device and per kernel: lds_optin.h; the limit itself, whatever this launch uses)
        static std::atomic<uint64_t> cfg[8] = {{0}, {0}, {0}, {0}, {0}, {0}, {0}, {0}};
        const int v = 4 * overlap + 2 * (ja->sample != 0) + (jb->sample != 0);
        if (const int e = pnmn::opt_in_lds(fwd_pair_variant(overlap, ja->sample != 0, jb->sample != 0), LDS_LIMIT, cfg[v])) return e;
    }
    const auto kernel = reinterpret_cast<void (*)(const MFwdArgs, const MFwdArgs, const int)>(
        const_cast<void*>(fwd_pair_variant(overlap, ja->sample != 0, jb->sample != 0)));
    int* sync = nullptr;
    hipError_t e = pnmn::cluster_sync_block(workspace, st, &sync);
    if (e != hipSuccess) return (int)e;
    const int tiles0 = padded_tiles(ja->B);
    auto args = [&](const pnmn_decoder_fwd_job* j, int* sy) {
        return MFwdArgs{j->xe, j->etable, j->enc, j->mask, j->h0, j->w_c, j->w_hh, j->w_p, j->b_p, j->hs, j->cs, j->act, j->ctx,
                        j->probs, j->tokens, (!j->sample && !j->xe) ? j->in_tokens : nullptr, (long)j->in_token_stride, sy, j->B,
                        j->T, j->S, j->V, (j->B + ROWS - 1) / ROWS, j->sample, j->pad_index, j->unk_index, j->start_index,
                        j->seed, j->row_offset};
    };
    const MFwdArgs a0 = args(ja, sync), a1 = args(jb, sync + tiles0 * pnmn::CLUSTER_COUNTER_STRIDE);
    const int groups = (tiles0 + padded_tiles(jb->B)) / 8;
    hipLaunchKernelGGL(kernel, dim3(8 * MEMBERS * groups), dim3(512), lds, st, a0, a1, tiles0);
    return (int)hipGetLastError();
}

int pnmn_attn_lstm_bwd_multi_pair(const pnmn_decoder_bwd_job* ja, const pnmn_decoder_bwd_job* jb, int hidden, void* workspace,
                                  void* stream) {
    if (!ja || !jb || !workspace) return PNMN_EINVAL;
    auto single = [&](const pnmn_decoder_bwd_job* j) {
        return pnmn_attn_lstm_bwd_multi(j->dhs, j->act, j->cs, j->hs, j->probs, j->enc, j->mask, j->h0, j->w_c_t, j->w_hh_t, j->dgates,
                                        j->dctx, j->dscore, j->weights, j->dh0, j->B, j->T, j->S, hidden, workspace, stream);
    };
    if (!pair_fits(ja->B, jb->B) || ja->T <= 0 || jb->T <= 0) {
        const int rc = single(ja);
        return rc != 0 ? rc : single(jb);
    }
    const pnmn_decoder_bwd_job* jobs[2] = {ja, jb};
    for (const pnmn_decoder_bwd_job* j : jobs) {
        if (!j->dhs || !j->act || !j->cs || !j->hs || !j->probs || !j->enc || !j->mask || !j->h0 || !j->w_c_t || !j->w_hh_t ||
            !j->dgates || !j->dctx || !j->dscore || !j->weights || !j->dh0)
            return PNMN_EINVAL;
        if (hidden != H || j->S < 1 || j->S > MAXS) return PNMN_ESHAPE;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int smax = ja->S > jb->S ? ja->S : jb->S;
    const size_t lds = BWD_FIXED_LDS + sizeof(float) * RW * smax * H;
    {
        static std::atomic<uint64_t> cfg{0};  // (per device: lds_optin.h)
        if (const int e = pnmn::opt_in_lds(reinterpret_cast<const void*>(attn_lstm_bwd_pair_kernel), 160 * 1024, cfg)) return e;
    }
    int* sync = nullptr;
    hipError_t e = pnmn::cluster_sync_block(workspace, st, &sync);
    if (e != hipSuccess) return (int)e;
    const int tiles0 = padded_tiles(ja->B), total = tiles0 + padded_tiles(jb->B);
    char* ws = static_cast<char*>(workspace);
    float* x1 = reinterpret_cast<float*>(ws + pnmn::CLUSTER_SYNC_BYTES);
    float* x2 = x1 + (size_t)total * 2 * MEMBERS * 2 * ROWS * H;
    auto args = [&](const pnmn_decoder_bwd_job* j, int first_tile) {
        return MBwdArgs{j->dhs, j->act, j->cs, j->hs, j->probs, j->enc, j->mask, j->h0, j->w_c_t, j->w_hh_t, j->dgates, j->dctx,
                        j->dscore, j->weights, j->dh0, x1 + (size_t)first_tile * 2 * MEMBERS * 2 * ROWS * H,
                        x2 + (size_t)first_tile * 2 * ROWS * H, sync + first_tile * pnmn::CLUSTER_COUNTER_STRIDE, j->B, j->T, j->S,
                        (j->B + ROWS - 1) / ROWS};
    };
    const MBwdArgs a0 = args(ja, 0), a1 = args(jb, tiles0);
    hipLaunchKernelGGL(attn_lstm_bwd_pair_kernel, dim3(8 * MEMBERS * (total / 8)), dim3(512), lds, st, a0, a1, tiles0);
    return (int)hipGetLastError();
}

// ---- three backward passes side by side ----------------------------------------------------------------------
static bool group3_fits(int Ba, int Bb, int Bc) {
    const int chunk = rows_per_launch();
    const int tiles = padded_tiles(Ba) + padded_tiles(Bb) + padded_tiles(Bc);
    return chunk > 0 && Ba > 0 && Bb > 0 && Bc > 0 && tiles * ROWS <= chunk && tiles <= 128;
}

int64_t pnmn_attn_lstm_group3_workspace_bytes(int Ba, int Bb, int Bc, int backward) {
    if (!group3_fits(Ba, Bb, Bc)) {
        const int64_t ab = pnmn_attn_lstm_pair_workspace_bytes(Ba, Bb, backward), c = pnmn_attn_lstm_multi_workspace_bytes(Bc, backward);
        return ab > c ? ab : c;
    }
    int64_t n = (int64_t)pnmn::CLUSTER_SYNC_BYTES;
    if (backward) n += (int64_t)(padded_tiles(Ba) + padded_tiles(Bb) + padded_tiles(Bc)) * (2 * MEMBERS * 2 + 2) * ROWS * H * sizeof(float);
    return n;
}

int pnmn_attn_lstm_bwd_multi_group3(const pnmn_decoder_bwd_job* ja, const pnmn_decoder_bwd_job* jb, const pnmn_decoder_bwd_job* jc,
                                    int hidden, void* workspace, void* stream) {
    if (!ja || !jb || !jc || !workspace) return PNMN_EINVAL;
    if (!group3_fits(ja->B, jb->B, jc->B) || ja->T <= 0 || jb->T <= 0 || jc->T <= 0) {  // (pair + single: same results)
        const int rc = pnmn_attn_lstm_bwd_multi_pair(ja, jb, hidden, workspace, stream);
        return rc != 0 ? rc
                       : pnmn_attn_lstm_bwd_multi(jc->dhs, jc->act, jc->cs, jc->hs, jc->probs, jc->enc, jc->mask, jc->h0, jc->w_c_t,
                                                  jc->w_hh_t, jc->dgates, jc->dctx, jc->dscore, jc->weights, jc->dh0, jc->B, jc->T, jc->S,
                                                  hidden, workspace, stream);
    }
    const pnmn_decoder_bwd_job* jobs[3] = {ja, jb, jc};
    int smax = 0;
    for (const pnmn_decoder_bwd_job* j : jobs) {
        if (!j->dhs || !j->act || !j->cs || !j->hs || !j->probs || !j->enc || !j->mask || !j->h0 || !j->w_c_t || !j->w_hh_t ||
            !j->dgates || !j->dctx || !j->dscore || !j->weights || !j->dh0)
            return PNMN_EINVAL;
        if (hidden != H || j->S < 1 || j->S > MAXS) return PNMN_ESHAPE;
        smax = j->S > smax ? j->S : smax;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t lds = BWD_FIXED_LDS + sizeof(float) * RW * smax * H;
    {
        static std::atomic<uint64_t> cfg{0};  // (per device: lds_optin.h)
        if (const int e = pnmn::opt_in_lds(reinterpret_cast<const void*>(attn_lstm_bwd_group3_kernel), 160 * 1024, cfg)) return e;
    }
    int* sync = nullptr;
    hipError_t e = pnmn::cluster_sync_block(workspace, st, &sync);
    if (e != hipSuccess) return (int)e;
    const int tiles0 = padded_tiles(ja->B), tiles01 = tiles0 + padded_tiles(jb->B), total = tiles01 + padded_tiles(jc->B);
    char* ws = static_cast<char*>(workspace);
    float* x1 = reinterpret_cast<float*>(ws + pnmn::CLUSTER_SYNC_BYTES);
    float* x2 = x1 + (size_t)total * 2 * MEMBERS * 2 * ROWS * H;
    auto args = [&](const pnmn_decoder_bwd_job* j, int first_tile) {
        return MBwdArgs{j->dhs, j->act, j->cs, j->hs, j->probs, j->enc, j->mask, j->h0, j->w_c_t, j->w_hh_t, j->dgates, j->dctx,
                        j->dscore, j->weights, j->dh0, x1 + (size_t)first_tile * 2 * MEMBERS * 2 * ROWS * H,
                        x2 + (size_t)first_tile * 2 * ROWS * H, sync + first_tile * pnmn::CLUSTER_COUNTER_STRIDE, j->B, j->T, j->S,
                        (j->B + ROWS - 1) / ROWS};
    };
    const MBwdArgs a0 = args(ja, 0), a1 = args(jb, tiles0), a2 = args(jc, tiles01);
    hipLaunchKernelGGL(attn_lstm_bwd_group3_kernel, dim3(8 * MEMBERS * (total / 8)), dim3(512), lds, st, a0, a1, a2, tiles0, tiles01);
    return (int)hipGetLastError();
}

}  // extern "C"
