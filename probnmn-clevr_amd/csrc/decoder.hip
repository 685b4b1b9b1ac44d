// Persistent attention-LSTM decoder (SimpleSeq2Seq's decoding loop, hidden = 256) for gfx950.
//
// Per decoding step the reference runs ~25 tiny torch ops per row batch: target embedding, dot-product
// attention over the encoder outputs (bmm, masked softmax, bmm), concat, LSTMCell, output projection,
// softmax, sampling (reference seq2seq_base.py:186-224 -> allennlp _prepare_output_projections).
// Everything in that loop depends on h_{t-1}, but nothing couples different batch rows -- so one
// workgroup owns 16 rows for ALL steps:
//     attention   scores = enc . h, AllenNLP masked softmax, ctx = w . enc   (enc rows from L2)
//     gates       xe_t + ctx W_c^T + h W_hh^T on the matrix cores (W_ih = [W_c | W_e]; the
//                 embedding half xe_t = e_t W_e^T + b comes precomputed for teacher forcing, or is
//                 gathered from the table E' = Emb W_e^T + b when the step's token is sampled here)
//     cell        lane-local (see the layout note in seq2seq.hip)
//     [sampling]  logits = h W_p^T + b, softmax, forbidden tokens, inverse-CDF draw (Philox), next token
// h / ctx / logits live in LDS, c in registers; W_c, W_hh (2 MiB) stream from L2 every step.
// Saved for the backward kernel: activated gates, c, h, ctx, the pre-mask softmax p.
// The output projection / log-softmax / cross-entropy over all steps are batched GEMMs done by the
// caller on the returned hidden states.
//
// Backward (attn_lstm_bwd_kernel) walks the steps in reverse with the same ownership: cell backward
// (lane-local) -> dgates; dctx = dgates W_c and dh = dgates W_hh on the matrix cores; attention
// backward (d weights, AllenNLP masked-softmax backward, d scores) adds into dh and into the rows'
// denc (only this workgroup touches them: plain read-modify-write).  Weight gradients are batched
// GEMMs over the saved dgates / ctx / h done by the caller.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/probnmn_hip.h"
#include "lds_optin.h"
#include "sampling.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int H = 256;
constexpr int G4 = 4 * H;
constexpr int ROWS = 16;
constexpr int LD = H + 4;
constexpr int MAXS = 64;   // encoder positions
constexpr int MAXV = 128;  // sampled vocabulary

using pnmn::sigm;
using pnmn::wmax;
using pnmn::wsum;

struct FwdArgs {
    const float* xe;        // [B][T][4H] teacher-forced embedding projection (+biases), or nullptr
    const float* etable;    // [V][4H]    Emb W_e^T + b  (sampling mode)
    const float* enc;       // [B][S][H]
    const float* mask;      // [B][S] 1/0
    const float* h0;        // [B][H]
    const float* w_c;       // [4H][H]
    const float* w_hh;      // [4H][H]
    const float* w_p;       // [V][H]   (sampling mode)
    const float* b_p;       // [V]
    float* hs;              // [B][T][H]
    float* cs;              // [B][T][H]
    float* act;             // [B][T][4H]
    float* ctx;             // [B][T][H]
    float* probs;           // [B][T][S]  softmax before masking
    int64_t* tokens;        // [B][T] sampled / arg-max tokens (sampling mode)
    const int64_t* in_tokens;  // teacher forcing without xe: step t's input is row in_tokens[row][t] of etable
    long in_stride;
    int B, T, S, V;
    int sample;             // 0: teacher forced, 1: sample, 2: greedy
    int pad, unk, start;
    uint64_t seed, row_offset;
};

__global__ __launch_bounds__(512) void attn_lstm_fwd_kernel(const FwdArgs a) {
    __shared__ __attribute__((aligned(16))) float hl[2][ROWS][LD];
    __shared__ __attribute__((aligned(16))) float cl[ROWS][LD];
    __shared__ float wl[ROWS][MAXS];
    __shared__ float logl[ROWS][MAXV];
    __shared__ int tokl[ROWS];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * ROWS;
    const int T = a.T, S = a.S;

    for (int i = tid; i < ROWS * H; i += 512) {
        const int r = i / H, k = i - r * H;
        hl[0][r][k] = (row0 + r < a.B) ? a.h0[(size_t)(row0 + r) * H + k] : 0.f;
    }
    if (tid < ROWS) tokl[tid] = a.start;
    float creg[2][4];
#pragma unroll
    for (int ut = 0; ut < 2; ++ut)
#pragma unroll
        for (int r = 0; r < 4; ++r) creg[ut][r] = 0.f;
    __syncthreads();

    for (int t = 0; t < T; ++t) {
        const int cur = t & 1, nxt = cur ^ 1;
        // ---------------- attention: wave w owns rows 2w, 2w+1 ----------------
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int rl = 2 * wave + rr;
            const int row = row0 + rl;
            if (row < a.B) {  // wave-uniform
                const f32x4 hv = *reinterpret_cast<const f32x4*>(&hl[cur][rl][4 * lane]);
                const float* er = a.enc + (size_t)row * S * H + 4 * lane;
                float myscore = 0.f;
                // encoder rows are fetched eight at a time so that their L2 latency overlaps
                for (int s0 = 0; s0 < S; s0 += 8) {
                    f32x4 e[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int s = s0 + k < S ? s0 + k : S - 1;
                        e[k] = *reinterpret_cast<const f32x4*>(er + (size_t)s * H);
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float p = wsum(e[k].x * hv.x + e[k].y * hv.y + e[k].z * hv.z + e[k].w * hv.w);
                        if (lane == s0 + k) myscore = p;
                    }
                }
                const float m = lane < S ? a.mask[(size_t)row * S + lane] : 0.f;
                const float v = myscore * m;  // allennlp masked_softmax: softmax(vector * mask) ...
                const float mx = wmax(lane < S ? v : -INFINITY);
                const float ex = lane < S ? expf(v - mx) : 0.f;
                const float p = ex / wsum(ex);
                const float q = p * m;        // ... * mask, renormalised with 1e-13
                const float wgt = q / (wsum(q) + 1e-13f);
                if (lane < S) {
                    wl[rl][lane] = wgt;
                    a.probs[((size_t)row * T + t) * S + lane] = p;
                }
                // context
                f32x4 c4 = f32x4{0.f, 0.f, 0.f, 0.f};
                for (int s0 = 0; s0 < S; s0 += 8) {
                    f32x4 e[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int s = s0 + k < S ? s0 + k : S - 1;
                        e[k] = *reinterpret_cast<const f32x4*>(er + (size_t)s * H);
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float ws = s0 + k < S ? __shfl(wgt, s0 + k) : 0.f;
                        c4 += e[k] * ws;
                    }
                }
                *reinterpret_cast<f32x4*>(&cl[rl][4 * lane]) = c4;
                *reinterpret_cast<f32x4*>(a.ctx + ((size_t)row * T + t) * H + 4 * lane) = c4;
            } else {
                *reinterpret_cast<f32x4*>(&cl[rl][4 * lane]) = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        __syncthreads();

        // ---------------- gates on the matrix cores ----------------
        f32x4 acc[4][2];
#pragma unroll
        for (int gate = 0; gate < 4; ++gate)
#pragma unroll
            for (int ut = 0; ut < 2; ++ut)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rl = 4 * g + r;
                    const int row = row0 + rl;
                    const int n = gate * H + 32 * wave + 16 * ut + li;
                    float v = 0.f;
                    if (row < a.B) {
                        v = a.sample ? a.etable[(size_t)tokl[rl] * G4 + n]
                                     : (a.xe ? a.xe[((size_t)row * T + t) * G4 + n]
                                             : a.etable[(size_t)a.in_tokens[(size_t)row * a.in_stride + t] * G4 + n]);
                    }
                    acc[gate][ut][r] = v;
                }
#pragma unroll 2
        for (int kb = 0; kb < H / 16; ++kb) {
            const f32x4 ac = *reinterpret_cast<const f32x4*>(&cl[li][kb * 16 + 4 * g]);
            const f32x4 ah = *reinterpret_cast<const f32x4*>(&hl[cur][li][kb * 16 + 4 * g]);
#pragma unroll
            for (int gate = 0; gate < 4; ++gate)
#pragma unroll
                for (int ut = 0; ut < 2; ++ut) {
                    const size_t fo = ((size_t)((gate * (H / 16) + 2 * wave + ut) * (H / 16) + kb) * 64 + lane) * 4;
                    const f32x4 bc = *reinterpret_cast<const f32x4*>(a.w_c + fo);  // weights are packed in
                    const f32x4 bh = *reinterpret_cast<const f32x4*>(a.w_hh + fo);  // fragment order (seq2seq.hip)
                    acc[gate][ut] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac.x, bc.x, acc[gate][ut], 0, 0, 0);
                    acc[gate][ut] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac.y, bc.y, acc[gate][ut], 0, 0, 0);
                    acc[gate][ut] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac.z, bc.z, acc[gate][ut], 0, 0, 0);
                    acc[gate][ut] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac.w, bc.w, acc[gate][ut], 0, 0, 0);
                    acc[gate][ut] = __builtin_amdgcn_mfma_f32_16x16x4f32(ah.x, bh.x, acc[gate][ut], 0, 0, 0);
                    acc[gate][ut] = __builtin_amdgcn_mfma_f32_16x16x4f32(ah.y, bh.y, acc[gate][ut], 0, 0, 0);
                    acc[gate][ut] = __builtin_amdgcn_mfma_f32_16x16x4f32(ah.z, bh.z, acc[gate][ut], 0, 0, 0);
                    acc[gate][ut] = __builtin_amdgcn_mfma_f32_16x16x4f32(ah.w, bh.w, acc[gate][ut], 0, 0, 0);
                }
        }
        // ---------------- cell ----------------
#pragma unroll
        for (int ut = 0; ut < 2; ++ut)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rl = 4 * g + r;
                const int row = row0 + rl;
                const int u = 32 * wave + 16 * ut + li;
                const float ig = sigm(acc[0][ut][r]), fg = sigm(acc[1][ut][r]);
                const float gg = tanhf(acc[2][ut][r]), og = sigm(acc[3][ut][r]);
                const float c = fg * creg[ut][r] + ig * gg;
                const float h = og * tanhf(c);
                creg[ut][r] = c;
                hl[nxt][rl][u] = h;
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
        __syncthreads();

        // ---------------- token choice for the next step ----------------
        if (a.sample) {
            const int V = a.V;
            if (16 * wave < V) {  // logits tile: 16 rows x 16 vocabulary entries per wave
                f32x4 lacc = f32x4{0.f, 0.f, 0.f, 0.f};
                const int vn = 16 * wave + li;
                const bool vok = vn < V;
#pragma unroll 4
                for (int kb = 0; kb < H / 16; ++kb) {
                    const f32x4 ah = *reinterpret_cast<const f32x4*>(&hl[nxt][li][kb * 16 + 4 * g]);
                    f32x4 bp = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (vok) bp = *reinterpret_cast<const f32x4*>(a.w_p + (size_t)vn * H + kb * 16 + 4 * g);  // (row-major: tiny)
                    lacc = __builtin_amdgcn_mfma_f32_16x16x4f32(ah.x, bp.x, lacc, 0, 0, 0);
                    lacc = __builtin_amdgcn_mfma_f32_16x16x4f32(ah.y, bp.y, lacc, 0, 0, 0);
                    lacc = __builtin_amdgcn_mfma_f32_16x16x4f32(ah.z, bp.z, lacc, 0, 0, 0);
                    lacc = __builtin_amdgcn_mfma_f32_16x16x4f32(ah.w, bp.w, lacc, 0, 0, 0);
                }
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
                    a.tokens[(size_t)row * T + t] = choice;
                }
            }
            __syncthreads();
        }
    }
}

struct BwdArgs {
    const float* dhs;     // [B][T][H] gradient wrt every h_t (from the batched output projection)
    const float* act;     // saved by the forward
    const float* cs;
    const float* hs;
    const float* ctx;
    const float* probs;
    const float* enc;     // [B][S][H]
    const float* mask;    // [B][S]
    const float* h0;      // [B][H]
    const float* w_c_t;   // [H][4H]  W_c^T
    const float* w_hh_t;  // [H][4H]  W_hh^T
    float* dgates;        // [B][T][4H]  (= d xe)
    float* denc;          // [B][S][H]   accumulated (must be zero on entry)
    float* dh0;           // [B][H]
    int B, T, S;
};

__global__ __launch_bounds__(512) void attn_lstm_bwd_kernel(const BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char braw[];
    float (*dgl)[G4 + 4] = reinterpret_cast<float (*)[G4 + 4]>(braw);                    // [16][1028]
    float (*dctxl)[LD] = reinterpret_cast<float (*)[LD]>(braw + sizeof(float) * ROWS * (G4 + 4));  // [16][260]
    float (*dhl)[LD] = dctxl + ROWS;                                                        // [16][260]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * ROWS;
    const int T = a.T, S = a.S;
    float dc_rec[2][4];
#pragma unroll
    for (int ut = 0; ut < 2; ++ut)
#pragma unroll
        for (int r = 0; r < 4; ++r) dc_rec[ut][r] = 0.f;
    for (int i = tid; i < ROWS * LD; i += 512) (&dhl[0][0])[i] = 0.f;
    __syncthreads();

    for (int t = T - 1; t >= 0; --t) {
        // ---- cell backward (lane-local); dh = dhs_t + recurrent/attention gradient left in dhl ----
#pragma unroll
        for (int ut = 0; ut < 2; ++ut)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rl = 4 * g + r;
                const int row = row0 + rl;
                const int u = 32 * wave + 16 * ut + li;
                float di = 0.f, df = 0.f, dg = 0.f, dout = 0.f, dcp = 0.f;
                if (row < a.B) {
                    const size_t o = ((size_t)row * T + t) * H + u;
                    const float* ar = a.act + ((size_t)row * T + t) * G4;
                    const float ig = ar[u], fg = ar[H + u], gg = ar[2 * H + u], og = ar[3 * H + u];
                    const float c = a.cs[o];
                    const float cp = t > 0 ? a.cs[o - H] : 0.f;
                    const float tc = tanhf(c);
                    const float dh = a.dhs[o] + dhl[rl][u];
                    const float dc = dc_rec[ut][r] + dh * og * (1.f - tc * tc);
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
                dc_rec[ut][r] = dcp;
                dgl[rl][u] = di;
                dgl[rl][H + u] = df;
                dgl[rl][2 * H + u] = dg;
                dgl[rl][3 * H + u] = dout;
            }
        __syncthreads();
        // ---- dctx = dgates W_c, dh_prev = dgates W_hh : [16 x 1024] x [1024 x 256] each ----
        f32x4 accc[2], acch[2];
#pragma unroll
        for (int ut = 0; ut < 2; ++ut) accc[ut] = acch[ut] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
        for (int kb = 0; kb < G4 / 16; ++kb) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(&dgl[li][kb * 16 + 4 * g]);
#pragma unroll
            for (int ut = 0; ut < 2; ++ut) {
                const size_t fo = ((size_t)((2 * wave + ut) * (G4 / 16) + kb) * 64 + lane) * 4;
                const f32x4 bc = *reinterpret_cast<const f32x4*>(a.w_c_t + fo);
                const f32x4 bh = *reinterpret_cast<const f32x4*>(a.w_hh_t + fo);
                accc[ut] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bc.x, accc[ut], 0, 0, 0);
                accc[ut] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bc.y, accc[ut], 0, 0, 0);
                accc[ut] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bc.z, accc[ut], 0, 0, 0);
                accc[ut] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bc.w, accc[ut], 0, 0, 0);
                acch[ut] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bh.x, acch[ut], 0, 0, 0);
                acch[ut] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bh.y, acch[ut], 0, 0, 0);
                acch[ut] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bh.z, acch[ut], 0, 0, 0);
                acch[ut] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bh.w, acch[ut], 0, 0, 0);
            }
        }
#pragma unroll
        for (int ut = 0; ut < 2; ++ut)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rl = 4 * g + r;
                const int u = 32 * wave + 16 * ut + li;
                dctxl[rl][u] = accc[ut][r];
                dhl[rl][u] = acch[ut][r];  // recurrent part; the attention part is added below
            }
        __syncthreads();
        // ---- attention backward: wave w owns rows 2w, 2w+1 ----
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int rl = 2 * wave + rr;
            const int row = row0 + rl;
            if (row >= a.B) continue;  // wave-uniform
            const f32x4 dc4 = *reinterpret_cast<const f32x4*>(&dctxl[rl][4 * lane]);
            const float* hp = t > 0 ? a.hs + ((size_t)row * T + (t - 1)) * H : a.h0 + (size_t)row * H;
            const f32x4 hv = *reinterpret_cast<const f32x4*>(hp + 4 * lane);
            const float* er = a.enc + (size_t)row * S * H + 4 * lane;
            float* dr = a.denc + (size_t)row * S * H + 4 * lane;
            // forward quantities of this (row, step): p (softmax before masking), mask, q, Z, w
            const float m = lane < S ? a.mask[(size_t)row * S + lane] : 0.f;
            const float p = lane < S ? a.probs[((size_t)row * T + t) * S + lane] : 0.f;
            const float q = p * m;
            const float Z = wsum(q) + 1e-13f;
            const float wgt = q / Z;
            // d weights: dw_s = dctx . enc_s
            float dw = 0.f;
            for (int s0 = 0; s0 < S; s0 += 8) {
                f32x4 e[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int s = s0 + k < S ? s0 + k : S - 1;
                    e[k] = *reinterpret_cast<const f32x4*>(er + (size_t)s * H);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float d = wsum(e[k].x * dc4.x + e[k].y * dc4.y + e[k].z * dc4.z + e[k].w * dc4.w);
                    if (lane == s0 + k) dw = d;
                }
            }
            // w = q / Z ; q = p * mask ; p = softmax(score * mask)
            const float dq = dw / Z - wsum(dw * q) / (Z * Z);
            const float dp = dq * m;
            const float dv = p * (dp - wsum(dp * p));
            const float dscore = dv * m;
            // denc_s += w_s * dctx + dscore_s * h_prev ;  dh_prev += sum_s dscore_s * enc_s
            f32x4 dh4 = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int s0 = 0; s0 < S; s0 += 4) {
                f32x4 e[4], d[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int s = s0 + k < S ? s0 + k : S - 1;
                    e[k] = *reinterpret_cast<const f32x4*>(er + (size_t)s * H);
                    d[k] = *reinterpret_cast<const f32x4*>(dr + (size_t)s * H);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (s0 + k < S) {  // wave-uniform
                        const float ws = __shfl(wgt, s0 + k);
                        const float ds = __shfl(dscore, s0 + k);
                        *reinterpret_cast<f32x4*>(dr + (size_t)(s0 + k) * H) = d[k] + dc4 * ws + hv * ds;
                        dh4 += e[k] * ds;
                    }
                }
            }
            f32x4* dst = reinterpret_cast<f32x4*>(&dhl[rl][4 * lane]);
            *dst = *dst + dh4;
        }
        __syncthreads();
    }
    for (int i = tid; i < ROWS * H; i += 512) {
        const int r = i / H, k = i - r * H;
        if (row0 + r < a.B) a.dh0[(size_t)(row0 + r) * H + k] = dhl[r][k];
    }
}

}  // namespace

extern "C" {

int pnmn_attn_lstm_fwd(const float* xe, const float* etable, const float* enc, const float* mask, const float* h0,
                       const float* w_c, const float* w_hh, const float* w_p, const float* b_p, float* hs, float* cs,
                       float* act, float* ctx, float* probs, int64_t* tokens, int B, int T, int S, int V, int hidden,
                       int sample, int pad_index, int unk_index, int start_index, uint64_t seed, uint64_t row_offset,
                       const int64_t* in_tokens, int64_t in_token_stride, void* stream) {
    if (B <= 0 || T <= 0) return 0;
    if (!enc || !mask || !h0 || !w_c || !w_hh || !hs || !cs || !act || !ctx || !probs) return PNMN_EINVAL;
    if (sample ? (!etable || !w_p || !b_p || !tokens) : (!xe && !(etable && in_tokens))) return PNMN_EINVAL;
    if (hidden != H || S < 1 || S > MAXS || (sample && (V < 1 || V > MAXV))) return PNMN_ESHAPE;
    FwdArgs a{xe, etable, enc, mask, h0, w_c, w_hh, w_p, b_p, hs, cs, act, ctx, probs, tokens, (!sample && !xe) ? in_tokens : nullptr,
              (long)in_token_stride, B, T, S, V, sample, pad_index, unk_index, start_index, seed, row_offset};
    hipLaunchKernelGGL(attn_lstm_fwd_kernel, dim3((B + ROWS - 1) / ROWS), dim3(512), 0,
                       static_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}

int pnmn_attn_lstm_bwd(const float* dhs, const float* act, const float* cs, const float* hs, const float* ctx,
                       const float* probs, const float* enc, const float* mask, const float* h0, const float* w_c_t,
                       const float* w_hh_t, float* dgates, float* denc, float* dh0, int B, int T, int S, int hidden,
                       void* stream) {
    if (B <= 0 || T <= 0) return 0;
    if (!dhs || !act || !cs || !hs || !ctx || !probs || !enc || !mask || !h0 || !w_c_t || !w_hh_t || !dgates ||
        !denc || !dh0)
        return PNMN_EINVAL;
    if (hidden != H || S < 1 || S > MAXS) return PNMN_ESHAPE;
    constexpr size_t lds = sizeof(float) * (ROWS * (G4 + 4) + 2 * ROWS * LD);
    static std::atomic<uint64_t> cfg{0};  // (per device: lds_optin.h)
    if (const int e = pnmn::opt_in_lds(reinterpret_cast<const void*>(attn_lstm_bwd_kernel), lds, cfg)) return e;
    BwdArgs a{dhs, act, cs, hs, ctx, probs, enc, mask, h0, w_c_t, w_hh_t, dgates, denc, dh0, B, T, S};
    hipLaunchKernelGGL(attn_lstm_bwd_kernel, dim3((B + ROWS - 1) / ROWS), dim3(512), lds,
                       static_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}

}  // extern "C"
