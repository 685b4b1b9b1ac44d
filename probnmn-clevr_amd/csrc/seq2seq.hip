// Seq2seq kernels for gfx950.  First the point kernels of the step-by-step fallback: the LSTM cell (gate
// non-linearities + state update, forward and backward) and the per-step token sampler (softmax,
// forbidden-token masking, inverse-CDF draw from a counter-based Philox stream or arg-max, log-prob
// gather).  Then the persistent LSTM-layer kernels (one workgroup per 16-row tile, and the multi-CU
// variants that share a tile among 4 or 8 workgroups) -- see the notes above them.
//
// All are bandwidth/latency-bound: one thread per (row, hidden unit) reading the four gate
// pre-activations (coalesced across units), one wave per sampled row with shuffle reductions.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/probnmn_hip.h"
#include "lds_optin.h"
#include "cluster.h"
#include "sampling.h"

namespace {

__device__ __forceinline__ float sigm(float z) { return 1.f / (1.f + expf(-z)); }

// gates: [B][4*Hd] pre-activations in torch order (i, f, g, o)
__global__ __launch_bounds__(256) void lstm_cell_fwd_kernel(const float* __restrict__ gates,
                                                            const float* __restrict__ c_prev, float* __restrict__ h,
                                                            float* __restrict__ c, float* __restrict__ act, int B,
                                                            int Hd) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * Hd) return;
    const int b = idx / Hd, u = idx - b * Hd;
    const float* gr = gates + (size_t)b * 4 * Hd;
    const float i = sigm(gr[u]);
    const float f = sigm(gr[Hd + u]);
    const float g = tanhf(gr[2 * Hd + u]);
    const float o = sigm(gr[3 * Hd + u]);
    const float cn = f * c_prev[idx] + i * g;
    c[idx] = cn;
    h[idx] = o * tanhf(cn);
    if (act) {
        float* ar = act + (size_t)b * 4 * Hd;
        ar[u] = i;
        ar[Hd + u] = f;
        ar[2 * Hd + u] = g;
        ar[3 * Hd + u] = o;
    }
}

__global__ __launch_bounds__(256) void lstm_cell_bwd_kernel(const float* __restrict__ act,
                                                            const float* __restrict__ c_prev,
                                                            const float* __restrict__ c, const float* __restrict__ dh,
                                                            const float* __restrict__ dc_in, float* __restrict__ dgates,
                                                            float* __restrict__ dc_prev, int B, int Hd) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * Hd) return;
    const int b = idx / Hd, u = idx - b * Hd;
    const float* ar = act + (size_t)b * 4 * Hd;
    const float i = ar[u], f = ar[Hd + u], g = ar[2 * Hd + u], o = ar[3 * Hd + u];
    const float tc = tanhf(c[idx]);
    const float gh = dh ? dh[idx] : 0.f;
    float dc = gh * o * (1.f - tc * tc);
    if (dc_in) dc += dc_in[idx];
    float* dg = dgates + (size_t)b * 4 * Hd;
    dg[u] = dc * g * i * (1.f - i);
    dg[Hd + u] = dc * c_prev[idx] * f * (1.f - f);
    dg[2 * Hd + u] = dc * i * (1.f - g * g);
    dg[3 * Hd + u] = gh * tc * o * (1.f - o);
    dc_prev[idx] = dc * f;
}

// ---- Philox4x32-10 (Salmon et al. 2011), one draw per (seed, row, step) ------------------------------
__device__ __forceinline__ void philox_round(uint32_t (&ctr)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * ctr[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * ctr[2];
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    ctr[0] = hi1 ^ ctr[1] ^ k0;
    ctr[1] = lo1;
    ctr[2] = hi0 ^ ctr[3] ^ k1;
    ctr[3] = lo0;
}

__device__ float philox_uniform(uint64_t seed, uint64_t row, uint32_t step) {
    uint32_t ctr[4] = {(uint32_t)row, (uint32_t)(row >> 32), step, 0x9E3779B9u};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(ctr, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return (float)(ctr[0] >> 8) * (1.0f / 16777216.0f);  // [0, 1)
}

__device__ __forceinline__ float wave_max(float v) { return pnmn::wmax(v); }  // (the decoder kernels' DPP reductions:
__device__ __forceinline__ float wave_sum(float v) { return pnmn::wsum(v); }  //  same summation order, same tokens)

// one wave per row; V up to 64*MAXV
constexpr int MAXV = 8;
__global__ __launch_bounds__(256) void sample_tokens_kernel(const float* __restrict__ logits, int64_t* __restrict__ tokens,
                                                            float* __restrict__ logprobs, int B, int V, int greedy,
                                                            uint64_t seed, uint64_t row_offset, uint32_t step,
                                                            int pad, int unk, int start) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= B) return;
    const float* z = logits + (size_t)row * V;
    float v[MAXV];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int j = lane + 64 * k;
        v[k] = j < V ? z[j] : -INFINITY;
        mx = fmaxf(mx, v[k]);
    }
    mx = wave_max(mx);
    float se = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) se += (lane + 64 * k < V) ? expf(v[k] - mx) : 0.f;
    se = wave_sum(se);
    const float lse = mx + logf(se);
    int choice;
    if (greedy) {
        // first index of the maximum
        int best = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            const int j = lane + 64 * k;
            if (j < V && v[k] == mx && j < best) best = j;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int other = __shfl_xor(best, o);
            best = other < best ? other : best;
        }
        choice = best;
    } else {
        // weights = softmax with the forbidden tokens zeroed (seq2seq_base.py:212-214); inverse CDF
        float w[MAXV];
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            const int j = lane + 64 * k;
            const bool ok = j < V && j != pad && j != unk && j != start;
            w[k] = ok ? expf(v[k] - lse) : 0.f;
            tot += w[k];
        }
        tot = wave_sum(tot);
        const float target = philox_uniform(seed, row_offset + (uint64_t)row, step) * tot;
        // token order = index order: chunk k holds indices [64k, 64k+64); scan chunk by chunk
        float before = 0.f;
        choice = -1;
        int last_ok = -1;
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            // inclusive prefix sum over lanes
            float inc = w[k];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const float t = __shfl_up(inc, o);
                if (lane >= o) inc += t;
            }
            const bool hit = (w[k] > 0.f) && (before + inc > target);
            const unsigned long long m = __ballot(hit);
            if (choice < 0 && m) choice = 64 * k + (int)__ffsll((long long)m) - 1;
            const unsigned long long pos = __ballot(w[k] > 0.f);
            if (pos) last_ok = 64 * k + 63 - __clzll((long long)pos);
            before += __shfl(inc, 63);
        }
        if (choice < 0) choice = last_ok;  // round-off at the very end of the CDF
    }
    if (lane == 0) {
        tokens[row] = choice;
        logprobs[row] = z[choice] - lse;  // from the UNMODIFIED log-softmax (seq2seq_base.py:204,220)
    }
}

}  // namespace

extern "C" {

int pnmn_lstm_cell_fwd(const float* gates, const float* c_prev, float* h, float* c, float* act, int B, int Hd,
                       void* stream) {
    if (B <= 0) return 0;
    if (!gates || !c_prev || !h || !c || Hd <= 0) return PNMN_EINVAL;
    const int n = B * Hd;
    hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                       gates, c_prev, h, c, act, B, Hd);
    return (int)hipGetLastError();
}

int pnmn_lstm_cell_bwd(const float* act, const float* c_prev, const float* c, const float* dh, const float* dc_in,
                       float* dgates, float* dc_prev, int B, int Hd, void* stream) {
    if (B <= 0) return 0;
    if (!act || !c_prev || !c || !dgates || !dc_prev || Hd <= 0) return PNMN_EINVAL;
    const int n = B * Hd;
    hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                       act, c_prev, c, dh, dc_in, dgates, dc_prev, B, Hd);
    return (int)hipGetLastError();
}

int pnmn_sample_tokens(const float* logits, int64_t* tokens, float* logprobs, int B, int V, int greedy,
                       uint64_t seed, uint64_t row_offset, uint32_t step, int pad_index, int unk_index,
                       int start_index, void* stream) {
    if (B <= 0) return 0;
    if (!logits || !tokens || !logprobs || V <= 0) return PNMN_EINVAL;
    if (V > 64 * MAXV) return PNMN_ESHAPE;
    hipLaunchKernelGGL(sample_tokens_kernel, dim3((B + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), logits,
                       tokens, logprobs, B, V, greedy, seed, row_offset, step, pad_index, unk_index, start_index);
    return (int)hipGetLastError();
}

}  // extern "C"

// =====================================================================================================
// Persistent LSTM layer (the recurrent half of nn.LSTM over a whole padded sequence).
//
// The input projection of ALL time steps is one big GEMM done by the caller (xp = X W_ih^T + b_ih +
// b_hh); what remains is inherently sequential: gates_t = xp_t + h_{t-1} W_hh^T, then the cell.
// Launching that per step costs T x (GEMM + cell) tiny kernels; here ONE workgroup owns 16 batch rows
// for all T steps: h lives in LDS (double buffered), c and the gate accumulators in registers, W_hh
// (1 MiB for H = 256) is streamed from L2 every step straight into MFMA B operands.  Rows are
// independent, so there is no inter-workgroup communication at all.
//
// Wave w (of 8) owns hidden units [32w, 32w+32) for all four gates: 8 accumulators of 16x16 whose
// lane (li, g) / register r holds (batch row 4g+r, unit 32w + 16*ut + li) -- the i, f, g, o values of
// one (row, unit) sit in the same lane and register of four accumulators, so the cell update is
// lane-local.  v_mfma_f32_16x16x4_f32, K consumed in the permuted order used by the conv kernels
// (lane group g takes k = 4g..4g+3 of each 16-block) so that A (h from LDS) and B (weights from
// global) are single 16-byte loads.  Weights come PRE-PACKED in fragment order
//     packed[n_tile][k_block][lane = 16*g + li][4] = W[16*n_tile + li][16*k_block + 4*g + 0..3]
// so that one wave-wide B load is one contiguous 1 KiB (8 full cache lines) -- with the natural
// row-major layout the same load touches 16 half-used lines and the kernel is bound by the
// texture-address path, 2.5x slower.
// =====================================================================================================
namespace {

typedef float f32x4_ __attribute__((ext_vector_type(4)));
constexpr int LH = 256;        // hidden size the kernel is built for
constexpr int LROWS = 16;      // batch rows per workgroup
constexpr int LLD = LH + 4;    // LDS row stride (floats): 16-byte aligned, breaks the 1 KiB bank period

__global__ __launch_bounds__(512) void lstm_seq_fwd_kernel(const float* __restrict__ xp, const int64_t* __restrict__ tokens,
                                                           long tstride, const float* __restrict__ w_hh,
                                                           float* __restrict__ hs, float* __restrict__ cs,
                                                           float* __restrict__ act, int B, int T) {
    __shared__ __attribute__((aligned(16))) float hl[2][LROWS][LLD];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * LROWS;
    float creg[2][4];
#pragma unroll
    for (int ut = 0; ut < 2; ++ut)
#pragma unroll
        for (int r = 0; r < 4; ++r) creg[ut][r] = 0.f;

    // this lane's four rows of xp (rows past the batch re-read the last one; never stored); with tokens the row of
    // step t + 1 is fetched during step t (see the multi-CU kernel below)
    int64_t xrow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = min(row0 + 4 * g + r, B - 1);
        xrow[r] = tokens ? tokens[(size_t)row * tstride] : (int64_t)row * T;
    }

    for (int t = 0; t < T; ++t) {
        const int cur = t & 1, nxt = cur ^ 1;
        f32x4_ acc[4][2];
        // start from the input projection
#pragma unroll
        for (int gate = 0; gate < 4; ++gate)
#pragma unroll
            for (int ut = 0; ut < 2; ++ut)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[gate][ut][r] = xp[(size_t)xrow[r] * (4 * LH) + gate * LH + 32 * wave + 16 * ut + li];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = min(row0 + 4 * g + r, B - 1);
            xrow[r] = tokens ? tokens[(size_t)row * tstride + min(t + 1, T - 1)] : xrow[r] + 1;
        }
        if (t > 0) {
#pragma unroll 4
            for (int kb = 0; kb < LH / 16; ++kb) {
                const f32x4_ a = *reinterpret_cast<const f32x4_*>(&hl[cur][li][kb * 16 + 4 * g]);
#pragma unroll
                for (int gate = 0; gate < 4; ++gate)
#pragma unroll
                    for (int ut = 0; ut < 2; ++ut) {
                        const int ntile = gate * (LH / 16) + 2 * wave + ut;
                        const f32x4_ b = *reinterpret_cast<const f32x4_*>(w_hh + ((size_t)(ntile * (LH / 16) + kb) * 64 + lane) * 4);
                        acc[gate][ut] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc[gate][ut], 0, 0, 0);
                        acc[gate][ut] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc[gate][ut], 0, 0, 0);
                        acc[gate][ut] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc[gate][ut], 0, 0, 0);
                        acc[gate][ut] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc[gate][ut], 0, 0, 0);
                    }
            }
        }
        // cell update, lane-local
#pragma unroll
        for (int ut = 0; ut < 2; ++ut)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rl = 4 * g + r;
                const int row = row0 + rl;
                const int u = 32 * wave + 16 * ut + li;
                const float ig = sigm(acc[0][ut][r]);
                const float fg = sigm(acc[1][ut][r]);
                const float gg = tanhf(acc[2][ut][r]);
                const float og = sigm(acc[3][ut][r]);
                const float c = fg * creg[ut][r] + ig * gg;
                const float h = og * tanhf(c);
                creg[ut][r] = c;
                hl[nxt][rl][u] = h;
                if (row < B) {
                    const size_t o = ((size_t)row * T + t) * LH + u;
                    hs[o] = h;
                    cs[o] = c;
                    if (act) {
                        float* ar = act + ((size_t)row * T + t) * (4 * LH);
                        ar[u] = ig;
                        ar[LH + u] = fg;
                        ar[2 * LH + u] = gg;
                        ar[3 * LH + u] = og;
                    }
                }
            }
        __syncthreads();  // h_t complete before anyone starts step t+1 (and everyone is done with h_{t-1})
    }
}

// backward: dgates[b][t][4H] (gradient wrt the gate pre-activations = wrt xp), given dhs (gradient wrt
// every output h_t), the saved activated gates and cell states, and W_hh^T ([H][4H], k contiguous).
__global__ __launch_bounds__(512) void lstm_seq_bwd_kernel(const float* __restrict__ dhs, const float* __restrict__ act,
                                                           const float* __restrict__ cs, const float* __restrict__ w_hh_t,
                                                           float* __restrict__ dgates, int B, int T) {
    extern __shared__ __attribute__((aligned(16))) char lraw[];
    float (*dgl)[4 * LH + 4] = reinterpret_cast<float (*)[4 * LH + 4]>(lraw);  // [16][1028]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * LROWS;
    float dh_rec[2][4], dc_rec[2][4];
#pragma unroll
    for (int ut = 0; ut < 2; ++ut)
#pragma unroll
        for (int r = 0; r < 4; ++r) dh_rec[ut][r] = dc_rec[ut][r] = 0.f;

    for (int t = T - 1; t >= 0; --t) {
        // cell backward, lane-local: this lane owns (rows 4g+r, units 32w+16ut+li)
#pragma unroll
        for (int ut = 0; ut < 2; ++ut)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rl = 4 * g + r;
                const int row = row0 + rl;
                const int u = 32 * wave + 16 * ut + li;
                float di = 0.f, df = 0.f, dg = 0.f, dout = 0.f, dcp = 0.f;
                if (row < B) {
                    const size_t o = ((size_t)row * T + t) * LH + u;
                    const float* ar = act + ((size_t)row * T + t) * (4 * LH);
                    const float ig = ar[u], fg = ar[LH + u], gg = ar[2 * LH + u], og = ar[3 * LH + u];
                    const float c = cs[o];
                    const float cp = t > 0 ? cs[o - LH] : 0.f;
                    const float tc = tanhf(c);
                    const float dh = dhs[o] + dh_rec[ut][r];
                    const float dc = dc_rec[ut][r] + dh * og * (1.f - tc * tc);
                    di = dc * gg * ig * (1.f - ig);
                    df = dc * cp * fg * (1.f - fg);
                    dg = dc * ig * (1.f - gg * gg);
                    dout = dh * tc * og * (1.f - og);
                    dcp = dc * fg;
                    float* dr = dgates + ((size_t)row * T + t) * (4 * LH);
                    dr[u] = di;
                    dr[LH + u] = df;
                    dr[2 * LH + u] = dg;
                    dr[3 * LH + u] = dout;
                }
                dc_rec[ut][r] = dcp;
                dgl[rl][u] = di;
                dgl[rl][LH + u] = df;
                dgl[rl][2 * LH + u] = dg;
                dgl[rl][3 * LH + u] = dout;
            }
        __syncthreads();
        // dh_{t-1} = dgates_t @ W_hh : [16 x 1024] x [1024 x 256]; this wave's 32 hidden units
        f32x4_ acc[2];
        acc[0] = f32x4_{0.f, 0.f, 0.f, 0.f};
        acc[1] = f32x4_{0.f, 0.f, 0.f, 0.f};
        if (t > 0) {
#pragma unroll 4
            for (int kb = 0; kb < (4 * LH) / 16; ++kb) {
                const f32x4_ a = *reinterpret_cast<const f32x4_*>(&dgl[li][kb * 16 + 4 * g]);
#pragma unroll
                for (int ut = 0; ut < 2; ++ut) {
                    const int ntile = 2 * wave + ut;
                    const f32x4_ b = *reinterpret_cast<const f32x4_*>(w_hh_t + ((size_t)(ntile * (4 * LH / 16) + kb) * 64 + lane) * 4);
                    acc[ut] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc[ut], 0, 0, 0);
                    acc[ut] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc[ut], 0, 0, 0);
                    acc[ut] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc[ut], 0, 0, 0);
                    acc[ut] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc[ut], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int ut = 0; ut < 2; ++ut)
#pragma unroll
            for (int r = 0; r < 4; ++r) dh_rec[ut][r] = acc[ut][r];
        __syncthreads();  // everyone done reading dgl before it is overwritten
    }
}


// -----------------------------------------------------------------------------------------------------
// Multi-CU variants.  One workgroup per 16 rows runs at the fp32 matrix rate of ONE CU (13.6 us per step
// for H = 256) while the other CUs idle whenever the batch has fewer than 16 x 256 rows.  Here S
// workgroups (S = 4 or 8, all resident at once: the host picks S so that the grid fits the chip) share
// a row tile: workgroup p owns hidden units [p H/S, (p+1) H/S) for all four gates, keeps ITS slice of
// W_hh in registers for the whole sequence (64 or 128 VGPRs) and exchanges only the 16 x H recurrent
// vector with its S-1 partners through L2 once per step:
//   forward   h_t is written to `hs` anyway; partners read h_{t-1} back from there
//   backward  dh_{t-1} = dgates_t W_hh is a sum over gate columns: every workgroup produces a 16 x H
//             partial from its columns, partners add the S slices they own (reduce-scatter through a
//             double-buffered workspace)
// Hand-off: stores -> s_waitcnt vmcnt(0) -> barrier -> one agent-scope release increment of the tile's
// step counter; consumers spin on the counter (bounded: a trap, not a hung GPU, if the partners never
// show up), one agent-scope acquire, barrier.  The counter is monotonic (S per step), so no reset race.
// blockIdx -> (tile, p) keeps a tile's workgroups on one XCD (round-robin dispatch: XCD = blockIdx % 8),
// which makes the exchange an L2 hit; correctness does not depend on that.
// (Measured and rejected, round 2: SIXTEEN members per tile for batches of at most 256 rows -- one 16-column tile
// per gate and member, the eight waves being four gates x two halves of the contraction: an LSTM layer alone runs
// 4.4 -> 3.5 us per step, but the joint step, where the NMN's convolutions share the chip on their own stream, gets
// slower (8.0 -> 8.2 ms at 128 questions): twice the workgroups have to find a free CU at once.)
// (Measured and rejected, round 2: two row tiles per workgroup, alternating, so that one tile's hand-off
// completes behind the other tile's step -- 1024 rows as 32 pairs x 8 members instead of 64 tiles x 4.  The
// hand-off is not what a step waits for: 13.4 us per step against 8.7, the two tiles' chains of wait -> load
// h -> MFMA -> LDS -> cell -> store simply add up in one instruction stream.)
// The operand order of every accumulation is the single-workgroup kernels'; results differ from theirs
// only by the compiler's fma contraction of the cell update (last ulps), and are run-to-run identical.
// -----------------------------------------------------------------------------------------------------

template <int S, bool TOK>
__global__ __launch_bounds__(512) void lstm_seq_fwd_cluster_kernel(const float* __restrict__ xp,
                                                                   const int64_t* __restrict__ tokens, long tstride,
                                                                   const float* __restrict__ w_hh,
                                                                   float* hs, float* __restrict__ cs,
                                                                   float* __restrict__ act, int* sync, int B, int T,
                                                                   int tiles) {
    constexpr int UW = LH / S;          // hidden units of this workgroup
    constexpr int UB = UW / 32;         // 16-unit blocks per wave: waves 2q and 2q+1 split gate q's UW columns
    constexpr int J = UW * 16 / 512;    // (row, unit) pairs per thread in the cell update
    constexpr int GLD = UW + 4;
    __shared__ float gl[4][LROWS][GLD];
    // h_{t-1} of the tile, staged ONCE per step by the whole workgroup (two coalesced 16-byte loads per thread)
    // and read back as MFMA operands from LDS: every wave needs all 16 x 256 values, and 16 scattered 16-byte
    // loads per lane straight from L2 kept the matrix pipe waiting ~4 000 cycles per step (cycle stamps)
    constexpr int HLD = LH + 4;
    __shared__ __attribute__((aligned(16))) float hl[LROWS][HLD];
    int tile, part;
    pnmn::cluster_coords<S>(tile, part);
    if (tile >= tiles) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int row0 = tile * LROWS, u0 = part * UW;
    const int gate = wave >> 1, ub0 = (wave & 1) * UB;
    pnmn::Cluster cl;
    cl.start(sync + tile * pnmn::CLUSTER_COUNTER_STRIDE, S);

    f32x4_ wreg[UB][LH / 16];
#pragma unroll
    for (int ub = 0; ub < UB; ++ub)
#pragma unroll
        for (int kb = 0; kb < LH / 16; ++kb) {
            const int ntile = gate * (LH / 16) + u0 / 16 + ub0 + ub;
            wreg[ub][kb] = *reinterpret_cast<const f32x4_*>(w_hh + ((size_t)(ntile * (LH / 16) + kb) * 64 + lane) * 4);
        }
    float creg[J];
#pragma unroll
    for (int j = 0; j < J; ++j) creg[j] = 0.f;

    // this lane's four rows of xp at step t (rows past the batch read the last row; their results are never
    // stored -- a `row < B ?` around each load makes the compiler issue them one after the other, each behind a
    // vmcnt(0)).  TOK: the row index is fetched one step ahead and only turned into an address at the top of the
    // step that uses it (a use right behind the load would make the wave wait for it -- and, the memory counter
    // being in order, for the xp loads in front of it -- BEFORE it starts polling the hand-off counter: +2.5 us
    // per step, measured)
    int64_t xrow[4];
    const int64_t* trow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = min(row0 + 4 * g + r, B - 1);
        trow[r] = TOK ? tokens + (size_t)row * tstride : nullptr;
        xrow[r] = TOK ? trow[r][0] : (int64_t)row * T;
    }

    for (int t = 0; t < T; ++t) {
        f32x4_ acc[UB];
#pragma unroll
        for (int ub = 0; ub < UB; ++ub)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc[ub][r] = xp[(size_t)xrow[r] * (4 * LH) + gate * LH + u0 + 16 * (ub0 + ub) + li];
#pragma unroll
        for (int r = 0; r < 4; ++r) xrow[r] = TOK ? trow[r][min(t + 1, T - 1)] : xrow[r] + 1;
        if (t > 0) {
            cl.wait();
            {   // both loads in flight before the first LDS store (written as a loop the compiler issues load,
                // vmcnt(0), store, load, vmcnt(0), store: two L2 round trips on the step's critical path)
                static_assert(LROWS * LH / 4 == 2 * 512, "two 16-byte pieces per thread");
                f32x4_ piece[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int i = tid + 512 * k, rl = i / (LH / 4), c4 = i % (LH / 4);
                    const int row = min(row0 + rl, B - 1);
                    piece[k] = *reinterpret_cast<const f32x4_*>(hs + ((size_t)row * T + (t - 1)) * LH + 4 * c4);
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int i = tid + 512 * k, rl = i / (LH / 4), c4 = i % (LH / 4);
                    *reinterpret_cast<f32x4_*>(&hl[rl][4 * c4]) = piece[k];
                }
            }
            __syncthreads();
            f32x4_ a[LH / 16];
#pragma unroll
            for (int kb = 0; kb < LH / 16; ++kb) a[kb] = *reinterpret_cast<const f32x4_*>(&hl[li][kb * 16 + 4 * g]);
#pragma unroll
            for (int kb = 0; kb < LH / 16; ++kb)
#pragma unroll
                for (int ub = 0; ub < UB; ++ub) {
                    acc[ub] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kb].x, wreg[ub][kb].x, acc[ub], 0, 0, 0);
                    acc[ub] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kb].y, wreg[ub][kb].y, acc[ub], 0, 0, 0);
                    acc[ub] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kb].z, wreg[ub][kb].z, acc[ub], 0, 0, 0);
                    acc[ub] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kb].w, wreg[ub][kb].w, acc[ub], 0, 0, 0);
                }
        }
#pragma unroll
        for (int ub = 0; ub < UB; ++ub)
#pragma unroll
            for (int r = 0; r < 4; ++r) gl[gate][4 * g + r][16 * (ub0 + ub) + li] = acc[ub][r];
        __syncthreads();
        // cell update: thread -> (row, unit) pairs, all four gates of a pair from LDS
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int idx = tid + 512 * j;
            const int rl = idx / UW, ul = idx % UW;
            const int row = row0 + rl, u = u0 + ul;
            const float ig = sigm(gl[0][rl][ul]);
            const float fg = sigm(gl[1][rl][ul]);
            const float gg = tanhf(gl[2][rl][ul]);
            const float og = sigm(gl[3][rl][ul]);
            const float c = fg * creg[j] + ig * gg;
            const float h = og * tanhf(c);
            creg[j] = c;
            if (row < B) {
                const size_t o = ((size_t)row * T + t) * LH + u;
                hs[o] = h;
                cs[o] = c;
                if (act) {
                    float* ar = act + ((size_t)row * T + t) * (4 * LH);
                    ar[u] = ig;
                    ar[LH + u] = fg;
                    ar[2 * LH + u] = gg;
                    ar[3 * LH + u] = og;
                }
            }
        }
        if (t + 1 < T) cl.signal();  // also the barrier that protects gl for the next step
    }
    cl.finish();
}

template <int S>
__global__ __launch_bounds__(512) void lstm_seq_bwd_cluster_kernel(const float* __restrict__ dhs,
                                                                   const float* __restrict__ act,
                                                                   const float* __restrict__ cs,
                                                                   const float* __restrict__ w_hh_t,
                                                                   float* __restrict__ dgates, float* px, int* sync,
                                                                   int B, int T, int tiles) {
    constexpr int UW = LH / S;
    constexpr int KB = 4 * UW / 16;     // k blocks of this workgroup's gate columns
    constexpr int J = UW * 16 / 512;
    constexpr int DLD = 4 * UW + 4;
    __shared__ __attribute__((aligned(16))) float dgl[LROWS][DLD];
    int tile, part;
    pnmn::cluster_coords<S>(tile, part);
    if (tile >= tiles) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int row0 = tile * LROWS, u0 = part * UW;
    pnmn::Cluster cl;
    cl.start(sync + tile * pnmn::CLUSTER_COUNTER_STRIDE, S);
    float* ptile = px + (size_t)tile * 2 * S * LROWS * LH;  // [parity][source part][16][H]

    // this wave's two 16-unit output tiles x this workgroup's gate columns, register resident
    f32x4_ wreg[2][KB];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int kl = 0; kl < KB; ++kl) {
            const int q = kl / (UW / 16), jj = kl % (UW / 16);
            const int kb = (q * LH + u0) / 16 + jj;
            wreg[nt][kl] = *reinterpret_cast<const f32x4_*>(w_hh_t + ((size_t)((2 * wave + nt) * (4 * LH / 16) + kb) * 64 + lane) * 4);
        }
    float dh_rec[J], dc_rec[J];
#pragma unroll
    for (int j = 0; j < J; ++j) dh_rec[j] = dc_rec[j] = 0.f;

    for (int t = T - 1; t >= 0; --t) {
        // everything of step t that does not depend on the recurrence, before the wait
        float ig[J], fg[J], gg[J], og[J], cc[J], cp[J], dho[J];
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int idx = tid + 512 * j;
            const int rl = idx / UW, ul = idx % UW;
            const int row = row0 + rl, u = u0 + ul;
            ig[j] = fg[j] = gg[j] = og[j] = cc[j] = cp[j] = dho[j] = 0.f;
            if (row < B) {
                const size_t o = ((size_t)row * T + t) * LH + u;
                const float* ar = act + ((size_t)row * T + t) * (4 * LH);
                ig[j] = ar[u], fg[j] = ar[LH + u], gg[j] = ar[2 * LH + u], og[j] = ar[3 * LH + u];
                cc[j] = cs[o];
                cp[j] = t > 0 ? cs[o - LH] : 0.f;
                dho[j] = dhs[o];
            }
        }
        if (t < T - 1) {
            cl.wait();
            const float* pp = ptile + (size_t)((t + 1) & 1) * S * LROWS * LH;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int idx = tid + 512 * j;
                const int rl = idx / UW, ul = idx % UW;
                float sum = 0.f;
#pragma unroll
                for (int s = 0; s < S; ++s) sum += pp[((size_t)s * LROWS + rl) * LH + u0 + ul];
                dh_rec[j] = sum;
            }
        }
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int idx = tid + 512 * j;
            const int rl = idx / UW, ul = idx % UW;
            const int row = row0 + rl, u = u0 + ul;
            float di = 0.f, df = 0.f, dg = 0.f, dout = 0.f, dcp = 0.f;
            if (row < B) {
                const float tc = tanhf(cc[j]);
                const float dh = dho[j] + dh_rec[j];
                const float dc = dc_rec[j] + dh * og[j] * (1.f - tc * tc);
                di = dc * gg[j] * ig[j] * (1.f - ig[j]);
                df = dc * cp[j] * fg[j] * (1.f - fg[j]);
                dg = dc * ig[j] * (1.f - gg[j] * gg[j]);
                dout = dh * tc * og[j] * (1.f - og[j]);
                dcp = dc * fg[j];
                float* dr = dgates + ((size_t)row * T + t) * (4 * LH);
                dr[u] = di;
                dr[LH + u] = df;
                dr[2 * LH + u] = dg;
                dr[3 * LH + u] = dout;
            }
            dc_rec[j] = dcp;
            dgl[rl][ul] = di;
            dgl[rl][UW + ul] = df;
            dgl[rl][2 * UW + ul] = dg;
            dgl[rl][3 * UW + ul] = dout;
        }
        if (t == 0) break;
        __syncthreads();
        f32x4_ acc[2];
        acc[0] = f32x4_{0.f, 0.f, 0.f, 0.f};
        acc[1] = f32x4_{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kl = 0; kl < KB; ++kl) {
            const f32x4_ a = *reinterpret_cast<const f32x4_*>(&dgl[li][kl * 16 + 4 * g]);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                // (weights as the A operand: the product comes out transposed, each lane holding four
                // consecutive units of one row -- one 16-byte store per tile instead of four 4-byte ones)
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[nt][kl].x, a.x, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[nt][kl].y, a.y, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[nt][kl].z, a.z, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[nt][kl].w, a.w, acc[nt], 0, 0, 0);
            }
        }
        float* po = ptile + ((size_t)(t & 1) * S + part) * LROWS * LH;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
            *reinterpret_cast<f32x4_*>(po + (size_t)li * LH + 16 * (2 * wave + nt) + 4 * g) = acc[nt];
        cl.signal();  // also: everyone is done reading dgl
    }
    cl.finish();
}

using pnmn::cluster_split;
constexpr size_t SYNC_BYTES = pnmn::CLUSTER_SYNC_BYTES;

}  // namespace

extern "C" {

// Rows one multi-CU launch covers: the whole batch when its tiles fit the chip, else (a large batch on a chip
// with CUs reserved for RCCL, pnmn_cluster_reserve_cus) as many whole groups of 8 tiles as fit with four members
// per tile -- the batch then runs as several launches over row ranges, one after the other.
static int lstm_chunk_tiles(int tiles) {
    if (cluster_split(tiles)) return tiles;
    const int most = pnmn::cluster_max_tiles();
    return most > 0 && most < tiles ? most : 0;
}

int64_t pnmn_lstm_seq_workspace_bytes(int B, int backward) {
    const int tiles = (B + LROWS - 1) / LROWS;
    const int ctiles = lstm_chunk_tiles(tiles);
    if (ctiles == 0) return 0;
    const int S = cluster_split(ctiles);
    return (int64_t)SYNC_BYTES + (backward ? (int64_t)ctiles * 2 * S * LROWS * LH * sizeof(float) : 0);
}

int pnmn_cluster_reserve_cus(int cus) {
    if (cus >= 0) pnmn::cluster_reserved_cus() = cus;
    return pnmn::device_cus();
}

int pnmn_lstm_seq_fwd(const float* xp, const int64_t* tokens, int64_t token_stride, const float* w_hh, float* hs,
                      float* cs, float* act, int B, int T, int hidden, void* workspace, void* stream) {
    if (B <= 0 || T <= 0) return 0;
    if (!xp || !w_hh || !hs || !cs) return PNMN_EINVAL;
    const long tstride = (long)token_stride;
    if (hidden != LH) return PNMN_ESHAPE;
    const int tiles = (B + LROWS - 1) / LROWS;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int ctiles = workspace ? lstm_chunk_tiles(tiles) : 0;
    if (ctiles) {
        int* sync = nullptr;
        hipError_t e = pnmn::cluster_sync_block(workspace, st, &sync);
        if (e != hipSuccess) return (int)e;
        for (int t0 = 0; t0 < tiles; t0 += ctiles) {
            const int nt = tiles - t0 < ctiles ? tiles - t0 : ctiles;
            const int S = cluster_split(nt);
            if (!S) return PNMN_ESHAPE;
            const size_t r0 = (size_t)t0 * LROWS;
            const int rows = (int)((size_t)B - r0 < (size_t)nt * LROWS ? (size_t)B - r0 : (size_t)nt * LROWS);
            const dim3 grid(8 * S * ((nt + 7) / 8));
            auto kern = S == 8 ? (tokens ? lstm_seq_fwd_cluster_kernel<8, true> : lstm_seq_fwd_cluster_kernel<8, false>)
                               : (tokens ? lstm_seq_fwd_cluster_kernel<4, true> : lstm_seq_fwd_cluster_kernel<4, false>);
            // (with tokens, xp is the per-token table: only the token rows move with the range)
            hipLaunchKernelGGL(kern, grid, dim3(512), 0, st, tokens ? xp : xp + r0 * T * (4 * LH),
                               tokens ? tokens + r0 * tstride : nullptr, tstride, w_hh, hs + r0 * T * LH, cs + r0 * T * LH,
                               act ? act + r0 * T * (4 * LH) : nullptr, sync, rows, T, nt);
            if ((e = hipGetLastError()) != hipSuccess) return (int)e;
        }
        return 0;
    }
    hipLaunchKernelGGL(lstm_seq_fwd_kernel, dim3(tiles), dim3(512), 0, st, xp, tokens, tstride, w_hh, hs, cs, act, B, T);
    return (int)hipGetLastError();
}

int pnmn_lstm_seq_bwd(const float* dhs, const float* act, const float* cs, const float* w_hh_t, float* dgates, int B,
                      int T, int hidden, void* workspace, void* stream) {
    if (B <= 0 || T <= 0) return 0;
    if (!dhs || !act || !cs || !w_hh_t || !dgates) return PNMN_EINVAL;
    if (hidden != LH) return PNMN_ESHAPE;
    const int tiles = (B + LROWS - 1) / LROWS;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int ctiles = workspace ? lstm_chunk_tiles(tiles) : 0;
    if (ctiles) {
        int* sync = nullptr;
        hipError_t e = pnmn::cluster_sync_block(workspace, st, &sync);
        if (e != hipSuccess) return (int)e;
        float* px = reinterpret_cast<float*>(static_cast<char*>(workspace) + SYNC_BYTES);
        for (int t0 = 0; t0 < tiles; t0 += ctiles) {  // (row ranges, as in pnmn_lstm_seq_fwd; px is reused)
            const int nt = tiles - t0 < ctiles ? tiles - t0 : ctiles;
            const int S = cluster_split(nt);
            if (!S) return PNMN_ESHAPE;
            const size_t r0 = (size_t)t0 * LROWS;
            const int rows = (int)((size_t)B - r0 < (size_t)nt * LROWS ? (size_t)B - r0 : (size_t)nt * LROWS);
            const dim3 grid(8 * S * ((nt + 7) / 8));
            const float* d = dhs + r0 * T * LH;
            const float* a = act + r0 * T * (4 * LH);
            const float* c = cs + r0 * T * LH;
            float* dg = dgates + r0 * T * (4 * LH);
            if (S == 8)
                hipLaunchKernelGGL(lstm_seq_bwd_cluster_kernel<8>, grid, dim3(512), 0, st, d, a, c, w_hh_t, dg, px, sync, rows, T, nt);
            else
                hipLaunchKernelGGL(lstm_seq_bwd_cluster_kernel<4>, grid, dim3(512), 0, st, d, a, c, w_hh_t, dg, px, sync, rows, T, nt);
            if ((e = hipGetLastError()) != hipSuccess) return (int)e;
        }
        return 0;
    }
    constexpr size_t lds = (size_t)LROWS * (4 * LH + 4) * sizeof(float);
    static std::atomic<uint64_t> cfg{0};  // (per device: lds_optin.h)
    if (const int e = pnmn::opt_in_lds(reinterpret_cast<const void*>(lstm_seq_bwd_kernel), lds, cfg)) return e;
    hipLaunchKernelGGL(lstm_seq_bwd_kernel, dim3(tiles), dim3(512), lds, st, dhs, act, cs, w_hh_t, dgates, B, T);
    return (int)hipGetLastError();
}

}  // extern "C"
