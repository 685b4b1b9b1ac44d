// Glue kernels of the seq2seq passes for gfx950: the token bookkeeping, masking, state selection and
// embedding gradients that the reference does with chains of tiny tensor ops (allennlp's
// add_sentence_boundary_token_ids / get_text_field_mask / get_final_encoder_states, reference
// probnmn/modules/seq2seq_base.py:97-141,278-293) -- one launch each.  At 128 questions per GPU the step is
// bound by the NUMBER of launches (host time and ~5 us of GPU time apiece), not by their work.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/probnmn_hip.h"

namespace {

// ---- sentence boundaries + masks -------------------------------------------------------------------------
// One wave per row.  tokens [B][T] (row stride `tstride`), right padded.
//   full[b] = [bos, tokens[b][0..T), 0]  with  full[b][1 + n_b] = eos,  n_b = #(tokens[b] != pad)      ([B][T+2])
// drop_first = 0: out = full (T+2 columns);  1: out = full[:, 1:] (T+1 columns: what an encoder reads).
// fmask = (out != pad) as float, last[b] = #(out[b] != pad) - 1 (the index get_final_encoder_states gathers).
__global__ __launch_bounds__(256) void token_prep_kernel(const int64_t* __restrict__ tokens, int64_t tstride, int B, int T,
                                                         int pad, int bos, int eos, int drop_first,
                                                         int64_t* __restrict__ out, float* __restrict__ fmask,
                                                         int* __restrict__ last) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (b >= B) return;
    const int64_t* row = tokens + (size_t)b * tstride;
    int n = 0;
    for (int t = lane; t < T; t += 64) n += row[t] != pad;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
    const int W = T + 2 - drop_first;
    int valid = 0;
    for (int j = lane; j < W; j += 64) {
        const int f = j + drop_first;  // column of the full row
        int64_t v = (f == 0) ? bos : (f <= T ? row[f - 1] : 0);
        if (f == n + 1) v = eos;
        out[(size_t)b * W + j] = v;
        const bool m = v != pad;
        if (fmask) fmask[(size_t)b * W + j] = m ? 1.f : 0.f;
        valid += m;
    }
    if (last) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) valid += __shfl_xor(valid, o);
        if (lane == 0) last[b] = valid - 1;
    }
}

// ---- rows of token matrices gathered / concatenated / padded into one matrix --------------------------------
// dst [sum of rows][W]: segment s contributes `rows` rows, row i = src[(idx ? idx[i] : i) * stride + 0 .. width) followed
// by `pad` up to W columns.  What a training iteration does with index_select / cat / F.pad on its question and program
// matrices (reference question_coding_trainer.py:128-160: the supervised / unsupervised row subsets) in one launch.
struct TokenSegs {
    pnmn_token_seg s[PNMN_TOKEN_SEGS];
    int n;
};
__global__ __launch_bounds__(256) void token_rows_kernel(const TokenSegs segs, int64_t* __restrict__ dst, int W, int64_t pad,
                                                         int total) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= total) return;
    int k = 0, first = 0;
    while (k + 1 < segs.n && r >= first + segs.s[k].rows) first += segs.s[k].rows, ++k;
    const pnmn_token_seg sg = segs.s[k];
    const int64_t i = sg.index ? sg.index[r - first] : (int64_t)(r - first);
    const int64_t* row = sg.src + i * sg.row_stride;
    for (int w = lane; w < W; w += 64) dst[(size_t)r * W + w] = w < sg.width ? row[w] : pad;
}

// ---- trim predictions at the first @end@ (reference seq2seq_base.py:278-293) -------------------------------
// keep a row up to and including its first `end`; a row that starts with `end` becomes all zeros; a row
// without `end` is kept whole.  One wave per row.
__global__ __launch_bounds__(256) void trim_predictions_kernel(const int64_t* __restrict__ raw, int B, int T, int end,
                                                               int64_t* __restrict__ out) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (b >= B) return;
    const int64_t* row = raw + (size_t)b * T;
    int first = T;  // first column holding `end`
    for (int t = lane; t < T; t += 64)
        if (row[t] == end && t < first) first = t;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) first = min(first, __shfl_xor(first, o));
    for (int t = lane; t < T; t += 64) {
        const bool keep = (first == T) || (first > 0 && t <= first);
        out[(size_t)b * T + t] = keep ? row[t] : 0;
    }
}

// ---- zero the padded steps of an encoder output and gather each row's last valid state ---------------------
//   enc[b][t] = hs[b][t] * fmask[b][t];   hlast[b] = enc[b][last[b]]   (last[b] < 0 counts from the end, as a
//   negative index does)
__global__ __launch_bounds__(256) void mask_last_fwd_kernel(const float* __restrict__ hs, const float* __restrict__ fmask,
                                                            const int* __restrict__ last, int T, int H,
                                                            float* __restrict__ enc, float* __restrict__ hlast) {
    const int b = blockIdx.x;
    int l = last[b];
    if (l < 0) l += T;
    const int h4 = H >> 2;
    const float4* src = reinterpret_cast<const float4*>(hs + (size_t)b * T * H);
    float4* dst = reinterpret_cast<float4*>(enc + (size_t)b * T * H);
    for (int i = blockIdx.y * 256 + threadIdx.x; i < T * h4; i += gridDim.y * 256) {
        const int t = i / h4;
        const float m = fmask[(size_t)b * T + t];
        float4 v = src[i];
        v.x *= m, v.y *= m, v.z *= m, v.w *= m;
        dst[i] = v;
        if (t == l) reinterpret_cast<float4*>(hlast + (size_t)b * H)[i - t * h4] = v;
    }
}

//   dhs[b][t] = (denc[b][t] + [t == last[b]] dhlast[b]) * fmask[b][t]        (denc / dhlast may be null)
__global__ __launch_bounds__(256) void mask_last_bwd_kernel(const float* __restrict__ denc, const float* __restrict__ dhlast,
                                                            const float* __restrict__ fmask, const int* __restrict__ last,
                                                            int T, int H, float* __restrict__ dhs) {
    const int b = blockIdx.x;
    int l = last[b];
    if (l < 0) l += T;
    const int h4 = H >> 2;
    const float4* src = denc ? reinterpret_cast<const float4*>(denc + (size_t)b * T * H) : nullptr;
    float4* dst = reinterpret_cast<float4*>(dhs + (size_t)b * T * H);
    for (int i = blockIdx.y * 256 + threadIdx.x; i < T * h4; i += gridDim.y * 256) {
        const int t = i / h4;
        const float m = fmask[(size_t)b * T + t];
        float4 v = src ? src[i] : float4{0.f, 0.f, 0.f, 0.f};
        if (dhlast && t == l) {
            const float4 d = reinterpret_cast<const float4*>(dhlast + (size_t)b * H)[i - t * h4];
            v.x += d.x, v.y += d.y, v.z += d.z, v.w += d.w;
        }
        v.x *= m, v.y *= m, v.z *= m, v.w *= m;
        dst[i] = v;
    }
}

// ---- embedding gradient for a small vocabulary ------------------------------------------------------------
//   dw[v][c] = sum over rows r with token(r) == v of dy[r][c]                       (V <= 128; dw need not be zeroed)
// rows = (b, t) of tokens [B][T]; `shift` = 1: row (b, t) takes the token of (b, t - 1) and `start` for t = 0 (the
// input token of decoding step t); token `skip` and tokens outside [0, V) contribute nothing.
// Two launches.  (1) one workgroup buckets the row indices by token (LDS histogram, prefix sums, scatter), cuts
// every bucket into chunks of 64 rows and zeroes dw; (2) one workgroup per (chunk, 1024 columns) adds its rows up
// in registers -- whole 4 KiB rows, eight in flight per thread -- and adds the result into dw with 16 bytes of
// global atomics per thread.  (Round 2's first version summed into an LDS table with ds_add_f32: 0.7 TB/s on dense
// gradients -- LDS float atomics retire a few lanes per cycle -- and 12 M global atomics for the table flushes.)
constexpr int EG_CHUNK = 64;

// (b, t) of row r by a multiply instead of a division: the bucketing kernel is ONE workgroup walking every row
// twice, ~40 instructions of integer division per row were most of its time.  magic = ceil(2^32 / T); exact
// for r T < 2^32 (the launcher checks).
__device__ __forceinline__ int eg_token(const int64_t* __restrict__ tokens, int64_t tok_bstride, int T, unsigned magic, int shift,
                                        int start, int skip, int V, int r) {
    int b = magic ? (int)__umulhi((unsigned)r, magic) : r;  // (magic == 0: T == 1)
    b -= (b * T > r);
    const int t = r - b * T;
    const int64_t raw = tokens[(size_t)b * tok_bstride + (shift ? max(t - 1, 0) : t)];  // (unconditional: loads of a batch overlap)
    const int64_t v = (shift && t == 0) ? (int64_t)start : raw;
    return (v == skip || v < 0 || v >= V) ? -1 : (int)v;
}

// workspace: order[R] | chunks[slices][3 * max_per] | count[slices]
__global__ __launch_bounds__(1024) void embedding_rows_kernel(const int64_t* __restrict__ tokens, int64_t tok_bstride, int B,
                                                              int T, unsigned magic, int V, int shift, int start, int skip, int C,
                                                              int* __restrict__ order, int* __restrict__ chunks, int max_per,
                                                              int* __restrict__ count, float* __restrict__ dw, int keep) {
    // one histogram per wave: 47 k rows through 93 shared counters would queue up behind each other
    __shared__ int whist[16][128], hist[128], rowbase[128], chbase[128];
    const int tid = threadIdx.x, wave = tid >> 6;
    // workgroup g buckets its own slice of the rows [lo, R) into its own part of `order` / `chunks`: no exchange
    // between workgroups, at the price of a few more partly filled chunks
    const int per = (B * T + gridDim.x - 1) / gridDim.x;
    const int lo = blockIdx.x * per, R = min(B * T, lo + per);
    chunks += 3 * (size_t)blockIdx.x * max_per;
    count += blockIdx.x;
    for (int i = tid; i < 16 * 128; i += 1024) (&whist[0][0])[i] = 0;
    __syncthreads();
    constexpr int UN = 8;  // token loads in flight per thread (one at a time, a pass is 46 L2 round trips at 47 k rows)
    for (int r0 = lo + tid; r0 < R; r0 += 1024 * UN) {
        int v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) v[u] = eg_token(tokens, tok_bstride, T, magic, shift, start, skip, V, min(r0 + 1024 * u, R - 1));
#pragma unroll
        for (int u = 0; u < UN; ++u)
            if (r0 + 1024 * u < R && v[u] >= 0) atomicAdd(&whist[wave][v[u]], 1);
    }
    if (!keep)  // (keep: the sums are added to what dw holds -- a second pass over other rows of the same table)
        for (size_t i = blockIdx.x * 1024 + tid; i < (size_t)V * C / 4; i += 1024 * gridDim.x)
            reinterpret_cast<float4*>(dw)[i] = float4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    if (tid < 128) {  // counts -> this wave's first position inside the token's bucket
        int run = 0;
        for (int w = 0; w < 16; ++w) {
            const int n = whist[w][tid];
            whist[w][tid] = run;
            run += n;
        }
        hist[tid] = run;
    }
    __syncthreads();
    if (tid == 0) {
        int rows = lo, ch = 0;
        for (int v = 0; v < V; ++v) {
            rowbase[v] = rows, chbase[v] = ch;
            rows += hist[v];
            ch += (hist[v] + EG_CHUNK - 1) / EG_CHUNK;
        }
        *count = ch;
    }
    __syncthreads();
    for (int r0 = lo + tid; r0 < R; r0 += 1024 * UN) {
        int v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) v[u] = eg_token(tokens, tok_bstride, T, magic, shift, start, skip, V, min(r0 + 1024 * u, R - 1));
#pragma unroll
        for (int u = 0; u < UN; ++u)
            if (r0 + 1024 * u < R && v[u] >= 0) order[rowbase[v[u]] + atomicAdd(&whist[wave][v[u]], 1)] = r0 + 1024 * u;
    }
    for (int v = tid; v < V; v += 1024) {
        const int n = hist[v];
        for (int j = 0; j * EG_CHUNK < n; ++j) {
            int* c = chunks + 3 * (chbase[v] + j);
            c[0] = v, c[1] = rowbase[v] + j * EG_CHUNK, c[2] = rowbase[v] + min(n, (j + 1) * EG_CHUNK);
        }
    }
}

__global__ __launch_bounds__(256) void embedding_sum_kernel(const float* __restrict__ dy, const int* __restrict__ order,
                                                            const int* __restrict__ chunks, int max_per,
                                                            const int* __restrict__ count, int C, float* __restrict__ dw) {
    const int slice = blockIdx.x / max_per, j = blockIdx.x % max_per;
    if (j >= count[slice]) return;
    const int c = (blockIdx.y * 256 + threadIdx.x) * 4;
    if (c >= C) return;
    const int* ch = chunks + 3 * ((size_t)slice * max_per + j);
    const int v = ch[0], b = ch[1], e = ch[2];
    constexpr int UN = 8;
    float4 acc = float4{0.f, 0.f, 0.f, 0.f};
    for (int i = b; i < e; i += UN) {
        float4 q[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) q[u] = *reinterpret_cast<const float4*>(dy + (size_t)order[min(i + u, e - 1)] * C + c);
#pragma unroll
        for (int u = 0; u < UN; ++u)
            if (i + u < e) acc.x += q[u].x, acc.y += q[u].y, acc.z += q[u].z, acc.w += q[u].w;
    }
    float* dst = dw + (size_t)v * C + c;
    unsafeAtomicAdd(dst, acc.x);
    unsafeAtomicAdd(dst + 1, acc.y);
    unsafeAtomicAdd(dst + 2, acc.z);
    unsafeAtomicAdd(dst + 3, acc.w);
}

// ---- derived parameters of the recurrent kernels, one launch for a whole model ------------------------------
// kind 0: dst = MFMA-fragment order of the [n][k] matrix src (row stride ld):
//             dst[i][j][a][r][c] = src[(16 i + r) ld + 16 j + 4 a + c]      ([n/16][k/16][4][16][4])
// kind 1: the same of the TRANSPOSE of src ([k][n] as stored, row stride ld): M[x][y] = src[y ld + x]
// kind 2: dst[i] = src[i] + src2[i], i < n   (b_ih + b_hh)
__global__ __launch_bounds__(256) void derive_params_kernel(const pnmn_derive_job* __restrict__ jobs) {
    const pnmn_derive_job jb = jobs[blockIdx.y];
    const int q = blockIdx.x * 256 + threadIdx.x;  // group of four consecutive dst elements
    if (jb.kind == 2) {
        if (q * 4 < jb.n) {
            const float4 a = reinterpret_cast<const float4*>(jb.src)[q], b = reinterpret_cast<const float4*>(jb.src2)[q];
            reinterpret_cast<float4*>(jb.dst)[q] = float4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w};
        }
        return;
    }
    if ((size_t)q * 4 >= (size_t)jb.n * jb.k) return;
    const int r = q & 15, a = (q >> 4) & 3, kb = jb.k >> 4;
    const int j = (q >> 6) % kb, i = (q >> 6) / kb;
    const int x = 16 * i + r, y = 16 * j + 4 * a;
    float4 v;
    if (jb.kind == 0) {
        v = *reinterpret_cast<const float4*>(jb.src + (size_t)x * jb.ld + y);
    } else {
        v.x = jb.src[(size_t)(y + 0) * jb.ld + x];
        v.y = jb.src[(size_t)(y + 1) * jb.ld + x];
        v.z = jb.src[(size_t)(y + 2) * jb.ld + x];
        v.w = jb.src[(size_t)(y + 3) * jb.ld + x];
    }
    reinterpret_cast<float4*>(jb.dst)[q] = v;
}


// ---- per-token projection table ---------------------------------------------------------------------------
// table[v][n] = bias[n] + sum_k emb[v][k] W[n][k]   (V <= 128 vocabulary rows, K = embedding size, N = 4H): the
// input projection of an embedding layer's V rows, from which the recurrent kernels take their step inputs.  The
// three GEMMs of it (forward; backward: d emb = d table W, d W = d table^T emb, d bias) have V x N x K = 26 MFLOP
// apiece -- as library calls they cost the host ~30 us each, fifteen per training step.  Here: one launch forward,
// one backward, fp32 MFMA 16x16x4 with both operands read straight from the row-major matrices (lane (li, g)
// holds row li, columns 16 kb + 4 g .. + 3: the four MFMAs of a k-block consume them in turn).
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma4(const f32x4 a, const f32x4 b, f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
    return acc;
}

// grid N / 16: workgroup -> 16 table columns, wave w -> vocabulary tiles w, w + 4
__global__ __launch_bounds__(256) void token_table_fwd_kernel(const float* __restrict__ emb, const float* __restrict__ w, long ldw,
                                                              const float* __restrict__ bias, int V, int K, int N,
                                                              float* __restrict__ table) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, g = lane >> 4;
    const int n0 = 16 * blockIdx.x;
    const float* wrow = w + (size_t)(n0 + li) * ldw + 4 * g;
    const float b = bias ? bias[n0 + li] : 0.f;
    for (int mt = wave; 16 * mt < V; mt += 4) {
        const float* arow = emb + (size_t)min(16 * mt + li, V - 1) * K + 4 * g;
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kb = 0; kb < K / 16; ++kb)
            acc = mfma4(*reinterpret_cast<const f32x4*>(arow + 16 * kb), *reinterpret_cast<const f32x4*>(wrow + 16 * kb), acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int v = 16 * mt + 4 * g + r;
            if (v < V) table[(size_t)v * N + n0 + li] = acc[r] + b;
        }
    }
}

// blocks [0, N / 16): d W rows n0 .. n0 + 15 (all K columns) and d bias of them -- contraction over the V rows;
// blocks behind: one 16 x 16 tile of d emb each -- contraction over the N columns, split over the four waves.
__global__ __launch_bounds__(256) void token_table_bwd_kernel(const float* __restrict__ dtable, const float* __restrict__ emb,
                                                              const float* __restrict__ w, long ldw, int V, int K, int N,
                                                              int padding_idx, float* __restrict__ demb,
                                                              float* __restrict__ dw, long lddw, float* __restrict__ dbias,
                                                              float* __restrict__ dbias2) {
    __shared__ float lds[128 * 17];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int V16 = (V + 15) & ~15;
    if ((int)blockIdx.x < N / 16) {
        const int n0 = 16 * blockIdx.x;
        for (int i = tid; i < V16 * 16; i += 256) {
            const int v = i >> 4, c = i & 15;
            lds[v * 17 + c] = v < V ? dtable[(size_t)v * N + n0 + c] : 0.f;
        }
        __syncthreads();
        if (dbias && tid < 16) {
            float sum = 0.f;
            for (int v = 0; v < V; ++v) sum += lds[v * 17 + tid];
            dbias[n0 + tid] = sum;
            if (dbias2) dbias2[n0 + tid] = sum;
        }
        if (!dw) return;
        for (int kt = wave; kt < K / 16; kt += 4) {
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int vb = 0; vb < V16 / 16; ++vb) {
                f32x4 a, b;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int v = 16 * vb + 4 * g + j;
                    a[j] = lds[v * 17 + li];
                    b[j] = v < V ? emb[(size_t)v * K + 16 * kt + li] : 0.f;
                }
                acc = mfma4(a, b, acc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) dw[(size_t)(n0 + 4 * g + r) * lddw + 16 * kt + li] = acc[r];
        }
        return;
    }
    if (!demb) return;
    const int tile = blockIdx.x - N / 16, mt = tile / (K / 16), kt = tile % (K / 16);
    const float* arow = dtable + (size_t)min(16 * mt + li, V - 1) * N + 4 * g;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    const int per = N / 4;  // columns of this wave (N % 64 == 0)
    for (int n = wave * per; n < (wave + 1) * per; n += 16) {
        f32x4 b;
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = w[(size_t)(n + 4 * g + j) * ldw + 16 * kt + li];
        acc = mfma4(*reinterpret_cast<const f32x4*>(arow + n), b, acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) lds[(wave * 16 + 4 * g + r) * 17 + li] = acc[r];
    __syncthreads();
    {
        const int r = tid >> 4, c = tid & 15, v = 16 * mt + r;
        if (v < V) {
            const float sum = (lds[r * 17 + c] + lds[(16 + r) * 17 + c]) + (lds[(32 + r) * 17 + c] + lds[(48 + r) * 17 + c]);
            demb[(size_t)v * K + 16 * kt + c] = v == padding_idx ? 0.f : sum;
        }
    }
}

}  // namespace

extern "C" int pnmn_token_prep(const int64_t* tokens, int64_t token_row_stride, int B, int T, int pad, int bos, int eos,
                               int drop_first, int64_t* out, float* fmask, int* last, void* stream) {
    if (B <= 0) return 0;
    if (!tokens || !out || T < 0) return PNMN_EINVAL;
    hipLaunchKernelGGL(token_prep_kernel, dim3((B + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), tokens,
                       token_row_stride, B, T, pad, bos, eos, drop_first, out, fmask, last);
    return (int)hipGetLastError();
}

extern "C" int pnmn_token_rows(const pnmn_token_seg* segs, int n_segs, int64_t* dst, int W, int64_t pad, void* stream) {
    if (n_segs <= 0 || W <= 0) return 0;
    if (!segs || !dst || n_segs > PNMN_TOKEN_SEGS) return PNMN_EINVAL;
    TokenSegs t;
    int total = 0;
    for (int k = 0; k < n_segs; ++k) {
        if (segs[k].rows < 0 || (segs[k].rows > 0 && !segs[k].src) || segs[k].width < 0) return PNMN_EINVAL;
        t.s[k] = segs[k];
        total += segs[k].rows;
    }
    t.n = n_segs;
    if (total == 0) return 0;
    hipLaunchKernelGGL(token_rows_kernel, dim3((total + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), t, dst, W, pad,
                       total);
    return (int)hipGetLastError();
}

extern "C" int pnmn_trim_predictions(const int64_t* raw, int B, int T, int end, int64_t* out, void* stream) {
    if (B <= 0 || T <= 0) return 0;
    if (!raw || !out) return PNMN_EINVAL;
    hipLaunchKernelGGL(trim_predictions_kernel, dim3((B + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), raw, B, T,
                       end, out);
    return (int)hipGetLastError();
}

extern "C" int pnmn_mask_last_fwd(const float* hs, const float* fmask, const int* last, int B, int T, int H, float* enc,
                                  float* hlast, void* stream) {
    if (B <= 0 || T <= 0) return 0;
    if (!hs || !fmask || !last || !enc || !hlast || (H & 3)) return PNMN_EINVAL;
    const int per_row = (T * (H >> 2) + 255) / 256;
    hipLaunchKernelGGL(mask_last_fwd_kernel, dim3(B, per_row < 8 ? per_row : 8), dim3(256), 0,
                       static_cast<hipStream_t>(stream), hs, fmask, last, T, H, enc, hlast);
    return (int)hipGetLastError();
}

extern "C" int pnmn_mask_last_bwd(const float* denc, const float* dhlast, const float* fmask, const int* last, int B, int T,
                                  int H, float* dhs, void* stream) {
    if (B <= 0 || T <= 0) return 0;
    if (!fmask || !last || !dhs || (H & 3)) return PNMN_EINVAL;
    const int per_row = (T * (H >> 2) + 255) / 256;
    hipLaunchKernelGGL(mask_last_bwd_kernel, dim3(B, per_row < 8 ? per_row : 8), dim3(256), 0,
                       static_cast<hipStream_t>(stream), denc, dhlast, fmask, last, T, H, dhs);
    return (int)hipGetLastError();
}

namespace {
constexpr int EG_SLICES = 8;
inline int eg_max_per(long rows, int V) { return (int)(((rows + EG_SLICES - 1) / EG_SLICES) / EG_CHUNK) + V; }
}  // namespace

extern "C" int64_t pnmn_embedding_grad_workspace_bytes(int B, int T, int V) {
    if (B <= 0 || T <= 0 || V <= 0) return 0;
    const long rows = (long)B * T;
    return (int64_t)sizeof(int) * (rows + (long)EG_SLICES * (3 * eg_max_per(rows, V) + 1));
}

extern "C" int pnmn_embedding_grad(const float* dy, const int64_t* tokens, int64_t token_row_stride, int B, int T, int C,
                                   int V, int shift, int start, int skip, int accumulate, float* dw, void* workspace,
                                   void* stream) {
    if (V <= 0 || C <= 0) return 0;
    if (!dw || (C & 3) || V > 128) return PNMN_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (B <= 0 || T <= 0) return accumulate ? 0 : (int)hipMemsetAsync(dw, 0, sizeof(float) * (size_t)V * C, s);
    if (!dy || !tokens || !workspace) return PNMN_EINVAL;
    const long rows = (long)B * T;
    if (rows * T >= (1L << 32)) return PNMN_ESHAPE;
    const unsigned magic = T == 1 ? 0u : (unsigned)(((1ULL << 32) + T - 1) / T);
    const int max_per = eg_max_per(rows, V);
    int* order = static_cast<int*>(workspace);
    int* chunks = order + rows;
    int* count = chunks + 3 * (size_t)EG_SLICES * max_per;
    hipLaunchKernelGGL(embedding_rows_kernel, dim3(EG_SLICES), dim3(1024), 0, s, tokens, token_row_stride, B, T, magic, V, shift,
                       start, skip, C, order, chunks, max_per, count, dw, accumulate);
    hipLaunchKernelGGL(embedding_sum_kernel, dim3(EG_SLICES * max_per, (C + 1023) / 1024), dim3(256), 0, s, dy, order, chunks,
                       max_per, count, C, dw);
    return (int)hipGetLastError();
}

extern "C" int pnmn_derive_params(const pnmn_derive_job* jobs, int n_jobs, int max_quads, void* stream) {
    if (n_jobs <= 0) return 0;
    if (!jobs || max_quads <= 0) return PNMN_EINVAL;
    hipLaunchKernelGGL(derive_params_kernel, dim3((max_quads + 255) / 256, n_jobs), dim3(256), 0,
                       static_cast<hipStream_t>(stream), jobs);
    return (int)hipGetLastError();
}

extern "C" int pnmn_token_table_fwd(const float* emb, const float* weight, int64_t weight_row_stride, const float* bias, int V,
                                    int K, int N, float* table, void* stream) {
    if (V <= 0) return 0;
    if (!emb || !weight || !table) return PNMN_EINVAL;
    if (V > 128 || K < 16 || K % 16 || N < 64 || N % 64 || weight_row_stride % 4) return PNMN_ESHAPE;
    hipLaunchKernelGGL(token_table_fwd_kernel, dim3(N / 16), dim3(256), 0, static_cast<hipStream_t>(stream), emb, weight,
                       (long)weight_row_stride, bias, V, K, N, table);
    return (int)hipGetLastError();
}

extern "C" int pnmn_token_table_bwd(const float* dtable, const float* emb, const float* weight, int64_t weight_row_stride, int V,
                                    int K, int N, int padding_idx, float* demb, float* dweight, int64_t dweight_row_stride,
                                    float* dbias, float* dbias2, void* stream) {
    if (V <= 0) return 0;
    if (!dtable || !emb || !weight) return PNMN_EINVAL;
    if (V > 128 || K < 16 || K % 16 || N < 64 || N % 64) return PNMN_ESHAPE;
    const int blocks = N / 16 + (demb ? ((V + 15) / 16) * (K / 16) : 0);
    hipLaunchKernelGGL(token_table_bwd_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), dtable, emb, weight,
                       (long)weight_row_stride, V, K, N, padding_idx, demb, dweight,
                       (long)(dweight_row_stride > 0 ? dweight_row_stride : K), dbias, dbias2);
    return (int)hipGetLastError();
}

// -----------------------------------------------------------------------------------------------------
// Encoder-output gradient of the attention decoder:  denc[b][s][:] = sum_t  w[b][t][s] dctx[b][t][:]
//                                                                       +  dscore[b][t][s] h_{t-1}[b][:]
// (enc_s enters step t through the context, weight w_ts, and through the score, whose gradient multiplies the
// previous hidden state).  The multi-CU backward kernel emits dctx / dscore / w; this used to be two strided-batched
// library GEMMs per decoder -- B products of [S x T] . [T x 256] with S, T <= 64: 0.46 ms each at 1024 rows
// (1.4 TFLOP/s) and 0.07-0.23 ms at 64-128 rows, on the backward pass's critical chain.  It is a bandwidth problem
// (dctx and hs are read once: 2 x B T H floats): one workgroup per (row, 16 source positions); thread c owns
// hidden unit c, walks the T steps with coalesced 1 KB row loads and keeps 16 accumulators; the per-(t, s)
// coefficients come from LDS.
// -----------------------------------------------------------------------------------------------------
namespace {
constexpr int DENC_H = 256, DENC_SB = 16, DENC_MAXT = 64;
__global__ __launch_bounds__(256) void attn_denc_kernel(const float* __restrict__ weights, const float* __restrict__ dscore,
                                                        const float* __restrict__ dctx, const float* __restrict__ hs,
                                                        const float* __restrict__ h0, float* __restrict__ denc, int T, int S) {
    __shared__ float wl[DENC_MAXT][DENC_SB], dl[DENC_MAXT][DENC_SB];
    const int b = blockIdx.x, s0 = blockIdx.y * DENC_SB, c = threadIdx.x;
    const int ns = (S - s0) < DENC_SB ? (S - s0) : DENC_SB;
    for (int i = threadIdx.x; i < T * DENC_SB; i += 256) {
        const int t = i / DENC_SB, s = i % DENC_SB;
        const bool ok = s < ns;
        wl[t][s] = ok ? weights[((size_t)b * T + t) * S + s0 + s] : 0.f;
        dl[t][s] = ok ? dscore[((size_t)b * T + t) * S + s0 + s] : 0.f;
    }
    __syncthreads();
    float acc[DENC_SB];
#pragma unroll
    for (int s = 0; s < DENC_SB; ++s) acc[s] = 0.f;
    const float* dc = dctx + (size_t)b * T * DENC_H + c;
    const float* hp = hs + (size_t)b * T * DENC_H + c;
    float hprev = h0[(size_t)b * DENC_H + c];
    for (int t = 0; t < T; ++t) {
        const float d = dc[(size_t)t * DENC_H];
        const float hnext = (t + 1 < T) ? hp[(size_t)t * DENC_H] : 0.f;  // h_t: the next step's h_{t-1}
#pragma unroll
        for (int s = 0; s < DENC_SB; ++s) acc[s] += wl[t][s] * d + dl[t][s] * hprev;
        hprev = hnext;
    }
    for (int s = 0; s < ns; ++s) denc[((size_t)b * S + s0 + s) * DENC_H + c] = acc[s];
}
}  // namespace

extern "C" int pnmn_attn_denc(const float* weights, const float* dscore, const float* dctx, const float* hs, const float* h0,
                              float* denc, int B, int T, int S, int hidden, void* stream) {
    if (B <= 0 || S <= 0) return 0;
    if (!weights || !dscore || !dctx || !hs || !h0 || !denc) return PNMN_EINVAL;
    if (hidden != DENC_H || T < 1 || T > DENC_MAXT) return PNMN_ESHAPE;
    hipLaunchKernelGGL(attn_denc_kernel, dim3(B, (S + DENC_SB - 1) / DENC_SB), dim3(256), 0, static_cast<hipStream_t>(stream),
                       weights, dscore, dctx, hs, h0, denc, T, S);
    return (int)hipGetLastError();
}
