// Glue kernels of the seq2seq passes for gfx950: the token bookkeeping, masking, state selection and
// embedding gradients that the reference does with chains of tiny tensor ops (allennlp's
// add_sentence_boundary_token_ids / get_text_field_mask / get_final_encoder_states, reference
// probnmn/modules/seq2seq_base.py:97-141,278-293) -- one launch each.  At 128 questions per GPU the step is
// bound by the NUMBER of launches (host time and ~5 us of GPU time apiece), not by their work.
#include <hip/hip_runtime.h>
#include <stdint.h>

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
//   dw[v][c] += sum over rows r with tokens[r] == v of dy[r][c]          (dw zeroed by the caller, V <= 128)
// grid (C / (64 VEC), row splits); a workgroup sums its rows into an LDS table [V][64 VEC] (ds_add_f32), then
// adds the table into dw.  A wave reads 256 VEC contiguous bytes of one row; the four waves take different rows,
// eight rows in flight each (the loop is bound by the latency of its loads, not by their bytes).  `shift` = 1:
// row (b, t) takes the token of (b, t - 1) and `start` for t = 0 (the input token of decoding step t),
// tokens being [B][T].
template <int VEC>
__global__ __launch_bounds__(256) void embedding_grad_kernel(const float* __restrict__ dy, const int64_t* __restrict__ tokens,
                                                             int64_t tok_bstride, int B, int T, int C, int V, int shift,
                                                             int start, int skip, float* __restrict__ dw) {
    extern __shared__ float table[];  // [V][64 * VEC]
    constexpr int W = 64 * VEC;
    constexpr int UN = 8;
    for (int i = threadIdx.x; i < V * W; i += 256) table[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const int c = blockIdx.x * W + lane * VEC;
    const int R = B * T;
    const int per = (R + gridDim.y - 1) / gridDim.y;
    const int r0 = blockIdx.y * per, r1 = min(R, r0 + per);
    for (int rb = r0 + sub; rb < r1; rb += 4 * UN) {
        float val[UN][VEC];
        int tok[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int r = rb + 4 * u;
            tok[u] = -1;
            if (r < r1) {
                const int b = r / T, t = r - b * T;
                int64_t v;
                if (shift)
                    v = t == 0 ? start : tokens[(size_t)b * tok_bstride + t - 1];
                else
                    v = tokens[(size_t)b * tok_bstride + t];
                tok[u] = (v == skip || v < 0 || v >= V) ? -1 : (int)v;
                const float* src = dy + (size_t)r * C + c;
                if (VEC == 4) {
                    const float4 q = *reinterpret_cast<const float4*>(src);
                    val[u][0] = q.x, val[u][1 % VEC] = q.y, val[u][2 % VEC] = q.z, val[u][3 % VEC] = q.w;
                } else if (VEC == 2) {
                    const float2 q = *reinterpret_cast<const float2*>(src);
                    val[u][0] = q.x, val[u][1 % VEC] = q.y;
                } else {
                    val[u][0] = *src;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (tok[u] < 0) continue;
#pragma unroll
            for (int k = 0; k < VEC; ++k)
                if (val[u][k] != 0.f) unsafeAtomicAdd(&table[tok[u] * W + lane * VEC + k], val[u][k]);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < V * W; i += 256) {
        const float s = table[i];
        if (s != 0.f) unsafeAtomicAdd(dw + (size_t)(i / W) * C + blockIdx.x * W + (i % W), s);
    }
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
                                                              float* __restrict__ dw, float* __restrict__ dbias) {
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
            for (int r = 0; r < 4; ++r) dw[(size_t)(n0 + 4 * g + r) * K + 16 * kt + li] = acc[r];
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

extern "C" int pnmn_embedding_grad(const float* dy, const int64_t* tokens, int64_t token_row_stride, int B, int T, int C,
                                   int V, int shift, int start, int skip, float* dw, void* stream) {
    if (B <= 0 || T <= 0) return 0;
    if (!dy || !tokens || !dw || (C & 63) || V < 1 || V > 128) return PNMN_EINVAL;
    const long rows = (long)B * T;
    // widest row segment per wave (1 KiB; the table then takes up to 128 KiB of the CU's LDS: one workgroup per
    // CU, eight loads in flight per lane)
    const int vec = (C % 256 == 0) ? 4 : (C % 128 == 0) ? 2 : 1;
    const int blocks = C / (64 * vec);
    int splits = (int)((rows + 127) / 128);  // >= 128 rows per workgroup
    const int cap = 512 / blocks > 1 ? 512 / blocks : 1;
    if (splits > cap) splits = cap;
    if (splits < 1) splits = 1;
    const size_t lds = (size_t)V * 64 * vec * sizeof(float);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid(blocks, splits);
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(embedding_grad_kernel<4>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 256 * 4);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    if (vec == 4)
        hipLaunchKernelGGL(embedding_grad_kernel<4>, grid, dim3(256), lds, s, dy, tokens, token_row_stride, B, T, C, V, shift,
                           start, skip, dw);
    else if (vec == 2)
        hipLaunchKernelGGL(embedding_grad_kernel<2>, grid, dim3(256), lds, s, dy, tokens, token_row_stride, B, T, C, V, shift,
                           start, skip, dw);
    else
        hipLaunchKernelGGL(embedding_grad_kernel<1>, grid, dim3(256), lds, s, dy, tokens, token_row_stride, B, T, C, V, shift,
                           start, skip, dw);
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
                                    int K, int N, int padding_idx, float* demb, float* dweight, float* dbias, void* stream) {
    if (V <= 0) return 0;
    if (!dtable || !emb || !weight) return PNMN_EINVAL;
    if (V > 128 || K < 16 || K % 16 || N < 64 || N % 64) return PNMN_ESHAPE;
    const int blocks = N / 16 + (demb ? ((V + 15) / 16) * (K / 16) : 0);
    hipLaunchKernelGGL(token_table_bwd_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), dtable, emb, weight,
                       (long)weight_row_stride, V, K, N, padding_idx, demb, dweight, dbias);
    return (int)hipGetLastError();
}
