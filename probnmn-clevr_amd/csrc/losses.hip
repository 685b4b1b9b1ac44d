// Per-sequence and per-batch loss kernels for gfx950: what the reference computes with a dozen tiny torch
// ops per call (log_softmax, gather, mask, sum, divide, ...; seq2seq_base.py:235-254,334-341,
// program_prior.py:146-151, elbo.py:28-34,61-89,253-270) as one launch each -- at 128 questions per GPU the
// step is bound by the number of launches, not by their work.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/probnmn_hip.h"
#include "sampling.h"

namespace {

__device__ __forceinline__ float wave_max(float v) { return pnmn::wmax(v); }  // (DPP reductions, sampling.h)
__device__ __forceinline__ float wave_sum(float v) { return pnmn::wsum(v); }

// log-sum-exp of one row of V logits, computed by one wave (every lane gets the result)
__device__ __forceinline__ float row_lse(const float* __restrict__ z, int V, int lane) {
    float m = -INFINITY;
    for (int k = lane; k < V; k += 64) m = fmaxf(m, z[k]);
    m = wave_max(m);
    float s = 0.f;
    for (int k = lane; k < V; k += 64) s += expf(z[k] - m);
    return m + logf(wave_sum(s));
}

// one workgroup (4 waves) per sequence; wave w takes steps w, w+4, ...
__global__ __launch_bounds__(256) void seq_nll_fwd_kernel(const float* __restrict__ logits, int64_t logits_bstride,
                                                          const int64_t* __restrict__ tokens, int64_t tok_bstride,
                                                          const int64_t* __restrict__ mask_tokens, int64_t mask_bstride,
                                                          int pad, float* __restrict__ loss, float* __restrict__ lse_out,
                                                          int T, int V, float eps) {
    __shared__ float red[2][4];
    const int b = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float num = 0.f, den = 0.f;
    for (int t = wave; t < T; t += 4) {
        const float* z = logits + (size_t)b * logits_bstride + (size_t)t * V;
        const float lse = row_lse(z, V, lane);
        if (lane == 0) lse_out[(size_t)b * T + t] = lse;
        const float w = mask_tokens[(size_t)b * mask_bstride + t] != pad ? 1.f : 0.f;
        const int64_t tok = tokens[(size_t)b * tok_bstride + t];
        num += w * (lse - z[tok]);
        den += w;
    }
    if (lane == 0) {
        red[0][wave] = num;
        red[1][wave] = den;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float n = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        const float d = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        loss[b] = n / (d + eps);
    }
}

__global__ __launch_bounds__(256) void seq_nll_bwd_kernel(const float* __restrict__ logits, int64_t logits_bstride,
                                                          const int64_t* __restrict__ tokens, int64_t tok_bstride,
                                                          const int64_t* __restrict__ mask_tokens, int64_t mask_bstride,
                                                          int pad, const float* __restrict__ lse,
                                                          const float* __restrict__ dloss, float* __restrict__ dlogits,
                                                          int64_t dlogits_bstride, int T, int V, float eps) {
    __shared__ float den_s;
    const int b = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave == 0) {
        float d = 0.f;
        for (int t = lane; t < T; t += 64) d += mask_tokens[(size_t)b * mask_bstride + t] != pad ? 1.f : 0.f;
        d = wave_sum(d);
        if (lane == 0) den_s = d;
    }
    __syncthreads();
    const float g = dloss[b] / (den_s + eps);
    for (int t = wave; t < T; t += 4) {
        const float* z = logits + (size_t)b * logits_bstride + (size_t)t * V;
        float* dz = dlogits + (size_t)b * dlogits_bstride + (size_t)t * V;
        const float coef = mask_tokens[(size_t)b * mask_bstride + t] != pad ? g : 0.f;
        const float l = lse[(size_t)b * T + t];
        const int64_t tok = tokens[(size_t)b * tok_bstride + t];
        for (int k = lane; k < V; k += 64) dz[k] = coef * (expf(z[k] - l) - (k == tok ? 1.f : 0.f));
    }
}

// REINFORCE / ELBO combination over the sampled rows: one workgroup, any n
//   logq = -pg, rec = -qr, prior = -pr, ans = -nmn      (per-row negative log-likelihoods in)
//   R = rec + beta * prior - beta * logq + gamma * ans ;  c = R - baseline
//   kl = logq * c - beta * logq ;  elbo = rec - kl
// sums[0..5] = sum rec, sum kl, sum elbo, sum R, sum nmn, sum c ; per-row derivatives of sum(elbo):
//   d/d pg[n] = c[n] - beta ;  d/d qr[n] = -1   (R is a constant of the estimator)
__global__ __launch_bounds__(256) void elbo_rows_kernel(const float* __restrict__ pg, const float* __restrict__ qr,
                                                        const float* __restrict__ pr, const float* __restrict__ nmn,
                                                        const float* __restrict__ baseline, float beta, float gamma,
                                                        int n, float* __restrict__ sums, float* __restrict__ dpg) {
    __shared__ float red[6][4];
    const float b = baseline[0];
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < n; i += 256) {
        const float logq = -pg[i], rec = -qr[i];
        const float prior = pr ? -pr[i] : 0.f;
        const float ans = nmn ? -nmn[i] : 0.f;
        const float R = rec + beta * prior - beta * logq + gamma * ans;
        const float c = R - b;
        const float kl = logq * c - beta * logq;
        acc[0] += rec;
        acc[1] += kl;
        acc[2] += rec - kl;
        acc[3] += R;
        acc[4] += nmn ? nmn[i] : 0.f;
        acc[5] += c;
        dpg[i] = c - beta;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const float s = wave_sum(acc[k]);
        if (lane == 0) red[k][wave] = s;
    }
    __syncthreads();
    if (threadIdx.x < 6) sums[threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// The whole scalar end of a question-coding / joint-training iteration in one launch (one workgroup): the ELBO
// combination above over the n sampled rows, the means of the m supervised rows' cross entropies, the objective
//     J = w_u (gamma mean(nmn) - mean(elbo)) + w_s alpha (mean(pg_s) + mean(qr_s))
// its per-row derivatives, and (single process) the moving-baseline update b += decay mean(c).  qr holds the n
// sampled rows followed by the m supervised ones (the reconstructor ran once over both).
//   stats[10] = mean rec, mean kl, mean elbo, mean R, mean nmn, sum c, mean pg_s, mean qr_s, J, n
//   grads: d_pg[n] = -w_u (c - beta) / n ; d_qr[0..n) = w_u / n, d_qr[n..n+m) = w_s alpha / m ;
//          d_nmn[n] = w_u gamma / n ; d_pgs[m] = w_s alpha / m
__global__ __launch_bounds__(256) void joint_objective_kernel(
    const float* __restrict__ pg, const float* __restrict__ qr, const float* __restrict__ pr, const float* __restrict__ nmn,
    const float* __restrict__ pgs, float* baseline, const float* __restrict__ w_u_ptr, const float* __restrict__ w_s_ptr,
    float alpha, float beta, float gamma, float decay, int update_baseline, int n, int m, float* __restrict__ stats,
    float* __restrict__ objective, float* __restrict__ d_pg, float* __restrict__ d_qr, float* __restrict__ d_nmn, float* __restrict__ d_pgs) {
    __shared__ float red[8][4];
    const float b = baseline[0];
    const float w_u = w_u_ptr ? w_u_ptr[0] : 1.f, w_s = w_s_ptr ? w_s_ptr[0] : 1.f;
    const float inv_n = n > 0 ? 1.f / (float)n : 0.f, inv_m = m > 0 ? 1.f / (float)m : 0.f;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < n; i += 256) {
        const float logq = -pg[i], rec = -qr[i];
        const float prior = pr ? -pr[i] : 0.f;
        const float ans = nmn ? -nmn[i] : 0.f;
        const float R = rec + beta * prior - beta * logq + gamma * ans;
        const float c = R - b;
        const float kl = logq * c - beta * logq;
        acc[0] += rec;
        acc[1] += kl;
        acc[2] += rec - kl;
        acc[3] += R;
        acc[4] += nmn ? nmn[i] : 0.f;
        acc[5] += c;
        d_pg[i] = -w_u * (c - beta) * inv_n;
        d_qr[i] = w_u * inv_n;
        if (d_nmn) d_nmn[i] = w_u * gamma * inv_n;
    }
    for (int j = threadIdx.x; j < m; j += 256) {
        acc[6] += pgs[j];
        acc[7] += qr[n + j];
        d_pgs[j] = w_s * alpha * inv_m;
        d_qr[n + j] = w_s * alpha * inv_m;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float s = wave_sum(acc[k]);
        if (lane == 0) red[k][wave] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = (red[k][0] + red[k][1]) + (red[k][2] + red[k][3]);
        stats[0] = t[0] * inv_n, stats[1] = t[1] * inv_n, stats[2] = t[2] * inv_n, stats[3] = t[3] * inv_n;
        stats[4] = t[4] * inv_n, stats[5] = t[5];
        stats[6] = t[6] * inv_m, stats[7] = t[7] * inv_m;
        stats[8] = w_u * (gamma * t[4] * inv_n - t[2] * inv_n) + w_s * alpha * (t[6] * inv_m + t[7] * inv_m);
        stats[9] = (float)n;
        objective[0] = stats[8];
        if (update_baseline && n > 0) baseline[0] = b + decay * t[5] * inv_n;
    }
}

}  // namespace

extern "C" {

int pnmn_joint_objective(const float* pg, const float* qr, const float* prior, const float* nmn, const float* pg_sup,
                         float* baseline, const float* w_unsup, const float* w_sup, float alpha, float beta, float gamma,
                         float decay, int update_baseline, int n, int m, float* stats, float* objective, float* d_pg,
                         float* d_qr, float* d_nmn, float* d_pg_sup, void* stream) {
    if (n < 0 || m < 0 || !baseline || !stats || !objective) return PNMN_EINVAL;
    if (n > 0 && (!pg || !qr || !d_pg || !d_qr)) return PNMN_EINVAL;
    if (m > 0 && (!pg_sup || !qr || !d_pg_sup || !d_qr)) return PNMN_EINVAL;
    if ((nmn != nullptr) != (d_nmn != nullptr)) return PNMN_EINVAL;
    hipLaunchKernelGGL(joint_objective_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), pg, qr, prior, nmn,
                       pg_sup, baseline, w_unsup, w_sup, alpha, beta, gamma, decay, update_baseline, n, m, stats, objective, d_pg,
                       d_qr, d_nmn, d_pg_sup);
    return (int)hipGetLastError();
}

int pnmn_seq_nll_fwd(const float* logits, int64_t logits_bstride, const int64_t* tokens, int64_t tok_bstride,
                     const int64_t* mask_tokens, int64_t mask_bstride, int pad, float* loss, float* lse, int B, int T,
                     int V, float eps, void* stream) {
    if (B <= 0) return 0;
    if (!logits || !tokens || !mask_tokens || !loss || !lse || T <= 0 || V <= 0) return PNMN_EINVAL;
    hipLaunchKernelGGL(seq_nll_fwd_kernel, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream), logits,
                       logits_bstride, tokens, tok_bstride, mask_tokens, mask_bstride, pad, loss, lse, T, V, eps);
    return (int)hipGetLastError();
}

int pnmn_seq_nll_bwd(const float* logits, int64_t logits_bstride, const int64_t* tokens, int64_t tok_bstride,
                     const int64_t* mask_tokens, int64_t mask_bstride, int pad, const float* lse, const float* dloss,
                     float* dlogits, int64_t dlogits_bstride, int B, int T, int V, float eps, void* stream) {
    if (B <= 0) return 0;
    if (!logits || !tokens || !mask_tokens || !lse || !dloss || !dlogits || T <= 0 || V <= 0) return PNMN_EINVAL;
    hipLaunchKernelGGL(seq_nll_bwd_kernel, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream), logits,
                       logits_bstride, tokens, tok_bstride, mask_tokens, mask_bstride, pad, lse, dloss, dlogits,
                       dlogits_bstride, T, V, eps);
    return (int)hipGetLastError();
}

int pnmn_elbo_rows(const float* pg_loss, const float* qr_loss, const float* prior_loss, const float* nmn_loss,
                   const float* baseline, float beta, float gamma, int n, float* sums, float* dpg, void* stream) {
    if (n <= 0) return 0;
    if (!pg_loss || !qr_loss || !baseline || !sums || !dpg) return PNMN_EINVAL;
    hipLaunchKernelGGL(elbo_rows_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), pg_loss, qr_loss,
                       prior_loss, nmn_loss, baseline, beta, gamma, n, sums, dpg);
    return (int)hipGetLastError();
}

}  // extern "C"
