// Wave-level helpers shared by the decoder kernels: reductions, the Philox4x32-10 uniform that
// pnmn_sample_tokens also uses, and one row's token choice from a row of logits in LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace pnmn {

__device__ __forceinline__ float sigm(float z) { return 1.f / (1.f + expf(-z)); }
// Wave-wide sum / max on the VALU's data-parallel primitives (DPP): four steps inside each row of 16 lanes
// (quad swaps, half-row and row mirrors), two row broadcasts, one v_readlane -- every lane gets the result.  The
// __shfl_xor butterfly these replace compiles to six ds_bpermute_b32, i.e. six dependent trips through the LDS
// crossbar (~60 cycles each) per reduction; the decoder kernels do ~25 reductions per time step.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move(float identity, float v) {  // lanes without a source keep `identity`
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, identity), __builtin_bit_cast(int, v),
                                                                  CTRL, ROW_MASK, 0xF, false));
}
template <typename Op>
__device__ __forceinline__ float wave_reduce(float v, float identity, Op op) {
    v = op(v, dpp_move<0xB1, 0xF>(identity, v));   // quad_perm [1,0,3,2]
    v = op(v, dpp_move<0x4E, 0xF>(identity, v));   // quad_perm [2,3,0,1]
    v = op(v, dpp_move<0x141, 0xF>(identity, v));  // row_half_mirror
    v = op(v, dpp_move<0x140, 0xF>(identity, v));  // row_mirror: every lane = its row of 16
    v = op(v, dpp_move<0x142, 0xA>(identity, v));  // row_bcast:15 into rows 1, 3
    v = op(v, dpp_move<0x143, 0xC>(identity, v));  // row_bcast:31 into rows 2, 3: lane 63 = the wave
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wsum(float v) {
    return wave_reduce(v, 0.f, [](float a, float b) { return a + b; });
}
__device__ __forceinline__ float wmax(float v) {
    return wave_reduce(v, -INFINITY, [](float a, float b) { return fmaxf(a, b); });
}

__device__ __forceinline__ void philox_round(uint32_t (&ctr)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * ctr[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * ctr[2];
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    ctr[0] = hi1 ^ ctr[1] ^ k0;
    ctr[1] = lo1;
    ctr[2] = hi0 ^ ctr[3] ^ k1;
    ctr[3] = lo0;
}
__device__ inline float philox_uniform(uint64_t seed, uint64_t row, uint32_t step) {  // same stream as pnmn_sample_tokens
    uint32_t ctr[4] = {(uint32_t)row, (uint32_t)(row >> 32), step, 0x9E3779B9u};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(ctr, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return (float)(ctr[0] >> 8) * (1.0f / 16777216.0f);
}

// One wave picks one row's token from `logits` (LDS, V <= 128 entries): mode 2 = first arg-max,
// mode 1 = inverse-CDF draw from softmax(logits) with pad / unk / start removed (reference
// seq2seq_base.py:203-220).  The result is wave-uniform.
__device__ inline int choose_row_token(const float* logits, int V, int mode, int pad, int unk, int start, uint64_t seed,
                                       uint64_t global_row, uint32_t t) {
    const int lane = threadIdx.x & 63;
    float v[2];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int j = lane + 64 * k;
        v[k] = j < V ? logits[j] : -INFINITY;
        mx = fmaxf(mx, v[k]);
    }
    mx = wmax(mx);
    int choice;
    if (mode == 2) {
        int best = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (lane + 64 * k < V && v[k] == mx && lane + 64 * k < best) best = lane + 64 * k;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int other = __shfl_xor(best, o);
            best = other < best ? other : best;
        }
        choice = best;
    } else {
        float se = 0.f;
#pragma unroll
        for (int k = 0; k < 2; ++k) se += (lane + 64 * k < V) ? expf(v[k] - mx) : 0.f;
        const float lse = mx + logf(wsum(se));
        float w[2], tot = 0.f;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int j = lane + 64 * k;
            const bool ok = j < V && j != pad && j != unk && j != start;
            w[k] = ok ? expf(v[k] - lse) : 0.f;
            tot += w[k];
        }
        tot = wsum(tot);
        const float target = philox_uniform(seed, global_row, t) * tot;
        float before = 0.f;
        choice = -1;
        int last_ok = -1;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            float inc = w[k];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const float tt = __shfl_up(inc, o);
                if (lane >= o) inc += tt;
            }
            const unsigned long long hit = __ballot((w[k] > 0.f) && (before + inc > target));
            if (choice < 0 && hit) choice = 64 * k + (int)__ffsll((long long)hit) - 1;
            const unsigned long long pos = __ballot(w[k] > 0.f);
            if (pos) last_ok = 64 * k + 63 - __clzll((long long)pos);
            before += __shfl(inc, 63);
        }
        if (choice < 0) choice = last_ok;
    }
    return choice;
}

}  // namespace pnmn
