// Several LSTM-layer passes in ONE launch, layers of one encoder as a WAVEFRONT (include/probnmn_hip.h: pnmn_lstm_stack_job).
//
// The reference's encoders are two-layer nn.LSTMs (probnmn/modules/seq2seq_base.py:56-60 via allennlp's
// PytorchSeq2SeqWrapper; probnmn/models/program_prior.py:50-56).  seq2seq.hip runs ONE layer over the whole sequence per
// launch (pnmn_lstm_seq_fwd / _bwd) with the second layer's input projection as a GEMM over all time steps in between:
// layer 1 (T steps) -> GEMM -> layer 2 (T steps), the same again backward.  But layer 2 at step t needs layer 1 at step
// t only, and the persistent kernels are bound by their per-step hand-off latency, not by the matrix pipe: at <= 256
// rows both layers' workgroups fit the chip side by side.  Here a launch takes a table of JOBS:
//   forward   FIRST  (dep < 0): a layer whose step inputs are rows of a per-token table / of a dense [B][T][4H] tensor --
//                               pnmn_lstm_seq_fwd's multi-CU kernel, which also tells its consumer "h_t is out";
//             SECOND (dep >= 0): the layer above job `dep`: per step  gates = b + h1_t W_ih^T + h2_{t-1} W_hh^T  with
//                               BOTH weight slices in registers (128 VGPRs); the product with h1_{t+1} is issued right
//                               behind the hand-off of step t, so it runs while the partners' h2_t travels;
//   backward  TOP    (dep < 0): pnmn_lstm_seq_bwd's multi-CU kernel (gradient wrt every output given) + "dgates_t is out";
//             BELOW  (dep >= 0): the layer under job `dep`: d h1_t = dgates2_t W_ih (the [16 x 1024] row block staged in
//                               LDS, K split over the eight waves) + its own recurrence.
// Independent jobs ride along (the two encoders' passes of a training iteration, the prior's): the launch takes as long as
// its longest chain.  Every job's tile t sits on XCD t % 8 (virtual tile = job base + t, bases multiples of 8), so linked
// tiles share an L2 and use the cheap same-XCD hand-off of cluster.h; linked tiles decide that TOGETHER (one shared word).
// Arithmetic: a FIRST / TOP job is bit-identical to pnmn_lstm_seq_{fwd,bwd}; a SECOND job's pre-activations are
// (b + h1 W_ih^T) + h2 W_hh^T summed in ascending k inside the MFMA chain where the separate path rounds the projection
// to memory first (same order of additions: bias first, then k ascending) -- and BELOW adds eight K-slice partials of
// dgates2 W_ih where the GEMM adds its own k tiles: equal to fp32 round-off (tests/test_lstm_stack_gpu.py).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/probnmn_hip.h"
#include "cluster.h"
#include "lds_optin.h"

namespace {

typedef float f32x4_ __attribute__((ext_vector_type(4)));
constexpr int LH = 256, LROWS = 16, S = 8, UW = LH / S;  // eight members per tile, 32 hidden units each
constexpr int MAXJ = PNMN_LSTM_STACK_JOBS;

__device__ __forceinline__ float sigm(float z) { return 1.f / (1.f + expf(-z)); }

struct Jobs {
    pnmn_lstm_stack_job j[MAXJ];
    int base[MAXJ + 1];   // first virtual tile of each job (multiples of 8)
    int consumer[MAXJ];   // job that depends on this one, or -1
    float* px[MAXJ];      // backward: the job's exchange buffer [tiles][2][S][16][H]
    int n;
};

// ---- hand-offs ----------------------------------------------------------------------------------------------------------
// line of a virtual tile (64 ints): [0] own arrivals  [1] started << 16 | XCC bits (unlinked jobs)  [2] finished
//                                   [3] cross arrivals: the producer's members count here (consumer's line)
//                                   [4] started << 16 | XCC bits of a LINKED pair (consumer's line, 2 S members)
struct Sync {
    int* line;        // this tile's line
    int* cross_out;   // consumer tile's word [3] (producer side) or nullptr
    int* cross_in;    // own word [3] (consumer side) or nullptr
    int own, cross;   // hand-offs waited for so far
    bool fast;

    __device__ __forceinline__ void start(int* mine, int* consumer_line, bool has_dep) {
        line = mine;
        cross_out = consumer_line ? consumer_line + 3 : nullptr;
        cross_in = has_dep ? mine + 3 : nullptr;
        own = cross = 0;
        __shared__ int decided;
        if (threadIdx.x == 0) {
            // a linked pair publishes into the consumer's word [4] (zeroed by the consumer's last member: it outlives
            // the producer by the dependency), an unlinked job into its own word [1]
            int* word = consumer_line ? consumer_line + 4 : (has_dep ? mine + 4 : mine + 1);
            const int expect = (consumer_line || has_dep) ? 2 * S : S;
            const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;  // HW_REG_XCC_ID[3:0]
            __hip_atomic_fetch_or(word, 1 << xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(word, 1 << 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0, w;
            while (((w = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 16) < expect) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 26)) __builtin_trap();
            }
            decided = __builtin_popcount(w & 0xFFFF) == 1;  // (the value that showed the full count carries every bit)
        }
        __syncthreads();
        fast = decided != 0;
    }

    __device__ __forceinline__ void publish(int* word) {
        if (fast)
            __hip_atomic_fetch_add(word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
            __hip_atomic_fetch_add(word, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }

    // this step's stores are out: tell the partners (`to_own`) and / or the consumer job
    __device__ __forceinline__ void signal(bool to_own) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            if (to_own) publish(line);
            if (cross_out) publish(cross_out);
        }
    }

    __device__ __forceinline__ void spin(int* word, int target) {
        if (threadIdx.x == 0) {
            int spins = 0;
            while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 26)) __builtin_trap();
            }
            if (fast)
                asm volatile("buffer_inv sc0\n\ts_waitcnt vmcnt(0)" ::: "memory");
            else
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
    __device__ __forceinline__ void wait_own() { spin(line, S * ++own); }
    __device__ __forceinline__ void wait_cross() { spin(cross_in, S * ++cross); }

    __device__ __forceinline__ void finish() {
        if (threadIdx.x == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int before = __hip_atomic_fetch_add(line + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (before == S - 1)
                for (int k = 0; k < 5; ++k) __hip_atomic_store(line + k, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
};

__device__ __forceinline__ f32x4_ mfma4(const f32x4_ a, const f32x4_ b, f32x4_ acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
    return acc;
}

// the tile's 16 x 256 vector of step `t` of a [B][T][256] tensor into LDS (two 16-byte pieces per thread, both requested
// before the first store: see lstm_seq_fwd_cluster_kernel)
constexpr int HLD = LH + 4;
__device__ __forceinline__ void stage_rows(const float* src, int row0, int B, int T, int t, float (*hl)[HLD]) {
    const int tid = threadIdx.x;
    f32x4_ piece[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i = tid + 512 * k, rl = i / (LH / 4), c4 = i % (LH / 4);
        const int row = min(row0 + rl, B - 1);
        piece[k] = *reinterpret_cast<const f32x4_*>(src + ((size_t)row * T + t) * LH + 4 * c4);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i = tid + 512 * k, rl = i / (LH / 4), c4 = i % (LH / 4);
        *reinterpret_cast<f32x4_*>(&hl[rl][4 * c4]) = piece[k];
    }
}

// =========================================================================================================================
// forward
// =========================================================================================================================
__global__ __launch_bounds__(512) void lstm_stack_fwd_kernel(const Jobs jobs, int* sync) {
    constexpr int GLD = UW + 4;
    __shared__ float gl[4][LROWS][GLD];
    __shared__ __attribute__((aligned(16))) float hl[LROWS][HLD];   // h_{t-1} of this layer
    __shared__ __attribute__((aligned(16))) float xl[LROWS][HLD];   // SECOND: h_t of the layer below
    const int slot = blockIdx.x >> 3;
    const int vtile = (blockIdx.x & 7) + 8 * (slot / S), part = slot % S;
    int ji = 0;
    while (ji + 1 < jobs.n && vtile >= jobs.base[ji + 1]) ++ji;
    const pnmn_lstm_stack_job jb = jobs.j[ji];
    const int tile = vtile - jobs.base[ji];
    const int B = jb.B, T = jb.T;
    if (tile * LROWS >= B) return;  // (padding tiles of a job: nobody waits for them)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int row0 = tile * LROWS, u0 = part * UW;
    const int gate = wave >> 1, ub = wave & 1;  // waves 2q, 2q + 1 split gate q's 32 columns
    const bool second = jb.dep >= 0;
    const int cons = jobs.consumer[ji];
    Sync sy;
    sy.start(sync + vtile * pnmn::CLUSTER_COUNTER_STRIDE,
             cons >= 0 ? sync + (jobs.base[cons] + tile) * pnmn::CLUSTER_COUNTER_STRIDE : nullptr, second);

    const int ntile = gate * (LH / 16) + u0 / 16 + ub;
    f32x4_ whh[LH / 16], wih[LH / 16];
#pragma unroll
    for (int kb = 0; kb < LH / 16; ++kb) {
        whh[kb] = *reinterpret_cast<const f32x4_*>(jb.w_hh + ((size_t)(ntile * (LH / 16) + kb) * 64 + lane) * 4);
        wih[kb] = second ? *reinterpret_cast<const f32x4_*>(jb.w_ih + ((size_t)(ntile * (LH / 16) + kb) * 64 + lane) * 4)
                         : f32x4_{0.f, 0.f, 0.f, 0.f};
    }
    float creg = 0.f;  // J = UW * 16 / 512 = 1 (row, unit) pair per thread
    const float* below = second ? jobs.j[jb.dep].hs : nullptr;
    const float bias = second ? jb.bias[gate * LH + u0 + 16 * ub + li] : 0.f;

    // FIRST: this lane's four rows of the step inputs (see lstm_seq_fwd_cluster_kernel for the token look-ahead)
    const bool tok = jb.tokens != nullptr;
    int64_t xrow[4];
    const int64_t* trow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = min(row0 + 4 * g + r, B - 1);
        trow[r] = tok ? jb.tokens + (size_t)row * jb.token_stride : nullptr;
        xrow[r] = tok ? trow[r][0] : (int64_t)row * T;
    }

    // SECOND: the input product of step t, from the layer below (ahead of us by construction)
    auto xpart = [&](int t) -> f32x4_ {
        sy.wait_cross();
        stage_rows(below, row0, B, T, t, xl);
        __syncthreads();
        f32x4_ acc = f32x4_{bias, bias, bias, bias};
#pragma unroll
        for (int kb = 0; kb < LH / 16; ++kb) acc = mfma4(*reinterpret_cast<const f32x4_*>(&xl[li][kb * 16 + 4 * g]), wih[kb], acc);
        return acc;
    };

    f32x4_ xacc = f32x4_{0.f, 0.f, 0.f, 0.f};
    if (second) xacc = xpart(0);
    for (int t = 0; t < T; ++t) {
        f32x4_ acc;
        if (second) {
            acc = xacc;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = jb.xp[(size_t)xrow[r] * (4 * LH) + gate * LH + u0 + 16 * ub + li];
#pragma unroll
            for (int r = 0; r < 4; ++r) xrow[r] = tok ? trow[r][min(t + 1, T - 1)] : xrow[r] + 1;
        }
        if (t > 0) {
            sy.wait_own();
            stage_rows(jb.hs, row0, B, T, t - 1, hl);
            __syncthreads();
            f32x4_ a[LH / 16];
#pragma unroll
            for (int kb = 0; kb < LH / 16; ++kb) a[kb] = *reinterpret_cast<const f32x4_*>(&hl[li][kb * 16 + 4 * g]);
#pragma unroll
            for (int kb = 0; kb < LH / 16; ++kb) acc = mfma4(a[kb], whh[kb], acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) gl[gate][4 * g + r][16 * ub + li] = acc[r];
        __syncthreads();
        {
            const int rl = tid / UW, ul = tid % UW;
            const int row = row0 + rl, u = u0 + ul;
            const float ig = sigm(gl[0][rl][ul]);
            const float fg = sigm(gl[1][rl][ul]);
            const float gg = tanhf(gl[2][rl][ul]);
            const float og = sigm(gl[3][rl][ul]);
            const float c = fg * creg + ig * gg;
            const float h = og * tanhf(c);
            creg = c;
            if (row < B) {
                const size_t o = ((size_t)row * T + t) * LH + u;
                jb.hs[o] = h;
                jb.cs[o] = c;
                if (jb.act) {
                    float* ar = jb.act + ((size_t)row * T + t) * (4 * LH);
                    ar[u] = ig;
                    ar[LH + u] = fg;
                    ar[2 * LH + u] = gg;
                    ar[3 * LH + u] = og;
                }
            }
        }
        // partners need h_t for step t + 1; the layer above needs it for ITS step t (also after the last step)
        if (t + 1 < T || sy.cross_out) sy.signal(t + 1 < T);
        if (second && t + 1 < T) xacc = xpart(t + 1);
    }
    sy.finish();
}

// =========================================================================================================================
// backward
// =========================================================================================================================
__global__ __launch_bounds__(512) void lstm_stack_bwd_kernel(const Jobs jobs, int* sync) {
    constexpr int KB = 4 * UW / 16;      // k blocks of this workgroup's gate columns (recurrence product)
    constexpr int DLD = 4 * UW + 4;
    constexpr int XLD = 4 * LH + 4;
    __shared__ __attribute__((aligned(16))) float dgl[LROWS][DLD];
    extern __shared__ __attribute__((aligned(16))) float dyn[];
    float (*dgx)[XLD] = reinterpret_cast<float (*)[XLD]>(dyn);                    // BELOW: dgates of the layer above, [16][1028]
    float (*red)[LROWS][UW + 1] = reinterpret_cast<float (*)[LROWS][UW + 1]>(dyn + LROWS * XLD);  // [8 waves][16][33]
    const int slot = blockIdx.x >> 3;
    const int vtile = (blockIdx.x & 7) + 8 * (slot / S), part = slot % S;
    int ji = 0;
    while (ji + 1 < jobs.n && vtile >= jobs.base[ji + 1]) ++ji;
    const pnmn_lstm_stack_job jb = jobs.j[ji];
    const int tile = vtile - jobs.base[ji];
    const int B = jb.B, T = jb.T;
    if (tile * LROWS >= B) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int row0 = tile * LROWS, u0 = part * UW;
    const bool below = jb.dep >= 0;
    const int cons = jobs.consumer[ji];
    Sync sy;
    sy.start(sync + vtile * pnmn::CLUSTER_COUNTER_STRIDE,
             cons >= 0 ? sync + (jobs.base[cons] + tile) * pnmn::CLUSTER_COUNTER_STRIDE : nullptr, below);
    float* ptile = jobs.px[ji] + (size_t)tile * 2 * S * LROWS * LH;  // [parity][source part][16][H]

    // recurrence: this wave's two 16-unit output tiles x this workgroup's gate columns
    f32x4_ wreg[2][KB];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int kl = 0; kl < KB; ++kl) {
            const int q = kl / (UW / 16), jj = kl % (UW / 16);
            const int kb = (q * LH + u0) / 16 + jj;
            wreg[nt][kl] = *reinterpret_cast<const f32x4_*>(jb.w_hh + ((size_t)((2 * wave + nt) * (4 * LH / 16) + kb) * 64 + lane) * 4);
        }
    // BELOW: W_ih of the layer above ([1024 gate columns] x [my 32 units]): this wave's K slice of 128 columns
    f32x4_ wx[2][8];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int kl = 0; kl < 8; ++kl)
            wx[nt][kl] = below ? *reinterpret_cast<const f32x4_*>(jb.w_ih + ((size_t)((u0 / 16 + nt) * (4 * LH / 16) + 8 * wave + kl) * 64 + lane) * 4)
                               : f32x4_{0.f, 0.f, 0.f, 0.f};
    const float* above = below ? jobs.j[jb.dep].dgates : nullptr;
    const int rl = tid / UW, ul = tid % UW;   // this thread's (row, unit) pair of the cell backward
    const int row = row0 + rl, u = u0 + ul;
    float dh_rec = 0.f, dc_rec = 0.f;

    // BELOW: gradient wrt h_t arriving from the layer above = dgates_above[:, t, :] W_ih, this member's 32 units
    auto xpart = [&](int t) -> float {
        sy.wait_cross();
        {   // 16 x 1024 floats: eight 16-byte pieces per thread, all in flight before the first LDS store
            f32x4_ piece[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = tid + 512 * k, r = i / LH, c4 = i % LH;
                const int rr = min(row0 + r, B - 1);
                piece[k] = *reinterpret_cast<const f32x4_*>(above + ((size_t)rr * T + t) * (4 * LH) + 4 * c4);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = tid + 512 * k, r = i / LH, c4 = i % LH;
                *reinterpret_cast<f32x4_*>(&dgx[r][4 * c4]) = piece[k];
            }
        }
        __syncthreads();
        f32x4_ acc[2] = {f32x4_{0.f, 0.f, 0.f, 0.f}, f32x4_{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int kl = 0; kl < 8; ++kl) {
            const f32x4_ a = *reinterpret_cast<const f32x4_*>(&dgx[li][(8 * wave + kl) * 16 + 4 * g]);
            acc[0] = mfma4(a, wx[0][kl], acc[0]);
            acc[1] = mfma4(a, wx[1][kl], acc[1]);
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][4 * g + r][16 * nt + li] = acc[nt][r];
        __syncthreads();
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) sum += red[w][rl][ul];  // K slices in ascending order
        return sum;
    };

    float dho_next = below ? xpart(T - 1) : 0.f;
    for (int t = T - 1; t >= 0; --t) {
        // everything of step t that does not depend on the recurrence, before the wait
        float ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f, cc = 0.f, cp = 0.f, dho = 0.f;
        if (row < B) {
            const size_t o = ((size_t)row * T + t) * LH + u;
            const float* ar = jb.act + ((size_t)row * T + t) * (4 * LH);
            ig = ar[u], fg = ar[LH + u], gg = ar[2 * LH + u], og = ar[3 * LH + u];
            cc = jb.cs[o];
            cp = t > 0 ? jb.cs[o - LH] : 0.f;
            dho = below ? dho_next : jb.dhs[o];
        }
        if (t < T - 1) {
            sy.wait_own();
            const float* pp = ptile + (size_t)((t + 1) & 1) * S * LROWS * LH;
            float sum = 0.f;
#pragma unroll
            for (int s = 0; s < S; ++s) sum += pp[((size_t)s * LROWS + rl) * LH + u0 + ul];
            dh_rec = sum;
        }
        float di = 0.f, df = 0.f, dg = 0.f, dout = 0.f, dcp = 0.f;
        if (row < B) {
            const float tc = tanhf(cc);
            const float dh = dho + dh_rec;
            const float dc = dc_rec + dh * og * (1.f - tc * tc);
            di = dc * gg * ig * (1.f - ig);
            df = dc * cp * fg * (1.f - fg);
            dg = dc * ig * (1.f - gg * gg);
            dout = dh * tc * og * (1.f - og);
            dcp = dc * fg;
            float* dr = jb.dgates + ((size_t)row * T + t) * (4 * LH);
            dr[u] = di;
            dr[LH + u] = df;
            dr[2 * LH + u] = dg;
            dr[3 * LH + u] = dout;
        }
        dc_rec = dcp;
        dgl[rl][ul] = di;
        dgl[rl][UW + ul] = df;
        dgl[rl][2 * UW + ul] = dg;
        dgl[rl][3 * UW + ul] = dout;
        if (t == 0) {
            if (sy.cross_out) sy.signal(false);  // the layer below still needs dgates_0
            break;
        }
        __syncthreads();
        f32x4_ acc[2] = {f32x4_{0.f, 0.f, 0.f, 0.f}, f32x4_{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int kl = 0; kl < KB; ++kl) {
            const f32x4_ a = *reinterpret_cast<const f32x4_*>(&dgl[li][kl * 16 + 4 * g]);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[nt] = mfma4(wreg[nt][kl], a, acc[nt]);  // (weights as A: see lstm_seq_bwd_cluster_kernel)
        }
        float* po = ptile + ((size_t)(t & 1) * S + part) * LROWS * LH;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) *reinterpret_cast<f32x4_*>(po + (size_t)li * LH + 16 * (2 * wave + nt) + 4 * g) = acc[nt];
        sy.signal(true);
        if (below) dho_next = xpart(t - 1);
    }
    sy.finish();
}

constexpr size_t BWD_DYN_LDS = ((size_t)LROWS * (4 * LH + 4) + (size_t)8 * LROWS * (UW + 1)) * sizeof(float);

struct Layout {
    Jobs jobs;
    int vtiles;
    bool ok;
};

Layout lay_out(const pnmn_lstm_stack_job* in, int n) {
    Layout L;
    L.ok = false;
    L.vtiles = 0;
    if (n <= 0 || n > MAXJ) return L;
    int at = 0;
    for (int k = 0; k < n; ++k) {
        L.jobs.j[k] = in[k];
        L.jobs.base[k] = at;
        L.jobs.consumer[k] = -1;
        L.jobs.px[k] = nullptr;
        if (in[k].B <= 0 || in[k].T <= 0) return L;
        at += (((in[k].B + LROWS - 1) / LROWS) + 7) / 8 * 8;
    }
    L.jobs.base[n] = at;
    L.jobs.n = n;
    for (int k = 0; k < n; ++k) {
        const int d = in[k].dep;
        if (d < 0) continue;
        if (d >= n || d == k || in[d].dep >= 0 || L.jobs.consumer[d] >= 0 || in[d].B != in[k].B || in[d].T != in[k].T) return L;
        L.jobs.consumer[d] = k;
    }
    L.vtiles = at;
    // every workgroup resident at once (one per CU), and a counter line per virtual tile
    L.ok = at * S <= pnmn::device_cus() && at <= 128;
    return L;
}

}  // namespace

extern "C" int64_t pnmn_lstm_stack_workspace_bytes(const pnmn_lstm_stack_job* jobs, int n, int backward) {
    const Layout L = lay_out(jobs, n);
    if (!L.ok) return 0;
    int64_t bytes = (int64_t)pnmn::CLUSTER_SYNC_BYTES;
    if (backward) {
        for (int k = 0; k < n; ++k) bytes += (int64_t)((jobs[k].B + LROWS - 1) / LROWS) * 2 * S * LROWS * LH * sizeof(float);
    }
    return bytes;
}

extern "C" int pnmn_lstm_stack_fwd(const pnmn_lstm_stack_job* jobs, int n, void* workspace, void* stream) {
    if (n <= 0) return 0;
    if (!jobs || !workspace) return PNMN_EINVAL;
    const Layout L = lay_out(jobs, n);
    if (!L.ok) return PNMN_ESHAPE;
    for (int k = 0; k < n; ++k) {
        const pnmn_lstm_stack_job& j = jobs[k];
        if (!j.w_hh || !j.hs || !j.cs) return PNMN_EINVAL;
        if (j.dep >= 0 ? (!j.w_ih || !j.bias) : !j.xp) return PNMN_EINVAL;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    int* sync = nullptr;
    hipError_t e = pnmn::cluster_sync_block(workspace, st, &sync);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(lstm_stack_fwd_kernel, dim3(S * L.vtiles), dim3(512), 0, st, L.jobs, sync);
    return (int)hipGetLastError();
}

extern "C" int pnmn_lstm_stack_bwd(const pnmn_lstm_stack_job* jobs, int n, void* workspace, void* stream) {
    if (n <= 0) return 0;
    if (!jobs || !workspace) return PNMN_EINVAL;
    const Layout L = lay_out(jobs, n);
    if (!L.ok) return PNMN_ESHAPE;
    for (int k = 0; k < n; ++k) {
        const pnmn_lstm_stack_job& j = jobs[k];
        if (!j.w_hh || !j.act || !j.cs || !j.dgates) return PNMN_EINVAL;
        if (j.dep >= 0 ? !j.w_ih : !j.dhs) return PNMN_EINVAL;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    int* sync = nullptr;
    hipError_t e = pnmn::cluster_sync_block(workspace, st, &sync);
    if (e != hipSuccess) return (int)e;
    // exchange buffers behind the (capture-time) counter block
    Jobs J = L.jobs;
    char* at = static_cast<char*>(workspace) + pnmn::CLUSTER_SYNC_BYTES;
    for (int k = 0; k < n; ++k) {
        J.px[k] = reinterpret_cast<float*>(at);
        at += (size_t)((jobs[k].B + LROWS - 1) / LROWS) * 2 * S * LROWS * LH * sizeof(float);
    }
    static std::atomic<uint64_t> cfg{0};
    if (const int rc = pnmn::opt_in_lds(reinterpret_cast<const void*>(lstm_stack_bwd_kernel), BWD_DYN_LDS, cfg)) return rc;
    hipLaunchKernelGGL(lstm_stack_bwd_kernel, dim3(S * L.vtiles), dim3(512), BWD_DYN_LDS, st, J, sync);
    return (int)hipGetLastError();
}
