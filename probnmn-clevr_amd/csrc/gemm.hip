// fp32 GEMM for gfx950, batched over up to PNMN_GEMM_MAX problems per launch (include/probnmn_hip.h: pnmn_gemm_desc).
//
// What it serves: everything in the seq2seq models that is a plain matrix product over ALL time steps -- the second
// LSTM layer's input projection, the output projections and their data gradients, every weight gradient (a product over
// B x T rows: split-K) -- i.e. what the reference reaches through nn.LSTM / nn.Linear / autograd
// (probnmn/modules/seq2seq_base.py:101-155 via allennlp) and rounds 1-5 of this build sent to hipBLASLt through torch.
// As one launch per GROUP of independent products (all weight gradients of a model's backward are one launch) the
// 128-question step no longer pays ~20 library calls of 15-30 us host time and a dependent dispatch each.
//
// Kernel shape: 256 threads = 2 x 2 waves, workgroup tile 128 x 128, k-tiles of 32, v_mfma_f32_32x32x2_f32 (a wave owns
// 64 x 64 = four accumulators of 16 registers).  An operand whose storage is contiguous along k ("KC": A [M][K], B given
// as [N][K]) is staged row-major [128][32 + 4] and read back with one ds_read_b128 per (32-row tile, 8 k): lane (i, h)
// takes k = 8 q + 4 h .. + 3, and the four MFMAs of that group contract k = 8 q + 4 h + s from both operands alike (a
// permutation of the k order inside a tile: the sum is the same, the rounding order fixed).  An operand stored with the
// OTHER index contiguous ("MC": A given as [K][M] -- a weight gradient's dy^T -- or B [K][N]) is staged as it lies,
// [32][128 + 4], with 16-byte stores, and read with ds_read_b32 at the same k.  Global loads of tile t + 1 are in flight
// while tile t is contracted (registers -> the other LDS buffer, one barrier per tile).
// Split-K: blockIdx walks (problem, tile, chunk); chunks write partial tiles to the problem's workspace and a second
// small launch (gemm_reduce_kernel) adds them in chunk order -- deterministic, no atomics on the output; the kernel
// boundary makes the partials visible (a fence + "last chunk reduces" in-kernel cost an L2 write-back per workgroup).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/probnmn_hip.h"
#include "global_ptr.h"
#include "lds_optin.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
using pnmn::as_global;
using pnmn::gfloat;

constexpr int TM = 128, TN = 128, TK = 32;
constexpr int KC_LD = TK + 4;   // row-major [128][36]: 144-byte rows, 16-byte aligned, conflict-free ds_read_b128
constexpr int MC_LD = TM + 4;   // k-major  [32][132]
constexpr int OP_FLOATS = TM * KC_LD > TK * MC_LD ? TM * KC_LD : TK * MC_LD;  // 4608 floats = 18 KB per operand buffer

struct Batch {
    pnmn_gemm_desc d[PNMN_GEMM_MAX];
    int first[PNMN_GEMM_MAX + 1];  // first block of each problem
    int n;
};

// One operand tile (128 rows x 32 k) from global memory into registers: four 16-byte pieces per thread.
// KC (k contiguous in storage): piece p = tid + 256 j -> row p / 8, k quad p % 8.
// MC (row index contiguous):    piece p -> k row p / 32, row quad p % 32.
// `rows` / `K`: bounds (pieces outside are zeros; rows past the end only feed outputs nobody stores, but k past the end
// would enter every sum).  shift_t > 0 (MC only): k row r reads storage row r - 1, or h0[r / shift_t] (zeros when h0 is
// null) where r % shift_t == 0 -- "the previous time step's state" of a [B][T][.] tensor without materialising it.
template <bool KC>
__device__ __forceinline__ void load_tile(const gfloat* base, int64_t ld, int row0, int k0, int rows, int K, bool vec, int tid,
                                          f32x4 (&r)[4], int shift_t, const gfloat* h0, int64_t ld_h0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = tid + 256 * j;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (KC) {
            const int row = row0 + (p >> 3), k = k0 + 4 * (p & 7);
            if (row < rows && k < K) {
                const gfloat* src = base + (int64_t)row * ld + k;
                if (vec && k + 3 < K) {
                    v = pnmn::load4(src);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (k + e < K) v[e] = src[e];
                }
            }
        } else {
            const int k = k0 + (p >> 5), row = row0 + 4 * (p & 31);
            if (k < K && row < rows) {
                const gfloat* src;
                bool zero = false;
                if (shift_t > 0) {
                    const int b = k / shift_t;
                    if (k - b * shift_t == 0) {
                        zero = h0 == nullptr;
                        src = h0 + (int64_t)b * ld_h0 + row;
                    } else {
                        src = base + (int64_t)(k - 1) * ld + row;
                    }
                } else {
                    src = base + (int64_t)k * ld + row;
                }
                if (!zero) {
                    if (vec && row + 3 < rows) {
                        v = pnmn::load4(src);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (row + e < rows) v[e] = src[e];
                    }
                }
            }
        }
        r[j] = v;
    }
}

// The same for a tile that lies wholly inside the operand (all 128 rows, all 32 k, 16-byte aligned rows): four
// unconditional 16-byte loads from per-thread pointers that advance by one k-tile per call -- the bounds-checked loader
// above compiles to a branch per piece (exec-masked scalar fall-backs), which serialises the eight loads of a k-tile.
template <bool KC>
__device__ __forceinline__ void load_tile_fast(const gfloat*& ptr, int64_t piece_stride, int64_t step, f32x4 (&r)[4]) {
    // (ONE per-thread pointer per operand: the pieces of a thread lie a uniform stride apart -- 32 rows (KC) / 8 k rows (MC)
    // -- which stays in scalar registers; four pointers per operand cost 16 VGPRs and the second workgroup of a CU)
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = pnmn::load4(ptr + j * piece_stride);
    ptr += step;
}

template <bool KC>
__device__ __forceinline__ const gfloat* tile_pointer(const gfloat* base, int64_t ld, int row0, int k0, int tid) {
    return KC ? base + (int64_t)(row0 + (tid >> 3)) * ld + k0 + 4 * (tid & 7) : base + (int64_t)(k0 + (tid >> 5)) * ld + row0 + 4 * (tid & 31);
}

template <bool KC>
__device__ __forceinline__ void store_tile(float* lds, int tid, const f32x4 (&r)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = tid + 256 * j;
        if (KC)
            *reinterpret_cast<f32x4*>(lds + (p >> 3) * KC_LD + 4 * (p & 7)) = r[j];
        else
            *reinterpret_cast<f32x4*>(lds + (p >> 5) * MC_LD + 4 * (p & 31)) = r[j];
    }
}

// the four operand values of k group q (k = 8 q + 4 h + s, s = 0..3) for the 32-row tile at `row`
template <bool KC>
__device__ __forceinline__ f32x4 frag(const float* lds, int row, int q, int h) {
    if (KC) return *reinterpret_cast<const f32x4*>(lds + row * KC_LD + 8 * q + 4 * h);
    const float* p = lds + (8 * q + 4 * h) * MC_LD + row;
    return f32x4{p[0], p[MC_LD], p[2 * MC_LD], p[3 * MC_LD]};
}

// `cs` (A stored [K][M] only): the thread's share of the COLUMN SUMS of A over this chunk's k range -- piece j of a tile is
// k row tid / 32 + 8 j, m quad tid % 32, so a thread meets the same four columns in every piece and adds them up as the
// registers arrive for the LDS store (sum_k A[k][m] = the bias gradient that goes with a weight gradient dy^T x).
template <bool AKC, bool BKC>
__device__ __forceinline__ void contract(const pnmn_gemm_desc& d, int m0, int n0, int kbeg, int kend, float* lds, f32x16 (&acc)[2][2],
                                         bool want_cs, f32x4& cs) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const gfloat* A = as_global(d.a);
    const gfloat* Bm = as_global(d.b);
    const gfloat* h0 = as_global(d.shift_h0);
    // operand A: AKC = stored [M][K] (lda = row stride), else [K][M];  operand B: BKC = stored [N][K], else [K][N]
    const bool avec = (d.lda & 3) == 0 && ((uintptr_t)d.a & 15) == 0;
    const bool bvec = (d.ldb & 3) == 0 && ((uintptr_t)d.b & 15) == 0 && (d.shift_t == 0 || d.shift_h0 == nullptr ||
                                                                       ((d.ld_h0 & 3) == 0 && ((uintptr_t)d.shift_h0 & 15) == 0));
    float* la = lds;                    // [2][OP_FLOATS]
    float* lb = lds + 2 * OP_FLOATS;    // [2][OP_FLOATS]
    f32x4 ra[4], rb[4];
    // interior tiles (the common case) take the straight-line loader; `kfull` = k-tiles that lie wholly below kend
    const bool afast = avec && m0 + TM <= d.M, bfast = bvec && n0 + TN <= d.N && (BKC || d.shift_t == 0);
    const int kfull = kbeg + (kend - kbeg) / TK * TK;
    const gfloat* pa = tile_pointer<AKC>(A, d.lda, m0, kbeg, tid);
    const gfloat* pb = tile_pointer<BKC>(Bm, d.ldb, n0, kbeg, tid);
    const int64_t sa = AKC ? TK : (int64_t)TK * d.lda, sb = BKC ? TK : (int64_t)TK * d.ldb;
    const int64_t ja = (AKC ? 32 : 8) * d.lda, jb = (BKC ? 32 : 8) * d.ldb;  // piece j = tid + 256 j: 32 rows / 8 k rows on
    auto fetch = [&](int k0) {
        if (afast && k0 < kfull)
            load_tile_fast<AKC>(pa, ja, sa, ra);
        else
            load_tile<AKC>(A, d.lda, m0, k0, d.M, kend, avec, tid, ra, 0, nullptr, 0);
        if (bfast && k0 < kfull)
            load_tile_fast<BKC>(pb, jb, sb, rb);
        else
            load_tile<BKC>(Bm, d.ldb, n0, k0, d.N, kend, bvec, tid, rb, BKC ? 0 : d.shift_t, h0, d.ld_h0);
    };
    fetch(kbeg);
    store_tile<AKC>(la, tid, ra);
    store_tile<BKC>(lb, tid, rb);
    if (!AKC && want_cs) cs += (ra[0] + ra[1]) + (ra[2] + ra[3]);
    __syncthreads();
    int cur = 0;
    for (int k0 = kbeg; k0 < kend; k0 += TK) {
        const bool more = k0 + TK < kend;
        if (more) fetch(k0 + TK);
        const float* ca = la + cur * OP_FLOATS;
        const float* cb = lb + cur * OP_FLOATS;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t] = frag<AKC>(ca, 64 * wm + 32 * t + i, q, h);
                b[t] = frag<BKC>(cb, 64 * wn + 32 * t + i, q, h);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][s], b[nt][s], acc[mt][nt], 0, 0, 0);
        }
        if (more) {
            store_tile<AKC>(la + (cur ^ 1) * OP_FLOATS, tid, ra);
            store_tile<BKC>(lb + (cur ^ 1) * OP_FLOATS, tid, rb);
            if (!AKC && want_cs) cs += (ra[0] + ra[1]) + (ra[2] + ra[3]);
        }
        __syncthreads();
        cur ^= 1;
    }
}

// (waves_per_eu 2: the register allocator must stay within 256 registers per wave, accumulators included -- at 260 the second
// workgroup of a CU is gone and every product of the plan ran 20 % slower)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_kernel(const Batch batch) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [4][OP_FLOATS] = 72 KB: two workgroups per CU
    // (a launch cut for fewer workgroups than it has units -- pnmn_gemm_cus: products that share the chip with another
    // stream's latency chain -- walks them; the LDS buffers are free again behind the last barrier of contract())
    for (int unit = blockIdx.x; unit < batch.first[batch.n]; unit += gridDim.x) {
    int pi = 0;
    while (pi + 1 < batch.n && unit >= batch.first[pi + 1]) ++pi;
    const pnmn_gemm_desc d = batch.d[pi];  // (into scalar registers: the kernel argument segment is read once)
    const int local = unit - batch.first[pi];
    const int tiles_n = (d.N + TN - 1) / TN, tiles_m = (d.M + TM - 1) / TM;
    const int split = d.split_k > 1 ? d.split_k : 1;
    // chunk fastest: the chunks of one output tile run together and the last one finds the others' partials in L2
    const int chunk = local % split, tile = local / split;
    const int tm = tile / tiles_n, tn = tile % tiles_n;
    const int m0 = tm * TM, n0 = tn * TN;
    // k range of this chunk: whole k-tiles, the first chunks one longer
    const int ktiles = (d.K + TK - 1) / TK;
    const int per = ktiles / split, extra = ktiles % split;
    const int kt0 = chunk * per + (chunk < extra ? chunk : extra);
    const int kt1 = kt0 + per + (chunk < extra ? 1 : 0);
    const int kbeg = kt0 * TK, kend = kt1 * TK < d.K ? kt1 * TK : d.K;

    f32x16 acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    const bool akc = !(d.flags & PNMN_GEMM_A_TRANSPOSED), bkc = (d.flags & PNMN_GEMM_B_TRANSPOSED) != 0;
    const bool want_cs = d.colsum != nullptr && !akc && tn == 0;  // (the first tile column of a row of tiles carries them)
    f32x4 cs = f32x4{0.f, 0.f, 0.f, 0.f};
    if (kbeg < kend) {
        if (akc && bkc)
            contract<true, true>(d, m0, n0, kbeg, kend, lds, acc, false, cs);
        else if (akc)
            contract<true, false>(d, m0, n0, kbeg, kend, lds, acc, false, cs);
        else if (bkc)
            contract<false, true>(d, m0, n0, kbeg, kend, lds, acc, want_cs, cs);
        else
            contract<false, false>(d, m0, n0, kbeg, kend, lds, acc, want_cs, cs);
    }

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    if (want_cs) {
        // eight threads (k rows u, u + 8, ...: u = tid / 32) hold partial sums of the same four columns: added in u order
        // through LDS (free behind contract()'s last barrier); a chunk's 128 sums go behind the partial tiles in the
        // workspace, an unsplit product's straight to the output
        *reinterpret_cast<f32x4*>(lds + (tid >> 5) * TM + 4 * (tid & 31)) = cs;
        __syncthreads();
        if (tid < TM) {
            float t = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) t += lds[u * TM + tid];
            if (split > 1) {
                as_global(d.workspace)[(size_t)tiles_m * tiles_n * split * (TM * TN) + ((size_t)tm * split + chunk) * TM + tid] = t;
            } else if (m0 + tid < d.M) {
                d.colsum[m0 + tid] = t;
                if (d.colsum2) d.colsum2[m0 + tid] = t;
            }
        }
        __syncthreads();  // (the next unit of a persistent workgroup stages into the same LDS)
    }
    // accumulator (mt, nt), register r: row 64 wm + 32 mt + 8 (r / 4) + 4 h + r % 4, column 64 wn + 32 nt + i
    if (split > 1) {
        gfloat* ws = as_global(d.workspace) + ((size_t)tile * split + chunk) * (TM * TN);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ws[(64 * wm + 32 * mt + 8 * (r >> 2) + 4 * h + (r & 3)) * TN + 64 * wn + 32 * nt + i] = acc[mt][nt][r];
        continue;  // (gemm_reduce_kernel, launched behind this one, adds the chunks up in chunk order)
    }
    gfloat* C = as_global(d.c);
    const gfloat* bias = as_global(d.bias);
    const bool accumulate = (d.flags & PNMN_GEMM_ACCUMULATE) != 0;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int col = n0 + 64 * wn + 32 * nt + i;
        if (col >= d.N) continue;
        const float bv = d.bias ? bias[col] : 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + 64 * wm + 32 * mt + 8 * (r >> 2) + 4 * h + (r & 3);
                if (row < d.M) {
                    gfloat* dst = C + (int64_t)row * d.ldc + col;
                    float v = acc[mt][nt][r] + bv;
                    if (accumulate) v += *dst;
                    *dst = v;
                }
            }
    }
    }
}

// Second launch of a split-K batch: every output tile's partial tiles added in chunk order (deterministic; the kernel
// boundary is what makes the partials visible -- a fence + "last chunk reduces" inside the product kernel cost an L2
// write-back per workgroup, 150 us for a 3 GFLOP weight gradient).  Block = a quarter tile (32 rows x 128 columns).
__global__ __launch_bounds__(256) void gemm_reduce_kernel(const Batch batch) {
    int pi = 0;
    while (pi + 1 < batch.n && (int)blockIdx.x >= batch.first[pi + 1]) ++pi;
    const pnmn_gemm_desc d = batch.d[pi];
    const int local = blockIdx.x - batch.first[pi];
    const int tiles_n = (d.N + TN - 1) / TN;
    const int quarter = local & 3, tile = local >> 2;
    const int m0 = (tile / tiles_n) * TM + 32 * quarter, n0 = (tile % tiles_n) * TN;
    const int split = d.split_k, tid = threadIdx.x;
    const gfloat* wt = as_global(d.workspace) + (size_t)tile * split * (TM * TN) + quarter * (32 * TN);
    f32x4 sum[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) sum[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < split; ++c) {
        const gfloat* wc = wt + (size_t)c * (TM * TN);
        f32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = pnmn::load4(wc + 4 * (tid + 256 * j));
#pragma unroll
        for (int j = 0; j < 4; ++j) sum[j] += v[j];
    }
    if (d.colsum && (d.flags & PNMN_GEMM_A_TRANSPOSED) && quarter == 0 && tile % tiles_n == 0 && tid < TM) {
        // the chunks' column sums of A (gemm_kernel), in chunk order
        const int tiles_m = (d.M + TM - 1) / TM, tm = tile / tiles_n;
        const gfloat* wc = as_global(d.workspace) + (size_t)tiles_m * tiles_n * split * (TM * TN) + (size_t)tm * split * TM + tid;
        float t = 0.f;
        for (int c = 0; c < split; ++c) t += wc[(size_t)c * TM];
        if (tm * TM + tid < d.M) {
            d.colsum[tm * TM + tid] = t;
            if (d.colsum2) d.colsum2[tm * TM + tid] = t;
        }
    }
    gfloat* C = as_global(d.c);
    const gfloat* bias = as_global(d.bias);
    const bool accumulate = (d.flags & PNMN_GEMM_ACCUMULATE) != 0;
    const bool cvec = (d.ldc & 3) == 0 && ((uintptr_t)d.c & 15) == 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = tid + 256 * j, row = m0 + (p >> 5), col = n0 + 4 * (p & 31);
        if (row >= d.M || col >= d.N) continue;
        gfloat* dst = C + (int64_t)row * d.ldc + col;
        f32x4 v = sum[j];
        if (cvec && col + 3 < d.N) {
            if (d.bias) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += bias[col + e];
            }
            if (accumulate) v += pnmn::load4(dst);
            pnmn::store4(dst, v);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (col + e < d.N) {
                    float x = v[e] + (d.bias ? bias[col + e] : 0.f);
                    if (accumulate) x += dst[e];
                    dst[e] = x;
                }
        }
    }
}

// column sums of a row-major [R][C] matrix: out[c] = sum_r x[r][c] (optionally also written to out2; bias gradients of an
// LSTM layer: b_ih and b_hh receive the same).  grid (column blocks of 256, slices of rows); a workgroup = 64 lanes x 16 bytes
// across (a 1 KiB row segment per wave and load) x 4 row phases, four loads in flight per thread; partials -> the last slice
// of a column block adds them in slice order (deterministic).  Scalar path for C % 4 != 0 (the vocabulary-wide dlogits).
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int64_t ld, int R, int C, float* __restrict__ partial,
                                                     int* __restrict__ counter, float* __restrict__ out, float* __restrict__ out2,
                                                     int accumulate, int vec) {
    __shared__ int last_flag;
    __shared__ float red[4][256];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c0 = blockIdx.x * 256 + 4 * tx;
    const int slices = gridDim.y, s = blockIdx.y;
    const int per = (R + slices - 1) / slices;
    const int r0 = s * per, r1 = min(R, r0 + per);
    f32x4 acc[4] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    if (c0 < C) {
        if (vec && c0 + 3 < C) {
            int r = r0 + ty;
            for (; r + 12 < r1; r += 16) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] += *reinterpret_cast<const f32x4*>(x + (int64_t)(r + 4 * e) * ld + c0);
            }
            for (; r < r1; r += 4) acc[0] += *reinterpret_cast<const f32x4*>(x + (int64_t)r * ld + c0);
        } else {
            for (int r = r0 + ty; r < r1; r += 4)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (c0 + k < C) acc[0][k] += x[(int64_t)r * ld + c0 + k];
        }
    }
    const f32x4 t = (acc[0] + acc[1]) + (acc[2] + acc[3]);
#pragma unroll
    for (int k = 0; k < 4; ++k) red[ty][4 * tx + k] = t[k];
    __syncthreads();
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < C) partial[(size_t)s * C + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        const int old = atomicAdd(counter + blockIdx.x, 1);
        last_flag = old == slices - 1;
        if (last_flag) counter[blockIdx.x] = 0;
    }
    __syncthreads();
    if (!last_flag || c >= C) return;
    __threadfence();
    float total = 0.f;
    for (int k = 0; k < slices; ++k) total += __builtin_nontemporal_load(partial + (size_t)k * C + c);
    if (accumulate) total += out[c];
    out[c] = total;
    if (out2) out2[c] = total;
}

}  // namespace

extern "C" int64_t pnmn_gemm_workspace_bytes(int M, int N, int split_k) {
    if (split_k <= 1) return 0;
    const int64_t tiles_m = (M + TM - 1) / TM, tiles = tiles_m * ((N + TN - 1) / TN);
    // partial tiles [tile][chunk][128 x 128], then the chunks' column sums of A [tile row][chunk][128]
    return (tiles * split_k * (TM * TN) + tiles_m * split_k * TM) * (int64_t)sizeof(float);
}

extern "C" int pnmn_gemm_split_k(int M, int N, int K, int cus) {
    // chunks so that tiles x chunks fills about two workgroups per CU, at least four k-tiles each
    const int64_t tiles = (int64_t)((M + TM - 1) / TM) * ((N + TN - 1) / TN);
    const int ktiles = (K + TK - 1) / TK;
    if (cus <= 0) cus = 256;
    int64_t want = (2 * (int64_t)cus + tiles - 1) / tiles;
    if (tiles >= cus) want = 1;
    const int most = ktiles / 4 > 0 ? ktiles / 4 : 1;
    if (want > most) want = most;
    if (want > 64) want = 64;
    return want < 1 ? 1 : (int)want;
}

extern "C" int pnmn_gemm(const pnmn_gemm_desc* descs, int n, void* stream) { return pnmn_gemm_cus(descs, n, 0, stream); }

extern "C" int pnmn_gemm_cus(const pnmn_gemm_desc* descs, int n, int max_workgroups, void* stream) {
    if (n <= 0) return 0;
    if (!descs || n > PNMN_GEMM_MAX) return PNMN_EINVAL;
    Batch b;
    int blocks = 0, live = 0;
    for (int k = 0; k < n; ++k) {
        const pnmn_gemm_desc& d = descs[k];
        if (d.M <= 0 || d.N <= 0) continue;
        if (!d.a || !d.b || !d.c || d.K < 0) return PNMN_EINVAL;
        if (d.split_k > 1 && !d.workspace) return PNMN_EINVAL;
        if (d.shift_t > 0 && (d.flags & PNMN_GEMM_B_TRANSPOSED)) return PNMN_ESHAPE;  // (the shifted operand is B as [K][N])
        if (d.colsum && !(d.flags & PNMN_GEMM_A_TRANSPOSED)) return PNMN_ESHAPE;      // (column sums: of A stored [K][M])
        b.d[live] = d;
        b.first[live] = blocks;
        const int tiles = ((d.M + TM - 1) / TM) * ((d.N + TN - 1) / TN);
        blocks += tiles * (d.split_k > 1 ? d.split_k : 1);
        ++live;
    }
    if (!live) return 0;
    b.first[live] = blocks;
    b.n = live;
    constexpr size_t lds = (size_t)4 * OP_FLOATS * sizeof(float);
    static std::atomic<uint64_t> cfg{0};  // (per device: lds_optin.h)
    if (const int e = pnmn::opt_in_lds(reinterpret_cast<const void*>(gemm_kernel), lds, cfg)) return e;
    const int grid = (max_workgroups > 0 && max_workgroups < blocks) ? max_workgroups : blocks;
    hipLaunchKernelGGL(gemm_kernel, dim3(grid), dim3(256), lds, static_cast<hipStream_t>(stream), b);
    // the split problems' reduction: four blocks per output tile
    Batch r;
    int rblocks = 0, rlive = 0;
    for (int k = 0; k < live; ++k) {
        if (b.d[k].split_k <= 1) continue;
        r.d[rlive] = b.d[k];
        r.first[rlive] = rblocks;
        rblocks += 4 * ((b.d[k].M + TM - 1) / TM) * ((b.d[k].N + TN - 1) / TN);
        ++rlive;
    }
    if (rlive) {
        r.first[rlive] = rblocks;
        r.n = rlive;
        hipLaunchKernelGGL(gemm_reduce_kernel, dim3(rblocks), dim3(256), 0, static_cast<hipStream_t>(stream), r);
    }
    return (int)hipGetLastError();
}

extern "C" int64_t pnmn_colsum_workspace_bytes(int R, int C) {
    (void)R;
    return (int64_t)128 * C * sizeof(float) + (int64_t)((C + 255) / 256) * sizeof(int);
}

extern "C" int pnmn_colsum(const float* x, int64_t ld, int R, int C, float* out, float* out2, int accumulate, void* workspace,
                           void* stream) {
    if (C <= 0) return 0;
    if (!out || !workspace || (R > 0 && !x)) return PNMN_EINVAL;
    int slices = (R + 63) / 64;
    if (slices > 128) slices = 128;
    if (slices < 1) slices = 1;
    float* partial = static_cast<float*>(workspace);
    int* counter = reinterpret_cast<int*>(partial + (size_t)128 * C);
    const int vec = (ld & 3) == 0 && ((uintptr_t)x & 15) == 0;
    hipLaunchKernelGGL(colsum_kernel, dim3((C + 255) / 256, slices), dim3(256), 0, static_cast<hipStream_t>(stream), x, ld, R, C,
                       partial, counter, out, out2, accumulate, vec);
    return (int)hipGetLastError();
}
