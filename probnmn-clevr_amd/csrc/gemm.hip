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
// 64 x 64 = four accumulators of 16 registers), two stages of 2 x 16 KB in LDS -> two workgroups per CU.  An operand whose
// storage is contiguous along k ("KC": A [M][K], B given as [N][K]) lies in LDS as [128 rows][32 k] with the 16-byte pieces of
// a row permuted by the row index (store_tile: conflict-free ds_read_b128 without padding) and is read back with one
// ds_read_b128 per (32-row tile, 8 k): lane (i, h) takes k = 8 q + 4 h .. + 3, and the four MFMAs of that group contract
// k = 8 q + 4 h + s from both operands alike (a permutation of the k order inside a tile: the sum is the same, the rounding
// order fixed).  An operand stored with the OTHER index contiguous ("MC": A given as [K][M] -- a weight gradient's dy^T -- or
// B [K][N]) lies as it is stored, [32][128], and is read with ds_read_b32 at the same k.  Interior tiles travel from global
// memory straight into that image (global_load_lds_dwordx4, struct Direct: no staging registers, no ds_write pass) while
// the tile before them is contracted; ragged tiles, unaligned operands and the shifted "previous state" operand go through
// registers (load_tile / load_tile_shift).  One barrier per k tile.
// Split-K: blockIdx walks (problem, tile, chunk); chunks write partial tiles to the problem's workspace and a second
// small launch (gemm_reduce_kernel) adds them in chunk order -- deterministic, no atomics on the output; the kernel
// boundary makes the partials visible (a fence + "last chunk reduces" in-kernel cost an L2 write-back per workgroup).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/probnmn_hip.h"
#include "global_ptr.h"
#include "lds_optin.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
using pnmn::as_global;
using pnmn::gfloat;

constexpr int TM = 128, TN = 128, TK = 32;
constexpr int OP_FLOATS = TM * TK;  // 4096 floats = 16 KB per operand tile and stage, no padding (see store_tile)
using lchar = __attribute__((address_space(3))) char;
using gchar = __attribute__((address_space(1))) char;

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

// The shifted operand's INTERIOR tiles (stored [K][N], all 32 k rows and 128 columns inside, 16-byte aligned rows): k row r
// reads storage row r - 1, or h0[r / shift_t] where r % shift_t == 0 (zeros when h0 is null) -- straight-line: the source
// pointer is selected, not branched on (the bounds-checked loader above is a branch per piece with an integer division).
__device__ __forceinline__ void load_tile_shift(const gfloat* base, int64_t ld, int col0, int k0, int tid, f32x4 (&r)[4], int shift_t,
                                                float rcp_t, const gfloat* h0, int64_t ld_h0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = k0 + (tid >> 5) + 8 * j, col = col0 + 4 * (tid & 31);
        const int b = (int)(((float)k + 0.5f) * rcp_t);  // k / shift_t (the true quotient of the half-integer is >= 1 / (2 shift_t) away from an integer, the float product within k * 1.2e-7 of it: exact below 2^22 rows; the caller stops at 2^21)
        const bool first = k == b * shift_t;
        const gfloat* prev = base + (int64_t)(k - 1) * ld + col;  // (k = 0 is always `first`: never dereferenced)
        const gfloat* init = h0 ? h0 + (int64_t)b * ld_h0 + col : base + col;
        f32x4 v = pnmn::load4(first ? init : prev);
        if (first && !h0) v = f32x4{0.f, 0.f, 0.f, 0.f};
        r[j] = v;
    }
}

// LDS image of an operand tile (16 KB, the same for the register loader and the direct-to-LDS loader):
//   KC  row r = 128 bytes = eight 16-byte pieces; piece g (k = 4 g .. 4 g + 3) sits at position g ^ ((r >> 1) & 7).
//       ds_read_b128 is served in four groups of 16 lanes ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32);
//       rows of equal parity share banks, and within a group the eight values of (r >> 1) & 7 occur once per parity: the
//       sixteen lanes of a group read sixteen different (parity, position) slots -- no conflicts, no padding;
//   MC  k row = 512 bytes as it lies; ds_read_b32 of consecutive lanes reads consecutive words.
template <bool KC>
__device__ __forceinline__ void store_tile(float* lds, int tid, const f32x4 (&r)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = tid + 256 * j;
        if (KC) {
            const int row = p >> 3;
            *reinterpret_cast<f32x4*>(lds + row * TK + 4 * ((p & 7) ^ ((row >> 1) & 7))) = r[j];
        } else {
            *reinterpret_cast<f32x4*>(lds + (p >> 5) * TM + 4 * (p & 31)) = r[j];
        }
    }
}

// the four operand values of k group q (k = 8 q + 4 h + s, s = 0..3) for the 32-row tile at `row`
template <bool KC>
__device__ __forceinline__ f32x4 frag(const float* lds, int row, int q, int h) {
    if (KC) return *reinterpret_cast<const f32x4*>(lds + row * TK + 4 * ((2 * q + h) ^ ((row >> 1) & 7)));
    const float* p = lds + (8 * q + 4 * h) * TM + row;
    return f32x4{p[0], p[TM], p[2 * TM], p[3 * TM]};
}

// Interior tiles go from global memory STRAIGHT into that image (global_load_lds_dwordx4: no staging registers, no
// ds_write pass).  A wave-load writes 1 KiB lane-linear (base + 16 lane), so the piece permutation is applied to the SOURCE
// address: wave w fills rows 32 w .. + 31 (KC: four loads of eight rows; lane l -> row 8 j + l / 8, position l % 8, i.e.
// piece (l % 8) ^ (4 (j & 1) + l / 16)) or k rows 8 w .. + 7 (MC: four loads of two k rows).
template <bool KC>
struct Direct {
    // source of load j = base (wave-uniform: scalar registers, advanced on the scalar unit) + j * piece + this lane's offset
    const gchar* base;
    uint32_t v0, v1;      // the lane's byte offset for the even / the odd loads (KC: their piece index differs by 4)
    int64_t piece, step;  // bytes from load j to j + 1, bytes per k tile (uniform)
    uint32_t off;         // byte offset of the wave's first load inside the operand tile (uniform)

    __device__ __forceinline__ void start(const gfloat* origin, int64_t ld, int row0, int k0, int tid) {
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
        // (the pointer comes out of a descriptor in the kernel arguments: uniform, but only readfirstlane says so)
        const uint64_t o64 = (uint64_t)(uintptr_t)origin;
        const gchar* o = (const gchar*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(o64 >> 32)) << 32) |
                                        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)o64));
        if (KC) {
            const int c = (lane & 7) ^ (lane >> 4);
            base = o + ((int64_t)(row0 + 32 * wave) * ld + k0) * 4;
            v0 = (uint32_t)((lane >> 3) * ld) * 4u + 16u * c, v1 = (uint32_t)((lane >> 3) * ld) * 4u + 16u * (c ^ 4);
            piece = 32 * ld, step = 4 * TK, off = (uint32_t)wave * 32u * 128u;
        } else {
            base = o + ((int64_t)(k0 + 8 * wave) * ld + row0) * 4;
            v0 = v1 = (uint32_t)((lane >> 5) * ld) * 4u + 16u * (lane & 31);
            piece = 8 * ld, step = 4 * (int64_t)TK * ld, off = (uint32_t)wave * 8u * 512u;
        }
    }
    // Issued as inline assembly: the compiler orders a __builtin_amdgcn_global_load_lds against EVERY later LDS read with
    // s_waitcnt vmcnt(0) (it cannot tell the stage being filled from the stage being read), which puts the whole global
    // latency of the prefetch in front of the first fragment read of every k tile.  contract() waits itself (vmcnt(0) +
    // barrier at the end of a k tile).  M0 = LDS byte address of the load's 1 KiB; nothing else in this kernel uses M0.
    __device__ __forceinline__ void issue(float* tile) {
        const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)((lchar*)tile)) + off;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint64_t src = (uint64_t)(uintptr_t)(base + j * piece);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                         :
                         : "v"(KC && (j & 1) ? v1 : v0), "s"(src), "s"(dst + 1024u * j)
                         : "memory");
        }
        base += step;
    }
};

// `cs` (A stored [K][M] only; waves of the first wave column): this lane's share of the COLUMN SUMS of A over the chunk's k
// range -- the A fragments a wave feeds its MFMAs with hold A[k][row] for all 32 k of a tile and the wave's 64 rows, lane
// (i, h) those of k = 8 q + 4 h + s: adding them up as they pass costs eight additions per k group and no memory access
// (sum_k A[k][m] = the bias gradient that goes with a weight gradient dy^T x).
template <bool AKC, bool BKC>
__device__ __forceinline__ void contract(const pnmn_gemm_desc& d, int m0, int n0, int kbeg, int kend, float* lds, f32x16 (&acc)[2][2],
                                         bool want_cs, float (&cs)[2]) {
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
    // interior tiles (the common case) go straight to LDS; `kfull` = k-tiles that lie wholly below kend; ragged tiles,
    // unaligned operands and the shifted operand take the bounds-checked loader through registers
    const bool adir = avec && m0 + TM <= d.M, bdir = bvec && n0 + TN <= d.N && (BKC || d.shift_t == 0);
    const int kfull = kbeg + (kend - kbeg) / TK * TK;
    const float rcp_t = d.shift_t > 0 ? 1.f / (float)d.shift_t : 0.f;
    Direct<AKC> da;
    Direct<BKC> db;
    da.start(A, d.lda, m0, kbeg, tid);
    db.start(Bm, d.ldb, n0, kbeg, tid);
    // One loop per loader combination (compile time: a loop that may take either path carries the staging registers through
    // every iteration, and the copies at the joins wait for ALL outstanding loads, the direct ones included).
    //   AD / BD: the operand goes straight to LDS.  k tiles [lo, hi); stage 0 first; every loop starts and ends with all of
    //   LDS free (behind a barrier).
    auto loop = [&](auto ad_tag, auto bd_tag, int lo, int hi) {
        constexpr bool AD = decltype(ad_tag)::value, BD = decltype(bd_tag)::value;
        f32x4 ra[4], rb[4];
        auto fetch = [&](int k0, int st) {
            if constexpr (AD)
                da.issue(la + st * OP_FLOATS);
            else
                load_tile<AKC>(A, d.lda, m0, k0, d.M, kend, avec, tid, ra, 0, nullptr, 0);
            if constexpr (BD)
                db.issue(lb + st * OP_FLOATS);
            else if constexpr (AD && !BKC)  // (direct A beside B through the registers: the shifted operand's interior tiles)
                load_tile_shift(Bm, d.ldb, n0, k0, tid, rb, d.shift_t, rcp_t, h0, d.ld_h0);
            else
                load_tile<BKC>(Bm, d.ldb, n0, k0, d.N, kend, bvec, tid, rb, BKC ? 0 : d.shift_t, h0, d.ld_h0);
        };
        auto commit = [&](int st) {
            if constexpr (!AD) store_tile<AKC>(la + st * OP_FLOATS, tid, ra);
            if constexpr (!BD) store_tile<BKC>(lb + st * OP_FLOATS, tid, rb);
        };
        if (lo >= hi) return;
        fetch(lo, 0);
        commit(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the direct loads' data is in LDS once the counter says so)
        __syncthreads();
        int cur = 0;
        for (int k0 = lo; k0 < hi; k0 += TK) {
            const bool more = k0 + TK < hi;
            if (more) fetch(k0 + TK, cur ^ 1);  // (stage cur ^ 1 was read last in the previous iteration: free behind its barrier)
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
                if (!AKC && want_cs) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) cs[t] += (a[t][0] + a[t][1]) + (a[t][2] + a[t][3]);
                }
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][s], b[nt][s], acc[mt][nt], 0, 0, 0);
            }
            if (more) commit(cur ^ 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            cur ^= 1;
        }
    };
    using yes = std::true_type;
    using no = std::false_type;
    if (adir && bdir) {
        loop(yes{}, yes{}, kbeg, kfull);
        loop(no{}, no{}, kfull, kend);  // (a last partial k tile)
    } else if (!BKC && adir && bvec && d.shift_t > 0 && n0 + TN <= d.N && kend <= (1 << 21)) {
        // a weight gradient against the "previous state": the shifted operand through the registers, dy^T direct
        if constexpr (!BKC) {
            loop(yes{}, no{}, kbeg, kfull);
            loop(no{}, no{}, kfull, kend);
        }
    } else {
        loop(no{}, no{}, kbeg, kend);
    }
}

// (waves_per_eu 2: the register allocator must stay within 256 registers per wave, accumulators included -- at 260 the second
// workgroup of a CU is gone and every product of the plan ran 20 % slower)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_kernel(const Batch batch) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [2 stages][A, B][OP_FLOATS] = 64 KB: two workgroups per CU
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
    // (the first tile column of a row of tiles carries them, and of its waves the first wave column)
    const bool want_cs = d.colsum != nullptr && !akc && tn == 0 && ((threadIdx.x >> 6) & 1) == 0;
    const bool tile_cs = d.colsum != nullptr && !akc && tn == 0;
    float cs[2] = {0.f, 0.f};
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

    // (everything the epilogue addresses with is derived from values the compiler cannot see through: it otherwise forms the
    // 32 row pointers of the output tile in front of the k loop and carries them -- or their spill slots -- through it)
    int tid = threadIdx.x, m0e = m0, n0e = n0;
    asm volatile("" : "+v"(tid), "+s"(m0e), "+s"(n0e));
    const int wave = tid >> 6, lane = tid & 63, i = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    if (tile_cs) {
        // lane (i, h) of wave (wm, 0) holds the sums of rows 64 wm + 32 t + i over the k of its half h: the halves meet in
        // LDS (free behind contract()'s last barrier) and are added in h order; a chunk's 128 sums go behind the partial
        // tiles in the workspace, an unsplit product's straight to the output
        if (want_cs) {
#pragma unroll
            for (int t = 0; t < 2; ++t) lds[h * TM + 64 * wm + 32 * t + i] = cs[t];
        }
        __syncthreads();
        if (tid < TM) {
            const float t = lds[tid] + lds[TM + tid];
            if (split > 1) {
                as_global(d.workspace)[(size_t)tiles_m * tiles_n * split * (TM * TN) + ((size_t)tm * split + chunk) * TM + tid] = t;
            } else if (m0e + tid < d.M) {
                d.colsum[m0e + tid] = t;
                if (d.colsum2) d.colsum2[m0e + tid] = t;
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
        const int col = n0e + 64 * wn + 32 * nt + i;
        if (col >= d.N) continue;
        const float bv = d.bias ? bias[col] : 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0e + 64 * wm + 32 * mt + 8 * (r >> 2) + 4 * h + (r & 3);
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
