// Streamed grouped NHWC convolution for gfx950 (see conv_nhwc.hip for the launch side).
//
// A workgroup is PERSISTENT and WAVE-SPECIALISED: 4 contraction waves + 4 loader waves, one of each per SIMD (512
// threads, one workgroup per CU).  It walks its share of the launch's units -- unit = (item, band of 196 output pixels, block of 128 / split
// output channels) -- and sees the input of every unit as a sequence of STAGES of 32 input channels:
//
//     14x14 maps, 28x28 with dilation 1 / 1x1:  stage = one quarter of a 128-channel chunk, all taps
//     28x28 maps with dilation 2 / 4 / 8:        stage = (tap row ky, quarter): the 7 image rows that tap row reads
//
// The loader waves move stages from global memory straight into a RING of LDS slots with 16-byte direct-to-LDS loads
// (global_load_lds_dwordx4: no registers, no ds_write pass), several stages ahead of the contraction waves and ACROSS
// unit boundaries, applies the fused prologue in place in LDS (attention-mask multiply; ReLU-backward gate, whose map
// travels through the ring slot behind its stage) and announces a stage with one s_barrier.  The contraction waves
// therefore never wait for memory: the next unit's first stage is resident when the previous unit's epilogue ends, and a
// stage hand-over is one barrier.  The loader has its own vmcnt queue, so the long-latency tile loads never stand in
// front of the contraction waves' weight stream (loads complete in order per wave).
//
// Inside a workgroup the waves split OUTPUT work only.  Split 1: a workgroup computes the unit's 128 output channels,
// a wave 32 of them (two 16-channel tiles sharing every A fragment) x all 13 m-tiles; split 2 / 4 / 8 (launches that do
// not fill the chip): 64 / 32 / 16 channels per workgroup and one channel tile x 13 / 7 / 4 m-tiles per wave.  Every wave contracts ALL
// input channels of its outputs, so there is no partial-sum exchange and the summation order of an output does not depend
// on the split (all splits are bit-identical).  One wave per SIMD: the MFMA stream of a wave is long runs over 14-26
// independent accumulators with ~0.3 other instructions per MFMA, which a single wave issues without gaps; two waves
// per SIMD (rounds 1-3) ran the pair at 78 % of the pipe and left the early finisher waiting at every barrier.
//
// Taps that fall outside the image contribute nothing: the loader publishes, per tap, which m-tiles have at least one
// pixel whose tap lands inside the staged rows; the contraction waves skip the others (dilation 8 on a 14x14 map: 42 of
// 117 (tap, m-tile) pairs; dilation 4: 21; the reference op is nn.Conv2d(..., dilation=d, padding=d),
// /root/reference/probnmn/modules/nmn_modules.py:146-160).
//
// LDS image of a slot: two SUB-SLOTS, one per 16-channel block kb of the stage, so that the two steps of a tap read at
// one register address + an immediate offset.  In a sub-slot pixel row r = 4 pieces of 16 bytes; piece g -- channels
// [4 g, +4) of the block -- sits at piece position
//     (g & 1) << 1  |  (((r >> 2) & 1) ^ (g >> 1))
// ds_read_b128 is served in groups of 16 lanes (measured in rounds 1-3): eight lanes of lane
// group g (pixels li in {0-3, 12-15}) and eight of g ^ 1 (li in {4-11}), sixteen different pixel rows.  Rows of equal
// r & 3 share banks: of each lane group two lanes fall into one such class, their rows 4 or 12 apart, so (r >> 2) & 1
// separates them and bit 1 separates g from g ^ 1: no bank conflicts for any tap or dilation.  A direct-to-LDS load
// writes lane-linear (base + 16 lane), so the permutation is applied to the SOURCE address: lane l of piece i fetches
// the channels that belong at position l & 3 of row 16 i + (l >> 2).  Rows a tap reads outside the staged region are
// the sub-slot's own eight zero rows (same r & 7, so the bank argument holds).
// The row a lane reads for (tap, output pixel) comes from a per-unit-geometry table [tap][208] of 16-bit row offsets the
// loader computes once (dilation / band changes between units are rare: a launch's items are sorted by weight), so a
// tap costs the contraction waves one ds_read_u16 and one v_xad per m-tile instead of ~10 VALU of border arithmetic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/probnmn_hip.h"
#include "global_ptr.h"

// Cycle accounting (make cycles -> lib/libprobnmn_cycles.so; scripts/r04_cycles.py reads the counters): off in the
// shipped build -- every stamp is an s_memtime the scalar unit waits for.
#ifdef PNMN_STREAM_CYCLES
#define PNMN_CYC() __builtin_readcyclecounter()
#else
#define PNMN_CYC() 0ull
#endif

namespace pnmn {
namespace stream {

constexpr int CB = 128;          // channels of a chunk / of an output block
constexpr int QC = 32;           // channels of a stage
constexpr int NTHREADS = 512;    // 4 contraction waves + 4 loader waves (one of each per SIMD)
constexpr int LOADER_WAVE = 4;   // the first loader wave
constexpr int NLOAD = 4;
constexpr int MTILES = 13;       // 196 output pixels = 12.25 tiles of 16
constexpr int TAB_ROWS = 208;    // table entries per tap (13 tiles x 16 pixels)

using lchar = __attribute__((address_space(3))) char;

template <int H, int W, int TH>
struct Geom {
    static constexpr bool WHOLE = (TH == H);
    static constexpr int NB = H / TH;                                  // bands per item
    static constexpr int RP = (WHOLE ? H : TH + 2) * W;                // pixel rows of the largest staged region
    static constexpr int Z0 = (RP + 7) & ~7;                           // first zero row of a sub-slot
    static constexpr int SUB_PIECES = (Z0 + 15) / 16;                  // 1 KiB wave-loads per sub-slot (16 rows each)
    static constexpr int PIECES = 2 * SUB_PIECES;                      // ... per slot
    static constexpr int SUB_BYTES = (Z0 + 8) * 64;                    // 13 312 (14x14) / 16 896 (28x28)
    static constexpr int SLOT_BYTES = 2 * SUB_BYTES;
    static constexpr int RING = WHOLE ? 6 : 4;
    // which loader wave requests (and prepares) piece i of a sub-slot: every fourth one -- except that a LAST piece which
    // overlaps its predecessor (it starts at Z0 - 16 to end in front of the zero rows) goes to the predecessor's wave:
    // two waves writing the same rows would let one's late load undo the other's prologue.
    static constexpr bool OVERLAP = (Z0 % 16) != 0;
    static constexpr int owner(int i) { return (OVERLAP && i == SUB_PIECES - 1) ? (SUB_PIECES - 2) % NLOAD : i % NLOAD; }
    static constexpr int owned(int lw) {
        int n = 0;
        for (int i = 0; i < SUB_PIECES; ++i) n += owner(i) == lw ? 1 : 0;
        return n;
    }
    static constexpr int TAB_OFF = RING * SLOT_BYTES;                  // uint16 [9][208]
    static constexpr int LDS_BYTES = TAB_OFF + 9 * TAB_ROWS * 2 + 128; // (+ the table rows of the m-tiles past the 13th)
    static_assert(LDS_BYTES <= 160 * 1024, "ring + table must fit the CU's LDS");
    static_assert((Z0 + 7) * 64 + 16 < 65536 && SUB_BYTES < 65536, "table entries and the kb offset are 16 bits");
};

// kernel argument: the launch's segments (conv_plan.h) and the convolution's uniform shape
struct Launch {
    int n_seg;
    int wg_begin[3], split[3], unit0[3], n_units[3], per_xcd[3];
    int wgs_x;       // virtual workgroups per output-channel block (sum over the segments, a multiple of 8)
    int total;       // wgs_x * cout_blocks
    int cin_chunks, ntaps, in_stride, out_stride, relu;
    unsigned long long* dbg;  // (PNMN_CONV_DBGPTR: cycle counters per contraction wave, scripts/r04_cycles.py)
};

__device__ __forceinline__ void lds_barrier() {
    // every LDS access of this wave has completed; no wait on the vector-memory counter (the contraction waves keep
    // their weight prefetch in flight across the hand-over, the loader its tile loads)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// image rows [rs, re) a stage of (band, dilation, pass) holds and the taps [t0, t1) that read them; false: the tap
// row lies wholly outside the image
template <int H, int W, int TH>
__device__ __forceinline__ bool pass_rows(int band, int dil, int npass, int pass, int ntaps, int& rs, int& re, int& t0, int& t1) {
    if (Geom<H, W, TH>::WHOLE) {
        rs = 0, re = H, t0 = 0, t1 = ntaps;
        return true;
    }
    const int y0 = band * TH;
    if (npass == 1) {
        const int halo = (ntaps == 1) ? 0 : 1;
        rs = y0 - halo < 0 ? 0 : y0 - halo;
        re = y0 + TH + halo > H ? H : y0 + TH + halo;
        t0 = 0, t1 = ntaps;
        return true;
    }
    const int a = y0 + (pass - 1) * dil;
    rs = a < 0 ? 0 : (a > H ? H : a);
    re = a + TH < 0 ? 0 : (a + TH > H ? H : a + TH);
    t0 = 3 * pass, t1 = t0 + 3;
    return re > rs;
}

// The (unit, stage) sequence of this workgroup.  Every wave -- and both cursors of the loader -- walks it with the same
// arithmetic, so all agree on the order without exchanging anything.  All members are wave-uniform.
template <int H, int W, int TH>
struct Walker {
    using G = Geom<H, W, TH>;
    const pnmn_conv_item* items;
    int vid, total;          // virtual workgroup id of the current unit (>= total: done)
    int item, band, sub, split, cb;
    int dil, npass, slots;   // of the unit's item; slots per stage: 1, or 2 with a gate map
    int chunk, pass, kq;     // stage cursor
    int rs, re, t0, t1;      // rows / taps of the current stage

    __device__ __forceinline__ bool valid() const { return vid < total; }

    __device__ __forceinline__ bool decode(const Launch& L) {
        // the output-channel blocks of a unit take CONSECUTIVE ids on one XCD (id & 7): they stream the same input, which
        // then comes out of that XCD's L2 for all but the first (block-major ids re-fetched the classifier conv's input
        // from HBM once per block: 841 MB fetched for 103 MB of input at 1024 items, scripts/r04_pmc_dispatch.sh)
        const int cout_blocks = L.total / L.wgs_x;
        const int s8 = vid >> 3;
        const int cbi = s8 % cout_blocks;
        const int x = (s8 / cout_blocks) * 8 + (vid & 7);
        int wb = L.wg_begin[0], sp = L.split[0], u0 = L.unit0[0], nu = L.n_units[0], per = L.per_xcd[0];
        if (L.n_seg > 1 && x >= L.wg_begin[1]) wb = L.wg_begin[1], sp = L.split[1], u0 = L.unit0[1], nu = L.n_units[1], per = L.per_xcd[1];
        if (L.n_seg > 2 && x >= L.wg_begin[2]) wb = L.wg_begin[2], sp = L.split[2], u0 = L.unit0[2], nu = L.n_units[2], per = L.per_xcd[2];
        const int local = x - wb;
        const int slot = local >> 3;
        const int j = slot / sp;
        const int unit = (local & 7) * per + j;   // XCD (vid & 7) takes a contiguous range of the segment's units
        if (unit >= nu || j >= per) return false;
        const int u = u0 + unit;
        item = u / G::NB, band = u % G::NB, sub = slot - j * sp, split = sp, cb = cbi;
        return true;
    }
    __device__ __forceinline__ void open_unit(const Launch& L) {
        while (vid < total && !decode(L)) vid += (int)gridDim.x;
        if (vid >= total) return;
        const pnmn_conv_item* it = items + item;
        dil = it->dilation;
        slots = it->gate ? 2 : 1;
        npass = (G::WHOLE || L.ntaps == 1 || dil == 1) ? 1 : 3;
        chunk = 0, pass = 0, kq = 0;
        while (!pass_rows<H, W, TH>(band, dil, npass, pass, L.ntaps, rs, re, t0, t1)) ++pass;  // (the centre row always exists)
    }
    __device__ __forceinline__ void start(const Launch& L, const pnmn_conv_item* its) {
        items = its, vid = (int)blockIdx.x, total = L.total;
        open_unit(L);
    }
    // next stage of the same unit; false: the unit is finished (the cursor is then undefined until next_unit())
    __device__ __forceinline__ bool next_stage(const Launch& L) {
        if (++kq < 4) return true;
        kq = 0;
        while (++pass < npass)
            if (pass_rows<H, W, TH>(band, dil, npass, pass, L.ntaps, rs, re, t0, t1)) return true;
        if (++chunk >= L.cin_chunks) return false;
        pass = 0;
        while (!pass_rows<H, W, TH>(band, dil, npass, pass, L.ntaps, rs, re, t0, t1)) ++pass;
        return true;
    }
    __device__ __forceinline__ void next_unit(const Launch& L) {
        vid += (int)gridDim.x;
        open_unit(L);
    }
    __device__ __forceinline__ int cbase() const { return chunk * CB + kq * QC; }  // first input channel of the stage
};

// ------------------------------------------------------------------------------------------------------------------
// loader wave
// ------------------------------------------------------------------------------------------------------------------

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// one stage (32 channels of the rows [rs, re) of one source map) into the slot at LDS byte address `slot`.
// The loader shares a SIMD with a contraction wave that keeps the matrix pipe busy, and there every VECTOR-ALU
// instruction of the loader waits for a gap between MFMAs (measured: ~150 cycles per instruction of the first version,
// which formed a 64-bit address per piece and lane).  So the per-lane part of the address is formed ONCE: the channel
// permutation of a lane does not depend on the piece (a piece starts on a multiple of 8 rows, so (row >> 2) & 1 is
// (lane >> 4) & 1), and a piece advances a wave-uniform base on the scalar unit.
template <int H, int W, int TH, int LW>
__device__ __forceinline__ void issue_rows(const float* src, int in_stride, int rs, int re, lchar* slot, int lane) {
    using G = Geom<H, W, TH>;
    const int nr = (re - rs) * W;
    using gchar = __attribute__((address_space(1))) char;
    // (the pointer comes out of an item record: uniform, but only the scalar unit can be told so)
    const uint64_t src64 = (uint64_t)(uintptr_t)src + (uint64_t)rs * W * in_stride * 4;
    const uint64_t base64 = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(src64 >> 32)) << 32) |
                            (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)src64);
    const gchar* base = (const gchar*)base64;
    const int s = lane & 3, lrow = lane >> 2;
    const int g = (((s & 1) ^ ((lane >> 4) & 1)) << 1) | (s >> 1);
    const uint32_t voff = (uint32_t)(lrow * in_stride + g * 4) * 4u;  // this lane's row and channels within a piece
    // the one piece that straddles the end of the region: its rows past the end re-read the last row (never addressed)
    const int edge = (nr & 15) && nr < G::Z0 ? (nr < G::Z0 - 16 ? nr & ~15 : G::Z0 - 16) : -1;
    const uint32_t eoff = edge < 0 ? voff : (uint32_t)(((edge + lrow < nr ? edge + lrow : nr - 1) - edge) * in_stride + g * 4) * 4u;
#pragma unroll
    for (int i = 0; i < G::SUB_PIECES; ++i) {
        if (G::owner(i) != LW) continue;  // (this loader wave's pieces)
        // Every lane of a direct-to-LDS load writes its 16 bytes, whatever EXEC says.  So every piece is 16 full rows,
        // and the last one starts early enough to end in front of the zero rows, overlapping its predecessor.
        const int r0 = i * 16 < G::Z0 - 16 ? i * 16 : G::Z0 - 16;
        const uint32_t off = r0 == edge ? eoff : voff;
        const gchar* pb = base + (size_t)(r0 < nr ? r0 : 0) * in_stride * 4;  // (uniform; pieces past the region: any rows)
        __builtin_amdgcn_global_load_lds((const gfloat*)(pb + off), slot + r0 * 64, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const gfloat*)(pb + (off + 64u)), slot + G::SUB_BYTES + r0 * 64, 16, 0, 0);
    }
}

// prologue in place: x *= mask[pixel], x = gate > 0 ? x : 0
template <int H, int W, int TH, int LW>
__device__ __forceinline__ void fixup(char* lds, int slot_x, int slot_g, const float* mask, int rs, int re, bool gated, int lane) {
    using G = Geom<H, W, TH>;
    const int nr = (re - rs) * W;
    const gfloat* msrc = mask ? as_global(mask) + rs * W : nullptr;
    // the mask values of all pieces are requested before the first is used (one round trip, not SUB_PIECES of them:
    // the first version kept the contraction waves waiting 18 % of their time at the hand-over)
    // this loader wave's pieces (Geom::owner); all their mask values are requested before the first is used
    float mk[G::SUB_PIECES];
#pragma unroll
    for (int i = 0; i < G::SUB_PIECES; ++i) {
        if (G::owner(i) != LW) continue;
        const int row = i * 16 + (lane >> 2);
        mk[i] = msrc ? msrc[row < nr ? row : nr - 1] : 1.f;
    }
    // one sub-slot at a time, all of its pieces read before the first is written back (LDS round trips in flight
    // instead of one after the other; the loaders have the registers)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        f32x4 v[G::SUB_PIECES], gt[G::SUB_PIECES];
#pragma unroll
        for (int i = 0; i < G::SUB_PIECES; ++i) {
            if (G::owner(i) != LW) continue;
            const int off = kb * G::SUB_BYTES + i * 1024 + lane * 16;
            v[i] = *reinterpret_cast<const f32x4*>(lds + slot_x + off);
            if (gated) gt[i] = *reinterpret_cast<const f32x4*>(lds + slot_g + off);
        }
#pragma unroll
        for (int i = 0; i < G::SUB_PIECES; ++i) {
            if (G::owner(i) != LW) continue;
            const int off = kb * G::SUB_BYTES + i * 1024 + lane * 16;
            f32x4 x = v[i] * mk[i];
            if (gated) {
                x.x = gt[i].x > 0.f ? x.x : 0.f;
                x.y = gt[i].y > 0.f ? x.y : 0.f;
                x.z = gt[i].z > 0.f ? x.z : 0.f;
                x.w = gt[i].w > 0.f ? x.w : 0.f;
            }
            if (i * 16 + (lane >> 2) < G::Z0) *reinterpret_cast<f32x4*>(lds + slot_x + off) = x;
        }
    }
}

// row table of a unit geometry (band, dilation): wave `w` of `nw` fills its share (one m-tile = 16 lanes)
template <int H, int W, int TH>
__device__ __forceinline__ void fill_table(char* lds, int band, int dil, int npass, int ntaps, int lane, int w, int nw) {
    using G = Geom<H, W, TH>;
    constexpr int HW = TH * W;
    uint16_t* tab = reinterpret_cast<uint16_t*>(lds + G::TAB_OFF);
    const int y0 = G::WHOLE ? 0 : band * TH;
    const int ntile = ntaps * MTILES;
    for (int t4 = 4 * w; t4 < ntile; t4 += 4 * nw) {  // four m-tiles per pass of a wave
        const int tile = t4 + (lane >> 4);
        const int tap = tile / MTILES;
        const int m = tile - tap * MTILES;
        const int p = m * 16 + (lane & 15);
        int rs, re, t0, t1;
        const int pass = npass == 1 ? 0 : tap / 3;
        const bool rows_ok = tile < ntile && pass_rows<H, W, TH>(band, dil, npass, pass, ntaps, rs, re, t0, t1);
        int dy = 0, dx = 0;
        if (ntaps == 9) dy = (tap / 3 - 1) * dil, dx = (tap % 3 - 1) * dil;
        const int yy = y0 + p / W + dy, xx = p % W + dx;
        const bool ok = rows_ok && p < HW && yy >= rs && yy < re && (unsigned)xx < (unsigned)W;
        const int qv = rows_ok ? (yy - rs) * W + xx : p;
        const int row = ok ? qv : G::Z0 + (qv & 7);
        if (tile < ntile) tab[tap * TAB_ROWS + p] = (uint16_t)(row * 64 + ((row >> 2) & 1) * 16);
    }
}

// Start-up work every wave shares: the zero rows of every slot (no load ever writes them) and the row table of the first
// unit (later geometries are the loaders', behind a unit's end).  Ends at the workgroup's first barrier.
template <int H, int W, int TH>
__device__ __forceinline__ void start_up(char* lds, const Walker<H, W, TH>& first, int ntaps, int wave, int lane) {
    using G = Geom<H, W, TH>;
    for (int t = wave * 64 + lane; t < G::RING * 64; t += NTHREADS) {  // (8 rows x 64 bytes per sub-slot: 32 lanes each)
        const int sl = t >> 6, kb = (t >> 5) & 1;
        *reinterpret_cast<f32x4*>(lds + sl * G::SLOT_BYTES + kb * G::SUB_BYTES + G::Z0 * 64 + (t & 31) * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (first.valid()) fill_table<H, W, TH>(lds, first.band, first.dil, first.npass, ntaps, lane, wave, NTHREADS / 64);
    lds_barrier();
}

template <int H, int W, int TH, int LW>
__device__ __forceinline__ void loader(const Launch& L, const pnmn_conv_item* items, char* lds, int lane) {
    using G = Geom<H, W, TH>;
    // direct-to-LDS loads of THIS wave per slot.  (One loader wave could not feed the ring: a wave keeps few such loads
    // in flight -- 250-650 cycles per load at the issue, measured -- so four of them, one per SIMD, take every fourth
    // piece each; a wave prepares exactly the pieces it requested, so its own vmcnt covers them.)
    constexpr int NP2 = 2 * G::owned(LW);
    lchar* ring = (lchar*)lds;
    // The loader shares its SIMD with a contraction wave whose MFMA stream would otherwise win nearly every issue
    // slot: at the default priority issuing one stage (26 loads and their addresses) took as long as contracting it.
    __builtin_amdgcn_s_setprio(3);
    Walker<H, W, TH> P, C;   // producer (issue) and consumer (hand-over) cursors
    P.start(L, items);
    C = P;
    int issued = 0, freed = 0, cstart = 0;  // in slots, cumulative
    const pnmn_conv_item* pit = P.valid() ? items + P.item : nullptr;
    // `limit`: slots that may be in use.  Before the first hand-over only the first stage is requested: the contraction
    // waves stand at its barrier, and everything issued in front of it delays it (launches of one unit per workgroup
    // -- the deep program levels -- spent a quarter of their time there).
    auto top_up = [&](int limit) {
        while (P.valid() && issued - freed + P.slots <= limit) {
            const float* src = (pit->in2 != nullptr && P.chunk > 0) ? pit->in2 + P.kq * QC : pit->in + P.cbase();
            issue_rows<H, W, TH, LW>(src, L.in_stride, P.rs, P.re, ring + (issued % G::RING) * G::SLOT_BYTES, lane);
            if (P.slots == 2) {
                issue_rows<H, W, TH, LW>(pit->gate + P.cbase(), L.in_stride, P.rs, P.re,
                                     ring + ((issued + 1) % G::RING) * G::SLOT_BYTES, lane);
            }
            issued += P.slots;
            if (!P.next_stage(L)) {
                P.next_unit(L);
                pit = P.valid() ? items + P.item : nullptr;
            }
        }
    };
    bool unit_end = false;   // the contraction waves stand (or will stand) at the end-of-unit barrier
    unsigned long long lc[4] = {0, 0, 0, 0};  // cycles: issuing, waiting for loads, prologue in place, at barriers
    unsigned long long lt = PNMN_CYC();
    auto lap = [&](int k) {
        const unsigned long long now = PNMN_CYC();
        lc[k] += now - lt;
        lt = now;
    };
    // the consumer's stage: its loads (and the gate map's) have landed -- everything issued up to its end -- and the
    // prologue is applied in place
    auto prepare = [&] {
        const pnmn_conv_item* cit = items + C.item;
        const int younger = issued - (cstart + C.slots);
        if (younger == 0) wait_vm<0>();
        else if (younger == 1) wait_vm<NP2>();
        else if (younger == 2) wait_vm<2 * NP2>();
        else wait_vm<3 * NP2>();
        lap(1);
        if (cit->mask != nullptr || C.slots == 2)
            fixup<H, W, TH, LW>(lds, (cstart % G::RING) * G::SLOT_BYTES, ((cstart + 1) % G::RING) * G::SLOT_BYTES, cit->mask, C.rs,
                            C.re, C.slots == 2, lane);
        lap(2);
    };
    // the first stage is requested BEFORE the start-up work: zero rows and table are written while it travels
    // (measured, round 5: requesting the SECOND stage up front too changes nothing, not even for the split-14 / 26
    // units that contract a stage in 1-2 us -- 18-19 us for a 7-item launch either way)
    top_up(P.valid() ? P.slots : 0);
    int tab_band = C.valid() ? C.band : -1, tab_dil = C.valid() ? C.dil : -1;
    start_up<H, W, TH>(lds, C, L.ntaps, LOADER_WAVE + LW, lane);
    lap(0);
    bool prepared = false;
    while (C.valid()) {
        if (!prepared) prepare();  // (only the first stage, and the second: requested behind the first hand-over)
        if (unit_end) {  // the previous unit's contraction is over: its table may go
            lds_barrier();
            unit_end = false;
            lap(3);
        }
        if (C.band != tab_band || C.dil != tab_dil) {
            fill_table<H, W, TH>(lds, C.band, C.dil, C.npass, L.ntaps, lane, LW, NLOAD);
            tab_band = C.band, tab_dil = C.dil;
        }
        lap(2);
        lds_barrier();        // hand-over: the stage is the contraction waves'; every stage before it is finished
        lap(3);
        freed = cstart;
        cstart += C.slots;
        if (!C.next_stage(L)) {
            C.next_unit(L);
            unit_end = true;
        }
        // The NEXT stage is prepared before anything new is requested: its loads went out a stage ago and have landed,
        // whereas behind a fresh request the in-place prologue -- whose LDS reads the compiler orders behind every
        // outstanding direct-to-LDS load -- would wait a full memory round trip per stage.
        prepared = C.valid() && issued >= cstart + C.slots;
        if (prepared) prepare();
        top_up(G::RING);
        lap(0);
    }
    if (unit_end) lds_barrier();
    if (L.dbg && lane == 0 && LW == 0) {
        unsigned long long* d = L.dbg + ((size_t)gridDim.x * 4 + blockIdx.x) * 8;
        d[0] = lc[0], d[1] = lc[1], d[2] = lc[2], d[3] = lc[3];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// contraction waves
// ------------------------------------------------------------------------------------------------------------------

// acc += a (x) b, IN PLACE.  The builtin leaves vDst free to differ from SrcC, and under this kernel's pinned order the
// register allocator then rotates the 13 accumulators through spare registers and spills them (measured: 25 scratch
// accesses per tap at a 256-register budget for ~160 live values).  The tied operand takes that freedom away.  No MFMA
// result is ever an A / B operand, dependent MFMAs are >= 2 issue slots apart (40 cycles dependent latency, 32 issue),
// and the compiler still sees the register dependencies on the loaded fragments (it inserts the s_waitcnt).
__device__ __forceinline__ void mfma(f32x4& acc, float a, float b) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}


// Epilogue of one wave: lane holds output channels n0 + 4g .. +3 of pixel (mbase + j) * 16 + li of its m-tiles.
// One straight-line variant per kind of item, chosen ONCE (uniform): on gfx9 stores count on vmcnt like loads, so a
// shared body -- "add the previous contents, which may or may not have been loaded" -- made every tile's store wait for
// the store of the tile before it: 13 write round trips in a row, 7 % of a unit.
template <int H, int W, int TH, int MTW>
__device__ __forceinline__ void epilogue(const pnmn_conv_item& it, const f32x4* acc, int mbase, int n0, int band, int out_stride,
                                         int relu, int lane, const f32x4 bias4) {
    constexpr int HW = TH * W;
    // (the pixel offsets below depend on nothing the contraction computes: left visible, the compiler forms all of them
    // BEFORE the stage loop, spills them across it and reloads them one scratch round trip at a time -- 27 000 cycles
    // per unit with two channel tiles.  An opaque lane id keeps them here.)
    asm volatile("" : "+v"(lane));
    const int li = lane & 15, g = lane >> 4;
    const int p_img = Geom<H, W, TH>::WHOLE ? 0 : band * TH * W;
    auto act = [&](f32x4 v) {
        v += bias4;
        if (relu) {
            v.x = fmaxf(v.x, 0.f);
            v.y = fmaxf(v.y, 0.f);
            v.z = fmaxf(v.z, 0.f);
            v.w = fmaxf(v.w, 0.f);
        }
        return v;
    };
    float* const out0 = it.out + (size_t)p_img * out_stride + n0 + 4 * g;
    if (!(it.flags & (PNMN_CONV_ACCUMULATE | PNMN_CONV_MASKBWD | PNMN_CONV_DATTN))) {
        // plain store
#pragma unroll
        for (int j = 0; j < MTW; ++j) {
            const int p = (mbase + j) * 16 + li;
            if (p < HW) store4(as_global(out0 + (size_t)p * out_stride), act(acc[j]));
        }
        return;
    }
    if (it.flags & PNMN_CONV_DATTN) {
        // dx is stored; d(attention)[p] += sum_c dx[p][c] * feats[p][c] (this wave's 16 channels: four lane groups of
        // four channels, then one atomic per pixel and wave).  mb_attn == nullptr: the all-ones attention, no gradient.
        const bool dattn = it.mb_attn != nullptr;
        f32x4 fv[MTW];
#pragma unroll
        for (int j = 0; j < MTW; ++j) {
            const int p = (mbase + j) * 16 + li;
            fv[j] = (dattn && p < HW) ? load4(as_global(it.mb_feats) + (size_t)(p_img + p) * CB + n0 + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < MTW; ++j) {
            const int p = (mbase + j) * 16 + li;
            if (p < HW) store4(as_global(out0 + (size_t)p * out_stride), act(acc[j]));
        }
        if (dattn) {
#pragma unroll
            for (int j = 0; j < MTW; ++j) {
                const int p = (mbase + j) * 16 + li;
                const f32x4 v = acc[j];
                float part = (p < HW) ? v.x * fv[j].x + v.y * fv[j].y + v.z * fv[j].z + v.w * fv[j].w : 0.f;
                part += __shfl_xor(part, 16);
                part += __shfl_xor(part, 32);
                if (p < HW && g == 0) unsafeAtomicAdd(it.mb_dattn + p_img + p, part);
            }
        }
        return;
    }
    if (it.flags & PNMN_CONV_MASKBWD) {
        // fused backward of (feats * attn): dfeats += dx * attn, dattn += sum_c dx * feats
        const bool sole = it.flags & PNMN_CONV_MB_SOLE;
        const gfloat* attn = as_global(it.mb_attn);
        float am[MTW];
        f32x4 fv[MTW], dold[MTW];
#pragma unroll
        for (int j = 0; j < MTW; ++j) {
            const int p = (mbase + j) * 16 + li;
            const bool ok = p < HW;
            am[j] = (ok && attn) ? attn[p_img + p] : 1.f;
            fv[j] = (ok && attn) ? load4(as_global(it.mb_feats) + (size_t)(p_img + p) * CB + n0 + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
            dold[j] = (ok && sole) ? load4(as_global(it.mb_dfeats) + (size_t)(p_img + p) * CB + n0 + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < MTW; ++j) {
            const int p = (mbase + j) * 16 + li;
            const bool ok = p < HW;
            const f32x4 v = acc[j];
            float part = v.x * fv[j].x + v.y * fv[j].y + v.z * fv[j].z + v.w * fv[j].w;
            if (ok) {
                float* d = it.mb_dfeats + (size_t)(p_img + p) * CB + n0 + 4 * g;
                if (sole) {  // only this wave touches these 4 channels of pixel p
                    store4(as_global(d), dold[j] + v * am[j]);
                } else {
                    unsafeAtomicAdd(d + 0, v.x * am[j]);
                    unsafeAtomicAdd(d + 1, v.y * am[j]);
                    unsafeAtomicAdd(d + 2, v.z * am[j]);
                    unsafeAtomicAdd(d + 3, v.w * am[j]);
                }
            }
            if (attn) {
                part += __shfl_xor(part, 16);
                part += __shfl_xor(part, 32);
                if (ok && g == 0) unsafeAtomicAdd(it.mb_dattn + p_img + p, part);
            }
        }
        return;
    }
    if (it.flags & PNMN_CONV_ATOMIC) {
#pragma unroll
        for (int j = 0; j < MTW; ++j) {
            const int p = (mbase + j) * 16 + li;
            if (p < HW) {
                const f32x4 v = act(acc[j]);
                float* d = out0 + (size_t)p * out_stride;
                unsafeAtomicAdd(d + 0, v.x);
                unsafeAtomicAdd(d + 1, v.y);
                unsafeAtomicAdd(d + 2, v.z);
                unsafeAtomicAdd(d + 3, v.w);
            }
        }
        return;
    }
    // out += result: every previous value requested before the first store
    f32x4 old[MTW];
#pragma unroll
    for (int j = 0; j < MTW; ++j) {
        const int p = (mbase + j) * 16 + li;
        old[j] = p < HW ? load4(as_global(out0 + (size_t)p * out_stride)) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < MTW; ++j) {
        const int p = (mbase + j) * 16 + li;
        if (p < HW) store4(as_global(out0 + (size_t)p * out_stride), act(acc[j]) + old[j]);
    }
}

// Plain-store epilogue of a wave with TWO channel tiles (split 1): the tiles are the two 64-byte halves of one 128-byte
// line per pixel, stored back to back.  Tile after tile (all of tile 0's m-tiles, then tile 1's) the halves of a line
// reach L2 hundreds of cycles apart, and per-dispatch PMC (scripts/r04_pmc_dispatch.sh) showed such launches writing
// 2.5-2.75x and fetching up to 4x their algorithmic bytes.
template <int H, int W, int TH, int MTW>
__device__ __forceinline__ void epilogue_plain_pair(const pnmn_conv_item& it, const f32x4* acc0, const f32x4* acc1, int mbase, int n0,
                                                    int band, int out_stride, int relu, int lane, const f32x4 bias0, const f32x4 bias1) {
    constexpr int HW = TH * W;
    asm volatile("" : "+v"(lane));  // (see epilogue())
    const int li = lane & 15, g = lane >> 4;
    const int p_img = Geom<H, W, TH>::WHOLE ? 0 : band * TH * W;
    auto act = [&](f32x4 v, const f32x4 b) {
        v += b;
        if (relu) {
            v.x = fmaxf(v.x, 0.f);
            v.y = fmaxf(v.y, 0.f);
            v.z = fmaxf(v.z, 0.f);
            v.w = fmaxf(v.w, 0.f);
        }
        return v;
    };
    float* const out0 = it.out + (size_t)p_img * out_stride + n0 + 4 * g;
#pragma unroll
    for (int j = 0; j < MTW; ++j) {
        const int p = (mbase + j) * 16 + li;
        const f32x4 v0 = act(acc0[j], bias0), v1 = act(acc1[j], bias1);
        if (p < HW) {
            store4(as_global(out0 + (size_t)p * out_stride), v0);
            store4(as_global(out0 + (size_t)p * out_stride + 16), v1);
        }
        __builtin_amdgcn_sched_barrier(0);  // (left alone, the compiler issues all of tile 0's stores, then all of tile 1's)
    }
}

// One unit at SPLIT workgroups per 128-channel block: all its stages, then the epilogue.  Leaves the walker at the next
// unit.  `cstart`: cumulative ring slots consumed by this workgroup so far (the same count the loader keeps).
//   SPLIT 1: a wave owns 32 output channels (two 16-channel tiles that share every A fragment and table row) x 13 m-tiles
//   SPLIT 2 / 4 / 8: one channel tile x 13 / 7 / 4 m-tiles
// KIND (chosen per UNIT, so that no two bodies meet inside the stage loop -- merging them there made the register
// allocator shuffle and spill accumulators):
//   0  every tap contracts all of the wave's m-tiles
//   1  dilation 8 on a 14x14 map: the taps of row -8 see only the m-tiles [7, 13), those of row +8 only [0, 6)
//      -- 39 of 117 (tap, m-tile) pairs are skipped (the zero rows would contribute exact zeros)
//   2  1x1 convolution: a stage is ONE tap (two steps); the weight sets rotate per stage
template <int H, int W, int TH, int SPLIT, int KIND>
__device__ __forceinline__ void run_unit(Walker<H, W, TH>& Wk, const Launch& L, char* lds, int& cstart, int wave, int lane,
                                         unsigned long long (&cyc)[4]) {
    using G = Geom<H, W, TH>;
    using std::integral_constant;
    const unsigned long long c_unit = PNMN_CYC();
    static_assert(SPLIT == 1 || SPLIT == 2 || SPLIT == 4 || SPLIT == 6 || SPLIT == 8 || SPLIT == 14 || SPLIT == 26,
                  "workgroups per 128-channel block");
    constexpr int NW = SPLIT == 1 ? 2 : 1;                        // 16-channel output tiles of a wave
    constexpr int MW = SPLIT <= 2 ? 1 : SPLIT / 2;                // waves that share a channel tile's m-tiles
    constexpr int MTW = (MTILES + MW - 1) / MW;                   // m-tiles per wave: 13 / 13 / 7 / 5 / 4 / 2 / 1
    constexpr int MH = (MTW + 1) / 2;
    constexpr int WSETS = NW == 2 ? 2 : 3;                        // weight sets in flight (taps ahead + 1)
    constexpr uint32_t FULL = (1u << MTW) - 1u;
    const int li = lane & 15, g = lane >> 4;
    // SPLIT 6 / 14 / 26 (round 5): the block's 8 channel tiles x 3 / 7 / 13 m-parts of 5 / 2 / 1 tiles are dealt to the
    // waves of the unit's workgroups in order, so a workgroup's waves straddle channel tiles -- a launch of 39 items fills
    // the chip with 234 workgroups of 5 tile-times where split 4 leaves 100 CUs idle for 7 and split 8 needs two rounds
    // of 4; the 7- and 14-item launches of the deep program levels (36 of a 1024-question step's 89 conv launches, 27 us
    // each at split 8: 17 us of MFMAs behind ~8 us of start-up, first-stage latency and epilogue) take 1 and 2.
    constexpr bool DEALT = (MW & (MW - 1)) != 0;
    const int gw = Wk.sub * 4 + wave;                             // (DEALT) this wave among the unit's waves
    const int nt = DEALT ? 0 : wave % (4 / (DEALT ? 1 : MW));     // which of the workgroup's wave-sized channel groups
    const int mbase = DEALT ? (gw % MW) * MTW : (wave / (4 / (DEALT ? 1 : MW))) * MTW;  // first m-tile of this wave
    const pnmn_conv_item it = Wk.items[Wk.item];
    // this wave's first output channel
    const int n0 = DEALT ? Wk.cb * CB + (gw / MW) * 16 : Wk.cb * CB + Wk.sub * (CB / SPLIT) + nt * 16 * NW;
    const int band = Wk.band;
    const int cin_total = L.cin_chunks * CB;
    const int slots = Wk.slots;

    f32x4 acc[NW][MTW];
#pragma unroll
    for (int n = 0; n < NW; ++n)
#pragma unroll
        for (int j = 0; j < MTW; ++j) acc[n][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // weight row of this lane: output channel n0 + li (+ 16 for the second tile), channels 4g.. of each 16-block
    constexpr int NTAPS = KIND == 2 ? 1 : 9;
    const gfloat* wrow = as_global(it.weight) + (size_t)(n0 + li) * NTAPS * cin_total + 4 * g;
    const size_t wtile = (size_t)16 * NTAPS * cin_total;
    const uint32_t gconst = (uint32_t)((g >> 1) * 16 | (g & 1) * 32);
    uint32_t tab_lane = (uint32_t)(G::TAB_OFF + (mbase * 16 + li) * 2);
    asm volatile("" : "+v"(tab_lane));  // (keep the table base in a register: the per-tile offsets are immediates)
    // tiles of this wave that exist at all (the 14th..16th lie outside the band)
    const uint32_t exist = (mbase >= MTILES) ? 0u : (MTILES - mbase >= MTW ? FULL : ((1u << (MTILES - mbase)) - 1u));

    // Weights: per channel tile two 16-byte pieces per lane and tap (the two 16-channel blocks of the stage), requested
    // WSETS - 1 taps ahead into a ring of sets (a lone wave per SIMD has only its own MFMAs to hide an L2 / MALL round
    // trip behind: a tap is 3 300 cycles with one channel tile, 6 700 with two).  A stage holds 9 or 3 taps and starts
    // at tap 0 / 3 / 6; with three sets, set = tap % 3; with two, the taps of a unit alternate (an odd stage flips the
    // parity, so the stage loop is unrolled by two).
    f32x4 bias4[NW];
#pragma unroll
    for (int n = 0; n < NW; ++n) {
        bias4[n] = f32x4{0.f, 0.f, 0.f, 0.f};  // (requested now: the epilogue is one memory round trip shorter)
        if (it.bias) bias4[n] = load4(as_global(it.bias) + n0 + 16 * n + 4 * g);
    }
    f32x4 wq[WSETS][NW][2];
    auto wload = [&](f32x4 (&dst)[NW][2], const gfloat* p) {
#pragma unroll
        for (int n = 0; n < NW; ++n) {
            dst[n][0] = load4(p + n * wtile);
            dst[n][1] = load4(p + n * wtile + 16);
        }
    };
    {
        const gfloat* w0 = wrow + (size_t)Wk.t0 * cin_total + Wk.cbase();
        wload(wq[0], w0);
        if constexpr (WSETS == 3) wload(wq[1], KIND == 2 ? w0 + QC : w0 + cin_total);  // (the next tap / the next stage)
    }

    uint32_t rb[MTW];
    f32x4 afrag[MTW];
    uint32_t slot_base = 0;
    // (rb holds ABSOLUTE LDS byte addresses -- the ring's base folded into slot_base -- so that a fragment read is one
    // ds_read_b128 with an immediate offset, without a per-read add of the workgroup's LDS base)
    using lf32x4 = __attribute__((address_space(3))) f32x4;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lchar*)lds;
    auto row_fetch = [&](int j, const char* tp) { rb[j] = *reinterpret_cast<const uint16_t*>(tp + j * 32); };
    auto row_apply = [&](int j) { rb[j] = (rb[j] ^ gconst) + slot_base; };
    auto frag_load = [&](int j, int kb) { afrag[j] = *reinterpret_cast<const lf32x4*>((uintptr_t)(rb[j] + kb * G::SUB_BYTES)); };

    // One half (tiles [lo, hi)) of a step, hand-ordered for a lone wave: the k-slices 0..2 of all its tiles with one
    // FILLER instruction behind each MFMA, then k-slice 3, where every tile's fragment register is re-loaded for the
    // NEXT step right behind its last reader.  k-slices outermost: consecutive MFMAs never share an accumulator (40
    // cycles dependent, 32 issue).  DO = false: the half's tiles see nothing of this tap (zero rows): fillers and
    // reloads only.  (Fillers are not free for a lone wave -- ~5 cycles each, measured -- which is what the second
    // channel tile per wave buys back: the same fragments and rows feed twice the MFMAs.)
    auto half = [&](auto DO, auto LO, auto HI, const f32x4 (&bw)[NW][2], auto KB, auto NFILL, auto&& fill, auto&& reload) {
        constexpr int lo = decltype(LO)::value, hi = decltype(HI)::value, nfill = decltype(NFILL)::value, kb = decltype(KB)::value;
        constexpr int per = (hi - lo) * NW;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int j = lo; j < hi; ++j)
#pragma unroll
                for (int n = 0; n < NW; ++n) {
                    if (decltype(DO)::value) mfma(acc[n][j], bw[n][kb][c], afrag[j][c]);
                    if (c * per + (j - lo) * NW + n < nfill) fill(c * per + (j - lo) * NW + n);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
        for (int j = lo; j < hi; ++j) {
#pragma unroll
            for (int n = 0; n < NW; ++n)
                if (decltype(DO)::value) mfma(acc[n][j], bw[n][kb][3], afrag[j][3]);
            reload(j);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int k = 3 * per; k < nfill; ++k) fill(k);
        __builtin_amdgcn_sched_barrier(0);
    };
    // One tap = two steps (kb = 0, 1).  HALVES: 0 both halves' MFMAs, 1 the first half's only, 2 the second's only.
    // SET / FAR: the weight sets of this tap and of the request it issues (`wfar`).  The fragments of the NEXT tap
    // are requested through its table row `tp` (after a stage's last tap they are discarded: the next stage's slot
    // may not be read before its barrier).
    auto tap_body = [&](auto HALVES, auto SET, auto FAR, const char* tp, const gfloat* wfar) {
        constexpr int hv = decltype(HALVES)::value;
        constexpr int set = decltype(SET)::value, far = decltype(FAR)::value;
        using D0 = integral_constant<bool, hv == 0 || hv == 1>;
        using D1 = integral_constant<bool, hv == 0 || hv == 2>;
        using LO0 = integral_constant<int, 0>;
        using MID = integral_constant<int, MH>;
        using END = integral_constant<int, MTW>;
        using K0 = integral_constant<int, 0>;
        using K1 = integral_constant<int, 1>;
        // step kb = 0; fillers: the far weights' loads
        auto fill0 = [&](int i) { wq[far][i >> 1][i & 1] = load4(wfar + (i >> 1) * wtile + (i & 1) * 16); };
        auto nofill = [&](int) {};
        auto reload1 = [&](int j) { frag_load(j, 1); };
        half(D0{}, LO0{}, MID{}, wq[set], K0{}, integral_constant<int, 2 * NW>{}, fill0, reload1);
        half(D1{}, MID{}, END{}, wq[set], K0{}, integral_constant<int, 0>{}, nofill, reload1);
        // step kb = 1; fillers: the next tap's table rows fetched, then applied (every load through the old rows has
        // been issued; a fetched row has a k-slice of MFMAs to arrive before it is applied)
        auto fill1 = [&](int i) {
            if (i < MTW) row_fetch(i, tp);
            else row_apply(i - MTW);
        };
        auto reload0 = [&](int j) { frag_load(j, 0); };
        constexpr int cap1 = 3 * MH * NW;  // filler places of the first half
        static_assert(2 * MTW <= cap1 || NW == 1, "the first half must hold the row fillers");
        half(D0{}, LO0{}, MID{}, wq[set], K1{}, integral_constant<int, (2 * MTW < cap1 ? 2 * MTW : cap1)>{}, fill1, reload0);
        half(D1{}, MID{}, END{}, wq[set], K1{}, integral_constant<int, (2 * MTW > cap1 ? 2 * MTW - cap1 : 0)>{},
             [&](int i) { fill1(i + cap1); }, reload0);
    };
    using I0 = integral_constant<int, 0>;
    using I1 = integral_constant<int, 1>;
    using I2 = integral_constant<int, 2>;

    bool more = true;
    // one stage: PAR (two weight sets only) = parity of the stage's first tap in the unit's tap order
    auto stage = [&](auto PAR) {
        constexpr int par = decltype(PAR)::value;
        const int t0 = Wk.t0, t1 = Wk.t1, cb0 = Wk.cbase();
        slot_base = lds0 + (uint32_t)((cstart % G::RING) * G::SLOT_BYTES);
        cstart += slots;
        more = Wk.next_stage(L);  // (the cursor now names the NEXT stage: its first weights are requested below)
        // first tap of the next stage (the unit's last stage re-requests its own: never used)
        const gfloat* wnext_stage = more ? wrow + (size_t)Wk.t0 * cin_total + Wk.cbase() : wrow + (size_t)t0 * cin_total + cb0;
        // weights of the tap `d` behind `tap` in the unit's order (the unit's last taps re-request their own)
        auto far_ptr = [&](int tap, int d) -> const gfloat* {
            if (tap + d < t1) return wrow + (size_t)(tap + d) * cin_total + cb0;
            return more ? wnext_stage + (size_t)(tap + d - t1) * cin_total : wrow + (size_t)tap * cin_total + cb0;
        };
        // table row of the tap behind `tap` (clamped: the fragments requested behind a stage's last tap are discarded)
        auto next_row = [&](int tap) -> const char* { return lds + tab_lane + (tap + 1 < NTAPS ? tap + 1 : NTAPS - 1) * (TAB_ROWS * 2); };

        {
            const unsigned long long c0 = PNMN_CYC();
            lds_barrier();  // the stage is resident (and the table, on a unit's first stage)
            cyc[0] += PNMN_CYC() - c0;
        }
#pragma unroll
        for (int j = 0; j < MTW; ++j) row_fetch(j, lds + tab_lane + t0 * (TAB_ROWS * 2));
#pragma unroll
        for (int j = 0; j < MTW; ++j) row_apply(j);
#pragma unroll
        for (int j = 0; j < MTW; ++j) frag_load(j, 0);
        if constexpr (KIND == 2) {
            // the stage's one tap uses set `par` and requests the stage WSETS - 1 ahead (the unit's last ones their own)
            using SA = integral_constant<int, par>;
            using SB = integral_constant<int, (par + WSETS - 1) % WSETS>;
            const int ahead = cb0 + (WSETS - 1) * QC;
            tap_body(I0{}, SA{}, SB{}, next_row(0), wrow + (ahead < cin_total ? ahead : cb0));
        } else if constexpr (WSETS == 3) {
            if (KIND == 1) {
                tap_body(I2{}, I0{}, I2{}, next_row(0), far_ptr(0, 2));
                tap_body(I2{}, I1{}, I0{}, next_row(1), far_ptr(1, 2));
                tap_body(I2{}, I2{}, I1{}, next_row(2), far_ptr(2, 2));
                tap_body(I0{}, I0{}, I2{}, next_row(3), far_ptr(3, 2));
                tap_body(I0{}, I1{}, I0{}, next_row(4), far_ptr(4, 2));
                tap_body(I0{}, I2{}, I1{}, next_row(5), far_ptr(5, 2));
                tap_body(I1{}, I0{}, I2{}, next_row(6), far_ptr(6, 2));
                tap_body(I1{}, I1{}, I0{}, next_row(7), far_ptr(7, 2));
                tap_body(I1{}, I2{}, I1{}, next_row(8), far_ptr(8, 2));
            } else {
                for (int ta = t0; ta < t1; ta += 3) {  // the three taps of a tap row
                    tap_body(I0{}, I0{}, I2{}, next_row(ta), far_ptr(ta, 2));
                    tap_body(I0{}, I1{}, I0{}, next_row(ta + 1), far_ptr(ta + 1, 2));
                    tap_body(I0{}, I2{}, I1{}, next_row(ta + 2), far_ptr(ta + 2, 2));
                }
            }
        } else {
            // two sets: tap k of the stage uses set (par + k) & 1 and requests the tap behind it into the other
            using SA = integral_constant<int, par>;
            using SB = integral_constant<int, par ^ 1>;
            if (KIND == 1) {
                tap_body(I2{}, SA{}, SB{}, next_row(0), far_ptr(0, 1));
                tap_body(I2{}, SB{}, SA{}, next_row(1), far_ptr(1, 1));
                tap_body(I2{}, SA{}, SB{}, next_row(2), far_ptr(2, 1));
                tap_body(I0{}, SB{}, SA{}, next_row(3), far_ptr(3, 1));
                tap_body(I0{}, SA{}, SB{}, next_row(4), far_ptr(4, 1));
                tap_body(I0{}, SB{}, SA{}, next_row(5), far_ptr(5, 1));
                tap_body(I1{}, SA{}, SB{}, next_row(6), far_ptr(6, 1));
                tap_body(I1{}, SB{}, SA{}, next_row(7), far_ptr(7, 1));
                tap_body(I1{}, SA{}, SB{}, next_row(8), far_ptr(8, 1));
            } else {
                // (a stage's 9 or 3 taps: 4 or 1 pairs and a last one)
                int ta = t0;
                for (; ta + 1 < t1; ta += 2) {
                    tap_body(I0{}, SA{}, SB{}, next_row(ta), far_ptr(ta, 1));
                    tap_body(I0{}, SB{}, SA{}, next_row(ta + 1), far_ptr(ta + 1, 1));
                }
                tap_body(I0{}, SA{}, SB{}, next_row(ta), far_ptr(ta, 1));
            }
        }
    };
    while (more) {
        stage(I0{});
        if constexpr (WSETS == 2 || KIND == 2) {  // (an odd number of taps per stage: the next one starts on the other set)
            if (more) stage(I1{});
        }
        if constexpr (WSETS == 3 && KIND == 2) {
            if (more) stage(I2{});
        }
    }
    const unsigned long long c_u = PNMN_CYC();
    lds_barrier();  // end of the unit's contraction: the loader may rewrite the row table
    const unsigned long long c_e = PNMN_CYC();
    if (exist != 0u) {
        if constexpr (NW == 2) {
            if (!(it.flags & (PNMN_CONV_ACCUMULATE | PNMN_CONV_MASKBWD | PNMN_CONV_DATTN))) {
                epilogue_plain_pair<H, W, TH, MTW>(it, acc[0], acc[1], mbase, n0, band, L.out_stride, L.relu, lane, bias4[0], bias4[NW - 1]);
            } else {
#pragma unroll
                for (int n = 0; n < NW; ++n)
                    epilogue<H, W, TH, MTW>(it, acc[n], mbase, n0 + 16 * n, band, L.out_stride, L.relu, lane, bias4[n]);
            }
        } else {
#pragma unroll
            for (int n = 0; n < NW; ++n)
                epilogue<H, W, TH, MTW>(it, acc[n], mbase, n0 + 16 * n, band, L.out_stride, L.relu, lane, bias4[n]);
        }
    }
    Wk.next_unit(L);
    const unsigned long long c_x = PNMN_CYC();
    cyc[0] += c_e - c_u;
    cyc[1] += c_x - c_e;
    cyc[2] += c_x - c_unit;
    cyc[3] += 1;
}

template <int H, int W, int TH, int TAPS>
__device__ __forceinline__ void conv_stream(const Launch& L, const pnmn_conv_item* items, char* lds) {
    static_assert(TAPS == 9 || TAPS == 1, "3x3 or 1x1");
    using G = Geom<H, W, TH>;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    if (wave >= LOADER_WAVE) {
        switch (wave - LOADER_WAVE) {
            case 0: loader<H, W, TH, 0>(L, items, lds, lane); break;
            case 1: loader<H, W, TH, 1>(L, items, lds, lane); break;
            case 2: loader<H, W, TH, 2>(L, items, lds, lane); break;
            default: loader<H, W, TH, 3>(L, items, lds, lane); break;
        }
        return;
    }
    Walker<H, W, TH> Wk;
    Wk.start(L, items);
    start_up<H, W, TH>(lds, Wk, L.ntaps, wave, lane);
    int cstart = 0;
    unsigned long long cyc[4] = {0, 0, 0, 0};  // barrier waits, epilogues, units, unit count
    const unsigned long long c_begin = PNMN_CYC();
    while (Wk.valid()) {
        // (all uniform over the workgroup)
        if constexpr (TAPS == 1) {
            switch (Wk.split) {
                case 1: run_unit<H, W, TH, 1, 2>(Wk, L, lds, cstart, wave, lane, cyc); break;
                case 2: run_unit<H, W, TH, 2, 2>(Wk, L, lds, cstart, wave, lane, cyc); break;
                case 4: run_unit<H, W, TH, 4, 2>(Wk, L, lds, cstart, wave, lane, cyc); break;
                default: run_unit<H, W, TH, 8, 2>(Wk, L, lds, cstart, wave, lane, cyc); break;
            }
        } else if (G::WHOLE && Wk.dil == 8 && Wk.split <= 2) {
            if (Wk.split == 1) run_unit<H, W, TH, 1, 1>(Wk, L, lds, cstart, wave, lane, cyc);
            else run_unit<H, W, TH, 2, 1>(Wk, L, lds, cstart, wave, lane, cyc);
        } else {
            switch (Wk.split) {
                case 1: run_unit<H, W, TH, 1, 0>(Wk, L, lds, cstart, wave, lane, cyc); break;
                case 2: run_unit<H, W, TH, 2, 0>(Wk, L, lds, cstart, wave, lane, cyc); break;
                case 4: run_unit<H, W, TH, 4, 0>(Wk, L, lds, cstart, wave, lane, cyc); break;
                case 6: run_unit<H, W, TH, 6, 0>(Wk, L, lds, cstart, wave, lane, cyc); break;
                case 14: run_unit<H, W, TH, 14, 0>(Wk, L, lds, cstart, wave, lane, cyc); break;
                case 26: run_unit<H, W, TH, 26, 0>(Wk, L, lds, cstart, wave, lane, cyc); break;
                default: run_unit<H, W, TH, 8, 0>(Wk, L, lds, cstart, wave, lane, cyc); break;
            }
        }
    }
    if (L.dbg && lane == 0) {
        unsigned long long* d = L.dbg + ((size_t)blockIdx.x * 4 + wave) * 8;
        d[0] = cyc[0], d[1] = cyc[1], d[2] = cyc[2], d[3] = cyc[3], d[4] = PNMN_CYC() - c_begin;
    }
}

}  // namespace stream
}  // namespace pnmn
