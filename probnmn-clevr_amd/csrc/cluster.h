// Hand-off primitives shared by the multi-CU recurrent kernels (seq2seq.hip, decoder_multi.hip).
//
// S workgroups that share one 16-row tile step in lockstep; between phases they exchange small vectors
// through L2.  One monotonic counter per tile: every member adds 1 per hand-off, so after the k-th
// hand-off the counter reads S * k -- no reset, no ABA.
//   signal: this wave's stores have reached L2 (s_waitcnt vmcnt(0)) -> workgroup barrier -> ONE
//           increment of the counter (agent-scope release)
//   wait:   thread 0 spins on the counter (bounded: a trap, not a hung GPU, if the partners never show
//           up -- the grid is sized by the host so that every member is resident), one agent-scope
//           acquire (also invalidates this CU's vector L1, which all waves of the workgroup share), barrier
// blockIdx -> (tile, member) keeps a tile's members on one XCD (dispatch is round-robin over the 8
// XCDs), which makes the exchange an L2 hit; correctness does not depend on it.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>

namespace pnmn {

// Hand-off state of one workgroup.  `fast` is decided once per launch (see start()): when every member
// of the tile runs on the same XCD, L2 is their common coherence point, so a hand-off needs neither the
// L2 write-back of an agent-scope release nor the L2 invalidate of an agent-scope acquire -- only
// "my stores have reached L2" on one side and "drop this CU's L1" on the other.  Measured on MI355X
// the agent-scope pair costs 6-9 us per hand-off with 256 workgroups resident (every release writes
// back all of the XCD's dirty output lines); the same-XCD pair costs a fraction of that.
struct Cluster {
    int* counter;   // [0] arrivals (monotonic), [1] OR of (1 << XCC id) over the members | members started << 16, [2] members that finished
    int handoffs;
    int members;
    bool fast;

    __device__ __forceinline__ void signal() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores have reached L2
        __syncthreads();
        if (threadIdx.x == 0) {
            if (fast)
                __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else
                __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    __device__ __forceinline__ void wait() {
        const int target = members * ++handoffs;
        if (threadIdx.x == 0) {
            int spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 26)) __builtin_trap();
            }
            if (fast)
                asm volatile("buffer_inv sc0\n\ts_waitcnt vmcnt(0)" ::: "memory");  // this CU's vector L1 only
            else
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }

    // Last act of a member.  The member that finishes last puts the tile's three words back to zero, so the
    // next launch on the stream finds the block as the library created it and no zeroing dispatch has to go
    // in front of every launch (16 per training step).  Every member has passed its last wait() when it gets
    // here, so nobody polls the words any more; vmcnt(0) first: this member's last signal() has been performed.
    __device__ __forceinline__ void finish() {
        if (threadIdx.x == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int before = __hip_atomic_fetch_add(counter + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (before == members - 1) {
                __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(counter + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(counter + 2, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }

    // Start of a launch: the members publish the XCD they run on and all take the same decision.  ONE word carries both
    // the OR of (1 << XCC id) (low half) and the number of members that have published (high half): a member's two
    // relaxed read-modify-writes of that word are ordered (same address), so whoever reads the full count also reads every
    // member's bit -- no release / acquire pair, which with 256 workgroups resident costs 6-9 us per launch (an agent-scope
    // release writes back the XCD's dirty lines; a step has ~20 such launches).  Nothing else is published here: what the
    // members read of earlier kernels is ordered by the kernel boundary.
    __device__ __forceinline__ void start(int* tile_counter, int n_members) {
        counter = tile_counter;
        handoffs = 0;
        members = n_members;
        fast = false;
        if (threadIdx.x == 0) {
            const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;  // HW_REG_XCC_ID[3:0]
            __hip_atomic_fetch_or(counter + 1, 1 << xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(counter + 1, 1 << 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while ((__hip_atomic_load(counter + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 16) < members) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 26)) __builtin_trap();
            }
        }
        __syncthreads();
        // (the word only grows until the last member's finish(): whoever reads it now reads every member's bit)
        const int word = __hip_atomic_load(counter + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        fast = __builtin_popcount(word & 0xFFFF) == 1;
    }
};

// (tile, member) of this workgroup for a grid of 8 * S * ceil(tiles / 8) workgroups
template <int S>
__device__ __forceinline__ void cluster_coords(int& tile, int& member) {
    const int slot = blockIdx.x >> 3;
    tile = (blockIdx.x & 7) + 8 * (slot / S);
    member = slot % S;
}

// CUs the multi-CU kernels leave alone.  A data-parallel run keeps RCCL's workgroups (one per channel, resident
// for the whole collective, waiting for their PEERS) off the CUs these kernels size their grids for: a grid that
// needs every CU while part of the chip is held by a kernel that itself waits is what stalled round 1.
// Set by pnmn_cluster_reserve_cus() or PNMN_CLUSTER_RESERVE_CUS; one instance per library (inline, C++17).
inline int& cluster_reserved_cus() {
    static int reserved = [] {
        const char* e = getenv("PNMN_CLUSTER_RESERVE_CUS");
        const int v = e ? atoi(e) : 0;
        return v > 0 ? v : 0;
    }();
    return reserved;
}

inline int physical_cus() {
    // cached per device: the design is one process per GPU, but a process that drives several devices must
    // not size a grid for device 1 from device 0's CU count
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
    if (!cus[dev] && hipDeviceGetAttribute(&cus[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        cus[dev] = -1;
    return cus[dev];
}

// CUs a multi-CU launch may count on (physical minus the reserve, never fewer than 32 of a real device)
inline int device_cus() {
    const int cus = physical_cus();
    if (cus <= 0) return cus;
    const int left = cus - cluster_reserved_cus();
    return left >= 32 ? left : (cus < 32 ? cus : 32);
}

// Largest number of 16-row tiles ONE multi-CU launch can take (four members per tile); 0 = none.
inline int cluster_max_tiles() {
    const int cus = device_cus();
    return cus >= 32 ? 8 * (cus / 32) : 0;
}

// Members per tile such that the whole grid is resident with one workgroup per CU; 0 = does not fit.
inline int cluster_split(int tiles, bool allow4 = true) {
    const int cus = device_cus();
    if (cus <= 0) return 0;
    const int groups = (tiles + 7) / 8;
    if (8 * groups * 8 <= cus) return 8;
    if (allow4 && 8 * groups * 4 <= cus) return 4;
    return 0;
}

// One counter line per tile, each on its own 256-byte line: the members of up to 128 tiles poll and increment
// concurrently, and counters sharing a line would serialise all of them on one L2 channel.
//
// The lines live in a block the LIBRARY owns, one per (device, stream), zeroed once when it is created; the
// kernels leave it zeroed (Cluster::finish), and launches on one stream run one after the other.  Only while
// the stream is being captured into a hipGraph do the counters go into the caller's workspace behind a
// zeroing KERNEL (a replayed graph may run next to eager launches of the stream it was captured on; and a
// memset node was seen to run out of order with the kernel nodes around it -- kernel nodes keep stream order).
static __global__ void cluster_zero_kernel(int* words, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) words[i] = 0;
}

constexpr int CLUSTER_COUNTER_STRIDE = 64;                                     // ints
constexpr size_t CLUSTER_SYNC_BYTES = 128 * CLUSTER_COUNTER_STRIDE * sizeof(int);  // up to 128 tiles

// The counter block a launch on `stream` uses; `workspace` = the caller's (>= CLUSTER_SYNC_BYTES), used under capture.
inline hipError_t cluster_sync_block(void* workspace, hipStream_t stream, int** out) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    hipError_t e = hipStreamIsCapturing(stream, &cap);
    if (e != hipSuccess) return e;
    if (cap != hipStreamCaptureStatusNone) {
        *out = static_cast<int*>(workspace);
        hipLaunchKernelGGL(cluster_zero_kernel, dim3(8), dim3(256), 0, stream, *out, (int)(CLUSTER_SYNC_BYTES / sizeof(int)));
        return hipGetLastError();
    }
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, int*> blocks;
    int dev = 0;
    if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    int*& b = blocks[{dev, stream}];
    if (!b) {
        void* p = nullptr;
        if ((e = hipMalloc(&p, CLUSTER_SYNC_BYTES)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(p, 0, CLUSTER_SYNC_BYTES, stream)) != hipSuccess) {
            (void)hipFree(p);
            return e;
        }
        b = static_cast<int*>(p);
    }
    *out = b;
    return hipSuccess;
}

}  // namespace pnmn
