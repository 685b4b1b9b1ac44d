// Hand-off primitives shared by the multi-CU recurrent kernels (seq2seq.hip, decoder_multi.hip).
//
// S workgroups that share one 16-row tile step in lockstep; between phases they exchange small vectors
// through L2.  One monotonic counter per tile: every member adds 1 per hand-off, so after the k-th
// hand-off the counter reads S * k -- no reset, no ABA.
//   signal: this wave's stores have reached L2 (s_waitcnt vmcnt(0)) -> workgroup barrier -> ONE
//           agent-scope release increment
//   wait:   thread 0 spins on the counter (bounded: a trap, not a hung GPU, if the partners never show
//           up -- the grid is sized by the host so that every member is resident), one agent-scope
//           acquire (invalidates this CU's vector L1, which all waves of the workgroup share), barrier
// blockIdx -> (tile, member) keeps a tile's members on one XCD (dispatch is round-robin over the 8
// XCDs), which makes the exchange an L2 hit; correctness does not depend on it.
#pragma once
#include <hip/hip_runtime.h>

namespace pnmn {

__device__ __forceinline__ void cluster_wait(const int* counter, int target) {
    if (threadIdx.x == 0) {
        int spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 26)) __builtin_trap();
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

__device__ __forceinline__ void cluster_signal(int* counter) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// (tile, member) of this workgroup for a grid of 8 * S * ceil(tiles / 8) workgroups
template <int S>
__device__ __forceinline__ void cluster_coords(int& tile, int& member) {
    const int slot = blockIdx.x >> 3;
    tile = (blockIdx.x & 7) + 8 * (slot / S);
    member = slot % S;
}

inline int device_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            cus = -1;
    }
    return cus;
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

constexpr size_t CLUSTER_SYNC_BYTES = 4096;  // step counters of up to 1024 tiles

}  // namespace pnmn
