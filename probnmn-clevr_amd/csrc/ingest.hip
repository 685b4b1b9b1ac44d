// Feature ingest for gfx950: a batch's image features go from a page-locked host store straight into
// the NHWC device buffer the stem reads -- one launch, no staging copy, no per-row memcpy calls.
//
// The reference keeps features as float64 (N, 1024, H, W) in HDF5, indexes one row per example on the
// host, casts to float and lets the DataLoader collate and the trainer `.to(device)` the batch
// (probnmn/data/readers.py:63-108, datasets.py:137-142, trainers/_trainer.py:272-287): three host passes
// over 0.8 MB per question before the H2D copy.  Here the store is pre-cast fp32 in hipHostMalloc memory
// (device-visible on ROCm); the kernel reads the selected rows over PCIe (coalesced 256-byte runs of one
// channel's pixels), transposes 64 channels x <= 392 pixels through LDS and writes full 256-byte NHWC
// pixel rows.  It replaces the gather, the cast, the H2D copy AND the NCHW -> NHWC pass of the stem's
// prologue.  Bound by PCIe Gen5 x16 (63 GB/s spec): 0.8 MB per question -> ~78 k questions/s per GPU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/probnmn_hip.h"

namespace {

// A FIXED, small number of workgroups walks the (example, 64-channel block, pixel range) work items: the kernel is
// bound by PCIe, not by the chip -- a few hundred KB in flight saturate the link -- and it runs on the loader's
// stream BESIDE a training step.  One workgroup per work item (round 2: 16 384 of them, 50 KB of LDS each) let the
// dispatcher park two or three of these on every CU, where they wait out microsecond PCIe round trips while their
// LDS keeps the step's convolution workgroups (98 KB) off the CU: the step ran 1.45x slower with its ingest beside
// it.  With `gridDim.x` <= 12 resident workgroups (16-byte loads, eight in flight per thread: 42.7 GB/s on an idle chip,
// the most this kernel reaches with any grid) and a 26 KB tile a convolution workgroup still fits next to one.  Measured
// beside the 1024-question step (round 3, gpurun_out r03k-r03n): 4 / 8 / 12 / 16 / 32 / 64 / 1024 workgroups -> 51.6 /
// 40.8 / 39.5 / 39.8 / 44.7 / 49.3 / 52.8 ms per step against 32.4 ms with resident features -- below ~12 the link is
// not filled (the ingest becomes the critical path), above it the step slows with the number of PCIe reads in flight
// (they hold memory-system queue entries for microseconds each), and ~7 ms of the step's 32 stay unhidden at best.
__global__ __launch_bounds__(256) void gather_features_kernel(const float* __restrict__ store,
                                                              const int64_t* __restrict__ indices,
                                                              float* __restrict__ dst, int64_t n_store, int Cn,
                                                              int HW, int PT, int n, int parts) {
    extern __shared__ float tile[];  // [64][PT+1]
    const int cblocks = (Cn + 63) / 64;
    const int total_items = n * cblocks * parts;
    const int ld = PT + 1;
    constexpr int NB = 8;  // loads in flight per thread: the PCIe round trip is microseconds
    for (int item = blockIdx.x; item < total_items; item += gridDim.x) {
        const int part = item % parts, cb = (item / parts) % cblocks, e = item / (parts * cblocks);
        const int c0 = cb * 64;
        const int p0 = part * PT;
        const int np = (HW - p0) < PT ? (HW - p0) : PT;
        const int cw = (Cn - c0) < 64 ? (Cn - c0) : 64;
        int64_t row = indices[e];
        if (row < 0 || row >= n_store) row = 0;  // (validated on the host; never index outside the store)
        const float* src = store + ((size_t)row * Cn + c0) * HW + p0;
        const int total = cw * np;
        if (((np | HW | p0) & 3) == 0) {
            // 16-byte loads (every channel's pixel run starts 16-byte aligned): a quarter of the load instructions per
            // byte -- the kernel shares its CUs' issue slots with the step it runs beside
            const int nq = np >> 2, total4 = cw * nq;
            typedef float f4 __attribute__((ext_vector_type(4)));
            for (int i0 = threadIdx.x; i0 < total4; i0 += 256 * NB) {
                f4 v[NB];
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    const int i = i0 + k * 256;
                    const int c = i / nq;
                    v[k] = i < total4 ? __builtin_nontemporal_load(reinterpret_cast<const f4*>(src + (size_t)c * HW + 4 * (i - c * nq)))
                                      : f4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    const int i = i0 + k * 256;
                    const int c = i / nq;
                    if (i < total4) {
                        float* t = tile + c * ld + 4 * (i - c * nq);
                        t[0] = v[k].x, t[1] = v[k].y, t[2] = v[k].z, t[3] = v[k].w;
                    }
                }
            }
        } else
        for (int i0 = threadIdx.x; i0 < total; i0 += 256 * NB) {
            float v[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int i = i0 + k * 256;
                const int c = i / np;
                v[k] = i < total ? __builtin_nontemporal_load(src + (size_t)c * HW + (i - c * np)) : 0.f;
            }
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int i = i0 + k * 256;
                const int c = i / np;
                if (i < total) tile[c * ld + (i - c * np)] = v[k];
            }
        }
        __syncthreads();
        float* out = dst + ((size_t)e * HW + p0) * Cn + c0;
        for (int i = threadIdx.x; i < total; i += 256) {
            const int p = i / cw, c = i - p * cw;
            out[(size_t)p * Cn + c] = tile[c * ld + p];
        }
        __syncthreads();  // the tile is refilled by the next item
    }
}

}  // namespace

extern "C" int pnmn_gather_features(const float* store, const int64_t* indices, float* dst, int n, int64_t n_store,
                                    int Cn, int HW, void* stream) {
    if (n <= 0) return 0;
    if (!store || !indices || !dst || Cn <= 0 || HW <= 0 || n_store <= 0) return PNMN_EINVAL;
    // pixel ranges of 100 pixels (a multiple of four: 16-byte loads): a 26 KB tile (see the kernel's header)
    const int PT = HW < 100 ? HW : 100;
    const int parts = (HW + PT - 1) / PT;
    const size_t lds = (size_t)64 * (PT + 1) * sizeof(float);
    static const int max_wgs = [] {  // (tuning hook)
        const char* e = getenv("PNMN_INGEST_WGS");
        const int v = e ? atoi(e) : 12;
        return v > 0 ? v : 12;
    }();
    const long items = (long)n * ((Cn + 63) / 64) * parts;
    hipLaunchKernelGGL(gather_features_kernel, dim3((unsigned)(items < max_wgs ? items : max_wgs)), dim3(256), lds,
                       static_cast<hipStream_t>(stream), store, indices, dst, n_store, Cn, HW, PT, n, parts);
    return (int)hipGetLastError();
}

// The same ingest on the COPY ENGINES: one hipMemcpyAsync per selected row (0.8 MB, contiguous in the store) into a
// plain NCHW batch, which the stem's layout pass reads like any other input.  No compute unit takes part: a kernel
// that reads over PCIe keeps its loads outstanding for microseconds, and with enough of them in flight to fill the
// link the step running beside it slowed down by 24-52 % whatever the grid size (r03h_ingest_step.txt); the DMA
// engines move the same bytes without touching the CUs' memory pipelines.  `indices` is a HOST array.
extern "C" int pnmn_copy_rows_h2d(const void* store, const int64_t* indices, void* dst, int n, int64_t n_store,
                                  int64_t row_bytes, void* stream) {
    if (n <= 0) return 0;
    if (!store || !indices || !dst || n_store <= 0 || row_bytes <= 0) return PNMN_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int i = 0; i < n; ++i) {
        const int64_t row = indices[i];
        if (row < 0 || row >= n_store) return PNMN_EINVAL;
        const hipError_t e = hipMemcpyAsync(static_cast<char*>(dst) + (size_t)i * row_bytes,
                                            static_cast<const char*>(store) + (size_t)row * row_bytes, (size_t)row_bytes,
                                            hipMemcpyHostToDevice, st);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}
