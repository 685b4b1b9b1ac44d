// Does s_getreg HW_REG_XCC_ID (what cluster.h's same-XCD check relies on) report the XCD a workgroup
// runs on, and is dispatch round-robin over the XCDs (XCD = linear workgroup id % 8)?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15;
}
int main() {
    const int n = 512;
    int *d, h[n];
    hipMalloc(&d, n * sizeof(int));
    hipLaunchKernelGGL(k, dim3(n), dim3(512), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int hist[16] = {0}, match = 0;
    for (int i = 0; i < n; ++i) {
        hist[h[i]]++;
        match += (h[i] == i % 8);
    }
    printf("first 16 workgroups:");
    for (int i = 0; i < 16; ++i) printf(" %d", h[i]);
    printf("\nhistogram:");
    for (int i = 0; i < 16; ++i) printf(" %d", hist[i]);
    printf("\nworkgroups with XCC id == blockIdx %% 8: %d of %d\n", match, n);
    return 0;
}
