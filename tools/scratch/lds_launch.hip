// micro-test: does a kernel's duration depend on its dynamic LDS request?  (256 workgroups x 256 threads, trivial body)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(uint32_t* out, uint32_t n) {
    extern __shared__ uint32_t s[];
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) s[i] = i;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = s[n - 1];
}
int main() {
    uint32_t* d; hipMalloc(&d, 4096 * 4);
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (size_t kb : {4, 16, 32, 64, 65, 96, 128, 139, 160}) {
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kb * 1024));
        for (uint32_t nwg : {64u, 256u, 1024u}) {
            for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k, dim3(nwg), dim3(256), kb * 1024, st, d, 256u);
            hipStreamSynchronize(st);
            hipEventRecord(a, st);
            for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(k, dim3(nwg), dim3(256), kb * 1024, st, d, 256u);
            hipEventRecord(b, st); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("lds %3zu KB  wgs %4u : %.1f us per launch\n", kb, nwg, ms * 1000 / 20);
        }
    }
    return 0;
}
