#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ void k(const uint32_t* __restrict__ src, uint32_t* __restrict__ out, uint32_t n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t* buf = reinterpret_cast<uint32_t*>(smem + wave * 2048);
    for (uint32_t t = 0; t < 2; ++t) {
        const uint32_t* g = src + (size_t)(blockIdx.x * 8 + wave * 2 + t) * 256 + lane * 4;
        if (lane * 4 < n)
            __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(buf + t * 256), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint4 a = *reinterpret_cast<const uint4*>(buf + lane * 4), b = *reinterpret_cast<const uint4*>(buf + 256 + lane * 4);
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w;
}
int main() {
    const int NB = 64; uint32_t *src, *out; size_t n = (size_t)NB * 8 * 256;
    hipMalloc(&src, n * 4); hipMalloc(&out, NB * 256 * 4);
    uint32_t* h = (uint32_t*)malloc(n * 4); for (size_t i = 0; i < n; ++i) h[i] = (uint32_t)(i * 2654435761u);
    hipMemcpy(src, h, n * 4, hipMemcpyHostToDevice);
    k<<<NB, 256, 4 * 2048>>>(src, out, 256);
    uint32_t* ho = (uint32_t*)malloc(NB * 256 * 4); hipMemcpy(ho, out, NB * 256 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < NB; ++b) for (int t = 0; t < 256; ++t) { int w = t >> 6, l = t & 63; uint32_t e = 0; for (int u = 0; u < 2; ++u) for (int i = 0; i < 4; ++i) e ^= h[(size_t)(b * 8 + w * 2 + u) * 256 + l * 4 + i]; bad += ho[b * 256 + t] != e; }
    printf("glds test bad=%d\n", bad); return bad != 0;
}
