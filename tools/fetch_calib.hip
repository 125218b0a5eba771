/* fetch_calib — calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on MI355X for the access patterns of the
 * match kernels.  MI355X_MICROARCH.md calibrates FETCH_SIZE only for wide coalesced streaming reads (reports 1/2 of
 * the bytes); "other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own
 * access pattern".  Every kernel below moves a KNOWN number of bytes / touches a known number of memory sectors
 * over a region far larger than the 256 MiB Infinity Cache; tools/calib.sh runs them under
 *   rocprofv3 --kernel-trace --pmc FETCH_SIZE    and    --pmc WRITE_SIZE
 * and profiles/rNN_fetch_calib.txt keeps counter ÷ known bytes per pattern.
 *
 *   stream16   : 16 B per lane, coalesced, every byte of the region once        (the bitmap / payload loads)
 *   gather1    : ONE byte per lane at a random place of the region               (an isolated container probe)
 *   probe8k    : one byte per lane, the 64 lanes of a wave at sorted random slots of ONE random 8 KiB array
 *                (a round of container probes: candidates of a stripe, ascending) — known distinct 64-B sectors
 *   gather4    : 4 B per lane at a random place                                  (a doclen gather)
 *   gather_pair: byte 0, then byte 64 of the same random 128-B line             (is a miss a 64-B or a 128-B fill?)
 *   write16    : 16 B per lane, coalesced stores                                 (candidate output)
 *   scratch96  : every lane stores and reloads 96 B of private (scratch) memory  (the spill of the 128-VGPR bound)
 *
 * Test/measurement tooling: not part of libxgm.so.  Build: hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip */
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void stream16(const uint4* __restrict__ p, size_t n16, unsigned long long* sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) atomicAdd(sink, 1ull);
}

__global__ void gather1(const unsigned char* __restrict__ p, size_t bytes, size_t n_access, unsigned long long* sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_access; i += (size_t)gridDim.x * blockDim.x)
        acc += p[mix(i) % bytes];
    if (acc == 0x12345678u) atomicAdd(sink, 1ull);
}

/* per wave and round: one random 8 KiB array; lane l reads slot floor(l * 128 + r_l) with r_l < 128: ascending slots,
 * exactly one per 128-B line → 64 distinct 128-B lines = 64 distinct (even or odd) 64-B sectors per round */
__global__ void probe8k_spread(const unsigned char* __restrict__ p, size_t bytes, size_t rounds_per_wave, unsigned long long* sink) {
    const uint32_t lane = threadIdx.x & 63u;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t acc = 0;
    for (size_t r = 0; r < rounds_per_wave; ++r) {
        const size_t arr = (mix(wave * 1315423911ull + r) % (bytes >> 13)) << 13;
        acc += p[arr + lane * 128u + (uint32_t)(mix(wave * 64u + lane + r * 7919u) & 127u)];
    }
    if (acc == 0x12345678u) atomicAdd(sink, 1ull);
}

/* the same, but the 64 lanes fall into 8 sectors: lane l reads slot (l / 8) * 1024 + (l % 8) * 8 → 8 distinct sectors */
__global__ void probe8k_dense(const unsigned char* __restrict__ p, size_t bytes, size_t rounds_per_wave, unsigned long long* sink) {
    const uint32_t lane = threadIdx.x & 63u;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t acc = 0;
    for (size_t r = 0; r < rounds_per_wave; ++r) {
        const size_t arr = (mix(wave * 1315423911ull + r) % (bytes >> 13)) << 13;
        acc += p[arr + (lane >> 3) * 1024u + (lane & 7u) * 8u];
    }
    if (acc == 0x12345678u) atomicAdd(sink, 1ull);
}

__global__ void gather4(const uint32_t* __restrict__ p, size_t n32, size_t n_access, unsigned long long* sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_access; i += (size_t)gridDim.x * blockDim.x)
        acc += p[mix(i) % n32];
    if (acc == 0x12345678u) atomicAdd(sink, 1ull);
}

__global__ void write16(uint4* __restrict__ p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        p[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}

/* 256 dwords of private memory per lane (too much for registers or an LDS promotion: it lives in scratch), of which
 * every round stores and reloads 24 dwords = 96 B per lane */
__global__ void scratch96(const uint32_t* __restrict__ idx, size_t rounds, unsigned long long* sink) {
    uint32_t priv[256];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (size_t r = 0; r < rounds; ++r) {
        const uint32_t o = (uint32_t)(r * 24u) & 255u;
#pragma unroll 1
        for (uint32_t i = 0; i < 24u; ++i) priv[(o + i + idx[(r + i) & 1023u]) & 255u] = t + i + (uint32_t)r;
#pragma unroll 1
        for (uint32_t i = 0; i < 24u; ++i) acc += priv[(o + i * 7u + idx[(r + 2u * i) & 1023u]) & 255u];
    }
    if (acc == 0x12345678u) atomicAdd(sink, 1ull);
}

/* Is an L2 miss a 64-B or a 128-B fill?  Every lane reads byte 0 of a random 128-B line and then byte 64 of the SAME
 * line.  128-B fills: the second read hits → as many requests as gather1 over the same number of lines;
 * 64-B fills: twice as many. */
__global__ void gather_pair(const unsigned char* __restrict__ p, size_t bytes, size_t n_access, unsigned long long* sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_access; i += (size_t)gridDim.x * blockDim.x) {
        const size_t line = (mix(i) % (bytes >> 7)) << 7;
        const uint32_t a = p[line];
        acc += a;
        acc += p[line + 64 + (a & 1u)];               /* depends on the first load: issued after it returned */
    }
    if (acc == 0x12345678u) atomicAdd(sink, 1ull);
}

int main(int argc, char** argv) {
    const size_t GiB = (size_t)1 << 30;
    const size_t bytes = (argc > 1 ? (size_t)atoll(argv[1]) : 8) * GiB;       /* region: >> the 256 MiB Infinity Cache */
    unsigned char* buf;
    unsigned long long* sink;
    uint32_t* idx;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&sink, 8));
    CK(hipMalloc(&idx, 4096));
    CK(hipMemset(buf, 1, bytes));
    CK(hipMemset(sink, 0, 8));
    CK(hipMemset(idx, 0, 4096));
    const dim3 block(256), grid(256 * 16);
    const size_t n_access = (size_t)1 << 28;                                     /* 268 M lane accesses */
    const size_t waves = (size_t)grid.x * block.x / 64, rounds = n_access / 64 / waves;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(stream16, grid, block, 0, 0, (const uint4*)buf, bytes / 16, sink);
        hipLaunchKernelGGL(gather1, grid, block, 0, 0, buf, bytes, n_access, sink);
        hipLaunchKernelGGL(probe8k_spread, grid, block, 0, 0, buf, bytes, rounds, sink);
        hipLaunchKernelGGL(probe8k_dense, grid, block, 0, 0, buf, bytes, rounds, sink);
        hipLaunchKernelGGL(gather4, grid, block, 0, 0, (const uint32_t*)buf, bytes / 4, n_access, sink);
        hipLaunchKernelGGL(gather_pair, grid, block, 0, 0, buf, bytes, n_access, sink);
        hipLaunchKernelGGL(write16, grid, block, 0, 0, (uint4*)buf, bytes / 16);
        hipLaunchKernelGGL(scratch96, grid, block, 0, 0, idx, (size_t)64, sink);
        CK(hipDeviceSynchronize());
    }
    printf("{\"region_bytes\": %zu, \"stream16_bytes\": %zu, \"gather1_accesses\": %zu, \"probe8k_rounds\": %zu, "
           "\"probe8k_spread_sectors\": %zu, \"probe8k_dense_sectors\": %zu, \"gather4_accesses\": %zu, \"write16_bytes\": %zu, "
           "\"scratch_lanes\": %zu, \"scratch_bytes_per_lane_round\": 96, \"scratch_rounds\": 64, \"gather_pair_lines\": %zu}\n",
           bytes, bytes, n_access, rounds * waves, rounds * waves * 64, rounds * waves * 8, n_access, bytes, (size_t)grid.x * block.x, n_access);
    return 0;
}
