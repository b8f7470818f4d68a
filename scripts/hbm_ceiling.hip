// Measured HBM ceilings on the actual MI355X for the access mixes the pyramid kernel uses
// (SURVEY.md 8d asks for a measured ceiling next to the 8 TB/s nominal).
//   build: hipcc --offload-arch=gfx950 -O3 -o hbm_ceiling scripts/hbm_ceiling.hip ; run: ./hbm_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void k_read(const u32x4 *in, size_t n, uint32_t *sink) {
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { u32x4 v = in[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}
template <bool NT> __global__ void k_write(u32x4 *out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        u32x4 v = {(uint32_t)i, 1u, 2u, 3u};
        if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v;
    }
}
__global__ void k_copy(const u32x4 *in, u32x4 *out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
// the pyramid level-0 mix: read 4 B (4 pixels), write 16 B of gradients + 1 B of the next level
template <bool NT> __global__ void k_mix(const uint32_t *in, u32x4 *out, uint8_t *out2, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t p = in[i];
        u32x4 v = {p, p >> 8, p >> 16, p >> 24};
        if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v;
        out2[i] = (uint8_t)p;
    }
}
template <class F> static double time_ms(F f, int reps = 10) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; i++) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main() {
    const size_t bytes = (size_t)1 << 30, n16 = bytes / 16;
    void *A, *B, *C; uint32_t *sink;
    hipMalloc(&A, bytes); hipMalloc(&B, bytes); hipMalloc(&C, bytes / 4); hipMalloc(&sink, 4);
    hipMemset(A, 1, bytes); hipMemset(B, 2, bytes);
    const dim3 grid(256 * 8), blk(256);
    double t;
    t = time_ms([&] { hipLaunchKernelGGL(k_read, grid, blk, 0, 0, (const u32x4 *)A, n16, sink); });
    printf("read   16B/lane            : %7.1f GB/s\n", bytes / t / 1e6);
    t = time_ms([&] { hipLaunchKernelGGL(k_write<false>, grid, blk, 0, 0, (u32x4 *)B, n16); });
    printf("write  16B/lane            : %7.1f GB/s\n", bytes / t / 1e6);
    t = time_ms([&] { hipLaunchKernelGGL(k_write<true>, grid, blk, 0, 0, (u32x4 *)B, n16); });
    printf("write  16B/lane nontemporal: %7.1f GB/s\n", bytes / t / 1e6);
    t = time_ms([&] { hipLaunchKernelGGL(k_copy, grid, blk, 0, 0, (const u32x4 *)A, (u32x4 *)B, n16); });
    printf("copy   (read+write bytes)  : %7.1f GB/s\n", 2.0 * bytes / t / 1e6);
    const size_t nm = bytes / 16;   // 16 B written per element, 4 B read, 1 B written to the next level
    t = time_ms([&] { hipLaunchKernelGGL(k_mix<false>, grid, blk, 0, 0, (const uint32_t *)A, (u32x4 *)B, (uint8_t *)C, nm); });
    printf("mix 4B read : 17B write    : %7.1f GB/s\n", 21.0 * nm / t / 1e6);
    t = time_ms([&] { hipLaunchKernelGGL(k_mix<true>, grid, blk, 0, 0, (const uint32_t *)A, (u32x4 *)B, (uint8_t *)C, nm); });
    printf("mix 4B read : 17B write  nt: %7.1f GB/s\n", 21.0 * nm / t / 1e6);
    return 0;
}
