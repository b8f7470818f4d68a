// Developer microbenchmark (not part of the library): issue cost and dependent latency of the f64
// operations the EKF kernels are built from, measured with s_memtime inside one workgroup.
//   hipcc --offload-arch=gfx950 -O3 -o f64_ubench f64_ubench.hip && ./f64_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double double4v __attribute__((ext_vector_type(4)));
#define N 480

__global__ void k_mfma_dep(double *out, long long *cyc, long long *wall)
{
    double4v acc = {0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
    __syncthreads();
    long long w0 = wall_clock64(), t0 = clock64();
#pragma unroll 16
    for (int i = 0; i < N; i++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    long long t1 = clock64(), w1 = wall_clock64();
    out[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    if (threadIdx.x == 0) { cyc[0] = t1 - t0; wall[0] = w1 - w0; }
}

__global__ void k_mfma_ind(double *out, long long *cyc)
{
    double4v c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
    __syncthreads();
    long long t0 = clock64();
#pragma unroll 4
    for (int i = 0; i < N / 4; i++) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    long long t1 = clock64();
    out[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

// 3 accumulators, 12 distinct A registers and 4 distinct B registers per block, operands from LDS
// (the shape of the H P loop in ekf.hip)
__global__ void k_mfma_lds(double *out, long long *cyc)
{
    __shared__ double hs[48 * 64];
    for (int i = threadIdx.x; i < 48 * 64; i += blockDim.x) hs[i] = 1e-3 * (i % 97);
    __syncthreads();
    const int lane = threadIdx.x & 63, kq = lane >> 4, cl = lane & 15;
    double4v acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    double bv[4] = {1.0 + lane, 2.0 + lane, 3.0 + lane, 4.0 + lane};
    long long t0 = clock64();
    for (int blk = 0; blk < N / 12; blk++) {
        double av[4][3];
#pragma unroll
        for (int sx = 0; sx < 4; sx++)
#pragma unroll
            for (int mt = 0; mt < 3; mt++) av[sx][mt] = hs[((blk & 3) * 16 + 4 * sx + kq) * 48 + 16 * mt + cl];
#pragma unroll
        for (int sx = 0; sx < 4; sx++)
#pragma unroll
            for (int mt = 0; mt < 3; mt++) acc[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[sx][mt], bv[sx], acc[mt], 0, 0, 0);
    }
    long long t1 = clock64();
    out[threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2];
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

// ---- MFMA vs plain LDS-tiled VALU on the EKF's own product shape: C(160x160) -= Y'(160x40) Y(40x160),
// Y resident in LDS (row c of Y at ys[c * 161 ...]), one 512-thread workgroup, C kept in registers and
// summed into `out` so that nothing is optimised away. This is the covariance downdate P -= Y'Y of
// ekf_update_kernel (n = 160, 40 measurement rows).
constexpr int YN = 160, YK = 40, YS = 161;
__global__ __launch_bounds__(512) void k_yty_mfma(double *out, long long *cyc)
{
    __shared__ double ys[YK * YS];
    for (int i = threadIdx.x; i < YK * YS; i += 512) ys[i] = 1e-3 * ((i * 7) % 113);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, kq = lane >> 4, cl = lane & 15;
    double acc_sum = 0;
    long long t0 = clock64();
    for (int tile = wave; tile < 100; tile += 8) {                  // 10 x 10 tiles of 16 x 16
        const int i0 = (tile % 10) * 16, j0 = (tile / 10) * 16;
        double4v acc = {0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < YK / 4; s++)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ys[(4 * s + kq) * YS + i0 + cl], ys[(4 * s + kq) * YS + j0 + cl], acc, 0, 0, 0);
        acc_sum += acc[0] + acc[1] + acc[2] + acc[3];
    }
    long long t1 = clock64();
    out[threadIdx.x] = acc_sum;
    if (lane == 0) cyc[wave] = t1 - t0;
}

__global__ __launch_bounds__(512) void k_yty_valu(double *out, long long *cyc)
{
    __shared__ double ys[YK * YS];
    for (int i = threadIdx.x; i < YK * YS; i += 512) ys[i] = 1e-3 * ((i * 7) % 113);
    __syncthreads();
    // 160 x 160 outputs = 1600 register tiles of 4 x 4; thread t owns tiles t, t + 512, ... (3.125 each):
    // per k: 4 + 4 LDS operands (two ds_read_b128 were it aligned; plain loads here) feed 16 FMAs
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double acc_sum = 0;
    long long t0 = clock64();
    for (int tile = threadIdx.x; tile < 1600; tile += 512) {
        const int i0 = (tile % 40) * 4, j0 = (tile / 40) * 4;
        double c[4][4] = {};
#pragma unroll 4
        for (int k = 0; k < YK; k++) {
            double a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { a[u] = ys[k * YS + i0 + u]; b[u] = ys[k * YS + j0 + u]; }
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int v = 0; v < 4; v++) c[u][v] = __builtin_fma(a[u], b[v], c[u][v]);
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int v = 0; v < 4; v++) acc_sum += c[u][v];
    }
    long long t1 = clock64();
    out[threadIdx.x] = acc_sum;
    if (lane == 0) cyc[wave] = t1 - t0;
}

__global__ void k_fma_dep(double *out, long long *cyc)
{
    double x = threadIdx.x * 1e-3, a = 1.0000001, b = 1e-9;
    long long t0 = clock64();
#pragma unroll 16
    for (int i = 0; i < N; i++) x = __builtin_fma(x, a, b);
    long long t1 = clock64();
    out[threadIdx.x] = x;
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

__global__ void k_fma_ind(double *out, long long *cyc)
{
    double x[8];
    for (int j = 0; j < 8; j++) x[j] = threadIdx.x * 1e-3 + j;
    const double a = 1.0000001, b = 1e-9;
    long long t0 = clock64();
#pragma unroll 2
    for (int i = 0; i < N / 8; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = __builtin_fma(x[j], a, b);
    long long t1 = clock64();
    double s = 0; for (int j = 0; j < 8; j++) s += x[j];
    out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

__global__ void k_rsq_dep(double *out, long long *cyc)
{
    double x = 2.0 + threadIdx.x * 1e-3;
    long long t0 = clock64();
#pragma unroll 16
    for (int i = 0; i < N; i++) x = __builtin_amdgcn_rsq(x) + 1.5;
    long long t1 = clock64();
    out[threadIdx.x] = x;
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

__global__ void k_rsqrt_full_dep(double *out, long long *cyc)
{
    double x = 2.0 + threadIdx.x * 1e-3;
    long long t0 = clock64();
#pragma unroll 16
    for (int i = 0; i < N; i++) x = rsqrt(x) + 1.5;
    long long t1 = clock64();
    out[threadIdx.x] = x;
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

__global__ void k_lds_chase(double *out, long long *cyc)
{
    __shared__ int nxt[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) nxt[i] = (i * 17 + 5) & 1023;
    __syncthreads();
    int p = threadIdx.x;
    long long t0 = clock64();
#pragma unroll 16
    for (int i = 0; i < N; i++) p = nxt[p];
    long long t1 = clock64();
    out[threadIdx.x] = p;
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

__global__ void k_readlane_fma(double *out, long long *cyc)
{
    double x = threadIdx.x * 1e-3 + 1.0, y = 0.5;
    long long t0 = clock64();
#pragma unroll 16
    for (int i = 0; i < N; i++) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(x), 3), hi = __builtin_amdgcn_readlane(__double2hiint(x), 3);
        x = __builtin_fma(__hiloint2double(hi, lo), 1e-9, x) ;
    }
    long long t1 = clock64();
    out[threadIdx.x] = x + y;
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

__global__ void k_barrier(double *out, long long *cyc)
{
    long long t0 = clock64();
#pragma unroll 16
    for (int i = 0; i < N; i++) __syncthreads();
    long long t1 = clock64();
    out[threadIdx.x] = 0;
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

int main()
{
    double *out; long long *cyc, *wall;
    hipMalloc(&out, 1024 * 8); hipMalloc(&cyc, 64 * 8); hipMalloc(&wall, 8);
    long long h[64], hw;
    auto report = [&](const char *name, int waves) {
        hipDeviceSynchronize();
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        long long mx = 0; for (int i = 0; i < waves; i++) if (h[i] > mx) mx = h[i];
        printf("%-34s waves=%2d  %8.1f ticks/op (max over waves)\n", name, waves, (double)mx / N);
    };
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_mfma_dep, dim3(1), dim3(64), 0, 0, out, cyc, wall);
        hipDeviceSynchronize();
        hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(&hw, wall, 8, hipMemcpyDeviceToHost);
        printf("mfma f64 16x16x4 dependent          waves= 1  %8.1f ticks/op   (clock64 ticks %lld, wall_clock64 ticks %lld -> clock64 = %.1f MHz if wall is 100 MHz)\n",
               (double)h[0] / N, h[0], hw, 100.0 * h[0] / hw);
    }
    for (int threads : {64, 256, 1024}) {
        hipLaunchKernelGGL(k_mfma_ind, dim3(1), dim3(threads), 0, 0, out, cyc); report("mfma f64 16x16x4 4 accumulators", threads / 64);
    }
    for (int threads : {64, 256, 512, 1024}) { hipLaunchKernelGGL(k_mfma_lds, dim3(1), dim3(threads), 0, 0, out, cyc); report("mfma f64, 3 acc, A from LDS per block", threads / 64); }
    // the same with every CU busy (blocks all write the same stamps; any block's value will do): does the chip
    // hold its clock under f64 MFMA load on all 256 CUs?
    for (int rep = 0; rep < 3; rep++) { hipLaunchKernelGGL(k_mfma_lds, dim3(1024), dim3(512), 0, 0, out, cyc); report("same, 1024 workgroups (all CUs busy)", 8); }
    {
        auto whole = [&](const char *name) {
            hipDeviceSynchronize();
            hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
            long long mx = 0; for (int i = 0; i < 8; i++) if (h[i] > mx) mx = h[i];
            printf("%-58s %8lld ticks for 1.02 M MACs (%.1f MAC/clk/CU)\n", name, mx, 160.0 * 160 * 40 / mx);
        };
        for (int rep = 0; rep < 2; rep++) {
            hipLaunchKernelGGL(k_yty_mfma, dim3(1), dim3(512), 0, 0, out, cyc); whole("P -= Y'Y shape, MFMA 16x16x4 tiles, operands from LDS");
            hipLaunchKernelGGL(k_yty_valu, dim3(1), dim3(512), 0, 0, out, cyc); whole("P -= Y'Y shape, LDS-tiled VALU (4x4 register tiles)");
        }
    }
    for (int threads : {64, 256, 1024}) { hipLaunchKernelGGL(k_fma_dep, dim3(1), dim3(threads), 0, 0, out, cyc); report("v_fma_f64 dependent", threads / 64); }
    for (int threads : {64, 256, 1024}) { hipLaunchKernelGGL(k_fma_ind, dim3(1), dim3(threads), 0, 0, out, cyc); report("v_fma_f64 8 independent", threads / 64); }
    hipLaunchKernelGGL(k_rsq_dep, dim3(1), dim3(64), 0, 0, out, cyc); report("v_rsq_f64 + add dependent", 1);
    hipLaunchKernelGGL(k_rsqrt_full_dep, dim3(1), dim3(64), 0, 0, out, cyc); report("rsqrt(double) + add dependent", 1);
    hipLaunchKernelGGL(k_lds_chase, dim3(1), dim3(64), 0, 0, out, cyc); report("ds_read_b32 pointer chase", 1);
    hipLaunchKernelGGL(k_readlane_fma, dim3(1), dim3(64), 0, 0, out, cyc); report("2x readlane + fma dependent", 1);
    for (int threads : {64, 256, 1024}) { hipLaunchKernelGGL(k_barrier, dim3(1), dim3(threads), 0, 0, out, cyc); report("s_barrier", threads / 64); }
    return 0;
}
