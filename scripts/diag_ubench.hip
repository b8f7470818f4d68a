// Developer microbenchmark: the in-register 16x16 diagonal factor (+inverse) of ekf.hip in isolation,
// with pieces switched off to see which part of a column step costs what.
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang fp contract(fast)

__device__ __forceinline__ double lane_bcast(double v, int src_lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}

template <int VARIANT>
__global__ void k_diag(const double *Tin, double *out, long long *cyc)
{
    __shared__ double T[16 * 17], W[256], col[320];
    const int lane = threadIdx.x, r = lane & 15;
    const bool ident = (lane & 16) != 0;
    for (int i = lane; i < 16 * 17; i += 64) T[i] = Tin[i];
    __syncthreads();
    long long t0 = clock64();
    double tr[16];
#pragma unroll
    for (int c = 0; c < 16; c++) {
        const double x = T[c * 17 + r];
        tr[c] = (!ident && c <= r) ? x : (c == r ? 1.0 : 0.0);
    }
    double mprev[16], lprev = 0.0;
    double *col_dst = lane < 16 ? col + r : col + 256 + lane;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const double d = lane_bcast(tr[k], k);
        if (VARIANT != 2 && k >= 1) {
#pragma unroll
            for (int c = k + 1; c < 16; c++) tr[c] -= lprev * mprev[c];
        }
        double inv;
        if (VARIANT == 1) inv = __builtin_amdgcn_rsq(d);      // raw approximation: no refinement
        else if (VARIANT == 3) {                               // hand-rolled refinement, no class test
            const double y0 = __builtin_amdgcn_rsq(d);
            const double e = __builtin_fma(-(d * y0), y0, 1.0);
            inv = __builtin_fma(y0 * e, __builtin_fma(e, 0.375, 0.5), y0);
        } else inv = rsqrt(d);
        const double lk = tr[k] * inv;
        tr[k] = lk;
        if (k + 1 < 16) {
            if (k + 2 < 16) col_dst[k * 16] = lk;
            tr[k + 1] -= lk * lane_bcast(lk, k + 1);
        }
#pragma unroll
        for (int c = k + 2; c < 16; c++) mprev[c] = col[k * 16 + c];
        lprev = lk;
#pragma unroll
        for (int c = k + 1; c < 16; c++) asm volatile("" : "+v"(tr[c]));
    }
    if (lane < 16) {
#pragma unroll
        for (int c = 0; c < 16; c++) if (c <= r) T[c * 17 + r] = tr[c];
    } else if (lane < 32) {
#pragma unroll
        for (int c = 0; c < 16; c++) W[r * 16 + c] = tr[c];
    }
    long long t1 = clock64();
    __syncthreads();
    out[lane] = T[lane] + W[lane];
    if (lane == 0) cyc[0] = t1 - t0;
}

int main()
{
    double h[16 * 17];
    for (int c = 0; c < 16; c++) for (int r = 0; r < 17; r++) h[c * 17 + r] = (r == c) ? 20.0 : 1.0 / (1 + r + c);
    double *Tin, *out; long long *cyc;
    hipMalloc(&Tin, sizeof(h)); hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
    hipMemcpy(Tin, h, sizeof(h), hipMemcpyHostToDevice);
    long long c;
    const char *names[] = {"full (as in ekf.hip)", "raw v_rsq_f64 (no refinement)", "no off-chain updates", "hand refinement without class test"};
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_diag<0>, dim3(1), dim3(64), 0, 0, Tin, out, cyc); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("%-40s %lld ticks\n", names[0], c);
        hipLaunchKernelGGL(k_diag<1>, dim3(1), dim3(64), 0, 0, Tin, out, cyc); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("%-40s %lld ticks\n", names[1], c);
        hipLaunchKernelGGL(k_diag<2>, dim3(1), dim3(64), 0, 0, Tin, out, cyc); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("%-40s %lld ticks\n", names[2], c);
        hipLaunchKernelGGL(k_diag<3>, dim3(1), dim3(64), 0, 0, Tin, out, cyc); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("%-40s %lld ticks\n", names[3], c);
    }
    return 0;
}
