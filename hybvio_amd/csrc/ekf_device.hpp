// Device building blocks shared by the EKF kernels (ekf.hip) and the fused prepare + chi2-gate kernel (vu_prepare.hip):
// state layout constants, the f64 MFMA tile product, the in-register 16 x 16 Cholesky block factor and the chi2 table.
// Everything lives in an anonymous namespace (force-inlined per translation unit).
#pragma once
#include <type_traits>
#include "chi2inv95.h"
#include "hv_internal.hpp"

// (both including translation units select fused multiply-adds for their f64 algebra; see ekf.hip)
#pragma clang fp contract(fast)

namespace hv {
namespace {

enum { POS = 0, VEL = 3, ORI = 6, BGA = 10, BAA = 13, BAT = 16, SFT = 19, CAM = 20, INER = 20, POSE = 7, QD = 12 };
enum { Q_ACC = 0, Q_GYRO = 3, Q_BGA_DRIFT = 6, Q_BAA_DRIFT = 9 };

typedef double double4v __attribute__((ext_vector_type(4)));

__device__ const double d_chi2inv95[HV_CHI2INV95_N] = { HV_CHI2INV95_VALUES };

// One wavefront: acc(16x16) = A(16 x K) * B(K x 16) with A(i, k) = Ap[i*sai + k*sak] and
// B(k, j) = Bp[k*sbk + j*sbj]. f64 MFMA operand layout: lane l carries A[l & 15][l >> 4] and
// B[l >> 4][l & 15]; result register q of lane l is C[(l >> 4) + 4 q][l & 15].
// Branch-free on purpose: rows i >= mi / columns j >= nj are clamped to the last valid one (they
// only produce output rows / columns that the caller never stores), the K tail is zeroed with a
// select, and the next U k-steps are prefetched while the current U MFMAs issue (U = 8 where an
// operand streams from HBM: ~2k cycles of latency against 64 cycles per MFMA and 4 waves per SIMD). (Bounds-checked
// lambdas made hipcc emit one exec-masked branch + 64-bit address chain per operand load.)
template <int U = 4>
__device__ __forceinline__ double4v mfma_tile(const double *Ap, int sai, int sak, int mi,
                                              const double *Bp, int sbk, int sbj, int nj, int K)
{
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const double *pa = Ap + min(r, mi - 1) * sai + q * sak;
    const double *pb = Bp + q * sbk + min(r, nj - 1) * sbj;
    const int da = 4 * sak, db = 4 * sbk;
    double4v acc = {0.0, 0.0, 0.0, 0.0};
    const int kfull = (K / (4 * U)) * (4 * U);
    int k0 = 0;
    if (kfull > 0) {
        double a0[U], b0[U];
#pragma unroll
        for (int u = 0; u < U; u++) { a0[u] = pa[u * da]; b0[u] = pb[u * db]; }
        for (k0 = 4 * U; k0 < kfull; k0 += 4 * U) {
            pa += U * da; pb += U * db;
            double a1[U], b1[U];
#pragma unroll
            for (int u = 0; u < U; u++) { a1[u] = pa[u * da]; b1[u] = pb[u * db]; }
#pragma unroll
            for (int u = 0; u < U; u++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], b0[u], acc, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < U; u++) { a0[u] = a1[u]; b0[u] = b1[u]; }
        }
#pragma unroll
        for (int u = 0; u < U; u++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], b0[u], acc, 0, 0, 0);
        pa += U * da; pb += U * db;
    }
    for (k0 = kfull; k0 < K; k0 += 4) {          // tail: clamp the address, zero the value
        const int k = k0 + q, back = max(k - (K - 1), 0);
        const double av = pa[-back * sak], bv = pb[-back * sbk];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(k < K ? av : 0.0, k < K ? bv : 0.0, acc, 0, 0, 0);
        pa += da; pb += db;
    }
    return acc;
}

// Cholesky factor of one 16 x 16 diagonal block AND the inverse of that factor, in one wavefront
// without LDS traffic or barriers on the dependency chain. Lane r < 16 holds row r of the block,
// lane 16 + i holds row i of the identity: running the same right-looking column steps over the
// stacked matrix [D; I] turns it into [L; L^-T] (the tall-matrix identity T L^-T applied to I).
// Pivot values travel through v_readlane (wave-uniform SGPRs), every array index is a constant.
// Rows / columns >= w (ragged last block) are padded with the identity.
__device__ __forceinline__ double lane_bcast(double v, int src_lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}

// NW = 16, or 8 for a ragged last block of at most 8 columns: only NW column steps are run (the padding
// columns are identity and every update they would receive is zero), W is still written 16 x 16.
template <int NW>
__device__ __forceinline__ void factor_diag_block(double *T, double *W, double *col, int R, int j0, int w, int lane)
{
    const int r = lane & 15;
    const bool ident = (lane & 16) != 0;                    // lanes 32..63 mirror 0..31 (results unused)
    double tr[NW];
#pragma unroll
    for (int c = 0; c < NW; c++) {
        const double x = T[(size_t)(j0 + min(c, w - 1)) * R + j0 + min(r, w - 1)];
        const bool from_t = !ident && c <= r && r < w;      // lower triangle of the block; r < w implies c < w
        tr[c] = from_t ? x : (c == r ? 1.0 : 0.0);
    }
    // Right-looking column steps. The dependency chain of step k -> k+1 is
    //   pivot d (readlane) -> rsqrt -> scale column k -> update column k+1 (readlane of L(k+1, k)),
    // all in registers. The updates of columns k+2.. are off that chain: their multipliers L(c, k)
    // are broadcast through LDS (one ds_write of the column, (15-k)/2 ds_read_b128 of uniform pairs)
    // instead of two v_readlane each, and they are applied one step LATE, after the chain work of
    // step k+1 has been issued: a wavefront issues in order, so a wait for the LDS round trip in
    // step k would stall the chain behind it.
    double mprev[NW], lprev = 0.0;
    double *col_dst = lane < 16 ? col + r : col + 256 + lane;   // col[256 .. 527]: dump area (an exec-masked store makes
                                                                  // hipcc wait for the store itself before the next use of LDS data)
#pragma unroll
    for (int k = 0; k < NW; k++) {
        const double d = lane_bcast(tr[k], k);
        // one rsqrt instead of sqrt + divide: v_rsq_f64 and one third-order correction (the device library's sequence without its
        // special-case select: a non-positive pivot gives inf / NaN either way, which the callers report as a failed gate). The
        // empty asm ties the late updates to the v_rsq result: in source order the compiler kept two of them -- with the wait for
        // their LDS operands -- IN FRONT of the v_rsq, i.e. inside the pivot-to-pivot chain (r03 ISA reading).
        double y0 = __builtin_amdgcn_rsq(d);
        asm volatile("" : "+v"(y0), "+v"(lprev));
        if (k >= 1) {                                          // step k-1's updates of columns k+1..: fill the rsqrt latency
#pragma unroll
            for (int c = k + 1; c < NW; c++) tr[c] -= lprev * mprev[c];
        }
        const double e0 = __builtin_fma(-d * y0, y0, 1.0);
        const double inv = __builtin_fma(y0 * e0, __builtin_fma(e0, 0.375, 0.5), y0);
        const double lk = tr[k] * inv;                        // lane k: d * rsqrt(d) = sqrt(d)
        tr[k] = lk;
        if (k + 1 < NW) {
            if (k + 2 < NW) col_dst[k * 16] = lk;            // branch-free: lanes >= 16 write to a dump row
            tr[k + 1] -= lk * lane_bcast(lk, k + 1);          // lane c < 16 holds L(c, k)
        }
#pragma unroll
        for (int c = k + 2; c < NW; c++) mprev[c] = col[k * 16 + c];         // requested now, used in step k+1
        lprev = lk;
        // pin the updates to this step: left alone, hipcc sinks each one to the last use of tr[c]
        // and keeps every multiplier read so far alive
#pragma unroll
        for (int c = k + 1; c < NW; c++) asm volatile("" : "+v"(tr[c]));
    }
    if (lane < 16) {
#pragma unroll
        for (int c = 0; c < NW; c++)
            if (c <= r && r < w) T[(size_t)(j0 + c) * R + j0 + r] = tr[c];
    } else if (lane < 32) {
#pragma unroll
        for (int c = 0; c < 16; c++) W[r * 16 + c] = c < NW ? tr[c < NW ? c : 0] : (c == r ? 1.0 : 0.0);   // W[i][c] = Linv(c, i)
    }
}


// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains every outstanding GLOBAL access of the wave (s_waitcnt
// vmcnt(0) in front of the s_barrier), i.e. exposes the HBM latency of stores nobody in the kernel reads back. For phases that talk
// through LDS alone (global loads are waited for where their registers are used, as always).
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Second half of a chi2 gate, by one workgroup of NT threads: T holds the lower triangle of H P H' (column-major, stride Rs) and v' in row
// nr; adds R = rd I, runs the blocked Cholesky of [S; v'] and returns chi2 = noise_scale z'z in every thread (a non-positive pivot leaves
// inf / NaN in it). work: >= 816 + NT / 64 doubles of LDS scratch. The caller has synchronised the workgroup behind its last write of T.
template <int NT>
__device__ __forceinline__ double gate_factor_chi2(double *T, int Rs, int nr, double rd, double noise_scale, double *work, long long *stamps = nullptr)
{
#ifdef HV_EKF_PHASE_STAMPS
#define GATE_STAMP(i) do { if (stamps && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) stamps[i] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define GATE_STAMP(i) do { (void)stamps; } while (0)
#endif
    const int t = threadIdx.x, lane = t & 63, kq = lane >> 4, cl = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    constexpr int nwaves = NT / 64;
    for (int i = t; i < nr; i += NT) T[(size_t)i * Rs + i] += rd;
    double *W = work, *col = work + 256, *red = work + 256 + 544;
    lds_barrier();
    // blocked Cholesky of [S; v'] (ekf_update_kernel phase C restricted to the measurement rows). The diagonal blocks are factored by ONE
    // wave (a ~340-cycle dependent chain per pivot that keeps its SIMD's issue port about half busy): the workgroups that share a CU use
    // different waves for it (wave w sits on SIMD w % 4; workgroups 256 apart in the grid land on the same CU in the first rounds), or
    // their chains queue up on SIMD 0 while the other three idle.
    const int chol_wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x >> 8) & 3u));
    const int Rlim = nr + 1;
    for (int j0 = 0; j0 < nr; j0 += 16) {
        const int w = min(16, nr - j0);
        const int ntile = (Rlim - j0 + 15) / 16;
        if (j0 > 0) {
            for (int tile = wave; tile < ntile; tile += nwaves) {
                const int i0 = j0 + 16 * tile, mi = Rlim - i0;
                const double4v acc = mfma_tile(T + i0, 1, Rs, mi, T + j0, Rs, 1, w, j0);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int i = kq + 4 * q;
                    if (i < mi && cl < w) T[(size_t)(j0 + cl) * Rs + i0 + i] -= acc[q];
                }
            }
            lds_barrier();
        }
        if (wave == chol_wave) { if (w <= 8) factor_diag_block<8>(T, W, col, Rs, j0, w, lane); else factor_diag_block<16>(T, W, col, Rs, j0, w, lane); }
        lds_barrier();
        for (int i0 = j0 + w + 16 * wave; i0 < Rlim; i0 += 16 * nwaves) {
            const int mi = Rlim - i0;
            const double4v acc = mfma_tile(T + (size_t)j0 * Rs + i0, 1, Rs, mi, W, 16, 1, 16, w);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int i = kq + 4 * q;
                if (i < mi && cl < w) T[(size_t)(j0 + cl) * Rs + i0 + i] = acc[q];
            }
        }
        lds_barrier();
    }
    GATE_STAMP(1);
    double sz = 0;
    for (int c = t; c < nr; c += NT) { const double z = T[(size_t)c * Rs + nr]; sz += z * z; }
    for (int o = 32; o > 0; o >>= 1) sz += __shfl_down(sz, o);
    if (lane == 0) red[wave] = sz;
    lds_barrier();
    double tot = 0;
#pragma unroll
    for (int w2 = 0; w2 < nwaves; w2++) tot += red[w2];
    GATE_STAMP(2);
    return tot * noise_scale;
#undef GATE_STAMP
}

// ---------------------------------------------------------------------------------------------
// S = Hc P(a, a) Hc' of ONE track from the FACTORS of its Jacobian (r04), by one workgroup of NT threads on the vector unit.
// On gfx950 v_mfma_f64_16x16x4 issues every 64 cycles per SIMD -- the f64 matrix peak IS the f64 vector peak (scripts/f64_ubench.hip) --
// so the dense products of sparse_gate buy nothing per flop, and prepareVisualUpdate's Jacobian (triangulation.cpp:940-980) has structure
// they cannot use. On its active columns
//     Hc = Dp + O4 F4
//   Dp (nr x na): rows 2i, 2i+1 (observation i) hold 7 values, in the columns of the observation's OWN pose k_i (-dip R | dip dRpt)
//   O4 (nr x 4):  dip R -- the projection's derivative w.r.t. the triangulated point -- and the negated feature velocity
//   F4 (4 x na):  the point's derivative w.r.t. every active column (dpf, and dpfi's time-shift column); the unit row of that column
// and with W = P(a, a), W(u', u) = P[a_u N + a_u'] (read as stored, no symmetry assumed), A = W Dp' (na x nr), WF = W F4' (na x 4):
//     S = Dp A + (Dp WF) O4' + O4 (F4 A + (F4 WF) O4')
// ONE pass over W: lane u' loads the 7 values W(u', columns of pose k) and forms the 2 ncam entries of A(u', rows of pose k) and its
// share of WF(u', :) from them; S1 = Dp A (7 multiply-adds per entry) and FA = F4 A follow per group of poses while A is in LDS, the
// rank-4 terms at the end: 0.25 M multiply-adds for a 21-pose stereo track where the dense form spends 3.1 M (2220 + 840 MFMAs: 79 k
// of the long gate's 208 k cycles; 16 poses 35 k of 130 k; 12 poses 25 k of 102 k -- profiles/r04/phase_stamps_one_track_visit.txt).
// Same S up to the summation order. A does not fit beside the factors for the longest tracks: the poses are then served in two groups.
// Deterministic: no atomics, partial sums meet in a fixed order.
// LDS: O4 [nr][4], DV [nr][7] (Dp's values), F4 [4][f4s], WF [na][4], WFp [slots][na][3], FA [4][nr], DWF [nr][4], FWF [16] and G (g_cap
// doubles) -- all disjoint from T; acol [na]. The caller has written O4 / DV / F4 and synchronised; T is zeroed, v' in row nr. Ends synchronised.
// ---------------------------------------------------------------------------------------------
// PREFETCH: the next pose's 7 values are requested before this pose's sums (14 more VGPRs: not in the 128-register two-per-CU build,
// where the other workgroup of the CU covers the round trip)
template <int NT, bool PREFETCH>
__device__ __forceinline__ void structured_S(const double *P, int N, const int *acol, int na, int n, int ncam, int nr,
                                             const double *O4, const double *DV, const double *F4, int f4s, double *WF, double *WFp, double *FA,
                                             double *DWF, double *FWF, double *G, int g_cap, double *T, int Rs, long long *stamps = nullptr)
{
#ifdef HV_EKF_PHASE_STAMPS
#define SS_STAMP(i) do { if (stamps && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) stamps[i] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define SS_STAMP(i) do { (void)stamps; } while (0)
#endif
    const int t = threadIdx.x;
    SS_STAMP(0);
    N = __builtin_amdgcn_readfirstlane(N); na = __builtin_amdgcn_readfirstlane(na); n = __builtin_amdgcn_readfirstlane(n);
    ncam = __builtin_amdgcn_readfirstlane(ncam); nr = __builtin_amdgcn_readfirstlane(nr); Rs = __builtin_amdgcn_readfirstlane(Rs);
    // lanes <-> rows u' of W (coalesced: a_u' runs through the 7 consecutive columns of a pose); slots of nap threads share the poses
    const int nap = (na + 63) & ~63, nslots = NT / nap;
    const int slot = t / nap, up = t - slot * nap;
    const bool lane_on = slot < nslots && up < na;
    const unsigned row_b = 8u * (unsigned)acol[lane_on ? up : 0];                  // byte offset of row a_u' inside a column of P
    auto load7 = [&](double (&w)[7], int k) {                                      // W(u', 7 k + j): scalar base + 32-bit lane offset
#pragma unroll
        for (int j = 0; j < 7; j++)
            w[j] = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(P) + ((unsigned)(acol[7 * k + j] * N) * 8u + row_b));
    };
    // pose groups: all at once where A (na x nr, odd stride) fits g_cap, else two
    const int ng = na * ((2 * ncam * n) | 1) <= g_cap ? 1 : 2;
    const int per = (n + ng - 1) / ng, gs = (2 * ncam * per) | 1;
    double wf0 = 0.0, wf1 = 0.0, wf2 = 0.0;
    if (lane_on && slot == 0)                                                       // WF(:, 3) = W's time-shift column (F4's unit row)
        WF[4 * up + 3] = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(P) + ((unsigned)(acol[7 * n] * N) * 8u + row_b));
    // the sums of one pose's values: WF's partial sums and the 2 ncam entries of A(u', rows of pose k)
    auto consume = [&](const double (&w)[7], int k, int k0, int npg) {
#pragma unroll
        for (int j = 0; j < 7; j++) {
            wf0 += w[j] * F4[7 * k + j]; wf1 += w[j] * F4[f4s + 7 * k + j]; wf2 += w[j] * F4[2 * f4s + 7 * k + j];
        }
        for (int cam = 0; cam < ncam; cam++) {
            const double *d0 = DV + 14 * (cam * n + k);
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int j = 0; j < 7; j++) { a0 += w[j] * d0[j]; a1 += w[j] * d0[7 + j]; }
            double *o = G + (size_t)up * gs + 2 * (cam * npg + k - k0);
            o[0] = a0; o[1] = a1;
        }
    };
    for (int g = 0; g < ng; g++) {
        const int k0 = g * per, k1 = min(n, k0 + per), npg = k1 - k0, rows_g = 2 * ncam * npg;
        // group-local row rl = 2 (cam npg + k - k0) + s  <->  global row r = 2 (cam n + k) + s
        auto row_of = [&](int rl) -> int { const int il = rl >> 1, cam = il >= npg ? 1 : 0; return 2 * (cam * n + k0 + il - cam * npg) + (rl & 1); };
        // ---- A = W Dp' for the poses of the group (into G), WF's partial sums ----
        // (requesting all of a lane's poses at once -- up to 3 x 7 values -- was measured SLOWER: 14.4 k against 10.0 k cycles for the first
        //  group of a 21-pose track, 26 spilled VGPRs; the gather is paced by the L1's line rate, not by exposed round trips)
        if (lane_on && k0 + slot < k1) {
            double wc[7], wn[7];
            load7(wc, k0 + slot);
            for (int k = k0 + slot; k < k1; k += nslots) {
                const bool more = PREFETCH && k + nslots < k1;
                if (more) load7(wn, k + nslots);                                   // (the next pose's values are in flight under this one's sums)
                consume(wc, k, k0, npg);
                if (more) {
#pragma unroll
                    for (int j = 0; j < 7; j++) wc[j] = wn[j];
                } else if (!PREFETCH && k + nslots < k1) load7(wc, k + nslots);
            }
        }
        if (g == ng - 1 && lane_on) { double *o = WFp + (size_t)(slot * na + up) * 3; o[0] = wf0; o[1] = wf1; o[2] = wf2; }
        lds_barrier();
        SS_STAMP(1 + 3 * g);
        // ---- FA = F4 A: (c, row) outputs, each a dot product over u' split over `parts` adjacent lanes ----
        {
            int parts = 1;
            while (parts < 8 && 2 * parts * 4 * rows_g <= NT) parts *= 2;
            const int part = t & (parts - 1);
            for (int q0 = 0; q0 < 4 * rows_g; q0 += NT / parts) {                  // (one round wherever 4 rows <= NT: every build there is)
                const int q = q0 + t / parts;
                const bool on = q < 4 * rows_g;
                const int c = on ? q / rows_g : 0, rl = on ? q - c * rows_g : 0;
                double sum = 0.0, sum2 = 0.0;                                       // (two chains: the LDS round trips overlap)
                int u = part;
                for (; u + parts < na; u += 2 * parts) {
                    sum += F4[c * f4s + u] * G[(size_t)u * gs + rl];
                    sum2 += F4[c * f4s + u + parts] * G[(size_t)(u + parts) * gs + rl];
                }
                if (u < na) sum += F4[c * f4s + u] * G[(size_t)u * gs + rl];
                sum += sum2;
                for (int o = 1; o < parts; o <<= 1) sum += __shfl_xor(sum, o);
                if (on && part == 0) FA[c * nr + row_of(rl)] = sum;
            }
        }
        SS_STAMP(2 + 3 * g);
        // ---- S1(r', r) = sum_j Dp(r', 7 k' + j) A(7 k' + j, r), r in the group, r' >= r (A is only read here: no barrier in between).
        // One group = the whole matrix: its lower triangle is walked as PAIRS of columns (r, nr - 1 - r: nr + 1 cells together, nr is even)
        // -- walked as a rectangle, half the lanes of every round sat idle. Two groups keep the rectangle (group column, row).
        {
            auto cell = [&](int rl, int r, int rp) {
                const int ip = rp >> 1, kp = ip >= n ? ip - n : ip;
                const double *d = DV + 7 * rp, *gc = G + (size_t)(7 * kp) * gs + rl;
                double sum = 0.0;
#pragma unroll
                for (int j = 0; j < 7; j++) sum += d[j] * gc[(size_t)j * gs];
                T[(size_t)r * Rs + rp] = sum;
            };
            if (ng == 1) {
                const unsigned inv = (unsigned)((0x100000000ull + (unsigned)nr) / (unsigned)(nr + 1));
                for (int e = t; e < (nr >> 1) * (nr + 1); e += NT) {
                    const int c2 = (int)__umulhi((unsigned)e, inv), off = e - c2 * (nr + 1);
                    const bool first = off < nr - c2;
                    const int r = first ? c2 : nr - 1 - c2, rp = first ? c2 + off : r + (off - (nr - c2));
                    cell(r, r, rp);                                                 // (one group: local row = global row)
                }
            } else {
                const unsigned inv = (unsigned)((0x100000000ull + (unsigned)nr - 1) / (unsigned)nr);
                for (int e = t; e < rows_g * nr; e += NT) {
                    const int rl = (int)__umulhi((unsigned)e, inv), rp = e - rl * nr, r = row_of(rl);
                    if (rp >= r) cell(rl, r, rp);
                }
            }
        }
        lds_barrier();
        SS_STAMP(3 + 3 * g);
    }
    // ---- the rank-4 terms: WF = sum of the slots' partial sums, DWF = Dp WF, FWF = F4 WF, FA += FWF O4', S += DWF O4' + O4 FA ----
    for (int e = t; e < 3 * na; e += NT) {                                         // (WFp: written in front of the last group's first barrier)                                         // (e = 3 u' + c)
        double sum = 0.0;
        for (int sl = 0; sl < nslots; sl++) sum += WFp[(size_t)sl * na * 3 + e];
        // the slots summed the pose columns u < 7 n; F4's last column (the point's derivative w.r.t. the time shift, non-zero with
        // estimateImuCameraTimeShift) meets W's time-shift column, which is WF(:, 3) (r04 advisor: this term was missing)
        const int up = e / 3, c = e - 3 * up;
        WF[4 * up + c] = sum + WF[4 * up + 3] * F4[c * f4s + 7 * n];
    }
    lds_barrier();
    SS_STAMP(7);
    for (int e = t; e < 4 * nr; e += NT) {                                         // DWF(r', c), e = 4 r' + c
        const int rp = e >> 2, c = e & 3, ip = rp >> 1, kp = ip >= n ? ip - n : ip;
        const double *d = DV + 7 * rp;
        double sum = 0.0;
#pragma unroll
        for (int j = 0; j < 7; j++) sum += d[j] * WF[4 * (7 * kp + j) + c];
        DWF[e] = sum;
    }
    {                                                                              // FWF(c', c): 16 dot products over u', 16 lanes each
        const int q = t >> 4, part = t & 15;
        double sum = 0.0;
        if (q < 16) for (int u = part; u < na; u += 16) sum += F4[(q >> 2) * f4s + u] * WF[4 * u + (q & 3)];
        for (int o = 1; o < 16; o <<= 1) sum += __shfl_xor(sum, o);
        if (q < 16 && part == 0) FWF[q] = sum;
    }
    lds_barrier();
    SS_STAMP(8);
    for (int e = t; e < 4 * nr; e += NT) {                                         // FA(c', r) += sum_c FWF(c', c) O4(r, c), e = c' nr + r
        const int cp = e / nr, r = e - cp * nr;
        const double *o = O4 + 4 * r, *f = FWF + 4 * cp;
        FA[e] += f[0] * o[0] + f[1] * o[1] + f[2] * o[2] + f[3] * o[3];
    }
    lds_barrier();
    SS_STAMP(9);
    {
        // the lower triangle only: the cells of column c2 (nr - c2 of them) and of column nr - 1 - c2 (c2 + 1) are nr + 1 together (nr is even)
        const unsigned inv = (unsigned)((0x100000000ull + (unsigned)nr) / (unsigned)(nr + 1));
        for (int e = t; e < (nr >> 1) * (nr + 1); e += NT) {
            const int c2 = (int)__umulhi((unsigned)e, inv), off = e - c2 * (nr + 1);
            const bool first = off < nr - c2;
            const int r = first ? c2 : nr - 1 - c2, rp = first ? c2 + off : r + (off - (nr - c2));
            const double *dw = DWF + 4 * rp, *o = O4 + 4 * r, *op = O4 + 4 * rp;
            double sum = T[(size_t)r * Rs + rp];
#pragma unroll
            for (int c = 0; c < 4; c++) sum += dw[c] * o[c] + op[c] * FA[c * nr + r];
            T[(size_t)r * Rs + rp] = sum;
        }
    }
    lds_barrier();
    SS_STAMP(10);
#undef SS_STAMP
}

// ---------------------------------------------------------------------------------------------
// Column-sparse chi2 gate of ONE track by one workgroup of NT threads: visualTrackOutlierCheck (ekf.cpp:787-819) on the active
// columns only. prepareVisualUpdate fills H in the 7 columns of every pose of the track and in the time-shift column and nowhere
// else (triangulation.cpp:908-921,958-980): with a = that column list (na = 7 poses + 1 entries) and Hc = H(:, a),
//     S = H P[0:l, 0:l] H' + R = Hc P(a, a) Hc' + R        -- the skipped terms are exact zeros, only the summation order differs.
// A 10-pose stereo track touches 71 of 160 columns: 40 KB of P and 0.6 MFLOP instead of 205 KB and 2.6 MFLOP.
// A wavefront owns 16-row blocks J of P(a, a) and chains two MFMA products through its accumulators (ekf_gate_stream_kernel's scheme):
//     G_J = P(a_J, a) Hc'   (A = P gathered from HBM / L2: lane (kq, cl) reads P[acol[4 s + kq] * n + acol[16 J + cl]], 7-double runs;
//                            B = Hc' from LDS)
//     S  += Hc(:, J) G_J    (A = Hc from LDS, B = the accumulator tile of the first product)
// so neither H P nor P(a, a) is ever stored. Then the blocked Cholesky of [S; v'] and chi2 = noise_scale z'z.
// LDS inputs: acol[na]; Hs = Hc k-major [4 ceil(na / 4)][16 TI], zero in rows >= nr and columns >= na; T = (nr + 1) x nr column-major
// with stride Rs, ZEROED except row nr = v'. work: >= 816 + NT / 64 doubles of scratch (may alias Hs: it is used after the products).
// Every thread returns the same chi2; a non-positive pivot leaves inf / NaN in it (the caller reports CHI2, as ekf_update_kernel does).
// ---------------------------------------------------------------------------------------------
// PIPE: the P values of an item rotate through one register buffer with the NEXT item's loads issued behind each MFMA (standalone gate
// kernel, ~110 VGPRs); !PIPE: chunks of 10 k-steps, double buffered (the fused prepare + gate kernel, which lives on 128 VGPRs).
// TIGHT: the staged Hc has nrp = 84 rows per column instead of 16 TI = 96 (the 84-row track of 21 stereo poses: Hc alone is 100 KB; rows
// nr .. 83 zero); the last row tile reads on into the next column instead of meeting zero padding.
// (ONE static array for every instantiation of sparse_gate: the two-per-CU build of the fused kernel has 192 bytes of LDS to spare)
__device__ __forceinline__ int *gate_turn_lds() { __shared__ int turn[8]; return turn; }
constexpr int HV_GATE_TIGHT_ROWS = 84;     // rows per staged column of the TIGHT layout: 21 stereo poses, the longest track there is
template <int TI, int NT, bool PIPE, bool TIGHT = false>
__device__ __forceinline__ double sparse_gate(const double *P, int n, const int *acol, int na, const double *Hs, double *T, int Rs, int nr,
                                              double rd, double noise_scale, double *work, long long *stamps = nullptr)
{
#ifdef HV_EKF_PHASE_STAMPS
#define GATE_STAMP(i) do { if (stamps && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) stamps[i] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define GATE_STAMP(i) do { (void)stamps; } while (0)
#endif
    const int t = threadIdx.x, lane = t & 63, kq = lane >> 4, cl = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    constexpr int nwaves = NT / 64;
    // the shape is the same for every lane: keep it on the scalar unit (the callers' values come out of per-lane loads)
    n = __builtin_amdgcn_readfirstlane(n); na = __builtin_amdgcn_readfirstlane(na); nr = __builtin_amdgcn_readfirstlane(nr);
    Rs = __builtin_amdgcn_readfirstlane(Rs);
    // rows per staged column of Hc: a compile-time constant, so that the LDS addresses of the k-steps are immediates off one lane base
    // (as run-time scalars the compiler hoisted one per k-step out of the item loop and spilled a thousand SGPRs)
    constexpr int nrp = TIGHT ? HV_GATE_TIGHT_ROWS : 16 * TI;
    // fewest k-steps a track that needs TI row tiles can have (<= 4 rows per pose: 4 (TI - 1) + 1 poses, 7 columns each + the SFT column):
    // the k-steps below it are live in every call and take no select
    constexpr int NK_MIN = (7 * (4 * (TI - 1) + 1) + 1 + 3) / 4;
    // Rows of a 16-row tile of Hs: in TIGHT layouts the last tile runs past row nrp into the next column (after the last column:
    // into whatever follows Hs in LDS): those values only reach rows / columns >= nr of G and S, which are never stored -- a garbage row i
    // of an MFMA's A operand stays in output row i, a garbage column c of its B operand in output column c -- so no predication is needed.
    // Columns beyond na4 (second product) are read unclamped -- somewhere inside the workgroup's LDS -- and replaced by zero in a register.
    const int nJ = (na + 15) >> 4, nk = (na + 3) >> 2, na4 = 4 * nk;
    // Work items (J, g): the 16-row block J of P(a, a) against the column tiles ct of group g of Hc' (one group of all TI tiles up to 48
    // rows, two groups of 3 / 3, 3 / 2 or 2 / 2 tiles in the big build); item = g * nJ + J. The P values of block J are the A operand of
    // EVERY column tile: as (J, ct) items -- r03's first form -- each of them was gathered TI times, and the scattered 8-byte gather
    // (~14 cache lines per wave-wide load), not the matrix pipe, set the pace: 52 k cycles of a 10-pose gate against 7 k of MFMA work.
    // Items are dealt to the waves BY SIMD (wave w runs on SIMD w % 4): a 6-wave workgroup has two waves on SIMDs 0 and 1, so a plain
    // round-robin over the waves would give those SIMDs twice the matrix work. Slots for 6 waves: 0 1 2 3 4 5 2 3 (SIMDs 0 1 2 3 0 1 2 3),
    // i.e. waves 2 and 3 take every 4th item, the others every 8th.
    constexpr int NCG = TI > 3 ? 2 : 1, CGS = (TI + NCG - 1) / NCG;
    int *gate_turn = gate_turn_lds();                               // per column tile: the next block J whose partial sums may enter S
    if (t < 8) gate_turn[t] = 0;
    lds_barrier();
    const int n_items = nJ * NCG, stride = nwaves == 6 ? ((wave == 2 || wave == 3) ? 4 : 8) : nwaves;
    // second product + LDS accumulation of one item, given G = P(a_J, a) Hc(tile ct)' in the accumulator layout. CT is a compile-time
    // copy of ct: with `if (rt >= ct)` around every MFMA each tile was its own basic block -- LDS read, full wait, MFMA, branch --
    // and the matrix pipe idled through one LDS round trip per instruction (r03 ISA reading of the 84-row build)
    auto finish_tiles = [&](auto ctc, int J, const double4v &accG) {
        constexpr int CT = decltype(ctc)::value;
        // S(rt, ct) += Hc(16 rt + i, a-index 16 J + k) G(k, 16 ct + c) for the lower tiles rt >= ct; k-step v: A = Hc(.., 16 J + 4 v + kq),
        // B = accG[v] -- the accumulator tile of the first product IS the B-operand layout (lane (kq, c) holds rows 4 v + kq).
        // (rows j >= na of G repeat row na - 1: they meet the zero columns of Hs, or the select below beyond na4)
        constexpr int NR = TI - CT;
        // (the A operands of k-step v + 1 are read while the MFMAs of step v run: two buffers of NR values)
        double ha[2][NR];
        auto load_ha = [&](double (&h)[NR], int v) {
            const int idx = 16 * J + 4 * v + kq;
            const bool live = 16 * J + 4 * v < na4;                            // (wave-uniform: na4 is a multiple of 4)
            const double *col_k = Hs + idx * nrp + cl;
#pragma unroll
            for (int r = 0; r < NR; r++) { const double x = col_k[16 * (CT + r)]; h[r] = live ? x : 0.0; }
        };
        load_ha(ha[0], 0);
        double4v accS[NR];
#pragma unroll
        for (int r = 0; r < NR; r++) accS[r] = double4v{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int v = 0; v < 4; v++) {
            if (v < 3) load_ha(ha[(v + 1) & 1], v + 1);
#pragma unroll
            for (int r = 0; r < NR; r++) accS[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(ha[v & 1][r], accG[v], accS[r], 0, 0, 0);
        }
        // The items' partial S meet in LDS IN THE ORDER OF J (r05): column tile CT receives one contribution per 16-row block J of
        // P(a, a), from whichever wave that item was dealt to; a turn counter per column tile lets block J in only after block J - 1, so the
        // sums are formed in one order whatever the waves' timing -- chi2 and the gate status are reproducible to the bit, run to run.
        // (r03 / r04 added them with ds_add_f64 as the waves arrived: up to ~1 ulp of S between two runs.) Deadlock-free: every wave
        // takes its items in rising order, an item only ever waits for a lower one. The matrix work is done before the wait.
        // (r05 advisor: the wait is only sound while the dealing above hands every wave its items in RISING order and all waves of the
        //  workgroup are resident -- both true by construction here; a change of the slot table that breaks it must fail loudly, not hang
        //  the GPU: the spin is bounded at ~30 ms, far beyond any gate, and traps)
        for (int spin = 0; __hip_atomic_load(&gate_turn[CT], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != J; ++spin) {
            if (spin > (1 << 19)) __builtin_trap();
            __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int r = 0; r < NR; r++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int i = 16 * (CT + r) + kq + 4 * q, c = 16 * CT + cl;
                if (i < nr && c < nr && i >= c) T[(size_t)c * Rs + i] += accS[r][q];
            }
        }
        if (lane == 0) __hip_atomic_store(&gate_turn[CT], J + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);   // (a wave's LDS writes complete in order)
    };
    // the second products of an item's column tiles CT0 .. CT0 + NCT - 1
    auto finish_group = [&](auto ct0c, auto nctc, int J, const double4v *accG) {
        constexpr int CT0 = decltype(ct0c)::value, NCT = decltype(nctc)::value;
        finish_tiles(std::integral_constant<int, CT0>{}, J, accG[0]);
        if constexpr (NCT > 1) finish_tiles(std::integral_constant<int, CT0 + 1>{}, J, accG[1]);
        if constexpr (NCT > 2) finish_tiles(std::integral_constant<int, CT0 + 2>{}, J, accG[2]);
        static_assert(NCT <= 3, "at most three column tiles per item");
    };
    // LDS address of the B operands of k-step u: B(k, c) = Hc(16 ct + c, k), column 4 u + kq. Steps that may lie beyond nk re-read the
    // last live column (their A operand is zero; an unclamped read could meet a NaN bit pattern elsewhere in LDS)
    auto hb_col = [&](int u) -> const double * { return Hs + (size_t)(4 * ((u < NK_MIN) ? u : min(u, nk - 1)) + kq) * nrp + cl; };
    if constexpr (PIPE) {
        // U k-steps of an item live in the rotating buffer: 24 cover 96 active columns (<= 13 poses); the big build (one 8-wave workgroup
        // per CU, 256 VGPRs per wave) holds all 37 k-steps of a 21-pose track -- an in-line remainder exposed one HBM round trip per item
        constexpr int G = 8, U = TI > 3 ? 40 : 24, NG = U / G;
        // element offsets of the K steps of THIS lane (they do not depend on the item): column a_k of P, k = 4 u + kq
        // (BYTE offsets, unsigned: scalar base + 32-bit lane offset is one global_load; a 64-bit lane address cost two adds and a VGPR pair)
        // Kept in registers by the small build; the big one (40 k-steps in flight) re-reads the column list from LDS in front of each
        // prefetch instead of holding 40 more VGPRs
        constexpr bool KOFF_REGS = U <= 24;
        unsigned koff_r[KOFF_REGS ? U : 1];
        if constexpr (KOFF_REGS) {
#pragma unroll
            for (int u = 0; u < U; u++) koff_r[u] = (unsigned)(acol[min(4 * u + kq, na - 1)] * n) * 8u;
        }
        auto koff = [&](int u) -> unsigned {
            if constexpr (KOFF_REGS) return koff_r[u];
            else return (unsigned)(acol[min(4 * u + kq, na - 1)] * n) * 8u;
        };
        auto p_at = [&](unsigned byte_off) -> double { return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(P) + byte_off); };
        // The P values of an item's k-steps rotate through ONE register buffer: as soon as the MFMAs of step u have read cur[u], the
        // load of the NEXT item's step u is issued into it, so its HBM / L2 round trip hides behind the rest of this item (the remaining
        // k-steps, the second products, the LDS atomics). Loaded item by item, every item exposed one full round trip.
        double cur[U];
        auto aj_of = [&](int item) -> unsigned {                               // row a_j of P(a, a): P(a_j, a_k) = P[a_k * n + a_j]
            const int J = item >= nJ ? item - nJ : item;
            return 8u * (unsigned)acol[min(16 * J + cl, na - 1)];
        };
        int it = wave;
        if (it < n_items) {
            const unsigned aj = aj_of(it);
#pragma unroll
            for (int u = 0; u < U; u++) cur[u] = p_at(koff(u) + aj);
        }
        // one item: NCT accumulator chains share every A operand; the Hc' operands of step u + 1 are read from LDS while the MFMAs of
        // step u run (read right in front of their MFMA, every k-step waited for an LDS round trip)
        auto pipe_item = [&](auto ct0c, auto nctc, int J, unsigned aj_next) {
            constexpr int CT0 = decltype(ct0c)::value, NCT = decltype(nctc)::value;
            double4v accG[NCT];
#pragma unroll
            for (int c = 0; c < NCT; c++) accG[c] = double4v{0.0, 0.0, 0.0, 0.0};
            double hq[2][NCT];
            auto load_hb = [&](double (&h)[NCT], int u) {
                const double *col = hb_col(u) + 16 * CT0;
#pragma unroll
                for (int c = 0; c < NCT; c++) h[c] = col[16 * c];
            };
            load_hb(hq[0], 0);
#pragma unroll
            for (int g = 0; g < NG; g++) {
                if (g * G < nk) {                                              // (groups beyond nk: one uniform branch each)
#pragma unroll
                    for (int j = 0; j < G; j++) {
                        const int u = g * G + j;
                        if (u + 1 < U) load_hb(hq[(u + 1) & 1], u + 1);
                        const double av = (u < NK_MIN || u < nk) ? cur[u] : 0.0;
#pragma unroll
                        for (int c = 0; c < NCT; c++) accG[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, hq[u & 1][c], accG[c], 0, 0, 0);
                        cur[u] = p_at(koff(u) + aj_next);
                    }
                }
            }
            if (nk > U) {                                                      // long mono tracks: the remaining k-steps, loaded in line
                const unsigned aj = 8u * (unsigned)acol[min(16 * J + cl, na - 1)];
                for (int s0 = U; s0 < nk; s0 += 8) {
                    double av[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const int s = s0 + u;
                        const double x = p_at((unsigned)(acol[min(4 * s + kq, na - 1)] * n) * 8u + aj);
                        av[u] = s < nk ? x : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const double *col = Hs + (size_t)(4 * min(s0 + u, nk - 1) + kq) * nrp + cl + 16 * CT0;
#pragma unroll
                        for (int c = 0; c < NCT; c++) accG[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], col[16 * c], accG[c], 0, 0, 0);
                    }
                }
            }
            finish_group(ct0c, nctc, J, accG);
        };
        for (; it < n_items; it += stride) {
            const int itn = it + stride;
            // (the last item of a wave re-requests its own values: cheaper than a second copy of the loop body without the prefetch)
            const unsigned aj_next = aj_of(itn < n_items ? itn : it);
            if constexpr (NCG == 1) pipe_item(std::integral_constant<int, 0>{}, std::integral_constant<int, TI>{}, it, aj_next);
            else if (it < nJ)       pipe_item(std::integral_constant<int, 0>{}, std::integral_constant<int, CGS>{}, it, aj_next);
            else                    pipe_item(std::integral_constant<int, CGS>{}, std::integral_constant<int, TI - CGS>{}, it - nJ, aj_next);
        }
    } else {
        // chunked form (the fused prepare + gate kernel on its 128 VGPRs, and the big build: with two or three column tiles per item an
        // item is >= 7 k cycles of matrix work against one exposed load round trip, which the SIMD's other wave covers -- the rotating
        // buffer of 40 k-steps plus three accumulator chains did not fit 256 VGPRs)
        constexpr int U = 10;                  // (the big build with chunks of 16: 21-pose gate 88 k -> 99 k cycles, r03)
        auto chunk_item = [&](auto ct0c, auto nctc, int J) {
            constexpr int CT0 = decltype(ct0c)::value, NCT = decltype(nctc)::value;
            const unsigned aj = (unsigned)acol[min(16 * J + cl, na - 1)];      // row a_j of P(a, a): P(a_j, a_k) = P[a_k * n + a_j]
            double4v accG[NCT];
#pragma unroll
            for (int c = 0; c < NCT; c++) accG[c] = double4v{0.0, 0.0, 0.0, 0.0};
            // k-steps in chunks of U, the next chunk's P values requested before this chunk's MFMAs; steps beyond nk read a clamped
            // address and are replaced by zero WHERE THEY ARE USED (a select next to the load sits in the block of the uniform branch
            // around it and made that block wait for its own loads: no prefetch at all)
            auto load_a = [&](double (&av)[U], int s0) {
#pragma unroll
                for (int u = 0; u < U; u++)                                    // (unsigned byte offset: scalar base + 32-bit lane offset)
                    av[u] = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(P) +
                                                              ((unsigned)acol[min(4 * (s0 + u) + kq, na - 1)] * (unsigned)n + aj) * 8u);
            };
            double a0[U], a1[U];
            load_a(a0, 0);
            for (int s0 = 0; s0 < nk; s0 += U) {
                if (s0 + U < nk) load_a(a1, s0 + U);                           // (uniform)
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const double *col = Hs + (size_t)(4 * min(s0 + u, nk - 1) + kq) * nrp + cl + 16 * CT0;
                    const double av = s0 + u < nk ? a0[u] : 0.0;
#pragma unroll
                    for (int c = 0; c < NCT; c++) accG[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, col[16 * c], accG[c], 0, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < U; u++) a0[u] = a1[u];
            }
            finish_group(ct0c, nctc, J, accG);
        };
        for (int it = wave; it < n_items; it += stride) {
            if constexpr (NCG == 1) chunk_item(std::integral_constant<int, 0>{}, std::integral_constant<int, TI>{}, it);
            else if (it < nJ)       chunk_item(std::integral_constant<int, 0>{}, std::integral_constant<int, CGS>{}, it);
            else                    chunk_item(std::integral_constant<int, CGS>{}, std::integral_constant<int, TI - CGS>{}, it - nJ);
        }
    }
    __syncthreads();
    GATE_STAMP(0);
    return gate_factor_chi2<NT>(T, Rs, nr, rd, noise_scale, work, stamps);
#undef GATE_STAMP
}
}  // namespace
}  // namespace hv
