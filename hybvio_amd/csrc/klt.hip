// Pyramidal Lucas-Kanade feature tracker, one 64-lane wavefront per feature.
//
// Replaces cv::calcOpticalFlowPyrLK as called at src/tracker/optical_flow.cpp:46-49 (OpenCV
// LKTrackerInvoker; algorithm restated in oracle/pyrlk_oracle.c). Design for CDNA4:
//   * a workgroup IS one wavefront: no workgroup barriers on the Gauss-Newton critical path;
//   * the level loop (coarse -> fine) stays inside the kernel, so one launch tracks every
//     feature of every (prev,next) pair of the batch through all levels;
//   * lane (half, cx) owns window column cx and rows half*16 .. half*16+15 of the 31x31 window;
//     its 16 template samples (I, Ix, Iy; 5+ fractional bits, int16 range) live in VGPRs for the
//     whole level, so an iteration reads only the J window;
//   * the J window comes from a 48x48-byte tile of the next image staged in LDS with coalesced
//     row loads; it is re-staged only when the window leaves the tile (8 px margin);
//   * the 2x2 normal equations are accumulated as exact integers per lane and reduced across the
//     wavefront with DPP row operations + v_readlane (no LDS, no atomics). Integer sums are
//     order independent, so status / positions are bit-reproducible against the CPU oracle;
//   * image borders are virtual: BORDER_REFLECT_101 index math for gray, zero for gradients
//     (OpenCV pads the pyramid by the window size instead).
// The per-point float sequence (weights, 2x2 solve, termination tests) is evaluated redundantly
// by all lanes in IEEE binary32 without FMA contraction (-ffp-contract=off).
#include <float.h>

#include "hv_internal.hpp"

namespace hv {

namespace {

constexpr int WIN = 31;
constexpr int HALF_ROWS = 16;          // rows per half-wave
constexpr int TS = 48;                 // staged J tile: TS x TS bytes
constexpr int TSD = TS / 4;
constexpr int MARGIN = 8;
constexpr int W_BITS = 14;

struct KltArgs {
    PyrLayout L;
    const uint8_t *slab;
    const uint8_t *const *l0_ptr;
    const int *l0_stride;
    const int *prev_slots, *next_slots;
    int pts_per_pair, n_points;
    const float2 *prev_xy;
    float2 *next_xy;
    uint8_t *status;
    float *err;
    int use_init, max_count;
    double epsilon;
    float min_eig;
};

typedef const __attribute__((address_space(1))) uint8_t *gptr_u8;     // global (not flat) loads
typedef const __attribute__((address_space(1))) uint32_t *gptr_u32;
typedef short short2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ gptr_u8 as_global(const uint8_t *p) { return (gptr_u8)(uintptr_t)p; }

// v_dot2_i32_i16: a.lo*b.lo + a.hi*b.hi + c on packed int16 pairs (full-rate; a 32-bit
// v_mul_lo_u32 is quarter-rate on CDNA and every operand here fits 16 bits).
__device__ __forceinline__ int dot2(uint32_t a, uint32_t b, int c)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, a), __builtin_bit_cast(short2v, b), c, false);
}
__device__ __forceinline__ uint32_t pack16(int lo, int hi) { return ((uint32_t)hi << 16) | ((uint32_t)lo & 0xFFFFu); }
__device__ __forceinline__ uint32_t lo16_pair(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x05040100u); }
__device__ __forceinline__ uint32_t hi16_pair(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }
__device__ __forceinline__ uint32_t pk_sub16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(short2v, a) - __builtin_bit_cast(short2v, b));
}

// Wavefront sum of a small integer (|sum| < 2^31) entirely in DPP: 4 in-row steps, then
// row_bcast:15 / row_bcast:31 carry the row totals forward; lane 63 holds the result.
__device__ __forceinline__ int wave_sum_small(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);    // quad_perm:[1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);    // quad_perm:[2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);   // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);   // row_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1,3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2,3
    return __builtin_amdgcn_readlane(v, 63);
}

// Exact wavefront sum of per-lane int32 partials, returned as the float nearest to it:
// the low 16 bits and the signed high part are reduced separately (no overflow), recombined
// exactly in f64 (< 2^53) and rounded once -- identical to (float)(int64 sum).
__device__ __forceinline__ float wave_sum_to_float(int v)
{
    const int lo = wave_sum_small(v & 0xFFFF);
    const int hi = wave_sum_small(v >> 16);
    return (float)((double)hi * 65536.0 + (double)lo);
}

__device__ __forceinline__ void bilinear_weights(float a, float b, int &iw00, int &iw01, int &iw10, int &iw11)
{
    iw00 = __float2int_rn((1.f - a) * (1.f - b) * (float)(1 << W_BITS));
    iw01 = __float2int_rn(a * (1.f - b) * (float)(1 << W_BITS));
    iw10 = __float2int_rn((1.f - a) * b * (float)(1 << W_BITS));
    iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
}

__global__ __launch_bounds__(64) void klt_kernel(KltArgs a)
{
    __shared__ uint32_t jt[(TS + 1) * TSD];   // +1 row: the unused 17th row read of half 1

    const int pt = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int lane = threadIdx.x;
    const int half = lane >> 5, cx = lane & 31;
    const bool col_ok = cx < WIN;
    const int nrows = half ? (WIN - HALF_ROWS) : HALF_ROWS;
    const PyrLayout &L = a.L;

    const int pair = pt / a.pts_per_pair;
    const int sp = a.prev_slots[pair], sn = a.next_slots[pair];
    const uint8_t *prev_base = a.slab + (long long)sp * L.slot_bytes;
    const uint8_t *next_base = a.slab + (long long)sn * L.slot_bytes;

    const float2 pp = a.prev_xy[pt];
    float2 guess = make_float2(0.f, 0.f);
    if (a.use_init) guess = a.next_xy[pt];

    const float half_win = (float)(WIN - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (float)(1 << 20);
    int st = 1;
    float errv = 0.f;
    float nx = 0.f, ny = 0.f;

    for (int level = L.levels - 1; level >= 0; --level) {
        const float lscale = 1.f / (float)(1 << level);
        float px = pp.x * lscale, py = pp.y * lscale;
        if (level == L.levels - 1) {
            if (a.use_init) { nx = guess.x * lscale; ny = guess.y * lscale; }
            else { nx = px; ny = py; }
        } else {
            nx = nx * 2.f; ny = ny * 2.f;
        }
        px -= half_win; py -= half_win;
        const int ipx = (int)floorf(px), ipy = (int)floorf(py);
        const int w = L.w[level], h = L.h[level];
        if (ipx < -WIN || ipx >= w || ipy < -WIN || ipy >= h) {
            if (level == 0) { st = 0; errv = 0.f; }
            continue;
        }

        gptr_u8 Ig, Jg;
        int Igs, Jgs;
        if (level == 0) {
            Ig = as_global(a.l0_ptr[sp]); Igs = a.l0_stride[sp];
            Jg = as_global(a.l0_ptr[sn]); Jgs = a.l0_stride[sn];
        } else {
            Ig = as_global(prev_base + L.goff[level]); Jg = as_global(next_base + L.goff[level]);
            Igs = Jgs = L.gstride[level];
        }
        const gptr_u32 Id = (gptr_u32)as_global(prev_base + L.doff[level]);
        const int Ids = L.dstride[level];

        // ---- template patch: bilinear samples of I and dI into registers, A = sum(dI dI^T) ----
        int iw00, iw01, iw10, iw11;
        bilinear_weights(px - (float)ipx, py - (float)ipy, iw00, iw01, iw10, iw11);

        // packed int16 pairs (rows 2m, 2m+1): template value, x-gradient, y-gradient
        uint32_t Ivp[HALF_ROWS / 2], IXp[HALF_ROWS / 2], IYp[HALF_ROWS / 2];
        int sA11 = 0, sA12 = 0, sA22 = 0;
        {
            const uint32_t wA = pack16(iw00, iw01), wB = pack16(iw10, iw11);
            const int xa = ipx + cx;
            const int rbase = ipy + half * HALF_ROWS;
            // Source rows rbase .. rbase+16 as packed pairs of columns (xa, xa+1). ALL loads of the
            // 17 rows are issued before the first use (one memory round trip per level, not 17).
            uint32_t gp[HALF_ROWS + 1], dxp[HALF_ROWS + 1], dyp[HALF_ROWS + 1];
            const bool inside = ipx >= 0 && ipx + 32 <= w && ipy >= 0 && ipy + 32 <= h;   // wave-uniform
            if (inside) {
                const unsigned g0 = (unsigned)rbase * (unsigned)Igs + (unsigned)xa;
                const unsigned d0 = (unsigned)rbase * (unsigned)Ids + (unsigned)xa;
                uint16_t graw[HALF_ROWS + 1];
                uint2 draw[HALF_ROWS + 1];
#pragma unroll
                for (int r = 0; r <= HALF_ROWS; ++r) {
                    __builtin_memcpy(&graw[r], (const void *)(uintptr_t)(Ig + (g0 + (unsigned)r * (unsigned)Igs)), 2);
                    __builtin_memcpy(&draw[r], (const void *)(uintptr_t)(Id + (d0 + (unsigned)r * (unsigned)Ids)), 8);
                }
#pragma unroll
                for (int r = 0; r <= HALF_ROWS; ++r) {
                    gp[r] = ((uint32_t)graw[r] & 0xFFu) | (((uint32_t)graw[r] & 0xFF00u) << 8);
                    dxp[r] = lo16_pair(draw[r].x, draw[r].y);
                    dyp[r] = hi16_pair(draw[r].x, draw[r].y);
                }
            } else {
                // virtual borders: BORDER_REFLECT_101 for gray, zero for the gradients; loads stay
                // unconditional (clamped addresses) and are masked afterwards -- no divergent branches
                const unsigned gxa = (unsigned)reflect101(xa, w), gxb = (unsigned)reflect101(xa + 1, w);
                const uint32_t ma = (unsigned)xa < (unsigned)w ? 0xFFFFFFFFu : 0u;
                const uint32_t mb = (unsigned)(xa + 1) < (unsigned)w ? 0xFFFFFFFFu : 0u;
                uint32_t ga[HALF_ROWS + 1], gb[HALF_ROWS + 1], da[HALF_ROWS + 1], db[HALF_ROWS + 1];
#pragma unroll
                for (int r = 0; r <= HALF_ROWS; ++r) {
                    const int ry = rbase + r;
                    const unsigned go = (unsigned)reflect101(ry, h) * (unsigned)Igs;
                    const unsigned dof = (unsigned)min(max(ry, 0), h - 1) * (unsigned)Ids;
                    ga[r] = Ig[go + gxa]; gb[r] = Ig[go + gxb];
                    da[r] = Id[dof + gxa]; db[r] = Id[dof + gxb];
                }
                __builtin_amdgcn_sched_barrier(0);   // keep all 68 loads in flight before the first use
#pragma unroll
                for (int r = 0; r <= HALF_ROWS; ++r) {
                    const uint32_t mr = (unsigned)(rbase + r) < (unsigned)h ? 0xFFFFFFFFu : 0u;
                    gp[r] = ga[r] | (gb[r] << 16);
                    dxp[r] = lo16_pair(da[r] & ma & mr, db[r] & mb & mr);
                    dyp[r] = hi16_pair(da[r] & ma & mr, db[r] & mb & mr);
                }
            }
            int iv_prev = 0, ix_prev = 0, iy_prev = 0;
#pragma unroll
            for (int k = 0; k < HALF_ROWS; ++k) {
                int ival = dot2(gp[k + 1], wB, dot2(gp[k], wA, 1 << (W_BITS - 5 - 1))) >> (W_BITS - 5);
                int ixv = dot2(dxp[k + 1], wB, dot2(dxp[k], wA, 1 << (W_BITS - 1))) >> W_BITS;
                int iyv = dot2(dyp[k + 1], wB, dot2(dyp[k], wA, 1 << (W_BITS - 1))) >> W_BITS;
                if (!(col_ok && k < nrows)) { ival = 0; ixv = 0; iyv = 0; }
                sA11 = __mul24(ixv, ixv) + sA11;
                sA12 = __mul24(ixv, iyv) + sA12;
                sA22 = __mul24(iyv, iyv) + sA22;
                if (k & 1) {
                    Ivp[k >> 1] = pack16(iv_prev, ival);
                    IXp[k >> 1] = pack16(ix_prev, ixv);
                    IYp[k >> 1] = pack16(iy_prev, iyv);
                }
                iv_prev = ival; ix_prev = ixv; iy_prev = iyv;
            }
        }
        const float A11 = wave_sum_to_float(sA11) * FLT_SCALE;
        const float A12 = wave_sum_to_float(sA12) * FLT_SCALE;
        const float A22 = wave_sum_to_float(sA22) * FLT_SCALE;

        float D = A11 * A22 - A12 * A12;
        const float min_eig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) /
                              (float)(2 * WIN * WIN);
        if (min_eig < a.min_eig || D < FLT_EPSILON) {
            if (level == 0) st = 0;
            continue;
        }
        D = 1.f / D;

        float cxn = nx - half_win, cyn = ny - half_win;
        float pdx = 0.f, pdy = 0.f;
        int tox = -(1 << 28), toy = -(1 << 28);
        const bool j_aligned = ((reinterpret_cast<uintptr_t>(Jg) | (uintptr_t)Jgs) & 3u) == 0;

        // Stage the J tile whose origin covers window origin (ix, iy) with an 8 px margin.
        auto ensure_tile = [&](int ix, int iy) {
            if (ix >= tox && ix + 32 <= tox + TS && iy >= toy && iy + 32 <= toy + TS) return;
            tox = (ix - MARGIN) & ~3; toy = iy - MARGIN;
            const bool fast = j_aligned && tox >= 0 && tox + TS <= w && toy >= 0 && toy + TS <= h;
            if (fast) {
#pragma unroll
                for (int i = 0; i < (TS * TSD) / 64; ++i) {
                    const int e = lane + 64 * i, r = e / TSD, c = e - r * TSD;
                    jt[e] = *(gptr_u32)(Jg + ((unsigned)(toy + r) * (unsigned)Jgs + (unsigned)(tox + 4 * c)));
                }
            } else {
#pragma unroll
                for (int i = 0; i < (TS * TSD) / 64; ++i) {
                    const int e = lane + 64 * i, r = e / TSD, c = e - r * TSD;
                    const gptr_u8 row = Jg + (unsigned)reflect101(toy + r, h) * (unsigned)Jgs;
                    const int x = tox + 4 * c;
                    jt[e] = (uint32_t)row[(unsigned)reflect101(x, w)] | ((uint32_t)row[(unsigned)reflect101(x + 1, w)] << 8) |
                            ((uint32_t)row[(unsigned)reflect101(x + 2, w)] << 16) | ((uint32_t)row[(unsigned)reflect101(x + 3, w)] << 24);
                }
            }
        };

        for (int j = 0; j < a.max_count; ++j) {
            const int inx = (int)floorf(cxn), iny = (int)floorf(cyn);
            if (inx < -WIN || inx >= w || iny < -WIN || iny >= h) {
                if (level == 0) st = 0;
                break;
            }
            ensure_tile(inx, iny);
            bilinear_weights(cxn - (float)inx, cyn - (float)iny, iw00, iw01, iw10, iw11);

            const uint8_t *jb = reinterpret_cast<const uint8_t *>(jt) +
                                (iny - toy + half * HALF_ROWS) * TS + (inx - tox + cx);
            const uint32_t wA = pack16(iw00, iw01), wB = pack16(iw10, iw11);
            uint32_t jp = (uint32_t)jb[0] | ((uint32_t)jb[1] << 16);
            int sb1 = 0, sb2 = 0;
#pragma unroll
            for (int m = 0; m < HALF_ROWS / 2; ++m) {
                const uint32_t r1 = (uint32_t)jb[(2 * m + 1) * TS] | ((uint32_t)jb[(2 * m + 1) * TS + 1] << 16);
                const uint32_t r2 = (uint32_t)jb[(2 * m + 2) * TS] | ((uint32_t)jb[(2 * m + 2) * TS + 1] << 16);
                const int t0 = dot2(r1, wB, dot2(jp, wA, 1 << (W_BITS - 5 - 1))) >> (W_BITS - 5);
                const int t1 = dot2(r2, wB, dot2(r1, wA, 1 << (W_BITS - 5 - 1))) >> (W_BITS - 5);
                const uint32_t dp = pk_sub16(pack16(t0, t1), Ivp[m]);     // |diff| <= 8160: no int16 overflow
                sb1 = dot2(dp, IXp[m], sb1);
                sb2 = dot2(dp, IYp[m], sb2);
                jp = r2;
            }
            const float b1 = wave_sum_to_float(sb1) * FLT_SCALE;
            const float b2 = wave_sum_to_float(sb2) * FLT_SCALE;
            const float dx = (A12 * b2 - A22 * b1) * D;
            const float dy = (A12 * b1 - A11 * b2) * D;

            cxn += dx; cyn += dy;
            nx = cxn + half_win; ny = cyn + half_win;

            if ((double)dx * (double)dx + (double)dy * (double)dy <= a.epsilon) break;
            if (j > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01) {
                nx -= dx * 0.5f; ny -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }

        // ---- level-0 epilogue (err requested, MIN_EIGENVALS flag unset): may clear status ----
        if (level == 0 && st) {
            const float ex = nx - half_win, ey = ny - half_win;
            const int inx = (int)floorf(ex), iny = (int)floorf(ey);
            if (inx < -WIN || inx >= w || iny < -WIN || iny >= h) {
                st = 0;
            } else {
                ensure_tile(inx, iny);
                bilinear_weights(ex - (float)inx, ey - (float)iny, iw00, iw01, iw10, iw11);
                const uint8_t *jb = reinterpret_cast<const uint8_t *>(jt) +
                                    (iny - toy + half * HALF_ROWS) * TS + (inx - tox + cx);
                int j0a = jb[0], j0b = jb[1];
                int sabs = 0;
#pragma unroll
                for (int k = 0; k < HALF_ROWS; ++k) {
                    const int j1a = jb[(k + 1) * TS], j1b = jb[(k + 1) * TS + 1];
                    const int tv = (__mul24(j0a, iw00) + __mul24(j0b, iw01) +
                                    __mul24(j1a, iw10) + __mul24(j1b, iw11) +
                                    (1 << (W_BITS - 5 - 1))) >> (W_BITS - 5);
                    const int iv = (int)(short)((k & 1) ? (Ivp[k >> 1] >> 16) : (Ivp[k >> 1] & 0xFFFFu));
                    sabs += (col_ok && k < nrows) ? abs(tv - iv) : 0;
                    j0a = j1a; j0b = j1b;
                }
                // sum |diff| <= 961 * 8160 < 2^24: the oracle's float accumulation is exact too
                errv = (float)wave_sum_small(sabs) * 1.f / (float)(32 * WIN * WIN);
            }
        }
    }

    if (lane == 0) {
        a.next_xy[pt] = make_float2(nx, ny);
        a.status[pt] = (uint8_t)st;
        a.err[pt] = errv;
    }
}

}  // namespace

int launch_klt(Ctx *c, int n_pairs, const int *prev_slots_dev, const int *next_slots_dev,
               int pts_per_pair, int n_points, const float *prev_xy, float *next_xy,
               uint8_t *status, float *err, int use_initial_flow, int max_iter)
{
    (void)n_pairs;
    if (n_points <= 0) return HV_OK;
    KltArgs a{};
    a.L = c->L;
    a.slab = c->slab;
    a.l0_ptr = c->d_l0_ptr; a.l0_stride = c->d_l0_stride;
    a.prev_slots = prev_slots_dev; a.next_slots = next_slots_dev;
    a.pts_per_pair = pts_per_pair; a.n_points = n_points;
    a.prev_xy = reinterpret_cast<const float2 *>(prev_xy);
    a.next_xy = reinterpret_cast<float2 *>(next_xy);
    a.status = status; a.err = err;
    a.use_init = use_initial_flow ? 1 : 0;
    int mc = max_iter; if (mc < 0) mc = 0; if (mc > 100) mc = 100;   // cv: clamp(maxCount, 0, 100)
    a.max_count = mc;
    double e = c->p.eps; if (e < 0.) e = 0.; if (e > 10.) e = 10.;   // cv: clamp(eps, 0, 10)^2
    a.epsilon = e * e;
    a.min_eig = (float)c->p.min_eig;
    ScopedKernelTime tm(c, HV_K_KLT);
    hipLaunchKernelGGL(klt_kernel, dim3((unsigned)n_points), dim3(64), 0, c->stream, a);
    HV_HIP(c, hipGetLastError());
    return HV_OK;
}

}  // namespace hv
