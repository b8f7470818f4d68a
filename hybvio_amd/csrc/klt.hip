// Pyramidal Lucas-Kanade feature tracker, one 64-lane wavefront per feature.
//
// Replaces cv::calcOpticalFlowPyrLK as called at src/tracker/optical_flow.cpp:46-49 (OpenCV
// LKTrackerInvoker; algorithm restated in oracle/pyrlk_oracle.c). Design for CDNA4:
//   * a workgroup IS one wavefront: no workgroup barriers on the Gauss-Newton critical path;
//   * the level loop (coarse -> fine) stays inside the kernel, so one launch tracks every
//     feature of every (prev,next) pair of the batch through all levels;
//   * lane (half, cx) owns window column cx and rows half*16 .. half*16+15 of the 31x31 window;
//     its 16 template samples (I, Ix, Iy; int16 range) live in VGPRs as packed pairs for the
//     whole level, so an iteration reads only the J window;
//   * the J window comes from a tile of the next image staged in LDS (default 40 x 36 pixels, template parameters TSX x TSY; r01
//     used 44 x 44 and before that 48 x 48) with coalesced row loads, requested together with the template rows of the level and
//     re-staged only when the window leaves it. The tile is stored as one
//     dword per pixel holding the packed pair (J[x], J[x+1]) << 7, so one ds_read_b32 per window
//     row feeds v_dot2_i32_i16 directly and the rounding shift ">> 9" becomes "take the high
//     half" (128*t + 32768 >> 16 == t + 256 >> 9). Gradients are stored x4 in the pyramid for
//     the same reason (4*s + 32768 >> 16 == s + 8192 >> 14). The kernel is VALU-issue bound
//     (rocprof: 93 % VALU busy), so the design minimises integer instructions per pixel;
//   * levels 0 and 1 have NO stored gradient plane (PyrLayout::grad_from, r02): the lane forms the Scharr gradients of its template
//     columns from the gray rows it loads anyway (scharr_row below), in the pre-scaled 4 d + 2 form the pyramid stores for levels >= 2;
//   * the 2x2 normal equations are accumulated as exact integers per lane and reduced across the
//     wavefront with DPP row operations + v_readlane (no LDS, no atomics): one 32-bit chain when
//     a ballot proves the total fits, else an exact 2-chain 64-bit path. Integer sums are
//     order independent, so status / positions are bit-reproducible against the CPU oracle;
//   * image borders: levels 0 and 1 keep VIRTUAL borders (BORDER_REFLECT_101 index math for gray, zero for gradients); levels >= 2
//     carry OpenCV's 31 (+1) pixel border PHYSICALLY in the slot (PyrLayout::pad, r02), so the coarse levels -- where most windows
//     cross the image edge -- run the border-free code paths.
// The per-point float sequence (weights, 2x2 solve, termination tests) is evaluated redundantly
// by all lanes in IEEE binary32 without FMA contraction (-ffp-contract=off).
#include <float.h>
#include <stdlib.h>

#include "hv_internal.hpp"

namespace hv {

namespace {

constexpr int WIN = 31;
constexpr int HALF_ROWS = 16;          // rows per half-wave
// staged J tile: TSX x TSY pixels (one dword each) around the 32 x 32 window, MX / MY pixels of slack on the low side (the
// rest on the high side); restaged only when an iteration moves the window out of it. Template parameters of the kernel:
// {44, 44, 6, 6} is the r01 shape (10 KB of LDS per wave with `region`), smaller tiles cost less to stage.
constexpr int RW = 48;                 // width of the in-image region used to build border tiles (>= TSX + 4)
static_assert(RW / 4 == 12, "the region staging loop divides by 12 with a 24-bit multiply");
constexpr int GRAY_SHIFT = 7;          // gray samples are pre-scaled by 128 (<= 32640: fits int16)
// The four bilinear weights always sum to 2^14, so adding 2 to every pre-scaled sample adds exactly
// 2^15 to the weighted sum: the rounding constant of CV_DESCALE rides along in the data and every
// dot2 chain starts from 0. Gradients are stored as 4*d + 2 for the same reason (pyramid.hip).
constexpr uint32_t ROUND_PAIR = 0x00020002u;
constexpr int W_BITS = 14;
// default build (Knobs::klt_tile = 5): 40 x 36 tile at 5 waves per SIMD (96 VGPRs, 7 dwords of loop-invariant lane constants in scratch, reloaded once per level): -4 % against the 4-wave build of the same tile (bench r02, C2 and C3 legs)

struct KltArgs {
    PyrLayout L;
    const uint8_t *slab;
    const uint8_t *const *l0_ptr;
    const int *l0_stride;
    const int *prev_slots, *next_slots;
    int pts_per_pair, n_points;
    const int *pts_in_pair;            // optional [n_pairs]: points of pair p that exist (<= pts_per_pair); the rest of its record is padding
    const float2 *prev_xy;
    float2 *next_xy;
    uint8_t *status;
    float *err;
    int use_init, max_count;
    double epsilon;
    float min_eig;
    // wave-uniform float constants evaluated on the host (binary32 results identical to the device's: single IEEE operations):
    // the two bands of the f32 convergence test and the eigenvalue bound -- as kernel arguments they live in SGPRs
    float eps_lo, eps_hi, eig_bound, eig_margin;   // eig_margin = 1e-5f * eig_bound (a term of the fast eigenvalue test's margin)
};

typedef const __attribute__((address_space(1))) uint8_t *gptr_u8;     // global (not flat) loads
typedef const __attribute__((address_space(1))) uint32_t *gptr_u32;
typedef short short2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ gptr_u8 as_global(const uint8_t *p) { return (gptr_u8)(uintptr_t)p; }

// Row loads of the border-free paths go through RAW BUFFER instructions (r06): address = descriptor base (4 SGPRs, wave-uniform) +
// soffset (an SGPR: the row's byte offset, one s_add per row) + voffset (the lane's byte offset, formed once per level). The global_load
// forms of r01 .. r05 cost a 64-bit VALU add per load wherever hipcc hoisted base + lane offset into a VGPR pair (v_lshl_add_u64: 68 per
// feature on the stored-gradient levels, profiles/r06/klt_static_census_r05.txt) and 4 scalar instructions per row for the 64-bit row base.
// Every offset is >= 0: the descriptor base is the SLOT (levels >= 1: the level's offset rides in soffset, padded rows lie behind the
// slot base) or the caller's level-0 image (inside paths only).
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const uint8_t *p)
{
    // the pointer is wave-uniform by construction (one feature per wave) but comes out of per-lane loads: v_readfirstlane pins the
    // descriptor to SGPRs once per level (left alone, every buffer_load re-reads its four dwords from VGPRs)
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<uint8_t *>(((uint64_t)hi << 32) | lo), 0, -1 /* no range limit */, 0x00020000);
}

// v_dot2_i32_i16: a.lo*b.lo + a.hi*b.hi + c on packed int16 pairs (a 32-bit v_mul_lo_u32 is
// quarter-rate on CDNA and every operand here fits 16 bits). clamp=true selects the 3-operand
// VOP3P encoding, which takes an inline 0 / a separate accumulator; without it hipcc emits the
// tied-accumulator v_dot2c + a v_mov per chain. No sum here can reach the int32 clamp.
__device__ __forceinline__ int dot2(uint32_t a, uint32_t b, int c)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, a), __builtin_bit_cast(short2v, b), c, true);
}
__device__ __forceinline__ uint32_t lo16_pair(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x05040100u); }
__device__ __forceinline__ uint32_t hi16_pair(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }
__device__ __forceinline__ uint32_t pk_sub16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(short2v, a) - __builtin_bit_cast(short2v, b));
}
// bytes (i, i+1) of the 8-byte string {lo, hi} as the pre-scaled pair (b_i << 7) | (b_i+1 << 23)
template <int I>
__device__ __forceinline__ uint32_t scaled_pair(uint32_t lo, uint32_t hi)
{
    constexpr uint32_t sel = 0x0C000C00u | (uint32_t)I | ((uint32_t)(I + 1) << 16);
    return __builtin_amdgcn_perm(hi, lo, sel) * (1u << GRAY_SHIFT) + ROUND_PAIR;   // no carry between halves
}

// Gradients of the fine levels formed in the kernel (level < PyrLayout::grad_from). q is the dword of gray bytes I(x-1 .. x+2) of one source
// row: the Scharr stencils of pixels x and x+1 (cv::calcSharrDeriv: smooth [3 10 3] across, difference [-1 0 1] along) are
// separable, so a row contributes its horizontal difference d = I(x+1) - I(x-1) and smoothed value s4 = 4 (3 I(x-1) + 10 I(x)
// + 3 I(x+1)) as packed 16-bit pairs for the two columns, and a gradient row is a vertical combination of three of those:
//   4 dx = 12 (d[r-1] + d[r+1]) + 40 d[r]        4 dy = s4[r+1] - s4[r-1]
// -- the pre-scaled form the pyramid stores (pyramid.hip) WITHOUT its + 2: the four bilinear weights sum to 2^14, so the + 2 of every
// sample is one + 32768 of the weighted sum, and the dot2 chains of the gradients formed here start from 32768 instead (window_row's
// ginit; r03: the OR per gradient row was 68 VALU instructions per feature). Every intermediate fits 16 bits (|.| <= 16322): v_pk_* arithmetic.
typedef unsigned short us2v __attribute__((ext_vector_type(2)));
struct ScharrRow { us2v d, s4; uint32_t g; };
__device__ __forceinline__ ScharrRow scharr_row(uint32_t q)
{
    const us2v l = __builtin_bit_cast(us2v, __builtin_amdgcn_perm(0u, q, 0x0C010C00u));    // (I(x-1), I(x))
    const us2v c = __builtin_bit_cast(us2v, __builtin_amdgcn_perm(0u, q, 0x0C020C01u));    // (I(x),   I(x+1))
    const us2v r = __builtin_bit_cast(us2v, __builtin_amdgcn_perm(0u, q, 0x0C030C02u));    // (I(x+1), I(x+2))
    ScharrRow o;
    o.d = r - l;
    o.s4 = (l + r) * us2v{12, 12} + c * us2v{40, 40};
    o.g = __builtin_bit_cast(uint32_t, c) * (1u << GRAY_SHIFT) + ROUND_PAIR;                // the pre-scaled gray pair of the row
    return o;
}
__device__ __forceinline__ uint32_t scharr_dx(const ScharrRow &a, const ScharrRow &b, const ScharrRow &c)
{
    return __builtin_bit_cast(uint32_t, (a.d + c.d) * us2v{12, 12} + b.d * us2v{40, 40});
}
__device__ __forceinline__ uint32_t scharr_dy(const ScharrRow &a, const ScharrRow &c)
{
    return __builtin_bit_cast(uint32_t, c.s4 - a.s4);
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

// Two such sums in one DPP chain (gfx950 v_permlane32_swap): the upper half-wave of `a` changes places with the lower half-wave of `b`,
// so that lanes 0 .. 31 hold a[l] + a[l + 32] and lanes 32 .. 63 b[l - 32] + b[l]; four in-row steps and one row_bcast:15 leave the
// totals in lanes 31 and 63. 9 VALU instructions for two sums instead of 14 (integer sums: any order gives the same bits).
__device__ __forceinline__ void wave_sum_pair_small(int a, int b, int &sum_a, int &sum_b)
{
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)a, (unsigned)b, false, false);
    int v = (int)r[0] + (int)r[1];
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);    // quad_perm:[1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);    // quad_perm:[2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);   // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);   // row_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1,3
    sum_a = __builtin_amdgcn_readlane(v, 31);
    sum_b = __builtin_amdgcn_readlane(v, 63);
}

// Exact sum of per-lane int32 partials as the nearest float (== (float)(int64 sum)): the low 16
// bits and the signed high part are reduced separately, recombined exactly in f64, rounded once.
__device__ __forceinline__ float wave_sum_wide(int v)
{
    const int lo = wave_sum_small(v & 0xFFFF);
    const int hi = wave_sum_small(v >> 16);
    return (float)((double)hi * 65536.0 + (double)lo);
}

// all |v| < 2^24 across the wavefront  =>  any 64-lane sum of them fits int32
__device__ __forceinline__ bool all_small(int a, int b, int c)
{
    const unsigned C = 1u << 24;
    const unsigned t = ((unsigned)a + C) | ((unsigned)b + C) | ((unsigned)c + C);
    return __builtin_amdgcn_ballot_w64(t >= 2 * C) == 0;
}

// cv: iw00 = cvRound((1-a)(1-b) 2^14), iw01 = cvRound(a (1-b) 2^14), iw10 = cvRound((1-a) b 2^14), iw11 = 2^14 - the rest.
// Scaling by 2^14 is exact in binary32 (no product here is subnormal), so it is applied to (1-b) and b once instead of to the
// three products; cvRound (round-half-even) of 0 <= x <= 2^14 is the low half of the bit pattern of x + 1.5 * 2^23 (the add
// rounds to the integer grid, ties to even), which is also where v_perm picks the packed weights from: 18 VALU, was 28.
__device__ __forceinline__ void bilinear_weights(float a, float b, uint32_t &wA, uint32_t &wB)
{
    const float MAGIC = 12582912.f;                        // 0x4B400000
    const float a1 = 1.f - a, b1 = 1.f - b;
    const float b1s = b1 * (float)(1 << W_BITS), bs = b * (float)(1 << W_BITS);
    const uint32_t f00 = __float_as_uint(a1 * b1s + MAGIC), f01 = __float_as_uint(a * b1s + MAGIC), f10 = __float_as_uint(a1 * bs + MAGIC);
    const uint32_t i11 = (uint32_t)(1 << W_BITS) + 3u * 0x4B400000u - (f00 + f01 + f10);   // mod 2^32: the three biases cancel
    wA = lo16_pair(f00, f01);    // top row weights (iw00, iw01)
    wB = lo16_pair(f10, i11);    // bottom row weights (iw10, iw11)
}

template <int TSX, int TSY, int MX, int MY, int WPS, bool RAGGED = false>
__global__ __launch_bounds__(64, WPS) void klt_kernel(KltArgs a)
{
    static_assert(TSX % 4 == 0 && TSX + 4 <= RW && TSX >= 32 + MX + 3 && TSY >= 32 + MY, "tile must hold the window at any alignment");
    // one dword per pixel: (J[x] << 7) | (J[x+1] << 23); +1 row for the unused 17th row of half 1
    __shared__ __attribute__((aligned(16))) uint32_t jt[(TSY + 1) * TSX];
    __shared__ __attribute__((aligned(16))) uint32_t region[TSY * (RW / 4)];   // raw bytes, border tiles only

    const int pt = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int lane = threadIdx.x;
    const int half = lane >> 5, cx = lane & 31;
    const bool col_ok = cx < WIN;
    const PyrLayout &L = a.L;

    const int pair = pt / a.pts_per_pair;
    const int sp = a.prev_slots[pair], sn = a.next_slots[pair];
    const uint8_t *prev_base = a.slab + (long long)sp * L.slot_bytes;
    const uint8_t *next_base = a.slab + (long long)sn * L.slot_bytes;

    float2 pp = a.prev_xy[pt];
    // ragged batch (its own instance of the kernel: even this one select costs the 5-wave build 3 more spilled registers, +2.7 %
    // time, which the uniform batches should not pay): the padding behind a pair's points is given a position no level contains,
    // so it falls through every level (status 0)
    if constexpr (RAGGED) {
        if (pt - pair * a.pts_per_pair >= a.pts_in_pair[pair]) pp = make_float2(-1.0e6f, -1.0e6f);
    }
    float2 guess = make_float2(0.f, 0.f);
    if (a.use_init) guess = a.next_xy[pt];

    // fast tile staging: lane (st_lr, st_c) = row-in-pass, 4-pixel group; lanes beyond RPI * G repeat the last item
    // (r06: 8-pixel groups, 16-byte loads -- G = 5 lanes per tile row, 12 rows per pass, 3 passes: half the vector-memory instructions
    //  of r01 .. r05's 4-pixel groups / 8-byte loads / 6 passes. The kernel is co-limited by the rate at which a CU's 20 waves get
    //  their VMEM instructions through address processing, profiles/r06/klt_planar_levels_ab.txt)
    constexpr int G = TSX / 8, RPI = 64 / G, NIT = TSY / RPI;
    static_assert(TSX % 8 == 0 && TSX + 8 <= RW, "8-pixel staging groups: the load footprint is TSX + 8 columns");
    static_assert(TSY % RPI == 0, "whole passes: no row clamp in the staging loop");
    const float half_win = (float)(WIN - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (float)(1 << 20);
    // uniform: pinned to SGPRs (the f64 -> f32 conversions are VALU instructions; left alone their results occupy two VGPRs for the
    // whole kernel, and the 5-waves-per-SIMD build has none to spare)
    const float eps_lo = a.eps_lo, eps_hi = a.eps_hi;
    int st = 1;
    float errv = 0.f;
    float nx = 0.f, ny = 0.f;

    for (int level = L.levels - 1; level >= 0; --level) {
        const float lscale = __uint_as_float((127u - (unsigned)level) << 23);      // 2^-level exactly (levels < 127): no division sequence
        float px = pp.x * lscale, py = pp.y * lscale;
        if (level == L.levels - 1) {
            if (a.use_init) { nx = guess.x * lscale; ny = guess.y * lscale; }
            else { nx = px; ny = py; }
        } else {
            nx = nx * 2.f; ny = ny * 2.f;
        }
        px -= half_win; py -= half_win;
        // wave-uniform by construction: v_readfirstlane moves them to SGPRs so that the address and
        // border arithmetic derived from them runs on the scalar unit (the kernel is VALU-issue bound)
        const int ipx = __builtin_amdgcn_readfirstlane((int)floorf(px));
        const int ipy = __builtin_amdgcn_readfirstlane((int)floorf(py));
        const int w = L.w[level], h = L.h[level];
        if (ipx < -WIN || ipx >= w || ipy < -WIN || ipy >= h) {
            if (level == 0) { st = 0; errv = 0.f; }
            continue;
        }

        // (r05, measured and dropped: touching the lines the NEXT finer level will ask for -- template rows at their exact position, the
        //  J tile around twice this level's estimate -- with global_load_lds into a dump area while this level runs: 1.397 against
        //  1.251 ms per launch, profiles/r05/klt_prefetch_ab.txt. The per-level round trip is not what the kernel waits for.)
        gptr_u8 Ig, Jg;
        int Igs, Jgs;
        if (level == 0) {
            Ig = as_global(a.l0_ptr[sp]); Igs = a.l0_stride[sp];
            Jg = as_global(a.l0_ptr[sn]); Jgs = a.l0_stride[sn];
        } else {
            Ig = as_global(prev_base + L.goff[level]); Jg = as_global(next_base + L.goff[level]);
            Igs = Jgs = L.gstride[level];
        }
        Igs = __builtin_amdgcn_readfirstlane(Igs); Jgs = __builtin_amdgcn_readfirstlane(Jgs);   // wave-uniform: keep them in SGPRs
        const gptr_u32 Id = (gptr_u32)as_global(prev_base + L.doff[level]);
        const int Ids = __builtin_amdgcn_readfirstlane(L.dstride[level]);
        // buffer descriptors of the level's planes and the byte offset of the plane's origin inside them (see make_rsrc)
        const __amdgpu_buffer_rsrc_t rI = make_rsrc(level == 0 ? a.l0_ptr[sp] : prev_base), rJ = make_rsrc(level == 0 ? a.l0_ptr[sn] : next_base),
                                     rD = make_rsrc(prev_base);
        const int g0 = level == 0 ? 0 : (int)L.goff[level], d0 = (int)L.doff[level], j0 = g0;
        // levels with a physical 32-px border (REFLECT_101 gray, zero gradients in memory, see PyrLayout): every window the
        // range test above lets through lies inside the padded rectangle, so the border-free paths serve all of them
        const bool padded = L.pad[level] != 0;                                             // wave-uniform
        // a level without a stored gradient plane: the template's gradients come from the gray rows (scharr_row above)
        const bool fly = level < L.grad_from;                                              // wave-uniform

        // ---- J tile bookkeeping (declared before the template: on the common path the first tile of the level is requested
        // together with the template rows, one memory round trip per level instead of three) ----
        float cxn = nx - half_win, cyn = ny - half_win;
        int tox = -(1 << 28), toy = -(1 << 28);
        // (r06) the staging lane constants are re-derived per level from an opaque copy of the lane id: hoisted to the kernel prologue
        // they were the three VGPRs the 5-wave build kept in scratch (12 VALU per level against 206 MB of scratch write-back per launch)
        int st_lane = lane;
        asm volatile("" : "+v"(st_lane));
        const int st_ln = min(st_lane, RPI * G - 1), st_lr = (int)(__umul24((unsigned)st_ln, (unsigned)((65536 + G - 1) / G)) >> 16), st_c = st_ln - st_lr * G;
        const int st_loff = st_lr * TSX + 8 * st_c;
        const unsigned st_goff = __umul24((unsigned)st_lr, (unsigned)Jgs) + 8u * (unsigned)st_c;   // lane part of the staging loads
        const bool j_aligned = ((reinterpret_cast<uintptr_t>(Jg) | (uintptr_t)Jgs) & 3u) == 0;
        // origin of the tile that holds window (ix, iy); true when the plain coalesced staging loop can load it
        auto plan_tile = [&](int ix, int iy, int &ox, int &oy) -> bool {
            ox = (ix - MX) & ~3; oy = iy - MY;
            // keep the (TSX+4) x TSY load footprint inside the image whenever the window allows it, so
            // only windows that really cross the border take the reflecting path
            const int bp = padded ? PYR_PAD : 0;                 // the rectangle that exists in memory: [-bp, w + bp) x [-bp, h + bp)
            const int cx0 = min(max(ox, -bp), (w + bp - TSX - 8) & ~3), cy0 = min(max(oy, -bp), h + bp - TSY);
            if (w + 2 * bp >= TSX + 8 && h + 2 * bp >= TSY && ix >= cx0 && ix + 32 <= cx0 + TSX && iy >= cy0 && iy + 32 <= cy0 + TSY) {
                ox = cx0; oy = cy0;
            }
            // padded level: where the clamp above cannot hold the window (right edge, ix > w + bp - 36) the load footprint runs a
            // few bytes past the row into the next row / the slab slack; those tile cells are never read by a window lane
            return j_aligned && (padded || (ox >= 0 && ox + TSX + 8 <= w && oy >= 0 && oy + TSY <= h));
        };
        // G = TSX / 4 lanes cover a tile row (8 source bytes -> 4 pixel pairs each), RPI = 64 / G rows per pass: pass i of
        // lane (lr, c) is row i * RPI + lr. The per-pass part of both addresses is scalar (row base) resp. an immediate
        // (LDS offset); the per-lane part (st_goff, st_loff) is formed once per level. r01 stepped a flat item index
        // through the tile instead: ~9 VALU of index arithmetic per item, 163 per staging against ~9 per pass here.
        auto tile_request = [&](uint4 (&raw)[NIT]) {
            int tb = j0 + toy * Jgs + tox, tstep = RPI * Jgs;
            asm volatile("" : "+s"(tb), "+s"(tstep));
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rJ, st_goff, tb + i * tstep, 0);   // (scalar row offset: one s_add per pass)
                raw[i] = make_uint4(v.x, v.y, v.z, v.w);
            }
        };
        auto tile_commit = [&](const uint4 (&raw)[NIT]) {
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                uint4 *dst = reinterpret_cast<uint4 *>(&jt[st_loff + i * RPI * TSX]);
                dst[0] = make_uint4(scaled_pair<0>(raw[i].x, raw[i].y), scaled_pair<1>(raw[i].x, raw[i].y),
                                    scaled_pair<2>(raw[i].x, raw[i].y), scaled_pair<3>(raw[i].x, raw[i].y));
                dst[1] = make_uint4(scaled_pair<0>(raw[i].y, raw[i].z), scaled_pair<1>(raw[i].y, raw[i].z),
                                    scaled_pair<2>(raw[i].y, raw[i].z), scaled_pair<3>(raw[i].y, raw[i].z));
            }
        };

        // ---- template patch: bilinear samples of I and dI into registers, A = sum(dI dI^T) ----
        // packed int16 pairs (window rows 2m, 2m+1 of this lane): value, x-gradient, y-gradient
        uint32_t Ivp[HALF_ROWS / 2], IXp[HALF_ROWS / 2], IYp[HALF_ROWS / 2];
        int sA11 = 0, sA12 = 0, sA22 = 0;
        {
            uint32_t wA, wB;
            bilinear_weights(px - (float)ipx, py - (float)ipy, wA, wB);
            if (!col_ok) { wA = 0; wB = 0; }          // zero weights => every sample of this lane is 0
            const int xa = ipx + cx;
            // border-free paths: every byte the lanes load exists. The loads span columns ipx .. ipx + 32 and rows ipy .. ipy + 32
            // (one more on each side when the gradients are formed here). Level 0 may be the caller's buffer -- nothing is
            // read past its last row or column; levels >= 1 live in the slab, where the one row / column beyond the image that
            // the zero-weight lanes touch is slab memory.
            const int in_lo = fly ? 1 : 0, in_hi = fly ? 34 : level == 0 ? 33 : 32;
            const bool inside = padded || (ipx >= in_lo && ipx + in_hi <= w && ipy >= in_lo && ipy + in_hi <= h);   // wave-uniform
            // per-lane part of every address; the per-row part is a scalar base (no VALU per load)
            // (on a padded level ipx / ipy may be negative: the signed part of every address is the scalar row base, the lane part
            // -- an unsigned VGPR offset of the saddr addressing mode -- stays >= 0)
            const unsigned vg = __umul24((unsigned)(half * HALF_ROWS), (unsigned)Igs) + (unsigned)cx;      // rows and strides < 2^24:
            const unsigned vd = __umul24((unsigned)(half * HALF_ROWS), (unsigned)Ids) + (unsigned)cx;      // full-rate v_mul_u32_u24
            const unsigned vd4 = 4u * vd;                                                                  // ... as the byte offset of the buffer loads

            // DESCALE(s, 9) == (128 s + 32768) >> 16 and DESCALE(s, 14) == (4 s + 32768) >> 16: the
            // inputs are pre-scaled by 128 / 4 and carry the +2 that sums to 32768, so each sample is
            // the high half of one dot2 chain started from 0. Rows r0 .. r0 + nk of the pre-scaled packed pairs -> window rows
            // r0 .. r0 + nk - 1 (r0 even).
            int vi_prev = 0, vx_prev = 0, vy_prev = 0;
            // ginit: 0 for gradients that carry the rounding constant (stored planes: 4 d + 2), 32768 for the ones formed here (4 d)
            auto window_row = [&](int k, int ginit, uint32_t g0, uint32_t g1, uint32_t dx0, uint32_t dx1, uint32_t dy0, uint32_t dy1) {
                const uint32_t wAk = (k == HALF_ROWS - 1 && half) ? 0u : wA;   // row 31 does not exist
                const uint32_t wBk = (k == HALF_ROWS - 1 && half) ? 0u : wB;
                const int vi = dot2(g1, wBk, dot2(g0, wAk, 0));
                const int vx = dot2(dx1, wBk, dot2(dx0, wAk, ginit));       // (zero weights: 32768 -> high half 0, as before)
                const int vy = dot2(dy1, wBk, dot2(dy0, wAk, ginit));
                if (k & 1) {
                    const int m = k >> 1;
                    Ivp[m] = hi16_pair((uint32_t)vi_prev, (uint32_t)vi);
                    IXp[m] = hi16_pair((uint32_t)vx_prev, (uint32_t)vx);
                    IYp[m] = hi16_pair((uint32_t)vy_prev, (uint32_t)vy);
                    sA11 = dot2(IXp[m], IXp[m], sA11);
                    sA12 = dot2(IXp[m], IYp[m], sA12);
                    sA22 = dot2(IYp[m], IYp[m], sA22);
                }
                vi_prev = vi; vx_prev = vx; vy_prev = vy;
            };

            // Where will the first iteration look? If that window's tile can be loaded by the plain staging loop, its loads are
            // issued FIRST, then all 17 template rows (63 VGPRs in flight): by the time the template rows are back the tile is
            // too, and the level has cost one HBM round trip. (A tile wasted on a level that fails the eigenvalue test is rare.)
            bool early = false;
            if (inside && a.max_count > 0) {
                const int inx0 = __builtin_amdgcn_readfirstlane((int)floorf(cxn)), iny0 = __builtin_amdgcn_readfirstlane((int)floorf(cyn));
                if (!(inx0 < -WIN || inx0 >= w || iny0 < -WIN || iny0 >= h)) {
                    int ox, oy;
                    early = plan_tile(inx0, iny0, ox, oy);
                    if (early) { tox = ox; toy = oy; }
                }
            }
            if (early && fly) {
                constexpr int NQ = HALF_ROWS + 3;        // source rows -1 .. 17 of this half
                uint4 raw[NIT];
                uint32_t q[NQ];
                tile_request(raw);
                int qb = g0 + (ipy - 1) * Igs + ipx - 1, qstep = Igs;                          // scalar: one s_add per row below
                asm volatile("" : "+s"(qb), "+s"(qstep));                                        // (opaque: see the gradient rows below)
#pragma unroll
                for (int r = 0; r < NQ; ++r) q[r] = __builtin_amdgcn_raw_buffer_load_b32(rI, vg, qb + r * qstep, 0);
                __builtin_amdgcn_sched_barrier(0);       // everything in flight before the first use
                tile_commit(raw);
                ScharrRow ra = scharr_row(q[0]), rb = scharr_row(q[1]), rc = scharr_row(q[2]);
                uint32_t g0 = rb.g, dx0 = scharr_dx(ra, rb, rc), dy0 = scharr_dy(ra, rc);
#pragma unroll
                for (int k = 0; k < HALF_ROWS; ++k) {
                    ra = rb; rb = rc; rc = scharr_row(q[k + 3]);
                    const uint32_t g1 = rb.g, dx1 = scharr_dx(ra, rb, rc), dy1 = scharr_dy(ra, rc);
                    window_row(k, 32768, g0, g1, dx0, dx1, dy0, dy1);
                    g0 = g1; dx0 = dx1; dy0 = dy1;
                }
            } else if (early) {
                constexpr int NR = HALF_ROWS + 1;
                uint4 raw[NIT];
                uint16_t graw[NR];
                uint2 draw[NR];
                tile_request(raw);
                int gb = g0 + ipy * Igs + ipx, gstep = Igs;
                int db = d0 + 4 * (ipy * Ids + ipx), dstep = 4 * Ids;                            // scalar: one s_add per load below
                asm volatile("" : "+s"(gb), "+s"(gstep), "+s"(db), "+s"(dstep));                 // (opaque: hipcc re-derives 4 * ((ipy + r) * Ids + ipx) per row otherwise: 3 scalar instructions per load)
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    graw[r] = __builtin_amdgcn_raw_buffer_load_b16(rI, vg, gb + r * gstep, 0);
                    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rD, vd4, db + r * dstep, 0);
                    draw[r] = make_uint2(v.x, v.y);
                }
                __builtin_amdgcn_sched_barrier(0);       // everything in flight before the first use
                tile_commit(raw);
                uint32_t g0 = scaled_pair<0>((uint32_t)graw[0], 0u), dx0 = lo16_pair(draw[0].x, draw[0].y), dy0 = hi16_pair(draw[0].x, draw[0].y);
#pragma unroll
                for (int k = 0; k < HALF_ROWS; ++k) {
                    const uint32_t g1 = scaled_pair<0>((uint32_t)graw[k + 1], 0u);
                    const uint32_t dx1 = lo16_pair(draw[k + 1].x, draw[k + 1].y), dy1 = hi16_pair(draw[k + 1].x, draw[k + 1].y);
                    window_row(k, 0, g0, g1, dx0, dx1, dy0, dy1);
                    g0 = g1; dx0 = dx1; dy0 = dy1;
                }
            } else {
            const unsigned gxa = (unsigned)reflect101(xa, w), gxb = (unsigned)reflect101(xa + 1, w);
            const bool ina = (unsigned)xa < (unsigned)w, inb = (unsigned)(xa + 1) < (unsigned)w;
            // The 17 source rows are processed in two batches of 9 (rows 0-8, 8-16): all loads of a
            // batch are issued before the first use (one memory round trip each) while the in-flight
            // registers stay within the budget of the border path (4 loads per row).
            constexpr int NB = HALF_ROWS / 2 + 1;
#pragma unroll
            for (int batch = 0; batch < 2; ++batch) {
                const int r0 = batch * (HALF_ROWS / 2);
                uint32_t gp[NB], dxp[NB], dyp[NB];     // pre-scaled packed pairs of columns (xa, xa+1)
                if (fly) {
                    // source rows r0 - 1 .. r0 + NB of this half as dwords of the gray bytes at columns xa - 1 .. xa + 2
                    uint32_t q[NB + 2];
                    if (inside) {
#pragma unroll
                        for (int r = 0; r < NB + 2; ++r) q[r] = __builtin_amdgcn_raw_buffer_load_b32(rI, vg, g0 + (ipy + r0 - 1) * Igs + ipx - 1 + r * Igs, 0);
                    } else {
                        // virtual border: BORDER_REFLECT_101 gray for the samples AND for the stencils of the pixels inside the
                        // image (what cv::calcSharrDeriv sees on the padded level); gradients outside the image are 0
                        const unsigned gxl = (unsigned)reflect101(xa - 1, w), gxr = (unsigned)reflect101(xa + 2, w);
                        uint32_t b0[NB + 2], b1[NB + 2], b2[NB + 2], b3[NB + 2];
#pragma unroll
                        for (int r = 0; r < NB + 2; ++r) {
                            const int ry0 = ipy + r0 - 1 + r, ry1 = ry0 + HALF_ROWS;                      // uniform
                            const unsigned go0 = __builtin_amdgcn_readfirstlane((unsigned)reflect101(ry0, h) * (unsigned)Igs);
                            const unsigned go1 = __builtin_amdgcn_readfirstlane((unsigned)reflect101(ry1, h) * (unsigned)Igs);
                            const unsigned go = half ? go1 : go0;
                            b0[r] = Ig[go + gxl]; b1[r] = Ig[go + gxa]; b2[r] = Ig[go + gxb]; b3[r] = Ig[go + gxr];
                        }
                        __builtin_amdgcn_sched_barrier(0);   // keep the whole batch in flight before the first use
#pragma unroll
                        for (int r = 0; r < NB + 2; ++r) q[r] = b0[r] | (b1[r] << 8) | (b2[r] << 16) | (b3[r] << 24);
                    }
                    ScharrRow ra = scharr_row(q[0]), rb = scharr_row(q[1]);
#pragma unroll
                    for (int r = 0; r < NB; ++r) {
                        const ScharrRow rc = scharr_row(q[r + 2]);
                        gp[r] = rb.g; dxp[r] = scharr_dx(ra, rb, rc); dyp[r] = scharr_dy(ra, rc);
                        ra = rb; rb = rc;
                    }
                    if (!inside) {
#pragma unroll
                        for (int r = 0; r < NB; ++r) {
                            const bool rin0 = (unsigned)(ipy + r0 + r) < (unsigned)h, rin1 = (unsigned)(ipy + r0 + r + HALF_ROWS) < (unsigned)h;
                            const bool rin = half ? rin1 : rin0;
                            const uint32_t keep = ((ina && rin) ? 0x0000FFFFu : 0u) | ((inb && rin) ? 0xFFFF0000u : 0u);
                            dxp[r] &= keep;                                       // border value 0 (no rounding constant in this form)
                            dyp[r] &= keep;
                        }
                    }
                } else if (inside) {
                    uint16_t graw[NB];
                    uint2 draw[NB];
#pragma unroll
                    for (int r = 0; r < NB; ++r) {
                        graw[r] = __builtin_amdgcn_raw_buffer_load_b16(rI, vg, g0 + (ipy + r0) * Igs + ipx + r * Igs, 0);
                        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rD, vd4, d0 + 4 * ((ipy + r0) * Ids + ipx) + r * (4 * Ids), 0);
                        draw[r] = make_uint2(v.x, v.y);
                    }
#pragma unroll
                    for (int r = 0; r < NB; ++r) {
                        gp[r] = scaled_pair<0>((uint32_t)graw[r], 0u);
                        dxp[r] = lo16_pair(draw[r].x, draw[r].y);
                        dyp[r] = hi16_pair(draw[r].x, draw[r].y);
                    }
                } else {
                    // virtual borders: BORDER_REFLECT_101 for gray, zero for the gradients; loads stay
                    // unconditional (clamped addresses) and are masked afterwards. Row offsets / masks
                    // are computed for both half-waves on the scalar unit and selected per lane.
                    uint32_t ga[NB], gb[NB], da[NB], db[NB];
#pragma unroll
                    for (int r = 0; r < NB; ++r) {
                        const int ry0 = ipy + r0 + r, ry1 = ry0 + HALF_ROWS;                              // uniform
                        // readfirstlane pins the four products to the scalar unit: left alone, hipcc turns the select
                        // of two products into a per-lane (quarter-rate) v_mul_lo_u32 of the selected row index
                        const unsigned go0 = __builtin_amdgcn_readfirstlane((unsigned)reflect101(ry0, h) * (unsigned)Igs);
                        const unsigned go1 = __builtin_amdgcn_readfirstlane((unsigned)reflect101(ry1, h) * (unsigned)Igs);
                        const unsigned do0 = __builtin_amdgcn_readfirstlane((unsigned)min(max(ry0, 0), h - 1) * (unsigned)Ids);
                        const unsigned do1 = __builtin_amdgcn_readfirstlane((unsigned)min(max(ry1, 0), h - 1) * (unsigned)Ids);
                        const unsigned go = half ? go1 : go0, dof = half ? do1 : do0;
                        ga[r] = Ig[go + gxa]; gb[r] = Ig[go + gxb];
                        da[r] = Id[dof + gxa]; db[r] = Id[dof + gxb];
                    }
                    __builtin_amdgcn_sched_barrier(0);   // keep the whole batch in flight before the first use
#pragma unroll
                    for (int r = 0; r < NB; ++r) {
                        const bool rin0 = (unsigned)(ipy + r0 + r) < (unsigned)h, rin1 = (unsigned)(ipy + r0 + r + HALF_ROWS) < (unsigned)h;
                        const bool rin = half ? rin1 : rin0;
                        gp[r] = ((ga[r] << GRAY_SHIFT) | (gb[r] << (16 + GRAY_SHIFT))) + ROUND_PAIR;
                        const uint32_t qa = (ina && rin) ? da[r] : ROUND_PAIR;     // border value 0 is stored as 4*0 + 2
                        const uint32_t qb = (inb && rin) ? db[r] : ROUND_PAIR;
                        dxp[r] = lo16_pair(qa, qb);
                        dyp[r] = hi16_pair(qa, qb);
                    }
                }
#pragma unroll
                for (int kk = 0; kk < HALF_ROWS / 2; ++kk)
                    window_row(r0 + kk, fly ? 32768 : 0, gp[kk], gp[kk + 1], dxp[kk], dxp[kk + 1], dyp[kk], dyp[kk + 1]);
            }
            }
        }
        float A11, A12, A22;
        // per-lane sums of squares are < 2^29; when all are < 2^24 one int32 chain each is exact
        if (all_small(sA11, sA12, sA22)) {
            int t11, t12;
            wave_sum_pair_small(sA11, sA12, t11, t12);
            A11 = (float)t11 * FLT_SCALE;
            A12 = (float)t12 * FLT_SCALE;
            A22 = (float)wave_sum_small(sA22) * FLT_SCALE;
        } else {
            A11 = wave_sum_wide(sA11) * FLT_SCALE;
            A12 = wave_sum_wide(sA12) * FLT_SCALE;
            A22 = wave_sum_wide(sA22) * FLT_SCALE;
        }

        float D = A11 * A22 - A12 * A12;
        // cv: minEig = (A22 + A11 - sqrt((A11-A22)^2 + 4 A12^2)) / (2 win^2) < minEigThreshold. Only the comparison is live
        // (the value would go to err under OPTFLOW_LK_GET_MIN_EIGENVALS, which the reference does not set), so it is decided
        // from the raw v_sqrt_f32 (1 ulp, no denormal fix-up, no division) unless the numerator lies within a few ulps of
        // its operands of the threshold; only then the correctly rounded sqrt and the IEEE division of the oracle run.
        const float tr = A22 + A11, disc = (A11 - A22) * (A11 - A22) + 4.f * A12 * A12;
        const float num_fast = tr - __builtin_amdgcn_sqrtf(disc), bound = a.eig_bound;
        const float margin = 4e-6f * fabsf(tr) + a.eig_margin + 1e-30f;
        bool weak;
        if (fabsf(num_fast - bound) > margin) weak = num_fast < bound;
        else weak = (tr - sqrtf(disc)) / (float)(2 * WIN * WIN) < a.min_eig;
        if (weak || D < FLT_EPSILON) {
            if (level == 0) st = 0;
            continue;
        }
        D = 1.f / D;

        float pdx = 0.f, pdy = 0.f;

        // Stage the J tile covering window origin (ix, iy) with MX / MY px of slack. A single wavefront
        // issues its LDS accesses in order, so no workgroup barrier is needed.
        auto ensure_tile = [&](int ix, int iy) {
            if (ix >= tox && ix + 32 <= tox + TSX && iy >= toy && iy + 32 <= toy + TSY) return;
            const bool fast = plan_tile(ix, iy, tox, toy);
            if (fast) {
                uint4 raw[NIT];
                tile_request(raw);
                tile_commit(raw);
            } else if (w >= RW && h >= TSY) {
                // The window crosses the image border. Every pixel the virtual (reflected) tile needs
                // lies in an RW x TS in-image region next to that border: copy the region with plain
                // coalesced loads into LDS, then build the tile by reflected LDS byte reads (row index
                // on the scalar unit, column index once per lane) -- ~3x fewer VALU than reflecting
                // every global load.
                const int rx0 = min(max(tox, 0), w - RW), ry0 = min(max(toy, 0), h - TSY);
                int ln = lane;
                asm volatile("" : "+v"(ln));
                {
                    uint32_t raw[(TSY * (RW / 4) + 63) / 64];
#pragma unroll
                    for (int i = 0; i < (TSY * (RW / 4) + 63) / 64; ++i) {
                        const int e = min(ln + 64 * i, TSY * (RW / 4) - 1), r = (int)(__umul24((unsigned)e, 5462u) >> 16), c = e - __umul24(r, RW / 4);   // e / 12, exact for e < 8190
                        __builtin_memcpy(&raw[i], (const void *)(uintptr_t)(Jg + (__umul24((unsigned)(ry0 + r), (unsigned)Jgs) +
                                                                               (unsigned)(rx0 + 4 * c))), 4);
                    }
#pragma unroll
                    for (int i = 0; i < (TSY * (RW / 4) + 63) / 64; ++i) {
                        const int e = min(ln + 64 * i, TSY * (RW / 4) - 1);
                        region[e] = raw[i];
                    }
                }
                const int c = min(ln, TSX - 1);                                    // lanes 0..TSX-1: one tile column each
                const uint8_t *rb = reinterpret_cast<const uint8_t *>(region);
                const int sx0 = reflect101(tox + c, w) - rx0, sx1 = reflect101(tox + c + 1, w) - rx0;
#pragma unroll 4
                for (int r = 0; r < TSY; ++r) {
                    const int sr = (reflect101(toy + r, h) - ry0) * RW;              // uniform
                    const uint32_t v = (((uint32_t)rb[sr + sx0] | ((uint32_t)rb[sr + sx1] << 16)) << GRAY_SHIFT) + ROUND_PAIR;
                    if (lane < TSX) jt[r * TSX + c] = v;
                }
            } else {
                int ln = lane;
                asm volatile("" : "+v"(ln));
#pragma unroll
                for (int i = 0; i < (TSY * TSX / 4 + 63) / 64; ++i) {
                    const int e = min(ln + 64 * i, TSY * TSX / 4 - 1), r = e / (TSX / 4), c = e - r * (TSX / 4);
                    const gptr_u8 row = Jg + __umul24((unsigned)reflect101(toy + r, h), (unsigned)Jgs);
                    const int x = tox + 4 * c;
                    uint32_t b[5];
#pragma unroll
                    for (int q = 0; q < 5; ++q) b[q] = ((uint32_t)row[(unsigned)reflect101(x + q, w)] << GRAY_SHIFT) + 2u;
                    *reinterpret_cast<uint4 *>(&jt[__umul24(r, TSX) + 4 * c]) =
                        make_uint4(b[0] | (b[1] << 16), b[1] | (b[2] << 16), b[2] | (b[3] << 16), b[3] | (b[4] << 16));
                }
            }
        };

        for (int j = 0; j < a.max_count; ++j) {
            const int inx = __builtin_amdgcn_readfirstlane((int)floorf(cxn));
            const int iny = __builtin_amdgcn_readfirstlane((int)floorf(cyn));
            if (inx < -WIN || inx >= w || iny < -WIN || iny >= h) {
                if (level == 0) st = 0;
                break;
            }
            int sb1 = 0, sb2 = 0;
            uint32_t wA, wB;
            // (r05, measured and dropped: requesting the 17 tile dwords of a lane's rows together in front of the arithmetic and keeping them
            //  while the integer window origin stays -- 128 VGPRs, 4 waves per SIMD -- ran 1.351 ms per launch against 1.324 for this loop at 4
            //  waves and 1.256 at 5: the loop is not waiting for LDS, profiles/r05/klt_rows_in_registers_ab.txt)
            ensure_tile(inx, iny);
            bilinear_weights(cxn - (float)inx, cyn - (float)iny, wA, wB);

            // (rows < 64: the 24-bit multiply is full rate, the v_mul_lo_u32 hipcc picks for a 32-bit product a quarter)
            const uint32_t *jrow = jt + __umul24((unsigned)(iny - toy + half * HALF_ROWS), (unsigned)TSX) + (unsigned)(inx - tox + cx);
            uint32_t jp = jrow[0];
#pragma unroll
            for (int m = 0; m < HALF_ROWS / 2; ++m) {
                const uint32_t r1 = jrow[(2 * m + 1) * TSX], r2 = jrow[(2 * m + 2) * TSX];
                const int v0 = dot2(r1, wB, dot2(jp, wA, 0));
                const int v1 = dot2(r2, wB, dot2(r1, wA, 0));
                const uint32_t dp = pk_sub16(hi16_pair((uint32_t)v0, (uint32_t)v1), Ivp[m]);   // |diff| <= 8160
                sb1 = dot2(dp, IXp[m], sb1);      // lanes / rows outside the window have IX = IY = 0
                sb2 = dot2(dp, IYp[m], sb2);
                jp = r2;
            }
            float b1, b2;
            if (all_small(sb1, sb2, 0)) {
                int t1, t2;
                wave_sum_pair_small(sb1, sb2, t1, t2);
                b1 = (float)t1 * FLT_SCALE;
                b2 = (float)t2 * FLT_SCALE;
            } else {
                b1 = wave_sum_wide(sb1) * FLT_SCALE;
                b2 = wave_sum_wide(sb2) * FLT_SCALE;
            }
            const float dx = (A12 * b2 - A22 * b1) * D;
            const float dy = (A12 * b1 - A11 * b2) * D;

            cxn += dx; cyn += dy;
            nx = cxn + half_win; ny = cyn + half_win;

            // cv: delta.ddot(delta) <= epsilon, evaluated in double. The f32 value decides unless it
            // falls within 1e-5 (relative) of the threshold -- f32 error is < 2e-7.
            const float s2 = dx * dx + dy * dy;
            bool converged;
            if (s2 > eps_hi) converged = false;
            else if (s2 < eps_lo) converged = true;
            else converged = (double)dx * (double)dx + (double)dy * (double)dy <= a.epsilon;
            if (converged) break;
            // cv: |float sum| < 0.01 (double). 0.01f < 0.01, so for a float x: x < 0.01 <=> x <= 0.01f
            if (j > 0 && fabsf(dx + pdx) <= 0.01f && fabsf(dy + pdy) <= 0.01f) {
                nx -= dx * 0.5f; ny -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }

        // ---- level-0 epilogue (err requested, MIN_EIGENVALS flag unset): may clear status ----
        if (level == 0 && st) {
            const float ex = nx - half_win, ey = ny - half_win;
            const int inx = __builtin_amdgcn_readfirstlane((int)floorf(ex));
            const int iny = __builtin_amdgcn_readfirstlane((int)floorf(ey));
            if (inx < -WIN || inx >= w || iny < -WIN || iny >= h) {
                st = 0;
            } else if (a.err != nullptr) {
                // the reference discards err (optical_flow.cpp:26 "ignored"); only its side effect on
                // status above is live, so the sum is skipped when the caller passes no err array
                ensure_tile(inx, iny);
                uint32_t wA, wB;
                bilinear_weights(ex - (float)inx, ey - (float)iny, wA, wB);
                // (rows < 64: the 24-bit multiply is full rate, the v_mul_lo_u32 hipcc picks for a 32-bit product a quarter)
            const uint32_t *jrow = jt + __umul24((unsigned)(iny - toy + half * HALF_ROWS), (unsigned)TSX) + (unsigned)(inx - tox + cx);
                uint32_t jp = jrow[0];
                int sabs = 0;
                // pixel (cx, half*16 + k) exists for k < 16 (half 0) / k < 15 (half 1) and cx < 31. Derived here from an opaque copy
                // of the lane id: hoisted to the kernel prologue these masks stay live across the whole level loop
                int ln = lane;
                asm volatile("" : "+v"(ln));
                const bool ok_e = (ln & 31) < WIN;
                const uint32_t ones = ok_e ? 0x00010001u : 0u;
                const uint32_t last_ok = (ok_e && ln < 32) ? 1u : 0u;
#pragma unroll
                for (int m = 0; m < HALF_ROWS / 2; ++m) {
                    const uint32_t r1 = jrow[(2 * m + 1) * TSX], r2 = jrow[(2 * m + 2) * TSX];
                    const int v0 = dot2(r1, wB, dot2(jp, wA, 0));
                    const int v1 = dot2(r2, wB, dot2(r1, wA, 0));
                    const short2v d = __builtin_bit_cast(short2v, pk_sub16(hi16_pair((uint32_t)v0, (uint32_t)v1), Ivp[m]));
                    const uint32_t ad = __builtin_bit_cast(uint32_t, __builtin_elementwise_abs(d));
                    const uint32_t msk = (m == HALF_ROWS / 2 - 1) ? ((ones & 0xFFFFu) | (last_ok << 16)) : ones;
                    sabs = dot2(ad, msk, sabs);
                    jp = r2;
                }
                // sum |diff| <= 961 * 8160 < 2^24: the oracle's float accumulation is exact too
                errv = (float)wave_sum_small(sabs) * 1.f / (float)(32 * WIN * WIN);
            }
        }
    }

    if (lane == 0) {
        a.next_xy[pt] = make_float2(nx, ny);
        a.status[pt] = (uint8_t)st;
        if (a.err != nullptr) a.err[pt] = errv;
    }
}

}  // namespace

int launch_klt(Ctx *c, int n_pairs, const int *prev_slots_dev, const int *next_slots_dev,
               int pts_per_pair, int n_points, const float *prev_xy, float *next_xy,
               uint8_t *status, float *err, int use_initial_flow, int max_iter, const int *pts_in_pair_dev)
{
    (void)n_pairs;
    if (n_points <= 0) return HV_OK;
    KltArgs a{};
    a.L = c->L;
    a.slab = c->slab;
    a.l0_ptr = c->d_l0_ptr; a.l0_stride = c->d_l0_stride;
    a.prev_slots = prev_slots_dev; a.next_slots = next_slots_dev;
    a.pts_per_pair = pts_per_pair; a.n_points = n_points; a.pts_in_pair = pts_in_pair_dev;
    a.prev_xy = reinterpret_cast<const float2 *>(prev_xy);
    a.next_xy = reinterpret_cast<float2 *>(next_xy);
    a.status = status; a.err = err;
    a.use_init = use_initial_flow ? 1 : 0;
    int mc = max_iter; if (mc < 0) mc = 0; if (mc > 100) mc = 100;   // cv: clamp(maxCount, 0, 100)
    a.max_count = mc;
    double e = c->p.eps; if (e < 0.) e = 0.; if (e > 10.) e = 10.;   // cv: clamp(eps, 0, 10)^2
    a.epsilon = e * e;
    a.min_eig = (float)c->p.min_eig;
    a.eps_lo = (float)(a.epsilon * (1.0 - 1e-5)); a.eps_hi = (float)(a.epsilon * (1.0 + 1e-5));
    a.eig_bound = a.min_eig * (float)(2 * WIN * WIN);
    a.eig_margin = 1e-5f * a.eig_bound;
    ScopedKernelTime tm(c, HV_K_KLT);
    // knob klt_tile (HV_KLT_TILE at hv_create; experiments only): 1 = the 40 x 36 tile (slack 2, 1 on the low side; 6 staging passes of 6 rows) compiled
    // for 4 waves per SIMD, 5 (default, and every other value) = the same for 5 waves per SIMD (96 VGPRs; its 7.6 KB of LDS allow 20 waves per CU).
    // r01 .. r03's other tile shapes (44 x 40, 40 x 42) measured slower and are gone
    const int tile_variant = c->knob.klt_tile;
    if (pts_in_pair_dev)        hipLaunchKernelGGL((klt_kernel<40, 36, 2, 1, 5, true>), dim3((unsigned)n_points), dim3(64), 0, c->stream, a);
    else if (tile_variant == 1) hipLaunchKernelGGL((klt_kernel<40, 36, 2, 1, 4>), dim3((unsigned)n_points), dim3(64), 0, c->stream, a);
    else if (tile_variant == 5) hipLaunchKernelGGL((klt_kernel<40, 36, 2, 1, 5>), dim3((unsigned)n_points), dim3(64), 0, c->stream, a);
    else                        hipLaunchKernelGGL((klt_kernel<40, 36, 2, 1, 5>), dim3((unsigned)n_points), dim3(64), 0, c->stream, a);
    HV_HIP(c, hipGetLastError());
    return HV_OK;
}

}  // namespace hv
