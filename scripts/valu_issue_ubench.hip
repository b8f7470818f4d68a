// Developer microbenchmark (not part of the library): wave64 ISSUE RATE of the instructions klt_kernel is built from, on gfx950,
// with every CU busy. Answers VERDICT r04 weak #3: is a wave64 VALU instruction 2 or 4 cycles of its SIMD?
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue_ubench valu_issue_ubench.hip && ./valu_issue_ubench
// Each kernel runs REPS x 32 copies of one instruction per wave, either as 8 independent chains ("ind") or as one dependent chain
// ("dep"), on CUS x 4 SIMDs x W waves (one wave per workgroup, so W = waves per SIMD when the launch fills every SIMD evenly).
// Reported: shader cycles per wave-instruction per SIMD = elapsed cycles x (waves per SIMD)^-1 ... i.e.
//     cyc_per_inst_simd = (t1 - t0 of a wave) / (insts per wave x W)          [timer: s_memtime, shader clock]
// and the shader clock itself (s_memtime against the 100 MHz s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <string>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int REPS = 256, UNR = 32;            // 8192 instructions per wave and kernel

#define R4(x) x x x x
#define R8(x) R4(x) R4(x)
#define R32(x) R8(x) R8(x) R8(x) R8(x)

// 8 independent destinations v[0..7]; sources a, b (never written)
#define IND8(OP, SFX) \
    asm volatile(R4(OP " %0, %8, %9" SFX "\n" OP " %1, %8, %9" SFX "\n" OP " %2, %8, %9" SFX "\n" OP " %3, %8, %9" SFX "\n" \
                    OP " %4, %8, %9" SFX "\n" OP " %5, %8, %9" SFX "\n" OP " %6, %8, %9" SFX "\n" OP " %7, %8, %9" SFX "\n") \
                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(a), "v"(b))
#define IND8_3(OP, SFX) \
    asm volatile(R4(OP " %0, %8, %9, %0" SFX "\n" OP " %1, %8, %9, %1" SFX "\n" OP " %2, %8, %9, %2" SFX "\n" OP " %3, %8, %9, %3" SFX "\n" \
                    OP " %4, %8, %9, %4" SFX "\n" OP " %5, %8, %9, %5" SFX "\n" OP " %6, %8, %9, %6" SFX "\n" OP " %7, %8, %9, %7" SFX "\n") \
                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(a), "v"(b))
#define IND8_1(OP, SFX) \
    asm volatile(R4(OP " %0, %8" SFX "\n" OP " %1, %8" SFX "\n" OP " %2, %8" SFX "\n" OP " %3, %8" SFX "\n" \
                    OP " %4, %8" SFX "\n" OP " %5, %8" SFX "\n" OP " %6, %8" SFX "\n" OP " %7, %8" SFX "\n") \
                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(a), "v"(b))
#define DEP32(OP, SFX) asm volatile(R32(OP " %0, %0, %1" SFX "\n") : "+v"(v0) : "v"(a))
#define DEP32_3(OP, SFX) asm volatile(R32(OP " %0, %1, %2, %0" SFX "\n") : "+v"(v0) : "v"(a), "v"(b))

#define KERNEL_HEAD(name) \
    __global__ __launch_bounds__(64) void name(unsigned *out, long long *cyc, long long *wall) { \
        unsigned v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7; \
        unsigned a = 0x00030001u + threadIdx.x, b = 0x00010002u; (void)b; (void)v1; (void)v2; (void)v3; (void)v4; (void)v5; (void)v6; (void)v7; \
        __shared__ unsigned lds[64 * 9]; lds[threadIdx.x] = v0; \
        const unsigned la = 4u * threadIdx.x; (void)la; \
        __builtin_amdgcn_s_sleep(8); \
        const long long w0 = wall_clock64(), t0 = clock64(); \
        for (int r = 0; r < REPS; ++r) {
#define KERNEL_TAIL \
        } \
        const long long t1 = clock64(), w1 = wall_clock64(); \
        out[blockIdx.x * 64 + threadIdx.x] = v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7; \
        if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; wall[blockIdx.x] = w1 - w0; } \
    }

KERNEL_HEAD(k_add_ind)    IND8("v_add_u32", "");                       KERNEL_TAIL
KERNEL_HEAD(k_add_dep)    DEP32("v_add_u32", "");                      KERNEL_TAIL
KERNEL_HEAD(k_fma_ind)    IND8_3("v_fma_f32", "");                     KERNEL_TAIL
KERNEL_HEAD(k_fma_dep)    DEP32_3("v_fma_f32", "");                    KERNEL_TAIL
KERNEL_HEAD(k_pkfma_ind)  { double d0 = v0, d1 = v1, d2 = v2, d3 = v3, da = 1.0 + 1e-9 * a, db = 1e-9; asm volatile(R4("v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3\n"
                                            "v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3\n")
                                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(da), "v"(db)); if (r == REPS - 1) v0 ^= (unsigned)__double2loint(d0 + d1 + d2 + d3); } KERNEL_TAIL
KERNEL_HEAD(k_dot2_ind)   IND8_3("v_dot2_i32_i16", " clamp");          KERNEL_TAIL
KERNEL_HEAD(k_dot2_dep)   DEP32_3("v_dot2_i32_i16", " clamp");         KERNEL_TAIL
KERNEL_HEAD(k_perm_ind)   IND8_3("v_perm_b32", "");                    KERNEL_TAIL
KERNEL_HEAD(k_perm_dep)   DEP32_3("v_perm_b32", "");                   KERNEL_TAIL
KERNEL_HEAD(k_pksub_ind)  IND8("v_pk_sub_i16", "");                    KERNEL_TAIL
KERNEL_HEAD(k_pksub_dep)  DEP32("v_pk_sub_i16", "");                   KERNEL_TAIL
KERNEL_HEAD(k_pkmul_ind)  IND8("v_pk_mul_lo_u16", "");                 KERNEL_TAIL
KERNEL_HEAD(k_lshlor_ind) IND8_3("v_lshl_or_b32", "");                 KERNEL_TAIL
KERNEL_HEAD(k_cnd_ind)    IND8("v_cndmask_b32", ", vcc");              KERNEL_TAIL
KERNEL_HEAD(k_mullo_ind)  IND8("v_mul_lo_u32", "");                    KERNEL_TAIL
KERNEL_HEAD(k_mul24_ind)  IND8("v_mul_u32_u24", "");                   KERNEL_TAIL
KERNEL_HEAD(k_dpp_ind)    IND8("v_add_u32_dpp", " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"); KERNEL_TAIL
KERNEL_HEAD(k_dpp_dep)    DEP32("v_add_u32_dpp", " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"); KERNEL_TAIL
KERNEL_HEAD(k_and_ind)    IND8("v_and_b32", "");                       KERNEL_TAIL
KERNEL_HEAD(k_or_ind)     IND8("v_or_b32", "");                        KERNEL_TAIL
KERNEL_HEAD(k_lshl_ind)   IND8("v_lshlrev_b32", "");                   KERNEL_TAIL
KERNEL_HEAD(k_sub_ind)    IND8("v_sub_u32", "");                       KERNEL_TAIL
KERNEL_HEAD(k_max_ind)    IND8("v_max_i32", "");                       KERNEL_TAIL
KERNEL_HEAD(k_mov_ind)    IND8_1("v_mov_b32", "");                     KERNEL_TAIL
KERNEL_HEAD(k_mulf_ind)   IND8("v_mul_f32", "");                       KERNEL_TAIL
KERNEL_HEAD(k_addf_ind)   IND8("v_add_f32", "");                       KERNEL_TAIL
KERNEL_HEAD(k_cvt_ind)    IND8_1("v_cvt_f32_i32", "");                 KERNEL_TAIL
KERNEL_HEAD(k_mad24_ind)  IND8_3("v_mad_u32_u24", "");                 KERNEL_TAIL
KERNEL_HEAD(k_bfe_ind)    IND8_3("v_bfe_u32", "");                     KERNEL_TAIL
KERNEL_HEAD(k_align_ind)  IND8_3("v_alignbit_b32", "");                KERNEL_TAIL
KERNEL_HEAD(k_add3_ind)   IND8_3("v_add3_u32", "");                    KERNEL_TAIL
KERNEL_HEAD(k_lshladd_ind) IND8_3("v_lshl_add_u32", "");               KERNEL_TAIL
KERNEL_HEAD(k_or3_ind)    IND8_3("v_or3_b32", "");                     KERNEL_TAIL
KERNEL_HEAD(k_pkadd_ind)  IND8("v_pk_add_u16", "");                    KERNEL_TAIL
KERNEL_HEAD(k_pkmad_ind)  IND8_3("v_pk_mad_u16", "");                  KERNEL_TAIL
KERNEL_HEAD(k_cnd64_ind)  IND8("v_cndmask_b32_e64", ", s[4:5]");       KERNEL_TAIL
// the VOP2 form reads VCC implicitly: once with VCC written by a v_cmp in front of every 32, once in the VOP3 encoding naming vcc
KERNEL_HEAD(k_cndvcc_cmp) { asm volatile("v_cmp_gt_u32 vcc, %0, %1" :: "v"(a), "v"(b) : "vcc"); IND8("v_cndmask_b32", ", vcc"); } KERNEL_TAIL
KERNEL_HEAD(k_cnd64_vcc)  IND8("v_cndmask_b32_e64", ", vcc");          KERNEL_TAIL
KERNEL_HEAD(k_addc32)     { asm volatile(R4("v_addc_co_u32_e32 %0, vcc, %8, %9, vcc\n v_addc_co_u32_e32 %1, vcc, %8, %9, vcc\n v_addc_co_u32_e32 %2, vcc, %8, %9, vcc\n v_addc_co_u32_e32 %3, vcc, %8, %9, vcc\n"
                                            "v_addc_co_u32_e32 %4, vcc, %8, %9, vcc\n v_addc_co_u32_e32 %5, vcc, %8, %9, vcc\n v_addc_co_u32_e32 %6, vcc, %8, %9, vcc\n v_addc_co_u32_e32 %7, vcc, %8, %9, vcc\n")
                                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(a), "v"(b) : "vcc"); } KERNEL_TAIL
KERNEL_HEAD(k_addc64)     { asm volatile(R4("v_addc_co_u32_e64 %0, vcc, %8, %9, vcc\n v_addc_co_u32_e64 %1, vcc, %8, %9, vcc\n v_addc_co_u32_e64 %2, vcc, %8, %9, vcc\n v_addc_co_u32_e64 %3, vcc, %8, %9, vcc\n"
                                            "v_addc_co_u32_e64 %4, vcc, %8, %9, vcc\n v_addc_co_u32_e64 %5, vcc, %8, %9, vcc\n v_addc_co_u32_e64 %6, vcc, %8, %9, vcc\n v_addc_co_u32_e64 %7, vcc, %8, %9, vcc\n")
                                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(a), "v"(b) : "vcc"); } KERNEL_TAIL
KERNEL_HEAD(k_addco32)    { asm volatile(R4("v_add_co_u32_e32 %0, vcc, %8, %9\n v_add_co_u32_e32 %1, vcc, %8, %9\n v_add_co_u32_e32 %2, vcc, %8, %9\n v_add_co_u32_e32 %3, vcc, %8, %9\n"
                                            "v_add_co_u32_e32 %4, vcc, %8, %9\n v_add_co_u32_e32 %5, vcc, %8, %9\n v_add_co_u32_e32 %6, vcc, %8, %9\n v_add_co_u32_e32 %7, vcc, %8, %9\n")
                                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(a), "v"(b) : "vcc"); } KERNEL_TAIL
KERNEL_HEAD(k_cmp32)      { asm volatile(R32("v_cmp_gt_u32_e32 vcc, %0, %1\n") :: "v"(a), "v"(b) : "vcc"); } KERNEL_TAIL
KERNEL_HEAD(k_cmp64)      { asm volatile(R32("v_cmp_gt_u32_e64 s[6:7], %0, %1\n") :: "v"(a), "v"(b) : "s6", "s7"); } KERNEL_TAIL
KERNEL_HEAD(k_movdpp_ind) IND8_1("v_mov_b32_dpp", " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"); KERNEL_TAIL
KERNEL_HEAD(k_sdwa_ind)   IND8("v_add_u32_sdwa", " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD"); KERNEL_TAIL
KERNEL_HEAD(k_dot4_ind)   IND8_3("v_dot4_i32_i8", "");                 KERNEL_TAIL
KERNEL_HEAD(k_dot2c_ind)  IND8("v_dot2c_i32_i16", "");                 KERNEL_TAIL
KERNEL_HEAD(k_sad_ind)    IND8_3("v_sad_u16", "");                     KERNEL_TAIL
// 4 VALU + 4 SALU per 8: does scalar work ride along for free? (s_mul_i32: an SALU instruction that leaves SCC alone -- the loop's
// compare sits in front of the asm block, its branch behind)
KERNEL_HEAD(k_valu_salu)  { asm volatile(R8("v_add_u32 %0, %8, %9\n s_mul_i32 s6, s6, s7\n v_add_u32 %1, %8, %9\n s_mul_i32 s7, s7, s6\n"
                                            "v_add_u32 %2, %8, %9\n s_mul_i32 s6, s6, s7\n v_add_u32 %3, %8, %9\n s_mul_i32 s7, s7, s6\n")
                                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(a), "v"(b) : "s6", "s7"); } KERNEL_TAIL
KERNEL_HEAD(k_dot2_salu)  { asm volatile(R8("v_dot2_i32_i16 %0, %8, %9, %0 clamp\n s_mul_i32 s6, s6, s7\n v_dot2_i32_i16 %1, %8, %9, %1 clamp\n s_mul_i32 s7, s7, s6\n"
                                            "v_dot2_i32_i16 %2, %8, %9, %2 clamp\n s_mul_i32 s6, s6, s7\n v_dot2_i32_i16 %3, %8, %9, %3 clamp\n s_mul_i32 s7, s7, s6\n")
                                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(a), "v"(b) : "s6", "s7"); } KERNEL_TAIL
// alternating 2-cycle and 4-cycle class instructions
KERNEL_HEAD(k_add_dot2)   { asm volatile(R8("v_dot2_i32_i16 %0, %8, %9, %0 clamp\n v_add_u32 %4, %8, %9\n v_dot2_i32_i16 %1, %8, %9, %1 clamp\n v_add_u32 %5, %8, %9\n")
                                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(a), "v"(b)); } KERNEL_TAIL
// ds_read_b32 (independent destinations, waits once per 32)
KERNEL_HEAD(k_dsread_ind) { asm volatile(R4("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:256\n ds_read_b32 %2, %8 offset:512\n ds_read_b32 %3, %8 offset:768\n"
                                            "ds_read_b32 %4, %8 offset:1024\n ds_read_b32 %5, %8 offset:1280\n ds_read_b32 %6, %8 offset:1536\n ds_read_b32 %7, %8 offset:1792\n")
                                         "s_waitcnt lgkmcnt(0)\n"
                                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(la)); } KERNEL_TAIL
// klt_kernel's iteration body as the compiler emits it (one pair of window rows = 2 ds_read_b32 + 4 dot2 + perm + pk_sub + 2 dot2),
// four pairs per unrolled block = 8 LDS reads + 32 VALU; the VALU count is what the rate is quoted on
KERNEL_HEAD(k_klt_mix)    { asm volatile(R4("ds_read_b32 %1, %8 offset:160\n ds_read_b32 %2, %8 offset:320\n s_waitcnt lgkmcnt(0)\n"
                                            "v_dot2_i32_i16 %3, %0, %9, 0 clamp\n v_dot2_i32_i16 %4, %1, %9, 0 clamp\n"
                                            "v_dot2_i32_i16 %3, %1, %10, %3 clamp\n v_dot2_i32_i16 %4, %2, %10, %4 clamp\n"
                                            "v_perm_b32 %3, %4, %3, %9\n v_pk_sub_i16 %3, %3, %10\n"
                                            "v_dot2_i32_i16 %5, %3, %9, %5 clamp\n v_dot2_i32_i16 %6, %3, %10, %6 clamp\n v_mov_b32 %0, %2\n")
                                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(la), "v"(a), "v"(b)); } KERNEL_TAIL
// the same with the two LDS reads of the NEXT pair issued before this pair's arithmetic (software pipelined: no exposed LDS latency)
KERNEL_HEAD(k_klt_mix_pipe) { asm volatile(R4("ds_read_b32 %6, %8 offset:160\n ds_read_b32 %7, %8 offset:320\n"
                                            "v_dot2_i32_i16 %3, %0, %9, 0 clamp\n v_dot2_i32_i16 %4, %1, %9, 0 clamp\n"
                                            "v_dot2_i32_i16 %3, %1, %10, %3 clamp\n v_dot2_i32_i16 %4, %2, %10, %4 clamp\n"
                                            "v_perm_b32 %3, %4, %3, %9\n v_pk_sub_i16 %3, %3, %10\n"
                                            "v_dot2_i32_i16 %5, %3, %9, %5 clamp\n v_dot2_i32_i16 %5, %3, %10, %5 clamp\n s_waitcnt lgkmcnt(0)\n v_mov_b32 %0, %2\n v_mov_b32 %1, %6\n v_mov_b32 %2, %7\n")
                                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(la), "v"(a), "v"(b)); } KERNEL_TAIL

typedef void (*kern_t)(unsigned *, long long *, long long *);
struct Case { const char *name; kern_t k; int valu_per_block; };      // VALU instructions per asm block (the quantity the rate is quoted on)

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("# device %s, %d CUs, clockRate %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
    const int max_blocks = cus * 4 * 8;
    unsigned *out; long long *cyc, *wall;
    CHECK(hipMalloc(&out, (size_t)max_blocks * 64 * 4)); CHECK(hipMalloc(&cyc, max_blocks * 8)); CHECK(hipMalloc(&wall, max_blocks * 8));
    std::vector<long long> hc(max_blocks), hw(max_blocks);
    const Case cases[] = {
        {"v_add_u32 ind", k_add_ind, 32}, {"v_add_u32 dep", k_add_dep, 32},
        {"v_fma_f32 ind", k_fma_ind, 32}, {"v_fma_f32 dep", k_fma_dep, 32},
        {"v_pk_fma_f32 ind (2 f32 FMAs per lane)", k_pkfma_ind, 32},
        {"v_dot2_i32_i16 ind", k_dot2_ind, 32}, {"v_dot2_i32_i16 dep", k_dot2_dep, 32},
        {"v_perm_b32 ind", k_perm_ind, 32}, {"v_perm_b32 dep", k_perm_dep, 32},
        {"v_pk_sub_i16 ind", k_pksub_ind, 32}, {"v_pk_sub_i16 dep", k_pksub_dep, 32},
        {"v_pk_mul_lo_u16 ind", k_pkmul_ind, 32}, {"v_lshl_or_b32 ind", k_lshlor_ind, 32},
        {"v_cndmask_b32 ind", k_cnd_ind, 32}, {"v_mul_lo_u32 ind", k_mullo_ind, 32}, {"v_mul_u32_u24 ind", k_mul24_ind, 32},
        {"v_add_u32_dpp ind", k_dpp_ind, 32}, {"v_add_u32_dpp dep", k_dpp_dep, 32},
        {"v_and_b32 ind", k_and_ind, 32}, {"v_or_b32 ind", k_or_ind, 32}, {"v_lshlrev_b32 ind", k_lshl_ind, 32}, {"v_sub_u32 ind", k_sub_ind, 32},
        {"v_max_i32 ind", k_max_ind, 32}, {"v_mov_b32 ind", k_mov_ind, 32}, {"v_mul_f32 ind", k_mulf_ind, 32}, {"v_add_f32 ind", k_addf_ind, 32},
        {"v_cvt_f32_i32 ind", k_cvt_ind, 32}, {"v_mad_u32_u24 ind", k_mad24_ind, 32}, {"v_bfe_u32 ind", k_bfe_ind, 32},
        {"v_alignbit_b32 ind", k_align_ind, 32}, {"v_add3_u32 ind", k_add3_ind, 32}, {"v_lshl_add_u32 ind", k_lshladd_ind, 32},
        {"v_or3_b32 ind", k_or3_ind, 32}, {"v_pk_add_u16 ind", k_pkadd_ind, 32}, {"v_pk_mad_u16 ind", k_pkmad_ind, 32},
        {"v_cndmask_b32_e64 (sgpr pair) ind", k_cnd64_ind, 32}, {"v_cndmask_b32 vcc, v_cmp in front of every 32", k_cndvcc_cmp, 33},
        {"v_cndmask_b32_e64 naming vcc ind", k_cnd64_vcc, 32},
        {"v_addc_co_u32_e32 (carry in and out through VCC)", k_addc32, 32}, {"v_addc_co_u32_e64 naming vcc", k_addc64, 32},
        {"v_add_co_u32_e32 (carry out to VCC)", k_addco32, 32}, {"v_cmp_gt_u32_e32 (writes VCC)", k_cmp32, 32}, {"v_cmp_gt_u32_e64 (writes an SGPR pair)", k_cmp64, 32}, {"v_mov_b32_dpp ind", k_movdpp_ind, 32}, {"v_add_u32_sdwa ind", k_sdwa_ind, 32},
        {"v_dot4_i32_i8 ind", k_dot4_ind, 32}, {"v_dot2c_i32_i16 (VOP2) ind", k_dot2c_ind, 32}, {"v_sad_u16 ind", k_sad_ind, 32},
        {"v_add_u32 + s_mul_i32 alternating (per VALU)", k_valu_salu, 32}, {"v_dot2 + s_mul_i32 alternating (per VALU)", k_dot2_salu, 32},
        {"v_dot2 + v_add_u32 alternating (per VALU)", k_add_dot2, 32},
        {"ds_read_b32 ind (per LDS read)", k_dsread_ind, 32},
        {"klt row-pair mix (8 VALU + 2 ds_read, wait per pair)", k_klt_mix, 36},
        {"klt row-pair mix, LDS reads one pair ahead", k_klt_mix_pipe, 44},
    };
    printf("%-58s", "instruction \\ waves per SIMD");
    const int wps_list[] = {1, 2, 4, 5, 8};
    for (int w : wps_list) printf(" %7d", w);
    printf("   (shader cycles per wave-instruction per SIMD; chip-wide G wave-inst/s at 8 waves)\n");
    for (const Case &c : cases) {
        printf("%-58s", c.name); fflush(stdout);
        double last_rate = 0, mhz = 0;
        for (int wps : wps_list) {
            const int blocks = cus * 4 * wps;
            for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(c.k, dim3(blocks), dim3(64), 0, 0, out, cyc, wall); }
            hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(c.k, dim3(blocks), dim3(64), 0, 0, out, cyc, wall);
            CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            CHECK(hipMemcpy(hc.data(), cyc, blocks * 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hw.data(), wall, blocks * 8, hipMemcpyDeviceToHost));
            std::vector<long long> s(hc.begin(), hc.begin() + blocks);
            std::sort(s.begin(), s.end());
            const double med = (double)s[blocks / 2];
            const double insts = (double)REPS * c.valu_per_block;
            printf(" %7.2f", med / (insts * wps));
            double wsum = 0, csum = 0; for (int i = 0; i < blocks; ++i) { wsum += hw[i]; csum += hc[i]; }
            mhz = csum / wsum * 100.0;                                  // s_memrealtime: 100 MHz
            last_rate = insts * blocks / (ms * 1e-3) * 1e-9;
            CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
        }
        printf("   %8.1f G/s (hipEvent, incl. launch), s_memtime clock %.0f MHz\n", last_rate, mhz);
    }
    return 0;
}
