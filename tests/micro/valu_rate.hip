// Issue cost of the integer vector instructions the coder kernels are made of, on gfx950:
//   hipcc --offload-arch=gfx950 -O2 tests/micro/valu_rate.hip -o build/valu_rate && build/valu_rate
// Each kernel runs REP x 64 copies of one instruction per wave (independent destinations), W waves per SIMD
// on every CU; reported: cycles per wave-instruction per SIMD (s_memtime ticks / instructions issued on it).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define REP 256
#define X4(s) s s s s
#define X16(s) X4(X4(s))
#define X64(s) X4(X16(s))

#define KERNEL(name, body)                                                                  \
    __global__ void __launch_bounds__(1024) name(unsigned long long *out, unsigned seed, int reps)    \
    {                                                                                       \
        unsigned a = threadIdx.x + seed, b = a * 3u + 1u, c = a ^ 0x55u, d = 0, e = 1, f = 2, g = 3, h = 4; \
        unsigned long long q = ((unsigned long long)a << 32) | b, r = 0, s = 0;             \
        unsigned long long t0 = __builtin_readcyclecounter();                               \
        for (int i = 0; i < reps; i++) { body }                                              \
        unsigned long long t1 = __builtin_readcyclecounter();                               \
        if (d + e + f + g + h + (unsigned)r + (unsigned)s == 0x12345u) out[1] = a + b + c + q;  \
        if (threadIdx.x == 0) atomicMax(out, t1 - t0);                                      \
    }

KERNEL(k_add, X16(asm volatile("v_add_u32 %0, %4, %5\n v_add_u32 %1, %4, %5\n v_add_u32 %2, %4, %5\n v_add_u32 %3, %4, %5" : "=v"(d), "=v"(e), "=v"(f), "=v"(g) : "v"(a), "v"(b));))
KERNEL(k_and, X16(asm volatile("v_and_b32 %0, %4, %5\n v_and_b32 %1, %4, %5\n v_and_b32 %2, %4, %5\n v_and_b32 %3, %4, %5" : "=v"(d), "=v"(e), "=v"(f), "=v"(g) : "v"(a), "v"(b));))
KERNEL(k_mov, X16(asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %5\n v_mov_b32 %2, %4\n v_mov_b32 %3, %5" : "=v"(d), "=v"(e), "=v"(f), "=v"(g) : "v"(a), "v"(b));))
KERNEL(k_shl32, X16(asm volatile("v_lshlrev_b32 %0, %4, %5\n v_lshlrev_b32 %1, %4, %5\n v_lshlrev_b32 %2, %4, %5\n v_lshlrev_b32 %3, %4, %5" : "=v"(d), "=v"(e), "=v"(f), "=v"(g) : "v"(a), "v"(b));))
KERNEL(k_shl64, X16(asm volatile("v_lshlrev_b64 %0, %2, %3\n v_lshlrev_b64 %1, %2, %3\n v_lshlrev_b64 %0, %2, %3\n v_lshlrev_b64 %1, %2, %3" : "=v"(r), "=v"(s) : "v"(a), "v"(q));))
KERNEL(k_shr64, X16(asm volatile("v_lshrrev_b64 %0, %2, %3\n v_lshrrev_b64 %1, %2, %3\n v_lshrrev_b64 %0, %2, %3\n v_lshrrev_b64 %1, %2, %3" : "=v"(r), "=v"(s) : "v"(a), "v"(q));))
KERNEL(k_ffbh, X16(asm volatile("v_ffbh_u32 %0, %4\n v_ffbh_u32 %1, %5\n v_ffbh_u32 %2, %4\n v_ffbh_u32 %3, %5" : "=v"(d), "=v"(e), "=v"(f), "=v"(g) : "v"(a), "v"(b));))
KERNEL(k_cndmask, X16(asm volatile("v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %4, %5, vcc\n v_cndmask_b32 %2, %4, %5, vcc\n v_cndmask_b32 %3, %4, %5, vcc" : "=v"(d), "=v"(e), "=v"(f), "=v"(g) : "v"(a), "v"(b) : "vcc");))
KERNEL(k_min, X16(asm volatile("v_min_u32 %0, %4, %5\n v_min_u32 %1, %4, %5\n v_min_u32 %2, %4, %5\n v_min_u32 %3, %4, %5" : "=v"(d), "=v"(e), "=v"(f), "=v"(g) : "v"(a), "v"(b));))
KERNEL(k_lshl_add, X16(asm volatile("v_lshl_add_u32 %0, %4, 3, %5\n v_lshl_add_u32 %1, %4, 3, %5\n v_lshl_add_u32 %2, %4, 3, %5\n v_lshl_add_u32 %3, %4, 3, %5" : "=v"(d), "=v"(e), "=v"(f), "=v"(g) : "v"(a), "v"(b));))
KERNEL(k_add3, X16(asm volatile("v_add3_u32 %0, %4, %5, %6\n v_add3_u32 %1, %4, %5, %6\n v_add3_u32 %2, %4, %5, %6\n v_add3_u32 %3, %4, %5, %6" : "=v"(d), "=v"(e), "=v"(f), "=v"(g) : "v"(a), "v"(b), "v"(c));))
KERNEL(k_mul_lo, X16(asm volatile("v_mul_lo_u32 %0, %4, %5\n v_mul_lo_u32 %1, %4, %5\n v_mul_lo_u32 %2, %4, %5\n v_mul_lo_u32 %3, %4, %5" : "=v"(d), "=v"(e), "=v"(f), "=v"(g) : "v"(a), "v"(b));))
KERNEL(k_perm, X16(asm volatile("v_perm_b32 %0, %4, %5, %6\n v_perm_b32 %1, %4, %5, %6\n v_perm_b32 %2, %4, %5, %6\n v_perm_b32 %3, %4, %5, %6" : "=v"(d), "=v"(e), "=v"(f), "=v"(g) : "v"(a), "v"(b), "v"(c));))
KERNEL(k_alignbit, X16(asm volatile("v_alignbit_b32 %0, %4, %5, %6\n v_alignbit_b32 %1, %4, %5, %6\n v_alignbit_b32 %2, %4, %5, %6\n v_alignbit_b32 %3, %4, %5, %6" : "=v"(d), "=v"(e), "=v"(f), "=v"(g) : "v"(a), "v"(b), "v"(c));))
KERNEL(k_bfe, X16(asm volatile("v_bfe_u32 %0, %4, %5, %6\n v_bfe_u32 %1, %4, %5, %6\n v_bfe_u32 %2, %4, %5, %6\n v_bfe_u32 %3, %4, %5, %6" : "=v"(d), "=v"(e), "=v"(f), "=v"(g) : "v"(a), "v"(b), "v"(c));))
KERNEL(k_pk_add, X16(asm volatile("v_pk_add_u16 %0, %4, %5\n v_pk_add_u16 %1, %4, %5\n v_pk_add_u16 %2, %4, %5\n v_pk_add_u16 %3, %4, %5" : "=v"(d), "=v"(e), "=v"(f), "=v"(g) : "v"(a), "v"(b));))
KERNEL(k_cmp, X16(asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %1, %0\n v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %1, %0" : : "v"(a), "v"(b) : "vcc");))
KERNEL(k_readlane, X16(asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5\n v_readlane_b32 s22, %0, 7\n v_readlane_b32 s23, %1, 9" : : "v"(a), "v"(b) : "s20", "s21", "s22", "s23");))
KERNEL(k_salu, X16(asm volatile("s_and_b64 s[20:21], s[20:21], exec\n s_or_b64 s[22:23], s[22:23], exec\n s_and_b64 s[24:25], s[24:25], exec\n s_or_b64 s[26:27], s[26:27], exec" : : : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc");))
KERNEL(k_mix, X16(asm volatile("v_add_u32 %0, %4, %5\n s_and_b64 s[20:21], s[20:21], exec\n v_and_b32 %1, %4, %5\n s_or_b64 s[22:23], s[22:23], exec" : "=v"(d), "=v"(e), "=v"(f), "=v"(g) : "v"(a), "v"(b) : "s20", "s21", "s22", "s23", "scc");))
KERNEL(k_dpp, X16(asm volatile("v_add_u32_dpp %0, %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %4, %5 row_shr:2 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %4, %5 row_shr:2 row_mask:0xf bank_mask:0xf" : "+v"(d), "+v"(e), "+v"(f), "+v"(g) : "v"(a), "v"(b));))

KERNEL(k_sub, X16(asm volatile("v_sub_u32 %0, %4, %5 \n v_sub_u32 %1, %4, %5 \n v_sub_u32 %2, %4, %5 \n v_sub_u32 %3, %4, %5" : "=v"(d), "=v"(e), "=v"(f), "=v"(g) : "v"(a), "v"(b));))
KERNEL(k_or, X16(asm volatile("v_or_b32 %0, %4, %5 \n v_or_b32 %1, %4, %5 \n v_or_b32 %2, %4, %5 \n v_or_b32 %3, %4, %5" : "=v"(d), "=v"(e), "=v"(f), "=v"(g) : "v"(a), "v"(b));))
KERNEL(k_xor, X16(asm volatile("v_xor_b32 %0, %4, %5 \n v_xor_b32 %1, %4, %5 \n v_xor_b32 %2, %4, %5 \n v_xor_b32 %3, %4, %5" : "=v"(d), "=v"(e), "=v"(f), "=v"(g) : "v"(a), "v"(b));))
KERNEL(k_shr32, X16(asm volatile("v_lshrrev_b32 %0, %4, %5 \n v_lshrrev_b32 %1, %4, %5 \n v_lshrrev_b32 %2, %4, %5 \n v_lshrrev_b32 %3, %4, %5" : "=v"(d), "=v"(e), "=v"(f), "=v"(g) : "v"(a), "v"(b));))
KERNEL(k_max, X16(asm volatile("v_max_u32 %0, %4, %5 \n v_max_u32 %1, %4, %5 \n v_max_u32 %2, %4, %5 \n v_max_u32 %3, %4, %5" : "=v"(d), "=v"(e), "=v"(f), "=v"(g) : "v"(a), "v"(b));))
KERNEL(k_and_or, X16(asm volatile("v_and_or_b32 %0, %4, %5, %6 \n v_and_or_b32 %1, %4, %5, %6 \n v_and_or_b32 %2, %4, %5, %6 \n v_and_or_b32 %3, %4, %5, %6" : "=v"(d), "=v"(e), "=v"(f), "=v"(g) : "v"(a), "v"(b), "v"(c));))
KERNEL(k_lshl_or, X16(asm volatile("v_lshl_or_b32 %0, %4, %5, %6 \n v_lshl_or_b32 %1, %4, %5, %6 \n v_lshl_or_b32 %2, %4, %5, %6 \n v_lshl_or_b32 %3, %4, %5, %6" : "=v"(d), "=v"(e), "=v"(f), "=v"(g) : "v"(a), "v"(b), "v"(c));))
KERNEL(k_max3, X16(asm volatile("v_max3_u32 %0, %4, %5, %6 \n v_max3_u32 %1, %4, %5, %6 \n v_max3_u32 %2, %4, %5, %6 \n v_max3_u32 %3, %4, %5, %6" : "=v"(d), "=v"(e), "=v"(f), "=v"(g) : "v"(a), "v"(b), "v"(c));))
KERNEL(k_mad24, X16(asm volatile("v_mad_u32_u24 %0, %4, %5, %6 \n v_mad_u32_u24 %1, %4, %5, %6 \n v_mad_u32_u24 %2, %4, %5, %6 \n v_mad_u32_u24 %3, %4, %5, %6" : "=v"(d), "=v"(e), "=v"(f), "=v"(g) : "v"(a), "v"(b), "v"(c));))
KERNEL(k_bfi, X16(asm volatile("v_bfi_b32 %0, %4, %5, %6 \n v_bfi_b32 %1, %4, %5, %6 \n v_bfi_b32 %2, %4, %5, %6 \n v_bfi_b32 %3, %4, %5, %6" : "=v"(d), "=v"(e), "=v"(f), "=v"(g) : "v"(a), "v"(b), "v"(c));))
KERNEL(k_xad, X16(asm volatile("v_xad_u32 %0, %4, %5, %6 \n v_xad_u32 %1, %4, %5, %6 \n v_xad_u32 %2, %4, %5, %6 \n v_xad_u32 %3, %4, %5, %6" : "=v"(d), "=v"(e), "=v"(f), "=v"(g) : "v"(a), "v"(b), "v"(c));))
KERNEL(k_or3, X16(asm volatile("v_or3_b32 %0, %4, %5, %6 \n v_or3_b32 %1, %4, %5, %6 \n v_or3_b32 %2, %4, %5, %6 \n v_or3_b32 %3, %4, %5, %6" : "=v"(d), "=v"(e), "=v"(f), "=v"(g) : "v"(a), "v"(b), "v"(c));))
KERNEL(k_add_lshl, X16(asm volatile("v_add_lshl_u32 %0, %4, %5, %6 \n v_add_lshl_u32 %1, %4, %5, %6 \n v_add_lshl_u32 %2, %4, %5, %6 \n v_add_lshl_u32 %3, %4, %5, %6" : "=v"(d), "=v"(e), "=v"(f), "=v"(g) : "v"(a), "v"(b), "v"(c));))
KERNEL(k_pk_lshr, X16(asm volatile("v_pk_lshrrev_b16 %0, %4, %5 \n v_pk_lshrrev_b16 %1, %4, %5 \n v_pk_lshrrev_b16 %2, %4, %5 \n v_pk_lshrrev_b16 %3, %4, %5" : "=v"(d), "=v"(e), "=v"(f), "=v"(g) : "v"(a), "v"(b));))
KERNEL(k_pk_max, X16(asm volatile("v_pk_max_u16 %0, %4, %5 \n v_pk_max_u16 %1, %4, %5 \n v_pk_max_u16 %2, %4, %5 \n v_pk_max_u16 %3, %4, %5" : "=v"(d), "=v"(e), "=v"(f), "=v"(g) : "v"(a), "v"(b));))
KERNEL(k_mul24, X16(asm volatile("v_mul_u32_u24 %0, %4, %5 \n v_mul_u32_u24 %1, %4, %5 \n v_mul_u32_u24 %2, %4, %5 \n v_mul_u32_u24 %3, %4, %5" : "=v"(d), "=v"(e), "=v"(f), "=v"(g) : "v"(a), "v"(b));))
KERNEL(k_ashr, X16(asm volatile("v_ashrrev_i32 %0, %4, %5 \n v_ashrrev_i32 %1, %4, %5 \n v_ashrrev_i32 %2, %4, %5 \n v_ashrrev_i32 %3, %4, %5" : "=v"(d), "=v"(e), "=v"(f), "=v"(g) : "v"(a), "v"(b));))
KERNEL(k_cnd64, X16(asm volatile("v_cndmask_b32 %0, %4, %5, s[20:21]\n v_cndmask_b32 %1, %4, %5, s[20:21]\n v_cndmask_b32 %2, %4, %5, s[22:23]\n v_cndmask_b32 %3, %4, %5, s[22:23]" : "=v"(d), "=v"(e), "=v"(f), "=v"(g) : "v"(a), "v"(b) : "s20", "s21", "s22", "s23");))
KERNEL(k_cmpcnd, X16(asm volatile("v_cmp_lt_u32 vcc, %4, %5\n v_cndmask_b32 %0, %4, %5, vcc\n v_cmp_lt_u32 vcc, %5, %4\n v_cndmask_b32 %1, %4, %5, vcc" : "=v"(d), "=v"(e), "=v"(f), "=v"(g) : "v"(a), "v"(b) : "vcc");))
KERNEL(k_cmp64, X16(asm volatile("v_cmp_lt_u32 s[20:21], %0, %1\n v_cmp_lt_u32 s[22:23], %1, %0\n v_cmp_lt_u32 s[20:21], %0, %1\n v_cmp_lt_u32 s[22:23], %1, %0" : : "v"(a), "v"(b) : "s20", "s21", "s22", "s23");))
KERNEL(k_not, X16(asm volatile("v_not_b32 %0, %4\n v_not_b32 %1, %5\n v_not_b32 %2, %4\n v_not_b32 %3, %5" : "=v"(d), "=v"(e), "=v"(f), "=v"(g) : "v"(a), "v"(b));))
KERNEL(k_bfei, X16(asm volatile("v_bfe_i32 %0, %4, 0, 1\n v_bfe_i32 %1, %5, 0, 1\n v_bfe_i32 %2, %4, 0, 1\n v_bfe_i32 %3, %5, 0, 1" : "=v"(d), "=v"(e), "=v"(f), "=v"(g) : "v"(a), "v"(b));))
KERNEL(k_addchain, X16(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %0, %0, %5\n v_add_u32 %0, %0, %4\n v_add_u32 %0, %0, %5" : "+v"(d) : "v"(e), "v"(f), "v"(a), "v"(b));))
KERNEL(k_shchain, X16(asm volatile("v_lshlrev_b32 %0, %4, %0\n v_lshrrev_b32 %0, %5, %0\n v_lshlrev_b32 %0, %4, %0\n v_lshrrev_b32 %0, %5, %0" : "+v"(d) : "v"(e), "v"(f), "v"(a), "v"(b));))

struct K { const char *name; void (*fn)(unsigned long long *, unsigned, int); };

int main()
{
    unsigned long long *d_out;
    hipMalloc(&d_out, 16);
    K ks[] = {{"v_add_u32", k_add}, {"v_and_b32", k_and}, {"v_mov_b32", k_mov}, {"v_lshlrev_b32", k_shl32},
              {"v_lshlrev_b64", k_shl64}, {"v_lshrrev_b64", k_shr64}, {"v_ffbh_u32", k_ffbh}, {"v_cndmask_b32", k_cndmask},
              {"v_min_u32", k_min}, {"v_lshl_add_u32", k_lshl_add}, {"v_add3_u32", k_add3}, {"v_mul_lo_u32", k_mul_lo},
              {"v_perm_b32", k_perm}, {"v_alignbit_b32", k_alignbit}, {"v_bfe_u32", k_bfe}, {"v_pk_add_u16", k_pk_add},
              {"v_cmp_lt_u32", k_cmp}, {"v_readlane_b32", k_readlane}, {"s_and/or_b64", k_salu},
              {"valu+salu 1:1", k_mix}, {"v_add_u32_dpp", k_dpp}, {"v_sub_u32", k_sub}, {"v_or_b32", k_or}, {"v_xor_b32", k_xor}, {"v_lshrrev_b32", k_shr32}, {"v_max_u32", k_max}, {"v_and_or_b32", k_and_or}, {"v_lshl_or_b32", k_lshl_or}, {"v_max3_u32", k_max3}, {"v_mad_u32_u24", k_mad24}, {"v_bfi_b32", k_bfi}, {"v_xad_u32", k_xad}, {"v_or3_b32", k_or3}, {"v_add_lshl_u32", k_add_lshl}, {"v_pk_lshrrev_b16", k_pk_lshr}, {"v_pk_max_u16", k_pk_max}, {"v_mul_u32_u24", k_mul24}, {"v_ashrrev_i32", k_ashr}, {"v_cndmask sgpr", k_cnd64}, {"cmp+cndmask vcc", k_cmpcnd}, {"v_cmp -> sgpr", k_cmp64}, {"v_not_b32", k_not}, {"v_bfe_i32", k_bfei}, {"v_add dependent", k_addchain}, {"v_shift dependent", k_shchain}};
    // waves per SIMD: 1, 2, 4 = one workgroup of 256 * w threads per CU; 8 = two workgroups of 1024
    const int wps[] = {1, 2, 4, 8};
    printf("%-18s", "waves/SIMD:");
    for (int w : wps) printf("%10d", w);
    printf("   (ns per wave-instruction on one SIMD, wall clock; last column: s_memtime ticks per ns)\n");
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int reps = 4096;
    for (auto &k : ks) {
        printf("%-18s", k.name);
        double tick_rate = 0;
        for (int w : wps) {
            const int wg = w == 8 ? 512 : 256, thr = w == 8 ? 1024 : 256 * w;
            hipLaunchKernelGGL(k.fn, dim3(wg), dim3(thr), 0, 0, d_out, 1u, 16);   // warm
            hipMemset(d_out, 0, 16);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k.fn, dim3(wg), dim3(thr), 0, 0, d_out, 1u, reps);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            unsigned long long ticks = 0;
            hipMemcpy(&ticks, d_out, 8, hipMemcpyDeviceToHost);
            printf("%10.3f", (double)ms * 1e6 / ((double)reps * 64 * w));
            tick_rate = (double)ticks / ((double)ms * 1e6);
        }
        printf("   %6.3f\n", tick_rate);
    }
    return 0;
}
