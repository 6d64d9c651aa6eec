// valu_dep.hip -- second look at VALU issue on gfx950: every loop body is ONE asm statement (.rept blocks), so that hipcc's hazard
// recognizer cannot put an s_nop between consecutive statements as it does in valu_mix.hip (there every instruction came with one, which
// is where that file's "8.25 cycles per wave-instruction at one wave per SIMD" comes from: 4 for the instruction, 4 for the s_nop).
// Questions: (1) issue interval of ONE wave, for a dependent chain and for independent instructions; (2) SIMD-level cost per class with
// W resident waves and no s_nop in the stream; (3) what the dependency structure of a DP cell costs (the fused extension cell against
// two cells interleaved).
// Every SIMD runs W waves (LDS-enforced occupancy as in valu_mix.hip), 256 instructions per loop trip.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_dep valu_dep.hip ; run: ./valu_dep [filter]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>

// %0..%3 accumulators (in/out), %4 %5 operands; each pattern is 4 instructions, repeated 64 times per trip
#define PATTERNS(X)                                                                                                      \
	X(add_dep, 4, "v_add_u32 %0, %0, %4\n v_add_u32 %0, %0, %5\n v_add_u32 %0, %0, %4\n v_add_u32 %0, %0, %5")           \
	X(add_ind4, 4, "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4")          \
	X(add_ind2, 4, "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %0, %0, %5\n v_add_u32 %1, %1, %5")          \
	X(max16_dep, 4, "v_max_i16 %0, %0, %4\n v_max_i16 %0, %0, %5\n v_max_i16 %0, %0, %4\n v_max_i16 %0, %0, %5")         \
	X(max16_ind4, 4, "v_max_i16 %0, %0, %4\n v_max_i16 %1, %1, %4\n v_max_i16 %2, %2, %4\n v_max_i16 %3, %3, %4")        \
	X(sub16c_ind4, 4, "v_sub_u16_e64 %0, %0, %4 clamp\n v_sub_u16_e64 %1, %1, %4 clamp\n v_sub_u16_e64 %2, %2, %4 clamp\n v_sub_u16_e64 %3, %3, %4 clamp") \
	X(bitop3_ind4, 4, "v_bitop3_b32 %0, %0, %4, %5 bitop3:0xf8\n v_bitop3_b32 %1, %1, %4, %5 bitop3:0xf8\n v_bitop3_b32 %2, %2, %4, %5 bitop3:0xf8\n v_bitop3_b32 %3, %3, %4, %5 bitop3:0xf8") \
	X(max32_dep, 4, "v_max_i32 %0, %0, %4\n v_max_i32 %0, %0, %5\n v_max_i32 %0, %0, %4\n v_max_i32 %0, %0, %5")         \
	X(max32_ind4, 4, "v_max_i32 %0, %0, %4\n v_max_i32 %1, %1, %4\n v_max_i32 %2, %2, %4\n v_max_i32 %3, %3, %4")        \
	X(lshlor_ind4, 4, "v_lshl_or_b32 %0, %0, 1, %4\n v_lshl_or_b32 %1, %1, 1, %4\n v_lshl_or_b32 %2, %2, 1, %4\n v_lshl_or_b32 %3, %3, 1, %4") \
	X(perm_dep, 4, "v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %0, %0, %4, %5") \
	X(perm_ind4, 4, "v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5") \
	X(sdwa_dep, 4, "v_add_u32_sdwa %0, sext(%4), %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:WORD_0\n v_add_u32_sdwa %0, sext(%4), %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:WORD_0\n v_add_u32_sdwa %0, sext(%4), %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:WORD_0\n v_add_u32_sdwa %0, sext(%4), %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:WORD_0") \
	X(sdwa_ind4, 4, "v_add_u32_sdwa %0, sext(%4), %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:WORD_0\n v_add_u32_sdwa %1, sext(%4), %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:WORD_0\n v_add_u32_sdwa %2, sext(%4), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:WORD_0\n v_add_u32_sdwa %3, sext(%4), %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:WORD_0") \
	X(sdwahi_ind4, 4, "v_max_i16_sdwa %0, %4, %5 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0\n v_max_i16_sdwa %1, %4, %5 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0\n v_max_i16_sdwa %2, %4, %5 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0\n v_max_i16_sdwa %3, %4, %5 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0") \
	X(pkmax_ind4, 4, "v_pk_max_u16 %0, %0, %4\n v_pk_max_u16 %1, %1, %4\n v_pk_max_u16 %2, %2, %4\n v_pk_max_u16 %3, %3, %4") \
	X(pkmax_dep, 4, "v_pk_max_u16 %0, %0, %4\n v_pk_max_u16 %0, %0, %5\n v_pk_max_u16 %0, %0, %4\n v_pk_max_u16 %0, %0, %5") \
	X(add_nop, 4, "v_add_u32 %0, %0, %4\n s_nop 0\n v_add_u32 %1, %1, %4\n s_nop 0")                                     \
	X(add_nop_dep, 4, "v_add_u32 %0, %0, %4\n s_nop 0\n v_add_u32 %0, %0, %5\n s_nop 0")                                 \
	X(fast_slow, 4, "v_add_u32 %0, %0, %4\n v_max_i32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_max_i32 %3, %3, %4")         \
	X(fast3_slow, 4, "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_max_i32 %3, %3, %4")        \
	X(fast_slow_dep, 4, "v_add_u32 %0, %0, %4\n v_max_i32 %0, %0, %5\n v_add_u32 %0, %0, %4\n v_max_i32 %0, %0, %5")     \
	/* does an s_nop, or grouping the slow ones, bring the 2-cycle rate of the fast class back in a mixed stream? */              \
	X(f3s_nop, 5, "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_max_i32 %3, %3, %4\n s_nop 0")  \
	X(f3s_nop_first, 5, "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n s_nop 0\n v_max_i32 %3, %3, %4") \
	X(f3s_nop2, 6, "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n s_nop 0\n v_max_i32 %3, %3, %4\n s_nop 0") \
	X(f3s_allnop, 8, "v_add_u32 %0, %0, %4\n s_nop 0\n v_add_u32 %1, %1, %4\n s_nop 0\n v_add_u32 %2, %2, %4\n s_nop 0\n v_max_i32 %3, %3, %4\n s_nop 0") \
	X(s2f6, 8, "v_max_i32 %0, %0, %4\n v_max_i32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %2, %2, %5\n v_add_u32 %3, %3, %5\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4") \
	X(s1f7, 8, "v_max_i32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %5\n v_add_u32 %2, %2, %5\n v_add_u32 %3, %3, %5\n v_add_u32 %1, %1, %4") \
	X(s1f15, 16, "v_max_i32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %5\n v_add_u32 %2, %2, %5\n v_add_u32 %3, %3, %5\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %5\n v_add_u32 %2, %2, %5\n v_add_u32 %3, %3, %5\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4") \
	X(f16_mix, 4, "v_add_u32 %0, %0, %4\n v_max_i16 %1, %1, %4\n v_sub_u16_e64 %2, %2, %4 clamp\n v_bitop3_b32 %3, %3, %4, %5 bitop3:0xf8") \
	X(f_vop2_vop3, 4, "v_add_u32 %0, %0, %4\n v_sub_u16_e64 %1, %1, %4 clamp\n v_add_u32 %2, %2, %4\n v_sub_u16_e64 %3, %3, %4 clamp") \
	X(f_lshr_add, 4, "v_lshrrev_b32 %0, 1, %0\n v_add_u32 %1, %1, %4\n v_lshrrev_b32 %2, 1, %2\n v_add_u32 %3, %3, %4") \
	X(cell_nop, 10, "v_add_u32_sdwa %0, sext(%4), %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:WORD_0\n s_nop 0\n v_lshrrev_b32 %2, 16, %1\n v_max_i16 %0, %0, %2\n v_max_i16 %0, %0, %3\n v_sub_u16_e64 %2, %2, %5 clamp\n v_sub_u16_e64 %3, %3, %5 clamp\n v_sub_u16_e64 %1, %0, %5 clamp\n v_max_u16 %2, %2, %1\n v_max_u16 %3, %3, %1") \
	X(cell_fastonly, 8, "v_lshrrev_b32 %2, 16, %1\n v_max_i16 %0, %0, %2\n v_max_i16 %0, %0, %3\n v_sub_u16_e64 %2, %2, %5 clamp\n v_sub_u16_e64 %3, %3, %5 clamp\n v_sub_u16_e64 %1, %0, %5 clamp\n v_max_u16 %2, %2, %1\n v_max_u16 %3, %3, %1") \
	X(s1f7_nopafter, 9, "v_max_i32 %0, %0, %4\n s_nop 0\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %5\n v_add_u32 %2, %2, %5\n v_add_u32 %3, %3, %5\n v_add_u32 %1, %1, %4") \
	X(s1f7_nopbefore, 9, "s_nop 0\n v_max_i32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %5\n v_add_u32 %2, %2, %5\n v_add_u32 %3, %3, %5\n v_add_u32 %1, %1, %4") \
	X(s1f7_nopmid, 9, "v_max_i32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %5\n s_nop 0\n v_add_u32 %2, %2, %5\n v_add_u32 %3, %3, %5\n v_add_u32 %1, %1, %4") \
	X(s1f7_nop1after, 9, "v_max_i32 %0, %0, %4\n s_nop 1\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %5\n v_add_u32 %2, %2, %5\n v_add_u32 %3, %3, %5\n v_add_u32 %1, %1, %4") \
	X(s1f7_salu, 9, "v_max_i32 %0, %0, %4\n s_add_u32 s10, s10, 1\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %5\n v_add_u32 %2, %2, %5\n v_add_u32 %3, %3, %5\n v_add_u32 %1, %1, %4") \
	X(s1f7_smov, 9, "v_max_i32 %0, %0, %4\n s_mov_b32 s10, 0\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %5\n v_add_u32 %2, %2, %5\n v_add_u32 %3, %3, %5\n v_add_u32 %1, %1, %4") \
	X(s2f6_nop, 9, "v_max_i32 %0, %0, %4\n v_max_i32 %1, %1, %4\n s_nop 0\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %2, %2, %5\n v_add_u32 %3, %3, %5\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4") \
	X(s2f6_2nop, 10, "v_max_i32 %0, %0, %4\n s_nop 0\n v_max_i32 %1, %1, %4\n s_nop 0\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %2, %2, %5\n v_add_u32 %3, %3, %5\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4") \
	X(sfsf_nop, 6, "v_max_i32 %0, %0, %4\n s_nop 0\n v_add_u32 %1, %1, %4\n v_max_i32 %2, %2, %4\n s_nop 0\n v_add_u32 %3, %3, %4") \
	X(sf3sf3_nop, 10, "v_max_i32 %0, %0, %4\n s_nop 0\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_max_i32 %0, %0, %4\n s_nop 0\n v_add_u32 %1, %1, %5\n v_add_u32 %2, %2, %5\n v_add_u32 %3, %3, %5") \
	X(sf3sf3_1nop, 9, "v_max_i32 %0, %0, %4\n s_nop 0\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_max_i32 %0, %0, %4\n v_add_u32 %1, %1, %5\n v_add_u32 %2, %2, %5\n v_add_u32 %3, %3, %5") \
	X(s4f12_nop, 17, "v_max_i32 %0, %0, %4\n v_max_i32 %1, %1, %4\n v_max_i32 %2, %2, %4\n v_max_i32 %3, %3, %4\n s_nop 0\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %5\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %5\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %5\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %5\n v_add_u32 %3, %3, %4\n v_add_u32 %1, %1, %5\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %5") \
	/* the arithmetic of one extension cell (bmh_device.h ext_cell, SYM), as one dependent block of 9, and two cells interleaved */ \
	X(cell_seq, 9, "v_add_u32_sdwa %0, sext(%4), %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:WORD_0\n v_lshrrev_b32 %2, 16, %1\n v_max_i16 %0, %0, %2\n v_max_i16 %0, %0, %3\n v_sub_u16_e64 %2, %2, %5 clamp\n v_sub_u16_e64 %3, %3, %5 clamp\n v_sub_u16_e64 %1, %0, %5 clamp\n v_max_u16 %2, %2, %1\n v_max_u16 %3, %3, %1") \
	X(cell_2way, 18, "v_add_u32_sdwa %0, sext(%4), %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:WORD_0\n v_add_u32_sdwa %6, sext(%4), %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:WORD_0\n v_lshrrev_b32 %2, 16, %1\n v_lshrrev_b32 %8, 16, %7\n v_max_i16 %0, %0, %2\n v_max_i16 %6, %6, %8\n v_max_i16 %0, %0, %3\n v_sub_u16_e64 %2, %2, %5 clamp\n v_sub_u16_e64 %8, %8, %5 clamp\n v_sub_u16_e64 %3, %3, %5 clamp\n v_sub_u16_e64 %1, %0, %5 clamp\n v_max_u16 %2, %2, %1\n v_max_u16 %3, %3, %1\n v_max_i16 %6, %6, %3\n v_sub_u16_e64 %3, %3, %5 clamp\n v_sub_u16_e64 %7, %6, %5 clamp\n v_max_u16 %8, %8, %7\n v_max_u16 %3, %3, %7")

constexpr int ITERS = 400;

#define KERNEL(name, n, text)                                                                                           \
	__global__ void __launch_bounds__(256) k_##name(uint32_t *out, long long *cyc)                                      \
	{                                                                                                                   \
		extern __shared__ uint32_t lds_keep[];                                                                          \
		uint32_t a0 = threadIdx.x, a1 = threadIdx.x * 3, a2 = threadIdx.x * 5, a3 = threadIdx.x * 7, b = threadIdx.x * 9 + 1,   \
		         c = threadIdx.x * 11 + 2, d0 = threadIdx.x + 1, d1 = threadIdx.x + 2, d2 = threadIdx.x + 3;                      \
		long long t0 = __builtin_amdgcn_s_memtime();                                                                    \
		long long r0 = __builtin_amdgcn_s_memrealtime();                                                                \
		for (int it = 0; it < ITERS; ++it)                                                                              \
			asm volatile(".rept 64\n " text "\n .endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b), "+v"(c), "+v"(d0), "+v"(d1), "+v"(d2) : : "s10", "scc"); \
		long long t1 = __builtin_amdgcn_s_memtime();                                                                    \
		long long r1 = __builtin_amdgcn_s_memrealtime();                                                                \
		out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + b + c + d0 + d1 + d2;                          \
		if (a0 == 0x12345) lds_keep[threadIdx.x] = a0;                                                                  \
		if (threadIdx.x == 0) cyc[blockIdx.x * 2] = t1 - t0, cyc[blockIdx.x * 2 + 1] = r1 - r0;                         \
	}
PATTERNS(KERNEL)

struct Pat { const char *name; int n; void (*fn)(uint32_t *, long long *); };
#define ENTRY(name, n, text) {#name, n, k_##name},
static const Pat pats[] = {PATTERNS(ENTRY)};

int main(int argc, char **argv)
{
	const char *filter = argc > 1 ? argv[1] : nullptr;
	uint32_t *out;
	long long *cyc;
	hipMalloc(&out, 256 << 20);
	hipMalloc(&cyc, 8 << 20);
	hipDeviceProp_t prop;
	hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount, rounds = 6;
	std::vector<long long> h((size_t)cus * 8 * rounds * 2);
	hipEvent_t e0, e1;
	hipEventCreate(&e0), hipEventCreate(&e1);
	const int lds_kb[9] = {0, 81, 54, 41, 33, 27, 23, 21, 19}; // W blocks of 256 threads (one wave per SIMD) fit a CU's 160 KiB
	printf("# %d CUs.  cyc = shader cycles (s_memtime, median block) per wave-instruction per SIMD at W resident waves per SIMD; ns = wall clock per\n", cus);
	printf("# wave-instruction per SIMD (HIP events over a grid of %d x W blocks per CU).  Loop bodies are single asm statements: no s_nop but the ones written.\n", rounds);
	printf("%-14s", "pattern");
	for (int w = 1; w <= 8; ++w) printf("  cyc W=%d", w);
	for (int w = 1; w <= 8; ++w) printf("   ns W=%d", w);
	printf("    GHz\n");
	for (const Pat &p : pats) {
		if (filter && !strstr(p.name, filter)) continue;
		double r[9], ns[9], ghz = 0;
		for (int wps = 1; wps <= 8; ++wps) {
			const int blocks = cus * wps * rounds;
			const size_t lds = (size_t)lds_kb[wps] * 1024;
			hipFuncSetAttribute((const void *)p.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
			hipLaunchKernelGGL(p.fn, dim3(blocks), dim3(256), lds, 0, out, cyc);
			hipEventRecord(e0, 0);
			hipLaunchKernelGGL(p.fn, dim3(blocks), dim3(256), lds, 0, out, cyc);
			hipEventRecord(e1, 0);
			hipEventSynchronize(e1);
			float ms = 0;
			hipEventElapsedTime(&ms, e0, e1);
			hipMemcpy(h.data(), cyc, sizeof(long long) * 2 * blocks, hipMemcpyDeviceToHost);
			std::vector<double> per(blocks);
			double rt = 0, st = 0;
			for (int b = 0; b < blocks; ++b) per[b] = (double)h[2 * b], st += (double)h[2 * b], rt += (double)h[2 * b + 1];
			std::sort(per.begin(), per.end());
			const double n_inst = (double)ITERS * 64 * p.n;
			r[wps] = per[blocks / 2] / ((double)wps * n_inst);
			ns[wps] = (double)ms * 1e6 / ((double)rounds * wps * n_inst);
			if (wps == 4) ghz = st / rt * 0.1;
		}
		printf("%-14s", p.name);
		for (int w = 1; w <= 8; ++w) printf(" %8.2f", r[w]);
		for (int w = 1; w <= 8; ++w) printf(" %8.2f", ns[w]);
		printf(" %6.2f\n", ghz);
		fflush(stdout);
	}
	return 0;
}
