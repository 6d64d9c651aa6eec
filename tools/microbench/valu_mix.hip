// valu_mix.hip -- issue cost of every VALU instruction class the DP kernels use (and the candidates to replace them), on gfx950.
// Every SIMD of the chip runs W resident waves (W = 1, 2, 3, 4, 8) of the same straight-line stream: 8 independent accumulators, 64
// instructions per loop trip, ITERS trips.  Two clocks: wall time from HIP events (-> cycles at the NOMINAL clock) and s_memtime
// (shader cycles actually spent, whatever the clock the chip holds under this load) / s_memrealtime (100 MHz) -> the held clock.
// Output: one line per instruction: cycles per wave-instruction per SIMD at each W (shader cycles), and the held clock.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_mix valu_mix.hip ; run: ./valu_mix [filter]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <algorithm>

// %0 = accumulator (in/out VGPR), %1 = a second VGPR, %2 = a third VGPR; vcc and s[10:11] are clobbered by all
#define OPS(X)                                                                                                          \
	/* VOP2 / VOP1 plain 32-bit */                                                                                      \
	X(add_u32, "v_add_u32 %0, %0, %1") X(sub_u32, "v_sub_u32 %0, %0, %1") X(subrev_u32, "v_subrev_u32 %0, %0, %1")      \
	X(and_b32, "v_and_b32 %0, %0, %1") X(or_b32, "v_or_b32 %0, %0, %1") X(xor_b32, "v_xor_b32 %0, %0, %1")              \
	X(lshlrev_b32, "v_lshlrev_b32 %0, 1, %0") X(lshrrev_b32, "v_lshrrev_b32 %0, 1, %0") X(ashrrev_i32, "v_ashrrev_i32 %0, 1, %0") \
	X(lshlrev_v, "v_lshlrev_b32 %0, %1, %0")                                                                            \
	X(max_u32, "v_max_u32 %0, %0, %1") X(max_i32, "v_max_i32 %0, %0, %1") X(min_u32, "v_min_u32 %0, %0, %1") X(min_i32, "v_min_i32 %0, %0, %1") \
	X(mov_b32, "v_mov_b32 %0, %1") X(not_b32, "v_not_b32 %0, %0")                                                       \
	X(cndmask_vcc, "v_cndmask_b32 %0, %0, %1, vcc") X(cndmask_sgpr, "v_cndmask_b32 %0, %0, %1, s[10:11]")               \
	X(cmp_vcc, "v_cmp_gt_i32 vcc, %0, %1") X(cmp_sgpr, "v_cmp_gt_i32 s[10:11], %0, %1")   \
	X(cmp_cnd_pair, "v_cmp_gt_i32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc")                                        \
	X(cmp_cnd_far, "v_cmp_gt_i32 vcc, %0, %1\n v_add_u32 %2, %2, %1\n v_cndmask_b32 %0, %0, %1, vcc")                  \
	X(mul_u32_u24, "v_mul_u32_u24 %0, %0, %1") X(mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")                                 \
	X(addc_co, "v_add_co_u32 %0, vcc, %0, %1") X(ffbh, "v_ffbh_u32 %0, %0") X(ffbl, "v_ffbl_b32 %0, %0") X(bcnt, "v_bcnt_u32_b32 %0, %0, %1") \
	X(bfrev, "v_bfrev_b32 %0, %0")                                                                                      \
	/* 16-bit VOP2 */                                                                                                   \
	X(add_u16, "v_add_u16 %0, %0, %1") X(sub_u16, "v_sub_u16 %0, %0, %1") X(max_u16, "v_max_u16 %0, %0, %1") X(max_i16, "v_max_i16 %0, %0, %1") \
	X(min_u16, "v_min_u16 %0, %0, %1") X(lshlrev_b16, "v_lshlrev_b16 %0, 1, %0")                                        \
	/* SDWA / DPP forms */                                                                                              \
	X(add_u32_sdwa, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1")    \
	X(max_i32_sdwa, "v_max_i32_sdwa %0, sext(%0), sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1") \
	X(max_i16_sdwa, "v_max_i16_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1") \
	X(add_u16_sdwa, "v_add_u16_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1") \
	X(mov_dpp_shr1, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf") X(max_dpp_shr1, "v_max_i32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf") \
	X(add_dpp_shr1, "v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")                                    \
	/* VOP3 (64-bit encodings) */                                                                                       \
	X(add_u32_e64, "v_add_u32_e64 %0, %0, %1") X(max_i32_e64, "v_max_i32_e64 %0, %0, %1") X(or_b32_e64, "v_or_b32_e64 %0, %0, %1") \
	X(add3_u32, "v_add3_u32 %0, %0, %1, %2") X(lshl_add_u32, "v_lshl_add_u32 %0, %0, 1, %1") X(add_lshl_u32, "v_add_lshl_u32 %0, %0, %1, 1") \
	X(and_or_b32, "v_and_or_b32 %0, %0, %1, %2") X(or3_b32, "v_or3_b32 %0, %0, %1, %2") X(lshl_or_b32, "v_lshl_or_b32 %0, %0, 1, %1") \
	X(xad_u32, "v_xad_u32 %0, %0, %1, %2") X(bfe_u32, "v_bfe_u32 %0, %0, 1, 8") X(bfe_i32, "v_bfe_i32 %0, %0, 1, 8") X(bfi_b32, "v_bfi_b32 %0, %1, %0, %2") \
	X(alignbit, "v_alignbit_b32 %0, %0, %1, 8") X(alignbyte, "v_alignbyte_b32 %0, %0, %1, 1") X(perm_b32, "v_perm_b32 %0, %0, %1, %2") \
	X(max3_i32, "v_max3_i32 %0, %0, %1, %2") X(max3_u32, "v_max3_u32 %0, %0, %1, %2") X(min3_i32, "v_min3_i32 %0, %0, %1, %2") X(med3_i32, "v_med3_i32 %0, %0, %1, %2") \
	X(max3_i16, "v_max3_i16 %0, %0, %1, %2") X(max3_u16, "v_max3_u16 %0, %0, %1, %2")                                   \
	X(mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2") X(mad_i32_i24, "v_mad_i32_i24 %0, %0, %1, %2") X(sad_u32, "v_sad_u32 %0, %0, %1, %2") \
	X(mad_u16, "v_mad_u16 %0, %0, %1, %2") X(mad_legacy_u16, "v_mad_legacy_u16 %0, %0, %1, %2")                         \
	X(bitop3_b32, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96") X(bitop3_b16, "v_bitop3_b16 %0, %0, %1, %2 bitop3:0x96")   \
	X(add_i32_clamp, "v_add_i32 %0, %0, %1 clamp") X(sub_i32_clamp, "v_sub_i32 %0, %0, %1 clamp") X(add_u32_clamp, "v_add_u32_e64 %0, %0, %1 clamp") \
	X(sub_u32_clamp, "v_sub_u32_e64 %0, %0, %1 clamp") X(add_u16_clamp, "v_add_u16_e64 %0, %0, %1 clamp") X(sub_u16_clamp, "v_sub_u16_e64 %0, %0, %1 clamp") \
	X(add_i16_clamp, "v_add_i16 %0, %0, %1 clamp") X(sub_i16_clamp, "v_sub_i16 %0, %0, %1 clamp")                       \
	X(mbcnt_lo, "v_mbcnt_lo_u32_b32 %0, %1, %0")                              \
	X(cndmask_e64_neg, "v_cndmask_b32_e64 %0, %0, %1, s[10:11]") X(dot4_i32_i8, "v_dot4_i32_i8 %0, %1, %2, %0") X(sad_u8, "v_sad_u8 %0, %0, %1, %2") \
	X(msad_u8, "v_msad_u8 %0, %0, %1, %2") X(lerp_u8, "v_lerp_u8 %0, %0, %1, %2")                                       \
	X(cvt_pk_u16_u32, "v_cvt_pk_u16_u32 %0, %0, %1") X(cvt_pk_i16_i32, "v_cvt_pk_i16_i32 %0, %0, %1") X(pack_b32_f16, "v_pack_b32_f16 %0, %0, %1") \
	/* VOP3P packed 16-bit */                                                                                           \
	X(pk_add_u16, "v_pk_add_u16 %0, %0, %1") X(pk_add_u16_clamp, "v_pk_add_u16 %0, %0, %1 clamp") X(pk_sub_u16_clamp, "v_pk_sub_u16 %0, %0, %1 clamp") \
	X(pk_add_i16, "v_pk_add_i16 %0, %0, %1") X(pk_sub_i16, "v_pk_sub_i16 %0, %0, %1") X(pk_max_i16, "v_pk_max_i16 %0, %0, %1") X(pk_max_u16, "v_pk_max_u16 %0, %0, %1") \
	X(pk_min_i16, "v_pk_min_i16 %0, %0, %1") X(pk_lshlrev_b16, "v_pk_lshlrev_b16 %0, 1, %0") X(pk_ashrrev_i16, "v_pk_ashrrev_i16 %0, 1, %0") \
	X(pk_mad_u16, "v_pk_mad_u16 %0, %0, %1, %2") X(pk_mad_i16, "v_pk_mad_i16 %0, %0, %1, %2") X(pk_mul_lo_u16, "v_pk_mul_lo_u16 %0, %0, %1") \
	X(pk_add_opsel, "v_pk_add_u16 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]")                                             \
	/* float ops as a yardstick (the guide: v_fma_f32 2 cycles with >= 2 waves) */                                      \
	X(add_f32, "v_add_f32 %0, %0, %1") X(fma_f32, "v_fma_f32 %0, %0, %1, %2") X(max_f32, "v_max_f32 %0, %0, %1") \
	X(max3_f32, "v_max3_f32 %0, %0, %1, %2") X(pk_max_f16, "v_pk_max_f16 %0, %0, %1") X(pk_add_f16, "v_pk_add_f16 %0, %0, %1") X(max_f16, "v_max_f16 %0, %0, %1") \
	X(fmac_f32, "v_fmac_f32 %0, %1, %2")                                                                                 \
	/* mixes */                                                                                                         \
	X(mix_add_max, "v_add_u32 %0, %0, %1\n v_max_i32 %0, %0, %2") X(mix_add_perm, "v_add_u32 %0, %0, %1\n v_perm_b32 %0, %0, %1, %2") \
	X(mix_or_pkmax, "v_or_b32 %0, %0, %1\n v_pk_max_u16 %0, %0, %2") X(mix_3add_max, "v_add_u32 %0, %0, %1\n v_sub_u32 %0, %0, %2\n v_and_b32 %0, %0, %1\n v_max_i32 %0, %0, %2") \
	X(snop0, "s_nop 0") X(snop1, "s_nop 1") X(mix_add_snop, "v_add_u32 %0, %0, %1\n s_nop 0")                           \
	X(mix_max_salu, "v_max_i32 %0, %0, %1\n s_add_u32 s10, s10, 1") X(mix_add_salu, "v_add_u32 %0, %0, %1\n s_add_u32 s10, s10, 1") \
	X(salu, "s_add_u32 s10, s10, 1")                                                                                     \
	X(dep1_add, "v_add_u32 %3, %3, %1") X(dep1_max, "v_max_i32 %3, %3, %1") X(dep1_perm, "v_perm_b32 %3, %3, %1, %2")     \
	X(dep2_add, "v_add_u32 %3, %3, %1\n v_add_u32 %4, %4, %1") X(dep2_max, "v_max_i32 %3, %3, %1\n v_max_i32 %4, %4, %1") \
	X(dep1_chain3, "v_add_u32 %3, %3, %1\n v_max_i32 %3, %3, %2\n v_sub_u32 %3, %3, %1")

constexpr int ITERS = 1500;
constexpr int UNROLL = 64;

#define KERNEL(name, text)                                                                                              \
	__global__ void __launch_bounds__(256) k_##name(uint32_t *out, long long *cyc)                                      \
	{                                                                                                                   \
		extern __shared__ uint32_t lds_keep[];                                                                          \
		uint32_t a[8];                                                                                                  \
		uint32_t b = threadIdx.x * 3 + 1, c = threadIdx.x * 5 + 2, d0 = threadIdx.x, d1 = threadIdx.x + 9;                                                      \
		for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 7 + i;                                                         \
		long long t0 = __builtin_amdgcn_s_memtime();                                                                    \
		long long r0 = __builtin_amdgcn_s_memrealtime();                                                                \
		for (int it = 0; it < ITERS; ++it) {                                                                            \
			_Pragma("unroll") for (int i = 0; i < UNROLL; ++i) asm volatile(text : "+v"(a[i & 7]), "+v"(b), "+v"(c), "+v"(d0), "+v"(d1) : : "vcc", "s10", "s11"); \
		}                                                                                                               \
		long long t1 = __builtin_amdgcn_s_memtime();                                                                    \
		long long r1 = __builtin_amdgcn_s_memrealtime();                                                                \
		uint32_t s = b + c + d0 + d1;                                                                                             \
		for (int i = 0; i < 8; ++i) s += a[i];                                       \
		out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                                 \
		if (s == 0x12345) lds_keep[threadIdx.x] = s;                                                                    \
		if (threadIdx.x == 0) {                                                                                         \
			cyc[blockIdx.x * 4] = t1 - t0, cyc[blockIdx.x * 4 + 1] = r1 - r0;                                           \
			cyc[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  /* HW_REG_HW_ID */  \
			cyc[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)); /* HW_REG_XCC_ID */ \
		}                                                                                                               \
	}
OPS(KERNEL)

struct Op { const char *name; const char *text; void (*fn)(uint32_t *, long long *); };
#define ENTRY(name, text) {#name, text, k_##name},
static const Op ops[] = {OPS(ENTRY)};

static int hwid_cu(unsigned id) { return (int)((id >> 8) & 0xf) | (int)((id >> 12) & 1) << 4 | (int)((id >> 13) & 7) << 5; } // cu | sh | se

int main(int argc, char **argv)
{
	const char *filter = argc > 1 ? argv[1] : nullptr;
	uint32_t *out;
	long long *cyc;
	hipMalloc(&out, 256 << 20);
	hipMalloc(&cyc, 8 << 20);
	hipDeviceProp_t prop;
	hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount;
	const int rounds = 6;
	std::vector<long long> h((size_t)cus * 8 * rounds * 4);
	hipEvent_t e0, e1;
	hipEventCreate(&e0), hipEventCreate(&e1);
	// W resident waves per SIMD are ENFORCED through LDS: a block is 256 threads (one wave per SIMD) and asks for so much dynamic LDS
	// that exactly W blocks fit the CU's 160 KiB; the grid holds `rounds` times what the chip can hold, so that in the steady state every
	// CU has W blocks whatever the dispatcher's placement.  Two clocks: per-block s_memtime (median; shader cycles one wave needs with
	// W-1 partners on its SIMD) and HIP events around the whole launch (wall).
	const int lds_kb[9] = {0, 81, 54, 41, 33, 27, 23, 21, 19};
	printf("# %d CUs, nominal %.2f GHz.  cyc = shader cycles (s_memtime, median block) per wave-instruction per SIMD with W resident waves per SIMD (LDS-enforced);\n", cus, prop.clockRate * 1e-6);
	printf("# ns = wall-clock nanoseconds per wave-instruction per SIMD from HIP events over a grid of %d x W blocks per CU; GHz = clock held at W=4\n", rounds);
	printf("%-18s", "instruction");
	for (int w = 1; w <= 8; ++w) printf("  cyc W=%d", w);
	for (int w = 1; w <= 8; ++w) printf("   ns W=%d", w);
	printf("    GHz  n  CUs-used\n");
	for (const Op &op : ops) {
		if (filter && !strstr(op.name, filter)) continue;
		int ninst = 1;
		for (const char *p = op.text; *p; ++p) ninst += *p == '\n';
		double r[9], ns[9], ghz = 0;
		int cu_used = 0;
		for (int wps = 1; wps <= 8; ++wps) {
			const int blocks = cus * wps * rounds;
			const size_t lds = (size_t)lds_kb[wps] * 1024;
			hipFuncSetAttribute((const void *)op.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
			hipLaunchKernelGGL(op.fn, dim3(blocks), dim3(256), lds, 0, out, cyc);  // warm
			hipEventRecord(e0, 0);
			hipLaunchKernelGGL(op.fn, dim3(blocks), dim3(256), lds, 0, out, cyc);
			hipEventRecord(e1, 0);
			hipEventSynchronize(e1);
			float ms = 0;
			hipEventElapsedTime(&ms, e0, e1);
			hipMemcpy(h.data(), cyc, sizeof(long long) * 4 * blocks, hipMemcpyDeviceToHost);
			std::vector<double> per(blocks);
			double rt = 0, st = 0;
			std::vector<int> seen(256 * 8, 0);
			for (int b = 0; b < blocks; ++b) {
				per[b] = (double)h[4 * b], st += (double)h[4 * b], rt += (double)h[4 * b + 1];
				seen[(hwid_cu((unsigned)h[4 * b + 2]) + 256 * (int)(h[4 * b + 3] & 7)) % (256 * 8)] = 1;
			}
			std::sort(per.begin(), per.end());
			r[wps] = per[blocks / 2] / ((double)wps * ITERS * UNROLL * ninst);
			ns[wps] = (double)ms * 1e6 / ((double)rounds * wps * ITERS * UNROLL * ninst);
			if (wps == 4) {
				ghz = st / rt * 0.1;
				for (int v : seen) cu_used += v;
			}
		}
		printf("%-18s", op.name);
		for (int w = 1; w <= 8; ++w) printf(" %8.2f", r[w]);
		for (int w = 1; w <= 8; ++w) printf(" %8.2f", ns[w]);
		printf(" %6.2f %2d %4d\n", ghz, ninst, cu_used);
		fflush(stdout);
	}
	return 0;
}
