// valu_rate.hip -- issue cost of the VALU instructions the DP kernels are made of, measured on one SIMD's stream:
// 8 independent accumulators per lane, REP x 8 instructions inside an unrolled loop, timed with s_memtime.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

#define OPS(X) X(pk_max_u16, "v_pk_max_u16 %0, %0, %1") X(pk_add_u16_clamp, "v_pk_add_u16 %0, %0, %1 clamp") \
	X(pk_sub_u16_clamp, "v_pk_sub_u16 %0, %0, %1 clamp") X(max_u32, "v_max_u32 %0, %0, %1") X(add_u32, "v_add_u32 %0, %0, %1") \
	X(perm_b32, "v_perm_b32 %0, %0, %1, %1") X(or_b32, "v_or_b32 %0, %0, %1") X(max3_i32, "v_max3_i32 %0, %0, %1, %1") \
	X(pk_mul_lo_u16, "v_pk_mul_lo_u16 %0, %0, %1") X(add3_u32, "v_add3_u32 %0, %0, %1, %1") X(pk_max_i16, "v_pk_max_i16 %0, %0, %1") \
	X(pk_add_u16_dep, "v_pk_add_u16 %0, %0, %1")

constexpr int ITERS = 2000;

#define KERNEL(name, text)                                                                                              \
	__global__ void k_##name(uint32_t *out, long long *cyc, int dep)                                                    \
	{                                                                                                                   \
		uint32_t a[8], b = threadIdx.x + 1;                                                                             \
		for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 7 + i;                                                         \
		long long t0 = __builtin_amdgcn_s_memtime();                                                                    \
		for (int it = 0; it < ITERS; ++it) {                                                                            \
			if (dep) {                                                                                                  \
				_Pragma("unroll") for (int i = 0; i < 64; ++i) asm volatile(text : "+v"(a[0]) : "v"(b));             \
			} else {                                                                                                    \
				_Pragma("unroll") for (int i = 0; i < 64; ++i) asm volatile(text : "+v"(a[i & 7]) : "v"(b));         \
			}                                                                                                           \
		}                                                                                                               \
		long long t1 = __builtin_amdgcn_s_memtime();                                                                    \
		uint32_t s = 0;                                                                                                 \
		for (int i = 0; i < 8; ++i) s += a[i];                                                                          \
		out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                                 \
		if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                                \
	}
OPS(KERNEL)

int main()
{
	uint32_t *out;
	long long *cyc;
	hipMalloc(&out, 64 << 20);
	hipMalloc(&cyc, 1 << 20);
	hipEvent_t e0, e1;
	hipEventCreate(&e0), hipEventCreate(&e1);
	hipDeviceProp_t prop;
	hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount;
	const double ghz = prop.clockRate * 1e-6;
	printf("%d CUs, %.2f GHz; cycles per wave-instruction per SIMD (all SIMDs busy)\n", cus, ghz);
	printf("%-20s %10s %10s %10s %10s %10s %10s\n", "instruction", "1w indep", "1w dep", "2w indep", "2w dep", "3w indep", "4w indep");
#define RUN(name, text)                                                                                                 \
	{                                                                                                                   \
		double r[6];                                                                                                    \
		for (int m = 0; m < 6; ++m) {                                                                                   \
			const int wps = m < 2 ? 1 : m < 4 ? 2 : m - 1, dep = m < 4 ? (m & 1) : 0;                                                                 \
			const int blocks = cus * 4, threads = 256 * wps; /* 4 blocks/CU x (4 or 8) waves... one block per CU below */ \
			(void)blocks;                                                                                               \
			hipLaunchKernelGGL(k_##name, dim3(cus), dim3(threads), 0, 0, out, cyc, dep);                                \
			hipEventRecord(e0, 0);                                                                                      \
			hipLaunchKernelGGL(k_##name, dim3(cus), dim3(threads), 0, 0, out, cyc, dep);                                \
			hipEventRecord(e1, 0);                                                                                      \
			hipEventSynchronize(e1);                                                                                    \
			float ms;                                                                                                   \
			hipEventElapsedTime(&ms, e0, e1);                                                                           \
			/* each SIMD ran wps waves x ITERS x 64 instructions */                                                     \
			r[m] = ms * 1e-3 * ghz * 1e9 / ((double)wps * ITERS * 64.0);                                                \
		}                                                                                                               \
		printf("%-20s %10.2f %10.2f %10.2f %10.2f %10.2f %10.2f\n", #name, r[0], r[1], r[2], r[3], r[4], r[5]);                                   \
	}
	OPS(RUN)
	return 0;
}
