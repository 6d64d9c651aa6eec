// extend_dispatch.hip -- routes every extension task to the kernel built for its query length.
//
//   bin 0: qlen <= 64    extend_reg_kernel<1>   (registers, 1 column per lane)
//   bin 1: qlen <= 128   extend_reg_kernel<2>
//   bin 2: qlen <= 256   extend_reg_kernel<4>
//   bin 3: longer, or qlen == 0                                                    -> extend_lds_kernel
//
// The bins are built ON THE DEVICE (one pass over the 32-byte task records, wave-aggregated
// atomics), and every extension kernel reads its bin size from device memory, so a launch needs no
// host round trip and the *_device entry point stays asynchronous on the caller's stream.
#include "bmh_ctx.h"
#include "bmh_device.h"

namespace bmh {

__global__ __launch_bounds__(256) void classify_kernel(const bmh_ext_task_t *__restrict__ tasks,
                                                       const uint32_t *__restrict__ order, long long n,
                                                       uint32_t *__restrict__ counts, uint32_t *__restrict__ lists,
                                                       int reg_ok)
{
	// one atomic per wave and bin: ballot the lanes of each bin, lane 0 of the bin reserves a range
	const int lane = threadIdx.x & 63;
	const long long stride = (long long)gridDim.x * blockDim.x;
	const long long first = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	for (long long k0 = first - lane; k0 < n; k0 += stride) { // k0: wave-uniform base index
		const long long k = k0 + lane;
		int bin = -1;
		uint32_t idx = 0;
		if (k < n) {
			idx = order ? order[k] : (uint32_t)k;
			const int qlen = tasks[idx].qlen;
			bin = (!reg_ok || qlen < 1) ? 3 : qlen <= 64 ? 0 : qlen <= 128 ? 1 : qlen <= 256 ? 2 : 3;
		}
#pragma unroll
		for (int b = 0; b < 4; ++b) {
			const unsigned long long m = __ballot(bin == b);
			if (m == 0) continue;
			uint32_t base = 0;
			if (lane == (int)__builtin_ctzll(m)) base = atomicAdd(&counts[b], (uint32_t)__builtin_popcountll(m));
			base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)__builtin_ctzll(m));
			if (bin == b) lists[(size_t)b * (size_t)n + base + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1))] = idx;
		}
	}
}

int launch_extend(bmh_ctx *ctx, const uint8_t *d_pool, const bmh_ext_task_t *d_tasks, int64_t n,
                  bmh_ext_result_t *d_res, const uint32_t *d_order, int qmax)
{
	if (n <= 0) return BMH_OK;
	int rc;
	const bool reg_ok = ctx->force_kernel != 1;
	if ((rc = ensure(ctx, ctx->d_bins, 64 + (size_t)4 * (size_t)n * 4))) return rc;
	uint32_t *counts = (uint32_t *)ctx->d_bins.p;
	uint32_t *lists = counts + 16;
	BMH_HIP(ctx, hipMemsetAsync(counts, 0, 64, ctx->stream));
	long long cg = (n + 255) / 256;
	if (cg > 2048) cg = 2048;
	hipLaunchKernelGGL(classify_kernel, dim3((unsigned)cg), dim3(256), 0, ctx->stream, d_tasks, d_order, (long long)n, counts,
	                   lists, reg_ok ? 1 : 0);
	BMH_HIP(ctx, hipGetLastError());
	const bool tm = ctx->timing;
	if (tm) BMH_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
	if (tm) BMH_HIP(ctx, hipEventRecord(ctx->ev_bin[0], ctx->stream));
	if (reg_ok && (rc = launch_extend_reg(ctx, 1, d_pool, d_tasks, n, d_res, lists, counts + 0))) return rc;
	if (tm) BMH_HIP(ctx, hipEventRecord(ctx->ev_bin[1], ctx->stream));
	if (reg_ok && qmax > 64 && (rc = launch_extend_reg(ctx, 2, d_pool, d_tasks, n, d_res, lists + (size_t)n, counts + 1))) return rc;
	if (tm) BMH_HIP(ctx, hipEventRecord(ctx->ev_bin[2], ctx->stream));
	if (reg_ok && qmax > 128 && (rc = launch_extend_reg(ctx, 4, d_pool, d_tasks, n, d_res, lists + 2 * (size_t)n, counts + 2)))
		return rc;
	if (tm) BMH_HIP(ctx, hipEventRecord(ctx->ev_bin[3], ctx->stream));
	// bin 3 always gets a launch: it also holds qlen == 0 tasks; an empty bin costs one idle grid
	if ((rc = launch_extend_lds(ctx, d_pool, d_tasks, reg_ok && qmax <= 256 ? 64 : n, d_res, lists + 3 * (size_t)n, counts + 3,
	                            qmax)))
		return rc;
	if (tm) {
		BMH_HIP(ctx, hipEventRecord(ctx->ev_bin[4], ctx->stream));
		BMH_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
		ctx->ev_valid = ctx->ev_bin_valid = true;
	}
	return BMH_OK;
}

} // namespace bmh
