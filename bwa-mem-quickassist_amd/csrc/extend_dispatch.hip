// extend_dispatch.hip -- routes every extension task to the kernel built for its query length.
//
//   bin 0: qlen <= 32    extend_lane_kernel<32>   (64 tasks per wave, one lane per task)
//   bin 1: qlen <= 64    extend_lane_kernel<64>
//   bin 2: qlen <= 128   extend_lane_kernel<128>
//   bin 3: qlen <= 256   extend_reg_kernel<4>     (one task per wave, 4 columns per lane)
//   bin 4: longer or qlen == 0                     extend_lds_kernel
// BMH_EXT_MODE=reg | grp | lds in the environment selects the one-task-per-wave register kernels, the
// four-tasks-per-wave group kernels, or the LDS kernel for bins 0-2 instead (A/B runs, profiles/).
//
// The bins are built ON THE DEVICE (a counting sort by bin and expected row count), and every
// extension kernel reads its bin size from device memory, so a launch needs no
// host round trip and the *_device entry point stays asynchronous on the caller's stream.
#include "bmh_ctx.h"
#include "bmh_device.h"

namespace bmh {

// Counting sort of the tasks by (bin, expected row count): three tiny kernels, no host round trip.
// The lane-per-task kernels want neighbouring tasks to run for a similar number of rows.
constexpr int kSortKeys = 1024;

__device__ __forceinline__ int ext_bin_of(int qlen, int tlen, int mode)
{
	// mode 0: lane-per-task kernels for qlen <= 128; 1: LDS kernel only; 2: one task per wave; 3: four tasks per wave
	if (mode == 1 || qlen < 1) return 4;
	if (mode == 3 && qlen <= 256 && tlen > kGrpTcapHost) return 3; // the group kernels stage the target in LDS
	return qlen <= 32 ? 0 : qlen <= 64 ? 1 : qlen <= 128 ? 2 : qlen <= 256 ? 3 : 4;
}

__global__ __launch_bounds__(256) void sort_hist_kernel(const bmh_ext_task_t *__restrict__ tasks,
                                                        const uint32_t *__restrict__ order, long long n,
                                                        uint32_t *__restrict__ hist, uint32_t *__restrict__ binkey,
                                                        int mode)
{
	for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long long)gridDim.x * blockDim.x) {
		const uint32_t idx = order ? order[k] : (uint32_t)k;
		const int qlen = tasks[idx].qlen, tlen = tasks[idx].tlen;
		const int bin = ext_bin_of(qlen, tlen, mode);
		// rows run at most to tlen, and the band leaves the query after ~qlen+w <= 2*qlen rows (ksw.c:418)
		const int key = bin <= 2 ? min(min(tlen, 2 * qlen + 8), kSortKeys - 1) : 0;
		binkey[k] = (uint32_t)(bin * kSortKeys + key);
		atomicAdd(&hist[bin * kSortKeys + key], 1u);
	}
}

// exclusive scan of each bin's histogram (in place -> scatter offsets) and the bin sizes
__global__ __launch_bounds__(1024) void sort_scan_kernel(uint32_t *__restrict__ hist, uint32_t *__restrict__ counts)
{
	__shared__ uint32_t part[1024];
	const int t = threadIdx.x;
	for (int b = 0; b < kExtBins; ++b) {
		const uint32_t v = hist[b * kSortKeys + t];
		part[t] = v;
		__syncthreads();
		for (int d = 1; d < 1024; d <<= 1) { // Hillis-Steele inclusive scan, 1024 threads
			const uint32_t add = t >= d ? part[t - d] : 0;
			__syncthreads();
			part[t] += add;
			__syncthreads();
		}
		hist[b * kSortKeys + t] = part[t] - v;
		if (t == 1023) counts[b] = part[t];
		__syncthreads();
	}
}

__global__ __launch_bounds__(256) void sort_scatter_kernel(const uint32_t *__restrict__ order, long long n,
                                                           uint32_t *__restrict__ offs,
                                                           const uint32_t *__restrict__ binkey,
                                                           uint32_t *__restrict__ lists)
{
	for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long long)gridDim.x * blockDim.x) {
		const uint32_t idx = order ? order[k] : (uint32_t)k;
		const uint32_t bk = binkey[k];
		const uint32_t pos = atomicAdd(&offs[bk], 1u);
		lists[(size_t)(bk / kSortKeys) * (size_t)n + pos] = idx;
	}
}

int launch_extend(bmh_ctx *ctx, const uint8_t *d_pool, const bmh_ext_task_t *d_tasks, int64_t n,
                  bmh_ext_result_t *d_res, const uint32_t *d_order, int qmax)
{
	if (n <= 0) return BMH_OK;
	int rc;
	const int mode = ctx->force_kernel; // 0 lane-per-task, 1 lds, 2 reg (1 task/wave), 3 grp (4 tasks/wave)
	const size_t N = (size_t)n;
	const size_t hist_words = (size_t)kExtBins * kSortKeys;
	if ((rc = ensure(ctx, ctx->d_bins, (16 + hist_words + N + (size_t)kExtBins * N) * 4))) return rc;
	uint32_t *counts = (uint32_t *)ctx->d_bins.p;
	uint32_t *hist = counts + 16, *binkey = hist + hist_words, *lists = binkey + N;
	BMH_HIP(ctx, hipMemsetAsync(counts, 0, (16 + hist_words) * 4, ctx->stream));
	long long cg = (n + 255) / 256;
	if (cg > 4096) cg = 4096;
	hipLaunchKernelGGL(sort_hist_kernel, dim3((unsigned)cg), dim3(256), 0, ctx->stream, d_tasks, d_order, (long long)n, hist,
	                   binkey, mode);
	hipLaunchKernelGGL(sort_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, hist, counts);
	hipLaunchKernelGGL(sort_scatter_kernel, dim3((unsigned)cg), dim3(256), 0, ctx->stream, d_order, (long long)n, hist, binkey,
	                   lists);
	BMH_HIP(ctx, hipGetLastError());
	const bool tm = ctx->timing;
	if (tm) BMH_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
	for (int b = 0; b < kExtBins; ++b) {
		if (tm) BMH_HIP(ctx, hipEventRecord(ctx->ev_bin[b], ctx->stream));
		const uint32_t *lst = lists + (size_t)b * N, *cnt = counts + b;
		const int qlo = b == 0 ? 0 : 16 << b; // bins 0..3 hold qlen <= 32,64,128,256
		if (b < 4 && (mode == 1 || qmax <= qlo)) continue; // provably empty bin
		rc = BMH_OK;
		if (b <= 2) {
			if (mode == 0) rc = launch_extend_lane(ctx, 32 << b, d_pool, d_tasks, n, d_res, lst, cnt);
			else if (mode == 3) rc = launch_extend_grp(ctx, 2 << b, d_pool, d_tasks, n, d_res, lst, cnt);
			else rc = launch_extend_reg(ctx, b == 2 ? 2 : 1, d_pool, d_tasks, n, d_res, lst, cnt);
		} else if (b == 3) rc = launch_extend_reg(ctx, 4, d_pool, d_tasks, n, d_res, lst, cnt);
		else rc = launch_extend_lds(ctx, d_pool, d_tasks, mode != 1 && qmax <= 256 ? 4096 : n, d_res, lst, cnt, qmax);
		if (rc) return rc;
	}
	if (tm) {
		BMH_HIP(ctx, hipEventRecord(ctx->ev_bin[kExtBins], ctx->stream));
		BMH_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
		ctx->ev_valid = ctx->ev_bin_valid = true;
	}
	return BMH_OK;
}

} // namespace bmh
