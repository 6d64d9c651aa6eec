// sw_dispatch.hip -- routes local Smith-Waterman tasks (bmh_sw_batch, ksw_align2 semantics) to their kernels.
#include <algorithm>

#include "bmh_ctx.h"
#include "bmh_device.h"
#include "sw_common.h"

namespace bmh {

// largest padded query and target length of a device-resident batch (the *_device entry point has no host view)
__global__ void sw_caps_kernel(const bmh_sw_task_t *__restrict__ tasks, long long n, int *caps)
{
	int q = 0, t = 0;
	for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long long)gridDim.x * blockDim.x) {
		q = max(q, (int)tasks[k].qlen), t = max(t, (int)min(tasks[k].tlen, 0x7fffffffu));
	}
	q = wave_reduce_max(q), t = wave_reduce_max(t);
	if ((threadIdx.x & 63) == 0) atomicMax(&caps[0], q), atomicMax(&caps[1], t);
}

int launch_sw(bmh_ctx *ctx, const uint8_t *d_pool, const bmh_sw_task_t *d_tasks, int64_t n, bmh_sw_result_t *d_res,
              int qcap, int tcap)
{
	if (n <= 0) return BMH_OK;
	if (ctx->params.o_ins < 1) {
		ctx->last_error = "the Smith-Waterman kernels need o_ins >= 1 (with o_ins == 0 the reference's lazy-F loop is not a closed recurrence)";
		return BMH_E_RANGE;
	}
	if (ctx->dev.max_mat < 1) {
		ctx->last_error = "the Smith-Waterman kernels need a positive match score";
		return BMH_E_RANGE;
	}
	int rc;
	if (qcap < 0 || tcap < 0) { // one small reduction + read-back
		if ((rc = ensure(ctx, ctx->d_scratch, 256))) return rc;
		int *caps = (int *)ctx->d_scratch.p, h[2] = {0, 0};
		BMH_HIP(ctx, hipMemsetAsync(caps, 0, 8, ctx->stream));
		hipLaunchKernelGGL(sw_caps_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 1024)), dim3(256), 0,
		                   ctx->stream, d_tasks, (long long)n, caps);
		BMH_HIP(ctx, hipMemcpyAsync(h, caps, 8, hipMemcpyDeviceToHost, ctx->stream));
		BMH_HIP(ctx, hipStreamSynchronize(ctx->stream));
		qcap = h[0], tcap = h[1];
	}
	qcap = std::max(qcap, 1) + 16, tcap = std::max(tcap, 1);
	if (ctx->timing) BMH_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
	if ((rc = launch_sw_generic(ctx, d_pool, d_tasks, n, d_res, nullptr, nullptr, qcap, tcap))) return rc;
	if (ctx->timing) {
		BMH_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
		ctx->ev_valid = true;
	}
	return BMH_OK;
}

} // namespace bmh
