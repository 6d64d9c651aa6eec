// sw_dispatch.hip -- routes local Smith-Waterman tasks (bmh_sw_batch, ksw_align2 semantics) to their kernels.
#include <algorithm>

#include "bmh_ctx.h"
#include "bmh_device.h"
#include "sw_common.h"

namespace bmh {

// largest padded query and target length of a device-resident batch (the *_device entry point has no host view)
__global__ void sw_caps_kernel(const bmh_sw_task_t *__restrict__ tasks, long long n, int *caps)
{
	int q = 0, t = 0, qi = 0; // qi = 65535 - shortest query
	for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long long)gridDim.x * blockDim.x) {
		q = max(q, (int)tasks[k].qlen), t = max(t, (int)min(tasks[k].tlen, 0x7fffffffu)), qi = max(qi, 65535 - (int)tasks[k].qlen);
	}
	q = wave_reduce_max(q), t = wave_reduce_max(t), qi = wave_reduce_max(qi);
	if ((threadIdx.x & 63) == 0) atomicMax(&caps[0], q), atomicMax(&caps[1], t), atomicMax(&caps[2], qi);
}

// ---- batches of up to 32 768 tasks: sw_wave_kernel (one wave per task, sw_wave.hip) takes every task the register kernels
//      would; the routing below then only feeds sw_generic_kernel (launch_sw)
// ---- device-side routing: a counting sort by (kernel, query length, target length), like the extension dispatcher
//   bin 0: byte mode, cannot overflow, padded query <= 80 columns   sw_lane_kernel<40>
//   bin 1: ... <= 160 columns                                        sw_lane_kernel<80>
//   bin 2: everything else                                           sw_generic_kernel (both passes)
//   bin 3, 4: like 0, 1 for queries holding an N                     sw_lane_kernel<.., CORR>
//   bin 5: nothing to do (second pass not wanted)
//   bin 6: byte mode, cannot overflow, padded query <= 256 columns  sw_lane_kernel<128, CORR>
//   bin 7: word mode, scores below 512, padded query <= 256 columns sw_lane_kernel<128, CORR, WORD> (250 bp reads)
__device__ __forceinline__ bool sw_lane_bin(int bin) { return bin < 2 || bin >= 6; }
__device__ __forceinline__ int sw_bin_of(const DevParams &P, int qlen, uint32_t xtra, int mode)
{
	if (mode == 1 || qlen < 1) return 2;
	if (!(xtra & BMH_SW_XBYTE)) { // ksw_i16: 8 segments; the two register halves hold roundup8(4*slen) columns each
		const int half = (((qlen + 7) >> 3) * 4 + 7) & ~7;
		return half <= 128 && qlen * P.max_mat + P.sw_shift < 512 ? 7 : 2;
	}
	if (qlen * P.max_mat + P.sw_shift >= 255) return 2;
	const int Qp = ((qlen + 15) >> 4) * 16;
	return Qp <= 80 ? 0 : Qp <= 160 ? 1 : Qp <= 256 ? 6 : 2;
}

constexpr int kSwSortThreads = 256;
__global__ __launch_bounds__(kSwSortThreads) void sw_hist_kernel(const bmh_sw_task_t *__restrict__ tasks, long long n,
                                                                 const bmh_sw_result_t *__restrict__ res, DevParams P,
                                                                 uint32_t *__restrict__ hist,
                                                                 uint16_t *__restrict__ binkey, int mode, int pass2,
                                                                 const uint8_t *__restrict__ pool, int qfine, int wave_cols)
{
	__shared__ uint32_t lh[kSortBins * kSortKeysHost];
	for (int t = threadIdx.x; t < kSortBins * kSortKeysHost; t += kSwSortThreads) lh[t] = 0;
	__syncthreads();
	const long long chunk = (n + gridDim.x - 1) / gridDim.x, lo = chunk * blockIdx.x, hi = min(lo + chunk, n);
	for (long long k = lo + threadIdx.x; k < hi; k += kSwSortThreads) {
		const uint32_t xtra = tasks[k].xtra;
		int qlen = tasks[k].qlen, rows = (int)min(tasks[k].tlen, 0xffffu);
		int bin = sw_bin_of(P, qlen, xtra, mode);
		bool has_n = false;
		const bool done = wave_cols > 0 && sw_wave_takes(P, qlen, xtra, wave_cols); // sw_wave_kernel has been through the batch
		if (!pass2 && bin < 2 && !done) { // an N anywhere in the query sends the task to the correcting instantiation
			const uint64_t q0 = tasks[k].q_off;
			const bool rev = tasks[k].flags & BMH_F_QREV;
			for (int x = 0; x < qlen; ++x) has_n |= (rev ? pool[q0 - x] : pool[q0 + x]) > 3;
		}
		if (pass2) { // ksw.c:354: only where start positions are wanted and the first pass reached the threshold
			const int score = res[k].score;
			const bool second = sw_lane_bin(bin) && (xtra & BMH_SW_XSTART) && !((xtra & BMH_SW_XSUBO) && score < (int)(xtra & 0xffff)) &&
			                    res[k].qe >= 0;
			qlen = res[k].qe + 1;
			bin = second ? sw_bin_of(P, qlen, xtra, mode) : 5;
			has_n = res[k].rsv != 0;
			rows = min(rows, 2 * qlen + 16); // the reversed pass stops once the score is reached
		}
		// lanes of a wave must share ceil(qlen/16) and should share qlen (uniform padding) and the row count
		// qfine >= 0: all queries of the batch lie in [qfine, qfine+64) (the usual case: one read length) -- six bits tell them
		// apart and five are left for the row count, so that the lanes of a wave finish within 32 rows of one another
		const int key = !sw_lane_bin(bin) ? 0 : (!pass2 && qfine >= 0) ? ((qlen - qfine) << 5) | min(rows >> 5, 31) : (min(qlen, 255) << 3) | min(rows >> 7, 7);
		if (bin < 2 && has_n) bin += 3;
		if (done) bin = 5;
		const int bk = bin * kSortKeysHost + (done ? 0 : key);
		binkey[k] = (uint16_t)bk;
		atomicAdd(&lh[bk], 1u);
	}
	__syncthreads();
	for (int t = threadIdx.x; t < kSortBins * kSortKeysHost; t += kSwSortThreads)
		if (lh[t]) atomicAdd(&hist[t], lh[t]);
}

int launch_sw(bmh_ctx *ctx, const uint8_t *d_pool, const bmh_sw_task_t *d_tasks, int64_t n, bmh_sw_result_t *d_res,
              int qcap, int tcap, int qmin)
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
		int *caps = (int *)ctx->d_scratch.p, h[3] = {0, 0, 0};
		BMH_HIP(ctx, hipMemsetAsync(caps, 0, 12, ctx->stream));
		hipLaunchKernelGGL(sw_caps_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 1024)), dim3(256), 0,
		                   ctx->stream, d_tasks, (long long)n, caps);
		BMH_HIP(ctx, hipMemcpyAsync(h, caps, 12, hipMemcpyDeviceToHost, ctx->stream));
		BMH_HIP(ctx, stream_wait(ctx, ctx->stream));
		qcap = h[0], tcap = h[1], qmin = 65535 - h[2];
	}
	const int qfine = (qmin >= 1 && qcap - qmin < 64) ? qmin : -1; // see sw_hist_kernel
	// a batch that cannot fill the chip with one lane per task goes to one wave per task (sw_wave.hip): the register
	// kernels' launches take as long as one lane's whole matrix, milliseconds however few the tasks.  What that kernel does
	// not take (arithmetic that can saturate) still goes through the routing below, to sw_generic_kernel.
	const bool wave = ctx->sw_mode == 0 && ctx->sw_wave && sw_wave_fits(n, qcap, tcap);
	const int wave_cols = wave ? ((std::max(qcap, 1) + 15) / 16 * 16 <= 192 ? 192 : 320) : 0;
	qcap = std::max(qcap, 1) + 16, tcap = std::max(tcap, 1);
	if (ctx->params.o_del > 255 || ctx->params.e_del > 255 || ctx->params.o_ins > 255 || ctx->params.e_ins > 255) {
		ctx->last_error = "the Smith-Waterman kernels need gap penalties below 256";
		return BMH_E_RANGE;
	}
	const int mode = ctx->sw_mode;
	const size_t N = (size_t)n;
	uint32_t *counts, *lists;
	long long cg = std::min<long long>((n + 1023) / 1024, 512);
	// per-wave slab of row maxima for the second-best score (ksw.c:181-189): [block][row][lane] u16
	const int grid = (int)std::min<long long>((n + 63) / 64, 2048 * ctx->grid_mult); // resident waves: 256 CUs x 4 SIMDs x 2
	if (!wave && (rc = ensure(ctx, ctx->d_swrm, (size_t)grid * ((size_t)tcap * 128 + 2048)))) return rc; // + kSwTableBytes per wave
	uint16_t *d_rm = (uint16_t *)ctx->d_swrm.p;
	if (ctx->timing) BMH_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
	if (wave && (rc = launch_sw_wave(ctx, d_pool, d_tasks, n, d_res, wave_cols, tcap))) return rc;
	for (int pass2 = 0; pass2 < (wave ? 1 : 2); ++pass2) { // (sw_generic_kernel runs its second pass itself)
		if ((rc = sort_tasks_begin(ctx, n, &counts, &lists))) return rc;
		uint32_t *hist = counts + 16;
		uint16_t *binkey = (uint16_t *)(hist + (size_t)kSortBins * kSortKeysHost);
		hipLaunchKernelGGL(sw_hist_kernel, dim3((unsigned)cg), dim3(kSwSortThreads), 0, ctx->stream, d_tasks, (long long)n,
		                   d_res, ctx->dev, hist, binkey, mode, pass2, d_pool, qfine, wave_cols);
		if ((rc = sort_tasks_finish(ctx, n, nullptr, (unsigned)cg))) return rc;
		if (wave) { // every bin but sw_generic_kernel's is empty
			if ((rc = launch_sw_generic(ctx, d_pool, d_tasks, n, d_res, lists + 2 * N, counts + 2, qcap, tcap))) return rc;
			break;
		}
		// (counts[b] = size of bin b, counts[8 + b] = its chunk cursor; longest columns first)
		if ((rc = launch_sw_lane(ctx, 128, true, true, d_pool, d_tasks, n, d_res, lists + 7 * N, counts + 7, d_rm, tcap, grid, pass2, counts + 15))) return rc;
		if (qcap > 160 + 16 && (rc = launch_sw_lane(ctx, 128, true, false, d_pool, d_tasks, n, d_res, lists + 6 * N, counts + 6, d_rm, tcap, grid, pass2, counts + 14))) return rc;
		if ((rc = launch_sw_lane(ctx, 80, false, false, d_pool, d_tasks, n, d_res, lists + N, counts + 1, d_rm, tcap, grid, pass2, counts + 9))) return rc;
		if ((rc = launch_sw_lane(ctx, 80, true, false, d_pool, d_tasks, n, d_res, lists + 4 * N, counts + 4, d_rm, tcap, grid, pass2, counts + 12))) return rc;
		if ((rc = launch_sw_lane(ctx, 40, false, false, d_pool, d_tasks, n, d_res, lists, counts, d_rm, tcap, grid, pass2, counts + 8))) return rc;
		if ((rc = launch_sw_lane(ctx, 40, true, false, d_pool, d_tasks, n, d_res, lists + 3 * N, counts + 3, d_rm, tcap, grid, pass2, counts + 11))) return rc;
		if (!pass2) {
			// what the register kernels do not take (more than 256 columns: reads of 260-300 bp) but one wave per task does
			// (up to 320 columns, arithmetic that cannot saturate) goes to sw_wave_kernel over the bin's list; the slab kernel
			// serves the rest (it used to serve all of it: 37 ms per launch for a handful of 300 bp mates, however few)
			const int wc = ctx->sw_mode == 0 && ctx->sw_wave && tcap <= 16384 && qcap - 16 > 256 ? 320 : 0;
			if (wc && (rc = launch_sw_wave(ctx, d_pool, d_tasks, n, d_res, wc, tcap, lists + 2 * N, counts + 2))) return rc;
			if ((rc = launch_sw_generic(ctx, d_pool, d_tasks, n, d_res, lists + 2 * N, counts + 2, qcap, tcap, wc))) return rc;
		}
	}
	if (ctx->timing) {
		BMH_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
		ctx->ev_valid = true;
	}
	return BMH_OK;
}

} // namespace bmh
