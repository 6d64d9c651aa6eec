// extend_dispatch.hip -- routes every extension task to the kernel built for its query length.
//
//   bin 0: qlen <= 32    extend_lane_kernel<32>   (64 tasks per wave, one lane per task)
//   bin 1: qlen <= 64    extend_lane_kernel<64>
//   bin 2: qlen <= 128   extend_lane_kernel<128>
//   bin 3: qlen <= 256   extend_lanex_kernel<2>   (32 tasks per wave, two lanes x 128 columns per task)
//   bin 4: qlen <= 512   extend_lanex_kernel<4>   (16 tasks per wave; only with BMH_EXT_MODE=lanex4: with the few
//                        such tasks a 150-300 bp run produces, one wave per task keeps more of the chip busy)
//   bin 5: longer or qlen == 0                     extend_lds_kernel  (one wave per task)
// BMH_EXT_MODE=reg | grp | lds in the environment selects the one-task-per-wave register kernels, the
// four-tasks-per-wave group kernels, or the LDS kernel for bins 0-2 instead (A/B runs, profiles/).
//
// The bins are built ON THE DEVICE (a counting sort by bin and expected row count), and every
// extension kernel reads its bin size from device memory, so a launch needs no
// host round trip and the *_device entry point stays asynchronous on the caller's stream.
#include <algorithm>

#include "bmh_ctx.h"
#include "bmh_device.h"

namespace bmh {

// Counting sort of the tasks by (bin, expected row count): three tiny kernels, no host round trip.
// The lane-per-task kernels want neighbouring tasks to run for a similar number of rows.
constexpr int kSortKeys = 2048;

__device__ __forceinline__ int ext_bin_of(int qlen, int tlen, int mode)
{
	// mode 0: lane-per-task kernels (qlen <= 256); 1: LDS kernel only; 2: one task per wave; 3: four tasks per wave;
	// 4: like 0 plus the four-lanes-per-task kernel for qlen <= 512
	if (mode == 1 || qlen < 1) return 5;
	if (mode == 3 && qlen <= 256 && tlen > kGrpTcapHost) return 3; // the group kernels stage the target in LDS
	return qlen <= 32 ? 0 : qlen <= 64 ? 1 : qlen <= 128 ? 2 : qlen <= 256 ? 3 : (qlen <= 512 && mode == 4) ? 4 : 5;
}

// sort key inside a bin: query-length bucket (major; lanes of a wave then share the unused leading columns,
// which the lane kernels skip), h0 bucket, and expected row count (minor; lanes of a wave then finish together).
// rows run at most to tlen, and the band leaves the query after ~qlen+w <= 2*qlen rows (ksw.c:418).
__device__ __forceinline__ int ext_sort_key(int bin, int qlen, int tlen, int h0)
{
	if (bin > 4) return 0;
	const int qlo = bin == 0 ? 1 : (16 << bin) + 1, qsh = bin < 2 ? 1 : bin; // 16 query-length buckets per bin
	const int rows = min(tlen, 2 * qlen + 8) >> (bin > 2 ? bin - 2 : 0);
	// h0 decides how wide the live interval is (cells stay non-zero within ~h0-o-e of the diagonal), so lanes
	// with a similar h0 need the same 8-column blocks
	return ((max(qlen - qlo, 0) >> qsh) * 8 + min(max(h0, 0) >> 4, 7)) * 16 + min(rows >> 4, 15);
}

constexpr int kSortBlocks = 512, kSortThreads = 256;
constexpr int kLanexMinTasks = 4096; // below this many 129-256 bp flanks one wave per task fills the chip better

// pass 1: per-block histogram in LDS over a contiguous chunk, flushed with one global atomic per used key
__global__ __launch_bounds__(kSortThreads) void sort_hist_kernel(const bmh_ext_task_t *__restrict__ tasks,
                                                                 const uint32_t *__restrict__ order, long long n,
                                                                 uint32_t *__restrict__ hist,
                                                                 uint16_t *__restrict__ binkey, int mode,
                                                                 const uint32_t *__restrict__ dn)
{
	__shared__ uint32_t lh[kSortBins * kSortKeys];
	for (int t = threadIdx.x; t < kSortBins * kSortKeys; t += kSortThreads) lh[t] = 0;
	__syncthreads();
	if (dn) n = min(n, (long long)*dn); // the list was built on the device (fused per-seed pipeline)
	const long long chunk = (n + gridDim.x - 1) / gridDim.x, lo = chunk * blockIdx.x, hi = min(lo + chunk, n);
	for (long long k = lo + threadIdx.x; k < hi; k += kSortThreads) {
		const uint32_t idx = order ? order[k] : (uint32_t)k;
		const int qlen = tasks[idx].qlen, tlen = tasks[idx].tlen;
		const int bin = ext_bin_of(qlen, tlen, mode);
		const int bk = bin * kSortKeys + ext_sort_key(bin, qlen, tlen, tasks[idx].h0);
		binkey[k] = (uint16_t)bk;
		atomicAdd(&lh[bk], 1u);
	}
	__syncthreads();
	for (int t = threadIdx.x; t < kSortBins * kSortKeys; t += kSortThreads)
		if (lh[t]) atomicAdd(&hist[t], lh[t]);
}

// pass 2: exclusive scan of each bin's histogram (in place -> cursors) and the bin sizes
__global__ __launch_bounds__(1024) void sort_scan_kernel(uint32_t *__restrict__ hist, uint32_t *__restrict__ counts)
{
	static_assert(kSortKeys == 2048, "two keys per thread");
	__shared__ uint32_t part[1024];
	const int t = threadIdx.x;
	for (int b = 0; b < kSortBins; ++b) {
		const uint32_t v0 = hist[b * kSortKeys + 2 * t], v1 = hist[b * kSortKeys + 2 * t + 1];
		part[t] = v0 + v1;
		__syncthreads();
		for (int d = 1; d < 1024; d <<= 1) { // Hillis-Steele inclusive scan over the pair sums
			const uint32_t add = t >= d ? part[t - d] : 0;
			__syncthreads();
			part[t] += add;
			__syncthreads();
		}
		const uint32_t excl = part[t] - (v0 + v1);
		hist[b * kSortKeys + 2 * t] = excl;
		hist[b * kSortKeys + 2 * t + 1] = excl + v0;
		if (t == 1023) counts[b] = part[t];
		__syncthreads();
	}
}

// pass 3: every block re-counts its chunk, reserves one range per used key from the global cursors and
// places its tasks inside those ranges with LDS atomics
__global__ __launch_bounds__(kSortThreads) void sort_scatter_kernel(const uint32_t *__restrict__ order, long long n,
                                                                    uint32_t *__restrict__ cursor,
                                                                    const uint16_t *__restrict__ binkey,
                                                                    uint32_t *__restrict__ lists, long long stride,
                                                                    const uint32_t *__restrict__ dn)
{
	__shared__ uint32_t lh[kSortBins * kSortKeys];
	for (int t = threadIdx.x; t < kSortBins * kSortKeys; t += kSortThreads) lh[t] = 0;
	__syncthreads();
	if (dn) n = min(n, (long long)*dn);
	const long long chunk = (n + gridDim.x - 1) / gridDim.x, lo = chunk * blockIdx.x, hi = min(lo + chunk, n);
	for (long long k = lo + threadIdx.x; k < hi; k += kSortThreads) atomicAdd(&lh[binkey[k]], 1u);
	__syncthreads();
	for (int t = threadIdx.x; t < kSortBins * kSortKeys; t += kSortThreads)
		if (lh[t]) lh[t] = atomicAdd(&cursor[t], lh[t]); // count -> start of this block's range
	__syncthreads();
	for (long long k = lo + threadIdx.x; k < hi; k += kSortThreads) {
		const uint32_t bk = binkey[k];
		const uint32_t pos = atomicAdd(&lh[bk], 1u);
		lists[(size_t)(bk / kSortKeys) * (size_t)stride + pos] = order ? order[k] : (uint32_t)k;
	}
}

// shared by the extension and the global dispatchers: workspace layout + the scan/scatter passes
static_assert(kSortKeys == kSortKeysHost, "keep bmh_ctx.h in sync");
int sort_tasks_begin(bmh_ctx *ctx, int64_t n, uint32_t **counts, uint32_t **lists)
{
	const size_t N = (size_t)n, hist_words = (size_t)kSortBins * kSortKeys;
	int rc = ensure(ctx, ctx->d_bins, (16 + hist_words + (N + 1) / 2 + 1 + (size_t)kSortBins * N) * 4);
	if (rc) return rc;
	*counts = (uint32_t *)ctx->d_bins.p;
	*lists = *counts + 16 + hist_words + (N + 1) / 2 + 1;
	BMH_HIP(ctx, hipMemsetAsync(*counts, 0, (16 + hist_words) * 4, ctx->stream));
	return BMH_OK;
}

int sort_tasks_finish(bmh_ctx *ctx, int64_t n, const uint32_t *d_order, unsigned blocks, const uint32_t *d_n)
{
	const size_t N = (size_t)n, hist_words = (size_t)kSortBins * kSortKeys;
	uint32_t *counts = (uint32_t *)ctx->d_bins.p, *hist = counts + 16;
	uint16_t *binkey = (uint16_t *)(hist + hist_words);
	uint32_t *lists = hist + hist_words + (N + 1) / 2 + 1;
	hipLaunchKernelGGL(sort_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, hist, counts);
	hipLaunchKernelGGL(sort_scatter_kernel, dim3(blocks), dim3(kSortThreads), 0, ctx->stream, d_order, (long long)n, hist, binkey,
	                   lists, (long long)n, d_n);
	BMH_HIP(ctx, hipGetLastError());
	return BMH_OK;
}

// ---- bin-size hints (bmh_ctx::BinHint): the counts of the previous launch of the same kind, if they have arrived
static void hint_poll(bmh_ctx *ctx, int kind)
{
	bmh_ctx::BinHint &h = ctx->hint[kind];
	if (h.pending && hipEventQuery(h.ev) == hipSuccess) {
		for (int b = 0; b < 8; ++b) h.cnt[b] = h.h[b];
		h.pending = false, h.valid = true;
	}
}

static int hint_post(bmh_ctx *ctx, int kind, const uint32_t *d_counts)
{
	bmh_ctx::BinHint &h = ctx->hint[kind];
	if (h.pending) return BMH_OK; // the previous copy has not landed yet; do not overwrite what it is writing
	if (!h.h) {
		if (hipHostMalloc((void **)&h.h, 64, hipHostMallocDefault) != hipSuccess || hipEventCreateWithFlags(&h.ev, hipEventDisableTiming) != hipSuccess) {
			(void)hipGetLastError();
			return BMH_OK; // no hints then
		}
	}
	BMH_HIP(ctx, hipMemcpyAsync(h.h, d_counts, 32, hipMemcpyDeviceToHost, ctx->stream));
	BMH_HIP(ctx, hipEventRecord(h.ev, ctx->stream));
	h.pending = true;
	return BMH_OK;
}

// a list whose length lives on the device and that the last launch of this kind found (almost) empty -- the band-doubling
// retries of the fused per-seed pipeline, for reads that need none: no sort, no bins, ONE launch of the any-length kernel
// (one wave per task) over the list as it stands.  Correct for any count; a count that has grown shows in the next hint.
constexpr uint32_t kTinyList = 2048;
static int launch_extend_tiny(bmh_ctx *ctx, const uint8_t *d_pool, const bmh_ext_task_t *d_tasks, int64_t n, bmh_ext_result_t *d_res,
                              const uint32_t *d_order, int qmax, const uint32_t *d_n, int kind)
{
	bmh_ctx::BinHint &h = ctx->hint[kind];
	int rc = launch_extend_lds(ctx, d_pool, d_tasks, n, d_res, d_order, d_n, qmax, 2 * kTinyList);
	if (rc) return rc;
	if (!h.pending && h.h) { // the list length stands in for the bin counts: it is all the next decision needs
		for (int b = 0; b < 8; ++b) h.h[b] = 0;
		BMH_HIP(ctx, hipMemcpyAsync(h.h + 5, d_n, 4, hipMemcpyDeviceToHost, ctx->stream));
		BMH_HIP(ctx, hipEventRecord(h.ev, ctx->stream));
		h.pending = true;
	}
	return BMH_OK;
}

constexpr int64_t kForkMinTasks = 131072; // batches below this run their bins on one stream (see launch_extend)

int launch_extend(bmh_ctx *ctx, const uint8_t *d_pool, const bmh_ext_task_t *d_tasks, int64_t n,
                  bmh_ext_result_t *d_res, const uint32_t *d_order, int qmax, const uint32_t *d_n, int kind)
{
	if (n <= 0) return BMH_OK;
	int rc;
	if (kind < 0 || kind >= bmh_ctx::kHintKinds) kind = 0;
	hint_poll(ctx, kind);
	const bmh_ctx::BinHint &hint = ctx->hint[kind];
	if (d_n && hint.valid && !ctx->ext_mode_forced) {
		uint32_t tot = 0;
		for (int b = 0; b < kExtBins; ++b) tot += hint.cnt[b];
		if (tot <= kTinyList) return launch_extend_tiny(ctx, d_pool, d_tasks, n, d_res, d_order, qmax, d_n, kind);
	}
	// expected size of bin b: the previous launch's count with a margin -- or, with no hint, the whole batch
	int64_t est[kExtBins], est_total = 0;
	for (int b = 0; b < kExtBins; ++b) {
		est[b] = hint.valid ? std::min<int64_t>(n, (int64_t)hint.cnt[b] + (hint.cnt[b] >> 4) + 64) : n;
		est_total += hint.valid ? hint.cnt[b] : 0;
	}
	if (!hint.valid) est_total = n;
	// 0 lane-per-task, 1 lds, 2 reg (1 task/wave), 3 grp (4 tasks/wave).  The lane-per-task kernels are built for
	// throughput: a wave walks ~100 rows x 128 columns for its 64 tasks, about half a millisecond however small the batch.
	// A driver round of a few thousand tasks (bmh_chain2aln_batch: 8 192 reads per call) cannot fill the chip anyway and
	// wants latency: one task per wave finishes in tens of microseconds (8 153 tasks: 0.76 -> 0.26 ms per call, 32 647:
	// 0.84 -> 0.51 ms; level at 65 k).  With a device-side count (d_n) the decision uses the hinted size.
	const int64_t n_eff = d_n ? std::min<int64_t>(n, est_total + (est_total >> 2) + 64) : n;
	const int mode = ctx->ext_mode_forced ? ctx->force_kernel : n_eff <= ctx->small_batch ? 2 : 0;
	const size_t N = (size_t)n;
	uint32_t *counts, *lists;
	if ((rc = sort_tasks_begin(ctx, n, &counts, &lists))) return rc;
	uint32_t *hist = counts + 16;
	uint16_t *binkey = (uint16_t *)(hist + (size_t)kSortBins * kSortKeys);
	long long cg = (n_eff + 1023) / 1024;
	if (cg > kSortBlocks) cg = kSortBlocks;
	if (cg < 1) cg = 1;
	hipLaunchKernelGGL(sort_hist_kernel, dim3((unsigned)cg), dim3(kSortThreads), 0, ctx->stream, d_tasks, d_order, (long long)n,
	                   hist, binkey, mode, d_n);
	if ((rc = sort_tasks_finish(ctx, n, d_order, (unsigned)cg, d_n))) return rc;
	if ((rc = hint_post(ctx, kind, counts))) return rc;
	const bool tm = ctx->timing;
	if (tm) BMH_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
	// Bins 0-2 (lane-per-task kernels, almost all tasks) run on the caller's stream; the few long flanks of bins 3-5
	// (few waves, each running for many rows) run BESIDE them on a second stream instead of as a serial tail.
	hipStream_t main_s = ctx->stream;
	// A small batch (the preload shim's: 33 k seeds) runs its bins one after the other on the caller's stream: the side
	// stream buys nothing there, costs 2.7 ms to create per context and one more stream to share the hardware queues with.
	const bool fork = n >= kForkMinTasks || ctx->ext_sched >= 0;
	if (fork && !ctx->aux_stream) BMH_HIP(ctx, hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking));
	hipStream_t side_s = fork ? ctx->aux_stream : main_s;
	if (fork) {
		BMH_HIP(ctx, hipEventRecord(ctx->ev_fork, main_s));
		BMH_HIP(ctx, hipStreamWaitEvent(side_s, ctx->ev_fork, 0));
	}
	static const int orders[5][kExtBins] = {{3, 4, 5, 0, 1, 2}, {3, 4, 5, 2, 1, 0}, {3, 4, 5, 2, 1, 0}, {3, 4, 5, 2, 1, 0}, {2, 1, 0, 3, 4, 5}};
	// 0: short bins first; 1: long bins first; 2: bins 0-1 behind the long flanks on the second stream, beside bin 2;
	// 3: bins 0-1 on a third stream of their own (A/B knob BMH_EXT_SCHED)
	// Default: long bins first.  When no query is longer than 160 (150 bp reads: bins 3-5 hold a handful of tasks) the two
	// short-query bins go behind them on the second stream and run beside the 128-column bin, whose two waves per SIMD leave
	// issue slots free: 4.61 -> 4.41 ms per 1 M reads.  With many long flanks (100-300 bp reads) that delays bins 3-4, which
	// are the critical path there (17.8 against 16.8 ms), so they keep the second stream to themselves.
	const int sched = ctx->ext_sched >= 0 ? ctx->ext_sched : !fork ? 1 : qmax <= 160 ? 2 : 1;
	const int *order = orders[sched];
	if (sched == 3) { // (an A/B knob: its stream is made when first asked for -- a stream costs 2.7 ms to create)
		if (!ctx->aux2_stream) BMH_HIP(ctx, hipStreamCreateWithFlags(&ctx->aux2_stream, hipStreamNonBlocking));
		BMH_HIP(ctx, hipStreamWaitEvent(ctx->aux2_stream, ctx->ev_fork, 0));
	}
	for (int k = 0; k < kExtBins; ++k) {
		const int b = order[k];
		ctx->stream = b >= 3 ? side_s : b < 2 && (sched == 2 || sched == 4) ? side_s : b < 2 && sched == 3 ? ctx->aux2_stream : main_s; // the launchers enqueue on ctx->stream
		if (tm) {
			rc = (int)hipEventRecord(ctx->ev_bin[b], ctx->stream);
			if (rc) { ctx->stream = main_s; return set_hip_error(ctx, (hipError_t)rc, "hipEventRecord"); }
		}
		const uint32_t *lst = lists + (size_t)b * N, *cnt = counts + b;
		const int qlo = b == 0 ? 0 : 16 << b; // bins 0..4 hold qlen <= 32,64,128,256,512
		// grid of the bin's kernel: sized for the expected count (every kernel strides over its bin, so any grid is
		// correct); a bin the hint calls empty still gets a few hundred waves in case the batch differs from the last one
		const int64_t eb = std::max<int64_t>(est[b], 16384);
		rc = BMH_OK;
		if (b < 5 && (mode == 1 || (qmax <= qlo && !(mode == 3 && b == 3)))) { // (mode 3 routes long targets of short queries to bin 3)
			// provably empty bin
		} else if (b <= 2) {
			if ((mode == 0 || mode == 4) && b == 2 && ctx->ext_split96) {
				// the 65-128 bin is sorted by query length first (16 buckets of 4 columns): its tasks of up to 96 columns are the head of
				// the list, and where it ends stands in the sort's cursor array -- the key range of buckets 0-7 closes at key 1023.  They go
				// to a 96-column instantiation, whose 120 state registers leave room for 3 waves per SIMD (the 128-column one: 2).
				const uint32_t *n96 = hist + (size_t)2 * kSortKeys + 1023;
				rc = launch_extend_lane(ctx, 96, d_pool, d_tasks, hint.valid ? est[b] : n, d_res, lst, n96, true); // (no pick-up launch: see `rem` in the kernel)
				if (!rc) rc = launch_extend_lane(ctx, 128, d_pool, d_tasks, hint.valid ? est[b] : n, d_res, lst, cnt, !hint.valid, n96);
			} else if (mode == 0 || mode == 4) rc = launch_extend_lane(ctx, 32 << b, d_pool, d_tasks, hint.valid ? est[b] : n, d_res, lst, cnt, !hint.valid);
			else if (mode == 3) rc = launch_extend_grp(ctx, 2 << b, d_pool, d_tasks, n, d_res, lst, cnt);
			else rc = launch_extend_reg(ctx, b == 2 ? 2 : 1, d_pool, d_tasks, n, d_res, lst, cnt, 0, est[b]);
		} else if (b == 3) {
			if (mode == 0 || mode == 4) {
				// the bin size (known on the device only) decides which kernel works: many 129-256 bp flanks -> two lanes per
				// task, a handful -> one wave per task.  With a hint exactly one of them is launched (either handles any count);
				// without one both are, and each looks at the count
				if (hint.valid) {
					if (hint.cnt[3] >= (uint32_t)kLanexMinTasks) rc = launch_extend_lanex(ctx, 2, d_pool, d_tasks, eb, d_res, lst, cnt, 0);
					else rc = launch_extend_reg(ctx, 4, d_pool, d_tasks, n, d_res, lst, cnt, 0, std::max<int64_t>(est[3], 1024));
				} else {
					rc = launch_extend_lanex(ctx, 2, d_pool, d_tasks, n, d_res, lst, cnt, kLanexMinTasks);
					if (!rc) rc = launch_extend_reg(ctx, 4, d_pool, d_tasks, n < kLanexMinTasks ? n : kLanexMinTasks, d_res, lst, cnt, kLanexMinTasks);
				}
			} else rc = launch_extend_reg(ctx, 4, d_pool, d_tasks, n, d_res, lst, cnt, 0, est[b]);
		} else if (b == 4 && mode == 4) rc = launch_extend_lanex(ctx, 4, d_pool, d_tasks, eb, d_res, lst, cnt, 0);
		else if (b == 5) rc = launch_extend_lds(ctx, d_pool, d_tasks, n, d_res, lst, cnt, qmax, hint.valid ? std::max<int64_t>(est[5], 512) : (mode != 1 && qmax <= 256 ? 4096 : 0));
		if (!rc && tm) rc = (int)hipEventRecord(ctx->ev_bin_end[b], ctx->stream) ? BMH_E_HIP : BMH_OK;
		if (rc) { ctx->stream = main_s; return rc; }
	}
	ctx->stream = main_s;
	if (fork) {
		BMH_HIP(ctx, hipEventRecord(ctx->ev_join, side_s));
		BMH_HIP(ctx, hipStreamWaitEvent(main_s, ctx->ev_join, 0));
	}
	if (sched == 3) {
		BMH_HIP(ctx, hipEventRecord(ctx->ev_join2, ctx->aux2_stream));
		BMH_HIP(ctx, hipStreamWaitEvent(main_s, ctx->ev_join2, 0));
	}
	if (tm) {
		BMH_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
		ctx->ev_valid = ctx->ev_bin_valid = true;
		// timing mode is a measurement mode: wait here and add this launch's per-bin durations to the running sums, so that a
		// caller of the fused per-seed pipeline (four dispatcher launches per call) can attribute its time to kernels
		BMH_HIP(ctx, hipEventSynchronize(ctx->ev1));
		for (int b = 0; b < kExtBins; ++b) {
			float ms = 0.f;
			if (hipEventElapsedTime(&ms, ctx->ev_bin[b], ctx->ev_bin_end[b]) == hipSuccess && ms > 0.f) ctx->ext_bin_ms_sum[b] += ms;
			else (void)hipGetLastError();
		}
		++ctx->ext_bin_launches;
	}
	return BMH_OK;
}

} // namespace bmh
