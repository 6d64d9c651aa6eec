// global_kernel.hip -- banded global (NW) affine alignment with traceback to a BAM CIGAR.
//
// Replaces ksw_global2 (reference bwa-0.7.8/ksw.c:501-584; spec SURVEY.md A.2).
// One wave64 per task, row-synchronous like the extension kernel:
//   * fixed band [max(0,i-w), min(qlen,i+w+1)) per row (ksw.c:528-529), 64 columns per chunk;
//   * H (shifted) and E as int32 in LDS (scores go far negative, -0x40000000 marks "outside"),
//     query profile 5 signed bytes per column in LDS;
//   * F(i,j+1)=max(F(i,j)-e_ins, M(i,j)-o_ins-e_ins) opens from the DIAGONAL score M
//     (ksw.c:538-541,557-560), so it is an exact max-plus prefix scan over lanes (6 DPP steps);
//   * one direction byte per cell, same encoding as the reference (ksw.c:547-561), written
//     row-contiguously either to LDS (ZLDS) or to a per-block HBM scratch slab;
//   * the traceback (ksw.c:566-581) is wave-parallel: the 64 lanes look 64 cells ahead along
//     the current direction (diagonal / column / row), a ballot finds where the run ends, so
//     one iteration emits a whole CIGAR run instead of one cell.
//   * CIGAR words are produced last-op-first and stored from the back of the task's slot
//     range, which leaves them in forward order; they are then moved to the front.
#include "bmh_ctx.h"
#include "bmh_device.h"

namespace bmh {

constexpr int kNegInf = -0x40000000; // MINUS_INF, ksw.c:487

struct CigarSink { // wave-uniform run-length CIGAR builder writing backwards from slot cap-1
	uint32_t *base;
	int cap, nw, last_op, last_len;
	__device__ __forceinline__ void flush(int lane)
	{
		if (last_len > 0) {
			if (nw < cap && lane == 0) base[cap - 1 - nw] = (uint32_t)last_len << 4 | (uint32_t)last_op;
			++nw;
		}
	}
	__device__ __forceinline__ void push(int op, int len, int lane)
	{
		if (len <= 0) return;
		if (last_len > 0 && op == last_op) last_len += len; // ksw.c:497
		else {
			flush(lane);
			last_op = op, last_len = len;
		}
	}
};

template <bool ZLDS>
__global__ __launch_bounds__(64) void global_kernel(const uint8_t *__restrict__ pool,
                                                    const bmh_glb_task_t *__restrict__ tasks,
                                                    const uint32_t *__restrict__ order,
                                                    const uint32_t *__restrict__ count, long long n,
                                                    bmh_glb_result_t *__restrict__ out,
                                                    uint32_t *__restrict__ cigar_pool, DevParams P, int qcap,
                                                    long long zcap, uint8_t *__restrict__ zscratch,
                                                    int *__restrict__ err_flag)
{
	extern __shared__ __align__(16) unsigned char smem[];
	int *H = (int *)smem;                   // [qcap+2] shifted H = eh[j].h
	int *E = H + (qcap + 2);                // [qcap+2]
	uint2 *PR = (uint2 *)(E + (qcap + 2));  // [qcap]  (8-byte aligned: 2*(qcap+2) ints precede)
	int8_t *smat = (int8_t *)(PR + qcap);   // [32]
	uint8_t *zl = (uint8_t *)(smat + 32);   // [zcap] when ZLDS
	uint8_t *z = ZLDS ? zl : zscratch + (size_t)blockIdx.x * (size_t)zcap;
	const int lane = threadIdx.x;
	const int oe_del = P.o_del + P.e_del, oe_ins = P.o_ins + P.e_ins;
	const int e_del = P.e_del, e_ins = P.e_ins;

	if (lane < 25) smat[lane] = (int8_t)mat_at(P, lane);

	if (count) n = *count; // bin size produced on the device by the dispatcher
	for (long long slot = blockIdx.x; slot < n; slot += gridDim.x) {
		const uint32_t idx = order ? order[slot] : (uint32_t)slot;
		const uint4 *tp = (const uint4 *)(tasks + idx);
		const uint4 ta = tp[0], tb = tp[1];
		const uint64_t q_off = (uint64_t)(uint32_t)uni(ta.y) << 32 | (uint32_t)uni(ta.x);
		const uint64_t t_off = (uint64_t)(uint32_t)uni(ta.w) << 32 | (uint32_t)uni(ta.z);
		const int qlen = uni(tb.x & 0xffff), tlen = uni(tb.x >> 16);
		const int w = uni((int)tb.y);
		const uint32_t cigar_off = (uint32_t)uni(tb.z);
		const int cigar_cap = uni(tb.w);
		const int n_col = min(qlen, 2 * w + 1); // ksw.c:509
		const bool want = cigar_cap > 0;

		if (qlen > qcap || w < 0 || (want && (long long)n_col * tlen > zcap)) {
			if (lane == 0) {
				out[idx].score = INT32_MIN, out[idx].n_cigar = 0;
				atomicExch(err_flag, BMH_E_RANGE);
			}
			continue;
		}

		// first row, ksw.c:519-522, and profile, ksw.c:514-517
		for (int j = lane; j <= qlen; j += 64) {
			H[j] = j == 0 ? 0 : (j <= w ? -(P.o_ins + e_ins * j) : kNegInf);
			E[j] = kNegInf;
			if (j < qlen) {
				const int qb = pool[q_off + (uint64_t)j];
				uint32_t lo = 0;
				for (int k = 0; k < 4; ++k) lo |= (uint32_t)(uint8_t)smat[k * 5 + qb] << (8 * k);
				PR[j] = make_uint2(lo, (uint32_t)(uint8_t)smat[20 + qb]);
			}
		}

		uint32_t tv = 0;
		for (int i = 0; i < tlen; ++i) { // ksw.c:524-564
			if ((i & 255) == 0) {
				tv = 0;
				for (int k = 0; k < 4; ++k) {
					const int r = i + lane * 4 + k;
					if (r < tlen) tv |= (uint32_t)pool[t_off + (uint64_t)r] << (8 * k);
				}
			}
			const int tw = __builtin_amdgcn_readlane((int)tv, (i >> 2) & 63);
			const int t = (tw >> ((i & 3) * 8)) & 0xff;
			const int beg = i > w ? i - w : 0;
			const int end = i + w + 1 < qlen ? i + w + 1 : qlen;
			int carry_h = beg == 0 ? -(P.o_del + e_del * (i + 1)) : kNegInf; // ksw.c:530
			int fin = kNegInf;                                                // F(i, cb)
			uint8_t *zi = z + (size_t)i * n_col;
			for (int cb = beg; cb < end; cb += 64) {
				const int j = cb + lane;
				const bool act = j < end;
				int hs = 0, e = 0;
				uint2 pr = make_uint2(0, 0);
				if (act) hs = H[j], e = E[j], pr = PR[j];
				const int s = t < 4 ? (int)(int8_t)(pr.x >> (t * 8)) : (int)(int8_t)pr.y;
				const int mm = hs + s;
				// F(i,j) = max(fin - (j-cb)*e_ins, max_{k<j}(mm_k - oe_ins - (j-1-k)*e_ins))
				const int g = act ? mm - oe_ins + lane * e_ins : INT32_MIN / 2;
				const int pm = wave_scan_max(g);
				const int pex = wave_shr1(pm, INT32_MIN / 2);
				const int f = max(pex - (lane - 1) * e_ins, fin - lane * e_ins);
				int d = mm >= e ? 0 : 1; // ksw.c:547-550
				int h = max(mm, e);
				d = h >= f ? d : 2;
				h = max(h, f);
				const int t1 = mm - oe_del, e2 = e - e_del; // ksw.c:552-556
				d |= e2 > t1 ? 1 << 2 : 0;
				const int en = max(e2, t1);
				d |= f - e_ins > mm - oe_ins ? 2 << 4 : 0; // ksw.c:557-559
				const int hprev = wave_shr1(h, carry_h);
				if (act) {
					H[j] = hprev, E[j] = en;
					if (want) zi[j - beg] = (uint8_t)d; // ksw.c:561
				}
				const int nact = end - cb;
				if (nact >= 64) {
					carry_h = __builtin_amdgcn_readlane(h, 63);
					fin = max(fin - 64 * e_ins, __builtin_amdgcn_readlane(pm, 63) - 63 * e_ins);
				} else carry_h = __builtin_amdgcn_readlane(h, nact - 1);
			}
			if (lane == 0 && end >= 0) H[end] = carry_h, E[end] = kNegInf; // ksw.c:563
		}
		const int score = uni(H[qlen]); // ksw.c:565

		int n_cigar = 0;
		if (want) { // traceback, ksw.c:566-581
			// the direction bytes were written by other lanes: drain our stores before reading them back
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
			CigarSink cs{cigar_pool + cigar_off, cigar_cap, 0, 0, 0};
			const long long zsize = (long long)n_col * tlen;
			int i = tlen - 1;
			int k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1;
			int which = 0;
			while (i >= 0 && k >= 0) {
				const int ii = which == 2 ? i : i - lane;
				const int kk = which == 1 ? k : k - lane;
				int f = 3; // "stop" for cells off the matrix
				if (ii >= 0 && kk >= 0) {
					long long zo = (long long)ii * n_col + (kk - (ii > w ? ii - w : 0));
					zo = zo < 0 ? 0 : (zo >= zsize ? zsize - 1 : zo); // stay inside the slab on out-of-domain input
					f = z[zo] >> (which << 1) & 3;
				}
				const unsigned long long bm = __ballot(f != which);
				const int run = bm ? __builtin_ctzll(bm) : 64;
				if (which == 0) {
					cs.push(0, run, lane), i -= run, k -= run;
					if (run < 64 && i >= 0 && k >= 0) {
						which = __builtin_amdgcn_readlane(f, run); // 1: came from E, 2: came from F
						if (which == 1) cs.push(2, 1, lane), --i;
						else cs.push(1, 1, lane), --k;
					}
				} else if (which == 1) {
					cs.push(2, run, lane), i -= run;
					if (run < 64 && i >= 0) cs.push(0, 1, lane), --i, --k, which = 0; // gap opened from the diagonal
				} else {
					cs.push(1, run, lane), k -= run;
					if (run < 64 && k >= 0) cs.push(0, 1, lane), --i, --k, which = 0;
				}
			}
			if (i >= 0) cs.push(2, i + 1, lane); // ksw.c:576-577
			if (k >= 0) cs.push(1, k + 1, lane);
			cs.flush(lane);
			n_cigar = cs.nw;
			// words sit at [cap-n, cap) in forward order; move them to [0, n)
			const int shift = cigar_cap - n_cigar;
			if (shift > 0 && n_cigar <= cigar_cap) {
				__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); // lane 0's words -> visible to all lanes
				for (int c = 0; c < n_cigar; c += 64) {
					uint32_t v = 0;
					if (c + lane < n_cigar) v = cs.base[shift + c + lane];
					if (c + lane < n_cigar) cs.base[c + lane] = v;
				}
			}
			if (n_cigar > cigar_cap && lane == 0) atomicExch(err_flag, BMH_E_CIGAR_CAP);
		}
		if (lane == 0) out[idx].score = score, out[idx].n_cigar = n_cigar;
	}
}

// ---- dispatcher -----------------------------------------------------------------------------------
// bin 0: w <= 31  -> global_lane_kernel<64>   (64 tasks per wave, band-relative registers)
// bin 1: w <= 63  -> global_lane_kernel<128>
// bin 3: 32 <= w <= 47 -> global_lane_kernel<96>
//        (an 80-slot instantiation for bands of 32-39 only pays at three waves per SIMD, and under 168 VGPRs its row loop spills
//        49 registers: not used)
// bin 2: wider bands, targets longer than the lane kernels' direction slab, or scores that could leave the
//        16-bit range -> global_kernel (one wave per task, int32 in LDS)
// Bins and the order inside them (by row count) come from the same device-side counting sort as the extension path.
__global__ __launch_bounds__(256) void glb_sort_hist_kernel(const bmh_glb_task_t *__restrict__ tasks,
                                                            const uint32_t *__restrict__ order, long long n,
                                                            uint32_t *__restrict__ hist, uint16_t *__restrict__ binkey,
                                                            DevParams P, int lane_ok, int rows_cap)
{
	__shared__ uint32_t lh[kSortBins * kSortKeysHost];
	for (int t = threadIdx.x; t < kSortBins * kSortKeysHost; t += 256) lh[t] = 0;
	__syncthreads();
	const long long chunk = (n + gridDim.x - 1) / gridDim.x, lo = chunk * blockIdx.x, hi = min(lo + chunk, n);
	const int emax = max(P.e_del, P.e_ins), smax = max(P.bias, P.max_mat);
	for (long long k = lo + threadIdx.x; k < hi; k += 256) {
		const uint32_t idx = order ? order[k] : (uint32_t)k;
		const int qlen = tasks[idx].qlen, tlen = tasks[idx].tlen, w = tasks[idx].w;
		const int worst = P.o_del + P.o_ins + emax * (qlen + tlen) + smax * max(qlen, tlen); // |score| bound of any cell
		int bin = 2;
		// (the lane kernels take the sign of 16-bit differences such as m - e - o_del: with |m|, |e| < 12000 and the -16384 sentinel that
		// stays inside +-32767 as long as the gap-open penalties are not absurd)
		if (lane_ok && tlen <= rows_cap && worst < 12000 && P.o_del + P.o_ins < 4000 && w >= 0) bin = w <= 31 ? 0 : w <= 47 ? 3 : (w <= 63 ? 1 : 2);
		// inside a lane bin: rows first (lanes of a wave run until their longest target ends), then band width (a wave
		// computes and stores the 8-slot blocks that ANY of its lanes needs, and its lanes' tracebacks share cache lines
		// when they sit in the same block)
		const int bk = bin * kSortKeysHost + (bin != 2 ? (min(tlen >> 3, 127) << 4 | (min(w, 63) >> (bin == 0 ? 1 : 2) & 15)) : 0);
		binkey[k] = (uint16_t)bk;
		atomicAdd(&lh[bk], 1u);
	}
	__syncthreads();
	for (int t = threadIdx.x; t < kSortBins * kSortKeysHost; t += 256)
		if (lh[t]) atomicAdd(&hist[t], lh[t]);
}

int launch_global(bmh_ctx *ctx, const uint8_t *d_pool, const bmh_glb_task_t *d_tasks, int64_t n,
                  bmh_glb_result_t *d_res, uint32_t *d_cigar, const uint32_t *d_order, int qmax, int tmax,
                  int wmax, int wgate)
{
	if (n <= 0) return BMH_OK;
	int rc;
	const size_t N = (size_t)n;
	const bool lane_ok = ctx->glb_mode == 0;
	const int rows_cap = tmax < 512 ? tmax : 512; // rows of the lane kernels' direction slab
	uint32_t *counts, *lists;
	if ((rc = sort_tasks_begin(ctx, n, &counts, &lists))) return rc;
	uint32_t *hist = counts + 16;
	uint16_t *binkey = (uint16_t *)(hist + (size_t)kSortBins * kSortKeysHost);
	long long cg = (n + 1023) / 1024;
	if (cg > 512) cg = 512;
	hipLaunchKernelGGL(glb_sort_hist_kernel, dim3((unsigned)cg), dim3(256), 0, ctx->stream, d_tasks, d_order, (long long)n, hist,
	                   binkey, ctx->dev, lane_ok ? 1 : 0, rows_cap);
	if ((rc = sort_tasks_finish(ctx, n, d_order, (unsigned)cg))) return rc;
	const bool tm = ctx->timing;
	if (tm) BMH_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
	if (lane_ok) {
		if (tm) BMH_HIP(ctx, hipEventRecord(ctx->ev_gbin[0], ctx->stream));
		if ((rc = launch_global_lane(ctx, 64, d_pool, d_tasks, n, d_res, d_cigar, lists, counts + 0, rows_cap))) return rc;
		if (tm) BMH_HIP(ctx, hipEventRecord(ctx->ev_gbin[1], ctx->stream));
		// bands of 32..47 take the 96-slot instantiation (two waves per SIMD, three quarters of the slots), 48..63 the 128-slot one
		if (wgate > 31 && (rc = launch_global_lane(ctx, 96, d_pool, d_tasks, n, d_res, d_cigar, lists + 3 * N, counts + 3, rows_cap)))
			return rc;
		if (wgate > 47 && (rc = launch_global_lane(ctx, 128, d_pool, d_tasks, n, d_res, d_cigar, lists + N, counts + 1, rows_cap)))
			return rc;
	}
	if (tm) BMH_HIP(ctx, hipEventRecord(ctx->ev_gbin[2], ctx->stream));
	{ // bin 2: the wave kernel
		const uint32_t *lst = lists + 2 * N, *cnt = counts + 2;
		const int qcap = (qmax + 63) & ~63;
		const size_t state = (size_t)8 * (qcap + 2) + (size_t)8 * qcap + 32;
		const long long ncol = qmax < 2LL * wmax + 1 ? qmax : 2LL * wmax + 1;
		long long zcap = ncol * (long long)tmax;
		zcap = (zcap + 15) & ~15LL;
		if (zcap < 16) zcap = 16;
		if (state > 160 * 1024) return BMH_E_RANGE;
		const bool zlds = state + (size_t)zcap <= 64 * 1024; // keep >= 2 blocks per CU in the LDS variant
		long long grid = n < (1LL << 20) ? n : (1LL << 20);
		if (lane_ok && grid > 8192) grid = 8192; // normally (almost) empty when the lane kernels are on
		if (zlds) {
			hipLaunchKernelGGL(global_kernel<true>, dim3((unsigned)grid), dim3(64), state + (size_t)zcap, ctx->stream, d_pool,
			                   d_tasks, lst, cnt, (long long)n, d_res, d_cigar, ctx->dev, qcap, zcap, (uint8_t *)nullptr, ctx->d_err);
		} else {
			long long budget = 8LL << 30; // HBM scratch for direction bytes, one slab per resident block
			long long g = budget / zcap;
			if (g < 1) return BMH_E_RANGE;
			if (g > 8192) g = 8192;
			if (grid > g) grid = g;
			if ((rc = ensure(ctx, ctx->d_scratch, (size_t)grid * (size_t)zcap))) return rc;
			hipLaunchKernelGGL(global_kernel<false>, dim3((unsigned)grid), dim3(64), state, ctx->stream, d_pool, d_tasks, lst, cnt,
			                   (long long)n, d_res, d_cigar, ctx->dev, qcap, zcap, (uint8_t *)ctx->d_scratch.p, ctx->d_err);
		}
		BMH_HIP(ctx, hipGetLastError());
	}
	if (tm) {
		BMH_HIP(ctx, hipEventRecord(ctx->ev_gbin[3], ctx->stream));
		BMH_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
		ctx->ev_valid = true, ctx->ev_gbin_valid = lane_ok;
	}
	return BMH_OK;
}

} // namespace bmh
