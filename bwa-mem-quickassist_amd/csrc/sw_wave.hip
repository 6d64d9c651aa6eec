// sw_wave.hip -- local Smith-Waterman (ksw_align2, reference bwa-0.7.8/ksw.c:114-364) with one WAVE per task, for
// batches too small to fill the chip with one lane per task.
//
// sw_lane.hip gives a task to one lane: 64 tasks per wave at full packed-16 throughput, but a launch takes as long as ONE
// lane needs for its whole matrix -- 3 to 8 ms for a 150 bp mate against its rescue window, however few tasks there are.
// The preload shim's phase 1 (a few dozen mem_chain2aln_short tasks per batch of reads) and its mate rescue (about 1 500
// tasks per slice) waited for exactly that.  Here the 64 lanes of a wave share one task's row: lane l owns columns
// [l*CPL, (l+1)*CPL) and a row costs about 100 instructions, so a task takes tenths of a millisecond and n tasks run on n
// waves side by side.  Both passes of ksw_align2 (ksw.c:341-364) run back to back in the same wave.
//
// A row is column-parallel because the recurrence sw_generic.hip states (the striped code's deviation included) has its
// only in-row dependency in F, and F is a max-plus prefix scan: with a(j) = max(M(i,j), E(i,j)) -- both from row i-1 --
//   Ffull(j) = max(0, max_{k<j}            a(k) - o_ins - e_ins*(j-k))      lazy-F value, ksw.c:165-176
//   Fseg(j)  = max(0, max_{seg(j)<=k<j}    a(k) - o_ins - e_ins*(j-k))      what the main loop sees, ksw.c:139,160-162
//   Hpre = max(a, Fseg),  H = max(Hpre, Ffull),  E' = max(0, max(E, Hpre - o_del) - e_del)
// (Fseg <= Ffull everywhere, so H - o_ins never beats the terms already in the scans.)  Both are exclusive prefix maxima
// of w(k) = a(k) - o_ins - e_ins + e_ins*k: the segmented one adds seg(k)*2^18 to w, which makes every entry of an earlier
// segment smaller than any of the current one, and is read back as "nothing yet" when it falls below the segment's floor.
// Scans run lane-locally over the CPL columns, then across lanes with six DPP steps (wave_scan_max).
//
// Only tasks the register kernels would take (sw_lane_bin: the striped byte or word arithmetic cannot saturate) run here;
// the others are left untouched for sw_generic_kernel.
#include <algorithm>

#include "bmh_ctx.h"
#include "bmh_device.h"
#include "sw_common.h"

namespace bmh {

constexpr int kSwWaveRows = 16384; // row maxima live in LDS (dynamic, two bytes per row of the batch's longest target)
constexpr int kSegBig = 1 << 18;   // > the span of w: scores < 32000, e_ins * column < 256 * 512

struct SwWaveSeq {
	const uint8_t *pool;
	uint64_t q_off, t_off;
	bool qrev, qcomp, trev, tpac;
	int qfold, tfold; // second pass: query base k = q(qfold-k); target row r = r <= tfold ? t(tfold-r) : t(r)
};

__device__ __forceinline__ int sww_qbase(const SwWaveSeq &s, int k)
{
	const int kk = s.qfold >= 0 ? s.qfold - k : k;
	int c = seq_base(s.pool, s.q_off, kk, s.qrev);
	c = c > 4 ? 4 : c;
	return s.qcomp && c < 4 ? 3 - c : c;
}

__device__ __forceinline__ int sww_tbase(const SwWaveSeq &s, const DevParams &P, int r)
{
	const int rr = r <= s.tfold ? s.tfold - r : r;
	const int c = tgt_base(s.pool, P, s.t_off, rr, s.trev, s.tpac);
	return c > 4 ? 4 : c;
}

// one pass of ksw_u8 / ksw_i16 by the whole wave; every lane returns the same SwCore
template <int CPL>
__device__ SwCore sw_pass_wave(const SwWaveSeq &seq, const DevParams &P, const uint2 *srow, int segs, int qlen, int tlen,
                               int minsc, int endsc, uint16_t *rm)
{
	const int lane = threadIdx.x & 63;
	const int slen = (qlen + segs - 1) / segs, Q = slen * segs;
	const int o_del = P.o_del, e_del = P.e_del, e_ins = P.e_ins, g_ins = P.o_ins + P.e_ins;
	// per column: the byte selector of its substitution score in the row's 8-byte table {A, C, G, T, N, pad = 0},
	// the constants of the two scans, and the tag that breaks ties towards the smallest column (ksw.c:204-206)
	uint32_t sel[CPL], tag[CPL];
	int wadd[CPL], sbig[CPL], fsub[CPL];
	bool real[CPL];
#pragma unroll
	for (int c = 0; c < CPL; ++c) {
		const int j = lane * CPL + c;
		const int code = j < qlen ? sww_qbase(seq, j) : 5; // 5: a pad column of the striped layout, scores 0 (ksw.c:98,107)
		real[c] = j < Q;
		sel[c] = 0x0c0c0c00u | (uint32_t)code;
		wadd[c] = e_ins * j - g_ins;                    // w(j) = a(j) + wadd: what column j offers the columns to its right
		sbig[c] = kSegBig * (slen > 0 ? j / slen : 0); // ... + this in the segmented scan
		fsub[c] = e_ins * (j - 1);                      // F(j) = prefix max of w - e_ins*(j-1): a(k) - o_ins - e_ins*(j-k)
		tag[c] = 511u - (uint32_t)j;
	}
	int H[CPL], E[CPL];
#pragma unroll
	for (int c = 0; c < CPL; ++c) H[c] = E[c] = 0;
	SwCore r;
	r.score = 0, r.te = -1, r.qe = -1, r.score2 = -1, r.te2 = -1;
	int gmax = 0, te = -1, qe = -1, nrows = 0;
	int tc = 4;
	constexpr int NEG = INT32_MIN / 2;
	for (int i = 0; i < tlen && slen > 0; ++i) {
		if ((i & 63) == 0) tc = i + lane < tlen ? sww_tbase(seq, P, i + lane) : 4; // the next 64 target bases, one per lane
		const int t = __builtin_amdgcn_readlane(tc, i & 63);
		const uint2 sr = srow[t];
		const int hleft = wave_shr1(H[CPL - 1], 0); // H(i-1, j-1) of the lane's first column
		int a[CPL], pre[CPL], spre[CPL];
		int loc = NEG, sloc = NEG;
#pragma unroll
		for (int c = 0; c < CPL; ++c) {
			const int diag = c == 0 ? hleft : H[c - 1];
			const int sc = (int)__builtin_amdgcn_perm(sr.y, sr.x, sel[c]) - 128;
			a[c] = max(max(diag + sc, 0), E[c]);
			pre[c] = loc, spre[c] = sloc;
			const int w = real[c] ? a[c] + wadd[c] : NEG, ws = real[c] ? w + sbig[c] : NEG;
			loc = max(loc, w), sloc = max(sloc, ws);
		}
		const int ex = wave_shr1(wave_scan_max(loc), NEG), sex = wave_shr1(wave_scan_max(sloc), NEG);
		uint32_t key = 0;
#pragma unroll
		for (int c = 0; c < CPL; ++c) {
			const int m = max(ex, pre[c]), ms = max(sex, spre[c]);
			const int ffull = max(m - fsub[c], 0);
			// an entry of the column's own segment is at least sbig - g_ins (a >= 0); anything smaller is an earlier segment's
			const int fseg = ms >= sbig[c] - g_ins ? max(ms - sbig[c] - fsub[c], 0) : 0;
			const int hpre = max(a[c], fseg), h = max(hpre, ffull);
			E[c] = max(max(E[c], max(hpre - o_del, 0)) - e_del, 0); // ksw.c:155-158, from the uncorrected H
			H[c] = h;
			if (real[c]) key = max(key, (uint32_t)h << 9 | tag[c]);
		}
		const uint32_t rk = (uint32_t)wave_reduce_max((int)key); // scores < 32000: the key stays positive
		const int imax = (int)(rk >> 9), arg = 511 - (int)(rk & 511);
		nrows = i + 1;
		if (rm && lane == 0) rm[i] = (uint16_t)imax;
		if (imax > gmax) { // ksw.c:190-195 (the byte-mode overflow exit cannot trigger for the tasks routed here)
			gmax = imax, te = i, qe = arg;
			if (gmax >= endsc) break;
		}
	}
	r.score = gmax, r.te = te, r.qe = te < 0 ? 0 : qe; // Hmax stays all zero when nothing scored: index 0 wins, ksw.c:204-206
	if (rm) {
		__builtin_amdgcn_s_waitcnt(0);
		__syncthreads();
		int s2 = -1, t2 = -1;
		sw_second_best(rm, 1, nrows, minsc, r.score, te, P.max_mat, &s2, &t2);
		r.score2 = s2, r.te2 = t2;
	}
	return r;
}

template <int CPL>
__global__ __launch_bounds__(64) void sw_wave_kernel(const uint8_t *__restrict__ pool, const bmh_sw_task_t *__restrict__ tasks,
                                                     long long n, bmh_sw_result_t *__restrict__ out, DevParams P,
                                                     int rows_cap, int *__restrict__ err_flag,
                                                     const uint32_t *__restrict__ order, const uint32_t *__restrict__ count)
{
	__shared__ uint2 srow[8];            // [t] = scores of target base t against {A, C, G, T, N, pad, -, -}, biased by 128
	extern __shared__ uint16_t rmax[];   // [rows_cap] row maxima of the task, for the second-best score (ksw.c:181-189)
	const int lane = threadIdx.x;
	if (lane < 8) {
		uint32_t lo = 0x80808080u, hi = 0x80808080u;
		if (lane < 5) {
			lo = 0;
			for (int q = 0; q < 4; ++q) lo |= (uint32_t)(uint8_t)(mat_at(P, lane * 5 + q) + 128) << (8 * q);
			hi = 0x80808000u | (uint32_t)(uint8_t)(mat_at(P, lane * 5 + 4) + 128);
		}
		srow[lane] = make_uint2(lo, hi);
	}
	__syncthreads();
	const long long cnt = count ? (long long)*count : n; // a bin of the dispatcher's sort (order, count) or the whole batch
	for (long long kk = blockIdx.x; kk < cnt; kk += gridDim.x) {
		const long long k = order ? (long long)order[kk] : kk;
		const bmh_sw_task_t tk = tasks[k];
		const int qlen = tk.qlen, tlen = (int)min(tk.tlen, 0x7fffffffu);
		const uint32_t xtra = tk.xtra;
		if (!sw_wave_takes(P, qlen, xtra, 64 * CPL)) continue; // sw_generic_kernel's (launch_sw sends it the rest)
		bmh_sw_result_t res;
		res.score = 0, res.te = res.qe = res.score2 = res.te2 = res.tb = res.qb = -1, res.rsv = 0;
		if (tlen > rows_cap) { // (launch_sw sizes rows_cap from the batch's longest target)
			res.score = INT32_MIN;
			if (lane == 0) out[k] = res, atomicExch(err_flag, BMH_E_RANGE);
			continue;
		}
		SwWaveSeq seq;
		seq.pool = pool, seq.q_off = tk.q_off, seq.t_off = tk.t_off;
		seq.qrev = tk.flags & BMH_F_QREV, seq.qcomp = tk.flags & BMH_F_QCOMP, seq.trev = tk.flags & BMH_F_TREV;
		seq.tpac = tk.flags & BMH_F_TPAC, seq.qfold = -1, seq.tfold = -1;
		const int segs = (xtra & BMH_SW_XBYTE) ? 16 : 8;
		const int thr = (int)(xtra & 0xffff);
		const int minsc = (xtra & BMH_SW_XSUBO) ? thr : 0x10000, endsc = (xtra & BMH_SW_XSTOP) ? thr : 0x10000; // ksw.c:131-132
		__syncthreads(); // the previous task's row maxima have been read
		const SwCore f = sw_pass_wave<CPL>(seq, P, srow, segs, qlen, tlen, minsc, endsc, (xtra & BMH_SW_XSUBO) ? rmax : nullptr);
		res.score = f.score, res.te = f.te, res.qe = f.qe, res.score2 = f.score2, res.te2 = f.te2;
		if ((xtra & BMH_SW_XSTART) && !((xtra & BMH_SW_XSUBO) && f.score < thr)) { // ksw.c:354-361
			seq.qfold = f.qe, seq.tfold = f.te;
			const SwCore rr = sw_pass_wave<CPL>(seq, P, srow, segs, f.qe + 1, tlen, 0x10000, f.score, nullptr);
			if (rr.score == f.score) res.tb = f.te - rr.te, res.qb = f.qe - rr.qe;
		}
		if (lane == 0) out[k] = res;
	}
}

// n tasks on n waves: does a batch with these longest query and target go this way?
bool sw_wave_fits(int64_t n, int qcap, int tcap)
{
	const int q = (std::max(qcap, 1) + 15) / 16 * 16;
	return n > 0 && n <= 32768 && q <= 64 * 5 && tcap <= kSwWaveRows;
}

int launch_sw_wave(bmh_ctx *ctx, const uint8_t *d_pool, const bmh_sw_task_t *d_tasks, int64_t n, bmh_sw_result_t *d_res, int max_cols,
                   int tcap, const uint32_t *d_order, const uint32_t *d_count)
{
	// a list whose length lives on the device (a bin of the sort) gets a grid for a few thousand tasks and strides over the rest
	const unsigned grid = (unsigned)std::min<int64_t>(n, d_count ? 16384 : 65536);
	const int rows_cap = (std::max(tcap, 1) + 63) & ~63;
	const size_t lds = (size_t)rows_cap * sizeof(uint16_t);
	if (max_cols <= 64 * 3)
		hipLaunchKernelGGL(sw_wave_kernel<3>, dim3(grid), dim3(64), lds, ctx->stream, d_pool, d_tasks, (long long)n, d_res, ctx->dev, rows_cap, ctx->d_err,
		                   d_order, d_count);
	else
		hipLaunchKernelGGL(sw_wave_kernel<5>, dim3(grid), dim3(64), lds, ctx->stream, d_pool, d_tasks, (long long)n, d_res, ctx->dev, rows_cap, ctx->d_err,
		                   d_order, d_count);
	BMH_HIP(ctx, hipGetLastError());
	return BMH_OK;
}

} // namespace bmh
