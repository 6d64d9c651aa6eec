// sw_generic.hip -- local Smith-Waterman (ksw_align2, reference bwa-0.7.8/ksw.c:341-364) for tasks of ANY size:
// one lane per task, the DP row in an HBM slab laid out [8-column chunk][lane][8] so that a lane moves 8 columns with
// one 32-byte access and the 64 lanes of a wave touch 2 KB contiguously.  This is the catch-all behind the register kernels of sw_lane.hip: simple, exact, slow.
//
// What is computed (derived from ksw_qinit/ksw_u8/ksw_i16, ksw.c:62-331, and pinned against them by the oracle):
// the striped SSE2 code equals the textbook affine-gap recurrence over the query PADDED to Q = p*slen columns
// (p = 16 in byte mode, 8 in word mode; slen = ceil(qlen/p); pad columns score 0) with ONE deviation: E(i+1,j) is
// taken from H before the lazy-F pass, i.e. from max(M, E, Fseg) where Fseg only knows gaps opened inside the
// column's own segment [k*slen,(k+1)*slen).  Needs o_ins > 0 (with o_ins == 0 the reference's lazy-F loop stops after
// one column and is not a closed recurrence; such parameter sets are refused on the host).
#include "bmh_ctx.h"
#include "bmh_device.h"
#include "sw_common.h"

namespace bmh {

// one pass of ksw_u8 / ksw_i16 for the calling lane.  he/qc/rm are this wave's slabs, already offset by the lane.
struct SwSeq {
	const uint8_t *pool;
	uint64_t q_off, t_off;
	bool qrev, qcomp, trev, tpac;
	int qfold, tfold; // second pass: query base k = q(qfold-k); target row r = r <= tfold ? t(tfold-r) : t(r)
};

__device__ __forceinline__ int sw_qbase(const SwSeq &s, int k)
{
	const int kk = s.qfold >= 0 ? s.qfold - k : k;
	int c = seq_base(s.pool, s.q_off, kk, s.qrev);
	c = c > 4 ? 4 : c;
	return s.qcomp && c < 4 ? 3 - c : c;
}

__device__ __forceinline__ int sw_tbase(const SwSeq &s, const DevParams &P, int r)
{
	const int rr = r <= s.tfold ? s.tfold - r : r;
	const int c = tgt_base(s.pool, P, s.t_off, rr, s.trev, s.tpac);
	return c > 4 ? 4 : c;
}

// he / qc hold 8 columns per lane contiguously ([chunk][lane][8]): one 32-byte and one 8-byte access per 8 cells, the 8
// cells themselves are computed from registers.
__device__ SwCore sw_pass_generic(const SwSeq &seq, const DevParams &P, const int *smat, bool byte_mode, int qlen,
                                  int tlen, int minsc, int endsc, uint32_t *he, uint8_t *qc, uint16_t *rm)
{
	const int p = byte_mode ? 16 : 8;
	const int slen = (qlen + p - 1) / p, Q = slen * p, NC = Q / 8; // Q is a multiple of 8
	const int o_del = P.o_del, e_del = P.e_del, o_ins = P.o_ins, e_ins = P.e_ins;
	SwCore r;
	r.score = 0, r.te = -1, r.qe = -1, r.score2 = -1, r.te2 = -1;
	for (int c = 0; c < NC; ++c) {
		uint32_t lo = 0, hi = 0;
		for (int k = 0; k < 8; ++k) {
			const int j = 8 * c + k;
			const uint32_t code = (uint32_t)(j < qlen ? sw_qbase(seq, j) : 5); // code 5 = pad column, scores 0 (ksw.c:98,107)
			if (k < 4) lo |= code << (8 * k);
			else hi |= code << (8 * (k - 4));
		}
		*(uint2 *)(qc + (size_t)c * 512) = make_uint2(lo, hi);
		uint4 *h = (uint4 *)(he + (size_t)c * 512);
		h[0] = make_uint4(0, 0, 0, 0), h[1] = make_uint4(0, 0, 0, 0);
	}
	int gmax = 0, te = -1, qe = -1, nrows = 0;
	bool ovf = false;
	for (int i = 0; i < tlen && slen > 0; ++i) {
		const int *row = smat + sw_tbase(seq, P, i) * 8;
		int fseg = 0, ffull = 0, diag = 0, imax = 0, arg = -1, js = 0;
		for (int c = 0; c < NC; ++c) {
			uint4 *hp4 = (uint4 *)(he + (size_t)c * 512);
			const uint4 h0 = hp4[0], h1 = hp4[1];
			const uint2 qq = *(const uint2 *)(qc + (size_t)c * 512);
			uint32_t x[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
			for (int k = 0; k < 8; ++k) {
				const int code = (int)((k < 4 ? qq.x >> (8 * k) : qq.y >> (8 * (k - 4))) & 0xff);
				const int hold = (int)(x[k] & 0xffff), e = (int)(x[k] >> 16);
				const int mm = max(diag + row[code], 0);
				if (js == 0) fseg = 0; // every vector lane of the reference starts its segment with f = 0 (ksw.c:139)
				js = js + 1 == slen ? 0 : js + 1;
				const int hpre = max(max(mm, e), fseg); // what the main loop stores, ksw.c:151-154
				const int h = max(hpre, ffull);         // after the lazy-F loop, ksw.c:165-176
				if (h > imax) imax = h, arg = 8 * c + k; // ties -> smallest query index, ksw.c:204-206
				diag = hold;
				const int en = max(max(e, max(hpre - o_del, 0)) - e_del, 0); // ksw.c:155-158, from the uncorrected H
				const int t = max(hpre - o_ins, 0);
				fseg = max(max(fseg, t) - e_ins, 0); // ksw.c:160-162
				ffull = max(max(ffull, t) - e_ins, 0);
				x[k] = (uint32_t)en << 16 | (uint32_t)h;
			}
			hp4[0] = make_uint4(x[0], x[1], x[2], x[3]), hp4[1] = make_uint4(x[4], x[5], x[6], x[7]);
		}
		nrows = i + 1;
		if (rm) rm[(size_t)i * 64] = (uint16_t)imax;
		if (imax > gmax) { // ksw.c:190-195
			gmax = imax, te = i, qe = arg;
			ovf = byte_mode && gmax + P.sw_shift >= 255; // ksw.c:194
			if (ovf || gmax >= endsc) break;
		}
	}
	r.score = ovf ? 255 : gmax, r.te = te;
	if (!ovf) {
		r.qe = te < 0 ? 0 : qe; // Hmax stays all zero when nothing scored: index 0 wins, ksw.c:204-206
		if (rm) sw_second_best(rm, 64, nrows, minsc, r.score, te, P.max_mat, &r.score2, &r.te2);
	}
	return r;
}

__global__ __launch_bounds__(64) void sw_generic_kernel(const uint8_t *__restrict__ pool,
                                                        const bmh_sw_task_t *__restrict__ tasks,
                                                        const uint32_t *__restrict__ order,
                                                        const uint32_t *__restrict__ count, long long n,
                                                        bmh_sw_result_t *__restrict__ out, DevParams P, uint8_t *slab,
                                                        long long slab_stride, int qcap, int tcap, int *err_flag, int wave_cols)
{
	__shared__ int smat[6 * 8];
	const int lane = threadIdx.x;
	if (lane < 48) {
		const int t = lane >> 3, q = lane & 7;
		smat[lane] = t < 5 && q < 5 ? mat_at(P, t * 5 + q) : 0;
	}
	__syncthreads();
	const long long cnt = count ? (long long)*count : n;
	uint8_t *base = slab + (size_t)blockIdx.x * (size_t)slab_stride;
	uint32_t *he = (uint32_t *)base + lane * 8;                                    // [qcap/8][64][8] u32
	uint16_t *rm = (uint16_t *)(base + (size_t)qcap * 256) + lane;                 // [tcap][64] u16
	uint8_t *qc = base + (size_t)qcap * 256 + (size_t)tcap * 128 + lane * 8;       // [qcap/8][64][8] u8
	for (long long c0 = (long long)blockIdx.x * 64; c0 < cnt; c0 += (long long)gridDim.x * 64) {
		const long long pos = c0 + lane;
		if (pos >= cnt) continue;
		const uint32_t idx = order ? order[pos] : (uint32_t)pos;
		const bmh_sw_task_t tk = tasks[idx];
		const int qlen = tk.qlen, tlen = (int)tk.tlen;
		const uint32_t xtra = tk.xtra;
		const bool byte_mode = xtra & BMH_SW_XBYTE;
		const int p = byte_mode ? 16 : 8;
		const int Q = (qlen + p - 1) / p * p;
		if (wave_cols > 0 && sw_wave_takes(P, qlen, xtra, wave_cols)) continue; // sw_wave_kernel has been through this list
		bmh_sw_result_t res;
		res.score = 0, res.te = res.qe = res.score2 = res.te2 = res.tb = res.qb = -1, res.rsv = 0;
		if (Q > qcap || tlen > tcap || sw_task_out_of_range(P, qlen, xtra)) {
			res.score = INT32_MIN;
			out[idx] = res;
			atomicExch(err_flag, BMH_E_RANGE);
			continue;
		}
		SwSeq seq;
		seq.pool = pool, seq.q_off = tk.q_off, seq.t_off = tk.t_off;
		seq.qrev = tk.flags & BMH_F_QREV, seq.qcomp = tk.flags & BMH_F_QCOMP, seq.trev = tk.flags & BMH_F_TREV;
		seq.tpac = tk.flags & BMH_F_TPAC, seq.qfold = -1, seq.tfold = -1;
		const int thr = (int)(xtra & 0xffff);
		const int minsc = (xtra & BMH_SW_XSUBO) ? thr : 0x10000, endsc = (xtra & BMH_SW_XSTOP) ? thr : 0x10000; // ksw.c:131-132
		const SwCore f = sw_pass_generic(seq, P, smat, byte_mode, qlen, tlen, minsc, endsc, he, qc,
		                                 (xtra & BMH_SW_XSUBO) ? rm : nullptr);
		res.score = f.score, res.te = f.te, res.qe = f.qe, res.score2 = f.score2, res.te2 = f.te2;
		const bool second = (xtra & BMH_SW_XSTART) && !((xtra & BMH_SW_XSUBO) && f.score < thr); // ksw.c:354
		if (second && f.qe < 0) { // byte overflow + start positions: the reference reads uninitialised memory here
			res.score = INT32_MIN;
			out[idx] = res;
			atomicExch(err_flag, BMH_E_RANGE);
			continue;
		}
		if (second) { // ksw.c:355-361
			seq.qfold = f.qe, seq.tfold = f.te;
			const SwCore rr = sw_pass_generic(seq, P, smat, byte_mode, f.qe + 1, tlen, 0x10000, f.score, he, qc, nullptr);
			if (rr.score == f.score) res.tb = f.te - rr.te, res.qb = f.qe - rr.qe;
		}
		out[idx] = res;
	}
}

int launch_sw_generic(bmh_ctx *ctx, const uint8_t *d_pool, const bmh_sw_task_t *d_tasks, int64_t n,
                      bmh_sw_result_t *d_res, const uint32_t *d_order, const uint32_t *d_count, int qcap, int tcap, int wave_cols)
{
	if (n <= 0) return BMH_OK;
	qcap = (qcap + 15) / 16 * 16;
	const size_t stride = ((size_t)qcap * (256 + 64) + (size_t)tcap * 128 + 255) & ~(size_t)255;
	long long grid = std::min<long long>((n + 63) / 64, 2048);
	while (grid > 1 && stride * (size_t)grid > ((size_t)8 << 30)) grid /= 2; // keep the workspace below 8 GB
	int rc;
	if ((rc = ensure(ctx, ctx->d_sw, stride * (size_t)grid))) return rc;
	hipLaunchKernelGGL(sw_generic_kernel, dim3((unsigned)grid), dim3(64), 0, ctx->stream, d_pool, d_tasks, d_order,
	                   d_count, (long long)n, d_res, ctx->dev, (uint8_t *)ctx->d_sw.p, (long long)stride, qcap, tcap,
	                   ctx->d_err, wave_cols);
	BMH_HIP(ctx, hipGetLastError());
	return BMH_OK;
}

} // namespace bmh
