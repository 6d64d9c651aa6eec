// extend_lds.hip -- banded z-drop affine-gap seed extension, any query length.
//
// Replaces ksw_extend2 (reference bwa-0.7.8/ksw.c:379-476; spec SURVEY.md A.1).
// One wave64 per task, ROW-SYNCHRONOUS: the reference's adaptive [beg,end)
// interval of row i+1 depends on the finished row i (ksw.c:463-466), so rows
// are walked in order and the 64 lanes own 64 consecutive query columns of the
// live interval (a window sliding with `beg`; wider rows take several chunks).
//
//   * H (shifted, = eh[j].h) and E (= eh[j].e) live in LDS as one packed dword
//     per column (u16|u16), the query profile as 5 signed bytes per column
//     (one ds_read_b64): per row and chunk 2 LDS reads + 1 LDS write per lane.
//   * The in-row dependency F(i,j+1)=max(F(i,j)-e_ins, H(i,j)-o_ins-e_ins) is a
//     max-plus prefix scan over lanes, done with 6 DPP steps (no LDS);
//     valid because o_ins >= 0 (SURVEY.md §7 hard part 1).
//   * Row maximum and its right-most column come from ONE wave max-reduction
//     of the key (h<<16 | j)  (ties -> larger j, ksw.c:434).
//   * beg/end, the z-drop test and the m==0 exit are wave-uniform scalars; the
//     interval update scans H for zeros with ballots.
//   * The target row base is wave-uniform: 4 bases per lane are kept in one
//     VGPR and fetched with v_readlane, reloaded every 256 rows.
//
// Integer only (no MFMA: this is a max-plus recurrence, not a contraction).
#include "bmh_ctx.h"
#include "bmh_device.h"

namespace bmh {

__device__ __forceinline__ void store_result(bmh_ext_result_t *o, int score, int qle, int tle, int gtle,
                                             int gscore, int max_off)
{
	int *p = (int *)o;
	p[0] = score, p[1] = qle, p[2] = tle, p[3] = gtle, p[4] = gscore, p[5] = max_off;
}

__global__ __launch_bounds__(64) void extend_lds_kernel(const uint8_t *__restrict__ pool,
                                                        const bmh_ext_task_t *__restrict__ tasks,
                                                        const uint32_t *__restrict__ order,
                                                        const uint32_t *__restrict__ count, long long n,
                                                        bmh_ext_result_t *__restrict__ out, DevParams P,
                                                        int qcap, int *__restrict__ err_flag)
{
	extern __shared__ __align__(16) unsigned char smem[];
	uint32_t *HE = (uint32_t *)smem;                                   // [qcap+2]  E<<16 | Hs
	uint2 *PR = (uint2 *)(smem + (((size_t)4 * (qcap + 2) + 15) & ~(size_t)15)); // [qcap] 5 score bytes
	int8_t *smat = (int8_t *)(PR + qcap);                              // [32]
	const int lane = threadIdx.x;
	const int oe_del = P.o_del + P.e_del, oe_ins = P.o_ins + P.e_ins;
	const int e_del = P.e_del, e_ins = P.e_ins;

	if (lane < 25) smat[lane] = (int8_t)mat_at(P, lane);

	if (count) n = *count; // bin size produced on the device by classify_kernel
	for (long long slot = blockIdx.x; slot < n; slot += gridDim.x) {
		const uint32_t idx = order ? order[slot] : (uint32_t)slot;
		const uint4 *tp = (const uint4 *)(tasks + idx);
		const uint4 ta = tp[0], tb = tp[1];
		const uint64_t q_off = (uint64_t)(uint32_t)uni(ta.y) << 32 | (uint32_t)uni(ta.x);
		const uint64_t t_off = (uint64_t)(uint32_t)uni(ta.w) << 32 | (uint32_t)uni(ta.z);
		const int qlen = uni(tb.x & 0xffff), tlen = uni(tb.x >> 16);
		int h0 = uni(tb.y);
		int w = uni((int)(int16_t)(tb.z & 0xffff));
		const int end_bonus = uni((int)(int16_t)(tb.z >> 16));
		const bool qrev = uni(tb.w) & BMH_F_QREV, trev = uni(tb.w) & BMH_F_TREV, tpac = uni(tb.w) & BMH_F_TPAC;
		if (h0 < 0) h0 = 0; // ksw.c:384

		if (qlen > qcap || h0 + qlen * P.max_mat > kScoreLimit) { // outside the supported range: fail loudly
			if (lane == 0) {
				store_result(out + idx, INT32_MIN, 0, 0, 0, 0, 0);
				atomicExch(err_flag, BMH_E_RANGE);
			}
			continue;
		}

		// first row (closed form of ksw.c:394-396) and query profile (ksw.c:389-392)
		for (int j = lane; j <= qlen; j += 64) {
			HE[j] = (uint32_t)(j == 0 ? h0 : max(0, h0 - P.o_ins - j * e_ins));
			if (j < qlen) {
				const int qb = seq_base(pool, q_off, j, qrev);
				uint32_t lo = 0;
				for (int k = 0; k < 4; ++k) lo |= (uint32_t)(uint8_t)smat[k * 5 + qb] << (8 * k);
				PR[j] = make_uint2(lo, (uint32_t)(uint8_t)smat[20 + qb]);
			}
		}

		// band clamp, ksw.c:398-406
		w = min(w, max(1, band_cap(qlen, P.max_mat, end_bonus, P.o_ins, e_ins)));
		w = min(w, max(1, band_cap(qlen, P.max_mat, end_bonus, P.o_del, e_del)));

		int beg = 0, end = qlen, best = h0, bi = -1, bj = -1, gi = -1, gscore = -1, max_off = 0;
		uint32_t tv = 0;

		for (int i = 0; i < tlen; ++i) {
			if ((i & 255) == 0) { // stage the next 256 target bases, 4 per lane
				tv = 0;
				for (int k = 0; k < 4; ++k) {
					const int r = i + lane * 4 + k;
					if (r < tlen) tv |= (uint32_t)tgt_base(pool, P, t_off, r, trev, tpac) << (8 * k);
				}
			}
			const int tw = __builtin_amdgcn_readlane((int)tv, (i >> 2) & 63);
			const int t = (tw >> ((i & 3) * 8)) & 0xff;
			const int left0 = max(0, h0 - (P.o_del + e_del * (i + 1))); // ksw.c:415-416
			beg = max(beg, i - w);                                       // ksw.c:418-420
			end = min(end, min(i + w + 1, qlen));

			int carry_h = left0; // H(i, cb-1): what column cb stores as its shifted H
			int fin = 0;         // F(i, cb)
			int rkey = -1;
			for (int cb = beg; cb < end; cb += 64) { // ksw.c:421-445, 64 columns at a time
				const int j = cb + lane;
				const bool act = j < end;
				uint32_t he = 0;
				uint2 pr = make_uint2(0, 0);
				if (act) he = HE[j], pr = PR[j];
				const int M = (int)(he & 0xffff), e = (int)(he >> 16);
				const int s = t < 4 ? (int)(int8_t)(pr.x >> (t * 8)) : (int)(int8_t)pr.y;
				const int hh = max(M + s, e);
				const int g = act ? max(hh - oe_ins, 0) + lane * e_ins : kNegInf16;
				const int pm = wave_scan_max(g);
				const int pex = wave_shr1(pm, kNegInf16);
				const int F = max(max(pex - (lane - 1) * e_ins, fin - lane * e_ins), 0);
				const int h = max(hh, F);
				const int en = max(max(e - e_del, h - oe_del), 0);
				const int hprev = wave_shr1(h, carry_h);
				if (act) HE[j] = (uint32_t)en << 16 | (uint32_t)hprev;
				const int nact = end - cb;
				if (nact >= 64) {
					carry_h = __builtin_amdgcn_readlane(h, 63);
					fin = max(fin - 64 * e_ins, __builtin_amdgcn_readlane(pm, 63) - 63 * e_ins);
				} else carry_h = __builtin_amdgcn_readlane(h, nact - 1);
				rkey = max(rkey, wave_reduce_max(act ? (h << 16 | j) : -1));
			}
			if (lane == 0) HE[end] = (uint32_t)carry_h; // eh[end].h = h1, eh[end].e = 0  (ksw.c:446)

			const int m = rkey < 0 ? 0 : rkey >> 16;
			const int mj = rkey < 0 ? -1 : rkey & 0xffff;
			if ((beg < end ? end : beg) == qlen) { // ksw.c:447-450 (`j == qlen` on the loop variable)
				if (!(gscore > carry_h)) gi = i;
				gscore = max(gscore, carry_h);
			}
			if (m == 0) break; // ksw.c:451
			if (m > best) {    // ksw.c:452-454
				best = m, bi = i, bj = mj;
				max_off = max(max_off, abs(mj - i));
			} else if (P.zdrop > 0) { // ksw.c:455-461
				const int di = i - bi, dj = mj - bj;
				if (di > dj) {
					if (best - m - (di - dj) * e_del > P.zdrop) break;
				} else {
					if (best - m - (dj - di) * e_ins > P.zdrop) break;
				}
			}
			// live-interval update, ksw.c:463-466: nearest zero of Hs left of mj / right of mj+2
			int nb = beg;
			for (int hi = mj; hi >= beg; hi -= 64) {
				const int j = hi - lane;
				const bool z = j >= beg && (HE[j] & 0xffff) == 0;
				const unsigned long long bm = __ballot(z);
				if (bm) {
					nb = hi - __builtin_ctzll(bm) + 1;
					break;
				}
			}
			int ne;
			for (int lo = mj + 2;; lo += 64) {
				const int j = lo + lane;
				bool z = true;
				if (j <= end) z = (HE[j] & 0xffff) == 0;
				const unsigned long long bm = __ballot(z);
				if (bm) {
					ne = lo + __builtin_ctzll(bm);
					break;
				}
			}
			beg = nb, end = ne;
		}
		if (lane == 0) store_result(out + idx, best, bj + 1, bi + 1, gi + 1, gscore, max_off);
	}
}

// ---- launcher: every task listed in d_order[0..*d_count) (or 0..n when d_count is null)
int launch_extend_lds(bmh_ctx *ctx, const uint8_t *d_pool, const bmh_ext_task_t *d_tasks, int64_t n,
                      bmh_ext_result_t *d_res, const uint32_t *d_order, const uint32_t *d_count, int qmax, long long grid_cap)
{
	if (n <= 0) return BMH_OK;
	const int qcap = (qmax + 63) & ~63;
	const size_t shmem = (((size_t)4 * (qcap + 2) + 15) & ~(size_t)15) + (size_t)8 * qcap + 32;
	if (shmem > 160 * 1024) return BMH_E_RANGE;
	long long grid = n < kPersistentGrid ? n : kPersistentGrid;
	if (grid_cap > 0 && grid > grid_cap) grid = grid_cap;
	hipLaunchKernelGGL(extend_lds_kernel, dim3((unsigned)grid), dim3(64), shmem, ctx->stream, d_pool, d_tasks, d_order,
	                   d_count, (long long)n, d_res, ctx->dev, qcap, ctx->d_err);
	BMH_HIP(ctx, hipGetLastError());
	return BMH_OK;
}

} // namespace bmh
