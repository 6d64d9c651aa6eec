// extend_lanex.hip -- the lane-per-task extension kernel for LONG flanks: a task is spread over LPT = 2 or 4
// adjacent lanes of a quad, 128 query columns per lane (qlen <= 256 / 512), 64/LPT tasks per wave.
//
// Same results as ksw_extend2 (reference bwa-0.7.8/ksw.c:379-476) and as extend_lane_kernel, whose row body it
// reuses unchanged.  What is added:
//   * the columns of a task are cut into LPT runs of 128; lane r of the quad owns run r (right-aligned as a whole,
//     so column qlen-1 is the last column of the last lane);
//   * F and the left neighbour are sequential along the row, so the runs are walked in LPT passes: in pass r only
//     the lanes with role r execute (EXEC-masked), the carries (F, H(i,j-1)) hop one lane up by quad_perm DPP
//     between passes; 8-column blocks that no participating lane needs are skipped as before, so a row costs about
//     the union of the live intervals, not LPT*128 columns;
//   * row maximum / zero-map searches are done per lane on its own run and combined inside the quad with
//     quad_perm butterflies; the bookkeeping is then identical in all lanes of the task.
#include "bmh_ctx.h"
#include "bmh_device.h"

namespace bmh {

__device__ __forceinline__ int bfx(int mask, int a, int b) { return __builtin_amdgcn_bitop3_b32(mask, a, b, 0xca); }

constexpr int DPPQ_FROM_LOWER = 0x90; // quad_perm:[0,0,1,2]  lane r <- lane r-1
constexpr int DPPQ_XOR1 = 0xB1;       // quad_perm:[1,0,3,2]
constexpr int DPPQ_XOR2 = 0x4E;       // quad_perm:[2,3,0,1]

template <int LPT>
__device__ __forceinline__ int quad_allmax(int v)
{
	v = max(v, dpp<DPPQ_XOR1>(v, v));
	if (LPT == 4) v = max(v, dpp<DPPQ_XOR2>(v, v));
	return v;
}
template <int LPT>
__device__ __forceinline__ int quad_allmin(int v)
{
	v = min(v, dpp<DPPQ_XOR1>(v, v));
	if (LPT == 4) v = min(v, dpp<DPPQ_XOR2>(v, v));
	return v;
}

template <int LPT, bool SYM>
__global__ __launch_bounds__(64, 2) void extend_lanex_kernel(const uint8_t *__restrict__ pool,
                                                             const bmh_ext_task_t *__restrict__ tasks,
                                                             const uint32_t *__restrict__ order,
                                                             const uint32_t *__restrict__ count, long long n,
                                                             bmh_ext_result_t *__restrict__ out, DevParams P,
                                                             int *__restrict__ err_flag, int min_count)
{
	constexpr int C = 128, TOT = C * LPT, TPW = 64 / LPT;
	constexpr int NW = C / 32, NB = C / 8;
	constexpr int INF = 0x7fff;
	__shared__ uint2 srow[8]; // srow[t] = the 5 signed score bytes mat[t*5 .. t*5+4]
	__shared__ uint32_t qsel4[32 * 64]; // query codes of columns 4v..4v+3 of lane l at [v*64 + l]: the selectors of one v_perm_b32 per four cells
	// the target, kStreamRows rows at a time, one byte per row at [row][lane]: a global load inside the row loop under an exec mask costs
	// an s_waitcnt vmcnt(0) -- a memory round trip -- per DP row (extend_lane.hip)
	constexpr int kStreamRows = 64;
	__shared__ uint8_t strm[kStreamRows * 64];
	const int lane = threadIdx.x, role = lane % LPT, c0 = role * C; // this lane owns global columns [c0, c0+C)
	const int oe_del = P.o_del + P.e_del, oe_ins = P.o_ins + P.e_ins;
	const int e_del = P.e_del, e_ins = P.e_ins;

	if (lane < 5) {
		uint32_t lo = 0;
		for (int q = 0; q < 4; ++q) lo |= (uint32_t)(uint8_t)mat_at(P, lane * 5 + q) << (8 * q);
		srow[lane] = make_uint2(lo, (uint32_t)(uint8_t)mat_at(P, lane * 5 + 4));
	}
	const long long cnt = count ? (long long)*count : n;
	if (cnt < min_count) return; // a handful of tasks is served better by one wave per task (extend_reg_kernel)
	// persistent grid: chunks of TPW tasks, grid stride (see extend_lane_kernel)
	for (long long base = (long long)blockIdx.x * TPW; base < cnt; base += (long long)gridDim.x * TPW) {
	const bool valid = base + lane / LPT < cnt;
	const long long pos = cnt - 1 - (valid ? base + lane / LPT : base); // sorted ascending: expensive waves first
	const uint32_t idx = order ? order[pos] : (uint32_t)pos;

	const uint4 *tp = (const uint4 *)(tasks + idx);
	const uint4 ta = tp[0], tb = tp[1];
	const uint64_t q_off = (uint64_t)ta.y << 32 | ta.x, t_off = (uint64_t)ta.w << 32 | ta.z;
	const int qlen = (int)(tb.x & 0xffff), tlen = (int)(tb.x >> 16);
	const int h0 = max((int)tb.y, 0); // ksw.c:384
	int w = (int)(int16_t)(tb.z & 0xffff);
	const int end_bonus = (int)(int16_t)(tb.z >> 16);
	const bool qrev = tb.w & BMH_F_QREV, trev = tb.w & BMH_F_TREV, tpac = tb.w & BMH_F_TPAC;
	const bool bad = qlen > TOT || qlen < 1 || h0 + qlen * P.max_mat > kScoreLimit;
	if (valid && bad && role == 0) {
		int *p = (int *)(out + idx);
		p[0] = INT32_MIN, p[1] = p[2] = p[3] = p[4] = p[5] = 0;
		atomicExch(err_flag, BMH_E_RANGE);
	}
	const int off = TOT - min(max(qlen, 1), TOT);

	// ---- per-lane column state (ksw.c:389-396), the whole query right-aligned in the TOT columns of the quad
	int HE[C];
#pragma unroll 2
	for (int v = 0; v < C / 4; ++v) {
		int sw = 0;
#pragma unroll
		for (int b = 0; b < 4; ++b) {
			const int j = c0 + 4 * v + b - off;
			int qb = 4;
			if (valid && !bad && j >= 0) qb = seq_base(pool, q_off, j, qrev);
			sw |= qb << (8 * b);
		}
		qsel4[v * 64 + lane] = (uint32_t)sw;
	}
#pragma unroll
	for (int p = 0; p < C; ++p) {
		const int j = c0 + p - off;
		HE[p] = j < 0 ? 0 : (j == 0 ? h0 : max(0, h0 - P.o_ins - j * e_ins));
	}
	w = min(w, max(1, band_cap(qlen, P.max_mat, end_bonus, P.o_ins, e_ins))); // ksw.c:398-406
	w = min(w, max(1, band_cap(qlen, P.max_mat, end_bonus, P.o_del, e_del)));

	int begp = off, endp = TOT, best = h0, bi = -1, bjp = off - 1, maxoff = 0, raw = h0 - P.o_del, gk = -1;
	bool alive = valid && !bad && tlen > 0;
	if (valid && !bad && tlen == 0 && role == 0) { // no rows at all
		int *p = (int *)(out + idx);
		p[0] = h0, p[1] = 0, p[2] = 0, p[3] = 0, p[4] = -1, p[5] = 0;
	}

	for (int i = 0; __builtin_amdgcn_ballot_w64(alive) != 0; ++i) { // i is wave-uniform: all tasks started together
		if ((i & (kStreamRows - 1)) == 0) { // wave-uniform: target rows [i, i + kStreamRows) of every lane still running
#pragma unroll 4
			for (int r = 0; r < kStreamRows; ++r) {
				int tb_ = 0;
				if (alive && i + r < tlen) tb_ = tgt_base(pool, P, t_off, i + r, trev, tpac);
				strm[r * 64 + lane] = (uint8_t)tb_;
			}
		}
		const int tcur = strm[(i & (kStreamRows - 1)) * 64 + lane];
		const uint2 row = srow[min(tcur, 4)];
		begp = max(begp, i - w + off);     // ksw.c:418-420
		endp = min(endp, i + w + 1 + off); // endp <= TOT covers the qlen clamp
		raw -= e_del;
		const int left = max(raw, 0); // first-column value, ksw.c:415-416
		const int gb = alive ? begp : TOT + 1, ge = alive ? endp : TOT + 1; // global interval
		const int lb = gb - c0, le = ge - c0;                               // the same in this lane's coordinates
		int am[NW];
#pragma unroll
		for (int v = 0; v < NW; ++v) {
			const int lo = min(max(lb - 32 * v, 0), 32), hi = min(max(le - 32 * v, 0), 32);
			am[v] = hi > lo ? (int)((0xffffffffu >> (32 - (hi - lo))) << lo) : 0;
		}
		int f = 0, hprev = left, kmax = -1, hlast = -1;
		int nz[NW];
#pragma unroll
		for (int v = 0; v < NW; ++v) nz[v] = 0;
#pragma nounroll
		for (int pass = 0; pass < LPT; ++pass) {
			// All lanes run the same straight-line code; only the lanes whose run is due (role == pass) may change
			// state.  (A divergent `if` here made hipcc shuttle the 160 state registers through AGPRs and scratch.)
			const int minev = role == pass ? -1 : 0;
			if (pass > 0) { // carries of the run to the left: F(i,j) and H(i,j-1), ksw.c:422
				const int fl = dpp<DPPQ_FROM_LOWER>(f, f), hl = dpp<DPPQ_FROM_LOWER>(hprev, hprev);
				f = fl, hprev = hl; // (junk in lanes that are not due; they never use it)
			}
			const int plb = minev ? lb : C + 1, ple = minev ? le : C + 1;
			int pam[NW], pnz[NW];
#pragma unroll
			for (int v = 0; v < NW; ++v) pam[v] = am[v] & minev, pnz[v] = nz[v];
			int pk = kmax, phl = hlast;
#pragma unroll
			for (int b = 0; b < NB; ++b) {
				// needed by a lane that is due iff [8b,8b+8) meets [beg,end]  (end itself receives eh[end])
				if (__builtin_amdgcn_ballot_w64(plb < 8 * b + 8 && ple >= 8 * b) == 0) continue;
				// the cell in 16-bit instructions (bmh_device.h; extend_lane.hip has the same one): every quantity of the recurrence is >= 0
				// except M = H(i-1,j-1) + S, which only enters a signed maximum with E >= 0
#pragma unroll
				for (int q4 = 0; q4 < 2; ++q4) {
					const int sc4 = (int)__builtin_amdgcn_perm(row.y, row.x, qsel4[(2 * b + q4) * 64 + lane]);
#pragma unroll
					for (int c4 = 0; c4 < 4; ++c4) {
						const int p = 8 * b + 4 * q4 + c4;
						const int actv = (pam[p / 32] << (31 - p % 32)) >> 31;
						const int m = c4 == 0 ? add_score<0>(sc4, HE[p]) : c4 == 1 ? add_score<1>(sc4, HE[p]) : c4 == 2 ? add_score<2>(sc4, HE[p]) : add_score<3>(sc4, HE[p]);
						const int e = (int)((unsigned)HE[p] >> 16);
						const int h = max16(max16(m, e), f);                              // ksw.c:430-432
						const int t = subc16(h, oe_del);                                  // max(h - oe_del, 0)
						const int en = maxu16(subc16(e, e_del), t) & actv;                // ksw.c:436-439
						f = maxu16(subc16(f, e_ins), SYM ? t : subc16(h, oe_ins)) & actv; // ksw.c:441-444
						HE[p] = bfx(minev, en << 16 | hprev, HE[p]);                      // eh[j] = {H(i,j-1), E(i+1,j)}
						const int ha = bfx(actv, h, -1);
						pk = max(pk, ha << 16 | (c0 + p));                                // global column in the key
						pnz[p / 32] |= nonzero16(h) << (p % 32);
						hprev = bfx(actv, h, left);
						if (p == C - 1) phl = ha;
					}
				}
			}
			kmax = bfx(minev, pk, kmax), hlast = bfx(minev, phl, hlast);
#pragma unroll
			for (int v = 0; v < NW; ++v) nz[v] = bfx(minev, pnz[v], nz[v]);
		}
		// ---- row end: combine the runs of the quad, then every lane of the task does the same bookkeeping
		kmax = quad_allmax<LPT>(kmax);
		if (role == LPT - 1) { // the lane that owns column qlen-1 (ksw.c:447-450, ties -> later row)
			gk = max(gk, hlast << 16 | i);
			if (alive && ge <= gb && gb == TOT) gk = max(gk, left << 16 | i); // empty row whose loop variable equals qlen
		}
		const int m = kmax < 0 ? 0 : kmax >> 16, mjp = kmax & 0xffff;
		const bool stop0 = kmax < 0x10000;                                   // m == 0 or empty row, ksw.c:451
		const bool upd = alive && m > best;                                 // ksw.c:452-454
		const int dd = (i - bi) - (mjp - bjp);
		const int pen = max(dd * e_del, -dd * e_ins);
		const bool zd = !upd && P.zdrop > 0 && best - m - pen > P.zdrop;    // ksw.c:455-461
		best = upd ? m : best;
		bi = upd ? i : bi;
		bjp = upd ? mjp : bjp;
		maxoff = upd ? max(maxoff, abs(mjp - off - i)) : maxoff;
		// live-interval update, ksw.c:463-466: per-lane zero-map searches in global columns, combined in the quad
		int lzm = -0x10000, fz = INF;
		const int mjl = mjp - c0;
#pragma unroll
		for (int v = 0; v < NW; ++v) {
			const int zz = ~nz[v] & am[v];
			const int lim = mjl - 32 * v; // columns < mj
			const int zl = lim <= 0 ? 0 : (lim >= 32 ? zz : zz & ((1 << lim) - 1));
			if (zl) lzm = c0 + 32 * v + 31 - __builtin_clz(zl);
		}
#pragma unroll
		for (int v = NW - 1; v >= 0; --v) {
			const int zz = ~nz[v] & am[v];
			const int lo = mjl + 1 - 32 * v; // columns > mj
			const int zr = lo >= 32 ? 0 : (lo <= 0 ? zz : zz & (-1 << lo));
			if (zr) fz = c0 + 32 * v + __builtin_ctz(zr);
		}
		lzm = max(quad_allmax<LPT>(lzm), gb - 2 + (left == 0));
		fz = quad_allmin<LPT>(fz);
		begp = lzm + 2;
		endp = min(fz == INF ? ge + 1 : fz + 1, TOT);
		const bool done = alive && (stop0 || zd || i + 1 >= tlen);
		if (done) { // results, ksw.c:470-475
			alive = false;
			int *p = (int *)(out + idx);
			if (role == 0) p[0] = best, p[1] = bjp - off + 1, p[2] = bi + 1, p[5] = maxoff;
			if (role == LPT - 1) p[3] = gk < 0 ? 0 : (gk & 0xffff) + 1, p[4] = gk < 0 ? -1 : gk >> 16;
		}
	}
	} // chunk loop
}

// ---- launcher: tasks listed in d_order[0..*d_count) must have 1 <= qlen <= 128*lpt
int launch_extend_lanex(bmh_ctx *ctx, int lpt, const uint8_t *d_pool, const bmh_ext_task_t *d_tasks, int64_t n,
                        bmh_ext_result_t *d_res, const uint32_t *d_order, const uint32_t *d_count, int min_count)
{
	if (n <= 0) return BMH_OK;
	const int tpw = 64 / lpt;
	long long grid = (n + tpw - 1) / tpw; // n = the dispatcher's upper bound of the bin size; the kernel strides
	const long long cap = ext_resident_waves(ctx, 2) * ctx->ext_grid_mult;
	if (grid > cap) grid = cap;
	const bool sym = ctx->dev.o_del + ctx->dev.e_del == ctx->dev.o_ins + ctx->dev.e_ins;
#define BMH_LAUNCH_LANEX(LL, SS)                                                                                        \
	hipLaunchKernelGGL((extend_lanex_kernel<LL, SS>), dim3((unsigned)grid), dim3(64), 0, ctx->stream, d_pool, d_tasks, d_order, d_count, \
	                   (long long)n, d_res, ctx->dev, ctx->d_err, min_count)
	if (lpt == 2 && sym) BMH_LAUNCH_LANEX(2, true);
	else if (lpt == 2) BMH_LAUNCH_LANEX(2, false);
	else if (lpt == 4 && sym) BMH_LAUNCH_LANEX(4, true);
	else if (lpt == 4) BMH_LAUNCH_LANEX(4, false);
#undef BMH_LAUNCH_LANEX
	else return BMH_E_ARG;
	BMH_HIP(ctx, hipGetLastError());
	return BMH_OK;
}

} // namespace bmh
