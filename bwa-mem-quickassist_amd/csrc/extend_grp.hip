// extend_grp.hip -- four seed extensions per wave64: one 16-lane DPP row per task.
//
// Same results as ksw_extend2 (reference bwa-0.7.8/ksw.c:379-476) and as the other kernels here.
// Motivation (profiles/r01_*): a 150 bp read flank has ~42 live cells per DP row, so a whole wave per
// task leaves most lanes idle while every row still pays the full price of the cross-lane scan, the
// row reduction and the bookkeeping.  Here a task owns ONE DPP row (16 lanes); each lane owns NV
// consecutive query columns (qlen <= 16*NV), right-aligned so that column qlen-1 is always the last
// column of lane 15.  One instruction stream therefore advances FOUR tasks by one DP row, and the
// cross-lane steps shrink to the 16-lane DPP forms (row_shr / quad_perm / row_mirror):
//   * F(i,j+1)=max(F(i,j)-e_ins, H(i,j)-o_ins-e_ins): sequential inside a lane, 5 DPP steps across
//     the 16 lanes (exclusive max-plus scan of lane totals);
//   * row max + right-most column, and the two "nearest zero" searches of the live-interval update
//     (ksw.c:463-466), are 4-step butterfly all-reduces that leave the result in every lane, so all
//     per-task bookkeeping (beg, end, best, its position, max_off, first-column value) is kept
//     replicated per lane and updated with VALU selects -- nothing per-task lives in SGPRs;
//   * groups are independent: when a task ends (m==0, z-drop, last row) its group writes the result
//     and immediately picks up its next task (static stride over the bin list) while the other three
//     keep going -- no sorting by length is needed;
//   * the row's target base comes from a per-group byte strip staged in LDS (1 ds_read_u8 per row),
//     the substitution score is one v_perm_b32 on the lane's biased profile bytes.
#include "bmh_ctx.h"
#include "bmh_device.h"

namespace bmh {

constexpr int kGrpTcap = 1024; // longest target a group can stage (LDS bytes per group)

constexpr int DPP_QP_1032 = 0xB1, DPP_QP_2301 = 0x4E, DPP_ROW_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140;

// max over the 16 lanes of a DPP row, result in every lane of the row
__device__ __forceinline__ int row_allmax(int v)
{
	constexpr int I = INT32_MIN;
	v = max(dpp<DPP_QP_1032>(I, v), v);
	v = max(dpp<DPP_QP_2301>(I, v), v);
	v = max(dpp<DPP_ROW_HALF_MIRROR>(I, v), v);
	v = max(dpp<DPP_ROW_MIRROR>(I, v), v);
	return v;
}

// exclusive prefix max over the lanes of a DPP row (lane 0 of the row gets `ident`)
__device__ __forceinline__ int row_exscan_max(int v, int ident)
{
	constexpr int I = INT32_MIN;
	v = dpp<DPP_ROW_SHR1>(ident, v);
	v = max(dpp<DPP_ROW_SHR1>(I, v), v);
	v = max(dpp<DPP_ROW_SHR2>(I, v), v);
	v = max(dpp<DPP_ROW_SHR4>(I, v), v);
	v = max(dpp<DPP_ROW_SHR8>(I, v), v);
	return v;
}

__device__ __forceinline__ int bfi(int mask, int a, int b) { return (a & mask) | (b & ~mask); }

template <int NV>
__global__ __launch_bounds__(64) void extend_grp_kernel(const uint8_t *__restrict__ pool,
                                                        const bmh_ext_task_t *__restrict__ tasks,
                                                        const uint32_t *__restrict__ order,
                                                        const uint32_t *__restrict__ count, long long n,
                                                        bmh_ext_result_t *__restrict__ out, DevParams P,
                                                        int *__restrict__ err_flag)
{
	constexpr int NC = 16 * NV; // columns per task slot
	constexpr int NEG = INT32_MIN / 2;
	__shared__ int smat[32];
	__shared__ uint8_t tstage[4][kGrpTcap];
	const int lane = threadIdx.x, l16 = lane & 15, g = lane >> 4;
	const int p0 = l16 * NV; // first (right-aligned) column position of this lane
	const int oe_del = P.o_del + P.e_del, oe_ins = P.o_ins + P.e_ins;
	const int e_del = P.e_del, e_ins = P.e_ins, bias = P.bias;

	if (lane < 25) smat[lane] = mat_at(P, lane) + bias;
	if (count) n = *count;

	const long long ngroups = (long long)gridDim.x * 4;
	long long next = (long long)blockIdx.x * 4 + g; // this group's next position in the bin list

	// per-task state, identical in the 16 lanes of a group
	uint32_t idx = 0;
	int qlen = 0, tlen = 0, h0 = 0, w = 0, off = 0;
	int i = 0, begp = 0, endp = 0, best = 0, bi = -1, bjp = -1, maxoff = 0, raw = 0, left = 0;
	int gk = -1; // (h<<16 | row) of column qlen-1, meaningful in lane 15 of the group
	bool alive = false;
	// per-column state
	int Hs[NV], E[NV], plo[NV], phi[NV];
#pragma unroll
	for (int k = 0; k < NV; ++k) Hs[k] = E[k] = plo[k] = phi[k] = 0;

	for (;;) {
		// ---- refill: every idle group takes its next task
#ifdef BMH_GRP_SYNC_REFILL
		const bool need = __builtin_amdgcn_ballot_w64(alive) == 0 && next < n; // experiment: refill only when all four are idle
#else
		const bool need = !alive && next < n;
#endif
		if (__builtin_amdgcn_ballot_w64(need)) {
			if (need) {
				idx = order ? order[next] : (uint32_t)next;
				next += ngroups;
				const uint4 *tp = (const uint4 *)(tasks + idx);
				const uint4 ta = tp[0], tb = tp[1];
				const uint64_t q_off = (uint64_t)ta.y << 32 | ta.x, t_off = (uint64_t)ta.w << 32 | ta.z;
				qlen = (int)(tb.x & 0xffff), tlen = (int)(tb.x >> 16);
				h0 = max((int)tb.y, 0); // ksw.c:384
				w = (int)(int16_t)(tb.z & 0xffff);
				const int end_bonus = (int)(int16_t)(tb.z >> 16);
				const bool qrev = tb.w & BMH_F_QREV, trev = tb.w & BMH_F_TREV, tpac = tb.w & BMH_F_TPAC;
				if (qlen > NC || qlen < 1 || tlen > kGrpTcap || h0 + qlen * P.max_mat > kScoreLimit) {
					if (l16 == 0) {
						int *p = (int *)(out + idx);
						p[0] = INT32_MIN, p[1] = p[2] = p[3] = p[4] = p[5] = 0;
						atomicExch(err_flag, BMH_E_RANGE);
					}
				} else {
					off = NC - qlen;
#pragma unroll
					for (int k = 0; k < NV; ++k) { // ksw.c:389-396
						const int j = p0 + k - off;
						int qb = 4;
						if (j >= 0) qb = seq_base(pool, q_off, j, qrev);
						plo[k] = smat[qb] | smat[5 + qb] << 8 | smat[10 + qb] << 16 | smat[15 + qb] << 24;
						phi[k] = smat[20 + qb];
						Hs[k] = j < 0 ? 0 : (j == 0 ? h0 : max(0, h0 - P.o_ins - j * e_ins));
						E[k] = 0;
					}
					for (int r = l16 * 4; r < tlen; r += 64) { // stage the target strip, 4 bases per lane and turn
#pragma unroll
						for (int c = 0; c < 4; ++c)
							if (r + c < tlen) tstage[g][r + c] = (uint8_t)tgt_base(pool, P, t_off, r + c, trev, tpac);
					}
					w = min(w, max(1, band_cap(qlen, P.max_mat, end_bonus, P.o_ins, e_ins))); // ksw.c:398-406
					w = min(w, max(1, band_cap(qlen, P.max_mat, end_bonus, P.o_del, e_del)));
					i = 0, begp = off, endp = NC, best = h0, bi = -1, bjp = off - 1, maxoff = 0;
					raw = h0 - P.o_del, left = 0, gk = -1;
					alive = tlen > 0;
					if (!alive && l16 == 0) { // no rows at all: ksw.c:411 never enters
						int *p = (int *)(out + idx);
						p[0] = h0, p[1] = 0, p[2] = 0, p[3] = 0, p[4] = -1, p[5] = 0;
					}
				}
			}
		}
		if (__builtin_amdgcn_ballot_w64(alive) == 0) {
			if (__builtin_amdgcn_ballot_w64(next < n) == 0) break;
			continue;
		}

		// ---- one DP row for the (up to) four live tasks
		const int tb = tstage[g][min(i, kGrpTcap - 1)];
		const int sel = 0x0c0c0c00 | tb;
		begp = max(begp, i - w + off);          // ksw.c:418-420
		endp = min(endp, i + w + 1 + off);      // (endp <= NC already covers the qlen clamp)
		raw -= e_del;
		left = max(raw, 0);                     // ksw.c:415-416
		const int rowb = begp, rowe = endp; // this row's interval, kept for the exit path
		const int lo = min(max(begp - p0, 0), NV), hi = min(max(endp - p0, 0), NV);
		const int am = hi > lo ? (int)(((1u << (hi - lo)) - 1u) << lo) : 0; // live slots of this lane
		const int kin = begp - p0;                                         // slot that holds column `beg`
		const int injm = (unsigned)kin < (unsigned)NV ? 1 << kin : 0;

		int hh[NV], Pm[NV];
#pragma unroll
		for (int k = 0; k < NV; ++k) {
			const int actv = (am << (31 - k)) >> 31; // all-ones if slot k is inside [beg,end)
			const int sc = (int)__builtin_amdgcn_perm((unsigned)phi[k], (unsigned)plo[k], (unsigned)sel);
			hh[k] = max(Hs[k] + sc - bias, E[k]);                              // ksw.c:430-431
			const int G = bfi(actv, max(hh[k] + (k * e_ins - oe_ins), k * e_ins), NEG); // max(hh-oe_ins,0)+k*e_ins
			Pm[k] = k ? max(Pm[k - 1], G) : G;
		}
		const int laneoff = p0 * e_ins;
		const int X = row_exscan_max(Pm[NV - 1] + laneoff, NEG) - laneoff; // best contribution of the lanes to the left
		int kj = -1, hm0[NV];
#pragma unroll
		for (int k = 0; k < NV; ++k) {
			const int actv = (am << (31 - k)) >> 31;
			const int pe = k ? max(X, Pm[k - 1]) : X;
			const int h = max(hh[k], pe + (e_ins - k * e_ins));               // max(hh, F), ksw.c:432
			const int en = max(max(E[k] - e_del, h - oe_del), 0);             // ksw.c:436-439
			E[k] = en & actv;
			const int ha = bfi(actv, h, -1);
			kj = max(kj, ha << 16 | (p0 + k));
			hm0[k] = max(ha, 0);
			if (k == NV - 1) gk = max(gk, ha << 16 | i); // lane 15: column qlen-1 (ksw.c:447-450, ties -> later row)
		}
		const int rkey = row_allmax(kj); // row max and its right-most column, ksw.c:434-435

		// next row's shifted H = eh[].h: first-column value at column beg (ksw.c:429), h(i,j-1) to its right, 0 outside
		int nzb = 0;
		{
			const int carry = dpp<DPP_ROW_SHR1>(0, hm0[NV - 1]);
#pragma unroll
			for (int k = NV - 1; k >= 0; --k) {
				const int src = k ? hm0[k - 1] : carry;
				const int injv = (injm << (31 - k)) >> 31;
				Hs[k] = bfi(injv, left, src);
				nzb |= min((unsigned)Hs[k], 1u) << k;
			}
		}
		const int zb = ~nzb & ((1 << NV) - 1); // zero map of this lane's columns

		const int m = rkey >> 16, mjp = rkey & 0xffff;
		const bool stop0 = rkey < 0x10000;    // m == 0 or empty row, ksw.c:451
		const bool upd = m > best;            // ksw.c:452-454
		const int dd = (i - bi) - (mjp - bjp);
		const int pen = max(dd * e_del, -dd * e_ins);
		const bool zd = !upd && P.zdrop > 0 && best - m - pen > P.zdrop; // ksw.c:455-461
		best = upd ? m : best;
		bi = upd ? i : bi;
		bjp = upd ? mjp : bjp;
		maxoff = upd ? max(maxoff, abs(mjp - off - i)) : maxoff;

		// live-interval update, ksw.c:463-466: nearest zero of eh[].h at or left of mj, and at or right of mj+2
		const int rl = mjp - p0;                                     // slots <= rl are at or left of mj
		const int zl = rl < 0 ? 0 : (rl >= NV - 1 ? zb : zb & ((2 << rl) - 1));
		const int candl = zl ? p0 + 31 - __builtin_clz(zl) : -1;
		const int rr = mjp + 2 - p0;                                 // slots >= rr are at or right of mj+2
		const int zr = rr >= NV ? 0 : (rr <= 0 ? zb : zb & (-1 << rr));
		const int candr = zr ? -(p0 + __builtin_ctz(zr)) : -NC;       // negated so that max picks the smallest
		begp = row_allmax(candl) + 1;
		endp = -row_allmax(candr);

		const bool last = i + 1 >= tlen;
		const bool done = alive && (stop0 || zd || last);
		if (__builtin_amdgcn_ballot_w64(done)) {
			if (done) { // results, ksw.c:470-475
				alive = false;
				if (l16 == 0) {
					int *p = (int *)(out + idx);
					p[0] = best, p[1] = bjp - off + 1, p[2] = bi + 1, p[5] = maxoff;
				}
				if (l16 == 15) {
					// an EMPTY row whose loop variable equals qlen still feeds gscore with the first-column value (ksw.c:447-450)
					if (rowe <= rowb && rowb == NC) gk = max(gk, left << 16 | i);
					int *p = (int *)(out + idx);
					p[3] = gk < 0 ? 0 : (gk & 0xffff) + 1;
					p[4] = gk < 0 ? -1 : gk >> 16;
				}
			}
		}
		++i;
	}
}

// ---- launcher: tasks listed in d_order[0..*d_count) must have 1 <= qlen <= 16*nv and tlen <= kGrpTcap
int launch_extend_grp(bmh_ctx *ctx, int nv, const uint8_t *d_pool, const bmh_ext_task_t *d_tasks, int64_t n,
                      bmh_ext_result_t *d_res, const uint32_t *d_order, const uint32_t *d_count)
{
	if (n <= 0) return BMH_OK;
	// persistent launch: exactly the resident waves (each group then walks ~n/(4*grid) tasks, which
	// averages out their lengths); more blocks than that would only start late and lengthen the tail
	static int occ[9] = {0};
	if (!occ[nv]) {
		int b = 0;
		hipError_t e = nv == 2 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, extend_grp_kernel<2>, 64, 0)
		             : nv == 4 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, extend_grp_kernel<4>, 64, 0)
		                       : hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, extend_grp_kernel<8>, 64, 0);
		occ[nv] = e == hipSuccess && b > 0 ? b : 8;
	}
	int ncu = 256;
	(void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device);
	long long grid = (n + 3) / 4;
	const long long resident = (long long)occ[nv] * ncu * ctx->grid_mult;
	if (grid > resident) grid = resident;
#define BMH_LAUNCH_GRP(NV)                                                                                           \
	hipLaunchKernelGGL(extend_grp_kernel<NV>, dim3((unsigned)grid), dim3(64), 0, ctx->stream, d_pool, d_tasks, d_order, \
	                   d_count, (long long)n, d_res, ctx->dev, ctx->d_err)
	switch (nv) {
	case 2: BMH_LAUNCH_GRP(2); break;
	case 4: BMH_LAUNCH_GRP(4); break;
	case 8: BMH_LAUNCH_GRP(8); break;
	default: return BMH_E_ARG;
	}
#undef BMH_LAUNCH_GRP
	BMH_HIP(ctx, hipGetLastError());
	return BMH_OK;
}

} // namespace bmh
