// extend_reg.hip -- register-resident fast path of the seed-extension kernel (qlen <= 64*NS).
//
// Same algorithm and results as extend_lds.hip / ksw_extend2 (reference bwa-0.7.8/ksw.c:379-476),
// re-shaped around what the rocprof counters of the first kernel showed: that kernel was bound by
// SCALAR issue (103 SALU + 27 branch instructions per DP row; one scalar unit per CU), not by VALU,
// LDS or HBM.  Here
//   * lane l of slot s owns query column 64*s+l for the whole task; H (shifted), E, the query
//     profile and the per-column constants stay in VGPRs -- no LDS traffic in the row loop;
//   * all per-row bookkeeping that the reference keeps in scalars (beg, end, best, its position,
//     max_off, the first-column value) is kept REPLICATED in VGPRs and updated with VALU selects,
//     so a row costs a handful of SALU instructions instead of ~130;
//   * the live-interval update (ksw.c:463-466) is two wave min-reductions over "distance to the
//     nearest zero of H" instead of scalar bit scans; lanes outside [beg,end] are rewritten with
//     H = E = 0 every row, which makes "outside the interval" and "H == 0" the same test;
//   * gscore / max_ie (ksw.c:447-450) are tracked by the lane that owns column qlen-1 as a
//     (h<<16 | row) key, ties -> later row;
//   * the target base of the row comes from a VGPR that holds one v_perm selector per lane and is
//     rotated by one lane per row (wave_rol:1) -- no scalar index arithmetic;
//   * the substitution score is one v_perm_b32 on the lane's 5 biased profile bytes.
// Rows still run in order (the adaptive band of row i+1 depends on row i).
#include "bmh_ctx.h"
#include "bmh_device.h"

namespace bmh {

constexpr int DPP_WAVE_ROL1 = 0x134;

// uniform value -> VGPR copy the compiler must treat as per-lane (keeps the arithmetic on the VALU)
__device__ __forceinline__ int vg(int s)
{
	int v;
	asm("v_mov_b32 %0, %1" : "=v"(v) : "s"(s));
	return v;
}

__device__ __forceinline__ unsigned wave_scan_min_u32(unsigned v)
{
	constexpr int I = -1; // identity of unsigned min
	v = min((unsigned)dpp<DPP_ROW_SHR1>(I, (int)v), v);
	v = min((unsigned)dpp<DPP_ROW_SHR2>(I, (int)v), v);
	v = min((unsigned)dpp<DPP_ROW_SHR4>(I, (int)v), v);
	v = min((unsigned)dpp<DPP_ROW_SHR8>(I, (int)v), v);
	v = min((unsigned)dpp<DPP_ROW_BCAST15, 0xa>(I, (int)v), v);
	v = min((unsigned)dpp<DPP_ROW_BCAST31, 0xc>(I, (int)v), v);
	return v;
}

template <int NS>
__global__ __launch_bounds__(64) void extend_reg_kernel(const uint8_t *__restrict__ pool,
                                                        const bmh_ext_task_t *__restrict__ tasks,
                                                        const uint32_t *__restrict__ order,
                                                        const uint32_t *__restrict__ count, long long n,
                                                        bmh_ext_result_t *__restrict__ out, DevParams P,
                                                        int *__restrict__ err_flag, int max_count)
{
	__shared__ int smat[32]; // biased scores, read once per task while building the lane profiles
	const int lane = threadIdx.x;
	const int oe_del = P.o_del + P.e_del, oe_ins = P.o_ins + P.e_ins;
	const int e_del = P.e_del, e_ins = P.e_ins;
	const int bias = P.bias;

	if (lane < 25) smat[lane] = mat_at(P, lane) + bias;

	if (count) n = *count; // bin size produced on the device by the dispatcher
	if (max_count > 0 && n >= max_count) return; // a large bin is served by the lanes-per-task kernel instead
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

		if (qlen > 64 * NS || qlen < 1 || h0 + qlen * P.max_mat > kScoreLimit) {
			if (lane == 0) {
				int *p = (int *)(out + idx);
				p[0] = INT32_MIN, p[1] = p[2] = p[3] = p[4] = p[5] = 0;
				atomicExch(err_flag, BMH_E_RANGE);
			}
			continue;
		}

		// ---- per-lane column state (ksw.c:389-396)
		int jv[NS], Hs[NS], E[NS], plo[NS], phi[NS], c1[NS], c2[NS], c3n[NS], gk[NS];
#pragma unroll
		for (int s = 0; s < NS; ++s) {
			const int j = 64 * s + lane;
			jv[s] = j;
			int qb = 4;
			if (j < qlen) qb = seq_base(pool, q_off, j, qrev);
			plo[s] = smat[qb] | smat[5 + qb] << 8 | smat[10 + qb] << 16 | smat[15 + qb] << 24;
			phi[s] = smat[20 + qb];
			Hs[s] = j == 0 ? h0 : (j <= qlen ? max(0, h0 - P.o_ins - j * e_ins) : 0);
			E[s] = 0;
			c2[s] = j * e_ins;
			c1[s] = c2[s] - oe_ins;
			c3n[s] = e_ins - c2[s];
			gk[s] = -1;
		}

		// band clamp, ksw.c:398-406
		w = min(w, max(1, band_cap(qlen, P.max_mat, end_bonus, P.o_ins, e_ins)));
		w = min(w, max(1, band_cap(qlen, P.max_mat, end_bonus, P.o_del, e_del)));
		const int w1 = w + 1;

		// ---- row state.  beg/end/m/mj are wave-uniform scalars (SGPRs); best, its position, max_off and the
		// first-column value are kept REPLICATED in VGPRs and updated with VALU ops -- the split keeps both the
		// vector and the scalar issue ports busy instead of saturating one of them.
		int beg = 0, end = qlen;
		int v_best = vg(h0), v_bi = vg(-1), v_bj = vg(-1), v_maxoff = vg(0);
		int v_raw = vg(h0 - P.o_del); // h0 - o_del - e_del*(i+1) after the row's decrement
		int v_left = vg(0);
		int i = 0;
		bool stopped = false;

		for (int ib = 0; ib < tlen && !stopped; ib += 64) {
			int tv; // lane l: v_perm selector of target row ib+l
			{
				const int r = ib + lane;
				int tbse = 0;
				if (r < tlen) tbse = tgt_base(pool, P, t_off, r, trev, tpac);
				tv = 0x0c0c0c00 | tbse;
			}
			const int nrow = min(64, tlen - ib);
			for (int k = 0; k < nrow; ++k, ++i) {
				const int sel = __builtin_amdgcn_readfirstlane(tv);
				tv = dpp<DPP_WAVE_ROL1>(tv, tv);
				// interval of this row, ksw.c:418-420
				beg = max(beg, i - w);
				end = min(min(end, i + w1), qlen);
				const unsigned wd = (unsigned)max(end - beg, 0);
				v_raw -= e_del;
				v_left = max(v_raw, 0); // first-column value, ksw.c:415-416

				int kj = -1, hm0[NS];
				int cin = INT32_MIN / 2;
#pragma unroll
				for (int s = 0; s < NS; ++s) {
					const bool act = (unsigned)(jv[s] - beg) < wd;
					const int sc = (int)__builtin_amdgcn_perm((unsigned)phi[s], (unsigned)plo[s], (unsigned)sel);
					const int hh = max(Hs[s] + sc - bias, E[s]); // max(M+S, E), ksw.c:430-431
					// max(hh-oe_ins,0) + j*e_ins; lanes outside [beg,end) must not feed F (a lane that was `beg` one
					// row earlier still holds the injected first-column value)
					const int G = act ? max(hh + c1[s], c2[s]) : INT32_MIN / 2;
					const int pm = wave_scan_max(G);
					int pex = wave_shr1(pm, INT32_MIN / 2);
					if (s > 0) pex = max(pex, cin);
					if (s + 1 < NS) cin = max(cin, __builtin_amdgcn_readlane(pm, 63));
					const int h = max(hh, pex + c3n[s]); // max(hh, F), ksw.c:432
					const int en = max(max(E[s] - e_del, h - oe_del), 0); // ksw.c:436-439
					E[s] = act ? en : 0;
					const int ha = act ? h : -1; // -1 outside the interval: its keys are negative and lose every max
					kj = max(kj, ha << 16 | jv[s]);
					gk[s] = max(gk[s], ha << 16 | i); // ksw.c:447-450 for the lane owning column qlen-1 (ties -> later row)
					hm0[s] = max(ha, 0);
				}
				const int rkey = wave_reduce_max(kj); // row max and its right-most column, ksw.c:434-435
				// next row's shifted H = eh[].h: first-column value at j==beg (ksw.c:429), h(i,j-1) right of it, 0 outside
				unsigned long long zmask[NS];
#pragma unroll
				for (int s = 0; s < NS; ++s) {
					const int prev_last = s > 0 ? __builtin_amdgcn_readlane(hm0[s - 1], 63) : 0;
					const int sh = dpp<DPP_WAVE_SHR1>(vg(prev_last), hm0[s]);
					Hs[s] = jv[s] == beg ? v_left : sh;
					zmask[s] = __ballot(Hs[s] == 0);
				}
				if (rkey < 0x10000) { // m == 0 (or empty row), ksw.c:451
					stopped = true;
					break;
				}
				const int m = rkey >> 16, mj = rkey & 0xffff;
				const bool upd = m > v_best; // ksw.c:452-454
				if (__builtin_amdgcn_ballot_w64(upd) == 0 && P.zdrop > 0) { // ksw.c:455-461
					const int dd = (i - v_bi) - (mj - v_bj);
					const int pen = max(dd * e_del, -dd * e_ins); // dd>0: dd*e_del ; dd<=0: -dd*e_ins (both e >= 1)
					if (__builtin_amdgcn_ballot_w64(v_best - m - pen > P.zdrop) != 0) {
						stopped = true;
						break;
					}
				}
				v_best = max(v_best, m);
				v_bi = upd ? i : v_bi;
				v_bj = upd ? mj : v_bj;
				v_maxoff = upd ? max(v_maxoff, abs(mj - i)) : v_maxoff;
				// live-interval update, ksw.c:463-466, on the zero masks of eh[].h (lanes outside [beg,end] hold 0):
				// nearest zero at or left of mj -> beg ; nearest zero at or right of mj+2 -> end
				int nb = 0, ne = qlen;
#pragma unroll
				for (int s = NS - 1; s >= 0; --s) {
					const int hi = mj - 64 * s; // highest bit of this word that may be looked at
					if (hi >= 0) {
						const unsigned long long z = hi >= 63 ? zmask[s] : zmask[s] & ((2ull << hi) - 1);
						if (z) {
							nb = 64 * s + 64 - __builtin_clzll(z);
							break;
						}
					}
				}
#pragma unroll
				for (int s = 0; s < NS; ++s) {
					const int lo = mj + 2 - 64 * s; // lowest bit of this word that may be looked at
					if (lo < 64) {
						const unsigned long long z = lo <= 0 ? zmask[s] : zmask[s] & (~0ull << lo);
						if (z) {
							ne = min(qlen, 64 * s + __builtin_ctzll(z));
							break;
						}
					}
				}
				beg = nb, end = ne;
			}
		}

		// ---- results (ksw.c:470-475)
		int gkey = -1;
#pragma unroll
		for (int s = 0; s < NS; ++s)
			if ((qlen - 1) >> 6 == s) gkey = __builtin_amdgcn_readlane(gk[s], (qlen - 1) & 63);
		int gscore = gkey < 0 ? -1 : gkey >> 16;
		int gi = gkey < 0 ? -1 : gkey & 0xffff;
		if (stopped && beg >= end && beg == qlen) { // empty row whose loop variable equals qlen, ksw.c:447
			const int left = __builtin_amdgcn_readfirstlane(v_left);
			if (!(gscore > left)) gi = i;
			gscore = max(gscore, left);
		}
		if (lane == 0) {
			int *p = (int *)(out + idx);
			p[0] = v_best, p[1] = v_bj + 1, p[2] = v_bi + 1, p[3] = gi + 1, p[4] = gscore, p[5] = v_maxoff;
		}
	}
}

// ---- launcher: every task listed in d_order[0..*d_count) (or 0..n) must have 1 <= qlen <= 64*ns
int launch_extend_reg(bmh_ctx *ctx, int ns, const uint8_t *d_pool, const bmh_ext_task_t *d_tasks, int64_t n,
                      bmh_ext_result_t *d_res, const uint32_t *d_order, const uint32_t *d_count, int max_count, long long grid_cap)
{
	if (n <= 0) return BMH_OK;
	long long grid = n < kPersistentGrid ? n : kPersistentGrid;
	if (grid_cap > 0 && grid > grid_cap) grid = grid_cap;
#define BMH_LAUNCH_REG(NS)                                                                                           \
	hipLaunchKernelGGL(extend_reg_kernel<NS>, dim3((unsigned)grid), dim3(64), 0, ctx->stream, d_pool, d_tasks, d_order, \
	                   d_count, (long long)n, d_res, ctx->dev, ctx->d_err, max_count)
	switch (ns) {
	case 1: BMH_LAUNCH_REG(1); break;
	case 2: BMH_LAUNCH_REG(2); break;
	case 4: BMH_LAUNCH_REG(4); break;
	default: return BMH_E_ARG;
	}
#undef BMH_LAUNCH_REG
	BMH_HIP(ctx, hipGetLastError());
	return BMH_OK;
}

} // namespace bmh
