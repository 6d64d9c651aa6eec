// extend_lane.hip -- 64 seed extensions per wave64: one LANE per task, DP row state in registers.
//
// Same results as ksw_extend2 (reference bwa-0.7.8/ksw.c:379-476) and as the other kernels here.
//
// Why (profiles/r01_*): a wave-instruction costs the same 4 cycles whether it advances one
// task by 64 cells or 64 tasks by one cell each.  With one task per wave (or per 16-lane row) every DP row
// pays ~60-100 vector instructions of cross-lane scan / reduction / bookkeeping for only ~40 live cells.
// Here each lane runs the reference's scalar recurrence for ITS OWN task:
//   * the C = 32/64/128 columns of the row (shifted H and E packed u16|u16 in ONE VGPR per column; the query
//     as v_perm selectors, 4 per VGPR) live in registers; the column loop is fully unrolled, so all register
//     indices are static; F and the left neighbour are carried sequentially through the unrolled columns --
//     no cross-lane operation is left in the kernel;
//   * the query is RIGHT-ALIGNED in the C columns (column qlen-1 is always position C-1), so the gscore column
//     is a compile-time position;
//   * the live interval [beg,end) of a lane is a per-lane bit mask; columns outside it run predicated (their
//     F / left-neighbour outputs are forced to 0 / first-column value, which is exactly what the first live
//     column must see, ksw.c:412-416,429); 8-column blocks that no lane of the wave needs are skipped with one
//     wave-uniform branch;
//   * the interval update (ksw.c:463-466) needs no second pass: while walking the columns a lane tracks the
//     last zero seen, the last zero left of the running maximum and the first zero right of it;
//   * all tasks of a wave start together at row 0, so the row index and the loop are wave-uniform; a lane that
//     finishes (m==0, z-drop, last row) writes its result and idles until the wave is done.  The dispatcher
//     hands this kernel tasks SORTED by expected row count, so lanes of a wave finish together.
#include <algorithm>
#include <type_traits>

#include "bmh_ctx.h"
#include "bmh_device.h"

namespace bmh {

#ifndef BMH_LANE_INTERIOR
#define BMH_LANE_INTERIOR 0 /* measured on the 20 M-read step: 81.7 ms with the unmasked body for interior blocks (27 spilled VGPRs at C = 64), 82.3 ms without */
#endif
constexpr bool kLaneInterior = BMH_LANE_INTERIOR; // unmasked body for blocks interior to every live lane's interval
// waves per SIMD the register allocator must leave room for (2nd launch-bounds argument)
#ifndef BMH_LANE_WAVES
#define BMH_LANE_WAVES(C) ((C) <= 32 ? 5 : (C) <= 64 ? 4 : (C) <= 96 ? 3 : 2)
#endif

// per-bit select: mask ? a : b  (one v_bitop3_b32 on gfx950)
__device__ __forceinline__ int bfi2(int mask, int a, int b) { return __builtin_amdgcn_bitop3_b32(mask, a, b, 0xca); }

// SYM: o_del+e_del == o_ins+e_ins (bwa's default) -> H-oe is computed once per cell for both gap states
// LOOP: the block walks the bin with a grid stride from chunk `chunk0` on (persistent grid; costs the register allocator
// some per-row scratch traffic at C = 128); !LOOP: one chunk per block, chunk = chunk0 + blockIdx.x.
template <int C, bool SYM, bool LOOP>
__global__ __launch_bounds__(64, BMH_LANE_WAVES(C)) void extend_lane_kernel(const uint8_t *__restrict__ pool,
                                                         const bmh_ext_task_t *__restrict__ tasks,
                                                         const uint32_t *__restrict__ order,
                                                         const uint32_t *__restrict__ count, long long n,
                                                         bmh_ext_result_t *__restrict__ out, DevParams P,
                                                         int *__restrict__ err_flag, long long chunk0,
                                                         const uint32_t *__restrict__ skip)
{
	constexpr int NW = C / 32, NQ = C / 4, NB = C / 8;
	constexpr int INF = 0x7fff;
	__shared__ uint2 srow[8]; // srow[t] = the 5 signed score bytes mat[t*5 .. t*5+4]
	// A task's sequences pass through LDS.  The query selectors are fetched in a ROLLED loop into `stage` and read back into QS[]:
	// unrolled, the C byte loads are all in flight at once and their 64-bit addresses and destinations (3 VGPRs each) set the kernel's
	// register count.  The target is then streamed through the same area kStreamRows rows at a time, one byte per row at [row][lane]:
	// loaded inside the row loop (under an exec mask, so that the compiler cannot count it) every row paid an `s_waitcnt vmcnt(0)` right
	// behind the load -- a full memory round trip per DP row.
	constexpr int kStreamRows = 64;
	__shared__ uint32_t stage[(C > kStreamRows ? C : kStreamRows) * 16]; // C/4 dwords x 64 lanes, or kStreamRows x 64 bytes
#ifdef BMH_LANE_LDS_SEL
	__shared__ uint8_t qsel[C * 64]; // query code of column p of lane l at [p*64+l]: one conflict-free ds_read_u8 per cell
#endif
	const int lane = threadIdx.x;
	const int oe_del = P.o_del + P.e_del, oe_ins = P.o_ins + P.e_ins;
	const int e_del = P.e_del, e_ins = P.e_ins;

	if (lane < 5) {
		uint32_t lo = 0;
		for (int q = 0; q < 4; ++q) lo |= (uint32_t)(uint8_t)mat_at(P, lane * 5 + q) << (8 * q);
		srow[lane] = make_uint2(lo, (uint32_t)(uint8_t)mat_at(P, lane * 5 + 4));
	}
	// The kernel's share of the (sorted) list: [0, count) without `skip`.  The 96-column instantiation takes the head of the 65-128 bin
	// (count = the number of its tasks with qlen <= 96, read from the sort's cursors), the 128-column one the rest: [skip, count) with
	// skip = that number.  The head is launched WITHOUT the strided pick-up launch -- an empty grid between two full ones still queues for
	// wave slots behind whatever runs on the other streams, 5-35 ms in the 100-300 bp shape -- so when the estimate `n` fell short of the
	// head (its one-chunk-per-block grid walks from the back and reached [skip - 64*ceil(n/64), skip)) the `rem` entries at the front
	// are this launch's to do as well: its positions [0, rem) are those, [rem, cnt) map to [skip, count).
	long long first = 0, rem = 0;
	if (skip) {
		first = (long long)*skip;
		rem = max(first - (((n + 63) >> 6) << 6), 0LL);
	}
	const long long cnt = (count ? (long long)*count : n) - first + rem;
	// persistent grid: a block walks the bin in chunks of 64 tasks with a grid stride (the launcher sizes the grid for
	// the machine, not for the batch, so an empty or small bin costs a few hundred waves instead of n/64)
	for (long long base = (chunk0 + (long long)blockIdx.x) * 64; base < cnt; base += LOOP ? (long long)gridDim.x * 64 : cnt) {
	const bool valid = base + lane < cnt;
	// the bin list is sorted ascending (short queries / few rows first); walk it from the back so the most
	// expensive waves are dispatched first and the cheap ones fill the tail
	const long long pos = cnt - 1 - (valid ? base + lane : base);
	const long long lp = pos < rem ? pos : pos + (first - rem);
	const uint32_t idx = order ? order[lp] : (uint32_t)lp;

	const uint4 *tp = (const uint4 *)(tasks + idx);
	const uint4 ta = tp[0], tb = tp[1];
	const uint64_t q_off = (uint64_t)ta.y << 32 | ta.x, t_off = (uint64_t)ta.w << 32 | ta.z;
	const int qlen = (int)(tb.x & 0xffff), tlen = (int)(tb.x >> 16);
	const int h0 = max((int)tb.y, 0); // ksw.c:384
	int w = (int)(int16_t)(tb.z & 0xffff);
	const int end_bonus = (int)(int16_t)(tb.z >> 16);
	const bool qrev = tb.w & BMH_F_QREV, trev = tb.w & BMH_F_TREV, tpac = tb.w & BMH_F_TPAC;
	const bool bad = qlen > C || qlen < 1 || h0 + qlen * P.max_mat > kScoreLimit;
	if (valid && bad) {
		int *p = (int *)(out + idx);
		p[0] = INT32_MIN, p[1] = p[2] = p[3] = p[4] = p[5] = 0;
		atomicExch(err_flag, BMH_E_RANGE);
	}
	const int off = C - min(max(qlen, 1), C);

	// ---- per-lane column state (ksw.c:389-396), query right-aligned
	int HE[C];
#ifdef BMH_LANE_LDS_SEL
	for (int p = 0; p < C; ++p) {
		const int j = p - off;
		int qb = 4;
		if (valid && !bad && j >= 0) qb = seq_base(pool, q_off, j, qrev);
		qsel[p * 64 + lane] = (uint8_t)qb;
	}
#else
	int QS[NQ];
#pragma unroll 2
	for (int v = 0; v < NQ; ++v) {
		int s = 0;
#pragma unroll
		for (int b = 0; b < 4; ++b) {
			const int j = 4 * v + b - off;
			int qb = 4;
			if (valid && !bad && j >= 0) qb = seq_base(pool, q_off, j, qrev);
			s |= qb << (8 * b);
		}
		stage[v * 64 + lane] = (uint32_t)s;
	}
#pragma unroll
	for (int v = 0; v < NQ; ++v) QS[v] = (int)stage[v * 64 + lane];
#endif
#pragma unroll
	for (int p = 0; p < C; ++p) {
		const int j = p - off;
		HE[p] = j < 0 ? 0 : (j == 0 ? h0 : max(0, h0 - P.o_ins - j * e_ins));
	}
	w = min(w, max(1, band_cap(qlen, P.max_mat, end_bonus, P.o_ins, e_ins))); // ksw.c:398-406
	w = min(w, max(1, band_cap(qlen, P.max_mat, end_bonus, P.o_del, e_del)));

	int begp = off, endp = C, best = h0, bi = -1, bjp = off - 1, maxoff = 0, raw = h0 - P.o_del, gk = -1;
	bool alive = valid && !bad && tlen > 0;
	if (valid && !bad && tlen == 0) { // no rows at all
		int *p = (int *)(out + idx);
		p[0] = h0, p[1] = 0, p[2] = 0, p[3] = 0, p[4] = -1, p[5] = 0;
	}
	uint8_t *strm = (uint8_t *)stage;

	for (int i = 0; __builtin_amdgcn_ballot_w64(alive) != 0; ++i) { // i is wave-uniform: all tasks started together
		if ((i & (kStreamRows - 1)) == 0) { // wave-uniform: target rows [i, i + kStreamRows) of every lane still running
#pragma unroll 4
			for (int r = 0; r < kStreamRows; ++r) {
				int tb = 0;
				if (alive && i + r < tlen) tb = tgt_base(pool, P, t_off, i + r, trev, tpac);
				strm[r * 64 + lane] = (uint8_t)tb;
			}
		}
		const int tcur = strm[(i & (kStreamRows - 1)) * 64 + lane];
		const uint2 row = srow[min(tcur, 4)];
		begp = max(begp, i - w + off);     // ksw.c:418-420
		endp = min(endp, i + w + 1 + off); // endp <= C covers the qlen clamp
		raw -= e_del;
		const int left = max(raw, 0); // first-column value, ksw.c:415-416
		const int lb = alive ? begp : C + 1, le = alive ? endp : C + 1;
		int am[NW];
#pragma unroll
		for (int v = 0; v < NW; ++v) {
			const int lo = min(max(lb - 32 * v, 0), 32), hi = min(max(le - 32 * v, 0), 32);
			am[v] = hi > lo ? (int)((0xffffffffu >> (32 - (hi - lo))) << lo) : 0;
		}
		int f = 0, hprev = left, kmax = -1, hlast = -1;
		int nz[NW]; // bit p set <=> h(i,p) != 0 (junk for columns outside the interval; masked with am[] below)
#pragma unroll
		for (int v = 0; v < NW; ++v) nz[v] = 0;
#pragma unroll
		for (int b = 0; b < NB; ++b) {
			// the block is needed by a lane iff [8b,8b+8) meets [beg,end]  (end itself receives eh[end])
			if (__builtin_amdgcn_ballot_w64(lb < 8 * b + 8 && le >= 8 * b) == 0) continue;
			// The cell in 16-bit instructions (bmh_device.h; issue classes: profiles/r03_valu_issue_classes.md).  Every quantity of the
			// recurrence is >= 0 except M = H(i-1,j-1) + S, which only enters a signed maximum with E >= 0.
			// An INTERIOR block -- all eight columns inside [beg,end) of every lane still running -- needs no activity masks: five of a
			// cell's nineteen instructions (finished lanes compute junk there; their results are already written).
			const bool edge = !kLaneInterior || __builtin_amdgcn_ballot_w64(alive && !(lb <= 8 * b && le >= 8 * b + 8)) != 0;
			auto cells = [&](auto MASKED) {
#pragma unroll
				for (int q4 = 0; q4 < 2; ++q4) {
#ifdef BMH_LANE_LDS_SEL
					int sc4 = 0;
#pragma unroll
					for (int c4 = 0; c4 < 4; ++c4) sc4 |= (int)(__builtin_amdgcn_perm(row.y, row.x, (unsigned)qsel[(8 * b + 4 * q4 + c4) * 64 + lane]) & 0xffu) << (8 * c4);
#else
					// the four substitution scores of columns 8b+4q4 .. +3 with one v_perm: the query codes are the selectors
					const int sc4 = (int)__builtin_amdgcn_perm(row.y, row.x, (unsigned)QS[2 * b + q4]);
#endif
#pragma unroll
					for (int c4 = 0; c4 < 4; ++c4) {
						const int p = 8 * b + 4 * q4 + c4;
						const int m = c4 == 0 ? add_score<0>(sc4, HE[p]) : c4 == 1 ? add_score<1>(sc4, HE[p]) : c4 == 2 ? add_score<2>(sc4, HE[p]) : add_score<3>(sc4, HE[p]);
						const int e = (int)((unsigned)HE[p] >> 16);
						const int h = max16(max16(m, e), f);                             // ksw.c:430-432
						const int t = subc16(h, oe_del);                                 // max(h - oe_del, 0)
						int en = maxu16(subc16(e, e_del), t);                            // ksw.c:436-439
						f = maxu16(subc16(f, e_ins), SYM ? t : subc16(h, oe_ins));       // ksw.c:441-444
						int ha = h, hn = h;
						if constexpr (decltype(MASKED)::value) {
							const int actv = (am[p / 32] << (31 - p % 32)) >> 31;
							en &= actv, f &= actv;
							ha = bfi2(actv, h, -1);                                      // -1 outside the interval
							hn = bfi2(actv, h, left);
						}
						HE[p] = en << 16 | hprev;                                        // eh[j] = {H(i,j-1), E(i+1,j)}, ksw.c:429,440
						kmax = max(kmax, ha << 16 | p);                                  // row max, ties -> larger j (ksw.c:434)
						nz[p / 32] |= nonzero16(h) << (p % 32);                         // only live columns are looked at later (& am)
						hprev = hn;
						if (p == C - 1) hlast = ha;
					}
				}
			};
			if (edge) cells(std::true_type{});
			else if constexpr (kLaneInterior) cells(std::false_type{});
		}
		// ---- row end, per lane
		gk = max(gk, hlast << 16 | i); // column qlen-1 live in this row: ksw.c:447-450 (ties -> later row)
		if (alive && le <= lb && lb == C) gk = max(gk, left << 16 | i); // empty row whose loop variable equals qlen
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
		// live-interval update, ksw.c:463-466, from the zero map of the live columns:
		//   eh[j].h == 0 for beg < j <= mj  <=>  h(i,j-1) == 0  -> last zero column left of mj, else eh[beg].h = first-column value
		//   eh[j].h == 0 for mj+2 <= j <= end <=> h(i,j-1) == 0  -> first zero column right of mj, else end+1
		int lzm = lb - 2 + (left == 0), fz = INF;
#pragma unroll
		for (int v = 0; v < NW; ++v) {
			const int zz = ~nz[v] & am[v];
			const int lim = mjp - 32 * v; // columns < mj
			const int zl = lim <= 0 ? 0 : (lim >= 32 ? zz : zz & ((1 << lim) - 1));
			if (zl) lzm = 32 * v + 31 - __builtin_clz(zl);
		}
#pragma unroll
		for (int v = NW - 1; v >= 0; --v) {
			const int zz = ~nz[v] & am[v];
			const int lo = mjp + 1 - 32 * v; // columns > mj
			const int zr = lo >= 32 ? 0 : (lo <= 0 ? zz : zz & (-1 << lo));
			if (zr) fz = 32 * v + __builtin_ctz(zr);
		}
		begp = lzm + 2;
		endp = min(fz == INF ? le + 1 : fz + 1, C);
		const bool done = alive && (stop0 || zd || i + 1 >= tlen);
		if (done) { // results, ksw.c:470-475
			alive = false;
			int *p = (int *)(out + idx);
			p[0] = best, p[1] = bjp - off + 1, p[2] = bi + 1;
			p[3] = gk < 0 ? 0 : (gk & 0xffff) + 1, p[4] = gk < 0 ? -1 : gk >> 16, p[5] = maxoff;
		}
	}
	} // chunk loop
}

// ---- launcher: tasks listed in d_order[0..*d_count) must have 1 <= qlen <= C.  `n` is the dispatcher's estimate of the
// bin size (exact or an upper bound without a hint).  persist: one strided launch with a capped grid.  Otherwise one
// chunk per block over the estimate, plus a small strided launch that picks up whatever lies beyond it.
int launch_extend_lane(bmh_ctx *ctx, int c, const uint8_t *d_pool, const bmh_ext_task_t *d_tasks, int64_t n,
                       bmh_ext_result_t *d_res, const uint32_t *d_order, const uint32_t *d_count, bool exact, const uint32_t *d_skip)
{
	if (n <= 0) return BMH_OK;
	const bool sym = ctx->dev.o_del + ctx->dev.e_del == ctx->dev.o_ins + ctx->dev.e_ins;
	const long long chunks = (n + 63) / 64;
	const long long cap = ext_resident_waves(ctx, BMH_LANE_WAVES(c)) * ctx->ext_grid_mult;
	// what the kernel takes for the reach of the head launch (see `rem` there): a persistent head grid reached everything
	const long long n_head = ctx->ext_persist && d_skip ? (1LL << 40) : (long long)n;
#define BMH_LAUNCH_LANE2(CC, SY, LP, GRID, C0)                                                                          \
	hipLaunchKernelGGL((extend_lane_kernel<CC, SY, LP>), dim3((unsigned)(GRID)), dim3(64), 0, ctx->stream, d_pool, d_tasks,  \
	                   d_order, d_count, n_head, d_res, ctx->dev, ctx->d_err, (long long)(C0), d_skip)
#define BMH_LAUNCH_LANE(CC, LP, GRID, C0)                                                                               \
	do {                                                                                                                \
		if (sym) BMH_LAUNCH_LANE2(CC, true, LP, GRID, C0);                                                              \
		else BMH_LAUNCH_LANE2(CC, false, LP, GRID, C0);                                                                 \
	} while (0)
#define BMH_LAUNCH_LANE_C(LP, GRID, C0)                                                                                 \
	do {                                                                                                                \
		switch (c) {                                                                                                    \
		case 32: BMH_LAUNCH_LANE(32, LP, GRID, C0); break;                                                              \
		case 64: BMH_LAUNCH_LANE(64, LP, GRID, C0); break;                                                              \
		case 96: BMH_LAUNCH_LANE(96, LP, GRID, C0); break;                                                              \
		case 128: BMH_LAUNCH_LANE(128, LP, GRID, C0); break;                                                            \
		default: return BMH_E_ARG;                                                                                      \
		}                                                                                                               \
	} while (0)
	if (ctx->ext_persist) BMH_LAUNCH_LANE_C(true, std::min(chunks, cap), 0);
	else {
		BMH_LAUNCH_LANE_C(false, chunks, 0);
		if (!exact) BMH_LAUNCH_LANE_C(true, 256, chunks); // beyond the estimate: usually nothing
	}
#undef BMH_LAUNCH_LANE_C
#undef BMH_LAUNCH_LANE
#undef BMH_LAUNCH_LANE2
	BMH_HIP(ctx, hipGetLastError());
	return BMH_OK;
}

} // namespace bmh
