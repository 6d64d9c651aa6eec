// global_lane.hip -- banded global alignment + traceback, 64 tasks per wave64 (one lane per task).
//
// Same results as ksw_global2 (reference bwa-0.7.8/ksw.c:501-584) and as global_kernel (one wave per task).
// The band of ksw_global2 is FIXED: row i covers columns [max(0,i-w), min(qlen,i+w+1)) (ksw.c:528-529).  So the
// row state is kept BAND-RELATIVE: slot s of row i is column j = i-w+s, 0 <= s <= 2w.  In these coordinates
//     the diagonal predecessor (i-1,j-1) is slot s   of the previous row  -> H is updated in place,
//     the vertical predecessor (i-1,j)   is slot s+1 of the previous row  -> E is read from the next register,
//     the horizontal predecessor (i,j-1) is slot s-1 of the same row      -> F is carried through the unrolled slots,
// and C = 64 (w <= 31) or 128 (w <= 63) registers hold the row of a task of ANY length: register s packs
// {H(i-1,j-1), E(i,j-1)} as two signed 16-bit halves (one v_perm_b32 to re-pack).  The query slides by one base per
// row: its v_perm selectors sit four per VGPR and the whole window is funnel-shifted by one byte per row
// (v_alignbyte_b32, C/4 instructions), the new byte coming from a per-lane global load issued one row ahead.
// Inactive slots are refilled every row with the first-column value -(o_del+e_del*(i+1)) (ksw.c:530) -- the slot just
// left of the band is the virtual column -1 that the next row's column 0 reads diagonally -- or with -inf.
// "-inf" is -16384: every comparison of the reference that involves MINUS_INF (ksw.c:487) has a finite value on the
// other side (a cell inside the band always has an in-band diagonal predecessor), so any sentinel below all finite
// scores reproduces it; the dispatcher sends tasks whose scores could reach -12000 to the int32 wave kernel.
// Direction state: the reference keeps one byte per cell (ksw.c:547-561: 2 bits "where H came from", 2 bits "E continues",
// 2 bits "F continues", of which 4 bits carry information).  Here a cell costs 4 bits -- the sign bits of the four
// differences the reference compares (see the fill) -- eight cells (one 8-slot block) per dword, and ONLY the blocks a wave computes are written, to a
// per-wave HBM slab laid out [block][row][lane]: the fill's stores are one coalesced 256-byte line per block and row,
// and the traceback (ksw.c:566-581: the reference's loop, one path per lane) walks up a block's rows, so the lanes of a
// wave -- sorted by band width, hence with their paths in the same block -- share the lines they fetch.
// (Round 1 stored a byte per cell for all C slots in [row][slot/4][lane] order: 3.35 GB of HBM traffic per 190 k tasks,
// profiles/traffic_latest.json; most of it the traceback pulling a 64-byte sector per 4-byte read.)
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "bmh_ctx.h"
#include "bmh_device.h"

namespace bmh {

constexpr int kNeg16 = -16384;
constexpr int kTbRows = 16;     // rows of a path's block fetched together by the traceback (LDS strip of kTbRows x 64 dwords = 4 KB)
constexpr int kStreamRows = 64; // rows of the {target base, incoming query base} stream staged in LDS at a time

__device__ __forceinline__ int sel3(int mask, int a, int b) { return __builtin_amdgcn_bitop3_b32(mask, a, b, 0xca); }

#ifndef BMH_GL_WAVES64
#define BMH_GL_WAVES64 3 /* round 3 (16-bit cell): 3 waves per SIMD hold the row loop without scratch traffic; 4 do not (100+ spills) */
#endif
#ifndef BMH_GL_WAVES128
#define BMH_GL_WAVES128 2
#endif
#ifndef BMH_GL_WAVES96
#define BMH_GL_WAVES96 2
#endif


// SYM: o_del + e_del == o_ins + e_ins (bwa's defaults): m - oe is shared by the E and F updates.  A run-time test of that inside the cell was
// compiled to a branch per cell.
template <int C, bool FAST, bool SYM>
__global__ __launch_bounds__(64, (C <= 64 ? BMH_GL_WAVES64 : C <= 96 ? BMH_GL_WAVES96 : BMH_GL_WAVES128)) void global_lane_kernel(
    const uint8_t *__restrict__ pool, const bmh_glb_task_t *__restrict__ tasks, const uint32_t *__restrict__ order,
    const uint32_t *__restrict__ count, long long n, bmh_glb_result_t *__restrict__ out, uint32_t *__restrict__ cigar_pool,
    DevParams P, uint32_t *__restrict__ zslab, int rows_cap, int *__restrict__ err_flag)
{
	constexpr int NW = (C + 31) / 32, NQ = C / 4, NB = C / 8;
	__shared__ uint2 srow[8];
	extern __shared__ __attribute__((aligned(16))) uint8_t dyn_lds[]; // max(C, kStreamRows) x 64 bytes: the window staging, then the per-row stream
	const int lane = threadIdx.x;
	const int oe_del = P.o_del + P.e_del, oe_ins = P.o_ins + P.e_ins;
	const int e_del = P.e_del, e_ins = P.e_ins;
	if (lane < 5) {
		uint32_t lo = 0;
		for (int q = 0; q < 4; ++q) lo |= (uint32_t)(uint8_t)mat_at(P, lane * 5 + q) << (8 * q);
		srow[lane] = make_uint2(lo, (uint32_t)(uint8_t)mat_at(P, lane * 5 + 4));
	}
	const long long cnt = count ? (long long)*count : n;
	// this wave's direction slab: a wave-uniform base (SGPRs); a store then needs only `lane` as its 32-bit vector offset
	uint32_t *__restrict__ zbase = zslab + (size_t)blockIdx.x * (size_t)rows_cap * (size_t)(NB * 64);

	for (long long base = (long long)blockIdx.x * 64; base < cnt; base += (long long)gridDim.x * 64) {
		const bool valid = base + lane < cnt;
		const long long pos = cnt - 1 - (valid ? base + lane : base); // sorted ascending by rows: longest first
		const uint32_t idx = order ? order[pos] : (uint32_t)pos;
		const uint4 *tp = (const uint4 *)(tasks + idx);
		const uint4 ta = tp[0], tb = tp[1];
		const uint64_t q_off = (uint64_t)ta.y << 32 | ta.x;
		const int qlen = (int)(tb.x & 0xffff), tlen = (int)(tb.x >> 16);
		const int w = (int)tb.y;
		const bool want = (int)tb.w > 0;
		const bool bad = w < 0 || 2 * w + 2 > C || tlen > rows_cap;
		if (valid && bad) {
			out[idx].score = INT32_MIN, out[idx].n_cigar = 0;
			atomicExch(err_flag, BMH_E_RANGE);
		}
		const bool live = valid && !bad;

		// ---- row state before row 0 (ksw.c:519-522): slot s is column s-w
		int R[C + 1], QW[NQ];
#pragma unroll
		for (int s = 0; s <= C; ++s) {
			const int j = s - w;
			const int hd = j < 0 ? kNeg16 : (j == 0 ? 0 : (j <= w && j <= qlen ? -(P.o_ins + e_ins * j) : kNeg16));
			R[s] = (int)((uint32_t)kNeg16 << 16 | ((uint32_t)hd & 0xffffu)); // {H(-1,j-1) as eh[j].h, E = -inf}
		}
		// ---- the sequences go through LDS.  (1) The query window: unrolled, its C byte loads are all in flight at once and their 64-bit
		// addresses and destinations set the kernel's register count (226 for C = 64); fetched in a rolled loop into LDS and read back
		// into QW[], the row loop fits 3 waves per SIMD.  (2) The per-row stream: row i consumes target base t[i] and the query base that
		// enters the window's top slot, q[i+C-w]; loaded from global memory inside the row loop they cost an `s_waitcnt vmcnt(0)` per row
		// (loads under an exec mask cannot be counted, and the counter they share with the direction-word stores runs in order: 40 % of
		// the waves' lifetime was spent waiting).  Both are staged in LDS, one byte per row (two nibbles) at [row][lane], kStreamRows rows
		// at a time: the row loop waits for memory once per kStreamRows rows.  The window staging area is reused for the stream.
		uint8_t *strm = dyn_lds;
		uint32_t *qstage = (uint32_t *)dyn_lds;
#pragma unroll 2
		for (int v = 0; v < NQ; ++v) {
			int sv = 0;
#pragma unroll
			for (int b = 0; b < 4; ++b) {
				const int j = 4 * v + b - w;
				int qb = 4;
				if (live && j >= 0 && j < qlen) qb = pool[q_off + (uint64_t)j];
				sv |= qb << (8 * b);
			}
			qstage[v * 64 + lane] = (uint32_t)sv;
		}
#pragma unroll
		for (int v = 0; v < NQ; ++v) QW[v] = (int)qstage[v * 64 + lane];
		// the stream holds kStreamRows rows at a time and is refilled by the row loop every kStreamRows rows (one wait per chunk)
		const int qsh = C - w; // row i takes q[i + qsh] into the top slot of the window of row i+1
		auto fill_stream = [&](int r0) {
			// the sequence offsets are read again from the task record here (once per kStreamRows rows) instead of living in four
			// VGPRs through the row loop: a spilled register costs a scratch reload per row, and that reload's s_waitcnt vmcnt(0)
			// also waits for every direction-word store in flight
			const uint4 tq = *(const uint4 *)(tasks + idx);
			const uint64_t q_off = (uint64_t)tq.y << 32 | tq.x, t_off = (uint64_t)tq.w << 32 | tq.z;
#pragma unroll 2
			for (int r4 = 0; r4 < kStreamRows; r4 += 4) { // four rows per trip: one (unaligned) dword of each sequence where it fits
				const int r = r0 + r4;
				uint32_t t4 = 0, q4 = 0x04040404u;
				if (live && r + 4 <= tlen) __builtin_memcpy(&t4, pool + t_off + (uint64_t)r, 4);
				else if (live) {
#pragma unroll
					for (int k = 0; k < 4; ++k)
						if (r + k < tlen) t4 |= (uint32_t)pool[t_off + (uint64_t)(r + k)] << (8 * k);
				}
				if (live && r + qsh + 4 <= qlen) __builtin_memcpy(&q4, pool + q_off + (uint64_t)(r + qsh), 4);
				else if (live) {
#pragma unroll
					for (int k = 0; k < 4; ++k)
						if (r + qsh + k < qlen) q4 = (q4 & ~(0xffu << (8 * k))) | (uint32_t)pool[q_off + (uint64_t)(r + qsh + k)] << (8 * k);
				}
				const uint32_t m4 = (t4 & 0x0f0f0f0fu) | (q4 & 0x0f0f0f0fu) << 4;
#pragma unroll
				for (int k = 0; k < 4; ++k) strm[(r4 + k) * 64 + lane] = (uint8_t)(m4 >> (8 * k));
			}
		};
		int i = 0, score = kNeg16;
		for (; __builtin_amdgcn_ballot_w64(live && i < tlen) != 0; ++i) { // ksw.c:524-564; i is wave-uniform
			const bool rowon = live && i < tlen;
			if ((i & (kStreamRows - 1)) == 0) fill_stream(i); // wave-uniform: rows [i, i + kStreamRows) of the {t, q} stream
			int lane_now; // (recomputed each row on purpose: kept live across the row it would be spilled, see fill_stream)
			asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_now));
			const int sbyte = strm[(i & (kStreamRows - 1)) * 64 + lane_now];
			const int tcur = sbyte & 15, qin = sbyte >> 4;
			const uint2 row = srow[min(tcur, 4)];
			const int slo = rowon ? max(0, w - i) : C + 1;
			const int shi = rowon ? min(2 * w + 1, qlen - i + w) : C + 1;
			int am[NW];
#pragma unroll
			for (int v = 0; v < NW; ++v) {
				const int lo = min(max(slo - 32 * v, 0), 32), hi = min(max(shi - 32 * v, 0), 32);
				am[v] = hi > lo ? (int)((0xffffffffu >> (32 - (hi - lo))) << lo) : 0;
			}
			const int fill = i < w ? -(P.o_del + e_del * (i + 1)) : kNeg16; // ksw.c:530: first-column value while beg == 0
			int f = kNeg16;
#pragma unroll
			for (int b = 0; b < NB; ++b) {
				// needed iff the block meets [slo-1, shi): slot slo-1 is the virtual column -1 that must receive `fill`
				if (__builtin_amdgcn_ballot_w64(slo <= 8 * b + 8 && shi > 8 * b) == 0) continue;
				// FAST: a block whose eight slots lie inside the band of EVERY lane still running needs no activity masks (no
				// select on F, E and H, no fill); lanes past their last row compute garbage into their own registers
				const bool masked = !FAST || __builtin_amdgcn_ballot_w64(rowon && !(slo <= 8 * b && shi >= 8 * b + 8)) != 0;
				// Direction state, 4 bits per cell, as the SIGN BITS of four differences (ksw.c:547-561):
				//   b1 = [m < e]  b2 = [max(m,e) < f]            -> d = b2 ? 2 : b1
				//   b3 = [e - e_del > m - oe_del] (E continues)    b4 = [f - e_ins > m - oe_ins] (F continues)
				// bit c of the four bytes of the block's dword = b1, b3, b2, b4 of cell c: two shift registers (dz12, dz34; fresh per block) take
				// two sign bits per cell at bits 15 and 31 and move down one place per cell.
				int dz12 = 0, dz34 = 0;
				auto cells = [&](auto MASKED) {
#pragma unroll
					for (int q4 = 0; q4 < 2; ++q4) {
						// the four substitution scores of slots 8b+4q4 .. +3 with one v_perm: the query codes are the selectors
						const int sc4 = (int)__builtin_amdgcn_perm(row.y, row.x, (unsigned)QW[2 * b + q4]);
#pragma unroll
						for (int c4 = 0; c4 < 4; ++c4) {
							const int s = 8 * b + 4 * q4 + c4;
							const int m = c4 == 0 ? add_score<0>(sc4, R[s]) : c4 == 1 ? add_score<1>(sc4, R[s]) : c4 == 2 ? add_score<2>(sc4, R[s]) : add_score<3>(sc4, R[s]); // ksw.c:546
							const int e = (int)((unsigned)R[s + 1] >> 16);  // E(i,j)
							const int h1 = max16(m, e);                     // ksw.c:547-550
							const int h = max16(h1, f);
							const int t1 = subk16(m, oe_del), e2 = subk16(e, e_del); // ksw.c:552-556
							const int t2 = SYM ? t1 : subk16(m, oe_ins), f2 = subk16(f, e_ins); // ksw.c:557-560
#ifdef BMH_GL_PERM_SIGNS // (round 3's first form: four differences, their high bytes gathered pairwise by v_perm)
							const int x1 = sub16(m, e), x2 = sub16(h1, f), x3 = sub16(t1, e2), x4 = sub16(t2, f2);
							const int z12 = (int)__builtin_amdgcn_perm((unsigned)x1, (unsigned)x2, 0x0c0c0105u);
							const int z34 = (int)__builtin_amdgcn_perm((unsigned)x3, (unsigned)x4, 0x0c0c0105u);
							dz12 = __builtin_amdgcn_bitop3_b32((int)((unsigned)dz12 >> 1), z12, 0x8080, 0xf8); // a | (b & c)
							dz34 = __builtin_amdgcn_bitop3_b32((int)((unsigned)dz34 >> 1), z34, 0x8080, 0xf8);
#else
							// two differences per register: the second one is subtracted straight into the high half (SDWA), so that one shift and
							// one v_bitop3 take both sign bits -- bits 15 and 31 -- into the shift register
							int x12 = sub16(m, e), x34 = sub16(t1, e2);
							sub16_into_hi(x12, h1, f);
							sub16_into_hi(x34, t2, f2);
							dz12 = __builtin_amdgcn_bitop3_b32((int)((unsigned)dz12 >> 1), x12, (int)0x80008000u, 0xf8); // a | (b & c)
							dz34 = __builtin_amdgcn_bitop3_b32((int)((unsigned)dz34 >> 1), x34, (int)0x80008000u, 0xf8);
#endif
							// register s <- {H(i,j) for the next row's diagonal, E(i+1,j) for the next row's slot s-1}
							if constexpr (decltype(MASKED)::value) {
								const int actv = (am[s / 32] << (31 - s % 32)) >> 31;
								f = sel3(actv, max16(f2, t2), kNeg16);
								R[s] = (int)__builtin_amdgcn_perm((unsigned)sel3(actv, max16(e2, t1), kNeg16), (unsigned)sel3(actv, h, fill), 0x05040100u);
							} else {
								f = max16(f2, t2);
								int r = h;
								max16_into_hi(r, e2, t1);
								R[s] = r;
							}
#ifdef BMH_GL_SCHED_CELL
							__builtin_amdgcn_sched_barrier(0);
#endif
						}
#ifdef BMH_GL_SCHED_Q4
						__builtin_amdgcn_sched_barrier(0);
#endif
					}
				};
				if (masked) cells(std::true_type{});
				else if constexpr (FAST) cells(std::false_type{});
#ifdef BMH_GL_PERM_SIGNS
				const uint32_t dz = (uint32_t)dz12 | (uint32_t)dz34 << 16; // bytes: b1 b2 b3 b4
#else
				const uint32_t dz = (uint32_t)dz12 >> 8 | (uint32_t)dz34; // eight cells on, b1 / b2 stand in bytes 1 / 3 of dz12, b3 / b4 of dz34: bytes b1 b3 b2 b4
#endif
				if (want) { // ksw.c:561, eight cells at once: wave-uniform line address in SGPRs + the lane's byte offset
					const uint32_t *line = zbase + ((size_t)b * (size_t)rows_cap + (size_t)i) * 64;
					asm volatile("global_store_dword %0, %1, %2" : : "v"(lane * 4), "v"(dz), "s"(line) : "memory");
				}
			}
			// score = eh[qlen].h after the LAST row of a lane = H(tlen-1, qlen-1), ksw.c:565: slot qlen-tlen+w of that row.
			// Picked up right here because the registers of a finished lane are refilled by the rows other lanes still run.
			if (__builtin_amdgcn_ballot_w64(live && i == tlen - 1)) {
				// register number ss picked by binary trees of selects on the bits of ss (C selects and a few compares; a compare per
				// register, hoisted out of the row loop by the compiler, used to cost 2 SGPRs per slot)
				const int ss = qlen - tlen + w;
				int pick = 0;
#pragma unroll
				for (int c0 = 0; c0 < C; c0 += 32) { // 32 slots at a time (16 temporaries), then the chunk by the high bits of ss
					int T[16];
#pragma unroll
					for (int k = 0; k < 16; ++k) {
						const int a0 = c0 + 2 * k < C ? R[c0 + 2 * k] : 0, a1 = c0 + 2 * k + 1 < C ? R[c0 + 2 * k + 1] : 0;
						T[k] = (ss & 1) ? a1 : a0;
					}
#pragma unroll
					for (int bit = 1, len = 8; len >= 1; ++bit, len >>= 1)
#pragma unroll
						for (int k = 0; k < len; ++k) T[k] = (ss >> bit & 1) ? T[2 * k + 1] : T[2 * k];
					pick = (ss >> 5) == c0 / 32 ? T[0] : pick;
				}
				if (live && i == tlen - 1 && ss >= 0 && ss < C) score = (int)(int16_t)(pick & 0xffff);
			}
			// slide the query window by one base (row i+1 looks at q[i+1-w+s])
#pragma unroll
			for (int v = 0; v + 1 < NQ; ++v) QW[v] = (int)__builtin_amdgcn_alignbyte((unsigned)QW[v + 1], (unsigned)QW[v], 1u);
			QW[NQ - 1] = (int)__builtin_amdgcn_alignbyte((unsigned)qin, (unsigned)QW[NQ - 1], 1u);
		}

		if (tlen == 0) score = qlen == 0 ? 0 : (qlen <= w ? -(P.o_ins + e_ins * qlen) : kNeg16); // eh[qlen].h of ksw.c:519-522
		if (score <= kNeg16 / 2) score = -0x40000000; // the reference's MINUS_INF (out-of-domain input only)

		// ---- traceback, ksw.c:566-581, one path per lane
		const uint2 tc = *(const uint2 *)((const uint8_t *)(tasks + idx) + 24); // {cigar_off, cigar_cap}, re-read (see fill_stream)
		const uint32_t cigar_off = tc.x;
		const int cigar_cap = (int)tc.y;
		int n_cigar = 0;
		if (__builtin_amdgcn_ballot_w64(live && want)) {
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); // direction words of this wave are in flight
			uint32_t *cg = cigar_pool + cigar_off;
			int ti = tlen - 1, tk = min(qlen, tlen - 1 + w + 1) - 1, which = 0, last_op = 0, last_len = 0, nw = 0;
			bool on = live && want;
			// A step needs one direction dword; read from the slab where the fill left it, every step is a dependent HBM/L2 round trip
			// (a third of the waves' lifetime, profiles/r03_global_lane.md).  A path moves up one row per step (two steps at an insertion) and
			// rarely leaves its 8-slot block, so the dwords of its block for the next kTbRows rows are fetched together -- kTbRows loads in
			// flight per lane -- into an LDS strip [k][lane], refilled for ALL lanes of the wave whenever one of them runs out (the lanes'
			// paths advance about a row per step, so they run out together).
			uint32_t *strip = (uint32_t *)dyn_lds;
			int cblk = -1, ctop = -1; // block and top row of this lane's strip: rows (ctop - kTbRows, ctop]
			while (__builtin_amdgcn_ballot_w64(on && ti >= 0 && tk >= 0)) {
				const bool act = on && ti >= 0 && tk >= 0;
				const int s = min(max(tk - (ti - w), 0), C - 1);
				if (__builtin_amdgcn_ballot_w64(act && ((s >> 3) != cblk || ti > ctop || ti <= ctop - kTbRows))) {
					if (act) {
						cblk = s >> 3, ctop = ti;
						const uint32_t *src = zbase + ((size_t)cblk * (size_t)rows_cap) * 64 + lane;
						uint32_t v[kTbRows];
#pragma unroll
						for (int k = 0; k < kTbRows; ++k) v[k] = ti - k >= 0 ? src[(size_t)(ti - k) * 64] : 0u;
#pragma unroll
						for (int k = 0; k < kTbRows; ++k) strip[k * 64 + lane] = v[k];
					}
				}
				if (act) {
					const uint32_t dzw = strip[(ctop - ti) * 64 + lane] >> (s & 7);
					// bit 0 = [m < e], bit 16 = [max(m,e) < f], bit 8 = E continues, bit 24 = F continues (see the fill)
#ifdef BMH_GL_PERM_SIGNS
					which = which == 0 ? ((dzw >> 8 & 1) ? 2 : (int)(dzw & 1)) : which == 1 ? (int)(dzw >> 16 & 1) : (int)(dzw >> 23 & 2);
#else
					which = which == 0 ? ((dzw >> 16 & 1) ? 2 : (int)(dzw & 1)) : which == 1 ? (int)(dzw >> 8 & 1) : (int)(dzw >> 23 & 2);
#endif
					const int op = which == 0 ? 0 : (which == 1 ? 2 : 1);
					if (last_len > 0 && op == last_op) ++last_len; // ksw.c:489-499
					else {
						if (last_len > 0) {
							if (nw < cigar_cap) cg[cigar_cap - 1 - nw] = (uint32_t)last_len << 4 | (uint32_t)last_op;
							++nw;
						}
						last_op = op, last_len = 1;
					}
					ti -= which != 2;
					tk -= which != 1;
				}
			}
			if (on) {
				for (int pass = 0; pass < 2; ++pass) { // leftovers, ksw.c:576-577: deletions then insertions
					const int op = pass == 0 ? 2 : 1, len = pass == 0 ? ti + 1 : tk + 1;
					if (len <= 0) continue;
					if (last_len > 0 && op == last_op) last_len += len;
					else {
						if (last_len > 0) {
							if (nw < cigar_cap) cg[cigar_cap - 1 - nw] = (uint32_t)last_len << 4 | (uint32_t)last_op;
							++nw;
						}
						last_op = op, last_len = len;
					}
				}
				if (last_len > 0) {
					if (nw < cigar_cap) cg[cigar_cap - 1 - nw] = (uint32_t)last_len << 4 | (uint32_t)last_op;
					++nw;
				}
				n_cigar = nw;
				// the words sit at [cap-n, cap) in forward order; move them to [0, n)
				const int shift = cigar_cap - nw;
				if (shift > 0 && nw <= cigar_cap) {
					__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
					for (int c = 0; c < nw; ++c) cg[c] = cg[shift + c];
				}
				if (nw > cigar_cap) atomicExch(err_flag, BMH_E_CIGAR_CAP);
			}
		}
		if (live) out[idx].score = score, out[idx].n_cigar = n_cigar;
	}
}

// ---- launcher: tasks listed in d_order[0..*d_count) must have 2w+2 <= C and tlen <= rows_cap
int launch_global_lane(bmh_ctx *ctx, int c, const uint8_t *d_pool, const bmh_glb_task_t *d_tasks, int64_t n,
                       bmh_glb_result_t *d_res, uint32_t *d_cigar, const uint32_t *d_order, const uint32_t *d_count,
                       int rows_cap)
{
	if (n <= 0) return BMH_OK;
	int ncu = 256;
	(void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device);
	long long grid = (n + 63) / 64;
	// persistent grid = the waves that are resident at once: every wave owns a private direction slab for as long as it
	// lives, so more blocks than that would only pin more HBM (2 048 waves x 170 rows x 8 blocks x 256 B = 0.7 GB at
	// 150 bp; a grow-only workspace per context, and the preload shim keeps one context per host thread)
	const long long resident = (long long)ncu * 4 * (c <= 64 ? BMH_GL_WAVES64 : c <= 96 ? BMH_GL_WAVES96 : BMH_GL_WAVES128);
	if (grid > resident) grid = resident;
	// one slab serves both lane kernels of a launch (they run back to back on the stream): [block][row][lane] dwords of
	// 8 cells, C/8 blocks per row
	const size_t slab = (size_t)grid * (size_t)rows_cap * (size_t)(c / 8) * 64 * 4;
	int rc = ensure(ctx, ctx->d_zslab, slab);
	if (rc) return rc;
	const bool fast = ctx->glb_fast != 0; // (A/B knob BMH_GL_FAST: 0 = masked body only)
	const bool sym = ctx->dev.o_del + ctx->dev.e_del == ctx->dev.o_ins + ctx->dev.e_ins;
	const size_t lds = (size_t)std::max(c, kStreamRows) * 64; // window staging (C/4 dwords per lane), then one byte per staged row and lane
#define BMH_LAUNCH_GL2(CC, FF, SS)                                                                                             \
	hipLaunchKernelGGL((global_lane_kernel<CC, FF, SS>), dim3((unsigned)grid), dim3(64), lds, ctx->stream, d_pool, d_tasks, d_order, \
	                   d_count, (long long)n, d_res, d_cigar, ctx->dev, (uint32_t *)ctx->d_zslab.p, rows_cap, ctx->d_err)
#define BMH_LAUNCH_GL(CC)                                                                                                      \
	do {                                                                                                                       \
		if (fast && sym) BMH_LAUNCH_GL2(CC, true, true);                                                                       \
		else if (fast) BMH_LAUNCH_GL2(CC, true, false);                                                                        \
		else if (sym) BMH_LAUNCH_GL2(CC, false, true);                                                                         \
		else BMH_LAUNCH_GL2(CC, false, false);                                                                                 \
	} while (0)
	if (c == 64) BMH_LAUNCH_GL(64);
	else if (c == 96) BMH_LAUNCH_GL(96);
	else if (c == 128) BMH_LAUNCH_GL(128);
	else return BMH_E_ARG;
#undef BMH_LAUNCH_GL
#undef BMH_LAUNCH_GL2
	BMH_HIP(ctx, hipGetLastError());
	return BMH_OK;
}

} // namespace bmh
