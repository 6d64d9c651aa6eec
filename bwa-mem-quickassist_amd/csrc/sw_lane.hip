// sw_lane.hip -- local Smith-Waterman (ksw_align2 byte mode, reference bwa-0.7.8/ksw.c:114-233,341-364) for
// mate rescue: 64 tasks per wave64, one LANE per task, the whole DP row in registers, packed 16-bit arithmetic.
//
// Same results as sw_generic.hip (whose header states the recurrence the striped reference amounts to).  Layout:
//   * a task's Q = 16*slen padded query columns are cut in two halves of Bs = roundup8(Q/2) columns.  Column jj of
//     half A (columns [0,Bs)) and column jj of half B ([Bs,2Bs)) share one VGPR as two u16 halves, and the two
//     halves run ONE ROW APART: in step s half A computes row s while half B computes row s-1, so what B needs
//     from A's last column (H(i-1,Bs-1), F(i,Bs)) was produced one step earlier and is handed over in registers.
//     Every v_pk_*_u16 instruction therefore advances two cells of the same task, and local alignment is exactly
//     unsigned saturating arithmetic: v_pk_add_u16/v_pk_sub_u16 with clamp are the reference's adds/subs_epu8.
//   * scores are kept x256 (tasks are only routed here when qlen*max(mat)+shift < 255, i.e. byte mode cannot
//     overflow): the low byte of each half is then free to carry the column tag 255-jj, and one v_pk_max_u16 per
//     column yields both the row maximum and the smallest column attaining it (ksw.c:204-206).
//   * substitution scores come from one v_perm_b32 per column over an 8-byte pool {4 biased scores of row s, 4 of
//     row s-1}; the per-lane selectors live in LDS (two columns per dword).  Columns holding N or (when the lanes of
//     a wave differ in length) padding take a slower corrected path, block-wise and only where some lane needs it.
//   * segment starts of the striped layout (F restarts at zero, ksw.c:139) and the end of the padded query are
//     folded into per-column subtrahends read from a small wave-uniform LDS table: sub-saturating 0xffff clears F.
//   * the lanes of a wave must agree on slen; the dispatcher sorts by query length, and a wave that still straddles
//     a boundary runs its groups one after the other.
// Pass 2 (start positions, ksw.c:355-361) is the same kernel over the reversed prefixes with KSW_XSTOP|score.
#include <algorithm>
#include <type_traits>

#include "bmh_ctx.h"
#include "bmh_device.h"
#include "sw_common.h"

namespace bmh {

constexpr int kSwTableBytes = 2048; // >= 16 * B
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ us2 as_us2(uint32_t x) { return __builtin_bit_cast(us2, x); }
__device__ __forceinline__ uint32_t as_u32(us2 x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ uint32_t pk_adds(uint32_t a, uint32_t b) { return as_u32(__builtin_elementwise_add_sat(as_us2(a), as_us2(b))); }
__device__ __forceinline__ uint32_t pk_subs(uint32_t a, uint32_t b) { return as_u32(__builtin_elementwise_sub_sat(as_us2(a), as_us2(b))); }
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) { return as_u32(__builtin_elementwise_max(as_us2(a), as_us2(b))); }
__device__ __forceinline__ uint32_t pk_shr(uint32_t a, int k) { return as_u32(as_us2(a) >> (us2)((unsigned short)k)); }
__device__ __forceinline__ uint32_t pk_mul(uint32_t a, uint32_t b) { return as_u32(as_us2(a) * as_us2(b)); }
__device__ __forceinline__ uint32_t pk_mad(uint32_t a, uint32_t b, uint32_t c) { return as_u32(as_us2(a) * as_us2(b) + as_us2(c)); }

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N-1>{})
template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F &&f)
{
	if constexpr (I < N) {
		f(std::integral_constant<int, I>{});
		static_for<N, I + 1>(f);
	}
}

struct SwLaneSeq { // where a lane's sequences come from
	const uint8_t *pool;
	uint64_t q_off, t_off;
	bool qrev, qcomp, trev, tpac;
	int qfold, tfold;
};

__device__ __forceinline__ int swl_qbase(const SwLaneSeq &s, int k)
{
	const int kk = s.qfold >= 0 ? s.qfold - k : k;
	int c = seq_base(s.pool, s.q_off, kk, s.qrev);
	c = c > 4 ? 4 : c;
	return s.qcomp && c < 4 ? 3 - c : c;
}

__device__ __forceinline__ int swl_tbase(const SwLaneSeq &s, const DevParams &P, int r)
{
	const int rr = r <= s.tfold ? s.tfold - r : r;
	const int c = tgt_base(s.pool, P, s.t_off, rr, s.trev, s.tpac);
	return c > 4 ? 4 : c;
}

// SYM: o_del == o_ins, H-o is shared by the E and F updates.  CORR: queries with N (and waves whose lanes differ in
// length) -- such columns take a per-lane score correction; the dispatcher keeps those tasks apart, and the plain
// instantiation groups its lanes by exact query length, so that padding is wave-uniform and needs no correction.
//
// WORD: ksw_i16's layout (8 segments instead of 16, no bias in the reference; the kernel keeps the bias, which cancels:
// max(H+S,0) == sat((H+S+shift)-shift)).  Scores are then kept x128 with a 7-bit column tag, good for scores below 512
// -- what bwa's callers send in word mode are queries of 250 columns and more (bwamem_pair.c:147), and B = 128 register
// pairs hold 256 of them.  The substitution byte is fetched into the LOW byte of each half and enters through
// v_pk_mad_u16 (x128 + H) in place of the add.
template <int B, bool SYM, bool CORR, bool WORD>
__global__ __launch_bounds__(64, B <= 40 ? 3 : B <= 80 ? 2 : 1) void sw_lane_kernel(const uint8_t *__restrict__ pool,
                                                                    const bmh_sw_task_t *__restrict__ tasks,
                                                                    const uint32_t *__restrict__ order,
                                                                    const uint32_t *__restrict__ count, long long n,
                                                                    bmh_sw_result_t *out, DevParams P,
                                                                    uint16_t *__restrict__ rmslab, int rows_cap,
                                                                    int pass2, int *__restrict__ err_flag,
                                                                    uint32_t *__restrict__ next_chunk)
{
	constexpr int NB = B / 8, NG = (B + 15) / 16;
	constexpr int SC = WORD ? 7 : 8, SEG = WORD ? 8 : 16; // score scale (bits), segments of the striped layout
	constexpr uint32_t TAGMAX = (1u << SC) - 1, SBND = WORD ? 512 : 255;
	__shared__ uint2 srow[8];              // [t] = {biased scores of target base t against A,C,G,T; against N}
	__shared__ uint32_t wl[(B / 2) * 64]; // v_perm selectors, two column pairs per dword, [word][lane]
	// the target, kSwStreamRows rows at a time, one byte per row at [row][lane]: fetched from global memory inside the row loop (under an
	// exec mask, so the compiler cannot count the load) every row paid an s_waitcnt vmcnt(0) -- a memory round trip plus the drain of
	// the row-maximum stores in flight (a third of the waves' lifetime in profiles/r03_*)
	constexpr int kSwStreamRows = 64;
	__shared__ uint8_t strm[kSwStreamRows * 64];
	__shared__ uint32_t ml[CORR ? 2 * ((B + 15) / 16) * 64 : 64]; // N / lane-specific padding bits per 16 columns, [2*g16+kind][lane]
	const int lane = threadIdx.x;
	if (count && *count == 0) return; // an empty bin (most launches of a small batch are): nothing to set up
	const uint32_t shift = (uint32_t)P.sw_shift;
	auto pair = [](uint32_t v) { return v << SC | v << (16 + SC); };
	const uint32_t odel = pair((uint32_t)P.o_del), edel = pair((uint32_t)P.e_del), oins = pair((uint32_t)P.o_ins);
	const uint32_t einspair = pair((uint32_t)P.e_ins);
	const uint32_t shpair = WORD ? shift | shift << 16 : shift << 8 | shift << 24; // as the substitution bytes are placed

	if (lane < 8) {
		uint32_t lo = 0, nn = 0;
		if (lane < 5) {
			for (int q = 0; q < 4; ++q) lo |= (uint32_t)(uint8_t)(mat_at(P, lane * 5 + q) + (int)shift) << (8 * q);
			nn = (uint32_t)(uint8_t)(mat_at(P, lane * 5 + 4) + (int)shift);
		}
		srow[lane] = make_uint2(lo, nn);
	}
	const long long cnt = count ? (long long)*count : n;
	// this wave's slab: the wave-uniform per-column subtrahend table {SH_j, Xseg_{j+1}} (written once per
	// group, then read back through the SCALAR cache: the values arrive in SGPRs, cost no VALU or LDS slot and no
	// VGPR), followed by the per-row maxima [row][lane]
	uint8_t *slab = (uint8_t *)rmslab + (size_t)blockIdx.x * ((size_t)rows_cap * 128 + kSwTableBytes);
	uint2 *xt = (uint2 *)slab;
	uint16_t *rm = (uint16_t *)(slab + kSwTableBytes) + lane;

	// chunks of 64 tasks are handed out dynamically (most expensive first): waves that drew short targets come back
	// for more instead of idling behind a static stride
	for (;;) {
		uint32_t ch = 0;
		if (lane == 0) ch = atomicAdd(next_chunk, 1u);
		const long long c0 = (long long)__builtin_amdgcn_readfirstlane(ch) * 64;
		if (c0 >= cnt) break;
		const bool valid = c0 + lane < cnt;
		const long long pos = cnt - 1 - (valid ? c0 + lane : c0); // long queries / long targets first
		const uint32_t idx = order ? order[pos] : (uint32_t)pos;
		const uint4 *tp = (const uint4 *)(tasks + idx);
		const uint4 ta = tp[0], tb = tp[1];
		SwLaneSeq seq;
		seq.pool = pool, seq.q_off = (uint64_t)ta.y << 32 | ta.x, seq.t_off = (uint64_t)ta.w << 32 | ta.z;
		const int tlen = (int)tb.x;
		const uint32_t flags = tb.y >> 16, xtra = tb.z;
		seq.qrev = flags & BMH_F_QREV, seq.trev = flags & BMH_F_TREV, seq.tpac = flags & BMH_F_TPAC, seq.qcomp = flags & BMH_F_QCOMP;
		seq.qfold = -1, seq.tfold = -1;
		int qlen = (int)(tb.y & 0xffff);
		const int thr = (int)(xtra & 0xffff);
		int minsc = (xtra & BMH_SW_XSUBO) ? thr : 0x10000, endsc = (xtra & BMH_SW_XSTOP) ? thr : 0x10000; // ksw.c:131-132
		int pscore = 0, pte = -1, pqe = -1;
		if (pass2 && valid) { // ksw.c:355-357: reversed prefixes, stop at the first pass's score
			const bmh_sw_result_t pr = out[idx];
			pscore = pr.score, pte = pr.te, pqe = pr.qe;
			qlen = pqe + 1, seq.qfold = pqe, seq.tfold = pte, minsc = 0x10000, endsc = pscore;
		}
		const bool want_rm = !pass2 && (xtra & BMH_SW_XSUBO);
		// the dispatcher only sends byte-mode tasks that cannot overflow and fit the register file
		const int slen = (qlen + SEG - 1) / SEG;
		const bool bad = valid && (qlen < 1 || ((slen * SEG / 2 + 7) & ~7) > B || !(xtra & BMH_SW_XBYTE) != WORD ||
		                           qlen * P.max_mat + (int)shift >= (int)SBND || tlen > rows_cap);
		if (bad) atomicExch(err_flag, BMH_E_RANGE);
		bool pending = valid && !bad;
		int r_score = 0, r_te = -1, r_qe = 0, r_rows = 0;

		while (__builtin_amdgcn_ballot_w64(pending) != 0) {
			const int first = __builtin_ctzll(__builtin_amdgcn_ballot_w64(pending));
			const int g = __builtin_amdgcn_readlane(slen, first), qlu = __builtin_amdgcn_readlane(qlen, first);
			const bool act = pending && slen == g && (CORR || qlen == qlu);
			pending = pending && !act;
			const int Qp = g * SEG;
			const int Bs = (Qp / 2 + 7) & ~7, nb = __builtin_amdgcn_readfirstlane(Bs / 8); // 8 columns per block
			const bool uni_q = __builtin_amdgcn_ballot_w64(act && qlen != qlu) == 0; // padding then is wave-uniform too
			__syncthreads();
			for (int jj = lane; jj < B + 8; jj += 64) { // (entries past Bs are never used, only prefetched)
				auto SH = [&](int c) { return c >= Qp ? 0xffffu : (uni_q && c >= qlu ? 0u : shift << SC); };
				auto XS = [&](int c) { return (c >= Qp || c % g == 0) ? 0xffffu : (uint32_t)P.e_ins << SC; };
				xt[jj] = make_uint2(SH(jj) | SH(Bs + jj) << 16, XS(jj + 1) | XS(Bs + jj + 1) << 16);
			}
			uint32_t MN[NG], MP[NG]; // per 16 columns: bit k = column is N / lane-specific padding (A low half, B high half)
#pragma unroll
			for (int v = 0; v < NG; ++v) MN[v] = MP[v] = 0;
#pragma unroll
			for (int w = 0; w < B / 2; ++w) {
				uint32_t word = 0x0c0c0c0cu;
				if (w < Bs / 2 && act) {
					word = 0;
#pragma unroll
					for (int k = 0; k < 2; ++k) {
						const int jj = 2 * w + k, ca = jj, cb = Bs + jj;
						const int qa = ca < qlen ? swl_qbase(seq, ca) : (ca < Qp ? 5 : 6);
						const int qb = cb < qlen ? swl_qbase(seq, cb) : (cb < Qp ? 5 : 6);
						word |= (uint32_t)(qa < 4 ? qa : 0x0c) << (16 * k) | (uint32_t)(qb < 4 ? 4 + qb : 0x0c) << (16 * k + 8);
						if (CORR) {
							MN[jj / 16] |= (uint32_t)(qa == 4) << (jj % 16) | (uint32_t)(qb == 4) << (16 + jj % 16);
							MP[jj / 16] |= (uint32_t)(qa == 5 && !uni_q) << (jj % 16) | (uint32_t)(qb == 5 && !uni_q) << (16 + jj % 16);
						}
					}
				}
				wl[w * 64 + lane] = word;
			}
			__threadfence();                 // the table is in L2 ...
			__builtin_amdgcn_s_dcache_inv(); // ... and no stale copy of it in the scalar cache
			uint32_t bflag = 0; // blocks in which some lane has a corrected column
#pragma unroll
			for (int b = 0; b < NB; ++b) {
				const uint32_t m = (MN[b / 2] | MP[b / 2]) & (0x00ff00ffu << (8 * (b & 1)));
				bflag |= (uint32_t)(__builtin_amdgcn_ballot_w64(m != 0) != 0) << b;
			}
#pragma unroll
			for (int v = 0; v < NG; ++v)
				if (CORR) ml[(2 * v) * 64 + lane] = MN[v], ml[(2 * v + 1) * 64 + lane] = MP[v];
			bflag = __builtin_amdgcn_readfirstlane(bflag);
			__syncthreads();

			uint32_t H[B], E[B];
#pragma unroll
			for (int jj = 0; jj < B; ++jj) H[jj] = E[jj] = 0;
			bool alive = act && tlen > 0;
			int gmax = 0, te = -1, qe = 0, nrows = 0;
			uint32_t hdB = 0, fsB = 0, ffB = 0, prevKA = TAGMAX;
			uint2 rB = make_uint2(0u, 0u);
			auto fill_stream = [&](int r0) {
#pragma unroll 4
				for (int r = 0; r < kSwStreamRows; ++r) strm[r * 64 + lane] = (uint8_t)(alive && r0 + r < tlen ? swl_tbase(seq, P, r0 + r) : 4);
			};
			fill_stream(0);
			int tn = strm[lane];
			uint32_t K = 0x0c0c0c0cu;
			int woff = lane;
			typedef const uint32_t __attribute__((address_space(4))) *sw_ctab_t;
			sw_ctab_t xc = (sw_ctab_t)(uintptr_t)xt;
			uint32_t k128 = 0x00800080u;
			asm volatile("" : "+s"(k128)); // opaque, or the multiply-add becomes a shift and an add

			for (int s = 0; __builtin_amdgcn_ballot_w64(alive) != 0; ++s) {
				const uint2 rA = srow[tn];
				if (((s + 1) & (kSwStreamRows - 1)) == 0) fill_stream(s + 1); // wave-uniform; row s's base is already in rA
				tn = strm[((s + 1) & (kSwStreamRows - 1)) * 64 + lane];
				const uint32_t plo = rA.x, phi = rB.x;
				const uint32_t vN = WORD ? rA.y | rB.y << 16 : rA.y << 8 | rB.y << 24;
				asm volatile("" : "+v"(K), "+v"(woff), "+s"(xc)); // keeps the row-invariant selector work inside the row loop
				uint32_t fs = fsB << 16, ff = ffB << 16, key = 0, hlast = hdB << 16;
				// the per-column subtrahend table comes through the scalar cache, one block (16 dwords) AHEAD of its use and issued from the
				// middle of the previous block: loaded where it is used, every block started with an s_waitcnt on a scalar-cache round trip
				// (ten per DP row), and issued next to an LDS read it is waited for with it (lgkmcnt counts both)
				uint32_t XT[2][16]; // block b reads XT[b & 1] and fills XT[(b + 1) & 1]: no copies (they would be hoisted to just behind the load)
#pragma unroll
				for (int k = 0; k < 16; ++k) XT[0][k] = xc[k];
#pragma unroll
				for (int b = 0; b < NB; ++b) {
					if (b >= nb) continue;
					{
						const bool flagged = CORR && ((bflag >> b) & 1);
						// the block's four selector words, read together at its start (read where they are used, each read is followed by an
						// s_waitcnt that also waits for the scalar-cache prefetch below)
						uint32_t WS[4];
#pragma unroll
						for (int k = 0; k < 4; ++k) WS[k] = wl[(4 * b + k) * 64 + woff];
						__builtin_amdgcn_sched_barrier(0);
						// S'(A) << 8 | S'(B) << 24 of column pair c of the block: one v_perm over {row s scores, row s-1 scores}
						auto subst = [&](int c) {
							const uint32_t sel = __builtin_amdgcn_perm(WS[c / 2], K,
							                                           WORD ? ((c & 1) ? 0x00070006u : 0x00050004u) : ((c & 1) ? 0x07000600u : 0x05000400u));
							uint32_t sp = __builtin_amdgcn_perm(phi, plo, sel);
							if (CORR && flagged) { // N columns score mat[t][4]; padding the other lanes do not share scores 0
								const int jj = 8 * b + c;
								const uint32_t xn = pk_shr(ml[(2 * (jj / 16)) * 64 + woff], jj % 16) & 0x00010001u;
								const uint32_t xp = pk_shr(ml[(2 * (jj / 16) + 1) * 64 + woff], jj % 16) & 0x00010001u;
								sp += pk_mul(xn, vN) + pk_mul(xp, shpair);
							}
							return sp;
						};
						// a = H(i-1,j-1) + S' of the column about to be computed; inside a block it is formed while the left
						// neighbour still holds its previous-row value, so that the new H is written in place; hlast carries
						// that value across block boundaries and, at the end, over to half B
						// (WORD: the byte sits in the low byte of each half and is scaled on the way in; nothing can overflow)
						auto enter = [&](uint32_t h, uint32_t sp) { return WORD ? pk_mad(sp, k128, h) : pk_adds(h, sp); };
						uint32_t a = enter(hlast, subst(0));
#pragma unroll
						for (int c = 0; c < 8; ++c) {
							const int jj = 8 * b + c;
							if (c == 5 && b + 1 < NB) { // behind the block's LDS waits (lgkmcnt counts both): six cells for the scalar cache to answer
								__builtin_amdgcn_sched_barrier(0);
#pragma unroll
								for (int k = 0; k < 16; ++k) XT[(b + 1) & 1][k] = xc[16 * (b + 1) + k];
								__builtin_amdgcn_sched_barrier(0);
							}
							const uint32_t xsh = XT[b & 1][2 * c], xseg = XT[b & 1][2 * c + 1];
							const uint32_t m = pk_subs(a, xsh);                  // ksw.c:149-150
							const uint32_t hp = pk_max(pk_max(m, E[jj]), fs);          // ksw.c:151-153
							const uint32_t h = pk_max(hp, ff);                          // lazy F, ksw.c:165-176
							const uint32_t tag = (TAGMAX - (uint32_t)jj) * 0x00010001u;
							key = pk_max(key, h | tag);
							if (c < 7) a = enter(H[jj], subst(c + 1));
							else hlast = H[jj];
							H[jj] = h;
							const uint32_t t1 = pk_subs(hp, odel);
							E[jj] = pk_subs(pk_max(E[jj], t1), edel);                   // ksw.c:155-158
							const uint32_t t2 = SYM ? t1 : pk_subs(hp, oins);
							fs = pk_subs(pk_max(fs, t2), xseg);                   // ksw.c:160-162; 0xffff restarts a segment
							ff = pk_subs(pk_max(ff, t2), einspair);
						}
					}
				}
				hdB = hlast & 0xffff, fsB = fs & 0xffff, ffB = ff & 0xffff;
				rB = rA;
				const uint32_t kA = prevKA, kB = key >> 16;
				prevKA = key & 0xffff;
				if (s >= 1 && alive) { // row i = s-1 is complete
					const int i = s - 1;
					const int ia = (int)(kA >> SC), ib = (int)(kB >> SC), imax = max(ia, ib);
					const int arg = ia >= ib ? (int)(TAGMAX - (kA & TAGMAX)) : Bs + (int)(TAGMAX - (kB & TAGMAX));
					if (want_rm) rm[(size_t)i * 64] = (uint16_t)imax;
					nrows = i + 1;
					if (imax > gmax) { // ksw.c:190-195
						gmax = imax, te = i, qe = arg;
						if (gmax >= endsc) alive = false;
					}
					if (i + 1 >= tlen) alive = false;
				}
			}
			if (act) r_score = gmax, r_te = te, r_qe = qe, r_rows = nrows;
		}

		// ---- results
		if (!pass2) {
			int s2 = -1, t2 = -1;
			if (__builtin_amdgcn_ballot_w64(want_rm && !bad) != 0) { // ksw.c:181-189,209-220 replayed for all lanes together
				const int maxrows = wave_reduce_max(valid && !bad && want_rm ? r_rows : 0);
				const int d = (r_score + P.max_mat - 1) / P.max_mat, low = r_te - d, high = r_te + d;
				bool have = false;
				int lsc = 0, li = 0;
				for (int i = 0; i < maxrows; ++i) {
					if (!(want_rm && i < r_rows)) continue;
					const int im = rm[(size_t)i * 64];
					if (im < minsc) continue;
					if (!have || li + 1 != i) {
						if (have && (li < low || li > high) && lsc > s2) s2 = lsc, t2 = li;
						have = true, lsc = im, li = i;
					} else if (lsc < im) lsc = im, li = i;
				}
				if (have && (li < low || li > high) && lsc > s2) s2 = lsc, t2 = li;
			}
			if (valid) {
				bmh_sw_result_t res;
				res.score = bad ? INT32_MIN : r_score, res.te = r_te, res.qe = r_qe, res.score2 = s2, res.te2 = t2;
				res.tb = -1, res.qb = -1, res.rsv = CORR; // tells the second pass's routing that the query holds an N
				out[idx] = res;
			}
		} else if (valid && !bad && r_score == pscore) { // ksw.c:360-361
			out[idx].tb = pte - r_te, out[idx].qb = pqe - r_qe;
		}
	}
}

int launch_sw_lane(bmh_ctx *ctx, int b, bool corr, bool word, const uint8_t *d_pool, const bmh_sw_task_t *d_tasks, int64_t n,
                   bmh_sw_result_t *d_res, const uint32_t *d_order, const uint32_t *d_count, uint16_t *d_rm, int rows_cap,
                   int grid, int pass2, uint32_t *d_next)
{
	if (n <= 0) return BMH_OK;
	const long long blocks = std::min<long long>((n + 63) / 64, grid);
	const bool sym = ctx->dev.o_del == ctx->dev.o_ins;
#define BMH_LAUNCH_SW(BB, SS, CC, WW)                                                                                    \
	hipLaunchKernelGGL((sw_lane_kernel<BB, SS, CC, WW>), dim3((unsigned)blocks), dim3(64), 0, ctx->stream, d_pool,       \
	                   d_tasks, d_order, d_count, (long long)n, d_res, ctx->dev, d_rm, rows_cap, pass2, ctx->d_err, d_next)
#define BMH_LAUNCH_SW2(BB, CC, WW)                                                                                       \
	do {                                                                                                                 \
		if (sym) BMH_LAUNCH_SW(BB, true, CC, WW);                                                                        \
		else BMH_LAUNCH_SW(BB, false, CC, WW);                                                                           \
	} while (0)
	if (b == 40 && !corr && !word) BMH_LAUNCH_SW2(40, false, false);
	else if (b == 40 && !word) BMH_LAUNCH_SW2(40, true, false);
	else if (b == 80 && !corr && !word) BMH_LAUNCH_SW2(80, false, false);
	else if (b == 80 && !word) BMH_LAUNCH_SW2(80, true, false);
	else if (b == 128 && corr && !word) BMH_LAUNCH_SW2(128, true, false); // the 256-column instantiations always correct
	else if (b == 128 && corr) BMH_LAUNCH_SW2(128, true, true);
	else return BMH_E_ARG;
#undef BMH_LAUNCH_SW2
#undef BMH_LAUNCH_SW
	BMH_HIP(ctx, hipGetLastError());
	return BMH_OK;
}

} // namespace bmh
