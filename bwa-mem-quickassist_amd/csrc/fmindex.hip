// fmindex.hip -- FM-index queries of BWA-MEM's seeding stage on the device (first slice):
//   smem_kernel  one lane per read: the bwt_smem1 calls (reference bwa-0.7.8/bwt.c:288-347) of smem_next2's iteration
//                (bwamem.c:118-162) over bwt_extend (bwt.c:261-274) / bwt_occ4 (bwt.c:159-177)
//   sa_kernel    one lane per suffix-array entry: bwt_sa (bwt.c:85-95) over bwt_invPsi (:52-58) / bwt_occ (:107-129)
// The index is the reference's own: the BWT with its interleaved occurrence counts (one 64-byte block per 128
// symbols: 4 x u64 counts + 8 words of 2-bit symbols, bwt.h:63-64) and the sampled suffix array, resident in HBM.
// Every occurrence query touches exactly one such block; counting is done with v_bcnt on equality masks instead of
// the reference's 256-entry byte table (same numbers).  This stage is the one part of the pipeline that is bound by
// random memory access rather than by instruction issue.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "bmh_ctx.h"
#include "bmh_device.h"

namespace bmh {

struct DevBwt {
	uint64_t primary, L2[5], seq_len;
	const uint32_t *bwt;
	const uint64_t *sa;
	uint64_t sa_mask; // sa_intv - 1
	int sa_shift;     // log2(sa_intv)
};

struct Intv { // == bwtintv_t
	uint64_t x0, x1, x2, info;
};

// Four counts / candidates as NAMED members, picked by a select chain: a per-lane index into an array would send the
// whole array through scratch memory on every extension (it did: 128 bytes written and re-read per bwt_extend).
struct U4 {
	uint64_t a, b, c, d;
};
__device__ __forceinline__ uint64_t pick(const U4 &v, int i) { return i == 0 ? v.a : i == 1 ? v.b : i == 2 ? v.c : v.d; }

// bwt_occ4, bwt.c:159-177.  The four counts of a word come from three population counts: with x = the symbols' high
// bits and y = their low bits (both moved to the even bit positions and cut to the prefix that counts),
// #3 = popc(x & y), #2 = popc(x) - #3, #1 = popc(y) - #3, #0 = symbols counted - the rest; v_bcnt accumulates for free.
struct Blk { // one 64-byte block: 4 x u64 counts before it, 128 symbols
	uint4 c0, c1, w0, w1;
};
__device__ __forceinline__ Blk fm_load(const DevBwt &B, uint64_t kk) // kk = position with the sentinel taken out
{
	const uint4 *blk = (const uint4 *)(B.bwt + ((kk >> 7) << 4));
	return Blk{blk[0], blk[1], blk[2], blk[3]};
}
__device__ __forceinline__ U4 fm_count4(const Blk &b, uint64_t kk)
{
	const uint32_t w[8] = {b.w0.x, b.w0.y, b.w0.z, b.w0.w, b.w1.x, b.w1.y, b.w1.z, b.w1.w};
	const int full = (int)((kk & 127) >> 4), rest = (int)(kk & 15) + 1; // whole words, symbols of the next one (symbol 0 = top bits)
	uint32_t px = 0, py = 0, pxy = 0;
#pragma unroll
	for (int j = 0; j < 8; ++j) {
		const uint32_t keep = (j < full ? 0xffffffffu : j == full ? 0xffffffffu << (32 - 2 * rest) : 0u) & 0x55555555u;
		const uint32_t x = (w[j] >> 1) & keep, y = w[j] & keep;
		px += (uint32_t)__popc(x), py += (uint32_t)__popc(y), pxy += (uint32_t)__popc(x & y);
	}
	const uint32_t n3 = pxy, n2 = px - pxy, n1 = py - pxy, n0 = (uint32_t)(kk & 127) + 1 - n1 - n2 - n3;
	U4 r;
	r.a = ((uint64_t)b.c0.y << 32 | b.c0.x) + (uint64_t)n0, r.b = ((uint64_t)b.c0.w << 32 | b.c0.z) + (uint64_t)n1;
	r.c = ((uint64_t)b.c1.y << 32 | b.c1.x) + (uint64_t)n2, r.d = ((uint64_t)b.c1.w << 32 | b.c1.z) + (uint64_t)n3;
	return r;
}
__device__ __forceinline__ U4 fm_occ4(const DevBwt &B, uint64_t k)
{
	if (k == (uint64_t)-1) return U4{0, 0, 0, 0};
	k -= (k >= B.primary); // the sentinel is not stored
	return fm_count4(fm_load(B, k), k);
}

// bwt_extend, bwt.c:261-274, for ONE base c (the only one of the four the caller goes on with)
__device__ __forceinline__ Intv fm_extend(const DevBwt &B, const Intv &ik, int c, bool is_back)
{
	const uint64_t a = is_back ? ik.x0 : ik.x1, b = is_back ? ik.x1 : ik.x0; // a = x[!is_back], b = x[is_back]
	// (reading the block once when both ends of a small interval share it -- bwt_2occ4, bwt.c:188-219 -- was measured and
	// is slower here: the second read hits the first one's cache line anyway and the branch costs more than it saves)
	const U4 tk = fm_occ4(B, a - 1), tl = fm_occ4(B, a - 1 + ik.x2);
	const U4 sz{tl.a - tk.a, tl.b - tk.b, tl.c - tk.c, tl.d - tk.d}; // ok[i].x[2]
	const U4 l2{B.L2[0], B.L2[1], B.L2[2], B.L2[3]};
	// ok[3].x[is_back] = b + [the sentinel lies in the interval]; ok[i].x[is_back] = ok[i+1].x[is_back] + ok[i+1].x[2]
	const uint64_t n3 = b + (a <= B.primary && a + ik.x2 - 1 >= B.primary), n2 = n3 + sz.d, n1 = n2 + sz.c, n0 = n1 + sz.b;
	const U4 nb{n0, n1, n2, n3};
	const uint64_t na = pick(l2, c) + 1 + pick(tk, c);
	Intv o;
	o.x2 = pick(sz, c);
	o.x0 = is_back ? na : pick(nb, c);
	o.x1 = is_back ? pick(nb, c) : na;
	o.info = 0;
	return o;
}

// a lane's three interval stacks live in a slab laid out [entry][lane]: lanes of a wave walking their stacks in step
// touch adjacent 32-byte records
struct Stack {
	Intv *base;
	__device__ __forceinline__ Intv &operator[](int j) const { return base[(size_t)j * 64]; }
};

// bwt_smem1, bwt.c:288-347.  mem receives the result in reference order; returns the next start; *n_mem = mem->n.
__device__ int fm_smem1(const DevBwt &B, int len, const uint8_t *q, int x, int min_intv, Stack prev, Stack curr, Stack mem,
                        int *n_mem)
{
	*n_mem = 0;
	if (q[x] > 3) return x + 1;
	if (min_intv < 1) min_intv = 1;
	Intv ik;
	int np, nc = 0, nm = 0, i;
	{
		const int c = q[x];
		const U4 l2{B.L2[0], B.L2[1], B.L2[2], B.L2[3]}, l2n{B.L2[1], B.L2[2], B.L2[3], B.L2[4]};
		ik.x0 = pick(l2, c) + 1, ik.x2 = pick(l2n, c) - pick(l2, c), ik.x1 = pick(l2, 3 - c) + 1, ik.info = (uint64_t)x + 1; // bwt_set_intv
	}
	for (i = x + 1; i < len; ++i) { // forward search
		if (q[i] < 4) {
			const Intv o = fm_extend(B, ik, 3 - q[i], false);
			if (o.x2 != ik.x2) {
				curr[nc++] = ik;
				if (o.x2 < (uint64_t)min_intv) break;
			}
			ik = o, ik.info = (uint64_t)i + 1;
		} else {
			curr[nc++] = ik;
			break;
		}
	}
	if (i == len) curr[nc++] = ik;
	for (int j = 0; j < nc >> 1; ++j) { // longer matches first
		const Intv t = curr[nc - 1 - j];
		curr[nc - 1 - j] = curr[j], curr[j] = t;
	}
	const int ret = (int)curr[0].info;
	{ const Stack t = curr; curr = prev, prev = t; }
	np = nc;
	uint64_t mem_start = 0; // mem[nm-1].info >> 32, kept in a register (it is tested once per interval and step)
	for (i = x - 1; i >= -1; --i) { // backward search for MEMs
		const int c = i < 0 ? -1 : q[i] < 4 ? q[i] : -1;
		uint64_t last_x2 = 0; // curr[nc-1].x2, likewise
		nc = 0;
		for (int j = 0; j < np; ++j) {
			const Intv p = prev[j];
			bool keep = c < 0; // the start of the read or an ambiguous base ends every interval: no extension needed
			Intv o{};
			if (!keep) {
				o = fm_extend(B, p, c, true);
				keep = o.x2 < (uint64_t)min_intv;
			}
			if (keep) {
				if (nc == 0 && (nm == 0 || (uint64_t)(i + 1) < mem_start)) {
					Intv m = p;
					m.info |= (uint64_t)(i + 1) << 32;
					mem[nm++] = m, mem_start = (uint64_t)(i + 1);
				}
			} else if (nc == 0 || o.x2 != last_x2) {
				Intv m = o;
				m.info = p.info;
				curr[nc++] = m, last_x2 = o.x2;
			}
		}
		if (nc == 0) break;
		{ const Stack t = curr; curr = prev, prev = t; }
		np = nc;
	}
	*n_mem = nm; // (still in reverse order; the caller writes it out back to front)
	return ret;
}

template <int WAVES> // resident waves per SIMD the register allocation aims at
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void smem_kernel(DevBwt B, const uint8_t *__restrict__ pool,
                                                  const uint64_t *__restrict__ read_off, const int *__restrict__ read_len,
                                                  int n_reads, bmh_smem_opt_t O, Intv *scratch, int lcap,
                                                  bmh_smem_call_t *calls, uint32_t *call_read, unsigned long long *cursors,
                                                  unsigned long long call_cap, Intv *intv, unsigned long long intv_cap,
                                                  int *overflow, int lanes)
{
	// `lanes` (a power of two <= 64) reads per wave: the reads of a wave sit in different loops of bwt_smem1 most of the
	// time and the wave pays for every path in turn, so a batch too small to fill the chip is spread over more, emptier
	// waves -- its latency then is that of a few reads' paths, not of 64
	const int lane = threadIdx.x;
	if (lane >= lanes) return;
	Intv *slab = scratch + (size_t)blockIdx.x * 3 * (size_t)lcap * 64 + lane;
	const Stack s0{slab}, s1{slab + (size_t)lcap * 64}, sm{slab + 2 * (size_t)lcap * 64};
	for (long long r = (long long)blockIdx.x * lanes + lane; r < n_reads; r += (long long)gridDim.x * lanes) {
		const uint8_t *q = pool + read_off[r];
		const int len = read_len[r];
		const int split_len = min(O.split_len, len); // bwamem.c:213
		int start = 0, seq = 0;
		while (start < len) { // smem_next2, bwamem.c:118-162
			while (start < len && q[start] > 3) ++start;
			if (start == len) break;
			int n, n2 = 0, x2 = 0, mi2 = 0, ret2 = 0;
			const int x1 = start;
			const int ret = fm_smem1(B, len, q, x1, O.start_width, s0, s1, sm, &n);
			start = ret;
			// the longest match, first of equals (sm holds them back to front)
			int mx = 0, mxk = 0, ne = 0; // ne = intervals long enough to be returned (bmh_smem_opt_t.min_emit_len)
			for (int i = 0; i < n; ++i) {
				const Intv v = sm[n - 1 - i];
				const int l = (int)((uint32_t)v.info - (uint32_t)(v.info >> 32));
				if (mx < l) mx = l, mxk = n - 1 - i;
				ne += l >= O.min_emit_len;
			}
			// write call 1
			unsigned long long ci = atomicAdd(&cursors[0], 1ull), base = atomicAdd(&cursors[1], (unsigned long long)ne);
			if (ci < call_cap && base + ne <= intv_cap) {
				bmh_smem_call_t c;
				c.x = x1, c.min_intv = O.start_width, c.ret = ret, c.n = ne, c.first = (uint32_t)base, c.rsv = (uint32_t)seq;
				calls[ci] = c, call_read[ci] = (uint32_t)r;
				for (int i = 0, k = 0; i < n; ++i) {
					const Intv v = sm[n - 1 - i];
					if ((int)((uint32_t)v.info - (uint32_t)(v.info >> 32)) >= O.min_emit_len) intv[base + k++] = v;
				}
			} else atomicExch(overflow, 1);
			++seq;
			bool split = false;
			if (n > 0 && split_len > 0 && mx >= split_len) {
				const Intv v = sm[mxk];
				if (v.x2 <= (uint64_t)O.split_width) split = true, x2 = (int)(((uint32_t)v.info + (uint32_t)(v.info >> 32)) >> 1), mi2 = (int)v.x2 + 1;
			}
			if (split) { // re-seeding from the middle of the longest match
				ret2 = fm_smem1(B, len, q, x2, mi2, s0, s1, sm, &n2);
				int ne2 = 0;
				for (int i = 0; i < n2; ++i) {
					const Intv v = sm[i];
					ne2 += (int)((uint32_t)v.info - (uint32_t)(v.info >> 32)) >= O.min_emit_len;
				}
				ci = atomicAdd(&cursors[0], 1ull), base = atomicAdd(&cursors[1], (unsigned long long)ne2);
				if (ci < call_cap && base + ne2 <= intv_cap) {
					bmh_smem_call_t c;
					c.x = x2, c.min_intv = mi2, c.ret = ret2, c.n = ne2, c.first = (uint32_t)base, c.rsv = (uint32_t)seq;
					calls[ci] = c, call_read[ci] = (uint32_t)r;
					for (int i = 0, k = 0; i < n2; ++i) {
						const Intv v = sm[n2 - 1 - i];
						if ((int)((uint32_t)v.info - (uint32_t)(v.info >> 32)) >= O.min_emit_len) intv[base + k++] = v;
					}
				} else atomicExch(overflow, 1);
				++seq;
			}
		}
	}
}

// ---- the same work with ONE bwt_extend site that all lanes reach together ---------------------------------------------------------
// In smem_kernel above the 64 reads of a wave sit in different loops of bwt_smem1 most of the time, and the wave issues every
// path for the few lanes on it: measured 25 % of the VALU lanes active (SQ_THREAD_CYCLES_VALU / 64 SQ_ACTIVE_INST_VALU), at
// about 200 instructions per extension.  Here a lane is a small machine {next read, next call, forward at i, backward at (i,j)}:
// per trip it first runs the transitions that need no extension (setting up a call, closing the forward pass, emitting a
// finished call, drawing the next read from a shared counter), then every lane that has an extension to do does it in the
// same instruction stream, then consumes the result.  Same calls, same intervals, same order within a read as above.
// Emitting a finished call is a chain of dependent memory operations (two returning atomics, the intervals read back from the
// lane's stack and written out, the next call's first bases): done the moment a lane gets there it would stall the other 63
// on almost every trip.  A lane that has finished a call therefore WAITS (PH_EMIT) until `emit_min` lanes of the wave do, or
// nobody has an extension left; then they emit, and set up their next calls, together.
enum { PH_READ = 0, PH_CALL = 1, PH_FWD = 2, PH_BWD = 3, PH_EMIT = 4, PH_DONE = 5 };

template <int WAVES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void smem_conv_kernel(
    DevBwt B, const uint8_t *__restrict__ pool, const uint64_t *__restrict__ read_off, const int *__restrict__ read_len, int n_reads,
    bmh_smem_opt_t O, Intv *scratch, int lcap, bmh_smem_call_t *calls, uint32_t *call_read, unsigned long long *cursors,
    unsigned long long call_cap, Intv *intv, unsigned long long intv_cap, int *overflow, int emit_min)
{
	const int lane = threadIdx.x;
	Intv *slab = scratch + (size_t)blockIdx.x * 3 * (size_t)lcap * 64 + lane;
	Stack prev{slab}, curr{slab + (size_t)lcap * 64};
	int emit_n = 0, emit_ret = 0;
	const Stack mem{slab + 2 * (size_t)lcap * 64};
	const U4 l2{B.L2[0], B.L2[1], B.L2[2], B.L2[3]}, l2n{B.L2[1], B.L2[2], B.L2[3], B.L2[4]};

	int phase = PH_READ;
	// read
	long long r = 0;
	const uint8_t *q = pool;
	int len = 0, split_len = 0, start = 0, seq = 0;
	bool pend_split = false, is_split = false;
	int x2 = 0, mi2 = 0;
	// call
	int x = 0, min_intv = 1, ret = 0, i = 0, j = 0, np = 0, nc = 0, nm = 0, cb = -1;
	bool rev = false; // the first backward row walks the forward pass's stack from its top (bwt.c:316 reverses it instead)
	uint64_t mem_start = 0, last_x2 = 0;
	Intv ik{}, p{};

	// the call is over: emit it (smem_next2's bookkeeping, bwamem.c:118-162) and decide what comes next
	auto finish_call = [&](int n, int ret_) {
		int mx = 0, mxk = 0, ne = 0; // ne = intervals long enough to be returned (bmh_smem_opt_t.min_emit_len)
		for (int k = 0; k < n; ++k) { // the longest match, first of equals (mem holds them back to front)
			const uint64_t info = mem[n - 1 - k].info;
			const int l = (int)((uint32_t)info - (uint32_t)(info >> 32));
			if (!is_split && mx < l) mx = l, mxk = n - 1 - k;
			ne += l >= O.min_emit_len;
		}
		const unsigned long long ci = atomicAdd(&cursors[0], 1ull), base = atomicAdd(&cursors[1], (unsigned long long)ne);
		if (ci < call_cap && base + ne <= intv_cap) {
			bmh_smem_call_t c;
			c.x = x, c.min_intv = is_split ? mi2 : O.start_width, c.ret = ret_, c.n = ne, c.first = (uint32_t)base, c.rsv = (uint32_t)seq;
			calls[ci] = c, call_read[ci] = (uint32_t)r;
			for (int k = 0, w = 0; k < n && w < ne; ++k) {
				const Intv v = mem[n - 1 - k];
				if ((int)((uint32_t)v.info - (uint32_t)(v.info >> 32)) >= O.min_emit_len) intv[base + w++] = v;
			}
		} else atomicExch(overflow, 1);
		++seq;
		if (!is_split) {
			start = ret_;
			if (n > 0 && split_len > 0 && mx >= split_len) {
				const Intv v = mem[mxk];
				if (v.x2 <= (uint64_t)O.split_width)
					pend_split = true, x2 = (int)(((uint32_t)v.info + (uint32_t)(v.info >> 32)) >> 1), mi2 = (int)v.x2 + 1;
			}
		}
		phase = PH_CALL;
	};
	// the forward pass is over (bwt.c:312-318): its stack becomes the first backward row, longest match first
	auto end_forward = [&]() {
		ret = (int)curr[nc - 1].info;
		np = nc, rev = true;
		{ const Stack t = curr; curr = prev, prev = t; }
		i = x - 1, cb = i < 0 ? -1 : q[i] < 4 ? q[i] : -1;
		j = 0, nc = 0, nm = 0, last_x2 = 0, mem_start = 0;
		phase = PH_BWD;
	};

	auto call_over = [&](int n, int ret_) { emit_n = n, emit_ret = ret_, phase = PH_EMIT; };

	for (;;) {
		// ---- finished calls, in batches
		{
			const unsigned long long waiting = __builtin_amdgcn_ballot_w64(phase == PH_EMIT);
			const unsigned long long busy = __builtin_amdgcn_ballot_w64(phase == PH_FWD || phase == PH_BWD);
			if (waiting != 0 && (__popcll(waiting) >= emit_min || busy == 0) && phase == PH_EMIT) finish_call(emit_n, emit_ret);
		}
		// ---- transitions that need no extension
		while (phase <= PH_BWD) {
			if (phase == PH_READ) {
				r = (long long)atomicAdd(&cursors[3], 1ull);
				if (r >= n_reads) { phase = PH_DONE; break; }
				q = pool + read_off[r], len = read_len[r];
				split_len = min(O.split_len, len); // bwamem.c:213
				start = 0, seq = 0, pend_split = false;
				phase = PH_CALL;
			}
			if (phase == PH_CALL) {
				if (pend_split) x = x2, min_intv = mi2, pend_split = false, is_split = true; // re-seeding from the middle of the longest match
				else {
					while (start < len && q[start] > 3) ++start;
					if (start >= len) { phase = PH_READ; continue; }
					x = start, min_intv = O.start_width, is_split = false;
				}
				if (min_intv < 1) min_intv = 1;
				if (q[x] > 3) { call_over(0, x + 1); break; } // bwt.c:295
				const int c = q[x];
				ik.x0 = pick(l2, c) + 1, ik.x2 = pick(l2n, c) - pick(l2, c), ik.x1 = pick(l2, 3 - c) + 1, ik.info = (uint64_t)x + 1; // bwt_set_intv
				i = x + 1, nc = 0;
				phase = PH_FWD;
			}
			if (phase == PH_FWD) {
				if (i < len && q[i] < 4) break; // an extension to do
				curr[nc++] = ik;                // an ambiguous base or the end of the read (bwt.c:308-312)
				end_forward();
			}
			if (phase == PH_BWD) {
				if (cb >= 0) { // an extension to do, of prev[j]
					p = prev[rev ? np - 1 - j : j];
					break;
				}
				// the start of the read or an ambiguous base ends every interval; only the first can still be new (bwt.c:327-333)
				if (nm == 0 || (uint64_t)(i + 1) < mem_start) {
					Intv m = prev[rev ? np - 1 : 0];
					m.info |= (uint64_t)(i + 1) << 32;
					mem[nm++] = m;
				}
				call_over(nm, ret);
			}
		}
		if (__builtin_amdgcn_ballot_w64(phase != PH_DONE) == 0) break;
		// ---- the extension, for every lane that has one
		Intv o{};
		if (phase == PH_FWD || phase == PH_BWD) { // ONE site, so the lanes go through it together
			const bool back = phase == PH_BWD;
			const Intv in = back ? p : ik;
			o = fm_extend(B, in, back ? cb : 3 - q[i], back);
		}
		// ---- what it means
		if (phase == PH_FWD) { // bwt.c:300-307
			bool over = false;
			if (o.x2 != ik.x2) {
				curr[nc++] = ik;
				over = o.x2 < (uint64_t)min_intv;
			}
			if (over) end_forward();
			else ik = o, ik.info = (uint64_t)i + 1, ++i;
		} else if (phase == PH_BWD) { // bwt.c:324-340
			if (o.x2 < (uint64_t)min_intv) {
				if (nc == 0 && (nm == 0 || (uint64_t)(i + 1) < mem_start)) {
					Intv m = p;
					m.info |= (uint64_t)(i + 1) << 32;
					mem[nm++] = m, mem_start = (uint64_t)(i + 1);
				}
			} else if (nc == 0 || o.x2 != last_x2) {
				Intv m = o;
				m.info = p.info;
				curr[nc++] = m, last_x2 = o.x2;
			}
			if (++j == np) { // the row is done
				if (nc == 0) call_over(nm, ret);
				else {
					{ const Stack t = curr; curr = prev, prev = t; }
					np = nc, rev = false, --i, cb = i < 0 ? -1 : q[i] < 4 ? q[i] : -1;
					j = 0, nc = 0, last_x2 = 0;
				}
			}
		}
	}
}

// bwt_sa, bwt.c:85-95
__global__ void sa_kernel(DevBwt B, const uint64_t *__restrict__ ks, long long n, uint64_t *__restrict__ pos)
{
	for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
		uint64_t k = ks[t], sa = 0;
		while (k & B.sa_mask) {
			const uint64_t x = k - (k > B.primary);
			const int c = (int)(B.bwt[((x >> 7) << 4) + 8 + ((x & 127) >> 4)] >> ((~x & 15) << 1) & 3); // bwt_B0, bwt.h:70
			++sa;
			const U4 l2{B.L2[0], B.L2[1], B.L2[2], B.L2[3]}, l2n{B.L2[1], B.L2[2], B.L2[3], B.L2[4]};
			if (k == B.primary) k = 0;                // bwt_invPsi, bwt.c:57
			else if (k == B.seq_len) k = pick(l2n, c); // bwt_occ's first special case, bwt.c:112: L2[c] + (L2[c+1] - L2[c])
			else k = pick(l2, c) + pick(fm_occ4(B, k), c);
		}
		pos[t] = sa + B.sa[k >> B.sa_shift];
	}
}

// bwt_sa of one entry (the loop of sa_kernel)
__device__ __forceinline__ uint64_t fm_sa(const DevBwt &B, uint64_t k)
{
	uint64_t sa = 0;
	while (k & B.sa_mask) {
		const uint64_t x = k - (k > B.primary);
		const int c = (int)(B.bwt[((x >> 7) << 4) + 8 + ((x & 127) >> 4)] >> ((~x & 15) << 1) & 3);
		++sa;
		const U4 l2{B.L2[0], B.L2[1], B.L2[2], B.L2[3]}, l2n{B.L2[1], B.L2[2], B.L2[3], B.L2[4]};
		if (k == B.primary) k = 0;
		else if (k == B.seq_len) k = pick(l2n, c);
		else k = pick(l2, c) + pick(fm_occ4(B, k), c);
	}
	return sa + B.sa[k >> B.sa_shift];
}

// The suffix-array entries mem_insert_seed will ask for (bwamem.c:218-225), looked up where the intervals are: one thread
// per interval the SMEM kernel wrote; an interval long and rare enough (length >= min_seed_len, x[2] <= max_occ) takes
// x[2] slots of `pos` (cursor[2]) and notes where they start in pos_base[slot], the others note UINT64_MAX.  The interval
// count is read from the SMEM kernel's cursor: no host round trip between the two kernels.
// When the SMEM kernel overflowed its output arrays (a cursor ran past its capacity) the slots behind the overflow were
// reserved but never written -- d_scratch is shared with other stages and holds stale bytes there -- and the host is going to
// retry with larger arrays anyway: nothing is looked up in that case (a stale x0/x2 would send fm_sa anywhere).
__global__ void sa_of_intervals_kernel(DevBwt B, const Intv *__restrict__ intv, const unsigned long long *__restrict__ cursors,
                                       unsigned long long call_cap, unsigned long long intv_cap, int min_seed_len, unsigned long long max_occ,
                                       uint64_t *__restrict__ pos_base, uint64_t *__restrict__ pos, unsigned long long pos_cap,
                                       unsigned long long *__restrict__ pos_cursor, int *__restrict__ overflow)
{
	if (cursors[0] > call_cap || cursors[1] > intv_cap) return; // overflowed attempt: its intervals are not all there
	const unsigned long long n = cursors[1];
	for (unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (unsigned long long)gridDim.x * blockDim.x) {
		const Intv v = intv[t];
		const int len = (int)((uint32_t)v.info - (uint32_t)(v.info >> 32));
		if (len < min_seed_len || v.x2 > max_occ) {
			pos_base[t] = ~0ull;
			continue;
		}
		const unsigned long long base = atomicAdd(pos_cursor, (unsigned long long)v.x2);
		pos_base[t] = base;
		if (base + v.x2 > pos_cap) {
			atomicExch(overflow, 1);
			continue;
		}
		for (uint64_t j = 0; j < v.x2; ++j) pos[base + j] = fm_sa(B, v.x0 + j);
	}
}

// ---- per-read order on the device.  The SMEM kernels append calls and intervals in completion order; what the caller gets is every
// read's calls in call order with their intervals behind one another (the order smem_next2 / mem_chain walk them in, bwamem.c:118-162,
// 208-243).  Putting them back was a pass over the downloaded arrays on the host -- 4 ms per 33 k-read batch, inside the device gate, a
// fifth of a seeding batch's wall time in the pipeline.  Four small kernels do it behind the SMEM kernel instead: count per read, one
// block's scan, place the calls (their sequence number within the read is in .rsv), then one lane per read lays out its intervals.
__global__ void smem_order_count(const bmh_smem_call_t *__restrict__ calls, const uint32_t *__restrict__ cread,
                                 const unsigned long long *__restrict__ cursors, unsigned long long call_cap, unsigned long long intv_cap,
                                 uint32_t *__restrict__ ccnt, uint32_t *__restrict__ icnt)
{
	if (cursors[0] > call_cap || cursors[1] > intv_cap) return; // overflowed attempt
	const unsigned long long n = cursors[0];
	for (unsigned long long c = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; c < n; c += (unsigned long long)gridDim.x * blockDim.x) {
		const uint32_t r = cread[c];
		atomicAdd(&ccnt[r], 1u);
		if (calls[c].n) atomicAdd(&icnt[r], (uint32_t)calls[c].n);
	}
}

// exclusive sums of both count arrays, n + 1 outputs each (the last one is the total); one block of 1024 threads
__global__ __launch_bounds__(1024) void smem_order_scan(const uint32_t *__restrict__ ccnt, const uint32_t *__restrict__ icnt, int n,
                                                        uint32_t *__restrict__ coff, uint64_t *__restrict__ ioff)
{
	__shared__ unsigned long long pc[1024], pi[1024];
	const int t = threadIdx.x, per = (n + 1023) / 1024, lo = min(t * per, n), hi = min(lo + per, n);
	unsigned long long sc = 0, si = 0;
	for (int k = lo; k < hi; ++k) sc += ccnt[k], si += icnt[k];
	pc[t] = sc, pi[t] = si;
	__syncthreads();
	for (int d = 1; d < 1024; d <<= 1) { // inclusive scan of the 1024 partial sums
		const unsigned long long ac = t >= d ? pc[t - d] : 0, ai = t >= d ? pi[t - d] : 0;
		__syncthreads();
		pc[t] += ac, pi[t] += ai;
		__syncthreads();
	}
	unsigned long long bc = pc[t] - sc, bi = pi[t] - si;
	for (int k = lo; k < hi; ++k) {
		coff[k] = (uint32_t)bc, ioff[k] = bi;
		bc += ccnt[k], bi += icnt[k];
	}
	if (t == 1023) coff[n] = (uint32_t)pc[1023], ioff[n] = pi[1023];
}

__global__ void smem_order_place(const bmh_smem_call_t *__restrict__ calls, const uint32_t *__restrict__ cread,
                                 const unsigned long long *__restrict__ cursors, unsigned long long call_cap, unsigned long long intv_cap,
                                 const uint32_t *__restrict__ coff, bmh_smem_call_t *__restrict__ calls2)
{
	if (cursors[0] > call_cap || cursors[1] > intv_cap) return;
	const unsigned long long n = cursors[0];
	for (unsigned long long c = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; c < n; c += (unsigned long long)gridDim.x * blockDim.x) {
		const bmh_smem_call_t cl = calls[c];
		calls2[coff[cread[c]] + cl.rsv] = cl; // .rsv = the call's sequence number within its read
	}
}

__global__ void smem_order_intervals(int n_reads, const uint32_t *__restrict__ coff, bmh_smem_call_t *__restrict__ calls2,
                                     const uint64_t *__restrict__ ioff, const Intv *__restrict__ intv, Intv *__restrict__ intv2,
                                     const uint64_t *__restrict__ pb, uint64_t *__restrict__ pb2, const unsigned long long *__restrict__ cursors,
                                     unsigned long long call_cap, unsigned long long intv_cap)
{
	if (cursors[0] > call_cap || cursors[1] > intv_cap) return;
	const int r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_reads) return;
	uint32_t local = 0;
	const uint64_t base = ioff[r];
	for (uint32_t c = coff[r]; c < coff[r + 1]; ++c) {
		bmh_smem_call_t cl = calls2[c];
		for (int k = 0; k < cl.n; ++k) {
			intv2[base + local + k] = intv[cl.first + k];
			if (pb) pb2[base + local + k] = pb[cl.first + k];
		}
		cl.first = local, cl.rsv = 0, local += (uint32_t)cl.n;
		calls2[c] = cl;
	}
}

// ---- host side: one resident copy of the index per (device, host arrays), shared by all contexts ----
struct BwtShare {
	int device;
	const uint32_t *h_bwt;
	const uint64_t *h_sa;
	void *d_bwt, *d_sa;
	DevBwt dev;
	int refs;
};
static std::mutex g_bwt_mu;
static std::vector<BwtShare> g_bwts;

} // namespace bmh

using namespace bmh;

struct bmh_bwt_binding { // hangs off the context (opaque pointer in bmh_ctx)
	DevBwt dev;
	const uint32_t *h_bwt;
};

extern "C" void free_bwt_binding(void *p) { delete (bmh_bwt_binding *)p; }

extern "C" {

int bmh_ctx_set_bwt(bmh_ctx_t *ctx, const bmh_bwt_t *b)
{
	if (!ctx || !b || !b->bwt || !b->sa || b->sa_intv < 1 || (b->sa_intv & (b->sa_intv - 1))) return BMH_E_ARG;
	BMH_HIP(ctx, hipSetDevice(ctx->device));
	std::lock_guard<std::mutex> lk(g_bwt_mu);
	if (ctx->bwt_bind && ((bmh_bwt_binding *)ctx->bwt_bind)->h_bwt == b->bwt) return BMH_OK;
	BwtShare *s = nullptr;
	for (auto &e : g_bwts)
		if (e.device == ctx->device && e.h_bwt == b->bwt && e.h_sa == b->sa) s = &e;
	if (!s) {
		BwtShare n{};
		n.device = ctx->device, n.h_bwt = b->bwt, n.h_sa = b->sa;
		const size_t bw = (size_t)b->bwt_size * 4 + 64, sb = (size_t)b->n_sa * 8 + 64;
		if (hipMalloc(&n.d_bwt, bw) != hipSuccess || hipMalloc(&n.d_sa, sb) != hipSuccess) {
			(void)hipGetLastError();
			if (n.d_bwt) (void)hipFree(n.d_bwt);
			ctx->last_error = "hipMalloc for the FM-index failed";
			return BMH_E_NOMEM;
		}
		if (hipMemcpy(n.d_bwt, b->bwt, (size_t)b->bwt_size * 4, hipMemcpyHostToDevice) != hipSuccess ||
		    hipMemcpy(n.d_sa, b->sa, (size_t)b->n_sa * 8, hipMemcpyHostToDevice) != hipSuccess) {
			(void)hipFree(n.d_bwt), (void)hipFree(n.d_sa);
			ctx->last_error = "uploading the FM-index failed";
			return BMH_E_HIP;
		}
		n.dev.primary = b->primary, n.dev.seq_len = b->seq_len;
		for (int i = 0; i < 5; ++i) n.dev.L2[i] = b->L2[i];
		n.dev.bwt = (const uint32_t *)n.d_bwt, n.dev.sa = (const uint64_t *)n.d_sa;
		n.dev.sa_mask = (uint64_t)b->sa_intv - 1, n.dev.sa_shift = __builtin_ctz((unsigned)b->sa_intv);
		n.refs = 0;
		g_bwts.push_back(n);
		s = &g_bwts.back();
	}
	++s->refs; // (kept for the life of the process: the index is as immutable as the reference)
	if (!ctx->bwt_bind) ctx->bwt_bind = new bmh_bwt_binding();
	((bmh_bwt_binding *)ctx->bwt_bind)->dev = s->dev, ((bmh_bwt_binding *)ctx->bwt_bind)->h_bwt = b->bwt;
	return BMH_OK;
}

int bmh_sa_batch(bmh_ctx_t *ctx, const uint64_t *k, int64_t n, uint64_t *pos)
{
	if (!ctx || n < 0 || (n > 0 && (!k || !pos))) return BMH_E_ARG;
	if (!ctx->bwt_bind) {
		ctx->last_error = "no FM-index on the device (bmh_ctx_set_bwt)";
		return BMH_E_ARG;
	}
	if (n == 0) return BMH_OK;
	const DevBwt &B = ((bmh_bwt_binding *)ctx->bwt_bind)->dev;
	for (int64_t i = 0; i < n; ++i)
		if (k[i] > B.seq_len) {
			ctx->last_error = "suffix-array index " + std::to_string(i) + " is beyond the index";
			return BMH_E_ARG;
		}
	GateGuard gate;
	int rc;
	BMH_HIP(ctx, hipSetDevice(ctx->device));
	if ((rc = ensure(ctx, ctx->d_tasks, (size_t)n * 8)) || (rc = ensure(ctx, ctx->d_res, (size_t)n * 8))) return rc;
	if ((rc = ensure_host(ctx, ctx->h_up, (size_t)n * 8)) || (rc = ensure_host(ctx, ctx->h_down, (size_t)n * 8))) return rc;
	memcpy(ctx->h_up.p, k, (size_t)n * 8);
	BMH_HIP(ctx, hipMemcpyAsync(ctx->d_tasks.p, ctx->h_up.p, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
	if (ctx->timing) BMH_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
	hipLaunchKernelGGL(sa_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 8192)), dim3(256), 0, ctx->stream, B,
	                   (const uint64_t *)ctx->d_tasks.p, (long long)n, (uint64_t *)ctx->d_res.p);
	BMH_HIP(ctx, hipGetLastError());
	if (ctx->timing) {
		BMH_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
		ctx->ev_valid = true;
	}
	BMH_HIP(ctx, hipMemcpyAsync(ctx->h_down.p, ctx->d_res.p, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
	BMH_HIP(ctx, stream_wait(ctx, ctx->stream));
	memcpy(pos, ctx->h_down.p, (size_t)n * 8);
	return BMH_OK;
}

// what bmh_seed_batch adds to bmh_smem_batch: the suffix-array positions of the intervals chaining will look up
struct SeedPos {
	int min_seed_len, max_occ;
	uint64_t *sa_off, *sa_pos; // sa_off[k] for interval k of the output, UINT64_MAX = never looked up
	size_t sa_cap;
	uint64_t n_pos;            // out: positions written
};

static int smem_impl(bmh_ctx_t *ctx, const bmh_smem_opt_t *o, int n_reads, const bmh_read_t *reads, uint32_t *call_off,
                     bmh_smem_call_t *calls, size_t call_cap, uint64_t *intv_off, bmh_smem_intv_t *intv, size_t intv_cap, SeedPos *sp)
{
	if (!ctx || !o || n_reads < 0 || (n_reads > 0 && (!reads || !call_off || !intv_off))) return BMH_E_ARG;
	if (o->min_emit_len < 0 || (sp && o->min_emit_len > o->min_seed_len)) {
		ctx->last_error = "bmh_smem_opt_t.min_emit_len must be >= 0 (and <= min_seed_len for bmh_seed_batch)";
		return BMH_E_ARG;
	}
	if (!ctx->bwt_bind) {
		ctx->last_error = "no FM-index on the device (bmh_ctx_set_bwt)";
		return BMH_E_ARG;
	}
	if (n_reads == 0) return BMH_OK;
	const DevBwt &B = ((bmh_bwt_binding *)ctx->bwt_bind)->dev;
	const bool trace = getenv("BMH_SMEM_TRACE") != nullptr;
	auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	double tt[6] = {now(), 0, 0, 0, 0, 0};
	int rc, lmax = 1;
	size_t bytes = 0;
	for (int r = 0; r < n_reads; ++r) {
		if (reads[r].l_seq < 0 || (reads[r].l_seq > 0 && !reads[r].seq)) return BMH_E_ARG;
		bytes += (size_t)reads[r].l_seq, lmax = std::max(lmax, reads[r].l_seq);
	}
	GateGuard gate;
	BMH_HIP(ctx, hipSetDevice(ctx->device));
	ctx->pool_resident = false;
	// host image of the input block [reads | offsets | lengths], built in pinned memory and uploaded with one copy
	const size_t in_off = (bytes + 16 + 63) & ~(size_t)63, in_len = in_off + (size_t)n_reads * 8, in_bytes = in_len + (size_t)n_reads * 4;
	if ((rc = ensure_host(ctx, ctx->h_up, in_bytes))) return rc;
	uint8_t *pool = (uint8_t *)ctx->h_up.p;
	uint64_t *off = (uint64_t *)(pool + in_off);
	int *len = (int *)(pool + in_len);
	memset(pool + bytes, 0, in_off - bytes);
	for (size_t r = 0, at = 0; r < (size_t)n_reads; ++r) {
		off[r] = at, len[r] = reads[r].l_seq;
		if (reads[r].l_seq) memcpy(pool + at, reads[r].seq, (size_t)reads[r].l_seq);
		at += (size_t)reads[r].l_seq;
	}
	const int lcap = lmax + 2;
	int lanes = 64; // reads per wave: fewer when the batch cannot fill the chip anyway (see smem_kernel)
	if (const char *e = getenv("BMH_SMEM_LANES")) lanes = atoi(e) >= 8 && atoi(e) <= 64 ? atoi(e) : 64; // (A/B knob; fewer measured slower)
	// 105 VGPRs -> 4 waves per SIMD; the 94-register build fits 5: worth it once the batch has more waves than 4 per SIMD
	// (500 k reads: 20.4 -> 18.9 ms), not below (200 k: 10.0 -> 10.3 ms)
	static const bool host_order = getenv("BMH_SMEM_HOST_ORDER") && atoi(getenv("BMH_SMEM_HOST_ORDER")) != 0; // (A/B: the per-read order made on the host)
	const bool dev_order = !host_order;
	int emit_min = 16;
	if (const char *e = getenv("BMH_SMEM_EMIT")) emit_min = atoi(e) >= 1 && atoi(e) <= 64 ? atoi(e) : 16; // (tuning knob)
	// Two kernels, same results.  smem_conv_kernel (one extension site, 46 % of the VALU lanes active against 25 %, but about
	// twice the instructions around each extension) has the shorter critical path: 8 192 reads 3.86 against 4.29 ms, 65 536
	// reads 4.80 against 5.36 ms -- the batch sizes of the pipeline.  Once the batch fills the chip several times over the
	// loop-per-lane kernel's lower instruction count wins: 1 M reads 32.1 against 38.1 ms.  BMH_SMEM_KERNEL=conv|loops forces one.
	bool conv = n_reads <= 128 * 1024;
	if (const char *e = getenv("BMH_SMEM_KERNEL")) conv = strcmp(e, "loops") != 0;
	int waves = n_reads > 4 * 1024 * 64 ? 5 : 4;
	if (const char *e = getenv("BMH_SMEM_WAVES")) waves = atoi(e) == 5 ? 5 : 4; // (A/B knob)
	int grid = (int)std::min<long long>(((long long)n_reads + lanes - 1) / lanes, 1024 * waves);
	while (grid > 1 && (size_t)grid * 3 * (size_t)lcap * 64 * sizeof(Intv) > ((size_t)6 << 30)) grid /= 2; // stacks: at most 6 GB
	// device outputs grow until everything fits (the totals are data dependent)
	// sized from the densest batch this context has seen (calls / intervals per base), so that a steady stream of
	// batches does not run the kernel twice
	size_t d_calls = std::max<size_t>((size_t)(ctx->smem_calls_per_base * 1.25 * (double)bytes) + 1024, 1024);
	size_t d_intv = std::max<size_t>((size_t)(ctx->smem_intv_per_base * 1.25 * (double)bytes) + 4096, 4096);
	size_t d_pos = sp ? std::max<size_t>((size_t)(ctx->smem_pos_per_base * 1.25 * (double)bytes) + 4096, 4096) : 0;
	if (const char *e = getenv("BMH_SMEM_INIT_CAP")) // (test knob: start with arrays this small, so that the first attempt overflows)
		if (atoll(e) > 0) d_calls = d_intv = (size_t)atoll(e), d_pos = sp ? (size_t)atoll(e) : 0;
	const bmh_smem_call_t *h_calls = nullptr; // (in the pinned download buffer)
	const uint32_t *h_read = nullptr;
	const Intv *h_intv = nullptr;
	const uint64_t *h_pb = nullptr, *h_pos = nullptr; // bmh_seed_batch: where an interval's positions start; the positions
	size_t n_calls = 0;
	unsigned long long totals[5] = {0, 0, 0, 0, 0}; // calls, intervals, (overflow flag), (read counter), positions
	if ((rc = ensure_host(ctx, ctx->h_down, 64))) return rc;
	tt[1] = now();
	for (int attempt = 0; attempt < 6; ++attempt) {
		const size_t hdr = 64, o_pool = hdr, o_off = o_pool + ((bytes + 16 + 63) & ~(size_t)63), o_len = o_off + (size_t)n_reads * 8,
		             o_calls = (o_len + (size_t)n_reads * 4 + 63) & ~(size_t)63, o_cr = o_calls + d_calls * sizeof(bmh_smem_call_t),
		             o_intv = (o_cr + d_calls * 4 + 63) & ~(size_t)63, o_pb = o_intv + d_intv * sizeof(Intv), o_pos = o_pb + (sp ? d_intv * 8 : 0),
		             // ... and the same in per-read order (smem_order_*): counts, offsets, calls, intervals, position bases
		             nr1 = ((size_t)n_reads + 1 + 15) & ~(size_t)15, o_ccnt = (o_pos + d_pos * 8 + 63) & ~(size_t)63, o_icnt = o_ccnt + nr1 * 4,
		             o_coff = o_icnt + nr1 * 4, o_ioff = o_coff + nr1 * 4, o_calls2 = o_ioff + nr1 * 8,
		             o_intv2 = (o_calls2 + d_calls * sizeof(bmh_smem_call_t) + 63) & ~(size_t)63, o_pb2 = o_intv2 + d_intv * sizeof(Intv),
		             total = o_pb2 + (sp ? d_intv * 8 : 0);
		if ((rc = ensure(ctx, ctx->d_scratch, total))) return rc;
		if ((rc = ensure(ctx, ctx->d_sw, (size_t)grid * 3 * (size_t)lcap * 64 * sizeof(Intv)))) return rc;
		uint8_t *d = (uint8_t *)ctx->d_scratch.p;
		BMH_HIP(ctx, hipMemsetAsync(d, 0, hdr, ctx->stream));
		static_assert(sizeof(uint64_t) == 8 && sizeof(int) == 4, "layout of the input block");
		BMH_HIP(ctx, hipMemcpyAsync(d + o_pool, pool, in_bytes, hipMemcpyHostToDevice, ctx->stream)); // o_off, o_len follow as in the image
		if (ctx->timing) BMH_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
#define BMH_SMEM_LAUNCH(W)                                                                                              \
	hipLaunchKernelGGL(smem_kernel<W>, dim3((unsigned)grid), dim3(64), 0, ctx->stream, B, (const uint8_t *)(d + o_pool), \
	                   (const uint64_t *)(d + o_off), (const int *)(d + o_len), n_reads, *o, (Intv *)ctx->d_sw.p, lcap,  \
	                   (bmh_smem_call_t *)(d + o_calls), (uint32_t *)(d + o_cr), (unsigned long long *)d,               \
	                   (unsigned long long)d_calls, (Intv *)(d + o_intv), (unsigned long long)d_intv, (int *)(d + 16), lanes)
#define BMH_SMEM_LAUNCH_CONV(W)                                                                                         \
	hipLaunchKernelGGL(smem_conv_kernel<W>, dim3((unsigned)grid), dim3(64), 0, ctx->stream, B, (const uint8_t *)(d + o_pool), \
	                   (const uint64_t *)(d + o_off), (const int *)(d + o_len), n_reads, *o, (Intv *)ctx->d_sw.p, lcap,  \
	                   (bmh_smem_call_t *)(d + o_calls), (uint32_t *)(d + o_cr), (unsigned long long *)d,               \
	                   (unsigned long long)d_calls, (Intv *)(d + o_intv), (unsigned long long)d_intv, (int *)(d + 16), emit_min)
		if (conv && waves == 5) BMH_SMEM_LAUNCH_CONV(5);
		else if (conv) BMH_SMEM_LAUNCH_CONV(4);
		else if (waves == 5) BMH_SMEM_LAUNCH(5);
		else BMH_SMEM_LAUNCH(4);
#undef BMH_SMEM_LAUNCH_CONV
#undef BMH_SMEM_LAUNCH
		BMH_HIP(ctx, hipGetLastError());
		if (sp) { // the positions of the intervals just written, straight behind them on the stream
			hipLaunchKernelGGL(sa_of_intervals_kernel, dim3((unsigned)std::min<size_t>((d_intv + 255) / 256, 4096)), dim3(256), 0, ctx->stream, B,
			                   (const Intv *)(d + o_intv), (const unsigned long long *)d, (unsigned long long)d_calls, (unsigned long long)d_intv, sp->min_seed_len,
			                   (unsigned long long)sp->max_occ, (uint64_t *)(d + o_pb), (uint64_t *)(d + o_pos), (unsigned long long)d_pos,
			                   (unsigned long long *)(d + 32), (int *)(d + 16));
			BMH_HIP(ctx, hipGetLastError());
		}
		if (dev_order) {
			const unsigned cb = (unsigned)std::min<size_t>((d_calls + 255) / 256, 2048);
			BMH_HIP(ctx, hipMemsetAsync(d + o_ccnt, 0, nr1 * 8, ctx->stream)); // both count arrays
			hipLaunchKernelGGL(smem_order_count, dim3(cb), dim3(256), 0, ctx->stream, (const bmh_smem_call_t *)(d + o_calls), (const uint32_t *)(d + o_cr),
			                   (const unsigned long long *)d, (unsigned long long)d_calls, (unsigned long long)d_intv, (uint32_t *)(d + o_ccnt),
			                   (uint32_t *)(d + o_icnt));
			hipLaunchKernelGGL(smem_order_scan, dim3(1), dim3(1024), 0, ctx->stream, (const uint32_t *)(d + o_ccnt), (const uint32_t *)(d + o_icnt), n_reads,
			                   (uint32_t *)(d + o_coff), (uint64_t *)(d + o_ioff));
			hipLaunchKernelGGL(smem_order_place, dim3(cb), dim3(256), 0, ctx->stream, (const bmh_smem_call_t *)(d + o_calls), (const uint32_t *)(d + o_cr),
			                   (const unsigned long long *)d, (unsigned long long)d_calls, (unsigned long long)d_intv, (const uint32_t *)(d + o_coff),
			                   (bmh_smem_call_t *)(d + o_calls2));
			hipLaunchKernelGGL(smem_order_intervals, dim3((unsigned)((n_reads + 255) / 256)), dim3(256), 0, ctx->stream, n_reads, (const uint32_t *)(d + o_coff),
			                   (bmh_smem_call_t *)(d + o_calls2), (const uint64_t *)(d + o_ioff), (const Intv *)(d + o_intv), (Intv *)(d + o_intv2),
			                   sp ? (const uint64_t *)(d + o_pb) : nullptr, (uint64_t *)(d + o_pb2), (const unsigned long long *)d,
			                   (unsigned long long)d_calls, (unsigned long long)d_intv);
			BMH_HIP(ctx, hipGetLastError());
		}
		if (ctx->timing) {
			BMH_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
			ctx->ev_valid = true;
		}
		BMH_HIP(ctx, hipMemcpyAsync(ctx->h_down.p, d, 40, hipMemcpyDeviceToHost, ctx->stream));
		BMH_HIP(ctx, stream_wait(ctx, ctx->stream));
		memcpy(totals, ctx->h_down.p, 40);
		if (sp) ctx->smem_pos_per_base = std::max(ctx->smem_pos_per_base, (double)totals[4] / (double)std::max<size_t>(bytes, 1));
		ctx->smem_calls_per_base = std::max(ctx->smem_calls_per_base, (double)totals[0] / (double)std::max<size_t>(bytes, 1));
		ctx->smem_intv_per_base = std::max(ctx->smem_intv_per_base, (double)totals[1] / (double)std::max<size_t>(bytes, 1));
		tt[2] = now();
		if (totals[0] <= d_calls && totals[1] <= d_intv && (!sp || totals[4] <= d_pos)) {
			n_calls = (size_t)totals[0];
			if (dev_order) { // everything is in its final order: offsets, calls, intervals, position bases, positions
				if (totals[0] > call_cap || totals[1] > intv_cap || (totals[0] && (!calls || (!intv && totals[1]))) || (sp && totals[4] > sp->sa_cap)) break;
				const size_t b_coff = (((size_t)n_reads + 1) * 4 + 63) & ~(size_t)63, b_ioff = ((size_t)n_reads + 1) * 8;
				const size_t b_c2 = (n_calls * sizeof(bmh_smem_call_t) + 63) & ~(size_t)63, b_i2 = (size_t)totals[1] * sizeof(Intv);
				const size_t b_p2 = sp ? (size_t)totals[1] * 8 : 0, b_ps = sp ? (size_t)totals[4] * 8 : 0;
				if ((rc = ensure_host(ctx, ctx->h_down, b_coff + b_ioff + b_c2 + b_i2 + b_p2 + b_ps + 64))) return rc;
				uint8_t *h = (uint8_t *)ctx->h_down.p;
				BMH_HIP(ctx, hipMemcpyAsync(h, d + o_coff, ((size_t)n_reads + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
				BMH_HIP(ctx, hipMemcpyAsync(h + b_coff, d + o_ioff, b_ioff, hipMemcpyDeviceToHost, ctx->stream));
				if (n_calls) BMH_HIP(ctx, hipMemcpyAsync(h + b_coff + b_ioff, d + o_calls2, n_calls * sizeof(bmh_smem_call_t), hipMemcpyDeviceToHost, ctx->stream));
				if (b_i2) BMH_HIP(ctx, hipMemcpyAsync(h + b_coff + b_ioff + b_c2, d + o_intv2, b_i2, hipMemcpyDeviceToHost, ctx->stream));
				if (b_p2) BMH_HIP(ctx, hipMemcpyAsync(h + b_coff + b_ioff + b_c2 + b_i2, d + o_pb2, b_p2, hipMemcpyDeviceToHost, ctx->stream));
				if (b_ps) BMH_HIP(ctx, hipMemcpyAsync(h + b_coff + b_ioff + b_c2 + b_i2 + b_p2, d + o_pos, b_ps, hipMemcpyDeviceToHost, ctx->stream));
				BMH_HIP(ctx, stream_wait(ctx, ctx->stream));
				tt[3] = now();
				memcpy(call_off, h, ((size_t)n_reads + 1) * 4);
				memcpy(intv_off, h + b_coff, b_ioff);
				if (n_calls) memcpy(calls, h + b_coff + b_ioff, n_calls * sizeof(bmh_smem_call_t));
				if (b_i2) memcpy(intv, h + b_coff + b_ioff + b_c2, b_i2);
				if (b_p2) memcpy(sp->sa_off, h + b_coff + b_ioff + b_c2 + b_i2, b_p2);
				if (b_ps) memcpy(sp->sa_pos, h + b_coff + b_ioff + b_c2 + b_i2 + b_p2, b_ps);
				if (sp) sp->n_pos = totals[4];
				tt[4] = now();
				if (trace)
					fprintf(stderr, "[bwamem_hip] bmh_smem_batch %d reads: prepare %.1f ms, upload+kernel %.1f ms, download %.1f ms, reorder %.1f ms\n", n_reads,
					        (tt[1] - tt[0]) * 1e3, (tt[2] - tt[1]) * 1e3, (tt[3] - tt[2]) * 1e3, (tt[4] - tt[3]) * 1e3);
				return BMH_OK;
			}
			const size_t b_calls = n_calls * sizeof(bmh_smem_call_t), b_read = (n_calls * 4 + 63) & ~(size_t)63, b_intv = (size_t)totals[1] * sizeof(Intv);
			const size_t b_pb = sp ? (size_t)totals[1] * 8 : 0, b_pos = sp ? (size_t)totals[4] * 8 : 0;
			if ((rc = ensure_host(ctx, ctx->h_down, b_calls + b_read + b_intv + b_pb + b_pos + 64))) return rc;
			uint8_t *h = (uint8_t *)ctx->h_down.p;
			h_calls = (const bmh_smem_call_t *)h, h_read = (const uint32_t *)(h + b_calls), h_intv = (const Intv *)(h + b_calls + b_read);
			h_pb = (const uint64_t *)(h + b_calls + b_read + b_intv), h_pos = (const uint64_t *)(h + b_calls + b_read + b_intv + b_pb);
			if (b_pb) BMH_HIP(ctx, hipMemcpyAsync(h + b_calls + b_read + b_intv, d + o_pb, b_pb, hipMemcpyDeviceToHost, ctx->stream));
			if (b_pos) BMH_HIP(ctx, hipMemcpyAsync(h + b_calls + b_read + b_intv + b_pb, d + o_pos, b_pos, hipMemcpyDeviceToHost, ctx->stream));
			if (n_calls) {
				BMH_HIP(ctx, hipMemcpyAsync(h, d + o_calls, b_calls, hipMemcpyDeviceToHost, ctx->stream));
				BMH_HIP(ctx, hipMemcpyAsync(h + b_calls, d + o_cr, n_calls * 4, hipMemcpyDeviceToHost, ctx->stream));
			}
			if (totals[1]) BMH_HIP(ctx, hipMemcpyAsync(h + b_calls + b_read, d + o_intv, b_intv, hipMemcpyDeviceToHost, ctx->stream));
			BMH_HIP(ctx, stream_wait(ctx, ctx->stream));
			break;
		}
		d_calls = std::max(d_calls, (size_t)totals[0] + 64), d_intv = std::max(d_intv, (size_t)totals[1] + 64);
		if (sp) d_pos = std::max(d_pos, (size_t)totals[4] + 64);
		if (attempt == 5) return BMH_E_NOMEM;
	}
	if (totals[0] > call_cap || totals[1] > intv_cap || (totals[0] && (!calls || (!intv && totals[1]))) || (sp && totals[4] > sp->sa_cap)) {
		ctx->last_error = "bmh_smem_batch: " + std::to_string(totals[0]) + " calls / " + std::to_string(totals[1]) +
		                  " intervals do not fit the caller's arrays";
		return BMH_E_CIGAR_CAP;
	}
	tt[3] = now();
	// the device appended in completion order; put every read's calls back in call order and its intervals behind one another
	std::vector<uint32_t> cnt((size_t)n_reads + 1, 0);
	for (size_t c = 0; c < n_calls; ++c) ++cnt[(size_t)h_read[c] + 1];
	for (int r = 0; r < n_reads; ++r) cnt[(size_t)r + 1] += cnt[(size_t)r];
	for (int r = 0; r <= n_reads; ++r) call_off[r] = cnt[(size_t)r];
	for (size_t c = 0; c < n_calls; ++c) calls[cnt[h_read[c]] + h_calls[c].rsv] = h_calls[c]; // rsv = sequence number in the read
	uint64_t used = 0;
	for (int r = 0; r < n_reads; ++r) {
		intv_off[r] = used;
		uint64_t local = 0;
		for (uint32_t c = call_off[r]; c < call_off[r + 1]; ++c) {
			bmh_smem_call_t &cl = calls[c];
			if (cl.n) memcpy(&intv[used + local], &h_intv[cl.first], (size_t)cl.n * sizeof(Intv));
			if (cl.n && sp) memcpy(&sp->sa_off[used + local], &h_pb[cl.first], (size_t)cl.n * 8);
			cl.first = (uint32_t)local, cl.rsv = 0, local += (uint64_t)cl.n;
		}
		used += local;
	}
	intv_off[n_reads] = used;
	if (sp) {
		if (totals[4]) memcpy(sp->sa_pos, h_pos, (size_t)totals[4] * 8);
		sp->n_pos = totals[4];
	}
	tt[4] = now();
	if (trace)
		fprintf(stderr, "[bwamem_hip] bmh_smem_batch %d reads: prepare %.1f ms, upload+kernel %.1f ms, download %.1f ms, reorder %.1f ms\n", n_reads,
		        (tt[1] - tt[0]) * 1e3, (tt[2] - tt[1]) * 1e3, (tt[3] - tt[2]) * 1e3, (tt[4] - tt[3]) * 1e3);
	return BMH_OK;
}

int bmh_smem_batch(bmh_ctx_t *ctx, const bmh_smem_opt_t *o, int n_reads, const bmh_read_t *reads, uint32_t *call_off,
                   bmh_smem_call_t *calls, size_t call_cap, uint64_t *intv_off, bmh_smem_intv_t *intv, size_t intv_cap)
{
	return smem_impl(ctx, o, n_reads, reads, call_off, calls, call_cap, intv_off, intv, intv_cap, nullptr);
}

int bmh_seed_batch(bmh_ctx_t *ctx, const bmh_smem_opt_t *o, int max_occ, int n_reads, const bmh_read_t *reads, uint32_t *call_off,
                   bmh_smem_call_t *calls, size_t call_cap, uint64_t *intv_off, bmh_smem_intv_t *intv, size_t intv_cap, uint64_t *sa_off,
                   uint64_t *sa_pos, size_t sa_cap, uint64_t *n_pos)
{
	if (!o || max_occ < 0 || (n_reads > 0 && (!sa_off || (!sa_pos && sa_cap)))) return BMH_E_ARG;
	SeedPos sp{o->min_seed_len, max_occ, sa_off, sa_pos, sa_cap, 0};
	const int rc = smem_impl(ctx, o, n_reads, reads, call_off, calls, call_cap, intv_off, intv, intv_cap, &sp);
	if (n_pos) *n_pos = sp.n_pos;
	return rc;
}

} // extern "C"
