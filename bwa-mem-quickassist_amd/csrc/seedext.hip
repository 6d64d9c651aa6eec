// seedext.hip -- the fused per-seed extension record: what mem_chain2aln does for ONE seed between
// bwamem.c:810 and :866 of the reference, for a whole batch of seeds, without returning to the host in between.
//
// The fork sketched this record as ext_param_t / ext_res_t (bwamem.c:553-577: both flanks of a seed in, the finished
// region out) and never used it.  On the device it is FOUR dependent rounds of ksw_extend2 batches, each one a run of the
// length-sorted lane-per-task kernels of extend_dispatch.hip over a task list that the previous round's results
// produced -- built by the small kernels below, on the device:
//
//   seed_left_make    left task of every seed with qbeg > 0 (reversed flanks, h0 = len*a, end_bonus = pen_clip5,
//                     band w), bwamem.c:810-826; seeds without a left flank get score = truesc = len*a (:839)
//   round L1          ksw_extend2 x (#left tasks)
//   seed_try<LEFT>    the retry rule of bwamem.c:828 (prev = -1, so: max_off >= 3w/4) -> compact list of tasks at 2w
//   round L2          ksw_extend2 x (#retries), usually a handful
//   seed_right_make   clip-or-reach-the-end decision of the left side (:831-837) from the LAST try's outputs; right
//                     task (forward flanks, h0 = sc0 = the left score, end_bonus = pen_clip3), :841-854; seeds that
//                     end at the read end are finished here (:866)
//   round R1          ksw_extend2 x (#right tasks)
//   seed_try<RIGHT>   retry rule of :856 (prev = sc0): score != sc0 && max_off >= 3w/4
//   round R2
//   seed_finish       the right side's decision (:859-865), a->w = max(aw0, aw1) (:875)
//
// Lists are appended with one atomic per wave (ballot + mbcnt); the dispatcher reads their lengths from device memory,
// so the host never learns them and never waits.
#include <algorithm>

#include "bmh_ctx.h"
#include "bmh_device.h"

namespace bmh {

struct SeedP { // the driver-level fields of bmh_params_t
	int a, w, pen_clip5, pen_clip3, max_mat;
};

struct SeedState { // what the left side leaves for the right side and the end (24 bytes)
	int qb, rbl, truesc, sc0, aw0;
	uint32_t r2; // 1 + index of the seed's retry task in T2/X2, 0 = no retry
};

__device__ __forceinline__ uint32_t wave_append(bool pred, uint32_t *counter)
{
	const unsigned long long m = __builtin_amdgcn_ballot_w64(pred);
	if (m == 0) return 0;
	const int leader = __builtin_ctzll(m);
	const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
	uint32_t base = 0;
	if (lane == leader) base = atomicAdd(counter, (uint32_t)__builtin_popcountll(m));
	base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
	return base + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
}

__device__ __forceinline__ void store_task(bmh_ext_task_t *t, uint64_t q_off, uint64_t t_off, int qlen, int tlen, int h0,
                                           int w, int end_bonus, unsigned flags)
{
	uint4 *p = (uint4 *)t;
	p[0] = make_uint4((uint32_t)q_off, (uint32_t)(q_off >> 32), (uint32_t)t_off, (uint32_t)(t_off >> 32));
	p[1] = make_uint4((uint32_t)qlen | (uint32_t)tlen << 16, (uint32_t)h0, (uint32_t)(w & 0xffff) | (uint32_t)end_bonus << 16, flags);
}

__global__ __launch_bounds__(256) void seed_left_make(const bmh_seed_task_t *__restrict__ S, long long n, SeedP sp,
                                                      bmh_ext_task_t *__restrict__ T, uint32_t *__restrict__ L,
                                                      uint32_t *__restrict__ cnt, SeedState *__restrict__ ST,
                                                      int *__restrict__ err_flag)
{
	const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
	bool has = false;
	if (i < n) {
		const bmh_seed_task_t s = S[i];
		const int rq = s.l_query - s.qbeg - s.len, rt = s.wlen - s.rbeg - s.len;
		if (s.qbeg < 0 || s.len <= 0 || rq < 0 || s.rbeg < 0 || rt < 0 || s.qbeg > 65535 || rq > 65535 || s.rbeg > 65535 || rt > 65535 ||
		    (long long)s.l_query * max(sp.max_mat, sp.a) > kScoreLimit) {
			atomicExch(err_flag, BMH_E_RANGE);
			SeedState st = {0, 0, -1, -1, sp.w, 0xffffffffu}; // poisoned: later stages skip the seed
			ST[i] = st;
		} else if (s.qbeg > 0) { // bwamem.c:810-826
			const uint64_t tl = (uint64_t)s.rbeg;
			store_task(&T[i], s.q_off + (uint64_t)(s.qbeg - 1), tl > 0 ? s.t_off + tl - 1 : s.t_off, s.qbeg, s.rbeg, s.len * sp.a, sp.w,
			           sp.pen_clip5, BMH_F_QREV | BMH_F_TREV | (s.flags & BMH_F_TPAC));
			SeedState st = {0, 0, -1, -1, sp.w, 0};
			ST[i] = st;
			has = true;
		} else { // bwamem.c:839
			SeedState st = {0, 0, s.len * sp.a, s.len * sp.a, sp.w, 0};
			ST[i] = st;
		}
	}
	const uint32_t pos = wave_append(has, &cnt[0]);
	if (has) L[pos] = (uint32_t)i;
}

// the band-doubling rule (bwamem.c:828 for the left side, :856 for the right): the tasks that must run again at 2w
template <bool RIGHT>
__global__ __launch_bounds__(256) void seed_try(const bmh_seed_task_t *__restrict__ S, long long n, SeedP sp,
                                                const bmh_ext_task_t *__restrict__ T, const bmh_ext_result_t *__restrict__ X,
                                                bmh_ext_task_t *__restrict__ T2, uint32_t *__restrict__ cnt,
                                                SeedState *__restrict__ ST)
{
	const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
	bool again = false;
	if (i < n && ST[i].r2 != 0xffffffffu) {
		const bmh_seed_task_t s = S[i];
		const bool has = RIGHT ? s.qbeg + s.len != s.l_query : s.qbeg > 0;
		if (has) {
			const bmh_ext_result_t x = X[i];
			const int prev = RIGHT ? ST[i].sc0 : -1;
			again = !(x.score == prev || x.max_off < (sp.w >> 1) + (sp.w >> 2));
		}
	}
	const uint32_t pos = wave_append(again, &cnt[RIGHT ? 3 : 1]);
	if (again) {
		const uint4 *src = (const uint4 *)&T[i];
		uint4 a = src[0], b = src[1];
		b.z = (b.z & 0xffff0000u) | (uint32_t)((sp.w << 1) & 0xffff);
		uint4 *dst = (uint4 *)&T2[pos];
		dst[0] = a, dst[1] = b;
		ST[i].r2 = pos + 1;
	}
}

__global__ __launch_bounds__(256) void seed_right_make(const bmh_seed_task_t *__restrict__ S, long long n, SeedP sp,
                                                       const bmh_ext_result_t *__restrict__ X, const bmh_ext_result_t *__restrict__ X2,
                                                       bmh_ext_task_t *__restrict__ T, uint32_t *__restrict__ L,
                                                       uint32_t *__restrict__ cnt, SeedState *__restrict__ ST,
                                                       bmh_seed_result_t *__restrict__ R)
{
	const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
	bool has = false;
	if (i < n) {
		const bmh_seed_task_t s = S[i];
		SeedState st = ST[i];
		if (st.r2 == 0xffffffffu) {
			bmh_seed_result_t r = {0, 0, 0, 0, INT32_MIN, INT32_MIN, 0, 0};
			R[i] = r;
		} else {
			int n_ext = 0;
			if (s.qbeg > 0) { // the left side's outcome: outputs of the LAST try, bwamem.c:831-837
				const bmh_ext_result_t x = st.r2 ? X2[st.r2 - 1] : X[i];
				n_ext = st.r2 ? 2 : 1;
				st.aw0 = st.r2 ? sp.w << 1 : sp.w;
				st.sc0 = x.score;
				if (x.gscore <= 0 || x.gscore <= x.score - sp.pen_clip5) st.qb = s.qbeg - x.qle, st.rbl = x.tle, st.truesc = x.score;
				else st.qb = 0, st.rbl = x.gtle, st.truesc = x.gscore;
			}
			st.r2 = 0;
			const int qe = s.qbeg + s.len;
			if (qe != s.l_query) { // bwamem.c:841-854
				store_task(&T[i], s.q_off + (uint64_t)qe, s.t_off + (uint64_t)(s.rbeg + s.len), s.l_query - qe, s.wlen - s.rbeg - s.len,
				           st.sc0, sp.w, sp.pen_clip3, s.flags & BMH_F_TPAC);
				has = true;
				R[i].n_ext = n_ext;
			} else { // bwamem.c:866
				bmh_seed_result_t r = {st.qb, s.l_query, s.rbeg - st.rbl, s.rbeg + s.len, st.sc0, st.truesc, max(st.aw0, sp.w), n_ext};
				R[i] = r;
			}
			ST[i] = st;
		}
	}
	const uint32_t pos = wave_append(has, &cnt[2]);
	if (has) L[pos] = (uint32_t)i;
}

__global__ __launch_bounds__(256) void seed_finish(const bmh_seed_task_t *__restrict__ S, long long n, SeedP sp,
                                                   const bmh_ext_result_t *__restrict__ X, const bmh_ext_result_t *__restrict__ X2,
                                                   const SeedState *__restrict__ ST, bmh_seed_result_t *__restrict__ R)
{
	const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	const bmh_seed_task_t s = S[i];
	const SeedState st = ST[i];
	const int qe = s.qbeg + s.len;
	if (st.r2 == 0xffffffffu || qe == s.l_query) return; // finished by seed_right_make
	const bmh_ext_result_t x = st.r2 ? X2[st.r2 - 1] : X[i];
	const int aw1 = st.r2 ? sp.w << 1 : sp.w;
	bmh_seed_result_t r;
	r.qb = st.qb, r.rb = s.rbeg - st.rbl, r.score = x.score, r.w = max(st.aw0, aw1), r.n_ext = R[i].n_ext + (st.r2 ? 2 : 1);
	if (x.gscore <= 0 || x.gscore <= x.score - sp.pen_clip3) // bwamem.c:859-865
		r.qe = qe + x.qle, r.re = s.rbeg + s.len + x.tle, r.truesc = st.truesc + (x.score - st.sc0);
	else r.qe = s.l_query, r.re = s.rbeg + s.len + x.gtle, r.truesc = st.truesc + (x.gscore - st.sc0);
	R[i] = r;
}

struct SeedWs {
	bmh_ext_task_t *T, *T2;
	bmh_ext_result_t *X, *X2;
	uint32_t *L, *cnt;
	SeedState *ST;
};

static int seed_workspace(bmh_ctx *ctx, int64_t n, SeedWs *w)
{
	const size_t N = (size_t)n, a256 = 255;
	size_t off = 0;
	auto take = [&](size_t bytes) {
		const size_t o = off;
		off = (off + bytes + a256) & ~a256;
		return o;
	};
	const size_t oc = take(64), oT = take(N * 32), oT2 = take(N * 32), oX = take(N * 24), oX2 = take(N * 24), oL = take(N * 4),
	             oS = take(N * sizeof(SeedState));
	int rc = ensure(ctx, ctx->d_seedws, off);
	if (rc) return rc;
	uint8_t *b = (uint8_t *)ctx->d_seedws.p;
	w->cnt = (uint32_t *)(b + oc), w->T = (bmh_ext_task_t *)(b + oT), w->T2 = (bmh_ext_task_t *)(b + oT2);
	w->X = (bmh_ext_result_t *)(b + oX), w->X2 = (bmh_ext_result_t *)(b + oX2), w->L = (uint32_t *)(b + oL), w->ST = (SeedState *)(b + oS);
	return BMH_OK;
}

// everything is enqueued on ctx->stream; the list lengths stay on the device (ws.cnt[0..3]: left, left retries, right,
// right retries -- read back by the host-buffer entry point for its statistics only)
int launch_seedext(bmh_ctx *ctx, const uint8_t *d_pool, const bmh_seed_task_t *d_tasks, int64_t n, bmh_seed_result_t *d_res,
                   int qmax)
{
	if (n <= 0) return BMH_OK;
	if (ctx->params.w < 1 || (ctx->params.w << 1) > 32767) {
		ctx->last_error = "the fused per-seed extension needs 1 <= w and 2*w <= 32767";
		return BMH_E_RANGE;
	}
	SeedWs ws;
	int rc = seed_workspace(ctx, n, &ws);
	if (rc) return rc;
	const SeedP sp = {ctx->params.a, ctx->params.w, ctx->params.pen_clip5, ctx->params.pen_clip3, ctx->dev.max_mat};
	const unsigned grid = (unsigned)((n + 255) / 256);
	hipStream_t s = ctx->stream;
	const bool tm = ctx->timing;
#define BMH_SROUND(k)                                                   \
	do {                                                                \
		if (tm) BMH_HIP(ctx, hipEventRecord(ctx->ev_sround[k], s));     \
	} while (0)
	BMH_HIP(ctx, hipMemsetAsync(ws.cnt, 0, 64, s));
	BMH_SROUND(0);
	hipLaunchKernelGGL(seed_left_make, dim3(grid), dim3(256), 0, s, d_tasks, (long long)n, sp, ws.T, ws.L, ws.cnt, ws.ST, ctx->d_err);
	if ((rc = launch_extend(ctx, d_pool, ws.T, n, ws.X, ws.L, qmax, ws.cnt + 0, 1))) return rc;
	BMH_SROUND(1);
	hipLaunchKernelGGL(seed_try<false>, dim3(grid), dim3(256), 0, s, d_tasks, (long long)n, sp, ws.T, ws.X, ws.T2, ws.cnt, ws.ST);
	if ((rc = launch_extend(ctx, d_pool, ws.T2, n, ws.X2, nullptr, qmax, ws.cnt + 1, 2))) return rc;
	BMH_SROUND(2);
	hipLaunchKernelGGL(seed_right_make, dim3(grid), dim3(256), 0, s, d_tasks, (long long)n, sp, ws.X, ws.X2, ws.T, ws.L, ws.cnt, ws.ST,
	                   d_res);
	if ((rc = launch_extend(ctx, d_pool, ws.T, n, ws.X, ws.L, qmax, ws.cnt + 2, 3))) return rc;
	BMH_SROUND(3);
	hipLaunchKernelGGL(seed_try<true>, dim3(grid), dim3(256), 0, s, d_tasks, (long long)n, sp, ws.T, ws.X, ws.T2, ws.cnt, ws.ST);
	if ((rc = launch_extend(ctx, d_pool, ws.T2, n, ws.X2, nullptr, qmax, ws.cnt + 3, 4))) return rc;
	hipLaunchKernelGGL(seed_finish, dim3(grid), dim3(256), 0, s, d_tasks, (long long)n, sp, ws.X, ws.X2, ws.ST, d_res);
	BMH_SROUND(4);
#undef BMH_SROUND
	if (tm) ctx->ev_sround_valid = true;
	BMH_HIP(ctx, hipGetLastError());
	return BMH_OK;
}

const uint32_t *seedext_counters(const bmh_ctx *ctx) { return (const uint32_t *)ctx->d_seedws.p; }

} // namespace bmh
