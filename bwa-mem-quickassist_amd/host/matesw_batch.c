/*
 * matesw_batch.c -- batched mate rescue: the loop of mem_sam_pe (reference bwa-0.7.8/bwamem_pair.c:251-263) over
 * mem_matesw (bwamem_pair.c:109-175) for a whole chunk of read pairs, its ksw_align2 calls run as GPU batches.
 *
 * What is sequential in the reference stays sequential here: every mem_matesw invocation first tests the four
 * orientations against the CURRENT content of the mate's region vector (:112-121), which earlier invocations of the
 * same pair may have changed.  So each pair is a small resumable machine {end i, hit j}; a ROUND plans, per unfinished
 * pair, its next few invocations that need Smith-Waterman (up to four ksw_align2 calls each: skip[] is fixed at
 * :112-121), runs all of them in one bmh_sw_batch, and folds the results in the reference's order (:150-166 insert,
 * :168 mem_sort_and_dedup after every orientation), re-deriving skip[] from the then-current vector before each
 * invocation.  The first planned invocation of a pair is never speculative; the ones planned ahead may turn out to be
 * skipped (their ksw_align2 results are then simply not used) -- the outcome is exactly the reference's.
 *
 * Sequences: the pool holds the reads of the pairs that need rescue, once each.  A reverse-complemented mate (:130-133) is BMH_F_QREV|BMH_F_QCOMP; with
 * the reference resident on the device the window bns_get_seq would return (:143) is a BMH_F_TPAC task, otherwise it
 * is decoded on the host into the round's pool.
 */
#include <stdlib.h>
#include <string.h>

#include "../../include/bwamem_hip.h"

const bmh_params_t *bmh_ctx_params_(const bmh_ctx_t *ctx);
int bmh_ctx_has_pac_(const bmh_ctx_t *ctx, const uint8_t *pac, int64_t l_pac);
void bmh_ctx_set_driver_stats_(bmh_ctx_t *ctx, const bmh_driver_stats_t *st);

enum { LOOKAHEAD = 8 }; /* invocations planned per pair and round */

typedef struct { /* one planned mem_matesw invocation: hit j of end i against the mate !i */
	int i, j;
	int plan[4]; /* per orientation: 0 not computed (was skipped when planned), -2 call on an empty window, -3 no call
	                (window inverted / bridging), > 0 index+1 of its ksw_align2 result */
	int64_t rb[4], re[4];
} inv_t;

typedef struct {
	bmh_alnreg_v b[2];  /* hits of each end within pen_unpaired of its best, copied up front (bwamem_pair.c:252-257) */
	int i, j;           /* next invocation to fold */
	int n;              /* sum of mem_matesw's return values */
	int done;
	int n_inv;
	inv_t inv[LOOKAHEAD];
} pair_t;

/* mem_infer_dir, bwamem_pair.c:23-30 */
static int infer_dir(int64_t l_pac, int64_t b1, int64_t b2, int64_t *dist)
{
	const int r1 = b1 >= l_pac, r2 = b2 >= l_pac;
	const int64_t p2 = r1 == r2 ? b2 : (l_pac << 1) - 1 - b2;
	*dist = p2 > b1 ? p2 - b1 : b1 - p2;
	return (r1 == r2 ? 0 : 1) ^ (p2 > b1 ? 0 : 3);
}

static void push_reg(bmh_alnreg_v *v, const bmh_alnreg_t *x) /* kv_push, kvec.h:68-74 */
{
	if (v->n == v->m) {
		v->m = v->m ? v->m << 1 : 2;
		v->a = (bmh_alnreg_t *)realloc(v->a, sizeof(bmh_alnreg_t) * v->m);
	}
	v->a[v->n++] = *x;
}

static void fetch_window(int64_t l_pac, const uint8_t *pac, int64_t beg, int64_t end, uint8_t *dst) /* bntseq.c:355-376 */
{
	int64_t k, l = 0;
	if (beg >= l_pac) {
		const int64_t lo = (l_pac << 1) - 1 - end, hi = (l_pac << 1) - 1 - beg;
		for (k = hi; k > lo; --k) dst[l++] = (uint8_t)(3 - (pac[k >> 2] >> ((~k & 3) << 1) & 3));
	} else
		for (k = beg; k < end; ++k) dst[l++] = (uint8_t)(pac[k >> 2] >> ((~k & 3) << 1) & 3);
}

int bmh_matesw_batch(bmh_ctx_t *ctx, int64_t l_pac, const uint8_t *pac, int n_pairs, const bmh_read_t *reads,
                     bmh_alnreg_v *regs, const bmh_pestat_t pes[4], const bmh_matesw_opt_t *o, bmh_dedup_fn dedup,
                     void *dedup_user, int *n_sw)
{
	const bmh_params_t *P;
	pair_t *ps = 0;
	uint64_t *read_off = 0; /* pool offsets of the two reads of each ACTIVE pair */
	int *act = 0, n_act = 0, q;
	uint8_t *pool = 0;
	bmh_sw_task_t *tasks = 0;
	bmh_sw_result_t *res = 0;
	size_t reads_bytes = 0, pool_cap = 0, task_cap = 0;
	int p, r, rc = BMH_OK, tpac, first_round = 1;
	bmh_driver_stats_t st;
	memset(&st, 0, sizeof(st));

	if (!ctx || !pac || !reads || !regs || !pes || !o || !dedup || n_pairs < 0 || l_pac <= 0) return BMH_E_ARG;
	if (!(P = bmh_ctx_params_(ctx))) return BMH_E_ARG;
	if (n_pairs == 0) return BMH_OK;
	tpac = bmh_ctx_has_pac_(ctx, pac, l_pac);
	/* Most pairs need no rescue at all: every candidate hit already has a properly placed mate, so each of its
	 * mem_matesw calls returns at bwamem_pair.c:122 and nothing ever changes.  Whether that is so can be read off the
	 * vectors as phase 1 left them (the first call that does NOT return there is the first that could change anything),
	 * so only the other pairs -- a few per cent -- get a machine, copies of their candidate hits, and room in the pool. */
	act = (int *)malloc(sizeof(int) * (size_t)n_pairs);
	if (!act) { rc = BMH_E_NOMEM; goto done; }
	for (p = 0; p < n_pairs; ++p) {
		int i, busy = 0;
		for (i = 0; i < 2 && !busy; ++i) {
			const bmh_alnreg_v *a = &regs[2 * p + i], *ma = &regs[2 * p + !i];
			size_t j, k;
			int nb = 0;
			for (j = 0; j < a->n && !busy; ++j) { /* the hits copied to b[i] (:252-257), the first max_matesw of them (:258-259) */
				int skip[4];
				if (a->a[j].score < a->a[0].score - o->pen_unpaired) continue;
				if (nb++ >= o->max_matesw) break;
				for (r = 0; r < 4; ++r) skip[r] = pes[r].failed ? 1 : 0;
				for (k = 0; k < ma->n; ++k) {
					int64_t dist;
					r = infer_dir(l_pac, a->a[j].rb, ma->a[k].rb, &dist);
					if (dist >= pes[r].low && dist <= pes[r].high) skip[r] = 1;
				}
				if (skip[0] + skip[1] + skip[2] + skip[3] != 4) busy = 1;
			}
		}
		if (busy) act[n_act++] = p;
	}
	if (n_sw) memset(n_sw, 0, sizeof(int) * (size_t)n_pairs);
	if (n_act == 0) goto done;
	ps = (pair_t *)calloc((size_t)n_act, sizeof(pair_t));
	read_off = (uint64_t *)malloc(sizeof(uint64_t) * 2 * (size_t)n_act);
	if (!ps || !read_off) { rc = BMH_E_NOMEM; goto done; }
	for (q = 0; q < n_act; ++q) {
		int i;
		size_t j;
		p = act[q];
		for (i = 0; i < 2; ++i) {
			const bmh_alnreg_v *a = &regs[2 * p + i];
			if (reads[2 * p + i].l_seq < 1 || reads[2 * p + i].l_seq > 65535) { rc = BMH_E_RANGE; goto done; }
			read_off[2 * q + i] = reads_bytes, reads_bytes += (size_t)reads[2 * p + i].l_seq;
			for (j = 0; j < a->n; ++j) /* bwamem_pair.c:252-257 */
				if (a->a[j].score >= a->a[0].score - o->pen_unpaired) push_reg(&ps[q].b[i], &a->a[j]);
		}
	}

	for (;;) {
		size_t n_tasks = 0, win_bytes = 0, used, want_tasks = 0;
		int active = 0;
		/* ---- plan: from each unfinished pair's cursor on, the next invocations that (as things stand) need ksw_align2.
		 * The first of them is planned against exactly the state it will be folded in; the later ones are planned AHEAD
		 * against the current state of the mate's vector, which earlier folds may still change -- ksw_align2 is pure and
		 * its inputs (hit, orientation, mate) do not depend on that state, so a result computed ahead is THE result; the
		 * fold below re-derives skip[] and only uses what it then really needs.  Without this, a pair with h candidate
		 * hits would cost h GPU round trips. */
		for (q = 0; q < n_act; ++q) {
			pair_t *s = &ps[q];
			int ii, jj;
			p = act[q];
			if (s->done) continue;
			s->n_inv = 0;
			for (ii = s->i, jj = s->j; ii < 2 && s->n_inv < LOOKAHEAD;) {
				const bmh_alnreg_t *a;
				const bmh_alnreg_v *ma;
				inv_t *e;
				int l_ms, skip[4];
				size_t k;
				if (!((size_t)jj < s->b[ii].n && jj < o->max_matesw)) { ++ii, jj = 0; continue; } /* :258-259 */
				a = &s->b[ii].a[jj], ma = &regs[2 * p + !ii], l_ms = reads[2 * p + !ii].l_seq;
				for (r = 0; r < 4; ++r) skip[r] = pes[r].failed ? 1 : 0; /* :112-121 */
				for (k = 0; k < ma->n; ++k) {
					int64_t dist;
					r = infer_dir(l_pac, a->rb, ma->a[k].rb, &dist);
					if (dist >= pes[r].low && dist <= pes[r].high) skip[r] = 1;
				}
				if (skip[0] + skip[1] + skip[2] + skip[3] == 4) { ++jj; continue; } /* :122, returns 0 */
				e = &s->inv[s->n_inv++];
				e->i = ii, e->j = jj;
				for (r = 0; r < 4; ++r) { /* :123-142 */
					int is_rev, is_larger;
					int64_t rb, re;
					e->plan[r] = 0;
					if (skip[r]) continue;
					is_rev = (r >> 1 != (r & 1)), is_larger = !(r >> 1);
					if (!is_rev) {
						rb = is_larger ? a->rb + pes[r].low : a->rb - pes[r].high;
						re = (is_larger ? a->rb + pes[r].high : a->rb - pes[r].low) + l_ms;
					} else {
						rb = (is_larger ? a->rb + pes[r].low : a->rb - pes[r].high) - l_ms;
						re = is_larger ? a->rb + pes[r].high : a->rb - pes[r].low;
					}
					if (rb < 0) rb = 0;
					if (re > l_pac << 1) re = l_pac << 1;
					e->rb[r] = rb, e->re[r] = re;
					/* bns_get_seq hands back re-rb bases unless the interval is inverted or bridges the two strands
					 * (bntseq.c:358-375); only then does mem_matesw call ksw_align2 (:144).  An empty window is still a call:
					 * it scores 0, inserts nothing and counts (no GPU work) */
					if (re == rb) e->plan[r] = -2;
					else if (re > rb && (rb >= l_pac || re <= l_pac)) e->plan[r] = 1, ++want_tasks, win_bytes += (size_t)(re - rb);
					else e->plan[r] = -3;
				}
				++jj;
			}
			++active;
		}
		if (!active) break;

		/* ---- build the round's tasks (and, without a resident reference, its windows) */
		if (want_tasks > task_cap) {
			task_cap = want_tasks + want_tasks / 2 + 64;
			free(tasks), free(res);
			tasks = (bmh_sw_task_t *)malloc(sizeof(bmh_sw_task_t) * task_cap);
			res = (bmh_sw_result_t *)malloc(sizeof(bmh_sw_result_t) * task_cap);
			if (!tasks || !res) { rc = BMH_E_NOMEM; goto done; }
		}
		used = reads_bytes;
		if (first_round || !tpac) {
			const size_t need_bytes = reads_bytes + (tpac ? 0 : win_bytes) + 16;
			if (need_bytes > pool_cap) {
				pool_cap = need_bytes + need_bytes / 2;
				free(pool);
				if (!(pool = (uint8_t *)malloc(pool_cap))) { rc = BMH_E_NOMEM; goto done; }
				first_round = 1;
			}
			if (first_round)
				for (q = 0; q < n_act; ++q) {
					memcpy(pool + read_off[2 * q], reads[2 * act[q]].seq, (size_t)reads[2 * act[q]].l_seq);
					memcpy(pool + read_off[2 * q + 1], reads[2 * act[q] + 1].seq, (size_t)reads[2 * act[q] + 1].l_seq);
				}
		}
		for (q = 0; q < n_act; ++q) {
			pair_t *s = &ps[q];
			int v;
			p = act[q];
			if (s->done) continue;
			for (v = 0; v < s->n_inv; ++v) {
				inv_t *e = &s->inv[v];
				const int mate = 2 * q + !e->i, l_ms = reads[2 * p + !e->i].l_seq; /* `mate` indexes read_off */
				for (r = 0; r < 4; ++r) {
					bmh_sw_task_t *t;
					const int is_rev = (r >> 1 != (r & 1));
					if (e->plan[r] <= 0) continue;
					t = &tasks[n_tasks];
					memset(t, 0, sizeof(*t));
					t->qlen = (uint16_t)l_ms, t->tlen = (uint32_t)(e->re[r] - e->rb[r]);
					t->q_off = is_rev ? read_off[mate] + (uint64_t)l_ms - 1 : read_off[mate]; /* :130-133 without the copy */
					t->flags = is_rev ? BMH_F_QREV | BMH_F_QCOMP : 0;
					if (tpac) t->t_off = (uint64_t)e->rb[r], t->flags |= BMH_F_TPAC;
					else {
						fetch_window(l_pac, pac, e->rb[r], e->re[r], pool + used);
						t->t_off = used, used += (size_t)(e->re[r] - e->rb[r]);
					}
					t->xtra = BMH_SW_XSUBO | BMH_SW_XSTART | (l_ms * P->a < 250 ? BMH_SW_XBYTE : 0) | (uint32_t)(o->min_seed_len * P->a); /* :147 */
					e->plan[r] = (int)n_tasks + 1; /* 1-based index of its result */
					++n_tasks;
				}
			}
		}
		if (n_tasks) {
			++st.rounds, st.ext_tasks += (int64_t)n_tasks; /* here: GPU rounds and ksw_align2 calls */
			memset(pool + used, 0, 16);
			if (first_round || !tpac) {
				st.pool_bytes += (int64_t)used + 16;
				if ((rc = bmh_upload_pool(ctx, pool, used + 16))) goto done;
				first_round = 0;
			}
			if ((rc = bmh_sw_batch(ctx, 0, 0, tasks, (int64_t)n_tasks, res))) goto done;
		}

		/* ---- fold, invocation by invocation in the reference's order (:109-175), as far as the planned results reach */
		for (q = 0; q < n_act; ++q) {
			pair_t *s = &ps[q];
			p = act[q];
			if (s->done) continue;
			for (;;) {
				const bmh_alnreg_t *a;
				bmh_alnreg_v *ma;
				const inv_t *e = 0;
				int skip[4], n = 0, l_ms, v, ok = 1;
				size_t k;
				while (s->i < 2 && !((size_t)s->j < s->b[s->i].n && s->j < o->max_matesw)) ++s->i, s->j = 0; /* :258-259 */
				if (s->i == 2) { s->done = 1; break; }
				a = &s->b[s->i].a[s->j], ma = &regs[2 * p + !s->i], l_ms = reads[2 * p + !s->i].l_seq;
				for (r = 0; r < 4; ++r) skip[r] = pes[r].failed ? 1 : 0; /* :112-121, against the vector as it is NOW */
				for (k = 0; k < ma->n; ++k) {
					int64_t dist;
					r = infer_dir(l_pac, a->rb, ma->a[k].rb, &dist);
					if (dist >= pes[r].low && dist <= pes[r].high) skip[r] = 1;
				}
				if (skip[0] + skip[1] + skip[2] + skip[3] == 4) { ++s->j; continue; } /* :122, returns 0 */
				for (v = 0; v < s->n_inv; ++v)
					if (s->inv[v].i == s->i && s->inv[v].j == s->j) e = &s->inv[v];
				if (!e) break; /* beyond this round's plan */
				for (r = 0; r < 4; ++r)
					if (!skip[r] && e->plan[r] == 0) ok = 0; /* an orientation that was skipped when planned is needed after all */
				if (!ok) break;    /* (a dedup removed the region that covered it): planned afresh in the next round */
				for (r = 0; r < 4; ++r) {
					if (skip[r]) continue;
					if (e->plan[r] > 0) {
						const bmh_sw_result_t *aln = &res[e->plan[r] - 1];
						const int is_rev = (r >> 1 != (r & 1));
						const int64_t rb = e->rb[r];
						if (aln->score >= o->min_seed_len && aln->qb >= 0) { /* :150-166 */
							bmh_alnreg_t b;
							size_t i, tmp;
							memset(&b, 0, sizeof(b));
							b.qb = is_rev ? l_ms - (aln->qe + 1) : aln->qb;
							b.qe = is_rev ? l_ms - aln->qb : aln->qe + 1;
							b.rb = is_rev ? (l_pac << 1) - (rb + aln->te + 1) : rb + aln->tb;
							b.re = is_rev ? (l_pac << 1) - (rb + aln->tb) : rb + aln->te + 1;
							b.score = aln->score, b.csub = aln->score2, b.secondary = -1;
							b.seedcov = (int32_t)((b.re - b.rb < b.qe - b.qb ? b.re - b.rb : b.qe - b.qb) >> 1);
							push_reg(ma, &b); /* make room, then move b so that ma stays sorted by score */
							for (i = 0; i < ma->n - 1; ++i)
								if (ma->a[i].score < b.score) break;
							tmp = i;
							for (i = ma->n - 1; i > tmp; --i) ma->a[i] = ma->a[i - 1];
							ma->a[i] = b;
						}
						++n;
					} else if (e->plan[r] == -2) ++n;
					if (n) ma->n = (size_t)dedup(dedup_user, (int)ma->n, ma->a); /* :168 */
				}
				s->n += n;
				++s->j;
			}
		}
	}
	if (n_sw)
		for (q = 0; q < n_act; ++q) n_sw[act[q]] = ps[q].n;
done:
	bmh_ctx_set_driver_stats_(ctx, &st);
	if (ps)
		for (q = 0; q < n_act; ++q) free(ps[q].b[0].a), free(ps[q].b[1].a);
	free(ps), free(read_off), free(pool), free(tasks), free(res), free(act);
	return rc;
}
