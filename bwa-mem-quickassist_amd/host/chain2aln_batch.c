/*
 * chain2aln_batch.c -- the batched extension driver (host side, plain C, above the C-ABI).
 *
 * Replaces, for a whole batch of reads, the reference's per-read loop over
 * mem_chain2aln() (bwa-0.7.8/bwamem.c:730-878; call sites :1105 and :1141) --
 * the function the fork meant to batch as mem_chain2aln_batched() (bwamem.c:580,
 * commented call at :1110) but never finished.
 *
 * Why a state machine and not "extend everything": inside one read the work is
 * sequential by construction -- the right extension starts from the left score
 * (bwamem.c:842,854), the second band try depends on the first (:828,:856), and
 * whether a seed is extended at all depends on the regions produced by earlier
 * seeds AND earlier chains of the same read (:769-802, shared `av`).  Different
 * reads are independent.  So every read runs the exact control flow of the
 * reference as a resumable state machine that stops whenever it needs a
 * ksw_extend2 result; each ROUND collects one pending extension per unfinished
 * read, runs them as one GPU batch (bmh_extend_batch on the resident sequence
 * pool), and feeds the results back.  Exactly the extensions the reference would
 * run are run -- no speculation, identical output.
 *
 * Sequence pool (uploaded once per batch): every read's codes, then one
 * reference window [rmax0,rmax1) per chain (bwamem.c:740-757), decoded on the
 * host from the 2-bit pac (bns_get_seq stays a host stage).
 * Left extensions read query and window backwards via BMH_F_QREV|BMH_F_TREV
 * instead of materialising reversed copies (bwamem.c:813-817).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../../include/bwamem_hip.h"

int bmh_upload_pool(bmh_ctx_t *ctx, const uint8_t *pool, size_t bytes);
const bmh_params_t *bmh_ctx_params_(const bmh_ctx_t *ctx);
int bmh_ctx_has_pac_(const bmh_ctx_t *ctx, const uint8_t *pac, int64_t l_pac);
void bmh_ctx_set_driver_stats_(bmh_ctx_t *ctx, const bmh_driver_stats_t *st);

#define MAX_BAND_TRY 2 /* bwamem.c:493 */

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* bwamem.c:544-551 */
static int cal_max_gap(const bmh_params_t *p, int qlen)
{
	int l_del = (int)((double)(qlen * p->a - p->o_del) / p->e_del + 1.);
	int l_ins = (int)((double)(qlen * p->a - p->o_ins) / p->e_ins + 1.);
	int l = imax(imax(l_del, l_ins), 1);
	return imin(l, p->w << 1);
}

/* 2-bit reference -> one code per byte over the doubled coordinate (bntseq.c:355-376).
 * The caller guarantees [beg,end) does not straddle l_pac (bwamem.c:752-755). */
static void fetch_window(int64_t l_pac, const uint8_t *pac, int64_t beg, int64_t end, uint8_t *dst)
{
	int64_t k, l = 0;
	if (beg >= l_pac) {
		int64_t lo = (l_pac << 1) - 1 - end, hi = (l_pac << 1) - 1 - beg;
		for (k = hi; k > lo; --k) dst[l++] = (uint8_t)(3 - (pac[k >> 2] >> ((~k & 3) << 1) & 3));
	} else {
		for (k = beg; k < end; ++k) dst[l++] = (uint8_t)(pac[k >> 2] >> ((~k & 3) << 1) & 3);
	}
}

typedef struct {
	int64_t rmax0, rmax1;
	uint64_t win_off; /* pool offset of the window's first base */
} chain_win_t;

enum { ST_NEXT_CHAIN, ST_NEXT_SEED, ST_LEFT_WAIT, ST_RIGHT_WAIT, ST_DONE };

typedef struct {
	int st, ci, k, tri;
	int aw0, aw1, sc0;
	size_t ai;          /* index of the region under construction in regs[r].a */
	size_t chain_base;  /* index of this read's first chain in the flat chain_win_t array */
	uint64_t read_off;  /* pool offset of the read */
	uint64_t *srt;
	const bmh_seed_t *s;
} rstate_t;

static bmh_alnreg_t *regs_push(bmh_alnreg_v *v) /* kv_pushp, kvec.h:82-86 */
{
	if (v->n == v->m) {
		v->m = v->m ? v->m << 1 : 2;
		v->a = (bmh_alnreg_t *)realloc(v->a, sizeof(bmh_alnreg_t) * v->m);
	}
	return &v->a[v->n++];
}

static int cmp_u64(const void *a, const void *b)
{
	uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
	return (x > y) - (x < y);
}

/* bwamem.c:769-784 */
static int seed_near_region(const bmh_params_t *p, const bmh_seed_t *s, const bmh_alnreg_v *av)
{
	size_t i;
	for (i = 0; i < av->n; ++i) {
		const bmh_alnreg_t *r = &av->a[i];
		int64_t rd;
		int qd, w, g;
		if (s->rbeg < r->rb || s->rbeg + s->len > r->re || s->qbeg < r->qb || s->qbeg + s->len > r->qe) continue;
		qd = s->qbeg - r->qb, rd = s->rbeg - r->rb;
		g = cal_max_gap(p, qd < rd ? qd : (int)rd);
		w = imin(g, p->w);
		if (qd - rd < w && rd - qd < w) return 1;
		qd = r->qe - (s->qbeg + s->len), rd = r->re - (s->rbeg + s->len);
		g = cal_max_gap(p, qd < rd ? qd : (int)rd);
		w = imin(g, p->w);
		if (qd - rd < w && rd - qd < w) return 1;
	}
	return 0;
}

/* bwamem.c:788-799: does another, not-skipped, long-enough seed overlap s off-diagonal? */
static int has_conflicting_seed(const bmh_chain_t *c, const uint64_t *srt, int k, const bmh_seed_t *s)
{
	int i;
	for (i = k + 1; i < c->n; ++i) {
		const bmh_seed_t *t;
		if (srt[i] == 0) continue;
		t = &c->seeds[(uint32_t)srt[i]];
		if (t->len < s->len * .95) continue;
		if (s->qbeg <= t->qbeg && s->qbeg + s->len - t->qbeg >= s->len >> 2 && t->qbeg - s->qbeg != t->rbeg - s->rbeg) return 1;
		if (t->qbeg <= s->qbeg && t->qbeg + t->len - s->qbeg >= s->len >> 2 && s->qbeg - t->qbeg != s->rbeg - t->rbeg) return 1;
	}
	return 0;
}

typedef struct {
	const bmh_params_t *p;
	const bmh_read_t *reads;
	const bmh_chain_v *chains;
	bmh_chain_pre_fn pre;
	void *pre_ud;
	bmh_alnreg_v *regs;
	const chain_win_t *wins;
	int tpac; /* targets come from the device-resident pac */
	bmh_driver_stats_t st;
	int err;
} drv_t;

static int emit_left(drv_t *d, int r, rstate_t *rs, bmh_ext_task_t *t)
{
	const chain_win_t *cw = &d->wins[rs->chain_base + (size_t)rs->ci];
	const bmh_seed_t *s = rs->s;
	int64_t tl = s->rbeg - cw->rmax0;
	if (s->qbeg > 65535 || tl > 65535 || tl < 0) return BMH_E_RANGE;
	rs->aw0 = d->p->w << rs->tri;
	if (rs->aw0 > 32767) return BMH_E_RANGE;
	memset(t, 0, sizeof(*t));
	t->q_off = rs->read_off + (uint64_t)(s->qbeg - 1); /* query[qbeg-1-i], bwamem.c:814 */
	if (d->tpac) t->t_off = (uint64_t)(tl > 0 ? s->rbeg - 1 : cw->rmax0); /* same bases, read from the resident pac */
	else t->t_off = cw->win_off + (uint64_t)(tl > 0 ? tl - 1 : 0);       /* rseq[tmp-1-i], bwamem.c:817 */
	t->qlen = (uint16_t)s->qbeg, t->tlen = (uint16_t)tl;
	t->h0 = s->len * d->p->a;
	t->w = (int16_t)rs->aw0, t->end_bonus = (int16_t)d->p->pen_clip5;
	t->flags = BMH_F_QREV | BMH_F_TREV | (d->tpac ? BMH_F_TPAC : 0);
	(void)r;
	return 0;
}

static int emit_right(drv_t *d, int r, rstate_t *rs, bmh_ext_task_t *t)
{
	const chain_win_t *cw = &d->wins[rs->chain_base + (size_t)rs->ci];
	const bmh_seed_t *s = rs->s;
	const int l_query = d->reads[r].l_seq, qe = s->qbeg + s->len;
	const int64_t re = s->rbeg + s->len - cw->rmax0, tl = cw->rmax1 - cw->rmax0 - re;
	if (l_query - qe > 65535 || tl > 65535 || tl < 0 || re < 0) return BMH_E_RANGE;
	rs->aw1 = d->p->w << rs->tri;
	if (rs->aw1 > 32767) return BMH_E_RANGE;
	memset(t, 0, sizeof(*t));
	t->q_off = rs->read_off + (uint64_t)qe;
	t->t_off = d->tpac ? (uint64_t)(cw->rmax0 + re) : cw->win_off + (uint64_t)re;
	t->flags = d->tpac ? BMH_F_TPAC : 0;
	t->qlen = (uint16_t)(l_query - qe), t->tlen = (uint16_t)tl;
	t->h0 = rs->sc0;
	t->w = (int16_t)rs->aw1, t->end_bonus = (int16_t)d->p->pen_clip3;
	return 0;
}

static void finish_seed(drv_t *d, int r, rstate_t *rs)
{
	const bmh_chain_t *c = &d->chains[r].a[rs->ci];
	bmh_alnreg_t *a = &d->regs[r].a[rs->ai];
	int i;
	for (i = 0, a->seedcov = 0; i < c->n; ++i) { /* bwamem.c:870-874 */
		const bmh_seed_t *t = &c->seeds[i];
		if (t->qbeg >= a->qb && t->qbeg + t->len <= a->qe && t->rbeg >= a->rb && t->rbeg + t->len <= a->re)
			a->seedcov += t->len;
	}
	a->w = imax(rs->aw0, rs->aw1); /* bwamem.c:875 */
	--rs->k;
	rs->st = ST_NEXT_SEED;
}

/* after the left side is known: start the right side or close the region (bwamem.c:841,866) */
static int begin_right(drv_t *d, int r, rstate_t *rs, bmh_ext_task_t *t)
{
	const bmh_seed_t *s = rs->s;
	bmh_alnreg_t *a = &d->regs[r].a[rs->ai];
	if (s->qbeg + s->len != d->reads[r].l_seq) {
		int rc;
		rs->sc0 = a->score;
		rs->tri = 0;
		if ((rc = emit_right(d, r, rs, t))) return rc;
		rs->st = ST_RIGHT_WAIT;
		return 1; /* task emitted */
	}
	a->qe = d->reads[r].l_seq, a->re = s->rbeg + s->len;
	finish_seed(d, r, rs);
	return 0;
}

/* Runs read r until it needs a GPU result (returns 1 with *t filled) or is done (returns 0); <0 on error. */
static int advance(drv_t *d, int r, rstate_t *rs, bmh_ext_task_t *t)
{
	for (;;) {
		if (rs->st == ST_NEXT_CHAIN) {
			const bmh_chain_t *c;
			int i;
			++rs->ci;
			if ((size_t)rs->ci >= d->chains[r].n) {
				rs->st = ST_DONE;
				return 0;
			}
			c = &d->chains[r].a[rs->ci];
			/* the caller's pre-step, e.g. mem_chain2aln_short (bwamem.c:1104/1140): <=0 means "chain done" */
			if (d->pre && d->pre(d->pre_ud, r, rs->ci, &d->regs[r]) <= 0) continue;
			if (c->n == 0) continue; /* bwamem.c:738 */
			rs->srt = (uint64_t *)malloc((size_t)c->n * 8); /* bwamem.c:760-763 */
			for (i = 0; i < c->n; ++i) rs->srt[i] = (uint64_t)c->seeds[i].len << 32 | (uint32_t)i;
			qsort(rs->srt, (size_t)c->n, 8, cmp_u64);
			rs->k = c->n - 1;
			rs->st = ST_NEXT_SEED;
		} else if (rs->st == ST_NEXT_SEED) {
			const bmh_chain_t *c = &d->chains[r].a[rs->ci];
			bmh_alnreg_t *a;
			int rc;
			if (rs->k < 0) {
				free(rs->srt);
				rs->srt = 0;
				rs->st = ST_NEXT_CHAIN;
				continue;
			}
			rs->s = &c->seeds[(uint32_t)rs->srt[rs->k]];
			if (seed_near_region(d->p, rs->s, &d->regs[r]) && !has_conflicting_seed(c, rs->srt, rs->k, rs->s)) {
				rs->srt[rs->k] = 0; /* bwamem.c:796-799 */
				--rs->k;
				++d->st.seeds_skipped;
				continue;
			}
			++d->st.seeds_extended;
			a = regs_push(&d->regs[r]); /* bwamem.c:804-807 */
			rs->ai = d->regs[r].n - 1;
			memset(a, 0, sizeof(*a));
			a->w = rs->aw0 = rs->aw1 = d->p->w;
			a->score = a->truesc = -1;
			if (rs->s->qbeg) { /* bwamem.c:810 */
				rs->tri = 0;
				if ((rc = emit_left(d, r, rs, t))) return rc;
				rs->st = ST_LEFT_WAIT;
				return 1;
			}
			a->score = a->truesc = rs->s->len * d->p->a, a->qb = 0, a->rb = rs->s->rbeg; /* bwamem.c:839 */
			if ((rc = begin_right(d, r, rs, t)) != 0) return rc;
		} else return rs->st == ST_DONE ? 0 : BMH_E_ARG;
	}
}

/* Feeds one extension result to read r; returns 1 if it immediately needs another (same seed), 0 otherwise. */
static int deliver(drv_t *d, int r, rstate_t *rs, const bmh_ext_result_t *x, bmh_ext_task_t *t)
{
	bmh_alnreg_t *a = &d->regs[r].a[rs->ai];
	const bmh_seed_t *s = rs->s;
	const int prev = a->score;
	int rc;
	a->score = x->score;
	if (rs->st == ST_LEFT_WAIT) {
		const int aw = rs->aw0;
		if (!(a->score == prev || x->max_off < (aw >> 1) + (aw >> 2)) && rs->tri + 1 < MAX_BAND_TRY) { /* bwamem.c:828 */
			++rs->tri;
			if ((rc = emit_left(d, r, rs, t))) return rc;
			return 1;
		}
		if (x->gscore <= 0 || x->gscore <= a->score - d->p->pen_clip5) { /* bwamem.c:831-837 */
			a->qb = s->qbeg - x->qle, a->rb = s->rbeg - x->tle;
			a->truesc = a->score;
		} else {
			a->qb = 0, a->rb = s->rbeg - x->gtle;
			a->truesc = x->gscore;
		}
		return begin_right(d, r, rs, t);
	} else { /* ST_RIGHT_WAIT */
		const chain_win_t *cw = &d->wins[rs->chain_base + (size_t)rs->ci];
		const int aw = rs->aw1, qe = s->qbeg + s->len;
		const int64_t re = s->rbeg + s->len - cw->rmax0;
		if (!(a->score == prev || x->max_off < (aw >> 1) + (aw >> 2)) && rs->tri + 1 < MAX_BAND_TRY) { /* bwamem.c:856 */
			++rs->tri;
			if ((rc = emit_right(d, r, rs, t))) return rc;
			return 1;
		}
		if (x->gscore <= 0 || x->gscore <= a->score - d->p->pen_clip3) { /* bwamem.c:859-865 */
			a->qe = qe + x->qle, a->re = cw->rmax0 + re + x->tle;
			a->truesc += a->score - rs->sc0;
		} else {
			a->qe = d->reads[r].l_seq, a->re = cw->rmax0 + re + x->gtle;
			a->truesc += x->gscore - rs->sc0;
		}
		finish_seed(d, r, rs);
		return 0;
	}
}

static double now_s(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int bmh_chain2aln_batch(bmh_ctx_t *ctx, int64_t l_pac, const uint8_t *pac, int n_reads, const bmh_read_t *reads,
                        const bmh_chain_v *chains, bmh_chain_pre_fn pre, void *pre_ud, bmh_alnreg_v *regs)
{
	const int trace = getenv("BMH_DRIVER_TRACE") != 0; /* where a call's time goes, on stderr */
	double t_trace[4] = {0, 0, 0, 0};
	drv_t d;
	rstate_t *rs = 0;
	chain_win_t *wins = 0;
	uint8_t *pool = 0;
	bmh_ext_task_t *tasks = 0;
	bmh_ext_result_t *res = 0;
	int *owner = 0;
	size_t n_chains = 0, pool_bytes = 0, ci_flat;
	int r, rc = BMH_OK, n_tasks;

	if (!ctx || n_reads < 0 || (n_reads > 0 && (!reads || !chains || !regs || !pac))) return BMH_E_ARG;
	memset(&d, 0, sizeof(d));
	d.p = bmh_ctx_params_(ctx);
	if (!d.p) return BMH_E_ARG;
	if (n_reads == 0) return BMH_OK;
	d.reads = reads, d.chains = chains, d.pre = pre, d.pre_ud = pre_ud, d.regs = regs;
	t_trace[0] = trace ? now_s() : 0;

	/* pass 1: window of every live chain (bwamem.c:740-755) and pool layout */
	rs = (rstate_t *)calloc((size_t)n_reads, sizeof(rstate_t));
	for (r = 0; r < n_reads; ++r) n_chains += chains[r].n;
	wins = (chain_win_t *)calloc(n_chains + 1, sizeof(chain_win_t));
	if (!rs || !wins) { rc = BMH_E_NOMEM; goto done; }
	for (r = 0, ci_flat = 0; r < n_reads; ++r) {
		size_t ci;
		rs[r].read_off = pool_bytes, rs[r].chain_base = ci_flat, rs[r].ci = -1, rs[r].st = ST_NEXT_CHAIN;
		pool_bytes += (size_t)reads[r].l_seq;
		for (ci = 0; ci < chains[r].n; ++ci, ++ci_flat) {
			const bmh_chain_t *c = &chains[r].a[ci];
			chain_win_t *cw = &wins[ci_flat];
			const int l_query = reads[r].l_seq;
			int i;
			if (c->n == 0) continue;
			cw->rmax0 = l_pac << 1, cw->rmax1 = 0;
			for (i = 0; i < c->n; ++i) {
				const bmh_seed_t *t = &c->seeds[i];
				const int rest = l_query - t->qbeg - t->len;
				const int64_t b = t->rbeg - (t->qbeg + cal_max_gap(d.p, t->qbeg));
				const int64_t e = t->rbeg + t->len + (rest + cal_max_gap(d.p, rest));
				if (b < cw->rmax0) cw->rmax0 = b;
				if (e > cw->rmax1) cw->rmax1 = e;
			}
			if (cw->rmax0 < 0) cw->rmax0 = 0;
			if (cw->rmax1 > l_pac << 1) cw->rmax1 = l_pac << 1;
			if (cw->rmax0 < l_pac && l_pac < cw->rmax1) {
				if (c->seeds[0].rbeg < l_pac) cw->rmax1 = l_pac;
				else cw->rmax0 = l_pac;
			}
			if (cw->rmax1 < cw->rmax0) { rc = BMH_E_ARG; goto done; }
		}
	}
	/* with the reference resident on the device (bmh_ctx_set_pac) the tasks address it directly and no window is
	 * decoded or shipped: bns_get_seq (bwamem.c:757) happens inside the kernels */
	d.tpac = bmh_ctx_has_pac_(ctx, pac, l_pac);
	for (ci_flat = 0; ci_flat < n_chains && !d.tpac; ++ci_flat) {
		wins[ci_flat].win_off = pool_bytes;
		pool_bytes += (size_t)(wins[ci_flat].rmax1 - wins[ci_flat].rmax0);
	}
	d.wins = wins;

	/* pass 2: fill and upload the pool once */
	pool = (uint8_t *)malloc(pool_bytes + 16);
	tasks = (bmh_ext_task_t *)malloc(sizeof(bmh_ext_task_t) * (size_t)n_reads);
	res = (bmh_ext_result_t *)malloc(sizeof(bmh_ext_result_t) * (size_t)n_reads);
	owner = (int *)malloc(sizeof(int) * (size_t)n_reads);
	if (!pool || !tasks || !res || !owner) { rc = BMH_E_NOMEM; goto done; }
	for (r = 0; r < n_reads; ++r) memcpy(pool + rs[r].read_off, reads[r].seq, (size_t)reads[r].l_seq);
	for (ci_flat = 0; ci_flat < n_chains && !d.tpac; ++ci_flat)
		if (wins[ci_flat].rmax1 > wins[ci_flat].rmax0)
			fetch_window(l_pac, pac, wins[ci_flat].rmax0, wins[ci_flat].rmax1, pool + wins[ci_flat].win_off);
	memset(pool + pool_bytes, 0, 16);
	d.st.pool_bytes = (int64_t)pool_bytes + 16;
	if ((rc = bmh_upload_pool(ctx, pool, pool_bytes + 16))) goto done;

	/* rounds */
	t_trace[1] = trace ? now_s() : 0;
	n_tasks = 0;
	for (r = 0; r < n_reads; ++r) {
		int k = advance(&d, r, &rs[r], &tasks[n_tasks]);
		if (k < 0) { rc = k; goto done; }
		if (k) owner[n_tasks++] = r;
	}
	while (n_tasks > 0) {
		int i, m = 0;
		++d.st.rounds;
		d.st.ext_tasks += n_tasks;
		t_trace[3] = trace ? now_s() : 0;
		if ((rc = bmh_extend_batch(ctx, 0, 0, tasks, n_tasks, res))) goto done;
		if (trace) t_trace[2] += now_s() - t_trace[3];
		for (i = 0; i < n_tasks; ++i) { /* compact in place: slot m <= i is free once task i is consumed */
			bmh_ext_task_t nt;
			int k;
			r = owner[i];
			k = deliver(&d, r, &rs[r], &res[i], &nt);
			if (k == 0) k = advance(&d, r, &rs[r], &nt);
			if (k < 0) { rc = k; goto done; }
			if (k) tasks[m] = nt, owner[m++] = r;
		}
		n_tasks = m;
	}
	if (trace)
		fprintf(stderr, "[bwamem_hip] bmh_chain2aln_batch %d reads: windows+pool+upload %.1f ms, %lld rounds: GPU calls %.1f ms, state machine %.1f ms\n",
		        n_reads, (t_trace[1] - t_trace[0]) * 1e3, (long long)d.st.rounds, t_trace[2] * 1e3, (now_s() - t_trace[1] - t_trace[2]) * 1e3);
done:
	if (rs) for (r = 0; r < n_reads; ++r) free(rs[r].srt);
	bmh_ctx_set_driver_stats_(ctx, &d.st);
	free(rs), free(wins), free(pool), free(tasks), free(res), free(owner);
	return rc;
}
