/*
 * chain2aln_batch.c -- the batched extension driver (host side, plain C, above the C-ABI).
 *
 * Replaces, for a whole batch of reads, the reference's per-read loop over
 * mem_chain2aln() (bwa-0.7.8/bwamem.c:730-878; call sites :1105 and :1141) --
 * the function the fork meant to batch as mem_chain2aln_batched() (bwamem.c:580,
 * commented call at :1110) but never finished.
 *
 * Inside one read the work is sequential by construction: the right extension
 * starts from the left score (bwamem.c:842,854), the second band try depends on
 * the first (:828,:856), and whether a seed is extended at all depends on the
 * regions produced by earlier seeds AND earlier chains of the same read
 * (:769-802, shared `av`).  Different reads are independent.
 *
 * The first two dependencies live on the device: bmh_seedext_batch runs a seed's
 * left extension, retries, clip decision and right extension as ONE record (the
 * ext_param_t/ext_res_t the fork sketched, bwamem.c:553-577).  The third one is
 * resolved here with a result cache and (bounded) speculation, which is exact
 * because a seed's extension is a pure function of the seed and its window:
 *   round 1  the longest seed of every chain of every read -- the one the
 *            reference extends first in that chain (bwamem.c:760-765);
 *   replay   every read runs the reference's exact control flow, taking
 *            extension results from the cache; a read that needs a result that
 *            is not there stops;
 *   round 2  for the stopped reads: the missing seed and every later seed the
 *            reference could still extend (all but those that are provably
 *            skipped: contained in a region that already exists, bwamem.c:769-784,
 *            with no conflicting seed even among the ones that were skipped,
 *            :788-799 -- regions only accumulate, so "contained" stays true);
 *   replay   completes every read.  (The loop would go on if a result were still
 *            missing; by the argument above it is not.)
 * Output is identical to the reference's; the extensions run are a superset of
 * the reference's (bmh_driver_stats: seeds_extended vs seeds_speculated).
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
	size_t seed_base; /* index of the chain's seed 0 in the flat result cache */
	/* mem_chain2aln_short (bwamem.c:495-542), when the driver runs it itself: -1 = it returns without a Smith-Waterman
	 * (the chain goes on to mem_chain2aln), else the index of its ksw_align2 in the batch's short-chain tasks */
	int32_t sw_idx, sqb, sqe, seedcov;
	int64_t srb, sre;
	uint64_t swin_off; /* pool offset of [srb,sre) when the reference is not resident */
} chain_win_t;

#define MEM_SHORT_EXT 50  /* bwamem.c:491-492 */
#define MEM_SHORT_LEN 200

enum { ST_NEXT_CHAIN, ST_NEXT_SEED, ST_DONE };

typedef struct {
	int st, ci, k;
	size_t chain_base;  /* index of this read's first chain in the flat chain_win_t array */
	uint64_t read_off;  /* pool offset of the read */
	uint64_t *srt;
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

/* the overlap test of bwamem.c:793-794 */
static inline int seeds_conflict(const bmh_seed_t *s, const bmh_seed_t *t)
{
	if (t->len < s->len * .95) return 0; /* double compare, bwamem.c:792 */
	if (s->qbeg <= t->qbeg && s->qbeg + s->len - t->qbeg >= s->len >> 2 && t->qbeg - s->qbeg != t->rbeg - s->rbeg) return 1;
	if (t->qbeg <= s->qbeg && t->qbeg + t->len - s->qbeg >= s->len >> 2 && s->qbeg - t->qbeg != s->rbeg - t->rbeg) return 1;
	return 0;
}

/* bwamem.c:788-799: does another, not-skipped, long-enough seed overlap s off-diagonal? */
static int has_conflicting_seed(const bmh_chain_t *c, const uint64_t *srt, int k, const bmh_seed_t *s)
{
	int i;
	for (i = k + 1; i < c->n; ++i) {
		if (srt[i] == 0) continue;
		if (seeds_conflict(s, &c->seeds[(uint32_t)srt[i]])) return 1;
	}
	return 0;
}

/* the same question asked BEFORE the chain's earlier seeds have been decided: every seed that sorts after `si`
 * (longer, or as long with a larger index) counts, skipped or not -- a superset of the conflicts the reference will see */
static int may_conflict(const bmh_chain_t *c, int si)
{
	const bmh_seed_t *s = &c->seeds[si];
	const uint64_t key = (uint64_t)s->len << 32 | (uint32_t)si;
	int i;
	for (i = 0; i < c->n; ++i) {
		const uint64_t ki = (uint64_t)c->seeds[i].len << 32 | (uint32_t)i;
		if (ki <= key || ki == 0) continue;
		if (seeds_conflict(s, &c->seeds[i])) return 1;
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
	/* extension results by (chain, seed): 0 = not asked for, 1 = asked for in the round being built, 2 = there */
	uint8_t *have;
	bmh_seed_result_t *cache;
	int short_msl; /* > 0: mem_chain2aln_short is the driver's own pre-step, with this opt->min_seed_len */
	const bmh_sw_result_t *sw_res;
} drv_t;

/* The part of mem_chain2aln_short before its ksw_align2 (bwamem.c:504-527): does the chain qualify, and for which
 * query / reference intervals?  Returns 1 and fills cw->s* if a Smith-Waterman is to be run. */
static int short_candidate(const bmh_params_t *p, int64_t l_pac, int l_query, const bmh_chain_t *c, chain_win_t *cw)
{
	int i, qb = l_query, qe = 0, cov = 0;
	int64_t rb = l_pac << 1, re = 0;
	cw->sw_idx = -1;
	if (c->n <= 0) return 0;
	for (i = 0; i < c->n; ++i) {
		const bmh_seed_t *s = &c->seeds[i];
		qb = qb < s->qbeg ? qb : s->qbeg;
		qe = qe > s->qbeg + s->len ? qe : s->qbeg + s->len;
		rb = rb < s->rbeg ? rb : s->rbeg;
		re = re > s->rbeg + s->len ? re : s->rbeg + s->len;
		cov += s->len;
	}
	qb -= MEM_SHORT_EXT, qe += MEM_SHORT_EXT;
	if (qb <= 10 || qe >= l_query - 10) return 0; /* ksw_align2 cannot align to the ends */
	rb -= MEM_SHORT_EXT, re += MEM_SHORT_EXT;
	rb = rb > 0 ? rb : 0;
	re = re < l_pac << 1 ? re : l_pac << 1;
	if (rb < l_pac && l_pac < re) {
		if (c->seeds[0].rbeg < l_pac) re = l_pac;
		else rb = l_pac;
	}
	if ((re - rb) - (qe - qb) > MEM_SHORT_EXT || (qe - qb) - (re - rb) > MEM_SHORT_EXT) return 0;
	if (qe - qb >= p->w * 4 || re - rb >= p->w * 4) return 0;
	if (qe - qb >= MEM_SHORT_LEN || re - rb >= MEM_SHORT_LEN) return 0;
	cw->sqb = qb, cw->sqe = qe, cw->srb = rb, cw->sre = re, cw->seedcov = cov;
	return 1;
}

/* Runs read r with the reference's control flow (bwamem.c:1101-1107 over :760-876) for as long as the extension results
 * it needs are cached.  Returns 0 when the read is finished, 1 when it stopped at a seed whose result is missing. */
static int run_read(drv_t *d, int r, rstate_t *rs)
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
			if (d->short_msl > 0) { /* ... or the driver's own: the second half of mem_chain2aln_short, bwamem.c:533-541 */
				const chain_win_t *cw = &d->wins[rs->chain_base + (size_t)rs->ci];
				if (cw->sw_idx >= 0) {
					const bmh_sw_result_t *x = &d->sw_res[cw->sw_idx];
					if (!(x->tb < MEM_SHORT_EXT >> 1 || x->te > cw->sre - cw->srb - (MEM_SHORT_EXT >> 1))) {
						bmh_alnreg_t *a = regs_push(&d->regs[r]);
						memset(a, 0, sizeof(*a));
						a->rb = cw->srb + x->tb, a->re = cw->srb + x->te + 1;
						a->qb = cw->sqb + x->qb, a->qe = cw->sqb + x->qe + 1;
						a->score = x->score, a->csub = x->score2, a->seedcov = cw->seedcov;
						continue; /* the chain is settled */
					}
				}
			}
			if (c->n == 0) continue; /* bwamem.c:738 */
			rs->srt = (uint64_t *)malloc((size_t)c->n * 8); /* bwamem.c:760-763 */
			for (i = 0; i < c->n; ++i) rs->srt[i] = (uint64_t)c->seeds[i].len << 32 | (uint32_t)i;
			qsort(rs->srt, (size_t)c->n, 8, cmp_u64);
			rs->k = c->n - 1;
			rs->st = ST_NEXT_SEED;
		} else if (rs->st == ST_NEXT_SEED) {
			const bmh_chain_t *c = &d->chains[r].a[rs->ci];
			const chain_win_t *cw = &d->wins[rs->chain_base + (size_t)rs->ci];
			const bmh_seed_t *s;
			const bmh_seed_result_t *x;
			bmh_alnreg_t *a;
			uint32_t si;
			int i;
			if (rs->k < 0) {
				free(rs->srt);
				rs->srt = 0;
				rs->st = ST_NEXT_CHAIN;
				continue;
			}
			si = (uint32_t)rs->srt[rs->k];
			s = &c->seeds[si];
			if (seed_near_region(d->p, s, &d->regs[r]) && !has_conflicting_seed(c, rs->srt, rs->k, s)) {
				rs->srt[rs->k] = 0; /* bwamem.c:796-799 */
				--rs->k;
				++d->st.seeds_skipped;
				continue;
			}
			if (d->have[cw->seed_base + si] != 2) return 1; /* the device has not extended this seed yet */
			++d->st.seeds_extended;
			x = &d->cache[cw->seed_base + si];
			a = regs_push(&d->regs[r]); /* bwamem.c:804-807 */
			memset(a, 0, sizeof(*a));
			a->qb = x->qb, a->qe = x->qe, a->rb = cw->rmax0 + x->rb, a->re = cw->rmax0 + x->re; /* bwamem.c:831-866 */
			a->score = x->score, a->truesc = x->truesc, a->w = x->w;                          /* ... and :875 */
			for (i = 0, a->seedcov = 0; i < c->n; ++i) { /* bwamem.c:870-874 */
				const bmh_seed_t *t = &c->seeds[i];
				if (t->qbeg >= a->qb && t->qbeg + t->len <= a->qe && t->rbeg >= a->rb && t->rbeg + t->len <= a->re)
					a->seedcov += t->len;
			}
			--rs->k;
		} else return 0;
	}
}

typedef struct {
	bmh_seed_task_t *t;
	size_t *slot; /* cache slot of each task */
	size_t n, m;
} req_t;

static int request(drv_t *d, req_t *q, int r, const rstate_t *rs, int ci, int si)
{
	const chain_win_t *cw = &d->wins[rs->chain_base + (size_t)ci];
	const bmh_seed_t *s = &d->chains[r].a[ci].seeds[si];
	bmh_seed_task_t *t;
	if (d->have[cw->seed_base + (size_t)si]) return 0;
	if (q->n == q->m) {
		q->m = q->m ? q->m << 1 : 1024;
		q->t = (bmh_seed_task_t *)realloc(q->t, sizeof(*q->t) * q->m);
		q->slot = (size_t *)realloc(q->slot, sizeof(*q->slot) * q->m);
		if (!q->t || !q->slot) return BMH_E_NOMEM;
	}
	if (s->rbeg < cw->rmax0 || s->rbeg + s->len > cw->rmax1 || cw->rmax1 - cw->rmax0 > 0x7fffffff) return BMH_E_RANGE;
	t = &q->t[q->n];
	memset(t, 0, sizeof(*t));
	t->q_off = rs->read_off;
	t->t_off = d->tpac ? (uint64_t)cw->rmax0 : cw->win_off; /* rseq[0], bwamem.c:757 */
	t->flags = d->tpac ? BMH_F_TPAC : 0;
	t->l_query = d->reads[r].l_seq, t->qbeg = s->qbeg, t->len = s->len;
	t->rbeg = (int32_t)(s->rbeg - cw->rmax0), t->wlen = (int32_t)(cw->rmax1 - cw->rmax0);
	q->slot[q->n++] = cw->seed_base + (size_t)si;
	d->have[cw->seed_base + (size_t)si] = 1;
	return 0;
}

static double now_s(void) /* the clock of the BMH_DRIVER_TRACE lines: wall time, or with BMH_TRACE_CPU this thread's CPU time */
{
	static int cpu = -1;
	struct timespec ts;
	if (cpu < 0) cpu = getenv("BMH_TRACE_CPU") != 0;
	clock_gettime(cpu ? CLOCK_THREAD_CPUTIME_ID : CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static int chains2regs(bmh_ctx_t *ctx, int64_t l_pac, const uint8_t *pac, int n_reads, const bmh_read_t *reads,
                       const bmh_chain_v *chains, bmh_chain_pre_fn pre, void *pre_ud, int short_msl, bmh_alnreg_v *regs);

int bmh_chain2aln_batch(bmh_ctx_t *ctx, int64_t l_pac, const uint8_t *pac, int n_reads, const bmh_read_t *reads,
                        const bmh_chain_v *chains, bmh_chain_pre_fn pre, void *pre_ud, bmh_alnreg_v *regs)
{
	return chains2regs(ctx, l_pac, pac, n_reads, reads, chains, pre, pre_ud, 0, regs);
}

int bmh_chains2regs_batch(bmh_ctx_t *ctx, int64_t l_pac, const uint8_t *pac, int n_reads, const bmh_read_t *reads,
                          const bmh_chain_v *chains, int min_seed_len, bmh_alnreg_v *regs)
{
	if (min_seed_len < 1) return BMH_E_ARG;
	return chains2regs(ctx, l_pac, pac, n_reads, reads, chains, 0, 0, min_seed_len, regs);
}

static int chains2regs(bmh_ctx_t *ctx, int64_t l_pac, const uint8_t *pac, int n_reads, const bmh_read_t *reads,
                       const bmh_chain_v *chains, bmh_chain_pre_fn pre, void *pre_ud, int short_msl, bmh_alnreg_v *regs)
{
	const int trace = getenv("BMH_DRIVER_TRACE") != 0; /* where a call's time goes, on stderr */
	double t_trace[4] = {0, 0, 0, 0}, t_fine[4] = {0, 0, 0, 0};
	drv_t d;
	rstate_t *rs = 0;
	chain_win_t *wins = 0;
	uint8_t *pool = 0;
	bmh_seed_result_t *res = 0;
	bmh_sw_task_t *sw_tasks = 0;
	bmh_sw_result_t *sw_res = 0;
	int *stopped = 0;
	req_t q;
	size_t n_chains = 0, n_seeds = 0, pool_bytes = 0, ci_flat, n_short = 0;
	int r, rc = BMH_OK, n_stopped;

	if (!ctx || n_reads < 0 || (n_reads > 0 && (!reads || !chains || !regs || !pac))) return BMH_E_ARG;
	memset(&d, 0, sizeof(d));
	memset(&q, 0, sizeof(q));
	d.p = bmh_ctx_params_(ctx);
	if (!d.p) return BMH_E_ARG;
	if (n_reads == 0) return BMH_OK;
	d.reads = reads, d.chains = chains, d.pre = pre, d.pre_ud = pre_ud, d.regs = regs, d.short_msl = short_msl;
	t_trace[0] = trace ? now_s() : 0;

	/* pass 1: window of every live chain (bwamem.c:740-755) and pool layout */
	rs = (rstate_t *)calloc((size_t)n_reads, sizeof(rstate_t));
	for (r = 0; r < n_reads; ++r) n_chains += chains[r].n;
	wins = (chain_win_t *)calloc(n_chains + 1, sizeof(chain_win_t));
	stopped = (int *)malloc(sizeof(int) * (size_t)n_reads);
	if (!rs || !wins || !stopped) { rc = BMH_E_NOMEM; goto done; }
	for (r = 0, ci_flat = 0; r < n_reads; ++r) {
		size_t ci;
		rs[r].read_off = pool_bytes, rs[r].chain_base = ci_flat, rs[r].ci = -1, rs[r].st = ST_NEXT_CHAIN;
		pool_bytes += (size_t)reads[r].l_seq;
		for (ci = 0; ci < chains[r].n; ++ci, ++ci_flat) {
			const bmh_chain_t *c = &chains[r].a[ci];
			chain_win_t *cw = &wins[ci_flat];
			const int l_query = reads[r].l_seq;
			int i;
			cw->seed_base = n_seeds, cw->sw_idx = -1;
			if (c->n <= 0) continue;
			if (short_msl > 0 && short_candidate(d.p, l_pac, l_query, c, cw)) cw->sw_idx = (int32_t)n_short++;
			n_seeds += (size_t)c->n;
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
		if (wins[ci_flat].sw_idx >= 0) wins[ci_flat].swin_off = pool_bytes, pool_bytes += (size_t)(wins[ci_flat].sre - wins[ci_flat].srb);
	}
	d.wins = wins;
	if (trace) t_fine[0] = now_s();

	/* pass 2: fill and upload the pool once */
	pool = (uint8_t *)malloc(pool_bytes + 16);
	d.have = (uint8_t *)calloc(n_seeds + 1, 1);
	d.cache = (bmh_seed_result_t *)malloc(sizeof(bmh_seed_result_t) * (n_seeds + 1));
	if (!pool || !d.have || !d.cache) { rc = BMH_E_NOMEM; goto done; }
	for (r = 0; r < n_reads; ++r) memcpy(pool + rs[r].read_off, reads[r].seq, (size_t)reads[r].l_seq);
	for (ci_flat = 0; ci_flat < n_chains && !d.tpac; ++ci_flat) {
		if (wins[ci_flat].rmax1 > wins[ci_flat].rmax0)
			fetch_window(l_pac, pac, wins[ci_flat].rmax0, wins[ci_flat].rmax1, pool + wins[ci_flat].win_off);
		if (wins[ci_flat].sw_idx >= 0) fetch_window(l_pac, pac, wins[ci_flat].srb, wins[ci_flat].sre, pool + wins[ci_flat].swin_off);
	}
	memset(pool + pool_bytes, 0, 16);
	d.st.pool_bytes = (int64_t)pool_bytes + 16;
	if (trace) t_fine[1] = now_s();
	if ((rc = bmh_upload_pool(ctx, pool, pool_bytes + 16))) goto done;
	if (trace) t_fine[2] = now_s();
	if (n_short) { /* every ksw_align2 of the batch's mem_chain2aln_short calls (bwamem.c:529-531) as one GPU batch */
		sw_tasks = (bmh_sw_task_t *)calloc(n_short, sizeof(bmh_sw_task_t));
		sw_res = (bmh_sw_result_t *)malloc(sizeof(bmh_sw_result_t) * n_short);
		if (!sw_tasks || !sw_res) { rc = BMH_E_NOMEM; goto done; }
		for (r = 0, ci_flat = 0; r < n_reads; ++r) {
			size_t ci;
			for (ci = 0; ci < chains[r].n; ++ci, ++ci_flat) {
				const chain_win_t *cw = &wins[ci_flat];
				bmh_sw_task_t *t;
				if (cw->sw_idx < 0) continue;
				t = &sw_tasks[cw->sw_idx];
				t->q_off = rs[r].read_off + (uint64_t)cw->sqb, t->qlen = (uint16_t)(cw->sqe - cw->sqb);
				t->t_off = d.tpac ? (uint64_t)cw->srb : cw->swin_off, t->tlen = (uint32_t)(cw->sre - cw->srb);
				t->flags = d.tpac ? BMH_F_TPAC : 0;
				t->xtra = BMH_SW_XSUBO | BMH_SW_XSTART | ((cw->sqe - cw->sqb) * d.p->a < 250 ? BMH_SW_XBYTE : 0) | (uint32_t)(short_msl * d.p->a);
			}
		}
		if ((rc = bmh_sw_batch(ctx, 0, 0, sw_tasks, (int64_t)n_short, sw_res))) goto done;
		d.sw_res = sw_res;
		d.st.short_sw = (int64_t)n_short;
	}
	t_trace[1] = trace ? now_s() : 0;

	/* round 1: the seed each chain is extended from first -- the last one in (len, index) order, bwamem.c:760-765 */
	for (r = 0; r < n_reads; ++r) {
		size_t ci;
		stopped[r] = r;
		for (ci = 0; ci < chains[r].n; ++ci) {
			const bmh_chain_t *c = &chains[r].a[ci];
			int i, top = 0;
			if (c->n <= 0) continue;
			for (i = 1; i < c->n; ++i)
				if (c->seeds[i].len >= c->seeds[top].len) top = i;
			if ((rc = request(&d, &q, r, &rs[r], (int)ci, top))) goto done;
		}
	}
	n_stopped = n_reads;
	for (;;) {
		bmh_seedext_stats_t ss;
		size_t j;
		int m = 0, i;
		if (q.n) {
			++d.st.rounds;
			t_trace[3] = trace ? now_s() : 0;
			res = (bmh_seed_result_t *)realloc(res, sizeof(*res) * q.n);
			if (!res) { rc = BMH_E_NOMEM; goto done; }
			if ((rc = bmh_seedext_batch(ctx, 0, 0, q.t, (int64_t)q.n, res))) goto done;
			if (trace) t_trace[2] += now_s() - t_trace[3];
			bmh_seedext_stats(ctx, &ss);
			d.st.ext_tasks += ss.left_tasks + ss.left_retries + ss.right_tasks + ss.right_retries;
			d.st.seeds_speculated += (int64_t)q.n;
			for (j = 0; j < q.n; ++j) d.cache[q.slot[j]] = res[j], d.have[q.slot[j]] = 2;
			q.n = 0;
		}
		/* replay: every unfinished read goes on until it is done or needs a seed that has not been extended */
		for (i = 0; i < n_stopped; ++i) {
			int k;
			r = stopped[i];
			k = run_read(&d, r, &rs[r]);
			if (k == 0) continue;
			stopped[m++] = r;
			{ /* what this read may still need: the seed it stopped at, the rest of its chain, the later chains */
				const bmh_chain_t *c = &chains[r].a[rs[r].ci];
				size_t ci;
				int kk;
				if ((rc = request(&d, &q, r, &rs[r], rs[r].ci, (int)(uint32_t)rs[r].srt[rs[r].k]))) goto done;
				for (kk = rs[r].k - 1; kk >= 0; --kk) {
					const int si = (int)(uint32_t)rs[r].srt[kk];
					if (seed_near_region(d.p, &c->seeds[si], &regs[r]) && !may_conflict(c, si)) continue; /* provably skipped */
					if ((rc = request(&d, &q, r, &rs[r], rs[r].ci, si))) goto done;
				}
				for (ci = (size_t)rs[r].ci + 1; ci < chains[r].n; ++ci) {
					const bmh_chain_t *c2 = &chains[r].a[ci];
					int si;
					for (si = 0; si < c2->n; ++si) {
						if (seed_near_region(d.p, &c2->seeds[si], &regs[r]) && !may_conflict(c2, si)) continue;
						if ((rc = request(&d, &q, r, &rs[r], (int)ci, si))) goto done;
					}
				}
			}
		}
		n_stopped = m;
		if (n_stopped == 0) break;
		if (q.n == 0) { rc = BMH_E_ARG; goto done; } /* cannot happen: a stopped read always asks for its seed */
	}
	d.st.seeds_speculated -= d.st.seeds_extended; /* extended on the device but never used */
	if (trace)
		fprintf(stderr, "[bwamem_hip] bmh_chain2aln_batch %d reads: windows+pool+upload %.1f ms (windows %.1f ms, pool %.1f ms, upload %.1f ms, short-chain SW %.1f ms), %lld rounds: GPU calls %.1f ms, replay %.1f ms; %lld seeds extended, %lld speculated in vain\n",
		        n_reads, (t_trace[1] - t_trace[0]) * 1e3, (t_fine[0] - t_trace[0]) * 1e3, (t_fine[1] - t_fine[0]) * 1e3, (t_fine[2] - t_fine[1]) * 1e3,
		        (t_trace[1] - t_fine[2]) * 1e3, (long long)d.st.rounds, t_trace[2] * 1e3, (now_s() - t_trace[1] - t_trace[2]) * 1e3,
		        (long long)d.st.seeds_extended, (long long)d.st.seeds_speculated);
done:
	if (rs) for (r = 0; r < n_reads; ++r) free(rs[r].srt);
	bmh_ctx_set_driver_stats_(ctx, &d.st);
	free(rs), free(wins), free(pool), free(res), free(stopped), free(q.t), free(q.slot), free(d.have), free(d.cache), free(sw_tasks), free(sw_res);
	return rc;
}
