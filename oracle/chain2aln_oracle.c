/*
 * oracle/chain2aln_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of the extension driver mem_chain2aln()
 * (reference: bwa-0.7.8/bwamem.c:730-878, helpers cal_max_gap :544-551 and
 * bns_get_seq bntseq.c:355-376), written from SURVEY.md Appendix A.3.
 * One read, one chain at a time, calling orc_extend() for every extension --
 * the sequential ground truth the batched GPU driver must reproduce.
 *
 * Parity status: PINNED against the compiled reference's mem_chain2aln
 * (oracle/_ref/libbwa_ref.so) by tests/test_oracle_vs_ref.py and the
 * committed fixtures tests/golden/chain2aln_*.bin.
 */
#include "chain2aln_oracle.h"

#include <stdlib.h>
#include <string.h>

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* bwamem.c:544-551 */
int orc_cal_max_gap(const bmh_params_t *p, int qlen)
{
	int l_del = (int)((double)(qlen * p->a - p->o_del) / p->e_del + 1.);
	int l_ins = (int)((double)(qlen * p->a - p->o_ins) / p->e_ins + 1.);
	int l = imax(imax(l_del, l_ins), 1);
	return imin(l, p->w << 1);
}

/* 2-bit packed reference, 4 bases per byte, first base in the top bits.  bntseq.c:191-192 */
static inline int pac_base(const uint8_t *pac, int64_t l) { return pac[l >> 2] >> ((~l & 3) << 1) & 3; }

/* bntseq.c:355-376: [beg,end) on the doubled (forward + reverse-complement)
 * coordinate; returns malloc'd codes and *len, or *len = 0 if it bridges l_pac. */
uint8_t *orc_get_seq(int64_t l_pac, const uint8_t *pac, int64_t beg, int64_t end, int64_t *len)
{
	uint8_t *seq = 0;
	int64_t k, l = 0;
	if (end < beg) { int64_t t = beg; beg = end, end = t; }
	if (end > l_pac << 1) end = l_pac << 1;
	if (beg < 0) beg = 0;
	if (!(beg >= l_pac || end <= l_pac)) { *len = 0; return 0; }
	*len = end - beg;
	seq = (uint8_t *)malloc((size_t)(end - beg) + 1);
	if (beg >= l_pac) { /* reverse strand: complement, walking backwards */
		int64_t lo = (l_pac << 1) - 1 - end, hi = (l_pac << 1) - 1 - beg;
		for (k = hi; k > lo; --k) seq[l++] = (uint8_t)(3 - pac_base(pac, k));
	} else {
		for (k = beg; k < end; ++k) seq[l++] = (uint8_t)pac_base(pac, k);
	}
	return seq;
}

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

/* bwamem.c:769-784: is seed s "around" an existing region? returns 1 if so */
static int seed_near_region(const bmh_params_t *p, const bmh_seed_t *s, const bmh_alnreg_v *av)
{
	size_t i;
	for (i = 0; i < av->n; ++i) {
		const bmh_alnreg_t *r = &av->a[i];
		int64_t rd;
		int qd, w, g;
		if (s->rbeg < r->rb || s->rbeg + s->len > r->re || s->qbeg < r->qb || s->qbeg + s->len > r->qe) continue;
		qd = s->qbeg - r->qb, rd = s->rbeg - r->rb;
		g = orc_cal_max_gap(p, qd < rd ? qd : (int)rd);
		w = imin(g, p->w);
		if (qd - rd < w && rd - qd < w) return 1;
		qd = r->qe - (s->qbeg + s->len), rd = r->re - (s->rbeg + s->len);
		g = orc_cal_max_gap(p, qd < rd ? qd : (int)rd);
		w = imin(g, p->w);
		if (qd - rd < w && rd - qd < w) return 1;
	}
	return 0;
}

/* What mem_chain2aln does for ONE seed once it has decided to extend it (bwamem.c:808-866): left extension on the
 * reversed flanks with up to MAX_BAND_TRY band widths, the clip-or-reach-the-end decision, right extension started from
 * the left score, its decision.  `rseq` is the chain's window [rmax0,rmax1), `rbeg` the seed's start inside it.
 * Result coordinates rb/re are window-relative.  This is the record the fork sketched as ext_param_t/ext_res_t
 * (bwamem.c:553-577); orc_chain2aln below is built on it, so the reference fixtures of mem_chain2aln pin it. */
void orc_seedext_one(const bmh_params_t *p, const uint8_t *query, int l_query, const uint8_t *rseq, int wlen, int qbeg, int len,
                     int rbeg, bmh_seed_result_t *r, int64_t *cells, int64_t *calls)
{
	orc_scoring_t sc;
	orc_extend_stats_t st;
	int i, aw0 = p->w, aw1 = p->w, score = -1, truesc = -1, n_ext = 0;
	sc.o_del = p->o_del, sc.e_del = p->e_del, sc.o_ins = p->o_ins, sc.e_ins = p->e_ins;
	sc.zdrop = p->zdrop, sc.m = 5, sc.mat = p->mat;
	if (qbeg) { /* left extension on reversed flanks.  bwamem.c:810-838 */
		int tl = rbeg, ql = qbeg;
		uint8_t *qs = (uint8_t *)malloc((size_t)ql + 1), *rs = (uint8_t *)malloc((size_t)tl + 1);
		orc_extend_out_t o;
		for (i = 0; i < ql; ++i) qs[i] = query[ql - 1 - i];
		for (i = 0; i < tl; ++i) rs[i] = rseq[tl - 1 - i];
		memset(&o, 0, sizeof(o));
		for (i = 0; i < 2; ++i) { /* MAX_BAND_TRY, bwamem.c:493 */
			int prev = score;
			aw0 = p->w << i;
			orc_extend(&sc, ql, qs, tl, rs, aw0, p->pen_clip5, len * p->a, &o, &st);
			++n_ext;
			if (cells) *cells += st.cells;
			score = o.score;
			if (score == prev || o.max_off < (aw0 >> 1) + (aw0 >> 2)) break;
		}
		if (o.gscore <= 0 || o.gscore <= score - p->pen_clip5) { /* clip.  bwamem.c:831-833 */
			r->qb = qbeg - o.qle, r->rb = rbeg - o.tle;
			truesc = score;
		} else { /* reach the read end.  bwamem.c:834-837 */
			r->qb = 0, r->rb = rbeg - o.gtle;
			truesc = o.gscore;
		}
		free(qs);
		free(rs);
	} else score = truesc = len * p->a, r->qb = 0, r->rb = rbeg; /* bwamem.c:839 */

	if (qbeg + len != l_query) { /* right extension.  bwamem.c:841-865 */
		int sc0 = score, qe = qbeg + len, re = rbeg + len;
		orc_extend_out_t o;
		memset(&o, 0, sizeof(o));
		for (i = 0; i < 2; ++i) {
			int prev = score;
			aw1 = p->w << i;
			orc_extend(&sc, l_query - qe, query + qe, wlen - re, rseq + re, aw1, p->pen_clip3, sc0, &o, &st);
			++n_ext;
			if (cells) *cells += st.cells;
			score = o.score;
			if (score == prev || o.max_off < (aw1 >> 1) + (aw1 >> 2)) break;
		}
		if (o.gscore <= 0 || o.gscore <= score - p->pen_clip3) {
			r->qe = qe + o.qle, r->re = re + o.tle;
			truesc += score - sc0;
		} else {
			r->qe = l_query, r->re = re + o.gtle;
			truesc += o.gscore - sc0;
		}
	} else r->qe = l_query, r->re = rbeg + len; /* bwamem.c:866 */
	r->score = score, r->truesc = truesc, r->w = imax(aw0, aw1), r->n_ext = n_ext;
	if (calls) *calls += n_ext;
}

/* ---- batch form over bmh_seed_task_t records (tests, bench CPU baseline): plain loop, optionally on pthreads */
#include <pthread.h>
typedef struct {
	const bmh_params_t *p;
	const uint8_t *pool, *pac;
	int64_t l_pac;
	const bmh_seed_task_t *t;
	bmh_seed_result_t *r;
	int64_t lo, hi, cells, calls;
} seed_job_t;

static void *seed_worker(void *arg)
{
	seed_job_t *j = (seed_job_t *)arg;
	int64_t k;
	for (k = j->lo; k < j->hi; ++k) {
		const bmh_seed_task_t *t = &j->t[k];
		const uint8_t *rseq = j->pool + t->t_off;
		uint8_t *own = 0;
		if (t->flags & BMH_F_TPAC) { /* window read from the 2-bit reference: bns_get_seq, bntseq.c:355-376 */
			int64_t len;
			own = orc_get_seq(j->l_pac, j->pac, (int64_t)t->t_off, (int64_t)t->t_off + t->wlen, &len);
			rseq = own;
		}
		orc_seedext_one(j->p, j->pool + t->q_off, t->l_query, rseq, t->wlen, t->qbeg, t->len, t->rbeg, &j->r[k], &j->cells, &j->calls);
		free(own);
	}
	return 0;
}

int orc_seedext_batch(const bmh_params_t *p, const uint8_t *pool, const uint8_t *pac, int64_t l_pac, const bmh_seed_task_t *tasks,
                      int64_t n, bmh_seed_result_t *results, int64_t *cells_out, int64_t *calls_out, int nthreads)
{
	pthread_t th[256];
	seed_job_t job[256];
	int k;
	if (nthreads < 1) nthreads = 1;
	if (nthreads > 256) nthreads = 256;
	for (k = 0; k < nthreads; ++k) {
		job[k].p = p, job[k].pool = pool, job[k].pac = pac, job[k].l_pac = l_pac, job[k].t = tasks, job[k].r = results;
		job[k].lo = n * k / nthreads, job[k].hi = n * (k + 1) / nthreads, job[k].cells = job[k].calls = 0;
		if (nthreads > 1) pthread_create(&th[k], 0, seed_worker, &job[k]);
		else seed_worker(&job[k]);
	}
	if (cells_out) *cells_out = 0;
	if (calls_out) *calls_out = 0;
	for (k = 0; k < nthreads; ++k) {
		if (nthreads > 1) pthread_join(th[k], 0);
		if (cells_out) *cells_out += job[k].cells;
		if (calls_out) *calls_out += job[k].calls;
	}
	return 0;
}

void orc_chain2aln(const bmh_params_t *p, int64_t l_pac, const uint8_t *pac, int l_query,
                   const uint8_t *query, const bmh_chain_t *c, bmh_alnreg_v *av,
                   orc_driver_trace_t *trace)
{
	int i, k;
	int64_t rmax0, rmax1, rlen;
	uint8_t *rseq;
	uint64_t *srt;

	if (c->n == 0) return; /* bwamem.c:738 */

	/* reference window covering every seed's maximal extension.  bwamem.c:740-755 */
	rmax0 = l_pac << 1, rmax1 = 0;
	for (i = 0; i < c->n; ++i) {
		const bmh_seed_t *t = &c->seeds[i];
		int rest = l_query - t->qbeg - t->len;
		int64_t b = t->rbeg - (t->qbeg + orc_cal_max_gap(p, t->qbeg));
		int64_t e = t->rbeg + t->len + (rest + orc_cal_max_gap(p, rest));
		if (b < rmax0) rmax0 = b;
		if (e > rmax1) rmax1 = e;
	}
	if (rmax0 < 0) rmax0 = 0;
	if (rmax1 > l_pac << 1) rmax1 = l_pac << 1;
	if (rmax0 < l_pac && l_pac < rmax1) { /* never straddle the strand boundary */
		if (c->seeds[0].rbeg < l_pac) rmax1 = l_pac;
		else rmax0 = l_pac;
	}
	rseq = orc_get_seq(l_pac, pac, rmax0, rmax1, &rlen); /* bwamem.c:757 */

	/* longest seed first (ties: larger index first).  bwamem.c:760-765 */
	srt = (uint64_t *)malloc((size_t)c->n * 8);
	for (i = 0; i < c->n; ++i) srt[i] = (uint64_t)c->seeds[i].len << 32 | (uint32_t)i;
	qsort(srt, (size_t)c->n, 8, cmp_u64);

	for (k = c->n - 1; k >= 0; --k) {
		const bmh_seed_t *s = &c->seeds[(uint32_t)srt[k]];
		bmh_alnreg_t *a;
		int aw0, aw1;

		if (seed_near_region(p, s, av)) { /* bwamem.c:785-802 */
			for (i = k + 1; i < c->n; ++i) {
				const bmh_seed_t *t;
				if (srt[i] == 0) continue; /* bwamem.c:790 (also skips seed #0 of length 0) */
				t = &c->seeds[(uint32_t)srt[i]];
				if (t->len < s->len * .95) continue; /* double compare, bwamem.c:792 */
				if (s->qbeg <= t->qbeg && s->qbeg + s->len - t->qbeg >= s->len >> 2 && t->qbeg - s->qbeg != t->rbeg - s->rbeg) break;
				if (t->qbeg <= s->qbeg && t->qbeg + t->len - s->qbeg >= s->len >> 2 && s->qbeg - t->qbeg != s->rbeg - t->rbeg) break;
			}
			if (i == c->n) {
				srt[k] = 0;
				if (trace) trace->seeds_skipped++;
				continue;
			}
		}
		if (trace) trace->seeds_extended++;

		a = regs_push(av); /* bwamem.c:804-807 */
		memset(a, 0, sizeof(*a));
		{ /* both extensions of the seed and their decisions: bwamem.c:808-866, see orc_seedext_one */
			bmh_seed_result_t sr;
			int64_t calls = 0;
			orc_seedext_one(p, query, l_query, rseq, (int)(rmax1 - rmax0), s->qbeg, s->len, (int)(s->rbeg - rmax0), &sr, 0, &calls);
			if (trace) trace->ext_calls += calls;
			a->qb = sr.qb, a->qe = sr.qe, a->rb = rmax0 + sr.rb, a->re = rmax0 + sr.re;
			a->score = sr.score, a->truesc = sr.truesc;
			aw0 = sr.w, aw1 = sr.w; /* a->w = max(aw0, aw1) below */
		}

		/* seed coverage.  bwamem.c:870-874 */
		for (i = 0, a->seedcov = 0; i < c->n; ++i) {
			const bmh_seed_t *t = &c->seeds[i];
			if (t->qbeg >= a->qb && t->qbeg + t->len <= a->qe && t->rbeg >= a->rb && t->rbeg + t->len <= a->re)
				a->seedcov += t->len;
		}
		a->w = imax(aw0, aw1); /* bwamem.c:875 */
	}
	free(srt);
	free(rseq);
}
