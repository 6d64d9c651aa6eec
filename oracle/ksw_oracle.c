/*
 * oracle/ksw_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * See ksw_oracle.h for scope, import rules and parity status (PINNED against
 * the compiled reference, oracle/_ref/).
 *
 * Written from SURVEY.md Appendix A.1 / A.2 (the behavioural spec), with the
 * reference lines each step follows cited as bwa-0.7.8/ksw.c:NNN.
 */
#include "ksw_oracle.h"
#include "../include/bwamem_hip.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

/* Largest gap length still worth opening for a query of `qlen` bases:
 * (int)((double)(qlen*mx + end_bonus - o) / e + 1.), floored at 1.  ksw.c:401-405 */
static int band_cap(int qlen, int mx, int end_bonus, int o, int e)
{
	int v = (int)((double)(qlen * mx + end_bonus - o) / e + 1.);
	return v > 1 ? v : 1;
}

/* ------------------------------------------------------------------ extend */

void orc_extend(const orc_scoring_t *sc, int qlen, const uint8_t *query,
                int tlen, const uint8_t *target, int w, int end_bonus, int h0,
                orc_extend_out_t *out, orc_extend_stats_t *stats)
{
	const int m = sc->m;
	const int oe_del = sc->o_del + sc->e_del, oe_ins = sc->o_ins + sc->e_ins;
	const int e_del = sc->e_del, e_ins = sc->e_ins;
	/* Hs[j] holds H(i-1, j-1) ("shifted" H), E[j] holds E(i, j) when row i starts. ksw.c:422 */
	int *Hs = (int *)calloc((size_t)qlen + 2, sizeof(int));
	int *E = (int *)calloc((size_t)qlen + 2, sizeof(int));
	int i, j, mx = 0;
	int best, bi = -1, bj = -1, gi = -1, gscore = -1, max_off = 0, beg = 0, end = qlen;
	int64_t cells = 0;
	int rows = 0;

	if (h0 < 0) h0 = 0; /* ksw.c:384 */

	/* first row: gap-in-query from the seed end.  ksw.c:394-396 */
	Hs[0] = h0;
	if (qlen >= 1) Hs[1] = h0 > oe_ins ? h0 - oe_ins : 0;
	for (j = 2; j <= qlen && Hs[j - 1] > e_ins; ++j) Hs[j] = Hs[j - 1] - e_ins;

	/* band clamp.  ksw.c:398-406 */
	for (i = 0; i < m * m; ++i) mx = imax(mx, sc->mat[i]);
	w = imin(w, band_cap(qlen, mx, end_bonus, sc->o_ins, e_ins));
	w = imin(w, band_cap(qlen, mx, end_bonus, sc->o_del, e_del));

	best = h0; /* ksw.c:408 */
	for (i = 0; i < tlen; ++i) {
		const int8_t *srow = sc->mat + (size_t)target[i] * m;
		int f = 0, rowmax = 0, mj = -1;
		/* first-column value H(i,-1) injected at j=beg whatever beg is.  ksw.c:415-416 */
		int left = imax(0, h0 - (sc->o_del + e_del * (i + 1)));
		beg = imax(beg, i - w);               /* ksw.c:418 */
		end = imin(imin(end, i + w + 1), qlen); /* ksw.c:419-420 */
		++rows;
		for (j = beg; j < end; ++j) { /* ksw.c:421-445 */
			int diag = Hs[j], e = E[j], h;
			Hs[j] = left;
			h = imax(imax(diag + srow[query[j]], e), f);
			left = h;
			if (!(rowmax > h)) mj = j; /* ties -> larger j; rowmax starts at 0.  ksw.c:434 */
			rowmax = imax(rowmax, h);
			E[j] = imax(e - e_del, imax(h - oe_del, 0));
			f = imax(f - e_ins, imax(h - oe_ins, 0));
			++cells;
		}
		/* after the loop j == end if the row was non-empty, else j == beg */
		Hs[end] = left; /* ksw.c:446 */
		E[end] = 0;
		if (j == qlen) { /* ksw.c:447-450: ties -> later i */
			if (!(gscore > left)) gi = i;
			gscore = imax(gscore, left);
		}
		if (rowmax == 0) break; /* ksw.c:451 */
		if (rowmax > best) {    /* ksw.c:452-454 */
			best = rowmax, bi = i, bj = mj;
			max_off = imax(max_off, abs(mj - i));
		} else if (sc->zdrop > 0) { /* ksw.c:455-461 */
			int di = i - bi, dj = mj - bj;
			if (di > dj) {
				if (best - rowmax - (di - dj) * e_del > sc->zdrop) break;
			} else {
				if (best - rowmax - (dj - di) * e_ins > sc->zdrop) break;
			}
		}
		/* shrink/grow the live interval around mj.  ksw.c:463-466 */
		for (j = mj; j >= beg && Hs[j]; --j) {}
		beg = j + 1;
		for (j = mj + 2; j <= end && Hs[j]; ++j) {}
		end = j;
	}
	free(Hs);
	free(E);
	out->score = best;
	out->qle = bj + 1;
	out->tle = bi + 1;
	out->gtle = gi + 1;
	out->gscore = gscore;
	out->max_off = max_off;
	if (stats) stats->cells = cells, stats->rows = rows;
}

/* ------------------------------------------------------------------ global */

#define NEG_INF (-0x40000000) /* ksw.c:487 */

typedef struct {
	uint32_t *a;
	int n, cap;
} cigar_buf_t;

/* run-length append, ksw.c:489-499 */
static void cigar_push(cigar_buf_t *c, int op, int len)
{
	if (c->n && (int)(c->a[c->n - 1] & 0xf) == op) {
		c->a[c->n - 1] += (uint32_t)len << 4;
		return;
	}
	if (c->n == c->cap) {
		c->cap = c->cap ? c->cap * 2 : 4;
		c->a = (uint32_t *)realloc(c->a, (size_t)c->cap * 4);
	}
	c->a[c->n++] = (uint32_t)len << 4 | (uint32_t)op;
}

int orc_global(const orc_scoring_t *sc, int qlen, const uint8_t *query,
               int tlen, const uint8_t *target, int w, int *n_cigar,
               uint32_t **cigar)
{
	const int m = sc->m;
	const int oe_del = sc->o_del + sc->e_del, oe_ins = sc->o_ins + sc->e_ins;
	const int e_del = sc->e_del, e_ins = sc->e_ins;
	const int n_col = imin(qlen, 2 * w + 1); /* ksw.c:509 */
	uint8_t *z = (uint8_t *)malloc((size_t)imax(n_col, 1) * (size_t)imax(tlen, 1));
	int *Hs = (int *)malloc(((size_t)qlen + 2) * sizeof(int));
	int *E = (int *)malloc(((size_t)qlen + 2) * sizeof(int));
	int i, j, score;

	if (n_cigar) *n_cigar = 0; /* ksw.c:507 */

	/* first row.  ksw.c:519-522 */
	Hs[0] = 0, E[0] = NEG_INF;
	for (j = 1; j <= qlen && j <= w; ++j) Hs[j] = -(sc->o_ins + e_ins * j), E[j] = NEG_INF;
	for (; j <= qlen; ++j) Hs[j] = E[j] = NEG_INF;

	for (i = 0; i < tlen; ++i) { /* ksw.c:524-564 */
		const int8_t *srow = sc->mat + (size_t)target[i] * m;
		const int beg = i > w ? i - w : 0;
		const int end = i + w + 1 < qlen ? i + w + 1 : qlen;
		int f = NEG_INF;
		int left = beg == 0 ? -(sc->o_del + e_del * (i + 1)) : NEG_INF; /* ksw.c:530 */
		uint8_t *zi = z + (size_t)i * n_col;
		for (j = beg; j < end; ++j) {
			int mm = Hs[j] + srow[query[j]], e = E[j], h, t;
			uint8_t d;
			Hs[j] = left;
			d = mm >= e ? 0 : 1; /* ksw.c:547-550 */
			h = mm >= e ? mm : e;
			d = h >= f ? d : 2;
			h = h >= f ? h : f;
			left = h;
			t = mm - oe_del; /* gaps open from the diagonal score, ksw.c:552-560 */
			e -= e_del;
			if (e > t) d |= 1 << 2;
			E[j] = e > t ? e : t;
			t = mm - oe_ins;
			f -= e_ins;
			if (f > t) d |= 2 << 4;
			f = f > t ? f : t;
			zi[j - beg] = d;
		}
		Hs[end] = left, E[end] = NEG_INF; /* ksw.c:563 */
	}
	score = Hs[qlen]; /* ksw.c:565 */

	if (n_cigar && cigar) { /* backtrack, ksw.c:566-581 */
		cigar_buf_t c = {0, 0, 0};
		int which = 0, k;
		i = tlen - 1;
		k = imin(qlen, i + w + 1) - 1;
		while (i >= 0 && k >= 0) {
			int off = k - (i > w ? i - w : 0);
			which = z[(size_t)i * n_col + off] >> (which << 1) & 3;
			if (which == 0) cigar_push(&c, 0, 1), --i, --k;
			else if (which == 1) cigar_push(&c, 2, 1), --i;
			else cigar_push(&c, 1, 1), --k;
		}
		if (i >= 0) cigar_push(&c, 2, i + 1);
		if (k >= 0) cigar_push(&c, 1, k + 1);
		for (i = 0; i < c.n >> 1; ++i) {
			uint32_t t = c.a[i];
			c.a[i] = c.a[c.n - 1 - i], c.a[c.n - 1 - i] = t;
		}
		*n_cigar = c.n, *cigar = c.a;
	}
	free(Hs);
	free(E);
	free(z);
	return score;
}

/* ------------------------------------------------------------ batch helper */

typedef struct {
	const orc_scoring_t *sc;
	const uint8_t *pool;
	const uint8_t *pac; /* 2-bit reference for BMH_F_TPAC tasks, or NULL */
	int64_t l_pac;
	const bmh_ext_task_t *tasks;
	bmh_ext_result_t *res;
	int lo, hi;
	int64_t cells;
} ext_job_t;

/* one base of bwa's doubled coordinate: forward strand pac[p], reverse strand the complement read backwards
 * (what bns_get_seq returns base by base, reference bntseq.c:355-376) */
static int pac_base(const uint8_t *pac, int64_t l_pac, int64_t p)
{
	if (p < l_pac) return pac[p >> 2] >> ((~p & 3) << 1) & 3;
	p = (l_pac << 1) - 1 - p;
	return 3 - (pac[p >> 2] >> ((~p & 3) << 1) & 3);
}

static void fetch_seq(const uint8_t *pool, uint64_t off, int len, int rev, uint8_t *dst)
{
	int k;
	if (!rev) memcpy(dst, pool + off, (size_t)len);
	else for (k = 0; k < len; ++k) dst[k] = pool[off - (uint64_t)k];
}

static void *ext_job_run(void *p)
{
	ext_job_t *job = (ext_job_t *)p;
	uint8_t *q = (uint8_t *)malloc(65536), *t = (uint8_t *)malloc(65536);
	int k;
	for (k = job->lo; k < job->hi; ++k) {
		const bmh_ext_task_t *tk = &job->tasks[k];
		orc_extend_out_t o;
		orc_extend_stats_t st;
		fetch_seq(job->pool, tk->q_off, tk->qlen, tk->flags & BMH_F_QREV, q);
		if (tk->flags & BMH_F_TPAC) {
			int x;
			for (x = 0; x < tk->tlen; ++x)
				t[x] = (uint8_t)pac_base(job->pac, job->l_pac, tk->flags & BMH_F_TREV ? (int64_t)tk->t_off - x : (int64_t)tk->t_off + x);
		} else fetch_seq(job->pool, tk->t_off, tk->tlen, tk->flags & BMH_F_TREV, t);
		orc_extend(job->sc, tk->qlen, q, tk->tlen, t, tk->w, tk->end_bonus, tk->h0, &o, &st);
		job->res[k].score = o.score, job->res[k].qle = o.qle, job->res[k].tle = o.tle;
		job->res[k].gtle = o.gtle, job->res[k].gscore = o.gscore, job->res[k].max_off = o.max_off;
		job->cells += st.cells;
	}
	free(q);
	free(t);
	return 0;
}

int orc_extend_batch(const orc_scoring_t *sc, const uint8_t *seqpool,
                     const struct bmh_ext_task *tasks, int n,
                     struct bmh_ext_result *results, int64_t *cells_out,
                     int nthreads)
{
	return orc_extend_batch_pac(sc, seqpool, 0, 0, tasks, n, results, cells_out, nthreads);
}

int orc_extend_batch_pac(const orc_scoring_t *sc, const uint8_t *seqpool, const uint8_t *pac, int64_t l_pac,
                         const struct bmh_ext_task *tasks, int n, struct bmh_ext_result *results, int64_t *cells_out,
                         int nthreads)
{
	int i;
	int64_t cells = 0;
	if (nthreads < 1) nthreads = 1;
	if (nthreads > 256) nthreads = 256;
	{
		ext_job_t jobs[256];
		pthread_t tid[256];
		for (i = 0; i < nthreads; ++i) {
			jobs[i].sc = sc, jobs[i].pool = seqpool, jobs[i].tasks = tasks, jobs[i].res = results;
			jobs[i].pac = pac, jobs[i].l_pac = l_pac;
			jobs[i].lo = (int)((int64_t)n * i / nthreads);
			jobs[i].hi = (int)((int64_t)n * (i + 1) / nthreads);
			jobs[i].cells = 0;
		}
		if (nthreads == 1) ext_job_run(&jobs[0]);
		else {
			for (i = 0; i < nthreads; ++i) pthread_create(&tid[i], 0, ext_job_run, &jobs[i]);
			for (i = 0; i < nthreads; ++i) pthread_join(tid[i], 0);
		}
		for (i = 0; i < nthreads; ++i) cells += jobs[i].cells;
	}
	if (cells_out) *cells_out = cells;
	return 0;
}

/* ------------------------------------------------------- global batch helper */

typedef struct {
	const orc_scoring_t *sc;
	const uint8_t *pool;
	const bmh_glb_task_t *tasks;
	bmh_glb_result_t *res;
	uint32_t *cigar_pool;
	int lo, hi;
	int64_t cells;
} glb_job_t;

static void *glb_job_run(void *p)
{
	glb_job_t *job = (glb_job_t *)p;
	int k;
	for (k = job->lo; k < job->hi; ++k) {
		const bmh_glb_task_t *tk = &job->tasks[k];
		int n = 0, i, w = tk->w, ncol = tk->qlen < 2 * w + 1 ? tk->qlen : 2 * w + 1;
		uint32_t *cg = 0;
		job->res[k].score = orc_global(job->sc, tk->qlen, job->pool + tk->q_off, tk->tlen, job->pool + tk->t_off, w,
		                               tk->cigar_cap ? &n : 0, tk->cigar_cap ? &cg : 0);
		job->res[k].n_cigar = n;
		for (i = 0; i < n && i < (int)tk->cigar_cap; ++i) job->cigar_pool[tk->cigar_off + i] = cg[i];
		free(cg);
		job->cells += (int64_t)ncol * tk->tlen; /* band cells, ksw.c:528-529 upper bound */
	}
	return 0;
}

int orc_global_batch(const orc_scoring_t *sc, const uint8_t *seqpool, const struct bmh_glb_task *tasks, int n,
                     struct bmh_glb_result *results, uint32_t *cigar_pool, int64_t *cells_out, int nthreads)
{
	int i;
	int64_t cells = 0;
	glb_job_t jobs[256];
	pthread_t tid[256];
	if (nthreads < 1) nthreads = 1;
	if (nthreads > 256) nthreads = 256;
	for (i = 0; i < nthreads; ++i) {
		jobs[i].sc = sc, jobs[i].pool = seqpool, jobs[i].tasks = tasks, jobs[i].res = results, jobs[i].cigar_pool = cigar_pool;
		jobs[i].lo = (int)((int64_t)n * i / nthreads), jobs[i].hi = (int)((int64_t)n * (i + 1) / nthreads), jobs[i].cells = 0;
	}
	if (nthreads == 1) glb_job_run(&jobs[0]);
	else {
		for (i = 0; i < nthreads; ++i) pthread_create(&tid[i], 0, glb_job_run, &jobs[i]);
		for (i = 0; i < nthreads; ++i) pthread_join(tid[i], 0);
	}
	for (i = 0; i < nthreads; ++i) cells += jobs[i].cells;
	if (cells_out) *cells_out = cells;
	return 0;
}
