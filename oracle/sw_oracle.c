/*
 * oracle/sw_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See sw_oracle.h.
 *
 * Written from the behaviour of bwa-0.7.8/ksw.c:62-364; no SIMD, no code shared with the reference.
 */
#include "sw_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "../include/bwamem_hip.h"

static inline int imax2(int a, int b) { return a > b ? a : b; }

typedef struct {
	int score, te, qe, score2, te2;
} core_out_t;

/* One pass of ksw_u8 (size 1) / ksw_i16 (size 2) over target rows 0..tlen-1, as a scalar emulation of the p
 * vector lanes: lane l owns the query segment [l*slen, (l+1)*slen) (ksw.c:87-89); X[j*p+l] below is element l of
 * vector j.  The lazy-F loop is kept literally, early exit included: with a zero gap-open penalty it stops after
 * the first column and the result is NOT the textbook DP. */
static core_out_t sw_core(int size, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m,
                          const int8_t *mat, int o_del, int e_del, int o_ins, int e_ins, int xtra, int64_t *cells)
{
	const int p = size == 1 ? 16 : 8;      /* values per vector, ksw.c:68 */
	const int slen = (qlen + p - 1) / p;   /* segment length, ksw.c:69 */
	const int Q = slen * p;                /* padded query length */
	const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
	const int minsc = (xtra & ORC_XSUBO) ? (xtra & 0xffff) : 0x10000; /* ksw.c:131-132 */
	const int endsc = (xtra & ORC_XSTOP) ? (xtra & 0xffff) : 0x10000;
	uint8_t shift = 127, mdiff = 0;
	int a, i, j, l, k, qmax, gmax = 0, te = -1, n_b = 0, m_b = 0;
	uint64_t *b = 0;
	int *H0 = (int *)calloc((size_t)Q + 1, sizeof(int)), *H1 = (int *)calloc((size_t)Q + 1, sizeof(int));
	int *E = (int *)calloc((size_t)Q + 1, sizeof(int)), *Hmax = (int *)calloc((size_t)Q + 1, sizeof(int));
	int f[16], h[16];
	core_out_t r = {0, -1, -1, -1, -1};

	for (a = 0; a < m * m; ++a) { /* ksw.c:78-85: bias and largest score */
		if (mat[a] < (int8_t)shift) shift = (uint8_t)mat[a];
		if (mat[a] > (int8_t)mdiff) mdiff = (uint8_t)mat[a];
	}
	qmax = mdiff;
	shift = (uint8_t)(256 - shift);

	for (i = 0; i < tlen && slen > 0; ++i) {
		const int8_t *row = mat + (int)target[i] * m;
		int imax = 0;
		for (l = 0; l < p; ++l) f[l] = 0, h[l] = l ? H0[(slen - 1) * p + l - 1] : 0; /* H(i-1,j-1): last vector shifted by one lane, ksw.c:140-141 */
		for (j = 0; j < slen; ++j) /* main loop, ksw.c:142-164 */
			for (l = 0; l < p; ++l) {
				const int qi = l * slen + j;
				const int s = qi < qlen ? row[query[qi]] : 0; /* pad columns score 0, ksw.c:98,107 */
				int v, e = E[j * p + l], t;
				if (size == 1) { /* biased unsigned bytes, ksw.c:149-150 */
					v = h[l] + (uint8_t)(s + shift);
					if (v > 255) v = 255;
					v = imax2(v - shift, 0);
				} else v = h[l] + s; /* signed words; the max with e >= 0 below clamps, ksw.c:257-259 */
				v = imax2(imax2(v, e), f[l]);
				imax = imax2(imax, v);
				H1[j * p + l] = v;
				t = imax2(v - oe_del, 0);
				E[j * p + l] = imax2(imax2(e - e_del, 0), t);
				t = imax2(v - oe_ins, 0);
				f[l] = imax2(imax2(f[l] - e_ins, 0), t);
				h[l] = H0[j * p + l];
			}
		for (k = 0; k < 16; ++k) { /* lazy F, ksw.c:165-176 */
			for (l = p - 1; l > 0; --l) f[l] = f[l - 1];
			f[0] = 0;
			for (j = 0; j < slen; ++j) {
				int all = 1;
				for (l = 0; l < p; ++l) {
					int v = imax2(H1[j * p + l], f[l]);
					H1[j * p + l] = v;
					v = imax2(v - oe_ins, 0);
					f[l] = imax2(f[l] - e_ins, 0);
					if (f[l] > v) all = 0;
				}
				if (all) goto lazy_done;
			}
		}
lazy_done:
		if (cells) *cells += qlen;
		if (imax >= minsc) { /* ksw.c:181-189: runs of rows above the threshold */
			if (n_b == 0 || (int32_t)b[n_b - 1] + 1 != i) {
				if (n_b == m_b) {
					m_b = m_b ? m_b << 1 : 8;
					b = (uint64_t *)realloc(b, 8 * (size_t)m_b);
				}
				b[n_b++] = (uint64_t)imax << 32 | (uint32_t)i;
			} else if ((int)(b[n_b - 1] >> 32) < imax) b[n_b - 1] = (uint64_t)imax << 32 | (uint32_t)i;
		}
		if (imax > gmax) { /* ksw.c:190-195 */
			gmax = imax, te = i;
			memcpy(Hmax, H1, sizeof(int) * (size_t)Q);
			if ((size == 1 && gmax + shift >= 255) || gmax >= endsc) break;
		}
		{ int *tmp = H1; H1 = H0; H0 = tmp; }
	}
	r.score = (size == 1 && gmax + shift >= 255) ? 255 : gmax; /* ksw.c:198 */
	r.te = te;
	if (!(size == 1 && r.score == 255)) { /* ksw.c:200-221 */
		int mx = -1, low, high;
		for (j = 0; j < slen; ++j) /* smallest query index among the maxima, ksw.c:204-206 */
			for (l = 0; l < p; ++l) {
				const int qi = l * slen + j, v = Hmax[j * p + l];
				if (v > mx || (v == mx && qi < r.qe)) mx = v, r.qe = qi;
			}
		if (b) {
			i = (r.score + qmax - 1) / qmax;
			low = te - i, high = te + i;
			for (i = 0; i < n_b; ++i) {
				const int e = (int32_t)b[i];
				if ((e < low || e > high) && (int)(b[i] >> 32) > r.score2) r.score2 = (int)(b[i] >> 32), r.te2 = e;
			}
		}
	}
	free(b), free(H0), free(H1), free(E), free(Hmax);
	return r;
}

orc_kswr_t orc_align2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                      int o_del, int e_del, int o_ins, int e_ins, int xtra, int *undefined, int64_t *cells)
{
	const int size = (xtra & ORC_XBYTE) ? 1 : 2; /* ksw.c:348 */
	orc_kswr_t r = {0, -1, -1, -1, -1, -1, -1};  /* g_defr, ksw.c:43 */
	core_out_t f, rr;
	uint8_t *rq, *rt;
	int k;
	if (undefined) *undefined = 0;
	f = sw_core(size, qlen, query, tlen, target, m, mat, o_del, e_del, o_ins, e_ins, xtra, cells);
	r.score = f.score, r.te = f.te, r.qe = f.qe, r.score2 = f.score2, r.te2 = f.te2;
	if ((xtra & ORC_XSTART) == 0 || ((xtra & ORC_XSUBO) && r.score < (xtra & 0xffff))) return r; /* ksw.c:354 */
	if (r.qe < 0) { /* byte overflow: the reference goes on with a zero-length query -- undefined */
		if (undefined) *undefined = 1;
		return r;
	}
	/* second pass over the reversed prefixes; the target keeps its untouched tail (ksw.c:355-357 pass tlen) */
	rq = (uint8_t *)malloc((size_t)r.qe + 2), rt = (uint8_t *)malloc((size_t)tlen + 1);
	for (k = 0; k <= r.qe; ++k) rq[k] = query[r.qe - k];
	for (k = 0; k < tlen; ++k) rt[k] = k <= r.te ? target[r.te - k] : target[k];
	rr = sw_core(size, r.qe + 1, rq, tlen, rt, m, mat, o_del, e_del, o_ins, e_ins, ORC_XSTOP | r.score, cells);
	free(rq), free(rt);
	if (r.score == rr.score) r.tb = r.te - rr.te, r.qb = r.qe - rr.qe; /* ksw.c:360-361 */
	return r;
}

/* ------------------------------------------------------------------ batch helper */

typedef struct {
	const bmh_params_t *p;
	const uint8_t *pool, *pac;
	int64_t l_pac;
	const bmh_sw_task_t *tasks;
	bmh_sw_result_t *res;
	int lo, hi;
	int64_t cells;
} sw_job_t;

static int pac_base(const uint8_t *pac, int64_t l_pac, int64_t p) /* bntseq.c:355-376 base by base */
{
	if (p < l_pac) return pac[p >> 2] >> ((~p & 3) << 1) & 3;
	p = (l_pac << 1) - 1 - p;
	return 3 - (pac[p >> 2] >> ((~p & 3) << 1) & 3);
}

static void *sw_job_run(void *ptr)
{
	sw_job_t *job = (sw_job_t *)ptr;
	const bmh_params_t *p = job->p;
	int k;
	for (k = job->lo; k < job->hi; ++k) {
		const bmh_sw_task_t *tk = &job->tasks[k];
		uint8_t *q = (uint8_t *)malloc((size_t)tk->qlen + 1), *t = (uint8_t *)malloc((size_t)tk->tlen + 1);
		orc_kswr_t r;
		int und = 0;
		uint32_t x;
		for (x = 0; x < tk->qlen; ++x) {
			int c = tk->flags & BMH_F_QREV ? job->pool[tk->q_off - x] : job->pool[tk->q_off + x];
			if (tk->flags & BMH_F_QCOMP) c = c < 4 ? 3 - c : 4;
			q[x] = (uint8_t)c;
		}
		for (x = 0; x < tk->tlen; ++x) {
			if (tk->flags & BMH_F_TPAC)
				t[x] = (uint8_t)pac_base(job->pac, job->l_pac, tk->flags & BMH_F_TREV ? (int64_t)tk->t_off - x : (int64_t)tk->t_off + x);
			else t[x] = tk->flags & BMH_F_TREV ? job->pool[tk->t_off - x] : job->pool[tk->t_off + x];
		}
		r = orc_align2(tk->qlen, q, (int)tk->tlen, t, 5, p->mat, p->o_del, p->e_del, p->o_ins, p->e_ins, (int)tk->xtra, &und,
		               &job->cells);
		job->res[k].score = r.score, job->res[k].te = r.te, job->res[k].qe = r.qe, job->res[k].score2 = r.score2;
		job->res[k].te2 = r.te2, job->res[k].tb = r.tb, job->res[k].qb = r.qb, job->res[k].rsv = und;
		free(q), free(t);
	}
	return 0;
}

int orc_sw_batch(const struct bmh_params *p, const uint8_t *pool, const uint8_t *pac, int64_t l_pac,
                 const struct bmh_sw_task *tasks, int n, struct bmh_sw_result *res, int64_t *cells_out, int nthreads)
{
	sw_job_t jobs[256];
	pthread_t tid[256];
	int i;
	int64_t cells = 0;
	if (nthreads < 1) nthreads = 1;
	if (nthreads > 256) nthreads = 256;
	for (i = 0; i < nthreads; ++i) {
		jobs[i].p = p, jobs[i].pool = pool, jobs[i].pac = pac, jobs[i].l_pac = l_pac, jobs[i].tasks = tasks;
		jobs[i].res = res, jobs[i].cells = 0;
		jobs[i].lo = (int)((int64_t)n * i / nthreads), jobs[i].hi = (int)((int64_t)n * (i + 1) / nthreads);
	}
	if (nthreads == 1) sw_job_run(&jobs[0]);
	else {
		for (i = 0; i < nthreads; ++i) pthread_create(&tid[i], 0, sw_job_run, &jobs[i]);
		for (i = 0; i < nthreads; ++i) pthread_join(tid[i], 0);
	}
	for (i = 0; i < nthreads; ++i) cells += jobs[i].cells;
	if (cells_out) *cells_out = cells;
	return 0;
}
