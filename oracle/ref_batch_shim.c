/*
 * oracle/ref_batch_shim.c -- TEST INFRASTRUCTURE.  Ours, not the reference's: a pthread loop that calls the
 * REFERENCE's ksw_align2 (compiled from /root/reference/bwa-0.7.8/ksw.c into the same oracle/_ref/libksw_ref.so)
 * over bmh_sw_task_t records, so that bench.py can time the reference's own SSE2 kernel on all host cores
 * (cpu_baseline kind "reference") without a Python call per task.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>

#include "../include/bwamem_hip.h"

typedef struct { int score, te, qe, score2, te2, tb, qb; } ref_kswr_t; /* kswr_t, reference ksw.h:14-19 */
struct _kswq_t;
extern ref_kswr_t ksw_align2(int qlen, uint8_t *query, int tlen, uint8_t *target, int m, const int8_t *mat, int o_del,
                             int e_del, int o_ins, int e_ins, int xtra, struct _kswq_t **qry); /* reference ksw.h:62 */

typedef struct {
	const bmh_params_t *p;
	const uint8_t *pool;
	const bmh_sw_task_t *tasks;
	bmh_sw_result_t *res;
	int lo, hi;
} job_t;

static void *run(void *ptr)
{
	job_t *j = (job_t *)ptr;
	int k;
	for (k = j->lo; k < j->hi; ++k) {
		const bmh_sw_task_t *t = &j->tasks[k];
		uint8_t *q = (uint8_t *)malloc((size_t)t->qlen + 1), *tg = (uint8_t *)malloc((size_t)t->tlen + 1);
		uint32_t x;
		ref_kswr_t r;
		for (x = 0; x < t->qlen; ++x) {
			int c = t->flags & BMH_F_QREV ? j->pool[t->q_off - x] : j->pool[t->q_off + x];
			q[x] = (uint8_t)((t->flags & BMH_F_QCOMP) && c < 4 ? 3 - c : c);
		}
		for (x = 0; x < t->tlen; ++x) tg[x] = t->flags & BMH_F_TREV ? j->pool[t->t_off - x] : j->pool[t->t_off + x];
		r = ksw_align2(t->qlen, q, (int)t->tlen, tg, 5, j->p->mat, j->p->o_del, j->p->e_del, j->p->o_ins, j->p->e_ins,
		               (int)t->xtra, 0);
		j->res[k].score = r.score, j->res[k].te = r.te, j->res[k].qe = r.qe, j->res[k].score2 = r.score2;
		j->res[k].te2 = r.te2, j->res[k].tb = r.tb, j->res[k].qb = r.qb, j->res[k].rsv = 0;
		free(q), free(tg);
	}
	return 0;
}

/* BMH_F_TPAC tasks are not supported here (the shim has no reference sequence); callers pass pool-addressed tasks. */
int ref_sw_batch_mt(const bmh_params_t *p, const uint8_t *pool, const bmh_sw_task_t *tasks, int n, bmh_sw_result_t *res,
                    int nthreads)
{
	job_t jobs[256];
	pthread_t tid[256];
	int i;
	if (nthreads < 1) nthreads = 1;
	if (nthreads > 256) nthreads = 256;
	for (i = 0; i < nthreads; ++i) {
		jobs[i].p = p, jobs[i].pool = pool, jobs[i].tasks = tasks, jobs[i].res = res;
		jobs[i].lo = (int)((int64_t)n * i / nthreads), jobs[i].hi = (int)((int64_t)n * (i + 1) / nthreads);
		pthread_create(&tid[i], 0, run, &jobs[i]);
	}
	for (i = 0; i < nthreads; ++i) pthread_join(tid[i], 0);
	return 0;
}
