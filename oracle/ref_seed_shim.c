/*
 * oracle/ref_seed_shim.c -- TEST INFRASTRUCTURE.  Ours, not the reference's: pthread loops around the REFERENCE's own
 * SMEM iterator (smem_itr_init / smem_set_query / smem_next2, reference bwamem.c:90-162) and bwt_sa (bwt.c:85), linked
 * against oracle/_ref/libbwa_ref.so, so that tools/fmindex_bench.py can time the reference's seeding queries on all
 * host cores (CPU baseline of kind "reference") without a Python call per read.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>

typedef struct { uint64_t x[3], info; } intv_t;            /* bwtintv_t, bwt.h:59-61 */
typedef struct { size_t n, m; intv_t *a; } intv_v;         /* bwtintv_v, bwt.h:63 */
extern void *smem_itr_init(const void *bwt);                /* bwamem.c:90 */
extern void smem_itr_destroy(void *itr);
extern void smem_set_query(void *itr, int len, const uint8_t *query);
extern const intv_v *smem_next2(void *itr, int split_len, int split_width, int start_width);
extern uint64_t bwt_sa(const void *bwt, uint64_t k);        /* bwt.c:85 */

typedef struct {
	const void *bwt;
	const uint8_t *pool;
	const uint64_t *off;
	const int *len;
	int lo, hi, split_len, split_width, start_width;
	uint64_t n_intv, checksum;
	const uint64_t *k;
	uint64_t *pos;
} job_t;

static void *run_smem(void *p)
{
	job_t *j = (job_t *)p;
	void *itr = smem_itr_init(j->bwt);
	int r;
	for (r = j->lo; r < j->hi; ++r) {
		const intv_v *a;
		const int sl = j->split_len < j->len[r] ? j->split_len : j->len[r]; /* bwamem.c:213 */
		smem_set_query(itr, j->len[r], j->pool + j->off[r]);
		while ((a = smem_next2(itr, sl, j->split_width, j->start_width)) != 0) {
			size_t i;
			j->n_intv += a->n;
			for (i = 0; i < a->n; ++i) j->checksum += a->a[i].x[0] * 31 + a->a[i].x[2] * 7 + a->a[i].info;
		}
	}
	smem_itr_destroy(itr);
	return 0;
}

static void *run_sa(void *p)
{
	job_t *j = (job_t *)p;
	int i;
	for (i = j->lo; i < j->hi; ++i) j->pos[i] = bwt_sa(j->bwt, j->k[i]);
	return 0;
}

static void spawn(job_t *jobs, int nthreads, int n, void *(*fn)(void *))
{
	pthread_t tid[256];
	int i;
	for (i = 0; i < nthreads; ++i) {
		jobs[i].lo = (int)((int64_t)n * i / nthreads), jobs[i].hi = (int)((int64_t)n * (i + 1) / nthreads);
		pthread_create(&tid[i], 0, fn, &jobs[i]);
	}
	for (i = 0; i < nthreads; ++i) pthread_join(tid[i], 0);
}

uint64_t ref_smem_iter_mt(const void *bwt, int n, const uint8_t *pool, const uint64_t *off, const int *len, int split_len,
                          int split_width, int start_width, int nthreads, uint64_t *checksum)
{
	job_t jobs[256];
	uint64_t tot = 0, cs = 0;
	int i;
	if (nthreads < 1) nthreads = 1;
	if (nthreads > 256) nthreads = 256;
	for (i = 0; i < nthreads; ++i) {
		jobs[i].bwt = bwt, jobs[i].pool = pool, jobs[i].off = off, jobs[i].len = len, jobs[i].split_len = split_len;
		jobs[i].split_width = split_width, jobs[i].start_width = start_width, jobs[i].n_intv = jobs[i].checksum = 0;
	}
	spawn(jobs, nthreads, n, run_smem);
	for (i = 0; i < nthreads; ++i) tot += jobs[i].n_intv, cs += jobs[i].checksum;
	if (checksum) *checksum = cs;
	return tot;
}

void ref_sa_mt(const void *bwt, const uint64_t *k, int n, uint64_t *pos, int nthreads)
{
	job_t jobs[256];
	int i;
	if (nthreads < 1) nthreads = 1;
	if (nthreads > 256) nthreads = 256;
	for (i = 0; i < nthreads; ++i) jobs[i].bwt = bwt, jobs[i].k = k, jobs[i].pos = pos;
	spawn(jobs, nthreads, n, run_sa);
}
