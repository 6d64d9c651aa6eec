/*
 * sam_post.c -- region post-processing and SAM text (host side, plain C, above the C-ABI).
 *
 * What mem_process_seqs does with a read's region vector after the extensions:
 *
 *   bmh_sort_and_dedup    mem_sort_and_dedup    reference bwa-0.7.8/bwamem.c:395-436
 *   bmh_mark_primary_se   mem_mark_primary_se   bwamem.c:445-475
 *   bmh_approx_mapq_se    mem_approx_mapq_se    bwamem.c:1023-1047
 *   bmh_pestat            mem_pestat            bwamem_pair.c:46-107   (cal_sub :34-44, mem_infer_dir :25-32)
 *   bmh_sam_batch         worker2 (bwamem.c:1281-1295) for a slice of a chunk: mem_reg2sam_se (:1049-1083) or
 *                         mem_sam_pe (bwamem_pair.c:240-332, without its rescue block) over mem_pair (:177-238),
 *                         mem_reg2aln (bwamem.c:1164-1236), bwa_fix_xref2 (bwa.c:179-222), mem_aln2sam (bwamem.c:904-1017)
 *
 * The reference interleaves decisions, global alignments and text per read.  Here a slice runs in three passes:
 *   A  per read / pair: primary marking, pairing, mapQ -> the exact list of regions that will be printed
 *   B  their global alignments as GPU batches: first the few regions that hang over the end of a reference sequence
 *      (bwa_fix_xref2 needs one bwa_gen_cigar2 each), then all of them through bmh_reg2cigar_batch (band inference and
 *      the <= 3 widening tries of mem_reg2aln)
 *   C  per read / pair: coordinates, clipping, flags, text
 * No per-region mallocs: alignments live in per-slice arrays, text grows in one buffer per read.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <time.h>

#include "../../include/bwamem_hip.h"
#include "sort_exact.h"

static double now_s(void) /* the clock of the BMH_DRIVER_TRACE lines: wall time, or with BMH_TRACE_CPU this thread's CPU time */
{
	static int cpu = -1;
	struct timespec ts;
	if (cpu < 0) cpu = getenv("BMH_TRACE_CPU") != 0;
	clock_gettime(cpu ? CLOCK_THREAD_CPUTIME_ID : CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

#define MIN_RATIO 0.8 /* bwamem_pair.c:14-18 */
#define MIN_DIR_CNT 10
#define MIN_DIR_RATIO 0.05
#define OUTLIER_BOUND 2.0
#define MAPPING_BOUND 3.0
#define MAX_STDDEV 4.0
#define MEM_MAPQ_COEF 30.0 /* bwamem.h:11 */

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

static inline uint64_t hash_64(uint64_t key) /* utils.h:98-109 */
{
	key += ~(key << 32);
	key ^= (key >> 22);
	key += ~(key << 13);
	key ^= (key >> 8);
	key += (key << 3);
	key ^= (key >> 15);
	key += ~(key << 27);
	key ^= (key >> 31);
	return key;
}

/* ---- orders (bwamem.c:386-393, utils.c:45) */
static int lt_re(const void *x, const void *y) { return ((const bmh_alnreg_t *)x)->re < ((const bmh_alnreg_t *)y)->re; }
static int lt_score_pos(const void *x, const void *y)
{
	const bmh_alnreg_t *a = (const bmh_alnreg_t *)x, *b = (const bmh_alnreg_t *)y;
	return a->score > b->score || (a->score == b->score && (a->rb < b->rb || (a->rb == b->rb && a->qb < b->qb)));
}
static int lt_score_hash(const void *x, const void *y)
{
	const bmh_alnreg_t *a = (const bmh_alnreg_t *)x, *b = (const bmh_alnreg_t *)y;
	return a->score > b->score || (a->score == b->score && a->hash < b->hash);
}
static int lt_u64(const void *x, const void *y) { return *(const uint64_t *)x < *(const uint64_t *)y; }
static void sort_u64(uint64_t *a, size_t n, uint64_t max) /* insert sizes: 1..max_ins -> a counting sort where that is small */
{
	if (max < (1u << 22) && n > 64) {
		uint32_t *cnt = (uint32_t *)calloc((size_t)max + 2, sizeof(uint32_t));
		size_t i, k = 0;
		uint64_t v;
		if (cnt) {
			for (i = 0; i < n; ++i) ++cnt[a[i] <= max ? a[i] : max + 1];
			for (v = 0; v <= max + 1; ++v)
				for (; cnt[v]; --cnt[v]) a[k++] = v;
			free(cnt);
			return;
		}
	}
	bmh_sort_exact(a, n, 8, lt_u64);
}
typedef struct { uint64_t x, y; } pair64_t;
static int lt_pair64(const void *p, const void *q)
{
	const pair64_t *a = (const pair64_t *)p, *b = (const pair64_t *)q;
	return a->x < b->x || (a->x == b->x && a->y < b->y);
}

/* ---- bwamem.c:395-436 */
int bmh_sort_and_dedup(int n, bmh_alnreg_t *a, float mask_level_redun)
{
	int m, i, j;
	if (n <= 1) return n;
	bmh_sort_exact(a, (size_t)n, sizeof(*a), lt_re);
	for (i = 1; i < n; ++i) {
		bmh_alnreg_t *p = &a[i];
		if (p->rb >= a[i - 1].re) continue;
		for (j = i - 1; j >= 0 && p->rb < a[j].re; --j) {
			bmh_alnreg_t *q = &a[j];
			int64_t orr, oq, mr, mq;
			if (q->qe == q->qb) continue; /* already excluded */
			orr = q->re - p->rb;                                  /* overlap on the reference */
			oq = q->qb < p->qb ? q->qe - p->qb : p->qe - q->qb;   /* overlap on the query */
			mr = q->re - q->rb < p->re - p->rb ? q->re - q->rb : p->re - p->rb;
			mq = q->qe - q->qb < p->qe - p->qb ? q->qe - q->qb : p->qe - p->qb;
			if (orr > mask_level_redun * mr && oq > mask_level_redun * mq) { /* one of the two is redundant */
				if (p->score < q->score) {
					p->qe = p->qb;
					break;
				} else q->qe = q->qb;
			}
		}
	}
	for (i = 0, m = 0; i < n; ++i)
		if (a[i].qe > a[i].qb) {
			if (m != i) a[m++] = a[i];
			else ++m;
		}
	n = m;
	bmh_sort_exact(a, (size_t)n, sizeof(*a), lt_score_pos);
	for (i = 1; i < n; ++i) /* identical hits */
		if (a[i].score == a[i - 1].score && a[i].rb == a[i - 1].rb && a[i].qb == a[i - 1].qb) a[i].qe = a[i].qb;
	for (i = 1, m = 1; i < n; ++i)
		if (a[i].qe > a[i].qb) {
			if (m != i) a[m++] = a[i];
			else ++m;
		}
	return m;
}

static inline int gap_tmp(const bmh_sam_opt_t *o) /* the largest single-event penalty, bwamem.c:455-457 */
{
	int tmp = o->a + o->b;
	tmp = o->o_del + o->e_del > tmp ? o->o_del + o->e_del : tmp;
	return o->o_ins + o->e_ins > tmp ? o->o_ins + o->e_ins : tmp;
}

/* ---- bwamem.c:445-475 */
void bmh_mark_primary_se(const bmh_sam_opt_t *o, int n, bmh_alnreg_t *a, int64_t id)
{
	int i, k, nz = 0, tmp, zs[64], *z = zs, zcap = 64;
	if (n == 0) return;
	for (i = 0; i < n; ++i) a[i].sub = 0, a[i].secondary = -1, a[i].hash = hash_64((uint64_t)(id + i));
	bmh_sort_exact(a, (size_t)n, sizeof(*a), lt_score_hash);
	tmp = gap_tmp(o);
	z[nz++] = 0;
	for (i = 1; i < n; ++i) {
		for (k = 0; k < nz; ++k) {
			const int j = z[k];
			const int b_max = imax(a[j].qb, a[i].qb), e_min = imin(a[j].qe, a[i].qe);
			if (e_min > b_max) { /* overlap on the query */
				const int min_l = imin(a[i].qe - a[i].qb, a[j].qe - a[j].qb);
				if (e_min - b_max >= min_l * o->mask_level) { /* significant */
					if (a[j].sub == 0) a[j].sub = a[i].score;
					if (a[j].score - a[i].score <= tmp) ++a[j].sub_n;
					break;
				}
			}
		}
		if (k == nz) {
			if (nz == zcap) {
				int *z2 = (int *)malloc(sizeof(int) * (size_t)zcap * 2);
				memcpy(z2, z, sizeof(int) * (size_t)nz);
				if (z != zs) free(z);
				z = z2, zcap *= 2;
			}
			z[nz++] = i;
		} else a[i].secondary = z[k];
	}
	if (z != zs) free(z);
}

/* ---- bwamem.c:1023-1047 */
int bmh_approx_mapq_se(const bmh_sam_opt_t *o, const bmh_alnreg_t *a)
{
	int mapq, l, sub = a->sub ? a->sub : o->min_seed_len * o->a;
	double identity;
	sub = a->csub > sub ? a->csub : sub;
	if (sub >= a->score) return 0;
	l = a->qe - a->qb > a->re - a->rb ? a->qe - a->qb : (int)(a->re - a->rb);
	identity = 1. - (double)(l * o->a - a->score) / (o->a + o->b) / l;
	if (a->score == 0) mapq = 0;
	else if (o->mapQ_coef_len > 0) {
		double tmp;
		tmp = l < o->mapQ_coef_len ? 1. : o->mapQ_coef_fac / log(l);
		tmp *= identity * identity;
		mapq = (int)(6.02 * (a->score - sub) / o->a * tmp * tmp + .499);
	} else {
		mapq = (int)(MEM_MAPQ_COEF * (1. - (double)sub / a->score) * log(a->seedcov) + .499);
		mapq = identity < 0.95 ? (int)(mapq * identity * identity + .499) : mapq;
	}
	if (a->sub_n > 0) mapq -= (int)(4.343 * log(a->sub_n + 1) + .499);
	if (mapq > 60) mapq = 60;
	if (mapq < 0) mapq = 0;
	return mapq;
}

/* ---- bwamem_pair.c:25-32 */
static inline int infer_dir(int64_t l_pac, int64_t b1, int64_t b2, int64_t *dist)
{
	const int r1 = b1 >= l_pac, r2 = b2 >= l_pac;
	const int64_t p2 = r1 == r2 ? b2 : (l_pac << 1) - 1 - b2; /* read 2 on the strand of read 1 */
	*dist = p2 > b1 ? p2 - b1 : b1 - p2;
	return (r1 == r2 ? 0 : 1) ^ (p2 > b1 ? 0 : 3);
}

/* ---- bwamem_pair.c:34-44 */
static int cal_sub(const bmh_sam_opt_t *o, const bmh_alnreg_v *r)
{
	size_t j;
	for (j = 1; j < r->n; ++j) {
		const int b_max = imax(r->a[j].qb, r->a[0].qb), e_min = imin(r->a[j].qe, r->a[0].qe);
		if (e_min > b_max) {
			const int min_l = imin(r->a[j].qe - r->a[j].qb, r->a[0].qe - r->a[0].qb);
			if (e_min - b_max >= min_l * o->mask_level) break;
		}
	}
	return j < r->n ? r->a[j].score : o->min_seed_len * o->a;
}

/* ---- bwamem_pair.c:46-107 */
void bmh_pestat(const bmh_sam_opt_t *o, int64_t l_pac, int n, const bmh_alnreg_v *regs, bmh_pestat_t pes[4], int verbose)
{
	struct { size_t n, m; uint64_t *a; } isize[4];
	size_t max;
	int i, d;
	memset(pes, 0, 4 * sizeof(bmh_pestat_t));
	memset(isize, 0, sizeof(isize));
	for (i = 0; i < n >> 1; ++i) {
		const bmh_alnreg_v *r0 = &regs[i << 1 | 0], *r1 = &regs[i << 1 | 1];
		int64_t is;
		int dir;
		if (i + 8 < n >> 1) /* every read's region vector is an allocation of its own: this serial loop is all cache misses */
			__builtin_prefetch(regs[(i + 8) << 1].a), __builtin_prefetch(regs[(i + 8) << 1 | 1].a);
		if (r0->n == 0 || r1->n == 0) continue;
		if (cal_sub(o, r0) > MIN_RATIO * r0->a[0].score) continue;
		if (cal_sub(o, r1) > MIN_RATIO * r1->a[0].score) continue;
		dir = infer_dir(l_pac, r0->a[0].rb, r1->a[0].rb, &is);
		if (is && is <= o->max_ins) {
			if (isize[dir].n == isize[dir].m) {
				isize[dir].m = isize[dir].m ? isize[dir].m << 1 : 2;
				isize[dir].a = (uint64_t *)realloc(isize[dir].a, 8 * isize[dir].m);
			}
			isize[dir].a[isize[dir].n++] = (uint64_t)is;
		}
	}
	if (verbose >= 3)
		fprintf(stderr, "[M::mem_pestat] # candidate unique pairs for (FF, FR, RF, RR): (%ld, %ld, %ld, %ld)\n", (long)isize[0].n, (long)isize[1].n,
		        (long)isize[2].n, (long)isize[3].n);
	for (d = 0; d < 4; ++d) {
		bmh_pestat_t *r = &pes[d];
		uint64_t *q = isize[d].a;
		const size_t qn = isize[d].n;
		size_t k;
		int p25, p50, p75, x;
		if (qn < MIN_DIR_CNT) {
			if (verbose >= 0) fprintf(stderr, "[M::mem_pestat] skip orientation %c%c as there are not enough pairs\n", "FR"[d >> 1 & 1], "FR"[d & 1]);
			r->failed = 1;
			continue;
		} else if (verbose >= 0) fprintf(stderr, "[M::mem_pestat] analyzing insert size distribution for orientation %c%c...\n", "FR"[d >> 1 & 1], "FR"[d & 1]);
		sort_u64(q, qn, (uint64_t)o->max_ins); /* ks_introsort_64 (:75): equal keys are indistinguishable, any sort gives its array */
		p25 = (int)q[(int)(.25 * qn + .499)];
		p50 = (int)q[(int)(.50 * qn + .499)];
		p75 = (int)q[(int)(.75 * qn + .499)];
		r->low = (int)(p25 - OUTLIER_BOUND * (p75 - p25) + .499);
		if (r->low < 1) r->low = 1;
		r->high = (int)(p75 + OUTLIER_BOUND * (p75 - p25) + .499);
		if (verbose >= 0) {
			fprintf(stderr, "[M::mem_pestat] (25, 50, 75) percentile: (%d, %d, %d)\n", p25, p50, p75);
			fprintf(stderr, "[M::mem_pestat] low and high boundaries for computing mean and std.dev: (%d, %d)\n", r->low, r->high);
		}
		for (k = 0, x = 0, r->avg = 0; k < qn; ++k)
			if (q[k] >= (uint64_t)r->low && q[k] <= (uint64_t)r->high) r->avg += q[k], ++x;
		r->avg /= x;
		for (k = 0, r->std = 0; k < qn; ++k)
			if (q[k] >= (uint64_t)r->low && q[k] <= (uint64_t)r->high) r->std += (q[k] - r->avg) * (q[k] - r->avg);
		r->std = sqrt(r->std / x);
		if (verbose >= 0) fprintf(stderr, "[M::mem_pestat] mean and std.dev: (%.2f, %.2f)\n", r->avg, r->std);
		r->low = (int)(p25 - MAPPING_BOUND * (p75 - p25) + .499);
		r->high = (int)(p75 + MAPPING_BOUND * (p75 - p25) + .499);
		if (r->low > r->avg - MAX_STDDEV * r->std) r->low = (int)(r->avg - MAX_STDDEV * r->std + .499);
		if (r->high < r->avg - MAX_STDDEV * r->std) r->high = (int)(r->avg + MAX_STDDEV * r->std + .499);
		if (r->low < 1) r->low = 1;
		if (verbose >= 0) fprintf(stderr, "[M::mem_pestat] low and high boundaries for proper pairs: (%d, %d)\n", r->low, r->high);
	}
	for (d = 0, max = 0; d < 4; ++d) max = max > isize[d].n ? max : isize[d].n;
	for (d = 0; d < 4; ++d) {
		if (pes[d].failed == 0 && isize[d].n < max * MIN_DIR_RATIO) {
			pes[d].failed = 1;
			if (verbose >= 0) fprintf(stderr, "[M::mem_pestat] skip orientation %c%c\n", "FR"[d >> 1 & 1], "FR"[d & 1]);
		}
		free(isize[d].a);
	}
}

/* ---- bwamem_pair.c:177-238.  Scratch vectors are the caller's (reused across pairs). */
typedef struct { size_t n, m; pair64_t *a; } pair64_v;
static pair64_t *pv_push(pair64_v *v)
{
	if (v->n == v->m) {
		v->m = v->m ? v->m << 1 : 16;
		v->a = (pair64_t *)realloc(v->a, sizeof(pair64_t) * v->m);
	}
	return &v->a[v->n++];
}

static int pair_ends(const bmh_sam_opt_t *o, int64_t l_pac, const bmh_pestat_t pes[4], const bmh_alnreg_v a[2], uint64_t id, int *sub,
                     int *n_sub, int z[2], pair64_v *v, pair64_v *u)
{
	int r, y[4], ret;
	size_t i;
	v->n = u->n = 0;
	for (r = 0; r < 2; ++r)
		for (i = 0; i < a[r].n; ++i) {
			const bmh_alnreg_t *e = &a[r].a[i];
			pair64_t *key = pv_push(v);
			key->x = (uint64_t)(e->rb < l_pac ? e->rb : (l_pac << 1) - 1 - e->rb); /* forward position */
			key->y = (uint64_t)e->score << 32 | (uint64_t)(i << 2) | (uint64_t)((e->rb >= l_pac) << 1) | (uint64_t)r;
		}
	bmh_sort_exact(v->a, v->n, sizeof(pair64_t), lt_pair64);
	y[0] = y[1] = y[2] = y[3] = -1;
	for (i = 0; i < v->n; ++i) {
		for (r = 0; r < 2; ++r) { /* direction */
			const int dir = r << 1 | (int)(v->a[i].y >> 1 & 1);
			int which, k;
			if (pes[dir].failed) continue;
			which = r << 1 | (int)((v->a[i].y & 1) ^ 1);
			if (y[which] < 0) continue; /* no earlier hit of that kind */
			for (k = y[which]; k >= 0; --k) {
				int64_t dist;
				int q;
				double ns;
				pair64_t *p;
				if ((int)(v->a[k].y & 3) != which) continue;
				dist = (int64_t)v->a[i].x - (int64_t)v->a[k].x;
				if (dist > pes[dir].high) break;
				if (dist < pes[dir].low) continue;
				ns = (dist - pes[dir].avg) / pes[dir].std;
				q = (int)((v->a[i].y >> 32) + (v->a[k].y >> 32) + .721 * log(2. * erfc(fabs(ns) * M_SQRT1_2)) * o->a + .499); /* .721 = 1/log(4) */
				if (q < 0) q = 0;
				p = pv_push(u);
				p->y = (uint64_t)k << 32 | i;
				/* the reference's mem_pair takes the pair id as an `int` (bwamem_pair.c:177) and shifts it as one */
				p->x = (uint64_t)q << 32 | (hash_64(p->y ^ (uint64_t)(int64_t)(int32_t)((uint32_t)(int32_t)id << 8)) & 0xffffffffU);
			}
		}
		y[v->a[i].y & 3] = (int)i;
	}
	if (u->n) { /* at least one proper pair */
		const int tmp = gap_tmp(o);
		long k2;
		size_t bi, bk;
		bmh_sort_exact(u->a, u->n, sizeof(pair64_t), lt_pair64);
		bi = (size_t)(u->a[u->n - 1].y >> 32), bk = (size_t)(u->a[u->n - 1].y << 32 >> 32);
		z[v->a[bi].y & 1] = (int)(v->a[bi].y << 32 >> 34); /* index of the best pair */
		z[v->a[bk].y & 1] = (int)(v->a[bk].y << 32 >> 34);
		ret = (int)(u->a[u->n - 1].x >> 32);
		*sub = u->n > 1 ? (int)(u->a[u->n - 2].x >> 32) : 0;
		for (k2 = (long)u->n - 2, *n_sub = 0; k2 >= 0; --k2)
			if (*sub - (int)(u->a[k2].x >> 32) <= tmp) ++*n_sub;
	} else ret = 0, *sub = 0, *n_sub = 0;
	return ret;
}

int bmh_pair(const bmh_sam_opt_t *o, int64_t l_pac, const bmh_pestat_t pes[4], const bmh_alnreg_v a[2], uint64_t id, int *sub, int *n_sub, int z[2])
{
	pair64_v v = {0, 0, 0}, u = {0, 0, 0};
	const int ret = pair_ends(o, l_pac, pes, a, id, sub, n_sub, z, &v, &u);
	free(v.a), free(u.a);
	return ret;
}

#define raw_mapq(diff, a) ((int)(6.02 * (diff) / (a) + .499)) /* bwamem_pair.c:238 */

/* ================================================================================================ alignments and text */

typedef struct { /* mem_aln_t (bwamem.h:72-82) with the CIGAR in a slice arena and the MD string kept apart */
	int64_t pos;
	int rid, flag, is_rev, mapq, NM, n_cigar, score, sub;
	uint32_t cig_off; /* first word in the slice's CIGAR arena */
	const char *md;
} aln_t;

typedef struct { /* text under construction: kstring_t without the per-call growth checks */
	char *s;
	size_t l, m;
} str_t;

static inline void st_room(str_t *t, size_t extra)
{
	if (t->l + extra + 1 > t->m) {
		t->m = (t->l + extra + 1) * 2;
		t->s = (char *)realloc(t->s, t->m);
	}
}
static inline void st_c(str_t *t, char c) { st_room(t, 1), t->s[t->l++] = c; }
static inline void st_n(str_t *t, const char *p, size_t n) { st_room(t, n), memcpy(t->s + t->l, p, n), t->l += n; }
static inline void st_s(str_t *t, const char *p) { st_n(t, p, strlen(p)); }
static void st_l(str_t *t, long c) /* kputw / kputl, kstring.h:62-111 */
{
	char buf[32];
	int l = 0;
	unsigned long x = c < 0 ? 0ul - (unsigned long)c : (unsigned long)c;
	if (c == 0) { st_c(t, '0'); return; }
	for (; x > 0; x /= 10) buf[l++] = (char)('0' + x % 10);
	if (c < 0) buf[l++] = '-';
	st_room(t, (size_t)l);
	while (l > 0) t->s[t->l++] = buf[--l];
}

static inline int rlen_of(int n_cigar, const uint32_t *cigar) /* get_rlen, bwamem.c:893-902 */
{
	int k, l;
	for (k = l = 0; k < n_cigar; ++k) {
		const int op = (int)(cigar[k] & 0xf);
		if (op == 0 || op == 2) l += (int)(cigar[k] >> 4);
	}
	return l;
}

/* ---- One SAM line (what mem_aln2sam prints, bwamem.c:904-1017), built the way this library is: the record and its mate are first
 * RESOLVED into a small value (flag bits, where an unmapped end is placed, clip lengths), then a fixed table of column emitters
 * writes the eleven mandatory columns and the tags from that value.  `cig` is the slice's CIGAR arena. */
typedef struct {
	int mapped;           /* has a reference sequence to print (its own, or the mate's for an unmapped end) */
	int rid, rev, n_op;   /* n_op = 0 when the end only borrows its mate's position */
	int64_t pos;
	const uint32_t *op;
} place_t;

typedef struct {
	const bmh_refidx_t *bns;
	const bmh_seq_t *read;
	const aln_t *all;     /* the read's records (for SA:Z) */
	int n_all, self;      /* ... and which one this line is */
	const aln_t *rec;
	const uint32_t *cig;
	const char *rg_id;
	int flag, has_mate;
	place_t me, mate;
} samline_t;

static place_t place_of(const aln_t *a, const uint32_t *cig)
{
	place_t p;
	p.mapped = a->rid >= 0, p.rid = a->rid, p.rev = a->is_rev, p.n_op = a->n_cigar, p.pos = a->pos, p.op = cig + a->cig_off;
	return p;
}
/* an unmapped end sits where its mapped mate is, without a CIGAR (bwamem.c:917-918) */
static void borrow_place(place_t *dst, const place_t *src) { dst->mapped = 1, dst->rid = src->rid, dst->pos = src->pos, dst->rev = src->rev, dst->n_op = 0; }

static int clip_len(const place_t *p, int at) /* length of a clip operation at CIGAR index `at`, else 0 */
{
	const int op = (int)(p->op[at] & 0xf);
	return op == 3 || op == 4 ? (int)(p->op[at] >> 4) : 0;
}
static int64_t far_end(const place_t *p) { return p->pos + (p->rev ? rlen_of(p->n_op, p->op) - 1 : 0); } /* 5' end on the reference */

/* bases or qualities of [from,to) in the orientation of the alignment; `code` maps a base code to its letter (NULL: copy bytes) */
static void put_oriented(str_t *t, const char *src, int from, int to, int rev, const char *code)
{
	int i, n = to > from ? to - from : 0;
	char *d;
	st_room(t, (size_t)n + 2);
	d = t->s + t->l;
	if (!rev) for (i = 0; i < n; ++i) d[i] = code ? code[(int)src[from + i]] : src[from + i];
	else for (i = 0; i < n; ++i) d[i] = code ? code[(int)src[to - 1 - i]] : src[to - 1 - i];
	t->l += (size_t)n;
}
enum { CLIP_SOFT, CLIP_HARD, CLIP_ASIS }; /* clips of the first line are S, of a supplementary line H; SA:Z prints what is stored */
static void put_ops(str_t *t, const uint32_t *op, int n, int clips)
{
	int i;
	for (i = 0; i < n; ++i) {
		int c = (int)(op[i] & 0xf);
		if (clips != CLIP_ASIS && (c == 3 || c == 4)) c = clips == CLIP_HARD ? 4 : 3;
		st_l(t, (long)(op[i] >> 4)), st_c(t, "MIDSH"[c]);
	}
}

static void col_qname_flag(const samline_t *L, str_t *t)
{
	st_s(t, L->read->name), st_c(t, '\t');
	st_l(t, (L->flag & 0xffff) | (L->flag & 0x10000 ? 0x100 : 0)); /* -M: a supplementary hit is shown as secondary (bwamem.c:928) */
}
static void col_rname_pos_mapq_cigar(const samline_t *L, str_t *t)
{
	if (!L->me.mapped) { st_n(t, "*\t0\t0\t*", 7); return; }
	st_s(t, L->bns->anns[L->me.rid].name), st_c(t, '\t');
	st_l(t, (long)(L->me.pos + 1)), st_c(t, '\t');
	st_l(t, L->rec->mapq), st_c(t, '\t');
	if (L->me.n_op) put_ops(t, L->me.op, L->me.n_op, L->self ? CLIP_HARD : CLIP_SOFT);
	else st_c(t, '*');
}
static void col_mate(const samline_t *L, str_t *t)
{
	if (!L->has_mate || !L->mate.mapped) { st_n(t, "*\t0\t0", 5); return; }
	if (L->me.rid == L->mate.rid) st_c(t, '=');
	else st_s(t, L->bns->anns[L->mate.rid].name);
	st_c(t, '\t'), st_l(t, (long)(L->mate.pos + 1)), st_c(t, '\t');
	if (L->me.rid == L->mate.rid && L->me.n_op && L->mate.n_op) { /* TLEN between the two 5' ends (bwamem.c:957-961) */
		const int64_t d = far_end(&L->me) - far_end(&L->mate);
		st_l(t, (long)-(d + (d > 0) - (d < 0)));
	} else st_c(t, '0');
}
static void col_seq_qual(const samline_t *L, str_t *t)
{
	int from = 0, to = L->read->l_seq;
	if (L->flag & 0x100) { st_n(t, "*\t*", 3); return; } /* none on secondary lines */
	if (L->self && L->me.n_op) { /* a supplementary line prints only what it aligns: its clips are hard (bwamem.c:971-976) */
		const int c5 = clip_len(&L->me, 0), c3 = clip_len(&L->me, L->me.n_op - 1);
		if (L->me.rev) from += c3, to -= c5;
		else from += c5, to -= c3;
	}
	put_oriented(t, L->read->seq, from, to, L->me.rev, L->me.rev ? "TGCAN" : "ACGTN");
	st_c(t, '\t');
	if (L->read->qual) put_oriented(t, L->read->qual, from, to, L->me.rev, 0);
	else st_c(t, '*');
}
static void col_tags(const samline_t *L, str_t *t)
{
	const aln_t *r = L->rec;
	int i, others = 0;
	if (L->me.n_op) st_n(t, "\tNM:i:", 6), st_l(t, r->NM), st_n(t, "\tMD:Z:", 6), st_s(t, r->md);
	if (r->score >= 0) st_n(t, "\tAS:i:", 6), st_l(t, r->score);
	if (r->sub >= 0) st_n(t, "\tXS:i:", 6), st_l(t, r->sub);
	if (L->rg_id && L->rg_id[0]) st_n(t, "\tRG:Z:", 6), st_s(t, L->rg_id);
	if (!(L->flag & 0x100)) { /* SA:Z lists the read's other non-secondary lines (bwamem.c:995-1013) */
		for (i = 0; i < L->n_all; ++i) others += i != L->self && !(L->all[i].flag & 0x100);
		if (others) st_n(t, "\tSA:Z:", 6);
		for (i = 0; others && i < L->n_all; ++i) {
			const aln_t *o = &L->all[i];
			if (i == L->self || (o->flag & 0x100) || o->rid < 0) continue;
			st_s(t, L->bns->anns[o->rid].name), st_c(t, ','), st_l(t, (long)(o->pos + 1)), st_c(t, ',');
			st_c(t, o->is_rev ? '-' : '+'), st_c(t, ',');
			put_ops(t, L->cig + o->cig_off, o->n_cigar, CLIP_ASIS);
			st_c(t, ','), st_l(t, o->mapq), st_c(t, ','), st_l(t, o->NM), st_c(t, ';');
		}
	}
	if (L->read->comment) st_c(t, '\t'), st_s(t, L->read->comment);
}

typedef void (*sam_col_fn)(const samline_t *, str_t *);
static const sam_col_fn k_sam_cols[] = {col_qname_flag, col_rname_pos_mapq_cigar, col_mate, col_seq_qual};

static void aln2sam(const bmh_refidx_t *bns, str_t *str, const bmh_seq_t *s, int n, const aln_t *list, int which, const aln_t *mate,
                    const uint32_t *cig, const char *rg_id)
{
	samline_t L;
	size_t c;
	L.bns = bns, L.read = s, L.all = list, L.n_all = n, L.self = which, L.rec = &list[which], L.cig = cig, L.rg_id = rg_id;
	L.has_mate = mate != 0;
	L.me = place_of(L.rec, cig);
	L.flag = L.rec->flag | (mate ? 0x1 : 0) | (L.me.mapped ? 0 : 0x4);
	if (mate) {
		L.mate = place_of(mate, cig);
		if (!L.mate.mapped) L.flag |= 0x8;
		if (!L.me.mapped && L.mate.mapped) borrow_place(&L.me, &L.mate);
		else if (!L.mate.mapped && L.me.mapped) borrow_place(&L.mate, &L.me);
		if (L.mate.rev) L.flag |= 0x20;
	}
	if (L.me.rev) L.flag |= 0x10;
	for (c = 0; c < sizeof(k_sam_cols) / sizeof(k_sam_cols[0]); ++c) k_sam_cols[c](&L, str), st_c(str, '\t');
	--str->l; /* the tags bring their own separators */
	col_tags(&L, str);
	st_c(str, '\n');
}

/* ---- bntseq.h:83-86, bntseq.c:316-330 */
static inline int64_t depos(int64_t l_pac, int64_t pos, int *is_rev) { return (*is_rev = pos >= l_pac) ? (l_pac << 1) - 1 - pos : pos; }
static int pos2rid(const bmh_refidx_t *bns, int64_t pos_f)
{
	int left = 0, mid = 0, right = bns->n_seqs;
	if (pos_f >= bns->l_pac) return -1;
	while (left < right) {
		mid = (left + right) >> 1;
		if (pos_f >= bns->anns[mid].offset) {
			if (mid == bns->n_seqs - 1) break;
			if (pos_f < bns->anns[mid + 1].offset) break;
			left = mid + 1;
		} else right = mid;
	}
	return mid;
}

typedef struct { /* one region that gets an alignment */
	int read, k;      /* read of the slice, region index in its (sorted) vector */
	int qb, qe;       /* after bwa_fix_xref2 */
	int64_t rb, re;
	int fix;          /* hangs over the end of its reference sequence: index into the fix batch, else -1 */
	int64_t cb, ce;   /* ... and the interval it has to be cut to */
} want_t;

typedef struct { /* pass A's verdict on a pair */
	int paired, z[2], q_se[2], extra_flag;
} pairdec_t;

static void unmapped(aln_t *a) /* mem_reg2aln(..., 0), bwamem.c:1171-1175 */
{
	memset(a, 0, sizeof(*a));
	a->rid = -1, a->pos = -1, a->flag |= 0x4;
}

typedef struct {
	want_t *a;
	size_t n, m;
} want_v;
static int want_push(want_v *w, int read, int k, const bmh_alnreg_t *ar)
{
	want_t *x;
	if (w->n == w->m) {
		w->m = w->m ? w->m << 1 : 1024;
		w->a = (want_t *)realloc(w->a, sizeof(want_t) * w->m);
		if (!w->a) return BMH_E_NOMEM;
	}
	x = &w->a[w->n++];
	x->read = read, x->k = k, x->qb = ar->qb, x->qe = ar->qe, x->rb = ar->rb, x->re = ar->re, x->fix = -1;
	return 0;
}

/* the regions mem_reg2sam_se prints, bwamem.c:1057-1062 (k = 0 first: it is also the `h` of mem_sam_pe's no_pairing) */
static int want_se(const bmh_sam_opt_t *o, want_v *w, int read, const bmh_alnreg_v *a)
{
	size_t k;
	int rc;
	for (k = 0; k < a->n; ++k) {
		const bmh_alnreg_t *p = &a->a[k];
		if (p->score < o->T) continue;
		if (p->secondary >= 0 && !(o->flag & BMH_MEM_F_ALL)) continue;
		if (p->secondary >= 0 && p->score < a->a[p->secondary].score * .5) continue;
		if (p->rb < 0 || p->re < 0) continue; /* mem_reg2aln then writes an unmapped record, bwamem.c:1172 */
		if ((rc = want_push(w, read, (int)k, p))) return rc;
	}
	return 0;
}

int bmh_sam_batch(bmh_ctx_t *ctx, const bmh_sam_opt_t *o, const bmh_refidx_t *bns, const uint8_t *pac, const bmh_pestat_t *pes,
                  int64_t id0, int n, bmh_seq_t *seqs, bmh_alnreg_v *regs, const char *rg_id)
{
	const int pe = (o->flag & BMH_MEM_F_PE) != 0;
	const int64_t l_pac = bns ? bns->l_pac : 0;
	want_v W = {0, 0, 0};
	pairdec_t *pd = 0;
	bmh_read_t *reads = 0;
	bmh_cigar_req_t *reqs = 0;
	bmh_cigar_res_t *res = 0;
	uint32_t *cig = 0, *arena = 0;
	char *md = 0;
	size_t *first = 0; /* first entry of W per read (+1 sentinel) */
	aln_t *alns = 0;
	pair64_v pv = {0, 0, 0}, pu = {0, 0, 0};
	str_t str = {0, 0, 0};
	size_t j, cw = 8, mb = 16, n_fix = 0, arena_words = 0;
	int i, rc = BMH_OK;
	const int trace = getenv("BMH_DRIVER_TRACE") != 0;
	double tt[4] = {0, 0, 0, 0};

	if (!ctx || !o || !bns || !pac || n < 0 || (n > 0 && (!seqs || !regs)) || (pe && ((n & 1) || !pes))) return BMH_E_ARG;
	if (n == 0) return BMH_OK;
	first = (size_t *)calloc((size_t)n + 1, sizeof(size_t));
	reads = (bmh_read_t *)malloc(sizeof(bmh_read_t) * (size_t)n);
	if (pe) pd = (pairdec_t *)calloc((size_t)(n >> 1), sizeof(pairdec_t));
	if (!first || !reads || (pe && !pd)) { rc = BMH_E_NOMEM; goto done; }
	for (i = 0; i < n; ++i) reads[i].l_seq = seqs[i].l_seq, reads[i].seq = (const uint8_t *)seqs[i].seq;

	if (trace) tt[0] = now_s();
	/* ---- pass A: decisions */
	if (!pe) {
		for (i = 0; i < n; ++i) { /* worker2's SE branch, bwamem.c:1285-1289 */
			if (i + 8 < n) __builtin_prefetch(regs[i + 8].a);
			bmh_mark_primary_se(o, (int)regs[i].n, regs[i].a, id0 + i);
			first[i] = W.n;
			if ((rc = want_se(o, &W, i, &regs[i]))) goto done;
		}
	} else {
		for (i = 0; i < n >> 1; ++i) { /* mem_sam_pe after its rescue block, bwamem_pair.c:264-331 */
			bmh_alnreg_v *a = &regs[i << 1];
			if (i + 6 < n >> 1) __builtin_prefetch(regs[(i + 6) << 1].a), __builtin_prefetch(regs[(i + 6) << 1 | 1].a);
			const uint64_t id = (uint64_t)(id0 >> 1) + (uint64_t)i;
			pairdec_t *d = &pd[i];
			int sub_o = 0, n_sub = 0, oo, r, go_pair = 0;
			bmh_mark_primary_se(o, (int)a[0].n, a[0].a, (int64_t)(id << 1 | 0));
			bmh_mark_primary_se(o, (int)a[1].n, a[1].a, (int64_t)(id << 1 | 1));
			d->extra_flag = 1;
			if (!(o->flag & BMH_MEM_F_NOPAIRING) && a[0].n && a[1].n &&
			    (oo = pair_ends(o, l_pac, pes, a, id, &sub_o, &n_sub, d->z, &pv, &pu)) > 0) {
				int is_multi[2], q_pe, score_un;
				size_t jj;
				for (r = 0; r < 2; ++r) { /* more than one good hit at an end even after rescue? */
					for (jj = 1; jj < a[r].n; ++jj)
						if (a[r].a[jj].secondary < 0 && a[r].a[jj].score >= o->T) break;
					is_multi[r] = jj < a[r].n;
				}
				if (!is_multi[0] && !is_multi[1]) {
					go_pair = 1;
					score_un = a[0].a[0].score + a[1].a[0].score - o->pen_unpaired;
					sub_o = sub_o > score_un ? sub_o : score_un;
					q_pe = raw_mapq(oo - sub_o, o->a);
					if (n_sub > 0) q_pe -= (int)(4.343 * log(n_sub + 1) + .499);
					if (q_pe < 0) q_pe = 0;
					if (q_pe > 60) q_pe = 60;
					if (oo > score_un) { /* the pair wins */
						bmh_alnreg_t *c[2];
						c[0] = &a[0].a[d->z[0]], c[1] = &a[1].a[d->z[1]];
						for (r = 0; r < 2; ++r) {
							if (c[r]->secondary >= 0) c[r]->sub = a[r].a[c[r]->secondary].score, c[r]->secondary = -2;
							d->q_se[r] = bmh_approx_mapq_se(o, c[r]);
						}
						for (r = 0; r < 2; ++r) d->q_se[r] = d->q_se[r] > q_pe ? d->q_se[r] : q_pe < d->q_se[r] + 40 ? q_pe : d->q_se[r] + 40;
						d->extra_flag |= 2;
						for (r = 0; r < 2; ++r) { /* cap at the tandem-repeat score */
							const int cap = raw_mapq(c[r]->score - c[r]->csub, o->a);
							d->q_se[r] = d->q_se[r] < cap ? d->q_se[r] : cap;
						}
					} else { /* the two best single-end hits win */
						d->z[0] = d->z[1] = 0;
						d->q_se[0] = bmh_approx_mapq_se(o, &a[0].a[0]);
						d->q_se[1] = bmh_approx_mapq_se(o, &a[1].a[0]);
					}
				}
			}
			d->paired = go_pair;
			for (r = 0; r < 2; ++r) {
				const int rd = i << 1 | r;
				first[rd] = W.n;
				if (go_pair) {
					const bmh_alnreg_t *ar = &a[r].a[d->z[r]];
					if (ar->rb >= 0 && ar->re >= 0 && (rc = want_push(&W, rd, d->z[r], ar))) goto done;
				} else if ((rc = want_se(o, &W, rd, &a[r]))) goto done;
			}
		}
	}
	first[n] = W.n;

	if (trace) tt[1] = now_s();
	/* ---- pass B: bwa_fix_xref2 (bwa.c:179-222), then the alignments */
	for (j = 0; j < W.n; ++j) {
		want_t *x = &W.a[j];
		const bmh_refann_t *ra;
		int is_rev;
		int64_t fm, cb, ce;
		if (x->rb < l_pac && x->re > l_pac) { /* bridges the strands: the reference gives up on the run (bwamem.c:1183-1186) */
			rc = BMH_E_ARG;
			goto done;
		}
		fm = depos(l_pac, (x->rb + x->re) >> 1, &is_rev);
		ra = &bns->anns[pos2rid(bns, fm)];
		cb = is_rev ? (l_pac << 1) - (ra->offset + ra->len) : ra->offset; /* its sequence, on the mapping strand */
		ce = cb + ra->len;
		if (cb > x->rb || ce < x->re) x->fix = (int)n_fix++, x->cb = cb > x->rb ? cb : x->rb, x->ce = ce < x->re ? ce : x->re;
	}
	if (n_fix) { /* one bwa_gen_cigar2(w_ = opt->w) per such region, then walk its CIGAR to the cut points (bwa.c:198-219) */
		size_t f = 0, fw = 8, fm_ = 16;
		bmh_cigar_req_t *fq = (bmh_cigar_req_t *)malloc(sizeof(*fq) * n_fix);
		bmh_cigar_res_t *fr = (bmh_cigar_res_t *)malloc(sizeof(*fr) * n_fix);
		uint32_t *fc;
		char *fmd;
		for (j = 0; j < W.n; ++j) {
			const want_t *x = &W.a[j];
			if (x->fix < 0) continue;
			fq[f].read = x->read, fq[f].qb = x->qb, fq[f].qe = x->qe, fq[f].rb = x->rb, fq[f].re = x->re, fq[f].truesc = INT32_MIN, fq[f].reg_w = o->w;
			fw += (size_t)(x->qe - x->qb) + (size_t)(x->re - x->rb) + 2, fm_ += 3 * ((size_t)(x->qe - x->qb) + (size_t)(x->re - x->rb)) + 16;
			++f;
		}
		fc = (uint32_t *)malloc(4 * fw), fmd = (char *)malloc(fm_);
		rc = fq && fr && fc && fmd ? bmh_reg2cigar_batch(ctx, l_pac, pac, reads, (int64_t)n_fix, fq, fr, fc, fw, fmd, fm_) : BMH_E_NOMEM;
		for (j = 0; j < W.n && !rc; ++j) {
			want_t *x = &W.a[j];
			const uint32_t *cg;
			int64_t xx;
			int k, y;
			if (x->fix < 0) continue;
			cg = fc + fr[x->fix].cigar_off;
			for (k = 0, xx = x->rb, y = x->qb; k < fr[x->fix].n_cigar; ++k) {
				const int op = (int)(cg[k] & 0xf), len = (int)(cg[k] >> 4);
				if (op == 0) {
					if (xx <= x->cb && x->cb < xx + len) x->qb = y + (int)(x->cb - xx), x->rb = x->cb;
					if (xx < x->ce && x->ce <= xx + len) {
						x->qe = y + (int)(x->ce - xx), x->re = x->ce;
						break;
					} else xx += len, y += len;
				} else if (op == 1) y += len;
				else if (op == 2) {
					if (xx <= x->cb && x->cb < xx + len) x->qb = y, x->rb = xx + len;
					if (xx < x->ce && x->ce <= xx + len) {
						x->qe = y, x->re = xx;
						break;
					} else xx += len;
				}
			}
			if (x->qb == x->qe || x->rb == x->re) rc = BMH_E_ARG; /* bwa_fix_xref2 returns -2: the reference aborts */
		}
		free(fq), free(fr), free(fc), free(fmd);
		if (rc) goto done;
	}
	if (W.n) {
		reqs = (bmh_cigar_req_t *)malloc(sizeof(*reqs) * W.n);
		res = (bmh_cigar_res_t *)malloc(sizeof(*res) * W.n);
		if (!reqs || !res) { rc = BMH_E_NOMEM; goto done; }
		for (j = 0; j < W.n; ++j) {
			const want_t *x = &W.a[j];
			const bmh_alnreg_t *ar = &regs[x->read].a[x->k];
			reqs[j].read = x->read, reqs[j].qb = x->qb, reqs[j].qe = x->qe, reqs[j].rb = x->rb, reqs[j].re = x->re;
			reqs[j].truesc = ar->truesc, reqs[j].reg_w = ar->w;
			cw += (size_t)(x->qe - x->qb) + (size_t)(x->re - x->rb) + 2, mb += 3 * ((size_t)(x->qe - x->qb) + (size_t)(x->re - x->rb)) + 16;
		}
		{ /* pools: a CIGAR may have ql+tl+1 operations and an MD 3 bytes per base, but they almost never have more than a
		   * few -- try with 16 words / 48 bytes per region (plus one region's worst case) and only fall back to the sizes
		   * that always suffice if the driver says BMH_E_CIGAR_CAP (hundreds of megabytes of untouched-but-mapped memory
		   * per slice, mapped and unmapped by every host thread at once, cost more than the alignments) */
			size_t cw2 = 16 * W.n + 70000, mb2 = 48 * W.n + 3 * 140000 + 64;
			if (cw2 > cw) cw2 = cw;
			if (mb2 > mb) mb2 = mb;
			cig = (uint32_t *)malloc(4 * cw2), md = (char *)malloc(mb2);
			if (!cig || !md) { rc = BMH_E_NOMEM; goto done; }
			rc = bmh_reg2cigar_batch(ctx, l_pac, pac, reads, (int64_t)W.n, reqs, res, cig, cw2, md, mb2);
			if (rc == BMH_E_CIGAR_CAP && (cw2 < cw || mb2 < mb)) {
				free(cig), free(md);
				cig = (uint32_t *)malloc(4 * cw), md = (char *)malloc(mb);
				if (!cig || !md) { rc = BMH_E_NOMEM; goto done; }
				rc = bmh_reg2cigar_batch(ctx, l_pac, pac, reads, (int64_t)W.n, reqs, res, cig, cw, md, mb);
			}
			if (rc) goto done;
		}
		for (j = 0; j < W.n; ++j) arena_words += (size_t)res[j].n_cigar + 2;
	}

	if (trace) tt[2] = now_s();
	/* ---- pass C: mem_reg2aln's second half (bwamem.c:1203-1235) for every wanted region, then the text */
	alns = (aln_t *)malloc(sizeof(aln_t) * (W.n + 2));
	arena = (uint32_t *)malloc(4 * (arena_words + 4));
	if (!alns || !arena) { rc = BMH_E_NOMEM; goto done; }
	arena_words = 0;
	for (j = 0; j < W.n; ++j) {
		const want_t *x = &W.a[j];
		const bmh_alnreg_t *ar = &regs[x->read].a[x->k];
		const int l_query = seqs[x->read].l_seq;
		aln_t *a = &alns[j];
		const uint32_t *src = cig + res[j].cigar_off;
		uint32_t *dst = arena + arena_words;
		int nc = res[j].n_cigar, is_rev, k, clip5, clip3;
		int64_t pos;
		if (res[j].NM < 0) { rc = BMH_E_ARG; goto done; } /* bwa_gen_cigar2 refused the region (bwa.c:99): cannot happen after the fix */
		memset(a, 0, sizeof(*a));
		a->mapq = ar->secondary < 0 ? bmh_approx_mapq_se(o, ar) : 0;
		if (ar->secondary >= 0) a->flag |= 0x100;
		a->NM = res[j].NM, a->md = md + res[j].md_off;
		pos = depos(l_pac, x->rb < l_pac ? x->rb : x->re - 1, &is_rev);
		a->is_rev = is_rev;
		if (nc > 0) { /* squeeze out a leading or trailing deletion */
			if ((src[0] & 0xf) == 2) pos += src[0] >> 4, ++src, --nc;
			else if ((src[nc - 1] & 0xf) == 2) --nc;
		}
		clip5 = is_rev ? l_query - x->qe : x->qb, clip3 = is_rev ? x->qb : l_query - x->qe;
		a->cig_off = (uint32_t)arena_words;
		k = 0;
		if (x->qb != 0 || x->qe != l_query) { /* soft clipping */
			if (clip5) dst[k++] = (uint32_t)clip5 << 4 | 3;
			memcpy(dst + k, src, 4 * (size_t)nc), k += nc;
			if (clip3) dst[k++] = (uint32_t)clip3 << 4 | 3;
		} else memcpy(dst, src, 4 * (size_t)nc), k = nc;
		a->n_cigar = k, arena_words += (size_t)k;
		a->rid = pos2rid(bns, pos);
		a->pos = pos - bns->anns[a->rid].offset;
		a->score = ar->score, a->sub = ar->sub > ar->csub ? ar->sub : ar->csub;
	}
	for (i = 0; i < n; i += pe ? 2 : 1) {
		const int nr = pe ? 2 : 1;
		aln_t h[2], un;
		int r, extra = 0;
		if (i + 8 < n) { /* a read's name, bases and qualities are three allocations of the host program's: scattered */
			const bmh_seq_t *nx = &seqs[i + 8];
			__builtin_prefetch(nx->name), __builtin_prefetch(nx->seq), __builtin_prefetch(nx->seq + 64), __builtin_prefetch(nx->seq + 128);
			if (nx->qual) __builtin_prefetch(nx->qual), __builtin_prefetch(nx->qual + 64), __builtin_prefetch(nx->qual + 128);
			if (pe) {
				++nx;
				__builtin_prefetch(nx->name), __builtin_prefetch(nx->seq), __builtin_prefetch(nx->seq + 64), __builtin_prefetch(nx->seq + 128);
				if (nx->qual) __builtin_prefetch(nx->qual), __builtin_prefetch(nx->qual + 64), __builtin_prefetch(nx->qual + 128);
			}
		}
		unmapped(&un);
		if (pe) {
			const pairdec_t *d = &pd[i >> 1];
			extra = d->extra_flag;
			if (strcmp(seqs[i].name, seqs[i + 1].name) != 0) {
				fprintf(stderr, "[bwamem_hip] paired reads have different names: \"%s\", \"%s\"\n", seqs[i].name, seqs[i + 1].name);
				rc = BMH_E_ARG;
				goto done;
			}
			for (r = 0; r < 2; ++r) { /* the mate records: the pair's two alignments, or each end's best hit (bwamem_pair.c:311-312,320-324) */
				const size_t f0 = first[i + r], f1 = first[i + r + 1];
				if (d->paired) {
					if (f1 > f0) h[r] = alns[f0], h[r].mapq = d->q_se[r];
					else unmapped(&h[r]);
					h[r].flag |= (r ? 0x80 : 0x40) | extra;
				} else if (f1 > f0 && W.a[f0].k == 0) h[r] = alns[f0]; /* regs[].a[0] with score >= T */
				else unmapped(&h[r]);
			}
			if (d->paired) {
				for (r = 0; r < 2; ++r) {
					str.l = 0;
					aln2sam(bns, &str, &seqs[i + r], 1, &h[r], 0, &h[!r], arena, rg_id);
					st_room(&str, 1), str.s[str.l] = 0;
					seqs[i + r].sam = (char *)malloc(str.l + 1);
					memcpy(seqs[i + r].sam, str.s, str.l + 1);
				}
				continue;
			}
			if (!(o->flag & BMH_MEM_F_NOPAIRING) && h[0].rid == h[1].rid && h[0].rid >= 0) { /* the two top hits make a proper pair? */
				int64_t dist;
				const int dd = infer_dir(l_pac, regs[i].a[0].rb, regs[i + 1].a[0].rb, &dist);
				if (!pes[dd].failed && dist >= pes[dd].low && dist <= pes[dd].high) extra |= 2;
			}
		}
		for (r = 0; r < nr; ++r) { /* mem_reg2sam_se, bwamem.c:1049-1083 */
			const int rd = i + r;
			const size_t f0 = first[rd], f1 = first[rd + 1];
			const int extra_flag = pe ? ((r ? 0x81 : 0x41) | extra) : 0;
			const aln_t *mate = pe ? &h[!r] : 0;
			size_t q;
			str.l = 0;
			for (q = f0; q < f1; ++q) { /* alns[f0..f1) is the reference's `aa` */
				const int k = W.a[q].k;
				const bmh_alnreg_t *p = &regs[rd].a[k];
				aln_t *a = &alns[q];
				a->flag |= extra_flag;
				if (p->secondary >= 0) a->sub = -1;
				if (k && p->secondary < 0) a->flag |= (o->flag & BMH_MEM_F_NO_MULTI) ? 0x10000 : 0x800; /* supplementary */
				if (k && a->mapq > alns[f0].mapq) a->mapq = alns[f0].mapq;
			}
			if (f1 == f0) {
				aln_t t = un;
				t.flag |= extra_flag;
				aln2sam(bns, &str, &seqs[rd], 1, &t, 0, mate, arena, rg_id);
			} else
				for (q = f0; q < f1; ++q) aln2sam(bns, &str, &seqs[rd], (int)(f1 - f0), &alns[f0], (int)(q - f0), mate, arena, rg_id);
			st_room(&str, 1), str.s[str.l] = 0;
			seqs[rd].sam = (char *)malloc(str.l + 1);
			memcpy(seqs[rd].sam, str.s, str.l + 1);
		}
	}
	if (trace)
		fprintf(stderr, "[bwamem_hip] bmh_sam_batch %d reads, %zu alignments: marking + pairing %.1f ms, global alignments (bmh_reg2cigar_batch) %.1f ms, coordinates + text %.1f ms\n",
		        n, W.n, (tt[1] - tt[0]) * 1e3, (tt[2] - tt[1]) * 1e3, (now_s() - tt[2]) * 1e3);
done:
	free(W.a), free(pd), free(reads), free(reqs), free(res), free(cig), free(md), free(first), free(alns), free(arena), free(pv.a), free(pu.a),
	    free(str.s);
	return rc;
}
