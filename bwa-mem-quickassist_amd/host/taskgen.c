/*
 * taskgen.c -- synthetic workload generator for bench.py and the full-size tests
 * (libbmh_taskgen.so).  Not part of the product path and not part of the oracle.
 *
 * There is no genome or FASTQ data on the build or GPU machines, and an hg38-scale
 * FM-index cannot be built there, so the extension kernel is driven
 * with the tasks mem_chain2aln() (reference bwa-0.7.8/bwamem.c:730-878) WOULD build
 * for simulated reads:
 *
 *   reference window = uniform random bases; read = window slice with substitutions,
 *   insertions, deletions (optionally N's and a chimeric random tail);
 *   seeds   = error-free runs of >= min_seed_len bases along the true diagonal (what
 *             SMEM seeding yields on a repeat-free genome, bwamem.c:118-157);
 *   window  = [rmax0,rmax1) over all seeds (bwamem.c:740-751);
 *   tasks   = left + right extension of the longest seed (the other seeds lie inside
 *             the resulting region and are skipped, bwamem.c:769-799):
 *             left : reversed flanks, h0 = len*a, end_bonus = pen_clip5   (bwamem.c:810-829)
 *             right: forward flanks,  h0 ~ left score, end_bonus = pen_clip3 (bwamem.c:841-857)
 *
 * The right task's h0 is the exact score of the true left alignment floored at the
 * seed score (the real value needs the left DP first); parity is unaffected because GPU
 * and oracle consume the same task records.
 */
#include <stdlib.h>
#include <string.h>

#include "../../include/bwamem_hip.h"

typedef struct {
	uint64_t seed;
	int32_t len_min, len_max;   /* read length range                         */
	int32_t min_seed_len;       /* 19, bwamem.c:58                            */
	int32_t max_indel;          /* indel lengths uniform in 1..max_indel     */
	double p_sub, p_ins, p_del; /* per-base event probabilities              */
	double p_n;                 /* per-base probability of an N in the read  */
	double p_chimera;           /* probability the read tail is unrelated    */
} bmh_taskgen_cfg_t;

static inline uint64_t splitmix(uint64_t *s)
{
	uint64_t z = (*s += 0x9e3779b97f4a7c15ULL);
	z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
	z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
	return z ^ (z >> 31);
}
static inline double urand(uint64_t *s) { return (double)(splitmix(s) >> 11) * (1.0 / 9007199254740992.0); }
static inline int irand(uint64_t *s, int lo, int hi) { return lo + (int)(splitmix(s) % (uint64_t)(hi - lo + 1)); }

static int cal_max_gap(const bmh_params_t *p, int qlen) /* bwamem.c:544-551 */
{
	int l_del = (int)((double)(qlen * p->a - p->o_del) / p->e_del + 1.);
	int l_ins = (int)((double)(qlen * p->a - p->o_ins) / p->e_ins + 1.);
	int l = l_del > l_ins ? l_del : l_ins;
	l = l > 1 ? l : 1;
	return l < p->w << 1 ? l : p->w << 1;
}

/* Upper bounds so callers can size buffers: bytes of pool and tasks per read. */
size_t bmh_taskgen_pool_bound(const bmh_taskgen_cfg_t *c, const bmh_params_t *p)
{
	return (size_t)c->len_max * 4 + (size_t)p->w * 4 + 64;
}

/* One simulated read: window `ref[0..R)`, read `rd[0..*L)`, clean-copy map `rpos`, the longest seed and the chain window
 * [rmax0,rmax1) over all seeds (bwamem.c:740-751).  Returns 0 for a read that yields no seed (no extension work). */
typedef struct {
	int L, best_q, best_len, best_r;
	int64_t rmax0, rmax1;
} sim_read_t;

static int sim_read(const bmh_taskgen_cfg_t *cfg, const bmh_params_t *p, uint64_t *sp, int G, int R, uint8_t *ref, uint8_t *rd,
                    int *rpos, sim_read_t *o)
{
	uint64_t s = *sp;
	const int L = irand(&s, cfg->len_min, cfg->len_max);
	int i, x, n = 0, best_q = -1, best_len = 0, best_r = -1, run_q = 0, run_len = 0;
	int chim_at = (cfg->p_chimera > 0 && urand(&s) < cfg->p_chimera && L > 40) ? irand(&s, 20, L - 1) : L + 1;
	int64_t rmax0 = 1 << 30, rmax1 = 0;
	for (i = 0; i < R; ++i) ref[i] = (uint8_t)(splitmix(&s) & 3);
	/* read = ref[G ...] with errors */
	for (x = G; n < L && x < R - 1;) {
		const double u = urand(&s);
		if (n >= chim_at) { rd[n] = (uint8_t)(splitmix(&s) & 3), rpos[n] = -1, ++n, ++x; continue; }
		if (u < cfg->p_sub) rd[n] = (uint8_t)((ref[x] + 1 + splitmix(&s) % 3) & 3), rpos[n] = -1, ++n, ++x;
		else if (u < cfg->p_sub + cfg->p_ins) {
			int k = irand(&s, 1, cfg->max_indel);
			for (; k > 0 && n < L; --k) rd[n] = (uint8_t)(splitmix(&s) & 3), rpos[n] = -1, ++n;
		} else if (u < cfg->p_sub + cfg->p_ins + cfg->p_del) x += irand(&s, 1, cfg->max_indel);
		else rd[n] = ref[x], rpos[n] = x, ++n, ++x;
	}
	if (n < L) { *sp = s; return 0; }
	if (cfg->p_n > 0)
		for (i = 0; i < L; ++i)
			if (urand(&s) < cfg->p_n) rd[i] = 4, rpos[i] = -1;
	*sp = s;
	/* seeds = maximal clean diagonal runs >= min_seed_len; chain window over all of them */
	for (i = 0; i <= L; ++i) {
		const int cont = i < L && rpos[i] >= 0 && run_len > 0 && rpos[i] == rpos[i - 1] + 1;
		if (cont) { ++run_len; continue; }
		if (run_len >= cfg->min_seed_len) {
			const int qb = run_q, rb = rpos[run_q], rest = L - qb - run_len;
			const int64_t b = rb - (qb + cal_max_gap(p, qb)), e = rb + run_len + (rest + cal_max_gap(p, rest));
			if (b < rmax0) rmax0 = b;
			if (e > rmax1) rmax1 = e;
			if (run_len > best_len) best_len = run_len, best_q = qb, best_r = rb;
		}
		if (i < L && rpos[i] >= 0) run_q = i, run_len = 1;
		else run_len = 0;
	}
	if (best_len == 0) return 0; /* unseeded read: no extension work */
	if (rmax0 < 0) rmax0 = 0;
	if (rmax1 > R) rmax1 = R;
	o->L = L, o->best_q = best_q, o->best_len = best_len, o->best_r = best_r, o->rmax0 = rmax0, o->rmax1 = rmax1;
	return 1;
}

/* Generates tasks for reads [0,n_reads).  Returns the number of tasks, or -1 if a
 * capacity is too small.  task_read (nullable) receives the read index of each task. */
int64_t bmh_taskgen_ext(const bmh_taskgen_cfg_t *cfg, const bmh_params_t *p, int64_t n_reads, uint8_t *pool,
                        size_t pool_cap, size_t *pool_used, bmh_ext_task_t *tasks, int64_t task_cap,
                        uint32_t *task_read)
{
	const int Lmax = cfg->len_max;
	const int G = (p->w << 1) + Lmax + 8; /* flank of reference kept on each side */
	const int R = Lmax * 2 + 2 * G + 64;
	uint8_t *ref = (uint8_t *)malloc((size_t)R), *rd = (uint8_t *)malloc((size_t)Lmax + 64);
	int *rpos = (int *)malloc(sizeof(int) * ((size_t)Lmax + 64)); /* ref index of a copied base, -1 = not a clean copy */
	int64_t nt = 0, r;
	size_t used = 0;
	uint64_t s = cfg->seed;

	for (r = 0; r < n_reads; ++r) {
		sim_read_t o;
		int i;
		if (!sim_read(cfg, p, &s, G, R, ref, rd, rpos, &o)) continue;
		{
			const int L = o.L, best_q = o.best_q, best_len = o.best_len, best_r = o.best_r;
			const int64_t rmax0 = o.rmax0, rmax1 = o.rmax1;
			if (used + (size_t)L + (size_t)(rmax1 - rmax0) + 16 > pool_cap || nt + 2 > task_cap) {
				nt = -1;
				break;
			}
			{
				const uint64_t read_off = used, win_off = used + (uint64_t)L;
				int lsc = best_len * p->a;
				memcpy(pool + read_off, rd, (size_t)L);
				memcpy(pool + win_off, ref + rmax0, (size_t)(rmax1 - rmax0));
				used += (size_t)L + (size_t)(rmax1 - rmax0);
				if (best_q > 0) { /* left extension */
					bmh_ext_task_t *t = &tasks[nt];
					const int tl = (int)(best_r - rmax0);
					int sc = 0, bestsc = 0;
					memset(t, 0, sizeof(*t));
					t->q_off = read_off + (uint64_t)(best_q - 1), t->t_off = win_off + (uint64_t)(tl > 0 ? tl - 1 : 0);
					t->qlen = (uint16_t)best_q, t->tlen = (uint16_t)tl, t->h0 = best_len * p->a;
					t->w = (int16_t)p->w, t->end_bonus = (int16_t)p->pen_clip5, t->flags = BMH_F_QREV | BMH_F_TREV;
					if (task_read) task_read[nt] = (uint32_t)r;
					++nt;
					for (i = best_q - 1; i >= 0; --i) { /* crude score of the true left alignment, for the right h0 */
						sc += rpos[i] >= 0 ? p->a : -4;
						if (sc > bestsc) bestsc = sc;
					}
					lsc += bestsc;
				}
				if (best_q + best_len < L) { /* right extension */
					bmh_ext_task_t *t = &tasks[nt];
					const int qe = best_q + best_len;
					const int64_t re = best_r + best_len - rmax0;
					memset(t, 0, sizeof(*t));
					t->q_off = read_off + (uint64_t)qe, t->t_off = win_off + (uint64_t)re;
					t->qlen = (uint16_t)(L - qe), t->tlen = (uint16_t)(rmax1 - rmax0 - re), t->h0 = lsc;
					t->w = (int16_t)p->w, t->end_bonus = (int16_t)p->pen_clip3;
					if (task_read) task_read[nt] = (uint32_t)r;
					++nt;
				}
			}
		}
	}
	free(ref), free(rd), free(rpos);
	if (pool_used) *pool_used = used;
	return nt;
}

/* The same simulated reads as ONE fused record per read (bmh_seed_task_t: the longest seed of the read's chain with its
 * window) -- what the per-seed extension entry point consumes: the right extension then starts from the score the
 * device computed for the left one (bwamem.c:842,854), not from a guess.  One task per seeded read; the read and its
 * window lie back to back in the pool.  Returns the number of tasks, -1 if a capacity is too small. */
int64_t bmh_taskgen_seed(const bmh_taskgen_cfg_t *cfg, const bmh_params_t *p, int64_t n_reads, uint8_t *pool, size_t pool_cap,
                         size_t *pool_used, bmh_seed_task_t *tasks, int64_t task_cap)
{
	const int Lmax = cfg->len_max;
	const int G = (p->w << 1) + Lmax + 8;
	const int R = Lmax * 2 + 2 * G + 64;
	uint8_t *ref = (uint8_t *)malloc((size_t)R), *rd = (uint8_t *)malloc((size_t)Lmax + 64);
	int *rpos = (int *)malloc(sizeof(int) * ((size_t)Lmax + 64));
	int64_t nt = 0, r;
	size_t used = 0;
	uint64_t s = cfg->seed;
	for (r = 0; r < n_reads; ++r) {
		sim_read_t o;
		bmh_seed_task_t *t;
		if (!sim_read(cfg, p, &s, G, R, ref, rd, rpos, &o)) continue;
		if (used + (size_t)o.L + (size_t)(o.rmax1 - o.rmax0) + 16 > pool_cap || nt + 1 > task_cap) {
			nt = -1;
			break;
		}
		t = &tasks[nt++];
		memset(t, 0, sizeof(*t));
		memcpy(pool + used, rd, (size_t)o.L);
		memcpy(pool + used + o.L, ref + o.rmax0, (size_t)(o.rmax1 - o.rmax0));
		t->q_off = used, t->t_off = used + (uint64_t)o.L, t->l_query = o.L, t->qbeg = o.best_q, t->len = o.best_len;
		t->rbeg = (int32_t)(o.best_r - o.rmax0), t->wlen = (int32_t)(o.rmax1 - o.rmax0);
		used += (size_t)o.L + (size_t)(o.rmax1 - o.rmax0);
	}
	free(ref), free(rd), free(rpos);
	if (pool_used) *pool_used = used;
	return nt;
}

/* Global-alignment tasks as bwa_gen_cigar2 hands them to ksw_global2 (reference bwa.c:116-132): the whole read
 * against the window it came from, band = |tlen-qlen| + 3 .. +3+wspread (measured mean 19).
 * Returns the number of tasks (one per read) or -1 if a capacity is too small. */
int64_t bmh_taskgen_glb(const bmh_taskgen_cfg_t *cfg, int64_t n_reads, int wspread, uint8_t *pool, size_t pool_cap,
                        size_t *pool_used, bmh_glb_task_t *tasks, int64_t task_cap, uint64_t *cigar_words)
{
	const int Lmax = cfg->len_max;
	uint8_t *ref = (uint8_t *)malloc((size_t)Lmax * 2 + 64), *rd = (uint8_t *)malloc((size_t)Lmax + 64);
	int64_t nt = 0, r;
	size_t used = 0;
	uint64_t s = cfg->seed, cw = 0;
	for (r = 0; r < n_reads; ++r) {
		const int L = irand(&s, cfg->len_min, cfg->len_max);
		int i, x, n = 0;
		for (i = 0; i < 2 * Lmax + 32; ++i) ref[i] = (uint8_t)(splitmix(&s) & 3);
		for (x = 0; n < L && x < 2 * Lmax;) {
			const double u = urand(&s);
			if (u < cfg->p_sub) rd[n++] = (uint8_t)((ref[x++] + 1 + splitmix(&s) % 3) & 3);
			else if (u < cfg->p_sub + cfg->p_ins) {
				int k = irand(&s, 1, cfg->max_indel);
				for (; k > 0 && n < L; --k) rd[n++] = (uint8_t)(splitmix(&s) & 3);
			} else if (u < cfg->p_sub + cfg->p_ins + cfg->p_del) x += irand(&s, 1, cfg->max_indel);
			else rd[n++] = ref[x++];
		}
		if (n < L || x < 1) continue;
		if (used + (size_t)L + (size_t)x + 16 > pool_cap || nt + 1 > task_cap) { nt = -1; break; }
		{
			bmh_glb_task_t *t = &tasks[nt++];
			memcpy(pool + used, rd, (size_t)L);
			memcpy(pool + used + L, ref, (size_t)x);
			t->q_off = used, t->t_off = used + (uint64_t)L, t->qlen = (uint16_t)L, t->tlen = (uint16_t)x;
			t->w = abs(x - L) + 3 + irand(&s, 0, wspread);
			t->cigar_off = (uint32_t)cw, t->cigar_cap = (uint32_t)(L + x + 2);
			cw += t->cigar_cap;
			used += (size_t)L + (size_t)x;
		}
	}
	free(ref), free(rd);
	if (pool_used) *pool_used = used;
	if (cigar_words) *cigar_words = cw;
	return nt;
}

/* Mate-rescue tasks as mem_matesw hands them to ksw_align2 (reference bwamem_pair.c:109-175): the mate (read length
 * from cfg) against the insert-size window [low,high] + l_ms of the reference; with probability p_hit the window holds
 * the mate's true locus (error model of cfg), otherwise it is unrelated sequence (the rescue fails, score < minsc).
 * xtra = KSW_XSUBO | KSW_XSTART | (l_ms*a < 250 ? KSW_XBYTE : 0) | min_seed_len*a  (bwamem_pair.c:147).
 * Returns the number of tasks (one per mate) or -1 if a capacity is too small. */
int64_t bmh_taskgen_sw(const bmh_taskgen_cfg_t *cfg, const bmh_params_t *p, int64_t n, int win_min, int win_max,
                       double p_hit, uint8_t *pool, size_t pool_cap, size_t *pool_used, bmh_sw_task_t *tasks,
                       int64_t task_cap)
{
	const int Lmax = cfg->len_max;
	uint8_t *ref = (uint8_t *)malloc((size_t)win_max + 2 * (size_t)Lmax + 64), *rd = (uint8_t *)malloc((size_t)Lmax + 64);
	int64_t nt = 0, r;
	size_t used = 0;
	uint64_t s = cfg->seed;
	for (r = 0; r < n; ++r) {
		const int L = irand(&s, cfg->len_min, cfg->len_max);
		const int W = irand(&s, win_min, win_max) + L;
		int i, x, k = 0;
		for (i = 0; i < W; ++i) ref[i] = (uint8_t)(splitmix(&s) & 3);
		if (urand(&s) < p_hit) {
			for (x = irand(&s, 0, W - L > 0 ? W - L : 0); k < L && x < W;) {
				const double u = urand(&s);
				if (u < cfg->p_sub) rd[k++] = (uint8_t)((ref[x++] + 1 + splitmix(&s) % 3) & 3);
				else if (u < cfg->p_sub + cfg->p_ins) {
					int g = irand(&s, 1, cfg->max_indel);
					for (; g > 0 && k < L; --g) rd[k++] = (uint8_t)(splitmix(&s) & 3);
				} else if (u < cfg->p_sub + cfg->p_ins + cfg->p_del) x += irand(&s, 1, cfg->max_indel);
				else rd[k++] = ref[x++];
			}
		}
		for (; k < L; ++k) rd[k] = (uint8_t)(splitmix(&s) & 3);
		if (used + (size_t)L + (size_t)W + 16 > pool_cap || nt + 1 > task_cap) { nt = -1; break; }
		{
			bmh_sw_task_t *t = &tasks[nt++];
			memset(t, 0, sizeof(*t));
			memcpy(pool + used, rd, (size_t)L);
			memcpy(pool + used + L, ref, (size_t)W);
			t->q_off = used, t->t_off = used + (uint64_t)L, t->qlen = (uint16_t)L, t->tlen = (uint32_t)W;
			t->xtra = BMH_SW_XSUBO | BMH_SW_XSTART | (L * p->a < 250 ? BMH_SW_XBYTE : 0) | (uint32_t)(cfg->min_seed_len * p->a);
			used += (size_t)L + (size_t)W;
		}
	}
	free(ref), free(rd);
	if (pool_used) *pool_used = used;
	return nt;
}

/* Re-lays a fused-record pool as [all reads | all windows] (bmh_taskgen_seed puts every read in front of its window) and rewrites the
 * records' offsets: the windows stand for reference bases, which the preload shim keeps resident in HBM (bmh_ctx_set_pac), so a
 * host-fed measurement uploads only the first region per batch.  *reads_bytes = size of the reads region (a multiple of 64).
 * Returns the bytes used in pool_out, 0 if it is too small. */
size_t bmh_taskgen_split(const uint8_t *pool_in, bmh_seed_task_t *tasks, int64_t n, uint8_t *pool_out, size_t out_cap, size_t *reads_bytes)
{
	size_t R = 0, W = 0, q = 0, w;
	int64_t k;
	for (k = 0; k < n; ++k) R += (size_t)tasks[k].l_query, W += (size_t)tasks[k].wlen;
	R = (R + 63) & ~(size_t)63;
	if (R + W + 16 > out_cap) return 0;
	w = R;
	for (k = 0; k < n; ++k) {
		bmh_seed_task_t *t = &tasks[k];
		memcpy(pool_out + q, pool_in + t->q_off, (size_t)t->l_query);
		memcpy(pool_out + w, pool_in + t->t_off, (size_t)t->wlen);
		t->q_off = q, t->t_off = w;
		q += (size_t)t->l_query, w += (size_t)t->wlen;
	}
	memset(pool_out + q, 0, R - q);
	memset(pool_out + w, 0, 16);
	if (reads_bytes) *reads_bytes = R;
	return w + 16;
}

/* The same for ksw_global2 tasks: pool_out = [all queries | all targets].  The targets stand for reference windows, which the region record
 * (bmh_region_cigar_batch) fetches from the resident 2-bit reference on the device, so a host-fed measurement ships only the queries.
 * *query_bytes = size of the queries region (a multiple of 64).  Returns the bytes used in pool_out, 0 if it is too small. */
size_t bmh_taskgen_split_glb(const uint8_t *pool_in, bmh_glb_task_t *tasks, int64_t n, uint8_t *pool_out, size_t out_cap, size_t *query_bytes)
{
	size_t Q = 0, T = 0, q = 0, w;
	int64_t k;
	for (k = 0; k < n; ++k) Q += (size_t)tasks[k].qlen, T += (size_t)tasks[k].tlen;
	Q = (Q + 63) & ~(size_t)63;
	if (Q + T + 16 > out_cap) return 0;
	w = Q;
	for (k = 0; k < n; ++k) {
		bmh_glb_task_t *t = &tasks[k];
		memcpy(pool_out + q, pool_in + t->q_off, (size_t)t->qlen);
		memcpy(pool_out + w, pool_in + t->t_off, (size_t)t->tlen);
		t->q_off = q, t->t_off = w;
		q += (size_t)t->qlen, w += (size_t)t->tlen;
	}
	memset(pool_out + q, 0, Q - q);
	memset(pool_out + w, 0, 16);
	if (query_bytes) *query_bytes = Q;
	return w + 16;
}
