/*
 * reg2cigar_batch.c -- batched CIGAR generation: the caller of ksw_global2 on BWA-MEM's path
 * (SURVEY.md §8 row a6), host side, plain C, above the C-ABI.
 *
 * For a batch of alignment regions it does what mem_reg2aln() does per region between
 * bwa_fix_xref2 and the clipping (reference bwa-0.7.8/bwamem.c:1187-1201):
 *     w2 = max(infer_bw(del), infer_bw(ins)) capped by the region's band     bwamem.c:884-891,1187-1191
 *     up to 3 x { bwa_gen_cigar2(w2); stop if the score repeats; w2 <<= 1 }
 *                 while score < truesc - a                                    bwamem.c:1193-1201
 * and inside each try what bwa_gen_cigar2() does (bwa.c:89-172): fetch [rb,re) from the 2-bit
 * reference, reverse query and reference when the hit is on the reverse strand (so indels are
 * left-aligned), pick the band (bwa.c:116-125), run the banded global alignment, derive NM and MD.
 * bwa_gen_cigar2 is a pure function of (region, band): the global alignments of ALL the bands a region could be
 * tried with form ONE bmh_global_batch() per call, and the loop is replayed over the results (round 1 ran one
 * batch per try).  Results are bit-identical to the per-region reference calls.
 *
 * Two paths.  With the 2-bit reference resident on the device (bmh_ctx_set_pac -- the preload shim's default) the call is ONE
 * bmh_region_cigar_batch(): the host lays out pools and computes bands, which needs no sequence byte; window fetch, orientation, the
 * no-gap score, the alignments, the replay of the loop, NM and MD are the device's (csrc/region_cigar.hip).  The few regions whose
 * CIGAR or MD outgrow the fixed slots coming back, and every call on a context without the resident reference, take the older
 * path below it: oriented copies made on the host, bmh_global_batch(), replay, NM and MD on the host.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../../include/bwamem_hip.h"

const bmh_params_t *bmh_ctx_params_(const bmh_ctx_t *ctx);
int bmh_upload_pool(bmh_ctx_t *ctx, const uint8_t *pool, size_t bytes);

static double now_s(void) /* the clock of the BMH_DRIVER_TRACE lines: wall time, or with BMH_TRACE_CPU this thread's CPU time */
{
	static int cpu = -1;
	struct timespec ts;
	if (cpu < 0) cpu = getenv("BMH_TRACE_CPU") != 0;
	clock_gettime(cpu ? CLOCK_THREAD_CPUTIME_ID : CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* [beg,end) of the 2-bit reference as one code per byte (comp: complemented), four bases per table look-up: the per-base
 * form (bntseq.c:355-376) is a third of this driver's host time on 150 bp regions. */
static uint32_t pac_lut[2][256];
static volatile int pac_lut_ready;
static void pac_codes(const uint8_t *pac, int64_t beg, int64_t end, int comp, uint8_t *dst)
{
	int64_t x = beg;
	const uint32_t *lut;
	if (!pac_lut_ready) { /* (filled with the same values by whoever gets here first: a benign race) */
		int b, j;
		for (b = 0; b < 256; ++b) {
			uint32_t f = 0, c = 0;
			for (j = 0; j < 4; ++j) {
				const uint32_t v = (uint32_t)(b >> ((3 - j) << 1) & 3);
				f |= v << (8 * j), c |= (3 - v) << (8 * j);
			}
			pac_lut[0][b] = f, pac_lut[1][b] = c;
		}
		__sync_synchronize();
		pac_lut_ready = 1;
	}
	lut = pac_lut[comp != 0];
	for (; x < end && (x & 3); ++x) *dst++ = (uint8_t)(lut[pac[x >> 2]] >> ((x & 3) << 3));
	for (; x + 4 <= end; x += 4, dst += 4) memcpy(dst, &lut[pac[x >> 2]], 4);
	for (; x < end; ++x) *dst++ = (uint8_t)(lut[pac[x >> 2]] >> ((x & 3) << 3));
}

enum { SMALL_CAP = 24 }; /* CIGAR slots reserved per task on the first attempt of a try */

static void put_num(char *s, size_t *l, int c) /* kputw, kstring.h:62-77, c >= 0 */
{
	char buf[16];
	int n = 0;
	if (c == 0) { s[(*l)++] = '0'; return; }
	for (; c > 0; c /= 10) buf[n++] = (char)('0' + c % 10);
	while (n) s[(*l)++] = buf[--n];
}

/* bwa.c:134-164 on oriented copies; md must have room for 3*(ql+tl)+16 bytes.  Returns NM, *md_len without the NUL.
 * `run` = matches since the last MD token. */
static int nm_md(int n_cigar, const uint32_t *cigar, const uint8_t *q, const uint8_t *t, int rev, char *md, size_t *md_len)
{
	const char *letter = rev ? "TGCAN" : "ACGTN";
	int c, i, qpos = 0, tpos = 0, run = 0, mismatches = 0, gap_bases = 0;
	size_t l = 0;
	for (c = 0; c < n_cigar; ++c) {
		const int op = (int)(cigar[c] & 0xf), len = (int)(cigar[c] >> 4);
		switch (op) {
		case 0: /* M: a mismatch closes the run and names the reference base */
			for (i = 0; i < len; ++i) {
				if (q[qpos + i] == t[tpos + i]) { ++run; continue; }
				put_num(md, &l, run);
				md[l++] = letter[t[tpos + i]];
				++mismatches, run = 0;
			}
			qpos += len, tpos += len;
			break;
		case 2: /* D: "^" + the deleted reference bases, unless it is the first or the last operation */
			if (c > 0 && c < n_cigar - 1) {
				put_num(md, &l, run);
				md[l++] = '^';
				for (i = 0; i < len; ++i) md[l++] = letter[t[tpos + i]];
				run = 0, gap_bases += len;
			}
			tpos += len;
			break;
		case 1: /* I */
			qpos += len, gap_bases += len;
			break;
		default: break;
		}
	}
	put_num(md, &l, run);
	md[l] = 0;
	*md_len = l;
	return mismatches + gap_bases;
}

static int infer_bw(int l1, int l2, int score, int a, int q, int r) /* bwamem.c:884-891 */
{
	int w;
	if (l1 == l2 && l1 * a - score < (q + r - a) << 1) return 0;
	w = (int)((double)((l1 < l2 ? l1 : l2) * a - score - q) / r + 2.);
	if (w < abs(l1 - l2)) w = abs(l1 - l2);
	return w;
}

typedef struct {
	uint64_t q_off, t_off; /* oriented copies in the pool */
	int ql, tl, w2, last_sc, tries, active, rev, valid, nodp;
	int score, n_cigar;
	int64_t tidx[3];    /* the GPU task of each of the (up to) three tries, -1 = none */
	int64_t final_task; /* ... and the one whose result the loop ends on */
	const uint32_t *cig; /* the final try's CIGAR words (in one of the scratch arrays); NULL = the no-DP case, one M run */
} cg_t;

/* band of try t_ of a region whose inferred band is w2 (bwa.c:116-125) */
static int try_band(const bmh_params_t *p, int ql, int tl, int w2)
{
	const int max_ins = (int)((double)(((ql + 1) >> 1) * p->mat[0] - p->o_ins) / p->e_ins + 1.);
	const int max_del = (int)((double)(((ql + 1) >> 1) * p->mat[0] - p->o_del) / p->e_del + 1.);
	int max_gap = max_ins > max_del ? max_ins : max_del, w, min_w;
	max_gap = max_gap > 1 ? max_gap : 1;
	w = (max_gap + abs(tl - ql) + 1) >> 1;
	w = w < w2 ? w : w2;
	min_w = abs(tl - ql) + 3;
	return w > min_w ? w : min_w;
}

/* the band mem_reg2aln starts with (bwamem.c:1187-1191), or reg_w for bwa_fix_xref2's single call (bwa.c:198) */
static int first_band(const bmh_params_t *p, const bmh_cigar_req_t *r, int ql, int tl)
{
	const int tmp = infer_bw(ql, tl, r->truesc, p->a, p->o_del, p->e_del);
	int w2 = infer_bw(ql, tl, r->truesc, p->a, p->o_ins, p->e_ins);
	w2 = w2 > tmp ? w2 : tmp;
	if (w2 > p->w) w2 = w2 < r->reg_w ? w2 : r->reg_w;
	if (r->truesc == INT32_MIN) w2 = r->reg_w;
	return w2;
}

/* ---- the path with oriented copies made on the host.  Packs its CIGARs and MD strings from *cig_used_ / *md_used_ on. */
static int reg2cigar_host_copies(bmh_ctx_t *ctx, const bmh_params_t *p, int64_t l_pac, const uint8_t *pac, const bmh_read_t *reads,
                                 int64_t n_req, const bmh_cigar_req_t *reqs, bmh_cigar_res_t *res, uint32_t *cigar_pool,
                                 size_t cigar_words, size_t *cig_used_, char *md_pool, size_t md_bytes, size_t *md_used_)
{
	cg_t *cg = 0;
	uint8_t *pool = 0;
	bmh_glb_task_t *tasks = 0;
	bmh_glb_result_t *gres = 0;
	uint32_t *scratch = 0, *bigscr = 0; /* CIGAR words of the batch's tasks (24 slots each); of the few redone with full slots */
	int64_t *owner = 0, k, task_cap = 0;
	size_t pool_bytes = 0, cig_used = *cig_used_, md_used = *md_used_;
	int rc = BMH_OK;
	const int trace = getenv("BMH_DRIVER_TRACE") != 0; /* where a call's time goes, on stderr */
	double tt[4] = {0, 0, 0, 0}, t0 = 0;

	if (n_req == 0) return BMH_OK;

	cg = (cg_t *)calloc((size_t)n_req, sizeof(cg_t));
	if (!cg) return BMH_E_NOMEM;
	if (trace) tt[0] = now_s();
	for (k = 0; k < n_req; ++k) { /* layout of the oriented sequence copies */
		const bmh_cigar_req_t *r = &reqs[k];
		cg_t *c = &cg[k];
		const int ql = r->qe - r->qb;
		const int64_t tl = r->re - r->rb;
		memset(&res[k], 0, sizeof(res[k]));
		res[k].NM = -1;
		c->valid = !(ql <= 0 || r->rb >= r->re || (r->rb < l_pac && r->re > l_pac) || r->rb < 0 || r->re > l_pac << 1); /* bwa.c:99-101 */
		if (!c->valid) continue;
		if (ql > 65535 || tl > 65535) { rc = BMH_E_RANGE; goto done; }
		c->ql = ql, c->tl = (int)tl, c->rev = r->rb >= l_pac;
		c->q_off = pool_bytes, pool_bytes += (size_t)ql;
		c->t_off = pool_bytes, pool_bytes += (size_t)tl;
	}
	pool = (uint8_t *)malloc(pool_bytes + 16);
	task_cap = n_req + n_req / 4 + 64;
	tasks = (bmh_glb_task_t *)malloc(sizeof(*tasks) * (size_t)task_cap);
	gres = (bmh_glb_result_t *)malloc(sizeof(*gres) * (size_t)task_cap);
	owner = (int64_t *)malloc(sizeof(int64_t) * (size_t)n_req);
	if (!pool || !tasks || !gres || !owner) { rc = BMH_E_NOMEM; goto done; }

	for (k = 0; k < n_req; ++k) { /* oriented copies (bwa.c:100-107) and the initial band (bwamem.c:1187-1191) */
		const bmh_cigar_req_t *r = &reqs[k];
		cg_t *c = &cg[k];
		const uint8_t *rq;
		uint8_t *q, *t;
		int i;
		if (k + 8 < n_req) { /* the reads are scattered allocations of the host program's, the reference is hundreds of megabytes */
			const bmh_cigar_req_t *nx = &reqs[k + 8];
			const uint8_t *ns = reads[nx->read].seq + nx->qb;
			const int64_t np = nx->rb >= l_pac ? (l_pac << 1) - nx->re : nx->rb;
			__builtin_prefetch(ns), __builtin_prefetch(ns + 64), __builtin_prefetch(ns + 128);
			if (np >= 0 && np < l_pac) __builtin_prefetch(pac + (np >> 2));
		}
		if (!c->valid) continue;
		rq = reads[r->read].seq + r->qb, q = pool + c->q_off, t = pool + c->t_off;
		if (!c->rev) {
			memcpy(q, rq, (size_t)c->ql);
			pac_codes(pac, r->rb, r->re, 0, t);
		} else { /* reverse-strand window = complement read backwards (bntseq.c:364-368), then reversed again (bwa.c:105) */
			const int64_t lo = (l_pac << 1) - 1 - r->re; /* forward coordinates (lo, hi] */
			for (i = 0; i < c->ql; ++i) q[i] = rq[c->ql - 1 - i];
			pac_codes(pac, lo + 1, lo + 1 + c->tl, 1, t);
		}
		c->w2 = first_band(p, r, c->ql, c->tl);
		c->last_sc = -(1 << 30), c->active = 1;
	}
	memset(pool + pool_bytes, 0, 16);
	if (trace) tt[1] = now_s();

	/* ---- bwamem.c:1194-1201.  The loop tries up to three bands (w2, 2*w2, 4*w2) and stops when the score repeats or is
	 * good enough.  bwa_gen_cigar2 is a pure function of (region, band), so all the bands a region COULD be tried with are
	 * aligned in ONE GPU batch and the loop is then replayed over the results: one device round trip per call instead of
	 * up to three (under eight host threads each costs ~10 ms whatever its size).  A try whose band equals an earlier
	 * try's (the band saturates at bwa.c:119-124) shares its task. */
	{
		size_t slot = 0;
		int64_t n_tasks = 0, n_big = 0;
		for (k = 0; k < n_req; ++k) {
			cg_t *c = &cg[k];
			const int single = reqs[k].truesc == INT32_MIN;
			int t_, prev_w = -1;
			c->tidx[0] = c->tidx[1] = c->tidx[2] = -1;
			if (!c->active) continue;
			if (c->ql == c->tl && c->w2 == 0) { /* no gap, no DP: bwa.c:108-114 (0 << k stays 0) */
				const uint8_t *q = pool + c->q_off, *t = pool + c->t_off;
				int i, sc = 0;
				for (i = 0; i < c->ql; ++i) sc += p->mat[t[i] * 5 + q[i]];
				c->nodp = 1, c->score = sc;
				continue;
			}
			for (t_ = 0; t_ < (single ? 1 : 3); ++t_) { /* band of this try, bwa.c:116-125 */
				const int w2 = c->w2 << t_;
				int w;
				bmh_glb_task_t *t;
				if (c->ql == c->tl && w2 == 0) break;
				w = try_band(p, c->ql, c->tl, w2);
				if (w == prev_w) { c->tidx[t_] = c->tidx[t_ - 1]; continue; }
				prev_w = w;
				if (n_tasks == task_cap) {
					task_cap = task_cap + task_cap / 2 + 1024;
					tasks = (bmh_glb_task_t *)realloc(tasks, sizeof(*tasks) * (size_t)task_cap);
					gres = (bmh_glb_result_t *)realloc(gres, sizeof(*gres) * (size_t)task_cap);
					if (!tasks || !gres) { rc = BMH_E_NOMEM; goto done; }
				}
				t = &tasks[n_tasks];
				t->q_off = c->q_off, t->t_off = c->t_off, t->qlen = (uint16_t)c->ql, t->tlen = (uint16_t)c->tl, t->w = w;
				/* a CIGAR can have ql+tl+1 operations but almost never has more than a few: reserve SMALL_CAP slots, so that the
				 * words coming back over PCIe are not 99 % padding, and redo the rare task that needs more (below) */
				t->cigar_off = (uint32_t)slot, t->cigar_cap = (uint32_t)(c->ql + c->tl + 2 < SMALL_CAP ? c->ql + c->tl + 2 : SMALL_CAP);
				slot += t->cigar_cap;
				c->tidx[t_] = n_tasks++;
			}
		}
		if (!(scratch = (uint32_t *)malloc(4 * (slot + 8)))) { rc = BMH_E_NOMEM; goto done; }
		if (n_tasks > 0) {
			if (trace) t0 = now_s();
			if ((rc = bmh_upload_pool(ctx, pool, pool_bytes + 16))) goto done;
			rc = bmh_global_batch(ctx, 0, 0, tasks, n_tasks, gres, scratch, slot + 4);
			if (trace) tt[2] += now_s() - t0;
			if (rc && rc != BMH_E_CIGAR_CAP) goto done;
			rc = BMH_OK;
		}
		for (k = 0; k < n_req; ++k) { /* replay the loop (bwamem.c:1194-1201) over the precomputed tries */
			cg_t *c = &cg[k];
			const int single = reqs[k].truesc == INT32_MIN;
			int t_;
			if (!c->active) continue;
			c->active = 0;
			if (c->nodp) { /* the same score at every band: try 1, and try 2 if try 1 was not good enough (then score == last_sc) */
				c->cig = 0; /* one match run of ql bases: written out at the end */
				c->n_cigar = 1, c->tries = (!single && c->score < reqs[k].truesc - p->a) ? 2 : 1;
				continue;
			}
			for (t_ = 0;; ++t_) {
				const int64_t ti = c->tidx[t_];
				++c->tries;
				c->score = gres[ti].score, c->n_cigar = gres[ti].n_cigar, c->final_task = ti;
				if (c->score == c->last_sc) break;                                                     /* bwamem.c:1198 */
				c->last_sc = c->score;
				if (single || !(c->tries < 3 && c->score < reqs[k].truesc - p->a)) break;              /* bwamem.c:1201 */
			}
			if ((uint32_t)c->n_cigar <= tasks[c->final_task].cigar_cap)
				c->cig = scratch + tasks[c->final_task].cigar_off;
			else owner[n_big++] = k;
		}
		if (n_big > 0) { /* the few long CIGARs again, with the full ql+tl+2 slots */
			bmh_glb_task_t *bt = (bmh_glb_task_t *)malloc(sizeof(*bt) * (size_t)n_big);
			bmh_glb_result_t *br = (bmh_glb_result_t *)malloc(sizeof(*br) * (size_t)n_big);
			size_t bs = 0;
			if (!bt || !br) { free(bt), free(br); rc = BMH_E_NOMEM; goto done; }
			for (k = 0; k < n_big; ++k) {
				const cg_t *c = &cg[owner[k]];
				bt[k] = tasks[c->final_task];
				bt[k].cigar_off = (uint32_t)bs, bt[k].cigar_cap = (uint32_t)(c->ql + c->tl + 2);
				bs += bt[k].cigar_cap;
			}
			bigscr = (uint32_t *)malloc(4 * (bs + 8));
			rc = bigscr ? bmh_global_batch(ctx, 0, 0, bt, n_big, br, bigscr, bs + 4) : BMH_E_NOMEM;
			for (k = 0; k < n_big && !rc; ++k) {
				cg_t *c = &cg[owner[k]];
				c->score = br[k].score, c->n_cigar = br[k].n_cigar;
				c->cig = bigscr + bt[k].cigar_off;
			}
			free(bt), free(br);
			if (rc) goto done;
		}
	}

	if (trace) tt[3] = now_s();
	for (k = 0; k < n_req; ++k) { /* NM / MD and the packed outputs */
		cg_t *c = &cg[k];
		size_t md_len = 0;
		if (!c->valid) continue;
		if (cig_used + (size_t)c->n_cigar > cigar_words || md_used + 3 * ((size_t)c->ql + (size_t)c->tl) + 16 > md_bytes) {
			rc = BMH_E_CIGAR_CAP;
			goto done;
		}
		if (c->cig) memcpy(cigar_pool + cig_used, c->cig, 4 * (size_t)c->n_cigar);
		else cigar_pool[cig_used] = (uint32_t)c->ql << 4;
		res[k].NM = nm_md(c->n_cigar, cigar_pool + cig_used, pool + c->q_off, pool + c->t_off, c->rev, md_pool + md_used, &md_len);
		res[k].score = c->score, res[k].n_cigar = c->n_cigar, res[k].tries = c->tries;
		res[k].cigar_off = (uint32_t)cig_used, res[k].md_off = (uint32_t)md_used, res[k].md_len = (uint32_t)md_len;
		cig_used += (size_t)c->n_cigar, md_used += md_len + 1;
	}
	if (trace)
		fprintf(stderr, "[bwamem_hip] bmh_reg2cigar_batch %lld regions: oriented copies %.1f ms, tries %.1f ms (of which upload + GPU calls %.1f ms), NM/MD %.1f ms\n",
		        (long long)n_req, (tt[1] - tt[0]) * 1e3, (tt[3] - tt[1]) * 1e3, tt[2] * 1e3, (now_s() - tt[3]) * 1e3);
done:
	*cig_used_ = cig_used, *md_used_ = md_used;
	free(cg), free(pool), free(tasks), free(gres), free(owner), free(scratch), free(bigscr);
	return rc;
}

/* ---- the path with the region record: everything that reads a sequence byte runs on the device */
enum { MD_SLOT = 96 }; /* bytes of MD coming back per region: "150", "75A74", ... (a 150 bp mate with 12 % substitutions: ~55); the rare longer one is redone by the host path */

int bmh_ctx_has_pac_(const bmh_ctx_t *ctx, const uint8_t *pac, int64_t l_pac);

static int reg2cigar_region_records(bmh_ctx_t *ctx, const bmh_params_t *p, int64_t l_pac, const uint8_t *pac, const bmh_read_t *reads,
                                    int64_t n_req, const bmh_cigar_req_t *reqs, bmh_cigar_res_t *res, uint32_t *cigar_pool,
                                    size_t cigar_words, char *md_pool, size_t md_bytes)
{
	bmh_region_req_t *rq = 0;
	bmh_region_res_t *rr = 0;
	bmh_glb_task_t *tasks = 0;
	bmh_cigar_req_t *redo_req = 0;
	bmh_cigar_res_t *redo_res = 0;
	int64_t *of = 0, *redo_of = 0; /* request index of region record v / of redone request j */
	uint8_t *rpool = 0;
	uint32_t *cout = 0;
	char *mout = 0;
	int64_t k, v, n_v = 0, n_tasks = 0, task_cap, n_redo = 0;
	size_t rbytes = 0, obytes = 0, slot = 0, cig_used = 0, md_used = 0;
	int rc = BMH_OK;
	const int trace = getenv("BMH_DRIVER_TRACE") != 0;
	double tt[4] = {0, 0, 0, 0};

	if (trace) tt[0] = now_s();
	rq = (bmh_region_req_t *)malloc(sizeof(*rq) * (size_t)n_req);
	rr = (bmh_region_res_t *)malloc(sizeof(*rr) * (size_t)n_req);
	of = (int64_t *)malloc(sizeof(*of) * (size_t)n_req);
	task_cap = n_req + n_req / 4 + 64;
	tasks = (bmh_glb_task_t *)malloc(sizeof(*tasks) * (size_t)task_cap);
	if (!rq || !rr || !of || !tasks) { rc = BMH_E_NOMEM; goto done; }
	for (k = 0; k < n_req; ++k) { /* records, pool layout, bands, tasks: arithmetic on coordinates only */
		const bmh_cigar_req_t *r = &reqs[k];
		const int ql = r->qe - r->qb, single = r->truesc == INT32_MIN;
		const int64_t tl = r->re - r->rb;
		bmh_region_req_t *q;
		int t_, w2, prev_w = -1;
		memset(&res[k], 0, sizeof(res[k]));
		res[k].NM = -1;
		if (ql <= 0 || r->rb >= r->re || (r->rb < l_pac && r->re > l_pac) || r->rb < 0 || r->re > l_pac << 1) continue; /* bwa.c:99-101 */
		if (ql > 65535 || tl > 65535) { rc = BMH_E_RANGE; goto done; }
		q = &rq[n_v];
		of[n_v++] = k;
		q->q_src = rbytes, rbytes += (size_t)ql;
		q->o_off = obytes, obytes += (size_t)ql + (size_t)tl;
		q->rb = r->rb, q->ql = ql, q->tl = (int)tl, q->truesc = r->truesc;
		q->task[0] = q->task[1] = q->task[2] = -1;
		w2 = first_band(p, r, ql, (int)tl);
		if (ql == tl && w2 == 0) continue; /* the no-gap case: bwa.c:108-114 */
		for (t_ = 0; t_ < (single ? 1 : 3); ++t_) {
			const int w = try_band(p, ql, (int)tl, w2 << t_);
			bmh_glb_task_t *t;
			if (w == prev_w) { q->task[t_] = q->task[t_ - 1]; continue; } /* the band saturates: the same alignment */
			prev_w = w;
			if (n_tasks == task_cap) {
				task_cap = task_cap + task_cap / 2 + 1024;
				tasks = (bmh_glb_task_t *)realloc(tasks, sizeof(*tasks) * (size_t)task_cap);
				if (!tasks) { rc = BMH_E_NOMEM; goto done; }
			}
			t = &tasks[n_tasks];
			t->q_off = q->o_off, t->t_off = q->o_off + (uint64_t)ql, t->qlen = (uint16_t)ql, t->tlen = (uint16_t)tl, t->w = w;
			t->cigar_off = (uint32_t)slot, t->cigar_cap = (uint32_t)(ql + (int)tl + 2 < SMALL_CAP ? ql + (int)tl + 2 : SMALL_CAP);
			slot += t->cigar_cap;
			q->task[t_] = (int32_t)n_tasks++;
		}
	}
	rpool = (uint8_t *)malloc(rbytes + 16);
	cout = (uint32_t *)malloc(4 * ((size_t)n_v * SMALL_CAP + 4));
	mout = (char *)malloc((size_t)n_v * MD_SLOT + 4);
	if (!rpool || !cout || !mout) { rc = BMH_E_NOMEM; goto done; }
	for (v = 0; v < n_v; ++v) { /* the query windows as the reads hold them */
		const bmh_cigar_req_t *r = &reqs[of[v]];
		if (v + 8 < n_v) { /* (the reads are scattered allocations of the host program's) */
			const bmh_cigar_req_t *nx = &reqs[of[v + 8]];
			const uint8_t *ns = reads[nx->read].seq + nx->qb;
			__builtin_prefetch(ns), __builtin_prefetch(ns + 64), __builtin_prefetch(ns + 128);
		}
		memcpy(rpool + rq[v].q_src, reads[r->read].seq + r->qb, (size_t)rq[v].ql);
	}
	if (trace) tt[1] = now_s();
	if (n_v > 0 && (rc = bmh_region_cigar_batch(ctx, rpool, rbytes, obytes, rq, n_v, tasks, n_tasks, slot + 4, SMALL_CAP, MD_SLOT, rr, cout, mout)))
		goto done;
	if (trace) tt[2] = now_s();
	for (v = 0; v < n_v; ++v) { /* the packed outputs */
		const bmh_region_res_t *x = &rr[v];
		bmh_cigar_res_t *o = &res[of[v]];
		if (x->flags) { ++n_redo; continue; }
		if (cig_used + (size_t)x->n_cigar > cigar_words || md_used + (size_t)x->md_len + 1 > md_bytes) { rc = BMH_E_CIGAR_CAP; goto done; }
		memcpy(cigar_pool + cig_used, cout + (size_t)v * SMALL_CAP, 4 * (size_t)x->n_cigar);
		memcpy(md_pool + md_used, mout + (size_t)v * MD_SLOT, (size_t)x->md_len);
		md_pool[md_used + (size_t)x->md_len] = 0;
		o->score = x->score, o->n_cigar = x->n_cigar, o->NM = x->NM, o->tries = x->tries;
		o->cigar_off = (uint32_t)cig_used, o->md_off = (uint32_t)md_used, o->md_len = (uint32_t)x->md_len;
		cig_used += (size_t)x->n_cigar, md_used += (size_t)x->md_len + 1;
	}
	if (n_redo > 0) { /* long CIGARs / MD strings: the few regions again, through the path with host copies and full capacities */
		int64_t j = 0;
		redo_req = (bmh_cigar_req_t *)malloc(sizeof(*redo_req) * (size_t)n_redo);
		redo_res = (bmh_cigar_res_t *)malloc(sizeof(*redo_res) * (size_t)n_redo);
		redo_of = (int64_t *)malloc(sizeof(*redo_of) * (size_t)n_redo);
		if (!redo_req || !redo_res || !redo_of) { rc = BMH_E_NOMEM; goto done; }
		for (v = 0; v < n_v; ++v)
			if (rr[v].flags) redo_req[j] = reqs[of[v]], redo_of[j++] = of[v];
		if ((rc = reg2cigar_host_copies(ctx, p, l_pac, pac, reads, n_redo, redo_req, redo_res, cigar_pool, cigar_words, &cig_used, md_pool,
		                                md_bytes, &md_used)))
			goto done;
		for (j = 0; j < n_redo; ++j) res[redo_of[j]] = redo_res[j];
	}
	if (trace)
		fprintf(stderr, "[bwamem_hip] bmh_reg2cigar_batch %lld regions, region records: layout + query windows %.1f ms, upload + GPU call %.1f ms, packing %.1f ms (%lld tasks, %lld regions redone with host copies)\n",
		        (long long)n_req, (tt[1] - tt[0]) * 1e3, (tt[2] - tt[1]) * 1e3, (now_s() - tt[2]) * 1e3, (long long)n_tasks, (long long)n_redo);
done:
	free(rq), free(rr), free(of), free(tasks), free(rpool), free(cout), free(mout), free(redo_req), free(redo_res), free(redo_of);
	return rc;
}

int bmh_reg2cigar_batch(bmh_ctx_t *ctx, int64_t l_pac, const uint8_t *pac, const bmh_read_t *reads, int64_t n_req,
                        const bmh_cigar_req_t *reqs, bmh_cigar_res_t *res, uint32_t *cigar_pool, size_t cigar_words,
                        char *md_pool, size_t md_bytes)
{
	static int host_only = -1; /* BMH_REG2CIGAR_HOST=1: the path with host copies whatever the context holds (A/B, tests) */
	const bmh_params_t *p;
	size_t cig_used = 0, md_used = 0;
	if (!ctx || n_req < 0 || (n_req > 0 && (!reads || !reqs || !res || !pac))) return BMH_E_ARG;
	p = bmh_ctx_params_(ctx);
	if (!p) return BMH_E_ARG;
	if (n_req == 0) return BMH_OK;
	if (host_only < 0) {
		const char *e = getenv("BMH_REG2CIGAR_HOST");
		host_only = e && *e && *e != '0';
	}
	if (!host_only && bmh_ctx_has_pac_(ctx, pac, l_pac))
		return reg2cigar_region_records(ctx, p, l_pac, pac, reads, n_req, reqs, res, cigar_pool, cigar_words, md_pool, md_bytes);
	return reg2cigar_host_copies(ctx, p, l_pac, pac, reads, n_req, reqs, res, cigar_pool, cigar_words, &cig_used, md_pool, md_bytes, &md_used);
}
