/*
 * interpose.c -- the reference-side binding, as an LD_PRELOAD-able object.
 *
 * It provides mem_align1_core_batched() with the fork's exact signature
 * (reference bwa-0.7.8/bwamem.c:1086) so that, preloaded in front of a build of the
 * reference (oracle/_ref/bwa + libbwa_ref.so), phase 1 of mem_process_seqs
 * (bwamem.c:1313 -> worker1_batched :1264) runs seeding/chaining on the CPU exactly
 * as before and hands every batch's chains to bmh_chain2aln_batch() -- the hook the
 * fork left commented out at bwamem.c:1110.  Run `bwa mem -b <batch>` to choose the
 * batch size.  INTEGRATION.md shows the same code as a patch to bwamem.c.
 *
 * Everything declared `extern` below is the reference's own symbol, resolved at load
 * time from libbwa_ref.so; nothing of the reference is compiled into this library.
 * The struct mirrors are layout-compatible re-declarations (file:line cited).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/bwamem_hip.h"
#include "tls_ctx.h"

typedef struct { /* mem_opt_t of the fork, bwamem.h:21-48 */
	int a, b, o_del, e_del, o_ins, e_ins, pen_unpaired, pen_clip5, pen_clip3, w, zdrop;
	int T, flag, min_seed_len;
	float split_factor;
	int split_width, max_occ, max_chain_gap, n_threads, batch_size, chunk_size;
	float mask_level, chain_drop_ratio, mask_level_redun, mapQ_coef_len;
	int mapQ_coef_fac, max_ins, max_matesw;
	int8_t mat[25];
} ref_mem_opt_t;
#define REF_MEM_F_NO_EXACT 0x40 /* bwamem.h:19 */

typedef struct { int64_t l_pac; /* first field of bntseq_t, bntseq.h:53 */ } ref_bntseq_head_t;
typedef struct { int l_seq; char *name, *comment, *seq, *qual, *sam; } ref_bseq1_t; /* bwa.h:18-22 */

/* reference functions this shim keeps calling on the CPU (bwamem.c:283,319,395,438,495; bntseq.c) */
extern bmh_chain_v mem_chain(const void *opt, const void *bwt, int64_t l_pac, int len, const uint8_t *seq);
extern int mem_chain_flt(const void *opt, int n_chn, bmh_chain_t *chains);
extern int mem_chain2aln_short(const void *opt, int64_t l_pac, const uint8_t *pac, int l_query, const uint8_t *query,
                               const bmh_chain_t *c, bmh_alnreg_v *av);
extern int mem_sort_and_dedup(int n, bmh_alnreg_t *a, float mask_level_redun);
extern int mem_test_and_remove_exact(const void *opt, int n, bmh_alnreg_t *a, int qlen);
extern unsigned char nst_nt4_table[256];

typedef struct {
	const ref_mem_opt_t *opt;
	int64_t l_pac;
	const uint8_t *pac;
	const bmh_read_t *reads;
	const bmh_chain_v *chains;
} pre_ud_t;

static int pre_short(void *user, int r, int ci, bmh_alnreg_v *av) /* bwamem.c:1104 */
{
	const pre_ud_t *u = (const pre_ud_t *)user;
	return mem_chain2aln_short(u->opt, u->l_pac, u->pac, u->reads[r].l_seq, u->reads[r].seq, &u->chains[r].a[ci], av);
}

bmh_alnreg_v *mem_align1_core_batched(const ref_mem_opt_t *opt, const void *bwt, const ref_bntseq_head_t *bns,
                                      const uint8_t *pac, ref_bseq1_t *seqs, int start, int batch_size)
{
	bmh_chain_v *chn = (bmh_chain_v *)malloc(sizeof(bmh_chain_v) * (size_t)batch_size);
	bmh_alnreg_v *regs = (bmh_alnreg_v *)calloc((size_t)batch_size, sizeof(bmh_alnreg_v));
	bmh_read_t *reads = (bmh_read_t *)malloc(sizeof(bmh_read_t) * (size_t)batch_size);
	bmh_params_t p;
	bmh_ctx_t *ctx;
	pre_ud_t ud;
	int b, i, rc;

	for (b = 0; b < batch_size; ++b) { /* CPU stages before the path, unchanged: bwamem.c:1093-1097 */
		ref_bseq1_t *s = &seqs[start + b];
		for (i = 0; i < s->l_seq; ++i) s->seq[i] = s->seq[i] < 4 ? s->seq[i] : (char)nst_nt4_table[(int)s->seq[i]];
		chn[b] = mem_chain(opt, bwt, bns->l_pac, s->l_seq, (uint8_t *)s->seq);
		chn[b].n = (size_t)mem_chain_flt(opt, (int)chn[b].n, chn[b].a);
		reads[b].l_seq = s->l_seq, reads[b].seq = (const uint8_t *)s->seq;
	}

	memset(&p, 0, sizeof(p)); /* the hot-path fields of mem_opt_t */
	p.o_del = opt->o_del, p.e_del = opt->e_del, p.o_ins = opt->o_ins, p.e_ins = opt->e_ins, p.zdrop = opt->zdrop;
	p.a = opt->a, p.w = opt->w, p.pen_clip5 = opt->pen_clip5, p.pen_clip3 = opt->pen_clip3;
	memcpy(p.mat, opt->mat, 25);
	ctx = bmh_tls_ctx(&p);
	{ /* reference resident in HBM, shared by all threads: the kernels do bns_get_seq themselves.  BMH_PAC_RESIDENT=0
	   * falls back to host-decoded windows in the pool. */
		const char *e = getenv("BMH_PAC_RESIDENT");
		if (!(e && e[0] == '0') && (rc = bmh_ctx_set_pac(ctx, pac, bns->l_pac))) bmh_tls_die(bmh_last_error(ctx), rc);
	}
	ud.opt = opt, ud.l_pac = bns->l_pac, ud.pac = pac, ud.reads = reads, ud.chains = chn;
	if ((rc = bmh_chain2aln_batch(ctx, bns->l_pac, pac, batch_size, reads, chn, pre_short, &ud, regs))) /* bwamem.c:1110 */
		bmh_tls_die(bmh_last_error(ctx), rc);

	for (b = 0; b < batch_size; ++b) { /* CPU stages after the path, unchanged: bwamem.c:1106,1112-1117 */
		for (i = 0; i < (int)chn[b].n; ++i) free(chn[b].a[i].seeds);
		free(chn[b].a);
		regs[b].n = (size_t)mem_sort_and_dedup((int)regs[b].n, regs[b].a, opt->mask_level_redun);
		if (opt->flag & REF_MEM_F_NO_EXACT)
			regs[b].n = (size_t)mem_test_and_remove_exact(opt, (int)regs[b].n, regs[b].a, seqs[start + b].l_seq);
	}
	free(chn);
	free(reads);
	return regs; /* caller copies and frees, bwamem.c:1272-1278 */
}

/* =====================================================================================================================
 * mem_process_seqs() with the reference's exact signature (bwamem.h:117, bwamem.c:1297-1327), for PAIRED-END runs:
 * the same three steps as the reference -- phase 1 through the batching seam above, insert-size statistics, phase 2 --
 * with ONE addition between them: mate rescue for the whole chunk in one bmh_matesw_batch() call, after which phase 2
 * runs with MEM_F_NO_RESCUE so that mem_sam_pe skips its own per-pair rescue block (bwamem_pair.c:251-263) and goes on
 * with the vectors the batch left -- which are, element for element, what that block would have produced.
 * Single-end runs and BMH_MATESW_BATCH=0 are forwarded to the reference's own mem_process_seqs.
 */

#define REF_MEM_F_PE 0x2         /* bwamem.h:14 */
#define REF_MEM_F_NO_RESCUE 0x20 /* bwamem.h:18 */

extern void kt_for(int n_threads, void (*func)(void *, int, int), void *data, int n);                       /* kthread.c */
extern void kt_for_batch(int n_threads, void (*func)(void *, int, int, int), void *data, int n, int batch);  /* kthread_batch.c:44 */
extern void mem_pestat(const void *opt, int64_t l_pac, int n, const bmh_alnreg_v *regs, bmh_pestat_t pes[4]); /* bwamem_pair.c:46 */
extern int mem_sam_pe(const void *opt, const void *bns, const uint8_t *pac, const bmh_pestat_t pes[4], uint64_t id,
                      ref_bseq1_t s[2], bmh_alnreg_v a[2]);                                                 /* bwamem_pair.c:238 */
extern double cputime(void), realtime(void); /* utils.c */
extern int bwa_verbose;

typedef struct {
	const ref_mem_opt_t *opt;
	const void *bwt;
	const ref_bntseq_head_t *bns;
	const uint8_t *pac;
	const bmh_pestat_t *pes;
	ref_bseq1_t *seqs;
	bmh_alnreg_v *regs;
	int64_t n_processed;
} qa_worker_t;

static void qa_worker1_batched(void *data, int start, int batch_size, int tid) /* == worker1_batched, bwamem.c:1264-1279 */
{
	qa_worker_t *w = (qa_worker_t *)data;
	bmh_alnreg_v *ret = mem_align1_core_batched(w->opt, w->bwt, w->bns, w->pac, w->seqs, start, batch_size);
	int i;
	(void)tid;
	for (i = start; i < start + batch_size; ++i) w->regs[i] = ret[i - start];
	free(ret);
}

static void qa_worker2_pe(void *data, int i, int tid) /* == the PE branch of worker2, bwamem.c:1290-1294 */
{
	qa_worker_t *w = (qa_worker_t *)data;
	(void)tid;
	mem_sam_pe(w->opt, w->bns, w->pac, w->pes, (uint64_t)(w->n_processed >> 1) + (uint64_t)i, &w->seqs[i << 1], &w->regs[i << 1]);
	free(w->regs[i << 1 | 0].a), free(w->regs[i << 1 | 1].a);
}

static int qa_dedup(void *user, int n, bmh_alnreg_t *a) /* bmh_dedup_fn over the reference's own function */
{
	return mem_sort_and_dedup(n, a, ((const ref_mem_opt_t *)user)->mask_level_redun);
}

typedef void (*process_seqs_fn)(const ref_mem_opt_t *, const void *, const ref_bntseq_head_t *, const uint8_t *, int64_t, int,
                                ref_bseq1_t *, const bmh_pestat_t *);

void mem_process_seqs(const ref_mem_opt_t *opt, const void *bwt, const ref_bntseq_head_t *bns, const uint8_t *pac,
                      int64_t n_processed, int n, ref_bseq1_t *seqs, const bmh_pestat_t *pes0)
{
	const char *e = getenv("BMH_MATESW_BATCH");
	qa_worker_t w;
	bmh_pestat_t pes[4];
	ref_mem_opt_t opt2;
	double ctime, rtime;
	if (!(opt->flag & REF_MEM_F_PE) || (opt->flag & REF_MEM_F_NO_RESCUE) || (e && e[0] == '0')) {
		static process_seqs_fn next;
		if (!next) next = (process_seqs_fn)dlsym(RTLD_NEXT, "mem_process_seqs");
		if (!next) bmh_tls_die("no other mem_process_seqs is loaded", BMH_E_ARG);
		next(opt, bwt, bns, pac, n_processed, n, seqs, pes0);
		return;
	}
	ctime = cputime(), rtime = realtime();
	w.opt = opt, w.bwt = bwt, w.bns = bns, w.pac = pac, w.seqs = seqs, w.n_processed = n_processed, w.pes = pes;
	w.regs = (bmh_alnreg_v *)malloc((size_t)n * sizeof(bmh_alnreg_v));
	kt_for_batch(opt->n_threads, qa_worker1_batched, &w, n, opt->batch_size); /* bwamem.c:1313 */
	if (pes0) memcpy(pes, pes0, 4 * sizeof(bmh_pestat_t));                    /* bwamem.c:1314-1317 */
	else mem_pestat(opt, bns->l_pac, n, w.regs, pes);
	{ /* the whole chunk's mate rescue in one call; reads are base codes by now (bwamem.c:1093-1094) */
		bmh_params_t p;
		bmh_matesw_opt_t mo;
		bmh_read_t *reads = (bmh_read_t *)malloc(sizeof(bmh_read_t) * (size_t)n);
		bmh_ctx_t *ctx;
		int i, rc;
		for (i = 0; i < n; ++i) reads[i].l_seq = seqs[i].l_seq, reads[i].seq = (const uint8_t *)seqs[i].seq;
		memset(&p, 0, sizeof(p));
		p.o_del = opt->o_del, p.e_del = opt->e_del, p.o_ins = opt->o_ins, p.e_ins = opt->e_ins, p.zdrop = opt->zdrop;
		p.a = opt->a, p.w = opt->w, p.pen_clip5 = opt->pen_clip5, p.pen_clip3 = opt->pen_clip3;
		memcpy(p.mat, opt->mat, 25);
		ctx = bmh_tls_ctx(&p);
		{
			const char *pr = getenv("BMH_PAC_RESIDENT");
			if (!(pr && pr[0] == '0') && (rc = bmh_ctx_set_pac(ctx, pac, bns->l_pac))) bmh_tls_die(bmh_last_error(ctx), rc);
		}
		mo.pen_unpaired = opt->pen_unpaired, mo.max_matesw = opt->max_matesw, mo.min_seed_len = opt->min_seed_len, mo.rsv = 0;
		if ((rc = bmh_matesw_batch(ctx, bns->l_pac, pac, n >> 1, reads, w.regs, pes, &mo, qa_dedup, (void *)opt, 0)))
			bmh_tls_die(bmh_last_error(ctx), rc);
		if (getenv("BMH_VERBOSE")) {
			bmh_driver_stats_t st;
			bmh_driver_stats(ctx, &st);
			fprintf(stderr, "[bwamem_hip] mate rescue: %d pairs, %lld ksw_align2 calls in %lld GPU rounds, %lld pool bytes\n", n >> 1,
			        (long long)st.ext_tasks, (long long)st.rounds, (long long)st.pool_bytes);
		}
		free(reads);
	}
	opt2 = *opt, opt2.flag |= REF_MEM_F_NO_RESCUE, w.opt = &opt2;
	kt_for(opt->n_threads, qa_worker2_pe, &w, n >> 1); /* bwamem.c:1319 */
	free(w.regs);
	if (bwa_verbose >= 3)
		fprintf(stderr, "[M::%s] Processed %d reads in %.3f CPU sec, %.3f real sec\n", __func__, n, cputime() - ctime, realtime() - rtime);
}
