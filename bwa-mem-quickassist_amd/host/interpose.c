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
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <malloc.h>

#include <pthread.h>
#include <semaphore.h>

#include "../../include/bwamem_hip.h"
#include "tls_ctx.h"

/* At most this many host threads are inside a GPU batch call at a time ($BMH_GPU_CONCURRENCY, default 12): the calls of
 * more threads than that only queue up behind one another on the device, and their workspace (re)allocations, which
 * synchronise the whole device, collide.  The host work around the calls (chaining, folding) is not limited. */
static sem_t g_gpu_sem;
static pthread_once_t g_gpu_once = PTHREAD_ONCE_INIT;
static int gpu_concurrency(void)
{
	const char *e = getenv("BMH_GPU_CONCURRENCY");
	return e && atoi(e) > 0 ? atoi(e) : 12;
}
static void gpu_sem_init(void) { sem_init(&g_gpu_sem, 0, (unsigned)gpu_concurrency()); }
static void gpu_enter(void)
{
	pthread_once(&g_gpu_once, gpu_sem_init);
	while (sem_wait(&g_gpu_sem) != 0) {}
}
static void gpu_leave(void) { sem_post(&g_gpu_sem); }

/* When the shim is loaded: create the contexts the run will use in the background, while the host program parses its
 * arguments and loads its index ($BMH_PREWARM=0 turns this off, =N creates N; default: the host program's -t). */
static int g_prewarm_n = 16; /* contexts to create ahead: one per host thread of the widest phase (a thread holds one while it is inside a driver call) */
/* The first chunk's host code runs on heaps that have never been touched: every page of the drivers' work arrays is a
 * page fault (chaining 1.05 thread-seconds in the first chunk of a run against 0.11 in the next ones).  While the host
 * program still parses its input, one short-lived thread per worker touches $BMH_HEAP_WARM_MB (64) megabytes through
 * malloc and frees them again: glibc keeps a finished thread's arena, mapped and resident (M_TRIM_THRESHOLD above), and
 * hands it to the next new thread -- the workers of kt_for. */
static void *heap_warm_thread(void *arg)
{
	const int mb = (int)(intptr_t)arg;
	void **p = (void **)malloc(sizeof(void *) * (size_t)(mb > 0 ? mb : 1));
	int k;
	if (!p) return 0;
	for (k = 0; k < mb; ++k)
		if ((p[k] = malloc(1 << 20)) != 0) memset(p[k], 0, 1 << 20);
	for (k = 0; k < mb; ++k) free(p[k]);
	free(p);
	return 0;
}
static void heap_warm(int threads)
{
	const char *e = getenv("BMH_HEAP_WARM_MB");
	const int mb = e ? atoi(e) : 64;
	int k;
	for (k = 0; k < threads && mb > 0; ++k) {
		pthread_t t;
		if (pthread_create(&t, 0, heap_warm_thread, (void *)(intptr_t)mb) == 0) pthread_detach(t);
	}
}

extern double cputime(void), realtime(void); /* the host program's (utils.c) */
static double g_t_loaded; /* realtime() when the shim was loaded */
static void qa_shim_exit(void);
static void *prewarm_thread(void *arg)
{
	int ndev = 0;
	(void)arg;
	/* The HIP runtime registers its own exit handlers when it is first initialised -- here, in this thread.  Exit handlers run
	 * last-registered-first, so the stop-and-join handler is registered AFTER that first HIP call: it then runs BEFORE the
	 * runtime's teardown whatever the order of the two libraries' constructors was.  (It is idempotent; the constructor
	 * registers it too, for a run that ends before this line.) */
	bmh_device_count(&ndev);
	atexit(qa_shim_exit);
	bmh_pool_prewarm(g_prewarm_n); /* (several threads creating contexts side by side are no faster: measured) */
	if (getenv("BMH_VERBOSE")) fprintf(stderr, "[bwamem_hip] %d contexts ready %.3f s after the shim was loaded\n", g_prewarm_n, realtime() - g_t_loaded);
	return 0;
}
/* A run shorter than the pre-warming (a few hundred reads) must not reach the runtime's teardown with that thread still
 * inside a HIP call: it is told to stop and waited for. */
static pthread_t g_prewarm;
static volatile int g_prewarm_on;
static void qa_shim_exit(void)
{
	if (!__sync_bool_compare_and_swap(&g_prewarm_on, 1, 0)) return; /* once, whichever registration fires first */
	bmh_pool_stop();
	pthread_join(g_prewarm, 0);
}
__attribute__((constructor)) static void qa_shim_loaded(void)
{
	const char *e = getenv("BMH_PREWARM"), *pl = getenv("LD_PRELOAD");
	pthread_t t;
	/* The HIP runtime multiplexes all streams of a process onto 4 hardware queues by default; with many host threads inside
	 * batch calls a thread's small extension kernels then wait behind another thread's seeding or Smith-Waterman kernel
	 * of milliseconds that happens to share its queue (measured, steady-state chunk of 1.07 M reads at 16 threads: 0.48 s
	 * with 4 queues, 0.35 s with 8, 0.28 s with 16 and 12 threads admitted to the GPU at a time; 24 is slower again).
	 * Must be in the environment before the runtime initialises; a value the user has set is left alone. */
	setenv("GPU_MAX_HW_QUEUES", "16", 0);
	bmh_set_device_gate(gpu_enter, gpu_leave); /* the library holds a GPU place for its device sections only */
	/* The drivers allocate and free a few buffers of megabytes per batch on every thread: keep that memory in the heaps
	 * instead of handing it back to the kernel and faulting it in again each time (mprotect/munmap/page faults were ~8 % of
	 * the CPU time of a chunk). */
	{
		const char *m = getenv("BMH_MALLOPT");
		const int bits = m ? atoi(m) : 3; /* (a larger M_TOP_PAD, bit 4, measured much slower) */
		if (bits & 1) mallopt(M_TRIM_THRESHOLD, 1 << 30);
		if (bits & 2) mallopt(M_MMAP_THRESHOLD, 32 << 20);
		if (bits & 4) mallopt(M_TOP_PAD, 16 << 20);
	}
	{ /* as many host threads as cores drive the library here: a thread waiting for the GPU sleeps instead of spinning
	   * (BMH_WAIT=spin restores the library's default) */
		const char *w = getenv("BMH_WAIT");
		bmh_set_wait_mode(!(w && !strcmp(w, "spin")));
	}
	if (e && e[0] == '0') return;
	if (!pl || !strstr(pl, "libbwamem_hip_dropin")) return; /* only when preloaded into a host program, not when merely dlopen()ed */
	{ /* ... and only into `<prog> mem ...`: index building, usage errors etc. never touch the GPU */
		char buf[512];
		FILE *f = fopen("/proc/self/cmdline", "rb");
		size_t n = f ? fread(buf, 1, sizeof(buf) - 1, f) : 0, i, first_len;
		int is_mem = 0;
		if (f) fclose(f);
		buf[n] = 0;
		first_len = strlen(buf);
		for (i = first_len + 1; i < n; i += strlen(buf + i) + 1)
			if (!strcmp(buf + i, "mem")) { is_mem = 1; break; }
		if (!is_mem) return;
		for (; i < n; i += strlen(buf + i) + 1) /* its -t, as `-t N` or `-tN` */
			if (buf[i] == '-' && buf[i + 1] == 't') {
				const char *v = buf[i + 2] ? buf + i + 2 : (i + 3 < n ? buf + i + 3 : "");
				if (atoi(v) > 0) g_prewarm_n = atoi(v);
			}
		{ /* phase 2 runs -t * 3/2 threads; $BMH_P1_THREADS / $BMH_P2_THREADS override the phases' thread counts; $BMH_PREWARM=N: N */
			const char *v[3] = {getenv("BMH_P1_THREADS"), getenv("BMH_P2_THREADS"), e};
			int k;
			g_prewarm_n += g_prewarm_n / 2;
			for (k = 0; k < 2; ++k)
				if (v[k] && atoi(v[k]) > g_prewarm_n) g_prewarm_n = atoi(v[k]);
			if (v[2] && atoi(v[2]) > 0) g_prewarm_n = atoi(v[2]);
			if (g_prewarm_n > 128) g_prewarm_n = 128;
		}
	}
	g_t_loaded = realtime();
	heap_warm(g_prewarm_n);
	if (pthread_create(&t, 0, prewarm_thread, 0) == 0) {
		g_prewarm = t, g_prewarm_on = 1;
		atexit(qa_shim_exit); /* a run that ends before the pre-warming thread's first HIP call; see prewarm_thread for the other case */
	}
}

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

/* what this shim still takes from the host program: its base-code table and its clocks (bntseq.c, utils.c) */
extern unsigned char nst_nt4_table[256];
static double stage_now(void) /* the clock of the BMH_VERBOSE thread-second sums: wall time, or with BMH_TRACE_CPU the thread's CPU time */
{
	static int cpu = -1;
	struct timespec ts;
	if (cpu < 0) cpu = getenv("BMH_TRACE_CPU") != 0;
	if (!cpu) return realtime();
	clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
	return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ---- seeding: the batch's FM-index queries on the GPU, chaining on top of them -----------------------------------------
 * mem_chain (bwamem.c:283) = SMEM search (smem_next2 -> bwt_smem1) + suffix-array look-ups (bwt_sa) + chaining in a
 * B-tree.  The first two are pure functions of (index, read): bmh_smem_batch and bmh_sa_batch compute them for the whole
 * batch; bmh_chain_reads (host/chain_batch.c) then does what mem_chain and mem_chain_flt do with them. */
typedef struct { /* bwt_t, bwt.h:45-57 */
	uint64_t primary, L2[5], seq_len, bwt_size;
	uint32_t *bwt;
	uint32_t cnt_table[256];
	int sa_intv;
	uint64_t n_sa;
	uint64_t *sa;
} ref_bwt_t;

typedef struct {
	const ref_bwt_t *bwt;
	int n_reads;
	const bmh_read_t *reads;
	uint32_t *call_off;
	bmh_smem_call_t *calls;
	uint64_t *intv_off;
	bmh_smem_intv_t *intv;
	uint64_t *sa_k, *sa_pos; /* per interval: where its run of positions starts (bmh_chain_sa_keys); the positions */
	size_t n_sa;
} qa_seed_t;
static double g_seed_density[3] = {0.06, 0.02, 0.03}; /* calls, intervals, suffix-array positions per base */
static long long g_seed_us[3]; /* thread-microseconds: bmh_smem_batch, building the look-up keys, bmh_sa_batch */
static __thread qa_seed_t *qa_seed; /* the batch this thread is chaining */

static void qa_seed_batch_begin(bmh_ctx_t *ctx, const ref_mem_opt_t *opt, const ref_bwt_t *bwt, int n, const bmh_read_t *reads)
{
	bmh_bwt_t ib;
	bmh_smem_opt_t so;
	qa_seed_t *S;
	size_t tot = 0, call_cap, intv_cap, nk = 0;
	uint64_t *keys = 0;
	double ts[4];
	int r, rc, i;
	qa_seed = 0;
	ib.primary = bwt->primary, ib.seq_len = bwt->seq_len, ib.bwt_size = bwt->bwt_size, ib.bwt = bwt->bwt;
	for (i = 0; i < 5; ++i) ib.L2[i] = bwt->L2[i];
	ib.sa_intv = bwt->sa_intv, ib.n_sa = bwt->n_sa, ib.sa = bwt->sa;
	if ((rc = bmh_ctx_set_bwt(ctx, &ib))) bmh_tls_die(bmh_last_error(ctx), rc);
	so.min_seed_len = opt->min_seed_len, so.split_len = (int)(opt->min_seed_len * opt->split_factor + .499); /* bwamem.c:211 */
	so.split_width = opt->split_width, so.start_width = (opt->flag & REF_MEM_F_NO_EXACT) ? 2 : 1;           /* bwamem.c:212 */
	so.min_emit_len = opt->min_seed_len; /* shorter intervals are never turned into seeds (bwamem.c:219): they stay on the device */
	for (r = 0; r < n; ++r) tot += (size_t)reads[r].l_seq;
	S = (qa_seed_t *)calloc(1, sizeof(*S));
	S->bwt = bwt, S->n_reads = n, S->reads = reads;
	S->call_off = (uint32_t *)malloc(4 * ((size_t)n + 1)), S->intv_off = (uint64_t *)malloc(8 * ((size_t)n + 1));
	/* output arrays sized from the densest batch seen so far in this process (calls / intervals / positions per base) */
	ts[0] = stage_now();
	if (!(getenv("BMH_SEED_FUSED") && getenv("BMH_SEED_FUSED")[0] == '0')) {
		/* SMEMs and the suffix-array entries chaining will ask for (bwamem.c:218-225) in one device round trip */
		size_t sa_cap;
		uint64_t n_pos = 0;
		for (call_cap = (size_t)(g_seed_density[0] * 1.3 * (double)tot) + 4 * (size_t)n + 64,
		    intv_cap = (size_t)(g_seed_density[1] * 1.3 * (double)tot) + 1024, sa_cap = (size_t)(g_seed_density[2] * 1.3 * (double)tot) + 4096;;
		     call_cap *= 2, intv_cap *= 2, sa_cap *= 2) {
			S->calls = (bmh_smem_call_t *)malloc(sizeof(bmh_smem_call_t) * call_cap);
			S->intv = (bmh_smem_intv_t *)malloc(sizeof(bmh_smem_intv_t) * intv_cap);
			S->sa_k = (uint64_t *)malloc(8 * intv_cap), S->sa_pos = (uint64_t *)malloc(8 * sa_cap);
			rc = bmh_seed_batch(ctx, &so, opt->max_occ, n, reads, S->call_off, S->calls, call_cap, S->intv_off, S->intv, intv_cap, S->sa_k, S->sa_pos,
			                    sa_cap, &n_pos);
			if (rc != BMH_E_CIGAR_CAP) break;
			free(S->calls), free(S->intv), free(S->sa_k), free(S->sa_pos);
		}
		if (rc) bmh_tls_die(bmh_last_error(ctx), rc);
		S->n_sa = (size_t)n_pos;
		ts[1] = ts[2] = ts[3] = stage_now();
		if (tot) { /* (a benign race: statistics that only size buffers) */
			const double dc = (double)S->call_off[n] / (double)tot, di = (double)S->intv_off[n] / (double)tot, dp = (double)n_pos / (double)tot;
			if (dc > g_seed_density[0]) g_seed_density[0] = dc;
			if (di > g_seed_density[1]) g_seed_density[1] = di;
			if (dp > g_seed_density[2]) g_seed_density[2] = dp;
		}
	} else { /* the same in three calls (the A/B path): bmh_smem_batch, bmh_chain_sa_keys, bmh_sa_batch */
		for (call_cap = (size_t)(g_seed_density[0] * 1.3 * (double)tot) + 4 * (size_t)n + 64,
		    intv_cap = (size_t)(g_seed_density[1] * 1.3 * (double)tot) + 1024;;
		     call_cap *= 2, intv_cap *= 2) {
			S->calls = (bmh_smem_call_t *)malloc(sizeof(bmh_smem_call_t) * call_cap);
			S->intv = (bmh_smem_intv_t *)malloc(sizeof(bmh_smem_intv_t) * intv_cap);
			rc = bmh_smem_batch(ctx, &so, n, reads, S->call_off, S->calls, call_cap, S->intv_off, S->intv, intv_cap);
			if (rc != BMH_E_CIGAR_CAP) break;
			free(S->calls), free(S->intv);
		}
		if (rc) bmh_tls_die(bmh_last_error(ctx), rc);
		ts[1] = stage_now();
		if (tot) {
			const double dc = (double)S->call_off[n] / (double)tot, di = (double)S->intv_off[n] / (double)tot;
			if (dc > g_seed_density[0]) g_seed_density[0] = dc;
			if (di > g_seed_density[1]) g_seed_density[1] = di;
		}
		{
			bmh_chain_opt_t co;
			memset(&co, 0, sizeof(co));
			co.min_seed_len = opt->min_seed_len, co.max_occ = opt->max_occ;
			S->sa_k = (uint64_t *)malloc(8 * (S->intv_off[n] + 1)); /* here: sa_off, one entry per interval */
			nk = (size_t)bmh_chain_sa_keys(&co, S->intv_off[n], S->intv, S->sa_k, 0);
			keys = (uint64_t *)malloc(8 * (nk + 1)), S->sa_pos = (uint64_t *)malloc(8 * (nk + 1));
			bmh_chain_sa_keys(&co, S->intv_off[n], S->intv, S->sa_k, keys);
			S->n_sa = nk;
		}
		ts[2] = stage_now();
		if ((rc = bmh_sa_batch(ctx, keys, (int64_t)nk, S->sa_pos))) bmh_tls_die(bmh_last_error(ctx), rc);
		free(keys);
		ts[3] = stage_now();
	}
	__sync_fetch_and_add(&g_seed_us[0], (long long)((ts[1] - ts[0]) * 1e6)), __sync_fetch_and_add(&g_seed_us[1], (long long)((ts[2] - ts[1]) * 1e6));
	__sync_fetch_and_add(&g_seed_us[2], (long long)((ts[3] - ts[2]) * 1e6));
	qa_seed = S;
}

static void qa_seed_batch_end(void)
{
	qa_seed_t *S = qa_seed;
	qa_seed = 0;
	if (!S) return;
	free(S->call_off), free(S->calls), free(S->intv_off), free(S->intv), free(S->sa_k), free(S->sa_pos), free(S);
}

static long long g_p1_cnt[4]; /* chains, seeds extended, seeds speculated in vain, short-chain Smith-Watermans */
static long long g_p1_us[5]; /* phase 1, thread-microseconds: wait for a GPU slot, seeding batch, chaining (host), wait, extension batch */

/* (a static body behind both exported names: inside a process that also holds the reference's own definitions -- the tests load
 * libbwa_ref.so next to this library -- a call through the exported name would bind to whichever was loaded first) */
static bmh_alnreg_v *align_batch(const ref_mem_opt_t *opt, const void *bwt, const ref_bntseq_head_t *bns,
                                 const uint8_t *pac, ref_bseq1_t *seqs, int start, int batch_size)
{
	bmh_chain_v *chn = (bmh_chain_v *)malloc(sizeof(bmh_chain_v) * (size_t)batch_size);
	bmh_alnreg_v *regs = (bmh_alnreg_v *)calloc((size_t)batch_size, sizeof(bmh_alnreg_v));
	bmh_read_t *reads = (bmh_read_t *)malloc(sizeof(bmh_read_t) * (size_t)batch_size);
	bmh_params_t p;
	bmh_ctx_t *ctx;
	double tq[6];
	int b, i, rc;

	for (b = 0; b < batch_size; ++b) { /* bwamem.c:1093-1094 */
		ref_bseq1_t *s = &seqs[start + b];
		for (i = 0; i < s->l_seq; ++i) s->seq[i] = s->seq[i] < 4 ? s->seq[i] : (char)nst_nt4_table[(int)s->seq[i]];
		reads[b].l_seq = s->l_seq, reads[b].seq = (const uint8_t *)s->seq;
	}
	memset(&p, 0, sizeof(p)); /* the hot-path fields of mem_opt_t */
	p.o_del = opt->o_del, p.e_del = opt->e_del, p.o_ins = opt->o_ins, p.e_ins = opt->e_ins, p.zdrop = opt->zdrop;
	p.a = opt->a, p.w = opt->w, p.pen_clip5 = opt->pen_clip5, p.pen_clip3 = opt->pen_clip3;
	memcpy(p.mat, opt->mat, 25);
	tq[0] = tq[1] = stage_now();
	ctx = bmh_pool_get(&p);
	qa_seed_batch_begin(ctx, opt, (const ref_bwt_t *)bwt, batch_size, reads); /* SMEMs + suffix-array look-ups of the batch on the GPU */
	bmh_pool_put(ctx);
	tq[2] = stage_now();
	{ /* chaining: mem_chain + mem_chain_flt (bwamem.c:1095-1097) over the batch's tables */
		const qa_seed_t *S = qa_seed;
		bmh_chain_opt_t co;
		co.w = opt->w, co.max_chain_gap = opt->max_chain_gap, co.min_seed_len = opt->min_seed_len, co.max_occ = opt->max_occ;
		co.split_len = (int)(opt->min_seed_len * opt->split_factor + .499), co.split_width = opt->split_width;
		co.mask_level = opt->mask_level, co.chain_drop_ratio = opt->chain_drop_ratio;
		if ((rc = bmh_chain_reads(&co, bns->l_pac, batch_size, reads, S->call_off, S->calls, S->intv_off, S->intv, S->sa_k, S->sa_pos, chn)))
			bmh_tls_die("the batch's seeding tables do not cover its chaining", rc);
		{
			long long nc = 0;
			for (b = 0; b < batch_size; ++b) nc += (long long)chn[b].n;
			__sync_fetch_and_add(&g_p1_cnt[0], nc);
		}
	}
	qa_seed_batch_end();
	tq[3] = stage_now();
	tq[4] = stage_now();
	ctx = bmh_pool_get(&p);
	{ /* reference resident in HBM, shared by all contexts: the kernels do bns_get_seq themselves.  BMH_PAC_RESIDENT=0
	   * falls back to host-decoded windows in the pool. */
		const char *e = getenv("BMH_PAC_RESIDENT");
		if (!(e && e[0] == '0') && (rc = bmh_ctx_set_pac(ctx, pac, bns->l_pac))) bmh_tls_die(bmh_last_error(ctx), rc);
	}
	if ((rc = bmh_chains2regs_batch(ctx, bns->l_pac, pac, batch_size, reads, chn, opt->min_seed_len, regs))) /* bwamem.c:1101-1110 */
		bmh_tls_die(bmh_last_error(ctx), rc);
	{
		bmh_driver_stats_t st;
		bmh_driver_stats(ctx, &st);
		__sync_fetch_and_add(&g_p1_cnt[1], st.seeds_extended), __sync_fetch_and_add(&g_p1_cnt[2], st.seeds_speculated);
		__sync_fetch_and_add(&g_p1_cnt[3], st.short_sw);
	}
	bmh_pool_put(ctx);
	tq[5] = stage_now();
	{ /* thread-seconds per stage, summed over the run (BMH_VERBOSE prints them per chunk) */
		static const int a_[5] = {0, 1, 2, 3, 4};
		int k;
		for (k = 0; k < 5; ++k) {
			const double d = tq[a_[k] + 1] - tq[a_[k]];
			long long us = (long long)(d * 1e6);
			__sync_fetch_and_add(&g_p1_us[k], us);
		}
	}

	for (b = 0; b < batch_size; ++b) { /* host stages after the path: bwamem.c:1106,1112-1117 */
		for (i = 0; i < (int)chn[b].n; ++i) free(chn[b].a[i].seeds);
		free(chn[b].a);
		regs[b].n = (size_t)bmh_sort_and_dedup((int)regs[b].n, regs[b].a, opt->mask_level_redun); /* bwamem.c:1114 */
		if ((opt->flag & REF_MEM_F_NO_EXACT) && regs[b].n && regs[b].a[0].truesc == seqs[start + b].l_seq * opt->a) { /* mem_test_and_remove_exact, bwamem.c:438-443 */
			memmove(regs[b].a, regs[b].a + 1, (regs[b].n - 1) * sizeof(bmh_alnreg_t));
			--regs[b].n;
		}
	}
	free(chn);
	free(reads);
	return regs; /* caller copies and frees, bwamem.c:1272-1278 */
}

bmh_alnreg_v *mem_align1_core_batched(const ref_mem_opt_t *opt, const void *bwt, const ref_bntseq_head_t *bns,
                                      const uint8_t *pac, ref_bseq1_t *seqs, int start, int batch_size)
{
	return align_batch(opt, bwt, bns, pac, seqs, start, batch_size);
}

/* mem_align1_core() with the reference's exact signature (bwamem.c:1122-1149; used by mem_align1 :1151 and example.c:44): one
 * read through the same drivers as a batch of one -- `seq` is converted to base codes in place as at :1128-1129, the result
 * vector is malloc'd and the caller's (as kv_push would have left it).  A batch of one cannot fill a GPU; the entry point
 * exists so that library users of the reference's API get the same regions from the same code path. */
bmh_alnreg_v mem_align1_core(const ref_mem_opt_t *opt, const void *bwt, const ref_bntseq_head_t *bns, const uint8_t *pac, int l_seq, char *seq)
{
	ref_bseq1_t one;
	bmh_alnreg_v *v, out;
	memset(&one, 0, sizeof(one));
	one.l_seq = l_seq, one.seq = seq;
	v = align_batch(opt, bwt, bns, pac, &one, 0, 1);
	out = v[0];
	free(v);
	return out;
}

/* =====================================================================================================================
 * mem_process_seqs() with the reference's exact signature (bwamem.h:117, bwamem.c:1297-1327): the same three steps as
 * the reference -- phase 1 through the batching seam above, insert-size statistics, phase 2 -- every data-parallel part
 * of them as GPU batches and the host code between them the library's own:
 *   phase 1   mem_align1_core_batched above (seeding batch, chaining, bmh_chain2aln_batch, bmh_sort_and_dedup)
 *   pairs     bmh_pestat, then the whole chunk's mate rescue (bmh_matesw_batch, one slice per GPU place)
 *   phase 2   bmh_sam_batch per slice of the chunk: primary marking, pairing, the global alignments of exactly the
 *             regions that get printed (bmh_reg2cigar_batch), SAM text
 * Nothing of the reference's bwamem.c / bwamem_pair.c runs after chaining.
 */

#define REF_MEM_F_PE 0x2         /* bwamem.h:14 */
#define REF_MEM_F_NO_RESCUE 0x20 /* bwamem.h:18 */

extern void kt_for(int n_threads, void (*func)(void *, int, int), void *data, int n);                       /* kthread.c */
extern void kt_for_batch(int n_threads, void (*func)(void *, int, int, int), void *data, int n, int batch);  /* kthread_batch.c:44 */
extern int bwa_verbose;
extern char bwa_rg_id[256]; /* bwa.c:16 */

typedef struct {
	const ref_mem_opt_t *opt;
	const void *bwt;
	const ref_bntseq_head_t *bns;
	const uint8_t *pac;
	ref_bseq1_t *seqs;
	bmh_alnreg_v *regs;
} qa_worker_t;

static void qa_worker1_batched(void *data, int start, int batch_size, int tid) /* == worker1_batched, bwamem.c:1264-1279 */
{
	qa_worker_t *w = (qa_worker_t *)data;
	bmh_alnreg_v *ret = align_batch(w->opt, w->bwt, w->bns, w->pac, w->seqs, start, batch_size);
	int i;
	(void)tid;
	for (i = start; i < start + batch_size; ++i) w->regs[i] = ret[i - start];
	free(ret);
}

static int qa_dedup(void *user, int n, bmh_alnreg_t *a) /* bmh_dedup_fn of the mate-rescue driver */
{
	return bmh_sort_and_dedup(n, a, ((const ref_mem_opt_t *)user)->mask_level_redun);
}

typedef struct {
	const ref_mem_opt_t *opt;
	const ref_bntseq_head_t *bns;
	const uint8_t *pac;
	int n, n_slices;
	int64_t n_processed;
	ref_bseq1_t *seqs;
	bmh_alnreg_v *regs;
	const bmh_read_t *reads;
	const bmh_params_t *params;
	const bmh_sam_opt_t *sopt;
	const bmh_pestat_t *pes;
	int resident;
} qa_slice_job_t;

static bmh_ctx_t *qa_slice_ctx(const qa_slice_job_t *J)
{
	bmh_ctx_t *ctx = bmh_pool_get(J->params);
	int rc;
	if (J->resident && (rc = bmh_ctx_set_pac(ctx, J->pac, J->bns->l_pac))) bmh_tls_die(bmh_last_error(ctx), rc);
	return ctx;
}

/* mate rescue, one slice of pairs per host thread */
static long long g_msw_calls, g_msw_rounds_max, g_msw_bytes;
static void qa_matesw_slice(void *data, int k, int tid)
{
	const qa_slice_job_t *J = (const qa_slice_job_t *)data;
	const int np = J->n >> 1, lo = (int)((int64_t)np * k / J->n_slices), hi = (int)((int64_t)np * (k + 1) / J->n_slices);
	bmh_matesw_opt_t mo;
	bmh_driver_stats_t st;
	bmh_ctx_t *ctx;
	int rc;
	(void)tid;
	if (hi <= lo) return;
	ctx = qa_slice_ctx(J);
	mo.pen_unpaired = J->opt->pen_unpaired, mo.max_matesw = J->opt->max_matesw, mo.min_seed_len = J->opt->min_seed_len, mo.rsv = 0;
	if ((rc = bmh_matesw_batch(ctx, J->bns->l_pac, J->pac, hi - lo, J->reads + 2 * lo, J->regs + 2 * lo, J->pes, &mo, qa_dedup,
	                           (void *)J->opt, 0)))
		bmh_tls_die(bmh_last_error(ctx), rc);
	bmh_driver_stats(ctx, &st);
	bmh_pool_put(ctx);
	__sync_fetch_and_add(&g_msw_calls, st.ext_tasks), __sync_fetch_and_add(&g_msw_bytes, st.pool_bytes);
	if (st.rounds > g_msw_rounds_max) g_msw_rounds_max = st.rounds; /* (a benign race: statistics only) */
}

/* phase 2, one slice of reads (whole pairs) per host thread: worker2 of the reference (bwamem.c:1281-1295) */
static long long g_sam_us[2]; /* thread-microseconds: waiting for a GPU place, bmh_sam_batch */
static void qa_sam_slice(void *data, int k, int tid)
{
	const qa_slice_job_t *J = (const qa_slice_job_t *)data;
	const int unit = (J->sopt->flag & BMH_MEM_F_PE) ? 2 : 1, nu = J->n / unit;
	const int lo = unit * (int)((int64_t)nu * k / J->n_slices), hi = unit * (int)((int64_t)nu * (k + 1) / J->n_slices);
	bmh_ctx_t *ctx;
	double t0, t1, t2;
	int rc, i;
	(void)tid;
	if (hi <= lo) return;
	t0 = t1 = stage_now();
	ctx = qa_slice_ctx(J);
	if ((rc = bmh_sam_batch(ctx, J->sopt, (const bmh_refidx_t *)J->bns, J->pac, J->pes, J->n_processed + lo, hi - lo, (bmh_seq_t *)(J->seqs + lo),
	                        J->regs + lo, bwa_rg_id)))
		bmh_tls_die(rc == BMH_E_ARG ? "a region could not be turned into an alignment (the reference aborts here too, bwamem.c:1183-1186)" : bmh_last_error(ctx), rc);
	bmh_pool_put(ctx);
	t2 = stage_now();
	for (i = lo; i < hi; ++i) free(J->regs[i].a);
	__sync_fetch_and_add(&g_sam_us[0], (long long)((t1 - t0) * 1e6)), __sync_fetch_and_add(&g_sam_us[1], (long long)((t2 - t1) * 1e6));
}

/* host threads of a phase: the program's -t unless the named variable says otherwise (a thread that waits for the GPU
 * sleeps, so more threads than cores can pay) */
static int qa_threads(const char *var, int dflt)
{
	const char *e = getenv(var);
	const int v = e ? atoi(e) : 0;
	return v > 0 ? v : dflt > 0 ? dflt : 1;
}

void mem_process_seqs(const ref_mem_opt_t *opt, const void *bwt, const ref_bntseq_head_t *bns, const uint8_t *pac,
                      int64_t n_processed, int n, ref_bseq1_t *seqs, const bmh_pestat_t *pes0)
{
	const int pe = (opt->flag & REF_MEM_F_PE) != 0;
	const int rescue = pe && !(opt->flag & REF_MEM_F_NO_RESCUE);
	qa_worker_t w;
	bmh_pestat_t pes[4];
	bmh_params_t p;
	bmh_sam_opt_t so;
	bmh_read_t *reads;
	qa_slice_job_t J;
	double ctime, rtime, t_[4], t_pes;
	int i, nt2;
	ctime = cputime(), rtime = realtime();
	t_[0] = rtime;
	if (n_processed == 0 && g_t_loaded > 0 && getenv("BMH_VERBOSE"))
		fprintf(stderr, "[bwamem_hip] the first chunk starts %.3f s after the shim was loaded\n", rtime - g_t_loaded);
	w.opt = opt, w.bwt = bwt, w.bns = bns, w.pac = pac, w.seqs = seqs;
	w.regs = (bmh_alnreg_v *)malloc((size_t)n * sizeof(bmh_alnreg_v));
	{ /* bwamem.c:1313, with the batch size evened out: -b 65536 cuts a chunk of 1 066 668 reads into 16 batches and a 17th
	   * of 18 092 that one thread runs alone after all others are done; the same reads in 16 batches of 66 667 are not a
	   * read slower per batch.  As many batches as -b asks for, rounded to a multiple of the thread count, at least 4 096
	   * reads each.  BMH_BATCH_EXACT=1 keeps -b to the read. */
		const int nt = qa_threads("BMH_P1_THREADS", opt->n_threads);
		int b = opt->batch_size > 0 ? opt->batch_size : 1;
		if (!getenv("BMH_BATCH_EXACT") && n > 0) {
			int64_t rounds = ((int64_t)n + (int64_t)b * nt / 2) / ((int64_t)b * nt), nb;
			if (rounds < 1) rounds = 1;
			nb = rounds * nt;
			if ((int64_t)n / nb < 4096) nb = ((int64_t)n + 4095) / 4096;
			b = (int)(((int64_t)n + nb - 1) / nb);
		}
		kt_for_batch(nt, qa_worker1_batched, &w, n, b);
	}
	t_[1] = realtime();
	memset(&so, 0, sizeof(so)); /* the fields of mem_opt_t phase 2 reads */
	so.a = opt->a, so.b = opt->b, so.o_del = opt->o_del, so.e_del = opt->e_del, so.o_ins = opt->o_ins, so.e_ins = opt->e_ins;
	so.pen_unpaired = opt->pen_unpaired, so.w = opt->w, so.T = opt->T, so.flag = opt->flag, so.min_seed_len = opt->min_seed_len;
	so.max_ins = opt->max_ins, so.mapQ_coef_fac = opt->mapQ_coef_fac, so.max_matesw = opt->max_matesw, so.mask_level = opt->mask_level;
	so.mask_level_redun = opt->mask_level_redun, so.mapQ_coef_len = opt->mapQ_coef_len;
	memcpy(so.mat, opt->mat, 25);
	if (pe) {                                                                /* bwamem.c:1314-1317 */
		if (pes0) memcpy(pes, pes0, 4 * sizeof(bmh_pestat_t));
		else bmh_pestat(&so, bns->l_pac, n, w.regs, pes, bwa_verbose);
	}
	t_pes = realtime();
	/* reads are base codes by now (bwamem.c:1093-1094) */
	reads = (bmh_read_t *)malloc(sizeof(bmh_read_t) * (size_t)n);
	for (i = 0; i < n; ++i) reads[i].l_seq = seqs[i].l_seq, reads[i].seq = (const uint8_t *)seqs[i].seq;
	memset(&p, 0, sizeof(p));
	p.o_del = opt->o_del, p.e_del = opt->e_del, p.o_ins = opt->o_ins, p.e_ins = opt->e_ins, p.zdrop = opt->zdrop;
	p.a = opt->a, p.w = opt->w, p.pen_clip5 = opt->pen_clip5, p.pen_clip3 = opt->pen_clip3;
	memcpy(p.mat, opt->mat, 25);
	{
		const char *pr = getenv("BMH_PAC_RESIDENT");
		J.resident = !(pr && pr[0] == '0');
	}
	J.opt = opt, J.bns = bns, J.pac = pac, J.n = n, J.seqs = seqs, J.regs = w.regs, J.reads = reads, J.params = &p, J.pes = pes;
	J.sopt = &so, J.n_processed = n_processed;
	/* Phase 2 and mate rescue are host code with a GPU batch in the middle of every slice; half again as many threads as -t
	 * keep the cores busy while some of them sleep on the GPU (measured at -t 16 on 16 cores: a chunk's rescue + phase 2
	 * 0.105 -> 0.080 s; twice as many is no better; phase 1 does not gain). */
	nt2 = qa_threads("BMH_P2_THREADS", opt->n_threads + opt->n_threads / 2);
	J.n_slices = qa_threads("BMH_P2_SLICES", nt2);
	if (rescue) { /* the whole chunk's mate rescue (the block of mem_sam_pe at bwamem_pair.c:251-263), one bmh_matesw_batch per slice */
		g_msw_calls = g_msw_rounds_max = g_msw_bytes = 0;
		kt_for(nt2, qa_matesw_slice, &J, J.n_slices);
		if (getenv("BMH_VERBOSE"))
			fprintf(stderr, "[bwamem_hip] mate rescue: %d pairs, %lld ksw_align2 calls in %lld GPU rounds, %lld pool bytes\n", n >> 1,
			        g_msw_calls, g_msw_rounds_max, g_msw_bytes);
	}
	t_[2] = realtime();
	kt_for(nt2, qa_sam_slice, &J, J.n_slices); /* bwamem.c:1318-1319 */
	t_[3] = realtime();
	free(reads);
	if (getenv("BMH_VERBOSE")) {
		fprintf(stderr, "[bwamem_hip] seeding batch thread-seconds so far: bmh_smem_batch %.3f, look-up keys %.3f, bmh_sa_batch %.3f\n",
		        g_seed_us[0] * 1e-6, g_seed_us[1] * 1e-6, g_seed_us[2] * 1e-6);
		fprintf(stderr, "[bwamem_hip] phase 1 thread-seconds so far: wait %.3f, seeding batch %.3f, chaining on the host %.3f, wait %.3f, extension batch %.3f\n",
		        g_p1_us[0] * 1e-6, g_p1_us[1] * 1e-6, g_p1_us[2] * 1e-6, g_p1_us[3] * 1e-6, g_p1_us[4] * 1e-6);
		fprintf(stderr, "[bwamem_hip] phase 1 so far: %lld chains from bmh_chain_reads, %lld seeds extended (+%lld speculated in vain), %lld short-chain Smith-Watermans batched\n",
		        g_p1_cnt[0], g_p1_cnt[1], g_p1_cnt[2], g_p1_cnt[3]);
		fprintf(stderr, "[bwamem_hip] phase 2 thread-seconds so far: wait %.3f, bmh_sam_batch %.3f\n", g_sam_us[0] * 1e-6, g_sam_us[1] * 1e-6);
		fprintf(stderr, "[bwamem_hip] chunk of %d reads: phase 1 %.3f s, pestat %.3f s + mate rescue %.3f s, phase 2 (marking, pairing, global alignments, SAM) %.3f s\n", n,
		        t_[1] - t_[0], t_pes - t_[1], t_[2] - t_pes, t_[3] - t_[2]);
	}
	free(w.regs);
	if (bwa_verbose >= 3)
		fprintf(stderr, "[M::%s] Processed %d reads in %.3f CPU sec, %.3f real sec\n", __func__, n, cputime() - ctime, realtime() - rtime);
}
