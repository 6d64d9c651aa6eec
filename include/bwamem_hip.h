/*
 * bwamem_hip.h -- C-ABI of libbwamem_hip.so: BWA-MEM's seed-extension hot
 * path (banded affine-gap DP) on AMD Instinct MI355X (gfx950).
 *
 * This is the drop-in boundary for the path named in BASELINE.json
 * (SURVEY.md §8b).  Plain C: pointers, sizes, fixed-width integers; no C++,
 * HIP or torch types appear in any signature.  Citations are relative to the
 * reference tree (peterpengwei/bwa-mem-quickassist, bwa-0.7.8/).
 *
 * Three levels, lowest first:
 *
 *   L2  batch DP         bmh_extend_batch*   replaces N calls of ksw_extend2 (ksw.h:108, ksw.c:379-476)
 *                        bmh_global_batch*   replaces N calls of ksw_global2 (ksw.h:84,  ksw.c:501-584)
 *   L2' per-call drop-in ksw_extend2 / ksw_global2 with the exact ksw.h signatures
 *                        (libbwamem_hip_dropin.so; each call is a batch of one)
 *   L3  extension driver bmh_chain2aln_batch and the fork's own seam
 *                        mem_chain2aln_batched(...) (bwamem.c:580, call site bwamem.c:1110),
 *                        which replace the per-read loop over mem_chain2aln (bwamem.c:730-878)
 *                        inside mem_align1_core_batched (bwamem.c:1086-1120).
 *
 * Error convention: every function returns BMH_OK (0) or a negative
 * BMH_E_* code; nothing is ever silently dropped.  The reference has no error
 * returns on this path (assert/exit: bwamem.c:758,845,1184; malloc_wrap.c:14-19),
 * so the integration stub in INTEGRATION.md aborts on a negative code.
 * There is NO CPU fallback inside this library: without a usable GPU every
 * compute entry point fails with BMH_E_NODEVICE.
 */
#ifndef BWAMEM_HIP_H
#define BWAMEM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BMH_VERSION 310 /* 0.3.1: bmh_region_cigar_batch (0.3.0: bmh_smem_opt_t grew min_emit_len (20 bytes), mem_align1_core, *_batch_sharded) */

enum {
	BMH_OK = 0,
	BMH_E_NODEVICE = -1, /* no HIP device / HIP runtime error at init        */
	BMH_E_HIP = -2,      /* a HIP call failed; see bmh_last_error()           */
	BMH_E_ARG = -3,      /* invalid argument (NULL, negative size, ...)       */
	BMH_E_RANGE = -4,    /* a task is outside the supported range (see below) */
	BMH_E_NOMEM = -5,    /* host or device allocation failed                  */
	BMH_E_CIGAR_CAP = -6 /* a global task produced more CIGAR ops than its cap */
};

/* ---- scoring / band parameters: the hot-path fields of mem_opt_t
 * (bwamem.h:21-48; defaults bwamem.c:45-75; matrix bwa.c:77-86). Constant for a run. */
typedef struct bmh_params {
	int32_t o_del, e_del, o_ins, e_ins; /* bwamem.h:23-24                             */
	int32_t zdrop;                      /* bwamem.h:28 ; <=0 disables (ksw.c:455)      */
	int32_t a;                          /* match score; only the L3 driver reads it    */
	int32_t w;                          /* opt->w;       only the L3 driver reads it    */
	int32_t pen_clip5, pen_clip3;       /* bwamem.h:26;  only the L3 driver reads it    */
	int8_t mat[25];                     /* bwamem.h:47 ; m = 5                           */
	int8_t pad_[3];
} bmh_params_t;

/* ---- one ksw_extend2 call (ksw.c:379).  32 bytes.
 * Sequences are base codes 0..4, one byte per base, living in a byte pool.
 * With BMH_F_QREV / BMH_F_TREV the sequence is read backwards from the offset:
 * base k = pool[off - k].  That is how the left extension's reversed copies
 * (bwamem.c:813-817) are expressed without materialising them. */
#define BMH_F_QREV 1u
#define BMH_F_TREV 2u
/* target taken from the 2-bit reference resident on the device (bmh_ctx_set_pac): t_off is then a position on
 * bwa's doubled coordinate [0, 2*l_pac) (reference bntseq.c:355-376), walked forwards or, with BMH_F_TREV,
 * backwards; no target bytes are needed in the pool */
#define BMH_F_TPAC 4u
typedef struct bmh_ext_task {
	uint64_t q_off;    /* pool offset of query base 0                     */
	uint64_t t_off;    /* pool offset of target base 0                    */
	uint16_t qlen;     /* ksw_extend2 arg 1                               */
	uint16_t tlen;     /* arg 3                                           */
	int32_t h0;        /* arg 14                                          */
	int16_t w;         /* arg 11 (before the clamp of ksw.c:398-406)      */
	int16_t end_bonus; /* arg 12                                          */
	uint16_t flags;    /* BMH_F_*                                         */
	uint16_t rsv_;
} bmh_ext_task_t;

/* ---- what ksw_extend2 returns (ksw.c:470-475).  24 bytes. */
typedef struct bmh_ext_result {
	int32_t score, qle, tle, gtle, gscore, max_off;
} bmh_ext_result_t;

/* ---- one ksw_global2 call (ksw.c:501).  32 bytes. */
typedef struct bmh_glb_task {
	uint64_t q_off, t_off;
	uint16_t qlen, tlen;
	int32_t w;          /* arg 11                                                   */
	uint32_t cigar_off; /* first uint32 slot of this task in the CIGAR output pool   */
	uint32_t cigar_cap; /* slots reserved; 0 = score only (cigar_==NULL, ksw.c:566)  */
} bmh_glb_task_t;

typedef struct bmh_glb_result {
	int32_t score;   /* return value, ksw.c:565                                  */
	int32_t n_cigar; /* *n_cigar_; if > cigar_cap the CIGAR was NOT fully written */
} bmh_glb_result_t;

/* Supported range (checked on the host, BMH_E_RANGE otherwise):
 *   extend: 0 <= qlen,tlen <= 65535; scores must fit int16 lanes:
 *           h0 + qlen*max(mat) <= 32000; o_ins >= 0 (SURVEY.md §7 hard part 1).
 *   global: qlen,tlen <= 65535. */

typedef struct bmh_ctx bmh_ctx_t;

int bmh_version(void);
const char *bmh_strerror(int code);
const char *bmh_last_error(const bmh_ctx_t *ctx); /* text of the last failing HIP call */

int bmh_device_count(int *n);
/* One context = one GPU + one HIP stream + grow-only device/pinned workspaces.
 * A context is used by one host thread at a time. */
int bmh_ctx_create(bmh_ctx_t **ctx, int device);
int bmh_ctx_destroy(bmh_ctx_t *ctx);
int bmh_ctx_set_params(bmh_ctx_t *ctx, const bmh_params_t *p);
/* Run on a caller-owned hipStream_t (passed as void*); NULL restores the context's own. */
int bmh_ctx_set_stream(bmh_ctx_t *ctx, void *hip_stream);
/* Make the 2-bit packed reference (bwaidx_t.pac, reference bwa.h:16; l_pac/4+1 bytes) resident in HBM: enables
 * BMH_F_TPAC tasks and lets the L3 drivers skip the host-side bns_get_seq.  hg38: 0.78 GB, uploaded once. */
int bmh_ctx_set_pac(bmh_ctx_t *ctx, const uint8_t *pac, int64_t l_pac);
int bmh_ctx_sync(bmh_ctx_t *ctx); /* waits for the stream; returns a pending BMH_E_RANGE/CIGAR_CAP of a *_device call */
/* Capacity hint for the *_device entry points, which cannot look at the tasks on the host:
 * the longest query the launch must handle (default 512).  Longer tasks fail with BMH_E_RANGE. */
int bmh_ctx_set_qcap(bmh_ctx_t *ctx, int max_qlen);

/* A process-wide gate around the DEVICE SECTIONS of the host-buffer entry points below (upload, kernels, download):
 * enter() is called before the first device operation of a call and leave() after its last.  A program that drives the
 * library from many host threads -- the reference runs phase 1 and 2 on n_threads pthreads -- can bound how many of
 * them are inside device sections at once (more only queue up behind one another on the GPU), while the host work of
 * the L3 drivers around those sections (state machines, band logic, text) runs on all threads.  NULL, NULL removes it. */
typedef void (*bmh_gate_fn)(void);
int bmh_set_device_gate(bmh_gate_fn enter, bmh_gate_fn leave);
/* How the host-buffer entry points wait for the GPU, process-wide.  0 (default): spinning on the completion signal --
 * the shortest latency, one core per waiting thread.  1: sleeping until the GPU's interrupt (an event created with
 * hipEventBlockingSync, and hipDeviceScheduleBlockingSync on the devices of contexts created afterwards) -- for programs
 * with as many or more runnable host threads than cores: measured on the preload shim at 16 threads on 16 cores, a third
 * of all CPU time was spent spinning inside these waits. */
void bmh_set_wait_mode(int blocking);

/* ---- L2, host buffers: H2D copy, launch, D2H copy, synchronous on return. */
int bmh_extend_batch(bmh_ctx_t *ctx, const uint8_t *seqpool, size_t pool_bytes,
                     const bmh_ext_task_t *tasks, int64_t n, bmh_ext_result_t *results);
int bmh_global_batch(bmh_ctx_t *ctx, const uint8_t *seqpool, size_t pool_bytes,
                     const bmh_glb_task_t *tasks, int64_t n, bmh_glb_result_t *results,
                     uint32_t *cigar_pool, size_t cigar_pool_words);

/* Leave a sequence pool resident on the device; afterwards bmh_extend_batch(ctx, NULL, 0, ...) and
 * bmh_global_batch(ctx, NULL, 0, ...) run tasks against it without re-uploading (the L3 drivers
 * upload once per batch and then only move task and result records per round). */
int bmh_upload_pool(bmh_ctx_t *ctx, const uint8_t *seqpool, size_t pool_bytes);

/* ---- L2, device-resident buffers: asynchronous on the context's stream.
 * `d_order` (nullable) is a device array of n task indices giving the launch
 * order (e.g. sorted by length for tail balance); results stay at task index. */
int bmh_extend_batch_device(bmh_ctx_t *ctx, const uint8_t *d_seqpool,
                            const bmh_ext_task_t *d_tasks, int64_t n,
                            bmh_ext_result_t *d_results, const uint32_t *d_order);
int bmh_global_batch_device(bmh_ctx_t *ctx, const uint8_t *d_seqpool,
                            const bmh_glb_task_t *d_tasks, int64_t n,
                            bmh_glb_result_t *d_results, uint32_t *d_cigar_pool,
                            const uint32_t *d_order);

/* ---- static shard of one host batch over several contexts (one per GPU,
 * SURVEY.md §8e): contiguous split, one stream per device, no collective. */
int bmh_extend_batch_sharded(bmh_ctx_t *const *ctxs, int n_ctx, const uint8_t *seqpool,
                             size_t pool_bytes, const bmh_ext_task_t *tasks, int64_t n,
                             bmh_ext_result_t *results);
/* The same static split for the other three batches of the DP path (declared here, records defined below): the fused per-seed
 * records, ksw_global2 + traceback, ksw_align2.  Contiguous task ranges like kt_for_batch's (reference kthread_batch.c:44-56,
 * bwamem.c:1313), every device gets the whole pool, everything is enqueued on every device before any is waited for, results
 * land at their task index; a shard's CIGAR words are copied into the caller's pool task by task. */
struct bmh_seed_task; struct bmh_seed_result; struct bmh_sw_task; struct bmh_sw_result;
int bmh_seedext_batch_sharded(bmh_ctx_t *const *ctxs, int n_ctx, const uint8_t *seqpool, size_t pool_bytes,
                              const struct bmh_seed_task *tasks, int64_t n, struct bmh_seed_result *results);
int bmh_global_batch_sharded(bmh_ctx_t *const *ctxs, int n_ctx, const uint8_t *seqpool, size_t pool_bytes,
                             const bmh_glb_task_t *tasks, int64_t n, bmh_glb_result_t *results,
                             uint32_t *cigar_pool, size_t cigar_pool_words);
int bmh_sw_batch_sharded(bmh_ctx_t *const *ctxs, int n_ctx, const uint8_t *seqpool, size_t pool_bytes,
                         const struct bmh_sw_task *tasks, int64_t n, struct bmh_sw_result *results);

/* ---- L2.5: one record per SEED -- the accelerator record the fork sketched and never used
 * (ext_param_t / ext_res_t, reference bwamem.c:553-577).  The device runs, for every seed, what
 * mem_chain2aln does between bwamem.c:810 and :866: the left extension on the reversed flanks (up to MAX_BAND_TRY = 2
 * band widths, retry rule :828), the clip-or-reach-the-end decision (:831-837), the right extension started from the
 * LEFT SCORE (:842,854; retry rule :856), its decision (:859-865) -- and returns the finished region fields.  Inside
 * the library that is four dependent rounds of ksw_extend2 batches (left, left at 2w, right, right at 2w), the
 * retry lists and the right tasks built ON THE DEVICE from the previous round's results; nothing returns to the host
 * in between, the whole call is asynchronous on the context's stream.
 * ext_param_t gave the flanks as four pointers + lengths; here they are expressed by the seed's position inside the
 * read and inside the chain's reference window [rmax0,rmax1) (bwamem.c:740-757), which is what the driver has. */
typedef struct bmh_seed_task { /* 40 bytes */
	uint64_t q_off;   /* pool offset of query base 0 (the whole read, base codes)                                  */
	uint64_t t_off;   /* pool offset of rseq[0], i.e. of window base rmax0; with BMH_F_TPAC the doubled-coordinate
	                     position rmax0 itself (the window is then read from the resident 2-bit reference)          */
	int32_t l_query;  /* read length                                                                                */
	int32_t qbeg, len;/* the seed on the query: mem_seed_t.qbeg / .len (bwamem.c:168-171); h0 = len * a            */
	int32_t rbeg;     /* the seed inside the window: mem_seed_t.rbeg - rmax0  (left target length, bwamem.c:815)    */
	int32_t wlen;     /* rmax1 - rmax0                                                                              */
	uint16_t flags;   /* BMH_F_TPAC or 0                                                                            */
	uint16_t rsv_;
} bmh_seed_task_t;

typedef struct bmh_seed_result { /* 32 bytes; ext_res_t (bwamem.c:568-577) in 32-bit fields */
	int32_t qb, qe;        /* mem_alnreg_t.qb / .qe                                    (bwamem.c:832,835,860,863,866) */
	int32_t rb, re;        /* mem_alnreg_t.rb / .re MINUS rmax0 (window-relative)                                     */
	int32_t score, truesc; /* mem_alnreg_t.score / .truesc                                                            */
	int32_t w;             /* max(aw[0], aw[1]), bwamem.c:875                                                         */
	int32_t n_ext;         /* ksw_extend2 calls the reference would have made for this seed (0..4)                    */
} bmh_seed_result_t;

/* Range: qbeg, l_query-qbeg-len, rbeg, wlen-rbeg-len <= 65535, 2*w <= 32767, and the extension limits above. */
int bmh_seedext_batch(bmh_ctx_t *ctx, const uint8_t *seqpool, size_t pool_bytes, const bmh_seed_task_t *tasks, int64_t n,
                      bmh_seed_result_t *results); /* pool == NULL: the pool left by bmh_upload_pool() */
int bmh_seedext_batch_device(bmh_ctx_t *ctx, const uint8_t *d_seqpool, const bmh_seed_task_t *d_tasks, int64_t n,
                             bmh_seed_result_t *d_results);
/* Two-step form of bmh_seedext_batch for callers that overlap host work with the device: _submit stages the tasks and
 * enqueues everything (uploads, the four rounds, the download into the context's pinned buffer) and returns at once;
 * _wait blocks until the results are there and copies them out.  One submission may be in flight per context. */
int bmh_seedext_submit(bmh_ctx_t *ctx, const bmh_seed_task_t *tasks, int64_t n);
int bmh_seedext_wait(bmh_ctx_t *ctx, bmh_seed_result_t *results);
typedef struct bmh_seedext_stats { /* of the last completed bmh_seedext_batch / _wait (host-buffer forms only) */
	int64_t seeds, left_tasks, left_retries, right_tasks, right_retries;
} bmh_seedext_stats_t;
int bmh_seedext_stats(const bmh_ctx_t *ctx, bmh_seedext_stats_t *st);

/* ---- kernel timing: HIP events recorded on the context's stream around the
 * dominant kernel of the last *_device call.  ms < 0 if none. */
int bmh_last_kernel_ms(bmh_ctx_t *ctx, float *ms);
int bmh_set_kernel_timing(bmh_ctx_t *ctx, int enable);
/* Per-kernel duration of the last extension launch, one entry per query-length bin of the dispatcher:
 * qlen <= 32, <= 64, <= 128, <= 256, <= 512, longer (LDS kernel).  -1 when timing was off. */
int bmh_last_extend_bin_ms(bmh_ctx_t *ctx, float ms[6]);
/* With timing on, every dispatcher launch (a fused per-seed call makes four) waits for its kernels and adds their per-bin durations
 * to running sums: ms[b] = the sum for bin b, *launches (nullable) = dispatcher launches counted; reset != 0 clears the sums. */
int bmh_extend_bin_ms_sum(bmh_ctx_t *ctx, double ms[6], long long *launches, int reset);
/* Per-kernel duration of the last global-alignment launch: the 64-slot lane kernel (w <= 31), the 128-slot one (w <= 63),
 * the one-wave-per-task kernel (everything else).  -1 when timing was off. */
int bmh_last_global_bin_ms(bmh_ctx_t *ctx, float ms[3]);
/* Duration of the four rounds of the last fused per-seed launch (left, left at 2w, right, right at 2w), each with the
 * small list-building kernel in front of it.  -1 when timing was off. */
int bmh_last_seedext_round_ms(bmh_ctx_t *ctx, float ms[4]);

/* ---- L3: data carriers of the extension driver, layout-compatible with the
 * reference so that its structs can be passed straight through. */
typedef struct bmh_seed {  /* == mem_seed_t, bwamem.c:168-171 */
	int64_t rbeg;
	int32_t qbeg, len;
} bmh_seed_t;

typedef struct bmh_chain { /* == mem_chain_t, bwamem.c:173-177 */
	int32_t n, m;
	int64_t pos;
	bmh_seed_t *seeds;
} bmh_chain_t;

typedef struct bmh_chain_v { /* == mem_chain_v, bwamem.c:179 */
	size_t n, m;
	bmh_chain_t *a;
} bmh_chain_v;

typedef struct bmh_alnreg { /* == mem_alnreg_t, bwamem.h:50-62 (64 bytes) */
	int64_t rb, re;
	int32_t qb, qe;
	int32_t score, truesc, sub, csub, sub_n, w, seedcov, secondary;
	uint64_t hash;
} bmh_alnreg_t;

typedef struct bmh_alnreg_v { /* == mem_alnreg_v, bwamem.h:64; a is malloc/realloc'd */
	size_t n, m;
	bmh_alnreg_t *a;
} bmh_alnreg_v;

typedef struct bmh_read { /* the two bseq1_t fields the driver reads (bwa.h:18-22) */
	int32_t l_seq;
	const uint8_t *seq; /* base codes 0..4 (after bwamem.c:1093-1094) */
} bmh_read_t;

/* Optional per-chain pre-step, called right before chain `chain` of read `read` would be
 * extended, in the reference's order.  It mirrors `ret = mem_chain2aln_short(...)`
 * (bwamem.c:1104 / :1140): it may append to *av; a return value <= 0 means the chain is
 * done and must not be extended, > 0 means "run mem_chain2aln on it" (bwamem.c:1105). */
typedef int (*bmh_chain_pre_fn)(void *user, int read, int chain, bmh_alnreg_v *av);

/* Replaces, for `n_reads` reads at once, the loop
 *     for each chain c of read r: [pre-step;] mem_chain2aln(opt,l_pac,pac,l_seq,seq,c,&regs[r])
 * (bwamem.c:1101-1107 / :1136-1143).  Regions are APPENDED to regs[r] (kv_pushp semantics,
 * bwamem.c:804; regs[r].a must be NULL or malloc'd), so entries already there take part in
 * the containment test of bwamem.c:769-802.  `pre` may be NULL (every chain is extended). */
int bmh_chain2aln_batch(bmh_ctx_t *ctx, int64_t l_pac, const uint8_t *pac, int n_reads,
                        const bmh_read_t *reads, const bmh_chain_v *chains, bmh_chain_pre_fn pre,
                        void *pre_user, bmh_alnreg_v *regs);

/* The whole loop body of mem_align1_core (bwamem.c:1101-1107 / :1136-1143) for n_reads reads: per chain first
 * mem_chain2aln_short (bwamem.c:495-542) -- its qualifying tests on the host, the ksw_align2 calls of all chains of the
 * batch as ONE bmh_sw_batch, its verdict folded in where the reference calls it -- then, unless that settled the chain,
 * mem_chain2aln as in bmh_chain2aln_batch.  min_seed_len = opt->min_seed_len (the XSUBO threshold, bwamem.c:529). */
int bmh_chains2regs_batch(bmh_ctx_t *ctx, int64_t l_pac, const uint8_t *pac, int n_reads, const bmh_read_t *reads,
                          const bmh_chain_v *chains, int min_seed_len, bmh_alnreg_v *regs);

/* ---- L3, phase 2: the caller of ksw_global2.
 * One request = one alignment region after bwa_fix_xref2, as mem_reg2aln sees it at bwamem.c:1187. */
typedef struct bmh_cigar_req {
	int32_t read;    /* index into reads[]                                              */
	int32_t qb, qe;  /* query interval [qb,qe)                    (mem_alnreg_t.qb/qe)   */
	int64_t rb, re;  /* reference interval, doubled coordinate    (mem_alnreg_t.rb/re)   */
	int32_t truesc;  /* mem_alnreg_t.truesc: band inference + retry test, bwamem.c:1187,1201;
	                    INT32_MIN = ONE bwa_gen_cigar2 call with w_ = reg_w instead of that loop (bwa_fix_xref2, bwa.c:198) */
	int32_t reg_w;   /* mem_alnreg_t.w:      cap of the inferred band, bwamem.c:1191     */
} bmh_cigar_req_t;

typedef struct bmh_cigar_res {
	int32_t score;      /* global alignment score of the last try (bwa_gen_cigar2 *score)   */
	int32_t n_cigar;    /* number of CIGAR words                                             */
	int32_t NM;         /* edit distance, bwa.c:162; -1 if the request was rejected (bwa.c:99) */
	int32_t tries;      /* bwa_gen_cigar2 calls the reference would have made (1..3)          */
	uint32_t cigar_off; /* first word in cigar_pool                                           */
	uint32_t md_off;    /* first byte of the NUL-terminated MD string in md_pool              */
	uint32_t md_len;    /* strlen(MD)                                                         */
	uint32_t rsv_;
} bmh_cigar_res_t;

/* Replaces, for n_req regions at once, the loop of mem_reg2aln (bwamem.c:1187-1201) over
 * bwa_gen_cigar2 (bwa.c:89-172): band inference, up to three banded global alignments per region
 * (one GPU batch per try), NM and MD.  The reference appends the MD string to the CIGAR buffer
 * (bwa.c:136,161-163); here both are returned separately -- INTEGRATION.md shows the two-line glue.
 * Pools: cigar_words >= sum(qlen+tlen+2), md_bytes >= sum(3*(qlen+tlen)+16) always suffice. */
int bmh_reg2cigar_batch(bmh_ctx_t *ctx, int64_t l_pac, const uint8_t *pac, const bmh_read_t *reads, int64_t n_req,
                        const bmh_cigar_req_t *reqs, bmh_cigar_res_t *results, uint32_t *cigar_pool,
                        size_t cigar_words, char *md_pool, size_t md_bytes);

/* ---- L2.6: one record per REGION -- everything bwa_gen_cigar2 (bwa.c:89-172) does with sequence bytes, and the replay of mem_reg2aln's
 * band loop (bwamem.c:1193-1201), on the device, against the 2-bit reference made resident by bmh_ctx_set_pac():
 *   - bns_get_seq of [rb, rb+tl) on the doubled coordinate and the reversal of query and window for reverse-strand hits (bwa.c:100-107):
 *     the oriented copies are written into a device pool, query k at o_off, its window right behind it at o_off + ql;
 *   - the no-gap shortcut's score, sum of mat[t*5+q] (bwa.c:108-114), for a region whose task[0] is -1;
 *   - ksw_global2 + traceback of `tasks` (their q_off / t_off address that oriented pool; cigar_off / cigar_cap a device-side scratch
 *     of `task_cigar_words` words);
 *   - per region the loop over its tries task[0..2]: stop when the score repeats or reaches truesc - a, or after one try when
 *     truesc == INT32_MIN (bwamem.c:1194-1201; equal bands share a task index);
 *   - NM and MD of the final try (bwa.c:134-164), MD with "ACGTN" or, on the reverse strand, "TGCAN".
 * The host keeps what needs no sequence: band inference and the band of each try (bwamem.c:884-891,1187-1191, bwa.c:116-125).
 * Returned per region: the record below, its CIGAR words at cigar_out[k*cig_cap ..] and its MD bytes (no NUL) at md_out[k*md_cap ..].
 * A CIGAR longer than the final task's cigar_cap or cig_cap, or an MD longer than md_cap, is flagged and not returned (the caller
 * redoes those few regions with full capacities, e.g. through bmh_global_batch); score, n_cigar and tries are exact either way. */
#define BMH_REGION_CIGAR_CUT 1u /* n_cigar exceeds the capacity: CIGAR, NM and MD not returned */
#define BMH_REGION_MD_CUT 2u    /* MD longer than md_cap: NM is exact, the MD bytes are not returned */
typedef struct bmh_region_req { /* 48 bytes */
	uint64_t q_src;  /* offset in readpool of query base qb, read orientation, one base code per byte */
	int64_t rb;      /* doubled coordinate of the window's first base; [rb, rb+tl) must not bridge l_pac (bwa.c:99) */
	uint64_t o_off;  /* where the oriented query goes in the device pool (the window follows at o_off + ql) */
	int32_t ql, tl;
	int32_t truesc;  /* INT32_MIN: one try */
	int32_t task[3]; /* index into tasks[] of try 0..2; -1 = no such try; task[0] == -1: the no-gap case */
} bmh_region_req_t;
typedef struct bmh_region_res { /* 24 bytes */
	int32_t score, n_cigar, tries, NM;
	int32_t md_len;
	uint32_t flags; /* BMH_REGION_* */
} bmh_region_res_t;
/* Bytes of a region's MD slot past md_len are unspecified.  readpool: the query windows; opool_bytes: size of the oriented pool the o_off / q_off / t_off fields address.  BMH_E_ARG without a
 * resident reference or with offsets outside the pools. */
int bmh_region_cigar_batch(bmh_ctx_t *ctx, const uint8_t *readpool, size_t readpool_bytes, size_t opool_bytes,
                           const bmh_region_req_t *reqs, int64_t n_req, const bmh_glb_task_t *tasks, int64_t n_tasks,
                           size_t task_cigar_words, int cig_cap, int md_cap, bmh_region_res_t *results, uint32_t *cigar_out,
                           char *md_out);

/* Counters of the last bmh_chain2aln_batch call (for the bench / logs). */
typedef struct bmh_driver_stats {
	int64_t rounds, ext_tasks, seeds_extended, seeds_skipped;
	int64_t pool_bytes; /* bytes of sequence shipped to the device for the call */
	int64_t seeds_speculated; /* seeds the device extended ahead of the replay whose result was not needed */
	int64_t short_sw;         /* ksw_align2 calls of mem_chain2aln_short run as one batch (bmh_chains2regs_batch) */
} bmh_driver_stats_t;
int bmh_driver_stats(const bmh_ctx_t *ctx, bmh_driver_stats_t *st);

/* The host-buffer entry points stage transfers of up to 64 MB per direction through pinned buffers owned by the context
 * (they grow on demand).  A caller that creates its contexts ahead of time -- the preload shim does, while the host
 * program is still loading its index -- can reserve them here, so that the first batch does not pay for the
 * allocation.  No counterpart in the reference (it has no device). */
int bmh_ctx_reserve_staging(bmh_ctx_t *ctx, size_t upload_bytes, size_t download_bytes);
/* The same for the device-side workspaces of the host-buffer entry points (sequence pool, task / result records, CIGAR
 * words, the dispatcher's sort lists): growing one of them in the middle of a run frees and re-allocates device memory,
 * which synchronises the device under every other host thread's batch. */
int bmh_ctx_reserve_device(bmh_ctx_t *ctx, size_t pool_bytes, int64_t max_tasks, size_t cigar_words);
/* ... and the two large kernel workspaces, which otherwise grow inside a context's first batches (a hipMalloc of hundreds
 * of megabytes under every other thread's work): the SMEM kernels' interval stacks for batches of up to seed_reads reads of
 * up to seed_read_len bases, and the ksw_global2 lane kernels' direction slab for batches of up to global_tasks tasks with
 * targets of up to global_rows rows.  0 skips either. */
int bmh_ctx_reserve_kernels(bmh_ctx_t *ctx, int seed_reads, int seed_read_len, int64_t global_tasks, int global_rows);

/* ------------------------------------------------------------------------------------------------------------
 * Local Smith-Waterman for mate rescue and short chains.
 * Replaces: kswr_t ksw_align2(qlen, query, tlen, target, m, mat, o_del, e_del, o_ins, e_ins, xtra, qry)
 *           reference bwa-0.7.8/ksw.h:62, ksw.c:341-364 (over ksw_qinit / ksw_u8 / ksw_i16, ksw.c:62-331);
 *           callers mem_matesw (bwamem_pair.c:109-175) and mem_chain2aln_short (bwamem.c:495-542).
 * One record per call; `xtra` is passed through unchanged (KSW_XBYTE/XSTOP/XSUBO/XSTART | threshold, ksw.h:6-9)
 * and the result is field for field kswr_t (ksw.h:14-19), including the byte-mode (16 columns per vector) versus
 * word-mode (8) differences of the striped reference, which are visible in score2/te2 and, rarely, the scores.
 * m = 5 and the matrix / gap penalties come from bmh_params_t.  Inputs outside the exact domain are refused with
 * BMH_E_RANGE: qlen*max(mat) >= 32000, and byte-mode tasks that can overflow (qlen*max(mat) + shift >= 255) combined
 * with KSW_XSTART, for which the reference reads uninitialised memory (ksw.c:355-357). */
#define BMH_SW_XBYTE 0x10000u
#define BMH_SW_XSTOP 0x20000u
#define BMH_SW_XSUBO 0x40000u
#define BMH_SW_XSTART 0x80000u
/* query base = complement of the pool byte (3-b; N stays N).  With BMH_F_QREV the query is the reverse complement
 * of the stored mate, so the host needs no second copy of it (reference bwamem_pair.c:130-133). */
#define BMH_F_QCOMP 8u

typedef struct bmh_sw_task {
	uint64_t q_off;  /* query base k = pool[q_off + k], or pool[q_off - k] with BMH_F_QREV */
	uint64_t t_off;  /* target likewise (BMH_F_TREV); with BMH_F_TPAC a doubled-coordinate position */
	uint32_t tlen;
	uint16_t qlen;
	uint16_t flags;  /* BMH_F_QREV | BMH_F_TREV | BMH_F_TPAC | BMH_F_QCOMP */
	uint32_t xtra;   /* ksw_align2's xtra */
	uint32_t rsv;
} bmh_sw_task_t; /* 32 bytes */

typedef struct bmh_sw_result {
	int32_t score, te, qe, score2, te2, tb, qb; /* kswr_t, ksw.h:14-19; unset values are -1 */
	int32_t rsv;
} bmh_sw_result_t; /* 32 bytes */

int bmh_sw_batch(bmh_ctx_t *ctx, const uint8_t *pool, size_t pool_bytes, const bmh_sw_task_t *tasks, int64_t n,
                 bmh_sw_result_t *results);
/* device-resident variant: nothing crosses PCIe; completion is ordered on the context's stream */
int bmh_sw_batch_device(bmh_ctx_t *ctx, const uint8_t *d_pool, const bmh_sw_task_t *d_tasks, int64_t n,
                        bmh_sw_result_t *d_results);

/* ------------------------------------------------------------------------------------------------------------
 * Batched mate rescue: the loop mem_sam_pe runs per pair (reference bwamem_pair.c:251-263) over mem_matesw
 * (bwamem_pair.c:109-175), for a whole chunk of pairs.
 *   reads[2p], reads[2p+1]   the two mates of pair p (base codes), regs[2p], regs[2p+1] their region vectors as
 *                            phase 1 left them (mem_alnreg_v, layout-compatible); rescued regions are inserted into
 *                            the MATE's vector exactly as the reference does (kv_push growth, score-sorted insert,
 *                            then mem_sort_and_dedup)
 *   pes[4]                   mem_pestat_t per orientation (bwamem.h:66-70), opt fields of mem_opt_t in `o`
 *   dedup                    the caller's mem_sort_and_dedup(n, a, opt->mask_level_redun) (bwamem.c:395); it is host
 *                            post-processing outside this path, so the reference's own function is passed in
 *   n_sw[p] (nullable)       the sum of mem_matesw's return values for pair p
 * Each mem_matesw invocation tests its four orientations against the current state of the mate's vector, so the
 * invocations of one pair are sequential; the driver runs them as rounds -- per unfinished pair its next few
 * invocations (planned ahead against the current state, re-checked when folded), their ksw_align2 calls as one GPU
 * batch over all pairs.  The outcome is exactly the reference's; a call planned ahead may go unused.
 * With the reference resident (bmh_ctx_set_pac, same pac pointer) the windows are BMH_F_TPAC tasks and the
 * reverse-complemented mate is BMH_F_QREV|BMH_F_QCOMP: the pool holds every read once and nothing else. */
typedef struct bmh_pestat { /* mem_pestat_t, bwamem.h:66-70 */
	int32_t low, high;
	int32_t failed;
	double avg, std;
} bmh_pestat_t;
typedef struct bmh_matesw_opt { /* the mem_opt_t fields mem_sam_pe / mem_matesw read beside the scoring (bwamem.h:21-48) */
	int32_t pen_unpaired, max_matesw, min_seed_len, rsv;
} bmh_matesw_opt_t;
typedef int (*bmh_dedup_fn)(void *user, int n, bmh_alnreg_t *a);
int bmh_matesw_batch(bmh_ctx_t *ctx, int64_t l_pac, const uint8_t *pac, int n_pairs, const bmh_read_t *reads,
                     bmh_alnreg_v *regs, const bmh_pestat_t pes[4], const bmh_matesw_opt_t *o, bmh_dedup_fn dedup,
                     void *dedup_user, int *n_sw);

/* ------------------------------------------------------------------------------------------------------------
 * Region post-processing and SAM text: everything mem_process_seqs does with a read's region
 * vector after the extensions -- host code (plain C, batched per chunk slice), with the one DP it contains
 * (ksw_global2 under mem_reg2aln / bwa_fix_xref2) run as GPU batches.
 * Replaces: mem_sort_and_dedup (reference bwamem.c:395-436), mem_mark_primary_se (:445-475), mem_approx_mapq_se
 *           (:1023-1047), mem_reg2aln (:1164-1236) over bwa_fix_xref2 (bwa.c:179-222), mem_aln2sam (:904-1017),
 *           mem_reg2sam_se (:1049-1083); mem_pestat (bwamem_pair.c:46-107), mem_pair (:177-238), mem_sam_pe (:240-332)
 *           -- i.e. worker2 of mem_process_seqs (bwamem.c:1281-1295).
 * Ties in the reference's unstable sorts decide which duplicate survives and which hit is primary; the library sorts
 * with the same comparison/exchange sequence (host/sort_exact.h), so the output order is the reference's. */
typedef struct bmh_sam_opt { /* the mem_opt_t fields phase 2 reads (bwamem.h:21-48); scoring as in bmh_params_t */
	int32_t a, b, o_del, e_del, o_ins, e_ins, pen_unpaired, w;
	int32_t T, flag, min_seed_len, max_ins, mapQ_coef_fac, max_matesw;
	float mask_level, mask_level_redun, mapQ_coef_len;
	int8_t mat[25];
	int8_t pad_[3];
} bmh_sam_opt_t;
#define BMH_MEM_F_PE 0x2        /* bwamem.h:14-20 */
#define BMH_MEM_F_NOPAIRING 0x4
#define BMH_MEM_F_ALL 0x8
#define BMH_MEM_F_NO_MULTI 0x10
#define BMH_MEM_F_NO_RESCUE 0x20

typedef struct bmh_refann { /* == bntann1_t, bntseq.h:39-45: one reference sequence */
	int64_t offset;
	int32_t len, n_ambs;
	uint32_t gi;
	char *name, *anno;
} bmh_refann_t;
typedef struct bmh_refidx { /* == the head of bntseq_t, bntseq.h:53-61 (a bntseq_t* may be passed) */
	int64_t l_pac;
	int32_t n_seqs;
	uint32_t seed;
	bmh_refann_t *anns;
} bmh_refidx_t;
typedef struct bmh_seq { /* == bseq1_t, bwa.h:19-22; seq holds base codes (after bwamem.c:1093-1094); sam is malloc'd */
	int32_t l_seq;
	char *name, *comment, *seq, *qual, *sam;
} bmh_seq_t;

int bmh_sort_and_dedup(int n, bmh_alnreg_t *a, float mask_level_redun);                       /* mem_sort_and_dedup  */
void bmh_mark_primary_se(const bmh_sam_opt_t *o, int n, bmh_alnreg_t *a, int64_t id);         /* mem_mark_primary_se */
int bmh_approx_mapq_se(const bmh_sam_opt_t *o, const bmh_alnreg_t *a);                         /* mem_approx_mapq_se  */
/* mem_pestat; with verbose >= 3 it prints the reference's "[M::mem_pestat] ..." lines to stderr */
void bmh_pestat(const bmh_sam_opt_t *o, int64_t l_pac, int n, const bmh_alnreg_v *regs, bmh_pestat_t pes[4], int verbose);
/* mem_pair (bwamem_pair.c:177-238): the best-scoring properly oriented pair of hits of the two ends; returns its score
 * (0 = none), *sub / *n_sub the runner-up and how many lie within one event of it, z[] the chosen hit of each end */
int bmh_pair(const bmh_sam_opt_t *o, int64_t l_pac, const bmh_pestat_t pes[4], const bmh_alnreg_v a[2], uint64_t id, int *sub,
             int *n_sub, int z[2]);
/* Phase 2 for reads [0,n) of a chunk slice: marks primaries, pairs (when o->flag has BMH_MEM_F_PE: n is even, reads 2p
 * and 2p+1 are mates, pes = the four insert-size models, mate rescue already done or off), runs the global alignments
 * of exactly the regions that get printed as GPU batches, and writes seqs[i].sam (malloc'd, the caller frees it as
 * fastmap.c:201 does).  id0 = n_processed + index of read 0 (the reference's per-read id, bwamem.c:1287,1291).
 * rg_id: the reference's bwa_rg_id ("" = none).  regs[i].a is left sorted/marked as the reference leaves it. */
int bmh_sam_batch(bmh_ctx_t *ctx, const bmh_sam_opt_t *o, const bmh_refidx_t *bns, const uint8_t *pac, const bmh_pestat_t *pes,
                  int64_t id0, int n, bmh_seq_t *seqs, bmh_alnreg_v *regs, const char *rg_id);

/* ------------------------------------------------------------------------------------------------------------
 * FM-index queries of the seeding stage (first slice): super-maximal exact matches and suffix-
 * array look-ups on the device, over the reference's own index arrays made resident in HBM.
 * Replaces: bwt_smem1 (reference bwa-0.7.8/bwt.c:288-347; over bwt_extend :261-274 and bwt_2occ4 :191-219), called in the
 *           order smem_next2 calls it for one read (bwamem.c:118-162 as driven by mem_insert_seed, bwamem.c:208-214), and
 *           bwt_sa (bwt.c:85-95; over bwt_invPsi :52-58 and bwt_occ :107-129).
 * Chaining (mem_insert_seed's B-tree, mem_chain_flt) stays host code; INTEGRATION.md shows how the reference's
 * mem_chain consumes these results unchanged. */
typedef struct bmh_bwt { /* the fields of bwt_t the queries read (bwt.h:45-57); arrays are borrowed, never modified */
	uint64_t primary, L2[5], seq_len, bwt_size; /* bwt_size in 32-bit words */
	const uint32_t *bwt;                        /* BWT with the interleaved occurrence counts (bwt.h:63-64 layout) */
	int32_t sa_intv;
	uint64_t n_sa;
	const uint64_t *sa;
} bmh_bwt_t;
typedef struct bmh_smem_intv { uint64_t x[3], info; } bmh_smem_intv_t; /* == bwtintv_t, bwt.h:59-61 */
typedef struct bmh_smem_opt {
	int32_t min_seed_len; /* mem_opt_t.min_seed_len (only used to bound the look-ups, not the calls)            */
	int32_t split_len;    /* (int)(min_seed_len*split_factor + .499), bwamem.c:211 (clamped to the read length there) */
	int32_t split_width;  /* mem_opt_t.split_width                                                              */
	int32_t start_width;  /* 2 with MEM_F_NO_EXACT, else 1, bwamem.c:212                                        */
	int32_t min_emit_len; /* 0: every interval of every call comes back, like bwt_smem1.  > 0: only intervals at least this
	                       * long do (call.n counts those) -- all that mem_insert_seed looks at with min_emit_len <=
	                       * min_seed_len (bwamem.c:219), about one interval in twenty on 150 bp reads; the calls made, their
	                       * arguments and return values are the same either way.  Must be >= 0; bmh_seed_batch and
	                       * bmh_chain_reads additionally need min_emit_len <= min_seed_len (chaining recomputes the longest
	                       * match and the re-seeding decision from the intervals it is given): BMH_E_ARG otherwise.
	                       * The struct is 20 bytes since BMH_VERSION 300 (16 before: rebuild callers)               */
} bmh_smem_opt_t;
typedef struct bmh_smem_call { /* one bwt_smem1 call and where its result intervals lie */
	int32_t x, min_intv; /* arguments (bwt.c:288)                         */
	int32_t ret, n;      /* return value and mem->n                       */
	uint32_t first;      /* first interval in the read's slice of the pool */
	uint32_t rsv;
} bmh_smem_call_t;

/* The index files of `bwa index` (<prefix>.bwt/.sa/.ann/.amb/.pac; reference bwt.c:380-421, bntseq.c:94-150, bwa.c:291)
 * read into plain arrays: what bmh_ctx_set_bwt / bmh_ctx_set_pac take.  Host I/O only, no GPU involved. */
typedef struct bmh_index {
	bmh_bwt_t bwt;    /* arrays owned by the index */
	int64_t l_pac;
	uint8_t *pac;     /* l_pac/4+1 bytes */
	int32_t n_seqs;   /* reference sequences (.ann): name, offset in the concatenation, length */
	char **names;
	int64_t *offsets;
	int32_t *lens;
	int32_t n_holes;  /* runs of ambiguous bases (.amb; bntamb1_t, bntseq.h:47-51): start, length, the letter */
	int64_t *hole_offsets;
	int32_t *hole_lens;
	char *hole_chars;
} bmh_index_t;
int bmh_index_load(const char *prefix, bmh_index_t **out); /* BMH_E_ARG if a file is missing or inconsistent */
void bmh_index_free(bmh_index_t *ix);

/* Make the index resident on the context's device (one copy per device and host array, shared by all contexts). */
int bmh_ctx_set_bwt(bmh_ctx_t *ctx, const bmh_bwt_t *bwt);
/* For every read: the bwt_smem1 calls of smem_next2's iteration, in order.  Read r's calls are
 * calls[call_off[r] .. call_off[r+1]) and their intervals lie in intv[intv_off[r] + call.first ...].  call_off and
 * intv_off have n_reads+1 entries.  BMH_E_CIGAR_CAP if a capacity is too small (nothing partial is returned). */
int bmh_smem_batch(bmh_ctx_t *ctx, const bmh_smem_opt_t *o, int n_reads, const bmh_read_t *reads, uint32_t *call_off,
                   bmh_smem_call_t *calls, size_t call_cap, uint64_t *intv_off, bmh_smem_intv_t *intv, size_t intv_cap);
/* bmh_smem_batch + the suffix-array look-ups chaining will make, in ONE device round trip: for every interval that comes
 * back and is long and rare enough to become seeds (length >= o->min_seed_len, x[2] <= max_occ; bwamem.c:218-225),
 * sa_off[k] = index of its first position in sa_pos (its x[2] positions follow one another: bwt_sa(x[0] + j)), for the
 * others UINT64_MAX -- what bmh_chain_sa_keys + bmh_sa_batch produce, in the form bmh_chain_reads takes (the runs lie in
 * sa_pos in no particular order).  sa_off has intv_cap entries.  *n_pos (nullable) = positions written.
 * BMH_E_CIGAR_CAP if a capacity is too small. */
int bmh_seed_batch(bmh_ctx_t *ctx, const bmh_smem_opt_t *o, int max_occ, int n_reads, const bmh_read_t *reads, uint32_t *call_off,
                   bmh_smem_call_t *calls, size_t call_cap, uint64_t *intv_off, bmh_smem_intv_t *intv, size_t intv_cap, uint64_t *sa_off,
                   uint64_t *sa_pos, size_t sa_cap, uint64_t *n_pos);
/* N x bwt_sa: pos[i] = position (doubled coordinate) of suffix-array entry k[i]. */
int bmh_sa_batch(bmh_ctx_t *ctx, const uint64_t *k, int64_t n, uint64_t *pos);

/* ------------------------------------------------------------------------------------------------------------
 * Seeds to chains (the rest of SURVEY.md §8(f) row 3): host code over the results of bmh_smem_batch / bmh_sa_batch.
 * Replaces: mem_chain (reference bwamem.c:283-306) = smem_next2's rounds (:118-157) + mem_insert_seed (:208-243, over
 *           test_and_merge :186-206 and klib's B-tree of chains) + the in-order read-out, followed by mem_chain_flt
 *           (:319-380) -- lines bwamem.c:1096-1097 of mem_align1_core_batched.
 * call_off / calls / intv_off / intv are bmh_smem_batch's outputs for the same reads.  The bwt_sa results come in
 * interval order: bmh_chain_sa_keys lists, for every interval that is long and rare enough (length >= min_seed_len,
 * x[2] <= max_occ), its suffix-array indices x[0]..x[0]+x[2]-1 and records in sa_off[k] where interval k's run starts
 * (UINT64_MAX = never looked up); the caller resolves the list with ONE bmh_sa_batch and passes the positions as sa_pos
 * -- no sorting, no searching.  chains[r] is filled like mem_chain's return value: a malloc'd array of chains in the
 * reference's order, each with a malloc'd seed array; the caller frees both.
 * Precondition: the intervals come from a bmh_smem_batch / bmh_seed_batch run with min_emit_len <= o->min_seed_len (0
 * included): with a larger filter the longest match of a call and the re-seeding decision derived from it would be
 * computed from a truncated list and chains would silently differ. */
typedef struct bmh_chain_opt { /* the mem_opt_t fields seeding + chaining read (bwamem.h:21-48) */
	int32_t w, max_chain_gap, min_seed_len, max_occ;
	int32_t split_len;   /* (int)(min_seed_len * split_factor + .499), bwamem.c:211 */
	int32_t split_width;
	float mask_level, chain_drop_ratio;
} bmh_chain_opt_t;
uint64_t bmh_chain_sa_keys(const bmh_chain_opt_t *o, uint64_t n_intv, const bmh_smem_intv_t *intv, uint64_t *sa_off /* [n_intv] */,
                           uint64_t *keys /* NULL: count only */);
int bmh_chain_reads(const bmh_chain_opt_t *o, int64_t l_pac, int n_reads, const bmh_read_t *reads, const uint32_t *call_off,
                    const bmh_smem_call_t *calls, const uint64_t *intv_off, const bmh_smem_intv_t *intv, const uint64_t *sa_off,
                    const uint64_t *sa_pos, bmh_chain_v *chains);

#ifdef __cplusplus
}
#endif
#endif
