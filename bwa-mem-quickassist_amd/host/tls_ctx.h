/* tls_ctx.h -- GPU contexts for the reference-side shims.
 * The reference runs phase 1 / phase 2 on n_threads pthreads that are created and joined PER CHUNK (kthread.c:51-52),
 * so thread-local contexts would be created and lost once per chunk and thread.  Contexts therefore live in a process-
 * wide pool: a caller takes one for the duration of a call and gives it back.  That also makes the shims re-entrant:
 * the reference calls ksw_align2 (via mem_chain2aln_short) from INSIDE the batched phase-1 call, and simply gets a
 * second context. */
#ifndef BMH_TLS_CTX_H
#define BMH_TLS_CTX_H
#include "../../include/bwamem_hip.h"

/* an idle context with `p` installed (device = $BMH_DEVICE, default 0); aborts on failure */
bmh_ctx_t *bmh_pool_get(const bmh_params_t *p);
void bmh_pool_put(bmh_ctx_t *ctx);
void bmh_pool_prewarm(int n);
void bmh_pool_stop(void); /* makes a running bmh_pool_prewarm return after the context it is creating */
void bmh_tls_die(const char *msg, int code);
#endif
