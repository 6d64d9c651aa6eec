/* tls_ctx.h -- one lazily created GPU context per host thread for the reference-side shims
 * (the reference calls phase 1 from n_threads pthreads on disjoint reads, bwamem.c:1313). */
#ifndef BMH_TLS_CTX_H
#define BMH_TLS_CTX_H
#include "../../include/bwamem_hip.h"

/* context of the calling thread with `p` installed (device = $BMH_DEVICE, default 0); aborts on failure */
bmh_ctx_t *bmh_tls_ctx(const bmh_params_t *p);
/* slot 0: the batched seam (interpose.c); slot 1: the per-call drop-ins, which the reference may call from INSIDE a
 * batched call (mem_chain2aln_short -> ksw_align2 runs in the driver's pre-callback) and must not disturb it */
bmh_ctx_t *bmh_tls_ctx_slot(const bmh_params_t *p, int slot);
void bmh_tls_die(const char *msg, int code);
#endif
