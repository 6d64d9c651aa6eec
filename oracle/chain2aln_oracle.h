/*
 * oracle/chain2aln_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * CPU restatement of mem_chain2aln (bwa-0.7.8/bwamem.c:730-878).  See the .c.
 * Uses the layout-compatible carriers of include/bwamem_hip.h so the same
 * buffers can be handed to the reference, the oracle and the GPU driver.
 */
#ifndef ORC_CHAIN2ALN_ORACLE_H
#define ORC_CHAIN2ALN_ORACLE_H

#include "../include/bwamem_hip.h"
#include "ksw_oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
	int64_t ext_calls, seeds_extended, seeds_skipped;
} orc_driver_trace_t;

int orc_cal_max_gap(const bmh_params_t *p, int qlen);
uint8_t *orc_get_seq(int64_t l_pac, const uint8_t *pac, int64_t beg, int64_t end, int64_t *len);
void orc_chain2aln(const bmh_params_t *p, int64_t l_pac, const uint8_t *pac, int l_query,
                   const uint8_t *query, const bmh_chain_t *c, bmh_alnreg_v *av,
                   orc_driver_trace_t *trace /* nullable */);

/* one seed: bwamem.c:808-866 (the fused record of SURVEY.md §8 row a5); window-relative rb/re */
void orc_seedext_one(const bmh_params_t *p, const uint8_t *query, int l_query, const uint8_t *rseq, int wlen, int qbeg, int len,
                     int rbeg, bmh_seed_result_t *r, int64_t *cells /* nullable, += */, int64_t *calls /* nullable, += */);
int orc_seedext_batch(const bmh_params_t *p, const uint8_t *pool, const uint8_t *pac, int64_t l_pac, const bmh_seed_task_t *tasks,
                      int64_t n, bmh_seed_result_t *results, int64_t *cells_out, int64_t *calls_out, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
