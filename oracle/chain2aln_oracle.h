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

#ifdef __cplusplus
}
#endif
#endif
