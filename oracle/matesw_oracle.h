/*
 * oracle/matesw_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Sequential CPU restatement of mate rescue for ONE read pair:
 *   orc_matesw_pair  <->  the block of mem_sam_pe at reference bwa-0.7.8/bwamem_pair.c:251-263
 *                         over mem_matesw (bwamem_pair.c:109-175), mem_infer_dir (:23-30),
 *                         bns_get_seq (bntseq.c:355-376) and ksw_align2 (via orc_align2).
 * mem_sort_and_dedup (bwamem.c:395-436) is host post-processing outside the path; it is passed
 * in as a callback -- the reference's own function where the compiled reference is available (that is how this file
 * is pinned, tests/test_oracle_vs_ref.py), orc_simple_dedup elsewhere (DUT and oracle then share it).
 */
#ifndef ORC_MATESW_ORACLE_H
#define ORC_MATESW_ORACLE_H
#include <stdint.h>

#include "../include/bwamem_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* regs[0], regs[1]: the two ends' region vectors (in/out, realloc'd like kvec).  Returns the sum of mem_matesw's
 * return values. */
int orc_matesw_pair(const bmh_params_t *p, const bmh_matesw_opt_t *o, int64_t l_pac, const uint8_t *pac,
                    const bmh_pestat_t pes[4], const bmh_read_t reads[2], bmh_alnreg_v regs[2], bmh_dedup_fn dedup,
                    void *user);

/* A deterministic stand-in for mem_sort_and_dedup where the reference is not available: stable order by
 * (score desc, rb, qb), exact duplicates of (score, rb, qb) dropped.  NOT the reference's function. */
int orc_simple_dedup(void *user, int n, bmh_alnreg_t *a);

#ifdef __cplusplus
}
#endif
#endif
