/*
 * oracle/fmindex_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the FM-index queries BWA-MEM's seeding is made of:
 *   orc_bwt_occ4    <->  bwt_occ4 / bwt_2occ4   (reference bwa-0.7.8/bwt.c:159-219)
 *   orc_bwt_extend  <->  bwt_extend             (bwt.c:261-274)
 *   orc_bwt_smem1   <->  bwt_smem1              (bwt.c:288-347)
 *   orc_bwt_sa      <->  bwt_sa, bwt_invPsi, bwt_occ  (bwt.c:52-58, 85-95, 107-129)
 *   orc_smem_calls  <->  the sequence of bwt_smem1 calls smem_next2 makes for one read (bwamem.c:118-162), as
 *                        mem_insert_seed drives it (bwamem.c:208-214)
 * Counting is restated with bit arithmetic on the packed BWT words (no lookup table); results are pinned against the
 * reference's own functions on a real index (tests/test_oracle_vs_ref.py) and by tests/golden/fmindex_golden.npz.
 */
#ifndef ORC_FMINDEX_ORACLE_H
#define ORC_FMINDEX_ORACLE_H
#include <stdint.h>

#include "../include/bwamem_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

void orc_bwt_occ4(const bmh_bwt_t *b, uint64_t k, uint64_t cnt[4]);
void orc_bwt_extend(const bmh_bwt_t *b, const bmh_smem_intv_t *ik, bmh_smem_intv_t ok[4], int is_back);
/* mem must have room for len+1 intervals; returns the next start (bwt.c:319) and *n_mem */
int orc_bwt_smem1(const bmh_bwt_t *b, int len, const uint8_t *q, int x, int min_intv, bmh_smem_intv_t *mem, int *n_mem);
uint64_t orc_bwt_sa(const bmh_bwt_t *b, uint64_t k);
uint64_t orc_fm_extends(int reset); /* number of orc_bwt_extend calls so far (bench statistics; not thread-safe) */
/* call log of one read: calls[c] = {x, min_intv, ret, n, first} with intervals in pool[first .. first+n).
 * Returns the number of calls, or -1 if a capacity is too small. */
int orc_smem_calls(const bmh_bwt_t *b, const bmh_smem_opt_t *o, int len, const uint8_t *q, bmh_smem_call_t *calls,
                   int call_cap, bmh_smem_intv_t *pool, int pool_cap, int *pool_used);

#ifdef __cplusplus
}
#endif
#endif
