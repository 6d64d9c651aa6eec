/*
 * oracle/sw_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of the local Smith-Waterman used for mate rescue and short chains
 * (SURVEY.md §8(f) row 2):
 *
 *   orc_align2  <->  ksw_align2   (reference: bwa-0.7.8/ksw.c:341-364)
 *                    over ksw_qinit / ksw_u8 / ksw_i16 (ksw.c:62-112, 114-233, 235-331)
 *
 * The reference computes the DP with Farrar's striped SSE2 layout.  Its RESULTS depend on that layout in
 * three places, all restated here without any SIMD:
 *   (1) the query is padded to p*slen columns (p = 16 bytes or 8 words per vector, slen = ceil(qlen/p));
 *       pad columns score 0 against every base and DO take part in the row maximum (ksw.c:98,107,152);
 *   (2) E(i+1,j) is derived from H before the lazy-F correction (ksw.c:155-158 precede :165-176), i.e. from
 *       max(M, E, F restricted to gaps opened inside the column's own segment [k*slen,(k+1)*slen));
 *   (3) ties for the query end resolve to the smallest query index (ksw.c:206-208).
 * The final H of a row equals the textbook affine-gap H given those E values (the lazy-F loop is exact).
 *
 * Parity status: PINNED against the reference itself compiled in-container (oracle/_ref/libksw_ref.so) by
 * tests/test_oracle_vs_ref.py and by the committed fixture tests/golden/sw_golden.npz.
 */
#ifndef ORC_SW_ORACLE_H
#define ORC_SW_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_XBYTE 0x10000  /* ksw.h:6-9 */
#define ORC_XSTOP 0x20000
#define ORC_XSUBO 0x40000
#define ORC_XSTART 0x80000

typedef struct { /* field for field kswr_t, ksw.h:14-19 */
	int score;
	int te, qe;
	int score2, te2;
	int tb, qb;
} orc_kswr_t;

/* ksw_align2 with qry == NULL.  Sequences are NOT modified (the reference reverses them in place and restores
 * them, ksw.c:355,358).  *undefined (nullable) is set when the reference's behaviour is undefined for the input:
 * byte mode overflowed (score 255, qe -1) and KSW_XSTART asks for a second pass over a zero-length query
 * (ksw.c:355-357 read H0[-1]). */
orc_kswr_t orc_align2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                      int o_del, int e_del, int o_ins, int e_ins, int xtra, int *undefined, int64_t *cells);

struct bmh_sw_task;
struct bmh_sw_result;
struct bmh_params;
/* Batch helper over bmh_sw_task_t / bmh_sw_result_t records (include/bwamem_hip.h); pac/l_pac serve BMH_F_TPAC. */
int orc_sw_batch(const struct bmh_params *p, const uint8_t *pool, const uint8_t *pac, int64_t l_pac,
                 const struct bmh_sw_task *tasks, int n, struct bmh_sw_result *res, int64_t *cells_out, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
