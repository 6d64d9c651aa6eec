/* oracle/gencigar_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See gencigar_oracle.c. */
#ifndef ORC_GENCIGAR_ORACLE_H
#define ORC_GENCIGAR_ORACLE_H
#include "../include/bwamem_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef struct {
	int score, n_cigar, NM, w_used;
	uint32_t *cigar; /* malloc'd, BAM encoding */
	char *md;        /* malloc'd, NUL terminated */
} orc_cigar_t;
int orc_infer_bw(int l1, int l2, int score, int a, int q, int r);
void orc_gen_cigar(const bmh_params_t *p, int w_, int64_t l_pac, const uint8_t *pac, int l_query, const uint8_t *query,
                   int64_t rb, int64_t re, orc_cigar_t *out);
void orc_reg2cigar(const bmh_params_t *p, int64_t l_pac, const uint8_t *pac, const uint8_t *read, int qb, int qe, int64_t rb,
                   int64_t re, int truesc, int reg_w, orc_cigar_t *out, int *rounds);
void orc_cigar_free(orc_cigar_t *c);
#ifdef __cplusplus
}
#endif
#endif
