/*
 * oracle/gencigar_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the caller of ksw_global2 on BWA-MEM's path:
 *   orc_gen_cigar   <->  bwa_gen_cigar2   reference bwa-0.7.8/bwa.c:89-172
 *   orc_infer_bw    <->  infer_bw         reference bwa-0.7.8/bwamem.c:884-891
 *   orc_reg2cigar   <->  the band choice + retry loop of mem_reg2aln, bwamem.c:1187-1201
 * sequential, one region at a time, calling orc_global().
 *
 * Parity status: PINNED against the compiled reference's bwa_gen_cigar2 (oracle/_ref/libbwa_ref.so)
 * by tests/test_oracle_vs_ref.py and the committed fixture tests/golden/cigar_golden.npz.
 */
#include <stdlib.h>
#include <string.h>

#include "chain2aln_oracle.h"
#include "gencigar_oracle.h"

static void put_num(char *s, int *l, int c) /* kputw, kstring.h:62-77 (c >= 0 here) */
{
	char buf[16];
	int n = 0;
	if (c == 0) { s[(*l)++] = '0'; return; }
	for (; c > 0; c /= 10) buf[n++] = (char)('0' + c % 10);
	while (n) s[(*l)++] = buf[--n];
}

/* NM and MD from a CIGAR and the two (already oriented) sequences.  bwa.c:134-164.  md must hold 3*(ql+tl)+16 bytes. */
static int nm_md(int n_cigar, const uint32_t *cigar, const uint8_t *q, const uint8_t *t, int rev, char *md)
{
	const char *int2base = rev ? "TGCAN" : "ACGTN";
	int k, i, x = 0, y = 0, u = 0, n_mm = 0, n_gap = 0, l = 0;
	for (k = 0; k < n_cigar; ++k) {
		const int op = cigar[k] & 0xf, len = (int)(cigar[k] >> 4);
		if (op == 0) {
			for (i = 0; i < len; ++i) {
				if (q[x + i] != t[y + i]) {
					put_num(md, &l, u);
					md[l++] = int2base[t[y + i]];
					++n_mm, u = 0;
				} else ++u;
			}
			x += len, y += len;
		} else if (op == 2) {
			if (k > 0 && k < n_cigar - 1) { /* not for a leading or trailing deletion, bwa.c:152 */
				put_num(md, &l, u);
				md[l++] = '^';
				for (i = 0; i < len; ++i) md[l++] = int2base[t[y + i]];
				u = 0, n_gap += len;
			}
			y += len;
		} else if (op == 1) x += len, n_gap += len;
	}
	put_num(md, &l, u);
	md[l] = 0;
	return n_mm + n_gap;
}

int orc_infer_bw(int l1, int l2, int score, int a, int q, int r) /* bwamem.c:884-891 */
{
	int w;
	if (l1 == l2 && l1 * a - score < (q + r - a) << 1) return 0;
	w = (int)((double)((l1 < l2 ? l1 : l2) * a - score - q) / r + 2.);
	if (w < abs(l1 - l2)) w = abs(l1 - l2);
	return w;
}

void orc_gen_cigar(const bmh_params_t *p, int w_, int64_t l_pac, const uint8_t *pac, int l_query, const uint8_t *query,
                   int64_t rb, int64_t re, orc_cigar_t *out)
{
	orc_scoring_t sc;
	uint8_t *q, *rseq;
	int64_t rlen;
	int i;
	memset(out, 0, sizeof(*out));
	out->NM = -1;
	if (l_query <= 0 || rb >= re || (rb < l_pac && re > l_pac)) return; /* bwa.c:99 */
	rseq = orc_get_seq(l_pac, pac, rb, re, &rlen);
	if (re - rb != rlen) { free(rseq); return; }
	q = (uint8_t *)malloc((size_t)l_query);
	memcpy(q, query, (size_t)l_query);
	if (rb >= l_pac) { /* reverse both so that indels end up left-most, bwa.c:102-107 */
		for (i = 0; i < l_query >> 1; ++i) { uint8_t t = q[i]; q[i] = q[l_query - 1 - i], q[l_query - 1 - i] = t; }
		for (i = 0; i < rlen >> 1; ++i) { uint8_t t = rseq[i]; rseq[i] = rseq[rlen - 1 - i], rseq[rlen - 1 - i] = t; }
	}
	if (l_query == re - rb && w_ == 0) { /* no gap possible, bwa.c:108-114 */
		out->cigar = (uint32_t *)malloc(4);
		out->cigar[0] = (uint32_t)l_query << 4;
		out->n_cigar = 1;
		for (i = 0; i < l_query; ++i) out->score += p->mat[rseq[i] * 5 + q[i]];
	} else { /* band, bwa.c:116-125 */
		int max_ins = (int)((double)(((l_query + 1) >> 1) * p->mat[0] - p->o_ins) / p->e_ins + 1.);
		int max_del = (int)((double)(((l_query + 1) >> 1) * p->mat[0] - p->o_del) / p->e_del + 1.);
		int max_gap = max_ins > max_del ? max_ins : max_del, w, min_w;
		max_gap = max_gap > 1 ? max_gap : 1;
		w = (max_gap + abs((int)rlen - l_query) + 1) >> 1;
		w = w < w_ ? w : w_;
		min_w = abs((int)rlen - l_query) + 3;
		w = w > min_w ? w : min_w;
		sc.o_del = p->o_del, sc.e_del = p->e_del, sc.o_ins = p->o_ins, sc.e_ins = p->e_ins, sc.zdrop = 0, sc.m = 5, sc.mat = p->mat;
		out->w_used = w;
		out->score = orc_global(&sc, l_query, q, (int)rlen, rseq, w, &out->n_cigar, &out->cigar);
	}
	out->md = (char *)malloc(3 * ((size_t)l_query + (size_t)rlen) + 16);
	out->NM = nm_md(out->n_cigar, out->cigar, q, rseq, rb >= l_pac, out->md);
	free(q);
	free(rseq);
}

void orc_cigar_free(orc_cigar_t *c)
{
	free(c->cigar), free(c->md);
	memset(c, 0, sizeof(*c));
}

/* bwamem.c:1187-1201: initial band from the region's true score, then up to 3 tries with doubled bands */
void orc_reg2cigar(const bmh_params_t *p, int64_t l_pac, const uint8_t *pac, const uint8_t *read, int qb, int qe, int64_t rb,
                   int64_t re, int truesc, int reg_w, orc_cigar_t *out, int *rounds)
{
	int i = 0, calls = 0, last_sc = -(1 << 30), w2, tmp;
	tmp = orc_infer_bw(qe - qb, (int)(re - rb), truesc, p->a, p->o_del, p->e_del);
	w2 = orc_infer_bw(qe - qb, (int)(re - rb), truesc, p->a, p->o_ins, p->e_ins);
	w2 = w2 > tmp ? w2 : tmp;
	if (w2 > p->w) w2 = w2 < reg_w ? w2 : reg_w;
	memset(out, 0, sizeof(*out));
	do {
		orc_cigar_free(out);
		orc_gen_cigar(p, w2, l_pac, pac, qe - qb, read + qb, rb, re, out);
		++calls;
		if (out->score == last_sc) break;
		last_sc = out->score;
		w2 <<= 1;
	} while (++i < 3 && out->score < truesc - p->a);
	if (rounds) *rounds = calls;
}
