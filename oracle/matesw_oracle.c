/*
 * oracle/matesw_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See matesw_oracle.h.
 */
#include "matesw_oracle.h"

#include <stdlib.h>
#include <string.h>

#include "sw_oracle.h"

static int dir_of(int64_t l_pac, int64_t b1, int64_t b2, int64_t *dist) /* bwamem_pair.c:23-30 */
{
	int s1 = b1 >= l_pac, s2 = b2 >= l_pac;
	int64_t q = s1 == s2 ? b2 : (l_pac << 1) - 1 - b2; /* mate position seen from the anchor's strand */
	*dist = q > b1 ? q - b1 : b1 - q;
	return (s1 == s2 ? 0 : 1) ^ (q > b1 ? 0 : 3);
}

static uint8_t *get_seq(int64_t l_pac, const uint8_t *pac, int64_t beg, int64_t end, int64_t *len) /* bntseq.c:355-376 */
{
	uint8_t *seq = 0;
	int64_t k, n = 0;
	if (end < beg) { int64_t t = beg; beg = end, end = t; }
	if (end > l_pac << 1) end = l_pac << 1;
	if (beg < 0) beg = 0;
	*len = 0;
	if (beg < l_pac && end > l_pac) return 0; /* bridges the two strands */
	*len = end - beg;
	seq = (uint8_t *)malloc((size_t)(end - beg) + 1);
	for (k = beg; k < end; ++k) {
		const int64_t f = k >= l_pac ? (l_pac << 1) - 1 - k : k;
		const int b = pac[f >> 2] >> ((~f & 3) << 1) & 3;
		seq[n++] = (uint8_t)(k >= l_pac ? 3 - b : b);
	}
	return seq;
}

static void vec_push(bmh_alnreg_v *v, const bmh_alnreg_t *x) /* kvec.h:68-74 */
{
	if (v->n == v->m) {
		v->m = v->m ? v->m << 1 : 2;
		v->a = (bmh_alnreg_t *)realloc(v->a, sizeof(bmh_alnreg_t) * v->m);
	}
	v->a[v->n++] = *x;
}

/* one mem_matesw call, bwamem_pair.c:109-175 */
static int matesw_one(const bmh_params_t *p, const bmh_matesw_opt_t *o, int64_t l_pac, const uint8_t *pac,
                      const bmh_pestat_t pes[4], const bmh_alnreg_t *a, int l_ms, const uint8_t *ms, bmh_alnreg_v *ma,
                      bmh_dedup_fn dedup, void *user)
{
	int skip[4], r, n = 0;
	size_t k;
	for (r = 0; r < 4; ++r) skip[r] = pes[r].failed ? 1 : 0;
	for (k = 0; k < ma->n; ++k) { /* which orientations already have a properly placed mate */
		int64_t d;
		r = dir_of(l_pac, a->rb, ma->a[k].rb, &d);
		if (d >= pes[r].low && d <= pes[r].high) skip[r] = 1;
	}
	if (skip[0] + skip[1] + skip[2] + skip[3] == 4) return 0;
	for (r = 0; r < 4; ++r) {
		const int is_rev = (r >> 1) != (r & 1), is_larger = !(r >> 1);
		uint8_t *seq, *ref;
		int64_t rb, re, len;
		int i;
		if (skip[r]) continue;
		seq = (uint8_t *)malloc((size_t)l_ms + 1);
		for (i = 0; i < l_ms; ++i) seq[i] = is_rev ? (ms[l_ms - 1 - i] < 4 ? 3 - ms[l_ms - 1 - i] : 4) : ms[i];
		if (!is_rev) {
			rb = is_larger ? a->rb + pes[r].low : a->rb - pes[r].high;
			re = (is_larger ? a->rb + pes[r].high : a->rb - pes[r].low) + l_ms;
		} else {
			rb = (is_larger ? a->rb + pes[r].low : a->rb - pes[r].high) - l_ms;
			re = is_larger ? a->rb + pes[r].high : a->rb - pes[r].low;
		}
		if (rb < 0) rb = 0;
		if (re > l_pac << 1) re = l_pac << 1;
		ref = get_seq(l_pac, pac, rb, re, &len);
		if (len == re - rb) {
			const int xtra = ORC_XSUBO | ORC_XSTART | (l_ms * p->a < 250 ? ORC_XBYTE : 0) | (o->min_seed_len * p->a);
			const orc_kswr_t x = orc_align2(l_ms, seq, (int)len, ref, 5, p->mat, p->o_del, p->e_del, p->o_ins, p->e_ins, xtra, 0, 0);
			if (x.score >= o->min_seed_len && x.qb >= 0) {
				bmh_alnreg_t b;
				size_t at;
				memset(&b, 0, sizeof(b));
				b.qb = is_rev ? l_ms - (x.qe + 1) : x.qb;
				b.qe = is_rev ? l_ms - x.qb : x.qe + 1;
				b.rb = is_rev ? (l_pac << 1) - (rb + x.te + 1) : rb + x.tb;
				b.re = is_rev ? (l_pac << 1) - (rb + x.tb) : rb + x.te + 1;
				b.score = x.score, b.csub = x.score2, b.secondary = -1;
				b.seedcov = (int32_t)((b.re - b.rb < b.qe - b.qb ? b.re - b.rb : b.qe - b.qb) >> 1);
				vec_push(ma, &b);
				for (at = 0; at < ma->n - 1; ++at) /* first slot whose score is lower */
					if (ma->a[at].score < b.score) break;
				memmove(&ma->a[at + 1], &ma->a[at], sizeof(bmh_alnreg_t) * (ma->n - 1 - at));
				ma->a[at] = b;
			}
			++n;
		}
		if (n) ma->n = (size_t)dedup(user, (int)ma->n, ma->a);
		free(seq), free(ref);
	}
	return n;
}

int orc_matesw_pair(const bmh_params_t *p, const bmh_matesw_opt_t *o, int64_t l_pac, const uint8_t *pac,
                    const bmh_pestat_t pes[4], const bmh_read_t reads[2], bmh_alnreg_v regs[2], bmh_dedup_fn dedup,
                    void *user)
{
	bmh_alnreg_v b[2] = {{0, 0, 0}, {0, 0, 0}};
	int i, n = 0;
	size_t j;
	for (i = 0; i < 2; ++i) /* bwamem_pair.c:254-257 */
		for (j = 0; j < regs[i].n; ++j)
			if (regs[i].a[j].score >= regs[i].a[0].score - o->pen_unpaired) vec_push(&b[i], &regs[i].a[j]);
	for (i = 0; i < 2; ++i) /* :258-260 */
		for (j = 0; j < b[i].n && (int)j < o->max_matesw; ++j)
			n += matesw_one(p, o, l_pac, pac, pes, &b[i].a[j], reads[!i].l_seq, reads[!i].seq, &regs[!i], dedup, user);
	free(b[0].a), free(b[1].a);
	return n;
}

int orc_simple_dedup(void *user, int n, bmh_alnreg_t *a)
{
	int i, j, m = 0;
	(void)user;
	for (i = 1; i < n; ++i) { /* stable insertion sort */
		const bmh_alnreg_t x = a[i];
		for (j = i - 1; j >= 0; --j) {
			const bmh_alnreg_t *y = &a[j];
			const int before = x.score > y->score || (x.score == y->score && (x.rb < y->rb || (x.rb == y->rb && x.qb < y->qb)));
			if (!before) break;
			a[j + 1] = a[j];
		}
		a[j + 1] = x;
	}
	for (i = 0; i < n; ++i)
		if (m == 0 || a[i].score != a[m - 1].score || a[i].rb != a[m - 1].rb || a[i].qb != a[m - 1].qb) a[m++] = a[i];
	return m;
}
