/*
 * oracle/fmindex_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See fmindex_oracle.h.
 * Written from the behaviour of bwa-0.7.8/bwt.c and bwamem.c:118-162; counting by bit arithmetic, no table.
 */
#include "fmindex_oracle.h"

#include <stdlib.h>
#include <string.h>

/* symbols equal to c among the first n (1..16) symbols of a packed word (symbol 0 in the top two bits) */
static int count_in_word(uint32_t w, int c, int n)
{
	uint32_t x = w ^ (uint32_t)(c * 0x55555555u); /* a symbol equals c <=> both of its bits are now 0 */
	uint32_t eq = ~(x | x >> 1) & 0x55555555u;
	eq &= 0xffffffffu << (32 - 2 * n);
	return __builtin_popcount(eq);
}

void orc_bwt_occ4(const bmh_bwt_t *b, uint64_t k, uint64_t cnt[4]) /* bwt.c:159-177 */
{
	const uint32_t *p;
	int c, j, full, rest;
	if (k == (uint64_t)-1) { cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0; return; }
	k -= (k >= b->primary); /* the sentinel is not stored */
	p = b->bwt + ((k >> 7) << 4); /* 16 words per 128 symbols: 4 x u64 counts, then 8 words of symbols (bwt.h:63-64) */
	memcpy(cnt, p, 32);
	p += 8;
	full = (int)((k & 127) >> 4), rest = (int)(k & 15) + 1;
	for (c = 0; c < 4; ++c) {
		int n = 0;
		for (j = 0; j < full; ++j) n += count_in_word(p[j], c, 16);
		n += count_in_word(p[full], c, rest);
		cnt[c] += (uint64_t)n;
	}
}

static uint64_t g_extends; /* statistics for the bench (not thread-safe; single-threaded use only) */
uint64_t orc_fm_extends(int reset)
{
	const uint64_t v = g_extends;
	if (reset) g_extends = 0;
	return v;
}

void orc_bwt_extend(const bmh_bwt_t *b, const bmh_smem_intv_t *ik, bmh_smem_intv_t ok[4], int is_back) /* bwt.c:261-274 */
{
	uint64_t tk[4], tl[4];
	int i;
	++g_extends;
	orc_bwt_occ4(b, ik->x[!is_back] - 1, tk);
	orc_bwt_occ4(b, ik->x[!is_back] - 1 + ik->x[2], tl);
	for (i = 0; i < 4; ++i) {
		ok[i].x[!is_back] = b->L2[i] + 1 + tk[i];
		ok[i].x[2] = tl[i] - tk[i];
	}
	ok[3].x[is_back] = ik->x[is_back] + (ik->x[!is_back] <= b->primary && ik->x[!is_back] + ik->x[2] - 1 >= b->primary);
	ok[2].x[is_back] = ok[3].x[is_back] + ok[3].x[2];
	ok[1].x[is_back] = ok[2].x[is_back] + ok[2].x[2];
	ok[0].x[is_back] = ok[1].x[is_back] + ok[1].x[2];
}

static void reverse(bmh_smem_intv_t *a, int n)
{
	int j;
	for (j = 0; j < n >> 1; ++j) {
		bmh_smem_intv_t t = a[n - 1 - j];
		a[n - 1 - j] = a[j], a[j] = t;
	}
}

int orc_bwt_smem1(const bmh_bwt_t *b, int len, const uint8_t *q, int x, int min_intv, bmh_smem_intv_t *mem, int *n_mem)
{
	bmh_smem_intv_t ik, ok[4], *prev, *curr, *sw;
	int i, j, c, ret, np = 0, nc = 0, nm = 0;
	*n_mem = 0;
	if (q[x] > 3) return x + 1;
	if (min_intv < 1) min_intv = 1;
	prev = (bmh_smem_intv_t *)malloc(sizeof(*prev) * ((size_t)len + 2));
	curr = (bmh_smem_intv_t *)malloc(sizeof(*curr) * ((size_t)len + 2));
	ik.x[0] = b->L2[q[x]] + 1, ik.x[2] = b->L2[q[x] + 1] - b->L2[q[x]], ik.x[1] = b->L2[3 - q[x]] + 1; /* bwt_set_intv, bwt.h:75 */
	ik.info = (uint64_t)x + 1;
	for (i = x + 1; i < len; ++i) { /* forward: extend to the right while the interval stays large enough (bwt.c:303-317) */
		if (q[i] < 4) {
			c = 3 - q[i];
			orc_bwt_extend(b, &ik, ok, 0);
			if (ok[c].x[2] != ik.x[2]) {
				curr[nc++] = ik;
				if (ok[c].x[2] < (uint64_t)min_intv) break;
			}
			ik = ok[c], ik.info = (uint64_t)i + 1;
		} else {
			curr[nc++] = ik;
			break;
		}
	}
	if (i == len) curr[nc++] = ik;
	reverse(curr, nc);
	ret = (int)curr[0].info;
	sw = curr, curr = prev, prev = sw, np = nc;
	for (i = x - 1; i >= -1; --i) { /* backward: longest matches first; keep those that cannot be extended (bwt.c:323-344) */
		c = i < 0 ? -1 : q[i] < 4 ? q[i] : -1;
		for (j = 0, nc = 0; j < np; ++j) {
			const bmh_smem_intv_t *p = &prev[j];
			orc_bwt_extend(b, p, ok, 1);
			if (c < 0 || ok[c].x[2] < (uint64_t)min_intv) {
				if (nc == 0 && (nm == 0 || (uint64_t)(i + 1) < mem[nm - 1].info >> 32)) {
					ik = *p, ik.info |= (uint64_t)(i + 1) << 32;
					mem[nm++] = ik;
				}
			} else if (nc == 0 || ok[c].x[2] != curr[nc - 1].x[2]) {
				ok[c].info = p->info;
				curr[nc++] = ok[c];
			}
		}
		if (nc == 0) break;
		sw = curr, curr = prev, prev = sw, np = nc;
	}
	reverse(mem, nm);
	free(prev), free(curr);
	*n_mem = nm;
	return ret;
}

static uint64_t occ1(const bmh_bwt_t *b, uint64_t k, int c) /* bwt_occ, bwt.c:107-129 */
{
	const uint32_t *p;
	uint64_t n;
	int j, full;
	if (k == b->seq_len) return b->L2[c + 1] - b->L2[c];
	if (k == (uint64_t)-1) return 0;
	k -= (k >= b->primary);
	p = b->bwt + ((k >> 7) << 4);
	memcpy(&n, p + 2 * c, 8);
	p += 8;
	full = (int)((k & 127) >> 4);
	for (j = 0; j < full; ++j) n += (uint64_t)count_in_word(p[j], c, 16);
	return n + (uint64_t)count_in_word(p[full], c, (int)(k & 15) + 1);
}

uint64_t orc_bwt_sa(const bmh_bwt_t *b, uint64_t k) /* bwt.c:85-95 with bwt_invPsi :52-58 */
{
	uint64_t sa = 0;
	const uint64_t mask = (uint64_t)b->sa_intv - 1;
	while (k & mask) {
		const uint64_t x = k - (k > b->primary);
		const int c = (int)(b->bwt[((x >> 7) << 4) + 8 + ((x & 127) >> 4)] >> ((~x & 15) << 1) & 3); /* bwt_B0, bwt.h:70 */
		++sa;
		k = k == b->primary ? 0 : b->L2[c] + occ1(b, k, c);
	}
	return sa + b->sa[k / (uint64_t)b->sa_intv];
}

int orc_smem_calls(const bmh_bwt_t *b, const bmh_smem_opt_t *o, int len, const uint8_t *q, bmh_smem_call_t *calls,
                   int call_cap, bmh_smem_intv_t *pool, int pool_cap, int *pool_used)
{
	const int split_len = o->split_len < len ? o->split_len : len; /* bwamem.c:213 */
	int start = 0, nc = 0, used = 0;
	while (start < len) { /* smem_next2, bwamem.c:118-162 */
		int n, i, mx = 0, mx_i = 0, ret, n_rec = 1; /* n_rec: calls of this round (2 with re-seeding) */
		while (start < len && q[start] > 3) ++start;
		if (start == len) break;
		if (nc + 2 > call_cap || used + 2 * (len + 1) > pool_cap) return -1;
		ret = orc_bwt_smem1(b, len, q, start, o->start_width, pool + used, &n);
		calls[nc].x = start, calls[nc].min_intv = o->start_width, calls[nc].ret = ret, calls[nc].n = n;
		calls[nc].first = (uint32_t)used, calls[nc].rsv = 0, ++nc;
		start = ret;
		for (i = 0; i < n; ++i) { /* the longest match, first of equals */
			const int l = (int)((uint32_t)pool[used + i].info - (uint32_t)(pool[used + i].info >> 32));
			if (mx < l) mx = l, mx_i = i;
		}
		if (n > 0 && split_len > 0 && mx >= split_len && pool[used + mx_i].x[2] <= (uint64_t)o->split_width) { /* re-seeding */
			const bmh_smem_intv_t *p = &pool[used + mx_i];
			const int mid = (int)(((uint32_t)p->info + (uint32_t)(p->info >> 32)) >> 1), mi = (int)p->x[2] + 1;
			int n2;
			ret = orc_bwt_smem1(b, len, q, mid, mi, pool + used + n, &n2);
			calls[nc].x = mid, calls[nc].min_intv = mi, calls[nc].ret = ret, calls[nc].n = n2;
			calls[nc].first = (uint32_t)(used + n), calls[nc].rsv = 0, ++nc;
			n += n2, n_rec = 2;
		}
		if (o->min_emit_len > 0) { /* bmh_smem_opt_t.min_emit_len: hand back the long intervals only, order kept */
			int k, w = used, rd = used;
			for (k = nc - n_rec; k < nc; ++k) {
				const int cn = calls[k].n;
				calls[k].first = (uint32_t)w, calls[k].n = 0;
				for (i = 0; i < cn; ++i, ++rd)
					if ((int)((uint32_t)pool[rd].info - (uint32_t)(pool[rd].info >> 32)) >= o->min_emit_len) pool[w++] = pool[rd], ++calls[k].n;
			}
			n = w - used;
		}
		used += n;
	}
	*pool_used = used;
	return nc;
}
