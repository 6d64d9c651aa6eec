/*
 * chain_batch.c -- seeds to chains (host side, plain C, above the C-ABI): the rest of SURVEY.md §8(f) row 3.
 *
 * Replaces, for a batch of reads whose FM-index queries were answered by bmh_smem_batch / bmh_sa_batch:
 *   smem_next2         reference bwa-0.7.8/bwamem.c:118-157  which bwt_smem1 results form one round of seeds (the
 *                      re-seeding call from the middle of a long unique match and the ordered merge of its result)
 *   mem_insert_seed    bwamem.c:208-243   every occurrence of every long-enough, rare-enough seed, in order, into a
 *                      B-tree of chains keyed by reference position (test_and_merge :186-206)
 *   mem_chain          bwamem.c:283-306   ... and out of it in key order
 *   mem_chain_flt      bwamem.c:319-380   (mem_chain_weight :245-263) dropping chains shadowed by better ones
 *
 * Two details decide the ORDER of the chains, which the extension stage and finally the SAM output depend on, and
 * both are reproduced literally:
 *   * chains with EQUAL keys: klib's B-tree (kbtree.h) puts a new key behind the first equal key of the leaf its
 *     descent ends in, and `kb_intervalp` returns the first equal key of the first node on its way down that has
 *     one -- both depend on how the tree has split so far.  So this file keeps the same tree: nodes of 2t-1 = 15 keys
 *     (t from kb_init with KB_DEFAULT_SIZE = 512 bytes and 24-byte keys, kbtree.h:54-66), pre-emptive splitting on the
 *     way down (kbtree.h:176-212), the two-sided binary search of __kb_getp_aux (kbtree.h:122-135).
 *   * chains of EQUAL weight: mem_chain_flt orders them with klib's unstable introsort (sort_exact.h).
 */
#include <stdlib.h>
#include <string.h>

#include "../../include/bwamem_hip.h"
#include "sort_exact.h"

/* ---- the B-tree of chains ------------------------------------------------------------------------------------------ */
enum { BT_T = 8, BT_MAX = 2 * BT_T - 1 }; /* kb_init(chn, 512): t = ((512-4-8)/(8+24)+1)>>1 = 8 */

typedef struct btnode {
	int n, internal;
	bmh_chain_t key[BT_MAX];
	struct btnode *child[BT_MAX + 1];
} btnode_t;

typedef struct {
	btnode_t *root;
	int n_keys;
	btnode_t **all; /* every node, for a flat release */
	int n_all, m_all;
} bt_t;

static btnode_t *bt_node(bt_t *b)
{
	btnode_t *x = (btnode_t *)calloc(1, sizeof(btnode_t));
	if (b->n_all == b->m_all) {
		b->m_all = b->m_all ? b->m_all << 1 : 16;
		b->all = (btnode_t **)realloc(b->all, sizeof(btnode_t *) * (size_t)b->m_all);
	}
	b->all[b->n_all++] = x;
	return x;
}

static inline int key_cmp(int64_t a, int64_t b) { return (b < a) - (a < b); } /* chain_cmp, bwamem.c:183 */

/* __kb_getp_aux, kbtree.h:122-135: the first key equal to k (*r = 0), else the last key below it (possibly -1; *r != 0) */
static int bt_find(const btnode_t *x, int64_t k, int *r)
{
	int begin = 0, end = x->n, dummy;
	if (!r) r = &dummy;
	if (x->n == 0) return -1;
	while (begin < end) {
		const int mid = (begin + end) >> 1;
		if (key_cmp(x->key[mid].pos, k) < 0) begin = mid + 1;
		else end = mid;
	}
	if (begin == x->n) {
		*r = 1;
		return x->n - 1;
	}
	if ((*r = key_cmp(k, x->key[begin].pos)) < 0) --begin;
	return begin;
}

/* kb_intervalp (kbtree.h:153-169), lower bound only: the closest chain at or below k */
static bmh_chain_t *bt_lower(bt_t *b, int64_t k)
{
	btnode_t *x = b->root;
	bmh_chain_t *lower = 0;
	while (x) {
		int r = 0;
		const int i = bt_find(x, k, &r);
		if (i >= 0 && r == 0) return &x->key[i];
		if (i >= 0) lower = &x->key[i];
		if (!x->internal) return lower;
		x = x->child[i + 1];
	}
	return lower;
}

/* __kb_split, kbtree.h:176-192: child y = x->child[i] is full; its upper half moves to a new right sibling */
static void bt_split(bt_t *b, btnode_t *x, int i, btnode_t *y)
{
	btnode_t *z = bt_node(b);
	z->internal = y->internal, z->n = BT_T - 1;
	memcpy(z->key, y->key + BT_T, sizeof(bmh_chain_t) * (BT_T - 1));
	if (y->internal) memcpy(z->child, y->child + BT_T, sizeof(btnode_t *) * BT_T);
	y->n = BT_T - 1;
	memmove(x->child + i + 2, x->child + i + 1, sizeof(btnode_t *) * (size_t)(x->n - i));
	x->child[i + 1] = z;
	memmove(x->key + i + 1, x->key + i, sizeof(bmh_chain_t) * (size_t)(x->n - i));
	x->key[i] = y->key[BT_T - 1];
	++x->n;
}

/* kb_putp / __kb_putp_aux, kbtree.h:193-227 (iterative: the recursion there is a plain descent) */
static void bt_put(bt_t *b, const bmh_chain_t *k)
{
	btnode_t *x = b->root;
	++b->n_keys;
	if (x->n == BT_MAX) { /* grow at the root */
		btnode_t *s = bt_node(b);
		b->root = s, s->internal = 1, s->n = 0, s->child[0] = x;
		bt_split(b, s, 0, x);
		x = s;
	}
	while (x->internal) {
		int i = bt_find(x, k->pos, 0) + 1;
		if (x->child[i]->n == BT_MAX) {
			bt_split(b, x, i, x->child[i]);
			if (key_cmp(k->pos, x->key[i].pos) > 0) ++i;
		}
		x = x->child[i];
	}
	{
		const int i = bt_find(x, k->pos, 0);
		if (i != x->n - 1) memmove(x->key + i + 2, x->key + i + 1, sizeof(bmh_chain_t) * (size_t)(x->n - i - 1));
		x->key[i + 1] = *k;
		++x->n;
	}
}

static void bt_walk(const btnode_t *x, bmh_chain_t *out, size_t *n) /* in key order, __kb_traverse */
{
	int i;
	for (i = 0; i < x->n; ++i) {
		if (x->internal) bt_walk(x->child[i], out, n);
		out[(*n)++] = x->key[i];
	}
	if (x->internal) bt_walk(x->child[x->n], out, n);
}

/* ---- bwamem.c:186-206 */
static int test_and_merge(const bmh_chain_opt_t *o, int64_t l_pac, bmh_chain_t *c, const bmh_seed_t *p)
{
	const bmh_seed_t *last = &c->seeds[c->n - 1];
	const int64_t qend = last->qbeg + last->len, rend = last->rbeg + last->len;
	int64_t x, y;
	if (p->qbeg >= c->seeds[0].qbeg && p->qbeg + p->len <= qend && p->rbeg >= c->seeds[0].rbeg && p->rbeg + p->len <= rend) return 1; /* contained */
	if ((last->rbeg < l_pac || c->seeds[0].rbeg < l_pac) && p->rbeg >= l_pac) return 0; /* other strand */
	x = p->qbeg - last->qbeg; /* never negative */
	y = p->rbeg - last->rbeg;
	if (y >= 0 && x - y <= o->w && y - x <= o->w && x - last->len < o->max_chain_gap && y - last->len < o->max_chain_gap) { /* grow */
		if (c->n == c->m) {
			c->m <<= 1;
			c->seeds = (bmh_seed_t *)realloc(c->seeds, (size_t)c->m * sizeof(bmh_seed_t));
		}
		c->seeds[c->n++] = *p;
		return 1;
	}
	return 0; /* a new chain */
}

/* ---- bwamem.c:245-263 */
static int chain_weight(const bmh_chain_t *c)
{
	int64_t end;
	int j, w = 0, tmp;
	for (j = 0, end = 0; j < c->n; ++j) {
		const bmh_seed_t *s = &c->seeds[j];
		if (s->qbeg >= end) w += s->len;
		else if (s->qbeg + s->len > end) w += (int)(s->qbeg + s->len - end);
		end = end > s->qbeg + s->len ? end : s->qbeg + s->len;
	}
	tmp = w;
	for (j = 0, end = 0; j < c->n; ++j) { /* (the reference adds the second pass onto w and advances `end` on the QUERY, :256-261) */
		const bmh_seed_t *s = &c->seeds[j];
		if (s->rbeg >= end) w += s->len;
		else if (s->rbeg + s->len > end) w += (int)(s->rbeg + s->len - end);
		end = end > s->qbeg + s->len ? end : s->qbeg + s->len;
	}
	return w < tmp ? w : tmp;
}

/* ---- bwamem.c:310-380 */
typedef struct {
	int beg, end, w;
	void *p, *p2;
} flt_aux_t;
static int flt_lt(const void *a, const void *b) { return ((const flt_aux_t *)a)->w > ((const flt_aux_t *)b)->w; }

static int chain_flt(const bmh_chain_opt_t *o, int n_chn, bmh_chain_t *chains)
{
	flt_aux_t *a;
	bmh_chain_t *swap;
	int i, j, n;
	if (n_chn <= 1) return n_chn;
	a = (flt_aux_t *)malloc(sizeof(flt_aux_t) * (size_t)n_chn);
	for (i = 0; i < n_chn; ++i) {
		bmh_chain_t *c = &chains[i];
		a[i].beg = c->seeds[0].qbeg;
		a[i].end = c->seeds[c->n - 1].qbeg + c->seeds[c->n - 1].len;
		a[i].w = chain_weight(c), a[i].p = c, a[i].p2 = 0;
	}
	bmh_sort_exact(a, (size_t)n_chn, sizeof(flt_aux_t), flt_lt);
	swap = (bmh_chain_t *)malloc(sizeof(bmh_chain_t) * (size_t)n_chn); /* best chain first */
	for (i = 0; i < n_chn; ++i) swap[i] = *(bmh_chain_t *)a[i].p, a[i].p = &chains[i];
	memcpy(chains, swap, sizeof(bmh_chain_t) * (size_t)n_chn);
	free(swap);
	for (i = 1, n = 1; i < n_chn; ++i) {
		for (j = 0; j < n; ++j) {
			const int b_max = a[j].beg > a[i].beg ? a[j].beg : a[i].beg, e_min = a[j].end < a[i].end ? a[j].end : a[i].end;
			if (e_min > b_max) { /* overlap on the query */
				const int min_l = a[i].end - a[i].beg < a[j].end - a[j].beg ? a[i].end - a[i].beg : a[j].end - a[j].beg;
				if (e_min - b_max >= min_l * o->mask_level) { /* significant */
					if (a[j].p2 == 0) a[j].p2 = a[i].p;
					if (a[i].w < a[j].w * o->chain_drop_ratio && a[j].w - a[i].w >= o->min_seed_len << 1) break;
				}
			}
		}
		if (j == n) a[n++] = a[i]; /* not shadowed by a better chain */
	}
	for (i = 0; i < n; ++i) { /* kept: the survivors and, for each, the first chain it shadows */
		bmh_chain_t *c = (bmh_chain_t *)a[i].p;
		if (c->n > 0) c->n = -c->n;
		c = (bmh_chain_t *)a[i].p2;
		if (c && c->n > 0) c->n = -c->n;
	}
	free(a);
	for (i = 0; i < n_chn; ++i) {
		bmh_chain_t *c = &chains[i];
		if (c->n >= 0) free(c->seeds), c->seeds = 0, c->n = c->m = 0;
		else c->n = -c->n;
	}
	for (i = n = 0; i < n_chn; ++i)
		if (chains[i].n > 0) {
			if (n != i) chains[n++] = chains[i];
			else ++n;
		}
	return n;
}

/* ---- one round of smem_next2 (bwamem.c:118-157) from the batch's call records: the intervals it returns */
typedef struct { /* a merged round: pointers into the batch's interval array (their index finds their bwt_sa results) */
	const bmh_smem_intv_t **a;
	size_t n, m;
} intv_v;
static inline void iv_push(intv_v *v, const bmh_smem_intv_t *x)
{
	if (v->n == v->m) {
		v->m = v->m ? v->m << 1 : 64;
		v->a = (const bmh_smem_intv_t **)realloc((void *)v->a, sizeof(*v->a) * v->m);
	}
	v->a[v->n++] = x;
}
static inline int iv_len(const bmh_smem_intv_t *p) { return (int)((uint32_t)p->info - (uint32_t)(p->info >> 32)); }

/* The suffix-array entries chaining will ask for (bwamem.c:218-225): every occurrence of every interval that is long
 * and rare enough -- taken over ALL intervals the batch returned, a superset of the merged lists that are walked.
 * sa_off[k] = index of interval k's first entry in the key list (and later in the position list), or UINT64_MAX for an
 * interval that is never looked up; keys (nullable: count only) receives x[0] + j for j < x[2].  Returns the key count. */
uint64_t bmh_chain_sa_keys(const bmh_chain_opt_t *o, uint64_t n_intv, const bmh_smem_intv_t *intv, uint64_t *sa_off, uint64_t *keys)
{
	uint64_t k, n = 0;
	for (k = 0; k < n_intv; ++k) {
		const bmh_smem_intv_t *p = &intv[k];
		if (iv_len(p) >= o->min_seed_len && p->x[2] <= (uint64_t)o->max_occ) {
			uint64_t j;
			if (sa_off) sa_off[k] = n;
			if (keys)
				for (j = 0; j < p->x[2]; ++j) keys[n + j] = p->x[0] + j;
			n += p->x[2];
		} else if (sa_off) sa_off[k] = UINT64_MAX;
	}
	return n;
}

int bmh_chain_reads(const bmh_chain_opt_t *o, int64_t l_pac, int n_reads, const bmh_read_t *reads, const uint32_t *call_off,
                    const bmh_smem_call_t *calls, const uint64_t *intv_off, const bmh_smem_intv_t *intv, const uint64_t *sa_off,
                    const uint64_t *sa_pos, bmh_chain_v *chains)
{
	intv_v merged = {0, 0, 0};
	int r, rc = BMH_OK;
	if (!o || n_reads < 0 || (n_reads > 0 && (!reads || !call_off || !calls || !intv_off || !intv || !sa_off || !sa_pos || !chains))) return BMH_E_ARG;
	for (r = 0; r < n_reads; ++r) {
		const int len = reads[r].l_seq;
		const bmh_smem_intv_t *iv = intv + intv_off[r];
		int split_len = o->split_len < len ? o->split_len : len; /* bwamem.c:213 */
		uint32_t c = call_off[r];
		bt_t bt;
		size_t k;
		chains[r].n = chains[r].m = 0, chains[r].a = 0;
		if (len < o->min_seed_len) continue; /* bwamem.c:291 */
		memset(&bt, 0, sizeof(bt));
		bt.root = bt_node(&bt);
		while (c < call_off[r + 1]) { /* one smem_next2 round per main bwt_smem1 call */
			const bmh_smem_call_t *mc = &calls[c++];
			const bmh_smem_intv_t *m = iv + mc->first;
			const bmh_smem_intv_t **list = 0; /* null: the main call's intervals as they stand */
			size_t n_list = (size_t)mc->n, i;
			int max = 0, max_i = 0;
			for (i = 0; i < (size_t)mc->n; ++i) /* the longest match, bwamem.c:130-134 */
				if (max < iv_len(&m[i])) max = iv_len(&m[i]), max_i = (int)i;
			if (mc->n > 0 && split_len > 0 && max >= split_len && m[max_i].x[2] <= (uint64_t)o->split_width) {
				/* long and rare: its middle was searched again with a higher occurrence floor (bwamem.c:135-155);
				 * that call is the next record */
				const bmh_smem_call_t *sc;
				const bmh_smem_intv_t *s;
				size_t j = 0;
				if (c >= call_off[r + 1]) { rc = BMH_E_ARG; goto fail; }
				sc = &calls[c++], s = iv + sc->first;
				if (sc->x != (int)(((uint32_t)m[max_i].info + (uint32_t)(m[max_i].info >> 32)) >> 1) || sc->min_intv != (int)(m[max_i].x[2] + 1)) {
					rc = BMH_E_ARG; /* the call list does not follow smem_next2's order */
					goto fail;
				}
				merged.n = 0, i = 0;
#define KEEP_SUB(p) (iv_len(p) >= max >> 1 && (int)(uint32_t)(p)->info > mc->x)
				while (i < (size_t)mc->n && j < (size_t)sc->n) { /* ordered merge by (start, len - end) */
					const int64_t xi = (int64_t)(m[i].info >> 32 << 32 | (uint64_t)(uint32_t)(len - (int)(uint32_t)m[i].info));
					const int64_t xj = (int64_t)(s[j].info >> 32 << 32 | (uint64_t)(uint32_t)(len - (int)(uint32_t)s[j].info));
					if (xi < xj) iv_push(&merged, &m[i]), ++i;
					else if (KEEP_SUB(&s[j])) iv_push(&merged, &s[j]), ++j;
					else ++j;
				}
				for (; i < (size_t)mc->n; ++i) iv_push(&merged, &m[i]);
				for (; j < (size_t)sc->n; ++j)
					if (KEEP_SUB(&s[j])) iv_push(&merged, &s[j]);
#undef KEEP_SUB
				list = merged.a, n_list = merged.n;
			}
			for (i = 0; i < n_list; ++i) { /* mem_insert_seed's loop body, bwamem.c:216-240 */
				const bmh_smem_intv_t *p = list ? list[i] : &m[i];
				const int slen = iv_len(p);
				uint64_t kk;
				if (slen < o->min_seed_len || p->x[2] > (uint64_t)o->max_occ) continue;
				if (sa_off[(size_t)(p - intv)] == UINT64_MAX) { rc = BMH_E_ARG; goto fail; } /* the caller's table must cover every such interval */
				for (kk = 0; kk < p->x[2]; ++kk) {
					bmh_chain_t tmp, *lower;
					bmh_seed_t sd;
					int to_add = 0;
					sd.rbeg = tmp.pos = (int64_t)sa_pos[sa_off[(size_t)(p - intv)] + kk];
					sd.qbeg = (int32_t)(p->info >> 32), sd.len = slen;
					if (sd.rbeg < l_pac && l_pac < sd.rbeg + sd.len) continue; /* bridges the strands */
					if (bt.n_keys) {
						lower = bt_lower(&bt, tmp.pos);
						if (!lower || !test_and_merge(o, l_pac, lower, &sd)) to_add = 1;
					} else to_add = 1;
					if (to_add) {
						tmp.n = 1, tmp.m = 4;
						tmp.seeds = (bmh_seed_t *)calloc((size_t)tmp.m, sizeof(bmh_seed_t));
						tmp.seeds[0] = sd;
						bt_put(&bt, &tmp);
					}
				}
			}
		}
		if (bt.n_keys) {
			chains[r].a = (bmh_chain_t *)malloc(sizeof(bmh_chain_t) * (size_t)bt.n_keys);
			chains[r].m = (size_t)bt.n_keys;
			bt_walk(bt.root, chains[r].a, &chains[r].n);
			chains[r].n = (size_t)chain_flt(o, (int)chains[r].n, chains[r].a); /* bwamem.c:1097 */
		}
		for (k = 0; k < (size_t)bt.n_all; ++k) free(bt.all[k]);
		free(bt.all);
		continue;
	fail:
		for (k = 0; k < (size_t)bt.n_all; ++k) {
			int q;
			for (q = 0; q < bt.all[k]->n; ++q) free(bt.all[k]->key[q].seeds);
			free(bt.all[k]);
		}
		free(bt.all);
		break;
	}
	free((void *)merged.a);
	return rc;
}
