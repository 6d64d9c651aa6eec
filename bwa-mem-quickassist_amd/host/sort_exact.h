/*
 * sort_exact.h -- an UNSTABLE sort whose output order is part of the contract.
 *
 * Several steps of bwa's post-processing sort records whose keys tie (regions with equal end coordinate before
 * de-duplication, equal scores before primary marking, equal chain weights before filtering) and then treat
 * neighbours asymmetrically, so WHICH of two equal records comes first decides the SAM output.  The reference sorts
 * them with klib's introsort (reference bwa-0.7.8/ksort.h:159-218: median-of-three quicksort with an explicit stack
 * that leaves runs of <= 16 elements for one final insertion sort, combsort when the recursion gets too deep; the
 * insertion sort is ksort.h:130-137, the combsort :138-158).  To reproduce its output on ties this file performs the
 * same comparisons and exchanges in the same order -- as one type-agnostic routine over byte records and a less-than
 * callback, instead of klib's per-type macro expansion.
 */
#ifndef BMH_SORT_EXACT_H
#define BMH_SORT_EXACT_H

#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BMH_SORT_MAXREC 128 /* records are copied through a stack buffer */

typedef int (*bmh_lt_fn)(const void *a, const void *b);

typedef struct {
	char *a;
	size_t sz;
	bmh_lt_fn lt;
	char tmp[BMH_SORT_MAXREC];
} bmh_sortctx_t;

static inline char *sx_at(const bmh_sortctx_t *c, ptrdiff_t i) { return c->a + (size_t)i * c->sz; }
static inline void sx_swap(bmh_sortctx_t *c, ptrdiff_t i, ptrdiff_t j)
{
	memcpy(c->tmp, sx_at(c, i), c->sz);
	memcpy(sx_at(c, i), sx_at(c, j), c->sz);
	memcpy(sx_at(c, j), c->tmp, c->sz);
}

/* ksort.h:130-137 over [s, t) */
static inline void sx_insertion(bmh_sortctx_t *c, ptrdiff_t s, ptrdiff_t t)
{
	ptrdiff_t i, j;
	for (i = s + 1; i < t; ++i)
		for (j = i; j > s && c->lt(sx_at(c, j), sx_at(c, j - 1)); --j) sx_swap(c, j, j - 1);
}

/* ksort.h:138-158 over [s, s+n) */
static inline void sx_comb(bmh_sortctx_t *c, ptrdiff_t s, size_t n)
{
	const double shrink = 1.2473309501039786540366528676643;
	size_t gap = n;
	int swapped;
	do {
		ptrdiff_t i;
		if (gap > 2) {
			gap = (size_t)((double)gap / shrink);
			if (gap == 9 || gap == 10) gap = 11;
		}
		swapped = 0;
		for (i = s; i < s + (ptrdiff_t)n - (ptrdiff_t)gap; ++i)
			if (c->lt(sx_at(c, i + (ptrdiff_t)gap), sx_at(c, i))) sx_swap(c, i, i + (ptrdiff_t)gap), swapped = 1;
	} while (swapped || gap > 2);
	if (gap != 1) sx_insertion(c, s, s + (ptrdiff_t)n);
}

/* ksort.h:159-218 */
static inline void bmh_sort_exact(void *base, size_t n, size_t sz, bmh_lt_fn lt)
{
	bmh_sortctx_t c;
	struct { ptrdiff_t s, t; int d; } *stack, *top;
	char pivot[BMH_SORT_MAXREC];
	ptrdiff_t s, t;
	int d;
	if (n < 1 || sz > BMH_SORT_MAXREC) return;
	c.a = (char *)base, c.sz = sz, c.lt = lt;
	if (n == 2) {
		if (lt(sx_at(&c, 1), sx_at(&c, 0))) sx_swap(&c, 0, 1);
		return;
	}
	for (d = 2; (1ul << d) < n; ++d) {}
	stack = malloc(sizeof(*stack) * (sizeof(size_t) * (size_t)d + 2));
	top = stack, s = 0, t = (ptrdiff_t)n - 1, d <<= 1;
	for (;;) {
		if (s < t) {
			ptrdiff_t i, j, k;
			if (--d == 0) { /* too deep: combsort the whole range */
				sx_comb(&c, s, (size_t)(t - s + 1));
				t = s;
				continue;
			}
			i = s, j = t, k = i + ((j - i) >> 1) + 1; /* median of first, middle+1, last */
			if (lt(sx_at(&c, k), sx_at(&c, i))) {
				if (lt(sx_at(&c, k), sx_at(&c, j))) k = j;
			} else k = lt(sx_at(&c, j), sx_at(&c, i)) ? i : j;
			memcpy(pivot, sx_at(&c, k), sz);
			if (k != t) sx_swap(&c, k, t);
			for (;;) {
				do ++i; while (lt(sx_at(&c, i), pivot));
				do --j; while (i <= j && lt(pivot, sx_at(&c, j)));
				if (j <= i) break;
				sx_swap(&c, i, j);
			}
			sx_swap(&c, i, t);
			if (i - s > t - i) { /* larger side onto the stack if it is longer than 16, go on with the smaller one */
				if (i - s > 16) top->s = s, top->t = i - 1, top->d = d, ++top;
				s = t - i > 16 ? i + 1 : t;
			} else {
				if (t - i > 16) top->s = i + 1, top->t = t, top->d = d, ++top;
				t = i - s > 16 ? i - 1 : s;
			}
		} else if (top == stack) {
			free(stack);
			sx_insertion(&c, 0, (ptrdiff_t)n);
			return;
		} else --top, s = top->s, t = top->t, d = top->d;
	}
}

#endif
