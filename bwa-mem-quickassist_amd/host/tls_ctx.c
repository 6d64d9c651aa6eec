/* tls_ctx.c -- see tls_ctx.h */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tls_ctx.h"

typedef struct {
	bmh_ctx_t *ctx;
	bmh_params_t params;
	int have, busy;
} slot_t;

#define BMH_POOL_MAX 1024
static slot_t g_slots[BMH_POOL_MAX];
static int g_n;
static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;

/* The GPUs the pool's contexts live on: $BMH_DEVICES = comma-separated ordinals (contexts are created on them in turn, so
 * the host threads' batches spread over the GPUs of a node; the resident reference and index are copied once per device),
 * else $BMH_DEVICE, else 0. */
static int pool_device(int nth)
{
	const char *list = getenv("BMH_DEVICES"), *one = getenv("BMH_DEVICE");
	int dev[64], n = 0;
	if (list && *list) {
		const char *p = list;
		while (*p && n < 64) {
			char *end;
			long v = strtol(p, &end, 10);
			if (end == p) break;
			dev[n++] = (int)v;
			p = *end == ',' ? end + 1 : end;
			if (*end != ',') break;
		}
	}
	if (n == 0) return one ? atoi(one) : 0;
	return dev[nth % n];
}

/* Workspaces of a context that will see the shim's batches -- up to 64 k reads of phase 1 (reads, windows, SMEM tables
 * coming back), a slice of a chunk in phase 2 -- sized once, so that no batch in the middle of a run has to grow them
 * ($BMH_RESERVE_MB scales the figures; 0 turns the reservation off). */
static void pool_reserve(bmh_ctx_t *ctx)
{
	const char *e = getenv("BMH_RESERVE_MB");
	const size_t mb = e ? (size_t)atoi(e) : 16;
	if (mb == 0) return;
	(void)bmh_ctx_reserve_staging(ctx, mb << 20, mb << 20);
	(void)bmh_ctx_reserve_device(ctx, mb << 20, (int64_t)(mb << 10) * 2, (mb << 20) / 8);
	{ /* the kernels' own workspaces for batches of the shim's size: 40 k reads of up to 160 bases per seeding batch, 70 k
	   * global alignments of up to 192 rows per slice of phase 2 ($BMH_RESERVE_KERNELS=0: grow on demand) */
		const char *k = getenv("BMH_RESERVE_KERNELS");
		if (!(k && k[0] == '0')) (void)bmh_ctx_reserve_kernels(ctx, (int)(mb * 2500), 160, (int64_t)mb * 4400, 192);
	}
}

void bmh_tls_die(const char *msg, int code)
{
	fprintf(stderr, "[bwamem_hip] fatal: %s (%s)\n", msg ? msg : "?", bmh_strerror(code));
	abort();
}

bmh_ctx_t *bmh_pool_get(const bmh_params_t *p)
{
	slot_t *s = 0;
	int i, rc;
	pthread_mutex_lock(&g_mu);
	for (i = 0; i < g_n && !s; ++i) /* prefer an idle context that already carries these parameters */
		if (!g_slots[i].busy && g_slots[i].have && memcmp(&g_slots[i].params, p, sizeof(*p)) == 0) s = &g_slots[i];
	for (i = 0; i < g_n && !s; ++i)
		if (!g_slots[i].busy) s = &g_slots[i];
	if (!s) {
		if (g_n == BMH_POOL_MAX) bmh_tls_die("context pool exhausted", BMH_E_NOMEM);
		s = &g_slots[g_n++];
	}
	s->busy = 1;
	pthread_mutex_unlock(&g_mu);
	if (!s->ctx) {
		if ((rc = bmh_ctx_create(&s->ctx, pool_device((int)(s - g_slots))))) bmh_tls_die("cannot create a GPU context", rc);
		pool_reserve(s->ctx);
	}
	if (!s->have || memcmp(&s->params, p, sizeof(*p)) != 0) {
		if ((rc = bmh_ctx_set_params(s->ctx, p))) bmh_tls_die(bmh_last_error(s->ctx), rc);
		s->params = *p, s->have = 1;
	}
	return s->ctx;
}

void bmh_pool_put(bmh_ctx_t *ctx)
{
	int i;
	pthread_mutex_lock(&g_mu);
	for (i = 0; i < g_n; ++i)
		if (g_slots[i].ctx == ctx) g_slots[i].busy = 0;
	pthread_mutex_unlock(&g_mu);
}

/* Create `n` idle contexts ahead of use (called from a background thread while the host program is still loading its
 * index): context creation costs tens of milliseconds each and contends inside the runtime when many threads do it at
 * once in the middle of the first chunk. */
static volatile int g_stop; /* the process is going down: stop creating contexts (see bmh_pool_stop) */
void bmh_pool_stop(void) { g_stop = 1; }

void bmh_pool_prewarm(int n)
{
	int k;
	for (k = 0; k < n && !g_stop; ++k) {
		bmh_ctx_t *ctx = 0;
		slot_t *s = 0;
		if (bmh_ctx_create(&ctx, pool_device(k))) return; /* no GPU: the first real call will say so loudly */
		pool_reserve(ctx);
		pthread_mutex_lock(&g_mu);
		if (g_n < BMH_POOL_MAX) s = &g_slots[g_n++], s->ctx = ctx, s->have = 0, s->busy = 0;
		pthread_mutex_unlock(&g_mu);
		if (!s) { bmh_ctx_destroy(ctx); return; }
	}
}
