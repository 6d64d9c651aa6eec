/* tls_ctx.c -- see tls_ctx.h */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tls_ctx.h"

static __thread bmh_ctx_t *tls_ctx;
static __thread bmh_params_t tls_params;
static __thread int tls_have;

void bmh_tls_die(const char *msg, int code)
{
	fprintf(stderr, "[bwamem_hip] fatal: %s (%s)\n", msg ? msg : "?", bmh_strerror(code));
	abort();
}

bmh_ctx_t *bmh_tls_ctx(const bmh_params_t *p)
{
	int rc;
	if (!tls_ctx) {
		const char *dev = getenv("BMH_DEVICE");
		if ((rc = bmh_ctx_create(&tls_ctx, dev ? atoi(dev) : 0))) bmh_tls_die("cannot create a GPU context", rc);
	}
	if (!tls_have || memcmp(&tls_params, p, sizeof(*p)) != 0) {
		if ((rc = bmh_ctx_set_params(tls_ctx, p))) bmh_tls_die(bmh_last_error(tls_ctx), rc);
		tls_params = *p, tls_have = 1;
	}
	return tls_ctx;
}
