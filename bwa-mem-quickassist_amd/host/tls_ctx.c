/* tls_ctx.c -- see tls_ctx.h */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tls_ctx.h"

static __thread bmh_ctx_t *tls_ctx_[2];
static __thread bmh_params_t tls_params_[2];
static __thread int tls_have_[2];

void bmh_tls_die(const char *msg, int code)
{
	fprintf(stderr, "[bwamem_hip] fatal: %s (%s)\n", msg ? msg : "?", bmh_strerror(code));
	abort();
}

bmh_ctx_t *bmh_tls_ctx_slot(const bmh_params_t *p, int slot)
{
	int rc;
	if (!tls_ctx_[slot]) {
		const char *dev = getenv("BMH_DEVICE");
		if ((rc = bmh_ctx_create(&tls_ctx_[slot], dev ? atoi(dev) : 0))) bmh_tls_die("cannot create a GPU context", rc);
	}
	if (!tls_have_[slot] || memcmp(&tls_params_[slot], p, sizeof(*p)) != 0) {
		if ((rc = bmh_ctx_set_params(tls_ctx_[slot], p))) bmh_tls_die(bmh_last_error(tls_ctx_[slot]), rc);
		tls_params_[slot] = *p, tls_have_[slot] = 1;
	}
	return tls_ctx_[slot];
}

bmh_ctx_t *bmh_tls_ctx(const bmh_params_t *p) { return bmh_tls_ctx_slot(p, 0); }
