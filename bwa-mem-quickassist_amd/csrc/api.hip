// api.hip -- the C-ABI of libbwamem_hip.so (include/bwamem_hip.h): contexts, parameter upload,
// host-buffer and device-resident batch entry points, static multi-GPU sharding.
// No CPU fallback lives here: if HIP is unusable every compute entry point returns an error.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <numeric>
#include <vector>

#include "bmh_ctx.h"

namespace bmh {

bmh_gate_fn g_gate_enter = nullptr, g_gate_leave = nullptr;
int g_wait_blocking = 0;

int set_hip_error(bmh_ctx *ctx, hipError_t e, const char *what)
{
	if (ctx) {
		ctx->last_error = std::string(what) + ": " + hipGetErrorString(e);
	}
	return BMH_E_HIP;
}

int ensure(bmh_ctx *ctx, DevBuf &b, size_t bytes)
{
	if (bytes <= b.cap) return BMH_OK;
	// growing means hipFree + hipMalloc, and both synchronise the whole device -- with many host threads in flight a
	// workspace that creeps up by a few per cent per batch stalls everybody again and again.  So: generous steps.
	size_t cap = std::max(bytes + bytes / 4, b.cap * 2);
	cap = (cap + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
	if (b.p) BMH_HIP(ctx, hipFree(b.p));
	b.p = nullptr, b.cap = 0;
	hipError_t e = hipMalloc(&b.p, cap);
	if (e != hipSuccess) {
		set_hip_error(ctx, e, "hipMalloc(workspace)");
		return BMH_E_NOMEM;
	}
	b.cap = cap;
	return BMH_OK;
}

int ensure_host(bmh_ctx *ctx, DevBuf &b, size_t bytes)
{
	if (bytes <= b.cap) return BMH_OK;
	size_t cap = std::max(bytes + bytes / 4, b.cap * 2); // pinned allocations are slow and synchronise too: see ensure()
	cap = (cap + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
	if (b.p) BMH_HIP(ctx, hipHostFree(b.p));
	b.p = nullptr, b.cap = 0;
	hipError_t e = hipHostMalloc(&b.p, cap, hipHostMallocDefault);
	if (e != hipSuccess) {
		set_hip_error(ctx, e, "hipHostMalloc(staging)");
		return BMH_E_NOMEM;
	}
	b.cap = cap;
	return BMH_OK;
}

static void free_buf(DevBuf &b)
{
	if (b.p) (void)hipFree(b.p);
	b.p = nullptr, b.cap = 0;
}

// read-and-clear the device error flag; the stream must be idle
static int fetch_err(bmh_ctx *ctx)
{
	BMH_HIP(ctx, hipMemcpyAsync(ctx->h_err, ctx->d_err, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
	BMH_HIP(ctx, stream_wait(ctx, ctx->stream));
	int e = *ctx->h_err;
	if (e != 0) {
		BMH_HIP(ctx, hipMemsetAsync(ctx->d_err, 0, sizeof(int), ctx->stream));
		BMH_HIP(ctx, stream_wait(ctx, ctx->stream));
		ctx->last_error = "a task was outside the supported range (see bwamem_hip.h)";
	}
	return e;
}

// The callers' host buffers are pageable.  Up to kStageMax bytes per direction and call they travel through the
// context's pinned staging buffers -- one memcpy on the host, then a DMA that does not depend on the runtime's own
// (slow, process-wide serialised) path for pageable memory; larger transfers go the direct way.
constexpr size_t kStageMax = (size_t)64 << 20;
struct Stager {
	bmh_ctx *ctx;
	bool up = false, down = false;
	size_t up_used = 0, down_used = 0;
	struct Pending {
		void *dst;
		size_t off, bytes;
	} pend[4];
	int n_pend = 0;
	int begin(bmh_ctx *c, size_t up_bytes, size_t down_bytes)
	{
		ctx = c;
		int rc;
		up = up_bytes <= kStageMax, down = down_bytes <= kStageMax;
		if (up && (rc = ensure_host(c, c->h_up, up_bytes + 256))) return rc;
		if (down && (rc = ensure_host(c, c->h_down, down_bytes + 256))) return rc;
		return BMH_OK;
	}
	int h2d(void *d_dst, const void *src, size_t bytes)
	{
		if (up) {
			uint8_t *h = (uint8_t *)ctx->h_up.p + up_used;
			memcpy(h, src, bytes);
			up_used += (bytes + 63) & ~(size_t)63, src = h;
		}
		BMH_HIP(ctx, hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
		return BMH_OK;
	}
	int d2h(void *dst, const void *d_src, size_t bytes)
	{
		void *to = dst;
		if (down && n_pend < 4) {
			to = (uint8_t *)ctx->h_down.p + down_used;
			pend[n_pend++] = Pending{dst, down_used, bytes};
			down_used += (bytes + 63) & ~(size_t)63;
		}
		BMH_HIP(ctx, hipMemcpyAsync(to, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
		return BMH_OK;
	}
	void finish() // after the stream has been synchronised
	{
		for (int k = 0; k < n_pend; ++k) memcpy(pend[k].dst, (const uint8_t *)ctx->h_down.p + pend[k].off, pend[k].bytes);
		n_pend = 0;
	}
};

} // namespace bmh

using namespace bmh;

extern "C" {

int bmh_version(void) { return BMH_VERSION; }

void bmh_set_wait_mode(int blocking) { bmh::g_wait_blocking = blocking != 0; }

int bmh_set_device_gate(bmh_gate_fn enter, bmh_gate_fn leave)
{
	if ((enter == nullptr) != (leave == nullptr)) return BMH_E_ARG;
	g_gate_enter = enter, g_gate_leave = leave;
	return BMH_OK;
}

const char *bmh_strerror(int code)
{
	switch (code) {
	case BMH_OK: return "ok";
	case BMH_E_NODEVICE: return "no usable HIP device";
	case BMH_E_HIP: return "HIP runtime error";
	case BMH_E_ARG: return "invalid argument";
	case BMH_E_RANGE: return "task outside the supported range";
	case BMH_E_NOMEM: return "out of memory";
	case BMH_E_CIGAR_CAP: return "CIGAR longer than the reserved slot range";
	default: return "unknown error";
	}
}

const char *bmh_last_error(const bmh_ctx_t *ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }

int bmh_device_count(int *n)
{
	int c = 0;
	if (!n) return BMH_E_ARG;
	if (hipGetDeviceCount(&c) != hipSuccess) {
		*n = 0;
		return BMH_E_NODEVICE;
	}
	*n = c;
	return BMH_OK;
}

int bmh_ctx_create(bmh_ctx_t **out, int device)
{
	if (!out) return BMH_E_ARG;
	*out = nullptr;
	int c = 0;
	if (hipGetDeviceCount(&c) != hipSuccess || c <= 0) return BMH_E_NODEVICE;
	if (device < 0 || device >= c) return BMH_E_ARG;
	bmh_ctx *ctx = new (std::nothrow) bmh_ctx();
	if (!ctx) return BMH_E_NOMEM;
	ctx->device = device;
	hipError_t e;
	if ((e = hipSetDevice(device)) == hipSuccess && g_wait_blocking) { // (also covers hipMemcpy / hipFree; refused on an active device by some runtimes: not an error)
		if (hipSetDeviceFlags(hipDeviceScheduleBlockingSync) != hipSuccess) (void)hipGetLastError();
	}
	if (e != hipSuccess || (e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)) != hipSuccess ||
	    (e = hipMalloc((void **)&ctx->d_err, sizeof(int))) != hipSuccess || (e = hipMemset(ctx->d_err, 0, sizeof(int))) != hipSuccess ||
	    (e = hipHostMalloc((void **)&ctx->h_err, sizeof(int), hipHostMallocDefault)) != hipSuccess ||
	    (e = hipEventCreate(&ctx->ev0)) != hipSuccess || (e = hipEventCreate(&ctx->ev1)) != hipSuccess ||
	    (e = hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming)) != hipSuccess ||
	    (e = hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming)) != hipSuccess ||
	    (e = hipEventCreateWithFlags(&ctx->ev_join2, hipEventDisableTiming)) != hipSuccess ||
	    (e = hipEventCreateWithFlags(&ctx->ev_wait, hipEventDisableTiming | hipEventBlockingSync)) != hipSuccess) {
		fprintf(stderr, "[bwamem_hip] context creation failed: %s\n", hipGetErrorString(e));
		bmh_ctx_destroy(ctx);
		return BMH_E_NODEVICE;
	}
	for (int b = 0; b <= kExtBinsMax; ++b)
		if (hipEventCreate(&ctx->ev_bin[b]) != hipSuccess || hipEventCreate(&ctx->ev_bin_end[b]) != hipSuccess) {
			bmh_ctx_destroy(ctx);
			return BMH_E_NODEVICE;
		}
	for (int b = 0; b < 5; ++b)
		if (hipEventCreate(&ctx->ev_sround[b]) != hipSuccess) {
			bmh_ctx_destroy(ctx);
			return BMH_E_NODEVICE;
		}
	for (int b = 0; b < 4; ++b)
		if (hipEventCreate(&ctx->ev_gbin[b]) != hipSuccess) {
			bmh_ctx_destroy(ctx);
			return BMH_E_NODEVICE;
		}
	ctx->stream = ctx->own_stream;
	(void)hipDeviceGetAttribute(&ctx->ncu, hipDeviceAttributeMultiprocessorCount, device);
	if (ctx->ncu <= 0) ctx->ncu = 256;
	if (const char *m = getenv("BMH_EXT_PERSIST")) ctx->ext_persist = atoi(m) != 0;
	if (const char *m = getenv("BMH_EXT_SPLIT96")) ctx->ext_split96 = atoi(m) != 0; // (A/B knob)
	if (const char *m = getenv("BMH_EXT_GRID_MULT")) ctx->ext_grid_mult = atoi(m) > 0 ? atoi(m) : ctx->ext_grid_mult;
	if (const char *m = getenv("BMH_EXT_SMALL")) ctx->small_batch = atoi(m) >= 0 ? atoi(m) : ctx->small_batch;
	if (getenv("BMH_EXT_MODE")) ctx->ext_mode_forced = true;
	if (const char *m = getenv("BMH_EXT_MODE")) ctx->force_kernel = !strcmp(m, "lds") ? 1 : !strcmp(m, "reg") ? 2 : !strcmp(m, "grp") ? 3 : !strcmp(m, "lanex4") ? 4 : 0;
	if (const char *m = getenv("BMH_GLB_MODE")) ctx->glb_mode = !strcmp(m, "wave") ? 1 : 0;
	if (const char *m = getenv("BMH_SW_MODE")) ctx->sw_mode = !strcmp(m, "generic") ? 1 : 0;
	if (const char *m = getenv("BMH_SW_WAVE")) ctx->sw_wave = atoi(m) != 0;
	if (const char *m = getenv("BMH_GL_FAST")) ctx->glb_fast = atoi(m) != 0;
	if (const char *m = getenv("BMH_GRID_MULT")) ctx->grid_mult = atoi(m) > 0 ? atoi(m) : 1;
	if (const char *m = getenv("BMH_EXT_SCHED")) ctx->ext_sched = atoi(m) >= 0 && atoi(m) <= 4 ? atoi(m) : -1;
	*out = ctx;
	return BMH_OK;
}

static void pac_release(bmh_ctx_t *ctx);
void free_bwt_binding(void *p); // fmindex.hip

int bmh_ctx_destroy(bmh_ctx_t *ctx)
{
	if (!ctx) return BMH_OK;
	(void)hipSetDevice(ctx->device);
	if (ctx->stream) (void)stream_wait(ctx, ctx->stream);
	free_buf(ctx->d_pool), free_buf(ctx->d_tasks), free_buf(ctx->d_res), free_buf(ctx->d_order);
	free_buf(ctx->d_cigar), free_buf(ctx->d_scratch), free_buf(ctx->d_bins), free_buf(ctx->d_zslab), free_buf(ctx->d_sw), free_buf(ctx->d_swrm);
	free_buf(ctx->d_seedws), free_buf(ctx->d_region);
	for (auto &h : ctx->hint) {
		if (h.ev) (void)hipEventDestroy(h.ev);
		if (h.h) (void)hipHostFree(h.h);
	}
	pac_release(ctx);
	if (ctx->bwt_bind) free_bwt_binding(ctx->bwt_bind);
	if (ctx->d_err) (void)hipFree(ctx->d_err);
	if (ctx->h_err) (void)hipHostFree(ctx->h_err);
	if (ctx->h_up.p) (void)hipHostFree(ctx->h_up.p);
	if (ctx->h_down.p) (void)hipHostFree(ctx->h_down.p);
	if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
	if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
	for (int b = 0; b <= kExtBinsMax; ++b)
	{
		if (ctx->ev_bin[b]) (void)hipEventDestroy(ctx->ev_bin[b]);
		if (ctx->ev_bin_end[b]) (void)hipEventDestroy(ctx->ev_bin_end[b]);
	}
	for (int b = 0; b < 4; ++b)
		if (ctx->ev_gbin[b]) (void)hipEventDestroy(ctx->ev_gbin[b]);
	for (int b = 0; b < 5; ++b)
		if (ctx->ev_sround[b]) (void)hipEventDestroy(ctx->ev_sround[b]);
	if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
	if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
	if (ctx->aux_stream) (void)hipStreamDestroy(ctx->aux_stream);
	if (ctx->ev_join2) (void)hipEventDestroy(ctx->ev_join2);
	if (ctx->ev_wait) (void)hipEventDestroy(ctx->ev_wait);
	if (ctx->aux2_stream) (void)hipStreamDestroy(ctx->aux2_stream);
	if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
	delete ctx;
	return BMH_OK;
}

int bmh_ctx_set_params(bmh_ctx_t *ctx, const bmh_params_t *p)
{
	if (!ctx || !p) return BMH_E_ARG;
	// e>=1: the reference divides by e (ksw.c:401,404); o_ins>=0: the scan form of F needs it
	if (p->e_del < 1 || p->e_ins < 1 || p->o_ins < 0 || p->o_del < 0) {
		ctx->last_error = "need e_del>=1, e_ins>=1, o_del>=0, o_ins>=0";
		return BMH_E_RANGE;
	}
	ctx->params = *p;
	DevParams &d = ctx->dev;
	d.o_del = p->o_del, d.e_del = p->e_del, d.o_ins = p->o_ins, d.e_ins = p->e_ins, d.zdrop = p->zdrop;
	d.max_mat = 0; // ksw.c:399-400 starts the maximum at 0
	int mn = 0;
	for (int i = 0; i < 25; ++i) d.max_mat = std::max(d.max_mat, (int)p->mat[i]), mn = std::min(mn, (int)p->mat[i]);
	d.bias = -mn;
	{
		int8_t lo = 127;
		for (int i = 0; i < 25; ++i) lo = std::min(lo, p->mat[i]);
		d.sw_shift = (uint8_t)(256 - (uint8_t)lo);
	}
	uint8_t bytes[28] = {0};
	memcpy(bytes, p->mat, 25);
	memcpy(d.matw, bytes, 28);
	ctx->have_params = true;
	return BMH_OK;
}

// The reference is immutable and large (hg38: 0.78 GB), and the reference program drives phase 1 from many host
// threads, each with its own context: one device copy per (device, host buffer) is shared by all of them.
struct PacShare {
	int device;
	const uint8_t *h;
	long long l_pac;
	void *d;
	int refs;
};
static std::mutex g_pac_mu;
static std::vector<PacShare> g_pacs;

static void pac_release(bmh_ctx_t *ctx)
{
	if (!ctx->h_pac) return;
	std::lock_guard<std::mutex> lk(g_pac_mu);
	for (size_t i = 0; i < g_pacs.size(); ++i)
		if (g_pacs[i].device == ctx->device && g_pacs[i].h == ctx->h_pac && g_pacs[i].l_pac == ctx->dev.l_pac) {
			if (--g_pacs[i].refs == 0) {
				(void)hipFree(g_pacs[i].d);
				g_pacs.erase(g_pacs.begin() + (long)i);
			}
			break;
		}
	ctx->h_pac = nullptr, ctx->dev.pac = nullptr, ctx->dev.l_pac = 0;
}

int bmh_ctx_set_pac(bmh_ctx_t *ctx, const uint8_t *pac, int64_t l_pac)
{
	if (!ctx || !pac || l_pac <= 0) return BMH_E_ARG;
	if (ctx->h_pac == pac && ctx->dev.l_pac == l_pac) return BMH_OK; // already resident
	BMH_HIP(ctx, hipSetDevice(ctx->device));
	pac_release(ctx);
	std::lock_guard<std::mutex> lk(g_pac_mu);
	for (auto &e : g_pacs)
		if (e.device == ctx->device && e.h == pac && e.l_pac == l_pac) {
			++e.refs;
			ctx->h_pac = pac, ctx->dev.pac = (const uint8_t *)e.d, ctx->dev.l_pac = l_pac;
			return BMH_OK;
		}
	const size_t bytes = (size_t)(l_pac / 4 + 1);
	void *d = nullptr;
	if (hipMalloc(&d, bytes + 16) != hipSuccess) {
		(void)hipGetLastError();
		ctx->last_error = "hipMalloc of " + std::to_string(bytes) + " bytes for the reference failed";
		return BMH_E_NOMEM;
	}
	if (hipMemcpy(d, pac, bytes, hipMemcpyHostToDevice) != hipSuccess) {
		(void)hipFree(d);
		ctx->last_error = "uploading the reference failed";
		return BMH_E_HIP;
	}
	g_pacs.push_back({ctx->device, pac, (long long)l_pac, d, 1});
	ctx->h_pac = pac, ctx->dev.pac = (const uint8_t *)d, ctx->dev.l_pac = l_pac;
	return BMH_OK;
}

// does the context hold exactly this reference?  (internal hook for the C drivers)
int bmh_ctx_has_pac_(const bmh_ctx_t *ctx, const uint8_t *pac, int64_t l_pac)
{
	return ctx && pac && ctx->h_pac == pac && ctx->dev.l_pac == l_pac;
}

int bmh_ctx_set_stream(bmh_ctx_t *ctx, void *s)
{
	if (!ctx) return BMH_E_ARG;
	ctx->stream = s ? (hipStream_t)s : ctx->own_stream;
	return BMH_OK;
}

int bmh_ctx_set_qcap(bmh_ctx_t *ctx, int qcap)
{
	if (!ctx || qcap < 1 || qcap > 65535) return BMH_E_ARG;
	ctx->qcap = qcap;
	return BMH_OK;
}

int bmh_ctx_sync(bmh_ctx_t *ctx)
{
	if (!ctx) return BMH_E_ARG;
	BMH_HIP(ctx, hipSetDevice(ctx->device));
	BMH_HIP(ctx, stream_wait(ctx, ctx->stream));
	return fetch_err(ctx);
}

int bmh_set_kernel_timing(bmh_ctx_t *ctx, int enable)
{
	if (!ctx) return BMH_E_ARG;
	ctx->timing = enable != 0;
	ctx->ev_valid = false;
	return BMH_OK;
}

int bmh_last_kernel_ms(bmh_ctx_t *ctx, float *ms)
{
	if (!ctx || !ms) return BMH_E_ARG;
	*ms = -1.f;
	if (!ctx->ev_valid) return BMH_OK;
	BMH_HIP(ctx, hipEventSynchronize(ctx->ev1));
	BMH_HIP(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
	return BMH_OK;
}

int bmh_last_extend_bin_ms(bmh_ctx_t *ctx, float ms[6])
{
	if (!ctx || !ms) return BMH_E_ARG;
	for (int b = 0; b < kExtBins; ++b) ms[b] = -1.f;
	if (!ctx->ev_bin_valid) return BMH_OK;
	BMH_HIP(ctx, hipEventSynchronize(ctx->ev1)); // recorded after both streams joined
	for (int b = 0; b < kExtBins; ++b) BMH_HIP(ctx, hipEventElapsedTime(&ms[b], ctx->ev_bin[b], ctx->ev_bin_end[b]));
	return BMH_OK;
}

int bmh_extend_bin_ms_sum(bmh_ctx_t *ctx, double ms[6], long long *launches, int reset)
{
	if (!ctx || !ms) return BMH_E_ARG;
	for (int b = 0; b < kExtBins; ++b) ms[b] = ctx->ext_bin_ms_sum[b];
	if (launches) *launches = ctx->ext_bin_launches;
	if (reset) {
		for (int b = 0; b < kExtBins; ++b) ctx->ext_bin_ms_sum[b] = 0.0;
		ctx->ext_bin_launches = 0;
	}
	return BMH_OK;
}

int bmh_last_seedext_round_ms(bmh_ctx_t *ctx, float ms[4])
{
	if (!ctx || !ms) return BMH_E_ARG;
	ms[0] = ms[1] = ms[2] = ms[3] = -1.f;
	if (!ctx->ev_sround_valid) return BMH_OK;
	BMH_HIP(ctx, hipEventSynchronize(ctx->ev_sround[4]));
	for (int b = 0; b < 4; ++b) BMH_HIP(ctx, hipEventElapsedTime(&ms[b], ctx->ev_sround[b], ctx->ev_sround[b + 1]));
	return BMH_OK;
}

int bmh_last_global_bin_ms(bmh_ctx_t *ctx, float ms[3])
{
	if (!ctx || !ms) return BMH_E_ARG;
	ms[0] = ms[1] = ms[2] = -1.f;
	if (!ctx->ev_gbin_valid) return BMH_OK;
	BMH_HIP(ctx, hipEventSynchronize(ctx->ev_gbin[3]));
	for (int b = 0; b < 3; ++b) BMH_HIP(ctx, hipEventElapsedTime(&ms[b], ctx->ev_gbin[b], ctx->ev_gbin[b + 1]));
	return BMH_OK;
}

// ------------------------------------------------------------------ extend

int bmh_extend_batch_device(bmh_ctx_t *ctx, const uint8_t *d_pool, const bmh_ext_task_t *d_tasks, int64_t n,
                            bmh_ext_result_t *d_res, const uint32_t *d_order)
{
	if (!ctx || n < 0 || (n > 0 && (!d_pool || !d_tasks || !d_res))) return BMH_E_ARG;
	if (!ctx->have_params) return BMH_E_ARG;
	if (n > 0xffffffffLL) return BMH_E_ARG;
	BMH_HIP(ctx, hipSetDevice(ctx->device));
	return launch_extend(ctx, d_pool, d_tasks, n, d_res, d_order, ctx->qcap);
}

static int validate_ext(bmh_ctx *ctx, const bmh_ext_task_t *t, int64_t n, size_t pool_bytes, int *qmax)
{
	int qm = 1;
	for (int64_t k = 0; k < n; ++k) {
		const bmh_ext_task_t &x = t[k];
		const bool qr = x.flags & BMH_F_QREV, tr = x.flags & BMH_F_TREV, tp = x.flags & BMH_F_TPAC;
		const uint64_t qlo = qr ? x.q_off - (x.qlen ? x.qlen - 1 : 0) : x.q_off, thi_len = x.tlen;
		const uint64_t tlo = tr ? x.t_off - (x.tlen ? x.tlen - 1 : 0) : x.t_off;
		const uint64_t tspace = tp ? (uint64_t)(ctx->dev.l_pac << 1) : (uint64_t)pool_bytes;
		if (tp && !ctx->dev.pac) {
			ctx->last_error = "task " + std::to_string(k) + " has BMH_F_TPAC but no reference was uploaded (bmh_ctx_set_pac)";
			return BMH_E_ARG;
		}
		if ((qr && x.qlen && x.q_off + 1 < x.qlen) || (tr && x.tlen && x.t_off + 1 < x.tlen) || qlo + x.qlen > pool_bytes ||
		    tlo + thi_len > tspace) {
			ctx->last_error = "task " + std::to_string(k) + " reads outside the sequence pool";
			return BMH_E_ARG;
		}
		const int h0 = x.h0 < 0 ? 0 : x.h0;
		if ((int64_t)h0 + (int64_t)x.qlen * ctx->dev.max_mat > kScoreLimit) {
			ctx->last_error = "task " + std::to_string(k) + ": h0 + qlen*max(mat) exceeds the 16-bit score range";
			return BMH_E_RANGE;
		}
		qm = std::max(qm, (int)x.qlen);
	}
	*qmax = qm;
	return BMH_OK;
}

int bmh_extend_batch(bmh_ctx_t *ctx, const uint8_t *pool, size_t pool_bytes, const bmh_ext_task_t *tasks, int64_t n,
                     bmh_ext_result_t *results)
{
	if (!ctx || n < 0 || (n > 0 && (!tasks || !results))) return BMH_E_ARG;
	if (!ctx->have_params) return BMH_E_ARG;
	if (n == 0) return BMH_OK;
	if (n > 0xffffffffLL) return BMH_E_ARG;
	const bool resident = pool == nullptr; // use the pool left on the device by bmh_upload_pool()
	if (resident) {
		if (!ctx->pool_resident) return BMH_E_ARG;
		pool_bytes = ctx->pool_bytes;
	}
	int qmax = 1, rc;
	if ((rc = validate_ext(ctx, tasks, n, pool_bytes, &qmax))) return rc;
	GateGuard gate;
	BMH_HIP(ctx, hipSetDevice(ctx->device));
	if (!resident) {
		ctx->pool_resident = false;
		if ((rc = ensure(ctx, ctx->d_pool, pool_bytes + 16))) return rc;
	}
	if ((rc = ensure(ctx, ctx->d_tasks, (size_t)n * sizeof(bmh_ext_task_t)))) return rc;
	if ((rc = ensure(ctx, ctx->d_res, (size_t)n * sizeof(bmh_ext_result_t)))) return rc;
	Stager st;
	if ((rc = st.begin(ctx, (resident ? 0 : pool_bytes + 64) + (size_t)n * sizeof(bmh_ext_task_t), (size_t)n * sizeof(bmh_ext_result_t)))) return rc;
	if (!resident && (rc = st.h2d(ctx->d_pool.p, pool, pool_bytes))) return rc;
	if ((rc = st.h2d(ctx->d_tasks.p, tasks, (size_t)n * sizeof(bmh_ext_task_t)))) return rc;
	// launch order: the dispatcher sorts the tasks on the device (bin, length bucket, row estimate)
	if ((rc = launch_extend(ctx, (const uint8_t *)ctx->d_pool.p, (const bmh_ext_task_t *)ctx->d_tasks.p, n,
	                        (bmh_ext_result_t *)ctx->d_res.p, nullptr, qmax)))
		return rc;
	if ((rc = st.d2h(results, ctx->d_res.p, (size_t)n * sizeof(bmh_ext_result_t)))) return rc;
	rc = fetch_err(ctx); // synchronises
	st.finish();
	return rc;
}

int bmh_upload_pool(bmh_ctx_t *ctx, const uint8_t *pool, size_t bytes)
{
	if (!ctx || !pool) return BMH_E_ARG;
	int rc;
	GateGuard gate;
	BMH_HIP(ctx, hipSetDevice(ctx->device));
	ctx->pool_resident = false;
	if ((rc = ensure(ctx, ctx->d_pool, bytes + 16))) return rc;
	Stager st;
	if ((rc = st.begin(ctx, bytes, 0)) || (rc = st.h2d(ctx->d_pool.p, pool, bytes))) return rc;
	BMH_HIP(ctx, stream_wait(ctx, ctx->stream)); // `pool` may be freed by the caller on return
	ctx->pool_resident = true, ctx->pool_bytes = bytes;
	return BMH_OK;
}

// internal hooks for host/chain2aln_batch.c (C cannot see inside bmh_ctx)
const bmh_params_t *bmh_ctx_params_(const bmh_ctx_t *ctx) { return ctx && ctx->have_params ? &ctx->params : nullptr; }
void bmh_ctx_set_driver_stats_(bmh_ctx_t *ctx, const bmh_driver_stats_t *st)
{
	if (ctx && st) ctx->dstats = *st;
}

int bmh_extend_batch_sharded(bmh_ctx_t *const *ctxs, int n_ctx, const uint8_t *pool, size_t pool_bytes,
                             const bmh_ext_task_t *tasks, int64_t n, bmh_ext_result_t *results)
{
	if (!ctxs || n_ctx < 1 || n < 0) return BMH_E_ARG;
	for (int g = 0; g < n_ctx; ++g)
		if (!ctxs[g] || !ctxs[g]->have_params) return BMH_E_ARG;
	if (n == 0) return BMH_OK;
	// contiguous static split; every shard gets the whole pool (offsets stay valid), its own task slice
	std::vector<int64_t> lo((size_t)n_ctx + 1);
	for (int g = 0; g <= n_ctx; ++g) lo[(size_t)g] = n * g / n_ctx;
	int rc;
	for (int g = 0; g < n_ctx; ++g) { // enqueue everything on every device before waiting on any
		bmh_ctx *c = ctxs[g];
		const int64_t m = lo[(size_t)g + 1] - lo[(size_t)g];
		if (m == 0) continue;
		const bmh_ext_task_t *t = tasks + lo[(size_t)g];
		int qmax = 1;
		if ((rc = validate_ext(c, t, m, pool_bytes, &qmax))) {
			for (int h = 0; h < g; ++h) {
				(void)hipSetDevice(ctxs[h]->device);
				(void)stream_wait(ctxs[h], ctxs[h]->stream);
			}
			return rc;
		}
		BMH_HIP(c, hipSetDevice(c->device));
		c->pool_resident = false; // the context's pool is overwritten below
		rc = BMH_OK;
		if (!rc) rc = ensure(c, c->d_pool, pool_bytes + 16);
		if (!rc) rc = ensure(c, c->d_tasks, (size_t)m * sizeof(bmh_ext_task_t));
		if (!rc) rc = ensure(c, c->d_res, (size_t)m * sizeof(bmh_ext_result_t));
		if (!rc && hipMemcpyAsync(c->d_pool.p, pool, pool_bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = BMH_E_HIP;
		if (!rc && hipMemcpyAsync(c->d_tasks.p, t, (size_t)m * sizeof(bmh_ext_task_t), hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = BMH_E_HIP;
		if (!rc)
			rc = launch_extend(c, (const uint8_t *)c->d_pool.p, (const bmh_ext_task_t *)c->d_tasks.p, m, (bmh_ext_result_t *)c->d_res.p, nullptr,
			                   qmax);
		if (!rc && hipMemcpyAsync(results + lo[(size_t)g], c->d_res.p, (size_t)m * sizeof(bmh_ext_result_t), hipMemcpyDeviceToHost, c->stream) !=
		               hipSuccess)
			rc = BMH_E_HIP;
		if (rc) { // copies into the caller's `results` may be in flight on the devices already served: drain them first
			for (int h = 0; h <= g; ++h) {
				(void)hipSetDevice(ctxs[h]->device);
				(void)stream_wait(ctxs[h], ctxs[h]->stream);
			}
			return rc;
		}
	}
	int first = BMH_OK;
	for (int g = 0; g < n_ctx; ++g) {
		if (lo[(size_t)g + 1] == lo[(size_t)g]) continue;
		BMH_HIP(ctxs[g], hipSetDevice(ctxs[g]->device));
		rc = fetch_err(ctxs[g]);
		if (rc && !first) first = rc;
	}
	return first;
}

// ------------------------------------------------------------------ fused per-seed extension (row a5)

int bmh_seedext_batch_device(bmh_ctx_t *ctx, const uint8_t *d_pool, const bmh_seed_task_t *d_tasks, int64_t n,
                             bmh_seed_result_t *d_res)
{
	if (!ctx || n < 0 || (n > 0 && (!d_pool || !d_tasks || !d_res))) return BMH_E_ARG;
	if (!ctx->have_params) return BMH_E_ARG;
	if (n > 0x7fffffffLL) return BMH_E_ARG;
	BMH_HIP(ctx, hipSetDevice(ctx->device));
	return launch_seedext(ctx, d_pool, d_tasks, n, d_res, ctx->qcap);
}

static int validate_seeds(bmh_ctx *ctx, const bmh_seed_task_t *t, int64_t n, size_t pool_bytes, int *qmax)
{
	int qm = 1;
	const int smax = std::max(ctx->dev.max_mat, ctx->params.a);
	for (int64_t k = 0; k < n; ++k) {
		const bmh_seed_task_t &x = t[k];
		const int64_t rq = (int64_t)x.l_query - x.qbeg - x.len, rt = (int64_t)x.wlen - x.rbeg - x.len;
		const bool tp = x.flags & BMH_F_TPAC;
		if (tp && !ctx->dev.pac) {
			ctx->last_error = "seed " + std::to_string(k) + " has BMH_F_TPAC but no reference was uploaded (bmh_ctx_set_pac)";
			return BMH_E_ARG;
		}
		if (x.l_query < 1 || x.qbeg < 0 || x.len < 1 || rq < 0 || x.rbeg < 0 || rt < 0 || x.wlen < 0) {
			ctx->last_error = "seed " + std::to_string(k) + " does not lie inside its read and window";
			return BMH_E_ARG;
		}
		const uint64_t tspace = tp ? (uint64_t)(ctx->dev.l_pac << 1) : (uint64_t)pool_bytes;
		if (x.q_off + (uint64_t)x.l_query > pool_bytes || x.t_off + (uint64_t)x.wlen > tspace) {
			ctx->last_error = "seed " + std::to_string(k) + " reads outside the sequence pool";
			return BMH_E_ARG;
		}
		if (x.qbeg > 65535 || rq > 65535 || x.rbeg > 65535 || rt > 65535 || (int64_t)x.l_query * smax > kScoreLimit) {
			ctx->last_error = "seed " + std::to_string(k) + ": flank longer than 65535 or scores beyond the 16-bit range";
			return BMH_E_RANGE;
		}
		qm = std::max(qm, std::max(x.qbeg, (int)rq));
	}
	*qmax = qm;
	return BMH_OK;
}

int bmh_seedext_submit(bmh_ctx_t *ctx, const bmh_seed_task_t *tasks, int64_t n)
{
	if (!ctx || n < 0 || (n > 0 && !tasks)) return BMH_E_ARG;
	if (!ctx->have_params || !ctx->pool_resident || ctx->seed_pending_n >= 0) return BMH_E_ARG;
	if (n > 0x7fffffffLL) return BMH_E_ARG;
	int qmax = 1, rc;
	if ((rc = validate_seeds(ctx, tasks, n, ctx->pool_bytes, &qmax))) return rc;
	BMH_HIP(ctx, hipSetDevice(ctx->device));
	const size_t tb = (size_t)n * sizeof(bmh_seed_task_t), rb = (size_t)n * sizeof(bmh_seed_result_t);
	if ((rc = ensure(ctx, ctx->d_tasks, tb + 64)) || (rc = ensure(ctx, ctx->d_res, rb + 64))) return rc;
	if ((rc = ensure_host(ctx, ctx->h_up, tb + 256)) || (rc = ensure_host(ctx, ctx->h_down, rb + 256))) return rc;
	if (n > 0) {
		memcpy(ctx->h_up.p, tasks, tb);
		BMH_HIP(ctx, hipMemcpyAsync(ctx->d_tasks.p, ctx->h_up.p, tb, hipMemcpyHostToDevice, ctx->stream));
		if ((rc = launch_seedext(ctx, (const uint8_t *)ctx->d_pool.p, (const bmh_seed_task_t *)ctx->d_tasks.p, n,
		                         (bmh_seed_result_t *)ctx->d_res.p, qmax)))
			return rc;
		BMH_HIP(ctx, hipMemcpyAsync(ctx->h_down.p, ctx->d_res.p, rb, hipMemcpyDeviceToHost, ctx->stream));
		BMH_HIP(ctx, hipMemcpyAsync((uint8_t *)ctx->h_down.p + ((rb + 63) & ~(size_t)63), seedext_counters(ctx), 16, hipMemcpyDeviceToHost,
		                            ctx->stream));
	}
	ctx->seed_pending_n = n;
	return BMH_OK;
}

int bmh_seedext_wait(bmh_ctx_t *ctx, bmh_seed_result_t *results)
{
	if (!ctx || ctx->seed_pending_n < 0 || (ctx->seed_pending_n > 0 && !results)) return BMH_E_ARG;
	const int64_t n = ctx->seed_pending_n;
	ctx->seed_pending_n = -1;
	if (n == 0) return BMH_OK;
	BMH_HIP(ctx, hipSetDevice(ctx->device));
	const int rc = fetch_err(ctx); // synchronises
	const size_t rb = (size_t)n * sizeof(bmh_seed_result_t);
	memcpy(results, ctx->h_down.p, rb);
	const uint32_t *c = (const uint32_t *)((const uint8_t *)ctx->h_down.p + ((rb + 63) & ~(size_t)63));
	ctx->sstats.seeds = n, ctx->sstats.left_tasks = c[0], ctx->sstats.left_retries = c[1], ctx->sstats.right_tasks = c[2],
	ctx->sstats.right_retries = c[3];
	return rc;
}

int bmh_seedext_batch(bmh_ctx_t *ctx, const uint8_t *pool, size_t pool_bytes, const bmh_seed_task_t *tasks, int64_t n,
                      bmh_seed_result_t *results)
{
	if (!ctx || n < 0 || (n > 0 && (!tasks || !results))) return BMH_E_ARG;
	if (!ctx->have_params) return BMH_E_ARG;
	if (n == 0) return BMH_OK;
	int rc;
	if (pool && (rc = bmh_upload_pool(ctx, pool, pool_bytes))) return rc;
	GateGuard gate;
	if ((rc = bmh_seedext_submit(ctx, tasks, n))) return rc;
	return bmh_seedext_wait(ctx, results);
}

int bmh_seedext_stats(const bmh_ctx_t *ctx, bmh_seedext_stats_t *st)
{
	if (!ctx || !st) return BMH_E_ARG;
	*st = ctx->sstats;
	return BMH_OK;
}

// ------------------------------------------------------------------ global

struct GlbShape {
	int qmax = 1, tmax = 1, wmax = 0, wraw = 0;
	size_t cig_lo = ~(size_t)0, cig_hi = 0; // the words of the CIGAR pool the tasks may write
};
static int validate_glb(bmh_ctx *ctx, const bmh_glb_task_t *tasks, int64_t n, size_t pool_bytes, bool have_cigar_pool, size_t cigar_words, GlbShape *o)
{
	GlbShape g;
	for (int64_t k = 0; k < n; ++k) {
		const bmh_glb_task_t &x = tasks[k];
		if (x.q_off + x.qlen > pool_bytes || x.t_off + x.tlen > pool_bytes || x.w < 0 ||
		    (x.cigar_cap && (!have_cigar_pool || (size_t)x.cigar_off + x.cigar_cap > cigar_words))) {
			ctx->last_error = "global task " + std::to_string(k) + " has out-of-range offsets";
			return BMH_E_ARG;
		}
		g.qmax = std::max(g.qmax, (int)x.qlen), g.tmax = std::max(g.tmax, (int)x.tlen);
		g.wmax = std::max(g.wmax, std::min(x.w, (int)x.qlen)); // only min(qlen,2w+1) columns are ever stored
		g.wraw = std::max(g.wraw, x.w);                        // ... but the device bins the tasks by their w as given
		if (x.cigar_cap) g.cig_lo = std::min(g.cig_lo, (size_t)x.cigar_off), g.cig_hi = std::max(g.cig_hi, (size_t)x.cigar_off + x.cigar_cap);
	}
	*o = g;
	return BMH_OK;
}

int bmh_global_batch_device(bmh_ctx_t *ctx, const uint8_t *d_pool, const bmh_glb_task_t *d_tasks, int64_t n,
                            bmh_glb_result_t *d_res, uint32_t *d_cigar, const uint32_t *d_order)
{
	if (!ctx || n < 0 || (n > 0 && (!d_pool || !d_tasks || !d_res))) return BMH_E_ARG;
	if (!ctx->have_params) return BMH_E_ARG;
	if (n > 0xffffffffLL) return BMH_E_ARG;
	BMH_HIP(ctx, hipSetDevice(ctx->device));
	// no host view of the tasks: size for the context's capacity hint (square band-limited matrix)
	return launch_global(ctx, d_pool, d_tasks, n, d_res, d_cigar, d_order, ctx->qcap, ctx->qcap + 2 * ctx->params.w + 64,
	                     std::max(ctx->params.w * 4, 100), std::max(ctx->params.w * 4, 100));
}

int bmh_global_batch(bmh_ctx_t *ctx, const uint8_t *pool, size_t pool_bytes, const bmh_glb_task_t *tasks, int64_t n,
                     bmh_glb_result_t *results, uint32_t *cigar_pool, size_t cigar_words)
{
	if (!ctx || n < 0 || (n > 0 && (!tasks || !results))) return BMH_E_ARG;
	if (!ctx->have_params) return BMH_E_ARG;
	if (n == 0) return BMH_OK;
	if (n > 0xffffffffLL) return BMH_E_ARG;
	const bool resident = pool == nullptr; // use the pool left on the device by bmh_upload_pool()
	if (resident) {
		if (!ctx->pool_resident) return BMH_E_ARG;
		pool_bytes = ctx->pool_bytes;
	}
	GlbShape gs;
	int rc;
	if ((rc = validate_glb(ctx, tasks, n, pool_bytes, cigar_pool != nullptr, cigar_words, &gs))) return rc;
	const int qmax = gs.qmax, tmax = gs.tmax, wmax = gs.wmax, wraw = gs.wraw;
	GateGuard gate;
	BMH_HIP(ctx, hipSetDevice(ctx->device));
	if (!resident) {
		ctx->pool_resident = false;
		if ((rc = ensure(ctx, ctx->d_pool, pool_bytes + 16))) return rc;
	}
	if ((rc = ensure(ctx, ctx->d_tasks, (size_t)n * sizeof(bmh_glb_task_t)))) return rc;
	if ((rc = ensure(ctx, ctx->d_res, (size_t)n * sizeof(bmh_glb_result_t)))) return rc;
	if ((rc = ensure(ctx, ctx->d_cigar, (cigar_words + 4) * 4))) return rc;
	Stager st;
	if ((rc = st.begin(ctx, (resident ? 0 : pool_bytes + 64) + (size_t)n * sizeof(bmh_glb_task_t),
	                   (size_t)n * sizeof(bmh_glb_result_t) + 64 + cigar_words * 4)))
		return rc;
	if (!resident && (rc = st.h2d(ctx->d_pool.p, pool, pool_bytes))) return rc;
	if ((rc = st.h2d(ctx->d_tasks.p, tasks, (size_t)n * sizeof(bmh_glb_task_t)))) return rc;
	if ((rc = launch_global(ctx, (const uint8_t *)ctx->d_pool.p, (const bmh_glb_task_t *)ctx->d_tasks.p, n,
	                        (bmh_glb_result_t *)ctx->d_res.p, (uint32_t *)ctx->d_cigar.p, nullptr, qmax, tmax, wmax, wraw)))
		return rc;
	if ((rc = st.d2h(results, ctx->d_res.p, (size_t)n * sizeof(bmh_glb_result_t)))) return rc;
	if (cigar_words && (rc = st.d2h(cigar_pool, ctx->d_cigar.p, cigar_words * 4))) return rc;
	rc = fetch_err(ctx);
	st.finish();
	return rc;
}

// ------------------------------------------------------------------ the region record (bwa_gen_cigar2 around ksw_global2)

int bmh_region_cigar_batch(bmh_ctx_t *ctx, const uint8_t *readpool, size_t readpool_bytes, size_t opool_bytes, const bmh_region_req_t *reqs,
                           int64_t n_req, const bmh_glb_task_t *tasks, int64_t n_tasks, size_t task_cigar_words, int cig_cap, int md_cap,
                           bmh_region_res_t *results, uint32_t *cigar_out, char *md_out)
{
	if (!ctx || n_req < 0 || n_tasks < 0 || cig_cap < 1 || md_cap < 1) return BMH_E_ARG;
	if (n_req > 0 && (!readpool || !reqs || !results || !cigar_out || !md_out)) return BMH_E_ARG;
	if (n_tasks > 0 && !tasks) return BMH_E_ARG;
	if (!ctx->have_params) return BMH_E_ARG;
	if (!ctx->dev.pac) {
		ctx->last_error = "bmh_region_cigar_batch needs the resident reference (bmh_ctx_set_pac)";
		return BMH_E_ARG;
	}
	if (n_req == 0) return BMH_OK;
	if (n_req > 0x7fffffffLL || n_tasks > 0xffffffffLL) return BMH_E_ARG;
	const int64_t l_pac = ctx->dev.l_pac;
	for (int64_t k = 0; k < n_req; ++k) { // every byte the kernels will address
		const bmh_region_req_t &r = reqs[k];
		bool ok = r.ql >= 1 && r.tl >= 1 && r.ql <= 65535 && r.tl <= 65535 && r.q_src + (uint64_t)r.ql <= readpool_bytes &&
		          r.o_off + (uint64_t)r.ql + (uint64_t)r.tl <= opool_bytes && r.rb >= 0 && r.rb + r.tl <= l_pac << 1 &&
		          !(r.rb < l_pac && r.rb + r.tl > l_pac);
		const int nt = r.truesc == INT32_MIN ? 1 : 3;
		if (ok && r.task[0] >= 0)
			for (int t = 0; t < nt; ++t) ok = ok && r.task[t] >= 0 && r.task[t] < n_tasks;
		if (ok && r.task[0] < 0) ok = r.ql == r.tl;
		if (!ok) {
			ctx->last_error = "region " + std::to_string(k) + " is outside its pools, the reference or the task list";
			return BMH_E_ARG;
		}
	}
	GlbShape gs;
	int rc;
	if (n_tasks > 0 && (rc = validate_glb(ctx, tasks, n_tasks, opool_bytes, true, task_cigar_words, &gs))) return rc;
	GateGuard gate;
	BMH_HIP(ctx, hipSetDevice(ctx->device));
	// device pool = [oriented copies | the query windows as uploaded]
	const size_t rpool_off = (opool_bytes + 16 + 63) & ~(size_t)63;
	ctx->pool_resident = false;
	if ((rc = ensure(ctx, ctx->d_pool, rpool_off + readpool_bytes + 16))) return rc;
	if ((rc = ensure(ctx, ctx->d_tasks, (size_t)std::max<int64_t>(n_tasks, 1) * sizeof(bmh_glb_task_t)))) return rc;
	if ((rc = ensure(ctx, ctx->d_res, (size_t)std::max<int64_t>(n_tasks, 1) * sizeof(bmh_glb_result_t)))) return rc;
	if ((rc = ensure(ctx, ctx->d_cigar, (task_cigar_words + 4) * 4))) return rc;
	const size_t req_b = ((size_t)n_req * sizeof(bmh_region_req_t) + 255) & ~(size_t)255;
	const size_t res_b = ((size_t)n_req * sizeof(bmh_region_res_t) + 255) & ~(size_t)255;
	const size_t cig_b = ((size_t)n_req * (size_t)cig_cap * 4 + 255) & ~(size_t)255;
	const size_t md_b = ((size_t)n_req * (size_t)md_cap + 255) & ~(size_t)255;
	if ((rc = ensure(ctx, ctx->d_region, req_b + res_b + cig_b + md_b))) return rc;
	uint8_t *d_reg = (uint8_t *)ctx->d_region.p;
	bmh_region_req_t *d_reqs = (bmh_region_req_t *)d_reg;
	bmh_region_res_t *d_rres = (bmh_region_res_t *)(d_reg + req_b);
	uint32_t *d_cout = (uint32_t *)(d_reg + req_b + res_b);
	char *d_md = (char *)(d_reg + req_b + res_b + cig_b);
	Stager st;
	if ((rc = st.begin(ctx, readpool_bytes + 64 + (size_t)n_req * sizeof(bmh_region_req_t) + 64 + (size_t)n_tasks * sizeof(bmh_glb_task_t) + 64,
	                   res_b + cig_b + md_b)))
		return rc;
	if ((rc = st.h2d((uint8_t *)ctx->d_pool.p + rpool_off, readpool, readpool_bytes))) return rc;
	if ((rc = st.h2d(d_reqs, reqs, (size_t)n_req * sizeof(bmh_region_req_t)))) return rc;
	if (n_tasks > 0 && (rc = st.h2d(ctx->d_tasks.p, tasks, (size_t)n_tasks * sizeof(bmh_glb_task_t)))) return rc;
	if ((rc = launch_region_orient(ctx, (uint8_t *)ctx->d_pool.p, rpool_off, d_reqs, n_req))) return rc;
	if (n_tasks > 0 && (rc = launch_global(ctx, (const uint8_t *)ctx->d_pool.p, (const bmh_glb_task_t *)ctx->d_tasks.p, n_tasks,
	                                       (bmh_glb_result_t *)ctx->d_res.p, (uint32_t *)ctx->d_cigar.p, nullptr, gs.qmax, gs.tmax, gs.wmax, gs.wraw)))
		return rc;
	if ((rc = launch_region_finish(ctx, (const uint8_t *)ctx->d_pool.p, d_reqs, n_req, (const bmh_glb_task_t *)ctx->d_tasks.p,
	                               (const bmh_glb_result_t *)ctx->d_res.p, (const uint32_t *)ctx->d_cigar.p, d_rres, d_cout, cig_cap, d_md, md_cap)))
		return rc;
	if ((rc = st.d2h(results, d_rres, (size_t)n_req * sizeof(bmh_region_res_t)))) return rc;
	if ((rc = st.d2h(cigar_out, d_cout, (size_t)n_req * (size_t)cig_cap * 4))) return rc;
	if ((rc = st.d2h(md_out, d_md, (size_t)n_req * (size_t)md_cap))) return rc;
	rc = fetch_err(ctx);
	st.finish();
	return rc == BMH_E_CIGAR_CAP ? BMH_OK : rc; // a task that outgrew its slots is reported per region (BMH_REGION_CIGAR_CUT)
}

// ------------------------------------------------------------------ local Smith-Waterman (ksw_align2)

int bmh_sw_batch_device(bmh_ctx_t *ctx, const uint8_t *d_pool, const bmh_sw_task_t *d_tasks, int64_t n,
                        bmh_sw_result_t *d_res)
{
	if (!ctx || n < 0 || (n > 0 && (!d_pool || !d_tasks || !d_res))) return BMH_E_ARG;
	if (!ctx->have_params) return BMH_E_ARG;
	if (n > 0xffffffffLL) return BMH_E_ARG;
	BMH_HIP(ctx, hipSetDevice(ctx->device));
	return launch_sw(ctx, d_pool, d_tasks, n, d_res, -1, -1, -1);
}

static int validate_sw(bmh_ctx *ctx, const bmh_sw_task_t *tasks, int64_t n, size_t pool_bytes, int *qmax_, int *tmax_, int *qmin_)
{
	int qmax = 1, tmax = 1, qmin = 65535;
	for (int64_t k = 0; k < n; ++k) {
		const bmh_sw_task_t &x = tasks[k];
		qmin = std::min(qmin, (int)x.qlen);
		const bool qr = x.flags & BMH_F_QREV, tr = x.flags & BMH_F_TREV, tp = x.flags & BMH_F_TPAC;
		const uint64_t qlo = qr ? x.q_off - (x.qlen ? x.qlen - 1 : 0) : x.q_off;
		const uint64_t tlo = tr ? x.t_off - (x.tlen ? x.tlen - 1 : 0) : x.t_off;
		const uint64_t tspace = tp ? (uint64_t)(ctx->dev.l_pac << 1) : (uint64_t)pool_bytes;
		if (tp && !ctx->dev.pac) {
			ctx->last_error = "task " + std::to_string(k) + " has BMH_F_TPAC but no reference was uploaded (bmh_ctx_set_pac)";
			return BMH_E_ARG;
		}
		if ((qr && x.qlen && x.q_off + 1 < x.qlen) || (tr && x.tlen && x.t_off + 1 < x.tlen) || qlo + x.qlen > pool_bytes ||
		    tlo + x.tlen > tspace || x.tlen > 0x7fffffffu) {
			ctx->last_error = "Smith-Waterman task " + std::to_string(k) + " reads outside the sequence pool";
			return BMH_E_ARG;
		}
		if (x.qlen < 1 || (int64_t)x.qlen * ctx->dev.max_mat >= kScoreLimit) {
			ctx->last_error = "Smith-Waterman task " + std::to_string(k) + ": qlen must be >= 1 and qlen*max(mat) below the 16-bit score range";
			return BMH_E_RANGE;
		}
		qmax = std::max(qmax, (int)x.qlen), tmax = std::max(tmax, (int)x.tlen);
	}
	*qmax_ = qmax, *tmax_ = tmax, *qmin_ = qmin;
	return BMH_OK;
}

int bmh_sw_batch(bmh_ctx_t *ctx, const uint8_t *pool, size_t pool_bytes, const bmh_sw_task_t *tasks, int64_t n,
                 bmh_sw_result_t *results)
{
	if (!ctx || n < 0 || (n > 0 && (!tasks || !results))) return BMH_E_ARG;
	if (!ctx->have_params) return BMH_E_ARG;
	if (n == 0) return BMH_OK;
	if (n > 0xffffffffLL) return BMH_E_ARG;
	const bool resident = pool == nullptr; // use the pool left on the device by bmh_upload_pool()
	if (resident) {
		if (!ctx->pool_resident) return BMH_E_ARG;
		pool_bytes = ctx->pool_bytes;
	}
	int qmax = 1, tmax = 1, qmin = 65535, rc;
	if ((rc = validate_sw(ctx, tasks, n, pool_bytes, &qmax, &tmax, &qmin))) return rc;
	GateGuard gate;
	BMH_HIP(ctx, hipSetDevice(ctx->device));
	if (!resident) {
		ctx->pool_resident = false;
		if ((rc = ensure(ctx, ctx->d_pool, pool_bytes + 16))) return rc;
	}
	if ((rc = ensure(ctx, ctx->d_tasks, (size_t)n * sizeof(bmh_sw_task_t)))) return rc;
	if ((rc = ensure(ctx, ctx->d_res, (size_t)n * sizeof(bmh_sw_result_t)))) return rc;
	Stager st;
	if ((rc = st.begin(ctx, (resident ? 0 : pool_bytes + 64) + (size_t)n * sizeof(bmh_sw_task_t), (size_t)n * sizeof(bmh_sw_result_t)))) return rc;
	if (!resident && (rc = st.h2d(ctx->d_pool.p, pool, pool_bytes))) return rc;
	if ((rc = st.h2d(ctx->d_tasks.p, tasks, (size_t)n * sizeof(bmh_sw_task_t)))) return rc;
	if ((rc = launch_sw(ctx, (const uint8_t *)ctx->d_pool.p, (const bmh_sw_task_t *)ctx->d_tasks.p, n,
	                    (bmh_sw_result_t *)ctx->d_res.p, qmax, tmax, qmin)))
		return rc;
	if ((rc = st.d2h(results, ctx->d_res.p, (size_t)n * sizeof(bmh_sw_result_t)))) return rc;
	rc = fetch_err(ctx); // synchronises
	st.finish();
	return rc;
}

// ------------------------------------------------------------------ static shards of the other batches
// One contiguous slice of the tasks per context (one context per GPU), every device given the whole sequence pool (offsets stay
// valid), everything -- uploads, kernels, downloads -- enqueued on every device's stream before any of them is waited for; no
// collective, no device-to-device traffic.  The same split as kt_for_batch's ranges (reference kthread_batch.c:44-56, bwamem.c:1313).
} // extern "C"
static void drain_shards(bmh_ctx_t *const *ctxs, int upto)
{
	for (int h = 0; h <= upto; ++h) {
		(void)hipSetDevice(ctxs[h]->device);
		(void)stream_wait(ctxs[h], ctxs[h]->stream);
	}
}
template <class Enqueue> static int run_sharded(bmh_ctx_t *const *ctxs, int n_ctx, int64_t n, Enqueue enqueue)
{
	if (!ctxs || n_ctx < 1 || n < 0) return BMH_E_ARG;
	for (int g = 0; g < n_ctx; ++g)
		if (!ctxs[g] || !ctxs[g]->have_params) return BMH_E_ARG;
	if (n == 0) return BMH_OK;
	for (int g = 0; g < n_ctx; ++g) {
		const int64_t lo = n * g / n_ctx, m = n * (g + 1) / n_ctx - lo;
		if (m == 0) continue;
		bmh_ctx *c = ctxs[g];
		int rc = hipSetDevice(c->device) == hipSuccess ? BMH_OK : BMH_E_HIP;
		if (!rc) rc = enqueue(c, g, lo, m);
		if (rc) { // copies into the caller's arrays may be in flight on the devices already served: drain them first
			drain_shards(ctxs, g);
			return rc;
		}
	}
	int first = BMH_OK;
	for (int g = 0; g < n_ctx; ++g) {
		if (n * (g + 1) / n_ctx == n * g / n_ctx) continue;
		BMH_HIP(ctxs[g], hipSetDevice(ctxs[g]->device));
		const int rc = fetch_err(ctxs[g]); // synchronises
		if (rc && !first) first = rc;
	}
	return first;
}
#define SH_TRY(expr)                                                                                                   \
	do {                                                                                                               \
		if ((expr) != hipSuccess) {                                                                                    \
			c->last_error = std::string(#expr) + ": " + hipGetErrorString(hipGetLastError());                          \
			return BMH_E_HIP;                                                                                          \
		}                                                                                                              \
	} while (0)

extern "C" {
int bmh_seedext_batch_sharded(bmh_ctx_t *const *ctxs, int n_ctx, const uint8_t *pool, size_t pool_bytes, const bmh_seed_task_t *tasks,
                              int64_t n, bmh_seed_result_t *results)
{
	if (n > 0 && (!pool || !tasks || !results)) return BMH_E_ARG;
	return run_sharded(ctxs, n_ctx, n, [&](bmh_ctx *c, int, int64_t lo, int64_t m) -> int {
		int qmax = 1, rc;
		if (m > 0x7fffffffLL) return BMH_E_ARG;
		if ((rc = validate_seeds(c, tasks + lo, m, pool_bytes, &qmax))) return rc;
		c->pool_resident = false;
		if ((rc = ensure(c, c->d_pool, pool_bytes + 16)) || (rc = ensure(c, c->d_tasks, (size_t)m * sizeof(bmh_seed_task_t) + 64)) ||
		    (rc = ensure(c, c->d_res, (size_t)m * sizeof(bmh_seed_result_t) + 64)))
			return rc;
		SH_TRY(hipMemcpyAsync(c->d_pool.p, pool, pool_bytes, hipMemcpyHostToDevice, c->stream));
		SH_TRY(hipMemcpyAsync(c->d_tasks.p, tasks + lo, (size_t)m * sizeof(bmh_seed_task_t), hipMemcpyHostToDevice, c->stream));
		if ((rc = launch_seedext(c, (const uint8_t *)c->d_pool.p, (const bmh_seed_task_t *)c->d_tasks.p, m, (bmh_seed_result_t *)c->d_res.p, qmax)))
			return rc;
		SH_TRY(hipMemcpyAsync(results + lo, c->d_res.p, (size_t)m * sizeof(bmh_seed_result_t), hipMemcpyDeviceToHost, c->stream));
		return BMH_OK;
	});
}

int bmh_sw_batch_sharded(bmh_ctx_t *const *ctxs, int n_ctx, const uint8_t *pool, size_t pool_bytes, const bmh_sw_task_t *tasks, int64_t n,
                         bmh_sw_result_t *results)
{
	if (n > 0 && (!pool || !tasks || !results)) return BMH_E_ARG;
	return run_sharded(ctxs, n_ctx, n, [&](bmh_ctx *c, int, int64_t lo, int64_t m) -> int {
		int qmax = 1, tmax = 1, qmin = 65535, rc;
		if ((rc = validate_sw(c, tasks + lo, m, pool_bytes, &qmax, &tmax, &qmin))) return rc;
		c->pool_resident = false;
		if ((rc = ensure(c, c->d_pool, pool_bytes + 16)) || (rc = ensure(c, c->d_tasks, (size_t)m * sizeof(bmh_sw_task_t))) ||
		    (rc = ensure(c, c->d_res, (size_t)m * sizeof(bmh_sw_result_t))))
			return rc;
		SH_TRY(hipMemcpyAsync(c->d_pool.p, pool, pool_bytes, hipMemcpyHostToDevice, c->stream));
		SH_TRY(hipMemcpyAsync(c->d_tasks.p, tasks + lo, (size_t)m * sizeof(bmh_sw_task_t), hipMemcpyHostToDevice, c->stream));
		if ((rc = launch_sw(c, (const uint8_t *)c->d_pool.p, (const bmh_sw_task_t *)c->d_tasks.p, m, (bmh_sw_result_t *)c->d_res.p, qmax, tmax, qmin)))
			return rc;
		SH_TRY(hipMemcpyAsync(results + lo, c->d_res.p, (size_t)m * sizeof(bmh_sw_result_t), hipMemcpyDeviceToHost, c->stream));
		return BMH_OK;
	});
}

int bmh_global_batch_sharded(bmh_ctx_t *const *ctxs, int n_ctx, const uint8_t *pool, size_t pool_bytes, const bmh_glb_task_t *tasks, int64_t n,
                             bmh_glb_result_t *results, uint32_t *cigar_pool, size_t cigar_words)
{
	if (n > 0 && (!pool || !tasks || !results)) return BMH_E_ARG;
	// a shard's CIGAR words come back into a buffer of its own first: the ranges of two shards may interleave in the caller's pool, and a
	// device only holds what its own tasks wrote
	std::vector<std::vector<uint32_t>> back((size_t)std::max(n_ctx, 0));
	std::vector<GlbShape> shape((size_t)std::max(n_ctx, 0));
	const int rc = run_sharded(ctxs, n_ctx, n, [&](bmh_ctx *c, int g, int64_t lo, int64_t m) -> int {
		int rc;
		GlbShape &gs = shape[(size_t)g];
		if (m > 0xffffffffLL) return BMH_E_ARG;
		if ((rc = validate_glb(c, tasks + lo, m, pool_bytes, cigar_pool != nullptr, cigar_words, &gs))) return rc;
		c->pool_resident = false;
		if ((rc = ensure(c, c->d_pool, pool_bytes + 16)) || (rc = ensure(c, c->d_tasks, (size_t)m * sizeof(bmh_glb_task_t))) ||
		    (rc = ensure(c, c->d_res, (size_t)m * sizeof(bmh_glb_result_t))) || (rc = ensure(c, c->d_cigar, (cigar_words + 4) * 4)))
			return rc;
		SH_TRY(hipMemcpyAsync(c->d_pool.p, pool, pool_bytes, hipMemcpyHostToDevice, c->stream));
		SH_TRY(hipMemcpyAsync(c->d_tasks.p, tasks + lo, (size_t)m * sizeof(bmh_glb_task_t), hipMemcpyHostToDevice, c->stream));
		if ((rc = launch_global(c, (const uint8_t *)c->d_pool.p, (const bmh_glb_task_t *)c->d_tasks.p, m, (bmh_glb_result_t *)c->d_res.p,
		                        (uint32_t *)c->d_cigar.p, nullptr, gs.qmax, gs.tmax, gs.wmax, gs.wraw)))
			return rc;
		SH_TRY(hipMemcpyAsync(results + lo, c->d_res.p, (size_t)m * sizeof(bmh_glb_result_t), hipMemcpyDeviceToHost, c->stream));
		if (gs.cig_hi > gs.cig_lo) {
			back[(size_t)g].resize(gs.cig_hi - gs.cig_lo);
			SH_TRY(hipMemcpyAsync(back[(size_t)g].data(), (const uint32_t *)c->d_cigar.p + gs.cig_lo, (gs.cig_hi - gs.cig_lo) * 4, hipMemcpyDeviceToHost, c->stream));
		}
		return BMH_OK;
	});
	if (rc && rc != BMH_E_CIGAR_CAP && rc != BMH_E_RANGE) return rc; // (a flagged task: the other tasks' results are still delivered)
	for (int g = 0; g < n_ctx && n > 0; ++g) {
		const int64_t lo = n * g / n_ctx, hi = n * (g + 1) / n_ctx;
		if (back[(size_t)g].empty()) continue;
		for (int64_t k = lo; k < hi; ++k) {
			const bmh_glb_task_t &x = tasks[k];
			const size_t nw = std::min<size_t>((size_t)std::max(results[k].n_cigar, 0), x.cigar_cap);
			if (nw) memcpy(cigar_pool + x.cigar_off, back[(size_t)g].data() + ((size_t)x.cigar_off - shape[(size_t)g].cig_lo), nw * 4);
		}
	}
	return rc;
}
#undef SH_TRY

int bmh_ctx_reserve_staging(bmh_ctx_t *ctx, size_t upload_bytes, size_t download_bytes)
{
	if (!ctx) return BMH_E_ARG;
	int rc;
	BMH_HIP(ctx, hipSetDevice(ctx->device));
	if ((rc = ensure_host(ctx, ctx->h_up, upload_bytes)) || (rc = ensure_host(ctx, ctx->h_down, download_bytes))) return rc;
	return BMH_OK;
}

int bmh_ctx_reserve_device(bmh_ctx_t *ctx, size_t pool_bytes, int64_t max_tasks, size_t cigar_words)
{
	if (!ctx || max_tasks < 0) return BMH_E_ARG;
	int rc;
	const size_t N = (size_t)max_tasks;
	BMH_HIP(ctx, hipSetDevice(ctx->device));
	if ((rc = ensure(ctx, ctx->d_pool, pool_bytes + 16)) || (rc = ensure(ctx, ctx->d_tasks, N * 40 + 64)) || (rc = ensure(ctx, ctx->d_res, N * 32 + 64)) ||
	    (rc = ensure(ctx, ctx->d_cigar, (cigar_words + 4) * 4)) ||
	    (rc = ensure(ctx, ctx->d_bins, (16 + (size_t)kSortBins * kSortKeysHost + (N + 1) / 2 + 1 + (size_t)kSortBins * N) * 4)))
		return rc;
	return BMH_OK;
}

int bmh_ctx_reserve_kernels(bmh_ctx_t *ctx, int seed_reads, int seed_read_len, int64_t global_tasks, int global_rows)
{
	if (!ctx || seed_reads < 0 || seed_read_len < 0 || global_tasks < 0 || global_rows < 0) return BMH_E_ARG;
	int rc;
	BMH_HIP(ctx, hipSetDevice(ctx->device));
	if (seed_reads > 0) { // the SMEM kernels' interval stacks (fmindex.hip): three per lane, read length + 2 entries of 32 bytes each
		const size_t grid = std::min<size_t>(((size_t)seed_reads + 63) / 64, 1024 * 4);
		if ((rc = ensure(ctx, ctx->d_sw, grid * 3 * ((size_t)seed_read_len + 2) * 64 * 32))) return rc;
	}
	if (global_tasks > 0) { // the lane kernels' direction slab (global_lane.hip): resident waves x rows x 8 blocks x 256 bytes
		const size_t grid = std::min<size_t>(((size_t)global_tasks + 63) / 64, (size_t)std::max(ctx->ncu, 1) * 4 * 2);
		if ((rc = ensure(ctx, ctx->d_zslab, grid * (size_t)std::min(global_rows, 512) * 8 * 256))) return rc;
	}
	return BMH_OK;
}

int bmh_driver_stats(const bmh_ctx_t *ctx, bmh_driver_stats_t *st)
{
	if (!ctx || !st) return BMH_E_ARG;
	*st = ctx->dstats;
	return BMH_OK;
}

} // extern "C"
