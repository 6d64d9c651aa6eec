// bmh_ctx.h -- host-side context shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/bwamem_hip.h"
#include "bmh_device.h"

constexpr int kExtBinsMax = 6;

struct DevBuf { // grow-only device allocation
	void *p = nullptr;
	size_t cap = 0;
};

struct bmh_ctx {
	int device = 0;
	hipStream_t own_stream = nullptr, stream = nullptr;
	bool have_params = false;
	bmh_params_t params{};
	bmh::DevParams dev{};
	int qcap = 512; // query-length capacity used to size the LDS kernel's window state for *_device calls
	// device workspaces of the host-buffer entry points
	DevBuf d_pool, d_tasks, d_res, d_order, d_cigar, d_scratch;
	void *bwt_bind = nullptr; // FM-index binding (fmindex.hip), or null
	double smem_calls_per_base = 0.06, smem_intv_per_base = 0.35, smem_pos_per_base = 0.03; // high-water marks of bmh_smem_batch's / bmh_seed_batch's output density
	const uint8_t *h_pac = nullptr; // host identity of the shared device copy of the 2-bit reference (bmh_ctx_set_pac)
	DevBuf d_swrm;  // per-wave row maxima of the register Smith-Waterman kernels
	DevBuf d_sw;    // row / row-maximum slabs of the local Smith-Waterman kernels
	DevBuf d_zslab; // direction words of the lane-per-task global kernels, one slab per resident wave
	int glb_mode = 0; // 0 lane-per-task global kernels, 1 one wave per task only (env BMH_GLB_MODE=wave)
	int sw_mode = 0;  // 0 register kernels where they fit, 1 slab kernel only (env BMH_SW_MODE=generic)
	int glb_fast = 1;   // unmasked body for blocks inside every lane's band (env BMH_GL_FAST=0 turns it off)
	int sw_wave = 1;  // batches of up to 32 k tasks: one wave per task (sw_wave.hip); env BMH_SW_WAVE=0 turns it off
	DevBuf d_bins; // per-launch bin lists of the extension dispatcher: 4 counters + 4 x n task indices
	int grid_mult = 1;    // env BMH_GRID_MULT: persistent grid = resident waves x this (tuning knob)
	int ext_sched = -1;   // env BMH_EXT_SCHED: launch order / streams of the extension bins, -1 = by query length (see launch_extend)
	bool ext_mode_forced = false; // BMH_EXT_MODE was given
	int small_batch = 49152;      // batches of at most this many extension tasks go to the one-task-per-wave kernels (env BMH_EXT_SMALL, 0 = never)
	int force_kernel = 0; // kernels for qlen<=128: 0 lane-per-task, 1 LDS kernel, 2 one task/wave, 3 four tasks/wave (env BMH_EXT_MODE=lds|reg|grp)
	bool pool_resident = false; // d_pool holds a pool uploaded by bmh_upload_pool()
	size_t pool_bytes = 0;
	// pinned staging of the entry points that move bulk data per call from many host threads at once (the runtime's own
	// path for pageable memory was the slowest part of a seeding batch: 5-36 ms for 11 MB with eight threads in flight)
	DevBuf h_up, h_down;
	int *d_err = nullptr; // device error flag (BMH_E_* or 0)
	int *h_err = nullptr; // pinned mirror
	// kernel timing
	bool timing = false;
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	bool ev_valid = false;
	hipEvent_t ev_bin[kExtBinsMax + 1] = {};     // start of each extension bin's kernels
	hipEvent_t ev_bin_end[kExtBinsMax + 1] = {}; // ... and their end (bins run on two streams, see launch_extend)
	hipStream_t aux_stream = nullptr;            // the few long flanks (bins 3-5) run beside the lane kernels
	hipEvent_t ev_fork = nullptr, ev_join = nullptr;
	hipStream_t aux2_stream = nullptr; // ... and the two short-query bins beside the 128-column one
	hipEvent_t ev_join2 = nullptr;
	hipEvent_t ev_wait = nullptr; // hipEventBlockingSync: what stream_wait() sleeps on in blocking mode
	bool ev_bin_valid = false;
	int ext_split96 = 1; // the 65-128 column bin sends its tasks of up to 96 columns to extend_lane_kernel<96> (3 waves/SIMD); BMH_EXT_SPLIT96=0:
	                     // one kernel for the bin.  Measured: a 1 M-read batch 3.55 against 3.83 ms; the 20 M-read step's extension stage level at
	                     // 4 M-read chunks (83.4 ms: the bin's 4-8 k waves per launch lose to wave quantisation what the third resident wave wins),
	                     // 76.1 against 80.3 ms at 10 M-read chunks
	double ext_bin_ms_sum[kExtBinsMax + 1] = {}; // timing mode: per-bin kernel time summed over dispatcher launches (bmh_extend_bin_ms_sum)
	long long ext_bin_launches = 0;
	hipEvent_t ev_gbin[4] = {}; // boundaries of the three kernels of a global-alignment launch (64-slot, 128-slot, wave)
	bool ev_gbin_valid = false;
	hipEvent_t ev_sround[5] = {}; // boundaries of the four rounds of a fused per-seed launch
	bool ev_sround_valid = false;
	std::string last_error;
	bmh_driver_stats_t dstats{};
	int ncu = 256; // compute units of the device (persistent grids are sized from it)
	bool ext_persist = false; // extension lane kernels as one strided launch with a capped grid (env BMH_EXT_PERSIST=1)
	int ext_grid_mult = 2; // extension lane kernels: grid cap = resident waves x this (env BMH_EXT_GRID_MULT)
	// Bin sizes of the extension dispatcher are only known on the device.  After every launch they are copied to pinned
	// memory WITHOUT waiting; the next launch of the same kind (plain API call, or stage k of the fused per-seed
	// pipeline) reads whatever has arrived and sizes its grids / picks the kernel of bin 3 from it.  A stale or missing
	// hint costs speed, never correctness: every kernel strides over its bin whatever the grid.
	static constexpr int kHintKinds = 6;
	struct BinHint {
		uint32_t *h = nullptr;   // pinned, 16 words: the bins' counts of the launch that wrote it
		hipEvent_t ev = nullptr; // recorded behind the copy
		bool pending = false, valid = false;
		uint32_t cnt[8] = {};
	} hint[kHintKinds];
	DevBuf d_seedws; // workspace of the fused per-seed extension (seedext.hip)
	DevBuf d_region; // region records, their results, CIGAR and MD slots (bmh_region_cigar_batch)
	bmh_seedext_stats_t sstats{};
	int64_t seed_pending_n = -1; // tasks of the bmh_seedext_submit() in flight, -1 = none
};

namespace bmh {

// the caller's gate around a device section (bmh_set_device_gate), held for the lifetime of the guard
extern bmh_gate_fn g_gate_enter, g_gate_leave;
struct GateGuard {
	bmh_gate_fn leave;
	GateGuard() : leave(g_gate_leave)
	{
		if (g_gate_enter) g_gate_enter();
	}
	~GateGuard()
	{
		if (leave) leave();
	}
	GateGuard(const GateGuard &) = delete;
	GateGuard &operator=(const GateGuard &) = delete;
};

int set_hip_error(bmh_ctx *ctx, hipError_t e, const char *what);
// every host wait for a stream goes through here (bmh_set_wait_mode)
extern int g_wait_blocking;
inline hipError_t stream_wait(bmh_ctx *ctx, hipStream_t s)
{
	if (!g_wait_blocking || !ctx->ev_wait) return hipStreamSynchronize(s);
	const hipError_t e = hipEventRecord(ctx->ev_wait, s);
	return e != hipSuccess ? e : hipEventSynchronize(ctx->ev_wait);
}
int ensure(bmh_ctx *ctx, DevBuf &b, size_t bytes);
int ensure_host(bmh_ctx *ctx, DevBuf &b, size_t bytes); // same, pinned host memory

#define BMH_HIP(ctx, call)                                                   \
	do {                                                                     \
		hipError_t e_ = (call);                                              \
		if (e_ != hipSuccess) return bmh::set_hip_error((ctx), e_, #call);   \
	} while (0)

// kernel launchers (defined next to the kernels)
// grid cap of the extension kernels: enough blocks to keep every wave slot refilled (256 CUs x 32 waves x 4),
// each block walks its bin with a grid stride
constexpr long long kPersistentGrid = 256LL * 32 * 4;

// dispatcher: classifies the tasks by query length on the device and runs each bin on its kernel
// d_n (nullable): device-side count <= n of the entries of d_order (or of d_tasks) that are tasks; kind: which
// BinHint slot the launch reads and refreshes
int launch_extend(bmh_ctx *ctx, const uint8_t *d_pool, const bmh_ext_task_t *d_tasks, int64_t n,
                  bmh_ext_result_t *d_res, const uint32_t *d_order, int qmax, const uint32_t *d_n = nullptr, int kind = 0);
int launch_seedext(bmh_ctx *ctx, const uint8_t *d_pool, const bmh_seed_task_t *d_tasks, int64_t n,
                   bmh_seed_result_t *d_res, int qmax);
const uint32_t *seedext_counters(const bmh_ctx *ctx); // the four list lengths of the last launch_seedext, on the device
inline long long ext_resident_waves(const bmh_ctx *ctx, int waves_per_simd) { return (long long)ctx->ncu * 4 * waves_per_simd; }
int launch_extend_lds(bmh_ctx *ctx, const uint8_t *d_pool, const bmh_ext_task_t *d_tasks, int64_t n,
                      bmh_ext_result_t *d_res, const uint32_t *d_order, const uint32_t *d_count, int qmax, long long grid_cap = 0);
constexpr int kSortKeysHost = 2048; // == kSortKeys in extend_dispatch.hip
int sort_tasks_begin(bmh_ctx *ctx, int64_t n, uint32_t **counts, uint32_t **lists);
int sort_tasks_finish(bmh_ctx *ctx, int64_t n, const uint32_t *d_order, unsigned blocks, const uint32_t *d_n = nullptr);
int launch_global_lane(bmh_ctx *ctx, int c, const uint8_t *d_pool, const bmh_glb_task_t *d_tasks, int64_t n,
                       bmh_glb_result_t *d_res, uint32_t *d_cigar, const uint32_t *d_order, const uint32_t *d_count,
                       int rows_cap);
constexpr int kExtBins = 6;        // length bins of the extension dispatcher
constexpr int kSortBins = 8;       // bins the shared counting sort can tell apart (extension 6, global 6, Smith-Waterman 8)
constexpr int kGrpTcapHost = 1024; // == kGrpTcap in extend_grp.hip
int launch_extend_grp(bmh_ctx *ctx, int nv, const uint8_t *d_pool, const bmh_ext_task_t *d_tasks, int64_t n,
                      bmh_ext_result_t *d_res, const uint32_t *d_order, const uint32_t *d_count);
int launch_extend_lanex(bmh_ctx *ctx, int lpt, const uint8_t *d_pool, const bmh_ext_task_t *d_tasks, int64_t n,
                        bmh_ext_result_t *d_res, const uint32_t *d_order, const uint32_t *d_count, int min_count);
int launch_sw(bmh_ctx *ctx, const uint8_t *d_pool, const bmh_sw_task_t *d_tasks, int64_t n, bmh_sw_result_t *d_res,
              int qcap, int tcap, int qmin);
int launch_sw_lane(bmh_ctx *ctx, int b, bool corr, bool word, const uint8_t *d_pool, const bmh_sw_task_t *d_tasks, int64_t n,
                   bmh_sw_result_t *d_res, const uint32_t *d_order, const uint32_t *d_count, uint16_t *d_rm, int rows_cap,
                   int grid, int pass2, uint32_t *d_next);
bool sw_wave_fits(int64_t n, int qcap, int tcap);
int launch_sw_wave(bmh_ctx *ctx, const uint8_t *d_pool, const bmh_sw_task_t *d_tasks, int64_t n, bmh_sw_result_t *d_res, int max_cols,
                   int tcap, const uint32_t *d_order = nullptr, const uint32_t *d_count = nullptr);
int launch_sw_generic(bmh_ctx *ctx, const uint8_t *d_pool, const bmh_sw_task_t *d_tasks, int64_t n,
                      bmh_sw_result_t *d_res, const uint32_t *d_order, const uint32_t *d_count, int qcap, int tcap, int wave_cols = 0);
int launch_extend_lane(bmh_ctx *ctx, int c, const uint8_t *d_pool, const bmh_ext_task_t *d_tasks, int64_t n,
                       bmh_ext_result_t *d_res, const uint32_t *d_order, const uint32_t *d_count, bool exact, const uint32_t *d_skip = nullptr);
int launch_extend_reg(bmh_ctx *ctx, int ns, const uint8_t *d_pool, const bmh_ext_task_t *d_tasks, int64_t n,
                      bmh_ext_result_t *d_res, const uint32_t *d_order, const uint32_t *d_count, int max_count = 0, long long grid_cap = 0);
// wmax: the widest band in stored columns (min(w, qlen)), sizes the wave kernel's direction matrix; wgate: the largest w
// as the tasks carry it -- the device bins by that, so it decides which lane kernels are launched
int launch_global(bmh_ctx *ctx, const uint8_t *d_pool, const bmh_glb_task_t *d_tasks, int64_t n,
                  bmh_glb_result_t *d_res, uint32_t *d_cigar, const uint32_t *d_order, int qmax, int tmax,
                  int wmax, int wgate);

int launch_region_orient(bmh_ctx *ctx, uint8_t *d_pool, size_t rpool_off, const bmh_region_req_t *d_reqs, int64_t n);
int launch_region_finish(bmh_ctx *ctx, const uint8_t *d_pool, const bmh_region_req_t *d_reqs, int64_t n, const bmh_glb_task_t *d_tasks,
                         const bmh_glb_result_t *d_gres, const uint32_t *d_tcig, bmh_region_res_t *d_out, uint32_t *d_cig_out, int cig_cap,
                         char *d_md_out, int md_cap);

} // namespace bmh
