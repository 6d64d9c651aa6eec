// region_cigar.hip -- the byte work of bwa_gen_cigar2 (reference bwa-0.7.8/bwa.c:89-172) around ksw_global2, on the device:
// bmh_region_cigar_batch()'s two kernels.  The global alignments between them are launch_global()'s (global_lane.hip).
//
//   region_orient_kernel   bns_get_seq over the resident 2-bit reference (bntseq.c:355-376) + the reversal of query and window for hits
//                          on the reverse strand (bwa.c:100-107): one wave per region writes the oriented copies the global kernels read.
//   region_finish_kernel   one lane per region: the no-gap score (bwa.c:108-114) or the replay of mem_reg2aln's band loop over the tries'
//                          results (bwamem.c:1194-1201), then NM and MD from the final CIGAR (bwa.c:134-164).
// Both are a few bytes per base once per region -- microseconds beside the alignments; what they buy is on the host, which no longer
// touches a sequence byte in phase 2 (oriented copies + NM/MD were 8 of a slice's 45 ms of CPU, profiles/r02_pipeline_stage_clocks_final.txt).
#include <hip/hip_runtime.h>

#include "bmh_ctx.h"
#include "bmh_device.h"

namespace bmh {

__global__ __launch_bounds__(256) void region_orient_kernel(uint8_t *__restrict__ pool, size_t rpool_off,
                                                            const bmh_region_req_t *__restrict__ reqs, long long n, DevParams P)
{
	const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
	const int lane = threadIdx.x & 63;
	if (r >= n) return;
	const bmh_region_req_t q = reqs[r];
	const bool rev = q.rb >= P.l_pac;
	const uint8_t *src = pool + rpool_off + q.q_src;
	uint8_t *dq = pool + q.o_off, *dt = dq + q.ql;
	for (int i = lane; i < q.ql; i += 64) dq[i] = rev ? src[q.ql - 1 - i] : src[i];
	// reverse strand: the window [rb, rb+tl) of the doubled coordinate, read backwards
	const uint64_t t0 = rev ? (uint64_t)(q.rb + q.tl - 1) : (uint64_t)q.rb;
	for (int i = lane; i < q.tl; i += 64) dt[i] = (uint8_t)tgt_base(nullptr, P, t0, i, rev, true);
}

struct MdOut { // kputw / kputc into the region's slot; past the cap only the length keeps counting
	char *s;
	int l, cap;
	__device__ void c(char ch)
	{
		if (l < cap) s[l] = ch;
		++l;
	}
	__device__ void num(int v) // kputw (kstring.h:62-77), v >= 0
	{
		char buf[12];
		int k = 0;
		if (v == 0) buf[k++] = '0';
		for (; v > 0; v /= 10) buf[k++] = (char)('0' + v % 10);
		while (k) c(buf[--k]);
	}
};

__global__ __launch_bounds__(64) void region_finish_kernel(const uint8_t *__restrict__ pool, const bmh_region_req_t *__restrict__ reqs,
                                                           long long n, const bmh_glb_task_t *__restrict__ tasks,
                                                           const bmh_glb_result_t *__restrict__ gres, const uint32_t *__restrict__ tcig,
                                                           bmh_region_res_t *__restrict__ out, uint32_t *__restrict__ cig_out, int cig_cap,
                                                           char *__restrict__ md_out, int md_cap, DevParams P, int a)
{
	const long long k = (long long)blockIdx.x * 64 + threadIdx.x;
	if (k >= n) return;
	const bmh_region_req_t rq = reqs[k];
	const bool rev = rq.rb >= P.l_pac;
	const uint8_t *q = pool + rq.o_off, *t = q + rq.ql;
	const bool single = rq.truesc == INT32_MIN;
	bmh_region_res_t r;
	r.flags = 0, r.NM = -1, r.md_len = 0;
	uint32_t *cg = cig_out + (size_t)k * cig_cap;
	if (rq.task[0] < 0) { // no gap, no DP: one match run (bwa.c:108-114); the same score at every band, so a second try ends the loop
		int sc = 0;
		for (int i = 0; i < rq.ql; ++i) sc += mat_at(P, t[i] * 5 + q[i]);
		r.score = sc, r.n_cigar = 1, r.tries = (!single && sc < rq.truesc - a) ? 2 : 1;
		cg[0] = (uint32_t)rq.ql << 4;
	} else { // bwamem.c:1194-1201 over the precomputed tries
		int last = -(1 << 30), tries = 0, fin = rq.task[0];
		for (int t_ = 0;; ++t_) {
			fin = rq.task[t_];
			++tries;
			r.score = gres[fin].score;
			if (r.score == last) break; // bwamem.c:1198
			last = r.score;
			if (single || !(tries < 3 && r.score < rq.truesc - a)) break; // bwamem.c:1201
		}
		r.tries = tries, r.n_cigar = gres[fin].n_cigar;
		const bmh_glb_task_t ft = tasks[fin];
		if ((uint32_t)r.n_cigar > ft.cigar_cap || r.n_cigar > cig_cap) {
			r.flags = BMH_REGION_CIGAR_CUT;
			out[k] = r;
			return;
		}
		for (int i = 0; i < r.n_cigar; ++i) cg[i] = tcig[ft.cigar_off + i];
	}
	// NM and MD, bwa.c:134-164: walk the CIGAR over the oriented copies; `run` = matches since the last MD token
	MdOut md{md_out + (size_t)k * md_cap, 0, md_cap};
	const char *letter = rev ? "TGCAN" : "ACGTN";
	int qpos = 0, tpos = 0, run = 0, mismatches = 0, gap_bases = 0;
	for (int c = 0; c < r.n_cigar; ++c) {
		const int op = (int)(cg[c] & 0xf), len = (int)(cg[c] >> 4);
		switch (op) {
		case 0: // M: a mismatch closes the run and names the reference base
			for (int i = 0; i < len; ++i) {
				if (q[qpos + i] == t[tpos + i]) {
					++run;
					continue;
				}
				md.num(run), md.c(letter[t[tpos + i]]);
				++mismatches, run = 0;
			}
			qpos += len, tpos += len;
			break;
		case 2: // D: "^" + the deleted reference bases, unless it is the first or the last operation
			if (c > 0 && c < r.n_cigar - 1) {
				md.num(run), md.c('^');
				for (int i = 0; i < len; ++i) md.c(letter[t[tpos + i]]);
				run = 0, gap_bases += len;
			}
			tpos += len;
			break;
		case 1: // I
			qpos += len, gap_bases += len;
			break;
		default: break;
		}
	}
	md.num(run);
	r.NM = mismatches + gap_bases, r.md_len = md.l;
	if (md.l > md_cap) r.flags |= BMH_REGION_MD_CUT;
	out[k] = r;
}

int launch_region_orient(bmh_ctx *ctx, uint8_t *d_pool, size_t rpool_off, const bmh_region_req_t *d_reqs, int64_t n)
{
	if (n <= 0) return BMH_OK;
	hipLaunchKernelGGL(region_orient_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, ctx->stream, d_pool, rpool_off, d_reqs, (long long)n,
	                   ctx->dev);
	BMH_HIP(ctx, hipGetLastError());
	return BMH_OK;
}

int launch_region_finish(bmh_ctx *ctx, const uint8_t *d_pool, const bmh_region_req_t *d_reqs, int64_t n, const bmh_glb_task_t *d_tasks,
                         const bmh_glb_result_t *d_gres, const uint32_t *d_tcig, bmh_region_res_t *d_out, uint32_t *d_cig_out, int cig_cap,
                         char *d_md_out, int md_cap)
{
	if (n <= 0) return BMH_OK;
	hipLaunchKernelGGL(region_finish_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, d_pool, d_reqs, (long long)n, d_tasks,
	                   d_gres, d_tcig, d_out, d_cig_out, cig_cap, d_md_out, md_cap, ctx->dev, ctx->params.a);
	BMH_HIP(ctx, hipGetLastError());
	return BMH_OK;
}

} // namespace bmh
