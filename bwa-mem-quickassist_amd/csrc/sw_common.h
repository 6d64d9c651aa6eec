// sw_common.h -- pieces shared by the local Smith-Waterman kernels (ksw_align2, reference ksw.c:62-364).
#pragma once
#include "bmh_device.h"

namespace bmh {

struct SwCore { // what one pass of ksw_u8 / ksw_i16 returns (kswr_t without tb/qb)
	int score, te, qe, score2, te2;
};

// 16-bit score range of the kernels
__device__ __forceinline__ bool sw_task_out_of_range(const DevParams &P, int qlen, uint32_t xtra)
{
	(void)xtra;
	return (long long)qlen * P.max_mat >= kScoreLimit || qlen < 1;
}

// Second-best score, reference ksw.c:181-189 (the b array: one entry per run of consecutive rows whose maximum
// reaches minsc, an entry being re-anchored at the row of a new run maximum) and ksw.c:209-220 (best entry outside
// [te-d, te+d], d = ceil(score/max(mat))), replayed from the per-row maxima rm[i*stride] after the pass.
__device__ __forceinline__ void sw_second_best(const uint16_t *rm, int stride, int nrows, int minsc, int score, int te,
                                               int qmax, int *score2, int *te2)
{
	const int d = (score + qmax - 1) / qmax, low = te - d, high = te + d;
	bool have = false;
	int lsc = 0, li = 0, s2 = -1, t2 = -1;
	for (int i = 0; i < nrows; ++i) {
		const int im = rm[(size_t)i * stride];
		if (im < minsc) continue;
		if (!have || li + 1 != i) {
			if (have && (li < low || li > high) && lsc > s2) s2 = lsc, t2 = li;
			have = true, lsc = im, li = i;
		} else if (lsc < im) lsc = im, li = i;
	}
	if (have && (li < low || li > high) && lsc > s2) s2 = lsc, t2 = li;
	*score2 = s2, *te2 = t2;
}

// Tasks sw_wave_kernel takes when a batch goes its way (launch_sw): those whose striped byte or word arithmetic cannot
// saturate -- the register kernels' condition (sw_dispatch.hip) -- and whose padded query fits max_cols columns.
__device__ __forceinline__ bool sw_wave_takes(const DevParams &P, int qlen, uint32_t xtra, int max_cols)
{
	if (qlen < 1) return false;
	const bool byte_mode = xtra & BMH_SW_XBYTE;
	const int segs = byte_mode ? 16 : 8;
	if ((qlen + segs - 1) / segs * segs > max_cols) return false;
	return byte_mode ? qlen * P.max_mat + P.sw_shift < 255 : qlen * P.max_mat + P.sw_shift < 512;
}

} // namespace bmh
