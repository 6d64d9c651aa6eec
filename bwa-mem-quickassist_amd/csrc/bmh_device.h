// bmh_device.h -- device-side helpers shared by the gfx950 kernels.
// wave64 only (CDNA4): every cross-lane primitive here assumes 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bwamem_hip.h"

namespace bmh {

// Scoring constants, passed by value in the kernarg segment (uniform -> SGPRs).
struct DevParams {
	int o_del, e_del, o_ins, e_ins, zdrop;
	int max_mat;      // max entry of mat[] (ksw.c:398-400), precomputed on the host
	int bias;         // -min(mat[]) floored at 0: profile bytes are stored as score+bias (unsigned)
	int sw_shift;     // ksw_qinit's byte-mode bias (uint8_t)(256 - min(mat)), reference ksw.c:78-85
	uint32_t matw[7]; // mat[25] as bytes, little endian, padded to 28
	const uint8_t *pac; // 2-bit reference resident in HBM (bmh_ctx_set_pac), or null
	long long l_pac;    // its length in bases; the doubled coordinate [l_pac, 2*l_pac) is the reverse strand
};

constexpr int kScoreLimit = 32000; // h0 + qlen*max_mat must stay below this (16-bit lanes)
constexpr int kNegInf16 = -16384;  // scan identity for 16-bit-ranged values held in int32

// ---- DPP controls (GFX9 encoding) ----
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118;
constexpr int DPP_WAVE_SHR1 = 0x138, DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;

template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ int dpp(int old, int src)
{
	return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xf, false);
}

// inclusive prefix max over the 64 lanes (6 VALU+DPP steps, no LDS).
// The DPP `old` operand is INT32_MIN, the identity of signed max, which lets
// hipcc's DPP combiner fold each step into a single v_max_i32_dpp.
__device__ __forceinline__ int wave_scan_max(int v)
{
	constexpr int I = INT32_MIN;
	v = max(dpp<DPP_ROW_SHR1>(I, v), v);
	v = max(dpp<DPP_ROW_SHR2>(I, v), v);
	v = max(dpp<DPP_ROW_SHR4>(I, v), v);
	v = max(dpp<DPP_ROW_SHR8>(I, v), v);
	v = max(dpp<DPP_ROW_BCAST15, 0xa>(I, v), v);
	v = max(dpp<DPP_ROW_BCAST31, 0xc>(I, v), v);
	return v;
}

// max over the wave, returned as a wave-uniform value
__device__ __forceinline__ int wave_reduce_max(int v)
{
	return __builtin_amdgcn_readlane(wave_scan_max(v), 63);
}

// value of lane-1 (lane 0 receives `lane0`)
__device__ __forceinline__ int wave_shr1(int v, int lane0)
{
	return dpp<DPP_WAVE_SHR1>(lane0, v);
}

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ int mat_at(const DevParams &P, int idx)
{
	return (int)(int8_t)(P.matw[idx >> 2] >> ((idx & 3) * 8));
}

// sequence base k of a task: forward pool[off+k], reversed pool[off-k]
__device__ __forceinline__ int seq_base(const uint8_t *pool, uint64_t off, int k, bool rev)
{
	return rev ? pool[off - (uint64_t)k] : pool[off + (uint64_t)k];
}

// target base k of a task.  With BMH_F_TPAC the offset is a position on bwa's doubled coordinate and the base is
// decoded from the 2-bit pac on the fly -- bns_get_seq (reference bntseq.c:355-376) done by the consumer:
// forward strand pac[p], reverse strand 3 - pac[2*l_pac-1-p].
__device__ __forceinline__ int tgt_base(const uint8_t *pool, const DevParams &P, uint64_t off, int k, bool rev, bool tpac)
{
	if (!tpac) return seq_base(pool, off, k, rev);
	const long long p = rev ? (long long)off - k : (long long)off + k;
	const bool rs = p >= P.l_pac;
	const long long f = rs ? (P.l_pac << 1) - 1 - p : p;
	const int b = P.pac[f >> 2] >> ((~f & 3) << 1) & 3;
	return rs ? 3 - b : b;
}

// ksw.c:401-405 without floating point: trunc((x)/e + 1) floored at 1 equals x/e+1 for x>=0, else 1 (e>=1)
__device__ __forceinline__ int band_cap(int qlen, int mx, int end_bonus, int o, int e)
{
	int x = qlen * mx + end_bonus - o;
	return x >= 0 ? x / e + 1 : 1;
}

} // namespace bmh
