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

// ---- the cell in 16-bit arithmetic.  What an instruction costs on gfx950 depends on its class (profiles/r03_valu_issue_classes.md,
// tools/microbench/valu_mix.hip): the 16-bit VOP2 forms (v_add/sub/max_i16 ...), v_lshrrev_b32 and v_bitop3_b32 issue every 2 cycles
// per SIMD once 4 waves are resident, while v_max_i32, v_cmp, v_cndmask, every SDWA/DPP form and every other VOP3 issue every 4.
// hipcc picks the 32-bit forms for this code, so the cell is spelled out.  Operands are 16-bit two's-complement values in the LOW
// half of a VGPR; the 16-bit instructions ignore the high half of their inputs and zero it in their result.
#define BMH_OP16(name, text)                                                                                            \
	__device__ __forceinline__ int name(int a, int b)                                                                   \
	{                                                                                                                   \
		int d;                                                                                                          \
		asm(text " %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));                                                             \
		return d;                                                                                                       \
	}
BMH_OP16(add16, "v_add_u16")   // a + b
BMH_OP16(sub16, "v_sub_u16")   // a - b: bit 15 = [a < b] while |a - b| < 32768
BMH_OP16(max16, "v_max_i16")   // signed
#undef BMH_OP16
// (hipcc has v_max_u16 / v_sub_u16 clamp for __builtin_elementwise_max / _sub_sat on uint16_t, but then also fuses pairs of them into
// v_pk_* and SDWA forms of the 4-cycle class and the kernels measured slower: extension stage 82.5 -> 87.4 ms.  Hence asm.)
__device__ __forceinline__ int maxu16(int a, int b)
{
	int d;
	asm("v_max_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
	return d;
}
// max(a - k, 0) on unsigned 16-bit values: one saturating subtraction (k wave-uniform)
__device__ __forceinline__ int subc16(int a, int k)
{
	int d;
	asm("v_sub_u16_e64 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "s"(k));
	return d;
}
__device__ __forceinline__ int nonzero16(int a) // min(a, 1) on an unsigned 16-bit value
{
	int d;
	asm("v_min_u16 %0, 1, %1" : "=v"(d) : "v"(a));
	return d;
}
__device__ __forceinline__ int subk16(int a, int k) // a - k, k wave-uniform (SGPR or inline constant as src0 of v_subrev)
{
	int d;
	asm("v_subrev_u16 %0, %1, %2" : "=v"(d) : "s"(k), "v"(a));
	return d;
}
// H(i-1,j-1) (low half of r) + the score byte B of sc4, sign-extended: low half of the result = M(i,j)
template <int B> __device__ __forceinline__ int add_score(int sc4, int r)
{
	int d;
	if constexpr (B == 0) asm("v_add_u32_sdwa %0, sext(%1), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:WORD_0" : "=v"(d) : "v"(sc4), "v"(r));
	if constexpr (B == 1) asm("v_add_u32_sdwa %0, sext(%1), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:WORD_0" : "=v"(d) : "v"(sc4), "v"(r));
	if constexpr (B == 2) asm("v_add_u32_sdwa %0, sext(%1), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:WORD_0" : "=v"(d) : "v"(sc4), "v"(r));
	if constexpr (B == 3) asm("v_add_u32_sdwa %0, sext(%1), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:WORD_0" : "=v"(d) : "v"(sc4), "v"(r));
	return d;
}
// high half of r = max(a, b) (signed 16-bit), low half kept: E(i+1,j) goes in beside the H(i,j) already in r
__device__ __forceinline__ void max16_into_hi(int &r, int a, int b)
{
	asm("v_max_i16_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0" : "+v"(r) : "v"(a), "v"(b));
}

// high half of r = a - b (16-bit), low half kept: two differences whose sign bits are wanted share a register
__device__ __forceinline__ void sub16_into_hi(int &r, int a, int b)
{
	asm("v_sub_u16_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0" : "+v"(r) : "v"(a), "v"(b));
}

// ksw.c:401-405 without floating point: trunc((x)/e + 1) floored at 1 equals x/e+1 for x>=0, else 1 (e>=1)
__device__ __forceinline__ int band_cap(int qlen, int mx, int end_bonus, int o, int e)
{
	int x = qlen * mx + end_bonus - o;
	return x >= 0 ? x / e + 1 : 1;
}

} // namespace bmh
