// kernels.hpp -- hand-written gfx950 (CDNA4, wave64) kernels of the vcf2multialign splice path.
//
// Integer copy/index work, HBM-bound: no MFMA anywhere.  The rules that matter here are
// coalesced 16-B/lane global accesses, LDS staging of byte-granular patches, keeping the
// shared inputs (REF row, edge tables) L2/Infinity-Cache resident across the rows that
// reuse them, and launching far more workgroups than the 256 CUs.
//
// Reference semantics (paths relative to the reference tree):
//   transpose_bits_kernel           libvcf2multialign/transpose_matrix.cc:41-109
//   expand_reference_row_kernel     libvcf2multialign/sequence_writer.cc:22-85 with copy = PLOIDY_MAX
//   resolve_effective_edges_kernel  the order-dependent part of sequence_writer.cc:51-67
//   splice_aligned_kernel           the byte-producing part of sequence_writer.cc:57-83
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace v2m {

typedef uint32_t u32;
typedef uint64_t u64;

constexpr int kWave = 64;
typedef u32 vec4u __attribute__((ext_vector_type(4)));   // native 16-B vector (nontemporal builtins need it)
typedef vec4u vec4u_unaligned8 __attribute__((aligned(8)));   // a 16-B access at an 8-byte aligned address (legal on gfx950: unaligned access mode)

// ---------------------------------------------------------------------------------------------
// Geometry of the aligned splice.
// ---------------------------------------------------------------------------------------------
constexpr int kTileBytes = 16384;                    // aligned positions per tile
constexpr int kSpliceThreads = 256;
constexpr int kTileChunks = kTileBytes / 16;          // 16-B chunks per tile (1024)
constexpr int kChunksPerThread = kTileChunks / kSpliceThreads;   // 4
constexpr int kLongPatch = 96;                        // patches longer than this are filled by a whole wave

// Per-edge patch descriptor: where the edge's label + padding lands in aligned co-ordinates.
struct __attribute__((aligned(16))) edge_patch {
	u32 aln_begin;    // aligned_positions[source node]
	u32 aln_end;      // aligned_positions[target node]
	u32 label_begin;  // offset into the label byte pool
	u32 label_len;
};

// Source and target node of an edge, packed for one 8-B load in the resolve scan.
struct __attribute__((aligned(8))) edge_span {
	u32 src;
	u32 tgt;
};


// ---------------------------------------------------------------------------------------------
// Bit-matrix transpose.  src: n_rows x n_cols bits, column-major, SW = n_rows/64 words per column.
// dst: n_cols x n_rows bits, DW = n_cols/64 words per column.
//
// One workgroup moves a panel of kR row-words x kC column groups through LDS, so that both HBM
// sides move contiguous 8*kR- and 8*kC-byte segments instead of the 8-B words, a whole column
// apart, that a tile-per-wave scheme would touch.  Each wave transposes 64 x 64-bit tiles in
// registers with the six-stage butterfly (lane = source column, bit = source row).
// ---------------------------------------------------------------------------------------------
constexpr int kTrThreads = 256;

__device__ __forceinline__ u64 shfl_xor_u64(u64 v, int mask)
{
	u32 lo = __shfl_xor((u32) v, mask, kWave);
	u32 hi = __shfl_xor((u32) (v >> 32), mask, kWave);
	return ((u64) hi << 32) | lo;
}

// 64 x 64 bit transpose across the wave: on return lane j bit i == (on entry) lane i bit j.
__device__ __forceinline__ u64 wave_transpose_64x64(u64 x, int lane)
{
#define V2M_TR_STAGE(D, LOWMASK)                                                   \
	{                                                                              \
		u64 const y = shfl_xor_u64(x, D);                                          \
		u64 const low = LOWMASK;                                                   \
		x = (lane & D) ? ((x & ~low) | ((y & ~low) >> D)) : ((x & low) | ((y & low) << D)); \
	}
	V2M_TR_STAGE(32, 0x00000000FFFFFFFFULL)
	V2M_TR_STAGE(16, 0x0000FFFF0000FFFFULL)
	V2M_TR_STAGE(8, 0x00FF00FF00FF00FFULL)
	V2M_TR_STAGE(4, 0x0F0F0F0F0F0F0F0FULL)
	V2M_TR_STAGE(2, 0x3333333333333333ULL)
	V2M_TR_STAGE(1, 0x5555555555555555ULL)
#undef V2M_TR_STAGE
	return x;
}

// The same butterfly with gfx950's register-to-register lane exchanges instead of ds_bpermute (which goes through the
// LDS crossbar and costs an LDS round trip per stage): DPP quad/row permutes for distances 1, 2, 4 and 8,
// v_permlane16_swap for 16, and for 32 a single v_permlane32_swap of the word's two halves, which IS the stage
// (lane i's high dword and lane i + 32's low dword change places).
template <int kDist> __device__ __forceinline__ u32 lane_xor_u32(u32 v, int lane)
{
	if constexpr (kDist == 1) return (u32) __builtin_amdgcn_mov_dpp((int) v, 0xB1, 0xF, 0xF, false);         // quad_perm [1,0,3,2]
	else if constexpr (kDist == 2) return (u32) __builtin_amdgcn_mov_dpp((int) v, 0x4E, 0xF, 0xF, false);    // quad_perm [2,3,0,1]
	else if constexpr (kDist == 4)   // i -> 7 - i (row_half_mirror), then i -> 3 - i within the quad: together i ^ 4
		return (u32) __builtin_amdgcn_mov_dpp(__builtin_amdgcn_mov_dpp((int) v, 0x141, 0xF, 0xF, false), 0x1B, 0xF, 0xF, false);
	else if constexpr (kDist == 8) return (u32) __builtin_amdgcn_mov_dpp((int) v, 0x128, 0xF, 0xF, false);   // row_ror:8
	else {
		static_assert(kDist == 16, "distance 32 is handled on the register pair");
		auto const r = __builtin_amdgcn_permlane16_swap(v, v, false, false);   // r[0]: odd 16-lane rows hold their lower neighbour's value; r[1]: even rows hold their upper neighbour's
		return (lane & 16) ? r[0] : r[1];
	}
}

template <int kDist> __device__ __forceinline__ u64 transpose_stage(u64 x, u64 low, int lane)
{
	u64 const y = ((u64) lane_xor_u32<kDist>((u32) (x >> 32), lane) << 32) | lane_xor_u32<kDist>((u32) x, lane);
	return (lane & kDist) ? ((x & ~low) | ((y & ~low) >> kDist)) : ((x & low) | ((y & low) << kDist));
}

__device__ __forceinline__ u64 wave_transpose_64x64_fast(u64 x, int lane)
{
	{
		auto const r = __builtin_amdgcn_permlane32_swap((u32) x, (u32) (x >> 32), false, false);   // lanes 32..63 of the low dword <-> lanes 0..31 of the high dword
		x = ((u64) r[1] << 32) | r[0];
	}
	x = transpose_stage<16>(x, 0x0000FFFF0000FFFFULL, lane);
	x = transpose_stage<8>(x, 0x00FF00FF00FF00FFULL, lane);
	x = transpose_stage<4>(x, 0x0F0F0F0F0F0F0F0FULL, lane);
	x = transpose_stage<2>(x, 0x3333333333333333ULL, lane);
	x = transpose_stage<1>(x, 0x5555555555555555ULL, lane);
	return x;
}

// The butterfly again, written for the instruction count (round 3: the transposes issue ~100 VALU instructions per 512-byte
// tile with the form above, which is most of what a wave does).  Per 64-bit word:
//   distance 32: one v_permlane32_swap of the two dwords (as above);
//   distance 16: the dwords are re-packed into (low halves, high halves) with two v_perm_b32, after which the exchange IS one
//                v_permlane16_swap of the two registers (odd 16-lane rows of the first <-> even rows of the second), and two
//                v_perm_b32 pack them back;
//   distance 8:  whole bytes move: one DPP row rotate per dword, one v_perm_b32 with a per-lane selector;
//   distance 4, 2, 1: with m = the lane's keep mask (low or ~low) the bits taken from the partner are (y & m) rotated by D
//                one way or the other (nothing crosses the dword's ends: low has its top D bits clear), so a dword costs
//                a DPP move + v_and, one v_alignbit_b32 with a per-lane amount and one v_and_or_b32.
// The per-lane constants (selector, masks, rotate amounts) depend on the lane only: a kernel computes them once.
struct lean_butterfly {
	u32 sel8;              // v_perm_b32 selector of the distance-8 stage
	u32 m4, m2, m1;        // keep masks
	u32 r4, r2, r1;        // rotate-right amounts
	__device__ __forceinline__ explicit lean_butterfly(int lane)
		: sel8((lane & 8) ? 0x03070105u : 0x06020400u),
		  m4((lane & 4) ? 0xF0F0F0F0u : 0x0F0F0F0Fu), m2((lane & 2) ? 0xCCCCCCCCu : 0x33333333u), m1((lane & 1) ? 0xAAAAAAAAu : 0x55555555u),
		  r4((lane & 4) ? 4u : 28u), r2((lane & 2) ? 2u : 30u), r1((lane & 1) ? 1u : 31u)
	{
	}

	template <int kDist> __device__ __forceinline__ static u32 partner(u32 v)
	{
		if constexpr (kDist == 1) return (u32) __builtin_amdgcn_mov_dpp((int) v, 0xB1, 0xF, 0xF, false);
		else if constexpr (kDist == 2) return (u32) __builtin_amdgcn_mov_dpp((int) v, 0x4E, 0xF, 0xF, false);
		else if constexpr (kDist == 4) return (u32) __builtin_amdgcn_mov_dpp(__builtin_amdgcn_mov_dpp((int) v, 0x141, 0xF, 0xF, false), 0x1B, 0xF, 0xF, false);
		else { static_assert(kDist == 8, "DPP reaches the lanes of a 16-lane row"); return (u32) __builtin_amdgcn_mov_dpp((int) v, 0x128, 0xF, 0xF, false); }
	}

	template <int kDist> __device__ __forceinline__ static u32 bit_stage(u32 x, u32 m, u32 r)
	{
		u32 const take = partner<kDist>(x) & m;
		return (x & m) | __builtin_amdgcn_alignbit(take, take, r);
	}

	__device__ __forceinline__ u64 operator()(u64 x) const
	{
		u32 lo, hi;
		{
			auto const r = __builtin_amdgcn_permlane32_swap((u32) x, (u32) (x >> 32), false, false);
			lo = r[0]; hi = r[1];
		}
		{
			u32 const e = __builtin_amdgcn_perm(hi, lo, 0x05040100u);     // the low halves of (lo, hi)
			u32 const o = __builtin_amdgcn_perm(hi, lo, 0x07060302u);     // the high halves
			auto const r = __builtin_amdgcn_permlane16_swap(e, o, false, false);   // e's odd rows <-> o's even rows
			lo = __builtin_amdgcn_perm(r[1], r[0], 0x05040100u);
			hi = __builtin_amdgcn_perm(r[1], r[0], 0x07060302u);
		}
		lo = __builtin_amdgcn_perm(partner<8>(lo), lo, sel8);
		hi = __builtin_amdgcn_perm(partner<8>(hi), hi, sel8);
		lo = bit_stage<4>(lo, m4, r4); hi = bit_stage<4>(hi, m4, r4);
		lo = bit_stage<2>(lo, m2, r2); hi = bit_stage<2>(hi, m2, r2);
		lo = bit_stage<1>(lo, m1, r1); hi = bit_stage<1>(hi, m1, r1);
		// (without this the compiler sinks the last stage's two halves to the word's use, a write-out a whole block later, and
		// keeps four registers per word alive instead of two)
		asm volatile("" : "+v"(lo), "+v"(hi));
		return ((u64) hi << 32) | lo;
	}
};

// Workgroups are handed to the 8 XCDs round-robin (block b runs on XCD b % 8) and every XCD has its own L2.  Panels that
// are neighbours in the matrix share 128-B lines on the side whose run is shorter than a line or not line-aligned, so the
// work items (numbered so that neighbours are consecutive) are cut into 8 contiguous chunks, one per XCD: a shared line
// is then fetched / written back by ONE L2, by workgroups that run at about the same time.
__device__ __forceinline__ bool xcd_chunked_item(u32 block, u64 n_items, u32 items_per_xcd, u64 &item)
{
	if (0 == items_per_xcd) { item = block; return item < n_items; }   // plain dispatch order (A/B switch of the tuning tools)
	u32 const xcd = block & 7, i = block >> 3;
	item = (u64) xcd * items_per_xcd + i;
	return i < items_per_xcd && item < n_items;
}

// Panel of kR source row-words x kC source column groups (64*kR x 64*kC bits).  Every source column contributes
// 8*kR contiguous bytes, every destination column receives 8*kC contiguous bytes.
template <int kR, int kC>
__global__ __launch_bounds__(kTrThreads) void transpose_bits_kernel(
	u64 const *__restrict__ src, u64 *__restrict__ dst, u64 SW, u64 DW, u64 src_pitch, u64 dst_pitch,
	u32 n_row_panels, u32 n_col_panels, u32 items_per_xcd, u32 rows_fastest)
{
	static_assert(kR % 4 == 0, "each of the 4 waves owns kR / 4 row-words");
	// phase 1 view: [source column within panel = 64*kC][kR + 1]; phase 2 view: [destination column = 64*kR][kC + 1].
	// The one-word pad makes the per-lane stride of the strided accesses odd in 8-byte units: conflict-free ds_read/write_b64.
	constexpr int kWords1 = 64 * kC * (kR + 1), kWords2 = 64 * kR * (kC + 1);
	__shared__ u64 panel[kWords1 > kWords2 ? kWords1 : kWords2];

	int const t = threadIdx.x;
	int const lane = t & 63;
	int const wave = t >> 6;
	u64 item;
	if (!xcd_chunked_item(blockIdx.x, (u64) n_row_panels * n_col_panels, items_per_xcd, item)) return;   // whole workgroup
	// rows_fastest: vertically adjacent panels (which share source lines) are consecutive items; otherwise horizontally
	// adjacent ones (which share destination lines).  The host picks per shape.
	u64 const rw0 = (rows_fastest ? item % n_row_panels : item / n_col_panels) * kR;    // first source row-word
	u64 const cg0 = (rows_fastest ? item / n_row_panels : item % n_col_panels) * kC;    // first source column group
	u64 const n_cols = DW * 64;

	// load: kR consecutive lanes fetch one column's 8*kR contiguous bytes
	for (int k = 0; k < (64 * kR * kC) / kTrThreads; ++k) {
		int const idx = t + kTrThreads * k;
		int const col = idx / kR, w = idx % kR;
		u64 const gcol = cg0 * 64 + col;
		u64 v = 0;
		if (gcol < n_cols && rw0 + w < SW)
			v = src[gcol * src_pitch + rw0 + w];
		panel[col * (kR + 1) + w] = v;
	}
	__syncthreads();

	// each wave owns kR / 4 row-words; pull their tiles into registers
	constexpr int kA = kR / 4;
	lean_butterfly const butterfly(lane);
	u64 x[kA][kC];
#pragma unroll
	for (int a = 0; a < kA; ++a)
#pragma unroll
		for (int cg = 0; cg < kC; ++cg)
			x[a][cg] = panel[(64 * cg + lane) * (kR + 1) + kA * wave + a];
	__syncthreads();

#pragma unroll
	for (int a = 0; a < kA; ++a)
#pragma unroll
		for (int cg = 0; cg < kC; ++cg)
			panel[(64 * (kA * wave + a) + lane) * (kC + 1) + cg] = wave_transpose_64x64(x[a][cg], lane);
	__syncthreads();

	// store: kC consecutive lanes write one destination column's 8*kC contiguous bytes
	for (int k = 0; k < (64 * kR * kC) / kTrThreads; ++k) {
		int const idx = t + kTrThreads * k;
		int const dcol = idx / kC, cw = idx % kC;
		u64 const grow = rw0 * 64 + dcol;          // destination column = source row
		if (grow < SW * 64 && cg0 + cw < DW)
			dst[grow * dst_pitch + cg0 + cw] = panel[dcol * (kC + 1) + cw];
	}
}


// Streaming variant: a workgroup owns kR = 16 source row-words (a full 128-B line of every source column) and
// kC = 16 column groups (a full 128-B line of every destination column), but only one 64-column group of the
// source is in LDS at a time (double-buffered, one barrier per group).  Each wave keeps the transposed tiles of
// its 4 row-words in registers across the 16 groups (lane = destination column, 16 words = 128 contiguous
// bytes of it) and writes them out through a wave-private LDS slab, so that 16 consecutive lanes store one
// destination column's 128 bytes.  Both HBM sides move whole 128-B runs.

// kTsDepth: source sub-panels in flight per workgroup (registers).  kSlabCols: destination columns of a wave that pass
// through its write-out slab at a time (64 = all of a tile's at once; 32 = two passes over a slab half the size, which
// takes the kernel from 52 KB of LDS -- 3 workgroups per CU -- to 35 KB -- 4).
// kTsR x kTsC: row-words x column groups of a workgroup's block (16 x 16: a 128-B line on both sides; 8 x 32: 64-B source runs,
// 256-B destination runs, for a dense destination whose columns start at arbitrary 8-byte offsets).
template <int kTsDepth, int kSlabCols, int kWaves = 4, int kTsR = 16, int kTsC = 16, bool kLean = true>
__global__ __launch_bounds__(64 * kWaves) void transpose_bits_stream_kernel(
	u64 const *__restrict__ src, u64 *__restrict__ dst, u64 SW, u64 DW, u64 src_pitch, u64 dst_pitch,
	u32 n_row_panels, u32 n_col_panels, u32 items_per_xcd, u32 rows_fastest)
{
	static_assert(64 % kSlabCols == 0 && kSlabCols >= 16, "the slab takes a whole fraction of a tile's 64 destination columns");
	constexpr int kThreads = 64 * kWaves;
	static_assert(kTsR % kWaves == 0 && (64 * kTsR) % kThreads == 0 && (kTsC * kSlabCols) % 64 == 0, "whole row-words per wave, whole words per thread");
	constexpr int kA = kTsR / kWaves;                    // row-words per wave
	__shared__ u64 in[2][64][kTsR + 1];
	__shared__ u64 slab[kWaves][kSlabCols][kTsC + 1];

	int const t = threadIdx.x;
	int const lane = t & 63;
	int const wave = t >> 6;
	u64 item;
	if (!xcd_chunked_item(blockIdx.x, (u64) n_row_panels * n_col_panels, items_per_xcd, item)) return;   // whole workgroup
	u64 const rw0 = (rows_fastest ? item % n_row_panels : item / n_col_panels) * kTsR;
	u64 const cg0 = (rows_fastest ? item / n_row_panels : item % n_col_panels) * kTsC;
	u64 const n_cols = DW * 64;

	// one sub-panel = 64 columns x 16 row-words; 16 consecutive lanes fetch one column's 128 contiguous bytes.
	// kTsDepth sub-panels are kept in flight in registers (the kernel is bound by bytes in flight, not by the shuffles).
	// Every load is issued unconditionally from an address clamped into the matrix and zeroed at the stash where it lies
	// outside: loads under branches made the compiler drain the whole load queue at every barrier (see the ring kernel).
	constexpr int kPer = (64 * kTsR) / kThreads;
	u64 stage[kTsDepth][kPer];
	u64 const last_col = n_cols - 1;
	auto fetch = [&](int cg, u64 (&st)[kPer]) {
#pragma unroll
		for (int k = 0; k < kPer; ++k) {
			int const idx = t + kThreads * k;
			int const col = idx / kTsR, w = idx % kTsR;
			u64 const gcol = (cg0 + cg) * 64 + col;
			st[k] = src[(gcol < n_cols ? gcol : last_col) * src_pitch + (rw0 + w < SW ? rw0 + w : SW - 1)];
		}
	};
	auto stash = [&](int buf, int cg, u64 const (&st)[kPer]) {
#pragma unroll
		for (int k = 0; k < kPer; ++k) {
			int const idx = t + kThreads * k;
			int const col = idx / kTsR, w = idx % kTsR;
			in[buf][col][w] = ((cg0 + cg) * 64 + col < n_cols && rw0 + w < SW) ? st[k] : 0;
		}
	};

	u64 y[kA][kTsC];
	lean_butterfly const butterfly(lane);
#pragma unroll
	for (int cg = 0; cg < kTsDepth; ++cg) fetch(cg, stage[cg]);
	stash(0, 0, stage[0]);
#pragma unroll
	for (int cg = 0; cg < kTsC; ++cg) {
		__syncthreads();                                 // in[cg & 1] is complete; everyone is done with in[(cg + 1) & 1]
		if (cg + kTsDepth < kTsC) fetch(cg + kTsDepth, stage[cg % kTsDepth]);   // stage[cg % kTsDepth] was stashed for this group already
#pragma unroll
		for (int a = 0; a < kA; ++a)
			y[a][cg] = kLean ? butterfly(in[cg & 1][lane][kA * wave + a]) : wave_transpose_64x64_fast(in[cg & 1][lane][kA * wave + a], lane);
		if (cg + 1 < kTsC) stash((cg + 1) & 1, cg + 1, stage[(cg + 1) % kTsDepth]);
		if (kLean) __builtin_amdgcn_sched_barrier(0);              // (the steps stay apart in the schedule: merged, the compiler drains the whole load queue -- s_waitcnt vmcnt(0) -- nine times per block instead of once)
	}

	// write-out, one row-word at a time through the wave's own slab (LDS operations of one wave execute in order)
#pragma unroll
	for (int a = 0; a < kA; ++a) {
		u64 const rw = rw0 + kA * wave + a;
#pragma unroll
		for (int part = 0; part < 64 / kSlabCols; ++part) {
			// the destination columns [kSlabCols * part, + kSlabCols) of the tile: their lanes drop their 16 words into the slab ...
			if (kSlabCols == 64 || lane / kSlabCols == part) {
#pragma unroll
				for (int cg = 0; cg < kTsC; ++cg)
					slab[wave][lane % kSlabCols][cg] = y[a][cg];
			}
			// ... and all 64 lanes store them, 16 consecutive lanes one column's 128 bytes
			if (rw < SW) {
#pragma unroll
				for (int k = 0; k < kTsC * kSlabCols / 64; ++k) {
					int const idx = lane + 64 * k;
					int const dcol = idx / kTsC, cw = idx % kTsC;
					if (cg0 + cw < DW)
						dst[(rw * 64 + kSlabCols * part + dcol) * dst_pitch + cg0 + cw] = slab[wave][dcol][cw];
				}
			}
		}
	}
}


// Sector-aligned streaming transpose ("ring" kernel).
//
// What the two kernels above lose at the reference's own padding (dimensions that are multiples of 64 and nothing
// more, transpose_matrix.cc:53-54) is alignment: a destination column starts at byte 8 * r * DW, so their 64-B / 128-B
// runs begin at arbitrary 8-byte offsets, straddle 64-B DRAM sectors and 128-B lines, and leave the L2s writing most
// sectors in two masked pieces.  Here the destination side is addressed as the flat word array it is: a workgroup
// owns kR source row-words and streams along a span of source column groups, one group (64 columns) per step; each wave
// transposes its tiles in registers (lane = destination row) and drops every lane's word into a wave-private LDS ring of
// kS words per destination row, at slot (flat word index mod kS).  A row whose slot kS - 1 has just been written owns a
// complete, naturally aligned kS * 8-byte sector: kS consecutive lanes store it with one coalesced piece of a store
// instruction.  Rows reach their boundaries at different steps (row r's phase is r * DW mod kS), so every step a few
// rows flush -- 64 / kS of them when DW is odd -- and the stores spread evenly over the stream.  Which rows: the set is an
// arithmetic progression of lanes (a solution set of j * DW = c mod kS), read off the ballot.
//
// Span edges: row r's window is the span moved back to r's nearest sector boundary, so that windows of neighbouring
// spans meet on sector boundaries; the workgroup therefore starts kS - 1 groups early and a row simply does not store
// sectors outside its window.  Only the two ends of a destination row (where its first / last sector is shared with
// the neighbouring rows) are written as partial sectors.
//
// Source side: the kR-word runs (128 B for kR = 16) of a column start at arbitrary 8-byte offsets too, so vertically
// adjacent panels share their first / last line; xcd_chunked_item() puts them on one XCD at the same time and the
// shared lines are fetched from HBM once.  Source words are prefetched kD steps ahead in registers and staged through
// a double-buffered LDS tile (one barrier per step); the ring needs no barrier (LDS operations of one wave execute in order).
// ---------------------------------------------------------------------------------------------
template <int kR, int kW, int kS, int kD, bool kFastLanes, bool kNonTemporal>
__global__ __launch_bounds__(64 * kW) void transpose_bits_ring_kernel(
	u64 const *__restrict__ src, u64 *__restrict__ dst, u64 SW, u64 DW, u64 src_pitch, u64 dst_pitch,
	u32 n_panels, u32 n_spans, u32 span_groups /* multiple of kS */, u32 items_per_xcd, u32 panel_fastest)
{
	static_assert(kR % kW == 0 && (kS & (kS - 1)) == 0 && kS >= 2 && kS <= 16 && kD % 2 == 0, "geometry");
	constexpr int kA = kR / kW;              // row-words (tiles per step) per wave; also source words per thread per step
	constexpr int kT = 64 * kW;
	constexpr int kLgS = kS == 2 ? 1 : kS == 4 ? 2 : kS == 8 ? 3 : 4;
	constexpr int kSectorsPerStore = 64 / kS;
	__shared__ u64 stage[2][64][kR + 1];     // [source column within the group][row-word], odd stride: conflict-free column reads
	__shared__ u64 ring[kW][kA][64][kS + 1]; // [wave][tile][destination row within the tile][slot]

	u64 item;
	if (!xcd_chunked_item(blockIdx.x, (u64) n_panels * n_spans, items_per_xcd, item)) return;   // whole workgroup
	u32 const panel = panel_fastest ? (u32) (item % n_panels) : (u32) (item / n_spans);
	u32 const span = panel_fastest ? (u32) (item / n_panels) : (u32) (item % n_spans);

	int const t = threadIdx.x, lane = t & 63, wave = t >> 6;
	u64 const rw0 = (u64) panel * kR;
	u64 const span_begin = (u64) span * span_groups;                                   // nominal window [span_begin, span_end)
	u64 const span_end = (span_begin + span_groups < DW) ? span_begin + span_groups : DW;
	bool const first_span = 0 == span, last_span = span_end == DW;
	u32 const cg_lo = first_span ? 0u : (u32) span_begin - (kS - 1);
	u32 const n_steps = (u32) span_end - cg_lo;

	// Source words of this thread: column (t + kT * k) / kR of the group, row-word (t + kT * k) % kR of the panel.
	// Every fetch is issued unconditionally, from an address clamped into the matrix (row-words past the matrix become
	// zeros at the stash, steps past the span re-read its last group): the number of loads between a fetch and its use is
	// then a compile-time constant and the compiler can wait with a counted s_waitcnt vmcnt(n) instead of vmcnt(0) --
	// with loads under branches it drained the whole queue every step and the kernel ran at one memory latency per step.
	u64 src_off[kA];
	bool src_ok[kA];
#pragma unroll
	for (int k = 0; k < kA; ++k) {
		int const idx = t + kT * k;
		src_ok[k] = rw0 + idx % kR < SW;
		src_off[k] = (u64) (idx / kR) * src_pitch + (src_ok[k] ? rw0 + idx % kR : SW - 1);
	}
	u32 const cg_last = (u32) span_end - 1;
	u64 pf[kD][kA];
	auto const fetch = [&](u64 (&r)[kA], u32 cg) {
		u64 const *const base = src + (u64) (cg < cg_last ? cg : cg_last) * 64 * src_pitch;
#pragma unroll
		for (int k = 0; k < kA; ++k) r[k] = kNonTemporal ? __builtin_nontemporal_load(base + src_off[k]) : base[src_off[k]];
	};
	auto const stash = [&](int buf, u64 const (&r)[kA]) {
#pragma unroll
		for (int k = 0; k < kA; ++k) {
			int const idx = t + kT * k;
			stage[buf][idx / kR][idx % kR] = src_ok[k] ? r[k] : 0;
		}
	};

	// destination rows of this wave's tiles: tile a = row-word rw0 + kA * wave + a, lane = row within it
	u64 tile_base[kA];                        // flat destination word index of the tile's row 0, column group 0
	u64 const lane_off = (u64) lane * dst_pitch;
#pragma unroll
	for (int a = 0; a < kA; ++a) tile_base[a] = (rw0 + kA * wave + a) * 64 * dst_pitch;

	lean_butterfly const butterfly(lane);
	auto const compute = [&](u32 cg, int buf) {
		u64 tv[kA];
#pragma unroll
		for (int a = 0; a < kA; ++a) tv[a] = stage[buf][lane][kA * wave + a];
#pragma unroll
		for (int a = 0; a < kA; ++a) tv[a] = kFastLanes ? butterfly(tv[a]) : wave_transpose_64x64(tv[a], lane);   // (round 5: the lean butterfly of the streaming kernels)
		bool done[kA];
#pragma unroll
		for (int a = 0; a < kA; ++a) {
			u32 const slot = ((u32) tile_base[a] + (u32) lane_off + cg) & (kS - 1);
			ring[wave][a][lane][slot] = tv[a];
			done[a] = slot == kS - 1 || cg + 1 == (u32) DW;
		}
		__builtin_amdgcn_wave_barrier();
#pragma unroll
		for (int a = 0; a < kA; ++a) {
			if (rw0 + kA * wave + a >= SW) continue;                                   // wave-uniform: row-words past the matrix
			u64 const mask = __ballot(done[a]);
			if (0 == mask) continue;
			// the rows that completed a sector at this step: lanes j0, j0 + stride, ... (n of them, n a power of two)
			u32 const n = (u32) __builtin_popcountll(mask), j0 = (u32) __builtin_ctzll(mask);
			u32 const lg_stride = 6 - (31 - (u32) __builtin_clz(n));
			u32 const w = lane & (kS - 1);
			for (u32 f = 0; f * kSectorsPerStore < n; ++f) {
				u32 const idx = f * kSectorsPerStore + (lane >> kLgS);
				if (idx >= n) continue;
				u32 const row = j0 + (idx << lg_stride);
				u64 const row_base = tile_base[a] + (u64) row * dst_pitch;              // flat index of the row's word 0
				u64 const g = row_base + cg;
				u32 const slot = (u32) g & (kS - 1);
				u64 const g0 = g - slot;                                                // the sector's first word
				bool ok = w <= slot && g0 + w >= row_base;                             // row end / row start: the rest belongs to the neighbouring rows
				if (!first_span) ok = ok && g0 >= ((row_base + span_begin) & ~(u64) (kS - 1));
				if (!last_span) ok = ok && g0 < ((row_base + span_end) & ~(u64) (kS - 1));
				if (ok) {
					u64 const v = ring[wave][a][row][w];
					if (kNonTemporal) __builtin_nontemporal_store(v, dst + g0 + w);
					else dst[g0 + w] = v;
				}
			}
		}
	};

#pragma unroll
	for (int j = 0; j < kD; ++j) fetch(pf[j], cg_lo + j);
	stash(0, pf[0]);
	for (u32 base = 0; base < n_steps; base += kD) {
#pragma unroll
		for (int j = 0; j < kD; ++j) {
			u32 const st = base + j;
			if (st >= n_steps) break;
			// The stash of the previous step must have landed in LDS before anyone passes the barrier.  __syncthreads() implies
			// that wait, but hipcc (ROCm 7.2) drops it on the loop's back edge here (the header's s_barrier came out with no
			// s_waitcnt lgkmcnt(0) on the path from the last step's ds_write): a few hundred wrong words per 79 M, now and then.
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
			__syncthreads();                       // stage[j & 1] is complete; everyone is done with the other buffer
			fetch(pf[j], cg_lo + st + kD);         // pf[j] was stashed during the previous step
			compute(cg_lo + st, j & 1);
			stash((j + 1) & 1, pf[(j + 1) % kD]);  // (after the last step: a stash nobody reads)
		}
	}
}


// Whole-line streaming transpose ("lines" kernel, round 3).
//
// What the dense form (destination columns at the reference's own padding, transpose_matrix.cc:53-54: a column starts at byte
// 8 * r * DW) loses with the streaming kernel is that each of its 128-byte destination runs straddles two 128-B lines, so
// every line is written in two masked pieces by two workgroups; the ring kernel above stores whole sectors but pays for its
// per-step bookkeeping.  This kernel keeps the streaming kernel's structure -- transposed tiles accumulated in registers over
// 16 column groups, written out through a wave-private LDS slab, 16 consecutive lanes per destination column -- and streams
// on along a span of several 16-group blocks, carrying the previous block's words in registers: destination column r's lines
// begin at its words s_r, s_r + 16, ... with s_r = (-r * dst_pitch) mod 16, so after block b the wave holds the whole line
// [16 (b - 1) + s_r, 16 b + s_r) of every one of its columns -- the previous block's words from s_r on, the current block's
// words before s_r.  Both blocks' words go into the slab side by side (32 words per column, no per-lane select anywhere) and
// the column's line is read back from word s_r of them: every store is a whole, aligned 128-byte line.  Only the first and
// last line of a span are written in two pieces (by the neighbouring spans' workgroups), and with a line-aligned pitch
// (the library's own path matrix) s_r is 0 and the kernel degenerates into the streaming kernel.
// A workgroup owns kTsR source row-words (the panels that share the source lines are neighbours in item order and on one XCD).
// Product geometries: "lines8" = 8 row-words on 8 waves (64-byte source runs, one tile per wave, 112 / 146 VGPRs) and, since round 5,
// "lines16" = 16 row-words on 8 waves, two tiles per wave (128-byte source runs, 185 / 253 VGPRs, one workgroup per CU): the L2's request
// counters (profiles/r05/transpose_pmc.txt) showed that the 64-byte runs of a dense source, seven in eight of which straddle a sector, cost
// lines8 14.4 M read requests per launch of the config-3 matrix where whole lines would be 5 M and 128-byte runs are 9.6 M -- and the
// transposes' times follow the request count.  Dense forward 0.30-0.34 -> 0.26-0.28 ms (profiles/r05/transpose_lines16.txt).
template <int kWaves, int kDepth, int kTsR = 8, int kSlabRows = 32, bool kMerge = false>
__global__ __launch_bounds__(64 * kWaves) void transpose_bits_lines_kernel(
	u64 const *__restrict__ src, u64 *__restrict__ dst, u64 SW, u64 DW, u64 src_pitch, u64 dst_pitch,
	u32 n_panels, u32 n_spans, u32 span_blocks, u32 items_per_xcd, u32 panel_fastest)
{
	constexpr int kThreads = 64 * kWaves, kA = kTsR / kWaves, kPer = (64 * kTsR) / kThreads;
	static_assert(kTsR % kWaves == 0 && 16 % kDepth == 0 && (64 * kTsR) % kThreads == 0 && (32 == kSlabRows || 16 == kSlabRows), "geometry");
	__shared__ u64 in[2][64][kTsR + 1];
	__shared__ u64 slab[kWaves][kSlabRows][33];  // [wave][destination column of the part of the tile][previous block's 16 words, this block's 16 words]

	u64 item64;
	if (!xcd_chunked_item(blockIdx.x, (u64) n_panels * n_spans, items_per_xcd, item64)) return;   // whole workgroup
	u32 const item = (u32) item64;                                 // (a grid has fewer than 2^31 workgroups)
	u32 const panel = panel_fastest ? item % n_panels : item / n_spans;
	u32 const span = panel_fastest ? item / n_panels : item % n_spans;

	int const t = threadIdx.x, lane = t & 63;
	int const wave = __builtin_amdgcn_readfirstlane(t >> 6);
	u64 const rw0 = (u64) panel * kTsR;
	u32 const n_blocks = ((u32) DW + 15) / 16;                     // (the host checks that DW fits comfortably)
	u32 const b_lo = span * span_blocks, b_hi = (b_lo + span_blocks < n_blocks) ? b_lo + span_blocks : n_blocks;
	u32 const span_lo = 16 * b_lo, span_hi = (16 * b_hi < (u32) DW) ? 16 * b_hi : (u32) DW;   // this workgroup's words of every destination column

	// source: as in the streaming kernel, every load unconditional from an address clamped into the matrix (what lies outside is
	// never stored: destination columns past SW * 64 and words past span_hi are masked at the store)
	u32 src_off[kPer];                                             // byte offset inside a column group (the host checks 512 * src_pitch < 2^32)
#pragma unroll
	for (int k = 0; k < kPer; ++k) {
		int const idx = t + kThreads * k;
		u64 const w = rw0 + idx % kTsR;
		src_off[k] = (u32) (((u64) (idx / kTsR) * src_pitch + (w < SW ? w : SW - 1)) * 8);
	}
	u32 const cg_max = (u32) DW - 1;
	auto const fetch = [&](u64 (&st)[kPer], u32 cg) {
		char const *const base = reinterpret_cast<char const *>(src + (u64) (cg < cg_max ? cg : cg_max) * 64 * src_pitch);   // uniform
#pragma unroll
		for (int k = 0; k < kPer; ++k) st[k] = *reinterpret_cast<u64 const *>(base + src_off[k]);
	};
	auto const stash = [&](int buf, u64 const (&st)[kPer]) {
#pragma unroll
		for (int k = 0; k < kPer; ++k) {
			int const idx = t + kThreads * k;
			in[buf][idx / kTsR][idx % kTsR] = st[k];
		}
	};

	// destination: lane -> (column of the part of the tile, word of its line) for the stores of a part.  Column 4 k + lane / 16
	// of the part (16 or 32 * part + that of the tile, and 16 * pitch = 0 mod 16) has s = (s_lane - 4 k * pitch) mod 16: four
	// values, k and k + 4 share one.  The stores take a uniform base (SGPRs) and a 32-bit byte offset per lane (the host checks
	// that 4 * dst_pitch + DW words stay below 2^29).
	u32 const lane_col = (u32) lane >> 4, lane_word = (u32) lane & 15;
	u32 const pitch_lo = (u32) dst_pitch & 15;
	u32 s_word[4];                                                 // s + the lane's word of the line
#pragma unroll
	for (int k = 0; k < 4; ++k) s_word[k] = ((0u - (lane_col + 4 * k) * pitch_lo) & 15) + lane_word;
	u32 const lane_col_off = lane_col * (u32) dst_pitch;

	u64 y_prev[kA][16], y_cur[kA][16];
#pragma unroll
	for (int a = 0; a < kA; ++a)
#pragma unroll
		for (int c = 0; c < 16; ++c) y_prev[a][c] = 0;

	// Odd spans stream BACKWARDS.  The line a span shares with its neighbour is written in two pieces, one by either workgroup;
	// when both stream forwards one piece is written at the end of a workgroup's life and the other at the start of the next
	// one's, a whole workgroup's run apart: the line has left the L2 in between and the memory sees two partial writes (measured:
	// this cost as much as a sixth of the kernel's time).  With the directions alternating, neighbours reach their common
	// boundary at the same end of their lives, and they start together (neighbouring items, same XCD).
	bool const reverse = 0 != (span & 1);

	// kMerge (the host picks it when a workgroup takes whole destination columns of a DENSE destination, dst_pitch == DW >= 16):
	// column r's last line and column r + 1's first are then one line of memory, and its two pieces would be written a whole
	// run apart by the same wave.  Instead every line is written once, by the column it BEGINS in: the wave keeps its first
	// block's words (y_first) to the end, and in the last two rounds column r + 1's first words go into column r's slab row
	// behind its last words -- lane l + 1 writes them there.  Only a tile's first column writes its head on its own (the
	// column before it belongs to another wave) and a tile's last column its tail.
	u64 y_first[kMerge ? kA : 1][16];
	u32 const n_blocks_all = n_blocks;

	// the lines that begin in block e: its words from s_r on (the slab's first 16 words), the next block's words before s_r (the
	// next 16).  Forwards e is the block before the one just computed, backwards the one just computed.
	u32 const slab_cur = reverse ? 0u : 16u, slab_prev = 16u - slab_cur;
	auto const emit = [&](u32 e) {
		bool const head_round = kMerge && e + 1 == 0;              // uniform
		bool const overlay = kMerge && e + 2 >= n_blocks_all;      // uniform: one of the last two rounds (or the head round of a one-block column)
		u32 const overlay_at = (u32) DW - 16 * e;                  // slab position of the next column's word 0
#pragma unroll
		for (int a = 0; a < kA; ++a) {
			u64 const rw = rw0 + (u64) (kA * wave + a);            // wave-uniform
			bool const rw_ok = rw < SW;
#pragma unroll
			for (int part = 0; part < 64 / kSlabRows; ++part) {
				if (head_round && part > 0) continue;
				if (lane / kSlabRows == part) {
#pragma unroll
					for (int c = 0; c < 16; ++c) {
						slab[wave][lane % kSlabRows][slab_prev + c] = y_prev[a][c];
						slab[wave][lane % kSlabRows][slab_cur + c] = y_cur[a][c];
					}
				}
				if (kMerge) {
					if (overlay && lane >= 1 && (lane - 1) / kSlabRows == part) {
#pragma unroll
						for (int c = 0; c < 16; ++c) {
							u32 const at = overlay_at + c;         // (position 32 is the row's pad word: what does not fit is not needed)
							slab[wave][(lane - 1) % kSlabRows][at < 32 ? at : 32] = y_first[kMerge ? a : 0][c];
						}
					}
				}
				__builtin_amdgcn_wave_barrier();                   // (LDS operations of one wave execute in order)
				u64 v[kSlabRows / 4];
#pragma unroll
				for (int k = 0; k < kSlabRows / 4; ++k) v[k] = slab[wave][lane_col + 4 * k][s_word[k & 3]];
#pragma unroll
				for (int k = 0; k < kSlabRows / 4; ++k) {
					u32 const c_rel = 16 * e + s_word[k & 3];      // (e = -1 wraps into a huge value or back into the first words: [span_lo, span_hi) decides)
					char *const base = reinterpret_cast<char *>(dst + (rw * 64 + kSlabRows * part + 4 * k) * dst_pitch);   // uniform
					bool ok;
					if (kMerge) {
						u32 const begins = c_rel - lane_word;      // the line's first word, column-relative (huge: it begins in the column before)
						bool const first_col = 0 == part && 0 == k && 0 == lane_col;
						bool const last_col = 64 / kSlabRows - 1 == part && kSlabRows / 4 - 1 == k && 3 == lane_col;
						bool const own = c_rel < (u32) DW;
						ok = own ? (begins < (u32) DW || first_col) : (begins < (u32) DW && !last_col);
					}
					else ok = c_rel >= span_lo && c_rel < span_hi;
					if (rw_ok && ok)
						*reinterpret_cast<u64 *>(base + (lane_col_off + c_rel) * 8u) = v[k];
				}
				__builtin_amdgcn_wave_barrier();
			}
		}
	};

	// step st of the span: block st / 16 from its first (forwards) or last (backwards) block, group st % 16 of the block
	u32 const n_span_blocks = b_hi - b_lo;
	auto const group_of = [&](u32 st) -> u32 {
		u32 const i = st >> 4;
		if (i >= n_span_blocks) return cg_max;                     // (prefetches past the span's end: any group will do)
		return 16 * (reverse ? b_hi - 1 - i : b_lo + i) + (st & 15);
	};

	lean_butterfly const butterfly(lane);
	u64 stage[kDepth][kPer];
#pragma unroll
	for (int j = 0; j < kDepth; ++j) fetch(stage[j], group_of(j));
	stash(0, stage[0]);
	// one round more than the span has blocks: the last one computes nothing (y_cur = y_prev = the span's last block) and emits
	// that block's words from s_r on (forwards) / before s_r (backwards: the line begins in the block before the span)
	for (u32 i = 0; i <= n_span_blocks; ++i) {
		if (i < n_span_blocks) {
#pragma unroll
			for (int c = 0; c < 16; ++c) {
				asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (see the ring kernel: the wait __syncthreads() implies can get lost on a loop's back edge)
				__syncthreads();                                   // in[c & 1] is complete; everyone is done with in[(c + 1) & 1]
				fetch(stage[c % kDepth], group_of(16 * i + c + kDepth));   // stage[c % kDepth] was stashed for this step already
#pragma unroll
				for (int a = 0; a < kA; ++a) y_cur[a][c] = butterfly(in[c & 1][lane][kA * wave + a]);
				stash((c + 1) & 1, stage[(c + 1) % kDepth]);
				__builtin_amdgcn_sched_barrier(0);                 // (as in the streaming kernel: the steps stay apart in the schedule)
			}
		}
		if (kMerge && 0 == i) {
#pragma unroll
			for (int a = 0; a < kA; ++a)
#pragma unroll
				for (int c = 0; c < 16; ++c) y_first[a][c] = y_cur[a][c];
		}
		emit(reverse ? b_hi - 1 - i : b_lo + i - 1);
#pragma unroll
		for (int a = 0; a < kA; ++a)
#pragma unroll
			for (int c = 0; c < 16; ++c) y_prev[a][c] = y_cur[a][c];
	}
}


#ifdef V2M_TUNING_BUILD
// Rotating-line streaming transpose ("rot" kernel, round 5): an experiment that LOST and lives in the tuning build only
// (profiles/r05/transpose_rot8_experiment.txt: 0.51 / 0.75 ms on the dense config-3 matrix against lines8's 0.30 / 0.32 -- a line stored as eight
// 16-byte pieces by ONE lane is eight requests to the L2 where a coalesced store is one).
//
// The whole-line kernel above pays for its whole lines with 64 registers of carried words (y_prev + y_cur) on top of everything else:
// 146-158 VGPRs, ONE workgroup of 8 waves per CU, 16 KB of loads in flight per CU.  But a destination column needs no carried block: lane l
// of a tile keeps the words of its destination column in a file of 16 registers indexed by the step (y[c] = the word of column group
// cg = 16 b + c, overwritten 16 steps later), and column r's line ends at the steps with (base + r * pitch + cg) % 16 == 15 -- one fixed c per
// lane, done_at.  At that step the 16 registers ARE the line, rotated: word j of the line is y[(c + 1 + j) % 16], a compile-time index in
// the unrolled step.  So every step the (four, for an odd pitch) lanes whose lines have just ended store them, each lane its own 128 bytes as
// 16-byte pieces straight from its registers (eight global_store_dwordx4 with immediate offsets; a line that starts at an odd register takes
// seven of them and two 8-byte ones, so that every piece is a naturally aligned register pair): no slab, no LDS on the way out, the stores
// spread evenly over the stream, and the L2 receives each line's pieces back to back from one wave.  32 registers of state per tile.
// A span's first 15 steps and the up to 15 lines still open at its end hold words of the neighbouring spans: those lines go out through a
// small per-wave LDS slab, word by word under the span's bounds (the neighbours write the rest), as do steps past the matrix.
// CPU replay of the indexing: tools/rot_transpose_model.py (also a -m "not gpu" test).
template <int kWaves, int kDepth, int kTsR>
__global__ __launch_bounds__(64 * kWaves) void transpose_bits_rot_kernel(
	u64 const *__restrict__ src, u64 *__restrict__ dst, u64 SW, u64 DW, u64 src_pitch, u64 dst_pitch,
	u32 n_panels, u32 n_spans, u32 span_blocks, u32 items_per_xcd, u32 panel_fastest)
{
	constexpr int kThreads = 64 * kWaves, kA = kTsR / kWaves, kPer = (64 * kTsR) / kThreads;
	static_assert(kTsR % kWaves == 0 && 16 % kDepth == 0 && (64 * kTsR) % kThreads == 0, "geometry");
	__shared__ u64 in[2][64][kTsR + 1];
	__shared__ u64 slab[kWaves][4][17];          // the guarded flavour: four lines at a time, [line][register of the file]
	__shared__ u32 slab_lane[kWaves][4];

	u64 item64;
	if (!xcd_chunked_item(blockIdx.x, (u64) n_panels * n_spans, items_per_xcd, item64)) return;   // whole workgroup
	u32 const item = (u32) item64;
	u32 const panel = panel_fastest ? item % n_panels : item / n_spans;
	u32 const span = panel_fastest ? item / n_panels : item % n_spans;

	int const t = threadIdx.x, lane = t & 63;
	int const wave = __builtin_amdgcn_readfirstlane(t >> 6);
	u64 const rw0 = (u64) panel * kTsR;
	u32 const n_blocks = ((u32) DW + 15) / 16;
	u32 const b_lo = span * span_blocks, b_hi = (b_lo + span_blocks < n_blocks) ? b_lo + span_blocks : n_blocks;
	u32 const span_lo = 16 * b_lo, span_hi = (16 * b_hi < (u32) DW) ? 16 * b_hi : (u32) DW;
	u32 const n_span_blocks = b_hi - b_lo;

	// source: as in the lines kernel (every load unconditional from an address clamped into the matrix)
	u32 src_off[kPer];
#pragma unroll
	for (int k = 0; k < kPer; ++k) {
		int const idx = t + kThreads * k;
		u64 const w = rw0 + idx % kTsR;
		src_off[k] = (u32) (((u64) (idx / kTsR) * src_pitch + (w < SW ? w : SW - 1)) * 8);
	}
	u32 const cg_max = (u32) DW - 1;
	auto const fetch = [&](u64 (&st)[kPer], u32 cg) {
		char const *const base = reinterpret_cast<char const *>(src + (u64) (cg < cg_max ? cg : cg_max) * 64 * src_pitch);   // uniform
#pragma unroll
		for (int k = 0; k < kPer; ++k) st[k] = *reinterpret_cast<u64 const *>(base + src_off[k]);
	};
	auto const stash = [&](int buf, u64 const (&st)[kPer]) {
#pragma unroll
		for (int k = 0; k < kPer; ++k) {
			int const idx = t + kThreads * k;
			in[buf][idx / kTsR][idx % kTsR] = st[k];
		}
	};

	// destination: per tile, the lane's column, the step (mod 16) at which its lines end, and the word address of the line that ends at
	// the current step (advanced by one word per step)
	u32 done_at[kA];
	u64 *line_ptr[kA];
	bool col_ok[kA];
#pragma unroll
	for (int a = 0; a < kA; ++a) {
		u64 const rw = rw0 + (u64) (kA * wave + a);
		col_ok[a] = rw < SW;                                           // (wave-uniform)
		u64 *const col = dst + (rw * 64 + (u64) lane) * dst_pitch;
		done_at[a] = (15u - (u32) ((reinterpret_cast<uintptr_t>(col) >> 3) & 15u)) & 15u;
		line_ptr[a] = col + span_lo - 15;                              // (never dereferenced outside [span_lo, span_hi))
	}

	u64 y[kA][16];
#pragma unroll
	for (int a = 0; a < kA; ++a)
#pragma unroll
		for (int c = 0; c < 16; ++c) y[a][c] = 0;

	// the guarded flavour: the lines that end at step c (cg = the step's column group), four at a time through the slab, every word under
	// the span's bounds.  `c` may be a run-time value: the whole register file goes into the slab, the reader knows which register is which word.
	auto const emit_guarded = [&](int a, u32 c, u32 cg) {
		bool const done = done_at[a] == c;
		u64 const mask = __ballot(done);
		u32 const n = (u32) __builtin_popcountll(mask);                  // (uniform)
		u32 const rank = __builtin_amdgcn_mbcnt_hi((u32) (mask >> 32), __builtin_amdgcn_mbcnt_lo((u32) mask, 0u));
		u64 const tile_col0 = (rw0 + (u64) (kA * wave + a)) * 64;
		for (u32 base = 0; base < n; base += 4) {
			if (done && rank - base < 4u) {
#pragma unroll
				for (int w = 0; w < 16; ++w) slab[wave][rank - base][w] = y[a][w];
				slab_lane[wave][rank - base] = (u32) lane;
			}
			__builtin_amdgcn_wave_barrier();                            // (LDS operations of one wave execute in order)
			u32 const rk = (u32) lane >> 4, w = (u32) lane & 15u;
			if (base + rk < n) {
				u32 const j = (w - c - 1u) & 15u;                        // register w holds word j of the line
				long long const k = (long long) cg - 15 + (long long) j; // column-relative word
				if (k >= (long long) span_lo && k < (long long) span_hi) {
					u64 const r = tile_col0 + slab_lane[wave][rk];
					dst[r * dst_pitch + (u64) k] = slab[wave][rk][w];
				}
			}
			__builtin_amdgcn_wave_barrier();
		}
	};

	lean_butterfly const butterfly(lane);
	u64 stage[kDepth][kPer];
	auto const group_of = [&](u32 st) -> u32 { u32 const cg = span_lo + st; return cg < cg_max ? cg : cg_max; };
#pragma unroll
	for (int j = 0; j < kDepth; ++j) fetch(stage[j], group_of(j));
	stash(0, stage[0]);
	for (u32 i = 0; i < n_span_blocks; ++i) {
#pragma unroll
		for (int c = 0; c < 16; ++c) {
			u32 const cg = span_lo + 16 * i + c;
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (see the ring kernel: the wait __syncthreads() implies can get lost on a loop's back edge)
			__syncthreads();                                           // in[c & 1] is complete; everyone is done with in[(c + 1) & 1]
			fetch(stage[c % kDepth], group_of(16 * i + c + kDepth));
			bool const real = cg < span_hi;                            // (uniform; false only past the matrix's last column group)
			if (real) {
#pragma unroll
				for (int a = 0; a < kA; ++a) y[a][c] = butterfly(in[c & 1][lane][kA * wave + a]);
			}
			stash((c + 1) & 1, stage[(c + 1) % kDepth]);
#pragma unroll
			for (int a = 0; a < kA; ++a) {
				// From the span's second block on, a line that ends at a real step lies inside the span: the fast flavour.  (The first block's
				// lines, the steps past the matrix and the lines still open at the end go through emit_guarded below.)
				if (real && i >= 1 && col_ok[a] && done_at[a] == (u32) c) {
					// the line, from the rotated register file: word j = y[(c + 1 + j) % 16]; every piece a naturally aligned register pair
					u64 *const p = line_ptr[a];
					if (c & 1) {
#pragma unroll
						for (int m = 0; m < 8; ++m) {
							u64 const lo = y[a][(c + 1 + 2 * m) & 15], hi = y[a][(c + 2 + 2 * m) & 15];
							vec4u v;
							v[0] = (u32) lo; v[1] = (u32) (lo >> 32); v[2] = (u32) hi; v[3] = (u32) (hi >> 32);
							*reinterpret_cast<vec4u *>(p + 2 * m) = v;
						}
					} else {
						p[0] = y[a][(c + 1) & 15];
#pragma unroll
						for (int m = 0; m < 7; ++m) {
							u64 const lo = y[a][(c + 2 + 2 * m) & 15], hi = y[a][(c + 3 + 2 * m) & 15];
							vec4u v;
							v[0] = (u32) lo; v[1] = (u32) (lo >> 32); v[2] = (u32) hi; v[3] = (u32) (hi >> 32);
							*reinterpret_cast<vec4u_unaligned8 *>(p + 1 + 2 * m) = v;
						}
						p[15] = y[a][c & 15];
					}
				}
				line_ptr[a] += 1;
			}
			__builtin_amdgcn_sched_barrier(0);                         // (the steps stay apart in the schedule, as in the streaming kernels)
		}
		if (0 == i) {
			// The first block's lines, all at once behind its steps: the words of a line that ended at step c which belong to this span are
			// y[0 .. c], still where they were put (the registers of the block before the span hold nothing of ours: the span before writes them).
			for (u32 c = 0; c < 16 && span_lo + c < span_hi; ++c) {
#pragma unroll
				for (int a = 0; a < kA; ++a)
					if (col_ok[a]) emit_guarded(a, c, span_lo + c);
			}
		}
	}
	// the lines still open at the span's end (and the steps past the matrix's last column group): their words before span_hi
	for (u32 e = 0; e < 15; ++e) {
		u32 const cg = span_hi + e;
#pragma unroll
		for (int a = 0; a < kA; ++a)
			if (col_ok[a]) emit_guarded(a, cg & 15u, cg);
	}
}
#endif   // V2M_TUNING_BUILD


// ---------------------------------------------------------------------------------------------
// The gap-aligned REF row ("template"): what output_sequence() emits for
// chromosome_copy_index == PLOIDY_MAX (sequence_writer.cc:49,70-81): for every node its reference
// segment followed by '-' up to the next node's aligned position.  Built once per uploaded
// graph; every aligned output row equals it outside the spans of its effective ALT edges.
// One thread per 16 output bytes; bytes past L are zero.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void expand_reference_row_kernel(
	char const *__restrict__ ref, u32 const *__restrict__ ref_pos, u32 const *__restrict__ aln_pos,
	u32 n_nodes, u32 L, u64 n_chunks, uint4 *__restrict__ out, char gap)
{
	u64 const c = (u64) blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= n_chunks) return;
	u64 const p0 = c * 16;
	unsigned char bytes[16];
	if (p0 >= L) {
		out[c] = make_uint4(0, 0, 0, 0);
		return;
	}
	// node containing p0: last n with aln_pos[n] <= p0
	u32 lo = 0, hi = n_nodes;   // invariant: aln_pos[lo] <= p0 < aln_pos[hi] (hi == n_nodes means +inf)
	while (hi - lo > 1) {
		u32 const mid = lo + (hi - lo) / 2;
		if (aln_pos[mid] <= p0) lo = mid; else hi = mid;
	}
	u32 n = lo;
	u32 a0 = aln_pos[n], r0 = ref_pos[n];
	u32 a1 = (n + 1 < n_nodes) ? aln_pos[n + 1] : 0xFFFFFFFFu;
	u32 r1 = (n + 1 < n_nodes) ? ref_pos[n + 1] : r0;
#pragma unroll
	for (int b = 0; b < 16; ++b) {
		u64 const p = p0 + b;
		unsigned char v = 0;
		if (p < L) {
			while (p >= a1) {
				++n;
				a0 = a1; r0 = r1;
				a1 = (n + 1 < n_nodes) ? aln_pos[n + 1] : 0xFFFFFFFFu;
				r1 = (n + 1 < n_nodes) ? ref_pos[n + 1] : r0;
			}
			u32 const off = (u32) p - a0;
			v = (off < r1 - r0) ? (unsigned char) ref[r0 + off] : (unsigned char) gap;
		}
		bytes[b] = v;
	}
	uint4 o;
	o.x = bytes[0] | (bytes[1] << 8) | (bytes[2] << 16) | ((u32) bytes[3] << 24);
	o.y = bytes[4] | (bytes[5] << 8) | (bytes[6] << 16) | ((u32) bytes[7] << 24);
	o.z = bytes[8] | (bytes[9] << 8) | (bytes[10] << 16) | ((u32) bytes[11] << 24);
	o.w = bytes[12] | (bytes[13] << 8) | (bytes[14] << 16) | ((u32) bytes[15] << 24);
	out[c] = o;
}


// ---------------------------------------------------------------------------------------------
// Effective-edge resolution: the order-dependent part of the walk.
//
// output_sequence() only honours a set path bit if the walk actually visits the edge's
// source node: following an ALT edge jumps to its target and silently skips the set bits of
// every node in between, and among several set edges of one node the lowest index wins
// (sequence_writer.cc:51-67).  Scanning the row's set bits in edge order with the current
// node `cur` (initially 0): edge e is effective iff src[e] >= cur, and then cur = tgt[e]
// (hard part 2).
//
// What makes this parallel is a property of the GRAPH, not of the row: an edge whose source
// node is at or after the target of every lower-numbered edge (src[e] >= max tgt[0..e)) can
// never be skipped, whatever bits a row has, because cur is always one of those targets.
// Only the other edges -- the "overlappable" ones, lying inside the span of an earlier edge:
// variants under a deletion, second and later ALTs of a multi-allelic site -- need the scan,
// and for them the scan can restart, with cur = 0, at the nearest earlier edge that is NOT
// overlappable (set or not: every earlier target is <= its source).  Both the mask and hence the
// restart points are properties of the uploaded graph, computed once.
//
// resolve_effective_edges_kernel: one thread per (row, 64-edge word), fully coalesced (grid.y = row).  A word
// with no set overlappable bit is copied through.  Otherwise the thread finds its restart
// point in the static mask and replays the row's set bits from there to the end of its own
// word.  A restart point more than kMaxBackWords away
// (graphs with chromosome-scale deletions) flags the row for resolve_rows_serial_kernel, the
// one-wave-per-row scan that carries cur across the whole row.
//
// Founder rows assemble their bit column from several chromosome copies, one per cut segment
// (founder_sequence_greedy_output.cc:106-114).
// ---------------------------------------------------------------------------------------------
struct row_segments {
	// segment k of a row covers edges [edge_begin[k], edge_begin[k+1]) and reads copy[k];
	// the last segment extends to the end.  Rows without cuts have exactly one segment.
	u32 const *seg_offsets;     // [n_rows + 1]
	u32 const *seg_edge_begin;  // [total segments]
	u32 const *seg_copy;        // [total segments]  (0xFFFFFFFF = follow REF)
	// bit columns of the rows that have more than one segment, put together by assemble_row_bits_kernel
	u64 const *assembled;       // [n_rows x assembled_words]; not read for single-segment rows
	u32 assembled_words;
};

__device__ __forceinline__ u64 load_row_word(
	u64 const *__restrict__ paths, u64 words_per_copy, row_segments const &rs, u32 row, u32 s_begin, u32 s_end, u32 wi)
{
	if (s_end - s_begin == 1) {
		u32 const copy = rs.seg_copy[s_begin];
		return (copy == 0xFFFFFFFFu) ? 0 : paths[(u64) copy * words_per_copy + wi];
	}
	return rs.assembled[(u64) row * rs.assembled_words + wi];
}

// Founder rows switch chromosome copy at every cut node -- hundreds of thousands of segments per row at 1KG scale,
// dozens per 64-edge word.  One wave puts 64 words of one row together: the segments that touch them are a
// contiguous piece of the row's (sorted) segment table, found once; the lanes then take one segment each, so the
// table is read coalesced and all the scattered path-word loads of the piece are in flight together.
__global__ __launch_bounds__(256) void assemble_row_bits_kernel(
	u64 const *__restrict__ paths, u64 words_per_copy, row_segments rs, u64 *__restrict__ assembled, u32 n_words, u32 row_base)
{
	__shared__ unsigned long long acc[256];
	u32 const row = blockIdx.y + row_base;
	u32 const s_begin = rs.seg_offsets[row], s_end = rs.seg_offsets[row + 1];
	if (s_end - s_begin <= 1) return;                        // whole workgroup: the row is read straight from its copy
	u32 const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	u32 const w0 = (blockIdx.x * 4 + wave) * 64;              // this wave's first word
	acc[threadIdx.x] = 0;
	__syncthreads();
	if (w0 < n_words) {
		u64 const e_lo = (u64) w0 * 64;
		u64 const e_hi = (u64) (w0 + 64 < n_words ? w0 + 64 : n_words) * 64;
		// last segment whose first edge is <= e_lo (the first segment of a row starts at edge 0)
		u32 lo = s_begin, hi = s_end;
		while (hi - lo > 1) {
			u32 const mid = lo + (hi - lo) / 2;
			if (rs.seg_edge_begin[mid] <= e_lo) lo = mid; else hi = mid;
		}
		for (u32 s = lo + lane; s < s_end; s += 64) {
			u64 b = rs.seg_edge_begin[s];
			if (b >= e_hi) break;
			u64 e = (s + 1 < s_end) ? rs.seg_edge_begin[s + 1] : ~0ULL;
			u32 const copy = rs.seg_copy[s];
			if (copy == 0xFFFFFFFFu) continue;
			b = b > e_lo ? b : e_lo;
			e = e < e_hi ? e : e_hi;
			if (b >= e) continue;
			u64 const *const column = paths + (u64) copy * words_per_copy;
			for (u64 wi = b >> 6; wi <= (e - 1) >> 6; ++wi) {
				u64 const base = wi * 64;
				u32 const from = b > base ? (u32) (b - base) : 0;   // first bit of this word in the segment
				u32 const to = e < base + 64 ? (u32) (e - base) : 64;   // one past the last
				u64 const mask = (to >= 64 ? ~0ULL : ((1ULL << to) - 1)) & ~((1ULL << from) - 1);
				u64 const v = column[wi] & mask;
				if (v) atomicOr(&acc[wave * 64 + (u32) (wi - w0)], (unsigned long long) v);
			}
		}
	}
	__syncthreads();
	if (w0 + lane < n_words) assembled[(u64) row * rs.assembled_words + w0 + lane] = acc[threadIdx.x];
}

constexpr int kResolveWordsPerThread = 8;
constexpr u32 kResolveQueueShards = 1024;
constexpr u32 kMaxBackWords = 2048;   // default max_back_words: restart points > 131072 edges back go to the serial kernel

// The exact rule for one (row, word): restart at the nearest earlier edge that is NOT overlappable (set or not: when the
// walk reaches such an edge every earlier target is <= its source node, so the scan state there is equivalent to
// cur = 0; the search runs over the graph-static mask only) and replay the row's set edges from there to the end of the word.
__device__ __forceinline__ u64 resolve_word_exact(
	u64 const *__restrict__ paths, u64 words_per_copy, row_segments const &rs, u32 row, u32 s_begin, u32 s_end,
	edge_span const *__restrict__ spans, u64 const *__restrict__ overlappable,
	u32 wi, u64 w, u64 ovl_w, u32 *__restrict__ needs_serial, u32 max_back_words)
{
	u64 const ov = w & ovl_w;
	if (0 == ov) return w;
	int const b0 = __builtin_ctzll(ov);
	u32 sw = wi;
	int sb = 0;
	u64 fixed = ~ovl_w & ((1ULL << b0) - 1);
	if (fixed) {
		sb = 63 - __builtin_clzll(fixed);
	} else {
		u32 steps = 0;
		for (;;) {
			if (0 == sw) { sb = 0; break; }              // edge 0 is never overlappable; defensive
			--sw;
			if (++steps > max_back_words) { atomicOr(&needs_serial[row], 1u); return w; }   // the row goes to resolve_rows_serial_kernel
			fixed = ~overlappable[sw];
			if (fixed) { sb = 63 - __builtin_clzll(fixed); break; }
		}
	}
	u64 out = (sw == wi) ? (w & ((1ULL << sb) - 1)) : 0;   // set bits before the restart point in this word are certain
	u32 cur = 0;
	for (u32 ww = sw; ww <= wi; ++ww) {
		u64 x = (ww == wi) ? w : load_row_word(paths, words_per_copy, rs, row, s_begin, s_end, ww);
		if (ww == sw) x &= ~((1ULL << sb) - 1);
		// the rule is sequential, the loads it needs are not: the spans of the next (up to) 8 set edges are fetched together
		while (x) {
			constexpr int kBatch = 8;
			edge_span sp[kBatch];
			int bit[kBatch];
			bool has[kBatch];
#pragma unroll
			for (int k = 0; k < kBatch; ++k) {
				has[k] = 0 != x;
				bit[k] = has[k] ? __builtin_ctzll(x) : 0;
				x &= x - 1;                                        // stays 0 once it is 0
				sp[k] = spans[ww * 64u + bit[k]];                  // always a valid edge of this word (its first one when the batch has run out)
			}
#pragma unroll
			for (int k = 0; k < kBatch; ++k) {
				if (has[k] && sp[k].src >= cur) {
					cur = sp[k].tgt;
					if (ww == wi) out |= 1ULL << bit[k];
				}
			}
		}
	}
	return out;
}

// Pass 1, streaming.  grid: x over rows, y over pieces of kResolveWordsPerThread x 256 consecutive words of a row (the
// row, and with it the segment table lookups, is uniform per workgroup).  Every load of the common path is issued before
// any result is looked at, in two rounds: the row's words with the static masks and ranks, then the blocker masks of
// each word's first two set overlappable edges.  A word is decided here when it has no set overlappable edge, or at most
// two and none of their possible blockers is set (then all its set edges are effective).  Everything else -- a possible
// blocker is set, or three and more overlappable edges are -- needs dependent loads, and ONE lane that takes them holds
// its whole wave up for microseconds: on config 5's graph nearly every wave has such a lane, which is what this kernel's
// time was made of (240 us per 251-row batch whether it handled 1 or 8 words per thread, replayed or looked masks up,
// against 70 us for a plain copy of the same words).  Those words are appended to a queue instead and decided by
// resolve_queued_words_kernel, where 64 of them share a wave.  The queue is cut into kResolveQueueShards segments with a
// counter each, a workgroup appends to the segment its index picks, one atomic per wave: a single counter would take
// every wave's atomic in turn (~88 per microsecond) and cost more than the kernel.
__global__ __launch_bounds__(256) void resolve_effective_edges_kernel(
	u64 const *__restrict__ paths, u64 words_per_copy, u32 n_edges,
	row_segments rs, edge_span const *__restrict__ spans, u64 const *__restrict__ overlappable,
	u32 const *__restrict__ ovl_rank, u64 const *__restrict__ blocker_masks,
	u64 *__restrict__ eff, u32 n_words, u32 eff_words_per_row, u32 row_base, u32 piece_base,
	u32 *__restrict__ queue, u32 *__restrict__ queue_counts /* [kResolveQueueShards] */, u32 shard_capacity, u32 *__restrict__ needs_serial, u32 max_back_words)
{
	u32 const row = blockIdx.x + row_base;
	u32 const shard = (blockIdx.y * gridDim.x + blockIdx.x) % kResolveQueueShards;
	u32 const s_begin = rs.seg_offsets[row], s_end = rs.seg_offsets[row + 1];
	u64 const tail_mask = (n_edges & 63) ? (1ULL << (n_edges & 63)) - 1 : ~0ULL;   // padding bits are zero by contract; do not trust them
	// where the row's bits come from: one chromosome copy, nothing (REF row), or the assembled row
	u64 const *row_words = nullptr;
	if (s_end - s_begin == 1) {
		u32 const copy = rs.seg_copy[s_begin];
		if (copy != 0xFFFFFFFFu) row_words = paths + (u64) copy * words_per_copy;
	} else {
		row_words = rs.assembled + (u64) row * rs.assembled_words;
	}
	u32 const w_first = (blockIdx.y + piece_base) * kResolveWordsPerThread * blockDim.x + threadIdx.x;
	u64 w_all[kResolveWordsPerThread], ovl_all[kResolveWordsPerThread], mask_a[kResolveWordsPerThread], mask_b[kResolveWordsPerThread];
	u32 rank_all[kResolveWordsPerThread];
#pragma unroll
	for (int piece = 0; piece < kResolveWordsPerThread; ++piece) {
		u32 const wi = w_first + piece * blockDim.x;
		u32 const wc = wi < n_words ? wi : n_words - 1;     // clamped: every load is issued, none sits under a branch
		w_all[piece] = row_words ? row_words[wc] : 0;
		ovl_all[piece] = overlappable[wc];
		rank_all[piece] = ovl_rank[wc];
		if (wi == n_words - 1) w_all[piece] &= tail_mask;
	}
#pragma unroll
	for (int piece = 0; piece < kResolveWordsPerThread; ++piece) {
		u64 const ov = w_all[piece] & ovl_all[piece], ov2 = ov & (ov - 1);
		auto const entry = [&](u64 m) {                     // the table entry of the lowest set bit of m (entry 0 when m is empty)
			return m ? rank_all[piece] + (u32) __builtin_popcountll(ovl_all[piece] & ((1ULL << __builtin_ctzll(m)) - 1)) : 0u;
		};
		mask_a[piece] = blocker_masks[entry(ov)];
		mask_b[piece] = blocker_masks[entry(ov2)];
	}
	int const lane = threadIdx.x & 63;
#pragma unroll
	for (int piece = 0; piece < kResolveWordsPerThread; ++piece) {
		u32 const wi = w_first + piece * blockDim.x;
		bool const in_range = wi < n_words;
		u64 const w = w_all[piece], ov = w & ovl_all[piece], ov2 = ov & (ov - 1);
		bool const hard = in_range && ov && ((w & mask_a[piece]) || (ov2 && ((w & mask_b[piece]) || (ov2 & (ov2 - 1)))));
		if (in_range && !hard) eff[(u64) row * eff_words_per_row + wi] = w;
		u64 const hard_lanes = __ballot(hard);
		if (hard_lanes) {                                    // wave-uniform
			u32 base = 0;
			if (0 == lane) base = atomicAdd(&queue_counts[shard], (u32) __builtin_popcountll(hard_lanes));
			base = __shfl(base, 0, kWave);
			if (hard) {
				u32 const slot = base + __builtin_amdgcn_mbcnt_hi((u32) (hard_lanes >> 32), __builtin_amdgcn_mbcnt_lo((u32) hard_lanes, 0));
				if (slot < shard_capacity) queue[(u64) shard * shard_capacity + slot] = (row - row_base) * n_words + wi;
				else eff[(u64) row * eff_words_per_row + wi] = resolve_word_exact(paths, words_per_copy, rs, row, s_begin, s_end, spans, overlappable, wi, w, ovl_all[piece], needs_serial, max_back_words);
			}
		}
	}
}

// Pass 2, dense: one thread per queued (row, word); workgroup b works on segment b % kResolveQueueShards.
__global__ __launch_bounds__(256) void resolve_queued_words_kernel(
	u64 const *__restrict__ paths, u64 words_per_copy, u32 n_edges,
	row_segments rs, edge_span const *__restrict__ spans, u64 const *__restrict__ overlappable,
	u64 *__restrict__ eff, u32 n_words, u32 eff_words_per_row, u32 row_base,
	u32 const *__restrict__ queue, u32 const *__restrict__ queue_counts, u32 shard_capacity, u32 *__restrict__ needs_serial, u32 max_back_words)
{
	u32 const shard = blockIdx.x % kResolveQueueShards, part = blockIdx.x / kResolveQueueShards, parts = gridDim.x / kResolveQueueShards;
	u32 const n = queue_counts[shard] < shard_capacity ? queue_counts[shard] : shard_capacity;
	u64 const tail_mask = (n_edges & 63) ? (1ULL << (n_edges & 63)) - 1 : ~0ULL;
	for (u32 i = part * blockDim.x + threadIdx.x; i < n; i += parts * blockDim.x) {
		u32 const entry = queue[(u64) shard * shard_capacity + i];
		u32 const row = row_base + entry / n_words, wi = entry % n_words;
		u32 const s_begin = rs.seg_offsets[row], s_end = rs.seg_offsets[row + 1];
		u64 w = load_row_word(paths, words_per_copy, rs, row, s_begin, s_end, wi);
		if (wi == n_words - 1) w &= tail_mask;
		eff[(u64) row * eff_words_per_row + wi] = resolve_word_exact(paths, words_per_copy, rs, row, s_begin, s_end, spans, overlappable, wi, w, overlappable[wi], needs_serial, max_back_words);
	}
}


// One wave per flagged row: the row-long scan that carries cur across 4096-edge chunks.
__global__ __launch_bounds__(256) void resolve_rows_serial_kernel(
	u64 const *__restrict__ paths, u64 words_per_copy, u32 n_edges,
	row_segments rs, edge_span const *__restrict__ spans,
	u64 *__restrict__ eff, u64 eff_words_per_row, u32 n_rows, u32 const *__restrict__ needs_serial)
{
	int const lane = threadIdx.x & 63;
	u32 const row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	if (row >= n_rows) return;   // whole wave exits together
	if (needs_serial && 0 == needs_serial[row]) return;

	u32 const s_begin = rs.seg_offsets[row], s_end = rs.seg_offsets[row + 1];
	u32 const n_words = (n_edges + 63) / 64;
	u64 *const eff_row = eff + (u64) row * eff_words_per_row;
	u32 cur = 0;   // current node of the walk at the start of this chunk (wave-uniform)

	for (u32 base = 0; base < n_words; base += 64) {
		u32 const wi = base + lane;
		u64 w = 0;
		if (wi < n_words) {
			w = load_row_word(paths, words_per_copy, rs, row, s_begin, s_end, wi);
			if (wi == n_words - 1 && (n_edges & 63))
				w &= (1ULL << (n_edges & 63)) - 1;       // padding bits are zero by contract; do not trust them
		}

		// pass 1: maximum target among this lane's set edges
		u32 lane_max = 0;
		for (u64 m = w; m; m &= m - 1) {
			u32 const e = wi * 64u + __builtin_ctzll(m);
			u32 const tgt = spans[e].tgt;
			lane_max = tgt > lane_max ? tgt : lane_max;
		}
		// exclusive prefix maximum over lanes
		u32 incl = lane_max;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			u32 const o = __shfl_up(incl, d, kWave);
			if (lane >= d) incl = o > incl ? o : incl;
		}
		u32 excl = __shfl_up(incl, 1, kWave);
		if (lane == 0) excl = 0;
		u32 const chunk_max = __shfl(incl, 63, kWave);

		// pass 2: every set edge must start at or after everything that precedes it
		u32 run = excl > cur ? excl : cur;
		bool uncertain = false;
		for (u64 m = w; m; m &= m - 1) {
			u32 const e = wi * 64u + __builtin_ctzll(m);
			edge_span const sp = spans[e];
			uncertain |= sp.src < run;
			run = sp.tgt > run ? sp.tgt : run;
		}

		u64 out = w;
		if (__any(uncertain)) {
			// replay the chunk serially, lane by lane, carrying the true current node
			out = 0;
			u32 c = cur;
			u64 busy = __ballot(w != 0);
			while (busy) {
				int const j = __builtin_ctzll(busy);
				busy &= busy - 1;
				if (lane == j) {
					for (u64 m = w; m; m &= m - 1) {
						int const b = __builtin_ctzll(m);
						edge_span const sp = spans[wi * 64u + b];
						if (sp.src >= c) {
							out |= 1ULL << b;
							c = sp.tgt;
						}
					}
				}
				c = __shfl(c, j, kWave);
			}
			cur = c;
		} else {
			cur = chunk_max > cur ? chunk_max : cur;
		}
		if (wi < eff_words_per_row)
			eff_row[wi] = out;
	}
	// words between n_words and eff_words_per_row (when the scratch row is wider) stay untouched:
	// the splice kernel never reads bits >= n_edges.
}


// ---------------------------------------------------------------------------------------------
// Aligned splice: out[row] = template, patched over the aligned span of every effective edge
// with the edge's label followed by '-' (sequence_writer.cc:57-65,80-83).
//
// Workgroup = one 16-KiB tile of aligned positions x a group of rows.  The tile of the
// template is fetched once into registers (4 x 16 B per thread) and reused for every row of
// the group; per row it is dropped into LDS, the few effective edges that touch the tile
// overwrite their spans byte-wise, and the tile streams out with 16-B/lane coalesced stores.
// Two LDS buffers alternate so that one barrier pair per row suffices.
//
// Which edges can touch tile t: those whose span begins inside it -- a contiguous index range,
// because edges are ordered by source node (variant_graph.cc:81,101) -- plus the edges that
// begin earlier and reach into it (an explicit per-tile list; at most one of them can be
// effective for any given row).
// ---------------------------------------------------------------------------------------------
struct tile_tables {
	u32 const *edge_begin;     // [n_tiles + 1] first edge with aln_begin >= t * kTileBytes
	u32 const *cross_offsets;  // [n_tiles + 1] CSR into cross_edges
	u32 const *cross_edges;    // edges with aln_begin < t*kTileBytes < aln_end, ascending
};

// Bytes [from, to) of an edge's aligned span: label bytes first, then padding.
template <typename LabelPtr>
__device__ __forceinline__ void fill_patch_bytes(
	unsigned char *tile, u32 tile_base, u32 aln_begin, u32 label_len, LabelPtr label,
	u32 from, u32 to, u32 start, u32 step, char gap)
{
	for (u32 pos = from + start; pos < to; pos += step) {
		u32 const off = pos - aln_begin;
		tile[pos - tile_base] = (off < label_len) ? (unsigned char) label[off] : (unsigned char) gap;
	}
}

// Per-workgroup cache of everything the row loop needs about the tile's candidate edges, so that the loop itself
// touches only LDS: the patch descriptors and label bytes of the edges that begin in the tile (a contiguous
// edge range, hence a contiguous slice of `patches` and of the label pool) and the effective-edge words of every
// row of the group for that range.  Without it each row pays three dependent L2 round trips (effective bits ->
// descriptor -> label bytes) between its two barriers, which is what bounds the kernel on dense graphs
// (config 5: one edge per 40 bp).  Edges that cross into the tile from the left, candidates beyond the cache
// and spans longer than kLongPatch take the global-memory path.
constexpr int kCandLds = 512;
constexpr int kLabelLds = 1024;
constexpr int kGroupRowsLds = 16;
constexpr int kEffWordsLds = kCandLds / 64 + 2;
constexpr int kLongQueueLds = 16;

// 8-byte form of an edge_patch relative to the tile; 0xFFFF in span or label_rel means "not representable here,
// read the full descriptor from global memory" (span >= 64 KiB, or label bytes outside the cached slice).
struct cached_patch {
	unsigned short begin_rel;   // aln_begin - tile_base (< kTileBytes: the edge begins in this tile)
	unsigned short span;        // aln_end - aln_begin
	unsigned short label_rel;   // label_begin - label_base
	unsigned short label_len;
};

struct patch_cache {
	cached_patch patch[kCandLds];
	u64 eff[kGroupRowsLds][kEffWordsLds];
	unsigned char labels[kLabelLds];
	u32 long_queue[kLongQueueLds];
	u32 long_count;
};

struct tile_job {
	u32 tile_base, cross_begin, n_cross, range_begin, n_range;
	u32 n_lds;          // candidates of the range cached in LDS
	u32 w0;             // first effective-bit word cached
};

// The effective-edge words of up to kGroupRowsLds rows, first_row onwards, for the tile's cached edge range.
__device__ __forceinline__ void load_eff_cache(
	patch_cache &pc, tile_job const &job, u64 const *__restrict__ eff, u64 eff_words_per_row, u32 first_row, u32 n_rows, int t)
{
	if (0 == job.n_lds) return;
	u32 const nw = ((job.range_begin + job.n_lds - 1) >> 6) - job.w0 + 1;   // <= kEffWordsLds
	u32 const rows = n_rows < (u32) kGroupRowsLds ? n_rows : (u32) kGroupRowsLds;
	for (u32 i = t; i < rows * nw; i += kSpliceThreads)
		pc.eff[i / nw][i % nw] = eff[(u64) (first_row + i / nw) * eff_words_per_row + job.w0 + i % nw];
}

__device__ __forceinline__ void load_patch_cache(
	patch_cache &pc, tile_job &job, tile_tables const &tt, edge_patch const *__restrict__ patches, char const *__restrict__ labels,
	u64 const *__restrict__ eff, u64 eff_words_per_row, u32 tile, u32 row_begin, u32 n_group_rows, int t)
{
	job.tile_base = tile * (u32) kTileBytes;
	job.cross_begin = tt.cross_offsets[tile];
	job.n_cross = tt.cross_offsets[tile + 1] - job.cross_begin;
	job.range_begin = tt.edge_begin[tile];
	job.n_range = tt.edge_begin[tile + 1] - job.range_begin;
	job.n_lds = job.n_range < (u32) kCandLds ? job.n_range : (u32) kCandLds;
	job.w0 = job.range_begin >> 6;
	if (job.n_lds) {
		u32 const label_base = patches[job.range_begin].label_begin;
		for (u32 i = t; i < job.n_lds; i += kSpliceThreads) {
			edge_patch const p = patches[job.range_begin + i];
			u32 const span = p.aln_end - p.aln_begin, rel = p.label_begin - label_base;
			cached_patch c;
			c.begin_rel = (unsigned short) (p.aln_begin - job.tile_base);
			c.span = span < 0xFFFFu ? (unsigned short) span : (unsigned short) 0xFFFF;
			bool const label_ok = rel + p.label_len <= (u32) kLabelLds;
			c.label_rel = label_ok ? (unsigned short) rel : (unsigned short) 0xFFFF;
			c.label_len = label_ok ? (unsigned short) p.label_len : (unsigned short) 0;
			pc.patch[i] = c;
		}
		edge_patch const last = patches[job.range_begin + job.n_lds - 1];
		u32 const label_span = last.label_begin + last.label_len - label_base;
		u32 const label_len = label_span < (u32) kLabelLds ? label_span : (u32) kLabelLds;
		for (u32 i = t; i < label_len; i += kSpliceThreads)
			pc.labels[i] = (unsigned char) labels[label_base + i];
	}
	load_eff_cache(pc, job, eff, eff_words_per_row, row_begin, n_group_rows, t);
	if (t == 0) pc.long_count = 0;
	// visibility: the caller's next __syncthreads() (after it has dropped the pristine tile into LDS)
}

// What an effective candidate edge contributes to the tile: its clipped aligned span and where its label bytes are.
struct tile_patch {
	u32 edge;
	u32 aln_begin;     // unclipped start of the span
	u32 from, to;      // span clipped to the tile
	u32 label_len;
	u32 label_begin;   // offset into the cached label slice (in_lds) or into the global label pool
	bool in_lds;
};

// Calls f(tile_patch) for every candidate edge of the tile that is effective in the row, each candidate on one thread.
// Candidate i of the tile (crossing edges first, then the range that begins in it): its edge index and whether the
// cached LDS copies (patch descriptor, effective-edge words) cover it.
__device__ __forceinline__ u32 candidate_edge(tile_job const &job, tile_tables const &tt, u32 i)
{
	return (i < job.n_cross) ? tt.cross_edges[job.cross_begin + i] : job.range_begin + (i - job.n_cross);
}

__device__ __forceinline__ bool candidate_in_lds(tile_job const &job, u32 i)
{
	return i >= job.n_cross && i - job.n_cross < job.n_lds;
}

// The part of candidate i's span that falls into the tile, from the cached descriptor when it is representable there.
__device__ __forceinline__ tile_patch candidate_patch(
	patch_cache const &pc, tile_job const &job, edge_patch const *__restrict__ patches, u32 i, u32 e, bool cached)
{
	u32 const tile_end = job.tile_base + kTileBytes;
	cached_patch c{};
	if (cached) {
		c = pc.patch[i - job.n_cross];
		cached = c.span != 0xFFFF && c.label_rel != 0xFFFF;
	}
	tile_patch tp;
	tp.edge = e;
	if (cached) {
		tp.aln_begin = tp.from = job.tile_base + c.begin_rel;
		tp.to = tp.from + c.span < tile_end ? tp.from + c.span : tile_end;
		tp.label_len = c.label_len;
		tp.label_begin = c.label_rel;
		tp.in_lds = true;
	} else {
		edge_patch const p = patches[e];
		tp.aln_begin = p.aln_begin;
		tp.from = p.aln_begin > job.tile_base ? p.aln_begin : job.tile_base;
		tp.to = p.aln_end < tile_end ? p.aln_end : tile_end;
		tp.label_len = p.label_len;
		tp.label_begin = p.label_begin;
		tp.in_lds = false;
	}
	return tp;
}

// Whether candidate i is effective for the row (local_row = its index within the rows whose words are cached).
__device__ __forceinline__ bool candidate_effective(
	patch_cache const &pc, tile_job const &job, tile_tables const &tt, u64 const *__restrict__ eff_row, u32 local_row, u32 i,
	u32 &e, bool &cached)
{
	cached = candidate_in_lds(job, i) && local_row < (u32) kGroupRowsLds;
	if (cached) {
		e = job.range_begin + (i - job.n_cross);
		return (pc.eff[local_row][(e >> 6) - job.w0] >> (e & 63)) & 1;
	}
	e = candidate_edge(job, tt, i);
	return (eff_row[e >> 6] >> (e & 63)) & 1;
}

template <typename F>
__device__ __forceinline__ void for_each_effective_candidate(
	patch_cache const &pc, tile_job const &job, tile_tables const &tt, edge_patch const *__restrict__ patches,
	u64 const *__restrict__ eff_row, u32 local_row, int t, F &&f)
{
	u32 const n_cand = job.n_cross + job.n_range;
	for (u32 i = t; i < n_cand; i += kSpliceThreads) {
		u32 e;
		bool cached;
		if (!candidate_effective(pc, job, tt, eff_row, local_row, i, e, cached)) continue;
		f(candidate_patch(pc, job, patches, i, e, cached));
	}
}

// Overwrites, in the LDS tile `buf`, the spans of the row's effective edges.  Contains the barriers that separate
// it from the tile's readers; must be called by all threads of the workgroup.
__device__ __forceinline__ void patch_row_tile(
	unsigned char *buf, patch_cache &pc, tile_job const &job, tile_tables const &tt,
	edge_patch const *__restrict__ patches, char const *__restrict__ labels,
	u64 const *__restrict__ eff_row, u32 local_row, int t, char gap)
{
	u32 const tile_end = job.tile_base + kTileBytes;
	for_each_effective_candidate(pc, job, tt, patches, eff_row, local_row, t, [&](tile_patch const &tp) {
		if (tp.to - tp.from > kLongPatch) {
			u32 const slot = atomicAdd(&pc.long_count, 1u);
			if (slot < (u32) kLongQueueLds) { pc.long_queue[slot] = tp.edge; return; }
		}
		if (tp.in_lds)
			fill_patch_bytes(buf, job.tile_base, tp.aln_begin, tp.label_len, pc.labels + tp.label_begin, tp.from, tp.to, 0, 1, gap);
		else
			fill_patch_bytes(buf, job.tile_base, tp.aln_begin, tp.label_len, labels + tp.label_begin, tp.from, tp.to, 0, 1, gap);
	});
	__syncthreads();

	u32 const n_long = pc.long_count < (u32) kLongQueueLds ? pc.long_count : (u32) kLongQueueLds;   // workgroup-uniform
	if (n_long) {
		// long spans (big deletions, long insertions): one wave per span, 64 bytes per step
		for (u32 q = t >> 6; q < n_long; q += kSpliceThreads >> 6) {
			edge_patch const p = patches[pc.long_queue[q]];
			u32 const from = p.aln_begin > job.tile_base ? p.aln_begin : job.tile_base;
			u32 const to = p.aln_end < tile_end ? p.aln_end : tile_end;
			fill_patch_bytes(buf, job.tile_base, p.aln_begin, p.label_len, labels + p.label_begin, from, to, t & 63, 64, gap);
		}
		__syncthreads();
		if (t == 0) pc.long_count = 0;   // next read is after the next row's first barrier
	}
}

// Workgroup -> (tile, row group).  The grid is cut into super-blocks of `tile_run` consecutive tiles x all row
// groups; inside one, consecutive workgroups take consecutive TILES of the same row group.  The ~1000
// workgroups in flight therefore write a few long contiguous runs per row (what a memset looks like to the
// TLB and to DRAM pages) instead of 16 KiB islands 100 MB apart, while a super-block's template tiles
// (tile_run x 16 KiB) stay L2-resident for the row groups that follow.  Workgroups are handed to the 8 XCDs round-robin
// and tile_run is a multiple of 8, so (tile, any group) always lands on XCD (tile - t0) % 8: each XCD's L2 serves one
// eighth of the super-block's tiles to all row groups.
__device__ __forceinline__ void map_block(u32 b, u32 n_groups, u32 n_tiles, u32 tile_run, u32 &tile, u32 &group)
{
	u32 const per_super = tile_run * n_groups;
	u32 const super = b / per_super, within = b % per_super;
	u32 const t0 = super * tile_run;
	u32 const run = (n_tiles - t0 < tile_run) ? n_tiles - t0 : tile_run;
	group = within / run;
	tile = t0 + within % run;
}

template <bool kNonTemporal>
__global__ __launch_bounds__(kSpliceThreads) void splice_aligned_kernel(
	vec4u const *__restrict__ tmpl, u64 const *__restrict__ eff, u64 eff_words_per_row,
	tile_tables tt, edge_patch const *__restrict__ patches, char const *__restrict__ labels,
	char *__restrict__ out, u64 row_pitch, u32 n_rows, u32 rows_per_group, u32 n_groups, u32 n_tiles, u32 tile_run,
	u64 store_limit /* aligned length rounded up to 16 */, char gap)
{
	__shared__ vec4u lds[2][kTileChunks];
	__shared__ patch_cache pc;

	int const t = threadIdx.x;
	u32 tile, group;
	map_block(blockIdx.x, n_groups, n_tiles, tile_run, tile, group);
	u32 const row_begin = group * rows_per_group;
	u32 const row_end = (row_begin + rows_per_group < n_rows) ? row_begin + rows_per_group : n_rows;

	vec4u pristine[kChunksPerThread];
#pragma unroll
	for (int k = 0; k < kChunksPerThread; ++k)
		pristine[k] = tmpl[(u64) tile * kTileChunks + t + kSpliceThreads * k];

	tile_job job;
	load_patch_cache(pc, job, tt, patches, labels, eff, eff_words_per_row, tile, row_begin, row_end - row_begin, t);
	u32 const tile_base = job.tile_base;

	for (u32 row = row_begin; row < row_end; ++row) {
		vec4u *const buf = lds[(row - row_begin) & 1];
		// A group may hold more rows than the cache of effective-edge words (kGroupRowsLds): the template tile and the patch cache are set up once per
		// group, the words are reloaded every kGroupRowsLds rows.  (No barrier before the reload: every read of the previous rows' words lies before
		// the barrier inside their patch_row_tile(), which every thread has passed; the barrier below orders the reload before this row's reads.)
		u32 const cached_row = (row - row_begin) % (u32) kGroupRowsLds;
		if (row != row_begin && 0 == cached_row)
			load_eff_cache(pc, job, eff, eff_words_per_row, row, row_end - row, t);
#pragma unroll
		for (int k = 0; k < kChunksPerThread; ++k)
			buf[t + kSpliceThreads * k] = pristine[k];
		__syncthreads();

		patch_row_tile((unsigned char *) buf, pc, job, tt, patches, labels, eff + (u64) row * eff_words_per_row, cached_row, t, gap);

		char *const dst = out + (u64) row * row_pitch + (u64) tile * kTileBytes;
#pragma unroll
		for (int k = 0; k < kChunksPerThread; ++k) {
			int const c = t + kSpliceThreads * k;
			if ((u64) tile_base + (u64) c * 16 < store_limit) {
				vec4u const nv = buf[c];
				if (kNonTemporal)
					__builtin_nontemporal_store(nv, (vec4u *) (dst + c * 16));
				else
					*(vec4u *) (dst + c * 16) = nv;
			}
		}
	}
}


// ---------------------------------------------------------------------------------------------
// Unaligned rows (should_output_unaligned, sequence_writer.cc:80: the '-' padding is not written).
//
// An unaligned row is the aligned row with the padding removed, so the same tiles are built in LDS
// -- from a second template whose padding byte is 0 instead of '-' (a byte neither FASTA text nor
// VCF alleles can contain), patched with 0 as padding -- and then compacted.  Where a tile's bytes
// land in the row needs the number of non-padding bytes of all tiles before it:
//   pass 1  count_unaligned_kernel     non-padding bytes of every (row, tile), WITHOUT building the rows: the
//                                      template tile's own count plus, per effective edge, the label bytes it
//                                      puts into the tile minus the template bytes its span removes
//           scan_tile_counts_kernel    exclusive prefix sum per row -> tile offsets, row lengths
//   pass 2  splice_unaligned_kernel    builds each tile and streams it out chunk by chunk: a 16-B chunk without padding
//                                      goes out as one 16-B store at its (byte-granular) destination, a chunk that
//                                      contains padding is packed in registers first (see the kernel).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 zero_bytes_mask(u32 x)
{
	// 0x80 in every byte of x that is zero (exact, no borrow artefacts)
	return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);
}

// Inclusive prefix sum across the wave in six DPP adds (four shifts inside each row of 16 lanes, then lane 15 of rows 0 / 2
// added into rows 1 / 3 and lane 31 into rows 2 and 3): no LDS round trips, unlike the __shfl_up form (ds_bpermute).
__device__ __forceinline__ u32 wave_inclusive_scan_u32(u32 v)
{
	v += (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x111, 0xf, 0xf, false);   // row_shr:1, zeros shifted in
	v += (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x112, 0xf, 0xf, false);   // row_shr:2
	v += (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x114, 0xf, 0xf, false);   // row_shr:4
	v += (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x118, 0xf, 0xf, false);   // row_shr:8
	v += (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
	v += (u32) __builtin_amdgcn_update_dpp(0, (int) v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
	return v;
}

// Four independent scans in lockstep, round by round: a DPP instruction that reads the register the instruction before it wrote costs two
// wait states (hipcc fills them with s_nop 1), and written one scan after the other -- as the source had them in rounds 2-5 -- the four scans
// came out as 24 dependent adds with 28 s_nops between them; side by side each round's four adds cover one another's wait states.
template <int kN>
__device__ __forceinline__ void wave_inclusive_scan_u32_lockstep(u32 (&v)[kN])
{
#define V2M_SCAN_ROUND(CTRL, ROWS) \
	_Pragma("unroll") for (int k = 0; k < kN; ++k) v[k] += (u32) __builtin_amdgcn_update_dpp(0, (int) v[k], CTRL, ROWS, 0xf, false);
	V2M_SCAN_ROUND(0x111, 0xf)   // row_shr:1, zeros shifted in
	V2M_SCAN_ROUND(0x112, 0xf)   // row_shr:2
	V2M_SCAN_ROUND(0x114, 0xf)   // row_shr:4
	V2M_SCAN_ROUND(0x118, 0xf)   // row_shr:8
	V2M_SCAN_ROUND(0x142, 0xa)   // row_bcast:15 into rows 1 and 3
	V2M_SCAN_ROUND(0x143, 0xc)   // row_bcast:31 into rows 2 and 3
#undef V2M_SCAN_ROUND
}

// Lane i receives lane i + 1's value, lane 63 receives 0 (v_mov_b32_dpp wave_shl:1, no LDS round trip).
__device__ __forceinline__ u64 wave_shift_left_u64(u64 v)
{
	u32 const lo = (u32) __builtin_amdgcn_update_dpp(0, (int) (u32) v, 0x130, 0xf, 0xf, true);
	u32 const hi = (u32) __builtin_amdgcn_update_dpp(0, (int) (u32) (v >> 32), 0x130, 0xf, 0xf, true);
	return (u64) hi << 32 | lo;
}

// Non-zero bytes of tile[lo, hi).
__device__ __forceinline__ u32 count_nonzero_bytes(unsigned char const *tile, u32 lo, u32 hi)
{
	u32 n = 0;
	while (lo < hi && (lo & 3)) n += tile[lo++] != 0;
	for (; lo + 4 <= hi; lo += 4)
		n += 4 - __builtin_popcount(zero_bytes_mask(*(u32 const *) (tile + lo)));
	while (lo < hi) n += tile[lo++] != 0;
	return n;
}

constexpr int kCountRowsMax = 256;   // rows per group the count kernel can hold (host clamps rows_per_group)
constexpr int kCandDeltaLds = 1024;  // candidates per tile whose (row-independent) count change is kept in LDS

__global__ __launch_bounds__(kSpliceThreads) void count_unaligned_kernel(
	vec4u const *__restrict__ tmpl0, u64 const *__restrict__ eff, u64 eff_words_per_row,
	tile_tables tt, edge_patch const *__restrict__ patches, char const *__restrict__ labels,
	u32 *__restrict__ tile_counts /* [n_rows][n_tiles] */, u32 n_tiles,
	u32 n_rows, u32 rows_per_group, u32 n_groups, u32 tile_run)
{
	__shared__ vec4u lds[kTileChunks];        // the pristine template tile, shared by all rows of the group
	__shared__ patch_cache pc;
	__shared__ int row_delta[kCountRowsMax];
	__shared__ int cand_delta[kCandDeltaLds];
	__shared__ u32 wave_sums[kSpliceThreads / 64];

	int const t = threadIdx.x;
	u32 tile, group;
	map_block(blockIdx.x, n_groups, n_tiles, tile_run, tile, group);
	u32 const row_begin = group * rows_per_group;
	u32 const row_end = (row_begin + rows_per_group < n_rows) ? row_begin + rows_per_group : n_rows;

	u32 mine = 0;
#pragma unroll
	for (int k = 0; k < kChunksPerThread; ++k) {
		vec4u const v = tmpl0[(u64) tile * kTileChunks + t + kSpliceThreads * k];
		lds[t + kSpliceThreads * k] = v;
#pragma unroll
		for (int d = 0; d < 4; ++d)
			mine += 4 - __builtin_popcount(zero_bytes_mask(v[d]));
	}
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) mine += __shfl_down(mine, d, kWave);
	if ((t & 63) == 0) wave_sums[t >> 6] = mine;
	for (u32 r = t; r < (u32) kCountRowsMax; r += kSpliceThreads) row_delta[r] = 0;

	tile_job job;
	load_patch_cache(pc, job, tt, patches, labels, eff, eff_words_per_row, tile, row_begin, row_end - row_begin, t);
	__syncthreads();
	u32 tile_count = 0;
#pragma unroll
	for (int wv = 0; wv < kSpliceThreads / 64; ++wv) tile_count += wave_sums[wv];

	unsigned char const *const tile_bytes = (unsigned char const *) lds;
	// What an edge changes in the tile's count does not depend on the row (the effective edges of a row have disjoint
	// spans): label bytes it puts into the tile minus template bytes its span removes.  Computed once per candidate
	// here, so that the row loop is one LDS read per effective edge.
	auto patch_delta = [&](tile_patch const &tp) {
		u32 const label_end = tp.aln_begin + tp.label_len;
		int const label_in_tile = label_end > tp.from ? (int) ((label_end < tp.to ? label_end : tp.to) - tp.from) : 0;
		return label_in_tile - (int) count_nonzero_bytes(tile_bytes, tp.from - job.tile_base, tp.to - job.tile_base);
	};
	u32 const n_cand = job.n_cross + job.n_range;
	for (u32 i = t; i < n_cand && i < (u32) kCandDeltaLds; i += kSpliceThreads)
		cand_delta[i] = patch_delta(candidate_patch(pc, job, patches, i, candidate_edge(job, tt, i), candidate_in_lds(job, i)));
	__syncthreads();
	// The group may hold more rows than the LDS cache of effective-edge words (kGroupRowsLds): the template tile, its count
	// and the per-candidate changes are set up once per group, the cached words are reloaded every kGroupRowsLds rows.
	for (u32 sub = row_begin; sub < row_end; sub += kGroupRowsLds) {
		u32 const sub_end = sub + kGroupRowsLds < row_end ? sub + kGroupRowsLds : row_end;
		if (sub != row_begin) {
			// (explicit wait: hipcc leaves it out at this loop header -- harmless here, the only LDS operations possibly in
			// flight are the row_delta atomics, but tests/test_kernel_isa.py holds every barrier to the same rule)
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
			__syncthreads();                                     // everyone is done with the previous rows' words
			load_eff_cache(pc, job, eff, eff_words_per_row, sub, sub_end - sub, t);
			__syncthreads();
		}
		u32 const sub_rows = sub_end - sub;
		// Candidates whose effective-edge words are in LDS: one thread per (row, word), walking the set bits only.
		u32 const nw = job.n_lds ? ((job.range_begin + job.n_lds - 1) >> 6) - job.w0 + 1 : 0;
		for (u32 idx = t; idx < sub_rows * nw; idx += kSpliceThreads) {
			u32 const r = idx / nw, w = idx % nw;
			u32 const first = (job.w0 + w) * 64u;                 // edge of bit 0 of this word
			u64 bits = pc.eff[r][w];
			if (first < job.range_begin) bits &= ~0ULL << (job.range_begin - first);
			u32 const cached_end = job.range_begin + job.n_lds;
			if (cached_end < first + 64u) bits &= (1ULL << (cached_end - first)) - 1;
			int delta = 0;
			for (; bits; bits &= bits - 1) {
				u32 const e = first + (u32) __builtin_ctzll(bits);
				u32 const i = job.n_cross + (e - job.range_begin);
				delta += i < (u32) kCandDeltaLds ? cand_delta[i] : patch_delta(candidate_patch(pc, job, patches, i, e, true));
			}
			if (delta) atomicAdd(&row_delta[sub - row_begin + r], delta);
		}
		// The others (edges crossing into the tile from the left, candidates beyond the cache): one thread per (row, candidate).
		u32 const n_other = n_cand - job.n_lds;
		for (u32 idx = t; idx < sub_rows * n_other; idx += kSpliceThreads) {
			u32 const r = idx / n_other, j = idx % n_other;
			u32 const i = j < job.n_cross ? j : j + job.n_lds;
			u32 const e = candidate_edge(job, tt, i);
			if (!((eff[(u64) (sub + r) * eff_words_per_row + (e >> 6)] >> (e & 63)) & 1)) continue;
			int const delta = i < (u32) kCandDeltaLds ? cand_delta[i] : patch_delta(candidate_patch(pc, job, patches, i, e, false));
			if (delta) atomicAdd(&row_delta[sub - row_begin + r], delta);
		}
	}
	__syncthreads();
	for (u32 r = t; r < row_end - row_begin; r += kSpliceThreads)
		tile_counts[(u64) (row_begin + r) * n_tiles + tile] = (u32) ((int) tile_count + row_delta[r]);
}

// v_perm_b32 selector that moves the bytes of a dword named by the 4-bit mask `keep` to its low end, in order (0x0c = a
// zero byte): entry `keep` of the 16-entry LDS table the stream-out kernel compacts padded chunks with.
__device__ __forceinline__ u32 compaction_selector(u32 keep)
{
	u32 sel = 0x0c0c0c0cu, n = 0;
	for (u32 b = 0; b < 4; ++b)
		if (keep >> b & 1) { sel = (sel & ~(0xFFu << (8 * n))) | (b << (8 * n)); ++n; }
	return sel;
}

typedef unsigned short u16;
typedef u16 u16_unaligned __attribute__((aligned(1)));
typedef u32 u32_unaligned __attribute__((aligned(1)));
typedef u64 u64_unaligned __attribute__((aligned(1)));
typedef vec4u vec4u_unaligned __attribute__((aligned(1)));   // 16-B access at any byte address (gfx950 / HSA unaligned access mode; tools/unaligned_store_test.hip)

#ifdef V2M_TUNING_BUILD
// The stream-out of rounds 2-5, every wave packing its own short chunks: kept for the A/B (V2M_UNALIGNED_KERNEL=wave with the tuning library;
// profiles/r05/unaligned_shared_pack.txt).
template <bool kNonTemporal>
__global__ __launch_bounds__(kSpliceThreads) void splice_unaligned_per_wave_kernel(
	vec4u const *__restrict__ tmpl0, u64 const *__restrict__ eff, u64 eff_words_per_row,
	tile_tables tt, edge_patch const *__restrict__ patches, char const *__restrict__ labels,
	u32 const *__restrict__ tile_offsets /* [n_rows][n_tiles]: where each tile's bytes start in its row */, u32 n_tiles,
	char *__restrict__ out, u64 row_pitch, u32 n_rows, u32 rows_per_group, u32 n_groups, u32 tile_run)
{
	__shared__ vec4u lds[kTileChunks];
	__shared__ patch_cache pc;
	__shared__ u32 wave_sums[kChunksPerThread * (kSpliceThreads / 64)];   // per 1-KiB slot, in stream order: [k][wave]
	__shared__ u32 compact_sel[16];
	constexpr u32 kPackQueue = 24;
	__shared__ u32 pack_queue[kSpliceThreads / 64][kPackQueue];    // per wave: the short chunks of the row tile waiting to be packed (see the stream-out below)

	int const t = threadIdx.x;
	int const lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);   // (scalar: slot numbers and LDS bases derived from it stay in SGPRs)
	if (t < 16) compact_sel[t] = compaction_selector((u32) t);   // first read follows the row loop's barriers
	u32 tile, group;
	map_block(blockIdx.x, n_groups, n_tiles, tile_run, tile, group);
	{
		// Neighbouring tiles of a row share the line at their (byte-granular) boundary, and map_block() deals consecutive tiles to
		// the 8 XCDs in turn: the two pieces of every such line would meet in no L2.  So within its super-block every XCD takes a run of
		// consecutive tiles instead of every eighth one (still a fixed tile -> XCD map: the template tiles keep their L2).  Config 3:
		// 11.4 -> 10.9 ms per 620 rows; config 5 (where the kernel is bound by instruction issue): no change.
		u32 const t0 = tile / tile_run * tile_run;
		u32 const run = (n_tiles - t0 < tile_run) ? n_tiles - t0 : tile_run;
		if (0 == run % 8) { u32 const j = tile - t0; tile = t0 + (j & 7) * (run / 8) + (j >> 3); }
	}
	u32 const row_begin = group * rows_per_group;
	u32 const row_end = (row_begin + rows_per_group < n_rows) ? row_begin + rows_per_group : n_rows;

	vec4u pristine[kChunksPerThread];
#pragma unroll
	for (int k = 0; k < kChunksPerThread; ++k)
		pristine[k] = tmpl0[(u64) tile * kTileChunks + t + kSpliceThreads * k];

	tile_job job;
	load_patch_cache(pc, job, tt, patches, labels, eff, eff_words_per_row, tile, row_begin, row_end - row_begin, t);

	for (u32 row = row_begin; row < row_end; ++row) {
		// (everybody read the previous row's tile before the barrier that follows its scan, so no barrier is needed here)
#pragma unroll
		for (int k = 0; k < kChunksPerThread; ++k)
			lds[t + kSpliceThreads * k] = pristine[k];
		__syncthreads();

		patch_row_tile((unsigned char *) lds, pc, job, tt, patches, labels, eff + (u64) row * eff_words_per_row, row - row_begin, t, 0);

		// Thread t owns the 16-B chunks t, t + 256, ... (lane-contiguous, like the aligned kernel).  Where a chunk's
		// surviving bytes go = the number of surviving bytes in all chunks before it, in chunk order.
		vec4u v[kChunksPerThread];
		u32 cnt[kChunksPerThread], incl[kChunksPerThread];
#pragma unroll
		for (int k = 0; k < kChunksPerThread; ++k) v[k] = lds[t + kSpliceThreads * k];
#pragma unroll
		for (int k = 0; k < kChunksPerThread; ++k) {
			// Non-padding bytes of the chunk: v_msad_u8 adds |a - b| over the bytes whose reference byte b is not 0, and
			// |(b ^ 1) - b| = 1 for every b: two instructions per dword instead of the five of zero-byte mask + popcount.
			// (Round 3: with the slot bases below, 78 -> 65 VGPRs, so 7 workgroups per CU instead of 6; that, more than the
			// ~30 instructions saved per wave and tile, is what took 8 % off the kernel.)
			u32 bytes = 0;
#pragma unroll
			for (int d = 0; d < 4; ++d) bytes = __builtin_amdgcn_msad_u8(v[k][d] ^ 0x01010101u, v[k][d], bytes);
			incl[k] = cnt[k] = bytes;
		}
		wave_inclusive_scan_u32_lockstep(incl);
#pragma unroll
		for (int k = 0; k < kChunksPerThread; ++k)
			if (lane == 63) wave_sums[k * (kSpliceThreads / 64) + wave] = incl[k];
		__syncthreads();

		// Where each of the 16 slots starts: their byte counts scanned in stream order by the first 16 lanes of every wave
		// (one LDS read and four DPP adds instead of four reads and a dozen adds per slot), picked out with v_readlane.
		u32 slot_end = lane < kChunksPerThread * (kSpliceThreads / 64) ? wave_sums[lane] : 0u;
		slot_end += (u32) __builtin_amdgcn_update_dpp(0, (int) slot_end, 0x111, 0xf, 0xf, false);   // row_shr:1 (lanes 0..15 are one DPP row)
		slot_end += (u32) __builtin_amdgcn_update_dpp(0, (int) slot_end, 0x112, 0xf, 0xf, false);   // row_shr:2
		slot_end += (u32) __builtin_amdgcn_update_dpp(0, (int) slot_end, 0x114, 0xf, 0xf, false);   // row_shr:4
		slot_end += (u32) __builtin_amdgcn_update_dpp(0, (int) slot_end, 0x118, 0xf, 0xf, false);   // row_shr:8

		char *const dst = out + (u64) row * row_pitch + tile_offsets[(u64) row * n_tiles + tile];

		// Stream-out.  A chunk without padding goes out as it is: one 16-B store at its byte-granular destination (the wave's 64 of them
		// cover one contiguous KiB at whatever byte phase the row is in).  A chunk that holds padding has to be packed first, and that
		// costs ~100 VALU instructions which a wave pays whether one lane needs them or all 64 -- on a dense graph (config 5: a gap
		// every 500 bases) nearly every 1-KiB slot holds ONE or TWO such chunks, and packing "where they are" made this kernel run at
		// 100 % VALU issue, 512 instructions per wave and row tile against the aligned kernel's 114 (profiles/r05/unaligned_pmc_*).
		// So the short chunks of the wave's four slots are first QUEUED -- descriptor = chunk | bytes << 10 | destination << 14, one
		// LDS word each, written under the slot's mask at mbcnt positions: no loop, no branch -- and ONE pass packs them all, a lane
		// per chunk, reading the chunks back from the row tile (still whole: only this wave rewrites its own slots, in program order).
		// A slot whose short chunks do not fit the queue any more (kPackQueue entries per wave: what fits beside the tile without
		// costing the seventh workgroup per CU; tiles inside long insertions) is packed where it is, by the same code: stage 2 below
		// is one loop over "the queued ones" and "dense slot k", so the pack code exists once.
		u32 offs[kChunksPerThread];
		u32 queued = 0, dense = 0;                                               // wave-uniform
#pragma unroll
		for (int k = 0; k < kChunksPerThread; ++k) {
			int const slot = k * (kSpliceThreads / 64) + wave;                       // wave-uniform
			u32 const slot_begin = slot ? (u32) __builtin_amdgcn_readlane((int) slot_end, slot - 1) : 0u;
			u32 const c = cnt[k];
			u32 const off = slot_begin + incl[k] - c;
			offs[k] = off;
			if (16 == c) {
				if (kNonTemporal) __builtin_nontemporal_store(v[k], (vec4u_unaligned *) (dst + off));
				else *(vec4u_unaligned *) (dst + off) = v[k];
			}
			bool const partial = c - 1u < 15u;                                       // 1 ... 15 surviving bytes (an all-padding chunk writes nothing)
			u64 const mask = __ballot(partial);
			u32 const n = (u32) __builtin_popcountll(mask);                           // (scalar)
			bool const fits = queued + n <= kPackQueue;                               // (scalar)
			u32 const pos = __builtin_amdgcn_mbcnt_hi((u32) (mask >> 32), __builtin_amdgcn_mbcnt_lo((u32) mask, queued));
			if (partial && pos < kPackQueue) pack_queue[wave][pos] = (u32) (t + kSpliceThreads * k) | c << 10 | off << 14;   // (entries of a slot that does not fit are overwritten or ignored)
			dense |= (n && !fits) ? 1u << k : 0u;
			queued = fits ? queued + n : queued;
		}
		u32 const pending = pack_queue[wave][lane < (int) kPackQueue ? lane : 0];   // (same wave wrote them: LDS operations of a wave execute in order)
		for (u32 todo = dense | (queued ? 1u << kChunksPerThread : 0u); todo; todo &= todo - 1) {   // wave-uniform
			u32 const j = (u32) __builtin_ctz(todo);
			u32 e = pending;
			bool active = (u32) lane < queued;
			if (j < (u32) kChunksPerThread) {                                         // (uniform) dense slot j, every lane its own chunk
				u32 const c = 0 == j ? cnt[0] : 1 == j ? cnt[1] : 2 == j ? cnt[2] : cnt[3];
				u32 const o = 0 == j ? offs[0] : 1 == j ? offs[1] : 2 == j ? offs[2] : offs[3];
				e = ((u32) t + (u32) kSpliceThreads * j) | c << 10 | o << 14;
				active = c - 1u < 15u;
			}
			if (!active) continue;
			u32 const c = (e >> 10) & 15u;
			vec4u const x = lds[e & 1023u];
			// the chunk's surviving bytes packed to the low end of a 16-B value: v_perm_b32 per dword with the selector table, then the four
			// pieces shifted together
			u32 piece[4], len[4];
#pragma unroll
			for (int d = 0; d < 4; ++d) {
				u32 const keep = ~zero_bytes_mask(x[d]) & 0x80808080u;             // 0x80 per surviving byte (bits 7, 15, 23, 31)
				// ... gathered into the top nibble: neighbours side by side at bits 14|15, 22|23, 30|31, then all four at 28..31.
				// (inline assembly because the compiler recognises x | x << 7 | ... as a multiplication by a constant and
				// emits v_mul_lo_u32, which issues at a quarter of the rate of these two)
				u32 pairs, quad;
				asm("v_lshl_or_b32 %0, %1, 7, %1" : "=v"(pairs) : "v"(keep));
				asm("v_lshl_or_b32 %0, %1, 14, %1" : "=v"(quad) : "v"(pairs));
				u32 const m = quad >> 28;
				piece[d] = __builtin_amdgcn_perm(0u, x[d], compact_sel[m]);
				len[d] = (u32) __builtin_popcount(m);
			}
			u64 const lo = (u64) piece[0] | ((u64) piece[1] << (8 * len[0]));
			u64 const hi = (u64) piece[2] | ((u64) piece[3] << (8 * len[2]));
			u32 const sh = 8 * (len[0] + len[1]);                                  // 0 ... 64
			u64 const packed_lo = lo | (sh < 64 ? hi << sh : 0);
			u64 const packed_hi = 0 == sh ? 0 : (64 == sh ? hi : hi >> (64 - sh));
			// ... and stored exactly: at most one store each of 8, 4, 2 and 1 bytes (the bytes around them belong to other chunks' stores)
			char *p = dst + (e >> 14);
			u64 rest = packed_lo;
			if (c & 8) { *(u64_unaligned *) p = packed_lo; p += 8; rest = packed_hi; }
			if (c & 4) { *(u32_unaligned *) p = (u32) rest; p += 4; rest >>= 32; }
			if (c & 2) { *(u16_unaligned *) p = (u16) rest; p += 2; rest >>= 16; }
			if (c & 1) *p = (char) rest;
		}
		// wave_sums is rewritten only after the next row's barriers
	}
}
#endif   // V2M_TUNING_BUILD

// The surviving (non-zero) bytes of the 16-B chunk x, c of them, packed to the low end of a 16-B value and stored at p exactly: at most one
// store each of 8, 4, 2 and 1 bytes (the bytes around them belong to other chunks' stores).  v_perm_b32 per dword with the 16-entry
// selector table, then the four pieces shifted together.
__device__ __forceinline__ void pack_chunk_and_store_exact(vec4u const x, u32 const c, char *p, u32 const *compact_sel)
{
	u32 piece[4], len[4];
#pragma unroll
	for (int d = 0; d < 4; ++d) {
		u32 const keep = ~zero_bytes_mask(x[d]) & 0x80808080u;             // 0x80 per surviving byte (bits 7, 15, 23, 31)
		// ... gathered into the top nibble (inline assembly: the compiler would make a v_mul_lo_u32 of it, a quarter-rate instruction)
		u32 pairs, quad;
		asm("v_lshl_or_b32 %0, %1, 7, %1" : "=v"(pairs) : "v"(keep));
		asm("v_lshl_or_b32 %0, %1, 14, %1" : "=v"(quad) : "v"(pairs));
		u32 const m = quad >> 28;
		piece[d] = __builtin_amdgcn_perm(0u, x[d], compact_sel[m]);
		len[d] = (u32) __builtin_popcount(m);
	}
	u64 const lo = (u64) piece[0] | ((u64) piece[1] << (8 * len[0]));
	u64 const hi = (u64) piece[2] | ((u64) piece[3] << (8 * len[2]));
	u32 const sh = 8 * (len[0] + len[1]);                                  // 0 ... 64
	u64 const packed_lo = lo | (sh < 64 ? hi << sh : 0);
	u64 const packed_hi = 0 == sh ? 0 : (64 == sh ? hi : hi >> (64 - sh));
	u64 rest = packed_lo;
	if (c & 8) { *(u64_unaligned *) p = packed_lo; p += 8; rest = packed_hi; }
	if (c & 4) { *(u32_unaligned *) p = (u32) rest; p += 4; rest >>= 32; }
	if (c & 2) { *(u16_unaligned *) p = (u16) rest; p += 2; rest >>= 16; }
	if (c & 1) *p = (char) rest;
}

// The unaligned stream-out with ONE packing pass per workgroup and row tile instead of one per wave (round 5).
//
// Counters on config 5 (profiles/r05/unaligned_pmc_config5_*): per wave and row tile the per-wave stream-out (rounds 2-5, now in the tuning build) issues 308 VALU + 213 SALU
// instructions against the aligned kernel's 114 + 117, and a SIMD gets 479 issue turns per row tile at the HBM-bound pace: the
// kernel is bound by instruction issue, and ~40 % of what it issues is the packing pass (pack + exact stores), which a wave
// pays in full whether one of its lanes has a short chunk or all 64 -- on a dense graph every wave has a few, every row tile.
// Here the short chunks of the whole 16-KiB row tile (config 3: ~35, config 5: ~70) go into ONE queue per workgroup --
// descriptor (bytes | destination << 4) and the chunk's 16 bytes, at positions that fall out of the byte-count scan itself (the
// short-chunk flag rides in the high half of the scanned word: no ballot, no mbcnt) -- and are packed 64 at a time by ONE wave,
// a row later: row r's queue is complete at the barrier that follows row r + 1's tile build, after which wave (r + pass) % 4
// packs it while the others go on with row r + 1 (so no barrier is added, and the packing wave rotates).  Slots whose short
// chunks no longer fit the queue (tiles inside long insertions) are packed where they are by their own wave.
// Measured (profiles/r05/unaligned_shared_pack.txt): config 5 11.5 -> 10.95 ms per 244 rows (aligned kernel, same rows: 9.03), VALU 308 -> 256, SALU 213 -> ~156
// per wave and row tile (the scans in lockstep took the rest); config 3 within what two boxes differ by.
template <bool kNonTemporal, u32 kQueue = 128>
__global__ __launch_bounds__(kSpliceThreads) void splice_unaligned_kernel(
	vec4u const *__restrict__ tmpl0, u64 const *__restrict__ eff, u64 eff_words_per_row,
	tile_tables tt, edge_patch const *__restrict__ patches, char const *__restrict__ labels,
	u32 const *__restrict__ tile_offsets /* [n_rows][n_tiles]: where each tile's bytes start in its row */, u32 n_tiles,
	char *__restrict__ out, u64 row_pitch, u32 n_rows, u32 rows_per_group, u32 n_groups, u32 tile_run)
{
	constexpr int kWaves = kSpliceThreads / 64, kSlots = kChunksPerThread * kWaves;
	// kQueue: short chunks of a row tile that wait for the packing wave(s)
	__shared__ vec4u lds[kTileChunks];
	__shared__ patch_cache pc;
	__shared__ __attribute__((aligned(16))) u32 wave_sums[kSlots];         // per 1-KiB slot: [wave][k]; bytes | short chunks << 16
	__shared__ u32 compact_sel[16];
	__shared__ vec4u queue_data[kQueue];
	__shared__ u32 queue_desc[kQueue];
	__shared__ u32 queue_count;

	int const t = threadIdx.x;
	int const lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
	if (t < 16) compact_sel[t] = compaction_selector((u32) t);   // first read follows the row loop's barriers
	u32 tile, group;
	map_block(blockIdx.x, n_groups, n_tiles, tile_run, tile, group);
	{
		// (as in the kernel above: within its super-block every XCD takes a run of consecutive tiles)
		u32 const t0 = tile / tile_run * tile_run;
		u32 const run = (n_tiles - t0 < tile_run) ? n_tiles - t0 : tile_run;
		if (0 == run % 8) { u32 const j = tile - t0; tile = t0 + (j & 7) * (run / 8) + (j >> 3); }
	}
	u32 const row_begin = group * rows_per_group;
	u32 const row_end = (row_begin + rows_per_group < n_rows) ? row_begin + rows_per_group : n_rows;

	vec4u pristine[kChunksPerThread];
#pragma unroll
	for (int k = 0; k < kChunksPerThread; ++k)
		pristine[k] = tmpl0[(u64) tile * kTileChunks + t + kSpliceThreads * k];

	tile_job job;
	load_patch_cache(pc, job, tt, patches, labels, eff, eff_words_per_row, tile, row_begin, row_end - row_begin, t);

	char *dst_prev = out;
	for (u32 row = row_begin; ; ++row) {                                      // (one round more than there are rows: the last one only packs the last row's queue)
		bool const past_end = row >= row_end;                                  // (uniform)
		u32 const tile_offset = past_end ? 0u : tile_offsets[(u64) row * n_tiles + tile];   // (needed after the third barrier: asked for here)
		if (!past_end) {
			if (row != row_begin && 0 == (row - row_begin) % (u32) kGroupRowsLds)   // (as in the aligned kernel: the next kGroupRowsLds rows' effective-edge words)
				load_eff_cache(pc, job, eff, eff_words_per_row, row, row_end - row, t);
#pragma unroll
			for (int k = 0; k < kChunksPerThread; ++k)
				lds[t + kSpliceThreads * k] = pristine[k];
		}
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // (the previous row's queue writes precede this barrier across the loop's back edge)
		__syncthreads();

		if (row != row_begin) {
			// the queue of the row before: packed by wave (row + pass) % 4, 64 entries a pass
			u32 const n = (u32) __builtin_amdgcn_readfirstlane((int) queue_count);
			for (u32 pass = 0; 64 * pass < n; ++pass) {
				if (((row + pass) & (u32) (kWaves - 1)) != (u32) wave) continue;
				u32 const idx = 64 * pass + (u32) lane;
				if (idx < n) {
					u32 const e = queue_desc[idx];
					pack_chunk_and_store_exact(queue_data[idx], e & 15u, dst_prev + (e >> 4), compact_sel);
				}
			}
		}
		if (past_end) break;

		patch_row_tile((unsigned char *) lds, pc, job, tt, patches, labels, eff + (u64) row * eff_words_per_row, (row - row_begin) % (u32) kGroupRowsLds, t, 0);

		// Thread t owns the 16-B chunks t, t + 256, ...  Scanned word: surviving bytes in the low half, "short chunk" (1 ... 15 of them) in the high half.
		vec4u v[kChunksPerThread];
		u32 mine[kChunksPerThread], incl[kChunksPerThread];
#pragma unroll
		for (int k = 0; k < kChunksPerThread; ++k) v[k] = lds[t + kSpliceThreads * k];
#pragma unroll
		for (int k = 0; k < kChunksPerThread; ++k) {
			u32 bytes = 0;
#pragma unroll
			for (int d = 0; d < 4; ++d) bytes = __builtin_amdgcn_msad_u8(v[k][d] ^ 0x01010101u, v[k][d], bytes);
			incl[k] = mine[k] = bytes | (bytes - 1u < 15u ? 0x10000u : 0u);
		}
		wave_inclusive_scan_u32_lockstep(incl);
		if (lane == 63) {                                                      // one 16-B store: the wave's four slot totals side by side
			vec4u sums;
#pragma unroll
			for (int k = 0; k < kChunksPerThread; ++k) sums[k] = incl[k];
			*(vec4u *) &wave_sums[kChunksPerThread * wave] = sums;
		}
		__syncthreads();

		// slot k * kWaves + wave (stream order) sits at wave_sums[kChunksPerThread * wave + k]
		u32 slot_end = lane < kSlots ? wave_sums[kChunksPerThread * (lane % kWaves) + lane / kWaves] : 0u;
		slot_end += (u32) __builtin_amdgcn_update_dpp(0, (int) slot_end, 0x111, 0xf, 0xf, false);   // row_shr:1 (lanes 0..15 are one DPP row)
		slot_end += (u32) __builtin_amdgcn_update_dpp(0, (int) slot_end, 0x112, 0xf, 0xf, false);   // row_shr:2
		slot_end += (u32) __builtin_amdgcn_update_dpp(0, (int) slot_end, 0x114, 0xf, 0xf, false);   // row_shr:4
		slot_end += (u32) __builtin_amdgcn_update_dpp(0, (int) slot_end, 0x118, 0xf, 0xf, false);   // row_shr:8

		char *const dst = out + (u64) row * row_pitch + tile_offset;
		// the slots whose short chunks fit the queue: a prefix of the slots (the counts only grow), read off a ballot
		u32 const n_fit = (u32) __builtin_popcountll(__ballot(lane < kSlots && (slot_end >> 16) <= kQueue));            // (uniform)
		u32 const n_queued = n_fit ? (u32) __builtin_amdgcn_readlane((int) slot_end, (int) n_fit - 1) >> 16 : 0u;   // (uniform)
		u32 offs[kChunksPerThread];
		u32 dense = 0;                                                         // (uniform) this wave's slots that do not fit
#pragma unroll
		for (int k = 0; k < kChunksPerThread; ++k) {
			int const slot = k * kWaves + wave;                                  // wave-uniform
			u32 const slot_begin = slot ? (u32) __builtin_amdgcn_readlane((int) slot_end, slot - 1) : 0u;
			u32 const c = mine[k] & 0xFFFFu;
			u32 const before = slot_begin + incl[k] - mine[k];                   // both halves at once: bytes and short chunks before this one
			u32 const off = before & 0xFFFFu;
			offs[k] = off;
			if (16 == c) {
				if (kNonTemporal) __builtin_nontemporal_store(v[k], (vec4u_unaligned *) (dst + off));
				else *(vec4u_unaligned *) (dst + off) = v[k];
			}
			if ((u32) slot < n_fit) {                                             // (uniform)
				if (mine[k] >> 16) {
					u32 const pos = before >> 16;
					queue_desc[pos] = c | off << 4;
					queue_data[pos] = v[k];
				}
			}
			else dense |= 1u << k;
		}
		for (; dense; dense &= dense - 1) {                                    // (rare: tiles inside long insertions) packed where they are, read back from the row tile
			u32 const j = (u32) __builtin_ctz(dense);
			u32 const c = (0 == j ? mine[0] : 1 == j ? mine[1] : 2 == j ? mine[2] : mine[3]) & 0xFFFFu;
			u32 const o = 0 == j ? offs[0] : 1 == j ? offs[1] : 2 == j ? offs[2] : offs[3];
			if (c - 1u < 15u) pack_chunk_and_store_exact(lds[(u32) t + (u32) kSpliceThreads * j], c, dst + o, compact_sel);
		}
		if (0 == t) queue_count = n_queued;
		dst_prev = dst;
		// wave_sums is rewritten only after the next row's barriers
	}
}

// Exclusive prefix sum of the per-tile byte counts of each row (in place) and the row lengths.
__global__ __launch_bounds__(256) void scan_tile_counts_kernel(u32 *__restrict__ tile_counts, u32 n_tiles, u64 *__restrict__ row_lengths)
{
	__shared__ u32 wave_sums[4];
	__shared__ u32 carry_s;
	u32 *const row = tile_counts + (u64) blockIdx.x * n_tiles;
	int const t = threadIdx.x, lane = t & 63, wave = t >> 6;
	if (t == 0) carry_s = 0;
	__syncthreads();
	for (u32 base = 0; base < n_tiles; base += 256) {
		u32 const i = base + t;
		u32 const mine = i < n_tiles ? row[i] : 0;
		u32 const incl = wave_inclusive_scan_u32(mine);
		if (lane == 63) wave_sums[wave] = incl;
		__syncthreads();
		u32 before = carry_s, total = 0;
#pragma unroll
		for (int wv = 0; wv < 4; ++wv) {
			u32 const ws = wave_sums[wv];
			if (wv < wave) before += ws;
			total += ws;
		}
		if (i < n_tiles) row[i] = before + incl - mine;
		__syncthreads();
		if (t == 0) carry_s += total;
		__syncthreads();
	}
	if (t == 0) row_lengths[blockIdx.x] = carry_s;
}


// ---------------------------------------------------------------------------------------------
// Write-rate probe for output buffers (v2m_alloc_output): the splice kernel's store pattern -- n_groups x 16
// rows `pitch` apart advancing together in 16-KiB segments, nontemporal 16-B/lane stores -- with no reads.
// On MI355X the rate this pattern reaches differs by ~25 % between physical regions of HBM
// (tools/streams_probe.hip), so output buffers are chosen by measurement.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kSpliceThreads) void probe_write_kernel(char *__restrict__ out, u64 pitch, u32 n_groups)
{
	u32 const tile = blockIdx.x / n_groups, group = blockIdx.x % n_groups;
	vec4u const v = {0, 0, 0, 0};
	for (u32 r = 0; r < 16; ++r) {
		char *const dst = out + (u64) (group * 16 + r) * pitch + (u64) tile * kTileBytes;
#pragma unroll
		for (int k = 0; k < kChunksPerThread; ++k)
			__builtin_nontemporal_store(v, (vec4u *) (dst + (threadIdx.x + kSpliceThreads * k) * 16));
	}
}


// ---------------------------------------------------------------------------------------------
// Row checksums (verification helper): sum over 8-byte little-endian words w (zero padded past
// the row's length) of mix64((w_index + 1) * GOLDEN ^ word), plus mix64(length).
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ u64 mix64(u64 z)
{
	z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
	z ^= z >> 27; z *= 0x94D049BB133111EBULL;
	z ^= z >> 31;
	return z;
}

constexpr u64 kGolden = 0x9E3779B97F4A7C15ULL;

__global__ __launch_bounds__(256) void checksum_rows_kernel(
	char const *__restrict__ rows, u64 row_pitch, u64 const *__restrict__ lengths, u64 fixed_length,
	u64 words_per_block, unsigned long long *__restrict__ sums)
{
	u32 const row = blockIdx.y;
	u64 const len = lengths ? lengths[row] : fixed_length;
	u64 const n_words = (len + 7) / 8;
	u64 const w_begin = (u64) blockIdx.x * words_per_block;
	u64 w_end = w_begin + words_per_block;
	if (w_end > n_words) w_end = n_words;
	u64 const *const base = (u64 const *) (rows + (u64) row * row_pitch);
	u64 acc = 0;
	for (u64 w = w_begin + threadIdx.x; w < w_end; w += blockDim.x) {
		u64 v = base[w];
		if (w == n_words - 1 && (len & 7))
			v &= (1ULL << (8 * (len & 7))) - 1;
		acc += mix64((w + 1) * kGolden ^ v);
	}
	if (blockIdx.x == 0 && threadIdx.x == 0) acc += mix64(len);
	// wave reduce, then one atomic per wave
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) {
		u32 lo = __shfl_down((u32) acc, d, kWave);
		u32 hi = __shfl_down((u32) (acc >> 32), d, kWave);
		acc += ((u64) hi << 32) | lo;
	}
	if ((threadIdx.x & 63) == 0 && acc)
		atomicAdd(&sums[row], (unsigned long long) acc);
}

} // namespace v2m
