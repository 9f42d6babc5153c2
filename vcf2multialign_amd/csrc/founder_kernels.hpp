// founder_kernels.hpp -- the chunk walks of the founder search on the GPU.
//
// The reference's cut search (find_initial_cut_positions_lambda_min, libvcf2multialign/find_cut_positions.cc:93-211) walks a
// positional BWT over the ALT edges (pbwt_context::update_divergence, include/vcf2multialign/pbwt.hh:77-134): one step per
// edge, each depending on the previous one, and at every candidate node a pass over the distinct divergence values from the
// largest down (find_cut_positions.cc:134-165).  The host (csrc/host/founder.cc) cuts the edges into chunks whose start
// state is built from scratch from the transposed path matrix; what a chunk's worker does from there -- thousands of pBWT
// steps over a few thousand chromosome copies and the per-candidate value walk -- is one workgroup's job here:
//
//   pBWT step   a stable partition of the copies by the edge's bit plus two running maxima with resets (Durbin's algorithm 2):
//               both are prefix scans, of a count and of functions "p -> max(p, a)" / "p -> c" composed left to right
//   candidate   class_count(v) = #{copies with divergence > v} only changes between candidates' edge ranges, so the walk
//               over distinct values collapses to: bin the copies by the candidate their divergence value points to (an LDS
//               hash; about a hundred bins at config 4), then for every bin the copies and the emitting bins among the bins
//               with a larger key -- one sum over the pairs of bins, spread over the whole workgroup; no sort, no scan
//
// Divergence values are BIASED by one as on the host (0 = "no match yet", founder.cc), so everything is plain u32 order.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace v2m {

constexpr int kPbwtThreads = 1024;                    // (512 / 256 threads with twice / four times the copies each: 40.5 + 22.0 ms / 71.7 + 39.9 ms at config 4 against 30.5 + 17.3)
constexpr int kPbwtWaves = kPbwtThreads / 64;
// Chromosome copies a workgroup can walk: its pBWT state lives in LDS -- order (u16) and divergence (u32), ONE array each (a step reads all
// of its copies into registers, passes a barrier and writes them back to their new places, so no second buffer is needed: round 5; with two
// the state of 8192 copies was all that fit).  The kernels are instantiated per copies-per-thread count and size their arrays for it:
// the cut search's kernel (state + the candidates' hash and bins) reaches 20 480 copies -- BASELINE config 5's 20 000 -- and so does the matching's:
// up to 12 288 copies its two class arrays and the joined classes' starts (6 bytes per copy, touched once per CUT, not per step) sit in LDS
// beside the state, above that in a scratch area of global memory the host provides (kPbwtClassesInLds).
constexpr int kPbwtMaxCopies = 20480;
constexpr int kPbwtMaxCopiesRecords = kPbwtMaxCopies;
constexpr int kPbwtPerThread = kPbwtMaxCopies / kPbwtThreads;          // 20
constexpr int kPbwtPerThreadRecords = kPbwtMaxCopiesRecords / kPbwtThreads;   // 20
template <int kPer> constexpr bool kPbwtClassesInLds = kPer <= 12;          // (12 288 copies: state + class arrays = 150 KB)
constexpr int kPbwtClassScratchArrays = 3;                                  // per chunk, of 1024 * kPer unsigned shorts each: the two class arrays, the joined classes' starts
// distinct candidate bins per candidate node: about a hundred in practice, at most kPbwtMaxBins; the largest instantiations take the smaller table
template <int kPer> constexpr int pbwt_hash_slots() { return kPer > 8 ? 2048 : 4096; }
constexpr int kPbwtMaxBins = kPbwtThreads;            // more distinct bins than this at one candidate (a thread per bin): the chunk is left to the host

// __syncthreads() with the LDS wait spelled out: on loop back edges hipcc (ROCm 7.2) has emitted the barrier without the
// s_waitcnt lgkmcnt(0) that __syncthreads() implies (kernels.hpp, ring transpose; tests/test_kernel_isa.py checks every
// barrier of every kernel), and these kernels are all loops around barriers.
__device__ __forceinline__ void pbwt_block_sync()
{
	asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	__syncthreads();
}

// "p -> is_const ? value : max(p, value)": what a run of copies does to a running maximum that is reset after every copy of
// its own class -- packed into one register: bit 31 = is_const, the low bits the value (biased divergence values are edge
// indices + 2, far below 2^31; the host refuses graphs where they are not).  0 is the identity.
constexpr uint32_t kChainConst = 0x80000000u;

// first `first`, after that `next`
__device__ __forceinline__ uint32_t chain_then(uint32_t first, uint32_t next)
{
	uint32_t const keep = (first & kChainConst) | next;               // v_and_or_b32
	uint32_t const m = first > keep ? first : keep;                   // a constant stays one, its value grows; otherwise a plain max
	return (int32_t) next < 0 ? next : m;
}

__device__ __forceinline__ uint32_t chain_apply(uint32_t c, uint32_t p) { return (int32_t) c < 0 ? (c & ~kChainConst) : (p > c ? p : c); }

struct pbwt_scan_item {
	uint32_t zeros;       // copies that do not use the edge
	uint32_t p, q;        // the running maxima of the two classes (pbwt.hh:93-131), as packed chains
};

__device__ __forceinline__ pbwt_scan_item scan_combine(pbwt_scan_item const &a, pbwt_scan_item const &b)   // a, then b
{
	return pbwt_scan_item{a.zeros + b.zeros, chain_then(a.p, b.p), chain_then(a.q, b.q)};
}

// v_mov_b32_dpp with zeros where a lane has no source (or its row is masked out): the identity of every scan below, so a
// round of a scan is "fetch, combine" with no per-lane select.  No LDS round trip, unlike __shfl_up (ds_bpermute_b32).
template <int kCtrl, int kRowMask>
__device__ __forceinline__ uint32_t pbwt_dpp(uint32_t v)
{
	// all rows enabled: bound_ctrl shifts zeros in, and the instruction needs no `old` operand set up beforehand; with rows masked out
	// (the two broadcasts) those rows must come out as 0, which is what `old` = 0 gives them
	return (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, kCtrl, kRowMask, 0xf, 0xf == kRowMask);
}

constexpr int kDppRowShr1 = 0x111, kDppRowShr2 = 0x112, kDppRowShr4 = 0x114, kDppRowShr8 = 0x118, kDppBcast15 = 0x142, kDppBcast31 = 0x143, kDppWaveShr1 = 0x138;

template <int kCtrl, int kRowMask>
__device__ __forceinline__ pbwt_scan_item scan_round(pbwt_scan_item const &x)
{
	pbwt_scan_item const up{pbwt_dpp<kCtrl, kRowMask>(x.zeros), pbwt_dpp<kCtrl, kRowMask>(x.p), pbwt_dpp<kCtrl, kRowMask>(x.q)};
	return scan_combine(up, x);
}

// inclusive scan inside every row of 16 lanes (four shifts)
__device__ __forceinline__ pbwt_scan_item row_inclusive_scan(pbwt_scan_item x)
{
	x = scan_round<kDppRowShr1, 0xf>(x);
	x = scan_round<kDppRowShr2, 0xf>(x);
	x = scan_round<kDppRowShr4, 0xf>(x);
	x = scan_round<kDppRowShr8, 0xf>(x);
	return x;
}

// ... and across the wave: lane 15 of rows 0 / 2 into rows 1 / 3, then lane 31 into rows 2 and 3 (the operator is not commutative;
// what arrives always lies BEFORE the lane's own run)
__device__ __forceinline__ pbwt_scan_item wave_inclusive_scan(pbwt_scan_item x)
{
	x = row_inclusive_scan(x);
	x = scan_round<kDppBcast15, 0xa>(x);
	x = scan_round<kDppBcast31, 0xc>(x);
	return x;
}

__device__ __forceinline__ uint32_t wave_inclusive_add(uint32_t v)
{
	v += pbwt_dpp<kDppRowShr1, 0xf>(v);
	v += pbwt_dpp<kDppRowShr2, 0xf>(v);
	v += pbwt_dpp<kDppRowShr4, 0xf>(v);
	v += pbwt_dpp<kDppRowShr8, 0xf>(v);
	v += pbwt_dpp<kDppBcast15, 0xa>(v);
	v += pbwt_dpp<kDppBcast31, 0xc>(v);
	return v;
}

__device__ __forceinline__ uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }

__device__ __forceinline__ uint32_t row_inclusive_max(uint32_t v)
{
	v = umax(v, pbwt_dpp<kDppRowShr1, 0xf>(v));
	v = umax(v, pbwt_dpp<kDppRowShr2, 0xf>(v));
	v = umax(v, pbwt_dpp<kDppRowShr4, 0xf>(v));
	v = umax(v, pbwt_dpp<kDppRowShr8, 0xf>(v));
	return v;
}

__device__ __forceinline__ uint32_t wave_inclusive_max(uint32_t v)
{
	v = row_inclusive_max(v);
	v = umax(v, pbwt_dpp<kDppBcast15, 0xa>(v));
	v = umax(v, pbwt_dpp<kDppBcast31, 0xc>(v));
	return v;
}

__device__ __forceinline__ uint32_t row_inclusive_add(uint32_t v)
{
	v += pbwt_dpp<kDppRowShr1, 0xf>(v);
	v += pbwt_dpp<kDppRowShr2, 0xf>(v);
	v += pbwt_dpp<kDppRowShr4, 0xf>(v);
	v += pbwt_dpp<kDppRowShr8, 0xf>(v);
	return v;
}

// The edge's bit column (words_per_edge <= 128 words, one per thread) is fetched a whole step ahead into a register and stashed in
// the LDS buffer the NEXT step reads while the current one's second pass runs, so a step never stands still for a global load and
// the stash needs no barrier of its own (steps always visit consecutive edges; `edge_limit` clamps the fetch after the last one).
__device__ __forceinline__ uint64_t pbwt_fetch_column(uint64_t const *__restrict__ paths_by_edge, uint32_t words_per_edge, uint32_t edge, uint32_t edge_limit, int t)
{
	if (0u == edge_limit) return 0;
	uint32_t const e = edge < edge_limit ? edge : edge_limit - 1u;
	return (uint32_t) t < words_per_edge ? paths_by_edge[(uint64_t) e * words_per_edge + (uint32_t) t] : 0;
}

struct pbwt_column_stream {
	uint64_t next_word;   // thread t's word of the edge after the one whose words are in column[buf]
	int buf;
};

// Before the first step: edge `edge`'s words into column[0], the next edge's on their way.  The caller's next barrier publishes them.
template <int kColumnWords>
__device__ __forceinline__ pbwt_column_stream pbwt_prime_columns(uint64_t const *__restrict__ paths_by_edge, uint32_t words_per_edge, uint32_t edge, uint32_t edge_limit,
	uint64_t (*column)[kColumnWords], int t)
{
	if ((uint32_t) t < words_per_edge) column[0][t] = pbwt_fetch_column(paths_by_edge, words_per_edge, edge, edge_limit, t);
	return pbwt_column_stream{pbwt_fetch_column(paths_by_edge, words_per_edge, edge + 1u, edge_limit, t), 0};
}

// One step of Durbin's algorithm 2 for edge `edge` (pbwt.hh:77-134) by the whole workgroup, in place: every thread has its copies in registers
// before the step's first barrier and writes them to their new places after it.  Thread t owns `my_count` <= kPer <= kPbwtPerThread consecutive copies of the order from
// my_begin on (kPer = ceil(copies / 1024) is a template parameter of the kernels: the per-copy loops unroll without a branch or a dead slot).  Returns how many copies do NOT use the edge (workgroup-uniform).  TWO barriers; the
// state is updated in place.
//   pass 1  the thread's copies, their divergence values and their bits of the edge into registers; what the run does to the zero
//           count and to the two running maxima, folded locally; an inclusive scan of that over the wave (DPP)      -- barrier --
//   pass 2  the 16 waves' totals scanned by every wave for itself (one LDS read + a row scan), the lane's exclusive prefix from its
//           neighbour (DPP), then every copy placed (stable partition) with its new divergence value; the next edge's column
//           stashed                                                                                                  -- barrier --
// Round 3's form of this step took 5 us at config 4 (three barriers, ds_bpermute scans over five registers, the 16 wave totals
// combined serially by all 1024 threads, every copy read twice).
template <int kPer>
__device__ __forceinline__ uint32_t pbwt_step(
	uint64_t const *__restrict__ paths_by_edge, uint32_t words_per_edge, uint32_t edge, uint32_t edge_limit, pbwt_column_stream &cols,
	unsigned short *order, uint32_t *divergence, uint64_t (*column)[kPer * (kPbwtThreads / 64)], uint4 *wave_items,
	uint32_t my_begin, uint32_t my_count, int t, int lane, int wave)
{
	constexpr uint32_t per = kPer;
	unsigned short const *const ord = order;
	uint32_t const *const dv = divergence;
	uint64_t const *const col = column[cols.buf];
	// A slot past the thread's last copy (only the last threads have any) counts as a copy of class 1 with divergence 0: it adds no zero,
	// leaves p as it is, and what it does to q only reaches threads that hold no copy at all.  So nothing below selects on validity but the
	// two stores of pass 2.
	uint32_t const *const col32 = reinterpret_cast<uint32_t const *>(col);
	uint32_t copy[kPer], d[kPer];
#pragma unroll
	for (int k = 0; k < kPer; ++k) {
		if ((uint32_t) k < per) {                                           // (uniform)
			copy[k] = ord[my_begin + k];                                     // (my_begin + k < 1024 * per = the arrays' size: inside either way)
			d[k] = dv[my_begin + k];
		}
	}
	uint32_t flags = 0;
	uint32_t chain_p = 0u, chain_q = 0u;
#pragma unroll
	for (int k = 0; k < kPer; ++k) {
		if ((uint32_t) k < per) {
			bool const valid = (uint32_t) k < my_count;
			uint32_t const c = valid ? copy[k] : 0u;
			uint32_t const bit = (col32[c >> 5] >> (c & 31u)) & 1u;
			uint32_t const f = valid ? bit : 1u;
			d[k] = valid ? d[k] : 0u;
			flags |= f << k;
			// a copy of class 1 folds its value into p and leaves biased 0 (= 1) behind in q; one of class 0 the other way round
			uint32_t const acc = f ? chain_p : chain_q;
			uint32_t const keep = (acc & kChainConst) | d[k];
			uint32_t const grown = acc > keep ? acc : keep;                   // chain_then(acc, d): d is never a constant
			chain_p = f ? grown : (kChainConst | 1u);
			chain_q = f ? (kChainConst | 1u) : grown;
		}
	}
	pbwt_scan_item const mine{per - (uint32_t) __builtin_popcount(flags), chain_p, chain_q};
	pbwt_scan_item const incl = wave_inclusive_scan(mine);
	if (lane == 63) wave_items[wave] = uint4{incl.zeros, incl.p, incl.q, 0u};
	pbwt_block_sync();

	// the waves' totals: every wave scans the 16 of them for itself, in each of its rows of 16 lanes
	uint4 const wi = wave_items[lane & (kPbwtWaves - 1)];
	pbwt_scan_item const waves_incl = row_inclusive_scan(pbwt_scan_item{wi.x, wi.y, wi.z});
	uint32_t const zeros_total = (uint32_t) __builtin_amdgcn_readlane((int) waves_incl.zeros, kPbwtWaves - 1);
	int const w = __builtin_amdgcn_readfirstlane(wave);
	pbwt_scan_item before{0u, 0u, 0u};                                       // identity: no copies
	if (w > 0) {
		before.zeros = (uint32_t) __builtin_amdgcn_readlane((int) waves_incl.zeros, w - 1);
		before.p = (uint32_t) __builtin_amdgcn_readlane((int) waves_incl.p, w - 1);
		before.q = (uint32_t) __builtin_amdgcn_readlane((int) waves_incl.q, w - 1);
	}
	// ... and the lanes before this one in its wave (lane 0: nothing = the identity)
	pbwt_scan_item const lanes_before{pbwt_dpp<kDppWaveShr1, 0xf>(incl.zeros), pbwt_dpp<kDppWaveShr1, 0xf>(incl.p), pbwt_dpp<kDppWaveShr1, 0xf>(incl.q)};
	before = scan_combine(before, lanes_before);

	// second pass: place the copies (stable partition) with their new divergence values
	uint32_t p = chain_apply(before.p, edge + 2u), q = chain_apply(before.q, edge + 2u);   // biased edge + 1 (pbwt.hh:93)
	uint32_t zero_at = before.zeros, one_at = zeros_total + (my_begin - before.zeros);
	unsigned short *const out_ord = order;                              // (everybody's reads lie before the barrier above)
	uint32_t *const out_dv = divergence;
#pragma unroll
	for (int k = 0; k < kPer; ++k) {
		if ((uint32_t) k < per) {
			uint32_t const fk = (flags >> k) & 1u;
			bool const f = 0u != fk;
			p = p > d[k] ? p : d[k];
			q = q > d[k] ? q : d[k];
			uint32_t const at = f ? one_at : zero_at;
			uint32_t const value = f ? q : p;
			if ((uint32_t) k < my_count) { out_ord[at] = (unsigned short) copy[k]; out_dv[at] = value; }
			one_at += fk;
			zero_at += fk ^ 1u;
			p = f ? p : 1u;
			q = f ? 1u : q;
		}
	}
	// the next edge's column into the other buffer (nobody reads that one during this step), the one after it on its way
	if ((uint32_t) t < words_per_edge) column[cols.buf ^ 1][t] = cols.next_word;
	cols.next_word = pbwt_fetch_column(paths_by_edge, words_per_edge, edge + 2u, edge_limit, t);
	cols.buf ^= 1;
	pbwt_block_sync();
	return zeros_total;
}

// Puts `count` copies into the bin's slot of the candidate's hash table (claiming a slot for a bin seen for the first time).
template <int kSlots>
__device__ __forceinline__ void pbwt_bin_add(uint32_t *hash_key, uint32_t *hash_count, uint32_t *bin_slot, uint32_t *n_bins_s, uint32_t *failed_s, uint32_t bin, uint32_t count)
{
	uint32_t slot = (bin * 2654435761u) >> 20 & (uint32_t) (kSlots - 1);
	bool placed = false;
	for (int probe = 0; probe < kSlots && !placed; ++probe) {   // (bounded: a full table ends the chunk)
		uint32_t const seen = atomicCAS(&hash_key[slot], 0u, bin + 1u);
		if (0u == seen) {                                       // claimed a free slot
			uint32_t const k = atomicAdd(n_bins_s, 1u);
			if (k < (uint32_t) kPbwtMaxBins) bin_slot[k] = slot;
			placed = true;
		} else if (seen == bin + 1u) {
			placed = true;
		} else {
			slot = (slot + 1u) & (uint32_t) (kSlots - 1);
		}
	}
	if (placed) atomicAdd(&hash_count[slot], count);
	else *failed_s = 1u;
}

// One workgroup per chunk.  See v2m_pbwt_cut_trials() in include/v2m_hip.h for the arguments.  kPer = ceil(n_copies / kPbwtThreads).
template <int kPer>
__global__ __launch_bounds__(kPbwtThreads) void pbwt_cut_trials_kernel(
	uint64_t const *__restrict__ paths_by_edge,          // edge-major bits: column e = words [e * words_per_edge, +words_per_edge), bit c = copy c
	uint32_t words_per_edge, uint32_t n_copies, uint32_t n_edges,
	uint32_t const *__restrict__ first_candidate_from_edge,   // [n_edges + 1]: first candidate whose edge index is >= e
	uint32_t const *__restrict__ cand_edge, uint64_t const *__restrict__ cand_aligned, uint64_t min_distance,
	uint64_t const *__restrict__ chunk_first,             // [n_chunks + 1] candidate indices
	uint32_t const *__restrict__ start_order, uint32_t const *__restrict__ start_divergence,   // [n_chunks][n_copies]
	uint64_t trial_capacity, uint32_t *__restrict__ trial_pred, uint32_t *__restrict__ trial_class_count,   // [n_chunks][trial_capacity]
	uint64_t *__restrict__ trial_end,                     // [n_candidates]: trials of the chunk up to and including this candidate
	uint32_t *__restrict__ chunk_status)                  // [n_chunks]: 0 = done, 1 = left to the host (too many bins / trials)
{
	constexpr int kCopies = kPer * kPbwtThreads, kHashSlots = pbwt_hash_slots<kPer>();
	__shared__ unsigned short order[kCopies];
	__shared__ uint32_t divergence[kCopies];
	__shared__ uint64_t column[2][kCopies / 64];
	__shared__ uint4 wave_items[kPbwtWaves];
	__shared__ uint32_t hash_key[kHashSlots];             // bin + 1; 0 = free
	__shared__ uint32_t hash_count[kHashSlots];
	__shared__ uint32_t bin_slot[kPbwtMaxBins];           // hash slots in use, in claiming order
	__shared__ __attribute__((aligned(16))) uint32_t bin_key[kPbwtMaxBins], bin_packed[kPbwtMaxBins];   // the candidate's bins, dense: key; copies | emits << 16
	__shared__ uint32_t bin_sum[kPbwtMaxBins];            // per bin: the same two fields summed over the bins with a larger key
	__shared__ uint32_t n_bins_s, reduce_max[kPbwtWaves], reduce_cnt[kPbwtWaves], failed_s, emitted_s, smallest_s;

	int const t = threadIdx.x, lane = t & 63, wave = t >> 6;
	uint32_t const chunk = blockIdx.x;
	uint64_t const cand_begin = chunk_first[chunk], cand_end = chunk_first[chunk + 1];
	constexpr uint32_t per = kPer;                                              // ceil(n_copies / kPbwtThreads) <= kPbwtPerThread (the host picks the instantiation)
	uint32_t const my_begin = (uint32_t) t * per < n_copies ? (uint32_t) t * per : n_copies;
	uint32_t const my_end = my_begin + per < n_copies ? my_begin + per : n_copies;
	uint32_t const my_count = my_end - my_begin;

	for (uint32_t i = t; i < n_copies; i += kPbwtThreads) {
		order[i] = (unsigned short) start_order[(uint64_t) chunk * n_copies + i];
		divergence[i] = start_divergence[(uint64_t) chunk * n_copies + i];
	}
	for (uint32_t i = t; i < (uint32_t) kHashSlots; i += kPbwtThreads) { hash_key[i] = 0; hash_count[i] = 0; }
	if (t == 0) { n_bins_s = 0; failed_s = 0; emitted_s = 0; smallest_s = 0xFFFFFFFFu; }
	uint32_t edge = cand_begin < cand_end ? cand_edge[cand_begin] : 0;
	pbwt_column_stream cols = pbwt_prime_columns<kCopies / 64>(paths_by_edge, words_per_edge, edge, n_edges, column, t);
	pbwt_block_sync();

	uint64_t n_trials = 0;                                 // (kept by every thread: all of them see the same counts)
	uint32_t *const my_pred = trial_pred + (uint64_t) chunk * trial_capacity;
	uint32_t *const my_class = trial_class_count + (uint64_t) chunk * trial_capacity;

	for (uint64_t cand = cand_begin; cand < cand_end; ++cand) {
		// ---- the edges before this candidate's node (find_cut_positions.cc:170-176 over pbwt.hh:77-134) -----------------
		uint32_t const upto = cand_edge[cand];
		for (; edge < upto; ++edge) {
			pbwt_step<kPer>(paths_by_edge, words_per_edge, edge, n_edges, cols, order, divergence, column, wave_items, my_begin, my_count, t, lane, wave);
		}

		// ---- the candidate (find_cut_positions.cc:134-165) ---------------------------------------------------------------
		uint32_t const next = (uint32_t) cand;
		uint32_t const *const dv = divergence;
		// the largest divergence value and how many copies hold it
		uint32_t d[kPer];
		uint32_t my_max = 0;
#pragma unroll
		for (int k = 0; k < kPer; ++k) {
			if ((uint32_t) k < per) {
				d[k] = (uint32_t) k < my_count ? dv[my_begin + k] : 0u;
				my_max = umax(my_max, d[k]);
			}
		}
		my_max = wave_inclusive_max(my_max);
		if (lane == 63) reduce_max[wave] = my_max;
		pbwt_block_sync();
		uint32_t const d_max = (uint32_t) __builtin_amdgcn_readlane((int) row_inclusive_max(reduce_max[lane & (kPbwtWaves - 1)]), kPbwtWaves - 1);
		// every other copy goes into the bin of the candidate its value points to (clipped to next + 1: "no earlier candidate");
		// biased 0 ("no match yet") is the SMALLEST value: the walk reaches it last, it points to no candidate, and the only
		// count that includes it is the final one (all copies): it takes no part in the bins
		uint32_t my_cnt = 0;
		uint32_t bin[kPer];
#pragma unroll
		for (int k = 0; k < kPer; ++k) {
			if ((uint32_t) k < per) {
				bool const valid = (uint32_t) k < my_count;
				bool const binned = valid && 0u != d[k] && d[k] != d_max;
				my_cnt += (valid && d[k] == d_max) ? 1u : 0u;
				uint32_t c = next + 1u;
				if (binned && d[k] - 1u <= n_edges) c = first_candidate_from_edge[d[k] - 1u];   // (all of a thread's loads go out before any is used)
				bin[k] = binned ? (c < next + 1u ? c : next + 1u) : 0xFFFFFFFFu;
			}
		}
		// About a hundred bins per candidate at config 4, a dozen of which hold most of the 5008 copies.  Measured there, against this
		// plain loop (30.5 ms for the kernel): counting a wave's most frequent bins with ballots first and adding them once, 34.5 ms
		// with one such round, 36.6 with two, 43.9 with four; a plain read of the key before the compare-and-swap, 31.9; all of a
		// thread's lookups issued together, 33.3; without any insertion the kernel takes 17.9 ms, its pBWT steps alone 11.5
		// (profiles/r04/founder_kernels_what_bounds_them.txt).
#pragma unroll
		for (int k = 0; k < kPer; ++k) {
			if ((uint32_t) k < per) {
				if (0xFFFFFFFFu != bin[k]) pbwt_bin_add<kHashSlots>(hash_key, hash_count, bin_slot, &n_bins_s, &failed_s, bin[k], 1u);
			}
		}
		my_cnt = wave_inclusive_add(my_cnt);
		if (lane == 63) reduce_cnt[wave] = my_cnt;
		pbwt_block_sync();
		uint32_t const count_max = (uint32_t) __builtin_amdgcn_readlane((int) row_inclusive_add(reduce_cnt[lane & (kPbwtWaves - 1)]), kPbwtWaves - 1);
		uint32_t const n_bins = n_bins_s;
		if (n_bins > (uint32_t) kPbwtMaxBins || failed_s) {                        // (workgroup-uniform) more bins than this kernel holds: the host takes the chunk
			if (t == 0) chunk_status[chunk] = 1u;
			return;
		}
		// The trials (find_cut_positions.cc:139-160): the bins in descending key order, bin r tried with class_count = copies in
		// the bins before it (larger values) + the largest value's copies, unless it is the clipped bin, the candidate itself, or
		// too close.  Config 4 has several hundred bins per candidate, so nothing here may walk the bins one by one (round 3 ranked
		// them with a dependent LDS loop per thread and summed the counts with two more: ~14 us per candidate).  What bin i needs --
		// the copies and the number of EMITTING bins among the bins with a larger key -- is one sum over the pairs (i, j) with
		// key_j > key_i, spread over all 1024 threads; no sort, no scan: a trial's slot is that number.
		// phase D: the table's entries into dense arrays (bin_key / bin_packed: copies | emits << 16), the table left empty
		uint32_t my_key = 0, my_packed = 0;
		bool const have = (uint32_t) t < n_bins;

		if (have) {
			uint32_t const slot = bin_slot[t];
			my_key = hash_key[slot] - 1u;
			uint32_t const cnt = hash_count[slot];
			hash_key[slot] = 0;                                      // leave the table empty for the next candidate
			hash_count[slot] = 0;
			bool const emits = my_key != next + 1u && my_key != next && min_distance <= cand_aligned[next] - cand_aligned[my_key];
			my_packed = cnt | (emits ? 0x10000u : 0u);
		}
		if ((uint32_t) t < ((n_bins + 3u) & ~3u)) { bin_key[t] = my_key; bin_packed[t] = my_packed; bin_sum[t] = 0u; }   // (padded to whole groups of four with key 0: larger than nothing)
		{
			uint32_t const emits_wave = wave_inclusive_add(my_packed >> 16);
			uint32_t smallest = have ? my_key : 0xFFFFFFFFu;
			smallest = ~wave_inclusive_max(~smallest);                   // (a running minimum)
			if (lane == 63 && emits_wave) atomicAdd(&emitted_s, emits_wave);
			if (lane == 63 && 0xFFFFFFFFu != smallest) atomicMin(&smallest_s, smallest);
		}
		pbwt_block_sync();
		// phase R: thread t = (bin i, part): the pairs (i, j) with j in the part's slice of the bins
		{
			uint32_t n_pad = 64u;
			while (n_pad < n_bins) n_pad <<= 1;                                 // 64 .. 1024: a wave's lanes share their slice (LDS broadcasts)
			uint32_t const parts = (uint32_t) kPbwtThreads / n_pad;          // (n_bins <= kPbwtMaxBins = the threads: at least one)
			uint32_t const i = (uint32_t) t & (n_pad - 1u), part = (uint32_t) t / n_pad;
			uint32_t const groups = (n_bins + 3u) / 4u, per_part = (groups + parts - 1u) / parts;
			uint32_t const g0 = part * per_part, g1 = g0 + per_part < groups ? g0 + per_part : groups;
			if (i < n_bins && g0 < g1) {
				uint32_t const key_i = bin_key[i];
				uint32_t acc = 0;
				for (uint32_t g = g0; g < g1; ++g) {
					uint4 const kj = reinterpret_cast<uint4 const *>(bin_key)[g], pj = reinterpret_cast<uint4 const *>(bin_packed)[g];
					acc += kj.x > key_i ? pj.x : 0u;
					acc += kj.y > key_i ? pj.y : 0u;
					acc += kj.z > key_i ? pj.z : 0u;
					acc += kj.w > key_i ? pj.w : 0u;
				}
				if (acc) atomicAdd(&bin_sum[i], acc);
			}
		}
		pbwt_block_sync();
		// phase W: the trials
		{
			uint32_t const emitted = emitted_s;
			// after the loop: the segment may reach further left still (find_cut_positions.cc:162-165)
			uint32_t right_bound = next + 1u;
			if (n_bins) right_bound = smallest_s < right_bound ? smallest_s : right_bound;
			bool const last_trial = 0u != right_bound && right_bound - 1u != next;
			if (n_trials + emitted + (last_trial ? 1u : 0u) > trial_capacity) {   // (workgroup-uniform) the trials do not fit: the host takes the chunk
				if (t == 0) chunk_status[chunk] = 1u;
				return;
			}
			if (have && (my_packed >> 16)) {
				uint32_t const before = bin_sum[t];                              // copies | emitting bins << 16 of the bins with a larger key
				my_pred[n_trials + (before >> 16)] = my_key;
				my_class[n_trials + (before >> 16)] = count_max + (before & 0xFFFFu);
			}
			n_trials += emitted;
			if (last_trial) {
				if (t == 0) { my_pred[n_trials] = right_bound - 1u; my_class[n_trials] = n_copies; }
				++n_trials;
			}
			if (t == 0) trial_end[cand] = n_trials;
		}
		pbwt_block_sync();                                                // bin_* and the two totals are rewritten by the next candidate
		if (t == 0) { n_bins_s = 0; emitted_s = 0; smallest_s = 0xFFFFFFFFu; }   // (read again only behind the next candidate's barriers)
	}
	if (t == 0) chunk_status[chunk] = 0u;
}


// ---------------------------------------------------------------------------------------------------------------------
// The same walk for find_matchings (founder_sequence_greedy_output.cc:154-512): at every cut position the path classes of the
// block that ends there and of the two-block span that ends there (:215-251).  A class starts at every copy (in pBWT order)
// whose divergence value lies past the block's / the span's first edge; the host's loop is a running "last such copy" plus a
// run-length count, i.e. a max-scan and a compaction.  The greedy assignment itself (:254-457) is strictly sequential and
// stays on the host, which also sorts each cut's joined classes (std::sort on the same input in the same order as the
// reference, so ties fall the same way).
// ---------------------------------------------------------------------------------------------------------------------
constexpr uint32_t kPbwtNoClass = 0xFFFFu;            // the host's kPloidyMax in the 16-bit class arrays

struct class_scan_item {
	uint32_t last_block_start;   // 1 + the last index with a block-class boundary, 0 = none yet (so that all-zero is the identity)
	uint32_t block_starts;       // boundaries of the block (prev_cut_edge < d)
	uint32_t span_starts;        // boundaries of the two-block span (cut_pair_edge < d)
};

__device__ __forceinline__ class_scan_item class_combine(class_scan_item const &a, class_scan_item const &b)
{
	return class_scan_item{umax(a.last_block_start, b.last_block_start), a.block_starts + b.block_starts, a.span_starts + b.span_starts};
}

template <int kCtrl, int kRowMask>
__device__ __forceinline__ class_scan_item class_round(class_scan_item const &x)
{
	class_scan_item const up{pbwt_dpp<kCtrl, kRowMask>(x.last_block_start), pbwt_dpp<kCtrl, kRowMask>(x.block_starts), pbwt_dpp<kCtrl, kRowMask>(x.span_starts)};
	return class_combine(up, x);
}

__device__ __forceinline__ class_scan_item class_row_scan(class_scan_item x)
{
	x = class_round<kDppRowShr1, 0xf>(x);
	x = class_round<kDppRowShr2, 0xf>(x);
	x = class_round<kDppRowShr4, 0xf>(x);
	x = class_round<kDppRowShr8, 0xf>(x);
	return x;
}

__device__ __forceinline__ class_scan_item class_wave_scan(class_scan_item x)
{
	x = class_row_scan(x);
	x = class_round<kDppBcast15, 0xa>(x);
	x = class_round<kDppBcast31, 0xc>(x);
	return x;
}

// "threshold < unbiased(d)" with d biased: 0 is the reference's DIVERGENCE_MAX ("no match yet": starts a class)
__device__ __forceinline__ bool past_edge(uint32_t d, uint32_t threshold) { return 0u == d || d - 1u > threshold; }

template <int kPer>
__global__ __launch_bounds__(kPbwtThreads) void pbwt_cut_records_kernel(
	uint64_t const *__restrict__ paths_by_edge, uint32_t words_per_edge, uint32_t n_copies, uint32_t n_edge_columns /* columns of paths_by_edge (>= every edge visited + 1 is not required: fetches are clamped) */,
	uint32_t const *__restrict__ cut_edge,                // [n_cuts]: edges before each cut node
	uint64_t const *__restrict__ chunk_first_cut,         // [n_chunks + 1]: chunk k handles the cuts [first[k], first[k + 1]); first[k] >= 1
	uint32_t const *__restrict__ start_edge,              // [n_chunks]: the given state is the one after this many edges (<= cut_edge[first[k] - 1])
	uint32_t const *__restrict__ start_order, uint32_t const *__restrict__ start_divergence,   // [n_chunks][n_copies]
	uint64_t pool_capacity, uint32_t *__restrict__ pool_lhs, uint32_t *__restrict__ pool_rhs, uint32_t *__restrict__ pool_size,   // [n_chunks][pool_capacity]
	uint64_t *__restrict__ rec_pool_end,                  // [n_cuts]: joined classes of the chunk up to and including this cut
	uint32_t *__restrict__ rec_distinct, uint32_t *__restrict__ rec_first_class, uint32_t *__restrict__ rec_first_is_ref,   // [n_cuts]
	uint32_t *__restrict__ chunk_status,
	unsigned short *__restrict__ class_scratch)           // [n_chunks][kPbwtClassScratchArrays][1024 * kPer] when the instantiation keeps its class arrays in global memory, else unused
{
	static_assert(kPer <= kPbwtPerThreadRecords, "one instantiation per copies-per-thread count up to the cut search's");
	constexpr int kCopies = kPer * kPbwtThreads;
	constexpr bool kInLds = kPbwtClassesInLds<kPer>;
	__shared__ unsigned short order[kCopies];
	__shared__ uint32_t divergence[kCopies];
	__shared__ uint64_t column[2][kCopies / 64];
	__shared__ uint4 wave_items[kPbwtWaves];
	__shared__ uint4 class_items[kPbwtWaves];
	// per copy: the representative of its class at the last / the previous cut; per joined class: where it starts in the order.  Written and read
	// once per cut (scattered by copy number), never inside a pBWT step: beyond 12 288 copies they live in global memory (L2-resident: 120 KB per
	// workgroup), where the same barriers order them (a workgroup's waves share their CU's L1).
	__shared__ unsigned short copy_class_lds[kInLds ? 2 : 1][kInLds ? kCopies : 1];
	__shared__ unsigned short span_start_index_lds[kInLds ? kCopies : 1];
	unsigned short *const scratch = kInLds ? nullptr : class_scratch + (uint64_t) blockIdx.x * kPbwtClassScratchArrays * kCopies;
	auto const copy_class = [&](int which) -> unsigned short * {
		if constexpr (kInLds) return copy_class_lds[which];
		else return scratch + (uint64_t) which * kCopies;
	};
	unsigned short *span_start_index;
	if constexpr (kInLds) span_start_index = span_start_index_lds;
	else span_start_index = scratch + 2 * kCopies;

	int const t = threadIdx.x, lane = t & 63, wave = t >> 6;
	uint32_t const chunk = blockIdx.x;
	uint64_t const cut_begin = chunk_first_cut[chunk], cut_end = chunk_first_cut[chunk + 1];
	if (cut_begin >= cut_end) { if (t == 0) chunk_status[chunk] = 0u; return; }
	constexpr uint32_t per = kPer;
	uint32_t const my_begin = (uint32_t) t * per < n_copies ? (uint32_t) t * per : n_copies;
	uint32_t const my_end = my_begin + per < n_copies ? my_begin + per : n_copies;
	uint32_t const my_count = my_end - my_begin;

	for (uint32_t i = t; i < n_copies; i += kPbwtThreads) {
		order[i] = (unsigned short) start_order[(uint64_t) chunk * n_copies + i];
		divergence[i] = start_divergence[(uint64_t) chunk * n_copies + i];
		copy_class(0)[i] = (unsigned short) kPbwtNoClass;
		copy_class(1)[i] = (unsigned short) kPbwtNoClass;
	}
	int rhs = 0;                                           // copy_class(rhs): the classes the last cut left behind
	uint32_t edge = start_edge[chunk];
	pbwt_column_stream cols = pbwt_prime_columns<kCopies / 64>(paths_by_edge, words_per_edge, edge, n_edge_columns, column, t);
	pbwt_block_sync();

	// up to the cut before the chunk's first one; the classes it left behind (founder.cc:scan_cut_chunk)
	uint64_t const start_cut = cut_begin - 1;
	for (uint32_t const upto = cut_edge[start_cut]; edge < upto; ++edge) {
		pbwt_step<kPer>(paths_by_edge, words_per_edge, edge, n_edge_columns, cols, order, divergence, column, wave_items, my_begin, my_count, t, lane, wave);
	}

	// The classes of a threshold: every copy's representative = the copy at the last boundary at or before it.
	// with_span: also the joined classes of the two-block span, written to the pool.  Returns false when the pool is full.
	uint64_t n_pool = 0;
	uint32_t *const my_lhs = pool_lhs + (uint64_t) chunk * pool_capacity, *const my_rhs = pool_rhs + (uint64_t) chunk * pool_capacity, *const my_size = pool_size + (uint64_t) chunk * pool_capacity;
	// Two barriers.  (What a call leaves in class_items / span_start_index / copy_class is only rewritten behind the first barrier of
	// whatever runs next -- a pBWT step or another call -- and every read of it here lies before this call's last one.)
	auto const classes_at_cut = [&](uint32_t block_threshold, bool with_span, uint32_t span_threshold, uint32_t &distinct_out) -> bool {
		unsigned short const *const ord = order;
		uint32_t const *const dv = divergence;
		uint32_t copy[kPer], d[kPer];
		class_scan_item mine{0u, 0u, 0u};
#pragma unroll
		for (int k = 0; k < kPer; ++k) {
			if ((uint32_t) k < per) {                                       // (uniform)
				bool const valid = (uint32_t) k < my_count;
				copy[k] = ord[my_begin + k];
				d[k] = dv[my_begin + k];
				if (valid && past_edge(d[k], block_threshold)) { mine.last_block_start = my_begin + (uint32_t) k + 1u; ++mine.block_starts; }
				if (valid && with_span && past_edge(d[k], span_threshold)) ++mine.span_starts;
			}
		}
		class_scan_item const incl = class_wave_scan(mine);
		if (lane == 63) class_items[wave] = uint4{incl.last_block_start, incl.block_starts, incl.span_starts, 0u};
		pbwt_block_sync();
		uint4 const wi = class_items[lane & (kPbwtWaves - 1)];
		class_scan_item const waves_incl = class_row_scan(class_scan_item{wi.x, wi.y, wi.z});
		class_scan_item const total{0u, (uint32_t) __builtin_amdgcn_readlane((int) waves_incl.block_starts, kPbwtWaves - 1), (uint32_t) __builtin_amdgcn_readlane((int) waves_incl.span_starts, kPbwtWaves - 1)};
		int const w = __builtin_amdgcn_readfirstlane(wave);
		class_scan_item before{0u, 0u, 0u};
		if (w > 0) {
			before.last_block_start = (uint32_t) __builtin_amdgcn_readlane((int) waves_incl.last_block_start, w - 1);
			before.block_starts = (uint32_t) __builtin_amdgcn_readlane((int) waves_incl.block_starts, w - 1);
			before.span_starts = (uint32_t) __builtin_amdgcn_readlane((int) waves_incl.span_starts, w - 1);
		}
		class_scan_item const lanes_before{pbwt_dpp<kDppWaveShr1, 0xf>(incl.last_block_start), pbwt_dpp<kDppWaveShr1, 0xf>(incl.block_starts), pbwt_dpp<kDppWaveShr1, 0xf>(incl.span_starts)};
		before = class_combine(before, lanes_before);
		distinct_out = total.block_starts;
		bool const fits = !with_span || n_pool + total.span_starts <= pool_capacity;     // (workgroup-uniform)
		// second pass: representatives, class arrays, joined-class heads
		uint32_t last = before.last_block_start;                       // 1 + index, 0 = none
		uint32_t span_at = before.span_starts;
		unsigned short const *const lhs_class = copy_class(rhs);       // (the previous cut's classes: read)
		unsigned short *const rhs_class = copy_class(rhs ^ 1);         // (this cut's: written; the arrays swap roles below)
#pragma unroll
		for (int k = 0; k < kPer; ++k) {
			if ((uint32_t) k < per) {
				bool const valid = (uint32_t) k < my_count;
				if (valid) {
					if (past_edge(d[k], block_threshold)) last = my_begin + (uint32_t) k + 1u;
					uint32_t const rep = last ? (uint32_t) ord[last - 1u] : kPbwtNoClass;
					if (with_span && fits && past_edge(d[k], span_threshold)) {
						uint32_t const l = lhs_class[copy[k]];
						my_lhs[n_pool + span_at] = kPbwtNoClass == l ? 0xFFFFFFFFu : l;
						my_rhs[n_pool + span_at] = kPbwtNoClass == rep ? 0xFFFFFFFFu : rep;
						span_start_index[span_at] = (unsigned short) (my_begin + (uint32_t) k);
						++span_at;
					}
					rhs_class[copy[k]] = (unsigned short) rep;
				}
			}
		}
		pbwt_block_sync();
		if (with_span && fits) {
			for (uint32_t j = t; j < total.span_starts; j += kPbwtThreads)
				my_size[n_pool + j] = (j + 1 < total.span_starts ? (uint32_t) span_start_index[j + 1] : n_copies) - (uint32_t) span_start_index[j];
			n_pool += total.span_starts;
		}
		rhs ^= 1;
		return fits;
	};

	if (start_cut >= 1) {
		uint32_t ignored;
		classes_at_cut(cut_edge[start_cut - 1], false, 0u, ignored);
	}

	bool first_is_ref = true;
	for (uint64_t cut = cut_begin; cut < cut_end; ++cut) {
		for (uint32_t const upto = cut_edge[cut]; edge < upto; ++edge) {
			uint32_t const zeros = pbwt_step<kPer>(paths_by_edge, words_per_edge, edge, n_edge_columns, cols, order, divergence, column, wave_items, my_begin, my_count, t, lane, wave);
			// the copy that is first in the order now uses the edge exactly when no copy does not (:454-462)
			first_is_ref = first_is_ref && 0u != zeros;
		}
		uint32_t distinct = 0;
		bool const fits = classes_at_cut(cut_edge[cut - 1], cut >= 2, cut >= 2 ? cut_edge[cut - 2] : 0u, distinct);
		if (!fits) {                                                    // (workgroup-uniform) the pool is full: the host takes the chunk
			if (t == 0) chunk_status[chunk] = 1u;
			return;
		}
		if (t == 0) {
			rec_pool_end[cut] = n_pool;
			rec_distinct[cut] = distinct;
			rec_first_class[cut] = order[0];
			rec_first_is_ref[cut] = first_is_ref ? 1u : 0u;
		}
		first_is_ref = true;
	}
	if (t == 0) chunk_status[chunk] = 0u;
}

// first_candidate_from_edge[e] = first candidate whose edge index is >= e (what std::lower_bound over the whole list returns,
// find_cut_positions.cc:141): candidate c covers the edges (cand_edge[c - 1], cand_edge[c]].
__global__ __launch_bounds__(256) void pbwt_first_candidate_kernel(uint32_t const *__restrict__ cand_edge, uint32_t n_candidates, uint32_t n_edges, uint32_t *__restrict__ out)
{
	uint32_t const e = blockIdx.x * 256u + threadIdx.x;
	if (e > n_edges) return;
	uint32_t lo = 0, hi = n_candidates;                               // first c with cand_edge[c] >= e
	while (lo < hi) { uint32_t const mid = (lo + hi) / 2; if (cand_edge[mid] < e) lo = mid + 1; else hi = mid; }
	out[e] = lo;
}

} // namespace v2m
