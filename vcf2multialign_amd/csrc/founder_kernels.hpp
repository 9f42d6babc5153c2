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
//               hash), sort the few dozen distinct bins, prefix-sum their counts
//
// Divergence values are BIASED by one as on the host (0 = "no match yet", founder.cc), so everything is plain u32 order.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace v2m {

constexpr int kPbwtThreads = 1024;
constexpr int kPbwtWaves = kPbwtThreads / 64;
constexpr int kPbwtMaxCopies = 8192;                  // chromosome copies a workgroup can walk (LDS-resident state)
constexpr int kPbwtPerThread = kPbwtMaxCopies / kPbwtThreads;
constexpr int kPbwtHashSlots = 4096;                  // distinct candidate bins per candidate node: far fewer in practice
constexpr int kPbwtMaxBins = 1024;                    // more distinct bins than this at one candidate: the chunk is left to the host

// __syncthreads() with the LDS wait spelled out: on loop back edges hipcc (ROCm 7.2) has emitted the barrier without the
// s_waitcnt lgkmcnt(0) that __syncthreads() implies (kernels.hpp, ring transpose; tests/test_kernel_isa.py checks every
// barrier of every kernel), and these kernels are all loops around barriers.
__device__ __forceinline__ void pbwt_block_sync()
{
	asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	__syncthreads();
}

// "p -> is_const ? value : max(p, value)": what a run of copies does to a running maximum that is reset after every copy of
// its own class.
struct max_chain {
	uint32_t value;
	uint32_t is_const;
};

// first `then`, after that `next`
__device__ __forceinline__ max_chain chain_then(max_chain first, max_chain next)
{
	if (next.is_const) return next;
	return max_chain{first.value > next.value ? first.value : next.value, first.is_const};
}

__device__ __forceinline__ uint32_t chain_apply(max_chain c, uint32_t p) { return c.is_const ? c.value : (p > c.value ? p : c.value); }

struct pbwt_scan_item {
	uint32_t zeros;       // copies that do not use the edge
	max_chain p, q;       // the running maxima of the two classes (pbwt.hh:93-131)
};

__device__ __forceinline__ pbwt_scan_item scan_combine(pbwt_scan_item const &a, pbwt_scan_item const &b)   // a, then b
{
	return pbwt_scan_item{a.zeros + b.zeros, chain_then(a.p, b.p), chain_then(a.q, b.q)};
}

__device__ __forceinline__ pbwt_scan_item scan_shfl_up(pbwt_scan_item const &x, int delta)
{
	pbwt_scan_item y;
	y.zeros = __shfl_up(x.zeros, delta, 64);
	y.p.value = __shfl_up(x.p.value, delta, 64);
	y.p.is_const = __shfl_up(x.p.is_const, delta, 64);
	y.q.value = __shfl_up(x.q.value, delta, 64);
	y.q.is_const = __shfl_up(x.q.is_const, delta, 64);
	return y;
}

// One step of Durbin's algorithm 2 for edge `edge` (pbwt.hh:77-134) by the whole workgroup: state order[cur] / divergence[cur]
// -> order[cur ^ 1] / divergence[cur ^ 1].  Thread t owns the copies [my_begin, my_end) of the order.  Returns how many copies do
// NOT use the edge (workgroup-uniform).  Contains three barriers; the caller flips `cur`.
// The edge's bit column (words_per_edge <= 128 words, one per thread) travels in a register: `column_word` holds thread t's word of
// THIS edge on entry -- fetched while the previous step ran, or by pbwt_fetch_column() before the first one -- and of the next
// edge on return (steps always visit consecutive edges; `edge_limit` clamps the fetch after the last one).  Fetched at the top of
// the step instead, every step stood still for one global-memory latency (config 4: 58.3 -> 56.9 ms and 28.3 -> 27.0 ms for the two
// kernels -- most of a step is barriers and dependent LDS passes, not this).
__device__ __forceinline__ uint64_t pbwt_fetch_column(uint64_t const *__restrict__ paths_by_edge, uint32_t words_per_edge, uint32_t edge, uint32_t edge_limit, int t)
{
	if (0u == edge_limit) return 0;
	uint32_t const e = edge < edge_limit ? edge : edge_limit - 1u;
	return (uint32_t) t < words_per_edge ? paths_by_edge[(uint64_t) e * words_per_edge + (uint32_t) t] : 0;
}

__device__ __forceinline__ uint32_t pbwt_step(
	uint64_t const *__restrict__ paths_by_edge, uint32_t words_per_edge, uint32_t edge, uint32_t edge_limit, uint64_t &column_word,
	unsigned short (*order)[kPbwtMaxCopies], uint32_t (*divergence)[kPbwtMaxCopies], uint64_t *column, pbwt_scan_item *wave_items,
	int cur, uint32_t my_begin, uint32_t my_end, int t, int lane, int wave)
{
	if ((uint32_t) t < words_per_edge) column[t] = column_word;
	pbwt_block_sync();
	column_word = pbwt_fetch_column(paths_by_edge, words_per_edge, edge + 1u, edge_limit, t);   // in flight while this step runs
	unsigned short const *const ord = order[cur];
	uint32_t const *const dv = divergence[cur];
	// what this thread's run of copies does to the zero count and to the two running maxima
	pbwt_scan_item mine{0u, max_chain{0u, 0u}, max_chain{0u, 0u}};
	uint32_t flags = 0;
	for (uint32_t i = my_begin; i < my_end; ++i) {
		uint32_t const copy = ord[i], d = dv[i];
		uint32_t const f = (uint32_t) (column[copy >> 6] >> (copy & 63)) & 1u;
		flags |= f << (i - my_begin);
		pbwt_scan_item one;
		one.zeros = 1u - f;
		one.p = f ? max_chain{d, 0u} : max_chain{1u, 1u};      // a copy of class 0 takes p and leaves biased 0 behind
		one.q = f ? max_chain{1u, 1u} : max_chain{d, 0u};
		mine = scan_combine(mine, one);
	}
	// exclusive scan over the threads: inside the wave, then over the waves' totals
	pbwt_scan_item incl = mine;
#pragma unroll
	for (int delta = 1; delta < 64; delta <<= 1) {
		pbwt_scan_item const up = scan_shfl_up(incl, delta);
		if (lane >= delta) incl = scan_combine(up, incl);
	}
	if (lane == 63) wave_items[wave] = incl;
	pbwt_block_sync();
	pbwt_scan_item before{0u, max_chain{0u, 0u}, max_chain{0u, 0u}};      // identity: no copies
	uint32_t zeros_total = 0;
	for (int w = 0; w < kPbwtWaves; ++w) {
		pbwt_scan_item const wi = wave_items[w];
		if (w < wave) before = scan_combine(before, wi);
		zeros_total += wi.zeros;
	}
	pbwt_scan_item const lane_before = scan_shfl_up(incl, 1);
	if (lane) before = scan_combine(before, lane_before);
	// second pass: place the copies (stable partition) with their new divergence values
	uint32_t p = chain_apply(before.p, edge + 2u), q = chain_apply(before.q, edge + 2u);   // biased edge + 1 (pbwt.hh:93)
	uint32_t zero_at = before.zeros, one_at = zeros_total + (my_begin - before.zeros);
	unsigned short *const out_ord = order[cur ^ 1];
	uint32_t *const out_dv = divergence[cur ^ 1];
	for (uint32_t i = my_begin; i < my_end; ++i) {
		uint32_t const copy = ord[i], d = dv[i];
		p = p > d ? p : d;
		q = q > d ? q : d;
		if (!((flags >> (i - my_begin)) & 1u)) { out_ord[zero_at] = (unsigned short) copy; out_dv[zero_at] = p; ++zero_at; p = 1u; }
		else { out_ord[one_at] = (unsigned short) copy; out_dv[one_at] = q; ++one_at; q = 1u; }
	}
	pbwt_block_sync();
	return zeros_total;
}

// One workgroup per chunk.  See v2m_pbwt_cut_trials() in include/v2m_hip.h for the arguments.
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
	__shared__ unsigned short order[2][kPbwtMaxCopies];
	__shared__ uint32_t divergence[2][kPbwtMaxCopies];
	__shared__ uint64_t column[kPbwtMaxCopies / 64];
	__shared__ pbwt_scan_item wave_items[kPbwtWaves];
	__shared__ uint32_t hash_key[kPbwtHashSlots];         // bin + 1; 0 = free
	__shared__ uint32_t hash_count[kPbwtHashSlots];
	__shared__ uint32_t bin_slot[kPbwtMaxBins];           // hash slots in use, in claiming order
	__shared__ uint32_t sorted_key[kPbwtMaxBins], sorted_count[kPbwtMaxBins], sorted_emit[kPbwtMaxBins];
	__shared__ uint32_t n_bins_s, reduce_max[kPbwtWaves], reduce_cnt[kPbwtWaves], failed_s;

	int const t = threadIdx.x, lane = t & 63, wave = t >> 6;
	uint32_t const chunk = blockIdx.x;
	uint64_t const cand_begin = chunk_first[chunk], cand_end = chunk_first[chunk + 1];
	uint32_t const per = (n_copies + kPbwtThreads - 1) / kPbwtThreads;          // <= kPbwtPerThread
	uint32_t const my_begin = (uint32_t) t * per < n_copies ? (uint32_t) t * per : n_copies;
	uint32_t const my_end = my_begin + per < n_copies ? my_begin + per : n_copies;

	for (uint32_t i = t; i < n_copies; i += kPbwtThreads) {
		order[0][i] = (unsigned short) start_order[(uint64_t) chunk * n_copies + i];
		divergence[0][i] = start_divergence[(uint64_t) chunk * n_copies + i];
	}
	for (uint32_t i = t; i < (uint32_t) kPbwtHashSlots; i += kPbwtThreads) { hash_key[i] = 0; hash_count[i] = 0; }
	if (t == 0) { n_bins_s = 0; failed_s = 0; }
	pbwt_block_sync();

	int cur = 0;
	uint64_t n_trials = 0;                                 // (kept by every thread: all of them see the same counts)
	uint32_t edge = cand_begin < cand_end ? cand_edge[cand_begin] : 0;
	uint64_t column_word = pbwt_fetch_column(paths_by_edge, words_per_edge, edge, n_edges, t);
	uint32_t *const my_pred = trial_pred + (uint64_t) chunk * trial_capacity;
	uint32_t *const my_class = trial_class_count + (uint64_t) chunk * trial_capacity;

	for (uint64_t cand = cand_begin; cand < cand_end; ++cand) {
		// ---- the edges before this candidate's node (find_cut_positions.cc:170-176 over pbwt.hh:77-134) -----------------
		uint32_t const upto = cand_edge[cand];
		for (; edge < upto; ++edge) {
			pbwt_step(paths_by_edge, words_per_edge, edge, n_edges, column_word, order, divergence, column, wave_items, cur, my_begin, my_end, t, lane, wave);
			cur ^= 1;
		}

		// ---- the candidate (find_cut_positions.cc:134-165) ---------------------------------------------------------------
		uint32_t const next = (uint32_t) cand;
		uint32_t const *const dv = divergence[cur];
		// the largest divergence value and how many copies hold it
		uint32_t my_max = 0;
		for (uint32_t i = my_begin; i < my_end; ++i) my_max = dv[i] > my_max ? dv[i] : my_max;
#pragma unroll
		for (int delta = 32; delta >= 1; delta >>= 1) { uint32_t const o = __shfl_xor(my_max, delta, 64); my_max = o > my_max ? o : my_max; }
		if (lane == 0) reduce_max[wave] = my_max;
		pbwt_block_sync();
		uint32_t d_max = 0;
		for (int w = 0; w < kPbwtWaves; ++w) d_max = reduce_max[w] > d_max ? reduce_max[w] : d_max;
		uint32_t my_cnt = 0;
		// every other copy goes into the bin of the candidate its value points to (clipped to next + 1: "no earlier candidate")
		for (uint32_t i = my_begin; i < my_end; ++i) {
			uint32_t const d = dv[i];
			if (d == d_max) { ++my_cnt; continue; }
			// biased 0 ("no match yet") is the SMALLEST value: the walk reaches it last, it points to no candidate, and the only
			// count that includes it is the final one (all copies): it takes no part in the bins
			if (0u == d) continue;
			uint32_t bin = next + 1u;
			if (d - 1u <= n_edges) { uint32_t const c = first_candidate_from_edge[d - 1u]; bin = c < bin ? c : bin; }
			uint32_t slot = (bin * 2654435761u) >> 20 & (uint32_t) (kPbwtHashSlots - 1);
			bool placed = false;
			for (int probe = 0; probe < kPbwtHashSlots && !placed; ++probe) {   // (bounded: a full table ends the chunk, below)
				uint32_t const seen = atomicCAS(&hash_key[slot], 0u, bin + 1u);
				if (0u == seen) {                                       // claimed a free slot
					uint32_t const k = atomicAdd(&n_bins_s, 1u);
					if (k < (uint32_t) kPbwtMaxBins) bin_slot[k] = slot;
					placed = true;
				} else if (seen == bin + 1u) {
					placed = true;
				} else {
					slot = (slot + 1u) & (uint32_t) (kPbwtHashSlots - 1);
				}
			}
			if (placed) atomicAdd(&hash_count[slot], 1u);
			else failed_s = 1u;
		}
#pragma unroll
		for (int delta = 32; delta >= 1; delta >>= 1) my_cnt += __shfl_xor(my_cnt, delta, 64);
		if (lane == 0) reduce_cnt[wave] = my_cnt;
		pbwt_block_sync();
		uint32_t count_max = 0;
		for (int w = 0; w < kPbwtWaves; ++w) count_max += reduce_cnt[w];
		uint32_t const n_bins = n_bins_s;
		if (n_bins > (uint32_t) kPbwtMaxBins || failed_s) {                        // (workgroup-uniform) more bins than this kernel sorts: the host takes the chunk
			if (t == 0) chunk_status[chunk] = 1u;
			return;
		}
		// sort the bins by key, descending: rank by counting (a few dozen bins)
		uint32_t key = 0, cnt = 0, rank = 0;
		if ((uint32_t) t < n_bins) {
			uint32_t const slot = bin_slot[t];
			key = hash_key[slot] - 1u;
			cnt = hash_count[slot];
			for (uint32_t j = 0; j < n_bins; ++j) rank += (hash_key[bin_slot[j]] - 1u) > key;
		}
		pbwt_block_sync();
		if ((uint32_t) t < n_bins) {
			sorted_key[rank] = key;
			sorted_count[rank] = cnt;
			uint32_t const slot = bin_slot[t];                       // leave the table empty for the next candidate
			hash_key[slot] = 0;
			hash_count[slot] = 0;
		}
		if (t == 0) n_bins_s = 0;
		pbwt_block_sync();
		// the trials: bin r is tried with class_count = copies in the bins before it (larger values) + the largest value's copies,
		// unless it is the clipped bin, the candidate itself, or too close (find_cut_positions.cc:139-160)
		uint32_t class_before = count_max, emit = 0;
		if ((uint32_t) t < n_bins) {
			for (uint32_t j = 0; j < (uint32_t) t; ++j) class_before += sorted_count[j];
			uint32_t const k = sorted_key[t];
			emit = (k != next + 1u && k != next && min_distance <= cand_aligned[next] - cand_aligned[k]) ? 1u : 0u;
			sorted_emit[t] = emit;
		}
		pbwt_block_sync();
		uint32_t emitted = 0;                                           // how many bins emit (every thread computes it: small)
		uint32_t my_slot = 0;
		for (uint32_t j = 0; j < n_bins; ++j) { if (j == (uint32_t) t) my_slot = emitted; emitted += sorted_emit[j]; }
		// after the loop: the segment may reach further left still (find_cut_positions.cc:162-165)
		uint32_t right_bound = next + 1u;
		if (n_bins) { uint32_t const smallest = sorted_key[n_bins - 1]; right_bound = smallest < right_bound ? smallest : right_bound; }
		bool const last_trial = 0u != right_bound && right_bound - 1u != next;
		if (n_trials + emitted + (last_trial ? 1u : 0u) > trial_capacity) {   // (workgroup-uniform)
			if (t == 0) chunk_status[chunk] = 1u;
			return;
		}
		if ((uint32_t) t < n_bins && emit) { my_pred[n_trials + my_slot] = sorted_key[t]; my_class[n_trials + my_slot] = class_before; }
		n_trials += emitted;
		if (last_trial) {
			if (t == 0) { my_pred[n_trials] = right_bound - 1u; my_class[n_trials] = n_copies; }
			++n_trials;
		}
		if (t == 0) trial_end[cand] = n_trials;
		pbwt_block_sync();                                                // sorted_* are rewritten by the next candidate
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
	int last_block_start;     // last index with a block-class boundary, -1 = none yet
	uint32_t block_starts;    // boundaries of the block (prev_cut_edge < d)
	uint32_t span_starts;     // boundaries of the two-block span (cut_pair_edge < d)
};

__device__ __forceinline__ class_scan_item class_combine(class_scan_item const &a, class_scan_item const &b)
{
	return class_scan_item{b.last_block_start > a.last_block_start ? b.last_block_start : a.last_block_start, a.block_starts + b.block_starts, a.span_starts + b.span_starts};
}

__device__ __forceinline__ class_scan_item class_shfl_up(class_scan_item const &x, int delta)
{
	return class_scan_item{__shfl_up(x.last_block_start, delta, 64), __shfl_up(x.block_starts, delta, 64), __shfl_up(x.span_starts, delta, 64)};
}

// "threshold < unbiased(d)" with d biased: 0 is the reference's DIVERGENCE_MAX ("no match yet": starts a class)
__device__ __forceinline__ bool past_edge(uint32_t d, uint32_t threshold) { return 0u == d || d - 1u > threshold; }

__global__ __launch_bounds__(kPbwtThreads) void pbwt_cut_records_kernel(
	uint64_t const *__restrict__ paths_by_edge, uint32_t words_per_edge, uint32_t n_copies, uint32_t n_edge_columns /* columns of paths_by_edge (>= every edge visited + 1 is not required: fetches are clamped) */,
	uint32_t const *__restrict__ cut_edge,                // [n_cuts]: edges before each cut node
	uint64_t const *__restrict__ chunk_first_cut,         // [n_chunks + 1]: chunk k handles the cuts [first[k], first[k + 1]); first[k] >= 1
	uint32_t const *__restrict__ start_edge,              // [n_chunks]: the given state is the one after this many edges (<= cut_edge[first[k] - 1])
	uint32_t const *__restrict__ start_order, uint32_t const *__restrict__ start_divergence,   // [n_chunks][n_copies]
	uint64_t pool_capacity, uint32_t *__restrict__ pool_lhs, uint32_t *__restrict__ pool_rhs, uint32_t *__restrict__ pool_size,   // [n_chunks][pool_capacity]
	uint64_t *__restrict__ rec_pool_end,                  // [n_cuts]: joined classes of the chunk up to and including this cut
	uint32_t *__restrict__ rec_distinct, uint32_t *__restrict__ rec_first_class, uint32_t *__restrict__ rec_first_is_ref,   // [n_cuts]
	uint32_t *__restrict__ chunk_status)
{
	__shared__ unsigned short order[2][kPbwtMaxCopies];
	__shared__ uint32_t divergence[2][kPbwtMaxCopies];
	__shared__ uint64_t column[kPbwtMaxCopies / 64];
	__shared__ pbwt_scan_item wave_items[kPbwtWaves];
	__shared__ class_scan_item class_items[kPbwtWaves];
	__shared__ unsigned short copy_class[2][kPbwtMaxCopies];      // per copy: the representative of its class at the last / the previous cut
	__shared__ unsigned short span_start_index[kPbwtMaxCopies];   // per joined class: where it starts in the order

	int const t = threadIdx.x, lane = t & 63, wave = t >> 6;
	uint32_t const chunk = blockIdx.x;
	uint64_t const cut_begin = chunk_first_cut[chunk], cut_end = chunk_first_cut[chunk + 1];
	if (cut_begin >= cut_end) { if (t == 0) chunk_status[chunk] = 0u; return; }
	uint32_t const per = (n_copies + kPbwtThreads - 1) / kPbwtThreads;
	uint32_t const my_begin = (uint32_t) t * per < n_copies ? (uint32_t) t * per : n_copies;
	uint32_t const my_end = my_begin + per < n_copies ? my_begin + per : n_copies;

	for (uint32_t i = t; i < n_copies; i += kPbwtThreads) {
		order[0][i] = (unsigned short) start_order[(uint64_t) chunk * n_copies + i];
		divergence[0][i] = start_divergence[(uint64_t) chunk * n_copies + i];
		copy_class[0][i] = (unsigned short) kPbwtNoClass;
		copy_class[1][i] = (unsigned short) kPbwtNoClass;
	}
	pbwt_block_sync();

	int cur = 0, rhs = 0;                                  // copy_class[rhs]: the classes the last cut left behind
	uint32_t edge = start_edge[chunk];
	uint64_t column_word = pbwt_fetch_column(paths_by_edge, words_per_edge, edge, n_edge_columns, t);
	// up to the cut before the chunk's first one; the classes it left behind (founder.cc:scan_cut_chunk)
	uint64_t const start_cut = cut_begin - 1;
	for (uint32_t const upto = cut_edge[start_cut]; edge < upto; ++edge) {
		pbwt_step(paths_by_edge, words_per_edge, edge, n_edge_columns, column_word, order, divergence, column, wave_items, cur, my_begin, my_end, t, lane, wave);
		cur ^= 1;
	}

	// The classes of a threshold: every copy's representative = the copy at the last boundary at or before it.
	// with_span: also the joined classes of the two-block span, written to the pool.  Returns false when the pool is full.
	uint64_t n_pool = 0;
	uint32_t *const my_lhs = pool_lhs + (uint64_t) chunk * pool_capacity, *const my_rhs = pool_rhs + (uint64_t) chunk * pool_capacity, *const my_size = pool_size + (uint64_t) chunk * pool_capacity;
	auto const classes_at_cut = [&](uint32_t block_threshold, bool with_span, uint32_t span_threshold, uint32_t &distinct_out) -> bool {
		unsigned short const *const ord = order[cur];
		uint32_t const *const dv = divergence[cur];
		class_scan_item mine{-1, 0u, 0u};
		for (uint32_t i = my_begin; i < my_end; ++i) {
			uint32_t const d = dv[i];
			if (past_edge(d, block_threshold)) { mine.last_block_start = (int) i; ++mine.block_starts; }
			if (with_span && past_edge(d, span_threshold)) ++mine.span_starts;
		}
		class_scan_item incl = mine;
#pragma unroll
		for (int delta = 1; delta < 64; delta <<= 1) {
			class_scan_item const up = class_shfl_up(incl, delta);
			if (lane >= delta) incl = class_combine(up, incl);
		}
		if (lane == 63) class_items[wave] = incl;
		pbwt_block_sync();
		class_scan_item before{-1, 0u, 0u}, total{-1, 0u, 0u};
		for (int w = 0; w < kPbwtWaves; ++w) {
			class_scan_item const wi = class_items[w];
			if (w < wave) before = class_combine(before, wi);
			total = class_combine(total, wi);
		}
		class_scan_item const lane_before = class_shfl_up(incl, 1);
		if (lane) before = class_combine(before, lane_before);
		distinct_out = total.block_starts;
		bool const fits = !with_span || n_pool + total.span_starts <= pool_capacity;     // (workgroup-uniform)
		// second pass: representatives, class arrays, joined-class heads
		int last = before.last_block_start;
		uint32_t span_at = before.span_starts;
		unsigned short const *const lhs_class = copy_class[rhs];       // (the previous cut's classes: read)
		unsigned short *const rhs_class = copy_class[rhs ^ 1];         // (this cut's: written; the arrays swap roles below)
		for (uint32_t i = my_begin; i < my_end; ++i) {
			uint32_t const d = dv[i], copy = ord[i];
			if (past_edge(d, block_threshold)) last = (int) i;
			uint32_t const rep = last >= 0 ? (uint32_t) ord[last] : kPbwtNoClass;
			if (with_span && fits && past_edge(d, span_threshold)) {
				uint32_t const l = lhs_class[copy];
				my_lhs[n_pool + span_at] = kPbwtNoClass == l ? 0xFFFFFFFFu : l;
				my_rhs[n_pool + span_at] = kPbwtNoClass == rep ? 0xFFFFFFFFu : rep;
				span_start_index[span_at] = (unsigned short) i;
				++span_at;
			}
			rhs_class[copy] = (unsigned short) rep;
		}
		pbwt_block_sync();
		if (with_span && fits) {
			for (uint32_t j = t; j < total.span_starts; j += kPbwtThreads)
				my_size[n_pool + j] = (j + 1 < total.span_starts ? (uint32_t) span_start_index[j + 1] : n_copies) - (uint32_t) span_start_index[j];
			n_pool += total.span_starts;
		}
		rhs ^= 1;
		pbwt_block_sync();                                                // class_items / span_start_index are rewritten by the next call
		return fits;
	};

	if (start_cut >= 1) {
		uint32_t ignored;
		classes_at_cut(cut_edge[start_cut - 1], false, 0u, ignored);
	}

	bool first_is_ref = true;
	for (uint64_t cut = cut_begin; cut < cut_end; ++cut) {
		for (uint32_t const upto = cut_edge[cut]; edge < upto; ++edge) {
			uint32_t const zeros = pbwt_step(paths_by_edge, words_per_edge, edge, n_edge_columns, column_word, order, divergence, column, wave_items, cur, my_begin, my_end, t, lane, wave);
			cur ^= 1;
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
			rec_first_class[cut] = order[cur][0];
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
