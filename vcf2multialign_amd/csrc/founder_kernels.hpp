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
	__syncthreads();

	int cur = 0;
	uint64_t n_trials = 0;                                 // (kept by every thread: all of them see the same counts)
	uint32_t edge = cand_begin < cand_end ? cand_edge[cand_begin] : 0;
	uint32_t *const my_pred = trial_pred + (uint64_t) chunk * trial_capacity;
	uint32_t *const my_class = trial_class_count + (uint64_t) chunk * trial_capacity;

	for (uint64_t cand = cand_begin; cand < cand_end; ++cand) {
		// ---- the edges before this candidate's node (find_cut_positions.cc:170-176 over pbwt.hh:77-134) -----------------
		uint32_t const upto = cand_edge[cand];
		for (; edge < upto; ++edge) {
			for (uint32_t w = t; w < words_per_edge; w += kPbwtThreads) column[w] = paths_by_edge[(uint64_t) edge * words_per_edge + w];
			__syncthreads();
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
			__syncthreads();
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
			__syncthreads();
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
		__syncthreads();
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
		__syncthreads();
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
		__syncthreads();
		if ((uint32_t) t < n_bins) {
			sorted_key[rank] = key;
			sorted_count[rank] = cnt;
			uint32_t const slot = bin_slot[t];                       // leave the table empty for the next candidate
			hash_key[slot] = 0;
			hash_count[slot] = 0;
		}
		if (t == 0) n_bins_s = 0;
		__syncthreads();
		// the trials: bin r is tried with class_count = copies in the bins before it (larger values) + the largest value's copies,
		// unless it is the clipped bin, the candidate itself, or too close (find_cut_positions.cc:139-160)
		uint32_t class_before = count_max, emit = 0;
		if ((uint32_t) t < n_bins) {
			for (uint32_t j = 0; j < (uint32_t) t; ++j) class_before += sorted_count[j];
			uint32_t const k = sorted_key[t];
			emit = (k != next + 1u && k != next && min_distance <= cand_aligned[next] - cand_aligned[k]) ? 1u : 0u;
			sorted_emit[t] = emit;
		}
		__syncthreads();
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
		__syncthreads();                                                // sorted_* are rewritten by the next candidate
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
