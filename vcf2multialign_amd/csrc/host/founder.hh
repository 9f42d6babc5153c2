// founder.hh -- host side of --founder-sequences: where to cut the graph and which chromosome copy each
// founder follows between consecutive cuts.
//
// Both walk a positional BWT with divergence counts over the ALT edges, every step depending on the previous
// one -- but the state after any number of edges can also be built from scratch from the transposed path matrix, so the walk is
// cut into chunks (founder.cc: pbwt_state_at) that run on several host threads or, with a founder_walker (round 3), as one GPU
// workgroup each (csrc/founder_kernels.hpp behind v2m_pbwt_cut_trials / v2m_pbwt_cut_records); the score updates and the
// greedy assignment, which are strictly sequential, stay here.  They restate the reference's
// find_initial_cut_positions_lambda_min (libvcf2multialign/find_cut_positions.cc:93-211),
// pbwt_context (include/vcf2multialign/pbwt.hh:21-145) and founder_sequence_greedy_output::find_matchings
// (libvcf2multialign/founder_sequence_greedy_output.cc:154-512); the rows they select are then spliced on the
// GPU (output.hh: founder_sequence_greedy_output).  They read paths_by_edge_and_chrom_copy, the builder's own
// (un-transposed) matrix.
#pragma once

#include <functional>
#include <memory>
#include <vector>

#include "variant_graph.hh"

namespace v2m::host {

constexpr u32 kCutPositionScoreMax = UINT32_MAX;   // find_cut_positions.hh:17

// Cut positions (node indices, first 0, last the sink) minimising the largest number of distinct path
// segments between two consecutive cuts, subject to a minimum aligned distance.  Returns the score
// (kCutPositionScoreMax if there is no solution).
// threads: as for find_matchings below.
u32 find_cut_positions(variant_graph const &graph, u64 min_distance, std::vector<u64> &cut_positions, unsigned threads = 1);

// Something that walks chunks of the two searches elsewhere (the GPU: gpu_path.cc:gpu_founder_walker over v2m_pbwt_cut_trials and
// v2m_pbwt_cut_records): given every chunk's start state it produces, per chunk, what the reference's loops collect at each
// candidate (find_cut_positions.cc:134-165) / at each cut (founder_sequence_greedy_output.cc:215-251), or marks the chunk as
// left undone.  The start states the cut search built are kept for the matching that usually follows it.
struct founder_walker {
	virtual ~founder_walker() {}
	virtual u64 max_copies() const = 0;
	// ... and in records() (the matching's walks keep two class arrays beside the state: fewer copies fit)
	virtual u64 max_copies_records() const { return max_copies(); }
	// how many chunks it likes to walk at once (one workgroup each on the GPU: a couple per compute unit); 0 = no preference
	virtual std::size_t preferred_chunks() const { return 0; }
	// cand_edge / cand_aligned: all candidates; chunk_first: n_chunks + 1 candidate indices; start_*: [n_chunks][n_copies];
	// trial_pred / trial_class: n_chunks x capacity; trial_end: per candidate, within its chunk; status: per chunk, 0 = done
	virtual void walk(u64 n_copies, u64 min_distance, std::vector<u32> const &cand_edge, std::vector<u64> const &cand_aligned,
		std::vector<u64> const &chunk_first, u32 const *start_order, u32 const *start_divergence,
		u64 capacity, u32 *trial_pred, u32 *trial_class, u64 *trial_end, u32 *status) = 0;
	// The same walks with every chunk's pairs handed to `take` (chunk, status, preds, class counts, number of pairs) in chunk order
	// instead of landing in arrays of the caller's; trial_end and status are complete before the first call.  Here: walk() into
	// arrays of its own, then the calls.  The GPU walker streams the pairs back while `take` works (v2m_pbwt_cut_trials_streamed).
	typedef std::function<void(std::size_t, u32, u32 const *, u32 const *, u64)> chunk_taker;
	virtual void walk_streamed(u64 n_copies, u64 min_distance, std::vector<u32> const &cand_edge, std::vector<u64> const &cand_aligned,
		std::vector<u64> const &chunk_first, u32 const *start_order, u32 const *start_divergence,
		u64 capacity, u64 *trial_end, u32 *status, chunk_taker const &take);
	// cut_edge: edges before every cut node; chunk_first_cut: n_chunks + 1 cut indices; start_edge: per chunk, the edges the given
	// state has seen; pool_*: n_chunks x pool_capacity joined classes; rec_*: per cut
	virtual void records(u64 n_copies, std::vector<u32> const &cut_edge, std::vector<u64> const &chunk_first_cut, std::vector<u32> const &start_edge,
		u32 const *start_order, u32 const *start_divergence, u64 pool_capacity, u32 *pool_lhs, u32 *pool_rhs, u32 *pool_size,
		u64 *rec_pool_end, u32 *rec_distinct, u32 *rec_first_class, u32 *rec_first_is_ref, u32 *status) = 0;
	// what the last search did: chunks walked by the walker / walked here after all
	u64 chunks_walked{}, chunks_left{};
	// called once the matching has made its last call to the walker (the rest -- sorting, the greedy assignment -- is host work):
	// the caller may use the device for something else from then on
	std::function<void()> on_last_walk;
	// pBWT states (order, biased divergence; [state][copy]) after state_edge[k] edges, left behind by the cut search
	std::vector<u32> state_edge;
	std::unique_ptr<u32[]> state_order, state_divergence;
	u64 state_copies{};
	// which graph the states were built for: its edge count and the address of its transposed matrix's words (a walker reused on
	// another graph with as many copies must not start from the first one's states)
	u64 state_edges{};
	void const *state_matrix{};
};
typedef founder_walker cut_trial_walker;

// A founder_walker that walks its chunks right here, edge by edge from the states it is given, on one thread: the walked
// searches' plumbing (chunk bounds, state reuse, capacities, chunks handed back) without a GPU -- for the CPU test suite and
// the sanitizer harness (tools/sanitize_host.sh).  It is also the plainest statement of what the GPU kernels compute.
// max_copies: pretend to hold no more copies than this.
std::unique_ptr<founder_walker> make_host_founder_walker(variant_graph const &graph, u64 max_copies = UINT64_MAX);
// The same search with the chunk walks handed to `walker` (the start states are still built here, on `threads` threads, and
// the score updates stay sequential); chunks the walker leaves undone, and graphs it cannot take, are walked here.
u32 find_cut_positions(variant_graph const &graph, u64 min_distance, std::vector<u64> &cut_positions, unsigned threads, founder_walker *walker);

// Greedy assignment of path equivalence classes to founders.  assigned_samples receives the
// (cut_positions.size() - 1) x founder_count matrix, column-major, one column per founder; slots that stay
// unassigned hold kPloidyMax.  Returns false when there is nothing to match.
// threads: 1 = the reference's sequential loop; more (0 = up to 16 hardware threads) splits the cuts into chunks whose
// pBWT state is built from scratch from paths_by_chrom_copy_and_edge (the transpose's result), with the same outcome;
// without that matrix the search is sequential whatever `threads` says.
bool find_matchings(
	variant_graph const &graph, std::vector<u64> const &cut_positions, u32 founder_count, bool keep_ref_edges,
	std::vector<u32> &assigned_samples, unsigned threads = 1);
// ... with the chunk walks handed to `walker` (the greedy assignment stays sequential, here).
bool find_matchings(
	variant_graph const &graph, std::vector<u64> const &cut_positions, u32 founder_count, bool keep_ref_edges,
	std::vector<u32> &assigned_samples, unsigned threads, founder_walker *walker);

// --output-cut-positions / --input-cut-positions (founder_sequence_greedy_output.cc:118-136).  The reference writes the
// struct {min_distance, cut_positions, score} (output.hh:89-97,133-139) through cereal's PortableBinaryOutputArchive;
// cereal is not part of the reference tree, so the layout below restates cereal's published portable-binary encoding
// and is not pinned by any file of the reference: one byte 1 (little-endian marker), u32 class version 0,
// u64 min_distance, u64 count, count x u64 node indices, u32 score; all little-endian.
struct cut_position_file {
	std::vector<u64> cut_positions;
	u64 min_distance{};
	u32 score{};
};
void write_cut_positions(cut_position_file const &cuts, char const *path);
cut_position_file read_cut_positions(char const *path);

} // namespace v2m::host
