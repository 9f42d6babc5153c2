// founder.hh -- host side of --founder-sequences: where to cut the graph and which chromosome copy each
// founder follows between consecutive cuts.
//
// Both are strictly sequential over the ALT edges (a positional BWT with divergence counts; every step
// depends on the previous one), which is why they stay on the host.  They
// restate the reference's find_initial_cut_positions_lambda_min (libvcf2multialign/find_cut_positions.cc:93-211),
// pbwt_context (include/vcf2multialign/pbwt.hh:21-145) and founder_sequence_greedy_output::find_matchings
// (libvcf2multialign/founder_sequence_greedy_output.cc:154-512); the rows they select are then spliced on the
// GPU (output.hh: founder_sequence_greedy_output).  They read paths_by_edge_and_chrom_copy, the builder's own
// (un-transposed) matrix.
#pragma once

#include <vector>

#include "variant_graph.hh"

namespace v2m::host {

constexpr u32 kCutPositionScoreMax = UINT32_MAX;   // find_cut_positions.hh:17

// Cut positions (node indices, first 0, last the sink) minimising the largest number of distinct path
// segments between two consecutive cuts, subject to a minimum aligned distance.  Returns the score
// (kCutPositionScoreMax if there is no solution).
u32 find_cut_positions(variant_graph const &graph, u64 min_distance, std::vector<u64> &cut_positions);

// Greedy assignment of path equivalence classes to founders.  assigned_samples receives the
// (cut_positions.size() - 1) x founder_count matrix, column-major, one column per founder; slots that stay
// unassigned hold kPloidyMax.  Returns false when there is nothing to match.
bool find_matchings(
	variant_graph const &graph, std::vector<u64> const &cut_positions, u32 founder_count, bool keep_ref_edges,
	std::vector<u32> &assigned_samples);

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
