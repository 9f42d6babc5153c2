// readers.hh -- plain-text input for the host driver: single-sequence FASTA and VCF -> variant graph.
//
// These stand in for the libbio readers the reference uses (lb::read_single_fasta_sequence at
// vcf2multialign/main.cc:381, vcf::reader at libvcf2multialign/variant_graph.cc:133-146,181); libbio is an
// absent submodule, so only the behaviour visible at those call sites is reproduced.
#pragma once

#include <cstdint>
#include <string>
#include <string_view>
#include <vector>

#include "variant_graph.hh"

namespace v2m::host {

typedef std::vector<char> sequence_type;   // variant_graph.hh:33

// First sequence of the file, or the one whose identifier (text after '>' up to the first blank) equals seq_id.
bool read_single_fasta_sequence(char const *path, sequence_type &seq, char const *seq_id = nullptr);

// variant_graph.hh:138-158
struct build_graph_delegate {
	virtual ~build_graph_delegate() {}
	virtual bool should_include(std::string_view sample_name, u32 chrom_copy_idx) const = 0;
	virtual void report_overlapping_alternative(
		u64 lineno, u64 ref_pos, std::string_view var_id, std::string_view sample_name, u32 chrom_copy_idx, u32 gt) = 0;
	// return false to stop building
	virtual bool ref_column_mismatch(u64 var_idx, u64 ref_pos, std::string_view ref_in_vcf, std::string_view expected) = 0;
};

// variant_graph.hh:167-171
struct build_graph_statistics {
	u64 handled_variants{};
	u64 chr_id_mismatches{};
};

// build_variant_graph (variant_graph.cc:108-454) up to, but not including, the final transpose at :453:
// paths_by_edge_and_chrom_copy is filled, paths_by_chrom_copy_and_edge is left for the GPU
// (gpu_path.hh: transpose_paths).  Throws std::runtime_error on malformed input.
//
// The reference parses and builds on one thread.  Here the genotype text -- 5 * 10^9 fields at config 3 -- is
// parsed by `threads` workers over 8-MB chunks of whole lines into sparse (copy, allele) lists, and one thread
// merges the chunks in file order through graph_builder, so the graph and the delegate calls are exactly those
// of a sequential pass.  threads == 0: one per hardware thread, at most 16.  path_alignment: see graph_builder.
void build_variant_graph(
	sequence_type const &ref_seq, char const *variants_path, char const *chr_id,
	variant_graph &graph, build_graph_statistics &stats, build_graph_delegate &delegate, unsigned threads = 0, u64 path_alignment = 64);

} // namespace v2m::host
