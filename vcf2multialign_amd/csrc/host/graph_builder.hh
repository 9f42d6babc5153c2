// graph_builder.hh -- streaming construction of the variant graph from variant records.
//
// Behaviour follows the reference's build_variant_graph() (libvcf2multialign/variant_graph.cc:108-454,
// SURVEY.md Appendix B) record for record; the structure does not: the reference is one function
// around a VCF-reader callback, this is a push-style builder that any record source can drive --
// the VCF text reader (vcf_reader.cc) and the synthetic generator (synth.cc) both do -- with the
// pending ALT targets in a binary heap instead of a multimap.
#pragma once

#include <functional>
#include <queue>
#include <string_view>
#include <vector>

#include <algorithm>
#include <new>
#include <stdexcept>

#include "variant_graph.hh"

namespace v2m::host {

enum class alt_kind { sequence, deletion, unhandled };   // vcf::sv_type NONE / DEL / everything else (variant_graph.cc:328-364)

alt_kind classify_alt(std::string_view alt);

struct alt_allele {
	alt_kind kind{alt_kind::unhandled};
	std::string_view sequence;   // for alt_kind::sequence
};

struct overlap_info {
	u64 ref_pos{};
	u32 copy_row{};      // chromosome copy (row of paths_by_edge_and_chrom_copy)
	u32 alt_number{};    // 1-based GT value
};

class graph_builder {
public:
	typedef std::function<void(overlap_info const &)> overlap_callback;

	// track_paths == false builds nodes/edges only (the path matrix is then produced elsewhere,
	// e.g. directly in HBM by the synthetic generator).
	// path_alignment: both dimensions of paths_by_edge_and_chrom_copy are padded to a multiple of it.  64 is what the
	// reference does (variant_graph.cc:277,449); 1024 makes every matrix column start on a 128-B line, which the GPU
	// transpose rewards.  Padding rows and columns are zero either way.
	graph_builder(variant_graph &graph, bool track_paths = true, u64 path_alignment = 64);

	// Must be called once before the first record, with the ploidy of every (included) sample
	// (variant_graph.cc:215-288: taken from the first matching record).
	void begin(std::vector<std::string> sample_names, std::vector<u32> const &ploidies);

	// One record.  Positions must be non-decreasing (variant_graph.cc:292-297; returns false otherwise).
	// After the call edge_for_alt(i) tells which edge ALT i became (kEdgeMax if it was skipped).
	bool add_record(u64 ref_pos, u64 ref_allele_length, alt_allele const *alts, std::size_t n_alts);
	u64 edge_for_alt(std::size_t alt_idx) const { return m_edges_by_alt[alt_idx]; }

	// What the reference has done with a record when its delegate's ref_column_mismatch() returns false and parsing stops
	// (variant_graph.cc:300-313): the pending targets up to ref_pos and the record's own node exist, none of its edges, and
	// the previous record's position is still the "previous position" that finish() measures the sink's distance from.
	void add_record_node_only(u64 ref_pos);
	// Whether a record at ref_pos would lie before the previous one (variant_graph.cc:293-297: an error in the reference).
	bool would_go_back(u64 ref_pos) const { return ref_pos < m_prev_ref_pos; }

	// Genotype of one chromosome copy for the record just added: alt_number is the 1-based GT value
	// (0 and missing are not passed).  Sets the path bit, reporting an overlap first when the copy is
	// still inside an earlier ALT (variant_graph.cc:399-424 -- the bit is set even then).
	void set_genotype(u32 copy_row, u32 alt_number);

	// Sink node and final column count (variant_graph.cc:437-451).  The transpose that follows in the
	// reference (:453) is NOT done here: it is the GPU's job (gpu_path.hh: transpose_paths()).
	void finish(u64 ref_length);

	void on_overlap(overlap_callback cb) { m_overlap_cb = std::move(cb); }

	// A hint, after begin(): about this many ALT edges are coming.  The path matrix grows by doubling (:368-376 grows it by 2048
	// columns at a time); with its final size reserved the doublings no longer copy it (config 3: 632 MB, 0.3 s of the merge stage).
	// The hint may be far off (a multi-chromosome VCF filtered to one chromosome: the caller's estimate counts every record), so it
	// is capped -- 8 GiB, more than the config-5 matrix needs -- and a reserve that fails is simply not made: growth by doubling
	// builds the same matrix.
	void expect_edges(u64 n_edges)
	{
		auto &m(m_graph.paths_by_edge_and_chrom_copy);
		if (!m_track_paths || !m.rows) return;
		u64 const words(m.rows / 64 * (m_path_alignment * ((n_edges + m_path_alignment - 1) / m_path_alignment)));
		u64 const cap_words((u64(8) << 30) / sizeof(u64));
		try { m.words.reserve(std::min(words, cap_words)); }
		catch (std::bad_alloc const &) {}
		catch (std::length_error const &) {}
	}

private:
	struct pending_target {
		u64 ref_pos;
		u64 seq;          // insertion order, to break ties like std::multimap does
		u64 edge;
		u64 aln_pos;
		bool operator>(pending_target const &o) const { return ref_pos != o.ref_pos ? ref_pos > o.ref_pos : seq > o.seq; }
	};

	u64 add_node(u64 ref_pos, u64 aln_pos);
	u64 add_or_update_node(u64 ref_pos, u64 aln_pos);
	u64 add_edge(std::string_view label);
	void flush_targets(u64 ref_pos);

	variant_graph &m_graph;
	bool m_track_paths;
	u64 m_path_alignment;
	std::priority_queue<pending_target, std::vector<pending_target>, std::greater<pending_target>> m_pending;
	u64 m_seq{};
	u64 m_aln_pos{};
	u64 m_prev_ref_pos{};
	u64 m_cur_ref_pos{};
	u64 m_min_edge{};
	std::vector<u64> m_edges_by_alt;
	std::vector<u64> m_current_edge_targets;
	std::vector<u64> m_target_ref_pos_by_copy;
	overlap_callback m_overlap_cb;
};

} // namespace v2m::host
