// output.hh -- the output classes of the reference (include/vcf2multialign/output.hh:26-130) over the GPU path.
//
// Same class names, constructor arguments and row/identifier conventions; output_a2m() batches all rows of a
// file into one v2m_splice_rows() call instead of calling output_sequence() per row.  With pipe_cmd the sequences go
// to the standard input of `pipe_cmd <name>` instead of a file (output.cc:26-38,49-68).
// founder_sequence_greedy_output takes its cut positions and matchings from founder.hh (find_cut_positions /
// find_matchings, sequential host algorithms) or from a cut-position file.
#pragma once

#include <cstdint>
#include <ostream>
#include <string>
#include <vector>

#include "gpu_path.hh"

namespace v2m::host {

struct output_delegate {
	virtual ~output_delegate() {}
	virtual void will_handle_sample(std::string const &sample, u32 sample_idx, u32 chr_copy_idx) = 0;
	virtual void will_handle_founder_sequence(u32 idx) = 0;
	virtual void handled_sequences(u32 sequence_count) = 0;
};

struct null_output_delegate final : output_delegate {
	void will_handle_sample(std::string const &, u32, u32) override {}
	void will_handle_founder_sequence(u32) override {}
	void handled_sequences(u32) override {}
};

class output {
public:
	output(gpu_context &gpu, char const *pipe_cmd, char const *chromosome_id, bool should_output_reference, bool should_output_unaligned, output_delegate &delegate);
	virtual ~output() {}

	// The graph must be the one uploaded to the context (upload_graph()).
	virtual void output_separate(variant_graph const &graph, bool should_include_fasta_header) = 0;
	void output_a2m(variant_graph const &graph, char const *dst_name);

	// More GPUs for the same output (each must hold the same uploaded graph): the rows of an aligned A2M file are
	// then sharded in contiguous blocks over all contexts, one host thread per context, and every thread writes its
	// rows straight to their final place in the file (every aligned row has the same length, so offsets are known
	// up front; there is no collective and no reordering buffer -- SURVEY.md section 8e).  Unaligned output and
	// std::ostream targets use the first context only.
	void add_gpu(gpu_context &gpu) { m_more_gpus.push_back(&gpu); }

	// The chromosome copies each context's path matrix holds (one entry per context, the first context first), when the
	// matrix was sharded with upload_path_slice(): rows of a plain haplotype batch then go to the context that owns their
	// copy, with the copy index re-based to the shard (the REF row to the first context).  Without shards every context is
	// expected to hold the whole matrix and rows are split evenly.
	void set_copy_shards(std::vector<copy_shard> shards) { m_copy_shards = std::move(shards); }

	// The other way to share the path matrix, for every output that has to leave in row order (std::ostream targets: pipes,
	// unaligned A2M; one file per sequence): the contexts hold the chromosome copies dealt round-robin in blocks
	// (upload_path_blocks()), every context splices ITS rows on its own thread, and the rows are handed to the one writer in
	// row order straight from the contexts' pinned buffers (no copy, no queue: a turnstile on the row index).  While one
	// context's rows are being written, the others' next blocks are crossing PCIe.
	void set_copy_interleave(copy_interleave deal) { m_copy_interleave = deal; m_interleaved = true; }
	virtual void output_a2m(variant_graph const &graph, std::ostream &stream) = 0;

protected:
	struct row_set {
		std::vector<std::string> ids;           // FASTA identifiers (a2m) or file names (separate)
		std::vector<std::uint32_t> copy_index;
		std::vector<std::uint64_t> cut_offsets{0}, cut_nodes;
		std::vector<std::uint32_t> cut_copies;
		bool any_cuts{};
	};
	void splice(row_set const &rows, v2m_sink_fn sink, void *user);             // the sink sees the rows in batch order, one at a time
	void splice_in_turns(row_set const &rows, v2m_sink_fn sink, void *user);
	// rows that need no order and a sink that keeps them until it calls v2m_row_release (a pool of writers): v2m_splice_rows_held per context
	void splice_held(row_set const &rows, v2m_hold_sink_fn sink, void *user);
	static std::vector<std::uint32_t> rebased_copies(row_set const &rows, std::uint64_t first, std::uint64_t last, copy_shard shard);
	void write_a2m(row_set const &rows, std::ostream &stream);
	void write_a2m_sharded(row_set const &rows, char const *dst_name);
	virtual row_set a2m_rows(variant_graph const &graph) = 0;
	void write_separate(row_set const &rows);
	std::string prefixed(std::string const &name, char sep) const;

	gpu_context &m_gpu;
	std::vector<gpu_context *> m_more_gpus;
	std::vector<copy_shard> m_copy_shards;
	copy_interleave m_copy_interleave;
	bool m_interleaved{};
	char const *m_pipe_cmd{};
	char const *m_chromosome_id{};
	output_delegate *m_delegate{};
	bool m_should_output_reference{};
	bool m_should_output_unaligned{};
};

class haplotype_output final : public output {
public:
	using output::output;
	using output::output_a2m;
	void output_separate(variant_graph const &graph, bool should_include_fasta_header) override;
	void output_a2m(variant_graph const &graph, std::ostream &stream) override;
private:
	row_set rows_for(variant_graph const &graph, char sep, char const *suffix);
	row_set a2m_rows(variant_graph const &graph) override { return rows_for(graph, '\t', ""); }
};

class founder_sequence_greedy_output final : public output {
public:
	using output::output;
	using output::output_a2m;
	// cut_positions: node indices, first 0, last the sink; assigned_samples: (cuts - 1) rows x founders columns,
	// column-major, one column per founder (founder_sequence_greedy_output.cc:171,544).
	void set_cut_positions(std::vector<u64> cut_positions) { m_cut_positions = std::move(cut_positions); }
	void set_assigned_samples(std::vector<u32> column_major, u32 founder_count) { m_assigned_samples = std::move(column_major); m_founder_count = founder_count; }
	void output_separate(variant_graph const &graph, bool should_include_fasta_header) override;
	void output_a2m(variant_graph const &graph, std::ostream &stream) override;
private:
	row_set rows_for(char sep, char const *suffix);
	row_set a2m_rows(variant_graph const &) override { return rows_for('\t', ""); }
	std::vector<u64> m_cut_positions;
	std::vector<u32> m_assigned_samples;
	u32 m_founder_count{};
};

} // namespace v2m::host
