// v2m_oracle.cc -- CPU ORACLE. TEST INFRASTRUCTURE ONLY.
//
// A single-threaded CPU restatement of the vcf2multialign hot path (and of the
// graph builder needed to reach the reference's golden vectors from its fixture
// files).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
// may load this library; the product (vcf2multialign_amd/) never links, imports
// or executes anything from oracle/.
//
// Parity status: PINNED.  The restatement reproduces every golden value the
// reference's own tests hold for this path (tests/test_oracle_goldens.py):
//   * 5 complete A2M texts            /root/reference/tests/founder_sequences.cc:122-186
//   * 5 node/edge tables              /root/reference/tests/variant_graph.cc:251-337
//   * the expected overlap report     /root/reference/tests/variant_graph.cc:267,288
//   * 3 fixed + property transposes   /root/reference/tests/transpose_matrix.cc:188-279
// The reference itself cannot be compiled here (libbio, cereal, boost, ragel,
// range-v3 absent), so there is no oracle/_ref build.
//
// Each function cites the reference file:line whose behaviour it restates
// (paths relative to /root/reference).  Nothing here is copied from there: the
// data model is flat arrays + std::string labels, the VCF/FASTA readers are
// minimal restatements of what the libbio call sites need.

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <map>
#include <ostream>
#include <sstream>
#include <string>
#include <string_view>
#include <vector>

namespace {

constexpr std::uint32_t kPloidyMax = UINT32_MAX;   // variant_graph.hh:55 (PLOIDY_MAX)
constexpr std::uint64_t kEdgeMax = UINT64_MAX;     // variant_graph.hh:53 (EDGE_MAX)

// ---------------------------------------------------------------------------
// Bit matrix: column-major, 64 rows per word, LSB-first inside a word.
// Restates the libbio::bit_matrix call sites listed in SURVEY.md 8c
// (operator()(row, col), number_of_rows/columns, word_at).  Bit order is our
// choice (LSB-first, matching the shift arithmetic at transpose_matrix.cc:81-84);
// the reference's tests cannot distinguish it.
// ---------------------------------------------------------------------------
struct bit_matrix {
	std::uint64_t rows{};   // multiple of 64 whenever the matrix is transposed
	std::uint64_t cols{};
	std::vector<std::uint64_t> words;

	bit_matrix() = default;
	bit_matrix(std::uint64_t r, std::uint64_t c) : rows(r), cols(c), words((r * c + 63) / 64, 0) {}

	bool get(std::uint64_t r, std::uint64_t c) const
	{
		std::uint64_t const idx(c * rows + r);
		return (words[idx >> 6] >> (idx & 63)) & 1;
	}

	void set(std::uint64_t r, std::uint64_t c)
	{
		std::uint64_t const idx(c * rows + r);
		words[idx >> 6] |= std::uint64_t(1) << (idx & 63);
	}

	// bit_matrix::resize(total_bits, 0) as used at variant_graph.cc:374,450:
	// the row count stays, the column count follows from the new total.
	void resize_total_bits(std::uint64_t total_bits)
	{
		words.resize((total_bits + 63) / 64, 0);
		cols = rows ? total_bits / rows : 0;
	}
};


// ---------------------------------------------------------------------------
// variant_graph.hh:57-80
// ---------------------------------------------------------------------------
struct overlap_report {
	std::uint64_t lineno{};
	std::uint64_t ref_pos{};
	std::string var_id;
	std::string sample_name;
	std::uint32_t chrom_copy_idx{};
	std::uint32_t gt{};
};

struct graph {
	std::vector<std::uint64_t> reference_positions;
	std::vector<std::uint64_t> aligned_positions;
	std::vector<std::uint64_t> alt_edge_targets;
	std::vector<std::uint64_t> alt_edge_count_csum;
	std::vector<std::string> alt_edge_labels;
	bit_matrix paths_by_chrom_copy_and_edge;   // rows = edges, cols = chromosome copies
	bit_matrix paths_by_edge_and_chrom_copy;   // rows = chromosome copies, cols = edges
	std::vector<std::string> sample_names;
	std::vector<std::uint32_t> ploidy_csum;

	// Not part of the reference's struct: what its delegates would have seen.
	std::vector<overlap_report> overlaps;
	std::uint64_t handled_variants{};
	std::uint64_t chr_id_mismatches{};

	// flattened views handed out through the C API
	std::vector<std::uint64_t> label_offsets;
	std::string label_bytes;
	std::string sample_name_blob;

	std::uint64_t node_count() const { return reference_positions.size(); }
	std::uint64_t edge_count() const { return alt_edge_targets.size(); }

	// variant_graph.cc:77-83
	std::uint64_t add_node(std::uint64_t ref_pos, std::uint64_t aln_pos)
	{
		reference_positions.push_back(ref_pos);
		aligned_positions.push_back(aln_pos);
		alt_edge_count_csum.push_back(alt_edge_count_csum.back());
		return reference_positions.size() - 1;
	}

	// variant_graph.cc:86-96
	std::uint64_t add_or_update_node(std::uint64_t ref_pos, std::uint64_t aln_pos)
	{
		if (reference_positions.back() < ref_pos)
			return add_node(ref_pos, aln_pos);
		aligned_positions.back() = std::max(aligned_positions.back(), aln_pos);
		return reference_positions.size() - 1;
	}

	// variant_graph.cc:99-105
	std::uint64_t add_edge(std::string_view label)
	{
		++alt_edge_count_csum.back();
		alt_edge_targets.push_back(0);
		alt_edge_labels.emplace_back(label);
		return alt_edge_targets.size() - 1;
	}

	void finish_views()
	{
		label_offsets.assign(1, 0);
		label_bytes.clear();
		for (auto const &l : alt_edge_labels) {
			label_bytes += l;
			label_offsets.push_back(label_bytes.size());
		}
		sample_name_blob.clear();
		for (auto const &s : sample_names) {
			sample_name_blob += s;
			sample_name_blob.push_back('\0');
		}
	}
};


// ---------------------------------------------------------------------------
// transpose_matrix.cc:18-38 -- 8x8 bit-block transpose.  Restated with the
// three-round masked swap (Hacker's Delight 7-3) rather than the reference's
// 15 diagonal masks; same function on a 64-bit word holding 8 rows of 8 bits.
// ---------------------------------------------------------------------------
inline std::uint64_t transpose8x8(std::uint64_t x)
{
	std::uint64_t t;
	t = (x ^ (x >> 7)) & 0x00AA00AA00AA00AAULL;  x ^= t ^ (t << 7);
	t = (x ^ (x >> 14)) & 0x0000CCCC0000CCCCULL; x ^= t ^ (t << 14);
	t = (x ^ (x >> 28)) & 0x00000000F0F0F0F0ULL; x ^= t ^ (t << 28);
	return x;
}


// transpose_matrix.cc:41-109.  Same traversal: for each 64-row word index, each
// of its 8 bytes, each group of 64 source columns, gather one byte per column
// into eight 8x8 blocks, transpose each, scatter bytes into 8 destination words.
int transpose_blocks(std::uint64_t const *src, std::uint64_t nrows, std::uint64_t ncols, std::uint64_t *dst)
{
	if (0 == ncols) return 0;                       // transpose_matrix.cc:48-49
	if (nrows % 64 || ncols % 64) return -1;        // transpose_matrix.cc:53-54 (asserts there)
	std::uint64_t const src_col_words(nrows / 64);
	std::uint64_t const dst_col_words(ncols / 64);
	std::fill(dst, dst + nrows * ncols / 64, 0);

	for (std::uint64_t rw(0); rw < src_col_words; ++rw) {
		for (unsigned byte_idx(0); byte_idx < 8; ++byte_idx) {
			for (std::uint64_t cg(0); cg < dst_col_words; ++cg) {
				std::uint64_t blocks[8];
				for (unsigned b(0); b < 8; ++b) {
					std::uint64_t acc(0);
					for (unsigned k(0); k < 8; ++k) {
						std::uint64_t const col(64 * cg + 8 * b + k);
						std::uint64_t const w(src[col * src_col_words + rw]);   // transpose_matrix.cc:79
						acc |= ((w >> (8 * byte_idx)) & 0xff) << (8 * k);
					}
					blocks[b] = transpose8x8(acc);
				}
				for (unsigned k(0); k < 8; ++k) {
					std::uint64_t out(0);
					for (unsigned b(0); b < 8; ++b)
						out |= ((blocks[b] >> (8 * k)) & 0xff) << (8 * b);
					dst[(64 * rw + 8 * byte_idx + k) * dst_col_words + cg] |= out;   // transpose_matrix.cc:93
				}
			}
		}
	}
	return 0;
}


// Definition of the transpose, bit by bit: dst(c, r) = src(r, c)
// (what tests/transpose_matrix.cc:170-179 builds as "expected").
int transpose_naive(std::uint64_t const *src, std::uint64_t nrows, std::uint64_t ncols, std::uint64_t *dst)
{
	if (0 == ncols) return 0;
	if (nrows % 64 || ncols % 64) return -1;
	std::fill(dst, dst + nrows * ncols / 64, 0);
	for (std::uint64_t c(0); c < ncols; ++c) {
		for (std::uint64_t rw(0); rw < nrows / 64; ++rw) {
			std::uint64_t w(src[c * (nrows / 64) + rw]);
			while (w) {
				unsigned const b(__builtin_ctzll(w));
				w &= w - 1;
				std::uint64_t const r(64 * rw + b);
				std::uint64_t const idx(r * ncols + c);   // dst has ncols rows; column r
				dst[idx >> 6] |= std::uint64_t(1) << (idx & 63);
			}
		}
	}
	return 0;
}


bit_matrix transpose_matrix(bit_matrix const &m)
{
	if (0 == m.cols) return bit_matrix{};
	bit_matrix dst(m.cols, m.rows);
	transpose_blocks(m.words.data(), m.rows, m.cols, dst.words.data());
	return dst;
}


// ---------------------------------------------------------------------------
// sequence_writer.hh:16-36 -- delegate with a virtual per-node hook.
// ---------------------------------------------------------------------------
struct sequence_writing_delegate {
	std::uint32_t chromosome_copy_index{kPloidyMax};
	virtual ~sequence_writing_delegate() {}
	virtual void handle_node(graph const &, std::uint64_t node) = 0;
};

// haplotype_output.cc:22-32 / founder_sequence_greedy_output.cc:72-75
struct fixed_copy_delegate final : sequence_writing_delegate {
	fixed_copy_delegate() = default;
	explicit fixed_copy_delegate(std::uint32_t copy) { chromosome_copy_index = copy; }
	void handle_node(graph const &, std::uint64_t) override {}
};

// founder_sequence_greedy_output.cc:78-115: when the walk visits the next cut
// node, switch to that segment's assigned copy.  A cut node that the walk jumps
// over is never matched again (the reference asserts node <= cut, :108).
struct founder_delegate final : sequence_writing_delegate {
	std::uint64_t const *cut_nodes{};
	std::uint32_t const *copies{};   // one per cut interval (stride 1: one matrix column)
	std::uint64_t n_cuts{};
	std::uint64_t idx{};

	void handle_node(graph const &, std::uint64_t node) override
	{
		if (idx < n_cuts && node == cut_nodes[idx]) {
			chromosome_copy_index = copies[idx];
			++idx;
		}
	}
};


// sequence_writer.cc:22-85.  Uses the same three stream write forms as the
// reference so that the timed CPU baseline carries the same ostream cost:
// operator<<(std::string) for labels (:62), operator<<(std::string_view) for
// reference parts (:73-74) and fill_n over an ostreambuf_iterator for gaps (:81).
void output_sequence(
	char const *ref_seq,
	graph const &g,
	std::ostream &stream,
	char const *fasta_identifier,
	bool should_output_unaligned,
	sequence_writing_delegate &delegate
)
{
	if (fasta_identifier)
		stream << '>' << fasta_identifier << '\n';                                  // :35-36

	std::uint64_t ref_pos(0), aln_pos(0), next_ref_pos(0), next_aln_pos(0), node(0);
	std::uint64_t const limit(g.node_count() - 1);
	auto const &paths(g.paths_by_chrom_copy_and_edge);
	while (node < limit) {                                                          // :44
		delegate.handle_node(g, node);                                              // :46
		std::uint64_t emitted(0);
		bool took_alt(false);
		if (kPloidyMax != delegate.chromosome_copy_index) {                         // :49
			std::uint64_t const e_end(g.alt_edge_count_csum[node + 1]);
			for (std::uint64_t e(g.alt_edge_count_csum[node]); e < e_end; ++e) {     // :51-52
				if (paths.get(e, delegate.chromosome_copy_index)) {                  // :54
					std::uint64_t const target(g.alt_edge_targets[e]);
					auto const &label(g.alt_edge_labels[e]);
					next_ref_pos = g.reference_positions[target];
					next_aln_pos = g.aligned_positions[target];
					stream << label;                                                 // :62
					node = target;
					emitted = label.size();
					took_alt = true;
					break;                                                           // first set edge wins
				}
			}
		}
		if (!took_alt) {                                                            // :70-77
			next_ref_pos = g.reference_positions[node + 1];
			next_aln_pos = g.aligned_positions[node + 1];
			std::string_view const ref_part(ref_seq + ref_pos, next_ref_pos - ref_pos);
			stream << ref_part;
			emitted = ref_part.size();
			++node;
		}
		if (!should_output_unaligned)                                               // :80-81
			std::fill_n(std::ostreambuf_iterator<char>(stream), next_aln_pos - aln_pos - emitted, '-');
		ref_pos = next_ref_pos;                                                     // :82-83
		aln_pos = next_aln_pos;
	}
}


// haplotype_output.cc:38-82: optional REF row, then sample-major, copy-minor;
// ids "[chr\t]REF" and "[chr\t]SAMPLE-N"; '\n' after every row body.
// first_copy/n_copies select a contiguous run of chromosome copies (the whole
// range for the reference's behaviour) so the baseline can time a bounded sample.
std::uint64_t haplotype_output_a2m(
	char const *ref_seq, graph const &g, std::ostream &stream,
	char const *chromosome_id, bool output_reference, bool unaligned,
	std::uint64_t first_copy, std::uint64_t n_copies
)
{
	std::uint64_t rows(0);
	if (output_reference) {                                                         // :48-59
		std::stringstream id;
		if (chromosome_id) id << chromosome_id << '\t';
		id << "REF";
		fixed_copy_delegate d;
		output_sequence(ref_seq, g, stream, id.str().data(), unaligned, d);
		stream << '\n';
		++rows;
	}
	for (std::size_t s(0); s < g.sample_names.size(); ++s) {                        // :62
		std::uint32_t const ploidy(g.ploidy_csum[s + 1] - g.ploidy_csum[s]);
		for (std::uint32_t c(0); c < ploidy; ++c) {                                 // :65
			std::uint64_t const copy(g.ploidy_csum[s] + c);                         // :28-31
			if (copy < first_copy || first_copy + n_copies <= copy) continue;
			std::stringstream id;
			if (chromosome_id) id << chromosome_id << '\t';
			id << g.sample_names[s] << '-' << (1 + c);                              // :69-72
			fixed_copy_delegate d(static_cast<std::uint32_t>(copy));
			output_sequence(ref_seq, g, stream, id.str().data(), unaligned, d);
			stream << '\n';                                                         // :76
			++rows;
		}
	}
	return rows;
}


// founder_sequence_greedy_output.cc:515-550.  assigned_samples is the
// (n_cuts-1) x n_founders matrix, one column per founder, column-major
// (pinned by tests/founder_sequences.cc:130 against the expected text).
void founder_output_a2m(
	char const *ref_seq, graph const &g, std::ostream &stream,
	char const *chromosome_id, bool output_reference, bool unaligned,
	std::uint64_t const *cut_positions, std::uint64_t n_cuts,
	std::uint32_t const *assigned_samples, std::uint64_t n_founders
)
{
	if (output_reference) {
		std::stringstream id;
		if (chromosome_id) id << chromosome_id << '\t';
		id << "REF";
		fixed_copy_delegate d;
		output_sequence(ref_seq, g, stream, id.str().data(), unaligned, d);
		stream << '\n';
	}
	std::uint64_t const col_rows(n_cuts ? n_cuts - 1 : 0);
	for (std::uint64_t f(0); f < n_founders; ++f) {
		std::stringstream id;
		if (chromosome_id) id << chromosome_id << '\t';
		id << (1 + f);                                                              // :542
		founder_delegate d;
		d.cut_nodes = cut_positions;
		d.copies = assigned_samples + f * col_rows;
		d.n_cuts = col_rows;   // the last cut position (the sink) has no segment of its own
		output_sequence(ref_seq, g, stream, id.str().data(), unaligned, d);
		stream << '\n';
	}
}


// ---------------------------------------------------------------------------
// Minimal readers standing in for libbio (absent): only what the call sites at
// main.cc:381 and variant_graph.cc:133-146,181 need for plain-text input.
// ---------------------------------------------------------------------------
bool read_single_fasta_sequence(char const *path, std::string &seq, char const *seq_id)
{
	std::ifstream is(path);
	if (!is) return false;
	std::string line;
	bool in_wanted(false), found(false);
	while (std::getline(is, line)) {
		if (!line.empty() && '\r' == line.back()) line.pop_back();
		if (!line.empty() && '>' == line[0]) {
			if (found) break;   // next record begins
			auto const end(line.find_first_of(" \t", 1));
			std::string const id(line.substr(1, std::string::npos == end ? std::string::npos : end - 1));
			in_wanted = (!seq_id) || id == seq_id;
			found = in_wanted;
			continue;
		}
		if (in_wanted) seq += line;
	}
	return found;
}


// ---------------------------------------------------------------------------
// Founder search: pBWT over the ALT edges, cut positions, greedy matching.
// Literal restatement -- every divergence-count update the reference makes is
// made here too; the product's host version (csrc/host/founder.cc) takes
// shortcuts and is tested against this one.
// ---------------------------------------------------------------------------

// pbwt.hh:32-49: divergence values order with DIVERGENCE_MAX first.
struct divergence_value {
	std::uint64_t value{};
	divergence_value() = default;
	divergence_value(std::uint64_t v) : value(v) {}
	bool operator<(divergence_value const other) const { return 1 + value < 1 + other.value; }
	operator std::uint64_t() const { return value; }
};

// pbwt.hh:22-145
struct pbwt_context {
	std::vector<std::uint32_t> permutation, prev_permutation;
	std::vector<divergence_value> divergence, prev_divergence;
	std::map<divergence_value, std::uint32_t> divergence_value_counts;

	explicit pbwt_context(std::uint32_t count) : permutation(count), divergence(count, divergence_value(UINT64_MAX))   // :62-73
	{
		if (count) {
			divergence[0] = 0;
			divergence_value_counts[0] = 1;
			if (1 < count) divergence_value_counts[UINT64_MAX] = count - 1;
		}
		for (std::uint32_t i(0); i < count; ++i) permutation[i] = i;
	}

	void swap_vectors()                                                       // :137-145
	{
		std::swap(permutation, prev_permutation);
		std::swap(divergence, prev_divergence);
		permutation.clear();
		divergence.clear();
	}

	// :76-134; `column` is one column of paths_by_edge_and_chrom_copy
	void update_divergence(bit_matrix const &m, std::uint64_t column, divergence_value const kk)
	{
		std::uint32_t zero_idx(0);
		std::uint32_t one_idx(std::uint32_t(prev_permutation.size()));
		for (std::uint64_t r(0); r < m.rows; ++r) one_idx -= m.get(r, column) ? 1 : 0;
		permutation.resize(prev_permutation.size());
		divergence.resize(prev_divergence.size());
		divergence_value pp(kk + 1), qq(kk + 1);
		for (std::size_t ii(0); ii < prev_permutation.size(); ++ii) {
			auto const val_idx(prev_permutation[ii]);
			auto const pd(prev_divergence[ii]);
			if (pp < pd) pp = pd;
			if (qq < pd) qq = pd;
			{
				auto const it(divergence_value_counts.find(pd));
				--it->second;
				if (0 == it->second) divergence_value_counts.erase(it);
			}
			if (!m.get(val_idx, column)) {
				++divergence_value_counts[pp];
				permutation[zero_idx] = val_idx;
				divergence[zero_idx] = pp;
				++zero_idx;
				pp = 0;
			} else {
				++divergence_value_counts[qq];
				permutation[one_idx] = val_idx;
				divergence[one_idx] = qq;
				++one_idx;
				qq = 0;
			}
		}
	}
};

constexpr std::uint32_t kCutPositionScoreMax = UINT32_MAX;   // find_cut_positions.hh (CUT_POSITION_SCORE_MAX)

// find_cut_positions.cc:29-63
struct cut_position {
	std::uint64_t edge{};
	std::uint64_t prev_edge{kEdgeMax};
	std::uint64_t node{};
	std::uint32_t score{};

	void update_if_needed(std::uint32_t eq_class_count, cut_position const &prev_cut)
	{
		auto const candidate_score(std::max(eq_class_count, prev_cut.score));
		if (candidate_score < score) {
			score = candidate_score;
			prev_edge = prev_cut.edge;
		}
	}
};

std::uint32_t total_chromosome_copies(graph const &g) { return g.ploidy_csum.empty() ? 0 : g.ploidy_csum.back(); }

// find_cut_positions.cc:93-211
std::uint32_t find_initial_cut_positions_lambda_min(graph const &g, std::uint64_t min_distance, std::vector<std::uint64_t> &out_cut_positions)
{
	out_cut_positions.clear();
	auto const path_count(total_chromosome_copies(g));
	std::uint64_t rightmost_seen_alt_edge_target(0), edge_idx(0), prev_cut_pos_id(kEdgeMax);
	pbwt_context pbwt_ctx(path_count);
	std::vector<cut_position> cut_positions;
	cut_positions.push_back({0, kEdgeMax, 0, 0});
	auto const by_edge([](cut_position const &lhs, std::uint64_t rhs) { return lhs.edge < rhs; });

	for (std::uint64_t node(0); node < g.node_count(); ++node) {               // variant_graph_walker
		if (rightmost_seen_alt_edge_target <= node && prev_cut_pos_id != edge_idx) {   // :126-129
			cut_positions.push_back({edge_idx, kEdgeMax, node, path_count});
			auto &current_cut(cut_positions.back());
			prev_cut_pos_id = edge_idx;

			auto const cut_pos_begin(cut_positions.begin());
			auto cut_pos_rb(cut_positions.end());
			auto const &dvc(pbwt_ctx.divergence_value_counts);
			auto eq_class_count(dvc.rbegin()->second);                          // :137
			// every entry but the last (largest), from the largest down (:115-121, :138)
			auto last(dvc.end()); --last;
			for (auto it(last); it != dvc.begin();) {
				--it;
				std::uint64_t const div_edge_idx(it->first);
				auto const pos(std::lower_bound(cut_pos_begin, cut_pos_rb, div_edge_idx, by_edge));
				if (pos != cut_pos_rb) {
					cut_pos_rb = pos;
					if (min_distance <= g.aligned_positions[node] - g.aligned_positions[pos->node])
						current_cut.update_if_needed(eq_class_count, *pos);
				}
				eq_class_count += it->second;
			}
			if (cut_pos_begin != cut_pos_rb) {                                  // :160-164
				--cut_pos_rb;
				current_cut.update_if_needed(eq_class_count, *cut_pos_rb);
			}
		}

		for (auto e(g.alt_edge_count_csum[node]); e < g.alt_edge_count_csum[node + 1]; ++e) {   // :169-176
			pbwt_ctx.swap_vectors();
			pbwt_ctx.update_divergence(g.paths_by_edge_and_chrom_copy, edge_idx, edge_idx);
			++edge_idx;
			rightmost_seen_alt_edge_target = std::max(rightmost_seen_alt_edge_target, g.alt_edge_targets[e]);
		}
	}

	if (cut_positions.size() <= 1) return kCutPositionScoreMax;               // :182-183

	auto it(cut_positions.cend() - 1);
	auto const retval(it->score);
	for (;;) {                                                                // :188-198
		out_cut_positions.push_back(it->node);
		auto const prev_edge(it->prev_edge);
		if (kEdgeMax == prev_edge) break;
		it = std::lower_bound(cut_positions.cbegin(), it, prev_edge, by_edge);
	}
	if (0 != out_cut_positions.back()) out_cut_positions.push_back(0);
	std::reverse(out_cut_positions.begin(), out_cut_positions.end());
	if (out_cut_positions.back() != g.node_count() - 1) out_cut_positions.back() = g.node_count() - 1;   // :205-207
	return retval;
}


// founder_sequence_greedy_output.cc:47-69
struct joined_path_eq_class {
	std::uint32_t lhs_rep{}, rhs_rep{}, size{};
	joined_path_eq_class(std::uint32_t l, std::uint32_t r) : lhs_rep(l), rhs_rep(r) {}
	explicit joined_path_eq_class(std::uint32_t r) : lhs_rep(kPloidyMax), rhs_rep(r) {}
	bool operator<(joined_path_eq_class const &other) const { return size < other.size; }
};

// founder_sequence_greedy_output.cc:154-512.  assigned_samples: (cuts - 1) rows x founder_count columns, column-major.
bool find_matchings(
	graph const &g, std::vector<std::uint64_t> const &cuts, std::uint32_t founder_count, bool should_keep_ref_edges,
	std::vector<std::uint32_t> &assigned_samples)
{
	if (cuts.size() < 2) return false;                                        // :162-166
	auto const copies(total_chromosome_copies(g));
	if (0 == copies) return false;

	std::size_t const n_rows(cuts.size() - 1);
	assigned_samples.assign(n_rows * founder_count, kPloidyMax);               // :170-172
	auto const assigned([&](std::size_t row, std::uint32_t col) -> std::uint32_t & { return assigned_samples[col * n_rows + row]; });

	std::multimap<std::uint32_t, std::uint32_t> assignments_by_eq_class;
	std::vector<char> reserved_assignments(copies, 0);
	std::vector<std::uint32_t> arbitrarily_connected_rhs;
	std::uint64_t edge_idx(0), prev_cut_edge_idx(0), cut_pair_edge_idx(0);
	std::vector<std::uint32_t> lhs_eq_classes(copies, kPloidyMax), rhs_eq_classes(copies, kPloidyMax);
	std::uint32_t lhs_distinct_eq_classes(0), rhs_distinct_eq_classes(0);
	std::vector<joined_path_eq_class> joined_path_eq_classes;
	bool lhs_first_path_is_ref(true), rhs_first_path_is_ref(true);
	std::uint32_t lhs_first_path_eq_class(0), rhs_first_path_eq_class(0);
	auto cut_pos_it(cuts.begin());
	++cut_pos_it;
	pbwt_context pbwt_ctx(copies);
	std::uint64_t cut_pos_idx(0);

	for (std::uint64_t node(0); node < g.node_count(); ++node) {
		if (cut_pos_it != cuts.end() && node == *cut_pos_it) {                  // :208
			std::swap(lhs_eq_classes, rhs_eq_classes);                          // :213-222
			std::fill(rhs_eq_classes.begin(), rhs_eq_classes.end(), kPloidyMax);
			lhs_distinct_eq_classes = rhs_distinct_eq_classes;
			lhs_first_path_eq_class = rhs_first_path_eq_class;
			rhs_distinct_eq_classes = 0;
			rhs_first_path_eq_class = pbwt_ctx.permutation.front();

			{
				std::uint32_t rep(kPloidyMax);                                  // :229-252
				joined_path_eq_classes.clear();
				for (std::size_t i(0); i < pbwt_ctx.permutation.size(); ++i) {
					auto const aa(pbwt_ctx.permutation[i]);
					std::uint64_t const dd(pbwt_ctx.divergence[i]);             // compared as a plain integer (:233)
					if (prev_cut_edge_idx < dd) { rep = aa; ++rhs_distinct_eq_classes; }
					rhs_eq_classes[aa] = rep;
					if (0 < cut_pos_idx) {
						if (cut_pair_edge_idx < dd) joined_path_eq_classes.emplace_back(lhs_eq_classes[aa], rep);
						++joined_path_eq_classes.back().size;
					}
				}
			}

			if (0 < cut_pos_idx) {
				std::sort(joined_path_eq_classes.begin(), joined_path_eq_classes.end());   // :256
				if (!should_keep_ref_edges && lhs_first_path_is_ref && rhs_first_path_is_ref)   // :259-264
					std::erase_if(joined_path_eq_classes, [&](auto const &c) { return lhs_first_path_eq_class == c.lhs_rep && rhs_first_path_eq_class == c.rhs_rep; });

				if (1 == cut_pos_idx) {                                         // :266-311
					auto remaining_founders(founder_count);
					auto remaining_reserved(std::min(remaining_founders, lhs_distinct_eq_classes));
					remaining_founders -= remaining_reserved;
					std::uint32_t founder_idx(0);
					auto const do_assign([&](joined_path_eq_class const &c) {
						assignments_by_eq_class.emplace(c.lhs_rep, founder_idx);
						assigned(0, founder_idx) = c.lhs_rep;
						++founder_idx;
					});
					for (auto c(joined_path_eq_classes.rbegin()); c != joined_path_eq_classes.rend(); ++c) {
						if (reserved_assignments[c->lhs_rep]) {
							if (remaining_founders) { --remaining_founders; do_assign(*c); }
						} else if (remaining_reserved) {
							--remaining_reserved;
							reserved_assignments[c->lhs_rep] = 1;
							do_assign(*c);
						}
					}
					// the reference spins here when there is no class at all; the oracle stops instead
					while (remaining_founders && !joined_path_eq_classes.empty())
						for (auto c(joined_path_eq_classes.rbegin()); c != joined_path_eq_classes.rend() && remaining_founders; ++c) {
							--remaining_founders;
							do_assign(*c);
						}
				}

				{                                                               // :321-437
					std::fill(reserved_assignments.begin(), reserved_assignments.end(), 0);
					arbitrarily_connected_rhs.clear();
					auto remaining_founders(founder_count);
					auto remaining_reserved(std::min(remaining_founders, rhs_distinct_eq_classes));
					remaining_founders -= remaining_reserved;

					auto const try_assign([&](joined_path_eq_class const &c) -> bool {
						auto const it(assignments_by_eq_class.find(c.lhs_rep));
						if (assignments_by_eq_class.end() == it) return false;
						auto const founder_idx(it->second);
						assignments_by_eq_class.erase(it);
						assigned(cut_pos_idx, founder_idx) = c.rhs_rep;
						return true;
					});
					auto const assign_arbitrary([&](std::uint32_t rhs_rep) {
						if (assignments_by_eq_class.empty()) return;            // asserted non-empty at :355
						auto const it(assignments_by_eq_class.begin());
						auto const founder_idx(it->second);
						assignments_by_eq_class.erase(it);
						assigned(cut_pos_idx, founder_idx) = rhs_rep;
					});

					bool is_first(true), leave(false);                          // steps 1-3 (:363-413)
					while (!leave) {
						bool did_assign(false);
						for (auto c(joined_path_eq_classes.rbegin()); c != joined_path_eq_classes.rend(); ++c) {
							if (reserved_assignments[c->rhs_rep]) {
								if (remaining_founders) {
									if (try_assign(*c)) { did_assign = true; --remaining_founders; }
								} else if (!is_first) {
									leave = true;                               // goto continue_subsequent_assignment
									break;
								}
							} else if (remaining_reserved) {
								--remaining_reserved;
								if (try_assign(*c)) reserved_assignments[c->rhs_rep] = 1;
								else arbitrarily_connected_rhs.push_back(c->rhs_rep);
							}
						}
						if (leave) break;
						if (!remaining_founders) break;
						if (is_first) { is_first = false; continue; }
						if (!did_assign) break;
					}

					for (auto const rhs_rep : arbitrarily_connected_rhs) {       // step 4 (:417-426)
						if (!reserved_assignments[rhs_rep]) {
							assign_arbitrary(rhs_rep);
							reserved_assignments[rhs_rep] = 1;
						}
					}

					while (!assignments_by_eq_class.empty() && !joined_path_eq_classes.empty())   // step 5 (:429-438)
						for (auto c(joined_path_eq_classes.rbegin()); c != joined_path_eq_classes.rend() && !assignments_by_eq_class.empty(); ++c)
							assign_arbitrary(c->rhs_rep);

					assignments_by_eq_class.clear();                            // :441-447
					for (std::uint32_t idx(0); idx < founder_count; ++idx)
						assignments_by_eq_class.insert({assigned(cut_pos_idx, idx), idx});
				}
			}

			++cut_pos_idx;                                                      // :451-457
			++cut_pos_it;
			cut_pair_edge_idx = prev_cut_edge_idx;
			prev_cut_edge_idx = edge_idx;
			lhs_first_path_is_ref = rhs_first_path_is_ref;
			rhs_first_path_is_ref = true;
		}

		for (auto e(g.alt_edge_count_csum[node]); e < g.alt_edge_count_csum[node + 1]; ++e) {   // :461-469
			pbwt_ctx.swap_vectors();
			pbwt_ctx.update_divergence(g.paths_by_edge_and_chrom_copy, edge_idx, edge_idx);
			rhs_first_path_is_ref = rhs_first_path_is_ref && !g.paths_by_edge_and_chrom_copy.get(pbwt_ctx.permutation.front(), edge_idx);
			++edge_idx;
		}
	}

	if (1 == cut_pos_idx) {                                                   // :475-508
		std::uint32_t rep(kPloidyMax);
		joined_path_eq_classes.clear();
		for (std::size_t i(0); i < pbwt_ctx.permutation.size(); ++i) {
			auto const aa(pbwt_ctx.permutation[i]);
			std::uint64_t const dd(pbwt_ctx.divergence[i]);
			if (0 < dd) {
				rep = aa;
				++rhs_distinct_eq_classes;
				joined_path_eq_classes.emplace_back(rep);
			}
			rhs_eq_classes[aa] = rep;
			++joined_path_eq_classes.back().size;
		}
		std::sort(joined_path_eq_classes.begin(), joined_path_eq_classes.end());
		if (!should_keep_ref_edges && rhs_first_path_is_ref)
			std::erase_if(joined_path_eq_classes, [&](auto const &c) { return rhs_first_path_eq_class == c.rhs_rep; });
		std::uint32_t founder_idx(0);
		for (auto c(joined_path_eq_classes.rbegin()); c != joined_path_eq_classes.rend() && founder_idx < founder_count; ++c, ++founder_idx)
			assigned(0, founder_idx) = c->rhs_rep;
	}
	return true;
}


std::vector<std::string_view> split(std::string_view s, char delim)
{
	std::vector<std::string_view> out;
	std::size_t pos(0);
	while (true) {
		auto const next(s.find(delim, pos));
		if (std::string_view::npos == next) { out.push_back(s.substr(pos)); break; }
		out.push_back(s.substr(pos, next - pos));
		pos = next + 1;
	}
	return out;
}


enum class alt_kind { sequence, deletion, unhandled };

// vcf::sv_type as consumed at variant_graph.cc:328-364: a plain base string is
// NONE, "<DEL>" is DEL, anything else (".", "*", "<CNV...>", breakends) is
// not turned into an edge.
alt_kind classify_alt(std::string_view alt)
{
	if (alt == "<DEL>") return alt_kind::deletion;
	if (alt.empty()) return alt_kind::unhandled;
	for (char c : alt) {
		switch (c) {
			case 'A': case 'C': case 'G': case 'T': case 'N':
			case 'a': case 'c': case 'g': case 't': case 'n':
				break;
			default:
				return alt_kind::unhandled;
		}
	}
	return alt_kind::sequence;
}


constexpr std::uint32_t kNullAllele = UINT32_MAX;

bool parse_gt(std::string_view field, std::vector<std::uint32_t> &alleles)
{
	alleles.clear();
	std::size_t pos(0);
	while (pos <= field.size()) {
		auto next(field.find_first_of("|/", pos));
		if (std::string_view::npos == next) next = field.size();
		auto const tok(field.substr(pos, next - pos));
		if (tok == ".") alleles.push_back(kNullAllele);
		else {
			if (tok.empty()) return false;
			std::uint32_t v(0);
			for (char c : tok) { if (c < '0' || '9' < c) return false; v = 10 * v + (c - '0'); }
			alleles.push_back(v);
		}
		pos = next + 1;
	}
	return true;
}


// variant_graph.cc:108-454.  REF mismatches are fatal.
// should_include(sample, copy) (build_graph_delegate, variant_graph.hh:143) is restated as one optional
// exclusion: every copy of `exclude_sample`, or only copy `exclude_copy` of it when that is >= 0.
bool build_variant_graph(std::string const &ref_seq, char const *vcf_path, char const *chr_id, graph &g, std::string &err,
	char const *exclude_sample = nullptr, int exclude_copy = -1)
{
	auto should_include([&](std::string const &sample, std::uint32_t copy) {
		if (!exclude_sample || sample != exclude_sample) return true;
		return exclude_copy >= 0 && int(copy) != exclude_copy;
	});
	std::ifstream is(vcf_path);
	if (!is) { err = "unable to open VCF"; return false; }

	g.alt_edge_count_csum.push_back(0);                                             // :147
	g.add_node(0, 0);                                                               // :148

	struct edge_destination { std::uint64_t edge_index; std::uint64_t position; };
	std::multimap<std::uint64_t, edge_destination> pending;                         // :157
	std::uint64_t aln_pos(0), prev_ref_pos(0), var_idx(0), lineno(0);
	bool is_first(true);
	std::vector<std::uint64_t> edges_by_alt, target_ref_positions_by_chrom_copy, current_edge_targets;
	std::vector<std::string> vcf_sample_names;
	struct included { std::uint32_t sample_in, sample_out, copy_in, copy_out; };
	std::vector<included> included_samples;
	std::vector<std::uint32_t> gt;

	auto add_target_nodes([&](std::uint64_t ref_pos) {                              // :160-179
		auto it(pending.begin());
		for (; it != pending.end() && it->first <= ref_pos; ++it) {
			aln_pos = std::max(aln_pos + (it->first - prev_ref_pos), it->second.position);
			auto const node(g.add_or_update_node(it->first, aln_pos));
			g.alt_edge_targets[it->second.edge_index] = node;
			prev_ref_pos = it->first;
		}
		pending.erase(pending.begin(), it);
	});

	std::string line;
	while (std::getline(is, line)) {
		++lineno;
		if (!line.empty() && '\r' == line.back()) line.pop_back();
		if (line.empty()) continue;
		if ('#' == line[0]) {
			if (line.rfind("#CHROM", 0) == 0) {
				auto const cols(split(line, '\t'));
				for (std::size_t i(9); i < cols.size(); ++i) vcf_sample_names.emplace_back(cols[i]);
				g.sample_names = vcf_sample_names;                                  // :146
			}
			continue;
		}

		++var_idx;
		auto const cols(split(line, '\t'));
		if (cols.size() < 8) { err = "malformed VCF record at line " + std::to_string(lineno); return false; }
		if (cols[0] != chr_id) { ++g.chr_id_mismatches; continue; }                  // :203-207
		if (cols.size() < 10) { err = "variant " + std::to_string(var_idx) + " does not have a genotype"; return false; } // :209-213

		std::size_t gt_field(SIZE_MAX);
		{
			auto const fmt(split(cols[8], ':'));
			for (std::size_t i(0); i < fmt.size(); ++i) if (fmt[i] == "GT") { gt_field = i; break; }
			if (SIZE_MAX == gt_field) { err = "variant " + std::to_string(var_idx) + " does not have a genotype"; return false; }
		}
		auto sample_gt([&](std::size_t sample_idx, std::vector<std::uint32_t> &dst) -> bool {
			auto const fields(split(cols[9 + sample_idx], ':'));
			if (fields.size() <= gt_field) return false;
			return parse_gt(fields[gt_field], dst);
		});

		if (is_first) {                                                             // :215-288
			is_first = false;
			g.ploidy_csum.assign(1 + g.sample_names.size(), 0);
			std::uint32_t out_idx(0);
			std::vector<std::uint32_t> removed_samples;   // no chromosome copy included
			for (std::size_t s(0); s < vcf_sample_names.size(); ++s) {
				if (!sample_gt(s, gt)) { err = "bad GT at line " + std::to_string(lineno); return false; }
				std::uint32_t included_count(0);
				for (std::uint32_t c(0); c < gt.size(); ++c) {
					if (should_include(vcf_sample_names[s], c)) {                   // :231
						included_samples.push_back({std::uint32_t(s), out_idx, c, included_count});
						++included_count;
					}
				}
				if (included_count) {                                               // :238-247
					g.ploidy_csum[1 + out_idx] = g.ploidy_csum[out_idx] + included_count;
					++out_idx;
				} else {
					removed_samples.push_back(std::uint32_t(s));
				}
			}
			if (!removed_samples.empty()) {                                         // :250-273
				g.ploidy_csum.resize(g.ploidy_csum.size() - removed_samples.size());
				removed_samples.push_back(UINT32_MAX);
				auto it(removed_samples.begin());
				std::vector<std::string> kept;
				for (std::size_t idx(0); idx < g.sample_names.size(); ++idx) {
					if (idx < *it) kept.push_back(g.sample_names[idx]);
					else ++it;
				}
				g.sample_names.swap(kept);
			}
			std::uint64_t const copies(g.ploidy_csum.back());
			std::uint64_t const rows(64 * ((copies + 63) / 64));                    // :277
			g.paths_by_edge_and_chrom_copy = rows ? bit_matrix(rows, 512) : bit_matrix(1, 0);
			target_ref_positions_by_chrom_copy.assign(copies, 0);
		}

		++g.handled_variants;
		std::uint64_t const ref_pos(std::stoull(std::string(cols[1])) - 1);         // zero_based_pos, :292
		if (ref_pos < prev_ref_pos) { err = "variant " + std::to_string(var_idx) + " has non-increasing position"; return false; } // :293-297

		add_target_nodes(ref_pos);                                                  // :300
		aln_pos += ref_pos - prev_ref_pos;                                          // :303-304
		g.add_or_update_node(ref_pos, aln_pos);                                     // :305

		auto const ref(cols[3]);
		if (ref_seq.size() < ref_pos + ref.size() || 0 != ref_seq.compare(ref_pos, ref.size(), ref)) {   // :307-314
			err = "REF column mismatch in variant " + std::to_string(var_idx);
			return false;
		}

		auto const alts(split(cols[4], ','));
		edges_by_alt.assign(alts.size(), kEdgeMax);                                 // :320-321
		current_edge_targets.clear();
		std::uint64_t min_edge(0), max_edge(0);
		bool first_edge(true);
		for (std::size_t a(0); a < alts.size(); ++a) {                              // :326-365
			auto const kind(classify_alt(alts[a]));
			if (alt_kind::unhandled == kind) continue;
			std::uint64_t const target_pos(ref_pos + ref.size());                   // :333
			std::uint64_t edge;
			if (alt_kind::sequence == kind) {
				edge = g.add_edge(alts[a]);
				pending.emplace(target_pos, edge_destination{edge, aln_pos + alts[a].size()});   // :338
			} else {
				edge = g.add_edge(std::string_view{});
				pending.emplace(target_pos, edge_destination{edge, aln_pos});       // :344
			}
			edges_by_alt[a] = edge;
			current_edge_targets.push_back(target_pos);
			if (first_edge) { min_edge = edge; first_edge = false; }
			max_edge = edge;
		}

		{                                                                           // :368-376
			auto &m(g.paths_by_edge_and_chrom_copy);
			if (m.cols && m.cols <= max_edge) {
				std::uint64_t const multiplier(4 + m.cols / 512);
				m.resize_total_bits(m.rows * multiplier * 512);
			}
		}

		for (auto const &inc : included_samples) {                                  // :379-425
			if (!sample_gt(inc.sample_in, gt) || gt.size() <= inc.copy_in) { err = "bad GT at line " + std::to_string(lineno); return false; }
			std::uint32_t const allele(gt[inc.copy_in]);
			if (0 == allele || kNullAllele == allele) continue;                     // :393-397
			if (edges_by_alt.size() < allele) { err = "GT allele out of range at line " + std::to_string(lineno); return false; }
			std::uint64_t const edge(edges_by_alt[allele - 1]);
			if (kEdgeMax == edge) continue;                                         // :401-403
			std::uint64_t const row(g.ploidy_csum[inc.sample_out] + inc.copy_out);
			if (ref_pos < target_ref_positions_by_chrom_copy[row])                  // :408-418
				g.overlaps.push_back({lineno, ref_pos, std::string(cols[2]), vcf_sample_names[inc.sample_in], inc.copy_in, allele});
			target_ref_positions_by_chrom_copy[row] = current_edge_targets[edge - min_edge];   // :422-423
			g.paths_by_edge_and_chrom_copy.set(row, edge);                          // :424 (set even when overlapping)
		}

		prev_ref_pos = ref_pos;                                                     // :427
	}

	{                                                                               // :437-443
		std::uint64_t const ref_pos(ref_seq.size());
		add_target_nodes(ref_pos);
		g.add_or_update_node(ref_pos, aln_pos + (ref_pos - prev_ref_pos));
	}

	if (!g.paths_by_edge_and_chrom_copy.words.empty()) {   // :445-451 (`if (graph.paths_by_edge_and_chrom_copy.size())`: the 1 x 0 matrix of a graph without chromosome copies, :280, stays as it is)
		auto &m(g.paths_by_edge_and_chrom_copy);
		std::uint64_t const ncol(64 * ((g.edge_count() + 63) / 64));
		m.resize_total_bits(m.rows * ncol);
	}
	g.paths_by_chrom_copy_and_edge = transpose_matrix(g.paths_by_edge_and_chrom_copy);   // :453
	g.finish_views();
	return true;
}


// An ostream that discards everything but still runs the full formatting path
// through a real streambuf with a put area (like the fd-backed stream the
// reference writes to, output.cc:62-64), counting the bytes.
class counting_null_buf final : public std::streambuf {
	char m_buf[1 << 16];
	std::uint64_t m_count{};
public:
	counting_null_buf() { setp(m_buf, m_buf + sizeof(m_buf)); }
	std::uint64_t count() { return m_count + (pptr() - pbase()); }
protected:
	int_type overflow(int_type ch) override
	{
		m_count += pptr() - pbase();
		setp(m_buf, m_buf + sizeof(m_buf));
		if (traits_type::eq_int_type(ch, traits_type::eof())) return traits_type::not_eof(ch);
		*pptr() = traits_type::to_char_type(ch);
		pbump(1);
		return ch;
	}
	std::streamsize xsputn(char const *s, std::streamsize n) override
	{
		std::streamsize left(n);
		while (left) {
			std::streamsize const room(epptr() - pptr());
			std::streamsize const k(std::min(room, left));
			std::memcpy(pptr(), s, k);
			pbump(int(k));
			s += k; left -= k;
			if (pptr() == epptr()) { m_count += pptr() - pbase(); setp(m_buf, m_buf + sizeof(m_buf)); }
		}
		return n;
	}
};


void set_err(char *err, std::size_t errlen, std::string const &msg)
{
	if (err && errlen) {
		std::snprintf(err, errlen, "%s", msg.c_str());
	}
}

} // namespace


// ===========================================================================
// C API for ctypes (tests/oracle.py).
// ===========================================================================
extern "C" {

typedef struct v2mo_graph v2mo_graph;

v2mo_graph *v2mo_build_variant_graph_ex(
	char const *fasta_path, char const *seq_id, char const *vcf_path, char const *chr_id,
	char const *exclude_sample, int exclude_copy,
	char **ref_out, std::uint64_t *ref_len_out, char *err, std::size_t errlen
);

v2mo_graph *v2mo_build_variant_graph(
	char const *fasta_path, char const *seq_id, char const *vcf_path, char const *chr_id,
	char **ref_out, std::uint64_t *ref_len_out, char *err, std::size_t errlen
)
{
	return v2mo_build_variant_graph_ex(fasta_path, seq_id, vcf_path, chr_id, nullptr, -1, ref_out, ref_len_out, err, errlen);
}

v2mo_graph *v2mo_build_variant_graph_ex(
	char const *fasta_path, char const *seq_id, char const *vcf_path, char const *chr_id,
	char const *exclude_sample, int exclude_copy,
	char **ref_out, std::uint64_t *ref_len_out, char *err, std::size_t errlen
)
{
	std::string ref;
	if (!read_single_fasta_sequence(fasta_path, ref, seq_id)) { set_err(err, errlen, "unable to read the reference sequence"); return nullptr; }
	auto *g(new graph);
	std::string msg;
	if (!build_variant_graph(ref, vcf_path, chr_id, *g, msg, exclude_sample, exclude_copy)) { set_err(err, errlen, msg); delete g; return nullptr; }
	if (ref_out) {
		char *buf(static_cast<char *>(std::malloc(ref.size() + 1)));
		std::memcpy(buf, ref.data(), ref.size());
		buf[ref.size()] = 0;
		*ref_out = buf;
	}
	if (ref_len_out) *ref_len_out = ref.size();
	return reinterpret_cast<v2mo_graph *>(g);
}

void v2mo_free(void *p) { std::free(p); }

// Build an oracle graph from flat arrays (labels as CSR, names NUL-separated).
// path words are paths_by_chrom_copy_and_edge (rows = edges, cols = copies).
v2mo_graph *v2mo_graph_from_arrays(
	std::uint64_t n_nodes, std::uint64_t n_edges,
	std::uint64_t const *ref_positions, std::uint64_t const *aln_positions,
	std::uint64_t const *edge_targets, std::uint64_t const *edge_count_csum,
	std::uint64_t const *label_offsets, char const *label_bytes,
	std::uint64_t const *path_words, std::uint64_t path_rows, std::uint64_t path_cols,
	std::uint64_t n_samples, char const *sample_name_blob, std::uint32_t const *ploidy_csum
)
{
	auto *g(new graph);
	g->reference_positions.assign(ref_positions, ref_positions + n_nodes);
	g->aligned_positions.assign(aln_positions, aln_positions + n_nodes);
	g->alt_edge_targets.assign(edge_targets, edge_targets + n_edges);
	g->alt_edge_count_csum.assign(edge_count_csum, edge_count_csum + n_nodes + 1);
	g->alt_edge_labels.reserve(n_edges);
	for (std::uint64_t e(0); e < n_edges; ++e)
		g->alt_edge_labels.emplace_back(label_bytes + label_offsets[e], label_offsets[e + 1] - label_offsets[e]);
	g->paths_by_chrom_copy_and_edge = bit_matrix(path_rows, path_cols);
	if (path_words)
		std::copy(path_words, path_words + path_rows * path_cols / 64, g->paths_by_chrom_copy_and_edge.words.begin());
	char const *p(sample_name_blob);
	for (std::uint64_t s(0); s < n_samples; ++s) { g->sample_names.emplace_back(p); p += g->sample_names.back().size() + 1; }
	if (ploidy_csum) g->ploidy_csum.assign(ploidy_csum, ploidy_csum + n_samples + 1);
	g->finish_views();
	return reinterpret_cast<v2mo_graph *>(g);
}

void v2mo_graph_free(v2mo_graph *h) { delete reinterpret_cast<graph *>(h); }

#define G(h) (*reinterpret_cast<graph *>(h))
std::uint64_t v2mo_node_count(v2mo_graph *h) { return G(h).node_count(); }
std::uint64_t v2mo_edge_count(v2mo_graph *h) { return G(h).edge_count(); }
std::uint64_t v2mo_sample_count(v2mo_graph *h) { return G(h).sample_names.size(); }
std::uint64_t const *v2mo_reference_positions(v2mo_graph *h) { return G(h).reference_positions.data(); }
std::uint64_t const *v2mo_aligned_positions(v2mo_graph *h) { return G(h).aligned_positions.data(); }
std::uint64_t const *v2mo_alt_edge_targets(v2mo_graph *h) { return G(h).alt_edge_targets.data(); }
std::uint64_t const *v2mo_alt_edge_count_csum(v2mo_graph *h) { return G(h).alt_edge_count_csum.data(); }
std::uint64_t const *v2mo_label_offsets(v2mo_graph *h) { return G(h).label_offsets.data(); }
char const *v2mo_label_bytes(v2mo_graph *h) { return G(h).label_bytes.data(); }
char const *v2mo_sample_name_blob(v2mo_graph *h) { return G(h).sample_name_blob.data(); }
std::uint64_t v2mo_sample_name_blob_size(v2mo_graph *h) { return G(h).sample_name_blob.size(); }
std::uint32_t const *v2mo_ploidy_csum(v2mo_graph *h) { return G(h).ploidy_csum.data(); }
std::uint64_t v2mo_handled_variants(v2mo_graph *h) { return G(h).handled_variants; }
std::uint64_t v2mo_chr_id_mismatches(v2mo_graph *h) { return G(h).chr_id_mismatches; }

// which: 0 = paths_by_chrom_copy_and_edge, 1 = paths_by_edge_and_chrom_copy
std::uint64_t const *v2mo_path_words(v2mo_graph *h, int which, std::uint64_t *rows, std::uint64_t *cols)
{
	auto const &m(which ? G(h).paths_by_edge_and_chrom_copy : G(h).paths_by_chrom_copy_and_edge);
	if (rows) *rows = m.rows;
	if (cols) *cols = m.cols;
	return m.words.data();
}

std::uint64_t v2mo_overlap_count(v2mo_graph *h) { return G(h).overlaps.size(); }
void v2mo_overlap_get(v2mo_graph *h, std::uint64_t i, std::uint64_t *lineno, std::uint64_t *ref_pos,
	char const **var_id, char const **sample, std::uint32_t *copy_idx, std::uint32_t *gt)
{
	auto const &o(G(h).overlaps[i]);
	*lineno = o.lineno; *ref_pos = o.ref_pos; *var_id = o.var_id.c_str(); *sample = o.sample_name.c_str();
	*copy_idx = o.chrom_copy_idx; *gt = o.gt;
}

int v2mo_transpose_matrix(std::uint64_t const *src, std::uint64_t nrows, std::uint64_t ncols, std::uint64_t *dst)
{
	return transpose_blocks(src, nrows, ncols, dst);
}

int v2mo_transpose_matrix_naive(std::uint64_t const *src, std::uint64_t nrows, std::uint64_t ncols, std::uint64_t *dst)
{
	return transpose_naive(src, nrows, ncols, dst);
}

// One row through output_sequence into a caller buffer.  Row selection:
// n_cuts == 0 -> constant copy_index (UINT32_MAX = REF); otherwise the founder
// delegate over (cut_nodes[i], cut_copies[i]).  Returns the byte count needed
// (written only if it fits).
std::int64_t v2mo_output_sequence(
	v2mo_graph *h, char const *ref_seq, char const *fasta_id, int unaligned,
	std::uint32_t copy_index, std::uint64_t const *cut_nodes, std::uint32_t const *cut_copies, std::uint64_t n_cuts,
	char *out, std::uint64_t out_cap
)
{
	std::ostringstream os;
	if (n_cuts) {
		founder_delegate d;
		d.cut_nodes = cut_nodes; d.copies = cut_copies; d.n_cuts = n_cuts;
		output_sequence(ref_seq, G(h), os, fasta_id, unaligned, d);
	} else {
		fixed_copy_delegate d(copy_index);
		output_sequence(ref_seq, G(h), os, fasta_id, unaligned, d);
	}
	auto const s(std::move(os).str());
	if (s.size() <= out_cap) std::memcpy(out, s.data(), s.size());
	return std::int64_t(s.size());
}

// Checksum of one row as the oracle writes it -- the formula of v2m_checksum_rows_device (include/v2m_hip.h): sum over the
// row's 8-byte little-endian words w (zero padded past the end) of mix64((w_index + 1) * GOLDEN ^ word), plus mix64(length).
// Lets full-size tests compare whole 100-250 MB rows without moving them through Python.  Thread-safe (the graph is only read).
std::uint64_t v2mo_row_checksum(
	v2mo_graph *h, char const *ref_seq, int unaligned,
	std::uint32_t copy_index, std::uint64_t const *cut_nodes, std::uint32_t const *cut_copies, std::uint64_t n_cuts,
	std::uint64_t *length_out
)
{
	// output_sequence() writes into a stream whose buffer folds the bytes into the checksum as they arrive (a 100-250 MB row is never
	// held: sixteen threads each growing a string of that size spent their time in the allocator, not in the walk)
	class checksum_streambuf final : public std::streambuf {
	public:
		checksum_streambuf() { setp(m_buf, m_buf + sizeof(m_buf)); }
		std::uint64_t finish(std::uint64_t &length)
		{
			take(true);
			length = m_total;
			return m_acc + mix64(m_total);
		}
	private:
		static std::uint64_t mix64(std::uint64_t z)
		{
			z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
			z ^= z >> 27; z *= 0x94D049BB133111EBULL;
			z ^= z >> 31;
			return z;
		}
		// whole 8-byte little-endian words of what has been written so far; the (< 8) bytes left over move to the front, and at
		// the row's end they make the last word, zero padded
		void take(bool last)
		{
			std::size_t const n(std::size_t(pptr() - pbase()));
			std::size_t const whole(n / 8);
			for (std::size_t w(0); w < whole; ++w) {
				std::uint64_t v;
				std::memcpy(&v, m_buf + 8 * w, 8);                                   // little-endian host
				m_acc += mix64((++m_words) * 0x9E3779B97F4A7C15ULL ^ v);
			}
			std::size_t const rest(n - 8 * whole);
			m_total += 8 * whole;
			if (last && rest) {
				std::uint64_t v(0);
				std::memcpy(&v, m_buf + 8 * whole, rest);
				m_acc += mix64((++m_words) * 0x9E3779B97F4A7C15ULL ^ v);
				m_total += rest;
				setp(m_buf, m_buf + sizeof(m_buf));
				return;
			}
			std::memmove(m_buf, m_buf + 8 * whole, rest);
			setp(m_buf, m_buf + sizeof(m_buf));
			pbump(int(rest));
		}
		int_type overflow(int_type ch) override
		{
			take(false);
			if (!traits_type::eq_int_type(ch, traits_type::eof())) { *pptr() = traits_type::to_char_type(ch); pbump(1); }
			return traits_type::not_eof(ch);
		}
		char m_buf[1 << 16];
		std::uint64_t m_acc{}, m_words{}, m_total{};
	};
	checksum_streambuf buf;
	std::ostream os(&buf);
	if (n_cuts) {
		founder_delegate d;
		d.cut_nodes = cut_nodes; d.copies = cut_copies; d.n_cuts = n_cuts;
		output_sequence(ref_seq, G(h), os, nullptr, unaligned, d);
	} else {
		fixed_copy_delegate d(copy_index);
		output_sequence(ref_seq, G(h), os, nullptr, unaligned, d);
	}
	std::uint64_t n(0);
	std::uint64_t const acc(buf.finish(n));
	if (length_out) *length_out = n;
	return acc;
}

// dst_path == NULL -> discard through the counting null stream (timed baseline).
// Returns bytes written, or -1.  seconds_out (optional) receives the wall time
// of the output loop alone.
std::int64_t v2mo_haplotype_output_a2m(
	v2mo_graph *h, char const *ref_seq, char const *chromosome_id, int output_reference, int unaligned,
	std::uint64_t first_copy, std::uint64_t n_copies, char const *dst_path, double *seconds_out
)
{
	auto const t0(std::chrono::steady_clock::now());
	std::int64_t bytes(-1);
	if (dst_path) {
		std::ofstream os(dst_path, std::ios::binary | std::ios::trunc);
		if (!os) return -1;
		haplotype_output_a2m(ref_seq, G(h), os, chromosome_id, output_reference, unaligned, first_copy, n_copies);
		os.flush();
		bytes = std::int64_t(os.tellp());
	} else {
		counting_null_buf buf;
		std::ostream os(&buf);
		haplotype_output_a2m(ref_seq, G(h), os, chromosome_id, output_reference, unaligned, first_copy, n_copies);
		bytes = std::int64_t(buf.count());
	}
	auto const t1(std::chrono::steady_clock::now());
	if (seconds_out) *seconds_out = std::chrono::duration<double>(t1 - t0).count();
	return bytes;
}

std::int64_t v2mo_founder_output_a2m(
	v2mo_graph *h, char const *ref_seq, char const *chromosome_id, int output_reference, int unaligned,
	std::uint64_t const *cut_positions, std::uint64_t n_cuts,
	std::uint32_t const *assigned_samples, std::uint64_t n_founders,
	char *out, std::uint64_t out_cap
)
{
	std::ostringstream os;
	founder_output_a2m(ref_seq, G(h), os, chromosome_id, output_reference, unaligned, cut_positions, n_cuts, assigned_samples, n_founders);
	auto const s(std::move(os).str());
	if (s.size() <= out_cap) std::memcpy(out, s.data(), s.size());
	return std::int64_t(s.size());
}

// find_cut_positions + find_matchings (founder_sequence_greedy_output.cc:139-152, 154-512).  cuts_out: node_count
// entries; assigned_out: (cuts - 1) * founder_count entries, column-major.  Returns the number of cut positions,
// 0 when there is no solution.  A graph made by v2mo_graph_from_arrays gets its copies-by-edge matrix here.
std::uint64_t v2mo_find_founders(
	v2mo_graph *h, std::uint64_t min_distance, std::uint32_t founder_count, int keep_ref_edges,
	std::uint64_t *cuts_out, std::uint32_t *assigned_out, std::uint64_t assigned_capacity, std::uint32_t *score_out
)
{
	auto &g(G(h));
	if (g.paths_by_edge_and_chrom_copy.words.empty() && !g.paths_by_chrom_copy_and_edge.words.empty())
		g.paths_by_edge_and_chrom_copy = transpose_matrix(g.paths_by_chrom_copy_and_edge);
	std::vector<std::uint64_t> cuts;
	auto const score(find_initial_cut_positions_lambda_min(g, min_distance, cuts));
	if (score_out) *score_out = score;
	if (kCutPositionScoreMax == score) return 0;
	std::vector<std::uint32_t> assigned;
	if (!find_matchings(g, cuts, founder_count, 0 != keep_ref_edges, assigned)) return 0;
	if (assigned.size() > assigned_capacity) return 0;
	std::copy(cuts.begin(), cuts.end(), cuts_out);
	std::copy(assigned.begin(), assigned.end(), assigned_out);
	return cuts.size();
}

} // extern "C"
