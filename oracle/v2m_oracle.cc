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

	if (!g.paths_by_edge_and_chrom_copy.words.empty() || g.paths_by_edge_and_chrom_copy.rows) {   // :445-451
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

} // extern "C"
