// variant_graph.hh -- host-side variant graph of the MI355X build.
//
// Same information as the reference's variant_graph (include/vcf2multialign/variant_graph.hh:57-66)
// in the layout the device wants: 64-bit integers as in the reference, ALT labels as one byte pool
// with CSR offsets (instead of a vector of strings), path matrices as plain column-major 64-bit
// words (rows padded to 64, LSB-first), so that v2m_upload_graph() takes the arrays as they are.
#pragma once

#include <cstdint>
#include <memory>
#include <new>
#include <string>
#include <string_view>
#include <type_traits>
#include <utility>
#include <vector>

namespace v2m::host {

typedef std::uint32_t u32;
typedef std::uint64_t u64;

constexpr u32 kPloidyMax = UINT32_MAX;   // variant_graph.hh:55
constexpr u64 kEdgeMax = UINT64_MAX;     // variant_graph.hh:53

// Column-major bit matrix; one column = rows/64 consecutive words; rows is a multiple of 64.
// std::allocator, except that resize(n) without a value leaves the new elements as they are instead of zeroing them:
// a matrix that is about to be overwritten completely (read_graph: 0.6 GB per matrix at BASELINE config 3) then is not
// written twice, and its pages are first touched by the threads that fill it.
template <typename T>
struct default_init_allocator : std::allocator<T> {
	template <typename U> struct rebind { using other = default_init_allocator<U>; };
	using std::allocator<T>::allocator;
	template <typename U> void construct(U *p) noexcept(std::is_nothrow_default_constructible_v<U>) { ::new (static_cast<void *>(p)) U; }
	template <typename U, typename... Args> void construct(U *p, Args &&... args) { ::new (static_cast<void *>(p)) U(std::forward<Args>(args)...); }
};

struct bit_matrix {
	u64 rows{};
	u64 cols{};
	std::vector<u64, default_init_allocator<u64>> words;

	bit_matrix() = default;
	bit_matrix(u64 rows_, u64 cols_) : rows(rows_), cols(cols_), words(rows_ / 64 * cols_, 0) {}
	// a matrix whose words the caller is going to overwrite, all of them
	static bit_matrix for_overwrite(u64 rows_, u64 cols_) { bit_matrix m; m.rows = rows_; m.cols = cols_; m.words.resize(rows_ / 64 * cols_); return m; }

	u64 words_per_column() const { return rows / 64; }
	bool test(u64 r, u64 c) const { return (words[c * (rows / 64) + (r >> 6)] >> (r & 63)) & 1; }
	void set(u64 r, u64 c) { words[c * (rows / 64) + (r >> 6)] |= u64(1) << (r & 63); }
	void set_column_count(u64 n) { words.resize(rows / 64 * n, 0); cols = n; }
};

struct variant_graph {
	std::vector<u64> reference_positions;   // by node
	std::vector<u64> aligned_positions;     // by node (MSA co-ordinates)
	std::vector<u64> alt_edge_targets;      // by edge
	std::vector<u64> alt_edge_count_csum;   // by 1-based node: edges of node n are [csum[n], csum[n+1])
	std::vector<u64> alt_edge_label_offsets{0};   // CSR over alt_edge_label_bytes, [edge_count + 1]
	std::string alt_edge_label_bytes;
	bit_matrix paths_by_chrom_copy_and_edge;   // rows = edges, cols = chromosome copies
	bit_matrix paths_by_edge_and_chrom_copy;   // rows = chromosome copies, cols = edges
	std::vector<std::string> sample_names;
	std::vector<u32> ploidy_csum;

	u64 node_count() const { return reference_positions.size(); }
	u64 edge_count() const { return alt_edge_targets.size(); }
	u64 aligned_length() const { return aligned_positions.empty() ? 0 : aligned_positions.back(); }
	u32 sample_ploidy(u64 sample) const { return ploidy_csum[sample + 1] - ploidy_csum[sample]; }
	u32 total_chromosome_copies() const { return ploidy_csum.empty() ? 0 : ploidy_csum.back(); }
	std::string_view label(u64 e) const
	{
		return std::string_view(alt_edge_label_bytes).substr(alt_edge_label_offsets[e], alt_edge_label_offsets[e + 1] - alt_edge_label_offsets[e]);
	}
};

} // namespace v2m::host
