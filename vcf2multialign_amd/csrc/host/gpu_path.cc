#include "gpu_path.hh"

#include <cstdlib>
#include <exception>

#include <algorithm>

namespace v2m::host {

gpu_context::gpu_context(int device)
{
	int const rc(v2m_ctx_create(device, &m_ctx));
	if (V2M_OK != rc) throw gpu_error(rc, v2m_last_error(nullptr));
}

gpu_context::~gpu_context() { if (m_owned) v2m_ctx_destroy(m_ctx); }

void gpu_context::check(int rc) const
{
	if (V2M_OK != rc) throw gpu_error(rc, v2m_last_error(m_ctx));
}


bit_matrix transpose_matrix(gpu_context &gpu, bit_matrix const &mat)
{
	if (0 == mat.cols) return bit_matrix{};                           // transpose_matrix.cc:48-49
	bit_matrix dst(mat.cols, mat.rows);
	gpu.check(v2m_transpose_bits(gpu.get(), mat.words.data(), mat.rows, mat.cols, dst.words.data()));
	return dst;
}


void transpose_paths(gpu_context &gpu, variant_graph &graph)
{
	graph.paths_by_chrom_copy_and_edge = transpose_matrix(gpu, graph.paths_by_edge_and_chrom_copy);
}


copy_shard shard_copies(u64 n_copies, u32 world, u32 rank)
{
	constexpr u64 granule(8);
	u64 const n_blocks((n_copies + granule - 1) / granule), base(n_blocks / world), extra(n_blocks % world);
	u64 const first_heavy(world - extra);
	u64 const b0(rank * base + (rank > first_heavy ? rank - first_heavy : 0));
	u64 const b1(b0 + base + (rank >= first_heavy ? 1 : 0));
	return {std::min(n_copies, granule * b0), std::min(n_copies, granule * b1)};
}


void upload_path_slice(gpu_context &gpu, variant_graph const &g, copy_shard shard)
{
	auto const &m(g.paths_by_edge_and_chrom_copy);
	gpu.check(v2m_upload_path_slice(gpu.get(), m.words.empty() ? nullptr : m.words.data(), m.rows, m.cols, shard.first, shard.end - shard.first));
}


void upload_path_blocks(gpu_context &gpu, variant_graph const &g, copy_interleave deal, u32 rank)
{
	auto const &m(g.paths_by_edge_and_chrom_copy);
	u64 const first(std::min<u64>(m.rows, deal.block * rank));
	gpu.check(v2m_upload_path_blocks(gpu.get(), m.words.empty() ? nullptr : m.words.data(), m.rows, m.cols, first, deal.block, deal.block * deal.world, m.rows));
}


void gpu_founder_walker::walk(u64 n_copies, u64 min_distance, std::vector<u32> const &cand_edge, std::vector<u64> const &cand_aligned,
	std::vector<u64> const &chunk_first, u32 const *start_order, u32 const *start_divergence,
	u64 capacity, u32 *trial_pred, u32 *trial_class, u64 *trial_end, u32 *status)
{
	m_gpu.check(v2m_pbwt_cut_trials(m_gpu.get(), n_copies, min_distance, cand_edge.size(), cand_edge.data(), cand_aligned.data(),
		chunk_first.size() - 1, chunk_first.data(), start_order, start_divergence, capacity, trial_pred, trial_class, trial_end, status));
}


void gpu_founder_walker::walk_streamed(u64 n_copies, u64 min_distance, std::vector<u32> const &cand_edge, std::vector<u64> const &cand_aligned,
	std::vector<u64> const &chunk_first, u32 const *start_order, u32 const *start_divergence,
	u64 capacity, u64 *trial_end, u32 *status, chunk_taker const &take)
{
	if (char const *const e = std::getenv("V2M_FOUNDER_ARRAYS")) if (*e && '0' != *e) {   // test knob: the array form of the call (v2m_pbwt_cut_trials), then the hand-over
		founder_walker::walk_streamed(n_copies, min_distance, cand_edge, cand_aligned, chunk_first, start_order, start_divergence, capacity, trial_end, status, take);
		return;
	}
	// the library calls back through C: what `take` throws is parked and thrown again once the call has returned
	struct relay { chunk_taker const &take; std::exception_ptr error; } state{take, nullptr};
	int const rc(v2m_pbwt_cut_trials_streamed(m_gpu.get(), n_copies, min_distance, cand_edge.size(), cand_edge.data(), cand_aligned.data(),
		chunk_first.size() - 1, chunk_first.data(), start_order, start_divergence, capacity, trial_end, status,
		[](void *user, uint64_t chunk, uint32_t chunk_status, uint32_t const *pred, uint32_t const *class_count, uint64_t n_pairs) -> int {
			relay &r(*static_cast<relay *>(user));
			try { r.take(std::size_t(chunk), chunk_status, pred, class_count, n_pairs); }
			catch (...) { r.error = std::current_exception(); return 1; }
			return 0;
		}, &state));
	if (state.error) std::rethrow_exception(state.error);
	m_gpu.check(rc);
}


void gpu_founder_walker::records(u64 n_copies, std::vector<u32> const &cut_edge, std::vector<u64> const &chunk_first_cut, std::vector<u32> const &start_edge,
	u32 const *start_order, u32 const *start_divergence, u64 pool_capacity, u32 *pool_lhs, u32 *pool_rhs, u32 *pool_size,
	u64 *rec_pool_end, u32 *rec_distinct, u32 *rec_first_class, u32 *rec_first_is_ref, u32 *status)
{
	m_gpu.check(v2m_pbwt_cut_records(m_gpu.get(), n_copies, cut_edge.size(), cut_edge.data(), chunk_first_cut.size() - 1, chunk_first_cut.data(), start_edge.data(),
		start_order, start_divergence, pool_capacity, pool_lhs, pool_rhs, pool_size, rec_pool_end, rec_distinct, rec_first_class, rec_first_is_ref, status));
}


void warm_up_sink(gpu_context &gpu, bool unaligned)
{
	// six REF rows into a sink that drops them: more than one slice of the sink path's default 512-MB slots whenever two
	// slices can occur at all, so both device slots and both pinned slots are allocated at the size later calls use
	u32 const copies[6] = {kPloidyMax, kPloidyMax, kPloidyMax, kPloidyMax, kPloidyMax, kPloidyMax};
	v2m_row_batch batch{};
	batch.n_rows = 6;
	batch.copy_index = copies;
	gpu.check(v2m_splice_rows(gpu.get(), &batch, unaligned ? V2M_SPLICE_UNALIGNED : 0u, [](void *, uint64_t, char const *, uint64_t) -> int { return 0; }, nullptr));
}


void upload_graph(gpu_context &gpu, sequence_type const &ref_seq, variant_graph const &g, bool with_paths)
{
	v2m_graph_view view{};
	view.node_count = g.node_count();
	view.edge_count = g.edge_count();
	view.reference_positions = g.reference_positions.data();
	view.aligned_positions = g.aligned_positions.data();
	view.alt_edge_targets = g.alt_edge_targets.data();
	view.alt_edge_count_csum = g.alt_edge_count_csum.data();
	view.alt_edge_label_offsets = g.alt_edge_label_offsets.data();
	view.alt_edge_label_bytes = g.alt_edge_label_bytes.data();
	auto const &paths(g.paths_by_chrom_copy_and_edge);
	view.paths_by_chrom_copy_and_edge = (!with_paths || paths.words.empty()) ? nullptr : paths.words.data();
	view.path_rows = with_paths ? paths.rows : 0;
	view.path_cols = with_paths ? paths.cols : 0;
	gpu.check(v2m_upload_graph(gpu.get(), &view, ref_seq.data(), ref_seq.size()));
}

} // namespace v2m::host
