// gpu_path.hh -- the host's view of the GPU hot path: thin C++ wrappers over the C ABI (include/v2m_hip.h).
#pragma once

#include <stdexcept>
#include <string>

#include "../../../include/v2m_hip.h"
#include "founder.hh"
#include "readers.hh"
#include "variant_graph.hh"

namespace v2m::host {

struct gpu_error : std::runtime_error {
	int code;
	gpu_error(int code_, std::string const &what) : std::runtime_error(what), code(code_) {}
};

class gpu_context {
public:
	explicit gpu_context(int device = 0);
	struct borrowed {};
	gpu_context(v2m_ctx *ctx, borrowed) : m_ctx(ctx), m_owned(false) {}   // somebody else's context (the Python binding's, in tests)
	~gpu_context();
	gpu_context(gpu_context const &) = delete;
	gpu_context &operator=(gpu_context const &) = delete;
	v2m_ctx *get() const { return m_ctx; }
	void check(int rc) const;   // throws gpu_error with v2m_last_error()
private:
	v2m_ctx *m_ctx{};
	bool m_owned{true};
};

// transpose_matrix (include/vcf2multialign/transpose_matrix.hh:14) on the GPU.
[[nodiscard]] bit_matrix transpose_matrix(gpu_context &gpu, bit_matrix const &mat);

// The last step of build_variant_graph (variant_graph.cc:453).
void transpose_paths(gpu_context &gpu, variant_graph &graph);

// Makes `graph` + `ref_seq` the resident graph of the context.  with_paths = false leaves the path matrix out (it then
// comes from upload_path_slice()).
void upload_graph(gpu_context &gpu, sequence_type const &ref_seq, variant_graph const &graph, bool with_paths = true);

// Pays the sink path's one-off costs ahead of time (a gigabyte of pinned host memory: 0.15 s of hipHostMalloc, the device
// slots, the templates) by splicing a few REF rows into a sink that drops them.  For callers that have something else
// to do before the first real row (the founder search); needs an uploaded graph.
void warm_up_sink(gpu_context &gpu, bool unaligned);

// The chromosome copies [first, end) a GPU owns when `n_copies` copies are sharded over `world` GPUs (SURVEY.md section 8e;
// the same rule as vcf2multialign_amd/sharding.py:shard_copies): contiguous blocks of whole bytes of the bit-packed path
// matrix (8 copies), the blocks that do not divide evenly go to the last GPUs because the first one also carries the REF row.
struct copy_shard { u64 first{}, end{}; };
copy_shard shard_copies(u64 n_copies, u32 world, u32 rank);

// This GPU's copies [first, end) of graph.paths_by_edge_and_chrom_copy: uploaded (only those bytes), transposed on the
// GPU and bound as the resident graph's path matrix (v2m_upload_path_slice).  Row batches for this context then use copy
// indices relative to `first`.  Nothing comes back to the host.
void upload_path_slice(gpu_context &gpu, variant_graph const &graph, copy_shard shard);

// Chromosome copies dealt to `world` GPUs in blocks of `block` copies, round-robin (v2m_upload_path_blocks): GPU `rank` owns the
// copies c with (c / block) % world == rank.  This is the sharding for outputs that have to leave in row order -- pipes,
// unaligned A2M, one file per sequence with a delegate that counts them -- where contiguous shards would leave all GPUs but
// one waiting for their turn: with blocks dealt round-robin every GPU always has rows that are due soon.
struct copy_interleave {
	u64 block{8};     // copies per block: a multiple of 8 (whole bytes of the bit-packed columns)
	u32 world{1};
	u32 owner(u64 copy) const { return u32((copy / block) % world); }
	u64 local(u64 copy) const { return (copy / (block * world)) * block + copy % block; }   // the copy's index in its owner's matrix
};
void upload_path_blocks(gpu_context &gpu, variant_graph const &graph, copy_interleave deal, u32 rank);

// The chunk walks of the founder searches on the GPU (v2m_pbwt_cut_trials, v2m_pbwt_cut_records).  The context must hold the
// uploaded graph with its path matrix (upload_graph(..., true)).
class gpu_founder_walker final : public founder_walker {
public:
	explicit gpu_founder_walker(gpu_context &gpu) : m_gpu(gpu) {}
	u64 max_copies() const override { return 20480; }          // v2m_pbwt_cut_trials: csrc/founder_kernels.hpp kPbwtMaxCopies
	u64 max_copies_records() const override { return 20480; }  // v2m_pbwt_cut_records: kPbwtMaxCopiesRecords (its class arrays in LDS up to 12 288 copies, in device memory above)
	std::size_t preferred_chunks() const override { return 256; }   // one workgroup per chunk, 1024 threads each: one round over 256 CUs
	void walk(u64 n_copies, u64 min_distance, std::vector<u32> const &cand_edge, std::vector<u64> const &cand_aligned,
		std::vector<u64> const &chunk_first, u32 const *start_order, u32 const *start_divergence,
		u64 capacity, u32 *trial_pred, u32 *trial_class, u64 *trial_end, u32 *status) override;
	void walk_streamed(u64 n_copies, u64 min_distance, std::vector<u32> const &cand_edge, std::vector<u64> const &cand_aligned,
		std::vector<u64> const &chunk_first, u32 const *start_order, u32 const *start_divergence,
		u64 capacity, u64 *trial_end, u32 *status, chunk_taker const &take) override;   // v2m_pbwt_cut_trials_streamed
	void records(u64 n_copies, std::vector<u32> const &cut_edge, std::vector<u64> const &chunk_first_cut, std::vector<u32> const &start_edge,
		u32 const *start_order, u32 const *start_divergence, u64 pool_capacity, u32 *pool_lhs, u32 *pool_rhs, u32 *pool_size,
		u64 *rec_pool_end, u32 *rec_distinct, u32 *rec_first_class, u32 *rec_first_is_ref, u32 *status) override;
private:
	gpu_context &m_gpu;
};
typedef gpu_founder_walker gpu_cut_trial_walker;

} // namespace v2m::host
