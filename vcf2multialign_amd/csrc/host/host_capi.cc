// host_capi.cc -- C API over the host-side graph builder so that the Python tests can compare it with the
// oracle's builder array by array (no GPU involved: the transpose is not part of the host builder).

#include <algorithm>
#include <cstring>
#include <stdexcept>
#include <string>

#include "founder.hh"
#include "gpu_path.hh"
#include "graph_file.hh"
#include "readers.hh"

namespace vh = v2m::host;

namespace {

struct recorded_overlap { vh::u64 lineno, ref_pos; std::string var_id, sample; vh::u32 copy, gt; };

struct host_graph {
	vh::variant_graph graph;
	vh::sequence_type ref;
	vh::build_graph_statistics stats;
	std::vector<recorded_overlap> overlaps;
	std::string sample_blob;
};

struct recording_delegate final : vh::build_graph_delegate {
	host_graph *hg{};
	std::string excluded_sample;   // optional: exclude every copy of this sample
	int excluded_copy{-1};         // or only this copy
	bool should_include(std::string_view sample_name, vh::u32 copy) const override
	{
		if (excluded_sample.empty() || sample_name != excluded_sample) return true;
		return excluded_copy >= 0 && int(copy) != excluded_copy;
	}
	void report_overlapping_alternative(vh::u64 lineno, vh::u64 ref_pos, std::string_view var_id, std::string_view sample_name, vh::u32 copy, vh::u32 gt) override
	{
		hg->overlaps.push_back({lineno, ref_pos, std::string(var_id), std::string(sample_name), copy, gt});
	}
	bool stop_at_mismatch{};       // what the reference's delegate may also answer: stop parsing here (variant_graph.cc:312-313)
	bool ref_column_mismatch(vh::u64, vh::u64, std::string_view, std::string_view) override
	{
		if (stop_at_mismatch) return false;
		throw std::runtime_error("REF column mismatch");
	}
};

bool g_stop_at_mismatch(false);

} // namespace

extern "C" {

void *v2mh_build_variant_graph(char const *fasta, char const *seq_id, char const *vcf, char const *chr, char const *exclude_sample, int exclude_copy, unsigned threads, char *err, size_t errlen)
{
	auto *hg(new host_graph);
	try {
		if (!vh::read_single_fasta_sequence(fasta, hg->ref, seq_id)) throw std::runtime_error("unable to read the reference sequence");
		recording_delegate d;
		d.hg = hg;
		if (exclude_sample) { d.excluded_sample = exclude_sample; d.excluded_copy = exclude_copy; }
		d.stop_at_mismatch = g_stop_at_mismatch;
		vh::build_variant_graph(hg->ref, vcf, chr, hg->graph, hg->stats, d, threads);
		for (auto const &s : hg->graph.sample_names) { hg->sample_blob += s; hg->sample_blob.push_back('\0'); }
		return hg;
	} catch (std::exception const &e) {
		if (err && errlen) { std::strncpy(err, e.what(), errlen - 1); err[errlen - 1] = 0; }
		delete hg;
		return nullptr;
	}
}

// A host graph from flat arrays (e.g. a synthetic dataset whose genotype matrix was generated on the GPU):
// paths_by_edge_and_chrom_copy words are column-major, rows = path_rows copies (multiple of 64), cols = path_cols edges.
void *v2mh_graph_from_arrays(
	uint64_t n_nodes, uint64_t n_edges, uint64_t const *ref_pos, uint64_t const *aln_pos, uint64_t const *targets, uint64_t const *csum,
	uint64_t const *label_offsets, char const *label_bytes, uint64_t const *paths_by_edge_and_chrom_copy, uint64_t path_rows, uint64_t path_cols,
	uint32_t n_samples, uint32_t ploidy)
{
	auto *hg(new host_graph);
	auto &g(hg->graph);
	g.reference_positions.assign(ref_pos, ref_pos + n_nodes);
	g.aligned_positions.assign(aln_pos, aln_pos + n_nodes);
	g.alt_edge_targets.assign(targets, targets + n_edges);
	g.alt_edge_count_csum.assign(csum, csum + n_nodes + 1);
	g.alt_edge_label_offsets.assign(label_offsets, label_offsets + n_edges + 1);
	g.alt_edge_label_bytes.assign(label_bytes, label_bytes + label_offsets[n_edges]);
	g.paths_by_edge_and_chrom_copy = vh::bit_matrix(path_rows, path_cols);
	if (paths_by_edge_and_chrom_copy)
		std::copy(paths_by_edge_and_chrom_copy, paths_by_edge_and_chrom_copy + path_rows / 64 * path_cols, g.paths_by_edge_and_chrom_copy.words.begin());
	g.ploidy_csum.assign(1, 0);
	for (uint32_t s(0); s < n_samples; ++s) { g.sample_names.push_back("S" + std::to_string(s)); g.ploidy_csum.push_back(g.ploidy_csum.back() + ploidy); }
	for (auto const &s : g.sample_names) { hg->sample_blob += s; hg->sample_blob.push_back('\0'); }
	return hg;
}

void v2mh_free(void *h) { delete static_cast<host_graph *>(h); }

// How the next v2mh_build_variant_graph() calls answer a REF column mismatch: 0 = it is an error (default), 1 = stop parsing there.
void v2mh_set_stop_at_ref_mismatch(int stop) { g_stop_at_mismatch = 0 != stop; }

int v2mh_write_graph(void *h, char const *path, char *err, size_t errlen)
{
	try { vh::write_graph(static_cast<host_graph *>(h)->graph, path); return 0; }
	catch (std::exception const &e) { if (err && errlen) { std::strncpy(err, e.what(), errlen - 1); err[errlen - 1] = 0; } return 1; }
}

void *v2mh_read_graph(char const *path, char *err, size_t errlen)
{
	auto *hg(new host_graph);
	try {
		vh::read_graph(path, hg->graph);
		for (auto const &s : hg->graph.sample_names) { hg->sample_blob += s; hg->sample_blob.push_back('\0'); }
		return hg;
	} catch (std::exception const &e) {
		if (err && errlen) { std::strncpy(err, e.what(), errlen - 1); err[errlen - 1] = 0; }
		delete hg;
		return nullptr;
	}
}

#define HG(h) (*static_cast<host_graph *>(h))
uint64_t v2mh_node_count(void *h) { return HG(h).graph.node_count(); }
uint64_t v2mh_edge_count(void *h) { return HG(h).graph.edge_count(); }
uint64_t v2mh_sample_count(void *h) { return HG(h).graph.sample_names.size(); }
uint64_t v2mh_ref_length(void *h) { return HG(h).ref.size(); }
char const *v2mh_reference(void *h) { return HG(h).ref.data(); }
uint64_t const *v2mh_reference_positions(void *h) { return HG(h).graph.reference_positions.data(); }
uint64_t const *v2mh_aligned_positions(void *h) { return HG(h).graph.aligned_positions.data(); }
uint64_t const *v2mh_alt_edge_targets(void *h) { return HG(h).graph.alt_edge_targets.data(); }
uint64_t const *v2mh_alt_edge_count_csum(void *h) { return HG(h).graph.alt_edge_count_csum.data(); }
uint64_t const *v2mh_label_offsets(void *h) { return HG(h).graph.alt_edge_label_offsets.data(); }
char const *v2mh_label_bytes(void *h) { return HG(h).graph.alt_edge_label_bytes.data(); }
char const *v2mh_sample_blob(void *h) { return HG(h).sample_blob.data(); }
uint64_t v2mh_sample_blob_size(void *h) { return HG(h).sample_blob.size(); }
uint32_t const *v2mh_ploidy_csum(void *h) { return HG(h).graph.ploidy_csum.data(); }
uint64_t v2mh_ploidy_csum_size(void *h) { return HG(h).graph.ploidy_csum.size(); }
uint64_t v2mh_handled_variants(void *h) { return HG(h).stats.handled_variants; }
uint64_t v2mh_chr_id_mismatches(void *h) { return HG(h).stats.chr_id_mismatches; }
uint64_t const *v2mh_paths_by_edge_and_chrom_copy(void *h, uint64_t *rows, uint64_t *cols)
{
	auto const &m(HG(h).graph.paths_by_edge_and_chrom_copy);
	*rows = m.rows; *cols = m.cols;
	return m.words.data();
}
uint64_t v2mh_overlap_count(void *h) { return HG(h).overlaps.size(); }
void v2mh_overlap_get(void *h, uint64_t i, uint64_t *lineno, uint64_t *ref_pos, char const **var_id, char const **sample, uint32_t *copy, uint32_t *gt)
{
	auto const &o(HG(h).overlaps[i]);
	*lineno = o.lineno; *ref_pos = o.ref_pos; *var_id = o.var_id.c_str(); *sample = o.sample.c_str(); *copy = o.copy; *gt = o.gt;
}

// find_cut_positions + find_matchings on a built graph.  cuts_out must hold node_count entries; assigned_out
// (cuts - 1) * founder_count entries (column-major).  Returns the number of cut positions, 0 if there is no
// solution; *score_out receives the segmentation score.
uint64_t v2mh_find_founders_mt(void *h, uint64_t min_distance, uint32_t founder_count, int keep_ref_edges,
	uint64_t *cuts_out, uint32_t *assigned_out, uint64_t assigned_capacity, uint32_t *score_out, unsigned threads)
{
	auto const &g(HG(h).graph);
	std::vector<vh::u64> cuts;
	vh::u32 const score(vh::find_cut_positions(g, min_distance, cuts, threads));
	if (score_out) *score_out = score;
	if (vh::kCutPositionScoreMax == score) return 0;
	std::vector<vh::u32> assigned;
	if (!vh::find_matchings(g, cuts, founder_count, 0 != keep_ref_edges, assigned, threads)) return 0;
	if (assigned.size() > assigned_capacity) return 0;
	std::copy(cuts.begin(), cuts.end(), cuts_out);
	std::copy(assigned.begin(), assigned.end(), assigned_out);
	return cuts.size();
}

// find_cut_positions with the chunk walks on the GPU context `ctx` (a v2m_ctx * that holds this graph with its path matrix).
// Returns the number of cut positions (0: no solution); cuts_out must hold node_count entries.
uint64_t v2mh_find_cut_positions_gpu(void *h, void *ctx, uint64_t min_distance, unsigned threads, uint64_t *cuts_out, uint32_t *score_out, uint64_t *chunks_walked_and_left, char *err, size_t errlen)
{
	try {
		vh::gpu_context gpu(static_cast<v2m_ctx *>(ctx), vh::gpu_context::borrowed{});
		vh::gpu_cut_trial_walker walker(gpu);
		std::vector<vh::u64> cuts;
		vh::u32 const score(vh::find_cut_positions(HG(h).graph, min_distance, cuts, threads, &walker));
		if (score_out) *score_out = score;
		if (chunks_walked_and_left) { chunks_walked_and_left[0] = walker.chunks_walked; chunks_walked_and_left[1] = walker.chunks_left; }
		if (vh::kCutPositionScoreMax == score) return 0;
		std::copy(cuts.begin(), cuts.end(), cuts_out);
		return cuts.size();
	} catch (std::exception const &e) {
		if (err && errlen) { std::strncpy(err, e.what(), errlen - 1); err[errlen - 1] = 0; }
		if (score_out) *score_out = vh::kCutPositionScoreMax;
		return 0;
	}
}

// find_cut_positions + find_matchings with the chunk walks of both on the GPU context `ctx`.  chunks: {cut search: walked on the
// GPU, left to the host; matching: walked, left}.  with_search = 0: the cut positions are given in cuts_out[0 .. n_cuts_in).
uint64_t v2mh_find_founders_gpu(void *h, void *ctx, uint64_t min_distance, uint32_t founder_count, int keep_ref_edges, unsigned threads,
	int with_search, uint64_t n_cuts_in, uint64_t *cuts_out, uint32_t *assigned_out, uint64_t assigned_capacity, uint32_t *score_out, uint64_t *chunks, char *err, size_t errlen)
{
	try {
		vh::gpu_context gpu(static_cast<v2m_ctx *>(ctx), vh::gpu_context::borrowed{});
		vh::gpu_founder_walker walker(gpu);
		auto const &g(HG(h).graph);
		std::vector<vh::u64> cuts;
		vh::u32 score(0);
		if (with_search) {
			score = vh::find_cut_positions(g, min_distance, cuts, threads, &walker);
			if (chunks) { chunks[0] = walker.chunks_walked; chunks[1] = walker.chunks_left; }
			if (score_out) *score_out = score;
			if (vh::kCutPositionScoreMax == score) return 0;
		} else {
			cuts.assign(cuts_out, cuts_out + n_cuts_in);
		}
		std::vector<vh::u32> assigned;
		walker.chunks_walked = walker.chunks_left = 0;
		if (!vh::find_matchings(g, cuts, founder_count, 0 != keep_ref_edges, assigned, threads, &walker)) return 0;
		if (chunks) { chunks[2] = walker.chunks_walked; chunks[3] = walker.chunks_left; }
		if (assigned.size() > assigned_capacity) return 0;
		std::copy(cuts.begin(), cuts.end(), cuts_out);
		std::copy(assigned.begin(), assigned.end(), assigned_out);
		return cuts.size();
	} catch (std::exception const &e) {
		if (err && errlen) { std::strncpy(err, e.what(), errlen - 1); err[errlen - 1] = 0; }
		return 0;
	}
}

// The walked searches with the chunk walks done by the host's own edge-by-edge walker (founder.hh: make_host_founder_walker):
// the plumbing of find_cut_positions_walked / find_matchings_walked without a GPU.  chunks as in v2mh_find_founders_gpu.
uint64_t v2mh_find_founders_walked_on_host(void *h, uint64_t min_distance, uint32_t founder_count, int keep_ref_edges, unsigned threads, uint64_t max_copies,
	uint64_t *cuts_out, uint32_t *assigned_out, uint64_t assigned_capacity, uint32_t *score_out, uint64_t *chunks, char *err, size_t errlen)
{
	try {
		auto const &g(HG(h).graph);
		auto walker(vh::make_host_founder_walker(g, max_copies));
		std::vector<vh::u64> cuts;
		vh::u32 const score(vh::find_cut_positions(g, min_distance, cuts, threads, walker.get()));
		if (chunks) { chunks[0] = walker->chunks_walked; chunks[1] = walker->chunks_left; }
		if (score_out) *score_out = score;
		if (vh::kCutPositionScoreMax == score) return 0;
		std::vector<vh::u32> assigned;
		walker->chunks_walked = walker->chunks_left = 0;
		if (!vh::find_matchings(g, cuts, founder_count, 0 != keep_ref_edges, assigned, threads, walker.get())) return 0;
		if (chunks) { chunks[2] = walker->chunks_walked; chunks[3] = walker->chunks_left; }
		if (assigned.size() > assigned_capacity) return 0;
		std::copy(cuts.begin(), cuts.end(), cuts_out);
		std::copy(assigned.begin(), assigned.end(), assigned_out);
		return cuts.size();
	} catch (std::exception const &e) {      // nothing may cross the C boundary (an exception would end the caller's process)
		if (err && errlen) { std::strncpy(err, e.what(), errlen - 1); err[errlen - 1] = 0; }
		return 0;
	}
}

uint64_t v2mh_find_founders(void *h, uint64_t min_distance, uint32_t founder_count, int keep_ref_edges,
	uint64_t *cuts_out, uint32_t *assigned_out, uint64_t assigned_capacity, uint32_t *score_out)
{
	return v2mh_find_founders_mt(h, min_distance, founder_count, keep_ref_edges, cuts_out, assigned_out, assigned_capacity, score_out, 1);
}

// The transpose's result (rows = edges, cols = copies, column-major words) for a graph whose builder ran without a GPU;
// the multi-threaded founder search reads copy prefixes from it.
void v2mh_set_paths_by_chrom_copy_and_edge(void *h, uint64_t const *words, uint64_t rows, uint64_t cols)
{
	auto &m(HG(h).graph.paths_by_chrom_copy_and_edge);
	m = vh::bit_matrix(rows, cols);
	std::copy(words, words + rows / 64 * cols, m.words.begin());
}

// Cut position files (founder.hh).  Return 0 on success, 1 with a message in err otherwise.
int v2mh_write_cut_positions(char const *path, uint64_t const *cuts, uint64_t n_cuts, uint64_t min_distance, uint32_t score, char *err, size_t errlen)
{
	try {
		vh::write_cut_positions({std::vector<vh::u64>(cuts, cuts + n_cuts), min_distance, score}, path);
		return 0;
	} catch (std::exception const &e) {
		if (err && errlen) { std::strncpy(err, e.what(), errlen - 1); err[errlen - 1] = 0; }
		return 1;
	}
}

// cuts_out may be NULL to query the count (returned through *n_cuts); otherwise it must hold *n_cuts entries.
int v2mh_read_cut_positions(char const *path, uint64_t *cuts_out, uint64_t *n_cuts, uint64_t *min_distance, uint32_t *score, char *err, size_t errlen)
{
	try {
		auto const f(vh::read_cut_positions(path));
		if (cuts_out && *n_cuts >= f.cut_positions.size()) std::copy(f.cut_positions.begin(), f.cut_positions.end(), cuts_out);
		*n_cuts = f.cut_positions.size();
		if (min_distance) *min_distance = f.min_distance;
		if (score) *score = f.score;
		return 0;
	} catch (std::exception const &e) {
		if (err && errlen) { std::strncpy(err, e.what(), errlen - 1); err[errlen - 1] = 0; }
		return 1;
	}
}


// shard_copies() of gpu_path.hh, for the test that it and vcf2multialign_amd/sharding.py agree on every boundary.
void v2mh_shard_copies(uint64_t n_copies, uint32_t world, uint32_t rank, uint64_t *first, uint64_t *end)
{
	auto const s(vh::shard_copies(n_copies, world, rank));
	*first = s.first;
	*end = s.end;
}

} // extern "C"
