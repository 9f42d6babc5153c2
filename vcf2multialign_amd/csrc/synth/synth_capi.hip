// synth_capi.hip -- C API of the synthetic-input generator (bench.py / scale tests; not part of
// the drop-in boundary).  Host generator + the HIP kernel that fills the genotype bit matrix in HBM.

#include <hip/hip_runtime.h>

#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "synth.hh"

namespace {

using v2m::synth::u32;
using v2m::synth::u64;

// paths_by_edge_and_chrom_copy: rows = chromosome copies (n_rows, multiple of 64), cols = edges.
// One thread per 64-bit word = 64 copies of one edge.
__global__ __launch_bounds__(256) void fill_paths_kernel(
	u64 *__restrict__ words, u64 words_per_col, u64 n_cols, u64 copy_base, u64 n_copies, u64 n_edges,
	u32 const *__restrict__ thresholds, u64 seed)
{
	u64 const idx = (u64) blockIdx.x * blockDim.x + threadIdx.x;
	if (idx >= words_per_col * n_cols) return;
	u64 const edge = idx / words_per_col, cw = idx % words_per_col;
	u64 w = 0;
	if (edge < n_edges) {
		u32 const thr = thresholds[edge];
		u64 const base = seed + edge * 0x9E3779B97F4A7C15ULL;
#pragma unroll 8
		for (int b = 0; b < 64; ++b) {
			u64 const copy = copy_base + cw * 64 + b;   // global chromosome copy; this matrix holds copies [copy_base, copy_base + n_rows)
			if (copy < n_copies) {
				u64 z = base + copy * 0xC2B2AE3D27D4EB4FULL;
				z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
				z ^= z >> 27; z *= 0x94D049BB133111EBULL;
				z ^= z >> 31;
				if ((u32) (z >> 32) < thr) w |= 1ULL << b;
			}
		}
	}
	words[idx] = w;
}

} // namespace


extern "C" {

struct v2ms_config {
	uint64_t seed, ref_length, n_variants;
	double frac_mnp, frac_insertion, frac_deletion, frac_multiallelic;
	uint32_t max_indel;
};

void *v2ms_generate(v2ms_config const *c)
{
	auto *d(new v2m::synth::dataset);
	v2m::synth::config cfg;
	cfg.seed = c->seed; cfg.ref_length = c->ref_length; cfg.n_variants = c->n_variants;
	cfg.frac_mnp = c->frac_mnp; cfg.frac_insertion = c->frac_insertion; cfg.frac_deletion = c->frac_deletion;
	cfg.frac_multiallelic = c->frac_multiallelic; cfg.max_indel = c->max_indel ? c->max_indel : 32;
	v2m::synth::generate(cfg, *d);
	return d;
}

void v2ms_free(void *h) { delete static_cast<v2m::synth::dataset *>(h); }

#define D(h) (*static_cast<v2m::synth::dataset *>(h))
uint64_t v2ms_node_count(void *h) { return D(h).graph.node_count(); }
uint64_t v2ms_edge_count(void *h) { return D(h).graph.edge_count(); }
uint64_t v2ms_ref_length(void *h) { return D(h).reference.size(); }
char const *v2ms_reference(void *h) { return D(h).reference.data(); }
uint64_t const *v2ms_reference_positions(void *h) { return D(h).graph.reference_positions.data(); }
uint64_t const *v2ms_aligned_positions(void *h) { return D(h).graph.aligned_positions.data(); }
uint64_t const *v2ms_alt_edge_targets(void *h) { return D(h).graph.alt_edge_targets.data(); }
uint64_t const *v2ms_alt_edge_count_csum(void *h) { return D(h).graph.alt_edge_count_csum.data(); }
uint64_t const *v2ms_label_offsets(void *h) { return D(h).graph.alt_edge_label_offsets.data(); }
char const *v2ms_label_bytes(void *h) { return D(h).graph.alt_edge_label_bytes.data(); }
uint32_t const *v2ms_edge_thresholds(void *h) { return D(h).edge_thresholds.data(); }

int v2ms_write_fasta_and_vcf(void *h, uint64_t seed, uint32_t samples, uint32_t ploidy, char const *chromosome, char const *fasta_path, char const *vcf_path)
{
	return v2m::synth::write_fasta_and_vcf(D(h), seed, samples, ploidy, chromosome, fasta_path, vcf_path) ? 0 : 1;
}

// Fills a (n_rows x n_cols)-bit column-major matrix in HBM (rows = the chromosome copies
// [copy_base, copy_base + n_rows) of n_copies in total, cols = edges) on `stream` (a hipStream_t).
// Returns 0 on success, a hipError_t otherwise.
int v2ms_fill_paths_device(void *stream, void *d_words, uint64_t n_rows, uint64_t n_cols, uint64_t copy_base, uint64_t n_copies, uint64_t n_edges, void const *d_thresholds, uint64_t seed)
{
	if (n_rows % 64) return -1;
	u64 const total(n_rows / 64 * n_cols);
	if (0 == total) return 0;
	hipLaunchKernelGGL(fill_paths_kernel, dim3(unsigned((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
		static_cast<u64 *>(d_words), n_rows / 64, n_cols, copy_base, n_copies, n_edges, static_cast<u32 const *>(d_thresholds), seed);
	return int(hipGetLastError());
}

// CPU re-derivation of one copy's column of paths_by_chrom_copy_and_edge (n_words = ceil(E/64) words).
void v2ms_copy_column(void *h, uint64_t seed, uint64_t copy, uint64_t *words_out, uint64_t n_words)
{
	auto const &d(D(h));
	std::memset(words_out, 0, n_words * 8);
	for (u64 e(0); e < d.graph.edge_count() && e / 64 < n_words; ++e)
		if (v2m::synth::path_bit(seed, e, copy, d.edge_thresholds[e]))
			words_out[e >> 6] |= u64(1) << (e & 63);
}

// A v2m_sink_fn implemented in C for end-to-end measurements of v2m_splice_rows (tools/sink_bench.py): writes
// '>' id '\n' body '\n' to a file descriptor like the A2M writer does (fd < 0: only counts).
struct v2ms_sink_state { int fd; uint64_t rows; uint64_t bytes; };

int v2ms_fd_sink(void *user, uint64_t row, char const *bytes, uint64_t length)
{
	auto *st(static_cast<v2ms_sink_state *>(user));
	++st->rows;
	st->bytes += length;
	if (st->fd < 0) return 0;
	char header[64];
	int const n(std::snprintf(header, sizeof(header), ">row%llu\n", (unsigned long long) row));
	if (n != ::write(st->fd, header, size_t(n))) return 1;
	uint64_t done(0);
	while (done < length) {
		ssize_t const w(::write(st->fd, bytes + done, length - done));
		if (w <= 0) return 1;
		done += uint64_t(w);
	}
	return 1 == ::write(st->fd, "\n", 1) ? 0 : 1;
}



} // extern "C"
