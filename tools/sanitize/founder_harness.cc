// Sanitizer harness for the host's founder search (csrc/host/founder.cc): a random graph with one ALT edge per
// second node, both path matrices filled; the sequential search against the chunked one on 2, 4 and 8 threads and against the
// walked one (chunks handed to a founder_walker: here the host's own).
// Built and run by tools/sanitize_host.sh with -fsanitize=address,undefined and -fsanitize=thread.
#include "founder.hh"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

using namespace v2m::host;

int main(int argc, char **argv)
{
	u64 const E(argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 3000);
	u32 const H(argc > 2 ? u32(std::strtoul(argv[2], nullptr, 10)) : 600);
	u64 const Hp((H + 63) / 64 * 64), Ep((E + 63) / 64 * 64), N(2 * E + 2);
	variant_graph g;
	g.reference_positions.resize(N);
	g.aligned_positions.resize(N);
	g.alt_edge_count_csum.assign(N + 1, 0);
	for (u64 i(0); i < N; ++i) {
		u64 const p(0 == i ? 0 : (i % 2 ? 10 * ((i + 1) / 2) : 10 * (i / 2) + 1));
		g.reference_positions[i] = g.aligned_positions[i] = p;
		g.alt_edge_count_csum[i + 1] = g.alt_edge_count_csum[i] + ((1 == i % 2 && i < 2 * E) ? 1 : 0);
	}
	g.alt_edge_targets.resize(E);
	g.alt_edge_label_offsets.resize(E + 1);
	for (u64 e(0); e < E; ++e) { g.alt_edge_targets[e] = 2 * e + 2; g.alt_edge_label_offsets[e + 1] = e + 1; }
	g.alt_edge_label_bytes.assign(E, 'A');
	g.sample_names.resize(H / 2);
	g.ploidy_csum.resize(H / 2 + 1);
	for (u32 i(0); i <= H / 2; ++i) g.ploidy_csum[i] = 2 * i;
	g.paths_by_edge_and_chrom_copy = bit_matrix(Hp, Ep);
	g.paths_by_chrom_copy_and_edge = bit_matrix(Ep, Hp);
	std::mt19937_64 rng(1);
	std::uniform_real_distribution<double> U(0, 1);
	for (u64 e(0); e < E; ++e) {
		double const f(0.5 * std::pow(10., -3 * U(rng)));
		for (u32 c(0); c < 2 * (H / 2); ++c)
			if (U(rng) < f) { g.paths_by_edge_and_chrom_copy.set(c, e); g.paths_by_chrom_copy_and_edge.set(e, c); }
	}

	std::vector<u64> cuts;
	u32 const score(find_cut_positions(g, 50, cuts, 1));
	std::vector<u32> assigned;
	find_matchings(g, cuts, 25, false, assigned, 1);
	std::printf("E=%llu copies=%u: %zu cut positions, score %u\n", (unsigned long long) E, 2 * (H / 2), cuts.size(), score);
	for (unsigned threads : {2u, 4u, 8u}) {
		std::vector<u64> cuts_mt;
		std::vector<u32> assigned_mt;
		u32 const score_mt(find_cut_positions(g, 50, cuts_mt, threads));
		find_matchings(g, cuts_mt, 25, false, assigned_mt, threads);
		bool const same(score_mt == score && cuts_mt == cuts && assigned_mt == assigned);
		std::printf("  %u threads: %s\n", threads, same ? "same" : "DIFFERENT");
		if (!same) return 1;
	}
	// the walked searches (what the GPU path runs around its kernels) with the host's own walker, several threads building the states
	for (unsigned threads : {1u, 4u}) {
		auto walker(make_host_founder_walker(g));
		std::vector<u64> cuts_w;
		std::vector<u32> assigned_w;
		u32 const score_w(find_cut_positions(g, 50, cuts_w, threads, walker.get()));
		find_matchings(g, cuts_w, 25, false, assigned_w, threads, walker.get());
		bool const same(score_w == score && cuts_w == cuts && assigned_w == assigned);
		std::printf("  walked, %u threads: %s (%llu chunks)\n", threads, same ? "same" : "DIFFERENT", (unsigned long long) walker->chunks_walked);
		if (!same) return 1;
	}
	return 0;
}
