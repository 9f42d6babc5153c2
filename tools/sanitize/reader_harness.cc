// Sanitizer harness for the host's readers, graph builder and graph checkpoint (csrc/host/readers.cc,
// graph_builder.cc, graph_file.cc): reader_harness FASTA VCF SCRATCH_FILE builds the graph on 1, 3 and 8 threads
// (with a sample filter and both matrix paddings), compares the results and round-trips the checkpoint, once more with a
// matrix large enough for the reader's threads.  A fourth argument names the input's kind: "chunks" = an input of several of the
// reader's 8-MB chunks (the worker window, recycled chunks and the merge across chunk boundaries run; the 72-MiB checkpoint is
// skipped), "stop" = the same with a REF column mismatch in a later chunk and a delegate that ends the build there
// (variant_graph.cc:307-314) -- the graphs of 1, 3 and 8 threads must agree and must end at the mismatch.
#include "graph_file.hh"
#include "readers.hh"

#include <cstdio>
#include <cstring>

using namespace v2m::host;

struct delegate final : build_graph_delegate {
	u64 overlaps{};
	bool should_include(std::string_view sample, u32 copy) const override { return !(sample == "S3" && 1 == copy); }
	void report_overlapping_alternative(u64, u64, std::string_view, std::string_view, u32, u32) override { ++overlaps; }
	bool stop_at_mismatch{};
	u64 mismatches{};
	bool ref_column_mismatch(u64, u64, std::string_view, std::string_view) override { ++mismatches; return !stop_at_mismatch; }
};

int main(int argc, char **argv)
{
	if (argc < 4) return 2;
	sequence_type ref;
	if (!read_single_fasta_sequence(argv[1], ref)) return 2;
	bool const many_chunks(argc > 4 && 0 == std::strcmp(argv[4], "chunks")), stop(argc > 4 && 0 == std::strcmp(argv[4], "stop"));
	u64 signature[3] = {};
	int k(0);
	for (unsigned threads : {1u, 3u, 8u}) {
		variant_graph g;
		build_graph_statistics stats;
		delegate d;
		d.stop_at_mismatch = stop;
		build_variant_graph(ref, argv[2], "1", g, stats, d, threads, 3 == threads ? 1024 : 64);
		std::printf("threads %u: %llu nodes, %llu edges, %llu overlaps, %llu records\n", threads, (unsigned long long) g.node_count(),
			(unsigned long long) g.edge_count(), (unsigned long long) d.overlaps, (unsigned long long) stats.handled_variants);
		signature[k++] = g.node_count() ^ (g.edge_count() << 20) ^ (d.overlaps << 40) ^ (stats.handled_variants << 50);
		if (stop && (1 != d.mismatches || g.reference_positions.back() != ref.size())) return 6;      // ended at the first mismatch, sink node added (:437-451)
		if (many_chunks && (0 != d.mismatches || 0 == d.overlaps)) return 7;
		write_graph(g, argv[3]);
		variant_graph back;
		read_graph(argv[3], back);
		if (back.alt_edge_targets != g.alt_edge_targets || back.paths_by_edge_and_chrom_copy.words != g.paths_by_edge_and_chrom_copy.words) return 3;
	}
	if (!many_chunks && !stop) {
		// a path matrix of 72 MiB: read_graph() fetches and checksums it in 32-MiB parts on three threads
		variant_graph g;
		g.reference_positions = {0, 1};
		g.aligned_positions = {0, 1};
		g.alt_edge_targets = {1};
		g.alt_edge_count_csum = {0, 1, 1};
		g.alt_edge_label_offsets = {0, 1};
		g.alt_edge_label_bytes = "A";
		g.sample_names.assign(4608, "S");
		g.ploidy_csum.resize(4609);
		for (u32 i(0); i < 4609; ++i) g.ploidy_csum[i] = 2 * i;
		g.paths_by_edge_and_chrom_copy = bit_matrix(9216, 65536);
		u64 x(88172645463325252ULL);
		for (auto &w : g.paths_by_edge_and_chrom_copy.words) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; w = x; }
		write_graph(g, argv[3]);
		variant_graph back;
		read_graph(argv[3], back);
		if (back.paths_by_edge_and_chrom_copy.words != g.paths_by_edge_and_chrom_copy.words || back.sample_names.size() != 4608) return 5;
		std::printf("72-MiB matrix: checkpoint round trip on several reader threads\n");
	}
	return (signature[0] == signature[1] && signature[1] == signature[2]) ? 0 : 4;
}
