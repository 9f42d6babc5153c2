// main.cc -- thin command-line driver with the reference's flag names (vcf2multialign/cmdline.ggo:4-55,
// vcf2multialign/main.cc:370-552) for the part this repository implements: --haplotypes and --founder-sequences with
// FASTA + VCF (or checkpointed graph) input, A2M / separate / unaligned / piped output, sample filters, overlap
// reports, cut-position files.  The per-row splicing and the path-matrix
// transpose run on the GPU (there is no CPU path); everything else is host code.

#include <getopt.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <future>
#include <thread>
#include <iostream>
#include <memory>
#include <set>
#include <sstream>
#include <string>
#include <tuple>
#include <vector>

#include <chrono>
#include <csignal>

#include "founder.hh"
#include "gpu_path.hh"
#include "graph_file.hh"
#include "output.hh"
#include "readers.hh"

namespace vh = v2m::host;

namespace {

struct options {
	bool haplotypes{};
	long founder_sequences{0};
	long minimum_distance{0};
	bool founder_mode{}, keep_ref_edges{};
	char const *input_reference{}, *reference_sequence{}, *input_variants{}, *chromosome{};
	char const *output_sequences_a2m{}, *dst_chromosome{}, *output_overlaps{};
	char const *include_samples{}, *exclude_samples{};
	char const *input_graph{}, *output_graph{};
	char const *pipe{}, *input_cut_positions{}, *output_cut_positions{};
	bool output_sequences_separate{}, separate_plain{}, omit_reference{}, unaligned{}, verbose{}, graph_statistics{};
	bool ref_mismatch_error{};
	std::vector<int> devices{0};
};

void usage()
{
	std::cerr <<
		"Usage: vcf2multialign --haplotypes --input-reference=filename.fa --input-variants=filename.vcf --chromosome=id [<options>]\n"
		"  -H, --haplotypes                   Produce predicted haplotype sequences\n"
		"  -r, --input-reference=filename     Reference FASTA file path\n"
		"  -e, --reference-sequence=id        Reference sequence identifier in the input FASTA\n"
		"  -a, --input-variants=filename      Variant call file path\n"
		"  -c, --chromosome=id                Chromosome identifier\n"
		"  -s, --output-sequences-a2m=file    Output reference-guided multiple alignment as A2M\n"
		"      --output-sequences-separate    Output one sequence per file\n"
		"      --separate-output-format=fmt   A2M (default) or plain\n"
		"  -m, --dst-chromosome=id            Chromosome identifier in output\n"
		"      --omit-reference               Omit the reference sequence from the output\n"
		"      --unaligned                    Output unaligned sequences instead of an MSA\n"
		"      --output-overlaps=file         Output overlapping variants as TSV instead of stdout\n"
		"      --output-graph-statistics      Output graph statistics to stdout\n"
		"      --ref-mismatch-handling=how    warning (default) or error\n"
		"      --include-samples=file         TSV (chrom, sample, copy_idx) of the only copies to include\n"
		"  -x, --exclude-samples=file         TSV (chrom, sample, copy_idx) of copies to exclude\n"
		"      --device=n[,m...]              HIP device(s) to run on (default 0); with several, the rows of an aligned\n"
		"                                     A2M file are sharded over them (graph replicated, no collective)\n"
		"      --verbose\n"
		"  -F, --founder-sequences=count      Produce founder sequences instead of haplotypes\n"
		"  -d, --minimum-distance=distance    Minimum node distance (MSA co-ordinates) between cut positions\n"
		"      --keep-ref-edges               Take the reference edges into account when matching\n"
		"  -g, --input-graph=filename         Variant graph input (this build's flat V2MGRAF1 format)\n"
		"  -f, --output-graph=filename        Output the variant graph\n"
		"  -p, --input-cut-positions=file     Cut position input\n"
		"  -t, --output-cut-positions=file    Output the cut positions\n"
		"      --pipe=command                 Instead of writing sequences to files, pipe them to `command <name>`\n"
		"  (--output-graphviz and --output-memory-breakdown are outside this build's scope: SURVEY.md section 2)\n";
}

typedef std::set<std::tuple<std::string, std::string, unsigned>> sample_set;


// main.cc:42-120 (sample filter lists): TSV rows (chrom, sample, copy_idx)
sample_set read_sample_list(char const *path, char const *chromosome)
{
	sample_set out;
	std::ifstream is(path);
	if (!is) { std::cerr << "ERROR: Unable to open " << path << '\n'; std::exit(EXIT_FAILURE); }
	std::string line;
	while (std::getline(is, line)) {
		if (line.empty()) continue;
		std::istringstream ls(line);
		std::string chrom, sample; unsigned copy{};
		if (!(std::getline(ls, chrom, '\t') && std::getline(ls, sample, '\t') && (ls >> copy))) continue;
		if (chromosome && chrom != chromosome) continue;
		out.emplace(chrom, sample, copy);
	}
	return out;
}

struct build_delegate final : vh::build_graph_delegate {
	sample_set const *included{}, *excluded{};
	std::string chromosome;
	std::ostream *overlaps{&std::cout};
	bool mismatch_is_error{};
	bool is_tsv{};

	bool should_include(std::string_view sample_name, vh::u32 chrom_copy_idx) const override
	{
		auto const key(std::make_tuple(chromosome, std::string(sample_name), unsigned(chrom_copy_idx)));
		if (included) return included->count(key) > 0;
		if (excluded) return excluded->count(key) == 0;
		return true;
	}

	static std::string joined_ids(std::string_view var_id, char const *sep)
	{
		std::string out;   // the VCF ID column holds ';'-separated identifiers
		for (char const c : var_id) { if (';' == c) out += sep; else out.push_back(c); }
		return out;
	}

	void report_overlapping_alternative(vh::u64 lineno, vh::u64 ref_pos, std::string_view var_id, std::string_view sample_name, vh::u32 chrom_copy_idx, vh::u32 gt) override
	{
		if (is_tsv)   // main.cc:159-165
			*overlaps << lineno << '\t' << ref_pos << '\t' << joined_ids(var_id, ",") << '\t' << sample_name << '\t' << chrom_copy_idx << '\t' << gt << '\n';
		else          // main.cc:168-170
			*overlaps << "Overlapping alternative alleles. Line number: " << lineno << " current variant position: " << ref_pos
				<< " variant identifiers: " << joined_ids(var_id, ", ") << " sample: " << sample_name << " chromosome copy: " << chrom_copy_idx << " genotype: " << gt << '\n';
	}

	bool ref_column_mismatch(vh::u64 var_idx, vh::u64 ref_pos, std::string_view ref_in_vcf, std::string_view expected) override
	{
		std::cerr << (mismatch_is_error ? "ERROR: " : "WARNING: ") << "REF column contents do not match the reference sequence in variant " << var_idx
			<< ", position " << (1 + ref_pos) << ". Expected: \"" << expected << "\" Actual: \"" << ref_in_vcf << "\"\n";   // main.cc:179-189
		if (mismatch_is_error) std::exit(EXIT_FAILURE);
		return true;
	}
};

struct progress_delegate final : vh::output_delegate {
	bool verbose{};
	void will_handle_sample(std::string const &, vh::u32, vh::u32) override {}
	void will_handle_founder_sequence(vh::u32) override {}
	void handled_sequences(vh::u32 count) override
	{
		if (verbose && 0 == count % 10) std::cerr << "Handled " << count << " sequences...\n";   // main.cc:313-330
	}
};

} // namespace


int main(int argc, char **argv)
{
	::signal(SIGPIPE, SIG_IGN);   // main.cc:372: a --pipe child that goes away shows up as a write error
	// A command-line run is one shot: the library's per-context calibrations (which transpose kernel for a matrix shape, plain
	// or nontemporal row stores: a handful of extra launches each) have nothing to amortise over, so the driver presets what
	// they choose on nearly every box.  A value already in the environment wins.
	::setenv("V2M_TRANSPOSE_PANEL", "stream16", 0);
	::setenv("V2M_NT_STORES", "1", 0);
	::setenv("V2M_UNALIGNED_STORE", "plain", 0);
	options opt;
	enum { o_keep_ref = 900, o_separate = 1000, o_sep_format, o_omit_ref, o_unaligned, o_overlaps, o_stats, o_mismatch, o_include, o_device, o_verbose, o_pipe, o_unsupported };
	static option const longopts[] = {
		{"haplotypes", no_argument, nullptr, 'H'}, {"founder-sequences", required_argument, nullptr, 'F'},
		{"input-reference", required_argument, nullptr, 'r'}, {"reference-sequence", required_argument, nullptr, 'e'},
		{"input-variants", required_argument, nullptr, 'a'}, {"chromosome", required_argument, nullptr, 'c'},
		{"output-sequences-a2m", required_argument, nullptr, 's'}, {"output-sequences-separate", no_argument, nullptr, o_separate},
		{"separate-output-format", required_argument, nullptr, o_sep_format}, {"dst-chromosome", required_argument, nullptr, 'm'},
		{"omit-reference", no_argument, nullptr, o_omit_ref}, {"unaligned", no_argument, nullptr, o_unaligned},
		{"output-overlaps", required_argument, nullptr, o_overlaps}, {"output-graph-statistics", no_argument, nullptr, o_stats},
		{"ref-mismatch-handling", required_argument, nullptr, o_mismatch}, {"include-samples", required_argument, nullptr, o_include},
		{"exclude-samples", required_argument, nullptr, 'x'}, {"device", required_argument, nullptr, o_device}, {"verbose", no_argument, nullptr, o_verbose},
		{"input-graph", required_argument, nullptr, 'g'}, {"output-graph", required_argument, nullptr, 'f'},
		{"output-graphviz", required_argument, nullptr, o_unsupported}, {"output-memory-breakdown", required_argument, nullptr, o_unsupported}, {"pipe", required_argument, nullptr, o_pipe},
		{"minimum-distance", required_argument, nullptr, 'd'}, {"input-cut-positions", required_argument, nullptr, 'p'},
		{"output-cut-positions", required_argument, nullptr, 't'}, {"keep-ref-edges", no_argument, nullptr, o_keep_ref},
		{"help", no_argument, nullptr, 'h'}, {nullptr, 0, nullptr, 0}};
	int c;
	while (-1 != (c = getopt_long(argc, argv, "HF:d:p:t:r:e:a:c:s:m:x:g:f:v:h", longopts, nullptr))) {
		switch (c) {
			case 'H': opt.haplotypes = true; break;
			case 'F': opt.founder_mode = true; opt.founder_sequences = std::atol(optarg); break;
			case 'd': opt.minimum_distance = std::atol(optarg); break;
			case o_keep_ref: opt.keep_ref_edges = true; break;
			case 'r': opt.input_reference = optarg; break;
			case 'e': opt.reference_sequence = optarg; break;
			case 'a': opt.input_variants = optarg; break;
			case 'c': opt.chromosome = optarg; break;
			case 's': opt.output_sequences_a2m = optarg; break;
			case 'm': opt.dst_chromosome = optarg; break;
			case 'x': opt.exclude_samples = optarg; break;
			case 'g': opt.input_graph = optarg; break;
			case 'f': opt.output_graph = optarg; break;
			case o_separate: opt.output_sequences_separate = true; break;
			case o_sep_format:
				if (0 == std::strcmp(optarg, "plain")) opt.separate_plain = true;
				else if (0 != std::strcmp(optarg, "A2M")) { std::cerr << "ERROR: --separate-output-format must be A2M or plain.\n"; return EXIT_FAILURE; }
				break;
			case o_omit_ref: opt.omit_reference = true; break;
			case o_unaligned: opt.unaligned = true; break;
			case o_overlaps: opt.output_overlaps = optarg; break;
			case o_stats: opt.graph_statistics = true; break;
			case o_mismatch:
				if (0 == std::strcmp(optarg, "error")) opt.ref_mismatch_error = true;
				else if (0 != std::strcmp(optarg, "warning")) { std::cerr << "ERROR: --ref-mismatch-handling must be warning or error.\n"; return EXIT_FAILURE; }
				break;
			case o_include: opt.include_samples = optarg; break;
			case o_device: {
				opt.devices.clear();
				std::istringstream ds(optarg);
				for (std::string tok; std::getline(ds, tok, ',');) if (!tok.empty()) opt.devices.push_back(std::atoi(tok.c_str()));
				if (opt.devices.empty()) { std::cerr << "ERROR: --device needs at least one device.\n"; return EXIT_FAILURE; }
				break;
			}
			case o_verbose: opt.verbose = true; break;
			case o_pipe: opt.pipe = optarg; break;
			case 'p': opt.input_cut_positions = optarg; break;
			case 't': opt.output_cut_positions = optarg; break;
			case 'h': usage(); return EXIT_SUCCESS;
			case 'v':   // --output-graphviz's short form (cmdline.ggo:38): the same answer
			case o_unsupported: std::cerr << "ERROR: option " << ('v' == c ? "-v / --output-graphviz" : argv[optind - 1]) << " is not supported by this build.\n"; return EXIT_FAILURE;
			default: usage(); return EXIT_FAILURE;
		}
	}

	// main.cc:577-611
	if (opt.haplotypes == opt.founder_mode) { std::cerr << "ERROR: exactly one of --haplotypes and --founder-sequences is required.\n"; return EXIT_FAILURE; }
	if (opt.founder_mode && opt.founder_sequences <= 0) { std::cerr << "ERROR: --founder-sequences must be positive.\n"; return EXIT_FAILURE; }   // main.cc:595-599
	if (opt.minimum_distance < 0) { std::cerr << "ERROR: --minimum-distance must be non-negative.\n"; return EXIT_FAILURE; }                    // main.cc:607-611
	if (!opt.input_reference) { std::cerr << "ERROR: --input-reference is required.\n"; return EXIT_FAILURE; }
	if (opt.input_variants && opt.input_graph) { std::cerr << "ERROR: Only one of --input-variants and --input-graph can be specified.\n"; return EXIT_FAILURE; }   // main.cc:577-581
	if (!opt.input_variants && !opt.input_graph) { std::cerr << "ERROR: One of --input-variants and --input-graph must be specified.\n"; return EXIT_FAILURE; }  // main.cc:583-587
	if (opt.input_variants && !opt.chromosome) { std::cerr << "ERROR: --chromosome must be specified with --input-variants.\n"; return EXIT_FAILURE; }          // main.cc:589-593
	if (opt.output_graph && !opt.input_variants) { std::cerr << "ERROR: --output-graph requires --input-variants.\n"; return EXIT_FAILURE; }                     // cmdline.ggo:40 (dependon)
	if (opt.include_samples && opt.exclude_samples) { std::cerr << "ERROR: --include-samples and --exclude-samples are mutually exclusive.\n"; return EXIT_FAILURE; }

	try {
		// The GPU contexts come up on a second thread (HIP runtime start-up and stream creation: 0.2 s) while this one reads the
		// reference and the graph; without a usable MI355X the run still fails, loudly, at the first use below.
		auto contexts_coming(std::async(std::launch::async, [&opt] {
			std::vector<std::unique_ptr<vh::gpu_context>> contexts;
			for (int const device : opt.devices) contexts.emplace_back(new vh::gpu_context(device));
			return contexts;
		}));
		std::vector<std::unique_ptr<vh::gpu_context>> contexts;
		auto const first_gpu([&]() -> vh::gpu_context & {
			if (contexts.empty()) contexts = contexts_coming.get();
			return *contexts.front();
		});

		vh::sequence_type ref_seq;
		std::cerr << "Reading the reference sequence..." << std::flush;
		if (!vh::read_single_fasta_sequence(opt.input_reference, ref_seq, opt.reference_sequence)) {
			std::cerr << " ERROR: Unable to read the reference sequence.\n";
			return EXIT_FAILURE;
		}
		std::cerr << " Done. Reference length is " << ref_seq.size() << ".\n";
		// A missing or unusable GPU, or a bad --device, should end the run here, before the variants are parsed, not minutes later.
		// HIP's start-up takes 0.2 s whether it ends in a context or in an error, so after a reference of any size (100 Mb: 0.07 s
		// + 30 ms of grace here; a genome: seconds) the outcome is usually in; when it is not, waiting for it would only take the
		// overlap with the graph's loading away (0.1 s at config 3), and the first use reports the failure all the same.
		if (std::future_status::ready == contexts_coming.wait_for(std::chrono::milliseconds(30))) (void) first_gpu();

		vh::variant_graph graph;
		if (opt.input_graph) {                                  // main.cc:392-401
			std::cerr << "Loading the variant graph from " << opt.input_graph << "..." << std::flush;
			vh::read_graph(opt.input_graph, graph);
			std::cerr << " Done.\n";
		} else {
			sample_set included, excluded;
			build_delegate delegate;
			delegate.chromosome = opt.chromosome;
			delegate.mismatch_is_error = opt.ref_mismatch_error;
			if (opt.include_samples) { included = read_sample_list(opt.include_samples, opt.chromosome); delegate.included = &included; }
			if (opt.exclude_samples) { excluded = read_sample_list(opt.exclude_samples, opt.chromosome); delegate.excluded = &excluded; }
			std::ofstream overlaps_os;
			if (opt.output_overlaps) {
				overlaps_os.open(opt.output_overlaps);
				delegate.overlaps = &overlaps_os;
				delegate.is_tsv = true;
			}
			std::cerr << "Building the variant graph...\n";
			vh::build_graph_statistics stats;
			vh::build_variant_graph(ref_seq, opt.input_variants, opt.chromosome, graph, stats, delegate, 0, 64);   // the reference's padding (variant_graph.cc:277,449)
			// variant_graph.cc:453 (the transpose) happens on the GPU(s) below; the transposed matrix only comes back to the host
			// when something on the host reads it: the founder search and the graph checkpoint.
			if (opt.founder_mode || opt.output_graph) vh::transpose_paths(first_gpu(), graph);
			std::cerr << "Done. Handled variants: " << stats.handled_variants << " Chromosome ID mismatches: " << stats.chr_id_mismatches << '\n';
			if (0 == stats.handled_variants) std::cerr << "WARNING: no variants matched the chromosome identifier \"" << opt.chromosome << "\".\n";
		}

		if (opt.output_graph) {                                 // main.cc:418-426
			std::cerr << "Outputting the variant graph..." << std::flush;
			vh::write_graph(graph, opt.output_graph);
			std::cerr << " Done.\n";
		}

		if (opt.graph_statistics) {   // main.cc:428-435
			std::cout << "Nodes:        " << graph.node_count() << '\n';
			std::cout << "ALT edges:    " << graph.edge_count() << '\n';
			std::cout << "Total ploidy: " << graph.total_chromosome_copies() << '\n';
		}

		// Graph and reference are replicated on every GPU; the path matrix is not.  With several GPUs, GPU k
		// receives the bits of its own chromosome copies only, transposes them itself and splices the rows of those copies:
		//  - an aligned A2M file as the only output: contiguous blocks of copies, every GPU's thread writes its rows at their
		//    final file offsets (all aligned rows have the same length);
		//  - every output that has to leave in row order (pipes, unaligned A2M, one file per sequence): the copies dealt
		//    round-robin in blocks of 8, so that every GPU always has rows that are due soon, handed to the writer in turns.
		// Founder rows read copies all over the matrix: every GPU then holds all of it.
		vh::gpu_context &gpu(first_gpu());
		std::vector<vh::gpu_context *> all_gpus;
		for (auto &g : contexts) all_gpus.push_back(g.get());
		bool const several(opt.haplotypes && all_gpus.size() > 1);
		bool const sharded(several && opt.output_sequences_a2m && !opt.pipe && !opt.unaligned && !opt.output_sequences_separate);
		bool const interleaved(several && !sharded);
		vh::copy_interleave const deal{8, vh::u32(all_gpus.size())};
		std::vector<vh::copy_shard> shards;
		std::future<void> founder_graph_uploaded;
		if (opt.founder_mode) {
			// The first context gets the graph with its (transposed) path matrix now: the cut search walks its chunks there
			// (v2m_pbwt_cut_trials).  The other contexts' uploads and the output path's buffers are set up while the search runs.
			vh::upload_graph(gpu, ref_seq, graph, true);
			founder_graph_uploaded = std::async(std::launch::async, [&] {
				for (std::size_t k(1); k < all_gpus.size(); ++k) vh::upload_graph(*all_gpus[k], ref_seq, graph, true);
				for (std::size_t k(1); k < all_gpus.size(); ++k) vh::warm_up_sink(*all_gpus[k], opt.unaligned);
			});
		}
		if (!opt.founder_mode) {
			// every context's graph and matrix share go up on that context's own thread (one PCIe link each)
			std::vector<std::exception_ptr> upload_errors(all_gpus.size());
			std::vector<std::thread> uploaders;
			for (std::size_t k(0); k < all_gpus.size(); ++k) {
				vh::copy_shard const shard(sharded
					? vh::shard_copies(graph.total_chromosome_copies(), vh::u32(all_gpus.size()), vh::u32(k))
					: vh::copy_shard{0, graph.paths_by_edge_and_chrom_copy.rows});
				if (sharded) shards.push_back(shard);
				if (opt.verbose && sharded) std::cerr << "GPU context " << k << " (device " << opt.devices[k] << "): chromosome copies [" << shard.first << ", " << shard.end << ")\n";
				if (opt.verbose && interleaved) std::cerr << "GPU context " << k << " (device " << opt.devices[k] << "): chromosome copies " << deal.block * k << " + " << deal.block * deal.world << " j + [0, " << deal.block << "), j = 0, 1, ...\n";
				uploaders.emplace_back([&, k, shard] {
					try {
						vh::upload_graph(*all_gpus[k], ref_seq, graph, false);
						if (interleaved) vh::upload_path_blocks(*all_gpus[k], graph, deal, vh::u32(k));
						else vh::upload_path_slice(*all_gpus[k], graph, shard);
					} catch (...) {
						upload_errors[k] = std::current_exception();
					}
				});
			}
			for (auto &u : uploaders) u.join();
			for (auto const &e : upload_errors) if (e) std::rethrow_exception(e);
		}
		progress_delegate delegate;
		delegate.verbose = opt.verbose;
		auto const do_output([&](vh::output &output) {   // main.cc:456-473
			if (opt.output_sequences_a2m) {
				std::cerr << "Outputting sequences as A2M...\n";
				output.output_a2m(graph, opt.output_sequences_a2m);
				std::cerr << "Done.\n";
			}
			if (opt.output_sequences_separate) {
				std::cerr << "Outputting sequences one by one..." << std::flush;
				output.output_separate(graph, !opt.separate_plain);
				std::cerr << " Done.\n";
			}
		});

		if (opt.haplotypes) {
			vh::haplotype_output output(gpu, opt.pipe, opt.dst_chromosome, !opt.omit_reference, opt.unaligned, delegate);
			for (std::size_t k(1); k < all_gpus.size(); ++k) output.add_gpu(*all_gpus[k]);
			if (sharded) output.set_copy_shards(shards);
			if (interleaved) output.set_copy_interleave(deal);
			do_output(output);
		} else {                                            // main.cc:487-550
			vh::founder_sequence_greedy_output output(gpu, opt.pipe, opt.dst_chromosome, !opt.omit_reference, opt.unaligned, delegate);
			for (std::size_t k(1); k < all_gpus.size(); ++k) output.add_gpu(*all_gpus[k]);
			std::vector<vh::u64> cuts;
			vh::u32 score(0);
			vh::gpu_founder_walker walker(gpu);   // the chunk walks of both searches run on the first context (v2m_pbwt_cut_trials / _records)
			vh::u64 cut_min_distance(vh::u64(opt.minimum_distance));
			if (opt.input_cut_positions) {                  // main.cc:499-500
				auto loaded(vh::read_cut_positions(opt.input_cut_positions));
				cuts = std::move(loaded.cut_positions);
				score = loaded.score;
				cut_min_distance = loaded.min_distance;
			} else {
				std::cerr << "Optimising cut positions...\n";
				score = vh::find_cut_positions(graph, vh::u64(opt.minimum_distance), cuts, 0, &walker);
				if (opt.verbose) std::cerr << "Cut search: " << walker.chunks_walked << " chunks walked on the GPU, " << walker.chunks_left << " on the host.\n";
				if (vh::kCutPositionScoreMax == score) { std::cerr << "ERROR: Unable to optimise cut positions.\n"; return EXIT_FAILURE; }
				if (opt.verbose) {
					std::cout << "Cut positions:";
					for (auto const cp : cuts) std::cout << ' ' << cp;
					std::cout << '\n';
				}
			}
			std::cout << "Maximum segmentation height: " << (1 + vh::u64(score)) << '\n';   // main.cc:520
			if (opt.output_cut_positions)                   // main.cc:522-523
				vh::write_cut_positions({cuts, cut_min_distance, score}, opt.output_cut_positions);
			std::cerr << "Finding matchings in the variant graph...\n";
			std::vector<vh::u32> assigned;
			// the first context is idle once the matching's chunk walks are back: its output buffers (a gigabyte of pinned memory,
			// 0.15 s) are set up on another thread while the greedy assignment runs here
			std::future<void> first_sink_warm;
			walker.on_last_walk = [&] { first_sink_warm = std::async(std::launch::async, [&] { vh::warm_up_sink(gpu, opt.unaligned); }); };
			if (!vh::find_matchings(graph, cuts, vh::u32(opt.founder_sequences), opt.keep_ref_edges, assigned, 0, &walker)) { std::cerr << "ERROR: Unable to find matchings.\n"; return EXIT_FAILURE; }
			if (opt.verbose) {                              // main.cc:534-545
				std::cout << "Matchings:\n";
				std::size_t const rows(cuts.size() - 1);
				for (long col(0); col < opt.founder_sequences; ++col) {
					std::cout << col << ':';
					for (std::size_t r(0); r < rows; ++r) std::cout << '\t' << assigned[col * rows + r];
					std::cout << '\n';
				}
			}
			output.set_cut_positions(std::move(cuts));
			output.set_assigned_samples(std::move(assigned), vh::u32(opt.founder_sequences));
			founder_graph_uploaded.get();
			if (first_sink_warm.valid()) first_sink_warm.get(); else vh::warm_up_sink(gpu, opt.unaligned);
			do_output(output);
		}
	} catch (vh::gpu_error const &e) {
		std::cerr << "ERROR (GPU path, code " << e.code << "): " << e.what() << '\n';
		return EXIT_FAILURE;
	} catch (std::exception const &e) {
		std::cerr << "ERROR: " << e.what() << '\n';
		return EXIT_FAILURE;
	}
	return EXIT_SUCCESS;
}
