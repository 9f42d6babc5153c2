// synth.hh -- deterministic synthetic inputs of the shapes BASELINE.json names (bench + scale tests).
//
// Reference bases iid over ACGT; V distinct variant sites uniform in [0, R-64), sorted; per site a
// record type drawn from the configured mix; indel length k ~ Geom(mean 3) capped at 32; per ALT an
// allele frequency f = 0.5 * 10^(-3u), u ~ U(0,1).  The records are pushed through the product's own
// graph_builder (so nodes/edges/aligned positions are exactly what the VCF path would produce), while
// the genotype bits -- 5 * 10^9 of them at config 3 -- are generated directly in HBM by
// fill_paths_kernel from a counter-based hash, which any row can be re-derived from on the CPU.
// PRNG: xoshiro256** seeded through splitmix64.
#pragma once

#include <string>
#include <vector>

#include "../host/variant_graph.hh"

namespace v2m::synth {

using host::u32;
using host::u64;

struct config {
	u64 seed{};
	u64 ref_length{};
	u64 n_variants{};
	// fractions of record types; the rest is SNV
	double frac_mnp{};        // REF and ALT of equal length 2-4
	double frac_insertion{};  // REF 1 base, ALT 1 + k
	double frac_deletion{};   // REF 1 + k, ALT 1 base
	double frac_multiallelic{};   // two ALTs: an SNV and an insertion
	u32 max_indel{32};
};

struct record {
	u64 pos{};            // 0-based
	u32 ref_length{};
	u32 n_alts{};
	u64 first_edge{};     // edge of ALT 1; ALT k is first_edge + k - 1
	std::string alts[2];
};

struct dataset {
	std::vector<record> records;
	std::string reference;
	host::variant_graph graph;            // nodes, edges, labels; no path matrices
	std::vector<u32> edge_thresholds;     // P(copy carries edge) * 2^32, per edge
};

void generate(config const &cfg, dataset &out);

// The same dataset as text: a single-sequence FASTA and a VCF with `samples` samples of the given ploidy whose
// genotypes follow the genotype hash (per copy: the first ALT of the record whose edge bit is set, else 0), so
// that the text pipeline can be run on exactly what the direct generator produces.  Returns false on I/O errors.
bool write_fasta_and_vcf(dataset const &ds, u64 seed, u32 samples, u32 ploidy, char const *chromosome, char const *fasta_path, char const *vcf_path);

// The genotype hash shared by the device kernel and the CPU check.
inline u64 mix64(u64 z)
{
	z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
	z ^= z >> 27; z *= 0x94D049BB133111EBULL;
	z ^= z >> 31;
	return z;
}

inline bool path_bit(u64 seed, u64 edge, u64 copy, u32 threshold)
{
	return u32(mix64(seed + edge * 0x9E3779B97F4A7C15ULL + copy * 0xC2B2AE3D27D4EB4FULL) >> 32) < threshold;
}

} // namespace v2m::synth
