#include "synth.hh"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "../host/graph_builder.hh"

namespace v2m::synth {

namespace {

struct xoshiro256ss {
	u64 s[4];
	explicit xoshiro256ss(u64 seed)
	{
		for (auto &v : s) { seed += 0x9E3779B97F4A7C15ULL; v = mix64(seed); }
	}
	static u64 rotl(u64 x, int k) { return (x << k) | (x >> (64 - k)); }
	u64 next()
	{
		u64 const result(rotl(s[1] * 5, 7) * 9);
		u64 const t(s[1] << 17);
		s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
		s[2] ^= t;
		s[3] = rotl(s[3], 45);
		return result;
	}
	double uniform() { return double(next() >> 11) * (1.0 / 9007199254740992.0); }
	u64 below(u64 n) { return u64((__uint128_t(next()) * n) >> 64); }
};

constexpr char kBases[4] = {'A', 'C', 'G', 'T'};

char other_base(xoshiro256ss &rng, char ref)
{
	char c;
	do c = kBases[rng.below(4)]; while (c == ref);
	return c;
}

u32 geometric_mean3(xoshiro256ss &rng, u32 cap)
{
	u32 k(1);
	while (k < cap && rng.uniform() >= 1.0 / 3.0) ++k;
	return k;
}

u32 draw_threshold(xoshiro256ss &rng)
{
	double const f(0.5 * std::pow(10.0, -3.0 * rng.uniform()));
	return u32(f * 4294967296.0);
}

} // namespace


void generate(config const &cfg, dataset &out)
{
	xoshiro256ss rng(cfg.seed);
	u64 const R(cfg.ref_length);

	out.reference.resize(R);
	for (u64 i(0); i < R; i += 32) {            // 2 bits per base, 32 bases per draw
		u64 bits(rng.next());
		for (u64 j(i); j < std::min(R, i + 32); ++j, bits >>= 2)
			out.reference[j] = kBases[bits & 3];
	}

	// distinct sorted sites
	std::vector<u64> sites;
	u64 const span(R > 64 ? R - 64 : 1);
	u64 const want(std::min(cfg.n_variants, span));
	sites.reserve(want + want / 8);
	while (sites.size() < want) {
		u64 const missing(want - sites.size());
		for (u64 i(0); i < missing + missing / 16 + 8; ++i) sites.push_back(rng.below(span));
		std::sort(sites.begin(), sites.end());
		sites.erase(std::unique(sites.begin(), sites.end()), sites.end());
	}
	if (sites.size() > want) {
		// drop the surplus at evenly spread ranks so the rest stays uniform
		std::vector<u64> kept;
		kept.reserve(want);
		u64 const n(sites.size());
		for (u64 i(0); i < want; ++i) kept.push_back(sites[i * n / want]);
		sites.swap(kept);
	}

	out.graph = host::variant_graph{};
	out.records.clear();
	out.edge_thresholds.clear();
	host::graph_builder builder(out.graph, /* track_paths */ false);
	std::string alt_a, alt_b;
	for (u64 const pos : sites) {
		double const u(rng.uniform());
		char const ref_base(out.reference[pos]);
		host::alt_allele alts[2];
		std::size_t n_alts(1);
		u64 ref_len(1);
		alt_a.clear();
		double edge(cfg.frac_mnp);
		if (u < edge) {
			ref_len = 2 + rng.below(3);
			for (u64 i(0); i < ref_len; ++i) alt_a.push_back(other_base(rng, out.reference[pos + i]));
		} else if (u < (edge += cfg.frac_insertion)) {
			u32 const k(geometric_mean3(rng, cfg.max_indel));
			alt_a.push_back(ref_base);
			for (u32 i(0); i < k; ++i) alt_a.push_back(kBases[rng.below(4)]);
		} else if (u < (edge += cfg.frac_deletion)) {
			ref_len = 1 + geometric_mean3(rng, cfg.max_indel);
			alt_a.push_back(ref_base);
		} else if (u < (edge += cfg.frac_multiallelic)) {
			alt_a.push_back(other_base(rng, ref_base));
			alt_b.clear();
			alt_b.push_back(ref_base);
			u32 const k(geometric_mean3(rng, cfg.max_indel));
			for (u32 i(0); i < k; ++i) alt_b.push_back(kBases[rng.below(4)]);
			alts[1] = {host::alt_kind::sequence, alt_b};
			n_alts = 2;
		} else {
			alt_a.push_back(other_base(rng, ref_base));
		}
		alts[0] = {host::alt_kind::sequence, alt_a};
		builder.add_record(pos, ref_len, alts, n_alts);
		{
			record r;
			r.pos = pos; r.ref_length = u32(ref_len); r.n_alts = u32(n_alts); r.first_edge = builder.edge_for_alt(0);
			r.alts[0] = alt_a;
			if (n_alts > 1) r.alts[1] = alt_b;
			out.records.push_back(std::move(r));
		}
		for (std::size_t a(0); a < n_alts; ++a) out.edge_thresholds.push_back(draw_threshold(rng));
	}
	builder.finish(R);
}


bool write_fasta_and_vcf(dataset const &ds, u64 seed, u32 samples, u32 ploidy, char const *chromosome, char const *fasta_path, char const *vcf_path)
{
	{
		FILE *f(std::fopen(fasta_path, "wb"));
		if (!f) return false;
		std::fprintf(f, ">%s synthetic\n", chromosome);
		for (u64 i(0); i < ds.reference.size(); i += 80) {
			u64 const n(std::min<u64>(80, ds.reference.size() - i));
			std::fwrite(ds.reference.data() + i, 1, n, f);
			std::fputc('\n', f);
		}
		if (0 != std::fclose(f)) return false;
	}
	FILE *f(std::fopen(vcf_path, "wb"));
	if (!f) return false;
	std::vector<char> buf(1 << 22);
	std::setvbuf(f, buf.data(), _IOFBF, buf.size());
	std::fputs("##fileformat=VCFv4.2\n##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT", f);
	for (u32 s(0); s < samples; ++s) std::fprintf(f, "\tS%u", s);
	std::fputc('\n', f);
	// The genotype text is most of the work (config 3: 10 GB, 5 G fields): blocks of records are formatted by several
	// threads at once and written in order.
	auto const format_block([&](u64 first, u64 end, std::string &text) {
		text.clear();
		for (u64 idx(first); idx < end; ++idx) {
			auto const &r(ds.records[idx]);
			text += chromosome; text += '\t'; text += std::to_string(r.pos + 1); text += "\tv"; text += std::to_string(idx); text += '\t';
			text.append(ds.reference, r.pos, r.ref_length); text += '\t';
			text += r.alts[0];
			if (r.n_alts > 1) { text += ','; text += r.alts[1]; }
			text += "\t.\tPASS\t.\tGT";
			std::size_t at(text.size());
			text.resize(at + std::size_t(samples) * 2 * ploidy);                // per sample: a tab, then alleles separated by '|'
			for (u32 s(0); s < samples; ++s) {
				text[at++] = '\t';
				for (u32 c(0); c < ploidy; ++c) {
					if (c) text[at++] = '|';
					u64 const copy(u64(s) * ploidy + c);
					char allele('0');
					for (u32 a(0); a < r.n_alts; ++a)
						if (path_bit(seed, r.first_edge + a, copy, ds.edge_thresholds[r.first_edge + a])) { allele = char('1' + a); break; }
					text[at++] = allele;
				}
			}
			text += '\n';
		}
	});
	u64 const n_records(ds.records.size());
	unsigned const n_threads(std::min(16u, std::max(1u, std::thread::hardware_concurrency())));
	u64 const block(std::max<u64>(1, std::min<u64>(4096, (u64(32) << 20) / (64 + u64(samples) * 2 * ploidy))));   // about 32 MB of text
	std::vector<std::string> texts(n_threads);
	for (u64 round_first(0); round_first < n_records; round_first += block * n_threads) {
		std::vector<std::thread> threads;
		for (unsigned t(0); t < n_threads; ++t) {
			u64 const first(round_first + block * t);
			if (first >= n_records) { texts[t].clear(); continue; }
			threads.emplace_back(format_block, first, std::min(n_records, first + block), std::ref(texts[t]));
		}
		for (auto &th : threads) th.join();
		for (auto const &text : texts)
			if (!text.empty() && text.size() != std::fwrite(text.data(), 1, text.size(), f)) { std::fclose(f); return false; }
	}
	return 0 == std::fclose(f);
}

} // namespace v2m::synth
