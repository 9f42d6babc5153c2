#include "readers.hh"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstring>
#include <fstream>
#include <stdexcept>

#include "graph_builder.hh"

namespace v2m::host {

bool read_single_fasta_sequence(char const *path, sequence_type &seq, char const *seq_id)
{
	std::ifstream is(path, std::ios::binary);
	if (!is) return false;
	seq.clear();
	std::string line;
	bool wanted(false), found(false);
	while (std::getline(is, line)) {
		if (!line.empty() && '\r' == line.back()) line.pop_back();
		if (!line.empty() && '>' == line.front()) {
			if (found) break;
			auto const stop(line.find_first_of(" \t", 1));
			std::string_view const id(std::string_view(line).substr(1, std::string::npos == stop ? std::string::npos : stop - 1));
			wanted = !seq_id || id == seq_id;
			found = wanted;
			continue;
		}
		if (wanted) seq.insert(seq.end(), line.begin(), line.end());
	}
	return found;
}


namespace {

// Read-only mapping of a whole file (the reference maps the VCF too: vcf::mmap_input, variant_graph.cc:133-134).
struct mapped_file {
	char const *data{};
	std::size_t size{};
	int fd{-1};

	explicit mapped_file(char const *path)
	{
		fd = ::open(path, O_RDONLY);
		if (fd < 0) throw std::runtime_error(std::string("unable to open ") + path);
		struct stat st;
		if (0 != ::fstat(fd, &st)) { ::close(fd); throw std::runtime_error(std::string("unable to stat ") + path); }
		size = std::size_t(st.st_size);
		if (size) {
			void *p(::mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0));
			if (MAP_FAILED == p) { ::close(fd); throw std::runtime_error(std::string("unable to map ") + path); }
			data = static_cast<char const *>(p);
			::madvise(p, size, MADV_SEQUENTIAL);
		}
	}
	~mapped_file()
	{
		if (data) ::munmap(const_cast<char *>(data), size);
		if (fd >= 0) ::close(fd);
	}
};

// Splits [begin, end) at `delim` without allocating: call next() until it returns false.
struct field_cursor {
	char const *p, *end;
	char delim;
	bool done{false};
	field_cursor(std::string_view s, char d) : p(s.data()), end(s.data() + s.size()), delim(d) {}
	bool next(std::string_view &out)
	{
		if (done) return false;
		char const *q(static_cast<char const *>(std::memchr(p, delim, std::size_t(end - p))));
		if (!q) { out = std::string_view(p, std::size_t(end - p)); done = true; return true; }
		out = std::string_view(p, std::size_t(q - p));
		p = q + 1;
		return true;
	}
};

struct included_copy {
	u32 sample_vcf_index, copy_vcf_index, row;   // row = ploidy_csum[output sample] + output copy
};

[[noreturn]] void bad(u64 lineno, char const *what)
{
	throw std::runtime_error("VCF line " + std::to_string(lineno) + ": " + what);
}

} // namespace


void build_variant_graph(
	sequence_type const &ref_seq, char const *variants_path, char const *chr_id,
	variant_graph &graph, build_graph_statistics &stats, build_graph_delegate &delegate)
{
	mapped_file const file(variants_path);
	std::string_view const text(file.data, file.size);
	std::string_view const ref_sv(ref_seq.data(), ref_seq.size());
	std::string_view const wanted_chr(chr_id);

	graph = variant_graph{};
	graph_builder builder(graph, /* track_paths */ true);

	std::vector<std::string> vcf_sample_names;
	std::vector<included_copy> included;          // sorted by sample_vcf_index, then copy
	std::vector<std::string_view> sample_fields;
	std::vector<alt_allele> alts;
	bool is_first(true);
	u64 lineno(0), var_idx(0);
	std::string_view cur_id;          // what the overlap callback reports: the record and copy being handled
	u32 cur_sample(0), cur_copy(0);

	std::size_t pos(0);
	while (pos < text.size()) {
		std::size_t eol(text.find('\n', pos));
		if (std::string_view::npos == eol) eol = text.size();
		std::string_view line(text.substr(pos, eol - pos));
		pos = eol + 1;
		++lineno;
		if (!line.empty() && '\r' == line.back()) line.remove_suffix(1);
		if (line.empty()) continue;
		if ('#' == line.front()) {
			if (line.substr(0, 6) == "#CHROM") {
				field_cursor fc(line, '\t');
				std::string_view f;
				for (unsigned i(0); fc.next(f); ++i)
					if (i >= 9) vcf_sample_names.emplace_back(f);
			}
			continue;
		}

		++var_idx;
		field_cursor fc(line, '\t');
		std::string_view chrom, pos_f, id, ref, alt_f, skip, format;
		if (!(fc.next(chrom) && fc.next(pos_f) && fc.next(id) && fc.next(ref) && fc.next(alt_f) && fc.next(skip) && fc.next(skip) && fc.next(skip)))
			bad(lineno, "fewer than 8 columns");
		if (chrom != wanted_chr) { ++stats.chr_id_mismatches; continue; }                 // variant_graph.cc:203-207
		if (!fc.next(format)) bad(lineno, "variant does not have a genotype");              // :209-213
		std::size_t gt_index(SIZE_MAX);
		{
			field_cursor ff(format, ':');
			std::string_view f;
			for (std::size_t i(0); ff.next(f); ++i) if (f == "GT") { gt_index = i; break; }
			if (SIZE_MAX == gt_index) bad(lineno, "variant does not have a genotype");
		}
		sample_fields.clear();
		for (std::string_view f; fc.next(f);) sample_fields.push_back(f);
		if (sample_fields.size() != vcf_sample_names.size()) bad(lineno, "sample column count differs from the header");

		auto gt_of([&](std::size_t sample) -> std::string_view {
			field_cursor sf(sample_fields[sample], ':');
			std::string_view f;
			for (std::size_t i(0); sf.next(f); ++i) if (i == gt_index) return f;
			bad(lineno, "sample without GT");
		});

		if (is_first) {                                                                     // :215-288
			is_first = false;
			std::vector<std::string> names;
			std::vector<u32> ploidies;
			u32 row(0);
			for (std::size_t s(0); s < vcf_sample_names.size(); ++s) {
				// alleles are separated by '|' or '/': count them
				std::string_view const gt(gt_of(s));
				u32 ploidy(1);
				for (char const c : gt) if ('|' == c || '/' == c) ++ploidy;
				u32 kept(0);
				for (u32 c(0); c < ploidy; ++c) {
					if (delegate.should_include(vcf_sample_names[s], c)) {
						included.push_back({u32(s), c, row++});
						++kept;
					}
				}
				if (kept) { names.push_back(vcf_sample_names[s]); ploidies.push_back(kept); }   // samples with no included copy are dropped (:250-273)
			}
			builder.begin(std::move(names), ploidies);
			builder.on_overlap([&](overlap_info const &o) {
				delegate.report_overlapping_alternative(lineno, o.ref_pos, cur_id, vcf_sample_names[cur_sample], cur_copy, o.alt_number);
			});
		}

		++stats.handled_variants;
		u64 ref_pos(0);
		if (pos_f.empty()) bad(lineno, "empty POS");
		for (char const c : pos_f) { if (c < '0' || '9' < c) bad(lineno, "bad POS"); ref_pos = 10 * ref_pos + u64(c - '0'); }
		if (0 == ref_pos) bad(lineno, "POS must be 1-based");
		--ref_pos;                                                                           // zero_based_pos (:292)

		{                                                                                    // :307-314
			std::string_view const expected(ref_pos <= ref_sv.size() ? ref_sv.substr(ref_pos, ref.size()) : std::string_view{});
			if (ref != expected && !delegate.ref_column_mismatch(var_idx, ref_pos, ref, expected))
				return;
		}

		cur_id = id;
		alts.clear();
		{
			field_cursor ac(alt_f, ',');
			for (std::string_view a; ac.next(a);) alts.push_back({classify_alt(a), a});
		}
		if (!builder.add_record(ref_pos, ref.size(), alts.data(), alts.size()))
			throw std::runtime_error("variant " + std::to_string(var_idx) + " has non-increasing position");   // :293-297

		// genotypes of the included copies (:379-425)
		std::size_t k(0);
		while (k < included.size()) {
			u32 const s(included[k].sample_vcf_index);
			std::string_view const gt(gt_of(s));
			// walk the alleles of this sample once
			u32 copy(0);
			std::size_t a(0);
			while (a <= gt.size() && k < included.size() && included[k].sample_vcf_index == s) {
				std::size_t b(a);
				while (b < gt.size() && '|' != gt[b] && '/' != gt[b]) ++b;
				if (included[k].copy_vcf_index == copy) {
					std::string_view const tok(gt.substr(a, b - a));
					if (tok.empty()) bad(lineno, "empty GT allele");
					if (tok != ".") {                                                        // NULL_ALLELE: skipped (:396-397)
						u32 allele(0);
						for (char const c : tok) { if (c < '0' || '9' < c) bad(lineno, "bad GT allele"); allele = 10 * allele + u32(c - '0'); }
						if (allele) {
							if (allele > alts.size()) bad(lineno, "GT allele exceeds the ALT count");
							cur_sample = s; cur_copy = copy;
							builder.set_genotype(included[k].row, allele);
						}
					}
					++k;
				}
				++copy;
				a = b + 1;
			}
			if (k < included.size() && included[k].sample_vcf_index == s)
				bad(lineno, "GT has fewer alleles than in the first record");              // libbio_assert_lt(chr_idx_input, gt.size()), :390
		}
	}

	if (is_first) {
		// No record on the requested chromosome: the reference leaves ploidy_csum empty and then reads it
		// out of bounds (hard part 10).  Here: all samples, ploidy 0 rows, REF only.
		builder.begin({}, {});
	}
	builder.finish(ref_seq.size());                                                          // :437-451
}

} // namespace v2m::host
