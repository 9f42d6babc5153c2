#include "readers.hh"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <condition_variable>
#include <cstdlib>
#include <cstdio>
#include <chrono>
#include <algorithm>
#include <memory>
#include <exception>
#include <functional>
#include <cstring>
#include <fstream>
#include <mutex>
#include <stdexcept>
#include <thread>

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

[[noreturn]] void bad(u64 lineno, char const *what)
{
	throw std::runtime_error("VCF line " + std::to_string(lineno) + ": " + what);
}

// ---- record parsing (runs on worker threads) ------------------------------------------------------------------

struct parsed_genotype { u32 row, alt_number; };   // (which sample and copy a row is comes from the context when an overlap is reported: 8 bytes x 360 M genotypes at config 3)

struct parsed_record {
	u64 line_in_chunk;        // 1-based within the chunk
	u64 data_line_in_chunk;   // counts only non-header lines
	u64 ref_pos;
	std::string_view id, ref;
	u32 alt_begin, n_alts;
	u64 geno_begin, geno_end;
	u64 chr_mismatches_before;   // records of other chromosomes seen in the chunk before this one
};

struct parsed_chunk {
	std::vector<parsed_record> records;
	std::vector<alt_allele> alts;
	std::vector<parsed_genotype> genos;
	u64 n_lines{}, n_data_lines{}, chr_mismatches{};
	std::string error;        // first error, with its chunk-relative line in error_line
	u64 error_line{};
};

struct parse_context {
	std::string_view wanted_chr;
	std::size_t n_samples{};
	// (sample, copy) -> row of paths_by_edge_and_chrom_copy, or -1 when the copy is not included;
	// copies of sample s are row_lookup[copy_begin[s] .. copy_begin[s + 1])
	std::vector<std::int32_t> row_lookup;
	std::vector<u32> copy_begin;
};

struct chunk_error { u64 line; char const *what; };

// "0|0\t" and "0/0\t" as the four bytes a little-endian load sees
constexpr std::uint32_t kRefRefPhased = 0x09307C30u, kRefRefUnphased = 0x09302F30u;
static_assert(__BYTE_ORDER__ == __ORDER_LITTLE_ENDIAN__, "the column fast path compares little-endian words");

// Parses the lines of text[begin, end) (whole lines) into `out`.
void parse_chunk(std::string_view text, parse_context const &ctx, parsed_chunk &out)
{
	std::size_t pos(0);
	try {
		while (pos < text.size()) {
			std::size_t eol(text.find('\n', pos));
			if (std::string_view::npos == eol) eol = text.size();
			std::string_view line(text.substr(pos, eol - pos));
			pos = eol + 1;
			++out.n_lines;
			if (!line.empty() && '\r' == line.back()) line.remove_suffix(1);
			if (line.empty() || '#' == line.front()) continue;
			++out.n_data_lines;

			field_cursor fc(line, '\t');
			std::string_view chrom, pos_f, id, ref, alt_f, skip, format;
			if (!(fc.next(chrom) && fc.next(pos_f) && fc.next(id) && fc.next(ref) && fc.next(alt_f) && fc.next(skip) && fc.next(skip) && fc.next(skip)))
				throw chunk_error{out.n_lines, "fewer than 8 columns"};
			if (chrom != ctx.wanted_chr) { ++out.chr_mismatches; continue; }              // variant_graph.cc:203-207
			if (!fc.next(format)) throw chunk_error{out.n_lines, "variant does not have a genotype"};   // :209-213
			std::size_t gt_index(SIZE_MAX);
			{
				field_cursor ff(format, ':');
				std::string_view f;
				for (std::size_t i(0); ff.next(f); ++i) if (f == "GT") { gt_index = i; break; }
				if (SIZE_MAX == gt_index) throw chunk_error{out.n_lines, "variant does not have a genotype"};
			}

			parsed_record rec{};
			rec.line_in_chunk = out.n_lines;
			rec.data_line_in_chunk = out.n_data_lines;
			rec.chr_mismatches_before = out.chr_mismatches;
			rec.id = id;
			rec.ref = ref;
			if (pos_f.empty()) throw chunk_error{out.n_lines, "empty POS"};
			for (char const c : pos_f) { if (c < '0' || '9' < c) throw chunk_error{out.n_lines, "bad POS"}; rec.ref_pos = 10 * rec.ref_pos + u64(c - '0'); }
			if (0 == rec.ref_pos) throw chunk_error{out.n_lines, "POS must be 1-based"};
			--rec.ref_pos;                                                                 // zero_based_pos (:292)

			rec.alt_begin = u32(out.alts.size());
			{
				field_cursor ac(alt_f, ',');
				for (std::string_view a; ac.next(a);) out.alts.push_back({classify_alt(a), a});
			}
			rec.n_alts = u32(out.alts.size()) - rec.alt_begin;

			// genotypes of the included copies (:379-425); allele 0 and '.' are not recorded (:393-397)
			rec.geno_begin = out.genos.size();
			std::size_t sample(0);
			for (std::string_view field; ; ++sample) {
				// Most of a population-scale VCF is "0|0\t": both copies on the reference allele, nothing to record (:393-397).  Runs of
				// such columns are skipped four bytes at a time instead of a memchr and three loops each (config 3: 2.5 G columns; GT must
				// be the first FORMAT key and the sample at most diploid, so that the general path below would find nothing either).
				if (0 == gt_index) {
					while (!fc.done && fc.end - fc.p >= 4 && sample < ctx.n_samples && ctx.copy_begin[sample + 1] - ctx.copy_begin[sample] <= 2) {
						std::uint32_t four;
						std::memcpy(&four, fc.p, 4);
						if (kRefRefPhased != four && kRefRefUnphased != four) break;
						fc.p += 4;
						++sample;
					}
				}
				if (!fc.next(field)) break;
				if (sample >= ctx.n_samples) throw chunk_error{out.n_lines, "more sample columns than in the header"};
				std::string_view gt(field);
				if (gt_index || std::string_view::npos != field.find(':')) {
					field_cursor sf(field, ':');
					std::size_t i(0);
					bool found(false);
					for (std::string_view f; sf.next(f); ++i) if (i == gt_index) { gt = f; found = true; break; }
					if (!found) throw chunk_error{out.n_lines, "sample without GT"};
				}
				u32 const c_begin(ctx.copy_begin[sample]), c_end(ctx.copy_begin[sample + 1]);
				u32 copy(0);
				std::size_t a(0);
				while (a <= gt.size()) {
					std::size_t b(a);
					while (b < gt.size() && '|' != gt[b] && '/' != gt[b]) ++b;
					if (c_begin + copy < c_end) {
						std::int32_t const row(ctx.row_lookup[c_begin + copy]);
						if (row >= 0) {
							std::string_view const tok(gt.substr(a, b - a));
							if (tok.empty()) throw chunk_error{out.n_lines, "empty GT allele"};
							if (tok != ".") {
								u32 allele(0);
								for (char const c : tok) { if (c < '0' || '9' < c) throw chunk_error{out.n_lines, "bad GT allele"}; allele = 10 * allele + u32(c - '0'); }
								if (allele) {
									if (allele > rec.n_alts) throw chunk_error{out.n_lines, "GT allele exceeds the ALT count"};
									out.genos.push_back({u32(row), allele});
								}
							}
						}
					}
					++copy;
					a = b + 1;
				}
				for (u32 c(copy); c_begin + c < c_end; ++c)                               // libbio_assert_lt(chr_idx_input, gt.size()), :390
					if (ctx.row_lookup[c_begin + c] >= 0) throw chunk_error{out.n_lines, "GT has fewer alleles than in the first record"};
			}
			if (sample != ctx.n_samples) throw chunk_error{out.n_lines, "sample column count differs from the header"};
			rec.geno_end = out.genos.size();
			out.records.push_back(rec);
		}
	} catch (chunk_error const &e) {
		out.error = e.what;
		out.error_line = e.line;
	}
}

} // namespace


void build_variant_graph(
	sequence_type const &ref_seq, char const *variants_path, char const *chr_id,
	variant_graph &graph, build_graph_statistics &stats, build_graph_delegate &delegate, unsigned threads, u64 path_alignment)
{
	mapped_file const file(variants_path);
	std::string_view const text(file.data, file.size);
	std::string_view const ref_sv(ref_seq.data(), ref_seq.size());

	graph = variant_graph{};
	graph_builder builder(graph, /* track_paths */ true, path_alignment);

	// ---- header, then the first record on the requested chromosome: it fixes ploidy and inclusion (:215-288) ----
	parse_context ctx;
	ctx.wanted_chr = chr_id;
	std::vector<std::string> vcf_sample_names;
	std::size_t body_begin(0);
	u64 header_lines(0);
	{
		std::size_t pos(0);
		while (pos < text.size()) {
			std::size_t eol(text.find('\n', pos));
			if (std::string_view::npos == eol) eol = text.size();
			std::string_view line(text.substr(pos, eol - pos));
			if (!line.empty() && '\r' == line.back()) line.remove_suffix(1);
			if (!line.empty() && '#' != line.front()) break;
			if (line.substr(0, 6) == "#CHROM") {
				field_cursor fc(line, '\t');
				std::string_view f;
				for (unsigned i(0); fc.next(f); ++i) if (i >= 9) vcf_sample_names.emplace_back(f);
			}
			pos = eol + 1;
			++header_lines;
		}
		body_begin = std::min(pos, text.size());
	}
	ctx.n_samples = vcf_sample_names.size();
	ctx.copy_begin.assign(ctx.n_samples + 1, 0);
	bool have_first(false);
	{
		std::size_t pos(body_begin);
		u64 lineno(header_lines);
		while (pos < text.size() && !have_first) {
			std::size_t eol(text.find('\n', pos));
			if (std::string_view::npos == eol) eol = text.size();
			std::string_view line(text.substr(pos, eol - pos));
			pos = eol + 1;
			++lineno;
			if (!line.empty() && '\r' == line.back()) line.remove_suffix(1);
			if (line.empty() || '#' == line.front()) continue;
			field_cursor fc(line, '\t');
			std::string_view f, format;
			if (!fc.next(f)) continue;
			if (f != ctx.wanted_chr) continue;
			for (int i(1); i < 8; ++i) if (!fc.next(f)) bad(lineno, "fewer than 8 columns");
			if (!fc.next(format)) bad(lineno, "variant does not have a genotype");
			std::size_t gt_index(SIZE_MAX);
			{
				field_cursor ff(format, ':');
				std::string_view x;
				for (std::size_t i(0); ff.next(x); ++i) if (x == "GT") { gt_index = i; break; }
				if (SIZE_MAX == gt_index) bad(lineno, "variant does not have a genotype");
			}
			std::vector<std::string> names;
			std::vector<u32> ploidies;
			u32 row(0);
			std::size_t s(0);
			for (std::string_view field; fc.next(field); ++s) {
				if (s >= ctx.n_samples) bad(lineno, "more sample columns than in the header");
				std::string_view gt;
				{
					field_cursor sf(field, ':');
					std::size_t i(0);
					bool found(false);
					for (std::string_view x; sf.next(x); ++i) if (i == gt_index) { gt = x; found = true; break; }
					if (!found) bad(lineno, "sample without GT");
				}
				u32 ploidy(1);
				for (char const c : gt) if ('|' == c || '/' == c) ++ploidy;
				u32 kept(0);
				for (u32 c(0); c < ploidy; ++c) {
					bool const inc(delegate.should_include(vcf_sample_names[s], c));       // :231
					ctx.row_lookup.push_back(inc ? std::int32_t(row) : -1);
					if (inc) { ++row; ++kept; }
				}
				ctx.copy_begin[s + 1] = u32(ctx.row_lookup.size());
				if (kept) { names.push_back(vcf_sample_names[s]); ploidies.push_back(kept); }   // samples with no included copy are dropped (:250-273)
			}
			if (s != ctx.n_samples) bad(lineno, "sample column count differs from the header");
			builder.begin(std::move(names), ploidies);
			have_first = true;
		}
	}
	if (!have_first) {
		// No record on the requested chromosome: the reference leaves ploidy_csum empty and then reads it
		// out of bounds (hard part 10).  Here: no samples, REF only.
		builder.begin({}, {});
		for (std::size_t pos(body_begin); pos < text.size();) {
			std::size_t eol(text.find('\n', pos));
			if (std::string_view::npos == eol) eol = text.size();
			if (eol > pos && '#' != text[pos]) ++stats.chr_id_mismatches;
			pos = eol + 1;
		}
		builder.finish(ref_seq.size());
		return;
	}

	// about one ALT edge per line of the size of the first record's (a hint for the path matrix's allocation, nothing more)
	{
		std::size_t const first_eol(text.find('\n', body_begin));
		std::size_t const line_bytes(std::max<std::size_t>(16, (std::string_view::npos == first_eol ? text.size() : first_eol) - body_begin + 1));
		builder.expect_edges(u64(double(text.size() - body_begin) / double(line_bytes) * 1.1) + 1024);
	}

	// ---- chunks of whole lines, parsed by worker threads, merged in file order -------------------------------------
	if (0 == threads) threads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
	std::size_t const target_chunk(std::size_t(8) << 20);
	std::vector<std::pair<std::size_t, std::size_t>> ranges;
	for (std::size_t b(body_begin); b < text.size();) {
		std::size_t e(std::min(text.size(), b + target_chunk));
		if (e < text.size()) {
			std::size_t const nl(text.find('\n', e));
			e = (std::string_view::npos == nl) ? text.size() : nl + 1;
		}
		ranges.emplace_back(b, e);
		b = e;
	}

	std::size_t const n_chunks(ranges.size());
	std::vector<parsed_chunk> chunks(n_chunks);
	std::vector<char> ready(n_chunks, 0);
	std::mutex mutex;
	std::condition_variable cv_ready, cv_window;
	std::size_t next_chunk(0), consumed(0);
	std::size_t const window(std::max<std::size_t>(2, 2 * threads));
	bool abort_workers(false);
	std::vector<std::vector<parsed_genotype>> spare_genos;           // (under `mutex`)

	auto const worker([&] {
		for (;;) {
			std::size_t idx;
			{
				std::unique_lock<std::mutex> lock(mutex);
				cv_window.wait(lock, [&] { return abort_workers || next_chunk >= n_chunks || next_chunk < consumed + window; });
				if (abort_workers || next_chunk >= n_chunks) return;
				idx = next_chunk++;
			}
			{
				// (a genotype list that an earlier chunk has been through: its pages are there already; fresh ones of this size
				// come from mmap every time and cost a fault per page)
				std::lock_guard<std::mutex> lock(mutex);
				if (!spare_genos.empty()) { chunks[idx].genos.swap(spare_genos.back()); spare_genos.pop_back(); }
			}
			parse_chunk(text.substr(ranges[idx].first, ranges[idx].second - ranges[idx].first), ctx, chunks[idx]);
			{
				std::lock_guard<std::mutex> lock(mutex);
				ready[idx] = 1;
			}
			cv_ready.notify_all();
		}
	});
	std::vector<std::thread> pool;
	if (threads > 1)
		for (unsigned t(0); t < std::min<std::size_t>(threads, n_chunks); ++t) pool.emplace_back(worker);
	struct joiner {
		std::vector<std::thread> &pool; std::mutex &mutex; std::condition_variable &cv; bool &abort_flag;
		~joiner()
		{
			{ std::lock_guard<std::mutex> lock(mutex); abort_flag = true; }
			cv.notify_all();
			for (auto &t : pool) t.join();
		}
	} const join_on_exit{pool, mutex, cv_window, abort_workers};

	u64 lineno_base(header_lines), var_idx(0);

	// row of the path matrix -> (sample column, copy of it), for the overlap reports
	std::vector<std::pair<u32, u32>> row_origin;
	for (std::size_t smp(0); smp < ctx.n_samples; ++smp)
		for (u32 c(ctx.copy_begin[smp]); c < ctx.copy_begin[smp + 1]; ++c)
			if (ctx.row_lookup[c] >= 0) {
				if (row_origin.size() <= std::size_t(ctx.row_lookup[c])) row_origin.resize(std::size_t(ctx.row_lookup[c]) + 1);
				row_origin[std::size_t(ctx.row_lookup[c])] = {u32(smp), c - ctx.copy_begin[smp]};
			}

	// The merge stage.  Per chunk, in file order: (A) the records go into the builder one by one -- nodes, edges, targets: cheap --
	// and what their genotypes will need is kept (graph_builder::snapshot_record); (B) the chunk's genotypes are applied, every
	// chromosome copy's in record order, by a few helper threads that own disjoint groups of 64 copies each (at population scale
	// the genotypes ARE the merge stage: config 3 has 360 M of them, 2 s of one thread); overlaps are reported afterwards in the
	// reference's order (record by record, copy by copy).
	struct apply_pool {
		std::vector<std::thread> threads;
		std::mutex mutex;
		std::condition_variable cv_start, cv_done;
		u64 generation{};
		unsigned remaining{};
		bool quit{};
		std::function<void(unsigned)> job;
		std::exception_ptr error;
		explicit apply_pool(unsigned n)
		{
			for (unsigned h(0); h < n; ++h) threads.emplace_back([this, h] {
				u64 seen(0);
				for (;;) {
					{
						std::unique_lock<std::mutex> lock(mutex);
						cv_start.wait(lock, [&] { return quit || generation != seen; });
						if (quit) return;
						seen = generation;
					}
					try { job(h); }
					catch (...) { std::lock_guard<std::mutex> lock(mutex); if (!error) error = std::current_exception(); }
					{
						std::lock_guard<std::mutex> lock(mutex);
						--remaining;
					}
					cv_done.notify_all();
				}
			});
		}
		~apply_pool()
		{
			{ std::lock_guard<std::mutex> lock(mutex); quit = true; }
			cv_start.notify_all();
			for (auto &t : threads) t.join();
		}
		void run(std::function<void(unsigned)> f)
		{
			std::unique_lock<std::mutex> lock(mutex);
			job = std::move(f);
			remaining = unsigned(threads.size());
			++generation;
			cv_start.notify_all();
			cv_done.wait(lock, [&] { return 0 == remaining; });
			if (error) { auto const e(error); error = nullptr; std::rethrow_exception(e); }
		}
	};
	u64 const n_copy_rows(builder.tracked_copies());
	unsigned const n_helpers(threads > 1 ? unsigned(std::min<u64>(std::min(threads, 8u), std::max<u64>(1, (n_copy_rows + 63) / 64))) : 1u);
	std::unique_ptr<apply_pool> helpers(n_helpers > 1 ? new apply_pool(n_helpers) : nullptr);
	u64 const rows_per_helper((((n_copy_rows + 63) / 64 + n_helpers - 1) / n_helpers) * 64);   // whole 64-copy groups: disjoint words of the bit matrix

	std::vector<graph_builder::record_snapshot> snaps;
	std::vector<u64> edge_slots;
	std::vector<std::vector<u64>> overlaps_found(n_helpers);         // per helper: indices into chunk.genos

	// V2M_READER_TIMING=1: where the merge stage's time went (waiting for parsed chunks / records into the builder / genotypes), to stderr
	bool const timing(nullptr != std::getenv("V2M_READER_TIMING"));
	double t_wait(0), t_records(0), t_genotypes(0);
	auto const now([] { return std::chrono::steady_clock::now(); });
	auto const since([&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(now() - t).count(); });
	struct report_timing {
		bool on; double const &w, &r, &g;
		~report_timing() { if (on) std::fprintf(stderr, "[vcf reader] merge stage: %.3f s waiting for parsed chunks, %.3f s records into the builder, %.3f s genotypes\n", w, r, g); }
	} const report{timing, t_wait, t_records, t_genotypes};

	for (std::size_t ci(0); ci < n_chunks; ++ci) {
		auto const t0(now());
		if (threads > 1) {
			std::unique_lock<std::mutex> lock(mutex);
			cv_ready.wait(lock, [&] { return 0 != ready[ci]; });
		} else {
			parse_chunk(text.substr(ranges[ci].first, ranges[ci].second - ranges[ci].first), ctx, chunks[ci]);
		}
		t_wait += since(t0);
		auto const t1(now());
		parsed_chunk &chunk(chunks[ci]);
		// (A) records parsed before an error are still merged first, so errors surface in file order
		snaps.clear();
		edge_slots.clear();
		parsed_record const *stopped_at(nullptr);
		for (auto const &rec : chunk.records) {
			++stats.handled_variants;
			u64 const this_var(var_idx + rec.data_line_in_chunk);
			// the reference's order: the position check (variant_graph.cc:293-297) comes before the REF comparison (:307-314)
			if (builder.would_go_back(rec.ref_pos))
				throw std::runtime_error("variant " + std::to_string(this_var) + " has non-increasing position");
			{                                                                            // :307-314
				std::string_view const expected(rec.ref_pos <= ref_sv.size() ? ref_sv.substr(rec.ref_pos, rec.ref.size()) : std::string_view{});
				if (rec.ref != expected && !delegate.ref_column_mismatch(this_var, rec.ref_pos, rec.ref, expected)) { stopped_at = &rec; break; }
			}
			if (!builder.add_record(rec.ref_pos, rec.ref.size(), chunk.alts.data() + rec.alt_begin, rec.n_alts))
				throw std::runtime_error("variant " + std::to_string(this_var) + " has non-increasing position");   // :293-297
			snaps.emplace_back();
			builder.snapshot_record(snaps.back(), edge_slots);
		}
		t_records += since(t1);
		auto const t2(now());
		// (B) the genotypes of the records that went in
		{
			auto const apply([&](unsigned h) {
				auto const usable([&](parsed_genotype const &gt, graph_builder::record_snapshot const &snap, u64 &edge) {
					if (0 == gt.alt_number || snap.n_alts < gt.alt_number) return false;             // (set_genotype's own checks)
					edge = edge_slots[snap.first_edge_slot + gt.alt_number - 1];
					return kEdgeMax != edge;                                                         // :401-403
				});
				// the path bits, by records: a run of records is a run of columns of the matrix, which nobody else writes
				{
					std::size_t const r_lo(snaps.size() * h / n_helpers), r_hi(snaps.size() * (h + 1) / n_helpers);
					for (std::size_t r(r_lo); r < r_hi; ++r) {
						auto const &rec(chunk.records[r]);
						for (u64 k(rec.geno_begin); k < rec.geno_end; ++k) {
							u64 edge;
							if (usable(chunk.genos[k], snaps[r], edge)) builder.set_path_bit(chunk.genos[k].row, edge);
						}
					}
				}
				// where every copy is after each record, by copies (a record's genotypes are in sample order = ascending rows: the
				// helper's rows are one run of them)
				u64 const row_lo(u64(h) * rows_per_helper), row_hi(n_helpers > 1 ? row_lo + rows_per_helper : UINT64_MAX);
				auto &found(overlaps_found[h]);
				found.clear();
				for (std::size_t r(0); r < snaps.size(); ++r) {
					auto const &rec(chunk.records[r]);
					u64 k(rec.geno_begin);
					if (n_helpers > 1)
						k = u64(std::lower_bound(chunk.genos.begin() + std::ptrdiff_t(rec.geno_begin), chunk.genos.begin() + std::ptrdiff_t(rec.geno_end), row_lo,
							[](parsed_genotype const &g, u64 lo) { return g.row < lo; }) - chunk.genos.begin());
					for (; k < rec.geno_end; ++k) {
						auto const &gt(chunk.genos[k]);
						if (gt.row >= row_hi) break;
						u64 edge;
						if (usable(gt, snaps[r], edge) && builder.move_copy(gt.row, snaps[r])) found.push_back(k);
					}
				}
			});
			u64 const n_genos(snaps.empty() ? 0 : chunk.records[snaps.size() - 1].geno_end);
			if (helpers && n_genos >= 32768) helpers->run(apply);
			else {
				// (few genotypes: this thread, as every "helper" in turn -- the row ranges keep the per-copy order either way)
				std::vector<u64> all;
				for (unsigned h(0); h < n_helpers; ++h) { apply(h); all.insert(all.end(), overlaps_found[h].begin(), overlaps_found[h].end()); overlaps_found[h].clear(); }
				overlaps_found[0].swap(all);
			}
			// overlaps in the reference's order: genotypes are stored record by record, copy by copy (variant_graph.cc:379-425)
			std::vector<u64> order;
			for (auto const &found : overlaps_found) order.insert(order.end(), found.begin(), found.end());
			std::sort(order.begin(), order.end());
			std::size_t r(0);
			for (u64 const k : order) {
				while (chunk.records[r].geno_end <= k) ++r;
				auto const &gt(chunk.genos[k]);
				delegate.report_overlapping_alternative(lineno_base + chunk.records[r].line_in_chunk, snaps[r].ref_pos, chunk.records[r].id, vcf_sample_names[row_origin[gt.row].first], row_origin[gt.row].second, gt.alt_number);
			}
		}
		t_genotypes += since(t2);
		if (stopped_at) {
			// the reference stops parsing here (variant_graph.cc:312-313) and still adds the sink node (:437-451); the
			// records of other chromosomes it had passed by then have been counted (:203-207)
			stats.chr_id_mismatches += stopped_at->chr_mismatches_before;
			builder.add_record_node_only(stopped_at->ref_pos);
			builder.finish(ref_seq.size());
			return;
		}
		if (!chunk.error.empty()) bad(lineno_base + chunk.error_line, chunk.error.c_str());
		stats.chr_id_mismatches += chunk.chr_mismatches;
		lineno_base += chunk.n_lines;
		var_idx += chunk.n_data_lines;
		parsed_chunk().records.swap(chunk.records);   // release the chunk's memory
		std::vector<alt_allele>().swap(chunk.alts);
		if (threads > 1) {
			{
				std::lock_guard<std::mutex> lock(mutex);
				consumed = ci + 1;
				chunk.genos.clear();
				spare_genos.emplace_back();
				spare_genos.back().swap(chunk.genos);
			}
			cv_window.notify_all();
		}
		else std::vector<parsed_genotype>().swap(chunk.genos);
	}
	builder.finish(ref_seq.size());                                                      // :437-451
}

} // namespace v2m::host
