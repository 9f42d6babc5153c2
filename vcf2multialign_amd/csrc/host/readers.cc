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

// What a worker leaves behind for a chunk.  Besides the records themselves, everything their genotypes amount to (at population
// scale the genotypes ARE the work: config 3 has 360 M of them), so that the one thread that merges chunks in file order never
// looks at a genotype:
//   - the path bits, as the chunk's own slice of paths_by_edge_and_chrom_copy: one column per ALT that becomes an edge (the
//     builder numbers the edges of consecutive records, and of a record's ALTs, consecutively: variant_graph.cc:328-364), in the
//     matrix's own layout, so merging them is one memcpy;
//   - per chromosome copy, its first ALT in the chunk (record, allele) and the reference position its last one reaches
//     (variant_graph.cc:422-423): what the overlap check (:408-418) of the NEXT chunk's first ALT of that copy needs, and what
//     this chunk's needs from the previous ones;
//   - the overlaps that lie inside the chunk (a copy's second, third, ... ALT here), in the reference's order.
struct chunk_overlap { u32 record, row, alt_number; };               // record = index in the chunk

struct parsed_record {
	u64 line_in_chunk;        // 1-based within the chunk
	u64 data_line_in_chunk;   // counts only non-header lines
	u64 ref_pos;
	std::string_view id, ref;
	u32 alt_begin, n_alts;
	u64 first_column;         // columns of the chunk's bit slice before this record's
	u64 overlap_begin;        // overlaps inside the chunk before this record's
	u64 chr_mismatches_before;   // records of other chromosomes seen in the chunk before this one
};

constexpr u32 kNoRecord = UINT32_MAX;

struct parsed_chunk {
	std::vector<parsed_record> records;
	std::vector<alt_allele> alts;
	std::vector<u64> bits;                      // [columns][words per column]
	u64 n_columns{};
	std::vector<u32> first_record, first_alt;   // per chromosome copy (row); kNoRecord: no ALT in this chunk
	std::vector<u64> last_target;               // per row, valid where first_record is
	std::vector<chunk_overlap> overlaps;
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
	u64 n_rows{};               // included chromosome copies
	u64 words_per_column{};     // of paths_by_edge_and_chrom_copy (its rows are padded)
};

struct chunk_error { u64 line; char const *what; };

// "0|0\t" and "0/0\t" as the four bytes a little-endian load sees
constexpr std::uint32_t kRefRefPhased = 0x09307C30u, kRefRefUnphased = 0x09302F30u;
static_assert(__BYTE_ORDER__ == __ORDER_LITTLE_ENDIAN__, "the column fast path compares little-endian words");

// Parses the lines of text[begin, end) (whole lines) into `out`.
void parse_chunk(std::string_view text, parse_context const &ctx, parsed_chunk &out)
{
	std::size_t pos(0);
	out.bits.clear();
	out.n_columns = 0;
	out.overlaps.clear();
	out.first_record.assign(ctx.n_rows, kNoRecord);
	out.first_alt.assign(ctx.n_rows, 0);
	out.last_target.assign(ctx.n_rows, 0);
	std::vector<u64> alt_column;                 // of the record at hand: ALT -> column of the chunk's slice, or UINT64_MAX (no edge)
	u64 columns_before_record(0);
	std::size_t overlaps_before_record(0);
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
			// the ALTs that become edges get the chunk's next columns, in ALT order (variant_graph.cc:328-364)
			rec.first_column = columns_before_record = out.n_columns;
			rec.overlap_begin = overlaps_before_record = out.overlaps.size();
			alt_column.assign(rec.n_alts, UINT64_MAX);
			for (u32 a(0); a < rec.n_alts; ++a) if (alt_kind::unhandled != out.alts[rec.alt_begin + a].kind) alt_column[a] = out.n_columns++;
			out.bits.resize(out.n_columns * ctx.words_per_column, 0);
			u32 const rec_index(u32(out.records.size()));
			u64 const target_ref_pos(rec.ref_pos + rec.ref.size());                       // :333

			// genotypes of the included copies (:379-425); allele 0 and '.' change nothing (:393-397)
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
									u64 const column(alt_column[allele - 1]);
									if (UINT64_MAX != column) {                                   // (an ALT without an edge: :401-403)
										out.bits[column * ctx.words_per_column + (u32(row) >> 6)] |= u64(1) << (u32(row) & 63);   // :424
										if (kNoRecord == out.first_record[row]) { out.first_record[row] = rec_index; out.first_alt[row] = allele; }
										else if (rec.ref_pos < out.last_target[row]) out.overlaps.push_back({rec_index, u32(row), allele});   // :408-418
										out.last_target[row] = target_ref_pos;                    // :422-423
									}
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
			out.records.push_back(rec);
		}
	} catch (chunk_error const &e) {
		out.error = e.what;
		out.error_line = e.line;
		// what the failing record had claimed before it failed (the records before it are still merged; the per-copy state it may
		// have touched no longer matters: the error ends the build right after them)
		out.n_columns = columns_before_record;
		out.overlaps.resize(overlaps_before_record);
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
			ctx.n_rows = row;
			ctx.words_per_column = graph.paths_by_edge_and_chrom_copy.words_per_column();
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
	std::vector<parsed_chunk> spare_chunks;                           // (under `mutex`) consumed chunks: their vectors' pages are there already

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
				// (the vectors of a chunk that has been merged: their pages are there already; fresh ones of this size come from
				// mmap every time and cost a fault per page)
				std::lock_guard<std::mutex> lock(mutex);
				if (!spare_chunks.empty()) { chunks[idx] = std::move(spare_chunks.back()); spare_chunks.pop_back(); }
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

	// The merge stage.  Per chunk, in file order: the records go into the builder one by one -- nodes, edges, targets: cheap --,
	// the chunk's slice of the path matrix is copied into place, and every chromosome copy's first ALT of the chunk is checked
	// against where the copy's last ALT before the chunk reached (the overlap check, variant_graph.cc:408-418, across the chunk
	// boundary; inside the chunk the parser has made it).  Overlaps are reported in the reference's order: record by record, copy by copy.
	std::vector<u64> reaches(ctx.n_rows, 0);                          // per copy: target_ref_positions_by_chrom_copy (:422-423)
	std::vector<chunk_overlap> reported;

	// V2M_READER_TIMING=1: where the merge stage's time went (waiting for parsed chunks / records into the builder / bits and overlaps), to stderr
	bool const timing(nullptr != std::getenv("V2M_READER_TIMING"));
	double t_wait(0), t_records(0), t_genotypes(0);
	auto const now([] { return std::chrono::steady_clock::now(); });
	auto const since([&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(now() - t).count(); });
	struct report_timing {
		bool on; double const &w, &r, &g;
		~report_timing() { if (on) std::fprintf(stderr, "[vcf reader] merge stage: %.3f s waiting for parsed chunks, %.3f s records into the builder, %.3f s path bits and overlaps\n", w, r, g); }
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
		// records parsed before an error are still merged first, so errors surface in file order
		u64 const first_edge(graph.edge_count());
		std::size_t n_merged(0);
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
			++n_merged;
		}
		t_records += since(t1);
		auto const t2(now());
		{
			// the merged records' columns of the chunk's slice (all of them unless the build stops inside the chunk)
			u64 const n_columns(n_merged < chunk.records.size() ? chunk.records[n_merged].first_column : chunk.n_columns);
			if (graph.edge_count() - first_edge != n_columns) throw std::logic_error("VCF reader: the builder made other edges than the parser counted");
			auto &m(graph.paths_by_edge_and_chrom_copy);
			// (no chromosome copy included -- every sample excluded: the matrix has no rows, graph_builder grows no columns for it,
			// and there is nothing to copy; the edge count above is still checked)
			if (n_columns && ctx.words_per_column) {
				if (m.cols < first_edge + n_columns || m.words_per_column() != ctx.words_per_column) throw std::logic_error("VCF reader: the path matrix is not what the parser filled its slice for");
				std::memcpy(m.words.data() + first_edge * ctx.words_per_column, chunk.bits.data(), n_columns * ctx.words_per_column * sizeof(u64));
			}
			// overlaps: the parser's (inside the chunk) and, here, every copy's first ALT of the chunk against the chunks before
			u64 const n_inside(n_merged < chunk.records.size() ? chunk.records[n_merged].overlap_begin : chunk.overlaps.size());
			reported.assign(chunk.overlaps.begin(), chunk.overlaps.begin() + std::ptrdiff_t(n_inside));
			for (u64 row(0); row < ctx.n_rows; ++row) {
				u32 const r(chunk.first_record[row]);
				if (kNoRecord == r || r >= n_merged) continue;
				if (chunk.records[r].ref_pos < reaches[row]) reported.push_back({r, u32(row), chunk.first_alt[row]});
				reaches[row] = chunk.last_target[row];
			}
			if (reported.size() > n_inside) {
				auto const before([](chunk_overlap const &a, chunk_overlap const &b) { return a.record != b.record ? a.record < b.record : a.row < b.row; });
				std::sort(reported.begin() + std::ptrdiff_t(n_inside), reported.end(), before);
				std::inplace_merge(reported.begin(), reported.begin() + std::ptrdiff_t(n_inside), reported.end(), [](chunk_overlap const &a, chunk_overlap const &b) { return a.record != b.record ? a.record < b.record : a.row < b.row; });
			}
			for (auto const &o : reported)
				delegate.report_overlapping_alternative(lineno_base + chunk.records[o.record].line_in_chunk, chunk.records[o.record].ref_pos, chunk.records[o.record].id,
					vcf_sample_names[row_origin[o.row].first], row_origin[o.row].second, o.alt_number);
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
		if (threads > 1) {
			{
				std::lock_guard<std::mutex> lock(mutex);
				consumed = ci + 1;
				chunk.records.clear(); chunk.alts.clear(); chunk.n_lines = chunk.n_data_lines = chunk.chr_mismatches = 0;
				spare_chunks.emplace_back(std::move(chunk));
				chunk = parsed_chunk{};
			}
			cv_window.notify_all();
		}
		else chunk = parsed_chunk{};
	}
	builder.finish(ref_seq.size());                                                      // :437-451
}

} // namespace v2m::host
