#include "graph_file.hh"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <exception>
#include <stdexcept>
#include <thread>
#include <vector>

namespace v2m::host {

namespace {

char const kMagic[8] = {'V', '2', 'M', 'G', 'R', 'A', 'F', '1'};

u64 mix64(u64 z)
{
	z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
	z ^= z >> 27; z *= 0x94D049BB133111EBULL;
	z ^= z >> 31;
	return z;
}

// Sum over the words p[0, n_words) of mix64((first_index + i) ^ word): the file's checksum is the sum over all its words, so
// parts can be summed separately -- large blocks on several threads (32 MiB each, at most 16 threads).
u64 checksum_words(unsigned char const *p, u64 n_words, u64 first_index)
{
	auto const part([p, first_index](u64 begin, u64 end) {
		u64 sum(0);
		for (u64 i(begin); i < end; ++i) { u64 w; std::memcpy(&w, p + 8 * i, 8); sum += mix64((first_index + i) ^ w); }
		return sum;
	});
	u64 const part_words(u64(4) << 20);
	unsigned const n_threads(unsigned(std::min<u64>({(n_words + part_words - 1) / part_words, 16, std::max(1u, std::thread::hardware_concurrency())})));
	if (n_threads <= 1) return part(0, n_words);
	std::atomic<u64> next(0);
	std::vector<u64> sums(n_threads, 0);
	std::vector<std::thread> threads;
	for (unsigned t(0); t < n_threads; ++t)
		threads.emplace_back([&, t] {
			for (u64 first(next.fetch_add(part_words)); first < n_words; first = next.fetch_add(part_words))
				sums[t] += part(first, std::min(n_words, first + part_words));
		});
	for (auto &th : threads) th.join();
	u64 sum(0);
	for (u64 const s : sums) sum += s;
	return sum;
}

struct writer {
	std::FILE *f;
	u64 index{}, sum{};
	void words(void const *p, u64 n_bytes)   // n_bytes is padded to 8 with zeros
	{
		auto const *b(static_cast<unsigned char const *>(p));
		u64 const full(n_bytes / 8);
		sum += checksum_words(b, full, index);
		index += full;
		if (full && 1 != std::fwrite(b, 8 * full, 1, f)) throw std::runtime_error("write error");
		if (n_bytes % 8) {
			u64 w(0);
			std::memcpy(&w, b + 8 * full, n_bytes % 8);
			sum += mix64(index++ ^ w);
			if (1 != std::fwrite(&w, 8, 1, f)) throw std::runtime_error("write error");
		}
	}
};

// Reads with pread() so that a large block (the path matrices: 0.6 GB each at BASELINE config 3) is fetched and
// checksummed by several threads at once; the checksum is a sum over (word index, word), so the parts just add up.
struct reader {
	int fd;
	u64 offset{}, index{}, sum{};

	static void read_exactly(int fd, unsigned char *dst, u64 n, u64 at)
	{
		while (n) {
			ssize_t const got(::pread(fd, dst, n, off_t(at)));
			if (got < 0 && EINTR == errno) continue;
			if (got <= 0) throw std::runtime_error("unexpected end of graph file");
			dst += got; at += u64(got); n -= u64(got);
		}
	}

	static u64 checksum(unsigned char const *p, u64 n_words, u64 first_index)
	{
		u64 sum(0);
		for (u64 i(0); i < n_words; ++i) { u64 w; std::memcpy(&w, p + 8 * i, 8); sum += mix64((first_index + i) ^ w); }
		return sum;
	}

	void words(void *p, u64 n_bytes)
	{
		u64 const full(n_bytes / 8);
		unsigned char *const dst(static_cast<unsigned char *>(p));
		u64 const part_words(u64(4) << 20);                                   // 32 MiB per task
		unsigned const n_threads(unsigned(std::min<u64>({(full + part_words - 1) / part_words, 16, std::max(1u, std::thread::hardware_concurrency())})));
		if (n_threads <= 1) {
			if (full) read_exactly(fd, dst, 8 * full, offset);
			sum += checksum(dst, full, index);
		} else {
			std::atomic<u64> next(0);
			std::vector<u64> sums(n_threads, 0);
			std::vector<std::exception_ptr> errors(n_threads);
			std::vector<std::thread> threads;
			for (unsigned t(0); t < n_threads; ++t)
				threads.emplace_back([&, t] {
					try {
						for (u64 first(next.fetch_add(part_words)); first < full; first = next.fetch_add(part_words)) {
							u64 const n(std::min(part_words, full - first));
							read_exactly(fd, dst + 8 * first, 8 * n, offset + 8 * first);
							sums[t] += checksum(dst + 8 * first, n, index + first);
						}
					} catch (...) { errors[t] = std::current_exception(); }
				});
			for (auto &th : threads) th.join();
			for (auto const &e : errors) if (e) std::rethrow_exception(e);
			for (u64 const part : sums) sum += part;
		}
		offset += 8 * full;
		index += full;
		if (n_bytes % 8) {                                                      // the last word is padded with zeros in the file
			unsigned char tail[8];
			read_exactly(fd, tail, 8, offset);
			sum += checksum(tail, 1, index);
			std::memcpy(dst + 8 * full, tail, n_bytes % 8);
			offset += 8;
			++index;
		}
	}
};

struct file_closer { std::FILE *f; ~file_closer() { if (f) std::fclose(f); } };
struct fd_closer { int fd; ~fd_closer() { if (fd >= 0) ::close(fd); } };

} // namespace


void write_graph(variant_graph const &g, char const *path)
{
	std::FILE *f(std::fopen(path, "wb"));
	if (!f) throw std::runtime_error(std::string("unable to open ") + path + " for writing");
	file_closer closer{f};
	writer w{f};
	std::string names;
	for (auto const &s : g.sample_names) { names += s; names.push_back('\0'); }
	u64 const counts[10] = {
		g.node_count(), g.edge_count(), g.alt_edge_label_bytes.size(), g.sample_names.size(), names.size(), g.ploidy_csum.size(),
		g.paths_by_chrom_copy_and_edge.rows, g.paths_by_chrom_copy_and_edge.cols, g.paths_by_edge_and_chrom_copy.rows, g.paths_by_edge_and_chrom_copy.cols};
	w.words(kMagic, 8);
	w.words(counts, sizeof(counts));
	w.words(g.reference_positions.data(), 8 * g.reference_positions.size());
	w.words(g.aligned_positions.data(), 8 * g.aligned_positions.size());
	w.words(g.alt_edge_targets.data(), 8 * g.alt_edge_targets.size());
	w.words(g.alt_edge_count_csum.data(), 8 * g.alt_edge_count_csum.size());
	w.words(g.alt_edge_label_offsets.data(), 8 * g.alt_edge_label_offsets.size());
	w.words(g.paths_by_chrom_copy_and_edge.words.data(), 8 * g.paths_by_chrom_copy_and_edge.words.size());
	w.words(g.paths_by_edge_and_chrom_copy.words.data(), 8 * g.paths_by_edge_and_chrom_copy.words.size());
	w.words(g.ploidy_csum.data(), 4 * g.ploidy_csum.size());
	w.words(g.alt_edge_label_bytes.data(), g.alt_edge_label_bytes.size());
	w.words(names.data(), names.size());
	u64 const sum(w.sum);
	if (1 != std::fwrite(&sum, 8, 1, f)) throw std::runtime_error("write error");
	closer.f = nullptr;
	if (0 != std::fclose(f)) throw std::runtime_error(std::string("error while closing ") + path);
}


void read_graph(char const *path, variant_graph &g)
{
	int const fd(::open(path, O_RDONLY | O_CLOEXEC));
	if (fd < 0) throw std::runtime_error(std::string("unable to open ") + path);
	fd_closer closer{fd};
	reader r{fd};
	char magic[8];
	r.words(magic, 8);
	if (0 != std::memcmp(magic, kMagic, 8)) throw std::runtime_error(std::string(path) + " is not a V2MGRAF1 graph file");
	u64 c[10];
	r.words(c, sizeof(c));
	u64 const limit(u64(1) << 40);
	for (u64 const v : c) if (v > limit) throw std::runtime_error("implausible counts in graph file");
	if (c[6] % 64 || c[7] % 64 || c[8] % 64 || c[9] % 64) throw std::runtime_error("path matrix dimensions must be multiples of 64");
	// the counts must agree with each other and with the size of the file before anything is allocated from them
	if (c[3] && c[5] != c[3] + 1) throw std::runtime_error("graph file: ploidy_csum must have one entry more than there are samples");
	{
		struct stat st;
		if (0 != ::fstat(fd, &st) || st.st_size < 0) throw std::runtime_error(std::string("unable to size ") + path);
		auto const pad8([](u64 n) { return (n + 7) & ~u64(7); });   // every block is padded to whole 8-byte words
		// (checked arithmetic: each count may be up to 2^40, so the matrix terms can wrap 64 bits -- a header with 2^40 x 2^30 bits
		// would otherwise add up to a small payload, pass, and leave matrices whose word arrays are empty)
		bool wrapped(false);
		auto const mul([&](u64 a, u64 b) { u64 out(0); wrapped |= __builtin_mul_overflow(a, b, &out); return out; });
		u64 payload(0);
		for (u64 const block : {8 * (2 * c[0] + c[1] + (c[0] + 1) + (c[1] + 1)), mul(mul(c[6] / 64, c[7]), 8), mul(mul(c[8] / 64, c[9]), 8), pad8(4 * c[5]), pad8(c[2]), pad8(c[4]), u64(8)})
			wrapped |= __builtin_add_overflow(payload, block, &payload);
		if (wrapped || u64(st.st_size) < r.offset || payload != u64(st.st_size) - r.offset) throw std::runtime_error(std::string(path) + ": the counts in the header do not match the size of the file (truncated or corrupted graph file)");
	}
	g = variant_graph{};
	g.reference_positions.resize(c[0]);
	g.aligned_positions.resize(c[0]);
	g.alt_edge_targets.resize(c[1]);
	g.alt_edge_count_csum.resize(c[0] + 1);
	g.alt_edge_label_offsets.resize(c[1] + 1);
	g.paths_by_chrom_copy_and_edge = bit_matrix::for_overwrite(c[6], c[7]);
	g.paths_by_edge_and_chrom_copy = bit_matrix::for_overwrite(c[8], c[9]);
	if (g.paths_by_chrom_copy_and_edge.words.size() != c[6] / 64 * c[7] || g.paths_by_edge_and_chrom_copy.words.size() != c[8] / 64 * c[9])
		throw std::runtime_error("graph file: path matrix allocation does not match its dimensions");
	g.ploidy_csum.resize(c[5]);
	g.alt_edge_label_bytes.resize(c[2]);
	std::string names(c[4], '\0');
	r.words(g.reference_positions.data(), 8 * c[0]);
	r.words(g.aligned_positions.data(), 8 * c[0]);
	r.words(g.alt_edge_targets.data(), 8 * c[1]);
	r.words(g.alt_edge_count_csum.data(), 8 * (c[0] + 1));
	r.words(g.alt_edge_label_offsets.data(), 8 * (c[1] + 1));
	r.words(g.paths_by_chrom_copy_and_edge.words.data(), 8 * g.paths_by_chrom_copy_and_edge.words.size());
	r.words(g.paths_by_edge_and_chrom_copy.words.data(), 8 * g.paths_by_edge_and_chrom_copy.words.size());
	r.words(g.ploidy_csum.data(), 4 * c[5]);
	r.words(g.alt_edge_label_bytes.data(), c[2]);
	r.words(names.data(), c[4]);
	u64 stored(0);
	reader::read_exactly(fd, reinterpret_cast<unsigned char *>(&stored), 8, r.offset);
	if (stored != r.sum) throw std::runtime_error(std::string(path) + ": checksum mismatch (truncated or corrupted graph file)");
	for (std::size_t pos(0); pos < names.size();) {
		std::size_t const end(names.find('\0', pos));
		if (std::string::npos == end) break;
		g.sample_names.emplace_back(names, pos, end - pos);
		pos = end + 1;
	}
	if (g.sample_names.size() != c[3]) throw std::runtime_error("sample name block does not match the sample count");
	if (g.alt_edge_label_offsets.back() != c[2]) throw std::runtime_error("label offsets do not match the label block");
	// what the readers of these arrays index with (variant_graph::label(), sample_ploidy(), the edge ranges of a node)
	auto const monotone([](auto const &v) { for (std::size_t i(1); i < v.size(); ++i) if (v[i] < v[i - 1]) return false; return true; });
	if (0 != g.alt_edge_label_offsets.front() || !monotone(g.alt_edge_label_offsets)) throw std::runtime_error("graph file: label offsets must start at 0 and not decrease");
	if (!monotone(g.ploidy_csum) || (!g.ploidy_csum.empty() && 0 != g.ploidy_csum.front())) throw std::runtime_error("graph file: ploidy_csum must start at 0 and not decrease");
	if (0 != g.alt_edge_count_csum.front() || !monotone(g.alt_edge_count_csum) || g.alt_edge_count_csum.back() != c[1]) throw std::runtime_error("graph file: alt_edge_count_csum must run from 0 to the edge count");
	for (u64 const t : g.alt_edge_targets) if (t >= c[0]) throw std::runtime_error("graph file: edge target outside the node range");
	// either path matrix may be absent (0 x 0); one that is there must cover every edge and every chromosome copy
	if ((c[6] || c[7]) && (c[6] < c[1] || c[7] < g.total_chromosome_copies())) throw std::runtime_error("graph file: paths_by_chrom_copy_and_edge is smaller than the graph");
	if ((c[8] || c[9]) && (c[9] < c[1] || c[8] < g.total_chromosome_copies())) throw std::runtime_error("graph file: paths_by_edge_and_chrom_copy is smaller than the graph");
}

} // namespace v2m::host
