// Sanitizer harness for the multi-context writers of csrc/host/output.cc (the sharded pwritev writer, the turnstile for ordered
// outputs, the unordered per-context writers of --output-sequences-separate, --pipe) and for the worker pool of the bench's
// checksumming sink (csrc/synth/sink.cc).
//
// output.cc reaches the GPU through five entry points of include/v2m_hip.h -- v2m_splice_rows, v2m_splice_rows_held +
// v2m_row_release (rows a pool of writers keeps for a while), v2m_aligned_length, v2m_last_error (gpu_context::check) -- and
// through gpu_context's constructor / destructor.  This file is a CPU-only MOCK of exactly those: a "context" synthesises its rows on the calling thread, a few rows per slice, in a heap buffer that is freed as
// soon as the slice's sink calls have returned (a sink that kept a pointer is a use-after-free under ASan, as it would be a stale
// pinned slot in the library), with slice sizes that do not divide the contexts' blocks, so turns change in the middle of slices.
// Nothing here is product code and nothing of it is linked into the product.
// Built and run by tools/sanitize_host.sh with -fsanitize=address,undefined and -fsanitize=thread.
#include "output.hh"

#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <sstream>
#include <streambuf>
#include <thread>

#include <sys/stat.h>
#include <unistd.h>

using namespace v2m::host;

// ---- the mock -----------------------------------------------------------------------------------------------------------------
struct v2m_ctx {
	u64 aligned_len{};
	u64 slice_rows{3};
	long fail_after_rows{-1};                       // >= 0: v2m_splice_rows returns V2M_ERR_HIP after this many rows
	std::function<u32(u32)> to_global;              // local copy index -> chromosome copy (shards / interleave)
	std::string err;
	unsigned delay_us{};                            // per row: lets the contexts' threads interleave differently
};

namespace {
	char base_of(u32 copy, u64 j) { return "ACGT-"[(copy * 2654435761u + j * 40503u + (j >> 3)) % 5]; }

	u64 row_length(v2m_ctx const &c, u32 copy, bool unaligned) { return unaligned ? c.aligned_len - (V2M_PLOIDY_MAX == copy ? 7 : copy % 5) : c.aligned_len; }

	std::string row_body(v2m_ctx const &c, u32 copy, bool unaligned)
	{
		std::string s(row_length(c, copy, unaligned), '?');
		for (u64 j(0); j < s.size(); ++j) s[j] = base_of(copy, j);
		return s;
	}
}

extern "C" {
uint64_t v2m_aligned_length(const v2m_ctx *ctx) { return ctx->aligned_len; }
const char *v2m_last_error(const v2m_ctx *ctx) { return ctx ? ctx->err.c_str() : "no context"; }

int v2m_splice_rows(v2m_ctx *ctx, const v2m_row_batch *rows, uint32_t flags, v2m_sink_fn sink, void *user)
{
	bool const unaligned(flags & V2M_SPLICE_UNALIGNED);
	u64 const pitch(ctx->aligned_len + 16);
	for (u64 r0(0); r0 < rows->n_rows; r0 += ctx->slice_rows) {
		u64 const n(std::min<u64>(ctx->slice_rows, rows->n_rows - r0));
		char *const slot(new char[n * pitch]);       // one "pinned slot": gone after the slice's last sink call
		struct freer { char *p; ~freer() { delete[] p; } } const free_slot{slot};
		std::vector<u64> lengths(n);
		for (u64 i(0); i < n; ++i) {
			u32 const local(rows->copy_index[r0 + i]);
			u32 const copy(V2M_PLOIDY_MAX == local ? local : ctx->to_global ? ctx->to_global(local) : local);
			std::string const body(row_body(*ctx, copy, unaligned));
			std::memcpy(slot + i * pitch, body.data(), body.size());
			lengths[i] = body.size();
		}
		for (u64 i(0); i < n; ++i) {
			if (ctx->fail_after_rows >= 0 && long(r0 + i) >= ctx->fail_after_rows) { ctx->err = "mock: device lost"; return V2M_ERR_HIP; }
			if (ctx->delay_us) std::this_thread::sleep_for(std::chrono::microseconds(ctx->delay_us));
			if (0 != sink(user, r0 + i, slot + i * pitch, lengths[i])) { ctx->err = "sink callback failed"; return V2M_ERR_SINK; }
		}
	}
	return V2M_OK;
}
}

// The held form: n_slots heap "pinned slots" in a ring; a slot is freed (so that a writer still reading it is a use-after-free under ASan)
// and refilled only when every accepted row of it has been released -- by whatever thread, which is what TSan watches.
struct v2m_row_hold {
	std::mutex *mutex; std::condition_variable *released;
	u64 outstanding{};
	char *bytes{};
};

extern "C" {
void v2m_row_release(v2m_row_hold *hold)
{
	// (notified under the lock: the waiter may free the slot ring the moment it sees zero, so nothing of it is touched after the unlock)
	std::lock_guard<std::mutex> const lock(*hold->mutex);
	if (0 == --hold->outstanding) hold->released->notify_all();
}

int v2m_splice_rows_held(v2m_ctx *ctx, const v2m_row_batch *rows, uint32_t flags, uint32_t n_slots, v2m_hold_sink_fn sink, void *user)
{
	bool const unaligned(flags & V2M_SPLICE_UNALIGNED);
	u64 const pitch(ctx->aligned_len + 16);
	std::mutex mutex;
	std::condition_variable released;
	std::vector<v2m_row_hold> slots(n_slots);
	for (auto &s : slots) { s.mutex = &mutex; s.released = &released; }
	auto const wait_free([&](v2m_row_hold &s) {
		std::unique_lock<std::mutex> lock(mutex);
		released.wait(lock, [&] { return 0 == s.outstanding; });
		delete[] s.bytes;
		s.bytes = nullptr;
	});
	int rc(V2M_OK);
	for (u64 r0(0), slice(0); r0 < rows->n_rows && V2M_OK == rc; r0 += ctx->slice_rows, ++slice) {
		u64 const n(std::min<u64>(ctx->slice_rows, rows->n_rows - r0));
		auto &slot(slots[slice % n_slots]);
		wait_free(slot);
		slot.bytes = new char[n * pitch];
		std::vector<u64> lengths(n);
		for (u64 i(0); i < n; ++i) {
			u32 const local(rows->copy_index[r0 + i]);
			u32 const copy(V2M_PLOIDY_MAX == local ? local : ctx->to_global ? ctx->to_global(local) : local);
			std::string const body(row_body(*ctx, copy, unaligned));
			std::memcpy(slot.bytes + i * pitch, body.data(), body.size());
			lengths[i] = body.size();
		}
		{ std::lock_guard<std::mutex> const lock(mutex); slot.outstanding = n; }
		for (u64 i(0); i < n; ++i) {
			bool const fail(ctx->fail_after_rows >= 0 && long(r0 + i) >= ctx->fail_after_rows);
			if (ctx->delay_us) std::this_thread::sleep_for(std::chrono::microseconds(ctx->delay_us));
			if (!fail && 0 == sink(user, r0 + i, slot.bytes + i * pitch, lengths[i], &slot)) continue;
			{ std::lock_guard<std::mutex> const lock(mutex); slot.outstanding -= n - i; }
			ctx->err = fail ? "mock: device lost" : "sink callback failed";
			rc = fail ? V2M_ERR_HIP : V2M_ERR_SINK;
			break;
		}
	}
	for (auto &s : slots) wait_free(s);                              // no row in a writer's hands when the call returns
	return rc;
}
}

// gpu_context's out-of-line members (csrc/host/gpu_path.cc in the product, which pulls in the rest of the ABI)
gpu_context::gpu_context(int) { std::abort(); }
gpu_context::~gpu_context() {}
void gpu_context::check(int rc) const { if (V2M_OK != rc) throw gpu_error(rc, v2m_last_error(m_ctx)); }


// ---- expectations ---------------------------------------------------------------------------------------------------------------
namespace {
	int failures(0);
	void expect(bool ok, char const *what) { if (!ok) { ++failures; std::fprintf(stderr, "FAILED: %s\n", what); } }

	variant_graph make_graph(u32 samples)
	{
		variant_graph g;
		g.sample_names.resize(samples);
		g.ploidy_csum.resize(samples + 1);
		for (u32 i(0); i < samples; ++i) g.sample_names[i] = "S" + std::to_string(i);
		for (u32 i(0); i <= samples; ++i) g.ploidy_csum[i] = 2 * i;
		return g;
	}

	std::string expected_a2m(v2m_ctx const &c, variant_graph const &g, bool unaligned, char const *chr = nullptr)
	{
		std::string const pre(chr ? std::string(chr) + "\t" : std::string());
		std::string out(">" + pre + "REF\n" + row_body(c, V2M_PLOIDY_MAX, unaligned) + "\n");
		for (u32 s(0); s < g.sample_names.size(); ++s)
			for (u32 k(0); k < 2; ++k)
				out += ">" + pre + g.sample_names[s] + "-" + std::to_string(1 + k) + "\n" + row_body(c, 2 * s + k, unaligned) + "\n";
		return out;
	}

	std::string slurp(std::string const &path)
	{
		std::ifstream f(path, std::ios::binary);
		std::stringstream ss;
		ss << f.rdbuf();
		return ss.str();
	}

	// an ostream target that goes bad after `limit` bytes
	struct failing_buf final : std::streambuf {
		std::size_t limit, seen{};
		explicit failing_buf(std::size_t l) : limit(l) {}
		int_type overflow(int_type ch) override { return ++seen > limit ? traits_type::eof() : traits_type::not_eof(ch); }
		std::streamsize xsputn(char const *, std::streamsize n) override { seen += std::size_t(n); return seen > limit ? 0 : n; }
	};

	struct counting_delegate final : output_delegate {
		std::atomic<u32> samples{}, handled{}, max_seen{};
		void will_handle_sample(std::string const &, u32, u32) override { ++samples; }
		void will_handle_founder_sequence(u32) override {}
		void handled_sequences(u32 n) override { ++handled; u32 m(max_seen.load()); while (n > m && !max_seen.compare_exchange_weak(m, n)) {} }
	};

	struct rig {
		std::vector<v2m_ctx> ctx;
		std::vector<std::unique_ptr<gpu_context>> gpus;
		rig(std::size_t n, u64 L) : ctx(n)
		{
			for (std::size_t k(0); k < n; ++k) {
				ctx[k].aligned_len = L;
				ctx[k].slice_rows = 2 + k;                   // 2, 3, 4 rows per slice: never a divisor of the 8-copy blocks + REF
				ctx[k].delay_us = unsigned(37 * (k + 1) % 50);
				gpus.emplace_back(new gpu_context(&ctx[k], gpu_context::borrowed{}));
			}
		}
		void attach(output &o) { for (std::size_t k(1); k < gpus.size(); ++k) o.add_gpu(*gpus[k]); }
		void shard(output &o, u32 n_copies)
		{
			std::vector<copy_shard> shards;
			for (u32 k(0); k < gpus.size(); ++k) {
				shards.push_back(shard_copies_mock(n_copies, u32(gpus.size()), k));
				u64 const first(shards.back().first);
				ctx[k].to_global = [first](u32 l) { return u32(first + l); };
			}
			o.set_copy_shards(shards);
		}
		void interleave(output &o)
		{
			copy_interleave deal;
			deal.block = 8;
			deal.world = u32(gpus.size());
			for (u32 k(0); k < gpus.size(); ++k)
				ctx[k].to_global = [deal, k](u32 l) { return u32((l / deal.block) * deal.block * deal.world + k * deal.block + l % deal.block); };
			o.set_copy_interleave(deal);
		}
		static copy_shard shard_copies_mock(u64 n_copies, u32 world, u32 rank)     // gpu_path.cc:shard_copies (kept in step by tests/test_host_cpu.py for the product's)
		{
			u64 const granule(8), n_blocks((n_copies + granule - 1) / granule), base(n_blocks / world), extra(n_blocks % world), first_heavy(world - extra);
			u64 const b0(rank * base + (rank > first_heavy ? rank - first_heavy : 0)), b1(b0 + base + (rank >= first_heavy ? 1 : 0));
			return {std::min(n_copies, granule * b0), std::min(n_copies, granule * b1)};
		}
	};

	template <typename F> bool throws(F &&f, char const *needle = nullptr)
	{
		try { f(); }
		catch (std::exception const &e) { return !needle || nullptr != std::strstr(e.what(), needle); }
		return false;
	}
}


// ---- csrc/synth/sink.cc -----------------------------------------------------------------------------------------------------------
extern "C" {
void *v2ms_checksum_sink_create(uint64_t capacity_rows, uint32_t threads);
void v2ms_checksum_sink_destroy(void *);
int v2ms_checksum_sink_fn(void *, uint64_t, char const *, uint64_t);
uint64_t v2ms_checksum_sink_rows(void *);
uint64_t v2ms_checksum_sink_bytes(void *);
uint64_t const *v2ms_checksum_sink_checksums(void *);
uint64_t const *v2ms_checksum_sink_lengths(void *);
void v2ms_checksum_sink_force_scalar(void *);
}

namespace {
	u64 mix64(u64 z) { z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL; z ^= z >> 27; z *= 0x94D049BB133111EBULL; z ^= z >> 31; return z; }
	u64 reference_checksum(std::string const &row)   // include/v2m_hip.h: v2m_checksum_rows_device's formula, byte by byte
	{
		u64 acc(0);
		for (u64 k(0); 8 * k < row.size(); ++k) {
			u64 w(0);
			std::memcpy(&w, row.data() + 8 * k, std::min<u64>(8, row.size() - 8 * k));
			acc += mix64((k + 1) * 0x9E3779B97F4A7C15ULL ^ w);
		}
		return acc + mix64(row.size());
	}

	void sink_pool_checks()
	{
		// rows shorter than a word (no part to hand out: the divide-by-zero of commit 24b531c), around word and vector boundaries, and long ones
		std::vector<u64> const lengths{0, 1, 7, 8, 9, 15, 16, 17, 63, 64, 65, 71, 511, 512, 513, 4099, 100003, 1u << 20, 3, 0, 1000001};
		v2m_ctx c;
		for (u32 threads : {1u, 2u, 5u, 16u}) {
			for (int scalar(0); scalar < 2; ++scalar) {
				void *const s(v2ms_checksum_sink_create(lengths.size(), threads));
				if (scalar) v2ms_checksum_sink_force_scalar(s);
				std::vector<u64> want;
				u64 total(0);
				for (u64 r(0); r < lengths.size(); ++r) {
					c.aligned_len = lengths[r];
					std::string const row(row_body(c, u32(r), false));
					char *const slot(new char[row.size() + 1]);          // valid during the call only
					std::memcpy(slot, row.data(), row.size());
					expect(0 == v2ms_checksum_sink_fn(s, r, slot, row.size()), "sink accepts the row");
					delete[] slot;
					want.push_back(reference_checksum(row));
					total += row.size();
				}
				expect(1 == v2ms_checksum_sink_fn(s, lengths.size(), "x", 1), "a row beyond the capacity is refused");
				expect(v2ms_checksum_sink_rows(s) == lengths.size() && v2ms_checksum_sink_bytes(s) == total, "sink counts rows and bytes");
				for (u64 r(0); r < lengths.size(); ++r)
					expect(v2ms_checksum_sink_checksums(s)[r] == want[r] && v2ms_checksum_sink_lengths(s)[r] == lengths[r], "sink checksum equals the formula");
				v2ms_checksum_sink_destroy(s);
			}
		}
		// a sink that is destroyed without ever seeing a row, and one destroyed right after a row (workers still parked / just woken)
		v2ms_checksum_sink_destroy(v2ms_checksum_sink_create(1, 8));
		void *const s(v2ms_checksum_sink_create(1, 8));
		std::string const row(4096, 'A');
		v2ms_checksum_sink_fn(s, 0, row.data(), row.size());
		v2ms_checksum_sink_destroy(s);
	}
}


int main(int argc, char **argv)
{
	std::signal(SIGPIPE, SIG_IGN);                       // as the command-line driver does: a reader that goes away is an error code, not a signal
	std::string const dir(argc > 1 ? argv[1] : "/tmp");
	u32 const samples(37);                               // 74 copies: 10 blocks of 8 (the last one ragged) over 3 contexts -> 3 / 3 / 4 blocks
	u64 const L(2500);
	variant_graph const g(make_graph(samples));
	counting_delegate counted;

	// (A) one aligned A2M file written by three contexts' threads at final offsets (write_a2m_sharded), sharded matrix and whole matrix
	for (int sharded(0); sharded < 2; ++sharded) {
		rig r(3, L);
		haplotype_output out(*r.gpus[0], nullptr, "chr1", true, false, counted);
		r.attach(out);
		if (sharded) r.shard(out, 2 * samples);
		std::string const path(dir + "/sharded.a2m");
		out.output_a2m(g, path.c_str());
		expect(slurp(path) == expected_a2m(r.ctx[0], g, false, "chr1"), "sharded pwritev writer: the file is the sequential one");
	}
	expect(counted.max_seen == 2 * samples + 1, "delegate saw every sequence handled");

	// (B) ordered outputs through the turnstile: a std::ostream target, aligned and unaligned, copies dealt round-robin in blocks of 8
	for (int unaligned(0); unaligned < 2; ++unaligned) {
		for (std::size_t n_ctx : {2u, 3u, 5u}) {
			rig r(n_ctx, L);
			haplotype_output out(*r.gpus[0], nullptr, nullptr, true, 0 != unaligned, counted);
			r.attach(out);
			r.interleave(out);
			std::ostringstream os;
			out.output_a2m(g, os);
			expect(os.str() == expected_a2m(r.ctx[0], g, 0 != unaligned), "turnstile: rows leave in row order");
		}
	}

	// (C) a file per sequence: the rows stay in the contexts' slots while a pool of writers (1, 3 and 8 threads) writes them out
	for (unsigned writers : {1u, 3u, 8u}) for (std::size_t n_ctx : {1u, 3u}) {
		::setenv("V2M_WRITER_THREADS", std::to_string(writers).c_str(), 1);
		::setenv("V2M_HELD_SLOTS", writers == 3 ? "2" : "5", 1);
		rig r(n_ctx, L);
		haplotype_output out(*r.gpus[0], nullptr, nullptr, true, false, counted);
		r.attach(out);
		if (n_ctx > 1) r.interleave(out);
		std::string const sub(dir + "/separate");
		::mkdir(sub.c_str(), 0755);
		char cwd[4096];
		expect(nullptr != ::getcwd(cwd, sizeof(cwd)) && 0 == ::chdir(sub.c_str()), "chdir");
		out.output_separate(g, true);
		expect(slurp("REF.a2m") == ">REF.a2m\n" + row_body(r.ctx[0], V2M_PLOIDY_MAX, false), "separate: REF file");
		bool all(true);
		for (u32 s(0); s < samples; ++s)
			for (u32 k(0); k < 2; ++k) {
				std::string const name("S" + std::to_string(s) + "." + std::to_string(1 + k) + ".a2m");
				all = all && slurp(name) == ">" + name + "\n" + row_body(r.ctx[0], 2 * s + k, false);
			}
		expect(all, "separate: every sequence's file, written by the pool");
		for (u32 q(0); q < samples; ++q) for (u32 k(0); k < 2; ++k) ::unlink(("S" + std::to_string(q) + "." + std::to_string(1 + k) + ".a2m").c_str());
		expect(0 == ::chdir(cwd), "chdir back");
	}

	// (D) a sink that fails in the middle of the run: ordered (stream goes bad), unordered (a file that cannot be created), sharded (/dev/full)
	{
		rig r(3, L);
		haplotype_output out(*r.gpus[0], nullptr, nullptr, true, false, counted);
		r.attach(out);
		r.interleave(out);
		failing_buf buf(20 * (L + 8));
		std::ostream os(&buf);
		expect(throws([&] { out.output_a2m(g, os); }, "sink"), "turnstile: a stream that goes bad ends every context's call");
	}
	{
		rig r(3, L);
		haplotype_output out(*r.gpus[0], nullptr, nullptr, true, false, counted);
		r.attach(out);
		r.interleave(out);
		char cwd[4096];
		expect(nullptr != ::getcwd(cwd, sizeof(cwd)) && 0 == ::chdir("/proc"), "chdir /proc");      // no file can be created there
		expect(throws([&] { out.output_separate(g, true); }), "separate: files that cannot be created end the run");
		expect(0 == ::chdir(cwd), "chdir back");
	}
	{
		rig r(3, L);
		haplotype_output out(*r.gpus[0], nullptr, nullptr, true, false, counted);
		r.attach(out);
		r.shard(out, 2 * samples);
		expect(throws([&] { out.output_a2m(g, "/dev/full"); }, "sink"), "sharded writer: ENOSPC ends the run");
	}

	// (E) a context that fails in the middle (the library returns an error on one GPU): nobody waits for its rows for ever
	for (int ordered(0); ordered < 2; ++ordered) {
		rig r(3, L);
		r.ctx[1].fail_after_rows = 5;
		haplotype_output out(*r.gpus[0], nullptr, nullptr, true, false, counted);
		r.attach(out);
		r.interleave(out);
		if (ordered) {
			std::ostringstream os;
			expect(throws([&] { out.output_a2m(g, os); }, "device lost"), "turnstile: the failing context's error is the one reported");
		} else {
			char cwd[4096];
			std::string const sub(dir + "/separate");
			expect(nullptr != ::getcwd(cwd, sizeof(cwd)) && 0 == ::chdir(sub.c_str()), "chdir");
			expect(throws([&] { out.output_separate(g, true); }, "device lost"), "unordered: the failing context's error is the one reported");
			expect(0 == ::chdir(cwd), "chdir back");
		}
	}
	{
		rig r(3, L);
		r.ctx[2].fail_after_rows = 0;
		haplotype_output out(*r.gpus[0], nullptr, nullptr, true, false, counted);
		r.attach(out);
		r.shard(out, 2 * samples);
		std::string const path(dir + "/sharded_fail.a2m");
		expect(throws([&] { out.output_a2m(g, path.c_str()); }, "device lost"), "sharded writer: a failing context's error is reported after all threads have joined");
	}

	// (F) --pipe: a reader that exits without reading (EPIPE, SIGPIPE ignored), one that fails, one that does not exist, one that reads everything
	{
		rig r(3, L);
		haplotype_output out(*r.gpus[0], "true", nullptr, true, false, counted);
		r.attach(out);
		r.interleave(out);
		expect(throws([&] { out.output_a2m(g, "unused-name"); }), "pipe: a reader that exits early is an error");
	}
	{
		rig r(2, L);
		haplotype_output out(*r.gpus[0], "false", nullptr, true, false, counted);
		r.attach(out);
		r.interleave(out);
		expect(throws([&] { out.output_separate(g, true); }, "exited with status"), "pipe per sequence: a failing subprocess is the error reported");
	}
	{
		rig r(1, L);
		haplotype_output out(*r.gpus[0], "/no/such/command", nullptr, true, false, counted);
		expect(throws([&] { out.output_a2m(g, "x"); }, "Unable to execute subprocess"), "pipe: a command that does not exist");
	}
	{
		rig r(3, L);
		std::string const script(dir + "/reader.sh"), result(dir + "/piped.a2m");
		{ std::ofstream f(script); f << "#!/bin/sh\ncat > " << result << "\n"; }
		::chmod(script.c_str(), 0755);
		haplotype_output out(*r.gpus[0], script.c_str(), nullptr, true, true, counted);
		r.attach(out);
		r.interleave(out);
		out.output_a2m(g, "piped");
		expect(slurp(result) == expected_a2m(r.ctx[0], g, true), "pipe: the reader gets the whole unaligned A2M in row order");
	}

	// (G) the bench's checksumming sink
	sink_pool_checks();

	if (failures) { std::fprintf(stderr, "output_harness: %d check(s) failed\n", failures); return 1; }
	std::printf("output_harness: sharded writer, turnstile (ordered / unordered), failing sinks, failing contexts, pipes and the checksum sink's pool: ok\n");
	return 0;
}
