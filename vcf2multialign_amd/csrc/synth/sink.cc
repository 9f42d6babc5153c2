// sink.cc -- the checksumming C sink of bench.py's end-to-end leg (host only; part of libv2m_synth.so, not of the drop-in boundary).
//
// Discards every row AFTER reading all of it: the checksum of v2m_checksum_rows_device (include/v2m_hip.h) of every row body as it
// arrives in the library's pinned slot, so that what crossed the link can be compared with the CPU oracle's rows.  A row has to be
// consumed before the sink returns (the slot is the library's) and the link delivers 56 GB/s, so the row is cut into parts for a small
// pool of threads that live as long as the sink (the parts' sums add up: the checksum is a sum over (index, word)).
//
// A GPU box gives a job 16 cores' worth of CPU time (a cgroup quota: threads beyond it get the whole process throttled), and the scalar
// loop needs 12 of them for the link's rate (7 GB/s per core: two 64-bit multiplies per word).  Where the CPU has AVX-512DQ the words
// go through vpmullq eight at a time (chosen at run time), which leaves most of the quota to the process's other threads.

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#if !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#endif

namespace {

inline uint64_t mix64(uint64_t z) { z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL; z ^= z >> 27; z *= 0x94D049BB133111EBULL; z ^= z >> 31; return z; }

constexpr uint64_t kGolden = 0x9E3779B97F4A7C15ULL;

// sum over k in [k0, k1) of mix64((k + 1) * kGolden ^ word k)
uint64_t sum_words_scalar(char const *bytes, uint64_t k0, uint64_t k1)
{
	uint64_t acc(0);
	for (uint64_t k(k0); k < k1; ++k) {
		uint64_t w;
		std::memcpy(&w, bytes + 8 * k, 8);
		acc += mix64((k + 1) * kGolden ^ w);
	}
	return acc;
}

typedef uint64_t (*sum_words_fn)(char const *, uint64_t, uint64_t);

#if !defined(__HIP_DEVICE_COMPILE__)      // (hipcc parses every input for gfx950 as well; there are no x86 builtins on that side)
__attribute__((target("avx512f,avx512dq"))) uint64_t sum_words_avx512(char const *bytes, uint64_t k0, uint64_t k1)
{
	uint64_t k(k0);
	__m512i acc(_mm512_setzero_si512());
	__m512i index_term(_mm512_set_epi64((long long) ((k0 + 8) * kGolden), (long long) ((k0 + 7) * kGolden), (long long) ((k0 + 6) * kGolden), (long long) ((k0 + 5) * kGolden),
		(long long) ((k0 + 4) * kGolden), (long long) ((k0 + 3) * kGolden), (long long) ((k0 + 2) * kGolden), (long long) ((k0 + 1) * kGolden)));
	__m512i const step(_mm512_set1_epi64((long long) (8 * kGolden)));
	__m512i const c1(_mm512_set1_epi64((long long) 0xBF58476D1CE4E5B9ULL)), c2(_mm512_set1_epi64((long long) 0x94D049BB133111EBULL));
	for (; k + 8 <= k1; k += 8) {
		__m512i z(_mm512_xor_si512(index_term, _mm512_loadu_si512(bytes + 8 * k)));
		z = _mm512_xor_si512(z, _mm512_srli_epi64(z, 30));
		z = _mm512_mullo_epi64(z, c1);
		z = _mm512_xor_si512(z, _mm512_srli_epi64(z, 27));
		z = _mm512_mullo_epi64(z, c2);
		z = _mm512_xor_si512(z, _mm512_srli_epi64(z, 31));
		acc = _mm512_add_epi64(acc, z);
		index_term = _mm512_add_epi64(index_term, step);
	}
	// (lanes added as unsigned words: _mm512_reduce_add_epi64 adds them as signed long long, which is an overflow UBSan reports)
	alignas(64) uint64_t lanes[8];
	_mm512_store_si512(lanes, acc);
	uint64_t total(sum_words_scalar(bytes, k, k1));
	for (uint64_t lane : lanes) total += lane;
	return total;
}

sum_words_fn pick_sum_words()
{
	__builtin_cpu_init();
	return (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq")) ? sum_words_avx512 : sum_words_scalar;
}
#else
sum_words_fn pick_sum_words() { return sum_words_scalar; }
#endif

} // namespace


extern "C" {

struct v2ms_checksum_sink {
	std::vector <uint64_t> checksums, lengths;
	uint64_t rows{}, bytes{};
	std::vector <std::thread> workers;
	std::mutex mutex;
	std::condition_variable wake, idle;
	// the row being summed; written by the sink only under `mutex` and only while no worker is active, read by a worker under
	// `mutex` when it takes notice of a new generation (a worker that wakes late finds either the finished row -- no part left --
	// or the next one, never a mixture)
	struct job { char const *bytes{}; uint64_t words{}, parts{}; } current;
	uint64_t generation{};
	bool stopping{};
	unsigned active{};                   // workers between noticing a generation and having finished with it
	uint64_t parts_done{};
	std::atomic <uint64_t> next_part{}, sum{};

	sum_words_fn sum_words{pick_sum_words()};
	char const *flavour() const { return sum_words == sum_words_scalar ? "scalar" : "avx512dq"; }

	// takes parts of `j` until none is left; returns how many it took
	uint64_t run_parts(job const &j)
	{
		if (0 == j.parts) return 0;
		uint64_t const per_part((j.words + j.parts - 1) / j.parts);
		uint64_t local(0), done(0);
		for (;;) {
			uint64_t const p(next_part.fetch_add(1, std::memory_order_relaxed));
			if (p >= j.parts) break;
			uint64_t const k0(p * per_part), k1(k0 + per_part < j.words ? k0 + per_part : j.words);
			local += sum_words(j.bytes, k0, k1);
			++done;
		}
		if (done) sum.fetch_add(local, std::memory_order_relaxed);
		return done;
	}

	void worker()
	{
		uint64_t seen(0);
		for (;;) {
			job j;
			{
				std::unique_lock <std::mutex> lock(mutex);
				wake.wait(lock, [&]{ return stopping || generation != seen; });
				if (stopping) return;
				seen = generation;
				j = current;
				++active;
			}
			uint64_t const done(run_parts(j));
			std::lock_guard <std::mutex> lock(mutex);
			parts_done += done;
			--active;
			if (0 == active) idle.notify_all();
		}
	}
};

void *v2ms_checksum_sink_create(uint64_t capacity_rows, uint32_t threads)
{
	auto *s(new v2ms_checksum_sink);
	s->checksums.assign(capacity_rows, 0);
	s->lengths.assign(capacity_rows, 0);
	for (uint32_t i(1); i < threads; ++i)
		s->workers.emplace_back([s]{ s->worker(); });
	return s;
}

void v2ms_checksum_sink_destroy(void *user)
{
	auto *s(static_cast<v2ms_checksum_sink *>(user));
	{
		std::lock_guard <std::mutex> lock(s->mutex);
		s->stopping = true;
	}
	s->wake.notify_all();
	for (auto &t : s->workers) t.join();
	delete s;
}

// a v2m_sink_fn; `user` is what v2ms_checksum_sink_create returned
int v2ms_checksum_sink_fn(void *user, uint64_t row, char const *bytes, uint64_t length)
{
	auto *s(static_cast<v2ms_checksum_sink *>(user));
	if (row >= s->checksums.size()) return 1;
	v2ms_checksum_sink::job j;
	j.bytes = bytes;
	j.words = length / 8;
	j.parts = j.words ? 4 * (s->workers.size() + 1) : 0;
	s->sum.store(0, std::memory_order_relaxed);
	if (j.parts) {
		{
			std::unique_lock <std::mutex> lock(s->mutex);
			s->idle.wait(lock, [&]{ return 0 == s->active; });        // (a worker that noticed the previous row late is let out first)
			s->current = j;
			s->parts_done = 0;
			s->next_part.store(0, std::memory_order_relaxed);
			++s->generation;
		}
		s->wake.notify_all();
		uint64_t const mine(s->run_parts(j));
		std::unique_lock <std::mutex> lock(s->mutex);
		s->parts_done += mine;
		s->idle.wait(lock, [&]{ return s->parts_done == j.parts && 0 == s->active; });
	}
	uint64_t acc(s->sum.load(std::memory_order_relaxed));
	if (length % 8) {
		uint64_t w(0);
		std::memcpy(&w, bytes + 8 * j.words, length % 8);
		acc += mix64((j.words + 1) * 0x9E3779B97F4A7C15ULL ^ w);
	}
	acc += mix64(length);
	s->checksums[row] = acc;
	s->lengths[row] = length;
	++s->rows;
	s->bytes += length;
	return 0;
}

uint64_t v2ms_checksum_sink_rows(void *user) { return static_cast<v2ms_checksum_sink *>(user)->rows; }
uint64_t v2ms_checksum_sink_bytes(void *user) { return static_cast<v2ms_checksum_sink *>(user)->bytes; }
uint64_t const *v2ms_checksum_sink_checksums(void *user) { return static_cast<v2ms_checksum_sink *>(user)->checksums.data(); }
uint64_t const *v2ms_checksum_sink_lengths(void *user) { return static_cast<v2ms_checksum_sink *>(user)->lengths.data(); }

// "scalar" or "avx512dq": which loop this sink sums words with (chosen at run time from the CPU's features)
char const *v2ms_checksum_sink_flavour(void *user) { return static_cast<v2ms_checksum_sink *>(user)->flavour(); }
// test hook: 0 = the scalar loop whatever the CPU offers
void v2ms_checksum_sink_force_scalar(void *user) { static_cast<v2ms_checksum_sink *>(user)->sum_words = sum_words_scalar; }

} // extern "C"
