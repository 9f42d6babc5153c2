#include "output.hh"
#include <cstdlib>
#include <cstdio>
#include <chrono>

#include <fcntl.h>
#include <sys/uio.h>
#include <sys/wait.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <exception>
#include <mutex>
#include <fstream>
#include <stdexcept>
#include <streambuf>
#include <thread>

namespace v2m::host {

output::output(gpu_context &gpu, char const *pipe_cmd, char const *chromosome_id, bool should_output_reference, bool should_output_unaligned, output_delegate &delegate):
	m_gpu(gpu), m_pipe_cmd(pipe_cmd), m_chromosome_id(chromosome_id), m_delegate(&delegate),
	m_should_output_reference(should_output_reference), m_should_output_unaligned(should_output_unaligned)
{
}


namespace {
	// --pipe (output.cc:26-38,49-68): `command dst_name` as a child process that reads the sequences from its standard
	// input instead of us writing the file.  libbio::subprocess<STDIN> with KEEP_STDERR there: the child's stderr is
	// ours; its stdout goes to /dev/null here (libbio is not part of the reference tree, so this detail is unpinned).
	class pipe_process {
	public:
		pipe_process(char const *command, char const *dst_name)
		{
			int in[2], status[2];
			if (0 != ::pipe2(in, O_CLOEXEC)) fail_errno("pipe");
			if (0 != ::pipe2(status, O_CLOEXEC)) { int const e(errno); ::close(in[0]); ::close(in[1]); errno = e; fail_errno("pipe"); }
			m_pid = ::fork();
			if (m_pid < 0) { int const e(errno); ::close(in[0]); ::close(in[1]); ::close(status[0]); ::close(status[1]); errno = e; fail_errno("fork"); }
			if (0 == m_pid) {
				// child: only async-signal-safe calls from here on
				if (::dup2(in[0], STDIN_FILENO) >= 0) {
					int const null_fd(::open("/dev/null", O_WRONLY));
					if (null_fd >= 0) { ::dup2(null_fd, STDOUT_FILENO); ::close(null_fd); }
					char *const argv[] = {const_cast<char *>(command), const_cast<char *>(dst_name), nullptr};
					::execvp(command, argv);
				}
				int const e(errno);
				(void) !::write(status[1], &e, sizeof(e));
				::_exit(127);
			}
			::close(in[0]);
			::close(status[1]);
			m_fd = in[1];
			int child_errno(0);
			ssize_t got;
			do { got = ::read(status[0], &child_errno, sizeof(child_errno)); } while (got < 0 && EINTR == errno);
			::close(status[0]);
			if (got > 0) {                                                  // exec failed (main.cc:333-338)
				::close(m_fd);
				m_fd = -1;
				int st;
				::waitpid(m_pid, &st, 0);
				throw std::runtime_error(std::string("Unable to execute subprocess. ") + command + ": " + std::strerror(child_errno));
			}
		}

		pipe_process(pipe_process const &) = delete;
		pipe_process &operator=(pipe_process const &) = delete;

		~pipe_process()
		{
			if (m_fd >= 0) ::close(m_fd);
			if (m_pid > 0 && !m_waited) { int st; ::waitpid(m_pid, &st, 0); }
		}

		int fd() const { return m_fd; }

		// Close the child's standard input and wait for it; anything but exit(0) is an error (main.cc:341-367).
		void finish()
		{
			::close(m_fd);
			m_fd = -1;
			int st(0);
			pid_t r;
			do { r = ::waitpid(m_pid, &st, 0); } while (r < 0 && EINTR == errno);
			m_waited = true;
			if (r == m_pid && WIFEXITED(st) && 0 == WEXITSTATUS(st)) return;
			std::string msg("Subprocess with PID " + std::to_string(m_pid) + " exited with status ");
			if (r != m_pid) msg += "0 (exiting reason not known)";
			else if (WIFSIGNALED(st)) msg += std::to_string(WTERMSIG(st)) + " (terminated by signal)";
			else if (WIFSTOPPED(st)) msg += std::to_string(WSTOPSIG(st)) + " (stopped by signal)";
			else msg += std::to_string(WEXITSTATUS(st));
			throw std::runtime_error(msg);
		}

	private:
		[[noreturn]] static void fail_errno(char const *what)
		{
			throw std::runtime_error(std::string("Unable to execute subprocess. ") + what + ": " + std::strerror(errno));
		}

		pid_t m_pid{-1};
		int m_fd{-1};
		bool m_waited{};
	};


	// std::ostream over a file descriptor (the reference opens a libbio::file_ostream on the child's stdin handle).
	class fd_streambuf final : public std::streambuf {
	public:
		explicit fd_streambuf(int fd) : m_fd(fd), m_buffer(1u << 20) { setp(m_buffer.data(), m_buffer.data() + m_buffer.size()); }
		~fd_streambuf() override { sync(); }

	protected:
		int_type overflow(int_type ch) override
		{
			if (!flush_buffer()) return traits_type::eof();
			if (!traits_type::eq_int_type(ch, traits_type::eof())) { *pptr() = traits_type::to_char_type(ch); pbump(1); }
			return traits_type::not_eof(ch);
		}

		std::streamsize xsputn(char const *s, std::streamsize n) override
		{
			if (n < std::streamsize(m_buffer.size() / 4)) return std::streambuf::xsputn(s, n);
			if (!flush_buffer() || !write_all(s, std::size_t(n))) return 0;      // row bodies go straight from the pinned ring
			return n;
		}

		int sync() override { return flush_buffer() ? 0 : -1; }

	private:
		bool flush_buffer()
		{
			bool const ok(write_all(pbase(), std::size_t(pptr() - pbase())));
			setp(m_buffer.data(), m_buffer.data() + m_buffer.size());
			return ok;
		}

		bool write_all(char const *p, std::size_t n)
		{
			while (n) {
				ssize_t const w(::write(m_fd, p, n));
				if (w < 0) { if (EINTR == errno) continue; return false; }   // EPIPE when the child has gone away (SIGPIPE is ignored)
				p += w;
				n -= std::size_t(w);
			}
			return true;
		}

		int m_fd;
		std::vector<char> m_buffer;
	};
}


std::string output::prefixed(std::string const &name, char sep) const
{
	return m_chromosome_id ? std::string(m_chromosome_id) + sep + name : name;
}


namespace {
	// Lets the contexts' sink calls through in row order.  A context whose row is not due waits inside its sink call; the
	// library meanwhile keeps filling that context's other pinned slot, so the wait costs nothing but the turn itself.
	struct turnstile {
		std::mutex mutex;
		std::condition_variable turn;
		std::uint64_t next_row{};
		bool failed{};
	};

	struct turn_state {
		turnstile *gate;
		std::vector<std::uint64_t> const *global_row;   // of each row of this context's batch
		v2m_sink_fn sink;
		void *user;
	};

	int turn_sink(void *user, uint64_t row, char const *bytes, uint64_t length)
	{
		auto &st(*static_cast<turn_state *>(user));
		std::uint64_t const global((*st.global_row)[row]);
		std::unique_lock<std::mutex> lock(st.gate->mutex);
		st.gate->turn.wait(lock, [&] { return st.gate->failed || st.gate->next_row == global; });
		if (st.gate->failed) return 1;
		int const rc(st.sink(st.user, global, bytes, length));   // (under the lock: one writer at a time, in order)
		if (rc) st.gate->failed = true; else ++st.gate->next_row;
		lock.unlock();
		st.gate->turn.notify_all();
		return rc;
	}
}


// splice() over several contexts that hold the chromosome copies dealt round-robin (set_copy_interleave): the sink sees the
// rows in the batch's order, exactly as from one context.
void output::splice_in_turns(row_set const &rows, v2m_sink_fn sink, void *user)
{
	if (rows.any_cuts) throw std::runtime_error("rows that switch copies need the whole path matrix on their GPU");
	std::vector<gpu_context *> gpus{&m_gpu};
	gpus.insert(gpus.end(), m_more_gpus.begin(), m_more_gpus.end());
	std::size_t const g(gpus.size());
	if (m_copy_interleave.world != g) throw std::runtime_error("the copy interleave was made for another number of GPU contexts");
	std::vector<std::vector<std::uint32_t>> local_copy(g);
	std::vector<std::vector<std::uint64_t>> global_row(g);
	for (std::uint64_t i(0); i < rows.copy_index.size(); ++i) {
		std::uint32_t const c(rows.copy_index[i]);
		bool const is_ref(V2M_PLOIDY_MAX == c);
		std::size_t const k(is_ref ? 0 : m_copy_interleave.owner(c));          // the REF row needs no path bits: first context
		local_copy[k].push_back(is_ref ? c : std::uint32_t(m_copy_interleave.local(c)));
		global_row[k].push_back(i);
	}
	turnstile gate;
	std::vector<std::exception_ptr> errors(g);
	std::vector<std::thread> threads;
	for (std::size_t k(0); k < g; ++k) {
		threads.emplace_back([&, k] {
			try {
				if (local_copy[k].empty()) return;
				v2m_row_batch batch{};
				batch.n_rows = local_copy[k].size();
				batch.copy_index = local_copy[k].data();
				turn_state st{&gate, &global_row[k], sink, user};
				gpus[k]->check(v2m_splice_rows(gpus[k]->get(), &batch, m_should_output_unaligned ? V2M_SPLICE_UNALIGNED : 0u, turn_sink, &st));
			} catch (...) {
				errors[k] = std::current_exception();
				{ std::lock_guard<std::mutex> const lock(gate.mutex); gate.failed = true; }   // nobody waits for this context's rows any longer
				gate.turn.notify_all();
			}
		});
	}
	for (auto &t : threads) t.join();
	// the first error in context order that is not just "my sink was told to stop"
	for (auto const &e : errors) {
		if (!e) continue;
		try { std::rethrow_exception(e); }
		catch (gpu_error const &ge) { if (V2M_ERR_SINK != ge.code) throw; }
	}
	for (auto const &e : errors) if (e) std::rethrow_exception(e);
}


void output::splice(row_set const &rows, v2m_sink_fn sink, void *user)
{
	if (m_interleaved && !m_more_gpus.empty()) { splice_in_turns(rows, sink, user); return; }
	v2m_row_batch batch{};
	batch.n_rows = rows.copy_index.size();
	batch.copy_index = rows.copy_index.data();
	std::vector<std::uint32_t> rebased;
	if (!m_copy_shards.empty() && 0 != m_copy_shards.front().first) {       // the first context's matrix starts at another copy
		rebased = rebased_copies(rows, 0, rows.copy_index.size(), m_copy_shards.front());
		batch.copy_index = rebased.data();
	}
	if (rows.any_cuts) {
		batch.cut_offsets = rows.cut_offsets.data();
		batch.cut_nodes = rows.cut_nodes.data();
		batch.cut_copies = rows.cut_copies.data();
	}
	m_gpu.check(v2m_splice_rows(m_gpu.get(), &batch, m_should_output_unaligned ? V2M_SPLICE_UNALIGNED : 0u, sink, user));
}


// Copy indices of rows [first, last) relative to the shard that holds them.
std::vector<std::uint32_t> output::rebased_copies(row_set const &rows, std::uint64_t first, std::uint64_t last, copy_shard shard)
{
	if (rows.any_cuts) throw std::runtime_error("rows that switch copies need the whole path matrix on their GPU");
	std::vector<std::uint32_t> out(rows.copy_index.begin() + first, rows.copy_index.begin() + last);
	for (auto &c : out) {
		if (V2M_PLOIDY_MAX == c) continue;
		if (c < shard.first || c >= shard.end) throw std::runtime_error("row of chromosome copy " + std::to_string(c) + " was given to a GPU that does not hold it");
		c -= std::uint32_t(shard.first);
	}
	return out;
}


namespace {
	struct a2m_state { std::ostream *stream; std::vector<std::string> const *ids; output_delegate *delegate; };

	int a2m_sink(void *user, uint64_t row, char const *bytes, uint64_t length)
	{
		auto &st(*static_cast<a2m_state *>(user));
		*st.stream << '>' << (*st.ids)[row] << '\n';                       // sequence_writer.cc:35-36
		st.stream->write(bytes, std::streamsize(length));
		*st.stream << '\n';                                                // haplotype_output.cc:57,76
		st.delegate->handled_sequences(u32(1 + row));                      // haplotype_output.cc:58,78-79
		return st.stream->good() ? 0 : 1;
	}

	struct separate_state { std::vector<std::string> const *names; char const *pipe_cmd; std::exception_ptr error; };

	// output_sequence_file always passes dst_name as the FASTA identifier, whatever should_include_fasta_header says
	// (output.cc:36,42), and no newline follows the body.
	void write_sequence_file(std::ostream &os, std::string const &name, char const *bytes, uint64_t length)
	{
		os << '>' << name << '\n';
		os.write(bytes, std::streamsize(length));
		os.flush();
	}

	int separate_sink(void *user, uint64_t row, char const *bytes, uint64_t length)
	{
		auto &st(*static_cast<separate_state *>(user));
		auto const &name((*st.names)[row]);
		if (st.pipe_cmd) {                                                  // output.cc:26-38
			try {
				pipe_process proc(st.pipe_cmd, name.c_str());
				bool good;
				{
					fd_streambuf buf(proc.fd());
					std::ostream os(&buf);
					write_sequence_file(os, name, bytes, length);
					good = os.good();
				}
				proc.finish();                                                  // throws if the child did not exit(0)
				if (!good) throw std::runtime_error("error while writing to the subprocess for " + name);
			} catch (...) {
				st.error = std::current_exception();
				return 1;
			}
			return 0;
		}
		std::ofstream os(name, std::ios::binary | std::ios::trunc);
		if (!os) return 1;
		write_sequence_file(os, name, bytes, length);
		return os.good() ? 0 : 1;
	}
}


namespace {
	// Writer threads behind a v2m_hold_sink_fn (v2m_splice_rows_held): the sink queues a row and returns, a writer writes the row's file from the
	// library's pinned slot and only then gives the row back.  One writer on one file moves 6-11 GB/s through the page cache, the link delivers
	// 56: a file per sequence scales with writers (profiles/r04/e2e_config2_file_destinations.txt), and with rows that may be kept ONE GPU
	// context is enough to feed them all.
	class row_writer_pool {
	public:
		typedef bool (*write_fn)(void *user, std::uint64_t row, char const *bytes, std::uint64_t length);

		row_writer_pool(unsigned n_threads, write_fn write, void *user) : m_write(write), m_user(user)
		{
			for (unsigned i(0); i < std::max(1u, n_threads); ++i) m_threads.emplace_back([this] { work(); });
		}

		~row_writer_pool() { finish(); }

		// the v2m_hold_sink_fn: `user` is the pool
		static int sink(void *user, uint64_t row, char const *bytes, uint64_t length, v2m_row_hold *hold)
		{
			auto &self(*static_cast<row_writer_pool *>(user));
			{
				std::lock_guard<std::mutex> const lock(self.m_mutex);
				if (self.m_failed) return 1;                                    // not accepted: the library ends the call
				self.m_jobs.push_back({row, bytes, length, hold});
			}
			self.m_wake.notify_one();
			return 0;
		}

		// every queued row written (or dropped after a failure) and released; false if a write failed
		bool finish()
		{
			{
				std::lock_guard<std::mutex> const lock(m_mutex);
				m_stopping = true;
			}
			m_wake.notify_all();
			for (auto &t : m_threads) t.join();
			m_threads.clear();
			return !m_failed;
		}

	private:
		struct job { std::uint64_t row; char const *bytes; std::uint64_t length; v2m_row_hold *hold; };

		void work()
		{
			for (;;) {
				job j;
				bool write;
				{
					std::unique_lock<std::mutex> lock(m_mutex);
					m_wake.wait(lock, [&] { return m_stopping || !m_jobs.empty(); });
					if (m_jobs.empty()) return;
					j = m_jobs.front();
					m_jobs.pop_front();
					write = !m_failed;
				}
				bool const ok(!write || m_write(m_user, j.row, j.bytes, j.length));
				v2m_row_release(j.hold);                                            // whatever happened to the file: the slot is the library's
				if (!ok) { std::lock_guard<std::mutex> const lock(m_mutex); m_failed = true; }
			}
		}

		write_fn m_write;
		void *m_user;
		std::mutex m_mutex;
		std::condition_variable m_wake;
		std::deque<job> m_jobs;
		std::vector<std::thread> m_threads;
		bool m_stopping{}, m_failed{};
	};

	unsigned env_count(char const *name, unsigned fallback, unsigned lo, unsigned hi)
	{
		char const *const e(std::getenv(name));
		long const v((e && *e) ? std::strtol(e, nullptr, 10) : 0);
		return unsigned(std::min<long>(hi, std::max<long>(lo, v > 0 ? v : long(fallback))));
	}

	struct held_turn_state { std::vector<std::uint64_t> const *global_row; v2m_hold_sink_fn sink; void *user; };

	int held_turn_sink(void *user, uint64_t row, char const *bytes, uint64_t length, v2m_row_hold *hold)
	{
		auto const &st(*static_cast<held_turn_state *>(user));
		return st.sink(st.user, (*st.global_row)[row], bytes, length, hold);
	}
}


// splice() for rows that need no order among themselves and a sink that keeps them for a while (v2m_splice_rows_held): one call per context,
// each on its own thread when there are several; the sink sees the batch's row indices.
void output::splice_held(row_set const &rows, v2m_hold_sink_fn sink, void *user)
{
	unsigned const n_slots(env_count("V2M_HELD_SLOTS", 4, 2, 64));
	std::uint32_t const flags(m_should_output_unaligned ? V2M_SPLICE_UNALIGNED : 0u);
	if (!(m_interleaved && !m_more_gpus.empty())) {
		v2m_row_batch batch{};
		batch.n_rows = rows.copy_index.size();
		batch.copy_index = rows.copy_index.data();
		std::vector<std::uint32_t> rebased;
		if (!m_copy_shards.empty() && 0 != m_copy_shards.front().first) {
			rebased = rebased_copies(rows, 0, rows.copy_index.size(), m_copy_shards.front());
			batch.copy_index = rebased.data();
		}
		if (rows.any_cuts) {
			batch.cut_offsets = rows.cut_offsets.data();
			batch.cut_nodes = rows.cut_nodes.data();
			batch.cut_copies = rows.cut_copies.data();
		}
		m_gpu.check(v2m_splice_rows_held(m_gpu.get(), &batch, flags, n_slots, sink, user));
		return;
	}
	if (rows.any_cuts) throw std::runtime_error("rows that switch copies need the whole path matrix on their GPU");
	std::vector<gpu_context *> gpus{&m_gpu};
	gpus.insert(gpus.end(), m_more_gpus.begin(), m_more_gpus.end());
	std::size_t const g(gpus.size());
	if (m_copy_interleave.world != g) throw std::runtime_error("the copy interleave was made for another number of GPU contexts");
	std::vector<std::vector<std::uint32_t>> local_copy(g);
	std::vector<std::vector<std::uint64_t>> global_row(g);
	for (std::uint64_t i(0); i < rows.copy_index.size(); ++i) {
		std::uint32_t const c(rows.copy_index[i]);
		bool const is_ref(V2M_PLOIDY_MAX == c);
		std::size_t const k(is_ref ? 0 : m_copy_interleave.owner(c));
		local_copy[k].push_back(is_ref ? c : std::uint32_t(m_copy_interleave.local(c)));
		global_row[k].push_back(i);
	}
	std::vector<std::exception_ptr> errors(g);
	std::vector<std::thread> threads;
	for (std::size_t k(0); k < g; ++k) {
		threads.emplace_back([&, k] {
			try {
				if (local_copy[k].empty()) return;
				v2m_row_batch batch{};
				batch.n_rows = local_copy[k].size();
				batch.copy_index = local_copy[k].data();
				held_turn_state st{&global_row[k], sink, user};
				gpus[k]->check(v2m_splice_rows_held(gpus[k]->get(), &batch, flags, n_slots, held_turn_sink, &st));
			} catch (...) {
				errors[k] = std::current_exception();
			}
		});
	}
	for (auto &t : threads) t.join();
	for (auto const &e : errors) {                                                      // the first error that is not just "my sink was told to stop"
		if (!e) continue;
		try { std::rethrow_exception(e); }
		catch (gpu_error const &ge) { if (V2M_ERR_SINK != ge.code) throw; }
	}
	for (auto const &e : errors) if (e) std::rethrow_exception(e);
}


void output::write_a2m(row_set const &rows, std::ostream &stream)
{
	a2m_state st{&stream, &rows.ids, m_delegate};
	splice(rows, a2m_sink, &st);
}


void output::write_separate(row_set const &rows)
{
	separate_state st{&rows.ids, m_pipe_cmd, nullptr};
	if (!m_pipe_cmd) {
		// Files of their own need no order among themselves and no single writer: the rows stay in the library's pinned slots while a pool
		// of threads writes them out (V2M_WRITER_THREADS, default 8 in all), so that ONE context's link is fed by as many writers as it takes.
		// With several GPU contexts each one's thread delivers into the same pool.
		struct file_writer {
			static bool write(void *user, std::uint64_t row, char const *bytes, std::uint64_t length) { return 0 == separate_sink(user, row, bytes, length); }
		};
		row_writer_pool pool(env_count("V2M_WRITER_THREADS", 8, 1, 64), &file_writer::write, &st);
		std::exception_ptr error;
		try { splice_held(rows, &row_writer_pool::sink, &pool); } catch (...) { error = std::current_exception(); }
		bool const written(pool.finish());
		if (!written) throw std::runtime_error("error while writing a sequence file");      // the telling error, not "sink failed"
		if (error) std::rethrow_exception(error);
		return;
	}
	try {
		// Subprocesses (--pipe) are started one after the other, as the reference starts them (output.cc:26-38).
		splice(rows, separate_sink, &st);
	} catch (...) {
		if (st.error) std::rethrow_exception(st.error);                     // what went wrong with the subprocess, not "sink failed"
		throw;
	}
}


namespace {
	// One shard of an aligned A2M file: rows [first, first + n) of the batch, written at precomputed file offsets.
	struct shard_state {
		int fd;
		std::vector<std::string> const *ids;
		std::uint64_t const *offsets;      // file offset of each row of the whole batch
		std::uint64_t first;               // batch index of the shard's row 0
		std::mutex *one_writer;            // buffered writes into ONE file run under its inode lock anyway; several threads inside the kernel at
		                                   // once only fight over it (tmpfs, two writers: 20 GB in 5.4 s against 3.2 s for one; profiles/r04/
		                                   // e2e_config2_file_destinations.txt), so they take turns out here
	};

	int shard_sink(void *user, uint64_t row, char const *bytes, uint64_t length)
	{
		auto const &st(*static_cast<shard_state *>(user));
		std::uint64_t const i(st.first + row);
		std::string const header(">" + (*st.ids)[i] + "\n");
		char newline('\n');
		// header, body, newline as one gather list, written in turns of at most kTurnBytes: the contexts' threads alternate at that grain,
		// so that on a file system whose writes to one file do run in parallel (not tmpfs / ext4 under buffered I/O, where the inode lock
		// serialises them anyway: profiles/r04/e2e_config2_file_destinations.txt) a 100-MB row does not hold the others up for its whole length
		constexpr std::size_t kTurnBytes(std::size_t(16) << 20);
		struct iovec iov[3] = {{const_cast<char *>(header.data()), header.size()}, {const_cast<char *>(bytes), length}, {&newline, 1}};
		std::uint64_t off(st.offsets[i]);
		int k(0);
		while (k < 3) {   // pwritev may write less than asked
			struct iovec turn[3];
			int n(0);
			std::size_t budget(kTurnBytes);
			for (int j(k); j < 3 && budget; ++j) {
				turn[n] = iov[j];
				if (turn[n].iov_len > budget) turn[n].iov_len = budget;
				budget -= turn[n].iov_len;
				++n;
			}
			ssize_t w;
			{
				std::lock_guard<std::mutex> const lock(*st.one_writer);
				w = ::pwritev(st.fd, turn, n, off_t(off));
			}
			if (w < 0 && EINTR == errno) continue;
			if (w <= 0) return 1;
			off += std::uint64_t(w);
			std::size_t left = std::size_t(w);
			while (k < 3 && left >= iov[k].iov_len) { left -= iov[k].iov_len; ++k; }
			if (k < 3) { iov[k].iov_base = static_cast<char *>(iov[k].iov_base) + left; iov[k].iov_len -= left; }
		}
		return 0;
	}
}


void output::write_a2m_sharded(row_set const &rows, char const *dst_name)
{
	std::vector<gpu_context *> gpus{&m_gpu};
	gpus.insert(gpus.end(), m_more_gpus.begin(), m_more_gpus.end());
	std::uint64_t const n(rows.copy_index.size()), L(v2m_aligned_length(m_gpu.get()));
	std::vector<std::uint64_t> offsets(n + 1, 0);
	for (std::uint64_t i(0); i < n; ++i) offsets[i + 1] = offsets[i] + 1 + rows.ids[i].size() + 1 + L + 1;   // '>' id '\n' body '\n'

	int const fd(::open(dst_name, O_WRONLY | O_CREAT | O_TRUNC, 0644));
	if (fd < 0) throw std::runtime_error(std::string("unable to open ") + dst_name + " for writing");
	struct closer { int fd; ~closer() { ::close(fd); } } const close_on_exit{fd};
	if (0 != ::ftruncate(fd, off_t(offsets[n]))) { /* not all targets can be sized (e.g. /dev/null); pwrite extends regular files anyway */ }

	std::size_t const g(gpus.size());
	// Which rows go to which context: with a sharded path matrix, the rows whose copy the context holds (REF: the first
	// context); otherwise equal contiguous blocks.  Rows are in copy order (haplotype_output.cc:62-65), so either way a
	// context's rows are one contiguous block of the file.
	std::vector<std::uint64_t> bounds(g + 1, n);
	bounds[0] = 0;
	bool const sharded(!m_copy_shards.empty());
	if (sharded) {
		if (m_copy_shards.size() != g) throw std::runtime_error("one copy shard per GPU context is needed");
		if (rows.any_cuts) throw std::runtime_error("rows that switch copies need the whole path matrix on every GPU");
		std::uint64_t i(0);
		for (std::size_t k(0); k < g; ++k) {
			while (i < n && (V2M_PLOIDY_MAX == rows.copy_index[i] ? 0 == k : rows.copy_index[i] < m_copy_shards[k].end)) ++i;
			bounds[k + 1] = i;
		}
		if (bounds[g] != n) throw std::runtime_error("a row's chromosome copy lies outside every GPU's shard");
	} else {
		for (std::size_t k(0); k <= g; ++k) bounds[k] = n * k / g;
	}
	std::vector<std::exception_ptr> errors(g);
	std::mutex one_writer;
	std::vector<std::thread> threads;
	for (std::size_t k(0); k < g; ++k) {
		std::uint64_t const first(bounds[k]), last(bounds[k + 1]);
		threads.emplace_back([&, k, first, last] {
			try {
				if (first == last) return;
				v2m_row_batch batch{};
				batch.n_rows = last - first;
				batch.copy_index = rows.copy_index.data() + first;
				std::vector<std::uint32_t> rebased;
				if (sharded) {
					rebased = rebased_copies(rows, first, last, m_copy_shards[k]);
					batch.copy_index = rebased.data();
				}
				std::vector<std::uint64_t> cut_offsets;
				if (rows.any_cuts) {   // re-base the CSR offsets of the shard
					cut_offsets.assign(rows.cut_offsets.begin() + first, rows.cut_offsets.begin() + last + 1);
					std::uint64_t const base(cut_offsets.front());
					for (auto &o : cut_offsets) o -= base;
					batch.cut_offsets = cut_offsets.data();
					batch.cut_nodes = rows.cut_nodes.data() + base;
					batch.cut_copies = rows.cut_copies.data() + base;
				}
				shard_state st{fd, &rows.ids, offsets.data(), first, &one_writer};
				gpus[k]->check(v2m_splice_rows(gpus[k]->get(), &batch, 0u, shard_sink, &st));
			} catch (...) {
				errors[k] = std::current_exception();
			}
		});
	}
	for (auto &t : threads) t.join();
	for (auto const &e : errors) if (e) std::rethrow_exception(e);
	for (std::uint64_t i(0); i < n; ++i) m_delegate->handled_sequences(u32(1 + i));
}


void output::output_a2m(variant_graph const &graph, char const *dst_name)          // output.cc:47-76
{
	if (m_pipe_cmd) {                                                               // :49-68
		pipe_process proc(m_pipe_cmd, dst_name);
		bool good(false);
		std::exception_ptr error;
		try {
			fd_streambuf buf(proc.fd());
			std::ostream stream(&buf);
			output_a2m(graph, stream);
			stream.flush();
			good = stream.good();
		} catch (...) {
			error = std::current_exception();
		}
		proc.finish();                                                              // a failed child is the more telling error
		if (error) std::rethrow_exception(error);
		if (!good) throw std::runtime_error(std::string("error while writing to the subprocess for ") + dst_name);
		return;
	}
	if (!m_more_gpus.empty() && !m_should_output_unaligned && !m_interleaved) {
		write_a2m_sharded(a2m_rows(graph), dst_name);
		return;
	}
	std::ofstream stream(dst_name, std::ios::binary | std::ios::trunc);
	if (!stream) throw std::runtime_error(std::string("unable to open ") + dst_name + " for writing");
	output_a2m(graph, stream);
	stream.flush();
	if (!stream) throw std::runtime_error(std::string("error while writing ") + dst_name);
}


// --- haplotype_output (haplotype_output.cc:38-132) ------------------------------------------------------------
output::row_set haplotype_output::rows_for(variant_graph const &graph, char sep, char const *suffix)
{
	row_set rows;
	bool const a2m('\t' == sep);
	if (m_should_output_reference) {                                                // :48-59 / :87-103
		rows.ids.push_back(prefixed("REF", sep) + suffix);
		rows.copy_index.push_back(V2M_PLOIDY_MAX);
	}
	for (std::size_t sample_idx(0); sample_idx < graph.sample_names.size(); ++sample_idx) {   // :62 / :106
		auto const &sample(graph.sample_names[sample_idx]);
		for (u32 chr_copy_idx(0); chr_copy_idx < graph.sample_ploidy(sample_idx); ++chr_copy_idx) {
			m_delegate->will_handle_sample(sample, u32(sample_idx), chr_copy_idx);
			rows.ids.push_back(prefixed(sample + (a2m ? '-' : '.') + std::to_string(1 + chr_copy_idx), sep) + suffix);   // :69-72 / :117
			rows.copy_index.push_back(graph.ploidy_csum[sample_idx] + chr_copy_idx);                                // :28-31
		}
	}
	return rows;
}


void haplotype_output::output_a2m(variant_graph const &graph, std::ostream &stream)
{
	write_a2m(rows_for(graph, '\t', ""), stream);
}


void haplotype_output::output_separate(variant_graph const &graph, bool should_include_fasta_header)
{
	char const *const suffix(should_include_fasta_header ? (m_should_output_unaligned ? ".fa" : ".a2m") : "");   // :94-100
	write_separate(rows_for(graph, '.', suffix));
}


// --- founder_sequence_greedy_output (founder_sequence_greedy_output.cc:515-597) ----------------------------------
output::row_set founder_sequence_greedy_output::rows_for(char sep, char const *suffix)
{
	if (m_cut_positions.empty() || 0 != m_cut_positions.front())
		throw std::runtime_error("cut positions must start with node 0");            // asserted at founder_sequence_greedy_output.cc:101-102
	std::size_t const col_rows(m_cut_positions.size() - 1);
	if (m_assigned_samples.size() != col_rows * m_founder_count)
		throw std::runtime_error("assigned_samples must have (cut positions - 1) x founders entries");

	row_set rows;
	rows.any_cuts = true;
	if (m_should_output_reference) {                                                // :519-531
		rows.ids.push_back(prefixed("REF", sep) + suffix);
		rows.copy_index.push_back(V2M_PLOIDY_MAX);
		rows.cut_offsets.push_back(0);
	}
	// the delegate's copy switch at each cut node (:106-114): one (node, copy) pair per cut and founder -- 17 M of them at config 4,
	// written by a few threads (every founder's run of the two arrays is its own)
	rows.cut_nodes.resize(std::size_t(m_founder_count) * col_rows);
	rows.cut_copies.resize(std::size_t(m_founder_count) * col_rows);
	for (u32 col(0); col < m_founder_count; ++col) {                                // :533-549
		m_delegate->will_handle_founder_sequence(col);
		rows.ids.push_back(prefixed(std::to_string(1 + col), sep) + suffix);        // :542
		rows.copy_index.push_back(V2M_PLOIDY_MAX);
		rows.cut_offsets.push_back(std::size_t(col + 1) * col_rows);                // (cut_offsets starts out as {0}: row r's cuts are [r], [r + 1])
	}
	{
		std::atomic<u32> next(0);
		auto const work([&] {
			for (u32 col; (col = next.fetch_add(1)) < m_founder_count;) {
				std::size_t const base(std::size_t(col) * col_rows);
				for (std::size_t k(0); k < col_rows; ++k) {
					rows.cut_nodes[base + k] = m_cut_positions[k];
					rows.cut_copies[base + k] = m_assigned_samples[base + k];
				}
			}
		});
		std::vector<std::thread> pool;
		unsigned const n_threads(std::size_t(m_founder_count) * col_rows >= (std::size_t(1) << 20) ? std::min<unsigned>(8, m_founder_count) : 1u);
		for (unsigned t(1); t < n_threads; ++t) pool.emplace_back(work);
		work();
		for (auto &t : pool) t.join();
	}
	return rows;
}


void founder_sequence_greedy_output::output_a2m(variant_graph const &, std::ostream &stream)
{
	bool const timing(nullptr != std::getenv("V2M_FOUNDER_TIMING"));
	auto const t0(std::chrono::steady_clock::now());
	row_set const rows(rows_for('\t', ""));
	auto const t1(std::chrono::steady_clock::now());
	write_a2m(rows, stream);
	if (timing) std::fprintf(stderr, "[founder] output: row tables %.3f s, rows through the GPU %.3f s\n", std::chrono::duration<double>(t1 - t0).count(), std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count());
}


void founder_sequence_greedy_output::output_separate(variant_graph const &, bool should_include_fasta_header)
{
	char const *const suffix(should_include_fasta_header ? (m_should_output_unaligned ? ".fa" : ".a2m") : "");
	write_separate(rows_for('.', suffix));
}

} // namespace v2m::host
