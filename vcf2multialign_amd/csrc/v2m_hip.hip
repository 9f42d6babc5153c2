// v2m_hip.hip -- implementation of the C ABI in include/v2m_hip.h for MI355X (gfx950).
//
// Host side of the device path: validates and narrows the variant graph, derives the device
// tables, owns all HBM allocations and the two HIP streams (compute + D2H), and launches the
// kernels in kernels.hpp.  There is no CPU fallback anywhere in this file: every entry point
// either runs on the GPU or returns an error.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include <type_traits>

#include "../../include/v2m_hip.h"
#include "kernels.hpp"
#include "founder_kernels.hpp"

using v2m::u32;
using v2m::u64;

namespace {

thread_local std::string g_create_error;

struct dev_buf {
	void *p{};
	size_t bytes{};
	dev_buf() = default;
	dev_buf(dev_buf const &) = delete;
	dev_buf &operator=(dev_buf const &) = delete;
	~dev_buf() { reset(); }
	void reset() { if (p) (void) hipFree(p); p = nullptr; bytes = 0; }
	hipError_t ensure(size_t n)
	{
		if (n <= bytes) return hipSuccess;
		reset();
		if (0 == n) return hipSuccess;
		hipError_t const st(hipMalloc(&p, n));
		if (hipSuccess == st) bytes = n; else p = nullptr;
		return st;
	}
	template <typename T> T *as() const { return static_cast<T *>(p); }
};

struct pinned_buf {
	void *p{};
	size_t bytes{};
	~pinned_buf() { reset(); }
	void reset() { if (p) (void) hipHostFree(p); p = nullptr; bytes = 0; }
	hipError_t ensure(size_t n)
	{
		if (n <= bytes) return hipSuccess;
		reset();
		if (0 == n) return hipSuccess;
		hipError_t const st(hipHostMalloc(&p, n, hipHostMallocDefault));
		if (hipSuccess == st) bytes = n; else p = nullptr;
		return st;
	}
	template <typename T> T *as() const { return static_cast<T *>(p); }
};

struct event_pair { hipEvent_t begin{}, end{}; };

} // namespace


// One pinned slot of v2m_splice_rows_held: the rows of one slice, and how many of them the sink still holds.  A row's hold IS its
// slot (include/v2m_hip.h: v2m_row_hold), so that a release is one decrement under the ring's mutex from whatever thread.
struct held_ring_state {
	std::mutex mutex;
	std::condition_variable released;
};

struct v2m_row_hold {
	pinned_buf host;
	hipEvent_t copied{};                // the slice's D2H copy has landed in `host`
	uint64_t outstanding{};             // delivered rows not yet released (under ring->mutex)
	held_ring_state *ring{};
};


struct v2m_ctx {
	int device{};
	u32 n_cus{256};          // compute units of the device (the lines16 transpose sizes its spans by it)
	hipStream_t stream{};
	hipStream_t copy_stream{};
	std::string err;

	// profiling
	bool profiling{};
	std::vector<event_pair> events[V2M_KERNEL_COUNT];
	std::vector<event_pair> free_events;

	// graph
	bool has_graph{};
	u64 n_nodes{}, n_edges{}, ref_len{}, aligned_len{}, label_bytes{};
	bool has_nul_byte{};                // ref_seq or a label holds a 0 byte: the unaligned kernels' padding marker (kernels.hpp), so --unaligned refuses
	u32 n_tiles{};
	std::vector<u32> h_csum;            // alt_edge_count_csum narrowed, [N + 1]
	std::vector<u32> h_tgt_prefix_max;  // [E + 1]: max target over edges < e (cut validation)
	dev_buf d_ref, d_ref_pos, d_aln_pos, d_spans, d_patches, d_labels, d_template, d_overlappable;
	dev_buf d_ovl_rank, d_blocker_masks;   // per word: overlappable edges before it; per overlappable edge: who can block it
	dev_buf d_template0;   // the REF row with 0 as padding byte (unaligned mode), built on first use
	bool has_template0{};
	dev_buf d_tile_edge_begin, d_cross_offsets, d_cross_edges;

	// paths_by_chrom_copy_and_edge
	u64 const *d_paths{};
	dev_buf owned_paths;
	dev_buf d_slice_src;     // v2m_upload_path_slice: the un-transposed slice (released after the transpose)
	u64 path_rows{}, path_cols{};
	u64 path_pitch{};        // words from one copy's column to the next (path_rows / 64 for caller-supplied matrices)
	// founder search: the bound matrix transposed back to edge-major bits (paths_by_edge_and_chrom_copy), made by the first of the
	// v2m_pbwt_* calls after a matrix is bound and kept for the following ones (one founder run makes two); dropped with the binding
	dev_buf d_by_edge;
	bool by_edge_valid{};

	struct transpose_pick { u64 rows, cols, src_pitch, dst_pitch; std::string kernel; };
	std::vector<transpose_pick> transpose_choice;   // per matrix shape: which transpose kernel measured fastest

	// store flavour of the aligned splice: -1 = not calibrated yet, 0 = plain, 1 = nontemporal
	int store_mode{-1};
	int unaligned_store_mode{-1};   // the same for the unaligned splice
	std::string info;

	// per-call scratch
	// segment tables of the row batch being resolved: two pinned staging areas used in turn, so that the host can prepare and
	// queue the next slice while the previous one is still running (each area is reused only after its uploads have left it)
	pinned_buf h_row_stage[2];
	hipEvent_t ev_row_stage[2]{};
	bool row_stage_in_flight[2]{};
	int row_stage_next{};
	dev_buf d_resolve_queue, d_resolve_count;   // (row, word) pairs the streaming resolve pass leaves to the dense one
	dev_buf d_eff, d_row_bits, d_seg_offsets, d_seg_edge_begin, d_seg_copy, d_sums, d_lengths, d_needs_serial, d_tile_counts, d_row_lengths;
	dev_buf ring[2];
	pinned_buf host_ring[2];
	// v2m_splice_rows_held: more pinned slots than the two above, each kept until the sink has released its rows
	held_ring_state held_state;
	std::vector<std::unique_ptr<v2m_row_hold>> held_ring;
	pinned_buf trials_stage[2];   // v2m_pbwt_cut_trials_streamed: the pairs' way back to the host
	hipEvent_t ev_compute[2]{}, ev_copy[2]{};
};


namespace {

int fail(v2m_ctx *ctx, int code, char const *fmt, ...)
{
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	std::vsnprintf(buf, sizeof(buf), fmt, ap);
	va_end(ap);
	if (ctx) ctx->err = buf; else g_create_error = buf;
	return code;
}

#define V2M_HIP_TRY(ctx, expr)                                                              \
	do {                                                                                    \
		hipError_t const st_ = (expr);                                                      \
		if (hipSuccess != st_) {                                                            \
			int const code_ = (hipErrorOutOfMemory == st_) ? V2M_ERR_OUT_OF_MEMORY : V2M_ERR_HIP; \
			return fail(ctx, code_, "%s failed: %s", #expr, hipGetErrorString(st_));       \
		}                                                                                   \
	} while (0)


// Brackets a launch with events when profiling is on.
struct timed_launch {
	v2m_ctx *ctx;
	int kernel;
	event_pair ev{};
	bool active{};

	timed_launch(v2m_ctx *c, int k) : ctx(c), kernel(k)
	{
		if (!ctx->profiling) return;
		if (!ctx->free_events.empty()) { ev = ctx->free_events.back(); ctx->free_events.pop_back(); }
		else if (hipSuccess != hipEventCreate(&ev.begin) || hipSuccess != hipEventCreate(&ev.end)) return;
		active = (hipSuccess == hipEventRecord(ev.begin, ctx->stream));
	}
	~timed_launch()
	{
		if (!active) return;
		(void) hipEventRecord(ev.end, ctx->stream);
		ctx->events[kernel].push_back(ev);
	}
};


template <typename T>
int upload_vec(v2m_ctx *ctx, dev_buf &dst, std::vector<T> const &src, size_t min_bytes = 16)
{
	size_t const bytes(std::max(src.size() * sizeof(T), min_bytes));
	V2M_HIP_TRY(ctx, dst.ensure(bytes));
	if (!src.empty())
		V2M_HIP_TRY(ctx, hipMemcpyAsync(dst.p, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
	return V2M_OK;
}


// Events that destroy themselves, and a guard that puts ctx->profiling back: the calibration paths below return early
// on any HIP error.
struct scoped_events {
	std::vector<hipEvent_t> ev;
	~scoped_events() { for (auto e : ev) (void) hipEventDestroy(e); }
	hipError_t create(std::size_t n)
	{
		for (std::size_t i(0); i < n; ++i) {
			hipEvent_t e{};
			hipError_t const st(hipEventCreate(&e));
			if (hipSuccess != st) return st;
			ev.push_back(e);
		}
		return hipSuccess;
	}
	hipEvent_t operator[](std::size_t i) const { return ev[i]; }
};

struct scoped_profiling_off {
	v2m_ctx *ctx;
	bool was;
	explicit scoped_profiling_off(v2m_ctx *c) : ctx(c), was(c->profiling) { c->profiling = false; }
	~scoped_profiling_off() { ctx->profiling = was; }
};


// 1-D grids in XCD chunks (kernels.hpp: xcd_chunked_item); xcd = false keeps the plain dispatch order (tuning A/B).
struct xcd_grid { unsigned blocks; u32 items_per_xcd; };

bool make_xcd_grid(u64 n_items, bool xcd, xcd_grid &g)
{
	u64 const per((n_items + 7) / 8), blocks(xcd ? per * 8 : n_items);
	if (0 == n_items || blocks > 0x7FFFFFFFull) return false;
	g.blocks = unsigned(blocks);
	g.items_per_xcd = xcd ? u32(per) : 0u;
	return true;
}

// Which dimension of the panel grid runs fastest in item order: 0 = the shorter one (both kinds of neighbours stay close
// in time), 1 = row panels, 2 = column panels / spans.
u32 rows_fastest_for(int order, u64 n_row_panels, u64 n_col_panels)
{
	if (1 == order) return 1;
	if (2 == order) return 0;
	return n_row_panels <= n_col_panels ? 1u : 0u;
}

template <int kR, int kC>
int launch_transpose_shape(v2m_ctx *ctx, u64 const *d_src, u64 SW, u64 DW, u64 SP, u64 DP, u64 *d_dst, bool xcd, int order)
{
	u64 const P((SW + kR - 1) / kR), Q((DW + kC - 1) / kC);
	xcd_grid g;
	if (P > 0xFFFFFFFFull || Q > 0xFFFFFFFFull || !make_xcd_grid(P * Q, xcd, g))
		return fail(ctx, V2M_ERR_UNSUPPORTED, "matrix too large for one transpose launch (%llu x %llu bits)", (unsigned long long) (SW * 64), (unsigned long long) (DW * 64));
	{
		timed_launch tl(ctx, V2M_KERNEL_TRANSPOSE);
		hipLaunchKernelGGL((v2m::transpose_bits_kernel<kR, kC>), dim3(g.blocks), dim3(v2m::kTrThreads), 0, ctx->stream, d_src, d_dst, SW, DW, SP, DP, u32(P), u32(Q), g.items_per_xcd, rows_fastest_for(order, P, Q));
	}
	V2M_HIP_TRY(ctx, hipGetLastError());
	return V2M_OK;
}

template <int kDepth, int kSlabCols, int kWaves = 4, int kTsR = 16, int kTsC = 16, bool kLean = true>
int launch_transpose_stream(v2m_ctx *ctx, u64 const *d_src, u64 SW, u64 DW, u64 SP, u64 DP, u64 *d_dst, bool xcd, int order)
{
	u64 const P((SW + kTsR - 1) / kTsR), Q((DW + kTsC - 1) / kTsC);
	xcd_grid g;
	if (P > 0xFFFFFFFFull || Q > 0xFFFFFFFFull || !make_xcd_grid(P * Q, xcd, g)) return fail(ctx, V2M_ERR_UNSUPPORTED, "matrix too large for one transpose launch");
	{
		timed_launch tl(ctx, V2M_KERNEL_TRANSPOSE);
		hipLaunchKernelGGL((v2m::transpose_bits_stream_kernel<kDepth, kSlabCols, kWaves, kTsR, kTsC, kLean>), dim3(g.blocks), dim3(64 * kWaves), 0, ctx->stream, d_src, d_dst, SW, DW, SP, DP, u32(P), u32(Q), g.items_per_xcd, rows_fastest_for(order, P, Q));
	}
	V2M_HIP_TRY(ctx, hipGetLastError());
	return V2M_OK;
}

// The sector-aligned streaming kernel: kR row-words per workgroup on kW waves, kS-word sectors, kD steps of prefetch,
// spans of `span_groups` column groups (0 = default).
template <int kR, int kW, int kS, int kD, bool kFast, bool kNT>
int launch_transpose_ring(v2m_ctx *ctx, u64 const *d_src, u64 SW, u64 DW, u64 SP, u64 DP, u64 *d_dst, u64 span_groups, bool xcd, int order)
{
	if (0 == span_groups) span_groups = 64;
	span_groups = (span_groups + kS - 1) / kS * kS;
	u64 const P((SW + kR - 1) / kR), NS((DW + span_groups - 1) / span_groups);
	xcd_grid g;
	if (P > 0xFFFFFFFFull || NS > 0xFFFFFFFFull || span_groups > 0x7FFFFFFFull || DW > 0xFFFFFFFFull || !make_xcd_grid(P * NS, xcd, g))
		return fail(ctx, V2M_ERR_UNSUPPORTED, "matrix too large for one transpose launch");
	{
		timed_launch tl(ctx, V2M_KERNEL_TRANSPOSE);
		hipLaunchKernelGGL((v2m::transpose_bits_ring_kernel<kR, kW, kS, kD, kFast, kNT>), dim3(g.blocks), dim3(64 * kW), 0, ctx->stream,
			d_src, d_dst, SW, DW, SP, DP, u32(P), u32(NS), u32(span_groups), g.items_per_xcd, rows_fastest_for(order, P, NS));
	}
	V2M_HIP_TRY(ctx, hipGetLastError());
	return V2M_OK;
}

// The whole-line streaming kernel: kTsR row-words per workgroup on kWaves waves, spans of `span_blocks` blocks of 16 column groups.
// span_blocks = 0: chosen here.  Long spans leave fewer lines written in two pieces, short ones more workgroups to balance: a
// whole destination column per workgroup where columns are short and there are plenty of panels (the inverse direction of a
// path matrix: 5 blocks at config 3, 20 at config 5), otherwise the longest of 32 / 16 / 8 / 4 blocks that still leaves
// 1024 workgroups = two rounds over the chip (measured, config 3 forward with its 10 panels: 8 or 10 blocks 0.295-0.306 ms,
// 4-7 blocks 0.300-0.322, 12 and more 0.334-0.347; config 5 forward: 16-32).
template <int kWaves, int kDepth, int kTsR = 8, int kSlabRows = 32, bool kMayMerge = false>
int launch_transpose_lines(v2m_ctx *ctx, u64 const *d_src, u64 SW, u64 DW, u64 SP, u64 DP, u64 *d_dst, u64 span_blocks, bool xcd, int order)
{
	u64 const P((SW + kTsR - 1) / kTsR), NB((DW + 15) / 16);
	if (0 == span_blocks && 16 == kTsR) {
		// lines16 (two tiles per wave: 185-253 VGPRs, ONE workgroup per CU): the grid is cut for whole rounds over the chip.  Among the span
		// counts that leave at least 4 blocks per span, the one with the best (share of the CUs busy in the last round) x (share of the lines a
		// span writes whole: K / (K + 1)); fewer, longer spans on a tie.  Config 3 forward (5 panels x 977 blocks): 49 spans of 20 blocks =
		// 245 workgroups, one round (measured: 20 blocks 0.263-0.286 ms, 10 blocks 0.276-0.302, 8 / 13 / 16 blocks -- 2.4, 1.5, 1.2 rounds --
		// 0.32-0.38); config 5 forward (20 panels x 6093 blocks): 51 spans of 120; whole columns when there are more panels than CUs.
		u64 const n_cus(std::max<u32>(1, ctx->n_cus));
		u64 const max_spans(std::max<u64>(1, std::min<u64>(NB / 4, 4 * n_cus / std::max<u64>(1, P) + 1)));
		double best(-1);
		for (u64 ns(1); ns <= max_spans; ++ns) {
			u64 const k((NB + ns - 1) / ns), spans((NB + k - 1) / k), wgs(P * spans);
			double const score(double(wgs) / double((wgs + n_cus - 1) / n_cus * n_cus) * double(k) / double(k + 1));
			if (score > best * 1.005) { best = score; span_blocks = k; }
		}
	}
	if (0 == span_blocks) {
		if (NB <= 32 && P >= 1024) span_blocks = NB;
		else {
			span_blocks = 32;
			while (span_blocks > 4 && P * ((NB + span_blocks - 1) / span_blocks) < 1024) span_blocks /= 2;
		}
	}
	u64 const NS((NB + span_blocks - 1) / span_blocks);
	xcd_grid g;
	if (P > 0xFFFFFFFFull || NS > 0xFFFFFFFFull || span_blocks > 0xFFFFull || SP >= (u64(1) << 23) || 4 * DP + DW >= (u64(1) << 29) || !make_xcd_grid(P * NS, xcd, g))
		return fail(ctx, V2M_ERR_UNSUPPORTED, "matrix too large for one transpose launch");
	{
		timed_launch tl(ctx, V2M_KERNEL_TRANSPOSE);
		u32 const panel_fastest(1 == order ? 1u : 2 == order ? 0u : 1u);
		// whole SHORT columns of a dense destination whose columns do not start on lines: every line written once, by the column it
		// begins in.  (Short: the column ends are 2 of 6 lines at config 3's 79 words, where this is worth 9 %; at config 5's 313
		// words they are 2 of 21 and the 34 registers the kernel needs for it cost more than they bring.)
		if (kMayMerge && 1 == NS && NB <= 8 && DP == DW && DW >= 16 && 0 != DP % 16)
			hipLaunchKernelGGL((v2m::transpose_bits_lines_kernel<kWaves, kDepth, kTsR, kSlabRows, kMayMerge>), dim3(g.blocks), dim3(64 * kWaves), 0, ctx->stream,
				d_src, d_dst, SW, DW, SP, DP, u32(P), u32(NS), u32(span_blocks), g.items_per_xcd, panel_fastest);
		else
			hipLaunchKernelGGL((v2m::transpose_bits_lines_kernel<kWaves, kDepth, kTsR, kSlabRows, false>), dim3(g.blocks), dim3(64 * kWaves), 0, ctx->stream,
				d_src, d_dst, SW, DW, SP, DP, u32(P), u32(NS), u32(span_blocks), g.items_per_xcd, panel_fastest);
	}
	V2M_HIP_TRY(ctx, hipGetLastError());
	return V2M_OK;
}

#ifdef V2M_TUNING_BUILD
// The rotating-line kernel (tuning build only: it lost): kTsR row-words per workgroup on kWaves waves, spans of `span_blocks` blocks of 16 column groups (0: chosen here:
// the longest of 32 / 16 / 8 / 4 blocks that still leaves 2048 workgroups -- with ~80 VGPRs three workgroups of 8 waves share a CU).
template <int kWaves, int kDepth, int kTsR>
int launch_transpose_rot(v2m_ctx *ctx, u64 const *d_src, u64 SW, u64 DW, u64 SP, u64 DP, u64 *d_dst, u64 span_blocks, bool xcd, int order)
{
	u64 const P((SW + kTsR - 1) / kTsR), NB((DW + 15) / 16);
	if (0 == span_blocks) {
		span_blocks = 32;
		while (span_blocks > 4 && P * ((NB + span_blocks - 1) / span_blocks) < 2048) span_blocks /= 2;
	}
	u64 const NS((NB + span_blocks - 1) / span_blocks);
	xcd_grid g;
	if (P > 0xFFFFFFFFull || NS > 0xFFFFFFFFull || span_blocks > 0xFFFFull || SP >= (u64(1) << 23) || DW >= (u64(1) << 28) || !make_xcd_grid(P * NS, xcd, g))
		return fail(ctx, V2M_ERR_UNSUPPORTED, "matrix too large for one transpose launch");
	{
		timed_launch tl(ctx, V2M_KERNEL_TRANSPOSE);
		u32 const panel_fastest(1 == order ? 1u : 2 == order ? 0u : 1u);
		hipLaunchKernelGGL((v2m::transpose_bits_rot_kernel<kWaves, kDepth, kTsR>), dim3(g.blocks), dim3(64 * kWaves), 0, ctx->stream,
			d_src, d_dst, SW, DW, SP, DP, u32(P), u32(NS), u32(span_blocks), g.items_per_xcd, panel_fastest);
	}
	V2M_HIP_TRY(ctx, hipGetLastError());
	return V2M_OK;
}
#endif   // V2M_TUNING_BUILD

// Kernel names: "8x8", "4x16", ... (LDS panel kR x kC), "stream16", "lines8[:K]" / "lines16[:K]" (whole lines, 8 / 16 row-words per workgroup, spans of K blocks), "ring:R,W,S,D[,K[,slow|nt]]" (tuning build; slow = ds_bpermute
// butterfly, nt = nontemporal loads and stores); trailing "/rr" keeps the plain round-robin dispatch order instead of XCD chunks, "/pf" / "/sf" make the row
// panels / the column panels (spans) run fastest in item order instead of the shorter dimension.
#ifdef V2M_TUNING_BUILD
constexpr bool kTuningBuild = true;
#else
constexpr bool kTuningBuild = false;
#endif

int launch_transpose_named(v2m_ctx *ctx, std::string shape, u64 const *d_src, u64 SW, u64 DW, u64 SP, u64 DP, u64 *d_dst)
{
	bool xcd(true);
	int order(0);
	for (bool again(true); again && shape.size() > 3;) {
		std::string const tail(shape.substr(shape.size() - 3));
		again = true;
		if (tail == "/rr") xcd = false;
		else if (tail == "/pf") order = 1;
		else if (tail == "/sf") order = 2;
		else again = false;
		if (again) shape.resize(shape.size() - 3);
	}
	if (0 == shape.compare(0, 5, "ring:")) {
		int R(0), W(0), S(0), D(0), K(0);
		char tail[16] = "";
		int const got(std::sscanf(shape.c_str() + 5, "%d,%d,%d,%d,%d,%15s", &R, &W, &S, &D, &K, tail));
		if (got < 4) return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "bad transpose kernel name '%s'", shape.c_str());
		[[maybe_unused]] bool const fast(0 != std::strcmp(tail, "slow")), nt(0 == std::strcmp(tail, "nt"));
		// The product build holds the kernels the library picks among (kTransposeCandidates): 8x8, stream16, lines8 and lines16.  The other
		// shapes and flavours measured on the way there (tools/tune_transpose.py) are compiled with -DV2M_TUNING_BUILD
		// only (vcf2multialign_amd/libv2m_hip_tuning.so, loaded with V2M_HIP_LIBRARY by the tuning tool and the variant tests).
#define V2M_RING_FLAVOUR(r, w, s, d, f, n) launch_transpose_ring<r, w, s, d, f, n>(ctx, d_src, SW, DW, SP, DP, d_dst, u64(K), xcd, order)
#ifdef V2M_TUNING_BUILD
#define V2M_RING(r, w, s, d)                                                                                         \
		if (R == r && W == w && S == s && D == d)                                                                    \
			return !fast ? V2M_RING_FLAVOUR(r, w, s, d, false, false) : nt ? V2M_RING_FLAVOUR(r, w, s, d, true, true) : V2M_RING_FLAVOUR(r, w, s, d, true, false);
		V2M_RING(16, 8, 8, 4) V2M_RING(16, 16, 8, 8) V2M_RING(16, 8, 4, 4) V2M_RING(16, 8, 16, 4)
		V2M_RING(8, 4, 8, 4) V2M_RING(8, 4, 8, 8) V2M_RING(8, 8, 8, 4) V2M_RING(8, 8, 8, 8) V2M_RING(8, 8, 8, 16)
		V2M_RING(16, 16, 16, 8) V2M_RING(8, 8, 16, 8)   // (round 5: whole-line sectors with the lean butterfly; profiles/r05/transpose_rot8_experiment.txt)
#undef V2M_RING
#endif
#undef V2M_RING_FLAVOUR
		return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "transpose kernel '%s' is not in this build%s", shape.c_str(), kTuningBuild ? "" : " (the product build has 8x8, stream16, lines8 and lines16; the rest needs -DV2M_TUNING_BUILD)");
	}
	if (shape == "stream16") return launch_transpose_stream<4, 64>(ctx, d_src, SW, DW, SP, DP, d_dst, xcd, order);
#ifdef V2M_TUNING_BUILD
	if (0 == shape.compare(0, 4, "rot8")) {
		// "rot8[:K[,V]]": spans of K blocks (0 / absent = chosen per shape); V = geometry variant
		int K(0), V(8);
		if (shape.size() > 4 && (':' != shape[4] || std::sscanf(shape.c_str() + 5, "%d,%d", &K, &V) < 1 || K < 0)) return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "bad transpose kernel name '%s'", shape.c_str());
		if (8 == V) return launch_transpose_rot<8, 4, 8>(ctx, d_src, SW, DW, SP, DP, d_dst, u64(K), xcd, order);
		if (88 == V) return launch_transpose_rot<8, 8, 8>(ctx, d_src, SW, DW, SP, DP, d_dst, u64(K), xcd, order);            // 8 steps of prefetch
		if (4 == V) return launch_transpose_rot<4, 4, 8>(ctx, d_src, SW, DW, SP, DP, d_dst, u64(K), xcd, order);             // 4 waves, two tiles each
		if (16 == V) return launch_transpose_rot<8, 4, 16>(ctx, d_src, SW, DW, SP, DP, d_dst, u64(K), xcd, order);           // 16 row-words (128-B source runs), two tiles per wave
		if (1616 == V) return launch_transpose_rot<16, 4, 16>(ctx, d_src, SW, DW, SP, DP, d_dst, u64(K), xcd, order);        // 16 row-words on 16 waves
		return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "bad transpose kernel name '%s'", shape.c_str());
	}
#endif
	if (0 == shape.compare(0, 7, "lines16")) {
		// "lines16[:K]": the whole-line kernel with 16 row-words per workgroup (128-byte source runs) on 8 waves, two tiles each; spans of K blocks (0 / absent = chosen per shape)
		int K(0);
		if (shape.size() > 7 && (':' != shape[7] || std::sscanf(shape.c_str() + 8, "%d", &K) < 1 || K < 0)) return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "bad transpose kernel name '%s'", shape.c_str());
		return launch_transpose_lines<8, 4, 16, 32, true>(ctx, d_src, SW, DW, SP, DP, d_dst, u64(K), xcd, order);
	}
	if (0 == shape.compare(0, 6, "lines8")) {
		// "lines8[:K]": spans of K blocks (0 / absent = chosen per shape); the tuning build also has "lines8:K,V" with V = another geometry
		int K(0), V(8);
		if (shape.size() > 6 && (':' != shape[6] || std::sscanf(shape.c_str() + 7, "%d,%d", &K, &V) < 1 || K < 0)) return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "bad transpose kernel name '%s'", shape.c_str());
		if (8 == V) return launch_transpose_lines<8, 4, 8, 32, true>(ctx, d_src, SW, DW, SP, DP, d_dst, u64(K), xcd, order);
#ifdef V2M_TUNING_BUILD
		if (4 == V) return launch_transpose_lines<4, 4>(ctx, d_src, SW, DW, SP, DP, d_dst, u64(K), xcd, order);                // 4 waves, two tiles each
		if (88 == V) return launch_transpose_lines<8, 8>(ctx, d_src, SW, DW, SP, DP, d_dst, u64(K), xcd, order);               // 8 steps of prefetch
		if (816 == V) return launch_transpose_lines<8, 4, 8, 16>(ctx, d_src, SW, DW, SP, DP, d_dst, u64(K), xcd, order);       // 16-column slab
		if (16 == V) return launch_transpose_lines<16, 4, 16, 16>(ctx, d_src, SW, DW, SP, DP, d_dst, u64(K), xcd, order);      // 16 row-words on 16 waves
		if (168 == V) return launch_transpose_lines<16, 8, 16, 16>(ctx, d_src, SW, DW, SP, DP, d_dst, u64(K), xcd, order);
		if (1 == V) return launch_transpose_lines<8, 4>(ctx, d_src, SW, DW, SP, DP, d_dst, u64(K), xcd, order);                // the product's geometry without the merged column ends
		if (28 == V) return launch_transpose_lines<8, 4, 16, 32>(ctx, d_src, SW, DW, SP, DP, d_dst, u64(K), xcd, order);       // 16 row-words (128-B source runs) on 8 waves, two tiles each (round 5, late: profiles/r05/transpose_pmc.txt)
		if (281 == V) return launch_transpose_lines<8, 4, 16, 32, true>(ctx, d_src, SW, DW, SP, DP, d_dst, u64(K), xcd, order);  // the same, merged column ends where they apply
		if (288 == V) return launch_transpose_lines<8, 8, 16, 32>(ctx, d_src, SW, DW, SP, DP, d_dst, u64(K), xcd, order);      // the same with 8 steps of prefetch
		if (282 == V) return launch_transpose_lines<8, 2, 16, 32>(ctx, d_src, SW, DW, SP, DP, d_dst, u64(K), xcd, order);      // the same with 2 steps of prefetch
		if (2816 == V) return launch_transpose_lines<8, 4, 16, 16>(ctx, d_src, SW, DW, SP, DP, d_dst, u64(K), xcd, order);     // the same with a 16-column slab
#endif
		return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "transpose kernel '%s' is not in this build%s", shape.c_str(), kTuningBuild ? "" : " (needs -DV2M_TUNING_BUILD)");
	}
#ifdef V2M_TUNING_BUILD
	if (shape == "stream16:old") return launch_transpose_stream<4, 64, 4, 16, 16, false>(ctx, d_src, SW, DW, SP, DP, d_dst, xcd, order);
	if (shape == "stream16:2,64") return launch_transpose_stream<2, 64>(ctx, d_src, SW, DW, SP, DP, d_dst, xcd, order);
	if (shape == "stream16:4,32") return launch_transpose_stream<4, 32>(ctx, d_src, SW, DW, SP, DP, d_dst, xcd, order);
	if (shape == "stream16:2,32") return launch_transpose_stream<2, 32>(ctx, d_src, SW, DW, SP, DP, d_dst, xcd, order);
	if (shape == "stream16:3,32") return launch_transpose_stream<3, 32>(ctx, d_src, SW, DW, SP, DP, d_dst, xcd, order);
	if (shape == "stream16:2,16") return launch_transpose_stream<2, 16>(ctx, d_src, SW, DW, SP, DP, d_dst, xcd, order);
	if (shape == "stream16:4,32,8") return launch_transpose_stream<4, 32, 8>(ctx, d_src, SW, DW, SP, DP, d_dst, xcd, order);
	if (shape == "stream16:4,16,8") return launch_transpose_stream<4, 16, 8>(ctx, d_src, SW, DW, SP, DP, d_dst, xcd, order);
	if (shape == "stream16:2,16,8") return launch_transpose_stream<2, 16, 8>(ctx, d_src, SW, DW, SP, DP, d_dst, xcd, order);
	if (shape == "stream16:4,64,8") return launch_transpose_stream<4, 64, 8>(ctx, d_src, SW, DW, SP, DP, d_dst, xcd, order);
	if (shape == "stream8x32") return launch_transpose_stream<4, 32, 4, 8, 32>(ctx, d_src, SW, DW, SP, DP, d_dst, xcd, order);
	if (shape == "stream8x32:8") return launch_transpose_stream<4, 16, 8, 8, 32>(ctx, d_src, SW, DW, SP, DP, d_dst, xcd, order);
	if (shape == "stream32x8") return launch_transpose_stream<4, 64, 4, 32, 8>(ctx, d_src, SW, DW, SP, DP, d_dst, xcd, order);
	if (shape == "stream4x64") return launch_transpose_stream<4, 16, 4, 4, 64>(ctx, d_src, SW, DW, SP, DP, d_dst, xcd, order);
#endif
	if (shape == "8x8") return launch_transpose_shape<8, 8>(ctx, d_src, SW, DW, SP, DP, d_dst, xcd, order);
#ifdef V2M_TUNING_BUILD
	if (shape == "4x16") return launch_transpose_shape<4, 16>(ctx, d_src, SW, DW, SP, DP, d_dst, xcd, order);
	if (shape == "16x8") return launch_transpose_shape<16, 8>(ctx, d_src, SW, DW, SP, DP, d_dst, xcd, order);
	if (shape == "16x4") return launch_transpose_shape<16, 4>(ctx, d_src, SW, DW, SP, DP, d_dst, xcd, order);
	if (shape == "4x8") return launch_transpose_shape<4, 8>(ctx, d_src, SW, DW, SP, DP, d_dst, xcd, order);
	if (shape == "8x4") return launch_transpose_shape<8, 4>(ctx, d_src, SW, DW, SP, DP, d_dst, xcd, order);
	if (shape == "8x16") return launch_transpose_shape<8, 16>(ctx, d_src, SW, DW, SP, DP, d_dst, xcd, order);
#endif
	return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "unknown transpose kernel '%s'", shape.c_str());
}

// Several kernels implement the transpose; which is fastest depends on the matrix shape, so matrices of at least 32 MiB
// are timed once per shape and context with each candidate (the result is the same either way) and the fastest is
// remembered.  V2M_TRANSPOSE_PANEL forces one; V2M_TRANSPOSE_CANDIDATES (comma-free list separated by ';') replaces the list.
char const *const kTransposeCandidates[] = {"8x8", "stream16", "lines8", "lines16"};

// src_pitch / dst_pitch: words from one column to the next (0 = dense: n_rows / 64 and n_cols / 64).
int launch_transpose(v2m_ctx *ctx, u64 const *d_src, u64 n_rows, u64 n_cols, u64 *d_dst, u64 src_pitch = 0, u64 dst_pitch = 0)
{
	u64 const SW(n_rows / 64), DW(n_cols / 64);
	u64 const SP(src_pitch ? src_pitch : SW), DP(dst_pitch ? dst_pitch : DW);
	char const *e(std::getenv("V2M_TRANSPOSE_PANEL"));
	if (e && *e) return launch_transpose_named(ctx, e, d_src, SW, DW, SP, DP, d_dst);
	if (SW * DW * 512 < (u64(32) << 20)) return launch_transpose_named(ctx, "8x8", d_src, SW, DW, SP, DP, d_dst);
	for (auto const &c : ctx->transpose_choice)
		if (c.rows == n_rows && c.cols == n_cols && c.src_pitch == SP && c.dst_pitch == DP) return launch_transpose_named(ctx, c.kernel, d_src, SW, DW, SP, DP, d_dst);

	std::vector<std::string> names;
	if (char const *list = std::getenv("V2M_TRANSPOSE_CANDIDATES")) {
		std::string cur;
		for (char const *p(list); ; ++p) {
			if (';' == *p || 0 == *p) { if (!cur.empty()) names.push_back(cur); cur.clear(); if (0 == *p) break; }
			else cur.push_back(*p);
		}
	}
	if (names.empty()) names.assign(std::begin(kTransposeCandidates), std::end(kTransposeCandidates));

	std::vector<float> best(names.size(), 1e30f);
	{
		scoped_events ev;
		V2M_HIP_TRY(ctx, ev.create(names.size() + 1));
		scoped_profiling_off const quiet(ctx);   // the calibration launches are not the caller's
		std::vector<char> usable(names.size(), 1);   // a candidate may decline a shape (its index arithmetic is 32-bit): it is left out, not an error
		for (int rep(0); rep < 3; ++rep) {        // the first round touches the pages; the better of the next two counts (one timing alone picked the wrong kernel now and then: the candidates are within 10 % of each other and a launch's time depends on what the one before it left in the caches)
			V2M_HIP_TRY(ctx, hipEventRecord(ev[0], ctx->stream));
			for (std::size_t k(0); k < names.size(); ++k) {
				if (usable[k]) {
					int const rc(launch_transpose_named(ctx, names[k], d_src, SW, DW, SP, DP, d_dst));
					if (V2M_ERR_UNSUPPORTED == rc) usable[k] = 0;
					else if (rc) return rc;
				}
				V2M_HIP_TRY(ctx, hipEventRecord(ev[k + 1], ctx->stream));
			}
			V2M_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
			for (std::size_t k(0); k < names.size(); ++k) {
				float ms(0);
				V2M_HIP_TRY(ctx, hipEventElapsedTime(&ms, ev[k], ev[k + 1]));
				if (rep && usable[k]) best[k] = std::min(best[k], ms);
			}
		}
		if (std::none_of(usable.begin(), usable.end(), [](char u) { return 0 != u; })) return fail(ctx, V2M_ERR_UNSUPPORTED, "matrix too large for one transpose launch");
		if (std::any_of(usable.begin(), usable.end(), [](char u) { return 0 == u; })) ctx->err.clear();   // (the declining candidate's message)
	}
	std::size_t const pick(std::size_t(std::min_element(best.begin(), best.end()) - best.begin()));
	ctx->transpose_choice.push_back({n_rows, n_cols, SP, DP, names[pick]});
	std::string note("transpose " + std::to_string(n_rows) + "x" + std::to_string(n_cols) + " bits" + ((SP != SW || DP != DW) ? " (column pitches " + std::to_string(SP) + " / " + std::to_string(DP) + " words)" : std::string()) + ": " + names[pick] + " (");
	for (std::size_t k(0); k < names.size(); ++k) {
		char buf[64];
		std::snprintf(buf, sizeof(buf), "%s%s %.3f ms", k ? ", " : "", names[k].c_str(), best[k]);
		note += buf;
	}
	note += ")";
	if (ctx->info.size() > 2000) ctx->info.clear();
	if (!ctx->info.empty()) ctx->info += "; ";
	ctx->info += note;
	// the calibration already produced the result; run the chosen kernel once more under the caller's profiling so
	// that its launch is accounted for like any other
	return launch_transpose_named(ctx, names[pick], d_src, SW, DW, SP, DP, d_dst);
}


// Host-side preparation of a row batch: validates it against the uploaded graph and flattens
// every row into (edge_begin, copy) segments.  The tables are written straight into the context's pinned
// staging area: founder batches carry hundreds of thousands of segments per row, and uploading those from pageable
// memory makes the runtime pin and unpin the pages as userptr memory, which stalls the queue for tens of ms.
struct prepared_rows {
	u32 *seg_offsets{};      // [n_rows + 1]
	u32 *seg_edge_begin{};   // [n_segments]
	u32 *seg_copy{};         // [n_segments]
	u64 n_rows{}, n_segments{};
	bool any_switching_row{};
};

int prepare_rows(v2m_ctx *ctx, v2m_row_batch const *rows, u64 row_begin, u64 row_end, prepared_rows &out)
{
	u64 const n_rows(row_end - row_begin);
	u64 max_segments(n_rows);                                 // a row without cuts is one segment ...
	if (rows->cut_offsets) {
		for (u64 r(row_begin); r < row_end; ++r) {
			if (rows->cut_offsets[r + 1] < rows->cut_offsets[r])
				return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "cut_offsets decrease at row %llu", (unsigned long long) r);
			max_segments += rows->cut_offsets[r + 1] - rows->cut_offsets[r];   // ... one with cuts has one per cut, plus a leading REF one
		}
	}
	if (max_segments >= 0xFFFFFFFFull)
		return fail(ctx, V2M_ERR_UNSUPPORTED, "row batch has too many cut segments (%llu) for one call", (unsigned long long) max_segments);
	auto const pad([](u64 n) { return (n * sizeof(u32) + 63) & ~u64(63); });
	int const area(ctx->row_stage_next);
	if (ctx->row_stage_in_flight[area]) {
		V2M_HIP_TRY(ctx, hipEventSynchronize(ctx->ev_row_stage[area]));   // the uploads of two slices ago
		ctx->row_stage_in_flight[area] = false;
	}
	V2M_HIP_TRY(ctx, ctx->h_row_stage[area].ensure(pad(n_rows + 1) + 2 * pad(max_segments)));
	char *const base(static_cast<char *>(ctx->h_row_stage[area].p));
	out.seg_offsets = reinterpret_cast<u32 *>(base);
	out.seg_edge_begin = reinterpret_cast<u32 *>(base + pad(n_rows + 1));
	out.seg_copy = reinterpret_cast<u32 *>(base + pad(n_rows + 1) + pad(max_segments));
	out.n_rows = n_rows;
	out.any_switching_row = false;

	// segment offsets first (a row with cuts: one segment per cut, plus a leading REF one unless its first cut is node 0), then the
	// rows are filled independently of each other: founder batches carry hundreds of thousands of cuts per row (config 4: 672 495 x 26
	// rows, each cut checked against the graph), which a few threads do in a quarter of the time
	out.seg_offsets[0] = 0;
	for (u64 r(row_begin); r < row_end; ++r) {
		u64 const c_begin(rows->cut_offsets ? rows->cut_offsets[r] : 0), c_end(rows->cut_offsets ? rows->cut_offsets[r + 1] : 0);
		u64 const n_seg(c_begin == c_end ? 1 : (c_end - c_begin) + (0 != rows->cut_nodes[c_begin] ? 1 : 0));
		out.seg_offsets[r - row_begin + 1] = u32(out.seg_offsets[r - row_begin] + n_seg);
		out.any_switching_row = out.any_switching_row || n_seg > 1;
	}
	out.n_segments = out.seg_offsets[n_rows];

	// (error messages are composed where the error is found; the first failing row in row order is the one reported)
	struct row_error { int code{V2M_OK}; std::string text; };
	auto const fill_row([&](u64 r, row_error &err) {
		auto const failed([&](int code, char const *fmt, unsigned long long a, unsigned long long b = 0) {
			char buf[256];
			std::snprintf(buf, sizeof(buf), fmt, a, b);
			err.code = code;
			err.text = buf;
		});
		u64 const c_begin(rows->cut_offsets ? rows->cut_offsets[r] : 0), c_end(rows->cut_offsets ? rows->cut_offsets[r + 1] : 0);
		u32 n(out.seg_offsets[r - row_begin]);
		if (c_begin == c_end) {
			if (!rows->copy_index) return failed(V2M_ERR_INVALID_ARGUMENT, "row %llu has no cuts and rows->copy_index is NULL", r);
			u32 const copy(rows->copy_index[r]);
			if (copy != V2M_PLOIDY_MAX && (!ctx->d_paths || copy >= ctx->path_cols))
				return failed(V2M_ERR_INVALID_ARGUMENT, "row %llu: chromosome copy %llu is outside the path matrix", r, copy);
			out.seg_edge_begin[n] = 0;
			out.seg_copy[n] = copy;
			return;
		}
		u64 prev(0);
		for (u64 k(c_begin); k < c_end; ++k) {
			u64 const node(rows->cut_nodes[k]);
			u32 const copy(rows->cut_copies[k]);
			if (node >= ctx->n_nodes) return failed(V2M_ERR_PRECONDITION, "row %llu: cut node %llu does not exist", r, node);
			if (k > c_begin && node <= prev) return failed(V2M_ERR_PRECONDITION, "row %llu: cut nodes must be strictly increasing", r);
			if (copy != V2M_PLOIDY_MAX && (!ctx->d_paths || copy >= ctx->path_cols))
				return failed(V2M_ERR_INVALID_ARGUMENT, "row %llu: chromosome copy %llu is outside the path matrix", r, copy);
			u32 const first_edge(ctx->h_csum[node]);
			// founder_sequence_greedy_output.cc:108 asserts the walk never jumps over a cut node
			if (node > 0 && ctx->h_tgt_prefix_max[first_edge] > node)
				return failed(V2M_ERR_PRECONDITION, "row %llu: cut node %llu lies inside the span of an ALT edge", r, node);
			if (k == c_begin && node != 0) {   // copy index is PLOIDY_MAX until the first cut is visited
				out.seg_edge_begin[n] = 0;
				out.seg_copy[n] = V2M_PLOIDY_MAX;
				++n;
			}
			out.seg_edge_begin[n] = first_edge;
			out.seg_copy[n] = copy;
			++n;
			prev = node;
		}
	});
	std::vector<row_error> errors(n_rows);
	unsigned const n_threads(out.n_segments >= (u64(1) << 20) ? unsigned(std::min<u64>(n_rows, 8)) : 1u);
	if (n_threads <= 1) {
		for (u64 r(row_begin); r < row_end; ++r) { fill_row(r, errors[r - row_begin]); if (V2M_OK != errors[r - row_begin].code) break; }
	}
	else {
		std::atomic<u64> next(row_begin);
		auto const work([&] { for (u64 r; (r = next.fetch_add(1)) < row_end;) fill_row(r, errors[r - row_begin]); });
		std::vector<std::thread> pool;
		for (unsigned t(1); t < n_threads; ++t) pool.emplace_back(work);
		work();
		for (auto &t : pool) t.join();
	}
	for (auto const &e : errors) if (V2M_OK != e.code) return fail(ctx, e.code, "%s", e.text.c_str());
	return V2M_OK;
}


// Tuning knobs (read per call so that one process can A/B them).
// V2M_NT_STORES=0/1 forces plain / nontemporal output stores; unset = calibrate once per context (below).
int forced_store_mode()
{
	char const *e = std::getenv("V2M_NT_STORES");
	if (!(e && *e)) return -1;
	return std::atoi(e) != 0 ? 1 : 0;
}

u32 rows_per_group_for(u64 n_rows)
{
	char const *e = std::getenv("V2M_ROWS_PER_GROUP");
	int const v((e && *e) ? std::atoi(e) : 0);
	if (v > 0) return u32(std::min(v, 256));   // count_unaligned_kernel holds at most 256 rows per group
	// 32 rows per workgroup: the template tile and the patch cache are set up once per group (a prologue of dependent L2 / HBM round trips during which
	// the workgroup stores nothing) and the effective-edge words are reloaded every 16 rows (kGroupRowsLds).  Measured per 620 rows of config 3 /
	// 244 of config 5 on one box (profiles/r05/rows_per_group.txt): 16 rows 9.39-9.57 / 9.07-9.35 ms (unaligned 10.65-10.84 / 11.0-11.7), 32 rows
	// 9.06-9.32 / 9.14-9.31 (10.16-10.30 / 10.9-11.6), 48 rows 9.31-9.51 / 9.20-9.42, 64 rows 9.12-9.33 / 9.35-9.44.
	return u32(std::min<u64>(32, std::max<u64>(1, n_rows)));
}


// Words per row of the effective-edge scratch: the edge count in 64-bit words, rounded up to a whole 128-B line so
// that every row starts line-aligned.
u64 eff_row_words(v2m_ctx const *ctx) { return (((ctx->n_edges + 63) / 64) + 15) & ~u64(15); }

// Effective-edge bits of rows [row_begin, row_end) of the batch into ctx->d_eff.
int resolve_slice(v2m_ctx *ctx, v2m_row_batch const *rows, u64 row_begin, u64 row_end)
{
	u64 const n_rows(row_end - row_begin);
	u64 const eff_words(eff_row_words(ctx)), n_words((ctx->n_edges + 63) / 64);
	if (0 == ctx->n_edges) return V2M_OK;

	prepared_rows pr;
	if (int const rc = prepare_rows(ctx, rows, row_begin, row_end, pr)) return rc;
	auto const upload([&](dev_buf &dst, u32 const *src, u64 count) -> int {
		V2M_HIP_TRY(ctx, dst.ensure(std::max<size_t>(count * sizeof(u32), 16)));
		if (count) V2M_HIP_TRY(ctx, hipMemcpyAsync(dst.p, src, count * sizeof(u32), hipMemcpyHostToDevice, ctx->stream));
		return V2M_OK;
	});
	// The staging area the tables came from is free again once these copies have run; the other one takes the next slice.  The
	// area counts as in flight from the first copy on, whether or not all three get queued: a failed call must not leave
	// copies behind that the next user of the area does not wait for.
	int const stage(ctx->row_stage_next);
	ctx->row_stage_in_flight[stage] = true;
	ctx->row_stage_next ^= 1;
	int upload_rc(upload(ctx->d_seg_offsets, pr.seg_offsets, n_rows + 1));
	if (!upload_rc) upload_rc = upload(ctx->d_seg_edge_begin, pr.seg_edge_begin, pr.n_segments);
	if (!upload_rc) upload_rc = upload(ctx->d_seg_copy, pr.seg_copy, pr.n_segments);
	hipError_t const recorded(hipEventRecord(ctx->ev_row_stage[stage], ctx->stream));
	if (upload_rc) return upload_rc;
	V2M_HIP_TRY(ctx, recorded);
	V2M_HIP_TRY(ctx, ctx->d_eff.ensure(n_rows * eff_words * sizeof(u64)));
	V2M_HIP_TRY(ctx, ctx->d_needs_serial.ensure(n_rows * sizeof(u32)));
	V2M_HIP_TRY(ctx, hipMemsetAsync(ctx->d_needs_serial.p, 0, n_rows * sizeof(u32), ctx->stream));

	// rows that switch copies (founder rows) get their bit column put together first
	bool const any_switching_row(pr.any_switching_row);
	if (any_switching_row) V2M_HIP_TRY(ctx, ctx->d_row_bits.ensure(n_rows * eff_words * sizeof(u64)));

	v2m::row_segments rs{ctx->d_seg_offsets.as<u32>(), ctx->d_seg_edge_begin.as<u32>(), ctx->d_seg_copy.as<u32>(),
		any_switching_row ? ctx->d_row_bits.as<u64>() : nullptr, u32(eff_words)};
	char const *const back_env(std::getenv("V2M_MAX_BACK_WORDS"));   // test knob: 0 forces the serial kernel for every cross-word restart
	u32 const max_back_words((back_env && *back_env) ? u32(std::strtoul(back_env, nullptr, 10)) : v2m::kMaxBackWords);
	{
		timed_launch tl(ctx, V2M_KERNEL_RESOLVE);
		// rows per launch: queue entries are 32-bit (row, word) indices
		u64 const rows_per_launch(std::max<u64>(1, std::min<u64>(65535, 0xFFFFFFFFull / std::max<u64>(1, n_words))));
		for (u64 r0(0); r0 < n_rows; r0 += rows_per_launch) {
			u64 const nr(std::min<u64>(rows_per_launch, n_rows - r0));
			if (any_switching_row)
				for (u64 y0(0); y0 < nr; y0 += 65535)   // grid.y limit
					hipLaunchKernelGGL(v2m::assemble_row_bits_kernel, dim3(unsigned((n_words + 255) / 256), unsigned(std::min<u64>(65535, nr - y0))), dim3(256), 0, ctx->stream,
						ctx->d_paths, ctx->path_pitch, rs, ctx->d_row_bits.as<u64>(), u32(n_words), u32(r0 + y0));
			// the queue's segments hold every word of the launch if they have to (iid random bits do that), up to 1 GiB in all;
			// a workgroup whose segment is full decides its overflow itself
			u64 const shard_capacity(std::max<u64>(256, std::min<u64>(nr * n_words, u64(1) << 28) / v2m::kResolveQueueShards));
			V2M_HIP_TRY(ctx, ctx->d_resolve_queue.ensure(shard_capacity * v2m::kResolveQueueShards * sizeof(u32)));
			V2M_HIP_TRY(ctx, ctx->d_resolve_count.ensure(v2m::kResolveQueueShards * sizeof(u32)));
			V2M_HIP_TRY(ctx, hipMemsetAsync(ctx->d_resolve_count.p, 0, v2m::kResolveQueueShards * sizeof(u32), ctx->stream));
			for (u64 piece0(0), pieces((n_words + 256 * v2m::kResolveWordsPerThread - 1) / (256 * v2m::kResolveWordsPerThread)); piece0 < pieces; piece0 += 65535)   // grid.y limit; rows run fastest
				hipLaunchKernelGGL(v2m::resolve_effective_edges_kernel, dim3(unsigned(nr), unsigned(std::min<u64>(65535, pieces - piece0))), dim3(256), 0, ctx->stream,
					ctx->d_paths, ctx->path_pitch, u32(ctx->n_edges), rs, ctx->d_spans.as<v2m::edge_span>(), ctx->d_overlappable.as<u64>(),
					ctx->d_ovl_rank.as<u32>(), ctx->d_blocker_masks.as<u64>(),
					ctx->d_eff.as<u64>(), u32(n_words), u32(eff_words), u32(r0), u32(piece0),
					ctx->d_resolve_queue.as<u32>(), ctx->d_resolve_count.as<u32>(), u32(shard_capacity), ctx->d_needs_serial.as<u32>(), max_back_words);
			hipLaunchKernelGGL(v2m::resolve_queued_words_kernel, dim3(2 * v2m::kResolveQueueShards), dim3(256), 0, ctx->stream,
				ctx->d_paths, ctx->path_pitch, u32(ctx->n_edges), rs, ctx->d_spans.as<v2m::edge_span>(), ctx->d_overlappable.as<u64>(),
				ctx->d_eff.as<u64>(), u32(n_words), u32(eff_words), u32(r0),
				ctx->d_resolve_queue.as<u32>(), ctx->d_resolve_count.as<u32>(), u32(shard_capacity), ctx->d_needs_serial.as<u32>(), max_back_words);
		}
		// rows whose restart point is too far back for the per-word kernel (chromosome-scale deletions)
		hipLaunchKernelGGL(v2m::resolve_rows_serial_kernel, dim3(unsigned((n_rows + 3) / 4)), dim3(256), 0, ctx->stream,
			ctx->d_paths, ctx->path_pitch, u32(ctx->n_edges), rs, ctx->d_spans.as<v2m::edge_span>(),
			ctx->d_eff.as<u64>(), eff_words, u32(n_rows), ctx->d_needs_serial.as<u32>());
	}
	V2M_HIP_TRY(ctx, hipGetLastError());
	return V2M_OK;
}


struct splice_grid {
	u32 rows_per_group, n_groups, tile_run;
	u64 n_blocks;
};

u32 tile_run_for(u32 n_tiles)
{
	char const *e = std::getenv("V2M_TILE_RUN");   // tuning knob
	int const v((e && *e) ? std::atoi(e) : 0);
	u32 const run(v > 0 ? u32(v) : 64u);
	return std::max<u32>(1, std::min(run, n_tiles));
}

int make_grid(v2m_ctx *ctx, u64 n_rows, splice_grid &g)
{
	g.rows_per_group = rows_per_group_for(n_rows);
	g.n_groups = u32((n_rows + g.rows_per_group - 1) / g.rows_per_group);
	g.n_blocks = u64(ctx->n_tiles) * g.n_groups;
	g.tile_run = tile_run_for(ctx->n_tiles);
	if (g.n_blocks > 0x7FFFFFFFull)
		return fail(ctx, V2M_ERR_UNSUPPORTED, "splice grid too large (%llu workgroups); use smaller batches", (unsigned long long) g.n_blocks);
	return V2M_OK;
}


// Resolve + aligned splice of rows [row_begin, row_end) of the batch into d_out.
int splice_aligned_slice(v2m_ctx *ctx, v2m_row_batch const *rows, u64 row_begin, u64 row_end, char *d_out, u64 row_pitch)
{
	u64 const n_rows(row_end - row_begin);
	if (0 == n_rows || 0 == ctx->aligned_len) return V2M_OK;
	if (int const rc = resolve_slice(ctx, rows, row_begin, row_end)) return rc;

	u64 const eff_words(eff_row_words(ctx));
	splice_grid g;
	if (int const rc = make_grid(ctx, n_rows, g)) return rc;
	v2m::tile_tables tt{ctx->d_tile_edge_begin.as<u32>(), ctx->d_cross_offsets.as<u32>(), ctx->d_cross_edges.as<u32>()};
	u64 const store_limit((ctx->aligned_len + 15) & ~u64(15));
	auto launch = [&](bool nt) {
		if (nt)
			hipLaunchKernelGGL(v2m::splice_aligned_kernel<true>, dim3(unsigned(g.n_blocks)), dim3(v2m::kSpliceThreads), 0, ctx->stream,
				ctx->d_template.as<v2m::vec4u>(), ctx->d_eff.as<u64>(), eff_words, tt, ctx->d_patches.as<v2m::edge_patch>(), ctx->d_labels.as<char>(),
				d_out, row_pitch, u32(n_rows), g.rows_per_group, g.n_groups, ctx->n_tiles, g.tile_run, store_limit, '-');
		else
			hipLaunchKernelGGL(v2m::splice_aligned_kernel<false>, dim3(unsigned(g.n_blocks)), dim3(v2m::kSpliceThreads), 0, ctx->stream,
				ctx->d_template.as<v2m::vec4u>(), ctx->d_eff.as<u64>(), eff_words, tt, ctx->d_patches.as<v2m::edge_patch>(), ctx->d_labels.as<char>(),
				d_out, row_pitch, u32(n_rows), g.rows_per_group, g.n_groups, ctx->n_tiles, g.tile_run, store_limit, '-');
	};

	// Output rows are written once and never re-read by the GPU, so nontemporal stores (which keep the
	// rows from displacing the template and edge tables in L2 / Infinity Cache) usually win: 7.2-7.6 ms
	// against 8.1-8.5 ms per 51-GB launch at config 3.  But on some boxes / memory layouts they are stuck
	// in a slower mode for the whole life of a process (9.1-9.3 ms, tools/probe_nt*.py), while plain stores
	// stay put.  The flavour is therefore calibrated once per context, on the first launch that writes
	// >= 1 GiB: that launch is issued twice per flavour (the output is the same every time) and the
	// faster one is kept.
	int mode(forced_store_mode());
	if (mode < 0) mode = ctx->store_mode;
	if (mode < 0 && n_rows * ctx->aligned_len >= (u64(1) << 30)) {
		scoped_events ev;
		V2M_HIP_TRY(ctx, ev.create(5));
		V2M_HIP_TRY(ctx, hipEventRecord(ev[0], ctx->stream));
		for (int i(0); i < 4; ++i) {
			launch(0 == (i & 1));   // nt, plain, nt, plain
			V2M_HIP_TRY(ctx, hipEventRecord(ev[i + 1], ctx->stream));
		}
		V2M_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		float t[4];
		for (int i(0); i < 4; ++i) V2M_HIP_TRY(ctx, hipEventElapsedTime(&t[i], ev[i], ev[i + 1]));
		float const nt_ms(std::min(t[0], t[2])), plain_ms(std::min(t[1], t[3]));
		ctx->store_mode = nt_ms <= plain_ms ? 1 : 0;
		char buf[160];
		std::snprintf(buf, sizeof(buf), "aligned splice stores: %s (calibrated on %llu rows: nontemporal %.3f ms, plain %.3f ms)",
			ctx->store_mode ? "nontemporal" : "plain", (unsigned long long) n_rows, nt_ms, plain_ms);
		if (!ctx->info.empty()) ctx->info += "; ";
		ctx->info += buf;
		mode = ctx->store_mode;
	}
	if (mode < 0) mode = 1;   // small launches before any calibration
	{
		timed_launch tl(ctx, V2M_KERNEL_SPLICE_ALIGNED);
		launch(0 != mode);
	}
	V2M_HIP_TRY(ctx, hipGetLastError());
	return V2M_OK;
}


// Resolve + unaligned splice (count, scan, compact) of rows [row_begin, row_end) into d_out;
// leaves the row lengths in ctx->d_row_lengths.
int splice_unaligned_slice(v2m_ctx *ctx, v2m_row_batch const *rows, u64 row_begin, u64 row_end, char *d_out, u64 row_pitch)
{
	u64 const n_rows(row_end - row_begin);
	if (0 == n_rows) return V2M_OK;
	V2M_HIP_TRY(ctx, ctx->d_row_lengths.ensure(n_rows * sizeof(u64)));
	if (0 == ctx->aligned_len) {
		V2M_HIP_TRY(ctx, hipMemsetAsync(ctx->d_row_lengths.p, 0, n_rows * sizeof(u64), ctx->stream));
		return V2M_OK;
	}
	u64 const n_chunks(u64(ctx->n_tiles) * v2m::kTileChunks);
	if (!ctx->has_template0) {
		V2M_HIP_TRY(ctx, ctx->d_template0.ensure(n_chunks * 16));
		{
			timed_launch tl(ctx, V2M_KERNEL_TEMPLATE);
			hipLaunchKernelGGL(v2m::expand_reference_row_kernel, dim3(unsigned((n_chunks + 255) / 256)), dim3(256), 0, ctx->stream,
				ctx->d_ref.as<char>(), ctx->d_ref_pos.as<u32>(), ctx->d_aln_pos.as<u32>(), u32(ctx->n_nodes), u32(ctx->aligned_len), n_chunks, ctx->d_template0.as<uint4>(), char(0));
		}
		V2M_HIP_TRY(ctx, hipGetLastError());
		ctx->has_template0 = true;
	}
	if (int const rc = resolve_slice(ctx, rows, row_begin, row_end)) return rc;

	u64 const eff_words(eff_row_words(ctx));
	splice_grid g;
	if (int const rc = make_grid(ctx, n_rows, g)) return rc;
	V2M_HIP_TRY(ctx, ctx->d_tile_counts.ensure(n_rows * ctx->n_tiles * sizeof(u32)));
	v2m::tile_tables tt{ctx->d_tile_edge_begin.as<u32>(), ctx->d_cross_offsets.as<u32>(), ctx->d_cross_edges.as<u32>()};
	{
		// pass 1 builds no row, so it takes more rows per workgroup than pass 2 (the template tile, its byte count and the candidates' changes
		// are set up once per group): as many as the kernel holds (kCountRowsMax = 256; measured per 620 / 244 rows of config 3 / 5: 32 rows
		// 0.64 / 0.84 ms, 64 rows 0.48 / 0.64, 128 rows 0.41 / 0.53, 256 rows 0.38 / 0.49).  V2M_COUNT_ROWS_PER_GROUP overrides.
		char const *const ce(std::getenv("V2M_COUNT_ROWS_PER_GROUP"));
		u32 const count_rows(u32(std::min<u64>(std::max<u64>(1, n_rows), (ce && *ce && std::atoi(ce) > 0) ? u64(std::min(std::atoi(ce), int(v2m::kCountRowsMax))) : u64(v2m::kCountRowsMax))));
		u32 const count_groups(u32((n_rows + count_rows - 1) / count_rows));
		timed_launch tl(ctx, V2M_KERNEL_UNALIGNED_COUNT);
		hipLaunchKernelGGL(v2m::count_unaligned_kernel, dim3(unsigned(u64(ctx->n_tiles) * count_groups)), dim3(v2m::kSpliceThreads), 0, ctx->stream,
			ctx->d_template0.as<v2m::vec4u>(), ctx->d_eff.as<u64>(), eff_words, tt, ctx->d_patches.as<v2m::edge_patch>(), ctx->d_labels.as<char>(),
			ctx->d_tile_counts.as<u32>(), ctx->n_tiles, u32(n_rows), count_rows, count_groups, g.tile_run);
		hipLaunchKernelGGL(v2m::scan_tile_counts_kernel, dim3(unsigned(n_rows)), dim3(256), 0, ctx->stream,
			ctx->d_tile_counts.as<u32>(), ctx->n_tiles, ctx->d_row_lengths.as<u64>());
	}
	auto const launch([&](bool nt) {
		auto const go([&](auto kernel) {
			hipLaunchKernelGGL(kernel, dim3(unsigned(g.n_blocks)), dim3(v2m::kSpliceThreads), 0, ctx->stream,
				ctx->d_template0.as<v2m::vec4u>(), ctx->d_eff.as<u64>(), eff_words, tt, ctx->d_patches.as<v2m::edge_patch>(), ctx->d_labels.as<char>(),
				ctx->d_tile_counts.as<u32>(), ctx->n_tiles, d_out, row_pitch, u32(n_rows), g.rows_per_group, g.n_groups, g.tile_run);
		});
#ifdef V2M_TUNING_BUILD
		// V2M_UNALIGNED_KERNEL=wave | shared64: the stream-out whose every wave packs its own short chunks / the product's with a queue of 64 (tools/unaligned_ab.sh)
		static int const flavour([] { char const *const e(std::getenv("V2M_UNALIGNED_KERNEL")); return !e ? 0 : 0 == std::strcmp(e, "wave") ? 1 : 0 == std::strcmp(e, "shared64") ? 2 : 0; }());
		if (1 == flavour) { if (nt) go(v2m::splice_unaligned_per_wave_kernel<true>); else go(v2m::splice_unaligned_per_wave_kernel<false>); return; }
		if (2 == flavour) { if (nt) go(v2m::splice_unaligned_kernel<true, 64>); else go(v2m::splice_unaligned_kernel<false, 64>); return; }
#endif
		if (nt) go(v2m::splice_unaligned_kernel<true>);
		else go(v2m::splice_unaligned_kernel<false>);
	});
	// Store flavour: as for the aligned splice, which of nontemporal and plain stores is faster depends on the box and the
	// buffer (7.2 vs 5.4 ms per 256 config-3 rows on one box, 5.6 vs 6.0 ms on another), so the first launch that writes
	// >= 1 GiB is issued twice per flavour (same output every time) and the faster one is kept.  V2M_UNALIGNED_STORE=plain|nt forces.
	int mode(ctx->unaligned_store_mode);
	if (char const *const mode_env = std::getenv("V2M_UNALIGNED_STORE")) {
		if (0 == std::strcmp(mode_env, "plain")) mode = 0;
		else if (0 == std::strcmp(mode_env, "nt")) mode = 1;
	}
	if (mode < 0 && n_rows * ctx->ref_len >= (u64(1) << 30)) {
		scoped_events ev;
		V2M_HIP_TRY(ctx, ev.create(5));
		V2M_HIP_TRY(ctx, hipEventRecord(ev[0], ctx->stream));
		for (int i(0); i < 4; ++i) {
			launch(0 == (i & 1));   // nt, plain, nt, plain
			V2M_HIP_TRY(ctx, hipEventRecord(ev[i + 1], ctx->stream));
		}
		V2M_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		float t[4];
		for (int i(0); i < 4; ++i) V2M_HIP_TRY(ctx, hipEventElapsedTime(&t[i], ev[i], ev[i + 1]));
		float const nt_ms(std::min(t[0], t[2])), plain_ms(std::min(t[1], t[3]));
		ctx->unaligned_store_mode = nt_ms <= plain_ms ? 1 : 0;
		char buf[160];
		std::snprintf(buf, sizeof(buf), "unaligned splice stores: %s (calibrated on %llu rows: nontemporal %.3f ms, plain %.3f ms)",
			ctx->unaligned_store_mode ? "nontemporal" : "plain", (unsigned long long) n_rows, nt_ms, plain_ms);
		if (ctx->info.size() > 2000) ctx->info.clear();
		if (!ctx->info.empty()) ctx->info += "; ";
		ctx->info += buf;
		mode = ctx->unaligned_store_mode;
	}
	if (mode < 0) mode = 1;   // small launches before any calibration
	{
		timed_launch tl(ctx, V2M_KERNEL_SPLICE_UNALIGNED);
		launch(0 != mode);
	}
	V2M_HIP_TRY(ctx, hipGetLastError());
	return V2M_OK;
}


int check_batch(v2m_ctx *ctx, v2m_row_batch const *rows, u32 flags)
{
	if (!ctx) return V2M_ERR_INVALID_ARGUMENT;
	if (!rows) return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "rows is NULL");
	if (flags & ~V2M_SPLICE_UNALIGNED) return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "unknown flags 0x%x", flags);
	if (!ctx->has_graph) return fail(ctx, V2M_ERR_STATE, "no graph uploaded");
	if (rows->n_rows && !rows->copy_index && !rows->cut_offsets) return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "rows->copy_index is NULL");
	if (rows->cut_offsets && rows->cut_offsets[rows->n_rows] && (!rows->cut_nodes || !rows->cut_copies))
		return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "cut arrays are NULL");
	if (rows->n_rows >= 0xFFFFFFFFull) return fail(ctx, V2M_ERR_UNSUPPORTED, "too many rows in one batch");
	// the founder searches' edge-major copy of the path matrix (as large as the matrix) is not kept through the output that follows them
	if (ctx->d_by_edge.p) { ctx->d_by_edge.reset(); ctx->by_edge_valid = false; }
	if ((flags & V2M_SPLICE_UNALIGNED) && ctx->has_nul_byte)
		return fail(ctx, V2M_ERR_UNSUPPORTED, "the reference sequence or an ALT label holds a NUL byte, which the unaligned kernels use as their padding marker; aligned mode keeps such bytes");
	return V2M_OK;
}

} // namespace


// =============================================================================================
namespace {

// Calls launch(std::integral_constant<int, kPer>) with the founder kernels' instantiation for `copies` chromosome copies (the larger of the copies
// walked and the bound matrix's columns: a thread owns kPer copies of the order, and the LDS arrays -- the staged edge column among them -- hold
// 1024 * kPer): every count up to 8, then 10, 12, 16 and 20.
template <int kMaxPer, typename t_launch>
void pbwt_dispatch_per_thread(u64 copies, t_launch &&launch)
{
	static_assert(20 == v2m::kPbwtPerThread && 20 == v2m::kPbwtPerThreadRecords && 20 == kMaxPer, "one case per instantiation");
	u64 const per((copies + v2m::kPbwtThreads - 1) / v2m::kPbwtThreads);      // (<= kMaxPer was checked)
	switch (per) {
		case 0: case 1: launch(std::integral_constant<int, 1>{}); return;
		case 2: launch(std::integral_constant<int, 2>{}); return;
		case 3: launch(std::integral_constant<int, 3>{}); return;
		case 4: launch(std::integral_constant<int, 4>{}); return;
		case 5: launch(std::integral_constant<int, 5>{}); return;
		case 6: launch(std::integral_constant<int, 6>{}); return;
		case 7: launch(std::integral_constant<int, 7>{}); return;
		case 8: launch(std::integral_constant<int, 8>{}); return;
		case 9: case 10: launch(std::integral_constant<int, 10>{}); return;
		case 11: case 12: launch(std::integral_constant<int, 12>{}); return;
		default: break;
	}
	if constexpr (kMaxPer > 12) {
		if (per <= 16) launch(std::integral_constant<int, 16>{});
		else launch(std::integral_constant<int, 20>{});
	}
}

} // namespace


extern "C" {

uint32_t v2m_abi_version(void) { return V2M_ABI_VERSION; }

int v2m_ctx_create(int device_id, v2m_ctx **ctx_out)
{
	if (!ctx_out) return fail(nullptr, V2M_ERR_INVALID_ARGUMENT, "ctx_out is NULL");
	*ctx_out = nullptr;
	int count(0);
	hipError_t st(hipGetDeviceCount(&count));
	if (hipSuccess != st || 0 == count)
		return fail(nullptr, V2M_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU fallback", hipSuccess != st ? hipGetErrorString(st) : "device count is 0");
	if (device_id < 0 || device_id >= count)
		return fail(nullptr, V2M_ERR_INVALID_ARGUMENT, "device %d out of range (%d devices)", device_id, count);
	hipDeviceProp_t prop;
	st = hipGetDeviceProperties(&prop, device_id);
	if (hipSuccess != st) return fail(nullptr, V2M_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(st));
	if (0 != std::strncmp(prop.gcnArchName, "gfx950", 6))
		return fail(nullptr, V2M_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 (MI355X) only", device_id, prop.gcnArchName);
	st = hipSetDevice(device_id);
	if (hipSuccess != st) return fail(nullptr, V2M_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(st));

	auto *ctx(new v2m_ctx);
	ctx->device = device_id;
	if (prop.multiProcessorCount > 0) ctx->n_cus = u32(prop.multiProcessorCount);
	if (hipSuccess != (st = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking))
		|| hipSuccess != (st = hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking))) {
		delete ctx;
		return fail(nullptr, V2M_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(st));
	}
	for (int i(0); i < 2 && hipSuccess == st; ++i) {
		st = hipEventCreateWithFlags(&ctx->ev_compute[i], hipEventDisableTiming);
		if (hipSuccess == st) st = hipEventCreateWithFlags(&ctx->ev_copy[i], hipEventDisableTiming);
		if (hipSuccess == st) st = hipEventCreateWithFlags(&ctx->ev_row_stage[i], hipEventDisableTiming);
	}
	if (hipSuccess != st) {
		std::string const what(hipGetErrorString(st));
		v2m_ctx_destroy(ctx);   // destroys whatever was created
		return fail(nullptr, V2M_ERR_HIP, "hipEventCreateWithFlags: %s", what.c_str());
	}
	*ctx_out = ctx;
	return V2M_OK;
}

void v2m_ctx_destroy(v2m_ctx *ctx)
{
	if (!ctx) return;
	(void) hipSetDevice(ctx->device);
	(void) hipStreamSynchronize(ctx->stream);
	(void) hipStreamSynchronize(ctx->copy_stream);
	for (auto &v : ctx->events) for (auto &e : v) { (void) hipEventDestroy(e.begin); (void) hipEventDestroy(e.end); }
	for (auto &e : ctx->free_events) { (void) hipEventDestroy(e.begin); (void) hipEventDestroy(e.end); }
	for (int i(0); i < 2; ++i) {
		if (ctx->ev_compute[i]) (void) hipEventDestroy(ctx->ev_compute[i]);
		if (ctx->ev_copy[i]) (void) hipEventDestroy(ctx->ev_copy[i]);
		if (ctx->ev_row_stage[i]) (void) hipEventDestroy(ctx->ev_row_stage[i]);
	}
	for (auto &slot : ctx->held_ring) if (slot && slot->copied) (void) hipEventDestroy(slot->copied);
	(void) hipStreamDestroy(ctx->stream);
	(void) hipStreamDestroy(ctx->copy_stream);
	delete ctx;
}

const char *v2m_last_error(const v2m_ctx *ctx)
{
	return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

int v2m_ctx_synchronize(v2m_ctx *ctx)
{
	if (!ctx) return V2M_ERR_INVALID_ARGUMENT;
	V2M_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	V2M_HIP_TRY(ctx, hipStreamSynchronize(ctx->copy_stream));
	return V2M_OK;
}

void *v2m_ctx_stream(v2m_ctx *ctx) { return ctx ? (void *) ctx->stream : nullptr; }

const char *v2m_ctx_info(const v2m_ctx *ctx) { return ctx ? ctx->info.c_str() : ""; }


// ---- transpose ------------------------------------------------------------------------------

int v2m_transpose_bits_device(v2m_ctx *ctx, const void *d_src_words, uint64_t n_rows, uint64_t n_cols, void *d_dst_words)
{
	if (!ctx) return V2M_ERR_INVALID_ARGUMENT;
	if (0 == n_cols) return V2M_OK;                          // transpose_matrix.cc:48-49
	if (n_rows % 64 || n_cols % 64)                          // transpose_matrix.cc:53-54
		return fail(ctx, V2M_ERR_PRECONDITION, "matrix dimensions must be multiples of 64 (got %llu x %llu)", (unsigned long long) n_rows, (unsigned long long) n_cols);
	if (0 == n_rows) return V2M_OK;
	if (!d_src_words || !d_dst_words) return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "NULL matrix pointer");
	if (((uintptr_t) d_src_words | (uintptr_t) d_dst_words) & 7) return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "matrix pointers must be 8-byte aligned");
	V2M_HIP_TRY(ctx, hipSetDevice(ctx->device));
	return launch_transpose(ctx, static_cast<u64 const *>(d_src_words), n_rows, n_cols, static_cast<u64 *>(d_dst_words));
}

int v2m_transpose_bits(v2m_ctx *ctx, const uint64_t *src_words, uint64_t n_rows, uint64_t n_cols, uint64_t *dst_words)
{
	if (!ctx) return V2M_ERR_INVALID_ARGUMENT;
	if (0 == n_cols) return V2M_OK;
	if (n_rows % 64 || n_cols % 64)
		return fail(ctx, V2M_ERR_PRECONDITION, "matrix dimensions must be multiples of 64 (got %llu x %llu)", (unsigned long long) n_rows, (unsigned long long) n_cols);
	if (0 == n_rows) return V2M_OK;
	if (!src_words || !dst_words) return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "NULL matrix pointer");
	V2M_HIP_TRY(ctx, hipSetDevice(ctx->device));
	size_t const bytes(n_rows / 64 * n_cols * sizeof(u64));
	dev_buf src, dst;
	V2M_HIP_TRY(ctx, src.ensure(bytes));
	V2M_HIP_TRY(ctx, dst.ensure(bytes));
	V2M_HIP_TRY(ctx, hipMemcpyAsync(src.p, src_words, bytes, hipMemcpyHostToDevice, ctx->stream));
	if (int const rc = launch_transpose(ctx, src.as<u64>(), n_rows, n_cols, dst.as<u64>())) return rc;
	V2M_HIP_TRY(ctx, hipMemcpyAsync(dst_words, dst.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
	V2M_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return V2M_OK;
}


// ---- graph ----------------------------------------------------------------------------------

int v2m_upload_graph(v2m_ctx *ctx, const v2m_graph_view *g, const char *ref_seq, uint64_t ref_len)
{
	if (!ctx) return V2M_ERR_INVALID_ARGUMENT;
	if (!g) return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "graph is NULL");
	u64 const N(g->node_count), E(g->edge_count);
	if (0 == N) return fail(ctx, V2M_ERR_PRECONDITION, "a variant graph has at least the source node");
	if (!g->reference_positions || !g->aligned_positions || !g->alt_edge_count_csum) return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "NULL node array");
	if (E && (!g->alt_edge_targets || !g->alt_edge_label_offsets)) return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "NULL edge array");
	if (ref_len && !ref_seq) return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "ref_seq is NULL");
	if (g->paths_by_chrom_copy_and_edge && (g->path_rows % 64 || g->path_cols % 64 || g->path_rows < E))
		return fail(ctx, V2M_ERR_PRECONDITION, "path matrix must be (>= edge_count) x copies with both dimensions multiples of 64");
	V2M_HIP_TRY(ctx, hipSetDevice(ctx->device));

	u64 const limit32(0xFFFFFFFFull - 2 * v2m::kTileBytes);
	u64 const L(g->aligned_positions[N - 1]);
	if (N >= limit32 || E >= limit32 || L >= limit32 || ref_len >= limit32)
		return fail(ctx, V2M_ERR_UNSUPPORTED, "graph does not fit 32-bit device indices (nodes %llu, edges %llu, aligned length %llu)", (unsigned long long) N, (unsigned long long) E, (unsigned long long) L);

	// --- validate what output_sequence() silently relies on, narrow to 32 bits -------------
	std::vector<u32> ref_pos(N), aln_pos(N), csum(N + 1);
	if (0 != g->reference_positions[0] || 0 != g->aligned_positions[0])
		return fail(ctx, V2M_ERR_PRECONDITION, "node 0 must be at reference and aligned position 0");
	for (u64 n(0); n < N; ++n) {
		u64 const r(g->reference_positions[n]), a(g->aligned_positions[n]);
		if (r > ref_len) return fail(ctx, V2M_ERR_PRECONDITION, "node %llu: reference position %llu is past the reference (%llu)", (unsigned long long) n, (unsigned long long) r, (unsigned long long) ref_len);
		if (n) {
			u64 const pr(g->reference_positions[n - 1]), pa(g->aligned_positions[n - 1]);
			if (r <= pr) return fail(ctx, V2M_ERR_PRECONDITION, "reference positions must increase strictly (node %llu)", (unsigned long long) n);
			if (a < pa || a - pa < r - pr)   // the '-' count at sequence_writer.cc:81 would underflow
				return fail(ctx, V2M_ERR_PRECONDITION, "node %llu: aligned distance is shorter than the reference distance", (unsigned long long) n);
		}
		ref_pos[n] = u32(r);
		aln_pos[n] = u32(a);
	}
	if (0 != g->alt_edge_count_csum[0] || E != g->alt_edge_count_csum[N])
		return fail(ctx, V2M_ERR_PRECONDITION, "alt_edge_count_csum must run from 0 to edge_count");
	for (u64 n(0); n <= N; ++n) {
		if (n && g->alt_edge_count_csum[n] < g->alt_edge_count_csum[n - 1]) return fail(ctx, V2M_ERR_PRECONDITION, "alt_edge_count_csum decreases at node %llu", (unsigned long long) n);
		csum[n] = u32(g->alt_edge_count_csum[n]);
	}
	if (N >= 1 && csum[N] != csum[N - 1])
		return fail(ctx, V2M_ERR_PRECONDITION, "the sink node cannot have ALT edges");

	std::vector<v2m::edge_span> spans(E);
	std::vector<v2m::edge_patch> patches(E);
	std::vector<u32> tgt_prefix_max(E + 1, 0);
	std::vector<u64> overlappable((E + 63) / 64, 0);   // edge e can be skipped by some row iff src[e] < max tgt[0..e)
	u64 label_total(0);
	if (E) {
		label_total = g->alt_edge_label_offsets[E];
		if (label_total >= limit32) return fail(ctx, V2M_ERR_UNSUPPORTED, "label pool too large");
		if (label_total && !g->alt_edge_label_bytes) return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "alt_edge_label_bytes is NULL");
		if (0 != g->alt_edge_label_offsets[0]) return fail(ctx, V2M_ERR_PRECONDITION, "alt_edge_label_offsets must start at 0");
	}
	for (u64 n(0); n < N; ++n) {
		for (u32 e(csum[n]); e < csum[n + 1]; ++e) {
			u64 const tgt(g->alt_edge_targets[e]);
			if (tgt <= n || tgt >= N) return fail(ctx, V2M_ERR_PRECONDITION, "edge %u: target %llu must lie after its source node %llu", e, (unsigned long long) tgt, (unsigned long long) n);
			u64 const lo(g->alt_edge_label_offsets[e]), hi(g->alt_edge_label_offsets[e + 1]);
			if (hi < lo || hi > label_total) return fail(ctx, V2M_ERR_PRECONDITION, "edge %u: bad label offsets", e);
			if (hi - lo > u64(aln_pos[tgt]) - aln_pos[n])   // libbio_assert_lte at sequence_writer.cc:61
				return fail(ctx, V2M_ERR_PRECONDITION, "edge %u: label (%llu) is longer than the aligned span (%u)", e, (unsigned long long) (hi - lo), aln_pos[tgt] - aln_pos[n]);
			spans[e] = {u32(n), u32(tgt)};
			patches[e] = {aln_pos[n], aln_pos[tgt], u32(lo), u32(hi - lo)};
			tgt_prefix_max[e + 1] = std::max(tgt_prefix_max[e], u32(tgt));
			if (n < tgt_prefix_max[e]) overlappable[e >> 6] |= u64(1) << (e & 63);
		}
	}

	// --- who can block an overlappable edge ----------------------------------------------------
	// Edge e is skipped only if an EFFECTIVE earlier edge e' ends past e's source node (tgt[e'] > src[e]); if none of those
	// edges is even set in a row, e is effective there without replaying the walk.  Per overlappable edge: the mask of such
	// edges inside e's own 64-edge word (all ones when one lies in an earlier word: the kernel then replays), indexed by
	// the edge's rank among the overlappable ones.
	std::vector<u32> ovl_rank(overlappable.size() + 1, 0);
	std::vector<u64> blocker_masks;
	for (u64 wi(0); wi < overlappable.size(); ++wi) {
		ovl_rank[wi] = u32(blocker_masks.size());
		for (u64 m(overlappable[wi]); m; m &= m - 1) {
			u64 const b(u64(__builtin_ctzll(m))), e(wi * 64 + b);
			u64 mask(0);
			if (tgt_prefix_max[wi * 64] > spans[e].src) mask = ~u64(0);
			else for (u64 k(0); k < b; ++k) if (spans[wi * 64 + k].tgt > spans[e].src) mask |= u64(1) << k;
			blocker_masks.push_back(mask);
		}
	}
	ovl_rank.back() = u32(blocker_masks.size());

	// --- per-tile edge tables ----------------------------------------------------------------
	u32 const n_tiles(u32(std::max<u64>(1, (L + v2m::kTileBytes - 1) / v2m::kTileBytes)));
	std::vector<u32> tile_edge_begin(n_tiles + 1), cross_offsets(n_tiles + 1, 0), cross_edges;
	{
		u32 e(0);
		for (u32 t(0); t <= n_tiles; ++t) {
			u64 const base(u64(t) * v2m::kTileBytes);
			while (e < E && patches[e].aln_begin < base) ++e;
			tile_edge_begin[t] = e;
		}
		tile_edge_begin[n_tiles] = u32(E);
		// an edge crosses into tile t when aln_begin < t*T < aln_end
		std::vector<u32> counts(n_tiles + 1, 0);
		auto for_each_crossing = [&](auto &&fn) {
			for (u32 e2(0); e2 < E; ++e2) {
				u64 const first(u64(patches[e2].aln_begin) / v2m::kTileBytes + 1);
				if (0 == patches[e2].aln_end) continue;
				u64 const last((u64(patches[e2].aln_end) - 1) / v2m::kTileBytes);
				for (u64 t(first); t <= last && t < n_tiles; ++t) fn(u32(t), e2);
			}
		};
		for_each_crossing([&](u32 t, u32) { ++counts[t]; });
		for (u32 t(0); t < n_tiles; ++t) cross_offsets[t + 1] = cross_offsets[t] + counts[t];
		cross_edges.resize(cross_offsets[n_tiles]);
		std::vector<u32> cursor(cross_offsets.begin(), cross_offsets.end() - 1);
		for_each_crossing([&](u32 t, u32 e2) { cross_edges[cursor[t]++] = e2; });
	}

	// --- upload ------------------------------------------------------------------------------
	ctx->has_graph = false;
	ctx->has_template0 = false;
	V2M_HIP_TRY(ctx, ctx->d_ref.ensure(std::max<u64>(ref_len, 16)));
	if (ref_len) V2M_HIP_TRY(ctx, hipMemcpyAsync(ctx->d_ref.p, ref_seq, ref_len, hipMemcpyHostToDevice, ctx->stream));
	V2M_HIP_TRY(ctx, ctx->d_labels.ensure(std::max<u64>(label_total, 16)));
	if (label_total) V2M_HIP_TRY(ctx, hipMemcpyAsync(ctx->d_labels.p, g->alt_edge_label_bytes, label_total, hipMemcpyHostToDevice, ctx->stream));
	if (int const rc = upload_vec(ctx, ctx->d_ref_pos, ref_pos)) return rc;
	if (int const rc = upload_vec(ctx, ctx->d_aln_pos, aln_pos)) return rc;
	if (int const rc = upload_vec(ctx, ctx->d_spans, spans)) return rc;
	if (int const rc = upload_vec(ctx, ctx->d_patches, patches)) return rc;
	if (int const rc = upload_vec(ctx, ctx->d_overlappable, overlappable)) return rc;
	if (int const rc = upload_vec(ctx, ctx->d_ovl_rank, ovl_rank)) return rc;
	if (int const rc = upload_vec(ctx, ctx->d_blocker_masks, blocker_masks)) return rc;
	if (int const rc = upload_vec(ctx, ctx->d_tile_edge_begin, tile_edge_begin)) return rc;
	if (int const rc = upload_vec(ctx, ctx->d_cross_offsets, cross_offsets)) return rc;
	if (int const rc = upload_vec(ctx, ctx->d_cross_edges, cross_edges)) return rc;

	u64 const n_chunks(u64(n_tiles) * v2m::kTileChunks);
	V2M_HIP_TRY(ctx, ctx->d_template.ensure(n_chunks * 16));
	{
		timed_launch tl(ctx, V2M_KERNEL_TEMPLATE);
		hipLaunchKernelGGL(v2m::expand_reference_row_kernel, dim3(unsigned((n_chunks + 255) / 256)), dim3(256), 0, ctx->stream,
			ctx->d_ref.as<char>(), ctx->d_ref_pos.as<u32>(), ctx->d_aln_pos.as<u32>(), u32(N), u32(L), n_chunks, ctx->d_template.as<uint4>(), '-');
	}
	V2M_HIP_TRY(ctx, hipGetLastError());

	ctx->d_paths = nullptr;
	ctx->by_edge_valid = false;
	ctx->path_rows = ctx->path_cols = ctx->path_pitch = 0;
	if (g->paths_by_chrom_copy_and_edge) {
		size_t const bytes(g->path_rows / 64 * g->path_cols * sizeof(u64));
		V2M_HIP_TRY(ctx, ctx->owned_paths.ensure(std::max<size_t>(bytes, 16)));
		if (bytes) V2M_HIP_TRY(ctx, hipMemcpyAsync(ctx->owned_paths.p, g->paths_by_chrom_copy_and_edge, bytes, hipMemcpyHostToDevice, ctx->stream));
		ctx->d_paths = ctx->owned_paths.as<u64>();
		ctx->path_rows = g->path_rows;
		ctx->path_cols = g->path_cols;
		ctx->path_pitch = g->path_rows / 64;
	}
	V2M_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));

	ctx->n_nodes = N; ctx->n_edges = E; ctx->ref_len = ref_len; ctx->aligned_len = L; ctx->label_bytes = label_total;
	// The reference streams whatever bytes the FASTA / VCF held (sequence_writer.cc:73-74).  Aligned mode does the same here; the
	// unaligned kernels mark padding with byte 0, so a graph that holds one is refused there (check_batch) instead of losing it.
	ctx->has_nul_byte = (ref_len && std::memchr(ref_seq, 0, ref_len)) || (label_total && std::memchr(g->alt_edge_label_bytes, 0, label_total));
	ctx->n_tiles = n_tiles;
	ctx->h_csum = std::move(csum);
	ctx->h_tgt_prefix_max = std::move(tgt_prefix_max);
	ctx->has_graph = true;
	return V2M_OK;
}

int v2m_set_paths_device(v2m_ctx *ctx, const void *d_words, uint64_t path_rows, uint64_t path_cols)
{
	if (!ctx) return V2M_ERR_INVALID_ARGUMENT;
	if (!ctx->has_graph) return fail(ctx, V2M_ERR_STATE, "no graph uploaded");
	if (path_rows % 64 || path_cols % 64 || path_rows < ctx->n_edges)
		return fail(ctx, V2M_ERR_PRECONDITION, "path matrix must be (>= edge_count) x copies with both dimensions multiples of 64");
	if (path_rows && path_cols && (!d_words || ((uintptr_t) d_words & 7))) return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "d_words must be a non-NULL 8-byte aligned device pointer");
	ctx->d_paths = static_cast<u64 const *>(d_words);
	ctx->by_edge_valid = false;
	ctx->path_rows = path_rows;
	ctx->path_cols = path_cols;
	ctx->path_pitch = path_rows / 64;
	return V2M_OK;
}

namespace {

// The library's own copy of paths_by_chrom_copy_and_edge keeps every copy's column on a 128-byte line: the column pitch is
// the word count rounded up to 16.  Nobody but the kernels of this file reads that buffer, and with line-aligned
// destination columns the transpose stores whole lines.
u64 aligned_pitch(u64 words) { return (words + 15) & ~u64(15); }

// Transposes a device-resident transpose input (n_rows copies x n_cols edges, column pitch src_pitch words) into the
// ctx-owned, line-aligned path matrix and binds it.  Asynchronous on the ctx's stream.
int transpose_into_owned_paths(v2m_ctx *ctx, u64 const *d_src, u64 n_rows, u64 n_cols, u64 src_pitch)
{
	u64 const pitch(aligned_pitch(n_cols / 64));
	V2M_HIP_TRY(ctx, ctx->owned_paths.ensure(pitch * n_rows * sizeof(u64)));
	// (the pad words between a column's n_cols / 64 words and its pitch are neither written here nor read by any kernel)
	if (int const rc = launch_transpose(ctx, d_src, n_rows, n_cols, ctx->owned_paths.as<u64>(), src_pitch, pitch)) return rc;
	ctx->d_paths = ctx->owned_paths.as<u64>();
	ctx->by_edge_valid = false;
	ctx->path_rows = n_cols;
	ctx->path_cols = n_rows;
	ctx->path_pitch = pitch;
	return V2M_OK;
}

} // namespace

int v2m_bind_path_matrix_device(v2m_ctx *ctx, const void *d_paths_by_edge_and_chrom_copy, uint64_t n_rows, uint64_t n_cols)
{
	if (!ctx) return V2M_ERR_INVALID_ARGUMENT;
	if (!ctx->has_graph) return fail(ctx, V2M_ERR_STATE, "no graph uploaded");
	if (n_rows % 64 || n_cols % 64)                            // transpose_matrix.cc:53-54
		return fail(ctx, V2M_ERR_PRECONDITION, "matrix dimensions must be multiples of 64 (got %llu x %llu)", (unsigned long long) n_rows, (unsigned long long) n_cols);
	if (n_cols < ctx->n_edges) return fail(ctx, V2M_ERR_PRECONDITION, "path matrix has %llu edge columns, the graph %llu edges", (unsigned long long) n_cols, (unsigned long long) ctx->n_edges);
	V2M_HIP_TRY(ctx, hipSetDevice(ctx->device));
	ctx->d_paths = nullptr;
	ctx->by_edge_valid = false;
	ctx->path_rows = ctx->path_cols = ctx->path_pitch = 0;
	if (0 == n_rows || 0 == n_cols) return V2M_OK;
	if (!d_paths_by_edge_and_chrom_copy || ((uintptr_t) d_paths_by_edge_and_chrom_copy & 7)) return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "the matrix must be a non-NULL 8-byte aligned device pointer");
	return transpose_into_owned_paths(ctx, static_cast<u64 const *>(d_paths_by_edge_and_chrom_copy), n_rows, n_cols, 0);
}

int v2m_upload_path_slice(v2m_ctx *ctx, const uint64_t *src_words, uint64_t n_rows, uint64_t n_cols, uint64_t first_copy, uint64_t n_copies)
{
	if (!ctx) return V2M_ERR_INVALID_ARGUMENT;
	if (first_copy > n_rows || n_copies > n_rows - first_copy) return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "copies [%llu, %llu) are outside the matrix (%llu rows)", (unsigned long long) first_copy, (unsigned long long) (first_copy + n_copies), (unsigned long long) n_rows);
	// one block: the contiguous slice
	u64 const block((n_copies + 7) & ~u64(7));
	return v2m_upload_path_blocks(ctx, src_words, n_rows, n_cols, first_copy, std::max<u64>(block, 8), std::max<u64>(block, 8), first_copy + n_copies);
}

int v2m_upload_path_blocks(v2m_ctx *ctx, const uint64_t *src_words, uint64_t n_rows, uint64_t n_cols, uint64_t first_copy, uint64_t block_copies, uint64_t stride_copies, uint64_t copy_end)
{
	if (!ctx) return V2M_ERR_INVALID_ARGUMENT;
	if (!ctx->has_graph) return fail(ctx, V2M_ERR_STATE, "no graph uploaded");
	if (n_rows % 64 || n_cols % 64)                            // transpose_matrix.cc:53-54
		return fail(ctx, V2M_ERR_PRECONDITION, "matrix dimensions must be multiples of 64 (got %llu x %llu)", (unsigned long long) n_rows, (unsigned long long) n_cols);
	if (n_cols < ctx->n_edges) return fail(ctx, V2M_ERR_PRECONDITION, "path matrix has %llu edge columns, the graph %llu edges", (unsigned long long) n_cols, (unsigned long long) ctx->n_edges);
	if (first_copy % 8 || block_copies % 8 || stride_copies % 8 || 0 == block_copies || stride_copies < block_copies)
		return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "first_copy, block_copies and stride_copies must be multiples of 8 (whole bytes of the bit-packed columns), 0 < block_copies <= stride_copies");
	if (copy_end > n_rows) return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "copies up to %llu are outside the matrix (%llu rows)", (unsigned long long) copy_end, (unsigned long long) n_rows);
	V2M_HIP_TRY(ctx, hipSetDevice(ctx->device));
	ctx->d_paths = nullptr;
	ctx->by_edge_valid = false;
	ctx->path_rows = ctx->path_cols = ctx->path_pitch = 0;

	// this GPU's copies: blocks j = 0, 1, ... at first_copy + j * stride_copies, the last one possibly cut short by copy_end
	u64 n_copies(0), n_blocks(0);
	for (u64 c(first_copy); c < copy_end; c += stride_copies, ++n_blocks) n_copies += std::min(block_copies, copy_end - c);
	if (0 == n_copies || 0 == n_cols) return V2M_OK;              // nothing to bind: rows of this ctx can only be REF rows
	if (!src_words) return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "NULL matrix pointer");

	u64 const hp((n_copies + 63) / 64 * 64);                      // the share's row count: this GPU's copies, padded
	size_t const take((n_copies + 7) / 8), src_col_bytes(n_rows / 8), block_bytes(block_copies / 8), stride_bytes(stride_copies / 8);
	bool const whole(take == src_col_bytes && 1 == n_blocks);
	// a packed share gets line-aligned columns too (pitch rounded up to 16 words); the whole matrix is sent as it is
	u64 const src_pitch(whole ? hp / 64 : aligned_pitch(hp / 64));
	size_t const col_bytes(src_pitch * 8), bytes(col_bytes * n_cols);
	V2M_HIP_TRY(ctx, ctx->d_slice_src.ensure(bytes));
	char const *const src(reinterpret_cast<char const *>(src_words) + first_copy / 8);
	if (whole) {
		// the whole matrix: one contiguous copy
		V2M_HIP_TRY(ctx, hipMemcpyAsync(ctx->d_slice_src.p, src_words, bytes, hipMemcpyHostToDevice, ctx->stream));
	} else {
		// bytes [first_copy / 8 + j * stride_bytes, + block_bytes) of every column, packed block after block into pinned slots by a few
		// host threads and sent slot by slot
		unsigned char const tail_mask((n_copies % 8) ? (unsigned char) ((1u << (n_copies % 8)) - 1) : (unsigned char) 0xFF);
		size_t const slot_cols(std::max<size_t>(1, std::min<size_t>(n_cols, (size_t(64) << 20) / col_bytes)));
		for (int i(0); i < 2; ++i) V2M_HIP_TRY(ctx, ctx->host_ring[i].ensure(slot_cols * col_bytes));
		scoped_events sent;
		V2M_HIP_TRY(ctx, sent.create(2));
		size_t slot_index(0);
		for (size_t c0(0); c0 < n_cols; c0 += slot_cols, ++slot_index) {
			size_t const nc(std::min(slot_cols, size_t(n_cols) - c0));
			int const b(int(slot_index & 1));
			if (slot_index >= 2) V2M_HIP_TRY(ctx, hipEventSynchronize(sent[b]));   // the slot's previous upload has left the host buffer
			char *const stage(static_cast<char *>(ctx->host_ring[b].p));
			auto const pack([&](size_t lo, size_t hi) {
				for (size_t c(lo); c < hi; ++c) {
					char *const d(stage + c * col_bytes);
					char const *const column(src + (c0 + c) * src_col_bytes);
					size_t done(0);
					for (u64 j(0); j < n_blocks; ++j) {
						size_t const n(std::min(block_bytes, take - done));
						std::memcpy(d + done, column + j * stride_bytes, n);
						done += n;
					}
					d[take - 1] = char((unsigned char) d[take - 1] & tail_mask);   // bits of copies past the share are another GPU's
					if (take < col_bytes) std::memset(d + take, 0, col_bytes - take);
				}
			});
			unsigned const n_threads(unsigned(std::max<size_t>(1, std::min<size_t>(8, nc * col_bytes >> 22))));
			if (n_threads <= 1) pack(0, nc);
			else {
				std::vector<std::thread> pool;
				for (unsigned t(0); t < n_threads; ++t) pool.emplace_back(pack, nc * t / n_threads, nc * (t + 1) / n_threads);
				for (auto &t : pool) t.join();
			}
			V2M_HIP_TRY(ctx, hipMemcpyAsync(static_cast<char *>(ctx->d_slice_src.p) + c0 * col_bytes, stage, nc * col_bytes, hipMemcpyHostToDevice, ctx->stream));
			V2M_HIP_TRY(ctx, hipEventRecord(sent[b], ctx->stream));
		}
	}
	if (int const rc = transpose_into_owned_paths(ctx, ctx->d_slice_src.as<u64>(), hp, n_cols, src_pitch)) return rc;
	V2M_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	ctx->d_slice_src.reset();
	return V2M_OK;
}

uint64_t v2m_aligned_length(const v2m_ctx *ctx) { return (ctx && ctx->has_graph) ? ctx->aligned_len : 0; }
uint64_t v2m_min_row_pitch(const v2m_ctx *ctx) { return (ctx && ctx->has_graph) ? ((ctx->aligned_len + 255) & ~u64(255)) : 0; }
uint64_t v2m_max_unaligned_length(const v2m_ctx *ctx) { return (ctx && ctx->has_graph) ? ctx->ref_len + ctx->label_bytes : 0; }


// ---- founder search: chunk walks ------------------------------------------------------------------

namespace {

// What both founder entry points ask of the ctx and of their start states, and the edge-major bits they walk.
int pbwt_check_state(v2m_ctx *ctx, uint64_t n_copies, uint64_t n_chunks, const uint32_t *start_order, int max_copies)
{
	if (0 == n_copies || n_copies > u64(max_copies)) return fail(ctx, V2M_ERR_UNSUPPORTED, "this GPU chunk walk holds at most %d chromosome copies (got %llu)", max_copies, (unsigned long long) n_copies);
	if (n_copies > ctx->path_cols) return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "the bound path matrix has %llu copies, %llu asked for", (unsigned long long) ctx->path_cols, (unsigned long long) n_copies);
	// (the kernels stage a whole edge column -- path_cols / 64 words -- in an LDS array sized with the state)
	if (ctx->path_cols > u64(max_copies)) return fail(ctx, V2M_ERR_UNSUPPORTED, "this GPU chunk walk reads edge columns of at most %d copies; the bound path matrix has %llu columns", max_copies, (unsigned long long) ctx->path_cols);
	// (biased divergence values are edge indices + 2, and the kernels keep bit 31 of a running maximum for "constant")
	if (ctx->n_edges >= 0x7FFFFFF0ull) return fail(ctx, V2M_ERR_UNSUPPORTED, "the GPU chunk walk keeps edge indices in 31 bits (the graph has %llu edges)", (unsigned long long) ctx->n_edges);
	// the start order indexes the workgroup's state arrays in LDS
	for (u64 i(0), n(n_chunks * n_copies); i < n; ++i)
		if (start_order[i] >= n_copies) return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "start_order[%llu] = %u is not a chromosome copy (%llu copies)", (unsigned long long) i, start_order[i], (unsigned long long) n_copies);
	return V2M_OK;
}

int edge_major_paths(v2m_ctx *ctx, u64 const **d_by_edge_out)
{
	u64 const rows(ctx->path_rows), cols(ctx->path_cols);              // both multiples of 64
	if (!ctx->by_edge_valid) {
		V2M_HIP_TRY(ctx, ctx->d_by_edge.ensure(std::max<size_t>(16, rows * (cols / 64) * sizeof(u64))));
		if (int const rc = launch_transpose(ctx, ctx->d_paths, rows, cols, ctx->d_by_edge.as<u64>(), ctx->path_pitch, 0)) return rc;
		ctx->by_edge_valid = true;
	}
	*d_by_edge_out = ctx->d_by_edge.as<u64>();
	return V2M_OK;
}

// v2m_pbwt_cut_trials (the pairs land in the caller's arrays) and v2m_pbwt_cut_trials_streamed (they pass through two pinned
// slots and a callback takes them chunk by chunk while the next slice is on its way).
int pbwt_cut_trials_impl(v2m_ctx *ctx, uint64_t n_copies, uint64_t min_distance,
	uint64_t n_candidates, const uint32_t *cand_edge, const uint64_t *cand_aligned_pos,
	uint64_t n_chunks, const uint64_t *chunk_first, const uint32_t *start_order, const uint32_t *start_divergence,
	uint64_t trial_capacity, uint32_t *trial_pred, uint32_t *trial_class_count, uint64_t *trial_end, uint32_t *chunk_status,
	v2m_trials_sink sink, void *sink_user)
{
	if (!ctx) return V2M_ERR_INVALID_ARGUMENT;
	if (!ctx->has_graph || !ctx->d_paths) return fail(ctx, V2M_ERR_STATE, "the founder search needs an uploaded graph with its path matrix");
	if (0 == n_chunks) return V2M_OK;
	if (!cand_edge || !cand_aligned_pos || !chunk_first || !start_order || !start_divergence || (!sink && (!trial_pred || !trial_class_count)) || !trial_end || !chunk_status)
		return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "NULL array");
	if (int const rc = pbwt_check_state(ctx, n_copies, n_chunks, start_order, v2m::kPbwtMaxCopies)) return rc;
	if (n_candidates >= 0xFFFFFFFFull || ctx->n_edges >= 0xFFFFFFFDull) return fail(ctx, V2M_ERR_UNSUPPORTED, "candidate and edge indices are kept in 32 bits");
	if (chunk_first[0] < 1 || chunk_first[n_chunks] > n_candidates) return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "chunk bounds outside the candidate list");
	for (u64 k(0); k < n_chunks; ++k) if (chunk_first[k] > chunk_first[k + 1]) return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "chunk bounds must not decrease");
	// the candidates as find_cut_positions.cc:113,126-151 makes them: the sentinel (edge 0, node 0), then one per distinct edge index in
	// node order -- both columns ascend (the first real candidate may share edge 0 with the sentinel)
	for (u64 c(0); c < n_candidates; ++c) {
		if (cand_edge[c] > ctx->n_edges || (c && cand_edge[c] < cand_edge[c - 1])) return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "candidate edges must ascend and stay inside the graph (candidate %llu)", (unsigned long long) c);
		if (c && cand_aligned_pos[c] < cand_aligned_pos[c - 1]) return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "candidate aligned positions must not decrease (candidate %llu)", (unsigned long long) c);
	}
	V2M_HIP_TRY(ctx, hipSetDevice(ctx->device));

	// edge-major bits: the bound matrix (rows = edges, columns = copies) transposed back on the device, once per binding
	u64 const cols(ctx->path_cols);
	u64 const *d_by_edge(nullptr);
	if (int const rc = edge_major_paths(ctx, &d_by_edge)) return rc;
	dev_buf d_first, d_cand_edge, d_cand_aln, d_chunk_first, d_order, d_div, d_pred, d_class, d_end, d_status;

	auto const up([&](dev_buf &dst, void const *src, size_t bytes) -> int {
		V2M_HIP_TRY(ctx, dst.ensure(std::max<size_t>(bytes, 16)));
		if (bytes) V2M_HIP_TRY(ctx, hipMemcpyAsync(dst.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
		return V2M_OK;
	});
	if (int const rc = up(d_cand_edge, cand_edge, n_candidates * sizeof(u32))) return rc;
	if (int const rc = up(d_cand_aln, cand_aligned_pos, n_candidates * sizeof(u64))) return rc;
	if (int const rc = up(d_chunk_first, chunk_first, (n_chunks + 1) * sizeof(u64))) return rc;
	if (int const rc = up(d_order, start_order, n_chunks * n_copies * sizeof(u32))) return rc;
	if (int const rc = up(d_div, start_divergence, n_chunks * n_copies * sizeof(u32))) return rc;
	u32 const n_edges(u32(ctx->n_edges));
	V2M_HIP_TRY(ctx, d_first.ensure((u64(n_edges) + 1) * sizeof(u32)));
	V2M_HIP_TRY(ctx, d_pred.ensure(std::max<u64>(16, n_chunks * trial_capacity * sizeof(u32))));
	V2M_HIP_TRY(ctx, d_class.ensure(std::max<u64>(16, n_chunks * trial_capacity * sizeof(u32))));
	V2M_HIP_TRY(ctx, d_end.ensure(n_candidates * sizeof(u64)));
	V2M_HIP_TRY(ctx, d_status.ensure(n_chunks * sizeof(u32)));
	V2M_HIP_TRY(ctx, hipMemsetAsync(d_status.p, 0xFF, n_chunks * sizeof(u32), ctx->stream));
	hipLaunchKernelGGL(v2m::pbwt_first_candidate_kernel, dim3(unsigned((u64(n_edges) + 1 + 255) / 256)), dim3(256), 0, ctx->stream,
		d_cand_edge.as<u32>(), u32(n_candidates), n_edges, d_first.as<u32>());
	// (the kernel is instantiated per copies-per-thread count, 1 .. 8, so that its per-copy loops unroll without branches)
	auto const launch_trials([&](auto per_tag) {
		hipLaunchKernelGGL((v2m::pbwt_cut_trials_kernel<decltype(per_tag)::value>), dim3(unsigned(n_chunks)), dim3(v2m::kPbwtThreads), 0, ctx->stream,
			d_by_edge, u32(cols / 64), u32(n_copies), n_edges, d_first.as<u32>(), d_cand_edge.as<u32>(), d_cand_aln.as<u64>(), min_distance,
			d_chunk_first.as<u64>(), d_order.as<u32>(), d_div.as<u32>(), trial_capacity, d_pred.as<u32>(), d_class.as<u32>(), d_end.as<u64>(), d_status.as<u32>());
	});
	pbwt_dispatch_per_thread<v2m::kPbwtPerThread>(ctx->path_cols, launch_trials);
	V2M_HIP_TRY(ctx, hipGetLastError());
	V2M_HIP_TRY(ctx, hipMemcpyAsync(chunk_status, d_status.p, n_chunks * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
	// (only the candidates of this call's chunks: a caller may be consuming an earlier call's part of the same array meanwhile)
	u64 const cand_lo(chunk_first[0]), cand_hi(chunk_first[n_chunks]);
	if (cand_hi > cand_lo) V2M_HIP_TRY(ctx, hipMemcpyAsync(trial_end + cand_lo, d_end.as<u64>() + cand_lo, (cand_hi - cand_lo) * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
	V2M_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	// only what the chunks produced comes back
	std::vector<u64> produced(n_chunks, 0);
	for (u64 k(0); k < n_chunks; ++k) {
		if (0 != chunk_status[k] || chunk_first[k] == chunk_first[k + 1]) { if (0 != chunk_status[k]) chunk_status[k] = 1; continue; }
		u64 const n(trial_end[chunk_first[k + 1] - 1]);
		if (n > trial_capacity) { chunk_status[k] = 1; continue; }
		produced[k] = n;
	}
	if (!sink) {
		for (u64 k(0); k < n_chunks; ++k) {
			if (0 == produced[k]) continue;
			V2M_HIP_TRY(ctx, hipMemcpyAsync(trial_pred + k * trial_capacity, d_pred.as<u32>() + k * trial_capacity, produced[k] * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
			V2M_HIP_TRY(ctx, hipMemcpyAsync(trial_class_count + k * trial_capacity, d_class.as<u32>() + k * trial_capacity, produced[k] * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
		}
		V2M_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		return V2M_OK;
	}

	// Streamed: runs of whole chunks travel through two pinned slots in turn (pairs of a slice: preds first, class counts behind them); while
	// the callback works through one slice the next one crosses the link.  Nothing of the caller's is ever the target of a copy:
	// hundreds of MB of pageable memory as a copy target cost more to fault in, pin and release than the pairs take to use.
	u64 const slot_pairs(std::max<u64>(trial_capacity, u64(4) << 20));
	for (auto &slot : ctx->trials_stage) V2M_HIP_TRY(ctx, slot.ensure(2 * slot_pairs * sizeof(u32)));
	scoped_events ev;
	V2M_HIP_TRY(ctx, ev.create(2));
	struct slice { u64 first, end; };
	std::vector<slice> slices;
	for (u64 k(0); k < n_chunks;) {
		u64 end(k), pairs(0);
		while (end < n_chunks && (end == k || pairs + produced[end] <= slot_pairs)) pairs += produced[end++];
		slices.push_back({k, end});
		k = end;
	}
	auto const issue([&](std::size_t i) -> int {
		u32 *const pred(ctx->trials_stage[i & 1].as<u32>()), *const cls(pred + slot_pairs);
		u64 at(0);
		for (u64 k(slices[i].first); k < slices[i].end; ++k) {
			if (0 == produced[k]) continue;
			V2M_HIP_TRY(ctx, hipMemcpyAsync(pred + at, d_pred.as<u32>() + k * trial_capacity, produced[k] * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
			V2M_HIP_TRY(ctx, hipMemcpyAsync(cls + at, d_class.as<u32>() + k * trial_capacity, produced[k] * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
			at += produced[k];
		}
		V2M_HIP_TRY(ctx, hipEventRecord(ev[i & 1], ctx->stream));
		return V2M_OK;
	});
	if (int const rc = issue(0)) return rc;
	for (std::size_t i(0); i < slices.size(); ++i) {
		if (i + 1 < slices.size()) if (int const rc = issue(i + 1)) return rc;          // (the other slot: slice i - 1 has been handed over)
		V2M_HIP_TRY(ctx, hipEventSynchronize(ev[i & 1]));
		u32 const *const pred(ctx->trials_stage[i & 1].as<u32>()), *const cls(pred + slot_pairs);
		u64 at(0);
		for (u64 k(slices[i].first); k < slices[i].end; ++k) {
			if (int const rc = sink(sink_user, k, chunk_status[k], pred + at, cls + at, produced[k])) {
				(void) hipStreamSynchronize(ctx->stream);
				return fail(ctx, V2M_ERR_SINK, "the trial sink returned %d at chunk %llu", rc, (unsigned long long) k);
			}
			at += produced[k];
		}
	}
	return V2M_OK;
}

} // namespace


int v2m_pbwt_cut_trials(v2m_ctx *ctx, uint64_t n_copies, uint64_t min_distance,
	uint64_t n_candidates, const uint32_t *cand_edge, const uint64_t *cand_aligned_pos,
	uint64_t n_chunks, const uint64_t *chunk_first, const uint32_t *start_order, const uint32_t *start_divergence,
	uint64_t trial_capacity, uint32_t *trial_pred, uint32_t *trial_class_count, uint64_t *trial_end, uint32_t *chunk_status)
{
	return pbwt_cut_trials_impl(ctx, n_copies, min_distance, n_candidates, cand_edge, cand_aligned_pos, n_chunks, chunk_first, start_order, start_divergence,
		trial_capacity, trial_pred, trial_class_count, trial_end, chunk_status, nullptr, nullptr);
}


int v2m_pbwt_cut_trials_streamed(v2m_ctx *ctx, uint64_t n_copies, uint64_t min_distance,
	uint64_t n_candidates, const uint32_t *cand_edge, const uint64_t *cand_aligned_pos,
	uint64_t n_chunks, const uint64_t *chunk_first, const uint32_t *start_order, const uint32_t *start_divergence,
	uint64_t trial_capacity, uint64_t *trial_end, uint32_t *chunk_status, v2m_trials_sink sink, void *sink_user)
{
	if (ctx && !sink) return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "NULL sink");
	return pbwt_cut_trials_impl(ctx, n_copies, min_distance, n_candidates, cand_edge, cand_aligned_pos, n_chunks, chunk_first, start_order, start_divergence,
		trial_capacity, nullptr, nullptr, trial_end, chunk_status, sink, sink_user);
}


int v2m_pbwt_cut_records(v2m_ctx *ctx, uint64_t n_copies, uint64_t n_cuts, const uint32_t *cut_edge,
	uint64_t n_chunks, const uint64_t *chunk_first_cut, const uint32_t *start_edge, const uint32_t *start_order, const uint32_t *start_divergence,
	uint64_t pool_capacity, uint32_t *pool_lhs, uint32_t *pool_rhs, uint32_t *pool_size,
	uint64_t *rec_pool_end, uint32_t *rec_distinct, uint32_t *rec_first_class, uint32_t *rec_first_is_ref, uint32_t *chunk_status)
{
	if (!ctx) return V2M_ERR_INVALID_ARGUMENT;
	if (!ctx->has_graph || !ctx->d_paths) return fail(ctx, V2M_ERR_STATE, "the founder search needs an uploaded graph with its path matrix");
	if (0 == n_chunks) return V2M_OK;
	if (!cut_edge || !chunk_first_cut || !start_edge || !start_order || !start_divergence || !pool_lhs || !pool_rhs || !pool_size
		|| !rec_pool_end || !rec_distinct || !rec_first_class || !rec_first_is_ref || !chunk_status)
		return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "NULL array");
	if (int const rc = pbwt_check_state(ctx, n_copies, n_chunks, start_order, v2m::kPbwtMaxCopiesRecords)) return rc;
	if (chunk_first_cut[0] < 1 || chunk_first_cut[n_chunks] > n_cuts) return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "chunk bounds outside the cut list");
	for (u64 k(0); k < n_chunks; ++k) {
		if (chunk_first_cut[k] > chunk_first_cut[k + 1]) return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "chunk bounds must not decrease");
		if (chunk_first_cut[k] < chunk_first_cut[k + 1] && start_edge[k] > cut_edge[chunk_first_cut[k] - 1]) return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "chunk %llu: the start state lies past the cut before its first one", (unsigned long long) k);
	}
	for (u64 j(0); j < n_cuts; ++j) if ((j && cut_edge[j] < cut_edge[j - 1]) || cut_edge[j] > ctx->n_edges) return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "cut edges must ascend and stay inside the graph");
	V2M_HIP_TRY(ctx, hipSetDevice(ctx->device));

	u64 const rows(ctx->path_rows), cols(ctx->path_cols);
	u64 const *d_by_edge(nullptr);
	if (int const rc = edge_major_paths(ctx, &d_by_edge)) return rc;
	dev_buf d_cut_edge, d_chunk_first, d_start_edge, d_order, d_div, d_lhs, d_rhs, d_size, d_end, d_distinct, d_first, d_ref, d_status;
	auto const up([&](dev_buf &dst, void const *src, size_t bytes) -> int {
		V2M_HIP_TRY(ctx, dst.ensure(std::max<size_t>(bytes, 16)));
		if (bytes) V2M_HIP_TRY(ctx, hipMemcpyAsync(dst.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
		return V2M_OK;
	});
	if (int const rc = up(d_cut_edge, cut_edge, n_cuts * sizeof(u32))) return rc;
	if (int const rc = up(d_chunk_first, chunk_first_cut, (n_chunks + 1) * sizeof(u64))) return rc;
	if (int const rc = up(d_start_edge, start_edge, n_chunks * sizeof(u32))) return rc;
	if (int const rc = up(d_order, start_order, n_chunks * n_copies * sizeof(u32))) return rc;
	if (int const rc = up(d_div, start_divergence, n_chunks * n_copies * sizeof(u32))) return rc;
	size_t const pool_bytes(std::max<u64>(16, n_chunks * pool_capacity * sizeof(u32)));
	V2M_HIP_TRY(ctx, d_lhs.ensure(pool_bytes));
	V2M_HIP_TRY(ctx, d_rhs.ensure(pool_bytes));
	V2M_HIP_TRY(ctx, d_size.ensure(pool_bytes));
	V2M_HIP_TRY(ctx, d_end.ensure(n_cuts * sizeof(u64)));
	V2M_HIP_TRY(ctx, d_distinct.ensure(n_cuts * sizeof(u32)));
	V2M_HIP_TRY(ctx, d_first.ensure(n_cuts * sizeof(u32)));
	V2M_HIP_TRY(ctx, d_ref.ensure(n_cuts * sizeof(u32)));
	V2M_HIP_TRY(ctx, d_status.ensure(n_chunks * sizeof(u32)));
	V2M_HIP_TRY(ctx, hipMemsetAsync(d_status.p, 0xFF, n_chunks * sizeof(u32), ctx->stream));
	dev_buf d_class_scratch;                                  // the class arrays of the instantiations that do not hold them in LDS (more than 12 288 copies)
	hipError_t scratch_error(hipSuccess);
	auto const launch_records([&](auto per_tag) {
		constexpr int kPer(decltype(per_tag)::value);
		if (!v2m::kPbwtClassesInLds<kPer>) {
			scratch_error = d_class_scratch.ensure(size_t(n_chunks) * v2m::kPbwtClassScratchArrays * v2m::kPbwtThreads * kPer * sizeof(unsigned short));
			if (hipSuccess != scratch_error) return;
		}
		hipLaunchKernelGGL((v2m::pbwt_cut_records_kernel<kPer>), dim3(unsigned(n_chunks)), dim3(v2m::kPbwtThreads), 0, ctx->stream,
			d_by_edge, u32(cols / 64), u32(n_copies), u32(rows), d_cut_edge.as<u32>(), d_chunk_first.as<u64>(), d_start_edge.as<u32>(), d_order.as<u32>(), d_div.as<u32>(),
			pool_capacity, d_lhs.as<u32>(), d_rhs.as<u32>(), d_size.as<u32>(), d_end.as<u64>(), d_distinct.as<u32>(), d_first.as<u32>(), d_ref.as<u32>(), d_status.as<u32>(),
			d_class_scratch.as<unsigned short>());
	});
	pbwt_dispatch_per_thread<v2m::kPbwtPerThreadRecords>(ctx->path_cols, launch_records);
	V2M_HIP_TRY(ctx, scratch_error);
	V2M_HIP_TRY(ctx, hipGetLastError());
	V2M_HIP_TRY(ctx, hipMemcpyAsync(chunk_status, d_status.p, n_chunks * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
	u64 const cut_lo(chunk_first_cut[0]), cut_hi(chunk_first_cut[n_chunks]);   // only the cuts of this call's chunks
	if (cut_hi > cut_lo) {
		V2M_HIP_TRY(ctx, hipMemcpyAsync(rec_pool_end + cut_lo, d_end.as<u64>() + cut_lo, (cut_hi - cut_lo) * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
		V2M_HIP_TRY(ctx, hipMemcpyAsync(rec_distinct + cut_lo, d_distinct.as<u32>() + cut_lo, (cut_hi - cut_lo) * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
		V2M_HIP_TRY(ctx, hipMemcpyAsync(rec_first_class + cut_lo, d_first.as<u32>() + cut_lo, (cut_hi - cut_lo) * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
		V2M_HIP_TRY(ctx, hipMemcpyAsync(rec_first_is_ref + cut_lo, d_ref.as<u32>() + cut_lo, (cut_hi - cut_lo) * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
	}
	V2M_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	for (u64 k(0); k < n_chunks; ++k) {
		if (0 != chunk_status[k]) { chunk_status[k] = 1; continue; }
		if (chunk_first_cut[k] == chunk_first_cut[k + 1]) continue;
		u64 const n(rec_pool_end[chunk_first_cut[k + 1] - 1]);
		if (n > pool_capacity) { chunk_status[k] = 1; continue; }
		V2M_HIP_TRY(ctx, hipMemcpyAsync(pool_lhs + k * pool_capacity, d_lhs.as<u32>() + k * pool_capacity, n * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
		V2M_HIP_TRY(ctx, hipMemcpyAsync(pool_rhs + k * pool_capacity, d_rhs.as<u32>() + k * pool_capacity, n * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
		V2M_HIP_TRY(ctx, hipMemcpyAsync(pool_size + k * pool_capacity, d_size.as<u32>() + k * pool_capacity, n * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
	}
	V2M_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return V2M_OK;
}


// ---- rows -----------------------------------------------------------------------------------

int v2m_splice_rows_device(v2m_ctx *ctx, const v2m_row_batch *rows, uint32_t flags, void *d_out, uint64_t row_pitch, uint64_t *row_lengths_out)
{
	if (int const rc = check_batch(ctx, rows, flags)) return rc;
	if (0 == rows->n_rows) return V2M_OK;
	bool const unaligned(flags & V2M_SPLICE_UNALIGNED);
	if (!d_out || ((uintptr_t) d_out & 15)) return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "d_out must be a 16-byte aligned device pointer");
	u64 const need(unaligned ? v2m_max_unaligned_length(ctx) : ctx->aligned_len);
	if (row_pitch % 16 || row_pitch < ((need + 15) & ~u64(15)))
		return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "row_pitch must be a multiple of 16 and at least %llu rounded up to 16", (unsigned long long) need);
	V2M_HIP_TRY(ctx, hipSetDevice(ctx->device));
	if (!unaligned) {
		if (int const rc = splice_aligned_slice(ctx, rows, 0, rows->n_rows, static_cast<char *>(d_out), row_pitch)) return rc;
		if (row_lengths_out)
			for (u64 r(0); r < rows->n_rows; ++r) row_lengths_out[r] = ctx->aligned_len;
		return V2M_OK;
	}
	if (int const rc = splice_unaligned_slice(ctx, rows, 0, rows->n_rows, static_cast<char *>(d_out), row_pitch)) return rc;
	if (row_lengths_out) {
		V2M_HIP_TRY(ctx, hipMemcpyAsync(row_lengths_out, ctx->d_row_lengths.p, rows->n_rows * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
		V2M_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	}
	return V2M_OK;
}

int v2m_splice_rows(v2m_ctx *ctx, const v2m_row_batch *rows, uint32_t flags, v2m_sink_fn sink, void *user)
{
	if (int const rc = check_batch(ctx, rows, flags)) return rc;
	if (!sink) return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "sink is NULL");
	if (0 == rows->n_rows) return V2M_OK;
	V2M_HIP_TRY(ctx, hipSetDevice(ctx->device));
	bool const unaligned(flags & V2M_SPLICE_UNALIGNED);

	u64 const L(ctx->aligned_len);
	if (0 == L) {
		for (u64 r(0); r < rows->n_rows; ++r)
			if (sink(user, r, "", 0)) return fail(ctx, V2M_ERR_SINK, "sink aborted at row %llu", (unsigned long long) r);
		return V2M_OK;
	}

	// Slices of the batch alternate between two device buffers and two pinned host buffers:
	// the D2H copy of slice s runs on copy_stream while the kernels of slice s+1 run on stream.
	u64 const pitch(unaligned ? ((v2m_max_unaligned_length(ctx) + 255) & ~u64(255)) : v2m_min_row_pitch(ctx));
	char const *const slot_env(std::getenv("V2M_RING_SLOT_BYTES"));   // test knob: force small slices
	// 512-MB slots; 128-MB ones for batches of less than 8 GB (a founder run's 26 rows): a gigabyte of pinned memory takes 0.15 s
	// to set up and as long to give back, which a run of a second notices (config 4 end to end: 1.39 -> 1.22 s)
	u64 const slot_default(rows->n_rows * pitch < (u64(8) << 30) ? (u64(128) << 20) : (u64(512) << 20));
	u64 const slot_target((slot_env && *slot_env) ? std::strtoull(slot_env, nullptr, 10) : slot_default);
	u64 const rows_per_slice(std::max<u64>(1, std::min<u64>(rows->n_rows, slot_target / pitch)));
	u64 const n_slices((rows->n_rows + rows_per_slice - 1) / rows_per_slice);
	u64 const slot_bytes(rows_per_slice * pitch);
	u64 const lengths_bytes(rows_per_slice * sizeof(u64));   // unaligned: row lengths ride at the end of the pinned slot
	for (int i(0); i < (n_slices > 1 ? 2 : 1); ++i) {
		V2M_HIP_TRY(ctx, ctx->ring[i].ensure(slot_bytes));
		V2M_HIP_TRY(ctx, ctx->host_ring[i].ensure(slot_bytes + lengths_bytes));
	}

	auto drain = [&](u64 s) -> int {
		int const b(int(s & 1));
		V2M_HIP_TRY(ctx, hipEventSynchronize(ctx->ev_copy[b]));
		u64 const r0(s * rows_per_slice), r1(std::min(rows->n_rows, r0 + rows_per_slice));
		char const *base(static_cast<char const *>(ctx->host_ring[b].p));
		u64 const *lengths(reinterpret_cast<u64 const *>(base + slot_bytes));
		for (u64 r(r0); r < r1; ++r)
			if (sink(user, r, base + (r - r0) * pitch, unaligned ? lengths[r - r0] : L)) return fail(ctx, V2M_ERR_SINK, "sink aborted at row %llu", (unsigned long long) r);
		return V2M_OK;
	};

	// V2M_SPLICE_TIMING=1: where the host's time of this call went (row tables + launches / waiting for copies + the sink), to stderr
	bool const timing(nullptr != std::getenv("V2M_SPLICE_TIMING"));
	double t_issue(0), t_drain(0);
	auto const now([] { return std::chrono::steady_clock::now(); });
	auto const since([&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(now() - t).count(); });
	auto const t_call(now());

	int rc(V2M_OK);
	u64 launched(0);
	for (u64 s(0); s < n_slices && V2M_OK == rc; ++s) {
		int const b(int(s & 1));
		u64 const r0(s * rows_per_slice), r1(std::min(rows->n_rows, r0 + rows_per_slice));
		auto const t_slice(now());
		rc = unaligned
			? splice_unaligned_slice(ctx, rows, r0, r1, ctx->ring[b].as<char>(), pitch)
			: splice_aligned_slice(ctx, rows, r0, r1, ctx->ring[b].as<char>(), pitch);
		t_issue += since(t_slice);
		if (V2M_OK != rc) break;
		hipError_t st(hipSuccess);
		if (unaligned)   // d_row_lengths is reused by the next slice: take the copy on the compute stream
			st = hipMemcpyAsync(static_cast<char *>(ctx->host_ring[b].p) + slot_bytes, ctx->d_row_lengths.p, (r1 - r0) * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream);
		if (hipSuccess == st) st = hipEventRecord(ctx->ev_compute[b], ctx->stream);
		if (hipSuccess == st) st = hipStreamWaitEvent(ctx->copy_stream, ctx->ev_compute[b], 0);
		if (hipSuccess == st) st = hipMemcpyAsync(ctx->host_ring[b].p, ctx->ring[b].p, (r1 - r0) * pitch, hipMemcpyDeviceToHost, ctx->copy_stream);
		if (hipSuccess == st) st = hipEventRecord(ctx->ev_copy[b], ctx->copy_stream);
		if (hipSuccess != st) { rc = fail(ctx, V2M_ERR_HIP, "D2H pipeline: %s", hipGetErrorString(st)); break; }
		launched = s + 1;
		auto const t_d(now());
		if (s >= 1) rc = drain(s - 1);
		t_drain += since(t_d);
	}
	auto const t_d(now());
	if (V2M_OK == rc && launched) rc = drain(launched - 1);
	t_drain += since(t_d);
	// leave both streams idle whatever happened
	(void) hipStreamSynchronize(ctx->stream);
	(void) hipStreamSynchronize(ctx->copy_stream);
	if (timing) std::fprintf(stderr, "[v2m_splice_rows] %llu rows in %llu slices: %.3f s in all, %.3f s preparing and launching, %.3f s waiting for copies and in the sink\n",
		(unsigned long long) rows->n_rows, (unsigned long long) n_slices, since(t_call), t_issue, t_drain);
	return rc;
}


// ---- rows the sink may keep (ABI 5) -------------------------------------------------------------------------------------------
//
// v2m_splice_rows hands a row over for the duration of the sink call, so whatever the sink does with it -- a write() into a file --
// happens on the calling thread, one row at a time, and the next slice cannot be launched meanwhile: one context, one writer.
// Here a row stays where it is (a pinned slot of a ring of n_slots) until the sink says it is done with it, from any thread:
// the sink queues the row for a pool of writers and returns, the call goes on launching slices and copies, and only a slot whose
// rows are all released is copied into again.
int v2m_splice_rows_held(v2m_ctx *ctx, const v2m_row_batch *rows, uint32_t flags, uint32_t n_slots, v2m_hold_sink_fn sink, void *user)
{
	if (int const rc = check_batch(ctx, rows, flags)) return rc;
	if (!sink) return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "sink is NULL");
	if (n_slots < 2 || n_slots > 64) return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "n_slots must be between 2 and 64 (got %u)", n_slots);
	if (0 == rows->n_rows) return V2M_OK;
	V2M_HIP_TRY(ctx, hipSetDevice(ctx->device));
	bool const unaligned(flags & V2M_SPLICE_UNALIGNED);
	u64 const L(ctx->aligned_len);

	auto &state(ctx->held_state);
	auto const wait_released([&](v2m_row_hold &slot) {
		std::unique_lock<std::mutex> lock(state.mutex);
		state.released.wait(lock, [&] { return 0 == slot.outstanding; });
	});
	// hands rows [r0, r1) of a slot to the sink; returns V2M_ERR_SINK when the sink refuses one (that row and the rows after it count
	// as never delivered)
	auto const deliver([&](v2m_row_hold &slot, u64 r0, u64 r1, u64 pitch, u64 slot_bytes) -> int {
		char const *const base(slot.host.as<char>());
		u64 const *const lengths(reinterpret_cast<u64 const *>(base + slot_bytes));
		{ std::lock_guard<std::mutex> const lock(state.mutex); slot.outstanding = r1 - r0; }
		for (u64 r(r0); r < r1; ++r) {
			if (0 == sink(user, r, L ? base + (r - r0) * pitch : "", (unaligned && L) ? lengths[r - r0] : L, &slot)) continue;
			{ std::lock_guard<std::mutex> const lock(state.mutex); slot.outstanding -= r1 - r; }   // (the only waiter is this thread)
			return fail(ctx, V2M_ERR_SINK, "sink aborted at row %llu", (unsigned long long) r);
		}
		return V2M_OK;
	});

	while (ctx->held_ring.size() < n_slots) {
		std::unique_ptr<v2m_row_hold> slot(new v2m_row_hold);
		slot->ring = &state;
		V2M_HIP_TRY(ctx, hipEventCreateWithFlags(&slot->copied, hipEventDisableTiming));
		ctx->held_ring.push_back(std::move(slot));
	}

	if (0 == L) {   // rows without a byte: nothing to copy, nothing to hold on to
		auto &slot(*ctx->held_ring[0]);
		int const rc(deliver(slot, 0, rows->n_rows, 0, 0));
		wait_released(slot);
		return rc;
	}

	u64 const pitch(unaligned ? ((v2m_max_unaligned_length(ctx) + 255) & ~u64(255)) : v2m_min_row_pitch(ctx));
	char const *const slot_env(std::getenv("V2M_RING_SLOT_BYTES"));   // test knob: force small slices
	u64 const slot_default(rows->n_rows * pitch < (u64(8) << 30) ? (u64(128) << 20) : (u64(512) << 20));
	u64 const slot_target((slot_env && *slot_env) ? std::strtoull(slot_env, nullptr, 10) : slot_default);
	u64 const rows_per_slice(std::max<u64>(1, std::min<u64>(rows->n_rows, slot_target / pitch)));
	u64 const n_slices((rows->n_rows + rows_per_slice - 1) / rows_per_slice);
	u64 const slot_bytes(rows_per_slice * pitch);
	u64 const lengths_bytes(rows_per_slice * sizeof(u64));
	int rc(V2M_OK);
	auto const hip_step([&](hipError_t st, char const *what) {
		if (hipSuccess != st && V2M_OK == rc) rc = fail(ctx, hipErrorOutOfMemory == st ? V2M_ERR_OUT_OF_MEMORY : V2M_ERR_HIP, "%s: %s", what, hipGetErrorString(st));
		return hipSuccess == st;
	});
	for (int i(0); i < (n_slices > 1 ? 2 : 1) && V2M_OK == rc; ++i) hip_step(ctx->ring[i].ensure(slot_bytes), "device slot");
	u64 const slots_used(std::min<u64>(n_slots, n_slices));

	u64 launched(0), delivered(0);
	for (u64 s(0); s < n_slices && V2M_OK == rc; ++s) {
		auto &slot(*ctx->held_ring[s % n_slots]);
		int const d(int(s & 1));
		u64 const r0(s * rows_per_slice), r1(std::min(rows->n_rows, r0 + rows_per_slice));
		wait_released(slot);                                              // the slice that was here n_slots slices ago
		if (!hip_step(slot.host.ensure(slot_bytes + lengths_bytes), "pinned slot")) break;
		// the device slot is written again only when the copy that reads it (slice s - 2) is over
		if (s >= 2 && !hip_step(hipStreamWaitEvent(ctx->stream, ctx->held_ring[(s - 2) % n_slots]->copied, 0), "device slot reuse")) break;
		rc = unaligned
			? splice_unaligned_slice(ctx, rows, r0, r1, ctx->ring[d].as<char>(), pitch)
			: splice_aligned_slice(ctx, rows, r0, r1, ctx->ring[d].as<char>(), pitch);
		if (V2M_OK != rc) break;
		bool ok(true);
		if (unaligned) ok = hip_step(hipMemcpyAsync(slot.host.as<char>() + slot_bytes, ctx->d_row_lengths.p, (r1 - r0) * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream), "row lengths");
		ok = ok && hip_step(hipEventRecord(ctx->ev_compute[d], ctx->stream), "event");
		ok = ok && hip_step(hipStreamWaitEvent(ctx->copy_stream, ctx->ev_compute[d], 0), "event");
		ok = ok && hip_step(hipMemcpyAsync(slot.host.p, ctx->ring[d].p, (r1 - r0) * pitch, hipMemcpyDeviceToHost, ctx->copy_stream), "D2H copy");
		ok = ok && hip_step(hipEventRecord(slot.copied, ctx->copy_stream), "event");
		if (!ok) break;
		launched = s + 1;
		if (s >= 1) {                                                     // the slice before this one: its copy ran under this one's kernels
			auto &prev(*ctx->held_ring[(s - 1) % n_slots]);
			if (!hip_step(hipEventSynchronize(prev.copied), "D2H copy")) break;
			rc = deliver(prev, r0 - rows_per_slice, r0, pitch, slot_bytes);
			delivered = s;
		}
	}
	if (V2M_OK == rc && launched > delivered) {
		auto &last(*ctx->held_ring[(launched - 1) % n_slots]);
		if (hip_step(hipEventSynchronize(last.copied), "D2H copy"))
			rc = deliver(last, (launched - 1) * rows_per_slice, std::min(rows->n_rows, launched * rows_per_slice), pitch, slot_bytes);
	}
	// whatever happened: both streams idle, and no row still in a writer's hands when the call returns (the slots are the library's)
	(void) hipStreamSynchronize(ctx->stream);
	(void) hipStreamSynchronize(ctx->copy_stream);
	for (u64 i(0); i < std::max<u64>(slots_used, 1); ++i) wait_released(*ctx->held_ring[i]);
	return rc;
}

void v2m_row_release(v2m_row_hold *hold)
{
	if (!hold) return;
	// Notified under the lock: the thread inside v2m_splice_rows_held returns once it sees the last slot at zero, and its caller may destroy
	// the ctx right after -- nothing of the ring is touched once the mutex is let go.
	std::lock_guard<std::mutex> const lock(hold->ring->mutex);
	if (hold->outstanding && 0 == --hold->outstanding) hold->ring->released.notify_all();
}



// ---- output buffers ------------------------------------------------------------------------------

namespace {

// ms of the probe pattern on a buffer (best of two after a warm-up that populates the page tables)
int probe_output(v2m_ctx *ctx, void *p, u64 pitch, u32 n_groups, float &ms_out)
{
	scoped_events ev;
	V2M_HIP_TRY(ctx, ev.create(2));
	u32 const n_tiles(u32(pitch / v2m::kTileBytes));
	ms_out = 1e30f;
	for (int rep(0); rep < 3; ++rep) {
		V2M_HIP_TRY(ctx, hipEventRecord(ev[0], ctx->stream));
		hipLaunchKernelGGL(v2m::probe_write_kernel, dim3(n_tiles * n_groups), dim3(v2m::kSpliceThreads), 0, ctx->stream, static_cast<char *>(p), pitch, n_groups);
		V2M_HIP_TRY(ctx, hipEventRecord(ev[1], ctx->stream));
		V2M_HIP_TRY(ctx, hipEventSynchronize(ev[1]));
		float t(0);
		V2M_HIP_TRY(ctx, hipEventElapsedTime(&t, ev[0], ev[1]));
		if (rep) ms_out = std::min(ms_out, t);
	}
	return V2M_OK;
}

} // namespace

int v2m_alloc_output(v2m_ctx *ctx, uint64_t bytes, int candidates, void **d_out)
{
	if (!ctx) return V2M_ERR_INVALID_ARGUMENT;
	if (!d_out || 0 == bytes) return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "bad v2m_alloc_output arguments");
	*d_out = nullptr;
	V2M_HIP_TRY(ctx, hipSetDevice(ctx->device));

	// the probe writes n_groups x 16 pseudo-rows of whole tiles; too small a buffer cannot be probed meaningfully
	u32 const n_groups(32);
	u64 const pitch((bytes / (n_groups * 16)) & ~u64(v2m::kTileBytes - 1));
	bool const can_probe(pitch >= u64(64) * v2m::kTileBytes);
	auto const note([&](std::string const &text) {
		if (ctx->info.size() > 2000) ctx->info.clear();   // keep the note bounded over many allocations
		if (!ctx->info.empty()) ctx->info += "; ";
		ctx->info += text;
	});

	// hipMalloc; with candidates > 1 several are held at once, the store pattern is timed on each and the fastest kept.
	// (Buffers mapped from physically contiguous chunks with hipMemCreate / hipMemMap write at the same rate as the best
	// hipMalloc'ed ones, but that API is not used: on this stack (ROCm 7.2) a range that is unmapped, freed and mapped again
	// keeps stale address translations -- rows written into a re-mapped buffer were silently lost, and the round-1 probe's
	// GPU memory access fault was the same thing -- see profiles/r02/output_buffer_vmm_reuse.txt.)
	struct candidate_set {
		std::vector<void *> bufs;
		void *keep{};
		~candidate_set() { for (void *p : bufs) if (p != keep) (void) hipFree(p); }
	} cs;
	bool const probe(candidates > 1 && can_probe);
	for (int c(0); c < (probe ? candidates : 1); ++c) {
		void *p(nullptr);
		hipError_t const st(hipMalloc(&p, bytes));
		if (hipSuccess != st) {
			(void) hipGetLastError();
			if (cs.bufs.empty()) return fail(ctx, V2M_ERR_OUT_OF_MEMORY, "hipMalloc of %llu bytes failed: %s", (unsigned long long) bytes, hipGetErrorString(st));
			break;   // keep what fits
		}
		cs.bufs.push_back(p);
	}
	std::size_t best(0);
	if (probe && cs.bufs.size() > 1) {
		std::vector<float> ms;
		for (void *p : cs.bufs) {
			float t(0);
			if (int const rc = probe_output(ctx, p, pitch, n_groups, t)) return rc;   // cs frees every candidate
			ms.push_back(t);
		}
		best = std::size_t(std::min_element(ms.begin(), ms.end()) - ms.begin());
		std::string text("output buffer chosen among " + std::to_string(cs.bufs.size()) + " hipMalloc candidates by probe write rate (GB/s):");
		for (float const t : ms) { char b2[32]; std::snprintf(b2, sizeof(b2), " %.0f", double(pitch) * n_groups * 16 / (t * 1e6)); text += b2; }
		note(text);
	}
	cs.keep = cs.bufs[best];
	*d_out = cs.keep;
	return V2M_OK;
}

int v2m_free_output(v2m_ctx *ctx, void *d_ptr)
{
	if (!ctx) return V2M_ERR_INVALID_ARGUMENT;
	if (!d_ptr) return V2M_OK;
	V2M_HIP_TRY(ctx, hipSetDevice(ctx->device));
	V2M_HIP_TRY(ctx, hipFree(d_ptr));
	return V2M_OK;
}


// ---- checksums ------------------------------------------------------------------------------

int v2m_checksum_rows_device(v2m_ctx *ctx, const void *d_rows, uint64_t row_pitch, uint64_t n_rows, uint64_t length, const uint64_t *lengths, uint64_t *checksums_out)
{
	if (!ctx) return V2M_ERR_INVALID_ARGUMENT;
	if (0 == n_rows) return V2M_OK;
	if (!d_rows || !checksums_out || ((uintptr_t) d_rows & 7) || (row_pitch & 7)) return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "bad checksum arguments");
	V2M_HIP_TRY(ctx, hipSetDevice(ctx->device));
	V2M_HIP_TRY(ctx, ctx->d_sums.ensure(n_rows * 8));
	V2M_HIP_TRY(ctx, hipMemsetAsync(ctx->d_sums.p, 0, n_rows * 8, ctx->stream));
	u64 max_len(length);
	u64 const *d_lengths(nullptr);
	if (lengths) {
		max_len = *std::max_element(lengths, lengths + n_rows);
		V2M_HIP_TRY(ctx, ctx->d_lengths.ensure(n_rows * 8));
		V2M_HIP_TRY(ctx, hipMemcpyAsync(ctx->d_lengths.p, lengths, n_rows * 8, hipMemcpyHostToDevice, ctx->stream));
		d_lengths = ctx->d_lengths.as<u64>();
	}
	for (u64 r(0); r < n_rows; ++r) {
		u64 const len(lengths ? lengths[r] : length);
		if (len > row_pitch) return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "row %llu is longer than the pitch", (unsigned long long) r);
	}
	u64 const words_per_block(256 * 32);
	u64 const n_words((max_len + 7) / 8);
	u64 const gx(std::max<u64>(1, (n_words + words_per_block - 1) / words_per_block));
	for (u64 r0(0); r0 < n_rows; r0 += 32768) {
		u64 const nr(std::min<u64>(32768, n_rows - r0));
		hipLaunchKernelGGL(v2m::checksum_rows_kernel, dim3(unsigned(gx), unsigned(nr)), dim3(256), 0, ctx->stream,
			static_cast<char const *>(d_rows) + r0 * row_pitch, row_pitch, d_lengths ? d_lengths + r0 : nullptr, length, words_per_block,
			ctx->d_sums.as<unsigned long long>() + r0);
		V2M_HIP_TRY(ctx, hipGetLastError());
	}
	V2M_HIP_TRY(ctx, hipMemcpyAsync(checksums_out, ctx->d_sums.p, n_rows * 8, hipMemcpyDeviceToHost, ctx->stream));
	V2M_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return V2M_OK;
}


// ---- profiling ------------------------------------------------------------------------------

int v2m_profile_enable(v2m_ctx *ctx, int enabled)
{
	if (!ctx) return V2M_ERR_INVALID_ARGUMENT;
	ctx->profiling = (0 != enabled);
	return V2M_OK;
}

int v2m_profile_reset(v2m_ctx *ctx)
{
	if (!ctx) return V2M_ERR_INVALID_ARGUMENT;
	V2M_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	for (auto &v : ctx->events) {
		ctx->free_events.insert(ctx->free_events.end(), v.begin(), v.end());
		v.clear();
	}
	return V2M_OK;
}

int v2m_profile_get(v2m_ctx *ctx, int kernel, uint64_t *launches_out, double *total_ms_out)
{
	if (!ctx) return V2M_ERR_INVALID_ARGUMENT;
	if (kernel < 0 || kernel >= V2M_KERNEL_COUNT) return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "unknown kernel id %d", kernel);
	V2M_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	double total(0);
	for (auto const &e : ctx->events[kernel]) {
		float ms(0);
		V2M_HIP_TRY(ctx, hipEventElapsedTime(&ms, e.begin, e.end));
		total += ms;
	}
	if (launches_out) *launches_out = ctx->events[kernel].size();
	if (total_ms_out) *total_ms_out = total;
	return V2M_OK;
}

int v2m_profile_get_launches(v2m_ctx *ctx, int kernel, double *ms_out, uint64_t capacity, uint64_t *launches_out)
{
	if (!ctx) return V2M_ERR_INVALID_ARGUMENT;
	if (kernel < 0 || kernel >= V2M_KERNEL_COUNT) return fail(ctx, V2M_ERR_INVALID_ARGUMENT, "unknown kernel id %d", kernel);
	V2M_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	auto const &ev(ctx->events[kernel]);
	for (u64 i(0); i < ev.size() && i < capacity && ms_out; ++i) {
		float ms(0);
		V2M_HIP_TRY(ctx, hipEventElapsedTime(&ms, ev[i].begin, ev[i].end));
		ms_out[i] = ms;
	}
	if (launches_out) *launches_out = ev.size();
	return V2M_OK;
}

} // extern "C"
