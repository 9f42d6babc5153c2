// resolve_probe.hip -- what does the memory system give the resolve kernel's access pattern?  n_rows rows of n_words 8-byte
// words, one row every `pitch` words, read from one buffer and written to another, as the kernel does it (a workgroup =
// K x 256 consecutive words of one row, rows fastest or pieces fastest in the grid), against a flat copy of the same bytes.
// Build: hipcc --offload-arch=gfx950 -O3 -o resolve_probe resolve_probe.hip      Run: ./resolve_probe [rows words]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int K, bool kRowsFastest, int kMode>   // mode 0: copy, 1: read only (one store per thread), 2: write only
__global__ __launch_bounds__(256) void rows_copy(u64 const *__restrict__ src, u64 *__restrict__ dst, u64 src_pitch, u64 dst_pitch, unsigned n_words)
{
	unsigned const row = kRowsFastest ? blockIdx.x : blockIdx.y, piece = kRowsFastest ? blockIdx.y : blockIdx.x;
	unsigned const w0 = piece * K * 256 + threadIdx.x;
	u64 v[K];
#pragma unroll
	for (int k = 0; k < K; ++k) {
		unsigned const wi = w0 + k * 256, wc = wi < n_words ? wi : n_words - 1;
		v[k] = kMode == 2 ? (u64) wi : src[(u64) row * src_pitch + wc];
	}
	if (kMode == 1) {
		u64 acc = 0;
#pragma unroll
		for (int k = 0; k < K; ++k) acc ^= v[k];
		if (acc == 0x123456789ULL) dst[(u64) row * dst_pitch + w0] = acc;
		return;
	}
#pragma unroll
	for (int k = 0; k < K; ++k) {
		unsigned const wi = w0 + k * 256;
		if (wi < n_words) dst[(u64) row * dst_pitch + wi] = v[k];
	}
}

__global__ __launch_bounds__(256) void flat_copy(uint4 const *__restrict__ src, uint4 *__restrict__ dst, size_t n16)
{
	for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t) gridDim.x * 256) dst[i] = src[i];
}

template <typename F> float timed(F f)
{
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	float best = 1e30f;
	for (int i = 0; i < 6; ++i) {
		CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
		float ms; CK(hipEventElapsedTime(&ms, a, b));
		if (i) best = std::min(best, ms);
	}
	return best;
}

int main(int argc, char **argv)
{
	unsigned const rows = argc > 2 ? atoi(argv[1]) : 251, words = argc > 2 ? atoi(argv[2]) : 97489;
	u64 const dst_pitch = (words + 15) & ~15u;
	size_t const bytes = (size_t) rows * words * 8;
	for (u64 src_rows : {(u64) rows, (u64) 20032}) {
		u64 const src_pitch = dst_pitch;
		u64 *src, *dst;
		CK(hipMalloc(&src, src_rows * src_pitch * 8)); CK(hipMalloc(&dst, (size_t) rows * dst_pitch * 8));
		CK(hipMemset(src, 1, src_rows * src_pitch * 8)); CK(hipMemset(dst, 0, (size_t) rows * dst_pitch * 8));
		printf("%u rows x %u words (%.0f MB in, %.0f MB out), source buffer %.1f GB\n", rows, words, bytes / 1e6, bytes / 1e6, src_rows * src_pitch * 8 / 1e9);
		auto report = [&](char const *name, float ms, double moved) { printf("  %-44s %.3f ms = %5.0f GB/s\n", name, ms, moved / ms / 1e6); };
		report("flat copy, 16 B per lane", timed([&] { hipLaunchKernelGGL(flat_copy, dim3(4096), dim3(256), 0, 0, (uint4 const *) src, (uint4 *) dst, bytes / 16); }), 2.0 * bytes);
#define RUN(K, RF, MODE, NAME) { unsigned const pieces = (words + K * 256 - 1) / (K * 256); dim3 g = RF ? dim3(rows, pieces) : dim3(pieces, rows); \
		report(NAME, timed([&] { hipLaunchKernelGGL((rows_copy<K, RF, MODE>), g, dim3(256), 0, 0, src, dst, src_pitch, dst_pitch, words); }), (MODE == 0 ? 2.0 : 1.0) * bytes); }
		RUN(1, false, 0, "copy, 1 word per thread, pieces fastest")
		RUN(1, true, 0, "copy, 1 word per thread, rows fastest")
		RUN(8, false, 0, "copy, 8 words per thread, pieces fastest")
		RUN(8, true, 0, "copy, 8 words per thread, rows fastest")
		RUN(8, true, 1, "read only, 8 words per thread, rows fastest")
		RUN(8, true, 2, "write only, 8 words per thread, rows fastest")
		RUN(8, false, 1, "read only, 8 words per thread, pieces fastest")
		RUN(8, false, 2, "write only, 8 words per thread, pieces fastest")
		CK(hipFree(src)); CK(hipFree(dst));
	}
	return 0;
}
