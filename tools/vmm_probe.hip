// vmm_probe.hip -- is the write rate of the splice store pattern a property of the physical memory chunk?
// Physical memory is taken in 2-GB chunks through the virtual-memory API (hipMemCreate), every chunk is mapped and
// probed on its own (the splice store pattern squeezed into the chunk, and a sequential fill), then three 64-GB
// buffers are put together from the fastest, the slowest and the first 32 chunks and the full pattern
// (627 rows x 100 MB, 16-KiB tiles, 16 rows per group) is timed on each.
// Build: hipcc --offload-arch=gfx950 -O3 -o vmm_probe vmm_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>

typedef unsigned int vec4u __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void rows_kernel(char *out, size_t pitch, unsigned n_groups, unsigned rows_per_group, unsigned n_rows)
{
	unsigned const tile = blockIdx.x / n_groups, group = blockIdx.x % n_groups;
	vec4u const v = {0x2d2d2d2du, 0x41434754u, tile, group};
	for (unsigned r = 0; r < rows_per_group; ++r) {
		unsigned const row = group * rows_per_group + r;
		if (row >= n_rows) break;
		char *dst = out + (size_t) row * pitch + (size_t) tile * 16384;
#pragma unroll
		for (int k = 0; k < 4; ++k)
			if ((size_t) tile * 16384 + (threadIdx.x + 256 * k) * 16 + 16 <= pitch)   // the last tile of a row is partial: stay inside the row
				__builtin_nontemporal_store(v, (vec4u *) (dst + (threadIdx.x + 256 * k) * 16));
	}
}

__global__ __launch_bounds__(256) void fill_kernel(char *out, size_t n16)
{
	vec4u const v = {1, 2, 3, 4};
	for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t) gridDim.x * 256)
		__builtin_nontemporal_store(v, (vec4u *) out + i);
}

template <typename F> float timed(F f, int reps = 3)
{
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	float best = 1e30f;
	for (int i = 0; i < reps + 1; ++i) {
		CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
		float ms; CK(hipEventElapsedTime(&ms, a, b));
		if (i) best = std::min(best, ms);
	}
	CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
	return best;
}

int main(int argc, char **argv)
{
	int const max_chunks = argc > 1 ? atoi(argv[1]) : 126;
	hipMemAllocationProp prop = {};
	prop.type = hipMemAllocationTypePinned;
	prop.location.type = hipMemLocationTypeDevice;
	prop.location.id = 0;
	size_t gran = 0;
	CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
	size_t const chunk = size_t(2) << 30;
	printf("allocation granularity %zu bytes; chunk %zu bytes\n", gran, chunk);
	if (chunk % gran) { printf("chunk is not a multiple of the granularity\n"); return 1; }

	std::vector<hipMemGenericAllocationHandle_t> handles;
	for (int i = 0; i < max_chunks; ++i) {
		hipMemGenericAllocationHandle_t h;
		if (hipSuccess != hipMemCreate(&h, chunk, &prop, 0)) { (void) hipGetLastError(); break; }
		handles.push_back(h);
	}
	size_t const n = handles.size();
	printf("%zu chunks = %.0f GB of physical memory\n", n, n * chunk / 1e9);
	hipDeviceptr_t va;
	CK(hipMemAddressReserve(&va, n * chunk, 0, 0, 0));
	hipMemAccessDesc access = {};
	access.location = prop.location;
	access.flags = hipMemAccessFlagsProtReadWrite;
	for (size_t i = 0; i < n; ++i) CK(hipMemMap((char *) va + i * chunk, chunk, 0, handles[i], 0));
	CK(hipMemSetAccess(va, n * chunk, &access, 1));

	// per chunk: 64 pseudo-rows of 32 MB advancing together, and a sequential fill
	size_t const small_pitch = chunk / 64;
	unsigned const small_tiles = unsigned(small_pitch / 16384);
	std::vector<float> rate(n), fill(n);
	for (size_t i = 0; i < n; ++i) {
		char *p = (char *) va + i * chunk;
		float const t = timed([&] { hipLaunchKernelGGL(rows_kernel, dim3(small_tiles * 4), dim3(256), 0, 0, p, small_pitch, 4u, 16u, 64u); });
		float const f = timed([&] { hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, p, chunk / 16); });
		rate[i] = chunk / t / 1e6; fill[i] = chunk / f / 1e6;
	}
	printf("chunk: rows-pattern GB/s (sequential fill GB/s)\n");
	for (size_t i = 0; i < n; ++i) printf("%3zu: %5.0f (%5.0f)%s", i, rate[i], fill[i], (i % 6 == 5 || i + 1 == n) ? "\n" : "   ");
	std::vector<size_t> order(n);
	std::iota(order.begin(), order.end(), 0);
	std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return rate[a] > rate[b]; });
	printf("rows pattern per chunk: best %.0f, median %.0f, worst %.0f GB/s\n", rate[order[0]], rate[order[n / 2]], rate[order[n - 1]]);

	// 64-GB buffers from chosen chunks: the real pattern
	size_t const take = 32;
	if (n < 2 * take) { printf("not enough chunks for the composition test\n"); return 0; }
	size_t const L = 100299831, pitch = (L + 255) & ~size_t(255);
	unsigned const rows = unsigned(take * chunk / pitch), n_groups = (rows + 15) / 16, n_tiles = unsigned((L + 16383) / 16384);
	CK(hipMemUnmap(va, n * chunk));
	auto compose = [&](char const *name, std::vector<size_t> const &pick) {
		for (size_t k = 0; k < take; ++k) CK(hipMemMap((char *) va + k * chunk, chunk, 0, handles[pick[k]], 0));
		CK(hipMemSetAccess(va, take * chunk, &access, 1));
		float const t = timed([&] { hipLaunchKernelGGL(rows_kernel, dim3(n_tiles * n_groups), dim3(256), 0, 0, (char *) va, pitch, n_groups, 16u, rows); }, 4);
		double mean = 0; for (size_t k = 0; k < take; ++k) mean += rate[pick[k]];
		printf("%-28s %u rows x %zu bytes: %.3f ms = %.0f GB/s   (mean chunk probe %.0f)\n", name, rows, pitch, t, double(rows) * L / t / 1e6, mean / take);
		CK(hipMemUnmap(va, take * chunk));
	};
	std::vector<size_t> best(order.begin(), order.begin() + take), worst(order.end() - take, order.end()), first(take), last(take);
	std::iota(first.begin(), first.end(), 0);
	std::iota(last.begin(), last.end(), n - take);
	std::sort(best.begin(), best.end());      // keep allocation order inside the buffer
	std::sort(worst.begin(), worst.end());
	compose("first 32 chunks", first);
	compose("last 32 chunks", last);
	compose("32 fastest chunks", best);
	compose("32 slowest chunks", worst);
	std::vector<size_t> best_by_rate(order.begin(), order.begin() + take);
	compose("32 fastest, fastest first", best_by_rate);
	for (auto h : handles) CK(hipMemRelease(h));
	CK(hipMemAddressFree(va, n * chunk));
	return 0;
}
