// streams_probe.hip -- characterise HBM write behaviour for the splice kernel's access pattern:
// many rows 100 MB apart advancing together in 16-KiB segments, versus one sequential fill,
// over slices of one large allocation.  Build: hipcc --offload-arch=gfx950 -O3 -o streams_probe streams_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef unsigned int vec4u __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool NT, int SEG>
__global__ __launch_bounds__(256) void rows_kernel(char *out, size_t pitch, unsigned n_groups, unsigned rows_per_group)
{
	unsigned const tile = blockIdx.x / n_groups, group = blockIdx.x % n_groups;
	vec4u const v = {0x2d2d2d2du, 0x41434754u, tile, group};
	for (unsigned r = 0; r < rows_per_group; ++r) {
		char *dst = out + (size_t)(group * rows_per_group + r) * pitch + (size_t) tile * SEG;
#pragma unroll
		for (int k = 0; k < SEG / 16 / 256; ++k) {
			if ((size_t) tile * SEG + (threadIdx.x + 256 * k) * 16 + 16 > pitch) continue;   // the last segment of a row is partial: stay inside the row
			vec4u *p = (vec4u *)(dst + (threadIdx.x + 256 * k) * 16);
			if (NT) __builtin_nontemporal_store(v, p); else *p = v;
		}
	}
}

template <bool NT>
__global__ __launch_bounds__(256) void fill_kernel(char *out, size_t n16)
{
	vec4u const v = {1, 2, 3, 4};
	for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t) gridDim.x * 256) {
		vec4u *p = (vec4u *) out + i;
		if (NT) __builtin_nontemporal_store(v, p); else *p = v;
	}
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
	return best;
}

int main(int argc, char **argv)
{
	size_t const L = 100299831, pitch = (L + 255) & ~size_t(255);
	unsigned const rows = 64, rpg = 16, n_groups = rows / rpg;
	unsigned const n_tiles = (L + 16383) / 16384;
	int const n_slices = argc > 1 ? atoi(argv[1]) : 32;   // 32 x 6.4 GB = 205 GB
	size_t const slice = (size_t) rows * pitch;
	char *buf; CK(hipMalloc(&buf, slice * n_slices));
	printf("one allocation of %.1f GB at %p; per 6.4-GB slice: 64 rows x 100 MB written as 16-KiB segments by the splice grid (nt / plain), and a sequential fill (nt)\n", slice * n_slices / 1e9, (void *) buf);
	for (int s = 0; s < n_slices; ++s) {
		char *p = buf + s * slice;
		float nt = timed([&] { hipLaunchKernelGGL((rows_kernel<true, 16384>), dim3(n_tiles * n_groups), dim3(256), 0, 0, p, pitch, n_groups, rpg); });
		float pl = timed([&] { hipLaunchKernelGGL((rows_kernel<false, 16384>), dim3(n_tiles * n_groups), dim3(256), 0, 0, p, pitch, n_groups, rpg); });
		float nt4 = timed([&] { hipLaunchKernelGGL((rows_kernel<true, 4096>), dim3(n_tiles * 4 * n_groups), dim3(256), 0, 0, p, pitch, n_groups, rpg); });
		float fl = timed([&] { hipLaunchKernelGGL((fill_kernel<true>), dim3(4096), dim3(256), 0, 0, p, slice / 16); });
		printf("slice %2d: rows nt %.3f ms (%.0f GB/s)  rows plain %.3f  rows nt 4K-seg %.3f  fill nt %.3f (%.0f GB/s)\n", s, nt, slice / nt / 1e6, pl, nt4, fl, slice / fl / 1e6);
	}
	// the whole 512-row pattern over consecutive 51-GB windows
	for (int w = 0; w + 8 <= n_slices; w += 8) {
		char *p = buf + w * slice;
		float nt = timed([&] { hipLaunchKernelGGL((rows_kernel<true, 16384>), dim3(n_tiles * 32), dim3(256), 0, 0, p, pitch, 32u, rpg); });
		printf("window slices %2d-%2d: 512 rows nt %.3f ms (%.0f GB/s)\n", w, w + 7, nt, 8 * slice / nt / 1e6);
	}
	return 0;
}
