// What does a read + write kernel reach on this memory system, and does the access width per lane matter?  (The transposes move 8 bytes per
// lane and instruction; MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy where torch's copy_ of the config-3 matrix gave 5.48.)
//   hipcc --offload-arch=gfx950 -O3 -o tools/copy_kernel_probe tools/copy_kernel_probe.hip && tools/copy_kernel_probe [MB = 632]
// Grid-stride copies of MB megabytes (source and destination each), widths 8 and 16 bytes per lane, plain and nontemporal, 1 / 2 / 4 / 8 loads in flight
// per lane, several grid sizes; then the same with the destination (or the source) shifted by 8 bytes, i.e. 16-byte accesses that are only 8-byte aligned
// and lines that do not start where a wave's KiB starts -- the dense path matrix's situation.  Best of 10 launches, HIP events.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
typedef unsigned u32;
typedef unsigned long long u64;
typedef u32 vec4u __attribute__((ext_vector_type(4)));
typedef vec4u vec4u_a8 __attribute__((aligned(8)));

template <typename T, int kInFlight, bool kNT>
__global__ __launch_bounds__(256) void copy_kernel(T const *__restrict__ src, T *__restrict__ dst, u64 n)
{
	u64 const stride = (u64) gridDim.x * 256;
	u64 i = (u64) blockIdx.x * 256 + threadIdx.x;
	for (; i + (kInFlight - 1) * stride < n; i += kInFlight * stride) {
		T v[kInFlight];
#pragma unroll
		for (int k = 0; k < kInFlight; ++k) v[k] = kNT ? __builtin_nontemporal_load(src + i + k * stride) : src[i + k * stride];
#pragma unroll
		for (int k = 0; k < kInFlight; ++k) {
			if (kNT) __builtin_nontemporal_store(v[k], dst + i + k * stride);
			else dst[i + k * stride] = v[k];
		}
	}
	for (; i < n; i += stride) dst[i] = src[i];
}

template <typename T, int kInFlight, bool kNT>
float time_copy(void const *src, void *dst, u64 bytes, int blocks)
{
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	float best = 1e30f;
	for (int rep = 0; rep < 10; ++rep) {
		hipEventRecord(e0, 0);
		hipLaunchKernelGGL((copy_kernel<T, kInFlight, kNT>), dim3(blocks), dim3(256), 0, 0, (T const *) src, (T *) dst, bytes / sizeof(T));
		hipEventRecord(e1, 0);
		hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		if (rep) best = std::min(best, ms);
	}
	hipEventDestroy(e0); hipEventDestroy(e1);
	return best;
}

int main(int argc, char **argv)
{
	u64 const mb = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 632;
	u64 const bytes = mb << 20;
	char *a, *b;
	if (hipSuccess != hipMalloc(&a, bytes + 4096) || hipSuccess != hipMalloc(&b, bytes + 4096)) { std::printf("hipMalloc failed\n"); return 1; }
	hipMemset(a, 0x5a, bytes + 4096); hipMemset(b, 0, bytes + 4096); hipDeviceSynchronize();
	std::printf("copy of %llu MB (read + write = %.3f GB); GB/s = bytes read + bytes written per second\n", mb, 2.0 * bytes / 1e9);
	for (int shift_case = 0; shift_case < 3; ++shift_case) {
		char const *src = a + (2 == shift_case ? 8 : 0);
		char *dst = b + (1 == shift_case ? 8 : 0);
		std::printf("%s\n", 0 == shift_case ? "both sides line-aligned" : 1 == shift_case ? "destination shifted by 8 bytes" : "source shifted by 8 bytes");
		for (int blocks : {1024, 2048, 4096, 8192}) {
#define V2M_ROW(T, name, F, NT) { float const ms = time_copy<T, F, NT>(src, dst, bytes, blocks); std::printf("  %-28s %5d workgroups: %.3f ms = %5.0f GB/s\n", name, blocks, ms, 2.0 * bytes / ms / 1e6); }
			V2M_ROW(u64, "8 B/lane, 1 in flight", 1, false)
			V2M_ROW(u64, "8 B/lane, 4 in flight", 4, false)
			V2M_ROW(u64, "8 B/lane, 8 in flight", 8, false)
			V2M_ROW(u64, "8 B/lane, 4 in flight, nt", 4, true)
			V2M_ROW(vec4u_a8, "16 B/lane, 1 in flight", 1, false)
			V2M_ROW(vec4u_a8, "16 B/lane, 2 in flight", 2, false)
			V2M_ROW(vec4u_a8, "16 B/lane, 4 in flight", 4, false)
			V2M_ROW(vec4u_a8, "16 B/lane, 4 in flight, nt", 4, true)
#undef V2M_ROW
		}
	}
	return 0;
}
