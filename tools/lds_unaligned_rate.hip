// How fast are LDS stores / loads at byte-granular addresses on gfx950?  (The staged unaligned stream-out stood or fell with it.)
//   hipcc --offload-arch=gfx950 -O3 -o tools/lds_unaligned_rate tools/lds_unaligned_rate.hip && tools/lds_unaligned_rate
// One workgroup of 256 threads per CU, every lane writes (or reads) WIDTH bytes at lane * 16 + shift, 4096 times; cycles per wave instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32;
typedef unsigned long long u64;
typedef unsigned short u16;
typedef u32 vec4u __attribute__((ext_vector_type(4)));
typedef vec4u vec4u_u __attribute__((aligned(1)));
typedef u64 u64_u __attribute__((aligned(1)));
typedef u32 u32_u __attribute__((aligned(1)));

template <int kWidth, bool kRead>
__global__ __launch_bounds__(256) void k(u32 *out, int shift, int iters, u64 *cycles)
{
	__shared__ __attribute__((aligned(16))) unsigned char buf[256 * 16 + 64];
	int const t = threadIdx.x;
	unsigned char *p = buf + t * 16 + shift;
	vec4u v = {(u32) t, 1, 2, 3};
	u32 acc = 0;
	__syncthreads();
	u64 const t0 = clock64();
	for (int i = 0; i < iters; ++i) {
		if (kRead) {
			if (16 == kWidth) { vec4u const r = *(volatile vec4u_u *) p; acc += r[0] + r[3]; }
			else if (8 == kWidth) acc += (u32) *(volatile u64_u *) p;
			else acc += *(volatile u32_u *) p;
		} else {
			if (16 == kWidth) *(volatile vec4u_u *) p = v;
			else if (8 == kWidth) *(volatile u64_u *) p = (u64) v[0];
			else *(volatile u32_u *) p = v[0];
		}
	}
	__syncthreads();
	u64 const t1 = clock64();
	if (0 == t && 0 == blockIdx.x) cycles[0] = t1 - t0;
	out[blockIdx.x * 256 + t] = acc + buf[t];
}

template <int kWidth, bool kRead> void run(u32 *d, u64 *c)
{
	for (int shift : {0, 1, 2, 3, 4, 5, 8, 12, 15}) {
		int const iters = 4096;
		hipLaunchKernelGGL((k<kWidth, kRead>), dim3(256), dim3(256), 0, 0, d, shift, iters, c);
		hipDeviceSynchronize();
		u64 h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
		printf("%s of %2d bytes at shift %2d: %6.1f clock64 ticks per instruction of one wave (4 waves issue in turn)\n", kRead ? "load " : "store", kWidth, shift, (double) h / iters / 4);
	}
}

int main()
{
	u32 *d; u64 *c; hipMalloc(&d, 256 * 256 * 4); hipMalloc(&c, 8);
	run<16, false>(d, c); run<8, false>(d, c); run<4, false>(d, c);
	run<16, true>(d, c); run<8, true>(d, c); run<4, true>(d, c);
	return 0;
}
