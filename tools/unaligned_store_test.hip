// Do 16-byte global stores at arbitrary byte alignment work on gfx950 (HSA unaligned access mode)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef unsigned int vec4u __attribute__((ext_vector_type(4)));
typedef vec4u vec4u_u __attribute__((aligned(1)));
__global__ void k(char *out, int shift)
{
	vec4u v = {0x03020100u + threadIdx.x * 0x10101010u, 0x07060504u, 0x0b0a0908u, 0x0f0e0d0cu};
	*(vec4u_u *) (out + shift + 16 * threadIdx.x) = v;   // lane-contiguous 16-B stores, all misaligned by `shift`
}
int main()
{
	char *d; hipMalloc(&d, 4096);
	int bad = 0;
	for (int shift = 0; shift < 16; ++shift) {
		hipMemset(d, 0xEE, 4096);
		hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, shift);
		if (hipDeviceSynchronize() != hipSuccess) { printf("fault at shift %d\n", shift); return 1; }
		std::vector<unsigned char> h(4096);
		hipMemcpy(h.data(), d, 4096, hipMemcpyDeviceToHost);
		for (int t = 0; t < 64; ++t) for (int b = 0; b < 16; ++b) {
			unsigned char exp = b < 4 ? (unsigned char) (b + ((t * 0x10) & 0xFF)) : (unsigned char) b;
			if (b < 4) exp = (unsigned char) ((0x03020100u + t * 0x10101010u) >> (8 * b));
			if (h[shift + 16 * t + b] != exp) ++bad;
		}
		for (int i = 0; i < shift; ++i) if (h[i] != 0xEE) ++bad;
		if (h[shift + 1024] != 0xEE) ++bad;
	}
	printf("unaligned 16-B stores: %s (%d mismatches)\n", bad ? "WRONG" : "ok", bad);
	return bad != 0;
}
