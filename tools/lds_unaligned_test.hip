// Does gfx950 take 16-, 8-, 4- and 2-byte LDS stores at ANY byte address (hipcc emits ds_write_b128 / b64 / b32 / b16 for under-aligned
// __shared__ accesses: unaligned access mode)?  The unaligned stream-out's staging buffer depends on it.  GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o tools/lds_unaligned_test tools/lds_unaligned_test.hip && tools/lds_unaligned_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef unsigned u32;
typedef unsigned long long u64;
typedef unsigned short u16;
typedef u32 vec4u __attribute__((ext_vector_type(4)));
typedef vec4u vec4u_u __attribute__((aligned(1)));
typedef u64 u64_u __attribute__((aligned(1)));
typedef u32 u32_u __attribute__((aligned(1)));
typedef u16 u16_u __attribute__((aligned(1)));

// every lane writes `width` bytes of its pattern at byte offset lane * 16 + shift (shift 0..15 per launch), then the block is read back with aligned loads
__global__ void k(unsigned char *out, int shift, int width)
{
	__shared__ __attribute__((aligned(16))) unsigned char buf[256 * 16 + 64];
	int const t = threadIdx.x;
	for (int i = t; i < (int) sizeof buf; i += 256) buf[i] = 0xEE;
	__syncthreads();
	vec4u v;
	for (int d = 0; d < 4; ++d) v[d] = 0x03020100u + 0x04040404u * d + 0x10101010u * (t & 7);
	unsigned char *p = buf + t * 16 + shift;
	if (16 == width) *(vec4u_u *) p = v;
	else if (8 == width) *(u64_u *) p = (u64) v[0] | (u64) v[1] << 32;
	else if (4 == width) *(u32_u *) p = v[0];
	else if (2 == width) *(u16_u *) p = (u16) v[0];
	else *p = (unsigned char) v[0];
	__syncthreads();
	for (int i = t; i < (int) (sizeof buf) / 16; i += 256) ((vec4u *) out)[i] = ((vec4u *) buf)[i];
	// and an unaligned 16-byte READ of what lane t wrote
	__syncthreads();
	if (16 == width) { vec4u const r = *(vec4u_u *) p; ((vec4u *) (out + sizeof buf))[t] = r; }
}

int main()
{
	size_t const n = 256 * 16 + 64, total = n + 256 * 16;
	unsigned char *d; hipMalloc(&d, total);
	std::vector<unsigned char> h(total), want(n);
	int bad = 0;
	for (int width : {16, 8, 4, 2, 1}) for (int shift = 0; shift < 16; ++shift) {
		hipMemset(d, 0, total);
		hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, shift, width);
		if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed width %d shift %d\n", width, shift); return 2; }
		hipMemcpy(h.data(), d, total, hipMemcpyDeviceToHost);
		std::fill(want.begin(), want.end(), 0xEE);
		// lanes in order: a later lane's bytes overwrite an earlier one's only when width + shift overlaps -- with width <= 16 and stride 16 they never do
		for (int t = 0; t < 256; ++t) for (int b = 0; b < width; ++b) want[t * 16 + shift + b] = (unsigned char) (b + 0x10 * (t & 7));
		if (memcmp(want.data(), h.data(), n)) { ++bad; printf("MISMATCH width %d shift %d\n", width, shift); }
		if (16 == width) for (int t = 0; t < 256; ++t) for (int b = 0; b < 16; ++b) if (h[n + t * 16 + b] != (unsigned char) (b + 0x10 * (t & 7))) { ++bad; printf("READ MISMATCH shift %d lane %d\n", shift, t); t = 256; break; }
	}
	printf(bad ? "lds unaligned: %d failures\n" : "lds unaligned: every width at every byte offset ok\n", bad);
	return bad ? 1 : 0;
}
