// dev aid: what does the shader clock do under a packed-FP32 load?  s_memtime (shader cycles) against s_memrealtime
// (100 MHz) inside one wave, for a grid of G workgroups of 256 threads running a packed-multiply loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ void spin(long long *out, int iters, float seed)
{
	v2f a[8];
	for (int i = 0; i < 8; ++i)
		a[i] = (v2f){seed + i, seed - i};
	const v2f m = {1.0000001f, 0.9999999f};
	const long long r0 = wall_clock64(), c0 = clock64();
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int i = 0; i < 8; ++i)
			asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
	}
	const long long c1 = clock64(), r1 = wall_clock64();
	float s = 0;
	for (int i = 0; i < 8; ++i)
		s += a[i].x + a[i].y;
	if (threadIdx.x == 0) {
		out[blockIdx.x * 3 + 0] = c1 - c0;
		out[blockIdx.x * 3 + 1] = r1 - r0;
		out[blockIdx.x * 3 + 2] = (long long)s;
	}
}
int main()
{
	long long *d;
	hipMalloc(&d, 8192 * 3 * sizeof(long long));
	int wc = 0;
	hipDeviceGetAttribute(&wc, hipDeviceAttributeWallClockRate, 0);
	printf("wall clock rate attribute: %d kHz\n", wc);
	const int grids[] = {1, 256, 1024, 4096};
	for (int rep = 0; rep < 2; ++rep)
		for (int G : grids)
			for (int iters : {20000, 200000}) {
				hipEvent_t e0, e1;
				hipEventCreate(&e0);
				hipEventCreate(&e1);
				hipEventRecord(e0);
				spin<<<G, 256>>>(d, iters, 1.0f);
				hipEventRecord(e1);
				hipEventSynchronize(e1);
				float ms;
				hipEventElapsedTime(&ms, e0, e1);
				std::vector<long long> h(G * 3);
				hipMemcpy(h.data(), d, G * 3 * sizeof(long long), hipMemcpyDeviceToHost);
				double c = 0, r = 0;
				for (int i = 0; i < G; ++i) {
					c += h[i * 3];
					r += h[i * 3 + 1];
				}
				c /= G;
				r /= G;
				const double us = r / (wc * 1e-3);
				printf("G=%5d iters=%6d  event %8.3f ms  wave: %10.0f memtime ticks in %8.1f us -> %7.1f MHz; ticks per pk op %5.2f; at event time %5.2f ns per op per wave\n",
				       G, iters, ms, c, us, c / us, c / (8.0 * iters), ms * 1e6 / (8.0 * iters));
			}
	return 0;
}
