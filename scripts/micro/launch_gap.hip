// dev aid: what a dependent launch, an event record and a cross-stream wait cost on the device timeline
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void spin(long long *out, int ticks)
{
	const long long t0 = wall_clock64();
	while (wall_clock64() - t0 < ticks)
		;
	if (out && threadIdx.x == 0 && blockIdx.x == 0)
		*out = t0;
}
static double run(int mode, int n, int ticks, int blocks)
{
	hipStream_t s, s2;
	hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
	hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
	hipEvent_t ev[4], a, b;
	for (auto &e : ev)
		hipEventCreateWithFlags(&e, hipEventDisableTiming);
	hipEventCreate(&a);
	hipEventCreate(&b);
	for (int rep = 0; rep < 2; ++rep) {
		hipEventRecord(a, s);
		for (int i = 0; i < n; ++i) {
			hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, s, nullptr, ticks);
			if (mode == 1)
				hipEventRecord(ev[i & 3], s);
			if (mode == 2) {	/* a kernel on the other stream that waits for this one, and we wait for it */
				hipEventRecord(ev[i & 1], s);
				hipStreamWaitEvent(s2, ev[i & 1], 0);
				hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s2, nullptr, 0);
				hipEventRecord(ev[2 + (i & 1)], s2);
				hipStreamWaitEvent(s, ev[2 + (i & 1)], 0);
			}
			if (mode == 3) {	/* wait for an event of the other stream that completed long ago */
				hipStreamWaitEvent(s, ev[0], 0);
			}
		}
		hipEventRecord(b, s);
		hipStreamSynchronize(s);
	}
	float ms = 0;
	hipEventElapsedTime(&ms, a, b);
	hipStreamDestroy(s);
	hipStreamDestroy(s2);
	return ms * 1e3 / n - ticks / 100.0;
}
int main()
{
	const char *names[] = {"back-to-back", "event record between", "round trip through a second stream", "wait on a completed event"};
	for (int blocks : {1, 2048})
		for (int ticks : {0, 2000})
			for (int mode = 0; mode < 4; ++mode)
				printf("blocks %4d kernel %5.1f us  %-38s overhead per launch %6.2f us\n", blocks, ticks / 100.0, names[mode],
				       run(mode, 400, ticks, blocks));
	return 0;
}
