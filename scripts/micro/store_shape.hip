// dev aid: what the channeliser's plane stores cost by shape.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 store_shape.hip -o store_shape.bin
// 181 MB (8 planes x 2.82 M float2) are written by single-wave workgroups, as k1_fast does: wave group w of a
// grid of G groups x 11 roles writes, in iteration q, outputs [(w + q G) 84 + 8 r, + 8) of every plane.
//   mode 0  k1_fast: lane = (window, plane): one 8-byte store per lane = 8 runs of 64 bytes per instruction
//   mode 1  the same bytes, lane = (plane, window pair): dwordx4 per lane, 32 lanes active (8 runs of 64 bytes)
//   mode 2  one plane per instruction: 8 instructions of 8 lanes x 8 bytes (64-byte run each)
//   mode 3  role-major blocks: blockIdx = role * G + group (the halves of a line go to different XCDs)
//   mode 4  full lines: a wave owns 16 windows x 4 planes (runs of 128 bytes), 6 roles
//   mode 5  1 KB per plane per instruction: a wave owns 128 consecutive outputs of one plane (the upper bound)
// and each of them with the 268.8 MB input stream read alongside (3 dword loads per lane and iteration), +16.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define PLANE (2822400LL + 4096)
#define NOUT 2822400LL

template <int MODE, bool READ> __global__ __launch_bounds__(64) void kern(float2 *dec, const unsigned *raw, int groups, int np, unsigned *sink)
{
	const int lane = threadIdx.x;
	int g, w;
	if (MODE == 3) {
		g = blockIdx.x / groups;
		w = blockIdx.x % groups;
	} else {
		g = (blockIdx.x >> 3) % 11;
		w = (blockIdx.x / 88) * 8 + (blockIdx.x & 7);
	}
	unsigned acc = 0;
	for (int q = 0; q < np; ++q) {
		const long long per = (long long)w + (long long)q * groups;
		if (READ) {
			const unsigned *p = raw + per * 2000 + g * 190;
			acc += p[lane] + p[lane + 64] + p[(lane + 128) < 190 ? lane + 128 : 189];
		}
		const float2 v = make_float2((float)acc, (float)q);
		if (MODE == 0 || MODE == 3) {
			const int kk = lane >> 3, c = lane & 7, k = g * 8 + kk;
			if (k < 84)
				dec[c * PLANE + per * 84 + k] = v;
		} else if (MODE == 1) {
			const int c = lane >> 2, k = g * 8 + 2 * (lane & 3);
			if (lane < 32 && k < 84)
				*reinterpret_cast<float4 *>(&dec[c * PLANE + per * 84 + k]) = make_float4(v.x, v.y, v.x, v.y);
		} else if (MODE == 2) {
			for (int c = 0; c < 8; ++c) {
				const int k = g * 8 + lane;
				if (lane < 8 && k < 84)
					dec[c * PLANE + per * 84 + k] = v;
			}
		} else if (MODE == 4) {
			/* 6 roles of 16 windows (the last one 4): lanes = (plane half 4, window 16); two half-waves = planes 0-3, 4-7 in two instructions */
			if (g < 6) {
				const int k = g * 16 + (lane & 15), c = lane >> 4;
				if (k < 84) {
					dec[c * PLANE + per * 84 + k] = v;
					dec[(c + 4) * PLANE + per * 84 + k] = v;
				}
			}
		} else if (MODE == 5) {
			/* 128 consecutive outputs of one plane per instruction: blocks of (w, q) cover the plane linearly */
			const long long base = ((long long)blockIdx.x * np + q) * 128;
			const int c = (int)(base / NOUT) & 7;
			const long long o = base % NOUT;
			if (o + 128 <= NOUT && base < 8 * NOUT)
				*reinterpret_cast<float4 *>(&dec[c * PLANE + o + 2 * lane]) = make_float4(v.x, v.y, v.x, v.y);
		}
	}
	if (acc == 0x12345678u)
		sink[0] = acc;
}

/* the read stream and the k1_fast store shape in ONE kernel, loads software-pipelined DEPTH iterations ahead (what k1_fast
 * does, minus its arithmetic): is 123 us for k1_fast without its mixer a property of the memory system or of the kernel? */
template <int DEPTH, int SHAPE> __global__ __launch_bounds__(64) void piped(float2 *dec, const unsigned *raw, int groups, int np, unsigned *sink, long long PL)
{
	const int lane = threadIdx.x;
	const int g = (blockIdx.x >> 3) % 11;
	const int w = (blockIdx.x / 88) * 8 + (blockIdx.x & 7);
	unsigned r[DEPTH][3];
	const int l2 = (lane + 128) < 190 ? lane + 128 : 189;
#pragma unroll
	for (int d = 0; d < DEPTH; ++d) {
		const unsigned *p = raw + ((long long)w + (long long)(d < np ? d : np - 1) * groups) * 2000 + g * 190;
		r[d][0] = p[lane];
		r[d][1] = p[lane + 64];
		r[d][2] = p[l2];
	}
	for (int q0 = 0; q0 < np; q0 += DEPTH) {
#pragma unroll
		for (int d = 0; d < DEPTH; ++d) {
			const int q = q0 + d;
			if (q < np) {
				const unsigned acc = r[d][0] + r[d][1] + r[d][2];
				const int qn = q + DEPTH < np ? q + DEPTH : np - 1;
				const unsigned *p = raw + ((long long)w + (long long)qn * groups) * 2000 + g * 190;
				r[d][0] = p[lane];
				r[d][1] = p[lane + 64];
				r[d][2] = p[l2];
				const long long per = (long long)w + (long long)q * groups;
				const float2 v = make_float2((float)acc, (float)q);
				if (SHAPE == 0) {
					const int kk = lane >> 3, c = lane & 7, k = g * 8 + kk;
					if (k < 84)
						dec[c * PL + per * 84 + k] = v;
				} else {
					const int c = lane >> 2, k = g * 8 + 2 * (lane & 3);
					if (lane < 32 && k < 84)
						*reinterpret_cast<float4 *>(&dec[c * PL + per * 84 + k]) = make_float4(v.x, v.y, v.x, v.y);
				}
			}
		}
	}
	if (r[0][0] == 0x12345678u)
		sink[0] = 1;
}

/* the same with ONE 16-byte load per lane (49 lanes cover the slice) instead of three 4-byte ones */
template <int DEPTH> __global__ __launch_bounds__(64) void piped16(float2 *dec, const unsigned *raw, int groups, int np, unsigned *sink, long long PL)
{
	const int lane = threadIdx.x;
	const int g = (blockIdx.x >> 3) % 11;
	const int w = (blockIdx.x / 88) * 8 + (blockIdx.x & 7);
	uint4 r[DEPTH];
	const int l4 = lane < 49 ? lane : 48;
#pragma unroll
	for (int d = 0; d < DEPTH; ++d) {
		const uint4 *p = reinterpret_cast<const uint4 *>(raw + ((long long)w + (long long)(d < np ? d : np - 1) * groups) * 2000 + (g * 190 & ~3));
		r[d] = p[l4];
	}
	for (int q0 = 0; q0 < np; q0 += DEPTH) {
#pragma unroll
		for (int d = 0; d < DEPTH; ++d) {
			const int q = q0 + d;
			if (q < np) {
				const unsigned acc = r[d].x + r[d].y + r[d].z + r[d].w;
				const int qn = q + DEPTH < np ? q + DEPTH : np - 1;
				const uint4 *p = reinterpret_cast<const uint4 *>(raw + ((long long)w + (long long)qn * groups) * 2000 + (g * 190 & ~3));
				r[d] = p[l4];
				const long long per = (long long)w + (long long)q * groups;
				const float2 v = make_float2((float)acc, (float)q);
				const int c = lane >> 2, k = g * 8 + 2 * (lane & 3);
				if (lane < 32 && k < 84)
					*reinterpret_cast<float4 *>(&dec[c * PL + per * 84 + k]) = make_float4(v.x, v.y, v.x, v.y);
			}
		}
	}
	if (r[0].x == 0x12345678u)
		sink[0] = 1;
}

/* full, aligned lines: a wave owns 16 consecutive outputs (one 128-byte line per plane) of a SUPERPERIOD of 4 periods
 * (336 outputs = 21 lines); 21 roles; two store instructions (planes 0-3, 4-7: 4 planes x 8 lanes x 16 bytes) and two
 * 16-byte loads per lane (the 16 windows' 381 samples) per iteration */
template <int DEPTH> __global__ __launch_bounds__(64) void piped_line(float2 *dec, const unsigned *raw, int groups, int np, unsigned *sink, long long PL)
{
	const int lane = threadIdx.x;
	const int g = (blockIdx.x >> 3) % 21;
	const int w = (blockIdx.x / 168) * 8 + (blockIdx.x & 7);
	uint4 r[DEPTH][2];
	const int l4 = lane < 32 ? lane + 64 : 95;
#pragma unroll
	for (int d = 0; d < DEPTH; ++d) {
		const uint4 *p = reinterpret_cast<const uint4 *>(raw + ((long long)w + (long long)(d < np ? d : np - 1) * groups) * 8000 + (g * 381 & ~3));
		r[d][0] = p[lane];
		r[d][1] = p[l4];
	}
	for (int q0 = 0; q0 < np; q0 += DEPTH) {
#pragma unroll
		for (int d = 0; d < DEPTH; ++d) {
			const int q = q0 + d;
			if (q < np) {
				const unsigned acc = r[d][0].x + r[d][0].w + r[d][1].y + r[d][1].z;
				const int qn = q + DEPTH < np ? q + DEPTH : np - 1;
				const uint4 *p = reinterpret_cast<const uint4 *>(raw + ((long long)w + (long long)qn * groups) * 8000 + (g * 381 & ~3));
				r[d][0] = p[lane];
				r[d][1] = p[l4];
				const long long sp = (long long)w + (long long)q * groups;
				const float4 v = make_float4((float)acc, (float)q, (float)acc, 1.0f);
				const int c = lane >> 3, k = 2 * (lane & 7);
				if (lane < 32) {
					*reinterpret_cast<float4 *>(&dec[c * PL + sp * 336 + g * 16 + k]) = v;
					*reinterpret_cast<float4 *>(&dec[(c + 4) * PL + sp * 336 + g * 16 + k]) = v;
				}
			}
		}
	}
	if (r[0][0].x == 0x12345678u)
		sink[0] = 1;
}

template <int DEPTH, int SHAPE> static void runp(float2 *dec, const unsigned *raw, unsigned *sink, const char *name, long long PL = PLANE)
{
	hipEvent_t a, b;
	hipEventCreate(&a);
	hipEventCreate(&b);
	float best = 1e9f;
	for (int rep = 0; rep < 5; ++rep) {
		hipEventRecord(a, 0);
		hipLaunchKernelGGL((piped<DEPTH, SHAPE>), dim3(1344 * 11), dim3(64), 0, 0, dec, raw, 1344, 25, sink, PL);
		hipEventRecord(b, 0);
		hipDeviceSynchronize();
		float ms = 0;
		hipEventElapsedTime(&ms, a, b);
		if (rep && ms < best)
			best = ms;
	}
	printf("%-72s one kernel   %7.1f us   (180.6 MB written + 268.8 MB read: %5.2f TB/s)\n", name, best * 1e3, 449.4e6 / (best * 1e-3) / 1e12);
}

/* upper bound of the same traffic in the friendliest shape: every wave reads 3 KB and writes 2 KB per iteration, 16 bytes per lane,
 * contiguous -- a copy kernel with k1_fast's read : write ratio */
__global__ __launch_bounds__(256) void copyish(float4 *dst, const float4 *src, long long niter_total)
{
	const long long wave = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6;
	const long long nwave = ((long long)gridDim.x * 256) >> 6;
	const int lane = threadIdx.x & 63;
	for (long long it = wave; it < niter_total; it += nwave) {
		const float4 a = src[it * 192 + lane], b = src[it * 192 + 64 + lane], c = src[it * 192 + 128 + lane];
		dst[it * 128 + lane] = make_float4(a.x + b.x, a.y + c.y, a.z, b.w);
		dst[it * 128 + 64 + lane] = make_float4(c.x, b.y, c.z, a.w);
	}
}

template <int MODE, bool READ> static void run(float2 *dec, const unsigned *raw, unsigned *sink, const char *name)
{
	const int periods = 33598;
	int groups = 1344, np = 25;	/* 1344 x 25 = 33600 */
	int blocks = groups * 11;
	if (MODE == 5) {
		blocks = 14784;
		np = (int)((8 * NOUT / 128 + blocks - 1) / blocks);
	}
	(void)periods;
	hipEvent_t a, b;
	hipEventCreate(&a);
	hipEventCreate(&b);
	float best = 1e9f;
	for (int rep = 0; rep < 5; ++rep) {
		hipEventRecord(a, 0);
		hipLaunchKernelGGL((kern<MODE, READ>), dim3(blocks), dim3(64), 0, 0, dec, raw, groups, np, sink);
		hipEventRecord(b, 0);
		hipDeviceSynchronize();
		float ms = 0;
		hipEventElapsedTime(&ms, a, b);
		if (rep && ms < best)
			best = ms;
	}
	printf("%-72s %s  %7.1f us   write %5.2f TB/s%s\n", name, READ ? "with reads" : "stores only", best * 1e3, 180.6e6 / (best * 1e-3) / 1e12,
	       READ ? "   (+268.8 MB read)" : "");
}

template <int MODE> static void both(float2 *dec, const unsigned *raw, unsigned *sink, const char *name)
{
	/* the stores of MODE and the read stream as two kernels on two streams at the same time: do reads and writes overlap? */
	hipStream_t sa, sb;
	hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
	hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
	hipEvent_t a, b, c;
	hipEventCreate(&a);
	hipEventCreate(&b);
	hipEventCreate(&c);
	float best = 1e9f;
	for (int rep = 0; rep < 5; ++rep) {
		hipDeviceSynchronize();
		hipEventRecord(a, sa);
		hipStreamWaitEvent(sb, a, 0);
		hipLaunchKernelGGL((kern<MODE, false>), dim3(MODE == 5 ? 14784 : 1344 * 11), dim3(64), 0, sa, dec, raw, 1344, MODE == 5 ? 12 : 25, sink);
		hipLaunchKernelGGL((kern<6, true>), dim3(1344 * 11), dim3(64), 0, sb, dec, raw, 1344, 25, sink);
		hipEventRecord(c, sb);
		hipStreamWaitEvent(sa, c, 0);
		hipEventRecord(b, sa);
		hipDeviceSynchronize();
		float ms = 0;
		hipEventElapsedTime(&ms, a, b);
		if (rep && ms < best)
			best = ms;
	}
	printf("%-72s two kernels  %7.1f us   (180.6 MB written + 268.8 MB read: %5.2f TB/s)\n", name, best * 1e3, 449.4e6 / (best * 1e-3) / 1e12);
}

int main()
{
	float2 *dec;
	unsigned *raw, *sink;
	hipMalloc(&dec, 8 * (PLANE + 600000) * sizeof(float2));
	hipMalloc(&raw, (size_t)33602 * 2000 * 4 + 4096);
	hipMalloc(&sink, 64);
	hipMemset(raw, 1, (size_t)33602 * 2000 * 4);
	hipMemset(dec, 0, 8 * PLANE * sizeof(float2));
	run<0, false>(dec, raw, sink, "0 k1_fast: 8 runs of 64 B per instruction, 8 B per lane");
	run<1, false>(dec, raw, sink, "1 8 runs of 64 B, 16 B per lane (32 lanes)");
	run<2, false>(dec, raw, sink, "2 one 64 B run per instruction (8 instructions)");
	run<3, false>(dec, raw, sink, "3 as 0, role-major blocks (line halves on different XCDs)");
	run<4, false>(dec, raw, sink, "4 runs of 128 B: 16 windows x 4 planes per instruction");
	run<5, false>(dec, raw, sink, "5 1 KB of one plane per instruction");
	run<6, true>(dec, raw, sink, "6 no stores: the read stream alone");
	both<0>(dec, raw, sink, "0 + read stream");
	both<1>(dec, raw, sink, "1 + read stream");
	both<4>(dec, raw, sink, "4 + read stream");
	both<5>(dec, raw, sink, "5 + read stream");
	{
		hipEvent_t a, b;
		hipEventCreate(&a);
		hipEventCreate(&b);
		for (int blocks : {2048, 8192}) {
			float best = 1e9f;
			for (int rep = 0; rep < 5; ++rep) {
				hipEventRecord(a, 0);
				hipLaunchKernelGGL(copyish, dim3(blocks), dim3(256), 0, 0, (float4 *)dec, (const float4 *)raw, 87500LL);
				hipEventRecord(b, 0);
				hipDeviceSynchronize();
				float ms = 0;
				hipEventElapsedTime(&ms, a, b);
				if (rep && ms < best)
					best = ms;
			}
			printf("copy shape: 3 KB read + 2 KB written per wave iteration, %d blocks              one kernel   %7.1f us   (179.2 MB written + 268.8 MB read: %5.2f TB/s)\n",
			       blocks, best * 1e3, 448.0e6 / (best * 1e-3) / 1e12);
		}
	}
	runp<1, 0>(dec, raw, sink, "k1_fast shape, loads 1 iteration ahead");
	runp<2, 0>(dec, raw, sink, "k1_fast shape, loads 2 iterations ahead");
	runp<4, 0>(dec, raw, sink, "k1_fast shape, loads 4 iterations ahead");
	runp<8, 0>(dec, raw, sink, "k1_fast shape, loads 8 iterations ahead");
	runp<4, 1>(dec, raw, sink, "16 B per lane stores, loads 4 iterations ahead");
	runp<8, 1>(dec, raw, sink, "16 B per lane stores, loads 8 iterations ahead");
	{
		hipEvent_t a, b;
		hipEventCreate(&a);
		hipEventCreate(&b);
		float best = 1e9f;
		for (int rep = 0; rep < 5; ++rep) {
			hipEventRecord(a, 0);
			hipLaunchKernelGGL((piped16<4>), dim3(1344 * 11), dim3(64), 0, 0, dec, raw, 1344, 25, sink, (long long)PLANE);
			hipEventRecord(b, 0);
			hipDeviceSynchronize();
			float ms = 0;
			hipEventElapsedTime(&ms, a, b);
			if (rep && ms < best)
				best = ms;
		}
		printf("one 16 B load per lane + 16 B per lane stores, depth 4                   one kernel   %7.1f us   (%5.2f TB/s)\n", best * 1e3, 449.4e6 / (best * 1e-3) / 1e12);
	}
	{
		hipEvent_t a, b;
		hipEventCreate(&a);
		hipEventCreate(&b);
		for (int groups : {336, 672}) {
			float best = 1e9f;
			const int np = 8400 / groups;
			for (int rep = 0; rep < 5; ++rep) {
				hipEventRecord(a, 0);
				hipLaunchKernelGGL((piped_line<4>), dim3(groups * 21), dim3(64), 0, 0, dec, raw, groups, np, sink, (long long)PLANE);
				hipEventRecord(b, 0);
				hipDeviceSynchronize();
				float ms = 0;
				hipEventElapsedTime(&ms, a, b);
				if (rep && ms < best)
					best = ms;
			}
			printf("full aligned 128 B lines: 21 roles x 4 periods, %4d wave groups                  one kernel   %7.1f us   (%5.2f TB/s)\n", groups, best * 1e3, 449.4e6 / (best * 1e-3) / 1e12);
		}
	}
	for (long long extra : {0LL, 16LL, 64LL, 528LL, 1040LL, 2064LL, 4112LL, 8208LL, 16400LL, 65552LL, 131088LL, 524304LL}) {
		char nm[96];
		snprintf(nm, sizeof nm, "16 B per lane stores, depth 4, plane stride + %lld outputs", extra);
		runp<4, 1>(dec, raw, sink, nm, PLANE + extra);
	}
	run<0, true>(dec, raw, sink, "0 k1_fast: 8 runs of 64 B per instruction, 8 B per lane");
	run<1, true>(dec, raw, sink, "1 8 runs of 64 B, 16 B per lane (32 lanes)");
	run<3, true>(dec, raw, sink, "3 as 0, role-major blocks (line halves on different XCDs)");
	run<4, true>(dec, raw, sink, "4 runs of 128 B: 16 windows x 4 planes per instruction");
	run<5, true>(dec, raw, sink, "5 1 KB of one plane per instruction");
	return 0;
}
