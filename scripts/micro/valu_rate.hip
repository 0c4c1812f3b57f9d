// dev aid: what the channeliser's packed-FP32 instruction mix costs on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off valu_rate.hip -o valu_rate.bin
// Every kernel is one wavefront per workgroup running the same unrolled body `iters` times; the grid is
// 1024 x W workgroups (W waves per SIMD on a 256-CU part).  Reported: shader cycles (s_memtime) a wave
// needs per VALU instruction of the body, times 1/W = cycles per instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int MODE> __global__ __launch_bounds__(64) void kern(float2 *out, const float2 *in, long long *cyc, int iters)
{
	__shared__ float2 xs[256];
	const int lane = threadIdx.x;
	for (int i = lane; i < 256; i += 64)
		xs[i] = in[i];
	__syncthreads();
	v2f x = {in[lane].x, in[lane].y}, w = {in[64 + lane].x, in[64 + lane].y};
	v2f acc[16], r[16];
#pragma unroll
	for (int i = 0; i < 16; ++i) {
		acc[i] = (v2f){in[128 + i].x, in[128 + i].y};
		r[i] = acc[i];
	}
	v2f ws = {in[200].x, in[200].y};	/* wave-uniform -> SGPR pair */
	const v2f *xp = reinterpret_cast<const v2f *>(&xs[(lane >> 3) * 24]);
	const long long t0 = clock64();
	for (int it = 0; it < iters; ++it) {
		if constexpr (MODE == 0) {	/* 8 independent v_pk_mul_f32 */
#define B(i) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r[i]) : "v"(x), "v"(w));
			REP8(B) REP8(B)
#undef B
		} else if constexpr (MODE == 1) {	/* 8 chains of v_pk_add_f32 */
#define B(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(x));
			REP8(B) REP8(B)
#undef B
		} else if constexpr (MODE == 2) {	/* 16 independent v_mul_f32 */
#define B(i) asm volatile("v_mul_f32 %0, %2, %3\n\tv_mul_f32 %1, %2, %3" : "=v"(r[i].x), "=v"(r[i].y) : "v"(x.x), "v"(w.y));
			REP8(B) REP8(B)
#undef B
		} else if constexpr (MODE == 3) {	/* 16 chains of v_add_f32 */
#define B(i) asm volatile("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %2" : "+v"(acc[i].x), "+v"(acc[i].y) : "v"(x.x));
			REP8(B) REP8(B)
#undef B
		} else if constexpr (MODE == 4) {	/* 16 chains of v_fma_f32 */
#define B(i) asm volatile("v_fma_f32 %0, %2, %3, %0\n\tv_fma_f32 %1, %2, %3, %1" : "+v"(acc[i].x), "+v"(acc[i].y) : "v"(x.x), "v"(w.y));
			REP8(B) REP8(B)
#undef B
		} else if constexpr (MODE == 5) {	/* 8 chains of v_pk_fma_f32 */
#define B(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(w));
			REP8(B) REP8(B)
#undef B
		} else if constexpr (MODE == 6) {	/* the exact complex MAC, one accumulator, no wait states */
#define B(i) asm volatile("v_pk_mul_f32 %1, %3, %4 op_sel_hi:[0,1]\n\t" \
			  "v_pk_mul_f32 %2, %3, %4 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t" \
			  "v_pk_add_f32 %1, %1, %2\n\tv_pk_add_f32 %0, %0, %1" : "+v"(acc[0]), "=&v"(r[0]), "=&v"(r[1]) : "v"(x), "v"(w));
			REP8(B)
#undef B
		} else if constexpr (MODE == 7) {	/* the same with the s_nop 0 the compiler puts in front of dependent packed ops */
#define B(i) asm volatile("v_pk_mul_f32 %1, %3, %4 op_sel_hi:[0,1]\n\t" \
			  "v_pk_mul_f32 %2, %3, %4 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\ts_nop 0\n\t" \
			  "v_pk_add_f32 %1, %1, %2\n\ts_nop 0\n\tv_pk_add_f32 %0, %0, %1" : "+v"(acc[0]), "=&v"(r[0]), "=&v"(r[1]) : "v"(x), "v"(w));
			REP8(B)
#undef B
		} else if constexpr (MODE == 8 || MODE == 9) {	/* four accumulators interleaved (8: w in VGPRs, 9: w in SGPRs) */
#define MUL4(W) "v_pk_mul_f32 %4, %12, " W " op_sel_hi:[0,1]\n\tv_pk_mul_f32 %5, %12, " W " op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t" \
		"v_pk_mul_f32 %6, %12, " W " op_sel_hi:[0,1]\n\tv_pk_mul_f32 %7, %12, " W " op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t" \
		"v_pk_mul_f32 %8, %12, " W " op_sel_hi:[0,1]\n\tv_pk_mul_f32 %9, %12, " W " op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t" \
		"v_pk_mul_f32 %10, %12, " W " op_sel_hi:[0,1]\n\tv_pk_mul_f32 %11, %12, " W " op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t" \
		"v_pk_add_f32 %4, %4, %5\n\tv_pk_add_f32 %6, %6, %7\n\tv_pk_add_f32 %8, %8, %9\n\tv_pk_add_f32 %10, %10, %11\n\t" \
		"v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %6\n\tv_pk_add_f32 %2, %2, %8\n\tv_pk_add_f32 %3, %3, %10"
			if constexpr (MODE == 8) {
#define B(i) asm volatile(MUL4("%13") : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), \
			  "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]) : "v"(x), "v"(w));
				REP8(B)
#undef B
			} else {
#define B(i) asm volatile(MUL4("%13") : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), \
			  "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]) : "v"(x), "s"(ws));
				REP8(B)
#undef B
			}
		} else if constexpr (MODE == 10) {	/* the same arithmetic in plain FP32: 4 mul, sub, add, 2 add; four accumulators */
#define B(i) asm volatile("v_mul_f32 %2, %6, %8\n\tv_mul_f32 %3, %7, %9\n\tv_mul_f32 %4, %6, %9\n\tv_mul_f32 %5, %7, %8\n\t" \
			  "v_sub_f32 %2, %2, %3\n\tv_add_f32 %4, %4, %5\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %4" \
			  : "+v"(acc[i & 3].x), "+v"(acc[i & 3].y), "=&v"(r[i].x), "=&v"(r[i].y), "=&v"(r[8 + i].x), "=&v"(r[8 + i].y) \
			  : "v"(x.x), "v"(x.y), "v"(w.x), "v"(w.y));
			REP8(B)
#undef B
		} else if constexpr (MODE == 11) {	/* what k1_fast compiles to: 24 samples from LDS (8 addresses per wave, broadcast), compiler-scheduled */
			v2f a = acc[0];
#pragma unroll
			for (int t = 0; t < 24; ++t) {
				v2f p, q;
				asm("v_pk_mul_f32 %0, %2, %3 op_sel_hi:[0,1]\n\tv_pk_mul_f32 %1, %2, %3 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]"
				    : "=&v"(p), "=&v"(q) : "v"(xp[t]), "v"(acc[1 + (t & 7)]));
				a += (p + q);
			}
			acc[0] = a;
			asm volatile("" ::: "memory");
		} else if constexpr (MODE == 12) {	/* the same, two windows per lane: two independent chains share every LO value */
			v2f a = acc[0], b = acc[9];
#pragma unroll
			for (int t = 0; t < 24; ++t) {
				v2f p, q, p2, q2;
				asm("v_pk_mul_f32 %0, %2, %3 op_sel_hi:[0,1]\n\tv_pk_mul_f32 %1, %2, %3 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]"
				    : "=&v"(p), "=&v"(q) : "v"(xp[t]), "v"(acc[1 + (t & 7)]));
				asm("v_pk_mul_f32 %0, %2, %3 op_sel_hi:[0,1]\n\tv_pk_mul_f32 %1, %2, %3 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]"
				    : "=&v"(p2), "=&v"(q2) : "v"(xp[t + 24]), "v"(acc[1 + (t & 7)]));
				a += (p + q);
				b += (p2 + q2);
			}
			acc[0] = a;
			acc[9] = b;
			asm volatile("" ::: "memory");
		}
	}
	if constexpr (MODE >= 13) {
		typedef float v16f __attribute__((ext_vector_type(16)));
		const float2 *lo = in;			/* wave-uniform */
		const unsigned xa0 = (unsigned)(size_t)(__attribute__((address_space(3))) const float2 *)&xs[(lane & 7) * 24];
		v2f a = acc[0];
		for (int it = 0; it < iters; ++it) {
			const float2 *lp = lo + (it & 7) * 8;
			unsigned xa = xa0;
#pragma unroll 1
			for (int b = 0; b < 3; ++b) {
				v16f w;
				v2f x8[8];
				asm volatile("s_load_dwordx16 %8, %10, 0x0\n\t"
					     "ds_read_b64 %0, %9\n\tds_read_b64 %1, %9 offset:8\n\tds_read_b64 %2, %9 offset:16\n\t"
					     "ds_read_b64 %3, %9 offset:24\n\tds_read_b64 %4, %9 offset:32\n\tds_read_b64 %5, %9 offset:40\n\t"
					     "ds_read_b64 %6, %9 offset:48\n\tds_read_b64 %7, %9 offset:56\n\ts_waitcnt lgkmcnt(0)"
					     : "=&v"(x8[0]), "=&v"(x8[1]), "=&v"(x8[2]), "=&v"(x8[3]), "=&v"(x8[4]), "=&v"(x8[5]), "=&v"(x8[6]), "=&v"(x8[7]), "=&s"(w)
					     : "v"(xa), "s"(lp) : "memory");
				if constexpr (MODE == 13) {
#pragma unroll
					for (int u = 0; u < 8; ++u) {
						v2f t1, t2;
						asm volatile("v_pk_mul_f32 %1, %3, %4 op_sel_hi:[0,1]\n\t"
							     "v_pk_mul_f32 %2, %3, %4 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\ts_nop 0\n\t"
							     "v_pk_add_f32 %1, %1, %2\n\ts_nop 0\n\tv_pk_add_f32 %0, %0, %1"
							     : "+v"(a), "=&v"(t1), "=&v"(t2) : "v"(x8[u]), "s"((v2f){w[2 * u], w[2 * u + 1]}));
					}
				}
				lp += 8;
				xa += 64;
			}
		}
		acc[0] = a;
	}
	const long long t1 = clock64();
	v2f s = {0.f, 0.f};
#pragma unroll
	for (int i = 0; i < 16; ++i)
		s += acc[i] + r[i];
	out[(size_t)blockIdx.x * 64 + lane] = make_float2(s.x, s.y);
	if (lane == 0)
		cyc[blockIdx.x] = t1 - t0;
}

static const char *names[] = {
	"v_pk_mul_f32 x16 independent", "v_pk_add_f32 x16, 8 chains", "v_mul_f32 x32 independent", "v_add_f32 x32, 16 chains",
	"v_fma_f32 x32, 16 chains", "v_pk_fma_f32 x16, 8 chains", "cmac x8, 1 chain, no nops (32 pk ops)", "cmac x8, 1 chain, 2 s_nop each (32 pk ops)",
	"cmac x32, 4 chains, w VGPR (128 pk ops)", "cmac x32, 4 chains, w SGPR (128 pk ops)", "cmac x8 plain f32, 4 chains (64 ops)",
	"k1 window: 24 cmac, x from LDS, compiler (96 pk ops)", "k1 two windows per lane: 48 cmac (192 pk ops)",
	"pp: 3 x (s_load + 8 ds_read + wait + 8 cmac) (96 pk ops)", "pp: 3 x (s_load + 8 ds_read + wait), no cmac (as if 96)",
};
static const int ninstr[] = {16, 16, 32, 32, 32, 16, 32, 32, 128, 128, 64, 96, 192, 96, 96};

template <int MODE> static void run(float2 *out, const float2 *in, long long *cyc, int W)
{
	const int blocks = 1024 * W, iters = 2000;
	hipEvent_t a, b;
	hipEventCreate(&a);
	hipEventCreate(&b);
	hipLaunchKernelGGL(kern<MODE>, dim3(blocks), dim3(64), 0, 0, out, in, cyc, 10);
	hipEventRecord(a, 0);
	hipLaunchKernelGGL(kern<MODE>, dim3(blocks), dim3(64), 0, 0, out, in, cyc, iters);
	hipEventRecord(b, 0);
	hipDeviceSynchronize();
	float ms = 0;
	hipEventElapsedTime(&ms, a, b);
	std::vector<long long> h(blocks);
	hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
	double mean = 0;
	for (long long v : h)
		mean += (double)v;
	mean /= blocks;
	const double per = mean / ((double)iters * ninstr[MODE]);
	printf("%-52s W=%d  wall %8.3f ms  wave cycles/instr %6.2f  per SIMD %5.2f  (wall-clock based per SIMD at 2.4 GHz: %5.2f)\n", names[MODE], W, ms,
	       per, per / W, ms * 1e-3 * 2.4e9 / ((double)iters * ninstr[MODE] * W));
	hipEventDestroy(a);
	hipEventDestroy(b);
}

int main()
{
	float2 *in, *out;
	long long *cyc;
	hipMalloc(&in, 256 * sizeof(float2));
	hipMalloc(&out, 1024 * 8 * 64 * sizeof(float2));
	hipMalloc(&cyc, 1024 * 8 * sizeof(long long));
	std::vector<float2> h(256);
	for (int i = 0; i < 256; ++i)
		h[i] = make_float2(1.0f + 1e-3f * (float)(rand() % 1000), 1e-3f * (float)(rand() % 1000) - 0.5f);
	hipMemcpy(in, h.data(), 256 * sizeof(float2), hipMemcpyHostToDevice);
	for (int W : {1, 2, 4, 6, 8}) {
		run<13>(out, in, cyc, W);
		run<14>(out, in, cyc, W);
		run<0>(out, in, cyc, W);
		run<1>(out, in, cyc, W);
		run<2>(out, in, cyc, W);
		run<3>(out, in, cyc, W);
		run<4>(out, in, cyc, W);
		run<5>(out, in, cyc, W);
		run<6>(out, in, cyc, W);
		run<7>(out, in, cyc, W);
		run<8>(out, in, cyc, W);
		run<9>(out, in, cyc, W);
		run<10>(out, in, cyc, W);
		run<11>(out, in, cyc, W);
		run<12>(out, in, cyc, W);
	}
	return 0;
}
