/* vdl2gpu_k1.h -- K1: channeliser kernels.  Part of the device side of libvdl2gpu.so; included by vdl2gpu_kernels.h only. */
#ifndef VDL2GPU_K1_H
#define VDL2GPU_K1_H

/* ======================================================================= K1
 * Channeliser: ingest conversion (rtl.c:285-292) + complex mix with the LO
 * table + integrate-and-dump to 84 kS/s (d8psk.c:366-381), all channels of a
 * stream from ONE read of the wideband samples.
 *
 * The dump schedule has a closed form: with c0 = decimator
 * clock at the start of the push, local output j ends at local input
 *     le(j) = ceil(((j+1)*SDRCLK - c0) / 21) - 1
 * so every output window is independent and the whole push is time-parallel.
 * Each lane owns one (output window, channel) and adds its 23/24 (2 MS/s) ..
 * 119/120 (10 MS/s) products in stream order, which keeps the float sum
 * identical to the reference's serial loop.  The window straddling a push
 * boundary continues from the partial sum carried in StreamState.acc.
 */
#define K1_THREADS 256
#define K1_OPB 32		/* outputs per pass (256 threads / 8 channel lanes) */
#define K1_PASSES 8

__device__ __forceinline__ long long k1_win_end(long long j, int sdrclk, int c0)
{
	return ((j + 1) * (long long)sdrclk - c0 + 20) / 21 - 1;
}

template <int FMT> __device__ __forceinline__ float2 k1_load(const char *raw, long long i)
{
	if (FMT == VDL2GPU_FMT_CU8) {
		const uchar2 b = reinterpret_cast<const uchar2 *>(raw)[i];
		return make_float2((float)b.x - (float)127.37, (float)b.y - (float)127.37);
	} else if (FMT == VDL2GPU_FMT_CS16) {
		const short2 v = reinterpret_cast<const short2 *>(raw)[i];
		return make_float2((float)v.x, (float)v.y);
	} else if (FMT == VDL2GPU_FMT_CF32) {
		return reinterpret_cast<const float2 *>(raw)[i];
	} else {
		return make_float2(reinterpret_cast<const float *>(raw)[i], 0.0f);
	}
}

template <int FMT> __global__ __launch_bounds__(K1_THREADS)
void k1_channelise(K1Params p)
{
	extern __shared__ float2 k1_smem[];
	float2 *lo_s = k1_smem;					/* [(L+maxwin)][8] */
	float2 *xs = k1_smem + (size_t)(p.L + p.maxwin) * VDL2_CS;	/* [32*maxwin] */
	const int tid = threadIdx.x;
	const int s = blockIdx.y;
	const float2 *lo = p.lo + (size_t)s * VDL2_CS * p.L;
	for (int idx = tid; idx < (p.L + p.maxwin) * VDL2_CS; idx += K1_THREADS) {
		const int n = idx >> 3, c = idx & 7;
		lo_s[idx] = lo[c * p.L + (n % p.L)];
	}
	const char *raw = (const char *)p.raw + (size_t)s * p.stream_stride;
	StreamState *ss = p.ss + s;
	const long long fill = ss->dec_fill;
	const long long jb = p.jbeg + (long long)blockIdx.x * (K1_OPB * K1_PASSES);
	if (blockIdx.x == 0 && tid == 0 && p.jbeg == 0) {
		ss->last_fill = fill;
		ss->last_J = p.J;
	}
	/* lane = (channel, output): 32 consecutive outputs of one channel per half-wave,
	 * so a plane store is a 256-byte run */
	const int o = tid & 31, c = tid >> 5;
	float2 *dec = p.dec + ((size_t)s * VDL2_CS + c) * p.cap + fill;
	for (int pass = 0; pass < K1_PASSES; ++pass) {
		const long long jp = jb + (long long)pass * K1_OPB;
		if (jp > p.jend)
			break;
		const long long jhi = (jp + K1_OPB - 1 < p.jend) ? jp + K1_OPB - 1 : p.jend;
		const long long in_lo = (jp == 0) ? 0 : k1_win_end(jp - 1, p.sdrclk, p.c0) + 1;
		const long long in_hi = (jhi == p.J) ? p.N - 1 : k1_win_end(jhi, p.sdrclk, p.c0);
		const int cnt = (int)(in_hi - in_lo + 1);
		__syncthreads();
		for (int i = tid; i < cnt; i += K1_THREADS)
			xs[i] = k1_load<FMT>(raw, in_lo + i);
		__syncthreads();
		const long long j = jp + o;
		if (j <= p.jend && c < p.nbch) {
			const long long a = (j == 0) ? 0 : k1_win_end(j - 1, p.sdrclk, p.c0) + 1;
			const long long b = (j == p.J) ? p.N - 1 : k1_win_end(j, p.sdrclk, p.c0);
			const int n = (int)(b - a + 1);
			const float2 *xp = xs + (int)(a - in_lo);
			const float2 *wp = lo_s + (size_t)((p.no0 + a) % p.L) * VDL2_CS + c;
			float dre = 0.0f, dim = 0.0f;
			int nf = n;
			if (j == 0) {
				const float2 cy = ss->acc[p.parity][c];
				dre = cy.x;
				dim = cy.y;
				nf += p.nf0;
			}
			if (FMT == VDL2GPU_FMT_F32R) {
				for (int t = 0; t < n; ++t) {
					const float x = xp[t].x;
					const float2 w = wp[t * VDL2_CS];
					dre += x * w.x;
					dim += x * w.y;
				}
			} else {
				for (int t = 0; t < n; ++t) {
					const float2 x = xp[t];
					const float2 w = wp[t * VDL2_CS];
					const float pr = x.x * w.x - x.y * w.y;
					const float pi = x.x * w.y + x.y * w.x;
					dre += pr;
					dim += pi;
				}
			}
			if (j == p.J) {
				ss->acc[p.parity ^ 1][c] = make_float2(dre, dim);
			} else {
				const float fn = (float)nf;
				dec[j] = make_float2(dre / fn, dim / fn);
			}
		}
	}
}


/* ---- K1 fast path: SDRINRATE 2 MS/s (SDRCLK 500, LO period 80) ------------------------
 * The dump schedule and the LO phase repeat every 2000 inputs = 84 outputs (1 ms of air
 * time).  One WAVEFRONT owns 8 consecutive windows of the period x 8 channels (lane =
 * window*8 + channel) for many periods.  A lane's 23/24 LO values never change, so they
 * live in VGPRs; the ~190 samples the wave's 8 windows cover are fetched by the wave itself
 * (3 coalesced loads per lane), converted once, and parked in a private double-buffered LDS
 * slice, from which each sample is read once per window and broadcast to the 8 channel
 * lanes.  Inner loop: 1 LDS read + 8 VALU ops per sample and channel.  No workgroup
 * barrier anywhere: wavefronts never wait for each other, 16 of them per CU hide HBM latency.
 * 84 = 10*8 + 4, so 11 wave roles cover a period (the last one half empty). */
typedef float v2f __attribute__((ext_vector_type(2)));

/* (re, im) += x * w for complex x, w with the reference's operation order
 *   pr = x.re*w.re - x.im*w.im;  pi = x.re*w.im + x.im*w.re;  acc += (pr, pi)
 * as four packed-FP32 VALU ops (gfx950 issues plain FP32 at half the packed rate):
 *   a = (x.re*w.re, x.re*w.im)          v_pk_mul_f32, op_sel picks x.re twice
 *   b = (x.im*(-w.im), x.im*w.re)       v_pk_mul_f32, halves of w swapped, low lane negated
 *   acc += (a + b)                      2 x v_pk_add_f32
 * a.lo + b.lo = x.re*w.re + (-(x.im*w.im)) is bit-identical to the subtraction. */
__device__ __forceinline__ void k1_cmac(v2f &acc, v2f x, v2f w)
{
	v2f a, b;
	asm("v_pk_mul_f32 %0, %2, %3 op_sel_hi:[0,1]\n\t"
	    "v_pk_mul_f32 %1, %2, %3 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]"
	    : "=&v"(a), "=&v"(b)
	    : "v"(x), "v"(w));
	acc += (a + b);
}

#define K1F_THREADS 64
#define K1F_PB 32		/* periods per wavefront */
#define K1F_DEPTH 4		/* periods of raw samples in flight per wavefront (registers) */
#define K1F_PER_IN 2000
#define K1F_PER_OUT 84
#define K1F_ROLES 11
#define K1F_SLICE 192		/* >= 8 windows x 24 samples */

template <int FMT> struct K1Raw;
template <> struct K1Raw<VDL2GPU_FMT_CU8> { typedef unsigned short T; };
template <> struct K1Raw<VDL2GPU_FMT_CS16> { typedef unsigned int T; };
template <> struct K1Raw<VDL2GPU_FMT_CF32> { typedef float2 T; };
template <> struct K1Raw<VDL2GPU_FMT_F32R> { typedef float T; };

template <int FMT> __device__ __forceinline__ typename K1Raw<FMT>::T k1_raw_load(const char *raw, long long i)
{
	return reinterpret_cast<const typename K1Raw<FMT>::T *>(raw)[i];
}

template <int FMT> __device__ __forceinline__ float2 k1_raw_cvt(typename K1Raw<FMT>::T v)
{
	if constexpr (FMT == VDL2GPU_FMT_CU8) {
		return make_float2((float)(v & 0xffu) - (float)127.37, (float)(v >> 8) - (float)127.37);
	} else if constexpr (FMT == VDL2GPU_FMT_CS16) {
		return make_float2((float)(short)(v & 0xffffu), (float)(short)(v >> 16));
	} else if constexpr (FMT == VDL2GPU_FMT_CF32) {
		return v;
	} else {
		return make_float2(v, 0.0f);
	}
}

template <int FMT> __global__ __launch_bounds__(K1F_THREADS)
void k1_fast(K1Params p)
{
	typedef typename K1Raw<FMT>::T raw_t;
	__shared__ float2 xs[K1F_SLICE];
	const int lane = threadIdx.x;
	const int s = blockIdx.y;
	const int g = blockIdx.x % K1F_ROLES;
	/* Wave group w = blockIdx.x / ROLES handles periods per_lo + w, + w + NW, + w + 2 NW, .. (NW =
	 * number of wave groups): at every loop iteration the whole grid reads one contiguous band of
	 * NW periods and writes one contiguous band of each plane, which keeps HBM pages open, instead
	 * of every wave streaming through its own distant range. */
	const long long nw = (long long)(gridDim.x / K1F_ROLES);
	const long long wgrp = (long long)(blockIdx.x / K1F_ROLES);
	const long long pp0 = p.per_lo + wgrp;
	if (wgrp >= p.per_n)
		return;
	const int np = (int)((p.per_n - wgrp + nw - 1) / nw);	/* periods pp0 + q*nw, q < np */
	const long long pstride = (long long)K1F_PER_IN * nw;	/* samples between this wave's periods */
	const int kk = lane >> 3, c = lane & 7;
	const int k = g * 8 + kk;
	const bool active = (k < K1F_PER_OUT) && (c < p.nbch);
	const char *raw = (const char *)p.raw + (size_t)s * p.stream_stride;
	const long long fill = p.ss[s].dec_fill;
	/* slice of this wave in period pp0: from the first sample of window 8g to the last of window 8g+7 */
	const long long j0 = pp0 * K1F_PER_OUT + g * 8;		/* >= 84 */
	const int klast = (g * 8 + 7 < K1F_PER_OUT) ? 7 : (K1F_PER_OUT - 1 - g * 8);
	const long long sbase = k1_win_end(j0 - 1, p.sdrclk, p.c0) + 1;
	const int slen = (int)(k1_win_end(j0 + klast, p.sdrclk, p.c0) - sbase + 1);
	int off = 0, nwin = 0;
	v2f w[24];
#pragma unroll
	for (int t = 0; t < 24; ++t)
		w[t] = (v2f){0.0f, 0.0f};
	if (active) {
		const long long j = j0 + kk;
		const long long a = k1_win_end(j - 1, p.sdrclk, p.c0) + 1;
		const long long b = k1_win_end(j, p.sdrclk, p.c0);
		off = (int)(a - sbase);
		nwin = (int)(b - a + 1);
		int ph = (int)((p.no0 + a) % 80);
		const float2 *lo = p.lo + ((size_t)s * VDL2_CS + c) * 80;
#pragma unroll
		for (int t = 0; t < 24; ++t) {
			const float2 q = lo[ph];
			w[t] = (v2f){q.x, q.y};
			ph = (ph + 1 == 80) ? 0 : ph + 1;
		}
	}
	const float fn = (float)nwin;
	const float rfn = 1.0f / (nwin ? fn : 1.0f);	/* RN(1/nf) for the exact FMA division below */
	float2 *dec = p.dec + ((size_t)s * VDL2_CS + c) * p.cap + fill + pp0 * K1F_PER_OUT + k;
	/* lanes fetch samples lane, lane+64, lane+128 of the slice (clamped: the tail lanes of the
	 * last load re-read the last sample instead of branching) */
	int li[3];
#pragma unroll
	for (int u = 0; u < 3; ++u) {
		const int i = lane + u * 64;
		li[u] = i < slen ? i : slen - 1;
	}
	raw_t rr[K1F_DEPTH][3];
#pragma unroll
	for (int d = 0; d < K1F_DEPTH; ++d)
#pragma unroll
		for (int u = 0; u < 3; ++u)
			rr[d][u] = k1_raw_load<FMT>(raw, sbase + pstride * (d < np ? d : np - 1) + li[u]);
	for (int q0 = 0; q0 < np; q0 += K1F_DEPTH) {
#pragma unroll
		for (int d = 0; d < K1F_DEPTH; ++d) {
			const int q = q0 + d;
			if (q < np) {
				/* period q: registers -> float -> LDS slice, then refill the registers
				 * with period q+DEPTH so that DEPTH periods stay in flight.  (A second LDS
				 * slice to take this write off the mixer's critical path measured slower.) */
#pragma unroll
				for (int u = 0; u < 3; ++u)
					xs[lane + u * 64] = k1_raw_cvt<FMT>(rr[d][u]);
				const int qn = (q + K1F_DEPTH < np) ? q + K1F_DEPTH : np - 1;
#pragma unroll
				for (int u = 0; u < 3; ++u)
					rr[d][u] = k1_raw_load<FMT>(raw, sbase + pstride * qn + li[u]);
				__syncthreads();	/* single-wave workgroup: LDS write -> read ordering */
				if (active) {
					const v2f *xp = reinterpret_cast<const v2f *>(&xs[off]);
					v2f acc = {0.0f, 0.0f};
					if (FMT == VDL2GPU_FMT_F32R) {
#pragma unroll
						for (int t = 0; t < 23; ++t) {
							const float x = xp[t].x;
							acc += (v2f){x, x} * w[t];
						}
						if (nwin == 24) {
							const float x = xp[23].x;
							acc += (v2f){x, x} * w[23];
						}
					} else {
#pragma unroll
						for (int t = 0; t < 23; ++t)
							k1_cmac(acc, xp[t], w[t]);
						if (nwin == 24)
							k1_cmac(acc, xp[23], w[23]);
					}
					/* D /= nf (d8psk.c:377).  q0 = x*RN(1/nf); q = fma(fma(-q0, nf, x), RN(1/nf), q0)
					 * is the correctly rounded quotient for every |x| >= 1e-30 (exhaustively
					 * checked for nf = 23, 24: tests/ctests/div_check.c); below that, and only
					 * then, the plain IEEE division is used */
					float qr, qi;
					if (__all(fabsf(acc.x) >= 1e-30f && fabsf(acc.y) >= 1e-30f)) {
						const float q0r = acc.x * rfn, q0i = acc.y * rfn;
						qr = fmaf(fmaf(-q0r, fn, acc.x), rfn, q0r);
						qi = fmaf(fmaf(-q0i, fn, acc.y), rfn, q0i);
					} else {
						qr = acc.x / fn;
						qi = acc.y / fn;
					}
					dec[(long long)q * K1F_PER_OUT * nw] = make_float2(qr, qi);
				}
				__syncthreads();	/* reads done before the slice is overwritten */
			}
		}
	}
}

#endif
