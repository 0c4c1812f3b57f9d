/*
 * vdl2gpu_kernels.h -- device side of libvdl2gpu.so (gfx950 only).
 *
 * Data layout in HBM
 *   raw      wideband IQ exactly as the SDR delivers it (cu8 / cs16 / cf32 /
 *            real f32), one contiguous run per stream; read ONCE by K1.
 *   lo       per (stream, channel) local-oscillator table, L = SDRINRATE/25000
 *            complex floats, computed on the host with libm (d8psk.c:353-357).
 *   dec      channel-interleaved 84 kS/s frames: frame m = 8 x float2, channel
 *            fastest (64 B per frame), two ping-pong buffers per stream.  K1
 *            appends, K2 reads, K3 moves the unconsumed tail to the other
 *            buffer.  Frame 0 of a buffer is stream time `dec_base`.
 *   state    StreamState (decimator carry) + ChanState (sync detector state:
 *            next evaluation instant, FIR sub-phase, last 68 phases, last two
 *            fit errors) -- the explicit, persistent form of the reference's
 *            stack-resident channel_t (vdlm2.h:56-79).
 *   bursts   ring of vdl2gpu_burst_t records + atomic counter.
 *
 * Arithmetic contract: every float/double operation below is written in the
 * order and width of the reference C expression it replaces and this file is
 * compiled with -ffp-contract=off, so all intermediate values (decimated
 * samples, FIR outputs, phases, fit errors, soft bits) are bit-identical to
 * the reference built -O2 on x86-64, not merely the decisions.
 */
#ifndef VDL2GPU_KERNELS_H
#define VDL2GPU_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vdl2_math.h"
#include "../../include/vdl2gpu.h"

#define VDL2_CS 8		/* channel slots per decimated frame */
#define VDL2_HIST 16		/* frames of history the 17-tap FIR needs */
#define VDL2_NPH 68		/* NBPH*D8DWN, vdlm2.h:54-55 */
#define VDL2_MAXSYM 5456	/* >= ceil((25 + 8*8*255)/3) symbols of the longest burst */
#define VDL2_CARRY_FRAMES 49152	/* >= longest burst (43592 frames) + history + slack */
#define VDL2_PN_BITS 16384 + 64

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

struct StreamState {
	long long dec_base;	/* stream time (84 kS/s index) of frame 0 of the current buffer */
	long long dec_fill;	/* frames present before this push's K1 output */
	long long last_fill;	/* diagnostics: where the last push's output starts */
	long long last_J;
	float2 acc[2][VDL2_CS];	/* integrate-and-dump partial sums carried across pushes */
};

struct ChanState {
	long long pos;		/* stream time of the next WSYNC evaluation */
	int r;			/* FIR sub-phase (channel_t.clk after the -=8), 0..3 */
	float perr, p2err, pfr;	/* channel_t.perr/p2err/pfr */
	float ring[VDL2_NPH];	/* channel_t.Ph in time order, ring[67] newest */
	unsigned long long n_eval, n_trig, n_reject, n_burst, n_defer;
};

struct ChanCfg {
	int chn, Fr, Fo, pad;
};

struct K1Params {
	const void *raw;
	size_t stream_stride;
	int fmt, nbch;
	int sdrclk, L, maxwin;
	int c0, no0, nf0, parity;
	long long N, J;
	const float2 *lo;	/* [S][8][L] */
	float2 *dec;		/* this push's buffer, [S][cap][8] */
	long long cap;
	StreamState *ss;
};

struct K2Params {
	const float2 *dec;
	long long cap;
	int nbch;
	long long J;
	StreamState *ss;
	ChanState *cs;
	const ChanCfg *cfg;
	const uint8_t *pn;
	vdl2gpu_burst_t *recs;
	unsigned *rec_count;
	unsigned rec_cap;
	unsigned *overflow;
};

struct K3Params {
	const float2 *src;
	float2 *dst;
	long long cap;
	int nbch;
	long long J;
	StreamState *ss;
	const ChanState *cs;
};

/* ---- constant data tables (d8psk.h:20-249) as bit patterns ------------- */
#define VDL2_TABLE_BEGIN(name, n) __constant__ uint32_t c_##name[n] = {
#define VDL2_F32(x) x,
#define VDL2_TABLE_END };
#include "vdl2_tables.inc"
#undef VDL2_TABLE_BEGIN
#undef VDL2_F32
#undef VDL2_TABLE_END

/* parity-check columns of the (25,20) header code (data, viterbi.c:29-35) */
__constant__ int c_hcol[25] = { 6, 7, 9, 10, 11, 12, 14, 15, 17, 19, 21, 22, 24, 25, 26, 27, 28, 29, 30, 31,
	16, 8, 4, 2, 1
};

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
	float2 *dec = p.dec + ((size_t)s * p.cap + fill) * VDL2_CS;
	const long long jb = (long long)blockIdx.x * (K1_OPB * K1_PASSES);
	if (blockIdx.x == 0 && tid == 0) {
		ss->last_fill = fill;
		ss->last_J = p.J;
	}
	const int o = tid >> 3, c = tid & 7;
	for (int pass = 0; pass < K1_PASSES; ++pass) {
		const long long jp = jb + (long long)pass * K1_OPB;
		if (jp > p.J)
			break;
		const long long jhi = (jp + K1_OPB - 1 < p.J) ? jp + K1_OPB - 1 : p.J;
		const long long in_lo = (jp == 0) ? 0 : k1_win_end(jp - 1, p.sdrclk, p.c0) + 1;
		const long long in_hi = (jhi == p.J) ? p.N - 1 : k1_win_end(jhi, p.sdrclk, p.c0);
		const int cnt = (int)(in_hi - in_lo + 1);
		__syncthreads();
		for (int i = tid; i < cnt; i += K1_THREADS)
			xs[i] = k1_load<FMT>(raw, in_lo + i);
		__syncthreads();
		const long long j = jp + o;
		if (j <= p.J && c < p.nbch) {
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
				dec[j * VDL2_CS + c] = make_float2(dre / fn, dim / fn);
			}
		}
	}
}

/* ======================================================================= K2
 * Demodulator: one workgroup per VDL channel.
 *   search  (time-parallel): for the next <=K2_THREADS evaluation instants of
 *           the idle detector compute the filtered phase (d8psk.c:219-230),
 *           then the 17-point sync-word fit error (d8psk.c:257-289), then find
 *           the first instant where `perr < 4 && err > perr` (d8psk.c:292).
 *   burst   (parallel over symbols): one-shot timing estimate (d8psk.c:303-306),
 *           header symbols -> soft bits -> (25,20) Viterbi in one wavefront
 *           (viterbi.c), then every payload symbol's phase, differential
 *           slice + Grey soft tables + descramble (d8psk.c:54-65, 211-217,
 *           321-331) and the column-major de-interleave (d8psk.c:117-206)
 *           as a closed-form scatter.
 */
#ifndef K2_THREADS
#define K2_THREADS 512
#endif

__device__ __forceinline__ float d_tab(const uint32_t *t, int i)
{
	return __uint_as_float(t[i]);
}

/* filteredphase(), d8psk.c:219-230: x points at frame n-16 (channel column) */
__device__ __forceinline__ float k2_fir_phase(const float2 *x, int tap0)
{
	float sr = 0.0f, si = 0.0f;
	for (int i = tap0, j = 0; i < 65; i += 4, ++j) {
		const float m = d_tab(c_mflt, i);
		const float2 v = x[(size_t)j * VDL2_CS];
		sr += v.x * m;
		si += v.y * m;
	}
	return vdl2_atan2f(si, sr);
}

/* d8psk.c:257-289: ph[0], ph[4], ... ph[64] are the 17 phases one symbol apart */
__device__ __forceinline__ float k2_sync_metric(const float *ph, float *slope)
{
	float pr[17];
	float pu = 0.0f;
	float pv = ph[0] - d_tab(c_sw, 0);
	float mean = pv;
	pr[0] = pv;
#pragma unroll
	for (int l = 1; l < 17; ++l) {
		const float pc = ph[4 * l] - d_tab(c_sw, l);
		const float pd = pc - pv;
		pv = pc;
		if ((double)pd > M_PI)
			pu = (float)((double)pu - 2 * M_PI);
		else if ((double)pd < -M_PI)
			pu = (float)((double)pu + 2 * M_PI);
		pr[l] = pc + pu;
		mean += pr[l];
	}
	mean /= 17.0f;
	float fr = 0.0f;
#pragma unroll
	for (int l = 0; l < 17; ++l) {
		pr[l] -= mean;
		fr += pr[l] * (float)(l - 8);
	}
	fr /= 408.0f;
	float err = 0.0f;
#pragma unroll
	for (int l = 0; l < 17; ++l) {
		const float e = pr[l] - (float)(l - 8) * fr;
		err += e * e;
	}
	*slope = fr;
	return err;
}

/* differential slice of one symbol -> Grey table index (d8psk.c:213, 323-327) */
__device__ __forceinline__ int k2_grey_index(float p, float pprev, float df)
{
	float d = (p - pprev) - df;
	if ((double)d > M_PI)
		d = (float)((double)d - 2 * M_PI);
	if ((double)d < -M_PI)
		d = (float)((double)d + 2 * M_PI);
	int i = (int)roundf((float)(128.0 * (double)d / M_PI + 128.0));
	return i < 0 ? 0 : (i > 256 ? 256 : i);
}

__device__ __forceinline__ float k2_soft_bit(int idx, int which, int pnbit)
{
	const float v = d_tab(which == 0 ? c_grey1 : (which == 1 ? c_grey2 : c_grey3), idx);
	return pnbit ? (float)(1.0 - (double)v) : v;	/* descrambler, d8psk.c:60-63 */
}

struct K2Shared {
	float pbuf[VDL2_NPH + K2_THREADS];	/* phases: [0,68) = history ring */
	float errs[K2_THREADS + 2];		/* errs[t+2] = err of eval t; [0],[1] = p2err, perr */
	float frs[K2_THREADS + 1];		/* frs[t+1] = slope of eval t; [0] = pfr */
	float psym[VDL2_MAXSYM];		/* burst symbol phases */
	uint8_t hbits[VDL2_MAXSYM];		/* 3 descrambled hard bits per symbol */
	float hsoft[25];			/* descrambled header soft bits */
	uint8_t vbk[26][32], vbv[26][32];	/* Viterbi back pointers / decided bits */
	int first;
	int ctl[16];
	float fctl[8];
};

__global__ __launch_bounds__(K2_THREADS)
void k2_demod(K2Params p)
{
	__shared__ K2Shared sh;
	const int tid = threadIdx.x;
	const int c = blockIdx.x, s = blockIdx.y;
	if (c >= p.nbch)
		return;
	ChanState *cs = p.cs + (size_t)s * VDL2_CS + c;
	const StreamState *ss = p.ss + s;
	const long long dec_base = ss->dec_base;
	const long long avail_end = dec_base + ss->dec_fill + p.J;
	const float2 *x0 = p.dec + (size_t)s * p.cap * VDL2_CS + c;	/* frame f, channel c: x0[f*8] */

	long long pos = cs->pos;
	int r = cs->r;
	unsigned long long n_eval = 0, n_trig = 0, n_reject = 0, n_burst = 0, n_defer = 0;
	if (tid < VDL2_NPH)
		sh.pbuf[tid] = cs->ring[tid];
	if (tid == 0) {
		sh.errs[0] = cs->p2err;
		sh.errs[1] = cs->perr;
		sh.frs[0] = cs->pfr;
	}
	__syncthreads();

	for (;;) {
		long long rem = (avail_end - pos + 1) / 2;
		const int nev = rem > K2_THREADS ? K2_THREADS : (int)rem;
		if (nev <= 0)
			break;
		/* ---- search window: evaluations at pos, pos+2, ... */
		if (tid < nev) {
			const long long n = pos + 2 * tid;
			sh.pbuf[VDL2_NPH + tid] = k2_fir_phase(x0 + (size_t)(n - VDL2_HIST - dec_base) * VDL2_CS, r);
		}
		if (tid == 0)
			sh.first = 0x7fffffff;
		__syncthreads();
		if (tid < nev) {
			float fr;
			const float err = k2_sync_metric(&sh.pbuf[tid + 4], &fr);
			sh.errs[tid + 2] = err;
			sh.frs[tid + 1] = fr;
		}
		__syncthreads();
		if (tid < nev) {
			const float perr = sh.errs[tid + 1];
			if (perr < 4.0f && sh.errs[tid + 2] > perr)
				atomicMin(&sh.first, tid);
		}
		__syncthreads();
		const int ts = sh.first;
		if (ts == 0x7fffffff) {
			/* no trigger: commit the whole window */
			float keep = 0.0f;
			if (tid < VDL2_NPH)
				keep = sh.pbuf[nev + tid];
			const float e0 = sh.errs[nev], e1 = sh.errs[nev + 1], f0 = sh.frs[nev];
			__syncthreads();
			if (tid < VDL2_NPH)
				sh.pbuf[tid] = keep;
			if (tid == 0) {
				sh.errs[0] = e0;
				sh.errs[1] = e1;
				sh.frs[0] = f0;
			}
			pos += 2LL * nev;
			n_eval += nev;
			__syncthreads();
			continue;
		}
		/* ---- sync trigger at evaluation ts (stream time nstar) */
		const long long nstar = pos + 2LL * ts;
		if (tid == 0) {
			const float p2err = sh.errs[ts], perr = sh.errs[ts + 1], err = sh.errs[ts + 2];
			/* parabolic interpolation of the error minimum, d8psk.c:303-305 */
			const float of = 4.0f * (p2err - 4.0f * perr + 3.0f * err) / (p2err - 2.0f * perr + err);
			int clk0 = (int)roundf(of);
			if (clk0 < 0)
				clk0 = 0;	/* unreachable for finite inputs: of is in [4,12] */
			if (clk0 > 68)
				clk0 = 68;
			int j0 = (32 - clk0 + 3) / 4;
			if (j0 < 1)
				j0 = 1;
			sh.ctl[0] = clk0;
			sh.ctl[1] = j0;
			sh.ctl[2] = clk0 + 4 * j0 - 32;	/* sub-phase during and after the burst */
			sh.fctl[0] = sh.frs[ts];	/* df = pfr, d8psk.c:301 */
		}
		__syncthreads();
		const int clk0 = sh.ctl[0], j0 = sh.ctl[1], rb = sh.ctl[2];
		const float df = sh.fctl[0];
		const long long nsym0 = nstar + j0;	/* stream time of burst symbol 0 */
		bool defer = (nsym0 + 64 >= avail_end);	/* 9 header symbols must be present */
		int accepted = 0, nbrow = 0, nlbyte = 0, nsym = 0;
		int nd_rows = 0, nd_last = 0, nf_rows = 0, nf_last = 0, ND = 0, NF = 0;
		if (!defer) {
			if (tid < 9)
				sh.psym[tid] = k2_fir_phase(x0 + (size_t)(nsym0 + 8 * tid - VDL2_HIST - dec_base) * VDL2_CS, rb);
			if (tid == 9)
				sh.fctl[1] = k2_fir_phase(x0 + (size_t)(nstar - VDL2_HIST - dec_base) * VDL2_CS, clk0);	/* P1 */
			__syncthreads();
			if (tid < 25) {
				const int k = tid / 3;
				const float pprev = k ? sh.psym[k - 1] : sh.fctl[1];
				const int idx = k2_grey_index(sh.psym[k], pprev, df);
				float v = k2_soft_bit(idx, tid % 3, p.pn[tid]);
				if (tid < 3)
					v = 0.0f;	/* reserved bits forced, d8psk.c:81-82 */
				sh.hsoft[tid] = v;
			}
			__syncthreads();
			if (tid < 64) {
				/* (25,20) code, 32 syndrome states = 32 lanes (viterbi.c:46-78).
				 * Target state t has two candidates: bit 0 from state t, bit 1
				 * from state t^H[n]; the reference visits sources in ascending
				 * order and replaces a survivor only by a strictly larger metric. */
				const int t = tid & 31;
				double pb = (t == 0) ? 1.0 : 0.0;
				for (int n = 0; n < 25; ++n) {
					const double v = (double)sh.hsoft[n];
					const int src1 = t ^ c_hcol[n];
					const double pb1 = __shfl(pb, src1, 32);
					const double m0 = pb * (1.0 - v);
					const double m1 = pb1 * v;
					const bool has0 = (pb != 0.0), has1 = (pb1 != 0.0);
					double nv = 0.0;
					int nb = 0, ns = 0;
					if (t < src1) {
						if (has0 && m0 > nv) { nv = m0; nb = 0; ns = t; }
						if (has1 && m1 > nv) { nv = m1; nb = 1; ns = src1; }
					} else {
						if (has1 && m1 > nv) { nv = m1; nb = 1; ns = src1; }
						if (has0 && m0 > nv) { nv = m0; nb = 0; ns = t; }
					}
					if (tid < 32) {
						sh.vbk[n + 1][t] = (uint8_t)ns;
						sh.vbv[n + 1][t] = (uint8_t)nb;
					}
					pb = nv;
				}
			}
			__syncthreads();
			if (tid == 0) {
				unsigned word = 0, mask = 1;
				int st = 0;
				for (int n = 25; n > 0; --n) {
					if (sh.vbv[n][st])
						word |= mask;
					st = sh.vbk[n][st];
					mask <<= 1;
				}
				word >>= 5;	/* drop the 5 parity bits, d8psk.c:90 */
				unsigned len = 0;
				for (int i = 0; i < 17; ++i)
					len |= ((word >> i) & 1u) << (16 - i);	/* reversebits(.,17) */
				const int nbr = (int)(len / 1992u) + 1;
				const int nlb = (int)((len % 1992u + 7u) / 8u);
				sh.ctl[3] = (len >= 96u && nbr <= 8) ? 1 : 0;
				sh.ctl[4] = nbr;
				sh.ctl[5] = nlb;
			}
			__syncthreads();
			accepted = sh.ctl[3];
			nbrow = sh.ctl[4];
			nlbyte = sh.ctl[5];
			if (accepted) {
				/* receiver's byte schedule, d8psk.c:117-206 */
				nd_rows = nbrow;
				nd_last = nlbyte ? nlbyte : 249;	/* nlbyte==0: zero-fill loop is skipped */
				ND = (nbrow - 1) * 249 + nd_last;
				if (nlbyte <= 2) {
					nf_rows = nbrow - 1;
					nf_last = 6;
				} else {
					nf_rows = nbrow;
					nf_last = (nlbyte <= 30) ? 2 : (nlbyte <= 67 ? 4 : 6);
				}
				NF = (nf_rows - 1) * 6 + nf_last;
				if (nf_rows <= 0)
					NF = 0;
				nsym = (25 + 8 * (ND + NF) + 2) / 3;
				if (nsym0 + 8LL * (nsym - 1) >= avail_end)
					defer = true;
			}
		}
		if (defer) {
			/* the burst is not completely inside the data we hold: commit the
			 * evaluations before the trigger and retry on the next push */
			float keep = 0.0f;
			if (tid < VDL2_NPH)
				keep = sh.pbuf[ts + tid];
			const float e0 = sh.errs[ts], e1 = sh.errs[ts + 1], f0 = sh.frs[ts];
			__syncthreads();
			if (tid < VDL2_NPH)
				sh.pbuf[tid] = keep;
			if (tid == 0) {
				sh.errs[0] = e0;
				sh.errs[1] = e1;
				sh.frs[0] = f0;
			}
			pos += 2LL * ts;
			n_eval += ts;
			n_defer++;
			__syncthreads();
			break;
		}
		n_trig++;
		long long nlast;
		if (!accepted) {
			n_reject++;
			nlast = nsym0 + 64;	/* state returns to WSYNC on the 25th bit (symbol 8) */
		} else {
			nlast = nsym0 + 8LL * (nsym - 1);
			for (int k = tid; k < nsym; k += K2_THREADS)
				sh.psym[k] = k2_fir_phase(x0 + (size_t)(nsym0 + 8LL * k - VDL2_HIST - dec_base) * VDL2_CS, rb);
			if (tid == 0) {
				unsigned slot = atomicAdd(p.rec_count, 1u);
				if (slot >= p.rec_cap) {
					atomicAdd(p.overflow, 1u);
					slot = 0xffffffffu;
				}
				sh.ctl[6] = (int)slot;
			}
			__syncthreads();
			const unsigned slot = (unsigned)sh.ctl[6];
			for (int k = tid; k < nsym; k += K2_THREADS) {
				const float pprev = k ? sh.psym[k - 1] : sh.fctl[1];
				const int idx = k2_grey_index(sh.psym[k], pprev, df);
				int hb = 0;
#pragma unroll
				for (int i = 0; i < 3; ++i) {
					const float v = k2_soft_bit(idx, i, p.pn[3 * k + i]);
					if ((double)v > 0.5)
						hb |= 1 << i;
				}
				sh.hbits[k] = (uint8_t)hb;
			}
			vdl2gpu_burst_t *rec = (slot != 0xffffffffu) ? p.recs + slot : nullptr;
			if (rec) {
				uint32_t *w = reinterpret_cast<uint32_t *>(&rec->data[0][0]);
				for (int i = tid; i < VDL2GPU_MAXROWS * VDL2GPU_ROWLEN / 4; i += K2_THREADS)
					w[i] = 0u;
			}
			__syncthreads();
			if (rec) {
				for (int b = tid; b < ND + NF; b += K2_THREADS) {
					const int q0 = 25 + 8 * b;
					unsigned byte = 0;
#pragma unroll
					for (int i = 0; i < 8; ++i) {
						const int q = q0 + i;
						byte |= (unsigned)((sh.hbits[q / 3] >> (q % 3)) & 1) << i;
					}
					/* column-major walk with a short last row -> (row, col) */
					int row, col;
					if (b < ND) {
						const int full = nd_last * nd_rows;
						if (b < full) {
							col = b / nd_rows;
							row = b % nd_rows;
						} else {
							const int bb = b - full;
							col = nd_last + bb / (nd_rows - 1);
							row = bb % (nd_rows - 1);
						}
					} else {
						const int bf = b - ND;
						const int full = nf_last * nf_rows;
						if (bf < full) {
							col = bf / nf_rows;
							row = bf % nf_rows;
						} else {
							const int bb = bf - full;
							col = nf_last + bb / (nf_rows - 1);
							row = bb % (nf_rows - 1);
						}
						col += 249;
					}
					rec->data[row][col] = (uint8_t)byte;
				}
				if (tid == 0) {
					const ChanCfg cf = p.cfg[(size_t)s * VDL2_CS + c];
					rec->stream = s;
					rec->chn = cf.chn;
					rec->Fr = cf.Fr;
					rec->nbrow = nbrow;
					rec->nlbyte = nlbyte;
					rec->df = df;
					rec->ppm = 0.0f;	/* host: d8psk.c:302 needs libm double math */
					rec->trig_dec = nstar;
					rec->end_dec = nlast;
					rec->trig_sample = 0;
					rec->end_sample = 0;
				}
			}
			n_burst++;
		}
		/* back to the idle detector: ring keeps the phases up to the trigger
		 * evaluation (Ph is not written during a burst), errors re-armed
		 * (d8psk.c:308), sub-phase sticks at rb */
		{
			float keep = 0.0f;
			if (tid < VDL2_NPH)
				keep = sh.pbuf[ts + 1 + tid];
			const float f0 = sh.frs[ts];
			__syncthreads();
			if (tid < VDL2_NPH)
				sh.pbuf[tid] = keep;
			if (tid == 0) {
				sh.errs[0] = 500.0f;
				sh.errs[1] = 500.0f;
				sh.frs[0] = f0;
			}
			n_eval += ts + 1;
			pos = nlast + 2;
			r = rb;
			__syncthreads();
		}
	}
	/* persist */
	if (tid < VDL2_NPH)
		cs->ring[tid] = sh.pbuf[tid];
	if (tid == 0) {
		cs->pos = pos;
		cs->r = r;
		cs->p2err = sh.errs[0];
		cs->perr = sh.errs[1];
		cs->pfr = sh.frs[0];
		cs->n_eval += n_eval;
		cs->n_trig += n_trig;
		cs->n_reject += n_reject;
		cs->n_burst += n_burst;
		cs->n_defer += n_defer;
	}
}

/* ======================================================================= K3
 * Move the frames no channel has consumed yet (plus FIR history) to the front
 * of the other ping-pong buffer and rebase stream time.  Normally ~20 frames;
 * up to one full burst when a channel is waiting for the end of a long burst.
 */
#define K3_THREADS 1024
__global__ __launch_bounds__(K3_THREADS)
void k3_compact(K3Params p)
{
	const int s = blockIdx.x;
	StreamState *ss = p.ss + s;
	__shared__ long long sh_base, sh_fill;
	if (threadIdx.x == 0) {
		long long mn = 0x7fffffffffffffffLL;
		for (int c = 0; c < p.nbch; ++c) {
			const long long q = p.cs[(size_t)s * VDL2_CS + c].pos;
			mn = q < mn ? q : mn;
		}
		sh_base = ss->dec_base;
		sh_fill = ss->dec_fill + p.J;
		const long long end = sh_base + sh_fill;
		long long nb = mn - VDL2_HIST;
		if (nb > end - VDL2_HIST)
			nb = end - VDL2_HIST;	/* always keep the FIR history */
		if (nb < sh_base)
			nb = sh_base;
		ss->dec_base = nb;
		ss->dec_fill = end - nb;
	}
	__syncthreads();
	const long long shift = ss->dec_base - sh_base;
	const long long keep = ss->dec_fill;
	const float4 *src = reinterpret_cast<const float4 *>(p.src + ((size_t)s * p.cap + shift) * VDL2_CS);
	float4 *dst = reinterpret_cast<float4 *>(p.dst + (size_t)s * p.cap * VDL2_CS);
	const long long n4 = keep * (VDL2_CS / 2);
	for (long long i = threadIdx.x; i < n4; i += K3_THREADS)
		dst[i] = src[i];
}

__global__ void k_atan2f(const float *y, const float *x, float *out, size_t n)
{
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n)
		out[i] = vdl2_atan2f(y[i], x[i]);
}

#endif
