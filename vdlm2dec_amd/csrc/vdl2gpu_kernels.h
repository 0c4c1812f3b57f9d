/*
 * vdl2gpu_kernels.h -- device side of libvdl2gpu.so (gfx950 only).
 *
 * Data layout in HBM
 *   raw      wideband IQ exactly as the SDR delivers it (cu8 / cs16 / cf32 /
 *            real f32), one contiguous run per stream; read ONCE by K1.
 *   lo       per (stream, channel) local-oscillator table, L = SDRINRATE/25000
 *            complex floats, computed on the host with libm (d8psk.c:353-357).
 *   dec      84 kS/s channel planes: plane (stream, channel) = `cap` float2,
 *            two ping-pong sets.  K1 appends, K2* read, K3 moves the
 *            unconsumed tail to the other set.  Frame 0 of a plane is stream
 *            time `dec_base`; VDL2_HIST frames of history are always kept.
 *   state    StreamState (decimator carry) + ChanState (sync detector state:
 *            next evaluation instant, FIR sub-phase, last 68 phases, last two
 *            fit errors) -- the explicit, persistent form of the reference's
 *            stack-resident channel_t (vdlm2.h:56-79).
 *   cands    per channel: sync-trigger candidates of the free-running detector
 *            under all 8 timing hypotheses (K2a) + what happens after each (K2b).
 *   bursts   staging pool (K2b) and output ring (K2c/K2d) of vdl2gpu_burst_t.
 *
 * Pipeline of one push
 *   K1  channelise     time-parallel over the whole GPU, the only full-rate kernel
 *   K2a sync scan      time-parallel: fit error of EVERY (sample, sub-phase) pair
 *   K2b burst clusters one workgroup per candidate: exact state machine from the
 *                      trigger until the detector is history-free again
 *   K2c resolve        one workgroup per VDL channel: walks the real chain of
 *                      bursts through the tables (sequential but O(bursts))
 *   K2d gather         copies the bursts on the real chain to the output ring
 *   K3  compact
 *
 * Why the tables are exact: between bursts the reference detector evaluates
 * every 2nd 84 kS/s sample with a sticky FIR sub-phase r = clk%4 and sample
 * parity, both of which only change at a sync trigger (d8psk.c:248-250,
 * 305, 317-319).  68 evaluations after a burst the phase ring Ph[] holds only
 * new values and perr/p2err are the two previous errors, so the detector's
 * decision at sample n is a pure function of (n, r): the "free-running" fit
 * error E_r(n).  K2a computes E_r(n) for all n and r with the same float
 * sequence the serial code uses; K2b/K2c replay the short history-dependent
 * stretches (stale ring after a burst) with the serial machine.
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

#define VDL2_CS 8		/* channel planes per stream */
#define VDL2_HIST 160		/* frames of history kept: 17-tap FIR + 17 symbols x 8 + slack */
#define VDL2_NPH 68		/* NBPH*D8DWN, vdlm2.h:54-55 */
#define VDL2_STEADY 68		/* evaluations after which the detector forgot the last burst */
#define VDL2_MAXSYM 5456	/* >= ceil((25 + 8*8*255)/3) symbols of the longest burst */
#define VDL2_CARRY_FRAMES 49152	/* >= longest burst (43592 frames) + history + slack */
#define VDL2_PN_BITS (16384 + 64)
#define VDL2_CAND_CAP 4096	/* trigger candidates per channel per push */
#define VDL2_CL_MAXB 4		/* bursts per cluster before the resolver takes over */
#define VDL2_SEL_CAP 16384	/* bursts on the real chain per channel per push */

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

struct StreamState {
	long long dec_base;	/* stream time (84 kS/s index) of frame 0 of the current planes */
	long long dec_fill;	/* frames present before this push's K1 output */
	long long last_fill;	/* diagnostics: where the last push's output starts */
	long long last_J;
	float2 acc[2][VDL2_CS];	/* integrate-and-dump partial sums carried across pushes */
};

struct ChanState {
	long long pos;		/* stream time of the next WSYNC evaluation */
	int r;			/* FIR sub-phase (channel_t.clk after the -=8), 0..3 */
	int fresh;		/* evaluations since the last trigger/reset, saturating */
	float perr, p2err, pfr;	/* channel_t.perr/p2err/pfr */
	float ring[VDL2_NPH];	/* channel_t.Ph in time order, ring[67] newest */
	unsigned long long n_eval, n_trig, n_reject, n_burst, n_defer, n_slow, n_cand, n_redo;
};

struct ChanCfg {
	int chn, Fr, Fo, pad;
};

struct Cand {			/* free-running detector fires at dec_base + nrel with sub-phase r */
	int nrel, r;
	float p2err, perr, err, pfr;
};

enum { CL_STEADY = 0, CL_DEFER_FIRST = 1, CL_NONSTEADY = 2, CL_INVALID = 3 };
struct Cluster {		/* what the resolver reads of a cluster is its 8-byte head (cl_pack); this is the rest */
	ChanState saved;	/* CL_NONSTEADY: explicit state to continue from */
};

/* resolver's view of a cluster: x = n_s - dec_base, y = status | r_s << 2 | nslots << 4 | ntrig << 8 | nrej << 16 | nburst << 24 */
__device__ __forceinline__ int2 cl_pack(int n_s_rel, int status, int r_s, int nslots, int ntrig, int nrej, int nburst)
{
	ntrig = ntrig > 255 ? 255 : ntrig;
	nrej = nrej > 255 ? 255 : nrej;
	nburst = nburst > 255 ? 255 : nburst;
	return make_int2(n_s_rel, status | (r_s << 2) | (nslots << 4) | (ntrig << 8) | (nrej << 16) | (nburst << 24));
}

struct BurstDesc {		/* a burst found by a cluster; payload decoded later if it is on the real chain */
	long long nstar;	/* stream time of the sync trigger */
	int sc;			/* stream*8 + channel */
	int clk0;		/* (int)roundf(of), d8psk.c:305 */
	float df;
	int nbrow, nlbyte, pad;
};

struct Seg {			/* the chain idled in class (r, parity of lo) over stream-relative [lo, hi) */
	int lo, hi, r, pad;
};

struct K1Params {
	const void *raw;
	size_t stream_stride;
	int fmt, nbch;
	int sdrclk, L, maxwin;
	int c0, no0, nf0, parity;
	long long N, J;
	long long jbeg, jend;	/* generic kernel: outputs [jbeg, jend] (jend may be J = the carried tail) */
	long long per_lo, per_n;	/* fast kernel: whole 84-output periods [per_lo, per_lo+per_n) */
	int per_pb;		/* periods per wavefront (chosen so that the waves fill the GPU evenly) */
	const float2 *lo;	/* [S][8][L] */
	float2 *dec;		/* this push's planes, [S][8][cap] */
	long long cap;
	StreamState *ss;
};

struct K2Params {
	const float2 *dec;
	long long cap;
	int nbch, nstreams;
	long long J;
	StreamState *ss;
	ChanState *cs;
	const ChanCfg *cfg;
	const uint8_t *pn;
	Cand *cands;		/* [S*8][CAND_CAP] */
	Cluster *clusters;	/* [S*8][CAND_CAP] */
	int2 *clhead;		/* [S*8][CAND_CAP] what the resolver needs of every cluster, 8 bytes: see cl_pack() */
	unsigned *ctl;		/* [0]=out count [1]=out overflow [2]=stage count [3]=k2b ticket [4]=stage overflow
				 * [8 + S*8 ...] cand counts, then cand overflow flags */
	BurstDesc *stage;	/* burst descriptors of all clusters */
	unsigned *sel_list;	/* descriptors on the real chain (K2c -> K2d) */
	unsigned stage_cap;
	vdl2gpu_burst_t *recs;	/* output ring of this push */
	unsigned *outc;		/* [0] = records written, [1] = records dropped (ring full) */
	unsigned *outc_total_redo;	/* running count of serial redos (host adapts the number of repair rounds) */
	unsigned rec_cap;
	int force_serial;	/* diagnostics: skip the tables, run the serial machine */
	int full_scan;		/* scan all four sub-phases everywhere (no regions / verify) */
	int test_noregion;	/* test hook: skip the region scan so that K2a-verify must catch the misses */
	int2 *regs;		/* [S*8][REG_CAP] (lo, count) stream-relative */
	Seg *segs;		/* [S*8][SEG_CAP] */
	int *fail;		/* [S*8] earliest unexpected hit (stream-relative), >= VDL2_VERIFIED = verified */
	int *redo;		/* [S*8] 1 = this channel is being re-resolved in the repair round */
	int round;		/* 0 = first pass over every channel; 1 = repair pass over the channels whose
				 * verify failed (the hits were appended to their candidate tables) */
	ChanState *cs_out;	/* resolver result, committed by K2f */
	int *skey;		/* [S*8][CAND_CAP] candidates sorted by time: nrel*4 + r */
	unsigned short *sidx;	/* [S*8][CAND_CAP] sorted rank -> candidate index */
	unsigned short *prim;	/* [S*8][CAND_CAP] candidates whose cluster K2b computes */
	int *seeds;		/* [S*8][CAND_CAP] probe instants around which all classes are scanned */
	unsigned long long *dbg;	/* diagnostics: cycle counters */
};
#define CTL_OUT 0
#define CTL_OUT_OVF 1
#define CTL_STAGE 2
#define CTL_TICKET 3
#define CTL_STAGE_OVF 4
#define CTL_CAND0 8		/* [S*8] candidate counts, [S*8] overflow flags, then: */
#define CTL_NREG0 (CTL_CAND0 + 2 * p.nstreams * VDL2_CS)
#define CTL_NSEG0 (CTL_CAND0 + 3 * p.nstreams * VDL2_CS)
#define CTL_NSEL0 (CTL_CAND0 + 4 * p.nstreams * VDL2_CS)
#define CTL_NPRIM0 (CTL_CAND0 + 5 * p.nstreams * VDL2_CS)
#define CTL_NSEED0 (CTL_CAND0 + 6 * p.nstreams * VDL2_CS)

struct K3Params {
	const float2 *src;
	float2 *dst;
	long long cap;
	int nbch;
	long long J;
	StreamState *ss;
	const ChanState *cs;
	const unsigned *outc;	/* device counters: [2*ring] records, [2*ring+1] dropped, [4] serial redos so far */
	unsigned *host_cnt;	/* the same, in pinned host memory, for this push's ring ([4], [5]: frame counters, written by k4_publish) */
	int ring;
};

struct KInitParams {		/* per-push reset of the demodulator's control words */
	unsigned *ctl;
	int ctl_words;
	unsigned *outc;		/* 2 words of this push's ring */
	int *fail, *redo;
	int nsc;
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

/* ============================================================ shared DSP pieces */
__device__ __forceinline__ float d_tab(const uint32_t *t, int i)
{
	return __uint_as_float(t[i]);
}

/* filteredphase(), d8psk.c:219-230: x points at sample n-16.  All 17 samples of the
 * ring are fetched up front (independent loads, one memory latency); the taps
 * mflt[tap0], mflt[tap0+4], .. < 65 are then applied oldest sample first, exactly the
 * reference's accumulation order. */
template <int R> __device__ __forceinline__ float k2_fir_phase_r(const float2 *x)
{
	float2 v[17];
#pragma unroll
	for (int j = 0; j < 17; ++j)
		v[j] = x[j];
	float sr = 0.0f, si = 0.0f;
#pragma unroll
	for (int j = 0; j < 17; ++j) {
		if (R + 4 * j < 65) {
			const float m = d_tab(c_mflt, R + 4 * j);
			sr += v[j].x * m;
			si += v[j].y * m;
		}
	}
	return vdl2_atan2f(si, sr);
}

__device__ __forceinline__ float k2_fir_phase(const float2 *x, int tap0)
{
	switch (tap0) {
	case 0: return k2_fir_phase_r<0>(x);
	case 1: return k2_fir_phase_r<1>(x);
	case 2: return k2_fir_phase_r<2>(x);
	case 3: return k2_fir_phase_r<3>(x);
	default: break;
	}
	/* trigger instant: clk = (int)roundf(of) in [4,12] -> 16..14 taps (d8psk.c:305-306) */
	float2 v[17];
#pragma unroll
	for (int j = 0; j < 17; ++j)
		v[j] = x[j];
	float sr = 0.0f, si = 0.0f;
#pragma unroll
	for (int j = 0; j < 17; ++j) {
		const int i = tap0 + 4 * j;
		if (i < 65) {
			const float m = d_tab(c_mflt, i);
			sr += v[j].x * m;
			si += v[j].y * m;
		}
	}
	return vdl2_atan2f(si, sr);
}

/* d8psk.c:257-289: ph[0], ph[STRIDE], ... ph[16*STRIDE] are the 17 phases one symbol apart */
/* The reference compares the float phase step with the DOUBLE constants +-M_PI.  M_PI lies strictly
 * between the adjacent floats 0x40490fda (3.14159250) and 0x40490fdb (3.14159274), so for a float x
 *     (double)x > M_PI   <=>  x > 0x40490fda      and      (double)x < -M_PI  <=>  x < -0x40490fda
 * and the comparison can be made in float without changing a single decision. */
#define VDL2_PI_BELOW 0x40490fdau

template <int STRIDE> __device__ __forceinline__ float k2_sync_metric(const float *ph, float *slope)
{
	const float pi_lo = __uint_as_float(VDL2_PI_BELOW);
	float pr[17];
	double pud = 0.0;	/* Pu: every update goes float -> double -> float like `Pu -= 2 * M_PI`; */
	float pu = 0.0f;	/* kept in both forms so that only the narrowing is paid per step */
	float pv = ph[0] - d_tab(c_sw, 0);
	float mean = pv;
	pr[0] = pv;
#pragma unroll
	for (int l = 1; l < 17; ++l) {
		const float pc = ph[STRIDE * l] - d_tab(c_sw, l);
		const float pd = pc - pv;
		pv = pc;
		/* -1 / 0 / +1 turns; k * 2pi is exact in double, so pu + k*2pi is the reference's sum */
		const float k = (pd > pi_lo) ? -1.0f : ((pd < -pi_lo) ? 1.0f : 0.0f);
		pu = (float)(pud + (double)k * (2 * M_PI));
		pud = (double)pu;
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

/* Screening form of the fit error for the scan kernels.  It takes exactly the same unwrap
 * decisions as k2_sync_metric (pc and pd are the same float operations) but counts turns and
 * applies them as turns * 2pi in one fused step instead of rounding Pu through double after
 * every turn, and it may fuse/reassociate the regression.  With the decisions equal, the two
 * differ only by rounding: |Pr - Pr'| < 8e-5 per point (16 roundings of Pu at |Pu| < 128 plus
 * one ulp), |M - M'|, 8|fr - fr'| < 1e-3, so for an exact error below 4 (every residual < 2)
 * |err - err'| < 2 * sqrt(17 * 4) * 1.2e-3 < 0.02.  The scan therefore treats
 * err' >= VDL2_SCREEN_ERR (4.25) as proof that the exact error is >= 4 and recomputes every
 * instant below it, and its two neighbours, with k2_sync_metric. */
#define VDL2_SCREEN_ERR 4.25f
template <int STRIDE> __device__ __forceinline__ float k2_sync_metric_screen(const float *ph)
{
	const float pi_lo = __uint_as_float(VDL2_PI_BELOW);
	const float two_pi = 6.28318530717958647692f;
	float pr[17];
	float pv = ph[0] - d_tab(c_sw, 0);
	float turns = 0.0f, sum = pv, sl = pv * -8.0f;
	pr[0] = pv;
#pragma unroll
	for (int l = 1; l < 17; ++l) {
		const float pc = ph[STRIDE * l] - d_tab(c_sw, l);
		const float pd = pc - pv;
		pv = pc;
		const float k = (fabsf(pd) > pi_lo) ? copysignf(1.0f, pd) : 0.0f;
		turns -= k;
		pr[l] = __fmaf_rn(turns, two_pi, pc);
		sum += pr[l];
		sl = __fmaf_rn(pr[l], (float)(l - 8), sl);
	}
	const float mean = sum * (1.0f / 17.0f);
	const float fr = sl * (1.0f / 408.0f);
	float err = 0.0f;
#pragma unroll
	for (int l = 0; l < 17; ++l) {
		const float e = __fmaf_rn((float)(8 - l), fr, pr[l] - mean);
		err = __fmaf_rn(e, e, err);
	}
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

/* ====================================================== the serial state machine
 * Exact replay of demodD8psk()/putbit() for one channel by one workgroup:
 *   search  for the next <=NT evaluation instants compute the filtered phase
 *           (d8psk.c:219-230), the 17-point sync-word fit (d8psk.c:257-289) and
 *           find the first instant where `perr < 4 && err > perr` (d8psk.c:292);
 *   burst   one-shot timing estimate (d8psk.c:303-306), header symbols -> soft
 *           bits -> (25,20) Viterbi in one wavefront (viterbi.c), then every
 *           payload symbol in parallel: differential slice + Grey soft tables +
 *           descramble (d8psk.c:54-65, 211-217, 321-331) and the column-major
 *           de-interleave (d8psk.c:117-206) as a closed-form scatter.
 * Used three ways: K2b (from a trigger candidate until history-free), K2c (from
 * a carried non-steady state), and as the whole demodulator when the candidate
 * tables overflow or force_serial is set.
 */
#define K2_NT 256		/* workgroup size of the serial machine in the resolver */
#define K2B_NT 64		/* one wavefront per burst cluster */
#define VDL2_XT 256		/* samples in the LDS tile: >= 152+1 (ring), 16+2*64+1 (window), 16+7+65 (header) */

/* receiver's byte schedule for a burst of nbrow rows / nlbyte bytes in the last row
 * (d8psk.c:117-206): ND data bytes then NF FEC bytes, column-major over the rows,
 * short last row */
struct BurstGeom {
	int nd_rows, nd_last, nf_rows, nf_last, ND, NF, nsym;
};

__device__ __forceinline__ BurstGeom burst_geom(int nbrow, int nlbyte)
{
	BurstGeom g;
	g.nd_rows = nbrow;
	g.nd_last = nlbyte ? nlbyte : 249;	/* nlbyte==0: the zero-fill loop is skipped */
	g.ND = (nbrow - 1) * 249 + g.nd_last;
	if (nlbyte <= 2) {			/* FEC shortening of the last row, d8psk.c:153-161 */
		g.nf_rows = nbrow - 1;
		g.nf_last = 6;
	} else {
		g.nf_rows = nbrow;
		g.nf_last = (nlbyte <= 30) ? 2 : (nlbyte <= 67 ? 4 : 6);
	}
	g.NF = (g.nf_rows > 0) ? (g.nf_rows - 1) * 6 + g.nf_last : 0;
	g.nsym = (25 + 8 * (g.ND + g.NF) + 2) / 3;
	return g;
}

__device__ __forceinline__ void burst_timing(int clk0, int *j0, int *rb)
{
	int j = (32 - clk0 + 3) / 4;	/* samples until clk0 + 4j >= 32 (d8psk.c:239, 317-319) */
	if (j < 1)
		j = 1;
	*j0 = j;
	*rb = clk0 + 4 * j - 32;	/* sub-phase during and after the burst */
}

/* Payload of one accepted burst -> output record (all NT threads of the workgroup).
 * One lane per transmitted byte: its 8 bits sit in 3 or 4 consecutive symbols; the lane takes
 * their phases itself (and the one before, for the differential slice): differential slice +
 * Grey soft tables + descramble + hard decision (d8psk.c:54-65, 119, 168, 211-217, 321-331),
 * then the column-major de-interleave as a closed-form scatter (d8psk.c:127-147, 176-197). */
/* sph: optional LDS buffer of VDL2_MAXSYM floats.  With it every symbol phase is computed once by
 * one lane and the byte lanes read them from LDS; without it (serial stretches of the resolver, which
 * have no LDS to spare) each byte lane computes the four or five phases it needs itself. */
#define VDL2_MAXSYM 5456	/* symbols 7 .. (25 + 8 * 2040 - 1) / 3 */
template <int NT> __device__ void burst_payload(vdl2gpu_burst_t *rec, const float2 *x0, const uint8_t *pn, long long nstar,
						  int clk0, float df, int nbrow, int nlbyte, int stream, ChanCfg cfg, float *sph = nullptr)
{
	const int tid = threadIdx.x;
	int j0, rb;
	burst_timing(clk0, &j0, &rb);
	const BurstGeom g = burst_geom(nbrow, nlbyte);
	const long long nsym0 = nstar + j0;
	uint32_t *w = reinterpret_cast<uint32_t *>(&rec->data[0][0]);
	for (int i = tid; i < VDL2GPU_MAXROWS * VDL2GPU_ROWLEN / 4; i += NT)
		w[i] = 0u;
	__syncthreads();
	const float2 *xs0 = x0 + (nsym0 - 16);
	if (sph) {
		const int kmax = (25 + 8 * (g.ND + g.NF) - 1) / 3;
		for (int k = 7 + tid; k <= kmax; k += NT)
			sph[k - 7] = k2_fir_phase(xs0 + 8LL * k, rb);
		__syncthreads();
	}
	for (int b = tid; b < g.ND + g.NF; b += NT) {
		const int q0 = 25 + 8 * b;
		const int k0 = q0 / 3;	/* >= 8: never needs P1 */
		int q = q0;
		unsigned byte = 0;
		float pprev = sph ? sph[k0 - 8] : k2_fir_phase(xs0 + 8LL * (k0 - 1), rb);
		for (int k = k0; q < q0 + 8; ++k) {
			const float pk = sph ? sph[k - 7] : k2_fir_phase(xs0 + 8LL * k, rb);
			const int idx = k2_grey_index(pk, pprev, df);
			pprev = pk;
			for (int i = q - 3 * k; i < 3 && q < q0 + 8; ++i, ++q) {
				const float v = k2_soft_bit(idx, i, pn[q]);
				if ((double)v > 0.5)
					byte |= 1u << (q - q0);
			}
		}
		int row, col;
		if (b < g.ND) {
			const int full = g.nd_last * g.nd_rows;
			if (b < full) {
				col = b / g.nd_rows;
				row = b % g.nd_rows;
			} else {
				const int bb = b - full;
				col = g.nd_last + bb / (g.nd_rows - 1);
				row = bb % (g.nd_rows - 1);
			}
		} else {
			const int bf = b - g.ND;
			const int full = g.nf_last * g.nf_rows;
			if (bf < full) {
				col = bf / g.nf_rows;
				row = bf % g.nf_rows;
			} else {
				const int bb = bf - full;
				col = g.nf_last + bb / (g.nf_rows - 1);
				row = bb % (g.nf_rows - 1);
			}
			col += 249;
		}
		rec->data[row][col] = (uint8_t)byte;
	}
	if (tid == 0) {
		rec->stream = stream;
		rec->chn = cfg.chn;
		rec->Fr = cfg.Fr;
		rec->nbrow = nbrow;
		rec->nlbyte = nlbyte;
		rec->df = df;
		rec->ppm = 0.0f;	/* host: d8psk.c:302 needs libm double math */
		rec->trig_dec = nstar;
		rec->end_dec = nsym0 + 8LL * (g.nsym - 1);
		rec->trig_sample = 0;
		rec->end_sample = 0;
	}
}

template <int NT> struct MachSharedT {
	float pbuf[VDL2_NPH + NT];	/* phases: [0,68) = history ring */
	float errs[NT + 2];		/* errs[t+2] = err of eval t; [0],[1] = p2err, perr */
	float frs[NT + 1];		/* frs[t+1] = slope of eval t; [0] = pfr */
	float psym[12];			/* header symbol phases */
	float2 xt[VDL2_XT];		/* LDS tile of the channel's samples (cluster mode) */
	float smf[72];			/* low-pass taps mflt[] (d8psk.h:28-45), zero padded */
	float hsoft[25];		/* descrambled header soft bits */
	uint8_t vbk[26][32], vbv[26][32];	/* Viterbi back pointers / decided bits */
	int first;
	int ctl[16];
	float fctl[8];
};

struct MachCtx {
	const float2 *x;	/* channel plane, frame 0 = stream time dec_base */
	long long dec_base, avail_end;
	const uint8_t *pn;
	vdl2gpu_burst_t *recs;	/* sink: output ring, payload decoded at once (K2c serial stretches) ... */
	BurstDesc *desc;	/* ... or descriptor pool, payload decoded by K2d if selected (K2b, K2c) */
	unsigned *sel, *nsel;	/* K2c: descriptors made by its serial stretches are on the real chain */
	unsigned dyn_base;	/* first dynamic descriptor slot */
	long long desc_static;	/* >= 0: descriptor slots are desc_static + burst index (K2b: no atomics) */
	int sc;
	unsigned long long *dbg;
	long long t_lo, t_hi;	/* stream-time range currently held in the LDS tile (cluster mode) */
	const float *grey;	/* 3 x 257 soft-bit tables in LDS, or nullptr -> constant memory */
	unsigned *rec_count, *rec_ovf;
	unsigned rec_cap;
	int stream;
	ChanCfg cfg;
};

struct MachState {
	long long pos;
	int r, fresh;
};

struct MachOut {
	int nslots, slots[VDL2_CL_MAXB];
	int ntrig, nrej, nburst, ndefer;
	long long neval;
};

enum { MR_END = 0, MR_DEFER = 1, MR_STEADY = 2, MR_LIMIT = 3 };

#define VDL2_PN_HEAD 0xa423d8c8u	/* first 32 scrambler bits from seed 0x4D4B (d8psk.c:54-65, 299) */

__device__ __forceinline__ float mach_soft_bit(const MachCtx &cx, int idx, int which, int pnbit)
{
	const float v = cx.grey ? cx.grey[which * 257 + idx]
				: d_tab(which == 0 ? c_grey1 : (which == 1 ? c_grey2 : c_grey3), idx);
	return pnbit ? (float)(1.0 - (double)v) : v;	/* descrambler, d8psk.c:60-63 */
}


/* XL = true: all sample reads of the machine go through the LDS tile sh.xt, which
 * mach_need() (re)fills from the channel plane whenever the next phase of work leaves it;
 * XL = false: samples are read from the plane in HBM/L2 directly. */
template <int NT, bool XL> __device__ __forceinline__ void mach_need(MachSharedT<NT> &sh, MachCtx &cx, long long lo, long long hi)
{
	if (!XL)
		return;
	if (lo >= cx.t_lo && hi <= cx.t_hi)
		return;		/* uniform: cx is the same in every lane */
	__syncthreads();
	long long cnt = cx.avail_end - lo;
	cnt = cnt > VDL2_XT ? VDL2_XT : cnt;
	const float2 *src = cx.x + (lo - cx.dec_base);
	for (int i = threadIdx.x; i < (int)cnt; i += NT)
		sh.xt[i] = src[i];
	cx.t_lo = lo;
	cx.t_hi = lo + (cnt > 0 ? cnt : 0);
	__syncthreads();
}

template <int NT> __device__ __forceinline__ void mach_init_taps(MachSharedT<NT> &sh)
{
	for (int i = threadIdx.x; i < 72; i += NT)
		sh.smf[i] = (i < 65) ? d_tab(c_mflt, i) : 0.0f;
	__syncthreads();
}

/* filteredphase() at the sample of stream time n with first tap `tap0` (d8psk.c:219-230) */
template <int NT, bool XL> __device__ __forceinline__ float mach_fir(const MachSharedT<NT> &sh, const MachCtx &cx, long long n, int tap0)
{
	float2 v[17];
	if (XL) {
		const float2 *x = &sh.xt[(int)(n - 16 - cx.t_lo)];
#pragma unroll
		for (int j = 0; j < 17; ++j)
			v[j] = x[j];
	} else {
		const float2 *x = cx.x + (n - 16 - cx.dec_base);
#pragma unroll
		for (int j = 0; j < 17; ++j)
			v[j] = x[j];
	}
	float sr = 0.0f, si = 0.0f;
#pragma unroll
	for (int j = 0; j < 17; ++j) {
		const int i = tap0 + 4 * j;
		if (i < 65) {
			const float m = sh.smf[i];
			sr += v[j].x * m;
			si += v[j].y * m;
		}
	}
	return vdl2_atan2f(si, sr);
}

template <int NT> __device__ __forceinline__ void mach_load(MachSharedT<NT> &sh, const ChanState *cs)
{
	const int tid = threadIdx.x;
	for (int i = tid; i < VDL2_NPH; i += NT)
		sh.pbuf[i] = cs->ring[i];
	if (tid == 0) {
		sh.errs[0] = cs->p2err;
		sh.errs[1] = cs->perr;
		sh.frs[0] = cs->pfr;
	}
	__syncthreads();
}

template <int NT> __device__ __forceinline__ void mach_store(const MachSharedT<NT> &sh, const MachState &st, ChanState *cs)
{
	const int tid = threadIdx.x;
	for (int i = tid; i < VDL2_NPH; i += NT)
		cs->ring[i] = sh.pbuf[i];
	if (tid == 0) {
		cs->pos = st.pos;
		cs->r = st.r;
		cs->fresh = st.fresh;
		cs->p2err = sh.errs[0];
		cs->perr = sh.errs[1];
		cs->pfr = sh.frs[0];
	}
}

/* shift the phase ring: new ring = pbuf[from .. from+67] (all threads call) */
template <int NT> __device__ __forceinline__ void mach_shift_ring(MachSharedT<NT> &sh, int from, float e0, float e1, float f0)
{
	const int tid = threadIdx.x;
	float keep[(VDL2_NPH + NT - 1) / NT];
#pragma unroll
	for (int k = 0; k < (VDL2_NPH + NT - 1) / NT; ++k) {
		const int i = tid + k * NT;
		keep[k] = (i < VDL2_NPH) ? sh.pbuf[from + i] : 0.0f;
	}
	__syncthreads();
#pragma unroll
	for (int k = 0; k < (VDL2_NPH + NT - 1) / NT; ++k) {
		const int i = tid + k * NT;
		if (i < VDL2_NPH)
			sh.pbuf[i] = keep[k];
	}
	if (tid == 0) {
		sh.errs[0] = e0;
		sh.errs[1] = e1;
		sh.frs[0] = f0;
	}
	__syncthreads();
}

/* Build the detector state at a history-free instant (n, r): the ring holds the
 * free-running phases of the previous 68 evaluations and perr/p2err/pfr are
 * those of evaluations n-2 and n-4.  Needs samples back to n-152. */
template <int NT, bool XL> __device__ __forceinline__ void mach_materialize(MachSharedT<NT> &sh, MachCtx &cx, long long n, int r)
{
	const int tid = threadIdx.x;
	mach_need<NT, XL>(sh, cx, n - 152, n + 1);
	for (int i = tid; i < VDL2_NPH; i += NT) {
		const long long q = n - 2LL * (VDL2_NPH - i);
		sh.pbuf[i] = mach_fir<NT, XL>(sh, cx, q, r);
	}
	__syncthreads();
	if (tid < 2) {
		float fr;
		const float e = k2_sync_metric<4>(&sh.pbuf[3 - tid], &fr);
		sh.errs[1 - tid] = e;	/* tid 0: evaluation n-2 -> perr; tid 1: n-4 -> p2err */
		if (tid == 0)
			sh.frs[0] = fr;
	}
	__syncthreads();
}

/* stop_steady: return MR_STEADY as soon as the detector is history-free and at least
 * `min_trig` triggers were handled.  first_nev: size of the first search window (a hint). */
template <int NT, bool XL> __device__ int machine_run(MachSharedT<NT> &sh, MachCtx &cx, MachState &st, bool stop_steady,
					     int min_trig, int max_bursts, int first_nev, MachOut &out)
{
	const int tid = threadIdx.x;
	const float2 *x0 = cx.x - cx.dec_base;	/* x0[n] = sample at stream time n */
	long long pos = st.pos;
	int r = st.r, fresh = st.fresh;
	int rc = MR_END;
	for (;;) {
		if (stop_steady && fresh >= VDL2_STEADY && out.ntrig >= min_trig) {
			rc = MR_STEADY;
			break;
		}
		if (out.nslots >= max_bursts) {
			rc = MR_LIMIT;
			break;
		}
		const long long rem = (cx.avail_end - pos + 1) / 2;
		int nev = rem > NT ? NT : (int)rem;
		if (nev <= 0) {
			rc = MR_END;
			break;
		}
		if (first_nev > 0) {
			nev = nev < first_nev ? nev : first_nev;
			first_nev = 0;
		} else if (stop_steady && out.ntrig >= min_trig && fresh < VDL2_STEADY) {
			const int need = VDL2_STEADY - fresh;
			nev = nev < need ? nev : need;
		}
		/* ---- search window: evaluations at pos, pos+2, ... */
		mach_need<NT, XL>(sh, cx, pos - 16, pos + 2LL * nev);
		if (tid < nev)
			sh.pbuf[VDL2_NPH + tid] = mach_fir<NT, XL>(sh, cx, pos + 2 * tid, r);
		if (tid == 0)
			sh.first = 0x7fffffff;
		__syncthreads();
		if (tid < nev) {
			float fr;
			const float err = k2_sync_metric<4>(&sh.pbuf[tid + 4], &fr);
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
			mach_shift_ring(sh, nev, sh.errs[nev], sh.errs[nev + 1], sh.frs[nev]);
			pos += 2LL * nev;
			out.neval += nev;
			fresh = fresh + nev > 1000000 ? 1000000 : fresh + nev;
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
			int j0, rb0;
			burst_timing(clk0, &j0, &rb0);
			sh.ctl[0] = clk0;
			sh.ctl[1] = j0;
			sh.ctl[2] = rb0;
			sh.fctl[0] = sh.frs[ts];	/* df = pfr, d8psk.c:301 */
		}
		__syncthreads();
		const int clk0 = sh.ctl[0], j0 = sh.ctl[1], rb = sh.ctl[2];
		const float df = sh.fctl[0];
		const long long nsym0 = nstar + j0;	/* stream time of burst symbol 0 */
		bool defer = (nsym0 + 64 >= cx.avail_end);	/* 9 header symbols must be present */
		int accepted = 0, nbrow = 0, nlbyte = 0, nsym = 0;
		if (!defer) {
			mach_need<NT, XL>(sh, cx, nstar - 16, nsym0 + 65);
			if (tid < 9)
				sh.psym[tid] = mach_fir<NT, XL>(sh, cx, nsym0 + 8 * tid, rb);
			if (tid == 9)
				sh.fctl[1] = mach_fir<NT, XL>(sh, cx, nstar, clk0);	/* P1 */
			__syncthreads();
			if (tid < 25) {
				const int k = tid / 3;
				const float pprev = k ? sh.psym[k - 1] : sh.fctl[1];
				const int idx = k2_grey_index(sh.psym[k], pprev, df);
				float v = mach_soft_bit(cx, idx, tid % 3, (int)((VDL2_PN_HEAD >> tid) & 1u));
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
				int sv = 0;
				for (int n = 25; n > 0; --n) {
					if (sh.vbv[n][sv])
						word |= mask;
					sv = sh.vbk[n][sv];
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
				nsym = burst_geom(nbrow, nlbyte).nsym;
				if (nsym0 + 8LL * (nsym - 1) >= cx.avail_end)
					defer = true;
			}
		}
		if (defer) {
			/* the burst is not completely inside the data we hold: commit the
			 * evaluations before the trigger and retry on the next push */
			mach_shift_ring(sh, ts, sh.errs[ts], sh.errs[ts + 1], sh.frs[ts]);
			pos += 2LL * ts;
			out.neval += ts;
			fresh = fresh + ts > 1000000 ? 1000000 : fresh + ts;
			out.ndefer++;
			rc = MR_DEFER;
			break;
		}
		out.ntrig++;
		long long nlast;
		if (!accepted) {
			out.nrej++;
			nlast = nsym0 + 64;	/* state returns to WSYNC on the 25th bit (symbol 8) */
		} else {
			nlast = nsym0 + 8LL * (nsym - 1);
			if (tid == 0) {
				unsigned slot;
				if (cx.desc_static >= 0)
					slot = (unsigned)(cx.desc_static + out.nslots);	/* out.nslots < VDL2_CL_MAXB here */
				else {
					/* dynamic slots live behind the static region of the pool */
					slot = atomicAdd(cx.rec_count, 1u);
					if (cx.desc)
						slot += cx.dyn_base;
				}
				if (slot >= cx.rec_cap) {
					atomicAdd(cx.rec_ovf, 1u);
					slot = 0xffffffffu;
				} else if (cx.desc) {
					BurstDesc d;
					d.nstar = nstar;
					d.sc = cx.sc;
					d.clk0 = clk0;
					d.df = df;
					d.nbrow = nbrow;
					d.nlbyte = nlbyte;
					d.pad = 0;
					cx.desc[slot] = d;
					if (cx.sel) {
						const unsigned q = atomicAdd(cx.nsel, 1u);
						if (q < VDL2_SEL_CAP)
							cx.sel[q] = slot;
						else
							atomicAdd(cx.rec_ovf, 1u);
					}
				}
				sh.ctl[6] = (int)slot;
			}
			__syncthreads();
			const unsigned slot = (unsigned)sh.ctl[6];
			if (!XL && !cx.desc && slot != 0xffffffffu)
				burst_payload<NT>(cx.recs + slot, x0, cx.pn, nstar, clk0, df, nbrow, nlbyte, cx.stream, cx.cfg);
#pragma unroll
			for (int i = 0; i < VDL2_CL_MAXB; ++i)
				if (out.nslots == i)
					out.slots[i] = (int)slot;
			out.nslots++;
			out.nburst++;
		}
		/* back to the idle detector: ring keeps the phases up to the trigger
		 * evaluation (Ph is not written during a burst), errors re-armed
		 * (d8psk.c:308), sub-phase sticks at rb */
		mach_shift_ring(sh, ts + 1, 500.0f, 500.0f, sh.frs[ts]);
		out.neval += ts + 1;
		pos = nlast + 2;
		r = rb;
		fresh = 0;
	}
	st.pos = pos;
	st.r = r;
	st.fresh = fresh;
	return rc;
}

__device__ __forceinline__ void mach_ctx(MachCtx &cx, const K2Params &p, int s, int c, bool to_stage)
{
	const StreamState *ss = p.ss + s;
	cx.x = p.dec + ((size_t)s * VDL2_CS + c) * p.cap;
	cx.dec_base = ss->dec_base;
	cx.avail_end = ss->dec_base + ss->dec_fill + p.J;
	cx.pn = p.pn;
	cx.sc = s * VDL2_CS + c;
	cx.dbg = to_stage ? p.dbg : nullptr;
	cx.t_lo = cx.t_hi = 0;
	cx.grey = nullptr;
	cx.sel = cx.nsel = nullptr;
	cx.desc_static = -1;
	if (to_stage) {
		cx.recs = nullptr;
		cx.dyn_base = (unsigned)p.nstreams * VDL2_CS * VDL2_CAND_CAP * VDL2_CL_MAXB;
		cx.desc = p.stage;
		cx.rec_count = p.ctl + CTL_STAGE;
		cx.rec_ovf = p.ctl + CTL_STAGE_OVF;
		cx.rec_cap = p.stage_cap;
	} else {
		cx.recs = p.recs;
		cx.dyn_base = 0;
		cx.desc = nullptr;
		cx.rec_count = p.outc;
		cx.rec_ovf = p.outc + 1;
		cx.rec_cap = p.rec_cap;
	}
	cx.stream = s;
	cx.cfg = p.cfg[(size_t)s * VDL2_CS + c];
}

/* ====================================================================== K2a
 * Sync scan.  For a run of evaluation instants n = nbase + S*i of one channel and a set of FIR
 * sub-phases r, compute the filtered phase P_r(n) and the free-running fit error E_r(n) from
 * P_r(n), P_r(n-8), .. P_r(n-128), and test where the idle detector would fire:
 *     E_r(n-2) < 4 && E_r(n) > E_r(n-2)
 * (S = 1: every sample, both parities; S = 2: one parity only -- n-8l and n-2 keep n's parity.)
 *
 * Three uses, all the same tile routine on K2A_TS instants staged in LDS:
 *   k2a_probe   every sample >= pos of each channel, but ONLY the sub-phase the channel's
 *               detector is in at the start of the push.  Finds every burst (a burst fires the
 *               detector in all 8 (sub-phase, parity) classes within a few samples) and is
 *               already the complete table for that sub-phase.
 *   k2a_region  the other three sub-phases, only in the neighbourhood of the probe's hits.
 *   k2a_verify  after the resolver: every stretch the real chain idled through in a class the
 *               probe did not cover is scanned in exactly that class; a hit means the tables
 *               missed an event and the channel is redone serially (K2f).  This is what makes
 *               the shortcut exact instead of heuristic.
 * VDL2GPU_F_FULLSCAN makes the probe cover all four sub-phases (no regions/verify needed).
 */
#ifndef K2A_THREADS
#define K2A_THREADS 256
#endif
#ifndef K2A_TS
#define K2A_TS 1024		/* evaluation instants per tile */
#endif
#define K2A_POFF 132		/* samples of phase history before the tile: 128 + 4 */
#define K2A_XOFF (K2A_POFF + 16)
#define K2A_XMAX (2 * K2A_TS + K2A_XOFF)
#define K2A_WL2 24		/* survivors of both screens whose exact phases fit in LDS at once */
#define K2A_DEF 320		/* survivors collected before they are worked off; must hold one more tile pass (K2A_WL) */
#ifndef K2A_WL
#define K2A_WL 192		/* screened-in evaluations per tile and sub-phase; more than that and the tile is done in pieces */
#endif
#define VDL2_REG_CAP 1024	/* probe-hit regions per channel per push */
#define VDL2_REG_PAD 40		/* samples scanned on either side of a probe hit */
#define VDL2_REG_GAP 96		/* hits closer than this share a region */
#define VDL2_SEG_CAP 4096	/* verify segments per channel per push */
#define VDL2_VERIFIED 0x7f000000	/* fail[] values at or above this mean: nothing unexpected found */
#define VDL2_SEED_ERR 7.0f	/* probe fit error below which a neighbourhood is scanned in every class
				 * (the detector itself needs < 4): catches marginal events that only some
				 * classes detect; what it still misses is caught by K2a-verify */

struct K2aDef {			/* an evaluation that needs the exact fit */
	int n;			/* its instant, stream-relative (samples) */
	int r;			/* FIR sub-phase */
	int lo, hi;		/* verify: only hits in [lo, hi) count */
};

struct K2aShared {
	float2 xs[K2A_XMAX + 8];	/* S = 1: samples in order; S = 2: even samples, then (at K2A_XODD) odd samples, so that
					 * both FIR tap parities are unit-stride across lanes */
	float2 wu[K2A_TS + K2A_POFF];	/* unit phasor of every filtered sample (history first), then in place the phasor
					 * of the symbol-spaced phase step */
	float smf[72];			/* low-pass taps mflt[] (d8psk.h:28-45) */
	float atab[VDL2_ATAN_ROWS * VDL2_ATAN_STRIDE];	/* atanf range constants (vdl2_math.h) */
	int wl[K2A_WL];			/* evaluations of the current tile the first screen lets through */
	K2aDef dl[K2A_DEF];		/* survivors of both screens, collected over tiles until there are enough to
					 * give every lane an exact phase to compute (k2a_flush) */
	float sph[K2A_WL2][3][17];	/* exact phases of one batch of survivors: evaluation before / at / after */
	float we[3][K2A_WL2], wf[K2A_WL2];	/* their exact fit errors, and the slope at the middle one */
	int nwl, ndl;
};
#define K2A_XODD (K2A_XMAX / 2 + 4)	/* 8-byte elements: an odd multiple of 64 bytes away, so the two halves use disjoint banks */

/* Screens for the 17-point fit (the expensive part of the scan).
 * With Pr[] the unwrapped, template-corrected phases the reference fits a line to (d8psk.c:257-289)
 * and e[] their residuals, the lag-1 phase steps satisfy D_l = Pr[l] - Pr[l-1] = fr + e_l - e_(l-1), and
 * D_l = (P_l - P_(l-1)) - (SW_l - SW_(l-1)) modulo 2pi whatever the unwrap decided.  Hence
 *      sum_l (D_l - mean D)^2 <= sum_l (e_l - e_(l-1))^2 <= 4 * err,
 * and with R = |sum_l exp(j D_l)| >= sum_l cos(D_l - mean D) >= 16 - sum_l (D_l - mean D)^2 / 2:
 *      err >= (16 - R) / 2.
 * The same argument on the 15 lag-2 steps Pr[l+2] - Pr[l] gives err >= (15 - R2) / 2, and on the
 * 14 lag-3 steps err >= (14 - R3) / 2.
 * Neither needs the unwrap or even a phase: exp(j D_l) = c_l * u_l with u = w * conj(w') the unit
 * phasor of two symbol-spaced FIR outputs and c_l the template step (an odd multiple of pi/8), and
 * the lag-2 and lag-3 phasors are products of neighbouring lag-1 ones.  An evaluation with R <= 7.5,
 * R2 <= 6.5 or R3 <= 5.5 has err >= 4.25 > 4 (the rounding in R, R2 is < 1e-4), so it can neither be the minimum
 * the detector fires after nor matter to it.  Every evaluation gets the first screen (32 packed
 * FMAs, passes ~2 % of noise), its survivors the second (passes ~6 % of those), and only what
 * survives both -- sync words, and about one noise evaluation in a thousand -- gets atan2f, the
 * exact unwrap and the exact fit, together with its two neighbours.  Non-finite values count as
 * surviving. */
#define VDL2_SCREEN_R2 56.25f	/* R^2: (16 - 7.5) / 2 = 4.25 */
#define VDL2_SCREEN_R22 42.25f	/* R2^2: (15 - 6.5) / 2 = 4.25 */
#define VDL2_SCREEN_R32 30.25f	/* R3^2: (14 - 5.5) / 2 = 4.25 */
__device__ __forceinline__ v2f k2_rot(v2f acc, v2f u, float cx, float cy)
{
	/* acc += (cx + j cy) * u */
	acc = __builtin_elementwise_fma((v2f){cx, cx}, u, acc);
	return __builtin_elementwise_fma((v2f){-cy, cy}, u.yx, acc);
}


/* detector test of one instant (d8psk.c:292) and what a hit means in each scan mode */
__device__ __forceinline__ void k2a_emit(const K2Params &p, int sc, long long dec_base, long long n, int r, int mode,
						  long long chk_lo, long long chk_hi, int *fail, int skip_r, int skip_par,
						  float p2err, float perr, float err, float pfr, unsigned *cntp, unsigned *ovf, Cand *cl)
{
	if (mode == 2 && perr < VDL2_SEED_ERR && err > perr) {
		const unsigned kk = atomicAdd(p.ctl + CTL_NSEED0 + sc, 1u);
		if (kk < VDL2_CAND_CAP)	/* surplus seeds are simply dropped: K2a-verify covers what they would have */
			p.seeds[(size_t)sc * VDL2_CAND_CAP + kk] = (int)(n - dec_base);
	}
	if (!(perr < 4.0f && err > perr))
		return;
	if (mode == 0 || mode == 2) {
		if (r == skip_r && (int)(n & 1) == skip_par)
			return;	/* that class is the probe's: already in the table */
	} else {
		if (n < chk_lo || n >= chk_hi)
			return;
		/* a detector hit the tables did not list: remember where, and make it a seed so that the
		 * repair round scans its neighbourhood in every class (that finds this hit again, and
		 * whatever else the detector does around it) */
		atomicMin(fail, (int)(n - dec_base));
		const unsigned kk = atomicAdd(p.ctl + CTL_NSEED0 + sc, 1u);
		if (kk < VDL2_CAND_CAP)
			p.seeds[(size_t)sc * VDL2_CAND_CAP + kk] = (int)(n - dec_base);
		return;
	}
	const unsigned kk = atomicAdd(cntp, 1u);
	if (kk < VDL2_CAND_CAP) {
		Cand cd;
		cd.nrel = (int)(n - dec_base);
		cd.r = r;
		cd.p2err = p2err;
		cd.perr = perr;
		cd.err = err;
		cd.pfr = pfr;
		cl[kk] = cd;
	} else
		*ovf = 1u;
}

/* The samples of a tile travel HBM -> registers -> LDS.  The registers of the *next* tile of the
 * same workgroup are loaded right after the current tile's have been parked in LDS, so that the
 * memory latency (several thousand cycles under load) is hidden behind the current tile's arithmetic. */
template <int S> struct K2aPre {
	static constexpr int NL = (S * (K2A_TS - 1) + 1 + K2A_XOFF + K2A_THREADS - 1) / K2A_THREADS;
	float2 v[NL];
	bool loaded;
};

/* once per workgroup, before its first tile */
__device__ __forceinline__ void k2a_tables(K2aShared &sh)
{
	for (int i = threadIdx.x; i < 72; i += K2A_THREADS)
		sh.smf[i] = (i < 65) ? d_tab(c_mflt, i) : 0.0f;
	if (threadIdx.x < VDL2_ATAN_ROWS * VDL2_ATAN_STRIDE)
		sh.atab[threadIdx.x] = vdl2_atan_tab_entry(threadIdx.x);
	if (threadIdx.x == 0)
		sh.ndl = 0;
	__syncthreads();
}

template <int S> __device__ __forceinline__ void k2a_fetch(K2aPre<S> &pre, const K2Params &p, int sc, long long dec_base, long long nbase, int cnt)
{
	const float2 *x = p.dec + (size_t)sc * p.cap + (nbase - K2A_XOFF - dec_base);
	const int nx = S * (cnt - 1) + 1 + K2A_XOFF;
#pragma unroll
	for (int k = 0; k < K2aPre<S>::NL; ++k) {
		const int i = (int)threadIdx.x + k * K2A_THREADS;
		if (i < nx)
			pre.v[k] = x[i];
	}
	pre.loaded = true;
}

/* mode 0: append candidates; mode 1: report hits in [chk_lo, chk_hi) to *fail and append them;
 * mode 2: probe (candidates + seeds).  One sub-phase per pass:
 *   FIR + unit phasor of every instant | phase-step phasors | first screen -> worklist |
 *   second screen of the worklist | exact phases of the survivors | exact fits | detector test. */
/* Work off the collected survivors: exact phases (FIR from the channel plane in HBM/L2 -- the tile
 * they came from has left LDS -- then atan2f, d8psk.c:219-229), exact fits (d8psk.c:257-289) of the
 * evaluation and its two neighbours, detector test (d8psk.c:292).  Every lane has work: 51 phases
 * per survivor. */
__device__ void k2a_flush(K2aShared &sh, const K2Params &p, int sc, long long dec_base, int mode, int *fail, int skip_r, int skip_par)
{
	const int tid = threadIdx.x;
	__syncthreads();
	const int nd = sh.ndl;
	const float2 *x0 = p.dec + (size_t)sc * p.cap;	/* x0[n] = sample at stream-relative time n */
	unsigned *cntp = p.ctl + CTL_CAND0 + sc;
	unsigned *ovf = p.ctl + CTL_CAND0 + p.nstreams * VDL2_CS + sc;
	Cand *cl = p.cands + (size_t)sc * VDL2_CAND_CAP;
	for (int b0 = 0; b0 < nd; b0 += K2A_WL2) {
		const int nb = (nd - b0 < K2A_WL2) ? nd - b0 : K2A_WL2;
		for (int t = tid; t < 51 * nb; t += K2A_THREADS) {
			const int slot = t / 51, rem = t - 51 * slot, w = rem / 17, l = rem - 17 * w;
			const K2aDef d = sh.dl[b0 + slot];
			const float2 *x = x0 + (d.n + (w - 1) * 2 - 8 * (16 - l) - 16);
			float2 xv[17];
#pragma unroll
			for (int j = 0; j < 17; ++j)
				xv[j] = x[j];
			v2f acc = {0.0f, 0.0f};
#pragma unroll
			for (int j = 0; j < 16; ++j) {
				const float m = sh.smf[d.r + 4 * j];
				acc += (v2f){xv[j].x, xv[j].y} * (v2f){m, m};
			}
			if (d.r == 0) {	/* mflt[r + 64] exists only for r == 0 */
				const float m = sh.smf[64];
				acc += (v2f){xv[16].x, xv[16].y} * (v2f){m, m};
			}
			sh.sph[slot][w][l] = vdl2_atan2f_tab(acc.y, acc.x, sh.atab);
		}
		__syncthreads();
		for (int k = tid; k < 3 * nb; k += K2A_THREADS) {
			const int slot = k / 3, w = k - 3 * slot;
			float fr;
			sh.we[w][slot] = k2_sync_metric<1>(&sh.sph[slot][w][0], &fr);
			if (w == 1)
				sh.wf[slot] = fr;
		}
		__syncthreads();
		for (int k = tid; k < nb; k += K2A_THREADS) {
			const K2aDef d = sh.dl[b0 + k];
			k2a_emit(p, sc, dec_base, dec_base + d.n + 2, d.r, mode, dec_base + d.lo, dec_base + d.hi, fail, skip_r, skip_par,
				 sh.we[0][k], sh.we[1][k], sh.we[2][k], sh.wf[k], cntp, ovf, cl);
		}
		__syncthreads();
	}
	if (tid == 0)
		sh.ndl = 0;
	__syncthreads();
}

/* filtered sample of tile instant q (sub-phase taps mf[], 17th tap only for r == 0): d8psk.c:219-228 */
template <int S> __device__ __forceinline__ v2f k2a_fir(const K2aShared &sh, int q, const float (&mf)[17], bool tap17)
{
	/* tap j multiplies sample (nbase + S*(q-PH)) - 16 + j = tile sample S*q + j */
	const v2f *xe = reinterpret_cast<const v2f *>(&sh.xs[q]);
	const v2f *xo = reinterpret_cast<const v2f *>(&sh.xs[K2A_XODD + q]);
	v2f xv[17];
#pragma unroll
	for (int j = 0; j < 17; ++j)	/* every LDS read in flight before the first multiply */
		xv[j] = (S == 2) ? ((j & 1) ? xo[j >> 1] : xe[j >> 1]) : xe[j];
	v2f acc = {0.0f, 0.0f};
#pragma unroll
	for (int j = 0; j < 16; ++j)
		acc += xv[j] * (v2f){mf[j], mf[j]};
	if (tap17)
		acc += xv[16] * (v2f){mf[16], mf[16]};
	return acc;
}

template <int S> __device__ void k2a_tile(K2aShared &sh, const K2Params &p, int sc, long long dec_base, long long nbase,
					   int cnt, unsigned rmask, int mode, long long chk_lo, long long chk_hi, int *fail,
					   K2aPre<S> &pre, long long next_nbase, int next_cnt, int skip_r = -1, int skip_par = 0)
{
	const int tid = threadIdx.x;
	constexpr int PH = K2A_POFF / S;	/* phase instants of history */
	constexpr int LSTR = 8 / S;		/* one symbol in instants */
	constexpr int E2 = 2 / S, E4 = 4 / S;	/* previous two evaluations in instants */
	constexpr int NQ = (K2A_TS + PH + K2A_THREADS - 1) / K2A_THREADS;
	static_assert(K2A_POFF == S * PH, "phase history must be a whole number of instants");
	/* exp(-j (SW[l] - SW[l-1])), l = 1..16: the template steps are 1,7,5,-7,1,3,-3,-7,3,-1,5,-5,-3,-5,-1,7 (x pi/8) */
	constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f;
	constexpr float rc[16] = {C1, -C1, -S1, -C1, C1, S1, S1, -C1, S1, C1, -S1, -S1, S1, -S1, C1, -C1};
	constexpr float rs[16] = {-S1, -S1, -C1, S1, -S1, -C1, C1, S1, -C1, S1, -C1, C1, C1, C1, S1, -S1};
	const int nx = S * (cnt - 1) + 1 + K2A_XOFF;
	const bool prof = p.dbg && mode == 2 && tid == 0 && (blockIdx.x & 7) == 0;
	long long tq = prof ? clock64() : 0;
#define K2A_STAMP(slot) do { if (prof) { const long long tn = clock64(); atomicAdd(p.dbg + 32 + (slot), (unsigned long long)(tn - tq)); tq = tn; } } while (0)
	if (!pre.loaded)
		k2a_fetch<S>(pre, p, sc, dec_base, nbase, cnt);
	__syncthreads();
#pragma unroll
	for (int k = 0; k < K2aPre<S>::NL; ++k) {
		const int i = tid + k * K2A_THREADS;
		if (i < nx)
			sh.xs[S == 2 ? (i & 1) * K2A_XODD + (i >> 1) : i] = pre.v[k];
	}
	pre.loaded = false;
	if (next_cnt > 0)
		k2a_fetch<S>(pre, p, sc, dec_base, next_nbase, next_cnt);
	__syncthreads();
	K2A_STAMP(0);
#pragma unroll 1
	for (int r = 0; r < 4; ++r) {
		if (!(rmask & (1u << r)))
			continue;
		float mf[17];	/* wave-uniform: scalar registers */
#pragma unroll
		for (int j = 0; j < 17; ++j)	/* mflt[r + 64] exists only for r == 0 (16 taps otherwise) */
			mf[j] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(sh.smf[r + 4 * j])));
		const bool tap17 = (r == 0);
		/* ---- unit phasors of the filtered samples of instants -PH .. cnt-1 */
		for (int q = tid; q < cnt + PH; q += K2A_THREADS) {
			const v2f acc = k2a_fir<S>(sh, q, mf, tap17);
			const float n2 = __fmaf_rn(acc.x, acc.x, acc.y * acc.y);
			v2f w = acc * __frsqrt_rn(n2);
			if (!(n2 >= 1e-30f && n2 <= 1e30f)) {	/* atan2f(0, 0) = 0; anything else odd: let it through */
				const float bad = (acc.x == 0.0f && acc.y == 0.0f) ? 0.0f : __builtin_nanf("");
				w = (v2f){1.0f + bad, bad};
			}
			sh.wu[q] = make_float2(w.x, w.y);
		}
		K2A_STAMP(1);
		__syncthreads();
		K2A_STAMP(2);
		/* ---- in place: wu[q] <- wu[q] * conj(wu[q - LSTR]) */
		{
			v2f u[NQ];
			float2 a[NQ], b[NQ];
			const int qmax = cnt + PH - 1;
#pragma unroll
			for (int k = 0; k < NQ; ++k) {	/* clamped, unpredicated: all reads in flight together */
				int q = tid + k * K2A_THREADS;
				q = q < LSTR ? LSTR : (q > qmax ? qmax : q);
				a[k] = sh.wu[q];
				b[k] = sh.wu[q - LSTR];
			}
#pragma unroll
			for (int k = 0; k < NQ; ++k)
				u[k] = (v2f){__fmaf_rn(a[k].x, b[k].x, a[k].y * b[k].y), __fmaf_rn(a[k].y, b[k].x, -(a[k].x * b[k].y))};
			__syncthreads();
#pragma unroll
			for (int k = 0; k < NQ; ++k) {
				const int q = tid + k * K2A_THREADS;
				if (q >= LSTR && q < cnt + PH)
					sh.wu[q] = make_float2(u[k].x, u[k].y);
			}
		}
		K2A_STAMP(3);
		/* ---- the instants of the tile, all at once unless the worklist overflows (pathological
		 *      input such as a constant-phase tone): then in pieces it cannot overflow on */
		int piece = cnt;
		for (int c0 = 0; c0 < cnt;) {
			const int c1 = (c0 + piece < cnt) ? c0 + piece : cnt;
			if (tid == 0)
				sh.nwl = 0;
			__syncthreads();
			const int ndl0 = sh.ndl;	/* nobody appends between this barrier and the next */
			/* first screen, of the evaluation that is the `perr` of instant i: j = i + E2 */
			for (int i = c0 + tid; i < c1; i += K2A_THREADS) {
				const int j = i + E2;
				const float2 *uq = &sh.wu[PH - E4 + j - 15 * LSTR];
				float2 uu[16];
#pragma unroll
				for (int l = 0; l < 16; ++l)	/* all sixteen LDS reads in flight before the arithmetic */
					uu[l] = uq[l * LSTR];
				v2f acc = {0.0f, 0.0f}, acc1 = {0.0f, 0.0f};
#pragma unroll
				for (int l = 0; l < 16; l += 2) {
					acc = k2_rot(acc, (v2f){uu[l].x, uu[l].y}, rc[l], rs[l]);
					acc1 = k2_rot(acc1, (v2f){uu[l + 1].x, uu[l + 1].y}, rc[l + 1], rs[l + 1]);
				}
				acc += acc1;
				const float r2 = __fmaf_rn(acc.x, acc.x, acc.y * acc.y);
				if (!(r2 <= VDL2_SCREEN_R2)) {
					const int k = atomicAdd(&sh.nwl, 1);
					if (k < K2A_WL)
						sh.wl[k] = j;
				}
			}
			K2A_STAMP(4);
			__syncthreads();
			K2A_STAMP(5);
			const int nwl = sh.nwl;
			if (prof)
				atomicAdd(p.dbg + 32 + 10, (unsigned long long)nwl);
			if (nwl > K2A_WL) {
				piece = K2A_WL;
				__syncthreads();	/* everyone has read nwl before it is reset */
				continue;
			}
			if (ndl0 + nwl > K2A_DEF)	/* no room for this pass's survivors: work the list off first */
				k2a_flush(sh, p, sc, dec_base, mode, fail, skip_r, skip_par);
			/* second screen: lag-2 steps as products of neighbouring rotated lag-1 phasors */
			for (int k = tid; k < nwl; k += K2A_THREADS) {
				const int j = sh.wl[k];
				const float2 *uq = &sh.wu[PH - E4 + j - 15 * LSTR];
				v2f v[16];
#pragma unroll
				for (int l = 0; l < 16; ++l) {
					const float2 u = uq[l * LSTR];
					v[l] = k2_rot((v2f){0.0f, 0.0f}, (v2f){u.x, u.y}, rc[l], rs[l]);
				}
				v2f acc = {0.0f, 0.0f}, acc3 = {0.0f, 0.0f};
#pragma unroll
				for (int l = 0; l < 15; ++l) {
					const v2f p2 = k2_rot((v2f){0.0f, 0.0f}, v[l], v[l + 1].x, v[l + 1].y);	/* lag-2 step l */
					acc += p2;
					if (l < 14)	/* third screen: lag-3 steps, 14 of them */
						acc3 = k2_rot(acc3, p2, v[l + 2].x, v[l + 2].y);
				}
				const float r2 = __fmaf_rn(acc.x, acc.x, acc.y * acc.y);
				const float r3 = __fmaf_rn(acc3.x, acc3.x, acc3.y * acc3.y);
				if (!(r2 <= VDL2_SCREEN_R22) && !(r3 <= VDL2_SCREEN_R32)) {
					K2aDef d;
					d.n = (int)(nbase - dec_base) + S * (j - E4);
					d.r = r;
					d.lo = (mode == 1) ? (int)(chk_lo - dec_base) : 0;
					d.hi = (mode == 1) ? (int)(chk_hi - dec_base) : 0;
					sh.dl[atomicAdd(&sh.ndl, 1)] = d;
				}
			}
			K2A_STAMP(6);
			c0 = c1;
		}
		K2A_STAMP(8);
		__syncthreads();
		K2A_STAMP(9);
		if (prof)
			atomicAdd(p.dbg + 32 + 11, 1ull);
	}
#undef K2A_STAMP
}

__global__ __launch_bounds__(K2A_THREADS) __attribute__((amdgpu_waves_per_eu(4, 8)))
void k2a_probe(K2Params p)
{
	__shared__ K2aShared sh;
	const int c = blockIdx.y, s = blockIdx.z;
	const int sc = s * VDL2_CS + c;
	const StreamState *ss = p.ss + s;
	const long long dec_base = ss->dec_base;
	const long long avail_end = dec_base + ss->dec_fill + p.J;
	if (p.force_serial)
		return;
	k2a_tables(sh);
	/* each workgroup walks tiles blockIdx.x, blockIdx.x + gridDim.x, ... of its channel */
	if (p.full_scan) {
		K2aPre<1> pre;
		pre.loaded = false;
		const long long step = (long long)gridDim.x * K2A_TS;
		for (long long n0 = p.cs[sc].pos + (long long)blockIdx.x * K2A_TS; n0 < avail_end; n0 += step) {
			const int nt = (int)((avail_end - n0 < K2A_TS) ? (avail_end - n0) : K2A_TS);
			const long long n1 = n0 + step;
			const int nt1 = n1 < avail_end ? (int)((avail_end - n1 < K2A_TS) ? (avail_end - n1) : K2A_TS) : 0;
			k2a_tile<1>(sh, p, sc, dec_base, n0, nt, 0xfu, 0, 0, 0, nullptr, pre, n1, nt1);
		}
		k2a_flush(sh, p, sc, dec_base, 0, nullptr, -1, 0);
		return;
	}
	/* the class the channel's detector is in right now: sub-phase r, parity of pos */
	K2aPre<2> pre;
	pre.loaded = false;
	const unsigned rmask = 1u << p.cs[sc].r;
	const long long step = 2LL * gridDim.x * K2A_TS;
	for (long long n0 = p.cs[sc].pos + 2LL * blockIdx.x * K2A_TS; n0 < avail_end; n0 += step) {
		const long long left = (avail_end - n0 + 1) / 2;
		const int nt = (int)(left < K2A_TS ? left : K2A_TS);
		const long long n1 = n0 + step;
		const long long left1 = (avail_end - n1 + 1) / 2;
		const int nt1 = n1 < avail_end ? (int)(left1 < K2A_TS ? left1 : K2A_TS) : 0;
		k2a_tile<2>(sh, p, sc, dec_base, n0, nt, rmask, 2, 0, 0, nullptr, pre, n1, nt1);
	}
	k2a_flush(sh, p, sc, dec_base, 2, nullptr, -1, 0);
}

/* ---- workgroup sort of up to VDL2_CAND_CAP 64-bit keys whose top bits are a time stamp.
 * Detector events are spread over the push (a few per burst, bursts are sparse), so a bucket
 * sort on time -- histogram, scan, scatter, then a short insertion sort inside each bucket -- needs
 * about a dozen barriers where a bitonic network needs log^2(n)/2 = 78.  A push whose events
 * pile up in one bucket (more than WGS_MAXB) falls back to the bitonic network.
 *   keys[] in/out (LDS), tmp[] scratch (LDS), both VDL2_CAND_CAP long; time = key >> tshift, < range. */
#define WGS_NBK 2048
#define WGS_MAXB 48
struct WgSortShared {
	unsigned long long tmp[VDL2_CAND_CAP];
	unsigned start[WGS_NBK + 1], cur[WGS_NBK];
	unsigned part[64];
	unsigned maxb;
};

template <int NT> __device__ void wg_sort_u64(unsigned long long *keys, WgSortShared &ws, int n, int tshift, unsigned range)
{
	const int tid = threadIdx.x;
	int bsh = 0;
	while ((range >> bsh) >= WGS_NBK)
		++bsh;
	for (int b = tid; b < WGS_NBK; b += NT)
		ws.cur[b] = 0;
	if (tid == 0)
		ws.maxb = 0;
	__syncthreads();
	for (int i = tid; i < n; i += NT) {
		unsigned b = (unsigned)(keys[i] >> tshift) >> bsh;
		b = b < WGS_NBK ? b : WGS_NBK - 1;
		const unsigned k = atomicAdd(&ws.cur[b], 1u);
		if (k + 1 > WGS_MAXB)
			ws.maxb = 1;
	}
	__syncthreads();
	if (ws.maxb) {
		/* crowded bucket: bitonic network over the next power of two */
		int npow = 1;
		while (npow < n)
			npow <<= 1;
		for (int i = n + tid; i < npow; i += NT)
			keys[i] = ~0ull;
		__syncthreads();
		for (int k = 2; k <= npow; k <<= 1)
			for (int j = k >> 1; j > 0; j >>= 1) {
				for (int i = tid; i < npow; i += NT) {
					const int l = i ^ j;
					if (l > i) {
						const unsigned long long a0 = keys[i], b0 = keys[l];
						if ((a0 > b0) == ((i & k) == 0)) {
							keys[i] = b0;
							keys[l] = a0;
						}
					}
				}
				__syncthreads();
			}
		return;
	}
	/* exclusive scan of the bucket counts: per-thread run of WGS_NBK / NT buckets, then a scan of the run sums */
	constexpr int RUN = (WGS_NBK + NT - 1) / NT;
	{
		unsigned sum = 0;
		for (int k = 0; k < RUN; ++k) {
			const int b = tid * RUN + k;
			if (b < WGS_NBK)
				sum += ws.cur[b];
		}
		/* wave-level inclusive scan, then the wave totals */
		unsigned incl = sum;
		for (int d = 1; d < 64; d <<= 1) {
			const unsigned o = __shfl_up(incl, d, 64);
			if ((tid & 63) >= d)
				incl += o;
		}
		if ((tid & 63) == 63)
			ws.part[tid >> 6] = incl;
		__syncthreads();
		unsigned base = 0;
		for (int w = 0; w < (tid >> 6); ++w)
			base += ws.part[w];
		unsigned run = base + incl - sum;
		for (int k = 0; k < RUN; ++k) {
			const int b = tid * RUN + k;
			if (b < WGS_NBK) {
				const unsigned cnt = ws.cur[b];
				ws.start[b] = run;
				ws.cur[b] = run;
				run += cnt;
			}
		}
		if (tid == NT - 1)
			ws.start[WGS_NBK] = run;
	}
	__syncthreads();
	for (int i = tid; i < n; i += NT) {
		const unsigned long long v = keys[i];
		unsigned b = (unsigned)(v >> tshift) >> bsh;
		b = b < WGS_NBK ? b : WGS_NBK - 1;
		ws.tmp[atomicAdd(&ws.cur[b], 1u)] = v;
	}
	__syncthreads();
	for (int b = tid; b < WGS_NBK; b += NT) {
		const int lo = (int)ws.start[b], hi = (int)ws.start[b + 1];
		for (int i = lo + 1; i < hi; ++i) {
			const unsigned long long v = ws.tmp[i];
			int j = i - 1;
			while (j >= lo && ws.tmp[j] > v) {
				ws.tmp[j + 1] = ws.tmp[j];
				--j;
			}
			ws.tmp[j + 1] = v;
		}
	}
	__syncthreads();
	for (int i = tid; i < n; i += NT)
		keys[i] = ws.tmp[i];
	__syncthreads();
}

/* ---- regions around the probe's hits (one workgroup per channel) */
#define K2R_NT 1024
__global__ __launch_bounds__(K2R_NT)
void k2r_regions(K2Params p)
{
	__shared__ unsigned long long key64[VDL2_CAND_CAP];
	__shared__ WgSortShared ws;
	__shared__ int key[VDL2_CAND_CAP];
	const int tid = threadIdx.x;
	const int c = blockIdx.x, s = blockIdx.y;
	const int sc = s * VDL2_CS + c;
	if (p.force_serial || p.full_scan)
		return;
	if (p.round > 0 && p.fail[sc] >= VDL2_VERIFIED)
		return;		/* repair round: only channels whose verify pass found something */
	int ncand = (int)p.ctl[CTL_NSEED0 + sc];
	ncand = ncand > VDL2_CAND_CAP ? VDL2_CAND_CAP : ncand;
	const int *seeds = p.seeds + (size_t)sc * VDL2_CAND_CAP;
	for (int i = tid; i < ncand; i += K2R_NT)
		key64[i] = (unsigned long long)(unsigned)seeds[i];
	__syncthreads();
	wg_sort_u64<K2R_NT>(key64, ws, ncand, 0, (unsigned)(p.ss[s].dec_fill + p.J));
	for (int i = tid; i < ncand; i += K2R_NT)
		key[i] = (int)key64[i];
	__syncthreads();
	{
		/* every run of hits closer than VDL2_REG_GAP becomes a region (order is irrelevant) */
		__shared__ int s_nreg;
		const StreamState *ss = p.ss + s;
		const int lo_lim = (int)(p.cs[sc].pos - ss->dec_base);
		const int hi_lim = (int)(ss->dec_fill + p.J);
		int2 *regs = p.regs + (size_t)sc * VDL2_REG_CAP;
		if (tid == 0)
			s_nreg = 0;
		__syncthreads();
		for (int i = tid; i < ncand; i += K2R_NT) {
			if (i > 0 && key[i] - key[i - 1] <= VDL2_REG_GAP)
				continue;	/* not the first hit of its run */
			int j = i;
			while (j + 1 < ncand && key[j + 1] - key[j] <= VDL2_REG_GAP)
				++j;
			int lo = key[i] - VDL2_REG_PAD, hi = key[j] + VDL2_REG_PAD + 1;
			lo = lo < lo_lim ? lo_lim : lo;
			hi = hi > hi_lim ? hi_lim : hi;
			if (hi <= lo)
				continue;
			/* long merged regions (bursts back to back) are cut into tile-sized pieces */
			const int nchunk = (hi - lo + K2A_TS - 1) / K2A_TS;
			const int base = atomicAdd(&s_nreg, nchunk);
			for (int k = 0; k < nchunk; ++k)
				if (base + k < VDL2_REG_CAP) {
					const int q = lo + k * K2A_TS;
					regs[base + k] = make_int2(q, (hi - q < K2A_TS) ? hi - q : K2A_TS);
				}
		}
		__syncthreads();
		if (tid == 0) {
			const int n = s_nreg;
			p.ctl[CTL_NREG0 + sc] = (unsigned)(n > VDL2_REG_CAP ? VDL2_REG_CAP : n);
			p.ctl[CTL_NSEED0 + sc] = 0;	/* the seed list now collects what K2a-verify finds */
			if (n > VDL2_REG_CAP)
				p.ctl[CTL_CAND0 + p.nstreams * VDL2_CS + sc] = 1u;	/* tables unusable -> serial */
		}
	}
}

__global__ __launch_bounds__(K2A_THREADS) __attribute__((amdgpu_waves_per_eu(4, 8)))
void k2a_region(K2Params p)
{
	__shared__ K2aShared sh;
	const int c = blockIdx.y, s = blockIdx.z;
	const int sc = s * VDL2_CS + c;
	if (p.force_serial || p.full_scan || (p.test_noregion && p.round == 0))
		return;
	if (p.round > 0 && p.fail[sc] >= VDL2_VERIFIED)
		return;
	const unsigned nreg = p.ctl[CTL_NREG0 + sc];
	const long long dec_base = p.ss[s].dec_base;
	const int2 *regs = p.regs + (size_t)sc * VDL2_REG_CAP;
	const int skip_r = p.cs[sc].r, skip_par = (int)(p.cs[sc].pos & 1);
	k2a_tables(sh);
	K2aPre<1> pre;
	pre.loaded = false;
	for (unsigned k = blockIdx.x; k < nreg; k += gridDim.x) {
		const int2 rg = regs[k];
		const int2 rn = (k + gridDim.x < nreg) ? regs[k + gridDim.x] : make_int2(0, 0);
		k2a_tile<1>(sh, p, sc, dec_base, dec_base + rg.x, rg.y, 0xfu, 0, 0, 0, nullptr, pre, dec_base + rn.x, rn.y, skip_r, skip_par);
	}
	k2a_flush(sh, p, sc, dec_base, 0, nullptr, skip_r, skip_par);
}

/* one workgroup = K2A_VRUN tiles of 2*K2A_TS samples; every piece of a verify segment inside a tile
 * is scanned in the segment's class */
#define K2A_VRUN 4
#define K2A_VITEMS 64
__global__ __launch_bounds__(K2A_THREADS) __attribute__((amdgpu_waves_per_eu(4, 8)))
void k2a_verify(K2Params p)
{
	__shared__ K2aShared sh;
	__shared__ int s_list[64], s_nl, s_ni;
	__shared__ int4 s_item[K2A_VITEMS];	/* lo, hi (stream-relative samples), sub-phase */
	const int tid = threadIdx.x;
	const int c = blockIdx.y, s = blockIdx.z;
	const int sc = s * VDL2_CS + c;
	if (p.force_serial || p.full_scan)
		return;
	if (p.round > 0 && !p.redo[sc])
		return;
	const StreamState *ss = p.ss + s;
	const long long dec_base = ss->dec_base;
	const int r_lo = (int)(p.cs[sc].pos - dec_base) + (int)blockIdx.x * K2A_VRUN * 2 * K2A_TS;
	const int t_end = (int)(ss->dec_fill + p.J);
	if (r_lo >= t_end)
		return;
	const int r_hi = r_lo + K2A_VRUN * 2 * K2A_TS < t_end ? r_lo + K2A_VRUN * 2 * K2A_TS : t_end;
	const int nseg = (int)p.ctl[CTL_NSEG0 + sc];
	const Seg *segs = p.segs + (size_t)sc * VDL2_SEG_CAP;
	k2a_tables(sh);
	if (tid == 0)
		s_nl = 0;
	__syncthreads();
	for (int k = tid; k < nseg && k < VDL2_SEG_CAP; k += K2A_THREADS) {
		const Seg g = segs[k];
		if (g.lo < r_hi && g.hi > r_lo && g.hi > g.lo) {
			const int q = atomicAdd(&s_nl, 1);
			if (q < 64)
				s_list[q] = k;
		}
	}
	__syncthreads();
	const int nl = s_nl;
	if (tid == 0) {
		int ni = 0;
		for (int q = 0; q < nl && q < 64; ++q) {
			const Seg g = segs[s_list[q]];
			for (int t_lo = r_lo; t_lo < r_hi; t_lo += 2 * K2A_TS) {
				const int t_hi = t_lo + 2 * K2A_TS < r_hi ? t_lo + 2 * K2A_TS : r_hi;
				int lo = g.lo > t_lo ? g.lo : t_lo;
				const int hi = g.hi < t_hi ? g.hi : t_hi;
				lo += (lo ^ g.lo) & 1;		/* keep the segment's parity */
				if (lo >= hi)
					continue;
				if (ni < K2A_VITEMS)
					s_item[ni] = make_int4(lo, hi, g.r, 0);
				++ni;
			}
		}
		s_ni = ni;
	}
	__syncthreads();
	const int ni = s_ni;
	if (nl > 64 || ni > K2A_VITEMS) {	/* absurdly fragmented stretch: give up on the tables for this channel */
		if (tid == 0)
			atomicMin(p.fail + sc, 0);
		return;
	}
	K2aPre<2> pre;
	pre.loaded = false;
	for (int q = 0; q < ni; ++q) {
		const int4 it = s_item[q];
		const int4 nx = (q + 1 < ni) ? s_item[q + 1] : make_int4(0, 0, 0, 0);
		k2a_tile<2>(sh, p, sc, dec_base, dec_base + it.x, (it.y - it.x + 1) / 2, 1u << it.z, 1, dec_base + it.x, dec_base + it.y,
			    p.fail + sc, pre, dec_base + nx.x, (nx.y - nx.x + 1) / 2);
	}
	k2a_flush(sh, p, sc, dec_base, 1, p.fail + sc, -1, 0);
}

/* ====================================================================== K2s
 * Per channel: sort the candidates by time (bitonic network in LDS) for the resolver, and pick the
 * ones whose cluster is worth precomputing: the first of its (sub-phase, parity) class within a
 * burst's worth of samples.  A later candidate of the same class can only be reached if the detector
 * turns history-free in the few samples between the two; the resolver computes such a cluster itself
 * when it ever needs one (status CL_INVALID), so this is a cost decision, never a correctness one.
 */
#define K2S_NT 1024
#define K2S_LOOKBACK 72		/* a triggered detector is busy for at least 9 symbols = 72 samples */
__global__ __launch_bounds__(K2S_NT)
void k2s_sort(K2Params p)
{
	__shared__ unsigned long long sbuf[VDL2_CAND_CAP];
	__shared__ WgSortShared ws;
	__shared__ int s_np;
	const int tid = threadIdx.x;
	const int c = blockIdx.x, s = blockIdx.y;
	const int sc = s * VDL2_CS + c;
	if (p.force_serial)
		return;
	if (p.round > 0 && p.fail[sc] >= VDL2_VERIFIED)
		return;
	int ncand = (int)p.ctl[CTL_CAND0 + sc];
	if (ncand > VDL2_CAND_CAP || p.ctl[CTL_CAND0 + p.nstreams * VDL2_CS + sc] != 0)
		return;		/* tables unusable: the resolver runs serially */
	const Cand *cands = p.cands + (size_t)sc * VDL2_CAND_CAP;
	for (int i = tid; i < ncand; i += K2S_NT)
		sbuf[i] = (((unsigned long long)(unsigned)(cands[i].nrel * 4 + cands[i].r)) << 16) | (unsigned)i;
	if (tid == 0)
		s_np = 0;
	__syncthreads();
	wg_sort_u64<K2S_NT>(sbuf, ws, ncand, 18, (unsigned)(p.ss[s].dec_fill + p.J));
	int *skey = p.skey + (size_t)sc * VDL2_CAND_CAP;
	unsigned short *sidx = p.sidx + (size_t)sc * VDL2_CAND_CAP;
	unsigned short *prim = p.prim + (size_t)sc * VDL2_CAND_CAP;
	for (int j = tid; j < ncand; j += K2S_NT) {
		const unsigned long long v = sbuf[j];
		const int key = (int)(v >> 16), idx = (int)(v & 0xffffu);
		skey[j] = key;
		sidx[j] = (unsigned short)idx;
		const int n = key >> 2, cls = (key & 3) * 2 + (n & 1);
		bool primary = true;
		for (int i = j - 1; i >= 0; --i) {
			const int ki = (int)(sbuf[i] >> 16), ni = ki >> 2;
			if (n - ni >= K2S_LOOKBACK)
				break;
			if ((ki & 3) * 2 + (ni & 1) == cls) {
				primary = false;
				break;
			}
		}
		if (primary)
			prim[atomicAdd(&s_np, 1)] = (unsigned short)idx;
		else
			p.clhead[(size_t)sc * VDL2_CAND_CAP + idx] = cl_pack(0, CL_INVALID, 0, 0, 0, 0, 0);
	}
	__syncthreads();
	if (tid == 0)
		p.ctl[CTL_NPRIM0 + sc] = (unsigned)s_np;
}

/* ====================================================================== K2b
 * One workgroup per trigger candidate (persistent workgroups pull tickets):
 * put the detector in the history-free state at the candidate, run the exact
 * machine through the burst (and any burst that follows before the detector is
 * history-free again) and record where and how the idle search resumes.
 */
#ifndef K2B_WAVES
#define K2B_WAVES 4
#endif
__global__ __launch_bounds__(K2B_NT) __attribute__((amdgpu_waves_per_eu(K2B_WAVES, 8)))
void k2b_clusters(K2Params p)
{
	__shared__ MachSharedT<K2B_NT> sh;
	__shared__ float sgrey[3 * 257];
	__shared__ unsigned s_pref[65];
	const int tid = threadIdx.x;
	const int nsc = p.nstreams * VDL2_CS;
	if (p.force_serial)
		return;
	for (int i = tid; i < 257; i += K2B_NT) {
		sgrey[i] = d_tab(c_grey1, i);
		sgrey[257 + i] = d_tab(c_grey2, i);
		sgrey[514 + i] = d_tab(c_grey3, i);
	}
	mach_init_taps(sh);
	/* blockIdx.y selects a group of up to 64 (stream, channel) slots; exclusive prefix of the
	 * group's cluster counts maps a ticket to (slot, primary candidate) */
	const int sc0 = (int)blockIdx.y * 64;
	const int nsc64 = (nsc - sc0) < 64 ? (nsc - sc0) : 64;
	if (tid == 0) {
		unsigned acc = 0;
		for (int k = 0; k < nsc64; ++k) {
			unsigned n = p.ctl[CTL_NPRIM0 + sc0 + k];
			n = n > VDL2_CAND_CAP ? VDL2_CAND_CAP : n;
			if (p.round > 0 && p.fail[sc0 + k] >= VDL2_VERIFIED)
				n = 0;
			s_pref[k] = acc;
			acc += n;
		}
		s_pref[nsc64] = acc;
	}
	__syncthreads();
	const unsigned total = s_pref[nsc64];
	for (unsigned tk = blockIdx.x; tk < total; tk += gridDim.x) {
		int scl = 0;
		while (scl + 1 < nsc64 && s_pref[scl + 1] <= tk)
			++scl;
		const int sc = sc0 + scl;
		const int idx = (int)p.prim[(size_t)sc * VDL2_CAND_CAP + (tk - s_pref[scl])];
		const int s = sc / VDL2_CS, c = sc % VDL2_CS;
		MachCtx cx;
		mach_ctx(cx, p, s, c, true);
		const Cand cd = p.cands[(size_t)sc * VDL2_CAND_CAP + idx];
		Cluster *cl = p.clusters + (size_t)sc * VDL2_CAND_CAP + idx;
		MachState st;
		st.pos = cx.dec_base + cd.nrel;
		st.r = cd.r;
		st.fresh = VDL2_STEADY;
		cx.grey = sgrey;
		cx.desc_static = ((long long)sc * VDL2_CAND_CAP + idx) * VDL2_CL_MAXB;
		MachOut out;
		out.nslots = out.ntrig = out.nrej = out.nburst = out.ndefer = 0;
		out.neval = 0;
#pragma unroll
		for (int i = 0; i < VDL2_CL_MAXB; ++i)
			out.slots[i] = 0;
		const long long t0 = wall_clock64();
		mach_materialize<K2B_NT, true>(sh, cx, st.pos, st.r);
		const long long t1 = wall_clock64();
		const int rc = machine_run<K2B_NT, true>(sh, cx, st, true, 1, VDL2_CL_MAXB, 1, out);
		const long long t2 = wall_clock64();
		if (tid == 0 && p.dbg && (tk & 15u) == 0) {
			atomicAdd(p.dbg + 0, (unsigned long long)(t1 - t0));
			atomicAdd(p.dbg + 1, (unsigned long long)(t2 - t1));
			atomicAdd(p.dbg + 2, 1ull);
			atomicAdd(p.dbg + 3, (unsigned long long)out.ntrig);
			atomicAdd(p.dbg + 4, (unsigned long long)out.neval);
			atomicMax(p.dbg + 5, (unsigned long long)(t2 - t1));
			atomicAdd(p.dbg + 6, (unsigned long long)(rc == MR_STEADY));
			atomicAdd(p.dbg + 7, (unsigned long long)out.nrej);
		}
		int status;
		if (rc == MR_STEADY)
			status = CL_STEADY;
		else if (rc == MR_DEFER && out.ntrig == 0)
			status = CL_DEFER_FIRST;
		else
			status = CL_NONSTEADY;
		bool bad = false;
#pragma unroll
		for (int i = 0; i < VDL2_CL_MAXB; ++i)
			if (i < out.nslots && out.slots[i] < 0)
				bad = true;	/* descriptor pool full */
		if (bad)
			status = CL_INVALID;
		if (status == CL_NONSTEADY)
			mach_store(sh, st, &cl->saved);
		if (tid == 0)	/* descriptors sit in static slots desc_static + i: the head only needs their number */
			p.clhead[(size_t)sc * VDL2_CAND_CAP + idx] =
			    cl_pack((int)(st.pos - cx.dec_base), status, st.r, out.nslots < VDL2_CL_MAXB ? out.nslots : VDL2_CL_MAXB,
				    out.ntrig, out.nrej, out.nburst);
		__syncthreads();
	}
}

/* ====================================================================== K2c
 * Resolver: one workgroup per VDL channel follows the real chain of events.
 * While the detector is history-free the next event is simply the first
 * candidate of the current (sub-phase, sample parity) at or after `pos`, and
 * its consequences were precomputed by K2b; otherwise the serial machine runs
 * until the detector is history-free again.
 *   1. rank-sort the channel's candidates by time                 (parallel)
 *   2. for every candidate: status + index of the candidate that  (parallel)
 *      follows its cluster  -> successor table in LDS
 *   3. walk the chain through the successor table                 (one lane, LDS only)
 *   4. mark the staged bursts of the visited clusters, add counters (parallel)
 */
#define K2C_NOCAND 0xffffu

/* first sorted candidate at/after stream-relative time `want` of class (r, parity), from `from` */
__device__ __forceinline__ int k2c_next(const int *skey, int ncand, int from, int want, int r)
{
	/* lower bound on time */
	int lo = from, hi = ncand;
	while (lo < hi) {
		const int mid = (lo + hi) >> 1;
		if ((skey[mid] >> 2) < want)
			lo = mid + 1;
		else
			hi = mid;
	}
	for (; lo < ncand; ++lo) {
		const int k = skey[lo];
		if ((k & 3) == r && (((k >> 2) - want) & 1) == 0)
			return lo;
	}
	return -1;
}

__global__ __launch_bounds__(K2_NT)
void k2c_resolve(K2Params p)
{
	__shared__ MachSharedT<K2_NT> sh;
	__shared__ int skey[VDL2_CAND_CAP];		/* sorted keys: nrel*4 + r */
	__shared__ unsigned short sidx[VDL2_CAND_CAP];	/* sorted rank -> candidate index */
	__shared__ unsigned short snext[VDL2_CAND_CAP];	/* rank of the candidate that follows the cluster */
	__shared__ uint8_t sstat[VDL2_CAND_CAP];	/* cluster status */
	__shared__ uint8_t ssel[VDL2_CAND_CAP];		/* visited by the real chain */
	__shared__ int2 shead[VDL2_CAND_CAP];		/* cl_pack() of every candidate's cluster, by sorted rank */
	__shared__ int s_walk[4];
	__shared__ int s_cnt[4];
	const int tid = threadIdx.x;
	const int c = blockIdx.x, s = blockIdx.y;
	const int sc = s * VDL2_CS + c;
	if (p.round > 0) {
		if (p.fail[sc] >= VDL2_VERIFIED)
			return;		/* verified in the first pass: nothing to repair */
		__syncthreads();
		if (tid == 0) {
			p.redo[sc] = 1;
			atomicAdd(p.outc_total_redo + 1, 1u);	/* channel-pushes that went through a repair round */
			if (p.dbg)
				atomicAdd(p.dbg + 24, 1ull);
			p.fail[sc] = 0x7f7f7f7f;	/* the repair pass is verified afresh */
			p.ctl[CTL_NSEL0 + sc] = 0;
			p.ctl[CTL_NSEG0 + sc] = 0;
		}
		__syncthreads();
	}
	const ChanState *cs = p.cs + sc;	/* input state: left untouched until K2f commits */
	ChanState *cs_out = p.cs_out + sc;
	unsigned *sel = p.sel_list + (size_t)sc * VDL2_SEL_CAP;
	unsigned *nsel = p.ctl + CTL_NSEL0 + sc;
	Seg *segs = p.segs + (size_t)sc * VDL2_SEG_CAP;
	unsigned *nseg = p.ctl + CTL_NSEG0 + sc;
	MachCtx cx;
	mach_ctx(cx, p, s, c, true);	/* bursts of serial stretches become descriptors too */
	cx.sel = sel;
	cx.nsel = nsel;
	cx.dbg = nullptr;
	MachState st;
	st.pos = cs->pos;
	st.r = cs->r;
	st.fresh = cs->fresh;
	const int r_probe = cs->r;
	const int par_probe = (int)(cs->pos & 1);	/* the probe scanned class (r_probe, par_probe) everywhere */
	const int t_end = (int)(cx.avail_end - cx.dec_base);
	const bool lazy = !p.full_scan;
	mach_init_taps(sh);
	mach_load(sh, cs);
	MachOut out;
	out.nslots = out.ntrig = out.nrej = out.nburst = out.ndefer = 0;
	out.neval = 0;
	unsigned long long n_slow = 0;
	int ncand = (int)p.ctl[CTL_CAND0 + sc];
	const bool tables_ok = !p.force_serial && ncand <= VDL2_CAND_CAP && p.ctl[CTL_CAND0 + p.nstreams * VDL2_CS + sc] == 0;
	if (!tables_ok)
		ncand = 0;
	const Cluster *clusters = p.clusters + (size_t)sc * VDL2_CAND_CAP;
	/* 1. candidates sorted by time (K2s) */
	const long long pos_in = st.pos;
	const long long tk0 = wall_clock64();
	{
		const int2 *head = p.clhead + (size_t)sc * VDL2_CAND_CAP;
		for (int i = tid; i < ncand; i += K2_NT) {
			const int idx = p.sidx[(size_t)sc * VDL2_CAND_CAP + i];
			skey[i] = p.skey[(size_t)sc * VDL2_CAND_CAP + i];
			sidx[i] = (unsigned short)idx;
			shead[i] = head[idx];
		}
	}
	__syncthreads();
	const long long tk1 = wall_clock64();
	/* 2. successor table */
	for (int j = tid; j < ncand; j += K2_NT) {
		const int2 hd = shead[j];
		const int status = hd.y & 3;
		int nx = -1;
		if (status == CL_STEADY)
			nx = k2c_next(skey, ncand, j + 1, hd.x, (hd.y >> 2) & 3);
		sstat[j] = (uint8_t)status;
		snext[j] = (nx < 0) ? (unsigned short)K2C_NOCAND : (unsigned short)nx;
		ssel[j] = 0;
	}
	__syncthreads();
	const long long tk2 = wall_clock64();
	bool steady_end = false;
	for (;;) {
		if (!tables_ok || st.fresh < VDL2_STEADY) {
			/* history-dependent stretch (or no tables): serial machine */
			const long long p0 = st.pos;
			const int rc = machine_run<K2_NT, false>(sh, cx, st, tables_ok, 0, 1 << 30, 0, out);
			n_slow += (unsigned long long)(st.pos - p0);
			if (rc != MR_STEADY)
				break;
			continue;
		}
		/* 3. history-free: walk the successor table until something special happens */
		if (tid == 0) {
			int cur = k2c_next(skey, ncand, 0, (int)(st.pos - cx.dec_base), st.r);
			int last = -1, why = 0;	/* why: 0 = no more candidates, 1 = special cluster at cur */
			if (lazy && (st.r != r_probe || (int)(st.pos & 1) != par_probe)) {
				/* the chain idles from here to the next candidate in a class the probe did not
				 * scan: K2a-verify must confirm there really is nothing in between */
				const unsigned q = atomicAdd(nseg, 1u);
				if (q < VDL2_SEG_CAP) {
					Seg g;
					g.lo = (int)(st.pos - cx.dec_base);
					g.hi = (cur >= 0) ? (skey[cur] >> 2) : t_end;
					g.r = st.r;
					g.pad = 0;
					segs[q] = g;
				} else
					atomicMin(p.fail + sc, 0);
			}
			while (cur >= 0) {
				const int stt = sstat[cur];
				if (stt != CL_STEADY) {
					why = 1;
					break;
				}
				ssel[cur] = 1;
				last = cur;
				const int nx = snext[cur];
				cur = (nx == K2C_NOCAND) ? -1 : nx;
			}
			s_walk[0] = cur;
			s_walk[1] = last;
			s_walk[2] = why;
		}
		__syncthreads();
		const int cur = s_walk[0], last = s_walk[1], why = s_walk[2];
		__syncthreads();
		if (last >= 0) {
			st.pos = cx.dec_base + shead[last].x;
			st.r = (shead[last].y >> 2) & 3;
		}
		if (!why) {
			/* idle to the end of the data: next evaluation is the first one past it */
			const long long rem = (cx.avail_end - st.pos + 1) / 2;
			if (rem > 0)
				st.pos += 2 * rem;
			steady_end = true;
			break;
		}
		const long long ncand_t = cx.dec_base + (skey[cur] >> 2);
		const Cluster *cl = clusters + sidx[cur];
		const int status = sstat[cur];
		if (status == CL_DEFER_FIRST) {
			st.pos = ncand_t;
			out.ndefer++;
			steady_end = true;
			break;
		}
		if (status == CL_INVALID) {
			/* staging pool was full: replay this stretch here */
			st.pos = ncand_t;
			mach_materialize<K2_NT, false>(sh, cx, st.pos, st.r);
			const int rc = machine_run<K2_NT, false>(sh, cx, st, true, 1, 1 << 30, 1, out);
			if (rc != MR_STEADY)
				break;
			continue;
		}
		/* CL_NONSTEADY: its bursts count, then continue from the explicit state it stopped in */
		if (tid == 0)
			ssel[cur] = 1;
		mach_load(sh, &cl->saved);
		st.pos = cl->saved.pos;
		st.r = cl->saved.r;
		st.fresh = cl->saved.fresh < VDL2_STEADY ? cl->saved.fresh : VDL2_STEADY - 1;
	}
	__syncthreads();
	const long long tk3 = wall_clock64();
	/* 4. publish the visited clusters: list positions come from LDS counters seeded with what the
	 *    serial stretches and the walk already listed; the global counters are written once */
	if (tid < 4)
		s_cnt[tid] = 0;
	if (tid == 0) {
		s_walk[0] = (int)*nsel;
		s_walk[1] = (int)*nseg;
	}
	__syncthreads();
	{
		int a = 0, b = 0, d = 0;
		for (int j = tid; j < ncand; j += K2_NT)
			if (ssel[j]) {
				const int2 hd = shead[j];
				const int ns = (hd.y >> 4) & 15;
				/* K2b's descriptors sit in static slots: (candidate index) * VDL2_CL_MAXB + burst */
				const unsigned slot0 = (unsigned)(((size_t)sc * VDL2_CAND_CAP + sidx[j]) * VDL2_CL_MAXB);
				if (ns) {
					const unsigned q = (unsigned)atomicAdd(&s_walk[0], ns);
					for (int i = 0; i < ns; ++i) {
						if (q + i < VDL2_SEL_CAP)
							sel[q + i] = slot0 + i;
						else
							atomicAdd(p.outc + 1, 1u);
					}
				}
				a += (hd.y >> 8) & 255;
				b += (hd.y >> 16) & 255;
				d += (hd.y >> 24) & 255;
				const int r_s = (hd.y >> 2) & 3;
				const long long n_s = cx.dec_base + hd.x;
				if (lazy && sstat[j] == CL_STEADY && (r_s != r_probe || (int)(n_s & 1) != par_probe)) {
					/* after this cluster the chain idles in class (r_s, parity of n_s) until
					 * the successor's trigger (or the end of the data) */
					const unsigned q = (unsigned)atomicAdd(&s_walk[1], 1);
					if (q < VDL2_SEG_CAP) {
						Seg g;
						g.lo = hd.x;
						g.hi = (snext[j] == K2C_NOCAND) ? t_end : (skey[snext[j]] >> 2);
						g.r = r_s;
						g.pad = 0;
						segs[q] = g;
					} else
						atomicMin(p.fail + sc, 0);
				}
			}
		if (a)
			atomicAdd(&s_cnt[0], a);
		if (b)
			atomicAdd(&s_cnt[1], b);
		if (d)
			atomicAdd(&s_cnt[2], d);
	}
	__syncthreads();
	if (tid == 0) {
		*nsel = (unsigned)s_walk[0];
		*nseg = (unsigned)s_walk[1];
	}
	__syncthreads();
	if (steady_end) {
		mach_materialize<K2_NT, false>(sh, cx, st.pos, st.r);
		st.fresh = VDL2_STEADY;
	}
	__syncthreads();
	mach_store(sh, st, cs_out);
	if (tid == 0) {
		cs_out->n_eval = cs->n_eval + (unsigned long long)((st.pos - pos_in) / 2);	/* evaluation instants covered */
		cs_out->n_trig = cs->n_trig + (unsigned long long)(out.ntrig + s_cnt[0]);
		cs_out->n_reject = cs->n_reject + (unsigned long long)(out.nrej + s_cnt[1]);
		cs_out->n_burst = cs->n_burst + (unsigned long long)(out.nburst + s_cnt[2]);
		cs_out->n_defer = cs->n_defer + (unsigned long long)out.ndefer;
		cs_out->n_slow = cs->n_slow + n_slow;
		cs_out->n_cand = cs->n_cand + (unsigned long long)ncand;
		cs_out->n_redo = cs->n_redo;
		if (p.dbg) {
			const long long tk4 = wall_clock64();
			atomicAdd(p.dbg + 16, (unsigned long long)(tk1 - tk0));
			atomicAdd(p.dbg + 17, (unsigned long long)(tk2 - tk1));
			atomicAdd(p.dbg + 18, (unsigned long long)(tk3 - tk2));
			atomicAdd(p.dbg + 19, (unsigned long long)(tk4 - tk3));
			atomicAdd(p.dbg + 20, 1ull);
		}
	}
}

/* ====================================================================== K2f
 * Commit.  If K2a-verify found nothing the resolver's result becomes the channel state.  If it
 * found a detector hit the tables did not contain, the channel's push is redone from its input
 * state by the serial machine alone (always exact; it writes its bursts itself) and the
 * resolver's selection for that channel is dropped.
 */
__global__ __launch_bounds__(K2_NT)
void k2f_commit(K2Params p)
{
	__shared__ MachSharedT<K2_NT> sh;
	const int tid = threadIdx.x;
	const int c = blockIdx.x, s = blockIdx.y;
	const int sc = s * VDL2_CS + c;
	ChanState *cs = p.cs + sc;
	if (p.fail[sc] >= VDL2_VERIFIED) {
		const uint32_t *src = reinterpret_cast<const uint32_t *>(p.cs_out + sc);
		uint32_t *dst = reinterpret_cast<uint32_t *>(cs);
		for (int i = tid; i < (int)(sizeof(ChanState) / 4); i += K2_NT)
			dst[i] = src[i];
		return;
	}
	MachCtx cx;
	mach_ctx(cx, p, s, c, false);
	cx.dbg = nullptr;
	MachState st;
	st.pos = cs->pos;
	st.r = cs->r;
	st.fresh = cs->fresh;
	const long long p0 = st.pos;
	mach_init_taps(sh);
	mach_load(sh, cs);
	MachOut out;
	out.nslots = out.ntrig = out.nrej = out.nburst = out.ndefer = 0;
	out.neval = 0;
	machine_run<K2_NT, false>(sh, cx, st, false, 0, 1 << 30, 0, out);
	__syncthreads();
	mach_store(sh, st, cs);
	if (tid == 0) {
		cs->n_eval += (unsigned long long)out.neval;
		cs->n_trig += (unsigned long long)out.ntrig;
		cs->n_reject += (unsigned long long)out.nrej;
		cs->n_burst += (unsigned long long)out.nburst;
		cs->n_defer += (unsigned long long)out.ndefer;
		cs->n_slow += (unsigned long long)(st.pos - p0);
		cs->n_redo += 1;
		atomicAdd(p.outc_total_redo, 1u);
		p.ctl[CTL_NSEL0 + sc] = 0;	/* K2d: nothing of the resolver's for this channel */
	}
}

/* ====================================================================== K2d
 * Payload decode of the bursts that lie on the real chain: one workgroup per
 * selected burst descriptor, one lane per byte.
 */
#define K2D_NT 256
#ifndef K2D_WAVES
#define K2D_WAVES 2
#endif
__global__ __launch_bounds__(K2D_NT) __attribute__((amdgpu_waves_per_eu(K2D_WAVES, 8)))
void k2d_payload(K2Params p)
{
	__shared__ unsigned s_slot;
	__shared__ float sph[VDL2_MAXSYM];
	const int sc = blockIdx.y;
	unsigned n = p.ctl[CTL_NSEL0 + sc];
	n = n > VDL2_SEL_CAP ? VDL2_SEL_CAP : n;
	const unsigned *sel = p.sel_list + (size_t)sc * VDL2_SEL_CAP;
	for (unsigned i = blockIdx.x; i < n; i += gridDim.x) {
		if (threadIdx.x == 0) {
			unsigned slot = atomicAdd(p.outc, 1u);
			if (slot >= p.rec_cap) {
				atomicAdd(p.outc + 1, 1u);
				slot = 0xffffffffu;
			}
			s_slot = slot;
		}
		__syncthreads();
		const unsigned slot = s_slot;
		if (slot != 0xffffffffu) {
			const BurstDesc d = p.stage[sel[i]];
			const int s = d.sc / VDL2_CS;
			const float2 *x0 = p.dec + (size_t)d.sc * p.cap - p.ss[s].dec_base;
			burst_payload<K2D_NT>(p.recs + slot, x0, p.pn, d.nstar, d.clk0, d.df, d.nbrow, d.nlbyte, s, p.cfg[d.sc], sph);
		}
		__syncthreads();
	}
}

/* ======================================================================= K3
 * Move the frames no channel has consumed yet (plus history) to the front of
 * the other ping-pong plane set and rebase stream time.  Normally ~170 frames
 * per plane; up to one full burst when a channel waits for the end of one.
 */
#define K3_THREADS 256
__global__ __launch_bounds__(K3_THREADS)
void k3_compact(K3Params p)
{
	const int s = blockIdx.y, c = blockIdx.x;
	const StreamState *ss = p.ss + s;
	long long mn = 0x7fffffffffffffffLL;
	for (int k = 0; k < p.nbch; ++k) {
		const long long q = p.cs[(size_t)s * VDL2_CS + k].pos;
		mn = q < mn ? q : mn;
	}
	const long long base = ss->dec_base;
	const long long end = base + ss->dec_fill + p.J;
	long long nb = mn - VDL2_HIST;
	if (nb > end - VDL2_HIST)
		nb = end - VDL2_HIST;	/* always keep the history */
	if (nb < base)
		nb = base;
	if (nb < end - VDL2_CARRY_FRAMES)
		nb = end - VDL2_CARRY_FRAMES;	/* cannot happen: no burst is that long */
	const long long keep = end - nb;
	/* the carry sits right-aligned below frame VDL2_CARRY_FRAMES of the other plane set, so that the
	 * channeliser of the next push -- which writes from that frame on -- does not depend on how
	 * much is carried and may run while this push is still being demodulated */
	const float2 *src = p.src + ((size_t)s * VDL2_CS + c) * p.cap + (nb - base);
	float2 *dst = p.dst + ((size_t)s * VDL2_CS + c) * p.cap + (VDL2_CARRY_FRAMES - keep);
	for (long long i = threadIdx.x; i < keep; i += K3_THREADS)
		dst[i] = src[i];
}

/* one launch instead of four memsets */
__global__ void k_push_init(KInitParams p)
{
	for (int i = threadIdx.x; i < p.ctl_words; i += blockDim.x)
		p.ctl[i] = 0u;
	for (int i = threadIdx.x; i < p.nsc; i += blockDim.x) {
		p.fail[i] = 0x7f7f7f7f;
		p.redo[i] = 0;
	}
	if (threadIdx.x < 2)
		p.outc[threadIdx.x] = 0u;
}

/* runs after k3_compact (same stream): publish the new time base and hand the push's counters to the host */
__global__ void k3_rebase(K3Params p)
{
	const int s = blockIdx.x;
	if (threadIdx.x != 0)
		return;
	if (s == 0) {
		p.host_cnt[0] = p.outc[2 * p.ring];
		p.host_cnt[1] = p.outc[2 * p.ring + 1];
		p.host_cnt[2] = p.outc[4];
		p.host_cnt[3] = p.outc[5];
	}
	StreamState *ss = p.ss + s;
	long long mn = 0x7fffffffffffffffLL;
	for (int k = 0; k < p.nbch; ++k) {
		const long long q = p.cs[(size_t)s * VDL2_CS + k].pos;
		mn = q < mn ? q : mn;
	}
	const long long base = ss->dec_base;
	const long long end = base + ss->dec_fill + p.J;
	long long nb = mn - VDL2_HIST;
	if (nb > end - VDL2_HIST)
		nb = end - VDL2_HIST;
	ss->dec_base = end - VDL2_CARRY_FRAMES;	/* frame VDL2_CARRY_FRAMES = first output of the next push */
	ss->dec_fill = VDL2_CARRY_FRAMES;
}

/* test hook: both device forms of atan2f; a disagreement between them comes back as NaN */
__global__ void k_atan2f(const float *y, const float *x, float *out, size_t n)
{
	__shared__ float atab[VDL2_ATAN_ROWS * VDL2_ATAN_STRIDE];
	if (threadIdx.x < VDL2_ATAN_ROWS * VDL2_ATAN_STRIDE)
		atab[threadIdx.x] = vdl2_atan_tab_entry(threadIdx.x);
	__syncthreads();
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) {
		const float a = vdl2_atan2f(y[i], x[i]), b = vdl2_atan2f_tab(y[i], x[i], atab);
		out[i] = (__float_as_uint(a) == __float_as_uint(b)) ? b : __uint_as_float(0x7fc00001u);
	}
}

#endif
