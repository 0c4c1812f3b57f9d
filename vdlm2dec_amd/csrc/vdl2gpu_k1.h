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
	const int s = (int)blockIdx.y;
	const float2 *lo = p.lo + (size_t)s * VDL2_CS * p.L;
	for (int idx = tid; idx < (p.L + p.maxwin) * VDL2_CS; idx += K1_THREADS) {
		const int n = idx >> 3, c = idx & 7;
		lo_s[idx] = lo[c * p.L + (n % p.L)];
	}
	const char *raw = (const char *)p.raw + (size_t)s * p.stream_stride;
	StreamState *ss = p.ss + s;
	const long long fill = VDL2_CARRY_FRAMES;	/* a push's output always starts at that frame */
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
		if (FMT == VDL2GPU_FMT_CU8 && p.quirk) {
			/* rtl.c:285-292 as written: within every hand-off block of 32768 samples, sample k lands in slot
			 * k + 1, slot 0 stays 0 (BSS) and the block's last sample is lost.  Pushes are whole
			 * blocks in this mode, so the position in the block is the position in the push modulo 32768. */
			for (int i = tid; i < cnt; i += K1_THREADS) {
				const long long gi = in_lo + i;
				xs[i] = (gi & 32767) ? k1_load<FMT>(raw, gi - 1) : make_float2(0.0f, 0.0f);
			}
		} else
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


/* ---- K1 fast path: whole periods of the schedule, any rate ----------------------------
 * The dump schedule and the LO phase repeat every PER = 4*SDRCLK inputs = 84 outputs (1 ms of air
 * time; 2000 inputs at 2 MS/s, 10000 at 10 MS/s), so 64 consecutive periods are 64 copies of the
 * same program: same window boundaries, same LO value at every step, different samples.  That is
 * the wavefront: LANE = PERIOD, WAVE = CHANNEL, eight waves (the stream's eight channels) to a
 * workgroup that shares the samples.
 *
 *   LO      wave-uniform, so it lives in SGPRs: one s_load_dwordx16 per 8 samples from the
 *           channel's table, and the mixer's packed multiplies take it as their scalar operand.
 *           No LO registers per lane, hence no limit on the window length (119/120 samples at
 *           10 MS/s cost what 23/24 do at 2 MS/s).
 *   x       a workgroup fetches a chunk of 32 samples of each of its 64 periods with ONE 16-byte
 *           load per thread (8 threads cover a period's 128 contiguous bytes), converts once
 *           (rtl.c:287-289 / SURVEY A.1) and parks float2 in LDS, row = period.  The next chunk is
 *           in flight in registers while this one is mixed.  Every lane reads its own row: one
 *           conflict-free ds_read_b64 per sample and wave, nothing to broadcast.
 *   mixer   per sample and channel the reference's eight roundings as four packed-FP32
 *           instructions (k1_cmac_s), accumulated in stream order: bit-identical to d8psk.c:368.
 *   out     a lane finishes a window of ITS period every 23/24 samples; eight of them are parked in
 *           a wave-private LDS tile [window][period] and leave as 64-byte runs of the channel plane
 *           (4 lanes x 16 bytes), instead of 8-byte scattered stores.
 *
 * A workgroup's task is (64 periods) x (a run of `wpt` windows of the period), so that a push is
 * several thousand tasks whatever its length.  All control flow is wave-uniform (scalar branches).
 * Gfx950 issues plain FP32 at 2 cycles and packed FP32 at 4 per wave64, so the mixer costs
 * 16 issue cycles per sample and channel either way: that, not HBM, bounds this kernel
 * (scripts/micro/valu_rate.hip measures 4.6-4.9 cycles per packed op with all SIMDs busy). */
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef const v16f __attribute__((address_space(4))) *k1_cptr16;	/* constant address space: scalar loads */

/* acc += x * w for complex x (VGPR pair), w (SGPR pair) with the reference's operation order
 *   pr = x.re*w.re - x.im*w.im;  pi = x.re*w.im + x.im*w.re;  acc += (pr, pi)
 * as four packed-FP32 VALU ops:
 *   a = (x.re*w.re, x.re*w.im)          v_pk_mul_f32, op_sel picks x.re twice
 *   b = (x.im*(-w.im), x.im*w.re)       v_pk_mul_f32, halves of w swapped, low lane negated
 *   acc += (a + b)                      2 x v_pk_add_f32
 * a.lo + b.lo = x.re*w.re + (-(x.im*w.im)) is bit-identical to the subtraction.  One volatile
 * block: the compiler must not turn the uniform branches around it into selects. */
__device__ __forceinline__ void k1_cmac_s(v2f &acc, v2f x, v2f w)
{
	v2f a, b;
	asm volatile("v_pk_mul_f32 %1, %3, %4 op_sel_hi:[0,1]\n\t"
		     "v_pk_mul_f32 %2, %3, %4 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
		     "s_nop 0\n\t"
		     "v_pk_add_f32 %1, %1, %2\n\t"
		     "s_nop 0\n\t"
		     "v_pk_add_f32 %0, %0, %1"
		     : "+v"(acc), "=&v"(a), "=&v"(b)
		     : "v"(x), "s"(w));
}

/* real input (air.c:206-208): D += x * wf is (x*w.re, x*w.im), no cross terms (d8psk.c:368 with a float Cbuff) */
__device__ __forceinline__ void k1_rmac_s(v2f &acc, float x, v2f w)
{
	v2f a;
	asm volatile("v_pk_mul_f32 %1, %2, %3 op_sel_hi:[0,1]\n\t"
		     "s_nop 0\n\t"
		     "v_pk_add_f32 %0, %0, %1"
		     : "+v"(acc), "=&v"(a)
		     : "v"((v2f){x, x}), "s"(w));
}

/* Eight samples in one block, software-pipelined so that no instruction reads what the one before it
 * wrote (gfx950 needs a wait state there for packed FP32, and a wave that spends it on s_nop gives the
 * slot away): A/B = the two products of a sample, T = their sum, S = acc += T, in the order
 *   A0 B0 A1 B1 T0 A2 B2 T1 S0 A3 B3 T2 S1 ... A7 B7 T6 S5 T7 S6 . S7
 * -- the S chain is the reference's accumulation order, sample by sample. */
#define K1_A(t, x, w) "v_pk_mul_f32 " t ", " x ", " w " op_sel_hi:[0,1]\n\t"
#define K1_B(t, x, w) "v_pk_mul_f32 " t ", " x ", " w " op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
#define K1_T(a, b) "v_pk_add_f32 " a ", " a ", " b "\n\t"
#define K1_S(a) "v_pk_add_f32 %0, %0, " a "\n\t"
__device__ __forceinline__ void k1_cmac8_s(v2f &acc, const v2f (&x)[8], const v16f w)
{
	v2f a0, b0, a1, b1, a2, b2;
	asm volatile(
		K1_A("%1", "%7", "%15") K1_B("%2", "%7", "%15")
		K1_A("%3", "%8", "%16") K1_B("%4", "%8", "%16")
		K1_T("%1", "%2")
		K1_A("%5", "%9", "%17") K1_B("%6", "%9", "%17")
		K1_T("%3", "%4")
		K1_S("%1")
		K1_A("%1", "%10", "%18") K1_B("%2", "%10", "%18")
		K1_T("%5", "%6")
		K1_S("%3")
		K1_A("%3", "%11", "%19") K1_B("%4", "%11", "%19")
		K1_T("%1", "%2")
		K1_S("%5")
		K1_A("%5", "%12", "%20") K1_B("%6", "%12", "%20")
		K1_T("%3", "%4")
		K1_S("%1")
		K1_A("%1", "%13", "%21") K1_B("%2", "%13", "%21")
		K1_T("%5", "%6")
		K1_S("%3")
		K1_A("%3", "%14", "%22") K1_B("%4", "%14", "%22")
		K1_T("%1", "%2")
		K1_S("%5")
		K1_T("%3", "%4")
		K1_S("%1")
		"s_nop 0\n\t"
		K1_S("%3")
		: "+v"(acc), "=&v"(a0), "=&v"(b0), "=&v"(a1), "=&v"(b1), "=&v"(a2), "=&v"(b2)
		: "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]),
		  "s"((v2f){w[0], w[1]}), "s"((v2f){w[2], w[3]}), "s"((v2f){w[4], w[5]}), "s"((v2f){w[6], w[7]}),
		  "s"((v2f){w[8], w[9]}), "s"((v2f){w[10], w[11]}), "s"((v2f){w[12], w[13]}), "s"((v2f){w[14], w[15]}));
}

/* What a block of 8 samples needs, requested in one go and waited for once: the 8 LO values of the wave's
 * channel (one s_load_dwordx16 into SGPRs) and 8 consecutive float2 of this lane's LDS row -- as eight
 * ds_read_b64: the LDS serves those at 256 bytes a clock, the ds_read2_b64 the compiler would merge them into
 * gets half of that, and at one read per sample and wave the LDS is nearly as busy as the VALU. */
__device__ __forceinline__ void k1_load_block(v16f &w, v2f (&x)[8], const float2 *lo, const unsigned a)
{
	asm volatile("s_load_dwordx16 %8, %10, 0x0\n\t"
		     "ds_read_b64 %0, %9\n\tds_read_b64 %1, %9 offset:8\n\tds_read_b64 %2, %9 offset:16\n\t"
		     "ds_read_b64 %3, %9 offset:24\n\tds_read_b64 %4, %9 offset:32\n\tds_read_b64 %5, %9 offset:40\n\t"
		     "ds_read_b64 %6, %9 offset:48\n\tds_read_b64 %7, %9 offset:56\n\ts_waitcnt lgkmcnt(0)"
		     : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7]), "=&s"(w)
		     : "v"(a), "s"(lo)
		     : "memory");
}

__device__ __forceinline__ void k1_load_block_nos(v16f &w, v2f (&x)[8], const float2 *lo, const unsigned a)
{
	asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:8\n\tds_read_b64 %2, %8 offset:16\n\t"
		     "ds_read_b64 %3, %8 offset:24\n\tds_read_b64 %4, %8 offset:32\n\tds_read_b64 %5, %8 offset:40\n\t"
		     "ds_read_b64 %6, %8 offset:48\n\tds_read_b64 %7, %8 offset:56\n\ts_waitcnt lgkmcnt(0)"
		     : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7])
		     : "v"(a)
		     : "memory");
	w = (v16f)(1.0f);
}
__device__ __forceinline__ void k1_load_block_nol(v16f &w, v2f (&x)[8], const float2 *lo, const unsigned a)
{
	asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(w) : "s"(lo) : "memory");
#pragma unroll
	for (int u = 0; u < 8; ++u)
		x[u] = (v2f){1.0f, 2.0f};
}
#define K1P_THREADS 512
#define K1P_CH 32		/* samples of every period per chunk */
#define K1P_XROW (K1P_CH + 1)	/* LDS row of a period: +1 keeps the lanes' reads on distinct banks */
#define K1P_OROW 65
#define K1P_PER_OUT 84

template <int FMT> struct K1Fmt;
template <> struct K1Fmt<VDL2GPU_FMT_CU8> { enum { BYTES = 2, SPB = 8 }; };	/* SPB: samples per 16-byte piece */
template <> struct K1Fmt<VDL2GPU_FMT_CS16> { enum { BYTES = 4, SPB = 4 }; };
template <> struct K1Fmt<VDL2GPU_FMT_CF32> { enum { BYTES = 8, SPB = 2 }; };
template <> struct K1Fmt<VDL2GPU_FMT_F32R> { enum { BYTES = 4, SPB = 4 }; };

/* one 16-byte piece of raw samples -> SPB converted samples */
template <int FMT> __device__ __forceinline__ void k1_piece_cvt(const uint4 v, float2 *out)
{
	const unsigned w[4] = {v.x, v.y, v.z, v.w};
	if constexpr (FMT == VDL2GPU_FMT_CU8) {
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			const unsigned h = (w[u >> 1] >> (16 * (u & 1))) & 0xffffu;
			out[u] = make_float2((float)(h & 0xffu) - (float)127.37, (float)(h >> 8) - (float)127.37);
		}
	} else if constexpr (FMT == VDL2GPU_FMT_CS16) {
#pragma unroll
		for (int u = 0; u < 4; ++u)
			out[u] = make_float2((float)(short)(w[u] & 0xffffu), (float)(short)(w[u] >> 16));
	} else if constexpr (FMT == VDL2GPU_FMT_CF32) {
		out[0] = make_float2(__uint_as_float(w[0]), __uint_as_float(w[1]));
		out[1] = make_float2(__uint_as_float(w[2]), __uint_as_float(w[3]));
	} else {
#pragma unroll
		for (int u = 0; u < 4; ++u)
			out[u] = make_float2(__uint_as_float(w[u]), 0.0f);
	}
}

template <int FMT> __global__ __launch_bounds__(K1P_THREADS, 6)
void k1_pp(K1PParams p)
{
	constexpr int B = K1Fmt<FMT>::BYTES, SPB = K1Fmt<FMT>::SPB;
#ifdef K1P_DBG
	const int dbg = p.dbg;	/* development switches (VDL2GPU_K1_DBG): 1 no mixing, 2 no loads after the first chunk, 4 no stores, .. */
#else
	constexpr int dbg = 0;	/* (as run-time tests they were seven branches in every block of 8 samples) */
#endif
	constexpr int NPIECE = K1P_CH / SPB;			/* 16-byte pieces per period and chunk */
	constexpr int NPT = (64 * NPIECE + K1P_THREADS - 1) / K1P_THREADS;	/* pieces per thread */
	__shared__ float2 xs[8 + 64 * K1P_XROW + 8];	/* 8 entries of pad on either side: blocks of 8 are read whole */
	__shared__ float2 os[8][8 * K1P_OROW];
	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int c = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int s = (int)blockIdx.y;
	const int blk = (int)(blockIdx.x / (unsigned)p.nsub), sub = (int)(blockIdx.x % (unsigned)p.nsub);
	const int k0 = sub * p.wpt, k1 = k0 + p.wpt;		/* this task's windows of the period */
	const int pb = blk * 64;				/* its first period, counted from per_lo */
	const int nper = (p.per_n - pb < 64) ? p.per_n - pb : 64;
	const long long fill = VDL2_CARRY_FRAMES;
	const bool active = c < p.nbch;
	const char *raw = (const char *)p.raw + (size_t)s * p.stream_stride + (size_t)p.sbase0 * B;	/* first sample of period per_lo */
	if (p.edge_state && blockIdx.x == 0 && tid < VDL2_CS) {	/* what k1_channelise leaves at a push's two ends (see k1_fast) */
		StreamState *ss = p.ss + s;
		if (tid == 0) {
			ss->last_fill = VDL2_CARRY_FRAMES;
			ss->last_J = p.J;
		}
		ss->acc[p.parity ^ 1][tid] = make_float2(0.0f, 0.0f);	/* the push ends on a window boundary: nothing carried */
	}

	/* loader role: NPT pieces (period lp, piece lj) */
	const char *lptr[NPT];
	int lcol[NPT];
	bool lval[NPT];
#pragma unroll
	for (int j = 0; j < NPT; ++j) {
		const int q = tid + j * K1P_THREADS;
		const int lp = q / NPIECE, lj = q % NPIECE;
		lval[j] = q < 64 * NPIECE;
		const int pp = lp < nper ? lp : nper - 1;	/* lanes beyond the last period re-read it (and store nothing) */
		lptr[j] = raw + ((long long)(pb + pp) * p.per_in - p.d) * B + lj * 16;
		lcol[j] = 8 + (lp < 64 ? lp : 63) * K1P_XROW + lj * SPB;
	}

	int k = k0;
	int i = (k0 == 0) ? 0 : p.wend[k0 - 1] + 1;		/* period-relative sample index */
	int wend = p.wend[k0];
	int nf = wend - i + 1;
	const int i_stop = p.wend[k1 - 1] + 1;
	int wi = (p.ph0 + i) % p.L;
	int slot = 0, kflush = k0;
	v2f acc = {0.0f, 0.0f};
	const float2 *lo = p.lo_ext + ((size_t)s * VDL2_CS + (active ? c : 0)) * p.lo_stride + 8;	/* 8 entries of front pad */
	const unsigned xrow = (unsigned)(size_t)(__attribute__((address_space(3))) const float2 *)&xs[8 + lane * K1P_XROW];
	const int m0 = (i + p.d) / K1P_CH, m1 = (i_stop - 1 + p.d) / K1P_CH;
	float2 *decp = p.dec + ((size_t)s * VDL2_CS + (active ? c : 0)) * p.cap + fill + (p.per_lo + pb) * K1P_PER_OUT;

	uint4 rr[NPT];
#pragma unroll
	for (int j = 0; j < NPT; ++j)
		rr[j] = *reinterpret_cast<const uint4 *>(lptr[j] + (long long)m0 * K1P_CH * B);
	for (int m = m0; m <= m1; ++m) {
		if (!(dbg & 8))
			__syncthreads();	/* the previous chunk has been read by every wave */
#pragma unroll
		for (int j = 0; j < NPT; ++j)
			if (lval[j]) {
				float2 cv[SPB];
				k1_piece_cvt<FMT>(rr[j], cv);
#pragma unroll
				for (int u = 0; u < SPB; ++u)
					xs[lcol[j] + u] = cv[u];
			}
		if (m < m1 && !(dbg & 2)) {
#pragma unroll
			for (int j = 0; j < NPT; ++j)
				rr[j] = *reinterpret_cast<const uint4 *>(lptr[j] + (long long)(m + 1) * K1P_CH * B);
		}
		if (!(dbg & 8))
			__syncthreads();
		if (!active || (dbg & 1))
			continue;
		const int cb = m * K1P_CH - p.d;	/* period-relative index of the chunk's first sample */
		const int hi = (i_stop < cb + K1P_CH) ? i_stop : cb + K1P_CH;
		while (i < hi) {
			/* a piece: the samples up to the window's or the chunk's end, as blocks of 8 and a tail */
			const int lim = (hi < wend + 1) ? hi : wend + 1;
			int n = lim - i;
			unsigned xa = xrow + (unsigned)(i - cb) * 8u;
			const float2 *lp = lo + wi;
			i = lim;
			wi += n;
			if (wi >= p.L)
				wi -= p.L;
			for (; n >= 8; n -= 8) {
				v16f w;
				v2f xr[8];
				if (dbg & 16) {
					w = (v16f)(1.0f);
#pragma unroll
					for (int u = 0; u < 8; ++u)
						xr[u] = acc;
				} else if (dbg & 512)
					k1_load_block_nos(w, xr, lp, xa);
				else if (dbg & 1024)
					k1_load_block_nol(w, xr, lp, xa);
				else
					k1_load_block(w, xr, lp, xa);
				if constexpr (FMT == VDL2GPU_FMT_F32R) {
#pragma unroll
					for (int u = 0; u < 8; ++u)
						k1_rmac_s(acc, xr[u].x, (v2f){w[2 * u], w[2 * u + 1]});
				} else if (!(dbg & 64))
					k1_cmac8_s(acc, xr, w);
				lp += 8;
				xa += 64;
			}
			if (n && !(dbg & 128)) {
				/* the tail: read the 8 entries that END with it (what lies before is the row's or the table's
				 * front pad or earlier samples) and enter the unrolled sequence n steps before its end */
				v16f w;
				v2f xr[8];
				if (dbg & 512)
					k1_load_block_nos(w, xr, lp - (8 - n), xa - (unsigned)(8 - n) * 8u);
				else if (dbg & 1024)
					k1_load_block_nol(w, xr, lp - (8 - n), xa - (unsigned)(8 - n) * 8u);
				else
					k1_load_block(w, xr, lp - (8 - n), xa - (unsigned)(8 - n) * 8u);
#define K1_TAIL(u) if constexpr (FMT == VDL2GPU_FMT_F32R) k1_rmac_s(acc, xr[u].x, (v2f){w[2 * (u)], w[2 * (u) + 1]}); \
		   else k1_cmac_s(acc, xr[u], (v2f){w[2 * (u)], w[2 * (u) + 1]});
				switch (n) {
				case 7: K1_TAIL(1)
				case 6: K1_TAIL(2)
				case 5: K1_TAIL(3)
				case 4: K1_TAIL(4)
				case 3: K1_TAIL(5)
				case 2: K1_TAIL(6)
				default: K1_TAIL(7)
				}
#undef K1_TAIL
			}
			if (i > wend && (dbg & 256)) {
				acc = (v2f){0.0f, 0.0f};
				++k;
				if (k < k1) {
					const int e = p.wend[k];
					nf = e - wend;
					wend = e;
				}
			} else if (i > wend) {
				/* D /= nf (d8psk.c:377).  q0 = x*RN(1/nf); q = fma(fma(-q0, nf, x), RN(1/nf), q0) is the
				 * correctly rounded quotient for every |x| >= 1e-30 and nf in {23,24,59,60,71,72,119,120}
				 * (exhaustively checked: tests/ctests/div_check.c); otherwise the plain IEEE division */
				const float fn = (float)nf;
				float qr, qi;
				if (p.fast_div && __all(fabsf(acc.x) >= 1e-30f && fabsf(acc.y) >= 1e-30f)) {
					const float rfn = (nf == p.nf_lo) ? p.rcp_lo : p.rcp_hi;
					const float q0r = acc.x * rfn, q0i = acc.y * rfn;
					qr = fmaf(fmaf(-q0r, fn, acc.x), rfn, q0r);
					qi = fmaf(fmaf(-q0i, fn, acc.y), rfn, q0i);
				} else {
					qr = acc.x / fn;
					qi = acc.y / fn;
				}
				os[c][slot * K1P_OROW + lane] = make_float2(qr, qi);
				acc = (v2f){0.0f, 0.0f};
				++slot;
				++k;
				if (slot == 8 || k == k1) {
					/* 8 windows x 64 periods -> 64-byte runs of the plane: lane = (period, pair of windows) */
					__builtin_amdgcn_wave_barrier();
#pragma unroll
					for (int it = 0; it < 4; ++it) {
						const int pp = it * 16 + (lane >> 2), q = lane & 3;
						if (pp < nper && 2 * q < slot && !(dbg & 4)) {
							const float2 v0 = os[c][(2 * q) * K1P_OROW + pp];
							float2 *dst = decp + (long long)pp * K1P_PER_OUT + kflush + 2 * q;
							if (2 * q + 1 < slot) {
								const float2 v1 = os[c][(2 * q + 1) * K1P_OROW + pp];
								typedef float k1_v4a8 __attribute__((ext_vector_type(4), aligned(8)));
								*reinterpret_cast<k1_v4a8 *>(dst) = (k1_v4a8){v0.x, v0.y, v1.x, v1.y};
							} else
								*dst = v0;
						}
					}
					__builtin_amdgcn_wave_barrier();
					kflush = k;
					slot = 0;
				}
				if (k < k1) {
					const int e = p.wend[k];
					nf = e - wend;
					wend = e;
				}
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

/* Eight (seven) samples with the LO values in VGPRs, ordered so that no instruction reads what the one before it wrote
 * (A/B = the two products of a sample, T = their sum, S = acc += T in sample order; see k1_cmac8_s). */
__device__ __forceinline__ void k1_cmac8_v(v2f &acc, const v2f (&x)[8], const v2f *w)
{
	v2f a0, b0, a1, b1, a2, b2;
	asm volatile(
		K1_A("%1", "%7", "%15") K1_B("%2", "%7", "%15")
		K1_A("%3", "%8", "%16") K1_B("%4", "%8", "%16")
		K1_T("%1", "%2")
		K1_A("%5", "%9", "%17") K1_B("%6", "%9", "%17")
		K1_T("%3", "%4")
		K1_S("%1")
		K1_A("%1", "%10", "%18") K1_B("%2", "%10", "%18")
		K1_T("%5", "%6")
		K1_S("%3")
		K1_A("%3", "%11", "%19") K1_B("%4", "%11", "%19")
		K1_T("%1", "%2")
		K1_S("%5")
		K1_A("%5", "%12", "%20") K1_B("%6", "%12", "%20")
		K1_T("%3", "%4")
		K1_S("%1")
		K1_A("%1", "%13", "%21") K1_B("%2", "%13", "%21")
		K1_T("%5", "%6")
		K1_S("%3")
		K1_A("%3", "%14", "%22") K1_B("%4", "%14", "%22")
		K1_T("%1", "%2")
		K1_S("%5")
		K1_T("%3", "%4")
		K1_S("%1")
		"s_nop 0\n\t"
		K1_S("%3")
		: "+v"(acc), "=&v"(a0), "=&v"(b0), "=&v"(a1), "=&v"(b1), "=&v"(a2), "=&v"(b2)
		: "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]),
		  "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]));
}
__device__ __forceinline__ void k1_cmac7_v(v2f &acc, const v2f (&x)[8], const v2f *w)
{
	v2f a0, b0, a1, b1, a2, b2;
	asm volatile(
		K1_A("%1", "%7", "%14") K1_B("%2", "%7", "%14")
		K1_A("%3", "%8", "%15") K1_B("%4", "%8", "%15")
		K1_T("%1", "%2")
		K1_A("%5", "%9", "%16") K1_B("%6", "%9", "%16")
		K1_T("%3", "%4")
		K1_S("%1")
		K1_A("%1", "%10", "%17") K1_B("%2", "%10", "%17")
		K1_T("%5", "%6")
		K1_S("%3")
		K1_A("%3", "%11", "%18") K1_B("%4", "%11", "%18")
		K1_T("%1", "%2")
		K1_S("%5")
		K1_A("%5", "%12", "%19") K1_B("%6", "%12", "%19")
		K1_T("%3", "%4")
		K1_S("%1")
		K1_A("%1", "%13", "%20") K1_B("%2", "%13", "%20")
		K1_T("%5", "%6")
		K1_S("%3")
		K1_T("%1", "%2")
		K1_S("%5")
		"s_nop 0\n\t"
		K1_S("%1")
		: "+v"(acc), "=&v"(a0), "=&v"(b0), "=&v"(a1), "=&v"(b1), "=&v"(a2), "=&v"(b2)
		: "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]),
		  "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]));
}
__device__ __forceinline__ void k1_cmac1_v(v2f &acc, v2f x, v2f w)
{
	v2f a, b;
	asm volatile(K1_A("%1", "%3", "%4") K1_B("%2", "%3", "%4") "s_nop 0\n\t" K1_T("%1", "%2") "s_nop 0\n\t" K1_S("%1")
		     : "+v"(acc), "=&v"(a), "=&v"(b)
		     : "v"(x), "v"(w));
}
/* eight consecutive float2 from LDS, requested and not waited for (ds_read_b64: 256 bytes a clock; the ds_read2_b64 the
 * compiler merges neighbouring reads into gets half of that) */
__device__ __forceinline__ void k1_lds_issue8(v2f (&x)[8], const unsigned a)
{
	asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:8\n\tds_read_b64 %2, %8 offset:16\n\t"
		     "ds_read_b64 %3, %8 offset:24\n\tds_read_b64 %4, %8 offset:32\n\tds_read_b64 %5, %8 offset:40\n\t"
		     "ds_read_b64 %6, %8 offset:48\n\tds_read_b64 %7, %8 offset:56"
		     : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7])
		     : "v"(a)
		     : "memory");
}

/* four consecutive float2 from LDS at byte offset OFS behind address a, requested and not waited for (the offset is an
 * immediate: as part of the address the compiler keeps one register per block and copy of the slice) */
template <int OFS> __device__ __forceinline__ void k1_lds_issue4(v2f (&x)[4], const unsigned a)
{
	asm volatile("ds_read_b64 %0, %4 offset:%5\n\tds_read_b64 %1, %4 offset:%6\n\tds_read_b64 %2, %4 offset:%7\n\tds_read_b64 %3, %4 offset:%8"
		     : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3])
		     : "v"(a), "n"(OFS), "n"(OFS + 8), "n"(OFS + 16), "n"(OFS + 24)
		     : "memory");
}
/* Four (three) samples with the LO values in VGPRs and two pairs of temporaries; no instruction reads what the one
 * before it wrote (the packed operations' forwarding hazard), one s_nop where that cannot be arranged. */
__device__ __forceinline__ void k1_cmac4_v(v2f &acc, const v2f (&x)[4], const v2f *w)
{
	v2f a0, b0, a1, b1;
	asm volatile(
		K1_A("%1", "%5", "%9") K1_B("%2", "%5", "%9")
		K1_A("%3", "%6", "%10") K1_B("%4", "%6", "%10")
		K1_T("%1", "%2")
		K1_T("%3", "%4")
		K1_S("%1")
		K1_A("%1", "%7", "%11")
		K1_S("%3")
		K1_B("%2", "%7", "%11")
		K1_A("%3", "%8", "%12") K1_B("%4", "%8", "%12")
		K1_T("%1", "%2")
		K1_T("%3", "%4")
		K1_S("%1")
		"s_nop 0\n\t"
		K1_S("%3")
		: "+v"(acc), "=&v"(a0), "=&v"(b0), "=&v"(a1), "=&v"(b1)
		: "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]));
}
__device__ __forceinline__ void k1_cmac3_v(v2f &acc, const v2f (&x)[4], const v2f *w)
{
	v2f a0, b0, a1, b1;
	asm volatile(
		K1_A("%1", "%5", "%8") K1_B("%2", "%5", "%8")
		K1_A("%3", "%6", "%9") K1_B("%4", "%6", "%9")
		K1_T("%1", "%2")
		K1_T("%3", "%4")
		K1_S("%1")
		K1_A("%1", "%7", "%10")
		K1_S("%3")
		K1_B("%2", "%7", "%10")
		"s_nop 0\n\t"
		K1_T("%1", "%2")
		"s_nop 0\n\t"
		K1_S("%1")
		: "+v"(acc), "=&v"(a0), "=&v"(b0), "=&v"(a1), "=&v"(b1)
		: "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(w[0]), "v"(w[1]), "v"(w[2]));
}

#define K1F_THREADS 128		/* two wavefronts: channels 0-3 and 4-7 of the same 16 windows */
#ifdef K1F_PROF
#define K1F_WAVES_OF(FMT_) 4	/* the stamps need registers of their own */
#else
#define K1F_WAVES_OF(FMT_) ((FMT_) == VDL2GPU_FMT_CF32 ? 4 : 5)
#endif	/* wavefronts per SIMD the kernel is built for: 96 registers (cf32 holds its
								 * raw samples in twice as many: 128, 4 wavefronts) */
#ifndef K1F_DEPTH
#define K1F_DEPTH 2		/* superperiods of raw samples in flight per wavefront (registers) */
#endif
#define K1F_CHUNK 8		/* superperiods per ticket (even: the two LDS copies and register sets alternate) */
#define K1F_PER_IN 8000		/* a SUPERPERIOD: 4 periods of the schedule = 336 outputs = 21 lines of 16 */
#define K1F_PER_OUT 336
#define K1F_ROLES 21

/* raw samples as the wave's loads deliver them: one 32-bit register per sample (64 for cf32) */
template <int FMT> struct K1Raw { typedef unsigned T; };
template <> struct K1Raw<VDL2GPU_FMT_CF32> { typedef unsigned T __attribute__((ext_vector_type(2))); };

/* The sample loads are written in assembly because their waits are: a wave keeps DEPTH periods of samples in
 * flight, and before it converts one it must only wait until the loads of THAT period have landed -- memory
 * operations of a wave complete in issue order, so "at most as many outstanding as were issued after them".
 * The compiler, seeing loads in a loop with a conditional body, waits for everything (vmcnt(0)): every period
 * then costs a full memory round trip, store acknowledgement included, and the kernel is latency-bound. */
template <int FMT, int OFS = 0> __device__ __forceinline__ void k1_raw_issue(typename K1Raw<FMT>::T &r, const unsigned voff, const char *sbase)
{
#ifdef K1F_LOAD_NT
#define K1F_LD_MOD " nt"
#else
#define K1F_LD_MOD ""
#endif
	if constexpr (FMT == VDL2GPU_FMT_CU8)
		asm volatile("global_load_ushort %0, %1, %2 offset:%3" K1F_LD_MOD : "=v"(r) : "v"(voff), "s"(sbase), "n"(OFS) : "memory");
	else if constexpr (FMT == VDL2GPU_FMT_CF32)
		asm volatile("global_load_dwordx2 %0, %1, %2 offset:%3" K1F_LD_MOD : "=v"(r) : "v"(voff), "s"(sbase), "n"(OFS) : "memory");
	else
		asm volatile("global_load_dword %0, %1, %2 offset:%3" K1F_LD_MOD : "=v"(r) : "v"(voff), "s"(sbase), "n"(OFS) : "memory");
}

template <int FMT> __device__ __forceinline__ float2 k1_raw_cvt(typename K1Raw<FMT>::T v)
{
	if constexpr (FMT == VDL2GPU_FMT_CU8) {
		return make_float2((float)(v & 0xffu) - (float)127.37, (float)((v >> 8) & 0xffu) - (float)127.37);
	} else if constexpr (FMT == VDL2GPU_FMT_CS16) {
		return make_float2((float)(short)(v & 0xffffu), (float)(short)(v >> 16));
	} else if constexpr (FMT == VDL2GPU_FMT_CF32) {
		return make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
	} else {
		return make_float2(__uint_as_float(v), 0.0f);
	}
}

/* one global_store_dwordx2 that is always ISSUED (the waits count it), with only the `on` lanes enabled; the address is
 * a workgroup-uniform base (scalar registers) plus the lane's 32-bit byte offset */
__device__ __forceinline__ void k1_store_masked(const float2 *sbase, unsigned voff, v2f v, bool on)
{
	const unsigned long long m = __ballot(on);
	asm volatile("s_mov_b64 s[2:3], exec\n\t"
		     "s_mov_b64 exec, %3\n\t"
		     "global_store_dwordx2 %0, %1, %2\n\t"
		     "s_mov_b64 exec, s[2:3]"
		     :: "v"(voff), "v"(v), "s"(sbase), "s"(m) : "memory", "s2", "s3");
}

#ifdef K1F_PROF
#define K1F_PROF_SLOTS 32768
__device__ unsigned k1f_prof[K1F_PROF_SLOTS][12];	/* development: shader cycles a wavefront spends in each phase (last launch) */
#define K1F_STAMP(I_) do { const unsigned t_ = (unsigned)__builtin_amdgcn_readfirstlane((int)clock64()); pf[I_] += t_ - tl; tl = t_; } while (0)
#else
#define K1F_STAMP(I_) do { } while (0)
#endif
template <int FMT> __global__ __launch_bounds__(K1F_THREADS, K1F_WAVES_OF(FMT))
void k1_fast(K1Params p)
{
	typedef typename K1Raw<FMT>::T raw_t;
	constexpr int B = (FMT == VDL2GPU_FMT_CU8) ? 2 : (FMT == VDL2GPU_FMT_CF32) ? 8 : 4;
#ifdef K1F_PROF
	unsigned pf[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
	unsigned tl = (unsigned)__builtin_amdgcn_readfirstlane((int)clock64());
	const unsigned wall0 = (unsigned)__builtin_amdgcn_readfirstlane((int)wall_clock64());
#endif
	/* LDS: every window has its own row of 25 float2 (24 samples + 1 of padding: rows of 50 dwords put the 8 windows of
	 * a half-wave read on 8 different bank pairs; laid end to end, windows 4 apart -- 95 or 96 samples -- shared banks),
	 * a pair of slices per iteration, two copies of the pair used in turn (one barrier per iteration) */
	__shared__ float2 xs[2][2][16 * 25 + 8];	/* [copy][half of the pair] */
	__shared__ int s_next;
	const int tid = threadIdx.x;
	const int lane = tid & 63, wv = tid >> 6;
	const int s = (int)blockIdx.y;
	/* A workgroup owns 16 consecutive outputs -- ONE 128-byte line of every channel plane -- of a superperiod (4
	 * periods of the schedule: 8000 inputs, 336 outputs, 21 lines) for many superperiods: lane = (window, channel),
	 * 16 windows x 4 channels to a wavefront, the two wavefronts share the windows' ~381 samples through LDS.  A
	 * wavefront's store is four whole, aligned lines.  (Runs of 64 bytes -- 8 windows per wavefront -- reached HBM as
	 * partial lines once the read stream pushed them out of the L2 before their other halves arrived: the same
	 * traffic moved in 128 us instead of 86, scripts/micro/store_shape.hip.)
	 *
	 * The kernel is built around what a SIMD needs to stay busy: one wavefront issues a packed operation every 9 cycles
	 * at best, four of them together one every 3.5, eight one every 1.5 - 2 (scripts/micro/clock_rate.hip, valu_rate.hip)
	 * -- three to four wavefronts per SIMD must be mixing at any time.  Hence 96 registers (5 wavefronts per SIMD: the
	 * 24 LO values of the lane's window take 48 of them, samples come from LDS four at a time), and a grid that is
	 * resident as a whole (the launch sizes it): a workgroup's start-up -- cold code, LO values, first samples -- is
	 * paid once per ~70 superperiods. */
	/* Work is handed out in TICKETS of K1F_CHUNK superperiods.  Workgroup b runs on XCD x = b % 8 and has role
	 * g = (b / 8) % 21; the workgroups of one (role, XCD) form a family that shares a counter and takes the superperiods
	 * per_lo + x + 8 i, i = 0, 1, .. in chunks: ticket t = i in [t C, t C + C).  The first ticket of a workgroup is its
	 * rank in the family, the next one comes from the counter while the current one is being worked on -- a workgroup on
	 * a SIMD that advances slowly (more wavefronts, a busier CU, another kernel's wavefronts beside it) simply takes
	 * fewer tickets.  With a fixed share each the launch lasted as long as its slowest SIMD: 104 us for wavefronts that
	 * lived 88 us on average.  A family's superperiods are neighbours of the other roles' on the same XCD: at any time the
	 * grid reads one contiguous band of the input and writes one contiguous band of each plane, each L2 its own eighth. */
	const int x = (int)(blockIdx.x & 7);
	const int g = (int)((blockIdx.x >> 3) % K1F_ROLES);
	const int rank = (int)(blockIdx.x / (8 * K1F_ROLES)), nfam = (int)(gridDim.x / (8 * K1F_ROLES));
	const int n_x = ((int)p.per_n - x + 7) >> 3;			/* superperiods of this XCD */
	if (rank * K1F_CHUNK >= n_x)	/* its first ticket is empty */
		return;
	if (p.edge_state && blockIdx.x == 0 && tid < VDL2_CS) {	/* (rank 0 of XCD 0: never empty) what k1_channelise leaves at a push's two ends */
		StreamState *ss = p.ss + s;
		if (tid == 0) {
			ss->last_fill = VDL2_CARRY_FRAMES;
			ss->last_J = p.J;
		}
		ss->acc[p.parity ^ 1][tid] = make_float2(0.0f, 0.0f);	/* the push ends on a window boundary: nothing carried */
	}
	const unsigned *ctr = p.tickets + ((size_t)s * K1F_ROLES + g) * 8 + x;	/* ticket = nfam + (old value - tbase[x]) */
	const unsigned tbase = p.tbase[x];
	const int kk = lane >> 2, c = wv * 4 + (lane & 3);
	const bool active = c < p.nbch;
	const char *raw = (const char *)p.raw + (size_t)s * p.stream_stride;
	/* The schedule repeats exactly every superperiod (336 * SDRCLK = 21 * 8000): window jr of ANY superperiod ends
	 * e(jr) samples behind the superperiod's nominal start pp * 8000, e(jr) = ceil(((jr + 1) * 500 - c0) / 21) - 1
	 * (k1_win_end with the superperiod's 168000 taken out; 21 * 32 keeps the division's numerator positive), and the
	 * sample at `rel` belongs to window ceil((21 (rel + 1) + c0 - 20) / 500) - 1.  Everything in front of the loop is
	 * 32-bit arithmetic on these two, no table and no barrier: every instruction here is executed exactly once and
	 * fetched cold (~330 cycles per 64-byte line of code), so this part is written for size.
	 * The slice of this workgroup: from the first sample of window 16g to the last of window 16g + 15. */
	const int c0 = p.c0;
	auto e_rel = [c0](int jr) { return ((jr + 1) * 500 - c0 + 20 + 21 * 32) / 21 - 32 - 1; };
	const int e0 = e_rel(g * 16 - 1);
	const int slen = e_rel(g * 16 + 15) - e0;
	const int ek = e_rel(g * 16 + kk - 1);
	const int off = ek - e0, nwin = e_rel(g * 16 + kk) - ek;
	/* threads fetch samples tid, tid+128, tid+256 of the slice (clamped: the tail re-reads the last sample) and park
	 * each in the row of the window it belongs to */
	unsigned vo[3];	/* [1] = [0] + 128 B is never clamped: the loads use [0] with an immediate offset */
	int xd[3];
#pragma unroll 1
	for (int u = 0; u < 3; ++u) {
		int i = tid + u * K1F_THREADS;
		i = i < slen ? i : slen - 1;
		const int rel = e0 + 1 + i;
		const int jr = (21 * (rel + 1) + c0 - 20 + 499) / 500 - 1;	/* numerator > 0 for every sample of the slice */
		const int x = (jr - g * 16) * 25 + (rel - e_rel(jr - 1) - 1);
		if (u == 0) { vo[0] = (unsigned)i * B; xd[0] = x; }
		else if (u == 1) { vo[1] = (unsigned)i * B; xd[1] = x; }
		else { vo[2] = (unsigned)i * B; xd[2] = x; }
	}
	/* index i of the family: superperiod per_lo + x + 8 i.  An ITERATION takes a pair (2 p, 2 p + 1): two slices in
	 * flight (one register set each), one wait, one barrier and one turn of the bookkeeping for two superperiods of
	 * mixing -- with one superperiod per iteration a wavefront spent as long outside the mixer as in it, and a SIMD
	 * needs three of its five mixing. */
	constexpr int CP = K1F_CHUNK / 2;	/* pairs per ticket */
	static_assert(K1F_DEPTH == 2 && K1F_CHUNK % 2 == 0 && CP >= 4, "the loop below is written for pairs and a ticket known at the fourth pair");
	const char *rbase = raw + ((p.per_lo + x) * K1F_PER_IN + e0 + 1) * B;	/* the slice in the family's first superperiod; workgroup-uniform */
	constexpr long long pbytes = (long long)K1F_PER_IN * B * 8;
	int p0 = rank * CP;	/* this iteration's pair */
	raw_t rr[2][3];
	{
		const int ia = 2 * p0, ib = ia + 1 < n_x ? ia + 1 : ia;
		const char *rb = rbase + pbytes * ia;
		k1_raw_issue<FMT>(rr[0][0], vo[0], rb);
		k1_raw_issue<FMT, K1F_THREADS * B>(rr[0][1], vo[0], rb);
		k1_raw_issue<FMT>(rr[0][2], vo[2], rb);
		rb = rbase + pbytes * ib;
		k1_raw_issue<FMT>(rr[1][0], vo[0], rb);
		k1_raw_issue<FMT, K1F_THREADS * B>(rr[1][1], vo[0], rb);
		k1_raw_issue<FMT>(rr[1][2], vo[2], rb);
	}
	K1F_STAMP(8);	/* prologue: addresses, first loads issued */
	/* the lane's LO values, behind the first samples' loads (one round trip for both); the table carries its own
	 * wrap-around (24 loads off one address) */
	v2f w[24];
	{
		const int ph = (p.no0 + e0 + 1 + off + 80) % 80;	/* 8000 = 100 LO periods: the same in every superperiod; e0 + 1 >= -23 */
		const float2 *lo = p.lo_ext + ((size_t)s * VDL2_CS + (active ? c : 0)) * p.lo_stride + 8 + ph;
#pragma unroll
		for (int t = 0; t < 24; ++t) {
			const float2 q = lo[t];
			w[t] = (v2f){q.x, q.y};
		}
	}
	const float fn = (float)nwin;
	const float rfn = 1.0f / fn;	/* RN(1/nf) for the exact FMA division below */
	const float2 *dec = p.dec + (size_t)s * VDL2_CS * p.cap + VDL2_CARRY_FRAMES + (p.per_lo + x) * K1F_PER_OUT + g * 16;	/* workgroup-uniform */
	const unsigned dvo = (unsigned)(((size_t)(active ? c : 0) * p.cap + kk) * sizeof(float2));	/* planes are < 4 GB apart */
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");	/* from here on the only memory operations are the counted ones below */
#pragma unroll
	for (int t = 0; t < 24; ++t)
		asm volatile("" : "+v"(w[t]));	/* loaded in front of the loop, once */
	K1F_STAMP(0);	/* prologue */
	const unsigned xa = (unsigned)(size_t)(__attribute__((address_space(3))) const float2 *)&xs[0][0][kk * 25];
	unsigned tkr = 0;	/* lane 0 of wavefront 0: the counter's answer, landing while the chunk is worked on */
	int nxt = 0x7fffffff;
	int pos = 0;	/* position in the current chunk */
	int buf = 0;	/* which copy of the slices this iteration writes and reads */
#ifdef K1F_PROF
	int nit = 0;
#endif
	while (p0 >= 0) {
#ifdef K1F_PROF
		nit += 2;
#endif
		/* pair p0: registers -> float -> LDS slices, then refill the registers with the next pair.  Every iteration
		 * issues exactly 6 loads and then 2 stores per wavefront: only the 2 stores have been issued after the loads
		 * this iteration waits for (a ticket request is issued BEFORE an iteration's loads, so they see it land). */
#ifndef K1F_NOPRIO
		/* the SIMD's arbiter serves its oldest wavefront first: left alone, the five wavefronts of a SIMD advance
		 * at very different rates.  Rotating priorities keep them together. */
		switch ((pos + (int)blockIdx.x) & 3) {
		case 0: __builtin_amdgcn_s_setprio(0); break;
		case 1: __builtin_amdgcn_s_setprio(1); break;
		case 2: __builtin_amdgcn_s_setprio(2); break;
		default: __builtin_amdgcn_s_setprio(3); break;
		}
#endif
#if defined(K1F_NOLOAD) || defined(K1F_NOSTORE)
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
		asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
#endif
		K1F_STAMP(1);	/* wait for the samples */
#pragma unroll
		for (int ab = 0; ab < 2; ++ab)
#pragma unroll
			for (int u = 0; u < 3; ++u)
				asm volatile("" : "+v"(rr[ab][u]));	/* read only behind the wait */
#pragma unroll
		for (int ab = 0; ab < 2; ++ab) {
			float2 *xb = xs[buf][ab];
#pragma unroll
			for (int u = 0; u < 3; ++u)
				xb[xd[u]] = k1_raw_cvt<FMT>(rr[ab][u]);
		}
		if (pos == 0) {
			/* ask for the next ticket: older than the loads issued below, so the next iteration's wait sees it land */
			const unsigned long long m = __ballot(tid == 0);
			asm volatile("s_mov_b64 s[2:3], exec\n\t"
				     "s_mov_b64 exec, %3\n\t"
				     "global_atomic_add %0, %1, %2, %4 sc0\n\t"
				     "s_mov_b64 exec, s[2:3]"
				     : "+v"(tkr) : "v"(0u), "v"(1u), "s"(m), "s"(ctr) : "memory", "s2", "s3");
		}
		if (pos == 1) {
			asm volatile("" : "+v"(tkr));	/* it has landed: the wait above was for loads issued after the request */
			if (tid == 0)
				s_next = ((int)(tkr - tbase) + nfam) * CP;
		}
		if (pos == 2)
			nxt = __builtin_amdgcn_readfirstlane(s_next);	/* first pair of the next ticket; written one barrier ago */
		/* the next pair: in this chunk, the next ticket's first, or none (the loads then fetch this one again) */
		int p1 = pos < CP - 1 ? p0 + 1 : nxt;
		p1 = 2 * p1 < n_x ? p1 : -1;
		{
			const int ia = 2 * (p1 >= 0 ? p1 : p0), ib = ia + 1 < n_x ? ia + 1 : ia;
#ifndef K1F_NOLOAD
			const char *rb = rbase + pbytes * ia;
			k1_raw_issue<FMT>(rr[0][0], vo[0], rb);
			k1_raw_issue<FMT, K1F_THREADS * B>(rr[0][1], vo[0], rb);
			k1_raw_issue<FMT>(rr[0][2], vo[2], rb);
			rb = rbase + pbytes * ib;
			k1_raw_issue<FMT>(rr[1][0], vo[0], rb);
			k1_raw_issue<FMT, K1F_THREADS * B>(rr[1][1], vo[0], rb);
			k1_raw_issue<FMT>(rr[1][2], vo[2], rb);
#endif
		}
		K1F_STAMP(2);	/* convert, park, issue the next loads */
#ifndef K1F_NOBARRIER
		__syncthreads();	/* the slices are written */
#endif
		K1F_STAMP(3);	/* barrier */
		const bool has_b = 2 * p0 + 1 < n_x;
		const unsigned xc = xa + (unsigned)buf * (unsigned)sizeof(xs[0]);
#pragma unroll
		for (int ab = 0; ab < 2; ++ab) {
			v2f res = {0.0f, 0.0f};
#ifdef K1F_NOMIX
			if (p.nbch > 8) {
#else
			if (ab == 0 || has_b) {
#endif
				v2f acc = {0.0f, 0.0f};
				if (FMT == VDL2GPU_FMT_F32R) {
					const v2f *xp = reinterpret_cast<const v2f *>(&xs[buf][ab][kk * 25]);
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
					/* six blocks of 4 samples; every block is mixed while the next one's samples are on their way
					 * from LDS (reads return in order: at most 4 outstanding = the previous block is there); which
					 * slice of the pair is part of the reads' immediate offsets */
					auto mix = [&](auto par) {
						constexpr int XO = (int)sizeof(xs[0][0]) * decltype(par)::value;
						v2f x0[4], x1[4];
						asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
						k1_lds_issue4<0 + XO>(x0, xc);
						k1_lds_issue4<32 + XO>(x1, xc);
						asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
						k1_cmac4_v(acc, x0, &w[0]);
						k1_lds_issue4<64 + XO>(x0, xc);
						asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
						k1_cmac4_v(acc, x1, &w[4]);
						k1_lds_issue4<96 + XO>(x1, xc);
						asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
						k1_cmac4_v(acc, x0, &w[8]);
						k1_lds_issue4<128 + XO>(x0, xc);
						asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
						k1_cmac4_v(acc, x1, &w[12]);
						k1_lds_issue4<160 + XO>(x1, xc);
						asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
						k1_cmac4_v(acc, x0, &w[16]);
						asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
						k1_cmac3_v(acc, x1, &w[20]);
						if (nwin == 24)
							k1_cmac1_v(acc, x1[3], w[23]);
					};
					if (ab)
						mix(std::integral_constant<int, 1>{});
					else
						mix(std::integral_constant<int, 0>{});
				}
				/* D /= nf (d8psk.c:377).  q0 = x*RN(1/nf); q = fma(fma(-q0, nf, x), RN(1/nf), q0)
				 * is the correctly rounded quotient for every |x| >= 1e-30 (exhaustively
				 * checked for nf = 23, 24: tests/ctests/div_check.c); below that, and only
				 * then, the plain IEEE division is used */
				if (__all(fabsf(acc.x) >= 1e-30f && fabsf(acc.y) >= 1e-30f)) {
					const float q0r = acc.x * rfn, q0i = acc.y * rfn;
					res.x = fmaf(fmaf(-q0r, fn, acc.x), rfn, q0r);
					res.y = fmaf(fmaf(-q0i, fn, acc.y), rfn, q0i);
				} else {
					res.x = acc.x / fn;
					res.y = acc.y / fn;
				}
			}
			K1F_STAMP(4);	/* mix + divide */
			/* exactly one store instruction per superperiod and wavefront: four whole lines (channels beyond nbch masked
			 * off; a wavefront without any channel, or the missing second half of the family's last pair, still issues
			 * it, with no lane enabled, so that the count above holds) */
#ifndef K1F_NOSTORE
			k1_store_masked(dec + (long long)(2 * p0 + ab) * (8 * K1F_PER_OUT), dvo, res, active && (ab == 0 || has_b));
#endif
			K1F_STAMP(5);	/* store issue */
		}
		/* no second barrier: the next iteration writes the other copy of the slices, and the one after that writes this
		 * one only behind the next iteration's barrier, which every wave reaches after its reads here */
		p0 = p1;
		pos = pos + 1 == CP ? 0 : pos + 1;
		buf ^= 1;
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef K1F_PROF
	K1F_STAMP(6);	/* drain */
	if (lane == 0 && blockIdx.y == 0 && blockIdx.x * 2 + wv < K1F_PROF_SLOTS) {
		for (int i = 0; i < 7; ++i)
			k1f_prof[blockIdx.x * 2 + wv][i] = pf[i];
		k1f_prof[blockIdx.x * 2 + wv][7] = (unsigned)nit;
		k1f_prof[blockIdx.x * 2 + wv][8] = (unsigned)wall_clock64() - wall0;	/* 100 MHz ticks of the wavefront's life */
		k1f_prof[blockIdx.x * 2 + wv][9] = wall0;
		k1f_prof[blockIdx.x * 2 + wv][10] = pf[7];
		k1f_prof[blockIdx.x * 2 + wv][11] = pf[8];
	}
#endif
}

#endif
