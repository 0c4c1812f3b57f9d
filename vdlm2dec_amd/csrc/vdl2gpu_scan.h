/* vdl2gpu_scan.h -- K2a: sync scan (screens, survivors, probe / regions / verify) and the workgroup sort.  Part of the device side of libvdl2gpu.so; included by vdl2gpu_kernels.h only. */
#ifndef VDL2GPU_SCAN_H
#define VDL2GPU_SCAN_H

/* ====================================================================== K2a
 * Sync scan.  For a run of evaluation instants n = nbase + S*i of one channel and a set of FIR
 * sub-phases r, compute the filtered phase P_r(n) and the free-running fit error E_r(n) from
 * P_r(n), P_r(n-8), .. P_r(n-128), and test where the idle detector would fire:
 *     E_r(n-2) < 4 && E_r(n) > E_r(n-2)
 * (S = 1: every sample, both parities; S = 2: one parity only -- n-8l and n-2 keep n's parity.)
 *
 * Three uses, all the same tile routine on K2A_TS instants staged in LDS:
 *   k2a_probe   every second sample of the push, carry included, in ONE fixed class (sub-phase 0, the parity of the first
 *               carried frame with history).  Finds every burst (a burst fires the
 *               detector in all 8 (sub-phase, parity) classes within a few samples) and is
 *               already the complete table for that sub-phase.
 *   k2a_region  the other three sub-phases, only in the neighbourhood of the probe's hits.
 *   k2a_verify  after the resolver: every stretch the real chain idled through in a class the
 *               probe did not cover is scanned in exactly that class; a hit means the tables
 *               missed an event: it is listed, and the repair round re-resolves the channel with it
 *               (what still fails is redone serially by K2f).  This is what makes the shortcut
 *               exact instead of heuristic.
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
#define VDL2_REG_CAP 4096	/* probe-hit regions per channel per push (noise alone seeds ~100 per million 84 kS/s samples) */
#ifndef VDL2_REG_PAD
#define VDL2_REG_PAD 40
#endif
//		/* samples scanned on either side of a probe hit */
#ifndef VDL2_REG_GAP
#define VDL2_REG_GAP 96
#endif
//		/* hits closer than this share a region */
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

struct alignas(16) K2aShared {
	float2 xs[K2A_XMAX + 8];	/* S = 1: samples in order; S = 2: even samples, then (at K2A_XODD) odd samples, so that
					 * both FIR tap parities are unit-stride across lanes */
	float2 wu[K2A_TS + K2A_POFF + 2];	/* unit phasor of every filtered sample (history first), then in place the phasor
					 * of the symbol-spaced phase step */
	float smf[72];			/* low-pass taps mflt[] (d8psk.h:28-45) */
	float atab[VDL2_ATAN_ROWS * VDL2_ATAN_STRIDE];	/* atanf range constants (vdl2_math.h) */
	int wl[K2A_WL];			/* evaluations of the current tile the first screen lets through */
	K2aDef dl[K2A_DEF];		/* survivors of both screens, collected over tiles until there are enough to
					 * give every lane an exact phase to compute (k2a_flush) */
	float sph[K2A_WL2][3][17];	/* exact phases of one batch of survivors: evaluation before / at / after */
	float we[3][K2A_WL2], wf[K2A_WL2];	/* their exact fit errors, and the slope at the middle one */
	int nwl, ndl;
	unsigned long long prof[16];	/* diagnostics: stage cycle counters of the workgroup's first lane, added to p.dbg at the end
					 * (a global atomic per stamp would sit in front of the tile's next s_waitcnt vmcnt) */
};
static_assert((K2A_XMAX / 2 + 4) % 2 == 0 && (K2A_XMAX + 8) % 2 == 0, "16-byte reads of xs[] and wu[]");
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
		/* The free-running test fires at EVERY rising step while the previous error is below 4 -- two or three instants
		 * in a row behind each minimum (17.6 candidates per burst: 8 classes x 2.2) --, but a real detector takes the
		 * first and is busy from then on; a later one of the run could only be taken by a chain that turns idle exactly
		 * between two of them (one burst ending inside another's sync word).  Only the first of a run is listed: the
		 * tables hold twice the bursts.  Should a chain ever stand between two steps of a run, the verify pass finds the
		 * unlisted hit (it tests every instant of the stretch the chain idled through) and the repair rounds, which list
		 * everything, put it in. */
		if (p.round == 0 && !p.full_scan && p2err < 4.0f && perr > p2err)	/* (VDL2GPU_F_FULLSCAN has no verify pass behind it: it lists everything) */
			return;
	} else {
		if (n < chk_lo || n >= chk_hi)
			return;
		/* a detector hit the tables did not list: remember where, and list it -- a candidate without a cluster: the repair
		 * round is the resolver alone, over the tables plus these (it replays a candidate that has no cluster with the
		 * serial machine and is back on the tables behind it).  Not a duplicate: the chain idled through this instant in
		 * this class, so the table had nothing here. */
		atomicMin(fail, (int)(n - dec_base));
		const unsigned kc = atomicAdd(cntp, 1u);
		if (kc < VDL2_CAND_CAP) {
			Cand cd;
			cd.nrel = (int)(n - dec_base);
			cd.r = r;
			cd.p2err = p2err;
			cd.perr = perr;
			cd.err = err;
			cd.pfr = pfr;
			cl[kc] = cd;
			p.clhead[(size_t)sc * VDL2_CAND_CAP + kc] = cl_pack(0, CL_INVALID, 0, 0, 0, 0, 0);
		} else
			*ovf = 1u;
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

/* Filtered samples of the tile instants q0 (even) and q0 + 1, sub-phase taps mf[] (d8psk.c:219-228), for
 * the screens only: fused multiply-adds,
 * and the 18 or 19 samples the two share are read once, 16 bytes per lane and read -- the filter pass is
 * bound by LDS bandwidth (one pipe per CU against four SIMDs), not by arithmetic.  A tap that does not
 * exist has mf[] = 0. */
template <int S> __device__ __forceinline__ void k2a_fir2(const K2aShared &sh, int q0, const float (&mf)[17], v2f &acc0, v2f &acc1)
{
	typedef float v4f __attribute__((ext_vector_type(4)));
	acc0 = (v2f){0.0f, 0.0f};
	acc1 = (v2f){0.0f, 0.0f};
	if (S == 2) {
		/* instant q, tap j: even j = 2m -> even sample q + m, odd j = 2m + 1 -> odd sample q + m */
		const v4f *pe = reinterpret_cast<const v4f *>(&sh.xs[q0]);
		const v4f *po = reinterpret_cast<const v4f *>(&sh.xs[K2A_XODD + q0]);
		v4f e[5], o[5];
#pragma unroll
		for (int i = 0; i < 5; ++i) {
			e[i] = pe[i];
			o[i] = po[i];
		}
		v2f xe[10], xo[10];
#pragma unroll
		for (int i = 0; i < 5; ++i) {
			xe[2 * i] = e[i].xy;
			xe[2 * i + 1] = e[i].zw;
			xo[2 * i] = o[i].xy;
			xo[2 * i + 1] = o[i].zw;
		}
#pragma unroll
		for (int m = 0; m < 9; ++m) {
			acc0 = __builtin_elementwise_fma(xe[m], (v2f){mf[2 * m], mf[2 * m]}, acc0);
			acc1 = __builtin_elementwise_fma(xe[m + 1], (v2f){mf[2 * m], mf[2 * m]}, acc1);
			if (m < 8) {
				acc0 = __builtin_elementwise_fma(xo[m], (v2f){mf[2 * m + 1], mf[2 * m + 1]}, acc0);
				acc1 = __builtin_elementwise_fma(xo[m + 1], (v2f){mf[2 * m + 1], mf[2 * m + 1]}, acc1);
			}
		}
	} else {
		const v4f *pe = reinterpret_cast<const v4f *>(&sh.xs[q0]);
		v4f e[9];
#pragma unroll
		for (int i = 0; i < 9; ++i)
			e[i] = pe[i];
		v2f x[18];
#pragma unroll
		for (int i = 0; i < 9; ++i) {
			x[2 * i] = e[i].xy;
			x[2 * i + 1] = e[i].zw;
		}
#pragma unroll
		for (int j = 0; j < 17; ++j) {
			acc0 = __builtin_elementwise_fma(x[j], (v2f){mf[j], mf[j]}, acc0);
			acc1 = __builtin_elementwise_fma(x[j + 1], (v2f){mf[j], mf[j]}, acc1);
		}
	}
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
	const bool prof = p.dbg && (mode == 2 || (mode == 0 && S == 1 && skip_r >= 0)) && tid == 0 && (blockIdx.x & 7) == 0;	/* probe, region scan */
	long long tq = prof ? clock64() : 0;
#define K2A_STAMP(slot) do { if (prof) { const long long tn = clock64(); sh.prof[slot] += (unsigned long long)(tn - tq); tq = tn; } } while (0)
	if (!pre.loaded)
		k2a_fetch<S>(pre, p, sc, dec_base, nbase, cnt);
	__syncthreads();
	K2A_STAMP(12);
#pragma unroll
	for (int k = 0; k < K2aPre<S>::NL; ++k) {
		const int i = tid + k * K2A_THREADS;
		if (i < nx)
			sh.xs[S == 2 ? (i & 1) * K2A_XODD + (i >> 1) : i] = pre.v[k];
	}
	pre.loaded = false;
	K2A_STAMP(13);
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
		/* ---- unit phasors of the filtered samples of instants -PH .. cnt-1 */
		for (int q0 = 2 * tid; q0 < cnt + PH; q0 += 2 * K2A_THREADS) {	/* (the odd one out at the end lands in wu[]'s padding) */
			v2f acc[2];
			k2a_fir2<S>(sh, q0, mf, acc[0], acc[1]);
			float4 out;
#pragma unroll
			for (int h = 0; h < 2; ++h) {
				const float n2 = __fmaf_rn(acc[h].x, acc[h].x, acc[h].y * acc[h].y);
				v2f w = acc[h] * __frsqrt_rn(n2);
				if (!(n2 >= 1e-30f && n2 <= 1e30f)) {	/* atan2f(0, 0) = 0; anything else odd: let it through */
					const float bad = (acc[h].x == 0.0f && acc[h].y == 0.0f) ? 0.0f : __builtin_nanf("");
					w = (v2f){1.0f + bad, bad};
				}
				if (h == 0) {
					out.x = w.x;
					out.y = w.y;
				} else {
					out.z = w.x;
					out.w = w.y;
				}
			}
			*reinterpret_cast<float4 *>(&sh.wu[q0]) = out;
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
				sh.prof[10] += (unsigned long long)nwl;
			if (nwl > K2A_WL) {
				piece = K2A_WL;
				__syncthreads();	/* everyone has read nwl before it is reset */
				continue;
			}
			if (ndl0 + nwl > K2A_DEF) {	/* no room for this pass's survivors: work the list off first */
				K2A_STAMP(6);
				k2a_flush(sh, p, sc, dec_base, mode, fail, skip_r, skip_par);
				K2A_STAMP(7);
			}
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
			sh.prof[11] += 1ull;
	}
#undef K2A_STAMP
}

__global__ __launch_bounds__(K2A_THREADS) __attribute__((amdgpu_waves_per_eu(4, 8)))
void k2a_probe(K2Params p)
{
	__shared__ K2aShared sh;
	const int c = blockIdx.y, s = blockIdx.z;
	const int sc = s * VDL2_CS + c;
	const long long dec_base = p.dec_base;
	const long long avail_end = dec_base + VDL2_CARRY_FRAMES + p.J;
	if (p.force_serial)
		return;
	if (p.round > 0 && (!p.full_round || p.fail[sc] >= VDL2_VERIFIED))
		return;		/* in a repair round the probe only runs as the complete scan of a failing channel */
	if (threadIdx.x < 16)
		sh.prof[threadIdx.x] = 0;
	k2a_tables(sh);
	/* each workgroup walks tiles blockIdx.x, blockIdx.x + gridDim.x, ... of its channel */
	if (p.full_scan || p.full_round) {
		K2aPre<1> pre;
		pre.loaded = false;
		const long long step = (long long)gridDim.x * K2A_TS;
		for (long long n0 = p.scan_lo + (long long)blockIdx.x * K2A_TS; n0 < avail_end; n0 += step) {
			const int nt = (int)((avail_end - n0 < K2A_TS) ? (avail_end - n0) : K2A_TS);
			const long long n1 = n0 + step;
			const int nt1 = n1 < avail_end ? (int)((avail_end - n1 < K2A_TS) ? (avail_end - n1) : K2A_TS) : 0;
			k2a_tile<1>(sh, p, sc, dec_base, n0, nt, 0xfu, 0, 0, 0, nullptr, pre, n1, nt1);
		}
		k2a_flush(sh, p, sc, dec_base, 0, nullptr, -1, 0);
		return;
	}
	/* ONE class everywhere: sub-phase probe_r at the instants of scan_lo's parity -- any class finds the bursts; which
	 * stretches the chain relied on in OTHER classes is what the verify pass re-scans */
	K2aPre<2> pre;
	pre.loaded = false;
	const unsigned rmask = 1u << p.probe_r;
	const long long step = 2LL * gridDim.x * K2A_TS;
	for (long long n0 = p.scan_lo + 2LL * blockIdx.x * K2A_TS; n0 < avail_end; n0 += step) {
		const long long left = (avail_end - n0 + 1) / 2;
		const int nt = (int)(left < K2A_TS ? left : K2A_TS);
		const long long n1 = n0 + step;
		const long long left1 = (avail_end - n1 + 1) / 2;
		const int nt1 = n1 < avail_end ? (int)(left1 < K2A_TS ? left1 : K2A_TS) : 0;
		k2a_tile<2>(sh, p, sc, dec_base, n0, nt, rmask, 2, 0, 0, nullptr, pre, n1, nt1);
	}
	k2a_flush(sh, p, sc, dec_base, 2, nullptr, -1, 0);
	__syncthreads();
	if (p.dbg && threadIdx.x < 16 && sh.prof[threadIdx.x])
		atomicAdd(p.dbg + 32 + threadIdx.x, sh.prof[threadIdx.x]);
}

/* ---- workgroup sort of up to VDL2_CAND_CAP 64-bit keys whose top bits are a time stamp.
 * Detector events are spread over the push (a few per burst, bursts are sparse), so a bucket
 * sort on time -- histogram, scan, scatter, then ranking inside each bucket -- needs
 * about a dozen barriers where a bitonic network needs log^2(n)/2 = 78.  A push whose events
 * pile up in one bucket (more than WGS_MAXB) falls back to the bitonic network.
 *   keys[] in/out (LDS), all different; tmp[] scratch (LDS), both VDL2_CAND_CAP long; time = key >> tshift, < range. */
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
	/* inside a bucket: every key finds its rank by counting the smaller ones (keys are unique; a burst puts
	 * a dozen or two into one bucket, so this is a handful of independent LDS reads per key where an
	 * insertion sort by one lane would be a long dependent chain) */
	for (int i = tid; i < n; i += NT) {
		const unsigned long long v = ws.tmp[i];
		unsigned b = (unsigned)(v >> tshift) >> bsh;
		b = b < WGS_NBK ? b : WGS_NBK - 1;
		const int lo = (int)ws.start[b], hi = (int)ws.start[b + 1];
		int rank = 0;
		for (int j = lo; j < hi; ++j)
			rank += (ws.tmp[j] < v) ? 1 : 0;
		keys[lo + rank] = v;
	}
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
	if (p.round > 0 && !p.full_round)
		return;		/* (launched in round 0, and in a complete round for the reset below) */
	if (p.full_round) {
		/* the channel's tables are made again from nothing, by a scan of every class at every instant */
		if (tid == 0) {
			p.ctl[CTL_CAND0 + sc] = 0;
			p.ctl[CTL_CAND0 + p.nstreams * VDL2_CS + sc] = 0;
			p.ctl[CTL_NCLUST0 + sc] = 0;
			p.ctl[CTL_NREG0 + sc] = 0;
			p.ctl[CTL_NSEED0 + sc] = 0;
		}
		return;
	}
	int ncand = (int)p.ctl[CTL_NSEED0 + sc];
	ncand = ncand > VDL2_CAND_CAP ? VDL2_CAND_CAP : ncand;
	const int *seeds = p.seeds + (size_t)sc * VDL2_CAND_CAP;
	for (int i = tid; i < ncand; i += K2R_NT)	/* the index makes equal instants distinct keys (the sort ranks keys) */
		key64[i] = ((unsigned long long)(unsigned)seeds[i] << 16) | (unsigned)i;
	__syncthreads();
	wg_sort_u64<K2R_NT>(key64, ws, ncand, 16, (unsigned)(VDL2_CARRY_FRAMES + p.J));
	for (int i = tid; i < ncand; i += K2R_NT)
		key[i] = (int)(key64[i] >> 16);
	__syncthreads();
	{
		/* every run of hits closer than VDL2_REG_GAP becomes a region (order is irrelevant) */
		__shared__ int s_nreg;
		const int lo_lim = (int)(p.scan_lo - p.dec_base);
		const int hi_lim = (int)(VDL2_CARRY_FRAMES + p.J);
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
			/* more regions than the list holds: the surplus is dropped -- regions are a cost decision, what a
			 * missing one would have found the verify pass finds (and a repair round scans the channel completely) */
			p.ctl[CTL_NREG0 + sc] = (unsigned)(n > VDL2_REG_CAP ? VDL2_REG_CAP : n);
			p.ctl[CTL_NSEED0 + sc] = 0;
		}
	}
}

__global__ __launch_bounds__(K2A_THREADS) __attribute__((amdgpu_waves_per_eu(4, 8)))
void k2a_region(K2Params p)
{
	__shared__ K2aShared sh;
	const int c = blockIdx.y, s = blockIdx.z;
	const int sc = s * VDL2_CS + c;
	if (p.force_serial || p.full_scan)
		return;
#ifdef VDL2GPU_TESTHOOKS
	if (p.test_noregion && p.round == 0)
		return;
#endif
	const unsigned nreg = p.ctl[CTL_NREG0 + sc];	/* (round 0 only: a repair round re-resolves or re-scans completely, it scans no regions) */
	if (blockIdx.x >= nreg)	/* nothing for this workgroup (64 channels x 128 workgroups, 40 regions each): not even the tables */
		return;
	const long long dec_base = p.dec_base;
	const int2 *regs = p.regs + (size_t)sc * VDL2_REG_CAP;
	const int skip_r = p.probe_r, skip_par = p.probe_par;
	if (threadIdx.x < 16)
		sh.prof[threadIdx.x] = 0;
	k2a_tables(sh);
	K2aPre<1> pre;
	pre.loaded = false;
	const bool prof = p.dbg && threadIdx.x == 0 && (blockIdx.x & 7) == 0;
	const long long t0 = prof ? clock64() : 0;
	for (unsigned k = blockIdx.x; k < nreg; k += gridDim.x) {
		const int2 rg = regs[k];
		const int2 rn = (k + gridDim.x < nreg) ? regs[k + gridDim.x] : make_int2(0, 0);
		k2a_tile<1>(sh, p, sc, dec_base, dec_base + rg.x, rg.y, 0xfu, 0, 0, 0, nullptr, pre, dec_base + rn.x, rn.y, skip_r, skip_par);
		if (prof)
			sh.prof[15] += (unsigned long long)rg.y;	/* instants */
	}
	const long long t1 = prof ? clock64() : 0;
	if (prof)
		sh.prof[13] += (unsigned long long)sh.ndl;	/* survivors left for the final flush */
	k2a_flush(sh, p, sc, dec_base, 0, nullptr, skip_r, skip_par);
	if (prof) {
		sh.prof[14] += (unsigned long long)(clock64() - t1);	/* final flush */
		sh.prof[12] += (unsigned long long)(t1 - t0);		/* all tiles */
	}
	__syncthreads();
	if (p.dbg && threadIdx.x < 16 && sh.prof[threadIdx.x])
		atomicAdd(p.dbg + 48 + threadIdx.x, sh.prof[threadIdx.x]);
}

/* one workgroup = K2A_VRUN tiles of 2*K2A_TS samples; every piece of a verify segment inside a tile
 * is scanned in the segment's class */
#ifndef K2A_VRUN
#define K2A_VRUN 4
#endif
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
	if (p.force_serial || p.full_scan || p.full_round)
		return;
	if (p.round > 0 && !p.redo[sc])
		return;
	const long long dec_base = p.dec_base;
	const int r_lo = (int)(p.cs[sc].pos - dec_base) + (int)blockIdx.x * K2A_VRUN * 2 * K2A_TS;
	const int t_end = (int)(VDL2_CARRY_FRAMES + p.J);
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

#endif
