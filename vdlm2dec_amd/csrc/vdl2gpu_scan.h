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
#ifndef K2A_PREFETCH
#define K2A_PREFETCH 0		/* 1: a workgroup loads its next tile's samples into registers behind the current tile's filter pass (twenty
				 * registers through the screens: the kernels then need more than the 80 that six wavefronts per SIMD leave) */
#endif
#ifndef K2A_FLUSH_ATTR
#define K2A_FLUSH_ATTR
#endif
#ifndef K2A_WPE
#define K2A_WPE 6		/* wavefronts per SIMD the scan kernels are compiled for */
#endif
#define K2A_POFF 132		/* samples of phase history before the tile: 128 + 4 */
#define K2A_XOFF (K2A_POFF + 16)
#define K2A_XMAX (2 * K2A_TS + K2A_XOFF)
#define K2X_WL 40		/* sparse stages (k2x_chunk): survivors whose exact phases are in LDS at once (2040 phases) */
#define K2X_NT 256		/* ... and the items of a chunk: a lane each */
#define K2X_CV 64		/* of which the fit screen of so many runs in ONE wavefront (11 % get that far) */
#define VDL2_ITEM_CAP 196608	/* (at most: K2Params.item_cap is sized by the longest part a handle can be given, vdl2gpu_create) evaluations per channel and scan that pass the first screen (2.7 % of the instants on noise and on
				 * payload symbols alike: 39 000 of a 33 s class-scan).  The list is in two parts: a private area per scan
				 * workgroup (p.surv_pch items each, filled through an LDS counter and worked off by the workgroup itself behind its last tile: k2a_tail) and behind
				 * them a common area for what a workgroup's own area does not hold (one device-scope atomic per wavefront
				 * and pass: slow, rare) */
#define VDL2_ITEM_PRIV 131072	/* ... of which private areas at most */
#define VDL2_MAXWG 512		/* scan workgroups per channel at most (private areas of >= 256 items) */
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

/* An evaluation that passed the first screen, as the scan's tile loop hands it to the sparse stages (k2x_chunk): five 16-byte words -- {n, r | odd << 8, lo, hi}
 * and its sixteen phase-step phasors (oldest first) as 2 x 16-bit fixed point (1 / 32767: 2e-5 rad, v_cvt_pknorm_i16_f32) --,
 * 80 bytes instead of 144 (the lists are written by one kernel and read by the next: 25 MB per class-scan of a 33 s push).
 * An item with a non-finite phasor carries the `odd` flag instead (the exact stage then looks at it).  Within an area of the
 * list (a scan workgroup's private area, or the common area) word w of item i lies at (w * area_items + i): consecutive items
 * -- a wavefront's lanes, on either side -- are consecutive in memory. */
#define K2A_ITEM_WORDS 5
struct K2aItem { float4 w[K2A_ITEM_WORDS]; };	/* (size only: see above for the layout) */

/* LDS of a scan workgroup: 19.1 KB (the verify pass: 20.4 KB), seven or eight workgroups per CU as far as LDS goes.
 *   xs[]  S = 2 (one class: probe, verify): even samples, then (at K2A_XODD) odd samples, so that both FIR tap parities are
 *         unit-stride across lanes; once the filter pass has read them the SAME memory takes the phase-step phasors
 *         (wu = xs: the filter's results wait in registers for the barrier).
 *         S = 1 (all classes: region scan, complete scan): samples in order in the lower half -- they are filtered once per
 *         sub-phase --, the phasors behind them (wu = xs + K2A_WU1). */
#define K2A_WU1 (K2A_TS + K2A_XOFF + 4)
#define K2A_XS_LEN (K2A_WU1 + K2A_TS + K2A_POFF + 4)
struct alignas(16) K2aShared {
	float2 xs[K2A_XS_LEN];
	float smf[72];			/* low-pass taps mflt[] (d8psk.h:28-45) */
	unsigned it_used, it_limit;	/* items the workgroup has put into its private area; where the first group that did not fit would have begun */
	unsigned short pend[K2A_THREADS / 64][K2A_TS / (K2A_THREADS / 64)];	/* per wavefront: the evaluations of a tile and sub-phase that passed the
					 * first screen (instant within the tile | 0x8000: a non-finite phasor), appended as ONE group behind the screen passes */
	unsigned long long prof[16];	/* diagnostics: stage cycle counters of the workgroup's first lane, added to p.dbg at the end
					 * (a global atomic per stamp would sit in front of the tile's next s_waitcnt vmcnt) */
};
static_assert(K2A_WU1 % 2 == 0 && K2A_XS_LEN >= K2A_XMAX + 8 && (K2A_XMAX / 2 + 4) % 2 == 0, "16-byte reads of xs[] and wu[]");
static_assert(K2A_TS + K2A_POFF / 2 + 2 <= K2A_XS_LEN && K2A_WU1 >= K2A_TS + K2A_XOFF, "phasor areas");
static_assert(K2A_TS <= 1024, "pend[] holds an instant within the tile in ten bits");
#define K2A_XODD (K2A_XMAX / 2 + 4)	/* 8-byte elements: an odd multiple of 64 bytes away, so the two halves use disjoint banks */
/* S = 4 (the probe since round 5: every FOURTH sample, one sub-phase -- it only has to FIND the bursts, see k2a_probe): a tile is
 * 512 instants (the same 2048 samples as an S = 2 tile), its samples lie de-interleaved by (index mod 4) in four runs of K2A_XQ4
 * elements (tap j = 4 m + t of instant q is element q + m of run t: unit stride across lanes for every tap), 64 bytes apart
 * modulo the banks' 256. */
#define K2A_TS4 (K2A_TS / 2)
#ifndef VDL2_PROBE_STRIDE
#define VDL2_PROBE_STRIDE 2	/* samples between the probe's instants: 2 = one class completely (round 4), 4 = every second instant of it, seeds only */
#endif
#define K2A_XQ4 552
static_assert(4 * K2A_XQ4 <= K2A_XS_LEN && K2A_XQ4 % 32 == 8 && (4 * (K2A_TS4 - 1) + 1 + K2A_XOFF + 3) / 4 <= K2A_XQ4 &&
	      K2A_TS4 + K2A_POFF / 4 - 2 + 6 <= K2A_XQ4, "S = 4 sample runs: a tile's samples, and the six elements a lane pair reads of each run");
template <int S> struct K2aTs { static constexpr int v = (S == 4) ? K2A_TS4 : K2A_TS; };

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
/* Fourth screen: the fit error itself, from approximate angles.  The reference's unwrapped phases (d8psk.c:262-274) differ
 * from one symbol to the next by the template-corrected phase step wrapped once into [-pi, pi] -- the angle D_l of the
 * rotated step phasor c_l * u_l the second screen has in registers (unless |D_l| is within rounding of pi, where the fit
 * error is far above any threshold anyway) -- so with q_l = D_1 + .. + D_l (q_0 = 0: the fit error does not depend on the
 * first phase) the reference's err is  sum q^2 - (sum q)^2 / 17 - (sum q (l - 8))^2 / 408  in real arithmetic.  With angles
 * good to 2e-5 rad (polynomial) + 1e-5 (fused filter, reciprocal square root, two phasor products) every q_l is good to
 * l * 3e-5, the residual vector to 3e-5 * sqrt(sum l^2) = 1.2e-3 in norm, err to 2 * sqrt(7.25) * 1.2e-3 + 3e-3 (float
 * cancellation in the three sums, q up to 50) < 0.01: an evaluation whose approximate err exceeds the detector's 4 (the
 * probe's seed limit 7) by VDL2_FIT_MARGIN = 0.25 cannot be below it exactly.  On noise 1.5 % of what the second and third
 * screens let through is below 7.25 and 0.05 % below 4.25 (numpy model of the chain over 2 M instants): the exact phases
 * (51 filters and 51 atan2f per survivor: a third of the probe's and the verify pass's vector instructions) are computed
 * around sync words only. */
#define VDL2_FIT_MARGIN 0.25f
#define VDL2_NB_MARGIN 0.1f	/* the sparse stages' fifth screen: what two approximate fit errors must differ by to be ordered */
__device__ __forceinline__ float k2_fast_angle(float y, float x)
{
	const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
	const float mn = __builtin_fminf(ax, ay), mx = __builtin_fmaxf(ax, ay);
	const float t = mn * __builtin_amdgcn_rcpf(mx);
	const float s = t * t;
	float p = __fmaf_rn(s, -0.0117212f, 0.05265332f);
	p = __fmaf_rn(s, p, -0.11643287f);
	p = __fmaf_rn(s, p, 0.19354346f);
	p = __fmaf_rn(s, p, -0.33262347f);
	p = __fmaf_rn(s, p, 0.99997726f) * t;
	p = ay > ax ? 1.57079637f - p : p;
	p = x < 0.0f ? 3.14159274f - p : p;
	return __builtin_copysignf(p, y);
}

__device__ __forceinline__ v2f k2_rot(v2f acc, v2f u, float cx, float cy)
{
	/* acc += (cx + j cy) * u */
	acc = __builtin_elementwise_fma((v2f){cx, cx}, u, acc);
	return __builtin_elementwise_fma((v2f){-cy, cy}, u.yx, acc);
}


/* detector test of one instant (d8psk.c:292) and what a hit means in each scan mode */
template <class PR> __device__ __forceinline__ void k2a_emit(PR p, int sc, long long dec_base, long long n, int r, int mode,
						  long long chk_lo, long long chk_hi, int *fail, int skip_r, int skip_par,
						  float p2err, float perr, float err, float pfr, unsigned *cntp, unsigned *ovf, Cand *cl)
{
	if (mode >= 2 && perr < VDL2_SEED_ERR && err > perr) {
		const unsigned kk = atomicAdd(p.ctl + CTL_NSEED0 + sc, 1u);
		if (kk < VDL2_CAND_CAP)	/* surplus seeds are simply dropped: K2a-verify covers what they would have */
			p.seeds[(size_t)sc * VDL2_CAND_CAP + kk] = (int)(n - dec_base);
	}
	if (mode == 3)	/* the probe: it finds the bursts, it lists no candidates (it looks at every second instant of its class only) */
		return;
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
 * same workgroup are loaded once the current tile's filter pass is through (its sample registers are free then), so that
 * the memory latency (several thousand cycles under load) is hidden behind the current tile's screens. */
template <int S> struct K2aPre {
	static constexpr int NL = (S * (K2aTs<S>::v - 1) + 1 + K2A_XOFF + K2A_THREADS - 1) / K2A_THREADS;
	float2 v[NL];
	bool loaded;
	int tiles;		/* tiles this workgroup has done: which of its wavefronts takes the sparse pass rotates */
};

/* once per workgroup, before its first tile */
__device__ __forceinline__ void k2a_tables(K2aShared &sh)
{
	for (int i = threadIdx.x; i < 72; i += K2A_THREADS)
		sh.smf[i] = (i < 65) ? d_tab(c_mflt, i) : 0.0f;
	if (threadIdx.x == 0) {
		sh.it_used = 0;
		sh.it_limit = 0xffffffffu;
	}
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

/* The tile loop of a scan kernel does the DENSE part only: filter, phasors and the first screen at every instant of a tile, all
 * four wavefronts of a workgroup in step.  What passes the first screen (2.7 % of the instants) is put aside as an item --
 * instant, class and the sixteen phase-step phasors it was screened on -- in the workgroup's private area of the channel's item
 * list, and behind its last tile the workgroup takes its area through the sparse stages with every lane busy (k2a_tail ->
 * k2x_chunk): second and third screen, the fit screen, then for the few survivors exact phases, exact fits and the detector
 * test.  [Until round 4 the sparse stages ran inside the tile loop: one wavefront of four worked on 27 items of 64 lanes while
 * the other three waited at the barrier -- a third of the probe's time -- and the exact work's live registers on top of the
 * tile loop's capped the scan kernels at four wavefronts per SIMD; round 4 ran them as a kernel of its own behind every scan:
 * four more launches a push, 30 us each as they ran.  Behind the tile loop the exact work has the registers to itself.]
 * mode 0: append candidates; mode 1: report hits in [chk_lo, chk_hi) to *fail and append them; mode 2: probe (candidates + seeds). */

/* The lanes of a wavefront whose evaluation passed the first screen append their items: one atomic per wavefront and pass,
 * the items in lane order (= time order: k2x_chunk finds an evaluation's neighbours next to it). */
__device__ __forceinline__ void k2a_append(K2aShared &sh, const K2Params &p, int sc, bool pass, int n_rel, int r, int lo, int hi, const float2 (&uu)[16], bool odd, int mode, int *fail)
{
	const unsigned long long m = __ballot(pass);
	if (m == 0)
		return;
	const unsigned lane_rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
	const unsigned cnt = (unsigned)__popcll(m);
	unsigned base = 0, stride = 0;	/* in 16-byte words: where word 0 of the group's first item goes, and from word to word */
	if (lane_rank == 0 && pass) {
		const unsigned pch = (unsigned)p.surv_pch;
		const unsigned off = atomicAdd(&sh.it_used, cnt);	/* LDS */
		if (off + cnt <= pch) {
			base = blockIdx.x * pch * K2A_ITEM_WORDS + off;
			stride = pch;
		} else {
			/* the workgroup's own area is full (a stretch where far more than 2.7 % pass: a carrier, a run of sync words):
			 * the common area behind the private ones, through the channel's counter in device memory */
			atomicMin(&sh.it_limit, off);
			const unsigned priv = (unsigned)p.surv_nwg * pch;
			const unsigned off2 = atomicAdd(p.ctl + CTL_NSURV0 + p.surv_slot * p.nstreams * VDL2_CS + sc, cnt);
			stride = p.item_cap - priv;
			const unsigned room = (p.surv_common_cap > 0 && (unsigned)p.surv_common_cap < stride) ? (unsigned)p.surv_common_cap : stride;
			base = (off2 + cnt <= room) ? priv * K2A_ITEM_WORDS + off2 : 0xffffffffu;
			if (base == 0xffffffffu)	/* refused: the drain must not read the common area from here on (nothing was written there) */
				atomicMax(p.ctl + CTL_NSURVLIM0 + p.surv_slot * p.nstreams * VDL2_CS + sc, ~off2);
		}
	}
	const int lead = __builtin_ctzll(m);
	base = (unsigned)__builtin_amdgcn_readlane((int)base, lead);
	stride = (unsigned)__builtin_amdgcn_readlane((int)stride, lead);
	if (!pass)
		return;
	if (base != 0xffffffffu) {
		float4 *dst = reinterpret_cast<float4 *>(p.items) + (size_t)sc * p.item_cap * K2A_ITEM_WORDS + base + lane_rank;
		dst[0] = make_float4(__int_as_float(n_rel), __int_as_float(r | (odd ? 0x100 : 0)), __int_as_float(lo), __int_as_float(hi));
#pragma unroll
		for (int w = 0; w < 4; ++w) {
			typedef short s2 __attribute__((ext_vector_type(2)));
			union { s2 v; int i; } c[4];
#pragma unroll
			for (int l = 0; l < 4; ++l)
				c[l].v = __builtin_amdgcn_cvt_pknorm_i16(uu[4 * w + l].x, uu[4 * w + l].y);
			dst[(size_t)(1 + w) * stride] = make_float4(__int_as_float(c[0].i), __int_as_float(c[1].i), __int_as_float(c[2].i), __int_as_float(c[3].i));
		}
	} else if (mode == 1)
		atomicMin(fail, 0);	/* the verify pass cannot vouch for the channel: it is re-resolved, in the end redone serially */
	else
		p.ctl[CTL_CAND0 + p.nstreams * VDL2_CS + sc] = 1u;	/* as if the candidate table had overflowed: the serial machine takes the channel */
}

/* the fourth screen's fit error from the sixteen rotated phase-step phasors (see k2_fast_angle) */
__device__ __forceinline__ float k2x_fit(const v2f (&v)[16])
{
	float q = 0.0f, sq = 0.0f, sqq = 0.0f, sql = 0.0f;
#pragma unroll
	for (int l = 0; l < 16; ++l) {
		q += k2_fast_angle(v[l].y, v[l].x);
		sq += q;
		sqq = __fmaf_rn(q, q, sqq);
		sql = __fmaf_rn(q, (float)(l - 7), sql);
	}
	return sqq - sq * sq * (1.0f / 17.0f) - sql * sql * (1.0f / 408.0f);
}

/* LDS of the sparse stages.  It overlays a scan workgroup's K2aShared once the workgroup's tiles are done (the scan works its own
 * items off itself: k2a_tail), or lies in the LDS of the one-workgroup-per-channel kernel that follows a scan on its stream and
 * drains the scan's common area (k2x_drain). */
struct alignas(8) K2xWork {
	float smf[72];
	float atab[VDL2_ATAN_ROWS * VDL2_ATAN_STRIDE];
	K2aDef dl[K2X_NT];		/* the chunk's survivors of all screens */
	int it_n[K2X_NT + 2];		/* the items' instants, ... */
	int it_st[K2X_NT + 2];		/* ... sub-phase | status << 4 (0: cannot fire, 1: fit error known approximately, 2: must be looked at exactly), ... */
	float it_ea[K2X_NT + 2];	/* ... approximate fit errors */
	union {				/* (never live together: a block barrier lies between the fit screen's last read and the exact stage's first write) */
		float2 cv[K2X_CV][17];		/* rotated phasors of the items that passed the second and third screen (row padded: a lane per row) */
		float sph[K2X_WL][3][17];	/* exact phases of one batch of survivors: evaluation before / at / after */
	} u;
	int cidx[K2X_CV];		/* ... whose the rows of cv[] are */
	int nc;
	int ns;
	int tabs;			/* smf[] and atab[] are in place (a workgroup loads them when its first chunk reaches the exact stage) */
	float we[3][K2X_WL], wf[K2X_WL];	/* exact fit errors of a batch, and the slope at the middle evaluation */
};

/* The sparse stages over one chunk of at most K2X_NT (256) consecutive items of an area of a channel's item list, by the first
 * 256 threads of a workgroup of NT (every thread of the workgroup makes the call: block barriers inside): a lane each through
 * the second, third and fourth screen (dense: the list holds nothing else), the fifth screen between neighbours, then together
 * through the exact stage for the survivors: exact phases (FIR in the reference's order from the channel plane, then atan2f,
 * d8psk.c:219-229), exact fits (d8psk.c:257-289) of the evaluation and its two neighbours, detector test (d8psk.c:292).
 * Survivors are rare since the fourth screen: a handful per sync word and class, one per 20 000 instants of noise.
 *   w0, stride   16-byte word index of the chunk's first item's word 0 in the channel's list; words from word to word of an item
 *   U            exact phases a lane works on at once (1 in the scan kernels, whose register budget is the tile loop's)
 * [Round 4 ran this as a kernel of its own behind every scan, four launches of 6144 workgroups a push most of which found
 * nothing and 30 us each as they ran; a scan workgroup's own area holds a chunk or two, and the workgroup is there anyway.] */
template <int NT, int U, class PR>
__device__ __forceinline__ void k2x_chunk(K2xWork &sh, PR p, int sc, int mode, int skip, unsigned w0, unsigned stride, unsigned nhere)
{
	static_assert(NT >= K2X_NT && (U == 1 || U == 2), "k2x_chunk");
	const int tid = threadIdx.x;
	const bool act = NT == K2X_NT || tid < K2X_NT;
	if (tid == 0) {
		sh.ns = 0;
		sh.nc = 0;
	}
	if (tid < 2) {
		sh.it_n[K2X_NT + tid] = 0x7fffffff;
		sh.it_st[K2X_NT + tid] = 0;
	}
	__syncthreads();
	/* exp(-j (SW[l] - SW[l-1])), l = 1..16: the template steps are 1,7,5,-7,1,3,-3,-7,3,-1,5,-5,-3,-5,-1,7 (x pi/8) */
	constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f;
	constexpr float rc[16] = {C1, -C1, -S1, -C1, C1, S1, S1, -C1, S1, C1, -S1, -S1, S1, -S1, C1, -C1};
	constexpr float rs[16] = {-S1, -S1, -C1, S1, -S1, -C1, C1, S1, -C1, S1, -C1, C1, C1, C1, S1, -S1};
	const float fit_limit = mode >= 2 ? VDL2_SEED_ERR + VDL2_FIT_MARGIN : 4.0f + VDL2_FIT_MARGIN;
	int st = 0;
	float erra = 0.0f;
	K2aDef d{};
	const bool have = act && (unsigned)tid < nhere;
	if (have) {
		const float4 *src = reinterpret_cast<const float4 *>(p.items) + (size_t)sc * p.item_cap * K2A_ITEM_WORDS + w0 + tid;
		float4 raw[K2A_ITEM_WORDS];
#pragma unroll
		for (int l = 0; l < K2A_ITEM_WORDS; ++l)
			raw[l] = src[(size_t)l * stride];
		d.n = __float_as_int(raw[0].x);
		const int rflags = __float_as_int(raw[0].y);
		d.r = rflags & 0xff;
		d.lo = __float_as_int(raw[0].z);
		d.hi = __float_as_int(raw[0].w);
		v2f v[16];
#pragma unroll
		for (int l = 0; l < 16; ++l) {
			const float4 q4 = raw[1 + l / 4];
			const int pk = __float_as_int((l & 3) == 0 ? q4.x : ((l & 3) == 1 ? q4.y : ((l & 3) == 2 ? q4.z : q4.w)));
			const v2f u = {(float)(short)(pk & 0xffff) * (1.0f / 32767.0f), (float)(pk >> 16) * (1.0f / 32767.0f)};
			v[l] = k2_rot((v2f){0.0f, 0.0f}, u, rc[l], rs[l]);
		}
		/* second screen: lag-2 steps as products of neighbouring rotated lag-1 phasors; third: lag-3 steps, 14 of them */
		v2f acc = {0.0f, 0.0f}, acc3 = {0.0f, 0.0f};
#pragma unroll
		for (int l = 0; l < 15; ++l) {
			const v2f p2 = k2_rot((v2f){0.0f, 0.0f}, v[l], v[l + 1].x, v[l + 1].y);
			acc += p2;
			if (l < 14)
				acc3 = k2_rot(acc3, p2, v[l + 2].x, v[l + 2].y);
		}
		const float r2 = __fmaf_rn(acc.x, acc.x, acc.y * acc.y);
		const float r3 = __fmaf_rn(acc3.x, acc3.x, acc3.y * acc3.y);
		if (rflags & 0x100)
			st = 2;		/* a non-finite phasor: the exact stage decides */
		else if (!(r2 <= VDL2_SCREEN_R22) && !(r3 <= VDL2_SCREEN_R32)) {
			/* fourth screen: the fit itself on approximate step angles (see k2_fast_angle) -- for the first K2X_CV of the
			 * chunk in one wavefront behind the barrier, a lane each; for more than that (a list of sync words) here */
			const int slot = atomicAdd(&sh.nc, 1);
			if (slot < K2X_CV) {
				sh.cidx[slot] = tid;
#pragma unroll
				for (int l = 0; l < 16; ++l)
					sh.u.cv[slot][l] = make_float2(v[l].x, v[l].y);
				st = 3;	/* (pending) */
			} else {
				erra = k2x_fit(v);
				st = (erra > fit_limit) ? 0 : 1;
			}
		}
	}
	if (act) {
		sh.it_n[tid] = have ? d.n : 0x7fffffff;
		sh.it_st[tid] = d.r | (st << 4);
		sh.it_ea[tid] = erra;
	}
	__syncthreads();
	{
		const int nc = sh.nc < K2X_CV ? sh.nc : K2X_CV;
		if (tid < nc) {
			v2f v[16];
#pragma unroll
			for (int l = 0; l < 16; ++l) {
				const float2 t = sh.u.cv[tid][l];
				v[l] = (v2f){t.x, t.y};
			}
			const float e = k2x_fit(v);
			const int who = sh.cidx[tid];
			sh.it_ea[who] = e;
			sh.it_st[who] = (sh.it_st[who] & 15) | ((e > fit_limit ? 0 : 1) << 4);
		}
	}
	__syncthreads();
	if (act) {
		st = sh.it_st[tid] >> 4;
		erra = sh.it_ea[tid];
	}
	if (st == 1) {
		/* Fifth screen: the detector's own test on the approximate errors of NEIGHBOURING evaluations.  A scan wavefront's items
		 * are in the list in time order, so the evaluation one step later (n + 2: the `err` to this evaluation's `perr`) is
		 * the next item or the one after, the evaluation one step earlier the previous or the one before -- or it is not in
		 * the list: then it failed the first screen (its error is above 4.25: a rising step) or it lies in another wavefront's
		 * group (unknown: treated the same, which keeps this evaluation).
		 *   falling step: err(n + 2) < err(n) -- neither the detector nor a seed can fire at n + 2 on this evaluation;
		 *   later step of a run: err(n - 2) < 4 and err(n) > err(n - 2) -- k2a_emit would not list a hit at n + 2 (only the
		 *   first firing of a run is listed: see there; the verify pass and the complete scan list everything).
		 * Both with VDL2_NB_MARGIN between the approximate values, which are good to 0.02.  Of the five evaluations around a
		 * sync word's minimum in a class, the one at the minimum is left. */
		const int same = d.r | (1 << 4);
		int nx = -1, pv = -1;
		if (sh.it_n[tid + 1] == d.n + 2 && sh.it_st[tid + 1] == same)
			nx = tid + 1;
		else if (sh.it_n[tid + 2] == d.n + 2 && sh.it_st[tid + 2] == same)
			nx = tid + 2;
		if (tid >= 1 && sh.it_n[tid - 1] == d.n - 2 && sh.it_st[tid - 1] == same)
			pv = tid - 1;
		else if (tid >= 2 && sh.it_n[tid - 2] == d.n - 2 && sh.it_st[tid - 2] == same)
			pv = tid - 2;
		if (nx >= 0 && sh.it_ea[nx] < erra - VDL2_NB_MARGIN)
			st = 0;
		const bool first_only = (mode == 0 || mode == 2) && p.round == 0 && !p.full_scan;	/* k2a_emit's condition */
		if (first_only && pv >= 0 && sh.it_ea[pv] < 4.0f - VDL2_NB_MARGIN && erra > sh.it_ea[pv] + VDL2_NB_MARGIN)
			st = 0;
	}
	if (st)
		sh.dl[atomicAdd(&sh.ns, 1)] = d;
	__syncthreads();
	const int nd = sh.ns;
#ifdef K2X_NO_EXACT
	return;
#endif
	if (nd == 0)	/* (block-uniform) */
		return;
	if (!sh.tabs) {	/* (block-uniform; the tables of the exact stage: only now, one chunk in ten gets here) */
		if (act) {
			for (int i = tid; i < 72; i += K2X_NT)
				sh.smf[i] = (i < 65) ? d_tab(c_mflt, i) : 0.0f;
			if (tid < VDL2_ATAN_ROWS * VDL2_ATAN_STRIDE)
				sh.atab[tid] = vdl2_atan_tab_entry(tid);
		}
		__syncthreads();
		if (tid == 0)
			sh.tabs = 1;
	}
	const long long dec_base = p.dec_base;
	const int skip_r = skip ? p.probe_r : -1, skip_par = p.probe_par;
	int *fail = p.fail + sc;
	const float2 *x0 = p.dec + (size_t)sc * p.cap;	/* x0[n] = sample at stream-relative time n */
	unsigned *cntp = p.ctl + CTL_CAND0 + sc;
	unsigned *ovf = p.ctl + CTL_CAND0 + p.nstreams * VDL2_CS + sc;
	Cand *cl = p.cands + (size_t)sc * VDL2_CAND_CAP;
	for (int b0 = 0; b0 < nd; b0 += K2X_WL) {
		const int nb = (nd - b0 < K2X_WL) ? nd - b0 : K2X_WL;
		/* U phases per lane and pass, their loads and dependent chains (17 taps, then the atan2f's divisions and polynomial)
		 * interleaved: the stage is a chain of latencies, not of throughput */
		for (int t0 = act ? tid : 51 * nb; t0 < 51 * nb; t0 += U * K2X_NT) {
			float2 xv[U][17];
			int rr[U], slot[U], ww[U], ll[U];
#pragma unroll
			for (int u = 0; u < U; ++u) {
				const int t = t0 + u * K2X_NT < 51 * nb ? t0 + u * K2X_NT : t0;	/* (the odd one out repeats its partner) */
				slot[u] = t / 51;
				const int rem = t - 51 * slot[u];
				ww[u] = rem / 17;
				ll[u] = rem - 17 * ww[u];
				const K2aDef dd = sh.dl[b0 + slot[u]];
				rr[u] = dd.r;
				const float2 *x = x0 + (dd.n + (ww[u] - 1) * 2 - 8 * (16 - ll[u]) - 16);
#pragma unroll
				for (int j = 0; j < 17; ++j)
					xv[u][j] = x[j];
			}
			v2f acc[U];
#pragma unroll
			for (int u = 0; u < U; ++u)
				acc[u] = (v2f){0.0f, 0.0f};
#pragma unroll
			for (int j = 0; j < 16; ++j) {
#pragma unroll
				for (int u = 0; u < U; ++u) {
					const float m = sh.smf[rr[u] + 4 * j];
					acc[u] += (v2f){xv[u][j].x, xv[u][j].y} * (v2f){m, m};
				}
			}
#pragma unroll
			for (int u = 0; u < U; ++u)
				if (rr[u] == 0) {	/* mflt[r + 64] exists only for r == 0 */
					const float m = sh.smf[64];
					acc[u] += (v2f){xv[u][16].x, xv[u][16].y} * (v2f){m, m};
				}
#pragma unroll
			for (int u = 0; u < U; ++u) {
				const float ph = vdl2_atan2f_tab(acc[u].y, acc[u].x, sh.atab);
				if (u == 0 || t0 + u * K2X_NT < 51 * nb)
					sh.u.sph[slot[u]][ww[u]][ll[u]] = ph;
			}
		}
		__syncthreads();
		for (int kk = act ? tid : 3 * nb; kk < 3 * nb; kk += K2X_NT) {
			const int slot = kk / 3, w = kk - 3 * slot;
			float fr;
			sh.we[w][slot] = k2_sync_metric<1>(&sh.u.sph[slot][w][0], &fr);
			if (w == 1)
				sh.wf[slot] = fr;
		}
		__syncthreads();
		for (int kk = act ? tid : nb; kk < nb; kk += K2X_NT) {
			const K2aDef dd = sh.dl[b0 + kk];
			k2a_emit<PR>(p, sc, dec_base, dec_base + dd.n + 2, dd.r, mode, dec_base + dd.lo, dec_base + dd.hi, fail, skip_r, skip_par,
				 sh.we[0][kk], sh.we[1][kk], sh.we[2][kk], sh.wf[kk], cntp, ovf, cl);
		}
		__syncthreads();
	}
}

/* Behind a scan workgroup's last tile: the sparse stages over what the workgroup itself put into its private area of the item
 * list, in chunks of 256 (a workgroup of the probe walks eight tiles and lists 220 items: one chunk).  The items were written
 * and are read by the same workgroup: a block barrier orders them.  K2xWork takes the place of the tile loop's LDS. */
static_assert(sizeof(K2xWork) <= sizeof(K2aShared), "the sparse stages' LDS overlays the tile loop's");
/* The tail reads the kernel's parameters AGAIN, from the kernarg segment, behind a point the compiler cannot move loads across:
 * as fields of the by-value `p` they are loaded at the kernel's entry and the dozen pointers and switches only the tail uses
 * would be held in scalar registers through the tile loop, which has none to spare (46-69 scalar and 23-47 vector registers
 * spilled when the tail took `p`; none this way).  K2Params MUST STAY the scan kernels' only explicit parameter (k2a_probe,
 * k2a_region, k2a_verify): it is read at offset 0 of the kernarg segment, and nothing but this comment and the marker test in
 * tests/test_build_resources.py would notice a second one in front of it. */
typedef const __attribute__((address_space(4))) K2Params K2ParamsK;
__device__ __forceinline__ K2ParamsK &k2_kernarg_again()
{
	unsigned long long ka = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
	asm volatile("; kernarg again" : "+s"(ka) : : "memory");
	return *(K2ParamsK *)ka;
}
__device__ __forceinline__ void k2a_tail(K2aShared &sh, int sc)
{
	__syncthreads();
	K2ParamsK &p = k2_kernarg_again();
	unsigned n = sh.it_used < sh.it_limit ? sh.it_used : sh.it_limit;
	n = n < (unsigned)p.surv_pch ? n : (unsigned)p.surv_pch;
	__syncthreads();	/* (everybody has read the counters the overlay is about to cover) */
	K2xWork &w = *reinterpret_cast<K2xWork *>(&sh);
	if (threadIdx.x == 0)
		w.tabs = 0;
	const unsigned base = blockIdx.x * (unsigned)p.surv_pch * K2A_ITEM_WORDS;
	/* the first chunk (all there is, as a rule) in straight-line code: inside a loop the compiler hoists the chunk's invariants in
	 * front of it and parks them in scratch -- 80 bytes per lane written and read back by every workgroup: 31 MB a scan */
	if (n > 0)
		k2x_chunk<K2A_THREADS, 1, K2ParamsK &>(w, p, sc, p.surv_mode, p.surv_skip, base, (unsigned)p.surv_pch, n < K2X_NT ? n : K2X_NT);
	for (unsigned off = K2X_NT; off < n; off += K2X_NT) {
		/* (every further chunk reads the parameters afresh as well: what the compiler cannot hoist out of the loop it does not park in scratch) */
		K2ParamsK &q = k2_kernarg_again();
		k2x_chunk<K2A_THREADS, 1, K2ParamsK &>(w, q, sc, q.surv_mode, q.surv_skip, base + off, (unsigned)q.surv_pch, n - off < K2X_NT ? n - off : K2X_NT);
	}
}

/* What the scan workgroups' private areas did not hold lies in the list's common area (k2a_append: a stretch where far more
 * than 2.7 % pass -- a carrier, a run of sync words; a test build's handicaps): the one-workgroup-per-channel kernel that
 * follows the scan on its stream works it off before it looks at the scan's results (a kernel boundary lies between the
 * writes and these reads: no fence).  p.drain_* name the scan (set by the host on the consumer's launch: enqueue_scan_drain);
 * nothing to do as a rule: one word read. */
template <int NT> __device__ void k2x_drain(K2xWork &w, const K2Params &p, int sc)
{
	if (p.drain_slot < 0)
		return;
	const unsigned pch = (unsigned)p.drain_pch, priv = (unsigned)p.drain_nwg * pch;
	const unsigned stride = p.item_cap - priv;
	unsigned nc = p.ctl[CTL_NSURV0 + p.drain_slot * p.nstreams * VDL2_CS + sc];
	{
		const unsigned lim = ~p.ctl[CTL_NSURVLIM0 + p.drain_slot * p.nstreams * VDL2_CS + sc];
		nc = nc > lim ? lim : nc;
	}
	nc = nc > stride ? stride : nc;	/* (a group that did not fit reserved past the end and wrote nothing: k2a_append; what lies below the first
					 * such group is complete only if no group was refused -- and then the channel is flagged unusable anyway) */
	if (nc == 0)	/* (block-uniform) */
		return;
	if (threadIdx.x == 0)
		w.tabs = 0;
	for (unsigned off = 0; off < nc; off += K2X_NT)
		k2x_chunk<NT, 1, const K2Params &>(w, p, sc, p.drain_mode, p.drain_skip, priv * K2A_ITEM_WORDS + off, stride, nc - off < K2X_NT ? nc - off : K2X_NT);
	/* what the chunks found went into the channel's counters by atomics (performed in the L2) while this CU's L1 may hold the
	 * counters' lines from the reads above: drop them before the caller looks at the counters (rare path: the price is fine) */
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
	__syncthreads();
}

/* Filtered samples of the tile instants q0 (even) and q0 + 1, sub-phase taps mf[] (d8psk.c:219-228), for
 * the screens only: fused multiply-adds,
 * and the 18 or 19 samples the two share are read once, 16 bytes per lane and read.  A tap that does not
 * exist has mf[] = 0. */
template <int S> __device__ __forceinline__ void k2a_fir2(const K2aShared &sh, int q0, const float (&mf)[17], v2f &acc0, v2f &acc1)
{
	typedef float v4f __attribute__((ext_vector_type(4)));
	acc0 = (v2f){0.0f, 0.0f};
	acc1 = (v2f){0.0f, 0.0f};
	if (S == 4) {
		/* instant q, tap j = 4 m + t: element q + m of run t; the neighbouring instant q + 1 one element on (q0 is even: 16-byte reads) */
		v2f x[4][6];
#pragma unroll
		for (int t = 0; t < 4; ++t) {
			const v4f *pt = reinterpret_cast<const v4f *>(&sh.xs[t * K2A_XQ4 + q0]);
#pragma unroll
			for (int i = 0; i < 3; ++i) {
				const v4f e = pt[i];
				x[t][2 * i] = e.xy;
				x[t][2 * i + 1] = e.zw;
			}
		}
#pragma unroll
		for (int j = 0; j < 17; ++j) {
			acc0 = __builtin_elementwise_fma(x[j & 3][j >> 2], (v2f){mf[j], mf[j]}, acc0);
			acc1 = __builtin_elementwise_fma(x[j & 3][(j >> 2) + 1], (v2f){mf[j], mf[j]}, acc1);
		}
	} else if (S == 2) {
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

/* unit phasor of a filtered sample (for the screens): atan2f(0, 0) = 0; anything else odd becomes NaN and passes every screen */
__device__ __forceinline__ v2f k2a_unit(v2f a)
{
	const float n2 = __fmaf_rn(a.x, a.x, a.y * a.y);
	v2f w = a * __frsqrt_rn(n2);
	if (!(n2 >= 1e-30f && n2 <= 1e30f)) {
		const float bad = (a.x == 0.0f && a.y == 0.0f) ? 0.0f : __builtin_nanf("");
		w = (v2f){1.0f + bad, bad};
	}
	return w;
}

__device__ __forceinline__ float k2a_lane_from(int byte_addr, float v)
{
	return __int_as_float(__builtin_amdgcn_ds_bpermute(byte_addr, __float_as_int(v)));
}

#ifdef K2A_NOBAR
#define K2A_SYNC() __builtin_amdgcn_wave_barrier()
#else
#define K2A_SYNC() __syncthreads()
#endif
template <int S> __device__ void k2a_tile(K2aShared &sh, const K2Params &p, int sc, long long dec_base, long long nbase,
					   int cnt, unsigned rmask, int mode, long long chk_lo, long long chk_hi, int *fail,
					   K2aPre<S> &pre, long long next_nbase, int next_cnt, int skip_r = -1, int skip_par = 0)
{
	const int tid = threadIdx.x;
	constexpr int PH = K2A_POFF / S;	/* phase instants of history */
	constexpr int LSTR = 8 / S;		/* one symbol in instants */
	constexpr int E2 = 2 / S, E4 = 4 / S;	/* previous two evaluations in instants */
	/* filter pass: a lane takes a PAIR of neighbouring instants; the phasor one symbol earlier is KH lanes down in the same
	 * wavefront, whose first KH lanes repeat the pairs of the wavefront before (they only supply, they do not store) */
	constexpr int KH = LSTR / 2;
	constexpr int PPW = 64 - KH;				/* pairs a wavefront owns per round */
	constexpr int PPI = (K2A_THREADS / 64) * PPW;		/* ... the workgroup */
	constexpr int TSS = K2aTs<S>::v;	/* instants of a full tile */
	constexpr int NIT = ((TSS + PH + 1) / 2 + PPI - 1) / PPI;
	static_assert(K2A_POFF == S * PH, "phase history must be a whole number of instants");
	float2 *const wu = (S != 1) ? sh.xs : sh.xs + K2A_WU1;	/* phase-step phasors (see K2aShared) */
	/* exp(-j (SW[l] - SW[l-1])), l = 1..16: the template steps are 1,7,5,-7,1,3,-3,-7,3,-1,5,-5,-3,-5,-1,7 (x pi/8) */
#ifdef K2A_SCREEN_FMA
	constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f;
	constexpr float rc[16] = {C1, -C1, -S1, -C1, C1, S1, S1, -C1, S1, C1, -S1, -S1, S1, -S1, C1, -C1};
	constexpr float rs[16] = {-S1, -S1, -C1, S1, -S1, -C1, C1, S1, -C1, S1, -C1, C1, C1, C1, S1, -S1};
#endif
	const int nx = S * (cnt - 1) + 1 + K2A_XOFF;
	const bool prof = p.dbg && (mode >= 2 || (mode == 0 && S == 1 && !p.full_scan && !p.full_round)) && tid == 0 && (blockIdx.x & 7) == 0;	/* probe, region scan */
	long long tq = prof ? clock64() : 0;
#define K2A_STAMP(slot) do { if (prof) { const long long tn = clock64(); sh.prof[slot] += (unsigned long long)(tn - tq); tq = tn; } } while (0)
	/* park slots: sample i = tid + k * K2A_THREADS of the tile goes to xs[pbase + k * pstep] */
	const int pbase = (S == 4) ? (tid & 3) * K2A_XQ4 + (tid >> 2) : ((S == 2) ? (tid & 1) * K2A_XODD + (tid >> 1) : tid);
	constexpr int pstep = K2A_THREADS / S;
#if K2A_PREFETCH
	if (!pre.loaded)
		k2a_fetch<S>(pre, p, sc, dec_base, nbase, cnt);
	K2A_SYNC();
	K2A_STAMP(12);
#pragma unroll
	for (int k = 0; k < K2aPre<S>::NL; ++k)
		if (tid + k * K2A_THREADS < nx)
			sh.xs[pbase + k * pstep] = pre.v[k];
	pre.loaded = false;
#else
	{
		/* no software prefetch: six workgroups per CU are in different phases of their tiles, one's wait for its samples is
		 * another's filter pass (the twenty registers a prefetch holds through the screens are what six wavefronts per SIMD
		 * do not leave) */
		const float2 *x = p.dec + (size_t)sc * p.cap + (nbase - K2A_XOFF - dec_base);
		float2 v[K2aPre<S>::NL];
#pragma unroll
		for (int k = 0; k < K2aPre<S>::NL; ++k)
			if (tid + k * K2A_THREADS < nx)
				v[k] = x[tid + k * K2A_THREADS];
		K2A_STAMP(12);
#pragma unroll
		for (int k = 0; k < K2aPre<S>::NL; ++k)
			if (tid + k * K2A_THREADS < nx)
				sh.xs[pbase + k * pstep] = v[k];
	}
#endif
	pre.tiles++;
	K2A_STAMP(13);
	K2A_SYNC();
	K2A_STAMP(0);
	const int npairs = (cnt + PH + 1) / 2;
	const int wv = tid >> 6, ln = tid & 63;
	const int src_lane = (ln - KH) * 4;	/* ds_bpermute address of the lane KH down (wraps for the first KH lanes: their result is not used) */
#pragma unroll 1
	for (int r = 0; r < 4; ++r) {
		if (!(rmask & (1u << r)))
			continue;
		float mf[17];	/* wave-uniform: scalar registers */
#pragma unroll
		for (int j = 0; j < 17; ++j)	/* mflt[r + 64] exists only for r == 0 (16 taps otherwise) */
			mf[j] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(sh.smf[r + 4 * j])));
		/* ---- phase-step phasors u[q] = w[q] * conj(w[q - LSTR]) of instants -PH + LSTR .. cnt-1, w = unit phasor of the
		 *      filtered sample */
		float4 hold[NIT];
#pragma unroll
		for (int it = 0; it < NIT; ++it) {
			const int pw = it * PPI + wv * PPW;	/* first pair this wavefront owns in this round */
			if (pw >= npairs) {	/* (wave-uniform) */
				hold[it] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
				continue;
			}
			int q0 = 2 * (pw + ln - KH);
			q0 = q0 < 0 ? 0 : (q0 > TSS + PH - 2 ? TSS + PH - 2 : q0);	/* keeps the reads inside xs[]; what a clamped lane computes is not stored */
			v2f acc[2];
			k2a_fir2<S>(sh, q0, mf, acc[0], acc[1]);
			const v2f w0 = k2a_unit(acc[0]), w1 = k2a_unit(acc[1]);
			const float p0x = k2a_lane_from(src_lane, w0.x), p0y = k2a_lane_from(src_lane, w0.y);
			const float p1x = k2a_lane_from(src_lane, w1.x), p1y = k2a_lane_from(src_lane, w1.y);
			hold[it] = make_float4(__fmaf_rn(w0.x, p0x, w0.y * p0y), __fmaf_rn(w0.y, p0x, -(w0.x * p0y)),
					       __fmaf_rn(w1.x, p1x, w1.y * p1y), __fmaf_rn(w1.y, p1x, -(w1.x * p1y)));
			__builtin_amdgcn_sched_barrier(0);	/* one round's nineteen samples in registers at a time */
		}
		K2A_STAMP(1);
#if K2A_PREFETCH
		if (next_cnt > 0 && !pre.loaded)	/* the next tile's samples: in flight during this tile's screens */
			k2a_fetch<S>(pre, p, sc, dec_base, next_nbase, next_cnt);
#endif
		if (S != 1)
			K2A_SYNC();	/* wu[] is xs[]: every wavefront is through with the samples */
		K2A_STAMP(2);
#pragma unroll
		for (int it = 0; it < NIT; ++it) {
			const int q0 = 2 * (it * PPI + wv * PPW + ln - KH);
			if (ln >= KH && q0 >= LSTR && q0 < cnt + PH)	/* (the odd one out at the end lands in wu[]'s padding) */
				*reinterpret_cast<float4 *>(&wu[q0]) = hold[it];
		}
		K2A_STAMP(3);
		K2A_SYNC();
		/* ---- first screen, of the evaluation that is the `perr` of instant i: j = i + E2.  What passes (2.7 % on noise: seven of a
		 *      wavefront's 256 evaluations) is only NOTED here, two bytes in the wavefront's own queue; the items -- sixteen phasors
		 *      re-read from wu[], packed, five 16-byte stores, one LDS atomic -- are written behind the passes as one group per
		 *      wavefront, tile and sub-phase [until round 4 every pass in which anything passed, 83 % of them, ran the whole append:
		 *      a quarter of a tile's instructions].  The queue is the wavefront's own: LDS operations of one wavefront execute in
		 *      order, no barrier is needed between its writes and its reads. */
		unsigned npend = 0;	/* (wave-uniform) */
		for (int i0 = 0; i0 < cnt; i0 += K2A_THREADS) {	/* (wave-uniform trip count) */
			const int i = i0 + tid;
			const int j = (i < cnt ? i : cnt - 1) + E2;
			const float2 *uq = &wu[PH - E4 + j - 15 * LSTR];
			float2 uu[16];
#pragma unroll
			for (int l = 0; l < 16; ++l)	/* all sixteen LDS reads in flight before the arithmetic */
				uu[l] = uq[l * LSTR];
#ifdef K2A_SCREEN_FMA
			v2f acc = {0.0f, 0.0f}, acc1 = {0.0f, 0.0f};
#pragma unroll
			for (int l = 0; l < 16; l += 2) {
				acc = k2_rot(acc, (v2f){uu[l].x, uu[l].y}, rc[l], rs[l]);
				acc1 = k2_rot(acc1, (v2f){uu[l + 1].x, uu[l + 1].y}, rc[l + 1], rs[l + 1]);
			}
			acc += acc1;
#else
			/* The sixteen template rotations are e^(j pi/8) e^(j k pi/4) with k = 7,4,5,3,7,6,1,3,6,0,5,2,1,2,0,4: the common factor
			 * does not change |R|, even k are quarter turns (+-1, +-j) and odd k the same times e^(j pi/4) -- so the phasors are
			 * ADDED into four sums (+-1 and +-j, with and without the eighth turn) and turned once at the end: 16 packed additions
			 * and a dozen plain operations where the rotation by multiplication took 32 packed multiply-adds. */
			constexpr int kq[16] = {7, 4, 5, 3, 7, 6, 1, 3, 6, 0, 5, 2, 1, 2, 0, 4};
			v2f s4[4] = {{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}};	/* [0] +-1, [1] +-j, [2] +-e^(j pi/4), [3] +-j e^(j pi/4) */
#pragma unroll
			for (int l = 0; l < 16; ++l) {
				const v2f u = {uu[l].x, uu[l].y};
				const int g = ((kq[l] & 1) << 1) | ((kq[l] >> 1) & 1);
				if (kq[l] & 4)
					s4[g] -= u;
				else
					s4[g] += u;
			}
			const v2f pa = {s4[0].x - s4[1].y, s4[0].y + s4[1].x};	/* s4[0] + j s4[1] */
			const v2f qa = {s4[2].x - s4[3].y, s4[2].y + s4[3].x};
			constexpr float RH = 0.70710678118654752f;
			const v2f acc = {__fmaf_rn(qa.x - qa.y, RH, pa.x), __fmaf_rn(qa.x + qa.y, RH, pa.y)};
#endif
			const float r2 = __fmaf_rn(acc.x, acc.x, acc.y * acc.y);
			bool pass = i < cnt && !(r2 <= VDL2_SCREEN_R2);
			if (mode == 0 && r == skip_r && (int)((nbase + S * (j - E4)) & 1) == skip_par)
				pass = false;	/* region scan: that class is the probe's, its hits are in the table already */
			const unsigned long long m = __ballot(pass);
			if (m != 0) {
				if (pass)
					sh.pend[wv][npend + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] =
						(unsigned short)(i | (!(r2 == r2) ? 0x8000 : 0));
				npend += (unsigned)__popcll(m);
			}
		}
		for (unsigned b = 0; b < npend; b += 64) {	/* (one turn, unless a carrier or a run of sync words lies in the tile) */
			const bool on = b + (unsigned)ln < npend;
			const int e = on ? (int)sh.pend[wv][b + (unsigned)ln] : 0;
			const int j = (e & 0x3ff) + E2;
			const float2 *uq = &wu[PH - E4 + j - 15 * LSTR];
			float2 uu[16];
#pragma unroll
			for (int l = 0; l < 16; ++l)
				uu[l] = uq[l * LSTR];
			const long long n_abs = nbase + S * (j - E4);	/* the evaluation's instant */
			k2a_append(sh, p, sc, on, (int)(n_abs - dec_base), r, (mode == 1) ? (int)(chk_lo - dec_base) : 0,
				   (mode == 1) ? (int)(chk_hi - dec_base) : 0, uu, (e & 0x8000) != 0, mode, fail);
		}
		K2A_STAMP(4);
		K2A_STAMP(8);
		K2A_SYNC();
		K2A_STAMP(9);
		if (prof)
			sh.prof[11] += 1ull;
	}
#undef K2A_STAMP
}

__global__ __launch_bounds__(K2A_THREADS) __attribute__((amdgpu_waves_per_eu(K2A_WPE, 8)))
void k2a_probe(K2Params p)
{
	__shared__ K2aShared sh;
	const int c = blockIdx.y, s = (int)blockIdx.z;
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
		pre.tiles = 0;
		const long long step = (long long)gridDim.x * K2A_TS;
		for (long long n0 = p.scan_lo + (long long)blockIdx.x * K2A_TS; n0 < avail_end; n0 += step) {
			const int nt = (int)((avail_end - n0 < K2A_TS) ? (avail_end - n0) : K2A_TS);
			const long long n1 = n0 + step;
			const int nt1 = n1 < avail_end ? (int)((avail_end - n1 < K2A_TS) ? (avail_end - n1) : K2A_TS) : 0;
			k2a_tile<1>(sh, p, sc, dec_base, n0, nt, 0xfu, 0, 0, 0, nullptr, pre, n1, nt1);
		}
		k2a_tail(sh, sc);
		return;
	}
#if VDL2_PROBE_STRIDE == 2
	/* ONE class everywhere: sub-phase probe_r at the instants of scan_lo's parity -- any class finds the bursts; which
	 * stretches the chain relied on in OTHER classes is what the verify pass re-scans (round 4's probe: its class is a complete
	 * table, the region scan skips it and the verify pass the stretches the chain idles through in it) */
	K2aPre<2> pre;
	pre.loaded = false;
	pre.tiles = 0;
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
#else
	/* The probe FINDS the bursts: sub-phase 0 at every FOURTH sample from scan_lo on, and what it hands on are seeds -- where the
	 * fit error of that class dips below VDL2_SEED_ERR (k2a_emit, mode 3) --, around which the region scan looks at every class
	 * at every instant.  A burst fires the detector in all eight classes within a few samples and its screen sum stays within
	 * 0.5 of the maximum two samples off (scripts/dev/probe_rate.py), so half the instants of one class find it as well as all of
	 * them did: half the filter and screen work of round 4's probe, which looked at every second sample and whose class was a
	 * complete table in exchange -- now the region scan lists the probe's class too and the verify pass covers the stretches
	 * the chain idles through in it (an eighth more of either).  What the probe misses the verify pass finds, as ever. */
	K2aPre<4> pre;
	pre.loaded = false;
	pre.tiles = 0;
	const long long step = 4LL * gridDim.x * K2A_TS4;
	for (long long n0 = p.scan_lo + 4LL * blockIdx.x * K2A_TS4; n0 < avail_end; n0 += step) {
		const long long left = (avail_end - n0 + 3) / 4;
		const int nt = (int)(left < K2A_TS4 ? left : K2A_TS4);
		const long long n1 = n0 + step;
		const long long left1 = (avail_end - n1 + 3) / 4;
		const int nt1 = n1 < avail_end ? (int)(left1 < K2A_TS4 ? left1 : K2A_TS4) : 0;
		k2a_tile<4>(sh, p, sc, dec_base, n0, nt, 1u, 3, 0, 0, nullptr, pre, n1, nt1);
	}
#endif
	k2a_tail(sh, sc);
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
	unsigned long long tmp[VDL2_CAND_CAP];	/* (k2r_regions, k2s_sort: K2xWork lies here while the kernel drains a scan's common area) */
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

static_assert(sizeof(K2xWork) <= sizeof(unsigned long long) * VDL2_CAND_CAP, "K2xWork overlays WgSortShared.tmp");
/* ---- regions around the probe's hits (one workgroup per channel) */
#define K2R_NT 1024
__global__ __launch_bounds__(K2R_NT)
void k2r_regions(K2Params p)
{
	__shared__ unsigned long long key64[VDL2_CAND_CAP];
	__shared__ WgSortShared ws;
	__shared__ int key[VDL2_CAND_CAP];
	const int tid = threadIdx.x;
	const int c = blockIdx.x, s = (int)blockIdx.y;
	const int sc = s * VDL2_CS + c;
	/* the common area of the scan in front: the probe's (round 0), the previous round's verify pass's (complete round) */
	k2x_drain<K2R_NT>(*reinterpret_cast<K2xWork *>(ws.tmp), p, sc);
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

#ifndef K2A_WPE_REGION
#define K2A_WPE_REGION 5	/* the region scan filters a tile once per sub-phase and skips the probe's class: at 80 registers the compiler
				 * spilled four of them inside the tile loop (12 bytes of scratch); at 96 none */
#endif
__global__ __launch_bounds__(K2A_THREADS) __attribute__((amdgpu_waves_per_eu(K2A_WPE_REGION, 8)))
void k2a_region(K2Params p)
{
	__shared__ K2aShared sh;
	const int c = blockIdx.y, s = (int)blockIdx.z;
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
	pre.tiles = 0;
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
		sh.prof[12] += (unsigned long long)(t1 - t0);		/* all tiles */
	k2a_tail(sh, sc);
	if (p.dbg && threadIdx.x < 16 && sh.prof[threadIdx.x])
		atomicAdd(p.dbg + 48 + threadIdx.x, sh.prof[threadIdx.x]);
}

/* one workgroup = K2A_VRUN tiles of 2*K2A_TS samples; every piece of a verify segment inside a tile
 * is scanned in the segment's class */
#ifndef K2A_VRUN
#define K2A_VRUN 4
#endif
#define K2A_VITEMS 64
__global__ __launch_bounds__(K2A_THREADS) __attribute__((amdgpu_waves_per_eu(K2A_WPE, 8)))
void k2a_verify(K2Params p)
{
	__shared__ K2aShared sh;
	__shared__ int s_list[64], s_nl, s_ni;
	__shared__ int4 s_item[K2A_VITEMS];	/* lo, hi (stream-relative samples), sub-phase */
	const int tid = threadIdx.x;
	const int c = blockIdx.y, s = (int)blockIdx.z;
	const int sc = s * VDL2_CS + c;
	if (p.force_serial || p.full_scan || p.full_round)
		return;
	if (p.round > 0 && !p.redo[sc])
		return;
	const long long dec_base = p.dec_base;
	const int t_end = (int)(VDL2_CARRY_FRAMES + p.J);
	const int nseg = (int)p.ctl[CTL_NSEG0 + sc];
	const Seg *segs = p.segs + (size_t)sc * VDL2_SEG_CAP;
	if (nseg == 0 || (int)(p.cs[sc].pos - dec_base) + (int)blockIdx.x * K2A_VRUN * 2 * K2A_TS >= t_end)	/* (block-uniform) nothing to look at: not even the tables */
		return;
	k2a_tables(sh);
	K2aPre<2> pre;
	pre.loaded = false;
	pre.tiles = 0;
	/* a workgroup's runs of K2A_VRUN tiles: blockIdx.x, blockIdx.x + gridDim.x, ... -- the first pass's grid has a workgroup per run;
	 * a repair round's pass, which looks at the few stretches a local repair changed, is launched with a handful of workgroups per
	 * channel (enqueue_back: 2 800 workgroups that find nothing each wait for a slot beside the other pushes' wide kernels) */
	for (int run = (int)blockIdx.x;; run += (int)gridDim.x) {
		const int r_lo = (int)(p.cs[sc].pos - dec_base) + run * K2A_VRUN * 2 * K2A_TS;
		if (r_lo >= t_end)
			break;
		const int r_hi = r_lo + K2A_VRUN * 2 * K2A_TS < t_end ? r_lo + K2A_VRUN * 2 * K2A_TS : t_end;
		__syncthreads();	/* (the lists of the run before have been read) */
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
		if (nl == 0)	/* (block-uniform) */
			continue;
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
			break;
		}
		for (int q = 0; q < ni; ++q) {
			const int4 it = s_item[q];
			const int4 nx = (q + 1 < ni) ? s_item[q + 1] : make_int4(0, 0, 0, 0);
			k2a_tile<2>(sh, p, sc, dec_base, dec_base + it.x, (it.y - it.x + 1) / 2, 1u << it.z, 1, dec_base + it.x, dec_base + it.y,
				    p.fail + sc, pre, dec_base + nx.x, (nx.y - nx.x + 1) / 2);
		}
	}
	k2a_tail(sh, sc);
}

#endif
