/* vdl2gpu_machine.h -- the exact serial detector / burst state machine.  Part of the device side of libvdl2gpu.so; included by vdl2gpu_kernels.h only. */
#ifndef VDL2GPU_MACHINE_H
#define VDL2GPU_MACHINE_H

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
/* (inlined: as a call it costs the resolver and the gather kernel 8 % -- callee-saved registers through scratch) */
/* TAB (K2d): sph, lds_tabs and pn8 are all there -- a compile-time fact, so that the kernel does not carry the other paths' code and
 * their scalar-register appetite (with run-time tests the compiler loaded all of mflt[] into 65 scalar registers for the path K2d never
 * takes and spilled 367 of them: 11 900 lines of ISA, 5 500 without) */
template <int NT, bool TAB = false> __device__ __forceinline__ void burst_payload(vdl2gpu_burst_t *rec, const float2 *x0, const uint8_t *pn, long long nstar,
						  int clk0, float df, int nbrow, int nlbyte, int stream, ChanCfg cfg, float *sph_ = nullptr,
						  int tag = 1, int slot = 0, const float *lds_tabs_ = nullptr, const uint8_t *pn8_ = nullptr,
						  const float *lds_fir = nullptr)
{
	/* lds_tabs: mflt[72], the atanf range table AND the three soft-bit tables in LDS (K2d); lds_fir: only the first two (the serial
	 * machine's MachSharedT.smf / .atab are adjacent): the phases then come from k2_fir_phase_tab -- the same result bits -- and no
	 * kernel that decodes a payload holds mflt[] in scalar registers */
	float *const sph = sph_;
	const float *const lds_tabs = lds_tabs_;
	const uint8_t *const pn8 = pn8_;
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
	if (TAB || sph) {
		const int kmax = (25 + 8 * (g.ND + g.NF) - 1) / 3;
		/* lds_tabs (K2d): mflt[72], the atanf range table and the three soft-bit tables in LDS -- the lanes' table look-ups are
		 * LDS reads instead of dependent loads from constant memory, the atan2f has no branches (same result bits) */
		if (TAB || lds_tabs)
			for (int k = 7 + tid; k <= kmax; k += NT)
				sph[k - 7] = k2_fir_phase_tab(xs0 + 8LL * k, rb, lds_tabs, lds_tabs + 72);
		else if (!TAB)
			for (int k = 7 + tid; k <= kmax; k += NT)
				sph[k - 7] = k2_fir_phase_tab(xs0 + 8LL * k, rb, lds_fir, lds_fir + 72);
		__syncthreads();
	}
	for (int b = tid; b < g.ND + g.NF; b += NT) {
		const int q0 = 25 + 8 * b;
		const int k0 = q0 / 3;	/* >= 8: never needs P1 */
		int q = q0;
		unsigned byte = 0;
		const float *grey = (TAB || lds_tabs) ? lds_tabs + 72 + VDL2_ATAN_ROWS * VDL2_ATAN_STRIDE : nullptr;
		const unsigned pnb = (TAB || pn8) ? pn8[b] : 0u;	/* the byte's eight scrambler bits in one load (pn[25 + 8b + i] << i) */
		float pprev, pk;
		if (TAB || sph)
			pprev = sph[k0 - 8];
		else
			pprev = k2_fir_phase_tab(xs0 + 8LL * (k0 - 1), rb, lds_fir, lds_fir + 72);
		for (int k = k0; q < q0 + 8; ++k) {
			if (TAB || sph)
				pk = sph[k - 7];
			else
				pk = k2_fir_phase_tab(xs0 + 8LL * k, rb, lds_fir, lds_fir + 72);
			const int idx = k2_grey_index(pk, pprev, df);
			pprev = pk;
			for (int i = q - 3 * k; i < 3 && q < q0 + 8; ++i, ++q) {
				const float v = k2_soft_bit(idx, i, (TAB || pn8) ? (int)((pnb >> (q - q0)) & 1u) : (int)pn[q], grey);
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
		rec->trig_sample = tag;	/* device-side use of the two host-filled fields: 0 = decoded by K2d from the resolver's
					 * selection (the host drops these for a channel that K2f then redid), 1 = final */
		rec->end_sample = slot;	/* stream * 8 + channel slot */
	}
}

template <int NT> struct MachSharedT {
	float pbuf[VDL2_NPH + NT];	/* phases: [0,68) = history ring */
	float errs[NT + 2];		/* errs[t+2] = err of eval t; [0],[1] = p2err, perr */
	float frs[NT + 1];		/* frs[t+1] = slope of eval t; [0] = pfr */
	float psym[12];			/* header symbol phases */
	float2 xt[VDL2_XT];		/* LDS tile of the channel's samples (cluster mode) */
	float smf[72];			/* low-pass taps mflt[] (d8psk.h:28-45), zero padded */
	float atab[VDL2_ATAN_ROWS * VDL2_ATAN_STRIDE];	/* atanf range constants: the table-driven atan2f (vdl2_math.h) has no data-dependent branches */
	float hsoft[25];		/* descrambled header soft bits */
	uint8_t vbk[26][32], vbv[26][32];	/* Viterbi back pointers / decided bits */
	int first;
	int ctl[16];
	float fctl[8];
};

static_assert(offsetof(MachSharedT<64>, atab) == offsetof(MachSharedT<64>, smf) + 72 * sizeof(float), "burst_payload's lds_fir: smf[72] with atab[] right behind it");
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
	HeadTap *headtap;	/* diagnostics, see K2Params */
	unsigned *headtap_n;
	unsigned headtap_cap;
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
	int nslots, badslot;	/* descriptors made; one of them found the pool full (an array of the slots became a scratch object: only its sign was ever read) */
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
	for (int i = threadIdx.x; i < VDL2_ATAN_ROWS * VDL2_ATAN_STRIDE; i += NT)
		sh.atab[i] = vdl2_atan_tab_entry(i);
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
#ifdef VDL2_MACH_ATAN_BRANCHY
	return vdl2_atan2f(si, sr);
#else
	return vdl2_atan2f_tab(si, sr, sh.atab);	/* the same result bits (tests/test_math.py, test_gpu_math.py), no divergence between the lanes */
#endif
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

/* What a sync trigger at stream time nstar sets in motion (d8psk.c:292-306, 77-116): one-shot timing
 * estimate from the three fit errors around the minimum, carrier estimate df = previous slope, the nine
 * header symbols, the (25,20) Viterbi decode, burst geometry.  Needs no detector history beyond those
 * four numbers -- which is what lets K2b take them from the scan's candidate record. */
struct MachTrig {
	long long nsym0;	/* stream time of burst symbol 0 */
	float df;
	int clk0, rb;		/* (int)roundf(of); FIR sub-phase of the burst */
	int accepted, nbrow, nlbyte, nsym;
	bool defer;		/* header or burst not completely inside the data held */
};

/* ring_r >= 0 (K2b): the caller is rebuilding the 68-phase ring in front of the trigger in sub-phase ring_r and has
 * filled sh.pbuf[0..63]; the last four phases (instants nstar-6 .. nstar) are computed here, in the same filter pass as
 * the header symbols (a pass of its own for four lanes costs a wavefront as much as one for sixty-four). */
template <int NT, bool XL> __device__ __forceinline__ MachTrig mach_trigger(MachSharedT<NT> &sh, MachCtx &cx, long long nstar,
							     float p2err, float perr, float err, float pfr, int ring_r = -1)
{
	const int tid = threadIdx.x;
	/* parabolic interpolation of the error minimum, d8psk.c:303-305 -- the same four numbers in every lane, so every
	 * lane computes it (through LDS it was a store, a barrier and four loads) */
	const float of = 4.0f * (p2err - 4.0f * perr + 3.0f * err) / (p2err - 2.0f * perr + err);
	int clk0 = (int)roundf(of);
	if (clk0 < 0)
		clk0 = 0;	/* unreachable for finite inputs: of is in [4,12] */
	if (clk0 > 68)
		clk0 = 68;
	int j0, rb;
	burst_timing(clk0, &j0, &rb);
	const float df = pfr;	/* df = pfr, d8psk.c:301 */
	const long long nsym0 = nstar + j0;	/* stream time of burst symbol 0 */
	bool defer = (nsym0 + 64 >= cx.avail_end);	/* 9 header symbols must be present */
	int accepted = 0, nbrow = 0, nlbyte = 0, nsym = 0;
	if (!defer) {
		mach_need<NT, XL>(sh, cx, nstar - (ring_r >= 0 ? 24 : 16), nsym0 + 65);
		/* ONE filter pass: header symbols 0..8 (sub-phase rb), P1 = the trigger instant filtered with clk0 as first
		 * tap (d8psk.c:306), and K2b's four ring phases (as separate passes each costs the wavefront a whole FIR + atan2f) */
		if (tid < (ring_r >= 0 ? 14 : 10)) {
			const long long n = tid < 9 ? nsym0 + 8 * tid : (tid == 9 ? nstar : nstar - 2LL * (13 - tid));
			const int tap = tid < 9 ? rb : (tid == 9 ? clk0 : ring_r);
			const float ph = mach_fir<NT, XL>(sh, cx, n, tap);
			if (tid < 9)
				sh.psym[tid] = ph;
			else if (tid == 9)
				sh.fctl[1] = ph;	/* P1 */
			else
				sh.pbuf[VDL2_NPH - 14 + tid] = ph;	/* ring entries 64..67 */
		}
		__syncthreads();
		if (tid < 64) {	/* the first wavefront */
			float v = 0.0f;
			if (tid < 25) {
				const int k = tid / 3;
				const float pprev = k ? sh.psym[k - 1] : sh.fctl[1];
				const int idx = k2_grey_index(sh.psym[k], pprev, df);
				v = mach_soft_bit(cx, idx, tid % 3, (int)((VDL2_PN_HEAD >> tid) & 1u));
				if (tid < 3)
					v = 0.0f;	/* reserved bits forced, d8psk.c:81-82 */
				sh.hsoft[tid] = v;
			}
			if (cx.headtap) {	/* diagnostics: uniform, off in production */
				unsigned k = 0;
				if (tid == 0)
					k = atomicAdd(cx.headtap_n, 1u);
				k = (unsigned)__shfl((int)k, 0, 64);
				if (k < cx.headtap_cap) {
					HeadTap *e = cx.headtap + k;
					if (tid < 25)
						e->soft[tid] = v;
					if (tid == 0) {
						e->nstar = nstar;
						e->sc = cx.sc;
						e->clk0 = clk0;
						e->p2err = p2err;
						e->perr = perr;
						e->err = err;
						e->pfr = pfr;
					}
				}
			}
			/* The (25,20) decode (viterbi.c:46-96) is a max-product search over the codewords: metric = product of
			 * v (bit 1) or 1 - v (bit 0) per position, in doubles.  If the hard decisions b_n = (v_n > 0.5) form a
			 * codeword (syndrome 0: the path ends in state 0) and every v_n is at least 1e-3 away from 0.5, that word
			 * is the decoder's output and the trellis need not be run: its metric takes the larger factor at every
			 * position, any other path the smaller one somewhere, which is >= 0.4 % less -- eleven orders of magnitude
			 * above what 25 roundings of a double product can move; products are monotone under rounding, so its
			 * prefix wins every compare-select on the way (strictly: the reference's `>` never sees a tie), and no
			 * metric underflows (v in [2e-6, 1-2e-6], d8psk.h).  Forced bits (v = 0) decide 0 in both.
			 * Else: the reference's trellis, below. */
			const bool one = tid < 25 && v > 0.5f;
			const bool unsure = tid >= 3 && tid < 25 && fabsf(v - 0.5f) < 1e-3f;
			const unsigned long long hb = __ballot(one);
			const int hc = one ? c_hcol[tid < 25 ? tid : 0] : 0;
			unsigned syn = 0;
#pragma unroll
			for (int k = 0; k < 5; ++k)
				syn |= (unsigned)(__popcll(__ballot((hc >> k) & 1)) & 1) << k;
			const bool quick = (syn == 0) && (__ballot(unsure) == 0ull);
			if (tid == 0) {
				sh.ctl[7] = quick ? 1 : 0;
				sh.ctl[8] = (int)((hb >> 3) & 0x1ffffu);	/* len: header bit 3 + i is bit i of the length (d8psk.c:90-93 undo exactly that order) */
			}
		}
		__syncthreads();
		if (!sh.ctl[7]) {	/* uniform */
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
				sh.ctl[8] = (int)(__brev(word & 0x1ffffu) >> 15);	/* reversebits(.,17) */
			}
			__syncthreads();
		}
		{
			const unsigned len = (unsigned)sh.ctl[8];
			nbrow = (int)(len / 1992u) + 1;
			nlbyte = (int)((len % 1992u + 7u) / 8u);
			accepted = (len >= 96u && nbrow <= 8) ? 1 : 0;
		}
		if (accepted) {
			nsym = burst_geom(nbrow, nlbyte).nsym;
			if (nsym0 + 8LL * (nsym - 1) >= cx.avail_end)
				defer = true;
		}
	}
	MachTrig tg;
	tg.nsym0 = nsym0;
	tg.df = df;
	tg.clk0 = clk0;
	tg.rb = rb;
	tg.accepted = accepted;
	tg.nbrow = nbrow;
	tg.nlbyte = nlbyte;
	tg.nsym = nsym;
	tg.defer = defer;
	return tg;
}

/* What a trigger that is not deferred leaves behind: counters and, for an accepted header, the burst's
 * descriptor (or its record, payload decoded at once).  Returns the stream time of the burst's last
 * symbol (of header symbol 8 for a rejected one): the idle search resumes two samples later. */
template <int NT, bool XL> __device__ __forceinline__ long long mach_commit_trigger(MachSharedT<NT> &sh, MachCtx &cx, const float2 *x0,
										       long long nstar, const MachTrig &tg, MachOut &out)
{
	const int tid = threadIdx.x;
	const int clk0 = tg.clk0, accepted = tg.accepted, nbrow = tg.nbrow, nlbyte = tg.nlbyte, nsym = tg.nsym;
	const float df = tg.df;
	const long long nsym0 = tg.nsym0;
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
			burst_payload<NT>(cx.recs + slot, x0, cx.pn, nstar, clk0, df, nbrow, nlbyte, cx.stream, cx.cfg, nullptr, 1, 0, nullptr, nullptr, sh.smf);
		if (slot == 0xffffffffu)
			out.badslot = 1;
		out.nslots++;
		out.nburst++;
	}
	return nlast;
}

/* stop_steady: return MR_STEADY as soon as the detector is history-free and at least
 * `min_trig` triggers were handled.  first_nev: size of the first search window (a hint). */
template <int NT, bool XL> __device__ __forceinline__ int machine_run(MachSharedT<NT> &sh, MachCtx &cx, MachState &st, bool stop_steady,
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
		const MachTrig tg = mach_trigger<NT, XL>(sh, cx, nstar, sh.errs[ts], sh.errs[ts + 1], sh.errs[ts + 2], sh.frs[ts]);
		const int rb = tg.rb;
		if (tg.defer) {
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
		const long long nlast = mach_commit_trigger<NT, XL>(sh, cx, x0, nstar, tg, out);
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
	cx.x = p.dec + ((size_t)s * VDL2_CS + c) * p.cap;
	cx.dec_base = p.dec_base;
	cx.avail_end = p.dec_base + VDL2_CARRY_FRAMES + p.J;
	cx.pn = p.pn;
	cx.sc = s * VDL2_CS + c;
	cx.dbg = to_stage ? p.dbg : nullptr;
	cx.headtap = p.headtap;
	cx.headtap_n = p.headtap_n;
	cx.headtap_cap = p.headtap_cap;
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

#endif
