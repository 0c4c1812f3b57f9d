/* vdl2gpu_resolve.h -- K2s, K2b, K2c, K2f, K2d: candidate order, clusters, the real chain, commit, payload.  Part of the device side of libvdl2gpu.so; included by vdl2gpu_kernels.h only. */
#ifndef VDL2GPU_RESOLVE_H
#define VDL2GPU_RESOLVE_H

/* ====================================================================== K2s
 * Per channel: sort the candidates by time (bitonic network in LDS) for the resolver, and pick the
 * ones whose cluster is worth precomputing: the first of its (sub-phase, parity) class within a
 * burst's worth of samples.  A later candidate of the same class can only be reached if the detector
 * turns history-free in the few samples between the two; the resolver computes such a cluster itself
 * when it ever needs one (status CL_INVALID), so this is a cost decision, never a correctness one.
 */
#define K2S_NT 1024
#define K2S_MERGE 256		/* candidates a repair round merges into the sorted table instead of sorting again */
#define K2S_LOOKBACK 72		/* a triggered detector is busy for at least 9 symbols = 72 samples */
__global__ __launch_bounds__(K2S_NT)
void k2s_sort(K2Params p)
{
	__shared__ unsigned long long sbuf[VDL2_CAND_CAP];
	__shared__ WgSortShared ws;
	__shared__ int s_np;
	const int tid = threadIdx.x;
	const int c = blockIdx.x, s = (int)blockIdx.y;
	const int sc = s * VDL2_CS + c;
	if (p.force_serial)
		return;
	k2x_drain<K2S_NT>(*reinterpret_cast<K2xWork *>(ws.tmp), p, sc);	/* the common area of the scan in front (region scan; a complete scan) */
	if (p.round > 0 && p.fail[sc] >= VDL2_VERIFIED)
		return;
	int ncand = (int)p.ctl[CTL_CAND0 + sc];
	if (ncand > VDL2_CAND_CAP || p.ctl[CTL_CAND0 + p.nstreams * VDL2_CS + sc] != 0)
		return;		/* tables unusable: the resolver runs serially */
	const Cand *cands = p.cands + (size_t)sc * VDL2_CAND_CAP;
	int *skey = p.skey + (size_t)sc * VDL2_CAND_CAP;
	unsigned short *sidx = p.sidx + (size_t)sc * VDL2_CAND_CAP;
	unsigned short *prim = p.prim + (size_t)sc * VDL2_CAND_CAP;
	for (int i = tid; i < ncand; i += K2S_NT)
		sbuf[i] = (((unsigned long long)(unsigned)(cands[i].nrel * 4 + cands[i].r)) << 16) | (unsigned)i;
	if (tid == 0)
		s_np = 0;
	__syncthreads();
	wg_sort_u64<K2S_NT>(sbuf, ws, ncand, 18, (unsigned)(VDL2_CARRY_FRAMES + p.J));
	/* repair round: candidates are only ever appended and a cluster depends on nothing but its own
	 * candidate and the samples, so the clusters of the earlier rounds stand -- only primaries that are new
	 * (or were not primary before) go to K2b */
	const int nold = (p.round > 0) ? (int)p.ctl[CTL_NCLUST0 + sc] : 0;
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
#ifdef VDL2GPU_TESTHOOKS
		if (p.prim_drop > 0 && j % p.prim_drop == p.prim_drop - 1)
			primary = false;
#endif
		int2 *head = p.clhead + (size_t)sc * VDL2_CAND_CAP + idx;
		if (!primary)
			*head = cl_pack(0, CL_INVALID, 0, 0, 0, 0, 0);
		else if (idx >= nold || (head->y & 3) == CL_INVALID)
			prim[atomicAdd(&s_np, 1)] = (unsigned short)idx;
	}
	__syncthreads();
	if (tid == 0) {
		p.ctl[CTL_NPRIM0 + sc] = (unsigned)s_np;
		p.ctl[CTL_NCLUST0 + sc] = (unsigned)ncand;
	}
}

/* K2s for a repair round that only re-resolves (K2Params.mini_round): the table is sorted but for the handful of candidates
 * the verify pass appended -- merge them in.  Every old entry moves up by the number of new keys below it, every new one goes
 * where a binary search of the old list plus its rank among the new ones puts it (keys are unique: what the verify pass lists
 * was not in the table).  Nothing else changes: the old clusters stand, the new candidates have none (their heads say so), no
 * primaries are chosen.  A kernel of its own because of its footprint: 2 KB of LDS where k2s_sort has 84 -- one workgroup per
 * channel that wants half a CU's LDS waits for the other stage's wide kernel to drain before it gets it. */
#define K2M_NT 256	/* four wavefronts of 40 registers fit beside anything */
__global__ __launch_bounds__(K2M_NT)
void k2s_merge(K2Params p)
{
	__shared__ unsigned long long knew[K2S_MERGE];
	__shared__ K2xWork xw;
	const int tid = threadIdx.x;
	const int c = blockIdx.x, s = (int)blockIdx.y;
	const int sc = s * VDL2_CS + c;
	if (p.force_serial)
		return;
	k2x_drain<K2M_NT>(xw, p, sc);	/* the common area of the verify pass in front: what it finds there fails the channel like any other hit */
	if (p.fail[sc] >= VDL2_VERIFIED)
		return;
	const int ncand = (int)p.ctl[CTL_CAND0 + sc];
	unsigned *ovf = p.ctl + CTL_CAND0 + p.nstreams * VDL2_CS + sc;
	if (ncand > VDL2_CAND_CAP || *ovf != 0)
		return;		/* tables unusable: the resolver runs serially */
	const int nold = (int)p.ctl[CTL_NCLUST0 + sc], nnew = ncand - nold;
	if (nnew <= 0)
		return;		/* (a channel that failed for another reason than an unlisted event: the resolver will find nothing new) */
	if (nnew > K2S_MERGE) {	/* a handicapped test build, a pathological input: not worth a sort kernel's footprint in every push */
		if (tid == 0)
			atomicOr(ovf, 2u);	/* the resolver takes the channel through the serial machine (a complete round resets this); bit 1, not
						 * bit 0: the tables were not full, the host must not shorten its parts for this (k3_rebase) */
		return;
	}
	const Cand *cands = p.cands + (size_t)sc * VDL2_CAND_CAP;
	int *skey = p.skey + (size_t)sc * VDL2_CAND_CAP;
	unsigned short *sidx = p.sidx + (size_t)sc * VDL2_CAND_CAP;
	if (tid < nnew)
		knew[tid] = (((unsigned long long)(unsigned)(cands[nold + tid].nrel * 4 + cands[nold + tid].r)) << 16) | (unsigned)(nold + tid);
	__syncthreads();
	/* everything is read before anything is written: a thread's share of the old list into registers, the new keys' places by
	 * binary search of the old list */
	constexpr int PER = VDL2_CAND_CAP / K2M_NT;
	unsigned long long v[PER];
	int up[PER];
#pragma unroll
	for (int k = 0; k < PER; ++k) {
		const int j = tid + k * K2M_NT;
		v[k] = j < nold ? (((unsigned long long)(unsigned)skey[j] << 16) | sidx[j]) : ~0ull;
	}
#pragma unroll
	for (int k = 0; k < PER; ++k) {
		up[k] = 0;
		for (int i = 0; i < nnew; ++i)
			up[k] += (knew[i] < v[k]) ? 1 : 0;
	}
	int at = 0;
	unsigned long long vn = 0;
	if (tid < nnew) {
		vn = knew[tid];
		const int kn = (int)(vn >> 16);
		int a = 0, b = nold;	/* old entries below vn */
		while (a < b) {
			const int m = (a + b) >> 1;
			if (skey[m] < kn)
				a = m + 1;
			else
				b = m;
		}
		at = a;
		for (int i = 0; i < nnew; ++i)
			at += (knew[i] < vn) ? 1 : 0;
	}
	__syncthreads();
#pragma unroll
	for (int k = 0; k < PER; ++k) {
		const int j = tid + k * K2M_NT;
		if (j < nold) {
			skey[j + up[k]] = (int)(v[k] >> 16);
			sidx[j + up[k]] = (unsigned short)(v[k] & 0xffffu);
		}
	}
	if (tid < nnew) {
		skey[at] = (int)(vn >> 16);
		sidx[at] = (unsigned short)(vn & 0xffffu);
	}
	if (tid == 0) {
		p.ctl[CTL_NPRIM0 + sc] = 0u;
		p.ctl[CTL_NCLUST0 + sc] = (unsigned)ncand;
	}
}
static_assert(VDL2_CAND_CAP % K2M_NT == 0 && K2S_MERGE <= K2M_NT, "k2s_merge");

/* ====================================================================== K2b
 * One wavefront per primary trigger candidate (persistent workgroups pull tickets):
 * handle the trigger from what the candidate record says the detector saw, rebuild
 * the phase ring it returns to, run the exact machine through the 68 evaluations
 * behind the burst (and any burst that follows before the detector is history-free
 * again) and record where and how the idle search resumes.
 */
#ifndef K2B_WAVES
#define K2B_WAVES 3	/* 163 registers: with 4 (128) the compiler spilled 12-20 of them to scratch; the kernel's time does not depend on 3 / 4 / 6 wavefronts per SIMD */
#endif
#ifndef K2B_GRIDW
#define K2B_GRIDW K2B_WAVES	/* wavefronts per SIMD the cluster kernel's (persistent) grid asks for */
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
	int ord = 0;	/* (development counters: the wavefront's n-th cluster) */
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
		out.badslot = 0;
		/* The candidate record holds what the detector saw when it fired -- the three fit errors and
		 * the slope (the scan computed them with the detector's own arithmetic) --, so the trigger is
		 * handled from those; what has to be rebuilt is the phase ring the detector returns to after
		 * the burst: the phases of the 68 evaluations up to and including this one. */
		const long long t0 = wall_clock64();
		const long long nstar = st.pos;
		mach_need<K2B_NT, true>(sh, cx, nstar - 152, nstar + 1);
		static_assert(K2B_NT == 64 && VDL2_NPH == 68, "ring rebuild: 64 phases here, the last 4 in mach_trigger's filter pass");
		sh.pbuf[tid] = mach_fir<K2B_NT, true>(sh, cx, nstar - 2LL * (VDL2_NPH - 1 - tid), st.r);
		if (tid == 0) {
			sh.errs[0] = 500.0f;	/* errors re-armed (d8psk.c:308) */
			sh.errs[1] = 500.0f;
			sh.frs[0] = cd.pfr;
		}
		__syncthreads();
		const long long t1 = wall_clock64();
		int rc;
		{
			const MachTrig tg = mach_trigger<K2B_NT, true>(sh, cx, nstar, cd.p2err, cd.perr, cd.err, cd.pfr, st.r);
			if (tg.defer) {
				out.ndefer++;
				rc = MR_DEFER;
			} else {
				const long long nlast = mach_commit_trigger<K2B_NT, true>(sh, cx, cx.x - cx.dec_base, nstar, tg, out);
				out.neval += 1;
				st.pos = nlast + 2;
				st.r = tg.rb;
				st.fresh = 0;
				rc = machine_run<K2B_NT, true>(sh, cx, st, true, 1, VDL2_CL_MAXB, 0, out);
			}
		}
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
		if (tid == 0 && p.dbg) {	/* is a wavefront's first cluster slower than its later ones (code fetched cold)? */
			const int o = ord < 3 ? ord : 3;
			atomicAdd(p.dbg + 8 + o, (unsigned long long)(t2 - t0));
			atomicAdd(p.dbg + 12 + o, 1ull);
		}
		++ord;
		int status;
		if (rc == MR_STEADY)
			status = CL_STEADY;
		else if (rc == MR_DEFER && out.ntrig == 0)
			status = CL_DEFER_FIRST;
		else
			status = CL_NONSTEADY;
		if (out.badslot)	/* descriptor pool full */
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

/* first sorted candidate at/after stream-relative time `want` of class (r, parity of want): a binary
 * search in that class's own list (ranks in time order, ascending), -1 if there is none.  (Scanning
 * the time-sorted list for the next entry of the class costs a candidate of a rare class a walk to the
 * end of the list, one LDS round trip per step.) */
__device__ __forceinline__ int k2c_next(const int *skey, const unsigned short *clist, const int *coff, int want, int r)
{
	const int cls = r * 2 + (want & 1);
	int lo = coff[cls], hi = coff[cls + 1];
	const int end = hi;
	while (lo < hi) {
		const int mid = (lo + hi) >> 1;
		if ((skey[clist[mid]] >> 2) < want)	/* (the class list holds ranks; their times are the sorted keys': four bytes a candidate less in LDS) */
			lo = mid + 1;
		else
			hi = mid;
	}
	return lo < end ? (int)clist[lo] : -1;
}

/* one hop of the walk in 16 bits: rank of the successor (13 bits) | its cluster status << 13, 0xffff = none */
#define K2C_RANK 0x1fffu
#define K2C_RBITS 13
#define K2C_HOP_NONE 0xffffu
#define K2C_VIS 4096	/* entries of the visited list; more than that (it would take ten thousand bursts in a
			 * channel's push) fails the channel over to the serial redo */
/* sjump[j], for a steady cluster j: bits 0-12 = the last steady cluster within four hops of j (j itself if
 * the first hop is not steady); K2C_J_CONT: all four hops were steady, go on from there; otherwise the
 * chain ends behind it (no successor) or, K2C_J_SPECIAL, at the non-steady cluster in bits 16-28 */
#define K2C_J_CONT 0x80000000u
#define K2C_J_SPECIAL 0x40000000u
static_assert(VDL2_CAND_CAP <= 8191, "a hop holds a 13-bit rank");

__global__ __launch_bounds__(K2_NT)
void k2c_resolve(K2Params p)
{
	__shared__ MachSharedT<K2_NT> sh;
	__shared__ int skey[VDL2_CAND_CAP];		/* sorted keys: nrel*4 + r */
	/* (sorted rank -> candidate index is read from device memory where it is needed -- publishing: a lane each --, and the rank of the
	 * candidate that follows a cluster is the first hop of swalk[]: 23 bytes of LDS per candidate, the tables hold 6144) */
	__shared__ uint8_t sstat[VDL2_CAND_CAP];	/* cluster status | sub-phase the idle search resumes in << 2 (bits 0-3 of the cluster's head) */
	__shared__ unsigned short svis[K2C_VIS];	/* what the real chain visited: rank | 0x8000 = that cluster; rank = the steady
							 * hops of swalk[rank] */
	__shared__ unsigned sjump[VDL2_CAND_CAP];	/* the walk's view of swalk[]: see K2C_J_* */
	__shared__ int sx[VDL2_CAND_CAP];		/* where the idle search resumes behind every candidate's cluster (cl_pack().x), by sorted rank; the
							 * rest of the head is read from device memory where it is needed (publishing: every lane its own) --
							 * together with the notes at the top: 23 bytes of LDS per candidate instead of 35 */
	__shared__ int s_walk[4];
	__shared__ int s_cnt[4];
	__shared__ unsigned short clist[VDL2_CAND_CAP];	/* ranks of the candidates, class by class, time order within */
	__shared__ unsigned long long swalk[VDL2_CAND_CAP];	/* the next four hops from a steady cluster: what the walk reads */
	unsigned short *const sw16 = reinterpret_cast<unsigned short *>(swalk);	/* (hop i of rank j: sw16[4 j + i]) */
	__shared__ int coff[9];				/* where each class (r * 2 + time parity) starts in clist */
	__shared__ int s_wcnt[K2_NT / 64][8];
#ifndef K2C_PRIO
#define K2C_PRIO 3
#endif
	__builtin_amdgcn_s_setprio(K2C_PRIO);	/* one workgroup per channel beside the channeliser's thousands of waves */
	const int tid = threadIdx.x;
	const int c = blockIdx.x, s = (int)blockIdx.y;
	const int sc = s * VDL2_CS + c;
	int seg_from = -0x7fffffff;	/* repair round: stretches that end at or before the earliest event the verify pass found lie on the unchanged
					 * part of the chain and have been verified -- only what lies behind is listed again */
	if (p.round > 0) {
		const int fail_in = p.fail[sc];
		if (fail_in >= VDL2_VERIFIED)
			return;		/* verified in the first pass: nothing to repair */
		seg_from = fail_in;
		__syncthreads();
		if (tid == 0) {
			p.redo[sc] = 1;
			atomicAdd(p.outc_total_redo + 1, 1u);	/* channel-pushes that went through a repair round */
			if (p.dbg)
				atomicAdd(p.dbg + 24, 1ull);
			p.fail[sc] = 0x7f7f7f7f;	/* the repair pass is verified afresh */
			p.ctl[CTL_NSEL1 + sc] = 0;	/* (the repair rounds' own selection: see CTL_NSEL1) */
			p.ctl[CTL_NSEG0 + sc] = 0;
			p.ctl[CTL_NWIN0 + sc] = VDL2_WIN_ALL;	/* resolved again from the input state: nothing of the first selection stands */
			/* if K2d ran ahead on the first pass's selection, what it made of this channel is void (K2d's second pass tags
			 * those records 2: the host and K4 drop them) and K2d decodes the repaired selection in that second pass */
			if (p.fmask && sc < 512)
				atomicOr(p.fmask + (sc >> 5), 1u << (sc & 31));
		}
		__syncthreads();
	}
	const ChanState *cs = p.cs + sc;	/* input state: left untouched until K2f commits */
	ChanState *cs_out = p.cs_out + sc;
	unsigned *sel = (p.round > 0 ? p.sel_list2 : p.sel_list) + (size_t)sc * VDL2_SEL_CAP;
	unsigned *nsel = p.ctl + (p.round > 0 ? CTL_NSEL1 : CTL_NSEL0) + sc;
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
	const int r_probe = p.probe_r, par_probe = p.probe_par;	/* the probe scanned class (r_probe, instants of parity par_probe) everywhere */
	const int t_end = (int)(cx.avail_end - cx.dec_base);
	const bool lazy = !p.full_scan && !p.full_round;	/* (a full round's tables hold every class: nothing is left to verify) */
	mach_init_taps(sh);
	mach_load(sh, cs);
	MachOut out;
	out.nslots = out.ntrig = out.nrej = out.nburst = out.ndefer = 0;
	out.badslot = 0;
	out.neval = 0;
	unsigned long long n_slow = 0;
	int ncand = (int)p.ctl[CTL_CAND0 + sc];
	const bool tables_ok = !p.force_serial && ncand <= VDL2_CAND_CAP && p.ctl[CTL_CAND0 + p.nstreams * VDL2_CS + sc] == 0;
	if (!tables_ok)
		ncand = 0;
	const Cluster *clusters = p.clusters + (size_t)sc * VDL2_CAND_CAP;
	/* What a local repair (k2p_patch) needs of this pass: which candidates the chain met in the history-free state (a repaired
	 * stretch that arrives at one of them has rejoined this chain), and what the serial stretches counted (no cluster head has it). */
	uint8_t *const onchain = p.onchain + (size_t)sc * VDL2_CAND_CAP;
	K2Slog *const slog = p.slog + (size_t)sc * VDL2_SLOG_CAP;
	int nslog = 0;
	const bool first_pass = p.round == 0;
	if (first_pass)
		for (int i = tid; i < (ncand + 3) / 4; i += K2_NT)
			reinterpret_cast<uint32_t *>(onchain)[i] = 0u;
	/* 1. candidates sorted by time (K2s) */
	const long long pos_in = st.pos;
	const long long tk0 = wall_clock64();
	{
		const int2 *head = p.clhead + (size_t)sc * VDL2_CAND_CAP;
		for (int i = tid; i < ncand; i += K2_NT) {
			const int idx = p.sidx[(size_t)sc * VDL2_CAND_CAP + i];
			skey[i] = p.skey[(size_t)sc * VDL2_CAND_CAP + i];
			const int2 hd = head[idx];
			sx[i] = hd.x;
			sstat[i] = (uint8_t)(hd.y & 15);
		}
	}
	__syncthreads();
	/* 1b. the class lists, a counting sort that keeps the time order: every thread counts the classes in
	 *     its own contiguous stretch of the sorted list, an exclusive scan over (class, thread) turns the
	 *     counts into positions, and the thread places its stretch */
	{
		const int per = (ncand + K2_NT - 1) / K2_NT;
		const int j0 = tid * per < ncand ? tid * per : ncand, j1 = j0 + per < ncand ? j0 + per : ncand;
		int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
		for (int j = j0; j < j1; ++j) {
			const int k = skey[j], cls = (k & 3) * 2 + ((k >> 2) & 1);
#pragma unroll
			for (int q = 0; q < 8; ++q)
				cnt[q] += (cls == q);
		}
		const int wave = tid >> 6, lane = tid & 63;
		int excl[8];
#pragma unroll
		for (int q = 0; q < 8; ++q) {
			int v = cnt[q];
			for (int d = 1; d < 64; d <<= 1) {
				const int o = __shfl_up(v, d, 64);
				if (lane >= d)
					v += o;
			}
			excl[q] = v - cnt[q];
			if (lane == 63)
				s_wcnt[wave][q] = v;
		}
		__syncthreads();
		if (tid == 0) {
			int acc = 0;
			for (int q = 0; q < 8; ++q) {
				coff[q] = acc;
				for (int w = 0; w < K2_NT / 64; ++w) {
					const int v = s_wcnt[w][q];
					s_wcnt[w][q] = acc;	/* where wave w's share of class q starts */
					acc += v;
				}
			}
			coff[8] = acc;
		}
		__syncthreads();
#pragma unroll
		for (int q = 0; q < 8; ++q)
			excl[q] += s_wcnt[wave][q];
		for (int j = j0; j < j1; ++j) {
			const int k = skey[j], cls = (k & 3) * 2 + ((k >> 2) & 1);
			int at = 0;
#pragma unroll
			for (int q = 0; q < 8; ++q)
				if (cls == q)
					at = excl[q]++;
			clist[at] = (unsigned short)j;
		}
		__syncthreads();
	}
	const long long tk1 = wall_clock64();
	/* 2. successor table.  (Interleaving several searches per thread to overlap their LDS round trips
	 *    was slower: this kernel, with two serial machines inlined, has no registers to spare.) */
	for (int j = tid; j < ncand; j += K2_NT) {
		const int status = sstat[j] & 3;
		int nx = -1;
		if (status == CL_STEADY)
			nx = k2c_next(skey, clist, coff, sx[j], (sstat[j] >> 2) & 3);	/* sx[j] lies behind the candidate's own time */
		/* the first hop: rank of the candidate that follows the cluster | that one's status (the other three: 2b) */
		sw16[4 * j] = (nx < 0) ? (unsigned short)K2C_HOP_NONE : (unsigned short)((unsigned)nx | ((sstat[nx] & 3u) << K2C_RBITS));
	}
	__syncthreads();
	/* 2b. four hops per table entry: the walk below is one lane chasing pointers through LDS, a round
	 *     trip per read, so it reads as rarely as possible */
	for (int j = tid; j < ncand; j += K2_NT) {
		unsigned jv = 0;
		int at = j;
		bool open = ((sstat[j] & 3) == CL_STEADY);
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			unsigned hop = K2C_HOP_NONE;
			if (open) {
				hop = sw16[4 * at];	/* (first hops only are read here, and nobody writes them in this step) */
				if (hop != K2C_HOP_NONE) {
					open = ((hop >> K2C_RBITS) == CL_STEADY);
					if (open)
						at = (int)(hop & K2C_RANK);
				} else
					open = false;
			}
			if (i > 0)
				sw16[4 * j + i] = (unsigned short)hop;
			if (!open && !(jv & K2C_J_SPECIAL) && hop != K2C_HOP_NONE)
				jv |= K2C_J_SPECIAL | ((hop & K2C_RANK) << 16);
		}
		sjump[j] = jv | (unsigned)at | (open ? K2C_J_CONT : 0u);
	}
	__syncthreads();
	const long long tk2 = wall_clock64();
	bool steady_end = false;
	int nvis = 0;	/* slots of svis[] in use (thread 0's copy counts) */
	bool replay = false;	/* a cluster K2b did not make (CL_INVALID) is replayed here, from its candidate's instant */
	for (;;) {
		if (!tables_ok || st.fresh < VDL2_STEADY || replay) {
			/* history-dependent stretch (or no tables): serial machine.  (ONE call site: inlined into the kernel,
			 * its context stays in registers; with two the compiler made it a function and passed MachCtx /
			 * MachState / MachOut through 264 bytes of scratch per lane.)  A replay runs until the detector is
			 * history-free again behind at least one more trigger than the push has counted so far (a plain 1
			 * returned at once, without progress, when an earlier trigger had been counted). */
			const long long p0 = st.pos;
			const int c_trig = out.ntrig, c_rej = out.nrej, c_burst = out.nburst;
			const int rc = machine_run<K2_NT, false>(sh, cx, st, tables_ok, replay ? out.ntrig + 1 : 0, 1 << 30, replay ? 1 : 0, out);
			if (!replay)
				n_slow += (unsigned long long)(st.pos - p0);
			if (first_pass && tables_ok && (out.ntrig != c_trig || out.nrej != c_rej || out.nburst != c_burst)) {
				if (tid == 0 && nslog < VDL2_SLOG_CAP) {
					K2Slog g;
					g.t = (int)(p0 - cx.dec_base);
					g.ntrig = out.ntrig - c_trig;
					g.nrej = out.nrej - c_rej;
					g.nburst = out.nburst - c_burst;
					slog[nslog] = g;
				}
				++nslog;
			}
			replay = false;
			if (rc != MR_STEADY)
				break;
			continue;
		}
		/* 3. history-free: walk the successor table until something special happens */
		if (tid == 0) {
			int cur = k2c_next(skey, clist, coff, (int)(st.pos - cx.dec_base), st.r);
			int last = -1, why = 0;	/* why: 0 = no more candidates, 1 = special cluster at cur */
			if (lazy && (st.r != r_probe || (int)(st.pos & 1) != par_probe)) {
				/* the chain idles from here to the next candidate in a class the probe did not
				 * scan: K2a-verify must confirm there really is nothing in between */
				const int g_hi = (cur >= 0) ? (skey[cur] >> 2) : t_end;
				if (g_hi > seg_from) {
					const unsigned q = atomicAdd(nseg, 1u);
					if (q < VDL2_SEG_CAP) {
						Seg g;
						g.lo = (int)(st.pos - cx.dec_base);
						g.hi = g_hi;
						g.r = st.r;
						g.pad = 0;
						segs[q] = g;
					} else
						atomicMin(p.fail + sc, 0);
				}
			}
			if (cur >= 0 && (sstat[cur] & 3) != CL_STEADY)
				why = 1;
			else if (cur >= 0) {
				/* A lone lane chasing pointers: a wavefront on its own issues an instruction every ~5
				 * cycles, so the loop is four instructions and one LDS round trip per four clusters;
				 * what was visited is worked out from the list afterwards, by everybody. */
				if (nvis < K2C_VIS)
					svis[nvis] = (unsigned short)(cur | 0x8000);
				++nvis;
				last = cur;
				unsigned v;
				for (;;) {
					v = sjump[last];
					if (nvis < K2C_VIS)
						svis[nvis] = (unsigned short)last;
					++nvis;
					if (!(v & K2C_J_CONT))
						break;
					last = (int)(v & K2C_RANK);
				}
				last = (int)(v & K2C_RANK);
				if (v & K2C_J_SPECIAL) {
					cur = (int)((v >> 16) & K2C_RANK);
					why = 1;
				} else
					cur = -1;
				if (nvis > K2C_VIS)
					atomicMin(p.fail + sc, 0);
			}
			s_walk[0] = cur;
			s_walk[1] = last;
			s_walk[2] = why;
		}
		__syncthreads();
		const int cur = s_walk[0], last = s_walk[1], why = s_walk[2];
		__syncthreads();
		if (last >= 0) {
			st.pos = cx.dec_base + sx[last];
			st.r = (sstat[last] >> 2) & 3;
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
		const Cluster *cl = clusters + p.sidx[(size_t)sc * VDL2_CAND_CAP + cur];
		const int status = sstat[cur] & 3;
		if (first_pass && tid == 0)	/* (the chain met this candidate history-free too: whatever its cluster is, what follows is determined) */
			onchain[p.sidx[(size_t)sc * VDL2_CAND_CAP + cur]] = 1;
		if (status == CL_DEFER_FIRST) {
			st.pos = ncand_t;
			out.ndefer++;
			steady_end = true;
			break;
		}
		if (status == CL_INVALID) {
			/* no cluster for this candidate (not a primary, or the staging pool was full): replay the stretch at the top of the loop */
			st.pos = ncand_t;
			mach_materialize<K2_NT, false>(sh, cx, st.pos, st.r);
			replay = true;
			if (tid == 0 && p.dbg)
				atomicAdd(p.dbg + 25, 1ull);	/* replays of clusters K2b did not make */
			continue;
		}
		/* CL_NONSTEADY: its bursts count, then continue from the explicit state it stopped in */
		if (tid == 0) {
			if (nvis < K2C_VIS)
				svis[nvis] = (unsigned short)(cur | 0x8000);
			else
				atomicMin(p.fail + sc, 0);
			++nvis;
		}
		if (tid == 0 && p.dbg)
			atomicAdd(p.dbg + 26, 1ull);	/* non-steady clusters continued serially */
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
		s_walk[3] = nvis < K2C_VIS ? nvis : K2C_VIS;
	}
	__syncthreads();
	{
		int a = 0, b = 0, d = 0;
		const int nv4 = s_walk[3] * 4;
		for (int t = tid; t < nv4; t += K2_NT) {
			const unsigned ve = svis[t >> 2];
			int j = -1;
			if (ve & 0x8000u)
				j = (t & 3) == 0 ? (int)(ve & K2C_RANK) : -1;
			else {
				const unsigned hop = (unsigned)(swalk[ve] >> (16 * (t & 3))) & 0xffffu;
				if (hop != K2C_HOP_NONE && (hop >> K2C_RBITS) == CL_STEADY)
					j = (int)(hop & K2C_RANK);
			}
			if (j >= 0) {
				const unsigned gidx = p.sidx[(size_t)sc * VDL2_CAND_CAP + j];	/* (from device memory: a lane each, all in flight together) */
				const int2 hd = p.clhead[(size_t)sc * VDL2_CAND_CAP + gidx];
				if (first_pass)
					onchain[gidx] = 1;
				const unsigned hop0 = sw16[4 * j];
				const int next_t = (hop0 == K2C_HOP_NONE) ? t_end : (skey[hop0 & K2C_RANK] >> 2);	/* the successor's trigger, or the end of the data */
				const int ns = (hd.y >> 4) & 15;
				/* K2b's descriptors sit in static slots: (candidate index) * VDL2_CL_MAXB + burst */
				const unsigned slot0 = (unsigned)(((size_t)sc * VDL2_CAND_CAP + gidx) * VDL2_CL_MAXB);
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
				if (lazy && (sstat[j] & 3) == CL_STEADY && (r_s != r_probe || (int)(n_s & 1) != par_probe) &&
				    next_t > seg_from) {
					/* after this cluster the chain idles in class (r_s, parity of n_s) until
					 * the successor's trigger (or the end of the data) */
					const unsigned q = (unsigned)atomicAdd(&s_walk[1], 1);
					if (q < VDL2_SEG_CAP) {
						Seg g;
						g.lo = hd.x;
						g.hi = next_t;
						g.r = r_s;
						g.pad = 0;
						segs[q] = g;
					} else
						atomicMin(p.fail + sc, 0);
				}
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
		/* the output records of the selected bursts, reserved in one piece: K2d's workgroups (one per burst) then need no
		 * atomic of their own -- a thousand of them asking one device-scope counter for a slot at the same moment took 80 us,
		 * more than decoding the bursts (the same effect as in k4_frames) */
		/* Only the FIRST pass reserves here: every record it reserves is written (K2d's first pass decodes every channel's first
		 * selection, void or not).  A repair round's selection may be superseded by the next round's or by K2f's serial redo --
		 * records reserved for it would stay unwritten: whatever an earlier push left in the ring there would be handed out
		 * as bursts [found by scripts/soak.py's dropped-region-scan modes: two rounds, or a round and a redo].  The repaired
		 * selection's records are reserved by K2f, when it is final. */
		if (s_walk[0] > 0 && p.sel_reserved && p.round == 0)
			p.ctl[CTL_SELBASE0 + sc] = atomicAdd(p.outc, (unsigned)s_walk[0]);
		if (first_pass)
			p.ctl[CTL_NSLOG0 + sc] = (unsigned)(tables_ok ? nslog : VDL2_SLOG_CAP + 1);	/* (no tables: nothing a local repair could stand on) */
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
			atomicAdd(p.dbg + 27, (unsigned long long)ncand);
			atomicAdd(p.dbg + 28, (unsigned long long)s_walk[3]);	/* visited-list entries */
			atomicAdd(p.dbg + 29, n_slow);
		}
	}
}

/* ====================================================================== K2p
 * Local repair (the first repair round).  The verify pass found detector events the tables lacked -- a noise trigger that exists
 * in one timing class only, once per hundred channel-seconds of ordinary traffic: most pushes have one somewhere -- and listed them
 * as candidates without clusters.  Until round 5 the round re-resolved the failing channel from the START of the push (k2s_merge,
 * k2c_resolve with a CU's whole LDS, a verify pass over everything behind the event, every payload decoded again: 0.3 ms on the
 * tail of every push, and the chain resolver -> verify -> round -> commit -> next push's resolver was the period).  But the chain
 * is a deterministic function of where the detector stands: the event lies in a stretch the old chain IDLED through history-free
 * (that is what the verify pass scans), so up to the event the old chain stands; from the event on the new chain is followed --
 * the event replayed by the serial machine, then cluster by cluster through the tables -- until it arrives, history-free, at a
 * candidate the OLD chain also met history-free (K2Params.onchain): from there on the two are the same chain again.  A burst or
 * two as a rule (the class a burst leaves the detector in depends on the burst, hardly on the class it was met in).
 *   What changes is confined to the WINDOW [event, rejoin): the old chain's bursts triggered inside it are void (CTL_NWIN0 /
 *   K2Params.win: K2d's second pass tags their records), the new chain's bursts there are the repair's selection (sel_list2), the
 *   stretches the new chain idles through inside it are what the round's verify pass scans, the counters move by the difference.
 *   A new chain that does not rejoin before the data end leaves the channel state itself (the window runs to the end).
 * One workgroup per channel, tables read from device memory where they lie (a hop is three or four dependent reads; the windows
 * are a few hops): 8 KB of LDS where the resolver wants 157 -- a workgroup that wants a CU's whole LDS waits for a wide kernel
 * of another push to END.  What this kernel cannot do locally -- tables unusable, more new candidates than K2S_MERGE, more windows
 * than VDL2_WIN_CAP, a first pass with more serial stretches than its log holds -- it leaves failed: a further round resolves the
 * channel from its input state as before (k2s_merge + k2c_resolve), or K2f redoes it serially.
 * Match: d8psk.c:292-313 (what a trigger changes), 97-107 (a rejected header), 317-319 (the sub-phase sticks). */
#define K2P_NT K2_NT
#ifdef K2P_TRACE
#define K2P_TR(...) do { if (tid == 0 && sc == K2P_TRACE) printf(__VA_ARGS__); } while (0)
#else
#define K2P_TR(...) do { } while (0)
#endif
struct K2pShared {
	MachSharedT<K2P_NT> m;
	unsigned long long knew[K2S_MERGE];	/* the new candidates by time: (nrel * 4 + r) << 16 | index */
	int2 win[VDL2_WIN_CAP];
	union {
		K2xWork xw;			/* (while the verify pass's common area is drained) */
		int red[3][K2P_NT / 64];
	} u;
};

/* first rank of the sorted table whose key is >= key (wave-parallel 64-ary search: three rounds for 6144 entries) */
__device__ __forceinline__ int k2p_lower_bound(const int *skey, int n, int key)
{
	const int lane = threadIdx.x & 63;
	int lo = 0, hi = n;
	while (hi > lo) {
		const int step = (hi - lo + 63) / 64;
		const int j = lo + lane * step;
		const bool ge = j >= hi || skey[j] >= key;
		const unsigned long long m = __ballot(ge);
		const int f = m ? __builtin_ctzll(m) : 64;
		if (f == 0) {
			hi = lo;
			break;
		}
		const int nhi = lo + f * step < hi ? lo + f * step : hi;
		lo = lo + (f - 1) * step + 1;
		hi = nhi;
	}
	return lo;
}

/* first rank >= from of the sorted table with time >= want in class cls (sub-phase * 2 + parity of the time); n if none */
__device__ __forceinline__ int k2p_next_old(const int *skey, int from, int n, int want, int cls)
{
	const int lane = threadIdx.x & 63;
	for (int b = from; b < n; b += 64) {
		const int j = b + lane;
		const int k = j < n ? skey[j] : 0;
		const int t = k >> 2;
		const bool hit = j < n && t >= want && ((k & 3) * 2 + (t & 1)) == cls;
		const unsigned long long m = __ballot(hit);
		if (m)
			return b + __builtin_ctzll(m);
	}
	return n;
}

/* the same among the new candidates (LDS, sorted by time) */
__device__ __forceinline__ int k2p_next_new(const unsigned long long *knew, int n, int want, int cls)
{
	const int lane = threadIdx.x & 63;
	for (int b = 0; b < n; b += 64) {
		const int j = b + lane;
		const int k = j < n ? (int)(knew[j] >> 16) : 0;
		const int t = k >> 2;
		const bool hit = j < n && t >= want && ((k & 3) * 2 + (t & 1)) == cls;
		const unsigned long long m = __ballot(hit);
		if (m)
			return b + __builtin_ctzll(m);
	}
	return n;
}

__global__ __launch_bounds__(K2P_NT)
void k2p_patch(K2Params p)
{
	__shared__ K2pShared ps;
	MachSharedT<K2P_NT> &sh = ps.m;
	const int tid = threadIdx.x;
	const int c = blockIdx.x, s = (int)blockIdx.y;
	const int sc = s * VDL2_CS + c;
	if (p.force_serial)
		return;
	k2x_drain<K2P_NT>(ps.u.xw, p, sc);	/* the common area of the verify pass in front: what it finds there fails the channel like any other hit */
	const int fail_in = p.fail[sc];
	if (fail_in >= VDL2_VERIFIED)
		return;		/* verified in the first pass: nothing to repair */
	__syncthreads();
	const int ncand = (int)p.ctl[CTL_CAND0 + sc];
	const unsigned ovf_in = p.ctl[CTL_CAND0 + p.nstreams * VDL2_CS + sc];
	const int nold = (int)p.ctl[CTL_NCLUST0 + sc], nnew = ncand - nold;
	const unsigned nslog_in = p.ctl[CTL_NSLOG0 + sc];
	if (tid == 0) {
		p.redo[sc] = 1;
		atomicAdd(p.outc_total_redo + 1, 1u);	/* channel-pushes that went through a repair round */
		if (p.dbg)
			atomicAdd(p.dbg + 24, 1ull);
		p.ctl[CTL_NSEL1 + sc] = 0;
		p.ctl[CTL_NSEG0 + sc] = 0;
		p.ctl[CTL_NWIN0 + sc] = 0;
		if (p.fmask && sc < 512)
			atomicOr(p.fmask + (sc >> 5), 1u << (sc & 31));
	}
	/* what cannot be repaired locally stays failed (fail[] keeps its value): the next round, or K2f, takes the channel from its input state */
	/* fail_in <= 0 is not an event's instant: the verify pass could not vouch for the channel (items refused by a full list, more stretches than
	 * the lists hold): stretches it was to look at have not been looked at, and a local repair verifies only what IT changes
	 * [found by scripts/soak.py on handles whose item lists are small: seeds 6064, 6068, 6077] */
	K2P_TR("k2p sc %d: fail_in %d ncand %d nold %d nslog %u ovf %u dec_base %lld\n", sc, fail_in, ncand, nold, nslog_in, ovf_in, (long long)p.dec_base);
	if (ncand > VDL2_CAND_CAP || ovf_in != 0 || nnew <= 0 || nnew > K2S_MERGE || nslog_in > VDL2_SLOG_CAP || fail_in <= 0)
		return;
	__syncthreads();
	if (tid == 0)
		p.fail[sc] = 0x7f7f7f7f;	/* the repair is verified afresh */
	const ChanState *cs = p.cs + sc;	/* input state: left untouched until K2f commits */
	ChanState *cs_out = p.cs_out + sc;	/* the first pass's result: what this kernel amends */
	const Cand *cands = p.cands + (size_t)sc * VDL2_CAND_CAP;
	const int *skey = p.skey + (size_t)sc * VDL2_CAND_CAP;
	const unsigned short *sidx = p.sidx + (size_t)sc * VDL2_CAND_CAP;
	const int2 *heads = p.clhead + (size_t)sc * VDL2_CAND_CAP;
	const uint8_t *onchain = p.onchain + (size_t)sc * VDL2_CAND_CAP;
	const Cluster *clusters = p.clusters + (size_t)sc * VDL2_CAND_CAP;
	unsigned *sel = p.sel_list2 + (size_t)sc * VDL2_SEL_CAP;
	unsigned *nsel = p.ctl + CTL_NSEL1 + sc;
	Seg *segs = p.segs + (size_t)sc * VDL2_SEG_CAP;
	/* the new candidates by time: every one finds its rank by counting (there are a handful) */
	for (int i = tid; i < nnew; i += K2P_NT) {
		const unsigned long long v = (((unsigned long long)(unsigned)(cands[nold + i].nrel * 4 + cands[nold + i].r)) << 16) | (unsigned)(nold + i);
		int at = 0;
		for (int j = 0; j < nnew; ++j) {
			const unsigned long long w = (((unsigned long long)(unsigned)(cands[nold + j].nrel * 4 + cands[nold + j].r)) << 16) | (unsigned)(nold + j);
			at += (w < v) ? 1 : 0;
		}
		ps.knew[at] = v;
	}
	MachCtx cx;
	mach_ctx(cx, p, s, c, true);	/* bursts of serial stretches become descriptors */
	cx.sel = sel;
	cx.nsel = nsel;
	cx.dbg = nullptr;
	const int r_probe = p.probe_r, par_probe = p.probe_par;
	const int t_end = (int)(cx.avail_end - cx.dec_base);
	mach_init_taps(sh);	/* (barriers inside: knew[] is in place behind it) */
	MachOut out;
	out.nslots = out.ntrig = out.nrej = out.nburst = out.ndefer = 0;
	out.badslot = 0;
	out.neval = 0;
	MachState st;
	st.pos = 0;
	st.r = 0;
	st.fresh = VDL2_STEADY;
	int add_trig = 0, add_rej = 0, add_burst = 0;	/* what the table clusters on the new chain count */
	int add_defer = 0;
	int nwin = 0, nsegs = 0;
	int rk = 0;		/* rank of the old candidate the walk is looking at */
	int ev = 0;		/* next new candidate to look at */
	int t_cur = -0x7fffffff;	/* events before this lie inside a window already handled */
	bool to_end = false, steady_end = false, failed = false;
	/* mode of the one serial-machine call site below (two sites and the compiler makes it a function: see k2c_resolve) */
	enum { GO_WALK, GO_REPLAY, GO_SERIAL } go = GO_WALK;
	bool in_window = false;
	int win_lo = 0;
	for (;;) {
		if (!in_window) {
			/* the next event on the chain as it stands: the earliest new candidate behind the windows handled so far (it lies in
			 * a stretch the old chain idles through in its class: the verify pass found it there) */
			while (ev < nnew && (int)(ps.knew[ev] >> 18) < t_cur)
				++ev;
			if (ev >= nnew)
				break;
			const int key = (int)(ps.knew[ev] >> 16);
			++ev;
			win_lo = key >> 2;
			K2P_TR("  event %d r %d (t_cur %d)\n", key >> 2, key & 3, t_cur);
			in_window = true;
			st.pos = cx.dec_base + win_lo;
			st.r = key & 3;
			st.fresh = VDL2_STEADY;
			go = GO_REPLAY;
		}
		if (go != GO_WALK) {
			if (go == GO_REPLAY)
				mach_materialize<K2P_NT, false>(sh, cx, st.pos, st.r);
			const bool replay = go == GO_REPLAY;
			const int rc = machine_run<K2P_NT, false>(sh, cx, st, true, replay ? out.ntrig + 1 : 0, 1 << 30, replay ? 1 : 0, out);
			go = GO_WALK;
			K2P_TR("  machine rc %d -> pos %lld r %d fresh %d ntrig %d nburst %d ndefer %d\n", rc, (long long)(st.pos - cx.dec_base), st.r, st.fresh, out.ntrig, out.nburst, out.ndefer);
			if (rc != MR_STEADY) {	/* the data end (or a deferred burst) inside a history-dependent stretch: the channel state is the machine's */
				to_end = true;
				break;
			}
		}
		/* history-free at (st.pos, st.r): the next event is the first candidate of that class at or behind st.pos, old or new */
		const int want = (int)(st.pos - cx.dec_base), cls = st.r * 2 + (want & 1);
		/* (the search starts at the first entry at or behind `want` whatever its class: the hit of one class says nothing about where the
		 * candidates of the class the chain is in after the NEXT burst lie -- round 6's first version went on from the hit and, once a class
		 * had no further candidate, found none of any class: scripts/soak.py seed 6068) */
		rk = k2p_next_old(skey, k2p_lower_bound(skey, nold, want * 4), nold, want, cls);
		const int jn = k2p_next_new(ps.knew, nnew, want, cls);
		const int t_old = rk < nold ? (skey[rk] >> 2) : 0x7fffffff;
		const int t_new = jn < nnew ? (int)(ps.knew[jn] >> 18) : 0x7fffffff;
		const int t_next = t_old < t_new ? t_old : t_new;
		K2P_TR("  walk at %d r %d cls %d: t_old %d (rk %d) t_new %d\n", want, st.r, cls, t_old, rk, t_new);
		if (st.r != r_probe || (int)(st.pos & 1) != par_probe) {	/* (the probe's parity is one of STREAM time: a push's time base may be odd) */
			/* the new chain idles from here to that candidate in a class the probe did not scan: the round's verify pass looks */
			const int g_hi = t_next < t_end ? t_next : t_end;
			if (g_hi > want) {
				if (nsegs < VDL2_SEG_CAP) {
					if (tid == 0) {
						Seg g;
						g.lo = want;
						g.hi = g_hi;
						g.r = st.r;
						g.pad = 0;
						segs[nsegs] = g;
					}
				} else
					failed = true;
				++nsegs;
			}
		}
		if (t_next == 0x7fffffff) {	/* idle to the end of the data */
			const long long rem = (cx.avail_end - st.pos + 1) / 2;
			if (rem > 0)
				st.pos += 2 * rem;
			steady_end = true;
			to_end = true;
			break;
		}
		if (t_new < t_old) {	/* another new candidate: no cluster, replayed like the event */
			st.pos = cx.dec_base + t_new;
			go = GO_REPLAY;
			continue;
		}
		const int idx = sidx[rk];
		K2P_TR("  cand idx %d onchain %d\n", idx, (int)onchain[idx]);
		if (onchain[idx]) {
			/* the old chain met this candidate history-free as well: the same chain from here on */
			if (nwin < VDL2_WIN_CAP) {
				if (tid == 0)
					ps.win[nwin] = make_int2(win_lo, t_old);
			} else
				failed = true;
			++nwin;
			t_cur = t_old;
			in_window = false;
			continue;
		}
		const int2 hd = heads[idx];
		const int status = hd.y & 3;
		K2P_TR("  old cand idx %d status %d x %d\n", idx, status, hd.x);
		if (status == CL_INVALID) {
			st.pos = cx.dec_base + t_old;
			go = GO_REPLAY;
			continue;
		}
		if (status == CL_DEFER_FIRST) {	/* the burst is not completely inside the data held: the chain ends in front of its trigger */
			st.pos = cx.dec_base + t_old;
			add_defer++;
			steady_end = true;
			to_end = true;
			break;
		}
		{	/* a cluster K2b made: its bursts are on the new chain */
			const int ns = (hd.y >> 4) & 15;
			const unsigned slot0 = (unsigned)(((size_t)sc * VDL2_CAND_CAP + idx) * VDL2_CL_MAXB);
			if (tid == 0 && ns) {
				const unsigned q = atomicAdd(nsel, (unsigned)ns);	/* (the serial machine appends through the same counter) */
				for (int i = 0; i < ns; ++i) {
					if (q + i < VDL2_SEL_CAP)
						sel[q + i] = slot0 + i;
					else
						atomicAdd(p.outc + 1, 1u);
				}
			}
			add_trig += (hd.y >> 8) & 255;
			add_rej += (hd.y >> 16) & 255;
			add_burst += (hd.y >> 24) & 255;
		}
		if (status == CL_STEADY) {
			st.pos = cx.dec_base + hd.x;
			st.r = (hd.y >> 2) & 3;
			continue;
		}
		/* CL_NONSTEADY: go on from the explicit state the cluster stopped in */
		__syncthreads();
		mach_load(sh, &clusters[idx].saved);
		st.pos = clusters[idx].saved.pos;
		st.r = clusters[idx].saved.r;
		st.fresh = clusters[idx].saved.fresh < VDL2_STEADY ? clusters[idx].saved.fresh : VDL2_STEADY - 1;
		go = GO_SERIAL;
	}
	__syncthreads();
	K2P_TR(" end: to_end %d steady_end %d failed %d nwin %d nsegs %d pos %lld r %d\n", (int)to_end, (int)steady_end, (int)failed, nwin, nsegs, (long long)(st.pos - cx.dec_base), st.r);
	if (to_end) {	/* the last window runs to the end of the data */
		if (nwin < VDL2_WIN_CAP) {
			if (tid == 0)
				ps.win[nwin] = make_int2(win_lo, 0x7fffffff);
		} else
			failed = true;
		++nwin;
	}
	if (out.badslot)
		failed = true;
	if (failed) {	/* more windows or stretches than the lists hold: not repaired */
		if (tid == 0) {
			atomicMin(p.fail + sc, 0);
			p.ctl[CTL_NSEL1 + sc] = 0;
			p.ctl[CTL_NSEG0 + sc] = 0;
		}
		return;
	}
	__syncthreads();
	/* what the OLD chain counted inside the windows: the clusters it met there (their heads), the serial stretches it began there */
	int sub_trig = 0, sub_rej = 0, sub_burst = 0;
	for (int w = 0; w < nwin; ++w) {
		const int2 wn = ps.win[w];
		const int r_lo = k2p_lower_bound(skey, nold, wn.x * 4);
		const int r_hi = wn.y == 0x7fffffff ? nold : k2p_lower_bound(skey, nold, wn.y * 4);
		for (int j = r_lo + tid; j < r_hi; j += K2P_NT) {
			const int idx = sidx[j];
			if (onchain[idx]) {
				const int2 hd = heads[idx];
				sub_trig += (hd.y >> 8) & 255;
				sub_rej += (hd.y >> 16) & 255;
				sub_burst += (hd.y >> 24) & 255;
			}
		}
		if (tid < (int)nslog_in) {
			const K2Slog g = p.slog[(size_t)sc * VDL2_SLOG_CAP + tid];
			if (g.t >= wn.x && g.t < wn.y) {
				sub_trig += g.ntrig;
				sub_rej += g.nrej;
				sub_burst += g.nburst;
			}
		}
	}
	for (int d = 32; d > 0; d >>= 1) {
		sub_trig += __shfl_xor(sub_trig, d, 64);
		sub_rej += __shfl_xor(sub_rej, d, 64);
		sub_burst += __shfl_xor(sub_burst, d, 64);
	}
	if ((tid & 63) == 0) {
		ps.u.red[0][tid >> 6] = sub_trig;
		ps.u.red[1][tid >> 6] = sub_rej;
		ps.u.red[2][tid >> 6] = sub_burst;
	}
	__syncthreads();
	sub_trig = sub_rej = sub_burst = 0;
	for (int w = 0; w < K2P_NT / 64; ++w) {
		sub_trig += ps.u.red[0][w];
		sub_rej += ps.u.red[1][w];
		sub_burst += ps.u.red[2][w];
	}
	for (int w = tid; w < nwin; w += K2P_NT)
		p.win[(size_t)sc * VDL2_WIN_CAP + w] = ps.win[w];
	/* old counters of the push as the first pass left them in cs_out (read before anything is stored) */
	const unsigned long long o_trig = cs_out->n_trig, o_rej = cs_out->n_reject, o_burst = cs_out->n_burst, o_cand = cs_out->n_cand;
	__syncthreads();
	if (to_end) {
		if (steady_end) {
			mach_materialize<K2P_NT, false>(sh, cx, st.pos, st.r);
			st.fresh = VDL2_STEADY;
		}
		__syncthreads();
		mach_store(sh, st, cs_out);
	}
	if (tid == 0) {
		p.ctl[CTL_NSEG0 + sc] = (unsigned)nsegs;
		p.ctl[CTL_NWIN0 + sc] = (unsigned)nwin;
		cs_out->n_trig = o_trig - (unsigned long long)sub_trig + (unsigned long long)(add_trig + out.ntrig);
		cs_out->n_reject = o_rej - (unsigned long long)sub_rej + (unsigned long long)(add_rej + out.nrej);
		cs_out->n_burst = o_burst - (unsigned long long)sub_burst + (unsigned long long)(add_burst + out.nburst);
		cs_out->n_cand = o_cand + (unsigned long long)nnew;
		if (to_end) {
			cs_out->n_eval = cs->n_eval + (unsigned long long)((st.pos - cs->pos) / 2);
			cs_out->n_defer = cs->n_defer + (unsigned long long)(out.ndefer + add_defer);
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
	__shared__ K2xWork xw;
	const int tid = threadIdx.x;
	const int c = blockIdx.x, s = (int)blockIdx.y;
	const int sc = s * VDL2_CS + c;
	ChanState *cs = p.cs + sc;
	k2x_drain<K2_NT>(xw, p, sc);	/* the common area of the last verify pass */
	if (p.fail[sc] >= VDL2_VERIFIED) {
		const uint32_t *src = reinterpret_cast<const uint32_t *>(p.cs_out + sc);
		uint32_t *dst = reinterpret_cast<uint32_t *>(cs);
		for (int i = tid; i < (int)(sizeof(ChanState) / 4); i += K2_NT)
			dst[i] = src[i];
		/* a channel the repair rounds re-resolved: its selection is final now -- the records K2d's second pass fills */
		if (tid == 0 && p.sel_reserved && sc < 512 && (p.fmask[sc >> 5] >> (sc & 31) & 1u)) {
			const unsigned n = p.ctl[CTL_NSEL1 + sc];
			if (n > 0)
				p.ctl[CTL_SELBASE1 + sc] = atomicAdd(p.outc, n > VDL2_SEL_CAP ? VDL2_SEL_CAP : n);
		}
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
	out.badslot = 0;
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
		p.ctl[CTL_NSEL1 + sc] = 0;	/* K2d: nothing of the resolver's for this channel (it is masked: K2d reads the repair rounds' selection) ... */
		p.ctl[CTL_NWIN0 + sc] = VDL2_WIN_ALL;
		p.redo[sc] = 1;			/* (... also where the mask does not reach: K2d's one pass behind the commit asks this word) */
		if (p.fmask && sc < 512)	/* ... and if K2d ran ahead of the verify pass, the host drops what it made of it */
			atomicOr(p.fmask + (sc >> 5), 1u << (sc & 31));
	}
}

/* ====================================================================== K2d
 * Payload decode of the bursts that lie on the real chain: one workgroup per
 * selected burst descriptor, one lane per byte.
 */
#define K2D_NT 256
#ifndef K2D_WAVES
#define K2D_WAVES 5	/* 91 registers, no spill (round 4: 119 / four wavefronts -- with the table paths a run-time choice the kernel carried the
			 * code and the scalar registers of the paths it never takes: burst_payload<NT, TAB>) */
#endif
__global__ __launch_bounds__(K2D_NT) __attribute__((amdgpu_waves_per_eu(K2D_WAVES, 8)))
void k2d_payload(K2Params p)
{
	__shared__ unsigned s_slot;
	__shared__ float sph[VDL2_MAXSYM];
	__shared__ float s_tabs[72 + VDL2_ATAN_ROWS * VDL2_ATAN_STRIDE + 3 * 257];	/* mflt[], atanf range table, Grey1/2/3 (see burst_payload) */
	const int sc = (int)blockIdx.y;	/* stream * VDL2_CS + channel slot, like everywhere else: the grid spans all VDL2_CS slots of every stream */
	if ((sc % VDL2_CS) >= p.nbch)
		return;
	/* Which selection.  The first resolver pass's (sel_list) stands but for what the repair rounds made void of it (CTL_NWIN0: the
	 * bursts triggered inside the windows of a local repair, or all of it), the rounds' own (sel_list2) comes on top.
	 *   sel_mode 0: the first selection of every channel, beside the verify pass (nothing is known of any repair yet; records tagged 0);
	 *   sel_mode 1: second pass, behind the commit: only the channels a round touched (masked) -- the first pass's records that have
	 *               become void are tagged 2 (the host and K4 drop those), the rounds' selection is decoded (records tagged 1);
	 *   sel_mode 2: the one pass behind the commit of handles that cannot run the decode ahead (more than 512 channel slots: the 16-word
	 *               mask does not cover them; complete scans): what stands of the first selection, then the rounds' selection. */
	const bool touched = p.sel_mode == 2 ? p.redo[sc] != 0 : (sc < 512 && (p.fmask[sc >> 5] >> (sc & 31) & 1u));
	if (p.sel_mode == 1 && !touched)
		return;
	const unsigned nwin = (p.sel_mode != 0 && touched) ? p.ctl[CTL_NWIN0 + sc] : 0u;
	const int2 *win = p.win + (size_t)sc * VDL2_WIN_CAP;
	unsigned n0 = p.ctl[CTL_NSEL0 + sc], n1 = (p.sel_mode != 0 && touched) ? p.ctl[CTL_NSEL1 + sc] : 0u;
	n0 = n0 > VDL2_SEL_CAP ? VDL2_SEL_CAP : n0;
	n1 = n1 > VDL2_SEL_CAP ? VDL2_SEL_CAP : n1;
	const unsigned *sel0 = p.sel_list + (size_t)sc * VDL2_SEL_CAP, *sel1 = p.sel_list2 + (size_t)sc * VDL2_SEL_CAP;
	if (p.sel_mode == 1 && nwin != 0 && p.sel_reserved) {
		/* the first pass's records of this channel lie in one piece from CTL_SELBASE0 on: tag what has become void */
		const unsigned base0 = p.ctl[CTL_SELBASE0 + sc];
		for (unsigned i = blockIdx.x * K2D_NT + threadIdx.x; i < n0; i += gridDim.x * K2D_NT) {
			if (base0 + i >= p.rec_cap)
				break;
			vdl2gpu_burst_t *rec = p.recs + base0 + i;
			bool dead = nwin == VDL2_WIN_ALL;
			if (!dead) {
				const long long t = rec->trig_dec - p.dec_base;
				for (unsigned w = 0; w < nwin && w < VDL2_WIN_CAP; ++w)
					dead = dead || (t >= win[w].x && t < win[w].y);
			}
			if (dead)
				rec->trig_sample = 2;
		}
	}
	/* the bursts this launch decodes for the channel: entries [0, na) of the first selection, then [0, n1) of the rounds' */
	const unsigned na = p.sel_mode == 1 ? 0u : ((p.sel_mode == 2 && nwin == VDL2_WIN_ALL) ? 0u : n0);
	const unsigned n = na + n1;
	if (blockIdx.x >= n)
		return;
	for (int i = threadIdx.x; i < 72; i += K2D_NT)
		s_tabs[i] = (i < 65) ? d_tab(c_mflt, i) : 0.0f;
	if (threadIdx.x < VDL2_ATAN_ROWS * VDL2_ATAN_STRIDE)
		s_tabs[72 + threadIdx.x] = vdl2_atan_tab_entry(threadIdx.x);
	for (int i = threadIdx.x; i < 257; i += K2D_NT) {
		float *g = s_tabs + 72 + VDL2_ATAN_ROWS * VDL2_ATAN_STRIDE;
		g[i] = d_tab(c_grey1, i);
		g[257 + i] = d_tab(c_grey2, i);
		g[514 + i] = d_tab(c_grey3, i);
	}
	__syncthreads();
	/* reserved by K2c (first selection) and K2f (the rounds' selection, when it is final): K2d fills them without atomics */
	const unsigned base0 = p.ctl[CTL_SELBASE0 + sc], base1 = p.ctl[CTL_SELBASE1 + sc];
	for (unsigned i = blockIdx.x; i < n; i += gridDim.x) {
		const bool second = i >= na;
		const BurstDesc d = p.stage[second ? sel1[i - na] : sel0[i]];
		if (!second && p.sel_mode == 2 && nwin != 0) {	/* (block-uniform) a burst of the first selection inside a repaired window: not on the chain any more */
			const long long t = d.nstar - p.dec_base;
			bool dead = false;
			for (unsigned w = 0; w < nwin && w < VDL2_WIN_CAP; ++w)
				dead = dead || (t >= win[w].x && t < win[w].y);
			if (dead)
				continue;
		}
		unsigned slot;
		if (p.sel_reserved)
			slot = (second ? base1 : base0) + (second ? i - na : i);
		else {
			__syncthreads();
			if (threadIdx.x == 0)
				s_slot = atomicAdd(p.outc, 1u);
			__syncthreads();
			slot = s_slot;
		}
		if (slot >= p.rec_cap) {	/* ring full: counted, never silent */
			if (threadIdx.x == 0)
				atomicAdd(p.outc + 1, 1u);
			slot = 0xffffffffu;
		}
		if (slot != 0xffffffffu) {
			const int s = d.sc / VDL2_CS;
			const float2 *x0 = p.dec + (size_t)d.sc * p.cap - p.dec_base;
			burst_payload<K2D_NT, true>(p.recs + slot, x0, p.pn, d.nstar, d.clk0, d.df, d.nbrow, d.nlbyte, s, p.cfg[d.sc], sph, p.sel_mode == 0 ? 0 : 1, d.sc, s_tabs, p.pn8);
		}
		__syncthreads();
	}
}

#endif
