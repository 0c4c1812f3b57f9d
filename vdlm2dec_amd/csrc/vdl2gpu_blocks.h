/*
 * vdl2gpu_blocks.h -- the reference's block path as one HIP kernel:
 *   blk_thread()  vdlm2.c:84-161   per msgblk_t: RS(255,249) per row, HDLC bit un-stuffing,
 *                                  flag hunt, frame check, out()
 *   rs()          rs.c:81-291      errors-and-erasures Berlekamp-Massey / Chien / Forney over
 *                                  GF(256)/0x187, first root alpha^120
 *   check_frame() vdlm2.c:38-61    length >= 13, FCS-16 (crc.c) residue 0xf0b8
 *
 * One wavefront per burst record.  Integer/byte work, bit-exact by construction:
 *   syndromes      64 lanes x 4 bytes each, XOR-reduced (a syndrome is a field sum: any order)
 *   BM/Chien/Forney  a coefficient (a candidate position, a root) per lane; each step performs the
 *                  reference's field operations on the operands the reference's loops read, so that
 *                  miscorrections, the early exits and the partially applied corrections of an
 *                  uncorrectable row agree
 *   un-stuffing    a stuffed zero is a zero after exactly five ones: the run of ones in front of a
 *                  bit does not depend on what was dropped before, so every lane knows the run it
 *                  starts in from a scan of (all ones?, trailing ones) and drops its own zeros;
 *                  a prefix sum of the kept counts places its bits in the output
 *   flag hunt      the reference ORs bits into hdata[0] until it equals 0x7e (vdlm2.c:124-139): a
 *                  prefix OR over the un-stuffed bytes finds that byte; flags right behind it are
 *                  swallowed; from then on every byte is stored and every 0x7e closes a candidate frame
 *                  that starts at hdata[1] -- all candidates share the start, so one running FCS serves
 * Records come from the output ring of the demodulator (device memory) or from the host.
 */
#ifndef VDL2GPU_BLOCKS_H
#define VDL2GPU_BLOCKS_H

#define K4_NT 64
#define K4_NOLOG 255
#define K4_MAXBY (VDL2GPU_MAXROWS * 249)
#define K4_SLOT 256

struct K4Params {
	const vdl2gpu_burst_t *recs;
	const unsigned *nrecs_dev;	/* number of records (device word), or nullptr: nrecs */
	unsigned nrecs;
	unsigned rec_cap;
	vdl2gpu_frame_t *frames;
	unsigned *nframes;		/* [0] frames written (compact: to the arena), [1] frames dropped (buffer full),
					 * [2] arena bytes used (compact) */
	unsigned frame_cap;		/* records, or bytes if compact */
	int compact;			/* 0: an array of vdl2gpu_frame_t.
					 * 1: entries of 56 header bytes (the struct's head) + len data bytes.  Record i owns
					 * the K4_SLOT bytes at i * K4_SLOT: the first frame of its burst goes there when it
					 * fits (len 0: none) -- no allocation, because two thousand waves asking one
					 * device-scope counter for space at the same moment cost more than the whole block
					 * path; longer and further frames go to the arena behind rec_cap slots, entries
					 * rounded up to 8 bytes, space taken from nframes[2] */
	const unsigned *fmask;		/* not nullptr: the records come from the pipeline, where K2d's second pass tags the void ones 2 (trig_sample) */
	unsigned long long *dbg;	/* diagnostics: stage cycle counters, or nullptr */
	const unsigned *tabs;		/* K4_TABW words from k4_tables: gexp[512], glog[256], crc_tab[256] */
};

#define K4_TABW ((512 + 256 + 512) / 4)
/* A block is one wavefront: its lanes meet at every instruction, LDS operations of a wave complete in
 * order, and all a barrier has to do is keep the compiler from moving LDS accesses across it.  (A
 * __syncthreads() would also wait for every global store in flight.) */
static_assert(K4_NT == 64, "k4_frames is written for one wavefront per block");
#define K4_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
		       __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#define K4_RECW ((int)(sizeof(vdl2gpu_burst_t) / 4))
#define K4_RECW0 192	/* words of a record fetched before its header has been looked at: header + two rows */
static_assert(offsetof(vdl2gpu_burst_t, data) % 4 == 0 && sizeof(vdl2gpu_burst_t) % 4 == 0, "rows are fetched as words");

struct K4Shared {
	uint8_t gexp[512], glog[256];	/* these three in the order of K4Params.tabs */
	unsigned short crc_tab[256];
	unsigned long long recw[(sizeof(vdl2gpu_burst_t) + 7) / 8];	/* the record; its rows are corrected in place */
	uint8_t src[K4_MAXBY + 8];		/* data bytes of all rows, in order */
	unsigned dst[(K4_MAXBY + 8) / 4 + 2];	/* un-stuffed bit stream */
	int eras[6];
	unsigned fm[16];	/* K4Params.fmask */
	int ctl[16];
	/* Berlekamp-Massey work arrays live here, not in (scratch-backed) private arrays */
	uint8_t lam[8], syn[8], bpoly[8], tpoly[8], omg[8], root[8], reg[8], loc[8];
	int rs_deg, rs_count;
};

/* GF(256)/0x187 (rs.c:17-79) and FCS-16 reflected 0x8408 (crc.c) tables, built once per handle */
__global__ void k4_tables(unsigned *out)
{
	__shared__ uint8_t t[512 + 256 + 512];
	uint8_t *gexp = t, *glog = t + 512;
	unsigned short *crc = reinterpret_cast<unsigned short *>(t + 768);
	if (threadIdx.x == 0) {
		unsigned x = 1;
		for (int i = 0; i < 255; i++) {
			gexp[i] = (uint8_t)x;
			glog[x] = (uint8_t)i;
			x <<= 1;
			if (x & 0x100)
				x ^= 0x187;
		}
		for (int i = 255; i < 512; i++)
			gexp[i] = gexp[i - 255];
		glog[0] = K4_NOLOG;
	}
	for (int v = threadIdx.x; v < 256; v += blockDim.x) {
		unsigned c = (unsigned)v;
		for (int i = 0; i < 8; i++)
			c = (c & 1) ? ((c >> 1) ^ 0x8408u) : (c >> 1);
		crc[v] = (unsigned short)c;
	}
	__syncthreads();
	for (int i = threadIdx.x; i < K4_TABW; i += blockDim.x)
		out[i] = reinterpret_cast<const unsigned *>(t)[i];
}

__device__ __forceinline__ int k4_m255(int x)
{
	while (x >= 255) {
		x -= 255;
		x = (x >> 8) + (x & 255);
	}
	return x;
}

/* rs.c:81-291 for one row whose syndromes are not all zero, in three parts, all of them spread over
 * the lanes (a single lane walking these little polynomials pays an LDS round trip per table look-up
 * and took longer than everything else in the kernel together).
 * k4_rs_bm: erasure locator and Berlekamp-Massey.  Lane i owns coefficient i of lambda (value form) and
 * of B (index form); one step of the reference's r loop is a discrepancy (XOR over the lanes), a shift
 * of B by one lane, and one update per lane -- the same field operations on the same operands as the
 * reference's loops, which read only values of the step before.  Leaves lambda in index form and its
 * degree in LDS. */
__device__ void k4_rs_bm(K4Shared &sh, const unsigned *synv, int no_eras, int lane)
{
	enum { NR = 6, NNN = 255 };
	const uint8_t *gexp = sh.gexp, *glog = sh.glog;
	{
		unsigned mine = 0;
#pragma unroll
		for (int i = 0; i < NR; i++)
			mine = (lane == i) ? synv[i] : mine;
		if (lane < NR)
			sh.syn[lane] = glog[mine];	/* index form, NOLOG for zero */
	}
	unsigned lam = (lane == 0) ? 1u : 0u;
	if (no_eras > 0) {
		if (lane == 1)
			lam = gexp[k4_m255(NNN - 1 - sh.eras[0])];
		for (int i = 1; i < no_eras; i++) {
			const int u = k4_m255(NNN - 1 - sh.eras[i]);
			const unsigned lp = __shfl_up(lam, 1, K4_NT);
			if (lane >= 1 && lane <= i + 1) {
				const unsigned lg = glog[lp];
				if (lg != K4_NOLOG)
					lam ^= gexp[k4_m255(u + lg)];
			}
		}
	}
	unsigned b = (lane <= NR) ? (unsigned)glog[lam] : (unsigned)K4_NOLOG;
	K4_SYNC();
	int el = no_eras;
	for (int r = no_eras + 1; r <= NR; r++) {
		unsigned term = 0;
		if (lane < r && lam) {
			const unsigned sy = sh.syn[r - lane - 1];
			if (sy != K4_NOLOG)
				term = gexp[k4_m255(glog[lam] + sy)];
		}
		term ^= __shfl_xor(term, 1, K4_NT);
		term ^= __shfl_xor(term, 2, K4_NT);
		term ^= __shfl_xor(term, 4, K4_NT);
		const unsigned disc = (unsigned)__builtin_amdgcn_readfirstlane((int)term);
		unsigned bprev = __shfl_up(b, 1, K4_NT);
		if (lane == 0)
			bprev = K4_NOLOG;
		if (disc == 0) {
			b = bprev;
			continue;
		}
		const unsigned dl = glog[disc];
		unsigned t = lam;
		if (lane >= 1 && lane <= NR && bprev != K4_NOLOG)
			t = lam ^ gexp[k4_m255(dl + bprev)];
		if (2 * el <= r + no_eras - 1) {
			el = r + no_eras - el;
			b = (lane <= NR && lam) ? (unsigned)k4_m255(glog[lam] - dl + NNN) : (unsigned)K4_NOLOG;
		} else
			b = bprev;
		lam = t;
	}
	const unsigned li = (lane <= NR) ? (unsigned)glog[lam] : (unsigned)K4_NOLOG;
	if (lane <= NR)
		sh.lam[lane] = (uint8_t)li;
	const unsigned long long m = __ballot(li != K4_NOLOG);	/* bit 0 is always set: lambda_0 = 1 */
	if (lane == 0) {
		sh.rs_deg = 63 - __clzll((long long)m);
		sh.rs_count = 0;
	}
}

/* Chien search, all lanes: position i (1..255) is a root when 1 + sum_j lambda_j alpha^(i j) = 0.  The
 * reference walks i upwards and stops at the deg-th root; a polynomial of degree deg has no more, so
 * listing all roots in the order of i is the same list. */
__device__ void k4_rs_chien(K4Shared &sh, int lane)
{
	const int deg = sh.rs_deg;
	int count = 0;	/* the same in every lane */
	for (int t = 0; t < 4; ++t) {
		const int i = 1 + lane + 64 * t;
		bool isroot = false;
		if (i <= 255) {
			unsigned q = 1;
			for (int j = 1; j <= deg; ++j) {
				const unsigned l = sh.lam[j];
				if (l != K4_NOLOG)
					q ^= sh.gexp[k4_m255((int)l + i * j)];
			}
			isroot = (q == 0);
		}
		const unsigned long long m = __ballot(isroot);
		if (isroot) {
			const int pos = count + __popcll(m & ((1ull << lane) - 1ull));
			if (pos < 6) {
				sh.root[pos] = (uint8_t)i;
				sh.loc[pos] = (uint8_t)((i - 1) % 255);
			}
		}
		count += __popcll(m);
	}
	if (lane == 0)
		sh.rs_count = count;
}

/* Omega and Forney.  Lane i computes omega_i, then lane j the correction for root j.  The reference
 * takes the roots last first and abandons the row at the first zero denominator, keeping the corrections
 * made so far: here, the roots above the highest one with a zero denominator.  eras[] in/out like the
 * reference's eras_pos[]. */
__device__ void k4_rs_forney(K4Shared &sh, uint8_t *data, int lane)
{
	enum { NR = 6, NNN = 255, FIRST = 120 };
	const uint8_t *gexp = sh.gexp, *glog = sh.glog;
	const uint8_t *lam = sh.lam, *syn = sh.syn;
	const int deg = sh.rs_deg, count = sh.rs_count;
	if (deg != count)
		return;
	unsigned tmp = 0;
	if (lane < NR) {
		for (int j = (deg < lane) ? deg : lane; j >= 0; j--)
			if (syn[lane - j] != K4_NOLOG && lam[j] != K4_NOLOG)
				tmp ^= gexp[k4_m255(syn[lane - j] + lam[j])];
		sh.omg[lane] = glog[tmp];
	}
	if (lane == NR)
		sh.omg[NR] = K4_NOLOG;
	const unsigned long long mo = __ballot(lane < NR && tmp != 0);
	const int dego = mo ? 63 - __clzll((long long)mo) : 0;
	K4_SYNC();
	unsigned num1 = 0, num2 = 0, den = 1;
	int loc = 0;
	if (lane < count) {
		const int root = sh.root[lane];
		loc = sh.loc[lane];
		for (int i = dego; i >= 0; i--)
			if (sh.omg[i] != K4_NOLOG)
				num1 ^= gexp[k4_m255(sh.omg[i] + i * root)];
		num2 = gexp[k4_m255(root * (FIRST - 1) + NNN)];
		den = 0;
		const int top = (deg < NR - 1 ? deg : NR - 1) & ~1;
		for (int i = top; i >= 0; i -= 2)
			if (lam[i + 1] != K4_NOLOG)
				den ^= gexp[k4_m255(lam[i + 1] + i * root)];
	}
	const unsigned long long bad = __ballot(lane < count && den == 0);
	const int jstop = bad ? 63 - __clzll((long long)bad) : -1;
	if (lane < count && lane > jstop && num1)
		data[loc] ^= gexp[k4_m255(glog[num1] + glog[num2] + NNN - glog[den])];
	if (!bad && lane < count)
		sh.eras[lane] = loc;
}

__global__ __launch_bounds__(K4_NT)
void k4_frames(K4Params p)
{
	__shared__ K4Shared sh;
	const int lane = threadIdx.x;
	const bool prof = p.dbg && lane == 0;
	long long tq = prof ? clock64() : 0;
	/* Everything the first record needs from memory is asked for at once -- the tables, the record
	 * count and the front of record blockIdx.x -- because this kernel runs beside the next push's
	 * channeliser, which keeps the memory queues full: every dependent round trip costs microseconds. */
	unsigned tw[(K4_TABW + K4_NT - 1) / K4_NT], rw[K4_RECW0 / K4_NT] = {};
	const unsigned fmw = (p.fmask && lane < 16) ? p.fmask[lane] : 0u;
#pragma unroll
	for (int k = 0; k < (K4_TABW + K4_NT - 1) / K4_NT; ++k)
		tw[k] = (lane + K4_NT * k < K4_TABW) ? p.tabs[lane + K4_NT * k] : 0u;
	if (blockIdx.x < p.rec_cap) {
		const unsigned *w = reinterpret_cast<const unsigned *>(p.recs + blockIdx.x);
#pragma unroll
		for (int k = 0; k < K4_RECW0 / K4_NT; ++k)
			rw[k] = w[lane + K4_NT * k];
	}
	unsigned nrecs = p.nrecs_dev ? *p.nrecs_dev : p.nrecs;
	nrecs = nrecs > p.rec_cap ? p.rec_cap : nrecs;
	if (blockIdx.x >= nrecs)
		return;
#pragma unroll
	for (int k = 0; k < (K4_TABW + K4_NT - 1) / K4_NT; ++k)
		if (lane + K4_NT * k < K4_TABW)
			reinterpret_cast<unsigned *>(sh.gexp)[lane + K4_NT * k] = tw[k];
	if (lane < 16)
		sh.fm[lane] = fmw;
	unsigned *const recw = reinterpret_cast<unsigned *>(sh.recw);
	const vdl2gpu_burst_t *const rec = reinterpret_cast<const vdl2gpu_burst_t *>(sh.recw);	/* the copy in LDS */
	uint8_t *const rows = reinterpret_cast<uint8_t *>(sh.recw) + offsetof(vdl2gpu_burst_t, data);
	/* exponents of the syndrome sums for this lane's four bytes, (120+i)(254-j) mod 255, a byte each */
	unsigned sexp[4][2];
#pragma unroll
	for (int b = 0; b < 4; ++b) {
		const int j = lane * 4 + b;
		sexp[b][0] = sexp[b][1] = 0;
#pragma unroll
		for (int i = 0; i < 6; ++i)
			sexp[b][i / 4] |= (unsigned)k4_m255((120 + i) * (254 - (j < 255 ? j : 254))) << (8 * (i % 4));
	}
#define K4_STAMP(slot) do { if (prof) { const long long tn = clock64(); atomicAdd(p.dbg + 48 + (slot), (unsigned long long)(tn - tq)); tq = tn; } } while (0)
	K4_STAMP(0);
	for (unsigned ib = blockIdx.x; ib < nrecs; ib += gridDim.x) {
		const unsigned *gw = reinterpret_cast<const unsigned *>(p.recs + ib);
		if (ib != blockIdx.x) {
			K4_SYNC();	/* the previous record's frame bytes have been read */
#pragma unroll
			for (int k = 0; k < K4_RECW0 / K4_NT; ++k)
				rw[k] = gw[lane + K4_NT * k];
		}
#pragma unroll
		for (int k = 0; k < K4_RECW0 / K4_NT; ++k)
			recw[lane + K4_NT * k] = rw[k];
		K4_SYNC();
		const long long tb0 = prof ? clock64() : 0;
		if (prof)
			tq = tb0;
		const int nbrow = rec->nbrow, nlbyte = rec->nlbyte;
		const unsigned hdr = (unsigned)offsetof(vdl2gpu_frame_t, data);
		vdl2gpu_frame_t *const myslot = p.compact ? reinterpret_cast<vdl2gpu_frame_t *>(reinterpret_cast<char *>(p.frames) + (size_t)ib * K4_SLOT) : nullptr;
		if (myslot && lane == 0)
			myslot->len = 0;
		if (p.fmask && rec->trig_sample == 2)	/* (device-side tag) a burst of the first selection that a repair round made void: K2d's second pass says so */
			continue;
		if (nbrow < 1 || nbrow > VDL2GPU_MAXROWS || nlbyte < 0 || nlbyte > 249)
			continue;
		/* ---- rows to LDS */
		for (int i = K4_RECW0 + lane; i < ((int)offsetof(vdl2gpu_burst_t, data) + nbrow * VDL2GPU_ROWLEN + 3) / 4; i += K4_NT)
			recw[i] = gw[i];
		if (lane < 6)
			sh.eras[lane] = 0;
		K4_SYNC();
		K4_STAMP(1);
		/* ---- RS per row (rows in order: eras_pos[] carries over, vdlm2.c:104-113) */
		int nby = 0;
		for (int r = 0; r < nbrow; ++r) {
			int by = 249, nera = 0;
			if (r == nbrow - 1) {
				by = nlbyte;
				if (lane == 0) {	/* set_eras(), vdlm2.c:63-82 */
					if (by <= 67) {
						sh.eras[0] = 253;
						sh.eras[1] = 254;
					}
					if (by <= 30) {
						sh.eras[0] = 251;
						sh.eras[1] = 252;
						sh.eras[2] = 253;
						sh.eras[3] = 254;
					}
				}
				nera = by <= 30 ? 4 : (by <= 67 ? 2 : 0);
			}
			/* syndromes S_i = sum_j data[j] alpha^((120+i)(254-j)) */
			unsigned syn[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
			for (int b = 0; b < 4; ++b) {
				const int j = lane * 4 + b;
				if (j < 255) {
					const unsigned d = rows[r * VDL2GPU_ROWLEN + j];
					if (d) {
						const unsigned lg = sh.glog[d];
#pragma unroll
						for (int i = 0; i < 6; ++i)
							syn[i] ^= sh.gexp[lg + ((sexp[b][i / 4] >> (8 * (i % 4))) & 0xffu)];
					}
				}
			}
			{	/* field sums: XOR over the lanes, four syndromes to a word */
				unsigned s03 = syn[0] | (syn[1] << 8) | (syn[2] << 16) | (syn[3] << 24), s45 = syn[4] | (syn[5] << 8);
				for (int d = 32; d > 0; d >>= 1) {
					s03 ^= __shfl_xor(s03, d, K4_NT);
					s45 ^= __shfl_xor(s45, d, K4_NT);
				}
				syn[0] = s03 & 0xffu;
				syn[1] = (s03 >> 8) & 0xffu;
				syn[2] = (s03 >> 16) & 0xffu;
				syn[3] = s03 >> 24;
				syn[4] = s45 & 0xffu;
				syn[5] = s45 >> 8;
			}
			const unsigned any = syn[0] | syn[1] | syn[2] | syn[3] | syn[4] | syn[5];
			K4_SYNC();
			if (any) {	/* wave-uniform: every lane holds the reduced syndromes */
				k4_rs_bm(sh, syn, nera, lane);
				K4_SYNC();
				k4_rs_chien(sh, lane);
				K4_SYNC();
				k4_rs_forney(sh, rows + r * VDL2GPU_ROWLEN, lane);
			}
			K4_SYNC();
			for (int i = lane; i < by; i += K4_NT)
				sh.src[nby + i] = rows[r * VDL2GPU_ROWLEN + i];
			nby += by;
		}
		K4_STAMP(2);
		for (int i = lane; i < nby / 4 + 2; i += K4_NT)
			sh.dst[i] = 0u;
		K4_SYNC();
		/* ---- HDLC bit un-stuffing (vdlm2.c:116-128): lane owns bytes [b0, b1) */
		const int per = (nby + K4_NT - 1) / K4_NT;
		const int b0 = lane * per < nby ? lane * per : nby;
		const int b1 = b0 + per < nby ? b0 + per : nby;
		int tin;
		{
			int all = 1, trail = 0;
			for (int i = b0; i < b1; ++i) {
				const unsigned v = sh.src[i];
				if (v == 0xffu)
					trail += 8;
				else {
					all = 0;
					trail = __clz((int)((~v & 0xffu) << 24));	/* ones above the highest zero (bit 7 downwards) */
				}
			}
			/* run of ones in front of the lane's first bit: scan of t' = all ? t + trail : trail */
			for (int d = 1; d < K4_NT; d <<= 1) {
				const int la = __shfl_up(all, d, K4_NT), lt = __shfl_up(trail, d, K4_NT);
				if (lane >= d) {
					trail = all ? lt + trail : trail;
					all &= la;
				}
			}
			tin = __shfl_up(trail, 1, K4_NT);
			if (lane == 0)
				tin = 0;
		}
		/* A zero is a stuffed one iff exactly five ones stand in front of it.  With the (at most six
		 * relevant) bits in front of a byte put below it, that is a bit pattern test on the whole byte. */
		auto drops = [](unsigned v, int t) -> unsigned {
			const int tt = t < 6 ? t : 6;
			const unsigned W = (v << 6) | (((1u << tt) - 1u) << (6 - tt));
			const unsigned D = ~W & (W << 1) & (W << 2) & (W << 3) & (W << 4) & (W << 5) & ~(W << 6);
			return (D >> 6) & 0xffu;
		};
		auto run_after = [](unsigned v, int t) -> int {	/* ones at the end of the stream after byte v */
			return v == 0xffu ? t + 8 : __clz((int)((~v & 0xffu) << 24));
		};
		int nkeep = 0;
		{
			int t = tin;
			for (int i = b0; i < b1; ++i) {
				const unsigned v = sh.src[i];
				nkeep += 8 - __popc(drops(v, t));
				t = run_after(v, t);
			}
		}
		int kept_incl = nkeep;	/* prefix sum of the kept bits */
		for (int d = 1; d < K4_NT; d <<= 1) {
			const int o = __shfl_up(kept_incl, d, K4_NT);
			if (lane >= d)
				kept_incl += o;
		}
		const int kept_all = __shfl(kept_incl, K4_NT - 1, K4_NT);
		{
			int t = tin, o = kept_incl - nkeep;
			unsigned long long acc = 0;	/* bits of the output word being filled, and what spills over */
			int w = o >> 5;
			for (int i = b0; i < b1; ++i) {
				unsigned v = sh.src[i];
				unsigned dr = drops(v, t);
				t = run_after(v, t);
				int nbits = 8;
				while (dr) {	/* squeeze the dropped zeros out, highest first (there are at most two) */
					const int d = 31 - __clz((int)dr);
					v = (v & ((1u << d) - 1u)) | ((v >> (d + 1)) << d);
					dr &= ~(1u << d);
					--nbits;
				}
				acc |= (unsigned long long)v << (o & 31);
				o += nbits;
				if ((o >> 5) != w) {
					atomicOr(&sh.dst[w], (unsigned)acc);
					acc >>= 32;
					++w;
				}
			}
			if ((unsigned)acc)
				atomicOr(&sh.dst[w], (unsigned)acc);
		}
		K4_SYNC();
		K4_STAMP(3);
		const int nb = kept_all >> 3;	/* whole un-stuffed bytes */
		const uint8_t *B = reinterpret_cast<const uint8_t *>(sh.dst);
		/* ---- first flag: the byte at which the OR of all bytes so far equals 0x7e (vdlm2.c:129-133) */
		const int per2 = (nb + K4_NT - 1) / K4_NT;
		const int c0 = lane * per2 < nb ? lane * per2 : nb;
		const int c1 = c0 + per2 < nb ? c0 + per2 : nb;
		unsigned lor = 0;	/* OR of all bytes in front of the lane's */
		{
			for (int i = c0; i < c1; ++i)
				lor |= B[i];
			for (int d = 1; d < K4_NT; d <<= 1) {
				const unsigned o = __shfl_up(lor, d, K4_NT);
				if (lane >= d)
					lor |= o;
			}
			lor = __shfl_up(lor, 1, K4_NT);
			if (lane == 0)
				lor = 0;
		}
		if (lane == 0) {
			sh.ctl[0] = 0x7fffffff;	/* m0 */
			sh.ctl[1] = 0x7fffffff;	/* m1 */
		}
		K4_SYNC();
		{
			unsigned o = lor;
			for (int i = c0; i < c1; ++i) {
				o |= B[i];
				if (o == 0x7eu) {
					atomicMin(&sh.ctl[0], i);
					break;
				}
				if (o & ~0x7eu)
					break;	/* a bit outside 0x7e is set for good: no flag will ever be seen */
			}
		}
		K4_SYNC();
		const int m0 = sh.ctl[0];
		if (m0 != 0x7fffffff) {
			/* flags right behind the first one are swallowed (k == 1, vdlm2.c:134-135) */
			for (int i = c0 > m0 + 1 ? c0 : m0 + 1; i < c1; ++i)
				if (B[i] != 0x7eu) {
					atomicMin(&sh.ctl[1], i);
					break;
				}
		}
		K4_SYNC();
		const int m1 = sh.ctl[1];
		if (m0 == 0x7fffffff || m1 == 0x7fffffff) {
			K4_SYNC();
			continue;
		}
		/* ---- hdata[] = 0x7e, B[m1], B[m1+1], ...; every later 0x7e closes a candidate frame
		 *      hdata[0..k] whose FCS runs over hdata[1..k-1] (check_frame, vdlm2.c:38-61).  All candidates
		 *      start at m1, so one running FCS serves -- computed lane-parallel: the FCS is linear, the state
		 *      at the start of a lane's bytes is (state one lane earlier, advanced over per2 zero bytes)
		 *      xor (FCS from zero of that lane's bytes); that recurrence is scanned in log2(64) doubling steps,
		 *      the advance map (16 columns, one per lane) being squared alongside. */
		K4_STAMP(4);
		unsigned col = 0;	/* the advance map, one state bit per lane */
		if (lane < 16) {
			col = 1u << lane;
			for (int i = 0; i < per2; ++i)
				col = (col >> 8) ^ sh.crc_tab[col & 0xffu];
		}
		const int L1 = m1 / per2;	/* lane that holds m1 (per2 >= 1 since nb > m1) */
		unsigned crc_in;
		{
			unsigned x = 0;	/* FCS over the lane's own bytes, from 0xffff at m1, from zero behind it */
			if (lane >= L1) {
				x = (lane == L1) ? 0xffffu : 0u;
				for (int i = (lane == L1) ? m1 : c0; i < c1; ++i)
					x = (x >> 8) ^ sh.crc_tab[(x ^ B[i]) & 0xffu];
			}
			for (int d = 1; d < K4_NT; d <<= 1) {
				unsigned xl = __shfl_up(x, d, K4_NT);
				if (lane < d)
					xl = 0;
				unsigned y = 0, nc = 0;
#pragma unroll
				for (int b = 0; b < 16; ++b) {
					const unsigned cb = (unsigned)__builtin_amdgcn_readlane((int)col, b);
					y ^= (0u - ((xl >> b) & 1u)) & cb;
					nc ^= (0u - ((col >> b) & 1u)) & cb;
				}
				x ^= y;	/* exact for whole lanes; the last, partial lane's end state is not needed */
				col = nc;
			}
			crc_in = __shfl_up(x, 1, K4_NT);	/* state at the lane's first byte */
		}
		if (lane == 0)
			sh.ctl[1] = 0;	/* number of frames */
		K4_SYNC();
		if (lane >= L1) {
			unsigned c = (lane == L1) ? 0xffffu : crc_in;
			for (int q = (lane == L1) ? m1 : c0; q < c1; ++q) {
				const unsigned v = B[q];
				if (v == 0x7eu && (q - m1 + 2) >= 13 && c == 0xf0b8u) {
					const int k = atomicAdd(&sh.ctl[1], 1);
					if (k < 12)
						sh.ctl[2 + k] = q;
				}
				c = (c >> 8) ^ sh.crc_tab[(c ^ v) & 0xffu];
			}
		}
		K4_SYNC();
		if (lane == 0) {	/* frames in stream order (there is almost never more than one) */
			int nf0 = sh.ctl[1] < 12 ? sh.ctl[1] : 12;
			if (sh.ctl[1] > 12)	/* more CRC-clean frames in one burst than the table holds: counted, not silent */
				atomicAdd(p.nframes + 1, (unsigned)(sh.ctl[1] - 12));
			for (int i = 1; i < nf0; ++i)
				for (int j = i; j > 0 && sh.ctl[2 + j] < sh.ctl[1 + j]; --j) {
					const int t = sh.ctl[2 + j];
					sh.ctl[2 + j] = sh.ctl[1 + j];
					sh.ctl[1 + j] = t;
				}
			sh.ctl[1] = nf0;
		}
		K4_SYNC();
		K4_STAMP(5);
		const int nf = sh.ctl[1];
		for (int f = 0; f < nf; ++f) {
			const int q = sh.ctl[2 + f];
			const int len = q - m1 + 2;
			const bool inslot = myslot && f == 0 && hdr + (unsigned)len <= K4_SLOT;	/* uniform */
			if (!inslot) {
				if (lane == 0) {
					unsigned slot;
					if (p.compact) {
						const unsigned sz = (hdr + (unsigned)len + 7u) & ~7u;
						const unsigned base = p.rec_cap * K4_SLOT;
						slot = base + atomicAdd(p.nframes + 2, sz);	/* byte offset */
						if (slot + sz > p.frame_cap) {
							atomicAdd(p.nframes + 1, 1u);
							slot = 0xffffffffu;
						} else
							atomicAdd(p.nframes, 1u);
					} else {
						slot = atomicAdd(p.nframes, 1u);
						if (slot >= p.frame_cap) {
							atomicAdd(p.nframes + 1, 1u);
							slot = 0xffffffffu;
						}
					}
					sh.ctl[0] = (int)slot;
				}
				K4_SYNC();
			}
			const unsigned slot = inslot ? 0u : (unsigned)sh.ctl[0];
			K4_STAMP(12);
			if (slot != 0xffffffffu) {
				vdl2gpu_frame_t *fr = inslot ? myslot
						     : (p.compact ? reinterpret_cast<vdl2gpu_frame_t *>(reinterpret_cast<char *>(p.frames) + slot) : p.frames + slot);
				if (lane == 0) {
					fr->stream = rec->stream;
					fr->chn = rec->chn;
					fr->Fr = rec->Fr;
					fr->nbrow = nbrow;
					fr->nlbyte = nlbyte;
					fr->len = len;
					fr->block = (int32_t)ib;
					fr->seq = f;
					fr->df = rec->df;
					fr->ppm = rec->ppm;
					fr->trig_dec = rec->trig_dec;
					fr->end_dec = rec->end_dec;
					fr->data[0] = 0x7e;
				}
				for (int i = lane; i < len - 1 && i + 1 < VDL2GPU_MAXFRAME; i += K4_NT)
					fr->data[i + 1] = B[m1 + i];
			}
			K4_SYNC();
		}
		K4_SYNC();
		K4_STAMP(6);
		if (prof) {
			atomicAdd(p.dbg + 48 + 7, 1ull);
			const unsigned long long tot = (unsigned long long)(tq - tb0);
			atomicMax(p.dbg + 48 + 8, tot);
			if (nbrow >= 3) {
				atomicAdd(p.dbg + 48 + 13, tot);
				atomicAdd(p.dbg + 48 + 14, 1ull);
			}
			atomicMax(p.dbg + 48 + 15, (unsigned long long)nbrow);
			if (tot > 400000ull) {
				atomicAdd(p.dbg + 48 + 9, 1ull);
				p.dbg[48 + 10] = ((unsigned long long)nbrow << 32) | (unsigned)nlbyte;
				p.dbg[48 + 11] = (unsigned long long)kept_all;
			}
		}
	}
#undef K4_STAMP
}

#endif
