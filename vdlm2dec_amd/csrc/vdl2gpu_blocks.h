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
 *   BM/Chien/Forney  one lane, the reference's update order, so that miscorrections, the early
 *                  exits and the partially applied corrections of an uncorrectable row agree
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

struct K4Params {
	const vdl2gpu_burst_t *recs;
	const unsigned *nrecs_dev;	/* number of records (device word), or nullptr: nrecs */
	unsigned nrecs;
	unsigned rec_cap;
	vdl2gpu_frame_t *frames;
	unsigned *nframes;		/* [0] frames written, [1] frames dropped (buffer full), [2] bytes used (compact) */
	unsigned frame_cap;		/* records, or bytes if compact */
	int compact;			/* 0: an array of vdl2gpu_frame_t; 1: entries of 56 header bytes (the struct's head) +
					 * len data bytes, each rounded up to 8 bytes -- a frame is rarely longer than 100 bytes */
	unsigned long long *dbg;	/* diagnostics: stage cycle counters, or nullptr */
};

struct K4Shared {
	uint8_t gexp[512], glog[256];
	unsigned short crc_tab[256];
	uint8_t row[VDL2GPU_MAXROWS][256];	/* the burst's rows, corrected in place */
	uint8_t src[K4_MAXBY + 8];		/* data bytes of all rows, in order */
	unsigned dst[(K4_MAXBY + 8) / 4 + 2];	/* un-stuffed bit stream */
	int lsum[K4_NT][2];			/* per lane: all ones?, trailing ones */
	int tin[K4_NT];				/* run of ones in front of the lane's first bit */
	int kept[K4_NT + 1];			/* exclusive prefix of kept bits */
	unsigned char lor[K4_NT + 1];		/* exclusive prefix OR of the lanes' bytes */
	int eras[6];
	int ctl[16];
	/* Berlekamp-Massey work arrays live here, not in (scratch-backed) private arrays */
	uint8_t lam[8], syn[8], bpoly[8], tpoly[8], omg[8], root[8], reg[8], loc[8];
	int rs_deg, rs_count;
	unsigned short crc_adv[16];	/* FCS state advanced over one lane's worth of zero bytes, per state bit */
	unsigned short crc_in[K4_NT + 1];	/* running FCS at the start of each lane's bytes */
	unsigned short crc_own[K4_NT];	/* FCS (from zero) of each lane's bytes */
};

__device__ __forceinline__ int k4_m255(int x)
{
	while (x >= 255) {
		x -= 255;
		x = (x >> 8) + (x & 255);
	}
	return x;
}

/* rs.c:81-291 for one row whose syndromes are not all zero, in three parts.
 * k4_rs_bm (one lane): erasure locator and Berlekamp-Massey in the reference's update order; leaves
 * lambda in index form and its degree in LDS. */
__device__ void k4_rs_bm(K4Shared &sh, const unsigned *synv, const int *eras_pos, int no_eras)
{
	enum { NR = 6, NNN = 255 };
	const uint8_t *gexp = sh.gexp, *glog = sh.glog;
	uint8_t *lam = sh.lam, *syn = sh.syn, *bpoly = sh.bpoly, *tpoly = sh.tpoly;
	for (int i = 0; i < NR; i++)
		syn[i] = glog[synv[i]];	/* index form, NOLOG for zero */
	for (int i = 0; i <= NR; i++)
		lam[i] = 0;
	lam[0] = 1;
	if (no_eras > 0) {
		lam[1] = gexp[k4_m255(NNN - 1 - eras_pos[0])];
		for (int i = 1; i < no_eras; i++) {
			const int u = k4_m255(NNN - 1 - eras_pos[i]);
			for (int j = i + 1; j > 0; j--) {
				const uint8_t lg = glog[lam[j - 1]];
				if (lg != K4_NOLOG)
					lam[j] ^= gexp[k4_m255(u + lg)];
			}
		}
	}
	for (int i = 0; i <= NR; i++)
		bpoly[i] = glog[lam[i]];
	int el = no_eras;
	for (int r = no_eras + 1; r <= NR; r++) {
		uint8_t disc = 0;
		for (int i = 0; i < r; i++)
			if (lam[i] && syn[r - i - 1] != K4_NOLOG)
				disc ^= gexp[k4_m255(glog[lam[i]] + syn[r - i - 1])];
		const uint8_t dl = glog[disc];
		if (dl == K4_NOLOG) {
			for (int i = NR; i > 0; i--)
				bpoly[i] = bpoly[i - 1];
			bpoly[0] = K4_NOLOG;
			continue;
		}
		tpoly[0] = lam[0];
		for (int i = 0; i < NR; i++)
			tpoly[i + 1] = (bpoly[i] != K4_NOLOG) ? (uint8_t)(lam[i + 1] ^ gexp[k4_m255(dl + bpoly[i])]) : lam[i + 1];
		if (2 * el <= r + no_eras - 1) {
			el = r + no_eras - el;
			for (int i = 0; i <= NR; i++)
				bpoly[i] = lam[i] ? (uint8_t)k4_m255(glog[lam[i]] - dl + NNN) : (uint8_t)K4_NOLOG;
		} else {
			for (int i = NR; i > 0; i--)
				bpoly[i] = bpoly[i - 1];
			bpoly[0] = K4_NOLOG;
		}
		for (int i = 0; i <= NR; i++)
			lam[i] = tpoly[i];
	}
	int deg = 0;
	for (int i = 0; i <= NR; i++) {
		lam[i] = glog[lam[i]];
		if (lam[i] != K4_NOLOG)
			deg = i;
	}
	sh.rs_deg = deg;
	sh.rs_count = 0;
}

/* Chien search, all lanes: position i (1..255) is a root when 1 + sum_j lambda_j alpha^(i j) = 0.  The
 * reference walks i upwards and stops at the deg-th root; a polynomial of degree deg has no more, so
 * listing all roots in the order of i is the same list. */
__device__ void k4_rs_chien(K4Shared &sh, int lane)
{
	const int deg = sh.rs_deg;
	for (int t = 0; t < 4; ++t) {
		const int i = 1 + lane + 64 * t;
		bool isroot = false;
		if (i <= 255) {
			unsigned q = 1;
			for (int j = 1; j <= deg; ++j) {
				const unsigned l = sh.lam[j];
				if (l != K4_NOLOG)
					q ^= sh.gexp[(l + (unsigned)(i * j)) % 255u];
			}
			isroot = (q == 0);
		}
		const unsigned long long m = __ballot(isroot);
		if (isroot) {
			const int pos = sh.rs_count + __popcll(m & ((1ull << lane) - 1ull));
			if (pos < 6) {
				sh.root[pos] = (uint8_t)i;
				sh.loc[pos] = (uint8_t)((i - 1) % 255);
			}
		}
		__syncthreads();
		if (lane == 0)
			sh.rs_count += __popcll(m);
		__syncthreads();
	}
}

/* Omega and Forney (one lane), last root first; a zero denominator abandons the row with the
 * corrections made so far, like the reference.  eras[] in/out like the reference. */
__device__ void k4_rs_forney(K4Shared &sh, uint8_t *data, int *eras_pos)
{
	enum { NR = 6, NNN = 255, FIRST = 120 };
	const uint8_t *gexp = sh.gexp, *glog = sh.glog;
	const uint8_t *lam = sh.lam, *syn = sh.syn, *root = sh.root, *loc = sh.loc;
	uint8_t *omg = sh.omg;
	const int deg = sh.rs_deg, count = sh.rs_count;
	if (deg != count)
		return;
	int dego = 0;
	for (int i = 0; i < NR; i++) {
		uint8_t tmp = 0;
		for (int j = (deg < i) ? deg : i; j >= 0; j--)
			if (syn[i - j] != K4_NOLOG && lam[j] != K4_NOLOG)
				tmp ^= gexp[k4_m255(syn[i - j] + lam[j])];
		if (tmp)
			dego = i;
		omg[i] = glog[tmp];
	}
	omg[NR] = K4_NOLOG;
	for (int j = count - 1; j >= 0; j--) {
		uint8_t num1 = 0;
		for (int i = dego; i >= 0; i--)
			if (omg[i] != K4_NOLOG)
				num1 ^= gexp[k4_m255(omg[i] + i * root[j])];
		const uint8_t num2 = gexp[k4_m255(root[j] * (FIRST - 1) + NNN)];
		uint8_t den = 0;
		const int top = (deg < NR - 1 ? deg : NR - 1) & ~1;
		for (int i = top; i >= 0; i -= 2)
			if (lam[i + 1] != K4_NOLOG)
				den ^= gexp[k4_m255(lam[i + 1] + i * root[j])];
		if (den == 0)
			return;
		if (num1)
			data[loc[j]] ^= gexp[k4_m255(glog[num1] + glog[num2] + NNN - glog[den])];
	}
	for (int i = 0; i < count; i++)
		eras_pos[i] = loc[i];
}

__global__ __launch_bounds__(K4_NT)
void k4_frames(K4Params p)
{
	__shared__ K4Shared sh;
	const int lane = threadIdx.x;
	const bool prof = p.dbg && lane == 0;
	long long tq = prof ? clock64() : 0;
	/* tables: GF(256)/0x187 (rs.c:17-79), FCS-16 reflected 0x8408 (crc.c) */
	if (lane == 0) {
		unsigned x = 1;
		for (int i = 0; i < 255; i++) {
			sh.gexp[i] = (uint8_t)x;
			sh.glog[x] = (uint8_t)i;
			x <<= 1;
			if (x & 0x100)
				x ^= 0x187;
		}
		for (int i = 255; i < 512; i++)
			sh.gexp[i] = sh.gexp[i - 255];
		sh.glog[0] = K4_NOLOG;
	}
	for (int v = lane; v < 256; v += K4_NT) {
		unsigned c = (unsigned)v;
		for (int i = 0; i < 8; i++)
			c = (c & 1) ? ((c >> 1) ^ 0x8408u) : (c >> 1);
		sh.crc_tab[v] = (unsigned short)c;
	}
	__syncthreads();
	unsigned nrecs = p.nrecs_dev ? *p.nrecs_dev : p.nrecs;
	nrecs = nrecs > p.rec_cap ? p.rec_cap : nrecs;
#define K4_STAMP(slot) do { if (prof) { const long long tn = clock64(); atomicAdd(p.dbg + 48 + (slot), (unsigned long long)(tn - tq)); tq = tn; } } while (0)
	K4_STAMP(0);
	for (unsigned ib = blockIdx.x; ib < nrecs; ib += gridDim.x) {
		const vdl2gpu_burst_t *rec = p.recs + ib;
		const long long tb0 = prof ? clock64() : 0;
		if (prof)
			tq = tb0;
		const int nbrow = rec->nbrow, nlbyte = rec->nlbyte;
		if (nbrow < 1 || nbrow > VDL2GPU_MAXROWS || nlbyte < 0 || nlbyte > 249)
			continue;
		/* ---- rows to LDS */
		for (int i = lane; i < nbrow * 64; i += K4_NT) {
			const int r = i >> 6, w = i & 63;
			if (w * 4 < 255) {
				const uint8_t *d = &rec->data[r][w * 4];
				for (int b = 0; b < 4 && w * 4 + b < 255; ++b)
					sh.row[r][w * 4 + b] = d[b];
			}
		}
		if (lane < 6)
			sh.eras[lane] = 0;
		__syncthreads();
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
			for (int b = 0; b < 4; ++b) {
				const int j = lane * 4 + b;
				if (j < 255) {
					const unsigned d = sh.row[r][j];
					if (d) {
						const int lg = sh.glog[d];
#pragma unroll
						for (int i = 0; i < 6; ++i)
							syn[i] ^= sh.gexp[(lg + (120 + i) * (254 - j)) % 255];
					}
				}
			}
#pragma unroll
			for (int i = 0; i < 6; ++i)
				for (int d = 32; d > 0; d >>= 1)
					syn[i] ^= __shfl_xor(syn[i], d, 64);
			const unsigned any = syn[0] | syn[1] | syn[2] | syn[3] | syn[4] | syn[5];
			__syncthreads();
			if (any) {	/* wave-uniform: every lane holds the reduced syndromes */
				if (lane == 0)
					k4_rs_bm(sh, syn, sh.eras, nera);
				__syncthreads();
				k4_rs_chien(sh, lane);
				if (lane == 0)
					k4_rs_forney(sh, sh.row[r], sh.eras);
			}
			__syncthreads();
			for (int i = lane; i < by; i += K4_NT)
				sh.src[nby + i] = sh.row[r][i];
			nby += by;
		}
		K4_STAMP(2);
		for (int i = lane; i < (K4_MAXBY + 8) / 4 + 2; i += K4_NT)
			sh.dst[i] = 0u;
		__syncthreads();
		/* ---- HDLC bit un-stuffing (vdlm2.c:116-128): lane owns bytes [b0, b1) */
		const int per = (nby + K4_NT - 1) / K4_NT;
		const int b0 = lane * per < nby ? lane * per : nby;
		const int b1 = b0 + per < nby ? b0 + per : nby;
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
			sh.lsum[lane][0] = all;
			sh.lsum[lane][1] = trail;
		}
		__syncthreads();
		if (lane == 0) {
			int t = 0;
			for (int l = 0; l < K4_NT; ++l) {
				sh.tin[l] = t;
				t = sh.lsum[l][0] ? t + sh.lsum[l][1] : sh.lsum[l][1];
			}
		}
		__syncthreads();
		int nkeep = 0;
		{
			int t = sh.tin[lane];
			for (int i = b0; i < b1; ++i) {
				const unsigned v = sh.src[i];
				for (int n = 0; n < 8; ++n) {
					if (v & (1u << n)) {
						++t;
						++nkeep;
					} else {
						if (t != 5)
							++nkeep;
						t = 0;
					}
				}
			}
		}
		sh.kept[lane + 1] = nkeep;
		__syncthreads();
		if (lane == 0) {
			int a = 0;
			sh.kept[0] = 0;
			for (int l = 1; l <= K4_NT; ++l) {
				a += sh.kept[l];
				sh.kept[l] = a;
			}
		}
		__syncthreads();
		{
			int t = sh.tin[lane], o = sh.kept[lane];
			unsigned acc = 0;	/* bits of the output word being filled */
			int w = o >> 5;
			for (int i = b0; i < b1; ++i) {
				const unsigned v = sh.src[i];
				for (int n = 0; n < 8; ++n) {
					const unsigned bit = (v >> n) & 1u;
					if (bit)
						++t;
					else {
						const bool stuffed = (t == 5);
						t = 0;
						if (stuffed)
							continue;
					}
					acc |= bit << (o & 31);
					++o;
					if ((o & 31) == 0) {
						atomicOr(&sh.dst[w], acc);
						acc = 0;
						++w;
					}
				}
			}
			if (acc)
				atomicOr(&sh.dst[w], acc);
		}
		__syncthreads();
		K4_STAMP(3);
		const int nb = sh.kept[K4_NT] >> 3;	/* whole un-stuffed bytes */
		const uint8_t *B = reinterpret_cast<const uint8_t *>(sh.dst);
		/* ---- first flag: the byte at which the OR of all bytes so far equals 0x7e (vdlm2.c:129-133) */
		const int per2 = (nb + K4_NT - 1) / K4_NT;
		const int c0 = lane * per2 < nb ? lane * per2 : nb;
		const int c1 = c0 + per2 < nb ? c0 + per2 : nb;
		{
			unsigned o = 0;
			for (int i = c0; i < c1; ++i)
				o |= B[i];
			sh.lor[lane + 1] = (unsigned char)o;
		}
		if (lane == 0) {
			sh.ctl[0] = 0x7fffffff;	/* m0 */
			sh.ctl[1] = 0x7fffffff;	/* m1 */
		}
		__syncthreads();
		if (lane == 0) {
			unsigned a = 0;
			sh.lor[0] = 0;
			for (int l = 1; l <= K4_NT; ++l) {
				a |= sh.lor[l];
				sh.lor[l] = (unsigned char)a;
			}
		}
		__syncthreads();
		{
			unsigned o = sh.lor[lane];
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
		__syncthreads();
		const int m0 = sh.ctl[0];
		if (m0 != 0x7fffffff) {
			/* flags right behind the first one are swallowed (k == 1, vdlm2.c:134-135) */
			for (int i = c0 > m0 + 1 ? c0 : m0 + 1; i < c1; ++i)
				if (B[i] != 0x7eu) {
					atomicMin(&sh.ctl[1], i);
					break;
				}
		}
		__syncthreads();
		const int m1 = sh.ctl[1];
		if (m0 == 0x7fffffff || m1 == 0x7fffffff) {
			__syncthreads();
			continue;
		}
		/* ---- hdata[] = 0x7e, B[m1], B[m1+1], ...; every later 0x7e closes a candidate frame
		 *      hdata[0..k] whose FCS runs over hdata[1..k-1] (check_frame, vdlm2.c:38-61).  All candidates
		 *      start at m1, so one running FCS serves -- computed lane-parallel: the FCS is linear, the state
		 *      at the start of a lane's bytes is (state one lane earlier, advanced over per2 zero bytes)
		 *      xor (FCS from zero of that lane's bytes). */
		K4_STAMP(4);
		if (lane < 16) {	/* the advance map, one state bit per lane */
			unsigned c = 1u << lane;
			for (int i = 0; i < per2; ++i)
				c = (c >> 8) ^ sh.crc_tab[c & 0xffu];
			sh.crc_adv[lane] = (unsigned short)c;
		}
		const int L1 = m1 / per2;	/* lane that holds m1 (per2 >= 1 since nb > m1) */
		{
			unsigned c = (lane == L1) ? 0xffffu : 0u;
			for (int i = (lane == L1) ? m1 : c0; i < c1; ++i)
				c = (c >> 8) ^ sh.crc_tab[(c ^ B[i]) & 0xffu];
			sh.crc_own[lane] = (unsigned short)c;
		}
		if (lane == 0)
			sh.ctl[1] = 0;	/* number of frames */
		__syncthreads();
		if (lane == 0) {
			unsigned st = sh.crc_own[L1];	/* state after lane L1's bytes (from 0xffff at m1) */
			for (int l = L1 + 1; l < K4_NT; ++l) {
				sh.crc_in[l] = (unsigned short)st;
				unsigned adv = 0;
				for (int b = 0; b < 16; ++b)
					if (st & (1u << b))
						adv ^= sh.crc_adv[b];
				st = adv ^ sh.crc_own[l];	/* exact for whole lanes; the last lane's state is not needed */
			}
		}
		__syncthreads();
		if (lane >= L1) {
			unsigned c = (lane == L1) ? 0xffffu : sh.crc_in[lane];
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
		__syncthreads();
		if (lane == 0) {	/* frames in stream order (there is almost never more than one) */
			int nf0 = sh.ctl[1] < 12 ? sh.ctl[1] : 12;
			for (int i = 1; i < nf0; ++i)
				for (int j = i; j > 0 && sh.ctl[2 + j] < sh.ctl[1 + j]; --j) {
					const int t = sh.ctl[2 + j];
					sh.ctl[2 + j] = sh.ctl[1 + j];
					sh.ctl[1 + j] = t;
				}
			sh.ctl[1] = nf0;
		}
		__syncthreads();
		K4_STAMP(5);
		const int nf = sh.ctl[1];
		for (int f = 0; f < nf; ++f) {
			const int q = sh.ctl[2 + f];
			const int len = q - m1 + 2;
			if (lane == 0) {
				unsigned slot;
				if (p.compact) {
					const unsigned sz = (unsigned)((offsetof(vdl2gpu_frame_t, data) + len + 7) & ~7);
					slot = atomicAdd(p.nframes + 2, sz);	/* byte offset */
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
			__syncthreads();
			const unsigned slot = (unsigned)sh.ctl[0];
			if (slot != 0xffffffffu) {
				vdl2gpu_frame_t *fr = p.compact ? reinterpret_cast<vdl2gpu_frame_t *>(reinterpret_cast<char *>(p.frames) + slot) : p.frames + slot;
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
			__syncthreads();
		}
		__syncthreads();
		K4_STAMP(6);
		if (prof) {
			atomicAdd(p.dbg + 48 + 7, 1ull);
			const unsigned long long tot = (unsigned long long)(tq - tb0);
			atomicMax(p.dbg + 48 + 8, tot);
			if (tot > 400000ull) {
				atomicAdd(p.dbg + 48 + 9, 1ull);
				p.dbg[48 + 10] = ((unsigned long long)nbrow << 32) | (unsigned)nlbyte;
				p.dbg[48 + 11] = (unsigned long long)sh.kept[K4_NT];
			}
		}
	}
#undef K4_STAMP
}

#endif
