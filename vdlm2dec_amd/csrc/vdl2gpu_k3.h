/* vdl2gpu_k3.h -- K3 and the small per-push kernels.  Part of the device side of libvdl2gpu.so; included by vdl2gpu_kernels.h only. */
#ifndef VDL2GPU_K3_H
#define VDL2GPU_K3_H

/* ======================================================================= K3
 * The carry: the last VDL2_CARRY_FRAMES frames of the previous push's planes (49152: the longest burst plus history) go
 * in front of this push's output in the other plane set, right-aligned below frame VDL2_CARRY_FRAMES where the
 * channeliser starts writing.  A FIXED amount: round 2 copied only what no channel had consumed yet, which made the
 * copy -- and with it the next push's scan -- wait for the resolver.  p.J = outputs of the previous push: its frames
 * [J, J + VDL2_CARRY_FRAMES) are the stretch (part of ITS carry if it was shorter than that).
 */
#define K3_THREADS 256
__global__ __launch_bounds__(K3_THREADS)
void k3_carry(K3Params p)
{
	const int c = blockIdx.y, s = (int)blockIdx.z;
	typedef float v4f __attribute__((ext_vector_type(4)));
	const size_t plane = ((size_t)s * VDL2_CS + c) * p.cap;
	const float2 *src = p.src + plane + p.J;
	float2 *dst = p.dst + plane;
	const int i0 = (int)(blockIdx.x * K3_THREADS + threadIdx.x), step = (int)(gridDim.x * K3_THREADS);
	if ((p.J & 1) == 0) {	/* planes start on 128-byte lines: an even offset keeps 16-byte alignment */
		const v4f *s4 = reinterpret_cast<const v4f *>(src);
		v4f *d4 = reinterpret_cast<v4f *>(dst);
		for (int i = i0; i < VDL2_CARRY_FRAMES / 2; i += step)
			d4[i] = s4[i];
	} else
		for (int i = i0; i < VDL2_CARRY_FRAMES; i += step)
			dst[i] = src[i];
}

/* one launch instead of four memsets */
__global__ void k_push_init(KInitParams p)
{
	if (blockIdx.x)
		return;
	for (int i = threadIdx.x; i < p.ctl_words; i += blockDim.x)
		p.ctl[i] = 0u;
	for (int i = threadIdx.x; i < p.nsc; i += blockDim.x) {
		p.fail[i] = 0x7f7f7f7f;
		p.redo[i] = 0;
	}
	if (threadIdx.x < 2)
		p.outc[threadIdx.x] = 0u;
	if (threadIdx.x < 16)
		p.fmask[threadIdx.x] = 0u;
	if (p.fcnt && threadIdx.x < 4)
		p.fcnt[threadIdx.x] = 0u;
}

/* runs after k3_compact (same stream): publish the new time base and hand the push's counters to the host */
__global__ void k3_rebase(K3Params p)
{
	const int s = (int)blockIdx.x;
	/* channels that went through the serial machine because their candidates did not fit the tables (the host then
	 * shortens the parts it cuts pushes into): one lane per channel slot, not one load after the other */
	unsigned novf = 0, maxc = 0;
	if (blockIdx.x == 0) {	/* (the counters of the launch's streams: the other streams' words of this table set are zero) */
		for (int sc = (int)threadIdx.x; sc < (p.nstreams * VDL2_CS + 63) / 64 * 64; sc += 64) {
			const bool live = sc < p.nstreams * VDL2_CS && sc % VDL2_CS < p.nbch;
			const unsigned nc = live ? p.ctl[CTL_CAND0 + sc] : 0u;
			const bool over = live && (nc > VDL2_CAND_CAP || (p.ctl[CTL_CAND0 + p.nstreams * VDL2_CS + sc] & 1u));	/* (bit 1: unusable for another reason than their size) */
			novf += (unsigned)__popcll(__ballot(over));
			maxc = nc > maxc ? nc : maxc;
		}
		for (int d = 32; d > 0; d >>= 1) {	/* the busiest channel's candidate count: what the host sizes the next parts by */
			const unsigned o = (unsigned)__shfl_xor((int)maxc, d, 64);
			maxc = o > maxc ? o : maxc;
		}
	}
	if (threadIdx.x != 0)
		return;
	if (blockIdx.x == 0) {
		p.host_cnt[0] = p.outc[2 * p.ring];
		p.host_cnt[1] = p.outc[2 * p.ring + 1];
		p.host_cnt[2] = p.outc[8];	/* running totals: serial redos, repairs */
		p.host_cnt[3] = p.outc[9];
		for (int i = 0; i < 3; ++i)
			p.host_cnt[4 + i] = p.fcnt ? p.fcnt[i] : 0u;
		for (int i = 0; i < 16; ++i)
			p.host_cnt[8 + i] = p.fmask[i];
		p.host_cnt[7] = novf;
		p.host_cnt[24] = maxc;
	}
	StreamState *ss = p.ss + s;	/* (diagnostics: the kernels take the time base from their parameters) */
	ss->dec_base = ss->dec_base + ss->dec_fill + p.J - VDL2_CARRY_FRAMES;	/* frame VDL2_CARRY_FRAMES = first output of the next push */
	ss->dec_fill = VDL2_CARRY_FRAMES;
}

/* The push's burst records, device ring -> page-locked host memory, by the GPU itself at the end of the back stage: when
 * the host sees the push's completion event the records are already where vdl2gpu_poll*() hands them out from (round 2:
 * hipMemcpyAsync into a bounce buffer, a stream synchronise and two memcpy per record on the collecting thread -- 1.5 ms per
 * push with 3500 bursts, more than the GPU needed for the push). */
struct KExportParams {
	const vdl2gpu_burst_t *recs;
	const unsigned *count;	/* records written (device counter of this push's ring) */
	vdl2gpu_burst_t *dst;	/* device address of the ring's slab */
	unsigned cap;		/* records the slab holds */
};
/* Only what a record uses travels: the header and the rows of the burst (a record is eight rows of 255 bytes; a typical
 * burst fills one or two) -- a sixth of the PCIe traffic and of what the collecting thread reads; the host zero-fills the
 * rest when it hands the record out (vdl2gpu.hip: rec_copy).  One wavefront per record. */
__global__ __launch_bounds__(256)
void k_export_records(KExportParams p)
{
	static_assert(sizeof(vdl2gpu_burst_t) % 8 == 0 && offsetof(vdl2gpu_burst_t, data) % 8 == 0, "records are copied as 8-byte words");
	unsigned n = *p.count;
	n = n < p.cap ? n : p.cap;
	const unsigned lane = threadIdx.x & 63u;
	for (unsigned r = blockIdx.x * 4u + (threadIdx.x >> 6); r < n; r += gridDim.x * 4u) {
		int nb = p.recs[r].nbrow;
		nb = nb < 0 ? 0 : (nb > VDL2GPU_MAXROWS ? VDL2GPU_MAXROWS : nb);
		const unsigned words = (unsigned)((offsetof(vdl2gpu_burst_t, data) + (size_t)nb * VDL2GPU_ROWLEN + 7) / 8);
		const unsigned long long *src = reinterpret_cast<const unsigned long long *>(p.recs + r);
		unsigned long long *dst = reinterpret_cast<unsigned long long *>(p.dst + r);
		for (unsigned i = lane; i < words; i += 64u)
			dst[i] = src[i];
	}
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
