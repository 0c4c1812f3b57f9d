/* vdl2gpu_k3.h -- K3 and the small per-push kernels.  Part of the device side of libvdl2gpu.so; included by vdl2gpu_kernels.h only. */
#ifndef VDL2GPU_K3_H
#define VDL2GPU_K3_H

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
	if (threadIdx.x < 16)
		p.fmask[threadIdx.x] = 0u;
	if (p.fcnt && threadIdx.x < 4)
		p.fcnt[threadIdx.x] = 0u;
}

/* runs after k3_compact (same stream): publish the new time base and hand the push's counters to the host */
__global__ void k3_rebase(K3Params p)
{
	const int s = blockIdx.x;
	/* channels that went through the serial machine because their candidates did not fit the tables (the host then
	 * shortens the parts it cuts pushes into): one lane per channel slot, not one load after the other */
	unsigned novf = 0, maxc = 0;
	if (s == 0) {
		for (int sc = (int)threadIdx.x; sc < (p.nstreams * VDL2_CS + 63) / 64 * 64; sc += 64) {
			const bool live = sc < p.nstreams * VDL2_CS && sc % VDL2_CS < p.nbch;
			const unsigned nc = live ? p.ctl[CTL_CAND0 + sc] : 0u;
			const bool over = live && (nc > VDL2_CAND_CAP || p.ctl[CTL_CAND0 + p.nstreams * VDL2_CS + sc]);
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
	if (s == 0) {
		p.host_cnt[0] = p.outc[2 * p.ring];
		p.host_cnt[1] = p.outc[2 * p.ring + 1];
		p.host_cnt[2] = p.outc[4];
		p.host_cnt[3] = p.outc[5];
		for (int i = 0; i < 3; ++i)
			p.host_cnt[4 + i] = p.fcnt ? p.fcnt[i] : 0u;
		for (int i = 0; i < 16; ++i)
			p.host_cnt[8 + i] = p.fmask[i];
		p.host_cnt[7] = novf;
		p.host_cnt[24] = maxc;
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
