/*
 * vdl2gpu.hip -- host side of libvdl2gpu.so: the C ABI of include/vdl2gpu.h.
 *
 * One handle = one MI355X, three pushes in the pipeline.  vdl2gpu_push() enqueues
 *   front stage (fstream):     [H2D copy] -> K1 channelise -> K2a probe + K2x second stage / regions / region scan + K2x -> K2s sort -> carry for the next push
 *   back stage  (stream):      K2b clusters -> K2c resolve -> K2a verify + K2x (K2d payload beside it, on the copy stream)
 *   tail        (pay_stream):  repair round (merge, resolve, verify) -> commit -> re-resolved payloads -> export -> counters
 * and returns; the front stage of one push runs beside the back stage of the one before and the tail of the one before that
 * (see vdl2gpu::Back and enqueue_back); plane sets, table sets and output rings exist three times, the slabs four times.
 * vdl2gpu_poll() synchronises and hands burst records back in stream-time order.  There is no CPU fallback: without a HIP device
 * vdl2gpu_create() fails with VDL2GPU_ENODEV.
 *
 * Build (see __graft_entry__.build()):
 *   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared \
 *         vdl2gpu.hip -o libvdl2gpu.so
 */
#include "vdl2gpu_kernels.h"
#include "vdl2gpu_blocks.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

extern "C" void sincosf(float, float *, float *);

#define HIPCHK(h, expr)                                                                         \
	do {                                                                                    \
		hipError_t e__ = (expr);                                                        \
		if (e__ != hipSuccess) {                                                        \
			(h)->err = std::string(#expr) + ": " + hipGetErrorString(e__);          \
			return VDL2GPU_EHIP;                                                    \
		}                                                                               \
	} while (0)

#ifdef VDL2GPU_TESTHOOKS
static int g_test_item_grid = 0, g_test_item_common = 0, g_test_item_verify = 0;	/* read once per vdl2gpu_create (test build only) */
#endif

#define NEV 8	/* before K1 | K1 | probe+regions | K2b | K2c | verify | K2f+K2d | K3 */
#define NEVX 24	/* e[15] = the verify pass (round 0) has ended; + e[8], e[9] bracket the k1_fast launch alone; e[10] = start of the demodulator chain; e[12] = before the verify pass
			 * (main stream, behind the wait for the resolver); e[13], e[14] = around the resolver (its own stream when hoisted) */
struct PushTiming {
	hipEvent_t e[NEVX];	/* before K1, after K1, after K2a, after K2b, after K2c+K2d, after K3 */
	uint64_t samples;
	uint64_t index;		/* which push of the handle (VDL2GPU_STAGE_DUMP) */
	bool fast;
	bool staged;
	int fast_parts;		/* fast-kernel launches of this push (e[8]..e[9]); e[11] = end of the first period's general launch */
};

struct vdl2gpu {
	/* Every public call on the handle takes this lock (see "Threads" in include/vdl2gpu.h): the reference's shape -- the SDR
	 * library's callback thread committing blocks (rtl.c:274-295, 302) while another thread collects msgblk_t's -- works on one
	 * handle.  Recursive because public calls use each other (get_stats -> sync ...).  The blocking collectors release it while
	 * they wait for the GPU (wait_harvest). */
	std::recursive_mutex mu;
	vdl2gpu_config_t cfg;
	std::vector<vdl2gpu_chan_t> chans;
	int S, C, L, maxwin, sdrclk;
	size_t sample_bytes;
	long long cap;		/* frames per stream per ping-pong buffer */
	hipStream_t stream = nullptr;
	/* host samples: two staging buffers in HBM, filled on a stream of their own, so that the copy of one
	 * push runs beside the channeliser of the one before */
	hipStream_t in_stream = nullptr;
	hipEvent_t raw_copied[2] = {nullptr, nullptr};
	void *d_raw[2] = {nullptr, nullptr};
	size_t raw_bytes[2] = {0, 0};
	bool k1_rec[2] = {false, false};	/* k1_done[i] has been recorded at least once */
	/* ingest ring (rtl.c:274-295, air.c:191-217): pinned host slots the producer fills in place */
	void *ring_host = nullptr;
	size_t ring_slot_samples = 0, ring_slot_bytes = 0;
	int ring_nslots = 0;
	unsigned long long ring_next = 0;
	bool ring_acquired = false;
	std::vector<hipEvent_t> ring_copied;
	std::vector<char> ring_inflight;
	float2 *d_lo = nullptr;
	unsigned *d_k1_tickets = nullptr;	/* k1_fast's work counters */
	std::vector<unsigned> k1_tbase;	/* [S][8] what they hold, per stream and XCD (the same for every role; the same for every stream between two calls) */
	int last_set = 0;		/* the table / plane set of the last push (the debug calls read it) */
	float2 *d_lo_ext = nullptr;	/* [S][8][8 + L + 40]: every LO table with its last 8 entries in front and its first 40 behind (k1_pp reads 8 at a time) */
	float2 *d_dec[VDL2_NSET] = {};	/* plane sets, used in turn (push % 3): a set is written by the channeliser two pushes after its last
							 * reader, the back stage's tail, was ENQUEUED -- with two sets the channeliser had to wait for that tail */
	StreamState *d_ss = nullptr;
	ChanState *d_cs = nullptr;
	ChanCfg *d_cfg = nullptr;
	uint8_t *d_pn = nullptr;
	uint8_t *d_pn8 = nullptr;	/* ... by payload byte (K2Params.pn8) */
	vdl2gpu_burst_t *d_recs[VDL2_NRING] = {};	/* output rings, used in turn (push % 3): the calling thread waits for the
									 * ring's previous push only three pushes later -- with two rings it waited for the tail of the push
									 * before last in every call, and the GPU's front stream waited for the calling thread */
	unsigned *d_outc = nullptr;	/* [2*ring + {0,1}] = records written, dropped; [8], [9] running totals: serial redos, repairs */
	bool ring_busy[VDL2_NRING] = {};
	hipEvent_t ring_done2[VDL2_NRING][2] = {};	/* the end of the push's tail: its records and counters are on the host.  Two events per ring, used in
						 * turn (ring_ev[]): a collector that waits for one with the handle lock released (wait_harvest) would
						 * otherwise wait on an event the producer may re-record for the push three later */
	int ring_ev[VDL2_NRING] = {};		/* which of the two the ring's current push recorded */
#define ring_done(r) ring_done2[r][h->ring_ev[r]]
	hipEvent_t in_read[VDL2_NRING] = {};	/* its channeliser has read the caller's device buffer */
	bool in_rec[VDL2_NRING] = {};
	uint64_t ring_push[VDL2_NRING] = {};	/* which push filled the ring */
	hipStream_t copy_stream = nullptr;
	unsigned *d_ctl[VDL2_NSET] = {};	/* control words, see CTL_* in vdl2gpu_kernels.h */
	size_t ctl_words = 0;
	unsigned rec_cap = 0;
	Cand *d_cands[VDL2_NSET] = {};
	Cluster *d_clusters[VDL2_NSET] = {};
	int2 *d_clhead[VDL2_NSET] = {};
	BurstDesc *d_stage[VDL2_NSET] = {};
	unsigned *d_sel_list[VDL2_NSET] = {};
	unsigned *d_sel_list2[VDL2_NSET] = {};	/* the repair rounds' selection (K2Params.sel_list2) */
	int2 *d_regs[VDL2_NSET] = {};
	Seg *d_segs[VDL2_NSET] = {};
	int *d_fail[VDL2_NSET] = {};
	int *d_redo[VDL2_NSET] = {};
	ChanState *d_cs_out[VDL2_NSET] = {};
	int *d_skey[VDL2_NSET] = {};
	unsigned short *d_sidx[VDL2_NSET] = {}, *d_prim[VDL2_NSET] = {};
	int *d_seeds[VDL2_NSET] = {};
	uint8_t *d_onchain[VDL2_NSET] = {};	/* K2Params.onchain, .slog, .win: what a local repair stands on and what it leaves (k2p_patch) */
	K2Slog *d_slog[VDL2_NSET] = {};
	int2 *d_win[VDL2_NSET] = {};
	unsigned item_cap = VDL2_ITEM_CAP, item_priv = VDL2_ITEM_PRIV;	/* K2Params.item_cap; of which private areas at most (create_impl) */
	K2aItem *d_items[VDL2_NSET] = {};	/* what passed the scans' first screen (worked off by the scan workgroups themselves; the common area by the next kernel) */
	int full_scan = 0;
	unsigned stage_cap = 0;
	int prim_drop = 0;	/* VDL2GPU_PRIM_DROP (tests) */
#ifndef VDL2_K2D_GRID
#ifndef VDL2_K2D_GRID
#define VDL2_K2D_GRID 64
#endif
#endif
	int k2d_grid = VDL2_K2D_GRID;	/* payload workgroups per channel (VDL2GPU_K2D_GRID): 64 -- half the chip's wavefront slots at the kernel's 119 registers.
					 * One workgroup per burst is a chain of latencies; 128 per channel held EVERY slot while the verify pass beside it
					 * waited for them (0.472 against 0.462 ms per step at 64, same box; 32 and 16: 0.468, 0.469) */
	int force_serial = 0;
	int quirk = 0;		/* VDL2GPU_F_RTL_QUIRK */
	int n_cu = 256;
	int probe_occ = 4;	/* resident k2a_probe workgroups per CU */
	bool stage_events = true;	/* per-stage HIP events (vdl2gpu_timing_t breakdown): an event record between two kernels of the chain
					 * costs ~3 us, so only every stage_every-th push carries them (the sums are scaled up in harvest) */
	int stage_every = 4;
	bool stage_dump = false;	/* VDL2GPU_STAGE_DUMP=1: events on every push, their times printed when the push is collected (harvest_timing) */
	hipEvent_t ev_origin = nullptr;
	hipEvent_t k1_done[2] = {nullptr, nullptr};	/* per staging buffer (host input) */
	hipEvent_t k2_done[VDL2_NSET] = {};	/* per table set: the end of the tail of the push that used it */
	bool k2_rec[VDL2_NSET] = {};
	hipStream_t pay_stream = nullptr;	/* K2d beside the verify pass; then the push's TAIL (repair rounds, commit, export, counters: see enqueue_back) */
	hipEvent_t verify_done = nullptr, k2f_done = nullptr;	/* main -> tail: the verify pass has run; tail -> main: the channel states are committed */
	bool k2f_rec = false;
	hipStream_t tail_prev = nullptr;	/* the stream the previous push's tail ran on */
	hipEvent_t k2c_done = nullptr, pay_done = nullptr;
	unsigned *d_fmask[VDL2_NSET] = {};	/* K2f's redo mask of the push in flight, 16 words */
	bool ring_spec[VDL2_NRING] = {};	/* that ring's K2d ran ahead of verify: honour the redo mask */
	int repair_rounds = 0;		/* adapted floor..4 from how often the serial fallback was needed */
	int rounds_floor = 1;		/* one (resolver-only) repair round is always scheduled, see enqueue_back */
	size_t split_samples = 0;	/* pushes longer than this are cut into parts (36 s of air time), see push_checked; halved
					 * whenever a channel's candidate tables overflow */
	size_t split_default = 0;
	unsigned long long last_ovf_push = 0;
	size_t ring_samples[VDL2_NRING] = {};	/* samples (per stream) of the push that filled each output ring */
	double cand_dens[4] = {0, 0, 0, 0};	/* candidates per input sample of the busiest channel, last four parts collected */
	unsigned cand_dens_n = 0;
	size_t split_unit = 32768;	/* parts are multiples of this (k1_fast takes whole superperiods; the RTL quirk needs whole blocks) */
	unsigned redos_seen = 0, repairs_seen = 0;
	uint64_t last_redo_push = 0;
	unsigned long long *d_dbg = nullptr;
	HeadTap *d_headtap = nullptr;	/* VDL2GPU_F_DEBUG_HEADS: every trigger of the last push (any kernel) */
	unsigned *d_headtap_n = nullptr;
	unsigned headtap_cap = 0;
	uint64_t total_in = 0;		/* samples per stream pushed so far */
	uint64_t pushes = 0;
	uint64_t overflowed = 0;
	std::vector<PushTiming> pending;
	std::vector<PushTiming> free_ev;
	vdl2gpu_timing_t tm{};
	std::vector<vdl2gpu_burst_t> ready;	/* fetched, not yet handed out (storage order) */
	/* hand-out order, consumed from ready_pos: bits 32-33 = where the record lies (0: `ready`, 1 + ring: that ring's slab of
	 * page-locked host memory, which the GPU itself fills at the end of the push -- k_export_records -- so that collecting a
	 * push's bursts is an index sort and ONE copy per record, into the caller's buffer), bits 0-31 = index there */
	std::vector<uint64_t> ready_idx;
	vdl2gpu_burst_t *h_slab[VDL2_NSLAB] = {}, *d_slab[VDL2_NSLAB] = {};	/* the slabs (push % 4) and their device addresses:
									 * one more than rings, so that a ring collected at the last moment -- by the call that is about to reuse it -- still lies
									 * untouched in its slab while the caller polls once more, instead of being copied aside at once */
	int ring_slab[VDL2_NRING] = {};	/* the slab of the push that filled the ring */
	size_t slab_lo[VDL2_NSLAB] = {}, slab_hi[VDL2_NSLAB] = {};	/* ready_idx[lo, hi): where the handles into each slab lie (a push's are contiguous) */
	unsigned slab_cap = 0;
	size_t ready_pos = 0;
	/* block path in the pipeline (VDL2GPU_F_FRAMES) */
	bool frames_on = false;
	vdl2gpu_frame_t *d_frames[VDL2_NRING] = {};	/* byte buffers of compact entries */
	unsigned *d_k4tab = nullptr;	/* GF(256) and FCS tables of the block path */
	unsigned *d_fcnt = nullptr;	/* [4*ring] frames written, [4*ring+1] dropped, [4*ring+2] bytes used */
	unsigned frame_cap = 0;	/* bytes of a frame buffer (slots + arena) */

	std::vector<uint8_t> fready;		/* compact frame entries as k4_frames wrote them, storage order */
	std::vector<size_t> fready_idx;	/* hand-out order: byte offsets into `fready`, consumed from fready_pos */
	size_t fready_pos = 0;
	uint64_t frames_dropped = 0;
	vdl2gpu_burst_t *h_pin = nullptr;	/* pinned bounce buffer for record read-back */
	unsigned *h_pin_cnt = nullptr;	/* pinned, written by k3_rebase: [32*ring + {0..6}] counters, [7] overflowed channels, [8..23] redo mask, [24] most candidates of any channel */
	unsigned *d_pin_cnt = nullptr;	/* its device address */
	unsigned pin_recs = 0;
	bool failed = false;	/* a HIP call failed while work was being enqueued: device and host state no longer agree */
	std::string err;
	double hprof[8] = {0, 0, 0, 0, 0, 0, 0, 0};	/* VDL2GPU_HOST_PROF: seconds of the calling thread in the segments of push_impl (printed at destroy) */
	bool hprof_on = false;
	uint64_t hprof_push0 = 0;	/* pushes at the last reset of hprof[] */
	/* Environment knobs, read ONCE in create_impl (INTEGRATION.md lists them): push_impl never calls getenv.
	 * The test handicaps (VDL2GPU_PRIM_DROP, VDL2GPU_SPLIT_SAMPLES, VDL2GPU_F_TEST_NOREGION) exist only in the
	 * library built with -DVDL2GPU_TESTHOOKS (libvdl2gpu_test.so, which the tests load). */
	struct {
		bool no_k1_fast = false;	/* VDL2GPU_NO_K1_FAST: general channeliser only */
		bool no_tail = false;		/* VDL2GPU_NO_TAIL: everything behind the verify pass stays on the main stream */
		bool no_whole_pp = false;	/* VDL2GPU_NO_WHOLE_PP: 5/6/10 MS/s pushes always run their first and last period through the general channeliser (round 4) */
		double table_fill = 0.90;	/* VDL2GPU_TABLE_FILL (percent): how full the busiest channel's candidate table may get before parts are shortened */
		bool k1_pp = false;		/* VDL2GPU_K1_PP: k1_pp at 2 MS/s as well */
		bool debug_counters = false;	/* VDL2GPU_DEBUG_COUNTERS: cycle counters of the demodulator kernels */
		bool split_fixed = false;	/* (test hook) the part length was given: do not adapt it */
		int k1f_nfam = 0;		/* VDL2GPU_K1F_NFAM */
		int verify2_wg = 8;		/* VDL2GPU_VERIFY2_WG: workgroups per channel of the local repair round's verify pass */
		int k1_dbg = 0;			/* VDL2GPU_K1_DBG */
		int k1_nsub = 0;		/* VDL2GPU_K1_NSUB */
	} knob;
	/* A push's work is three stages on three streams, and three pushes are in the pipeline at once -- the planes, the tables
	 * (candidates, clusters, descriptors, control words ...) and the output rings exist three times:
	 *   FRONT (fstream): channeliser, scan of one class + regions, sort, carry copy -- wide kernels that need
	 *                    nothing of the previous push's RESULT: the scan starts at the first carried frame and in a fixed
	 *                    class (what the resolver consumes is decided by where it stands, not by where the scan began),
	 *                    the stream-time base is the host's arithmetic, the carry is a fixed 49152 frames;
	 *   BACK  (stream):  clusters, resolver, verify pass (+ payload decode beside it): the resolver needs the channel state
	 *                    the previous push's tail committed (k2f_done);
	 *   TAIL  (pay_stream): repair round, commit, block path, export, counters (enqueue_back).
	 * FRONT(N+1) runs beside BACK(N) and TAIL(N-1): the one-workgroup-per-channel kernels (resolver, commit ...) no longer
	 * leave the GPU idle, and the channeliser's planes are still in the Infinity Cache when the scan reads them.  Pushes too short
	 * for the parallel path (the live path: one SDR block) keep everything on the one stream: a hop costs ~30 us. */
	struct Back {
		bool valid = false;
		K2Params k2{};
		int64_t J = 0;
		int par = 0, ring = 0, slab = 0;
		bool staged = false, serial = false, two_streams = false;
		unsigned tiles = 0;
		size_t pt_index = 0;	/* its PushTiming in `pending` */
	} back;
	hipStream_t fstream = nullptr;
	hipEvent_t f_done[VDL2_NSET] = {};	/* FRONT of the push on that plane / table set has been enqueued up to its last kernel */
	hipEvent_t k1_ev = nullptr;	/* channeliser + carry copy of the latest push that kept to the main stream */
	hipEvent_t f_tail = nullptr;	/* the end of the latest front stage on fstream (carry copy included) */
	bool k1_ev_rec = false, last_two_streams = false;
	int64_t last_J = 0;	/* outputs of the previous push: where its last 49152 frames lie */
	/* stage sums of the pushes that carried stage events, unscaled, and how many those were */
	double st_scan = 0, st_cluster = 0, st_resolve = 0, st_demod = 0, st_other = 0, st_k1 = 0;
	uint64_t st_pushes = 0;
};

/* ------------------------------------------------------------ pure host helpers */
extern "C" unsigned int reversebits(const unsigned int bits, const int n)
{
	unsigned int r = 0;
	for (int i = 0; i < n; ++i)
		r |= ((bits >> i) & 1u) << (n - 1 - i);
	return r;
}

/* d8psk.c:353-357: wf[n] = cexpf(-n*Fo*I) with Fo narrowed to float.  cexpf of a
 * purely imaginary argument is (cos, sin) from libm's sincosf. */
extern "C" int vdl2gpu_lo_table(unsigned sdrinrate, int fo_hz, float *out_re_im, int max_complex)
{
	const int L = (int)(sdrinrate / 25000u);
	if (L <= 0 || L > max_complex)
		return VDL2GPU_EINVAL;
	const float w = (float)((double)((float)fo_hz / (float)sdrinrate) * 2.0 * M_PI);
	for (int n = 0; n < L; ++n) {
		const float y = (float)(-n) * w;
		float sn, cs;
		sincosf(y, &sn, &cs);
		out_re_im[2 * n] = cs;
		out_re_im[2 * n + 1] = sn;
	}
	return L;
}

/* Decimation schedule of one push, pure integer arithmetic:
 * given the number of samples already consumed, where does the push start in
 * the 21/SDRCLK clock, the LO period and the current integrate-and-dump window,
 * and how many 84 kS/s outputs complete inside it. */
extern "C" int vdl2gpu_plan(uint64_t total_in, uint64_t n, unsigned sdrclk, unsigned lo_len,
			    int *c0, int *no0, int *nf0, int64_t *nout)
{
	if (!sdrclk || !lo_len || sdrclk <= 21)
		return VDL2GPU_EINVAL;
	const unsigned __int128 t21 = (unsigned __int128)total_in * 21u;
	const uint64_t done = (uint64_t)(t21 / sdrclk);	/* outputs completed before this push */
	*c0 = (int)(uint64_t)(t21 % sdrclk);
	*no0 = (int)(total_in % lo_len);
	/* first input index after output (done-1): ceil(done*sdrclk/21) */
	const unsigned __int128 num = (unsigned __int128)done * sdrclk;
	const uint64_t first = (uint64_t)((num + 20) / 21);
	*nf0 = (int)(total_in - first);
	*nout = (int64_t)(((uint64_t)*c0 + 21ull * n) / sdrclk);
	return VDL2GPU_OK;
}

/* chooseFc() of rtl.c:123-160 (the tuner centre for a list of channel frequencies) and the mixer offsets of
 * rtl.c:245-247.  The reference walks Fc down in 1 Hz steps from max + 50 kHz to min - 50 kHz and takes the first
 * value at which every channel is within SDRINRATE/2 - 50 kHz of Fc, none is closer than 50 kHz, and no two
 * neighbouring (sorted) channels are mirror images; if none qualifies the loop ends with Fc = min - 50 kHz, which
 * the reference then uses (reproduced, not fixed).  Parity unpinned: rtl.c needs rtl-sdr.h. */
extern "C" int vdl2gpu_choose_fc_rtl(const unsigned *fr, int nbch, unsigned sdrinrate, unsigned *fc, int *fo)
{
	if (!fr || !fc || nbch < 1 || nbch > VDL2GPU_MAXCH || sdrinrate < 200000)
		return VDL2GPU_EINVAL;
	const int step = 25000;		/* STEPRATE, vdlm2.h:33 */
	unsigned fd[VDL2GPU_MAXCH];
	for (int n = 0; n < nbch; ++n)
		fd[n] = fr[n];
	std::sort(fd, fd + nbch);	/* rtl.c:128-140 */
	*fc = 0;
	if (fd[nbch - 1] - fd[0] > sdrinrate - 4u * step) {	/* rtl.c:142-145: "Frequencies too far apart" */
		if (fo)
			for (int n = 0; n < nbch; ++n)
				fo[n] = 0;
		return VDL2GPU_OK;
	}
	int c, n = 0;
	for (c = (int)fd[nbch - 1] + 2 * step; c > (int)fd[0] - 2 * step; --c) {	/* rtl.c:147-158, int arithmetic as there */
		for (n = 0; n < nbch; ++n) {
			if (std::abs(c - (int)fd[n]) > (int)(sdrinrate / 2) - 2 * step)
				break;
			if (std::abs(c - (int)fd[n]) < 2 * step)
				break;
			if (n > 0 && c - (int)fd[n - 1] == (int)fd[n] - c)
				break;
		}
		if (n == nbch)
			break;
	}
	*fc = (unsigned)c;
	if (fo)
		for (int i = 0; i < nbch; ++i)
			fo[i] = (int)fr[i] - c;	/* rtl.c:245-247 */
	return VDL2GPU_OK;
}

/* chooseFc() of air.c:47-70 and the mixer offsets of air.c:182-184.  Parity unpinned: air.c needs airspy.h. */
extern "C" int vdl2gpu_choose_fc_air(const unsigned *fr, int nbch, unsigned sdrinrate, unsigned *fc, int *fo, int *r10, int *r11)
{
	if (!fr || !fc || nbch < 1 || nbch > VDL2GPU_MAXCH)
		return VDL2GPU_EINVAL;
	static const unsigned hf[] = {1953050, 1980748, 2001344, 2032592, 2060291, 2087988};	/* r820t_hf, air.c:44 */
	static const unsigned lf[] = {525548, 656935, 795424, 898403, 1186034, 1502073, 1715133, 1853622};	/* r820t_lf, air.c:45 */
	const unsigned step = 25000;
	unsigned minf = 140000000u, maxf = 0;	/* air.c:77, 96-97 */
	for (int n = 0; n < nbch; ++n) {
		minf = std::min(minf, fr[n]);
		maxf = std::max(maxf, fr[n]);
	}
	const unsigned bw = maxf - minf + 2 * step;
	unsigned off = 0;
	if (r10)
		*r10 = 0;
	if (r11)
		*r11 = 0;
	*fc = 0;
	if (sdrinrate == 5000000) {	/* the R820T2 of the Airspy R2, air.c:53-66 */
		int i, j;
		for (i = 7; i >= 0; --i)
			if (hf[5] - lf[i] >= bw)
				break;
		if (i < 0) {	/* air.c:57: return 0 */
			if (fo)
				for (int n = 0; n < nbch; ++n)
					fo[n] = 0;
			return VDL2GPU_OK;
		}
		for (j = 5; j >= 0; --j)
			if (hf[j] - lf[i] <= bw)
				break;
		++j;
		if (j > 5)	/* cannot happen: hf[5] - lf[i] >= bw and the test is <=; only equality leaves j = 5 -> 6 */
			j = 5;
		off = (hf[j] + lf[i]) / 2 - sdrinrate / 4;
		if (r10)
			*r10 = 0xB0 | (15 - j);
		if (r11)
			*r11 = 0xE0 | (15 - i);
	}
	*fc = ((maxf + minf) / 2 + off + step / 2) / step * step;	/* air.c:69 */
	if (fo) {
		const unsigned f0 = *fc + sdrinrate / 4;	/* air.c:182 */
		for (int n = 0; n < nbch; ++n)
			fo[n] = (int)(fr[n] - f0);
	}
	return VDL2GPU_OK;
}

/* stream time (84 kS/s index) -> index of the input sample that completed it */
static int64_t dec_to_sample(int64_t m, unsigned sdrclk)
{
	if (m < 0)
		return m;
	return (int64_t)((((unsigned __int128)(uint64_t)(m + 1)) * sdrclk + 20) / 21) - 1;
}

extern "C" int vdl2gpu_abi_version(void)
{
	return VDL2GPU_ABI_VERSION;
}

extern "C" const char *vdl2gpu_strerror(int code)
{
	switch (code) {
	case VDL2GPU_OK: return "ok";
	case VDL2GPU_EINVAL: return "invalid argument";
	case VDL2GPU_EHIP: return "HIP runtime error";
	case VDL2GPU_ENOMEM: return "out of memory";
	case VDL2GPU_EOVERFLOW: return "burst record ring overflow";
	case VDL2GPU_ENODEV: return "no HIP device (there is no CPU fallback)";
	default: return "unknown error";
	}
}

#define HLOCK(h) std::unique_lock<std::recursive_mutex> hlock_((h)->mu)

extern "C" const char *vdl2gpu_last_error(vdl2gpu_t *h)
{
	if (!h)
		return "null handle";
	HLOCK(h);
	return h->err.c_str();	/* (valid until the next call on the handle, from any thread) */
}

/* msgblk_t, vdlm2.h:39-47, LP64: prev@0(8) chn@8 Fr@12 tv@16(16) ppm@32 nbrow@36 nlbyte@40 data@44 */
extern "C" int vdl2gpu_burst_to_msgblk(const vdl2gpu_burst_t *b, void *msgblk, size_t msgblk_size)
{
	if (!b || !msgblk || msgblk_size < VDL2GPU_MSGBLK_SIZE)
		return VDL2GPU_EINVAL;
	char *m = (char *)msgblk;	/* the offsets are held against the reference's own header at build time (see VDL2GPU_MSGBLK_SIZE in include/vdl2gpu.h) */
	memcpy(m + VDL2GPU_MSGBLK_OFF_CHN, &b->chn, 4);
	memcpy(m + VDL2GPU_MSGBLK_OFF_FR, &b->Fr, 4);
	memcpy(m + VDL2GPU_MSGBLK_OFF_PPM, &b->ppm, 4);
	memcpy(m + VDL2GPU_MSGBLK_OFF_NBROW, &b->nbrow, 4);
	memcpy(m + VDL2GPU_MSGBLK_OFF_NLBYTE, &b->nlbyte, 4);
	memcpy(m + VDL2GPU_MSGBLK_OFF_DATA, b->data, VDL2GPU_MAXROWS * VDL2GPU_ROWLEN);
	return VDL2GPU_OK;
}

/* ------------------------------------------------------------------- lifecycle */
static size_t fmt_bytes(int fmt)
{
	switch (fmt) {
	case VDL2GPU_FMT_CU8: return 2;
	case VDL2GPU_FMT_CS16: return 4;
	case VDL2GPU_FMT_CF32: return 8;
	case VDL2GPU_FMT_F32R: return 4;
	default: return 0;
	}
}

extern "C" void vdl2gpu_destroy(vdl2gpu_t *h)
{
	if (h && h->hprof_on && h->pushes)
		fprintf(stderr, "vdl2gpu host profile, ms per push over %llu pushes: wait for the buffer of the push before last %.3f, channeliser enqueue %.3f, "
				"collect the ring %.3f, front stage enqueue %.3f, spill %.3f, back stage enqueue %.3f; waiting for rings' events (push and poll calls) %.3f\n", (unsigned long long)h->pushes,
			h->hprof[0] / h->pushes * 1e3, h->hprof[1] / h->pushes * 1e3, h->hprof[2] / h->pushes * 1e3, h->hprof[3] / h->pushes * 1e3,
			h->hprof[4] / h->pushes * 1e3, h->hprof[5] / h->pushes * 1e3, h->hprof[6] / h->pushes * 1e3);
#ifdef K1F_PROF
	{
		static unsigned raw[K1F_PROF_SLOTS][12];
		(void)hipDeviceSynchronize();
		if (hipMemcpyFromSymbol(raw, HIP_SYMBOL(k1f_prof), sizeof raw) == hipSuccess) {
			double pf[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
			unsigned first = 0xffffffffu, last_start = 0, last_end = 0;
			int nwaves = 0;
			for (int b = 0; b < K1F_PROF_SLOTS; ++b) {
				if (!raw[b][7])
					continue;
				++nwaves;
				for (int i = 0; i < 9; ++i)
					pf[i] += raw[b][i];
				first = std::min(first, raw[b][9]);
				last_start = std::max(last_start, raw[b][9]);
				last_end = std::max(last_end, raw[b][9] + raw[b][8]);
			}
			const char *nm[7] = {"prologue", "wait samples", "convert+park+issue", "barrier", "mix+divide", "store issue", "drain"};
			if (pf[7] > 0) {
				double cyc = 0;
				for (int i = 0; i < 7; ++i)
					cyc += pf[i];
				fprintf(stderr, "k1_fast phases, shader cycles per wavefront-iteration (%.0f wavefront-iterations, %d wavefronts in the last launch):", pf[7], nwaves);
				for (int i = 0; i < 7; ++i)
					fprintf(stderr, " %s %.0f;", nm[i], pf[i] / pf[7]);
				{
					double a = 0, b = 0;
					for (int bb = 0; bb < K1F_PROF_SLOTS; ++bb)
						if (raw[bb][7]) {
							a += raw[bb][10];
							b += raw[bb][11];
						}
					cyc += a + b;
					fprintf(stderr, " [prologue per wavefront: window table %.0f, addresses + first loads %.0f, LO values + wait %.0f]", a / nwaves, b / nwaves, pf[0] / nwaves);
				}
				fprintf(stderr, " shader clock %.0f MHz; a wavefront lives %.1f us; first start to last start %.1f us, to last end %.1f us\n",
					cyc / (pf[8] / 100.0), pf[8] / nwaves / 100.0, (last_start - first) / 100.0, (last_end - first) / 100.0);
				std::vector<unsigned> life;
				double by_xcd[8] = {0}, by_role[K1F_ROLES] = {0}, by_wv[2] = {0};
				int n_xcd[8] = {0}, n_role[K1F_ROLES] = {0}, n_wv[2] = {0};
				for (int b = 0; b < K1F_PROF_SLOTS; ++b) {
					if (!raw[b][7])
						continue;
					life.push_back(raw[b][8]);
					const int blk = b / 2;
					by_xcd[blk & 7] += raw[b][8]; ++n_xcd[blk & 7];
					by_role[(blk >> 3) % K1F_ROLES] += raw[b][8]; ++n_role[(blk >> 3) % K1F_ROLES];
					by_wv[b & 1] += raw[b][8]; ++n_wv[b & 1];
				}
				std::sort(life.begin(), life.end());
				fprintf(stderr, "   lifetime us: min %.1f p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f\n   by XCD:", life[0] / 100.0, life[life.size() / 10] / 100.0,
					life[life.size() / 2] / 100.0, life[life.size() * 9 / 10] / 100.0, life[life.size() * 99 / 100] / 100.0, life.back() / 100.0);
				for (int i = 0; i < 8; ++i)
					fprintf(stderr, " %.1f", by_xcd[i] / std::max(1, n_xcd[i]) / 100.0);
				fprintf(stderr, "\n   by role:");
				for (int i = 0; i < K1F_ROLES; ++i)
					fprintf(stderr, " %.1f", by_role[i] / std::max(1, n_role[i]) / 100.0);
				fprintf(stderr, "\n   by wavefront of the workgroup: %.1f %.1f\n", by_wv[0] / std::max(1, n_wv[0]) / 100.0, by_wv[1] / std::max(1, n_wv[1]) / 100.0);
			}
		}
	}
#endif
	if (!h)
		return;
	(void)hipSetDevice(h->cfg.device);
	if (h->in_stream)
		(void)hipStreamSynchronize(h->in_stream);
	if (h->fstream)
		(void)hipStreamSynchronize(h->fstream);
	if (h->stream)
		(void)hipStreamSynchronize(h->stream);
	if (h->pay_stream)
		(void)hipStreamSynchronize(h->pay_stream);
	if (h->copy_stream)
		(void)hipStreamSynchronize(h->copy_stream);
	for (auto &pt : h->pending)
		for (auto &e : pt.e)
			(void)hipEventDestroy(e);
	for (auto &pt : h->free_ev)
		for (auto &e : pt.e)
			(void)hipEventDestroy(e);
	(void)hipFree(h->d_raw[0]);
	(void)hipFree(h->d_raw[1]);
	if (h->ring_host)
		(void)hipHostFree(h->ring_host);
	for (hipEvent_t e : h->ring_copied)
		(void)hipEventDestroy(e);
	for (int i = 0; i < 2; ++i)
		if (h->raw_copied[i])
			(void)hipEventDestroy(h->raw_copied[i]);
	if (h->in_stream)
		(void)hipStreamDestroy(h->in_stream);
	(void)hipFree(h->d_lo);
	(void)hipFree(h->d_lo_ext);
	(void)hipFree(h->d_k1_tickets);
	for (int r = 0; r < VDL2_NSET; ++r)
		(void)hipFree(h->d_dec[r]);
	(void)hipFree(h->d_ss);
	(void)hipFree(h->d_cs);
	(void)hipFree(h->d_cfg);
	(void)hipFree(h->d_pn);
	(void)hipFree(h->d_pn8);
	for (int r = 0; r < VDL2_NRING; ++r) {
		(void)hipFree(h->d_recs[r]);
		(void)hipFree(h->d_frames[r]);
	}
	(void)hipFree(h->d_fcnt);
	(void)hipFree(h->d_k4tab);
	(void)hipFree(h->d_outc);
	if (h->copy_stream)
		(void)hipStreamDestroy(h->copy_stream);
	for (int r = 0; r < 2; ++r)
		if (h->k1_done[r])
			(void)hipEventDestroy(h->k1_done[r]);
	for (int r = 0; r < VDL2_NSET; ++r)
		if (h->k2_done[r])
			(void)hipEventDestroy(h->k2_done[r]);
	for (int r = 0; r < VDL2_NRING; ++r) {
		for (int k = 0; k < 2; ++k)
			if (h->ring_done2[r][k])
				(void)hipEventDestroy(h->ring_done2[r][k]);
		if (h->in_read[r])
			(void)hipEventDestroy(h->in_read[r]);
	}
	if (h->fstream) {
		(void)hipStreamSynchronize(h->fstream);
		(void)hipStreamDestroy(h->fstream);
	}
	for (int r = 0; r < VDL2_NSET; ++r)
		if (h->f_done[r])
			(void)hipEventDestroy(h->f_done[r]);
	if (h->k1_ev)
		(void)hipEventDestroy(h->k1_ev);
	if (h->f_tail)
		(void)hipEventDestroy(h->f_tail);
	if (h->pay_stream) {
		(void)hipStreamSynchronize(h->pay_stream);
		(void)hipStreamDestroy(h->pay_stream);
	}
	if (h->k2c_done)
		(void)hipEventDestroy(h->k2c_done);
	if (h->pay_done)
		(void)hipEventDestroy(h->pay_done);
	if (h->verify_done)
		(void)hipEventDestroy(h->verify_done);
	if (h->k2f_done)
		(void)hipEventDestroy(h->k2f_done);

	if (h->ev_origin)
		(void)hipEventDestroy(h->ev_origin);
	for (int r = 0; r < VDL2_NSET; ++r)
		(void)hipFree(h->d_fmask[r]);
	for (int r = 0; r < VDL2_NSET; ++r)
		(void)hipFree(h->d_ctl[r]);
	for (int r = 0; r < VDL2_NSET; ++r)
		(void)hipFree(h->d_cands[r]);
	for (int r = 0; r < VDL2_NSET; ++r)
		(void)hipFree(h->d_clusters[r]);
	for (int r = 0; r < VDL2_NSET; ++r)
		(void)hipFree(h->d_clhead[r]);
	for (int r = 0; r < VDL2_NSET; ++r)
		(void)hipFree(h->d_stage[r]);
	for (int r = 0; r < VDL2_NSET; ++r)
		(void)hipFree(h->d_sel_list[r]);
	for (int r = 0; r < VDL2_NSET; ++r)
		(void)hipFree(h->d_sel_list2[r]);
	for (int r = 0; r < VDL2_NSET; ++r)
		(void)hipFree(h->d_regs[r]);
	for (int r = 0; r < VDL2_NSET; ++r)
		(void)hipFree(h->d_segs[r]);
	for (int r = 0; r < VDL2_NSET; ++r)
		(void)hipFree(h->d_fail[r]);
	for (int r = 0; r < VDL2_NSET; ++r)
		(void)hipFree(h->d_redo[r]);
	for (int r = 0; r < VDL2_NSET; ++r)
		(void)hipFree(h->d_cs_out[r]);
	for (int r = 0; r < VDL2_NSET; ++r)
		(void)hipFree(h->d_skey[r]);
	for (int r = 0; r < VDL2_NSET; ++r)
		(void)hipFree(h->d_sidx[r]);
	for (int r = 0; r < VDL2_NSET; ++r)
		(void)hipFree(h->d_prim[r]);
	for (int r = 0; r < VDL2_NSET; ++r) {
		(void)hipFree(h->d_seeds[r]);
		(void)hipFree(h->d_onchain[r]);
		(void)hipFree(h->d_slog[r]);
		(void)hipFree(h->d_win[r]);
	}
	for (int r = 0; r < VDL2_NSET; ++r)
		(void)hipFree(h->d_items[r]);
	(void)hipFree(h->d_dbg);
	(void)hipFree(h->d_headtap);
	(void)hipFree(h->d_headtap_n);
	if (h->h_pin)
		(void)hipHostFree(h->h_pin);
	for (int r = 0; r < VDL2_NSLAB; ++r)
		if (h->h_slab[r])
			(void)hipHostFree(h->h_slab[r]);
	if (h->h_pin_cnt)
		(void)hipHostFree(h->h_pin_cnt);
	if (h->stream)
		(void)hipStreamDestroy(h->stream);
	delete h;
}

static int create_impl(vdl2gpu_t *h)
{
	const vdl2gpu_config_t &cfg = h->cfg;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg.device >= ndev) {
		h->err = "no HIP device";
		return VDL2GPU_ENODEV;
	}
	HIPCHK(h, hipSetDevice(cfg.device));
	{
		hipDeviceProp_t prop;
		if (hipGetDeviceProperties(&prop, cfg.device) == hipSuccess && prop.multiProcessorCount > 0)
			h->n_cu = prop.multiProcessorCount;
		int occ = 0;
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k2a_probe, K2A_THREADS, 0) == hipSuccess && occ > 0)
			h->probe_occ = occ;
	}
	{
		int prio_lo = 0, prio_hi = 0;
		HIPCHK(h, hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
		HIPCHK(h, hipStreamCreateWithPriority(&h->stream, hipStreamNonBlocking, prio_hi));
	}
	const int S = h->S, L = h->L;
	h->k1_tbase.assign((size_t)S * 8, 0u);
	const long long jmax = (long long)((21ull * cfg.max_push) / (unsigned)h->sdrclk) + 2;
	h->cap = (VDL2_CARRY_FRAMES + jmax + 64 + 15) / 16 * 16;	/* planes start on 128-byte lines */
	{
		/* The item lists (what passes a scan's first screen: 80 bytes an item, three sets) by the longest PART the handle can be given
		 * -- max_push, or what push_checked cuts longer pushes into (36 s of air time, a third more in a test build): 64 items of private
		 * areas per 1024-instant tile (the verify pass's workgroups take four tiles and an area of 256 each; the probe's 42 a tile), the
		 * common area half of that again.  A 67 MS push at 2 MS/s keeps round 5's 131 072 + 65 536 items per channel (126 MB a set and
		 * stream); a handle for pushes of a few MS 32 768 + 32 768 (42 MB). */
		const long long jcap = 48LL * 84000;
		const long long tiles_max = (VDL2_CARRY_FRAMES + std::min(jmax, jcap)) / K2A_TS + 2;
		const unsigned priv = (unsigned)std::min<long long>(VDL2_ITEM_PRIV, std::max<long long>(32768, (64 * tiles_max + 4095) / 4096 * 4096));
		h->item_priv = priv;
		h->item_cap = priv + std::max(priv / 2, 32768u);	/* (the common area: what a stretch of sync words or a carrier sends past the private areas) */
#ifdef VDL2_ITEMS_FULL_VALUES
		h->item_priv = VDL2_ITEM_PRIV;
		h->item_cap = VDL2_ITEM_CAP;
#endif
	}
	const size_t dec_bytes = (size_t)S * (size_t)h->cap * VDL2_CS * sizeof(float2);
	for (int r = 0; r < VDL2_NSET; ++r) {
		HIPCHK(h, hipMalloc(&h->d_dec[r], dec_bytes));
		HIPCHK(h, hipMemsetAsync(h->d_dec[r], 0, dec_bytes, h->stream));
	}
	HIPCHK(h, hipMalloc(&h->d_lo, (size_t)S * VDL2_CS * L * sizeof(float2)));
	HIPCHK(h, hipMalloc(&h->d_lo_ext, (size_t)S * VDL2_CS * (L + 48) * sizeof(float2)));
	HIPCHK(h, hipMalloc(&h->d_k1_tickets, (size_t)S * 21 * 8 * sizeof(unsigned)));
	HIPCHK(h, hipMemsetAsync(h->d_k1_tickets, 0, (size_t)S * 21 * 8 * sizeof(unsigned), h->stream));
	HIPCHK(h, hipMalloc(&h->d_ss, (size_t)S * sizeof(StreamState)));
	HIPCHK(h, hipMalloc(&h->d_cs, (size_t)S * VDL2_CS * sizeof(ChanState)));
	HIPCHK(h, hipMalloc(&h->d_cfg, (size_t)S * VDL2_CS * sizeof(ChanCfg)));
	HIPCHK(h, hipMalloc(&h->d_pn, VDL2_PN_BITS));
	for (int r = 0; r < VDL2_NRING; ++r) {
		HIPCHK(h, hipMalloc(&h->d_recs[r], (size_t)h->rec_cap * sizeof(vdl2gpu_burst_t)));
	}
	HIPCHK(h, hipMalloc(&h->d_outc, 16 * sizeof(unsigned)));
	HIPCHK(h, hipMemsetAsync(h->d_outc, 0, 16 * sizeof(unsigned), h->stream));
	HIPCHK(h, hipStreamCreateWithPriority(&h->copy_stream, hipStreamNonBlocking, 0));
	/* (HIP multiplexes its streams onto four hardware queues: a fifth stream shares one with another, and kernels that
	 * were meant to run side by side then run one behind the other -- this handle makes exactly main, copy, resolver, payload;
	 * host input adds one for its copies, which may share a queue with the record read-back) */
	for (int r = 0; r < 2; ++r)
		HIPCHK(h, hipEventCreateWithFlags(&h->k1_done[r], hipEventDisableTiming));
	for (int r = 0; r < VDL2_NSET; ++r)
		HIPCHK(h, hipEventCreateWithFlags(&h->k2_done[r], hipEventDisableTiming));
	for (int r = 0; r < VDL2_NRING; ++r) {
		for (int k = 0; k < 2; ++k)
			HIPCHK(h, hipEventCreateWithFlags(&h->ring_done2[r][k], hipEventDisableTiming));
		HIPCHK(h, hipEventCreateWithFlags(&h->in_read[r], hipEventDisableTiming));
	}
	{
		int prio_lo = 0, prio_hi = 0;
		HIPCHK(h, hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
		HIPCHK(h, hipStreamCreateWithPriority(&h->fstream, hipStreamNonBlocking, prio_lo));	/* the back stage (main stream, high priority) is the shorter one: it goes first */
	}
	for (int r = 0; r < VDL2_NSET; ++r)
		HIPCHK(h, hipEventCreateWithFlags(&h->f_done[r], hipEventDisableTiming));
	HIPCHK(h, hipEventCreateWithFlags(&h->k1_ev, hipEventDisableTiming));
	HIPCHK(h, hipEventCreateWithFlags(&h->f_tail, hipEventDisableTiming));
	HIPCHK(h, hipStreamCreateWithFlags(&h->pay_stream, hipStreamNonBlocking));
	HIPCHK(h, hipEventCreateWithFlags(&h->k2c_done, hipEventDisableTiming));
	HIPCHK(h, hipEventCreateWithFlags(&h->pay_done, hipEventDisableTiming));
	HIPCHK(h, hipEventCreateWithFlags(&h->verify_done, hipEventDisableTiming));
	HIPCHK(h, hipEventCreateWithFlags(&h->k2f_done, hipEventDisableTiming));

	for (int r = 0; r < VDL2_NSET; ++r)
		HIPCHK(h, hipMalloc(&h->d_fmask[r], 16 * sizeof(unsigned)));
	for (int r = 0; r < VDL2_NSET; ++r)
		HIPCHK(h, hipMemsetAsync(h->d_fmask[r], 0, 16 * sizeof(unsigned), h->stream));
	h->ctl_words = VDL2_CTL_WORDS((size_t)S * VDL2_CS);
	for (int r = 0; r < VDL2_NSET; ++r)
		HIPCHK(h, hipMalloc(&h->d_ctl[r], h->ctl_words * sizeof(unsigned)));
	for (int r = 0; r < VDL2_NSET; ++r)
		HIPCHK(h, hipMemsetAsync(h->d_ctl[r], 0, h->ctl_words * sizeof(unsigned), h->stream));
	for (int r = 0; r < VDL2_NSET; ++r)
		HIPCHK(h, hipMalloc(&h->d_cands[r], (size_t)S * VDL2_CS * VDL2_CAND_CAP * sizeof(Cand)));
	for (int r = 0; r < VDL2_NSET; ++r)
		HIPCHK(h, hipMalloc(&h->d_clusters[r], (size_t)S * VDL2_CS * VDL2_CAND_CAP * sizeof(Cluster)));
	for (int r = 0; r < VDL2_NSET; ++r)
		HIPCHK(h, hipMalloc(&h->d_clhead[r], (size_t)S * VDL2_CS * VDL2_CAND_CAP * sizeof(int2)));
	h->stage_cap = (unsigned)S * VDL2_CS * VDL2_CAND_CAP * VDL2_CL_MAXB + 65536u;	/* static slots + dynamic tail */
	for (int r = 0; r < VDL2_NSET; ++r)
		HIPCHK(h, hipMalloc(&h->d_stage[r], (size_t)h->stage_cap * sizeof(BurstDesc)));
	for (int r = 0; r < VDL2_NSET; ++r)
		HIPCHK(h, hipMalloc(&h->d_sel_list[r], (size_t)S * VDL2_CS * VDL2_SEL_CAP * sizeof(unsigned)));
	for (int r = 0; r < VDL2_NSET; ++r)
		HIPCHK(h, hipMalloc(&h->d_sel_list2[r], (size_t)S * VDL2_CS * VDL2_SEL_CAP * sizeof(unsigned)));
	for (int r = 0; r < VDL2_NSET; ++r)
		HIPCHK(h, hipMalloc(&h->d_regs[r], (size_t)S * VDL2_CS * VDL2_REG_CAP * sizeof(int2)));
	for (int r = 0; r < VDL2_NSET; ++r)
		HIPCHK(h, hipMalloc(&h->d_segs[r], (size_t)S * VDL2_CS * VDL2_SEG_CAP * sizeof(Seg)));
	for (int r = 0; r < VDL2_NSET; ++r)
		HIPCHK(h, hipMalloc(&h->d_fail[r], (size_t)S * VDL2_CS * sizeof(int)));
	for (int r = 0; r < VDL2_NSET; ++r)
		HIPCHK(h, hipMalloc(&h->d_redo[r], (size_t)S * VDL2_CS * sizeof(int)));
	for (int r = 0; r < VDL2_NSET; ++r)
		HIPCHK(h, hipMalloc(&h->d_cs_out[r], (size_t)S * VDL2_CS * sizeof(ChanState)));
	for (int r = 0; r < VDL2_NSET; ++r)
		HIPCHK(h, hipMalloc(&h->d_skey[r], (size_t)S * VDL2_CS * VDL2_CAND_CAP * sizeof(int)));
	for (int r = 0; r < VDL2_NSET; ++r)
		HIPCHK(h, hipMalloc(&h->d_sidx[r], (size_t)S * VDL2_CS * VDL2_CAND_CAP * sizeof(unsigned short)));
	for (int r = 0; r < VDL2_NSET; ++r)
		HIPCHK(h, hipMalloc(&h->d_prim[r], (size_t)S * VDL2_CS * VDL2_CAND_CAP * sizeof(unsigned short)));
	for (int r = 0; r < VDL2_NSET; ++r)
		HIPCHK(h, hipMalloc(&h->d_seeds[r], (size_t)S * VDL2_CS * VDL2_CAND_CAP * sizeof(int)));
	for (int r = 0; r < VDL2_NSET; ++r) {
		HIPCHK(h, hipMalloc(&h->d_onchain[r], (size_t)S * VDL2_CS * VDL2_CAND_CAP));
		HIPCHK(h, hipMalloc(&h->d_slog[r], (size_t)S * VDL2_CS * VDL2_SLOG_CAP * sizeof(K2Slog)));
		HIPCHK(h, hipMalloc(&h->d_win[r], (size_t)S * VDL2_CS * VDL2_WIN_CAP * sizeof(int2)));
	}
	for (int r = 0; r < VDL2_NSET; ++r)
#ifdef VDL2_ITEMS_ALLOC_FULL
		HIPCHK(h, hipMalloc(&h->d_items[r], (size_t)S * VDL2_CS * VDL2_ITEM_CAP * sizeof(K2aItem)));
#else
		HIPCHK(h, hipMalloc(&h->d_items[r], (size_t)S * VDL2_CS * h->item_cap * sizeof(K2aItem)));
#endif
	/* every environment knob is read here, once */
	auto env_int = [](const char *name, int dflt) { const char *v = getenv(name); return v ? atoi(v) : dflt; };
	h->full_scan = ((cfg.flags & VDL2GPU_F_FULLSCAN) || getenv("VDL2GPU_FULL_SCAN")) ? 1 : 0;
	h->stage_every = std::max(1, env_int("VDL2GPU_STAGE_EVERY", h->stage_every));
	h->stage_dump = env_int("VDL2GPU_STAGE_DUMP", 0) != 0;
	if (h->stage_dump)
		h->stage_every = 1;
	h->k2d_grid = std::max(1, env_int("VDL2GPU_K2D_GRID", h->k2d_grid));
	h->knob.no_k1_fast = getenv("VDL2GPU_NO_K1_FAST") != nullptr;
	h->hprof_on = getenv("VDL2GPU_HOST_PROF") != nullptr;
	h->knob.no_tail = getenv("VDL2GPU_NO_TAIL") != nullptr;
	h->knob.no_whole_pp = getenv("VDL2GPU_NO_WHOLE_PP") != nullptr;
	h->knob.table_fill = std::min(100, std::max(10, env_int("VDL2GPU_TABLE_FILL", 90))) / 100.0;
	h->knob.k1_pp = getenv("VDL2GPU_K1_PP") != nullptr;
	h->knob.debug_counters = getenv("VDL2GPU_DEBUG_COUNTERS") != nullptr;
	h->knob.k1f_nfam = env_int("VDL2GPU_K1F_NFAM", 0);
	h->knob.verify2_wg = std::max(1, env_int("VDL2GPU_VERIFY2_WG", 8));
	h->knob.k1_dbg = env_int("VDL2GPU_K1_DBG", 0);
	h->knob.k1_nsub = env_int("VDL2GPU_K1_NSUB", 0);
#ifdef VDL2GPU_TESTHOOKS
	h->prim_drop = env_int("VDL2GPU_PRIM_DROP", 0);
	g_test_item_grid = env_int("VDL2GPU_TEST_ITEM_GRID", 0);
	g_test_item_common = env_int("VDL2GPU_TEST_ITEM_COMMON", 0);
	g_test_item_verify = env_int("VDL2GPU_TEST_ITEM_VERIFY", 0);	/* the two above for the verify passes only */
#endif
	/* A push in which a channel's verify pass fails with no round scheduled costs a serial redo of that channel's whole
	 * push (milliseconds), an idle round 30 us: with 16 channels or more an event somewhere is frequent enough that one
	 * round is always scheduled. */
	/* Parts (see push_checked): at most 36 s of air time; until the first pushes have been collected and their candidate
	 * density is known, 8.4 s -- a saturated channel (250 candidates a second) fills half of the tables in that long. */
	h->split_unit = ((cfg.flags & VDL2GPU_F_RTL_QUIRK) || h->sdrclk != 500 || h->L != 80) ? 32768 : K1F_PER_IN;
	/* other rates: whole periods of the dump schedule (4 * SDRCLK samples), so that a part that starts on a schedule boundary is ONE
	 * k1_pp launch like a whole push (push_impl: whole_pp) -- unless the quirk wants whole 32768-sample blocks */
	if (!(cfg.flags & VDL2GPU_F_RTL_QUIRK) && h->split_unit == 32768 && (4 * h->sdrclk) % h->L == 0 && ((size_t)4 * h->sdrclk * h->sample_bytes) % 16 == 0)
		h->split_unit = (size_t)4 * h->sdrclk * ((32768 + (size_t)4 * h->sdrclk - 1) / ((size_t)4 * h->sdrclk));	/* (about the block's size) */
	h->split_default = (size_t)(36.0 * (double)h->cfg.sdrinrate) / h->split_unit * h->split_unit;
	{
		/* the verify pass maps one workgroup to K2A_VRUN tiles and the item list has room for VDL2_MAXWG private areas: a part
		 * must not have more tiles than that covers (36 s of air time are 3003 tiles, the bound is 4088) */
		const double max_frames = ((double)VDL2_MAXWG * K2A_VRUN - 4.0) * 2.0 * K2A_TS - 2.0 * K2A_TS - (double)VDL2_CARRY_FRAMES;
		const size_t max_part = (size_t)(max_frames * (double)h->sdrclk / 21.0) / h->split_unit * h->split_unit;
		h->split_default = std::min(h->split_default, max_part);
	}
	h->split_samples = std::max(h->split_unit, (size_t)(8.4 * (double)h->cfg.sdrinrate) / h->split_unit * h->split_unit);
#ifdef VDL2GPU_TESTHOOKS
	if (getenv("VDL2GPU_SPLIT_SAMPLES")) {
		h->split_samples = std::min((size_t)atoll(getenv("VDL2GPU_SPLIT_SAMPLES")), h->split_default * 4 / 3);	/* (still inside the verify grid's bound) */
		h->knob.split_fixed = true;
	}
#endif
	h->rounds_floor = 1;	/* see enqueue_back */
	h->repair_rounds = env_int("VDL2GPU_REPAIR_ROUNDS", h->rounds_floor);
	h->force_serial = (cfg.flags & VDL2GPU_F_SERIAL) ? 1 : 0;
	h->quirk = (cfg.flags & VDL2GPU_F_RTL_QUIRK) ? 1 : 0;
	h->pin_recs = std::min<unsigned>(h->rec_cap, 8192u);
	HIPCHK(h, hipHostMalloc(&h->h_pin, (size_t)h->pin_recs * sizeof(vdl2gpu_burst_t), hipHostMallocDefault));
	h->slab_cap = std::min<unsigned>(h->rec_cap, 16384u);	/* 34 MB of page-locked memory per slab at most (four slabs); a push with more bursts takes the bounce buffer for the rest */
#ifdef VDL2GPU_TESTHOOKS
	h->slab_cap = std::max(1u, std::min<unsigned>(h->slab_cap, (unsigned)env_int("VDL2GPU_SLAB_CAP", (int)h->slab_cap)));	/* (tests: force the bounce path) */
#endif
	for (int r = 0; r < VDL2_NSLAB; ++r) {
		HIPCHK(h, hipHostMalloc(&h->h_slab[r], (size_t)h->slab_cap * sizeof(vdl2gpu_burst_t), hipHostMallocMapped));
		HIPCHK(h, hipHostGetDevicePointer((void **)&h->d_slab[r], h->h_slab[r], 0));
	}
	HIPCHK(h, hipHostMalloc(&h->h_pin_cnt, 32 * VDL2_NRING * sizeof(unsigned), hipHostMallocMapped));
	memset(h->h_pin_cnt, 0, 32 * VDL2_NRING * sizeof(unsigned));
	h->frames_on = (cfg.flags & VDL2GPU_F_FRAMES) != 0;
	HIPCHK(h, hipMalloc(&h->d_k4tab, K4_TABW * sizeof(unsigned)));
	hipLaunchKernelGGL(k4_tables, dim3(1), dim3(64), 0, h->stream, h->d_k4tab);
	HIPCHK(h, hipGetLastError());
	if (h->frames_on) {
		h->frame_cap = h->rec_cap * K4_SLOT + (4u << 20);	/* bytes: a slot per record, and the arena (see K4Params) */
		for (int r = 0; r < VDL2_NRING; ++r)
			HIPCHK(h, hipMalloc((void **)&h->d_frames[r], (size_t)h->frame_cap));
		HIPCHK(h, hipMalloc(&h->d_fcnt, 4 * VDL2_NRING * sizeof(unsigned)));
		HIPCHK(h, hipMemsetAsync(h->d_fcnt, 0, 4 * VDL2_NRING * sizeof(unsigned), h->stream));
	}
	HIPCHK(h, hipHostGetDevicePointer((void **)&h->d_pin_cnt, h->h_pin_cnt, 0));
	HIPCHK(h, hipMalloc(&h->d_dbg, 64 * sizeof(unsigned long long)));
	HIPCHK(h, hipMemsetAsync(h->d_dbg, 0, 64 * sizeof(unsigned long long), h->stream));
	if (cfg.flags & VDL2GPU_F_DEBUG_HEADS) {
		h->headtap_cap = 1u << 18;
		HIPCHK(h, hipMalloc(&h->d_headtap, (size_t)h->headtap_cap * sizeof(HeadTap)));
		HIPCHK(h, hipMalloc(&h->d_headtap_n, sizeof(unsigned)));
		HIPCHK(h, hipMemsetAsync(h->d_headtap_n, 0, sizeof(unsigned), h->stream));
	}

	std::vector<float2> lo((size_t)S * VDL2_CS * L, make_float2(0.f, 0.f));
	std::vector<ChanCfg> cc((size_t)S * VDL2_CS, ChanCfg{ 0, 0, 0, 0 });
	std::vector<float> tmp(2 * (size_t)L);
	for (int s = 0; s < S; ++s)
		for (int c = 0; c < h->C; ++c) {
			const vdl2gpu_chan_t &ch = h->chans[(size_t)s * h->C + c];
			vdl2gpu_lo_table(cfg.sdrinrate, ch.Fo, tmp.data(), L);
			for (int n = 0; n < L; ++n)
				lo[((size_t)s * VDL2_CS + c) * L + n] = make_float2(tmp[2 * n], tmp[2 * n + 1]);
			cc[(size_t)s * VDL2_CS + c] = ChanCfg{ ch.chn, ch.Fr, ch.Fo, 0 };
		}
	HIPCHK(h, hipMemcpyAsync(h->d_lo, lo.data(), lo.size() * sizeof(float2), hipMemcpyHostToDevice, h->stream));
	std::vector<float2> loe((size_t)S * VDL2_CS * (L + 48), make_float2(0.f, 0.f));
	for (size_t sc = 0; sc < (size_t)S * VDL2_CS; ++sc)
		for (int n = 0; n < L + 48; ++n)
			loe[sc * (L + 48) + n] = lo[sc * L + ((n + L - 8) % L)];
	HIPCHK(h, hipMemcpyAsync(h->d_lo_ext, loe.data(), loe.size() * sizeof(float2), hipMemcpyHostToDevice, h->stream));
	HIPCHK(h, hipMemcpyAsync(h->d_cfg, cc.data(), cc.size() * sizeof(ChanCfg), hipMemcpyHostToDevice, h->stream));

	/* scrambler sequence from seed 0x4D4B (d8psk.c:54-65, 299): identical for every burst */
	std::vector<uint8_t> pn(VDL2_PN_BITS);
	unsigned sc = 0x4D4B;
	for (size_t i = 0; i < pn.size(); ++i) {
		const unsigned b = (sc ^ (sc >> 14)) & 1u;
		sc = (sc << 1) | b;
		pn[i] = (uint8_t)b;
	}
	HIPCHK(h, hipMemcpyAsync(h->d_pn, pn.data(), pn.size(), hipMemcpyHostToDevice, h->stream));
	std::vector<uint8_t> pn8(2048, 0);	/* 2040 payload bytes at most (8 rows of 255) */
	for (size_t b = 0; b < pn8.size(); ++b)
		for (int i = 0; i < 8; ++i)
			if (25 + 8 * b + i < pn.size())
				pn8[b] |= (uint8_t)(pn[25 + 8 * b + i] << i);
	HIPCHK(h, hipMalloc(&h->d_pn8, pn8.size()));
	HIPCHK(h, hipMemcpyAsync(h->d_pn8, pn8.data(), pn8.size(), hipMemcpyHostToDevice, h->stream));

	/* canonical start state: everything zero except initD8psk's perr=100 (d8psk.c:28-37);
	 * 16 zero frames stand for the empty Inbuff ring */
	std::vector<StreamState> ss(S);
	memset(ss.data(), 0, ss.size() * sizeof(StreamState));
	for (auto &x : ss) {
		x.dec_base = -VDL2_CARRY_FRAMES;	/* new output always starts at frame VDL2_CARRY_FRAMES; zeros before it */
		x.dec_fill = VDL2_CARRY_FRAMES;
	}
	std::vector<ChanState> cs((size_t)S * VDL2_CS);
	memset(cs.data(), 0, cs.size() * sizeof(ChanState));
	for (auto &x : cs) {
		x.pos = 1;	/* clk: 0 -> 4 (sample 0, idle) -> 8 (sample 1, evaluate) */
		x.r = 0;
		x.fresh = 0;	/* the all-zero start ring counts as history */
		x.perr = 100.0f;
	}
	HIPCHK(h, hipMemcpyAsync(h->d_ss, ss.data(), ss.size() * sizeof(StreamState), hipMemcpyHostToDevice, h->stream));
	HIPCHK(h, hipMemcpyAsync(h->d_cs, cs.data(), cs.size() * sizeof(ChanState), hipMemcpyHostToDevice, h->stream));
	HIPCHK(h, hipStreamSynchronize(h->stream));
	return VDL2GPU_OK;
}

extern "C" int vdl2gpu_create(const vdl2gpu_config_t *cfg, vdl2gpu_t **out)
{
	if (!cfg || !out || cfg->struct_size != sizeof(vdl2gpu_config_t))
		return VDL2GPU_EINVAL;
	if (cfg->nbch < 1 || cfg->nbch > VDL2GPU_MAXCH || cfg->nstreams < 1 || !cfg->chan || !cfg->max_push)
		return VDL2GPU_EINVAL;
	if (fmt_bytes(cfg->fmt) == 0 || cfg->sdrinrate < 100000 || cfg->sdrinrate % 25000)
		return VDL2GPU_EINVAL;
	if ((cfg->flags & VDL2GPU_F_RTL_QUIRK) && cfg->fmt != VDL2GPU_FMT_CU8)
		return VDL2GPU_EINVAL;	/* the quirk is in_callback()'s, and that only ever sees cu8 */
#ifndef VDL2GPU_TESTHOOKS
	if (cfg->flags & VDL2GPU_F_TEST_NOREGION)
		return VDL2GPU_EINVAL;	/* test handicaps are compiled into libvdl2gpu_test.so only */
#endif
	const unsigned sdrclk = cfg->sdrclk ? cfg->sdrclk : cfg->sdrinrate / 4000;
	if (sdrclk <= 21 || sdrclk > 1000000)
		return VDL2GPU_EINVAL;
	vdl2gpu_t *h = new(std::nothrow) vdl2gpu;
	if (!h)
		return VDL2GPU_ENOMEM;
	h->cfg = *cfg;
	h->S = cfg->nstreams;
	h->C = cfg->nbch;
	h->L = (int)(cfg->sdrinrate / 25000);	/* SDRINRATE/STEPRATE, d8psk.c:348 */
	h->sdrclk = (int)sdrclk;
	h->maxwin = (h->sdrclk + 20) / 21;
	h->sample_bytes = fmt_bytes(cfg->fmt);
	h->rec_cap = cfg->max_bursts ? cfg->max_bursts : 65536u;
	h->chans.assign(cfg->chan, cfg->chan + (size_t)cfg->nstreams * cfg->nbch);
	h->cfg.chan = h->chans.data();
	const int rc = create_impl(h);
	if (rc != VDL2GPU_OK) {
		std::string e = h->err;
		if (getenv("VDL2GPU_VERBOSE"))	/* (there is no handle to ask vdl2gpu_last_error() of) */
			fprintf(stderr, "vdl2gpu_create: %s\n", e.c_str());
		vdl2gpu_destroy(h);
		*out = nullptr;
		return rc;
	}
	*out = h;
	return VDL2GPU_OK;
}

/* ------------------------------------------------------------------------ push */
static int get_events(vdl2gpu_t *h, PushTiming &pt)
{
	if (!h->free_ev.empty()) {
		pt = h->free_ev.back();
		h->free_ev.pop_back();
		return VDL2GPU_OK;
	}
	for (auto &e : pt.e)
		HIPCHK(h, hipEventCreate(&e));
	return VDL2GPU_OK;
}

static int harvest_timing(vdl2gpu_t *h)
{
	for (auto &pt : h->pending) {
		float d[NEV - 1] = {0};
		if (h->stage_dump && pt.staged && h->ev_origin) {
			/* VDL2GPU_STAGE_DUMP=1: where every stage of every push began and ended on the GPU's clock, in us since the handle's
			 * first push -- a Gantt chart of the pipeline as it runs WITHOUT a profiler (under rocprofv3 the calling thread is
			 * what the streams wait for).  e0 K1 begins | e1 K1 ends | e10 scan begins | e4 front ends | e2 clusters begin |
			 * e3 = e13 clusters end | e14 resolver ends | e12 verify begins | e15 verify ends | e5 rounds end | e6 commit..export end | e7 tail ends */
			/* (VDL2GPU_STAGE_DUMP only) e23 resolver begins (the previous push's commit has been seen) | e16 tail begins (the verify pass has
			 * been seen on the tail's stream) | e17 merge ends | e18 round's resolver ends | e5 rounds end | e20 commit ends | e21 second
			 * payload pass ends | e22 export ends */
			static const int order[] = {0, 1, 10, 4, 2, 13, 23, 14, 12, 15, 16, 17, 18, 5, 20, 21, 22, 6, 7};
			fprintf(stderr, "vdl2gpu stage dump push %llu:", (unsigned long long)pt.index);
			for (int k : order) {
				float t = -1.0f;
				if (hipEventElapsedTime(&t, h->ev_origin, pt.e[k]) != hipSuccess) {
					(void)hipGetLastError();
					t = -1.0f;
				}
				fprintf(stderr, " e%d=%.1f", k, t * 1e3f);
			}
			fprintf(stderr, "\n");
		}
		for (int i = 0; i + 1 < NEV; ++i) {	/* the demodulator chain starts at e[10], not where the channeliser ended */
			if (!pt.staged)
				break;
			if (i == 3)	/* the resolver alone (it may have run on its own stream, the next push's channeliser beside it) */
				HIPCHK(h, hipEventElapsedTime(&d[i], pt.e[13], pt.e[14]));
			else
				HIPCHK(h, hipEventElapsedTime(&d[i], i == 1 ? pt.e[10] : (i == 4 ? pt.e[12] : pt.e[i]), i == 1 ? pt.e[4] : pt.e[i + 1]));
		}
		if (pt.fast && pt.staged) {	/* kernel intervals only: first period | fast kernel | tail */
			float a = 0, b = 0, c = 0;
			HIPCHK(h, hipEventElapsedTime(&a, pt.e[0], pt.e[11]));
			HIPCHK(h, hipEventElapsedTime(&b, pt.e[8], pt.e[9]));
			HIPCHK(h, hipEventElapsedTime(&c, pt.e[9], pt.e[1]));
			d[0] = a + b + c;
		}
		if (pt.staged) {	/* only some pushes carry events (each costs the stream ~3 us, and the channeliser now sits on the main
					 * stream): their mean stands for all (see vdl2gpu_get_timing) */
			h->st_k1 += d[0];
			h->st_scan += d[1] + d[4];
			h->st_cluster += d[2];
			h->st_resolve += d[3] + d[5];
			h->st_demod += d[1] + d[2] + d[3] + d[4] + d[5];
			h->st_other += d[6];
			h->st_pushes++;
		}
		if (pt.fast && pt.staged) {
			float f = 0;
			HIPCHK(h, hipEventElapsedTime(&f, pt.e[8], pt.e[9]));
			h->tm.channelise_fast_ms += f;
			h->tm.fast_pushes++;	/* counts the fast-kernel launches that were timed */
		}
		h->tm.pushes++;
		h->tm.samples += pt.samples;
		h->free_ev.push_back(pt);
	}
	h->pending.clear();
	return VDL2GPU_OK;
}

static int harvest_ring(vdl2gpu_t *h, int ring, bool blocking);
static int enqueue_back(vdl2gpu_t *h);
static void spill_slab(vdl2gpu_t *h, int slab);

/* A scan kernel.  Its workgroups work what passes their first screen off themselves, behind their last tile (k2a_tail); what
 * their private areas of the item list did not hold is left in the list's common area for the one-workgroup-per-channel kernel
 * that follows on the stream -- `drain` says where that is (put into that kernel's parameters with scan_drain()). */
enum { SCAN_PROBE, SCAN_REGION, SCAN_VERIFY };
struct ScanDrain { int slot = -1, mode = 0, skip = 0, pch = 0, nwg = 0; };
static void scan_drain(K2Params &q, const ScanDrain &d)
{
	q.drain_slot = d.slot;
	q.drain_mode = d.mode;
	q.drain_skip = d.skip;
	q.drain_pch = d.pch;
	q.drain_nwg = d.nwg;
}
static ScanDrain launch_scan(int which, const K2Params &k2, dim3 grid, hipStream_t st, int slot, int mode, int skip, unsigned tiles_per_wg)
{
	K2Params q = k2;
	/* a scan workgroup's private part of the item list: one and a half times what its tiles yield at the first screen's 2.7 %
	 * (28 per tile and class; the region scan's tiles are sync words: far more pass) plus a sync word's worth; what it does
	 * not hold goes to the common area */
	const unsigned item_priv = k2.item_priv;
	grid.x = std::min<unsigned>(grid.x, VDL2_MAXWG);
	if (which != SCAN_VERIFY)	/* (the verify pass's grid IS its map of the part: 256 items x its workgroups fit by construction; the others walk their work with any grid) */
		grid.x = std::max(1u, std::min<unsigned>(grid.x, item_priv / 256u));
	unsigned want = (tiles_per_wg * (which == SCAN_REGION ? 400u : 42u * (K2A_TS / 1024u)) + 128u + 255u) / 256u * 256u;	/* (42 of a 1024-instant tile pass: 2.7 % x 1.5) */
	q.surv_common_cap = 0;	/* (0: whatever the list has left behind the private areas) */
#ifdef VDL2GPU_TESTHOOKS
	const bool test_items = !g_test_item_verify || which == SCAN_VERIFY;
	if (test_items && g_test_item_grid > 0) {	/* VDL2GPU_TEST_ITEM_GRID: few scan workgroups with the smallest private areas -- most items take the common area's path */
		grid.x = std::min<unsigned>(grid.x, (unsigned)g_test_item_grid);
		want = 256u;
	}
	if (test_items && g_test_item_common > 0)	/* VDL2GPU_TEST_ITEM_COMMON: a common area of so many items -- the list overflows */
		q.surv_common_cap = g_test_item_common;
#endif
	q.surv_nwg = (int)grid.x;
	q.surv_pch = (int)std::max(256u, std::min(want, item_priv / grid.x / 256u * 256u));
	q.surv_slot = slot;
	q.surv_mode = mode;
	q.surv_skip = skip;
	q.drain_slot = -1;
	switch (which) {
	case SCAN_PROBE: hipLaunchKernelGGL(k2a_probe, grid, dim3(K2A_THREADS), 0, st, q); break;
	case SCAN_REGION: hipLaunchKernelGGL(k2a_region, grid, dim3(K2A_THREADS), 0, st, q); break;
	default: hipLaunchKernelGGL(k2a_verify, grid, dim3(K2A_THREADS), 0, st, q); break;
	}
	ScanDrain d;
	d.slot = slot;
	d.mode = mode;
	d.skip = skip;
	d.pch = q.surv_pch;
	d.nwg = q.surv_nwg;
	return d;
}

template <int FMT> static void launch_k1(const K1Params &p, dim3 grid, size_t smem, hipStream_t st)
{
	hipLaunchKernelGGL(k1_channelise<FMT>, grid, dim3(K1_THREADS), smem, st, p);
}

static int push_impl(vdl2gpu_t *h, const void *iq, size_t nsamples, size_t stream_stride_bytes, int memkind, bool wait_copy);

static int push_checked(vdl2gpu_t *h, const void *iq, size_t nsamples, size_t stream_stride_bytes, int memkind, bool wait_copy)
{
	if (h && h->failed) {
		h->err = "an earlier HIP error left the handle unusable: " + h->err;
		return VDL2GPU_EHIP;
	}
	/* A push carries at most ~36 s of air time through the tables, and less on busy channels: the tables hold 6144 trigger
	 * candidates per channel, and a channel that overflows them is handled by the serial machine for the whole part
	 * (exact, ~15 ms for 34 s of 8 channels).  Longer pushes are cut into equal parts whose length follows the candidate
	 * density of the pushes collected lately (harvest_ring) -- multiples of a k1_fast superperiod (8000 samples at
	 * 2 MS/s: the parts then need no general channeliser launch at their edges) or of the reference's 32768-sample block
	 * (other rates; VDL2GPU_F_RTL_QUIRK needs whole blocks anyway); any cut gives the same bursts. */
	int rc = VDL2GPU_OK;
	const size_t lim = h ? h->split_samples : 0;
	auto passes = [&](const void *q, size_t n) -> int { return push_impl(h, q, n, stream_stride_bytes, memkind, wait_copy); };
	if (!h || !iq || lim == 0 || nsamples <= lim)
		rc = passes(iq, nsamples);
	else {
		if (nsamples > h->cfg.max_push)
			return VDL2GPU_EINVAL;
		const size_t parts = (nsamples + lim - 1) / lim;
		const size_t unit = h->split_unit;
		const size_t part = ((nsamples + parts - 1) / parts + unit - 1) / unit * unit;
		for (size_t off = 0; off < nsamples && rc == VDL2GPU_OK; off += part)
			rc = passes((const char *)iq + off * h->sample_bytes, std::min(part, nsamples - off));
	}
	if (rc == VDL2GPU_EHIP)
		h->failed = true;	/* part of the push may be enqueued: nothing after it can be trusted */
	return rc;
}

extern "C" int vdl2gpu_push(vdl2gpu_t *h, const void *iq, size_t nsamples, size_t stream_stride_bytes, int memkind)
{
	if (!h)
		return VDL2GPU_EINVAL;
	HLOCK(h);
	/* the caller may reuse a host buffer as soon as we return (the reference's producer refills Cbuff
	 * right after the consumers pass Bar1, d8psk.c:383): wait for the copy, not for the kernels */
	return push_checked(h, iq, nsamples, stream_stride_bytes, memkind, true);
}

/* ---------------------------------------------------------------- ingest ring */
extern "C" int vdl2gpu_ring_init(vdl2gpu_t *h, size_t slot_samples, int nslots)
{
	if (!h || slot_samples == 0 || slot_samples > h->cfg.max_push || nslots < 2 || nslots > 64)
		return VDL2GPU_EINVAL;
	HLOCK(h);
	if (h->ring_host) {
		h->err = "vdl2gpu_ring_init: the ring exists already";
		return VDL2GPU_EINVAL;
	}
	HIPCHK(h, hipSetDevice(h->cfg.device));
	h->ring_slot_samples = slot_samples;
	h->ring_slot_bytes = slot_samples * h->sample_bytes * (size_t)h->S;
	HIPCHK(h, hipHostMalloc(&h->ring_host, h->ring_slot_bytes * (size_t)nslots, hipHostMallocDefault));
	h->ring_nslots = nslots;
	h->ring_copied.resize((size_t)nslots);
	h->ring_inflight.assign((size_t)nslots, 0);
	for (int i = 0; i < nslots; ++i)
		HIPCHK(h, hipEventCreateWithFlags(&h->ring_copied[(size_t)i], hipEventDisableTiming));
	return VDL2GPU_OK;
}

extern "C" void *vdl2gpu_ring_acquire(vdl2gpu_t *h, size_t *stream_stride_bytes)
{
	if (!h)
		return nullptr;
	HLOCK(h);
	if (!h->ring_host || h->ring_acquired)
		return nullptr;
	const size_t slot = (size_t)(h->ring_next % (unsigned long long)h->ring_nslots);
	if (h->ring_inflight[slot]) {	/* its copy to the GPU must have left the slot */
		if (hipSetDevice(h->cfg.device) != hipSuccess || hipEventSynchronize(h->ring_copied[slot]) != hipSuccess) {
			h->err = "vdl2gpu_ring_acquire: waiting for the slot failed";
			return nullptr;
		}
		h->ring_inflight[slot] = 0;
	}
	if (stream_stride_bytes)
		*stream_stride_bytes = h->ring_slot_samples * h->sample_bytes;
	h->ring_acquired = true;
	return (char *)h->ring_host + slot * h->ring_slot_bytes;
}

extern "C" int vdl2gpu_ring_commit(vdl2gpu_t *h, size_t nsamples)
{
	if (!h)
		return VDL2GPU_EINVAL;
	HLOCK(h);
	if (!h->ring_host || !h->ring_acquired || nsamples > h->ring_slot_samples)
		return VDL2GPU_EINVAL;
	const size_t slot = (size_t)(h->ring_next % (unsigned long long)h->ring_nslots);
	h->ring_acquired = false;
	h->ring_next++;
	if (nsamples == 0)
		return VDL2GPU_OK;	/* e.g. a short USB read: the block is dropped (rtl.c:278-281) */
	const int rc = push_checked(h, (char *)h->ring_host + slot * h->ring_slot_bytes, nsamples,
				    h->ring_slot_samples * h->sample_bytes, VDL2GPU_MEM_HOST, false);
	if (h->in_stream) {	/* whatever was enqueued from the slot, its end is marked: acquire() waits for it */
		if (hipEventRecord(h->ring_copied[slot], h->in_stream) == hipSuccess)
			h->ring_inflight[slot] = 1;
		else if (rc == VDL2GPU_OK) {
			h->err = "hipEventRecord(ring_copied)";
			h->failed = true;
			return VDL2GPU_EHIP;
		}
	}
	return rc;
}


/* The BACK stage of a push (see vdl2gpu::Back): resolver, payload decode beside the verify pass, repair rounds, commit,
 * block path, counters -- on the main stream, behind the previous push's back stage and behind this push's front. */
static int enqueue_back(vdl2gpu_t *h)
{
	if (!h->back.valid)
		return VDL2GPU_OK;
	h->back.valid = false;
	const K2Params &k2 = h->back.k2;
	const int64_t J = h->back.J;
	const int par = h->back.par, ring = h->back.ring;
	const bool staged = h->back.staged, serial = h->back.serial;
	const unsigned tiles = h->back.tiles;
	const int GS = h->S;
	PushTiming &pt = h->pending[h->back.pt_index];
	const dim3 gch((unsigned)h->C, (unsigned)GS);
	hipStream_t rs = h->stream;
	if (h->back.two_streams)
		HIPCHK(h, hipStreamWaitEvent(rs, h->f_done[par], 0));
	/* the cluster kernel needs nothing of the previous push's result either, but it is wide, and the stages are better
	 * balanced with it here: FRONT = channeliser + scan, BACK = clusters + resolver + verify */
	if (staged)
		HIPCHK(h, hipEventRecord(pt.e[2], rs));
	if (!serial)
		hipLaunchKernelGGL(k2b_clusters, dim3((unsigned)(h->n_cu * 4 * K2B_GRIDW), (unsigned)((GS * VDL2_CS + 63) / 64)), dim3(K2B_NT), 0, rs, k2);
	HIPCHK(h, hipGetLastError());
	if (staged)
		HIPCHK(h, hipEventRecord(pt.e[3], rs));
	if (staged)
		HIPCHK(h, hipEventRecord(pt.e[13], rs));
	const dim3 vgrid0((tiles / 2 + 1 + K2A_VRUN - 1) / K2A_VRUN, (unsigned)h->C, (unsigned)GS);
	ScanDrain vdrain;	/* the verify pass whose common area the next one-workgroup-per-channel kernel of the tail has to drain */
	if (h->k2f_rec)		/* the channel states the resolver starts from are committed on the previous push's tail */
		HIPCHK(h, hipStreamWaitEvent(rs, h->k2f_done, 0));
	if (staged && h->stage_dump)
		HIPCHK(h, hipEventRecord(pt.e[23], rs));
	hipLaunchKernelGGL(k2c_resolve, gch, dim3(K2_NT), 0, rs, k2);
	HIPCHK(h, hipGetLastError());
	if (staged)
		HIPCHK(h, hipEventRecord(pt.e[14], rs));
	/* The resolver's selection is final unless the verify pass fails -- then a repair round re-resolves the
	 * channel, or K2f redoes it serially, and the host drops what K2d made of it (see harvest_ring; K2d decodes
	 * a repaired selection in a second pass behind the rounds): decode the payloads beside the verify pass
	 * instead of behind it. */
	const bool spec = !h->full_scan && !serial && h->S * VDL2_CS <= 512;
	h->ring_spec[ring] = spec;
	if (spec)
		HIPCHK(h, hipEventRecord(h->k2c_done, rs));
	/* The payload decode beside the verify pass: on the copy stream (a hardware queue of its own), so that the push's TAIL on the
	 * payload stream -- the repair round and the commit, which the NEXT push's resolver waits for: the pipeline's loop-carried
	 * dependency -- starts when the verify pass ends, not when this latency-bound kernel has found CUs between the verify pass's
	 * workgroups and finished.  The repair rounds write a selection of their own (K2Params.sel_list2), so nothing the decode
	 * reads changes under it; the tail waits for it only where it needs its records: in front of the second payload pass and the
	 * export.  (VDL2GPU_PAY_TAIL=1: on the payload stream in front of the tail, round 3's arrangement.) */
	hipStream_t ps = h->copy_stream;	/* (four hardware queues: the copy stream has one job) */
	if (spec) {
		HIPCHK(h, hipStreamWaitEvent(ps, h->k2c_done, 0));
		hipLaunchKernelGGL(k2d_payload, dim3((unsigned)h->k2d_grid, (unsigned)(VDL2_CS * GS)), dim3(K2D_NT), 0, ps, k2);
		HIPCHK(h, hipEventRecord(h->pay_done, ps));
	}
	if (staged)
		HIPCHK(h, hipEventRecord(pt.e[12], h->stream));
	if (!serial)
		vdrain = launch_scan(SCAN_VERIFY, k2, vgrid0, h->stream, VDL2_SURV_VERIFY, 1, 0, K2A_VRUN);
	HIPCHK(h, hipGetLastError());
	if (staged && h->stage_dump)
		HIPCHK(h, hipEventRecord(pt.e[15], h->stream));
	/* ---- the TAIL: everything behind the verify pass -- repair rounds, commit, the payloads a round re-resolved, block path,
	 * export, counters: a chain of one-workgroup-per-channel kernels and a PCIe copy, 0.1 ms while nothing fails and 0.3 ms when
	 * a channel is repaired (most pushes of ordinary traffic).  On the main stream it stood between this push's verify pass and
	 * the NEXT push's cluster kernel, which needs none of it; it runs on the payload stream instead (behind the payload decode
	 * it would have had to wait for anyway), the main stream goes on with the next push, and only that push's resolver waits --
	 * for the commit (k2f_done), which the cluster kernel in front of it covers.  k2_done, which frees the plane and table sets
	 * and tells the host the ring is complete, is recorded at the tail's end. */
	hipStream_t ts = (spec && h->back.two_streams && !h->knob.no_tail) ? h->pay_stream : h->stream;
	if (ts != h->stream) {
		HIPCHK(h, hipEventRecord(h->verify_done, h->stream));
		HIPCHK(h, hipStreamWaitEvent(ts, h->verify_done, 0));
	}
	if (h->tail_prev && h->tail_prev != ts && h->k2_rec[(par + VDL2_NSET - 1) % VDL2_NSET])	/* tails follow each other (running totals, StreamState) */
		HIPCHK(h, hipStreamWaitEvent(ts, h->k2_done[(par + VDL2_NSET - 1) % VDL2_NSET], 0));
	h->tail_prev = ts;
	if (staged && h->stage_dump)
		HIPCHK(h, hipEventRecord(pt.e[16], ts));
	if (!h->full_scan && !serial) {
		/* Repair rounds.  The verify pass has appended what it found to the failing channel's table (candidates without
		 * clusters): a round re-sorts the table, re-resolves the channel -- the resolver replays the new candidates with the
		 * serial machine and returns to the tables behind each -- and verifies the stretches the new chain idles through
		 * (other classes than before from the first new event on).  Every other channel's workgroups exit at once: three
		 * launches, ~12 us, when nothing failed -- which is why one round is ALWAYS scheduled: an unlisted event (a noise
		 * trigger that exists in one class only) turns up about once per 100 channel-seconds of ordinary traffic, and without
		 * a round it costs a serial redo of the channel's whole push (K2f: 8 ms at 33 s of air time).  If more than one round
		 * is scheduled (after a serial redo: adapted below), the LAST one does not repair, it starts over: the channels that
		 * still fail are scanned completely -- every class at every instant, like VDL2GPU_F_FULLSCAN but for them alone
		 * (~0.1 ms for a channel of a 67 MS push) -- their tables rebuilt from nothing, which leaves nothing to verify and
		 * nothing to cascade.  What still fails after the last round is redone serially by K2f. */
		K2Params k2r = k2;
		/* (a repair round writes its own selection, sel_list2: the payload decode of the first one goes on beside it -- unless the
		 * last round is a complete one: that re-makes the failing channels' clusters, whose descriptors the decode may be reading) */
		if (spec && h->repair_rounds >= 2 && ts != ps)
			HIPCHK(h, hipStreamWaitEvent(ts, h->pay_done, 0));
		const dim3 vgrid((tiles / 2 + 1 + K2A_VRUN - 1) / K2A_VRUN, (unsigned)h->C, (unsigned)GS);
		for (int rr = 1; rr <= h->repair_rounds; ++rr) {
			k2r.round = rr;
			k2r.full_round = (rr == h->repair_rounds && h->repair_rounds >= 2) ? 1 : 0;
			k2r.mini_round = k2r.full_round ? 0 : 1;
			if (k2r.full_round) {
				scan_drain(k2r, vdrain);
				hipLaunchKernelGGL(k2r_regions, gch, dim3(K2R_NT), 0, ts, k2r);	/* (resets the channel's tables) */
				scan_drain(k2r, ScanDrain());
				const unsigned want = tiles;
				unsigned per = (unsigned)h->n_cu;	/* few channels fail: each may use the whole GPU (the others' workgroups leave at once) */
				per = per > want ? want : per;
				const ScanDrain pdrain = launch_scan(SCAN_PROBE, k2r, dim3(per, (unsigned)h->C, (unsigned)GS), ts, VDL2_SURV_FULL, 0, 0, 4 * ((want + per - 1) / per));
				scan_drain(k2r, pdrain);
				hipLaunchKernelGGL(k2s_sort, gch, dim3(K2S_NT), 0, ts, k2r);
				scan_drain(k2r, ScanDrain());
				hipLaunchKernelGGL(k2b_clusters, dim3((unsigned)(h->n_cu * 4 * K2B_GRIDW), (unsigned)((GS * VDL2_CS + 63) / 64)), dim3(K2B_NT), 0, ts, k2r);
				hipLaunchKernelGGL(k2c_resolve, gch, dim3(K2_NT), 0, ts, k2r);
				vdrain = ScanDrain();	/* (nothing is verified behind a complete round) */
			} else if (rr == 1) {
				/* the first round repairs locally: from each event the verify pass listed to where the new chain rejoins the old one
				 * (k2p_patch: one narrow kernel, tables read where they lie), and the verify pass looks at what changed */
				scan_drain(k2r, vdrain);
				hipLaunchKernelGGL(k2p_patch, gch, dim3(K2P_NT), 0, ts, k2r);
				if (staged && h->stage_dump)
					HIPCHK(h, hipEventRecord(pt.e[18], ts));
				scan_drain(k2r, ScanDrain());
				{
					/* a handful of workgroups per channel, each walking its share of the part's runs of tiles (k2a_verify): the stretches a
					 * local repair changed are a few tiles, and every workgroup of the launch, the 2 800 that find nothing included, has to
					 * wait for a slot beside the other pushes' wide kernels before it can leave */
					const unsigned nv = std::min<unsigned>(vgrid.x, (unsigned)h->knob.verify2_wg);
					const dim3 vnarrow(nv, vgrid.y, vgrid.z);
					vdrain = launch_scan(SCAN_VERIFY, k2r, vnarrow, ts, VDL2_SURV_VERIFY + rr, 1, 0, K2A_VRUN * ((vgrid.x + nv - 1) / nv));
				}
			} else {
				/* a further round resolves the channels that still fail again from their input state, with everything listed so far */
				scan_drain(k2r, vdrain);
				hipLaunchKernelGGL(k2s_merge, gch, dim3(K2M_NT), 0, ts, k2r);
				scan_drain(k2r, ScanDrain());
				hipLaunchKernelGGL(k2c_resolve, gch, dim3(K2_NT), 0, ts, k2r);
				vdrain = launch_scan(SCAN_VERIFY, k2r, vgrid, ts, VDL2_SURV_VERIFY + rr, 1, 0, K2A_VRUN);
			}
		}
		HIPCHK(h, hipGetLastError());
	}
	if (staged)
		HIPCHK(h, hipEventRecord(pt.e[5], ts));
	{
		K2Params k2f = k2;
		scan_drain(k2f, vdrain);
		hipLaunchKernelGGL(k2f_commit, gch, dim3(K2_NT), 0, ts, k2f);
	}

	HIPCHK(h, hipEventRecord(h->k2f_done, ts));
	h->k2f_rec = true;
	if (staged && h->stage_dump)
		HIPCHK(h, hipEventRecord(pt.e[20], ts));
	if (h->ring_spec[ring]) {
		if (ts != ps)
			HIPCHK(h, hipStreamWaitEvent(ts, h->pay_done, 0));	/* the export needs the first pass's records, K3 publishes the record count */
		if (!h->full_scan && !serial) {
			K2Params k2p = k2;	/* what the repair rounds (or K2f's serial redo) made void of the first selection is tagged now, what they selected is decoded */
			k2p.sel_mode = 1;
			hipLaunchKernelGGL(k2d_payload, dim3((unsigned)h->k2d_grid, (unsigned)(VDL2_CS * GS)), dim3(K2D_NT), 0, ts, k2p);
		}
	} else {
		K2Params k2p = k2;	/* one pass behind the commit: the repaired selection where there is one */
		k2p.sel_mode = 2;
		hipLaunchKernelGGL(k2d_payload, dim3((unsigned)h->k2d_grid, (unsigned)(VDL2_CS * GS)), dim3(K2D_NT), 0, ts, k2p);
	}
	HIPCHK(h, hipGetLastError());
	if (staged && h->stage_dump)
		HIPCHK(h, hipEventRecord(pt.e[21], ts));
	if (h->frames_on) {
		/* block path on the records where they lie (vdlm2.c:84-161).  In the chain, not beside it:
		 * a latency-bound kernel like this one and the next push's scan slow each other down by
		 * more than the overlap saves. */
		K4Params k4{};
		k4.recs = h->d_recs[ring];
		k4.nrecs_dev = h->d_outc + 2 * ring;
		k4.rec_cap = h->rec_cap;
		k4.frames = h->d_frames[ring];
		k4.nframes = h->d_fcnt + 4 * ring;
		k4.frame_cap = h->frame_cap;
		k4.compact = 1;
		k4.tabs = h->d_k4tab;
		k4.fmask = h->ring_spec[ring] ? h->d_fmask[par] : nullptr;
		k4.dbg = h->knob.debug_counters ? h->d_dbg : nullptr;
		hipLaunchKernelGGL(k4_frames, dim3((unsigned)h->n_cu * 16), dim3(K4_NT), 0, ts, k4);
		HIPCHK(h, hipGetLastError());
	}
	if (staged)
		HIPCHK(h, hipEventRecord(pt.e[6], ts));
	{
		K3Params k3{};
		k3.src = nullptr;	/* (the counters kernel copies nothing) */
		k3.dst = nullptr;
		k3.cap = h->cap;
		k3.nbch = h->C;
		k3.J = J;
		k3.ss = h->d_ss;
		k3.cs = h->d_cs;
		k3.outc = h->d_outc;
		k3.fmask = h->d_fmask[par];
		k3.fcnt = h->frames_on ? h->d_fcnt + 4 * ring : nullptr;
		k3.host_cnt = h->d_pin_cnt + 32 * ring;
		k3.ring = ring;
		k3.ctl = h->d_ctl[par];
		k3.nstreams = h->S;
		{
			/* the push's records go to the host by the GPU's own hand: page-locked memory, coalesced 8-byte stores */
			KExportParams ke{};
			ke.recs = h->d_recs[ring];
			ke.count = h->d_outc + 2 * ring;
			ke.dst = h->d_slab[h->back.slab];
			ke.cap = std::min(h->slab_cap, h->rec_cap);
			hipLaunchKernelGGL(k_export_records, dim3((unsigned)h->n_cu), dim3(256), 0, ts, ke);
			HIPCHK(h, hipGetLastError());
			if (staged && h->stage_dump)
				HIPCHK(h, hipEventRecord(pt.e[22], ts));
		}
		hipLaunchKernelGGL(k3_rebase, dim3((unsigned)GS), dim3(64), 0, ts, k3);
		HIPCHK(h, hipGetLastError());
	}
	if (staged)
		HIPCHK(h, hipEventRecord(pt.e[7], ts));
	HIPCHK(h, hipEventRecord(h->k2_done[par], ts));
	h->k2_rec[par] = true;
	h->ring_ev[ring] ^= 1;
	HIPCHK(h, hipEventRecord(h->ring_done(ring), ts));
	return VDL2GPU_OK;
}

static int push_impl(vdl2gpu_t *h, const void *iq, size_t nsamples, size_t stream_stride_bytes, int memkind, bool wait_copy)
{
	if (!h || (!iq && nsamples))
		return VDL2GPU_EINVAL;
	if (nsamples == 0)
		return VDL2GPU_OK;
	if (nsamples > h->cfg.max_push)
		return VDL2GPU_EINVAL;
	if (h->S > 1 && stream_stride_bytes < nsamples * h->sample_bytes)
		return VDL2GPU_EINVAL;
	if (h->quirk && nsamples % 32768) {
		h->err = "VDL2GPU_F_RTL_QUIRK: every push must be whole 32768-sample blocks";
		return VDL2GPU_EINVAL;
	}
	HIPCHK(h, hipSetDevice(h->cfg.device));
	if (h->pending.size() >= 256) {	/* bound the event backlog */
		HIPCHK(h, hipStreamSynchronize(h->fstream));
		HIPCHK(h, hipStreamSynchronize(h->stream));
		HIPCHK(h, hipStreamSynchronize(h->pay_stream));
		int rc = harvest_timing(h);
		if (rc)
			return rc;
	}
	/* include/vdl2gpu.h: a device buffer must stay unchanged "until the second push after this one has been issued": that push
	 * is this call, for the buffer of the push before last (with two output rings the wait for that push's ring implied it) */
	const int GS = h->S;
	auto hnow = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	double hp_t = hnow();	/* (always on: six clock reads a push; vdl2gpu_get_host_profile() hands the sums out, VDL2GPU_HOST_PROF prints them at destroy) */
	auto hp = [&](int k) { const double t = hnow(); h->hprof[k] += t - hp_t; hp_t = t; };
	if (h->in_rec[(h->pushes + 1) % VDL2_NRING])
		HIPCHK(h, hipEventSynchronize(h->in_read[(h->pushes + 1) % VDL2_NRING]));
	hp(0);
	const int slab = (int)(h->pushes % VDL2_NSLAB);	/* page-locked slab this push's records are exported to */
	const int ring = (int)(h->pushes % VDL2_NRING);	/* output ring of this push (collected below, once the GPU has been given work to do meanwhile) */
	const void *src = iq;
	size_t stride = stream_stride_bytes;
	bool staged_in = false;
	const int stg = (int)(h->pushes & 1);	/* staging buffer of this push; its channeliser's events are k1_done[stg] */
	if (memkind == VDL2GPU_MEM_HOST) {
		const size_t per = nsamples * h->sample_bytes;
		const size_t need = per * (size_t)h->S;
		if (!h->in_stream) {
			HIPCHK(h, hipStreamCreateWithFlags(&h->in_stream, hipStreamNonBlocking));
			for (int i = 0; i < 2; ++i)
				HIPCHK(h, hipEventCreateWithFlags(&h->raw_copied[i], hipEventDisableTiming));
		}
		if (need > h->raw_bytes[stg]) {
			HIPCHK(h, hipStreamSynchronize(h->fstream));
			HIPCHK(h, hipStreamSynchronize(h->stream));
			HIPCHK(h, hipStreamSynchronize(h->pay_stream));
			HIPCHK(h, hipStreamSynchronize(h->in_stream));
			(void)hipFree(h->d_raw[stg]);
			h->d_raw[stg] = nullptr;
			h->raw_bytes[stg] = 0;
			HIPCHK(h, hipMalloc(&h->d_raw[stg], need));
			h->raw_bytes[stg] = need;
		}
		/* the channeliser of the push before last has read this buffer */
		if (h->k1_rec[stg])
			HIPCHK(h, hipStreamWaitEvent(h->in_stream, h->k1_done[stg], 0));
		for (int s = 0; s < GS; ++s)
			HIPCHK(h, hipMemcpyAsync((char *)h->d_raw[stg] + (size_t)s * per,
						 (const char *)iq + (size_t)s * stream_stride_bytes, per,
						 hipMemcpyHostToDevice, h->in_stream));
		HIPCHK(h, hipEventRecord(h->raw_copied[stg], h->in_stream));
		if (wait_copy)
			HIPCHK(h, hipEventSynchronize(h->raw_copied[stg]));
		staged_in = true;
		src = h->d_raw[stg];
		stride = per;
	} else if (memkind != VDL2GPU_MEM_DEVICE)
		return VDL2GPU_EINVAL;

	K1Params k1{};
	int64_t J = 0;
	vdl2gpu_plan(h->total_in, nsamples, (unsigned)h->sdrclk, (unsigned)h->L, &k1.c0, &k1.no0, &k1.nf0, &J);
	const int par = (int)(h->pushes % VDL2_NSET);	/* table set, plane set and output ring of this push */
	const int pset = par;
	k1.raw = src;
	k1.stream_stride = stride;
	k1.fmt = h->cfg.fmt;
	k1.nbch = h->C;
	k1.sdrclk = h->sdrclk;
	k1.L = h->L;
	k1.maxwin = h->maxwin;
	k1.parity = (int)(h->pushes & 1);	/* the carried partial window is double-buffered in StreamState.acc: read [parity], written [parity ^ 1] */
	k1.quirk = h->quirk;
	k1.N = (long long)nsamples;
	k1.J = J;
	k1.lo = h->d_lo;
	k1.dec = h->d_dec[pset];
	k1.cap = h->cap;
	k1.ss = h->d_ss;
	/* A short push (a live SDR block is 1376 frames per channel) is cheaper on the serial machine alone
	 * than through the scan's ten launches: the parallel path only pays from a few thousand frames on. */
#ifdef VDL2GPU_TESTHOOKS
	const bool noregion = (h->cfg.flags & VDL2GPU_F_TEST_NOREGION) != 0;
#else
	const bool noregion = false;
#endif
	const bool serial = h->force_serial || (J <= VDL2_SERIAL_BELOW && !h->full_scan && !noregion);
	const bool two_streams = !serial;	/* see vdl2gpu::Back */
	hipStream_t fs = two_streams ? h->fstream : h->stream;
	/* stream time of frame 0 of this push's planes: the outputs completed before it, minus the carried frames in front */
	const long long dec_base = (long long)(((unsigned __int128)h->total_in * 21u) / (unsigned)h->sdrclk) - VDL2_CARRY_FRAMES;

	PushTiming pt{};
	int rc = get_events(h, pt);
	if (rc)
		return rc;
	pt.samples = nsamples;	/* (per stream) */
	pt.fast = false;
	pt.staged = h->stage_events && (h->pushes % (uint64_t)h->stage_every) == 0;
	pt.index = h->pushes;
	const bool staged = pt.staged;
	/* The channeliser opens the front stage (fstream; the main stream for a push that takes the serial path). */
	hipStream_t ks = fs;
	if (!two_streams && h->last_two_streams)	/* the previous push's channeliser state and carry were written on the front stream */
		HIPCHK(h, hipStreamWaitEvent(fs, h->f_tail, 0));
	if (!two_streams && h->k2_rec[(par + VDL2_NSET - 1) % VDL2_NSET])	/* ... and its tail may have run on the payload stream */
		HIPCHK(h, hipStreamWaitEvent(fs, h->k2_done[(par + VDL2_NSET - 1) % VDL2_NSET], 0));
	if (two_streams) {
		/* the plane set this push's channeliser writes, the table set and the output ring were last used by the push three
		 * back: by its tail (the payload decode of a repaired channel reads the planes to the very end of it) */
		if (h->k2_rec[par])
			HIPCHK(h, hipStreamWaitEvent(fs, h->k2_done[par], 0));
		if (h->k1_ev_rec && !h->last_two_streams)	/* the previous push's channeliser ran on the main stream */
			HIPCHK(h, hipStreamWaitEvent(fs, h->k1_ev, 0));
	}
	if (staged_in)
		HIPCHK(h, hipStreamWaitEvent(ks, h->raw_copied[stg], 0));
	if (staged && h->stage_dump && !h->ev_origin) {	/* the origin of the dump's times: in front of the first push's first event, on its stream */
		HIPCHK(h, hipEventCreate(&h->ev_origin));
		HIPCHK(h, hipEventRecord(h->ev_origin, ks));
	}
	if (staged)
		HIPCHK(h, hipEventRecord(pt.e[0], ks));
	{
		const long long per_block = K1_OPB * K1_PASSES;
		const size_t smem = ((size_t)(h->L + h->maxwin) * VDL2_CS + (size_t)K1_OPB * h->maxwin) * sizeof(float2);
		auto generic = [&](long long jbeg, long long jend) {
			if (jend < jbeg)
				return;
			K1Params q = k1;
			q.jbeg = jbeg;
			q.jend = jend;
			const unsigned gx = (unsigned)((jend - jbeg + 1 + per_block - 1) / per_block);
			const dim3 grid(gx, (unsigned)GS);
			switch (h->cfg.fmt) {
			case VDL2GPU_FMT_CU8: launch_k1<VDL2GPU_FMT_CU8>(q, grid, smem, ks); break;
			case VDL2GPU_FMT_CS16: launch_k1<VDL2GPU_FMT_CS16>(q, grid, smem, ks); break;
			case VDL2GPU_FMT_CF32: launch_k1<VDL2GPU_FMT_CF32>(q, grid, smem, ks); break;
			default: launch_k1<VDL2GPU_FMT_F32R>(q, grid, smem, ks); break;
			}
		};
		/* whole periods of the schedule (4*SDRCLK inputs = 84 outputs, the LO table a whole number of times:
		 * SDRINRATE = 4000*SDRCLK, air.c:138) on the period-parallel kernel; the first period (carried partial
		 * window) and the tail on the general one */
		const long long periods = J / K1P_PER_OUT;
		const int per_in = 4 * h->sdrclk;
		bool fast = (per_in % h->L == 0 && periods >= 4 && !h->quirk && !h->knob.no_k1_fast &&
			     std::min(K1P_CH, h->maxwin) <= h->L);	/* k1_pp steps its LO index by a piece (<= a chunk, <= a window) and wraps it once */
		K1PParams kp{};
		/* a push that starts on a window boundary of the schedule (nothing carried in) and is a whole number of periods (nothing
		 * carried out) needs no general launch at either end: the period-parallel kernel takes all of it (as k1_fast does below) */
		const bool whole_pp = k1.c0 == 0 && nsamples % (size_t)per_in == 0 && J == periods * K1P_PER_OUT && !h->knob.no_whole_pp;
		if (fast) {
			auto wend_abs = [&](long long j) { return ((j + 1) * (long long)h->sdrclk - k1.c0 + 20) / 21 - 1; };
			kp.per_lo = whole_pp ? 0 : 1;
			kp.sbase0 = wend_abs(K1P_PER_OUT * kp.per_lo - 1) + 1;
			/* 16-byte pieces: a period's first sample sits d samples above a 16-byte boundary, the same d for
			 * every period (a period is a whole number of 16-byte pieces) and every stream */
			const uintptr_t a0 = (uintptr_t)src + (uintptr_t)kp.sbase0 * h->sample_bytes;
			if ((a0 % 16) % h->sample_bytes || (h->S > 1 && stride % 16) || ((size_t)per_in * h->sample_bytes) % 16)
				fast = false;
			kp.d = (int)((a0 % 16) / h->sample_bytes);
			if (whole_pp && kp.d != 0) {	/* (the kernel reads a period from the 16-byte boundary below its first sample: that would lie in front of the buffer) */
				kp.per_lo = 1;
				kp.sbase0 = wend_abs(K1P_PER_OUT * kp.per_lo - 1) + 1;
				const uintptr_t a1 = (uintptr_t)src + (uintptr_t)kp.sbase0 * h->sample_bytes;
				if ((a1 % 16) % h->sample_bytes)
					fast = false;
				kp.d = (int)((a1 % 16) / h->sample_bytes);
			}
		}
		const long long nsp = periods / 4;	/* superperiods of 4 periods = 336 outputs = 21 lines of the planes */
		const bool fast2m = fast && h->sdrclk == 500 && h->L == 80 && nsp >= 3 && !h->knob.k1_pp &&
				    (size_t)h->cap * VDL2_CS * sizeof(float2) < (1ull << 32);	/* k1_fast addresses a stream's planes with 32-bit offsets */
		if (fast2m) {
			/* 2 MS/s: the LO values of a window fit a lane's registers (lane = window x channel).  Whole superperiods in
			 * the middle; the first one (carried partial window) and the tail on the general kernel -- unless the push
			 * starts on a window boundary of the schedule (c0 == 0: nothing carried in) and is a whole number of
			 * superperiods (nothing carried out): then the fast kernel takes all of it and the two general launches
			 * (36 us each for 0.02 % of the samples: launch and latency, not work) are not made at all. */
			const bool whole = k1.c0 == 0 && nsamples % K1F_PER_IN == 0 && J == nsp * K1F_PER_OUT;
			if (!whole)
				generic(0, K1F_PER_OUT - 1);
			if (staged)
				(void)hipEventRecord(pt.e[11], ks);	/* the wait for the resolver that follows is not channeliser time */
			pt.fast = true;
			k1.per_lo = whole ? 0 : 1;
			k1.per_n = whole ? nsp : nsp - 2;
			k1.edge_state = whole ? 1 : 0;
			k1.lo_ext = h->d_lo_ext;
			k1.lo_stride = h->L + 48;
			if (staged)
				(void)hipEventRecord(pt.e[8], ks);
			/* The grid is resident as a whole: n_cu * 2 * K1F_WAVES_OF(fmt) workgroups of two wavefronts fit.  Per stream
			 * 21 roles x 8 XCDs families of `nfam` workgroups each, which take the family's tickets in turn (see k1_fast);
			 * a family needs no more workgroups than it has tickets.  With several streams the families are many and
			 * small: rather two workgroups each and a twentieth of them waiting for a slot than one each and half the
			 * SIMDs' wavefront slots empty. */
			long long ngrp;
			{
				const long long slots = (long long)h->n_cu * 2 * K1F_WAVES_OF(h->cfg.fmt);
				const long long per_fam = (long long)K1F_ROLES * 8 * GS;
				long long nfam = slots / per_fam;
				if (nfam < 4 && (nfam + 1) * per_fam * 100 <= slots * 108)
					++nfam;
				if (h->knob.k1f_nfam > 0)
					nfam = h->knob.k1f_nfam;
				const long long tickets = ((k1.per_n + 7) / 8 + K1F_CHUNK - 1) / K1F_CHUNK;	/* of the family with the most */
				nfam = std::max<long long>(1, std::min(nfam, tickets));
				ngrp = nfam * 8;
			}
			/* the counters are never reset: a launch makes exactly one request per ticket of a family (k1_fast), so the
			 * host knows where each one stands */
			k1.tickets = h->d_k1_tickets;
			for (int x = 0; x < 8; ++x) {
				k1.tbase[x] = h->k1_tbase[x];	/* (every stream stands where the first does: all have seen the same pushes) */
				const long long n_x = (k1.per_n - x + 7) >> 3;
				if (n_x > 0)
					for (int sg = 0; sg < GS; ++sg)
						h->k1_tbase[(size_t)sg * 8 + x] += (unsigned)((n_x + K1F_CHUNK - 1) / K1F_CHUNK);
			}
			const dim3 grid((unsigned)ngrp * K1F_ROLES, (unsigned)GS);
			switch (h->cfg.fmt) {
			case VDL2GPU_FMT_CU8: hipLaunchKernelGGL(k1_fast<VDL2GPU_FMT_CU8>, grid, dim3(K1F_THREADS), 0, ks, k1); break;
			case VDL2GPU_FMT_CS16: hipLaunchKernelGGL(k1_fast<VDL2GPU_FMT_CS16>, grid, dim3(K1F_THREADS), 0, ks, k1); break;
			case VDL2GPU_FMT_CF32: hipLaunchKernelGGL(k1_fast<VDL2GPU_FMT_CF32>, grid, dim3(K1F_THREADS), 0, ks, k1); break;
			default: hipLaunchKernelGGL(k1_fast<VDL2GPU_FMT_F32R>, grid, dim3(K1F_THREADS), 0, ks, k1); break;
			}
			if (staged)
				(void)hipEventRecord(pt.e[9], ks);
			pt.fast_parts = 1;
			if (!whole)
				generic((nsp - 1) * K1F_PER_OUT, J);
		} else if (fast) {
			const bool whole = kp.per_lo == 0;
			if (!whole)
				generic(0, K1P_PER_OUT - 1);
			if (staged)
				(void)hipEventRecord(pt.e[11], ks);
			pt.fast = true;
			kp.edge_state = whole ? 1 : 0;
			kp.parity = k1.parity;
			kp.J = J;
			kp.raw = src;
			kp.stream_stride = stride;
			kp.nbch = h->C;
			kp.per_in = per_in;
			kp.L = h->L;
			kp.ph0 = (int)(((long long)k1.no0 + kp.sbase0) % h->L);
			kp.per_n = (int)(whole ? periods : periods - 2);
			kp.lo_ext = h->d_lo_ext;
			kp.lo_stride = h->L + 48;
			kp.dec = k1.dec;
			kp.cap = h->cap;
			kp.ss = h->d_ss;
			auto wend_abs = [&](long long j) { return ((j + 1) * (long long)h->sdrclk - k1.c0 + 20) / 21 - 1; };
			int nfmin = 1 << 30, nfmax = 0;
			for (int k = 0; k < K1P_PER_OUT; ++k) {
				kp.wend[k] = (int)(wend_abs(K1P_PER_OUT * kp.per_lo + k) - kp.sbase0);
				const int nf = kp.wend[k] - (k ? kp.wend[k - 1] : -1);
				nfmin = std::min(nfmin, nf);
				nfmax = std::max(nfmax, nf);
			}
			auto proven = [](int nf) { return nf == 23 || nf == 24 || nf == 59 || nf == 60 || nf == 71 || nf == 72 || nf == 119 || nf == 120; };
			kp.fast_div = proven(nfmin) && proven(nfmax) && nfmax - nfmin <= 1;
			kp.nf_lo = nfmin;
			kp.dbg = h->knob.k1_dbg;
			kp.rcp_lo = 1.0f / (float)nfmin;
			kp.rcp_hi = 1.0f / (float)(nfmin + 1);
			/* tasks = (blocks of 64 periods) x (runs of wpt windows): enough of them that the last round of
			 * workgroups is a small share of the launch, as long as possible otherwise */
			const long long blocks = (kp.per_n + 63) / 64;
			const int divs[] = {1, 2, 3, 4, 6, 7, 12, 14, 21, 28};
			const long long resident = (long long)h->n_cu * 3;
			int best = 1;
			double best_eff = -1;
			for (int nsub : divs) {
				const long long tasks = blocks * nsub * GS;
				const long long rounds = (tasks + resident - 1) / resident;
				const double eff = (double)tasks / (double)(rounds * resident) - 0.004 * nsub;	/* shorter tasks pay their start-up more often */
				if (eff > best_eff) {
					best_eff = eff;
					best = nsub;
				}
			}
			if (h->knob.k1_nsub > 0)
				best = h->knob.k1_nsub;
			kp.nsub = best;
			kp.wpt = K1P_PER_OUT / best;
			if (staged)
				(void)hipEventRecord(pt.e[8], ks);
			const dim3 grid((unsigned)(blocks * kp.nsub), (unsigned)GS);
			switch (h->cfg.fmt) {
			case VDL2GPU_FMT_CU8: hipLaunchKernelGGL(k1_pp<VDL2GPU_FMT_CU8>, grid, dim3(K1P_THREADS), 0, ks, kp); break;
			case VDL2GPU_FMT_CS16: hipLaunchKernelGGL(k1_pp<VDL2GPU_FMT_CS16>, grid, dim3(K1P_THREADS), 0, ks, kp); break;
			case VDL2GPU_FMT_CF32: hipLaunchKernelGGL(k1_pp<VDL2GPU_FMT_CF32>, grid, dim3(K1P_THREADS), 0, ks, kp); break;
			default: hipLaunchKernelGGL(k1_pp<VDL2GPU_FMT_F32R>, grid, dim3(K1P_THREADS), 0, ks, kp); break;
			}
			if (staged)
				(void)hipEventRecord(pt.e[9], ks);
			pt.fast_parts = 1;
			if (!whole)
				generic((periods - 1) * K1P_PER_OUT, J);
		} else
			generic(0, J);
		HIPCHK(h, hipGetLastError());
	}
	if (staged)
		HIPCHK(h, hipEventRecord(pt.e[1], ks));
	if (memkind != VDL2GPU_MEM_HOST) {	/* the caller's device buffer has been read: see the wait at the top */
		HIPCHK(h, hipEventRecord(h->in_read[ring], ks));
		h->in_rec[ring] = true;
	} else
		h->in_rec[ring] = false;
	if (staged_in) {	/* (only the staging copy of the push after next waits for it) */
		HIPCHK(h, hipEventRecord(h->k1_done[stg], ks));
		h->k1_rec[stg] = true;
	}
	/* The output ring of this push: if the push that last used it (three back) has not been collected yet, collect it now --
	 * the GPU has the two pushes in between and this push's channeliser to work on while this thread waits for that push's
	 * tail. */
	hp(1);	/* channeliser enqueued */
	if (h->ring_busy[ring]) {
		const int rch = harvest_ring(h, ring, true);
		if (rch < 0)
			return rch;
	}
	hp(2);	/* ring collected */
	{
		KInitParams ki{};
		ki.ctl = h->d_ctl[par] + CTL_STAGE;
		ki.ctl_words = (int)(h->ctl_words - CTL_STAGE);
		ki.outc = h->d_outc + 2 * ring;
		ki.fail = h->d_fail[par];
		ki.redo = h->d_redo[par];
		ki.nsc = h->S * VDL2_CS;
		ki.fmask = h->d_fmask[par];
		ki.fcnt = h->frames_on ? h->d_fcnt + 4 * ring : nullptr;
		hipLaunchKernelGGL(k_push_init, dim3(1), dim3(1024), 0, fs, ki);
	}
	if (staged)
		HIPCHK(h, hipEventRecord(pt.e[10], fs));
	{
		K2Params k2{};
		k2.dec = h->d_dec[pset];
		k2.cap = h->cap;
		k2.nbch = h->C;
		k2.nstreams = h->S;
		k2.J = J;
		k2.ss = h->d_ss;
		k2.cs = h->d_cs;
		k2.cfg = h->d_cfg;
		k2.pn = h->d_pn;
		k2.pn8 = h->d_pn8;
		k2.cands = h->d_cands[par];
		k2.clusters = h->d_clusters[par];
		k2.clhead = h->d_clhead[par];
		k2.ctl = h->d_ctl[par];
		k2.stage = h->d_stage[par];
		k2.sel_list = h->d_sel_list[par];
		k2.sel_list2 = h->d_sel_list2[par];
		k2.sel_mode = 0;
		k2.stage_cap = h->stage_cap;
		k2.recs = h->d_recs[ring];
		k2.outc = h->d_outc + 2 * ring;
		k2.outc_total_redo = h->d_outc + 8;
		k2.fmask = h->d_fmask[par];
		k2.rec_cap = h->rec_cap;
		k2.dec_base = dec_base;
		k2.scan_lo = dec_base + VDL2_HIST;	/* the scan starts at the first carried frame that has its history */
#if VDL2_PROBE_STRIDE == 2
		k2.probe_r = 0;				/* the one class scanned everywhere: fixed, not the class the channel is in */
#else
		k2.probe_r = -1;			/* no class is scanned everywhere: the probe only finds the bursts (every fourth sample of sub-phase 0), the
							 * region scan lists every class around them, the verify pass covers every stretch the chain idles through */
#endif
		k2.probe_par = (int)((dec_base + VDL2_HIST) & 1);
		k2.force_serial = serial ? 1 : 0;
		k2.sel_reserved = (!h->full_scan && !serial && h->S * VDL2_CS <= 512) ? 1 : 0;	/* (enqueue_back's `spec`) */
		k2.prim_drop = h->prim_drop;
		k2.dbg = h->knob.debug_counters ? h->d_dbg : nullptr;
		k2.headtap = h->d_headtap;
		k2.headtap_n = h->d_headtap_n;
		k2.headtap_cap = h->headtap_cap;
		if (h->d_headtap) {
			/* VDL2GPU_F_DEBUG_HEADS: one tap buffer for the handle, so the pipeline is drained first -- the back stage and the
			 * tail of the two pushes before would otherwise still be appending to it ("every trigger of the LAST push") */
			HIPCHK(h, hipStreamSynchronize(h->stream));
			HIPCHK(h, hipStreamSynchronize(h->pay_stream));
			HIPCHK(h, hipStreamSynchronize(h->copy_stream));
			HIPCHK(h, hipMemsetAsync(h->d_headtap_n, 0, sizeof(unsigned), fs));
		}
		k2.full_scan = h->full_scan;
		k2.test_noregion = noregion ? 1 : 0;
		k2.regs = h->d_regs[par];
		k2.segs = h->d_segs[par];
		k2.fail = h->d_fail[par];
		k2.redo = h->d_redo[par];
		k2.round = 0;
		k2.cs_out = h->d_cs_out[par];
		k2.skey = h->d_skey[par];
		k2.sidx = h->d_sidx[par];
		k2.prim = h->d_prim[par];
		k2.seeds = h->d_seeds[par];
		k2.onchain = h->d_onchain[par];
		k2.slog = h->d_slog[par];
		k2.win = h->d_win[par];
		k2.items = h->d_items[par];
		k2.item_cap = h->item_cap;
		k2.item_priv = h->item_priv;
		k2.drain_slot = -1;
		const unsigned tiles = (unsigned)((VDL2_CARRY_FRAMES + J) / K2A_TS + 2);
		const dim3 gch((unsigned)h->C, (unsigned)GS);
		ScanDrain pdrain, rdrain;
		if (!serial) {
			{
				/* as many workgroups as are resident at once, each walking its share of the channel's tiles */
				const unsigned want = h->full_scan ? tiles : tiles / 2 + 1;
				unsigned per = (unsigned)((h->n_cu * h->probe_occ + h->C * GS - 1) / (h->C * GS));
				per = per < 1 ? 1 : (per > want ? want : per);
				per = std::min<unsigned>(per, VDL2_MAXWG);
				/* the probe needs the carry the push before made (the first 49152 frames of this plane set); with the front
				 * stage on two streams (below) that copy is not on this stream */
				pdrain = launch_scan(SCAN_PROBE, k2, dim3(per, (unsigned)h->C, (unsigned)GS), fs, VDL2_SURV_PROBE, h->full_scan ? 0 : (VDL2_PROBE_STRIDE == 2 ? 2 : 3), 0, (h->full_scan ? 4 : 1) * ((want + per - 1) / per));
			}
			{
				K2Params k2d = k2;	/* (k2r_regions works the probe's common area off first, k2s_sort the region scan's) */
				scan_drain(k2d, pdrain);
				hipLaunchKernelGGL(k2r_regions, gch, dim3(K2R_NT), 0, fs, k2d);
			}
			rdrain = launch_scan(SCAN_REGION, k2, dim3(128, (unsigned)h->C, (unsigned)GS), fs, VDL2_SURV_REGION, 0, 1, 2);
			HIPCHK(h, hipGetLastError());
		}
		if (!serial) {
			K2Params k2d = k2;
			scan_drain(k2d, rdrain);
			hipLaunchKernelGGL(k2s_sort, gch, dim3(K2S_NT), 0, fs, k2d);
		}
		if (staged)
		HIPCHK(h, hipEventRecord(pt.e[4], fs));	/* end of the front stage's scan + sort (e[4] is free: the verify pass is timed from e[12]) */
		HIPCHK(h, hipGetLastError());
		/* ---- end of the FRONT stage */
		if (two_streams)
			HIPCHK(h, hipEventRecord(h->f_done[par], fs));
		{
			/* the carry for the NEXT push: the last 49152 frames of this push's planes (its own carry included if it is
			 * shorter) go in front of where the next push's output will start, in the other plane set -- a fixed amount,
			 * so that it does not wait for the resolver to say how much is still unconsumed (3 MB per stream).  Behind
			 * this push's scan rather than in front of the next push's channeliser: there the copy sat for 100 us
			 * behind the cluster kernel, which has the higher priority. */
			/* the next plane set's head was last read by the tail of the push two back (a repaired channel's payloads are
			 * decoded late: a burst at the very start of that push lies in its head); the next push's channeliser, right
			 * behind this copy, waits for that same tail anyway */
			if (two_streams && h->k2_rec[(par + 1) % VDL2_NSET])
				HIPCHK(h, hipStreamWaitEvent(fs, h->k2_done[(par + 1) % VDL2_NSET], 0));
			K3Params k3{};
			k3.src = h->d_dec[pset];
			k3.dst = h->d_dec[(pset + 1) % VDL2_NSET];
			k3.cap = h->cap;
			k3.nbch = h->C;
			k3.J = J;
			hipLaunchKernelGGL(k3_carry, dim3(24, (unsigned)h->C, (unsigned)GS), dim3(K3_THREADS), 0, fs, k3);
			HIPCHK(h, hipGetLastError());
			if (two_streams)	/* a following push that keeps to the main stream must see the carry (and with two front streams: the next probe) */
				HIPCHK(h, hipEventRecord(h->f_tail, fs));
			else {	/* ... and a following push's front stage this push's channeliser state and carry, made on the main stream */
				HIPCHK(h, hipEventRecord(h->k1_ev, fs));
				h->k1_ev_rec = true;
			}
		}
		h->back.valid = true;
		h->back.k2 = k2;
		h->back.J = J;
		h->back.par = par;
		h->back.ring = ring;
		h->back.slab = slab;
		h->back.staged = staged;
		h->back.serial = serial;
		h->back.two_streams = two_streams;
		h->back.tiles = tiles;
		h->back.pt_index = h->pending.size();
	}
	h->pending.push_back(pt);
	hp(3);	/* rest of the front stage enqueued */
	spill_slab(h, slab);	/* (this push's export will write the slab of the push four back: whatever of it the caller has not taken yet moves aside) */
	hp(4);
	{
		const int rcb = enqueue_back(h);
		if (rcb)
			return rcb;
	}
	hp(5);	/* back stage enqueued */
	h->last_J = J;
	h->last_two_streams = two_streams;
	h->ring_busy[ring] = true;
	h->ring_slab[ring] = slab;
	h->ring_push[ring] = h->pushes;
	h->ring_samples[ring] = nsamples;
	h->last_set = par;
	h->total_in += nsamples;
	h->pushes++;
	return VDL2GPU_OK;
}

extern "C" int vdl2gpu_sync(vdl2gpu_t *h)
{
	if (!h)
		return VDL2GPU_EINVAL;
	HLOCK(h);
	HIPCHK(h, hipSetDevice(h->cfg.device));
	HIPCHK(h, hipStreamSynchronize(h->fstream));
	HIPCHK(h, hipStreamSynchronize(h->stream));
	HIPCHK(h, hipStreamSynchronize(h->pay_stream));
	HIPCHK(h, hipStreamSynchronize(h->copy_stream));
	return harvest_timing(h);
}

/* A record in a slab holds what k_export_records sent: the header and the rows the burst uses; everything behind is zero by
 * definition (burst_payload clears a record before it fills it) and is not read from the slab. */
static inline void rec_copy(vdl2gpu_burst_t *dst, const vdl2gpu_burst_t *src, bool from_slab)
{
	if (!from_slab) {
		*dst = *src;
		return;
	}
	const int nb = std::min(std::max(src->nbrow, 0), (int)VDL2GPU_MAXROWS);
	const size_t used = offsetof(vdl2gpu_burst_t, data) + (size_t)nb * VDL2GPU_ROWLEN;
	memcpy(dst, src, used);
	memset(reinterpret_cast<char *>(dst) + used, 0, sizeof *dst - used);
}

static inline vdl2gpu_burst_t *rec_of(vdl2gpu_t *h, uint64_t hd)
{
	const unsigned src = (unsigned)(hd >> 32) & 7u;
	return (src ? h->h_slab[src - 1] : h->ready.data()) + (uint32_t)hd;
}

/* A ring's slab is about to be written again (its push's back stage is being enqueued): whatever of it has not been
 * handed out yet moves to the pageable queue.  A consumer that polls after every push never gets here with anything. */
static void spill_slab(vdl2gpu_t *h, int slab)
{
	/* only the stretch of the hand-out order that the slab's push put there (a consumer that polls rarely may have 4 x max_bursts
	 * unread entries: walking all of them in every push cost the calling thread more than enqueueing the push) */
	const size_t lo = std::max(h->slab_lo[slab], h->ready_pos), hi = std::min(h->slab_hi[slab], h->ready_idx.size());
	h->slab_lo[slab] = h->slab_hi[slab] = 0;
	for (size_t i = lo; i < hi; ++i)
		if (((h->ready_idx[i] >> 32) & 7u) == (unsigned)(1 + slab)) {
			h->ready.emplace_back();
			rec_copy(&h->ready.back(), &h->h_slab[slab][(uint32_t)h->ready_idx[i]], true);
			h->ready_idx[i] = (uint64_t)(h->ready.size() - 1);
		}
}

/* Move the records of the push that filled `ring` to the host queue.  blocking = false: only if
 * that push has finished (returns 1 if it has not).  The copy runs on its own stream, so a later
 * push keeps the GPU busy meanwhile. */
static int harvest_ring(vdl2gpu_t *h, int ring, bool blocking)
{
	if (!h->ring_busy[ring])
		return 0;
	if (!blocking) {
		const hipError_t q = hipEventQuery(h->ring_done(ring));
		if (q == hipErrorNotReady)
			return 1;
		if (q != hipSuccess) {
			h->err = std::string("hipEventQuery: ") + hipGetErrorString(q);
			return VDL2GPU_EHIP;
		}
	}
	const double hq0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
	HIPCHK(h, hipEventSynchronize(h->ring_done(ring)));
	h->hprof[6] += std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - hq0;
	const unsigned c0 = h->h_pin_cnt[32 * ring], c1 = h->h_pin_cnt[32 * ring + 1];
	const unsigned n = std::min(c0, h->rec_cap);
	h->overflowed += c1;
	if (!h->knob.split_fixed && h->ring_samples[ring]) {
		/* How long a part may be follows from how many trigger candidates the busiest channel produced per input sample
		 * in the parts collected lately (the highest of the last four): parts are sized to fill 90 % of the tables, so
		 * that traffic may grow by a tenth from one push to the next before a channel overflows them.  A channel that does
		 * overflow is handled by the serial machine for that part (exact, milliseconds); its density then counts as
		 * twice what the tables hold.  Round 2 halved the parts on an overflow and doubled them again after 1024 quiet
		 * pushes: busy channels ended up in parts a quarter full. */
		const unsigned novf = h->h_pin_cnt[32 * ring + 7];
		const unsigned maxc = h->h_pin_cnt[32 * ring + 24];
		/* (a part's scan starts at the first carried frame: its count covers the part plus 49152 frames of the one before) */
		const double span = (double)h->ring_samples[ring] + (double)VDL2_CARRY_FRAMES * (double)h->sdrclk / 21.0;
		double d = (double)std::min<unsigned>(maxc, VDL2_CAND_CAP) / span;
		if (novf)
			d = 2.0 * (double)VDL2_CAND_CAP / span;
		h->cand_dens[h->cand_dens_n++ & 3u] = d;
		double dmax = 0.0;
		for (double x : h->cand_dens)
			dmax = std::max(dmax, x);
		size_t lim = h->split_default;
		if (dmax > 0.0)
			lim = (size_t)std::min((double)h->split_default, std::max(0.0, h->knob.table_fill * (double)VDL2_CAND_CAP / dmax - (double)VDL2_CARRY_FRAMES * (double)h->sdrclk / 21.0));
		h->split_samples = std::max(h->split_unit, lim / h->split_unit * h->split_unit);
		if (novf)
			h->last_ovf_push = h->ring_push[ring];
	}
	{
		/* one round always (enqueue_back); one more after every serial redo -- a repaired chain failed its own verify pass
		 * as often as rounds were scheduled --, and back down one at a time after 256 pushes without a serial redo (what
		 * the first round repairs does not count: it is always there) */
		const unsigned redos = h->h_pin_cnt[32 * ring + 2], repairs = h->h_pin_cnt[32 * ring + 3];
		if (redos != h->redos_seen) {
			h->redos_seen = redos;
			h->last_redo_push = h->ring_push[ring];
			h->repair_rounds = std::min(4, h->repair_rounds + 1);
		} else if (h->repair_rounds > h->rounds_floor && h->ring_push[ring] > h->last_redo_push + 256) {
			h->repair_rounds--;
			h->last_redo_push = h->ring_push[ring];
		}
		h->repairs_seen = repairs;
	}
	if (n) {
		/* bounded: a consumer that never collects bursts (only frames) loses the oldest ones, counted */
		const size_t qmax = 4 * (size_t)h->rec_cap;
		if (h->ready_idx.size() - h->ready_pos > qmax) {
			const size_t drop = h->ready_idx.size() - h->ready_pos - qmax;
			h->ready_pos += drop;
			h->overflowed += drop;
		}
		if (h->ready_pos == h->ready_idx.size()) {	/* everything handed out: recycle storage */
			h->ready.clear();
			h->ready_idx.clear();
			h->ready_pos = 0;
			for (int k = 0; k < VDL2_NSLAB; ++k)
				h->slab_lo[k] = h->slab_hi[k] = 0;
		} else if (h->ready_pos > 1024 && h->ready_pos > h->ready_idx.size() / 2) {	/* the handed-out prefix is the larger part of the storage: drop it
											 * (so the storage never exceeds 2 x the unread records + one push: <= (8 + 1) x max_bursts records) */
			std::vector<vdl2gpu_burst_t> keep;
			keep.reserve(h->ready_idx.size() - h->ready_pos);
			for (size_t i = h->ready_pos; i < h->ready_idx.size(); ++i) {
				keep.emplace_back();
				rec_copy(&keep.back(), rec_of(h, h->ready_idx[i]), ((h->ready_idx[i] >> 32) & 7u) != 0);
			}
			h->ready.swap(keep);
			h->ready_idx.resize(h->ready.size());
			for (size_t i = 0; i < h->ready_idx.size(); ++i)
				h->ready_idx[i] = (uint64_t)i;
			h->ready_pos = 0;
			for (int k = 0; k < VDL2_NSLAB; ++k)	/* (no handle points into a slab any more) */
				h->slab_lo[k] = h->slab_hi[k] = 0;
		}
		/* the first slab_cap records are already in this ring's slab (k_export_records ran before the event this call
		 * waited for); a push with more than that brings the rest through the bounce buffer */
		const unsigned ns = std::min(n, h->slab_cap);
		const size_t old = h->ready.size();
		for (unsigned done = ns; done < n; done += h->pin_recs) {
			const unsigned m = std::min(h->pin_recs, n - done);
			HIPCHK(h, hipMemcpyAsync(h->h_pin, h->d_recs[ring] + done, (size_t)m * sizeof(vdl2gpu_burst_t),
						 hipMemcpyDeviceToHost, h->copy_stream));
			HIPCHK(h, hipStreamSynchronize(h->copy_stream));
			const vdl2gpu_burst_t *pin = reinterpret_cast<const vdl2gpu_burst_t *>(h->h_pin);
			h->ready.insert(h->ready.end(), pin, pin + m);
		}
		/* K2d ran ahead of the verify pass: what a repair round (or K2f's serial redo) made void of the first selection
		 * K2d's second pass has tagged (trig_sample == 2 on the device); everything else is a burst of the chain */
		const bool any = h->ring_spec[ring];
		const size_t iold = h->ready_idx.size();
		auto take = [&](vdl2gpu_burst_t &b, uint64_t handle) {
			if (any && b.trig_sample == 2)
				return;
			b.trig_sample = dec_to_sample(b.trig_dec, (unsigned)h->sdrclk);
			b.end_sample = dec_to_sample(b.end_dec, (unsigned)h->sdrclk);
			/* d8psk.c:302, same mixed float/double expression */
			b.ppm = (float)((double)(10500.0f * b.df) / (2.0 * M_PI * (double)b.Fr) * 1e6);
			h->ready_idx.push_back(handle);
		};
		for (unsigned i = 0; i < ns; ++i)
			take(h->h_slab[h->ring_slab[ring]][i], ((uint64_t)(1 + h->ring_slab[ring]) << 32) | i);
		for (size_t i = old; i < h->ready.size(); ++i)
			take(h->ready[i], (uint64_t)i);
		h->slab_lo[h->ring_slab[ring]] = iold;
		h->slab_hi[h->ring_slab[ring]] = h->ready_idx.size();
		std::sort(h->ready_idx.begin() + iold, h->ready_idx.end(), [h](uint64_t x, uint64_t y) {
			const vdl2gpu_burst_t &a = *rec_of(h, x), &b = *rec_of(h, y);
			if (a.end_dec != b.end_dec)
				return a.end_dec < b.end_dec;
			if (a.stream != b.stream)
				return a.stream < b.stream;
			return a.chn < b.chn;
		});
	}
	if (h->frames_on) {
		const unsigned arena0 = h->rec_cap * K4_SLOT;
		const unsigned nbytes = std::min(h->h_pin_cnt[32 * ring + 6], h->frame_cap - arena0);
		h->frames_dropped += h->h_pin_cnt[32 * ring + 5];
		if (n) {
			{	/* bounded like the burst queue: the oldest frames go, counted */
				const size_t qmax = 4 * (size_t)h->rec_cap;
				if (h->fready_idx.size() - h->fready_pos > qmax) {
					const size_t drop = h->fready_idx.size() - h->fready_pos - qmax;
					h->fready_pos += drop;
					h->frames_dropped += drop;
				}
			}
			if (h->fready_pos == h->fready_idx.size()) {
				h->fready.clear();
				h->fready_idx.clear();
				h->fready_pos = 0;
			} else if (h->fready_pos > 1024 && h->fready_pos > h->fready_idx.size() / 2) {	/* compact: entries are self-delimiting */
				std::vector<uint8_t> keep;
				std::vector<size_t> kidx;
				const size_t hdr0 = offsetof(vdl2gpu_frame_t, data);
				for (size_t i = h->fready_pos; i < h->fready_idx.size(); ++i) {
					const uint8_t *e = h->fready.data() + h->fready_idx[i];
					const size_t sz = (hdr0 + (size_t)reinterpret_cast<const vdl2gpu_frame_t *>(e)->len + 7) & ~(size_t)7;
					kidx.push_back(keep.size());
					keep.insert(keep.end(), e, e + sz);
				}
				h->fready.swap(keep);
				h->fready_idx.swap(kidx);
				h->fready_pos = 0;
			}
			const size_t old = h->fready.size();
			const size_t hdr = offsetof(vdl2gpu_frame_t, data);
			const size_t pin_bytes = (size_t)h->pin_recs * sizeof(vdl2gpu_burst_t);
			/* the records' slots: keep the occupied ones, packed like arena entries */
			const size_t slot_bytes = (size_t)n * K4_SLOT;
			for (size_t done = 0; done < slot_bytes; done += pin_bytes) {
				const size_t m = std::min(pin_bytes, slot_bytes - done);
				HIPCHK(h, hipMemcpyAsync(h->h_pin, (const char *)h->d_frames[ring] + done, m, hipMemcpyDeviceToHost, h->copy_stream));
				HIPCHK(h, hipStreamSynchronize(h->copy_stream));
				const uint8_t *pin = reinterpret_cast<const uint8_t *>(h->h_pin);
				for (size_t o = 0; o < m; o += K4_SLOT) {
					const vdl2gpu_frame_t *f = reinterpret_cast<const vdl2gpu_frame_t *>(pin + o);
					if (f->len <= 0 || hdr + (size_t)f->len > K4_SLOT)
						continue;
					const size_t sz = (hdr + (size_t)f->len + 7) & ~(size_t)7;
					h->fready.insert(h->fready.end(), pin + o, pin + o + sz);
				}
			}
			if (nbytes) {
				const size_t at = h->fready.size();
				h->fready.resize(at + nbytes);
				for (size_t done = 0; done < nbytes; done += pin_bytes) {
					const size_t m = std::min(pin_bytes, (size_t)nbytes - done);
					HIPCHK(h, hipMemcpyAsync(h->h_pin, (const char *)h->d_frames[ring] + arena0 + done, m, hipMemcpyDeviceToHost,
								 h->copy_stream));
					HIPCHK(h, hipStreamSynchronize(h->copy_stream));
					memcpy(h->fready.data() + at + done, h->h_pin, m);
				}
			}
			/* walk the entries (56 header bytes + len data bytes, rounded up to 8) */
			const size_t iold = h->fready_idx.size();
			for (size_t off = old; off + hdr <= h->fready.size();) {
				vdl2gpu_frame_t *f = reinterpret_cast<vdl2gpu_frame_t *>(h->fready.data() + off);
				if (f->len < 0 || f->len > VDL2GPU_MAXFRAME || off + hdr + (size_t)f->len > h->fready.size())
					break;
				f->ppm = (float)((double)(10500.0f * f->df) / (2.0 * M_PI * (double)f->Fr) * 1e6);	/* d8psk.c:302 */
				f->block = -1;
				h->fready_idx.push_back((uint32_t)off);
				off += (hdr + (size_t)f->len + 7) & ~(size_t)7;
			}
			const uint8_t *fd = h->fready.data();
			std::sort(h->fready_idx.begin() + iold, h->fready_idx.end(), [fd](size_t x, size_t y) {
				const vdl2gpu_frame_t &a = *reinterpret_cast<const vdl2gpu_frame_t *>(fd + x);
				const vdl2gpu_frame_t &b = *reinterpret_cast<const vdl2gpu_frame_t *>(fd + y);
				if (a.end_dec != b.end_dec)
					return a.end_dec < b.end_dec;
				if (a.stream != b.stream)
					return a.stream < b.stream;
				if (a.chn != b.chn)
					return a.chn < b.chn;
				return a.seq < b.seq;
			});
		}
	}
	h->ring_busy[ring] = false;
	return 0;
}

static int harvest_all(vdl2gpu_t *h, bool blocking)
{
	/* oldest push first */
	int order[VDL2_NRING], n = 0;
	for (int r = 0; r < VDL2_NRING; ++r)
		if (h->ring_busy[r])
			order[n++] = r;
	std::sort(order, order + n, [&](int a, int b) { return h->ring_push[a] < h->ring_push[b]; });
	for (int k = 0; k < n; ++k) {
		const int rc = harvest_ring(h, order[k], blocking);
		if (rc < 0)
			return rc;
		if (rc == 1)
			break;	/* not finished yet; anything newer is not finished either */
	}
	return 0;
}

/* Collect everything pushed before this call, waiting for the GPU where it has to -- with the handle's lock RELEASED during every
 * wait, so that a producer thread keeps committing blocks while a consumer thread sits in vdl2gpu_poll().  A ring that the
 * producer reuses meanwhile was collected by the producer's own call first (push_impl), in order. */
static int wait_harvest(vdl2gpu_t *h, std::unique_lock<std::recursive_mutex> &lk)
{
	const uint64_t upto = h->pushes;
	for (;;) {
		int r = -1;
		for (int k = 0; k < VDL2_NRING; ++k)
			if (h->ring_busy[k] && h->ring_push[k] < upto && (r < 0 || h->ring_push[k] < h->ring_push[r]))
				r = k;
		if (r < 0)
			return 0;
		const int rc = harvest_ring(h, r, false);
		if (rc < 0)
			return rc;
		if (rc == 1) {
			hipEvent_t ev = h->ring_done(r);
			lk.unlock();
			const hipError_t e = hipEventSynchronize(ev);
			lk.lock();
			if (e != hipSuccess) {
				h->err = std::string("hipEventSynchronize(ring_done): ") + hipGetErrorString(e);
				return VDL2GPU_EHIP;
			}
		}
	}
}

static int hand_out(vdl2gpu_t *h, vdl2gpu_burst_t *out, int max)
{
	const int n = std::min<int>(max, (int)(h->ready_idx.size() - h->ready_pos));
	for (int i = 0; i < n; ++i) {
		const uint64_t hd = h->ready_idx[h->ready_pos + i];
		rec_copy(out + i, rec_of(h, hd), ((hd >> 32) & 7u) != 0);
	}
	h->ready_pos += (size_t)n;
	return n;
}

/* ---------------------------------------------------------------- block path */
extern "C" int vdl2gpu_decode_blocks(vdl2gpu_t *h, const vdl2gpu_burst_t *blocks, int n,
				     vdl2gpu_frame_t *frames, int max_frames, int *dropped)
{
	if (!h || n < 0 || max_frames < 0 || (n > 0 && !blocks) || (max_frames > 0 && !frames))
		return VDL2GPU_EINVAL;
	if (dropped)
		*dropped = 0;
	if (n == 0 || max_frames == 0)
		return 0;
	HLOCK(h);
	HIPCHK(h, hipSetDevice(h->cfg.device));
	vdl2gpu_burst_t *d_blk = nullptr;
	vdl2gpu_frame_t *d_fr = nullptr;
	unsigned *d_cnt = nullptr;
	auto cleanup = [&]() {
		(void)hipFree(d_blk);
		(void)hipFree(d_fr);
		(void)hipFree(d_cnt);
	};
	hipError_t e = hipMalloc(&d_blk, (size_t)n * sizeof(vdl2gpu_burst_t));
	if (e == hipSuccess)
		e = hipMalloc(&d_fr, (size_t)max_frames * sizeof(vdl2gpu_frame_t));
	if (e == hipSuccess)
		e = hipMalloc(&d_cnt, 4 * sizeof(unsigned));
	/* on the stream the kernel runs on: that stream does not wait for the legacy default stream, and a
	 * plain hipMemset() that lands after the kernel's first atomics loses frames */
	if (e == hipSuccess)
		e = hipMemcpyAsync(d_blk, blocks, (size_t)n * sizeof(vdl2gpu_burst_t), hipMemcpyHostToDevice, h->copy_stream);
	if (e == hipSuccess)
		e = hipMemsetAsync(d_cnt, 0, 4 * sizeof(unsigned), h->copy_stream);
	unsigned cnt[2] = {0, 0};
	if (e == hipSuccess) {
		K4Params k4{};
		k4.recs = d_blk;
		k4.nrecs_dev = nullptr;
		k4.nrecs = (unsigned)n;
		k4.rec_cap = (unsigned)n;
		k4.frames = d_fr;
		k4.nframes = d_cnt;
		k4.frame_cap = (unsigned)max_frames;
		k4.compact = 0;
		k4.tabs = h->d_k4tab;
		const unsigned grid = (unsigned)std::min<long long>(n, (long long)h->n_cu * 32);
		hipLaunchKernelGGL(k4_frames, dim3(grid), dim3(K4_NT), 0, h->copy_stream, k4);
		e = hipGetLastError();
		if (e == hipSuccess)
			e = hipStreamSynchronize(h->copy_stream);
	}
	if (e == hipSuccess)
		e = hipMemcpyAsync(cnt, d_cnt, sizeof cnt, hipMemcpyDeviceToHost, h->copy_stream);
	if (e == hipSuccess)
		e = hipStreamSynchronize(h->copy_stream);
	const int nf = (int)std::min<unsigned>(cnt[0], (unsigned)max_frames);
	if (e == hipSuccess && nf > 0) {
		e = hipMemcpyAsync(frames, d_fr, (size_t)nf * sizeof(vdl2gpu_frame_t), hipMemcpyDeviceToHost, h->copy_stream);
		if (e == hipSuccess)
			e = hipStreamSynchronize(h->copy_stream);
	}
	cleanup();
	if (e != hipSuccess) {
		h->err = std::string("vdl2gpu_decode_blocks: ") + hipGetErrorString(e);
		return VDL2GPU_EHIP;
	}
	if (dropped)
		*dropped = (int)cnt[1];
	std::sort(frames, frames + nf, [](const vdl2gpu_frame_t &a, const vdl2gpu_frame_t &b) {
		return a.block != b.block ? a.block < b.block : a.seq < b.seq;
	});
	return nf;
}

static int poll_frames_impl(vdl2gpu_t *h, vdl2gpu_frame_t *out, int max, bool blocking)
{
	if (!h || (max > 0 && !out) || max < 0)
		return VDL2GPU_EINVAL;
	if (!h->frames_on) {
		h->err = "vdl2gpu_poll_frames needs VDL2GPU_F_FRAMES";
		return VDL2GPU_EINVAL;
	}
	HLOCK(h);
	HIPCHK(h, hipSetDevice(h->cfg.device));
	int rc = blocking ? wait_harvest(h, hlock_) : harvest_all(h, false);
	if (rc)
		return rc;
	const int n = std::min<int>(max, (int)(h->fready_idx.size() - h->fready_pos));
	for (int i = 0; i < n; ++i) {	/* only what is meaningful of the record: its head and data[0..len) */
		const uint8_t *e = h->fready.data() + h->fready_idx[h->fready_pos + i];
		const vdl2gpu_frame_t *f = reinterpret_cast<const vdl2gpu_frame_t *>(e);
		memcpy(&out[i], e, offsetof(vdl2gpu_frame_t, data) + (size_t)f->len);
	}
	h->fready_pos += (size_t)n;
	return n;
}

extern "C" int vdl2gpu_poll_frames(vdl2gpu_t *h, vdl2gpu_frame_t *out, int max)
{
	return poll_frames_impl(h, out, max, true);
}

extern "C" int vdl2gpu_poll_frames_ready(vdl2gpu_t *h, vdl2gpu_frame_t *out, int max)
{
	return poll_frames_impl(h, out, max, false);
}

extern "C" int vdl2gpu_pending(vdl2gpu_t *h)
{
	if (!h)
		return VDL2GPU_EINVAL;
	HLOCK(h);
	HIPCHK(h, hipSetDevice(h->cfg.device));
	int rc = wait_harvest(h, hlock_);
	if (rc)
		return rc;
	return (int)(h->ready_idx.size() - h->ready_pos);
}

/* Pushes the GPU has not finished yet (0..3).  Never waits. */
extern "C" int vdl2gpu_inflight(vdl2gpu_t *h)
{
	if (!h)
		return VDL2GPU_EINVAL;
	HLOCK(h);
	if (h->failed)
		return VDL2GPU_EHIP;
	HIPCHK(h, hipSetDevice(h->cfg.device));
	int n = 0;
	for (int r = 0; r < VDL2_NRING; ++r)
		if (h->ring_busy[r]) {
			const hipError_t q = hipEventQuery(h->ring_done(r));
			if (q == hipErrorNotReady)
				++n;
			else if (q != hipSuccess) {
				h->err = std::string("hipEventQuery: ") + hipGetErrorString(q);
				return VDL2GPU_EHIP;
			}
		}
	return n;
}

extern "C" int vdl2gpu_poll(vdl2gpu_t *h, vdl2gpu_burst_t *out, int max)
{
	if (!h || (max > 0 && !out) || max < 0)
		return VDL2GPU_EINVAL;
	HLOCK(h);
	if (h->failed)
		return VDL2GPU_EHIP;
	HIPCHK(h, hipSetDevice(h->cfg.device));
	int rc = wait_harvest(h, hlock_);
	if (rc)
		return rc;
	return hand_out(h, out, max);
}

extern "C" int vdl2gpu_poll_ready(vdl2gpu_t *h, vdl2gpu_burst_t *out, int max)
{
	if (!h || (max > 0 && !out) || max < 0)
		return VDL2GPU_EINVAL;
	HLOCK(h);
	if (h->failed)
		return VDL2GPU_EHIP;
	HIPCHK(h, hipSetDevice(h->cfg.device));
	int rc = harvest_all(h, false);
	if (rc)
		return rc;
	return hand_out(h, out, max);
}

extern "C" int vdl2gpu_get_stats(vdl2gpu_t *h, vdl2gpu_stats_t *out)
{
	if (!h || !out)
		return VDL2GPU_EINVAL;
	HLOCK(h);
	int rc = vdl2gpu_sync(h);
	if (rc)
		return rc;
	std::vector<ChanState> cs((size_t)h->S * VDL2_CS);
	HIPCHK(h, hipMemcpy(cs.data(), h->d_cs, cs.size() * sizeof(ChanState), hipMemcpyDeviceToHost));
	memset(out, 0, sizeof *out);
	out->samples_in = h->total_in;
	out->dec_samples = (uint64_t)(((unsigned __int128)h->total_in * 21u) / (unsigned)h->sdrclk);
	for (int s = 0; s < h->S; ++s)
		for (int c = 0; c < h->C; ++c) {
			const ChanState &x = cs[(size_t)s * VDL2_CS + c];
			out->sync_evals += x.n_eval;
			out->triggers += x.n_trig;
			out->header_rejects += x.n_reject;
			out->bursts += x.n_burst;
			out->deferrals += x.n_defer;
			out->serial_samples += x.n_slow;
			out->candidates += x.n_cand;
			out->serial_redos += x.n_redo;
		}
	out->overflowed = h->overflowed;
	out->frames_dropped = h->frames_dropped;
	{	/* the running total on the device (k2f_commit adds to it), like the channel counters above: current whether or not
		 * the caller has collected the pushes yet */
		unsigned rep = 0;
		HIPCHK(h, hipMemcpy(&rep, h->d_outc + 9, sizeof rep, hipMemcpyDeviceToHost));
		out->repairs = rep;
	}
	return VDL2GPU_OK;
}

/* Where the calling thread's time inside vdl2gpu_push() went, in seconds, summed over the pushes since the last reset (no pipeline
 * drain: it is host bookkeeping): out[0] waiting for the device buffer of the push before last, [1] enqueueing the channeliser,
 * [2] collecting the output ring this push will reuse (mostly: WAITING for the push three back to finish -- the GPU is the slower
 * side then), [3] enqueueing the rest of the front stage, [4] moving uncollected records out of the slab's way, [5] enqueueing the
 * back stage and the tail, [6] of [2] and of the polling calls: waiting for a ring's completion event, [7] pushes counted. */
extern "C" int vdl2gpu_get_host_profile(vdl2gpu_t *h, double *out8, int reset)
{
	if (!h || !out8)
		return VDL2GPU_EINVAL;
	HLOCK(h);
	for (int i = 0; i < 7; ++i)
		out8[i] = h->hprof[i];
	out8[7] = (double)(h->pushes - h->hprof_push0);
	if (reset) {
		for (int i = 0; i < 7; ++i)
			h->hprof[i] = 0.0;
		h->hprof_push0 = h->pushes;
	}
	return VDL2GPU_OK;
}

extern "C" int vdl2gpu_get_timing(vdl2gpu_t *h, vdl2gpu_timing_t *out, int reset)
{
	if (!h || !out)
		return VDL2GPU_EINVAL;
	HLOCK(h);
	int rc = vdl2gpu_sync(h);
	if (rc)
		return rc;
	*out = h->tm;
	{
		/* stage sums: mean of the pushes that carried stage events x all pushes (the staged ones are every
		 * stage_every-th: scaling their sum by stage_every was biased whenever pushes % stage_every != 0) */
		const double k = h->st_pushes ? (double)h->tm.pushes / (double)h->st_pushes : 0.0;
		out->channelise_ms = k * h->st_k1;
		out->scan_ms = k * h->st_scan;
		out->cluster_ms = k * h->st_cluster;
		out->resolve_ms = k * h->st_resolve;
		out->demod_ms = k * h->st_demod;
		out->other_ms = k * h->st_other;
	}
	if (reset) {
		h->tm = vdl2gpu_timing_t{};
		h->st_scan = h->st_cluster = h->st_resolve = h->st_demod = h->st_other = h->st_k1 = 0;
		h->st_pushes = 0;
	}
	return VDL2GPU_OK;
}

/* ----------------------------------------------------------------- diagnostics */
extern "C" int64_t vdl2gpu_debug_dec(vdl2gpu_t *h, int stream, int ch, float *out, int64_t max_complex)
{
	if (!h || stream < 0 || stream >= h->S || ch < 0 || ch >= h->C || !out)
		return VDL2GPU_EINVAL;
	HLOCK(h);
	if (!h->pushes)
		return VDL2GPU_EINVAL;
	int rc = vdl2gpu_sync(h);
	if (rc)
		return rc;
	StreamState ss;
	HIPCHK(h, hipMemcpy(&ss, h->d_ss + stream, sizeof ss, hipMemcpyDeviceToHost));
	const int par = h->last_set;
	const int64_t n = std::min<int64_t>(ss.last_J, max_complex);
	if (n <= 0)
		return 0;
	HIPCHK(h, hipMemcpy(out, h->d_dec[par] + ((size_t)stream * VDL2_CS + ch) * h->cap + ss.last_fill,
			    (size_t)n * sizeof(float2), hipMemcpyDeviceToHost));
	return n;
}

extern "C" int vdl2gpu_debug_lo(vdl2gpu_t *h, int stream, int ch, float *out, int max_complex)
{
	if (!h || stream < 0 || stream >= h->S || ch < 0 || ch >= h->C || !out || max_complex < h->L)
		return VDL2GPU_EINVAL;
	HLOCK(h);
	HIPCHK(h, hipSetDevice(h->cfg.device));
	HIPCHK(h, hipStreamSynchronize(h->stream));
	HIPCHK(h, hipMemcpy(out, h->d_lo + ((size_t)stream * VDL2_CS + ch) * h->L, (size_t)h->L * sizeof(float2), hipMemcpyDeviceToHost));
	return h->L;
}

extern "C" int vdl2gpu_debug_atan2f(vdl2gpu_t *h, const float *y, const float *x, float *out, size_t n)
{
	if (!h || !y || !x || !out)
		return VDL2GPU_EINVAL;
	HLOCK(h);
	if (!n)
		return VDL2GPU_OK;
	HIPCHK(h, hipSetDevice(h->cfg.device));
	float *d = nullptr;
	HIPCHK(h, hipMalloc(&d, 3 * n * sizeof(float)));
	hipError_t e = hipMemcpyAsync(d, y, n * sizeof(float), hipMemcpyHostToDevice, h->stream);
	if (e == hipSuccess)
		e = hipMemcpyAsync(d + n, x, n * sizeof(float), hipMemcpyHostToDevice, h->stream);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_atan2f, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, d, d + n, d + 2 * n, n);
		e = hipStreamSynchronize(h->stream);
	}
	if (e == hipSuccess)
		e = hipMemcpy(out, d + 2 * n, n * sizeof(float), hipMemcpyDeviceToHost);
	(void)hipFree(d);
	if (e != hipSuccess) {
		h->err = hipGetErrorString(e);
		return VDL2GPU_EHIP;
	}
	return VDL2GPU_OK;
}

extern "C" int vdl2gpu_debug_counters(vdl2gpu_t *h, unsigned long long *out, int n, int reset)
{
	if (!h || !out || n < 0 || n > 64)
		return VDL2GPU_EINVAL;
	HLOCK(h);
	int rc = vdl2gpu_sync(h);
	if (rc)
		return rc;
	HIPCHK(h, hipMemcpy(out, h->d_dbg, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
	if (reset) {
		HIPCHK(h, hipMemsetAsync(h->d_dbg, 0, 64 * sizeof(unsigned long long), h->stream));
		HIPCHK(h, hipStreamSynchronize(h->stream));
	}
	return VDL2GPU_OK;
}

/* candidates of (stream, channel index) found by the last push's sync scan: 6 ints per candidate
 * {nrel, r, bits(p2err), bits(perr), bits(err), bits(pfr)}; returns the count */
extern "C" int vdl2gpu_debug_cands(vdl2gpu_t *h, int stream, int ch, int *out, int max_cands)
{
	if (!h || stream < 0 || stream >= h->S || ch < 0 || ch >= h->C || !out)
		return VDL2GPU_EINVAL;
	HLOCK(h);
	int rc = vdl2gpu_sync(h);
	if (rc)
		return rc;
	const int sc = stream * VDL2_CS + ch;
	unsigned n = 0;
	HIPCHK(h, hipMemcpy(&n, h->d_ctl[h->last_set] + CTL_CAND0 + sc, sizeof n, hipMemcpyDeviceToHost));
	n = std::min<unsigned>(n, VDL2_CAND_CAP);
	n = std::min<unsigned>(n, (unsigned)max_cands);
	if (n)
		HIPCHK(h, hipMemcpy(out, h->d_cands[h->last_set] + (size_t)sc * VDL2_CAND_CAP, (size_t)n * sizeof(Cand), hipMemcpyDeviceToHost));
	return (int)n;
}

/* verify result of the last push per (stream, channel slot): stream-relative position of the first
 * detector hit the tables lacked, or >= 0x7f000000 when the push verified */
/* diagnostics: the cluster heads (cl_pack) of the last push's candidates, in vdl2gpu_debug_cands()'s order */
extern "C" int vdl2gpu_debug_clheads(vdl2gpu_t *h, int stream, int ch, int *out, int max_cands)
{
	if (!h || stream < 0 || stream >= h->S || ch < 0 || ch >= h->C || !out)
		return VDL2GPU_EINVAL;
	HLOCK(h);
	int rc = vdl2gpu_sync(h);
	if (rc)
		return rc;
	const int sc = stream * VDL2_CS + ch;
	unsigned n = 0;
	HIPCHK(h, hipMemcpy(&n, h->d_ctl[h->last_set] + CTL_CAND0 + sc, sizeof n, hipMemcpyDeviceToHost));
	n = std::min<unsigned>(n, VDL2_CAND_CAP);
	n = std::min<unsigned>(n, (unsigned)max_cands);
	if (n)
		HIPCHK(h, hipMemcpy(out, h->d_clhead[h->last_set] + (size_t)sc * VDL2_CAND_CAP, (size_t)n * sizeof(int2), hipMemcpyDeviceToHost));
	return (int)n;
}

extern "C" int vdl2gpu_debug_fail(vdl2gpu_t *h, int *out, int n)
{
	if (!h || !out || n < h->S * VDL2_CS)
		return VDL2GPU_EINVAL;
	HLOCK(h);
	int rc = vdl2gpu_sync(h);
	if (rc)
		return rc;
	for (int st = 0; st < h->S; ++st)	/* (every stream from the set of its last pass) */
		HIPCHK(h, hipMemcpy(out + (size_t)st * VDL2_CS, h->d_fail[h->last_set] + (size_t)st * VDL2_CS, VDL2_CS * sizeof(int), hipMemcpyDeviceToHost));
	return h->S * VDL2_CS;
}

extern "C" int vdl2gpu_debug_segs(vdl2gpu_t *h, int stream, int ch, int *out, int max_segs)
{
	if (!h || stream < 0 || stream >= h->S || ch < 0 || ch >= h->C || !out)
		return VDL2GPU_EINVAL;
	HLOCK(h);
	int rc = vdl2gpu_sync(h);
	if (rc)
		return rc;
	const int sc = stream * VDL2_CS + ch;
	unsigned n = 0;
	HIPCHK(h, hipMemcpy(&n, h->d_ctl[h->last_set] + CTL_CAND0 + 3 * (size_t)h->S * VDL2_CS + sc, sizeof n, hipMemcpyDeviceToHost));
	n = std::min<unsigned>(n, (unsigned)std::min(max_segs, VDL2_SEG_CAP));
	if (n)
		HIPCHK(h, hipMemcpy(out, h->d_segs[h->last_set] + (size_t)sc * VDL2_SEG_CAP, (size_t)n * sizeof(Seg), hipMemcpyDeviceToHost));
	return (int)n;
}

/* VDL2GPU_F_DEBUG_HEADS: the header soft bits of every sync trigger any kernel of the LAST push handled -- the
 * clusters of all eight timing classes, the resolver's serial stretches, a serial redo --, 34 x 32-bit words each:
 * {nstar lo, nstar hi, stream * 8 + channel slot, clk0, p2err, perr, err, pfr (float bits), soft[25] (float bits), pad}.
 * Returns the number of entries written to `out` (<= max_entries), in no particular order. */
extern "C" int vdl2gpu_debug_heads(vdl2gpu_t *h, uint32_t *out, int max_entries)
{
	if (!h || !out || max_entries < 0)
		return VDL2GPU_EINVAL;
	HLOCK(h);
	if (!h->d_headtap) {
		h->err = "vdl2gpu_debug_heads needs VDL2GPU_F_DEBUG_HEADS";
		return VDL2GPU_EINVAL;
	}
	int rc = vdl2gpu_sync(h);
	if (rc)
		return rc;
	unsigned n = 0;
	HIPCHK(h, hipMemcpy(&n, h->d_headtap_n, sizeof n, hipMemcpyDeviceToHost));
	n = std::min(n, h->headtap_cap);
	n = std::min<unsigned>(n, (unsigned)max_entries);
	std::vector<HeadTap> tmp(n);
	if (n)
		HIPCHK(h, hipMemcpy(tmp.data(), h->d_headtap, (size_t)n * sizeof(HeadTap), hipMemcpyDeviceToHost));
	for (unsigned i = 0; i < n; ++i) {
		uint32_t *o = out + 34 * (size_t)i;
		const HeadTap &e = tmp[i];
		o[0] = (uint32_t)((unsigned long long)e.nstar & 0xffffffffu);
		o[1] = (uint32_t)((unsigned long long)e.nstar >> 32);
		o[2] = (uint32_t)e.sc;
		o[3] = (uint32_t)e.clk0;
		memcpy(o + 4, &e.p2err, 4);
		memcpy(o + 5, &e.perr, 4);
		memcpy(o + 6, &e.err, 4);
		memcpy(o + 7, &e.pfr, 4);
		memcpy(o + 8, e.soft, 100);
		o[33] = 0;
	}
	return (int)n;
}
