/*
 * vdl2gpu.h -- C ABI of libvdl2gpu.so: the MI355X-native VDL Mode 2 front end.
 *
 * Drop-in boundary.  The reference has no plugin API; its
 * seam is the translation-unit pair d8psk.c + viterbi.c, entered through
 *
 *     void *rcv_thread(void *arg)            d8psk.c:335   (one pthread per channel)
 *     int   initD8psk(channel_t *ch)         d8psk.c:28
 *     unsigned reversebits(unsigned, int)    d8psk.c:39    (also used by out.c:429-432)
 *
 * consuming the shared sample block `Cbuff` (rtl.c:273 / air.c:190, 32768 samples
 * per hand-off, vdlm2.h:35) with per-channel `thread_param_t{chn,Fr,Fo}`
 * (vdlm2.h:49-52) and producing `msgblk_t` records (vdlm2.h:39-47) through
 * `decodeVdlm2(channel_t*)` (vdlm2.c:189).  This library replaces exactly that:
 * raw sample blocks in, burst records out, everything between on the GPU.
 *
 *     reference interface                      replaced by
 *     ---------------------------------------  -----------------------------------
 *     rcv_thread() x nbch + Bar1/Bar2           vdl2gpu_create() + vdl2gpu_push()
 *     in_callback() cu8->float, rtl.c:285-292   fused into the channeliser kernel
 *     rx_callback() real f32,  air.c:206-208    VDL2GPU_FMT_F32R
 *     thread_param_t            vdlm2.h:49-52   vdl2gpu_chan_t (same three ints)
 *     decodeVdlm2(ch)           vdlm2.c:189     vdl2gpu_poll() -> vdl2gpu_burst_t,
 *                                               vdl2gpu_burst_to_msgblk() fills the
 *                                               reference's msgblk_t byte layout
 *     initD8psk / viterbi_*     d8psk.c:28,     internal to the kernels
 *                               viterbi.c:37-96
 *     reversebits               d8psk.c:39      reversebits() still exported
 *
 * Plain C types only; no torch / HIP types cross this boundary.  `iq` may be a
 * host pointer (copied with hipMemcpyAsync) or a device pointer (used in place).
 * All functions return 0 on success or a negative VDL2GPU_E* code; none aborts.
 */
#ifndef VDL2GPU_H
#define VDL2GPU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* libvdl2gpu.so is built -fvisibility=hidden with an export map (csrc/vdl2gpu.map): what this header declares is what it exports */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define VDL2GPU_ABI_VERSION 6	/* 3: vdl2gpu_debug_heads, VDL2GPU_F_DEBUG_HEADS, VDL2GPU_MSGBLK_*; 4: vdl2gpu_stats_t.repairs; 5: vdl2gpu_inflight, the handle lock ("Threads");
				 * 6: vdl2gpu_get_host_profile, vdl2gpu_debug_clheads */
#define VDL2GPU_MAXCH 8		/* MAXNBCHANNELS vdlm2.h:26 */
#define VDL2GPU_MAXROWS 8	/* bursts with more rows are rejected, d8psk.c:103 */
#define VDL2GPU_ROWLEN 255

enum {
	VDL2GPU_OK = 0,
	VDL2GPU_EINVAL = -1,	/* bad argument / configuration */
	VDL2GPU_EHIP = -2,	/* a HIP runtime call failed (see vdl2gpu_last_error) */
	VDL2GPU_ENOMEM = -3,
	VDL2GPU_EOVERFLOW = -4,	/* reserved; record overflow is reported through vdl2gpu_stats_t.overflowed, never as an error */
	VDL2GPU_ENODEV = -5	/* no usable GPU: the library never falls back to the CPU */
};

/* sample formats of the wideband stream */
enum {
	VDL2GPU_FMT_CU8 = 0,	/* interleaved u8 I,Q; x = (float)b - 127.37f  (rtl.c:287-289) */
	VDL2GPU_FMT_CS16 = 1,	/* interleaved s16 I,Q; x = (float)v              */
	VDL2GPU_FMT_CF32 = 2,	/* interleaved f32 I,Q  (what Cbuff holds with WITH_RTL)     */
	VDL2GPU_FMT_F32R = 3	/* real f32 (Cbuff with WITH_AIR, air.c:190)                 */
};

enum { VDL2GPU_MEM_HOST = 0, VDL2GPU_MEM_DEVICE = 1 };

/* == thread_param_t, vdlm2.h:49-52 */
typedef struct {
	int32_t chn;		/* channel number reported back in every burst */
	int32_t Fr;		/* channel frequency, Hz (only used for ppm, d8psk.c:302) */
	int32_t Fo;		/* offset from the tuner centre, Hz (d8psk.c:354) */
} vdl2gpu_chan_t;

typedef struct {
	uint32_t struct_size;	/* sizeof(vdl2gpu_config_t), for ABI growth */
	uint32_t sdrinrate;	/* SDRINRATE: 2000000 (rtl.c:36), 5/6 MS/s (air.c:134), 10 MS/s */
	uint32_t sdrclk;	/* SDRCLK; 0 = sdrinrate/4000 (rtl.c:37, air.c:138) */
	int32_t fmt;		/* VDL2GPU_FMT_* */
	int32_t nbch;		/* channels per wideband stream, 1..8 */
	int32_t nstreams;	/* independent wideband streams decoded side by side (>=1) */
	const vdl2gpu_chan_t *chan;	/* nstreams*nbch entries, stream-major */
	uint64_t max_push;	/* largest nsamples a single vdl2gpu_push() will carry.  Throughput grows with the push (≈ 0.4 ms of
				 * fixed work per push).  The parallel sync tables of a push hold 4096 trigger candidates per channel
				 * (a burst leaves about 18: a channel at 4 bursts a second fills them in 50 s, a saturated one in
				 * 16 s); a channel that exceeds them is handled by the serial machine for that stretch (exact, ~100x
				 * slower; stats.serial_samples shows it).  Long pushes are therefore cut into equal parts inside the
				 * library -- the bursts are the same for any cut --: 8.4 s of air time until the first pushes have
				 * been collected, then as long as fills 90 % of the tables at the candidate density of the busiest
				 * channel over the last four parts, at most 36 s (72 MS at 2 MS/s). */
	int32_t device;		/* HIP device ordinal */
	uint32_t max_bursts;	/* burst-record ring capacity (0 = default 65536) */
	uint32_t flags;		/* VDL2GPU_F_* */
} vdl2gpu_config_t;

#define VDL2GPU_F_KEEP_DEC 1u	/* accepted for compatibility: the last push's decimated stream is always kept (vdl2gpu_debug_dec) */
#define VDL2GPU_F_FULLSCAN 4u	/* scan all four FIR sub-phases everywhere instead of probe + regions + verify */
#define VDL2GPU_F_TEST_NOREGION 8u	/* test hook: drop the region scan; the verify pass must then redo channels serially.  Only
					 * libvdl2gpu_test.so (the same sources with -DVDL2GPU_TESTHOOKS) honours it, together with the
					 * VDL2GPU_PRIM_DROP / VDL2GPU_SPLIT_SAMPLES environment handicaps; libvdl2gpu.so rejects the
					 * flag with VDL2GPU_EINVAL and never reads those variables */
#define VDL2GPU_F_FRAMES 16u	/* run the block path (RS, HDLC, FCS) on every push's bursts as well: vdl2gpu_poll_frames() */
#define VDL2GPU_F_SERIAL 2u	/* diagnostics: skip the parallel sync tables, one serial machine per channel */
#define VDL2GPU_F_RTL_QUIRK 32u	/* cu8 only: reproduce in_callback() as written (rtl.c:285-292): in every hand-off block of
				 * 32768 samples, sample k is stored at index k+1, index 0 stays 0 and the last sample is
				 * lost -- what the reference decodes on a real RTL stick.  Every push must
				 * then be a whole number of 32768-sample blocks (RTLINBUFSZ/2, vdlm2.h:35). */

#define VDL2GPU_F_DEBUG_HEADS 64u	/* diagnostics: keep the header soft bits of every sync trigger of the last push (vdl2gpu_debug_heads) */

/* One decoded burst = the msgblk_t fields the DSP fills (vdlm2.h:39-47). */
typedef struct {
	int32_t stream;
	int32_t chn;		/* msgblk_t.chn */
	int32_t Fr;		/* msgblk_t.Fr */
	int32_t nbrow;		/* msgblk_t.nbrow  (header value, d8psk.c:94) */
	int32_t nlbyte;		/* msgblk_t.nlbyte (header value, d8psk.c:95) */
	float df;		/* carrier estimate at sync, rad/symbol (channel_t.df) */
	float ppm;		/* msgblk_t.ppm, d8psk.c:302 */
	int64_t trig_dec;	/* 84 kS/s sample index of the sync trigger (stream time) */
	int64_t end_dec;	/* 84 kS/s sample index of the last symbol of the burst */
	int64_t trig_sample;	/* the same instants in input samples */
	int64_t end_sample;
	uint8_t data[VDL2GPU_MAXROWS][VDL2GPU_ROWLEN];	/* msgblk_t.data rows 0..7 */
} vdl2gpu_burst_t;

typedef struct {
	uint64_t samples_in;	/* per stream */
	uint64_t dec_samples;	/* 84 kS/s samples produced per channel */
	uint64_t sync_evals;	/* WSYNC evaluations, all channels */
	uint64_t triggers;	/* sync triggers */
	uint64_t header_rejects;	/* d8psk.c:97-107 */
	uint64_t bursts;	/* records handed out */
	uint64_t deferrals;	/* bursts that waited for a later push to complete */
	uint64_t candidates;	/* sync-trigger candidates found by the parallel scan (all timing hypotheses) */
	uint64_t serial_redos;	/* channel-pushes redone serially because the verify pass found an unlisted event */
	uint64_t serial_samples;	/* 84 kS/s samples handled by the serial machine (history-dependent stretches) */
	uint64_t overflowed;	/* burst records dropped: device ring full, or host queue never drained */
	uint64_t frames_dropped;	/* VDL2GPU_F_FRAMES: frames dropped (arena full, more than 12 frames in a burst, host queue never drained) */
	uint64_t repairs;	/* channel-pushes a repair round re-resolved because the verify pass found an unlisted event (cheap; serial_redos
				 * counts the ones no scheduled round could settle) */
} vdl2gpu_stats_t;

typedef struct {
	double channelise_ms;	/* sum over pushes of the channeliser kernels (HIP events on every stage_every-th push, scaled to all) */
	double demod_ms;	/* sum of the demodulator kernels (scan + cluster + resolve) */
	double scan_ms;		/* K2a sync scan */
	double cluster_ms;	/* K2b burst clusters */
	double resolve_ms;	/* K2c resolver + K2d gather */
	double other_ms;	/* compaction / bookkeeping kernels */
	double channelise_fast_ms;	/* the k1_fast launches alone (2 MS/s path): sum over fast_pushes launches */
	uint64_t fast_pushes;	/* number of fast-kernel launches behind channelise_fast_ms (the timed ones) */
	uint64_t pushes;
	uint64_t samples;	/* input samples per stream covered by the sums */
} vdl2gpu_timing_t;

typedef struct vdl2gpu vdl2gpu_t;

int vdl2gpu_abi_version(void);
int vdl2gpu_create(const vdl2gpu_config_t *cfg, vdl2gpu_t **out);
void vdl2gpu_destroy(vdl2gpu_t *h);

/* ---- Threads ------------------------------------------------------------------------------------------------
 * A handle may be used from several threads at once.  The reference has a producer thread -- the SDR library's callback,
 * in_callback() under rtlsdr_read_async() (rtl.c:274-295, 302) / rx_callback() (air.c:191-217) -- and consumer threads behind
 * it (main.c:225-231, vdlm2.c:84); a shim built on this library keeps that shape on ONE handle:
 *   - every call that takes a vdl2gpu_t* locks the handle for its duration; calls from different threads are serialised, none is
 *     lost, and each burst / frame record is handed out exactly once, whichever thread asks;
 *   - PRODUCER side: vdl2gpu_push(), or vdl2gpu_ring_acquire() / vdl2gpu_ring_commit().  One producer at a time: acquire and
 *     commit alternate, and pushes are decoded in the order the calls were made.  A producer call holds the lock while it
 *     enqueues (~0.1 ms) and, when three pushes are already in the pipeline, while it waits for the oldest of them;
 *   - CONSUMER side: vdl2gpu_poll_ready() / vdl2gpu_poll_frames_ready() never wait.  vdl2gpu_poll(), vdl2gpu_poll_frames() and
 *     vdl2gpu_pending() wait for everything that was pushed BEFORE the call -- with the lock released while they wait for the
 *     GPU, so a producer thread keeps committing blocks meanwhile (what it commits after the call began is not waited for);
 *   - per (stream, channel) the bursts come out in time order across calls, as decodeVdlm2() receives them (d8psk.c:201);
 *   - vdl2gpu_sync(), vdl2gpu_get_stats(), vdl2gpu_get_timing() and the vdl2gpu_debug_*() calls drain the pipeline under the
 *     lock: fine from any thread, but they hold the others up for that long;
 *   - vdl2gpu_last_error()'s string is valid until the next call on the handle from any thread;
 *   - vdl2gpu_destroy() must not run beside any other call on the handle (join the threads first);
 *   - different handles share nothing.
 * tests/ctests/thread_stress.c is this shape (one pthread committing ring slots, one collecting), held to the oracle ten times
 * over by tests/test_gpu_dropin.py. */

/* Feed `nsamples` samples of every stream.  Stream s starts at
 * (const char*)iq + s*stream_stride_bytes.  Asynchronous: returns once the
 * work is enqueued on the handle's HIP stream.  Replaces one Bar2/Bar1
 * hand-off of Cbuff (d8psk.c:360-383) for all channels at once, with any
 * block length instead of the fixed 32768.
 * Buffer lifetime: a VDL2GPU_MEM_HOST buffer may be reused as soon as the call returns (it has been copied).
 * A VDL2GPU_MEM_DEVICE buffer is read in place by the channeliser, asynchronously: it must stay valid and
 * unchanged until the second push after this one has been issued, or until vdl2gpu_sync() / vdl2gpu_poll()
 * returns -- whichever comes first (three pushes can be in the pipeline; the call that issues the second push after this one
 * waits until this one's channeliser has read the buffer).
 * Errors are sticky: after a call has returned VDL2GPU_EHIP the handle only accepts vdl2gpu_destroy(). */
int vdl2gpu_push(vdl2gpu_t *h, const void *iq, size_t nsamples, size_t stream_stride_bytes, int memkind);

/* Ingest ring: replaces the producer side of the hand-off -- in_callback()
 * converting a USB block into Cbuff between Bar1 and Bar2 (rtl.c:274-295), rx_callback() copying airspy
 * samples (air.c:191-217) -- by `nslots` slots of page-locked host memory that the producer fills in place
 * (e.g. rtlsdr_read_sync() straight into the slot; the cu8/cs16 conversion happens in the channeliser
 * kernel).  acquire() hands out the next slot -- stream s starts at slot + s * *stream_stride_bytes -- and
 * waits only if that slot's previous contents are still on their way to the GPU; commit(nsamples) enqueues
 * the copy and the whole decode behind it and returns at once, so the copy of one block runs beside the
 * kernels of the one before.  commit(0) drops the block (a short USB read, rtl.c:278-281).  One producer:
 * acquire and commit alternate.  Bursts come out through vdl2gpu_poll*() as with vdl2gpu_push(). */
int vdl2gpu_ring_init(vdl2gpu_t *h, size_t slot_samples, int nslots);
void *vdl2gpu_ring_acquire(vdl2gpu_t *h, size_t *stream_stride_bytes);
int vdl2gpu_ring_commit(vdl2gpu_t *h, size_t nsamples);
/* Wait until everything pushed so far has been demodulated. */
int vdl2gpu_sync(vdl2gpu_t *h);
/* Collect finished bursts (waits for everything pushed so far).  Bursts come out ordered by
 * (end_sample, stream, chn).  Returns the count (>=0) or a negative error.
 * The host keeps what has been fetched from the GPU and not yet handed out in two bounded queues (bursts;
 * with VDL2GPU_F_FRAMES also frames): a consumer that drains only one of them loses the OLDEST entries of the
 * other once it holds more than 4 x max_bursts unread ones (counted in vdl2gpu_stats_t.overflowed / frames
 * dropped).  Memory: the storage behind a queue is compacted as soon as its handed-out prefix is the larger part,
 * so it holds at most 2 x the unread records plus one push's worth -- <= 9 x max_bursts records of 2104 bytes
 * (1.2 GB with the default max_bursts = 65536 and a consumer that never polls; a few MB with one that does). */
int vdl2gpu_poll(vdl2gpu_t *h, vdl2gpu_burst_t *out, int max);
/* Same, but never waits: hands out only the bursts of pushes the GPU has already finished.  Lets
 * a caller keep the next pushes running while it consumes the earlier ones (three pushes can be in the pipeline;
 * a fourth first collects the oldest). */
int vdl2gpu_poll_ready(vdl2gpu_t *h, vdl2gpu_burst_t *out, int max);
/* Number of bursts a poll would currently return (implies vdl2gpu_sync). */
int vdl2gpu_pending(vdl2gpu_t *h);
/* Pushes the GPU has not finished yet (0..3); never waits.  For a producer that is not bound to real time and wants to
 * size its pushes by what the pipeline can take (a push of a million samples costs the calling thread the same 0.1 ms to enqueue
 * as one of 32768). */
int vdl2gpu_inflight(vdl2gpu_t *h);

int vdl2gpu_get_stats(vdl2gpu_t *h, vdl2gpu_stats_t *out);
int vdl2gpu_get_timing(vdl2gpu_t *h, vdl2gpu_timing_t *out, int reset);
/* Where the calling thread's time inside vdl2gpu_push() went (host bookkeeping: no pipeline drain), seconds summed since the last
 * reset: out8[0] waiting for the device buffer of the push before last; [1] enqueueing the channeliser; [2] collecting the output
 * ring this push reuses (mostly WAITING for the push three back to finish: the GPU is the slower side then); [3] enqueueing the rest
 * of the front stage; [4] moving uncollected records out of the slab's way; [5] enqueueing the back stage and the tail; [6] (part of
 * [2] and of the polling calls) waiting for a ring's completion event; [7] the pushes counted.  Replaces nothing of the reference. */
int vdl2gpu_get_host_profile(vdl2gpu_t *h, double *out8, int reset);
const char *vdl2gpu_last_error(vdl2gpu_t *h);
const char *vdl2gpu_strerror(int code);

/* Fill a reference msgblk_t (vdlm2.h:39-47, LP64 layout: prev@0 chn@8 Fr@12
 * tv@16 ppm@32 nbrow@36 nlbyte@40 data@44, sizeof 16624 -- the x86-64 / aarch64 Linux layout; the offsets
 * are compile-time constants of this library, so a host with another ABI must copy the fields itself)
 * from a burst record.  `msgblk` must point at sizeof(msgblk_t) zeroed bytes (calloc, as vdlm2.c:201);
 * msgblk_size is checked against 16624. */
#define VDL2GPU_MSGBLK_OFF_CHN 8
#define VDL2GPU_MSGBLK_OFF_FR 12
#define VDL2GPU_MSGBLK_OFF_TV 16
#define VDL2GPU_MSGBLK_OFF_PPM 32
#define VDL2GPU_MSGBLK_OFF_NBROW 36
#define VDL2GPU_MSGBLK_OFF_NLBYTE 40
#define VDL2GPU_MSGBLK_OFF_DATA 44
#define VDL2GPU_MSGBLK_SIZE 16624	/* oracle/ref_layout_check.c holds these against offsetof(msgblk_t, ..) of the reference's own vdlm2.h
					 * with _Static_assert: `make -C oracle ref` fails if they ever disagree */
int vdl2gpu_burst_to_msgblk(const vdl2gpu_burst_t *b, void *msgblk, size_t msgblk_size);

/* ---- block path: what the reference's blk_thread does with a msgblk_t ----
 * vdlm2.c:84-161 -- RS(255,249) of every row (rs.c), HDLC bit un-stuffing, flag hunt, check_frame()
 * (length >= 13, FCS-16 of crc.c) -- for a whole batch of bursts in one kernel.  One record per
 * frame the reference would pass to out(msgblk_t*, unsigned char *hdata, int l) (vdlm2.h:134):
 * data[0..len) is hdata, the other fields are the msgblk_t's.  Frames come out ordered by
 * (block, seq).  At the rates this library demodulates at, the reference's single blk_thread is
 * the bottleneck; a shim can call this instead of decodeVdlm2() and go to out(). */
#define VDL2GPU_MAXFRAME 2000
typedef struct {
	int32_t stream, chn, Fr;	/* of the burst the frame came in */
	int32_t nbrow, nlbyte;
	int32_t len;			/* l of out(): opening flag .. closing flag */
	int32_t block;			/* index of the burst in the batch */
	int32_t seq;			/* 0 for the burst's first frame, ... */
	float df, ppm;
	int64_t trig_dec, end_dec;
	uint8_t data[VDL2GPU_MAXFRAME];	/* hdata[0..len) */
} vdl2gpu_frame_t;
/* Decode `n` bursts (host array, e.g. straight from vdl2gpu_poll) into CRC-clean frames.
 * Returns the number of frames (<= max_frames; more are dropped and counted in *dropped if given)
 * or a negative error. */
int vdl2gpu_decode_blocks(vdl2gpu_t *h, const vdl2gpu_burst_t *blocks, int n,
			  vdl2gpu_frame_t *frames, int max_frames, int *dropped);
/* With VDL2GPU_F_FRAMES the same kernel runs on every push's burst records where they lie in device
 * memory, right behind the demodulator.  Collect the frames of everything pushed so far (waits like
 * vdl2gpu_poll; the bursts themselves stay available through vdl2gpu_poll).  Frames come out ordered
 * by (end_dec, stream, chn, seq); `block` is -1 here, and of data[] only [0, len) is written.  Returns the
 * count or a negative error. */
int vdl2gpu_poll_frames(vdl2gpu_t *h, vdl2gpu_frame_t *out, int max);
/* Same, but never waits (like vdl2gpu_poll_ready). */
int vdl2gpu_poll_frames_ready(vdl2gpu_t *h, vdl2gpu_frame_t *out, int max);

/* d8psk.c:39-52; the host path keeps calling it (out.c:429-432, outxid.c:122). */
unsigned int reversebits(const unsigned int bits, const int n);

/* ---- pure host helpers (usable without a GPU) ---- */
/* Local-oscillator table of one channel, d8psk.c:353-357 (libm sincosf of the
 * float-narrowed phase step).  Returns the table length SDRINRATE/25000. */
int vdl2gpu_lo_table(unsigned sdrinrate, int fo_hz, float *out_re_im, int max_complex);
/* Integrate-and-dump schedule of one push (d8psk.c:374-381 in closed form). */
int vdl2gpu_plan(uint64_t total_in, uint64_t n, unsigned sdrclk, unsigned lo_len,
		 int *c0, int *no0, int *nf0, int64_t *nout);
/* Frequency planning: the tuner centre the reference picks for a list of channel
 * frequencies, and the per-channel mixer offsets Fo it derives (thread_param_t.Fo).
 *   rtl: chooseFc() of rtl.c:123-160 -- the highest Fc (1 Hz steps, downwards from max + 50 kHz) that keeps every
 *        channel within SDRINRATE/2 - 50 kHz, none closer than 50 kHz, and no two adjacent (sorted) channels
 *        mirror images of each other; Fo = Fr - Fc (rtl.c:245-247).  Returns 0 and Fc = 0 when the channels span
 *        more than SDRINRATE - 100 kHz, like the reference.
 *   air: chooseFc() of air.c:47-70 -- the midpoint rounded to the 25 kHz grid, at 5 MS/s (Airspy R2) shifted by
 *        the R820T2 filter pair that just covers the span; Fo = Fr - (Fc + SDRINRATE/4) (air.c:182-184).  The two
 *        tuner register values the reference writes (R10 = 0xB0 | (15-j), R11 = 0xE0 | (15-i)) are returned as
 *        well (0 when none is written).
 * fr[] is NOT reordered (the reference sorts a scratch copy, rtl.c:218-241).  Parity: unpinned -- rtl.c / air.c
 * need the SDR vendor headers and cannot be compiled in the build container; tested against hand-derived
 * cases and an independent brute-force statement of the same rules. */
int vdl2gpu_choose_fc_rtl(const unsigned *fr, int nbch, unsigned sdrinrate, unsigned *fc, int *fo);
int vdl2gpu_choose_fc_air(const unsigned *fr, int nbch, unsigned sdrinrate, unsigned *fc, int *fo, int *r10, int *r11);

/* ---- diagnostics (P3 taps, tests only) ---- */
/* Decimated samples of the LAST push of (stream, channel index), interleaved
 * re,im; needs VDL2GPU_F_KEEP_DEC.  Returns the number of complex samples. */
int64_t vdl2gpu_debug_dec(vdl2gpu_t *h, int stream, int ch, float *out, int64_t max_complex);
/* Local-oscillator table of (stream, channel index): len complex values. */
int vdl2gpu_debug_lo(vdl2gpu_t *h, int stream, int ch, float *out, int max_complex);
/* Trigger candidates of the last push's sync scan, 6 x int32 each {nrel, r, p2err, perr, err, pfr bits}. */
int vdl2gpu_debug_cands(vdl2gpu_t *h, int stream, int ch, int *out, int max_cands);
/* ... and what follows each of them (one int2 per candidate: where the idle search resumes, relative to the push's planes; status |
 * sub-phase << 2 | descriptors << 4 | triggers << 8 | rejects << 16 | bursts << 24) */
int vdl2gpu_debug_clheads(vdl2gpu_t *h, int stream, int ch, int *out, int max_cands);
/* Verify pass of the last push: first unexpected detector hit per (stream, channel slot), and the
 * idle segments the resolver asked to have verified {lo, hi, r, pad}. */
int vdl2gpu_debug_fail(vdl2gpu_t *h, int *out, int n);
int vdl2gpu_debug_segs(vdl2gpu_t *h, int stream, int ch, int *out, int max_segs);
/* Development cycle counters of the demodulator kernels (meaning is internal). */
int vdl2gpu_debug_counters(vdl2gpu_t *h, unsigned long long *out, int n, int reset);
/* With VDL2GPU_F_DEBUG_HEADS: the descrambled header soft bits (what viterbi_add() is given, d8psk.c:81-83) of every
 * sync trigger ANY kernel of the last push handled -- the clusters of all timing classes, on the real chain or not,
 * and the serial stretches --, 34 x uint32 per entry: nstar (lo, hi), stream * 8 + channel slot, clk0, then the float
 * bits of p2err, perr, err, pfr and soft[25], one word of padding.  Unordered.  Returns the count (<= max_entries). */
int vdl2gpu_debug_heads(vdl2gpu_t *h, uint32_t *out, int max_entries);
/* Device build of the fixed-sequence atan2f, elementwise (host arrays). */
int vdl2gpu_debug_atan2f(vdl2gpu_t *h, const float *y, const float *x, float *out, size_t n);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
