/*
 * vdl2gpu_rcv.c -- drop-in replacement for the reference's d8psk.c + viterbi.c.
 *
 * Build the reference with this file instead of those two (everything else unchanged) and
 * link libvdl2gpu.so:
 *
 *     cc -DWITH_RTL ... main.c rtl.c vdlm2.c crc.c rs.c out*.c label.c cJSON.c \
 *        dropin/vdl2gpu_rcv.c -Iinclude -L. -lvdl2gpu -lpthread -lm
 *
 * It keeps the three entry points the rest of the program uses (vdlm2.h:113-128):
 *     void *rcv_thread(void *arg)        main.c:230 still spawns one per channel
 *     int   initD8psk(channel_t *ch)
 *     unsigned reversebits(unsigned,int) comes from libvdl2gpu.so (out.c:429 keeps working)
 * and the reference's hand-off protocol: producer `Bar1 -> fill Cbuff -> Bar2`
 * (rtl.c:283-294, air.c:203-212), consumers `Bar2 -> read Cbuff -> Bar1` (d8psk.c:360-383),
 * barrier count nbch+1 (main.c:225-226).  The thread of channel 0 feeds the block to the GPU
 * for ALL channels; the other rcv_threads only keep the barrier count.  Every finished burst
 * is copied into that channel's `ch->blk` and handed to the unchanged decodeVdlm2()
 * (vdlm2.c:189), so RS / HDLC / CRC / ACARS / output run exactly as before.
 *
 * With -DVDL2GPU_FRAMES the block path runs on the GPU as well (vdl2gpu_decode_blocks: RS, HDLC
 * un-stuffing, FCS for a whole batch of bursts in one kernel) and the shim goes straight to
 * out(msgblk_t*, unsigned char *hdata, int l) (vdlm2.h:134, called by check_frame, vdlm2.c:60):
 * then vdlm2.c, rs.c and crc.c drop out of the build too.  At the rates the GPU front end
 * demodulates at, the reference's single blk_thread cannot keep up.
 *
 * This file contains no DSP.  It is compiled against the reference's own vdlm2.h.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <time.h>
#include <complex.h>
#include "vdlm2.h"
#include "vdl2gpu.h"

extern int nbch;		/* main.c:59 */

static channel_t g_ch[MAXNBCHANNELS];

/* d8psk.c:295 stamps a burst with gettimeofday() when its sync word triggers, i.e. while the hand-off block that holds
 * the trigger is being processed; out.c prints that stamp with every message.  Here bursts come back one or more blocks
 * later, so the wall time of every push is kept (first input sample, timeval) and a burst gets the stamp of the push
 * its trigger sample (vdl2gpu_burst_t.trig_sample) arrived in.  The longest burst spans 33 blocks at 2 MS/s. */
#define NSTAMP 1024	/* >= BATCH_BLOCKS hand-offs per push x (three pushes in the pipeline + the slot being filled) + the 33 blocks of the longest
			 * burst + slack: a burst collected that many hand-offs after its trigger block still finds that block's stamp */
static struct { unsigned long long first; struct timeval tv; } g_stamp[NSTAMP];
static unsigned long long g_npush, g_nsamples;

static void stamp_push(unsigned long long nsamples)
{
	gettimeofday(&g_stamp[g_npush % NSTAMP].tv, NULL);
	g_stamp[g_npush % NSTAMP].first = g_nsamples;
	g_npush++;
	g_nsamples += nsamples;
}

static struct timeval stamp_of(long long sample)
{
	unsigned long long k = g_npush, oldest = g_npush > NSTAMP ? g_npush - NSTAMP : 0;
	while (k > oldest + 1 && g_stamp[(k - 1) % NSTAMP].first > (unsigned long long)(sample < 0 ? 0 : sample))
		k--;
	return g_stamp[(k - 1) % NSTAMP].tv;	/* (a trigger older than the ring gets the oldest stamp kept) */
}
static volatile int g_ready;	/* channels initialised so far (channel 0 must be first, vdlm2.c:172) */

/* Hand-offs of Cbuff are collected in a slot of the library's ingest ring (page-locked memory) and the slot is committed when
 * it is full or the source is live (the previous hand-off came a millisecond or more ago): a live source is committed block by
 * block with no added latency, a file replay in pushes of BATCH_BLOCKS blocks -- enqueueing a push costs the feeding thread
 * 0.1 ms whatever its size, a hand-off is 32768 samples, and the pipeline behind the ring wants pushes of a million to run at
 * its rate.  Bursts are collected with the never-waiting call after every hand-off [round 3: the waiting
 * vdl2gpu_poll(), one pipeline drain per 32768 samples: the CPU reference's own 133 MS/s]; vdl2gpu_rcv_flush() -- for the
 * program's shutdown path, next to stopVdlm2() (main.c:106-110) -- commits what is collected and waits for the rest. */
#define BATCH_BLOCKS 64
_Static_assert(NSTAMP >= BATCH_BLOCKS * 5 + 64, "the stamp ring must reach back over the pushes in flight");
#define LIVE_GAP_NS 1000000	/* hand-offs at least this far apart are a live source: committed at once (a block is 16.4 ms of air time
				 * at 2 MS/s, 3.3 ms at 10 MS/s); closer together the source is a replay and the slot fills first */
static vdl2gpu_t *g_h;
static long long g_last_ns;	/* when the previous hand-off came */
static int g_null;		/* VDL2GPU_RCV_NULL=1: keep the barrier protocol, do nothing else (measures what the protocol alone allows) */
static char *g_slot;		/* the slot being filled, or NULL */
static size_t g_fill;		/* samples in it */
static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;	/* deliver() and the slot: the feeding thread against vdl2gpu_rcv_flush() */
#ifdef WITH_RTL
#define SAMPLE_BYTES sizeof(complex float)
#else
#define SAMPLE_BYTES sizeof(float)
#endif

int initD8psk(channel_t *ch)
{
	(void)ch;
	return 0;		/* all detector state lives on the GPU */
}

#ifdef VDL2GPU_FRAMES
static void deliver(vdl2gpu_t *h, int wait)
{
	static vdl2gpu_burst_t b[64];
	static vdl2gpu_frame_t f[128];
	static msgblk_t blk;	/* what out() reads of it: chn, Fr, tv, ppm, nbrow, nlbyte */
	int n, nf, i, dropped = 0;
	while ((n = wait ? vdl2gpu_poll(h, b, 64) : vdl2gpu_poll_ready(h, b, 64)) > 0) {
		nf = vdl2gpu_decode_blocks(h, b, n, f, 128, &dropped);
		if (dropped)	/* more frames than the buffer holds (or than 12 in one burst): never silent */
			fprintf(stderr, "vdl2gpu_decode_blocks: %d frame(s) dropped\n", dropped);
		for (i = 0; i < nf; i++) {
			memset(&blk, 0, sizeof blk);
			vdl2gpu_burst_to_msgblk(&b[f[i].block], &blk, sizeof blk);
			blk.tv = stamp_of(b[f[i].block].trig_sample);	/* d8psk.c:295 */
			out(&blk, f[i].data, f[i].len);
		}
		if (nf < 0)
			fprintf(stderr, "vdl2gpu_decode_blocks: %s\n", vdl2gpu_strerror(nf));
		if (n < 64)
			break;
	}
	if (n < 0)
		fprintf(stderr, "vdl2gpu_poll: %s\n", vdl2gpu_strerror(n));
}
#else
static void deliver(vdl2gpu_t *h, int wait)
{
	static vdl2gpu_burst_t b[64];
	int n, i;
	while ((n = wait ? vdl2gpu_poll(h, b, 64) : vdl2gpu_poll_ready(h, b, 64)) > 0) {
		for (i = 0; i < n; i++) {
			channel_t *ch = &g_ch[b[i].chn];
			vdl2gpu_burst_to_msgblk(&b[i], ch->blk, sizeof(msgblk_t));
			ch->df = b[i].df;			/* channel_t.df as d8psk.c:301 leaves it */
			ch->blk->tv = stamp_of(b[i].trig_sample);	/* d8psk.c:295: the time the block holding the sync trigger was handed over */
			decodeVdlm2(ch);			/* takes ch->blk, installs a fresh zeroed one */
		}
		if (n < 64)
			break;
	}
	if (n < 0)
		fprintf(stderr, "vdl2gpu_poll: %s\n", vdl2gpu_strerror(n));
}

#endif

static void commit_slot(void)
{
	if (g_slot && g_fill) {
		const int rc = vdl2gpu_ring_commit(g_h, g_fill);
		if (rc)
			fprintf(stderr, "vdl2gpu_ring_commit: %s (%s)\n", vdl2gpu_strerror(rc), vdl2gpu_last_error(g_h));
		g_slot = NULL;
		g_fill = 0;
	}
}

/* End of stream / shutdown: decode what has been handed over so far and pass every burst on.  Safe from any thread. */
void vdl2gpu_rcv_flush(void)
{
	pthread_mutex_lock(&g_mu);
	if (g_h) {
		commit_slot();
		deliver(g_h, 1);
	}
	pthread_mutex_unlock(&g_mu);
}

static void flush_at_exit(void)
{
	int flushed = 0;
	if (pthread_mutex_trylock(&g_mu) == 0) {	/* (a thread caught in the middle of a hand-off keeps the lock: then nothing can be flushed safely) */
		if (g_h) {
			commit_slot();
			deliver(g_h, 1);
			flushed = 1;
		}
		pthread_mutex_unlock(&g_mu);
	}
#ifndef VDL2GPU_FRAMES
	/* decodeVdlm2() only QUEUES a burst for the reference's block thread (vdlm2.c:189-201); main.c's own stopVdlm2() has
	 * returned long before an atexit handler runs, so what was queued just now would die with the process: wait for the
	 * queue once more (stopVdlm2, vdlm2.c:182-187: it polls once a second, up to five times). */
	if (flushed)
		stopVdlm2();
#else
	(void)flushed;
#endif
}

void *rcv_thread(void *arg)
{
	thread_param_t *param = (thread_param_t *) arg;
	channel_t *ch = &g_ch[param->chn];
	vdl2gpu_t *h = NULL;

	ch->chn = param->chn;
	ch->Fr = param->Fr;
	while (g_ready != param->chn)	/* initVdlm2 of channel 0 creates the block thread */
		sched_yield();
	initD8psk(ch);
#ifndef VDL2GPU_FRAMES
	initVdlm2(ch);		/* block queue + blk_thread of the unchanged host path */
#endif
	__sync_fetch_and_add(&g_ready, 1);

	if (param->chn == 0) {
		/* main.c passes &tparam[n]; the array is contiguous, so channel 0 sees the whole plan */
		vdl2gpu_chan_t plan[MAXNBCHANNELS];
		vdl2gpu_config_t cfg;
		int n, rc;
		while (g_ready != nbch)
			sched_yield();
		for (n = 0; n < nbch; n++) {
			plan[n].chn = param[n].chn;
			plan[n].Fr = param[n].Fr;
			plan[n].Fo = param[n].Fo;
		}
		memset(&cfg, 0, sizeof cfg);
		cfg.struct_size = sizeof cfg;
		cfg.sdrinrate = SDRINRATE;
		cfg.sdrclk = SDRCLK;
#ifdef WITH_RTL
		cfg.fmt = VDL2GPU_FMT_CF32;	/* Cbuff is complex float, rtl.c:273 */
#else
		cfg.fmt = VDL2GPU_FMT_F32R;	/* Cbuff is real float, air.c:190 */
#endif
		cfg.nbch = nbch;
		cfg.nstreams = 1;
		cfg.chan = plan;
		cfg.max_push = (size_t)BATCH_BLOCKS * (RTLINBUFSZ / 2);
		rc = vdl2gpu_create(&cfg, &h);
		if (rc) {
			fprintf(stderr, "vdl2gpu_create: %s\n", vdl2gpu_strerror(rc));
			exit(1);
		}
		rc = vdl2gpu_ring_init(h, cfg.max_push, 4);
		if (rc) {
			fprintf(stderr, "vdl2gpu_ring_init: %s\n", vdl2gpu_strerror(rc));
			exit(1);
		}
		g_null = getenv("VDL2GPU_RCV_NULL") != NULL;
		pthread_mutex_lock(&g_mu);
		g_h = h;
		pthread_mutex_unlock(&g_mu);
		/* the reference's main.c never calls vdl2gpu_rcv_flush(): what a fast source left in the slot being filled and what
		 * the pipeline still holds is decoded and handed over when the program exits (a maintainer who adds the call next
		 * to stopVdlm2() gets it earlier: INTEGRATION.md) */
		atexit(flush_at_exit);
	}

	pthread_barrier_wait(&Bar1);
	for (;;) {
		pthread_barrier_wait(&Bar2);
		int live = 0;
		if (h && !g_null) {
			/* Cbuff is copied out before the producer is let go; demodulation happens asynchronously */
			struct timespec ts;
			long long now;
			clock_gettime(CLOCK_MONOTONIC, &ts);
			now = (long long)ts.tv_sec * 1000000000ll + ts.tv_nsec;
			live = g_last_ns == 0 || now - g_last_ns >= LIVE_GAP_NS;
			g_last_ns = now;
			pthread_mutex_lock(&g_mu);
			stamp_push(RTLINBUFSZ / 2);
			if (!g_slot) {
				size_t stride;
				g_slot = vdl2gpu_ring_acquire(h, &stride);
				g_fill = 0;
				if (!g_slot)
					fprintf(stderr, "vdl2gpu_ring_acquire: %s\n", vdl2gpu_last_error(h));
			}
			if (g_slot) {
				memcpy(g_slot + g_fill * SAMPLE_BYTES, (const void *)Cbuff, (RTLINBUFSZ / 2) * SAMPLE_BYTES);
				g_fill += RTLINBUFSZ / 2;
			} else {
				/* no slot: the block still goes in, by the copying call (stamp_push() has counted its samples: dropping it
				 * would shift every later burst's time stamp) */
				const int rc = vdl2gpu_push(h, (const void *)Cbuff, RTLINBUFSZ / 2, 0, VDL2GPU_MEM_HOST);
				if (rc)
					fprintf(stderr, "vdl2gpu_push: %s (%s)\n", vdl2gpu_strerror(rc), vdl2gpu_last_error(h));
			}
			pthread_mutex_unlock(&g_mu);
		}
		pthread_barrier_wait(&Bar1);	/* producer may refill Cbuff */
		if (h && !g_null) {
			pthread_mutex_lock(&g_mu);
			if (live || g_fill >= (size_t)BATCH_BLOCKS * (RTLINBUFSZ / 2))
				commit_slot();
			deliver(h, 0);
			pthread_mutex_unlock(&g_mu);
		}
	}
	return NULL;
}
