/*
 * ref_harness.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * File-replay driver that plays the SDR-producer role of the reference
 * (rtl.c:274-295 in_callback / air.c:191-217 rx_callback) against the
 * reference's OWN, UNMODIFIED translation units d8psk.c, viterbi.c, vdlm2.c,
 * crc.c, rs.c, which are compiled from where they lie under /root/reference
 * by oracle/Makefile into oracle/_ref/.  Nothing from the reference is copied
 * here: this file only supplies the globals vdlm2.h:30-31,84-96 declares
 * extern, an out() sink (vdlm2.h:134), and link-time taps.
 *
 * Canonical oracle semantics:
 *   - ONE channel per process (avoids the shared static Viterbi tables race,
 *     viterbi.c:25-27); the channel always runs as chn 0 because only
 *     channel 0's initVdlm2() spawns blk_thread (vdlm2.c:172-177).
 *   - fresh thread => rcv_thread's stack channel_t is all-zero.
 *   - gettimeofday interposed (timestamps are not part of parity).
 *   - cu8 conversion WITHOUT the rtl.c:291 store-index off-by-one unless
 *     quirk=1 is requested.
 *
 * Taps (link-time, reference sources untouched):
 *   --wrap=decodeVdlm2  every msgblk_t handed to the host path (d8psk.c:201)
 *   --wrap=viterbi_add  every descrambled header soft bit (d8psk.c:83)
 *   --wrap=free         lets us know when blk_thread finished a block
 *   atan2f / roundf     defined here, forward to libm via dlsym(RTLD_NEXT)
 *   out()               every CRC-clean frame (vdlm2.c:61)
 *
 * usage: ref_xxx <iqfile> <fmt:cu8|cs16|cf32|f32> <SDRINRATE> <Fo> <Fr> <outfile> [quirk] [tapfile]
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <unistd.h>
#include <dlfcn.h>
#include <math.h>
#include <pthread.h>
#include <complex.h>
#include <sys/time.h>
#include <time.h>
#include "vdlm2.h"

unsigned int SDRINRATE = 2000000;
unsigned int SDRCLK = 500;
unsigned int Fc = 0;
int ppm = 0;
int verbose = 0;
int nbch = 1;		/* main.c:59 (only the GPU drop-in build reads it) */
int grndmess = 1, emptymess = 1, undecmess = 1;
FILE *logfd;
pthread_barrier_t Bar1, Bar2;
#ifdef WITH_RTL
complex float Cbuff[RTLINBUFSZ / 2];
#else
float Cbuff[RTLINBUFSZ / 2];
#endif

static FILE *outfd;
static FILE *tapfd;
static pthread_mutex_t outmtx = PTHREAD_MUTEX_INITIALIZER;
static volatile long n_enq, n_freed;
static volatile uint64_t sample_clock;	/* input samples handed over so far */

/* ---- taps ------------------------------------------------------------- */
static void tap(uint32_t type, float a, float b, float c)
{
	struct { uint32_t t; float a, b, c; } r = { type, a, b, c };
	if (tapfd)
		fwrite(&r, sizeof r, 1, tapfd);
}

float atan2f(float y, float x)
{
	static float (*real) (float, float);
	if (!real)
		real = (float (*)(float, float))dlsym(RTLD_NEXT, "atan2f");
	float r = real(y, x);
	tap(1, y, x, r);
	return r;
}

float cargf(complex float z)
{
	return atan2f(cimagf(z), crealf(z));
}

float roundf(float x)
{
	static float (*real) (float);
	if (!real)
		real = (float (*)(float))dlsym(RTLD_NEXT, "roundf");
	float r = real(x);
	tap(2, x, r, 0);
	return r;
}

int gettimeofday(struct timeval *tv, void *tz)
{
	(void)tz;
	/* deterministic "time": block-granular input sample clock */
	tv->tv_sec = (time_t) (sample_clock / 1000000);
	tv->tv_usec = (suseconds_t) (sample_clock % 1000000);
	return 0;
}

#ifndef VDL2GPU_DROPIN
extern void __real_viterbi_add(float V, int n);
void __wrap_viterbi_add(float V, int n)
{
	tap(3, V, (float)n, 0);
	__real_viterbi_add(V, n);
}
#endif

#ifndef VDL2GPU_FRAMES
extern void __real_decodeVdlm2(channel_t * ch);
void __wrap_decodeVdlm2(channel_t * ch)
{
	int r, i;
	msgblk_t *b = ch->blk;
	pthread_mutex_lock(&outmtx);
	/* tv: what d8psk.c:295 stamped at the sync trigger -- gettimeofday() is the sample clock here (below), so the
	 * stamp says which hand-off block the trigger was seen in, deterministically */
	fprintf(outfd, "B %d %d %.9g %08x t%ld.%06ld ", b->nbrow, b->nlbyte, b->ppm,
		*(uint32_t *) & ch->df, (long)b->tv.tv_sec, (long)b->tv.tv_usec);
	if (nbch > 1)
		fprintf(outfd, "c%d ", b->chn);
	for (r = 0; r < 8; r++)
		for (i = 0; i < 255; i++)
			fprintf(outfd, "%02x", b->data[r][i]);
	fprintf(outfd, "\n");
	n_enq++;
	pthread_mutex_unlock(&outmtx);
	tap(4, (float)b->nbrow, (float)b->nlbyte, ch->df);
	__real_decodeVdlm2(ch);
}

extern void __real_free(void *p);
void __wrap_free(void *p)
{
	__real_free(p);
	__sync_fetch_and_add(&n_freed, 1);
}
#endif

void out(msgblk_t * blk, unsigned char *hdata, int l)
{
	int i;
	pthread_mutex_lock(&outmtx);
	fprintf(outfd, "F %d %d %d ", blk->nbrow, blk->nlbyte, l);
	if (nbch > 1)
		fprintf(outfd, "c%d ", blk->chn);
	for (i = 0; i < l; i++)
		fprintf(outfd, "%02x", hdata[i]);
	fprintf(outfd, "\n");
	pthread_mutex_unlock(&outmtx);
}

/* ---- producer --------------------------------------------------------- */
int main(int argc, char **argv)
{
	if (argc < 7) {
		fprintf(stderr,
			"usage: %s iqfile fmt rate Fo Fr outfile [quirk] [tapfile]\n",
			argv[0]);
		return 2;
	}
	const char *fmt = argv[2];
	SDRINRATE = (unsigned)atoi(argv[3]);
	SDRCLK = SDRINRATE / 4000;	/* air.c:138; 500 at 2 MS/s as rtl.c:37 */
	int quirk = argc > 7 ? atoi(argv[7]) : 0;
	FILE *f = fopen(argv[1], "rb");
	if (!f) {
		perror(argv[1]);
		return 1;
	}
	outfd = fopen(argv[6], "w");
	if (argc > 8 && argv[8][0])
		tapfd = fopen(argv[8], "wb");
	logfd = stderr;

	/* Fo and Fr may be comma-separated lists: several channels in one process is only used
	 * by the GPU drop-in build (the reference itself is run one channel per process) */
	static thread_param_t tp[MAXNBCHANNELS];
	{
		char *fo = strdup(argv[4]), *fr = strdup(argv[5]), *s1, *s2;
		char *a = strtok_r(fo, ",", &s1), *b = strtok_r(fr, ",", &s2);
		nbch = 0;
		while (a && b && nbch < MAXNBCHANNELS) {
			tp[nbch].chn = nbch;
			tp[nbch].Fo = atoi(a);
			tp[nbch].Fr = atoi(b);
			nbch++;
			a = strtok_r(NULL, ",", &s1);
			b = strtok_r(NULL, ",", &s2);
		}
	}
	Fc = (unsigned)(tp[0].Fr - tp[0].Fo);

	pthread_barrier_init(&Bar1, NULL, nbch + 1);
	pthread_barrier_init(&Bar2, NULL, nbch + 1);
	pthread_t th[MAXNBCHANNELS];
	for (int n = 0; n < nbch; n++)
		pthread_create(&th[n], NULL, rcv_thread, &tp[n]);

	const int NB = RTLINBUFSZ / 2;
	size_t ssz;
#ifdef WITH_RTL
	if (!strcmp(fmt, "cu8"))
		ssz = 2;
	else if (!strcmp(fmt, "cs16"))
		ssz = 4;
	else if (!strcmp(fmt, "cf32"))
		ssz = 8;
	else {
		fprintf(stderr, "bad fmt for complex build\n");
		return 2;
	}
#else
	if (!strcmp(fmt, "f32"))
		ssz = 4;
	else {
		fprintf(stderr, "bad fmt for real build\n");
		return 2;
	}
#endif
	unsigned char *raw = malloc(ssz * NB);
	struct timespec t_replay0;
	clock_gettime(CLOCK_MONOTONIC, &t_replay0);
	for (;;) {
		size_t got = fread(raw, ssz, NB, f);
		if (got != (size_t) NB)
			break;	/* only whole blocks, like rtl.c:278-281 */
		pthread_barrier_wait(&Bar1);
		int i;
#ifdef WITH_RTL
		if (ssz == 2) {
			if (quirk) {	/* rtl.c:285-292 as written: sample k lands in slot k+1 */
				Cbuff[0] = 0;
				for (i = 0; i < NB - 1; i++)
					Cbuff[i + 1] =
					    ((float)raw[2 * i] - (float)127.37) +
					    ((float)raw[2 * i + 1] -
					     (float)127.37) * I;
			} else
				for (i = 0; i < NB; i++)
					Cbuff[i] =
					    ((float)raw[2 * i] - (float)127.37) +
					    ((float)raw[2 * i + 1] -
					     (float)127.37) * I;
		} else if (ssz == 4) {
			const int16_t *s = (const int16_t *)raw;
			for (i = 0; i < NB; i++)
				Cbuff[i] = (float)s[2 * i] + (float)s[2 * i + 1] * I;
		} else {
			const float *s = (const float *)raw;
			for (i = 0; i < NB; i++)
				Cbuff[i] = s[2 * i] + s[2 * i + 1] * I;
		}
#else
		memcpy(Cbuff, raw, sizeof(float) * NB);
#endif
		sample_clock += NB;
		pthread_barrier_wait(&Bar2);
	}
	pthread_barrier_wait(&Bar1);	/* consumer finished the last block */
#ifdef VDL2GPU_DROPIN
	{
		extern void vdl2gpu_rcv_flush(void);	/* dropin/vdl2gpu_rcv.c: the shutdown path's call, next to stopVdlm2() */
		struct timespec t1;
		vdl2gpu_rcv_flush();
		clock_gettime(CLOCK_MONOTONIC, &t1);
		/* file replay through the drop-in, first hand-off to last burst delivered (bench.py's dropin_replay leg reads this) */
		fprintf(stderr, "replay %llu samples %.6f s\n", (unsigned long long)sample_clock,
			(double)(t1.tv_sec - t_replay0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t_replay0.tv_nsec));
	}
#endif
	/* drain blk_thread: every enqueued block is freed at vdlm2.c:157 */
	int spins = 0;
	while (n_freed < n_enq && spins++ < 20000)
		usleep(500);
	pthread_mutex_lock(&outmtx);
	fclose(outfd);
	if (tapfd)
		fclose(tapfd);
	_exit(0);
}
