/* The reference's threading shape on ONE handle (include/vdl2gpu.h, "Threads"): a producer thread that plays the SDR library's
 * callback -- acquire a slot of the ingest ring, fill it, commit (rtl.c:274-295 under rtlsdr_read_async, rtl.c:302) -- and a
 * consumer thread that collects msgblk_t records meanwhile, alternating the never-waiting and the waiting call, like the
 * reference's consumers behind decodeVdlm2().  Neither thread waits for the other except through the library.  Prints one line
 * per burst ("B chn nbrow nlbyte fnv(data)") in delivery order; the test compares with the oracle's over ten runs.
 *   thread_stress <raw file> <fmt: cu8|cs16> <rate> <nch> <Fo...> <block samples> <slots>
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "vdl2gpu.h"

static unsigned long long fnv(const unsigned char *p, size_t n)
{
	unsigned long long h = 1469598103934665603ull;
	for (size_t i = 0; i < n; ++i)
		h = (h ^ p[i]) * 1099511628211ull;
	return h;
}

static vdl2gpu_t *h;
static FILE *fp;
static size_t sb, blk;
static volatile int produced_all, failed;
static long long last_end[8] = { -1, -1, -1, -1, -1, -1, -1, -1 };

static void *producer(void *arg)
{
	(void)arg;
	for (;;) {
		size_t stride;
		void *slot = vdl2gpu_ring_acquire(h, &stride);
		if (!slot) {
			fprintf(stderr, "acquire: %s\n", vdl2gpu_last_error(h));
			failed = 1;
			break;
		}
		const size_t got = fread(slot, sb, blk, fp);
		const int rc = vdl2gpu_ring_commit(h, got);
		if (rc) {
			fprintf(stderr, "commit: %s\n", vdl2gpu_strerror(rc));
			failed = 1;
			break;
		}
		if (got < blk)
			break;
	}
	__sync_synchronize();
	produced_all = 1;
	return NULL;
}

static int take(int n, const vdl2gpu_burst_t *b)
{
	for (int i = 0; i < n; ++i) {
		if (b[i].chn < 0 || b[i].chn >= 8 || b[i].end_sample < last_end[b[i].chn]) {	/* per channel in time order, d8psk.c:201 */
			fprintf(stderr, "burst of channel %d out of order\n", b[i].chn);
			return -1;
		}
		last_end[b[i].chn] = b[i].end_sample;
		printf("B %d %d %d %016llx\n", b[i].chn, b[i].nbrow, b[i].nlbyte, fnv(&b[i].data[0][0], (size_t)b[i].nbrow * VDL2GPU_ROWLEN));
	}
	return 0;
}

static void *consumer(void *arg)
{
	static vdl2gpu_burst_t b[64];
	unsigned turn = 0;
	(void)arg;
	for (;;) {
		const int done = produced_all;	/* read BEFORE collecting: what is collected after this covers every block */
		__sync_synchronize();
		const int n = (done || (++turn & 7) == 0) ? vdl2gpu_poll(h, b, 64) : vdl2gpu_poll_ready(h, b, 64);
		if (n < 0 || take(n, b)) {
			fprintf(stderr, "poll: %s\n", n < 0 ? vdl2gpu_strerror(n) : "order");
			failed = 1;
			return NULL;
		}
		if (done && n == 0)
			return NULL;
	}
}

int main(int argc, char **argv)
{
	if (argc < 8)
		return 2;
	const int fmt = !strcmp(argv[2], "cu8") ? VDL2GPU_FMT_CU8 : VDL2GPU_FMT_CS16;
	sb = fmt == VDL2GPU_FMT_CU8 ? 2 : 4;
	const unsigned rate = (unsigned)atoi(argv[3]);
	const int nch = atoi(argv[4]);
	if (nch < 1 || nch > 8 || argc < 7 + nch)
		return 2;
	vdl2gpu_chan_t plan[8];
	for (int i = 0; i < nch; ++i) {
		plan[i].chn = i;
		plan[i].Fo = atoi(argv[5 + i]);
		plan[i].Fr = 136975000 + plan[i].Fo;
	}
	blk = (size_t)atol(argv[5 + nch]);
	const int slots = atoi(argv[6 + nch]);
	vdl2gpu_config_t cfg;
	memset(&cfg, 0, sizeof cfg);
	cfg.struct_size = sizeof cfg;
	cfg.sdrinrate = rate;
	cfg.fmt = fmt;
	cfg.nbch = nch;
	cfg.nstreams = 1;
	cfg.chan = plan;
	cfg.max_push = blk;
	int rc = vdl2gpu_create(&cfg, &h);
	if (rc) {
		fprintf(stderr, "create: %s\n", vdl2gpu_strerror(rc));
		return 1;
	}
	rc = vdl2gpu_ring_init(h, blk, slots);
	if (rc) {
		fprintf(stderr, "ring_init: %s\n", vdl2gpu_strerror(rc));
		return 1;
	}
	fp = fopen(argv[1], "rb");
	if (!fp)
		return 1;
	pthread_t tp, tc;
	pthread_create(&tc, NULL, consumer, NULL);
	pthread_create(&tp, NULL, producer, NULL);
	pthread_join(tp, NULL);
	pthread_join(tc, NULL);
	fclose(fp);
	vdl2gpu_destroy(h);
	return failed ? 1 : 0;
}
