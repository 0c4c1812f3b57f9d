/* A caller at C speed: a recording goes through the ingest ring in small blocks with the block path in the
 * pipeline, bursts and frames are collected with the never-waiting calls while blocks keep coming.  Prints
 * one line per burst ("B chn nbrow nlbyte fnv(data)") and per frame ("F chn len fnv(hdata)"), sorted by the
 * library's own delivery order; the test compares the lines with the oracle's and across repeated runs
 * (races between streams show up only when nothing slows the caller down).
 *   ring_stress <raw file> <fmt: cu8|cs16> <rate> <nch> <Fo...> <block samples> <slots>
 */
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

static vdl2gpu_burst_t b[256];
static vdl2gpu_frame_t f[256];

static int drain(vdl2gpu_t *h, int ready)
{
	int n;
	for (;;) {
		n = ready ? vdl2gpu_poll_ready(h, b, 256) : vdl2gpu_poll(h, b, 256);
		if (n < 0)
			return n;
		for (int i = 0; i < n; ++i)
			printf("B %d %d %d %016llx\n", b[i].chn, b[i].nbrow, b[i].nlbyte,
			       fnv(&b[i].data[0][0], (size_t)b[i].nbrow * VDL2GPU_ROWLEN));
		if (n < 256)
			break;
	}
	for (;;) {
		n = ready ? vdl2gpu_poll_frames_ready(h, f, 256) : vdl2gpu_poll_frames(h, f, 256);
		if (n < 0)
			return n;
		for (int i = 0; i < n; ++i)
			printf("F %d %d %016llx\n", f[i].chn, f[i].len, fnv(f[i].data, (size_t)f[i].len));
		if (n < 256)
			break;
	}
	return 0;
}

int main(int argc, char **argv)
{
	if (argc < 8)
		return 2;
	const char *path = argv[1];
	const int fmt = !strcmp(argv[2], "cu8") ? VDL2GPU_FMT_CU8 : VDL2GPU_FMT_CS16;
	const size_t sb = fmt == VDL2GPU_FMT_CU8 ? 2 : 4;
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
	const size_t blk = (size_t)atol(argv[5 + nch]);
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
	cfg.flags = VDL2GPU_F_FRAMES;
	vdl2gpu_t *h = NULL;
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
	FILE *fp = fopen(path, "rb");
	if (!fp)
		return 1;
	for (;;) {
		size_t stride;
		void *slot = vdl2gpu_ring_acquire(h, &stride);
		if (!slot) {
			fprintf(stderr, "acquire: %s\n", vdl2gpu_last_error(h));
			return 1;
		}
		const size_t got = fread(slot, sb, blk, fp);
		rc = vdl2gpu_ring_commit(h, got);
		if (rc) {
			fprintf(stderr, "commit: %s\n", vdl2gpu_strerror(rc));
			return 1;
		}
		if (drain(h, 1))
			return 1;
		if (got < blk)
			break;
	}
	fclose(fp);
	if (drain(h, 0))
		return 1;
	vdl2gpu_destroy(h);
	return 0;
}
