/*
 * vdl2_oracle.c -- TEST INFRASTRUCTURE ONLY (see vdl2_oracle.h).
 *
 * Scalar CPU restatement of the reference's hot path, written from the
 * behaviour documented in SURVEY.md Appendix A, one stage per function, with an
 * explicit zero-initialised state record instead of the reference's thread
 * stack.  Every float operation is kept in the reference's order and width
 * (float vs double), and the library is compiled -O2 -ffp-contract=off, so
 * that results are bit-identical to the reference built -O2 on x86-64.
 *
 *   stage                         reference
 *   ----------------------------  ------------------------------------------
 *   ingest_*()                    rtl.c:285-292 (cu8), air.c:206-208 (real f32)
 *   lo_table()                    d8psk.c:353-357
 *   mix_sample()                  d8psk.c:366-381
 *   demod_sample()                d8psk.c:232-333
 *   fir_phase()                   d8psk.c:219-230
 *   sync_metric()                 d8psk.c:257-289
 *   slice_symbol()                d8psk.c:211-217, 321-331
 *   take_bit() + header/payload   d8psk.c:54-209
 *   header_viterbi_*()            viterbi.c:37-96 (state made per-channel)
 *   vo_block_frames()             vdlm2.c:64-161, crc.c
 *   vo_rs_decode()                rs.c:81-291
 */
#define _GNU_SOURCE
#include <complex.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "vdl2_oracle.h"

/* ---- constant data (d8psk.h:20-249), as bit patterns ------------------- */
typedef union { uint32_t u; float f; } f32bits;
#define VDL2_TABLE_BEGIN(name, n) static const f32bits T_##name[n] = {
#define VDL2_F32(x) {x},
#define VDL2_TABLE_END };
#include "../vdlm2dec_amd/csrc/vdl2_tables.inc"
#undef VDL2_TABLE_BEGIN
#undef VDL2_F32
#undef VDL2_TABLE_END

#define NTAP 65			/* MFLTLEN vdlm2.h:37 */
#define NRING 17		/* MBUFLEN vdlm2.h:38 */
#define NSYNC 17		/* NBPH vdlm2.h:54 */
#define NPH 68			/* NBPH*D8DWN vdlm2.h:54-55 */

enum { ST_WSYNC, ST_HEAD, ST_DATA, ST_FEC };

/* (25,20) header code parity-check columns: data table viterbi.c:29-35 */
static const int HCOL[25] = {
	6, 7, 9, 10, 11, 12, 14, 15, 17, 19, 21, 22, 24, 25, 26, 27, 28, 29, 30, 31,
	16, 8, 4, 2, 1
};

struct vo_chan {
	unsigned rate, sdrclk;
	int lo_len, fo, fr;
	float *lo_re, *lo_im;
	/* mixer / integrate-and-dump (d8psk.c:343-347) */
	int mclk, nf, no;
	float acc_re, acc_im;
	/* demodulator (vdlm2.h:56-79) */
	float in_re[NRING], in_im[NRING];
	int ink;
	float ph[NPH];
	int phidx;
	float df, p2err, perr, pfr, p1;
	int clk;
	unsigned scrambler, nbits, nbyte, nrow, nbrow, nlbyte;
	unsigned char bits;
	int state;
	/* header Viterbi (viterbi.c:25-27) */
	double pb[26][32];
	int bk[26][32], bv[26][32];
	/* outputs */
	vo_block cur;
	vo_block *blocks;
	size_t nblocks, capblocks;
	vo_trigger *trigs;
	size_t ntrigs, captrigs;
	int64_t ndec;
	/* taps */
	int tap_dec, tap_ph;
	float *dec;
	size_t capdec;
	float *phs;
	size_t nphs, capphs;
};

/* ------------------------------------------------------------------ helpers */
unsigned vo_reversebits(unsigned bits, int n)
{
	unsigned r = 0;
	for (int i = 0; i < n; i++)
		r |= ((bits >> i) & 1u) << (n - 1 - i);
	return r;
}

void vo_pn_bits(uint8_t *out, size_t n)
{
	unsigned s = 0x4D4B;
	for (size_t i = 0; i < n; i++) {
		unsigned b = (s ^ (s >> 14)) & 1u;
		s = (s << 1) | b;
		out[i] = (uint8_t) b;
	}
}

float vo_atan2f(float y, float x)
{
	return atan2f(y, x);
}

static void *grow(void *p, size_t *cap, size_t need, size_t elt)
{
	if (need <= *cap)
		return p;
	size_t nc = *cap ? *cap * 2 : 64;
	while (nc < need)
		nc *= 2;
	p = realloc(p, nc * elt);
	if (!p)
		abort();
	*cap = nc;
	return p;
}

/* --------------------------------------------------------- create / destroy */
static void lo_table(vo_chan *c)
{
	/* d8psk.c:354: float <- (float/float) * 2.0 * M_PI evaluated in double */
	float w = (float)((double)((float)c->fo / (float)c->rate) * 2.0 * M_PI);
	for (int n = 0; n < c->lo_len; n++) {
		float y = (float)(-n) * w;
		float complex z = cexpf(CMPLXF(0.0f, y));	/* d8psk.c:356 */
		c->lo_re[n] = crealf(z);
		c->lo_im[n] = cimagf(z);
	}
}

vo_chan *vo_create(unsigned sdrinrate, int fo_hz, int fr_hz)
{
	vo_chan *c = calloc(1, sizeof *c);	/* all-zero = canonical start state */
	c->rate = sdrinrate;
	c->sdrclk = sdrinrate / 4000;	/* rtl.c:37, air.c:138 */
	c->lo_len = (int)(sdrinrate / 25000);	/* STEPRATE vdlm2.h:33 */
	c->fo = fo_hz;
	c->fr = fr_hz;
	c->lo_re = malloc(sizeof(float) * c->lo_len);
	c->lo_im = malloc(sizeof(float) * c->lo_len);
	lo_table(c);
	/* initD8psk d8psk.c:28-37 */
	c->ink = 0;
	c->phidx = 0;
	c->df = 0;
	c->perr = 100;
	c->p1 = 0;
	c->state = ST_WSYNC;	/* initVdlm2 vdlm2.c:167 */
	return c;
}

void vo_destroy(vo_chan *c)
{
	if (!c)
		return;
	free(c->lo_re);
	free(c->lo_im);
	free(c->blocks);
	free(c->trigs);
	free(c->dec);
	free(c->phs);
	free(c);
}

void vo_enable_taps(vo_chan *c, int dec_samples, int phases)
{
	c->tap_dec = dec_samples;
	c->tap_ph = phases;
}

void vo_lo_table(const vo_chan *c, float *out, int *len)
{
	for (int n = 0; n < c->lo_len; n++) {
		out[2 * n] = c->lo_re[n];
		out[2 * n + 1] = c->lo_im[n];
	}
	*len = c->lo_len;
}

size_t vo_num_blocks(const vo_chan *c) { return c->nblocks; }
const vo_block *vo_blocks(const vo_chan *c) { return c->blocks; }
size_t vo_num_triggers(const vo_chan *c) { return c->ntrigs; }
const vo_trigger *vo_triggers(const vo_chan *c) { return c->trigs; }
size_t vo_num_dec(const vo_chan *c) { return (size_t) c->ndec; }
const float *vo_dec_tap(const vo_chan *c) { return c->dec; }
size_t vo_num_phase_tap(const vo_chan *c) { return c->nphs; }
const float *vo_phase_tap(const vo_chan *c) { return c->phs; }

/* ------------------------------------------------------------ header Viterbi */
static void header_viterbi_start(vo_chan *c)
{
	c->pb[0][0] = 1.0;
	for (int s = 1; s < 32; s++)
		c->pb[0][s] = 0.0;
}

static void header_viterbi_step(vo_chan *c, float v, int n)
{
	double *cur = c->pb[n], *nxt = c->pb[n + 1];
	for (int s = 0; s < 32; s++)
		nxt[s] = 0.0;
	for (int s = 0; s < 32; s++) {
		if (cur[s] == 0.0)
			continue;
		/* bit = 1 moves the syndrome by the column, bit = 0 keeps it;
		 * the survivor is replaced only by a strictly larger metric */
		double m1 = cur[s] * v;
		int t = s ^ HCOL[n];
		if (m1 > nxt[t]) {
			nxt[t] = m1;
			c->bk[n + 1][t] = s;
			c->bv[n + 1][t] = 1;
		}
		double m0 = cur[s] * (1.0 - v);
		if (m0 > nxt[s]) {
			nxt[s] = m0;
			c->bk[n + 1][s] = s;
			c->bv[n + 1][s] = 0;
		}
	}
}

static unsigned header_viterbi_finish(vo_chan *c)
{
	unsigned word = 0, mask = 1;
	int s = 0;
	for (int n = 25; n > 0; n--) {
		if (c->bv[n][s])
			word |= mask;
		s = c->bk[n][s];
		mask <<= 1;
	}
	return word;
}

unsigned vo_header_decode(const float soft[25], uint32_t *bits_out)
{
	vo_chan *c = calloc(1, sizeof *c);
	header_viterbi_start(c);
	for (int n = 0; n < 25; n++)
		header_viterbi_step(c, soft[n], n);
	unsigned w = header_viterbi_finish(c);
	free(c);
	if (bits_out)
		*bits_out = w;
	return vo_reversebits(w >> 5, 17);
}

/* --------------------------------------------------------------- burst bits */
static void emit_block(vo_chan *c)
{
	c->blocks = grow(c->blocks, &c->capblocks, c->nblocks + 1, sizeof(vo_block));
	c->cur.df = c->df;
	c->cur.end_dec = c->ndec - 1;
	c->blocks[c->nblocks++] = c->cur;
	/* decodeVdlm2 vdlm2.c:201-203 hands the DSP a fresh zeroed block */
	memset(&c->cur, 0, sizeof c->cur);
}

static void take_bit(vo_chan *c, float sv)
{
	/* descrambler d8psk.c:54-65 (runs in every state, d8psk.c:71) */
	unsigned pn = (c->scrambler ^ (c->scrambler >> 14)) & 1u;
	c->scrambler = (c->scrambler << 1) | pn;
	float v = pn ? (float)(1.0 - (double)sv) : sv;

	switch (c->state) {
	case ST_WSYNC:
		return;
	case ST_HEAD:
		if (c->nbits < 3)
			v = 0;
		{
			vo_trigger *th = &c->trigs[c->ntrigs - 1];	/* tap: tests/golden/*.json hold the real reference's digest of these */
			th->head[th->nhead++] = v;
		}
		header_viterbi_step(c, v, (int)c->nbits);
		if (++c->nbits < 25)
			return;
		{
			unsigned word = header_viterbi_finish(c) >> 5;
			unsigned len = vo_reversebits(word, 17);
			c->nbrow = len / 1992 + 1;
			c->nlbyte = (len % 1992 + 7) / 8;
			c->cur.nbrow = (int32_t) c->nbrow;
			c->cur.nlbyte = (int32_t) c->nlbyte;
			vo_trigger *t = &c->trigs[c->ntrigs - 1];
			t->len_bits = (int32_t) len;
			if (len < 12 * 8 || c->nbrow > 8) {
				t->accepted = 0;
				c->state = ST_WSYNC;
				return;
			}
			t->accepted = 1;
			c->state = ST_DATA;
			c->nrow = c->nbyte = 0;
			c->nbits = 0;
			c->bits = 0;
		}
		return;
	case ST_DATA:
	case ST_FEC:
		{
			const int fec = (c->state == ST_FEC);
			const unsigned ncol = fec ? 6u : 249u;
			const unsigned base = fec ? 249u : 0u;
			if ((double)v > 0.5)
				c->bits |= (unsigned char)(1u << c->nbits);
			if (++c->nbits < 8)
				return;
			c->cur.data[c->nrow][base + c->nbyte] = c->bits;
			c->nbits = 0;
			c->bits = 0;
			/* column-major walk over the rows, short last row zero-filled */
			if (++c->nrow == c->nbrow) {
				c->nrow = 0;
				c->nbyte++;
			}
			if (c->nlbyte)
				while (c->nrow == c->nbrow - 1 && c->nbyte >= c->nlbyte && c->nbyte < ncol) {
					c->cur.data[c->nrow][base + c->nbyte] = 0;
					c->nrow = 0;
					c->nbyte++;
				}
			if (c->nbyte != ncol)
				return;
			if (!fec) {
				c->state = ST_FEC;
				c->nrow = c->nbyte = 0;
				/* FEC shortening of the last row, d8psk.c:153-161 */
				if (c->nlbyte <= 2) {
					c->nlbyte = 0;
					c->nbrow--;
				} else if (c->nlbyte <= 30)
					c->nlbyte = 2;
				else if (c->nlbyte <= 67)
					c->nlbyte = 4;
				else
					c->nlbyte = 0;
			} else {
				emit_block(c);
				c->state = ST_WSYNC;
			}
		}
		return;
	}
}

/* ------------------------------------------------------------ demodulation */
static float fir_phase(vo_chan *c)
{
	float sr = 0, si = 0;
	int k = c->ink;
	for (int i = c->clk; i < NTAP; i += 4) {
		float m = T_mflt[i].f;
		sr += c->in_re[k] * m;
		si += c->in_im[k] * m;
		k = (k + 1) % NRING;
	}
	float p = atan2f(si, sr);
	if (c->tap_ph) {
		c->phs = grow(c->phs, &c->capphs, c->nphs + 1, sizeof(float));
		c->phs[c->nphs++] = p;
	}
	return p;
}

/* 17-point unwrap + straight-line fit of (phase - sync word); d8psk.c:257-289 */
static float sync_metric(const vo_chan *c, float *slope)
{
	float pr[NSYNC];
	float pu = 0;
	float pv = c->ph[(c->phidx + 4) % NPH] - T_sw[0].f;
	float mean = pv;
	pr[0] = pv;
	for (int l = 1; l < NSYNC; l++) {
		float pc = c->ph[(c->phidx + (l + 1) * 4) % NPH] - T_sw[l].f;
		float pd = pc - pv;
		pv = pc;
		if ((double)pd > M_PI)
			pu = (float)((double)pu - 2 * M_PI);
		else if ((double)pd < -M_PI)
			pu = (float)((double)pu + 2 * M_PI);
		pr[l] = pc + pu;
		mean += pr[l];
	}
	mean /= 17.0f;
	float fr = 0;
	for (int l = 0; l < NSYNC; l++) {
		pr[l] -= mean;
		fr += pr[l] * (float)(l - 8);
	}
	fr /= 408.0f;
	float err = 0;
	for (int l = 0; l < NSYNC; l++) {
		float e = pr[l] - (float)(l - 8) * fr;
		err += e * e;
	}
	*slope = fr;
	return err;
}

static void slice_symbol(vo_chan *c, float p)
{
	float d = (p - c->p1) - c->df;
	if ((double)d > M_PI)
		d = (float)((double)d - 2 * M_PI);
	if ((double)d < -M_PI)
		d = (float)((double)d + 2 * M_PI);
	int i = (int)roundf((float)(128.0 * (double)d / M_PI + 128.0));
	take_bit(c, T_grey1[i].f);
	take_bit(c, T_grey2[i].f);
	take_bit(c, T_grey3[i].f);
	c->p1 = p;
}

static void demod_sample(vo_chan *c, float re, float im)
{
	if (c->tap_dec) {
		c->dec = grow(c->dec, &c->capdec, 2 * (size_t) (c->ndec + 1), sizeof(float));
		c->dec[2 * c->ndec] = re;
		c->dec[2 * c->ndec + 1] = im;
	}
	c->ndec++;
	c->in_re[c->ink] = re;
	c->in_im[c->ink] = im;
	c->ink = (c->ink + 1) % NRING;
	c->clk += 4;

	if (c->state != ST_WSYNC) {	/* one symbol every 32 clock units */
		if (c->clk < 32)
			return;
		c->clk -= 32;
		slice_symbol(c, fir_phase(c));
		return;
	}
	if (c->clk < 8)		/* idle: evaluate every 2nd sample (T/4) */
		return;
	c->clk -= 8;
	float p = fir_phase(c);
	c->phidx = (c->phidx + 1) % NPH;
	c->ph[c->phidx] = p;
	float fr;
	float err = sync_metric(c, &fr);
	if ((double)c->perr < 4.0 && err > c->perr) {
		/* one step past the minimum of the fit error: burst start */
		c->trigs = grow(c->trigs, &c->captrigs, c->ntrigs + 1, sizeof(vo_trigger));
		vo_trigger *t = &c->trigs[c->ntrigs++];
		memset(t, 0, sizeof *t);
		t->dec_index = c->ndec - 1;
		t->p2err = c->p2err;
		t->perr = c->perr;
		t->err = err;
		t->pfr = c->pfr;
		c->state = ST_HEAD;
		c->nbits = 0;
		c->scrambler = 0x4D4B;
		header_viterbi_start(c);
		c->df = c->pfr;
		c->cur.ppm = (float)((double)(10500.0f * c->df) / (2.0 * M_PI * (double)c->fr) * 1e6);
		c->cur.trig_dec = c->ndec - 1;
		/* parabolic interpolation of the minimum -> sampling instant */
		float of = 4.0f * (c->p2err - 4.0f * c->perr + 3.0f * err) / (c->p2err - 2.0f * c->perr + err);
		c->clk = (int)roundf(of);
		t->of = of;
		t->clk = c->clk;
		t->accepted = -1;
		c->p1 = fir_phase(c);
		c->perr = c->p2err = 500;
	} else {
		c->p2err = c->perr;
		c->perr = err;
		c->pfr = fr;
	}
}

/* --------------------------------------------------------- mixer + decimator */
static inline void dump_if_due(vo_chan *c)
{
	c->nf++;
	c->no = (c->no + 1) % c->lo_len;
	c->mclk += 21;
	if (c->mclk >= (int)c->sdrclk) {
		c->mclk %= (int)c->sdrclk;
		float n = (float)c->nf;
		float re = c->acc_re / n, im = c->acc_im / n;
		c->acc_re = c->acc_im = 0;
		c->nf = 0;
		demod_sample(c, re, im);
	}
}

static inline void mix_complex(vo_chan *c, float xr, float xi)
{
	float wr = c->lo_re[c->no], wi = c->lo_im[c->no];
	float pr = xr * wr - xi * wi;
	float pi = xr * wi + xi * wr;
	c->acc_re += pr;
	c->acc_im += pi;
	dump_if_due(c);
}

static inline void mix_real(vo_chan *c, float x)
{
	/* WITH_AIR build: real sample times complex LO (d8psk.c:368 with float Cbuff) */
	c->acc_re += x * c->lo_re[c->no];
	c->acc_im += x * c->lo_im[c->no];
	dump_if_due(c);
}

void vo_feed(vo_chan *c, const void *raw, size_t n, int fmt)
{
	size_t i;
	switch (fmt) {
	case VO_FMT_CU8:{
			const uint8_t *b = raw;
			for (i = 0; i < n; i++)
				mix_complex(c, (float)b[2 * i] - (float)127.37, (float)b[2 * i + 1] - (float)127.37);
			break;
		}
	case VO_FMT_CU8_QUIRK:{
			/* rtl.c:285-292 as written: per 32768-sample block the consumer
			 * sees [0, s0 .. s32766] */
			const uint8_t *b = raw;
			for (i = 0; i < n; i++) {
				size_t k = i % 32768;
				if (k == 0)
					mix_complex(c, 0.0f, 0.0f);
				else {
					size_t j = i - 1;
					mix_complex(c, (float)b[2 * j] - (float)127.37, (float)b[2 * j + 1] - (float)127.37);
				}
			}
			break;
		}
	case VO_FMT_CS16:{
			const int16_t *s = raw;
			for (i = 0; i < n; i++)
				mix_complex(c, (float)s[2 * i], (float)s[2 * i + 1]);
			break;
		}
	case VO_FMT_CF32:{
			const float *s = raw;
			for (i = 0; i < n; i++)
				mix_complex(c, s[2 * i], s[2 * i + 1]);
			break;
		}
	case VO_FMT_F32R:{
			const float *s = raw;
			for (i = 0; i < n; i++)
				mix_real(c, s[i]);
			break;
		}
	default:
		abort();
	}
}

/* =================================================================== host path
 * RS(255,249) over GF(256)/0x187, first consecutive root alpha^120
 * (rs.c:17-79 tables, rs.c:81-291 decoder).  Errors-and-erasures
 * Berlekamp-Massey, Chien search, Forney; the update order follows the
 * reference so that even miscorrections and partial updates agree. */
static uint8_t gf_exp[512];
static uint8_t gf_log[256];
static int gf_ready;

static void gf_init(void)
{
	if (gf_ready)
		return;
	unsigned x = 1;
	for (int i = 0; i < 255; i++) {
		gf_exp[i] = (uint8_t) x;
		gf_log[x] = (uint8_t) i;
		x <<= 1;
		if (x & 0x100)
			x ^= 0x187;
	}
	for (int i = 255; i < 512; i++)
		gf_exp[i] = gf_exp[i - 255];
	gf_log[0] = 255;	/* "A0" */
	gf_ready = 1;
}

#define NOLOG 255
static inline int m255(int x)
{
	while (x >= 255) {
		x -= 255;
		x = (x >> 8) + (x & 255);
	}
	return x;
}

int vo_rs_decode(uint8_t *data, int *eras_pos, int no_eras)
{
	enum { NR = 6, NNN = 255, FIRST = 120 };
	uint8_t lam[NR + 1], syn[NR], bpoly[NR + 1], tpoly[NR + 1], omg[NR + 1];
	uint8_t root[NR], reg[NR + 1], loc[NR];
	int count = 0;
	gf_init();

	/* syndromes by Horner at alpha^(120+i) */
	for (int i = 0; i < NR; i++)
		syn[i] = data[0];
	for (int j = 1; j < NNN; j++)
		for (int i = 0; i < NR; i++)
			syn[i] = syn[i] ? (uint8_t) (data[j] ^ gf_exp[m255(gf_log[syn[i]] + FIRST + i)]) : data[j];
	int any = 0;
	for (int i = 0; i < NR; i++) {
		any |= syn[i];
		syn[i] = gf_log[syn[i]];	/* index form, NOLOG for zero */
	}
	if (!any)
		goto done;

	memset(lam, 0, sizeof lam);
	lam[0] = 1;
	if (no_eras > 0) {
		lam[1] = gf_exp[m255(NNN - 1 - eras_pos[0])];
		for (int i = 1; i < no_eras; i++) {
			int u = m255(NNN - 1 - eras_pos[i]);
			for (int j = i + 1; j > 0; j--) {
				uint8_t lg = gf_log[lam[j - 1]];
				if (lg != NOLOG)
					lam[j] ^= gf_exp[m255(u + lg)];
			}
		}
	}
	for (int i = 0; i <= NR; i++)
		bpoly[i] = gf_log[lam[i]];

	int el = no_eras;
	for (int r = no_eras + 1; r <= NR; r++) {
		uint8_t disc = 0;
		for (int i = 0; i < r; i++)
			if (lam[i] && syn[r - i - 1] != NOLOG)
				disc ^= gf_exp[m255(gf_log[lam[i]] + syn[r - i - 1])];
		uint8_t dl = gf_log[disc];
		if (dl == NOLOG) {
			memmove(&bpoly[1], bpoly, NR);
			bpoly[0] = NOLOG;
			continue;
		}
		tpoly[0] = lam[0];
		for (int i = 0; i < NR; i++)
			tpoly[i + 1] = (bpoly[i] != NOLOG) ? (uint8_t) (lam[i + 1] ^ gf_exp[m255(dl + bpoly[i])]) : lam[i + 1];
		if (2 * el <= r + no_eras - 1) {
			el = r + no_eras - el;
			for (int i = 0; i <= NR; i++)
				bpoly[i] = lam[i] ? (uint8_t) m255(gf_log[lam[i]] - dl + NNN) : NOLOG;
		} else {
			memmove(&bpoly[1], bpoly, NR);
			bpoly[0] = NOLOG;
		}
		memcpy(lam, tpoly, NR + 1);
	}

	int deg = 0;
	for (int i = 0; i <= NR; i++) {
		lam[i] = gf_log[lam[i]];
		if (lam[i] != NOLOG)
			deg = i;
	}
	/* Chien search */
	memcpy(&reg[1], &lam[1], NR);
	for (int i = 1, k = 0; i <= NNN; i++, k = m255(k + 1)) {
		uint8_t q = 1;
		for (int j = deg; j > 0; j--)
			if (reg[j] != NOLOG) {
				reg[j] = (uint8_t) m255(reg[j] + j);
				q ^= gf_exp[reg[j]];
			}
		if (q)
			continue;
		root[count] = (uint8_t) i;
		loc[count] = (uint8_t) k;
		if (++count == deg)
			break;
	}
	if (deg != count) {
		count = -1;
		goto done;
	}
	int dego = 0;
	for (int i = 0; i < NR; i++) {
		uint8_t tmp = 0;
		for (int j = (deg < i) ? deg : i; j >= 0; j--)
			if (syn[i - j] != NOLOG && lam[j] != NOLOG)
				tmp ^= gf_exp[m255(syn[i - j] + lam[j])];
		if (tmp)
			dego = i;
		omg[i] = gf_log[tmp];
	}
	omg[NR] = NOLOG;
	/* Forney, last root first */
	for (int j = count - 1; j >= 0; j--) {
		uint8_t num1 = 0;
		for (int i = dego; i >= 0; i--)
			if (omg[i] != NOLOG)
				num1 ^= gf_exp[m255(omg[i] + i * root[j])];
		uint8_t num2 = gf_exp[m255(root[j] * (FIRST - 1) + NNN)];
		uint8_t den = 0;
		int top = (deg < NR - 1 ? deg : NR - 1) & ~1;
		for (int i = top; i >= 0; i -= 2)
			if (lam[i + 1] != NOLOG)
				den ^= gf_exp[m255(lam[i + 1] + i * root[j])];
		if (den == 0) {
			count = -1;
			goto done;
		}
		if (num1)
			data[loc[j]] ^= gf_exp[m255(gf_log[num1] + gf_log[num2] + NNN - gf_log[den])];
	}
done:
	if (eras_pos)
		for (int i = 0; i < count; i++)
			eras_pos[i] = loc[i];
	return count;
}

static uint16_t fcs_step(uint16_t crc, uint8_t c)
{
	/* reflected CRC-16 poly 0x8408: the table at crc.c is this function tabulated */
	crc ^= c;
	for (int i = 0; i < 8; i++)
		crc = (crc & 1) ? (uint16_t) ((crc >> 1) ^ 0x8408) : (uint16_t) (crc >> 1);
	return crc;
}

static int frame_ok(const uint8_t *h, int l)
{
	if (l < 13)
		return 0;
	uint16_t crc = 0xffff;
	for (int i = 1; i < l - 1; i++)
		crc = fcs_step(crc, h[i]);
	return crc == 0xf0b8;
}

int vo_block_frames(vo_block *b, uint8_t *out, size_t out_cap, size_t *out_used)
{
	uint8_t h[8 * 249 + 8];
	int k = 0, s = 0, t = 0, nframes = 0;
	int eras[6] = { 0 };
	size_t used = out_used ? *out_used : 0;
	h[0] = 0;
	for (int r = 0; r < b->nbrow; r++) {
		int by = 249, nera = 0;
		if (r == b->nbrow - 1) {
			by = b->nlbyte;
			if (by <= 67) {
				nera = 2;
				eras[0] = 253;
				eras[1] = 254;
			}
			if (by <= 30) {
				nera = 4;
				eras[0] = 251;
				eras[1] = 252;
				eras[2] = 253;
				eras[3] = 254;
			}
		}
		vo_rs_decode(b->data[r], eras, nera);
		for (int i = 0; i < by; i++)
			for (int n = 0; n < 8; n++) {
				if (b->data[r][i] & (1 << n)) {
					h[k] |= (uint8_t) (1 << s);
					t++;
				} else {
					if (t == 5) {	/* stuffed zero */
						t = 0;
						continue;
					}
					t = 0;
				}
				if (++s < 8)
					continue;
				s = 0;
				if (h[k] == 0x7e) {
					if (k == 0) {
						h[++k] = 0;
					} else if (k == 1) {
						h[1] = 0;
					} else {
						if (frame_ok(h, k + 1)) {
							nframes++;
							if (out && used + 2 + (size_t) (k + 1) <= out_cap) {
								out[used] = (uint8_t) ((k + 1) & 0xff);
								out[used + 1] = (uint8_t) ((k + 1) >> 8);
								memcpy(out + used + 2, h, (size_t) (k + 1));
								used += 2 + (size_t) (k + 1);
							}
						}
						h[++k] = 0;
					}
				} else if (k > 0) {
					h[++k] = 0;
				}
			}
	}
	if (out_used)
		*out_used = used;
	return nframes;
}
