/*
 * vdl2_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement ("port") of the reference's per-channel IQ->bits path and of
 * the host block path that consumes its output.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load this library; the
 * product (libvdl2gpu.so) never links or calls it.
 *
 * Parity status: PINNED.  oracle/Makefile target `ref` builds the real
 * reference (d8psk.c, viterbi.c, vdlm2.c, crc.c, rs.c compiled where they lie)
 * into oracle/_ref/; tests/test_oracle_vs_ref.py checks this restatement
 * against it bit-for-bit (every atan2f argument/result, every header soft bit,
 * every msgblk_t, every CRC-clean frame) and tests/golden/ holds vectors
 * generated from that reference build for machines without /root/reference.
 */
#ifndef VDL2_ORACLE_H
#define VDL2_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VO_MAXROWS 8
#define VO_ROWLEN 255

/* one burst as handed to decodeVdlm2() (d8psk.c:201): the fields of msgblk_t
 * (vdlm2.h:39-47) the DSP fills, minus the wall-clock timestamp */
typedef struct {
	int32_t nbrow, nlbyte;	/* header values, d8psk.c:94-95 */
	float df;		/* channel_t.df at hand-off */
	float ppm;		/* d8psk.c:302 */
	int64_t trig_dec;	/* decimated-sample index (0-based) of the sync trigger */
	int64_t end_dec;	/* decimated-sample index of the symbol that completed the burst */
	uint8_t data[VO_MAXROWS][VO_ROWLEN];
} vo_block;

typedef struct {
	int64_t dec_index;	/* decimated-sample index of the trigger evaluation */
	float p2err, perr, err, pfr, of;
	int32_t clk;		/* (int)roundf(of) */
	int32_t accepted;	/* 1 = header accepted, 0 = rejected (d8psk.c:97-107) */
	int32_t len_bits;
	int32_t nhead;		/* header soft bits taken so far (25 unless the stream ended inside the header) */
	float head[25];		/* what viterbi_add() is given, in call order: descrambled, bits 0..2 forced to 0 (d8psk.c:81-83) */
} vo_trigger;

typedef struct vo_chan vo_chan;

enum { VO_FMT_CU8 = 0, VO_FMT_CS16 = 1, VO_FMT_CF32 = 2, VO_FMT_F32R = 3, VO_FMT_CU8_QUIRK = 4 };

vo_chan *vo_create(unsigned sdrinrate, int fo_hz, int fr_hz);
void vo_destroy(vo_chan *c);
/* keep every decimated sample / every WSYNC phase for diagnostics (P3 taps) */
void vo_enable_taps(vo_chan *c, int dec_samples, int phases);
/* n = number of SAMPLES (complex pairs, or real samples for VO_FMT_F32R).
 * VO_FMT_CU8_QUIRK reproduces rtl.c:285-292 as written (n must be a multiple
 * of 32768: each block becomes [0, s0 .. s32766]). */
void vo_feed(vo_chan *c, const void *raw, size_t n, int fmt);

size_t vo_num_blocks(const vo_chan *c);
const vo_block *vo_blocks(const vo_chan *c);
size_t vo_num_triggers(const vo_chan *c);
const vo_trigger *vo_triggers(const vo_chan *c);
size_t vo_num_dec(const vo_chan *c);		/* decimated samples emitted so far */
const float *vo_dec_tap(const vo_chan *c);	/* interleaved re,im (if enabled) */
size_t vo_num_phase_tap(const vo_chan *c);
const float *vo_phase_tap(const vo_chan *c);	/* every filteredphase() result in call order */
void vo_lo_table(const vo_chan *c, float *re_im_out, int *len_out);

/* host block path (vdlm2.c:64-161 + rs.c:81-291 + crc.c): RS-decode the rows in
 * place, HDLC-unstuff, FCS-check.  Frames are appended to `out` as
 * [u16 length][bytes...]; returns the number of CRC-clean frames. */
int vo_block_frames(vo_block *b, uint8_t *out, size_t out_cap, size_t *out_used);
/* errors-and-erasures RS(255,249) decoder, same contract as rs.c:81 */
int vo_rs_decode(uint8_t *data, int *eras_pos, int no_eras);

/* small pure helpers exposed for known-answer tests */
unsigned vo_reversebits(unsigned bits, int n);				/* d8psk.c:39-52 */
void vo_pn_bits(uint8_t *out, size_t n);				/* d8psk.c:54-65 seed 0x4D4B */
unsigned vo_header_decode(const float soft[25], uint32_t *bits_out);	/* viterbi.c + d8psk.c:88-92 */
float vo_atan2f(float y, float x);	/* the libm atan2f this build links (for vdl2_math.h checks) */

#ifdef __cplusplus
}
#endif
#endif
