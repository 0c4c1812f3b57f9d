/* vdl2gpu_types.h -- types, parameters and constant tables of the device side.  Part of the device side of libvdl2gpu.so; included by vdl2gpu_kernels.h only. */
#ifndef VDL2GPU_TYPES_H
#define VDL2GPU_TYPES_H


#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vdl2_math.h"
#include "../../include/vdl2gpu.h"

#define VDL2_CS 8		/* channel planes per stream */
#ifndef VDL2_NSET
#define VDL2_NSET 3		/* plane sets, table sets and output rings, used in turn (push % VDL2_NSET): so many pushes in the pipeline, the oldest in its tail */
#endif
#define VDL2_NRING VDL2_NSET	/* output rings (records, frames, counters) */
#define VDL2_NSLAB (VDL2_NSET + 1)	/* page-locked slabs the records are exported to, used in turn */
static_assert(VDL2_NSET >= 3 && VDL2_NSET <= 4, "d_outc[] has the per-ring counters in front of the running totals at [8]; three bits of a record handle name its slab");
#define VDL2_HIST 160		/* frames of history kept: 17-tap FIR + 17 symbols x 8 + slack */
#define VDL2_NPH 68		/* NBPH*D8DWN, vdlm2.h:54-55 */
#define VDL2_STEADY 68		/* evaluations after which the detector forgot the last burst */
#define VDL2_MAXSYM 5456	/* >= ceil((25 + 8*8*255)/3) symbols of the longest burst */
#define VDL2_SERIAL_BELOW 4096	/* pushes of at most this many 84 kS/s frames go to the serial machine directly */
#define VDL2_CARRY_FRAMES 49152	/* >= longest burst (43592 frames) + history + slack */
#define VDL2_PN_BITS (16384 + 64)
#define VDL2_CAND_CAP 6144	/* trigger candidates per channel per push (round 5: 4096 -> 6144, what the resolver's 23 bytes of LDS per candidate allow) */
#define VDL2_CL_MAXB 4		/* bursts per cluster before the resolver takes over */
#define VDL2_SEL_CAP 16384	/* bursts on the real chain per channel per push */

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

struct StreamState {
	long long dec_base;	/* stream time (84 kS/s index) of frame 0 of the current planes */
	long long dec_fill;	/* frames present before this push's K1 output */
	long long last_fill;	/* diagnostics: where the last push's output starts */
	long long last_J;
	float2 acc[2][VDL2_CS];	/* integrate-and-dump partial sums carried across pushes */
};

struct ChanState {
	long long pos;		/* stream time of the next WSYNC evaluation */
	int r;			/* FIR sub-phase (channel_t.clk after the -=8), 0..3 */
	int fresh;		/* evaluations since the last trigger/reset, saturating */
	float perr, p2err, pfr;	/* channel_t.perr/p2err/pfr */
	float ring[VDL2_NPH];	/* channel_t.Ph in time order, ring[67] newest */
	unsigned long long n_eval, n_trig, n_reject, n_burst, n_defer, n_slow, n_cand, n_redo;
};

struct ChanCfg {
	int chn, Fr, Fo, pad;
};

struct Cand {			/* free-running detector fires at dec_base + nrel with sub-phase r */
	int nrel, r;
	float p2err, perr, err, pfr;
};

enum { CL_STEADY = 0, CL_DEFER_FIRST = 1, CL_NONSTEADY = 2, CL_INVALID = 3 };
struct Cluster {		/* what the resolver reads of a cluster is its 8-byte head (cl_pack); this is the rest */
	ChanState saved;	/* CL_NONSTEADY: explicit state to continue from */
};

/* resolver's view of a cluster: x = n_s - dec_base, y = status | r_s << 2 | nslots << 4 | ntrig << 8 | nrej << 16 | nburst << 24 */
__device__ __forceinline__ int2 cl_pack(int n_s_rel, int status, int r_s, int nslots, int ntrig, int nrej, int nburst)
{
	ntrig = ntrig > 255 ? 255 : ntrig;
	nrej = nrej > 255 ? 255 : nrej;
	nburst = nburst > 255 ? 255 : nburst;
	return make_int2(n_s_rel, status | (r_s << 2) | (nslots << 4) | (ntrig << 8) | (nrej << 16) | (nburst << 24));
}

struct BurstDesc {		/* a burst found by a cluster; payload decoded later if it is on the real chain */
	long long nstar;	/* stream time of the sync trigger */
	int sc;			/* stream*8 + channel */
	int clk0;		/* (int)roundf(of), d8psk.c:305 */
	float df;
	int nbrow, nlbyte, pad;
};

struct Seg {			/* the chain idled in class (r, parity of lo) over stream-relative [lo, hi) */
	int lo, hi, r, pad;
};

struct HeadTap {		/* diagnostics (VDL2GPU_F_DEBUG_HEADS): what one sync trigger gave the header decoder */
	long long nstar;	/* stream time of the trigger */
	int sc;			/* stream * 8 + channel slot */
	int clk0;		/* (int)roundf(of), d8psk.c:305 */
	float p2err, perr, err, pfr;	/* the detector's view at the trigger (d8psk.c:292-305) */
	float soft[25];		/* descrambled header soft bits as viterbi_add() gets them (d8psk.c:81-83) */
};
static_assert(sizeof(HeadTap) == 132 || sizeof(HeadTap) == 136, "HeadTap layout");

struct K1Params {
	const void *raw;
	size_t stream_stride;
	int fmt, nbch;
	int sdrclk, L, maxwin;
	int c0, no0, nf0, parity;
	int quirk;		/* cu8 only: the store-index off-by-one of rtl.c:291, per 32768-sample block */
	long long N, J;
	long long jbeg, jend;	/* generic kernel: outputs [jbeg, jend] (jend may be J = the carried tail) */
	long long per_lo, per_n;	/* k1_fast: superperiods [per_lo, per_lo+per_n) */
	int edge_state;		/* k1_fast covers the whole push (it starts and ends on a window boundary of the schedule: no carried
				 * partial window): the launch also leaves the stream state the general kernel would (last_fill, last_J, empty carry) */
	unsigned *tickets;	/* k1_fast: [S][21 roles][8 XCDs] work counters; never reset: */
	unsigned tbase[8];	/* ... what the counters of XCD x hold when this launch starts (every launch adds the number of its tickets) */
	const float2 *lo;	/* [S][8][L] */
	const float2 *lo_ext;	/* k1_fast: [S][8][lo_stride], entry 8 + i = lo[i mod L] for -8 <= i < L + 40 */
	int lo_stride;
	float2 *dec;		/* this push's planes, [S][8][cap] */
	long long cap;
	StreamState *ss;
};

struct K1PParams {		/* k1_pp: whole periods (PER = 4*SDRCLK inputs = 84 outputs) [per_lo, per_lo + per_n) of a push */
	const void *raw;
	size_t stream_stride;
	int nbch;
	int per_in;		/* inputs per period */
	int L;			/* LO table length */
	int ph0;		/* LO index of a period's first sample */
	int d;			/* samples between the 16-byte boundary below a period's first sample and that sample */
	int fast_div;		/* both window lengths are in the set the reciprocal division is proven exact for */
	int dbg;		/* development: 1 = no mixing, 2 = no sample loads after the first chunk, 4 = no stores */
	int nf_lo;		/* the shorter of the two window lengths; rcp_lo = RN(1/nf_lo), rcp_hi = RN(1/(nf_lo+1)) */
	float rcp_lo, rcp_hi;
	int per_n;
	int nsub, wpt;		/* a period's 84 windows are split into nsub tasks of wpt */
	long long per_lo;
	long long sbase0;	/* push-relative index of the first sample of period per_lo */
	const float2 *lo_ext;	/* [S][8][lo_stride]: the LO table followed by its first 16 entries again */
	int lo_stride;
	float2 *dec;
	long long cap;
	StreamState *ss;
	int edge_state;		/* the launch covers the whole push (it starts and ends on a window boundary of the schedule): it also leaves the
				 * stream state the general kernel would (last_fill, last_J, an empty carry in acc[parity ^ 1]) */
	int parity;
	long long J;
	int wend[84];		/* last sample of each window, relative to the period's first sample */
};

struct K2Slog {			/* a stretch the first resolver pass handled with the serial machine: where it began (stream-relative) and what it counted */
	int t, ntrig, nrej, nburst;
};
struct K2aItem;
struct K2Params {
	const float2 *dec;
	long long cap;
	int nbch, nstreams;
	long long J;
	long long dec_base;	/* stream time of frame 0 of the planes (host arithmetic: outputs before this push - VDL2_CARRY_FRAMES) */
	long long scan_lo;	/* first instant the scans look at: dec_base + VDL2_HIST (NOT the channel's position: the front stage of a
				 * push runs before the previous push has committed where its channels stand) */
	int probe_r, probe_par;	/* the class the probe scans everywhere (fixed), parity of its instants */
	StreamState *ss;
	ChanState *cs;
	const ChanCfg *cfg;
	const uint8_t *pn;
	const uint8_t *pn8;	/* the scrambler sequence by payload byte: bit i of pn8[b] = pn[25 + 8 b + i] */
	Cand *cands;		/* [S*8][CAND_CAP] */
	Cluster *clusters;	/* [S*8][CAND_CAP] */
	int2 *clhead;		/* [S*8][CAND_CAP] what the resolver needs of every cluster, 8 bytes: see cl_pack() */
	unsigned *ctl;		/* [0]=out count [1]=out overflow [2]=stage count [3]=k2b ticket [4]=stage overflow
				 * [8 + S*8 ...] cand counts, then cand overflow flags */
	BurstDesc *stage;	/* burst descriptors of all clusters */
	unsigned *sel_list;	/* descriptors on the real chain (K2c -> K2d) */
	unsigned *sel_list2;	/* ... as the repair rounds re-resolved them (channels in fmask) */
	int sel_mode;		/* K2d: 0 = the first selection of every channel (beside the verify pass: the mask is not final yet);
				 * 1 = the repaired selection of the masked channels (second pass);
				 * 2 = one pass behind the commit: the repaired selection for masked channels, the first for the others */
	unsigned stage_cap;
	vdl2gpu_burst_t *recs;	/* output ring of this push */
	unsigned *outc;		/* [0] = records written, [1] = records dropped (ring full) */
	unsigned *outc_total_redo;	/* running count of serial redos (host adapts the number of repair rounds) */
	unsigned *fmask;	/* [16] bit per (stream, channel slot) that a repair round re-resolved or K2f redid serially in this push */
	int full_round;		/* this repair round scans the failing channels completely (all classes, every instant): no regions, no verify */
	int mini_round;		/* this repair round re-resolves with what the verify pass found and appended to the tables, nothing else: no scan, no
				 * clusters (the resolver replays the few new candidates itself) */
	int sel_reserved;	/* K2c reserves the output records of a channel's selected bursts in one piece (CTL_SELBASE0), K2d fills them without atomics */
	unsigned rec_cap;
	int force_serial;	/* diagnostics: skip the tables, run the serial machine */
	int prim_drop;		/* test handicap: every prim_drop-th candidate gets no precomputed cluster (the resolver builds it) */
	int full_scan;		/* scan all four sub-phases everywhere (no regions / verify) */
	int test_noregion;	/* test hook: skip the region scan so that K2a-verify must catch the misses */
	int2 *regs;		/* [S*8][REG_CAP] (lo, count) stream-relative */
	Seg *segs;		/* [S*8][SEG_CAP] */
	int *fail;		/* [S*8] earliest unexpected hit (stream-relative), >= VDL2_VERIFIED = verified */
	int *redo;		/* [S*8] 1 = this channel is being re-resolved in the repair round */
	int round;		/* 0 = first pass over every channel; 1 = repair pass over the channels whose
				 * verify failed (the hits were appended to their candidate tables) */
	ChanState *cs_out;	/* resolver result, committed by K2f */
	int *skey;		/* [S*8][CAND_CAP] candidates sorted by time: nrel*4 + r */
	unsigned short *sidx;	/* [S*8][CAND_CAP] sorted rank -> candidate index */
	unsigned short *prim;	/* [S*8][CAND_CAP] candidates whose cluster K2b computes */
	uint8_t *onchain;	/* [S*8][CAND_CAP] by candidate: 1 = the first resolver pass's chain met this candidate in the history-free state (k2p_patch: a
				 * repaired stretch of the chain that arrives at such a candidate has rejoined the old chain) */
	K2Slog *slog;		/* [S*8][VDL2_SLOG_CAP] the first pass's serial stretches (their counts are not in any cluster head) */
	int2 *win;		/* [S*8][VDL2_WIN_CAP] windows of stream-relative time [x, y): bursts of the first selection triggered inside are void (CTL_NWIN0) */
	int *seeds;		/* [S*8][CAND_CAP] probe instants around which all classes are scanned */
	struct K2aItem *items;	/* [S*8][ITEM_CAP] what passed a scan's first screen: a private area per scan workgroup (worked off by that workgroup behind
				 * its last tile), a common area behind them (worked off by the next kernel on the stream) */
	unsigned item_cap;	/* items per channel of the list (private areas + common area): sized by the longest part the handle can be given */
	unsigned item_priv;	/* ... of which private areas at most (host side: launch_scan) */
	int surv_pch, surv_nwg;	/* items a private area holds (a multiple of 256), scan workgroups per channel (= private areas) */
	int surv_common_cap;	/* test hook: the common area holds only so many items (0: all that is left of the list) */
	int surv_slot;		/* which of the push's scans this is: its item counters are ctl[CTL_NSURV0 + slot * S*8 ...] (VDL2_SURV_*) */
	int surv_mode;		/* sparse stages: what a detector hit among the survivors means (k2a_emit: 0 candidates, 1 verify, 2 probe) */
	int surv_skip;		/* sparse stages: hits in the probe's class are in the table already (region scan) */
	/* the one-workgroup-per-channel kernel behind a scan drains that scan's common area first (k2x_drain): which scan, and how its list was laid out */
	int drain_slot;		/* VDL2_SURV_* of the scan in front of this launch, -1: nothing to drain */
	int drain_mode, drain_skip;	/* that scan's surv_mode, surv_skip */
	int drain_pch, drain_nwg;	/* ... surv_pch, surv_nwg */
	unsigned long long *dbg;	/* diagnostics: cycle counters */
	HeadTap *headtap;	/* diagnostics: every trigger any kernel of the push handled (nullptr: off) */
	unsigned *headtap_n;
	unsigned headtap_cap;
};
#define CTL_OUT 0
#define CTL_OUT_OVF 1
#define CTL_STAGE 2
#define CTL_TICKET 3
#define CTL_STAGE_OVF 4
#define CTL_CAND0 8		/* [S*8] candidate counts, [S*8] overflow flags, then: */
#define CTL_NREG0 (CTL_CAND0 + 2 * p.nstreams * VDL2_CS)
#define CTL_NSEG0 (CTL_CAND0 + 3 * p.nstreams * VDL2_CS)
#define CTL_NSEL0 (CTL_CAND0 + 4 * p.nstreams * VDL2_CS)
#define CTL_NPRIM0 (CTL_CAND0 + 5 * p.nstreams * VDL2_CS)
#define CTL_NSEED0 (CTL_CAND0 + 6 * p.nstreams * VDL2_CS)
#define CTL_NCLUST0 (CTL_CAND0 + 7 * p.nstreams * VDL2_CS)	/* candidates [0, n) had their clusters decided by the previous K2s/K2b of this push */
#define CTL_NSURV0 (CTL_CAND0 + 8 * p.nstreams * VDL2_CS)	/* [VDL2_SURV_SLOTS][S*8] item counts, one set per scan of the push */
enum { VDL2_SURV_PROBE = 0, VDL2_SURV_REGION = 1, VDL2_SURV_VERIFY = 2 /* + repair round (1..4) */, VDL2_SURV_FULL = 7, VDL2_SURV_SLOTS = 8 };
#define CTL_SELBASE0 (CTL_CAND0 + (8 + VDL2_SURV_SLOTS) * p.nstreams * VDL2_CS)	/* first output record of the channel's selected bursts (K2c reserves, K2d fills) */
/* The selection exists twice: the first resolver pass writes sel_list / CTL_NSEL0 / CTL_SELBASE0, the repair rounds sel_list2 and
 * the words below -- the payload decode of the first selection runs beside the verify pass AND beside the repair round (on a
 * stream of its own), which therefore must not touch what it reads. */
#define CTL_NSEL1 (CTL_CAND0 + (9 + VDL2_SURV_SLOTS) * p.nstreams * VDL2_CS)
#define CTL_SELBASE1 (CTL_CAND0 + (10 + VDL2_SURV_SLOTS) * p.nstreams * VDL2_CS)
#define CTL_NSURVLIM0 (CTL_CAND0 + (11 + VDL2_SURV_SLOTS) * p.nstreams * VDL2_CS)	/* [VDL2_SURV_SLOTS][S*8] ~(where the first group that the common area
											 * refused would have begun): 0 = none refused; items below it are complete */
#define CTL_NWIN0 (CTL_CAND0 + (11 + 2 * VDL2_SURV_SLOTS) * p.nstreams * VDL2_CS)	/* [S*8] what the repair rounds made void of the channel's FIRST selection: 0 nothing,
											 * 1..VDL2_WIN_CAP: the bursts triggered inside so many windows of stream time (K2Params.win:
											 * a local repair, k2p_patch), VDL2_WIN_ALL: all of it (the channel was resolved again from its
											 * input state, or redone serially) */
#define CTL_NSLOG0 (CTL_CAND0 + (12 + 2 * VDL2_SURV_SLOTS) * p.nstreams * VDL2_CS)	/* [S*8] entries of K2Params.slog the first resolver pass made (VDL2_SLOG_CAP + 1: more than it holds) */
#define VDL2_CTL_WORDS(nsc) (CTL_CAND0 + (13 + 2 * VDL2_SURV_SLOTS) * (size_t)(nsc))
#define VDL2_WIN_CAP 32
#define VDL2_WIN_ALL 0xffffffffu
#define VDL2_SLOG_CAP 16

struct K3Params {
	const float2 *src;
	float2 *dst;
	long long cap;
	int nbch;
	long long J;
	StreamState *ss;
	const ChanState *cs;
	const unsigned *outc;	/* device counters: [2*ring] records, [2*ring+1] dropped, [4] serial redos so far */
	const unsigned *fmask;	/* K2f's redo mask of this push (16 words) */
	const unsigned *fcnt;	/* block path in the pipeline (VDL2GPU_F_FRAMES): frames, dropped, bytes; else nullptr */
	unsigned *host_cnt;	/* the same, in pinned host memory, for this push's ring ([4..6]: frame counters, [7]: channels whose
				 * candidate tables overflowed in this push) */
	int ring;
	const unsigned *ctl;	/* the push's control words */
	int nstreams;
};

struct KInitParams {		/* per-push reset of the demodulator's control words */
	unsigned *ctl;
	int ctl_words;
	unsigned *outc;		/* 2 words of this push's ring */
	int *fail, *redo;
	int nsc;
	unsigned *fmask;	/* 16 words */
	unsigned *fcnt;		/* 4 words: this ring's block-path counters, or nullptr */
};

/* ---- constant data tables (d8psk.h:20-249) as bit patterns ------------- */
#define VDL2_TABLE_BEGIN(name, n) __constant__ uint32_t c_##name[n] = {
#define VDL2_F32(x) x,
#define VDL2_TABLE_END };
#include "vdl2_tables.inc"
#undef VDL2_TABLE_BEGIN
#undef VDL2_F32
#undef VDL2_TABLE_END

/* parity-check columns of the (25,20) header code (data, viterbi.c:29-35) */
__constant__ int c_hcol[25] = { 6, 7, 9, 10, 11, 12, 14, 15, 17, 19, 21, 22, 24, 25, 26, 27, 28, 29, 30, 31,
	16, 8, 4, 2, 1
};

#endif
