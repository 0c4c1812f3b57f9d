/*
 * vdl2gpu_kernels.h -- device side of libvdl2gpu.so (gfx950 only).
 *
 * Data layout in HBM
 *   raw      wideband IQ exactly as the SDR delivers it (cu8 / cs16 / cf32 /
 *            real f32), one contiguous run per stream; read ONCE by K1.
 *   lo       per (stream, channel) local-oscillator table, L = SDRINRATE/25000
 *            complex floats, computed on the host with libm (d8psk.c:353-357).
 *   dec      84 kS/s channel planes: plane (stream, channel) = `cap` float2,
 *            three sets used in turn.  K1 writes from frame VDL2_CARRY_FRAMES on, K2* read, K3 copies
 *            the last VDL2_CARRY_FRAMES frames below frame VDL2_CARRY_FRAMES of the next
 *            set, where the next push's K1 output starts.  Frame 0 of a plane is stream
 *            time `dec_base`; VDL2_HIST frames of history are always kept.
 *   state    StreamState (decimator carry) + ChanState (sync detector state:
 *            next evaluation instant, FIR sub-phase, last 68 phases, last two
 *            fit errors) -- the explicit, persistent form of the reference's
 *            stack-resident channel_t (vdlm2.h:56-79).
 *   cands    per channel: sync-trigger candidates of the free-running detector
 *            under all 8 timing hypotheses (K2a) + what happens after each (K2b).
 *   bursts   staging pool (K2b) and output ring (K2c/K2d) of vdl2gpu_burst_t.
 *
 * Pipeline of one push.  Three stages on three streams, planes / tables / output rings exist three times: the FRONT stage
 * of push N+1 runs beside the BACK stage of push N and the TAIL of push N-1 (host side: vdl2gpu.hip, enqueue_back).
 *  front
 *   K1   channelise       time-parallel over the whole GPU, the only full-rate kernel (vdl2gpu_k1.h)
 *   K2a  sync scan        probe (ONE fixed detector class over the whole push, carry included), regions (all classes
 *                         around what it found).  Screens that prove where the detector cannot fire; exact FIR /
 *                         atan2f / fit only for what survives them; only the first firing of a run is listed (vdl2gpu_scan.h)
 *   K2s  sort             candidates by time, primaries marked (vdl2gpu_resolve.h)
 *   K3   carry            the last 49152 frames of every plane to the next plane set (a fixed amount: depends on nothing
 *                         the resolver decides)
 *  back
 *   K2b  burst clusters   one wavefront per primary candidate: exact state machine (vdl2gpu_machine.h)
 *                         from the trigger until the detector is history-free again
 *   K2c  resolve          one workgroup per VDL channel: walks the real chain of bursts through the tables
 *   K2a  verify           every stretch the chain idled through, in the class it idled in, unless the probe covered it;
 *                         what it finds joins the table (a candidate without a cluster)
 *   K2d  payload          symbols, slicer, descrambler, de-interleaver of the bursts on the chain (beside the verify pass)
 *  tail
 *   K2s' K2c K2a          repair round, always scheduled: merge what the verify pass listed into the sorted table,
 *                         resolve the failing channels again, verify what changed (all exit at once when nothing failed)
 *   K2f  commit           (or serial redo of a channel whose verify pass still fails)
 *   K2d  payload          of the channels a round re-resolved
 *   K4   block path       optional: RS / HDLC / FCS per burst (vdl2gpu_blocks.h)
 *   K3   export, rebase   records into page-locked host memory by the GPU's own hand; counters to the host
 *
 * Why the tables are exact: between bursts the reference detector evaluates
 * every 2nd 84 kS/s sample with a sticky FIR sub-phase r = clk%4 and sample
 * parity, both of which only change at a sync trigger (d8psk.c:248-250,
 * 305, 317-319).  68 evaluations after a burst the phase ring Ph[] holds only
 * new values and perr/p2err are the two previous errors, so the detector's
 * decision at sample n is a pure function of (n, r): the "free-running" fit
 * error E_r(n).  The scan finds every (n, r) at which it can fire; K2b/K2c replay
 * the short history-dependent stretches (stale ring after a burst)
 * with the serial machine; K2a-verify re-scans what the chain relied on.
 *
 * Arithmetic contract: every float/double operation below is written in the
 * order and width of the reference C expression it replaces and this file is
 * compiled with -ffp-contract=off, so all intermediate values (decimated
 * samples, FIR outputs, phases, fit errors, soft bits) are bit-identical to
 * the reference built -O2 on x86-64, not merely the decisions.
 */
#ifndef VDL2GPU_KERNELS_H
#define VDL2GPU_KERNELS_H

#include "vdl2gpu_types.h"	/* layouts, parameter blocks, tables */
#include "vdl2gpu_k1.h"		/* K1  channeliser */
#include "vdl2gpu_dsp.h"		/* FIR + phase, fit error, Grey slicer, burst geometry */
#include "vdl2gpu_machine.h"	/* exact serial machine (clusters, resolver, fallback) */
#include "vdl2gpu_scan.h"	/* K2a scan: screens, survivors, probe / regions / verify */
#include "vdl2gpu_resolve.h"	/* K2s sort, K2b clusters, K2c resolve, K2f commit, K2d payload */
#include "vdl2gpu_k3.h"		/* K3 compact / rebase, per-push init, test hook */

#endif
