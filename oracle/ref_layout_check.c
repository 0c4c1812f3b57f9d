/*
 * ref_layout_check.c -- TEST INFRASTRUCTURE ONLY: compiled (never linked or run) by `make -C oracle ref`.
 *
 * vdl2gpu_burst_to_msgblk() (include/vdl2gpu.h) writes the reference's msgblk_t (vdlm2.h:39-47) through byte offsets
 * that are compile-time constants of the library.  This translation unit includes the REFERENCE'S OWN vdlm2.h and holds
 * every one of those constants against offsetof()/sizeof there: if the reference's struct, or the ABI of the machine the
 * drop-in is built on, ever disagrees, the build of oracle/_ref fails here instead of a field landing in the wrong place.
 */
#define _GNU_SOURCE
#include <stddef.h>
#include <stdio.h>
#include <pthread.h>
#include <complex.h>
#include <sys/time.h>
#include "vdlm2.h"
#include "vdl2gpu.h"

_Static_assert(offsetof(msgblk_t, chn) == VDL2GPU_MSGBLK_OFF_CHN, "msgblk_t.chn");
_Static_assert(offsetof(msgblk_t, Fr) == VDL2GPU_MSGBLK_OFF_FR, "msgblk_t.Fr");
_Static_assert(offsetof(msgblk_t, tv) == VDL2GPU_MSGBLK_OFF_TV, "msgblk_t.tv");
_Static_assert(offsetof(msgblk_t, ppm) == VDL2GPU_MSGBLK_OFF_PPM, "msgblk_t.ppm");
_Static_assert(offsetof(msgblk_t, nbrow) == VDL2GPU_MSGBLK_OFF_NBROW, "msgblk_t.nbrow");
_Static_assert(offsetof(msgblk_t, nlbyte) == VDL2GPU_MSGBLK_OFF_NLBYTE, "msgblk_t.nlbyte");
_Static_assert(offsetof(msgblk_t, data) == VDL2GPU_MSGBLK_OFF_DATA, "msgblk_t.data");
_Static_assert(sizeof(msgblk_t) == VDL2GPU_MSGBLK_SIZE, "sizeof(msgblk_t)");
_Static_assert(sizeof(((msgblk_t *)0)->chn) == 4 && sizeof(((msgblk_t *)0)->Fr) == 4 && sizeof(((msgblk_t *)0)->ppm) == 4 &&
	       sizeof(((msgblk_t *)0)->nbrow) == 4 && sizeof(((msgblk_t *)0)->nlbyte) == 4, "msgblk_t field widths");
_Static_assert(sizeof(((msgblk_t *)0)->data[0]) == VDL2GPU_ROWLEN && sizeof(((msgblk_t *)0)->data) >= VDL2GPU_MAXROWS * VDL2GPU_ROWLEN,
	       "msgblk_t.data rows");
/* thread_param_t (vdlm2.h:49-52) == vdl2gpu_chan_t: dropin/vdl2gpu_rcv.c copies field by field, this keeps them the same shape */
_Static_assert(sizeof(thread_param_t) == sizeof(vdl2gpu_chan_t) && offsetof(thread_param_t, Fr) == offsetof(vdl2gpu_chan_t, Fr) &&
	       offsetof(thread_param_t, Fo) == offsetof(vdl2gpu_chan_t, Fo), "thread_param_t");
_Static_assert(MAXNBCHANNELS == VDL2GPU_MAXCH, "MAXNBCHANNELS");
