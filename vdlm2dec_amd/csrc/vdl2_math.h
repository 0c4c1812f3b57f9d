/*
 * vdl2_math.h -- float atan2 with a FIXED operation sequence.
 *
 * The reference takes the phase of every FIR output with libm's cargf/atan2f
 * (d8psk.c:229).  All downstream decisions (sync trigger, timing, Grey index)
 * hang on those floats, so the device must return the same bits as the host
 * libm the reference links (glibc 2.35 on this image: the classic fdlibm
 * single-precision algorithm -- argument reduction to one of four breakpoints
 * followed by an 11-term odd polynomial, all in float, no FMA).
 *
 * This header restates that published algorithm with explicit float
 * operations only (+ - * / compare, bit tests).  Compiled with
 * -ffp-contract=off on host or device it is bit-identical to glibc's atan2f;
 * tests/test_math.py checks that exhaustively-at-random on the CPU
 * (>10^8 points incl. axes, signed zeros, huge/tiny ratios) and
 * tests/test_gpu_math.py re-checks the device build against it.
 *
 * Usable from C, C++ and HIP device code.
 */
#ifndef VDL2_MATH_H
#define VDL2_MATH_H
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__) || defined(__HIP_DEVICE_COMPILE__)
#define VDL2_HD __host__ __device__ static inline
#else
#define VDL2_HD static inline
#endif

VDL2_HD uint32_t vdl2_f2u(float f)
{
	uint32_t u;
#if defined(__HIP_DEVICE_COMPILE__)
	u = __float_as_uint(f);
#else
	memcpy(&u, &f, 4);
#endif
	return u;
}

VDL2_HD float vdl2_u2f(uint32_t u)
{
	float f;
#if defined(__HIP_DEVICE_COMPILE__)
	f = __uint_as_float(u);
#else
	memcpy(&f, &u, 4);
#endif
	return f;
}

/* atan(x) for finite or infinite x (NaN is not produced by the pipeline).
 * Written without data-dependent branches (selects only) so that a 64-lane
 * wavefront does not serialise the four argument-reduction ranges; every
 * selected expression is the reference's, so the result bits are unchanged:
 * the unreduced range uses hi = lo = 0 and den = 1 (x/1 == x,
 * 0 - ((t*s - 0) - t) == t - t*s exactly), and sign symmetry is exact. */
VDL2_HD float vdl2_atanf(float x)
{
	/* breakpoints atan(0.5), atan(1), atan(1.5), atan(inf): hi + lo parts */
	const float hi0 = vdl2_u2f(0x3eed6338u), lo0 = vdl2_u2f(0x31ac3769u);
	const float hi1 = vdl2_u2f(0x3f490fdau), lo1 = vdl2_u2f(0x33222168u);
	const float hi2 = vdl2_u2f(0x3f7b985eu), lo2 = vdl2_u2f(0x33140fb4u);
	const float hi3 = vdl2_u2f(0x3fc90fdau), lo3 = vdl2_u2f(0x33a22168u);
	/* odd polynomial coefficients */
	const float a0 = vdl2_u2f(0x3eaaaaabu), a1 = vdl2_u2f(0xbe4ccccdu);
	const float a2 = vdl2_u2f(0x3e124925u), a3 = vdl2_u2f(0xbde38e38u);
	const float a4 = vdl2_u2f(0x3dba2e6eu), a5 = vdl2_u2f(0xbd9d8795u);
	const float a6 = vdl2_u2f(0x3d886b35u), a7 = vdl2_u2f(0xbd6ef16bu);
	const float a8 = vdl2_u2f(0x3d4bda59u), a9 = vdl2_u2f(0xbd15a221u);
	const float a10 = vdl2_u2f(0x3c8569d7u);

	const uint32_t hx = vdl2_f2u(x);
	const uint32_t ix = hx & 0x7fffffffu;
	const float ax = vdl2_u2f(ix);
	const int c0 = ix < 0x3ee00000u;	/* |x| < 7/16: no reduction */
	const int c1 = ix < 0x3f300000u;	/* < 11/16 */
	const int c2 = ix < 0x3f980000u;	/* < 19/16 */
	const int c3 = ix < 0x401c0000u;	/* < 39/16 */
	const float num = c0 ? ax : (c1 ? (2.0f * ax - 1.0f) : (c2 ? (ax - 1.0f) : (c3 ? (ax - 1.5f) : -1.0f)));
	const float den = c0 ? 1.0f : (c1 ? (2.0f + ax) : (c2 ? (ax + 1.0f) : (c3 ? (1.0f + 1.5f * ax) : ax)));
	const float hi = c0 ? 0.0f : (c1 ? hi0 : (c2 ? hi1 : (c3 ? hi2 : hi3)));
	const float lo = c0 ? 0.0f : (c1 ? lo0 : (c2 ? lo1 : (c3 ? lo2 : lo3)));
	const float t = num / den;
	const float z = t * t;
	const float w = z * z;
	const float s1 = z * (a0 + w * (a2 + w * (a4 + w * (a6 + w * (a8 + w * a10)))));
	const float s2 = w * (a1 + w * (a3 + w * (a5 + w * (a7 + w * a9))));
	float r = hi - ((t * (s1 + s2) - lo) - t);
	r = (ix >= 0x4c000000u) ? (hi3 + lo3) : r;	/* |x| >= 2^25: +-pi/2 */
	return vdl2_u2f(vdl2_f2u(r) ^ (hx & 0x80000000u));
}

VDL2_HD float vdl2_atan2f(float y, float x)
{
	const float pi = vdl2_u2f(0x40490fdbu);
	const float pi_lo = vdl2_u2f(0xb3bbbd2eu);
	const float pi_o_2 = vdl2_u2f(0x3fc90fdbu);
	const uint32_t hx = vdl2_f2u(x), hy = vdl2_f2u(y);
	const uint32_t ix = hx & 0x7fffffffu, iy = hy & 0x7fffffffu;
	const int m = (int)((hy >> 31) & 1u) | (int)((hx >> 30) & 2u);
	float z;

	if (hx == 0x3f800000u)	/* x == 1 */
		return vdl2_atanf(y);
	if (iy == 0) {		/* y == +-0 */
		if (m < 2)
			return y;
		return (m == 2) ? pi : -pi;
	}
	if (ix == 0)		/* x == +-0 */
		return (hy >> 31) ? -pi_o_2 : pi_o_2;
	if (ix == 0x7f800000u) {	/* x infinite */
		if (iy == 0x7f800000u) {
			const float q = vdl2_u2f(0x3f490fdbu);	/* pi/4 */
			switch (m) {
			case 0: return q;
			case 1: return -q;
			case 2: return 3.0f * q;
			default: return -3.0f * q;
			}
		}
		switch (m) {
		case 0: return 0.0f;
		case 1: return -0.0f;
		case 2: return pi;
		default: return -pi;
		}
	}
	if (iy == 0x7f800000u)
		return (hy >> 31) ? -pi_o_2 : pi_o_2;
	{
		const int k = ((int)iy - (int)ix) >> 23;
		if (k > 60)
			z = pi_o_2 + 0.5f * pi_lo;
		else if ((hx >> 31) && k < -60)
			z = 0.0f;
		else {
			const float q = y / x;
			z = vdl2_atanf(vdl2_u2f(vdl2_f2u(q) & 0x7fffffffu));
		}
	}
	switch (m) {
	case 0: return z;
	case 1: return vdl2_u2f(vdl2_f2u(z) ^ 0x80000000u);
	case 2: return pi - (z - pi_lo);
	default: return (z - pi_lo) - pi;
	}
}

/* ---- table-driven form for wide SIMD ------------------------------------------------
 * Same result bits as vdl2_atan2f(), but with no data-dependent control flow on the common
 * path: the four argument-reduction ranges of atanf differ only in five constants
 * (num = |q|*A + B, den = |q|*C + A, hi, lo -- each product/sum below is the very operation
 * the range's formula performs, or an exact identity such as |q|*1, |q|*0 + 1, x + (-y)),
 * so a lane looks its row up in a 5 x 8 float table (LDS on the device) instead of
 * branching.  Zero / infinite arguments take one rarely-entered branch.
 * tests/ctests/atan2_check.c checks this form against libm too. */
#define VDL2_ATAN_ROWS 5
#define VDL2_ATAN_STRIDE 8
VDL2_HD float vdl2_atan_tab_entry(int i)	/* i = row * VDL2_ATAN_STRIDE + column */
{
	const uint32_t t[VDL2_ATAN_ROWS][VDL2_ATAN_STRIDE] = {
		/* A           B            C            hi           lo */
		{0x3f800000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0, 0, 0},	/* |q| < 7/16 : t = |q| / 1 */
		{0x40000000u, 0xbf800000u, 0x3f800000u, 0x3eed6338u, 0x31ac3769u, 0, 0, 0},	/* < 11/16: (2q-1)/(2+q) */
		{0x3f800000u, 0xbf800000u, 0x3f800000u, 0x3f490fdau, 0x33222168u, 0, 0, 0},	/* < 19/16: (q-1)/(q+1) */
		{0x3f800000u, 0xbfc00000u, 0x3fc00000u, 0x3f7b985eu, 0x33140fb4u, 0, 0, 0},	/* < 39/16: (q-1.5)/(1+1.5q) */
		{0x00000000u, 0xbf800000u, 0x3f800000u, 0x3fc90fdau, 0x33a22168u, 0, 0, 0},	/* else   : -1/q */
	};
	return vdl2_u2f(t[i / VDL2_ATAN_STRIDE][i % VDL2_ATAN_STRIDE]);
}

VDL2_HD float vdl2_atan2f_tab(float y, float x, const float *tab)
{
	const float pi = vdl2_u2f(0x40490fdbu);
	const float pi_lo = vdl2_u2f(0xb3bbbd2eu);
	const float pi_o_2 = vdl2_u2f(0x3fc90fdbu);
	const float a0 = vdl2_u2f(0x3eaaaaabu), a1 = vdl2_u2f(0xbe4ccccdu);
	const float a2 = vdl2_u2f(0x3e124925u), a3 = vdl2_u2f(0xbde38e38u);
	const float a4 = vdl2_u2f(0x3dba2e6eu), a5 = vdl2_u2f(0xbd9d8795u);
	const float a6 = vdl2_u2f(0x3d886b35u), a7 = vdl2_u2f(0xbd6ef16bu);
	const float a8 = vdl2_u2f(0x3d4bda59u), a9 = vdl2_u2f(0xbd15a221u);
	const float a10 = vdl2_u2f(0x3c8569d7u);
	const uint32_t hx = vdl2_f2u(x), hy = vdl2_f2u(y);
	const uint32_t ix = hx & 0x7fffffffu, iy = hy & 0x7fffffffu;
	const uint32_t sy = hy & 0x80000000u;
	const int xneg = (int)(hx >> 31);

	/* atanf(|y/x|) */
	const uint32_t iq = vdl2_f2u(y / x) & 0x7fffffffu;
	const float aq = vdl2_u2f(iq);
	const int row = (int)(iq >= 0x3ee00000u) + (int)(iq >= 0x3f300000u) + (int)(iq >= 0x3f980000u) + (int)(iq >= 0x401c0000u);
	const float *c = tab + row * VDL2_ATAN_STRIDE;
	const float cA = c[0], cB = c[1], cC = c[2], hi = c[3], lo = c[4];
	const float t = (aq * cA + cB) / (aq * cC + cA);
	const float z2 = t * t;
	const float w = z2 * z2;
	const float s1 = z2 * (a0 + w * (a2 + w * (a4 + w * (a6 + w * (a8 + w * a10)))));
	const float s2 = w * (a1 + w * (a3 + w * (a5 + w * (a7 + w * a9))));
	float z = hi - ((t * (s1 + s2) - lo) - t);
	z = (iq >= 0x4c000000u) ? pi_o_2 : z;	/* atanhi[3] + atanlo[3] rounds to pi/2 */

	/* quadrant (e_atan2f.c): exponent-difference shortcuts, then m = 2*sign(x) + sign(y) */
	const int k = ((int)iy - (int)ix) >> 23;
	z = (k > 60) ? pi_o_2 : z;		/* pi_o_2 + 0.5f * pi_lo == pi_o_2 */
	z = (xneg && k < -60) ? 0.0f : z;
	const float zr = pi - (z - pi_lo);	/* m = 2; m = 3 is its exact negation */
	float r = vdl2_u2f(vdl2_f2u(xneg ? zr : z) ^ sy);

	const uint32_t mn = ix < iy ? ix : iy, mx = ix < iy ? iy : ix;
	if (mn == 0 || mx >= 0x7f800000u) {	/* a zero or an infinity (NaN is not produced by the pipeline) */
		if (iy == 0)
			r = xneg ? vdl2_u2f(0x40490fdbu | sy) : y;
		else if (ix == 0 || ix != 0x7f800000u)
			r = vdl2_u2f(0x3fc90fdbu | sy);
		else if (iy == 0x7f800000u)
			r = vdl2_u2f((xneg ? 0x4016cbe4u : 0x3f490fdbu) | sy);	/* 3pi/4, pi/4 */
		else
			r = vdl2_u2f((xneg ? 0x40490fdbu : 0u) | sy);
	}
	return r;
}

#endif /* VDL2_MATH_H */
