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

/* atan(x) for finite or infinite x (NaN is not produced by the pipeline). */
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
	const int neg = (int)(hx >> 31);
	float hi, lo;
	int reduced = 1;

	if (ix >= 0x4c000000u) {	/* |x| >= 2^25: atan = +-pi/2 */
		float r = hi3 + lo3;
		return neg ? -r : r;
	}
	if (ix < 0x3ee00000u) {	/* |x| < 7/16: no reduction */
		if (ix < 0x31000000u)	/* |x| < 2^-29: atan(x) == x in float */
			return x;
		reduced = 0;
		hi = lo = 0.0f;
	} else {
		float ax = vdl2_u2f(ix);
		if (ix < 0x3f980000u) {	/* |x| < 19/16 */
			if (ix < 0x3f300000u) {	/* 7/16 <= |x| < 11/16 */
				hi = hi0;
				lo = lo0;
				x = (2.0f * ax - 1.0f) / (2.0f + ax);
			} else {
				hi = hi1;
				lo = lo1;
				x = (ax - 1.0f) / (ax + 1.0f);
			}
		} else if (ix < 0x401c0000u) {	/* |x| < 39/16 */
			hi = hi2;
			lo = lo2;
			x = (ax - 1.5f) / (1.0f + 1.5f * ax);
		} else {
			hi = hi3;
			lo = lo3;
			x = -1.0f / ax;
		}
	}
	{
		const float z = x * x;
		const float w = z * z;
		const float s1 = z * (a0 + w * (a2 + w * (a4 + w * (a6 + w * (a8 + w * a10)))));
		const float s2 = w * (a1 + w * (a3 + w * (a5 + w * (a7 + w * a9))));
		if (!reduced)
			return x - x * (s1 + s2);
		{
			const float r = hi - ((x * (s1 + s2) - lo) - x);
			return neg ? -r : r;
		}
	}
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

#endif /* VDL2_MATH_H */
