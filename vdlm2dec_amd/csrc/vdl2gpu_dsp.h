/* vdl2gpu_dsp.h -- DSP pieces shared by the scan, the serial machine and the payload decoder.  Part of the device side of libvdl2gpu.so; included by vdl2gpu_kernels.h only. */
#ifndef VDL2GPU_DSP_H
#define VDL2GPU_DSP_H

/* ============================================================ shared DSP pieces */
__device__ __forceinline__ float d_tab(const uint32_t *t, int i)
{
	return __uint_as_float(t[i]);
}

/* filteredphase(), d8psk.c:219-230: x points at sample n-16.  All 17 samples of the
 * ring are fetched up front (independent loads, one memory latency); the taps
 * mflt[tap0], mflt[tap0+4], .. < 65 are then applied oldest sample first, exactly the
 * reference's accumulation order. */
template <int R> __device__ __forceinline__ float k2_fir_phase_r(const float2 *x)
{
	float2 v[17];
#pragma unroll
	for (int j = 0; j < 17; ++j)
		v[j] = x[j];
	float sr = 0.0f, si = 0.0f;
#pragma unroll
	for (int j = 0; j < 17; ++j) {
		if (R + 4 * j < 65) {
			const float m = d_tab(c_mflt, R + 4 * j);
			sr += v[j].x * m;
			si += v[j].y * m;
		}
	}
	return vdl2_atan2f(si, sr);
}

__device__ __forceinline__ float k2_fir_phase(const float2 *x, int tap0)
{
	switch (tap0) {
	case 0: return k2_fir_phase_r<0>(x);
	case 1: return k2_fir_phase_r<1>(x);
	case 2: return k2_fir_phase_r<2>(x);
	case 3: return k2_fir_phase_r<3>(x);
	default: break;
	}
	/* trigger instant: clk = (int)roundf(of) in [4,12] -> 16..14 taps (d8psk.c:305-306) */
	float2 v[17];
#pragma unroll
	for (int j = 0; j < 17; ++j)
		v[j] = x[j];
	float sr = 0.0f, si = 0.0f;
#pragma unroll
	for (int j = 0; j < 17; ++j) {
		const int i = tap0 + 4 * j;
		if (i < 65) {
			const float m = d_tab(c_mflt, i);
			sr += v[j].x * m;
			si += v[j].y * m;
		}
	}
	return vdl2_atan2f(si, sr);
}

/* The same with the table-driven atan2f (vdl2_math.h: the same result bits, no data-dependent branches) and the taps from an
 * LDS copy of mflt[] (72 floats, zero padded): for the payload decode, where 256 lanes take a symbol each. */
__device__ __forceinline__ float k2_fir_phase_tab(const float2 *x, int tap0, const float *smf, const float *atab)
{
	float2 v[17];
#pragma unroll
	for (int j = 0; j < 17; ++j)
		v[j] = x[j];
	float sr = 0.0f, si = 0.0f;
#pragma unroll
	for (int j = 0; j < 16; ++j) {	/* tap0 <= 3: sixteen taps always exist */
		const float m = smf[tap0 + 4 * j];
		sr += v[j].x * m;
		si += v[j].y * m;
	}
	if (tap0 == 0) {	/* mflt[64] exists only for tap0 == 0 */
		const float m = smf[64];
		sr += v[16].x * m;
		si += v[16].y * m;
	}
	return vdl2_atan2f_tab(si, sr, atab);
}

/* d8psk.c:257-289: ph[0], ph[STRIDE], ... ph[16*STRIDE] are the 17 phases one symbol apart */
/* The reference compares the float phase step with the DOUBLE constants +-M_PI.  M_PI lies strictly
 * between the adjacent floats 0x40490fda (3.14159250) and 0x40490fdb (3.14159274), so for a float x
 *     (double)x > M_PI   <=>  x > 0x40490fda      and      (double)x < -M_PI  <=>  x < -0x40490fda
 * and the comparison can be made in float without changing a single decision. */
#define VDL2_PI_BELOW 0x40490fdau

template <int STRIDE> __device__ __forceinline__ float k2_sync_metric(const float *ph, float *slope)
{
	const float pi_lo = __uint_as_float(VDL2_PI_BELOW);
	float pr[17];
	double pud = 0.0;	/* Pu: every update goes float -> double -> float like `Pu -= 2 * M_PI`; */
	float pu = 0.0f;	/* kept in both forms so that only the narrowing is paid per step */
	float pv = ph[0] - d_tab(c_sw, 0);
	float mean = pv;
	pr[0] = pv;
#pragma unroll
	for (int l = 1; l < 17; ++l) {
		const float pc = ph[STRIDE * l] - d_tab(c_sw, l);
		const float pd = pc - pv;
		pv = pc;
		/* -1 / 0 / +1 turns; k * 2pi is exact in double, so pu + k*2pi is the reference's sum */
		const float k = (pd > pi_lo) ? -1.0f : ((pd < -pi_lo) ? 1.0f : 0.0f);
		pu = (float)(pud + (double)k * (2 * M_PI));
		pud = (double)pu;
		pr[l] = pc + pu;
		mean += pr[l];
	}
	mean /= 17.0f;
	float fr = 0.0f;
#pragma unroll
	for (int l = 0; l < 17; ++l) {
		pr[l] -= mean;
		fr += pr[l] * (float)(l - 8);
	}
	fr /= 408.0f;
	float err = 0.0f;
#pragma unroll
	for (int l = 0; l < 17; ++l) {
		const float e = pr[l] - (float)(l - 8) * fr;
		err += e * e;
	}
	*slope = fr;
	return err;
}

/* differential slice of one symbol -> Grey table index (d8psk.c:213, 323-327) */
__device__ __forceinline__ int k2_grey_index(float p, float pprev, float df)
{
	float d = (p - pprev) - df;
	if ((double)d > M_PI)
		d = (float)((double)d - 2 * M_PI);
	if ((double)d < -M_PI)
		d = (float)((double)d + 2 * M_PI);
	int i = (int)roundf((float)(128.0 * (double)d / M_PI + 128.0));
	return i < 0 ? 0 : (i > 256 ? 256 : i);
}

__device__ __forceinline__ float k2_soft_bit(int idx, int which, int pnbit, const float *grey = nullptr)
{
	const float v = grey ? grey[which * 257 + idx] : d_tab(which == 0 ? c_grey1 : (which == 1 ? c_grey2 : c_grey3), idx);
	return pnbit ? (float)(1.0 - (double)v) : v;	/* descrambler, d8psk.c:60-63 */
}

#endif
