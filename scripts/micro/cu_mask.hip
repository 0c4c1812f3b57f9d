// micro: which CUs does a stream created with hipExtStreamCreateWithCUMask use?  (mask bit -> XCC / SE / CU mapping on gfx950)
// build: hipcc --offload-arch=gfx950 -O2 scripts/micro/cu_mask.hip -o scripts/micro/cu_mask.bin ; run: cu_mask.bin <bits cleared from the bottom> [first cleared bit]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>
__global__ void where(unsigned *out, int spin)
{
	unsigned hw, xcc;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
	if (threadIdx.x == 0)
		out[blockIdx.x] = (hw & 0xffffu) | ((xcc & 0xfu) << 16);
	long long t0 = clock64();
	while (clock64() - t0 < spin) { }
}
int main(int argc, char **argv)
{
	const int nclear = argc > 1 ? atoi(argv[1]) : 8, first = argc > 2 ? atoi(argv[2]) : 0;
	hipDeviceProp_t prop;
	hipGetDeviceProperties(&prop, 0);
	const int ncu = prop.multiProcessorCount;
	std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
	for (int i = 0; i < ncu; ++i)
		if (i < first || i >= first + nclear)
			mask[i / 32] |= 1u << (i % 32);
	hipStream_t st;
	hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data());
	printf("ncu %d, mask words %zu, create: %s\n", ncu, mask.size(), hipGetErrorString(e));
	const int nb = ncu * 16;
	unsigned *d, *h = (unsigned *)malloc(nb * sizeof(unsigned));
	hipMalloc(&d, nb * sizeof(unsigned));
	for (int pass = 0; pass < 2; ++pass) {
		hipStream_t s = pass ? st : 0;
		hipMemsetAsync(d, 0xff, nb * sizeof(unsigned), s);
		hipLaunchKernelGGL(where, dim3(nb), dim3(256), 0, s, d, 20000);
		hipStreamSynchronize(s);
		hipMemcpy(h, d, nb * sizeof(unsigned), hipMemcpyDeviceToHost);
		std::map<unsigned, std::set<unsigned>> per_xcc;
		for (int i = 0; i < nb; ++i) {
			const unsigned xcc = h[i] >> 16, cu = (h[i] >> 8) & 0xff;	/* cu_id | sh_id << 4 | se_id << 5 */
			per_xcc[xcc].insert(cu);
		}
		size_t tot = 0;
		printf("%s stream:", pass ? "masked" : "null");
		for (auto &kv : per_xcc) {
			printf(" xcc%u:%zu", kv.first, kv.second.size());
			tot += kv.second.size();
		}
		printf("  total CUs %zu\n", tot);
	}
	return 0;
}
