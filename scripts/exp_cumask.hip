// Experiment (not part of the product): which CUs does a stream created with
// hipExtStreamCreateWithCUMask run on? One wave per workgroup records XCC_ID and HW_ID; the host
// prints how many distinct CUs per XCC the mask gave. Mask layouts tried: the low 24 bits, and the
// set that is balanced under both plausible bit orders (bit = cu * 8 + xcc or bit = xcc * 32 + cu).
// Outcome: profiles/r03_cumask.txt (the mapping works; the streams made the step slower and
// rocprofv3 crash, so the table does not use them).
//   hipcc -O3 --offload-arch=gfx950 scripts/exp_cumask.hip -o scripts/bin/exp_cumask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <set>
#include <map>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void k_where(uint32_t* out, int spin) {
  uint32_t xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  // stay resident for a while so that the grid spreads over every CU the queue may use
  long long t0 = clock64();
  while (clock64() - t0 < spin) {}
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = xcc;
    out[2 * blockIdx.x + 1] = hw;
  }
}

static void run(const char* name, hipStream_t st, uint32_t* d, int nwg) {
  CK(hipMemsetAsync(d, 0xFF, nwg * 8, st));
  hipLaunchKernelGGL(k_where, dim3(nwg), dim3(1024), 0, st, d, 200000);
  CK(hipGetLastError());
  CK(hipStreamSynchronize(st));
  std::vector<uint32_t> h(2 * nwg);
  CK(hipMemcpy(h.data(), d, nwg * 8, hipMemcpyDeviceToHost));
  std::map<uint32_t, std::set<uint32_t>> per_xcc;
  for (int i = 0; i < nwg; ++i) {
    const uint32_t xcc = h[2 * i] & 0xF, hw = h[2 * i + 1];
    const uint32_t cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    per_xcc[xcc].insert(se << 8 | sh << 4 | cu);
  }
  printf("%-46s", name);
  int total = 0;
  for (auto& kv : per_xcc) {
    printf(" xcc%u:%2zu", kv.first, kv.second.size());
    total += (int)kv.second.size();
  }
  printf("  total %d CUs\n", total);
}

int main() {
  uint32_t* d;
  const int nwg = 4096;
  CK(hipMalloc(&d, nwg * 8));
  hipStream_t s0;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  run("no mask", s0, d, nwg);
  {
    uint32_t m[8] = {0x00FFFFFFu, 0, 0, 0, 0, 0, 0, 0};
    hipStream_t s;
    CK(hipExtStreamCreateWithCUMask(&s, 8, m));
    run("low 24 bits", s, d, nwg);
    CK(hipStreamDestroy(s));
  }
  {
    uint32_t m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int x = 0; x < 8; ++x)
      for (int j = 0; j < 3; ++j) {
        const int b = 32 * x + 8 * j + ((x + j) % 8);
        m[b / 32] |= 1u << (b % 32);
      }
    hipStream_t s;
    CK(hipExtStreamCreateWithCUMask(&s, 8, m));
    run("balanced 24 (3 per xcc under either order)", s, d, nwg);
    uint32_t c[8];
    for (int i = 0; i < 8; ++i) c[i] = ~m[i];
    hipStream_t s2;
    CK(hipExtStreamCreateWithCUMask(&s2, 8, c));
    run("its complement (232)", s2, d, nwg);
    CK(hipStreamDestroy(s));
    CK(hipStreamDestroy(s2));
  }
  {
    uint32_t m[8] = {0xFFFFFFFFu, 0, 0, 0, 0, 0, 0, 0};
    hipStream_t s;
    CK(hipExtStreamCreateWithCUMask(&s, 8, m));
    run("bits 0..31", s, d, nwg);
    CK(hipStreamDestroy(s));
  }
  printf("done\n");
  return 0;
}
