// Ceiling probe 2 (not part of the product): does a second 64-B line near the first one come
// cheaper than a second fully random line?  pairs/s for offsets of 1 line, 8 lines (512 B),
// 15 lines (<1 KB), 32 lines (2 KB), 1024 lines (64 KB), random.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ inline uint64_t mix(uint64_t x){x^=x>>30;x*=0xbf58476d1ce4e5b9ULL;x^=x>>27;x*=0x94d049bb133111ebULL;x^=x>>31;return x;}
template<int ILP>
__global__ void __launch_bounds__(256) k(const float4* __restrict__ tab, uint64_t lines_mask, int64_t n, int64_t delta, float4* __restrict__ out){
  const int l4 = threadIdx.x & 3;
  const int64_t ngroups = (int64_t)gridDim.x * 64;
  float4 acc = make_float4(0,0,0,0);
  for (int64_t i0 = ((int64_t)blockIdx.x*64 + threadIdx.x/4)*ILP; i0 < n; i0 += ngroups*ILP){
    float4 v[ILP], w[ILP];
#pragma unroll
    for(int q=0;q<ILP;++q){
      uint64_t line = mix(i0+q) & lines_mask;
      uint64_t l2 = delta >= 0 ? ((line & ~(uint64_t)2047) | ((line + delta) & 2047)) : (mix(line) & lines_mask);
      if (delta > 0 && delta < 32) l2 = (line & ~(uint64_t)31) | ((line + delta) & 31);
      v[q] = tab[line*4 + l4]; w[q] = tab[l2*4 + l4]; }
#pragma unroll
    for(int q=0;q<ILP;++q){ acc.x+=v[q].x+w[q].y; }
  }
  if (acc.x==12345.678f) out[0]=acc;
}
int main(){
  const uint64_t lines = 1ull<<29;
  float4 *tab,*out; hipMalloc(&tab, lines*64); hipMemset(tab, 0, lines*64); hipMalloc(&out, 64);
  const int64_t n = 1<<21;
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int64_t delta : {0, 1, 2, 8, 15, 31, 64, 1024, -1}) {
    k<4><<<4096,256>>>(tab,lines-1,n,delta,out); hipDeviceSynchronize();
    hipEventRecord(e0); for(int r=0;r<5;++r) k<4><<<4096,256>>>(tab,lines-1,n,delta,out); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms,e0,e1); ms/=5;
    printf("second line at +%lld lines: %.1f us  %.1f G pairs/s\n", (long long)delta, ms*1e3, n/ms/1e6);
  }
  return 0;
}
