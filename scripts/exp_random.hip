// Ceiling probe (not part of the product): random 64-byte line reads / writes over a large table
// with `ILP` independent requests in flight per 4-lane group.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
__device__ inline uint64_t mix(uint64_t x){x^=x>>30;x*=0xbf58476d1ce4e5b9ULL;x^=x>>27;x*=0x94d049bb133111ebULL;x^=x>>31;return x;}
template<int ILP, int MODE>   // MODE 0 read-only (sum), 1 read random + write streaming, 2 write random
__global__ void __launch_bounds__(256) k(const float4* __restrict__ tab, uint64_t lines_mask, int64_t n, float4* __restrict__ out, float4* __restrict__ wtab){
  const int l4 = threadIdx.x & 3;
  const int64_t ngroups = (int64_t)gridDim.x * 64;
  float4 acc = make_float4(0,0,0,0);
  for (int64_t i0 = ((int64_t)blockIdx.x*64 + threadIdx.x/4)*ILP; i0 < n; i0 += ngroups*ILP){
    float4 v[ILP];
#pragma unroll
    for(int q=0;q<ILP;++q){ uint64_t line = mix(i0+q) & lines_mask; if (MODE!=2) v[q] = tab[line*4 + l4]; else v[q]=make_float4(i0,q,0,0); }
#pragma unroll
    for(int q=0;q<ILP;++q){
      if (MODE==0){ acc.x+=v[q].x; acc.y+=v[q].y; }
      else if (MODE==1) out[(i0+q)*4 + l4] = v[q];
      else { uint64_t line = mix(i0+q) & lines_mask; wtab[line*4+l4] = v[q]; }
    }
  }
  if (MODE==0 && acc.x==12345.678f) out[0]=acc;
}
template<int ILP,int MODE> void run(const float4* tab, uint64_t mask, int64_t n, float4* out, float4* wtab, const char* name){
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int grid = (int)((n/ILP + 63)/64); if (grid > 4096) grid = 4096;
  k<ILP,MODE><<<grid,256>>>(tab,mask,n,out,wtab); hipDeviceSynchronize();
  hipEventRecord(e0); for(int r=0;r<5;++r) k<ILP,MODE><<<grid,256>>>(tab,mask,n,out,wtab); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms,e0,e1); ms/=5;
  printf("%-28s ILP %d n %lld: %.1f us  %.1f G lines/s  %.0f GB/s random side\n", name, ILP, (long long)n, ms*1e3, n/ms/1e6, n*64.0/ms/1e6);
}
int main(){
  const uint64_t lines = 1ull<<29;   // 32 GiB table
  float4 *tab,*out; hipMalloc(&tab, lines*64); hipMemset(tab, 0, lines*64);
  const int64_t n = 1<<21; hipMalloc(&out, n*64);
  run<1,0>(tab,lines-1,n,out,tab,"random read"); run<4,0>(tab,lines-1,n,out,tab,"random read"); run<8,0>(tab,lines-1,n,out,tab,"random read"); run<16,0>(tab,lines-1,n,out,tab,"random read");
  run<1,1>(tab,lines-1,n,out,tab,"random read + stream write"); run<4,1>(tab,lines-1,n,out,tab,"random read + stream write"); run<8,1>(tab,lines-1,n,out,tab,"random read + stream write");
  run<1,2>(tab,lines-1,n,out,tab,"random write"); run<4,2>(tab,lines-1,n,out,tab,"random write"); run<8,2>(tab,lines-1,n,out,tab,"random write");
  const int64_t n2 = 1<<24;
  hipFree(out); hipMalloc(&out, n2*64);
  run<8,0>(tab,lines-1,n2,out,tab,"random read (16M)"); run<8,1>(tab,lines-1,n2,out,tab,"random read + stream write (16M)"); run<8,2>(tab,lines-1,n2,out,tab,"random write (16M)");
  return 0;
}
