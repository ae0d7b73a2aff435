// Ceiling probe (not part of the product): what this MI355X sustains for the access patterns of
// the embedding hot path, so that bench.py reads its "random-line ceiling" from a measured file
// (profiles/r02_ceiling.txt) instead of a literal. One access = W contiguous bytes at a random
// W-aligned offset of a large buffer, moved by W/16 lanes with 16 B each.
//
//   hipcc -O3 --offload-arch=gfx950 scripts/exp_ceiling.hip -o scripts/bin/exp_ceiling
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __host__ inline uint64_t mix(uint64_t x) {
  x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL; x ^= x >> 27; x *= 0x94d049bb133111ebULL; x ^= x >> 31;
  return x;
}

// ---- A / D / G: single random reads of W bytes ------------------------------------------------
template <int W, int ILP>
__global__ void __launch_bounds__(256) k_read(const float4* __restrict__ tab, uint64_t units_mask,
                                              int64_t n, float4* __restrict__ out) {
  constexpr int L = W / 16;  // lanes per access
  const int l = threadIdx.x % L;
  const int64_t ngroups = (int64_t)gridDim.x * (256 / L);
  float acc = 0.f;
  for (int64_t i0 = ((int64_t)blockIdx.x * (256 / L) + threadIdx.x / L) * ILP; i0 < n; i0 += ngroups * ILP) {
    float4 v[ILP];
#pragma unroll
    for (int q = 0; q < ILP; ++q) v[q] = tab[(mix(i0 + q) & units_mask) * L + l];
#pragma unroll
    for (int q = 0; q < ILP; ++q) acc += v[q].x + v[q].w;
  }
  if (acc == 12345.678f) out[0] = make_float4(acc, 0, 0, 0);
}

// ---- B / C: a 64-B read, then a second 64-B read `delta` lines further inside the same aligned
// block of `blk` lines (blk = 0: an unrelated random line). DEP: the second address waits for the
// first line's data (probe -> row).
template <int ILP, int DEP>
__global__ void __launch_bounds__(256) k_pair(const float4* __restrict__ tab, uint64_t lines_mask,
                                              int64_t n, int blk, int delta, float4* __restrict__ out) {
  const int l = threadIdx.x & 3;
  const int64_t ngroups = (int64_t)gridDim.x * 64;
  float acc = 0.f;
  for (int64_t i0 = ((int64_t)blockIdx.x * 64 + threadIdx.x / 4) * ILP; i0 < n; i0 += ngroups * ILP) {
    float4 v[ILP], w[ILP];
    uint64_t line[ILP];
#pragma unroll
    for (int q = 0; q < ILP; ++q) {
      line[q] = mix(i0 + q) & lines_mask;
      v[q] = tab[line[q] * 4 + l];
    }
#pragma unroll
    for (int q = 0; q < ILP; ++q) {
      uint64_t dep = 0;
      if (DEP) dep = (uint64_t)(__float_as_uint(v[q].x) & 1u);  // table is zero-filled: always 0
      uint64_t l2;
      if (blk > 0) l2 = (line[q] & ~(uint64_t)(blk - 1)) | ((line[q] + delta + dep) & (uint64_t)(blk - 1));
      else l2 = (mix(line[q] ^ 0x1234567ull) + dep) & lines_mask;
      w[q] = tab[l2 * 4 + l];
    }
#pragma unroll
    for (int q = 0; q < ILP; ++q) acc += v[q].x + w[q].y;
  }
  if (acc == 12345.678f) out[0] = make_float4(acc, 0, 0, 0);
}

// ---- E: row update: read W bytes, write them back (+ optionally a 4-B stamp). TSMODE 0 none,
// 1 = 4-B write at a random place of a separate 4-B array, 2 = 4-B write inside the same 1-KiB
// aligned block as the row (offset 0 of the block).
template <int W, int ILP, int TSMODE>
__global__ void __launch_bounds__(256) k_rmw(float4* __restrict__ tab, uint64_t units_mask, int64_t n,
                                             uint32_t* __restrict__ ts, uint64_t ts_mask) {
  constexpr int L = W / 16;
  const int l = threadIdx.x % L;
  const int64_t ngroups = (int64_t)gridDim.x * (256 / L);
  for (int64_t i0 = ((int64_t)blockIdx.x * (256 / L) + threadIdx.x / L) * ILP; i0 < n; i0 += ngroups * ILP) {
    float4 v[ILP];
    uint64_t u[ILP];
#pragma unroll
    for (int q = 0; q < ILP; ++q) {
      // distinct units per access (no two groups update one unit): multiplicative permutation
      u[q] = ((uint64_t)(i0 + q) * 0x9E3779B97F4A7C15ull >> 20) & units_mask;
      v[q] = tab[u[q] * L + l];
    }
#pragma unroll
    for (int q = 0; q < ILP; ++q) {
      v[q].x += 1.f;
      tab[u[q] * L + l] = v[q];
      if (TSMODE == 1 && l == 0) ts[mix(u[q]) & ts_mask] = (uint32_t)i0;
      // 3 / 4 / 5: a 16- / 32- / 64-byte stamp record at a random place of the stamp array
      if (TSMODE == 3 && l == 0) reinterpret_cast<float4*>(ts)[mix(u[q]) & (ts_mask >> 2)] = v[q];
      if (TSMODE == 4 && l < 2) reinterpret_cast<float4*>(ts)[((mix(u[q]) & (ts_mask >> 3)) << 1) + l] = v[q];
      if (TSMODE == 5 && l < 4) reinterpret_cast<float4*>(ts)[((mix(u[q]) & (ts_mask >> 4)) << 2) + l] = v[q];
      if (TSMODE == 2 && l == 0) {
        uint64_t byte = (u[q] * W) & ~(uint64_t)1023;  // start of the row's 1-KiB block
        if (((u[q] * W) & 1023) == 0) byte += 512;      // not the row's own line
        reinterpret_cast<uint32_t*>(tab)[byte / 4 + 15] = (uint32_t)i0;
      }
    }
  }
}

// ---- F: sparse ascending sweep: access i reads the 64-B line i * stride + (mix(i) % stride):
// the order a batch sorted by slot would visit the table in.
template <int ILP>
__global__ void __launch_bounds__(256) k_sweep(const float4* __restrict__ tab, int64_t n, int stride,
                                               float4* __restrict__ out) {
  const int l = threadIdx.x & 3;
  const int64_t ngroups = (int64_t)gridDim.x * 64;
  float acc = 0.f;
  for (int64_t i0 = ((int64_t)blockIdx.x * 64 + threadIdx.x / 4) * ILP; i0 < n; i0 += ngroups * ILP) {
    float4 v[ILP];
#pragma unroll
    for (int q = 0; q < ILP; ++q) {
      const uint64_t i = (uint64_t)(i0 + q);
      v[q] = tab[(i * stride + mix(i) % (uint64_t)stride) * 4 + l];
    }
#pragma unroll
    for (int q = 0; q < ILP; ++q) acc += v[q].x;
  }
  if (acc == 12345.678f) out[0] = make_float4(acc, 0, 0, 0);
}

// ---- W: random 64-B line writes into a buffer (full lines)
template <int ILP>
__global__ void __launch_bounds__(256) k_write(float4* __restrict__ tab, uint64_t lines_mask, int64_t n) {
  const int l = threadIdx.x & 3;
  const int64_t ngroups = (int64_t)gridDim.x * 64;
  for (int64_t i0 = ((int64_t)blockIdx.x * 64 + threadIdx.x / 4) * ILP; i0 < n; i0 += ngroups * ILP) {
#pragma unroll
    for (int q = 0; q < ILP; ++q)
      tab[(((uint64_t)(i0 + q) * 0x9E3779B97F4A7C15ull >> 20) & lines_mask) * 4 + l] = make_float4(i0, q, 0, 0);
  }
}

static hipEvent_t e0, e1;
template <typename F>
static float timed(F f, int reps = 4) {
  f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) f();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}
static void report(const char* name, int64_t n, double bytes_per, float ms) {
  printf("%-64s n %9lld  %8.1f us  %6.2f G acc/s  %7.0f GB/s\n", name, (long long)n, ms * 1e3,
         n / ms / 1e6, n * bytes_per / ms / 1e6);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const bool only_stamp = argc > 1 && argv[1][0] == 's';
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  size_t free_b, total_b;
  CK(hipMemGetInfo(&free_b, &total_b));
  printf("device memory: %.1f GiB free of %.1f GiB\n", free_b / 1073741824.0, total_b / 1073741824.0);
  const uint64_t big = only_stamp ? 1ull << 35 : 1ull << 37;  // 128 GiB (32 GiB for the stamp rows only)
  float4 *tab, *out;
  CK(hipMalloc(&tab, big));
  CK(hipMemset(tab, 0, big));
  CK(hipMalloc(&out, 1 << 20));
  uint32_t* ts;
  const uint64_t ts_n = 1ull << 28;
  CK(hipMalloc(&ts, ts_n * 4));
  CK(hipMemset(ts, 0, ts_n * 4));
  const int64_t n = 1 << 23;
  const int grid = 8192;
  char nm[160];

  printf("\n[A] single random reads, 32 GiB footprint, by access width\n");
  if (!only_stamp) {
    const uint64_t fp = 1ull << 35;
    report("read 64 B  (ILP 8)", n, 64, timed([&] { k_read<64, 8><<<grid, 256>>>(tab, fp / 64 - 1, n, out); }));
    report("read 64 B  (ILP 16)", n, 64, timed([&] { k_read<64, 16><<<grid, 256>>>(tab, fp / 64 - 1, n, out); }));
    report("read 128 B (ILP 8)", n, 128, timed([&] { k_read<128, 8><<<grid, 256>>>(tab, fp / 128 - 1, n, out); }));
    report("read 256 B (ILP 4)", n, 256, timed([&] { k_read<256, 4><<<grid, 256>>>(tab, fp / 256 - 1, n, out); }));
    report("read 256 B (ILP 8)", n, 256, timed([&] { k_read<256, 8><<<grid, 256>>>(tab, fp / 256 - 1, n, out); }));
    report("read 512 B (ILP 4)", n, 512, timed([&] { k_read<512, 4><<<grid, 256>>>(tab, fp / 512 - 1, n, out); }));
    report("read 1024 B (ILP 4)", n / 2, 1024, timed([&] { k_read<1024, 4><<<grid, 256>>>(tab, fp / 1024 - 1, n / 2, out); }));
  }
  printf("\n[D] random 64-B reads by footprint\n");
  for (int lg : {27, 28, 30, 32, 34, 35, 37}) {
    if (only_stamp) break;
    const uint64_t fp = 1ull << lg;
    snprintf(nm, sizeof nm, "read 64 B over %6.2f GiB", fp / 1073741824.0);
    report(nm, n, 64, timed([&] { k_read<64, 8><<<grid, 256>>>(tab, fp / 64 - 1, n, out); }));
  }
  printf("\n[D2] random 128-B reads by footprint\n");
  for (int lg : {27, 30, 35, 37}) {
    if (only_stamp) break;
    const uint64_t fp = 1ull << lg;
    snprintf(nm, sizeof nm, "read 128 B over %6.2f GiB", fp / 1073741824.0);
    report(nm, n, 128, timed([&] { k_read<128, 8><<<grid, 256>>>(tab, fp / 128 - 1, n, out); }));
  }
  printf("\n[B] pairs of 64-B reads (32 GiB): second line inside the same aligned block\n");
  if (!only_stamp) {
    const uint64_t lm = (1ull << 35) / 64 - 1;
    struct { int blk, delta; } cs[] = {{2, 1}, {4, 1}, {4, 2}, {8, 4}, {16, 8}, {16, 1}, {32, 16}, {64, 32},
                                       {128, 64}, {1024, 512}, {0, 0}};
    for (auto c : cs) {
      snprintf(nm, sizeof nm, "pair: 2nd line +%d in a %d-B block%s", c.delta, c.blk * 64, c.blk ? "" : " (unrelated random)");
      report(nm, n / 2, 128, timed([&] { k_pair<8, 0><<<grid, 256>>>(tab, lm, n / 2, c.blk, c.delta, out); }));
    }
    printf("\n[C] the same, second address dependent on the first line's data\n");
    for (auto c : cs) {
      if (c.blk != 2 && c.blk != 16 && c.blk != 64 && c.blk != 0) continue;
      snprintf(nm, sizeof nm, "dep pair: 2nd line +%d in a %d-B block%s", c.delta, c.blk * 64, c.blk ? "" : " (unrelated random)");
      report(nm, n / 2, 128, timed([&] { k_pair<8, 1><<<grid, 256>>>(tab, lm, n / 2, c.blk, c.delta, out); }));
    }
  }
  printf("\n[E] row update (read W, write W back) over 32 GiB, 1M / 4M distinct rows\n");
  for (int64_t m : {(int64_t)1 << 20, (int64_t)1 << 22}) {
    const uint64_t fp = 1ull << 35;
    report("rmw 64 B", m, 128, timed([&] { k_rmw<64, 4, 0><<<grid, 256>>>(tab, fp / 64 - 1, m, ts, ts_n - 1); }));
    report("rmw 128 B", m, 256, timed([&] { k_rmw<128, 4, 0><<<grid, 256>>>(tab, fp / 128 - 1, m, ts, ts_n - 1); }));
    report("rmw 128 B + 4-B stamp, random place of a 1-GiB array", m, 260, timed([&] { k_rmw<128, 4, 1><<<grid, 256>>>(tab, fp / 128 - 1, m, ts, ts_n - 1); }));
    report("rmw 128 B + 4-B stamp in the row's 1-KiB block", m, 260, timed([&] { k_rmw<128, 4, 2><<<grid, 256>>>(tab, fp / 128 - 1, m, ts, ts_n - 1); }));
    report("rmw 128 B + 16-B stamp record, random place of a 1-GiB array", m, 272, timed([&] { k_rmw<128, 4, 3><<<grid, 256>>>(tab, fp / 128 - 1, m, ts, ts_n - 1); }));
    report("rmw 128 B + 32-B stamp record, random place of a 1-GiB array", m, 288, timed([&] { k_rmw<128, 4, 4><<<grid, 256>>>(tab, fp / 128 - 1, m, ts, ts_n - 1); }));
    report("rmw 128 B + 64-B stamp record, random place of a 1-GiB array", m, 320, timed([&] { k_rmw<128, 4, 5><<<grid, 256>>>(tab, fp / 128 - 1, m, ts, ts_n - 1); }));
    report("rmw 256 B", m, 512, timed([&] { k_rmw<256, 4, 0><<<grid, 256>>>(tab, fp / 256 - 1, m, ts, ts_n - 1); }));
    report("rmw 512 B", m, 1024, timed([&] { k_rmw<512, 2, 0><<<grid, 256>>>(tab, fp / 512 - 1, m, ts, ts_n - 1); }));
  }
  printf("\n[F] ascending sparse sweep of 64-B lines (one line per `stride` lines), 32 GiB span at most\n");
  for (int stride : {2, 8, 20, 64, 256}) {
    if (only_stamp) break;
    int64_t m = n;
    while ((uint64_t)m * stride * 64 > (1ull << 35)) m >>= 1;
    snprintf(nm, sizeof nm, "sweep: one 64-B line per %d lines", stride);
    report(nm, m, 64, timed([&] { k_sweep<8><<<grid, 256>>>(tab, m, stride, out); }));
  }
  printf("\n[G] random 64-B reads / writes over a 109 MB buffer (gradient rows, lookup output)\n");
  {
    const uint64_t lm = (1ull << 27) / 64 - 1;  // 128 MiB
    report("read 64 B over 128 MiB", n, 64, timed([&] { k_read<64, 8><<<grid, 256>>>(tab, lm, n, out); }));
    report("write 64 B lines over 128 MiB", n, 64, timed([&] { k_write<8><<<grid, 256>>>(tab, lm, n); }));
    report("write 64 B lines over 32 GiB", n, 64, timed([&] { k_write<8><<<grid, 256>>>(tab, (1ull << 35) / 64 - 1, n); }));
  }
  return 0;
}
