// Microbenchmark: DEPENDENT random-line chains, the access shape of backward search.  Each group of
// LPG lanes walks its own chain: load one 16*LPG-byte granule (one dwordx4 per lane), reduce it
// across the group (popcount + DPP-style shuffle adds), derive the next granule from the result.
// CH independent chains per group are kept in flight.  Reports granule requests/s at full occupancy:
// the ceiling a search kernel with that geometry can reach when it does no other work.
//   hipcc --offload-arch=gfx950 -O3 chain.hip -o chain && ./chain [table GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ inline uint64_t mix(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

template <int LPG>
__device__ inline uint32_t group_sum(uint32_t v) {
#pragma unroll
  for (int d = 1; d < LPG; d <<= 1) v += __shfl_xor(v, d, 64);
  return v;
}

template <int LPG, int CH>
__global__ __launch_bounds__(256) void k_chain(const uint4 *__restrict__ tab, uint64_t ngran, uint32_t steps,
                                                uint32_t *__restrict__ out, uint64_t seed) {
  const uint32_t t = threadIdx.x % LPG;
  const uint64_t grp = ((uint64_t)blockIdx.x * 256 + threadIdx.x) / LPG;
  uint64_t g[CH];
#pragma unroll
  for (int u = 0; u < CH; u++) g[u] = mix(seed + grp * CH + u) % ngran;
  uint32_t acc = 0;
  for (uint32_t s = 0; s < steps; s++) {
    uint4 w[CH];
#pragma unroll
    for (int u = 0; u < CH; u++) w[u] = tab[g[u] * LPG + t];
#pragma unroll
    for (int u = 0; u < CH; u++) {
      uint32_t p = __builtin_popcount(w[u].x) + __builtin_popcount(w[u].y) + __builtin_popcount(w[u].z) + __builtin_popcount(w[u].w);
      p = group_sum<LPG>(p);
      acc += p;
      g[u] = mix(g[u] + p) % ngran;
    }
  }
  if (acc == 0xFFFFFFFFu) out[0] = acc;
}

template <int LPG, int CH>
void run(const uint4 *tab, uint64_t bytes, uint32_t *out, int blocks_per_cu) {
  const uint64_t ngran = bytes / (16 * LPG);
  const int grid = 256 * blocks_per_cu;
  const uint64_t groups = (uint64_t)grid * 256 / LPG;
  const uint32_t steps = (uint32_t)((1ull << 25) / (groups * CH)) + 1;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9;
  for (int r = 0; r < 4; r++) {
    CK(hipEventRecord(a));
    k_chain<LPG, CH><<<grid, 256>>>(tab, ngran, steps, out, 1234 + r);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); if (r && ms < best) best = ms;
  }
  double n = (double)steps * groups * CH;
  printf("granule %3d B  chains/group %d  blocks/CU %d  in flight %7.0f : %7.3f ms  %6.2f G req/s  %6.2f TB/s  step %5.0f ns\n",
         16 * LPG, CH, blocks_per_cu, (double)groups * CH, best, n / best / 1e6, n * 16 * LPG / best / 1e9, best * 1e6 / steps);
}

// One lane = one chain, each step reads a whole 64-byte line with four dwordx4 loads (no lane group): the shape a
// frontier kernel with one element per lane would have if it did its rank queries in the element's own lane.
template <int CH>
__global__ __launch_bounds__(256) void k_chain_lane64(const uint4 *__restrict__ tab, uint64_t nline, uint32_t steps,
                                                       uint32_t *__restrict__ out, uint64_t seed) {
  const uint64_t id = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  uint64_t g[CH];
#pragma unroll
  for (int u = 0; u < CH; u++) g[u] = mix(seed + id * CH + u) % nline;
  uint32_t acc = 0;
  for (uint32_t s = 0; s < steps; s++) {
    uint4 w[CH][4];
#pragma unroll
    for (int u = 0; u < CH; u++)
#pragma unroll
      for (int j = 0; j < 4; j++) w[u][j] = tab[g[u] * 4 + j];
#pragma unroll
    for (int u = 0; u < CH; u++) {
      uint32_t p = 0;
#pragma unroll
      for (int j = 0; j < 4; j++)
        p += __builtin_popcount(w[u][j].x) + __builtin_popcount(w[u][j].y) + __builtin_popcount(w[u][j].z) + __builtin_popcount(w[u][j].w);
      acc += p;
      g[u] = mix(g[u] + p) % nline;
    }
  }
  if (acc == 0xFFFFFFFFu) out[0] = acc;
}

template <int CH>
void run_lane64(const uint4 *tab, uint64_t bytes, uint32_t *out, int blocks_per_cu) {
  const uint64_t nline = bytes / 64;
  const int grid = 256 * blocks_per_cu;
  const uint64_t chains = (uint64_t)grid * 256 * CH;
  const uint32_t steps = (uint32_t)((1ull << 25) / chains) + 1;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9;
  for (int r = 0; r < 4; r++) {
    CK(hipEventRecord(a));
    k_chain_lane64<CH><<<grid, 256>>>(tab, nline, steps, out, 99 + r);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); if (r && ms < best) best = ms;
  }
  double n = (double)steps * chains;
  printf("lane-owned 64 B line  chains/lane %d  blocks/CU %d  in flight %7.0f : %7.3f ms  %6.2f G lines/s  %6.2f TB/s  step %5.0f ns\n",
         CH, blocks_per_cu, (double)chains, best, n / best / 1e6, n * 64 / best / 1e9, best * 1e6 / steps);
}

int main(int argc, char **argv) {
  double gib = argc > 1 ? atof(argv[1]) : 64.0;
  uint64_t bytes = (uint64_t)(gib * (1ull << 30));
  uint4 *tab; uint32_t *out;
  CK(hipMalloc(&tab, bytes)); CK(hipMalloc(&out, 4));
  CK(hipMemset(tab, 0x5A, bytes));
  CK(hipDeviceSynchronize());
  printf("table %.1f GiB\n", gib);
  run<8, 1>(tab, bytes, out, 4);  run<8, 1>(tab, bytes, out, 8);  run<8, 2>(tab, bytes, out, 8);  run<8, 4>(tab, bytes, out, 8);
  run<4, 1>(tab, bytes, out, 4);  run<4, 1>(tab, bytes, out, 8);  run<4, 2>(tab, bytes, out, 8);  run<4, 4>(tab, bytes, out, 8);
  run<2, 1>(tab, bytes, out, 8);  run<2, 2>(tab, bytes, out, 8);
  run<1, 1>(tab, bytes, out, 8);  run<1, 2>(tab, bytes, out, 8);
  run_lane64<1>(tab, bytes, out, 2);  run_lane64<1>(tab, bytes, out, 4);  run_lane64<1>(tab, bytes, out, 8);  run_lane64<2>(tab, bytes, out, 4);
  return 0;
}
