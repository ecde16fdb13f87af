// Microbenchmark: random gather of G-byte granules (G = 32/64/128/256) from a large table,
// each granule fetched whole by G/16 adjacent lanes with one global_load_dwordx4 per lane.
// Measures the random-line ceiling the rank dictionary can hope for, per granule size.
//   hipcc --offload-arch=gfx950 -O3 gather.hip -o gather && ./gather [table GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ inline uint64_t mix(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// LPG lanes per granule; ILP independent granules per lane group per trip
template <int LPG, int ILP>
__global__ __launch_bounds__(256) void k_gather(const uint4 *__restrict__ tab, uint64_t ngran, uint64_t per_group,
                                                 uint32_t *__restrict__ out, uint64_t seed) {
  const uint32_t t = threadIdx.x % LPG;
  const uint64_t grp = ((uint64_t)blockIdx.x * 256 + threadIdx.x) / LPG;
  uint32_t acc = 0;
  for (uint64_t it = 0; it < per_group; it += ILP) {
    uint4 w[ILP];
#pragma unroll
    for (int u = 0; u < ILP; u++) {
      uint64_t g = mix(seed + grp * per_group + it + u) % ngran;
      w[u] = tab[g * LPG + t];
    }
#pragma unroll
    for (int u = 0; u < ILP; u++) acc += __builtin_popcount(w[u].x) + __builtin_popcount(w[u].y) + __builtin_popcount(w[u].z) + __builtin_popcount(w[u].w);
  }
  if (acc == 0xFFFFFFFFu) out[0] = acc;
}

template <int LPG, int ILP>
void run(const uint4 *tab, uint64_t bytes, uint32_t *out, int blocks_per_cu) {
  const uint64_t ngran = bytes / (16 * LPG);
  const int grid = 256 * blocks_per_cu;
  const uint64_t groups = (uint64_t)grid * 256 / LPG;
  const uint64_t total = 1ull << 25;                 // granules fetched per launch
  uint64_t per_group = total / groups; per_group -= per_group % ILP; if (per_group < ILP) per_group = ILP;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9;
  for (int r = 0; r < 4; r++) {
    CK(hipEventRecord(a));
    k_gather<LPG, ILP><<<grid, 256>>>(tab, ngran, per_group, out, 1234 + r);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); if (r && ms < best) best = ms;
  }
  double n = (double)per_group * groups;
  printf("granule %3d B  ilp %d  blocks/CU %d : %7.3f ms  %6.2f G granules/s  %6.2f TB/s\n", 16 * LPG, ILP, blocks_per_cu,
         best, n / best / 1e6, n * 16 * LPG / best / 1e9);
}

// One lane per query, two 16-byte pieces of the same random 128-byte line (header piece + one of the
// seven chunk pieces): the access shape of the per-lane rank primitive.
template <int ILP>
__global__ __launch_bounds__(256) void k_lane2(const uint4 *__restrict__ tab, uint64_t nlines, uint64_t per_lane,
                                                uint32_t *__restrict__ out, uint64_t seed) {
  const uint64_t lane = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t acc = 0;
  for (uint64_t it = 0; it < per_lane; it += ILP) {
    uint4 a[ILP], b[ILP];
#pragma unroll
    for (int u = 0; u < ILP; u++) {
      uint64_t h = mix(seed + lane * per_lane + it + u);
      uint64_t g = h % nlines;
      uint32_t q = 1 + (uint32_t)(h >> 40) % 7;
      a[u] = tab[g * 8];
      b[u] = tab[g * 8 + q];
    }
#pragma unroll
    for (int u = 0; u < ILP; u++) acc += __builtin_popcount(a[u].x ^ b[u].y) + __builtin_popcount(a[u].z ^ b[u].w);
  }
  if (acc == 0xFFFFFFFFu) out[0] = acc;
}

template <int ILP>
void run_lane2(const uint4 *tab, uint64_t bytes, uint32_t *out) {
  const uint64_t nlines = bytes / 128;
  const int grid = 256 * 8;
  const uint64_t lanes = (uint64_t)grid * 256;
  uint64_t per_lane = (1ull << 26) / lanes; per_lane -= per_lane % ILP; if (per_lane < ILP) per_lane = ILP;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9;
  for (int r = 0; r < 4; r++) {
    CK(hipEventRecord(a));
    k_lane2<ILP><<<grid, 256>>>(tab, nlines, per_lane, out, 99 + r);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); if (r && ms < best) best = ms;
  }
  double n = (double)per_lane * lanes;
  printf("per-lane 2x16B in one 128-B line, ilp %d : %7.3f ms  %6.2f G lines/s  (%6.2f TB/s if whole lines move)\n", ILP, best,
         n / best / 1e6, n * 128 / best / 1e9);
}

int main(int argc, char **argv) {
  double gib = argc > 1 ? atof(argv[1]) : 64.0;
  uint64_t bytes = (uint64_t)(gib * (1ull << 30));
  uint4 *tab; uint32_t *out;
  CK(hipMalloc(&tab, bytes)); CK(hipMalloc(&out, 4));
  CK(hipMemset(tab, 0x5A, bytes));
  CK(hipDeviceSynchronize());
  printf("table %.1f GiB\n", gib);
  run_lane2<1>(tab, bytes, out); run_lane2<2>(tab, bytes, out); run_lane2<4>(tab, bytes, out);
  run<1, 1>(tab, bytes, out, 8);  run<1, 2>(tab, bytes, out, 8);  run<1, 4>(tab, bytes, out, 8);
  run<2, 1>(tab, bytes, out, 8);  run<2, 2>(tab, bytes, out, 8);  run<2, 4>(tab, bytes, out, 8);
  run<4, 1>(tab, bytes, out, 8);  run<4, 2>(tab, bytes, out, 8);  run<4, 4>(tab, bytes, out, 8);
  run<8, 1>(tab, bytes, out, 8);  run<8, 2>(tab, bytes, out, 8);  run<8, 4>(tab, bytes, out, 8);
  run<16, 1>(tab, bytes, out, 8); run<16, 2>(tab, bytes, out, 8); run<16, 4>(tab, bytes, out, 8);
  return 0;
}
