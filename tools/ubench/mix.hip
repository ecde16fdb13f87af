// Microbenchmark: dependent chains with k_search4's REQUEST MIX (round 4; VERDICT r3 "what binds k_search4").
//
// chain.hip measures one kind of request -- a 64-byte block of one table, 16 bytes per lane of a quad -- and gives
// 51-54 G requests/s.  The search kernel at C3 asks four kinds of four tables (fmx_search.hip):
//   D : a 64-byte block of the rank dictionary (77 GiB), 16 bytes per lane of the quad          (general / one-row step)
//   J : a 16-byte entry of the row jump table (64 GiB), the four lanes of the quad the same address    (8 steps)
//   R : an 8-byte word of the three-step row table (32 GiB), lane 0 of the quad only                    (3 steps)
//   K : a 16-byte entry of the k-mer table (4 GiB), the four lanes the same address                     (first 4 steps)
// and a pattern is the chain K D D R J J J.  This program walks such chains, 16 per wave in lockstep like the
// kernel's lane groups, with nothing else on the chain (no pattern bytes, no LDS table, no popcounts to speak of), from
//   sep : four hipMalloc allocations (what the library does)
//   one : one hipMalloc carved into the four tables
//   vmm : one hipMemAddressReserve range, physical memory from hipMemCreate mapped at the recommended granularity
// so that the request rate of the MIX is known apart from the kernel's own work.
//   hipcc --offload-arch=gfx950 -O3 mix.hip -o mix && ./mix [D J R K GiB] [programs ...]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e__)); exit(1); } } while (0)

__device__ inline uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

struct Tables {
  const uint8_t *base[4];      // D, J, R, K
  uint64_t entries[4];         // 64-, 16-, 8-, 16-byte entries
};
struct Program {
  uint32_t len;
  uint8_t kind[16];            // 0 D, 1 J, 2 R, 3 K
};

__global__ __launch_bounds__(256) void k_mix(Tables tb, Program pg, uint32_t rounds, uint32_t *__restrict__ out, uint64_t seed) {
  const uint32_t t = threadIdx.x & 3u;
  const uint64_t grp = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 2;
  uint64_t state = mix64(seed + grp);
  uint32_t acc = 0;
  for (uint32_t r = 0; r < rounds; r++) {
    for (uint32_t s = 0; s < pg.len; s++) {              // wave-uniform
      const uint32_t kd = pg.kind[s];
      const uint64_t e = state % tb.entries[kd];
      uint32_t v;
      if (kd == 0) {
        const uint4 w = *reinterpret_cast<const uint4 *>(tb.base[0] + e * 64 + t * 16);
        v = __builtin_popcount(w.x) + __builtin_popcount(w.y) + __builtin_popcount(w.z) + __builtin_popcount(w.w);
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
      } else if (kd == 2) {
        unsigned long long w = 0;
        if (t == 0) w = *reinterpret_cast<const unsigned long long *>(tb.base[2] + e * 8);
        v = (uint32_t)w ^ (uint32_t)(w >> 32);
        v = __shfl(v, (threadIdx.x & 63u) & ~3u, 64);
      } else {
        const uint4 w = *reinterpret_cast<const uint4 *>(tb.base[kd] + e * 16);
        v = w.x ^ w.y ^ w.z ^ w.w;
      }
      acc += v;
      state = mix64(state + v);
    }
  }
  if (acc == 0xFFFFFFFFu) out[0] = acc;
}

static Program parse(const char *s) {
  Program p{};
  for (; *s && p.len < 16; s++) {
    const char *k = strchr("DJRK", *s);
    if (!k) { printf("bad program character %c\n", *s); exit(1); }
    p.kind[p.len++] = (uint8_t)(k - "DJRK");
  }
  return p;
}

static void run(const Tables &tb, const char *prog, int blocks_per_cu, uint32_t *out, const char *mode) {
  const Program pg = parse(prog);
  const int grid = 256 * blocks_per_cu;
  const uint64_t groups = (uint64_t)grid * 64;
  const uint32_t rounds = (uint32_t)((1ull << 26) / (groups * pg.len)) + 1;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9;
  for (int r = 0; r < 4; r++) {
    CK(hipEventRecord(a));
    k_mix<<<grid, 256>>>(tb, pg, rounds, out, 4321 + r);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); if (r && ms < best) best = ms;
  }
  const double n = (double)rounds * pg.len * groups;
  printf("%-4s program %-10s blocks/CU %d  chains %7.0f : %8.3f ms  %6.2f G req/s  trip %5.0f ns\n", mode, prog, blocks_per_cu,
         (double)groups, best, n / best / 1e6, best * 1e6 / (rounds * pg.len));
  fflush(stdout);
}

int main(int argc, char **argv) {
  double gib[4] = {77, 64, 32, 4};
  int ai = 1;
  for (int i = 0; i < 4 && ai < argc && (argv[ai][0] >= '0' && argv[ai][0] <= '9'); i++, ai++) gib[i] = atof(argv[ai]);
  std::vector<std::string> progs;
  for (; ai < argc; ai++) progs.push_back(argv[ai]);
  if (progs.empty()) progs = {"D", "J", "R", "K", "KDDRJJJ", "DRJJJ", "JJJ", "DJ", "DK"};
  const uint64_t esz[4] = {64, 16, 8, 16};
  uint64_t bytes[4], total = 0;
  for (int i = 0; i < 4; i++) { bytes[i] = (uint64_t)(gib[i] * (1ull << 30)) & ~((2ull << 20) - 1); total += bytes[i]; }
  printf("tables D %.0f J %.0f R %.0f K %.0f GiB (total %.0f GiB)\n", gib[0], gib[1], gib[2], gib[3], total / (double)(1ull << 30));
  uint32_t *out; CK(hipMalloc(&out, 4));
  const char *only = getenv("MIX_MODE");
  // ---- sep
  if (!only || !strcmp(only, "sep")) {
    Tables tb{};
    void *p[4];
    for (int i = 0; i < 4; i++) { CK(hipMalloc(&p[i], bytes[i])); CK(hipMemset(p[i], 0x5A, bytes[i])); tb.base[i] = (const uint8_t *)p[i]; tb.entries[i] = bytes[i] / esz[i]; }
    CK(hipDeviceSynchronize());
    for (const auto &pr : progs) { run(tb, pr.c_str(), 8, out, "sep"); }
    for (const auto &pr : progs) if (pr.size() > 1 || pr == "D") { run(tb, pr.c_str(), 7, out, "sep"); run(tb, pr.c_str(), 4, out, "sep"); run(tb, pr.c_str(), 2, out, "sep"); }
    for (int i = 0; i < 4; i++) CK(hipFree(p[i]));
  }
  // ---- one
  if (!only || !strcmp(only, "one")) {
    Tables tb{};
    void *p;
    CK(hipMalloc(&p, total)); CK(hipMemset(p, 0x5A, total)); CK(hipDeviceSynchronize());
    uint64_t o = 0;
    for (int i = 0; i < 4; i++) { tb.base[i] = (const uint8_t *)p + o; tb.entries[i] = bytes[i] / esz[i]; o += bytes[i]; }
    for (const auto &pr : progs) run(tb, pr.c_str(), 8, out, "one");
    CK(hipFree(p));
  }
  // ---- vmm
  if (!only || !strcmp(only, "vmm")) {
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gmin = 0, grec = 0;
    hipError_t e = hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum);
    if (e == hipSuccess) e = hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended);
    printf("vmm granularity: minimum %zu recommended %zu (%s)\n", gmin, grec, hipGetErrorString(e));
    if (e == hipSuccess && grec) {
      const uint64_t chunk = std::max<uint64_t>(grec, 1ull << 30);       // 1 GiB physical chunks
      const uint64_t tot = (total + chunk - 1) / chunk * chunk;
      void *va = nullptr;
      e = hipMemAddressReserve(&va, tot, chunk, nullptr, 0);
      printf("reserve %.0f GiB aligned to %.0f MiB: %s\n", tot / (double)(1ull << 30), chunk / (double)(1 << 20), hipGetErrorString(e));
      std::vector<hipMemGenericAllocationHandle_t> hs;
      bool ok = e == hipSuccess;
      for (uint64_t o = 0; ok && o < tot; o += chunk) {
        hipMemGenericAllocationHandle_t h;
        e = hipMemCreate(&h, chunk, &prop, 0);
        if (e == hipSuccess) e = hipMemMap((char *)va + o, chunk, 0, h, 0);
        if (e != hipSuccess) { printf("vmm map at %.0f GiB: %s\n", o / (double)(1ull << 30), hipGetErrorString(e)); ok = false; break; }
        hs.push_back(h);
      }
      if (ok) {
        hipMemAccessDesc ad{};
        ad.location = prop.location;
        ad.flags = hipMemAccessFlagsProtReadWrite;
        e = hipMemSetAccess(va, tot, &ad, 1);
        if (e != hipSuccess) { printf("hipMemSetAccess: %s\n", hipGetErrorString(e)); ok = false; }
      }
      if (ok) {
        CK(hipMemset(va, 0x5A, total)); CK(hipDeviceSynchronize());
        Tables tb{};
        uint64_t o = 0;
        for (int i = 0; i < 4; i++) { tb.base[i] = (const uint8_t *)va + o; tb.entries[i] = bytes[i] / esz[i]; o += bytes[i]; }
        for (const auto &pr : progs) run(tb, pr.c_str(), 8, out, "vmm");
      }
    }
  }
  return 0;
}
