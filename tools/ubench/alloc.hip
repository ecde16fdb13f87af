// alloc.hip -- what device memory costs to GET on this box, by API, piece size and number of host threads.
// Background (profiles/r04_alloc_time.txt): hipMalloc takes 20-30 ms per GiB here, so the 164 GiB of derived tables of a
// C3-size index cost 3-5 s to allocate and 1 s to fill.  Questions: is the cost per byte whatever the API (hipMalloc, the
// virtual-memory API, the stream-ordered pool)?  Does it overlap across host threads?  Does a pool hand memory back for free?
//
//   hipcc -O2 --offload-arch=gfx950 -o tools/ubench/alloc tools/ubench/alloc.hip -lpthread && tools/ubench/alloc [GiB=64]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); } } while (0)

__global__ void touch(unsigned long long *p, size_t words, size_t stride) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i * stride < words; i += (size_t)gridDim.x * blockDim.x) p[i * stride] = i;
}

int main(int argc, char **argv) {
  const size_t G = 1ull << 30;
  const size_t total = (argc > 1 ? (size_t)atoi(argv[1]) : 64) * G;
  CK(hipSetDevice(0));
  CK(hipFree(nullptr));
  size_t fr = 0, tot = 0;
  CK(hipMemGetInfo(&fr, &tot));
  printf("device: %.1f GiB free of %.1f; experiment size %.0f GiB\n", fr / 1073741824.0, tot / 1073741824.0, total / 1073741824.0);

  // 1. one hipMalloc, twice (is memory the process has just given back cheaper?)
  for (int rep = 0; rep < 2; rep++) {
    void *p = nullptr;
    double t0 = now();
    CK(hipMalloc(&p, total));
    double t1 = now();
    touch<<<1024, 256>>>((unsigned long long *)p, total / 8, 512);      // one word per 4 KiB: is anything deferred to first touch?
    CK(hipDeviceSynchronize());
    double t2 = now();
    CK(hipFree(p));
    double t3 = now();
    printf("hipMalloc(%zu GiB) #%d: alloc %.3f s (%.1f ms/GiB), first touch %.3f s, free %.3f s\n", total / G, rep, t1 - t0, (t1 - t0) * 1e3 / (total / G), t2 - t1, t3 - t2);
  }
  // 2. the same bytes as 1 GiB hipMallocs: one thread, then T threads at once
  for (int T : {1, 4, 8, 16}) {
    const size_t pieces = total / G;
    std::vector<void *> ps(pieces, nullptr);
    double t0 = now();
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
      th.emplace_back([&, t]() {
        CK(hipSetDevice(0));
        for (size_t i = t; i < pieces; i += T) CK(hipMalloc(&ps[i], G));
      });
    for (auto &x : th) x.join();
    double t1 = now();
    for (void *p : ps) CK(hipFree(p));
    double t2 = now();
    printf("hipMalloc 1 GiB x %zu on %2d host threads: %.3f s (%.1f ms/GiB), free %.3f s\n", pieces, T, t1 - t0, (t1 - t0) * 1e3 / pieces, t2 - t1);
  }
  // 3. the virtual-memory API: one reserved range, physical chunks created and mapped by T threads
  {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf("vmm: recommended granularity %zu KiB\n", gran >> 10);
    for (size_t chunk : {G, (size_t)256 << 20}) {
      for (int T : {1, 8}) {
        void *va = nullptr;
        double t0 = now();
        CK(hipMemAddressReserve(&va, total, chunk, nullptr, 0));
        double t1 = now();
        const size_t pieces = total / chunk;
        std::vector<hipMemGenericAllocationHandle_t> hs(pieces);
        std::vector<double> tc(T, 0.0), tm(T, 0.0);
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++)
          th.emplace_back([&, t]() {
            CK(hipSetDevice(0));
            for (size_t i = t; i < pieces; i += T) {
              double a = now();
              CK(hipMemCreate(&hs[i], chunk, &prop, 0));
              double b = now();
              CK(hipMemMap((char *)va + i * chunk, chunk, 0, hs[i], 0));
              double c = now();
              tc[t] += b - a;
              tm[t] += c - b;
            }
          });
        for (auto &x : th) x.join();
        double t2 = now();
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        CK(hipMemSetAccess(va, total, &acc, 1));
        double t3 = now();
        touch<<<1024, 256>>>((unsigned long long *)va, total / 8, 512);
        CK(hipDeviceSynchronize());
        double t4 = now();
        for (size_t i = 0; i < pieces; i++) { CK(hipMemUnmap((char *)va + i * chunk, chunk)); CK(hipMemRelease(hs[i])); }
        CK(hipMemAddressFree(va, total));
        double t5 = now();
        printf("vmm %4zu MiB chunks, %d threads: reserve %.3f, create+map %.3f s (create %.3f + map %.3f per thread), set access %.3f, first touch %.3f, teardown %.3f: total %.3f s (%.1f ms/GiB)\n",
               chunk >> 20, T, t1 - t0, t2 - t1, tc[0], tm[0], t3 - t2, t4 - t3, t5 - t4, t3 - t0, (t3 - t0) * 1e3 / (total / G));
      }
    }
  }
  // 4. the stream-ordered pool: first allocation, then again after a free that the pool keeps
  {
    hipMemPool_t pool = nullptr;
    CK(hipDeviceGetDefaultMemPool(&pool, 0));
    unsigned long long keep = ~0ull;
    CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    for (int rep = 0; rep < 3; rep++) {
      void *p = nullptr;
      double t0 = now();
      CK(hipMallocAsync(&p, total, st));
      CK(hipStreamSynchronize(st));
      double t1 = now();
      touch<<<1024, 256, 0, st>>>((unsigned long long *)p, total / 8, 512);
      CK(hipStreamSynchronize(st));
      double t2 = now();
      CK(hipFreeAsync(p, st));
      CK(hipStreamSynchronize(st));
      double t3 = now();
      printf("hipMallocAsync(%zu GiB) #%d (pool keeps what is freed): alloc %.3f s, first touch %.3f s, free %.3f s\n", total / G, rep, t1 - t0, t2 - t1, t3 - t2);
    }
    CK(hipMemPoolTrimTo(pool, 0));
    CK(hipStreamDestroy(st));
  }
  // 5. allocation beside a running kernel: does hipMalloc wait for the device, or the device for it?
  {
    void *busy = nullptr;
    CK(hipMalloc(&busy, 8 * G));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    double t0 = now();
    for (int i = 0; i < 40; i++) touch<<<4096, 256, 0, st>>>((unsigned long long *)busy, 8 * G / 8, 1);      // ~8 GiB of stores per launch
    void *p = nullptr;
    double t1 = now();
    CK(hipMalloc(&p, 32 * G));
    double t2 = now();
    CK(hipStreamSynchronize(st));
    double t3 = now();
    printf("hipMalloc(32 GiB) while 40 x 8 GiB store kernels run on another stream: malloc %.3f s; kernels done %.3f s after their launch (alone: see below)\n", t2 - t1, t3 - t0);
    CK(hipFree(p));
    t0 = now();
    for (int i = 0; i < 40; i++) touch<<<4096, 256, 0, st>>>((unsigned long long *)busy, 8 * G / 8, 1);
    CK(hipStreamSynchronize(st));
    printf("the 40 kernels alone: %.3f s\n", now() - t0);
    CK(hipFree(busy));
    CK(hipStreamDestroy(st));
  }
  return 0;
}
