// writevalue.hip -- what zeroing ONE 8-byte word in stream order costs: hipMemsetAsync (a fill kernel) against
// hipStreamWriteValue64 (a command-processor packet), each in front of a kernel that takes ~130 us, 200 steps back to back.
//   hipcc -O2 --offload-arch=gfx950 -o tools/ubench/writevalue tools/ubench/writevalue.hip && tools/ubench/writevalue
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void busy(unsigned long long *p, unsigned long long ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
  if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(p, 1ull);
}
int main() {
  unsigned long long *d = nullptr;
  hipMalloc(&d, 4096);
  hipMemset(d, 0xff, 4096);
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  const int steps = 200;
  for (int mode = 0; mode < 3; mode++) {
    hipError_t first = hipSuccess;
    for (int rep = 0; rep < 2; rep++) {
      hipStreamSynchronize(st);
      const double t0 = now();
      for (int i = 0; i < steps; i++) {
        hipError_t e = hipSuccess;
        if (mode == 1) e = hipMemsetAsync(d + 8, 0, 8, st);
        if (mode == 2) e = hipStreamWriteValue64(st, d + 8, 0ull, 0);
        if (e != hipSuccess && first == hipSuccess) first = e;
        busy<<<1536, 256, 0, st>>>(d + 8, 13000);      // 130 us
      }
      hipStreamSynchronize(st);
      const double dt = (now() - t0) / steps * 1e3;
      if (rep) {
        unsigned long long v = 0;
        hipMemcpy(&v, d + 8, 8, hipMemcpyDeviceToHost);
        printf("%-24s %.4f ms per step (word afterwards: %llu; first error: %s)\n", mode == 0 ? "kernel alone" : mode == 1 ? "hipMemsetAsync(8) + kernel" : "hipStreamWriteValue64 + kernel",
               dt, v, hipGetErrorString(first));
      }
    }
  }
  return 0;
}
