// What device memory costs to get on this box: hipMalloc of fresh memory, of memory this process freed before, and through a
// stream-ordered pool that keeps what is freed.  (a one-go file job pays its context's allocations once.)
//   hipcc --offload-arch=gfx950 -O2 scripts/alloc_probe.hip -o /tmp/alloc_probe && /tmp/alloc_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
  hipFree(nullptr);
  size_t fr = 0, tot = 0; hipMemGetInfo(&fr, &tot);
  printf("free %.1f GB of %.1f\n", fr / 1073741824.0, tot / 1073741824.0);
  const size_t G = 1ull << 30;
  for (int round = 0; round < 3; round++) {
    std::vector<void *> p;
    double t_all = now();
    for (int i = 0; i < 12; i++) {
      void *q = nullptr; double t = now();
      hipError_t e = hipMalloc(&q, 10 * G);
      double dt = now() - t;
      printf("round %d: hipMalloc 10 GB #%d: %.1f ms%s\n", round, i, dt, e == hipSuccess ? "" : " FAILED");
      if (e == hipSuccess) p.push_back(q);
    }
    printf("round %d: 120 GB in %.1f ms\n", round, now() - t_all);
    double t = now();
    for (void *q : p) hipFree(q);
    printf("round %d: frees %.1f ms\n", round, now() - t);
  }
  {   // one piece
    void *q = nullptr; double t = now(); hipMalloc(&q, 64 * G); printf("one 64 GB piece: %.1f ms\n", now() - t);
    t = now(); hipMemset(q, 0, 64 * G); hipDeviceSynchronize(); printf("memset of it: %.1f ms\n", now() - t);
    hipFree(q);
  }
  {   // stream-ordered pool with a high release threshold
    hipStream_t st; hipStreamCreate(&st);
    hipMemPool_t pool; hipDeviceGetDefaultMemPool(&pool, 0);
    uint64_t thr = ~0ull; hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr);
    for (int round = 0; round < 2; round++) {
      std::vector<void *> p; double t = now();
      for (int i = 0; i < 6; i++) { void *q = nullptr; hipMallocAsync(&q, 10 * G, st); p.push_back(q); }
      hipStreamSynchronize(st);
      printf("pool round %d: 60 GB in %.1f ms\n", round, now() - t);
      for (void *q : p) hipFreeAsync(q, st);
      hipStreamSynchronize(st);
    }
  }
  {   // two threads would tell whether allocations overlap; here: small pieces
    double t = now(); std::vector<void *> p;
    for (int i = 0; i < 64; i++) { void *q = nullptr; hipMalloc(&q, G / 4); p.push_back(q); }
    printf("64 x 0.25 GB: %.1f ms\n", now() - t);
    for (void *q : p) hipFree(q);
  }
  return 0;
}
