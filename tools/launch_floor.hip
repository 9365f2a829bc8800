// Micro-benchmark: per-kernel floor of dependent launches on one stream, eager vs hipGraph (gfx950).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_empty() {}
__global__ void k_one(unsigned long* c) { *c += 1; }
__global__ void k_wide(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  unsigned long* c; CK(hipMalloc(&c, 8)); CK(hipMemset(c, 0, 8));
  float* p; int n = 1 << 20; CK(hipMalloc(&p, n * 4)); CK(hipMemset(p, 0, n * 4));
  const int CH = 64, REP = 200;
  for (int variant = 0; variant < 3; ++variant) {
    auto launch = [&](hipStream_t st) {
      if (variant == 0) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st);
      else if (variant == 1) hipLaunchKernelGGL(k_one, dim3(1), dim3(1), 0, st, c);
      else hipLaunchKernelGGL(k_wide, dim3(n / 256), dim3(256), 0, st, p, n);
    };
    // eager
    for (int i = 0; i < 200; ++i) launch(s);
    CK(hipStreamSynchronize(s));
    double t0 = now();
    for (int r = 0; r < REP; ++r) for (int i = 0; i < CH; ++i) launch(s);
    CK(hipStreamSynchronize(s));
    double eager = (now() - t0) / (REP * CH);
    // graph
    hipGraph_t g; hipGraphExec_t ex;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < CH; ++i) launch(s);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    for (int r = 0; r < 20; ++r) CK(hipGraphLaunch(ex, s));
    CK(hipStreamSynchronize(s));
    t0 = now();
    for (int r = 0; r < REP; ++r) CK(hipGraphLaunch(ex, s));
    CK(hipStreamSynchronize(s));
    double graph = (now() - t0) / (REP * CH);
    printf("variant %d (%s): eager %.2f us/kernel, graph %.2f us/kernel\n", variant,
           variant == 0 ? "empty" : variant == 1 ? "1-thread RMW" : "4MB elementwise", eager, graph);
  }
  return 0;
}
