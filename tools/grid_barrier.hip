// Micro-benchmark (gfx950): cost of one dependent phase as (a) a kernel boundary inside a hipGraph and (b) a
// device-wide barrier inside one persistent kernel.  Each phase, every workgroup reads the 4 KB slice another
// workgroup (on another XCD) wrote in the previous phase, adds 1 and writes its own slice.
// The spin is bounded: a barrier that does not complete within ~2^20 polls sets an error flag and every
// workgroup leaves the kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

constexpr int SLICE = 1024;   // floats per workgroup

__device__ __forceinline__ void phase_body(const float* __restrict__ src, float* __restrict__ dst, int nb) {
  const int peer = (blockIdx.x * 37 + 11) % nb;
  for (int i = threadIdx.x; i < SLICE; i += blockDim.x) dst[blockIdx.x * SLICE + i] = src[peer * SLICE + i] + 1.f;
}

__global__ void k_phase(const float* src, float* dst, int nb) { phase_body(src, dst, nb); }

__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned target, unsigned* err) {
  __threadfence();   // agent-scope release of this wave's stores (L2 write-back: the XCD L2s are not coherent)
  __syncthreads();
  __shared__ int ok;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int good = 0;
    for (int spin = 0; spin < (1 << 20); ++spin) {
      if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) { good = 1; break; }
      if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
      __builtin_amdgcn_s_sleep(1);
    }
    if (!good) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ok = good;
  }
  __syncthreads();
  __threadfence();   // acquire: drop stale lines
  return ok != 0;
}

__global__ void k_persistent(float* a, float* b, int nb, int phases, unsigned* ctr, unsigned* err, unsigned base) {
  for (int ph = 0; ph < phases; ++ph) {
    phase_body((ph & 1) ? b : a, (ph & 1) ? a : b, nb);
    if (!grid_barrier(ctr, base + (unsigned)(ph + 1) * nb, err)) return;
  }
}

// Variant without fences: the phase data bypasses the (non-coherent) XCD L2s -- device-scope relaxed atomic stores
// and loads, one dword each -- so the barrier is only the counter: one atomic add + polling.
__device__ __forceinline__ bool grid_barrier_nofence(unsigned* ctr, unsigned target, unsigned* err) {
  __builtin_amdgcn_s_waitcnt(0);   // this wave's stores have left (vmcnt 0)
  __syncthreads();
  __shared__ int ok2;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int good = 0;
    for (int spin = 0; spin < (1 << 20); ++spin) {
      if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) { good = 1; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    if (!good) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ok2 = good;
  }
  __syncthreads();
  return ok2 != 0;
}

__global__ void k_persistent_nofence(float* a, float* b, int nb, int phases, unsigned* ctr, unsigned* err, unsigned base) {
  for (int ph = 0; ph < phases; ++ph) {
    const float* src = (ph & 1) ? b : a;
    float* dst = (ph & 1) ? a : b;
    const int peer = (blockIdx.x * 37 + 11) % nb;
    for (int i = threadIdx.x; i < SLICE; i += blockDim.x) {
      const float v = __hip_atomic_load(src + peer * SLICE + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(dst + blockIdx.x * SLICE + i, v + 1.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!grid_barrier_nofence(ctr, base + (unsigned)(ph + 1) * nb, err)) return;
  }
}

__global__ void k_barrier_only_nofence(int nb, int phases, unsigned* ctr, unsigned* err, unsigned base) {
  for (int ph = 0; ph < phases; ++ph)
    if (!grid_barrier_nofence(ctr, base + (unsigned)(ph + 1) * nb, err)) return;
}

// barrier only (no data): the pure synchronisation cost
__global__ void k_barrier_only(int nb, int phases, unsigned* ctr, unsigned* err, unsigned base) {
  for (int ph = 0; ph < phases; ++ph)
    if (!grid_barrier(ctr, base + (unsigned)(ph + 1) * nb, err)) return;
}

int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  const int PH = 64, REP = 50;
  for (int nb : {64, 256, 512}) {
    for (int threads : {256, 512}) {
      float *a, *b; unsigned *ctr, *err;
      CK(hipMalloc(&a, nb * SLICE * 4)); CK(hipMalloc(&b, nb * SLICE * 4));
      CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&err, 4));
      CK(hipMemset(a, 0, nb * SLICE * 4)); CK(hipMemset(b, 0, nb * SLICE * 4));
      CK(hipMemset(ctr, 0, 4)); CK(hipMemset(err, 0, 4));
      // (a) graph of PH kernels
      hipGraph_t g; hipGraphExec_t ex;
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      for (int ph = 0; ph < PH; ++ph)
        hipLaunchKernelGGL(k_phase, dim3(nb), dim3(threads), 0, s, (ph & 1) ? b : a, (ph & 1) ? a : b, nb);
      CK(hipStreamEndCapture(s, &g));
      CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
      for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ex, s));
      CK(hipStreamSynchronize(s));
      double t0 = now();
      for (int r = 0; r < REP; ++r) CK(hipGraphLaunch(ex, s));
      CK(hipStreamSynchronize(s));
      const double t_graph = (now() - t0) / (REP * PH);
      // (b) persistent kernel, PH phases per launch
      CK(hipMemset(a, 0, nb * SLICE * 4)); CK(hipMemset(b, 0, nb * SLICE * 4));
      unsigned base = 0;
      auto run = [&](bool data) {
        if (data) hipLaunchKernelGGL(k_persistent, dim3(nb), dim3(threads), 0, s, a, b, nb, PH, ctr, err, base);
        else hipLaunchKernelGGL(k_barrier_only, dim3(nb), dim3(threads), 0, s, nb, PH, ctr, err, base);
        base += (unsigned)PH * nb;
      };
      auto run_nf = [&](bool data) {
        if (data) hipLaunchKernelGGL(k_persistent_nofence, dim3(nb), dim3(threads), 0, s, a, b, nb, PH, ctr, err, base);
        else hipLaunchKernelGGL(k_barrier_only_nofence, dim3(nb), dim3(threads), 0, s, nb, PH, ctr, err, base);
        base += (unsigned)PH * nb;
      };
      run(true);
      CK(hipStreamSynchronize(s));
      std::vector<float> h(nb * SLICE);
      CK(hipMemcpy(h.data(), a, nb * SLICE * 4, hipMemcpyDeviceToHost));   // PH even -> result in a
      int bad = 0;
      for (float v : h) bad += (v != (float)PH);
      unsigned herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
      t0 = now();
      for (int r = 0; r < REP; ++r) run(true);
      CK(hipStreamSynchronize(s));
      const double t_pers = (now() - t0) / (REP * PH);
      t0 = now();
      for (int r = 0; r < REP; ++r) run(false);
      CK(hipStreamSynchronize(s));
      const double t_bar = (now() - t0) / (REP * PH);
      // fence-free variant
      CK(hipMemset(a, 0, nb * SLICE * 4)); CK(hipMemset(b, 0, nb * SLICE * 4));
      run_nf(true);
      CK(hipStreamSynchronize(s));
      CK(hipMemcpy(h.data(), a, nb * SLICE * 4, hipMemcpyDeviceToHost));
      int bad_nf = 0;
      for (float v : h) bad_nf += (v != (float)PH);
      t0 = now();
      for (int r = 0; r < REP; ++r) run_nf(true);
      CK(hipStreamSynchronize(s));
      const double t_pers_nf = (now() - t0) / (REP * PH);
      t0 = now();
      for (int r = 0; r < REP; ++r) run_nf(false);
      CK(hipStreamSynchronize(s));
      const double t_bar_nf = (now() - t0) / (REP * PH);
      printf("   fence-free (device-scope atomic loads/stores for the data): %.2f us/phase (barrier alone %.2f), wrong %d\n",
             t_pers_nf, t_bar_nf, bad_nf);
      CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
      printf("workgroups %4d x %3d threads: graph kernel boundary %.2f us/phase | persistent + grid barrier %.2f us/phase "
             "(barrier alone %.2f) | wrong %d, timeout flag %u\n", nb, threads, t_graph, t_pers, t_bar, bad, herr);
      CK(hipGraphExecDestroy(ex)); CK(hipGraphDestroy(g));
      CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(ctr)); CK(hipFree(err));
    }
  }
  return 0;
}
