// Microbenchmark: cost of an in-kernel exchange between two workgroups on different CUs of the same XCD
// (blockIdx b and b ^ 8: workgroups are dealt round-robin over the 8 XCDs), as the paired whole-block kernel needs it.
//   mode 0: formal agent-scope release / acquire fences (buffer_wbl2 sc1 / buffer_inv sc1)
//   mode 3: as 1, but no acquire fence: partner data read with relaxed agent-scope atomic loads (sc1)
//   mode 4: as 1, but no acquire fence: plain loads (the lines were never in this CU's L1 during this launch)
//   mode 2: no exchange at all (launch + filler floor)
//   mode 1: s_waitcnt vmcnt(0) + barrier + relaxed agent-scope flag; consumer invalidates its L1 only (buffer_inv sc1)
// Prints us per launch for `rounds` exchanges of `kb` KB per workgroup, the number of pairs whose halves sat on different
// XCDs, and the number of timed-out waits.   hipcc --offload-arch=gfx950 -O3 -o /tmp/pair_sync pair_sync.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(512) void pair_kernel(float* xbuf, int* flags, int* stats, float* sink, int kb, int rounds, int filler, unsigned long long* ts) {
  const int b = blockIdx.x, h = (b >> 3) & 1, pair = (b >> 4) * 8 + (b & 7);
  const int tid = threadIdx.x;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 0xf;
  if (tid == 0) stats[4 + b] = (int)xcc;
  const size_t per = (size_t)kb * 256;                                  // floats per half per round
  float acc = 0.f;
  for (int r = 0; r < rounds; ++r) {
    // some ALU filler between exchanges (the phases of the real kernel)
    float t = (float)tid;
    for (int i = 0; i < filler; ++i) t = t * 1.0001f + 0.5f;
    if (MODE == 2) { acc += t; continue; }
    unsigned long long T0 = wall_clock64();
    float* mine = xbuf + ((size_t)(pair * 2 + h) * rounds + r) * per;
    const float* theirs = xbuf + ((size_t)(pair * 2 + (h ^ 1)) * rounds + r) * per;
    for (size_t i = (size_t)tid * 4; i < per; i += 512 * 4) *reinterpret_cast<f32x4*>(mine + i) = (f32x4){t, (float)b, (float)r, (float)i};
    if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // modes 1, 3, 4
    unsigned long long T1 = wall_clock64();
    __syncthreads();
    unsigned long long T2 = wall_clock64(), T3 = 0;
    int* myflag = flags + (pair * 2 + h) * rounds + r;
    int* theirflag = flags + (pair * 2 + (h ^ 1)) * rounds + r;
    if (tid == 0) {
      __hip_atomic_store(myflag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int spins = 0;
      while (__hip_atomic_load(theirflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 20)) { atomicAdd(stats + 0, 1); break; }
      }
      T3 = wall_clock64();
      __hip_atomic_store(theirflag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // consumer resets: no epoch needed
    }
    __syncthreads();
    if (MODE <= 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    else asm volatile("" ::: "memory");
    for (size_t i = (size_t)tid * 4; i < per; i += 512 * 4) {
      f32x4 v;
      if (MODE == 3) {                      // every dword as a relaxed agent-scope atomic load (sc1: misses the L1)
        v.x = __hip_atomic_load(theirs + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); v.y = __hip_atomic_load(theirs + i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v.z = __hip_atomic_load(theirs + i + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); v.w = __hip_atomic_load(theirs + i + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else v = *reinterpret_cast<const f32x4*>(theirs + i);   // MODE 4: plain loads (these lines were never in this CU's L1 during this launch)
      if (v.y != (float)(b ^ 8) || v.z != (float)r || v.w != (float)i) atomicAdd(stats + 1, 1);
      acc += v.x;
    }
    if (tid == 0 && r == rounds - 1) { unsigned long long T4 = wall_clock64(); ts[b * 4 + 0] = T1 - T0; ts[b * 4 + 1] = T2 - T1; ts[b * 4 + 2] = T3 - T2; ts[b * 4 + 3] = T4 - T3; }
  }
  if (acc == 12345.678f) sink[0] = acc;
}

int main(int argc, char** argv) {
  const int kb = argc > 1 ? atoi(argv[1]) : 20, rounds = argc > 2 ? atoi(argv[2]) : 2, filler = argc > 3 ? atoi(argv[3]) : 2000;
  const int nblk = 256;
  float* xbuf; int* flags; int* stats; float* sink; unsigned long long* ts;
  CK(hipMalloc(&ts, nblk * 4 * sizeof(unsigned long long)));
  CK(hipMalloc(&xbuf, (size_t)nblk * rounds * kb * 1024)); CK(hipMalloc(&flags, nblk * rounds * sizeof(int)));
  CK(hipMalloc(&stats, (4 + nblk) * sizeof(int))); CK(hipMalloc(&sink, 4));
  CK(hipMemset(flags, 0, nblk * rounds * sizeof(int))); CK(hipMemset(stats, 0, (4 + nblk) * sizeof(int)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 5; ++mode) {
    float best = 1e9f, sum = 0.f; const int reps = 30;
    for (int it = 0; it < reps + 5; ++it) {
      CK(hipEventRecord(e0));
      if (mode == 0) hipLaunchKernelGGL(pair_kernel<0>, dim3(nblk), dim3(512), 0, 0, xbuf, flags, stats, sink, kb, rounds, filler, ts);
      else if (mode == 1) hipLaunchKernelGGL(pair_kernel<1>, dim3(nblk), dim3(512), 0, 0, xbuf, flags, stats, sink, kb, rounds, filler, ts);
      else if (mode == 3) hipLaunchKernelGGL(pair_kernel<3>, dim3(nblk), dim3(512), 0, 0, xbuf, flags, stats, sink, kb, rounds, filler, ts);
      else if (mode == 4) hipLaunchKernelGGL(pair_kernel<4>, dim3(nblk), dim3(512), 0, 0, xbuf, flags, stats, sink, kb, rounds, filler, ts);
      else hipLaunchKernelGGL(pair_kernel<2>, dim3(nblk), dim3(512), 0, 0, xbuf, flags, stats, sink, kb, rounds, filler, ts);   // no exchange: floor
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (it >= 5) { sum += ms; if (ms < best) best = ms; }
    }
    std::vector<int> st(4 + nblk);
    CK(hipMemcpy(st.data(), stats, st.size() * sizeof(int), hipMemcpyDeviceToHost));
    int cross = 0;
    for (int b = 0; b < nblk; ++b) if (st[4 + b] != st[4 + (b ^ 8)]) ++cross;
    printf("mode %d: %d KB x %d rounds: avg %.1f us, best %.1f us; timeouts %d, bad words %d, halves on different XCDs %d / %d; xcc of blocks 0..15:", mode, kb, rounds,
           sum / reps * 1000.f, best * 1000.f, st[0], st[1], cross, nblk);
    for (int b = 0; b < 16; ++b) printf(" %d", st[4 + b]);
    printf("\n");
    { std::vector<unsigned long long> t(nblk * 4); CK(hipMemcpy(t.data(), ts, t.size() * 8, hipMemcpyDeviceToHost)); double a[4] = {0, 0, 0, 0}; for (int b = 0; b < nblk; ++b) for (int i = 0; i < 4; ++i) a[i] += (double)t[b * 4 + i] / nblk;
      printf("   last round, thread 0, x10 ns: stores+fence %.0f  barrier %.0f  flag store -> partner flag seen %.0f  barrier+inv+read %.0f\n", a[0], a[1], a[2], a[3]); }
    CK(hipMemset(stats, 0, 4 * sizeof(int)));
  }
  return 0;
}
