// Probe of global_load_lds semantics on gfx950: per-lane global source, lane-linear LDS destination
// (wave-uniform base + lane*16), counted vmcnt + raw barrier.  Prints OK/FAIL.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half_t;
__global__ void k(const half_t* __restrict__ g, const half_t* __restrict__ zero, half_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // each wave copies 2 x 1 KiB: piece j of wave w -> LDS bytes [(w*2+j)*1024, +1024); lane i's source = row (w*2+j)*64+ (63-i) (reversed!)
  for (int j = 0; j < 2; ++j) {
    const int piece = wave * 2 + j;
    const int srcrow = piece * 64 + (63 - lane);
    const half_t* src = (lane % 7 == 3) ? zero : g + (size_t)srcrow * 8;    // some lanes read the zero page
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(smem + piece * 1024), 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int i = tid; i < 4 * 2 * 64 * 8; i += 256) out[i] = ((half_t*)smem)[i];
}
int main() {
  const int n = 4 * 2 * 64 * 8;
  std::vector<half_t> h(n), o(n);
  for (int i = 0; i < n; ++i) h[i] = (half_t)(i % 2048);
  half_t *g, *z, *out;
  hipMalloc(&g, n * 2); hipMalloc(&z, 256); hipMalloc(&out, n * 2);
  hipMemcpy(g, h.data(), n * 2, hipMemcpyHostToDevice); hipMemset(z, 0, 256);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 8192, 0, g, z, out);
  hipMemcpy(o.data(), out, n * 2, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int piece = 0; piece < 8; ++piece) for (int lane = 0; lane < 64; ++lane) for (int e = 0; e < 8; ++e) {
    const int srcrow = piece * 64 + (63 - lane);
    const float want = (lane % 7 == 3) ? 0.f : (float)((srcrow * 8 + e) % 2048);
    const float got = (float)o[(piece * 64 + lane) * 8 + e];
    if (want != got) { if (bad < 5) printf("mismatch piece %d lane %d e %d: got %f want %f\n", piece, lane, e, got, want); ++bad; }
  }
  printf(bad ? "FAIL %d\n" : "OK lane-linear LDS dest, per-lane source, zero page\n", bad);
  return bad != 0;
}
