// r05a_mfma_k16_probe.hip -- does accumulating a v_mfma_f32_16x16x16_f16 onto the result of a v_mfma_f32_16x16x32_f16 give
// run-to-run identical results on gfx950 / hipcc 7.2?  (Round 4's af_xattn320t_kernel saw single cells change between runs with that
// pair and switched to zero-padded 32-deep MFMAs: csrc/af_xattn_fused.hip, mfma_k16.)  No LDS, no barriers: operands come straight from
// global memory, so anything that changes between launches is the MFMA pair itself (a missing hazard wait), not a race.
//
//   hipcc --offload-arch=gfx950 -O3 r05a_mfma_k16_probe.hip -o r05a_mfma_k16_probe && ./r05a_mfma_k16_probe
//
// Variants (each launched NREP times on the same inputs, every launch compared bit for bit with the first and with a host fp32 sum):
//   0  acc = mfma32(a, b, 0);  acc = mfma16(a', b', acc)                      the plain dependent pair
//   1  the same with the pair repeated 8 times in a chain (acc feeds acc)
//   2  as the kernel does: S = mfma32(k, q, 0); S = mfma16(k', q', S); P = half(S); O = mfma32(v, P.., O); O = mfma16(v', P', O)
//      (the accumulator of one product converted and used as the B operand of the next)
//   3  variant 0 with the 16-deep step written as a zero-padded 32-deep MFMA (what the kernel ships): the control
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef _Float16 half_t;
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(2);                                                                  \
    }                                                                           \
  } while (0)

__device__ __forceinline__ floatx4 mfma16_padded(const half4_t& a, const half4_t& b, const floatx4& c) {
  const half8_t a8 = {a[0], a[1], a[2], a[3], (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
  const half8_t b8 = {b[0], b[1], b[2], b[3], (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c, 0, 0, 0);
}

// per wave: A8 [64 lanes][8], B8 [64][8], A4 [64][4], B4 [64][4] -> out [64][4]
template <int VARIANT>
__global__ __launch_bounds__(256) void probe_kernel(const half_t* __restrict__ a8g, const half_t* __restrict__ b8g, const half_t* __restrict__ a4g,
                                                    const half_t* __restrict__ b4g, float* __restrict__ out) {
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  const size_t o8 = ((size_t)wave * 64 + lane) * 8, o4 = ((size_t)wave * 64 + lane) * 4;
  const half8_t a8 = *reinterpret_cast<const half8_t*>(a8g + o8), b8 = *reinterpret_cast<const half8_t*>(b8g + o8);
  const half4_t a4 = *reinterpret_cast<const half4_t*>(a4g + o4), b4 = *reinterpret_cast<const half4_t*>(b4g + o4);
  const floatx4 z = {0.f, 0.f, 0.f, 0.f};
  floatx4 acc = z;
  if constexpr (VARIANT == 0) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, z, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc, 0, 0, 0);
  } else if constexpr (VARIANT == 1) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc, 0, 0, 0);
    }
  } else if constexpr (VARIANT == 2) {
    floatx4 s = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, z, 0, 0, 0);
    s = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, s, 0, 0, 0);
    const half4_t p = {(half_t)(s[0] * 0.01f), (half_t)(s[1] * 0.01f), (half_t)(s[2] * 0.01f), (half_t)(s[3] * 0.01f)};
    // P as the B operand of a 16-deep product onto a 32-deep result (the O^T = V^T P^T step with its 16-key tail)
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(b8, a8, z, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, p, acc, 0, 0, 0);
  } else {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, z, 0, 0, 0);
    acc = mfma16_padded(a4, b4, acc);
  }
  *reinterpret_cast<floatx4*>(out + o4) = acc;
}

static void host_ref(int variant, const half_t* a8, const half_t* b8, const half_t* a4, const half_t* b4, float* out) {
  // one wave: lane l holds A[row l&15][k = 8 (l>>4) + j], B[k][col l&15]; D: col = l & 15, row = 4 (l >> 4) + r
  auto A8 = [&](int row, int k) { return (float)a8[((k >> 3) * 16 + row) * 8 + (k & 7)]; };
  auto B8 = [&](int k, int col) { return (float)b8[((k >> 3) * 16 + col) * 8 + (k & 7)]; };
  auto A4 = [&](int row, int k) { return (float)a4[((k >> 2) * 16 + row) * 4 + (k & 3)]; };
  auto B4 = [&](int k, int col) { return (float)b4[((k >> 2) * 16 + col) * 4 + (k & 3)]; };
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const int col = l & 15, row = 4 * (l >> 4) + r;
      double s = 0.0;
      for (int k = 0; k < 32; ++k) s += (double)A8(row, k) * B8(k, col);
      for (int k = 0; k < 16; ++k) s += (double)A4(row, k) * B4(k, col);
      out[l * 4 + r] = (float)(variant == 1 ? 8.0 * s : s);
    }
}

template <int V>
static int run_variant(int nwaves, int nrep, const half_t* a8, const half_t* b8, const half_t* a4, const half_t* b4, float* out,
                       const std::vector<half_t>& ha8, const std::vector<half_t>& hb8, const std::vector<half_t>& ha4, const std::vector<half_t>& hb4) {
  const size_t n = (size_t)nwaves * 64 * 4;
  std::vector<float> first(n), cur(n);
  long diff_launches = 0, diff_cells = 0;
  for (int rep = 0; rep < nrep; ++rep) {
    CK(hipMemset(out, 0xFF, n * 4));
    hipLaunchKernelGGL(probe_kernel<V>, dim3(nwaves / 4), dim3(256), 0, 0, a8, b8, a4, b4, out);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(rep == 0 ? first.data() : cur.data(), out, n * 4, hipMemcpyDeviceToHost));
    if (rep > 0) {
      long d = 0;
      for (size_t i = 0; i < n; ++i) d += memcmp(&first[i], &cur[i], 4) != 0;
      diff_launches += d > 0;
      diff_cells += d;
    }
  }
  double maxerr = 0.0;
  if (V != 2) {
    std::vector<float> ref(256);
    for (int w = 0; w < nwaves; w += nwaves / 16) {
      host_ref(V, &ha8[(size_t)w * 512], &hb8[(size_t)w * 512], &ha4[(size_t)w * 256], &hb4[(size_t)w * 256], ref.data());
      for (int i = 0; i < 256; ++i) {
        const double e = fabs((double)first[(size_t)w * 256 + i] - ref[i]) / (1.0 + fabs((double)ref[i]));
        if (e > maxerr) maxerr = e;
      }
    }
  }
  printf("variant %d: %d launches x %d waves: %ld launches differ from the first (%ld cells); max rel err vs host fp64 sum %.3e\n", V, nrep, nwaves,
         diff_launches, diff_cells, maxerr);
  return diff_launches > 0;
}

int main() {
  const int nwaves = 256 * 16 * 4, nrep = 200;      // 16 workgroups per CU: the chip is full and waves share SIMDs
  const size_t n8 = (size_t)nwaves * 64 * 8, n4 = (size_t)nwaves * 64 * 4;
  std::vector<half_t> ha8(n8), hb8(n8), ha4(n4), hb4(n4);
  srand(5);
  auto rnd = [] { return (half_t)((rand() % 2001 - 1000) / 500.0f); };
  for (auto& v : ha8) v = rnd();
  for (auto& v : hb8) v = rnd();
  for (auto& v : ha4) v = rnd();
  for (auto& v : hb4) v = rnd();
  half_t *a8, *b8, *a4, *b4;
  float* out;
  CK(hipMalloc(&a8, n8 * 2));
  CK(hipMalloc(&b8, n8 * 2));
  CK(hipMalloc(&a4, n4 * 2));
  CK(hipMalloc(&b4, n4 * 2));
  CK(hipMalloc(&out, n4 * 4));
  CK(hipMemcpy(a8, ha8.data(), n8 * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(b8, hb8.data(), n8 * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(a4, ha4.data(), n4 * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(b4, hb4.data(), n4 * 2, hipMemcpyHostToDevice));
  int bad = 0;
  bad += run_variant<0>(nwaves, nrep, a8, b8, a4, b4, out, ha8, hb8, ha4, hb4);
  bad += run_variant<1>(nwaves, nrep, a8, b8, a4, b4, out, ha8, hb8, ha4, hb4);
  bad += run_variant<2>(nwaves, nrep, a8, b8, a4, b4, out, ha8, hb8, ha4, hb4);
  bad += run_variant<3>(nwaves, nrep, a8, b8, a4, b4, out, ha8, hb8, ha4, hb4);
  printf(bad ? "RESULT: the 16x16x16-after-16x16x32 pair is NOT run-to-run stable in isolation (compiler / hardware hazard)\n"
             : "RESULT: every variant is bit-identical across launches: the pair is stable in isolation\n");
  return 0;
}
