// Probe (not product): (1) how v_smfmac_f32_32x32x32_bf16 (gfx950 2:4 structured-sparse MFMA) maps a lane's eight
// compressed A values + index bits onto the 32 dense k of its row — learnt by multiplying with an identity B; (2) what it
// buys under the chip's power budget: a loop of sparse MFMAs against the dense v_mfma_f32_32x32x16_bf16 loop of the same
// dense-equivalent work, random operands.   hipcc -O2 --offload-arch=gfx950 tools/probes/smfmac_probe.hip -o /tmp/smfmac_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x16 __attribute__((ext_vector_type(16)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void layout_kernel(float* out, int idx) {
  const int lane = threadIdx.x;
  bf16x8 a;
  for (int j = 0; j < 8; ++j) a[j] = (__bf16)(float)(j + 1);          // compressed value j of this lane = j + 1
  bf16x16 b;
  for (int e = 0; e < 16; ++e) b[e] = (__bf16)((16 * (lane >> 5) + e) == (lane & 31) ? 1.f : 0.f);   // B[k][n] = (k == n), k = 16 (lane / 32) + e
  f32x16 c = {};
  c = __builtin_amdgcn_smfmac_f32_32x32x32_bf16(a, b, c, idx, 0, 0);
  for (int i = 0; i < 16; ++i) out[lane * 16 + i] = c[i];
}

template <bool SPARSE>
__global__ __launch_bounds__(256) void loop_kernel(const unsigned* seed, float* sink, int iters) {
  const int lane = threadIdx.x & 63;
  unsigned s = seed[(blockIdx.x * 256 + threadIdx.x) & 4095] | 1u;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return (float)(int)(s & 0xffff) / 65536.f - 0.5f; };
  bf16x8 a8[4];
  bf16x16 b16[2];
  bf16x8 b8[4];
  for (int q = 0; q < 4; ++q)
    for (int j = 0; j < 8; ++j) { a8[q][j] = (__bf16)rnd(); b8[q][j] = (__bf16)rnd(); }
  for (int q = 0; q < 2; ++q)
    for (int j = 0; j < 16; ++j) b16[q][j] = (__bf16)rnd();
  f32x16 acc[4] = {};
  const int idx = 0x4e4e4e4e ^ (lane * 0x01010101 & 0x44444444);
  for (int it = 0; it < iters; ++it) {
    if (SPARSE) {   // one sparse MFMA = 32 x 32 x 32 dense-equivalent
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_smfmac_f32_32x32x32_bf16(a8[q], b16[q & 1], acc[q], idx, 0, 0);
    } else {        // the same dense work: two 32 x 32 x 16 MFMAs
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[q], b8[q], acc[q], 0, 0, 0);
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[(q + 1) & 3], b8[(q + 2) & 3], acc[q], 0, 0, 0);
      }
    }
  }
  float t = 0.f;
  for (int q = 0; q < 4; ++q)
    for (int i = 0; i < 16; ++i) t += acc[q][i];
  if (t == 123.456f) sink[0] = t;
}

int main() {
  float* out;
  CK(hipMalloc(&out, 64 * 16 * 4));
  std::vector<float> h(64 * 16);
  for (int idx : {0x44444444, (int)0xeeeeeeee, (int)0x8888cccc, 0x4e4e0000, 0x00004e4e}) {
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, out, idx);
    CK(hipMemcpy(h.data(), out, 64 * 16 * 4, hipMemcpyDeviceToHost));
    // D[row][col]: acc i of lane l -> row = (i / 4) * 8 + (l / 32) * 4 + i % 4, col = l % 32.  With B = identity, row r of D is
    // the dense A row r: print rows 0, 1, 5 (the dense k positions 0 .. 31 and which compressed value landed there)
    printf("idx %08x:\n", (unsigned)idx);
    for (int row : {0, 1, 5}) {
      printf("  dense A row %d:", row);
      for (int col = 0; col < 32; ++col) {
        const int i = (row / 8) * 4 + row % 4, l = col + 32 * ((row / 4) % 2);
        printf(" %g", h[l * 16 + i]);
      }
      printf("\n");
    }
  }
  unsigned* seed;
  float* sink;
  CK(hipMalloc(&seed, 4096 * 4));
  CK(hipMalloc(&sink, 4));
  std::vector<unsigned> hs(4096);
  for (auto& v : hs) v = (unsigned)rand() * 2654435761u;
  CK(hipMemcpy(seed, hs.data(), 4096 * 4, hipMemcpyHostToDevice));
  const int iters = 20000;
  for (int rep = 0; rep < 2; ++rep)
    for (int sparse = 0; sparse < 2; ++sparse) {
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0));
      if (sparse) hipLaunchKernelGGL(loop_kernel<true>, dim3(256), dim3(256), 0, 0, seed, sink, iters);
      else hipLaunchKernelGGL(loop_kernel<false>, dim3(256), dim3(256), 0, 0, seed, sink, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double flops = 2.0 * 32 * 32 * 32 * 4.0 * iters * 4 * 256;     // dense-equivalent
      printf("%s: %.2f ms, %.0f dense-equivalent TFLOP/s (4 waves x 256 CUs, %d x 4 per wave)\n", sparse ? "sparse 32x32x32" : "dense 2 x 32x32x16", ms, flops / ms / 1e9, iters);
    }
  return 0;
}
