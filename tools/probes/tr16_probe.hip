// Probe of ds_read_b64_tr_b16 (gfx950 LDS transpose read): every lane reads 8 bytes at its own address;
// prints which LDS elements each lane receives.  Build: hipcc --offload-arch=gfx950 -O2 tr16_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned short* in, unsigned short* out, const int* addr) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = in[i];
  __syncthreads();
  v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(lds + addr[threadIdx.x]));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
  std::vector<unsigned short> in(8192);
  for (int i = 0; i < 8192; ++i) in[i] = (unsigned short)i;
  unsigned short *din, *dout; int* daddr;
  hipMalloc(&din, 8192 * 2); hipMalloc(&dout, 64 * 4 * 2); hipMalloc(&daddr, 64 * 4);
  hipMemcpy(din, in.data(), 8192 * 2, hipMemcpyHostToDevice);
  for (int pat = 0; pat < 3; ++pat) {
    std::vector<int> addr(64);
    for (int l = 0; l < 64; ++l) {
      if (pat == 0) addr[l] = l * 4;                       // lane l -> elements [4l, 4l+4)
      if (pat == 1) addr[l] = (l % 16) * 64 + (l / 16) * 4; // row = l%16 (pitch 64), col block = l/16
      if (pat == 2) addr[l] = (l / 16) * 256 + (l % 16) * 4; 
    }
    hipMemcpy(daddr, addr.data(), 64 * 4, hipMemcpyHostToDevice);
    k<<<1, 64>>>(din, dout, daddr);
    std::vector<unsigned short> out(256);
    hipMemcpy(out.data(), dout, 512, hipMemcpyDeviceToHost);
    printf("pattern %d\n", pat);
    for (int l = 0; l < 64; ++l)
      printf("lane %2d addr %4d -> %4d %4d %4d %4d\n", l, addr[l], out[l * 4], out[l * 4 + 1], out[l * 4 + 2], out[l * 4 + 3]);
  }
  return 0;
}
