// Probe: lane/element mapping of ds_read_b64_tr_b16 on gfx950.  LDS holds u16 value == element index;
// lane l supplies byte address addr[l]; prints the 4 values each lane receives.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const int* addr, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)((char*)lds + addr[threadIdx.x]));
  unsigned short* pv = (unsigned short*)&v;
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = pv[j];
}
int main() {
  int h_addr[64]; unsigned short h_out[256];
  int *d_addr; unsigned short* d_out;
  hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
  for (int mode = 0; mode < 2; ++mode) {
    for (int l = 0; l < 64; ++l) {
      int i = l & 15, g = l >> 4;
      // mode 0: natural (lane*8 bytes).  mode 1: row (8g + (i>>2)) of a 288-byte-stride tile, col chunk (i&3)*8 bytes
      h_addr[l] = mode == 0 ? l * 8 : (8 * g + (i >> 2)) * 288 + (i & 3) * 8;
    }
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d_addr, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) printf("lane %2d addr %5d : %5d %5d %5d %5d\n", l, h_addr[l], h_out[4*l], h_out[4*l+1], h_out[4*l+2], h_out[4*l+3]);
  }
  return 0;
}
