// Hardware probe: semantics of ds_read_b64_tr_b16 on gfx950 (used by the wgrad kernel).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const short* in, short* out) {
  __shared__ __attribute__((aligned(16))) short lds[64*64];
  int l = threadIdx.x;
  for (int i = l; i < 64*64; i += 64) lds[i] = in[i];
  __syncthreads();
  int g = l >> 4, i = l & 15, q = i >> 2, pp = i & 3;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + (8*g+q)*64 + 4*pp));
  out[l*4+0]=v[0]; out[l*4+1]=v[1]; out[l*4+2]=v[2]; out[l*4+3]=v[3];
}
int main() {
  short h[64*64], o[256];
  for (int r = 0; r < 64; ++r) for (int c = 0; c < 64; ++c) h[r*64+c] = (short)(r*100 + c);
  short *din, *dout;
  hipMalloc(&din, sizeof(h)); hipMalloc(&dout, sizeof(o));
  hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1,64>>>(din, dout);
  hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, o[l*4], o[l*4+1], o[l*4+2], o[l*4+3]);
  return 0;
}
