// pk_check.hip -- the packed-fp32 helpers with op_sel / neg modifiers on known values
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_sum_diff(f32x2 pq) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(r) : "v"(pq));
  return r;
}
__device__ __forceinline__ f32x2 pk_sum_diff2(f32x2 pq) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(r) : "v"(pq), "v"(pq));
  return r;
}
__global__ void k(float* o, const float* in) {
  f32x2 a = {in[0], in[1]};
  f32x2 r = pk_sum_diff(a), r2 = pk_sum_diff2(a);
  o[0] = r[0]; o[1] = r[1]; o[2] = r2[0]; o[3] = r2[1];
}
int main() {
  float *o, *in, h[4], hi[2] = {3.f, 5.f};
  hipMalloc(&o, 16); hipMalloc(&in, 8);
  hipMemcpy(in, hi, 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, in);
  hipMemcpy(h, o, 16, hipMemcpyDeviceToHost);
  printf("pk_sum_diff(3,5) = (%g, %g) [want 8, -2]; two-operand form (%g, %g)\n", h[0], h[1], h[2], h[3]);
  return 0;
}
