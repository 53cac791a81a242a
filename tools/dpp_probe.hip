#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* out){
  int x = threadIdx.x;
  int a = __builtin_amdgcn_update_dpp(-1, x, 0x138, 0xf, 0xf, false); // wave_shr:1
  int b = __builtin_amdgcn_update_dpp(-1, x, 0x130, 0xf, 0xf, false); // wave_shl:1
  out[threadIdx.x] = a; out[64+threadIdx.x] = b;
}
int main(){ int* d; hipMalloc(&d, 128*4); k<<<1,64>>>(d); int h[128]; hipMemcpy(h,d,512,hipMemcpyDeviceToHost);
 printf("shr: lane0=%d lane1=%d lane16=%d lane32=%d lane63=%d\n",h[0],h[1],h[16],h[32],h[63]);
 printf("shl: lane0=%d lane1=%d lane15=%d lane31=%d lane62=%d lane63=%d\n",h[64],h[65],h[64+15],h[64+31],h[64+62],h[64+63]); return 0;}
