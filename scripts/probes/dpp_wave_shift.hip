#include <hip/hip_runtime.h>
__global__ void k(unsigned *p){ unsigned a=p[threadIdx.x];
 unsigned b = __builtin_amdgcn_update_dpp(0u, a, 0x130, 0xf, 0xf, true);
 unsigned c = __builtin_amdgcn_update_dpp(0u, a, 0x138, 0xf, 0xf, true);
 p[threadIdx.x+64]=b; p[threadIdx.x+128]=c; }
int main(){ unsigned *d; hipMalloc(&d, 192*4); unsigned h[192]; for(int i=0;i<64;i++)h[i]=100+i; hipMemcpy(d,h,256,hipMemcpyHostToDevice);
 hipLaunchKernelGGL(k,1,64,0,0,d); hipMemcpy(h,d,768,hipMemcpyDeviceToHost);
 printf("shl(0x130): lane0 %u lane1 %u lane15 %u lane16 %u lane31 %u lane32 %u lane62 %u lane63 %u\n",h[64],h[65],h[79],h[80],h[95],h[96],h[126],h[127]);
 printf("shr(0x138): lane0 %u lane1 %u lane15 %u lane16 %u lane31 %u lane32 %u lane62 %u lane63 %u\n",h[128],h[129],h[143],h[144],h[159],h[160],h[190],h[191]); }
