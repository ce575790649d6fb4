// Probe: do DPP wave_shr:1 / wave_shl:1 behave as lane i <- lane i-1 / i+1 on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const int* in, int* a, int* b, int* c, int* d){
  int t = threadIdx.x;
  int v = in[t];
  a[t] = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, false); // wave_shr:1
  b[t] = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xf, 0xf, false); // wave_shl:1
  c[t] = __shfl_up(v,1); d[t] = __shfl_down(v,1);
}
int main(){
  int h[64], *in,*a,*b,*c,*d; int ra[64],rb[64],rc[64],rd[64];
  for(int i=0;i<64;i++) h[i]=100+i;
  hipMalloc(&in,256);hipMalloc(&a,256);hipMalloc(&b,256);hipMalloc(&c,256);hipMalloc(&d,256);
  hipMemcpy(in,h,256,hipMemcpyHostToDevice);
  k<<<1,64>>>(in,a,b,c,d);
  hipMemcpy(ra,a,256,hipMemcpyDeviceToHost);hipMemcpy(rb,b,256,hipMemcpyDeviceToHost);
  hipMemcpy(rc,c,256,hipMemcpyDeviceToHost);hipMemcpy(rd,d,256,hipMemcpyDeviceToHost);
  int ok_shr=1, ok_shl=1;
  for(int i=1;i<64;i++) if(ra[i]!=h[i-1]) ok_shr=0;
  for(int i=0;i<63;i++) if(rb[i]!=h[i+1]) ok_shl=0;
  printf("wave_shr ok=%d lane0=%d | wave_shl ok=%d lane63=%d\n", ok_shr, ra[0], ok_shl, rb[63]);
  printf("shr: %d %d %d %d ... %d %d | 15,16,17: %d %d %d 31,32,33: %d %d %d\n", ra[0],ra[1],ra[2],ra[3],ra[62],ra[63],ra[15],ra[16],ra[17],ra[31],ra[32],ra[33]);
  printf("shl: %d %d %d %d ... %d %d | 15,16,17: %d %d %d 31,32,33: %d %d %d\n", rb[0],rb[1],rb[2],rb[3],rb[62],rb[63],rb[15],rb[16],rb[17],rb[31],rb[32],rb[33]);
  printf("shfl_up lane0=%d lane1=%d; shfl_down lane63=%d lane62=%d\n", rc[0], rc[1], rd[63], rd[62]);
  hipDeviceProp_t p; hipGetDeviceProperties(&p,0);
  printf("dev=%s arch=%s CUs=%d clock=%d memclk=%d bus=%d L2=%d smem/blk=%zu\n", p.name,p.gcnArchName,p.multiProcessorCount,p.clockRate,p.memoryClockRate,p.memoryBusWidth,p.l2CacheSize,p.sharedMemPerBlock);
  return 0;
}
