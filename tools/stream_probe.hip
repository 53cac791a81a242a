// Bandwidth ceiling probe for the access patterns of the multigrid kernels on one MI355X.
// Arrays are laid out like a 513^3 level: idx = plane*i + pitch*j + k, pitch = 528, plane = 528*513.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

__global__ void __launch_bounds__(256) lin_copy(const double2* __restrict__ a, double2* __restrict__ o, size_t n2){
  for(size_t i = blockIdx.x*(size_t)256+threadIdx.x; i<n2; i += (size_t)gridDim.x*256) o[i]=a[i];
}
__global__ void __launch_bounds__(256) lin_2r1w(const double2* __restrict__ a, const double2* __restrict__ b, double2* __restrict__ o, size_t n2){
  for(size_t i = blockIdx.x*(size_t)256+threadIdx.x; i<n2; i += (size_t)gridDim.x*256){ double2 x=a[i], y=b[i]; o[i]=make_double2(x.x+y.x,x.y+y.y);} 
}
// march along i: block = 4 waves; wave w owns RJ rows j0+w*RJ.., 128 k per wave-row (double2 per lane); tile in k by blockIdx.x
template<int RJ> __global__ void __launch_bounds__(256) march_i(const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ o,
   int N, int pitch, long long plane, int CI){
  const int lane=threadIdx.x&63, w=threadIdx.x>>6;
  const int ntk = (N+127)/128; int bid=blockIdx.x; const int tk=bid%ntk; bid/=ntk; const int ntj=(N+4*RJ-1)/(4*RJ); const int tj=bid%ntj; const int ci=bid/ntj;
  const int k = tk*128+2*lane; const int j0=tj*4*RJ+w*RJ;
  const int i0=ci*CI, i1=min(N,i0+CI);
  if(k>=N) return;
  for(int i=i0;i<i1;i++){
    #pragma unroll
    for(int rr=0;rr<RJ;rr++){ int j=j0+rr; if(j<N){ long long p=plane*i+(long long)pitch*j+k; double2 x=*(const double2*)(a+p), y=*(const double2*)(b+p); *(double2*)(o+p)=make_double2(x.x+y.x,x.y+y.y);} }
  }
}
// march along j: wave owns RJ planes i0+w*RJ.., marches j
template<int RJ> __global__ void __launch_bounds__(256) march_j(const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ o,
   int N, int pitch, long long plane, int CJ){
  const int lane=threadIdx.x&63, w=threadIdx.x>>6;
  const int ntk = (N+127)/128; int bid=blockIdx.x; const int tk=bid%ntk; bid/=ntk; const int nti=(N+4*RJ-1)/(4*RJ); const int ti=bid%nti; const int cj=bid/nti;
  const int k = tk*128+2*lane; const int i0=ti*4*RJ+w*RJ;
  const int j0=cj*CJ, j1=min(N,j0+CJ);
  if(k>=N) return;
  for(int j=j0;j<j1;j++){
    #pragma unroll
    for(int rr=0;rr<RJ;rr++){ int i=i0+rr; if(i<N){ long long p=plane*i+(long long)pitch*j+k; double2 x=*(const double2*)(a+p), y=*(const double2*)(b+p); *(double2*)(o+p)=make_double2(x.x+y.x,x.y+y.y);} }
  }
}
int main(){
  const int N=513, pitch=528; const long long plane=(long long)pitch*N; const size_t n=(size_t)plane*N;
  double *a,*b,*o; CK(hipMalloc(&a,n*8)); CK(hipMalloc(&b,n*8)); CK(hipMalloc(&o,n*8));
  CK(hipMemset(a,0,n*8)); CK(hipMemset(b,0,n*8)); CK(hipMemset(o,0,n*8));
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run=[&](const char* name, double bytes, auto f){ f(); CK(hipDeviceSynchronize()); float best=1e9; for(int r=0;r<5;r++){ CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1)); if(ms<best)best=ms;} printf("%-28s %8.3f ms  %7.1f GB/s\n",name,best,bytes/best/1e6); };
  const double useful = (double)N*N*N*8;
  for(int g : {2048, 8192, 65536})
  { char nm[64]; snprintf(nm,64,"lin_copy grid %d",g); run(nm, 2.0*n*8, [&]{ hipLaunchKernelGGL(lin_copy,dim3(g),dim3(256),0,0,(const double2*)a,(double2*)o,n/2); }); 
    snprintf(nm,64,"lin_2r1w grid %d",g); run(nm, 3.0*n*8, [&]{ hipLaunchKernelGGL(lin_2r1w,dim3(g),dim3(256),0,0,(const double2*)a,(const double2*)b,(double2*)o,n/2); }); }
  for(int CI : {32,64,513}){
    { const int RJ=4; int nb=((N+127)/128)*((N+4*RJ-1)/(4*RJ))*((N+CI-1)/CI); char nm[64]; snprintf(nm,64,"march_i RJ4 CI %d (%d blk)",CI,nb); run(nm,3.0*useful,[&]{ hipLaunchKernelGGL(march_i<4>,dim3(nb),dim3(256),0,0,a,b,o,N,pitch,plane,CI);}); }
    { const int RJ=8; int nb=((N+127)/128)*((N+4*RJ-1)/(4*RJ))*((N+CI-1)/CI); char nm[64]; snprintf(nm,64,"march_i RJ8 CI %d (%d blk)",CI,nb); run(nm,3.0*useful,[&]{ hipLaunchKernelGGL(march_i<8>,dim3(nb),dim3(256),0,0,a,b,o,N,pitch,plane,CI);}); }
    { const int RJ=4; int nb=((N+127)/128)*((N+4*RJ-1)/(4*RJ))*((N+CI-1)/CI); char nm[64]; snprintf(nm,64,"march_j RJ4 CJ %d (%d blk)",CI,nb); run(nm,3.0*useful,[&]{ hipLaunchKernelGGL(march_j<4>,dim3(nb),dim3(256),0,0,a,b,o,N,pitch,plane,CI);}); }
    { const int RJ=8; int nb=((N+127)/128)*((N+4*RJ-1)/(4*RJ))*((N+CI-1)/CI); char nm[64]; snprintf(nm,64,"march_j RJ8 CJ %d (%d blk)",CI,nb); run(nm,3.0*useful,[&]{ hipLaunchKernelGGL(march_j<8>,dim3(nb),dim3(256),0,0,a,b,o,N,pitch,plane,CI);}); }
  }
  return 0;
}
