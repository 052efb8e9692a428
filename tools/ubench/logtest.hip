#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "/root/repo/koopman-online-updated-mpc_amd/csrc/plant_device.h"
__global__ void k(const double* x, double* y, int n) { int i = blockIdx.x*blockDim.x+threadIdx.x; if (i<n) y[i] = kmpc::kmpc_log(x[i]); }
int main() {
  const int n = 1<<20; double *hx = new double[n], *hy = new double[n]; 
  for (int i=0;i<n;++i) { double u = (double)rand()/RAND_MAX; hx[i] = (i%3==0) ? 1e-4 + u*1e-3 : (i%3==1 ? u*10 + 1e-9 : exp(40*u-20)); }
  double *dx,*dy; hipMalloc(&dx,n*8); hipMalloc(&dy,n*8); hipMemcpy(dx,hx,n*8,hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n/256), dim3(256), 0, 0, dx, dy, n); hipMemcpy(hy,dy,n*8,hipMemcpyDeviceToHost);
  double worst=0; for (int i=0;i<n;++i) { double r = log(hx[i]); double e = fabs(hy[i]-r)/fmax(fabs(r),1e-300); double ulp = fabs(hy[i]-r)/(nextafter(fabs(r),INFINITY)-fabs(r)+1e-320); if (ulp>worst) worst=ulp; }
  printf("worst error %.2f ulp\n", worst); return 0; }
