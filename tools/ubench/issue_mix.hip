// micro-benchmark: issue rates of the instruction kinds of the step body, alone and mixed, for 1 / 2 / 4 / 8 waves per SIMD.
// Every wave runs `iters` rounds of a straight-line block of 32 (or 32 + 32) independent instructions.
// hipcc --offload-arch=gfx950 -O3 issue_mix.hip -o issue_mix.bin && ./issue_mix.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define R4(x) x x x x
template <int MODE> __global__ __launch_bounds__(1024) void k(double* out, int iters, unsigned long long* clk) {
  __shared__ double lds[2048];
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  double f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
  const double m = 0.999 + 1e-9 * threadIdx.x, c = 1e-3 + 1e-9 * threadIdx.x;
  double sm = 0.999;  // wave-uniform operand
  asm volatile("" : "+s"(sm));
  float g0 = threadIdx.x, g1 = g0 + 1, g2 = g0 + 2, g3 = g0 + 3, g4 = g0 + 4, g5 = g0 + 5, g6 = g0 + 6, g7 = g0 + 7;
  const float gm = 0.999f + 1e-6f * threadIdx.x, gc = 1e-3f + 1e-6f * threadIdx.x;
  unsigned s0 = 1, s1 = 2, s2 = 3, s3 = 4, s4 = 5, s5 = 6, s6 = 7, s7 = 8;
  lds[threadIdx.x] = f0; lds[threadIdx.x + 1024] = f1;
  __syncthreads();
  const unsigned la = (threadIdx.x & 1023) * 8;
  double l0 = 0, l1 = 0, l2 = 0, l3 = 0;
#define FMA3 "v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
#define FMA2 "v_fma_f64 %0, %0, %10, %9\n v_fma_f64 %1, %1, %10, %9\n v_fma_f64 %2, %2, %10, %9\n v_fma_f64 %3, %3, %10, %9\n v_fma_f64 %4, %4, %10, %9\n v_fma_f64 %5, %5, %10, %9\n v_fma_f64 %6, %6, %10, %9\n v_fma_f64 %7, %7, %10, %9\n"
#define MUL2 "v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n"
#define ADD2 "v_add_f64 %0, %0, %9\n v_add_f64 %1, %1, %9\n v_add_f64 %2, %2, %9\n v_add_f64 %3, %3, %9\n v_add_f64 %4, %4, %9\n v_add_f64 %5, %5, %9\n v_add_f64 %6, %6, %9\n v_add_f64 %7, %7, %9\n"
#define FDPP "v_fmac_f64_dpp %0, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %2, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %4, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %5, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %6, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %7, %8, %9 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
#define DARGS : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(m), "v"(c), "s"(sm)
#define F32 "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
#define MOV "v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n"
#define DPP "v_mov_b32_dpp %0, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define GARGS : "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g3), "+v"(g4), "+v"(g5), "+v"(g6), "+v"(g7) : "v"(gm), "v"(gc)
#define SAL "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n s_add_u32 %4, %4, 1\n s_add_u32 %5, %5, 1\n s_add_u32 %6, %6, 1\n s_add_u32 %7, %7, 1\n"
#define SARGS : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) : : "scc"
#define LDSR "ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:8192\n ds_read_b64 %2, %4\n ds_read_b64 %3, %4 offset:8192\n ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:8192\n ds_read_b64 %2, %4\n ds_read_b64 %3, %4 offset:8192\n"
#define LARGS : "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3) : "v"(la)
  for (int i = 0; i < iters; ++i) {
    if constexpr (MODE == 0) asm volatile(R4(FMA3) DARGS);
    if constexpr (MODE == 1) asm volatile(R4(FMA2) DARGS);
    if constexpr (MODE == 2) asm volatile(R4(MUL2) DARGS);
    if constexpr (MODE == 3) asm volatile(R4(ADD2) DARGS);
    if constexpr (MODE == 4) asm volatile(R4(F32) GARGS);
    if constexpr (MODE == 5) asm volatile(R4(MOV) GARGS);
    if constexpr (MODE == 6) asm volatile(R4(DPP) GARGS);
    if constexpr (MODE == 7) asm volatile(R4(SAL) SARGS);
    if constexpr (MODE == 8) asm volatile(R4(LDSR) "s_waitcnt lgkmcnt(0)\n" LARGS);
    if constexpr (MODE == 9) {  // 32 f64 FMA + 32 SALU, alternating blocks of 8
      asm volatile(FMA2 DARGS); asm volatile(SAL SARGS); asm volatile(FMA2 DARGS); asm volatile(SAL SARGS);
      asm volatile(FMA2 DARGS); asm volatile(SAL SARGS); asm volatile(FMA2 DARGS); asm volatile(SAL SARGS);
    }
    if constexpr (MODE == 10) {  // 32 f64 FMA + 32 ds_read_b64
      asm volatile(LDSR LARGS); asm volatile(FMA2 DARGS); asm volatile(LDSR LARGS); asm volatile(FMA2 DARGS);
      asm volatile(LDSR LARGS); asm volatile(FMA2 DARGS); asm volatile(LDSR LARGS); asm volatile(FMA2 DARGS);
      asm volatile("s_waitcnt lgkmcnt(0)\n" ::: "memory");
    }
    if constexpr (MODE == 11) {  // 32 f64 FMA + 32 v_mov_b32
      asm volatile(FMA2 DARGS); asm volatile(MOV GARGS); asm volatile(FMA2 DARGS); asm volatile(MOV GARGS);
      asm volatile(FMA2 DARGS); asm volatile(MOV GARGS); asm volatile(FMA2 DARGS); asm volatile(MOV GARGS);
    }
    if constexpr (MODE == 13) asm volatile(R4(FDPP) DARGS);
    if constexpr (MODE == 12) asm volatile(R4("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n"));
  }
  if (clk && blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_amdgcn_s_memtime() - c0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
  out[blockIdx.x * 1024 + threadIdx.x] = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + (double)(s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7) + l0 + l1 + l2 + l3 +
                                         (double)(g0 + g1 + g2 + g3 + g4 + g5 + g6 + g7);
}
static const char* names[] = {"v_fma_f64, 3 VGPR sources", "v_fma_f64, 2 VGPR + 1 SGPR source", "v_mul_f64", "v_add_f64", "v_fma_f32", "v_mov_b32", "v_mov_b32_dpp row_shr",
                              "s_add_u32", "ds_read_b64", "32 v_fma_f64 + 32 s_add_u32", "32 v_fma_f64 + 32 ds_read_b64", "32 v_fma_f64 + 32 v_mov_b32", "s_nop 0", "v_fmac_f64_dpp row_newbcast"};
template <int MODE> void run(double* out, unsigned long long* clk) {
  const int iters = 4000;
  for (int wps = 1; wps <= 8; wps *= 2) {  // waves per SIMD
    const int threads = wps >= 4 ? 1024 : 256 * wps, blocks = wps == 8 ? 512 : 256;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters, clk);
    (void)hipEventRecord(e0);
    for (int j = 0; j < 5; ++j) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters, clk);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    unsigned long long hc[2]; (void)hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
    const double ghz = (double)hc[0] / (double)hc[1] * 0.1;
    const int per_round = MODE >= 9 && MODE <= 11 ? 64 : 32;
    const double ns = ms * 1e6 / iters / per_round / wps;  // per instruction and SIMD
    printf("%-36s %d waves/SIMD: %6.2f ns = %5.2f cycles per instruction and SIMD   (clock %.2f GHz)\n", names[MODE], wps, ns, ns * ghz, ghz);
    fflush(stdout);
  }
}
int main() {
  double* out; (void)hipMalloc(&out, 512 * 1024 * 8);
  unsigned long long* clk; (void)hipMalloc(&clk, 16);
  for (int i = 0; i < 40; ++i) hipLaunchKernelGGL(k<0>, dim3(256), dim3(1024), 0, 0, out, 20000, (unsigned long long*)nullptr);  // clock ramp
  (void)hipDeviceSynchronize();
  if (getenv("ALL")) { run<0>(out, clk); run<1>(out, clk); run<2>(out, clk); run<3>(out, clk); run<4>(out, clk); run<5>(out, clk); run<6>(out, clk); }
  if (getenv("ALL")) { run<7>(out, clk); run<8>(out, clk); run<9>(out, clk); run<10>(out, clk); run<11>(out, clk); run<12>(out, clk); }
  run<1>(out, clk); run<13>(out, clk);
  return 0;
}
