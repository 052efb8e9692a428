// rollout_jit.hip -- ONE instantiation of the fused roll-out kernel as a plug-in of libkoopmpc.so.
//
// libkoopmpc.so carries the fused roll-out (rollout_kernel.hip: `steps` iterations of the reference loop, duffing.py:823-1012, in one
// launch) for the dimension sets BASELINE.json and the reference's scripts use.  The reference's dimensions are edit-in-source
// constants (duffing.py:66 Nlift, :632-633 MPCHorizon; Koopman_update.m:67, 70, 113 runs L = 10, N = 10), so every other set gets its
// instantiation when a handle of that set is created: rollout_plugin.hip compiles THIS file with
//
//   hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -shared -DKMPC_JIT_L=.. -DKMPC_JIT_N=.. -DKMPC_JIT_Q=..
//         -DKMPC_JIT_NW=4|8|16 -DKMPC_JIT_KS=-1|0|25 -DKMPC_JIT_IO32=0|1 -DKMPC_JIT_TERM=0|1  rollout_jit.hip -o <cache>/rollout_....so
//
// -- the compiler, the flags and the sources the built-in instantiations are made of --, keeps the shared object in the kernel cache
// and loads it with dlopen.  The plug-in is self-contained (no symbol of the library): the library decides the workgroup size
// (rollout_waves) and hands it over with the launch arguments.
#define KMPC_ROLLOUT_JIT_TU
#if !defined(KMPC_JIT_L) || !defined(KMPC_JIT_N) || !defined(KMPC_JIT_Q) || !defined(KMPC_JIT_NW) || !defined(KMPC_JIT_KS) || !defined(KMPC_JIT_IO32) || !defined(KMPC_JIT_TERM)
#error "rollout_jit.hip is compiled by rollout_plugin.hip with the KMPC_JIT_* dimension macros"
#endif
#include "rollout_kernel.hip"

extern "C" {
// layout version of RolloutArgs / StepArgs the plug-in was compiled against (kernels.h KMPC_PLUGIN_ABI)
int kmpc_rollout_plugin_abi(void) { return KMPC_PLUGIN_ABI; }
int kmpc_rollout_plugin_args_bytes(void) { return (int)sizeof(kmpc::RolloutArgs<double>); }
hipError_t kmpc_rollout_plugin_launch(const kmpc::RolloutArgs<double>* a, int waves, hipStream_t s) {
  if (!a || a->s.L != KMPC_JIT_L || a->s.N != KMPC_JIT_N || a->s.q != KMPC_JIT_Q || (a->io_f32 != 0) != (KMPC_JIT_IO32 != 0) || waves <= 0 ||
      (a->term_every > 0) != (KMPC_JIT_TERM != 0))
    return hipErrorInvalidValue;
#if KMPC_JIT_IO32
  return kmpc::launch_rollout_impl<KMPC_JIT_L, KMPC_JIT_N, KMPC_JIT_Q, float>(*a, s, waves);
#else
  return kmpc::launch_rollout_impl<KMPC_JIT_L, KMPC_JIT_N, KMPC_JIT_Q, double>(*a, s, waves);
#endif
}
}
