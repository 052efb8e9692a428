// rollout_kernel_io32.hip -- the fused roll-out with float32 panels around the float64 state (row g2: BASELINE configs[1] names fp32,
// SURVEY G6): the IOT = float instantiations of rollout_kernel for the register-state dimension sets, compiled as a translation unit
// of their own (in parallel with the float64 ones).
#define KMPC_ROLLOUT_IO32_TU
#include "rollout_kernel.hip"
