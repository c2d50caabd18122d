// dev_launch.h -- what the launchers of both kernel translation units share: grid size, the (task, mode) ->
// template instantiation switch, the LEAN / tuned predicates.  Device code of copterstep_kernels.hip and
// copterstep_serve.hip (included there, inside the cs namespace's anonymous part); not a stand-alone header.
#pragma once

namespace cs {
namespace {

inline int grid_for(uint32_t n) { return (int)((n + kBlock - 1) / kBlock); }

// (task, mode) -> template instantiation
#define CS_CASE3(FN, TASK, ...)                                         \
  case TASK * 3 + CS_STATE_F32G:                                        \
    return FN<TASK, CS_STATE_F32G>(__VA_ARGS__);                        \
  case TASK * 3 + CS_STATE_F32_RN:                                      \
    return FN<TASK, CS_STATE_F32_RN>(__VA_ARGS__);                      \
  case TASK * 3 + CS_STATE_F64:                                         \
    return FN<TASK, CS_STATE_F64>(__VA_ARGS__);
#define CS_DISPATCH(FN, ...)                      \
  switch (task * 3 + mode) {                      \
    CS_CASE3(FN, CS_TASK_LANDER3D, __VA_ARGS__)   \
    CS_CASE3(FN, CS_TASK_HOVER3D, __VA_ARGS__)    \
    CS_CASE3(FN, CS_TASK_LANDER2D, __VA_ARGS__)   \
    CS_CASE3(FN, CS_TASK_LANDER1D, __VA_ARGS__)   \
    CS_CASE3(FN, CS_TASK_HOVER2D, __VA_ARGS__)    \
    CS_CASE3(FN, CS_TASK_HOVER1D, __VA_ARGS__)    \
    default:                                      \
      return hipErrorInvalidValue;                \
  }
static_assert(CS_STATE_F32G == 0 && CS_STATE_F32_RN == 1 && CS_STATE_F64 == 2, "dispatch index");
#define CS_MODE_LAUNCH(KERNEL, ...)                                                               \
  do {                                                                                            \
    const dim3 grid(grid_for(s.n)), block(kBlock);                                                \
    if (mode == CS_STATE_F32G)                                                                    \
      hipLaunchKernelGGL((KERNEL<CS_STATE_F32G>), grid, block, 0, stream, __VA_ARGS__);           \
    else if (mode == CS_STATE_F32_RN)                                                             \
      hipLaunchKernelGGL((KERNEL<CS_STATE_F32_RN>), grid, block, 0, stream, __VA_ARGS__);         \
    else if (mode == CS_STATE_F64)                                                                \
      hipLaunchKernelGGL((KERNEL<CS_STATE_F64>), grid, block, 0, stream, __VA_ARGS__);            \
    else                                                                                          \
      return hipErrorInvalidValue;                                                                \
    return hipGetLastError();                                                                     \
  } while (0)

// LEAN = the common configuration: the optional features are compiled out of the kernel (step_body)
inline bool lean_config(const DevConst& c, const DevState& s) {
  return c.autoreset != CS_AUTORESET_SAME_STEP && !c.stats && !c.tl_trunc && s.veh == nullptr && !c.gyro &&
         !c.act_f32 && !c.ticks;
}

// The headline combinations get every specialised instantiation of the lean kernel; the others one
// generic lean build (keeps the code object and its build time in bounds).
constexpr bool is_tuned(int task, int mode) {
  return (task == CS_TASK_LANDER3D || task == CS_TASK_HOVER3D) && mode == CS_STATE_F32G;
}

}  // namespace
}  // namespace cs
