// wbc_k_general.hip — the general tick kernel wbc_tick_kernel<MODE, WARM, ORTH> (one instance per wavefront; tick / assemble / FK outputs).
#include "wbc_common.h"

namespace wbc {

// ------------------------------------------------------------------------------------------------
// kernels: single-wave workgroups, one per instance for the tick kernels (the QP / integrate kernels walk the batch with a
// grid-stride loop whose exit, b >= B, every wave reaches).
// ------------------------------------------------------------------------------------------------
// WARM: the variant that reads / writes working sets (warm start, KernelArgs.ws_in / ws_out); the cold variant carries none of it
// ORTH: the variant that carries contact_presolve_orth (chosen by launch_tick when a plan of the batch asks for it: the other
// variants keep their register allocation — with the extra code inlined the general kernel went from 198 VGPRs to 256 + spills)
template <int MODE, bool WARM = false, bool ORTH = false>
__global__ void __launch_bounds__(64, 2) wbc_tick_kernel(const KernelArgs A, const DevModel* __restrict__ models,
                                                         const WbcConfig* __restrict__ cfgs, const DevPlan* __restrict__ plans) {
  // models / cfgs are separate __restrict__ const parameters so that the compiler may read them with scalar loads
  // (as members of A it must assume the kernel's own stores clobber them: every access became a vector load + full wait).
  // ONE instance per single-wave workgroup, no loop: inside a persistent loop the compiler hoists hundreds of
  // "invariants" (polynomial coefficients, masks, addresses) out of the tick, spills them to scratch and reloads them
  // one by one with full memory waits (profiles/r01_phase_cycles_v6: 60k cycles in one atan2). The hardware's
  // workgroup dispatcher does the batch loop instead; other resident waves cover this wave's input latency.
  __shared__ Smem S;
  const int lane = threadIdx.x;
  const int b = blockIdx.x;
#ifdef WBC_PROFILE
  const unsigned long long t_entry = clock64();
#else
  const unsigned long long t_entry = 0;
#endif
  S.cl[lane] = 0.0;                            // zero padding (never written above entry 25)
  const bool has2 = A.in.trunk_target || A.in.prev_trunk_target || A.in.trunk_ref_euler || A.in.trunk_prev_rot ||
                    A.in.com_target || A.in.com_target_vel;
  const bool has3 = A.in.ee_ref_rot != nullptr;
  // the model index is wave-uniform: say so, or every M.* / cfg.* access becomes a vector load
  const int mid = model_index(A.in.model_id, b, A.n_models);
  const InRegs cur = load_inputs(A.in, A.dbg_alias ? 0 : b, lane, has2, has3);   // dbg_alias: diagnostic, every wave reads instance 0
  const LaneConst lc = load_lane_const(models[mid], cfgs[mid], lane);   // L1/L2-resident 3 KB table
  stage_inputs(S, cur, lane, has2, has3);
  WSYNC();
  process_instance<MODE, WARM, ORTH>(S, A, models[mid], cfgs[mid], plans[mid], lc, cur, b, lane, t_entry);
}

// One translation unit per PART (csrc/Makefile compiles this file once per part, in parallel): each part instantiates some of the kernel's
// variants; part 0 also holds the launcher and sees the other parts' variants as explicit-instantiation declarations.
#ifndef GENERAL_PART
#define GENERAL_PART -1      // -1: everything in one unit
#endif
#define KINST(...) template __global__ void wbc_tick_kernel<__VA_ARGS__>(const KernelArgs, const DevModel* __restrict__, const WbcConfig* __restrict__, const DevPlan* __restrict__);
#define KDECL(...) extern template __global__ void wbc_tick_kernel<__VA_ARGS__>(const KernelArgs, const DevModel* __restrict__, const WbcConfig* __restrict__, const DevPlan* __restrict__);
#if GENERAL_PART == 0 || GENERAL_PART == -1
KINST(MODE_TICK)
#endif
#if GENERAL_PART == 1 || GENERAL_PART == -1
KINST(MODE_TICK, true)
#elif GENERAL_PART == 0
KDECL(MODE_TICK, true)
#endif
#if GENERAL_PART == 2 || GENERAL_PART == -1
KINST(MODE_TICK, false, true)
#elif GENERAL_PART == 0
KDECL(MODE_TICK, false, true)
#endif
#if GENERAL_PART == 3 || GENERAL_PART == -1
KINST(MODE_ASSEMBLE)
KINST(MODE_FK)
#elif GENERAL_PART == 0
KDECL(MODE_ASSEMBLE)
KDECL(MODE_FK)
#endif
#undef KINST
#undef KDECL
#if GENERAL_PART <= 0
int launch_tick(const KernelArgs& a, int mode, int grid, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (mode == MODE_TICK && (a.ws_in || a.ws_out)) hipLaunchKernelGGL((wbc_tick_kernel<MODE_TICK, true>), dim3(grid), dim3(64), 0, s, a, a.models, a.cfgs, a.plans);
  else if (mode == MODE_TICK && a.presolve && a.presolve_orth == 2) hipLaunchKernelGGL((wbc_tick_kernel<MODE_TICK, false, true>), dim3(grid), dim3(64), 0, s, a, a.models, a.cfgs, a.plans);
  else if (mode == MODE_TICK) hipLaunchKernelGGL(wbc_tick_kernel<MODE_TICK>, dim3(grid), dim3(64), 0, s, a, a.models, a.cfgs, a.plans);
  else if (mode == MODE_ASSEMBLE) hipLaunchKernelGGL(wbc_tick_kernel<MODE_ASSEMBLE>, dim3(grid), dim3(64), 0, s, a, a.models, a.cfgs, a.plans);
  else hipLaunchKernelGGL(wbc_tick_kernel<MODE_FK>, dim3(grid), dim3(64), 0, s, a, a.models, a.cfgs, a.plans);
  return check_launch("tick");
}
int tick_lds_bytes() { return (int)sizeof(Smem); }
#endif

}  // namespace wbc
