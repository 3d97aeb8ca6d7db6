// glrm_testhooks.hip -- the engine's test hooks in ONE place, compiled twice (lowrankmodels.jl_amd/build.py):
//   libglrm_hip.so           the product library: every hook is a constant -- no environment variable can inject a failure, point the engine at
//                            another RCCL, or switch the link emulator on (these were env-gated code in the production library)
//   libglrm_hip_testing.so   the same objects with this file and csrc/glrm_multigpu.hip rebuilt under -DGLRM_HIP_TESTING: GLRM_HIP_TEST_FAIL_FINALIZE,
//                            GLRM_HIP_RCCL_LIB / GLRM_HIP_RCCL_ALLOW_SHARED (tests/stubs/rccl_stub.cpp) and the link emulator
//                            (GLRM_EXCHANGE_EMULATE_*) exist only there; tests and `bench.py --emulate-link-gbps` load it by name.
#include <cstdlib>

#include "glrm_engine.hpp"

// (not part of the C ABI of include/glrm_hip.h: how a test tells the two builds apart)
extern "C" int glrm_build_is_testing(void) {
#ifdef GLRM_HIP_TESTING
  return 1;
#else
  return 0;
#endif
}

int glrm_test_fail_finalize() {
#ifdef GLRM_HIP_TESTING
  const char* v = getenv("GLRM_HIP_TEST_FAIL_FINALIZE");
  return v && *v && atoi(v) != 0;
#else
  return 0;
#endif
}
