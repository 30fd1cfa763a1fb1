// kernels.h — internal launch functions of libesmdiff_hip.so (gfx950 only).
// Every function enqueues on `stream` and returns the hipError_t of the launch.
#pragma once
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/esmdiff_hip.h"

#include <map>
#include <mutex>
#include <utility>

typedef uint16_t bf16_t;  // raw bfloat16 bits

// Types shared by the two operand-type builds of the kernels (namespace ed = bf16, ed16 = f16, see ed_half.h).
struct EdGemmWorkspace {
  float* partial;
  size_t partial_floats;
};
struct EdGemmPartials {
  const float* p;
  int S;
  int64_t stride;
};

#include "kernels_ns.inc"
