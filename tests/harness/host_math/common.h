// TEST INFRASTRUCTURE (tests/test_device_math_on_host.py): stands in for csrc/common.h so that the PRODUCT's
// gaussian_math.h -- copied next to this file at test time, never edited -- compiles for the host with g++.  Only what
// that header needs: the qualifiers as no-ops, the C ABI's argument block, and the one gfx950 builtin it calls.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>

#include "hgs.h"

#define __device__
#define __host__
#define __forceinline__ inline
// v_rcp_f32 is accurate to 1 ulp; the double chain refines it by a Newton step (rcp_d), so a correctly rounded
// reciprocal here changes its result by ~1e-14 relative at most
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
struct float4 { float x, y, z, w; };           // (HIP's vector type, as far as the header uses it: member access)
using std::max;
using std::min;
