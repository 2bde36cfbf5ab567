// common.h -- shared device/host helpers for libacoss_mi355x (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/acoss_mi355x.h"

#define ACOSS_WAVE 64

namespace acoss {

// Thread-local last-error text (capi.hip owns the storage).
void set_error(const char *fmt, ...);

inline int hip_check(hipError_t e, const char *what)
{
    if (e == hipSuccess) return ACOSS_OK;
    set_error("%s: %s", what, hipGetErrorString(e));
    return ACOSS_EIO;
}

#define ACOSS_HIP(expr)                                   \
    do {                                                  \
        int rc__ = ::acoss::hip_check((expr), #expr);     \
        if (rc__ != ACOSS_OK) return rc__;                \
    } while (0)

// Launch check: kernels are asynchronous; only launch-time errors surface here.
inline int launch_check(const char *name)
{
    return hip_check(hipGetLastError(), name);
}

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// The strip kernels address a pair's result matrix with 32-bit byte offsets (buffer stores: wave-uniform row offset + lane
// offset, {crp,strip32}_*.hip), rows of the last, partly filled step included.  Host-side bound on the matrix a launch may
// be given: rows padded by one 64-row step, pitches of up to max_ny + 256 elements (acoss_plan_pairs pads to pitch_align).
inline bool strip_offsets_fit(int max_nx, int max_ny, int cell_bytes)
{
    return (int64_t)cell_bytes * ((int64_t)max_nx + 64) * ((int64_t)max_ny + 256) < 0x7fffffffLL;
}
// Compute units of the current device (256 on MI355X): the grid unit of the persistent kernels.  capi.hip.
int device_cus();

__host__ __device__ inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// CRPUtils.py:186-193 neighbour count (host side; kappa*ncols rounded half-to-even like np.round)
long nneighbs(double kappa, long ncols);

}  // namespace acoss
