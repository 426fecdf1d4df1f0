// csrc/spg_internal.h — declarations shared between the HIP translation unit and the host code.
#pragma once
#include <cstddef>
#include "../../include/spg.h"

namespace spg {

// HIP backend (spg_kernels.hip)
int hip_backend_create(int device, spg_backend *out, char *errbuf, size_t errlen);
void hip_backend_destroy(spg_backend *b);
void *hip_backend_stream(spg_backend *b);
const char *hip_backend_error(spg_backend *b);
int hip_backend_launches(spg_backend *b);
void hip_backend_profile(spg_backend *b, int enable);
void hip_backend_profile_read(spg_backend *b, double *ms, double *bytes, long long *launches, long long *blankets);

}  // namespace spg
