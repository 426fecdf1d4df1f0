// stubs of the HIP translation units for a sanitizer build of the host code (CPU only, injected backend)
#include "spg_internal.h"
namespace spg {
int hip_backend_create(int, spg_backend *, char *, size_t) { return SPG_ENODEV; }
void hip_backend_destroy(spg_backend *) {}
void *hip_backend_stream(spg_backend *) { return nullptr; }
const char *hip_backend_error(spg_backend *) { return ""; }
int hip_backend_launches(spg_backend *) { return 0; }
int hip_backend_device(spg_backend *) { return -1; }
void hip_backend_profile(spg_backend *, int) {}
void hip_backend_profile_read(spg_backend *, double *, double *, long long *, long long *) {}
void hip_backend_profile_read_worker(spg_backend *, double *, double *, long long *, long long *) {}
void hip_backend_profile_read_big(spg_backend *, double *, double *, long long *, int *) {}
int hip_backend_end_of_call(spg_backend *) { return 0; }
int hip_stream_open(spg_backend *, int, int, int, StreamPort *) { return 1; }
void hip_stream_close(spg_backend *, const StreamPort *, double, long long) {}
int hip_la_test(int, int, int, int, int, int, double *, int, int, double *, int, int, double *, int, int, int *) { return SPG_ENODEV; }
int hip_big_glc_dense(void *, const DenseGraphIn &, int, int, int, int64_t, double *, int, int, double *, double *, char *, size_t) { return SPG_ENODEV; }
int hip_dense_information(void *, const DenseGraphIn &, int, double *, char *, size_t) { return SPG_ENODEV; }
int hip_dense_covariance(void *, const DenseGraphIn &, int, double *, char *, size_t) { return SPG_ENODEV; }
int rccl_get_unique_id(void *, char *, size_t) { return SPG_ENODEV; }
int rccl_comm_create(int, int, int, const void *, void **, char *, size_t) { return SPG_ENODEV; }
int rccl_allgather_f64(void *, void *, int64_t, int64_t, void *, char *, size_t) { return SPG_ENODEV; }
void rccl_comm_destroy(void *) {}
int hip_dense_kld(void *, const DenseGraphIn &, const DenseGraphIn &, int, int, const int64_t *, const int64_t *, double *, double *, char *, size_t) { return SPG_ENODEV; }
int hip_dense_optimize(void *, const DenseGraphIn &, int, int, double *, double *, char *, size_t) { return SPG_ENODEV; }
int hip_sparse_optimize(void *, const DenseGraphIn &, int, int, double *, double *, double *, char *, size_t) { return SPG_ENODEV; }
int hip_sparse_kld(void *, const DenseGraphIn &, const DenseGraphIn &, const uint8_t *, const int32_t *, const int32_t *, int, const int64_t *, const int64_t *, double *, double *, double *, char *, size_t) { return SPG_ENODEV; }
}
