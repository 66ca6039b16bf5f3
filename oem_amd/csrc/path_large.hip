// path_large.hip -- eigenvalue + path engine for p > SMALL_P_MAX (multi-workgroup, multi-launch).
#include "common.hpp"

namespace oemgpu {

size_t path_large_work_doubles(int p, int nsteps) { (void)nsteps; return (size_t)p * 8 + 1024; }

int run_path_large(hipStream_t s, const PathArgs &a, double *host_scratch)
{
    (void)s; (void)a; (void)host_scratch;
    set_error("p = %d > %d: the large-p engine is not built yet", a.p, SMALL_P_MAX);
    return OEMGPU_ERR_UNSUPPORTED;
}

}  // namespace oemgpu
