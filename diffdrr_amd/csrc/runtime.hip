// runtime.hip -- shared host-side pieces of the C ABI (include/diffdrr_hip.h): error string,
// argument checks, launch geometry of the per-ray kernels, ABI version.
#include "runtime.h"

namespace ddrr_rt {

namespace {
thread_local char g_err[512] = "";
}

int fail(int code, const char *what) {
    snprintf(g_err, sizeof(g_err), "%s", what);
    return code;
}
int fail_hip(hipError_t e, const char *where) {
    snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
    return (int)e;
}
int finish(const char *where) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, where);
    return 0;
}
const char *last_error() { return g_err; }

int g_xcd_swizzle = 1;

int check_common(const float *volume, int dx, int dy, int dz, const float *source, int src_n,
                 const float *target, int B, int N) {
    if (!volume || !source || !target) return fail(-1, "null volume/source/target pointer");
    if (dx < 1 || dy < 1 || dz < 1) return fail(-1, "volume dims must be >= 1");
    if ((long)dx * dy * dz > (1L << 30)) return fail(-1, "volume larger than 2^30 voxels");
    if (B < 0 || N < 0) return fail(-1, "negative batch or ray count");
    if (src_n != 1 && src_n != N) return fail(-1, "src_n must be 1 or N");
    return 0;
}

RayArgs make_args(const float *volume, int dx, int dy, int dz, const float *source, int src_n,
                  const float *target, const float *img, int B, int N, float shift, float eps,
                  int det_h, int det_w, int tile_h, int tile_w) {
    RayArgs p;
    p.vol = volume;
    p.D = Dims{dx, dy, dz};
    p.source = source;
    p.src_n = src_n;
    p.target = target;
    p.img = img;
    p.B = B;
    p.N = N;
    p.shift = shift;
    p.eps = eps;
    p.tm = make_tilemap(N, det_h, det_w, tile_h, tile_w);
    p.total_waves = B * p.tm.waves_per_pose;
    p.xcd_swizzle = g_xcd_swizzle;
    return p;
}

}  // namespace ddrr_rt

extern "C" {

int ddrr_abi_version(void) { return DDRR_ABI_VERSION; }
const char *ddrr_last_error(void) { return ddrr_rt::last_error(); }

#if defined(DDRR_EXPERIMENTS) || defined(DDRR_BRICK_PROFILE)
// tools/ builds only: 0/1 XCD-contiguous workgroup mapping of the per-ray kernels.
int ddrr_set_xcd_swizzle(int on) {
    int old = ddrr_rt::g_xcd_swizzle;
    ddrr_rt::g_xcd_swizzle = on ? 1 : 0;
    return old;
}
#endif

}  // extern "C"
