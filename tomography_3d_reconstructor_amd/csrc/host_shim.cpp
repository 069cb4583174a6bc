// host_shim.cpp -- host-side pieces of libtomo_hip.so: ABI version, error strings and a CPU build of
// the per-cell marching-cubes evaluator (mc_cell.h, the very code the device runs) so the CPU-only test
// tier can check the MC33 decision logic against the golden cells without a GPU.
#include <stdint.h>
#include "../../include/tomo_hip.h"

#define MC_LUT_QUAL static const
#define MC_FN static inline
#include "mc_cell.h"

#define TOMO_API extern "C" __attribute__((visibility("default")))

TOMO_API int tomo_abi_version(void) { return 6; }

TOMO_API const char *tomo_error_string(int code)
{
    switch (code) {
    case TOMO_OK: return "ok";
    case TOMO_E_ARG: return "bad argument";
    case TOMO_E_LAUNCH: return "HIP launch error";
    case TOMO_E_SIZE: return "size exceeds index range";
    case TOMO_E_WORKSPACE: return "workspace too small";
    default: return "unknown error";
    }
}

TOMO_API int tomo_host_mc_cell(const float *h_v, double iso, int8_t *h_tris, int *h_uses_centre)
{
    if (!h_v || !h_tris || !h_uses_centre) return TOMO_E_ARG;
    double v[8];
    int index = 0;
    for (int i = 0; i < 8; i++) {
        v[i] = (double)h_v[i] - iso;
        if (v[i] > 0.0) index |= 1 << i;
    }
    *h_uses_centre = 0;
    if (index == 0 || index == 255) return 0;
    McTiling t = mc_cell_tiling(v, index);
    for (int i = 0; i < 3 * t.ntri; i++) h_tris[i] = t.tris[i];
    *h_uses_centre = t.centre;
    return t.ntri;
}

// vertex offsets along an edge / of the centre vertex, for the same tests
TOMO_API double tomo_host_mc_edge_offset(double va, double vb) { return mc_edge_offset(va, vb); }
TOMO_API void tomo_host_mc_centre_offset(const double *v, double *out3) { mc_centre_offset(v, out3, out3 + 1, out3 + 2); }
