// skm_common.h -- shared host/device helpers of libskm (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <vector>

#include "../../include/skm.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

#define SKM_MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// Optional per-kernel timing with HIP events recorded on the launch stream (bench.py's roofline).
struct SkmProfRec {
    const char *name;
    hipEvent_t a, b;
};

struct SkmCtx {
    int device;
    int cu_count;
    int prof_on;
    std::vector<SkmProfRec> prof;
    // address, on this ctx's GPU, of the file-scope scratch array of skm_gmm.hip (skm_gmm_estep_noalloc);
    // looked up once in skm_ctx_create with the device current
    void *scratch;
};

struct SkmProfScope {
    SkmCtx *ctx;
    hipStream_t st;
    hipEvent_t a, b;
    const char *name;
    bool on;
    SkmProfScope(SkmCtx *c, const char *n, hipStream_t s) : ctx(c), st(s), name(n), on(c && c->prof_on)
    {
        if (on) {
            on = hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess;
            if (on) (void)hipEventRecord(a, st);
        }
    }
    ~SkmProfScope()
    {
        if (on) {
            (void)hipEventRecord(b, st);
            ctx->prof.push_back(SkmProfRec{name, a, b});
        }
    }
};

void skm_set_error(const char *fmt, ...);

#define SKM_HIP(call)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            skm_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                          __LINE__);                                                       \
            return SKM_ERR_HIP;                                                            \
        }                                                                                  \
    } while (0)

#define SKM_REQUIRE(cond, ...)        \
    do {                              \
        if (!(cond)) {                \
            skm_set_error(__VA_ARGS__); \
            return SKM_ERR_ARG;       \
        }                             \
    } while (0)

static inline size_t skm_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Geometry of the packed GMM image and of the kernels, derived from (C, D) only.
struct GmmGeom {
    int C, D;
    int CT;     // component tiles of 32
    int CTpad;  // CT rounded up to a multiple of 4 (pass-1 block); padding comps have kc=-1e30
    int NCH;    // feature chunks of 4 = number of 4-step K groups (NG)
    int NP;     // full chunk pairs
    int tail;   // NCH odd: last group mixes x (lanes 0-31) and x^2 (lanes 32-63)
    int NJT;    // 32-wide feature tiles of the statistics GEMM, column D carries the ones
    int XS;     // LDS row stride (floats) of a staged frame tile
};

static inline GmmGeom skm_geom(int C, int D)
{
    GmmGeom g;
    g.C = C;
    g.D = D;
    g.CT = (C + 31) / 32;
    g.CTpad = (g.CT + 3) / 4 * 4;
    g.NCH = (D + 3) / 4;
    g.NP = g.NCH / 2;
    g.tail = g.NCH & 1;
    g.NJT = (D + 1 + 31) / 32;
    g.XS = g.NJT * 32 + 4;
    return g;
}
