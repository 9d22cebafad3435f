// skm_gmm.hip -- diagonal-GMM E-step / Baum-Welch statistics for gfx950 (MI355X).
//
// Replaces (reference = SIDEKIT 1.3.2 fork, paths relative to its root):
//   sidekit/mixture.py:461-487   compute_log_posterior_probabilities  (2 GEMMs T x D x C)
//   sidekit/mixture.py:52-64     sum_log_probabilities                (row logsumexp, posteriors)
//   sidekit/mixture.py:537-568   _expectation                         (N, F, S accumulation)
//   sidekit/statserver.py:659-667 accumulate_stat body                (per-session N, F)
//   sidekit/mixture.py:631-632   _init moments
//
// Formulation.  With x~ = [x, x^2] and per component the packed columns
//   Wlin = log2e * mu*invcov,  Wsq = -1/2 log2e * invcov,  kc = log2e * (log w + log cst - 1/2 sum mu^2 invcov)
// the base-2 logit is  l[t,c] = kc[c] + x~[t,:] . W[:,c]  (one f32 MFMA GEMM, K = 2D).
// gamma[t,c] = exp2(l[t,c] - lse[t]) needs the row logsumexp over ALL components, while the
// statistics sum gamma over ALL frames, so the path runs as two sweeps over the frames:
//   sweep 1  k_lse_partial : per (component block, frame) partial (max, sum exp2) -> k_lse_combine -> lse[t]
//   sweep 2  k_stats       : recompute l - lse, gamma = exp2(.), [N|F|S] += gamma^T [1, x, x^2]  (second MFMA GEMM,
//                            the accumulator tile of GEMM 1 is fed straight back as the A operand of GEMM 2)
// Both sweeps keep their component block's W resident in LDS for the life of the workgroup, stage 32-frame
// tiles of X through wave-private LDS, and never write a T x C matrix to HBM.
//
// MFMA: v_mfma_f32_32x32x2_f32 (exact f32, 64 cycles/SIMD).  Operand maps (lane l, c = l&31, h = l>>5):
//   A[i=c][k=h], B[k=h][j=c], D[row = (r&3) + 8*(r>>2) + 4*h][col = c], r = accumulator register 0..15.
#include "skm_common.h"

#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

// ------------------------------------------------------------------------------------------------
// error handling
// ------------------------------------------------------------------------------------------------
// csrc/skm_gmm_b3.hip: the split-precision (bf16 x 3) variant of the fat E-step kernel
size_t skm_estep_b3_wb_bytes(int CTpad);
int skm_estep_b3_launch(SkmCtx *ctx, const float *X, long T, int D, int CTpad, int NCH, int NP, const float *wp,
                        const float *kc, void *wb, double *slabs, double *llk_part, int n_llk, int nblk, int flush_tiles,
                        int n_cb, float *pm, float *ps, float *lse, hipStream_t st);

static thread_local char g_err[512] = "";

void skm_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *skm_last_error(void) { return g_err; }
extern "C" int skm_version(void) { return 101; }

// Scratch of skm_gmm_estep_noalloc: the E-step workspace of the single-sweep kernels (per-wave float64 slabs,
// log-likelihood partials, team hand-off rows and flags) as a file-scope device array, so that the entry point
// needs no workspace argument and allocates nothing.  Every GPU the library runs on holds its own copy until the
// process ends.  In host code this NAME is only a handle for hipGetSymbolAddress: the address used everywhere
// (kernel arguments, hipMemsetAsync) is SkmCtx::scratch, looked up once per device in skm_ctx_create.
#define SKM_SCRATCH_BYTES ((size_t)144 << 20)
static __device__ __align__(256) unsigned char scratch[SKM_SCRATCH_BYTES];

extern "C" int skm_ctx_create(int device, void **ctx)
{
    SKM_REQUIRE(ctx != nullptr, "skm_ctx_create: ctx is NULL");
    int n = 0;
    SKM_HIP(hipGetDeviceCount(&n));
    SKM_REQUIRE(device >= 0 && device < n, "skm_ctx_create: device %d out of range (%d visible)", device, n);
    hipDeviceProp_t prop;
    SKM_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        skm_set_error("skm_ctx_create: device %d is %s, this library is built for gfx950 only", device,
                      prop.gcnArchName);
        return SKM_ERR_UNSUPPORTED;
    }
    // this GPU's copy of the scratch array: its address is per device, so it is taken with `device` current
    int prev = 0;
    SKM_HIP(hipGetDevice(&prev));
    SKM_HIP(hipSetDevice(device));
    void *sp = nullptr;
    const hipError_t se = hipGetSymbolAddress(&sp, HIP_SYMBOL(scratch));
    (void)hipSetDevice(prev);
    if (se != hipSuccess || sp == nullptr) {
        skm_set_error("skm_ctx_create: hipGetSymbolAddress(scratch) failed on device %d: %s", device,
                      hipGetErrorString(se));
        return SKM_ERR_HIP;
    }
    SkmCtx *c = new SkmCtx();
    c->device = device;
    c->cu_count = prop.multiProcessorCount;
    c->prof_on = 0;
    c->scratch = sp;
    *ctx = c;
    return SKM_OK;
}

extern "C" int skm_ctx_destroy(void *ctx)
{
    delete static_cast<SkmCtx *>(ctx);
    return SKM_OK;
}

extern "C" int skm_ctx_cu_count(void *ctx) { return ctx ? static_cast<SkmCtx *>(ctx)->cu_count : 0; }

extern "C" int skm_prof_enable(void *ctx_, int on)
{
    SkmCtx *ctx = static_cast<SkmCtx *>(ctx_);
    SKM_REQUIRE(ctx != nullptr, "skm_prof_enable: NULL ctx");
    ctx->prof_on = on ? 1 : 0;
    return SKM_OK;
}

// Waits for every recorded event, writes one line "name count total_ms" per kernel name into buf
// and clears the records.  Returns the number of bytes written (excluding the NUL), <0 on error.
extern "C" long skm_prof_read(void *ctx_, char *buf, size_t cap)
{
    SkmCtx *ctx = static_cast<SkmCtx *>(ctx_);
    SKM_REQUIRE(ctx && buf && cap > 0, "skm_prof_read: bad argument");
    struct Agg {
        const char *name;
        long n;
        double ms;
    };
    std::vector<Agg> agg;
    for (SkmProfRec &r : ctx->prof) {
        float ms = 0.f;
        SKM_HIP(hipEventSynchronize(r.b));
        SKM_HIP(hipEventElapsedTime(&ms, r.a, r.b));
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
        bool found = false;
        for (Agg &g : agg)
            if (strcmp(g.name, r.name) == 0) {
                g.n++;
                g.ms += ms;
                found = true;
            }
        if (!found) agg.push_back(Agg{r.name, 1, ms});
    }
    ctx->prof.clear();
    size_t off = 0;
    buf[0] = 0;
    for (Agg &g : agg) {
        int w = snprintf(buf + off, cap - off, "%s %ld %.6f\n", g.name, g.n, g.ms);
        if (w < 0 || (size_t)w >= cap - off) break;
        off += (size_t)w;
    }
    return (long)off;
}

#define SKM_GMM_MAIN_TU
#include "skm_gmm_kernels.h"

// defined in skm_gmm_team.hip / skm_gmm_seg.hip / skm_gmm_seg2.hip / skm_gmm_seg3.hip
SKM_ESTEP_TEAM_INSTANCES(SKM_DECLARE_ESTEP_TEAM)
SKM_SEG_TEAM_INSTANCES_A(SKM_DECLARE_SEG_TEAM)
SKM_SEG_TEAM_INSTANCES_B(SKM_DECLARE_SEG_TEAM)
SKM_SEG_TEAM_INSTANCES_C(SKM_DECLARE_SEG_TEAM)

// Baum-Welch team path: a hand-off poll that ran out raised *err and the team carried on with stale partial rows;
// its statistics are not valid, so every output row becomes NaN (the EM path does the same to the
// log-likelihood in k_reduce_slabs).  Keeps skm_bw_stats asynchronous: no host read of the flag.
__global__ __launch_bounds__(256) void k_bw_poison(const int *__restrict__ err, double *__restrict__ stat0, long n0,
                                                   double *__restrict__ stat1, long n1)
{
    if (*err == 0) return;
    const double nan = __builtin_nan("");
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n0; i += (long)gridDim.x * blockDim.x) stat0[i] = nan;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n1; i += (long)gridDim.x * blockDim.x) stat1[i] = nan;
}

// Sum the per-wave f64 slabs in a fixed order into out = [N | F | S | llk].  A workgroup covers 16 consecutive
// outputs with 16 threads each: thread (o, q) sums slots q, q+16, ... (8 loads in flight, combined in a fixed
// order), the 16 partial sums meet in LDS.  Workgroup 0 also sums the log-likelihood partials the same way.
#define SKM_RED_OUT 16
__global__ __launch_bounds__(256) void k_reduce_slabs(const double *__restrict__ slabs, int n_cb, int n_fs, int NW,
                                                      int NCT, int NJT, int NF, int C, int D,
                                                      double *__restrict__ out, int accumulate,
                                                      const double *__restrict__ llk_part, int n_llk,
                                                      const int *__restrict__ err = nullptr)
{
    __shared__ double red[256];
    const int tid = threadIdx.x;
    if (blockIdx.x == 0) {
        double l = 0.0;
        for (int i = tid; i < n_llk; i += 256) l += llk_part[i];
        red[tid] = l;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            __syncthreads();
        }
        if (tid == 0) {
            l = red[0];
            if (err && *err) l = __builtin_nan("");   // a team hand-off timed out: the statistics are not valid
            double *p = out + C + 2L * C * D;
            *p = accumulate ? (*p + l) : l;
        }
        __syncthreads();
    }
    const int NFW = NF * 32;
    const int o = tid & (SKM_RED_OUT - 1), q0 = tid / SKM_RED_OUT;
    constexpr int NQ = 256 / SKM_RED_OUT;
    const long idx = (long)blockIdx.x * SKM_RED_OUT + o;
    const int comp = (int)(idx / NFW), col = (int)(idx - (long)comp * NFW);
    double *dst = nullptr;
    if (comp < C) {
        if (col < NJT * 32) {
            if (col < D)
                dst = out + C + (long)comp * D + col;
            else if (col == D)
                dst = out + comp;
        } else {
            const int d = col - NJT * 32;
            if (d < D) dst = out + C + (long)C * D + (long)comp * D + d;
        }
    }
    double sum = 0.0;
    if (dst) {
        const int cb = (comp >> 5) / NCT;
        const int cl = comp - cb * NCT * 32;
        const long slab_elems = (long)NCT * 32 * NFW;
        // slot (fs, w) of this component block lives at ((fs * n_cb + cb) * NW + w)
        const double *src = slabs + (long)cl * NFW + col;
        const int n_slots = n_fs * NW;
        double part[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int i = q0;
        for (; i + 7 * NQ < n_slots; i += 8 * NQ) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int q = i + u * NQ, fs = q / NW, w = q - fs * NW;
                part[u] += src[(((long)fs * n_cb + cb) * NW + w) * slab_elems];
            }
        }
        for (int u = 0; i < n_slots; i += NQ, ++u) {
            const int fs = i / NW, w = i - fs * NW;
            part[u] += src[(((long)fs * n_cb + cb) * NW + w) * slab_elems];
        }
        sum = ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
    }
    red[tid] = sum;
    __syncthreads();
    if (q0 == 0 && dst) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) t += red[q * SKM_RED_OUT + o];
        *dst = accumulate ? (*dst + t) : t;
    }
}

// ------------------------------------------------------------------------------------------------
// column moments (sum x, sum x^2) -- HBM-bound
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_moments(const float *__restrict__ X, long T, int D, double *__restrict__ part)
{
    // each wave owns a contiguous range of rows; lane handles columns lane, lane+64, ...
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long wslot = (long)blockIdx.x * (blockDim.x >> 6) + wave;
    const long nslots = (long)gridDim.x * (blockDim.x >> 6);
    const long ra = T * wslot / nslots, rb = T * (wslot + 1) / nslots;
    double *dst = part + wslot * 2L * D;
    for (int d0 = 0; d0 < D; d0 += 64) {
        const int d = d0 + lane;
        double s1 = 0.0, s2 = 0.0;
        if (d < D) {
            for (long t = ra; t < rb; ++t) {
                const double v = (double)X[t * D + d];
                s1 += v;
                s2 = fma(v, v, s2);
            }
            dst[d] = s1;
            dst[D + d] = s2;
        }
    }
}

__global__ void k_moments_reduce(const double *__restrict__ part, int nslots, int D, double *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * D) return;
    double s = 0.0;
    for (int k = 0; k < nslots; ++k) s += part[(long)k * 2 * D + i];
    out[i] = s;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static const double LOG2E = 1.4426950408889634074;

extern "C" size_t skm_gmm_packed_bytes(int C, int D)
{
    if (C <= 0 || D <= 0) return 0;
    const GmmGeom g = skm_geom(C, D);
    return ((size_t)g.CTpad * g.NCH * 256 + (size_t)g.CTpad * 32) * sizeof(float);
}

// Image relative to a reference Gaussian R = N(0, diag(1 / riv)) (skm_gmm_pack_ref, include/skm.h): the square rows
// hold -1/2 (invcov - riv) and kc drops log R's constant, so the kernels' logit is log2(w_c N_c(x) / R(x)).  A
// per-frame term common to all components changes neither the posteriors nor N / F / S; what it changes is the size
// of the float32 partial sums (the x^2 / v part of every logit, about D/2 nats, is gone), i.e. their rounding error.
extern "C" int skm_gmm_pack(void *ctx, const double *w, const double *mu, const double *ic, int C, int D,
                            void *packed_dev, void *stream)
{
    return skm_gmm_pack_ref(ctx, w, mu, ic, nullptr, C, D, packed_dev, stream);
}

extern "C" int skm_gmm_pack_ref(void *ctx, const double *w, const double *mu, const double *ic, const double *riv,
                                int C, int D, void *packed_dev, void *stream)
{
    SKM_REQUIRE(ctx && w && mu && ic && packed_dev, "skm_gmm_pack: NULL argument");
    SKM_REQUIRE(C > 0 && D > 0, "skm_gmm_pack: bad shape C=%d D=%d", C, D);
    const GmmGeom g = skm_geom(C, D);
    const size_t nW = (size_t)g.CTpad * g.NCH * 256, nK = (size_t)g.CTpad * 32;
    std::vector<float> img(nW + nK, 0.f);
    float *wp = img.data(), *kc = img.data() + nW;
    for (int tile = 0; tile < g.CTpad; ++tile)
        for (int grp = 0; grp < g.NCH; ++grp)
            for (int lane = 0; lane < 64; ++lane) {
                const int comp = tile * 32 + (lane & 31), h = lane >> 5;
                for (int i = 0; i < 4; ++i) {
                    int d, kind;
                    if (grp < 2 * g.NP) {
                        d = 8 * (grp >> 1) + 4 * h + i;
                        kind = grp & 1;
                    } else {
                        d = 8 * g.NP + i;
                        kind = h;
                    }
                    double v = 0.0;
                    if (comp < C && d < D) {
                        const double icv = ic[(size_t)comp * D + d];
                        v = kind ? (-0.5 * LOG2E * (icv - (riv ? riv[d] : 0.0))) : (LOG2E * mu[(size_t)comp * D + d] * icv);
                    }
                    wp[(((size_t)tile * g.NCH + grp) * 64 + lane) * 4 + i] = (float)v;
                }
            }
    const double half_d_log2pi = riv ? 0.0 : 0.5 * D * log(2.0 * M_PI);
    for (int comp = 0; comp < g.CTpad * 32; ++comp) {
        double v = -1.0e30;
        if (comp < C) {
            double logdet_inv = 0.0, quad = 0.0;
            for (int d = 0; d < D; ++d) {
                const double icv = ic[(size_t)comp * D + d], m = mu[(size_t)comp * D + d];
                logdet_inv += (riv && riv[d] > 0.0) ? log(icv / riv[d]) : log(icv);
                quad += m * m * icv;
            }
            v = LOG2E * (log(w[comp]) + 0.5 * logdet_inv - half_d_log2pi - 0.5 * quad);
            if (!(v > -1.0e30)) v = -1.0e30;  // w = 0 or NaN: component never wins
        }
        kc[comp] = (float)v;
    }
    SKM_HIP(hipMemcpyAsync(packed_dev, img.data(), img.size() * sizeof(float), hipMemcpyHostToDevice,
                           (hipStream_t)stream));
    SKM_HIP(hipStreamSynchronize((hipStream_t)stream));
    return SKM_OK;
}

// ------------------------------------------------------------------------------------------------
// M-step + re-pack on the device: the statistics of an EM iteration never visit the host.
//   params = [ w (C) | mu (C*D) | invcov (C*D) | sum N (1) ]  float64
// ------------------------------------------------------------------------------------------------
// sum of N[C] in a fixed order (one workgroup), and the iteration's mean log-likelihood llk / sum N
__global__ __launch_bounds__(256) void k_mstep_sum(const double *__restrict__ stats, int C, int D, double *__restrict__ params,
                                                    double *__restrict__ llk_mean)
{
    __shared__ double part[256];
    double s = 0.0;
    for (int c = threadIdx.x; c < C; c += 256) s += stats[c];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) part[threadIdx.x] += part[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        params[(size_t)C + 2 * (size_t)C * D] = part[0];
        if (llk_mean) *llk_mean = stats[(size_t)C + 2 * (size_t)C * D] / part[0];
    }
}

// w = N / sum N, mu = F / N, cov = S / N - mu^2 clamped to [floor, ceil] * ctl (inclusive), invcov = 1 / cov
// (Mixture._maximization + variance_control, sidekit/mixture.py:590-611, 489-506)
__global__ __launch_bounds__(256) void k_mstep_params(const double *__restrict__ stats, const double *__restrict__ ctl,
                                                       int C, int D, double ceil_cov, double floor_cov,
                                                       double *__restrict__ params)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)C * D) return;
    const int c = (int)(e / D);
    const double n = stats[c], sumN = params[(size_t)C + 2 * (size_t)C * D];
    if (e % D == 0) params[c] = n / sumN;
    const double m = stats[(size_t)C + e] / n;
    double cov = stats[(size_t)C + (size_t)C * D + e] / n - m * m;
    const double fl = floor_cov * ctl[e], ce = ceil_cov * ctl[e];
    if (cov <= fl) cov = fl;
    if (cov >= ce) cov = ce;
    params[(size_t)C + e] = m;
    params[(size_t)C + (size_t)C * D + e] = 1.0 / cov;
}

// the packed image of skm_gmm_pack, from parameters in device memory
__global__ __launch_bounds__(256) void k_pack_w(const double *__restrict__ params, int C, int D, int CTpad, int NCH, int NP,
                                                 float *__restrict__ wp, const double *__restrict__ riv)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;      // ((tile * NCH + grp) * 64 + lane) * 4 + i
    if (e >= (long)CTpad * NCH * 256) return;
    const int i = (int)(e & 3), lane = (int)((e >> 2) & 63);
    const long tg = e >> 8;
    const int grp = (int)(tg % NCH), tile = (int)(tg / NCH);
    const int comp = tile * 32 + (lane & 31), h = lane >> 5;
    int d, kind;
    if (grp < 2 * NP) {
        d = 8 * (grp >> 1) + 4 * h + i;
        kind = grp & 1;
    } else {
        d = 8 * NP + i;
        kind = h;
    }
    double v = 0.0;
    if (comp < C && d < D) {
        const double icv = params[(size_t)C + (size_t)C * D + (size_t)comp * D + d];
        v = kind ? (-0.5 * 1.4426950408889634074 * (icv - (riv ? riv[d] : 0.0)))
                 : (1.4426950408889634074 * params[(size_t)C + (size_t)comp * D + d] * icv);
    }
    wp[e] = (float)v;
}

__global__ __launch_bounds__(256) void k_pack_kc(const double *__restrict__ params, int C, int D, int ncomp,
                                                  float *__restrict__ kc, const double *__restrict__ riv)
{
    const int comp = blockIdx.x * blockDim.x + threadIdx.x;
    if (comp >= ncomp) return;
    double v = -1.0e30;
    if (comp < C) {
        const double *mu = params + C + (size_t)comp * D, *ic = params + C + (size_t)C * D + (size_t)comp * D;
        double logdet_inv = 0.0, quad = 0.0;
        for (int d = 0; d < D; ++d) {
            logdet_inv += (riv && riv[d] > 0.0) ? log(ic[d] / riv[d]) : log(ic[d]);
            quad += mu[d] * mu[d] * ic[d];
        }
        v = 1.4426950408889634074 *
            (log(params[comp]) + 0.5 * logdet_inv - (riv ? 0.0 : 0.5 * D * log(2.0 * M_PI)) - 0.5 * quad);
        if (!(v > -1.0e30)) v = -1.0e30;  // w = 0 or NaN: component never wins
    }
    kc[comp] = (float)v;
}

extern "C" int skm_gmm_mstep_pack(void *ctx_, const double *stats_dev, const double *cov_var_ctl_dev, int C, int D,
                                  double ceil_cov, double floor_cov, double *params_dev, void *packed_dev,
                                  double *llk_mean_dev, const double *ref_invvar_dev, void *stream)
{
    SkmCtx *ctx = static_cast<SkmCtx *>(ctx_);
    SKM_REQUIRE(ctx && stats_dev && cov_var_ctl_dev && params_dev && packed_dev, "skm_gmm_mstep_pack: NULL argument");
    SKM_REQUIRE(C > 0 && D > 0, "skm_gmm_mstep_pack: bad shape C=%d D=%d", C, D);
    hipStream_t st = (hipStream_t)stream;
    const GmmGeom g = skm_geom(C, D);
    float *wp = static_cast<float *>(packed_dev);
    float *kc = wp + (size_t)g.CTpad * g.NCH * 256;
    SkmProfScope ps(ctx, "k_mstep_pack", st);
    hipLaunchKernelGGL(k_mstep_sum, dim3(1), dim3(256), 0, st, stats_dev, C, D, params_dev, llk_mean_dev);
    hipLaunchKernelGGL(k_mstep_params, dim3((unsigned)(((long)C * D + 255) / 256)), dim3(256), 0, st, stats_dev,
                       cov_var_ctl_dev, C, D, ceil_cov, floor_cov, params_dev);
    hipLaunchKernelGGL(k_pack_w, dim3((unsigned)(((long)g.CTpad * g.NCH * 256 + 255) / 256)), dim3(256), 0, st, params_dev, C, D,
                       g.CTpad, g.NCH, g.NP, wp, ref_invvar_dev);
    hipLaunchKernelGGL(k_pack_kc, dim3((g.CTpad * 32 + 255) / 256), dim3(256), 0, st, params_dev, C, D, g.CTpad * 32, kc,
                       ref_invvar_dev);
    SKM_HIP(hipGetLastError());
    return SKM_OK;
}

// Launch plan shared by every entry point.
struct Plan {
    GmmGeom g;
    TileGeom tg;
    int nct1, nw1, n_cb1, n_fs1;
    size_t lds1;
    int nct2, nw2, n_cb2, n_fs2, nf2;
    size_t lds2;
    int n_llk;
    int team_nwv;       // waves per workgroup of the Baum-Welch segment kernel (2: models of one or two component tiles)
    // single-sweep kernel (7..16 component tiles, D <= 63)
    int fused, fused_nwv, fused_blocks, fused_fat, fat_blocks;
    // opt-in split-precision variant of the fat kernel (SKM_SPLIT=bf16x3; csrc/skm_gmm_b3.hip): 16 component tiles, 16-byte rows
    int b3;
    // ... and in two sweeps over blocks of 512 components above that (C = 1024, 2048, 4096, or ending inside the last block)
    int b3two, b3_ncb, b3_blocks;
    size_t off_wb;
    // wide rows (65 <= D+1 <= 160, C <= 128): the fat kernel with ONE component tile per wave, 2 or 4 waves
    int wide, wide_nwv, wide_blocks;
    size_t wide_lds;
    size_t fused_lds;
    // one-wave single-sweep kernel (1..4 component tiles): component tiles per wave, workgroups (= waves)
    int solo, solo_nct, solo_blocks;
    size_t solo_lds;
    // team kernel (C > 512 in multiples of 512): B workgroups per team, teams of one XCD share blockIdx % 8
    int team, team_B, team_per_xcd, team_wgs, team_nct;
    size_t team_lds, off_xch, off_flags, off_err;
    // workspace offsets (bytes)
    size_t off_pm, off_ps, off_lse, off_llk, off_slabs, total;
};

static size_t lds_bytes(int nct, int nw, const GmmGeom &g)
{
    return ((size_t)nct * g.NCH * 256 + (size_t)nw * (32 * g.XS + 32)) * sizeof(float);
}

// frame tiles a wave accumulates in f32 registers between two folds into its f64 slab
static int skm_env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

static int skm_flush_tiles()
{
    const char *e = getenv("SKM_FLUSH");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : SKM_FLUSH_TILES;
}

static int make_plan(const SkmCtx *ctx, long T, int C, int D, bool with_s, Plan *p)
{
    SKM_REQUIRE(T >= 0 && C > 0 && D > 0, "bad shape T=%ld C=%d D=%d", T, C, D);
    p->g = skm_geom(C, D);
    const GmmGeom &g = p->g;
    if (g.NJT > 5) {
        skm_set_error("feature dimension %d not supported (max 159)", D);
        return SKM_ERR_UNSUPPORTED;
    }
    p->tg.D = D;
    p->tg.XS = g.XS;
    p->tg.NG = g.NCH;
    p->tg.NP = g.NP;
    p->tg.tail = g.tail;
    p->tg.vec4 = (D % 4 == 0) ? 1 : 0;
    // sweep 1: largest component block / wave count that fits the LDS
    // (a model of one or two component tiles -- the first stages of EM_split -- takes a block of that size
    // instead of padding tiles)
    p->nct1 = 0;
    const int nct_max = g.CT >= 3 ? 4 : g.CT;
    for (int nw = 8; nw >= 1 && !p->nct1; nw >>= 1)
        for (int nct = nct_max; nct >= 1; nct >>= 1)
            if (lds_bytes(nct, nw, g) <= SKM_LDS_BUDGET) {
                p->nct1 = nct;
                p->nw1 = nw;
                break;
            }
    // sweep 2: component tiles per wave limited by the accumulator registers (NCT*NF*16 <= 128..160)
    p->nf2 = g.NJT * (with_s ? 2 : 1);
    p->nct2 = (p->nf2 <= 4) ? 2 : 1;
    if (with_s && g.CT == 1 && g.NJT <= 2) p->nct2 = 1;
    p->nw2 = 0;
    for (int nw = 8; nw >= 1; nw >>= 1)
        if (lds_bytes(p->nct2, nw, g) <= SKM_LDS_BUDGET) {
            p->nw2 = nw;
            break;
        }
    if (!p->nct1 || !p->nw2) {
        skm_set_error("no launch configuration fits the LDS for D=%d", D);
        return SKM_ERR_UNSUPPORTED;
    }
    p->lds1 = lds_bytes(p->nct1, p->nw1, g);
    p->lds2 = lds_bytes(p->nct2, p->nw2, g);
    p->n_cb1 = g.CTpad / p->nct1;
    // padding tiles only matter up to a multiple of the block; blocks that are all padding are skipped
    p->n_cb1 = (g.CT + p->nct1 - 1) / p->nct1;
    p->n_cb2 = (g.CT + p->nct2 - 1) / p->nct2;
    const int wgs = ctx->cu_count > 0 ? ctx->cu_count : 256;
    p->n_fs1 = wgs / p->n_cb1 > 0 ? wgs / p->n_cb1 : 1;
    p->n_fs2 = wgs / p->n_cb2 > 0 ? wgs / p->n_cb2 : 1;
    p->n_llk = 2048;
    // one wave per workgroup holds all components (C <= 128); staging needs 32*D <= 32 floats * 64 lanes
    {
        const char *e = getenv("SKM_SOLO");
        p->solo = (with_s && g.CT <= 4 && g.NJT <= 2 && 32 * g.D <= 2048 && (e ? atoi(e) != 0 : 1)) ? 1 : 0;
        p->solo_nct = g.CT >= 3 ? 4 : g.CT;
        p->solo_blocks = wgs * 4 * (p->solo_nct == 1 ? 2 : 1);
        if (p->solo_blocks > p->n_llk) p->solo_blocks = p->n_llk;
        p->solo_lds = ((size_t)2 * 32 * g.XS + 64) * sizeof(float);
    }
    // 5 or 6 component tiles run as 8 (two waves of four, padding tiles with kc = -1e30): C = 160 5.6 -> 4.2 ms per 4 M
    // frames against the two sweeps, C = 192 5.5 -> 4.1
    p->fused = (with_s && g.CT >= skm_env_int("SKM_FUSED_MIN_CT", 5) && g.CT <= 16 && g.NJT <= 2 && !getenv("SKM_NO_FUSED")) ? 1 : 0;
    p->fused_nwv = (g.CT + 1) / 2;
    p->fused_blocks = wgs * (p->fused_nwv <= 4 ? 2 : 1);
    if (p->fused_blocks > p->n_llk) p->fused_blocks = p->n_llk;
    // one 512-register wave per SIMD with 4 component tiles each; staging needs 32*D <= 8 floats * threads
    {
        const int fw = g.CTpad / 4;
        const char *e = getenv("SKM_FAT");
        // 3 waves would leave one SIMD of the CU idle (a 512-register wave owns its SIMD): 2 or 4 only
        p->fused_fat = (p->fused && (fw == 2 || fw == 4) && 32 * g.D <= 2048 && (e ? atoi(e) != 0 : 1)) ? 1 : 0;
        p->fat_blocks = wgs * (fw <= 2 ? 2 : 1);
        if (p->fat_blocks > p->n_llk) p->fat_blocks = p->n_llk;
    }
    {
        const char *e = getenv("SKM_SPLIT");
        const bool on = e && !strcmp(e, "bf16x3");
        p->b3 = (p->fused_fat && g.CTpad == 16 && g.NJT == 2 && D % 4 == 0 && D <= 60 && on) ? 1 : 0;
        const int ncb = g.CTpad / 16;
        p->b3_ncb = ncb;
        p->b3_blocks = ncb > 0 ? (wgs / (8 * ncb)) * 8 * ncb : 0;
        p->b3two = (with_s && on && g.CTpad % 16 == 0 && (ncb == 2 || ncb == 4 || ncb == 8) && p->b3_blocks > 0 &&
                    g.NJT == 2 && D % 4 == 0 && D <= 60) ? 1 : 0;
    }
    p->fused_lds = ((size_t)3 * 32 * g.XS + (size_t)4 * p->fused_nwv * 32 + (size_t)p->fused_nwv * 32 +
                    (size_t)p->fused_nwv * 2 * 16 * 64) * sizeof(float);
    // wide rows: 3..5 statistics column tiles leave room for ONE component tile per 512-register wave (2 * NJT
    // accumulator tiles = 96..160 AGPRs), so a workgroup of 2 or 4 waves holds up to 128 components; staging
    // needs 32*D <= 20 floats * 256 threads (40 * 128)
    {
        const char *e = getenv("SKM_WIDE");
        p->wide = (with_s && g.NJT >= 3 && g.NJT <= 5 && g.CT <= 4 && D % 4 == 0 && 32 * D <= 5120 &&
                   (e ? atoi(e) != 0 : 1)) ? 1 : 0;
        p->wide_nwv = g.CT <= 2 ? 2 : 4;
        p->wide_blocks = wgs * (p->wide_nwv == 2 ? 2 : 1);
        if (p->wide_blocks > p->n_llk) p->wide_blocks = p->n_llk;
        p->wide_lds = ((size_t)3 * 32 * g.XS + (size_t)4 * p->wide_nwv * 32 + (size_t)p->wide_nwv * 32) * sizeof(float);
    }
    // team kernel: EM statistics, C a multiple of 512 above 512, 16-byte rows with D <= 64, and a CU count
    // that splits into teams of B workgroups per XCD (blocks are dealt round-robin over 8 XCDs)
    {
        const int B = g.CTpad / 16;
        const char *e = getenv("SKM_TEAM");
        p->team_B = B;
        p->team_per_xcd = (B > 0) ? (wgs / 8) / B : 0;
        p->team_wgs = 8 * p->team_per_xcd * B;
        // (the last workgroup's last wave may hold padding tiles: kc = -1e30, gamma = 0, rows at or beyond C are skipped
        // by k_reduce_slabs and flush_session_agpr)
        p->team = (g.CTpad % 16 == 0 && (B == 1 || B == 2 || B == 4 || B == 8) && p->team_per_xcd >= 1 &&
                   D <= 64 && g.NJT <= 2 && (e ? atoi(e) != 0 : 1)) ? 1 : 0;     // D % 4 != 0: run-time-geometry kernels
        p->team_nct = 4;
        p->team_nwv = 4;
        // EM on wide rows above 128 components: ONE component tile per wave (2 * NJT accumulator tiles), teams of
        // C/128 workgroups; the last workgroup's waves may sit on padding tiles (k_reduce_slabs reads comp < C only)
        if (with_s && !p->team && !p->wide && g.NJT >= 3 && g.NJT <= 5 && D % 4 == 0 && 32 * D <= 5120) {
            const int Bw = g.CTpad / 4;
            const int per_xcd = Bw > 0 ? (wgs / 8) / Bw : 0;
            if ((Bw == 2 || Bw == 4 || Bw == 8 || Bw == 16) && per_xcd >= 1 && (e ? atoi(e) != 0 : 1) &&
                !getenv("SKM_NO_WIDE_TEAM")) {
                p->team = 1;
                p->team_nct = 1;
                p->team_B = Bw;
                p->team_per_xcd = per_xcd;
                p->team_wgs = 8 * per_xcd * Bw;
            }
        }
        // Baum-Welch (no S accumulators: NCT * NJT tiles) also runs the segment kernel on smaller component blocks:
        // ONE component tile per wave (C <= 128, a team of one workgroup) or two (blocks of 256 components, teams
        // of 1, 2, 4 or 8), and on wide rows (3..5 column tiles, 32 * D <= 20 floats * 256 threads).  The last
        // block may be padding (kc = -1e30): flush_session_agpr skips rows at or beyond C.
        if (!with_s && !p->team) {
            const char *es = getenv("SKM_BW_SMALL");
            const bool rows_ok = g.NJT <= 2 || (D % 4 == 0 && g.NJT <= 5 && 32 * D <= 5120);
            // measured against the two sweeps (4 M frames, sessions of 2000): D = 60: C = 64 2.5 vs 1.5 ms (two of four
            // waves idle), C = 96 2.5 vs 2.9, C = 160 3.9 vs 4.9, C = 224 3.9 vs 5.8; D = 132: C = 32 4.3 vs 3.6,
            // C = 64 4.3 vs 6.0, C = 512 15.5 vs 45.1
            // One or two component tiles (C <= 64, the reference's own UBM): workgroups of TWO waves, two per CU (round
            // 3) -- every SIMD holds a busy wave where the four-wave workgroup left two on padding tiles (0.28 of the
            // peak at C = 64, D = 132).  SKM_BW_PAIR=0 restores the old choice.
            const char *ep = getenv("SKM_BW_PAIR");
            // measured against the two sweeps (4 M frames, sessions of 2000, ms): C = 64, D = 132: 2.52 vs 6.08 (and 4.50
            // for the four-wave workgroup); C = 32, D = 132: 2.50 vs 3.65; C = 64, D = 60: 1.46 vs 1.54; but C = 32,
            // D = 60: 1.42 vs 1.28 and C = 40, D = 20: 0.94 vs 0.71 -- wide rows always, two tiles from 32 columns on
            const bool pair = g.CT <= 2 && (g.NJT >= 3 || (g.CT == 2 && g.NJT == 2)) && (ep ? atoi(ep) != 0 : 1);
            const int min_ct1 = g.NJT <= 2 ? 3 : 2;
            const int nct = pair ? 1 : (g.CT <= 4 ? (g.CT >= min_ct1 ? 1 : 0) : (g.CTpad % 8 == 0 ? 2 : 0));
            const int Bs = nct ? (pair ? 1 : g.CTpad / (4 * nct)) : 0;
            const int per_xcd = Bs > 0 ? ((pair ? 2 * wgs : wgs) / 8) / Bs : 0;
            if (rows_ok && nct && (Bs == 1 || ((Bs == 2 || Bs == 4 || Bs == 8) && g.NJT >= 3)) && per_xcd >= 1 &&
                (e ? atoi(e) != 0 : 1) && (es ? atoi(es) != 0 : 1)) {
                p->team = 1;
                p->team_nct = nct;
                p->team_B = Bs;
                p->team_per_xcd = per_xcd;
                p->team_wgs = 8 * per_xcd * Bs;
                p->team_nwv = pair ? 2 : 4;
            }
        }
        p->team_lds = p->team_nwv == 2
                          ? ((size_t)3 * 32 * g.XS + 2 * 32 + (size_t)2 * p->team_nct * 4 * 64 * 4) * sizeof(float)
                          : ((size_t)4 * 32 * g.XS + 4 * 32 + (size_t)4 * p->team_nct * 4 * 64 * 4) * sizeof(float);
    }
    size_t off = 0;
    p->off_pm = off;
    off += skm_align_up((size_t)p->n_cb1 * (size_t)T * sizeof(float), 256);
    p->off_ps = off;
    off += skm_align_up((size_t)p->n_cb1 * (size_t)T * sizeof(float), 256);
    p->off_lse = off;
    off += skm_align_up((size_t)T * sizeof(float), 256);
    p->off_llk = off;
    off += skm_align_up((size_t)p->n_llk * sizeof(double), 256);
    p->off_slabs = off;
    // slabs sized for the EM (with S) configuration, the largest user
    {
        const int nf = g.NJT * 2, nct = (nf <= 4) ? 2 : 1;
        int nw = 8;
        while (nw > 1 && lds_bytes(nct, nw, g) > SKM_LDS_BUDGET) nw >>= 1;
        const int ncb = (g.CT + nct - 1) / nct;
        const int nfs = wgs / ncb > 0 ? wgs / ncb : 1;
        size_t sl = (size_t)ncb * nfs * nw * nct * 32 * nf * 32 * sizeof(double);
        size_t slf = (size_t)p->fused_blocks * p->fused_nwv * 64 * nf * 32 * sizeof(double);
        const size_t slfat = (size_t)p->fat_blocks * (g.CTpad / 4) * 128 * nf * 32 * sizeof(double);
        if (p->fused_fat && slfat > slf) slf = slfat;
        if (p->fused && slf > sl) sl = slf;
        const size_t slsolo = (size_t)p->solo_blocks * p->solo_nct * 32 * nf * 32 * sizeof(double);
        if (p->solo && slsolo > sl) sl = slsolo;
        const size_t slteam = (size_t)p->team_wgs * 4 * p->team_nct * 32 * nf * 32 * sizeof(double);
        if (p->team && slteam > sl) sl = slteam;
        const size_t slwide = (size_t)p->wide_blocks * p->wide_nwv * 32 * nf * 32 * sizeof(double);
        if (p->wide && slwide > sl) sl = slwide;
        const size_t slb3 = (size_t)p->b3_blocks * 4 * 128 * nf * 32 * sizeof(double);
        if (p->b3two && slb3 > sl) sl = slb3;
        off += skm_align_up(sl, 256);
    }
    p->off_flags = off;     // flags and err are cleared by one memset: keep them adjacent
    off += skm_align_up((size_t)(p->team ? 8 * p->team_per_xcd * 4 * p->team_nwv * p->team_B : 0) * sizeof(unsigned), 256);
    p->off_err = off;
    off += 256;
    p->off_xch = off;
    off += skm_align_up((size_t)(p->team ? 8 * p->team_per_xcd * 4 * p->team_nwv * p->team_B * 64 : 0) * sizeof(float), 256);
    p->off_wb = off;
    off += skm_align_up((p->b3 || p->b3two) ? skm_estep_b3_wb_bytes(g.CTpad) : 0, 256);
    p->total = off;
    return SKM_OK;
}

extern "C" size_t skm_gmm_workspace_bytes(void *ctx, long T, int C, int D)
{
    if (!ctx) return 0;
    Plan p, q;      // one workspace serves the EM entry points and skm_bw_stats (its plan may use a team where EM's does not)
    if (make_plan(static_cast<SkmCtx *>(ctx), T, C, D, true, &p) != SKM_OK) return 0;
    if (make_plan(static_cast<SkmCtx *>(ctx), T, C, D, false, &q) != SKM_OK) return 0;
    return p.total > q.total ? p.total : q.total;
}

template <typename K>
static int set_lds(K kernel, size_t bytes)
{
    SKM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)bytes));
    return SKM_OK;
}

static int launch_lse(SkmCtx *ctx, const Plan &p, const float *X, long T, const float *wp, const float *kc,
                      char *ws, float *loglik, hipStream_t st, double *lp = nullptr, int C = 0)
{
    LseArgs a;
    a.lp = lp;
    a.C = C;
#ifdef SKM_TRACE
    a.trace = g_trace;
#endif
    a.X = X;
    a.T = T;
    a.g = p.tg;
    a.wp = wp;
    a.kc = kc;
    a.pm = reinterpret_cast<float *>(ws + p.off_pm);
    a.ps = reinterpret_cast<float *>(ws + p.off_ps);
    a.n_cb = p.n_cb1;
    a.n_fs = p.n_fs1;
    const dim3 grid(p.n_cb1 * p.n_fs1), block(64 * p.nw1);
    int rc = SKM_OK;
    {
        SkmProfScope ps(ctx, "k_lse_partial", st);
        switch (p.nct1) {
        case 4:
            if ((rc = set_lds(k_lse_partial<4>, p.lds1))) return rc;
            hipLaunchKernelGGL(k_lse_partial<4>, grid, block, p.lds1, st, a);
            break;
        case 2:
            if ((rc = set_lds(k_lse_partial<2>, p.lds1))) return rc;
            hipLaunchKernelGGL(k_lse_partial<2>, grid, block, p.lds1, st, a);
            break;
        default:
            if ((rc = set_lds(k_lse_partial<1>, p.lds1))) return rc;
            hipLaunchKernelGGL(k_lse_partial<1>, grid, block, p.lds1, st, a);
            break;
        }
    }
    SKM_HIP(hipGetLastError());
    long nb = (T + 255) / 256;
    if (nb > p.n_llk) nb = p.n_llk;
    if (nb < 1) nb = 1;
    // unused partial slots must read as zero
    SKM_HIP(hipMemsetAsync(ws + p.off_llk, 0, p.n_llk * sizeof(double), st));
    {
        SkmProfScope ps(ctx, "k_lse_combine", st);
        hipLaunchKernelGGL(k_lse_combine, dim3((unsigned)nb), dim3(256), 0, st, a.pm, a.ps, p.n_cb1, T,
                           reinterpret_cast<float *>(ws + p.off_lse), loglik,
                           reinterpret_cast<double *>(ws + p.off_llk));
    }
    SKM_HIP(hipGetLastError());
    return SKM_OK;
}

template <int NJT, bool WITH_S, bool BW, int NCT_ = 0>
static int launch_stats_t(SkmCtx *ctx, const Plan &p, const StatsArgs &a, hipStream_t st)
{
    constexpr int NF = NJT * (WITH_S ? 2 : 1);
    constexpr int NCT = NCT_ ? NCT_ : ((NF <= 4) ? 2 : 1);
    if (p.nct2 != NCT) {
        skm_set_error("statistics sweep: plan wants %d component tiles per wave, kernel has %d", p.nct2, NCT);
        return SKM_ERR_UNSUPPORTED;
    }
    int rc = set_lds(k_stats<NJT, NCT, WITH_S, BW>, p.lds2);
    if (rc) return rc;
    SkmProfScope ps(ctx, BW ? "k_stats_bw" : "k_stats_em", st);
    hipLaunchKernelGGL((k_stats<NJT, NCT, WITH_S, BW>), dim3(p.n_cb2 * p.n_fs2), dim3(64 * p.nw2), p.lds2, st, a);
    SKM_HIP(hipGetLastError());
    return SKM_OK;
}

template <bool WITH_S, bool BW>
static int launch_stats(SkmCtx *ctx, const Plan &p, const StatsArgs &a, hipStream_t st)
{
    // a single component tile (C <= 32, the first stages of EM_split): one tile per wave instead of a padding tile
    if (WITH_S && !BW && p.nct2 == 1 && p.g.NJT == 1) return launch_stats_t<1, WITH_S, BW, 1>(ctx, p, a, st);
    if (WITH_S && !BW && p.nct2 == 1 && p.g.NJT == 2) return launch_stats_t<2, WITH_S, BW, 1>(ctx, p, a, st);
    switch (p.g.NJT) {
    case 1: return launch_stats_t<1, WITH_S, BW>(ctx, p, a, st);
    case 2: return launch_stats_t<2, WITH_S, BW>(ctx, p, a, st);
    case 3: return launch_stats_t<3, WITH_S, BW>(ctx, p, a, st);
    case 4: return launch_stats_t<4, WITH_S, BW>(ctx, p, a, st);
    case 5: return launch_stats_t<5, WITH_S, BW>(ctx, p, a, st);
    }
    skm_set_error("unsupported NJT=%d", p.g.NJT);
    return SKM_ERR_UNSUPPORTED;
}

static int check_common(void *ctx, const float *X, long T, int C, int D, const void *packed, void *ws,
                        size_t ws_bytes, const Plan &p, const char *who)
{
    SKM_REQUIRE(ctx && packed, "%s: NULL ctx/packed", who);
    SKM_REQUIRE(T == 0 || X != nullptr, "%s: X is NULL", who);
    SKM_REQUIRE(((uintptr_t)X & 15) == 0, "%s: X must be 16-byte aligned", who);
    SKM_REQUIRE(((uintptr_t)packed & 15) == 0, "%s: packed image must be 16-byte aligned", who);
    SKM_REQUIRE(ws != nullptr && ((uintptr_t)ws & 255) == 0, "%s: workspace NULL or not 256-byte aligned", who);
    if (ws_bytes < p.total) {
        skm_set_error("%s: workspace too small (%zu < %zu)", who, ws_bytes, p.total);
        return SKM_ERR_WORKSPACE;
    }
    (void)C;
    (void)D;
    return SKM_OK;
}

// E-step with a given plan.  single_sweep_only: the caller's workspace has no room for the T-sized regions of the
// two-sweep path (plan made with T = 0), so a team launch that cannot be co-resident is an error, not a fallback.
static int estep_run(SkmCtx *ctx, const Plan &p, const float *X, long T, int C, int D, const void *packed, double *out,
                     int accumulate, char *ws, bool single_sweep_only, hipStream_t st)
{
    int rc;
    const float *wp = static_cast<const float *>(packed);
    const float *kc = wp + (size_t)p.g.CTpad * p.g.NCH * 256;
    if (p.solo) {
        // single sweep, all components in one wave (C <= 128)
        FusedArgs f;
        memset(&f, 0, sizeof(f));
        f.X = X;
        f.T = T;
        f.g = p.tg;
        f.wp = wp;
        f.kc = kc;
        f.slabs = reinterpret_cast<double *>(ws + p.off_slabs);
        f.llk_part = reinterpret_cast<double *>(ws + p.off_llk);
        f.flush_tiles = skm_flush_tiles();
        const int nblk = p.solo_blocks, nf = 2 * p.g.NJT;
        SKM_HIP(hipMemsetAsync(f.llk_part, 0, p.n_llk * sizeof(double), st));
        {
            SkmProfScope ps(ctx, "k_estep_solo", st);
            const int geo = ((p.tg.NP & 1) ? 2 : 0) | (p.tg.tail ? 1 : 0);
#define SKM_SOLO_LAUNCH(NCT_, NJT_, GEO_)                                                                      \
    do {                                                                                                       \
        if ((rc = set_lds(k_estep_solo<NCT_, NJT_, GEO_>, p.solo_lds))) return rc;                             \
        hipLaunchKernelGGL((k_estep_solo<NCT_, NJT_, GEO_>), dim3(nblk), dim3(64), p.solo_lds, st, f);         \
    } while (0)
#define SKM_SOLO_GEO(NCT_)                                                                                     \
    do {                                                                                                       \
        /* the geometry is a template argument for every D: the run-time form of the logits ending (branches that  \
           merge behind asm MFMAs) gave wrong logits in this kernel for an even pair count with a tail group */    \
        if (p.g.NJT == 1) {                                                                                    \
            if (geo == 0) SKM_SOLO_LAUNCH(NCT_, 1, 0);                                                         \
            else if (geo == 1) SKM_SOLO_LAUNCH(NCT_, 1, 1);                                                    \
            else if (geo == 2) SKM_SOLO_LAUNCH(NCT_, 1, 2);                                                    \
            else SKM_SOLO_LAUNCH(NCT_, 1, 3);                                                                  \
        } else if (geo == 0) SKM_SOLO_LAUNCH(NCT_, 2, 0);                                                      \
        else if (geo == 1) SKM_SOLO_LAUNCH(NCT_, 2, 1);                                                        \
        else if (geo == 2) SKM_SOLO_LAUNCH(NCT_, 2, 2);                                                        \
        else SKM_SOLO_LAUNCH(NCT_, 2, 3);                                                                      \
    } while (0)
            if (p.solo_nct == 1) SKM_SOLO_GEO(1);
            else if (p.solo_nct == 2) SKM_SOLO_GEO(2);
            else SKM_SOLO_GEO(4);
#undef SKM_SOLO_GEO
#undef SKM_SOLO_LAUNCH
        }
        SKM_HIP(hipGetLastError());
        const long n_out = (long)C * nf * 32;
        SkmProfScope ps(ctx, "k_reduce_slabs", st);
        hipLaunchKernelGGL(k_reduce_slabs, dim3((unsigned)((n_out + SKM_RED_OUT - 1) / SKM_RED_OUT)), dim3(256), 0, st, f.slabs, 1, nblk,
                           1, p.solo_nct, p.g.NJT, nf, C, D, out, accumulate, f.llk_part, p.n_llk);
        SKM_HIP(hipGetLastError());
        return SKM_OK;
    }
    if (p.wide) {
        // single sweep for wide rows (D = 64..159, C <= 128): the fat kernel with one component tile per wave
        FusedArgs f;
        memset(&f, 0, sizeof(f));
        f.X = X;
        f.T = T;
        f.g = p.tg;
        f.wp = wp;
        f.kc = kc;
        f.slabs = reinterpret_cast<double *>(ws + p.off_slabs);
        f.llk_part = reinterpret_cast<double *>(ws + p.off_llk);
        f.stagger = 1;
        f.flush_tiles = skm_flush_tiles();
        const int nwv = p.wide_nwv, nblk = p.wide_blocks, nf = 2 * p.g.NJT;
        SKM_HIP(hipMemsetAsync(f.llk_part, 0, p.n_llk * sizeof(double), st));
        {
            SkmProfScope ps(ctx, "k_estep_fat", st);
            const int geo = ((p.tg.NP & 1) ? 2 : 0) | (p.tg.tail ? 1 : 0);
#define SKM_WIDE_LAUNCH(NJT_, GEO_)                                                                             \
    do {                                                                                                       \
        if (nwv == 4) {                                                                                        \
            if ((rc = set_lds(k_estep_fat<1, NJT_, GEO_, 20>, p.wide_lds))) return rc;                         \
            hipLaunchKernelGGL((k_estep_fat<1, NJT_, GEO_, 20>), dim3(nblk), dim3(256), p.wide_lds, st, f);    \
        } else {                                                                                               \
            if ((rc = set_lds(k_estep_fat<1, NJT_, GEO_, 40>, p.wide_lds))) return rc;                         \
            hipLaunchKernelGGL((k_estep_fat<1, NJT_, GEO_, 40>), dim3(nblk), dim3(128), p.wide_lds, st, f);    \
        }                                                                                                      \
    } while (0)
#define SKM_WIDE_GEO(NJT_)                                                                                      \
    do {                                                                                                       \
        if (geo == 0) SKM_WIDE_LAUNCH(NJT_, 0);                                                                \
        else if (geo == 1) SKM_WIDE_LAUNCH(NJT_, 1);                                                           \
        else if (geo == 2) SKM_WIDE_LAUNCH(NJT_, 2);                                                           \
        else SKM_WIDE_LAUNCH(NJT_, 3);                                                                         \
    } while (0)
            if (p.g.NJT == 3) SKM_WIDE_GEO(3);
            else if (p.g.NJT == 4) SKM_WIDE_GEO(4);
            else SKM_WIDE_GEO(5);
#undef SKM_WIDE_GEO
#undef SKM_WIDE_LAUNCH
        }
        SKM_HIP(hipGetLastError());
        const long n_out = (long)C * nf * 32;
        SkmProfScope ps(ctx, "k_reduce_slabs", st);
        hipLaunchKernelGGL(k_reduce_slabs, dim3((unsigned)((n_out + SKM_RED_OUT - 1) / SKM_RED_OUT)), dim3(256), 0, st, f.slabs, nwv, nblk,
                           1, 1, p.g.NJT, nf, C, D, out, accumulate, f.llk_part, p.n_llk);
        SKM_HIP(hipGetLastError());
        return SKM_OK;
    }
    if (p.fused) {
        // single sweep: all components in one workgroup (C <= 512)
        FusedArgs f;
        f.X = X;
        f.T = T;
        f.g = p.tg;
        f.wp = wp;
        f.kc = kc;
        f.slabs = reinterpret_cast<double *>(ws + p.off_slabs);
        f.llk_part = reinterpret_cast<double *>(ws + p.off_llk);
#ifdef SKM_TRACE
        f.trace = g_trace;
#endif
        {
            const char *e = getenv("SKM_STAGGER");
            f.stagger = e ? atoi(e) : 1;
        }
        f.flush_tiles = skm_flush_tiles();
        const bool fat = p.fused_fat;
        const int nwv = fat ? p.g.CTpad / 4 : p.fused_nwv, nblk = fat ? p.fat_blocks : p.fused_blocks, nf = 2 * p.g.NJT;
        // the fat kernel's first flush writes its slab; the older fused kernel adds to a cleared one
        if (!fat) SKM_HIP(hipMemsetAsync(f.slabs, 0, (size_t)nblk * nwv * 64 * nf * 32 * sizeof(double), st));
        SKM_HIP(hipMemsetAsync(f.llk_part, 0, p.n_llk * sizeof(double), st));
        if (p.b3) {
            if ((rc = skm_estep_b3_launch(ctx, X, T, D, p.g.CTpad, p.g.NCH, p.g.NP, wp, kc, ws + p.off_wb, f.slabs,
                                          f.llk_part, p.n_llk, nblk, f.flush_tiles, 1, nullptr, nullptr, nullptr, st)))
                return rc;
        } else {
            SkmProfScope ps(ctx, fat ? "k_estep_fat" : "k_estep_fused", st);
            if (fat) {
                const int geo = ((p.tg.NP & 1) ? 2 : 0) | (p.tg.tail ? 1 : 0);
#define SKM_FAT_LAUNCH(NJT_, GEO_)                                                                              \
    do {                                                                                                       \
        if (nwv == 4) {                                                                                        \
            if ((rc = set_lds(k_estep_fat<4, NJT_, GEO_, 8>, p.fused_lds))) return rc;                         \
            hipLaunchKernelGGL((k_estep_fat<4, NJT_, GEO_, 8>), dim3(nblk), dim3(256), p.fused_lds, st, f);    \
        } else {                                                                                               \
            if ((rc = set_lds(k_estep_fat<4, NJT_, GEO_, 16>, p.fused_lds))) return rc;                        \
            hipLaunchKernelGGL((k_estep_fat<4, NJT_, GEO_, 16>), dim3(nblk), dim3(128), p.fused_lds, st, f);   \
        }                                                                                                      \
    } while (0)
                if (p.g.NJT == 1) {
                    if (geo == 0) SKM_FAT_LAUNCH(1, 0);
                    else if (geo == 1) SKM_FAT_LAUNCH(1, 1);
                    else if (geo == 2) SKM_FAT_LAUNCH(1, 2);
                    else SKM_FAT_LAUNCH(1, 3);
                } else {
                    if (geo == 0) SKM_FAT_LAUNCH(2, 0);
                    else if (geo == 1) SKM_FAT_LAUNCH(2, 1);
                    else if (geo == 2) SKM_FAT_LAUNCH(2, 2);
                    else SKM_FAT_LAUNCH(2, 3);
                }
#undef SKM_FAT_LAUNCH
            } else if (p.g.NJT == 1) {
                if ((rc = set_lds(k_estep_fused<1>, p.fused_lds))) return rc;
                hipLaunchKernelGGL(k_estep_fused<1>, dim3(nblk), dim3(64 * nwv), p.fused_lds, st, f);
            } else {
                if ((rc = set_lds(k_estep_fused<2>, p.fused_lds))) return rc;
                hipLaunchKernelGGL(k_estep_fused<2>, dim3(nblk), dim3(64 * nwv), p.fused_lds, st, f);
            }
        }
        SKM_HIP(hipGetLastError());
        const long n_out = (long)C * nf * 32;
        SkmProfScope ps(ctx, "k_reduce_slabs", st);
        hipLaunchKernelGGL(k_reduce_slabs, dim3((unsigned)((n_out + SKM_RED_OUT - 1) / SKM_RED_OUT)), dim3(256), 0, st, f.slabs, nwv, nblk,
                           1, fat ? 4 : 2, p.g.NJT, nf, C, D, out, accumulate, f.llk_part, p.n_llk);
        SKM_HIP(hipGetLastError());
        return SKM_OK;
    }
    if (p.b3two && !single_sweep_only) {
        // split precision above 512 components: two sweeps of k_estep_b3 over blocks of 512 (csrc/skm_gmm_b3.hip)
        double *slabs = reinterpret_cast<double *>(ws + p.off_slabs);
        double *llk_part = reinterpret_cast<double *>(ws + p.off_llk);
        SKM_HIP(hipMemsetAsync(llk_part, 0, p.n_llk * sizeof(double), st));
        if ((rc = skm_estep_b3_launch(ctx, X, T, D, p.g.CTpad, p.g.NCH, p.g.NP, wp, kc, ws + p.off_wb, slabs, llk_part,
                                      p.n_llk, p.b3_blocks, skm_flush_tiles(), p.b3_ncb,
                                      reinterpret_cast<float *>(ws + p.off_pm), reinterpret_cast<float *>(ws + p.off_ps),
                                      reinterpret_cast<float *>(ws + p.off_lse), st)))
            return rc;
        const int nf = 2 * p.g.NJT;
        const long n_out = (long)C * nf * 32;
        SkmProfScope ps(ctx, "k_reduce_slabs", st);
        hipLaunchKernelGGL(k_reduce_slabs, dim3((unsigned)((n_out + SKM_RED_OUT - 1) / SKM_RED_OUT)), dim3(256), 0, st, slabs,
                           4 * p.b3_ncb, p.b3_blocks / p.b3_ncb, 1, 4, p.g.NJT, nf, C, D, out, accumulate, llk_part, p.n_llk);
        SKM_HIP(hipGetLastError());
        return SKM_OK;
    }
    if (p.team && p.team_B >= 2) {
        TeamArgs t;
        t.X = X;
        t.T = T;
        t.g = p.tg;
        t.wp = wp;
        t.kc = kc;
        t.slabs = reinterpret_cast<double *>(ws + p.off_slabs);
        t.llk_part = reinterpret_cast<double *>(ws + p.off_llk);
        t.xch = reinterpret_cast<float *>(ws + p.off_xch);
        t.flags = reinterpret_cast<unsigned *>(ws + p.off_flags);
        t.err = reinterpret_cast<int *>(ws + p.off_err);
        t.B = p.team_B;
        t.teams_per_xcd = p.team_per_xcd;
        t.w_bytes = p.g.CTpad * p.g.NCH * 1024;
        t.seg_off = nullptr;
        t.S = 0;
        t.C = C;
        t.stat0 = t.stat1 = nullptr;
        t.debug_drop = 0;
#ifdef SKM_TRACE
        t.trace = g_trace;
#endif
        const int nf = 2 * p.g.NJT;
        SKM_HIP(hipMemsetAsync(t.slabs, 0, (size_t)p.team_wgs * 4 * p.team_nct * 32 * nf * 32 * sizeof(double), st));
        SKM_HIP(hipMemsetAsync(t.llk_part, 0, p.n_llk * sizeof(double), st));
        SKM_HIP(hipMemsetAsync(t.flags, 0, p.off_xch - p.off_flags, st));
        void *kargs[] = {&t};
        hipError_t le;
        {
            SkmProfScope ps(ctx, "k_estep_team", st);
#define SKM_TEAM_LAUNCH(NJT_, B_, GEO_)                                                                            \
    do {                                                                                                          \
        if ((rc = set_lds(k_estep_team<4, NJT_, B_, GEO_>, p.team_lds))) return rc;                               \
        le = hipLaunchCooperativeKernel(reinterpret_cast<const void *>(k_estep_team<4, NJT_, B_, GEO_>),          \
                                        dim3(p.team_wgs), dim3(256), kargs, (unsigned)p.team_lds, st);            \
    } while (0)
#define SKM_TEAM_GEO(B_)                                                                                           \
    do {                                                                                                          \
        if (geo == 0) SKM_TEAM_LAUNCH(2, B_, 0);                                                                  \
        else if (geo == 1) SKM_TEAM_LAUNCH(2, B_, 1);                                                             \
        else if (geo == 2) SKM_TEAM_LAUNCH(2, B_, 2);                                                             \
        else SKM_TEAM_LAUNCH(2, B_, 3);                                                                           \
    } while (0)
            le = hipErrorInvalidValue;
            const int geo = ((p.tg.NP & 1) ? 2 : 0) | (p.tg.tail ? 1 : 0);
#define SKM_TEAMW_LAUNCH(NJT_, B_)                                                                                 \
    do {                                                                                                          \
        if ((rc = set_lds(k_estep_team<1, NJT_, B_, -1>, p.team_lds))) return rc;                                 \
        le = hipLaunchCooperativeKernel(reinterpret_cast<const void *>(k_estep_team<1, NJT_, B_, -1>),            \
                                        dim3(p.team_wgs), dim3(256), kargs, (unsigned)p.team_lds, st);            \
    } while (0)
#define SKM_TEAMW_B(NJT_)                                                                                          \
    do {                                                                                                          \
        if (p.team_B == 2) SKM_TEAMW_LAUNCH(NJT_, 2);                                                             \
        else if (p.team_B == 4) SKM_TEAMW_LAUNCH(NJT_, 4);                                                        \
        else if (p.team_B == 8) SKM_TEAMW_LAUNCH(NJT_, 8);                                                        \
        else SKM_TEAMW_LAUNCH(NJT_, 16);                                                                          \
    } while (0)
            if (p.team_nct == 1) {  // wide rows: one component tile per wave, run-time geometry
                if (p.g.NJT == 3) SKM_TEAMW_B(3);
                else if (p.g.NJT == 4) SKM_TEAMW_B(4);
                else SKM_TEAMW_B(5);
            } else
            if (p.g.NJT == 1) {     // D < 32: geometry stays a run-time branch (fewer instantiations)
                if (p.team_B == 2) SKM_TEAM_LAUNCH(1, 2, -1);
                else if (p.team_B == 4) SKM_TEAM_LAUNCH(1, 4, -1);
                else if (p.team_B == 8) SKM_TEAM_LAUNCH(1, 8, -1);
            } else if (!p.tg.vec4) {    // rows that are not a multiple of 16 bytes: the staging path for any D
#define SKM_TEAMG_LAUNCH(B_, GEO_)                                                                                 \
    do {                                                                                                          \
        if ((rc = set_lds(k_estep_team<4, 2, B_, GEO_, true>, p.team_lds))) return rc;                            \
        le = hipLaunchCooperativeKernel(reinterpret_cast<const void *>(k_estep_team<4, 2, B_, GEO_, true>),       \
                                        dim3(p.team_wgs), dim3(256), kargs, (unsigned)p.team_lds, st);            \
    } while (0)
#define SKM_TEAMG_GEO(B_)                                                                                          \
    do {                                                                                                          \
        if (geo == 0) SKM_TEAMG_LAUNCH(B_, 0);                                                                    \
        else if (geo == 1) SKM_TEAMG_LAUNCH(B_, 1);                                                               \
        else if (geo == 2) SKM_TEAMG_LAUNCH(B_, 2);                                                               \
        else SKM_TEAMG_LAUNCH(B_, 3);                                                                             \
    } while (0)
                if (p.team_B == 2) SKM_TEAMG_GEO(2);
                else if (p.team_B == 4) SKM_TEAMG_GEO(4);
                else if (p.team_B == 8) SKM_TEAMG_GEO(8);
#undef SKM_TEAMG_GEO
#undef SKM_TEAMG_LAUNCH
            } else {
                if (p.team_B == 2) SKM_TEAM_GEO(2);
                else if (p.team_B == 4) SKM_TEAM_GEO(4);
                else if (p.team_B == 8) SKM_TEAM_GEO(8);
            }
#undef SKM_TEAM_GEO
#undef SKM_TEAM_LAUNCH
#undef SKM_TEAMW_B
#undef SKM_TEAMW_LAUNCH
        }
        if (le == hipSuccess) {
            const long n_out = (long)C * nf * 32;
            SkmProfScope ps(ctx, "k_reduce_slabs", st);
            hipLaunchKernelGGL(k_reduce_slabs, dim3((unsigned)((n_out + SKM_RED_OUT - 1) / SKM_RED_OUT)), dim3(256), 0, st, t.slabs,
                               4 * p.team_B, 8 * p.team_per_xcd, 1, p.team_nct, p.g.NJT, nf, C, D, out, accumulate, t.llk_part,
                               p.n_llk, t.err);
            SKM_HIP(hipGetLastError());
            return SKM_OK;
        }
        (void)hipGetLastError();   // the teams cannot all be resident on this device: two sweeps instead
        if (single_sweep_only) {
            skm_set_error("E-step: the cooperative team launch failed (%s) and the scratch has no room for two sweeps",
                          hipGetErrorString(le));
            return SKM_ERR_HIP;
        }
    }
    if ((rc = launch_lse(ctx, p, X, T, wp, kc, ws, nullptr, st))) return rc;
    StatsArgs a;
    memset(&a, 0, sizeof(a));
    a.X = X;
    a.T = T;
    a.g = p.tg;
    a.wp = wp;
    a.kc = kc;
    a.lse2 = reinterpret_cast<const float *>(ws + p.off_lse);
    a.n_cb = p.n_cb2;
    a.n_fs = p.n_fs2;
    a.C = C;
    a.slabs = reinterpret_cast<double *>(ws + p.off_slabs);
    SKM_HIP(hipMemsetAsync(a.slabs, 0,
                           (size_t)p.n_cb2 * p.n_fs2 * p.nw2 * p.nct2 * 32 * p.nf2 * 32 * sizeof(double), st));
    if ((rc = launch_stats<true, false>(ctx, p, a, st))) return rc;
    const long n_out = (long)C * p.nf2 * 32;
    {
        SkmProfScope ps(ctx, "k_reduce_slabs", st);
        hipLaunchKernelGGL(k_reduce_slabs, dim3((unsigned)((n_out + SKM_RED_OUT - 1) / SKM_RED_OUT)), dim3(256), 0, st, a.slabs,
                           p.n_cb2, p.n_fs2, p.nw2, p.nct2, p.g.NJT, p.nf2, C, D, out, accumulate,
                           reinterpret_cast<const double *>(ws + p.off_llk), p.n_llk);
    }
    SKM_HIP(hipGetLastError());
    return SKM_OK;
}

extern "C" int skm_gmm_estep(void *ctx_, const float *X, long T, int C, int D, const void *packed, double *out,
                             int accumulate, void *ws_, size_t ws_bytes, void *stream)
{
    SkmCtx *ctx = static_cast<SkmCtx *>(ctx_);
    SKM_REQUIRE(ctx != nullptr, "skm_gmm_estep: NULL ctx");
    Plan p;
    int rc = make_plan(ctx, T, C, D, true, &p);
    if (rc) return rc;
    if ((rc = check_common(ctx_, X, T, C, D, packed, ws_, ws_bytes, p, "skm_gmm_estep"))) return rc;
    SKM_REQUIRE(out != nullptr, "skm_gmm_estep: out is NULL");
    return estep_run(ctx, p, X, T, C, D, packed, out, accumulate, static_cast<char *>(ws_), false, (hipStream_t)stream);
}

// skm_gmm_estep without a workspace argument and without any allocation: the kernels take their scratch (slabs,
// partials, hand-off rows) from this GPU's copy of the file-scope `scratch` array through SkmCtx::scratch.
// Shapes served by a single-sweep kernel (C in {256, 512} fused, C in {1024, 2048, 4096} team; D % 4 == 0, D <= 64)
// need a scratch that does not grow with T; any other shape takes two sweeps whose per-frame partials must fit too,
// which bounds T.  Everything that must start at zero is cleared on the stream at every call (the array keeps the
// previous call's contents); calls that overlap on one GPU share the array and race.
extern "C" size_t skm_gmm_estep_noalloc_max_bytes(void) { return SKM_SCRATCH_BYTES; }

extern "C" int skm_gmm_estep_noalloc(void *ctx_, const float *X, long T, int C, int D, const void *packed, double *out,
                                     int accumulate, void *stream)
{
    SkmCtx *ctx = static_cast<SkmCtx *>(ctx_);
    SKM_REQUIRE(ctx != nullptr, "skm_gmm_estep_noalloc: NULL ctx");
    SKM_REQUIRE(ctx->scratch != nullptr, "skm_gmm_estep_noalloc: no scratch address for device %d", ctx->device);
    Plan p;
    int rc = make_plan(ctx, 0, C, D, true, &p);          // layout without the T-sized two-sweep regions
    if (rc) return rc;
    const bool single = p.solo || p.wide || p.fused || (p.team && p.team_B >= 2);
    if (!single && (rc = make_plan(ctx, T, C, D, true, &p))) return rc;
    if (p.total > SKM_SCRATCH_BYTES) {
        skm_set_error("skm_gmm_estep_noalloc: T=%ld C=%d D=%d needs %zu bytes of scratch, the built-in array holds %zu "
                      "(use skm_gmm_estep with a workspace)", T, C, D, p.total, (size_t)SKM_SCRATCH_BYTES);
        return SKM_ERR_WORKSPACE;
    }
    if ((rc = check_common(ctx_, X, T, C, D, packed, ctx->scratch, SKM_SCRATCH_BYTES, p, "skm_gmm_estep_noalloc"))) return rc;
    SKM_REQUIRE(out != nullptr, "skm_gmm_estep_noalloc: out is NULL");
    return estep_run(ctx, p, X, T, C, D, packed, out, accumulate, static_cast<char *>(ctx->scratch), single,
                     (hipStream_t)stream);
}

extern "C" int skm_gmm_loglik(void *ctx_, const float *X, long T, int C, int D, const void *packed, float *loglik,
                              void *ws_, size_t ws_bytes, void *stream)
{
    SkmCtx *ctx = static_cast<SkmCtx *>(ctx_);
    SKM_REQUIRE(ctx != nullptr, "skm_gmm_loglik: NULL ctx");
    Plan p;
    int rc = make_plan(ctx, T, C, D, true, &p);
    if (rc) return rc;
    if ((rc = check_common(ctx_, X, T, C, D, packed, ws_, ws_bytes, p, "skm_gmm_loglik"))) return rc;
    SKM_REQUIRE(loglik != nullptr, "skm_gmm_loglik: loglik is NULL");
    const float *wp = static_cast<const float *>(packed);
    const float *kc = wp + (size_t)p.g.CTpad * p.g.NCH * 256;
    return launch_lse(ctx, p, X, T, wp, kc, static_cast<char *>(ws_), loglik, (hipStream_t)stream);
}

extern "C" int skm_gmm_logprob(void *ctx_, const float *X, long T, int C, int D, const void *packed, double *lp,
                               float *loglik, void *ws_, size_t ws_bytes, void *stream)
{
    SkmCtx *ctx = static_cast<SkmCtx *>(ctx_);
    SKM_REQUIRE(ctx != nullptr, "skm_gmm_logprob: NULL ctx");
    Plan p;
    int rc = make_plan(ctx, T, C, D, true, &p);
    if (rc) return rc;
    if ((rc = check_common(ctx_, X, T, C, D, packed, ws_, ws_bytes, p, "skm_gmm_logprob"))) return rc;
    SKM_REQUIRE(lp != nullptr, "skm_gmm_logprob: lp is NULL");
    const float *wp = static_cast<const float *>(packed);
    const float *kc = wp + (size_t)p.g.CTpad * p.g.NCH * 256;
    return launch_lse(ctx, p, X, T, wp, kc, static_cast<char *>(ws_), loglik, (hipStream_t)stream, lp, C);
}

extern "C" int skm_bw_stats(void *ctx_, const float *X, long T, const int64_t *seg_off, int S, int C, int D,
                            const void *packed, double *stat0, double *stat1, void *ws_, size_t ws_bytes,
                            void *stream)
{
    SkmCtx *ctx = static_cast<SkmCtx *>(ctx_);
    SKM_REQUIRE(ctx != nullptr, "skm_bw_stats: NULL ctx");
    Plan p;
    int rc = make_plan(ctx, T, C, D, false, &p);
    if (rc) return rc;
    if ((rc = check_common(ctx_, X, T, C, D, packed, ws_, ws_bytes, p, "skm_bw_stats"))) return rc;
    SKM_REQUIRE(S >= 0 && (S == 0 || (seg_off && stat0 && stat1)), "skm_bw_stats: NULL output or offsets");
    if (S == 0) return SKM_OK;
    hipStream_t st = (hipStream_t)stream;
    char *ws = static_cast<char *>(ws_);
    const float *wp = static_cast<const float *>(packed);
    const float *kc = wp + (size_t)p.g.CTpad * p.g.NCH * 256;
    if (p.team) {
        TeamArgs t;
        memset(&t, 0, sizeof(t));
        t.X = X;
        t.T = T;
        t.g = p.tg;
        t.wp = wp;
        t.kc = kc;
        t.xch = reinterpret_cast<float *>(ws + p.off_xch);
        t.flags = reinterpret_cast<unsigned *>(ws + p.off_flags);
        t.err = reinterpret_cast<int *>(ws + p.off_err);
        t.B = p.team_B;
        t.teams_per_xcd = p.team_per_xcd;
        t.w_bytes = p.g.CTpad * p.g.NCH * 1024;
        t.seg_off = seg_off;
        t.S = S;
        t.C = C;
        t.stat0 = stat0;
        t.stat1 = stat1;
        {
            const char *e = getenv("SKM_DEBUG_TEAM_DROP");
            t.debug_drop = (e && atoi(e) != 0) ? 1 : 0;
        }
#ifdef SKM_TRACE
        t.trace = g_trace;
#endif
        SKM_HIP(hipMemsetAsync(t.flags, 0, p.off_xch - p.off_flags, st));
        void *kargs[] = {&t};
        hipError_t le = hipErrorInvalidValue;
        {
            SkmProfScope ps(ctx, "k_bw_team", st);
#define SKM_TEAM_LAUNCH(NJT_, B_, GEO_)                                                                            \
    do {                                                                                                          \
        if ((rc = set_lds(k_seg_team<4, NJT_, B_, true, GEO_>, p.team_lds))) return rc;                           \
        le = hipLaunchCooperativeKernel(reinterpret_cast<const void *>(k_seg_team<4, NJT_, B_, true, GEO_>),      \
                                        dim3(p.team_wgs), dim3(256), kargs, (unsigned)p.team_lds, st);            \
    } while (0)
#define SKM_TEAM_GEO(B_)                                                                                           \
    do {                                                                                                          \
        if (geo == 0) SKM_TEAM_LAUNCH(2, B_, 0);                                                                  \
        else if (geo == 1) SKM_TEAM_LAUNCH(2, B_, 1);                                                             \
        else if (geo == 2) SKM_TEAM_LAUNCH(2, B_, 2);                                                             \
        else SKM_TEAM_LAUNCH(2, B_, 3);                                                                           \
    } while (0)
            const int geo = ((p.tg.NP & 1) ? 2 : 0) | (p.tg.tail ? 1 : 0);
#define SKM_BWS_LAUNCH(NCT_, NJT_, B_)                                                                             \
    do {                                                                                                          \
        if ((rc = set_lds(k_seg_team<NCT_, NJT_, B_, true, -1>, p.team_lds))) return rc;                          \
        le = hipLaunchCooperativeKernel(reinterpret_cast<const void *>(k_seg_team<NCT_, NJT_, B_, true, -1>),     \
                                        dim3(p.team_wgs), dim3(256), kargs, (unsigned)p.team_lds, st);            \
    } while (0)
#define SKM_BWS_WIDE(NJT_)                                                                                         \
    do {                                                                                                          \
        if (p.team_nct == 1) SKM_BWS_LAUNCH(1, NJT_, 1);                                                          \
        else if (p.team_B == 1) SKM_BWS_LAUNCH(2, NJT_, 1);                                                       \
        else if (p.team_B == 2) SKM_BWS_LAUNCH(2, NJT_, 2);                                                       \
        else if (p.team_B == 4) SKM_BWS_LAUNCH(2, NJT_, 4);                                                       \
        else SKM_BWS_LAUNCH(2, NJT_, 8);                                                                          \
    } while (0)
            if (p.team_nwv == 2) {      // one or two component tiles: two-wave workgroups, two per CU
#define SKM_BWP_LAUNCH(NJT_, GEN_)                                                                                 \
    do {                                                                                                          \
        if ((rc = set_lds(k_seg_team<1, NJT_, 1, true, -1, GEN_, 2>, p.team_lds))) return rc;                     \
        le = hipLaunchCooperativeKernel(reinterpret_cast<const void *>(k_seg_team<1, NJT_, 1, true, -1, GEN_, 2>), \
                                        dim3(p.team_wgs), dim3(128), kargs, (unsigned)p.team_lds, st);            \
    } while (0)
                if (p.g.NJT == 1) SKM_BWP_LAUNCH(1, true);
                else if (p.g.NJT == 2) SKM_BWP_LAUNCH(2, true);
                else if (p.g.NJT == 3) SKM_BWP_LAUNCH(3, false);
                else if (p.g.NJT == 4) SKM_BWP_LAUNCH(4, false);
                else SKM_BWP_LAUNCH(5, false);
#undef SKM_BWP_LAUNCH
            } else if (p.team_nct != 4) {      // smaller component blocks / wide rows (run-time geometry of the logits ending)
                if (p.g.NJT == 1) {
                    if (p.team_nct == 1) SKM_BWS_LAUNCH(1, 1, 1);
                    else SKM_BWS_LAUNCH(2, 1, 1);
                } else if (p.g.NJT == 2) {
                    if (p.team_nct == 1) SKM_BWS_LAUNCH(1, 2, 1);
                    else SKM_BWS_LAUNCH(2, 2, 1);
                } else if (p.g.NJT == 3) SKM_BWS_WIDE(3);
                else if (p.g.NJT == 4) SKM_BWS_WIDE(4);
                else SKM_BWS_WIDE(5);
            } else
            if (p.g.NJT == 1) {
                if (p.team_B == 1) SKM_TEAM_LAUNCH(1, 1, -1);
                else if (p.team_B == 2) SKM_TEAM_LAUNCH(1, 2, -1);
                else if (p.team_B == 4) SKM_TEAM_LAUNCH(1, 4, -1);
                else if (p.team_B == 8) SKM_TEAM_LAUNCH(1, 8, -1);
            } else if (!p.tg.vec4) {    // rows that are not a multiple of 16 bytes: the staging path for any D
#define SKM_TEAMG_LAUNCH(B_, GEO_)                                                                                 \
    do {                                                                                                          \
        if ((rc = set_lds(k_seg_team<4, 2, B_, true, GEO_, true>, p.team_lds))) return rc;                        \
        le = hipLaunchCooperativeKernel(reinterpret_cast<const void *>(k_seg_team<4, 2, B_, true, GEO_, true>),   \
                                        dim3(p.team_wgs), dim3(256), kargs, (unsigned)p.team_lds, st);            \
    } while (0)
#define SKM_TEAMG_GEO(B_)                                                                                          \
    do {                                                                                                          \
        if (geo == 0) SKM_TEAMG_LAUNCH(B_, 0);                                                                    \
        else if (geo == 1) SKM_TEAMG_LAUNCH(B_, 1);                                                               \
        else if (geo == 2) SKM_TEAMG_LAUNCH(B_, 2);                                                               \
        else SKM_TEAMG_LAUNCH(B_, 3);                                                                             \
    } while (0)
                if (p.team_B == 1) SKM_TEAMG_GEO(1);
                else if (p.team_B == 2) SKM_TEAMG_GEO(2);
                else if (p.team_B == 4) SKM_TEAMG_GEO(4);
                else if (p.team_B == 8) SKM_TEAMG_GEO(8);
#undef SKM_TEAMG_GEO
#undef SKM_TEAMG_LAUNCH
            } else {
                if (p.team_B == 1) SKM_TEAM_GEO(1);
                else if (p.team_B == 2) SKM_TEAM_GEO(2);
                else if (p.team_B == 4) SKM_TEAM_GEO(4);
                else if (p.team_B == 8) SKM_TEAM_GEO(8);
            }
#undef SKM_TEAM_GEO
#undef SKM_TEAM_LAUNCH
#undef SKM_BWS_WIDE
#undef SKM_BWS_LAUNCH
        }
        if (le == hipSuccess) {
            // a timed-out hand-off must not return numbers: consume the flag on the stream
            SkmProfScope ps(ctx, "k_bw_poison", st);
            hipLaunchKernelGGL(k_bw_poison, dim3(ctx->cu_count * 4), dim3(256), 0, st, t.err, stat0, (long)S * C, stat1,
                               (long)S * C * D);
            SKM_HIP(hipGetLastError());
            return SKM_OK;
        }
        (void)hipGetLastError();   // the teams cannot all be resident on this device: two sweeps instead
    }
    if ((rc = launch_lse(ctx, p, X, T, wp, kc, ws, nullptr, st))) return rc;
    StatsArgs a;
    memset(&a, 0, sizeof(a));
    a.X = X;
    a.T = T;
    a.g = p.tg;
    a.wp = wp;
    a.kc = kc;
    a.lse2 = reinterpret_cast<const float *>(ws + p.off_lse);
    a.n_cb = p.n_cb2;
    a.n_fs = p.n_fs2;
    a.C = C;
    a.seg_off = seg_off;
    a.S = S;
    a.stat0 = stat0;
    a.stat1 = stat1;
    return launch_stats<false, true>(ctx, p, a, st);
}

extern "C" int skm_bw_team_count(void *ctx_, int C, int D)
{
    SkmCtx *ctx = static_cast<SkmCtx *>(ctx_);
    Plan p;
    if (!ctx || C <= 0 || D <= 0 || make_plan(ctx, 0, C, D, false, &p) != SKM_OK) return 0;
    return p.team ? 8 * p.team_per_xcd : 0;
}

extern "C" size_t skm_moments_workspace_bytes(void *ctx_, int D)
{
    SkmCtx *ctx = static_cast<SkmCtx *>(ctx_);
    if (!ctx || D <= 0) return 0;
    const size_t nslots = (size_t)ctx->cu_count * 4 * 4;
    return skm_align_up(nslots * 2 * (size_t)D * sizeof(double), 256);
}

extern "C" int skm_moments(void *ctx_, const float *X, long T, int D, double *out, void *ws, size_t ws_bytes,
                           void *stream)
{
    SkmCtx *ctx = static_cast<SkmCtx *>(ctx_);
    SKM_REQUIRE(ctx && out && ws && (T == 0 || X), "skm_moments: NULL argument");
    SKM_REQUIRE(D > 0 && T >= 0, "skm_moments: bad shape");
    if (ws_bytes < skm_moments_workspace_bytes(ctx_, D)) {
        skm_set_error("skm_moments: workspace too small");
        return SKM_ERR_WORKSPACE;
    }
    const int blocks = ctx->cu_count * 4, nslots = blocks * 4;
    hipStream_t st = (hipStream_t)stream;
    SKM_HIP(hipMemsetAsync(ws, 0, (size_t)nslots * 2 * D * sizeof(double), st));
    hipLaunchKernelGGL(k_moments, dim3(blocks), dim3(256), 0, st, X, T, D, static_cast<double *>(ws));
    SKM_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_moments_reduce, dim3((2 * D + 127) / 128), dim3(128), 0, st,
                       static_cast<const double *>(ws), nslots, D, out);
    SKM_HIP(hipGetLastError());
    return SKM_OK;
}

// ------------------------------------------------------------------------------------------------
// Frame centring for un-normalised features.  The logits are accumulated in float32, so their absolute error grows
// with sum_d mu_d^2 / sigma_d^2; raw cepstra (c0 / log-energy offsets of tens, no cmvn) put that in the thousands and
// the posteriors lose the 1e-4 bar.  Everything on this path is translation invariant: with x' = x - m and
// mu' = mu - m the posteriors, N and the log-likelihood are unchanged, and
//   F = F' + N m,   S = S' + 2 m F' + N m^2   (per component),   stat1 = stat1' + stat0 (x) m   (per session).
// The shift every kernel applies is the float32 rounding of shift[d] (so that x - m is the same number on both
// sides of the identity); the host packs the model with the same rounded value.
// ------------------------------------------------------------------------------------------------
// Log-likelihood of an E-step run on an image relative to the reference Gaussian R = N(0, diag(1 / riv)): the kernels
// summed log(p(x_t) / R(x_t)); sum_t log R(x_t) = T (-D/2 log 2 pi + 1/2 sum_d log riv_d) - 1/2 sum_d riv_d sum_t x_td^2,
// and sum_t x_td^2 = sum_c S_cd because the posteriors of a frame sum to one.  One workgroup, fixed order, float64.
__global__ __launch_bounds__(256) void k_llk_ref(double *__restrict__ stats, int C, int D, double T,
                                                 const double *__restrict__ riv)
{
    __shared__ double part[256];
    const double *S = stats + (size_t)C + (size_t)C * D;
    double acc = 0.0;
    for (int d = threadIdx.x; d < D; d += 256) {
        double col = 0.0;
        for (int c = 0; c < C; ++c) col += S[(size_t)c * D + d];
        const double r = riv[d];
        acc += (r > 0.0 ? 0.5 * T * log(r) : 0.0) - 0.5 * r * col;
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) part[threadIdx.x] += part[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) stats[(size_t)C + 2 * (size_t)C * D] += part[0] - 0.5 * T * D * log(2.0 * M_PI);
}

extern "C" int skm_gmm_llk_ref(void *ctx_, double *stats_dev, int C, int D, long T, const double *ref_invvar_dev,
                               void *stream)
{
    SkmCtx *ctx = static_cast<SkmCtx *>(ctx_);
    SKM_REQUIRE(ctx && stats_dev && ref_invvar_dev && C > 0 && D > 0 && T >= 0, "skm_gmm_llk_ref: bad argument");
    SkmProfScope ps(ctx, "k_llk_ref", (hipStream_t)stream);
    hipLaunchKernelGGL(k_llk_ref, dim3(1), dim3(256), 0, (hipStream_t)stream, stats_dev, C, D, (double)T, ref_invvar_dev);
    SKM_HIP(hipGetLastError());
    return SKM_OK;
}

__global__ __launch_bounds__(256) void k_frames_center(const float *__restrict__ X, long n, int D,
                                                       const double *__restrict__ shift, float *__restrict__ out)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = X[i] - (float)shift[(int)(i % D)];
}

// rows of D % 4 == 0 floats: 16 bytes per lane, the four columns of a vector stay inside one row
__global__ __launch_bounds__(256) void k_frames_center4(const f32x4 *__restrict__ X, long n4, int D4,
                                                        const double *__restrict__ shift, f32x4 *__restrict__ out)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const int d = 4 * (int)(i % D4);
        f32x4 v = X[i];
        v.x -= (float)shift[d];
        v.y -= (float)shift[d + 1];
        v.z -= (float)shift[d + 2];
        v.w -= (float)shift[d + 3];
        out[i] = v;
    }
}

extern "C" int skm_frames_center(void *ctx_, const float *X, long T, int D, const double *shift_dev, float *out,
                                 void *stream)
{
    SkmCtx *ctx = static_cast<SkmCtx *>(ctx_);
    SKM_REQUIRE(ctx && shift_dev && D > 0 && T >= 0 && (T == 0 || (X && out)), "skm_frames_center: bad argument");
    if (T == 0) return SKM_OK;
    hipStream_t st = (hipStream_t)stream;
    SkmProfScope ps(ctx, "k_frames_center", st);
    const long n = T * D;
    if (D % 4 == 0 && (((uintptr_t)X | (uintptr_t)out) & 15) == 0) {
        const long n4 = n / 4;
        long blocks = (n4 + 255) / 256;
        if (blocks > (long)ctx->cu_count * 16) blocks = (long)ctx->cu_count * 16;
        hipLaunchKernelGGL(k_frames_center4, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<const f32x4 *>(X),
                           n4, D / 4, shift_dev, reinterpret_cast<f32x4 *>(out));
    } else {
        long blocks = (n + 255) / 256;
        if (blocks > (long)ctx->cu_count * 16) blocks = (long)ctx->cu_count * 16;
        hipLaunchKernelGGL(k_frames_center, dim3((unsigned)blocks), dim3(256), 0, st, X, n, D, shift_dev, out);
    }
    SKM_HIP(hipGetLastError());
    return SKM_OK;
}

__global__ __launch_bounds__(256) void k_stats_unshift(double *__restrict__ stats, int C, int D,
                                                       const double *__restrict__ shift)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)C * D) return;
    const int c = (int)(e / D), d = (int)(e - (long)c * D);
    const double m = (double)(float)shift[d];
    const double n = stats[c], f = stats[(size_t)C + e], s = stats[(size_t)C + (size_t)C * D + e];
    stats[(size_t)C + e] = f + n * m;
    stats[(size_t)C + (size_t)C * D + e] = s + m * (2.0 * f + n * m);
}

extern "C" int skm_gmm_stats_unshift(void *ctx_, double *stats_dev, int C, int D, const double *shift_dev, void *stream)
{
    SkmCtx *ctx = static_cast<SkmCtx *>(ctx_);
    SKM_REQUIRE(ctx && stats_dev && shift_dev && C > 0 && D > 0, "skm_gmm_stats_unshift: bad argument");
    SkmProfScope ps(ctx, "k_stats_unshift", (hipStream_t)stream);
    hipLaunchKernelGGL(k_stats_unshift, dim3((unsigned)(((long)C * D + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       stats_dev, C, D, shift_dev);
    SKM_HIP(hipGetLastError());
    return SKM_OK;
}

__global__ __launch_bounds__(256) void k_bw_unshift(const double *__restrict__ stat0, double *__restrict__ stat1, long S,
                                                    int C, int D, const double *__restrict__ shift)
{
    const long CD = (long)C * D, n = S * CD;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long s = i / CD;
        const int r = (int)(i - s * CD), c = r / D, d = r - c * D;
        stat1[i] += stat0[s * C + c] * (double)(float)shift[d];
    }
}

extern "C" int skm_bw_unshift(void *ctx_, const double *stat0_dev, double *stat1_dev, long S, int C, int D,
                              const double *shift_dev, void *stream)
{
    SkmCtx *ctx = static_cast<SkmCtx *>(ctx_);
    SKM_REQUIRE(ctx && shift_dev && C > 0 && D > 0 && S >= 0 && (S == 0 || (stat0_dev && stat1_dev)),
                "skm_bw_unshift: bad argument");
    if (S == 0) return SKM_OK;
    SkmProfScope ps(ctx, "k_bw_unshift", (hipStream_t)stream);
    long blocks = (S * (long)C * D + 255) / 256;
    if (blocks > (long)ctx->cu_count * 16) blocks = (long)ctx->cu_count * 16;
    hipLaunchKernelGGL(k_bw_unshift, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, stat0_dev, stat1_dev, S, C,
                       D, shift_dev);
    SKM_HIP(hipGetLastError());
    return SKM_OK;
}

// ------------------------------------------------------------------------------------------------
// sum_log_probabilities on a T x C float64 matrix the caller already holds (sidekit/mixture.py:52-64,
// sidekit/statserver.py:83-95): row maximum, log_lk = max + log sum exp(lp - max) (the maximum itself where it is not
// finite), pp = exp(lp - log_lk).  One wave per row, float64, fixed order.  The statistics kernels never need this --
// they fuse it -- it serves callers that import the helper by name.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_row_lse(const double *__restrict__ lp, long T, int C, double *__restrict__ pp,
                                                 double *__restrict__ log_lk)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    const double *src = lp + row * C;
    double m = -__builtin_inf();
    bool has_nan = false;
    for (int c = lane; c < C; c += 64) {
        const double v = src[c];
        has_nan |= (v != v);
        m = fmax(m, v);
    }
#pragma unroll
    for (int k = 32; k > 0; k >>= 1) {
        m = fmax(m, __shfl_xor(m, k));
        has_nan |= (bool)__shfl_xor((int)has_nan, k);
    }
    if (has_nan) m = __builtin_nan("");               // numpy.max propagates NaN
    double s = 0.0;
    for (int c = lane; c < C; c += 64) s += exp(src[c] - m);
#pragma unroll
    for (int k = 32; k > 0; k >>= 1) s += __shfl_xor(s, k);
    double l = m + log(s);
    if (!isfinite(m)) l = m;
    if (lane == 0 && log_lk) log_lk[row] = l;
    if (pp)
        for (int c = lane; c < C; c += 64) pp[row * C + c] = exp(src[c] - l);
}

extern "C" int skm_row_logsumexp(void *ctx_, const double *lp_dev, long T, int C, double *pp_dev, double *loglk_dev,
                                 void *stream)
{
    SkmCtx *ctx = static_cast<SkmCtx *>(ctx_);
    SKM_REQUIRE(ctx && T >= 0 && C > 0 && (T == 0 || lp_dev), "skm_row_logsumexp: bad argument");
    if (T == 0) return SKM_OK;
    SkmProfScope ps(ctx, "k_row_lse", (hipStream_t)stream);
    hipLaunchKernelGGL(k_row_lse, dim3((unsigned)((T + 3) / 4)), dim3(256), 0, (hipStream_t)stream, lp_dev, T, C, pp_dev,
                       loglk_dev);
    SKM_HIP(hipGetLastError());
    return SKM_OK;
}

// ------------------------------------------------------------------------------------------------
// per-segment mean of a float32 vector in float64 (gmm_scoring: mean frame log-likelihood of a test
// segment, sidekit/gmm_scoring.py:107,116).  One workgroup per segment, fixed summation order.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_segment_mean(const float *__restrict__ v, const int64_t *__restrict__ seg_off,
                                                      double *__restrict__ out)
{
    __shared__ double part[4];
    const long a = seg_off[blockIdx.x], b = seg_off[blockIdx.x + 1];
    double s = 0.0;
    for (long t = a + threadIdx.x; t < b; t += 256) s += (double)v[t];
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) s += __shfl_xor(s, m);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double tot = (part[0] + part[1]) + (part[2] + part[3]);
        out[blockIdx.x] = tot / (double)(b - a);   // empty segment: 0/0 = NaN, numpy's mean of nothing
    }
}

extern "C" int skm_segment_mean(void *ctx_, const float *v, const int64_t *seg_off, int S, double *out, void *stream)
{
    SkmCtx *ctx = static_cast<SkmCtx *>(ctx_);
    SKM_REQUIRE(ctx && S >= 0 && (S == 0 || (seg_off && out)), "skm_segment_mean: NULL argument");
    if (S == 0) return SKM_OK;
    SkmProfScope ps(ctx, "k_segment_mean", (hipStream_t)stream);
    hipLaunchKernelGGL(k_segment_mean, dim3(S), dim3(256), 0, (hipStream_t)stream, v, seg_off, out);
    SKM_HIP(hipGetLastError());
    return SKM_OK;
}
