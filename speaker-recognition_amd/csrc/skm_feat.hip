// skm_feat.hip -- feature post-processing in front of the statistics path, on the device (gfx950).
//
// Replaces FeaturesServer.post_processing (sidekit/features_server.py:202-246) for a batch of sessions whose
// raw frames are already in HBM: column mask (:248-258), RASTA filtering (frontend/normfeat.py:45-74 and
// features_server.py:308-324), delta / double delta (frontend/features.py:136-169), cms / cmvn
// (frontend/normfeat.py:77-126) and the selection of the VAD-labelled frames (:241-244), so that the frames
// the E-step and Baum-Welch kernels read never visit the host.
//
// The work is HBM-bound streaming (a 5-tap FIR + one pole, two 7-tap FIRs and an affine map per value).
// One lane owns one (session, column) pair and walks the session's frames in order, which keeps the
// recursion of the RASTA pole and the running count of labelled frames private to the lane; adjacent lanes
// own adjacent columns, so a wave reads and writes whole rows.  Two walks: `k_feat_stats` (mean and
// standard deviation over the labelled frames; skipped without cms/cmvn or with global statistics) and
// `k_feat_write` (recomputes the filters -- cheaper than a round trip of the 3x wider un-normalised
// matrix through HBM -- and stores the normalised rows, compacted when only labelled frames are kept).
// All arithmetic is float64; deltas are rounded to float32 where the reference stores them in a float32
// array (frontend/features.py:159).  No atomics: results are bit-identical from run to run.
#include "skm_common.h"

struct FeatFilt {
    double b[5];   // RASTA numerator
    double pole;   // 0.94
    double f[7];   // delta filter
};

__device__ __forceinline__ double feat_load(const void *X, int in_f64, long idx)
{
    return in_f64 ? static_cast<const double *>(X)[idx] : (double)static_cast<const float *>(X)[idx];
}

// Label of frame t of a session (features_server.py:323: under RASTA the first two labels are the third one).
__device__ __forceinline__ bool feat_label(const uint8_t *lab, long base, long t, long T, int rasta)
{
    if (!lab) return true;
    if (rasta && t < 2) t = (T > 2) ? 2 : T - 1;
    return lab[base + t] != 0;
}

// MODE 0: statistics (mu, 1/sd per session and output column).  MODE 1: write.
template <int NDELTA, int RASTA, int MODE>
__global__ __launch_bounds__(256) void k_feat_walk(
    const void *__restrict__ X, int in_f64, long ldx, const int *__restrict__ cols,
    const uint8_t *__restrict__ lab, const int64_t *__restrict__ seg_off, int S, int D0, FeatFilt ff,
    int norm,                      // 0 none, 1 cms, 2 cmvn
    double *__restrict__ mu, double *__restrict__ isd, long stat_stride,   // [S or 1][Dout]; stride 0 = global
    const int64_t *__restrict__ out_off, int keep_all, void *__restrict__ out, int out_f64,
    uint8_t *__restrict__ lab_out)
{
    constexpr int LA = 3 * NDELTA;          // frames of look-ahead before a row is complete
    constexpr int NO = NDELTA + 1;          // output blocks: [static | delta | double delta]
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (long)S * D0) return;
    const int s = (int)(g / D0), c = (int)(g % D0);
    const long base = seg_off[s];
    const long T = seg_off[s + 1] - base;
    if (T <= 0) return;
    const int col = cols ? cols[c] : c;
    const int Dout = NO * D0;

    double xw[5] = {0, 0, 0, 0, 0};
    double yprev = 0.0;
    double yw[7] = {0, 0, 0, 0, 0, 0, 0};
    float dw[7] = {0, 0, 0, 0, 0, 0, 0};

    // MODE 0 accumulators: sums of (v - shift) and (v - shift)^2 over the labelled frames
    double sh[NO], m1[NO], m2[NO];
#pragma unroll
    for (int k = 0; k < NO; ++k) sh[k] = m1[k] = m2[k] = 0.0;
    long nlab = 0;

    // MODE 1 constants
    double cm[NO], ci[NO];
#pragma unroll
    for (int k = 0; k < NO; ++k) {
        cm[k] = 0.0;
        ci[k] = 1.0;
    }
    long row0 = 0;
    if constexpr (MODE == 1) {
        if (norm) {
#pragma unroll
            for (int k = 0; k < NO; ++k) {
                cm[k] = mu[(long)s * stat_stride + k * D0 + c];
                ci[k] = isd[(long)s * stat_stride + k * D0 + c];
            }
        }
        row0 = out_off[s];
    }

    for (long n = 0; n < T + LA; ++n) {
        double y = 0.0;
        if (n < T) {
            const double x = feat_load(X, in_f64, (base + n) * ldx + col);
            if (RASTA) {
                xw[0] = xw[1]; xw[1] = xw[2]; xw[2] = xw[3]; xw[3] = xw[4]; xw[4] = x;
                if (n >= 4) {
                    // normfeat.py:64-72: frames 0..3 only prime the FIR state; the pole starts at frame 4
                    y = ff.b[0] * xw[4] + ff.b[1] * xw[3] + ff.b[2] * xw[2] + ff.b[3] * xw[1] + ff.b[4] * xw[0]
                        + ff.pole * yprev;
                }
                yprev = y;
            } else {
                y = x;
            }
        }
        double v[3] = {0.0, 0.0, 0.0};
        if constexpr (NDELTA == 0) {
            v[0] = y;
        } else {
#pragma unroll
            for (int k = 0; k < 6; ++k) yw[k] = yw[k + 1];
            yw[6] = y;
            float d = 0.f;
            const long td = n - 3;
            if (td >= 0 && td < T) {
                double a = 0.0;
#pragma unroll
                for (int k = 0; k < 7; ++k) a += ff.f[k] * yw[6 - k];
                d = (float)a;                               // features.py:159,167: stored in a float32 array
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) dw[k] = dw[k + 1];
            dw[6] = d;
            v[0] = yw[6 - LA];
            v[1] = (double)dw[6 - (LA - 3)];
            if constexpr (NDELTA == 2) {
                double a = 0.0;
#pragma unroll
                for (int k = 0; k < 7; ++k) a += ff.f[k] * (double)dw[6 - k];
                v[2] = (double)(float)a;
            }
        }
        const long te = n - LA;
        if (te < 0) continue;
        const bool on = feat_label(lab, base, te, T, RASTA);
        if constexpr (MODE == 0) {
            if (on) {
                if (nlab == 0) {
#pragma unroll
                    for (int k = 0; k < NO; ++k) sh[k] = v[k];
                }
#pragma unroll
                for (int k = 0; k < NO; ++k) {
                    const double dd = v[k] - sh[k];
                    m1[k] += dd;
                    m2[k] += dd * dd;
                }
                ++nlab;
            }
        } else {
            if (keep_all || on) {
                const long row = row0 + (keep_all ? te : nlab);
#pragma unroll
                for (int k = 0; k < NO; ++k) {
                    const double o = (v[k] - cm[k]) * ci[k];
                    const long oi = row * Dout + k * D0 + c;
                    if (out_f64) static_cast<double *>(out)[oi] = o;
                    else static_cast<float *>(out)[oi] = (float)o;
                }
                if (c == 0 && lab_out) lab_out[row] = on ? 1 : 0;
                if (on) ++nlab;
            }
        }
    }

    if constexpr (MODE == 0) {
#pragma unroll
        for (int k = 0; k < NO; ++k) {
            double mean, inv;
            if (nlab == 0 && norm == 2) {
                mean = 0.0;                                 // normfeat.py:122: cmvn leaves a session without speech alone
                inv = 1.0;
            } else {
                const double nn = (double)nlab;             // cms without speech: 0/0 = NaN like numpy.mean of nothing
                const double a = m1[k] / nn;
                mean = sh[k] + a;
                double var = m2[k] / nn - a * a;
                if (var < 0.0) var = 0.0;
                inv = (norm == 2) ? 1.0 / sqrt(var) : 1.0;  // numpy.std: population standard deviation
            }
            mu[(long)s * stat_stride + k * D0 + c] = mean;
            isd[(long)s * stat_stride + k * D0 + c] = inv;
        }
    }
}

// Labelled frames per session (one wave per session), with the RASTA label fix.
__global__ __launch_bounds__(256) void k_feat_count(const uint8_t *__restrict__ lab, const int64_t *__restrict__ seg_off,
                                                     int S, int rasta, int64_t *__restrict__ cnt)
{
    const int lane = threadIdx.x & 63;
    const long s = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= S) return;
    const long base = seg_off[s], T = seg_off[s + 1] - base;
    long n = 0;
    for (long t = lane; t < T; t += 64) n += feat_label(lab, base, t, T, rasta) ? 1 : 0;
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) n += __shfl_xor(n, m);
    if (lane == 0) cnt[s] = n;
}

// Exclusive scan of cnt[S] into off[S+1] (one workgroup; S is a session count).
__global__ __launch_bounds__(1024) void k_feat_scan(const int64_t *__restrict__ cnt, int S, int64_t *__restrict__ off)
{
    __shared__ long part[1024];
    const int t = threadIdx.x;
    const long per = ((long)S + 1023) / 1024;
    const long a = (long)t * per, b = (a + per < S) ? a + per : S;
    long sum = 0;
    for (long i = a; i < b; ++i) sum += cnt[i];
    part[t] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        long v = (t >= d) ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    long run = (t == 0) ? 0 : part[t - 1];
    for (long i = a; i < b; ++i) {
        off[i] = run;
        run += cnt[i];
    }
    if (t == 1023) off[S] = part[1023];
}

// off[s] = seg_off[s] - seg_off[0] (all frames kept: output rows mirror the input rows of the batch)
__global__ __launch_bounds__(256) void k_feat_rebase(const int64_t *__restrict__ seg_off, int S, int64_t *__restrict__ off)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= S) off[i] = seg_off[i] - seg_off[0];
}

template <int MODE>
static void feat_launch(int ndelta, int rasta, dim3 grid, hipStream_t st, const void *X, int in_f64, long ldx,
                        const int *cols, const uint8_t *lab, const int64_t *seg_off, int S, int D0, const FeatFilt &ff,
                        int norm, double *mu, double *isd, long stat_stride, const int64_t *out_off, int keep_all,
                        void *out, int out_f64, uint8_t *lab_out)
{
#define SKM_FEAT_CASE(ND, RA)                                                                                        \
    if (ndelta == ND && rasta == RA) {                                                                               \
        hipLaunchKernelGGL((k_feat_walk<ND, RA, MODE>), grid, dim3(256), 0, st, X, in_f64, ldx, cols, lab, seg_off, \
                           S, D0, ff, norm, mu, isd, stat_stride, out_off, keep_all, out, out_f64, lab_out);        \
        return;                                                                                                      \
    }
    SKM_FEAT_CASE(0, 0)
    SKM_FEAT_CASE(0, 1)
    SKM_FEAT_CASE(1, 0)
    SKM_FEAT_CASE(1, 1)
    SKM_FEAT_CASE(2, 0)
    SKM_FEAT_CASE(2, 1)
#undef SKM_FEAT_CASE
}

extern "C" size_t skm_feat_workspace_bytes(int S, int D0, int ndelta)
{
    if (S < 0 || D0 <= 0 || ndelta < 0 || ndelta > 2) return 0;
    const size_t dout = (size_t)D0 * (ndelta + 1);
    return skm_align_up((size_t)S * dout * 8, 256) * 2 + skm_align_up(((size_t)S + 1) * 8, 256);
}

extern "C" int skm_feat_postproc(void *ctx_, const void *X_dev, int in_f64, long ldx, const int *cols_dev, int D0,
                                 const uint8_t *label_dev, const int64_t *seg_off_dev, int S, int rasta, int ndelta,
                                 const double *delta_filter_host, int norm, const double *gmean_dev,
                                 const double *ginvstd_dev, int keep_all, void *out_dev, int out_f64,
                                 uint8_t *label_out_dev, int64_t *out_off_dev, void *ws_dev, size_t ws_bytes,
                                 void *stream)
{
    SkmCtx *ctx = static_cast<SkmCtx *>(ctx_);
    hipStream_t st = (hipStream_t)stream;
    SKM_REQUIRE(ctx && S >= 0 && D0 > 0 && ldx >= D0, "skm_feat_postproc: bad shape (S=%d, D0=%d, ldx=%ld)", S, D0, ldx);
    SKM_REQUIRE(ndelta >= 0 && ndelta <= 2 && norm >= 0 && norm <= 2 && (rasta == 0 || rasta == 1),
                "skm_feat_postproc: bad option (rasta=%d, ndelta=%d, norm=%d)", rasta, ndelta, norm);
    SKM_REQUIRE(seg_off_dev && out_off_dev && (S == 0 || (X_dev && out_dev)), "skm_feat_postproc: NULL pointer");
    SKM_REQUIRE(norm != 2 || (gmean_dev == nullptr) == (ginvstd_dev == nullptr),
                "skm_feat_postproc: cmvn takes both global statistics or none");
    SKM_REQUIRE((long)S * D0 < (1L << 31) * 256, "skm_feat_postproc: too many (session, column) pairs");
    if (ws_bytes < skm_feat_workspace_bytes(S, D0, ndelta)) {
        skm_set_error("skm_feat_postproc: workspace too small");
        return SKM_ERR_WORKSPACE;
    }
    FeatFilt ff;
    // numpy.arange(.2, -.3, -.1) as the reference builds it (normfeat.py:54): start + i*step in float64
    for (int i = 0; i < 5; ++i) ff.b[i] = 0.2 + i * -0.1;
    ff.pole = 0.94;
    static const double dflt[7] = {.25, .5, .25, 0, -.25, -.5, -.25};          // features_server.py:129
    for (int i = 0; i < 7; ++i) ff.f[i] = delta_filter_host ? delta_filter_host[i] : dflt[i];

    if (S == 0) {
        SKM_HIP(hipMemsetAsync(out_off_dev, 0, 8, st));
        return SKM_OK;
    }
    const size_t dout = (size_t)D0 * (ndelta + 1);
    char *ws = static_cast<char *>(ws_dev);
    double *mu = reinterpret_cast<double *>(ws);
    double *isd = reinterpret_cast<double *>(ws + skm_align_up((size_t)S * dout * 8, 256));
    int64_t *cnt = reinterpret_cast<int64_t *>(ws + 2 * skm_align_up((size_t)S * dout * 8, 256));
    long stat_stride = (long)dout;
    const bool global = norm && gmean_dev && (norm == 1 || ginvstd_dev);
    const dim3 grid((unsigned)(((long)S * D0 + 255) / 256));

    if (keep_all) {
        SkmProfScope ps(ctx, "k_feat_rebase", st);
        hipLaunchKernelGGL(k_feat_rebase, dim3((S + 256) / 256), dim3(256), 0, st, seg_off_dev, S, out_off_dev);
    } else {
        SkmProfScope ps(ctx, "k_feat_count_scan", st);
        hipLaunchKernelGGL(k_feat_count, dim3((S + 3) / 4), dim3(256), 0, st, label_dev, seg_off_dev, S, rasta, cnt);
        hipLaunchKernelGGL(k_feat_scan, dim3(1), dim3(1024), 0, st, cnt, S, out_off_dev);
    }
    SKM_HIP(hipGetLastError());
    const double *mu_use = mu, *isd_use = isd;
    if (global) {
        // global statistics (normfeat.py:93-94, :116-120): one row for every session; cms has no scale
        mu_use = gmean_dev;
        isd_use = (norm == 2) ? ginvstd_dev : nullptr;
        stat_stride = 0;
        if (norm == 1) {
            // a row of ones for the scale
            std::vector<double> ones(dout, 1.0);
            SKM_HIP(hipMemcpyAsync(isd, ones.data(), dout * 8, hipMemcpyHostToDevice, st));
            SKM_HIP(hipStreamSynchronize(st));
            isd_use = isd;
        }
    } else if (norm) {
        SkmProfScope ps(ctx, "k_feat_stats", st);
        feat_launch<0>(ndelta, rasta, grid, st, X_dev, in_f64, ldx, cols_dev, label_dev, seg_off_dev, S, D0, ff, norm,
                       mu, isd, stat_stride, nullptr, 1, nullptr, 0, nullptr);
        SKM_HIP(hipGetLastError());
    }
    {
        SkmProfScope ps(ctx, "k_feat_write", st);
        feat_launch<1>(ndelta, rasta, grid, st, X_dev, in_f64, ldx, cols_dev, label_dev, seg_off_dev, S, D0, ff, norm,
                       const_cast<double *>(mu_use), const_cast<double *>(isd_use), stat_stride, out_off_dev, keep_all,
                       out_dev, out_f64, label_out_dev);
        SKM_HIP(hipGetLastError());
    }
    return SKM_OK;
}
