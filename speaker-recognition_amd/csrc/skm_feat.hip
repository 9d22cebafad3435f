// skm_feat.hip -- feature post-processing in front of the statistics path, on the device (gfx950).
//
// Replaces FeaturesServer.post_processing (sidekit/features_server.py:202-246) for a batch of sessions whose
// raw frames are already in HBM: column mask (:248-258), RASTA filtering (frontend/normfeat.py:45-74 and
// features_server.py:308-324), delta / double delta (frontend/features.py:136-169), cms / cmvn
// (frontend/normfeat.py:77-126) and the selection of the VAD-labelled frames (:241-244), so that the frames
// the E-step and Baum-Welch kernels read never visit the host.
//
// The work is HBM-bound streaming (a 5-tap FIR + one pole, two 7-tap FIRs and an affine map per value), with
// float64 VALU work of the same order.  Sessions are cut into chunks of FEAT_L frames; one lane owns one
// (chunk, column) pair and walks its frames in order, adjacent lanes own adjacent columns, so a wave reads and
// writes whole rows and a 2000-frame session gives 8x more waves than lanes-per-session would.  What crosses a
// chunk boundary is made explicit:
//   * the RASTA pole: `k_feat_carry` computes every chunk's zero-state response at its last frame; the state a
//     chunk starts from is E[j-1] + a^L E[j-2] + a^2L E[j-3] (0.94^768 < 1e-20: three terms are exact in float64);
//   * the delta windows: a chunk starts its walk 6 frames early and runs 3*ndelta frames past its end;
//   * the row index of compacted output: labelled frames are counted per chunk and scanned over the batch;
//   * mean / variance: per-chunk (n, mean, M2) merged in chunk order (Chan et al.) by `k_feat_combine`.
// Two walks over the frames: `k_feat_stats` (skipped without cms/cmvn or with global statistics) and
// `k_feat_write` (recomputes the filters -- cheaper than a round trip of the 3x wider un-normalised matrix
// through HBM -- and stores the normalised rows).  All arithmetic is float64; deltas are rounded to float32
// where the reference stores them in a float32 array (frontend/features.py:159).  No atomics anywhere:
// results are bit-identical from run to run.
#include "skm_common.h"

#define FEAT_L 256     // frames per chunk
#define FEAT_H 6       // frames of history a chunk re-walks (two 7-tap windows)
#define FEAT_PF 7      // frames a lane loads ahead of the one it is working on (= one turn of the 7-deep rings)

struct FeatFilt {
    double b[5];    // RASTA numerator
    double pole;    // 0.94
    double poleL;   // pole^FEAT_L        (carry across a full chunk)
    double poleL0;  // pole^(FEAT_L - FEAT_H)  (carry across chunk 0, which is FEAT_H frames shorter)
    double f[7];    // delta filter
};

struct FeatItem {
    int s;          // session
    long j;         // chunk of the session
    long base, T;   // first row of the session in X, its length
    long n0, n1;    // frames [n0, n1) of the session are this chunk's
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

// item -> (session, chunk): largest s with chunk_base[s] <= item (sessions without frames own no chunk).
__device__ __forceinline__ bool feat_item(const int64_t *__restrict__ chunk_base, const int64_t *__restrict__ seg_off,
                                          int S, long item, FeatItem &it)
{
    if (item >= chunk_base[S]) return false;
    int lo = 0, hi = S;                       // invariant: chunk_base[lo] <= item < chunk_base[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (chunk_base[mid] <= item) lo = mid;
        else hi = mid;
    }
    it.s = lo;
    it.j = item - chunk_base[lo];
    it.base = seg_off[lo];
    it.T = seg_off[lo + 1] - it.base;
    it.n0 = it.j * FEAT_L;
    it.n1 = (it.n0 + FEAT_L < it.T) ? it.n0 + FEAT_L : it.T;
    return true;
}

// first frame of the RASTA recursion interval of chunk j (intervals tile the session; chunk 0's is shorter)
__device__ __forceinline__ long feat_b(long j) { return j == 0 ? 0 : j * FEAT_L - FEAT_H; }

// chunks per session
__global__ __launch_bounds__(256) void k_feat_nchunks(const int64_t *__restrict__ seg_off, int S, int64_t *__restrict__ nch)
{
    const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s < S) {
        const long T = seg_off[s + 1] - seg_off[s];
        nch[s] = T > 0 ? (T + FEAT_L - 1) / FEAT_L : 0;
    }
}

// Exclusive scan of cnt[n] into off[n+1] (one workgroup).
__global__ __launch_bounds__(1024) void k_feat_scan(const int64_t *__restrict__ cnt, long n, int64_t *__restrict__ off)
{
    __shared__ long part[1024];
    const int t = threadIdx.x;
    const long per = (n + 1023) / 1024;
    const long a = (long)t * per < n ? (long)t * per : n, b = (a + per < n) ? a + per : n;
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
    if (t == 1023) off[n] = part[1023];
}

// Labelled frames per chunk (one wave per item), with the RASTA label fix; 0 for items past the end.
__global__ __launch_bounds__(256) void k_feat_count(const uint8_t *__restrict__ lab, const int64_t *__restrict__ seg_off,
                                                     const int64_t *__restrict__ chunk_base, int S, long items_max,
                                                     int rasta, int64_t *__restrict__ cnt)
{
    const int lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= items_max) return;
    FeatItem it;
    long n = 0;
    if (feat_item(chunk_base, seg_off, S, item, it)) {
        for (long t = it.n0 + lane; t < it.n1; t += 64) n += feat_label(lab, it.base, t, it.T, rasta) ? 1 : 0;
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) n += __shfl_xor(n, m);
    if (lane == 0) cnt[item] = n;
}

// out_off[s]: first output row of session s (S+1 entries)
__global__ __launch_bounds__(256) void k_feat_outoff(const int64_t *__restrict__ seg_off, const int64_t *__restrict__ chunk_base,
                                                      const int64_t *__restrict__ row_base, int S, int keep_all,
                                                      int64_t *__restrict__ off)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= S) off[i] = keep_all ? seg_off[i] - seg_off[0] : row_base[chunk_base[i]];
}

// Zero-state response of the RASTA filter over a chunk's recursion interval, at the interval's last frame.
__global__ __launch_bounds__(256) void k_feat_carry(const void *__restrict__ X, int in_f64, long ldx,
                                                     const int *__restrict__ cols, const int64_t *__restrict__ seg_off,
                                                     const int64_t *__restrict__ chunk_base, int S, int D0, long items_max,
                                                     FeatFilt ff, double *__restrict__ E)
{
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long item = g / D0;
    const int c = (int)(g % D0);
    if (item >= items_max) return;
    FeatItem it;
    if (!feat_item(chunk_base, seg_off, S, item, it)) return;
    const long b0 = feat_b(it.j), b1 = feat_b(it.j + 1);
    if (b1 >= it.T) return;                   // the last chunk of a session hands nothing on
    const int col = cols ? cols[c] : c;
    double xw[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 1; k <= 4; ++k)
        if (b0 - k >= 0) xw[5 - k] = feat_load(X, in_f64, (it.base + b0 - k) * ldx + col);
    double y = 0.0;
    auto loadx = [&](long n) -> double {                    // clamped: always a legal address
        return feat_load(X, in_f64, (it.base + (n < it.T ? n : it.T - 1)) * ldx + col);
    };
    double xq[FEAT_PF], xn[FEAT_PF];
#pragma unroll
    for (int u = 0; u < FEAT_PF; ++u) xq[u] = loadx(b0 + u);
    for (long nb = b0; nb < b1; nb += FEAT_PF) {
#pragma unroll
        for (int u = 0; u < FEAT_PF; ++u) xn[u] = loadx(nb + FEAT_PF + u);
#pragma unroll
        for (int u = 0; u < FEAT_PF; ++u) {
            const long n = nb + u;
            xw[0] = xw[1]; xw[1] = xw[2]; xw[2] = xw[3]; xw[3] = xw[4];
            xw[4] = xq[u];
            const double yn = ff.b[0] * xw[4] + ff.b[1] * xw[3] + ff.b[2] * xw[2] + ff.b[3] * xw[1] + ff.b[4] * xw[0]
                              + ff.pole * y;
            y = (n >= b1) ? y : ((n >= 4) ? yn : 0.0);
        }
#pragma unroll
        for (int u = 0; u < FEAT_PF; ++u) xq[u] = xn[u];
    }
    E[item * D0 + c] = y;
}

// One frame step at ring position U (compile-time).  The 7-deep rings of x, y and delta are indexed with constants
// (a block of the frame loop is exactly 7 steps, so ring position = step within the block): no value is ever
// moved between registers.  i = frame index relative to the start of the walk (32-bit).
template <int NDELTA, int RASTA, int U>
struct FeatStep {
    static constexpr int LA = 3 * NDELTA;
    static constexpr int at(int back) { return ((U - back) % 7 + 7) % 7; }

    __device__ __forceinline__ static void run(int i, int i4, int iT, int itd0, const FeatFilt &ff, double x,
                                               double (&xr)[7], double (&yr)[7], double (&dr)[7], double &yprev,
                                               double (&v)[3])
    {
        double y;
        if constexpr (RASTA != 0) {
            xr[U] = x;
            // normfeat.py:64-72: frames 0..3 only prime the FIR state; the pole starts at frame 4
            y = ff.b[0] * xr[at(0)] + ff.b[1] * xr[at(1)] + ff.b[2] * xr[at(2)] + ff.b[3] * xr[at(3)] + ff.b[4] * xr[at(4)]
                + ff.pole * yprev;
            y = (i >= i4 && i < iT) ? y : 0.0;
            yprev = y;
        } else {
            y = (i < iT) ? x : 0.0;
        }
        if constexpr (NDELTA == 0) {
            v[0] = y;
        } else {
            yr[U] = y;
            double a = 0.0;
#pragma unroll
            for (int k = 0; k < 7; ++k) a += ff.f[k] * yr[at(k)];
            // features.py:159,167: the convolution is stored in a float32 array; zero outside the session
            dr[U] = (i >= itd0 && i < iT + 3) ? (double)(float)a : 0.0;      // delta of frame i - 3
            v[0] = yr[at(LA)];
            v[1] = dr[at(LA - 3)];
            if constexpr (NDELTA == 2) {
                double a2 = 0.0;
#pragma unroll
                for (int k = 0; k < 7; ++k) a2 += ff.f[k] * dr[at(k)];
                v[2] = (double)(float)a2;
            }
        }
    }
};

// MODE 0: per-chunk statistics (n, mean, M2 per output column).  MODE 1: write.
// The frame loop is kept free of divergent work (every lane of a wave walks another chunk): loads are made
// unconditional by clamping the frame index and every "outside the session" case is a select.  It runs in blocks of
// FEAT_PF = 7 frames: the loads of the next block are issued before the current block is processed (a lane's own
// chain of dependent steps would otherwise expose the memory latency once per frame), and 7 steps per block make
// the ring positions compile-time constants.
template <int NDELTA, int RASTA, int MODE, int IN64, int OUT64>
__global__ __launch_bounds__(256, 4) void k_feat_walk(
    const void *__restrict__ X, long ldx, const int *__restrict__ cols,
    const uint8_t *__restrict__ lab, const int64_t *__restrict__ seg_off, const int64_t *__restrict__ chunk_base,
    int S, int D0, long items_max, FeatFilt ff, const double *__restrict__ E,
    int64_t *__restrict__ part_n, double *__restrict__ part,                  // MODE 0 out: [item], [item][Dout][2]
    int norm, const double *__restrict__ mu, const double *__restrict__ isd, long stat_stride,   // MODE 1 in
    const int64_t *__restrict__ row_base, int keep_all, void *__restrict__ out, uint8_t *__restrict__ lab_out)
{
    static_assert(FEAT_PF == 7, "a block of the frame loop must be one turn of the 7-deep rings");
    constexpr int LA = 3 * NDELTA;          // frames of look-ahead before a row is complete
    constexpr int NO = NDELTA + 1;          // output blocks: [static | delta | double delta]
    constexpr int XB = IN64 ? 8 : 4, OB = OUT64 ? 8 : 4;
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long item = g / D0;
    const int c = (int)(g % D0);
    if (item >= items_max) return;
    FeatItem it;
    if (!feat_item(chunk_base, seg_off, S, item, it)) return;
    const int col = cols ? cols[c] : c;
    const int Dout = NO * D0;
    const long T = it.T, base = it.base, n0 = it.n0;
    const long w0 = feat_b(it.j);           // the walk starts FEAT_H frames before the chunk
    // 32-bit indices relative to w0: frame n = w0 + i
    const long rest = T - w0;
    const int iT = (int)(rest < (1L << 30) ? rest : (1L << 30));     // n < T       <=>  i < iT
    const int i4 = (w0 == 0) ? 4 : 0;                                // n >= 4      <=>  i >= i4   (w0 = 0 or >= FEAT_L - 6)
    const int itd0 = (w0 == 0) ? 3 : 0;                              // n - 3 >= 0  <=>  i >= itd0
    const int i_emit = (int)(n0 - w0) + LA, i_end = (int)(it.n1 - w0) + LA;
    const char *xw0 = static_cast<const char *>(X) + ((base + w0) * ldx + col) * XB;
    const unsigned xstep = (unsigned)(ldx * XB);
    auto loadx = [&](int i) -> double {
        const int ii = i < iT ? i : iT - 1;                          // clamped: always a legal address (iT >= 1)
        const char *p = xw0 + (unsigned long)(unsigned)ii * xstep;
        return IN64 ? *reinterpret_cast<const double *>(p) : (double)*reinterpret_cast<const float *>(p);
    };

    double xr[7] = {0, 0, 0, 0, 0, 0, 0};   // rings: slot (i % 7) holds x[i], y[i]; dr: delta[i - 3]
    double yr[7] = {0, 0, 0, 0, 0, 0, 0};
    double dr[7] = {0, 0, 0, 0, 0, 0, 0};
    double yprev = 0.0;
    if constexpr (RASTA != 0) {
        if (it.j > 0) {
            // x[w0 - k] sits k slots behind slot 0 (w0 >= FEAT_L - FEAT_H > 4 here)
#pragma unroll
            for (int k = 1; k <= 4; ++k) {
                const char *p = xw0 - (long)k * xstep;
                xr[7 - k] = IN64 ? *reinterpret_cast<const double *>(p) : (double)*reinterpret_cast<const float *>(p);
            }
            // state of the pole at frame w0 - 1: the last three intervals (0.94^768 < 1e-20)
            yprev = E[(item - 1) * D0 + c];
            if (it.j >= 2) {
                const double p1 = (it.j - 1 == 0) ? ff.poleL0 : ff.poleL;       // length of interval j-1
                double tail = E[(item - 2) * D0 + c];
                if (it.j >= 3) tail += ((it.j - 2 == 0) ? ff.poleL0 : ff.poleL) * E[(item - 3) * D0 + c];
                yprev += p1 * tail;
            }
        }
    }
    // label of the row completed at step i (frame w0 + i - LA), with the RASTA fix of the first two frames
    const uint8_t *labw = lab ? lab + base + w0 : nullptr;
    const int lfix = (int)((T > 2 ? 2 : T - 1));
    auto label_at = [&](int i) -> bool {
        if (!labw) return true;
        int t = i - LA;
        if (RASTA != 0 && w0 == 0 && t < 2) t = lfix;
        return labw[t] != 0;
    };

    double v[3] = {0.0, 0.0, 0.0};
    double xq[FEAT_PF], xn[FEAT_PF];
#pragma unroll
    for (int u = 0; u < FEAT_PF; ++u) xq[u] = loadx(u);

#define SKM_FEAT_BLOCK(BODY)                                                                          \
    for (int ib = 0; ib < i_end; ib += FEAT_PF) {                                                     \
        _Pragma("unroll") for (int u = 0; u < FEAT_PF; ++u) xn[u] = loadx(ib + FEAT_PF + u);          \
        SKM_FEAT_ONE(0, BODY) SKM_FEAT_ONE(1, BODY) SKM_FEAT_ONE(2, BODY) SKM_FEAT_ONE(3, BODY)       \
        SKM_FEAT_ONE(4, BODY) SKM_FEAT_ONE(5, BODY) SKM_FEAT_ONE(6, BODY)                             \
        _Pragma("unroll") for (int u = 0; u < FEAT_PF; ++u) xq[u] = xn[u];                            \
    }
#define SKM_FEAT_ONE(U, BODY)                                                                         \
    {                                                                                                 \
        const int i = ib + U;                                                                         \
        if (i < i_end) {                                                                              \
            FeatStep<NDELTA, RASTA, U>::run(i, i4, iT, itd0, ff, xq[U], xr, yr, dr, yprev, v);        \
            BODY                                                                                      \
        }                                                                                             \
    }

    if constexpr (MODE == 0) {
        double sh[NO], m1[NO], m2[NO];      // sums of (v - shift), (v - shift)^2 over the labelled frames
#pragma unroll
        for (int k = 0; k < NO; ++k) sh[k] = m1[k] = m2[k] = 0.0;
        int nlab = 0;
        SKM_FEAT_BLOCK({
            const bool on = i >= i_emit && label_at(i >= i_emit ? i : i_emit);
            const bool first = on && nlab == 0;
            _Pragma("unroll") for (int k = 0; k < NO; ++k) {
                sh[k] = first ? v[k] : sh[k];              /* shift = the chunk's first labelled value */
                const double dd = on ? v[k] - sh[k] : 0.0;
                m1[k] += dd;
                m2[k] += dd * dd;
            }
            nlab += on ? 1 : 0;
        })
        if (c == 0) part_n[item] = nlab;
#pragma unroll
        for (int k = 0; k < NO; ++k) {
            double mean = 0.0, M2 = 0.0;
            if (nlab > 0) {
                const double nn = (double)nlab;
                mean = sh[k] + m1[k] / nn;
                M2 = m2[k] - m1[k] * m1[k] / nn;
                if (M2 < 0.0) M2 = 0.0;
            }
            part[(item * Dout + k * D0 + c) * 2 + 0] = mean;
            part[(item * Dout + k * D0 + c) * 2 + 1] = M2;
        }
    } else {
        double cm[NO], ci[NO];              // the affine map of this session and column
#pragma unroll
        for (int k = 0; k < NO; ++k) {
            cm[k] = norm ? mu[(long)it.s * stat_stride + k * D0 + c] : 0.0;
            ci[k] = norm ? isd[(long)it.s * stat_stride + k * D0 + c] : 1.0;
        }
        long orow = keep_all ? (base - seg_off[0]) + n0 : row_base[item];
        char *op = static_cast<char *>(out) + (orow * Dout + c) * OB;
        const long ostep = (long)Dout * OB, oblk = (long)D0 * OB;
        SKM_FEAT_BLOCK({
            if (i >= i_emit) {
                const bool on = label_at(i);
                if (keep_all || on) {
                    _Pragma("unroll") for (int k = 0; k < NO; ++k) {
                        const double o = (v[k] - cm[k]) * ci[k];
                        if constexpr (OUT64 != 0) *reinterpret_cast<double *>(op + k * oblk) = o;
                        else *reinterpret_cast<float *>(op + k * oblk) = (float)o;
                    }
                    if (c == 0 && lab_out) lab_out[orow] = on ? 1 : 0;
                    op += ostep;
                    ++orow;
                }
            }
        })
    }
#undef SKM_FEAT_ONE
#undef SKM_FEAT_BLOCK
}

// Per session and output column: merge the chunks' (n, mean, M2) in chunk order, emit mean and 1/std.
__global__ __launch_bounds__(256) void k_feat_combine(const int64_t *__restrict__ chunk_base, const int64_t *__restrict__ part_n,
                                                       const double *__restrict__ part, int S, int Dout, int norm,
                                                       double *__restrict__ mu, double *__restrict__ isd)
{
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (long)S * Dout) return;
    const int s = (int)(g / Dout), kc = (int)(g % Dout);
    double n = 0.0, mean = 0.0, M2 = 0.0;
    for (long item = chunk_base[s]; item < chunk_base[s + 1]; ++item) {
        const double nb = (double)part_n[item];
        if (nb == 0.0) continue;
        const double mb = part[(item * Dout + kc) * 2], Mb = part[(item * Dout + kc) * 2 + 1];
        if (n == 0.0) {
            n = nb; mean = mb; M2 = Mb;
        } else {
            const double d = mb - mean, nt = n + nb;
            mean += d * (nb / nt);
            M2 += Mb + d * d * (n * nb / nt);
            n = nt;
        }
    }
    double m_out, inv;
    if (n == 0.0 && norm == 2) {
        m_out = 0.0;                                // normfeat.py:122: cmvn leaves a session without speech alone
        inv = 1.0;
    } else {
        m_out = (n == 0.0) ? (0.0 / n) : mean;      // cms without speech: NaN like numpy.mean of nothing
        inv = (norm == 2) ? 1.0 / sqrt(M2 / n) : 1.0;   // numpy.std: population standard deviation
    }
    mu[g] = m_out;
    isd[g] = inv;
}

struct FeatWs {
    int64_t *nch, *chunk_base, *cnt, *row_base, *part_n;
    double *E, *part, *mu, *isd;
    size_t bytes;
};

static FeatWs feat_ws(char *p, long T, int S, int D0, int ndelta)
{
    const size_t items = (size_t)(T / FEAT_L) + (size_t)S + 1;
    const size_t dout = (size_t)D0 * (ndelta + 1);
    FeatWs w;
    size_t o = 0;
    auto take = [&](size_t n) { size_t at = o; o += skm_align_up(n, 256); return p + at; };
    w.nch = reinterpret_cast<int64_t *>(take(((size_t)S + 1) * 8));
    w.chunk_base = reinterpret_cast<int64_t *>(take(((size_t)S + 1) * 8));
    w.cnt = reinterpret_cast<int64_t *>(take(items * 8));
    w.row_base = reinterpret_cast<int64_t *>(take((items + 1) * 8));
    w.part_n = reinterpret_cast<int64_t *>(take(items * 8));
    w.E = reinterpret_cast<double *>(take(items * D0 * 8));
    w.part = reinterpret_cast<double *>(take(items * dout * 16));
    w.mu = reinterpret_cast<double *>(take((size_t)S * dout * 8));
    w.isd = reinterpret_cast<double *>(take(((size_t)S + 1) * dout * 8));
    w.bytes = o;
    return w;
}

template <int MODE, int IN64, int OUT64, typename... Args>
static void feat_launch2(int ndelta, int rasta, dim3 grid, hipStream_t st, Args... args)
{
#define SKM_FEAT_CASE(ND, RA)                                                                              \
    if (ndelta == ND && rasta == RA) {                                                                     \
        hipLaunchKernelGGL((k_feat_walk<ND, RA, MODE, IN64, OUT64>), grid, dim3(256), 0, st, args...);    \
        return;                                                                                            \
    }
    SKM_FEAT_CASE(0, 0)
    SKM_FEAT_CASE(0, 1)
    SKM_FEAT_CASE(1, 0)
    SKM_FEAT_CASE(1, 1)
    SKM_FEAT_CASE(2, 0)
    SKM_FEAT_CASE(2, 1)
#undef SKM_FEAT_CASE
}

template <int MODE, typename... Args>
static void feat_launch(int ndelta, int rasta, int in64, int out64, dim3 grid, hipStream_t st, Args... args)
{
    if (MODE == 0) out64 = 0;               // the statistics walk stores nothing in the output type
    if (in64 && out64) feat_launch2<MODE, 1, 1>(ndelta, rasta, grid, st, args...);
    else if (in64) feat_launch2<MODE, 1, 0>(ndelta, rasta, grid, st, args...);
    else if (out64) feat_launch2<MODE, 0, 1>(ndelta, rasta, grid, st, args...);
    else feat_launch2<MODE, 0, 0>(ndelta, rasta, grid, st, args...);
}

extern "C" size_t skm_feat_workspace_bytes(long T, int S, int D0, int ndelta)
{
    if (T < 0 || S < 0 || D0 <= 0 || ndelta < 0 || ndelta > 2) return 0;
    return feat_ws(nullptr, T, S, D0, ndelta).bytes;
}

extern "C" int skm_feat_postproc(void *ctx_, const void *X_dev, int in_f64, long ldx, const int *cols_dev, int D0,
                                 const uint8_t *label_dev, const int64_t *seg_off_dev, int S, long T, int rasta,
                                 int ndelta, const double *delta_filter_host, int norm, const double *gmean_dev,
                                 const double *ginvstd_dev, int keep_all, void *out_dev, int out_f64,
                                 uint8_t *label_out_dev, int64_t *out_off_dev, void *ws_dev, size_t ws_bytes,
                                 void *stream)
{
    SkmCtx *ctx = static_cast<SkmCtx *>(ctx_);
    hipStream_t st = (hipStream_t)stream;
    SKM_REQUIRE(ctx && S >= 0 && T >= 0 && D0 > 0 && ldx >= D0, "skm_feat_postproc: bad shape (S=%d, T=%ld, D0=%d, ldx=%ld)",
                S, T, D0, ldx);
    SKM_REQUIRE(ndelta >= 0 && ndelta <= 2 && norm >= 0 && norm <= 2 && (rasta == 0 || rasta == 1),
                "skm_feat_postproc: bad option (rasta=%d, ndelta=%d, norm=%d)", rasta, ndelta, norm);
    SKM_REQUIRE(seg_off_dev && out_off_dev && (T == 0 || (X_dev && out_dev)), "skm_feat_postproc: NULL pointer");
    SKM_REQUIRE(norm != 2 || (gmean_dev == nullptr) == (ginvstd_dev == nullptr),
                "skm_feat_postproc: cmvn takes both global statistics or none");
    const long items_max = T / FEAT_L + S;                   // >= sum_s ceil(T_s / FEAT_L)
    SKM_REQUIRE((items_max * D0 + 255) / 256 < (1L << 31), "skm_feat_postproc: too many (chunk, column) pairs");
    if (ws_bytes < skm_feat_workspace_bytes(T, S, D0, ndelta)) {
        skm_set_error("skm_feat_postproc: workspace too small");
        return SKM_ERR_WORKSPACE;
    }
    FeatFilt ff;
    // numpy.arange(.2, -.3, -.1) as the reference builds it (normfeat.py:54): start + i*step in float64
    for (int i = 0; i < 5; ++i) ff.b[i] = 0.2 + i * -0.1;
    ff.pole = 0.94;
    ff.poleL = pow(0.94, (double)FEAT_L);
    ff.poleL0 = pow(0.94, (double)(FEAT_L - FEAT_H));
    static const double dflt[7] = {.25, .5, .25, 0, -.25, -.5, -.25};          // features_server.py:129
    for (int i = 0; i < 7; ++i) ff.f[i] = delta_filter_host ? delta_filter_host[i] : dflt[i];

    if (S == 0 || T == 0) {
        SKM_HIP(hipMemsetAsync(out_off_dev, 0, ((size_t)S + 1) * 8, st));
        return SKM_OK;
    }
    const int dout = D0 * (ndelta + 1);
    FeatWs w = feat_ws(static_cast<char *>(ws_dev), T, S, D0, ndelta);
    const dim3 grid((unsigned)((items_max * D0 + 255) / 256));
    {
        SkmProfScope ps(ctx, "k_feat_index", st);
        hipLaunchKernelGGL(k_feat_nchunks, dim3((S + 255) / 256), dim3(256), 0, st, seg_off_dev, S, w.nch);
        hipLaunchKernelGGL(k_feat_scan, dim3(1), dim3(1024), 0, st, w.nch, (long)S, w.chunk_base);
        if (!keep_all) {
            hipLaunchKernelGGL(k_feat_count, dim3((unsigned)((items_max + 3) / 4)), dim3(256), 0, st, label_dev, seg_off_dev,
                               w.chunk_base, S, items_max, rasta, w.cnt);
            hipLaunchKernelGGL(k_feat_scan, dim3(1), dim3(1024), 0, st, w.cnt, items_max, w.row_base);
        }
        hipLaunchKernelGGL(k_feat_outoff, dim3((S + 256) / 256), dim3(256), 0, st, seg_off_dev, w.chunk_base, w.row_base, S,
                           keep_all, out_off_dev);
        SKM_HIP(hipGetLastError());
    }
    if (rasta) {
        SkmProfScope ps(ctx, "k_feat_carry", st);
        hipLaunchKernelGGL(k_feat_carry, grid, dim3(256), 0, st, X_dev, in_f64, ldx, cols_dev, seg_off_dev, w.chunk_base, S,
                           D0, items_max, ff, w.E);
        SKM_HIP(hipGetLastError());
    }
    const double *mu_use = w.mu, *isd_use = w.isd;
    long stat_stride = dout;
    const bool global = norm && gmean_dev && (norm == 1 || ginvstd_dev);
    if (global) {
        // global statistics (normfeat.py:93-94, :116-120): one row for every session; cms has no scale
        mu_use = gmean_dev;
        isd_use = ginvstd_dev;
        stat_stride = 0;
        if (norm == 1) {
            std::vector<double> ones(dout, 1.0);
            SKM_HIP(hipMemcpyAsync(w.isd, ones.data(), (size_t)dout * 8, hipMemcpyHostToDevice, st));
            SKM_HIP(hipStreamSynchronize(st));
            isd_use = w.isd;
        }
    } else if (norm) {
        {
            SkmProfScope ps(ctx, "k_feat_stats", st);
            feat_launch<0>(ndelta, rasta, in_f64, 0, grid, st, X_dev, ldx, cols_dev, label_dev, seg_off_dev,
                           (const int64_t *)w.chunk_base, S, D0, items_max, ff, (const double *)w.E, w.part_n, w.part, norm,
                           (const double *)nullptr, (const double *)nullptr, 0L, (const int64_t *)nullptr, 1,
                           (void *)nullptr, (uint8_t *)nullptr);
            SKM_HIP(hipGetLastError());
        }
        SkmProfScope ps(ctx, "k_feat_combine", st);
        hipLaunchKernelGGL(k_feat_combine, dim3((unsigned)(((long)S * dout + 255) / 256)), dim3(256), 0, st, w.chunk_base,
                           w.part_n, w.part, S, dout, norm, w.mu, w.isd);
        SKM_HIP(hipGetLastError());
    }
    {
        SkmProfScope ps(ctx, "k_feat_write", st);
        feat_launch<1>(ndelta, rasta, in_f64, out_f64, grid, st, X_dev, ldx, cols_dev, label_dev, seg_off_dev,
                       (const int64_t *)w.chunk_base, S, D0, items_max, ff, (const double *)w.E, (int64_t *)nullptr,
                       (double *)nullptr, norm, mu_use, isd_use, stat_stride, (const int64_t *)w.row_base, keep_all, out_dev,
                       label_out_dev);
        SKM_HIP(hipGetLastError());
    }
    return SKM_OK;
}

// ------------------------------------------------------------------------------------------------
// Index of the labelled frames of a batch (the bookkeeping of the sliding normalisation): idx[k] = row of the k-th
// labelled frame (rows < T in order), coff[s] = labelled frames in front of row seg_off[s] (S + 1 entries;
// coff[S] = labelled frames in front of seg_off[S]).  Three steps, no atomics: labelled frames per block of
// FEAT_IB rows -> exclusive scan -> every wave ranks the labelled frames of its 64-row groups with ballots.
// ------------------------------------------------------------------------------------------------
#define FEAT_IB 4096
__global__ __launch_bounds__(256) void k_label_count(const uint8_t *__restrict__ lab, long T, int64_t *__restrict__ cnt)
{
    __shared__ int part[4];
    const long r0 = (long)blockIdx.x * FEAT_IB;
    int n = 0;
    for (long r = r0 + threadIdx.x; r < r0 + FEAT_IB && r < T; r += 256) n += lab[r] ? 1 : 0;
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) n += __shfl_xor(n, m);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) cnt[blockIdx.x] = (long)part[0] + part[1] + part[2] + part[3];
}

__global__ __launch_bounds__(256) void k_label_index(const uint8_t *__restrict__ lab, long T, const int64_t *__restrict__ boff,
                                                     int64_t *__restrict__ idx)
{
    __shared__ int wcnt[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long r0 = (long)blockIdx.x * FEAT_IB;
    long base = boff[blockIdx.x];
    // 16 rounds of 256 rows: rank inside the wave by ballot, waves of a round by a 4-entry table
    for (int it = 0; it < FEAT_IB / 256; ++it) {
        const long r = r0 + it * 256 + threadIdx.x;
        const bool on = r < T && lab[r] != 0;
        const unsigned long long m = __ballot(on);
        if (lane == 0) wcnt[wave] = __popcll(m);
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            before += (w < wave) ? wcnt[w] : 0;
            total += wcnt[w];
        }
        if (on) idx[base + before + __popcll(m & ((1ull << lane) - 1ull))] = r;
        base += total;
        __syncthreads();
    }
}

// coff[s] = boff[block of seg_off[s]] + labelled rows of that block in front of seg_off[s]   (one wave per entry)
__global__ __launch_bounds__(256) void k_label_coff(const uint8_t *__restrict__ lab, long T, const int64_t *__restrict__ boff,
                                                    const int64_t *__restrict__ seg_off, int S, int64_t *__restrict__ coff)
{
    const int lane = threadIdx.x & 63;
    const long s = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s > S) return;
    const long row = seg_off[s] < T ? seg_off[s] : T;
    const long b = row / FEAT_IB;
    long n = 0;
    for (long r = b * FEAT_IB + lane; r < row; r += 64) n += lab[r] ? 1 : 0;
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) n += __shfl_xor(n, m);
    if (lane == 0) coff[s] = boff[b] + n;
}

extern "C" size_t skm_label_index_workspace_bytes(long T) { return (size_t)(2 * ((T + FEAT_IB - 1) / FEAT_IB + 1) + 2) * 8; }

extern "C" int skm_label_index(void *ctx_, const uint8_t *label, long T, const int64_t *seg_off, int S, int64_t *idx,
                               int64_t *coff, void *ws, size_t ws_bytes, void *stream)
{
    SkmCtx *ctx = static_cast<SkmCtx *>(ctx_);
    SKM_REQUIRE(ctx && label && T > 0 && seg_off && S >= 0 && idx && coff && ws, "skm_label_index: bad argument");
    SKM_REQUIRE(ws_bytes >= skm_label_index_workspace_bytes(T), "skm_label_index: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const long nblk = (T + FEAT_IB - 1) / FEAT_IB;
    int64_t *cnt = static_cast<int64_t *>(ws), *boff = cnt + nblk + 1;
    SkmProfScope ps(ctx, "k_feat_index", st);
    hipLaunchKernelGGL(k_label_count, dim3((unsigned)nblk), dim3(256), 0, st, label, T, cnt);
    hipLaunchKernelGGL(k_feat_scan, dim3(1), dim3(1024), 0, st, cnt, nblk, boff);
    hipLaunchKernelGGL(k_label_index, dim3((unsigned)nblk), dim3(256), 0, st, label, T, boff, idx);
    hipLaunchKernelGGL(k_label_coff, dim3((unsigned)((S + 1 + 3) / 4)), dim3(256), 0, st, label, T, boff, seg_off, S, coff);
    SKM_HIP(hipGetLastError());
    return SKM_OK;
}

// ------------------------------------------------------------------------------------------------
// Sliding-window normalisation (features_server.py:282-287 -> frontend/normfeat.py:194-229): mean and sample
// standard deviation over `win` LABELLED frames centred on each labelled frame, the first / last win/2 positions
// take the nearest full window; only labelled frames change.  A session with at most `win` labelled frames takes
// the whole-session cms / cmvn result Yn (computed by skm_feat_postproc) for ALL its frames.
// One lane per (chunk of 128 positions, column): the window sums are built once per chunk (relative to a pivot
// sample, so the variance does not cancel) and then slid by one frame per step.  Reads Y, writes out (another
// buffer: windows reach into neighbouring chunks).  idx: row of every labelled frame, coff: labelled frames before
// each session.  Float64, no atomics.
// ------------------------------------------------------------------------------------------------
// One output of the sliding normalisation from the window sums (relative to `pivot`).  A float64 result divides as
// the reference does; a float32 result (the frames the statistics kernels read) takes the reciprocals in float32 --
// the float64 divide and square root were most of the kernel's vector instructions.
template <typename TO>
__device__ __forceinline__ TO slide_value(double x, double pivot, double sum, double ssq, double w, double inv_w, int reduce)
{
    const double v = x - (pivot + (sizeof(TO) == 8 ? sum / w : sum * inv_w));
    if (!reduce) return (TO)v;
    if (sizeof(TO) == 8) {
        double var = (ssq - sum * sum / w) / (w - 1.0);
        if (var < 0.0) var = 0.0;
        return (TO)(v / sqrt(var));
    }
    float var = (float)((ssq - sum * sum * inv_w) * (1.0 / (w - 1.0)));
    if (var < 0.f) var = 0.f;
    return (TO)((float)v * __builtin_amdgcn_rsqf(var));     // v_rsq_f32 (1 ulp) instead of sqrt + divide (~20 instructions)
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void k_feat_sliding(const TI *__restrict__ Y, const TI *__restrict__ Yn,
                                                      TO *__restrict__ out, int compact, int D,
                                                      const int64_t *__restrict__ idx,
                                                      const int64_t *__restrict__ coff, const int64_t *__restrict__ seg,
                                                      const int *__restrict__ chunk_sess, const int *__restrict__ chunk_p0,
                                                      int nchunks, int chunk_len, int win, int reduce)
{
    const int lane = threadIdx.x & 63;
    const int chunk = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (chunk >= nchunks) return;
    const int s = chunk_sess[chunk];
    const long p0 = chunk_p0[chunk];
    const long n = coff[s + 1] - coff[s];
    if (n <= win && compact) {      // whole-session normalisation, labelled frames only
        const int64_t *ixs = idx + coff[s];
        const long pe = (p0 + chunk_len < n) ? p0 + chunk_len : n;
        for (long p = p0; p < pe; ++p)
            for (int j = lane; j < D; j += 64) out[(coff[s] + p) * D + j] = (TO)Yn[ixs[p] * D + j];
        return;
    }
    if (n <= win) {     // whole-session normalisation: p0 counts frames of the session
        const long r0 = seg[s] + p0;
        const long r1 = (r0 + chunk_len < seg[s + 1]) ? r0 + chunk_len : seg[s + 1];
        for (long r = r0; r < r1; ++r)
            for (int j = lane; j < D; j += 64) out[r * D + j] = (TO)Yn[r * D + j];
        return;
    }
    const int64_t *ix = idx + coff[s];
    const long dw = win / 2;
    const long p1 = (p0 + chunk_len < n) ? p0 + chunk_len : n;
    const double w = (double)win, inv_w = 1.0 / w;
    for (int j = lane; j < D; j += 64) {
        long c = p0 < dw ? dw : (p0 > n - dw - 1 ? n - dw - 1 : p0);
        const double pivot = (double)Y[ix[c - dw] * D + j];
        double sum = 0.0, ssq = 0.0;
        for (long q = c - dw; q <= c + dw; ++q) {
            const double v = (double)Y[ix[q] * D + j] - pivot;
            sum += v;
            ssq += v * v;
        }
        for (long p = p0; p < p1; ++p) {
            const long cn = p < dw ? dw : (p > n - dw - 1 ? n - dw - 1 : p);
            if (cn != c) {      // the window moves by one labelled frame
                const double vin = (double)Y[ix[cn + dw] * D + j] - pivot;
                const double vout = (double)Y[ix[c - dw] * D + j] - pivot;
                sum += vin - vout;
                ssq += vin * vin - vout * vout;
                c = cn;
            }
            const long r = ix[p];
            out[(compact ? coff[s] + p : r) * D + j] = slide_value<TO>((double)Y[r * D + j], pivot, sum, ssq, w, inv_w, reduce);
        }
    }
}

// The same walk with the window in LDS (sessions with more than `win` labelled frames): a workgroup owns a chunk of
// 16 * SL labelled positions of one session and a block of 16 columns, loads the rows its windows touch --
// positions [c(p0) - dw, c(p1 - 1) + dw], at most 16 SL + win - 1 rows -- ONCE (64-byte pieces of the rows, gathered
// through idx), and 16 x 16 threads = (sub-chunk of SL positions, column) slide their window sums over LDS.
// HBM reads per output row: (16 SL + win) / (16 SL) = 1.35 at SL = 53 (two workgroups per CU) against three row reads per output in the
// lane-per-column kernel above.  SL = 1 (mod 4) with a row stride of 16 words: the four sub-chunks of a wave start
// 16 banks apart.
#define FEAT_SB 16      // columns per block
template <typename TI, typename TO, int SL>
__global__ __launch_bounds__(256) void k_feat_sliding_lds(const TI *__restrict__ Y, TO *__restrict__ out, int compact, int D,
                                                          const int64_t *__restrict__ idx, const int64_t *__restrict__ coff,
                                                          const int *__restrict__ chunk_sess, const int *__restrict__ chunk_p0,
                                                          int nchunks, int win, int reduce)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    TI *tile = reinterpret_cast<TI *>(lds_raw);             // [rows][FEAT_SB]
    // workgroups w, w + 8, w + 16, ... run on one XCD: the column blocks of ONE chunk take ids 8 apart, so the 128-byte
    // lines their 64-byte row pieces share are fetched into one L2 once (as a 2-D grid, chunk-major, the blocks of a
    // chunk ran on different XCDs at different times: 6.1 GB read per launch for 2.4 GB of tiles)
    const int ncb = (D + FEAT_SB - 1) / FEAT_SB;
    const int chunk = (int)(blockIdx.x / (8 * ncb)) * 8 + (int)(blockIdx.x & 7), cb = (int)(blockIdx.x >> 3) % ncb;
    if (chunk >= nchunks) return;
    const int s = chunk_sess[chunk];
    const long p0 = chunk_p0[chunk];
    const long n = coff[s + 1] - coff[s];
    const int64_t *ix = idx + coff[s];
    const long dw = win / 2;
    const long p1 = (p0 + 16 * SL < n) ? p0 + 16 * SL : n;
    auto centre = [&](long p) { return p < dw ? dw : (p > n - dw - 1 ? n - dw - 1 : p); };
    const long qa = centre(p0) - dw, qb = centre(p1 - 1) + dw + 1;     // rows [qa, qb) of the session's labelled frames
    const int j0 = cb * FEAT_SB;
    const int col = threadIdx.x & (FEAT_SB - 1), sub = threadIdx.x / FEAT_SB;
    const bool col_ok = j0 + col < D;
    // Stage 1: the row numbers of the tile's labelled frames, one coalesced pass into LDS.  Stage 2: the row pieces,
    // 16 rows per step and 16 steps in flight per thread (a row number followed by its dependent row load, ~80 times
    // in a row per thread, was most of the kernel's time with two workgroups per CU to hide it).
    const int nrows = (int)(qb - qa);
    int *rown = reinterpret_cast<int *>(tile + (size_t)(16 * SL + win) * FEAT_SB);
    for (int i = threadIdx.x; i < nrows; i += 256) rown[i] = (int)ix[qa + i];
    __syncthreads();
    constexpr int RPS = 256 / FEAT_SB, NFL = 16;
    for (int i0 = sub; i0 < nrows; i0 += RPS * NFL) {
        TI v[NFL];
#pragma unroll
        for (int k = 0; k < NFL; ++k) {
            const int i = i0 + k * RPS;
            v[k] = col_ok ? Y[(long)rown[i < nrows ? i : nrows - 1] * D + j0 + col] : (TI)0;
        }
#pragma unroll
        for (int k = 0; k < NFL; ++k) {
            const int i = i0 + k * RPS;
            if (i < nrows) tile[i * FEAT_SB + col] = v[k];
        }
    }
    __syncthreads();
    const long a = p0 + (long)sub * SL, b = (a + SL < p1) ? a + SL : p1;
    if (a >= b || !col_ok) return;
    // positions relative to the first row of the tile, 32-bit: the slide loop was 122 vector instructions per output,
    // a third of them 64-bit index arithmetic (PMC: SQ_INSTS_VALU 8.0e8 per launch, LDS bank conflicts < 1 %)
    const TI *tc = tile + col;                               // tc[r * FEAT_SB] = value of row r of the tile
    const int ra = (int)(a - qa), rb = (int)(b - qa), idw = (int)dw;
    const int c_lo = (int)(dw - qa), c_hi = (int)(n - dw - 1 - qa);     // centre(p) - qa = clamp(p - qa, c_lo, c_hi)
    auto centre_r = [&](int r) { return r < c_lo ? c_lo : (r > c_hi ? c_hi : r); };
    const double w = (double)win, inv_w = 1.0 / w;
    int c = centre_r(ra);
    const double pivot = (double)tc[(c - idw) * FEAT_SB];
    double sum = 0.0, ssq = 0.0;
#pragma unroll 8
    for (int q = c - idw; q <= c + idw; ++q) {
        const double v = (double)tc[q * FEAT_SB] - pivot;
        sum += v;
        ssq += v * v;
    }
    TO *orow = out + (coff[s] + a) * D + j0 + col;          // compact: consecutive labelled frames, consecutive rows
    for (int r = ra; r < rb; ++r) {
        const int cn = centre_r(r);
        if (cn != c) {      // the window moves by one labelled frame
            const double vin = (double)tc[(cn + idw) * FEAT_SB] - pivot;
            const double vout = (double)tc[(c - idw) * FEAT_SB] - pivot;
            sum += vin - vout;
            ssq += vin * vin - vout * vout;
            c = cn;
        }
        const TO o = slide_value<TO>((double)tc[r * FEAT_SB], pivot, sum, ssq, w, inv_w, reduce);
        if (compact) {
            *orow = o;
            orow += D;
        } else {
            out[(long)rown[r] * D + j0 + col] = o;
        }
    }
}

template <typename TI, typename TO, int SL>
static int feat_sliding_lds_launch(const void *Y, void *out, int compact, int D, const int64_t *idx, const int64_t *coff,
                                   const int *chunk_sess, const int *chunk_p0, int nchunks, int win, int reduce,
                                   hipStream_t st)
{
    const size_t lds = (size_t)(16 * SL + win) * FEAT_SB * sizeof(TI) + (size_t)(16 * SL + win) * sizeof(int);
    SKM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_feat_sliding_lds<TI, TO, SL>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned ncb = (unsigned)((D + FEAT_SB - 1) / FEAT_SB);
    hipLaunchKernelGGL((k_feat_sliding_lds<TI, TO, SL>), dim3((unsigned)((nchunks + 7) / 8) * 8 * ncb), dim3(256), lds, st,
                       (const TI *)Y, (TO *)out, compact, D, idx, coff, chunk_sess, chunk_p0, nchunks, win, reduce);
    SKM_HIP(hipGetLastError());
    return SKM_OK;
}

// positions per chunk of the LDS kernel for an input type (the host cuts the work list with it)
extern "C" int skm_feat_sliding_lds_chunk(int in_f64) { return in_f64 ? 16 * 29 : 16 * 53; }

// chunks of sessions with MORE than `win` labelled frames, chunk length skm_feat_sliding_lds_chunk(in_f64); win <= 301
// keeps the tile inside the 160 KB of LDS
extern "C" int skm_feat_sliding_norm_lds(void *ctx_, const void *Y, int in_f64, void *out, int out_f64, int compact, int D,
                                         const int64_t *idx, const int64_t *coff, const int *chunk_sess,
                                         const int *chunk_p0, int nchunks, int win, int reduce, void *stream)
{
    SkmCtx *ctx = static_cast<SkmCtx *>(ctx_);
    SKM_REQUIRE(ctx && D > 0 && nchunks >= 0 && win >= 3 && (win & 1) && win <= 601 && Y && out && Y != out &&
                    (out_f64 || compact) && (in_f64 || !out_f64) && idx && coff && (nchunks == 0 || (chunk_sess && chunk_p0)),
                "skm_feat_sliding_norm_lds: bad argument");
    if (nchunks == 0) return SKM_OK;
    hipStream_t st = (hipStream_t)stream;
    SkmProfScope ps(ctx, "k_feat_sliding", st);
    if (in_f64 && out_f64)
        return feat_sliding_lds_launch<double, double, 29>(Y, out, compact, D, idx, coff, chunk_sess, chunk_p0, nchunks, win, reduce, st);
    if (in_f64)
        return feat_sliding_lds_launch<double, float, 29>(Y, out, compact, D, idx, coff, chunk_sess, chunk_p0, nchunks, win, reduce, st);
    return feat_sliding_lds_launch<float, float, 53>(Y, out, compact, D, idx, coff, chunk_sess, chunk_p0, nchunks, win, reduce, st);
}

extern "C" int skm_feat_sliding_norm(void *ctx_, const void *Y, const void *Yn, int in_f64, void *out, int out_f64,
                                     int compact, int D,
                                     const int64_t *idx, const int64_t *coff, const int64_t *seg_off,
                                     const int *chunk_sess, const int *chunk_p0, int nchunks, int chunk_len, int win,
                                     int reduce, void *stream)
{
    SkmCtx *ctx = static_cast<SkmCtx *>(ctx_);
    SKM_REQUIRE(ctx && D > 0 && nchunks >= 0 && chunk_len > 0 && win >= 3 && (win & 1) && Y && out && Y != out && (out_f64 || compact) && (in_f64 || !out_f64) && coff && seg_off &&
                    (nchunks == 0 || (chunk_sess && chunk_p0)),
                "skm_feat_sliding_norm: bad argument");
    if (nchunks == 0) return SKM_OK;
    SkmProfScope ps(ctx, "k_feat_sliding", (hipStream_t)stream);
    const dim3 grid((unsigned)((nchunks + 3) / 4));
#define SKM_SLIDE(TI_, TO_)                                                                                         \
    hipLaunchKernelGGL((k_feat_sliding<TI_, TO_>), grid, dim3(256), 0, (hipStream_t)stream, (const TI_ *)Y,            \
                       (const TI_ *)Yn, (TO_ *)out, compact, D, idx, coff, seg_off, chunk_sess, chunk_p0, nchunks,      \
                       chunk_len, win, reduce)
    if (in_f64 && out_f64)
        SKM_SLIDE(double, double);
    else if (in_f64)
        SKM_SLIDE(double, float);
    else
        SKM_SLIDE(float, float);
#undef SKM_SLIDE
    SKM_HIP(hipGetLastError());
    return SKM_OK;
}
