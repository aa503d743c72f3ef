// dsp_fit.hip -- linear_slope_fit (processors/linear_slope_fit.py:11-91) with one waveform per LANE.
//
// The fit's mean and variance are float32 Welford recurrences that round after every sample and contract a shift of their state, so
// a waveform's samples cannot be shared out among lanes (dsp_vm.hip, op_linear_slope_fit: every lane of the wavefront runs the same
// chain, a float64 division's worth of dependent operations per sample).  Here the 64 lanes of a wavefront run 64 different
// waveforms' chains instead: the same operation sequence per waveform, 64 independent chains per instruction.  The rows are staged
// through LDS in tiles of 64 rows x 64 samples (loaded along the rows, read along the columns, pitch 65).
//
// The fits of the LEGEND recipes read the waveform after the baseline subtraction (bl_subtract.py:11-46 or numpy.subtract) and
// after the pole-zero correction (pole_zero.py:24-77, float64 state, sequential here exactly as the reference runs it); both are
// per-sample recurrences of one waveform, so the lane carries them along and up to DSP_FIT_MAX fits on windows of either stage are
// done in one pass over the row.  Arithmetic of a fit: as op_linear_slope_fit (same typing, same div_by_count, same closing formulas).
#include <hip/hip_runtime.h>

#include "dsp_wave.h"

namespace {

template <typename T>
__device__ __forceinline__ T fit_scalar(const void* p, int dtype, int64_t at) {
    switch (dtype) {
        case DSP_F32: return (T)((const float*)p)[at];
        case DSP_F64: return (T)((const double*)p)[at];
        case DSP_I32: return (T)((const int32_t*)p)[at];
        case DSP_I16: return (T)((const int16_t*)p)[at];
        case DSP_U16: return (T)((const uint16_t*)p)[at];
        default: return (T)((const uint32_t*)p)[at];
    }
}

// A stretch [a, b) of a tile in which the same fits are active: none (K = -1) or exactly fit K -- what the recipes have (a baseline
// window at the start, a tail window after the rise).  Groups of eight samples: their LDS reads first (the lane's column of the tile,
// the reciprocals of the counts), then the recurrences; with the window tests and the LDS reads inside a per-sample loop every sample
// paid two LDS round trips and four uniform branches on top of its dependent chain.
template <typename T, int K>
__device__ __forceinline__ void fit_stretch(const T* mine, const double* invk, int a, int b, int j_a, bool sub, T bsub, bool has_pz, double c, bool stage1,
                                            double& acc, double& xp, bool& nan_y, T& m, T& s, double& sy, double& sxy) {
    constexpr int G = 8;
    auto one = [&](T x, double inv, int j) {
        const T y = sub ? x - bsub : x;  // bl_subtract.py:45 / numpy.subtract: one float subtraction
        nan_y |= (y != y);
        T z = y;
        if (has_pz) {  // pole_zero.py:60-72: float64 state, the store rounds
            const double yd = (double)y;
            acc = (acc + yd) - xp * c;
            z = (T)acc;
            xp = yd;
        }
        if constexpr (K >= 0) {
            const T v = stage1 ? z : y;
            const T temp = v - m;
            m = (T)((double)m + div_by_count((double)temp, (double)(j + 1), inv));
            s = s + temp * (v - m);
            sy += (double)v;
            sxy += (double)v * (double)j;
        }
    };
    int u = a;
    for (; u + G <= b; u += G) {
        T xs[G];
        double iv[G];
#pragma unroll
        for (int k = 0; k < G; ++k) {
            xs[k] = mine[u + k];
            iv[k] = K >= 0 ? invk[u + k] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < G; ++k) one(xs[k], iv[k], j_a + (u - a) + k);
    }
    for (; u < b; ++u) one(mine[u], K >= 0 ? invk[u] : 0.0, j_a + (u - a));
}

template <typename T, typename InT>
__global__ __launch_bounds__(64) void dsp_fit_rows_kernel(FitArgs A) {
    constexpr int TS = 64, PITCH = TS + 1;
    __shared__ T tile[64 * PITCH];
    __shared__ double inv[DSP_FIT_MAX][TS];
    const int lane = lane_id();
    const int64_t r0 = (int64_t)blockIdx.x * 64, row = r0 + lane;
    const bool live = row < A.n_wf;
    const int rows_here = (int)(A.n_wf - r0 < 64 ? A.n_wf - r0 : 64);
    const InT* __restrict__ g = (const InT*)A.wf;
    T b = (T)A.sub_const;
    if (A.sub_mode && A.sub && live) b = fit_scalar<T>(A.sub, A.sub_dtype, row);
    const double c = A.pz_c;
    const bool sub = A.sub_mode != 0, has_pz = A.has_pz != 0;
    double acc = 0.0, xp = 0.0;
    bool nan_y = false;
    T m[DSP_FIT_MAX], s[DSP_FIT_MAX];
    double sy[DSP_FIT_MAX], sxy[DSP_FIT_MAX];
#pragma unroll
    for (int k = 0; k < DSP_FIT_MAX; ++k) {
        m[k] = s[k] = (T)0;
        sy[k] = sxy[k] = 0.0;
    }
    // Whole tiles of whole blocks are fetched with 16-byte loads -- V samples of a row per lane, V rows per instruction -- one tile ahead,
    // into registers, and written to LDS (transposed: one row of the tile per waveform) when the tile before has been consumed; the
    // last rows of a batch, the last samples of a row and rows that are not 16-byte aligned take the element-wise path.
    constexpr int V = 16 / (int)sizeof(InT), NV = TS / V;  // samples per load, loads per lane and tile
    typedef InT vec_t __attribute__((ext_vector_type(V)));
    const bool wide = rows_here == 64 && (A.row_stride * (int64_t)sizeof(InT)) % 16 == 0 && ((uintptr_t)g & 15u) == 0;
    vec_t pf[NV];
    const int pr = lane / (TS / V), pc = (lane % (TS / V)) * V;  // row within a group of V rows, first sample of this lane's piece
    auto fetch = [&](int s0) {
#pragma unroll
        for (int q = 0; q < NV; ++q) pf[q] = *(const vec_t*)(g + (r0 + q * V + pr) * A.row_stride + s0 + pc);
    };
    auto commit = [&]() {
#pragma unroll
        for (int q = 0; q < NV; ++q)
#pragma unroll
            for (int e = 0; e < V; ++e) tile[(q * V + pr) * PITCH + pc + e] = (T)pf[q][e];
    };
    if (wide && A.n_scan >= TS) fetch(0);
    for (int s0 = 0; s0 < A.n_scan; s0 += TS) {
        const int w = A.n_scan - s0 < TS ? A.n_scan - s0 : TS;
        if (wide && w == TS) {
            commit();
            if (s0 + 2 * TS <= A.n_scan) fetch(s0 + TS);
        } else {
            // stage: 64 samples of each of the rows, along the rows (one 64-element coalesced load per row)
#pragma unroll 8
            for (int r = 0; r < rows_here; ++r) tile[r * PITCH + lane] = lane < w ? (T)g[(r0 + r) * A.row_stride + s0 + lane] : (T)0;
        }
        // 1 / (j + 1) for the samples of this tile, one correctly rounded division per lane and fit, off the chains
#pragma unroll
        for (int k = 0; k < DSP_FIT_MAX; ++k) {
            const int j = s0 + lane - A.first[k];
            inv[k][lane] = (k < A.n_fits && j >= 0 && j < A.count[k]) ? 1.0 / (double)(j + 1) : 0.0;
        }
        wave_sync();
        if (live) {
            const T* mine = tile + lane * PITCH;
            // the tile in stretches with a constant set of active fits (all of this is uniform)
            int lo[DSP_FIT_MAX], hi[DSP_FIT_MAX];
#pragma unroll
            for (int k = 0; k < DSP_FIT_MAX; ++k) {
                int l = A.first[k] - s0, h = A.first[k] + A.count[k] - s0;
                l = l < 0 ? 0 : (l > w ? w : l);
                h = h < 0 ? 0 : (h > w ? w : h);
                lo[k] = k < A.n_fits ? l : 0;
                hi[k] = k < A.n_fits ? h : 0;
            }
            for (int cur = 0; cur < w;) {
                int nxt = w, n_act = 0, which = -1;
#pragma unroll
                for (int k = 0; k < DSP_FIT_MAX; ++k) {
                    if (lo[k] > cur && lo[k] < nxt) nxt = lo[k];
                    if (hi[k] > cur && hi[k] < nxt) nxt = hi[k];
                    if (lo[k] <= cur && cur < hi[k]) {
                        ++n_act;
                        which = k;
                    }
                }
                if (n_act == 0) {
                    T dm = (T)0, ds = (T)0;
                    double dy = 0.0, dxy = 0.0;
                    fit_stretch<T, -1>(mine, inv[0], cur, nxt, 0, sub, b, has_pz, c, false, acc, xp, nan_y, dm, ds, dy, dxy);
                } else if (n_act == 1) {
                    const int j_a = s0 + cur - A.first[which];
                    switch (which) {
                        case 0: fit_stretch<T, 0>(mine, inv[0], cur, nxt, j_a, sub, b, has_pz, c, A.stage[0] != 0, acc, xp, nan_y, m[0], s[0], sy[0], sxy[0]); break;
                        case 1: fit_stretch<T, 1>(mine, inv[1], cur, nxt, j_a, sub, b, has_pz, c, A.stage[1] != 0, acc, xp, nan_y, m[1], s[1], sy[1], sxy[1]); break;
                        case 2: fit_stretch<T, 2>(mine, inv[2], cur, nxt, j_a, sub, b, has_pz, c, A.stage[2] != 0, acc, xp, nan_y, m[2], s[2], sy[2], sxy[2]); break;
                        default: fit_stretch<T, 3>(mine, inv[3], cur, nxt, j_a, sub, b, has_pz, c, A.stage[3] != 0, acc, xp, nan_y, m[3], s[3], sy[3], sxy[3]); break;
                    }
                } else {  // windows that overlap: sample by sample, every fit tested
                    for (int u = cur; u < nxt; ++u) {
                        const T x = mine[u];
                        const T y = sub ? x - b : x;
                        nan_y |= (y != y);
                        T z = y;
                        if (has_pz) {
                            const double yd = (double)y;
                            acc = (acc + yd) - xp * c;
                            z = (T)acc;
                            xp = yd;
                        }
#pragma unroll
                        for (int k = 0; k < DSP_FIT_MAX; ++k) {
                            const int j = s0 + u - A.first[k];
                            if (k < A.n_fits && j >= 0 && j < A.count[k]) {  // (uniform)
                                const T v = A.stage[k] ? z : y;
                                const T temp = v - m[k];
                                m[k] = (T)((double)m[k] + div_by_count((double)temp, (double)(j + 1), inv[k][u]));
                                s[k] = s[k] + temp * (v - m[k]);
                                sy[k] += (double)v;
                                sxy[k] += (double)v * (double)j;
                            }
                        }
                    }
                }
                cur = nxt;
            }
        }
        wave_sync();
    }
    if (!live) return;
    T* out = (T*)A.out;
#pragma unroll
    for (int k = 0; k < DSP_FIT_MAX; ++k) {
        if (k >= A.n_fits) break;
        const int n = A.count[k];
        T sk = (T)((double)s[k] / (double)(n - 1));
        sk = (T)sqrt((double)sk);
        const long long nn = n, sum_x = nn * (nn - 1) / 2, sum_x2 = (nn - 1) * nn * (2 * nn - 1) / 6;
        T mean = m[k];
        T slope = (T)(((double)nn * sxy[k] - (double)sum_x * sy[k]) / (double)(nn * sum_x2 - sum_x * sum_x));
        T icpt = (T)((sy[k] - (double)sum_x * (double)slope) / (double)nn);
        // a processor upstream that turns a NaN anywhere into a NaN waveform (bl_subtract, pole_zero) makes the fit NaN; numpy.subtract and
        // a plain slice keep NaN samples single, and one inside the window has made the recurrences NaN by itself
        const bool whole = (A.sub_mode == 1) || A.stage[k];
        if ((whole && nan_y) || (A.stage[k] && A.pz_nan)) mean = sk = slope = icpt = quiet_nan<T>();
        T* o = out + (int64_t)k * 4 * A.n_wf + row;
        o[0] = mean;
        o[A.n_wf] = sk;
        o[2 * A.n_wf] = slope;
        o[3 * A.n_wf] = icpt;
    }
}

template <typename T>
int launch_fit(const FitArgs& A, int wf_dtype, hipStream_t stream) {
    const dim3 grid((unsigned)((A.n_wf + 63) / 64)), block(64);
    switch (wf_dtype) {
        case DSP_F32: hipLaunchKernelGGL((dsp_fit_rows_kernel<T, float>), grid, block, 0, stream, A); break;
        case DSP_F64: hipLaunchKernelGGL((dsp_fit_rows_kernel<T, double>), grid, block, 0, stream, A); break;
        case DSP_I16: hipLaunchKernelGGL((dsp_fit_rows_kernel<T, int16_t>), grid, block, 0, stream, A); break;
        case DSP_U16: hipLaunchKernelGGL((dsp_fit_rows_kernel<T, uint16_t>), grid, block, 0, stream, A); break;
        case DSP_I32: hipLaunchKernelGGL((dsp_fit_rows_kernel<T, int32_t>), grid, block, 0, stream, A); break;
        default: hipLaunchKernelGGL((dsp_fit_rows_kernel<T, uint32_t>), grid, block, 0, stream, A); break;
    }
    return (int)hipGetLastError();
}

}  // namespace

extern "C" int dsp_internal_launch_fit_rows(const FitArgs* A, int wf_dtype, int compute_dtype, hipStream_t stream) {
    if (A->n_wf <= 0) return 0;
    return compute_dtype == DSP_F64 ? launch_fit<double>(*A, wf_dtype, stream) : launch_fit<float>(*A, wf_dtype, stream);
}
