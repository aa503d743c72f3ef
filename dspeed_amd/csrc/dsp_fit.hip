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
    double acc = 0.0, xp = 0.0;
    bool nan_y = false;
    T m[DSP_FIT_MAX], s[DSP_FIT_MAX];
    double sy[DSP_FIT_MAX], sxy[DSP_FIT_MAX];
#pragma unroll
    for (int k = 0; k < DSP_FIT_MAX; ++k) {
        m[k] = s[k] = (T)0;
        sy[k] = sxy[k] = 0.0;
    }
    for (int s0 = 0; s0 < A.n_scan; s0 += TS) {
        const int w = A.n_scan - s0 < TS ? A.n_scan - s0 : TS;
        // stage: 64 samples of each of the rows, along the rows (one 64-element coalesced load per row)
#pragma unroll 8
        for (int r = 0; r < rows_here; ++r) tile[r * PITCH + lane] = lane < w ? (T)g[(r0 + r) * A.row_stride + s0 + lane] : (T)0;
        // 1 / (j + 1) for the samples of this tile, one correctly rounded division per lane and fit, off the chains
#pragma unroll
        for (int k = 0; k < DSP_FIT_MAX; ++k) {
            const int j = s0 + lane - A.first[k];
            inv[k][lane] = (k < A.n_fits && j >= 0 && j < A.count[k]) ? 1.0 / (double)(j + 1) : 0.0;
        }
        wave_sync();
        if (live) {
            const T* mine = tile + lane * PITCH;
            for (int u = 0; u < w; ++u) {
                const T x = mine[u];
                const T y = A.sub_mode ? x - b : x;  // bl_subtract.py:45 / numpy.subtract: one float subtraction
                nan_y |= (y != y);
                T z = y;
                if (A.has_pz) {  // pole_zero.py:60-72: float64 state, the store rounds
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
