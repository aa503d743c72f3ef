// dsp_reduce.hip -- per-event values read straight off rows in HBM: min_max (processors/min_max.py:11-82), numpy.amax, a sample at a
// constant index (fixed_time_pickoff at an integral time, fixed_time_pickoff.py:68-80; `wf[k]`), and time_point_thresh walks that start
// at a constant sample or at the extremes just found (time_point_thresh.py:12-92: the t0 estimate of the Ge recipes).
//
// Every Ge recipe asks for tp_min / tp_max / wf_min / wf_max of the raw waveform and for the maximum and one sample of a filtered one.
// On the waveform VM that is a LOAD of the whole row into LDS, an LDS pass and -- because the 8192-sample image leaves room for one
// wavefront per SIMD -- no latency hiding: 1.6 ms of the recipe's main program per 131 072 rows for the raw waveform alone.  Nothing here
// needs the row twice: a wavefront streams its row through registers, 16 bytes per lane and load, four loads in flight, every lane keeping
// (value, index) of its first minimum and maximum; one exchange across the wavefront at the end.  HBM-bound: the row is read once.
#include <hip/hip_runtime.h>

#include "dsp_program.h"
#include "dsp_wave.h"

namespace {

template <typename IN>
struct RowVec;  // 16 bytes of a row
template <>
struct RowVec<float> {
    static constexpr int N = 4;
    typedef float vec __attribute__((ext_vector_type(4)));
};
template <>
struct RowVec<int16_t> {
    static constexpr int N = 8;
    typedef short vec __attribute__((ext_vector_type(8)));
};
template <>
struct RowVec<uint16_t> {
    static constexpr int N = 8;
    typedef unsigned short vec __attribute__((ext_vector_type(8)));
};

struct Extremes {
    float vmin, vmax;
    int imin, imax;
    bool nan;
};

// strict comparisons in ascending index order: the first occurrence stays (min_max.py:73-77)
__device__ __forceinline__ void take(Extremes& e, float v, int i, bool live) {
    const bool lt = live && v < e.vmin, gt = live && v > e.vmax;
    e.vmin = lt ? v : e.vmin;
    e.imin = lt ? i : e.imin;
    e.vmax = gt ? v : e.vmax;
    e.imax = gt ? i : e.imax;
    e.nan |= live && (v != v);
}

// VEC: row starts, stride and length are whole 16-byte vectors (every Ge waveform is); otherwise a sample per lane and load
template <typename IN, bool VEC>
__global__ void __launch_bounds__(256) dsp_reduce_kernel(ReduceArgs A, int64_t n_wf) {
    constexpr int N = VEC ? RowVec<IN>::N : 1;
    typedef typename RowVec<IN>::vec vec;
    const int lane = lane_id();
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n_wf) return;  // (whole wavefronts: no barrier in this kernel)
    const IN* w = (const IN*)A.wf + row * A.wf_stride + A.wf_offset;
    const int n = A.len;
    Extremes e;
    e.vmin = e.vmax = (float)w[0];  // (sample 0 as every lane's start: a lane that sees no sample never wins -- equal value, larger or equal index)
    e.imin = e.imax = 0;
    e.nan = false;
    // samples [g * 64 * N + lane * N, + N) of group g; four groups requested before the first is looked at
    const int per_group = 64 * N, n_groups = (n + per_group - 1) / per_group;
    if constexpr (VEC) {
        const vec* wv = (const vec*)w;
        for (int g0 = 0; g0 < n_groups; g0 += 4) {
            vec x[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int at = (g0 + k) * per_group + lane * N;
                if (at < n) x[k] = __builtin_nontemporal_load(wv + ((g0 + k) * 64 + lane));  // (a vector that starts inside the row lies inside it whole)
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int at = (g0 + k) * per_group + lane * N;
#pragma unroll
                for (int j = 0; j < N; ++j) take(e, (float)x[k][j], at + j, at < n);
            }
        }
    } else {
        for (int g0 = 0; g0 < n_groups; g0 += 8) {
            IN x[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int at = (g0 + k) * 64 + lane;
                x[k] = at < n ? w[at] : (IN)0;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int at = (g0 + k) * 64 + lane;
                take(e, (float)x[k], at, at < n);
            }
        }
    }
    // across the wavefront: smaller value, then smaller index
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const float ovmin = __shfl_xor(e.vmin, m), ovmax = __shfl_xor(e.vmax, m);
        const int oimin = __shfl_xor(e.imin, m), oimax = __shfl_xor(e.imax, m);
        const bool tmin = ovmin < e.vmin || (ovmin == e.vmin && oimin < e.imin);
        const bool tmax = ovmax > e.vmax || (ovmax == e.vmax && oimax < e.imax);
        e.vmin = tmin ? ovmin : e.vmin;
        e.imin = tmin ? oimin : e.imin;
        e.vmax = tmax ? ovmax : e.vmax;
        e.imax = tmax ? oimax : e.imax;
    }
    const bool any_nan = wave_any(e.nan);
    // time_point_thresh (time_point_thresh.py:12-92) from a sample known by now: 64 consecutive samples per step, away from the start, until a
    // step holds a crossing -- the row was just read, the walk finds it in the caches.  Comparisons only.
    float walked[DSP_REDUCE_WALKS];
#pragma unroll
    for (int k = 0; k < DSP_REDUCE_WALKS; ++k) {
        walked[k] = quiet_nan<float>();
        if (!A.walk_out[k]) continue;  // (uniform)
        const float thr = A.walk_thr[k] ? A.walk_thr[k][row * A.walk_thr_stride[k]] : A.walk_thr_const[k];
        const int ts = A.walk_from[k] == 1 ? e.imin : (A.walk_from[k] == 2 ? e.imax : A.walk_start[k]);
        if (any_nan || thr != thr) continue;
        int found = -1;
        if (A.walk_forward[k]) {  // smallest i in [ts, n - 2] with w[i] <= thr < w[i+1] or w[i] >= thr > w[i+1]
            for (int b = ts; b <= n - 2 && found < 0; b += 64) {
                const int i = b + lane;
                bool hit = false;
                if (i <= n - 2) {
                    const float cur = (float)w[i], nxt = (float)w[i + 1];
                    hit = (cur <= thr && thr < nxt) || (cur >= thr && thr > nxt);
                }
                const unsigned long long m = __ballot(hit);
                if (m) found = b + __builtin_ctzll(m);
            }
        } else {  // largest i in [1, ts] with w[i-1] < thr <= w[i] or w[i-1] > thr >= w[i]
            for (int b = ts; b >= 1 && found < 0; b -= 64) {
                const int i = b - lane;
                bool hit = false;
                if (i >= 1) {
                    const float cur = (float)w[i], prv = (float)w[i - 1];
                    hit = (prv < thr && thr <= cur) || (prv > thr && thr >= cur);
                }
                const unsigned long long m = __ballot(hit);
                if (m) found = b - __builtin_ctzll(m);
            }
        }
        if (found >= 0) walked[k] = (float)found;
    }
    if (lane == 0) {
        const float nanv = quiet_nan<float>();
        // min_max: NaN anywhere -> four NaNs (min_max.py:62-68); numpy.amax of a row with a NaN is NaN
        const float v[5] = {any_nan ? nanv : (float)e.imin, any_nan ? nanv : (float)e.imax, any_nan ? nanv : e.vmin, any_nan ? nanv : e.vmax,
                            any_nan ? nanv : e.vmax};
#pragma unroll
        for (int k = 0; k < 5; ++k)
            if (A.out[k]) ((float*)A.out[k])[row * A.out_stride[k]] = v[k];
#pragma unroll
        for (int k = 0; k < DSP_REDUCE_WALKS; ++k)
            if (A.walk_out[k]) ((float*)A.walk_out[k])[row * A.walk_stride[k]] = walked[k];
#pragma unroll
        for (int k = 0; k < DSP_REDUCE_PICKS; ++k) {
            if (!A.pick_out[k]) continue;
            // a sample at a constant index: fixed_time_pickoff's NaN rule (a NaN anywhere, or a time outside the waveform) or the plain sample
            float s = nanv;
            if (A.pick_at[k] >= 0 && A.pick_at[k] < n && !(A.pick_rule[k] && any_nan)) s = (float)w[A.pick_at[k]];
            ((float*)A.pick_out[k])[row * A.pick_stride[k]] = s;
        }
    }
}

}  // namespace

template <typename IN>
static void launch_reduce(const ReduceArgs* A, int64_t n_wf, int vec, hipStream_t stream) {
    const unsigned blocks = (unsigned)((n_wf + 3) / 4);  // a wavefront per row, four to a workgroup
    if (vec)
        hipLaunchKernelGGL((dsp_reduce_kernel<IN, true>), dim3(blocks), dim3(256), 0, stream, *A, n_wf);
    else
        hipLaunchKernelGGL((dsp_reduce_kernel<IN, false>), dim3(blocks), dim3(256), 0, stream, *A, n_wf);
}

extern "C" int dsp_internal_launch_reduce(const ReduceArgs* A, int64_t n_wf, int dtype, int vec, hipStream_t stream) {
    if (n_wf <= 0) return 0;
    if (dtype == DSP_F32)
        launch_reduce<float>(A, n_wf, vec, stream);
    else if (dtype == DSP_I16)
        launch_reduce<int16_t>(A, n_wf, vec, stream);
    else
        launch_reduce<uint16_t>(A, n_wf, vec, stream);
    return (int)hipGetLastError();
}

extern "C" const char* dsp_internal_reduce_kernel_name() { return "dsp_reduce_kernel"; }
