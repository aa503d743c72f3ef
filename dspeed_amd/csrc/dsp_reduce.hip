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
#include "dsp_reduce_tail.h"
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

// VEC: row starts, stride and length are whole 16-byte vectors (every Ge waveform is); otherwise a sample per lane and load
template <typename IN, bool VEC>
__global__ void __launch_bounds__(256) dsp_reduce_kernel(ReduceArgs A, int64_t n_wf, int* err) {
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
    if (!A.need_stream) {
        // walks only, on rows that are NaN from the first sample on or NaN-free (what pole_zero writes): nothing reads the whole row
        e.nan = (float)w[0] != (float)w[0];
    } else if constexpr (VEC) {
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
    reduce_finish<IN>(A, row, e, w, n, lane, err);
}

}  // namespace

template <typename IN>
static void launch_reduce(const ReduceArgs* A, int64_t n_wf, int vec, int* err, hipStream_t stream) {
    const unsigned blocks = (unsigned)((n_wf + 3) / 4);  // a wavefront per row, four to a workgroup
    if (vec)
        hipLaunchKernelGGL((dsp_reduce_kernel<IN, true>), dim3(blocks), dim3(256), 0, stream, *A, n_wf, err);
    else
        hipLaunchKernelGGL((dsp_reduce_kernel<IN, false>), dim3(blocks), dim3(256), 0, stream, *A, n_wf, err);
}

extern "C" int dsp_internal_launch_reduce(const ReduceArgs* A, int64_t n_wf, int dtype, int vec, int* err, hipStream_t stream) {
    if (n_wf <= 0) return 0;
    if (dtype == DSP_F32)
        launch_reduce<float>(A, n_wf, vec, err, stream);
    else if (dtype == DSP_I16)
        launch_reduce<int16_t>(A, n_wf, vec, err, stream);
    else
        launch_reduce<uint16_t>(A, n_wf, vec, err, stream);
    return (int)hipGetLastError();
}

extern "C" const char* dsp_internal_reduce_kernel_name() { return "dsp_reduce_kernel"; }
