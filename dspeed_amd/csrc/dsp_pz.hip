// dsp_pz.hip -- [bl_subtract ->] pole_zero of whole rows, written back as rows: what a recipe stages in HBM for the kernels that read the
// pole-zero corrected waveform as rows (long FIRs, lane-per-waveform kernels, the program's reductions).
//
// pole_zero (processors/pole_zero.py:24-77) is a prefix sum: with S the inclusive prefix sum of the input, the reference's accumulator after
// sample k is  S(k) - c S(k-1)  (c = exp(-1/tau), float64), rounded to float32 on the way out.  On the waveform VM the row goes through an
// LDS image (lane-contiguous chunks, the recurrence inside a chunk) -- and an 8192-sample image leaves LDS for one wavefront per SIMD: 2.2 ms
// per 131 072 rows where the bytes (16 kB in, 32 kB out) take 1.2.  Here a wavefront walks its row in the order it lies in memory -- 8 samples
// per lane and group of 512 -- with the sums in float64 registers: a local prefix over the lane's 8 samples, one scan across the wavefront per
// group, a carry from group to group; no LDS, eight wavefronts per SIMD, loads and stores of 16 / 32 bytes per lane.
// The sums are the same real numbers in another order than the VM's (float64; the float32 outputs differ in a last place once in a while):
// within the filter bar of the oracle like every scan formulation of this recurrence here.
#include <hip/hip_runtime.h>

#include "dsp_program.h"
#include "dsp_reduce_tail.h"
#include "dsp_wave.h"

#define PZ_GLOBAL __attribute__((address_space(1)))
#define PZ_KARG __attribute__((address_space(4)))

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void pz_report(int* err, int code, int64_t row) {
    if (atomicCAS(&err[0], 0, code) == 0) {
        err[1] = (int)(row & 0xffffffffll);
        err[2] = (int)(row >> 32);
    }
}

// IN: 0 float32, 1 int16, 2 uint16 rows
template <int IN>
__global__ void __launch_bounds__(256) dsp_pz_rows_kernel(PzArgs A_, int64_t n_wf, int* err) {
    const PZ_KARG PzArgs& A = *(const PZ_KARG PzArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    (void)A_;
    const int lane = lane_id();
    const int64_t row = (int64_t)blockIdx.x * 4 + ((int)threadIdx.x >> 6);
    if (row >= n_wf) return;  // (whole wavefronts: no barrier in this kernel)
    constexpr int ESZ = IN == 0 ? 4 : 2, NV = IN == 0 ? 2 : 1;
    const int n = A.len;
    const PZ_GLOBAL char* rowp = (const PZ_GLOBAL char*)A.wf + (row * A.wf_stride + A.wf_offset) * ESZ;
    PZ_GLOBAL float* outp = (PZ_GLOBAL float*)A.out + row * A.out_stride;
    const bool sub = A.sub_mode != 0;
    const float bl = sub ? (A.bl ? ((const PZ_GLOBAL float*)A.bl)[row * A.bl_stride] : A.bl_const) : 0.0f;
    // a time constant per event (pole_zero.py:24-30, the "()" slot filled by a variable): the constant of pole_zero.py:60 formed here in
    // float64, as the interpreter's op forms it
    const float tau_row = A.tau ? ((const PZ_GLOBAL float*)A.tau)[row * A.tau_stride] : 0.0f;
    const double c = A.tau ? exp(-1.0 / (double)tau_row) : A.c;
    const bool tau_nan = A.tau ? (tau_row != tau_row) : (A.tau_nan != 0);
    const int n_groups = (n + 511) / 512;

    // two register sets of two groups each, filled in turn: group g + 4 is asked for while g .. g + 3 are worked on, and no set is ever copied
    // (a copy of a set would have to wait for its loads the moment it is made)
    u4 ra[2][NV], rb[2][NV];
    auto fetch = [&](u4 (&dst)[NV], int g) {
        int at = g * 512 + lane * 8;
        at = at < n ? at : 0;  // (a lane beyond the row's end asks for something inside it and ignores it: no branch around a load, which
                               //  would cost a wait for everything in flight where it joins)
        const PZ_GLOBAL u4* src = (const PZ_GLOBAL u4*)(rowp + (size_t)at * ESZ);
#pragma unroll
#if defined(PZ_DIAG_PLAIN_LOAD)
        for (int v = 0; v < NV; ++v) dst[v] = src[v];
#else
        for (int v = 0; v < NV; ++v) dst[v] = __builtin_nontemporal_load(src + v);
#endif
    };
    fetch(ra[0], 0);
    fetch(ra[1], 1);

    double carry = 0.0;  // S at the end of the previous group
    bool in_nan = false, out_nan = false;
    // min_max of the rows as they are read (A.mm_on: every Ge recipe asks for it of the raw waveform; a kernel of its own read the rows once
    // more): first-occurrence extremes per lane, in index order, one exchange across the wavefront at the end
    const bool mm_on = A.mm_on != 0;
    Extremes e;
    e.vmin = __builtin_inff();
    e.vmax = -__builtin_inff();
    e.imin = e.imax = 0;
    e.nan = false;
    float mx = 0.0f;  // largest |sample written| (a NaN never raises it): the scale of a float16 FIR behind (A.row_scale)
    // the same of the samples [in_lo, in_hi) as they are READ, minus the baseline: the scale of a float16 FIR that filters that slice of
    // these rows (A.in_scale; integer rows)
    const bool in_on = A.in_scale != nullptr;
    const int in_lo = A.in_lo, in_hi = A.in_hi;
    float mx_in = 0.0f;
    auto group = [&](const u4 (&raw)[NV], int g) {
        const int at = g * 512 + lane * 8;
        const bool live = at < n;  // (n is a multiple of 8: a lane's vector lies inside the row whole or not at all)
        float x[8];
        if (IN == 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                x[u] = __uint_as_float(raw[0][u]);
                x[4 + u] = __uint_as_float(raw[NV - 1][u]);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned r = raw[0][u];
                x[2 * u] = IN == 1 ? (float)(short)(r & 0xffffu) : (float)(r & 0xffffu);
                x[2 * u + 1] = IN == 1 ? (float)(short)(r >> 16) : (float)(r >> 16);
            }
        }
        if (mm_on && live) {  // (uniform flag; most groups change no lane's extremes: the largest and smallest of the eight first)
            const float hi8 = fmaxf(fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3])), fmaxf(fmaxf(x[4], x[5]), fmaxf(x[6], x[7])));
            const float lo8 = fminf(fminf(fminf(x[0], x[1]), fminf(x[2], x[3])), fminf(fminf(x[4], x[5]), fminf(x[6], x[7])));
            if (hi8 > e.vmax || lo8 < e.vmin) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const bool lt = x[u] < e.vmin, gt = x[u] > e.vmax;
                    e.vmin = lt ? x[u] : e.vmin;
                    e.imin = lt ? at + u : e.imin;
                    e.vmax = gt ? x[u] : e.vmax;
                    e.imax = gt ? at + u : e.imax;
                }
            }
            if (IN == 0) {
#pragma unroll
                for (int u = 0; u < 8; ++u) e.nan |= x[u] != x[u];  // (fmaxf / fminf skip a NaN: looked for by itself)
            }
        }
        double p[8];
        double run = 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float v = live ? (sub ? x[u] - bl : x[u]) : 0.0f;  // bl_subtract.py:45, in float32
            if (IN == 0) in_nan |= (v != v);                          // (integer samples: only the baseline can be a NaN, looked at below)
            if (in_on && at + u >= in_lo && at + u < in_hi) {          // (x - baseline in float32: what dsp_fir_f16_rows_kernel looks at)
                const float a = __builtin_fabsf(v);
                mx_in = (live && a > mx_in) ? a : mx_in;
            }
            run += (double)v;
            p[u] = run;
        }
        const double inc = wave_scan_add(run);
        const double base = wave_prev(inc) + carry;  // S at the sample in front of this lane's first one
        double before = base;
        f4 y0, y1;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double s = base + p[u];
            const float y = (float)(s - c * before);  // pole_zero.py:69-73: the accumulator after this sample, rounded on the way out
            if (u < 4) y0[u] = y; else y1[u - 4] = y;
            const float a = __builtin_fabsf(y);
            mx = (live && a > mx) ? a : mx;
            before = s;
        }
        // a NaN of the recurrence stays: S is NaN or infinite from there on and every later S - c S' is NaN again -- so the last sample
        // of every lane's eight tells (the row's last sample is one of them)
        out_nan |= live && (y1[3] != y1[3]);
        carry += readlane(inc, 63);
        if (live) {  // plain stores: measured 1.18 ms per 131 072 rows of 8192 against 1.66 with non-temporal ones (0.49 with none)
            PZ_GLOBAL f4* dst = (PZ_GLOBAL f4*)(outp + at);
            dst[0] = y0;
            dst[1] = y1;
        }
    };
    for (int g = 0; g < n_groups; g += 4) {
        fetch(rb[0], g + 2);
        fetch(rb[1], g + 3);
        group(ra[0], g);
        group(ra[1], g + 1);  // (beyond the last group: no live lane, sums of zeros)
        fetch(ra[0], g + 4);
        fetch(ra[1], g + 5);
        if (g + 2 < n_groups) {  // (uniform; no load inside)
            group(rb[0], g + 2);
            group(rb[1], g + 3);
        }
    }
    // ---- what the whole row decides: a NaN anywhere in the input (or a NaN baseline / time constant) makes the waveform NaN
    // (pole_zero.py:55-58); a NaN of the recurrence's own making (inf - inf) is a DSPFatal (:76-77) -- and a NaN waveform here, like the VM's op
    const bool bad_in = wave_any(in_nan) || tau_nan || (sub && bl != bl);
    const bool bad_out = wave_any(out_nan);
    if (bad_in || bad_out) {
        if (!bad_in && lane == 0) pz_report(err, DSP_E_PZ_NAN, row);
        const f4 nanv = {quiet_nan<float>(), quiet_nan<float>(), quiet_nan<float>(), quiet_nan<float>()};
        for (int at = lane * 4; at < n; at += 256) *(PZ_GLOBAL f4*)(outp + at) = nanv;
    }
    if (in_on) {  // (uniform) exactly what dsp_fir_f16_rows_kernel leaves for the slice: same samples, same rule
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) mx_in = fmaxf(mx_in, __shfl_xor(mx_in, sft));
        int e2 = (int)((__float_as_uint(mx_in) >> 23) & 0xffu) - 127;
        const bool bad = !(mx_in <= 3.4028234663852886e38f) || (mx_in > 0.0f && (e2 < -100 || e2 > 100));
        if (!(mx_in > 0.0f) || bad) e2 = 14;
        if (lane == 0) {
            A.in_scale[row] = __uint_as_float((unsigned)(127 + 14 - e2) << 23);
            A.in_flags[row] = (bad ? 1u : 0u) | ((sub && bl != bl) ? 2u : 0u);  // (integer rows: a NaN can only come in through the baseline)
        }
    }
    if (mm_on) {  // min_max.py:62-77: four NaNs for a row with a NaN, else the first occurrence of each extreme
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) {
            const float ovmin = __shfl_xor(e.vmin, sft), ovmax = __shfl_xor(e.vmax, sft);
            const int oimin = __shfl_xor(e.imin, sft), oimax = __shfl_xor(e.imax, sft);
            const bool tmin = ovmin < e.vmin || (ovmin == e.vmin && oimin < e.imin);
            const bool tmax = ovmax > e.vmax || (ovmax == e.vmax && oimax < e.imax);
            e.vmin = tmin ? ovmin : e.vmin;
            e.imin = tmin ? oimin : e.imin;
            e.vmax = tmax ? ovmax : e.vmax;
            e.imax = tmax ? oimax : e.imax;
        }
        const bool any = wave_any(e.nan);
        if (lane == 0) {
            const float nanv = quiet_nan<float>();
            const float v[4] = {any ? nanv : (float)e.imin, any ? nanv : (float)e.imax, any ? nanv : e.vmin, any ? nanv : e.vmax};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (A.mm_out[k]) ((PZ_GLOBAL float*)A.mm_out[k])[row * A.mm_stride[k]] = v[k];
        }
    }
    if (A.row_scale) {  // (uniform) exactly what dsp_fir_f16_rows_kernel leaves for these rows: same samples, same rule
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) mx = fmaxf(mx, __shfl_xor(mx, sft));
        const bool all_nan = bad_in || bad_out;
        if (all_nan) mx = 0.0f;
        int e = (int)((__float_as_uint(mx) >> 23) & 0xffu) - 127;
        const bool extreme = !(mx <= 3.4028234663852886e38f) || (mx > 0.0f && (e < -100 || e > 100));
        if (!(mx > 0.0f) || extreme) e = 14;
        if (lane == 0) {
            A.row_scale[row] = __uint_as_float((unsigned)(127 + 14 - e) << 23);
            A.row_flags[row] = (extreme ? 1u : 0u) | (all_nan ? 2u : 0u);
        }
    }
}

}  // namespace

extern "C" int dsp_internal_launch_pz_rows(const PzArgs* A, int64_t n_wf, int* err, hipStream_t stream) {
    if (n_wf <= 0) return 0;
    const unsigned blocks = (unsigned)((n_wf + 3) / 4);  // a wavefront per row, four to a workgroup
    switch (A->in_kind) {
        case 0: hipLaunchKernelGGL(dsp_pz_rows_kernel<0>, dim3(blocks), dim3(256), 0, stream, *A, n_wf, err); break;
        case 1: hipLaunchKernelGGL(dsp_pz_rows_kernel<1>, dim3(blocks), dim3(256), 0, stream, *A, n_wf, err); break;
        default: hipLaunchKernelGGL(dsp_pz_rows_kernel<2>, dim3(blocks), dim3(256), 0, stream, *A, n_wf, err); break;
    }
    return (int)hipGetLastError();
}

extern "C" const char* dsp_internal_pz_rows_kernel_name() { return "dsp_pz_rows_kernel"; }
