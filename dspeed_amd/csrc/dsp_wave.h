// dsp_wave.h -- wavefront-level building blocks shared by the device kernels (gfx950, wave64):
// DPP scans, lane shifts, the reference's per-sample step functions.  Internal header.
#pragma once
#include <hip/hip_runtime.h>

#include "dsp_program.h"

namespace {

// ------------------------------------------------------------------------------------------------
// wavefront primitives
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

// Orders this wavefront's LDS traffic for the compiler; the hardware executes one wave's LDS
// instructions in issue order, so no instruction is needed beyond the waitcnt the fence implies.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// lanes without a source get 0: with all rows written the hardware's own zero fill (bound_ctrl) -- no register to clear first --, with a
// row mask the "old" value 0 of the rows that are not written
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ int dpp0(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, ROW_MASK == 0xf);
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp0(float v) {
    return __int_as_float(dpp0<CTRL, ROW_MASK>(__float_as_int(v)));
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp0(double v) {
    int lo = dpp0<CTRL, ROW_MASK>(__double2loint(v));
    int hi = dpp0<CTRL, ROW_MASK>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118;
constexpr int DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143, DPP_WAVE_SHR1 = 0x138;

// inclusive + over the 64 lanes (lanes without a source add the DPP "old" value 0)
__device__ __forceinline__ double wave_scan_add(double v) {
    v += dpp0<DPP_ROW_SHR1>(v);
    v += dpp0<DPP_ROW_SHR2>(v);
    v += dpp0<DPP_ROW_SHR4>(v);
    v += dpp0<DPP_ROW_SHR8>(v);
    v += dpp0<DPP_ROW_BCAST15, 0xa>(v);
    v += dpp0<DPP_ROW_BCAST31, 0xc>(v);
    return v;
}
// value of the previous lane, 0 for lane 0
template <typename V>
__device__ __forceinline__ V wave_prev(V v) {
    return dpp0<DPP_WAVE_SHR1>(v);
}
__device__ __forceinline__ double wave_exscan_add(double v) { return wave_prev(wave_scan_add(v)); }

__device__ __forceinline__ float readlane(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ int readlane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ double readlane(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// value held `sh` lanes below (uniform sh >= 0); lanes without a source get 0
__device__ __forceinline__ double wave_shift_up(double v, int sh) {
    int src = lane_id() - sh;
    int lo = __builtin_amdgcn_ds_bpermute(src << 2, __double2loint(v));
    int hi = __builtin_amdgcn_ds_bpermute(src << 2, __double2hiint(v));
    return src >= 0 ? __hiloint2double(hi, lo) : 0.0;
}
__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = min(v, __shfl_xor(v, m));
    return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = max(v, __shfl_xor(v, m));
    return v;
}
__device__ __forceinline__ bool wave_any(bool p) { return __any(p) != 0; }

__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// LDS element index of sample e of a slot (e / C via a float reciprocal: exact for e < 2^20, C % 8 == 0)
template <typename SlotRef>  // (DevSlot in any address space)
__device__ __forceinline__ int padded_index(const SlotRef& s, int e) {
    int q = (int)(((float)e + 0.5f) * s.invC);
    return s.off + e + q * s.padw;
}

template <typename T>
__device__ __forceinline__ T quiet_nan() {
    return (T)__builtin_nanf("");
}

// ------------------------------------------------------------------------------------------------
// per-sample step of the trapezoidal filters, in the reference's operation order
// (processors/trap_filters.py :62-76 trap_filter, :130-149 trap_norm, :211-227 asym_trap_filter)
// ------------------------------------------------------------------------------------------------
enum { TRAP_FILTER = 0, TRAP_NORM = 1, TRAP_ASYM = 2, TRAP_ASYM_P2 = 3 };

// x / d for a positive integer-valued float64 d (a rise or fall time in samples), correctly rounded like the IEEE division
// the reference performs: q = RN(x * RN(1/d)), one exact residual, one correction (Markstein: with a correctly rounded
// reciprocal and fused multiply-adds the corrected quotient equals RN(x / d) when nothing over- or underflows -- the
// operands here are float32 differences over integers below 2^31).  3 float64 operations instead of the ~30 of a division.
__device__ __forceinline__ double div_by_count(double x, double d, double inv_d) {
    const double q = x * inv_d;
    const double r = __builtin_fma(-q, d, x);
    const double c = __builtin_fma(r, inv_d, q);
    return (__builtin_fabs(q) <= 1.7976931348623157e308) ? c : q;  // +-inf / NaN: the residual is NaN, x * (1/d) already has the result
}
// the same with the special cases (an infinite or NaN x, whose residual is NaN; the sign of a zero) restored by the hardware's own division
// fix-up instead of a compare and two selects: one instruction for three in the trapezoids' per-sample step
__device__ __forceinline__ double div_by_count_fx(double x, double d, double inv_d) {
    const double q = x * inv_d;
    const double r = __builtin_fma(-q, d, x);
    return __builtin_amdgcn_div_fixup(__builtin_fma(r, inv_d, q), d, x);
}

template <typename T, int KIND>
__device__ __forceinline__ T trap_step(T y, T a, T b1, T b2, T b3, double rr, double ll) {
    if (KIND == TRAP_FILTER) {
        return (((y + a) - b1) - b2) + b3;
    } else if (KIND == TRAP_NORM) {
        const T e = ((a - b1) - b2) + b3;
        return (T)((double)y + (double)e / rr);
    } else {
        const T e1 = a - b1, e2 = b2 - b3;
        return (T)(((double)y + (double)e1 / rr) - (double)e2 / ll);
    }
}

// the same step with the reciprocals of rise / fall precomputed once per waveform (1.0 / d is itself a correctly rounded division)
template <typename T, int KIND>
__device__ __forceinline__ T trap_step_r(T y, T a, T b1, T b2, T b3, double rr, double ll, double inv_rr, double inv_ll) {
    if (KIND == TRAP_FILTER) {
        return (((y + a) - b1) - b2) + b3;
    } else if (KIND == TRAP_NORM) {
        const T e = ((a - b1) - b2) + b3;
        return (T)((double)y + div_by_count_fx((double)e, rr, inv_rr));
    } else if (KIND == TRAP_ASYM_P2) {  // rise is a power of two (the usual 128 ns at 16 ns): e1 * (1 / rise) IS the correctly rounded quotient
        const T e1 = a - b1, e2 = b2 - b3;
        return (T)(((double)y + (double)e1 * inv_rr) - div_by_count_fx((double)e2, ll, inv_ll));
    } else {
        const T e1 = a - b1, e2 = b2 - b3;
        return (T)(((double)y + div_by_count_fx((double)e1, rr, inv_rr)) - div_by_count_fx((double)e2, ll, inv_ll));
    }
}

// ------------------------------------------------------------------------------------------------
// fixed_time_pickoff  (processors/fixed_time_pickoff.py:12-125), modes i n f c l h (float64 interpolation weights)
// w4 = samples at i0-1, i0, i0+1, i0+2 (only the in-range ones are used); fatal_code receives a DSP_E_* if the
// reference would raise DSPFatal for this (non-integer) t_in
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T pickoff_eval(T t_in, int mode, int n, const T* w4, int& fatal_code) {  // (inlined: w4 / fatal_code are the caller's registers)
    const int i0 = (int)t_in;
    if ((T)i0 == t_in) return w4[1];
    const double t0 = (double)t_in - (double)i0;
    const double t1 = 1.0 - t0;
    switch (mode) {
        case 'n': return (t0 < 0.5) ? w4[1] : w4[2];
        case 'f': return w4[1];
        case 'c': return w4[2];
        case 'l': return (T)(t1 * (double)w4[1] + t0 * (double)w4[2]);
        case 'h': {
            const double m0 = (i0 == 0) ? (double)(T)(w4[2] - w4[1]) : (double)(T)(w4[2] - w4[0]) / 2.0;
            const double m1 = (i0 == n - 2) ? (double)(T)(w4[2] - w4[1]) : (double)(T)(w4[3] - w4[1]) / 2.0;
            const double t1_2 = t1 * t1, t1_3 = t1 * t1_2, t0_2 = t0 * t0, t0_3 = t0 * t0_2;
            return (T)(((((-2.0 * t1_3 + 3.0 * t1_2) * (double)w4[1] + (-2.0 * t0_3 + 3.0 * t0_2) * (double)w4[2]) - (t1_3 - t1_2) * m0)) +
                       (t0_3 - t0_2) * m1);
        }
        case 's': return w4[1];  // (only reached for integer t_in, handled above: the spline itself is pickoff_spline in dsp_vm.hip)
        case 'i': fatal_code = DSP_E_FTP_INT; return quiet_nan<T>();
        default: fatal_code = DSP_E_FTP_MODE; return quiet_nan<T>();
    }
}

// in-range test of fixed_time_pickoff.py:68-74; returns false when the output must be NaN
template <typename T>
__device__ __forceinline__ bool pickoff_in_range(T t_in, int n) {
    return !(t_in != t_in) && !(t_in < (T)0) && !(t_in > (T)(n - 1));
}

// numpy.floor_divide's float loops (numpy/_core/src/npymath: npy_divmod): the quotient is formed from fmod's exact remainder, so it is the
// floor of the TRUE quotient where floor(a / b) can land one above it (a / b rounding up to an integer)
template <typename T>
__device__ __forceinline__ T np_floor_divide(T a, T b) {
    if (b == (T)0) return a / b;  // (inf or NaN, as the division gives it)
    T mod = sizeof(T) == 8 ? (T)fmod((double)a, (double)b) : (T)fmodf((float)a, (float)b);
    T div = (a - mod) / b;
    if (mod != (T)0 && ((b < (T)0) != (mod < (T)0))) div -= (T)1;
    if (div != (T)0) {
        T fl = floor(div);
        if (div - fl > (T)0.5) fl += (T)1;
        return fl;
    }
    return copysign((T)0, a / b);
}

// ------------------------------------------------------------------------------------------------
// NumPy's integer ufunc loops on values held in the float loop type (DSP_FN_IADD ... DSP_FN_ICAST, dspeed_hip.h): exact 64-bit integer
// arithmetic, then the wrap to the loop's integer type
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int64_t int_loop_value(double a) {
    // (integer variables hold integers; anything else -- a NaN waveform of a processor upstream -- becomes x86-64's "integer indefinite")
    return (a == a && __builtin_fabs(a) < 9223372036854775808.0) ? (int64_t)a : INT64_MIN;
}

__device__ __forceinline__ double int_loop_wrap(int64_t r, int meta) {
    const int bits = DSP_FN_INT_BITS(meta);  // 8, 16, 32; 64 (the float64 chain, waveforms: the host admits the loop only where the operands' types
    if (bits >= 64) return (double)r;        // bound the result below 2^53 -- nothing wraps and every value is a float64, processing_chain.py _wide_wf_loop)
    const uint64_t m = (uint64_t)r & ((1ull << bits) - 1ull);
    int64_t v = (int64_t)m;
    if (DSP_FN_INT_SIGNED(meta) && ((m >> (bits - 1)) & 1ull)) v -= (int64_t)1 << bits;
    return (double)v;
}

template <typename T>
__device__ __forceinline__ T int_loop_apply(int fn, T a_, T b_, int meta) {
    if (fn == DSP_FN_ICAST) {
        const double t = __builtin_trunc((double)a_);
        int64_t r;
        if (DSP_FN_INT_BITS(meta) == 32 && !DSP_FN_INT_SIGNED(meta))
            r = int_loop_value(t);  // (npy_uint)x goes through the 64-bit conversion
        else
            r = (t >= -2147483648.0 && t <= 2147483647.0) ? (int64_t)t : (int64_t)INT32_MIN;  // ... the others through the 32-bit one
        return (T)int_loop_wrap(r, meta);
    }
    const int64_t a = int_loop_value((double)a_), b = int_loop_value((double)b_);
    uint64_t r;
    if (fn == DSP_FN_IADD) r = (uint64_t)a + (uint64_t)b;
    else if (fn == DSP_FN_ISUB) r = (uint64_t)a - (uint64_t)b;
    else if (fn == DSP_FN_IMUL) r = (uint64_t)a * (uint64_t)b;
    else {  // DSP_FN_IFLOORDIV
        if (b == 0) r = 0;
        else if (b == -1) r = 0ull - (uint64_t)a;  // (the type's minimum // -1 wraps back to the minimum, as NumPy's loop returns it)
        else {
            int64_t q = a / b;
            if ((a % b != 0) && ((a < 0) != (b < 0))) q -= 1;
            r = (uint64_t)q;
        }
    }
    return (T)int_loop_wrap((int64_t)r, meta);
}

}  // namespace
