// dsp_vm.hip -- the waveform VM: one wavefront per waveform, all intermediates in LDS.
//
// Replaces the reference's ProcessingChain inner loop (processing_chain.py:665-673, 1144-1163) and the
// numba gufunc bodies it calls (processors/*.py, cited per op below) for one batch of rows.
//
// Execution model (gfx950 / CDNA4, wave64):
//   * a workgroup is `waves_per_block` independent wavefronts; wavefront w processes rows
//     w, w + total_waves, ... ; there is no inter-wave communication, hence no s_barrier anywhere;
//   * a waveform variable ("slot") lives in LDS; lane j owns the contiguous chunk [j*C, (j+1)*C) of it,
//     stored at pitch C+1 so that 64 lanes touching the same offset of their chunks hit 64 banks;
//   * recursions (pole-zero, trapezoids, IIR) are evaluated chunk-serially per lane with the carry
//     between chunks supplied by a 6-step DPP wavefront scan (row_shr 1/2/4/8, row_bcast 15/31);
//   * the kernel is memory/LDS bound: no MFMA anywhere.
//
// Arithmetic contract: every op reproduces the numba typing of the reference body:
// float32 (op) float32 stays float32; anything numba promotes is float64 here and is rounded to
// float32 exactly where the reference stores into a float32 array.  Compiled with -ffp-contract=off.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "dsp_program.h"
#include "dsp_wave.h"

// pointers into device memory (the program, I/O buffers) carry their address space in the type, see Ctx
#define DSP_GLOBAL __attribute__((address_space(1)))
// The device program (ops, slot and binding descriptors) is read-only for the kernel: read through the constant address space its uniform
// loads are scalar loads (SMEM, the scalar cache) instead of vector loads that every op start had to wait for with vmcnt(0).
#define DSP_PROG __attribute__((address_space(4)))

namespace {

// ------------------------------------------------------------------------------------------------
// per-wave context
// ------------------------------------------------------------------------------------------------
template <typename T>
struct Ctx {
    // LDS pointers carry their address space in the type: an op that the compiler decides not to inline into the interpreter would
    // otherwise see a generic pointer and access LDS through flat_load / flat_store (measured: the FIR op 5x slower)
    typedef __attribute__((address_space(3))) T LT;
    LT* lds;                 // this wavefront's LDS region
    const DSP_PROG DevProgram* prog;  // device copy of the program
    // I/O device pointers of this launch: read straight from the kernel-argument segment (constant address space, scalar loads with a
    // run-time index); taking the address of the by-value IoPtrs argument instead makes the compiler copy it to scratch
    const __attribute__((address_space(4))) uint64_t* kptrs;
    // (global address space spelled out for the same reason as LDS above: global_load / global_store, not flat)
    __device__ __forceinline__ uint64_t io_addr(int i) const { return kptrs[i]; }
    template <typename U>
    __device__ __forceinline__ DSP_GLOBAL U* io_ptr(int i) const {
        return (DSP_GLOBAL U*)kptrs[i];
    }
    int64_t row;             // waveform being processed
    int* err;                // device error word
    // nan_all bit s: slot s is "all NaN" (the reference's NaN-propagation state; its content is then not maintained);
    // nan_some bit s: slot s holds real content with NaN samples in it (windower output reaching past the input) -- consumers
    // treat it like all-NaN (np.isnan(w_in).any()), a store writes the content
    uint32_t nan_all, nan_some;

    __device__ __forceinline__ LT* chunk(const DSP_PROG DevSlot& s) const { return lds + s.off + lane_id() * s.pitch; }
    __device__ __forceinline__ LT* sregs() const { return lds + prog->sreg_off; }
    __device__ __forceinline__ bool slot_nan(int s) const { return ((nan_all | nan_some) >> s) & 1u; }
    __device__ __forceinline__ bool slot_all_nan(int s) const { return (nan_all >> s) & 1u; }
    __device__ __forceinline__ void set_nan(int s, bool v) {
        nan_all &= ~(1u << s);
        nan_some &= ~(1u << s);
        if (v) nan_all |= 1u << s;
    }
    __device__ __forceinline__ void set_some_nan(int s) {
        nan_all &= ~(1u << s);
        nan_some |= 1u << s;
    }
    __device__ __forceinline__ void fatal(int code) const {
        if (lane_id() == 0 && atomicCAS(&err[0], 0, code) == 0) {
            err[1] = (int)(row & 0xffffffffll);
            err[2] = (int)(row >> 32);
        }
    }
    // scalar operand: constant, per-waveform input column, or scalar register
    __device__ __forceinline__ T scalar(const DSP_PROG dsp_scalar_arg& a) const { return make_uniform(scalar_raw(a)); }
    static __device__ __forceinline__ float make_uniform(float v) {
        return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
    }
    static __device__ __forceinline__ double make_uniform(double v) {
        return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
    }
    __device__ __forceinline__ T scalar_raw(const DSP_PROG dsp_scalar_arg& a) const {
        if (a.kind == DSP_ARG_CONST) return (T)a.value;
        if (a.kind == DSP_ARG_REG) return sregs()[a.index];
        const DSP_PROG DevIO& io = prog->io[a.index];
        const int64_t at = (int64_t)io.offset + row * io.row_stride;
        if (io.dtype == DSP_F32) return (T)io_ptr<const float>(a.index)[at];
        if (io.dtype == DSP_F64) return (T)io_ptr<const double>(a.index)[at];
        if (io.dtype == DSP_I32) return (T)io_ptr<const int32_t>(a.index)[at];
        if (io.dtype == DSP_I16) return (T)io_ptr<const int16_t>(a.index)[at];
        if (io.dtype == DSP_U16) return (T)io_ptr<const uint16_t>(a.index)[at];
        if (io.dtype == DSP_BOOL) return (T)(io_ptr<const uint8_t>(a.index)[at] != 0);
        if (io.dtype == DSP_I64) return (T)io_ptr<const int64_t>(a.index)[at];  // (the value converted, as NumPy's cast into a float loop does)
        if (io.dtype == DSP_U64) return (T)io_ptr<const uint64_t>(a.index)[at];
        return (T)io_ptr<const uint32_t>(a.index)[at];
    }
};

// ------------------------------------------------------------------------------------------------
// LOAD / STORE: coalesced 16-byte global accesses <-> chunked LDS layout
// ------------------------------------------------------------------------------------------------
// TEAM == 2: the two wavefronts of a row's team load alternate batches (`member` 0 / 1) -- half the loads and half the LDS writes each; the
// caller's barrier makes the image whole for both
template <typename T, typename InT, int TEAM = 1>
__device__ __forceinline__ bool load_slot(Ctx<T>& cx, const DSP_PROG DevSlot& s, const DSP_GLOBAL InT* __restrict__ g, int len, bool vec_ok, bool sub,
                                          T bsub, int member = 0) {
    constexpr int V = 16 / (int)sizeof(InT);
    typedef InT vec_t __attribute__((ext_vector_type(V)));
    const int total = 64 * s.C;
    bool nan = false;
    if (vec_ok) {
        // B loads in flight per lane before the LDS writes.  A batch that lies inside the row whole (a uniform test) is straight-line
        // code: with a bounds test in front of every load the compiler waits for each load before the next branch and the loads go to
        // HBM one at a time (C2 on the VM: 121 -> 128 M waveforms/s).
        constexpr int B = 8;
        auto put = [&](const vec_t& vv, int e, bool inside = false) {
            const int a = padded_index(s, e);
#pragma unroll
            for (int m = 0; m < V; ++m) {
                T x = (T)vv[m];
                x = (sub && (inside || e + m < len)) ? x - bsub : x;  // (a BL_SUBTRACT folded into the load; the zero fill past the end stays zero)
                nan |= (x != x);
                cx.lds[a + m] = x;
            }
        };
        for (int base = (TEAM > 1 ? member : 0) * 64 * V * B; base < total; base += TEAM * 64 * V * B) {
            vec_t v[B];
            const int e0 = base + lane_id() * V;
            if (base + 64 * V * B <= len) {
#pragma unroll
                for (int b = 0; b < B; ++b) v[b] = *(const DSP_GLOBAL vec_t*)(g + e0 + b * 64 * V);
#pragma unroll
                for (int b = 0; b < B; ++b) put(v[b], e0 + b * 64 * V, true);
            } else {  // the batch that holds the end of the row (and the zero fill of the last chunk)
#pragma unroll
                for (int b = 0; b < B; ++b) {
                    const int e = e0 + b * 64 * V;
                    if (e + V <= len) {
                        v[b] = *(const DSP_GLOBAL vec_t*)(g + e);
                    } else {
#pragma unroll
                        for (int m = 0; m < V; ++m) v[b][m] = (e + m < len) ? g[e + m] : (InT)0;
                    }
                }
#pragma unroll
                for (int b = 0; b < B; ++b) {
                    const int e = e0 + b * 64 * V;
                    if (e < total) put(v[b], e);
                }
            }
        }
    } else {
        for (int e = lane_id() + (TEAM > 1 ? member : 0) * 64; e < total; e += TEAM * 64) {
            T x = (e < len) ? (T)g[e] : (T)0;
            x = (sub && e < len) ? x - bsub : x;
            nan |= (x != x);
            cx.lds[padded_index(s, e)] = x;
        }
    }
    return nan;
}

template <typename T, int TEAM = 1>
__device__ __forceinline__ void op_load(Ctx<T>& cx, const DSP_PROG DevOp& op, int member = 0) {
    const DSP_PROG DevSlot& s = cx.prog->slots[op.dst];
    const DSP_PROG DevIO& io = cx.prog->io[op.io];
    const int64_t at = cx.row * io.row_stride + io.offset;
    const bool vec_ok = io.vec_ok && ((cx.io_addr(op.io) & 15u) == 0);
    const bool sub = op.ic[0] != 0;  // the BL_SUBTRACT behind this load, folded in by dsp_chain_create
    const T bsub = sub ? cx.scalar(op.sp[0]) : (T)0;
    bool nan;
    switch (io.dtype) {
        case DSP_F32: nan = load_slot<T, float, TEAM>(cx, s, cx.template io_ptr<const float>(op.io) + at, io.len, vec_ok, sub, bsub, member); break;
        case DSP_I16: nan = load_slot<T, int16_t, TEAM>(cx, s, cx.template io_ptr<const int16_t>(op.io) + at, io.len, vec_ok, sub, bsub, member); break;
        case DSP_U16: nan = load_slot<T, uint16_t, TEAM>(cx, s, cx.template io_ptr<const uint16_t>(op.io) + at, io.len, vec_ok, sub, bsub, member); break;
        case DSP_I32: nan = load_slot<T, int32_t, TEAM>(cx, s, cx.template io_ptr<const int32_t>(op.io) + at, io.len, vec_ok, sub, bsub, member); break;
        case DSP_U32: nan = load_slot<T, uint32_t, TEAM>(cx, s, cx.template io_ptr<const uint32_t>(op.io) + at, io.len, vec_ok, sub, bsub, member); break;
        default: nan = load_slot<T, double, TEAM>(cx, s, cx.template io_ptr<const double>(op.io) + at, io.len, vec_ok, sub, bsub, member); break;
    }
    if (op.ip[0] > 0 || op.ip[1] > 0) {  // the slice of a longer waveform, first read by a processor whose NaN rule covers all of it
        for (int part = 0; part < 2; ++part) {
            const int cnt = op.ip[part];
            const int64_t first = part == 0 ? at - cnt : at + io.len;
            for (int e = lane_id(); e < cnt; e += 64) {
                switch (io.dtype) {
                    case DSP_F32: { const float x = cx.template io_ptr<const float>(op.io)[first + e]; nan |= (x != x); break; }
                    case DSP_F64: { const double x = cx.template io_ptr<const double>(op.io)[first + e]; nan |= (x != x); break; }
                    default: break;  // (integer rows hold no NaN)
                }
            }
        }
    }
    bool any_nan = wave_any(nan);
    if (TEAM > 1) {
        // each member saw half of the row: what they found meets in two words of the op scratch area, between two barriers -- the first also
        // makes the image whole for both, the second keeps the words until both have read them
        auto* sc = cx.lds + cx.prog->scratch_off;
        if (lane_id() == 0) sc[member] = any_nan ? (T)1 : (T)0;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        any_nan = false;
#pragma unroll
        for (int m = 0; m < TEAM; ++m) any_nan |= sc[m] != (T)0;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (sub)  // bl_subtract.py:41-44: a NaN anywhere (or a NaN baseline, which made every sample NaN) is a NaN waveform
        cx.set_nan(op.dst, any_nan);
    else if (any_nan)  // (the samples are all there: what reads the slot as a whole sees a NaN waveform, a store writes it as it is)
        cx.set_some_nan(op.dst);
    else
        cx.set_nan(op.dst, false);
    wave_sync();
}

template <typename T>
__device__ __forceinline__ void op_store(Ctx<T>& cx, const DSP_PROG DevOp& op) {
    const DSP_PROG DevSlot& s = cx.prog->slots[op.src];
    const DSP_PROG DevIO& io = cx.prog->io[op.io];
    DSP_GLOBAL T* g = cx.template io_ptr<T>(op.io) + cx.row * io.row_stride + io.offset;
    const int len = io.len;
    const bool nan = cx.slot_all_nan(op.src);  // (a slot with some NaN samples is stored as it is)
    if (io.dtype == DSP_BOOL) {  // truth values: one byte each (NaN is true, like ndarray.astype(bool))
        DSP_GLOBAL uint8_t* gb = cx.template io_ptr<uint8_t>(op.io) + cx.row * io.row_stride + io.offset;
        for (int e = lane_id(); e < len; e += 64) gb[e] = (nan || cx.lds[padded_index(s, e)] != (T)0) ? 1 : 0;
        return;
    }
    constexpr int V = 16 / (int)sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(V)));
    if (io.vec_ok && ((cx.io_addr(op.io) & 15u) == 0)) {
        for (int e = lane_id() * V; e < len; e += 64 * V) {
            const int a = padded_index(s, e);
            vec_t v;
#pragma unroll
            for (int m = 0; m < V; ++m) v[m] = nan ? quiet_nan<T>() : cx.lds[a + m];
            if (e + V <= len) {
                *(DSP_GLOBAL vec_t*)(g + e) = v;
            } else {
                for (int m = 0; m < V && e + m < len; ++m) g[e + m] = v[m];
            }
        }
    } else {
        for (int e = lane_id(); e < len; e += 64) g[e] = nan ? quiet_nan<T>() : cx.lds[padded_index(s, e)];
    }
}

template <typename T>
__device__ __forceinline__ void op_store_scalar(Ctx<T>& cx, const DSP_PROG DevOp& op) {
    const DSP_PROG DevIO& io = cx.prog->io[op.io];
    if (io.dtype == DSP_BOOL) {
        if (lane_id() == 0) cx.template io_ptr<uint8_t>(op.io)[cx.row * io.row_stride + io.offset] = cx.sregs()[op.ip[0]] != (T)0 ? 1 : 0;
        return;
    }
    if (lane_id() == 0) cx.template io_ptr<T>(op.io)[cx.row * io.row_stride + io.offset] = cx.sregs()[op.ip[0]];
}

// host-made (dsp_chain_create): a run of STORE_SCALARs as one op, lane j stores register ic[j] >> 16 to binding ic[j] & 0xffff
template <typename T>
__device__ __forceinline__ void op_store_scalars(Ctx<T>& cx, const DSP_PROG DevOp& op) {
    const int lane = lane_id();
    if (lane < op.dst) {
        int pair = op.ic[0];
#pragma unroll
        for (int j = 1; j < DSP_IC; ++j) pair = lane == j ? op.ic[j] : pair;  // (the op is read with scalar loads: a select per entry)
        const int k = pair & 0xffff, reg = pair >> 16;
        // this lane's binding: descriptor and pointer with the lane's own index (vector loads from the constant address space)
        const DSP_PROG DevIO& io = cx.prog->io[k];
        const int64_t at = cx.row * io.row_stride + io.offset;
        const T v = cx.sregs()[reg];
        if (io.dtype == DSP_BOOL)
            cx.template io_ptr<uint8_t>(k)[at] = v != (T)0 ? 1 : 0;
        else
            cx.template io_ptr<T>(k)[at] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// NumPy ufuncs the recipe language adds as processors (processing_chain.py:832-947, 1266-1420): one IEEE operation per sample in the
// loop type, truth values as 0 / 1
// ------------------------------------------------------------------------------------------------
template <int FN, typename T>
__device__ __forceinline__ T ew_apply(T a, T b, T c, int meta) {
    if constexpr (FN >= DSP_FN_IADD && FN <= DSP_FN_ICAST) return int_loop_apply<T>(FN, a, b, meta);  // (the integer loops: dsp_wave.h)
    else if constexpr (FN == DSP_FN_ADD) return a + b;
    else if constexpr (FN == DSP_FN_SUB) return a - b;
    else if constexpr (FN == DSP_FN_MUL) return a * b;
    else if constexpr (FN == DSP_FN_DIV) return a / b;
    else if constexpr (FN == DSP_FN_LT) return (T)(a < b);
    else if constexpr (FN == DSP_FN_LE) return (T)(a <= b);
    else if constexpr (FN == DSP_FN_GT) return (T)(a > b);
    else if constexpr (FN == DSP_FN_GE) return (T)(a >= b);
    else if constexpr (FN == DSP_FN_EQ) return (T)(a == b);
    else if constexpr (FN == DSP_FN_NE) return (T)(a != b);
    else if constexpr (FN == DSP_FN_WHERE) return a != (T)0 ? b : c;
    else if constexpr (FN == DSP_FN_ISNAN) return (T)(a != a);
    else if constexpr (FN == DSP_FN_ISFINITE) return (T)((a - a) == (T)0);
    else if constexpr (FN == DSP_FN_NEG) return -a;
    else if constexpr (FN == DSP_FN_FLOORDIV) return np_floor_divide<T>(a, b);
    else if constexpr (FN == DSP_FN_RINT) return rint(a);
    else if constexpr (FN == DSP_FN_FLOOR) return floor(a);
    else if constexpr (FN == DSP_FN_CEIL) return ceil(a);
    else if constexpr (FN == DSP_FN_TRUNC) return trunc(a);
    else if constexpr (FN == DSP_FN_LOR) return (T)(a != (T)0 || b != (T)0);
    else if constexpr (FN == DSP_FN_LAND) return (T)(a != (T)0 && b != (T)0);
    else return a;
}

template <typename T, typename F>
__device__ __forceinline__ void ew_dispatch(int fn, F&& f) {
    switch (DSP_FN_CODE(fn)) {
        case DSP_FN_IADD: f(std::integral_constant<int, DSP_FN_IADD>()); break;
        case DSP_FN_ISUB: f(std::integral_constant<int, DSP_FN_ISUB>()); break;
        case DSP_FN_IMUL: f(std::integral_constant<int, DSP_FN_IMUL>()); break;
        case DSP_FN_IFLOORDIV: f(std::integral_constant<int, DSP_FN_IFLOORDIV>()); break;
        case DSP_FN_ICAST: f(std::integral_constant<int, DSP_FN_ICAST>()); break;
        case DSP_FN_ADD: f(std::integral_constant<int, DSP_FN_ADD>()); break;
        case DSP_FN_SUB: f(std::integral_constant<int, DSP_FN_SUB>()); break;
        case DSP_FN_MUL: f(std::integral_constant<int, DSP_FN_MUL>()); break;
        case DSP_FN_DIV: f(std::integral_constant<int, DSP_FN_DIV>()); break;
        case DSP_FN_LT: f(std::integral_constant<int, DSP_FN_LT>()); break;
        case DSP_FN_LE: f(std::integral_constant<int, DSP_FN_LE>()); break;
        case DSP_FN_GT: f(std::integral_constant<int, DSP_FN_GT>()); break;
        case DSP_FN_GE: f(std::integral_constant<int, DSP_FN_GE>()); break;
        case DSP_FN_EQ: f(std::integral_constant<int, DSP_FN_EQ>()); break;
        case DSP_FN_NE: f(std::integral_constant<int, DSP_FN_NE>()); break;
        case DSP_FN_WHERE: f(std::integral_constant<int, DSP_FN_WHERE>()); break;
        case DSP_FN_ISNAN: f(std::integral_constant<int, DSP_FN_ISNAN>()); break;
        case DSP_FN_ISFINITE: f(std::integral_constant<int, DSP_FN_ISFINITE>()); break;
        case DSP_FN_NEG: f(std::integral_constant<int, DSP_FN_NEG>()); break;
        case DSP_FN_FLOORDIV: f(std::integral_constant<int, DSP_FN_FLOORDIV>()); break;
        case DSP_FN_LOR: f(std::integral_constant<int, DSP_FN_LOR>()); break;
        case DSP_FN_LAND: f(std::integral_constant<int, DSP_FN_LAND>()); break;
        case DSP_FN_RINT: f(std::integral_constant<int, DSP_FN_RINT>()); break;
        case DSP_FN_FLOOR: f(std::integral_constant<int, DSP_FN_FLOOR>()); break;
        case DSP_FN_CEIL: f(std::integral_constant<int, DSP_FN_CEIL>()); break;
        case DSP_FN_TRUNC: f(std::integral_constant<int, DSP_FN_TRUNC>()); break;
        default: f(std::integral_constant<int, DSP_FN_COPY>()); break;
    }
}

template <typename T>
__device__ __forceinline__ void op_elementwise(Ctx<T>& cx, const DSP_PROG DevOp& op) {
    const DSP_PROG DevSlot& sd = cx.prog->slots[op.dst];
    const int sl[3] = {op.src, op.ip[1], op.ip[2]};
    const typename Ctx<T>::LT* ps[3];
    T k[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        // an all-NaN waveform (the NaN rule of a processor upstream) has no content: every sample reads NaN; a per-event operand is the
        // same number for every sample
        const bool slot = sl[j] >= 0;
        k[j] = slot ? quiet_nan<T>() : cx.scalar(op.sp[j]);
        ps[j] = (slot && !cx.slot_all_nan(sl[j])) ? cx.chunk(cx.prog->slots[sl[j]]) : nullptr;
    }
    auto* pd = cx.chunk(sd);
    const int first = lane_id() * sd.C;
    bool nan = false;
    ew_dispatch<T>(op.ip[0], [&](auto fn) {
        // (the loads of four samples ahead of their four stores: the result may take the place of an operand, so the compiler keeps every
        // load behind the store before it; C is a multiple of 4)
        for (int t0 = 0; t0 < sd.C; t0 += 4) {
            T a[4], b[4], c[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[j] = ps[0] ? ps[0][t0 + j] : k[0];
                b[j] = ps[1] ? ps[1][t0 + j] : k[1];
                c[j] = ps[2] ? ps[2][t0 + j] : k[2];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                T v = ew_apply<decltype(fn)::value, T>(a[j], b[j], c[j], op.ip[0]);
                if (first + t0 + j >= sd.len) v = (T)0;  // (beyond the waveform: kept finite)
                nan |= (v != v);
                pd[t0 + j] = v;
            }
        }
    });
    if (wave_any(nan))
        cx.set_some_nan(op.dst);  // (sample by sample: NaN samples stay single samples)
    else
        cx.set_nan(op.dst, false);
    wave_sync();
}

template <typename T>
__device__ __forceinline__ void op_scalar_func(Ctx<T>& cx, const DSP_PROG DevOp& op) {
    const T a = cx.scalar(op.sp[0]), b = cx.scalar(op.sp[1]), c = cx.scalar(op.sp[2]);
    T v = (T)0;
    ew_dispatch<T>(op.ip[0], [&](auto fn) { v = ew_apply<decltype(fn)::value, T>(a, b, c, op.ip[0]); });
    if (lane_id() == 0) cx.sregs()[op.dst] = v;
    wave_sync();
}

// ------------------------------------------------------------------------------------------------
// bl_subtract  (processors/bl_subtract.py:11-46):  w_out = w_in - a_baseline, both T
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void op_bl_subtract(Ctx<T>& cx, const DSP_PROG DevOp& op) {
    const DSP_PROG DevSlot& ss = cx.prog->slots[op.src];
    const DSP_PROG DevSlot& sd = cx.prog->slots[op.dst];
    const T b = cx.scalar(op.sp[0]);
    // ip[0] = 1: numpy.subtract(w, scalar) -- the same subtraction sample by sample, but NaN samples stay where they are instead of
    // making the whole waveform NaN (bl_subtract.py:41-44 checks np.isnan(w_in).any(); the ufunc does not)
    const bool elementwise = op.ip[0] == 1;
    if ((elementwise ? cx.slot_all_nan(op.src) : cx.slot_nan(op.src)) || b != b) {
        cx.set_nan(op.dst, true);
        return;
    }
    const auto* ps = cx.chunk(ss);
    auto* pd = cx.chunk(sd);
    bool nan = false;
    // (eight loads, then eight stores: source and target may be the same slot, so the compiler orders every load behind the store before
    // it and each sample would wait out a full LDS round trip; C is a multiple of 8)
    for (int t0 = 0; t0 < ss.C; t0 += 8) {
        T v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = ps[t0 + k];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            v[k] = v[k] - b;
            nan |= (v[k] != v[k]);
            pd[t0 + k] = v[k];
        }
    }
    if (elementwise && wave_any(nan))
        cx.set_some_nan(op.dst);
    else
        cx.set_nan(op.dst, wave_any(nan));
    wave_sync();
}

// ------------------------------------------------------------------------------------------------
// min_max_norm  (processors/min_max.py:85-140):  w_out = w_in / max(|a_min|, |a_max|), w_in itself if either bound is 0
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void op_min_max_norm(Ctx<T>& cx, const DSP_PROG DevOp& op) {
    const DSP_PROG DevSlot& ss = cx.prog->slots[op.src];
    const DSP_PROG DevSlot& sd = cx.prog->slots[op.dst];
    const T lo = cx.scalar(op.sp[0]), hi = cx.scalar(op.sp[1]);
    const T amin = lo < (T)0 ? -lo : lo, amax = hi < (T)0 ? -hi : hi;
    const bool copy = amax == (T)0 || amin == (T)0;
    // (neither comparison holds with a NaN bound: the output stays NaN, like the reference's if / elif chain)
    if (cx.slot_nan(op.src) || (!copy && !(amax >= amin) && !(amax < amin))) {
        cx.set_nan(op.dst, true);
        return;
    }
    const T d = amax >= amin ? amax : amin;
    const auto* ps = cx.chunk(ss);
    auto* pd = cx.chunk(sd);
    bool nan = false;
#pragma unroll 8
    for (int t = 0; t < ss.C; ++t) {
        const T v = copy ? (T)ps[t] : (T)ps[t] / d;
        nan |= (v != v);
        pd[t] = v;
    }
    cx.set_nan(op.dst, wave_any(nan));
    wave_sync();
}

// ------------------------------------------------------------------------------------------------
// pole_zero  (processors/pole_zero.py:24-77)
//   y[0] = x[0];  acc_i = (acc_{i-1} + x[i]) - x[i-1]*c  in float64, y[i] = (T)acc_i,  c = exp(-1/tau) (float64, host libm).
// Parallel form: acc at the end of sample k is  S(k) - c*S(k-1)  with S = inclusive prefix sum of x, so the carry into
// lane j's chunk is  E_j - c*(E_j - x[jC-1])  with E_j the exclusive scan of the per-chunk sums; inside the chunk the
// recurrence runs with the reference's operation order.  fc[0] = c, ic[0] = tau is NaN.
// ------------------------------------------------------------------------------------------------
// exp(-1 / tau) in float64 for a time constant that varies per event: out of line, so that the 200 instructions of the device's exp do
// not sit in the middle of the op every recipe runs with a constant tau
__device__ __attribute__((noinline)) double pz_decay(double tau) { return exp(-1.0 / tau); }

template <typename T>
__device__ __forceinline__ void op_pole_zero(Ctx<T>& cx, const DSP_PROG DevOp& op) {
    const DSP_PROG DevSlot& ss = cx.prog->slots[op.src];
    const DSP_PROG DevSlot& sd = cx.prog->slots[op.dst];
    double c = op.fc[0];
    bool tau_nan = op.ic[0] != 0;
    if (op.ic[1]) {  // one time constant per event (the gufunc's "()" slot filled by a per-event variable, pole_zero.py:24-30): the
        // constant of pole_zero.py:60 is formed here, in float64 like numba forms it (the device's exp may differ from libm's in the last
        // bit: 1e-16 on the constant, far below the float32 output)
        const T tau = cx.scalar(op.sp[0]);
        tau_nan = tau != tau;
        c = pz_decay((double)tau);
    }
    if (cx.slot_nan(op.src) || tau_nan) {
        cx.set_nan(op.dst, true);
        return;
    }
    const auto* ps = cx.chunk(ss);
    auto* pd = cx.chunk(sd);
    const int C = ss.C;
    double X = 0.0;
#pragma unroll 8
    for (int t = 0; t < C; ++t) X += (double)ps[t];
    const double E = wave_exscan_add(X);
    const T xlast = ps[C - 1];
    const double xprev = (double)wave_prev(xlast);
    double acc = E - c * (E - xprev);
    double xp = xprev;
    bool nan = false;
    for (int t0 = 0; t0 < C; t0 += 8) {  // (loads of a batch ahead of its stores, as in bl_subtract: the filter usually runs in place)
        T xs[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) xs[k] = ps[t0 + k];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const double x = (double)xs[k];
            acc = (acc + x) - xp * c;
            const T y = (T)acc;
            nan |= (y != y);
            pd[t0 + k] = y;
            xp = x;
        }
    }
    if (wave_any(nan)) {  // pole_zero.py:76-77: NaN out of non-NaN input (inf - inf)
        cx.fatal(DSP_E_PZ_NAN);
        cx.set_nan(op.dst, true);
    } else {
        cx.set_nan(op.dst, false);
    }
    wave_sync();
}

// ------------------------------------------------------------------------------------------------
// double_pole_zero  (processors/pole_zero.py:82-198): second-order IIR, float64 state
//   y[n] = x[n] + n1 x[n-1] + n2 x[n-2] - d1 y[n-1] - d2 y[n-2],   y[0] = x[0], y[1] = x[1]
// State s = (y[n-1], y[n-2]) advances as s' = M s + (v, 0), M = [[-d1, -d2], [1, 0]].  Each lane first runs its chunk from a
// zero state (lane 0 from the true initial state) to get the chunk's forced response r_j; the carry into chunk j is
// sum_{k<j} M^{C(j-1-k)} r_k, a Hillis-Steele scan with the uniform matrices M^{C*2^d} (host, float64).
// fc: 0 n1, 1 n2, 2 d1, 3 d2, 4.. six 2x2 matrices M^{C}, M^{2C}, ... M^{32C} (row major); ic[0] = parameter NaN.
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void op_double_pole_zero(Ctx<T>& cx, const DSP_PROG DevOp& op) {
    const DSP_PROG DevSlot& ss = cx.prog->slots[op.src];
    const DSP_PROG DevSlot& sd = cx.prog->slots[op.dst];
    double n1 = op.fc[0], n2 = op.fc[1], d1 = op.fc[2], d2 = op.fc[3];
    bool par_nan = op.ic[0] != 0;
    const int C = ss.C, lane = lane_id();
    // powers of the recursion's linear part for the scan over the lanes: M^(C 2^d), d = 0..5; precomputed on the host for constant
    // parameters (read from the op where the scan uses them), formed here when a time constant or the fraction varies per event -- every
    // lane the same 2 x 2 products in float64 -- and parked in the LDS scratch area: 24 doubles held in registers across the passes
    // below cost the whole interpreter its register budget (spills in every op: C2 on the VM 129 -> 106 M waveforms/s)
    typedef __attribute__((address_space(3))) double lds_f64;
    auto* mscr = (lds_f64*)(cx.lds + cx.prog->scratch_off);
    const bool per_event = op.ic[1] != 0;
    if (per_event) {
        const T tau1 = cx.scalar(op.sp[0]), tau2 = cx.scalar(op.sp[1]), fr_t = cx.scalar(op.sp[2]);
        par_nan = (tau1 != tau1) || (tau2 != tau2) || (fr_t != fr_t);
        const double a = pz_decay((double)tau1), b = pz_decay((double)tau2), fr = (double)fr_t;  // pole_zero.py:168-174
        d1 = ((fr * b - fr * a) - b) - 1.0;
        d2 = -1.0 * ((fr * b - fr * a) - b);
        n1 = -1.0 * (a + b);
        n2 = a * b;
        struct M2 {
            double a, b, c, d;  // [[a, b], [c, d]]
        };
        auto mul = [](M2 x, M2 y) { return M2{x.a * y.a + x.b * y.c, x.a * y.b + x.b * y.d, x.c * y.a + x.d * y.c, x.c * y.b + x.d * y.d}; };
        M2 base{-d1, -d2, 1.0, 0.0}, pw{1.0, 0.0, 0.0, 1.0};  // (values, not arrays: an indexed local array would live in scratch memory)
        for (int e = C; e; e >>= 1) {  // pw = M^C
            if (e & 1) pw = mul(pw, base);
            base = mul(base, base);
        }
#pragma unroll 1
        for (int d = 0; d < 6; ++d) {
            if (lane == 0) {
                mscr[4 * d] = pw.a;
                mscr[4 * d + 1] = pw.b;
                mscr[4 * d + 2] = pw.c;
                mscr[4 * d + 3] = pw.d;
            }
            pw = mul(pw, pw);
        }
        wave_sync();
    }
    if (cx.slot_nan(op.src) || par_nan) {
        cx.set_nan(op.dst, true);
        return;
    }
    const auto* ps = cx.chunk(ss);
    auto* pd = cx.chunk(sd);
    // the two samples preceding this chunk
    const double xm1_in = (double)wave_prev(ps[C - 1]);
    const double xm2_in = (double)wave_prev(ps[C - 2]);
    // pass 1: forced response (lane 0: true response from y[0], y[1])
    double y1 = 0.0, y0 = 0.0, xm1 = xm1_in, xm2 = xm2_in;
    int t_begin = 0;
    if (lane == 0) {
        y0 = (double)ps[0];
        y1 = (double)ps[1];
        xm2 = y0;
        xm1 = y1;
        t_begin = 2;
    }
#pragma unroll 4
    for (int t = t_begin; t < C; ++t) {
        const double x = (double)ps[t];
        const double y2 = (((x + n1 * xm1) + n2 * xm2) - d1 * y1) - d2 * y0;
        y0 = y1;
        y1 = y2;
        xm2 = xm1;
        xm1 = x;
    }
    // scan of the affine maps with a common linear part
    double r1 = y1, r0 = y0;
#pragma unroll
    for (int d = 0; d < 6; ++d) {
        double M[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) M[k] = per_event ? (double)mscr[4 * d + k] : op.fc[4 + 4 * d + k];
        const double p1 = wave_shift_up(r1, 1 << d), p0 = wave_shift_up(r0, 1 << d);
        r1 += M[0] * p1 + M[1] * p0;
        r0 += M[2] * p1 + M[3] * p0;
    }
    wave_sync();  // (the scratch area is free again)
    // carry in = state at the end of the previous chunk
    y1 = wave_prev(r1);
    y0 = wave_prev(r0);
    xm1 = xm1_in;
    xm2 = xm2_in;
    bool nan = false;
    if (lane == 0) {
        const T a = ps[0], b = ps[1];
        y0 = (double)a;
        y1 = (double)b;
        xm2 = y0;
        xm1 = y1;
        pd[0] = a;
        pd[1] = b;
    }
#pragma unroll 4
    for (int t = t_begin; t < C; ++t) {
        const double x = (double)ps[t];
        const double y2 = (((x + n1 * xm1) + n2 * xm2) - d1 * y1) - d2 * y0;
        const T y = (T)y2;
        nan |= (y != y);
        pd[t] = y;
        y0 = y1;
        y1 = y2;
        xm2 = xm1;
        xm1 = x;
    }
    cx.set_nan(op.dst, wave_any(nan));
    wave_sync();
}

// ------------------------------------------------------------------------------------------------
// trapezoidal filters  (processors/trap_filters.py: trap_filter :12-76, trap_norm :79-149, asym_trap_filter :152-227)
//
// The reference accumulates y[i] = y[i-1] (+) increments with the feedback going through the float32 output array, i.e. it
// rounds to float32 after every operation (trap_filter) or after every sample (trap_norm/asym).  That rounding noise is
// ~1e-6 of the flat-top value, so a mathematically exact scan does NOT reproduce the reference to 1e-6.  Instead every lane
// replays the reference's operation sequence over its chunk starting from a speculative carry g_j (the exact value of the
// filter at the chunk boundary, from float64 prefix sums); because float32 addition commutes with a shift by a multiple of
// the current ulp, the true carries follow from the per-chunk increments h_j - g_j by one more exact scan, and the replayed
// values are corrected by the (ulp-multiple) offset.  Residual deviations come only from binade crossings / ties.
//
// lags:  FILTER/NORM  L1 = rise, L2 = rise+flat, L3 = 2 rise+flat        signs + - - +
//        ASYM         L1 = rise, L2 = rise+flat, L3 = rise+flat+fall
// ic: 0..2 lags, 3..5 q_k = L_k / C, 6..8 rho_k = L_k % C;  fc: 0 rise, 1 fall
// ------------------------------------------------------------------------------------------------
#ifndef VM_FULL_AMAX
#define VM_FULL_AMAX 1  // (0: A/B builds without the maximum-only replay of full rows)
#endif
constexpr int TRAP_NCAP = 4;
__device__ __forceinline__ float keep_larger(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ double keep_larger(double a, double b) { return __builtin_fmax(a, b); }

// Runs the trap emulation over slot `ss`.  If STORE, writes the filtered waveform into slot `sd` (must differ from ss).
// cap_idx[c] (uniform, -1 = unused): sample indices whose filtered value is wanted; returned in cap_val[c] (uniform).
// RED (with STORE false): the filtered waveform only feeds reductions -- min_max and / or time_point_thresh -- and is never stored
// (DSP_OP_TRAP_REDUCE): extremes are tracked during the replay (the correction by delta_j is monotone, so the extreme of the
// corrected values is the corrected extreme), the threshold walk is a second replay comparing consecutive corrected samples.
// RED = 2: only the maximum is wanted (numpy.amax of the trapezoid): one compare per sample instead of two extremes with their indices.
// RED = 3: only the threshold walk is wanted (no min_max registers): the first replay tracks nothing, it just yields the carries.
// RED = 4: as 2 on rows that fill their chunks (len = 64 C: every index is a sample): one v_max per sample -- a NaN never replaces the
// running maximum in either form; the two differ in the sign of a maximum that is zero.
template <typename T, int KIND, bool STORE, int RED = 0>
__device__ __forceinline__ void trap_core(Ctx<T>& cx, const DSP_PROG DevOp& op, const DSP_PROG DevSlot& ss, const DSP_PROG DevSlot& sd, const int* cap_idx, T* cap_val) {
    const int C = ss.C, lane = lane_id();
    const auto* ps = cx.chunk(ss);
    const double rr = op.fc[0], ll = op.fc[1];
    const double inv_rr = 1.0 / rr, inv_ll = 1.0 / ll;  // (rise or fall == 0 never gets here: ZeroDivisionError at chain creation)
    int q[3], rho[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        q[k] = op.ic[3 + k];
        rho[k] = op.ic[6 + k];
    }
    // ---- pass A: running sum of the input in T (it only feeds the speculative carries, which need not be exact),
    // captured at the three offsets the lagged chunk boundaries fall on
    T run = (T)0, cap[3] = {(T)0, (T)0, (T)0};
    int coff[3];  // offset in the chunk at which the lagged chunk boundary falls: (C - rho) mod C without the integer division (0 <= rho < C)
#pragma unroll
    for (int k = 0; k < 3; ++k) coff[k] = rho[k] ? C - rho[k] : 0;
    {
        int t = 0;
        while (t < C) {
            int nb = C;
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (coff[k] > t && coff[k] < nb) nb = coff[k];
#pragma unroll 8
            for (int u = t; u < nb; ++u) run += ps[u];
            t = nb;
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (t == coff[k]) cap[k] = run;
        }
    }
    const double E = wave_exscan_add((double)run);
    double G = E;  // sum of all samples before this chunk
    double A[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int o = coff[k];
        const double ak = E + (o == 0 ? 0.0 : (double)cap[k]);  // prefix up to (chunk, offset o)
        A[k] = wave_shift_up(ak, q[k] + (rho[k] > 0 ? 1 : 0));  // the chunk that index jC - L_k falls into
    }
    if (KIND == TRAP_FILTER)
        G = ((E - A[0]) - A[1]) + A[2];
    else if (KIND == TRAP_NORM)
        G = (((E - A[0]) - A[1]) + A[2]) / rr;
    else
        G = (E - A[0]) / rr - (A[1] - A[2]) / ll;
    const T g = (lane == 0) ? (T)-0.0 : (T)G;

    // ---- pass B: replay the reference's rounding sequence from the speculative carry
    // lagged sample i - L_k lives in chunk (lane - q_k - 1) at offset C - rho_k + t, one element further once t >= rho_k
    // (the chunk pad).  Lanes whose lagged chunk index is negative read the zero guard below the slot.
    const typename Ctx<T>::LT* lag[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int jj = lane - q[k] - 1;
        // jj >= 0: always inside the slot; jj == -1: offsets t < rho_k fall into the guard chunk right below the slot;
        // jj < -1: every lagged sample of this lane is before sample 0 -> park the pointer at the bottom of the guard
        lag[k] = (jj >= -1) ? cx.lds + ss.off + jj * ss.pitch + (C - rho[k]) : cx.lds + ss.off - 2 * ss.pitch;
    }
    int cap_lane[TRAP_NCAP], cap_off[TRAP_NCAP];
    T capv[TRAP_NCAP];
#pragma unroll
    for (int c = 0; c < TRAP_NCAP; ++c) {
        const int ci = cap_idx[c];
        const int cl = ci >= 0 ? (int)(((float)ci + 0.5f) * ss.invC) : -1;
        cap_lane[c] = cl;
        cap_off[c] = ci >= 0 ? ci - cl * C : -1;
        capv[c] = (T)0;
    }
    typename Ctx<T>::LT* __restrict__ pd = STORE ? cx.chunk(sd) : nullptr;  // (another slot than the ones the steps read: loads may pass stores)
    T y = g;
    const int n_valid = ss.len, i_first = lane * C;
    T vmin = __builtin_huge_val(), vmax = -__builtin_huge_val();
    int imin = 0x7fffffff, imax = 0x7fffffff;
    {
        int t = 0;
        while (t < C) {
            int nb = C;
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (rho[k] > t && rho[k] < nb) nb = rho[k];
#pragma unroll
            for (int c = 0; c < TRAP_NCAP; ++c)
                if (cap_off[c] + 1 > t && cap_off[c] + 1 < nb) nb = cap_off[c] + 1;
            const auto* l0 = lag[0] + (t >= rho[0] ? 1 : 0);
            const auto* l1 = lag[1] + (t >= rho[1] ? 1 : 0);
            const auto* l2 = lag[2] + (t >= rho[2] ? 1 : 0);
#pragma unroll 8
            for (int u = t; u < nb; ++u) {
                y = trap_step_r<T, KIND>(y, ps[u], l0[u], l1[u], l2[u], rr, ll, inv_rr, inv_ll);
                if (STORE) pd[u] = y;
                if (RED == 4) {
                    vmax = keep_larger(vmax, y);
                } else if (RED == 2) {
                    vmax = (i_first + u < n_valid && y > vmax) ? y : vmax;
                } else if (RED == 1) {  // (selects, no branches: strict comparisons keep the first occurrence)
                    const int idx = i_first + u;
                    const bool lt = idx < n_valid && y < vmin, gt = idx < n_valid && y > vmax;
                    vmin = lt ? y : vmin;
                    imin = lt ? idx : imin;
                    vmax = gt ? y : vmax;
                    imax = gt ? idx : imax;
                }
            }
            t = nb;
#pragma unroll
            for (int c = 0; c < TRAP_NCAP; ++c)
                if (t == cap_off[c] + 1) capv[c] = y;
        }
    }
    // ---- true carries: exact scan of the per-chunk increments
    const double D = (double)y - (double)g;
    const double tstart = wave_exscan_add(D);
    const double delta = tstart - (double)g;
    if (STORE) {
        if (delta != 0.0) {
#pragma unroll 8
            for (int t = 0; t < C; ++t) pd[t] = (T)((double)pd[t] + delta);
        }
    }
#pragma unroll
    for (int c = 0; c < TRAP_NCAP; ++c) {
        const T v = (T)((double)capv[c] + delta);
        cap_val[c] = cap_lane[c] >= 0 ? readlane(v, cap_lane[c]) : (T)0;
    }
    if constexpr (RED == 2 || RED == 4) {
        T cmax = vmax != -__builtin_huge_val() ? (T)((double)vmax + delta) : vmax;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            const T o = __shfl_xor(cmax, m);
            cmax = o > cmax ? o : cmax;
        }
        if (lane == 0) cx.sregs()[op.dst + 3] = cmax;
        wave_sync();
    } else if constexpr (RED != 0) {
      if constexpr (RED == 1) {
        // ---- min_max (min_max.py:11-82) over the corrected values: lowest index wins ties
        T cmin = imin != 0x7fffffff ? (T)((double)vmin + delta) : vmin, cmax = imax != 0x7fffffff ? (T)((double)vmax + delta) : vmax;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            const T omin = __shfl_xor(cmin, m), omax = __shfl_xor(cmax, m);
            const int oimin = __shfl_xor(imin, m), oimax = __shfl_xor(imax, m);
            if (omin < cmin || (omin == cmin && oimin < imin)) {
                cmin = omin;
                imin = oimin;
            }
            if (omax > cmax || (omax == cmax && oimax < imax)) {
                cmax = omax;
                imax = oimax;
            }
        }
        if (op.dst >= 0 && lane == 0) {
            auto* r = cx.sregs();
            r[op.dst] = (T)imin;
            r[op.dst + 1] = (T)imax;
            r[op.dst + 2] = cmin;
            r[op.dst + 3] = cmax;
        }
        wave_sync();
      }
        // ---- time_point_thresh (time_point_thresh.py:12-92) on the corrected values: second replay, consecutive samples compared
        if (op.io >= 0) {
            const T thr = cx.scalar(op.sp[0]), ts_f = cx.scalar(op.sp[1]), walk_f = cx.scalar(op.sp[2]);  // (t_start may be the t_max just written)
            T out = quiet_nan<T>();
            if (!(thr != thr || ts_f != ts_f || walk_f != walk_f)) {
                if (floor((double)ts_f) != (double)ts_f) {
                    cx.fatal(DSP_E_TPT_START_INT);
                } else if (floor((double)walk_f) != (double)walk_f) {
                    cx.fatal(DSP_E_TPT_WALK_INT);
                } else if ((long long)ts_f < 0 || (long long)ts_f >= n_valid) {
                    cx.fatal(DSP_E_TPT_RANGE);
                } else {
                    const int ts = (int)ts_f;
                    const bool forward = (long long)walk_f == 1;
                    T prv = wave_prev((T)((double)y + delta));  // last corrected sample of the previous lane's chunk
                    int best = forward ? 0x7fffffff : -1;
                    T yy = g;
                    int t = 0;
                    while (t < C) {
                        int nb = C;
#pragma unroll
                        for (int k = 0; k < 3; ++k)
                            if (rho[k] > t && rho[k] < nb) nb = rho[k];
                        const auto* l0 = lag[0] + (t >= rho[0] ? 1 : 0);
                        const auto* l1 = lag[1] + (t >= rho[1] ? 1 : 0);
                        const auto* l2 = lag[2] + (t >= rho[2] ? 1 : 0);
#pragma unroll 8
                        for (int u = t; u < nb; ++u) {
                            yy = trap_step_r<T, KIND>(yy, ps[u], l0[u], l1[u], l2[u], rr, ll, inv_rr, inv_ll);
                            const T cur = (T)((double)yy + delta);
                            const int idx = i_first + u;
                            if (forward) {  // smallest i = idx - 1 in [ts, n-2] with (w[i] <= thr < w[i+1]) or (w[i] >= thr > w[i+1])
                                const bool hit = ((prv <= thr && thr < cur) || (prv >= thr && thr > cur)) && idx - 1 >= ts && idx >= 1 && idx < n_valid;
                                best = (hit && best == 0x7fffffff) ? idx - 1 : best;
                            } else {        // largest i = idx in [1, ts] with (w[i-1] < thr <= w[i]) or (w[i-1] > thr >= w[i])
                                const bool hit = ((prv < thr && thr <= cur) || (prv > thr && thr >= cur)) && idx >= 1 && idx <= ts && idx < n_valid;
                                best = hit ? idx : best;
                            }
                            prv = cur;
                        }
                        t = nb;
                    }
                    if (forward) {
                        best = wave_min(best);
                        if (best != 0x7fffffff) out = (T)best;
                    } else {
                        best = wave_max(best);
                        if (best >= 0) out = (T)best;
                    }
                }
            }
            if (lane == 0) cx.sregs()[op.io] = out;
            wave_sync();
        }
    }
}

// TRAP_REDUCE: trap filter whose only consumers are min_max and / or time_point_thresh -- the filtered waveform never exists
// (no second 8192-sample slot: the LEGEND t0 chain asym_trap_filter -> min_max -> time_point_thresh fits twice as many wavefronts).
// dst = first of the four min_max registers (t_min, t_max, a_min, a_max) or -1; io = time_point_thresh register or -1;
// sp[0..2] = threshold, t_start, walk_forward; ip[0..2] = rise, flat, fall; ip[3] = trapezoid opcode
template <typename T>
__device__ __forceinline__ void op_trap_reduce(Ctx<T>& cx, const DSP_PROG DevOp& op) {
    const DSP_PROG DevSlot& ss = cx.prog->slots[op.src];
    // ip[3]: bits 0-7 the trapezoid's opcode; bits 8-15 the mode of a fixed_time_pickoff that also reads the trapezoid (0: none), at
    // sp[3], into register bits 16-29 minus one; bit 30: of the four min_max values only a_max is wanted (numpy.amax)
    const int kind = op.ip[3] & 0xff, pk_mode = (op.ip[3] >> 8) & 0xff, pk_reg = ((op.ip[3] >> 16) & 0x3fff) - 1;
    const bool amax_only = ((op.ip[3] >> 30) & 1) != 0;
    if (cx.slot_nan(op.src) || op.ic[9]) {
        if (lane_id() == 0) {
            auto* r = cx.sregs();
            if (op.dst >= 0)
                for (int k = 0; k < 4; ++k) r[op.dst + k] = quiet_nan<T>();
            if (op.io >= 0) r[op.io] = quiet_nan<T>();
            if (pk_reg >= 0) r[pk_reg] = quiet_nan<T>();
        }
        wave_sync();
        return;
    }
    int idx[TRAP_NCAP] = {-1, -1, -1, -1};
    T w4[TRAP_NCAP];
    T t_in = (T)0;
    bool pick = false;
    if (pk_reg >= 0) {
        t_in = cx.scalar(op.sp[3]);
        pick = pickoff_in_range(t_in, ss.len);
        if (pick) {
            const int i0 = (int)t_in;
#pragma unroll
            for (int k = 0; k < TRAP_NCAP; ++k) {
                const int e = i0 - 1 + k;
                const bool need = (k == 1) || (k == 2) || pk_mode == 'h';
                idx[k] = (need && e >= 0 && e < ss.len) ? e : -1;
            }
        }
    }
    const bool full = VM_FULL_AMAX && ss.len == 64 * ss.C;  // (every Ge recipe's rows: 8192 = 64 x 128)
    if (amax_only && kind == DSP_OP_TRAP_FILTER && full)
        trap_core<T, TRAP_FILTER, false, 4>(cx, op, ss, ss, idx, w4);
    else if (amax_only && kind == DSP_OP_TRAP_NORM && full)
        trap_core<T, TRAP_NORM, false, 4>(cx, op, ss, ss, idx, w4);
    else if (amax_only && kind == DSP_OP_TRAP_FILTER)
        trap_core<T, TRAP_FILTER, false, 2>(cx, op, ss, ss, idx, w4);
    else if (amax_only && kind == DSP_OP_TRAP_NORM)
        trap_core<T, TRAP_NORM, false, 2>(cx, op, ss, ss, idx, w4);
    else if (kind == DSP_OP_TRAP_FILTER)
        trap_core<T, TRAP_FILTER, false, 1>(cx, op, ss, ss, idx, w4);
    else if (kind == DSP_OP_TRAP_NORM)
        trap_core<T, TRAP_NORM, false, 1>(cx, op, ss, ss, idx, w4);
    else if (op.dst < 0 && op.ic[10])  // the t0 chain of the Ge recipes: asymmetric trapezoid, rise a power of two, threshold walk only
        trap_core<T, TRAP_ASYM_P2, false, 3>(cx, op, ss, ss, idx, w4);
    else if (op.dst < 0)
        trap_core<T, TRAP_ASYM, false, 3>(cx, op, ss, ss, idx, w4);
    else
        trap_core<T, TRAP_ASYM, false, 1>(cx, op, ss, ss, idx, w4);
    if (pk_reg >= 0) {
        T out = quiet_nan<T>();
        if (pick) {
            int fc = 0;
            out = pickoff_eval(t_in, pk_mode, ss.len, w4, fc);
            if (fc) cx.fatal(fc);
        }
        if (lane_id() == 0) cx.sregs()[pk_reg] = out;
        wave_sync();
    }
}

template <typename T>
__device__ __forceinline__ void op_trap(Ctx<T>& cx, const DSP_PROG DevOp& op) {
    const DSP_PROG DevSlot& ss = cx.prog->slots[op.src];
    const DSP_PROG DevSlot& sd = cx.prog->slots[op.dst];
    // ic[9]: static "output is all NaN" (rise == 0: the reference reads w_out[-1] = NaN, trap_filters.py:65-66)
    if (cx.slot_nan(op.src) || op.ic[9]) {
        cx.set_nan(op.dst, true);
        return;
    }
    const int none[TRAP_NCAP] = {-1, -1, -1, -1};
    T dummy[TRAP_NCAP];
    if (op.opcode == DSP_OP_TRAP_FILTER)
        trap_core<T, TRAP_FILTER, true>(cx, op, ss, sd, none, dummy);
    else if (op.opcode == DSP_OP_TRAP_NORM)
        trap_core<T, TRAP_NORM, true>(cx, op, ss, sd, none, dummy);
    else
        trap_core<T, TRAP_ASYM, true>(cx, op, ss, sd, none, dummy);
    cx.set_nan(op.dst, false);
    wave_sync();
}

// ------------------------------------------------------------------------------------------------
// fixed_time_pickoff  (processors/fixed_time_pickoff.py:12-125), modes i n f c l h (float64 interpolation weights)
// w4 = samples at i0-1, i0, i0+1, i0+2 (only the in-range ones are used)
// ------------------------------------------------------------------------------------------------
// mode 's' (fixed_time_pickoff.py:107-123): natural cubic spline through the whole waveform, evaluated at t_in.  The reference
// runs the tridiagonal sweep over all samples; its two recurrences (u forward, the second derivatives backward) contract by
// |w2| -> 2 - sqrt(3) = 0.268 per step, so a perturbation 48 samples away is scaled by 3e-28 -- far below one float64 ulp.
// Lane l therefore rebuilds u[i0 + l] from a 48-sample warm-up in the reference's operation order (exact from sample 1 when
// the window reaches it), and the back substitution runs over the 64 lanes' values (the tail beyond them is dropped the same way).
// The forward coefficient w2[i] = -0.5 / (0.5 w2[i-1] + 2) is data independent and stationary in float64 from i = 15 on.
// Out of line on purpose, with everything it needs passed by value: the spline mode is rare, and an outlined function that took the
// interpreter state by reference would force that state into memory for every op (which is what happened when the compiler outlined it
// by itself: C2 on the VM 129 -> 113 M waveforms/s).
struct SlotView {
    int off, len, padw;
    float invC;
};
template <typename T>
__device__ __attribute__((noinline)) T pickoff_spline(const typename Ctx<T>::LT* lds, SlotView ss, T t_in) {
    const int n = ss.len, lane = lane_id();
    const int i0 = (int)t_in;  // 0 <= i0 <= n - 2: the caller handles integer t_in
    const double t0 = (double)t_in - (double)i0, t1 = 1.0 - t0;
    auto X = [&](int i) -> double { return (double)lds[padded_index(ss, i)]; };
    constexpr int WARM = 48;
    constexpr double W2_FIX = -0.2679491924311227;
    const int j = i0 + lane;
    double u = 0.0, w = 0.0;
    if (j >= 1 && j <= n - 2) {
        int s = j - WARM;
        if (s < 1) s = 1;
        if (s - 1 >= 15) {
            w = W2_FIX;
        } else {
            for (int i = 1; i <= s - 1; ++i) w = -0.5 / (0.5 * w + 2.0);
        }
        for (int i = s; i <= j; ++i) {
            const double p = 0.5 * w + 2.0;
            w = -0.5 / p;
            const double d = (X(i + 1) - 2.0 * X(i)) + X(i - 1);
            u = (3.0 * d - 0.5 * u) / p;
        }
    }
    // w2[i] = w2[i] * w2[i+1] + u[i] from the far end down to i0 (w2 = u = 0 outside 1..n-2, as np.zeros leaves them)
    double W = 0.0, W1 = 0.0;
    for (int l = 63; l >= 0; --l) {
        const double wl = readlane(w, l), ul = readlane(u, l);
        W = wl * W + ul;
        if (l == 1) W1 = W;
    }
    const double t1_3 = t1 * (t1 * t1), t0_3 = t0 * (t0 * t0);
    return (T)((t1 * X(i0) + t0 * X(i0 + 1)) + ((t1_3 - t1) * W + (t0_3 - t0) * W1) / 6.0);
}

template <typename T>
__device__ __forceinline__ void op_pickoff(Ctx<T>& cx, const DSP_PROG DevOp& op) {
    const DSP_PROG DevSlot& ss = cx.prog->slots[op.src];
    const T t_in = cx.scalar(op.sp[0]);
    T out = quiet_nan<T>();
    if (op.ip[1] == 1) {  // wf[i] in a recipe (processing_chain.py:986-990): a view of one sample, not the processor -- no NaN rule
        if (!cx.slot_all_nan(op.src)) out = cx.lds[padded_index(ss, (int)t_in)];
    } else if (op.ip[1] == 2) {  // wf[variable]: get_default (processors/get.py:50-92); the default is a constant (sp[1], checked at creation)
        out = (T)op.sp[1].value;
        const T n_f = (T)ss.len;
        if (t_in > -n_f - (T)1 && t_in < n_f && !cx.slot_all_nan(op.src)) {  // (a NaN index fails both comparisons)
            int i = (int)t_in;  // truncation toward zero, like int()
            i = i < 0 ? i + ss.len : i;
            if (i >= 0 && i < ss.len) {
                const T v = cx.lds[padded_index(ss, i)];
                out = (v != v) ? out : v;
            }
        }
    } else if (op.ip[0] == 's' && !cx.slot_nan(op.src) && pickoff_in_range(t_in, ss.len) && (T)(int)t_in != t_in) {
        out = pickoff_spline<T>(cx.lds, SlotView{ss.off, ss.len, ss.padw, ss.invC}, t_in);
    } else if (!cx.slot_nan(op.src) && pickoff_in_range(t_in, ss.len)) {
        const int i0 = (int)t_in;
        T w4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int e = i0 - 1 + k;
            e = e < 0 ? 0 : (e > ss.len - 1 ? ss.len - 1 : e);
            w4[k] = cx.lds[padded_index(ss, e)];
        }
        int fc = 0;
        out = pickoff_eval(t_in, op.ip[0], ss.len, w4, fc);
        if (fc) cx.fatal(fc);
    }
    if (lane_id() == 0) cx.sregs()[op.dst] = out;
    wave_sync();
}

// TRAP_PICKOFF: trap filter whose only consumer is a fixed_time_pickoff -- the filtered waveform never exists.
template <typename T>
__device__ __forceinline__ void op_trap_pickoff(Ctx<T>& cx, const DSP_PROG DevOp& op) {
    const DSP_PROG DevSlot& ss = cx.prog->slots[op.src];
    const T t_in = cx.scalar(op.sp[0]);
    T out = quiet_nan<T>();
    if (!cx.slot_nan(op.src) && !op.ic[9] && pickoff_in_range(t_in, ss.len)) {
        const int i0 = (int)t_in;
        int idx[TRAP_NCAP];
        const bool wide = (op.io == 'h');
#pragma unroll
        for (int k = 0; k < TRAP_NCAP; ++k) {
            const int e = i0 - 1 + k;
            const bool need = (k == 1) || (k == 2) || wide;
            idx[k] = (need && e >= 0 && e < ss.len) ? e : -1;
        }
        T w4[TRAP_NCAP];
        if (op.ip[3] == DSP_OP_TRAP_FILTER)
            trap_core<T, TRAP_FILTER, false>(cx, op, ss, ss, idx, w4);
        else if (op.ip[3] == DSP_OP_TRAP_NORM)
            trap_core<T, TRAP_NORM, false>(cx, op, ss, ss, idx, w4);
        else
            trap_core<T, TRAP_ASYM, false>(cx, op, ss, ss, idx, w4);
        int fc = 0;
        out = pickoff_eval(t_in, op.io, ss.len, w4, fc);
        if (fc) cx.fatal(fc);
    }
    if (lane_id() == 0) cx.sregs()[op.dst] = out;
    wave_sync();
}

// ------------------------------------------------------------------------------------------------
// windower (processors/windower.py:12-54): w_out[k] = w_in[int(t0) + k], NaN where the window reaches outside the input
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void op_windower(Ctx<T>& cx, const DSP_PROG DevOp& op) {
    const DSP_PROG DevSlot& ss = cx.prog->slots[op.src];
    const DSP_PROG DevSlot& sd = cx.prog->slots[op.dst];
    const T t0 = cx.scalar(op.sp[0]);
    if (cx.slot_nan(op.src) || t0 != t0) {
        cx.set_nan(op.dst, true);
        return;
    }
    const int n = ss.len, m = sd.len;
    // int(t0) truncates toward zero; anything at or beyond the ends gives an all-NaN window either way
    const int beg = t0 >= (T)n ? n : (t0 <= (T)(-m) ? -m : (int)t0);
    const bool inside = beg >= 0 && beg + m <= n;
    for (int e = lane_id(); e < 64 * sd.C; e += 64) {
        const int si = beg + e;
        T v = (T)0;
        if (e < m) v = (si >= 0 && si < n) ? cx.lds[padded_index(ss, si)] : quiet_nan<T>();
        cx.lds[padded_index(sd, e)] = v;
    }
    if (inside)
        cx.set_nan(op.dst, false);
    else
        cx.set_some_nan(op.dst);
    wave_sync();
}

// ------------------------------------------------------------------------------------------------
// avg_current (processors/moving_windows.py:206-249): w_out = (w_in[L:] - w_in[:-L]) / length in T; ic[0] = L, fc[0] = length
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void op_avg_current(Ctx<T>& cx, const DSP_PROG DevOp& op) {
    const DSP_PROG DevSlot& ss = cx.prog->slots[op.src];
    const DSP_PROG DevSlot& sd = cx.prog->slots[op.dst];
    if (cx.slot_nan(op.src)) {
        cx.set_nan(op.dst, true);
        return;
    }
    const int L = op.ic[0], m = sd.len;
    const T length = (T)op.fc[0];
    bool nan = false;
    for (int e = lane_id(); e < 64 * sd.C; e += 64) {
        T v = (T)0;
        if (e < m) {
            v = (T)(cx.lds[padded_index(ss, e + L)] - cx.lds[padded_index(ss, e)]) / length;
            nan |= (v != v);  // (inf - inf)
        }
        cx.lds[padded_index(sd, e)] = v;
    }
    if (wave_any(nan))
        cx.set_some_nan(op.dst);
    else
        cx.set_nan(op.dst, false);
    wave_sync();
}

// ------------------------------------------------------------------------------------------------
// trap_pickoff (processors/trap_filters.py:230-293): (float64 sum of the `rise` samples ending at the pick-off sample minus the
// `rise` samples ending rise + flat earlier) / rise.  The reference adds them one by one; here lanes take strided elements and a
// wavefront scan adds the partial sums (float64 sums of one waveform's float32 samples are exact, so order does not matter).
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void op_trap_window_pickoff(Ctx<T>& cx, const DSP_PROG DevOp& op) {
    const DSP_PROG DevSlot& ss = cx.prog->slots[op.src];
    const T tp = cx.scalar(op.sp[0]);
    const int n = ss.len, rise = op.ip[0], flat = op.ip[1];
    T out = quiet_nan<T>();
    if (!cx.slot_nan(op.src) && !(tp != tp)) {
        if (floor((double)tp) != (double)tp) {
            cx.fatal(DSP_E_TPO_INT);
        } else {
            const double startd = (double)tp + 1.0;
            if (startd <= (double)n && startd >= (double)(2 * (long long)rise + flat)) {
                const int start = (int)startd;
                double s1 = 0.0, s2 = 0.0;
                for (int k = lane_id(); k < rise; k += 64) {
                    s1 += (double)cx.lds[padded_index(ss, start - rise + k)];
                    s2 += (double)cx.lds[padded_index(ss, start - 2 * rise - flat + k)];
                }
                s1 = readlane(wave_scan_add(s1), 63);
                s2 = readlane(wave_scan_add(s2), 63);
                if (rise == 0)
                    cx.fatal(DSP_E_ZERODIV);
                else
                    out = (T)((s1 - s2) / (double)rise);
            }
        }
    }
    if (lane_id() == 0) cx.sregs()[op.dst] = out;
    wave_sync();
}

// ------------------------------------------------------------------------------------------------
// upsampler (processors/upsampler.py:13-56): input sample t lands on the int(upsample) outputs starting at
// int(t * upsample - floor(upsample / 2)); computed per output by inverting that map (ranges of consecutive t do not overlap: their
// starts are at least int(upsample) apart).  fc[0] = upsample, fc[1] = floor(upsample / 2), ic[0] = int(upsample)
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void op_upsampler(Ctx<T>& cx, const DSP_PROG DevOp& op) {
    const DSP_PROG DevSlot& ss = cx.prog->slots[op.src];
    const DSP_PROG DevSlot& sd = cx.prog->slots[op.dst];
    if (cx.slot_nan(op.src)) {
        cx.set_nan(op.dst, true);
        return;
    }
    const int n = ss.len, m = sd.len, cnt = op.ic[0];
    const double up = op.fc[0], half = op.fc[1];
    bool nan = false;
    if ((double)cnt == up) {
        // a whole-number factor (the usual case, 16 in the LEGEND recipes): input t lands on outputs [t * cnt - half, t * cnt - half + cnt),
        // the ranges tile the output, so output j takes input (j + half) / cnt -- integer arithmetic instead of float64 per output
        const int ihalf = (int)half;
        const float inv = 1.0f / (float)cnt;
        // (four loads, then four stores: both go through the same LDS pointer, so a load is never moved ahead of the store before it and
        // every element would wait out a full LDS round trip; 64 * C is a multiple of 256)
        for (int j0 = lane_id(); j0 < 64 * sd.C; j0 += 256) {
            T v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = j0 + 64 * k, x = j + ihalf;
                int t = (int)((float)x * inv);  // x / cnt up to one either way; x < 2^23
                t = (t + 1) * cnt <= x ? t + 1 : t;
                t = t * cnt > x ? t - 1 : t;
                const int tr = t < n ? t : n - 1;
                const T w = cx.lds[padded_index(ss, tr)];
                v[k] = j < m ? (t < n ? w : quiet_nan<T>()) : (T)0;
                nan |= (v[k] != v[k]);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) cx.lds[padded_index(sd, j0 + 64 * k)] = v[k];
        }
        if (wave_any(nan))
            cx.set_some_nan(op.dst);
        else
            cx.set_nan(op.dst, false);
        wave_sync();
        return;
    }
    for (int j = lane_id(); j < 64 * sd.C; j += 64) {
        T v = (T)0;
        if (j < m) {
            v = quiet_nan<T>();
            const long long tc = (long long)floor(((double)j + half) / up);
#pragma unroll
            for (int d = 1; d >= -1; --d) {  // the last input sample that reaches output j wins, as in the reference's loop order
                const long long t = tc + d;
                const long long s0 = (long long)((double)t * up - half);  // (truncation toward zero, like int())
                if (v != v && t >= 0 && t < n && s0 <= j && j < s0 + cnt) v = cx.lds[padded_index(ss, (int)t)];
            }
            nan |= (v != v);
        }
        cx.lds[padded_index(sd, j)] = v;
    }
    if (wave_any(nan))
        cx.set_some_nan(op.dst);
    else
        cx.set_nan(op.dst, false);
    wave_sync();
}

// ------------------------------------------------------------------------------------------------
// moving_window_multi (processors/moving_windows.py:117-204): num_mw moving averages, alternately from the left and from the right.
// One pass:  y[0] = x[0],  y[v] = y[v-1] + (x[v] - (v < L ? x[0] : x[v-L])) / length,  every operation rounded to T and the sum fed
// back through the output array -- the same structure as trap_filter, so the same rounding replay: lane j runs the reference
// recurrence over its chunk from a speculative start (the exact running sum of the increments before the chunk, rounded), the
// true starts are the exact scan of the per-chunk increments, outputs are shifted by the difference.  A pass from the right is the
// same on the mirrored index.  ic[0] = L, ic[1] = num_mw, ic[2] = mw_type, fc[0] = length
// ------------------------------------------------------------------------------------------------
// a / d for an integer-valued window length d in the loop type, correctly rounded: the same three-operation form as div_by_count
// (checked against the IEEE quotient on 200 000 random float32 cases, lengths 1 .. 8192); tiny and non-finite operands divide plainly
__device__ __forceinline__ float mw_abs(float x) { return __builtin_fabsf(x); }
__device__ __forceinline__ double mw_abs(double x) { return __builtin_fabs(x); }
__device__ __forceinline__ float mw_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double mw_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <typename T> __device__ __forceinline__ T mw_qmax();
template <> __device__ __forceinline__ float mw_qmax<float>() { return 3.0e38f; }
template <> __device__ __forceinline__ double mw_qmax<double>() { return 1.0e308; }
template <typename T> __device__ __forceinline__ T mw_amin();
template <> __device__ __forceinline__ float mw_amin<float>() { return 1.0e-30f; }
template <> __device__ __forceinline__ double mw_amin<double>() { return 1.0e-290; }
__device__ __forceinline__ float div_by_length(float a, float d, float inv_d) {
    const float q = a * inv_d;
    if (__builtin_expect(!(__builtin_fabsf(q) <= 3.0e38f && __builtin_fabsf(a) >= 1.0e-30f), 0)) return a / d;
    return __builtin_fmaf(__builtin_fmaf(-q, d, a), inv_d, q);
}
__device__ __forceinline__ double div_by_length(double a, double d, double inv_d) {
    const double q = a * inv_d;
    if (__builtin_expect(!(__builtin_fabs(q) <= 1.0e308 && __builtin_fabs(a) >= 1.0e-290), 0)) return a / d;
    return __builtin_fma(__builtin_fma(-q, d, a), inv_d, q);
}

// One pass.  A lane walks its samples v = v0 + t, t = 0 .. C - 1, i.e. elements e = v (from the left) or n - 1 - v (from the right) of
// the slots, and e -/+ L for the sample leaving the window: consecutive LDS elements with at most one chunk pad on the way, at the same
// t for every lane (the lanes start C elements apart) -- so each stream is a base pointer plus t, and the loops are cut where a stream
// steps over its pad instead of turning every index into an address (that arithmetic was a third of the op's instructions).
// INPLACE: out is in.  What stands in the way is the sample leaving the window, x[v - L]: inside the lane's own chunk it is still there if
// the chunk is walked downwards (what has been overwritten lies above), and the L samples below the chunk -- the end of the previous
// lane's chunk, which that lane overwrites first -- are copied to `side` (64 x Ls elements) before anybody writes.  Needs L <= C.  One
// 4784-sample waveform less in LDS for the current branch of the Ge recipes: four waveforms per CU instead of three.
template <typename T, bool RIGHT, bool INPLACE = false>
__device__ __forceinline__ void mw_pass_dir(Ctx<T>& cx, const DSP_PROG DevSlot& in, const DSP_PROG DevSlot& out, int L, T length,
                                            typename Ctx<T>::LT* side = nullptr, int Ls = 0) {
    typedef typename Ctx<T>::LT LT;
    const int n = in.len, C = in.C, lane = lane_id(), v0 = lane * C;
    constexpr int D = RIGHT ? -1 : 1;
    const T x0 = cx.lds[padded_index(in, RIGHT ? n - 1 : 0)];
    const T inv_len = (T)1 / length;
    // stream of slot s starting at element e0 (this lane; may lie outside the slot: such elements are never read): base pointer, the t at
    // which it enters the next chunk (C: never)
    auto stream = [&](const DSP_PROG DevSlot& s, int e0, LT*& base, int& brk) {
        const int q = (e0 >= 0 ? e0 : e0 - (C - 1)) / C, r = e0 - q * C;  // floor division: r is the same for every lane
        base = cx.lds + s.off + q * s.pitch + r;
        brk = RIGHT ? r + 1 : (r == 0 ? C : C - r);
    };
    LT *p_in, *p_lag, *p_out;
    int brk_io, brk_lag, brk_o;
    const int e0 = RIGHT ? n - 1 - v0 : v0;
    stream(in, e0, p_in, brk_io);
    stream(in, RIGHT ? e0 + L : e0 - L, p_lag, brk_lag);
    stream(out, e0, p_out, brk_o);  // (same length, same chunks: brk_o == brk_io)
    brk_io = uniform(brk_io);
    brk_lag = uniform(brk_lag);
    const int pad_in = in.padw, pad_out = out.padw;
    auto next_cut = [&](int t, bool with_lag) {
        int nb = C;
        if (brk_io > t && brk_io < nb) nb = brk_io;
        if (with_lag && brk_lag > t && brk_lag < nb) nb = brk_lag;
        return nb;
    };
    // Every loop below works on groups of eight samples: the loads of a group first, then its arithmetic with selects instead of branches
    // (v == 0, v < L, v >= n), then its stores -- with a branch per sample the compiler waits for every LDS access on its own, three
    // round trips per sample and pass.  Lanes past the end of the waveform compute on whatever their elements hold and store nothing
    // (predicated stores: in a pass from the right their elements would lie below the slot, in another slot or another wavefront's
    // region); a lagged element before the slot (v < L) is read -- the guard below, or another region: LDS reads cannot fault -- and
    // dropped by the select.
    constexpr int G = 8;
    // pass A: the increments, parked in the output buffer, and their exact (float64) sum over this chunk -> speculative start
    double S = 0.0;
    if constexpr (INPLACE) {
        auto own = [&](int t) -> LT* { return p_in + D * (t + (t >= brk_io ? pad_in : 0)); };  // (t and the break are uniform: scalar selects)
        for (int k0 = 0; k0 < L; k0 += G) {  // the end of every chunk, before anybody overwrites it
            T tmp[G];
#pragma unroll
            for (int k = 0; k < G; ++k) tmp[k] = k0 + k < L ? (T)*own(C - L + k0 + k) : (T)0;
#pragma unroll
            for (int k = 0; k < G; ++k)
                if (k0 + k < L) side[lane * Ls + k0 + k] = tmp[k];
        }
        wave_sync();
        const LT* below = side + (lane > 0 ? lane - 1 : 0) * Ls;  // (lane 0: v < L there, the value is dropped for x0)
        for (int tb = C - G; tb >= 0; tb -= G) {
            T a[G], bq[G], d[G];
#pragma unroll
            for (int k = 0; k < G; ++k) {
                const int t = tb + k;
                a[k] = *own(t);
                if (tb >= L)
                    bq[k] = *own(t - L);
                else if (tb + G <= L)
                    bq[k] = below[t];
                else
                    bq[k] = t >= L ? (T)*own(t - L) : (T)below[t];
            }
            bool odd = false;
#pragma unroll
            for (int k = 0; k < G; ++k) {
                const int v = v0 + tb + k;
                a[k] = a[k] - (v < L ? x0 : bq[k]);
                const T q = a[k] * inv_len;
                odd |= !(mw_abs(q) <= mw_qmax<T>() && mw_abs(a[k]) >= mw_amin<T>()) && a[k] != (T)0;
                d[k] = mw_fma(mw_fma(-q, length, a[k]), inv_len, q);
            }
            if (wave_any(odd)) {
#pragma unroll
                for (int k = 0; k < G; ++k) d[k] = a[k] / length;
            }
#pragma unroll
            for (int k = 0; k < G; ++k) {
                const int v = v0 + tb + k;
                d[k] = v == 0 ? x0 : d[k];
                d[k] = v < n ? d[k] : (T)0;
                if (v < n) *own(tb + k) = d[k];
                S += (double)d[k];
            }
        }
    } else
    for (int t = 0; t < C;) {
        const int nb = next_cut(t, true);
        const LT* xi = p_in + D * (t >= brk_io ? pad_in : 0);
        const LT* xl = p_lag + D * (t >= brk_lag ? pad_in : 0);
        LT* __restrict__ yo = p_out + D * (t >= brk_io ? pad_out : 0);  // (another slot than the samples: loads may pass stores)
        int u = t;
        for (; u + G <= nb; u += G) {
            T a[G], bq[G], d[G];
#pragma unroll
            for (int k = 0; k < G; ++k) {
                a[k] = xi[D * (u + k)];
                bq[k] = xl[D * (u + k)];
            }
            bool odd = false;  // an operand the three-operation quotient is not proven for: tiny, infinite or NaN
#pragma unroll
            for (int k = 0; k < G; ++k) {
                const int v = v0 + u + k;
                a[k] = a[k] - (v < L ? x0 : bq[k]);
                const T q = a[k] * inv_len;
                odd |= !(mw_abs(q) <= mw_qmax<T>() && mw_abs(a[k]) >= mw_amin<T>()) && a[k] != (T)0;
                d[k] = mw_fma(mw_fma(-q, length, a[k]), inv_len, q);
            }
            if (wave_any(odd)) {
#pragma unroll
                for (int k = 0; k < G; ++k) d[k] = a[k] / length;
            }
#pragma unroll
            for (int k = 0; k < G; ++k) {
                const int v = v0 + u + k;
                d[k] = v == 0 ? x0 : d[k];
                d[k] = v < n ? d[k] : (T)0;
                if (v < n) yo[D * (u + k)] = d[k];  // (a lane past the end stores nothing: from the right its elements lie below the slot)
                S += (double)d[k];
            }
        }
        for (; u < nb; ++u) {
            const int v = v0 + u;
            const T xa = xi[D * u], xb = xl[D * u];
            const T df = xa - (v < L ? x0 : xb);
            T d = div_by_length(df, length, inv_len);
            d = v == 0 ? x0 : d;
            d = v < n ? d : (T)0;
            if (v < n) yo[D * u] = d;
            S += (double)d;
        }
        t = nb;
    }
    const double E = wave_exscan_add(S);
    const T g = (lane == 0) ? (T)-0.0 : (T)E;
    // pass B: the reference recurrence from g over the parked increments (each lane reads back what it wrote)
    T y = g;
    for (int t = 0; t < C;) {
        const int nb = next_cut(t, false);
        LT* yo = p_out + D * (t >= brk_io ? pad_out : 0);
        int u = t;
        for (; u + G <= nb; u += G) {
            T d[G];
#pragma unroll
            for (int k = 0; k < G; ++k) d[k] = yo[D * (u + k)];
#pragma unroll
            for (int k = 0; k < G; ++k) {
                y = v0 + u + k < n ? y + d[k] : y;
                d[k] = y;
            }
#pragma unroll
            for (int k = 0; k < G; ++k)
                if (v0 + u + k < n) yo[D * (u + k)] = d[k];
        }
        for (; u < nb; ++u) {
            if (v0 + u < n) {
                y = y + yo[D * u];
                yo[D * u] = y;
            }
        }
        t = nb;
    }
    // true starts: exact scan of the per-chunk increments (y before sample 0 is 0)
    const double Dd = (double)y - (double)g;
    const double delta = wave_exscan_add(Dd) - (double)g;
    wave_sync();
    if (delta != 0.0) {
        for (int t = 0; t < C;) {
            const int nb = next_cut(t, false);
            LT* yo = p_out + D * (t >= brk_io ? pad_out : 0);
            int u = t;
            for (; u + G <= nb; u += G) {
                T d[G];
#pragma unroll
                for (int k = 0; k < G; ++k) d[k] = yo[D * (u + k)];
#pragma unroll
                for (int k = 0; k < G; ++k) d[k] = (T)((double)d[k] + delta);
#pragma unroll
                for (int k = 0; k < G; ++k)
                    if (v0 + u + k < n) yo[D * (u + k)] = d[k];
            }
            for (; u < nb; ++u)
                if (v0 + u < n) yo[D * u] = (T)((double)yo[D * u] + delta);
            t = nb;
        }
    }
    wave_sync();
}

template <typename T>
__device__ __forceinline__ void mw_pass(Ctx<T>& cx, const DSP_PROG DevSlot& in, const DSP_PROG DevSlot& out, int L, T length, bool right) {
    if (right)
        mw_pass_dir<T, true>(cx, in, out, L, length);
    else
        mw_pass_dir<T, false>(cx, in, out, L, length);
}
template <typename T>
__device__ __forceinline__ void mw_pass_inplace(Ctx<T>& cx, const DSP_PROG DevSlot& io, typename Ctx<T>::LT* side, int Ls, int L, T length, bool right) {
    if (right)
        mw_pass_dir<T, true, true>(cx, io, io, L, length, side, Ls);
    else
        mw_pass_dir<T, false, true>(cx, io, io, L, length, side, Ls);
}

template <typename T>
__device__ __forceinline__ void op_moving_window_multi(Ctx<T>& cx, const DSP_PROG DevOp& op) {
    const DSP_PROG DevSlot& ss = cx.prog->slots[op.src];
    const DSP_PROG DevSlot& sd = cx.prog->slots[op.dst];
    const DSP_PROG DevSlot& sq = cx.prog->slots[op.ip[2]];
    const int L = op.ic[0], num = op.ic[1], type = op.ic[2];
    if (cx.slot_nan(op.src) || num == 0) {  // (no window at all leaves the output as it was initialised: NaN)
        cx.set_nan(op.dst, true);
        return;
    }
    if (op.ip[3] == 1) {  // in place (dst is src), ip[2] = a side slot of 64 * L samples for the chunk ends: every pass in the same buffer
        const T length = (T)op.fc[0];
        for (int p = 0; p < num; ++p)
            mw_pass_inplace<T>(cx, ss, cx.lds + sq.off, L | 1, L, length, ((p % 2 == 1) && type == 0) || type == 2);
        cx.set_nan(op.dst, false);
        wave_sync();
        return;
    }
    for (int e = lane_id(); e < 64 * sd.C; e += 64) {  // pads of both targets stay finite
        cx.lds[padded_index(sd, e)] = (T)0;
        if (num > 1 && op.ip[2] != op.src) cx.lds[padded_index(sq, e)] = (T)0;  // (the source as scratch: written whole by its producer)
    }
    wave_sync();
    const T length = (T)op.fc[0];
    for (int p = 0; p < num; ++p) {
        const bool right = ((p % 2 == 1) && type == 0) || type == 2;
        const bool to_dst = ((num - 1 - p) % 2) == 0;  // the last pass lands in dst
        const DSP_PROG DevSlot& in = p == 0 ? ss : (to_dst ? sq : sd);
        mw_pass(cx, in, to_dst ? sd : sq, L, length, right);
    }
    cx.set_nan(op.dst, false);
    wave_sync();
}

// ------------------------------------------------------------------------------------------------
// linear_slope_fit (processors/linear_slope_fit.py:11-91): Welford mean / variance and a least-squares line.
// The mean and the variance feed back through float32 (T) accumulators that round after every sample, and the update
// mean += (x - mean) / (i + 1) does not commute with a shift of its state (it contracts it), so unlike the trapezoids there is no
// parallel replay: the recurrence is run sample by sample, by every lane redundantly (the LDS reads are broadcasts).  That makes
// this the slowest op of the VM by far (a float64 division per sample on a dependent chain); a lane-per-waveform kernel is the
// right shape for it.  The regression sums are float64 and order-free: lanes take their chunks, one wavefront scan.
// Typing as numba resolves it (PARITY UNPINNED, see oracle/dsp_oracle_impl.h): temp, the products and the accumulators in T,
// temp / (i + 1) and stdev / (n - 1) in float64 rounded back to T.
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void op_linear_slope_fit(Ctx<T>& cx, const DSP_PROG DevOp& op) {
    const DSP_PROG DevSlot& ss = cx.prog->slots[op.src];
    // the fit runs on samples [first, first + n) of the slot: a constant slice wf[a:b] of an intermediate costs no copy
    const int first = op.ic[0], n = op.ic[1], C = ss.C, lane = lane_id();
    T o_mean = quiet_nan<T>(), o_std = o_mean, o_slope = o_mean, o_icpt = o_mean;
    if (!cx.slot_nan(op.src)) {
        // regression sums
        const auto* ps = cx.chunk(ss);
        double sy = 0.0, sxy = 0.0;
#pragma unroll 8
        for (int t = 0; t < C; ++t) {
            const int i = lane * C + t - first;
            if (i >= 0 && i < n) {
                const double x = (double)ps[t];
                sy += x;
                sxy += x * (double)i;
            }
        }
        sy = readlane(wave_scan_add(sy), 63);
        sxy = readlane(wave_scan_add(sxy), 63);
        // Welford, sequential, by every lane redundantly.  What shortens it: the samples are read as LDS broadcasts at addresses that
        // stay in scalar registers (chunk by chunk), the reciprocals of the counts are formed by the lanes in parallel (one division
        // per lane per segment of at most 64 samples, off the chain) and handed over through the scratch area, and the chain divides
        // with div_by_count -- the correctly rounded quotient in 3 dependent operations.  A waveform with an infinity in it (sy not
        // finite) keeps the plain division: there the residual of the short form is NaN.
        T m = (T)0, s = (T)0;
        const bool finite = (sy - sy) == 0.0;
        typedef __attribute__((address_space(3))) double lds_f64;
        auto* scr = (lds_f64*)(cx.lds + cx.prog->scratch_off);
        int c0 = first / C, t = first - c0 * C;                   // (uniform: chunk and offset of the next sample)
        const auto* row = cx.lds + ss.off + c0 * ss.pitch;
        for (int i = 0; i < n;) {
            int cnt = C - t;
            if (cnt > 64) cnt = 64;
            if (cnt > n - i) cnt = n - i;
            scr[lane] = 1.0 / (double)(i + lane + 1);
            wave_sync();
            const auto* p = row + t;
            if (finite) {
#pragma unroll 8
                for (int u = 0; u < cnt; ++u) {
                    const T x = p[u];
                    const double inv = scr[u], d = (double)(i + u + 1);
                    const T temp = x - m;
                    const double td = (double)temp, q = td * inv;
                    m = (T)((double)m + __builtin_fma(__builtin_fma(-q, d, td), inv, q));
                    s = s + temp * (x - m);
                }
            } else {
                for (int u = 0; u < cnt; ++u) {
                    const T x = p[u];
                    const T temp = x - m;
                    m = (T)((double)m + (double)temp / (double)(i + u + 1));
                    s = s + temp * (x - m);
                }
            }
            wave_sync();
            i += cnt;
            t += cnt;
            if (t == C) {
                t = 0;
                row += ss.pitch;
            }
        }
        s = (T)((double)s / (double)(n - 1));
        s = (T)sqrt((double)s);
        const long long nn = n, sum_x = nn * (nn - 1) / 2, sum_x2 = (nn - 1) * nn * (2 * nn - 1) / 6;
        o_mean = m;
        o_std = s;
        o_slope = (T)(((double)nn * sxy - (double)sum_x * sy) / (double)(nn * sum_x2 - sum_x * sum_x));
        o_icpt = (T)((sy - (double)sum_x * (double)o_slope) / (double)nn);
    }
    if (lane == 0) {
        auto* r = cx.sregs();
        r[op.dst] = o_mean;
        r[op.dst + 1] = o_std;
        r[op.dst + 2] = o_slope;
        r[op.dst + 3] = o_icpt;
    }
    wave_sync();
}

// ------------------------------------------------------------------------------------------------
// mean_below_threshold (processors/arithmetic.py:9-62): float64 total of the samples below the threshold / their count.
// The reference adds them one by one; here per-lane partial sums and a wavefront scan.  For float32 samples of one
// waveform's dynamic range the float64 sums are exact, hence order independent; otherwise the last float64 bit may differ.
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void op_mean_below(Ctx<T>& cx, const DSP_PROG DevOp& op) {
    const DSP_PROG DevSlot& ss = cx.prog->slots[op.src];
    const T thr = cx.scalar(op.sp[0]);
    T out = quiet_nan<T>();
    if (!cx.slot_nan(op.src) && !(thr != thr)) {
        const int n = ss.len, C = ss.C, i0 = lane_id() * C;
        const auto* ps = cx.chunk(ss);
        double total = 0.0, count = 0.0;
#pragma unroll 8
        for (int t = 0; t < C; ++t) {
            const T v = ps[t];
            if (i0 + t < n && v < thr) {
                total += (double)v;
                count += 1.0;
            }
        }
        total = readlane(wave_scan_add(total), 63);
        count = readlane(wave_scan_add(count), 63);
        if (count > 0.0) out = (T)(total / count);
    }
    if (lane_id() == 0) cx.sregs()[op.dst] = out;
    wave_sync();
}

// ------------------------------------------------------------------------------------------------
// the threshold walk of time_point_thresh / interpolated_time_point_thresh: 64 consecutive samples per step, starting at t_start and
// moving away from it, and the walk ends with the first step that holds a crossing -- in the recipes the start is the previous time
// point and the answer lies a few samples away (a walk that finds nothing visits the whole range, like the reference's loop).
//   forward:  smallest i in [ts, n - 2] with (w[i] <= thr < w[i+1]) or (w[i] >= thr > w[i+1])
//   backward: largest  i in [back_lo, ts] with (w[i-1] < thr <= w[i]) or (w[i-1] > thr >= w[i])
// Returns i or -1.  Comparisons only: bit exact whatever the order of the visits.
// ------------------------------------------------------------------------------------------------
template <typename T, typename SlotRef>
__device__ __forceinline__ int find_crossing(Ctx<T>& cx, const SlotRef& ss, T thr, int ts, bool forward, int back_lo) {
    const int n = ss.len, lane = lane_id();
    // The first two steps one at a time (that is where the recipes' walks end); a walk that goes on takes four steps per round, their
    // eight LDS reads in flight together -- a threshold that is never reached (tp_100 of a pulse below its own flat top) used to cost
    // an LDS round trip for each of the 128 steps of an 8192-sample waveform.
    if (forward) {
        int b = ts;
        for (int k = 0; k < 2 && b <= n - 2; ++k, b += 64) {
            const int i = b + lane;
            bool hit = false;
            if (i <= n - 2) {
                const T cur = cx.lds[padded_index(ss, i)], nxt = cx.lds[padded_index(ss, i + 1)];
                hit = (cur <= thr && thr < nxt) || (cur >= thr && thr > nxt);
            }
            const unsigned long long found = __ballot(hit);
            if (found) return b + __builtin_ctzll(found);
        }
        for (; b <= n - 2; b += 256) {
            T cur[4], nxt[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = b + 64 * k + lane, ic = i <= n - 2 ? i : n - 2;  // (clamped: the read is always inside the waveform)
                cur[k] = cx.lds[padded_index(ss, ic)];
                nxt[k] = cx.lds[padded_index(ss, ic + 1)];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = b + 64 * k + lane;
                const bool hit = i <= n - 2 && ((cur[k] <= thr && thr < nxt[k]) || (cur[k] >= thr && thr > nxt[k]));
                const unsigned long long found = __ballot(hit);
                if (found) return b + 64 * k + __builtin_ctzll(found);
            }
        }
    } else {
        int b = ts;
        for (int k = 0; k < 2 && b >= back_lo; ++k, b -= 64) {
            const int i = b - lane;  // (lane 0 looks at the sample nearest to the start)
            bool hit = false;
            if (i >= back_lo) {
                const T cur = cx.lds[padded_index(ss, i)], prv = cx.lds[padded_index(ss, i - 1)];
                hit = (prv < thr && thr <= cur) || (prv > thr && thr >= cur);
            }
            const unsigned long long found = __ballot(hit);
            if (found) return b - __builtin_ctzll(found);
        }
        for (; b >= back_lo; b -= 256) {
            T cur[4], prv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = b - 64 * k - lane, ic = i >= back_lo ? i : back_lo;
                cur[k] = cx.lds[padded_index(ss, ic)];
                prv[k] = cx.lds[padded_index(ss, ic - 1)];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = b - 64 * k - lane;
                const bool hit = i >= back_lo && ((prv[k] < thr && thr <= cur[k]) || (prv[k] > thr && thr >= cur[k]));
                const unsigned long long found = __ballot(hit);
                if (found) return b - 64 * k - __builtin_ctzll(found);
            }
        }
    }
    return -1;
}

// ------------------------------------------------------------------------------------------------
// time_point_thresh  (processors/time_point_thresh.py:12-92): comparisons only -> bit exact
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void op_time_point_thresh(Ctx<T>& cx, const DSP_PROG DevOp& op) {
    const DSP_PROG DevSlot& ss = cx.prog->slots[op.src];
    T thr = cx.scalar(op.sp[0]);
    if (op.ic[0]) thr = thr * (T)op.fc[0];  // (the planner folded `threshold = value * constant` into the walk: dsp_plan.cpp)
    const T ts_f = cx.scalar(op.sp[1]), walk_f = cx.scalar(op.sp[2]);
    T out = quiet_nan<T>();
    const int n = ss.len, lane = lane_id();
    if (!(cx.slot_nan(op.src) || thr != thr || ts_f != ts_f || walk_f != walk_f)) {
        if (floor((double)ts_f) != (double)ts_f) {
            cx.fatal(DSP_E_TPT_START_INT);
        } else if (floor((double)walk_f) != (double)walk_f) {
            cx.fatal(DSP_E_TPT_WALK_INT);
        } else if ((long long)ts_f < 0 || (long long)ts_f >= n) {
            cx.fatal(DSP_E_TPT_RANGE);
        } else {
            const int found = find_crossing(cx, ss, thr, (int)ts_f, (long long)walk_f == 1, 1);
            if (found >= 0) out = (T)found;
        }
    }
    if (lane == 0) cx.sregs()[op.dst] = out;
    wave_sync();
}

// ------------------------------------------------------------------------------------------------
// interpolated_time_point_thresh  (processors/time_point_thresh.py:95-222): the same crossing search with three differences -- the
// start is truncated by int() and a start outside the waveform gives NaN instead of DSPFatal, the backward walk stops at sample 2
// (range(int(t_start), 1, -1)) -- and the result placed between samples i_cross and i_cross + 1 by mode.  Typing as numba resolves
// it: mode 'l' divides in T, adds the int64 index in float64 and rounds to T; 'n' is index + 0.5 in float64.
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void op_interp_time_point_thresh(Ctx<T>& cx, const DSP_PROG DevOp& op) {
    const DSP_PROG DevSlot& ss = cx.prog->slots[op.src];
    const T thr = cx.scalar(op.sp[0]), ts_f = cx.scalar(op.sp[1]), walk_f = cx.scalar(op.sp[2]);
    const int mode = op.ip[0];
    T out = quiet_nan<T>();
    const int n = ss.len, lane = lane_id();
    if (!(cx.slot_nan(op.src) || thr != thr || ts_f != ts_f) && ts_f >= (T)0 && ts_f < (T)n) {
        const bool forward = walk_f > (T)0;
        int ic = find_crossing(cx, ss, thr, (int)ts_f, forward, 2);
        if (!forward && ic >= 0) ic -= 1;  // (the crossing lies between i - 1 and i)
        if (ic >= 0) {
            const T w0 = cx.lds[padded_index(ss, ic)], w1 = cx.lds[padded_index(ss, ic + 1)];
            if (mode == 'i' || mode == 'b' || mode == 'c') {
                out = (T)ic;
            } else if (mode == 'a' || mode == 'f') {
                out = (T)(ic + 1);
            } else if (mode == 'r') {
                const T d0 = thr - w0, d1 = thr - w1;
                out = (T)((d0 < (T)0 ? -d0 : d0) < (d1 < (T)0 ? -d1 : d1) ? ic : ic + 1);
            } else if (mode == 'n') {
                out = (T)((double)ic + 0.5);
            } else if (mode == 'l') {
                const T q = (thr - w0) / (w1 - w0);
                out = (T)((double)ic + (double)q);
            } else {
                cx.fatal(DSP_E_FTP_MODE);  // "Unrecognized interpolation mode": raised by the reference once a crossing is found
            }
        }
    }
    if (lane == 0) cx.sregs()[op.dst] = out;
    wave_sync();
}

// ------------------------------------------------------------------------------------------------
// min_max  (processors/min_max.py:11-82): first occurrence of the extremes (strict comparisons)
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void op_min_max(Ctx<T>& cx, const DSP_PROG DevOp& op, bool amax_only) {
    const DSP_PROG DevSlot& ss = cx.prog->slots[op.src];
    const int n = ss.len, C = ss.C, lane = lane_id();
    T o_tmin = quiet_nan<T>(), o_tmax = o_tmin, o_amin = o_tmin, o_amax = o_tmin;
    if (!cx.slot_nan(op.src)) {
        const auto* ps = cx.chunk(ss);
        const int i0 = lane * C;
        // lanes entirely beyond n start from sample 0 (broadcast) so they never win
        const T first = cx.lds[ss.off];
        T vmin = (i0 < n) ? ps[0] : first, vmax = vmin;
        int imin = (i0 < n) ? i0 : 0, imax = imin;
#pragma unroll 8
        for (int t = 1; t < C; ++t) {
            const T v = ps[t];
            const int i = i0 + t;
            const bool lt = i < n && v < vmin, gt = i < n && v > vmax;  // (selects, no branches: strict comparisons keep the first occurrence)
            vmin = lt ? v : vmin;
            imin = lt ? i : imin;
            vmax = gt ? v : vmax;
            imax = gt ? i : imax;
        }
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            const T ovmin = __shfl_xor(vmin, m), ovmax = __shfl_xor(vmax, m);
            const int oimin = __shfl_xor(imin, m), oimax = __shfl_xor(imax, m);
            if (ovmin < vmin || (ovmin == vmin && oimin < imin)) {
                vmin = ovmin;
                imin = oimin;
            }
            if (ovmax > vmax || (ovmax == vmax && oimax < imax)) {
                vmax = ovmax;
                imax = oimax;
            }
        }
        // value = the sample at the winning index (keeps the sign of a zero like w_in[min_index])
        o_amin = cx.lds[padded_index(ss, imin)];
        o_amax = cx.lds[padded_index(ss, imax)];
        o_tmin = (T)imin;
        o_tmax = (T)imax;
    }
    if (lane == 0) {
        auto* r = cx.sregs();
        if (amax_only) {
            r[op.dst] = o_amax;
        } else {
            r[op.dst + 0] = o_tmin;
            r[op.dst + 1] = o_tmax;
            r[op.dst + 2] = o_amin;
            r[op.dst + 3] = o_amax;
        }
    }
    wave_sync();
}

// ------------------------------------------------------------------------------------------------
// discrete_wavelet_transform, Haar  (processors/dwt.py:13-81 -> pywt.downcoef, 'symmetric' extension)
//   level by level: out[k] = fl(fl(f0*x[2k+1]) + fl(f1*x[2k])), f = (c, c) for 'a', (-c, c) for the last level of 'd'.
// Runs in place on the scratch slot ip[2] (a copy of src, or src itself when it is dead afterwards).
// ------------------------------------------------------------------------------------------------
// Lengths that are a multiple of 2^level and chunks that are, too (4096 / 8192 samples, level <= 5): every output is the tree of one
// run of G = 2^level consecutive samples of one lane's chunk -- no symmetric extension, no exchange between lanes, no intermediate
// level in LDS.  The same products and sums per level as the level-by-level form.
template <typename T, int LEVEL>
__device__ __forceinline__ void dwt_haar_local(Ctx<T>& cx, const DSP_PROG DevSlot& ss, const DSP_PROG DevSlot& sd, int part) {
    constexpr int G = 1 << LEVEL;
    const T c = (T)0.7071067811865476;
    const int lane = lane_id(), C = ss.C, first = lane * C;
    const auto* ps = cx.chunk(ss);
    for (int g0 = 0; g0 < C; g0 += G) {
        if (first + g0 >= ss.len) break;
        T v[G];
#pragma unroll
        for (int k = 0; k < G; ++k) v[k] = ps[g0 + k];
#pragma unroll
        for (int l = 0; l < LEVEL; ++l) {
            const T f0 = (l == LEVEL - 1 && part == 'd') ? -c : c;
#pragma unroll
            for (int k = 0; k < (G >> (l + 1)); ++k) v[k] = (T)(f0 * v[2 * k + 1]) + (T)(c * v[2 * k]);
        }
        cx.lds[padded_index(sd, (first + g0) >> LEVEL)] = v[0];
    }
}

template <typename T>
__device__ __forceinline__ void op_dwt_haar(Ctx<T>& cx, const DSP_PROG DevOp& op) {
    const DSP_PROG DevSlot& ss = cx.prog->slots[op.src];
    const DSP_PROG DevSlot& st = cx.prog->slots[op.ip[2]];
    const DSP_PROG DevSlot& sd = cx.prog->slots[op.dst];
    if (cx.slot_nan(op.src)) {
        cx.set_nan(op.dst, true);
        return;
    }
    const int level = op.ip[0], part = op.ip[1], lane = lane_id();
    constexpr int MAX_LOCAL = sizeof(T) == 8 ? 4 : 5;  // (32 float64 values in registers would spill the lean float64 build)
    if (level <= MAX_LOCAL && ss.len % (1 << level) == 0 && ss.C % (1 << level) == 0) {
        switch (level) {
            case 1: dwt_haar_local<T, 1>(cx, ss, sd, part); break;
            case 2: dwt_haar_local<T, 2>(cx, ss, sd, part); break;
            case 3: dwt_haar_local<T, 3>(cx, ss, sd, part); break;
            case 4: dwt_haar_local<T, 4>(cx, ss, sd, part); break;
            default:
                if constexpr (MAX_LOCAL >= 5) dwt_haar_local<T, 5>(cx, ss, sd, part);
                break;
        }
        wave_sync();
        for (int e = sd.len + lane; e < 64 * sd.C; e += 64) cx.lds[padded_index(sd, e)] = (T)0;
        cx.set_nan(op.dst, false);
        wave_sync();
        return;
    }
    const T c = (T)0.7071067811865476;
    int len = ss.len;
    for (int l = 0; l < level; ++l) {
        const DSP_PROG DevSlot& in = (l == 0) ? ss : st;
        const bool last = (l == level - 1);
        const DSP_PROG DevSlot& out = last ? sd : st;
        const int half = (len + 1) >> 1;
        const T f0 = (last && part == 'd') ? -c : c;
        // NB rounds of 64 outputs at a time: their 2 * NB * 64 inputs are all read before any of their outputs is written (the
        // outputs of rounds r .. r+NB-1 land below sample 64 (r + NB), the unread inputs start at 128 (r + NB): in-place halving is
        // safe), so the LDS latency is paid once per batch instead of once per round
        constexpr int NB = 8;
        for (int k0 = 0; k0 < half; k0 += 64 * NB) {
            T lo[NB], hi[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int k = k0 + b * 64 + lane;
                lo[b] = hi[b] = (T)0;
                if (k < half) {
                    lo[b] = cx.lds[padded_index(in, 2 * k)];
                    hi[b] = cx.lds[padded_index(in, (2 * k + 1 < len) ? 2 * k + 1 : len - 1)];
                }
            }
            wave_sync();  // all reads of this batch precede its writes
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int k = k0 + b * 64 + lane;
                if (k < half) cx.lds[padded_index(out, k)] = (T)(f0 * hi[b]) + (T)(c * lo[b]);
            }
        }
        wave_sync();
        len = half;
    }
    // keep the pad of dst finite
    for (int e = sd.len + lane; e < 64 * sd.C; e += 64) cx.lds[padded_index(sd, e)] = (T)0;
    cx.set_nan(op.dst, false);
    wave_sync();
}

// dst[k] = src[k + ip[0]]
template <typename T>
__device__ __forceinline__ void op_copy(Ctx<T>& cx, const DSP_PROG DevOp& op) {
    const DSP_PROG DevSlot& ss = cx.prog->slots[op.src];
    const DSP_PROG DevSlot& sd = cx.prog->slots[op.dst];
    if (cx.slot_all_nan(op.src)) {
        cx.set_nan(op.dst, true);
        return;
    }
    const int total = 64 * sd.C, lane = lane_id(), step = op.ip[1] != 0 ? op.ip[1] : 1;  // (a negative step walks the source backwards)
    if (!cx.slot_nan(op.src)) {
        // (eight loads, then eight stores -- written out, because through the one LDS pointer the compiler keeps every load behind the
        // store before it; total is a multiple of 512)
        for (int e0 = lane; e0 < total; e0 += 512) {
            T v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int e = e0 + 64 * k, se = e * step + op.ip[0];
                const bool ok = e < sd.len && se >= 0 && se < ss.len;
                const T w = cx.lds[padded_index(ss, ok ? se : 0)];
                v[k] = ok ? w : (T)0;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) cx.lds[padded_index(sd, e0 + 64 * k)] = v[k];
        }
        cx.set_nan(op.dst, false);
    } else {  // a slice is a view: of a waveform with NaN samples it holds the ones inside it (rare: one element at a time)
        bool nan = false;
#pragma unroll 1
        for (int e = lane; e < total; e += 64) {
            const int se = e * step + op.ip[0];
            const T v = (e < sd.len && se >= 0 && se < ss.len) ? cx.lds[padded_index(ss, se)] : (T)0;
            nan |= (v != v);
            cx.lds[padded_index(sd, e)] = v;
        }
        if (wave_any(nan))
            cx.set_some_nan(op.dst);
        else
            cx.set_nan(op.dst, false);
    }
    wave_sync();
}

// ------------------------------------------------------------------------------------------------
// convolve_wf / fft_convolve_wf  (processors/convolutions.py:14-72, :75-119): direct-form FIR
//   out[o] = sum_k w[o + start - k] * kern[k]   (np.convolve flips the kernel);  taps are wave-uniform -> scalar loads;
//   each lane produces R consecutive outputs from a sliding register window, so one LDS read feeds R FMAs.
// Accumulation: float32 FMA over blocks of 64 taps, block sums added into a float64 total (keeps the error well inside the
// 1e-6-of-peak bar; NumPy's own float32 summation order is library-internal).
// ic[0] = start offset in the 'full' convolution, ic[1] = kernel length, ic[2] = taps contain NaN
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }

// AMAX: fused with numpy.amax over the output (DSP_OP_CONVOLVE_AMAX): nothing is stored, sreg[dst] receives the maximum
template <typename T>
__device__ __forceinline__ void op_convolve(Ctx<T>& cx, const DSP_PROG DevOp& op, const bool AMAX) {  // (one body for both: the tap loop is the big part)
    const DSP_PROG DevSlot& ss = cx.prog->slots[op.src];
    const DSP_PROG DevSlot& sd = cx.prog->slots[AMAX ? op.src : op.dst];  // (unused when AMAX)
    if (cx.slot_nan(op.src) || op.ic[2]) {
        if (AMAX) {
            if (lane_id() == 0) cx.sregs()[op.dst] = quiet_nan<T>();
            wave_sync();
        } else {
            cx.set_nan(op.dst, true);
        }
        return;
    }
    T vmax = -__builtin_huge_val();
    bool vnan = false;
    constexpr int R = 5;   // consecutive outputs per lane (odd: the lanes' windows start 5 elements apart -> conflict-free reads)
    constexpr int U = 16;  // taps per block of the fast path
    constexpr int W = R + U - 1;
    const DSP_GLOBAL T* __restrict__ kern = cx.template io_ptr<const T>(op.io);
    const int n = ss.len, m = op.ic[1], start = op.ic[0], p = op.ic[3], lane = lane_id();
    // the binding may hold zeros after the m taps (to a multiple of the tap block): the blocked path then runs over them instead of leaving
    // the last m % 16 taps to the tap-by-tap path -- unless the waveform holds an infinity, which a zero tap would turn into a NaN
    int m_blk = m;
    if (op.ic[6] > m) {
        const auto* pc = cx.chunk(ss);
        bool nonfinite = false;
#pragma unroll 8
        for (int t = 0; t < ss.C; ++t) {
            const T v = pc[t];
            nonfinite |= !((v - v) == (T)0);
        }
        if (!wave_any(nonfinite)) m_blk = op.ic[6];
    }
    const bool linear = ss.padw == 0;  // the host lays FIR inputs out without chunk pads: a window is R + U - 1 consecutive elements
    const auto* x0 = cx.lds + ss.off;
    for (int o0 = 0; o0 < p; o0 += 64 * R) {
        const int ob_true = o0 + lane * R;  // first output of this lane
        // lanes past the end redo the last R outputs (discarded): their windows then stay inside the waveform like everyone's
        const int ob = (ob_true + R > p && p >= R) ? p - R : ob_true;
        double tot[R];
        T acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            tot[r] = 0.0;
            acc[r] = (T)0;
        }
        auto flush = [&]() {  // every 64 taps: block sums (T) into the float64 totals
#pragma unroll
            for (int r = 0; r < R; ++r) {
                tot[r] += (double)acc[r];
                acc[r] = (T)0;
            }
        };
        auto slow_taps = [&](int ka, int kb) {  // taps [ka, kb) one by one, bounds checked (edges of the 'full' and 'same' modes)
            for (int k = ka; k < kb; ++k) {
                const T kv = kern[k];
#pragma unroll
                for (int r = 0; r < R; ++r) {  // (a sample outside the waveform is left out, not multiplied as a zero: 0 * inf is NaN)
                    const int idx = ob + r + start - k;
                    if (idx >= 0 && idx < n) acc[r] = fma_t(cx.lds[padded_index(ss, idx)], kv, acc[r]);
                }
                if ((k & 63) == 63) flush();
            }
        };
        // tap k pairs with input index (ob + r + start - k).  A block of U taps starting at k0 touches inputs
        // [ob + start - k0 - (U-1), ob + (R-1) + start - k0]: inside the waveform for every lane iff kA <= k0 <= kB
        // (the slot reads 0 for zero_below elements under sample 0 and zero_above over sample n - 1: exactly what the reference's
        // zero padding of the 'same' and 'full' modes supplies, so windows may reach that far)
        // (ob grows with the lane: its extremes sit in lanes 63 and 0 -- two readlanes instead of two wavefront reductions; SGPRs: tap
        // addresses stay scalar)
        int kA = __builtin_amdgcn_readlane(ob, 63) + (R - 1) + start - (n - 1) - (op.ic[5] ? 0 : ss.zero_above);
        int kB = __builtin_amdgcn_readfirstlane(ob) + start - (U - 1) + (op.ic[5] ? 0 : ss.zero_below);
        kA = kA < 0 ? 0 : ((kA + U - 1) / U) * U;
        if (kB > m_blk - U) kB = m_blk - U;
        if ((!linear && ss.C < 2 * U) || kB < kA) {
            slow_taps(0, m);
        } else {
            slow_taps(0, kA);
            // ---- fast blocks: R + U - 1 inputs at immediate offsets, U taps (one broadcast 16-byte load each 4), R * U FMAs in tap
            // order; the next block's loads are issued before this block's arithmetic
            T w[2][W], h[2][U];
            // (taps as scalar loads -- constant address space, SGPR operands -- were measured slower: v_pk_fma_f32 wants its
            // multiplier pair in VGPRs, so every tap was moved back; the 16-byte vector loads below hit one cache line per wave)
            auto load_block_as = [&](auto lin, int k0, T (&wb)[W], T (&hb)[U]) {  // lin: the layout as a compile-time constant
                const int i0 = ob + start - k0 - (U - 1);  // first sample of the window
                if (decltype(lin)::value) {
                    const auto* base = x0 + i0;
#pragma unroll
                    for (int j = 0; j < W; ++j) wb[j] = base[j];
                } else {
                    // chunk-padded input (a waveform the recursive filters also work on): the W samples are W + 1 consecutive LDS
                    // elements minus the one pad the window may cross (C >= 2U > W: at most one) -- read them all at immediate
                    // offsets, then drop the pad with one select per sample
                    const int q = (int)(((float)i0 + 0.5f) * ss.invC);
                    const int jc = (q + 1) * ss.C - i0;  // first j that lies in the next chunk
                    const auto* base = x0 + (i0 + q);
                    T ph[W + 1];
#pragma unroll
                    for (int j = 0; j <= W; ++j) ph[j] = base[j];
#pragma unroll
                    for (int j = 0; j < W; ++j) wb[j] = j < jc ? ph[j] : ph[j + 1];
                }
#pragma unroll
                for (int u = 0; u < U; ++u) hb[u] = kern[k0 + u];
            };
            auto load_block = [&](int k0, T (&wb)[W], T (&hb)[U]) {
                if (linear)
                    load_block_as(std::true_type{}, k0, wb, hb);
                else
                    load_block_as(std::false_type{}, k0, wb, hb);
            };
            auto fma_block = [&](const T (&wb)[W], const T (&hb)[U]) {
                // (five independent sums of plain FMAs: packing outputs pairwise into v_pk_fma_f32 -- explicitly, also with even / odd
                // taps in separate sums -- measured 18 % slower: the pair shuffles and the fewer chains cost more than the packing saves)
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int r = 0; r < R; ++r) acc[r] = fma_t(wb[U - 1 - u + r], hb[u], acc[r]);
            };
            const int nblk = (kB - kA) / U + 1;
            load_block(kA, w[0], h[0]);
            int k0 = kA;
            int b = 0;
            // steady state: groups of four blocks (64 taps, one flush) as straight-line code -- every block's loads are issued one
            // block ahead and nothing in the body branches, so the waits the compiler inserts can leave the newest loads in flight
            // (with branches in the body it waits for everything, which defeats the prefetch).  Entered on a 64-tap boundary with at
            // least five blocks left, so the prefetch of "the next block" is always a real block.
            auto groups = [&](auto lin) {
                for (; b + 5 <= nblk; b += 4) {
                    load_block_as(lin, k0 + U, w[1], h[1]);
                    fma_block(w[0], h[0]);
                    load_block_as(lin, k0 + 2 * U, w[0], h[0]);
                    fma_block(w[1], h[1]);
                    load_block_as(lin, k0 + 3 * U, w[1], h[1]);
                    fma_block(w[0], h[0]);
                    load_block_as(lin, k0 + 4 * U, w[0], h[0]);
                    fma_block(w[1], h[1]);
                    flush();
                    k0 += 4 * U;
                }
            };
            if constexpr (sizeof(T) == 4) {  // (the float64 build is at its register limit without the second copy of the loop)
                if ((k0 & 63) == 0) {
                    if (linear)
                        groups(std::true_type{});
                    else
                        groups(std::false_type{});
                }
            }
            for (; b + 1 < nblk; b += 2) {
                load_block(k0 + U, w[1], h[1]);
                fma_block(w[0], h[0]);
                if (((k0 + U) & 63) == 0) flush();
                if (b + 2 < nblk) load_block(k0 + 2 * U, w[0], h[0]);
                fma_block(w[1], h[1]);
                if (((k0 + 2 * U) & 63) == 0) flush();
                k0 += 2 * U;
            }
            if (b < nblk) {
                fma_block(w[0], h[0]);
                if (((k0 + U) & 63) == 0) flush();
                k0 += U;
            }
            slow_taps(k0, m);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int o = ob + r;
            if (o >= ob_true && o < p) {
                const T v = (T)(tot[r] + (double)acc[r]);
                if (AMAX) {
                    vnan |= (v != v);
                    vmax = v > vmax ? v : vmax;
                } else {
                    cx.lds[padded_index(sd, o)] = v;
                }
            }
        }
    }
    if (AMAX) {
#pragma unroll
        for (int msk = 1; msk < 64; msk <<= 1) {
            const T other = __shfl_xor(vmax, msk);
            vmax = other > vmax ? other : vmax;
        }
        if (lane == 0) cx.sregs()[op.dst] = wave_any(vnan) ? quiet_nan<T>() : vmax;  // numpy.amax propagates NaN
    } else {
        for (int e = p + lane; e < 64 * sd.C; e += 64) cx.lds[padded_index(sd, e)] = (T)0;
        cx.set_nan(op.dst, false);
    }
    wave_sync();
}

template <typename T>
__device__ __forceinline__ void op_scalar_affine(Ctx<T>& cx, const DSP_PROG DevOp& op) {
    const T a = cx.scalar(op.sp[0]), b = cx.scalar(op.sp[1]), c = cx.scalar(op.sp[2]);
    if (lane_id() == 0) cx.sregs()[op.dst] = a * b + c;
    wave_sync();
}

// numpy.true_divide between two per-event variables (processing_chain.py:832-891 adds the ufunc as a processor)
template <typename T>
__device__ __forceinline__ void op_scalar_div(Ctx<T>& cx, const DSP_PROG DevOp& op) {
    const T a = cx.scalar(op.sp[0]), b = cx.scalar(op.sp[1]);
    if (lane_id() == 0) cx.sregs()[op.dst] = a / b;
    wave_sync();
}

// A time coordinate moved from one CoordinateGrid to another (unit_conversion.py:16-79): the offsets and the period ratio are float64
// arguments there, so the arithmetic is float64 whatever the loop type; the result takes the variable's type.
template <typename T>
__device__ __forceinline__ void op_scalar_convert(Ctx<T>& cx, const DSP_PROG DevOp& op) {
    auto f64_of = [&](const DSP_PROG dsp_scalar_arg& a) { return a.kind == DSP_ARG_CONST ? a.value : (double)cx.scalar(a); };
    const double x = (double)cx.scalar(op.sp[0]), off_in = f64_of(op.sp[1]), off_out = f64_of(op.sp[2]), ratio = op.sp[3].value;
    double r = (x + off_in) * ratio;  // (separate roundings like the reference's expression: no contraction into an fma)
    asm volatile("" : "+v"(r));
    r = r - off_out;
    const int mode = op.ip[0];
    if (mode == 1) r = __builtin_rint(r);
    else if (mode == 2) r = __builtin_floor(r);
    else if (mode == 3) r = __builtin_ceil(r);
    else if (mode == 4) r = __builtin_trunc(r);
    if (lane_id() == 0) cx.sregs()[op.dst] = (T)r;
    wave_sync();
}

// host-inserted: a slot that shares its LDS region with others starts from the all-zero state the kernel prologue gives the rest
template <typename T>
__device__ __forceinline__ void op_zero_region(Ctx<T>& cx, const DSP_PROG DevOp& op) {
    const int base = op.ic[0], n = op.ic[1];  // (both multiples of 4 elements: 16-byte stores)
    typedef T vec4_t __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) vec4_t lds_vec4;
    auto* p4 = (lds_vec4*)(cx.lds + base);
    const vec4_t z = {(T)0, (T)0, (T)0, (T)0};
#pragma unroll 4
    for (int e = lane_id(); e < n / 4; e += 64) p4[e] = z;
    cx.set_nan(op.dst, false);
    wave_sync();
}

// ------------------------------------------------------------------------------------------------
// the interpreter
// ------------------------------------------------------------------------------------------------
// FIR: the interpreter with the long-FIR op compiled in.  That op's register needs (183 VGPRs with it, 113 without) would cap every
// chain at 2 wavefronts per SIMD, so programs without a CONVOLVE run the lean build: 4 wavefronts per SIMD where LDS allows.
template <typename T, bool FIR, int TEAM>
// (the build without the long-FIR op: 3 wavefronts per SIMD = 168 registers.  At 4 per SIMD -- 128 registers -- the interpreter spilled in
// every op once the per-event pole-zero forms and the recipe language's pick-off modes were in: C2 on the VM 129 -> 106 M waveforms/s; and a
// waveform of 4096 samples leaves LDS for 9 wavefronts per CU anyway)
__global__ void __launch_bounds__(256 * TEAM, TEAM > 1 ? 1 : (FIR ? 2 : 3)) dsp_vm_kernel(const DevProgram* __restrict__ prog, IoPtrs ptrs, int64_t n_wf, int* err) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // TEAM == 2: wavefronts 2 s and 2 s + 1 of the workgroup share row slot s -- one LDS image, the ops of the row's program dealt out between
    // them (DevOp.member; the LOAD is shared: each member loads every other batch, a barrier makes the image whole).  One more
    // workgroup barrier per row keeps a member from loading the next row into an image its partner still reads.
    const int wave_raw = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // provably wave-uniform -> rows, pointers in SGPRs
    const int wave = TEAM == 1 ? wave_raw : wave_raw / TEAM;
    const int member = TEAM == 1 ? 0 : wave_raw % TEAM;
    const int wpb = (int)(blockDim.x >> 6) / TEAM;
    auto* lds = (typename Ctx<T>::LT*)smem_raw + (size_t)wave * prog->lds_elems_per_wave;  // (C cast: generic -> LDS address space)
    // zero the whole region once: guards below each slot must read as 0 forever, pads start finite
    for (int e = lane_id(); e < prog->lds_elems_per_wave; e += 64) lds[e] = (T)0;
    if (TEAM == 1)
        wave_sync();
    else
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (the partner may still be clearing what this member would load into)

    Ctx<T> cx;
    cx.lds = lds;
    cx.prog = (const DSP_PROG DevProgram*)prog;
    // kernel arguments: (const DevProgram*, IoPtrs, int64_t, int*) -> the pointer table starts 8 bytes into the segment
    static_assert(sizeof(const DevProgram*) == 8 && alignof(IoPtrs) == 8, "kernel-argument layout");
    cx.kptrs = (const __attribute__((address_space(4))) uint64_t*)__builtin_amdgcn_kernarg_segment_ptr() + 1;
    (void)ptrs;
    cx.err = err;
    const int64_t total_waves = (int64_t)gridDim.x * wpb;
    const int n_ops = prog->n_ops;
    // dsp_chain_profile: the first wavefront of every workgroup times each op of its waveforms with the shader clock
    unsigned long long* prof = prog->prof;
    // (of a team both members of the workgroup's first row slot are sampled: each closes an interval into the op IT ran last, so the cycles of
    // an op land on that op whichever member ran it; an op both run -- the LOAD -- collects both members' cycles)
    const bool sampled = prof != nullptr && wave_raw < TEAM;
    unsigned long long t_prev = 0;
    int prof_last = -1;
    for (int64_t base = (int64_t)blockIdx.x * wpb;; base += total_waves) {
        const int64_t row = base + wave;
        if (TEAM == 1 ? row >= n_wf : base >= n_wf) break;  // (a team's workgroup leaves together: every wavefront meets every barrier)
        if (TEAM > 1 && row >= n_wf) {  // (an idle row slot of the last round meets the barriers of a row: the one at its end and the load's two)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
            continue;
        }
        cx.row = row;
        cx.nan_all = cx.nan_some = 0;
        if (sampled) {
            t_prev = __builtin_amdgcn_s_memtime();
            prof_last = -1;
        }
        // wave priority rises with the progress through the row's program (levels 0..3 over the op list): of the wavefronts that share a
        // SIMD the one closest to finishing its row issues first, the others' loads and stores fill its gaps -- the arbitration that took the
        // specialised energy kernel from 63 % to 69 % of the HBM peak (dsp_energy.hip); C2 on this interpreter 137 -> 148 M waveforms/s
        int prio_level = 0;
        __builtin_amdgcn_s_setprio(0);
        for (int i = 0; i < n_ops; ++i) {
            const DSP_PROG DevOp& op = cx.prog->ops[i];
            if (TEAM > 1) {
                const int m = op.member;  // (uniform)
                if (m != DSP_MEMBER_ALL && m != member) continue;
            }
            {
                const int lvl = op.prio;
                if (lvl != prio_level) {  // (uniform; s_setprio takes an immediate)
                    prio_level = lvl;
                    if (lvl == 1) __builtin_amdgcn_s_setprio(1);
                    else if (lvl == 2) __builtin_amdgcn_s_setprio(2);
                    else __builtin_amdgcn_s_setprio(3);
                }
            }
            if (sampled) {  // close the interval of the op this wavefront ran last (one s_memtime per op when profiling, none otherwise)
                const unsigned long long now = __builtin_amdgcn_s_memtime();
                if (prof_last >= 0 && lane_id() == 0) atomicAdd(prof + prof_last, now - t_prev);
                t_prev = now;
                prof_last = i;
            }
            if constexpr (TEAM > 1) {
                // a team's program holds nothing but these (dsp_plan.cpp: what makes a program eligible): the build is a fifth of the full
                // interpreter's 280 kB -- the instruction cache (64 kB per two CUs) sees eight wavefronts of a CU in different ops
                switch (op.opcode) {
                    case DSP_OP_LOAD: op_load<T, TEAM>(cx, op, member); break;
                    case DSP_OP_STORE_SCALAR: op_store_scalar(cx, op); break;
                    case DSP_OP_INTERNAL_STORES: op_store_scalars(cx, op); break;
                    case DSP_OP_PICKOFF: op_pickoff(cx, op); break;
                    case DSP_OP_TRAP_PICKOFF: op_trap_pickoff(cx, op); break;
                    case DSP_OP_TRAP_REDUCE: op_trap_reduce(cx, op); break;
                    case DSP_OP_TIME_POINT_THRESH: op_time_point_thresh(cx, op); break;
                    case DSP_OP_MIN_MAX: op_min_max(cx, op, false); break;
                    case DSP_OP_AMAX: op_min_max(cx, op, true); break;
                    case DSP_OP_SCALAR_AFFINE: op_scalar_affine(cx, op); break;
                    case DSP_OP_SCALAR_DIV: op_scalar_div(cx, op); break;
                    case DSP_OP_SCALAR_FUNC: op_scalar_func(cx, op); break;
                    case DSP_OP_SCALAR_CONVERT: op_scalar_convert(cx, op); break;
                    default: break;
                }
            } else
            switch (op.opcode) {
                case DSP_OP_LOAD: op_load<T, TEAM>(cx, op, member); break;
                case DSP_OP_STORE: op_store(cx, op); break;
                case DSP_OP_STORE_SCALAR: op_store_scalar(cx, op); break;
                case DSP_OP_INTERNAL_STORES: op_store_scalars(cx, op); break;
                case DSP_OP_BL_SUBTRACT: op_bl_subtract(cx, op); break;
                case DSP_OP_MIN_MAX_NORM: op_min_max_norm(cx, op); break;
                case DSP_OP_POLE_ZERO: op_pole_zero(cx, op); break;
                case DSP_OP_DOUBLE_POLE_ZERO: op_double_pole_zero(cx, op); break;
                case DSP_OP_TRAP_FILTER:
                case DSP_OP_TRAP_NORM:
                case DSP_OP_ASYM_TRAP: op_trap(cx, op); break;
                case DSP_OP_PICKOFF: op_pickoff(cx, op); break;
                case DSP_OP_TRAP_PICKOFF: op_trap_pickoff(cx, op); break;
                case DSP_OP_TRAP_REDUCE: op_trap_reduce(cx, op); break;
                case DSP_OP_UPSAMPLER: op_upsampler(cx, op); break;
                case DSP_OP_LINEAR_SLOPE_FIT: op_linear_slope_fit(cx, op); break;
                case DSP_OP_MOVING_WINDOW_MULTI: op_moving_window_multi(cx, op); break;
                case DSP_OP_TIME_POINT_THRESH: op_time_point_thresh(cx, op); break;
                case DSP_OP_INTERP_TIME_POINT_THRESH: op_interp_time_point_thresh(cx, op); break;
                case DSP_OP_MIN_MAX: op_min_max(cx, op, false); break;
                case DSP_OP_AMAX: op_min_max(cx, op, true); break;
                case DSP_OP_MEAN_BELOW: op_mean_below(cx, op); break;
                case DSP_OP_WINDOWER: op_windower(cx, op); break;
                case DSP_OP_AVG_CURRENT: op_avg_current(cx, op); break;
                case DSP_OP_TRAP_WINDOW_PICKOFF: op_trap_window_pickoff(cx, op); break;
                case DSP_OP_DWT_HAAR: op_dwt_haar(cx, op); break;
                case DSP_OP_COPY: op_copy(cx, op); break;
                case DSP_OP_CONVOLVE:
                case DSP_OP_CONVOLVE_AMAX:
                    if constexpr (FIR) op_convolve<T>(cx, op, op.opcode == DSP_OP_CONVOLVE_AMAX);
                    break;
                case DSP_OP_SCALAR_AFFINE: op_scalar_affine(cx, op); break;
                case DSP_OP_SCALAR_DIV: op_scalar_div(cx, op); break;
                case DSP_OP_ELEMENTWISE: op_elementwise(cx, op); break;
                case DSP_OP_SCALAR_FUNC: op_scalar_func(cx, op); break;
                case DSP_OP_SCALAR_CONVERT: op_scalar_convert(cx, op); break;
                case DSP_OP_INTERNAL_ZERO: op_zero_region(cx, op); break;
                case DSP_OP_INTERNAL_NOP: break;
                default: break;
            }
        }
        if (sampled) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            if (lane_id() == 0) {
                if (prof_last >= 0) atomicAdd(prof + prof_last, now - t_prev);
                if (member == 0) atomicAdd(prof + n_ops, 1ull);
            }
        }
        if (TEAM > 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the image is free for the next row
    }
}

// ------------------------------------------------------------------------------------------------
// synthetic batch generator (bench.py): counter-based, so any row can be regenerated
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float u01(uint32_t h) { return (float)(h >> 8) * (1.0f / 16777216.0f); }

template <typename OutT>
__global__ void __launch_bounds__(256) dsp_synth_kernel(OutT* wf, int64_t n_wf, int wf_len, int64_t row_stride, float* baseline,
                                                        float* t_pick, uint32_t seed_lo, uint32_t seed_hi, int64_t first_row,
                                                        float inv_tau, float sigma, float pick_offset, float bl_lo, float bl_hi,
                                                        float amp_lo, float amp_hi, float rise_lo, float rise_hi) {
    for (int64_t r = blockIdx.x; r < n_wf; r += gridDim.x) {
        const uint64_t gr = (uint64_t)(first_row + r);
        const uint32_t k0 = mix32((uint32_t)gr ^ seed_lo), k1 = mix32((uint32_t)(gr >> 32) ^ seed_hi ^ k0);
        const float B = bl_lo + (bl_hi - bl_lo) * u01(mix32(k1 ^ 0x1234567u));
        const float A = amp_lo + (amp_hi - amp_lo) * u01(mix32(k1 ^ 0x89abcdeu));
        const float t0 = floorf((0.45f + 0.10f * u01(mix32(k1 ^ 0x5555aaau))) * (float)wf_len);
        // charge collection: the step reaches its height over `rise` samples, linearly (rise_hi <= 1: within one sample, as dsp_synth_waveforms)
        const float rise = rise_hi > 1.0f ? rise_lo + (rise_hi - rise_lo) * u01(mix32(k1 ^ 0x3141592u)) : 1.0f;
        const float inv_rise = 1.0f / rise;
        if (threadIdx.x == 0) {
            if (baseline) baseline[r] = B;
            if (t_pick) t_pick[r] = t0 + pick_offset;
        }
        for (int i = threadIdx.x; i < wf_len; i += blockDim.x) {
            // sum of four uniforms: variance 4/12, rescaled to unit variance
            uint32_t h = mix32(k1 + 0x9e3779b9u * (uint32_t)(i + 1));
            const float n = ((u01(h) + u01(mix32(h ^ 0xa5a5a5a5u)) + u01(mix32(h + 0x3c6ef372u)) + u01(mix32(h ^ 0x1b873593u))) - 2.0f) *
                            1.7320508f;
            const float d = (float)i - t0;
            float v = B + sigma * n;
            if (d >= 0.0f) v += A * __expf(-d * inv_tau) * fminf((d + 1.0f) * inv_rise, 1.0f);
            if (sizeof(OutT) == 2)
                wf[r * row_stride + i] = (OutT)__float2int_rn(v);
            else
                wf[r * row_stride + i] = (OutT)v;
        }
    }
}

// read-only streaming pass (measurement only): the HBM ceiling bench.py reports beside the energy kernel.  Same access shape as the
// energy kernel's prefetch: one wavefront pulls `row_bytes` contiguous bytes (16 bytes per lane and load) per row.
__global__ void __launch_bounds__(256) dsp_stream_read_kernel(const uint4* src, int64_t n_vec, uint32_t* sink) {
    uint32_t acc = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x * 4 + threadIdx.x; i < n_vec; i += stride) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t j = i + (int64_t)u * blockDim.x;
            if (j < n_vec) {
                const uint4 v = src[j];
                acc ^= v.x ^ v.y ^ v.z ^ v.w;
            }
        }
    }
    if (acc == 0x9e3779b9u) sink[0] = acc;  // (keeps the loads alive; practically never true)
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// launchers (called from dsp_host.cpp)
// ------------------------------------------------------------------------------------------------
extern "C" int dsp_internal_launch_vm_f32(const DevProgram* dev_prog, const IoPtrs* ptrs, int64_t n_wf, int* err, int blocks,
                                          int threads, int lds_bytes, int with_fir, int team, hipStream_t stream) {
    if (team == 3)
        hipLaunchKernelGGL((dsp_vm_kernel<float, false, 3>), dim3(blocks), dim3(threads), lds_bytes, stream, dev_prog, *ptrs, n_wf, err);
    else if (team == 2)
        hipLaunchKernelGGL((dsp_vm_kernel<float, false, 2>), dim3(blocks), dim3(threads), lds_bytes, stream, dev_prog, *ptrs, n_wf, err);
    else if (with_fir)
        hipLaunchKernelGGL((dsp_vm_kernel<float, true, 1>), dim3(blocks), dim3(threads), lds_bytes, stream, dev_prog, *ptrs, n_wf, err);
    else
        hipLaunchKernelGGL((dsp_vm_kernel<float, false, 1>), dim3(blocks), dim3(threads), lds_bytes, stream, dev_prog, *ptrs, n_wf, err);
    return (int)hipGetLastError();
}

extern "C" int dsp_internal_launch_vm_f64(const DevProgram* dev_prog, const IoPtrs* ptrs, int64_t n_wf, int* err, int blocks,
                                          int threads, int lds_bytes, int with_fir, hipStream_t stream) {
    if (with_fir)
        hipLaunchKernelGGL((dsp_vm_kernel<double, true, 1>), dim3(blocks), dim3(threads), lds_bytes, stream, dev_prog, *ptrs, n_wf, err);
    else
        hipLaunchKernelGGL((dsp_vm_kernel<double, false, 1>), dim3(blocks), dim3(threads), lds_bytes, stream, dev_prog, *ptrs, n_wf, err);
    return (int)hipGetLastError();
}

extern "C" int dsp_internal_set_vm_lds(int lds_bytes) {
    const void* k[6] = {reinterpret_cast<const void*>(&dsp_vm_kernel<float, true, 1>), reinterpret_cast<const void*>(&dsp_vm_kernel<float, false, 1>),
                        reinterpret_cast<const void*>(&dsp_vm_kernel<double, true, 1>), reinterpret_cast<const void*>(&dsp_vm_kernel<double, false, 1>),
                        reinterpret_cast<const void*>(&dsp_vm_kernel<float, false, 2>), reinterpret_cast<const void*>(&dsp_vm_kernel<float, false, 3>)};
    for (int i = 0; i < 6; ++i) {
        const int rc = (int)hipFuncSetAttribute(k[i], hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (rc != 0) return rc;
    }
    return 0;
}

extern "C" const char* dsp_internal_vm_kernel_name() { return "dsp_vm_kernel<float>"; }

extern "C" int dsp_internal_launch_synth(void* wf, int out_dtype, int64_t n_wf, int wf_len, int64_t row_stride, float* baseline,
                                         float* t_pick, uint64_t seed, int64_t first_row, float tau, float sigma, float pick_offset,
                                         float bl_lo, float bl_hi, float amp_lo, float amp_hi, float rise_lo, float rise_hi, hipStream_t stream) {
    const int blocks = (int)(n_wf < 8192 ? (n_wf > 0 ? n_wf : 1) : 8192);
    const uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
    if (out_dtype == DSP_I16)
        hipLaunchKernelGGL(dsp_synth_kernel<int16_t>, dim3(blocks), dim3(256), 0, stream, (int16_t*)wf, n_wf, wf_len, row_stride,
                           baseline, t_pick, lo, hi, first_row, 1.0f / tau, sigma, pick_offset, bl_lo, bl_hi, amp_lo, amp_hi, rise_lo, rise_hi);
    else
        hipLaunchKernelGGL(dsp_synth_kernel<float>, dim3(blocks), dim3(256), 0, stream, (float*)wf, n_wf, wf_len, row_stride, baseline,
                           t_pick, lo, hi, first_row, 1.0f / tau, sigma, pick_offset, bl_lo, bl_hi, amp_lo, amp_hi, rise_lo, rise_hi);
    return (int)hipGetLastError();
}

extern "C" int dsp_internal_launch_stream_read(const void* src, int64_t bytes, uint32_t* sink, int blocks, hipStream_t stream) {
    hipLaunchKernelGGL(dsp_stream_read_kernel, dim3(blocks), dim3(256), 0, stream, (const uint4*)src, bytes / 16, sink);
    return (int)hipGetLastError();
}
