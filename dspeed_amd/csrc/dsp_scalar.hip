// dsp_scalar.hip -- programs that only do arithmetic between per-event values and store them, one ROW per LANE.
//
// A whole recipe ends in dozens of such ops: thresholds scaled, times converted from sample indices to the units their columns are
// written in, results stored (the Ge recipe: 49 of the main program's 71 ops).  On the waveform VM a wavefront owns ONE row: every such op
// is an interpreter dispatch, an LDS round trip for the register file and one lane's worth of arithmetic -- 500-700 cycles per op and row,
// a quarter of the recipe's main program.  The recipe builder therefore cuts the all-scalar tail off a program (processing_chain.py,
// _split_scalar_tail); the head hands the registers the tail reads over as columns, and this kernel runs the tail with a row per lane:
// 64 rows per interpreter dispatch, registers in a lane-strided LDS file, the same expressions in the same type as the VM's ops
// (dsp_vm.hip: op_scalar_affine / _div / _convert / _func, op_store_scalar; reference: each is one NumPy ufunc processor,
// processing_chain.py:832-947, or a coordinate conversion, unit_conversion.py:16-79).  Compiled with -ffp-contract=off.
#include <hip/hip_runtime.h>

#include "dsp_program.h"
#include "dsp_wave.h"

#define SC_PROG __attribute__((address_space(4)))
#define SC_GLOBAL __attribute__((address_space(1)))

namespace {

template <typename T>
__device__ __forceinline__ T sc_apply(int ip0, T a, T b, T c) {
    const int fn = DSP_FN_CODE(ip0);
    if (fn >= DSP_FN_IADD && fn <= DSP_FN_ICAST) return int_loop_apply<T>(fn, a, b, ip0);  // (the integer loops: dsp_wave.h)
    switch (fn) {
        case DSP_FN_ADD: return a + b;
        case DSP_FN_SUB: return a - b;
        case DSP_FN_MUL: return a * b;
        case DSP_FN_DIV: return a / b;
        case DSP_FN_LT: return (T)(a < b);
        case DSP_FN_LE: return (T)(a <= b);
        case DSP_FN_GT: return (T)(a > b);
        case DSP_FN_GE: return (T)(a >= b);
        case DSP_FN_EQ: return (T)(a == b);
        case DSP_FN_NE: return (T)(a != b);
        case DSP_FN_WHERE: return a != (T)0 ? b : c;
        case DSP_FN_ISNAN: return (T)(a != a);
        case DSP_FN_ISFINITE: return (T)((a - a) == (T)0);
        case DSP_FN_NEG: return -a;
        case DSP_FN_FLOORDIV: return np_floor_divide<T>(a, b);
        case DSP_FN_RINT: return rint(a);
        case DSP_FN_FLOOR: return floor(a);
        case DSP_FN_CEIL: return ceil(a);
        case DSP_FN_TRUNC: return trunc(a);
        case DSP_FN_LOR: return (T)(a != (T)0 || b != (T)0);
        case DSP_FN_LAND: return (T)(a != (T)0 && b != (T)0);
        default: return a;
    }
}

// ---- integer programs (compute_dtype DSP_I64, dspeed_hip.h): NumPy's integer ufunc loops on 64-bit integer registers.  A register holds its
// value sign- or zero-extended from the type of the loop that made it; uint64 values are held as their bit pattern.
__device__ __forceinline__ int64_t sc_wrap(uint64_t r, int meta) {
    const int bits = DSP_FN_INT_BITS(meta);
    if (bits >= 64) return (int64_t)r;
    const uint64_t m = r & ((1ull << bits) - 1ull);
    return (DSP_FN_INT_SIGNED(meta) && ((m >> (bits - 1)) & 1ull)) ? (int64_t)(m | ~((1ull << bits) - 1ull)) : (int64_t)m;
}

template <>
__device__ __forceinline__ int64_t sc_apply<int64_t>(int ip0, int64_t a, int64_t b, int64_t c) {
    const int fn = DSP_FN_CODE(ip0);
    const bool u64 = DSP_FN_INT_BITS(ip0) == 64 && !DSP_FN_INT_SIGNED(ip0);  // the loop's type is uint64: division and order are unsigned
    switch (fn) {
        case DSP_FN_IADD: return sc_wrap((uint64_t)a + (uint64_t)b, ip0);
        case DSP_FN_ISUB: return sc_wrap((uint64_t)a - (uint64_t)b, ip0);
        case DSP_FN_IMUL: return sc_wrap((uint64_t)a * (uint64_t)b, ip0);
        case DSP_FN_IFLOORDIV: {
            if (b == 0) return 0;  // (numpy.floor_divide's integer loops give 0 and a warning)
            if (u64) return (int64_t)((uint64_t)a / (uint64_t)b);
            if (b == -1) return sc_wrap(0ull - (uint64_t)a, ip0);  // (the type's minimum // -1 wraps back to the minimum)
            int64_t q = a / b;
            if ((a % b != 0) && ((a < 0) != (b < 0))) q -= 1;
            return sc_wrap((uint64_t)q, ip0);
        }
        case DSP_FN_ICAST: return sc_wrap((uint64_t)a, ip0);
        case DSP_FN_LT: return u64 ? (int64_t)((uint64_t)a < (uint64_t)b) : (int64_t)(a < b);
        case DSP_FN_LE: return u64 ? (int64_t)((uint64_t)a <= (uint64_t)b) : (int64_t)(a <= b);
        case DSP_FN_GT: return u64 ? (int64_t)((uint64_t)a > (uint64_t)b) : (int64_t)(a > b);
        case DSP_FN_GE: return u64 ? (int64_t)((uint64_t)a >= (uint64_t)b) : (int64_t)(a >= b);
        case DSP_FN_EQ: return (int64_t)(a == b);
        case DSP_FN_NE: return (int64_t)(a != b);
        case DSP_FN_WHERE: return a != 0 ? b : c;
        case DSP_FN_ISNAN: return 0;     // (an integer is never NaN ...
        case DSP_FN_ISFINITE: return 1;  //  ... and always finite: numpy.isnan / isfinite through their float loops)
        case DSP_FN_LOR: return (int64_t)(a != 0 || b != 0);
        case DSP_FN_LAND: return (int64_t)(a != 0 && b != 0);
        default: return a;  // DSP_FN_COPY
    }
}

template <typename T>
__global__ void __launch_bounds__(64) dsp_scalar_kernel(const DevProgram* __restrict__ prog_, IoPtrs ptrs_, int64_t n_wf) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sc_smem[];
    const SC_PROG DevProgram* prog = (const SC_PROG DevProgram*)prog_;
    // I/O pointers straight from the kernel-argument segment (scalar loads with a run-time index; see dsp_vm.hip)
    const __attribute__((address_space(4))) uint64_t* kptrs = (const __attribute__((address_space(4))) uint64_t*)__builtin_amdgcn_kernarg_segment_ptr() + 1;
    (void)ptrs_;
    const int tid = (int)threadIdx.x;
    typedef __attribute__((address_space(3))) T LT;
    LT* regs = (LT*)sc_smem + tid;  // register r of this row at regs[r * 64]: one wavefront per block, a lane-strided register file
    const int64_t row = (int64_t)blockIdx.x * 64 + tid;
    const bool live = row < n_wf;
    const int64_t rowc = live ? row : n_wf - 1;
    const int n_ops = prog->n_ops;
    constexpr bool INT = sizeof(T) == 8 && (T)0.5 == (T)0;  // int64 registers: an integer program
    for (int r = 0; r < prog->n_sregs; ++r) regs[r * 64] = (T)0;  // (the VM's register file starts zeroed too)

    auto column = [&](int io_index) { return (const SC_GLOBAL char*)(uintptr_t)kptrs[io_index]; };
    auto operand = [&](const SC_PROG dsp_scalar_arg& a) -> T {
        if (a.kind == DSP_ARG_CONST) return (T)a.value;
        if (a.kind == DSP_ARG_REG) return regs[a.index * 64];
        const SC_PROG DevIO& io = prog->io[a.index];
        const int64_t at = (int64_t)io.offset + rowc * io.row_stride;
        const SC_GLOBAL char* p = column(a.index);
        switch (io.dtype) {
            case DSP_F32: return (T)((const SC_GLOBAL float*)p)[at];
            case DSP_F64: return (T)((const SC_GLOBAL double*)p)[at];
            case DSP_I32: return (T)((const SC_GLOBAL int32_t*)p)[at];
            case DSP_I16: return (T)((const SC_GLOBAL int16_t*)p)[at];
            case DSP_U16: return (T)((const SC_GLOBAL uint16_t*)p)[at];
            case DSP_BOOL: return (T)(((const SC_GLOBAL uint8_t*)p)[at] != 0);
            case DSP_I64: return (T)((const SC_GLOBAL int64_t*)p)[at];
            case DSP_U64:  // (an integer program keeps the bit pattern; a float loop converts the value, as NumPy's cast does)
                if constexpr (INT) return (T)((const SC_GLOBAL int64_t*)p)[at];
                else return (T)((const SC_GLOBAL uint64_t*)p)[at];
            default: return (T)((const SC_GLOBAL uint32_t*)p)[at];
        }
    };

    for (int i = 0; i < n_ops; ++i) {
        const SC_PROG DevOp& op = prog->ops[i];
        switch (op.opcode) {  // (uniform)
            case DSP_OP_SCALAR_AFFINE: {
                if constexpr (INT) break;  // (float ops: the planner admits none into an integer program)
                const T a = operand(op.sp[0]), b = operand(op.sp[1]), c = operand(op.sp[2]);
                regs[op.dst * 64] = a * b + c;
                break;
            }
            case DSP_OP_SCALAR_DIV: {
                if constexpr (INT) break;
                const T a = operand(op.sp[0]), b = operand(op.sp[1]);
                regs[op.dst * 64] = a / b;
                break;
            }
            case DSP_OP_SCALAR_CONVERT: {  // float64 whatever the loop type (unit_conversion.py:16-79); separate roundings
                if constexpr (INT) break;
                const double x = (double)operand(op.sp[0]);
                const double off_in = op.sp[1].kind == DSP_ARG_CONST ? op.sp[1].value : (double)operand(op.sp[1]);
                const double off_out = op.sp[2].kind == DSP_ARG_CONST ? op.sp[2].value : (double)operand(op.sp[2]);
                double r = (x + off_in) * op.sp[3].value;
                asm volatile("" : "+v"(r));
                r = r - off_out;
                const int mode = op.ip[0];
                if (mode == 1) r = __builtin_rint(r);
                else if (mode == 2) r = __builtin_floor(r);
                else if (mode == 3) r = __builtin_ceil(r);
                else if (mode == 4) r = __builtin_trunc(r);
                regs[op.dst * 64] = (T)r;
                break;
            }
            case DSP_OP_SCALAR_FUNC: {
                const T a = operand(op.sp[0]), b = operand(op.sp[1]), c = operand(op.sp[2]);
                regs[op.dst * 64] = sc_apply<T>(op.ip[0], a, b, c);
                break;
            }
            case DSP_OP_STORE_SCALAR: {
                const SC_PROG DevIO& io = prog->io[op.io];
                const T v = regs[op.ip[0] * 64];
                if (live) {
                    SC_GLOBAL char* p = (SC_GLOBAL char*)(uintptr_t)kptrs[op.io];
                    const int64_t at = (int64_t)io.offset + row * io.row_stride;
                    if (io.dtype == DSP_BOOL)
                        ((SC_GLOBAL uint8_t*)p)[at] = v != (T)0 ? 1 : 0;
                    else if constexpr (INT) {  // an integer program writes the binding's own type (dspeed_hip.h)
                        const bool is_u64 = op.ip[1] == 1;
                        switch (io.dtype) {
                            case DSP_I16: case DSP_U16: ((SC_GLOBAL uint16_t*)p)[at] = (uint16_t)v; break;
                            case DSP_I32: case DSP_U32: ((SC_GLOBAL uint32_t*)p)[at] = (uint32_t)v; break;
                            case DSP_F32: ((SC_GLOBAL float*)p)[at] = is_u64 ? (float)(uint64_t)v : (float)v; break;
                            case DSP_F64: ((SC_GLOBAL double*)p)[at] = is_u64 ? (double)(uint64_t)v : (double)v; break;
                            default: ((SC_GLOBAL int64_t*)p)[at] = (int64_t)v; break;
                        }
                    } else
                        ((SC_GLOBAL T*)p)[at] = v;
                }
                break;
            }
            default: break;  // (the host gives this kernel scalar programs only)
        }
    }
}

}  // namespace

extern "C" int dsp_internal_launch_scalar(const DevProgram* dev_prog, const IoPtrs* ptrs, int64_t n_wf, int n_sregs, int type, hipStream_t stream) {
    if (n_wf <= 0) return 0;
    const unsigned blocks = (unsigned)((n_wf + 63) / 64);
    const int lds = (n_sregs > 0 ? n_sregs : 1) * 64 * (type ? 8 : 4);  // at most 128 registers: 64 KB
    if (type == 2)
        hipLaunchKernelGGL(dsp_scalar_kernel<int64_t>, dim3(blocks), dim3(64), lds, stream, dev_prog, *ptrs, n_wf);
    else if (type == 1)
        hipLaunchKernelGGL(dsp_scalar_kernel<double>, dim3(blocks), dim3(64), lds, stream, dev_prog, *ptrs, n_wf);
    else
        hipLaunchKernelGGL(dsp_scalar_kernel<float>, dim3(blocks), dim3(64), lds, stream, dev_prog, *ptrs, n_wf);
    return (int)hipGetLastError();
}

extern "C" int dsp_internal_set_scalar_lds(int lds_bytes) {
    int rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&dsp_scalar_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (rc) return rc;
    rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&dsp_scalar_kernel<int64_t>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (rc) return rc;
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&dsp_scalar_kernel<double>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
}

extern "C" const char* dsp_internal_scalar_kernel_name() { return "dsp_scalar_kernel"; }
