// planner_fuzz.cpp -- test infrastructure: random programs through the device-free planner (dspeed_amd/csrc/dsp_plan.cpp), built for the
// CPU with -fsanitize=address,undefined by tests/test_planner_fuzz.py.
//
//   planner_fuzz <programs> <seed>
//
// Two kinds of programs: (1) op lists drawn from every opcode with operands that are valid most of the time (so that the validation, the
// constant evaluation and the LDS packing behind it are reached) and garbage some of the time (so that the validation itself is); (2) the
// shapes of the specialised kernels (energy chain, lane-per-waveform rows, matrix-core FIR, pole-zero rows, reductions, current branch)
// with random lengths, offsets, strides and one field mutated.  A program the planner accepts must satisfy its invariants:
//   * LDS: every slot's region inside the wavefront's slot area, regions of slots alive at the same time disjoint, sample 0 behind its
//     zero guard, the register file and the scratch area behind the slots, the whole at most a CU's 160 kB;
//   * a specialised kernel's argument block carries the bindings' offsets and strides (a binding offset is applied, or the shape refused);
//   * the device program's ops stay inside the table, team members are 0 / 1 / 2.
// Exit code 0 and a one-line summary when every program passed; a message and exit code 1 on the first violated invariant (the sanitizers
// end the process themselves).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <random>
#include <vector>

#include "../dspeed_amd/csrc/dsp_plan.h"

// ---- the kernels' side of the planner's interface: tile geometry and names (the library links the .hip files' own; these follow them)
extern "C" int dsp_internal_current_lds_bytes(int ma_len) { return (ma_len / 16 + 1) * 16 * 64 * 4; }
extern "C" int dsp_internal_fir_mfma_lds_bytes(int kend) { return (((320 + kend + 3) & ~3) + 2 * 64 * 36 + 64 * 4 * 2) * 4; }
extern "C" int dsp_internal_fir_store_lds_bytes(int kend) { return (((320 + kend + 3) & ~3) + 2 * 64 * 36) * 4; }
extern "C" int dsp_internal_fir_f16_tz(int kend) { return ((kend + 8 + 63) / 64) * 64 + 400 + 16; }
extern "C" size_t dsp_internal_fir_f16_taps_bytes(int kend) { return (size_t)16 * dsp_internal_fir_f16_tz(kend) * 2 + 16; }
extern "C" int dsp_internal_fir_f16_lds_bytes() { return 84 * 1024; }
#define NAME(fn, text) extern "C" const char* fn() { return text; }
NAME(dsp_internal_vm_kernel_name, "dsp_vm_kernel<float>")
NAME(dsp_internal_energy_kernel_name, "dsp_energy_kernel")
NAME(dsp_internal_energy_rr_kernel_name, "dsp_energy_rr_kernel")
NAME(dsp_internal_rows_kernel_name, "dsp_rows_kernel")
NAME(dsp_internal_pz_rows_kernel_name, "dsp_pz_rows_kernel")
NAME(dsp_internal_reduce_kernel_name, "dsp_reduce_kernel")
NAME(dsp_internal_scalar_kernel_name, "dsp_scalar_kernel")
NAME(dsp_internal_current_kernel_name, "dsp_current_kernel")
NAME(dsp_internal_fir_f16_kernel_name, "dsp_fir_f16_kernel")
NAME(dsp_internal_fir_mfma_kernel_name, "dsp_fir_mfma_kernel")
NAME(dsp_internal_fir_store_kernel_name, "dsp_fir_store_kernel")
NAME(dsp_internal_fir_runs_kernel_name, "dsp_fir_runs_kernel")
extern "C" int dsp_internal_fir_runs_lds_bytes(int m) { return (4 * ((((m + 63) & ~63) + 512) * 9 / 8) + 64) * 8; }

namespace {

std::mt19937_64 rng;
int rnd(int lo, int hi) { return lo + (int)(rng() % (uint64_t)(hi - lo + 1)); }  // inclusive
bool chance(double p) { return (double)(rng() >> 11) * (1.0 / 9007199254740992.0) < p; }
template <typename T>
T pick(std::initializer_list<T> v) { return *(v.begin() + rnd(0, (int)v.size() - 1)); }

struct Prog {
    std::vector<dsp_op> ops;
    std::vector<dsp_io_desc> io;
    std::vector<int32_t> slots;
    int n_sregs = 0;
    int dtype = DSP_F32;
};

int wild() { return pick({-1, 0, 1, 7, 64, 1000, 70000, -2147483647 - 1, 2147483647, rnd(-40, 400)}); }

int add_io(Prog& p, int kind, int dtype, int len, int offset, int64_t stride) {
    dsp_io_desc d{};
    d.kind = kind, d.dtype = dtype, d.len = len, d.offset = offset, d.row_stride = stride;
    p.io.push_back(d);
    return (int)p.io.size() - 1;
}
int scalar_in(Prog& p) { return add_io(p, DSP_IO_SCALAR_IN, pick({DSP_F32, DSP_F32, DSP_F64, DSP_I16, DSP_U16, DSP_I32, DSP_U32}), 1, chance(0.1) ? rnd(0, 3) : 0, pick({1, 1, 1, 0, 2})); }
int scalar_out(Prog& p) { return add_io(p, DSP_IO_SCALAR_OUT, chance(0.1) ? DSP_BOOL : p.dtype, 1, chance(0.1) ? rnd(0, 3) : 0, pick({1, 1, 2})); }
int new_sregs(Prog& p, int n) {
    const int r = p.n_sregs;
    p.n_sregs += n;
    return r;
}
dsp_scalar_arg sarg(Prog& p) {
    dsp_scalar_arg a{};
    const int k = rnd(0, 9);
    if (k < 5 || (k >= 8 && p.n_sregs == 0)) {
        a.kind = DSP_ARG_CONST;
        a.value = pick({0.0, 1.0, -1.0, 0.5, 3.0, 16.0, 100.25, 1716.28, 1e30, -1e30, (double)NAN, (double)INFINITY, (double)rnd(-50, 5000)});
    } else if (k < 8) {
        a.kind = DSP_ARG_INPUT;
        a.index = scalar_in(p);
    } else {
        a.kind = DSP_ARG_REG;
        a.index = rnd(0, p.n_sregs - 1);
    }
    if (chance(0.01)) a.kind = rnd(-1, 4);
    if (chance(0.01)) a.index = wild();
    return a;
}
dsp_scalar_arg f32_col(Prog& p) {
    dsp_scalar_arg a{};
    a.kind = DSP_ARG_INPUT;
    a.index = add_io(p, DSP_IO_SCALAR_IN, DSP_F32, 1, 0, pick({1, 1, 1, 0, 2}));
    return a;
}
dsp_scalar_arg cst(double v);
// operand of a specialised shape: what the kernels take (a finite constant, a float32 column) four times out of five
dsp_scalar_arg shape_arg(Prog& p) { return chance(0.4) ? f32_col(p) : chance(0.67) ? cst(pick({0.0, 1.0, 12.5, 100.0, 1716.28, 4000.0})) : sarg(p); }
int f32_out(Prog& p) { return chance(0.9) ? add_io(p, DSP_IO_SCALAR_OUT, DSP_F32, 1, 0, 1) : scalar_out(p); }
dsp_scalar_arg cst(double v) {
    dsp_scalar_arg a{};
    a.kind = DSP_ARG_CONST;
    a.value = v;
    return a;
}
int in_dtype(const Prog& p) { return p.dtype == DSP_F64 ? pick({DSP_F64, DSP_I32, DSP_U32, DSP_F32, DSP_I16, DSP_U16}) : pick({DSP_F32, DSP_F32, DSP_I16, DSP_U16}); }
int new_slot(Prog& p, int len) {
    p.slots.push_back(len);
    return (int)p.slots.size() - 1;
}
int some_len() { return pick({1, 2, 3, 4, 8, 16, 63, 64, 65, 100, 256, 301, 1000, 1024, 2048, 4096, 4784, 6092, 8192, rnd(1, 9000)}); }
int any_slot(Prog& p) { return p.slots.empty() ? 0 : rnd(0, (int)p.slots.size() - 1); }
dsp_op op0(int opcode) {
    dsp_op o{};
    o.opcode = opcode;
    return o;
}
int load_slot(Prog& p, int len) {  // LOAD of a fresh slot from a fresh input binding
    const int s = new_slot(p, len);
    const int off = chance(0.3) ? rnd(0, 40) : 0;
    const int tail = chance(0.3) ? rnd(0, 40) : 0;
    dsp_op o = op0(DSP_OP_LOAD);
    o.dst = s;
    o.io = add_io(p, DSP_IO_WF_IN, in_dtype(p), len, off, (int64_t)off + len + tail);
    if (chance(0.2)) o.ip[0] = rnd(0, off), o.ip[1] = rnd(0, tail);
    if (chance(0.2)) o.ip[2] = 1;
    p.ops.push_back(o);
    return s;
}

// one random op on the program so far; mostly well-formed
void random_op(Prog& p) {
    if (p.slots.empty() || chance(0.15)) {
        if (p.slots.size() < DSP_MAX_SLOTS) load_slot(p, some_len());
        return;
    }
    const int src = any_slot(p), n = p.slots[src];
    dsp_op o{};
    switch (rnd(0, 27)) {
        case 0: {
            o = op0(DSP_OP_STORE);
            o.src = src;
            o.io = add_io(p, DSP_IO_WF_OUT, chance(0.1) ? DSP_BOOL : p.dtype, n, 0, n + (chance(0.2) ? rnd(0, 9) : 0));
            break;
        }
        case 1: o = op0(DSP_OP_STORE_SCALAR), o.io = scalar_out(p), o.ip[0] = p.n_sregs ? rnd(0, p.n_sregs - 1) : 0; break;
        case 2: o = op0(pick({DSP_OP_BL_SUBTRACT, DSP_OP_MIN_MAX_NORM})), o.src = src, o.dst = chance(0.6) ? src : new_slot(p, n), o.sp[0] = sarg(p), o.sp[1] = sarg(p), o.ip[0] = chance(0.2); break;
        case 3: o = op0(DSP_OP_POLE_ZERO), o.src = src, o.dst = chance(0.6) ? src : new_slot(p, n), o.sp[0] = sarg(p); break;
        case 4: o = op0(DSP_OP_DOUBLE_POLE_ZERO), o.src = src, o.dst = chance(0.6) ? src : new_slot(p, n), o.sp[0] = sarg(p), o.sp[1] = sarg(p), o.sp[2] = sarg(p); break;
        case 5: {
            o = op0(pick({DSP_OP_TRAP_FILTER, DSP_OP_TRAP_NORM, DSP_OP_ASYM_TRAP}));
            o.src = src, o.dst = new_slot(p, n);
            o.ip[0] = rnd(-1, n / 3 + 2), o.ip[1] = rnd(-1, n / 3 + 2), o.ip[2] = rnd(-1, n / 3 + 2);
            break;
        }
        case 6: {
            o = op0(DSP_OP_TRAP_PICKOFF);
            o.src = src, o.dst = new_sregs(p, 1), o.io = pick({'l', 'n', 'f', 'c', 'h', 'i', 's', 'x'});
            o.ip[0] = rnd(0, n / 3 + 1), o.ip[1] = rnd(0, n / 3 + 1), o.ip[2] = rnd(0, n / 3 + 1), o.ip[3] = pick({DSP_OP_TRAP_FILTER, DSP_OP_TRAP_NORM, DSP_OP_ASYM_TRAP, 0});
            o.sp[0] = sarg(p);
            break;
        }
        case 7: {
            o = op0(DSP_OP_TRAP_REDUCE);
            o.src = src, o.dst = chance(0.7) ? new_sregs(p, 4) : -1, o.io = chance(0.6) ? new_sregs(p, 1) : -1;
            int code = pick({DSP_OP_TRAP_FILTER, DSP_OP_TRAP_NORM, DSP_OP_ASYM_TRAP});
            if (chance(0.3)) code |= (pick({'l', 'n', 'i', 'h', 's'}) << 8) | ((new_sregs(p, 1) + 1) << 16);
            if (chance(0.3)) code |= 1 << 30;
            o.ip[0] = rnd(0, n / 3 + 1), o.ip[1] = rnd(0, n / 3 + 1), o.ip[2] = rnd(0, n / 3 + 1), o.ip[3] = code;
            for (int k = 0; k < 4; ++k) o.sp[k] = sarg(p);
            break;
        }
        case 8: {
            o = op0(DSP_OP_PICKOFF);
            o.src = src, o.dst = new_sregs(p, 1), o.ip[0] = pick({'l', 'n', 'f', 'c', 'h', 'i', 's', 'q'}), o.ip[1] = pick({0, 0, 1, 2, 3});
            o.sp[0] = o.ip[1] == 1 ? cst(rnd(-1, n)) : sarg(p), o.sp[1] = chance(0.8) ? cst(NAN) : sarg(p);
            break;
        }
        case 9: o = op0(pick({DSP_OP_TIME_POINT_THRESH, DSP_OP_INTERP_TIME_POINT_THRESH})), o.src = src, o.dst = new_sregs(p, 1), o.ip[0] = pick({'i', 'b', 'c', 'a', 'f', 'r', 'n', 'l', 'z'}), o.sp[0] = sarg(p), o.sp[1] = sarg(p), o.sp[2] = sarg(p); break;
        case 10: o = op0(pick({DSP_OP_MEAN_BELOW, DSP_OP_AMAX})), o.src = src, o.dst = new_sregs(p, 1), o.sp[0] = sarg(p); break;
        case 11: o = op0(pick({DSP_OP_MIN_MAX, DSP_OP_LINEAR_SLOPE_FIT})), o.src = src, o.dst = new_sregs(p, 4), o.ip[0] = chance(0.5) ? rnd(0, n) : 0, o.ip[1] = chance(0.5) ? rnd(0, n) : 0; break;
        case 12: o = op0(DSP_OP_WINDOWER), o.src = src, o.dst = new_slot(p, chance(0.9) ? rnd(1, n > 1 ? n - 1 : 1) : n), o.sp[0] = sarg(p); break;
        case 13: {
            const int L = rnd(0, n > 1 ? n - 1 : 0);
            o = op0(DSP_OP_AVG_CURRENT), o.src = src, o.dst = new_slot(p, chance(0.9) ? (n - L > 0 ? n - L : 1) : some_len()), o.sp[0] = chance(0.9) ? cst(L + (chance(0.3) ? 0.5 : 0.0)) : sarg(p);
            break;
        }
        case 14: {
            const int up = pick({1, 2, 4, 8, 16, 3, 0, -1});
            o = op0(DSP_OP_UPSAMPLER), o.src = src, o.dst = new_slot(p, chance(0.9) && up > 0 && (int64_t)n * up < 20000 ? n * up : some_len()), o.sp[0] = chance(0.9) ? cst(up) : sarg(p);
            break;
        }
        case 15: {
            o = op0(DSP_OP_MOVING_WINDOW_MULTI);
            const int win = pick({1, 3, 16, 48, 112, 0, -1, rnd(1, n)}), num = pick({0, 1, 2, 3, 4, -1});
            o.src = src, o.sp[0] = chance(0.9) ? cst(win + (chance(0.05) ? 0.5 : 0.0)) : sarg(p), o.ip[0] = pick({0, 1, 2}), o.ip[1] = num;
            if (chance(0.3)) {
                o.dst = src, o.ip[3] = 1, o.ip[2] = new_slot(p, chance(0.9) && win > 0 ? 64 * win : some_len());
            } else {
                o.dst = new_slot(p, n), o.ip[2] = chance(0.8) ? new_slot(p, n) : any_slot(p);
            }
            break;
        }
        case 16: o = op0(DSP_OP_TRAP_WINDOW_PICKOFF), o.src = src, o.dst = new_sregs(p, 1), o.ip[0] = rnd(-1, n / 2 + 1), o.ip[1] = rnd(-1, n / 2 + 1), o.sp[0] = sarg(p); break;
        case 17: {
            const int level = rnd(0, 9);
            int len = n;
            for (int l = 0; l < level; ++l) len = (len + 1) / 2;
            o = op0(DSP_OP_DWT_HAAR), o.src = src, o.dst = new_slot(p, chance(0.9) ? (len > 0 ? len : 1) : some_len()), o.ip[0] = level, o.ip[1] = pick({'a', 'd', 'a', 'x'});
            o.ip[2] = chance(0.5) ? src : new_slot(p, chance(0.9) ? n : some_len());
            break;
        }
        case 18: {
            const int step = pick({1, 1, 2, 3, -1, -2, 0}), first = rnd(0, n - 1);
            const int cnt = step >= 0 ? (n - first + (step ? step : 1) - 1) / (step ? step : 1) : first / -step + 1;
            o = op0(DSP_OP_COPY), o.src = src, o.dst = new_slot(p, chance(0.9) ? (cnt > 0 ? rnd(1, cnt) : 1) : some_len()), o.ip[0] = first, o.ip[1] = step;
            break;
        }
        case 19: {
            o = op0(DSP_OP_ELEMENTWISE);
            int fn = rnd(0, DSP_FN_LAST + 1);
            if (fn >= DSP_FN_IADD && fn <= DSP_FN_ICAST) fn |= DSP_FN_INT(pick({8, 16, 32, 64, 0}), chance(0.5));
            o.ip[0] = fn, o.src = chance(0.8) ? src : -1, o.ip[1] = chance(0.4) ? src : -1, o.ip[2] = chance(0.2) ? any_slot(p) : -1;
            o.dst = chance(0.5) ? src : new_slot(p, n);
            for (int k = 0; k < 3; ++k) o.sp[k] = sarg(p);
            break;
        }
        case 20: {
            o = op0(DSP_OP_SCALAR_FUNC);
            int fn = rnd(0, DSP_FN_LAST + 1);
            if (fn >= DSP_FN_IADD && fn <= DSP_FN_ICAST) fn |= DSP_FN_INT(pick({8, 16, 32, 64}), chance(0.5));
            o.ip[0] = fn, o.dst = new_sregs(p, 1);
            for (int k = 0; k < 3; ++k) o.sp[k] = sarg(p);
            break;
        }
        case 21:
        case 22: {
            const int mode = pick({'v', 's', 'f', 'v', 'x'}), m = rnd(1, n + 2);
            const int outlen = mode == 'v' ? n - m + 1 : mode == 's' ? n : n + m - 1;
            o = op0(pick({DSP_OP_CONVOLVE, DSP_OP_CONVOLVE_AMAX}));
            const int padded = ((m + 15) / 16) * 16;
            o.src = src, o.io = add_io(p, DSP_IO_TAPS, p.dtype, chance(0.7) ? padded : m, 0, 0), o.ip[0] = mode, o.ip[1] = rnd(0, 3), o.ip[3] = chance(0.7) ? m : 0;
            if (o.opcode == DSP_OP_CONVOLVE) o.dst = new_slot(p, chance(0.9) && outlen > 0 ? outlen : some_len());
            else o.dst = new_sregs(p, 1), o.ip[2] = chance(0.9) ? outlen : wild();
            break;
        }
        case 23: o = op0(DSP_OP_SCALAR_AFFINE), o.dst = new_sregs(p, 1), o.sp[0] = sarg(p), o.sp[1] = sarg(p), o.sp[2] = sarg(p); break;
        case 24: o = op0(DSP_OP_SCALAR_DIV), o.dst = new_sregs(p, 1), o.sp[0] = sarg(p), o.sp[1] = sarg(p); break;
        case 25: o = op0(DSP_OP_SCALAR_CONVERT), o.dst = new_sregs(p, 1), o.ip[0] = rnd(0, 4), o.sp[0] = sarg(p), o.sp[1] = sarg(p), o.sp[2] = sarg(p), o.sp[3] = cst(pick({1.0, 16.0, 0.0625})); break;
        default: o = op0(rnd(-2, 40)), o.dst = wild(), o.src = wild(), o.io = wild(); break;
    }
    // now and then: garbage in one field
    if (chance(0.03)) o.dst = wild();
    if (chance(0.03)) o.src = wild();
    if (chance(0.03)) o.io = wild();
    if (chance(0.03)) o.ip[rnd(0, 3)] = wild();
    if (p.slots.size() > DSP_MAX_SLOTS) p.slots.resize(DSP_MAX_SLOTS);
    p.ops.push_back(o);
}

Prog random_program() {
    Prog p;
    p.dtype = chance(0.25) ? DSP_F64 : DSP_F32;
    const int n = rnd(1, chance(0.1) ? 60 : 14);
    for (int i = 0; i < n && (int)p.ops.size() < DSP_MAX_OPS && (int)p.io.size() < DSP_MAX_IO - 8 && p.n_sregs < DSP_MAX_SREGS - 8; ++i) random_op(p);
    if (p.ops.empty()) random_op(p);
    return p;
}

// ---- the specialised kernels' shapes, with random geometry
Prog energy_shape() {
    Prog p;
    const int len = pick({1024, 2048, 4096, 8192, 4096, 1000, 512});
    const int off = pick({0, 0, 0, 4, 8, 3}), stride = off + len + pick({0, 0, 0, 4, 16, 1});
    const int s = new_slot(p, len), r = new_sregs(p, 1);
    dsp_op ld = op0(DSP_OP_LOAD);
    ld.dst = s, ld.io = add_io(p, DSP_IO_WF_IN, pick({DSP_F32, DSP_I16, DSP_U16}), len, off, stride);
    p.ops.push_back(ld);
    if (chance(0.8)) {
        dsp_op b = op0(DSP_OP_BL_SUBTRACT);
        b.dst = b.src = s, b.sp[0] = shape_arg(p);
        p.ops.push_back(b);
    }
    dsp_op pz = op0(DSP_OP_POLE_ZERO);
    pz.dst = pz.src = s, pz.sp[0] = chance(0.7) ? cst(1716.28) : shape_arg(p);
    p.ops.push_back(pz);
    dsp_op tp = op0(DSP_OP_TRAP_PICKOFF);
    tp.dst = r, tp.src = s, tp.io = pick({'l', 'n', 'h', 'i', 'f', 'c'}), tp.ip[0] = rnd(0, len / 3), tp.ip[1] = rnd(0, len / 3), tp.ip[3] = pick({DSP_OP_TRAP_FILTER, DSP_OP_TRAP_NORM}), tp.sp[0] = shape_arg(p);
    p.ops.push_back(tp);
    dsp_op st = op0(DSP_OP_STORE_SCALAR);
    st.io = f32_out(p), st.ip[0] = r;
    p.ops.push_back(st);
    return p;
}

Prog rows_shape() {
    Prog p;
    const int len = pick({8192, 4096, 1024, 64, 1000, 8200});
    const int off = pick({0, 0, 8, 16, 2}), stride = off + len + pick({0, 8, 1});
    const int s = new_slot(p, len);
    dsp_op ld = op0(DSP_OP_LOAD);
    ld.dst = s, ld.io = add_io(p, DSP_IO_WF_IN, pick({DSP_F32, DSP_I16, DSP_U16}), len, off, stride), ld.ip[2] = chance(0.3);
    p.ops.push_back(ld);
    if (chance(0.3)) {
        dsp_op b = op0(DSP_OP_BL_SUBTRACT);
        b.dst = b.src = s, b.sp[0] = shape_arg(p);
        p.ops.push_back(b);
    }
    int wf = s;
    if (chance(0.7)) {
        dsp_op z = op0(pick({DSP_OP_POLE_ZERO, DSP_OP_DOUBLE_POLE_ZERO}));
        z.dst = z.src = s, z.sp[0] = cst(1716.28), z.sp[1] = cst(62.5), z.sp[2] = cst(0.02);
        p.ops.push_back(z);
    }
    if (chance(0.4)) {
        const int level = rnd(2, 9), d = new_slot(p, (len >> level) > 0 ? len >> level : 1);
        dsp_op w = op0(DSP_OP_DWT_HAAR);
        w.src = wf, w.dst = d, w.ip[0] = level, w.ip[1] = pick({'a', 'd'}), w.ip[2] = wf;
        p.ops.push_back(w);
        dsp_op st = op0(DSP_OP_STORE);
        st.src = d, st.io = add_io(p, DSP_IO_WF_OUT, DSP_F32, p.slots[d], pick({0, 4, 1}), p.slots[d] + pick({0, 4, 1}) + 4);
        p.ops.push_back(st);
    }
    dsp_op tr = op0(DSP_OP_TRAP_REDUCE);
    tr.src = wf, tr.dst = chance(0.7) ? new_sregs(p, 4) : -1, tr.io = chance(0.7) ? new_sregs(p, 1) : -1;
    tr.ip[0] = pick({8, 16, 4, 100}), tr.ip[1] = pick({4, 8, 0}), tr.ip[2] = pick({125, 8, 400}), tr.ip[3] = pick({DSP_OP_TRAP_FILTER, DSP_OP_TRAP_NORM, DSP_OP_ASYM_TRAP});
    tr.sp[0] = shape_arg(p), tr.sp[1] = (tr.dst >= 0 && chance(0.5)) ? dsp_scalar_arg{DSP_ARG_REG, tr.dst + rnd(0, 1), 0.0} : shape_arg(p), tr.sp[2] = cst(pick({0.0, 1.0, 0.0, 1.0, 0.5, (double)NAN}));
    p.ops.push_back(tr);
    for (int k = 0; k < 4 && tr.dst >= 0; ++k)
        if (chance(0.7)) {
            dsp_op st = op0(DSP_OP_STORE_SCALAR);
            st.io = f32_out(p), st.ip[0] = tr.dst + k;
            p.ops.push_back(st);
        }
    if (tr.io >= 0) {
        dsp_op st = op0(DSP_OP_STORE_SCALAR);
        st.io = f32_out(p), st.ip[0] = tr.io;
        p.ops.push_back(st);
    }
    return p;
}

Prog fir_shape() {
    Prog p;
    const int full = pick({8192, 4096, 1000}), off = pick({0, 0, 8, 100, 3}), n = pick({6092, 4000, 512, full - off});
    const int len = n > 0 && off + n <= full ? n : full - off;
    const int s = new_slot(p, len);
    dsp_op ld = op0(DSP_OP_LOAD);
    ld.dst = s, ld.io = add_io(p, DSP_IO_WF_IN, pick({DSP_F32, DSP_I16, DSP_U16}), len, off, full);
    if (chance(0.5)) ld.ip[0] = off, ld.ip[1] = full - off - len;
    p.ops.push_back(ld);
    if (chance(0.6)) {
        dsp_op b = op0(DSP_OP_BL_SUBTRACT);
        b.dst = b.src = s, b.sp[0] = shape_arg(p);
        p.ops.push_back(b);
    }
    const bool store = chance(0.4);
    const int nk = store ? 1 : rnd(1, 5);
    for (int k = 0; k < nk; ++k) {
        const int m = pick({64, 133, 5792, 300, 63, len, len + 1}), mode = store ? pick({'v', 's', 'f'}) : 'v';
        const int outlen = mode == 'v' ? len - m + 1 : mode == 's' ? len : len + m - 1;
        const int padded = ((m + 15) / 16) * 16;
        dsp_op c = op0(store ? DSP_OP_CONVOLVE : DSP_OP_CONVOLVE_AMAX);
        c.src = s, c.io = add_io(p, DSP_IO_TAPS, DSP_F32, padded, 0, 0), c.ip[0] = mode, c.ip[1] = chance(0.1) ? rnd(1, 3) : 0, c.ip[3] = m;
        if (store) {
            c.dst = new_slot(p, outlen > 0 ? outlen : 1);
            p.ops.push_back(c);
            dsp_op st = op0(DSP_OP_STORE);
            st.src = c.dst, st.io = add_io(p, DSP_IO_WF_OUT, DSP_F32, p.slots[c.dst], pick({0, 0, 4}), p.slots[c.dst] + 8);
            p.ops.push_back(st);
        } else {
            c.dst = new_sregs(p, 1), c.ip[2] = outlen;
            p.ops.push_back(c);
            dsp_op st = op0(DSP_OP_STORE_SCALAR);
            st.io = f32_out(p), st.ip[0] = c.dst;
            p.ops.push_back(st);
        }
    }
    return p;
}

// LOAD, CONVOLVE with the piecewise-constant hint, [STORE], [reductions of the filtered waveform] (dsp_fir_runs.hip)
Prog fir_runs_shape() {
    Prog p;
    const int len = pick({8192, 4096, 1000, 512, 8}), off = pick({0, 0, 8, 4, 1}), stride = off + len + pick({0, 8, 3});
    const int s = new_slot(p, len);
    dsp_op ld = op0(DSP_OP_LOAD);
    ld.dst = s, ld.io = add_io(p, DSP_IO_WF_IN, pick({DSP_F32, DSP_F32, DSP_F32, DSP_I16}), len, off, stride);
    if (chance(0.1)) ld.ip[0] = off;
    p.ops.push_back(ld);
    const int m = pick({133, 16, 1, 512, 513, 64, len}), mode = pick({'v', 's', 'f', 'x'});
    const int outlen = mode == 'v' ? len - m + 1 : mode == 's' ? len : len + m - 1;
    dsp_op c = op0(DSP_OP_CONVOLVE);
    c.src = s, c.dst = new_slot(p, outlen > 0 ? outlen : 1), c.io = add_io(p, DSP_IO_TAPS, DSP_F32, ((m + 15) / 16) * 16, 0, 0);
    c.ip[0] = mode, c.ip[1] = chance(0.1) ? rnd(1, 3) : 0, c.ip[2] = chance(0.9) ? 1 : rnd(0, 2), c.ip[3] = m;
    p.ops.push_back(c);
    const bool keep = chance(0.5), store_first = chance(0.5);
    dsp_op st = op0(DSP_OP_STORE);
    st.src = c.dst, st.io = add_io(p, DSP_IO_WF_OUT, DSP_F32, p.slots[c.dst], pick({0, 0, 4}), p.slots[c.dst] + 8);
    if (keep && store_first) p.ops.push_back(st);
    int mm = -1;
    std::vector<int> regs;
    for (int k = rnd(keep ? 0 : 1, 5); k > 0; --k) {
        dsp_op o{};
        switch (rnd(0, 3)) {
            case 0: o = op0(DSP_OP_MIN_MAX), o.src = c.dst, o.dst = mm = new_sregs(p, 4); for (int q = 0; q < 4; ++q) regs.push_back(mm + q); break;
            case 1: o = op0(DSP_OP_AMAX), o.src = c.dst, o.dst = new_sregs(p, 1), regs.push_back(o.dst); break;
            case 2: o = op0(DSP_OP_PICKOFF), o.src = c.dst, o.dst = new_sregs(p, 1), o.ip[0] = pick({'i', 'l', 'n'}), o.ip[1] = pick({0, 1}), o.sp[0] = cst(pick({0.0, 50.0, 50.5, (double)outlen, -1.0})), regs.push_back(o.dst); break;
            default:
                o = op0(DSP_OP_TIME_POINT_THRESH), o.src = c.dst, o.dst = new_sregs(p, 1), o.sp[0] = shape_arg(p);
                o.sp[1] = (mm >= 0 && chance(0.5)) ? dsp_scalar_arg{DSP_ARG_REG, mm + rnd(0, 1), 0.0} : cst(pick({0.0, 100.0, 100.5, -1.0}));
                o.sp[2] = cst(pick({0.0, 1.0, 0.5}));
                regs.push_back(o.dst);
        }
        p.ops.push_back(o);
    }
    for (int r : regs)
        if (chance(0.9)) {
            dsp_op sc = op0(DSP_OP_STORE_SCALAR);
            sc.io = f32_out(p), sc.ip[0] = r;
            p.ops.push_back(sc);
        }
    if (keep && !store_first) p.ops.push_back(st);
    return p;
}

Prog pz_shape() {
    Prog p;
    const int len = pick({8192, 4096, 1000, 8}), off = pick({0, 0, 8, 4, 1}), stride = off + len + pick({0, 8, 3});
    const int s = new_slot(p, len);
    dsp_op ld = op0(DSP_OP_LOAD);
    ld.dst = s, ld.io = add_io(p, DSP_IO_WF_IN, pick({DSP_F32, DSP_I16, DSP_U16}), len, off, stride);
    p.ops.push_back(ld);
    int mm = -1;
    if (chance(0.4)) {  // min_max of the rows as they are read
        dsp_op m = op0(DSP_OP_MIN_MAX);
        m.src = s, m.dst = mm = new_sregs(p, 4);
        p.ops.push_back(m);
    }
    if (chance(0.6)) {
        dsp_op b = op0(DSP_OP_BL_SUBTRACT);
        b.dst = b.src = s, b.sp[0] = shape_arg(p);
        p.ops.push_back(b);
    }
    dsp_op pz = op0(DSP_OP_POLE_ZERO);
    pz.dst = pz.src = s, pz.sp[0] = chance(0.7) ? cst(pick({1716.28, 1716.28, 0.0, (double)NAN})) : shape_arg(p);
    p.ops.push_back(pz);
    const bool stores_first = chance(0.5);
    auto stores = [&]() {
        for (int k = 0; mm >= 0 && k < 4; ++k)
            if (chance(0.85)) {
                dsp_op sc = op0(DSP_OP_STORE_SCALAR);
                sc.io = f32_out(p), sc.ip[0] = mm + k;
                p.ops.push_back(sc);
            }
    };
    if (stores_first) stores();
    dsp_op st = op0(DSP_OP_STORE);
    const int ooff = pick({0, 0, 4, 1});
    st.src = s, st.io = add_io(p, DSP_IO_WF_OUT, DSP_F32, len, ooff, ooff + len + pick({0, 4}));
    p.ops.push_back(st);
    if (!stores_first) stores();
    return p;
}

Prog reduce_shape() {
    Prog p;
    const int len = pick({8192, 4096, 301, 1000}), off = pick({0, 0, 8, 3}), stride = off + len + pick({0, 8, 1});
    const int s = new_slot(p, len);
    dsp_op ld = op0(DSP_OP_LOAD);
    ld.dst = s, ld.io = add_io(p, DSP_IO_WF_IN, pick({DSP_F32, DSP_I16, DSP_U16}), len, off, stride);
    p.ops.push_back(ld);
    int mm = -1, last_walk = -1;
    std::vector<int> regs;
    ld.ip[2] = chance(0.5);  // (the promise about the rows' NaNs: walks alone then need no pass over the row)
    p.ops[0] = ld;
    for (int k = rnd(1, 8); k > 0; --k) {
        dsp_op o{};
        switch (rnd(0, 3)) {
            case 0: o = op0(DSP_OP_MIN_MAX), o.src = s, o.dst = mm = new_sregs(p, 4); for (int q = 0; q < 4; ++q) regs.push_back(mm + q); break;
            case 1: o = op0(DSP_OP_AMAX), o.src = s, o.dst = new_sregs(p, 1), regs.push_back(o.dst); break;
            case 2: o = op0(DSP_OP_PICKOFF), o.src = s, o.dst = new_sregs(p, 1), o.ip[0] = pick({'i', 'l', 'n'}), o.ip[1] = pick({0, 1}), o.sp[0] = cst(pick({0.0, 50.0, 50.5, (double)len, -1.0})), regs.push_back(o.dst); break;
            default: {
                // a threshold: a constant, a column, a fraction of a column (SCALAR_AFFINE in front); a start: a constant, an extreme, a column,
                // where an earlier walk ended
                dsp_scalar_arg thr = shape_arg(p);
                if (chance(0.3)) {
                    dsp_op a = op0(DSP_OP_SCALAR_AFFINE);
                    a.dst = new_sregs(p, 1), a.sp[0] = shape_arg(p), a.sp[1] = cst(pick({0.9, 0.5, 1.0})), a.sp[2] = cst(pick({0.0, 0.0, -0.0, 1.0}));
                    p.ops.push_back(a);
                    thr = dsp_scalar_arg{DSP_ARG_REG, a.dst, 0.0};
                }
                o = op0(DSP_OP_TIME_POINT_THRESH), o.src = s, o.dst = new_sregs(p, 1), o.sp[0] = thr;
                if (mm >= 0 && chance(0.3)) o.sp[1] = dsp_scalar_arg{DSP_ARG_REG, mm + rnd(0, 1), 0.0};
                else if (last_walk >= 0 && chance(0.4)) o.sp[1] = dsp_scalar_arg{DSP_ARG_REG, last_walk, 0.0};
                else if (chance(0.3)) o.sp[1] = shape_arg(p);
                else o.sp[1] = cst(pick({0.0, 100.0, 100.5, -1.0}));
                o.sp[2] = cst(pick({0.0, 1.0, 0.5}));
                regs.push_back(o.dst);
                last_walk = o.dst;
            }
        }
        p.ops.push_back(o);
    }
    for (int r : regs)
        if (chance(0.9)) {
            dsp_op st = op0(DSP_OP_STORE_SCALAR);
            st.io = f32_out(p), st.ip[0] = r;
            p.ops.push_back(st);
        }
    return p;
}

Prog current_shape() {
    Prog p;
    const int len = pick({8192, 4096}), off = pick({0, 0, 4, 1}), stride = off + len + pick({0, 4});
    const int win = pick({301, 301, 101, 300}), lag = pick({1, 1, 2, 0}), up = pick({16, 16, 8, 1, 3}), ma = pick({48, 48, 16, 112, 40, 128});
    const int reach = (win - lag) * up - up / 2;  // upsampled samples an input sample lands on
    const int s = new_slot(p, len), w = new_slot(p, win), c = new_slot(p, win - lag), u = new_slot(p, pick({reach / 16 * 16, reach / 16 * 16, (win - lag) * up, 4784}));
    dsp_op ld = op0(DSP_OP_LOAD);
    ld.dst = s, ld.io = add_io(p, DSP_IO_WF_IN, pick({DSP_F32, DSP_F32, DSP_F32, DSP_I16}), len, off, stride), ld.ip[2] = chance(0.5);
    p.ops.push_back(ld);
    dsp_op a = op0(DSP_OP_WINDOWER);
    a.src = s, a.dst = w, a.sp[0] = shape_arg(p);
    p.ops.push_back(a);
    dsp_op b = op0(DSP_OP_AVG_CURRENT);
    b.src = w, b.dst = c, b.sp[0] = cst(lag);
    p.ops.push_back(b);
    dsp_op d = op0(DSP_OP_UPSAMPLER);
    d.src = c, d.dst = u, d.sp[0] = cst(up);
    p.ops.push_back(d);
    dsp_op m = op0(DSP_OP_MOVING_WINDOW_MULTI);
    const bool inplace = chance(0.6);
    m.src = u, m.dst = inplace ? u : new_slot(p, p.slots[u]), m.sp[0] = cst(ma), m.ip[0] = pick({0, 0, 0, 1}), m.ip[1] = pick({3, 3, 3, 2, 1});
    m.ip[2] = inplace ? new_slot(p, 64 * ma) : u, m.ip[3] = inplace;
    p.ops.push_back(m);
    dsp_op mmx = op0(DSP_OP_MIN_MAX);
    mmx.src = m.dst, mmx.dst = new_sregs(p, 4);
    p.ops.push_back(mmx);
    for (int k = 0; k < 4; ++k)
        if (chance(0.8)) {
            dsp_op st = op0(DSP_OP_STORE_SCALAR);
            st.io = f32_out(p), st.ip[0] = mmx.dst + k;
            p.ops.push_back(st);
        }
    return p;
}

// arithmetic between per-event values and stores only: the row-per-lane kernel's programs (the tail of a recipe)
// an integer program (compute type DSP_I64, dspeed_hip.h): SCALAR_FUNC and STORE_SCALAR ops only, integer / bool input columns, outputs of any type
Prog integer_shape() {
    Prog p;
    p.dtype = DSP_I64;
    auto in_col = [&]() {
        dsp_scalar_arg a{};
        a.kind = DSP_ARG_INPUT;
        a.index = add_io(p, DSP_IO_SCALAR_IN, chance(0.02) ? DSP_F32 : pick({DSP_I64, DSP_U64, DSP_I32, DSP_U32, DSP_I16, DSP_U16, DSP_BOOL, DSP_I64}), 1, 0, pick({1, 1, 0, 2}));
        return a;
    };
    auto opnd = [&]() {
        if (p.n_sregs && chance(0.4)) return dsp_scalar_arg{DSP_ARG_REG, rnd(0, p.n_sregs - 1), 0.0};
        if (chance(0.5)) return in_col();
        return cst((double)pick({0, 1, -1, 7, 65535, 1 << 30, rnd(-1000, 1000)}));
    };
    for (int k = rnd(1, 30); k > 0 && p.n_sregs < DSP_MAX_SREGS - 4 && (int)p.io.size() < DSP_MAX_IO - 6; --k) {
        dsp_op o{};
        if (p.n_sregs && chance(0.3)) {
            o = op0(DSP_OP_STORE_SCALAR);
            o.io = add_io(p, DSP_IO_SCALAR_OUT, pick({DSP_I64, DSP_U64, DSP_I32, DSP_U32, DSP_I16, DSP_U16, DSP_BOOL, DSP_F32, DSP_F64}), 1, 0, 1);
            o.ip[0] = rnd(0, p.n_sregs - 1), o.ip[1] = chance(0.2);
        } else {
            int fn = pick({DSP_FN_IADD, DSP_FN_ISUB, DSP_FN_IMUL, DSP_FN_IFLOORDIV, DSP_FN_ICAST, DSP_FN_LT, DSP_FN_GE, DSP_FN_EQ, DSP_FN_WHERE, DSP_FN_LOR, DSP_FN_LAND, DSP_FN_COPY});
            if (chance(0.02)) fn = pick({DSP_FN_ADD, DSP_FN_RINT, DSP_FN_DIV});  // (float functions: refused in an integer program)
            if (fn >= DSP_FN_IADD && fn <= DSP_FN_ICAST) fn |= DSP_FN_INT(chance(0.02) ? 24 : pick({8, 16, 32, 64, 64}), chance(0.5));
            else if (chance(0.3)) fn |= DSP_FN_INT(64, chance(0.5));
            o = op0(chance(0.97) ? DSP_OP_SCALAR_FUNC : DSP_OP_SCALAR_AFFINE), o.ip[0] = fn;
            for (int q = 0; q < 3; ++q) o.sp[q] = opnd();
            o.dst = new_sregs(p, 1);
        }
        p.ops.push_back(o);
    }
    if (p.ops.empty()) p.ops.push_back(op0(DSP_OP_SCALAR_FUNC)), p.n_sregs = 1;
    return p;
}

Prog scalar_shape() {
    if (chance(0.3)) return integer_shape();
    Prog p;
    p.dtype = chance(0.3) ? DSP_F64 : DSP_F32;
    for (int k = rnd(1, 40); k > 0 && p.n_sregs < DSP_MAX_SREGS - 4 && (int)p.io.size() < DSP_MAX_IO - 6; --k) {
        dsp_op o{};
        switch (rnd(0, 4)) {
            case 0: o = op0(DSP_OP_SCALAR_AFFINE), o.sp[0] = sarg(p), o.sp[1] = sarg(p), o.sp[2] = sarg(p), o.dst = new_sregs(p, 1); break;
            case 1: o = op0(DSP_OP_SCALAR_DIV), o.sp[0] = sarg(p), o.sp[1] = sarg(p), o.dst = new_sregs(p, 1); break;
            case 2: o = op0(DSP_OP_SCALAR_CONVERT), o.ip[0] = rnd(0, 4), o.sp[0] = sarg(p), o.sp[1] = sarg(p), o.sp[2] = sarg(p), o.sp[3] = cst(pick({1.0, 16.0, 0.0625})), o.dst = new_sregs(p, 1); break;
            case 3: {
                int fn = rnd(0, DSP_FN_LAST);
                if (fn >= DSP_FN_IADD && fn <= DSP_FN_ICAST) fn |= DSP_FN_INT(pick({8, 16, 32}), chance(0.5));
                o = op0(DSP_OP_SCALAR_FUNC), o.ip[0] = fn;
                for (int q = 0; q < 3; ++q) o.sp[q] = sarg(p);
                o.dst = new_sregs(p, 1);
                break;
            }
            default:
                if (!p.n_sregs) continue;
                o = op0(DSP_OP_STORE_SCALAR), o.io = scalar_out(p), o.ip[0] = rnd(0, p.n_sregs - 1);
        }
        p.ops.push_back(o);
    }
    if (p.ops.empty()) p.ops.push_back(op0(DSP_OP_SCALAR_AFFINE)), p.n_sregs = 1;
    return p;
}

void mutate(Prog& p) {
    if (p.ops.empty()) return;
    dsp_op& o = p.ops[rnd(0, (int)p.ops.size() - 1)];
    switch (rnd(0, 7)) {
        case 0: o.dst = wild(); break;
        case 1: o.src = wild(); break;
        case 2: o.io = wild(); break;
        case 3: o.ip[rnd(0, 3)] = wild(); break;
        case 4: o.sp[rnd(0, 3)] = sarg(p); break;
        case 5:
            if (!p.io.empty()) {
                dsp_io_desc& d = p.io[rnd(0, (int)p.io.size() - 1)];
                switch (rnd(0, 3)) {
                    case 0: d.offset = wild(); break;
                    case 1: d.row_stride = wild(); break;
                    case 2: d.len = wild(); break;
                    default: d.dtype = rnd(-1, 8);
                }
            }
            break;
        case 6:
            if (!p.slots.empty()) p.slots[rnd(0, (int)p.slots.size() - 1)] = wild();
            break;
        default: o.opcode = rnd(0, 40);
    }
}

#define REQUIRE(cond, ...)                                             \
    do {                                                               \
        if (!(cond)) {                                                 \
            fprintf(stderr, "INVARIANT VIOLATED: %s -- ", #cond);     \
            fprintf(stderr, __VA_ARGS__);                              \
            fprintf(stderr, "\n");                                    \
            return false;                                              \
        }                                                              \
    } while (0)

bool check(const Prog& p, const ChainPlan& c) {
    const DevProgram& P = c.host;
    const int ns = (int)p.slots.size();
    const int esz = (c.f64 || c.i64) ? 8 : 4;
    REQUIRE(c.lds_bytes_per_wave == P.lds_elems_per_wave * esz && c.lds_bytes_per_wave <= LDS_BYTES_PER_CU, "LDS %d bytes", c.lds_bytes_per_wave);
    REQUIRE(P.sreg_off + p.n_sregs <= P.scratch_off && P.scratch_off + DSP_SCRATCH_ELEMS <= P.lds_elems_per_wave, "register file / scratch: %d + %d, %d, %d", P.sreg_off,
            p.n_sregs, P.scratch_off, P.lds_elems_per_wave);
    REQUIRE(P.scratch_off % 4 == 0, "scratch alignment %d", P.scratch_off);
    for (int s = 0; s < ns; ++s) {
        const DevSlot& d = P.slots[s];
        REQUIRE(c.slot_base[s] >= 0 && c.slot_base[s] + c.slot_foot[s] <= P.sreg_off, "slot %d region [%d, +%d) beyond the slot area %d", s, c.slot_base[s], c.slot_foot[s], P.sreg_off);
        REQUIRE(c.slot_base[s] % 4 == 0 && c.slot_foot[s] % 4 == 0, "slot %d region alignment", s);
        REQUIRE(d.len == p.slots[s] && d.C % 16 == 0 && d.C * 64 >= d.len && d.pitch == d.C + d.padw && (d.padw == 0 || d.padw == 1), "slot %d geometry len %d C %d pitch %d", s, d.len, d.C, d.pitch);
        REQUIRE(d.off == c.slot_base[s] + 2 * d.pitch && d.off + 64 * d.pitch <= c.slot_base[s] + c.slot_foot[s], "slot %d image outside its region", s);
        REQUIRE(d.zero_below >= 0 && d.zero_below <= 2 * d.pitch && d.zero_above >= 0 && d.off + 64 * d.pitch + d.zero_above <= c.slot_base[s] + c.slot_foot[s], "slot %d zero margins", s);
        for (int t = 0; t < s; ++t) {
            const bool together = c.slot_first_op[s] <= c.slot_last_op[t] && c.slot_first_op[t] <= c.slot_last_op[s];
            const bool overlap = c.slot_base[s] < c.slot_base[t] + c.slot_foot[t] && c.slot_base[t] < c.slot_base[s] + c.slot_foot[s];
            REQUIRE(!(together && overlap), "slots %d and %d are alive together (ops %d..%d, %d..%d) and share LDS", s, t, c.slot_first_op[s], c.slot_last_op[s], c.slot_first_op[t],
                    c.slot_last_op[t]);
            if (overlap) REQUIRE(c.slot_shares[s] && c.slot_shares[t], "slots %d and %d share LDS without the clearing op", s, t);
        }
    }
    if (c.i64) {  // an integer program: the row-per-lane kernel, 8-byte registers, nothing but functions of per-event values and stores
        REQUIRE(c.scalar_ok && ns == 0 && esz == 8, "integer program on %s", dsp_plan_kernel_name(&c));
        for (size_t i = 0; i < p.ops.size(); ++i) REQUIRE(p.ops[i].opcode == DSP_OP_SCALAR_FUNC || p.ops[i].opcode == DSP_OP_STORE_SCALAR, "integer program holds opcode %d", p.ops[i].opcode);
        for (size_t k = 0; k < p.io.size(); ++k)
            if (p.io[k].kind == DSP_IO_SCALAR_IN) REQUIRE(p.io[k].dtype != DSP_F32 && p.io[k].dtype != DSP_F64, "integer program reads a float column");
    } else {
        for (size_t k = 0; k < p.io.size(); ++k)
            if (p.io[k].kind == DSP_IO_SCALAR_OUT || p.io[k].kind == DSP_IO_WF_OUT) REQUIRE(p.io[k].dtype == p.dtype || p.io[k].dtype == DSP_BOOL, "output %zu of type %d in a chain of type %d", k, p.io[k].dtype, p.dtype);
    }
    REQUIRE(P.n_ops >= 1 && P.n_ops <= DSP_MAX_OPS + DSP_MAX_SLOTS, "device ops %d", P.n_ops);
    REQUIRE(P.team >= 1 && P.team <= 3, "team %d", P.team);
    for (int i = 0; i < P.n_ops; ++i) REQUIRE((P.ops[i].member >= 0 && P.ops[i].member < P.team) || P.ops[i].member == DSP_MEMBER_ALL, "op %d member %d", i, P.ops[i].member);
    REQUIRE(c.waves_per_block >= 1 && c.waves_per_block <= 4 && P.waves_per_block == c.waves_per_block, "waves per block %d", c.waves_per_block);
    auto io_ok = [&](int k, int kind) { return k >= 0 && k < (int)p.io.size() && p.io[k].kind == kind; };
    if (c.rr_ok || c.fused_ok) {
        REQUIRE(io_ok(c.io_wf, DSP_IO_WF_IN) && io_ok(c.io_out, DSP_IO_SCALAR_OUT), "energy kernel bindings");
        const EnergyArgs& F = c.rr_ok ? c.rr : c.fused;
        REQUIRE(F.wf_offset == p.io[c.io_wf].offset && F.wf_stride == p.io[c.io_wf].row_stride && F.len == p.io[c.io_wf].len, "energy kernel: the row binding's offset / stride / length");
        REQUIRE((p.io[c.io_wf].offset * dsp_elem_size(p.io[c.io_wf].dtype)) % 16 == 0 && (p.io[c.io_wf].row_stride * dsp_elem_size(p.io[c.io_wf].dtype)) % 16 == 0, "energy kernel on unaligned rows");
        REQUIRE(F.out_stride == p.io[c.io_out].row_stride, "energy kernel: output stride");
        if (c.io_bl >= 0) REQUIRE(io_ok(c.io_bl, DSP_IO_SCALAR_IN) && F.bl_stride == p.io[c.io_bl].row_stride, "energy kernel: baseline column");
        if (c.io_tp >= 0) REQUIRE(io_ok(c.io_tp, DSP_IO_SCALAR_IN) && F.tp_stride == p.io[c.io_tp].row_stride, "energy kernel: pick-off column");
        if (c.rr_ok) REQUIRE(c.rr_lds_bytes > 0 && c.rr_lds_bytes <= LDS_BYTES_PER_CU, "energy kernel LDS %d", c.rr_lds_bytes);
    }
    if (c.rows_ok) {
        REQUIRE(io_ok(c.rio_wf, DSP_IO_WF_IN), "rows kernel: row binding");
        REQUIRE(c.rows.wf_offset == p.io[c.rio_wf].offset && c.rows.wf_stride == p.io[c.rio_wf].row_stride, "rows kernel: the row binding's offset / stride");
        REQUIRE(c.rows.len > 0 && c.rows.len % 8 == 0 && c.rows.len + c.rows.wf_offset <= c.rows.wf_stride, "rows kernel: length %d", c.rows.len);
        REQUIRE(c.rows_lds_bytes > 0 && c.rows_lds_bytes <= LDS_BYTES_PER_CU && c.rows.ring_entries % 8 == 0, "rows kernel LDS %d", c.rows_lds_bytes);
        for (int k = 0; k < 3; ++k) REQUIRE(c.rows.lag[k] >= 8 && c.rows.lag[k] + 8 <= c.rows.ring_entries, "rows kernel: lag %d of a ring of %d", c.rows.lag[k], c.rows.ring_entries);
    }
    if (c.fir_ok) {
        REQUIRE(io_ok(c.fio_wf, DSP_IO_WF_IN), "FIR kernel: row binding");
        REQUIRE(c.fir.wf_offset == p.io[c.fio_wf].offset && c.fir.wf_stride == p.io[c.fio_wf].row_stride && c.fir.n == p.io[c.fio_wf].len, "FIR kernel: the row binding's offset / stride / length");
        REQUIRE(c.fir.n_kernels >= 1 && c.fir.n_kernels <= DSP_FIR_MAXK, "FIR kernels %d", c.fir.n_kernels);
        for (int k = 0; k < c.fir.n_kernels; ++k) {
            REQUIRE(io_ok(c.fio_taps[k], DSP_IO_TAPS) && c.fir.m[k] >= 64 && c.fir.m[k] <= p.io[c.fio_taps[k]].len && c.fir.m[k] <= c.fir.n, "FIR kernel %d: %d taps", k, c.fir.m[k]);
            REQUIRE(c.fio_out[k] >= 0 && c.fio_out[k] < (int)p.io.size(), "FIR kernel %d: output binding", k);
        }
        REQUIRE(c.fir.kend % 32 == 0 && c.fir.kend > 0, "FIR kernel: kend %d", c.fir.kend);
        // (amax form: the staging loads run to the end of the last 32-sample stage, inside the row; kept output: kend is a tile's window)
        if (!c.fir.store) REQUIRE(c.fir.kend + c.fir.wf_offset <= c.fir.wf_stride, "FIR kernel: stages reach %d samples into a row of %lld", c.fir.kend + c.fir.wf_offset, (long long)c.fir.wf_stride);
        REQUIRE(c.fir_lds_bytes > 0 && c.fir_lds_bytes <= LDS_BYTES_PER_CU, "FIR kernel LDS %d", c.fir_lds_bytes);
    }
    if (c.pz_ok) {
        REQUIRE(io_ok(c.pio_wf, DSP_IO_WF_IN) && io_ok(c.pio_out, DSP_IO_WF_OUT), "pole-zero rows: bindings");
        REQUIRE(c.pz.wf_offset == p.io[c.pio_wf].offset && c.pz.wf_stride == p.io[c.pio_wf].row_stride && c.pz.len == p.io[c.pio_wf].len && c.pz.out_stride == p.io[c.pio_out].row_stride,
                "pole-zero rows: the bindings' offset / stride / length");
        REQUIRE(c.pz.len % 8 == 0, "pole-zero rows: length %d", c.pz.len);
        for (int k = 0; k < 4; ++k)
            if (c.pio_mm[k] >= 0) REQUIRE(c.pz.mm_on && io_ok(c.pio_mm[k], DSP_IO_SCALAR_OUT) && c.pz.mm_stride[k] == p.io[c.pio_mm[k]].row_stride, "pole-zero rows: min_max output %d", k);
    }
    if (c.red_ok) {
        REQUIRE(io_ok(c.dio_wf, DSP_IO_WF_IN), "reduce kernel: row binding");
        REQUIRE(c.red.wf_offset == p.io[c.dio_wf].offset && c.red.wf_stride == p.io[c.dio_wf].row_stride && c.red.len == p.io[c.dio_wf].len, "reduce kernel: the row binding's offset / stride / length");
        for (int k = 0; k < DSP_REDUCE_PICKS; ++k)
            if (c.dio_pick[k] >= 0) REQUIRE(c.red.pick_at[k] >= -1 && c.red.pick_at[k] < c.red.len, "reduce kernel: pick-off at %d of %d", c.red.pick_at[k], c.red.len);
        for (int k = 0; k < DSP_REDUCE_WALKS; ++k)
            if (c.dio_walk[k] >= 0 && c.red.walk_from[k] == 0) REQUIRE(c.red.walk_start[k] >= 0 && c.red.walk_start[k] < c.red.len, "reduce kernel: walk from %d of %d", c.red.walk_start[k], c.red.len);
        REQUIRE(c.red.n_walks >= 0 && c.red.n_walks <= DSP_REDUCE_WALKS, "reduce kernel: %d walks", c.red.n_walks);
        for (int k = 0; k < c.red.n_walks; ++k) {
            const int from = c.red.walk_from[k];
            REQUIRE(from >= 0 && from < 4 + k, "reduce kernel: walk %d starts from %d", k, from);
            if (from == 3) REQUIRE(io_ok(c.dio_walk_ts[k], DSP_IO_SCALAR_IN) && p.io[c.dio_walk_ts[k]].dtype == DSP_F32 && c.red.walk_ts_stride[k] == p.io[c.dio_walk_ts[k]].row_stride, "reduce kernel: walk %d's start column", k);
            if (c.red.walk_thr_scaled[k]) REQUIRE(io_ok(c.dio_walk_thr[k], DSP_IO_SCALAR_IN), "reduce kernel: walk %d scales no column", k);
        }
        if (!c.red.need_stream) REQUIRE(c.dio_out[0] < 0 && c.dio_out[4] < 0 && (p.io[c.dio_wf].dtype != DSP_F32 || (p.ops[0].ip[2] & 1)), "reduce kernel: no pass over rows that need one");
    }
    if (c.runs_ok) {
        REQUIRE(!c.red_ok && !c.fir_ok, "run-length FIR beside another kernel of the same program");
        REQUIRE(io_ok(c.uio_wf, DSP_IO_WF_IN) && p.io[c.uio_wf].dtype == DSP_F32 && io_ok(c.uio_taps, DSP_IO_TAPS), "run-length FIR: bindings");
        const FirRunsArgs& R = c.runs;
        REQUIRE(R.wf_offset == p.io[c.uio_wf].offset && R.wf_stride == p.io[c.uio_wf].row_stride && R.n == p.io[c.uio_wf].len, "run-length FIR: the row binding's offset / stride / length");
        REQUIRE(R.n % 8 == 0 && R.wf_offset % 4 == 0 && R.wf_stride % 4 == 0 && R.wf_offset + R.n <= R.wf_stride, "run-length FIR: rows of %d samples at %d of %lld", R.n, R.wf_offset, (long long)R.wf_stride);
        REQUIRE(R.m >= 1 && R.m <= DSP_FIR_RUNS_MAX_TAPS && R.m <= R.n && R.m <= p.io[c.uio_taps].len, "run-length FIR: %d taps", R.m);
        REQUIRE(R.start >= 0 && R.start <= R.m - 1 && R.p >= 1 && R.p + R.start <= R.n + R.m - 1 + R.start && R.p <= R.n + R.m - 1, "run-length FIR: %d outputs from %d", R.p, R.start);
        REQUIRE(R.keep ? (io_ok(c.uio_out, DSP_IO_WF_OUT) && p.io[c.uio_out].len == R.p && R.out_stride == p.io[c.uio_out].row_stride) : (R.has_red && R.out_stride >= R.p),
                "run-length FIR: where the filtered waveform goes");
        if (R.has_red) {
            REQUIRE(R.red.len == R.p, "run-length FIR: reductions over %d of %d outputs", R.red.len, R.p);
            for (int k = 0; k < DSP_REDUCE_PICKS; ++k)
                if (c.dio_pick[k] >= 0) REQUIRE(R.red.pick_at[k] >= -1 && R.red.pick_at[k] < R.p, "run-length FIR: pick-off at %d of %d", R.red.pick_at[k], R.p);
            for (int k = 0; k < DSP_REDUCE_WALKS; ++k)
                if (c.dio_walk[k] >= 0 && R.red.walk_from[k] == 0) REQUIRE(R.red.walk_start[k] >= 0 && R.red.walk_start[k] < R.p, "run-length FIR: walk from %d of %d", R.red.walk_start[k], R.p);
        }
    }
    if (c.cur_ok) {
        REQUIRE(io_ok(c.cio_wf, DSP_IO_WF_IN), "current kernel: row binding");
        REQUIRE(c.cur.wf_offset == p.io[c.cio_wf].offset && c.cur.wf_stride == p.io[c.cio_wf].row_stride && c.cur.n_in == p.io[c.cio_wf].len, "current kernel: the row binding's offset / stride / length");
        REQUIRE(c.cur.win_len < c.cur.n_in && c.cur.n_c == c.cur.win_len - c.cur.ac_lag && c.cur.n_up % 16 == 0 && c.cur.ma_len % 16 == 0 && c.cur.ma_len <= 112 && c.cur.up_shift >= 0 && c.cur.up_shift <= 4,
                "current kernel geometry");
        REQUIRE(c.cur_lds_bytes > 0 && c.cur_lds_bytes <= LDS_BYTES_PER_CU, "current kernel LDS %d", c.cur_lds_bytes);
    }
    return true;
}

void dump(const Prog& p) {
    fprintf(stderr, "program: dtype %d, %zu ops, %zu io, %zu slots, %d sregs\n", p.dtype, p.ops.size(), p.io.size(), p.slots.size(), p.n_sregs);
    for (size_t s = 0; s < p.slots.size(); ++s) fprintf(stderr, "  slot %zu: %d\n", s, p.slots[s]);
    for (size_t k = 0; k < p.io.size(); ++k) fprintf(stderr, "  io %zu: kind %d dtype %d len %d offset %d stride %lld\n", k, p.io[k].kind, p.io[k].dtype, p.io[k].len, p.io[k].offset, (long long)p.io[k].row_stride);
    for (size_t i = 0; i < p.ops.size(); ++i) {
        const dsp_op& o = p.ops[i];
        fprintf(stderr, "  op %zu: opcode %d dst %d src %d io %d ip %d %d %d %d sp", i, o.opcode, o.dst, o.src, o.io, o.ip[0], o.ip[1], o.ip[2], o.ip[3]);
        for (int k = 0; k < 4; ++k) fprintf(stderr, " (%d %d %g)", o.sp[k].kind, o.sp[k].index, o.sp[k].value);
        fprintf(stderr, "\n");
    }
}

}  // namespace

int main(int argc, char** argv) {
    const long n_programs = argc > 1 ? atol(argv[1]) : 10000;
    const unsigned long long seed = argc > 2 ? strtoull(argv[2], nullptr, 0) : 0xD5BEEDull;
    rng.seed(seed);
    long accepted = 0, by_kind[9] = {0}, n_integer = 0;
    long kernels[11] = {0};
    for (long it = 0; it < n_programs; ++it) {
        Prog p;
        const int kind = rnd(0, 14);
        switch (kind) {
            case 0: p = energy_shape(); break;
            case 1: p = rows_shape(); break;
            case 2: p = fir_shape(); break;
            case 3: p = pz_shape(); break;
            case 4: p = reduce_shape(); break;
            case 5: p = current_shape(); break;
            case 6: p = scalar_shape(); break;
            case 7: p = fir_runs_shape(); break;
            default: p = random_program();
        }
        if (kind <= 7 && chance(0.3)) mutate(p);
        if (kind > 7 && chance(0.05)) mutate(p);
        if (p.slots.size() > DSP_MAX_SLOTS && chance(0.9)) p.slots.resize(DSP_MAX_SLOTS);
        std::unique_ptr<ChainPlan> plan(new ChainPlan());
        const int rc = dsp_plan_build(plan.get(), p.ops.data(), (int)p.ops.size(), p.io.data(), (int)p.io.size(), p.slots.data(), (int)p.slots.size(), p.n_sregs, p.dtype);
        if (rc != DSP_OK) {
            if (!dsp_plan_last_error()[0] && rc != DSP_E_ZERODIV) {
                fprintf(stderr, "program %ld refused with code %d and no message\n", it, rc);
                dump(p);
                return 1;
            }
            continue;
        }
        ++accepted;
        ++by_kind[kind <= 7 ? kind : 8];
        n_integer += plan->i64 ? 1 : 0;
        kernels[plan->scalar_ok ? 0 : plan->pz_ok ? 1 : plan->red_ok ? 2 : plan->runs_ok ? 10 : plan->cur_ok ? 3 : plan->fir_ok ? 4 : plan->rows_ok ? 5 : plan->rr_ok ? 6 : plan->fused_ok ? 7 : plan->host.team >= 2 ? 8 : 9]++;
        if (!check(p, *plan)) {
            fprintf(stderr, "program %ld (seed %llu, kind %d, kernel %s)\n", it, seed, kind, dsp_plan_kernel_name(plan.get()));
            dump(p);
            return 1;
        }
    }
    printf("{\"programs\": %ld, \"accepted\": %ld, \"accepted_by_generator\": {\"energy\": %ld, \"rows\": %ld, \"fir\": %ld, \"pz\": %ld, \"reduce\": %ld, \"current\": %ld, \"scalar\": %ld, \"fir_runs\": %ld, \"random\": %ld}, "
           "\"kernels\": {\"scalar\": %ld, \"pz_rows\": %ld, \"reduce\": %ld, \"current\": %ld, \"fir\": %ld, \"rows\": %ld, \"energy_rr\": %ld, \"energy\": %ld, \"vm_team\": %ld, \"vm\": %ld, \"fir_runs\": %ld}, \"integer_programs\": %ld}\n",
           n_programs, accepted, by_kind[0], by_kind[1], by_kind[2], by_kind[3], by_kind[4], by_kind[5], by_kind[6], by_kind[7], by_kind[8], kernels[0], kernels[1], kernels[2], kernels[3], kernels[4], kernels[5],
           kernels[6], kernels[7], kernels[8], kernels[9], kernels[10], n_integer);
    return 0;
}
