// dsp_plan.cpp -- chain translation without the device: validation, host-evaluated constants, LDS packing by slot lifetime, the shape
// matchers of the specialised kernels and the launch geometry (see dsp_plan.h).  No HIP header, no HIP call: this file is also built for
// the CPU under AddressSanitizer / UBSan and fuzzed (tools/planner_fuzz.cpp).
//
// Reference behaviour mirrored here: constant-only DSPFatal conditions are raised at chain creation with the reference's own
// codes / messages (processors/*.py, cited in include/dspeed_hip.h).
#include "dsp_plan.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

namespace {
thread_local std::string g_last_error;
}

int dsp_fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}
void dsp_set_last_error(const char* text) { g_last_error = text ? text : ""; }
const char* dsp_plan_last_error() { return g_last_error.c_str(); }

int dsp_elem_size(int dtype) {
    switch (dtype) {
        case DSP_F32: case DSP_I32: case DSP_U32: return 4;
        case DSP_F64: case DSP_I64: case DSP_U64: return 8;
        case DSP_I16: case DSP_U16: return 2;
        case DSP_BOOL: return 1;
        default: return 0;
    }
}

extern "C" const char* dsp_fatal_message(int code) {
    switch (code) {
        case DSP_E_PZ_NAN: return "Pole-zero filter produced nans in output.";
        case DSP_E_DPZ_SHORT: return "The length of the waveform must be larger than 3 for the filter to work safely";
        case DSP_E_TRAP_RISE: return "The number of samples in the rise section must be positive";
        case DSP_E_TRAP_FLAT: return "The number of samples in the flat section must be positive";
        case DSP_E_TRAP_FALL: return "The number of samples in the fall section must be positive";
        case DSP_E_TRAP_WIDE: return "The trapezoid width is wider than the waveform";
        case DSP_E_FTP_INT: return "fixed_time_pickoff requires integer t_in when using mode 'i'";
        case DSP_E_FTP_MODE: return "Unrecognized interpolation mode";
        case DSP_E_TPT_START_INT: return "The starting index must be an integer";
        case DSP_E_TPT_WALK_INT: return "The search direction must be an integer";
        case DSP_E_TPT_RANGE: return "The starting index is out of range";
        case DSP_E_CONV_LONG: return "The filter is longer than the input waveform";
        case DSP_E_CONV_OUTLEN: return "Output waveform has the wrong length for this convolution mode";
        case DSP_E_CONV_MODE: return "Invalid mode";
        case DSP_E_DWT_LEVEL: return "The level must be a positive integer";
        case DSP_E_DWT_OUTLEN: return "Output waveform has the wrong length for this wavelet level";
        case DSP_E_ZERODIV: return "division by zero";
        case DSP_E_WINDOW_LONG: return "The windowed waveform must be smaller than the input waveform";
        case DSP_E_AVGCUR_RANGE: return "length is out of range, must be between 0 and the length of the waveform";
        case DSP_E_TPO_INT: return "The pick-off index must be an integer";
        case DSP_E_UPSAMPLE: return "Upsample must be greater than 0";
        case DSP_E_MW_LEN_INT: return "The length of the moving window must be an integer";
        case DSP_E_MW_NUM_INT: return "The number of moving windows must be an integer";
        case DSP_E_MW_LEN_RANGE: return "The length of the moving window is out of range";
        case DSP_E_MW_NUM_NEG: return "The number of moving windows much be positive";
        default: return "";
    }
}

#define fail dsp_fail
#define elem_size dsp_elem_size

// ip[0] of ELEMENTWISE / SCALAR_FUNC: a DSP_FN_* code; the integer loops carry their type (8, 16 or 32 bits; 32 only in the float64 chain,
// whose values hold every 32-bit integer), the float ones nothing
static void note(ChainPlan* ch, const char* fmt, ...) {
    if (!ch->note.empty()) return;  // (the first reason stands)
    char buf[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    ch->note = buf;
}

static bool fn_code_ok(int ip0, bool f64, bool i64 = false) {
    const int code = DSP_FN_CODE(ip0), bits = DSP_FN_INT_BITS(ip0);
    if (ip0 < 0 || code > DSP_FN_LAST || (ip0 >> 17) != 0) return false;
    // (64 bits outside an integer program: the float64 chain's waveform loops, exact while results stay below 2^53 -- the caller's promise)
    if (code >= DSP_FN_IADD && code <= DSP_FN_ICAST) return bits == 8 || bits == 16 || (bits == 32 && (f64 || i64)) || (bits == 64 && (i64 || (f64 && code != DSP_FN_ICAST)));
    if (i64) {  // an integer program: no float arithmetic; a comparison may name its loop's type (uint64: unsigned)
        if (code <= DSP_FN_DIV || code == DSP_FN_NEG || code == DSP_FN_FLOORDIV || code >= DSP_FN_RINT) return false;
        return (ip0 >> 8) == 0 || bits == 8 || bits == 16 || bits == 32 || bits == 64;
    }
    return (ip0 >> 8) == 0;
}

// Is the program  LOAD s; [MIN_MAX of s;] [BL_SUBTRACT s <- s;]  POLE_ZERO s <- s;  STORE s  [+ STORE_SCALARs of MIN_MAX's values]  on 16-byte aligned rows
// (dsp_pz.hip)?  The MIN_MAX is that of the rows as they are read: every Ge recipe asks for tp_min / tp_max / wf_min / wf_max of the raw waveform.
static bool match_pz_rows_shape(ChainPlan* ch, const dsp_op* ops, int n_ops, const dsp_io_desc* io, const int32_t* slot_len, int n_slots,
                                const std::vector<int>& dev_index, bool f64) {
    if (f64 || n_slots != 1 || n_ops < 3 || n_ops > 9 || ops[0].opcode != DSP_OP_LOAD) return false;
    const dsp_op& ld = ops[0];
    const dsp_io_desc& w = io[ld.io];
    const int len = slot_len[ld.dst];
    if ((w.dtype != DSP_F32 && w.dtype != DSP_I16 && w.dtype != DSP_U16) || ld.ip[0] != 0 || ld.ip[1] != 0 || w.len != len || len % 8 != 0 || len < 8)
        return false;
    const int es = w.dtype == DSP_F32 ? 4 : 2;
    if ((w.row_stride * es) % 16 != 0 || (w.offset * es) % 16 != 0) return false;
    PzArgs& A = ch->pz;
    memset(&A, 0, sizeof A);
    ch->pio_bl = -1;
    for (int k = 0; k < 4; ++k) ch->pio_mm[k] = -1;
    int i = 1, mm_reg = -1;
    if (ops[i].opcode == DSP_OP_MIN_MAX) {  // min_max of the rows as they are read: the kernel streams them anyway
        if (ops[i].src != ld.dst) return false;
        mm_reg = ops[i++].dst;
        A.mm_on = 1;
    }
    // the stores of its four values, wherever they stand behind it
    std::vector<dsp_op> rest(ops, ops + i);
    int pz_at = -1;  // (the pole-zero op's place in the caller's program: its device op holds the constants)
    for (int k = i; k < n_ops; ++k) {
        const dsp_op& o = ops[k];
        if (o.opcode == DSP_OP_POLE_ZERO && pz_at < 0) pz_at = k;
        if (o.opcode != DSP_OP_STORE_SCALAR) {
            rest.push_back(o);
            continue;
        }
        const int which = o.ip[0] - mm_reg;
        if (mm_reg < 0 || which < 0 || which > 3 || ch->pio_mm[which] >= 0 || io[o.io].dtype != DSP_F32) return false;
        ch->pio_mm[which] = o.io;
        A.mm_stride[which] = io[o.io].row_stride;
    }
    if (pz_at < 0 || (int)rest.size() < i + 2) return false;
    ops = rest.data();  // (the program without those stores)
    n_ops = (int)rest.size();
    if (ops[i].opcode == DSP_OP_BL_SUBTRACT) {
        const dsp_op& bs = ops[i++];
        if (bs.dst != ld.dst || bs.src != ld.dst || bs.ip[0] != 0) return false;
        if (bs.sp[0].kind == DSP_ARG_INPUT && io[bs.sp[0].index].dtype == DSP_F32) {
            ch->pio_bl = bs.sp[0].index;
            A.bl_stride = io[ch->pio_bl].row_stride;
        } else if (bs.sp[0].kind == DSP_ARG_CONST) {
            A.bl_const = (float)bs.sp[0].value;
        } else {
            return false;
        }
        A.sub_mode = 1;
    }
    if (i + 2 != n_ops || ops[i].opcode != DSP_OP_POLE_ZERO || ops[i + 1].opcode != DSP_OP_STORE) return false;
    const dsp_op &pz = ops[i], &st = ops[i + 1];
    const bool tau_col = pz.sp[0].kind == DSP_ARG_INPUT && io[pz.sp[0].index].dtype == DSP_F32;
    if (pz.src != ld.dst || pz.dst != ld.dst || (pz.sp[0].kind != DSP_ARG_CONST && !tau_col) || st.src != ld.dst) return false;
    ch->pio_tau = tau_col ? pz.sp[0].index : -1;
    if (tau_col) A.tau_stride = io[ch->pio_tau].row_stride;
    const dsp_io_desc& o = io[st.io];
    if (o.dtype != DSP_F32 || o.len != len || o.row_stride % 4 != 0 || o.offset % 4 != 0) return false;
    const DevOp& dpz = ch->host.ops[dev_index[pz_at]];
    A.c = dpz.fc[0];
    A.tau_nan = dpz.ic[0];
    A.wf_stride = w.row_stride;
    A.wf_offset = w.offset;
    A.len = len;
    A.in_kind = w.dtype == DSP_F32 ? 0 : (w.dtype == DSP_I16 ? 1 : 2);
    A.out_stride = o.row_stride;
    ch->pio_wf = ld.io;
    ch->pio_out = st.io;
    return true;
}

// Does the program only read per-event values off rows (dsp_reduce.hip)?
//   LOAD s;  then any of  MIN_MAX of s (once),  AMAX of s (once),  PICKOFF of s at a constant integral time (fixed_time_pickoff, or the plain
//   sample wf[k]; up to DSP_REDUCE_PICKS),  TIME_POINT_THRESH of s from a constant sample or from MIN_MAX's t_min / t_max (up to
//   DSP_REDUCE_WALKS);  then STORE_SCALARs of the registers those made, float32 columns
// The reductions themselves: ops[first ..] on waveform slot `slot` of `len` samples, then the stores; fills A and ch->dio_* (dsp_fir_runs.hip
// runs the same on the waveform it has just filtered)
static bool match_reduce_ops(ChainPlan* ch, const dsp_op* ops, int first, int n_ops, int slot, int len, const dsp_io_desc* io, ReduceArgs& A) {
    memset(&A, 0, sizeof A);
    int reg_of_out[5] = {-1, -1, -1, -1, -1}, reg_of_pick[DSP_REDUCE_PICKS] = {-1, -1, -1, -1};
    int reg_of_walk[DSP_REDUCE_WALKS], walk_thr_io[DSP_REDUCE_WALKS], walk_ts_io[DSP_REDUCE_WALKS];
    for (int k = 0; k < DSP_REDUCE_WALKS; ++k) reg_of_walk[k] = walk_thr_io[k] = walk_ts_io[k] = -1;
    // thresholds that are a fraction of a per-event column: SCALAR_AFFINE d <- column * constant + 0 in front of the walk that reads d
    struct Scaled { int reg, io; float factor; };
    std::vector<Scaled> scaled;
    int n_pick = 0, n_walk = 0, i = first;
    for (; i < n_ops; ++i) {
        const dsp_op& o = ops[i];
        if (o.opcode == DSP_OP_MIN_MAX && o.src == slot && reg_of_out[0] < 0) {
            for (int k = 0; k < 4; ++k) reg_of_out[k] = o.dst + k;
        } else if (o.opcode == DSP_OP_AMAX && o.src == slot && reg_of_out[4] < 0) {
            reg_of_out[4] = o.dst;
        } else if (o.opcode == DSP_OP_PICKOFF && o.src == slot && n_pick < DSP_REDUCE_PICKS && o.sp[0].kind == DSP_ARG_CONST && (o.ip[1] == 0 || o.ip[1] == 1)) {
            const double t = (double)(float)o.sp[0].value;
            if (!(t == std::floor(t)) || std::fabs(t) > 1e9) return false;  // (between samples: the interpolating modes stay with the program)
            if (o.ip[1] == 1 && (t < 0 || t >= len)) return false;
            reg_of_pick[n_pick] = o.dst;
            A.pick_at[n_pick] = (t >= 0 && t <= len - 1) ? (int)t : -1;  // fixed_time_pickoff.py:68-74
            A.pick_rule[n_pick] = o.ip[1] == 0;
            ++n_pick;
        } else if (o.opcode == DSP_OP_SCALAR_AFFINE && o.sp[0].kind == DSP_ARG_INPUT && io[o.sp[0].index].dtype == DSP_F32 && o.sp[1].kind == DSP_ARG_CONST &&
                   o.sp[2].kind == DSP_ARG_CONST && o.sp[2].value == 0.0) {
            scaled.push_back({o.dst, o.sp[0].index, (float)o.sp[1].value});
        } else if (o.opcode == DSP_OP_TIME_POINT_THRESH && o.src == slot && n_walk < DSP_REDUCE_WALKS && o.sp[2].kind == DSP_ARG_CONST &&
                   (o.sp[2].value == 0.0 || o.sp[2].value == 1.0)) {
            // threshold: a constant, a float32 column or a fraction of one; start: a constant sample inside the waveform, where MIN_MAX found an
            // extreme (an integer inside the waveform by construction -- the checks of time_point_thresh.py:67-74 cannot fail), a float32
            // column or where an earlier walk of this program ended (checked per event, as the processor does)
            if (o.sp[0].kind == DSP_ARG_INPUT && io[o.sp[0].index].dtype == DSP_F32) {
                walk_thr_io[n_walk] = o.sp[0].index;
                A.walk_thr_stride[n_walk] = io[o.sp[0].index].row_stride;
            } else if (o.sp[0].kind == DSP_ARG_CONST) {
                A.walk_thr_const[n_walk] = (float)o.sp[0].value;
            } else if (o.sp[0].kind == DSP_ARG_REG) {
                const Scaled* sc = nullptr;
                for (const Scaled& c : scaled)
                    if (c.reg == o.sp[0].index) sc = &c;  // (the latest value of the register)
                if (!sc) return false;
                walk_thr_io[n_walk] = sc->io;
                A.walk_thr_stride[n_walk] = io[sc->io].row_stride;
                A.walk_thr_factor[n_walk] = sc->factor;
                A.walk_thr_scaled[n_walk] = 1;
            } else {
                return false;
            }
            int from_walk = -1;
            for (int k = 0; k < n_walk; ++k)
                if (o.sp[1].kind == DSP_ARG_REG && reg_of_walk[k] == o.sp[1].index) from_walk = k;
            if (o.sp[1].kind == DSP_ARG_REG && reg_of_out[0] >= 0 && (o.sp[1].index == reg_of_out[0] || o.sp[1].index == reg_of_out[1])) {
                A.walk_from[n_walk] = o.sp[1].index == reg_of_out[0] ? 1 : 2;
            } else if (o.sp[1].kind == DSP_ARG_CONST) {
                const double t = (double)(float)o.sp[1].value;
                if (!(t == std::floor(t)) || t < 0 || t >= len) return false;
                A.walk_start[n_walk] = (int)t;
            } else if (o.sp[1].kind == DSP_ARG_INPUT && io[o.sp[1].index].dtype == DSP_F32) {
                A.walk_from[n_walk] = 3;
                walk_ts_io[n_walk] = o.sp[1].index;
                A.walk_ts_stride[n_walk] = io[o.sp[1].index].row_stride;
            } else if (from_walk >= 0) {
                A.walk_from[n_walk] = 4 + from_walk;
            } else {
                return false;
            }
            A.walk_forward[n_walk] = o.sp[2].value == 1.0;
            reg_of_walk[n_walk] = o.dst;
            ++n_walk;
        } else {
            break;
        }
    }
    if (i == first || i == n_ops) return false;
    for (int k = 0; k < DSP_REDUCE_WALKS; ++k) {
        ch->dio_walk[k] = -1;
        ch->dio_walk_thr[k] = walk_thr_io[k];
        ch->dio_walk_ts[k] = walk_ts_io[k];
    }
    A.n_walks = n_walk;
    for (const Scaled& c : scaled)  // (a fraction that something other than a walk reads -- a store -- is the program's business)
        for (int j = i; j < n_ops; ++j)
            if (ops[j].opcode == DSP_OP_STORE_SCALAR && ops[j].ip[0] == c.reg) return false;
    for (int k = 0; k < 5; ++k) ch->dio_out[k] = -1;
    for (int k = 0; k < DSP_REDUCE_PICKS; ++k) ch->dio_pick[k] = -1;
    for (; i < n_ops; ++i) {
        const dsp_op& o = ops[i];
        if (o.opcode != DSP_OP_STORE_SCALAR || io[o.io].dtype != DSP_F32) return false;
        bool placed = false;
        for (int k = 0; k < 5 && !placed; ++k)
            if (reg_of_out[k] == o.ip[0] && ch->dio_out[k] < 0) {
                ch->dio_out[k] = o.io;
                A.out_stride[k] = io[o.io].row_stride;
                placed = true;
            }
        for (int k = 0; k < n_pick && !placed; ++k)
            if (reg_of_pick[k] == o.ip[0] && ch->dio_pick[k] < 0) {
                ch->dio_pick[k] = o.io;
                A.pick_stride[k] = io[o.io].row_stride;
                placed = true;
            }
        for (int k = 0; k < n_walk && !placed; ++k)
            if (reg_of_walk[k] == o.ip[0] && ch->dio_walk[k] < 0) {
                ch->dio_walk[k] = o.io;
                A.walk_stride[k] = io[o.io].row_stride;
                placed = true;
            }
        if (!placed) return false;  // (a register stored twice, or one nothing here made)
    }
    for (int k = 0; k < n_walk; ++k) {
        bool feeds = false;
        for (int j = k + 1; j < n_walk; ++j) feeds |= A.walk_from[j] == 4 + k;
        if (ch->dio_walk[k] < 0 && !feeds) return false;  // (a walk nobody stores or starts from: the program's business)
    }
    A.len = len;
    // does anything need the whole row?  The extremes, a pick-off's NaN rule -- or the walks' own NaN rule unless the caller says a row is NaN
    // from its first sample on or not at all (the caller of match_reduce_ops sets need_stream = 1 where it has no such promise)
    A.need_stream = (reg_of_out[0] >= 0 || reg_of_out[4] >= 0 || n_pick > 0) ? 1 : 0;
    return true;
}

static bool match_reduce_shape(ChainPlan* ch, const dsp_op* ops, int n_ops, const dsp_io_desc* io, const int32_t* slot_len, bool f64) {
    if (f64 || n_ops < 3 || ops[0].opcode != DSP_OP_LOAD) return false;
    const dsp_op& ld = ops[0];
    const dsp_io_desc& w = io[ld.io];
    if ((w.dtype != DSP_F32 && w.dtype != DSP_I16 && w.dtype != DSP_U16) || ld.ip[0] != 0 || ld.ip[1] != 0 || w.len < 1 || slot_len[ld.dst] != w.len)
        return false;
    ReduceArgs& A = ch->red;
    if (!match_reduce_ops(ch, ops, 1, n_ops, ld.dst, w.len, io, A)) return false;
    if (!(ld.ip[2] & 1) && w.dtype == DSP_F32) A.need_stream = 1;  // (float rows without the promise of LOAD ip[2]: a NaN may sit anywhere)
    const int es = w.dtype == DSP_F32 ? 4 : 2;
    A.wf_stride = w.row_stride;
    A.wf_offset = w.offset;
    ch->red_dtype = w.dtype;
    ch->red_vec = (w.row_stride * es) % 16 == 0 && (w.offset * es) % 16 == 0 && (w.len * es) % 16 == 0;
    ch->dio_wf = ld.io;
    return true;
}

// Does the program have the shape of the current-branch kernel (dsp_current.hip)?
//   LOAD s0;  WINDOWER s1 <- s0 (start: constant or float32 column);  AVG_CURRENT s2 <- s1;  UPSAMPLER s3 <- s2;
//   MOVING_WINDOW_MULTI d <- s3 (3 windows, alternating);  MIN_MAX of d;  STORE_SCALARs of its four registers
static bool match_current_shape(ChainPlan* ch, const dsp_op* ops, int n_ops, const dsp_io_desc* io, const int32_t* slot_len, bool f64) {
    if (f64 || n_ops < 7) return false;
    const dsp_op &ld = ops[0], &wi = ops[1], &ac = ops[2], &up = ops[3], &mw = ops[4], &mm = ops[5];
    if (ld.opcode != DSP_OP_LOAD || wi.opcode != DSP_OP_WINDOWER || ac.opcode != DSP_OP_AVG_CURRENT || up.opcode != DSP_OP_UPSAMPLER ||
        mw.opcode != DSP_OP_MOVING_WINDOW_MULTI || mm.opcode != DSP_OP_MIN_MAX)
        return false;
    if (wi.src != ld.dst || ac.src != wi.dst || up.src != ac.dst || mw.src != up.dst || mm.src != mw.dst) return false;
    const dsp_io_desc& w = io[ld.io];
    if (w.dtype != DSP_F32 || ld.ip[0] != 0 || ld.ip[1] != 0 || (w.row_stride % 4) != 0 || (w.offset % 4) != 0 || (w.len % 4) != 0) return false;
    CurrentArgs& A = ch->cur;
    memset(&A, 0, sizeof A);
    A.wf_stride = w.row_stride;
    A.wf_offset = w.offset;
    A.n_in = w.len;
    A.scan_rows = (ld.ip[2] & 1) ? 0 : 1;
    if (wi.sp[0].kind == DSP_ARG_INPUT && io[wi.sp[0].index].dtype == DSP_F32) {
        ch->cio_t0 = wi.sp[0].index;
        A.t0_stride = io[ch->cio_t0].row_stride;
    } else if (wi.sp[0].kind == DSP_ARG_CONST) {
        A.t0_const = (float)wi.sp[0].value;
    } else {
        return false;
    }
    A.win_len = slot_len[wi.dst];
    if (A.win_len < 2 || A.win_len >= A.n_in) return false;
    // avg_current: an integer-valued length inside the window
    if (ac.sp[0].kind != DSP_ARG_CONST) return false;
    const float acl = (float)ac.sp[0].value;
    if (!(acl >= 1.0f) || std::floor(acl) != acl || acl >= (float)A.win_len) return false;
    A.ac_lag = (int)acl;
    A.ac_length = acl;
    A.n_c = A.win_len - A.ac_lag;
    if (slot_len[ac.dst] != A.n_c) return false;
    // upsampler: a factor in {1, 2, 4, 8, 16}, every output sample reached by an input sample
    if (up.sp[0].kind != DSP_ARG_CONST) return false;
    const float upf = (float)up.sp[0].value;
    int shift = -1;
    for (int k = 0; k <= 4; ++k)
        if (upf == (float)(1 << k)) shift = k;
    if (shift < 0) {
        note(ch, "the current branch with an upsampling factor of %g: the lane-per-waveform kernel takes 1, 2, 4, 8 or 16", (double)upf);
        return false;
    }
    A.up_shift = shift;
    A.up_half = (1 << shift) / 2;
    A.n_up = slot_len[up.dst];
    if (A.n_up < 32 || A.n_up % 16 != 0 || ((A.n_up - 1 + A.up_half) >> shift) >= A.n_c) {
        if (A.n_up % 16 != 0) note(ch, "the current branch with %d upsampled samples: the lane-per-waveform kernel takes a multiple of 16", A.n_up);
        return false;
    }
    // moving_window_multi: three alternating windows whose length is a multiple of 16 samples
    if (mw.sp[0].kind != DSP_ARG_CONST || mw.ip[0] != 0 || mw.ip[1] != 3 || slot_len[mw.dst] != A.n_up) {
        if (mw.sp[0].kind == DSP_ARG_CONST && (mw.ip[0] != 0 || mw.ip[1] != 3))
            note(ch, "the current branch with %d moving windows of type %d: the lane-per-waveform kernel takes three alternating ones", mw.ip[1], mw.ip[0]);
        return false;
    }
    const float mal = (float)mw.sp[0].value;
    if (!(mal >= 16.0f) || std::floor(mal) != mal || mal > 112.0f || ((int)mal % 16) != 0 || (int)mal >= A.n_up) {
        note(ch, "the current branch with moving windows of %g samples: the lane-per-waveform kernel takes multiples of 16 up to 112", (double)mal);
        return false;
    }
    A.ma_len = (int)mal;
    A.ma_length = mal;
    for (int i = 6; i < n_ops; ++i) {
        const dsp_op& o = ops[i];
        if (o.opcode != DSP_OP_STORE_SCALAR || io[o.io].dtype != DSP_F32) return false;
        const int k = o.ip[0] - mm.dst;
        if (k < 0 || k > 3 || ch->cio_out[k] >= 0) return false;
        ch->cio_out[k] = o.io;
        A.out_stride[k] = io[o.io].row_stride;
    }
    A.scratch_per_wave = (int64_t)(A.n_c + 2 * (A.n_up / 16)) * 64;
    ch->cio_wf = ld.io;
    ch->cur_lds_bytes = dsp_internal_current_lds_bytes(A.ma_len);
    return true;
}

static bool match_fir_shape(ChainPlan* ch, const dsp_op* ops, int n_ops, const dsp_io_desc* io, const int32_t* slot_len, int n_slots, bool f64) {
    const DevProgram& P = ch->host;
    if (f64 || n_ops < 3 || n_slots != 1 || ops[0].opcode != DSP_OP_LOAD) return false;
    const dsp_op& ld = ops[0];
    const int s = ld.dst, wdt = io[ld.io].dtype, n = slot_len[s];
    if (wdt != DSP_F32 && wdt != DSP_I16 && wdt != DSP_U16) return false;
    const int es = wdt == DSP_F32 ? 4 : 2, align = wdt == DSP_F32 ? 16 : 8;  // 4 samples per staging load
    if ((io[ld.io].row_stride * es) % align != 0 || (io[ld.io].offset * es) % align != 0) return false;
    FirArgs& A = ch->fir;
    memset(&A, 0, sizeof A);
    int i = 1;
    if (ops[i].opcode == DSP_OP_BL_SUBTRACT) {
        const dsp_op& bs = ops[i++];
        if (bs.dst != s || bs.src != s) return false;
        if (bs.sp[0].kind == DSP_ARG_INPUT && io[bs.sp[0].index].dtype == DSP_F32) {
            ch->fio_bl = bs.sp[0].index;
            A.bl_stride = io[ch->fio_bl].row_stride;
        } else if (bs.sp[0].kind == DSP_ARG_CONST) {
            A.bl_const = (float)bs.sp[0].value;
        } else {
            return false;
        }
        A.sub_mode = 1;
    }
    int regs[DSP_FIR_MAXK], nk = 0, max_m = 0;
    for (; i < n_ops && ops[i].opcode == DSP_OP_CONVOLVE_AMAX; ++i) {
        const dsp_op& o = ops[i];
        if (nk == DSP_FIR_MAXK || o.src != s || o.ip[0] != 'v' || o.ip[1] != 0 || io[o.io].dtype != DSP_F32) return false;
        const int m = o.ip[3] > 0 ? o.ip[3] : io[o.io].len, p = n - m + 1;
        if (m < 64 || p < 1 || p > 320 || o.ip[2] != p) {  // (a short kernel is the VM's business)
            if (m < 64) note(ch, "a %d-tap 'valid' convolution and its maximum: the matrix-core FIR takes kernels of 64 taps and more", m);
            else if (p > 320) note(ch, "'valid' convolution with %d outputs and their maximum: the matrix-core FIR with a maximum takes up to 320", p);
            return false;
        }
        ch->fio_taps[nk] = o.io;
        A.m[nk] = m;
        A.p[nk] = p;
        regs[nk] = o.dst;
        if (m > max_m) max_m = m;
        ++nk;
    }
    if (nk == 0) return false;
    for (; i < n_ops; ++i) {
        const dsp_op& o = ops[i];
        if (o.opcode != DSP_OP_STORE_SCALAR || io[o.io].dtype != DSP_F32) return false;
        int k = 0;
        while (k < nk && (regs[k] != o.ip[0] || ch->fio_out[k] >= 0)) ++k;
        if (k == nk) return false;
        ch->fio_out[k] = o.io;
        A.out_stride[k] = io[o.io].row_stride;
    }
    for (int k = 0; k < nk; ++k)
        if (ch->fio_out[k] < 0) return false;
    int kend = n < 319 + max_m ? n : 319 + max_m;
    kend = ((kend + 31) / 32) * 32;
    if (io[ld.io].offset + kend > io[ld.io].row_stride) return false;  // the staging loads run to the end of the last 32-sample stage
    A.wf_stride = io[ld.io].row_stride;
    A.wf_offset = io[ld.io].offset;
    A.n = n;
    A.in_kind = wdt == DSP_F32 ? 0 : (wdt == DSP_I16 ? 1 : 2);
    A.n_kernels = nk;
    A.kend = kend;
    A.scan_before = ld.ip[0];
    A.scan_after = ld.ip[1];
    ch->fio_wf = ld.io;
    ch->fir_lds_bytes = dsp_internal_fir_mfma_lds_bytes(kend);
    (void)P;
    return ch->fir_lds_bytes <= 80 * 1024;
}

// The matrix-core FIR with its output kept (dsp_fir_store_kernel):  LOAD s; [BL_SUBTRACT s <- s]; CONVOLVE d <- s (any mode, >= 64 finite taps);
// STORE d.  What whole recipes run ahead of their program for the filters other processors read (the t0 filter).
static bool match_fir_store_shape(ChainPlan* ch, const dsp_op* ops, int n_ops, const dsp_io_desc* io, const int32_t* slot_len, int n_slots, bool f64) {
    if (f64 || n_ops < 3 || n_ops > 4 || n_slots != 2 || ops[0].opcode != DSP_OP_LOAD) return false;
    const dsp_op& ld = ops[0];
    const int s = ld.dst, wdt = io[ld.io].dtype, n = slot_len[s];
    if (wdt != DSP_F32 && wdt != DSP_I16 && wdt != DSP_U16) return false;
    const int es = wdt == DSP_F32 ? 4 : 2, align = wdt == DSP_F32 ? 16 : 8;
    if ((io[ld.io].row_stride * es) % align != 0 || (io[ld.io].offset * es) % align != 0) return false;
    FirArgs& A = ch->fir;
    memset(&A, 0, sizeof A);
    ch->fio_bl = -1;
    int i = 1;
    if (ops[i].opcode == DSP_OP_BL_SUBTRACT) {
        const dsp_op& bs = ops[i++];
        if (bs.dst != s || bs.src != s || bs.ip[0] != 0) return false;
        if (bs.sp[0].kind == DSP_ARG_INPUT && io[bs.sp[0].index].dtype == DSP_F32) {
            ch->fio_bl = bs.sp[0].index;
            A.bl_stride = io[ch->fio_bl].row_stride;
        } else if (bs.sp[0].kind == DSP_ARG_CONST) {
            A.bl_const = (float)bs.sp[0].value;
        } else {
            return false;
        }
        A.sub_mode = 1;
    }
    if (i + 2 != n_ops || ops[i].opcode != DSP_OP_CONVOLVE || ops[i + 1].opcode != DSP_OP_STORE) return false;
    const dsp_op& o = ops[i];
    const dsp_op& st = ops[i + 1];
    if (o.src != s || o.dst == s || (o.ip[1] & 3) != 0 || io[o.io].dtype != DSP_F32 || st.src != o.dst || io[st.io].dtype != DSP_F32) return false;
    const int m = o.ip[3] > 0 ? o.ip[3] : io[o.io].len;
    if (m < 64 || m > n) {
        if (m < 64) note(ch, "a %d-tap convolution: the matrix-core FIR takes kernels of 64 taps and more", m);
        return false;
    }
    const int mode = o.ip[0];
    const int P = mode == 'v' ? n - m + 1 : (mode == 's' ? n : (mode == 'f' ? n + m - 1 : -1));
    if (P < 1 || slot_len[o.dst] != P || io[st.io].len != P) return false;
    A.store = 1;
    A.dshift = mode == 'v' ? 0 : (mode == 's' ? m / 2 : m - 1);
    A.m[0] = m;
    A.p[0] = P;
    A.n_kernels = 1;
    A.out_stride[0] = io[st.io].row_stride;
    A.wf_stride = io[ld.io].row_stride;
    A.wf_offset = io[ld.io].offset;
    A.n = n;
    A.in_kind = wdt == DSP_F32 ? 0 : (wdt == DSP_I16 ? 1 : 2);
    A.kend = ((320 + m - 1 + 3 + 31) / 32) * 32;  // the longest window of a 320-column tile (dsp_fir_mfma.hip)
    A.scan_before = ld.ip[0];
    A.scan_after = ld.ip[1];
    ch->fio_wf = ld.io;
    ch->fio_taps[0] = o.io;
    ch->fio_out[0] = st.io;
    ch->fir_lds_bytes = dsp_internal_fir_store_lds_bytes(A.kend);
    return ch->fir_lds_bytes <= 80 * 1024;
}

// The run-length FIR (dsp_fir_runs.hip):  LOAD s (float32 rows);  CONVOLVE d <- s with ip[2] = 1 (the caller found the kernel piecewise constant:
// at most DSP_FIR_RUNS_MAX runs) and at most DSP_FIR_RUNS_MAX_TAPS finite taps, any mode;  [STORE d];  [the reductions of match_reduce_ops on d].
// At least one of the two: a filtered waveform nobody keeps and nobody reads is no program.
static bool match_fir_runs_shape(ChainPlan* ch, const dsp_op* ops, int n_ops, const dsp_io_desc* io, const int32_t* slot_len, int n_slots, bool f64) {
    if (f64 || n_ops < 3 || n_slots != 2 || ops[0].opcode != DSP_OP_LOAD || ops[1].opcode != DSP_OP_CONVOLVE) return false;
    const dsp_op &ld = ops[0], &o = ops[1];
    const dsp_io_desc& w = io[ld.io];
    const int s = ld.dst, n = slot_len[s];
    if (o.ip[2] != 1 || (o.ip[1] & 3) != 0 || o.src != s || o.dst == s || io[o.io].dtype != DSP_F32) return false;
    if (w.dtype != DSP_F32 || ld.ip[0] != 0 || ld.ip[1] != 0 || w.len != n || n % 8 != 0 || n < 8 || w.row_stride % 4 != 0 || w.offset % 4 != 0) {
        note(ch, "a piecewise-constant kernel on rows that are not float32, 16-byte aligned and a multiple of 8 samples long: the run-length FIR kernel takes those");
        return false;
    }
    const int m = o.ip[3] > 0 ? o.ip[3] : io[o.io].len;
    if (m < 1 || m > n || m > DSP_FIR_RUNS_MAX_TAPS) return false;
    const int mode = o.ip[0];
    const int P = mode == 'v' ? n - m + 1 : (mode == 's' ? n : (mode == 'f' ? n + m - 1 : -1));
    if (P < 1 || slot_len[o.dst] != P) return false;
    FirRunsArgs& A = ch->runs;
    memset(&A, 0, sizeof A);
    // the STORE of the filtered waveform, wherever it stands behind the CONVOLVE; everything else must be the reductions and their stores
    ch->uio_out = -1;
    std::vector<dsp_op> rest;
    for (int i = 2; i < n_ops; ++i) {
        if (ops[i].opcode == DSP_OP_STORE) {
            const dsp_op& st = ops[i];
            if (A.keep || st.src != o.dst || io[st.io].dtype != DSP_F32 || io[st.io].len != P) return false;
            ch->uio_out = st.io;
            A.keep = 1;
            A.out_stride = io[st.io].row_stride;
        } else {
            rest.push_back(ops[i]);
        }
    }
    for (int k = 0; k < 5; ++k) ch->dio_out[k] = -1;
    for (int k = 0; k < DSP_REDUCE_PICKS; ++k) ch->dio_pick[k] = -1;
    for (int k = 0; k < DSP_REDUCE_WALKS; ++k) ch->dio_walk[k] = ch->dio_walk_thr[k] = ch->dio_walk_ts[k] = -1;
    if (!rest.empty()) {
        if (!match_reduce_ops(ch, rest.data(), 0, (int)rest.size(), o.dst, P, io, A.red)) return false;
        A.red.need_stream = 1;
        A.has_red = 1;
    } else if (!A.keep) {
        return false;
    }
    A.wf_stride = w.row_stride;
    A.wf_offset = w.offset;
    A.n = n;
    A.m = m;
    A.p = P;
    A.start = mode == 'v' ? m - 1 : (mode == 's' ? (m - 1) / 2 : 0);
    if (!A.keep) A.out_stride = (P + 63) & ~63;  // the pitch of the wavefronts' scratch rows
    ch->uio_wf = ld.io;
    ch->uio_taps = o.io;
    return true;
}

// Does the program have the shape of the lane-per-waveform kernel?  Fills ch->rows / ch->rio_* and returns true if so.
//   LOAD s;  [BL_SUBTRACT s <- s];  POLE_ZERO | DOUBLE_POLE_ZERO s <- s;  then in any order: one TRAP_REDUCE of s, at most one DWT_HAAR of s
//   into a slot that is stored, STORE_SCALARs of the reduction's registers.
static bool match_rows_shape(ChainPlan* ch, const dsp_op* ops, int n_ops, const dsp_io_desc* io, int n_io, const int32_t* slot_len, int n_slots,
                             const std::vector<int>& dev_index, bool f64) {
    const DevProgram& P = ch->host;
    if (f64 || n_ops < 3 || n_slots < 1 || n_slots > 2) return false;
    int i = 0;
    if (ops[i].opcode != DSP_OP_LOAD) return false;
    const dsp_op& ld = ops[i++];
    const int s = ld.dst;
    const int wdt = io[ld.io].dtype, len = slot_len[s];
    if (ld.ip[0] != 0 || ld.ip[1] != 0) return false;
    const bool row_layout = (wdt == DSP_F32 || wdt == DSP_I16 || wdt == DSP_U16) && P.io[ld.io].vec_ok && len % 8 == 0 && len >= 16;
    auto f32_or_const = [&](const dsp_scalar_arg& a) { return a.kind == DSP_ARG_CONST || (a.kind == DSP_ARG_INPUT && io[a.index].dtype == DSP_F32); };
    const dsp_op* bs = nullptr;
    if (ops[i].opcode == DSP_OP_BL_SUBTRACT) {
        bs = &ops[i++];
        if (bs->dst != s || bs->src != s || !f32_or_const(bs->sp[0])) return false;
    }
    if (i >= n_ops) return false;
    // (no pole-zero step at all: the rows are a waveform another program already corrected -- what whole recipes stage in HBM)
    const bool has_pz = ops[i].opcode == DSP_OP_POLE_ZERO || ops[i].opcode == DSP_OP_DOUBLE_POLE_ZERO;
    const dsp_op& pz = ops[has_pz ? i : 0];
    const DevOp& dpz = P.ops[dev_index[has_pz ? i : 0]];
    if (has_pz) {
        ++i;
        if (pz.dst != s || pz.src != s) return false;
        for (int k = 0; k < (pz.opcode == DSP_OP_POLE_ZERO ? 1 : 3); ++k)
            if (pz.sp[k].kind != DSP_ARG_CONST) {  // (per-event time constants: the VM's ops form the coefficients per row)
                note(ch, "a pole-zero time constant per event: the lane-per-waveform rows kernel takes constants");
                return false;
            }
    }
    const dsp_op *tr = nullptr, *dw = nullptr, *st_wf = nullptr;
    int tr_at = -1;
    std::vector<const dsp_op*> st_sc;
    for (; i < n_ops; ++i) {
        const dsp_op& o = ops[i];
        if (o.opcode == DSP_OP_TRAP_REDUCE && !tr && o.src == s) {
            tr = &o;
            tr_at = i;
        } else if (o.opcode == DSP_OP_DWT_HAAR && !dw && o.src == s && o.dst != s) {
            dw = &o;
        } else if (o.opcode == DSP_OP_STORE_SCALAR) {
            st_sc.push_back(&o);
        } else if (o.opcode == DSP_OP_STORE && !st_wf) {
            st_wf = &o;
        } else {
            return false;
        }
    }
    if (!tr || (dw != nullptr) != (st_wf != nullptr)) return false;
    if (tr->ip[3] >> 8) return false;  // (a pick-off or the amax-only form of the reduction: the VM's)
    if (dw && (st_wf->src != dw->dst || io[st_wf->io].dtype != DSP_F32)) return false;
    if (!row_layout) {
        note(ch, "rows of %d samples: the lane-per-waveform rows kernel takes float32 / int16 / uint16 rows of a multiple of 8 samples (16 and more) that start on 16-byte boundaries", len);
        return false;
    }
    const DevOp& dtr = P.ops[dev_index[tr_at]];
    RowsArgs& A = ch->rows;
    memset(&A, 0, sizeof A);
    A.wf_stride = io[ld.io].row_stride;
    A.wf_offset = io[ld.io].offset;
    A.len = len;
    A.in_kind = wdt == DSP_F32 ? 0 : (wdt == DSP_I16 ? 1 : 2);
    ch->rio_wf = ld.io;
    if (bs) {
        A.sub_mode = 1;
        if (bs->sp[0].kind == DSP_ARG_INPUT) {
            ch->rio_bl = bs->sp[0].index;
            A.bl_stride = io[ch->rio_bl].row_stride;
        } else {
            A.bl_const = (float)bs->sp[0].value;
        }
    }
    A.pz_kind = !has_pz ? 0 : (pz.opcode == DSP_OP_POLE_ZERO ? 1 : 2);
    A.pz_param_nan = has_pz ? dpz.ic[0] : 0;
    if (A.pz_kind == 0) {
        if (dw) return false;  // (the Haar transform of the kernel is the one of the pole-zero output)
    } else if (A.pz_kind == 1) {
        A.pz_c = dpz.fc[0];
    } else {
        A.n1 = dpz.fc[0];
        A.n2 = dpz.fc[1];
        A.d1 = dpz.fc[2];
        A.d2 = dpz.fc[3];
    }
    // trapezoid: every lag at least one block of the kernel (8 samples), the history ring within half a CU's LDS
    A.trap_kind = tr->ip[3] == DSP_OP_TRAP_FILTER ? 0 : (tr->ip[3] == DSP_OP_TRAP_NORM ? 1 : 2);
    int maxlag = 0;
    for (int k = 0; k < 3; ++k) {
        A.lag[k] = dtr.ic[k];
        if (A.lag[k] < 8) {
            note(ch, "a trapezoid with a rise or flat top of %d samples: the lane-per-waveform rows kernel takes 8 and more", A.lag[k]);
            return false;
        }
        if (A.lag[k] > maxlag) maxlag = A.lag[k];
    }
    const int R = ((maxlag + 8 + 7) / 8) * 8;  // R > largest lag + 7, a whole number of blocks
    if ((R + 8) * 256 > LDS_BYTES_PER_CU / 2) return false;
    A.ring_entries = R;
    ch->rows_lds_bytes = (R + 8) * 256;
    A.trap_all_nan = dtr.ic[9];
    A.rr = dtr.fc[0];
    A.ll = dtr.fc[1];
    A.inv_rr = 1.0 / A.rr;
    A.inv_ll = 1.0 / A.ll;
    const int rise = tr->ip[0];
    A.rise_pow2 = (rise > 0 && (rise & (rise - 1)) == 0) ? 1 : 0;
    // reductions
    const int mm = tr->dst, tpt_reg = tr->io;
    if (tpt_reg >= 0) {
        if (!f32_or_const(tr->sp[0]) || tr->sp[2].kind != DSP_ARG_CONST) return false;
        if (tr->sp[0].kind == DSP_ARG_INPUT) {
            ch->rio_thr = tr->sp[0].index;
            A.thr_stride = io[ch->rio_thr].row_stride;
        } else {
            A.thr_const = (float)tr->sp[0].value;
        }
        const double walk = (double)(float)tr->sp[2].value;
        A.walk_nan = std::isnan(walk) ? 1 : 0;
        A.walk_frac = (!A.walk_nan && std::floor(walk) != walk) ? 1 : 0;
        const bool forward = !A.walk_nan && !A.walk_frac && (long long)walk == 1;
        const dsp_scalar_arg& t = tr->sp[1];
        if (t.kind == DSP_ARG_REG) {
            if (mm < 0 || (t.index != mm && t.index != mm + 1)) return false;
            A.tpt_use_min = t.index == mm ? 1 : 0;
            A.tpt_mode = forward ? 4 : 2;
        } else {
            if (!f32_or_const(t)) return false;
            if (t.kind == DSP_ARG_INPUT) {
                ch->rio_ts = t.index;
                A.ts_stride = io[ch->rio_ts].row_stride;
            } else {
                A.ts_const = (float)t.value;
            }
            A.tpt_mode = forward ? 3 : 1;
        }
    }
    for (const dsp_op* st : st_sc) {
        const int r = st->ip[0];
        if (io[st->io].dtype != DSP_F32) return false;
        if (mm >= 0 && r >= mm && r < mm + 4) {
            if (ch->rio_mm[r - mm] >= 0) return false;  // (one column per value)
            ch->rio_mm[r - mm] = st->io;
            A.out_mm_stride[r - mm] = io[st->io].row_stride;
        } else if (tpt_reg >= 0 && r == tpt_reg && ch->rio_tpt < 0) {
            ch->rio_tpt = st->io;
            A.out_tpt_stride = io[st->io].row_stride;
        } else {
            return false;
        }
    }
    if (dw) {
        const int level = dw->ip[0], outlen = slot_len[dw->dst];
        const dsp_io_desc& d = io[st_wf->io];
        if (level < 3 || level > 8 || len % (1 << level) != 0 || outlen != (len >> level) || outlen % 4 != 0 || d.row_stride % 4 != 0 ||
            d.offset % 4 != 0 || d.len != outlen)
            return false;
        A.dwt_level = level;
        A.dwt_part = dw->ip[1];
        A.dwt_stride = d.row_stride;
        ch->rio_dwt = st_wf->io;
    }
    // a walk backward from a known start on rows that are NaN-free or NaN from the first sample on (the LOAD's promise), and nothing else
    // asked for: what lies behind the start cannot change the answer -- the group stops there (the t0 trapezoid of the Ge recipes: half a row)
    bool any_mm = false;
    for (int k = 0; k < 4; ++k) any_mm |= ch->rio_mm[k] >= 0;
    A.stop_at_start = (A.tpt_mode == 1 && !any_mm && !dw && !has_pz && (ld.ip[2] & 1) && !A.walk_nan && !A.walk_frac && !A.trap_all_nan) ? 1 : 0;
    return true;
}


static int check_slot(const DevProgram& P, int s) { return s >= 0 && s < P.n_slots; }

static void mat2_mul(const long double* a, const long double* b, long double* o) {
    long double r[4] = {a[0] * b[0] + a[1] * b[2], a[0] * b[1] + a[1] * b[3], a[2] * b[0] + a[3] * b[2], a[2] * b[1] + a[3] * b[3]};
    memcpy(o, r, sizeof r);
}

// lag geometry shared by the three trapezoids (ic/fc layout documented in dsp_vm.hip)
static int setup_trap(DevOp& d, int kind_opcode, int rise, int flat, int fall, int len, int C) {
    if (rise < 0) return DSP_E_TRAP_RISE;
    if (flat < 0) return DSP_E_TRAP_FLAT;
    int L[3];
    if (kind_opcode == DSP_OP_ASYM_TRAP) {
        if (fall < 0) return DSP_E_TRAP_FALL;
        if ((int64_t)rise + flat + fall > len) return DSP_E_TRAP_WIDE;
        if (len > 0 && (rise == 0 || (fall == 0 && rise + flat < len))) return DSP_E_ZERODIV;
        L[0] = rise;
        L[1] = rise + flat;
        L[2] = rise + flat + fall;
    } else {
        if (2 * (int64_t)rise + flat > len) return DSP_E_TRAP_WIDE;
        if (kind_opcode == DSP_OP_TRAP_NORM && len > 0 && rise == 0) return DSP_E_ZERODIV;
        L[0] = rise;
        L[1] = rise + flat;
        L[2] = 2 * rise + flat;
    }
    for (int k = 0; k < 3; ++k) {
        d.ic[k] = L[k];
        d.ic[3 + k] = L[k] / C;
        d.ic[6 + k] = L[k] % C;
    }
    // trap_filter with rise == 0 reads w_out[-1] (the NaN fill) in its first step: the whole output is NaN
    d.ic[9] = (kind_opcode == DSP_OP_TRAP_FILTER && rise == 0) ? 1 : 0;
    d.ic[10] = (rise > 0 && (rise & (rise - 1)) == 0) ? 1 : 0;  // rise is a power of two: x / rise == x * (1 / rise), exactly
    d.fc[0] = (double)rise;
    d.fc[1] = (double)fall;
    return DSP_OK;
}

// waveform slots an op reads or writes (for the lifetime analysis of the LDS packing)
static int op_slots(const dsp_op& o, int out[4]) {
    switch (o.opcode) {
        case DSP_OP_LOAD: out[0] = o.dst; return 1;
        case DSP_OP_STORE:
        case DSP_OP_TRAP_PICKOFF:
        case DSP_OP_TRAP_REDUCE:
        case DSP_OP_PICKOFF:
        case DSP_OP_TIME_POINT_THRESH:
        case DSP_OP_INTERP_TIME_POINT_THRESH:
        case DSP_OP_MEAN_BELOW:
        case DSP_OP_TRAP_WINDOW_PICKOFF:
        case DSP_OP_MIN_MAX:
        case DSP_OP_LINEAR_SLOPE_FIT:
        case DSP_OP_AMAX:
        case DSP_OP_CONVOLVE_AMAX: out[0] = o.src; return 1;
        case DSP_OP_BL_SUBTRACT:
        case DSP_OP_MIN_MAX_NORM:
        case DSP_OP_POLE_ZERO:
        case DSP_OP_DOUBLE_POLE_ZERO:
        case DSP_OP_TRAP_FILTER:
        case DSP_OP_TRAP_NORM:
        case DSP_OP_ASYM_TRAP:
        case DSP_OP_COPY:
        case DSP_OP_WINDOWER:
        case DSP_OP_AVG_CURRENT:
        case DSP_OP_UPSAMPLER:
        case DSP_OP_CONVOLVE: out[0] = o.src; out[1] = o.dst; return 2;
        case DSP_OP_DWT_HAAR: out[0] = o.src; out[1] = o.dst; out[2] = o.ip[2]; return 3;
        case DSP_OP_MOVING_WINDOW_MULTI:
            out[0] = o.src;
            out[1] = o.dst;
            if (o.ip[1] > 1 || o.ip[3] == 1) {
                out[2] = o.ip[2];
                return 3;
            }
            return 2;
        case DSP_OP_ELEMENTWISE: {
            int n = 0;
            out[n++] = o.dst;
            if (o.src >= 0) out[n++] = o.src;
            if (o.ip[1] >= 0) out[n++] = o.ip[1];
            if (o.ip[2] >= 0) out[n++] = o.ip[2];
            return n;
        }
        default: return 0;  // scalar ops
    }
}


int dsp_plan_build(ChainPlan* ch, const dsp_op* ops, int n_ops, const dsp_io_desc* io, int n_io, const int32_t* slot_len, int n_slots,
                   int n_sregs, int compute_dtype) {
    if (!ops || !ch || n_ops <= 0 || n_ops > DSP_MAX_OPS) return fail(DSP_ERR_ARG, "n_ops=%d out of range (1..%d)", n_ops, DSP_MAX_OPS);
    if (n_io < 0 || n_io > DSP_MAX_IO) return fail(DSP_ERR_ARG, "n_io=%d out of range", n_io);
    if (n_slots < 0 || n_slots > DSP_MAX_SLOTS) return fail(DSP_ERR_ARG, "n_slots=%d out of range", n_slots);
    if (n_sregs < 0 || n_sregs > DSP_MAX_SREGS) return fail(DSP_ERR_ARG, "n_sregs=%d out of range", n_sregs);
    if (compute_dtype != DSP_F32 && compute_dtype != DSP_F64 && compute_dtype != DSP_I64)
        return fail(DSP_ERR_ARG, "compute_dtype must be DSP_F32, DSP_F64 or DSP_I64");
    const bool i64 = compute_dtype == DSP_I64;  // an integer program of per-event values (dspeed_hip.h): 64-bit integer registers
    const int esz = compute_dtype == DSP_F32 ? 4 : 8;
    const bool f64 = compute_dtype == DSP_F64;
    if (i64) {
        if (n_slots != 0) return fail(DSP_ERR_ARG, "an integer program (DSP_I64) has no waveform slots");
        for (int i = 0; i < n_ops; ++i)
            if (ops[i].opcode != DSP_OP_SCALAR_FUNC && ops[i].opcode != DSP_OP_STORE_SCALAR)
                return fail(DSP_ERR_ARG, "op %d: an integer program (DSP_I64) holds SCALAR_FUNC and STORE_SCALAR ops only", i);
    }

    DevProgram& P = ch->host;
    P.n_ops = n_ops;
    P.n_slots = n_slots;
    P.n_io = n_io;
    P.n_sregs = n_sregs;
    ch->f64 = f64;
    ch->i64 = i64;

    // ---- LDS layout: [guard 2*pitch][slot 64*pitch][tail 8] per slot, then the scalar registers
    // FIR inputs are laid out linearly (no chunk pad): a slot qualifies when a CONVOLVE reads it and everything else that touches
    // it is layout-agnostic (load, store, copy, bl_subtract); the chunk-serial filters keep the padded, conflict-free layout
    bool linear[DSP_MAX_SLOTS] = {false};
    for (int s = 0; s < n_slots; ++s) {
        bool fir_in = false, only_plain = true;
        for (int i = 0; i < n_ops; ++i) {
            const dsp_op& o = ops[i];
            int touched[4];
            const int nt = op_slots(o, touched);
            bool uses = false;
            for (int k = 0; k < nt; ++k) uses |= touched[k] == s;
            if (!uses) continue;
            const bool conv = o.opcode == DSP_OP_CONVOLVE || o.opcode == DSP_OP_CONVOLVE_AMAX;
            const bool reads = o.opcode != DSP_OP_LOAD && o.src == s;
            const bool writes = o.dst == s && (o.opcode == DSP_OP_LOAD || nt >= 2);  // (ELEMENTWISE: dst is touched[0], nt >= 2)  // (one-slot ops other than LOAD only read)
            if (conv && reads) fir_in = true;
            const bool plain = o.opcode == DSP_OP_LOAD || o.opcode == DSP_OP_STORE || o.opcode == DSP_OP_COPY ||
                               o.opcode == DSP_OP_BL_SUBTRACT || o.opcode == DSP_OP_MIN_MAX_NORM || (conv && reads && !writes);
            if (!plain) only_plain = false;
        }
        linear[s] = fir_in && only_plain;
    }
    // Slots whose lifetimes (first .. last op that touches them) do not overlap share LDS: the ICPC recipe has 14 waveform variables
    // and at most four alive at a time.  Interval packing, first fit by first use; a slot that shares its region gets a
    // DSP_OP_INTERNAL_ZERO in front of its first op (guards must read 0, pads finite -- also for the next row, which finds the region
    // as the last tenant of the previous row left it).
    int first_op[DSP_MAX_SLOTS], last_op[DSP_MAX_SLOTS], foot[DSP_MAX_SLOTS], base[DSP_MAX_SLOTS];
    for (int s = 0; s < n_slots; ++s) {
        first_op[s] = n_ops;
        last_op[s] = -1;
    }
    for (int i = 0; i < n_ops; ++i) {
        int touched[4];
        const int nt = op_slots(ops[i], touched);
        for (int k = 0; k < nt; ++k) {
            const int s = touched[k];
            if (s < 0 || s >= n_slots) continue;  // (rejected by the op checks below)
            if (i < first_op[s]) first_op[s] = i;
            if (i > last_op[s]) last_op[s] = i;
        }
    }
    for (int s = 0; s < n_slots; ++s) {
        const int len = slot_len[s];
        if (len <= 0) return fail(DSP_ERR_ARG, "slot %d has length %d", s, len);
        // (40 000 float32 samples fill a CU's LDS; the bound keeps the layout arithmetic below -- and the kernels' index -> (lane, offset)
        // split, exact for indices under 2^20 -- inside int)
        if (len > (1 << 20)) return fail(DSP_ERR_TOO_LONG, "slot %d: %d samples; a waveform variable lives in LDS (at most 2^20 samples are addressable)", s, len);
        int C = (len + 63) / 64;
        C = ((C + 15) / 16) * 16;
        DevSlot& d = P.slots[s];
        d.len = len;
        d.C = C;
        d.padw = linear[s] ? 0 : 1;
        d.pitch = C + d.padw;
        d.invC = 1.0f / (float)C;
        // a short FIR kernel (the t0 filter) in 'same' / 'full' mode reads up to m - 1 samples past either end of its input: give the
        // slot a zero tail long enough that those windows need no bounds checks (below sample 0 the guard serves)
        int fir_taps = 0;
        for (int i = 0; i < n_ops; ++i)
            if ((ops[i].opcode == DSP_OP_CONVOLVE || ops[i].opcode == DSP_OP_CONVOLVE_AMAX) && ops[i].src == s && ops[i].io >= 0 &&
                ops[i].io < n_io && io[ops[i].io].kind == DSP_IO_TAPS && io[ops[i].io].len <= 1024 && io[ops[i].io].len > fir_taps)
                fir_taps = io[ops[i].io].len;
        const int tail = 40 + (fir_taps ? fir_taps + 32 : 0);
        d.zero_below = 2 * d.pitch - 8;
        d.zero_above = (len == 64 * C) ? tail - 8 : 0;
        foot[s] = 2 * d.pitch + 64 * d.pitch + tail;  // guard, chunks, tail: the pipelined loops read up to 2 groups + 1 past the last chunk
        foot[s] = ((foot[s] + 3) / 4) * 4;          // (regions stay 16-byte aligned for the wide clears)
        if (last_op[s] < 0) {                        // never used: alive throughout, so nothing is placed on top of it
            first_op[s] = 0;
            last_op[s] = n_ops - 1;
        }
    }
    int order[DSP_MAX_SLOTS];
    for (int s = 0; s < n_slots; ++s) order[s] = s;
    std::stable_sort(order, order + n_slots, [&](int a, int b) { return first_op[a] < first_op[b]; });
    int cursor = 0;
    bool shares[DSP_MAX_SLOTS] = {false};
    for (int oi = 0; oi < n_slots; ++oi) {
        const int s = order[oi];
        int at = 0;
        for (bool moved = true; moved;) {  // lowest offset where no slot alive at the same time lies
            moved = false;
            for (int oj = 0; oj < oi; ++oj) {
                const int t = order[oj];
                const bool alive_together = first_op[s] <= last_op[t] && first_op[t] <= last_op[s];
                if (alive_together && at < base[t] + foot[t] && base[t] < at + foot[s]) {
                    at = base[t] + foot[t];
                    moved = true;
                }
            }
        }
        base[s] = at;
        for (int oj = 0; oj < oi; ++oj) {
            const int t = order[oj];
            if (at < base[t] + foot[t] && base[t] < at + foot[s]) shares[s] = shares[t] = true;
        }
        P.slots[s].off = at + 2 * P.slots[s].pitch;
        if (at + foot[s] > cursor) cursor = at + foot[s];
    }
    P.sreg_off = cursor;
    cursor += ((n_sregs + 7) / 8) * 8 + 8;
    cursor = ((cursor + 3) / 4) * 4;
    P.scratch_off = cursor;
    cursor += DSP_SCRATCH_ELEMS;
    P.lds_elems_per_wave = cursor;
    ch->lds_bytes_per_wave = cursor * esz;
    if (ch->lds_bytes_per_wave > LDS_BYTES_PER_CU)
        return fail(DSP_ERR_TOO_LONG, "chain needs %d bytes of LDS per waveform; a CU has %d", ch->lds_bytes_per_wave, LDS_BYTES_PER_CU);
    // wavefronts per workgroup (1..4): the size that fits the most wavefronts into a CU's LDS and register budget (25 KB per
    // waveform: 3 per group and 2 groups = 6 wavefronts, where 4 per group would leave one group of 4); ties go to the larger group.
    // Register budget: the VM without the FIR op and 3 wavefronts per SIMD = 12 per CU, everything else 2 per SIMD = 8 per CU.
    for (int i = 0; i < n_ops; ++i) ch->has_fir |= (ops[i].opcode == DSP_OP_CONVOLVE || ops[i].opcode == DSP_OP_CONVOLVE_AMAX);
    auto pick_wpb = [&](int cap_waves) {
        int best_w = 1, best_waves = 0;
        for (int w = 1; w <= 4; ++w) {
            int groups = LDS_BYTES_PER_CU / (w * ch->lds_bytes_per_wave);
            if (groups > cap_waves / w) groups = cap_waves / w;  // whole groups only
            const int waves = groups * w;
            if (groups >= 1 && waves >= best_waves) {
                best_waves = waves;
                best_w = w;
            }
        }
        return best_w;
    };
    int wpb = pick_wpb(ch->has_fir ? 8 : 12);
    ch->classic_wpb = pick_wpb(8);
    ch->waves_per_block = wpb;
    P.waves_per_block = wpb;

    // ---- I/O bindings
    for (int k = 0; k < n_io; ++k) {
        const dsp_io_desc& a = io[k];
        DevIO& d = P.io[k];
        const int es = elem_size(a.dtype);
        if (!es) return fail(DSP_ERR_ARG, "io %d: unknown dtype %d", k, a.dtype);
        if (a.kind < DSP_IO_WF_IN || a.kind > DSP_IO_TAPS) return fail(DSP_ERR_ARG, "io %d: unknown kind %d", k, a.kind);
        if (a.len <= 0 || a.offset < 0) return fail(DSP_ERR_ARG, "io %d: bad len/offset", k);
        // rows lie row_stride elements apart and hold offset + len elements (dspeed_hip.h; 0: one row / value for every event); the bounds
        // keep the address arithmetic of the kernels and of this file inside 64 / 32 bits
        if (a.row_stride < 0 || a.row_stride > ((int64_t)1 << 40) || a.offset > (1 << 28) || a.len > (1 << 28))
            return fail(DSP_ERR_ARG, "io %d: row_stride / offset / len out of range", k);
        if ((a.kind == DSP_IO_WF_IN || a.kind == DSP_IO_WF_OUT) && (a.row_stride != 0 || a.kind == DSP_IO_WF_OUT) && (int64_t)a.offset + a.len > a.row_stride)
            return fail(DSP_ERR_ARG, "io %d: rows of %lld elements do not hold offset %d + len %d", k, (long long)a.row_stride, a.offset, a.len);
        // which rows may feed which loop: NumPy's can_cast rule as ProcessorManager applies it (processing_chain.py:1565-1572)
        if (a.kind == DSP_IO_WF_IN && !f64 && (a.dtype == DSP_I32 || a.dtype == DSP_U32 || a.dtype == DSP_F64))
            return fail(DSP_ERR_ARG, "io %d: int32/uint32/float64 rows select the float64 loop (compute_dtype DSP_F64)", k);
        const bool is_out = a.kind == DSP_IO_WF_OUT || a.kind == DSP_IO_SCALAR_OUT;
        if ((is_out || a.kind == DSP_IO_TAPS) && a.dtype != compute_dtype && !(is_out && a.dtype == DSP_BOOL) && !(i64 && a.kind == DSP_IO_SCALAR_OUT))
            return fail(DSP_ERR_ARG, "io %d: outputs have the chain's compute type or DSP_BOOL, taps the compute type", k);
        if ((a.dtype == DSP_BOOL || a.dtype == DSP_I64 || a.dtype == DSP_U64) && a.kind != DSP_IO_SCALAR_IN && a.kind != DSP_IO_SCALAR_OUT && !(a.dtype == DSP_BOOL && is_out))
            return fail(DSP_ERR_ARG, "io %d: DSP_BOOL / DSP_I64 / DSP_U64 are types of per-event columns (DSP_BOOL also of waveform outputs)", k);
        if (i64 && a.kind == DSP_IO_SCALAR_IN && (a.dtype == DSP_F32 || a.dtype == DSP_F64))
            return fail(DSP_ERR_ARG, "io %d: an integer program (DSP_I64) reads integer and DSP_BOOL columns", k);
        if (!i64 && is_out && (a.dtype == DSP_I64 || a.dtype == DSP_U64))
            return fail(DSP_ERR_ARG, "io %d: 64-bit integer outputs are written by integer programs (compute_dtype DSP_I64)", k);
        d.kind = a.kind;
        d.dtype = a.dtype;
        d.len = a.len;
        d.offset = a.offset;
        d.row_stride = a.row_stride;
        d.vec_ok = ((a.row_stride * es) % 16 == 0) && (((int64_t)a.offset * es) % 16 == 0);
    }

    // ---- ops
    std::vector<int> dev_index(n_ops);  // caller's op -> position in the device program
    int n_dev_ops = 0;
    for (int i = 0; i < n_ops; ++i) {
        const dsp_op& o = ops[i];
        for (int s = 0; s < n_slots; ++s)
            if (shares[s] && first_op[s] == i) {
                DevOp& z = P.ops[n_dev_ops++];
                memset(&z, 0, sizeof z);
                z.opcode = DSP_OP_INTERNAL_ZERO;
                z.dst = s;
                z.ic[0] = base[s];
                z.ic[1] = foot[s];
            }
        dev_index[i] = n_dev_ops;
        DevOp& d = P.ops[n_dev_ops++];
        memset(&d, 0, sizeof d);
        d.opcode = o.opcode;
        d.dst = o.dst;
        d.src = o.src;
        d.io = o.io;
        memcpy(d.ip, o.ip, sizeof d.ip);
        memcpy(d.sp, o.sp, sizeof d.sp);
        for (int k = 0; k < 4; ++k) {
            const dsp_scalar_arg& a = o.sp[k];
            if (a.kind == DSP_ARG_INPUT && (a.index < 0 || a.index >= n_io || io[a.index].kind != DSP_IO_SCALAR_IN))
                return fail(DSP_ERR_ARG, "op %d: scalar operand %d is not a scalar input binding", i, k);
            if (a.kind == DSP_ARG_REG && (a.index < 0 || a.index >= n_sregs)) return fail(DSP_ERR_ARG, "op %d: bad scalar register", i);
            if (a.kind < DSP_ARG_CONST || a.kind > DSP_ARG_REG) return fail(DSP_ERR_ARG, "op %d: bad scalar operand kind", i);
        }
        auto need_io = [&](int kind) { return o.io >= 0 && o.io < n_io && io[o.io].kind == kind; };
        // the float32 loop receives float32 scalars, the float64 loop float64 ones
        auto cst = [&](int k) { return f64 ? o.sp[k].value : (double)(float)o.sp[k].value; };
        switch (o.opcode) {
            case DSP_OP_LOAD:
                if (!check_slot(P, o.dst) || !need_io(DSP_IO_WF_IN)) return fail(DSP_ERR_ARG, "op %d: bad LOAD", i);
                if (io[o.io].len != slot_len[o.dst]) return fail(DSP_ERR_ARG, "op %d: LOAD length mismatch", i);
                if (o.ip[0] < 0 || o.ip[1] < 0 || o.ip[0] > io[o.io].offset || (int64_t)io[o.io].offset + io[o.io].len + o.ip[1] > io[o.io].row_stride)
                    return fail(DSP_ERR_ARG, "op %d: LOAD screens samples outside the row (ip[0], ip[1])", i);
                break;
            case DSP_OP_STORE:
                if (!check_slot(P, o.src) || !need_io(DSP_IO_WF_OUT)) return fail(DSP_ERR_ARG, "op %d: bad STORE", i);
                if (io[o.io].len != slot_len[o.src]) return fail(DSP_ERR_ARG, "op %d: STORE length mismatch", i);
                break;
            case DSP_OP_STORE_SCALAR:
                if (!need_io(DSP_IO_SCALAR_OUT) || o.ip[0] < 0 || o.ip[0] >= n_sregs) return fail(DSP_ERR_ARG, "op %d: bad STORE_SCALAR", i);
                break;
            case DSP_OP_BL_SUBTRACT:
            case DSP_OP_MIN_MAX_NORM:
                if (!check_slot(P, o.src) || !check_slot(P, o.dst) || slot_len[o.src] != slot_len[o.dst])
                    return fail(DSP_ERR_ARG, "op %d: bad element-wise op", i);
                break;
            case DSP_OP_POLE_ZERO: {
                if (!check_slot(P, o.src) || !check_slot(P, o.dst) || slot_len[o.src] != slot_len[o.dst])
                    return fail(DSP_ERR_ARG, "op %d: bad POLE_ZERO", i);
                if (o.sp[0].kind != DSP_ARG_CONST) {  // one tau per event: the constant is formed on the device (op_pole_zero)
                    d.ic[1] = 1;
                    break;
                }
                const double tau = cst(0);
                d.ic[0] = std::isnan(tau) ? 1 : 0;
                d.fc[0] = std::exp(-1.0 / tau);  // pole_zero.py:60 -- float64 via libm, like numba's lowering
                break;
            }
            case DSP_OP_DOUBLE_POLE_ZERO: {
                if (!check_slot(P, o.src) || !check_slot(P, o.dst) || slot_len[o.src] != slot_len[o.dst])
                    return fail(DSP_ERR_ARG, "op %d: bad DOUBLE_POLE_ZERO", i);
                bool per_event = false;
                for (int k = 0; k < 3; ++k) per_event |= o.sp[k].kind != DSP_ARG_CONST;
                if (per_event) {  // a time constant or the fraction per event: coefficients and scan matrices are formed on the device
                    if (slot_len[o.src] <= 3) return fail(DSP_E_DPZ_SHORT, "%s", dsp_fatal_message(DSP_E_DPZ_SHORT));
                    d.ic[1] = 1;
                    break;
                }
                const double tau1 = cst(0), tau2 = cst(1), fr = cst(2);
                d.ic[0] = (std::isnan(tau1) || std::isnan(tau2) || std::isnan(fr)) ? 1 : 0;
                if (!d.ic[0] && slot_len[o.src] <= 3) return fail(DSP_E_DPZ_SHORT, "%s", dsp_fatal_message(DSP_E_DPZ_SHORT));
                if (slot_len[o.src] <= 3) d.ic[0] = 1;
                const double a = std::exp(-1.0 / tau1), b = std::exp(-1.0 / tau2);  // pole_zero.py:168-174
                const double den1 = ((fr * b - fr * a) - b) - 1.0;
                const double den2 = -1.0 * ((fr * b - fr * a) - b);
                const double num1 = -1.0 * (a + b);
                const double num2 = a * b;
                d.fc[0] = num1;
                d.fc[1] = num2;
                d.fc[2] = den1;
                d.fc[3] = den2;
                // M^(C*2^d), d = 0..5, M = [[-den1, -den2], [1, 0]]
                long double M[4] = {-(long double)den1, -(long double)den2, 1.0L, 0.0L}, Pw[4] = {1, 0, 0, 1};
                const int C = P.slots[o.src].C;
                long double base[4];
                memcpy(base, M, sizeof base);
                for (int e = C; e; e >>= 1) {  // Pw = M^C
                    if (e & 1) mat2_mul(Pw, base, Pw);
                    mat2_mul(base, base, base);
                }
                for (int dd = 0; dd < 6; ++dd) {
                    for (int k = 0; k < 4; ++k) d.fc[4 + 4 * dd + k] = (double)Pw[k];
                    mat2_mul(Pw, Pw, Pw);
                }
                break;
            }
            case DSP_OP_TRAP_FILTER:
            case DSP_OP_TRAP_NORM:
            case DSP_OP_ASYM_TRAP: {
                if (!check_slot(P, o.src) || !check_slot(P, o.dst) || o.src == o.dst || slot_len[o.src] != slot_len[o.dst])
                    return fail(DSP_ERR_ARG, "op %d: bad trapezoid (source and destination slots must differ)", i);
                int rc = setup_trap(d, o.opcode, o.ip[0], o.ip[1], o.ip[2], slot_len[o.src], P.slots[o.src].C);
                if (rc) return fail(rc, "%s", dsp_fatal_message(rc));
                break;
            }
            case DSP_OP_TRAP_PICKOFF: {
                if (!check_slot(P, o.src) || o.dst < 0 || o.dst >= n_sregs) return fail(DSP_ERR_ARG, "op %d: bad TRAP_PICKOFF", i);
                if (o.ip[3] != DSP_OP_TRAP_FILTER && o.ip[3] != DSP_OP_TRAP_NORM && o.ip[3] != DSP_OP_ASYM_TRAP)
                    return fail(DSP_ERR_ARG, "op %d: TRAP_PICKOFF ip[3] must name a trapezoid opcode", i);
                if (o.io == 's')
                    return fail(DSP_ERR_UNSUPPORTED, "TRAP_PICKOFF cannot take mode 's' (the spline needs the whole filtered waveform): use TRAP_FILTER + PICKOFF");
                int rc = setup_trap(d, o.ip[3], o.ip[0], o.ip[1], o.ip[2], slot_len[o.src], P.slots[o.src].C);
                if (rc) return fail(rc, "%s", dsp_fatal_message(rc));
                break;
            }
            case DSP_OP_TRAP_REDUCE: {
                if (!check_slot(P, o.src) || (o.dst < 0 && o.io < 0) || (o.dst >= 0 && o.dst > n_sregs - 4) || o.io >= n_sregs)
                    return fail(DSP_ERR_ARG, "op %d: bad TRAP_REDUCE", i);
                const int tk = o.ip[3] & 0xff, pk_mode = (o.ip[3] >> 8) & 0xff, pk_reg = ((o.ip[3] >> 16) & 0x3fff) - 1;
                if (tk != DSP_OP_TRAP_FILTER && tk != DSP_OP_TRAP_NORM && tk != DSP_OP_ASYM_TRAP)
                    return fail(DSP_ERR_ARG, "op %d: TRAP_REDUCE ip[3] must name a trapezoid opcode", i);
                if (pk_reg >= n_sregs || (pk_reg >= 0 && (pk_mode == 0 || pk_mode == 's')) || (pk_reg < 0 && pk_mode != 0))
                    return fail(DSP_ERR_ARG, "op %d: bad pick-off in TRAP_REDUCE (register %d, mode %d)", i, pk_reg, pk_mode);
                if (((o.ip[3] >> 30) & 1) && (o.dst < 0 || tk == DSP_OP_ASYM_TRAP))
                    return fail(DSP_ERR_ARG, "op %d: TRAP_REDUCE amax-only needs the min_max registers and trap_filter / trap_norm", i);
                int rc = setup_trap(d, tk, o.ip[0], o.ip[1], o.ip[2], slot_len[o.src], P.slots[o.src].C);
                if (rc) return fail(rc, "%s", dsp_fatal_message(rc));
                break;
            }
            case DSP_OP_PICKOFF:
                if (!check_slot(P, o.src) || o.dst < 0 || o.dst >= n_sregs) return fail(DSP_ERR_ARG, "op %d: bad PICKOFF", i);
                if (o.ip[1] < 0 || o.ip[1] > 2) return fail(DSP_ERR_ARG, "op %d: PICKOFF ip[1] must be 0, 1 or 2", i);
                if (o.ip[1] == 2 && o.sp[1].kind != DSP_ARG_CONST) return fail(DSP_ERR_ARG, "op %d: PICKOFF ip[1] = 2 takes a constant default (sp[1])", i);
                if (o.ip[1] == 1 && (o.sp[0].kind != DSP_ARG_CONST || o.sp[0].value < 0 || o.sp[0].value >= slot_len[o.src] ||
                                     o.sp[0].value != std::floor(o.sp[0].value)))
                    return fail(DSP_ERR_ARG, "op %d: PICKOFF of one sample (ip[1] = 1) needs a constant index inside the waveform", i);
                break;
            case DSP_OP_TIME_POINT_THRESH:
            case DSP_OP_INTERP_TIME_POINT_THRESH:
                if (!check_slot(P, o.src) || o.dst < 0 || o.dst >= n_sregs) return fail(DSP_ERR_ARG, "op %d: bad TIME_POINT_THRESH", i);
                break;
            case DSP_OP_MEAN_BELOW:
                if (!check_slot(P, o.src) || o.dst < 0 || o.dst >= n_sregs) return fail(DSP_ERR_ARG, "op %d: bad MEAN_BELOW", i);
                break;
            case DSP_OP_WINDOWER:
                if (!check_slot(P, o.src) || !check_slot(P, o.dst) || o.src == o.dst) return fail(DSP_ERR_ARG, "op %d: bad WINDOWER", i);
                if (slot_len[o.dst] >= slot_len[o.src]) return fail(DSP_E_WINDOW_LONG, "%s", dsp_fatal_message(DSP_E_WINDOW_LONG));
                break;
            case DSP_OP_AVG_CURRENT: {
                if (!check_slot(P, o.src) || !check_slot(P, o.dst) || o.src == o.dst) return fail(DSP_ERR_ARG, "op %d: bad AVG_CURRENT", i);
                if (o.sp[0].kind != DSP_ARG_CONST) return fail(DSP_ERR_UNSUPPORTED, "avg_current: the window length must be a constant");
                const double length = f64 ? o.sp[0].value : (double)(float)o.sp[0].value;
                const int n = slot_len[o.src];
                if (!(length >= 0) || !(length < (double)n)) return fail(DSP_E_AVGCUR_RANGE, "%s", dsp_fatal_message(DSP_E_AVGCUR_RANGE));
                const int L = (int)length;
                if (L <= 0 || slot_len[o.dst] != n - L)
                    return fail(DSP_ERR_ARG, "avg_current: the output must hold len(w_in) - int(length) = %d samples (it holds %d)", n - L,
                                slot_len[o.dst]);
                d.ic[0] = L;
                d.fc[0] = length;
                break;
            }
            case DSP_OP_UPSAMPLER: {
                if (!check_slot(P, o.src) || !check_slot(P, o.dst) || o.src == o.dst) return fail(DSP_ERR_ARG, "op %d: bad UPSAMPLER", i);
                if (o.sp[0].kind != DSP_ARG_CONST) return fail(DSP_ERR_UNSUPPORTED, "upsampler: the factor must be a constant");
                const double up = f64 ? o.sp[0].value : (double)(float)o.sp[0].value;
                if (!(up > 0)) return fail(DSP_E_UPSAMPLE, "%s", dsp_fatal_message(DSP_E_UPSAMPLE));
                d.fc[0] = up;
                d.fc[1] = floor(up / 2.0);
                d.ic[0] = (int)up;
                break;
            }
            case DSP_OP_MOVING_WINDOW_MULTI: {
                const bool in_place = o.ip[3] == 1;
                if (!check_slot(P, o.src) || !check_slot(P, o.dst) || (o.src == o.dst) != in_place || slot_len[o.src] != slot_len[o.dst])
                    return fail(DSP_ERR_ARG, "op %d: bad MOVING_WINDOW_MULTI", i);
                if (o.sp[0].kind != DSP_ARG_CONST) return fail(DSP_ERR_UNSUPPORTED, "moving_window_multi: the window length must be a constant");
                const double length = f64 ? o.sp[0].value : (double)(float)o.sp[0].value;
                const int num = o.ip[1], n = slot_len[o.src];
                if (floor(length) != length) return fail(DSP_E_MW_LEN_INT, "%s", dsp_fatal_message(DSP_E_MW_LEN_INT));
                if ((long long)length < 0 || (long long)length >= n) return fail(DSP_E_MW_LEN_RANGE, "%s", dsp_fatal_message(DSP_E_MW_LEN_RANGE));
                if (num < 0) return fail(DSP_E_MW_NUM_NEG, "%s", dsp_fatal_message(DSP_E_MW_NUM_NEG));
                if (num > 0 && (long long)length == 0) return fail(DSP_E_ZERODIV, "%s", dsp_fatal_message(DSP_E_ZERODIV));
                // (the scratch may be the source itself when the number of windows is odd: the first pass goes source -> target, so the
                // source is free from the second pass on -- and is overwritten)
                if (in_place) {  // ip[2]: a side slot for the ends of the chunks, 64 lanes x (L | 1) elements; the window inside one chunk
                    const long long Lw = (long long)length;
                    if (!check_slot(P, o.ip[2]) || o.ip[2] == o.src || Lw > P.slots[o.src].C || 64 * (Lw | 1) > 64LL * P.slots[o.ip[2]].pitch)
                        return fail(DSP_ERR_ARG, "op %d: MOVING_WINDOW_MULTI in place needs a window of at most %d samples and a side slot of 64 x window samples (ip[2])",
                                    i, P.slots[o.src].C);
                } else
                if (num > 1 && (!check_slot(P, o.ip[2]) || (o.ip[2] == o.src && num % 2 == 0) || o.ip[2] == o.dst || slot_len[o.ip[2]] != n))
                    return fail(DSP_ERR_ARG, "op %d: MOVING_WINDOW_MULTI with several windows needs a scratch slot of the same length (ip[2])", i);
                d.ic[0] = (int)length;
                d.ic[1] = num;
                d.ic[2] = o.ip[0];
                d.fc[0] = length;
                break;
            }
            case DSP_OP_TRAP_WINDOW_PICKOFF: {
                if (!check_slot(P, o.src) || o.dst < 0 || o.dst >= n_sregs) return fail(DSP_ERR_ARG, "op %d: bad TRAP_WINDOW_PICKOFF", i);
                const int rise = o.ip[0], flat = o.ip[1];
                if (rise < 0) return fail(DSP_E_TRAP_RISE, "%s", dsp_fatal_message(DSP_E_TRAP_RISE));
                if (flat < 0) return fail(DSP_E_TRAP_FLAT, "%s", dsp_fatal_message(DSP_E_TRAP_FLAT));
                if (2 * (long long)rise + flat > slot_len[o.src]) return fail(DSP_E_TRAP_WIDE, "%s", dsp_fatal_message(DSP_E_TRAP_WIDE));
                break;
            }
            case DSP_OP_MIN_MAX:
                if (!check_slot(P, o.src) || o.dst < 0 || o.dst > n_sregs - 4) return fail(DSP_ERR_ARG, "op %d: bad MIN_MAX", i);
                break;
            case DSP_OP_LINEAR_SLOPE_FIT: {
                if (!check_slot(P, o.src) || o.dst < 0 || o.dst > n_sregs - 4) return fail(DSP_ERR_ARG, "op %d: bad LINEAR_SLOPE_FIT", i);
                const int first = o.ip[0];
                if (first < 0 || first > slot_len[o.src]) return fail(DSP_ERR_ARG, "op %d: LINEAR_SLOPE_FIT slice out of range", i);
                const int count = o.ip[1] > 0 ? o.ip[1] : slot_len[o.src] - first;  // ip[1] == 0: to the end of the slot
                if (count < 0 || count > slot_len[o.src] - first) return fail(DSP_ERR_ARG, "op %d: LINEAR_SLOPE_FIT slice out of range", i);
                if (count < 2) return fail(DSP_E_ZERODIV, "%s", dsp_fatal_message(DSP_E_ZERODIV));  // the line fit's denominator
                d.ic[0] = first;
                d.ic[1] = count;
                break;
            }
            case DSP_OP_AMAX:
                if (!check_slot(P, o.src) || o.dst < 0 || o.dst >= n_sregs) return fail(DSP_ERR_ARG, "op %d: bad AMAX", i);
                break;
            case DSP_OP_DWT_HAAR: {
                if (!check_slot(P, o.src) || !check_slot(P, o.dst) || !check_slot(P, o.ip[2]) || o.dst == o.src || o.dst == o.ip[2])
                    return fail(DSP_ERR_ARG, "op %d: bad DWT_HAAR slots", i);
                if (o.ip[0] <= 0) return fail(DSP_E_DWT_LEVEL, "%s", dsp_fatal_message(DSP_E_DWT_LEVEL));
                if (o.ip[1] != 'a' && o.ip[1] != 'd') return fail(DSP_ERR_ARG, "op %d: DWT coefficient must be 'a' or 'd'", i);
                int len = slot_len[o.src];
                if (slot_len[o.ip[2]] < (len + 1) / 2 && o.ip[0] > 1) return fail(DSP_ERR_ARG, "op %d: DWT scratch slot too small", i);
                for (int l = 0; l < o.ip[0]; ++l) len = (len + 1) / 2;
                if (len != slot_len[o.dst]) return fail(DSP_E_DWT_OUTLEN, "%s (got %d, expect %d)", dsp_fatal_message(DSP_E_DWT_OUTLEN), slot_len[o.dst], len);
                break;
            }
            case DSP_OP_COPY: {  // dst[k] = src[ip[0] + k * step], step = ip[1] (0 means 1; negative: backwards)
                const int64_t step = o.ip[1] != 0 ? o.ip[1] : 1;
                const int64_t last = check_slot(P, o.dst) ? (int64_t)o.ip[0] + (int64_t)(slot_len[o.dst] - 1) * step : -1;
                if (!check_slot(P, o.src) || !check_slot(P, o.dst) || o.src == o.dst || o.ip[0] < 0 || o.ip[0] >= slot_len[o.src] || last < 0 ||
                    last >= slot_len[o.src])
                    return fail(DSP_ERR_ARG, "op %d: bad COPY", i);
                break;
            }
            case DSP_OP_ELEMENTWISE: {
                if (!check_slot(P, o.dst) || !fn_code_ok(o.ip[0], f64)) return fail(DSP_ERR_ARG, "op %d: bad ELEMENTWISE", i);
                const int opnd[3] = {o.src, o.ip[1], o.ip[2]};
                int n_wf_opnd = 0;
                for (int k = 0; k < 3; ++k) {
                    if (opnd[k] < 0) continue;
                    if (!check_slot(P, opnd[k]) || slot_len[opnd[k]] != slot_len[o.dst])
                        return fail(DSP_ERR_ARG, "op %d: ELEMENTWISE operand %d is not a waveform of the length of the result", i, k);
                    ++n_wf_opnd;
                }
                if (!n_wf_opnd) return fail(DSP_ERR_ARG, "op %d: ELEMENTWISE needs a waveform operand (DSP_OP_SCALAR_FUNC otherwise)", i);
                break;
            }
            case DSP_OP_SCALAR_FUNC:
                if (o.dst < 0 || o.dst >= n_sregs || !fn_code_ok(o.ip[0], f64, i64)) return fail(DSP_ERR_ARG, "op %d: bad SCALAR_FUNC", i);
                break;
            case DSP_OP_CONVOLVE:
            case DSP_OP_CONVOLVE_AMAX: {
                const bool fusedmax = o.opcode == DSP_OP_CONVOLVE_AMAX;
                if (!check_slot(P, o.src) || !need_io(DSP_IO_TAPS)) return fail(DSP_ERR_ARG, "op %d: bad CONVOLVE", i);
                if (fusedmax ? (o.dst < 0 || o.dst >= n_sregs || o.ip[2] <= 0) : (!check_slot(P, o.dst) || o.src == o.dst))
                    return fail(DSP_ERR_ARG, "op %d: bad CONVOLVE destination", i);
                // ip[3] > 0: the kernel has ip[3] taps and the binding holds zeros after them (up to a multiple of the tap block, so the
                // blocked path covers every tap; the op falls back to the true length for a waveform with an infinity in it: 0 * inf)
                if (o.ip[3] < 0 || o.ip[3] > io[o.io].len) return fail(DSP_ERR_ARG, "op %d: CONVOLVE kernel length beyond its binding", i);
                const int m = o.ip[3] > 0 ? o.ip[3] : io[o.io].len;
                const int n = slot_len[o.src], p = fusedmax ? o.ip[2] : slot_len[o.dst], mode = o.ip[0];
                d.ic[1] = m;
                d.ic[6] = io[o.io].len;
                d.ic[2] = (o.ip[1] & 1) ? 1 : 0;  // caller found a NaN among the taps -> output NaN (convolutions.py:45-46)
                d.ic[5] = (o.ip[1] & 2) ? 1 : 0;  // ... an infinity: 0 * inf is NaN, so windows must not reach into the zero margins
                d.ic[3] = p;
                if (m > n) return fail(DSP_E_CONV_LONG, "%s", dsp_fatal_message(DSP_E_CONV_LONG));
                if (mode == 'f') {
                    if (p != n + m - 1) return fail(DSP_E_CONV_OUTLEN, "Output waveform has length %d; expect %d", p, n + m - 1);
                    d.ic[0] = 0;
                } else if (mode == 'v') {
                    if (p != n - m + 1) return fail(DSP_E_CONV_OUTLEN, "Output waveform has length %d; expect %d", p, n - m + 1);
                    d.ic[0] = m - 1;
                } else if (mode == 's') {
                    if (p != n) return fail(DSP_E_CONV_OUTLEN, "Output waveform has length %d; expect %d", p, n);
                    d.ic[0] = (m - 1) / 2;
                } else {
                    return fail(DSP_E_CONV_MODE, "%s", dsp_fatal_message(DSP_E_CONV_MODE));
                }
                break;
            }
            case DSP_OP_SCALAR_AFFINE:
            case DSP_OP_SCALAR_DIV:
                if (o.dst < 0 || o.dst >= n_sregs) return fail(DSP_ERR_ARG, "op %d: bad scalar arithmetic op", i);
                break;
            case DSP_OP_SCALAR_CONVERT:
                if (o.dst < 0 || o.dst >= n_sregs || o.ip[0] < 0 || o.ip[0] > 4 || o.sp[3].kind != DSP_ARG_CONST)
                    return fail(DSP_ERR_ARG, "op %d: bad SCALAR_CONVERT (ip[0] = rounding 0..4, sp[3] = constant period ratio)", i);
                break;
            default: return fail(DSP_ERR_ARG, "op %d: unknown opcode %d", i, o.opcode);
        }
    }
    // LOAD s; BL_SUBTRACT s <- s (what nearly every recipe starts with): the load subtracts while it writes the samples to LDS and the
    // second op becomes a no-op -- one pass over the waveform in LDS less.  Same results: one float subtraction per sample in the loop's
    // type, and the NaN rule of bl_subtract.py:41-44 (a NaN anywhere, or a NaN baseline: a NaN waveform) is the load's own flag.
    for (int i = 0; i + 1 < n_ops; ++i) {
        const dsp_op &ld = ops[i], &bs = ops[i + 1];
        if (ld.opcode == DSP_OP_LOAD && bs.opcode == DSP_OP_BL_SUBTRACT && bs.src == ld.dst && bs.dst == ld.dst && bs.ip[0] == 0 &&
            dev_index[i + 1] == dev_index[i] + 1) {
            DevOp& L = P.ops[dev_index[i]];
            DevOp& B = P.ops[dev_index[i + 1]];
            L.ic[0] = 1;
            L.sp[0] = B.sp[0];
            B.opcode = DSP_OP_INTERNAL_NOP;
        }
    }
    P.n_ops = n_dev_ops;

    // ---- does the program have the shape LOAD [-> BL_SUBTRACT] -> POLE_ZERO -> TRAP_PICKOFF -> STORE_SCALAR on one slot?
    {
        int i = 0;
        const dsp_op* ld = (n_ops > i && ops[i].opcode == DSP_OP_LOAD) ? &ops[i++] : nullptr;
        const dsp_op* bs = (ld && n_ops > i && ops[i].opcode == DSP_OP_BL_SUBTRACT) ? &ops[i++] : nullptr;
        const dsp_op* pz = (ld && n_ops > i && ops[i].opcode == DSP_OP_POLE_ZERO) ? &ops[i++] : nullptr;
        const dsp_op* tp = (pz && n_ops > i && ops[i].opcode == DSP_OP_TRAP_PICKOFF) ? &ops[i++] : nullptr;
        const dsp_op* st = (tp && n_ops > i && ops[i].opcode == DSP_OP_STORE_SCALAR) ? &ops[i++] : nullptr;
        const int wdt = ld ? io[ld->io].dtype : -1;
        // (a time constant per event, a float32 column: the register-resident kernel's TAU builds form exp(-1/tau) per row like the interpreter's op)
        const bool tau_col = pz && pz->sp[0].kind == DSP_ARG_INPUT && io[pz->sp[0].index].dtype == DSP_F32;
        const bool shape = !f64 && st && i == n_ops && n_slots == 1 && ld->ip[0] == 0 && ld->ip[1] == 0 && (wdt == DSP_F32 || wdt == DSP_I16 || wdt == DSP_U16) && P.io[ld->io].vec_ok &&
                           (!bs || (bs->dst == 0 && bs->src == 0 && bs->sp[0].kind != DSP_ARG_REG)) && pz->dst == 0 && pz->src == 0 &&
                           (pz->sp[0].kind == DSP_ARG_CONST || tau_col) &&
                           tp->src == 0 && tp->sp[0].kind != DSP_ARG_REG && st->ip[0] == tp->dst &&
                           (slot_len[0] == 1024 || slot_len[0] == 2048 || slot_len[0] == 4096 || slot_len[0] == 8192);
        if (!shape && st && i == n_ops && n_slots == 1) {  // the ops of the energy chain, but not the kernels' case of it
            if (f64) note(ch, "the energy chain in a float64 loop: the fused energy kernels are float32");
            else if (pz->sp[0].kind != DSP_ARG_CONST && !tau_col)
                note(ch, "a pole-zero time constant per event that is not a float32 column: the fused energy kernels take a constant or such a column");
            else if (!(slot_len[0] == 1024 || slot_len[0] == 2048 || slot_len[0] == 4096 || slot_len[0] == 8192))
                note(ch, "waveforms of %d samples: the fused energy kernels take 1024, 2048, 4096 or 8192", slot_len[0]);
            else if (!P.io[ld->io].vec_ok) note(ch, "rows that do not start on 16-byte boundaries: the fused energy kernels read 16 bytes per lane");
        }
        auto f32_col = [&](const dsp_scalar_arg& a) { return a.kind != DSP_ARG_INPUT || io[a.index].dtype == DSP_F32; };
        if (shape && (!bs || f32_col(bs->sp[0])) && f32_col(tp->sp[0]) && io[st->io].dtype == DSP_F32) {  // (the kernels store a float)
            EnergyArgs& F = ch->fused;
            const DevOp& dpz = P.ops[dev_index[pz - ops]];
            const DevOp& dtp = P.ops[dev_index[tp - ops]];
            F.wf_stride = io[ld->io].row_stride;
            F.wf_offset = io[ld->io].offset;
            F.len = slot_len[0];
            F.has_bl = bs ? 1 : 0;
            if (bs) {
                if (bs->sp[0].kind == DSP_ARG_INPUT) {
                    ch->io_bl = bs->sp[0].index;
                    F.bl_stride = io[ch->io_bl].row_stride;
                } else {
                    F.bl_const = (float)bs->sp[0].value;
                }
            }
            if (tp->sp[0].kind == DSP_ARG_INPUT) {
                ch->io_tp = tp->sp[0].index;
                F.tp_stride = io[ch->io_tp].row_stride;
            } else {
                F.tp_const = (float)tp->sp[0].value;
            }
            F.mode = tp->io;
            F.out_stride = io[st->io].row_stride;
            F.c = dpz.fc[0];
            F.tau_nan = dpz.ic[0];
            F.rr = dtp.fc[0];
            F.ll = dtp.fc[1];
            F.all_nan = dtp.ic[9];
            F.C = P.slots[0].C;
            F.pitch = P.slots[0].pitch;
            F.invC = P.slots[0].invC;
            for (int k = 0; k < 3; ++k) {
                F.q[k] = dtp.ic[3 + k];
                F.rho[k] = dtp.ic[6 + k];
            }
            F.lds_elems_per_wave = P.lds_elems_per_wave;
            F.slot_off = P.slots[0].off;
#ifdef DSPEED_HIP_DIAG
            if (const char* ab = getenv("DSPEED_HIP_ABLATE")) F.ablate = atoi(ab);  // diagnostic library only: skip passes / stamp phases
#endif
            ch->io_wf = ld->io;
            ch->io_out = st->io;
            ch->fused_trap = tp->ip[3];
            ch->fused_npf = slot_len[0] / 256;          // one wavefront per waveform: len == 256 * npf, npf in {4, 8, 16}
            ch->fused_ok = slot_len[0] <= 4096 && wdt == DSP_F32 && !tau_col;  // the classic kernel reads float32 rows only, one time constant
            if (tau_col) {
                ch->io_tau = pz->sp[0].index;
                F.tau_stride = io[ch->io_tau].row_stride;
            }
            ch->wf_dtype = wdt;
            const char* env = getenv("DSPEED_HIP_NO_FUSED");
            ch->fused_on = !(env && env[0] == '1');
            {  // register-resident kernel: C = len/64 + 1 samples per lane, linear LDS image of the waveform (1024 .. 8192 samples)
                EnergyArgs& I = ch->rr;
                I = F;
                const int Ci = slot_len[0] / 64 + 1;
                I.C = Ci;
                I.pitch = Ci;
                I.invC = 1.0f / (float)Ci;
                int guard = 2 * Ci + 8;  // zeros below the image: lagged reads before sample 0
                guard = ((guard + 3) / 4) * 4;
                I.slot_off = guard;
                const int ng1 = (Ci - 1) / 8 + 1, auxp = ng1 <= 9 ? 9 : (ng1 | 1);
                int elems = guard + 64 * Ci + 16 + 64 * auxp + 32;  // image, tail, per-lane side array (auxp per lane), capture buffer (2 x 16)
                elems = ((elems + 3) / 4) * 4;
                I.lds_elems_per_wave = elems;
                ch->rr_lds_bytes = elems * 4;
                for (int k = 0; k < 3; ++k) {
                    I.q[k] = dtp.ic[k];  // the lags themselves
                    I.rho[k] = 0;
                }
                for (int S = 1; S <= 2; ++S) {
                    const int CS = (Ci - 1) / S;
                    for (int k = 0; k < 3; ++k)
                        for (int sidx = 0; sidx < S; ++sidx) {
                            const int pos = sidx * CS - dtp.ic[k];          // samples before the sub-chain start, relative to the chunk
                            const int r = ((pos % Ci) + Ci) % Ci;           // ... = r samples into the chunk of the lane `shift` below
                            const int shift = (r - pos) / Ci;
                            int cs = r / CS;
                            if (cs > S - 1) cs = S - 1;
                            ch->plan[S - 1].shift[k][sidx] = shift;
                            ch->plan[S - 1].cs[k][sidx] = cs;
                            ch->plan[S - 1].local[k][sidx] = r - cs * CS;
                        }
                }
                ch->rr_ok = true;
                ch->variant = 6;
                if (const char* venv = getenv("DSPEED_HIP_VARIANT")) ch->variant = atoi(venv);  // A/B runs: 1, 6, 8
                if (ch->variant != 1 && ch->variant != 8) ch->variant = 6;
            }
        }
    }

    ch->fir_ok = match_fir_shape(ch, ops, n_ops, io, slot_len, n_slots, f64);
    if (!ch->fir_ok) {
        for (int k = 0; k < DSP_FIR_MAXK; ++k) ch->fio_taps[k] = ch->fio_out[k] = -1;
        ch->fir_ok = match_fir_store_shape(ch, ops, n_ops, io, slot_len, n_slots, f64);
    }
    if (ch->fir_ok) {
        const char* env = getenv("DSPEED_HIP_NO_FUSED");
        ch->fused_on = !(env && env[0] == '1');
        // the amax form runs on the float16 matrix instructions (two-way split operands, float32-accurate: dsp_fir_f16.hip) unless asked
        // for the float32 ones (DSPEED_HIP_FIR_F32=1: A/B and the exact float32 chain); its staging loads are 16 bytes of any row type
        const char* f32 = getenv("DSPEED_HIP_FIR_F32");
        const dsp_io_desc& w = io[ch->fio_wf];
        const int es = w.dtype == DSP_F32 ? 4 : 2;
        const int n_slice = ch->fir.n;  // (its 8-sample vectors are read whole: the last one must end inside the row)
        if (!(f32 && f32[0] == '1') && (w.row_stride * es) % 16 == 0 && (w.offset * es) % 16 == 0 &&
            ((n_slice & 7) == 0 || w.offset + ((n_slice + 7) & ~7) <= w.row_stride)) {
            ch->fir_f16 = true;
            ch->f16.tz = dsp_internal_fir_f16_tz(ch->fir.kend);
        }
    }
    ch->rows_ok = match_rows_shape(ch, ops, n_ops, io, n_io, slot_len, n_slots, dev_index, f64);
    if (!ch->rows_ok) {
        ch->rio_wf = ch->rio_bl = ch->rio_thr = ch->rio_ts = ch->rio_tpt = ch->rio_dwt = -1;
        for (int k = 0; k < 4; ++k) ch->rio_mm[k] = -1;
    } else {
        const char* env = getenv("DSPEED_HIP_NO_FUSED");
        ch->fused_on = !(env && env[0] == '1');
    }

    {  // nothing but arithmetic between per-event values and stores: one row per lane instead of one per wavefront
        bool only_scalar = true;
        for (int i = 0; i < n_ops; ++i) {
            const int oc = ops[i].opcode;
            only_scalar &= oc == DSP_OP_SCALAR_AFFINE || oc == DSP_OP_SCALAR_DIV || oc == DSP_OP_SCALAR_CONVERT || oc == DSP_OP_SCALAR_FUNC ||
                           oc == DSP_OP_STORE_SCALAR;
        }
        ch->scalar_ok = only_scalar && n_dev_ops == n_ops;
        if (ch->scalar_ok) {
            const char* env = getenv("DSPEED_HIP_NO_FUSED");
            ch->fused_on = !(env && env[0] == '1');
        }
    }
    ch->pz_ok = match_pz_rows_shape(ch, ops, n_ops, io, slot_len, n_slots, dev_index, f64);
    if (ch->pz_ok) {
        const char* env = getenv("DSPEED_HIP_NO_FUSED");
        ch->fused_on = !(env && env[0] == '1');
    }
    ch->red_ok = match_reduce_shape(ch, ops, n_ops, io, slot_len, f64);
    if (ch->red_ok) {
        const char* env = getenv("DSPEED_HIP_NO_FUSED");
        ch->fused_on = !(env && env[0] == '1');
    }
    {
        const char* off = getenv("DSPEED_HIP_NO_FIR_RUNS");  // A/B runs: the matrix-core FIR for piecewise-constant kernels too
        if (!ch->red_ok && !(off && off[0] == '1')) ch->runs_ok = match_fir_runs_shape(ch, ops, n_ops, io, slot_len, n_slots, f64);
        if (ch->runs_ok) {
            const char* env = getenv("DSPEED_HIP_NO_FUSED");
            ch->fused_on = !(env && env[0] == '1');
            ch->fir_ok = ch->fir_f16 = false;  // (LOAD, CONVOLVE, STORE has the matrix-core kernel's shape as well)
        }
    }
    ch->cur_ok = match_current_shape(ch, ops, n_ops, io, slot_len, f64);
    if (!ch->cur_ok) {
        ch->cio_wf = ch->cio_t0 = -1;
        for (int k = 0; k < 4; ++k) ch->cio_out[k] = -1;
    } else {
        const char* env = getenv("DSPEED_HIP_NO_FUSED");
        ch->fused_on = !(env && env[0] == '1');
    }

    // ---- a team of two wavefronts per row?  A program that loads ONE waveform and then only reads it -- the trapezoid reductions, pick-offs and
    // walks a whole recipe runs on its pole-zero rows -- whose image leaves LDS for one wavefront per SIMD: its ops fall into groups that share
    // no scalar register, and two wavefronts can run two groups on the one image at the same time.
    // ---- a threshold that is a fraction of a per-event value (the rise-time walks of the Ge recipes: time_point_thresh at 0.99 / 0.9 / 0.5 /
    // 0.1 of the trapezoid's maximum) costs the interpreter an op of its own -- ~ 1 700 cycles of a row's lone wavefront, whatever the op
    // computes -- for one multiplication.  SCALAR_AFFINE d <- x * const + 0 whose result only TIME_POINT_THRESH ops read as their threshold
    // is folded into them: they multiply (the same float multiplication: x * b + (+-0) is x * b) and the op becomes a no-op.
    if (const char* nf = getenv("DSPEED_HIP_NO_THRESHOLD_FOLD"); !(nf && nf[0] == '1')) {  // (A/B runs and the bit-identity test)
        auto writes_reg = [&](const DevOp& o, int r) {
            switch (o.opcode) {
                case DSP_OP_MIN_MAX: return r >= o.dst && r < o.dst + 4;
                case DSP_OP_TRAP_REDUCE:
                    return (o.dst >= 0 && r >= o.dst && r < o.dst + 4) || (o.io >= 0 && r == o.io) || r == ((o.ip[3] >> 16) & 0x3fff) - 1;
                case DSP_OP_LOAD: case DSP_OP_STORE: case DSP_OP_STORE_SCALAR: case DSP_OP_INTERNAL_NOP: case DSP_OP_INTERNAL_ZERO: return false;
                default: return o.dst == r;  // (waveform ops name a slot there: a spurious match only ends a window early)
            }
        };
        for (int i = 0; i < P.n_ops; ++i) {
            DevOp& a = P.ops[i];
            if (a.opcode != DSP_OP_SCALAR_AFFINE || a.sp[0].kind != DSP_ARG_REG || a.sp[1].kind != DSP_ARG_CONST || a.sp[2].kind != DSP_ARG_CONST ||
                a.sp[2].value != 0.0 || a.sp[0].index == a.dst)
                continue;
            const int r = a.dst, x = a.sp[0].index;
            std::vector<int> users;
            bool fine = true;
            for (int j = i + 1; j < P.n_ops && fine; ++j) {
                const DevOp& o = P.ops[j];
                bool reads_r = false;
                for (int k = 0; k < 4; ++k)
                    if (o.sp[k].kind == DSP_ARG_REG && o.sp[k].index == r) {
                        reads_r = true;
                        if (!(o.opcode == DSP_OP_TIME_POINT_THRESH && k == 0 && o.ic[0] == 0)) fine = false;
                    }
                if (o.opcode == DSP_OP_STORE_SCALAR && o.ip[0] == r) fine = false;
                if (reads_r && fine) users.push_back(j);
                if (writes_reg(o, r)) break;                        // the register starts another life: what follows reads that
                if (writes_reg(o, x) && fine) {                     // the value the walks would multiply changes here: nothing may read r after it
                    for (int j2 = j + 1; j2 < P.n_ops; ++j2) {
                        const DevOp& o2 = P.ops[j2];
                        for (int k = 0; k < 4; ++k)
                            if (o2.sp[k].kind == DSP_ARG_REG && o2.sp[k].index == r) fine = false;
                        if (o2.opcode == DSP_OP_STORE_SCALAR && o2.ip[0] == r) fine = false;
                        if (writes_reg(o2, r)) break;
                    }
                    break;
                }
            }
            if (!fine || users.empty()) continue;
            for (int j : users) {
                P.ops[j].sp[0] = a.sp[0];
                P.ops[j].ic[0] = 1;
                P.ops[j].fc[0] = a.sp[1].value;
            }
            a.opcode = DSP_OP_INTERNAL_NOP;
        }
    }

    P.team = 1;
    for (int i = 0; i < P.n_ops; ++i) P.ops[i].member = DSP_MEMBER_ALL;
    {
        const char* env = getenv("DSPEED_HIP_NO_TEAMS");
        const bool vm_runs = !(ch->fused_ok || ch->rr_ok || ch->rows_ok || ch->fir_ok || ch->cur_ok || ch->red_ok || ch->pz_ok || ch->scalar_ok);
        bool ok = vm_runs && !(env && env[0] == '1') && !f64 && !ch->has_fir && n_slots == 1 && ch->lds_bytes_per_wave > 0 &&
                  LDS_BYTES_PER_CU / ch->lds_bytes_per_wave <= 4 && ch->waves_per_block * 2 <= 16 && n_sregs <= 128;
        int first = 0;
        while (first < P.n_ops && P.ops[first].opcode == DSP_OP_INTERNAL_NOP) ++first;
        ok = ok && first < P.n_ops && P.ops[first].opcode == DSP_OP_LOAD;
        // registers an op writes / reads (scalar registers only: after the load nothing writes the waveform)
        auto writes = [&](const DevOp& o, int* r) {
            int n = 0;
            switch (o.opcode) {
                case DSP_OP_MIN_MAX: for (int k = 0; k < 4; ++k) r[n++] = o.dst + k; break;
                case DSP_OP_TRAP_REDUCE:
                    if (o.dst >= 0) for (int k = 0; k < 4; ++k) r[n++] = o.dst + k;
                    if (o.io >= 0) r[n++] = o.io;
                    if ((((o.ip[3] >> 16) & 0x3fff) - 1) >= 0) r[n++] = ((o.ip[3] >> 16) & 0x3fff) - 1;  // (a pick-off that reads the same trapezoid)
                    break;
                case DSP_OP_AMAX: case DSP_OP_TRAP_PICKOFF: case DSP_OP_TIME_POINT_THRESH: case DSP_OP_PICKOFF: case DSP_OP_SCALAR_AFFINE:
                case DSP_OP_SCALAR_DIV: case DSP_OP_SCALAR_CONVERT: case DSP_OP_SCALAR_FUNC: r[n++] = o.dst; break;
                default: break;
            }
            return n;
        };
        auto reads = [&](const DevOp& o, int* r) {
            int n = 0;
            for (int k = 0; k < 4; ++k)
                if (o.sp[k].kind == DSP_ARG_REG) r[n++] = o.sp[k].index;
            if (o.opcode == DSP_OP_STORE_SCALAR) r[n++] = o.ip[0];
            return n;
        };
        for (int i = first + 1; ok && i < P.n_ops; ++i) {
            switch (P.ops[i].opcode) {
                case DSP_OP_TRAP_REDUCE: case DSP_OP_TRAP_PICKOFF: case DSP_OP_TIME_POINT_THRESH: case DSP_OP_PICKOFF: case DSP_OP_MIN_MAX: case DSP_OP_AMAX:
                case DSP_OP_SCALAR_AFFINE: case DSP_OP_SCALAR_DIV: case DSP_OP_SCALAR_CONVERT: case DSP_OP_SCALAR_FUNC: case DSP_OP_STORE_SCALAR:
                case DSP_OP_INTERNAL_NOP: break;
                default: ok = false;
            }
        }
        if (ok) {
            // groups: ops joined by any register one writes and the other reads or writes
            std::vector<int> parent(P.n_ops);
            for (int i = 0; i < P.n_ops; ++i) parent[i] = i;
            auto find = [&](int x) { while (parent[x] != x) x = parent[x] = parent[parent[x]]; return x; };
            std::vector<int> owner(DSP_MAX_SREGS + 8, -1);  // register -> an op that touched it
            for (int i = first + 1; i < P.n_ops; ++i) {
                int regs[16], n = writes(P.ops[i], regs);
                n += reads(P.ops[i], regs + n);
                for (int k = 0; k < n; ++k) {
                    const int r = regs[k];
                    if (r < 0 || r >= (int)owner.size()) { ok = false; break; }
                    if (owner[r] < 0) owner[r] = i;
                    else parent[find(i)] = find(owner[r]);
                }
            }
            auto weight = [&](int oc) {
                switch (oc) {
                    case DSP_OP_TRAP_REDUCE: return 27;
                    case DSP_OP_TRAP_PICKOFF: return 20;
                    case DSP_OP_MIN_MAX: return 16;
                    case DSP_OP_TIME_POINT_THRESH: case DSP_OP_AMAX: return 4;
                    case DSP_OP_PICKOFF: return 3;
                    case DSP_OP_STORE_SCALAR: return 1;
                    default: return 2;
                }
            };
            std::vector<int> w(P.n_ops, 0);
            for (int i = first + 1; i < P.n_ops; ++i) w[find(i)] += weight(P.ops[i].opcode);
            std::vector<int> roots;
            for (int i = first + 1; i < P.n_ops; ++i)
                if (find(i) == i) roots.push_back(i);
            std::sort(roots.begin(), roots.end(), [&](int a, int b) { return w[a] > w[b]; });
            // groups dealt out to two or three members, heaviest first, each to the member with the least so far (three: when the third
            // member still gets a real share -- the Ge recipe's program is a trapezoid with its five walks, a trapezoid with a pick-off and
            // a pick-off of a third: the recurrences are latency-bound, a third wavefront on the image fills a SIMD's idle issue slots)
            int load[3] = {0, 0, 0}, n_team = 2;
            std::vector<int> side(P.n_ops, 0);
            auto deal = [&](int members) {
                load[0] = load[1] = load[2] = 0;
                for (int r : roots) {
                    int m = 0;
                    for (int k = 1; k < members; ++k)
                        if (load[k] < load[m]) m = k;
                    side[r] = m;
                    load[m] += w[r];
                }
            };
            const char* t3 = getenv("DSPEED_HIP_TEAM_MAX");  // (A/B runs: 2 keeps teams of two)
            deal(3);
            {
                const int total = load[0] + load[1] + load[2];
                int least = load[0] < load[1] ? load[0] : load[1];
                least = load[2] < least ? load[2] : least;
                if ((int)roots.size() >= 3 && 6 * least >= total && !(t3 && atoi(t3) < 3) && ch->waves_per_block * 3 <= 16)
                    n_team = 3;
                else
                    deal(2);
            }
            // worth more wavefronts only when each takes a real share of the work
            if (ok && load[0] > 0 && load[1] > 0 && (n_team == 3 || 5 * (load[0] < load[1] ? load[0] : load[1]) >= load[0] + load[1])) {
                P.team = n_team;
                for (int i = first + 1; i < P.n_ops; ++i) P.ops[i].member = side[find(i)];
                // a workgroup per team: the barrier at the end of a row then holds the two members of one row, not four rows' worth of
                // wavefronts whose walks take different times (DSPEED_HIP_TEAM_WPB: the teams per workgroup, for the A/B)
                int twpb = 1;
                if (const char* tw = getenv("DSPEED_HIP_TEAM_WPB")) twpb = atoi(tw);
                if (twpb >= 1 && twpb < ch->waves_per_block) ch->waves_per_block = P.waves_per_block = twpb;
            }
        }
    }

    // The interpreter pays a dispatch per op and row (a lone wavefront: op fetch, decode, a few hundred cycles), and a recipe ends in
    // dozens of one-lane stores.  Last step, after every shape matcher has read the ops: a run of STORE_SCALARs becomes one op whose lanes
    // store one value each, and the no-ops a folded BL_SUBTRACT left are dropped.  (The row-per-lane kernel keeps its plain stores.)
    if (!ch->scalar_ok) {
        int w = 0;
        for (int r = 0; r < P.n_ops;) {
            if (P.ops[r].opcode == DSP_OP_INTERNAL_NOP) {
                ++r;
                continue;
            }
            int e = r;
            while (e < P.n_ops && P.ops[e].opcode == DSP_OP_STORE_SCALAR && e - r < DSP_IC && P.ops[e].member == P.ops[r].member) ++e;
            if (e - r >= 2) {
                DevOp m = P.ops[r];
                m.opcode = DSP_OP_INTERNAL_STORES;
                m.dst = e - r;
                for (int j = 0; j < e - r; ++j) m.ic[j] = P.ops[r + j].io | (P.ops[r + j].ip[0] << 16);
                P.ops[w++] = m;
                r = e;
                continue;
            }
            if (w != r) P.ops[w] = P.ops[r];
            ++w;
            ++r;
        }
        P.n_ops = w;
    }

    for (int i = 0; i < P.n_ops; ++i) P.ops[i].prio = (i * 4) / P.n_ops;
    for (int s = 0; s < n_slots; ++s) {
        ch->slot_base[s] = base[s];
        ch->slot_foot[s] = foot[s];
        ch->slot_first_op[s] = first_op[s];
        ch->slot_last_op[s] = last_op[s];
        ch->slot_shares[s] = shares[s] ? 1 : 0;
    }
    return DSP_OK;
}

const char* dsp_plan_kernel_name(const ChainPlan* ch) {
    if (ch && ch->scalar_ok && ch->fused_on) return dsp_internal_scalar_kernel_name();
    if (ch && ch->pz_ok && ch->fused_on) return dsp_internal_pz_rows_kernel_name();
    if (ch && ch->red_ok && ch->fused_on) return dsp_internal_reduce_kernel_name();
    if (ch && ch->runs_ok && ch->fused_on) return dsp_internal_fir_runs_kernel_name();
    if (ch && ch->cur_ok && ch->fused_on) return dsp_internal_current_kernel_name();
    if (ch && ch->fir_ok && ch->fused_on)
        return ch->fir_f16 ? dsp_internal_fir_f16_kernel_name() : (ch->fir.store ? dsp_internal_fir_store_kernel_name() : dsp_internal_fir_mfma_kernel_name());
    if (ch && ch->rows_ok && ch->fused_on) return dsp_internal_rows_kernel_name();
    if (ch && ch->rr_ok && ch->fused_on && ch->variant != 1) return dsp_internal_energy_rr_kernel_name();
    return (ch && ch->fused_ok && ch->fused_on) ? dsp_internal_energy_kernel_name() : dsp_internal_vm_kernel_name();
}

extern "C" int dsp_chain_plan(const dsp_op* ops, int n_ops, const dsp_io_desc* io, int n_io, const int32_t* slot_len, int n_slots, int n_sregs,
                              int compute_dtype, dsp_plan_info* info) {
    if (!info) return dsp_fail(DSP_ERR_ARG, "null info");
    memset(info, 0, sizeof *info);
    std::unique_ptr<ChainPlan> ch(new ChainPlan());
    const int rc = dsp_plan_build(ch.get(), ops, n_ops, io, n_io, slot_len, n_slots, n_sregs, compute_dtype);
    if (rc != DSP_OK) return rc;
    snprintf(info->kernel, sizeof info->kernel, "%s", dsp_plan_kernel_name(ch.get()));
    const bool specialised = ch->fused_on && (ch->scalar_ok || ch->pz_ok || ch->red_ok || ch->runs_ok || ch->cur_ok || ch->fir_ok || ch->rows_ok || ch->rr_ok || ch->fused_ok);
    snprintf(info->note, sizeof info->note, "%s", specialised ? "" : ch->note.c_str());
    const DevProgram& P = ch->host;
    info->lds_bytes_per_wave = ch->lds_bytes_per_wave;
    info->waves_per_block = ch->waves_per_block;
    info->team = P.team;
    info->n_device_ops = P.n_ops;
    info->lds_elems_per_wave = P.lds_elems_per_wave;
    info->sreg_off = P.sreg_off;
    info->scratch_off = P.scratch_off;
    info->n_slots = P.n_slots;
    for (int s = 0; s < P.n_slots; ++s) {
        info->slot_base[s] = ch->slot_base[s];
        info->slot_elems[s] = ch->slot_foot[s];
        info->slot_first_op[s] = ch->slot_first_op[s];
        info->slot_last_op[s] = ch->slot_last_op[s];
        info->slot_off[s] = P.slots[s].off;
        info->slot_pitch[s] = P.slots[s].pitch;
        info->slot_chunk[s] = P.slots[s].C;
    }
    return DSP_OK;
}
