// dsp_program.h -- device-side program representation shared by the host translator (dsp_host.cpp)
// and the waveform VM kernel (dsp_vm.hip).  Internal: the public contract is include/dspeed_hip.h.
#pragma once
#include <stdint.h>

#include "../../include/dspeed_hip.h"

#define DSP_WAVE 64
#define DSP_FC 40 /* host-precomputed float64 constants per op */
#define DSP_OP_INTERNAL_NOP 101  /* host-made: a BL_SUBTRACT that the LOAD in front of it does while it writes the samples (DevOp ic[0] of the LOAD) */
#define DSP_OP_INTERNAL_STORES 102 /* host-made: a run of STORE_SCALAR ops as one op -- dst = count (<= DSP_IC), ic[j] = binding | register << 16: lane j stores */
#define DSP_OP_INTERNAL_ZERO 100 /* host-inserted: clear slot dst's whole LDS region (guard, chunks, pads, tail) before its first use */
#define DSP_SCRATCH_ELEMS 128
#define DSP_IC 12 /* host-precomputed integer constants per op */

// One waveform variable living in LDS.  Lane j of the wavefront owns samples [j*C, (j+1)*C) ("chunk");
// chunk j starts at element off + j*pitch with pitch = C+1 (odd) so that "every lane reads offset t of
// its chunk" hits 64 different banks.  Samples >= len up to 64*C are kept finite (zero-filled at load).
// FIR inputs are laid out without the pad (padw = 0, pitch = C: sample i at element off + i) so that a lane's
// window of consecutive samples is addressed with immediate offsets.
struct DevSlot {
    int32_t off;   // element offset inside the wavefront's LDS region
    int32_t len;   // logical number of samples
    int32_t C;     // samples per lane, multiple of 8
    int32_t pitch; // C + 1
    float invC;    // 1/C for index -> (lane, offset) splits
    int32_t padw;  // pitch - C: 1 (chunk pad) or 0 (linear)
    // elements that are guaranteed to read 0 below sample 0 (the guard) and above sample len - 1 (the tail; 0 unless len == 64 * C:
    // a partial last chunk is only "finite"): a FIR window may reach that far outside the waveform without bounds checks
    int32_t zero_below, zero_above;
};

struct DevIO {
    int32_t kind, dtype, len, offset;
    int64_t row_stride;
    int32_t vec_ok; // row base, stride and offset keep 16-byte alignment -> wide loads/stores
    int32_t pad_;
};

#define DSP_MEMBER_ALL 15
struct DevOp {
    int32_t opcode, dst, src, io;
    int32_t ip[4];
    dsp_scalar_arg sp[4];
    int32_t ic[DSP_IC];
    int32_t member;  // DevProgram.team > 1: which wavefront of a row's team runs the op (0 .. team - 1; DSP_MEMBER_ALL: every member)
    int32_t prio;    // the wave priority the interpreter runs the op at: 0 .. 3 over the op list (the planner's division, not one per op and row)
    double fc[DSP_FC];
};

struct DevProgram {
    int32_t n_ops, n_slots, n_io, n_sregs;
    int32_t lds_elems_per_wave; // waveform slots + scalar registers, in elements of the compute type
    int32_t sreg_off;           // element offset of the scalar register file
    int32_t waves_per_block;
    int32_t scratch_off;        // element offset of DSP_SCRATCH_ELEMS elements any op may use while it runs (16-byte aligned)
    // 2, 3: a TEAM of wavefronts per row -- a program that loads one waveform and then only reads it (reductions, walks, pick-offs of long
    // waveforms, whose image leaves LDS for one wavefront per SIMD) splits into groups of ops that share no register; each member runs
    // its groups on the shared image (DevOp.member).  1: a wavefront per row
    int32_t team;
    int32_t pad_;
    // per-op cycle counters (dsp_chain_profile): n_ops + 1 device words, the last counts the waveforms sampled; null = off
    unsigned long long* prof;
    DevSlot slots[DSP_MAX_SLOTS];
    DevIO io[DSP_MAX_IO];
    DevOp ops[DSP_MAX_OPS + DSP_MAX_SLOTS]; // + one region-clearing op per slot that shares LDS (DSP_OP_INTERNAL_ZERO)
};

struct IoPtrs {
    void* p[DSP_MAX_IO];
};

// device error word layout: err[0] = DSP_E_* code (0 = none), err[1..2] = row (lo, hi)
#define DSP_ERR_WORDS 20 /* 4 error words + 6 x 64-bit diagnostic phase-cycle sums */

// arguments of the lane-per-waveform fit kernel (dsp_fit.hip), filled by dsp_linear_slope_fit_rows
struct FitArgs {
    const void* wf;
    int64_t n_wf, row_stride;
    int32_t wf_len, n_scan;  // samples per row; samples that have to be visited (the NaN rules of bl_subtract / pole_zero look at all)
    const void* sub;         // per-row value subtracted first (device column) or null: sub_const
    int32_t sub_dtype, sub_mode;  // 0 nothing, 1 bl_subtract (NaN anywhere -> NaN waveform), 2 numpy.subtract (sample by sample)
    double sub_const;
    int32_t has_pz, pz_nan;
    double pz_c;
    int32_t n_fits;
    int32_t stage[DSP_FIT_MAX], first[DSP_FIT_MAX], count[DSP_FIT_MAX];
    void* out;  // [n_fits][4][n_wf] of the compute type: mean, stdev, slope, intercept
};


// arguments of the lane-per-waveform chain kernel (dsp_rows.hip), filled by dsp_chain_execute when a program has that shape
struct RowsArgs {
    const void* wf;          // rows
    int64_t wf_stride;       // elements between rows
    int32_t wf_offset, len;  // first sample used, samples per waveform (a multiple of 8)
    int32_t in_kind;         // 0 float32, 1 int16, 2 uint16 rows
    int32_t sub_mode;        // 1: bl_subtract first
    const float* bl;         // per-row baseline column or null: bl_const
    int64_t bl_stride;
    float bl_const;
    int32_t pz_kind;         // 1 pole_zero, 2 double_pole_zero
    int32_t pz_param_nan;    // a NaN time constant: everything downstream is NaN
    double pz_c;             // pole_zero: exp(-1/tau)
    double n1, n2, d1, d2;   // double_pole_zero: numerator / denominator coefficients (pole_zero.py:168-174)
    int32_t trap_kind;       // 0 trap_filter, 1 trap_norm, 2 asym_trap_filter
    int32_t rise_pow2;       // rise is a power of two: x / rise == x * (1 / rise) exactly
    int32_t lag[3];
    int32_t trap_all_nan;    // trap_filter with rise == 0
    double rr, ll, inv_rr, inv_ll;
    void* out_mm[4];         // t_min, t_max, a_min, a_max columns (null: not requested)
    int64_t out_mm_stride[4];
    int32_t tpt_mode;        // 0 none, 1 backward from a known start, 2 backward from the arg-extremum, 3 / 4 the same walking forward
    int32_t tpt_use_min;     // the start is t_min (else t_max)
    const float* thr;        // threshold column or null: thr_const
    int64_t thr_stride;
    const float* ts;         // start column or null: ts_const (modes 1, 3)
    int64_t ts_stride;
    float thr_const, ts_const;
    int32_t walk_nan, walk_frac;  // walk_forward is NaN (output NaN) / not an integer (DSPFatal)
    void* out_tpt;
    int64_t out_tpt_stride;
    int32_t dwt_level, dwt_part;  // Haar level (0: none, else 3..8), 'a' or 'd'
    void* dwt_out;
    int64_t dwt_stride;
    int32_t ring_entries;    // R: history samples kept per lane, a multiple of 8, >= largest lag + 16
    // 1: nothing behind the start of a backward walk matters -- the only output is a walk backward from a known start and the rows are
    // NaN-free or NaN from the first sample on (DSP_OP_LOAD ip[2]): the group stops at the block behind its 64 rows' latest start
    int32_t stop_at_start;
};

// arguments of the lane-per-waveform current-branch kernel (dsp_current.hip), filled by dsp_chain_execute when a program has that shape:
//   LOAD -> WINDOWER -> AVG_CURRENT -> UPSAMPLER -> MOVING_WINDOW_MULTI (3 alternating windows) -> MIN_MAX -> STORE_SCALARs
struct CurrentArgs {
    const void* wf;          // float32 rows
    int64_t wf_stride;
    int32_t wf_offset, n_in; // first sample, samples per row (a multiple of 4)
    const float* t0;         // window start column or null: t0_const
    int64_t t0_stride;
    float t0_const;
    int32_t win_len;         // samples of the window
    int32_t ac_lag;          // avg_current: int(length)
    float ac_length;
    int32_t n_c;             // samples of the current waveform: win_len - ac_lag
    int32_t up_shift, up_half;  // upsampling factor 1 << up_shift, floor(factor / 2)
    int32_t n_up;            // samples of the upsampled waveform (a multiple of 16)
    int32_t ma_len;          // moving-window length (a multiple of 16, at most 112)
    float ma_length;
    int32_t scan_rows;       // 1: the kernel screens the whole rows for NaN; 0: a NaN anywhere means NaN everywhere (DSP_OP_LOAD ip[2])
    void* out[4];            // t_min, t_max, a_min, a_max columns (null: not requested)
    int64_t out_stride[4];
    float* scratch;          // scratch_per_wave floats per resident wavefront: the current waveform and the checkpoints of two passes
    int64_t scratch_per_wave;
};

// arguments of the pole-zero rows kernel (dsp_pz.hip):  LOAD -> [BL_SUBTRACT] -> POLE_ZERO (constant tau) -> STORE
struct PzArgs {
    const void* wf;          // float32 / int16 / uint16 rows, 16-byte aligned
    int64_t wf_stride;
    int32_t wf_offset, len;  // first sample, samples (a multiple of 8)
    int32_t in_kind;         // 0 float32, 1 int16, 2 uint16
    int32_t sub_mode;        // 1: bl_subtract first
    const float* bl;         // per-row baseline column or null: bl_const
    int64_t bl_stride;
    float bl_const;
    int32_t tau_nan;
    double c;                // exp(-1/tau)
    void* out;               // float32 rows
    int64_t out_stride;
    const float* tau;        // or null: the time constant per event (a float32 column) instead of c / tau_nan
    int64_t tau_stride;
    float* row_scale;        // or null: what dsp_fir_f16_rows_kernel would find on the rows written here (FirF16Taps), for a float16 FIR behind
    uint32_t* row_flags;
    void* mm_out[4];         // or null each: min_max (min_max.py:11-82) of the rows as they are READ (t_min, t_max, a_min, a_max; float32 columns)
    int64_t mm_stride[4];
    int32_t mm_on, pad_;
    // or null: what dsp_fir_f16_rows_kernel would find on samples [in_lo, in_hi) of the rows READ here, minus the baseline -- for a float16 FIR
    // that filters that slice of the same rows with the same baseline (the cusp filter of the Ge recipes: waveform[0:6092] - baseline)
    float* in_scale;
    uint32_t* in_flags;
    int32_t in_lo, in_hi;
};

// arguments of the streaming reductions (dsp_reduce.hip), filled by dsp_chain_execute when a program has the shape
//   LOAD -> {MIN_MAX | AMAX | PICKOFF at a constant integral time | TIME_POINT_THRESH from a constant sample or from the extremes}+ -> STORE_SCALARs
#define DSP_REDUCE_PICKS 4
#define DSP_REDUCE_WALKS 6
struct ReduceArgs {
    const void* wf;          // float32 / int16 / uint16 rows
    int64_t wf_stride;
    int32_t wf_offset, len;  // first sample, samples
    void* out[5];            // t_min, t_max, a_min, a_max of MIN_MAX, the maximum of AMAX (null: not requested); float32 columns
    int64_t out_stride[5];
    void* pick_out[DSP_REDUCE_PICKS];
    int64_t pick_stride[DSP_REDUCE_PICKS];
    int32_t pick_at[DSP_REDUCE_PICKS];    // sample index, -1: outside the waveform (NaN)
    int32_t pick_rule[DSP_REDUCE_PICKS];  // 1: fixed_time_pickoff (a NaN anywhere in the row -> NaN), 0: the plain sample
    // time_point_thresh walks that start at a constant sample or at the minimum / maximum found above
    void* walk_out[DSP_REDUCE_WALKS];
    int64_t walk_stride[DSP_REDUCE_WALKS];
    const float* walk_thr[DSP_REDUCE_WALKS];  // threshold column, or null: walk_thr_const
    int64_t walk_thr_stride[DSP_REDUCE_WALKS];
    float walk_thr_const[DSP_REDUCE_WALKS];
    int32_t walk_from[DSP_REDUCE_WALKS];      // 0: walk_start, 1: t_min, 2: t_max, 3: the column walk_ts, 4 + j: where walk j (an earlier one) ended
    int32_t walk_start[DSP_REDUCE_WALKS];
    int32_t walk_forward[DSP_REDUCE_WALKS];
    // round 4: the rise-time walks of a recipe as a launch of their own behind its program (thousands of rows in flight instead of four a CU) --
    // thresholds that are a fraction of a per-event value (the column times walk_thr_factor: one float multiplication, as SCALAR_AFFINE makes it),
    // starts that are a column or where an earlier walk ended (checked like the processor's: DSPFatal for a fractional or outside start)
    float walk_thr_factor[DSP_REDUCE_WALKS];
    int32_t walk_thr_scaled[DSP_REDUCE_WALKS];
    const float* walk_ts[DSP_REDUCE_WALKS];
    int64_t walk_ts_stride[DSP_REDUCE_WALKS];
    int32_t n_walks;
    int32_t need_stream;  // 0: nothing asks for the whole row (walks only, on rows that are NaN from the first sample on or NaN-free: LOAD ip[2])
};

// arguments of the matrix-core FIR kernel (dsp_fir_mfma.hip): convolve_wf 'v' + numpy.amax of up to DSP_FIR_MAXK kernels on one waveform
#define DSP_FIR_MAXK 4
struct FirArgs {
    const void* wf;
    int64_t wf_stride;       // elements between rows
    int32_t wf_offset, n;    // first sample, samples of the slice the kernels run over
    int32_t in_kind;         // 0 float32, 1 int16, 2 uint16 rows
    int32_t sub_mode;        // 1: bl_subtract first
    const float* bl;
    int64_t bl_stride;
    float bl_const;
    int32_t n_kernels;
    const float* taps[DSP_FIR_MAXK];  // the kernels as the recipe holds them (np.convolve flips them)
    int32_t m[DSP_FIR_MAXK], p[DSP_FIR_MAXK];  // taps, valid outputs n - m + 1
    void* out[DSP_FIR_MAXK];
    int64_t out_stride[DSP_FIR_MAXK];
    int32_t kend;            // samples the product runs over: a multiple of 32, <= the row's length
    int32_t scan_before, scan_after;  // samples before / after the slice that are screened for NaN (DSP_OP_LOAD ip[0..1])
    int32_t store;           // 1: ONE kernel whose p[0] outputs are written to out[0] as a waveform (dsp_fir_store_kernel), any mode
    int32_t dshift;          // store: output c is the sum over samples c - dshift .. c - dshift + m - 1 ('v' 0, 's' m / 2, 'f' m - 1)
    const uint32_t* row_flags;  // dsp_fir_fixup_kernel: the rows' flags when something has looked at the rows already (dsp_fir_f16.hip), else null
};

// the float16 tap images of dsp_fir_f16.hip (one per kernel: 16 shifted / split copies of tz halfs, then the inverse scale), chain-owned
struct FirF16Taps {
    const void* taps16[DSP_FIR_MAXK];
    int32_t tz;
    const void* row_scale;  // float per row: the power of two that brings the row's largest magnitude into [2^14, 2^15)
    const void* row_flags;  // uint32 per row: bit 0 an infinity, bit 1 a NaN
    int32_t rows_done;      // the kernel that wrote the rows left both already (dsp_pz.hip): no pass over the rows for them
    int32_t pad_;
};


// the run-length FIR of dsp_fir_runs.hip: a kernel that is piecewise constant (the t0 filter of the Ge recipes: a ramp of 8 taps and a
// plateau of 125) written as  sum_b weight[b] * P[c + start + 1 - t[b]]  over its breakpoints t[0] = 0 < t[1] < .. < t[n_break - 1] = m,
// P the float64 prefix sums of the row.  Written by dsp_fir_runs_prep_kernel from the taps binding ahead of every launch.
#define DSP_FIR_RUNS_MAX 24      /* runs a kernel may have (breakpoints: one more) */
#define DSP_FIR_RUNS_MAX_TAPS 512
#define DSP_FIR_RUNS_GROUP 8     /* breakpoints at consecutive taps are taken together, up to this many */
struct FirRunsTable {
    int32_t n_break;   // 0: the taps are not of this form after all (more runs, a NaN or an infinity among them): every row tap by tap
    int32_t taps_nan;  // a NaN among the taps: every output NaN (convolutions.py:45-46)
    int32_t t[DSP_FIR_RUNS_MAX + 2];
    double weight[DSP_FIR_RUNS_MAX + 2];
    // breakpoints at consecutive taps (a ramp: every tap differs from the one before) as groups: first[k] .. first[k] + count[k] - 1
    int32_t n_groups, pad_;
    int32_t first[DSP_FIR_RUNS_MAX + 2], count[DSP_FIR_RUNS_MAX + 2];
};

// arguments of dsp_fir_runs_kernel:  LOAD -> CONVOLVE (any mode, ip[2] = 1) -> [STORE] -> the reductions of ReduceArgs on the filtered waveform
struct FirRunsArgs {
    const float* wf;         // float32 rows, 16-byte aligned
    int64_t wf_stride;
    int32_t wf_offset, n;    // first sample, samples (a multiple of 8)
    const float* taps;       // the kernel as the recipe holds it
    int32_t m, p;            // taps, outputs
    int32_t start;           // output c is np.convolve's full output c + start ('f' 0, 's' (m - 1) / 2, 'v' m - 1)
    int32_t keep;            // 1: the filtered waveform is an output (out), 0: it lives in the wavefront's scratch row until the reductions are done
    float* out;              // float32 rows (keep) or the scratch area: p_pitch floats per resident wavefront
    int64_t out_stride;      // keep: elements between rows; else p_pitch
    const FirRunsTable* table;
    int32_t has_red, pad_;
    ReduceArgs red;          // (wf / wf_stride / wf_offset unused: the reductions read the filtered row)
};
