// dsp_plan.h -- the planner's half of a chain: what dsp_chain_create decides about a dsp_op program WITHOUT touching the device.
//
// dsp_plan_build validates the program (every constant-only DSPFatal condition of the reference included), evaluates the constants the
// reference evaluates once per call, packs the waveform slots into LDS by lifetime, matches the program against the specialised kernels'
// shapes and picks the launch geometry.  It makes no HIP call and includes no HIP header: the same translation unit is compiled for the
// CPU with -fsanitize=address,undefined and fuzzed there (tests/test_planner_fuzz.py, tools/planner_fuzz.cpp).  dsp_host.cpp puts the
// device resources on top (struct dsp_chain : ChainPlan).  Internal header; the public contract is include/dspeed_hip.h.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>

#include "dsp_program.h"

constexpr int LDS_BYTES_PER_CU = 160 * 1024;

// mirror of the struct in dsp_energy.hip
struct EnergyArgs {
    const void* wf;
    int64_t wf_stride;
    int32_t wf_offset;
    int32_t len;
    const float* bl;
    int64_t bl_stride;
    float bl_const;
    int32_t has_bl;
    const float* tp;
    int64_t tp_stride;
    float tp_const;
    int32_t mode;
    float* out;
    int64_t out_stride;
    double c;
    double rr, ll;
    int32_t tau_nan;
    int32_t all_nan;
    int32_t C, pitch;
    float invC;
    int32_t q[3], rho[3];
    int32_t lds_elems_per_wave;
    int32_t slot_off;
    const float* tau;
    int64_t tau_stride;
    int32_t ablate;
};

struct EnergyPlan {
    int32_t shift[3][4];
    int32_t cs[3][4];
    int32_t local[3][4];
};

// geometry of the kernels, defined beside them (host arithmetic on their tile constants: dsp_current.hip, dsp_fir_mfma.hip, dsp_fir_f16.hip)
extern "C" int dsp_internal_current_lds_bytes(int ma_len);
extern "C" int dsp_internal_fir_mfma_lds_bytes(int kend);
extern "C" int dsp_internal_fir_store_lds_bytes(int kend);
extern "C" int dsp_internal_fir_f16_tz(int kend);
extern "C" size_t dsp_internal_fir_f16_taps_bytes(int kend);
extern "C" int dsp_internal_fir_f16_lds_bytes();
extern "C" const char* dsp_internal_vm_kernel_name();
extern "C" const char* dsp_internal_energy_kernel_name();
extern "C" const char* dsp_internal_energy_rr_kernel_name();
extern "C" const char* dsp_internal_rows_kernel_name();
extern "C" const char* dsp_internal_pz_rows_kernel_name();
extern "C" const char* dsp_internal_reduce_kernel_name();
extern "C" const char* dsp_internal_scalar_kernel_name();
extern "C" const char* dsp_internal_current_kernel_name();
extern "C" const char* dsp_internal_fir_f16_kernel_name();
extern "C" const char* dsp_internal_fir_mfma_kernel_name();
extern "C" const char* dsp_internal_fir_store_kernel_name();
extern "C" const char* dsp_internal_fir_runs_kernel_name();
extern "C" int dsp_internal_fir_runs_lds_bytes(int m);

// thread-local text behind dsp_last_error(); dsp_fail formats it and hands `code` back
int dsp_fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
void dsp_set_last_error(const char* text);
const char* dsp_plan_last_error();
int dsp_elem_size(int dtype);

// Everything the planner decides.  Plain data apart from the note; the pointers inside the kernels' argument blocks stay null here (the
// I/O pointers of a launch are filled in by dsp_chain_execute, the device images of the float16 FIR taps by dsp_chain_create).
struct ChainPlan {
    DevProgram host{};
    int lds_bytes_per_wave = 0;
    int waves_per_block = 0;   // generic VM launch
    int classic_wpb = 0;       // classic energy kernel launch (built for 2 wavefronts per SIMD)
    bool has_fir = false;      // the program holds a CONVOLVE: the VM build with the FIR op (2 wavefronts per SIMD instead of 4)
    bool f64 = false;  // the float64 gufunc loop (LDS elements are 8 bytes)
    bool i64 = false;  // an integer program of per-event values (compute_dtype DSP_I64): the row-per-lane kernel with 64-bit integer registers
    // specialised energy-chain kernel (dsp_energy.hip), selected when the program has exactly that shape
    bool fused_ok = false, fused_on = true;
    EnergyArgs fused{};
    int fused_trap = 0, fused_npf = 0, wf_dtype = DSP_F32;
    // register-resident kernel (pad-free LDS image): the default for 1024/2048/4096-sample energy chains
    bool rr_ok = false;
    // 6: register-resident kernel (default where it applies), 8: the same with two replay sub-chains per lane (A/B only),
    // 1: classic kernel (VM layout, bit-identical to the VM)
    int variant = 1;
    EnergyArgs rr{};
    EnergyPlan plan[2]{};  // [S - 1]
    int rr_lds_bytes = 0;
    int io_wf = -1, io_bl = -1, io_tp = -1, io_out = -1, io_tau = -1;
    // lane-per-waveform kernel (dsp_rows.hip): [bl_subtract ->] pole_zero | double_pole_zero -> short trapezoid -> min_max /
    // time_point_thresh, + Haar DWT of the pole-zero corrected waveform
    bool rows_ok = false;
    RowsArgs rows{};
    int rows_lds_bytes = 0;
    int rio_wf = -1, rio_bl = -1, rio_thr = -1, rio_ts = -1, rio_mm[4] = {-1, -1, -1, -1}, rio_tpt = -1, rio_dwt = -1;
    // matrix-core FIR kernel (dsp_fir_mfma.hip): LOAD [-> BL_SUBTRACT] -> CONVOLVE_AMAX ('v') x 1..4 -> STORE_SCALARs
    bool fir_ok = false;
    FirArgs fir{};
    int fir_lds_bytes = 0;
    int fio_wf = -1, fio_bl = -1, fio_taps[DSP_FIR_MAXK] = {-1, -1, -1, -1}, fio_out[DSP_FIR_MAXK] = {-1, -1, -1, -1};
    // the amax form on the float16 matrix instructions (dsp_fir_f16.hip): the default where the rows keep 16-byte alignment
    bool fir_f16 = false;
    FirF16Taps f16{};  // device images of the kernels' taps (rewritten by every launch: the taps are a binding), the rows' scales and flags
    // why a program that all but has the shape of a specialised kernel runs on the interpreter instead (dsp_chain_kernel_note)
    std::string note;
    // a program of scalar ops only (dsp_scalar.hip: a row per lane)
    bool scalar_ok = false;
    // lane-per-waveform current-branch kernel (dsp_current.hip)
    bool cur_ok = false;
    CurrentArgs cur{};
    int cur_lds_bytes = 0, cio_wf = -1, cio_t0 = -1, cio_out[4] = {-1, -1, -1, -1};
    // pole-zero rows written back as rows (dsp_pz.hip)
    bool pz_ok = false;
    PzArgs pz{};
    int pio_wf = -1, pio_bl = -1, pio_out = -1, pio_tau = -1, pio_mm[4] = {-1, -1, -1, -1};
    // streaming reductions of rows (dsp_reduce.hip)
    bool red_ok = false;
    ReduceArgs red{};
    int dio_wf = -1, dio_out[5] = {-1, -1, -1, -1, -1}, dio_pick[DSP_REDUCE_PICKS] = {-1, -1, -1, -1}, red_dtype = DSP_F32;
    int dio_walk[DSP_REDUCE_WALKS] = {-1, -1, -1, -1, -1, -1}, dio_walk_thr[DSP_REDUCE_WALKS] = {-1, -1, -1, -1, -1, -1},
        dio_walk_ts[DSP_REDUCE_WALKS] = {-1, -1, -1, -1, -1, -1};
    bool red_vec = false;  // rows keep 16-byte alignment and hold whole 16-byte vectors
    // run-length FIR with the reductions of its output (dsp_fir_runs.hip); the reductions' bindings are dio_* above
    bool runs_ok = false;
    FirRunsArgs runs{};
    int uio_wf = -1, uio_taps = -1, uio_out = -1;
    // LDS packing as decided (for dsp_chain_plan and the fuzzer's invariants): region of slot s = [slot_base[s], slot_base[s] + slot_foot[s])
    // elements of the compute type, alive from op slot_first_op[s] to slot_last_op[s] of the caller's program
    int32_t slot_base[DSP_MAX_SLOTS] = {0}, slot_foot[DSP_MAX_SLOTS] = {0}, slot_first_op[DSP_MAX_SLOTS] = {0}, slot_last_op[DSP_MAX_SLOTS] = {0};
    uint8_t slot_shares[DSP_MAX_SLOTS] = {0};
};

// 0 or the DSP_ERR_* / DSP_E_* code (text in dsp_plan_last_error()); `ch` must be a freshly constructed plan
int dsp_plan_build(ChainPlan* ch, const dsp_op* ops, int n_ops, const dsp_io_desc* io, int n_io, const int32_t* slot_len, int n_slots,
                   int n_sregs, int compute_dtype);
// the kernel a planned chain launches on rows that keep 16-byte alignment (what rocprofv3 --kernel-trace lists)
const char* dsp_plan_kernel_name(const ChainPlan* ch);
