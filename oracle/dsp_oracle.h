/* dsp_oracle.h -- CPU oracle for the dspeed hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference's numba kernel bodies (legend-exp/dspeed,
 * src/dspeed/processors/<name>.py; every function in dsp_oracle_impl.h cites file:line).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
 * and only as the checker / the reported CPU baseline -- the product (dspeed_amd + libdspeed_hip.so)
 * never links, imports or calls it.
 *
 * Pinning: checked against (1) the reference's own known-answer tests restated in
 * tests/test_oracle_golden.py (tests/processors/test_pole_zero.py, test_fixed_time_pickoff.py,
 * test_time_point_thresh.py, test_dwt.py of the reference) and (2) the fixtures in tests/golden/ (npz files),
 * which oracle/gen_golden.py produced by executing the reference's kernel bodies in the build container.
 */
#ifndef DSP_ORACLE_H
#define DSP_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* DSPFatal conditions; numeric values are shared with include/dspeed_hip.h (DSP_E_*) */
enum {
    ORC_OK = 0,
    ORC_E_PZ_NAN = 1,        /* pole_zero.py:76-77 */
    ORC_E_DPZ_SHORT = 2,     /* pole_zero.py:163-166 */
    ORC_E_TRAP_RISE = 3,     /* trap_filters.py:53-54 */
    ORC_E_TRAP_FLAT = 4,     /* trap_filters.py:56-57 */
    ORC_E_TRAP_FALL = 5,     /* trap_filters.py:205-206 */
    ORC_E_TRAP_WIDE = 6,     /* trap_filters.py:59-60 */
    ORC_E_FTP_INT = 7,       /* fixed_time_pickoff.py:84-85 */
    ORC_E_FTP_MODE = 8,      /* fixed_time_pickoff.py:124-125 */
    ORC_E_TPT_START_INT = 9, /* time_point_thresh.py:67-68 */
    ORC_E_TPT_WALK_INT = 10, /* time_point_thresh.py:70-71 */
    ORC_E_TPT_RANGE = 11,    /* time_point_thresh.py:73-74 */
    ORC_E_CONV_LONG = 12,    /* convolutions.py:48-49 */
    ORC_E_CONV_OUTLEN = 13,  /* convolutions.py:52-67 */
    ORC_E_CONV_MODE = 14,    /* convolutions.py:69-70 */
    ORC_E_DWT_LEVEL = 15,    /* dwt.py:67-68 */
    ORC_E_DWT_OUTLEN = 16,   /* numpy broadcast error in dwt.py:81 */
    ORC_E_ZERODIV = 17,      /* numba error_model='python': x / 0 raises ZeroDivisionError */
    ORC_E_WINDOW_LONG = 18,  /* windower.py:36-37 */
    ORC_E_AVGCUR_RANGE = 19, /* moving_windows.py:243-246 */
    ORC_E_TPO_INT = 20,      /* trap_filters.py:270-271 */
    ORC_E_UPSAMPLE = 21,     /* upsampler.py:47-48 */
    ORC_E_MW_LEN_INT = 22,   /* moving_windows.py:167-168 */
    ORC_E_MW_NUM_INT = 23,   /* moving_windows.py:170-171 */
    ORC_E_MW_LEN_RANGE = 24, /* moving_windows.py:173-174 */
    ORC_E_MW_NUM_NEG = 25    /* moving_windows.py:176-177 */
};

#define ORC_DECL(T, S)                                                                                                         \
    int orc_bl_subtract_##S(const T* in, long n_wf, int len, const T* bl, int bl_stride, T* out, long* err_row);              \
    int orc_pole_zero_##S(const T* in, long n_wf, int len, T tau, T* out, long* err_row);                                     \
    int orc_double_pole_zero_##S(const T* in, long n_wf, int len, T tau1, T tau2, T frac, T* out, long* err_row);             \
    int orc_trap_filter_##S(const T* in, long n_wf, int len, int rise, int flat, T* out, long* err_row);                      \
    int orc_trap_norm_##S(const T* in, long n_wf, int len, int rise, int flat, T* out, long* err_row);                        \
    int orc_asym_trap_filter_##S(const T* in, long n_wf, int len, int rise, int flat, int fall, T* out, long* err_row);       \
    int orc_fixed_time_pickoff_##S(const T* in, long n_wf, int len, const T* t_in, int t_in_stride, int mode, T* out,         \
                                   long* err_row);                                                                            \
    int orc_time_point_thresh_##S(const T* in, long n_wf, int len, const T* thr, int thr_stride, const T* t_start,            \
                                  int t_start_stride, T walk_forward, T* out, long* err_row);                                 \
    int orc_interpolated_time_point_thresh_##S(const T* in, long n_wf, int len, const T* thr, int thr_stride, const T* t_start,   \
                                               int t_start_stride, long walk_forward, int mode, T* out, long* err_row);       \
    int orc_min_max_##S(const T* in, long n_wf, int len, T* t_min, T* t_max, T* a_min, T* a_max, long* err_row);              \
    int orc_min_max_norm_##S(const T* in, long n_wf, int len, const T* a_min, int a_min_stride, const T* a_max, int a_max_stride,   \
                             T* out, long* err_row);                                                                          \
    int orc_windower_##S(const T* in, long n_wf, int len, const T* t0, int t0_stride, T* out, int m, long* err_row);          \
    int orc_avg_current_##S(const T* in, long n_wf, int len, T length, T* out, int m, long* err_row);                         \
    int orc_trap_pickoff_##S(const T* in, long n_wf, int len, int rise, int flat, const T* tp, int tp_stride, T* out,         \
                             long* err_row);                                                                                  \
    int orc_upsampler_##S(const T* in, long n_wf, int len, T upsample, T* out, int m, long* err_row);                         \
    int orc_moving_window_multi_##S(const T* in, long n_wf, int len, T length, T num_mw, int mw_type, T* out, long* err_row); \
    int orc_linear_slope_fit_##S(const T* in, long n_wf, int len, T* mean, T* stdev, T* slope, T* intercept, long* err_row);  \
    int orc_mean_below_threshold_##S(const T* in, long n_wf, int len, const T* thr, int thr_stride, T* out, long* err_row);   \
    int orc_convolve_##S(const T* in, long n_wf, int len, long in_row_stride, const T* kern, int m, int mode, T* out, int p,  \
                         long* err_row);                                                                                      \
    int orc_dwt_haar_##S(const T* in, long n_wf, int len, int level, int part, T* out, int p, long* err_row);

ORC_DECL(float, f32)
ORC_DECL(double, f64)
#undef ORC_DECL

/* Ge energy chain of BASELINE.json config 2, run the way the reference's ProcessingChain runs it
 * (processing_chain.py:665-673, 1144-1163): blocks of `block_width` rows, one processor call per block,
 * every intermediate waveform materialised in a (block_width, len) scratch buffer.
 * n_threads <= 1: single thread (dspeed as shipped); > 1: OpenMP over blocks.
 * Returns the first DSPFatal code met (0 if none). */
int orc_chain_energy_f32(const float* wf, long n_wf, int len, const float* baseline, const float* t_pick, float tau, int rise,
                         int flat, int mode, float* e_out, int block_width, int n_threads);

/* C1: pole_zero -> trap_filter (plumbing config) */
int orc_chain_pz_trap_f32(const float* wf, long n_wf, int len, float tau, int rise, int flat, float* out, int block_width,
                          int n_threads);

int orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
